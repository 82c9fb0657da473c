/*
 * stemgnn.h — C ABI of the MI355X (gfx950) STEM-GNN hot-path library (libstemgnn_hip.so).
 *
 * Drop-in boundary for the encoder message-passing aggregation + vector-quantize
 * pretraining path of GXG-CS/STEM-GNN.  The reference has no FFI of its own: the
 * boundary it exposes is the Python nn.Module surface, and the device
 * operations behind it are whatever PyTorch / PyG / torch-scatter dispatch to.  Each entry
 * point below names the reference operation (file:line, relative to
 * /root/reference/STEM-GNN) whose device work it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (hipMalloc / PyTorch caching allocator) unless the
 *    parameter name ends in _host; matrices are dense row-major fp32; indices as stated.
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing
 *    synchronises, allocates or frees (graph-capture safe).  Scratch comes from the caller
 *    (`workspace`, sized by the matching *_workspace_bytes function).
 *  - return value: STEMGNN_OK (0) or a negative STEMGNN_ERR_* code.  Shape / range
 *    violations that can be seen on the host are rejected before any launch.
 *  - inputs are borrowed and never written; outputs are fully overwritten unless documented
 *    as in-place.
 */
#ifndef STEMGNN_H_
#define STEMGNN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STEMGNN_ABI_VERSION 1

/* element kinds of operands that may be stored as bf16 (the *_k entry points); values widen exactly at the load and
 * are rounded to nearest-even at the store, arithmetic is fp32 */
enum { STEMGNN_F32 = 0, STEMGNN_BF16 = 1 };

enum {
  STEMGNN_OK = 0,
  STEMGNN_ERR_INVALID_ARG = -1, /* null pointer, negative size, unsupported D / K / Dc */
  STEMGNN_ERR_TOO_LARGE = -2,   /* N or E >= 2^31 - 1 (int32 CSR) */
  STEMGNN_ERR_WORKSPACE = -3,   /* workspace too small */
  STEMGNN_ERR_HIP = -4          /* a HIP runtime call failed; see stemgnn_last_hip_error() */
};

int stemgnn_abi_version(void);
const char* stemgnn_status_string(int status);
/* hipError_t (as int) of the last failing HIP call on the calling thread, 0 if none. */
int stemgnn_last_hip_error(void);

/* ------------------------------------------------------------------------------------
 * Graph structure: int64 COO -> int32 CSR.
 * Replaces the implicit per-call index handling of PyG's MessagePassing.propagate
 * (model/encoder.py:82: index_select over edge_index[0], scatter over edge_index[1]).
 * ------------------------------------------------------------------------------------ */

/* Scratch bytes needed by stemgnn_csr_build for this problem size. */
size_t stemgnn_csr_workspace_bytes(int64_t num_nodes, int64_t num_edges);

/*
 * Stable counting/radix sort of the edges by one endpoint.
 *   edge_index [2, E] int64 row-major, row 0 = source, row 1 = target (PyG convention).
 *   key_row    1: group by target  (CSR used by the forward aggregation, K1)
 *              0: group by source  (transposed CSR used by the backward, K2)
 *   rowptr [N+1]  segment offsets;  other [E] the opposite endpoint of each slot;
 *   eid [E]       original edge position of each slot (stable: ascending inside a segment).
 *   bad_count [1] number of edges with an endpoint outside [0, N); such edges are left
 *                 out (rowptr[N] = E - bad) so no later kernel can index out of range.
 */
int stemgnn_csr_build(const int64_t* edge_index, int64_t num_edges, int64_t num_nodes, int key_row,
                      int32_t* rowptr, int32_t* other, int32_t* eid, int32_t* bad_count,
                      void* workspace, size_t workspace_bytes, void* stream);

/*
 * dropout_adj(..., force_undirected=True) (reference pretrain.py:42-44, PyG 2.3.0) applied to a
 * graph given by BOTH of its CSR views, producing both CSR views of the augmented graph with
 * no host synchronisation.  keep(e) = philox(seed, offset)[e] >= p (the mask of
 * stemgnn_dropout_keep_mask(E, p, seed, offset)) or keep[e] != 0 when `keep` is given; an edge
 * survives when keep(e) && src(e) <= dst(e) and is emitted in both directions.  Output arrays
 * hold up to 2E slots (a_rowptr[N] = number of augmented edges); slots carry the ORIGINAL edge
 * id so the original dense edge_attr / edge types address them directly.
 */
size_t stemgnn_graph_dropout_workspace_bytes(int64_t num_nodes);
int stemgnn_graph_dropout_undirected(const int32_t* rowptr, const int32_t* src, const int32_t* eid,
                                     const int32_t* etype_slot, const int32_t* rowptr_t, const int32_t* dst_t,
                                     const int32_t* eid_t, const int32_t* etype_slot_t, int64_t num_nodes,
                                     int64_t num_edges, float p, uint64_t seed, uint64_t offset, const uint8_t* keep,
                                     int32_t* a_rowptr, int32_t* a_src, int32_t* a_eid, int32_t* a_etype_slot,
                                     int32_t* a_dst_t, int32_t* a_eid_t, int32_t* a_etype_slot_t, float* a_inv_deg,
                                     void* workspace, size_t workspace_bytes, void* stream);
/* The same for a graph whose rows >= active_rows have no in-edges (a neighbour-sampled batch: only the expanded nodes,
 * which come first, receive edges).  A surviving edge has source <= target < active_rows, so only rows [0, active_rows)
 * are examined; the other rows of the augmented views are empty.  active_rows < 0 or > num_nodes: every row.  Two
 * launches (count + block totals scanned by the last block to finish; offsets + fill). */
int stemgnn_graph_dropout_undirected_rows(const int32_t* rowptr, const int32_t* src, const int32_t* eid,
                                          const int32_t* etype_slot, const int32_t* rowptr_t, const int32_t* dst_t,
                                          const int32_t* eid_t, const int32_t* etype_slot_t, int64_t num_nodes,
                                          int64_t num_edges, int64_t active_rows, float p, uint64_t seed,
                                          uint64_t offset, const uint8_t* keep, int32_t* a_rowptr, int32_t* a_src,
                                          int32_t* a_eid, int32_t* a_etype_slot, int32_t* a_dst_t, int32_t* a_eid_t,
                                          int32_t* a_etype_slot_t, float* a_inv_deg, void* workspace,
                                          size_t workspace_bytes, void* stream);

/* randperm(n)[:k] as the reference uses it (model/pt_model.py:55-57,75-78, model/vq.py:1024): k
 * distinct ids of [0, n), the first k outputs of a keyed pseudo-random permutation (Feistel
 * network + cycle walking); no sort.  out int64 [k]. */
int stemgnn_sample_subset(int64_t n, int64_t k, uint64_t seed, uint64_t offset, int64_t* out, void* stream);

/* The k-subset of stemgnn_sample_subset (same picks for the same n = num_edges, seed, offset) applied to an edge
 * list in the launch that draws it (reference model/pt_model.py:55-57,75-80): perm[i] = the pick,
 * sel_index[0 or sel_stride + i] = edge_index[:, perm[i]] (sel_stride >= k lets the caller place the picks in the
 * first k columns of a wider [2, *] buffer), sel_type[i] = edge_type[perm[i]], selected[perm[i]] = 1 after the call
 * has zeroed selected[0, num_edges).  sel_index, sel_type (+ edge_type) and selected may be NULL.
 * edge_index int64 [2, num_edges] contiguous. */
int stemgnn_sample_edges(const int64_t* edge_index, const int64_t* edge_type, int64_t num_edges, int64_t k,
                         uint64_t seed, uint64_t offset, int64_t* perm, int64_t* sel_index, int64_t sel_stride,
                         int64_t* sel_type, uint8_t* selected, void* stream);

/* mask_feature(x, p, mode='col') (reference pretrain.py:41): zero the feature columns whose
 * Philox draw is < p (keep mask = stemgnn_dropout_keep_mask(D, p, seed, offset)). */
int stemgnn_mask_columns(const float* x, int64_t num_rows, int64_t dim, float p, uint64_t seed, uint64_t offset,
                         float* out, void* stream);

/* negative_sampling(pos_edge_index, N) (reference model/pt_model.py:60; PyG structured sparse
 * sampling): k pairs (r, c), r != c, uniform over the N(N-1) non-loop pairs, rejecting pairs
 * that are SELECTED positive edges (selected[e] != 0 over the graph's edges, looked up through
 * the by-target CSR).  out int64 [2, k]. */
int stemgnn_negative_sample(const int32_t* rowptr, const int32_t* src, const int32_t* eid, const uint8_t* selected,
                            int64_t num_nodes, int64_t k, uint64_t seed, uint64_t offset, int64_t* out, void* stream);
/* The same with the two output rows out_stride (>= k) elements apart (writes into a slice of a wider buffer). */
int stemgnn_negative_sample_into(const int32_t* rowptr, const int32_t* src, const int32_t* eid,
                                 const uint8_t* selected, int64_t num_nodes, int64_t k, uint64_t seed,
                                 uint64_t offset, int64_t* out, int64_t out_stride, void* stream);

/* ------------------------------------------------------------------------------------
 * Mini-batch neighbour sampler (replaces NeighborLoader(num_neighbors=[f]*L) on the host,
 * reference pretrain.py:151-153).  The full graph is given by its by-target CSR (int32, resident
 * in HBM; etype = edge type per CSR slot, may be NULL); `local_of` [num_nodes] is a persistent
 * scratch map initialised once by stemgnn_sampler_init_map and left clean by every call.
 * Per hop each newly reached node draws min(deg, fanout) in-neighbours uniformly without
 * replacement (Philox keyed by seed/offset); seeds come first in the local numbering, then new
 * nodes hop by hop in order of first appearance.  Outputs: n_id [cap_nodes] (global id per
 * local node), the batch's by-target CSR b_rowptr [cap_nodes+1] / b_src / b_type [cap_edges],
 * its COO b_coo int64 [2, cap_edges] (row stride cap_edges; edge j == CSR slot j) and
 * counts[3] = (N_b, E_b, A_b) on the device, A_b = the number of leading local nodes that were expanded (only
 * those can have in-edges: rows >= A_b of the batch CSR are empty).  cap_nodes >= B(1 + f + .. + f^L),
 * cap_edges >= B(f + .. + f^L).
 * ------------------------------------------------------------------------------------ */
int stemgnn_sampler_init_map(int32_t* local_of, int64_t num_nodes, void* stream);
size_t stemgnn_sampler_workspace_bytes(int64_t batch_size, int64_t hops, int64_t max_fanout);
int stemgnn_sample_batch(const int32_t* rowptr, const int32_t* src, const int32_t* etype, int64_t num_nodes,
                         const int64_t* seeds, int64_t batch_size, const int32_t* fanouts_host, int64_t hops,
                         uint64_t seed, uint64_t offset, int32_t* local_of, int64_t cap_nodes, int64_t cap_edges,
                         int32_t* n_id, int32_t* b_rowptr, int32_t* b_src, int32_t* b_type, int64_t* b_coo,
                         int32_t* counts, void* workspace, size_t workspace_bytes, void* stream);
/* The same batch plus everything its consumers would otherwise derive with further launches (one call = 11 launches for
 * two hops): the by-SOURCE CSR rowptr_t [cap_nodes+1] / dst_t / eid_t / type_t [cap_edges] -- identical to
 * stemgnn_csr_build(key_row = 0) of b_coo followed by a gather of the types (rows in edge order) --, inv_deg
 * [cap_nodes] = 1 / max(in-degree, 1) (stemgnn_inv_degree of b_rowptr), and the int64 forms the reference's batch
 * object carries (pretrain.py:166-176 reads batch.n_id / batch.x / batch.xe): n_id64 [cap_nodes], type64 [cap_edges],
 * x_out[i] = x[n_id[i]] (x = the dataset's node -> feature-row table, int64 [num_nodes]; NULL: x_out = n_id).
 * Here b_coo is written as a CONTIGUOUS [2, E_b] (row stride E_b, not cap_edges), and `counts` is only ever written
 * (by one thread of the last-but-one launch): it may be device-accessible pinned host memory (hipHostMalloc), which
 * saves the size copy -- the caller then waits for an event recorded behind the call instead.  n_id64 / type64 /
 * x_out may each be NULL.  rowptr_t / dst_t / eid_t / type_t may ALL be NULL: no by-source view is built (its per-row
 * ordering pass is quadratic in a row's length: callers skip it for graphs whose out-degrees are not small and sort
 * the COO instead, stemgnn_csr_build). */
int stemgnn_sample_batch_views(const int32_t* rowptr, const int32_t* src, const int32_t* etype, int64_t num_nodes,
                               const int64_t* seeds, int64_t batch_size, const int32_t* fanouts_host, int64_t hops,
                               uint64_t seed, uint64_t offset, int32_t* local_of, int64_t cap_nodes, int64_t cap_edges,
                               int32_t* n_id, int32_t* b_rowptr, int32_t* b_src, int32_t* b_type, int64_t* b_coo,
                               int32_t* counts, int32_t* rowptr_t, int32_t* dst_t, int32_t* eid_t, int32_t* type_t,
                               float* inv_deg, int64_t* n_id64, int64_t* type64, const int64_t* x, int64_t* x_out,
                               void* workspace, size_t workspace_bytes, void* stream);

/* Fan-out -1 (every in-neighbour: the reference's evaluation loaders, utils/loader.py:18-25, finetune.py:237) or any
 * per-hop mix of -1 and positive fan-outs.  A hop's entries are as many as its frontier's in-degrees add up to, so the
 * batch is built in steps and the HOST reads one size per hop:
 *   begin(seeds)  ->  per hop h: hop_sizes (entries per frontier node, their exclusive scan, *total)  -> the caller
 *   allocates s_src / s_type [total] and makes sure n_id holds n_cap >= nodes so far + min(total, num_nodes) entries  ->
 *   hop_expand (copy / sample the entries, number the new nodes; *nodes_after, if given, = nodes known now)  ->  after
 *   the last hop the caller reads state[hops + 1] = N_b and state[16 + hops] = E_b, allocates the outputs at exactly
 *   those sizes  ->  finish (by-target CSR, contiguous COO [2, E_b], 1 / in-degree, the int64 vectors; the scratch map
 *   is left clean; no by-source view: an evaluation batch has no backward).
 * state: 32 int32 on the device, owned by the caller for the duration of the batch.  frontier_cap: an upper bound of the
 * hop's frontier (hop 0: batch_size; then min(previous total, num_nodes)).  cnt / ent_base / wins / new_base:
 * [frontier_cap] int32 each; the arrays given to hop_expand are handed to finish again (host arrays of device
 * pointers, one per hop).  total / nodes_after / counts may be device-accessible pinned host memory. */
int stemgnn_sampler_full_begin(const int64_t* seeds, int64_t batch_size, int64_t num_nodes, int32_t* local_of,
                               int32_t* n_id, int32_t* state, void* stream);
int stemgnn_sampler_full_hop_sizes(const int32_t* rowptr, const int32_t* n_id, int32_t* state, int32_t hop,
                                   int32_t fanout, int64_t frontier_cap, int32_t* cnt, int32_t* ent_base,
                                   int32_t* total, void* stream);
int stemgnn_sampler_full_hop_expand(const int32_t* rowptr, const int32_t* src, const int32_t* etype, int32_t* n_id,
                                    int64_t n_cap, int32_t* state, int32_t hop, int32_t fanout, uint64_t seed,
                                    uint64_t offset, int64_t batch_size, int64_t frontier_cap, int64_t entries,
                                    int32_t* cnt, const int32_t* ent_base, int32_t* s_src, int32_t* s_type,
                                    int32_t* wins, int32_t* new_base, int32_t* local_of, int32_t* nodes_after,
                                    void* stream);
size_t stemgnn_sampler_full_finish_workspace_bytes(int64_t num_batch_edges);
int stemgnn_sampler_full_finish(int32_t* state, int32_t hops, const int32_t* fanouts_host,
                                const int64_t* frontier_caps_host, int32_t* const* cnt_h, int32_t* const* ent_base_h,
                                int32_t* const* s_src_h, int32_t* const* s_type_h, int32_t* local_of, const int32_t* n_id,
                                int64_t num_batch_nodes, int64_t num_batch_edges, int32_t* b_rowptr, int32_t* b_src,
                                int32_t* b_type, int64_t* b_coo, float* inv_deg, int64_t* n_id64, int64_t* type64,
                                const int64_t* x, int64_t* x_out, int32_t* counts, void* workspace,
                                size_t workspace_bytes, void* stream);

/* out[i] = table[index[i]] for int32 tables (edge-type id per CSR slot = xe[eid[slot]]). */
int stemgnn_gather_i32(const int32_t* table, const int32_t* index, int64_t n, int32_t* out, void* stream);

/* Same sort applied to arbitrary int32 keys in [0, num_keys): rowptr [num_keys+1], perm [n]. */
int stemgnn_group_by_key(const int32_t* keys, int64_t n, int64_t num_keys, int32_t* rowptr, int32_t* perm,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * K1: MySAGEConv message + mean aggregation, forward.
 * Replaces MessagePassing.propagate -> message -> MeanAggregation (model/encoder.py:82,
 * 94-97):  agg[i] = (1/max(deg(i),1)) * sum_{slots of i} relu(x[src] + ea).
 * Edge attribute source, exactly one of:
 *   edge_attr != NULL : dense [E, D] rows addressed by ORIGINAL edge id (drop-in signature)
 *   etab != NULL      : type table [T, D] + etype_slot [E] (type id per CSR slot)
 *   both NULL         : no edge term (edge_attr=None branch, encoder.py:95)
 * D must be a multiple of 4 and <= 2048; rows must be 16-byte aligned.
 * ------------------------------------------------------------------------------------ */
int stemgnn_sage_agg_fwd(const float* x, int64_t num_nodes, int64_t dim,
                         const int32_t* rowptr, const int32_t* src, const int32_t* eid,
                         const float* edge_attr, const float* etab, const int32_t* etype_slot, int64_t num_types,
                         float* agg, void* stream);

/* Measurement aid (bench.py roofline leg): while enabled, every stemgnn_sage_agg_fwd launch is
 * stamped with its own begin/end HIP events (hipExtLaunchKernelGGL); collect() waits for them
 * and returns the summed kernel time in ms and the launch count through HOST pointers. */
int stemgnn_profile_k1(int enable);
int stemgnn_profile_k1_collect(double* total_ms_host, int64_t* launches_host);
/* The same per launch, in launch order: ms_host [capacity] (HOST array), *launches_host = launches recorded. */
int stemgnn_profile_k1_collect_each(float* ms_host, int64_t capacity, int64_t* launches_host);

/*
 * K2: backward of K1 w.r.t. x (PyG autograd: index_select backward = index_add,
 * scatter-mean backward = gather / count).  Deterministic, no atomics:
 *   gx[s] = sum_{slots of s in the by-source CSR} 1[x[s] + ea > 0] * g_agg[dst] * inv_deg[dst]
 * inv_deg [N] = 1 / max(in-degree, 1) (stemgnn_inv_degree).
 */
int stemgnn_sage_agg_bwd(const float* g_agg, const float* x, int64_t num_nodes, int64_t dim,
                         const int32_t* rowptr_t, const int32_t* dst_t, const int32_t* eid_t,
                         const float* inv_deg,
                         const float* edge_attr, const float* etab, const int32_t* etype_slot_t, int64_t num_types,
                         float* g_x, void* stream);

/* The same, ADDING into g_x (which already holds lin_r's share of the gradient, encoder.py:83-87) instead of
 * overwriting it: rows without out-edges are left untouched.  Replaces autograd's separate accumulation add. */
int stemgnn_sage_agg_bwd_acc(const float* g_agg, const float* x, int64_t num_nodes, int64_t dim,
                             const int32_t* rowptr_t, const int32_t* dst_t, const int32_t* eid_t,
                             const float* inv_deg,
                             const float* edge_attr, const float* etab, const int32_t* etype_slot_t, int64_t num_types,
                             float* g_x, void* stream);

int stemgnn_inv_degree(const int32_t* rowptr, int64_t num_nodes, float* inv_deg, void* stream);

/* Plain mean aggregation agg[i] = mean_{slots of i} x[src] (no edge term, no relu) and its
 * backward: torch_scatter.scatter_mean(x[col], row) of MixtureSageLayer (model/encoder.py:124),
 * evaluated on the CSR of the FLIPPED graph. */
int stemgnn_mean_agg_fwd(const float* x, int64_t num_nodes, int64_t dim, const int32_t* rowptr, const int32_t* src,
                         float* agg, void* stream);
int stemgnn_mean_agg_bwd(const float* g_agg, int64_t num_nodes, int64_t dim, const int32_t* rowptr_t,
                         const int32_t* dst_t, const float* inv_deg, float* g_x, void* stream);

/* K1 / K2 on skewed graphs: rows with more than `heavy_above` edges (hubs) are cut into chunks of
 * `chunk` edges, reduced by separate groups into `partial` rows and summed per row in chunk order
 * (bit-reproducible).  Same operands and results (to fp32 summation order) as
 * stemgnn_sage_agg_fwd/bwd; relu = 0 gives the plain mean of stemgnn_mean_agg_fwd/bwd (pass NULL
 * edge operands; x may be NULL in the backward).  num_edges bounds the CSR's live edge count.
 * Caller-allocated plan buffers, sized so they cannot overflow:
 *   cap_items >= 2*(num_edges/chunk) + 2 : item_row, item_beg [cap_items], partial [cap_items, dim]
 *   cap_heavy >=    num_edges/chunk  + 1 : heavy_row [cap_heavy], heavy_span [cap_heavy, 2]
 *   counts [2]
 * build_plan != 0: the call fills the plan from rowptr as it goes; 0: reuse the plan an earlier
 * call over the SAME rowptr/chunk/heavy_above filled.  heavy_above >= chunk > 0. */
int stemgnn_sage_agg_fwd_split(const float* x, int64_t num_nodes, int64_t dim, int64_t num_edges,
                               const int32_t* rowptr, const int32_t* src, const int32_t* eid,
                               const float* edge_attr, const float* etab, const int32_t* etype_slot,
                               int64_t num_types, float* agg, int32_t relu,
                               int32_t chunk, int32_t heavy_above, int32_t build_plan,
                               int64_t cap_items, int64_t cap_heavy,
                               int32_t* item_row, int32_t* item_beg, int32_t* heavy_row, int32_t* heavy_span,
                               int32_t* counts, float* partial, void* stream);
int stemgnn_sage_agg_bwd_split(const float* g_agg, const float* x, int64_t num_nodes, int64_t dim, int64_t num_edges,
                               const int32_t* rowptr_t, const int32_t* dst_t, const int32_t* eid_t,
                               const float* inv_deg,
                               const float* edge_attr, const float* etab, const int32_t* etype_slot_t,
                               int64_t num_types, float* g_x, int32_t relu,
                               int32_t chunk, int32_t heavy_above, int32_t build_plan,
                               int64_t cap_items, int64_t cap_heavy,
                               int32_t* item_row, int32_t* item_beg, int32_t* heavy_row, int32_t* heavy_span,
                               int32_t* counts, float* partial, void* stream);

/* ------------------------------------------------------------------------------------
 * K4: BatchNorm1d (training statistics) + ReLU/LeakyReLU + Dropout
 * (model/encoder.py:173,313-317).
 * ------------------------------------------------------------------------------------ */
size_t stemgnn_bn_workspace_bytes(int64_t num_rows, int64_t dim);

/* Column mean / biased variance of y [N, D]; writes mean [D], rstd [D] = 1/sqrt(var+eps);
 * when running_mean/var != NULL updates them in place with `momentum` (unbiased variance,
 * torch.nn.BatchNorm1d semantics). */
int stemgnn_bn_stats(const float* y, int64_t num_rows, int64_t dim, float eps,
                     float* mean, float* rstd, float* running_mean, float* running_var, float momentum,
                     void* workspace, size_t workspace_bytes, void* stream);

/* Same finalisation from per-block column partials [blocks][2][D] (sum, sum of squares) that a
 * producer kernel already wrote (stemgnn_linear_fwd's fused statistics).  num_batches_tracked != NULL:
 * BatchNorm1d's int64 call counter, incremented by one in the same launch. */
int stemgnn_bn_stats_from_partials(const float* partial, int64_t blocks, int64_t num_rows, int64_t dim, float eps,
                                   float* mean, float* rstd, float* running_mean, float* running_var,
                                   float momentum, int64_t* num_batches_tracked, void* stream);

/* out = dropout(act((y - mean) * rstd * gamma + beta)).
 * act: 0 none, 1 relu / leaky-relu with `negative_slope`.  p = 0 disables dropout; the keep
 * decision of element (r, c) is philox(seed, offset)[r*D + c] >= p, reproducible by
 * stemgnn_dropout_keep_mask.  mean/rstd == NULL skips the normalisation (normalize='none'). */
int stemgnn_bn_act_drop_fwd(const float* y, int64_t num_rows, int64_t dim,
                            const float* mean, const float* rstd, const float* gamma, const float* beta,
                            int act, float negative_slope, float p, uint64_t seed, uint64_t offset,
                            float* out, void* stream);

/* Backward: g_y [N, D], g_gamma [D], g_beta [D] from g_out, recomputing x_hat, the
 * activation sign and the dropout mask.  */
int stemgnn_bn_act_drop_bwd(const float* g_out, const float* y, int64_t num_rows, int64_t dim,
                            const float* mean, const float* rstd, const float* gamma, const float* beta,
                            int act, float negative_slope, float p, uint64_t seed, uint64_t offset,
                            float* g_y, float* g_gamma, float* g_beta,
                            void* workspace, size_t workspace_bytes, void* stream);

/* keep[i] (uint8) for i in [0, n): the mask the two kernels above use. */
int stemgnn_dropout_keep_mask(int64_t n, float p, uint64_t seed, uint64_t offset, uint8_t* keep, void* stream);

/* ------------------------------------------------------------------------------------
 * K3 / K5: dense projections on the matrix cores, fp32 results (see stemgnn_linear_set_mode).
 * Replaces lin_l(agg) + lin_r(x) (model/encoder.py:83-87), project_in / project_out
 * (model/vq.py:881,1041) and the decoders' nn.Linear (model/pt_model.py:42,80,94).
 * All matrices dense row-major fp32; K1, K2, N multiples of 4.
 * ------------------------------------------------------------------------------------ */

/* The three dense products below run on the bf16 matrix cores by default: every fp32 operand is cut exactly
 * into three bf16 pieces and the six significant piece products are accumulated in fp32 -- no further from the
 * fp64 result than the fp32-MFMA kernels (csrc/linear.hip).  mode 0 selects the fp32-MFMA kernels, 1 the
 * default, 2 the bf16 GEMM mode of BASELINE config 5 (what autocast would do to the reference's nn.Linear calls:
 * both operands of every product rounded to bf16, one matrix pass, fp32 accumulation, fp32 bias / output / bias
 * gradient; the quantiser's similarity / arg-max core keeps the exact form, model/vq.py:623,634); any other value only
 * queries.  Returns the previous mode.  The tile plan (stemgnn_linear_stats_blocks) depends on the mode: set it
 * before sizing buffers.  STEMGNN_GEMM=f32 / bf16 in the environment starts the process in mode 0 / 2. */
int stemgnn_linear_set_mode(int mode);
/* Large products (>= 8 192 rows, both feature extents >= 256 and multiples of 64, >= 1e10 flop: the D = 768
 * configurations) run on the big-tile core (csrc/bigtile.hip: one hand-written 256 x 256 x 64 bf16 MFMA GEMM over
 * operands cut into bf16 planes by a cut pass) in mode 1 (the six exact piece products as one contraction of 6 K) and
 * mode 2 (one rounded plane) alike.  Its scratch -- the planes, the weight gradient's split slabs -- comes from the
 * CALLER: one arena per (device, stream), registered with stemgnn_linear_set_scratch and sized by
 * stemgnn_linear_scratch_bytes(rows, a, b) (an upper bound for every product y = x w^T, its backward-data and its weight
 * gradient with at most `rows` rows and feature extents a x b, either way round) and, for the quantiser's code
 * assignment at large codebooks, stemgnn_vq_assign_scratch_bytes.  The library never allocates, frees or synchronises
 * for it; the arena must stay valid until the stream has run the products enqueued while it was registered (NULL
 * unregisters).  A qualifying product that finds no arena, or one too small, runs on the 128-row tile kernels and is
 * counted in stemgnn_linear_bigtile_fallbacks -- a capacity miss the host can see, never an error swallowed: every other
 * failure of the core is returned.  stemgnn_linear_set_bigtile(0) switches the core off (1 = default; other values only
 * query; returns the previous setting); stemgnn_linear_bigtile_calls counts the products it has served. */
size_t stemgnn_linear_scratch_bytes(int64_t max_rows, int64_t dim_a, int64_t dim_b);
size_t stemgnn_vq_assign_scratch_bytes(int64_t num_rows, int64_t heads, int64_t code_dim, int64_t codebook_size);
int stemgnn_linear_set_scratch(void* scratch, size_t bytes, void* stream);
int stemgnn_linear_set_bigtile(int on);
/* Exact mode on the big-tile core: the products that contract along their operands' ROWS' columns (forward,
 * backward-data, the code assignment) read operands cut into TWO fp16 pieces of rows scaled by a power of two (three
 * matrix passes, fp32-accurate: csrc/bigtile.hip "pair format"; 1 = default) or into the three bf16 pieces (six passes; 0).
 * The weight gradient, which contracts over the rows themselves, always takes the bf16 pieces.  Other values only query;
 * returns the previous setting. */
int stemgnn_linear_set_pair(int on);
/* Measurement aid (bench.py's matrix-roofline leg), like stemgnn_profile_k1: while enabled, every launch of the big-tile
 * core is stamped with its own begin / end HIP events; collect() waits for them and returns (HOST pointers) the summed
 * kernel time in ms, the matrix work those launches EXECUTED in flop (2 x rows x rows x contraction over all segments:
 * six piece products per fp32 product in mode 1, one in mode 2) and the launch count. */
int stemgnn_profile_bigtile(int enable);
int stemgnn_profile_bigtile_collect(double* total_ms_host, double* total_flop_host, int64_t* launches_host);
int64_t stemgnn_linear_bigtile_calls(void);
int64_t stemgnn_linear_bigtile_fallbacks(void);

/* y [M, N] = x1 [M, K1] w1[N, K1]^T (+ x2 [M, K2] w2 [N, K2]^T when K2 > 0) + bias [N] (NULL: none).
 * stats_partial (may be NULL): receives per-row-block column sums / sums of squares of y,
 * [stemgnn_linear_stats_blocks(M, N)][2][N]; *stats_blocks_host (host pointer, may be NULL)
 * receives that block count.  x1_rows: rows >= x1_rows of x1 COUNT AS ZERO and are never read (the aggregate of a
 * sampled batch, whose edges all end in the leading, expanded nodes; x1 may be a [x1_rows, K1] buffer): row tiles
 * past them skip the x1 half of the contraction.  Pass -1 (or M) when x1 has M meaningful rows. */
size_t stemgnn_linear_stats_partial_bytes(int64_t num_rows, int64_t out_dim);
/* Two stemgnn_sample_edges draws over the same edge list in one launch (same picks as two separate calls with
 * (seed, offset_a) and (seed, offset_b)): sample a marks its picks in selected_a (cleared first; may be NULL), sample b
 * also returns the picked edges' types (sel_type_b may be NULL). */
int stemgnn_sample_edges2(const int64_t* edge_index, const int64_t* edge_type, int64_t num_edges, uint64_t seed,
                          int64_t k_a, uint64_t offset_a, int64_t* perm_a, int64_t* sel_index_a, int64_t sel_stride_a,
                          uint8_t* selected_a, int64_t k_b, uint64_t offset_b, int64_t* perm_b, int64_t* sel_index_b,
                          int64_t sel_stride_b, int64_t* sel_type_b, void* stream);

/* stemgnn_edge_concat_fwd (out[e] = [z[u_e], z[v_e]]) and target[e] = table[type[e]] (a row gather; an id outside
 * [0, T) gives a zero row) in one launch: the inputs and targets of the topo-sem head (reference pt_model.py:72-81). */
int stemgnn_edge_concat_gather(const float* z, int64_t num_nodes, int64_t dim, const int64_t* edge_index,
                               int64_t num_edges, float* out, const float* table, int64_t table_rows,
                               const int64_t* type, float* target, void* stream);

/* stemgnn_edge_concat_bwd, and in the same launch g_z[i] += add_a[i] + add_b[i] for the first add_n elements (the heads
 * phase: the gradients of the two seed-row heads join the query's gradient there). */
int stemgnn_edge_concat_bwd_add(const float* g_out, int64_t num_nodes, int64_t dim, const int64_t* edge_index,
                                int64_t num_edges, float* g_z, const float* add_a, const float* add_b, int64_t add_n,
                                void* stream);

/* stemgnn_edge_dot_fwd + stemgnn_edge_bce_loss in one launch (reference model/encoder.py:364-366 + model/pt_model.py:62-65):
 * scores of kp positive then kn negative edges, loss[0] = mean -log(sigmoid + EPS) + mean -log(1 - sigmoid + EPS),
 * coef[e] = d loss / d score_e.  The scores are not kept.  workspace: stemgnn_edge_dot_bce_workspace_bytes(kp + kn). */
size_t stemgnn_edge_dot_bce_workspace_bytes(int64_t num_edges);
int stemgnn_edge_dot_bce(const float* z, int64_t num_nodes, int64_t dim, const int64_t* edge_index, int64_t kp, int64_t kn,
                         float* loss, float* coef, void* workspace, size_t workspace_bytes, void* stream);

/* Deterministic forms of stemgnn_edge_dot_bwd_scaled / stemgnn_edge_concat_bwd (reference model/encoder.py:346-354
 * InnerProductDecoder backward, model/pt_model.py:72-81 cat(z_u, z_v) backward; ATen scatters with atomics there too):
 * the edge list is grouped by node (stable radix sort, twice: by each endpoint row) and one lane group per node adds
 * its edges in that order.  _dot_: g_z[n] = g_scalar * sum_e coef[e] z[other endpoint of e]  (g_z is overwritten);
 * _concat_: g_z[n] += sum_e g_out[e][:D] (n first endpoint) + g_out[e][D:] (n second).  workspace:
 * stemgnn_edge_det_workspace_bytes(N, E).  stemgnn_set_deterministic(1) (or STEMGNN_DETERMINISTIC=1) makes the heads
 * phase use them always (0, the default = the atomic forms for the usual sampled batch and the sorted forms once the
 * scatters are large enough for them to be the faster ones, k * D >= 5e7 elements; negative = query); returns the
 * previous setting. */
size_t stemgnn_edge_det_workspace_bytes(int64_t num_nodes, int64_t num_edges);
int stemgnn_edge_dot_bwd_det(const float* coef, const float* g_scalar, const float* z, int64_t num_nodes, int64_t dim,
                             const int64_t* edge_index, int64_t num_edges, float* g_z, void* workspace,
                             size_t workspace_bytes, void* stream);
int stemgnn_edge_concat_bwd_det(const float* g_out, int64_t num_nodes, int64_t dim, const int64_t* edge_index,
                                int64_t num_edges, float* g_z, void* workspace, size_t workspace_bytes, void* stream);
int stemgnn_set_deterministic(int on);

/* y[M, N] = x[M, K] w^T + bias (w [N, K]), or y = x w with w given as [K, N] (weight_is_kn != 0: the backward-data
 * form dx = dy w), for products over FEW rows (M <= 65536, N % 32 == 0, K % 16 == 0; exact-bf16 mode only): one wave
 * per 32 x 32 output tile, operands read straight from global memory (csrc/wsgemm.hip).  Same bits as
 * stemgnn_linear_fwd / stemgnn_linear_bwd_data.  The heads phase runs its three seed-row / sampled-pair products as
 * one such launch. */
int stemgnn_linear_few_rows(const float* x, const float* w, const float* bias, int64_t num_rows, int64_t out_dim,
                            int64_t in_dim, float* y, int32_t weight_is_kn, void* stream);

/* The weight-stationary kernels take the products with one 128-column operand and at least `min_tiles` 128-row tiles
 * (default 128; 0 = never: every product on the tile kernel).  Negative: query only.  Returns the previous value.
 * With stemgnn_linear_set_pair(1) (the default) that is csrc/wspair.hip: operands in the pair format (two fp16 pieces
 * of a row scaled by a power of two, three matrix passes, fp32-accurate), which also takes a sampled batch's two-operand
 * layer product (x1_rows < num_rows: the leading rows' aggregate is multiplied in the blocks' prologues);
 * stemgnn_linear_wsp_calls counts its launches.  With stemgnn_linear_set_pair(0): csrc/wsgemm.hip, three exact bf16
 * pieces and six passes -- the same bits as the tile kernel; the switches exist for A/B runs and tests. */
int stemgnn_linear_set_ws(int min_tiles);
int64_t stemgnn_linear_wsp_calls(void);
int64_t stemgnn_linear_stats_blocks(int64_t num_rows, int64_t out_dim);
int stemgnn_linear_fwd(const float* x1, const float* w1, int64_t k1, const float* x2, const float* w2, int64_t k2,
                       const float* bias, int64_t num_rows, int64_t out_dim, float* y, float* stats_partial,
                       int64_t* stats_blocks_host, int64_t x1_rows, void* stream);

/* The same with a row limit on the OUTPUT: rows >= store_rows of y are computed (they count in stats_partial) but
 * not written -- a BatchNorm layer whose caller reads only its leading rows (the EMA teacher's last layer,
 * pt_model.py:93-97: statistics over all rows, values of the seed rows).  y may be a [store_rows, N] buffer. */
int stemgnn_linear_fwd_rows(const float* x1, const float* w1, int64_t k1, const float* x2, const float* w2, int64_t k2,
                            const float* bias, int64_t num_rows, int64_t out_dim, float* y, float* stats_partial,
                            int64_t* stats_blocks_host, int64_t x1_rows, int64_t store_rows, void* stream);

/* Backward w.r.t. the input of y = x w^T: dx[M, K] = dy[M, N] w[N, K] (autograd of nn.Linear, reference
 * model/encoder.py:83-87, model/vq.py:881,1041).  The weight is read as stored; no transposed copy is made.
 * N, K multiples of 4. */
int stemgnn_linear_bwd_data(const float* dy, const float* w, int64_t num_rows, int64_t out_features,
                            int64_t in_features, float* dx, void* stream);

/* dw [N, K] = dy [M, N]^T x [M, K];  db [N] = column sums of dy (NULL: skip).  Deterministic
 * two-stage reduction over row splits (no atomics). */
size_t stemgnn_linear_bwd_weight_workspace_bytes(int64_t num_rows, int64_t out_dim, int64_t in_dim);
int stemgnn_linear_bwd_weight(const float* dy, const float* x, int64_t num_rows, int64_t out_dim, int64_t in_dim,
                              float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream);

/* out [cols, rows] = in [rows, cols]^T (weights; backward-data = stemgnn_linear_fwd on w^T). */
int stemgnn_transpose(const float* in, int64_t rows, int64_t cols, float* out, void* stream);

/* ------------------------------------------------------------------------------------
 * K6+K7+K8: cosine-similarity codebook assignment, fused.
 * Replaces l2norm (model/vq.py:891) + CosineSimCodebook.forward's einsum / argmax /
 * one-hot / einsum (vq.py:650-657) + straight-through and commitment MSE
 * (vq.py:931-937,1007-1009).  fp32-accurate similarity product (same mode as the dense products).
 *   xp      [N, H*Dc]  project_in output, head h in columns [h*Dc, (h+1)*Dc)
 *   embed   [H, K, Dc]
 *   xn      [N, H*Dc]  optional (may be NULL): l2-normalised input
 *   norm    [N, H]     ||xp_h|| before clamping at 1e-12 (saved for backward with xp)
 *   ind     [N, H]     int64 arg-max code (lowest index wins ties)
 *   quant   [N, H*Dc]  training: xn + (q - xn) (straight-through value); eval: q
 *   sqerr   [1]        sqerr_scale * sum over all elements of (q - xn)^2 (commitment numerator; pass
 *                      commitment_weight / (N*H*Dc) for the weighted mean of vq.py:1007-1009)
 * Dc must be a multiple of 4 and <= 1024, K <= 65536.
 * ------------------------------------------------------------------------------------ */
size_t stemgnn_vq_workspace_bytes(int64_t num_rows, int64_t heads, int64_t code_dim, int64_t codebook_size);
/* Which kernel the calling thread's most recent stemgnn_vq_assign_fwd / _lean call launched: 0 none yet, 1 the tile
 * form (k_vq_assign), 2 the weight-stationary form (k_vq_assign_ws: K = Dc = 128, >= 16 384 rows), 4 the large-codebook
 * form (K >= 512, Dc >= 256, >= 8 192 rows, a scratch arena registered: the six exact piece products of all heads as one
 * launch of the big-tile core, the arg-max taken from the accumulators per 256-code tile -- no [N, K] similarity matrix
 * --, one finishing pass; stemgnn_linear_set_bigtile(0) keeps it on the tile form).  The golden tests use it to make
 * sure a production-shape fixture was served by the production kernel. */
int stemgnn_vq_assign_last_path(void);

int stemgnn_vq_assign_fwd(const float* xp, int64_t num_rows, int64_t heads, int64_t code_dim,
                          const float* embed, int64_t codebook_size, int training,
                          float* xn, float* norm, int64_t* ind, float* quant, float* sqerr, float sqerr_scale,
                          void* workspace, size_t workspace_bytes, void* stream);

/* g_xp = d/dxp [ <g_quant, quant> + g_loss * commit_weight * mean((q - xn)^2) ]
 * (straight-through: d quant / d xn = I; q is constant; xn is recomputed from xp and norm).
 * g_loss [1] device scalar, may be NULL (no commitment term). */
int stemgnn_vq_assign_bwd(const float* g_quant, const float* g_loss, float commit_weight,
                          const float* xp, const float* norm, const int64_t* ind, const float* embed,
                          int64_t num_rows, int64_t heads, int64_t code_dim, int64_t codebook_size,
                          float* g_xp, void* stream);

/* The same assignment without any [N, H*Dc] output: ind, norm and the commitment sum only.  The squared error of a
 * row follows from its arg-max, |q - xn|^2 = |q|^2 + |xn|^2 - 2 <q, xn>, with esq [H, K] = |embed|^2 per code
 * (stemgnn_code_sqnorm).  For callers that consume project_out(codes) (stemgnn_codes_project) and not the codes. */
int stemgnn_vq_assign_lean(const float* xp, int64_t num_rows, int64_t heads, int64_t code_dim, const float* embed,
                           const float* esq, int64_t codebook_size, float* norm, int64_t* ind, float* sqerr,
                           float sqerr_scale, void* workspace, size_t workspace_bytes, void* stream);
int stemgnn_code_sqnorm(const float* embed, int64_t num_codes, int64_t code_dim, float* esq, void* stream);

/* project_out of quantised rows (vq.py:1041) read off a table: out[n] = bias + sum_h table[h][ind[n][h]], table
 * [H, K, D] = embed[h, k] W_out[:, h*Dc:(h+1)*Dc]^T (stemgnn_small_gemm), bias [D] or NULL. */
int stemgnn_codes_project(const float* table, const int64_t* ind, const float* bias, int64_t num_rows, int64_t heads,
                          int64_t codebook_size, int64_t dim, float* out, void* stream);

/* sums[h*K + k][:] = sum of the rows g[m] (g [M, D]) whose code in head h is k (ind [M, H] int64): the one-hot^T g
 * product on the matrix cores, with the one-hot operand built in registers.  Deterministic (row splits reduced in a
 * fixed order).  db[d] = sum_k sums[k][d] over one head's K rows = the column sums of g (stemgnn_segment_colsum). */
size_t stemgnn_code_segment_sums_workspace_bytes(int64_t num_rows, int64_t heads, int64_t codebook_size, int64_t dim);
int stemgnn_code_segment_sums(const int64_t* ind, int64_t heads, int64_t codebook_size, const float* g, int64_t num_rows,
                              int64_t dim, float* sums, void* workspace, size_t workspace_bytes, void* stream);
int stemgnn_segment_colsum(const float* sums, int64_t codebook_size, int64_t dim, float* db, void* stream);

/* c(m, n) = sum_k a(m, k) b(k, n) for small operands with arbitrary ELEMENT strides, `batches` independent products
 * (pointer offsets *_batch per batch): plain fp32 FMA in a fixed order.  For codebook-sized products only. */
int stemgnn_small_gemm(const float* a, int64_t a_m, int64_t a_k, int64_t a_batch, const float* b, int64_t b_k,
                       int64_t b_n, int64_t b_batch, float* c, int64_t c_m, int64_t c_n, int64_t c_batch, int64_t M,
                       int64_t N, int64_t K, int64_t batches, void* stream);

/* stemgnn_vq_assign_bwd with project_out's backward-data product inside (code_dim <= 128, default matrix-core
 * mode): g_out [N, D] is the gradient of project_out's OUTPUT, w_out [D, H*Dc] its weight as stored; the
 * [N, H*Dc] gradient of the quantised rows (g_out w_out) lives only in the kernel's accumulators. */
int stemgnn_vq_assign_bwd_fused(const float* g_out, int64_t dim, const float* w_out, const float* g_loss,
                                float commit_weight, const float* xp, const float* norm, const int64_t* ind,
                                const float* embed, int64_t num_rows, int64_t heads, int64_t code_dim,
                                int64_t codebook_size, float* g_xp, void* stream);

/* K10: EMA statistics (vq.py:661-672): bins [H, K] = #rows per code, embed_sum [H, K, Dc]
 * = sum of the normalised rows xp/max(norm,1e-12) per code.  Deterministic (sorted segment sums). */
size_t stemgnn_vq_ema_workspace_bytes(int64_t num_rows, int64_t heads, int64_t code_dim, int64_t codebook_size);
int stemgnn_vq_ema_stats(const float* xp, const float* norm, const int64_t* ind, int64_t num_rows, int64_t heads, int64_t code_dim,
                         int64_t codebook_size, float* bins, float* embed_sum,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * K11: InnerProductDecoder edge scores (model/encoder.py:364-366):
 *   out[e] = <z[u_e], z[v_e]>,  edge_index [2, E] int64.  Backward scatters with fp32
 *   atomics into g_z (which must be zero-initialised by the caller).
 * ------------------------------------------------------------------------------------ */
int stemgnn_edge_dot_fwd(const float* z, int64_t num_nodes, int64_t dim, const int64_t* edge_index,
                         int64_t num_edges, float* out, void* stream);
int stemgnn_edge_dot_bwd(const float* g_out, const float* z, int64_t num_nodes, int64_t dim,
                         const int64_t* edge_index, int64_t num_edges, float* g_z, void* stream);

/* topo_recon_loss (model/pt_model.py:62-65) from the edge scores of kp positive then kn negative
 * edges: loss[0] = mean -log(sigmoid(d)+1e-15) over positives + mean -log(1-sigmoid(d)+1e-15)
 * over negatives; coef[e] = d loss / d dots[e].  The backward is stemgnn_edge_dot_bwd_scaled:
 * g_z += (g_scalar[0] * coef[e]) * z[other endpoint]  (g_z zero-initialised by the caller). */
int stemgnn_edge_bce_loss(const float* dots, int64_t num_pos, int64_t num_neg, float* loss, float* coef,
                          void* stream);
int stemgnn_edge_dot_bwd_scaled(const float* coef, const float* g_scalar, const float* z, int64_t num_nodes,
                                int64_t dim, const int64_t* edge_index, int64_t num_edges, float* g_z,
                                void* stream);

/* ------------------------------------------------------------------------------------
 * Fused scalar losses (each replaces a chain of tiny ATen kernels and its autograd).
 * loss / g_loss are 1-element device scalars; reductions are fixed-order (reproducible).
 * ------------------------------------------------------------------------------------ */
/* loss = scale * mean((pred - target)^2) over n elements (F.mse_loss, model/pt_model.py:43,81) */
size_t stemgnn_loss_workspace_bytes(int64_t n);  /* n = elements of the largest reduction list (rows, heads * ids, 256) */
int stemgnn_mse_loss_fwd(const float* pred, const float* target, int64_t n, float scale, float* loss, void* workspace,
                         size_t workspace_bytes, void* stream);
int stemgnn_mse_loss_bwd(const float* pred, const float* target, int64_t n, float scale, const float* g_loss,
                         float* g_pred, void* stream);
/* loss = scale * mean_r (1 - cos(z_r, h_r)) with F.normalize's eps clamp (model/pt_model.py:96-100);
 * save [rows, 3] carries (cos, 1/|z|, |h|) to the backward, which returns the gradient w.r.t. h. */
int stemgnn_cosine_loss_fwd(const float* z, const float* h, int64_t rows, int64_t dim, float scale, float* loss,
                            float* save, void* workspace, size_t workspace_bytes, void* stream);
int stemgnn_cosine_loss_bwd(const float* z, const float* h, int64_t rows, int64_t dim, float scale,
                            const float* g_loss, const float* save, float* g_h, void* stream);
/* orthogonal_loss_fn(embed[:, ids]) * scale (model/vq.py:232-237,1011-1028): embed [H, K, Dc], ids
 * int64 [M] distinct, Dc <= 1024.  The backward writes the dense gradient g_embed [H, K, Dc] (zero off the ids).
 * Workspace: stemgnn_loss_workspace_bytes(H * M). */
int stemgnn_ortho_loss_fwd(const float* embed, const int64_t* ids, int64_t heads, int64_t codebook_size,
                           int64_t code_dim, int64_t num_ids, float scale, float* loss, void* workspace,
                           size_t workspace_bytes, void* stream);
int stemgnn_ortho_loss_bwd(const float* embed, const int64_t* ids, int64_t heads, int64_t codebook_size,
                           int64_t code_dim, int64_t num_ids, float scale, const float* g_loss, float* g_embed,
                           void* stream);

/* K12 feed: out[e] = concat(z[u_e], z[v_e]) ([E, 2D]); backward scatters g_out back
 * (atomics; g_z zero-initialised by the caller).  (model/pt_model.py:80) */
int stemgnn_edge_concat_fwd(const float* z, int64_t num_nodes, int64_t dim, const int64_t* edge_index,
                            int64_t num_edges, float* out, void* stream);
int stemgnn_edge_concat_bwd(const float* g_out, int64_t num_nodes, int64_t dim, const int64_t* edge_index,
                            int64_t num_edges, float* g_z, void* stream);

/* Row gather: out[i, :] = table[index[i], :] (host-side feature lookup of pretrain.py:33-38,
 * moved to the device).  index int64, rows fp32. */
int stemgnn_gather_rows(const float* table, int64_t num_table_rows, int64_t dim, const int64_t* index,
                        int64_t n, float* out, void* stream);
/* The same, counting the indices outside [0, num_table_rows) into bad_count [1] (device int32, zeroed by the call):
 * the reference's node_text_feat[data.x] (pretrain.py:33) raises IndexError for them; out-of-range rows are
 * written as zeros here and the caller raises when bad_count != 0. */
int stemgnn_gather_rows_checked(const float* table, int64_t num_table_rows, int64_t dim, const int64_t* index,
                                int64_t n, float* out, int32_t* bad_count, void* stream);

/* K14: teacher EMA (model/pt_model.py:104-106) on flat parameter buffers, in place:
 *   teacher = teacher * decay + student * (1 - decay). */
int stemgnn_ema_lerp(float* teacher, const float* student, int64_t n, float decay, void* stream);

/* ------------------------------------------------------------------------------------
 * torch.nn.utils.clip_grad_norm_(params, max_norm) (reference pretrain.py:62), L2 norm.
 * grads / sizes are HOST arrays of `count` (<= stemgnn_clip_grad_max_tensors()) device pointers
 * and element counts.  out[0] = total norm, out[1] = the factor applied (<= 1), both on the
 * device; the gradients are scaled in place.  fp64 accumulation in a fixed order.
 * ------------------------------------------------------------------------------------ */
int32_t stemgnn_clip_grad_max_tensors(void);
size_t stemgnn_clip_grad_workspace_bytes(int64_t total_elements, int32_t count);
int stemgnn_clip_grad_norm(float* const* grads, const int64_t* sizes, int32_t count, float max_norm, float* out,
                           void* workspace, size_t workspace_bytes, void* stream);

/* out[0] = sum_i weights[i] * terms[i][0] over up to 8 scalar loss terms (reference pretrain.py:51-58: the
 * lambda-weighted total), fixed order; terms / weights are HOST arrays (device pointers / floats).
 * Backward: g_terms[i] = g_out[0] * weights[i]. */
int stemgnn_weighted_sum(const float* const* terms, const float* weights, int32_t count, float* out, void* stream);
int stemgnn_weighted_sum_bwd(const float* weights, int32_t count, const float* g_out, float* g_terms, void* stream);

/* out[0] = total L2 norm of the gradients, out[1] = min(1, max_norm / (out[0] + 1e-6)); nothing is scaled.
 * Same host arrays and workspace as stemgnn_clip_grad_norm. */
int stemgnn_grad_norm_coef(const float* const* grads, const int64_t* sizes, int32_t count, float max_norm, float* out,
                           void* workspace, size_t workspace_bytes, void* stream);

/* One AdamW update (torch.optim.AdamW semantics, amsgrad off; reference pretrain.py:134-136,63) of `count`
 * (<= stemgnn_clip_grad_max_tensors()) tensors in one launch.  params / grads / exp_avg / exp_avg_sq / sizes are
 * HOST arrays of device pointers and element counts; `step` is the 1-based update count (bias correction);
 * grad_coef (device, may be NULL) multiplies every gradient as it is read -- pass out + 1 of
 * stemgnn_grad_norm_coef to apply clip_grad_norm_ without a pass over the gradients. */
int stemgnn_adamw_step(float* const* params, const float* const* grads, float* const* exp_avg,
                       float* const* exp_avg_sq, const int64_t* sizes, int32_t count, float lr, float beta1,
                       float beta2, float eps, float weight_decay, int64_t step, const float* grad_coef,
                       void* stream);

/* ------------------------------------------------------------------------------------
 * bf16 feature storage (BASELINE config 5): the *_k forms of the entry points above take the node-feature /
 * layer-output operand as either fp32 or bf16 (`kind`: STEMGNN_F32 / STEMGNN_BF16; `void*` = float* or uint16_t*).
 * ------------------------------------------------------------------------------------ */
int stemgnn_sage_agg_fwd_k(const void* x, int32_t x_kind, int64_t num_nodes, int64_t dim, const int32_t* rowptr,
                           const int32_t* src, const int32_t* eid, const float* edge_attr, const float* etab,
                           const int32_t* etype_slot, int64_t num_types, float* agg, void* stream);
int stemgnn_sage_agg_bwd_acc_k(const float* g_agg, const void* x, int32_t x_kind, int64_t num_nodes, int64_t dim,
                               const int32_t* rowptr_t, const int32_t* dst_t, const int32_t* eid_t, const float* inv_deg,
                               const float* edge_attr, const float* etab, const int32_t* etype_slot_t, int64_t num_types,
                               float* g_x, void* stream);
int stemgnn_linear_fwd_rows_k(const float* x1, const float* w1, int64_t k1, const void* x2, int32_t x2_kind,
                              const float* w2, int64_t k2, const float* bias, int64_t num_rows, int64_t out_dim, float* y,
                              float* stats_partial, int64_t* stats_blocks_host, int64_t x1_rows, int64_t store_rows,
                              void* stream);
int stemgnn_linear_bwd_weight_k(const float* dy, const void* x, int32_t x_kind, int64_t num_rows, int64_t out_dim,
                                int64_t in_dim, float* dw, float* db, void* workspace, size_t workspace_bytes,
                                void* stream);
int stemgnn_bn_act_drop_fwd_k(const float* y, int64_t num_rows, int64_t dim, const float* mean, const float* rstd,
                              const float* gamma, const float* beta, int act, float negative_slope, float p,
                              uint64_t seed, uint64_t offset, void* out, int32_t out_kind, void* stream);
int stemgnn_mask_columns_k(const void* x, int32_t kind, int64_t num_rows, int64_t dim, float p, uint64_t seed,
                           uint64_t offset, void* out, void* stream);
/* bad_count may be NULL (no range check) */
int stemgnn_gather_rows_k(const void* table, int32_t kind, int64_t num_table_rows, int64_t dim, const int64_t* index,
                          int64_t n, void* out, int32_t* bad_count, void* stream);

/* ====================================================================================
 * Phase entry points: one call enqueues a whole module forward (or its backward) of the pretraining path, so the
 * host pays one crossing per module instead of one per kernel.  Same arithmetic as the single-op entry points
 * above (they are what these functions launch); structs carry plain device pointers and sizes.
 * ==================================================================================== */

/* Both CSR views of one (sub)graph (stem_gnn_amd.graph.GraphStructure). */
typedef struct stemgnn_graph_view {
  int64_t num_nodes;
  int64_t active_rows;               /* rows >= active_rows have no in-edges (sampled batch); num_nodes if no promise */
  const int32_t* rowptr;             /* by target: forward aggregation */
  const int32_t* src;
  const int32_t* eid;
  const int32_t* etype_slot;         /* NULL unless the edge attribute is (type table, type id) */
  const int32_t* rowptr_t;           /* by source: backward (may be NULL for a forward-only call) */
  const int32_t* dst_t;
  const int32_t* eid_t;
  const int32_t* etype_slot_t;
  const float* inv_deg;              /* [num_nodes] 1 / max(in-degree, 1) (backward) */
} stemgnn_graph_view;

/* One MySAGEConv + BatchNorm1d layer of Encoder (model/encoder.py:72-97,173,313-317). */
typedef struct stemgnn_sage_layer {
  int64_t in_dim, out_dim;
  const float* w_l;                  /* lin_l.weight [out, in] */
  const float* b_l;                  /* lin_l.bias [out] or NULL */
  const float* w_r;                  /* lin_r.weight [out, in] */
  const float* bn_weight;            /* [out]; NULL with normalize == 'none' */
  const float* bn_bias;
  float* bn_running_mean;            /* NULL: statistics not tracked */
  float* bn_running_var;
  int64_t* bn_num_batches_tracked;
  float bn_eps, bn_momentum;
  uint64_t drop_seed, drop_offset;   /* Philox key of this layer's dropout */
  float* g_w_l;                      /* gradients, backward only (NULL: not wanted) */
  float* g_b_l;
  float* g_w_r;
  float* g_bn_weight;
  float* g_bn_bias;
} stemgnn_sage_layer;

typedef struct stemgnn_encoder_cfg {
  int32_t num_layers;
  int32_t use_bn;                    /* normalize == 'batch' */
  int32_t training;                  /* batch statistics + dropout; 0: running statistics, no dropout (forward only) */
  int32_t act;                       /* 1: relu / leaky relu between layers */
  float negative_slope;
  float dropout_p;
  int64_t out_rows;                  /* forward only: z holds rows [0, out_rows) of the output (the last layer's
                                        product and normalisation are not written past them; its BatchNorm statistics
                                        still run over every row).  <= 0 or >= N: all rows */
  int32_t feature_kind;              /* STEMGNN_F32 (0) or STEMGNN_BF16 (1): how x and the outputs of the layers before
                                        the last one are STORED (BASELINE config 5: bf16 feature storage).  All
                                        arithmetic is fp32 either way; z, the aggregates, the pre-normalisation
                                        products and every gradient stay fp32; the input takes no gradient. */
} stemgnn_encoder_cfg;

/* Encoder.forward (model/encoder.py:279-323) for the 'sage' backbone without MoE layers: per layer K1 over the
 * rows that can receive edges, lin_l(agg) + lin_r(h) with the BatchNorm statistics in its epilogue, statistics
 * finalisation, normalise + activation + dropout (none after the last layer).  Exactly one of edge_attr (dense
 * [E, D]) / etab ([T, D], with graph->etype_slot) may be given, or neither.  The graph must not need the heavy-row
 * split (no row with more than 128 edges).  `save` keeps what the backward needs (stemgnn_encoder_save_bytes);
 * z [N (or cfg->out_rows), out_dim of the last layer]. */
size_t stemgnn_encoder_save_bytes(int64_t num_nodes, int64_t active_rows, const stemgnn_sage_layer* layers,
                                  const stemgnn_encoder_cfg* cfg);
int stemgnn_encoder_fwd(const stemgnn_graph_view* graph, const void* x, const float* edge_attr, const float* etab,
                        int64_t num_types, const stemgnn_sage_layer* layers, const stemgnn_encoder_cfg* cfg, float* z,
                        void* save, size_t save_bytes, void* stream);
/* Its backward: g_z [N, out] -> parameter gradients (layer structs) and, when g_x != NULL, the input gradient.
 * g_z is only read.  `scratch`: stemgnn_encoder_bwd_scratch_bytes. */
size_t stemgnn_encoder_bwd_scratch_bytes(int64_t num_nodes, int64_t active_rows, const stemgnn_sage_layer* layers,
                                         const stemgnn_encoder_cfg* cfg);
int stemgnn_encoder_bwd(const stemgnn_graph_view* graph, const void* x, const float* edge_attr, const float* etab,
                        int64_t num_types, const stemgnn_sage_layer* layers, const stemgnn_encoder_cfg* cfg,
                        const float* g_z, float* g_x, const void* save, size_t save_bytes, void* scratch,
                        size_t scratch_bytes, void* stream);

/* VectorQuantize (model/vq.py:692-1064) with a cosine codebook per head and projections in / out. */
typedef struct stemgnn_vq_params {
  int64_t dim, heads, code_dim, codebook_size;
  const float* w_in;                 /* project_in.weight [H*Dc, dim] */
  const float* b_in;                 /* [H*Dc] or NULL */
  const float* w_out;                /* project_out.weight [dim, H*Dc] */
  const float* b_out;                /* [dim] or NULL */
  const float* embed;                /* _codebook.embed [H, K, Dc] */
  float commitment_weight;           /* 0: no commitment term */
  float ortho_weight;                /* 0: no orthogonal regulariser */
  const int64_t* ortho_ids;          /* code ids the regulariser runs on (distinct) */
  int64_t num_ortho_ids;
  float* g_w_in;                     /* gradients, backward only (NULL: not wanted) */
  float* g_b_in;
  float* g_w_out;
  float* g_b_out;
  float* g_embed;
} stemgnn_vq_params;

/* VectorQuantize.forward for callers that use (quantize, embed_ind, loss) and not the per-head codes (the pretraining
 * step, pt_model.py:113): project_in -> fused cosine assignment -> loss terms -> project_out read off the projected
 * code table.  quantize [N, dim], ind [N, H] int64, loss [1] = commitment_weight * mse + ortho_weight * ortho
 * (zero in eval mode).  `save`: stemgnn_vq_save_bytes. */
size_t stemgnn_vq_save_bytes(const stemgnn_vq_params* p, int64_t num_rows);
int stemgnn_vq_fwd(const stemgnn_vq_params* p, const float* z, int64_t num_rows, int training, float* quantize,
                   int64_t* ind, float* loss, void* save, size_t save_bytes, void* stream);
/* Backward: g_quantize [N, dim] (NULL: zero), g_loss [1] device scalar (NULL: zero) -> g_z [N, dim] and the parameter
 * gradients of the struct.  `scratch`: stemgnn_vq_bwd_scratch_bytes. */
size_t stemgnn_vq_bwd_scratch_bytes(const stemgnn_vq_params* p, int64_t num_rows);
int stemgnn_vq_bwd(const stemgnn_vq_params* p, const float* z, int64_t num_rows, const int64_t* ind,
                   const float* g_quantize, const float* g_loss, float* g_z, const void* save, size_t save_bytes,
                   void* scratch, size_t scratch_bytes, void* stream);

/* The decoders that read the decoder query in PretrainModel.forward (model/pt_model.py:39-102,128-131). */
typedef struct stemgnn_heads_params {
  int64_t dim;                       /* hidden width D of the query */
  int64_t in_dim;                    /* out_features of feat_recon_decoder (= feature width of x) */
  const float* w_feat;               /* feat_recon_decoder [in_dim, D] */
  const float* b_feat;
  const float* w_topo;               /* topo_recon_decoder.lin [D, D] (InnerProductDecoder, encoder.py:353-366) */
  const float* b_topo;
  const float* w_ts;                 /* topo_sem_recon_decoder [D, 2D] */
  const float* b_ts;
  const float* w_sem;                /* sem_projector [D, D] */
  const float* b_sem;
  float* g_w_feat;                   /* gradients (backward only) */
  float* g_b_feat;
  float* g_w_topo;
  float* g_b_topo;
  float* g_w_ts;
  float* g_b_ts;
  float* g_w_sem;
  float* g_b_sem;
} stemgnn_heads_params;

/* The four reconstruction losses of one step, losses[4] = (feat_recon, topo_recon, topo_sem_recon, sem_recon), each
 * UNWEIGHTED (the caller applies the lambdas).  k = number of sampled edges (max(int(E * ratio), 1)); the draws are
 * keyed by (seed, off_*) like stemgnn_sample_edges / stemgnn_negative_sample and returned for replay: topo_perm [k],
 * topo_edges int64 [2, 2k] (columns [0, k) the sampled positives, [k, 2k) the negatives), ts_perm [k], ts_edges
 * [2, k], ts_type [k].  edge_index int64 [2, E] / edge_type int64 [E] are the batch graph's COO; `graph` its
 * by-target CSR (negative sampling looks positives up through it).  x_feat [N, in_dim], z_teacher [>= bs, D].
 * Everything is enqueued on `stream` (with STEMGNN_HEADS_LANES=1 the three heads' chains run on internal side
 * streams forked from and joined to it). */
size_t stemgnn_heads_save_bytes(const stemgnn_heads_params* p, int64_t num_nodes, int64_t num_edges, int64_t bs, int64_t k);
int stemgnn_heads_fwd(const stemgnn_heads_params* p, const stemgnn_graph_view* graph, const int64_t* edge_index,
                      const int64_t* edge_type, int64_t num_edges, const float* etab, int64_t num_types, const float* q,
                      const float* x_feat, const float* z_teacher, int64_t bs, int64_t k, uint64_t seed,
                      uint64_t off_topo, uint64_t off_neg, uint64_t off_ts, int64_t* topo_perm, int64_t* topo_edges,
                      int64_t* ts_perm, int64_t* ts_edges, int64_t* ts_type, float* losses, void* save,
                      size_t save_bytes, void* stream);
/* Backward: g_losses [4] device scalars -> g_q [N, D] (overwritten) and the parameter gradients of the struct. */
size_t stemgnn_heads_bwd_scratch_bytes(const stemgnn_heads_params* p, int64_t num_nodes, int64_t bs, int64_t k);
int stemgnn_heads_bwd(const stemgnn_heads_params* p, int64_t num_nodes, const float* q, const float* x_feat,
                      const float* z_teacher, int64_t bs, int64_t k, const int64_t* topo_edges, const int64_t* ts_edges,
                      const float* g_losses, float* g_q, const void* save, size_t save_bytes, int64_t num_edges,
                      void* scratch, size_t scratch_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STEMGNN_H_ */
