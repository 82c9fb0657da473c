// Augmented-graph construction and loss-edge sampling on the device, without host round trips.
//
// Reference: torch_geometric.utils.dropout_adj(edge_index, edge_attr, p, force_undirected=True)
// (STEM-GNN/pretrain.py:42-44) and torch_geometric.utils.negative_sampling (model/pt_model.py:60).
// PyG's dropout_adj builds a new COO with boolean-mask indexing (a device->host sync for the
// output size) that the aggregation then has to re-sort.  Here the survivors are selected
// straight out of the ORIGINAL graph's two CSR views, which already hold every edge grouped by
// target and by source in edge order:
//
//   valid(e) = keep(e) && src(e) <= dst(e)            keep(e) = philox(seed, offset)[e] >= p
//   augmented in-list of v  = [valid in-edges of v, sources]  ++ [valid out-edges of v, targets]
//   augmented out-list of u = [valid out-edges of u, targets] ++ [valid in-edges of u, sources]
//
// which is exactly PyG's [row;col] ++ [col;row] emission stably grouped by target / by source.
// Slots carry the ORIGINAL edge id, so the dense edge_attr of the original graph (or its edge
// types) serves the augmented graph without the cat([ea, ea]) copy.  count -> scan -> fill,
// all sizes stay on the device (arrays are allocated at the 2E upper bound).
#include "common.h"

#include <algorithm>

#include <cstring>

namespace stemgnn {
namespace {

constexpr int kThreads = 256;

__device__ inline bool keep_edge(uint64_t seed, uint64_t offset, float p, int e) {
  if (p <= 0.f) return true;
  uint32_t r[4];
  Philox::gen(seed, offset, static_cast<uint64_t>(e) >> 2, r);
  return Philox::to_unit(r[e & 3]) >= p;
}

// Exclusive scan of one value per thread over the block (256 threads): wave scans by shuffles, the four wave totals
// through LDS.  Returns the exclusive prefix; `total` = the block's sum (in every thread).
__device__ inline int block_exclusive_scan(int x, int& total) {
  __shared__ int s_wave[kThreads / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = x;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(inc, o);
    if (lane >= o) inc += y;
  }
  __syncthreads();  // s_wave may still be read by a previous call
  if (lane == 63) s_wave[w] = inc;
  __syncthreads();
  int before = 0;
  total = 0;
#pragma unroll
  for (int k = 0; k < kThreads / 64; ++k) {
    const int t = s_wave[k];
    if (k < w) before += t;
    total += t;
  }
  return before + inc - x;
}

// Both launches give a row to FOUR neighbouring lanes (lane j takes slots j, j + 4, ... of the row's two lists): a row
// walked by one thread is a chain of ~20 dependent reads and as many Philox blocks -- 11-14 us whether 11k or 102k rows
// were walked.  A block covers kRows = 64 rows.
constexpr int kRowLanes = 4, kRows = kThreads / kRowLanes;

// Launch 1 of 2: the survivors per node (rows [0, A): only those can keep an edge, see the entry point), each block's
// total, and -- by whichever block finishes last (common.h: ticket_last) -- the exclusive scan of the block totals and
// the grand total.  Launch 2 redoes the scan inside its block from the per-node counts.  This replaces count + a
// two-launch device scan + fill (4 launches, 36 us on the 102k-node batch) without a spin-wait between blocks.
__global__ void __launch_bounds__(kThreads)
k_aug_count(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src, const int32_t* __restrict__ eid,
            const int32_t* __restrict__ rowptr_t, const int32_t* __restrict__ dst_t,
            const int32_t* __restrict__ eid_t, int64_t A, float p, uint64_t seed, uint64_t offset,
            const uint8_t* __restrict__ keep_in, int32_t* __restrict__ cnt_a, int32_t* __restrict__ deg,
            int32_t* __restrict__ block_sum, int32_t* __restrict__ block_off, int32_t* __restrict__ total_out,
            unsigned int* __restrict__ counter) {
  const int sub = threadIdx.x & (kRowLanes - 1);
  const int64_t v = static_cast<int64_t>(blockIdx.x) * kRows + (threadIdx.x >> 2);
  int a = 0, b = 0;
  if (v < A) {
    const int s1 = rowptr[v + 1], t1 = rowptr_t[v + 1];
    for (int s = rowptr[v] + sub; s < s1; s += kRowLanes) {
      const int e = eid[s];
      const bool k = keep_in ? keep_in[e] != 0 : keep_edge(seed, offset, p, e);
      a += (k && src[s] <= v) ? 1 : 0;
    }
    for (int t = rowptr_t[v] + sub; t < t1; t += kRowLanes) {
      const int e = eid_t[t];
      const bool k = keep_in ? keep_in[e] != 0 : keep_edge(seed, offset, p, e);
      b += (k && v <= dst_t[t]) ? 1 : 0;
    }
  }
  a += __shfl_xor(a, 1);
  a += __shfl_xor(a, 2);
  b += __shfl_xor(b, 1);
  b += __shfl_xor(b, 2);
  if (v < A && sub == 0) {
    cnt_a[v] = a;
    deg[v] = a + b;
  }
  int total;
  (void)block_exclusive_scan(sub == 0 ? a + b : 0, total);
  if (threadIdx.x == 0) {
    st_agent(&block_sum[blockIdx.x], total);
    wait_stores();
  }
  if (!ticket_last(counter)) return;
  const int nb = static_cast<int>(gridDim.x);
  int carry = 0;
  for (int base = 0; base < nb; base += kThreads) {
    const int i = base + static_cast<int>(threadIdx.x);
    const int x = i < nb ? ld_agent(&block_sum[i]) : 0;
    int chunk;
    const int ex = block_exclusive_scan(x, chunk);
    if (i < nb) block_off[i] = carry + ex;
    carry += chunk;
  }
  if (threadIdx.x == 0) total_out[0] = carry;
}

// Launch 2 of 2, over ALL N nodes: row offsets (rows >= A are empty: they close at the total), 1 / degree, and the
// survivors of rows < A written to both augmented views.  The four lanes of a row look at four consecutive slots at a
// time; a survivor's position is the row's running count + the survivors among the lanes before it (one ballot).
__global__ void __launch_bounds__(kThreads)
k_aug_fill(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src, const int32_t* __restrict__ eid,
           const int32_t* __restrict__ etype_slot, const int32_t* __restrict__ rowptr_t,
           const int32_t* __restrict__ dst_t, const int32_t* __restrict__ eid_t,
           const int32_t* __restrict__ etype_slot_t, int64_t N, int64_t A, float p, uint64_t seed, uint64_t offset,
           const uint8_t* __restrict__ keep_in, const int32_t* __restrict__ cnt_a, const int32_t* __restrict__ deg,
           const int32_t* __restrict__ block_off, int32_t* __restrict__ arowptr /* [N+1]; [N] = total, already there */,
           int32_t* __restrict__ a_src, int32_t* __restrict__ a_eid, int32_t* __restrict__ a_type,
           int32_t* __restrict__ a_dst_t, int32_t* __restrict__ a_eid_t, int32_t* __restrict__ a_type_t,
           float* __restrict__ inv_deg) {
  const int lane = threadIdx.x & 63, sub = threadIdx.x & (kRowLanes - 1), shift = lane & ~(kRowLanes - 1);
  const int64_t block0 = static_cast<int64_t>(blockIdx.x) * kRows;
  const int64_t v = block0 + (threadIdx.x >> 2);
  const int d = v < A ? deg[v] : 0;
  int block_total;
  const int ex = block_exclusive_scan(sub == 0 ? d : 0, block_total);  // lane 0 of the row holds the row's offset
  const int base = (block0 < A ? block_off[blockIdx.x] : arowptr[N]) + __shfl(ex, shift);
  if (v < N && sub == 0) {
    arowptr[v] = base;
    inv_deg[v] = 1.0f / static_cast<float>(d < 1 ? 1 : d);
  }
  const bool live = v < A;
  const int na = live ? cnt_a[v] : 0;
  const int s0 = live ? rowptr[v] : 0, s1 = live ? rowptr[v + 1] : 0;
  const int t0 = live ? rowptr_t[v] : 0, t1 = live ? rowptr_t[v + 1] : 0;
  // in-edges of v (u -> v, u <= v): first in the by-target list, second in the by-source list
  int ia = 0;
  for (int s = s0 + sub; __any(s - sub < s1); s += kRowLanes) {
    int e = 0, u = 0, ty = 0;
    bool f = false;
    if (s < s1) {
      e = eid[s];
      u = src[s];
      ty = etype_slot ? etype_slot[s] : 0;
      f = (keep_in ? keep_in[e] != 0 : keep_edge(seed, offset, p, e)) && u <= v;
    }
    const unsigned four = static_cast<unsigned>(__ballot(f) >> shift) & 15u;
    if (f) {
      const int q = ia + __popc(four & ((1u << sub) - 1u));
      const int pd = base + q, ps = base + (d - na) + q;
      a_src[pd] = u; a_eid[pd] = e; if (a_type) a_type[pd] = ty;
      a_dst_t[ps] = u; a_eid_t[ps] = e; if (a_type_t) a_type_t[ps] = ty;
    }
    ia += __popc(four);
  }
  // out-edges of v (v -> w, v <= w): second in the by-target list, first in the by-source list
  int ib = 0;
  for (int t = t0 + sub; __any(t - sub < t1); t += kRowLanes) {
    int e = 0, w = 0, ty = 0;
    bool f = false;
    if (t < t1) {
      e = eid_t[t];
      w = dst_t[t];
      ty = etype_slot_t ? etype_slot_t[t] : 0;
      f = (keep_in ? keep_in[e] != 0 : keep_edge(seed, offset, p, e)) && v <= w;
    }
    const unsigned four = static_cast<unsigned>(__ballot(f) >> shift) & 15u;
    if (f) {
      const int q = ib + __popc(four & ((1u << sub) - 1u));
      const int pd = base + na + q, ps = base + q;
      a_src[pd] = w; a_eid[pd] = e; if (a_type) a_type[pd] = ty;
      a_dst_t[ps] = w; a_eid_t[ps] = e; if (a_type_t) a_type_t[ps] = ty;
    }
    ib += __popc(four);
  }
}

__global__ void __launch_bounds__(kThreads) k_zero_u8(uint8_t* __restrict__ p, int64_t n) {
  // 16 bytes per thread where the address allows it, single bytes at the ragged ends
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  const int64_t head = std::min<int64_t>(n, static_cast<int64_t>((16 - (a & 15)) & 15));
  const int64_t n16 = (n - head) / 16;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  uint4* q = reinterpret_cast<uint4*>(p + head);
  for (int64_t i = t; i < n16; i += stride) q[i] = make_uint4(0u, 0u, 0u, 0u);
  for (int64_t i = t; i < head; i += stride) p[i] = 0;
  for (int64_t i = head + 16 * n16 + t; i < n; i += stride) p[i] = 0;
}

// PyG negative_sampling (structured, sparse, directed): uniform over the N*(N-1) non-self-loop
// pairs, rejecting pairs that are SELECTED positive edges (selected[e] != 0).  Membership is
// looked up in the target's in-list of the original graph.  Each thread redraws until accepted
// (<= 64 tries; the acceptance probability is 1 - k/(N(N-1)) ~ 1).
__global__ void __launch_bounds__(kThreads)
k_negative_sample(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src,
                  const int32_t* __restrict__ eid, const uint8_t* __restrict__ selected, int64_t N, int64_t k,
                  uint64_t seed, uint64_t offset, int64_t* __restrict__ out /*[2][k], rows out_stride apart*/,
                  int64_t out_stride) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= k) return;
  const uint64_t population = static_cast<uint64_t>(N) * static_cast<uint64_t>(N - 1);
  int64_t r = 0, c = 1 % N;
  for (int attempt = 0; attempt < 64; ++attempt) {
    uint32_t u[4];
    // retries live in the counter block's high bits, not in the key: (seed, offset + 1) belongs to the next op
    Philox::gen(seed, offset, (static_cast<uint64_t>(attempt) << 40) | static_cast<uint64_t>(i), u);
    const uint64_t x = (static_cast<uint64_t>(u[0]) << 32 | u[1]) % population;
    r = static_cast<int64_t>(x / static_cast<uint64_t>(N - 1));
    c = static_cast<int64_t>(x % static_cast<uint64_t>(N - 1));
    if (r <= c) ++c;  // vector_to_edge_index: skip the diagonal
    bool hit = false;
    for (int s = rowptr[c]; s < rowptr[c + 1] && !hit; ++s) hit = (src[s] == r) && selected[eid[s]] != 0;
    if (!hit) break;
  }
  out[i] = r;
  out[out_stride + i] = c;
}

// Uniform k-subset of [0, n) without replacement = the first k outputs of a keyed pseudo-random
// PERMUTATION of [0, n): a 6-round Feistel network on ceil(log2 n) bits (rounded up to even),
// cycle-walked back into range (expected < 4 walks).  This is what randperm(n)[:k] is used for at
// reference model/pt_model.py:55-57,75-78 and model/vq.py:1024, without sorting n keys.
__device__ inline uint32_t feistel_round_fn(uint32_t v, uint32_t key) {
  uint32_t x = v * 0x9E3779B1u + key;
  x ^= x >> 15; x *= 0x85EBCA77u;
  x ^= x >> 13; x *= 0xC2B2AE3Du;
  x ^= x >> 16;
  return x;
}

__device__ inline int64_t feistel_pick(int64_t i, int64_t n, int half_bits, uint64_t seed, uint64_t offset) {
  uint32_t keys[8];
  Philox::gen(seed, offset, 0, *reinterpret_cast<uint32_t(*)[4]>(&keys[0]));
  Philox::gen(seed, offset, 1, *reinterpret_cast<uint32_t(*)[4]>(&keys[4]));
  const uint64_t mask = (1ull << half_bits) - 1;
  uint64_t x = static_cast<uint64_t>(i);
  do {
    uint64_t l = x >> half_bits, r = x & mask;
#pragma unroll
    for (int rd = 0; rd < 6; ++rd) {
      const uint64_t nl = r;
      r = l ^ (feistel_round_fn(static_cast<uint32_t>(r), keys[rd]) & mask);
      l = nl;
    }
    x = (l << half_bits) | r;
  } while (x >= static_cast<uint64_t>(n));
  return static_cast<int64_t>(x);
}

__global__ void __launch_bounds__(kThreads)
k_sample_subset(int64_t n, int64_t k, int half_bits, uint64_t seed, uint64_t offset, int64_t* __restrict__ out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= k) return;
  out[i] = feistel_pick(i, n, half_bits, seed, offset);
}

// The same k picks applied to an edge list in the launch that draws them: the picked columns of edge_index, the
// picked edge types and the membership mask the negative sampler needs -- instead of four indexing launches.
__global__ void __launch_bounds__(kThreads)
k_sample_edges(const int64_t* __restrict__ edge_index, const int64_t* __restrict__ edge_type, int64_t E, int64_t k,
               int half_bits, uint64_t seed, uint64_t offset, int64_t* __restrict__ perm,
               int64_t* __restrict__ sel_index, int64_t sel_stride, int64_t* __restrict__ sel_type,
               uint8_t* __restrict__ selected) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= k) return;
  const int64_t e = feistel_pick(i, E, half_bits, seed, offset);
  perm[i] = e;
  if (sel_index) {
    sel_index[i] = edge_index[e];
    sel_index[sel_stride + i] = edge_index[E + e];
  }
  if (sel_type) sel_type[i] = edge_type[e];
  if (selected) selected[e] = 1;
}

// two edge samples of the same graph in one launch (the heads phase: topology positives, marked in `selected`, and the
// topo-sem pairs with their types); blocks [0, blocks_a) serve the first
__global__ void __launch_bounds__(kThreads)
k_sample_edges2(const int64_t* __restrict__ edge_index, const int64_t* __restrict__ edge_type, int64_t E, int half_bits,
                uint64_t seed, int blocks_a, int64_t k_a, uint64_t offset_a, int64_t* __restrict__ perm_a,
                int64_t* __restrict__ sel_a, int64_t stride_a, uint8_t* __restrict__ selected_a, int64_t k_b,
                uint64_t offset_b, int64_t* __restrict__ perm_b, int64_t* __restrict__ sel_b, int64_t stride_b,
                int64_t* __restrict__ type_b) {
  const bool first = static_cast<int>(blockIdx.x) < blocks_a;
  const int64_t i = static_cast<int64_t>(first ? blockIdx.x : blockIdx.x - blocks_a) * kThreads + threadIdx.x;
  if (i >= (first ? k_a : k_b)) return;
  const int64_t e = feistel_pick(i, E, half_bits, seed, first ? offset_a : offset_b);
  if (first) {
    perm_a[i] = e;
    sel_a[i] = edge_index[e];
    sel_a[stride_a + i] = edge_index[E + e];
    if (selected_a) selected_a[e] = 1;
  } else {
    perm_b[i] = e;
    sel_b[i] = edge_index[e];
    sel_b[stride_b + i] = edge_index[E + e];
    if (type_b) type_b[i] = edge_type[e];
  }
}

// mask_feature(x, p, mode='col') (reference pretrain.py:41): out = x with column c zeroed when
// philox(seed, offset)[c] < p  (the keep mask of stemgnn_dropout_keep_mask(D, p, seed, offset)).
__global__ void __launch_bounds__(kThreads)
k_mask_columns(const float* __restrict__ x, int64_t N, int D, float p, uint64_t seed, uint64_t offset,
               float* __restrict__ out, int kind) {
  const int nvec = D / 4;
  const int64_t total = N * nvec;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
  const int64_t i0 = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (stride % nvec == 0) {
    // the mask is a function of the COLUMN, and with this stride a thread stays on one column group: one Philox block
    // per thread instead of one per element (the generator was most of the kernel's instructions)
    uint32_t r[4];
    Philox::gen(seed, offset, static_cast<uint64_t>(i0 % nvec), r);
    const bool z0 = Philox::to_unit(r[0]) < p, z1 = Philox::to_unit(r[1]) < p, z2 = Philox::to_unit(r[2]) < p,
               z3 = Philox::to_unit(r[3]) < p;
    int64_t i = i0;
    for (; i + stride < total; i += 2 * stride) {  // two rows in flight
      float4 v = ld4_kind(x, 4 * i, kind), w = ld4_kind(x, 4 * (i + stride), kind);
      if (z0) v.x = w.x = 0.f;
      if (z1) v.y = w.y = 0.f;
      if (z2) v.z = w.z = 0.f;
      if (z3) v.w = w.w = 0.f;
      st4_kind(out, 4 * i, kind, v);  // a bf16 value masked or kept is still a bf16 value: no rounding happens
      st4_kind(out, 4 * (i + stride), kind, w);
    }
    if (i < total) {
      float4 v = ld4_kind(x, 4 * i, kind);
      if (z0) v.x = 0.f;
      if (z1) v.y = 0.f;
      if (z2) v.z = 0.f;
      if (z3) v.w = 0.f;
      st4_kind(out, 4 * i, kind, v);
    }
    return;
  }
  for (int64_t i = i0; i < total; i += stride) {
    const int c = static_cast<int>(i % nvec);
    uint32_t r[4];
    Philox::gen(seed, offset, static_cast<uint64_t>(c), r);
    float4 v = ld4_kind(x, 4 * i, kind);
    if (Philox::to_unit(r[0]) < p) v.x = 0.f;
    if (Philox::to_unit(r[1]) < p) v.y = 0.f;
    if (Philox::to_unit(r[2]) < p) v.z = 0.f;
    if (Philox::to_unit(r[3]) < p) v.w = 0.f;
    st4_kind(out, 4 * i, kind, v);  // a bf16 value masked or kept is still a bf16 value: no rounding happens
  }
}

}  // namespace
}  // namespace stemgnn

using namespace stemgnn;

extern "C" {

size_t stemgnn_graph_dropout_workspace_bytes(int64_t N) {
  if (N < 0) return 0;
  const size_t blocks = static_cast<size_t>((N + kRows - 1) / kRows) + 1;
  return 2 * align_up(static_cast<size_t>(N + 1) * sizeof(int32_t), 256) + 2 * align_up(blocks * sizeof(int32_t), 256) + 512;
}

int stemgnn_graph_dropout_undirected_rows(const int32_t* rowptr, const int32_t* src, const int32_t* eid,
                                          const int32_t* etype_slot, const int32_t* rowptr_t, const int32_t* dst_t,
                                          const int32_t* eid_t, const int32_t* etype_slot_t, int64_t N, int64_t E,
                                          int64_t active_rows, float p, uint64_t seed, uint64_t offset,
                                          const uint8_t* keep, int32_t* a_rowptr, int32_t* a_src, int32_t* a_eid,
                                          int32_t* a_etype_slot, int32_t* a_dst_t, int32_t* a_eid_t,
                                          int32_t* a_etype_slot_t, float* a_inv_deg, void* workspace,
                                          size_t workspace_bytes, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (N < 0 || E < 0 || p < 0.f || p > 1.f || !a_rowptr) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(N) || !fits_i32(2 * E)) return STEMGNN_ERR_TOO_LARGE;
  if (N == 0) {
    STEMGNN_HIP_TRY(hipMemsetAsync(a_rowptr, 0, sizeof(int32_t), st));
    return STEMGNN_OK;
  }
  if (!rowptr || !rowptr_t || !a_inv_deg || !workspace) return STEMGNN_ERR_INVALID_ARG;
  if (E > 0 && (!src || !eid || !dst_t || !eid_t || !a_src || !a_eid || !a_dst_t || !a_eid_t))
    return STEMGNN_ERR_INVALID_ARG;
  if ((etype_slot == nullptr) != (etype_slot_t == nullptr)) return STEMGNN_ERR_INVALID_ARG;
  if (workspace_bytes < stemgnn_graph_dropout_workspace_bytes(N)) return STEMGNN_ERR_WORKSPACE;
  const int64_t A = (active_rows < 0 || active_rows > N) ? N : active_rows;
  uintptr_t base = align_up(reinterpret_cast<uintptr_t>(workspace), 256);
  const size_t arr = align_up(static_cast<size_t>(N + 1) * sizeof(int32_t), 256);
  const size_t blocks = static_cast<size_t>((N + kRows - 1) / kRows) + 1;
  int32_t* cnt_a = reinterpret_cast<int32_t*>(base);
  int32_t* deg = reinterpret_cast<int32_t*>(base + arr);
  int32_t* block_sum = reinterpret_cast<int32_t*>(base + 2 * arr);
  int32_t* block_off = reinterpret_cast<int32_t*>(base + 2 * arr + align_up(blocks * sizeof(int32_t), 256));
  if (A > 0) {
    unsigned int* counter = ticket_counter(st);
    if (!counter) return STEMGNN_ERR_HIP;
    k_aug_count<<<static_cast<unsigned>((A + kRows - 1) / kRows), kThreads, 0, st>>>(
        rowptr, src, eid, rowptr_t, dst_t, eid_t, A, p, seed, offset, keep, cnt_a, deg, block_sum, block_off, a_rowptr + N,
        counter);
    STEMGNN_LAUNCH_CHECK();
  } else {
    STEMGNN_HIP_TRY(hipMemsetAsync(a_rowptr + N, 0, sizeof(int32_t), st));
  }
  k_aug_fill<<<static_cast<unsigned>((N + kRows - 1) / kRows), kThreads, 0, st>>>(
      rowptr, src, eid, etype_slot, rowptr_t, dst_t, eid_t, etype_slot_t, N, A, p, seed, offset, keep, cnt_a, deg, block_off,
      a_rowptr, a_src, a_eid, etype_slot ? a_etype_slot : nullptr, a_dst_t, a_eid_t, etype_slot ? a_etype_slot_t : nullptr,
      a_inv_deg);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_graph_dropout_undirected(const int32_t* rowptr, const int32_t* src, const int32_t* eid,
                                     const int32_t* etype_slot, const int32_t* rowptr_t, const int32_t* dst_t,
                                     const int32_t* eid_t, const int32_t* etype_slot_t, int64_t N, int64_t E,
                                     float p, uint64_t seed, uint64_t offset, const uint8_t* keep,
                                     int32_t* a_rowptr, int32_t* a_src, int32_t* a_eid, int32_t* a_etype_slot,
                                     int32_t* a_dst_t, int32_t* a_eid_t, int32_t* a_etype_slot_t, float* a_inv_deg,
                                     void* workspace, size_t workspace_bytes, void* stream_) {
  return stemgnn_graph_dropout_undirected_rows(rowptr, src, eid, etype_slot, rowptr_t, dst_t, eid_t, etype_slot_t, N, E, N,
                                               p, seed, offset, keep, a_rowptr, a_src, a_eid, a_etype_slot, a_dst_t,
                                               a_eid_t, a_etype_slot_t, a_inv_deg, workspace, workspace_bytes, stream_);
}

int stemgnn_sample_subset(int64_t n, int64_t k, uint64_t seed, uint64_t offset, int64_t* out, void* stream_) {
  if (n < 0 || k < 0 || k > n) return STEMGNN_ERR_INVALID_ARG;
  if (k == 0) return STEMGNN_OK;
  if (!out || n >= (1ll << 60)) return STEMGNN_ERR_INVALID_ARG;
  int bits = 2;
  while ((1ll << bits) < n) ++bits;
  if (bits & 1) ++bits;
  k_sample_subset<<<static_cast<unsigned>((k + kThreads - 1) / kThreads), kThreads, 0,
                    static_cast<hipStream_t>(stream_)>>>(n, k, bits / 2, seed, offset, out);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_mask_columns(const float* x, int64_t N, int64_t D, float p, uint64_t seed, uint64_t offset, float* out,
                         void* stream_) {
  if (N < 0 || D <= 0 || D % 4 != 0 || p < 0.f || p > 1.f) return STEMGNN_ERR_INVALID_ARG;
  if (N == 0) return STEMGNN_OK;
  if (!x || !out) return STEMGNN_ERR_INVALID_ARG;
  int64_t g = (N * (D / 4) + kThreads - 1) / kThreads;
  if (g > 4096) g = 4096;
  {  // a grid stride that is a multiple of the row length keeps a thread on one column group (see the kernel)
    int64_t a = D / 4, b = kThreads;
    while (b) { const int64_t t = a % b; a = b; b = t; }
    const int64_t m = (D / 4) / a;
    if (g >= m) g = g / m * m;
  }
  k_mask_columns<<<static_cast<unsigned>(g), kThreads, 0, static_cast<hipStream_t>(stream_)>>>(
      x, N, static_cast<int>(D), p, seed, offset, out, kF32);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_mask_columns_k(const void* x, int32_t kind, int64_t N, int64_t D, float p, uint64_t seed, uint64_t offset,
                           void* out, void* stream_) {
  if (N < 0 || D <= 0 || D % 4 != 0 || p < 0.f || p > 1.f || (kind != kF32 && kind != kBF16)) return STEMGNN_ERR_INVALID_ARG;
  if (N == 0) return STEMGNN_OK;
  if (!x || !out) return STEMGNN_ERR_INVALID_ARG;
  int64_t g = (N * (D / 4) + kThreads - 1) / kThreads;
  if (g > 4096) g = 4096;
  {  // a grid stride that is a multiple of the row length keeps a thread on one column group (see the kernel)
    int64_t a = D / 4, b = kThreads;
    while (b) { const int64_t t = a % b; a = b; b = t; }
    const int64_t m = (D / 4) / a;
    if (g >= m) g = g / m * m;
  }
  k_mask_columns<<<static_cast<unsigned>(g), kThreads, 0, static_cast<hipStream_t>(stream_)>>>(
      static_cast<const float*>(x), N, static_cast<int>(D), p, seed, offset, static_cast<float*>(out), kind);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_sample_edges(const int64_t* edge_index, const int64_t* edge_type, int64_t E, int64_t k, uint64_t seed,
                         uint64_t offset, int64_t* perm, int64_t* sel_index, int64_t sel_stride, int64_t* sel_type,
                         uint8_t* selected, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (E < 0 || k < 0 || k > E || (sel_index && sel_stride < k) || (sel_type && !edge_type)) return STEMGNN_ERR_INVALID_ARG;
  if (selected && E > 0) {
    // a byte-count memset goes through the runtime's slow fill path (20 us for 112 KB); a plain kernel takes 2
    k_zero_u8<<<static_cast<unsigned>(std::min<int64_t>((E + kThreads * 16 - 1) / (kThreads * 16), 1024)), kThreads, 0, st>>>(
        selected, E);
    STEMGNN_LAUNCH_CHECK();
  }
  if (k == 0) return STEMGNN_OK;
  if (!edge_index || !perm || E >= (1ll << 60)) return STEMGNN_ERR_INVALID_ARG;
  int bits = 2;
  while ((1ll << bits) < E) ++bits;
  if (bits & 1) ++bits;
  k_sample_edges<<<static_cast<unsigned>((k + kThreads - 1) / kThreads), kThreads, 0, st>>>(
      edge_index, edge_type, E, k, bits / 2, seed, offset, perm, sel_index, sel_stride, sel_type, selected);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_sample_edges2(const int64_t* edge_index, const int64_t* edge_type, int64_t E, uint64_t seed, int64_t k_a,
                          uint64_t offset_a, int64_t* perm_a, int64_t* sel_index_a, int64_t sel_stride_a,
                          uint8_t* selected_a, int64_t k_b, uint64_t offset_b, int64_t* perm_b, int64_t* sel_index_b,
                          int64_t sel_stride_b, int64_t* sel_type_b, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (E <= 0 || k_a <= 0 || k_b <= 0 || k_a > E || k_b > E || sel_stride_a < k_a || sel_stride_b < k_b)
    return STEMGNN_ERR_INVALID_ARG;
  if (!edge_index || !perm_a || !perm_b || !sel_index_a || !sel_index_b || (sel_type_b && !edge_type) ||
      E >= (1ll << 60))
    return STEMGNN_ERR_INVALID_ARG;
  if (selected_a) {
    k_zero_u8<<<static_cast<unsigned>(std::min<int64_t>((E + kThreads * 16 - 1) / (kThreads * 16), 1024)), kThreads, 0, st>>>(
        selected_a, E);
    STEMGNN_LAUNCH_CHECK();
  }
  int bits = 2;
  while ((1ll << bits) < E) ++bits;
  if (bits & 1) ++bits;
  const int ba = static_cast<int>((k_a + kThreads - 1) / kThreads), bb = static_cast<int>((k_b + kThreads - 1) / kThreads);
  k_sample_edges2<<<static_cast<unsigned>(ba + bb), kThreads, 0, st>>>(edge_index, edge_type, E, bits / 2, seed, ba, k_a,
                                                                     offset_a, perm_a, sel_index_a, sel_stride_a,
                                                                     selected_a, k_b, offset_b, perm_b, sel_index_b,
                                                                     sel_stride_b, sel_type_b);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_negative_sample(const int32_t* rowptr, const int32_t* src, const int32_t* eid, const uint8_t* selected,
                            int64_t N, int64_t k, uint64_t seed, uint64_t offset, int64_t* out, void* stream_) {
  return stemgnn_negative_sample_into(rowptr, src, eid, selected, N, k, seed, offset, out, k, stream_);
}

int stemgnn_negative_sample_into(const int32_t* rowptr, const int32_t* src, const int32_t* eid,
                                 const uint8_t* selected, int64_t N, int64_t k, uint64_t seed, uint64_t offset,
                                 int64_t* out, int64_t out_stride, void* stream_) {
  if (N < 0 || k < 0 || out_stride < k) return STEMGNN_ERR_INVALID_ARG;
  if (k == 0) return STEMGNN_OK;
  if (N < 2 || !rowptr || !src || !eid || !selected || !out) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(N)) return STEMGNN_ERR_TOO_LARGE;
  k_negative_sample<<<static_cast<unsigned>((k + kThreads - 1) / kThreads), kThreads, 0,
                      static_cast<hipStream_t>(stream_)>>>(rowptr, src, eid, selected, N, k, seed, offset, out,
                                                                           out_stride);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

}  // extern "C"
