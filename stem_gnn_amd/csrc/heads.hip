// Decoder / loss block of PretrainModel.forward (reference STEM-GNN/model/pt_model.py:39-102,128-131) as one call per
// direction: the four reconstruction heads that read the decoder query (quantize, or z with no_codebook):
//   feature      mse(feat_recon_decoder(q[:bs]), x[:bs])                                   pt_model.py:42-43
//   topology     BCE of sigmoid(<lin(q)_u, lin(q)_v>) on 10 % sampled edges + as many negatives   :51-65, encoder.py:353-366
//   topo-sem     mse(topo_sem_recon_decoder(cat(q_u, q_v)), edge_attr) on 10 % sampled edges  :72-81
//   semantic     mean(1 - cos(teacher[:bs], sem_projector(q[:bs])))                         :93-100
// The heads are independent chains of small, latency-bound launches (1 024-row and 11 k-row products, one-block
// reductions).  One call enqueues all of them on the caller's stream (the arithmetic is that of the single-op entry
// points).  Forking the three chains onto side streams was measured in round 2 (2.04-2.09 vs 1.97 ms per C4 step: the
// cross-stream event waits cost more than the overlap of these short chains returns) and is gone from the library.
#include "common.h"

#include <algorithm>

namespace stemgnn {
namespace {

constexpr int kBlock = 256;

inline size_t a256(size_t v) { return align_up(v, 256); }

struct Carver {
  uintptr_t base;
  size_t off = 0;
  explicit Carver(const void* p) : base(reinterpret_cast<uintptr_t>(p)) { off = a256(base) - base; }
  template <typename T>
  T* take(size_t count) {
    T* p = reinterpret_cast<T*>(base + off);
    off += a256(count * sizeof(T));
    return p;
  }
};

#define STEMGNN_TRY(expr)                 \
  do {                                    \
    const int rc__ = (expr);              \
    if (rc__ != STEMGNN_OK) return rc__;  \
  } while (0)

__global__ void __launch_bounds__(kBlock) k_zero16(uint4* __restrict__ p, int64_t n16) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n16;
       i += static_cast<int64_t>(gridDim.x) * kBlock)
    p[i] = make_uint4(0u, 0u, 0u, 0u);
}

int zero_bytes(void* p, size_t bytes, hipStream_t st) {  // bytes rounded up to 16: the buffers here are 256-aligned carves
  const int64_t n16 = static_cast<int64_t>((bytes + 15) / 16);
  if (n16 == 0) return STEMGNN_OK;
  int64_t g = (n16 + kBlock - 1) / kBlock;
  if (g > 2048) g = 2048;
  k_zero16<<<static_cast<unsigned>(g), kBlock, 0, st>>>(reinterpret_cast<uint4*>(p), n16);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

// g[r] += a[r] + b[r] for the leading rows (the two [:bs] consumers of the query)
__global__ void __launch_bounds__(kBlock) k_add2(float* __restrict__ g, const float* __restrict__ a,
                                                 const float* __restrict__ b, int64_t n4) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n4;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    float4 v = ld4(g + 4 * i);
    const float4 x = ld4(a + 4 * i), y = ld4(b + 4 * i);
    v.x += x.x + y.x; v.y += x.y + y.y; v.z += x.z + y.z; v.w += x.w + y.w;
    st4(g + 4 * i, v);
  }
}

struct HeadsSave {
  float *zl, *dots, *coef, *zz, *h_ts, *target, *h_f, *h_s, *cos_save;
  uint8_t* selected;
  void* ws[3];
  size_t ws_bytes;
  void* bce_ws;
  size_t bce_bytes;
  size_t bytes;
};

inline bool heads_ok(const stemgnn_heads_params* p) {
  return p && p->dim > 0 && p->dim % 4 == 0 && p->in_dim > 0 && p->in_dim % 4 == 0 && p->w_feat && p->w_topo && p->w_ts &&
         p->w_sem;
}

inline HeadsSave plan_heads(const void* save, const stemgnn_heads_params* p, int64_t N, int64_t E, int64_t bs, int64_t k) {
  HeadsSave s{};
  Carver c(save);
  const size_t n1 = std::max<int64_t>(N, 1), k1 = std::max<int64_t>(k, 1), b1 = std::max<int64_t>(bs, 1);
  const int64_t D = p->dim;
  s.zl = c.take<float>(n1 * D);
  s.dots = c.take<float>(2 * k1);
  s.coef = c.take<float>(2 * k1);
  s.zz = c.take<float>(k1 * 2 * D);
  s.h_ts = c.take<float>(k1 * D);
  s.target = c.take<float>(k1 * D);
  s.h_f = c.take<float>(b1 * p->in_dim);
  s.h_s = c.take<float>(b1 * D);
  s.cos_save = c.take<float>(b1 * 3);
  s.selected = c.take<uint8_t>(std::max<int64_t>(E, 1) + 16);
  s.ws_bytes = stemgnn_loss_workspace_bytes(std::max<int64_t>(bs, 256));
  for (int i = 0; i < 3; ++i) s.ws[i] = c.take<unsigned char>(s.ws_bytes);
  s.bce_bytes = stemgnn_edge_dot_bce_workspace_bytes(2 * k1);
  s.bce_ws = c.take<unsigned char>(s.bce_bytes);
  s.bytes = c.off + 256;
  return s;
}

}  // namespace
}  // namespace stemgnn

using namespace stemgnn;

extern "C" {

size_t stemgnn_heads_save_bytes(const stemgnn_heads_params* p, int64_t N, int64_t E, int64_t bs, int64_t k) {
  if (!heads_ok(p) || N < 0 || E < 0 || bs < 0 || k < 0) return 0;
  return plan_heads(nullptr, p, N, E, bs, k).bytes;
}

int stemgnn_heads_fwd(const stemgnn_heads_params* p, const stemgnn_graph_view* g, const int64_t* edge_index,
                      const int64_t* edge_type, int64_t E, const float* etab, int64_t T, const float* q,
                      const float* x_feat, const float* z_teacher, int64_t bs, int64_t k, uint64_t seed,
                      uint64_t off_topo, uint64_t off_neg, uint64_t off_ts, int64_t* topo_perm, int64_t* topo_edges,
                      int64_t* ts_perm, int64_t* ts_edges, int64_t* ts_type, float* losses, void* save,
                      size_t save_bytes, void* stream) {
  if (!heads_ok(p) || !g || E <= 0 || k <= 0 || k > E || bs <= 0 || T <= 0) return STEMGNN_ERR_INVALID_ARG;
  const int64_t N = g->num_nodes, D = p->dim;
  if (N < 2 || bs > N) return STEMGNN_ERR_INVALID_ARG;
  if (!edge_index || !edge_type || !etab || !q || !x_feat || !z_teacher || !topo_perm || !topo_edges || !ts_perm ||
      !ts_edges || !ts_type || !losses || !save || !g->rowptr || !g->src || !g->eid)
    return STEMGNN_ERR_INVALID_ARG;
  if (save_bytes < stemgnn_heads_save_bytes(p, N, E, bs, k)) return STEMGNN_ERR_WORKSPACE;
  hipStream_t s0 = static_cast<hipStream_t>(stream);
  const HeadsSave s = plan_heads(save, p, N, E, bs, k);

  // ---- topology head.  Sampled positives into the left half of the [2, 2k] edge buffer and flagged, negatives that
  // avoid them into the right half (pt_model.py:53-60); edge scores on lin(q); BCE.  Both edge samples of the phase
  // (this head's and the topo-sem head's) come from one launch.
  STEMGNN_TRY(stemgnn_sample_edges2(edge_index, edge_type, E, seed, k, off_topo, topo_perm, topo_edges, 2 * k, s.selected,
                                    k, off_ts, ts_perm, ts_edges, k, ts_type, s0));
  STEMGNN_TRY(stemgnn_negative_sample_into(g->rowptr, g->src, g->eid, s.selected, N, k, seed, off_neg, topo_edges + k,
                                           2 * k, s0));
  STEMGNN_TRY(stemgnn_linear_fwd(q, p->w_topo, D, nullptr, nullptr, 0, p->b_topo, N, D, s.zl, nullptr, nullptr, -1, s0));
  // edge scores and the BCE terms / coefficients in one launch
  STEMGNN_TRY(stemgnn_edge_dot_bce(s.zl, N, D, topo_edges, k, k, losses + 1, s.coef, s.bce_ws, s.bce_bytes, s0));

  // The three products over few rows (sampled pairs, seed rows) share one launch (DirectBatch, csrc/wsgemm.hip): on
  // their own they are 8 - 90 tiles on 256 CUs, 12 - 20 us each.
  const bool direct = stemgnn_linear_set_mode(-1) == 1 && linear_direct_ok(k, D, 2 * D) &&
                      linear_direct_ok(bs, p->in_dim, D) && linear_direct_ok(bs, D, D);
  DirectBatch small;

  // ---- topo-sem head.  The target rows are the type table's; cat(q_u, q_v) -> Linear -> mse (pt_model.py:72-81)
  STEMGNN_TRY(stemgnn_edge_concat_gather(q, N, D, ts_edges, k, s.zz, etab, T, ts_type, s.target, s0));
  if (direct) {
    STEMGNN_TRY(small.add(s.zz, p->w_ts, p->b_ts, k, D, 2 * D, s.h_ts, false, s0));
    STEMGNN_TRY(small.add(q, p->w_feat, p->b_feat, bs, p->in_dim, D, s.h_f, false, s0));
    STEMGNN_TRY(small.add(q, p->w_sem, p->b_sem, bs, D, D, s.h_s, false, s0));
    STEMGNN_TRY(small.flush(s0));
  } else {
    STEMGNN_TRY(stemgnn_linear_fwd(s.zz, p->w_ts, 2 * D, nullptr, nullptr, 0, p->b_ts, k, D, s.h_ts, nullptr, nullptr, -1, s0));
    // ---- the two heads on the seed rows q[:bs] (pt_model.py:42-43, 93-100)
    STEMGNN_TRY(stemgnn_linear_fwd(q, p->w_feat, D, nullptr, nullptr, 0, p->b_feat, bs, p->in_dim, s.h_f, nullptr, nullptr,
                                   -1, s0));
    STEMGNN_TRY(stemgnn_linear_fwd(q, p->w_sem, D, nullptr, nullptr, 0, p->b_sem, bs, D, s.h_s, nullptr, nullptr, -1, s0));
  }
  // the three losses of these heads in one launch (csrc/loss_ops.hip: same sums, same order as the single-op calls)
  HeadLossJobs lj{};
  lj.pred_a = s.h_ts; lj.tgt_a = s.target; lj.n_a = k * D; lj.loss_a = losses + 2; lj.ws_a = s.ws[0];
  lj.pred_b = s.h_f; lj.tgt_b = x_feat; lj.n_b = bs * p->in_dim; lj.loss_b = losses + 0; lj.ws_b = s.ws[1];
  lj.z = z_teacher; lj.h = s.h_s; lj.rows = bs; lj.D = D; lj.loss_c = losses + 3; lj.cos_save = s.cos_save;
  lj.ws_c = s.ws[2];
  STEMGNN_TRY(head_losses_fwd(lj, s0));
  return STEMGNN_OK;
}

size_t stemgnn_heads_bwd_scratch_bytes(const stemgnn_heads_params* p, int64_t N, int64_t bs, int64_t k) {
  if (!heads_ok(p) || N < 0 || bs < 0 || k < 0) return 0;
  const size_t n1 = std::max<int64_t>(N, 1), k1 = std::max<int64_t>(k, 1), b1 = std::max<int64_t>(bs, 1);
  const int64_t D = p->dim, I = p->in_dim;
  return a256(n1 * D * 4) + a256(k1 * D * 4) + a256(k1 * 2 * D * 4) + a256(b1 * I * 4) + 3 * a256(b1 * D * 4) +
         a256(stemgnn_linear_bwd_weight_workspace_bytes(N, D, D)) + a256(stemgnn_linear_bwd_weight_workspace_bytes(k, D, 2 * D)) +
         a256(stemgnn_linear_bwd_weight_workspace_bytes(bs, I, D)) + a256(stemgnn_linear_bwd_weight_workspace_bytes(bs, D, D)) +
         a256(stemgnn_edge_det_workspace_bytes(N, 2 * k)) + 2048;
}

int stemgnn_heads_bwd(const stemgnn_heads_params* p, int64_t N, const float* q, const float* x_feat,
                      const float* z_teacher, int64_t bs, int64_t k, const int64_t* topo_edges, const int64_t* ts_edges,
                      const float* g_losses, float* g_q, const void* save, size_t save_bytes, int64_t E, void* scratch,
                      size_t scratch_bytes, void* stream) {
  if (!heads_ok(p) || N < 2 || bs <= 0 || bs > N || k <= 0 || E < k) return STEMGNN_ERR_INVALID_ARG;
  if (!q || !x_feat || !z_teacher || !topo_edges || !ts_edges || !g_losses || !g_q || !save || !scratch)
    return STEMGNN_ERR_INVALID_ARG;
  if (!p->g_w_feat || !p->g_w_topo || !p->g_w_ts || !p->g_w_sem) return STEMGNN_ERR_INVALID_ARG;
  if (save_bytes < stemgnn_heads_save_bytes(p, N, E, bs, k) || scratch_bytes < stemgnn_heads_bwd_scratch_bytes(p, N, bs, k))
    return STEMGNN_ERR_WORKSPACE;
  const int64_t D = p->dim, I = p->in_dim;
  hipStream_t s0 = static_cast<hipStream_t>(stream);
  const HeadsSave s = plan_heads(save, p, N, E, bs, k);
  BtScope plane_scope(static_cast<hipStream_t>(stream));  // a head's output gradient: one cut for its two backward products
  Carver c(scratch);
  float* g_zl = c.take<float>(static_cast<size_t>(N) * D);
  float* g_hts = c.take<float>(static_cast<size_t>(k) * D);
  float* g_zz = c.take<float>(static_cast<size_t>(k) * 2 * D);
  float* g_hf = c.take<float>(static_cast<size_t>(bs) * I);
  float* g_hs = c.take<float>(static_cast<size_t>(bs) * D);
  float* g_head_f = c.take<float>(static_cast<size_t>(bs) * D);
  float* g_head_s = c.take<float>(static_cast<size_t>(bs) * D);
  const size_t wb_t = stemgnn_linear_bwd_weight_workspace_bytes(N, D, D);
  const size_t wb_ts = stemgnn_linear_bwd_weight_workspace_bytes(k, D, 2 * D);
  const size_t wb_f = stemgnn_linear_bwd_weight_workspace_bytes(bs, I, D);
  const size_t wb_s = stemgnn_linear_bwd_weight_workspace_bytes(bs, D, D);
  void* ws_t = c.take<unsigned char>(wb_t);
  void* ws_ts = c.take<unsigned char>(wb_ts);
  void* ws_f = c.take<unsigned char>(wb_f);
  void* ws_s = c.take<unsigned char>(wb_s);
  const size_t det_bytes = stemgnn_edge_det_workspace_bytes(N, 2 * k);
  void* det_ws = c.take<unsigned char>(det_bytes);
  // STEMGNN_DETERMINISTIC=1 / stemgnn_set_deterministic(1): the two scatters over sampled edges add in a fixed order
  // (edges grouped by node, two sorts each) instead of with fp32 atomics -- bit-reproducible steps, ~0.15 ms slower on a
  // C4 batch (k = 11 k edges, D = 128).  Atomics run at ~1.3 TB/s of added bytes on this part, plain gathers at 5+: once
  // the scatters move hundreds of MB the sorted form is the FASTER one (C3: k = 231 k, D = 768: 3.9 ms of atomics ->
  // 1.9 ms, 76.0 -> 74.0 ms per step; at k * D = 8.6e6, the refdefault batch, the atomics still win by 0.2 ms), so it
  // is taken on its own from k * D >= 5e7 elements.
  const bool det = stemgnn_set_deterministic(-1) == 1 || static_cast<double>(k) * static_cast<double>(D) >= 5e7;
  bool zeroed = false;  // g_zl already cleared (by the loss-gradient launch)

  // the four weight gradients run as one split-product launch and one reduction at the end (DwBatch)
  DwBatch dws;
  const bool direct = stemgnn_linear_set_mode(-1) == 1 && linear_direct_ok(k, 2 * D, D) && linear_direct_ok(bs, D, I) &&
                      linear_direct_ok(bs, D, D);
  DirectBatch small;  // the three backward-data products over few rows, one launch (see stemgnn_heads_fwd)

  // ---- the three loss gradients in one launch
  HeadLossJobs lj{};
  lj.pred_a = s.h_ts; lj.tgt_a = s.target; lj.n_a = k * D; lj.g_a = g_losses + 2; lj.gp_a = g_hts;
  lj.pred_b = s.h_f; lj.tgt_b = x_feat; lj.n_b = bs * I; lj.g_b = g_losses + 0; lj.gp_b = g_hf;
  lj.z = z_teacher; lj.h = s.h_s; lj.rows = bs; lj.D = D; lj.g_c = g_losses + 3; lj.cos_save = s.cos_save; lj.gh = g_hs;
  if (!det) {  // the same launch clears the topology head's scatter target
    lj.zero_ptr = g_zl;
    lj.zero_bytes = (static_cast<int64_t>(N) * D * 4 + 15) / 16 * 16;
    zeroed = true;
  }
  STEMGNN_TRY(head_losses_bwd(lj, s0));
  // ---- topo-sem head back to the gathered pairs; the two seed-row heads
  STEMGNN_TRY(dws.add(g_hts, s.zz, kF32, k, D, 2 * D, p->g_w_ts, p->b_ts ? p->g_b_ts : nullptr, ws_ts, wb_ts, s0));
  STEMGNN_TRY(dws.add(g_hf, q, kF32, bs, I, D, p->g_w_feat, p->b_feat ? p->g_b_feat : nullptr, ws_f, wb_f, s0));
  STEMGNN_TRY(dws.add(g_hs, q, kF32, bs, D, D, p->g_w_sem, p->b_sem ? p->g_b_sem : nullptr, ws_s, wb_s, s0));
  if (direct) {
    // dx[M, K] = dy[M, N] w[N, K]: the weight as stored is the [contraction][output] layout
    STEMGNN_TRY(small.add(g_hts, p->w_ts, nullptr, k, 2 * D, D, g_zz, true, s0));
    STEMGNN_TRY(small.add(g_hf, p->w_feat, nullptr, bs, D, I, g_head_f, true, s0));
    STEMGNN_TRY(small.add(g_hs, p->w_sem, nullptr, bs, D, D, g_head_s, true, s0));
    STEMGNN_TRY(small.flush(s0));
  } else {
    STEMGNN_TRY(stemgnn_linear_bwd_data(g_hts, p->w_ts, k, D, 2 * D, g_zz, s0));
    STEMGNN_TRY(stemgnn_linear_bwd_data(g_hf, p->w_feat, bs, I, D, g_head_f, s0));
    STEMGNN_TRY(stemgnn_linear_bwd_data(g_hs, p->w_sem, bs, D, D, g_head_s, s0));
  }

  // ---- topology head; its backward-data product lays down the dense gradient the other heads add into
  if (det) {
    STEMGNN_TRY(stemgnn_edge_dot_bwd_det(s.coef, g_losses + 1, s.zl, N, D, topo_edges, 2 * k, g_zl, det_ws, det_bytes, s0));
  } else {
    if (!zeroed) STEMGNN_TRY(zero_bytes(g_zl, static_cast<size_t>(N) * D * 4, s0));
    STEMGNN_TRY(stemgnn_edge_dot_bwd_scaled(s.coef, g_losses + 1, s.zl, N, D, topo_edges, 2 * k, g_zl, s0));
  }
  STEMGNN_TRY(stemgnn_linear_bwd_data(g_zl, p->w_topo, N, D, D, g_q, s0));
  STEMGNN_TRY(dws.add(g_zl, q, kF32, N, D, D, p->g_w_topo, p->b_topo ? p->g_b_topo : nullptr, ws_t, wb_t, s0));
  if (!det) {
    // the scatter of the sampled pairs' gradients and the two seed-row heads' gradients, one launch
    STEMGNN_TRY(stemgnn_edge_concat_bwd_add(g_zz, N, D, ts_edges, k, g_q, g_head_f, g_head_s, bs * D, s0));
  } else {
    STEMGNN_TRY(stemgnn_edge_concat_bwd_det(g_zz, N, D, ts_edges, k, g_q, det_ws, det_bytes, s0));
    const int64_t n4 = bs * D / 4;
    int64_t grid = (n4 + kBlock - 1) / kBlock;
    if (grid > 2048) grid = 2048;
    k_add2<<<static_cast<unsigned>(grid), kBlock, 0, s0>>>(g_q, g_head_f, g_head_s, n4);
    STEMGNN_LAUNCH_CHECK();
  }
  return dws.flush(s0);
}

}  // extern "C"
