// Phase entry points: a whole module forward / backward of the pretraining path enqueued by ONE call.
//
// Reference: Encoder.forward (STEM-GNN/model/encoder.py:279-323), VectorQuantize.forward (model/vq.py:849-1064) and
// the decoder / loss block of PretrainModel.forward (model/pt_model.py:39-102,116-142), each of which the reference
// runs as dozens of framework ops.  The single-op entry points of this library (include/stemgnn.h) already fuse those
// into kernels; called one by one from Python autograd they cost more host time than device time at batch size
// (111 launches, ~2 ms of issue time per 2.3 ms step).  Here the launch sequence of a module lives on the C side:
// the host pays one crossing per module and direction, the kernels and their arithmetic are unchanged.
#include "common.h"

#include <algorithm>

namespace stemgnn {
namespace {

inline size_t a256(size_t v) { return align_up(v, 256); }

struct Carver {  // hands out 256-byte aligned pieces of a caller buffer; only counts when base == 0
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

struct LayerSave {
  float *agg, *y, *h, *mean, *rstd;
};

struct EncoderPlan {
  int64_t N, A, R;           // rows, rows that receive edges, rows of the output the caller wants
  bool tail_identity;        // last layer without BatchNorm: its product IS z
  std::vector<LayerSave> layers;
  float* stats_partial;      // forward scratch (column partials of the widest layer)
  float* eval_rstd;          // eval-mode 1/sqrt(running_var + eps) of the widest layer
  size_t bytes;
};

inline bool encoder_args_ok(const stemgnn_sage_layer* layers, const stemgnn_encoder_cfg* cfg) {
  if (!layers || !cfg || cfg->num_layers <= 0 || cfg->num_layers > 64) return false;
  for (int l = 0; l < cfg->num_layers; ++l) {
    const stemgnn_sage_layer& y = layers[l];
    if (y.in_dim <= 0 || y.out_dim <= 0 || y.in_dim % 4 || y.out_dim % 4 || !y.w_l || !y.w_r) return false;
    if (l > 0 && layers[l - 1].out_dim != y.in_dim) return false;
    if (cfg->use_bn && (!y.bn_weight || !y.bn_bias)) return false;
    if ((y.bn_running_mean == nullptr) != (y.bn_running_var == nullptr)) return false;
  }
  if (cfg->feature_kind != kF32 && cfg->feature_kind != kBF16) return false;
  return cfg->dropout_p >= 0.f && cfg->dropout_p < 1.f;
}

// Layout of the save buffer (identical in forward and backward: both derive it from the same arguments).
inline EncoderPlan plan_encoder(const void* save, int64_t N, int64_t A, const stemgnn_sage_layer* layers,
                                const stemgnn_encoder_cfg* cfg, float* z) {
  EncoderPlan p;
  p.N = N;
  p.A = A;
  p.R = (cfg->out_rows > 0 && cfg->out_rows < N) ? cfg->out_rows : N;
  const int L = cfg->num_layers;
  p.tail_identity = !cfg->use_bn;
  Carver c(save);
  int64_t widest = 0;
  for (int l = 0; l < L; ++l) {
    const stemgnn_sage_layer& y = layers[l];
    const bool last = l == L - 1;
    LayerSave s{};
    s.agg = c.take<float>(static_cast<size_t>(std::max<int64_t>(A, 1)) * y.in_dim);
    if (last && p.tail_identity) {
      s.y = z;  // no normalisation, activation or dropout after the last layer: the product is the output
    } else {
      s.y = c.take<float>(static_cast<size_t>(std::max<int64_t>(last ? p.R : N, 1)) * y.out_dim);
    }
    // layer outputs before the last one are stored in the feature kind (bf16: half the bytes); z is always fp32
    s.h = last ? z : c.take<float>(static_cast<size_t>(std::max<int64_t>(N, 1)) * y.out_dim / (cfg->feature_kind == kBF16 ? 2 : 1) + 4);
    if (cfg->use_bn) {
      s.mean = c.take<float>(y.out_dim);
      s.rstd = c.take<float>(y.out_dim);
    }
    p.layers.push_back(s);
    widest = std::max(widest, y.out_dim);
  }
  p.stats_partial = reinterpret_cast<float*>(c.take<unsigned char>(stemgnn_linear_stats_partial_bytes(N, widest)));
  p.eval_rstd = c.take<float>(widest);
  p.bytes = c.off + 256;
  return p;
}

__global__ void k_rstd_from_var(const float* __restrict__ var, int n, float eps, float* __restrict__ rstd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rstd[i] = 1.0f / sqrtf(var[i] + eps);
}

#define STEMGNN_TRY(expr)                 \
  do {                                    \
    const int rc__ = (expr);              \
    if (rc__ != STEMGNN_OK) return rc__;  \
  } while (0)

// out[0] = (a ? a[0] : 0) + (b ? b[0] : 0)
__global__ void k_scalar_add(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (a ? a[0] : 0.f) + (b ? b[0] : 0.f);
}

struct VqSave {
  float *xp, *norm, *table, *esq, *terms;  // terms[0] commitment, terms[1] orthogonal
  void* assign_ws;
  size_t assign_ws_bytes;
  void* loss_ws;
  size_t loss_ws_bytes;
  size_t bytes;
};

inline bool vq_params_ok(const stemgnn_vq_params* p) {
  return p && p->dim > 0 && p->dim % 4 == 0 && p->heads > 0 && p->code_dim > 0 && p->code_dim % 4 == 0 &&
         p->codebook_size > 0 && p->w_in && p->w_out && p->embed && p->commitment_weight >= 0.f &&
         (p->ortho_weight == 0.f || (p->ortho_ids && p->num_ortho_ids > 0));
}

inline VqSave plan_vq(const void* save, const stemgnn_vq_params* p, int64_t N) {
  VqSave s{};
  Carver c(save);
  const size_t n1 = static_cast<size_t>(std::max<int64_t>(N, 1));
  const int64_t HD = p->heads * p->code_dim;
  s.xp = c.take<float>(n1 * HD);
  s.norm = c.take<float>(n1 * p->heads);
  s.table = c.take<float>(static_cast<size_t>(p->heads) * p->codebook_size * p->dim);
  s.esq = c.take<float>(static_cast<size_t>(p->heads) * p->codebook_size);
  s.terms = c.take<float>(4);
  s.assign_ws_bytes = stemgnn_vq_workspace_bytes(N, p->heads, p->code_dim, p->codebook_size);
  s.assign_ws = c.take<unsigned char>(s.assign_ws_bytes);
  s.loss_ws_bytes = stemgnn_loss_workspace_bytes(std::max<int64_t>(p->heads * std::max<int64_t>(p->num_ortho_ids, 1), 256));
  s.loss_ws = c.take<unsigned char>(s.loss_ws_bytes);
  s.bytes = c.off + 256;
  return s;
}

}  // namespace
}  // namespace stemgnn

using namespace stemgnn;

extern "C" {

size_t stemgnn_encoder_save_bytes(int64_t N, int64_t A, const stemgnn_sage_layer* layers,
                                  const stemgnn_encoder_cfg* cfg) {
  if (N < 0 || A < 0 || A > N || !encoder_args_ok(layers, cfg)) return 0;
  return plan_encoder(nullptr, N, A, layers, cfg, nullptr).bytes;
}

int stemgnn_encoder_fwd(const stemgnn_graph_view* g, const void* x, const float* edge_attr, const float* etab,
                        int64_t T, const stemgnn_sage_layer* layers, const stemgnn_encoder_cfg* cfg, float* z, void* save,
                        size_t save_bytes, void* stream) {
  if (!g || !encoder_args_ok(layers, cfg)) return STEMGNN_ERR_INVALID_ARG;
  const int64_t N = g->num_nodes;
  const int64_t A = (g->active_rows < 0 || g->active_rows > N) ? N : g->active_rows;
  if (N < 0) return STEMGNN_ERR_INVALID_ARG;
  if (N == 0) return STEMGNN_OK;
  if (!x || !z || !save || !g->rowptr) return STEMGNN_ERR_INVALID_ARG;
  if (save_bytes < stemgnn_encoder_save_bytes(N, A, layers, cfg)) return STEMGNN_ERR_WORKSPACE;
  if (cfg->use_bn && cfg->training && N <= 1) return STEMGNN_ERR_INVALID_ARG;  // BatchNorm1d raises on one row
  hipStream_t st = static_cast<hipStream_t>(stream);
  const EncoderPlan p = plan_encoder(save, N, A, layers, cfg, z);
  const int L = cfg->num_layers;
  const int fk = cfg->feature_kind;
  const void* h = x;
  for (int l = 0; l < L; ++l) {
    const stemgnn_sage_layer& y = layers[l];
    const LayerSave& s = p.layers[l];
    const bool last = l == L - 1;
    // K1 over the rows that can receive edges; the product below skips the aggregate's half past them
    if (A > 0)
      STEMGNN_TRY(stemgnn_sage_agg_fwd_k(h, fk, A, y.in_dim, g->rowptr, g->src, g->eid, edge_attr, etab, g->etype_slot, T,
                                         s.agg, stream));
    const bool batch_stats = cfg->use_bn && cfg->training;
    const int64_t rows_out = last ? p.R : N;  // the last layer's values are wanted for the leading rows only
    STEMGNN_TRY(stemgnn_linear_fwd_rows_k(s.agg, y.w_l, y.in_dim, h, fk, y.w_r, y.in_dim, y.b_l, N, y.out_dim, s.y,
                                          batch_stats ? p.stats_partial : nullptr, nullptr, A, rows_out, stream));
    if (last && p.tail_identity) break;
    const float *mean = nullptr, *rstd = nullptr;
    if (batch_stats) {
      STEMGNN_TRY(stemgnn_bn_stats_from_partials(p.stats_partial, stemgnn_linear_stats_blocks(N, y.out_dim), N,
                                                 y.out_dim, y.bn_eps, s.mean, s.rstd, y.bn_running_mean,
                                                 y.bn_running_var, y.bn_momentum, y.bn_num_batches_tracked, stream));
      mean = s.mean;
      rstd = s.rstd;
    } else if (cfg->use_bn) {
      if (!y.bn_running_mean) return STEMGNN_ERR_INVALID_ARG;  // eval mode needs tracked statistics
      const int n = static_cast<int>(y.out_dim);
      k_rstd_from_var<<<(n + 255) / 256, 256, 0, st>>>(y.bn_running_var, n, y.bn_eps, p.eval_rstd);
      STEMGNN_LAUNCH_CHECK();
      mean = y.bn_running_mean;
      rstd = p.eval_rstd;
    }
    const int act = last ? 0 : cfg->act;
    const float pdrop = (last || !cfg->training) ? 0.f : cfg->dropout_p;
    STEMGNN_TRY(stemgnn_bn_act_drop_fwd_k(s.y, rows_out, y.out_dim, mean, rstd, cfg->use_bn ? y.bn_weight : nullptr,
                                          cfg->use_bn ? y.bn_bias : nullptr, act, cfg->negative_slope, pdrop,
                                          y.drop_seed, y.drop_offset, s.h, last ? kF32 : fk, stream));
    h = s.h;
  }
  return STEMGNN_OK;
}

size_t stemgnn_encoder_bwd_scratch_bytes(int64_t N, int64_t A, const stemgnn_sage_layer* layers,
                                         const stemgnn_encoder_cfg* cfg) {
  if (N < 0 || A < 0 || A > N || !encoder_args_ok(layers, cfg)) return 0;
  int64_t wi = 0, wo = 0;
  size_t dw = 0;
  for (int l = 0; l < cfg->num_layers; ++l) {
    wi = std::max(wi, layers[l].in_dim);
    wo = std::max(wo, layers[l].out_dim);
    // every layer's weight gradients keep their slabs until the one reduction at the end of the phase
    dw += a256(stemgnn_linear_bwd_weight_workspace_bytes(N, layers[l].out_dim, layers[l].in_dim)) +
          a256(stemgnn_linear_bwd_weight_workspace_bytes(std::max<int64_t>(A, 1), layers[l].out_dim, layers[l].in_dim));
  }
  const size_t n1 = static_cast<size_t>(std::max<int64_t>(N, 1)), a1 = static_cast<size_t>(std::max<int64_t>(A, 1));
  // ... and every layer its own pre-activation gradient (an operand of those products)
  return cfg->num_layers * a256(n1 * wo * 4) + 2 * a256(n1 * wi * 4) + a256(a1 * wi * 4) + dw +
         a256(stemgnn_bn_workspace_bytes(N, wo)) + 1024;
}

int stemgnn_encoder_bwd(const stemgnn_graph_view* g, const void* x, const float* edge_attr, const float* etab,
                        int64_t T, const stemgnn_sage_layer* layers, const stemgnn_encoder_cfg* cfg, const float* g_z,
                        float* g_x, const void* save, size_t save_bytes, void* scratch, size_t scratch_bytes,
                        void* stream) {
  if (!g || !encoder_args_ok(layers, cfg)) return STEMGNN_ERR_INVALID_ARG;
  const int64_t N = g->num_nodes;
  const int64_t A = (g->active_rows < 0 || g->active_rows > N) ? N : g->active_rows;
  if (N <= 0) return N == 0 ? STEMGNN_OK : STEMGNN_ERR_INVALID_ARG;
  if (!cfg->training && cfg->use_bn) return STEMGNN_ERR_INVALID_ARG;  // running-statistics backward: single-op path
  if (cfg->out_rows > 0 && cfg->out_rows < N) return STEMGNN_ERR_INVALID_ARG;  // a truncated forward has no backward
  if (cfg->feature_kind == kBF16 && g_x) return STEMGNN_ERR_INVALID_ARG;         // stored features take no gradient
  if (!x || !g_z || !save || !scratch || !g->rowptr_t || !g->inv_deg) return STEMGNN_ERR_INVALID_ARG;
  if (save_bytes < stemgnn_encoder_save_bytes(N, A, layers, cfg) ||
      scratch_bytes < stemgnn_encoder_bwd_scratch_bytes(N, A, layers, cfg))
    return STEMGNN_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // large products on the big-tile core share their operands' planes inside this phase: a layer's pre-activation gradient
  // is cut once for its two backward-data products and its two weight gradients (operands are not rewritten below)
  BtScope plane_scope(st);
  const int L = cfg->num_layers;
  const EncoderPlan p = plan_encoder(save, N, A, layers, cfg, nullptr);
  int64_t wi = 0, wo = 0;
  for (int l = 0; l < L; ++l) {
    wi = std::max(wi, layers[l].in_dim);
    wo = std::max(wo, layers[l].out_dim);
  }
  Carver c(scratch);
  float* g_h_buf[2] = {c.take<float>(static_cast<size_t>(N) * wi), c.take<float>(static_cast<size_t>(N) * wi)};
  float* g_agg = c.take<float>(static_cast<size_t>(std::max<int64_t>(A, 1)) * wi);
  const size_t bn_ws_bytes = stemgnn_bn_workspace_bytes(N, wo);
  void* bn_ws = c.take<unsigned char>(bn_ws_bytes);
  // The weight gradients of all layers run as ONE split-product launch and ONE reduction at the end of the phase
  // (DwBatch, csrc/linear.hip): each keeps its operand g_y and its slabs until then.
  DwBatch dws;

  const float* g_h = g_z;  // gradient w.r.t. the current layer's output
  for (int l = L - 1; l >= 0; --l) {
    const stemgnn_sage_layer& y = layers[l];
    const LayerSave& s = p.layers[l];
    const bool last = l == L - 1;
    const void* h_prev = l == 0 ? x : static_cast<const void*>(p.layers[l - 1].h);  // stored in the feature kind
    const float* g_y = g_h;
    float* g_y_buf = c.take<float>(static_cast<size_t>(N) * wo);
    const size_t dwb_r = stemgnn_linear_bwd_weight_workspace_bytes(N, y.out_dim, y.in_dim);
    const size_t dwb_l = stemgnn_linear_bwd_weight_workspace_bytes(std::max<int64_t>(A, 1), y.out_dim, y.in_dim);
    void* dw_ws_r = c.take<unsigned char>(dwb_r);
    void* dw_ws_l = c.take<unsigned char>(dwb_l);
    if (!(last && p.tail_identity)) {
      const int act = last ? 0 : cfg->act;
      const float pdrop = last ? 0.f : cfg->dropout_p;
      STEMGNN_TRY(stemgnn_bn_act_drop_bwd(g_h, s.y, N, y.out_dim, cfg->use_bn ? s.mean : nullptr,
                                          cfg->use_bn ? s.rstd : nullptr, cfg->use_bn ? y.bn_weight : nullptr,
                                          cfg->use_bn ? y.bn_bias : nullptr, act, cfg->negative_slope, pdrop,
                                          y.drop_seed, y.drop_offset, g_y_buf, y.g_bn_weight, y.g_bn_bias, bn_ws,
                                          bn_ws_bytes, stream));
      g_y = g_y_buf;
    }
    // lin_r's weight gradient (and lin_l's bias gradient) contract over every row, lin_l's over the rows that carry
    // an aggregate
    if (y.g_w_r || y.g_b_l) {
      if (!y.g_w_r) return STEMGNN_ERR_INVALID_ARG;
      STEMGNN_TRY(dws.add(g_y, h_prev, cfg->feature_kind, N, y.out_dim, y.in_dim, y.g_w_r, y.b_l ? y.g_b_l : nullptr,
                          dw_ws_r, dwb_r, st));
    }
    if (y.g_w_l) {
      if (A > 0) {
        STEMGNN_TRY(dws.add(g_y, s.agg, kF32, A, y.out_dim, y.in_dim, y.g_w_l, nullptr, dw_ws_l, dwb_l, st));
      } else {
        STEMGNN_HIP_TRY(hipMemsetAsync(y.g_w_l, 0, sizeof(float) * y.out_dim * y.in_dim, st));
      }
    }
    float* g_prev = l == 0 ? g_x : g_h_buf[l & 1];
    if (!g_prev) break;  // the input needs no gradient
    STEMGNN_TRY(stemgnn_linear_bwd_data(g_y, y.w_r, N, y.out_dim, y.in_dim, g_prev, stream));
    if (A > 0) {
      // the rows that carry an aggregate are few (a sampled batch: seeds + first hop): the few-row product, if it fits
      // (chosen by shape and mode up front: a launch error of either path is an error of the phase, not a fallback)
      if (stemgnn_linear_set_mode(-1) == 1 && linear_direct_ok(A, y.in_dim, y.out_dim))
        STEMGNN_TRY(stemgnn_linear_few_rows(g_y, y.w_l, nullptr, A, y.in_dim, y.out_dim, g_agg, 1, stream));
      else
        STEMGNN_TRY(stemgnn_linear_bwd_data(g_y, y.w_l, A, y.out_dim, y.in_dim, g_agg, stream));
      // the aggregation's backward adds onto lin_r's share (no separate accumulation pass)
      STEMGNN_TRY(stemgnn_sage_agg_bwd_acc_k(g_agg, h_prev, cfg->feature_kind, N, y.in_dim, g->rowptr_t, g->dst_t,
                                             g->eid_t, g->inv_deg, edge_attr, etab, g->etype_slot_t, T, g_prev, stream));
    }
    g_h = g_prev;
  }
  return dws.flush(st);
}

size_t stemgnn_vq_save_bytes(const stemgnn_vq_params* p, int64_t N) {
  if (!vq_params_ok(p) || N < 0) return 0;
  return plan_vq(nullptr, p, N).bytes;
}

int stemgnn_vq_fwd(const stemgnn_vq_params* p, const float* z, int64_t N, int training, float* quantize, int64_t* ind,
                   float* loss, void* save, size_t save_bytes, void* stream) {
  if (!vq_params_ok(p) || N < 0 || !loss) return STEMGNN_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (N == 0) {
    STEMGNN_HIP_TRY(hipMemsetAsync(loss, 0, sizeof(float), st));
    return STEMGNN_OK;
  }
  if (!z || !quantize || !ind || !save) return STEMGNN_ERR_INVALID_ARG;
  if (save_bytes < stemgnn_vq_save_bytes(p, N)) return STEMGNN_ERR_WORKSPACE;
  const VqSave s = plan_vq(save, p, N);
  const int64_t H = p->heads, Dc = p->code_dim, K = p->codebook_size, D = p->dim, HD = H * Dc;
  // project_in (vq.py:881)
  STEMGNN_TRY(stemgnn_linear_fwd(z, p->w_in, D, nullptr, nullptr, 0, p->b_in, N, HD, s.xp, nullptr, nullptr, -1, stream));
  // codebook-side tables: |e|^2 per code and the projected code rows table[h][k] = W_out[:, h-block] embed[h][k]
  // (the pair-format assignment kernel takes the codes' squared norms from the rows it cuts: one launch less)
  if (!vq_assign_takes_own_sqnorm(N, H, Dc, K)) STEMGNN_TRY(stemgnn_code_sqnorm(p->embed, H * K, Dc, s.esq, stream));
  STEMGNN_TRY(stemgnn_small_gemm(p->embed, Dc, 1, K * Dc, p->w_out, 1, HD, Dc, s.table, D, 1, K * D, K, D, Dc, H, stream));
  // l2norm + similarity + arg-max + commitment sum (vq.py:891, 650-657, 1007-1009), nothing of size [N, H*Dc] written
  const bool commit = training && p->commitment_weight > 0.f;
  const float scale = commit ? p->commitment_weight / static_cast<float>(static_cast<double>(N) * HD) : 0.f;
  STEMGNN_TRY(stemgnn_vq_assign_lean(s.xp, N, H, Dc, p->embed, s.esq, K, s.norm, ind, s.terms, scale, s.assign_ws,
                                     s.assign_ws_bytes, stream));
  const bool ortho = training && p->ortho_weight > 0.f;
  if (ortho) {
    // the regulariser's finishing block also writes loss = commitment term + regulariser
    STEMGNN_TRY(ortho_loss_fwd_plus(p->embed, p->ortho_ids, H, K, Dc, p->num_ortho_ids, p->ortho_weight, s.terms + 1,
                                    commit ? s.terms : nullptr, loss, s.loss_ws, s.loss_ws_bytes, stream));
  } else {
    k_scalar_add<<<1, 64, 0, st>>>(commit ? s.terms : nullptr, nullptr, loss);
    STEMGNN_LAUNCH_CHECK();
  }
  // project_out of the quantised heads (vq.py:1041), read off the table
  STEMGNN_TRY(stemgnn_codes_project(s.table, ind, p->b_out, N, H, K, D, quantize, stream));
  return STEMGNN_OK;
}

size_t stemgnn_vq_bwd_scratch_bytes(const stemgnn_vq_params* p, int64_t N) {
  if (!vq_params_ok(p) || N < 0) return 0;
  const size_t n1 = static_cast<size_t>(std::max<int64_t>(N, 1));
  const int64_t HD = p->heads * p->code_dim;
  return 2 * a256(n1 * HD * 4) + a256(static_cast<size_t>(p->heads) * p->codebook_size * p->dim * 4) +
         a256(stemgnn_code_segment_sums_workspace_bytes(N, p->heads, p->codebook_size, p->dim)) +
         a256(stemgnn_linear_bwd_weight_workspace_bytes(N, HD, p->dim)) + 1024;
}

int stemgnn_vq_bwd(const stemgnn_vq_params* p, const float* z, int64_t N, const int64_t* ind, const float* g_quantize,
                   const float* g_loss, float* g_z, const void* save, size_t save_bytes, void* scratch,
                   size_t scratch_bytes, void* stream) {
  if (!vq_params_ok(p) || N < 0) return STEMGNN_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t H = p->heads, Dc = p->code_dim, K = p->codebook_size, D = p->dim, HD = H * Dc;
  if (N == 0) {
    if (p->g_w_in) STEMGNN_HIP_TRY(hipMemsetAsync(p->g_w_in, 0, sizeof(float) * HD * D, st));
    if (p->g_b_in) STEMGNN_HIP_TRY(hipMemsetAsync(p->g_b_in, 0, sizeof(float) * HD, st));
    if (p->g_w_out) STEMGNN_HIP_TRY(hipMemsetAsync(p->g_w_out, 0, sizeof(float) * HD * D, st));
    if (p->g_b_out) STEMGNN_HIP_TRY(hipMemsetAsync(p->g_b_out, 0, sizeof(float) * D, st));
    if (p->g_embed) STEMGNN_HIP_TRY(hipMemsetAsync(p->g_embed, 0, sizeof(float) * H * K * Dc, st));
    return STEMGNN_OK;
  }
  if (!z || !ind || !save || !scratch) return STEMGNN_ERR_INVALID_ARG;
  if (save_bytes < stemgnn_vq_save_bytes(p, N) || scratch_bytes < stemgnn_vq_bwd_scratch_bytes(p, N))
    return STEMGNN_ERR_WORKSPACE;
  const VqSave s = plan_vq(save, p, N);
  BtScope plane_scope(st);  // g_xp is cut once for project_in's backward-data product and its weight gradient
  Carver c(scratch);
  float* g_q = c.take<float>(static_cast<size_t>(N) * HD);
  float* g_xp = c.take<float>(static_cast<size_t>(N) * HD);
  float* sums = c.take<float>(static_cast<size_t>(H) * K * D);
  const size_t seg_ws_bytes = stemgnn_code_segment_sums_workspace_bytes(N, H, K, D);
  void* seg_ws = c.take<unsigned char>(seg_ws_bytes);
  const size_t dw_ws_bytes = stemgnn_linear_bwd_weight_workspace_bytes(N, HD, D);
  void* dw_ws = c.take<unsigned char>(dw_ws_bytes);

  // one head per 128-column tile and the exact-bf16 matrix-core mode: project_out's backward-data product runs inside
  // the assignment's backward and the [N, H*Dc] gradient of the quantised rows never reaches memory
  const bool fuse = g_quantize && Dc <= 128 && stemgnn_linear_set_mode(-1) == 1;
  if (g_quantize) {
    // straight-through (vq.py:937): the gradient of project_out's input reaches the normalised rows unchanged
    if (!fuse) STEMGNN_TRY(stemgnn_linear_bwd_data(g_quantize, p->w_out, N, D, HD, g_q, stream));
    // project_out's parameters: its input rows are code rows, so dW_out_h = (segment sums of g by code)^T embed_h
    if (p->g_w_out || p->g_b_out) {
      STEMGNN_TRY(stemgnn_code_segment_sums(ind, H, K, g_quantize, N, D, sums, seg_ws, seg_ws_bytes, stream));
      if (p->g_w_out)
        STEMGNN_TRY(stemgnn_small_gemm(sums, 1, D, K * D, p->embed, Dc, 1, K * Dc, p->g_w_out, HD, 1, Dc, D, Dc, K, H, stream));
      if (p->g_b_out) STEMGNN_TRY(stemgnn_segment_colsum(sums, K, D, p->g_b_out, stream));
    }
  } else {
    STEMGNN_HIP_TRY(hipMemsetAsync(g_q, 0, sizeof(float) * N * HD, st));
    if (p->g_w_out) STEMGNN_HIP_TRY(hipMemsetAsync(p->g_w_out, 0, sizeof(float) * HD * D, st));
    if (p->g_b_out) STEMGNN_HIP_TRY(hipMemsetAsync(p->g_b_out, 0, sizeof(float) * D, st));
  }
  // through the commitment term and the l2 normalisation
  const float* gl = p->commitment_weight > 0.f ? g_loss : nullptr;
  // (fused: the [N, HD] buffer of g_q is free -- it takes the row maxima of g_xp and the cut weight of project_in's
  // backward-data product in the pair format, csrc/wspair.hip: k_linear_ksp)
  bool have_rowmax = false;
  float* rowmax = g_q;
  void* ksp_scratch = reinterpret_cast<void*>(a256(reinterpret_cast<uintptr_t>(g_q + static_cast<size_t>(N) * H)));
  if (fuse) {
    const bool want = g_z && linear_ksp_ok(N, HD, D) &&
                      static_cast<size_t>(N) * H * 4 + 256 + linear_ksp_scratch_bytes(HD) <= static_cast<size_t>(N) * HD * 4;
    STEMGNN_TRY(vq_assign_bwd_fused_rowmax(g_quantize, D, p->w_out, gl, p->commitment_weight, s.xp, s.norm, ind, p->embed, N,
                                           H, Dc, K, g_xp, want ? rowmax : nullptr, &have_rowmax, stream));
  } else {
    STEMGNN_TRY(stemgnn_vq_assign_bwd(g_q, gl, p->commitment_weight, s.xp, s.norm, ind, p->embed, N, H, Dc, K, g_xp, stream));
  }
  if (g_z) {
    if (have_rowmax) STEMGNN_TRY(linear_ksp_launch(g_xp, rowmax, p->w_in, N, g_z, ksp_scratch, st));
    else STEMGNN_TRY(stemgnn_linear_bwd_data(g_xp, p->w_in, N, HD, D, g_z, stream));
  }
  if (p->g_w_in || p->g_b_in) {
    if (!p->g_w_in) return STEMGNN_ERR_INVALID_ARG;
    STEMGNN_TRY(stemgnn_linear_bwd_weight(g_xp, z, N, HD, D, p->g_w_in, p->b_in ? p->g_b_in : nullptr, dw_ws, dw_ws_bytes,
                                          stream));
  }
  if (p->g_embed) {
    if (p->ortho_weight > 0.f && g_loss) {
      STEMGNN_TRY(stemgnn_ortho_loss_bwd(p->embed, p->ortho_ids, H, K, Dc, p->num_ortho_ids, p->ortho_weight, g_loss,
                                         p->g_embed, stream));
    } else {
      STEMGNN_HIP_TRY(hipMemsetAsync(p->g_embed, 0, sizeof(float) * H * K * Dc, st));
    }
  }
  return STEMGNN_OK;
}

}  // extern "C"
