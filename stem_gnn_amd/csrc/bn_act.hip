// K4: BatchNorm1d (training statistics) + ReLU/LeakyReLU + Dropout, fused, for gfx950.
//
// Reference: STEM-GNN/model/encoder.py:173 (nn.BatchNorm1d per layer), :313-317
// (norm -> activation -> dropout; activation/dropout skipped after the last layer).
// The reference issues one kernel per op (stats, normalise, relu, dropout mask, multiply),
// i.e. ~5 passes over [N, D]; here: one column-reduction pass + one fused apply pass, and the
// dropout mask is never stored (Philox counter keyed by (seed, offset, element)).
// All kernels are HBM-bound: stats reads N*D*4 B; apply reads + writes N*D*4 B each.
#include "common.h"

namespace stemgnn {
namespace {

constexpr int kBlock = 256;
constexpr int kMaxPartialBlocks = 512;
constexpr int kFinCols = 1;     // finalize: 1 column x 256 partial slices per 256-thread block
constexpr int kFinSlices = kBlock / kFinCols;

__device__ inline float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ inline void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

struct ColGeom {
  int tx;       // lanes along the (float4) column axis, power of two <= 256
  int ty;       // row lanes per block = 256 / tx
  int ctiles;   // column tiles of width tx covering D/4
};

inline ColGeom col_geom(int64_t D) {
  int nvec = static_cast<int>(D / 4);
  int tx = 1;
  while (tx < nvec && tx < kBlock) tx <<= 1;
  ColGeom g;
  g.tx = tx;
  g.ty = kBlock / tx;
  g.ctiles = (nvec + tx - 1) / tx;
  return g;
}

inline int partial_blocks(int64_t N, const ColGeom& g) {
  int64_t rows_per_block = static_cast<int64_t>(g.ty) * 16;  // >=16 rows per row-lane before adding blocks
  int64_t b = (N + rows_per_block - 1) / rows_per_block;
  if (b < 1) b = 1;
  if (b > kMaxPartialBlocks) b = kMaxPartialBlocks;
  return static_cast<int>(b);
}

// The per-element backward signal entering the BN: g_out * dropout_scale * act'(pre).
struct Epilogue {
  int act;
  float slope, p, scale;
  uint64_t seed, offset;
};

__device__ inline void keep4(const Epilogue& ep, uint64_t vec_index, bool (&keep)[4]) {
  if (ep.p <= 0.f) {
    keep[0] = keep[1] = keep[2] = keep[3] = true;
    return;
  }
  uint32_t r[4];
  Philox::gen(ep.seed, ep.offset, vec_index, r);
#pragma unroll
  for (int k = 0; k < 4; ++k) keep[k] = Philox::to_unit(r[k]) >= ep.p;
}

__device__ inline float act_fwd(float v, int act, float slope) {
  if (act == 0) return v;
  return v > 0.f ? v : (slope == 0.f ? (v != v ? v : 0.f) : v * slope);
}

__device__ inline float act_grad(float pre, int act, float slope) {
  if (act == 0) return 1.f;
  return pre > 0.f ? 1.f : slope;
}

// MODE 0: sum(y), sum(y*y).  MODE 1: sum(g_bn), sum(g_bn * x_hat).
template <int MODE>
__global__ void __launch_bounds__(kBlock)
k_col_partials(const float* __restrict__ y, const float* __restrict__ g_out, int64_t N, int D, int tx, int ty,
               const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
               const float* __restrict__ beta, Epilogue ep, float* __restrict__ partial /*[blocks][2][D]*/) {
  __shared__ float4 red[2][kBlock];
  const int nvec = D / 4;
  const int cx = threadIdx.x % tx;
  const int ry = threadIdx.x / tx;
  const int64_t rows_per_block = (N + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * rows_per_block;
  const int64_t r1 = min(N, r0 + rows_per_block);
  for (int ct = 0; ct * tx < nvec; ++ct) {
    const int c = ct * tx + cx;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (c < nvec) {
      float4 m = make_float4(0.f, 0.f, 0.f, 0.f), rs = make_float4(1.f, 1.f, 1.f, 1.f), ga = rs, be = m;
      if (MODE == 1) {
        if (mean) { m = ld4(mean + 4 * c); rs = ld4(rstd + 4 * c); }
        if (gamma) { ga = ld4(gamma + 4 * c); be = ld4(beta + 4 * c); }
      }
#pragma unroll 4
      for (int64_t r = r0 + ry; r < r1; r += ty) {
        const float4 v = ld4(y + r * D + 4 * c);
        if (MODE == 0) {
          a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
          b.x += v.x * v.x; b.y += v.y * v.y; b.z += v.z * v.z; b.w += v.w * v.w;
        } else {
          const float4 g = ld4(g_out + r * D + 4 * c);
          bool keep[4];
          keep4(ep, static_cast<uint64_t>(r) * nvec + c, keep);
          const float xh[4] = {(v.x - m.x) * rs.x, (v.y - m.y) * rs.y, (v.z - m.z) * rs.z, (v.w - m.w) * rs.w};
          const float gg[4] = {g.x, g.y, g.z, g.w};
          const float gam[4] = {ga.x, ga.y, ga.z, ga.w}, bet[4] = {be.x, be.y, be.z, be.w};
          float gb[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float pre = xh[k] * gam[k] + bet[k];
            gb[k] = keep[k] ? gg[k] * ep.scale * act_grad(pre, ep.act, ep.slope) : 0.f;
          }
          a.x += gb[0]; a.y += gb[1]; a.z += gb[2]; a.w += gb[3];
          b.x += gb[0] * xh[0]; b.y += gb[1] * xh[1]; b.z += gb[2] * xh[2]; b.w += gb[3] * xh[3];
        }
      }
    }
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
    if (ry == 0 && c < nvec) {
      for (int j = 1; j < ty; ++j) {
        const float4 a2 = red[0][j * tx + cx], b2 = red[1][j * tx + cx];
        a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
        b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
      }
      float* p0 = partial + (static_cast<int64_t>(blockIdx.x) * 2) * D;
      st4(p0 + 4 * c, a);
      st4(p0 + D + 4 * c, b);
    }
    __syncthreads();
  }
}

// Combine block partials in fp64: kFinCols columns x kFinSlices slices per block, LDS tree over the slices.
__device__ inline void reduce_partials(const float* __restrict__ partial, int blocks, int D, int c, int slice,
                                       double* s_out, double* q_out) {
  __shared__ double red_s[kBlock], red_q[kBlock];
  double s = 0.0, q = 0.0;
  if (c < D) {
    // eight loads in flight per thread (the slabs sit in L2; a one-at-a-time walk is a chain of its latencies), four
    // running sums combined in a fixed order
    double sa[4] = {0.0, 0.0, 0.0, 0.0}, qa[4] = {0.0, 0.0, 0.0, 0.0};
    int b = slice;
    for (; b + 3 * kFinSlices < blocks; b += 4 * kFinSlices) {
      float v[4], u[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = partial[(static_cast<int64_t>(b + j * kFinSlices) * 2) * D + c];
        u[j] = partial[(static_cast<int64_t>(b + j * kFinSlices) * 2 + 1) * D + c];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { sa[j] += v[j]; qa[j] += u[j]; }
    }
    for (int j = 0; b < blocks; b += kFinSlices, ++j) {
      sa[j] += partial[(static_cast<int64_t>(b) * 2) * D + c];
      qa[j] += partial[(static_cast<int64_t>(b) * 2 + 1) * D + c];
    }
    s = (sa[0] + sa[1]) + (sa[2] + sa[3]);
    q = (qa[0] + qa[1]) + (qa[2] + qa[3]);
  }
  red_s[threadIdx.x] = s;
  red_q[threadIdx.x] = q;
  __syncthreads();
  for (int o = kFinSlices / 2; o > 0; o >>= 1) {
    if (slice < o) {
      red_s[threadIdx.x] += red_s[threadIdx.x + o * kFinCols];
      red_q[threadIdx.x] += red_q[threadIdx.x + o * kFinCols];
    }
    __syncthreads();
  }
  *s_out = red_s[threadIdx.x];
  *q_out = red_q[threadIdx.x];
}

__global__ void __launch_bounds__(kBlock)
k_stats_finalize(const float* __restrict__ partial, int blocks, int64_t N, int D, float eps,
                 float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ running_mean,
                 float* __restrict__ running_var, float momentum, int64_t* __restrict__ num_batches_tracked) {
  const int cl = threadIdx.x % kFinCols, slice = threadIdx.x / kFinCols;
  const int c = blockIdx.x * kFinCols + cl;
  if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) num_batches_tracked[0] += 1;  // BatchNorm1d's counter
  double s, q;
  reduce_partials(partial, blocks, D, c, slice, &s, &q);
  if (slice != 0 || c >= D) return;
  const double n = static_cast<double>(N);
  const double m = s / n;
  double var = q / n - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = static_cast<float>(m);
  rstd[c] = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
  if (running_mean) {
    const double unbiased = N > 1 ? var * n / (n - 1.0) : var;
    running_mean[c] = static_cast<float>((1.0 - momentum) * running_mean[c] + momentum * m);
    running_var[c] = static_cast<float>((1.0 - momentum) * running_var[c] + momentum * unbiased);
  }
}

__global__ void __launch_bounds__(kBlock)
k_grad_finalize(const float* __restrict__ partial, int blocks, int D, float* __restrict__ g_gamma,
                float* __restrict__ g_beta) {
  const int cl = threadIdx.x % kFinCols, slice = threadIdx.x / kFinCols;
  const int c = blockIdx.x * kFinCols + cl;
  double s, q;
  reduce_partials(partial, blocks, D, c, slice, &s, &q);
  if (slice != 0 || c >= D) return;
  g_beta[c] = static_cast<float>(s);
  g_gamma[c] = static_cast<float>(q);
}

// FWD: out = dropout(act(bn(y))).  BWD: g_y = gamma*rstd*(g_bn - sum_gb/N - x_hat*sum_gbx/N).
template <bool BWD>
__global__ void __launch_bounds__(kBlock)
k_apply(const float* __restrict__ y, const float* __restrict__ g_out, int64_t N, int D,
        const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
        const float* __restrict__ beta, const float* __restrict__ sum_gb, const float* __restrict__ sum_gbx,
        Epilogue ep, float* __restrict__ out, int out_kind) {
  const int nvec = D / 4;
  const int64_t total = N * nvec;
  const float inv_n = 1.0f / static_cast<float>(N);
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  const int64_t i0 = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  // with a stride that is a multiple of the row length (elementwise_grid) a thread stays on one column group: its
  // per-column parameters are loaded once, not once per element (and no 64-bit modulo per element)
  const bool fixed_col = stride % nvec == 0;
  int c = static_cast<int>(i0 % nvec);
  float4 m = make_float4(0.f, 0.f, 0.f, 0.f), rs = make_float4(1.f, 1.f, 1.f, 1.f), ga = rs, be = m;
  float4 s1 = m, s2 = m;
  auto load_params = [&]() {
    if (mean) { m = ld4(mean + 4 * c); rs = ld4(rstd + 4 * c); }
    if (gamma) { ga = ld4(gamma + 4 * c); be = ld4(beta + 4 * c); }
    if (BWD && mean) { s1 = ld4(sum_gb + 4 * c); s2 = ld4(sum_gbx + 4 * c); }
  };
  if (i0 < total) load_params();
  for (int64_t i = i0; i < total; i += stride) {
    if (!fixed_col) {
      c = static_cast<int>(i % nvec);
      load_params();
    }
    const float4 v = ld4(y + 4 * i);
    bool keep[4];
    keep4(ep, static_cast<uint64_t>(i), keep);
    const float vv[4] = {v.x, v.y, v.z, v.w};
    const float mm[4] = {m.x, m.y, m.z, m.w}, rr[4] = {rs.x, rs.y, rs.z, rs.w};
    const float gam[4] = {ga.x, ga.y, ga.z, ga.w}, bet[4] = {be.x, be.y, be.z, be.w};
    float o[4];
    if (!BWD) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float pre = (vv[k] - mm[k]) * rr[k] * gam[k] + bet[k];
        const float a = act_fwd(pre, ep.act, ep.slope);
        o[k] = keep[k] ? a * ep.scale : 0.f;
      }
    } else {
      const float4 g = ld4(g_out + 4 * i);
      const float gg[4] = {g.x, g.y, g.z, g.w};
      const float sb[4] = {s1.x, s1.y, s1.z, s1.w}, sx[4] = {s2.x, s2.y, s2.z, s2.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xh = (vv[k] - mm[k]) * rr[k];
        const float pre = xh * gam[k] + bet[k];
        const float gb = keep[k] ? gg[k] * ep.scale * act_grad(pre, ep.act, ep.slope) : 0.f;
        o[k] = mean ? gam[k] * rr[k] * (gb - sb[k] * inv_n - xh * sx[k] * inv_n) : gb * gam[k];
      }
    }
    st4_kind(out, 4 * i, out_kind, make_float4(o[0], o[1], o[2], o[3]));  // bf16 storage rounds here (RNE)
  }
}

__global__ void __launch_bounds__(kBlock)
k_keep_mask(int64_t n, float p, uint64_t seed, uint64_t offset, uint8_t* __restrict__ keep) {
  const int64_t nv = (n + 3) / 4;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < nv;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    uint32_t r[4];
    Philox::gen(seed, offset, static_cast<uint64_t>(i), r);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (4 * i + k < n) keep[4 * i + k] = (p <= 0.f || Philox::to_unit(r[k]) >= p) ? 1 : 0;
  }
}

inline Epilogue make_epilogue(int act, float slope, float p, uint64_t seed, uint64_t offset) {
  Epilogue ep;
  ep.act = act; ep.slope = slope; ep.p = p; ep.seed = seed; ep.offset = offset;
  ep.scale = p <= 0.f ? 1.f : (p >= 1.f ? 0.f : 1.0f / (1.0f - p));
  return ep;
}

inline int elementwise_grid(int64_t nvec_total, int64_t nvec_row = 1) {
  int64_t g = (nvec_total + kBlock - 1) / kBlock;
  if (g > 256 * 16) g = 256 * 16;
  // a grid stride that is a multiple of the row length keeps a thread on one column group: k_apply then loads the
  // per-column parameters once instead of once per element
  int64_t a = nvec_row, b = kBlock;
  while (b) { const int64_t t = a % b; a = b; b = t; }
  const int64_t m = nvec_row / a;
  if (g >= m) g = g / m * m;
  return static_cast<int>(g < 1 ? 1 : g);
}

inline bool dims_ok(int64_t N, int64_t D) { return N >= 0 && D > 0 && D % 4 == 0 && D <= 16384; }

}  // namespace
}  // namespace stemgnn

using namespace stemgnn;

extern "C" {

size_t stemgnn_bn_workspace_bytes(int64_t N, int64_t D) {
  if (!dims_ok(N, D)) return 0;
  return static_cast<size_t>(kMaxPartialBlocks) * 2 * D * sizeof(float) + 2 * D * sizeof(float) + 512;
}

int stemgnn_bn_stats(const float* y, int64_t N, int64_t D, float eps, float* mean, float* rstd, float* running_mean,
                     float* running_var, float momentum, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (!dims_ok(N, D) || N == 0 || !y || !mean || !rstd || !workspace) return STEMGNN_ERR_INVALID_ARG;
  if ((running_mean == nullptr) != (running_var == nullptr)) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(N)) return STEMGNN_ERR_TOO_LARGE;
  if (workspace_bytes < stemgnn_bn_workspace_bytes(N, D)) return STEMGNN_ERR_WORKSPACE;
  float* partial = reinterpret_cast<float*>(align_up(reinterpret_cast<uintptr_t>(workspace), 256));
  ColGeom g = col_geom(D);
  int blocks = partial_blocks(N, g);
  Epilogue ep = make_epilogue(0, 0.f, 0.f, 0, 0);
  k_col_partials<0><<<blocks, kBlock, 0, st>>>(y, nullptr, N, static_cast<int>(D), g.tx, g.ty, nullptr, nullptr,
                                               nullptr, nullptr, ep, partial);
  STEMGNN_LAUNCH_CHECK();
  k_stats_finalize<<<static_cast<int>((D + kFinCols - 1) / kFinCols), kBlock, 0, st>>>(
      partial, blocks, N, static_cast<int>(D), eps, mean, rstd, running_mean, running_var, momentum, nullptr);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_bn_stats_from_partials(const float* partial, int64_t blocks, int64_t N, int64_t D, float eps, float* mean,
                                   float* rstd, float* running_mean, float* running_var, float momentum,
                                   int64_t* num_batches_tracked, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (!dims_ok(N, D) || N == 0 || blocks <= 0 || blocks > (1 << 24) || !partial || !mean || !rstd)
    return STEMGNN_ERR_INVALID_ARG;
  if ((running_mean == nullptr) != (running_var == nullptr)) return STEMGNN_ERR_INVALID_ARG;
  k_stats_finalize<<<static_cast<int>((D + kFinCols - 1) / kFinCols), kBlock, 0, st>>>(
      partial, static_cast<int>(blocks), N, static_cast<int>(D), eps, mean, rstd, running_mean, running_var, momentum,
      num_batches_tracked);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_bn_act_drop_fwd(const float* y, int64_t N, int64_t D, const float* mean, const float* rstd,
                            const float* gamma, const float* beta, int act, float negative_slope, float p,
                            uint64_t seed, uint64_t offset, float* out, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (!dims_ok(N, D) || (act != 0 && act != 1) || p < 0.f || p > 1.f) return STEMGNN_ERR_INVALID_ARG;
  if ((mean == nullptr) != (rstd == nullptr) || (gamma == nullptr) != (beta == nullptr)) return STEMGNN_ERR_INVALID_ARG;
  if (N == 0) return STEMGNN_OK;
  if (!y || !out) return STEMGNN_ERR_INVALID_ARG;
  Epilogue ep = make_epilogue(act, negative_slope, p, seed, offset);
  k_apply<false><<<elementwise_grid(N * (D / 4), D / 4), kBlock, 0, st>>>(y, nullptr, N, static_cast<int>(D), mean, rstd,
                                                                  gamma, beta, nullptr, nullptr, ep, out, kF32);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_bn_act_drop_fwd_k(const float* y, int64_t N, int64_t D, const float* mean, const float* rstd,
                              const float* gamma, const float* beta, int act, float negative_slope, float p,
                              uint64_t seed, uint64_t offset, void* out, int32_t out_kind, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (!dims_ok(N, D) || (act != 0 && act != 1) || p < 0.f || p > 1.f) return STEMGNN_ERR_INVALID_ARG;
  if ((mean == nullptr) != (rstd == nullptr) || (gamma == nullptr) != (beta == nullptr)) return STEMGNN_ERR_INVALID_ARG;
  if (out_kind != kF32 && out_kind != kBF16) return STEMGNN_ERR_INVALID_ARG;
  if (N == 0) return STEMGNN_OK;
  if (!y || !out) return STEMGNN_ERR_INVALID_ARG;
  Epilogue ep = make_epilogue(act, negative_slope, p, seed, offset);
  k_apply<false><<<elementwise_grid(N * (D / 4), D / 4), kBlock, 0, st>>>(y, nullptr, N, static_cast<int>(D), mean, rstd,
                                                                  gamma, beta, nullptr, nullptr, ep,
                                                                  static_cast<float*>(out), out_kind);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_bn_act_drop_bwd(const float* g_out, const float* y, int64_t N, int64_t D, const float* mean,
                            const float* rstd, const float* gamma, const float* beta, int act, float negative_slope,
                            float p, uint64_t seed, uint64_t offset, float* g_y, float* g_gamma, float* g_beta,
                            void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (!dims_ok(N, D) || (act != 0 && act != 1) || p < 0.f || p > 1.f) return STEMGNN_ERR_INVALID_ARG;
  if ((mean == nullptr) != (rstd == nullptr) || (gamma == nullptr) != (beta == nullptr)) return STEMGNN_ERR_INVALID_ARG;
  if (N == 0) return STEMGNN_OK;
  if (!g_out || !y || !g_y) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(N)) return STEMGNN_ERR_TOO_LARGE;
  Epilogue ep = make_epilogue(act, negative_slope, p, seed, offset);
  const float *sum_gb = nullptr, *sum_gbx = nullptr;
  if (mean) {
    if (!workspace || workspace_bytes < stemgnn_bn_workspace_bytes(N, D)) return STEMGNN_ERR_WORKSPACE;
    float* partial = reinterpret_cast<float*>(align_up(reinterpret_cast<uintptr_t>(workspace), 256));
    float* sums = partial + static_cast<size_t>(kMaxPartialBlocks) * 2 * D;  // [2][D] when the caller wants no grads
    float* gb = g_beta ? g_beta : sums;
    float* gg = g_gamma ? g_gamma : sums + D;
    ColGeom g = col_geom(D);
    int blocks = partial_blocks(N, g);
    k_col_partials<1><<<blocks, kBlock, 0, st>>>(y, g_out, N, static_cast<int>(D), g.tx, g.ty, mean, rstd, gamma, beta,
                                                 ep, partial);
    STEMGNN_LAUNCH_CHECK();
    k_grad_finalize<<<static_cast<int>((D + kFinCols - 1) / kFinCols), kBlock, 0, st>>>(partial, blocks,
                                                                                   static_cast<int>(D), gg, gb);
    STEMGNN_LAUNCH_CHECK();
    sum_gb = gb;
    sum_gbx = gg;
  }
  k_apply<true><<<elementwise_grid(N * (D / 4), D / 4), kBlock, 0, st>>>(y, g_out, N, static_cast<int>(D), mean, rstd, gamma,
                                                                 beta, sum_gb, sum_gbx, ep, g_y, kF32);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_dropout_keep_mask(int64_t n, float p, uint64_t seed, uint64_t offset, uint8_t* keep, void* stream_) {
  if (n < 0 || p < 0.f || p > 1.f) return STEMGNN_ERR_INVALID_ARG;
  if (n == 0) return STEMGNN_OK;
  if (!keep) return STEMGNN_ERR_INVALID_ARG;
  k_keep_mask<<<elementwise_grid((n + 3) / 4), kBlock, 0, static_cast<hipStream_t>(stream_)>>>(n, p, seed, offset, keep);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

}  // extern "C"
