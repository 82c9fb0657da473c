// Small fused loss kernels of the pretraining step (each replaces 10-30 tiny ATen launches of
// the reference's loss arithmetic and its autograd):
//   * mean squared error + gradient        (F.mse_loss at model/pt_model.py:43,81)
//   * mean(1 - cos(z, h)) + gradient wrt h (model/pt_model.py:96-100)
//   * orthogonal codebook regulariser + gradient (model/vq.py:232-237,1011-1028)
// All are launch-latency bound (inputs of 0.1-6 MB); every reduction is a fixed-order tree
// inside one block, so results are bitwise reproducible.
#include "common.h"

namespace stemgnn {
namespace {

constexpr int kRed = 1024;
constexpr float kNormEps = 1e-12f;  // F.normalize eps

__device__ inline double block_sum(double v, double* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int o = kRed / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

// loss[0] = scale * mean((p - t)^2)
__global__ void __launch_bounds__(kRed) k_mse_fwd(const float* __restrict__ p, const float* __restrict__ t, int64_t n,
                                                  float scale, float* __restrict__ loss) {
  __shared__ double red[kRed];
  double s = 0.0;
  const int64_t n4 = n / 4;
  for (int64_t i = threadIdx.x; i < n4; i += kRed) {
    const float4 a = *reinterpret_cast<const float4*>(p + 4 * i), b = *reinterpret_cast<const float4*>(t + 4 * i);
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z, dw = a.w - b.w;
    s += static_cast<double>(dx * dx + dy * dy + dz * dz + dw * dw);
  }
  for (int64_t i = 4 * n4 + threadIdx.x; i < n; i += kRed) {
    const float d = p[i] - t[i];
    s += static_cast<double>(d * d);
  }
  const double tot = block_sum(s, red);
  if (threadIdx.x == 0) loss[0] = static_cast<float>(tot / static_cast<double>(n > 0 ? n : 1)) * scale;
}

// g_p = g[0] * scale * 2 (p - t) / n
__global__ void __launch_bounds__(256) k_mse_bwd(const float* __restrict__ p, const float* __restrict__ t, int64_t n,
                                                 float scale, const float* __restrict__ g, float* __restrict__ gp) {
  const float c = g[0] * scale * 2.0f / static_cast<float>(n);
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * 256)
    gp[i] = c * (p[i] - t[i]);
}

// loss = scale * mean_r (1 - <z_r, h_r> / (max(|z_r|,eps) max(|h_r|,eps))); one wave per row.
// Saves per row: dot of the normalised vectors (cos), 1/max(|z|,eps), |h| for the backward.
__global__ void __launch_bounds__(kRed) k_cos_fwd(const float* __restrict__ z, const float* __restrict__ h, int64_t rows,
                                                  int D, float scale, float* __restrict__ loss,
                                                  float* __restrict__ save /*[rows][3]*/) {
  __shared__ double red[kRed];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double acc = 0.0;
  for (int64_t r = wave; r < rows; r += kRed / 64) {
    float zz = 0.f, hh = 0.f, zh = 0.f;
    for (int c = lane; c < D; c += 64) {
      const float a = z[r * D + c], b = h[r * D + c];
      zz += a * a; hh += b * b; zh += a * b;
    }
    zz = wave_sum(zz); hh = wave_sum(hh); zh = wave_sum(zh);
    const float nz = sqrtf(zz), nh = sqrtf(hh);
    const float cz = fmaxf(nz, kNormEps), ch = fmaxf(nh, kNormEps);
    const float cosv = zh / (cz * ch);
    if (lane == 0) {
      save[r * 3 + 0] = cosv; save[r * 3 + 1] = 1.0f / cz; save[r * 3 + 2] = nh;
      acc += static_cast<double>(1.0f - cosv);
    }
  }
  const double tot = block_sum(acc, red);
  if (threadIdx.x == 0) loss[0] = static_cast<float>(tot / static_cast<double>(rows > 0 ? rows : 1)) * scale;
}

// g_h[r] = -(g scale / rows) * d cos / d h,  cos = <zn, h / max(|h|, eps)>
__global__ void __launch_bounds__(256) k_cos_bwd(const float* __restrict__ z, const float* __restrict__ h, int64_t rows,
                                                 int D, float scale, const float* __restrict__ g,
                                                 const float* __restrict__ save, float* __restrict__ gh) {
  const int lane = threadIdx.x & 63;
  const int64_t r = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float c = -g[0] * scale / static_cast<float>(rows);
  const float cosv = save[r * 3 + 0], inv_z = save[r * 3 + 1], nh = save[r * 3 + 2];
  const bool clamped = nh < kNormEps;
  const float inv_h = 1.0f / fmaxf(nh, kNormEps);
  for (int col = lane; col < D; col += 64) {
    const float zn = z[r * D + col] * inv_z, hn = h[r * D + col] * inv_h;
    // d/dh of <zn, h/|h|> = (zn - cos * hn) / |h|; in the eps-clamped branch h/eps is linear: zn / eps
    gh[r * D + col] = c * (clamped ? zn * inv_h : (zn - cosv * hn) * inv_h);
  }
}

// Orthogonal regulariser on embed[:, ids] ([H, M, Dc] selected codes): one block per head.
//   c_i = e_i / max(|e_i|, eps);  loss = scale * (sum_h sum_ij <c_i, c_j>^2 / (H M^2) - 1/M)
// Backward (same kernel family): d/dc_i = 4/(H M^2) sum_j <c_i,c_j> c_j, pushed through the
// normalisation and scattered into the dense codebook gradient (zero elsewhere).
template <bool BWD>
__global__ void __launch_bounds__(256)
k_ortho(const float* __restrict__ embed, const int64_t* __restrict__ ids, int H, int K, int Dc, int M, float scale,
        const float* __restrict__ g, float* __restrict__ loss_partial /*[H]*/, float* __restrict__ g_embed) {
  extern __shared__ float sm[];  // cn [M][Dc], gram [M][M], inv_norm [M]
  float* cn = sm;
  float* gram = cn + M * Dc;
  float* inv = gram + M * M;
  __shared__ double red[256];
  const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* eh = embed + static_cast<int64_t>(h) * K * Dc;
  for (int i = wave; i < M; i += 4) {
    const float* e = eh + ids[i] * Dc;
    float ss = 0.f;
    for (int c = lane; c < Dc; c += 64) ss += e[c] * e[c];
    ss = wave_sum(ss);
    const float iv = 1.0f / fmaxf(sqrtf(ss), kNormEps);
    if (lane == 0) inv[i] = iv;
    for (int c = lane; c < Dc; c += 64) cn[i * Dc + c] = e[c] * iv;
  }
  __syncthreads();
  for (int p = tid; p < M * M; p += 256) {
    const int i = p / M, j = p - i * M;
    float d = 0.f;
    for (int c = 0; c < Dc; ++c) d += cn[i * Dc + c] * cn[j * Dc + c];
    gram[p] = d;
  }
  __syncthreads();
  if (!BWD) {
    double s = 0.0;
    for (int p = tid; p < M * M; p += 256) s += static_cast<double>(gram[p]) * gram[p];
    red[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    if (tid == 0) loss_partial[h] = static_cast<float>(red[0]);
  } else {
    const float coef = g[0] * scale * 4.0f / (static_cast<float>(H) * M * M);
    for (int i = wave; i < M; i += 4) {
      // gc = coef * sum_j gram[i][j] c_j ;  ge = (gc - c_i <gc, c_i>) * inv_i
      float dotp = 0.f;
      for (int c = lane; c < Dc; c += 64) {
        float gc = 0.f;
        for (int j = 0; j < M; ++j) gc += gram[i * M + j] * cn[j * Dc + c];
        dotp += gc * cn[i * Dc + c];
      }
      dotp = wave_sum(dotp);
      float* ge = g_embed + (static_cast<int64_t>(h) * K + ids[i]) * Dc;
      for (int c = lane; c < Dc; c += 64) {
        float gc = 0.f;
        for (int j = 0; j < M; ++j) gc += gram[i * M + j] * cn[j * Dc + c];
        ge[c] = coef * (gc - cn[i * Dc + c] * dotp) * inv[i];
      }
    }
  }
}

__global__ void k_ortho_finish(const float* __restrict__ partial, int H, int M, float scale, float* __restrict__ loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int h = 0; h < H; ++h) s += partial[h];
    loss[0] = static_cast<float>(s / (static_cast<double>(H) * M * M) - 1.0 / M) * scale;
  }
}

}  // namespace
}  // namespace stemgnn

using namespace stemgnn;

extern "C" {

int stemgnn_mse_loss_fwd(const float* pred, const float* target, int64_t n, float scale, float* loss, void* stream_) {
  if (n < 0 || !loss || (n > 0 && (!pred || !target))) return STEMGNN_ERR_INVALID_ARG;
  k_mse_fwd<<<1, kRed, 0, static_cast<hipStream_t>(stream_)>>>(pred, target, n, scale, loss);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_mse_loss_bwd(const float* pred, const float* target, int64_t n, float scale, const float* g_loss,
                         float* g_pred, void* stream_) {
  if (n < 0) return STEMGNN_ERR_INVALID_ARG;
  if (n == 0) return STEMGNN_OK;
  if (!pred || !target || !g_loss || !g_pred) return STEMGNN_ERR_INVALID_ARG;
  int64_t grid = (n + 255) / 256;
  if (grid > 2048) grid = 2048;
  k_mse_bwd<<<static_cast<unsigned>(grid), 256, 0, static_cast<hipStream_t>(stream_)>>>(pred, target, n, scale, g_loss,
                                                                                      g_pred);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_cosine_loss_fwd(const float* z, const float* h, int64_t rows, int64_t dim, float scale, float* loss,
                            float* save, void* stream_) {
  if (rows < 0 || dim <= 0 || !loss || (rows > 0 && (!z || !h || !save))) return STEMGNN_ERR_INVALID_ARG;
  k_cos_fwd<<<1, kRed, 0, static_cast<hipStream_t>(stream_)>>>(z, h, rows, static_cast<int>(dim), scale, loss, save);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_cosine_loss_bwd(const float* z, const float* h, int64_t rows, int64_t dim, float scale,
                            const float* g_loss, const float* save, float* g_h, void* stream_) {
  if (rows < 0 || dim <= 0) return STEMGNN_ERR_INVALID_ARG;
  if (rows == 0) return STEMGNN_OK;
  if (!z || !h || !g_loss || !save || !g_h) return STEMGNN_ERR_INVALID_ARG;
  k_cos_bwd<<<static_cast<unsigned>((rows + 3) / 4), 256, 0, static_cast<hipStream_t>(stream_)>>>(
      z, h, rows, static_cast<int>(dim), scale, g_loss, save, g_h);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

size_t stemgnn_ortho_loss_workspace_bytes(int64_t heads) { return heads > 0 ? static_cast<size_t>(heads) * 4 + 256 : 0; }

static size_t ortho_lds(int64_t M, int64_t Dc) { return static_cast<size_t>(M * Dc + M * M + M) * sizeof(float); }

int stemgnn_ortho_loss_fwd(const float* embed, const int64_t* ids, int64_t heads, int64_t codebook_size,
                           int64_t code_dim, int64_t num_ids, float scale, float* loss, void* workspace,
                           size_t workspace_bytes, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (heads <= 0 || codebook_size <= 0 || code_dim <= 0 || num_ids <= 0 || !embed || !ids || !loss || !workspace)
    return STEMGNN_ERR_INVALID_ARG;
  if (ortho_lds(num_ids, code_dim) > 150 * 1024) return STEMGNN_ERR_INVALID_ARG;
  if (workspace_bytes < stemgnn_ortho_loss_workspace_bytes(heads)) return STEMGNN_ERR_WORKSPACE;
  float* partial = reinterpret_cast<float*>(align_up(reinterpret_cast<uintptr_t>(workspace), 256));
  const size_t lds = ortho_lds(num_ids, code_dim);
  if (lds > 64 * 1024) {
    static bool configured = false;
    if (!configured) {
      STEMGNN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ortho<false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      STEMGNN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ortho<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      configured = true;
    }
  }
  k_ortho<false><<<static_cast<unsigned>(heads), 256, lds, st>>>(embed, ids, static_cast<int>(heads),
                                                                 static_cast<int>(codebook_size),
                                                                 static_cast<int>(code_dim), static_cast<int>(num_ids),
                                                                 scale, nullptr, partial, nullptr);
  STEMGNN_LAUNCH_CHECK();
  k_ortho_finish<<<1, 64, 0, st>>>(partial, static_cast<int>(heads), static_cast<int>(num_ids), scale, loss);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_ortho_loss_bwd(const float* embed, const int64_t* ids, int64_t heads, int64_t codebook_size,
                           int64_t code_dim, int64_t num_ids, float scale, const float* g_loss, float* g_embed,
                           void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (heads <= 0 || codebook_size <= 0 || code_dim <= 0 || num_ids <= 0 || !embed || !ids || !g_loss || !g_embed)
    return STEMGNN_ERR_INVALID_ARG;
  if (ortho_lds(num_ids, code_dim) > 150 * 1024) return STEMGNN_ERR_INVALID_ARG;
  STEMGNN_HIP_TRY(hipMemsetAsync(g_embed, 0, sizeof(float) * heads * codebook_size * code_dim, st));
  k_ortho<true><<<static_cast<unsigned>(heads), 256, ortho_lds(num_ids, code_dim), st>>>(
      embed, ids, static_cast<int>(heads), static_cast<int>(codebook_size), static_cast<int>(code_dim),
      static_cast<int>(num_ids), scale, g_loss, nullptr, g_embed);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

}  // extern "C"
