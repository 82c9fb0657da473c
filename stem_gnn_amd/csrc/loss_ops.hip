// Small fused loss kernels of the pretraining step (each replaces 10-30 tiny ATen launches of
// the reference's loss arithmetic and its autograd):
//   * mean squared error + gradient        (F.mse_loss at model/pt_model.py:43,81)
//   * mean(1 - cos(z, h)) + gradient wrt h (model/pt_model.py:96-100)
//   * orthogonal codebook regulariser + gradient (model/vq.py:232-237,1011-1028)
// All are launch-latency bound (inputs of 0.1-6 MB); every reduction is a fixed-order tree
// inside one block, so results are bitwise reproducible.
#include "common.h"

#include <map>
#include <mutex>
#include <utility>

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

// loss[0] = (sum_i partial[i]) * mul + add, fixed order, by the 256 threads of the calling block (the last to arrive)
__device__ inline void finish_sum_block(const double* partial, int n, double mul, double add, float* loss, double* red) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += ld_agent(partial + i);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = static_cast<float>(red[0] * mul + add);
}

// partial[b] = sum over block b's grid-stride share of (p - t)^2 (nb blocks); the last block to arrive finishes
__device__ inline void mse_partial_body(const float* __restrict__ p, const float* __restrict__ t, int64_t n,
                                        double* __restrict__ partial, unsigned int* counter, double mul,
                                        float* __restrict__ loss, int b, int nb, double* red) {
  double s = 0.0;
  const int64_t n4 = n / 4;
  // four loads in flight per thread (a full-batch loss walks 1e8 elements with at most 256 blocks), consumed in the
  // order a one-at-a-time loop would: the sum's bits do not change
  const int64_t stride = static_cast<int64_t>(nb) * 256;
  int64_t i = static_cast<int64_t>(b) * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    float4 a[4], c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = *reinterpret_cast<const float4*>(p + 4 * (i + u * stride));
      c[u] = *reinterpret_cast<const float4*>(t + 4 * (i + u * stride));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float dx = a[u].x - c[u].x, dy = a[u].y - c[u].y, dz = a[u].z - c[u].z, dw = a[u].w - c[u].w;
      s += static_cast<double>(dx * dx + dy * dy + dz * dz + dw * dw);
    }
  }
  for (; i < n4; i += stride) {
    const float4 a = *reinterpret_cast<const float4*>(p + 4 * i), c = *reinterpret_cast<const float4*>(t + 4 * i);
    const float dx = a.x - c.x, dy = a.y - c.y, dz = a.z - c.z, dw = a.w - c.w;
    s += static_cast<double>(dx * dx + dy * dy + dz * dz + dw * dw);
  }
  if (b == 0)
    for (int64_t i = 4 * n4 + threadIdx.x; i < n; i += 256) {
      const float d = p[i] - t[i];
      s += static_cast<double>(d * d);
    }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) { st_agent(partial + b, red[0]); wait_stores(); }
  if (ticket_last(counter, nb)) finish_sum_block(partial, nb, mul, 0.0, loss, red);
}

__global__ void __launch_bounds__(256) k_mse_partial(const float* __restrict__ p, const float* __restrict__ t, int64_t n,
                                                     double* __restrict__ partial, unsigned int* counter, double mul,
                                                     float* __restrict__ loss) {
  __shared__ double red[256];
  mse_partial_body(p, t, n, partial, counter, mul, loss, blockIdx.x, gridDim.x, red);
}

// loss[0] = (sum_b partial[b]) * mul + add   (fixed order)
__global__ void __launch_bounds__(256) k_finish_sum(const double* __restrict__ partial, int n, double mul, double add,
                                                    float* __restrict__ loss, const float* __restrict__ plus = nullptr,
                                                    float* __restrict__ total = nullptr) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float v = static_cast<float>(red[0] * mul + add);
    loss[0] = v;
    if (total) total[0] = (plus ? plus[0] : 0.f) + v;  // a sum of two loss terms in the same launch (VQ: commit + ortho)
  }
}

// g_p = g[0] * scale * 2 (p - t) / n
__device__ inline void mse_bwd_body(const float* __restrict__ p, const float* __restrict__ t, int64_t n, float scale,
                                    const float* __restrict__ g, float* __restrict__ gp, int b, int nb) {
  const float c = g[0] * scale * 2.0f / static_cast<float>(n);
  if (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(t) | reinterpret_cast<uintptr_t>(gp)) & 15) == 0) {
    // 16-byte accesses (the element-wise result is the same); the few elements behind the last whole float4 below
    const int64_t n4 = n / 4;
    for (int64_t i = static_cast<int64_t>(b) * 256 + threadIdx.x; i < n4; i += static_cast<int64_t>(nb) * 256) {
      const float4 a = *reinterpret_cast<const float4*>(p + 4 * i), d = *reinterpret_cast<const float4*>(t + 4 * i);
      *reinterpret_cast<float4*>(gp + 4 * i) = make_float4(c * (a.x - d.x), c * (a.y - d.y), c * (a.z - d.z), c * (a.w - d.w));
    }
    if (b == 0)
      for (int64_t i = 4 * n4 + threadIdx.x; i < n; i += 256) gp[i] = c * (p[i] - t[i]);
    return;
  }
  for (int64_t i = static_cast<int64_t>(b) * 256 + threadIdx.x; i < n; i += static_cast<int64_t>(nb) * 256)
    gp[i] = c * (p[i] - t[i]);
}
__global__ void __launch_bounds__(256) k_mse_bwd(const float* __restrict__ p, const float* __restrict__ t, int64_t n,
                                                 float scale, const float* __restrict__ g, float* __restrict__ gp) {
  mse_bwd_body(p, t, n, scale, g, gp, blockIdx.x, gridDim.x);
}

// One wave per row: cos of the normalised rows (F.normalize eps clamp); save [rows][3] = (cos, 1/|z|, |h|),
// row_loss[r] = 1 - cos.
__device__ inline void cos_rows_body(const float* __restrict__ z, const float* __restrict__ h, int64_t rows, int D,
                                     float* __restrict__ save, double* __restrict__ row_loss, unsigned int* counter,
                                     double mul, float* __restrict__ loss, int b, int nb, double* red) {
  const int lane = threadIdx.x & 63;
  const int64_t r = static_cast<int64_t>(b) * 4 + (threadIdx.x >> 6);
  if (r < rows) {
    float zz = 0.f, hh = 0.f, zh = 0.f;
    for (int c = lane; c < D; c += 64) {
      const float a = z[r * D + c], bb = h[r * D + c];
      zz += a * a; hh += bb * bb; zh += a * bb;
    }
    zz = wave_sum(zz); hh = wave_sum(hh); zh = wave_sum(zh);
    const float nz = sqrtf(zz), nh = sqrtf(hh);
    const float cz = fmaxf(nz, kNormEps), ch = fmaxf(nh, kNormEps);
    const float cosv = zh / (cz * ch);
    if (lane == 0) {
      save[r * 3 + 0] = cosv; save[r * 3 + 1] = 1.0f / cz; save[r * 3 + 2] = nh;
      st_agent(row_loss + r, static_cast<double>(1.0f - cosv));
      wait_stores();
    }
  }
  if (ticket_last(counter, nb)) finish_sum_block(row_loss, static_cast<int>(rows), mul, 0.0, loss, red);
}
__global__ void __launch_bounds__(256) k_cos_rows(const float* __restrict__ z, const float* __restrict__ h, int64_t rows,
                                                  int D, float* __restrict__ save, double* __restrict__ row_loss,
                                                  unsigned int* counter, double mul, float* __restrict__ loss) {
  __shared__ double red[256];
  cos_rows_body(z, h, rows, D, save, row_loss, counter, mul, loss, blockIdx.x, gridDim.x, red);
}

// g_h[r] = -(g scale / rows) * d cos / d h,  cos = <zn, h / max(|h|, eps)>
__device__ inline void cos_bwd_body(const float* __restrict__ z, const float* __restrict__ h, int64_t rows, int D,
                                    float scale, const float* __restrict__ g, const float* __restrict__ save,
                                    float* __restrict__ gh, int b) {
  const int lane = threadIdx.x & 63;
  const int64_t r = static_cast<int64_t>(b) * 4 + (threadIdx.x >> 6);
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
__global__ void __launch_bounds__(256) k_cos_bwd(const float* __restrict__ z, const float* __restrict__ h, int64_t rows,
                                                 int D, float scale, const float* __restrict__ g,
                                                 const float* __restrict__ save, float* __restrict__ gh) {
  cos_bwd_body(z, h, rows, D, scale, g, save, gh, blockIdx.x);
}

// The heads phase's three losses in one launch each way: blocks [0, ba) the first mean squared error, [ba, ba + bb)
// the second, the rest the cosine rows; every reduction keeps its own partials, ticket and summation order.
struct HeadLossTable {
  HeadLossJobs j;
  double *part_a, *part_b, *part_c;
  unsigned int *cnt_a, *cnt_b, *cnt_c;
  int ba, bb, bc, bz;
};
__global__ void __launch_bounds__(256) k_head_losses_fwd(const HeadLossTable t) {
  __shared__ double red[256];
  const int b = blockIdx.x;
  if (b < t.ba)
    mse_partial_body(t.j.pred_a, t.j.tgt_a, t.j.n_a, t.part_a, t.cnt_a, 1.0 / static_cast<double>(t.j.n_a), t.j.loss_a, b,
                     t.ba, red);
  else if (b < t.ba + t.bb)
    mse_partial_body(t.j.pred_b, t.j.tgt_b, t.j.n_b, t.part_b, t.cnt_b, 1.0 / static_cast<double>(t.j.n_b), t.j.loss_b,
                     b - t.ba, t.bb, red);
  else
    cos_rows_body(t.j.z, t.j.h, t.j.rows, static_cast<int>(t.j.D), t.j.cos_save, t.part_c, t.cnt_c,
                  1.0 / static_cast<double>(t.j.rows), t.j.loss_c, b - t.ba - t.bb, t.bc, red);
}
__global__ void __launch_bounds__(256) k_head_losses_bwd(const HeadLossTable t) {
  const int b = blockIdx.x;
  if (b < t.ba) mse_bwd_body(t.j.pred_a, t.j.tgt_a, t.j.n_a, 1.0f, t.j.g_a, t.j.gp_a, b, t.ba);
  else if (b < t.ba + t.bb) mse_bwd_body(t.j.pred_b, t.j.tgt_b, t.j.n_b, 1.0f, t.j.g_b, t.j.gp_b, b - t.ba, t.bb);
  else if (b < t.ba + t.bb + t.bc)
    cos_bwd_body(t.j.z, t.j.h, t.j.rows, static_cast<int>(t.j.D), 1.0f, t.j.g_c, t.j.cos_save, t.j.gh, b - t.ba - t.bb);
  else {  // the clearing job: plain 16-byte stores, grid-stride over its bz blocks
    uint4* zp = reinterpret_cast<uint4*>(t.j.zero_ptr);
    const int64_t n16 = t.j.zero_bytes / 16;
    for (int64_t i = static_cast<int64_t>(b - t.ba - t.bb - t.bc) * 256 + threadIdx.x; i < n16;
         i += static_cast<int64_t>(t.bz) * 256)
      zp[i] = make_uint4(0u, 0u, 0u, 0u);
  }
}

// Orthogonal regulariser on embed[:, ids] ([H, M, Dc] selected codes).  One WAVE per (head, i):
//   c_i = e_i / max(|e_i|, eps);  g_ij = <c_i, c_j>;  loss = scale * (sum_h sum_ij g_ij^2 / (H M^2) - 1/M)
// forward: partial[h*M + i] = sum_j g_ij^2.
// backward: d/dc_i = 4/(H M^2) sum_j g_ij c_j, pushed through the normalisation of e_i and written to
// row ids[i] of the dense codebook gradient (zero elsewhere; ids are distinct).
constexpr int kOrthoMaxPerLane = 16;  // Dc <= 1024

// One wave per (head, selected code i); Q = columns per lane (Dc <= 64 Q).  The loop over the other codes j is
// latency-bound (id -> row -> two wave reductions per j), so the ids of a 64-wide chunk are read at once and four
// rows are in flight per round.
template <bool BWD, int Q>
__global__ void __launch_bounds__(64)
k_ortho(const float* __restrict__ embed, const int64_t* __restrict__ ids, int H, int K, int Dc, int M, float scale,
        const float* __restrict__ g, double* __restrict__ partial /*[H*M]*/, float* __restrict__ g_embed) {
  constexpr int U = 4;
  const int h = blockIdx.x / M, i = blockIdx.x - h * M, lane = threadIdx.x;
  const float* eh = embed + static_cast<int64_t>(h) * K * Dc;
  const float* ei = eh + ids[i] * Dc;
  float ci[Q], gc[Q];
  float ss = 0.f;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int c = lane + 64 * q;
    ci[q] = c < Dc ? ei[c] : 0.f;
    gc[q] = 0.f;
    ss += ci[q] * ci[q];
  }
  ss = wave_sum(ss);
  const float inv_i = 1.0f / fmaxf(sqrtf(ss), kNormEps);
#pragma unroll
  for (int q = 0; q < Q; ++q) ci[q] *= inv_i;
  double acc = 0.0;
  for (int j0 = 0; j0 < M; j0 += 64) {
    const int cnt = min(64, M - j0);
    const int my_id = lane < cnt ? static_cast<int>(ids[j0 + lane]) : 0;
    for (int jj = 0; jj < cnt; jj += U) {
      float ej_v[U][Q];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int id = __shfl(my_id, jj + u < cnt ? jj + u : jj, 64);  // clamp: duplicate row, masked below
        const float* ej = eh + static_cast<int64_t>(id) * Dc;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const int c = lane + 64 * q;
          ej_v[u][q] = c < Dc ? ej[c] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float sj = 0.f, dij = 0.f;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          sj += ej_v[u][q] * ej_v[u][q];
          dij += ci[q] * ej_v[u][q];
        }
        sj = wave_sum(sj);
        dij = wave_sum(dij);
        if (jj + u < cnt) {  // wave-uniform
          const float inv_j = 1.0f / fmaxf(sqrtf(sj), kNormEps);
          const float gij = dij * inv_j;
          acc += static_cast<double>(gij) * gij;
          if (BWD) {
#pragma unroll
            for (int q = 0; q < Q; ++q) gc[q] += gij * ej_v[u][q] * inv_j;
          }
        }
      }
    }
  }
  if (!BWD) {
    if (lane == 0) partial[blockIdx.x] = acc;
  } else {
    const float coef = g[0] * scale * 4.0f / (static_cast<float>(H) * M * M);
    float dotp = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) dotp += gc[q] * ci[q];
    dotp = wave_sum(dotp);
    float* ge = g_embed + (static_cast<int64_t>(h) * K + ids[i]) * Dc;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int c = lane + 64 * q;
      if (c < Dc) ge[c] = coef * (gc[q] - ci[q] * dotp) * inv_i;
    }
  }
}

// The same regulariser for the usual small selection (M <= 32 codes, Dc <= 256): ONE block per head keeps the
// normalised selected rows in LDS, forms the M x M Gram matrix with all its threads and -- backward -- pushes
// G C through the normalisation; it also clears the head's slice of the dense codebook gradient itself (no memset
// launch).  The wave-per-row kernel above walks 32 dependent row pairs per wave: 15 us forward and backward; this one
// is a few microseconds.  forward: partial[h][b] = sum over the block's rows i and all j of g_ij^2.
template <bool BWD>
__global__ void __launch_bounds__(256)
k_ortho_block(const float* __restrict__ embed, const int64_t* __restrict__ ids, int H, int K, int Dc, int M, float scale,
              const float* __restrict__ g, double* __restrict__ partial /*[H][4]*/, float* __restrict__ g_embed,
              unsigned int* counter = nullptr, double mul = 0.0, double add = 0.0, float* __restrict__ loss = nullptr,
              const float* __restrict__ plus = nullptr, float* __restrict__ total = nullptr) {
  // grid (heads, 4): block (h, b) owns selected rows 8 b .. 8 b + 7 -- a row per 32 lanes -- against all M rows
  constexpr int kMaxM = 32, kMaxD = 256;
  __shared__ __attribute__((aligned(16))) float cs[kMaxM][kMaxD + 4];  // normalised rows
  __shared__ float gm[8][kMaxM + 1];
  __shared__ double red[256];
  const int h = blockIdx.x, tid = threadIdx.x;
  const float* eh = embed + static_cast<int64_t>(h) * K * Dc;
  const int nv = Dc / 4;  // float4 per row (<= 64)
  if (BWD) {
    // clear this block's quarter of the head's slice of the dense gradient -- except the selected rows, which some
    // block of the head writes (no order between blocks: a cleared row must never be a written one)
    __shared__ int s_ids[kMaxM];
    if (tid < M) s_ids[tid] = static_cast<int>(ids[tid]);
    __syncthreads();
    float4* gz = reinterpret_cast<float4*>(g_embed + static_cast<int64_t>(h) * K * Dc);
    const int64_t total = static_cast<int64_t>(K) * nv, q0 = total * blockIdx.y / 4, q1 = total * (blockIdx.y + 1) / 4;
    for (int64_t q = q0 + tid; q < q1; q += 256) {
      const int row = static_cast<int>(q / nv);
      bool selected = false;
      for (int j = 0; j < M; ++j) selected |= s_ids[j] == row;
      if (!selected) gz[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  {  // all M rows, normalised: 8 threads per row
    const int r = tid >> 3, part = tid & 7;
    float ss = 0.f;
    if (r < M) {
      const float* er = eh + ids[r] * Dc;
      for (int v = part; v < nv; v += 8) {
        const float4 a = *reinterpret_cast<const float4*>(er + 4 * v);
        *reinterpret_cast<float4*>(&cs[r][4 * v]) = a;
        ss += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
      }
    }
    ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64);
    const float inv = 1.0f / fmaxf(sqrtf(ss), kNormEps);
    if (r < M) {
      for (int v = part; v < nv; v += 8) {
        float4 a = *reinterpret_cast<float4*>(&cs[r][4 * v]);
        a.x *= inv; a.y *= inv; a.z *= inv; a.w *= inv;
        *reinterpret_cast<float4*>(&cs[r][4 * v]) = a;
      }
      if (part == 0) cs[r][kMaxD] = inv;  // the spare column keeps 1 / |e_r|
    }
  }
  __syncthreads();
  const int il = tid >> 5, lane = tid & 31;  // local row, lane of the row's 32
  const int i = blockIdx.y * 8 + il;
  // Gram entry (i, j = lane), fixed order over the columns
  double acc = 0.0;
  if (i < M && lane < M) {
    float d = 0.f;
    for (int v = 0; v < nv; ++v) {
      const float4 a = *reinterpret_cast<const float4*>(&cs[i][4 * v]);
      const float4 b = *reinterpret_cast<const float4*>(&cs[lane][4 * v]);
      d += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    gm[il][lane] = d;
    acc = static_cast<double>(d) * d;
  }
  if (!BWD) {
    red[tid] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    if (tid == 0) { st_agent(partial + h * 4 + blockIdx.y, red[0]); wait_stores(); }
    // the block that arrives last adds the 4 H block sums in index order and writes the loss (and, for the quantiser's
    // phase, the sum with the commitment term): no finishing launch
    if (!ticket_last(counter)) return;
    if (tid == 0) {
      double s = 0.0;
      for (int i = 0; i < 4 * H; ++i) s += ld_agent(partial + i);
      const float v = static_cast<float>(s * mul + add);
      loss[0] = v;
      if (total) total[0] = (plus ? plus[0] : 0.f) + v;
    }
    return;
  }
  __syncthreads();
  if (i >= M) return;
  // d loss / d c_i = 4 scale / (H M^2) sum_j g_ij c_j, through c_i = e_i / |e_i|; a lane holds columns 4 lane (+ 128)
  const float coef = g[0] * scale * 4.0f / (static_cast<float>(H) * M * M);
  float4 gc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  float dotp = 0.f;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int v = lane + 32 * u;
    if (v < nv) {
      for (int j = 0; j < M; ++j) {
        const float w = gm[il][j];
        const float4 b = *reinterpret_cast<const float4*>(&cs[j][4 * v]);
        gc[u].x += w * b.x; gc[u].y += w * b.y; gc[u].z += w * b.z; gc[u].w += w * b.w;
      }
      const float4 c = *reinterpret_cast<const float4*>(&cs[i][4 * v]);
      dotp += gc[u].x * c.x + gc[u].y * c.y + gc[u].z * c.z + gc[u].w * c.w;
    }
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) dotp += __shfl_xor(dotp, o, 64);
  const float inv = cs[i][kMaxD];
  float* ge = g_embed + (static_cast<int64_t>(h) * K + ids[i]) * Dc;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int v = lane + 32 * u;
    if (v < nv) {
      const float4 c = *reinterpret_cast<const float4*>(&cs[i][4 * v]);
      *reinterpret_cast<float4*>(ge + 4 * v) =
          make_float4(coef * (gc[u].x - c.x * dotp) * inv, coef * (gc[u].y - c.y * dotp) * inv,
                      coef * (gc[u].z - c.z * dotp) * inv, coef * (gc[u].w - c.w * dotp) * inv);
    }
  }
}

inline bool ortho_block_ok(int M, int Dc) { return M >= 4 && M <= 32 && Dc <= 256 && Dc % 4 == 0; }

template <bool BWD>
void launch_ortho(int grid, hipStream_t st, const float* embed, const int64_t* ids, int H, int K, int Dc, int M,
                  float scale, const float* g, double* partial, float* g_embed) {
  const int q = (Dc + 63) / 64;
  if (q <= 2) k_ortho<BWD, 2><<<grid, 64, 0, st>>>(embed, ids, H, K, Dc, M, scale, g, partial, g_embed);
  else if (q <= 4) k_ortho<BWD, 4><<<grid, 64, 0, st>>>(embed, ids, H, K, Dc, M, scale, g, partial, g_embed);
  else if (q <= 8) k_ortho<BWD, 8><<<grid, 64, 0, st>>>(embed, ids, H, K, Dc, M, scale, g, partial, g_embed);
  else k_ortho<BWD, 16><<<grid, 64, 0, st>>>(embed, ids, H, K, Dc, M, scale, g, partial, g_embed);
}

// total = sum_i w_i * term_i over up to 8 scalar loss terms living in separate device buffers (reference
// pretrain.py:51-58), and its backward g_i = g * w_i, one launch each.
struct ScalarTable {
  const float* p[8];
  float w[8];
  int32_t count;
};

__global__ void k_weighted_sum(ScalarTable t, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s = 0.f;
  for (int i = 0; i < t.count; ++i) s += t.w[i] * t.p[i][0];  // fixed order
  out[0] = s;
}

__global__ void k_weighted_sum_bwd(ScalarTable t, const float* __restrict__ g, float* __restrict__ g_terms) {
  const int i = threadIdx.x;
  if (blockIdx.x == 0 && i < t.count) g_terms[i] = g[0] * t.w[i];
}

}  // namespace
}  // namespace stemgnn

namespace stemgnn {
// Per-(device, stream) scratch of the "last block finishes" reductions (common.h): kTicketSlots zero-initialised counter
// words and 512 doubles for partial sums.  Kernels of one stream run one after the other and every counter is back at
// zero when its kernel ends, so all reductions enqueued on a stream share its words; two streams never share any.
namespace {
struct StreamScratch {
  unsigned int* counters = nullptr;
  double* partials = nullptr;
};
std::mutex g_scratch_mu;
std::map<std::pair<int, hipStream_t>, StreamScratch> g_scratch;

StreamScratch stream_scratch(hipStream_t st) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return StreamScratch{};
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  auto it = g_scratch.find({dev, st});
  if (it != g_scratch.end()) return it->second;
  // not cached on failure: the next call tries again
  constexpr size_t kCounterBytes = 256, kBytes = kCounterBytes + 512 * sizeof(double);
  static_assert(kTicketSlots * sizeof(unsigned int) <= kCounterBytes, "counter words fit their line pair");
  unsigned char* p = nullptr;
  if (hipMalloc(&p, kBytes) != hipSuccess) return StreamScratch{};
  if (hipMemset(p, 0, kBytes) != hipSuccess) {
    (void)hipFree(p);
    return StreamScratch{};
  }
  StreamScratch s;
  s.counters = reinterpret_cast<unsigned int*>(p);
  s.partials = reinterpret_cast<double*>(p + kCounterBytes);
  g_scratch[{dev, st}] = s;
  return s;
}
}  // namespace

unsigned int* ticket_counter(hipStream_t st, int slot) {
  if (slot < 0 || slot >= kTicketSlots) return nullptr;
  const StreamScratch s = stream_scratch(st);
  return s.counters ? s.counters + slot : nullptr;
}
double* stream_partials(hipStream_t st) { return stream_scratch(st).partials; }

static inline int mse_blocks(int64_t n) {
  int64_t b = (n / 4 + 255) / 256;
  return static_cast<int>(b < 1 ? 1 : (b > 256 ? 256 : b));
}
static inline int mse_bwd_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  return static_cast<int>(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}
int head_losses_fwd(const HeadLossJobs& j, hipStream_t st) {
  if (j.n_a <= 0 || j.n_b <= 0 || j.rows <= 0 || j.D <= 0 || j.rows >= (1 << 30)) return STEMGNN_ERR_INVALID_ARG;
  HeadLossTable t;
  t.j = j;
  t.part_a = reinterpret_cast<double*>(align_up(reinterpret_cast<uintptr_t>(j.ws_a), 256));
  t.part_b = reinterpret_cast<double*>(align_up(reinterpret_cast<uintptr_t>(j.ws_b), 256));
  t.part_c = reinterpret_cast<double*>(align_up(reinterpret_cast<uintptr_t>(j.ws_c), 256));
  t.cnt_a = ticket_counter(st, 0);  // three reductions in one launch: three words of the stream's pool
  t.cnt_b = ticket_counter(st, 1);
  t.cnt_c = ticket_counter(st, 2);
  if (!t.cnt_a || !t.cnt_b || !t.cnt_c) return STEMGNN_ERR_HIP;
  t.ba = mse_blocks(j.n_a);
  t.bb = mse_blocks(j.n_b);
  t.bc = static_cast<int>((j.rows + 3) / 4);
  t.bz = 0;
  k_head_losses_fwd<<<static_cast<unsigned>(t.ba + t.bb + t.bc), 256, 0, st>>>(t);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}
int head_losses_bwd(const HeadLossJobs& j, hipStream_t st) {
  if (j.n_a <= 0 || j.n_b <= 0 || j.rows <= 0 || j.D <= 0) return STEMGNN_ERR_INVALID_ARG;
  HeadLossTable t;
  t.j = j;
  t.part_a = t.part_b = t.part_c = nullptr;
  t.cnt_a = t.cnt_b = t.cnt_c = nullptr;
  t.ba = mse_bwd_blocks(j.n_a);
  t.bb = mse_bwd_blocks(j.n_b);
  t.bc = static_cast<int>((j.rows + 3) / 4);
  t.bz = 0;
  if (j.zero_ptr && j.zero_bytes > 0) {
    if (j.zero_bytes % 16 != 0 || reinterpret_cast<uintptr_t>(j.zero_ptr) % 16 != 0) return STEMGNN_ERR_INVALID_ARG;
    const int64_t b = (j.zero_bytes / 16 + 255) / 256;
    t.bz = static_cast<int>(b > 2048 ? 2048 : b);
  }
  k_head_losses_bwd<<<static_cast<unsigned>(t.ba + t.bb + t.bc + t.bz), 256, 0, st>>>(t);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}
}  // namespace stemgnn

using namespace stemgnn;

extern "C" {

size_t stemgnn_loss_workspace_bytes(int64_t n) {
  // mse: <= 256 block partials; cosine: one double per row; ortho: one per (head, id)
  return static_cast<size_t>(n < 256 ? 256 : n) * sizeof(double) + 512;
}

int stemgnn_weighted_sum(const float* const* terms, const float* weights, int32_t count, float* out, void* stream_) {
  if (count < 0 || count > 8 || !out || (count > 0 && (!terms || !weights))) return STEMGNN_ERR_INVALID_ARG;
  ScalarTable t;
  t.count = count;
  for (int i = 0; i < count; ++i) {
    if (!terms[i]) return STEMGNN_ERR_INVALID_ARG;
    t.p[i] = terms[i];
    t.w[i] = weights[i];
  }
  k_weighted_sum<<<1, 64, 0, static_cast<hipStream_t>(stream_)>>>(t, out);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_weighted_sum_bwd(const float* weights, int32_t count, const float* g_out, float* g_terms, void* stream_) {
  if (count < 0 || count > 8 || (count > 0 && (!weights || !g_out || !g_terms))) return STEMGNN_ERR_INVALID_ARG;
  if (count == 0) return STEMGNN_OK;
  ScalarTable t;
  t.count = count;
  for (int i = 0; i < count; ++i) { t.p[i] = nullptr; t.w[i] = weights[i]; }
  k_weighted_sum_bwd<<<1, 64, 0, static_cast<hipStream_t>(stream_)>>>(t, g_out, g_terms);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_mse_loss_fwd(const float* pred, const float* target, int64_t n, float scale, float* loss, void* workspace,
                         size_t workspace_bytes, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (n < 0 || !loss || !workspace || (n > 0 && (!pred || !target))) return STEMGNN_ERR_INVALID_ARG;
  if (workspace_bytes < stemgnn_loss_workspace_bytes(256)) return STEMGNN_ERR_WORKSPACE;
  double* partial = reinterpret_cast<double*>(align_up(reinterpret_cast<uintptr_t>(workspace), 256));
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256) blocks = 256;
  unsigned int* counter = ticket_counter(st);
  if (!counter) return STEMGNN_ERR_HIP;
  k_mse_partial<<<static_cast<unsigned>(blocks), 256, 0, st>>>(pred, target, n, partial, counter,
                                                               static_cast<double>(scale) / (n > 0 ? n : 1), loss);
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
                            float* save, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (rows < 0 || dim <= 0 || !loss || !workspace || (rows > 0 && (!z || !h || !save))) return STEMGNN_ERR_INVALID_ARG;
  if (rows >= (1 << 30)) return STEMGNN_ERR_TOO_LARGE;
  if (workspace_bytes < stemgnn_loss_workspace_bytes(rows)) return STEMGNN_ERR_WORKSPACE;
  double* row_loss = reinterpret_cast<double*>(align_up(reinterpret_cast<uintptr_t>(workspace), 256));
  if (rows > 0) {
    unsigned int* counter = ticket_counter(st);
    if (!counter) return STEMGNN_ERR_HIP;
    k_cos_rows<<<static_cast<unsigned>((rows + 3) / 4), 256, 0, st>>>(z, h, rows, static_cast<int>(dim), save, row_loss,
                                                                      counter, static_cast<double>(scale) / rows, loss);
    STEMGNN_LAUNCH_CHECK();
  } else {
    k_finish_sum<<<1, 256, 0, st>>>(row_loss, 0, 0.0, 0.0, loss);
    STEMGNN_LAUNCH_CHECK();
  }
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

}  // extern "C"

// the regulariser, and optionally total[0] = plus[0] + loss[0] from the same finishing launch (csrc/phases.hip)
int stemgnn::ortho_loss_fwd_plus(const float* embed, const int64_t* ids, int64_t heads, int64_t codebook_size,
                                 int64_t code_dim, int64_t num_ids, float scale, float* loss, const float* plus,
                                 float* total, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (heads <= 0 || codebook_size <= 0 || code_dim <= 0 || code_dim > 64 * kOrthoMaxPerLane || num_ids <= 0 ||
      heads * num_ids > (1 << 20) || !embed || !ids || !loss || !workspace)
    return STEMGNN_ERR_INVALID_ARG;
  if (workspace_bytes < stemgnn_loss_workspace_bytes(heads * num_ids)) return STEMGNN_ERR_WORKSPACE;
  double* partial = reinterpret_cast<double*>(align_up(reinterpret_cast<uintptr_t>(workspace), 256));
  const int H = static_cast<int>(heads), M = static_cast<int>(num_ids);
  int nparts = H * M;
  if (ortho_block_ok(M, static_cast<int>(code_dim))) {
    unsigned int* counter = ticket_counter(st);
    if (!counter) return STEMGNN_ERR_HIP;
    k_ortho_block<false><<<dim3(H, 4), 256, 0, st>>>(embed, ids, H, static_cast<int>(codebook_size),
                                                     static_cast<int>(code_dim), M, scale, nullptr, partial, nullptr, counter,
                                                     static_cast<double>(scale) / (static_cast<double>(H) * M * M),
                                                     -static_cast<double>(scale) / M, loss, plus, total);
    STEMGNN_LAUNCH_CHECK();
    return STEMGNN_OK;
  } else {
    launch_ortho<false>(H * M, st, embed, ids, H, static_cast<int>(codebook_size), static_cast<int>(code_dim), M, scale,
                        nullptr, partial, nullptr);
  }
  STEMGNN_LAUNCH_CHECK();
  k_finish_sum<<<1, 256, 0, st>>>(partial, nparts, static_cast<double>(scale) / (static_cast<double>(H) * M * M),
                                  -static_cast<double>(scale) / M, loss, plus, total);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

extern "C" {

int stemgnn_ortho_loss_fwd(const float* embed, const int64_t* ids, int64_t heads, int64_t codebook_size,
                           int64_t code_dim, int64_t num_ids, float scale, float* loss, void* workspace,
                           size_t workspace_bytes, void* stream_) {
  return ortho_loss_fwd_plus(embed, ids, heads, codebook_size, code_dim, num_ids, scale, loss, nullptr, nullptr, workspace,
                             workspace_bytes, stream_);
}

int stemgnn_ortho_loss_bwd(const float* embed, const int64_t* ids, int64_t heads, int64_t codebook_size,
                           int64_t code_dim, int64_t num_ids, float scale, const float* g_loss, float* g_embed,
                           void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (heads <= 0 || codebook_size <= 0 || code_dim <= 0 || code_dim > 64 * kOrthoMaxPerLane || num_ids <= 0 ||
      !embed || !ids || !g_loss || !g_embed)
    return STEMGNN_ERR_INVALID_ARG;
  const int H = static_cast<int>(heads), M = static_cast<int>(num_ids);
  if (ortho_block_ok(M, static_cast<int>(code_dim))) {
    k_ortho_block<true><<<dim3(H, 4), 256, 0, st>>>(embed, ids, H, static_cast<int>(codebook_size), static_cast<int>(code_dim), M,
                                           scale, g_loss, nullptr, g_embed);
    STEMGNN_LAUNCH_CHECK();
    return STEMGNN_OK;
  }
  STEMGNN_HIP_TRY(hipMemsetAsync(g_embed, 0, sizeof(float) * heads * codebook_size * code_dim, st));
  launch_ortho<true>(H * M, st, embed, ids, H, static_cast<int>(codebook_size), static_cast<int>(code_dim), M, scale,
                     g_loss, nullptr, g_embed);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

}  // extern "C"
