// K3 / K5: dense per-layer projections on the fp32 matrix cores of gfx950.
//
// Reference: MySAGEConv's lin_l(agg) + lin_r(x) (STEM-GNN/model/encoder.py:83-87),
// VectorQuantize.project_in / project_out (model/vq.py:881,1041) and the decoders' nn.Linear
// layers (model/pt_model.py:42,80,94; model/encoder.py:364).  The reference leaves these to
// ATen/cuBLAS plus separate bias-add, add and bias-gradient reduction kernels.  The activations
// here are tall and skinny (M ~ 1e5 rows, N, K in {128, 256, 512}), a shape the vendor GEMM
// library serves badly on this chip (12-30 TFLOP/s for the weight-gradient product), so the
// three products are hand-written around v_mfma_f32_32x32x2_f32 (exact fp32; the reference runs
// fp32 without autocast):
//
//   forward     Y[M,N]  = X1[M,K1] W1[N,K1]^T (+ X2[M,K2] W2[N,K2]^T) + b      (fused K-concat)
//                         optional per-column sum / sum-of-squares partials of Y (BatchNorm stats)
//   backward-W  dW[N,K] = dY[M,N]^T X[M,K],  db[N] = colsum(dY)   (split over M, two-stage,
//                         deterministic: no atomics)
//   backward-X  dX[M,K] = dY[M,N] W[N,K]  = forward with the transposed weight
//
// Tiling: 256-thread block = 4 waves in a 2x2 grid, 128x128 output tile, each wave 64x64 =
// 2x2 MFMA tiles (64 accumulator registers); K (or M for backward-W) is consumed in chunks of
// 32 staged through LDS with register prefetch of the next chunk.  MFMA-bound:
// 2*M*N*K flop at 157 TFLOP/s vs (M*K + M*N)*4 bytes of HBM traffic.
#include "common.h"

#include <cstdlib>

namespace stemgnn {
namespace {

constexpr int kBlock = 256;
constexpr int kBM = 128, kBN = 128, kKC = 32;
constexpr int kLd = kKC + 4;  // 36-dword row stride: conflict-free ds_read_b128 of 32 rows x 16 B
constexpr int kMaxSplits = 512;

typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ inline float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ inline void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ inline float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// C-tile register r of lane (lj, hi) is row (r&3) + 8*(r>>2) + 4*hi, column lj.
__device__ inline int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// ---------------------------------------------------------------------------------------
// forward: Y = X1 W1^T (+ X2 W2^T) + b
// ---------------------------------------------------------------------------------------
template <bool STATS>
__global__ void __launch_bounds__(kBlock, 2)
k_linear_fwd(const float* __restrict__ x1, const float* __restrict__ w1, int K1, const float* __restrict__ x2,
             const float* __restrict__ w2, int K2, const float* __restrict__ bias, int64_t M, int N,
             float* __restrict__ y, float* __restrict__ stats_partial /*[gridDim.x][2][N]*/) {
  __shared__ __attribute__((aligned(16))) float sA[kBM * kLd];
  __shared__ __attribute__((aligned(16))) float sB[kBN * kLd];
  __shared__ float s_stats[2][2][kBN];  // [wave_m][sum|sumsq][n]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, hi = lane >> 5, lj = lane & 31;
  const int64_t m0 = static_cast<int64_t>(blockIdx.x) * kBM;
  const int n0 = blockIdx.y * kBN;
  const int c1 = (K1 + kKC - 1) / kKC, c2 = (K2 + kKC - 1) / kKC;
  const int steps = c1 + c2;

  float4 ra[4], rb[4];
  auto fetch = [&](int step) {
    const bool second = step >= c1;
    const float* xs = second ? x2 : x1;
    const float* ws = second ? w2 : w1;
    const int K = second ? K2 : K1;
    const int k0 = (second ? step - c1 : step) * kKC;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = t * kBlock + tid;  // 1024 float4 per operand chunk: row = idx / 8, 16-byte column = idx % 8
      const int r = idx >> 3, k = k0 + 4 * (idx & 7);
      const int64_t m = m0 + r;
      const int n = n0 + r;
      ra[t] = (m < M && k < K) ? ld4(xs + m * K + k) : zero4();
      rb[t] = (n < N && k < K) ? ld4(ws + static_cast<int64_t>(n) * K + k) : zero4();
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = t * kBlock + tid;
      st4(sA + (idx >> 3) * kLd + 4 * (idx & 7), ra[t]);
      st4(sB + (idx >> 3) * kLd + 4 * (idx & 7), rb[t]);
    }
  };

  floatx16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float bias_v[2];
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int n = n0 + wn * 64 + tn * 32 + lj;
    bias_v[tn] = (bias != nullptr && n < N) ? bias[n] : 0.f;
  }

  fetch(0);
  for (int step = 0; step < steps; ++step) {
    stash();
    __syncthreads();
    if (step + 1 < steps) fetch(step + 1);
#pragma unroll
    for (int ms = 0; ms < kKC / 8; ++ms) {
      const int ko = ms * 8 + hi * 4;  // lane half 0 takes k 0..3, half 1 takes k 4..7 of the micro-step
      float4 a[2], b[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = ld4(sA + (wm * 64 + t * 32 + lj) * kLd + ko);
        b[t] = ld4(sB + (wn * 64 + t * 32 + lj) * kLd + ko);
      }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].x, b[tn].x, acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].y, b[tn].y, acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].z, b[tn].z, acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].w, b[tn].w, acc[tm][tn], 0, 0, 0);
        }
    }
    __syncthreads();
  }

  // ---- epilogue: bias, store (two 128-byte row segments per store instruction), column stats.
  // The bias was loaded before the main loop: the epilogue issues stores only, so no
  // s_waitcnt vmcnt(0) ever serialises them (vmcnt counts stores too).
  const bool interior = (m0 + kBM <= M) && (n0 + kBN <= N);
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int nl = wn * 64 + tn * 32 + lj;
    const int n = n0 + nl;
    const float bv = bias_v[tn];
    float s1 = 0.f, s2 = 0.f;
    if (interior) {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        float* yp = y + (m0 + wm * 64 + tm * 32 + 4 * hi) * N + n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[tm][tn][r] + bv;
          yp[static_cast<int64_t>((r & 3) + 8 * (r >> 2)) * N] = v;
          if (STATS) { s1 += v; s2 += v * v; }
        }
      }
    } else {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t m = m0 + wm * 64 + tm * 32 + acc_row(r, hi);
          const float v = acc[tm][tn][r] + bv;
          if (m < M && n < N) {
            y[m * N + n] = v;
            if (STATS) { s1 += v; s2 += v * v; }
          }
        }
      }
    }
    if (STATS) {
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (hi == 0) { s_stats[wm][0][nl] = s1; s_stats[wm][1][nl] = s2; }
    }
  }
  if (STATS) {
    __syncthreads();
    if (tid < kBN && n0 + tid < N) {
      float* p = stats_partial + static_cast<int64_t>(blockIdx.x) * 2 * N;
      p[n0 + tid] = s_stats[0][0][tid] + s_stats[1][0][tid];
      p[N + n0 + tid] = s_stats[0][1][tid] + s_stats[1][1][tid];
    }
  }
}

// ---------------------------------------------------------------------------------------
// forward, weight-resident variant (K1 + K2 <= 256): the 128-column weight tile stays in LDS
// for the whole life of a PERSISTENT block (one 512-thread block per CU, 2 waves per SIMD),
// which then only streams activation chunks: BM x 32 floats per step through a double-buffered
// LDS ring with ONE barrier per step and the next chunk's global loads in flight behind the
// MFMAs, continuously across row-tile boundaries (no per-tile prologue bubble).
//   BM = 128 (K <= 128): waves 4(m) x 2(n), 32x64 per wave;  BM = 64 (K <= 256): 2 x 4, 32x32.
// ---------------------------------------------------------------------------------------
constexpr int kResThreads = 512;
// Measured on MI355X (tools/kbench.py linear, M = 102400): 55-56 us vs 45-47 us (K = 128) and 92 us vs
// 81 us (K = 256) for the 3-blocks-per-CU tile kernel above, with or without a 3-deep prefetch
// ring: kept for reference and further tuning, not dispatched.
constexpr bool kUseResidentVariant = false;

template <int BM, bool STATS>
__global__ void __launch_bounds__(kResThreads)
k_linear_fwd_res(const float* __restrict__ x1, const float* __restrict__ w1, int K1, const float* __restrict__ x2,
                 const float* __restrict__ w2, int K2, const float* __restrict__ bias, int64_t M, int N,
                 float* __restrict__ y, float* __restrict__ stats_partial /*[ceil(M/BM)][2][N]*/) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int WM = BM / 32;          // waves along m (each wave owns 32 rows)
  constexpr int WN = 8 / WM;           // waves along n
  constexpr int TN = 4 / WN;           // 32-column accumulator tiles per wave
  constexpr int FPT = BM * 8 / kResThreads;  // float4 of an activation chunk per thread

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, hi = lane >> 5, lj = lane & 31;
  const int Ktot = K1 + K2;
  const int chunks = (Ktot + kKC - 1) / kKC;
  const int ldb = chunks * kKC + 4;
  float* sB = smem;                               // [128][ldb]
  float* sA = sB + kBN * ldb;                     // [2][BM][kLd]
  float* s_stats = sA + 2 * BM * kLd;             // [WM][2][128]
  const int n0 = blockIdx.y * kBN;

  // resident weight tile (zero beyond N and beyond Ktot)
  {
    const int vec_per_row = chunks * kKC / 4;
    for (int idx = tid; idx < kBN * vec_per_row; idx += kResThreads) {
      const int r = idx / vec_per_row, k = 4 * (idx - r * vec_per_row);
      const int n = n0 + r;
      float4 v = zero4();
      if (n < N && k < Ktot)
        v = k < K1 ? ld4(w1 + static_cast<int64_t>(n) * K1 + k) : ld4(w2 + static_cast<int64_t>(n) * K2 + (k - K1));
      st4(sB + r * ldb + k, v);
    }
  }
  float bias_v[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + wn * 32 * TN + tn * 32 + lj;
    bias_v[tn] = (bias != nullptr && n < N) ? bias[n] : 0.f;
  }

  const int64_t num_tiles = (M + BM - 1) / BM;
  const int64_t my_tiles = blockIdx.x < num_tiles ? (num_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  const int64_t total = my_tiles * chunks;

  // Register ring three chunks deep: the loads of chunk s+3 are issued while chunk s is being
  // multiplied, so an HBM round trip has three chunk times to land (a single persistent block
  // per CU has no other block to hide it behind).
  auto fetch = [&](int64_t step, float4 (&slot)[FPT]) {
    const int64_t tile = blockIdx.x + (step / chunks) * gridDim.x;
    const int k0 = static_cast<int>(step % chunks) * kKC;
#pragma unroll
    for (int t = 0; t < FPT; ++t) {
      const int idx = t * kResThreads + tid;
      const int r = idx >> 3, k = k0 + 4 * (idx & 7);
      const int64_t m = tile * BM + r;
      float4 v = zero4();
      if (m < M && k < Ktot) v = k < K1 ? ld4(x1 + m * K1 + k) : ld4(x2 + m * K2 + (k - K1));
      slot[t] = v;
    }
  };

  floatx16 acc[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tn][r] = 0.f;

  auto body = [&](int64_t step, float4 (&slot)[FPT]) {
    float* buf = sA + (step & 1) * BM * kLd;
#pragma unroll
    for (int t = 0; t < FPT; ++t) {
      const int idx = t * kResThreads + tid;
      st4(buf + (idx >> 3) * kLd + 4 * (idx & 7), slot[t]);
    }
    __syncthreads();  // the only barrier of the step (also fences the first use of the resident tile)
    if (step + 3 < total) fetch(step + 3, slot);
    const int kc = static_cast<int>(step % chunks);
    const float* bbase = sB + kc * kKC;
#pragma unroll
    for (int ms = 0; ms < kKC / 8; ++ms) {
      const int ko = ms * 8 + hi * 4;
      const float4 a = ld4(buf + (wm * 32 + lj) * kLd + ko);
      float4 b[TN];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) b[tn] = ld4(bbase + (wn * 32 * TN + tn * 32 + lj) * ldb + ko);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[tn].x, acc[tn], 0, 0, 0);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[tn].y, acc[tn], 0, 0, 0);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[tn].z, acc[tn], 0, 0, 0);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[tn].w, acc[tn], 0, 0, 0);
    }
    if (kc == chunks - 1) {
      // ---- tile epilogue (stores only; the bias was loaded up front)
      const int64_t tile = blockIdx.x + (step / chunks) * gridDim.x;
      const int64_t mrow0 = tile * BM + wm * 32 + 4 * hi;
      const bool interior = (tile * BM + BM <= M) && (n0 + kBN <= N);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int nl = wn * 32 * TN + tn * 32 + lj;
        const int n = n0 + nl;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t m = mrow0 + (r & 3) + 8 * (r >> 2);
          const float v = acc[tn][r] + bias_v[tn];
          if (interior || (m < M && n < N)) {
            y[m * N + n] = v;
            if (STATS) { s1 += v; s2 += v * v; }
          }
          acc[tn][r] = 0.f;
        }
        if (STATS) {
          s1 += __shfl_xor(s1, 32, 64);
          s2 += __shfl_xor(s2, 32, 64);
          if (hi == 0) { s_stats[(wm * 2 + 0) * kBN + nl] = s1; s_stats[(wm * 2 + 1) * kBN + nl] = s2; }
        }
      }
      if (STATS) {
        __syncthreads();
        if (tid < kBN && n0 + tid < N) {
          float t1 = 0.f, t2 = 0.f;
#pragma unroll
          for (int w = 0; w < WM; ++w) { t1 += s_stats[(w * 2 + 0) * kBN + tid]; t2 += s_stats[(w * 2 + 1) * kBN + tid]; }
          float* p = stats_partial + tile * 2 * N;
          p[n0 + tid] = t1;
          p[N + n0 + tid] = t2;
        }
      }
    }
  };

  float4 r0[FPT], r1[FPT], r2[FPT];
  if (total > 0) fetch(0, r0);
  if (total > 1) fetch(1, r1);
  if (total > 2) fetch(2, r2);
  for (int64_t step = 0; step < total; step += 3) {
    body(step, r0);
    if (step + 1 < total) body(step + 1, r1);
    if (step + 2 < total) body(step + 2, r2);
  }
}

inline size_t res_lds_bytes(int BM, int Ktot) {
  const int chunks = (Ktot + kKC - 1) / kKC;
  return static_cast<size_t>(kBN * (chunks * kKC + 4) + 2 * BM * kLd + (BM / 32) * 2 * kBN) * sizeof(float);
}

template <int BM, bool STATS>
int launch_res(dim3 grid, hipStream_t st, const float* x1, const float* w1, int K1, const float* x2, const float* w2,
               int K2, const float* bias, int64_t M, int N, float* y, float* stats) {
  const size_t lds = res_lds_bytes(BM, K1 + K2);
  static bool configured = false;  // per instantiation
  if (!configured) {
    STEMGNN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_linear_fwd_res<BM, STATS>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    configured = true;
  }
  k_linear_fwd_res<BM, STATS><<<grid, kResThreads, lds, st>>>(x1, w1, K1, x2, w2, K2, bias, M, N, y, stats);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

// ---------------------------------------------------------------------------------------
// backward-W: partial[s] = dY[rows of split s]^T X[rows of split s]; partial_db[s] = colsum dY
// C[i = n][j = k]; the reduction index (row m) is the slow dimension of BOTH operands, so the
// MFMA fragments are 4-byte LDS reads along a row (conflict-free: 32 consecutive dwords).
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock, 2)
k_linear_bwd_weight(const float* __restrict__ dy, const float* __restrict__ x, int64_t M, int N, int K,
                    int64_t rows_per_split, float* __restrict__ partial_dw /*[S][N][K]*/,
                    float* __restrict__ partial_db /*[S][N]*/) {
  __shared__ __attribute__((aligned(16))) float sA[kKC * kBN];  // dY chunk [32 m][128 n]
  __shared__ __attribute__((aligned(16))) float sB[kKC * kBN];  // X  chunk [32 m][128 k]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1, hi = lane >> 5, lj = lane & 31;
  const int split = blockIdx.x;
  const int n0 = blockIdx.y * kBN, k0 = blockIdx.z * kBN;
  const int64_t mbeg = split * rows_per_split;
  const int64_t mend = min(M, mbeg + rows_per_split);
  const int steps = static_cast<int>((mend - mbeg + kKC - 1) / kKC);

  float4 ra[4], rb[4];
  auto fetch = [&](int step) {
    const int64_t mm = mbeg + static_cast<int64_t>(step) * kKC;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = t * kBlock + tid;  // 32 rows x 32 float4
      const int r = idx >> 5, c = 4 * (idx & 31);
      const int64_t m = mm + r;
      ra[t] = (m < mend && n0 + c < N) ? ld4(dy + m * N + n0 + c) : zero4();
      rb[t] = (m < mend && k0 + c < K) ? ld4(x + m * K + k0 + c) : zero4();
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = t * kBlock + tid;
      st4(sA + (idx >> 5) * kBN + 4 * (idx & 31), ra[t]);
      st4(sB + (idx >> 5) * kBN + 4 * (idx & 31), rb[t]);
    }
  };

  floatx16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float colsum = 0.f;

  if (steps > 0) fetch(0);
  for (int step = 0; step < steps; ++step) {
    stash();
    __syncthreads();
    if (step + 1 < steps) fetch(step + 1);
#pragma unroll 4
    for (int ks = 0; ks < kKC / 2; ++ks) {
      const int row = 2 * ks + hi;
      float a[2], b[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = sA[row * kBN + wi * 64 + t * 32 + lj];
        b[t] = sB[row * kBN + wj * 64 + t * 32 + lj];
      }
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
          acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
    }
    if (blockIdx.z == 0 && tid < kBN) {
#pragma unroll 8
      for (int r = 0; r < kKC; ++r) colsum += sA[r * kBN + tid];
    }
    __syncthreads();
  }

  float* pw = partial_dw + static_cast<int64_t>(split) * N * K;
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const int k = k0 + wj * 64 + tj * 32 + lj;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wi * 64 + ti * 32 + acc_row(r, hi);
        if (n < N && k < K) pw[static_cast<int64_t>(n) * K + k] = acc[ti][tj][r];
      }
    }
  if (blockIdx.z == 0 && tid < kBN && n0 + tid < N && partial_db != nullptr)
    partial_db[static_cast<int64_t>(split) * N + n0 + tid] = colsum;
}

// out[i] = sum_s partial[s][i]: 16 float4 columns x 16 split-slices per block, fixed-order LDS
// tree over the slices (deterministic).
__global__ void __launch_bounds__(kBlock)
k_reduce_splits(const float* __restrict__ partial, int splits, int64_t n, float* __restrict__ out) {
  __shared__ float4 red[kBlock];
  const int col = threadIdx.x & 15, slice = threadIdx.x >> 4;
  const int64_t i = (static_cast<int64_t>(blockIdx.x) * 16 + col) * 4;
  float4 a = zero4();
  if (i < n) {
    for (int s = slice; s < splits; s += 16) {
      const float4 v = ld4(partial + static_cast<int64_t>(s) * n + i);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = 8; o > 0; o >>= 1) {
    if (slice < o) {
      const float4 b = red[threadIdx.x + o * 16];
      float4 c = red[threadIdx.x];
      c.x += b.x; c.y += b.y; c.z += b.z; c.w += b.w;
      red[threadIdx.x] = c;
    }
    __syncthreads();
  }
  if (slice == 0 && i < n) st4(out + i, red[threadIdx.x]);
}

// out[c][r] = in[r][c] for small weight matrices (backward-X uses the forward kernel on W^T).
__global__ void __launch_bounds__(kBlock)
k_transpose(const float* __restrict__ in, int R, int C, float* __restrict__ out) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int r = by + j, c = bx + tx;
    tile[j][tx] = (r < R && c < C) ? in[static_cast<int64_t>(r) * C + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = bx + j, r = by + tx;
    if (r < R && c < C) out[static_cast<int64_t>(c) * R + r] = tile[tx][j];
  }
}

inline bool lin_dims_ok(int64_t M, int64_t N, int64_t K) {
  return M >= 0 && N > 0 && K > 0 && K % 4 == 0 && N <= 65536 && K <= 65536;
}

// rows per tile of the weight-resident forward variant (also the granularity of its statistics partials)
inline int res_block_rows(int64_t Ktot) { return Ktot <= 128 ? 128 : 64; }

// Row splits of the weight-gradient product.  Measured on MI355X (M = 102400): the kernel wants
// ~512 blocks in total (2 resident per CU) but the partial slabs (S * N * K floats, written and
// re-read) want S small: S = 512 / (#output tiles), capped at 256 (one tile: 58.8 us at S = 247
// vs 65 us at S = 458; four tiles: 134-149 us at S = 128 vs 176-183 us at S = 458).
inline int pick_splits(int64_t M, int64_t tiles) {
  int64_t target = 512 / (tiles < 1 ? 1 : tiles);
  if (target > 256) target = 256;
  if (target < 1) target = 1;
  int64_t rows = (M + target - 1) / target;
  rows = (rows + kKC - 1) / kKC * kKC;
  if (rows < kKC) rows = kKC;
  int64_t s = (M + rows - 1) / rows;
  if (s < 1) s = 1;
  if (s > kMaxSplits) s = kMaxSplits;
  return static_cast<int>(s);
}

inline int64_t out_tiles(int64_t N, int64_t K) { return ((N + kBN - 1) / kBN) * ((K + kBN - 1) / kBN); }

}  // namespace
}  // namespace stemgnn

using namespace stemgnn;

extern "C" {

size_t stemgnn_linear_stats_partial_bytes(int64_t M, int64_t N) {
  if (M < 0 || N <= 0) return 0;
  return static_cast<size_t>((M + 63) / 64) * 2 * N * sizeof(float) + 256;  // upper bound over both variants
}

int64_t stemgnn_linear_stats_blocks(int64_t M, int64_t k_total) {
  if (M < 0 || k_total <= 0) return 0;
  const int64_t rows = (kUseResidentVariant && k_total <= 256) ? res_block_rows(k_total) : kBM;
  return (M + rows - 1) / rows;
}

int stemgnn_linear_fwd(const float* x1, const float* w1, int64_t K1, const float* x2, const float* w2, int64_t K2,
                       const float* bias, int64_t M, int64_t N, float* y, float* stats_partial,
                       int64_t* stats_blocks_host, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (!lin_dims_ok(M, N, K1) || K2 < 0 || K2 % 4 != 0) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(M)) return STEMGNN_ERR_TOO_LARGE;
  if (stats_blocks_host) *stats_blocks_host = stemgnn_linear_stats_blocks(M, K1 + K2);
  if (M == 0) return STEMGNN_OK;
  if (!x1 || !w1 || !y || (K2 > 0 && (!x2 || !w2))) return STEMGNN_ERR_INVALID_ARG;
  if (N % 4 != 0) return STEMGNN_ERR_INVALID_ARG;
  const int gy = static_cast<int>((N + kBN - 1) / kBN);
  const int64_t Ktot = K1 + K2;
  if (kUseResidentVariant && Ktot <= 256 && (K2 == 0 || K1 % 4 == 0)) {
    // weight-resident persistent variant: one 512-thread block per CU
    const int BM = res_block_rows(Ktot);
    const int64_t tiles = (M + BM - 1) / BM;
    int gx = 256 / gy;
    if (gx < 1) gx = 1;
    if (gx > tiles) gx = static_cast<int>(tiles);
    dim3 grid(static_cast<unsigned>(gx), static_cast<unsigned>(gy));
    const int k1 = static_cast<int>(K1), k2 = static_cast<int>(K2), n = static_cast<int>(N);
    if (BM == 128)
      return stats_partial ? launch_res<128, true>(grid, st, x1, w1, k1, x2, w2, k2, bias, M, n, y, stats_partial)
                           : launch_res<128, false>(grid, st, x1, w1, k1, x2, w2, k2, bias, M, n, y, nullptr);
    return stats_partial ? launch_res<64, true>(grid, st, x1, w1, k1, x2, w2, k2, bias, M, n, y, stats_partial)
                         : launch_res<64, false>(grid, st, x1, w1, k1, x2, w2, k2, bias, M, n, y, nullptr);
  }
  dim3 grid(static_cast<unsigned>((M + kBM - 1) / kBM), static_cast<unsigned>(gy));
  if (stats_partial)
    k_linear_fwd<true><<<grid, kBlock, 0, st>>>(x1, w1, static_cast<int>(K1), x2, w2, static_cast<int>(K2), bias, M,
                                                static_cast<int>(N), y, stats_partial);
  else
    k_linear_fwd<false><<<grid, kBlock, 0, st>>>(x1, w1, static_cast<int>(K1), x2, w2, static_cast<int>(K2), bias, M,
                                                 static_cast<int>(N), y, nullptr);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

size_t stemgnn_linear_bwd_weight_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (!lin_dims_ok(M, N, K)) return 0;
  return static_cast<size_t>(pick_splits(M, out_tiles(N, K))) * (N * K + N) * sizeof(float) + 512;
}

int stemgnn_linear_bwd_weight(const float* dy, const float* x, int64_t M, int64_t N, int64_t K, float* dw, float* db,
                              void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (!lin_dims_ok(M, N, K) || N % 4 != 0 || !dw) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(M)) return STEMGNN_ERR_TOO_LARGE;
  if (M == 0) {
    STEMGNN_HIP_TRY(hipMemsetAsync(dw, 0, sizeof(float) * N * K, st));
    if (db) STEMGNN_HIP_TRY(hipMemsetAsync(db, 0, sizeof(float) * N, st));
    return STEMGNN_OK;
  }
  if (!dy || !x || !workspace) return STEMGNN_ERR_INVALID_ARG;
  if (workspace_bytes < stemgnn_linear_bwd_weight_workspace_bytes(M, N, K)) return STEMGNN_ERR_WORKSPACE;
  const int S = pick_splits(M, out_tiles(N, K));
  int64_t rows = (M + S - 1) / S;
  rows = (rows + kKC - 1) / kKC * kKC;
  float* pw = reinterpret_cast<float*>(align_up(reinterpret_cast<uintptr_t>(workspace), 256));
  float* pb = pw + static_cast<size_t>(S) * N * K;
  dim3 grid(static_cast<unsigned>(S), static_cast<unsigned>((N + kBN - 1) / kBN), static_cast<unsigned>((K + kBN - 1) / kBN));
  k_linear_bwd_weight<<<grid, kBlock, 0, st>>>(dy, x, M, static_cast<int>(N), static_cast<int>(K), rows, pw,
                                               db ? pb : nullptr);
  STEMGNN_LAUNCH_CHECK();
  const int64_t nk = N * K;
  k_reduce_splits<<<static_cast<unsigned>((nk / 4 + 15) / 16), kBlock, 0, st>>>(pw, S, nk, dw);
  STEMGNN_LAUNCH_CHECK();
  if (db) {
    k_reduce_splits<<<static_cast<unsigned>((N / 4 + 15) / 16), kBlock, 0, st>>>(pb, S, N, db);
    STEMGNN_LAUNCH_CHECK();
  }
  return STEMGNN_OK;
}

int stemgnn_transpose(const float* in, int64_t rows, int64_t cols, float* out, void* stream_) {
  if (rows <= 0 || cols <= 0 || !in || !out) return STEMGNN_ERR_INVALID_ARG;
  dim3 grid(static_cast<unsigned>((cols + 31) / 32), static_cast<unsigned>((rows + 31) / 32));
  k_transpose<<<grid, kBlock, 0, static_cast<hipStream_t>(stream_)>>>(in, static_cast<int>(rows),
                                                                     static_cast<int>(cols), out);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

}  // extern "C"
