// K3 / K5: dense per-layer projections on the matrix cores of gfx950, fp32 results.
//
// Reference: MySAGEConv's lin_l(agg) + lin_r(x) (STEM-GNN/model/encoder.py:83-87),
// VectorQuantize.project_in / project_out (model/vq.py:881,1041) and the decoders' nn.Linear
// layers (model/pt_model.py:42,80,94; model/encoder.py:364).  The reference leaves these to
// ATen/cuBLAS plus separate bias-add, add, transpose and bias-gradient reduction kernels.  The
// activations here are tall and skinny (M ~ 1e5 rows, N, K in {128, 256, 512}), a shape the vendor GEMM
// library serves badly on this chip (12-30 TFLOP/s for the weight-gradient product), so the three
// products are hand-written:
//
//   forward     Y[M,N]  = X1[M,K1] W1[N,K1]^T (+ X2[M,K2] W2[N,K2]^T) + b      (fused K-concat)
//                         optional per-column sum / sum-of-squares partials of Y (BatchNorm stats)
//   backward-W  dW[N,K] = dY[M,N]^T X[M,K],  db[N] = colsum(dY)   (split over M, two-stage,
//                         deterministic: no atomics)
//   backward-X  dX[M,K] = dY[M,N] W[N,K]  = the forward tile reading W as stored (BT)
//
// Two instruction paths, both fp32-accurate (the reference runs fp32 without autocast):
//   *_x3 (default)  v_mfma_f32_32x32x16_bf16 on exact three-way bf16 pieces of the fp32 operands
//                   (common.h: split3 / mfma_x3): six MFMAs of 32 cycles per 16 k;
//   plain           v_mfma_f32_32x32x2_f32: eight MFMAs of 64 cycles per 16 k; kept as the cross-check
//                   (stemgnn_linear_set_mode(0) / STEMGNN_GEMM=f32).
//
// Tiling: 256-thread block = 4 waves in a 2x2 grid, 128x128 output tile, each wave 64x64 =
// 2x2 MFMA tiles (64 accumulator registers); K (or M for backward-W) is consumed in chunks of
// 32 staged through LDS with register prefetch of the next chunk.
#include "common.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>

namespace stemgnn {
namespace {

constexpr int kBlock = 256;
constexpr int kBM = 128, kBN = 128, kKC = 32;
constexpr int kLd = kKC + 4;  // 36-dword row stride: conflict-free ds_read_b128 of 32 rows x 16 B
constexpr int kMaxSplits = 512;

// C-tile register r of lane (lj, hi) is row (r&3) + 8*(r>>2) + 4*hi, column lj.
__device__ inline int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// ---------------------------------------------------------------------------------------
// forward: Y = X1 W1^T (+ X2 W2^T) + b
// ---------------------------------------------------------------------------------------
template <int BM, bool STATS, bool BT = false>
__global__ void __launch_bounds__(kBlock, 2)
k_linear_fwd(const float* __restrict__ x1, const float* __restrict__ w1, int K1, const float* __restrict__ x2,
             const float* __restrict__ w2, int K2, const float* __restrict__ bias, int64_t M, int N,
             float* __restrict__ y, float* __restrict__ stats_partial /*[blocks][2][N]*/, int64_t row_base,
             int64_t stats_block0, int64_t x1_rows, int64_t store_rows) {
  // BM = 128: waves 2(m) x 2(n), 64x64 per wave.  BM = 32 (tail tiles): waves 1 x 4, 32x32 per wave.
  constexpr int WM = BM == 128 ? 2 : 1;
  constexpr int WN = 4 / WM;
  constexpr int TM = BM / (32 * WM);
  constexpr int TN = kBN / (32 * WN);
  constexpr int FA = BM * 8 / kBlock;  // float4 of an activation chunk per thread
  __shared__ __attribute__((aligned(16))) float sA[BM * kLd];
  __shared__ __attribute__((aligned(16))) float sB[kBN * kLd];
  __shared__ float s_stats[WM][2][kBN];  // [wave_m][sum|sumsq][n]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, hi = lane >> 5, lj = lane & 31;
  const int64_t m0 = row_base + static_cast<int64_t>(blockIdx.x) * BM;
  const int n0 = blockIdx.y * kBN;
  const int c1 = (K1 + kKC - 1) / kKC, c2 = (K2 + kKC - 1) / kKC;
  const int steps = c1 + c2;

  float4 ra[FA], rb[4];
  auto fetch = [&](int step) {
    const bool second = step >= c1;
    const float* xs = second ? x2 : x1;
    const float* ws = second ? w2 : w1;
    const int K = second ? K2 : K1;
    const int k0 = (second ? step - c1 : step) * kKC;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = t * kBlock + tid;  // 16-byte column = idx % 8, row = idx / 8
      const int r = idx >> 3, k = k0 + 4 * (idx & 7);
      const int n = n0 + r;
      if (BT) {
        // weight given as [K][N] (backward-data: dX = dY W): rows k0 + 4 (tid & 7) + t, columns n0 + 4 (tid >> 3) ..+3
        const int kk = k0 + 4 * (tid & 7) + t, nn = n0 + 4 * (tid >> 3);
        rb[t] = (kk < K && nn < N) ? ld4(ws + static_cast<int64_t>(kk) * N + nn) : zero4();
      } else {
        rb[t] = (n < N && k < K) ? ld4(ws + static_cast<int64_t>(n) * K + k) : zero4();
      }
      if (t < FA) {
        const int64_t m = m0 + r;
        // rows >= x1_rows of the first operand are zero by promise and are never read (the buffer may end there)
        ra[t] = (m < (second ? M : x1_rows) && k < K) ? ld4(xs + m * K + k) : zero4();
      }
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = t * kBlock + tid;
      if (BT) {  // rb[t] = W[k0 + 4 kq + t][n0 + 4 cq .. +3]  ->  sB[n][k]
        const int kq = tid & 7, cq = tid >> 3;
        sB[(4 * cq + 0) * kLd + 4 * kq + t] = rb[t].x;
        sB[(4 * cq + 1) * kLd + 4 * kq + t] = rb[t].y;
        sB[(4 * cq + 2) * kLd + 4 * kq + t] = rb[t].z;
        sB[(4 * cq + 3) * kLd + 4 * kq + t] = rb[t].w;
      } else {
        st4(sB + (idx >> 3) * kLd + 4 * (idx & 7), rb[t]);
      }
      if (t < FA) st4(sA + (idx >> 3) * kLd + 4 * (idx & 7), ra[t]);
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float bias_v[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + wn * 32 * TN + tn * 32 + lj;
    bias_v[tn] = (bias != nullptr && n < N) ? bias[n] : 0.f;
  }

  // rows >= x1_rows of the first operand are zero (an aggregate over a sampled batch: only the leading, expanded
  // nodes receive edges): a tile past them starts at the second operand's chunks
  const int first_step = (m0 >= x1_rows) ? c1 : 0;
  if (first_step < steps) fetch(first_step);
  for (int step = first_step; step < steps; ++step) {
    stash();
    __syncthreads();
    if (step + 1 < steps) fetch(step + 1);
#pragma unroll
    for (int ms = 0; ms < kKC / 8; ++ms) {
      const int ko = ms * 8 + hi * 4;  // lane half 0 takes k 0..3, half 1 takes k 4..7 of the micro-step
      float4 a[TM], b[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) a[t] = ld4(sA + (wm * 32 * TM + t * 32 + lj) * kLd + ko);
#pragma unroll
      for (int t = 0; t < TN; ++t) b[t] = ld4(sB + (wn * 32 * TN + t * 32 + lj) * kLd + ko);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
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
  const bool interior = (m0 + BM <= M) && (n0 + kBN <= N);
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int nl = wn * 32 * TN + tn * 32 + lj;
    const int n = n0 + nl;
    const float bv = bias_v[tn];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int64_t mrow0 = m0 + wm * 32 * TM + tm * 32 + 4 * hi;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = mrow0 + (r & 3) + 8 * (r >> 2);
        const float v = acc[tm][tn][r] + bv;
        if (interior || (m < M && n < N)) {
          if (m < store_rows) y[m * N + n] = v;  // rows past store_rows only feed the column statistics
          if (STATS) { s1 += v; s2 += v * v; }
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
      // one slab per 64-row half of a 128-row tile (what the weight-stationary kernel's tiles are), one per 32-row tile
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        float* p = stats_partial + (stats_block0 + static_cast<int64_t>(blockIdx.x) * WM + w) * 2 * N;
        p[n0 + tid] = s_stats[w][0][tid];
        p[N + n0 + tid] = s_stats[w][1][tid];
      }
    }
  }
}

// The same tile on the bf16 matrix cores (split3 / mfma_x3 above): chunks are split while they are staged.
// Staging row of float4 slot idx (8 slots per row).  ds_write_b64 is served in groups of 16 consecutive lanes = two
// rows; with the 80-byte plane stride rows r and r+1 share 12 of their 16 banks (2-way conflict: a third of the LDS
// cycles of this kernel, SQ_LDS_BANK_CONFLICT), rows r and r+4 share none -- so the rows of every group of eight
// are taken in the order 0 4 1 5 2 6 3 7.  Global loads are unaffected: 8 lanes still read one 128-byte row segment.
__device__ __forceinline__ int stage_row(int idx) {
  const int r = idx >> 3;
  return (r & ~7) | ((r & 1) << 2) | ((r >> 1) & 3);
}
// PIECES = 1: the bf16 GEMM mode -- operands rounded to bf16 while staged, one matrix instruction per tile and k step.
template <int BM, bool STATS, bool BT = false, int X2K = kF32, int PIECES = 3>
__global__ void __launch_bounds__(kBlock, 2)
k_linear_fwd_x3(const float* __restrict__ x1, const float* __restrict__ w1, int K1, const float* __restrict__ x2,
             const float* __restrict__ w2, int K2, const float* __restrict__ bias, int64_t M, int N,
             float* __restrict__ y, float* __restrict__ stats_partial /*[blocks][2][N]*/, int64_t row_base,
             int64_t stats_block0, int64_t x1_rows, int64_t store_rows) {
  // BM = 128: waves 2(m) x 2(n), 64x64 per wave.  BM = 32 (tail tiles): waves 1 x 4, 32x32 per wave.
  constexpr int WM = BM == 128 ? 2 : 1;
  constexpr int WN = 4 / WM;
  constexpr int TM = BM / (32 * WM);
  constexpr int TN = kBN / (32 * WN);
  constexpr int FA = BM * 8 / kBlock;  // float4 of an activation chunk per thread
  constexpr int PA = BM * kLdP, PB = kBN * kLdP;  // bytes of one bf16 plane
  // planes h, m, l of the activation chunk, then of the weight chunk; the epilogue reuses the memory as an fp32 tile
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * PA + 3 * PB];
  unsigned char* const sA = smem;
  unsigned char* const sB = smem + 3 * PA;
  __shared__ float s_stats[WM][2][kBN];  // [wave_m][sum|sumsq][n]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, hi = lane >> 5, lj = lane & 31;
  const int64_t m0 = row_base + static_cast<int64_t>(blockIdx.x) * BM;
  const int n0 = blockIdx.y * kBN;
  const int c1 = (K1 + kKC - 1) / kKC, c2 = (K2 + kKC - 1) / kKC;
  const int steps = c1 + c2;

  float4 ra[FA], rb[4];
  bool held_bf16 = false;  // ra holds raw bf16 bits of the second operand (X2K == kBF16 only)
  auto fetch = [&](int step) {
    const bool second = step >= c1;
    const float* xs = second ? x2 : x1;
    const float* ws = second ? w2 : w1;
    const int K = second ? K2 : K1;
    const int k0 = (second ? step - c1 : step) * kKC;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = t * kBlock + tid;  // 16-byte column = idx % 8, row = idx / 8
      const int r = stage_row(idx), k = k0 + 4 * (idx & 7);
      const int n = n0 + r;
      if (BT) {
        // weight given as [K][N] (backward-data: dX = dY W): rows k0 + 4 (tid & 7) + t, columns n0 + 4 (tid >> 3) ..+3
        const int kk = k0 + 4 * (tid & 7) + t, nn = n0 + 4 * (tid >> 3);
        rb[t] = (kk < K && nn < N) ? ld4(ws + static_cast<int64_t>(kk) * N + nn) : zero4();
      } else {
        rb[t] = (n < N && k < K) ? ld4(ws + static_cast<int64_t>(n) * K + k) : zero4();
      }
      if (t < FA) {
        const int64_t m = m0 + r;
        // rows >= x1_rows of the first operand are zero by promise and are never read (the buffer may end there)
        // the second operand (the layer input h) may be stored as bf16: widened here, the cut that follows is exact
        // the bits stay as loaded (two dwords of ra[t]) until stash(): a widening here would sit between the loads
        if (X2K == kBF16 && second) {
          const uint2 raw = (m < M && k < K) ? *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(xs) + m * K + k)
                                             : make_uint2(0u, 0u);
          ra[t] = make_float4(__uint_as_float(raw.x), __uint_as_float(raw.y), 0.f, 0.f);
        } else {
          ra[t] = (m < (second ? M : x1_rows) && k < K) ? ld4(xs + m * K + k) : zero4();
        }
      }
    }
    held_bf16 = X2K == kBF16 && second;
  };
  auto stash = [&]() {
    if (BT) stash_transposed<PIECES>(rb, sB, PB, tid);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = t * kBlock + tid;
      const int off = stage_row(idx) * kLdP + 8 * (idx & 7);
      uint2 h, m, l;
      if (!BT) {
        if (PIECES == 1) {
          *reinterpret_cast<uint2*>(sB + off) = pack_rne(rb[t]);
        } else {
          split3(rb[t], h, m, l);
          *reinterpret_cast<uint2*>(sB + off) = h;
          *reinterpret_cast<uint2*>(sB + PB + off) = m;
          *reinterpret_cast<uint2*>(sB + 2 * PB + off) = l;
        }
      }
      if (t < FA) {
        if (X2K == kBF16 && held_bf16) {  // four bf16 values ARE the h plane of their fp32 widening; m = l = 0 and
          // never read (the matrix loop takes the one-piece form for these chunks)
          *reinterpret_cast<uint2*>(sA + off) = make_uint2(__float_as_uint(ra[t].x), __float_as_uint(ra[t].y));
          continue;
        }
        if (PIECES == 1) {
          *reinterpret_cast<uint2*>(sA + off) = pack_rne(ra[t]);
          continue;
        }
        split3(ra[t], h, m, l);
        *reinterpret_cast<uint2*>(sA + off) = h;
        *reinterpret_cast<uint2*>(sA + PA + off) = m;
        *reinterpret_cast<uint2*>(sA + 2 * PA + off) = l;
      }
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float bias_v[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + wn * 32 * TN + tn * 32 + lj;
    bias_v[tn] = (bias != nullptr && n < N) ? bias[n] : 0.f;
  }

  // rows >= x1_rows of the first operand are zero (an aggregate over a sampled batch: only the leading, expanded
  // nodes receive edges): a tile past them starts at the second operand's chunks
  const int first_step = (m0 >= x1_rows) ? c1 : 0;
  if (first_step < steps) fetch(first_step);
  for (int step = first_step; step < steps; ++step) {
    stash();
    __syncthreads();
    if (step + 1 < steps) fetch(step + 1);
#pragma unroll
    for (int ks = 0; ks < kKC / 16; ++ks) {
      const int ko = ks * 32 + hi * 16;  // bytes: lane half 0 takes k 0..7, half 1 k 8..15 of the 16-wide step
      bf16x8 a[TM][3], b[TN][3];
      // chunks of a bf16-stored second operand carry one piece (stash() left the m / l planes alone): three products
      const bool one_piece = PIECES == 1 || (X2K == kBF16 && step >= c1);  // block-uniform
#pragma unroll
      for (int p = 0; p < PIECES; ++p) {
        if (!(one_piece && p > 0)) {
#pragma unroll
          for (int t = 0; t < TM; ++t)
            a[t][p] = *reinterpret_cast<const bf16x8*>(sA + p * PA + (wm * 32 * TM + t * 32 + lj) * kLdP + ko);
        }
#pragma unroll
        for (int t = 0; t < TN; ++t)
          b[t][p] = *reinterpret_cast<const bf16x8*>(sB + p * PB + (wn * 32 * TN + t * 32 + lj) * kLdP + ko);
      }
      if (PIECES == 1) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][0], acc[tm][tn], 0, 0, 0);
      } else if (one_piece) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma_x3_a1(a[tm][0], b[tn], acc[tm][tn]);
      } else {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma_x3(a[tm], b[tn], acc[tm][tn]);
      }
    }
    __syncthreads();
  }

  // ---- epilogue (128-row tiles): the accumulators hold one column and 16 scattered rows per lane, which as direct
  // stores is 64 four-byte store instructions per lane; measured on the plane-operand twin of this tile
  // (tools/pgemm_ablate.py) those stores cost more than the tile's matrix work.  The tile goes through LDS instead,
  // 64 rows at a time, and leaves as 16-byte stores of whole 512-byte rows (8 per thread).
  if (BM == 128) {
    constexpr int kLdT = kBN + 4;  // fp32 row stride of the staging tile (64 x 132 x 4 B = 33 KB of the 60 KB)
    float* tile = reinterpret_cast<float*>(smem);
    float s1[TN], s2[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) s1[tn] = s2[tn] = 0.f;
    const int64_t row_end = store_rows < M ? store_rows : M;  // rows past it only feed the column statistics
    for (int half = 0; half < 2; ++half) {
      const bool stored = m0 + half * 64 < row_end;  // block-uniform
      if (wm == half) {
        // rows of this lane's register sequence that lie below M (a whole half: >= 64); one 32-bit compare per
        // element instead of a 64-bit row test (the statistics variant was 6 us slower than the plain one)
        const int64_t live64 = M - m0 - half * 64 - 4 * hi;
        const int live = live64 > 64 ? 64 : (live64 < 0 ? 0 : static_cast<int>(live64));
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int nl = wn * 32 * TN + tn * 32 + lj;
          const bool col_ok = n0 + nl < N;
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int rs = tm * 32 + (r & 3) + 8 * (r >> 2);  // row of the sequence, without the lane half's 4 hi
              const float v = acc[tm][tn][r] + bias_v[tn];
              if (stored) tile[(rs + 4 * hi) * kLdT + nl] = v;
              if (STATS && col_ok && rs < live) { s1[tn] += v; s2[tn] += v * v; }
            }
        }
      }
      if (!stored) continue;
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rl = (tid >> 5) + 8 * i, c4 = tid & 31;
        const int64_t m = m0 + half * 64 + rl;
        if (m < row_end && n0 + 4 * c4 < N) st4(y + m * N + n0 + 4 * c4, ld4(tile + rl * kLdT + 4 * c4));
      }
      __syncthreads();
    }
    if (STATS) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int nl = wn * 32 * TN + tn * 32 + lj;
        float a = s1[tn], b = s2[tn];
        a += __shfl_xor(a, 32, 64);
        b += __shfl_xor(b, 32, 64);
        if (hi == 0) { s_stats[wm][0][nl] = a; s_stats[wm][1][nl] = b; }
      }
      __syncthreads();
      if (tid < kBN && n0 + tid < N) {
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          float* p = stats_partial + (stats_block0 + static_cast<int64_t>(blockIdx.x) * WM + w) * 2 * N;
          p[n0 + tid] = s_stats[w][0][tid];
          p[N + n0 + tid] = s_stats[w][1][tid];
        }
      }
    }
    return;
  }
  // ---- epilogue: bias, store (two 128-byte row segments per store instruction), column stats.
  // The bias was loaded before the main loop: the epilogue issues stores only, so no
  // s_waitcnt vmcnt(0) ever serialises them (vmcnt counts stores too).
  const bool interior = (m0 + BM <= M) && (n0 + kBN <= N);
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int nl = wn * 32 * TN + tn * 32 + lj;
    const int n = n0 + nl;
    const float bv = bias_v[tn];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int64_t mrow0 = m0 + wm * 32 * TM + tm * 32 + 4 * hi;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = mrow0 + (r & 3) + 8 * (r >> 2);
        const float v = acc[tm][tn][r] + bv;
        if (interior || (m < M && n < N)) {
          if (m < store_rows) y[m * N + n] = v;  // rows past store_rows only feed the column statistics
          if (STATS) { s1 += v; s2 += v * v; }
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
      // one slab per 64-row half of a 128-row tile (what the weight-stationary kernel's tiles are), one per 32-row tile
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        float* p = stats_partial + (stats_block0 + static_cast<int64_t>(blockIdx.x) * WM + w) * 2 * N;
        p[n0 + tid] = s_stats[w][0][tid];
        p[N + n0 + tid] = s_stats[w][1][tid];
      }
    }
  }
}

// (A weight-resident persistent variant -- 128-column weight tile kept in LDS, one 512-thread block
// per CU streaming activation chunks through a ring with one barrier per step -- was built and
// measured slower than the kernel above: 55 vs 46 us at K = 128, 92 vs 81 us at K = 256, M = 102400,
// with or without a three-chunk-deep prefetch.  Removed; see git history.)

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

// The same product on the bf16 matrix cores.  Both operands are contracted over their SLOW dimension (rows m),
// and a v_mfma_f32_32x32x16_bf16 lane wants 8 consecutive m: each thread stages a 4 (m) x 4 (n or k) micro-tile,
// splits it and stores it transposed, so the LDS planes are [n][32 m] / [k][32 m] and the fragment reads are the
// forward kernel's.  db comes from the staged registers (shuffle over the 8 lanes that share a column group).
// The same segment sums with the work cut the other way (the form the step runs): a block owns 32 FEATURE columns and
// four 128-code tiles (one per wave), so a chunk of g is read, cut and transposed ONCE per 32 columns instead of once
// per code tile (four times at H = 4), and the one-hot operand never touches LDS -- a lane compares the eight row
// assignments of its k-step (staged as int32 [head][rows]) with its own code and packs the result.  Same order of
// additions within a split (the splits are cut at multiples of 128 rows here).
__global__ void __launch_bounds__(kBlock, 2)
k_code_segment_sums_cols(const int64_t* __restrict__ ind, int H, int K, const float* __restrict__ g, int64_t M, int D,
                         int64_t rows_per_split, float* __restrict__ partial /*[S][H*K][D]*/) {
  // a step is 128 rows: every wave cuts and transposes ITS 32-row chunk of g (four plane sets), then every wave
  // multiplies all four chunks with its own code tile -- one pair of barriers per 128 rows
  constexpr int PL = 32 * kLdP, kStep = 128;
  __shared__ __attribute__((aligned(16))) unsigned char sB[4 * 3 * PL];  // [chunk][plane] g^T [32 d][32 m]
  __shared__ __attribute__((aligned(16))) int s_ind[16 * kStep];         // [head][128 rows], H <= 16
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, lj = lane & 31;
  const int split = blockIdx.x, d0 = blockIdx.y * 32;
  const int NC = H * K;
  const int cb = (blockIdx.z * 4 + wave) * 128;  // this wave's code tile
  const int64_t mbeg = split * rows_per_split;
  const int64_t mend = min(M, mbeg + rows_per_split);
  const int steps = mend > mbeg ? static_cast<int>((mend - mbeg + kStep - 1) / kStep) : 0;
  int ch[4], ck[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int c = cb + t * 32 + lj;
    ch[t] = c < NC ? c / K : 0;
    ck[t] = c < NC ? c % K : -7;  // never matches
  }
  const int mq = lane & 7, cq = lane >> 3;  // within the wave's chunk: rows 4 mq .. + 3, columns d0 + 4 cq .. + 3
  float4 rb[4];
  int ri[8];  // 128 H assignments per step over 256 threads: H / 2 each
  auto fetch = [&](int step) {
    const int64_t mm = mbeg + static_cast<int64_t>(step) * kStep;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = mm + 32 * wave + 4 * mq + i;
      rb[i] = m < mend ? ld4(g + m * D + d0 + 4 * cq) : zero4();
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int idx = q * kBlock + tid;
      ri[q] = -1;
      if (idx < kStep * H) {
        const int64_t m = mm + idx / H;
        if (m < mend) ri[q] = static_cast<int>(ind[m * H + idx % H]);
      }
    }
  };
  floatx16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  if (steps > 0) fetch(0);
  for (int step = 0; step < steps; ++step) {
    stash_transposed(rb, sB + wave * 3 * PL, PL, lane);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int idx = q * kBlock + tid;
      if (idx < kStep * H) s_ind[(idx % H) * kStep + idx / H] = ri[q];
    }
    __syncthreads();
    if (step + 1 < steps) fetch(step + 1);
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
#pragma unroll
      for (int ks = 0; ks < kKC / 16; ++ks) {
        const int ko = ks * 32 + hi * 16;
        bf16x8 b[3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
          b[p] = *reinterpret_cast<const bf16x8*>(sB + (c4 * 3 + p) * PL + lj * kLdP + ko);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int* iv = s_ind + ch[t] * kStep + c4 * 32 + ks * 16 + hi * 8;
          const int4 i0 = *reinterpret_cast<const int4*>(iv);
          const int4 i1 = *reinterpret_cast<const int4*>(iv + 4);
          const uint32_t one = 0x3F80u;  // bf16 1.0
          const int kk = ck[t];
          const uint4 av = make_uint4((i0.x == kk ? one : 0u) | (i0.y == kk ? one << 16 : 0u),
                                      (i0.z == kk ? one : 0u) | (i0.w == kk ? one << 16 : 0u),
                                      (i1.x == kk ? one : 0u) | (i1.y == kk ? one << 16 : 0u),
                                      (i1.z == kk ? one : 0u) | (i1.w == kk ? one << 16 : 0u));
          const bf16x8 a = __builtin_bit_cast(bf16x8, av);
          floatx16 c = acc[t];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[2], c, 0, 0, 0);  // small pieces first
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[0], c, 0, 0, 0);
          acc[t] = c;
        }
      }
    }
    __syncthreads();
  }
  float* pw = partial + static_cast<int64_t>(split) * NC * D;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = cb + t * 32 + acc_row(r, hi);
      if (n < NC) pw[static_cast<int64_t>(n) * D + d0 + lj] = acc[t][r];
    }
}

// (Round 3 measured a form of these sums without matrix instructions -- per chunk the (row, head) entries counting-sorted
// by code bin in LDS, each bin owned by one eight-lane group, sums in registers, bit-reproducible: correct, but 53 us in
// the step against this kernel's 49: staging 13 + sort 16 + walk 12 us of LDS round trips per 128-row chunk.  Dropped;
// the kernel is kept as a record in tools/micro/segsum_bins.hip.)

// One launch serves several products (a phase's weight gradients): a block finds its job in a table passed by value.
constexpr int kDwJobs = 8;
struct DwJob {
  const float* dy;
  const void* x;
  float* pw;   // [S][N][K] slabs
  float* pb;   // [S][N] or null
  int64_t M, rows_per_split;
  int N, K, splits, ktiles, block_end;  // blocks [block_end of the job before, block_end)
};
struct DwTable {
  DwJob job[kDwJobs];
  int count;
};

template <int XK, int PIECES = 3>  // element kind of x (common.h: kF32 / kBF16), compile-time: a run-time kind cost the
// fp32 launch 18 %; PIECES = 1: the bf16 GEMM mode (dY and x rounded to bf16 while staged; db stays the fp32 column sum)
__global__ void __launch_bounds__(kBlock, 2)
k_linear_bwd_weight_x3(const DwTable tab) {
  constexpr int PL = kBN * kLdP;
  __shared__ __attribute__((aligned(16))) unsigned char sA[3 * PL];  // dY^T planes [128 n][32 m]
  __shared__ __attribute__((aligned(16))) unsigned char sB[3 * PL];  // X^T  planes [128 k][32 m]

  int j = 0, begin = 0;
  while (j + 1 < tab.count && static_cast<int>(blockIdx.x) >= tab.job[j].block_end) { begin = tab.job[j].block_end; ++j; }
  const DwJob& jb = tab.job[j];
  const float* __restrict__ dy = jb.dy;
  const float* __restrict__ x = static_cast<const float*>(jb.x);
  const int64_t M = jb.M, rows_per_split = jb.rows_per_split;
  const int N = jb.N, K = jb.K;
  float* __restrict__ partial_dw = jb.pw;
  float* __restrict__ partial_db = jb.pb;
  const int local = static_cast<int>(blockIdx.x) - begin;
  const int split = local % jb.splits, tile = local / jb.splits;  // splits of one output tile are neighbours

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1, hi = lane >> 5, lj = lane & 31;
  const int n0 = (tile / jb.ktiles) * kBN, k0 = (tile % jb.ktiles) * kBN;
  const int64_t mbeg = split * rows_per_split;
  const int64_t mend = min(M, mbeg + rows_per_split);
  const int steps = static_cast<int>((mend - mbeg + kKC - 1) / kKC);
  const int mq = tid & 7, cq = tid >> 3;  // rows 4 mq .. 4 mq + 3 of the chunk, columns 4 cq .. 4 cq + 3 of the tile
  const bool do_db = k0 == 0 && partial_db != nullptr;

  float4 ra[4];
  Raw4<XK> rb[4];  // widened at the stash: no conversion between the prefetch loads
  auto fetch = [&](int step) {
    const int64_t mm = mbeg + static_cast<int64_t>(step) * kKC + 4 * mq;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = mm + i;
      ra[i] = (m < mend && n0 + 4 * cq < N) ? ld4(dy + m * N + n0 + 4 * cq) : zero4();
      if (m < mend && k0 + 4 * cq < K) rb[i].load(x, m * K + k0 + 4 * cq);
      else rb[i].clear();
    }
  };
  auto stash_t = [&](const float4 (&q)[4], unsigned char* planes) { stash_transposed<PIECES>(q, planes, PL, tid); };

  floatx16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float4 colsum = zero4();  // columns 4 cq .. 4 cq + 3 (lanes with mq == 0 hold the total)

  if (steps > 0) fetch(0);
  for (int step = 0; step < steps; ++step) {
    stash_t(ra, sA);
    {
      const float4 wb[4] = {rb[0].widen(), rb[1].widen(), rb[2].widen(), rb[3].widen()};
      stash_t(wb, sB);
    }
    if (do_db) {
      float4 c = make_float4(ra[0].x + ra[1].x + ra[2].x + ra[3].x, ra[0].y + ra[1].y + ra[2].y + ra[3].y,
                             ra[0].z + ra[1].z + ra[2].z + ra[3].z, ra[0].w + ra[1].w + ra[2].w + ra[3].w);
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) {
        c.x += __shfl_xor(c.x, o, 64); c.y += __shfl_xor(c.y, o, 64);
        c.z += __shfl_xor(c.z, o, 64); c.w += __shfl_xor(c.w, o, 64);
      }
      colsum.x += c.x; colsum.y += c.y; colsum.z += c.z; colsum.w += c.w;
    }
    __syncthreads();
    if (step + 1 < steps) fetch(step + 1);
#pragma unroll
    for (int ks = 0; ks < kKC / 16; ++ks) {
      const int ko = ks * 32 + hi * 16;
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int p = 0; p < PIECES; ++p)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          a[t][p] = *reinterpret_cast<const bf16x8*>(sA + p * PL + (wi * 64 + t * 32 + lj) * kLdP + ko);
          b[t][p] = *reinterpret_cast<const bf16x8*>(sB + p * PL + (wj * 64 + t * 32 + lj) * kLdP + ko);
        }
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
          if (PIECES == 1)
            acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti][0], b[tj][0], acc[ti][tj], 0, 0, 0);
          else
            acc[ti][tj] = XK == kBF16 ? mfma_x3_b1(a[ti], b[tj][0], acc[ti][tj])  // a bf16-stored x is its own h piece
                                      : mfma_x3(a[ti], b[tj], acc[ti][tj]);
        }
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
  if (do_db && mq == 0 && n0 + 4 * cq < N)  // N % 4 == 0: the four columns are in range together
    st4(partial_db + static_cast<int64_t>(split) * N + n0 + 4 * cq, colsum);
}

// Segment sums by code on the matrix cores: S[h * K + k][d] = sum over rows m with ind[m][h] == k of g[m][d], i.e. the
// weight-gradient product above with dY^T replaced by the ONE-HOT matrix of the code assignment.  A one-hot entry is
// exact in a single bf16 piece, so the A operand is built in registers from `ind` (no [M, H*K] or [M, H*Dc] operand is
// ever read) and a 16-row step takes three MFMAs (g's three pieces) instead of six.  Used for project_out's weight
// gradient: quantize = sum_h embed[h, ind[:, h]] W_out_h^T, so dW_out_h = S_h^T embed_h (reference model/vq.py:1041).
// Deterministic: row splits + the fixed-order slab reduction below.
__global__ void __launch_bounds__(kBlock, 2)
k_code_segment_sums(const int64_t* __restrict__ ind, int H, int K, const float* __restrict__ g, int64_t M, int D,
                    int64_t rows_per_split, float* __restrict__ partial /*[S][H*K][D]*/) {
  constexpr int PL = kBN * kLdP;
  __shared__ __attribute__((aligned(16))) unsigned char sA[PL];      // one-hot^T plane [128 codes][32 m]
  __shared__ __attribute__((aligned(16))) unsigned char sB[3 * PL];  // g^T planes   [128 d][32 m]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1, hi = lane >> 5, lj = lane & 31;
  const int split = blockIdx.x;
  const int n0 = blockIdx.y * kBN, k0 = blockIdx.z * kBN;  // code tile, feature tile
  const int NC = H * K;
  const int64_t mbeg = split * rows_per_split;
  const int64_t mend = min(M, mbeg + rows_per_split);
  const int steps = static_cast<int>((mend - mbeg + kKC - 1) / kKC);
  const int mq = tid & 7, cq = tid >> 3;
  // the four code columns this thread stages: (head, code) of each
  int ch[4], ck[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + 4 * cq + j;
    ch[j] = n < NC ? n / K : -1;
    ck[j] = n < NC ? n % K : -1;
  }

  int code[4][4];  // code[i][j]: assignment of row i in the head of column j
  float4 rb[4];
  auto fetch = [&](int step) {
    const int64_t mm = mbeg + static_cast<int64_t>(step) * kKC + 4 * mq;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = mm + i;
      rb[i] = (m < mend && k0 + 4 * cq < D) ? ld4(g + m * D + k0 + 4 * cq) : zero4();
#pragma unroll
      for (int j = 0; j < 4; ++j)
        code[i][j] = (m < mend && ch[j] >= 0) ? static_cast<int>(ind[m * H + ch[j]]) : -2;
    }
  };

  floatx16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  if (steps > 0) fetch(0);
  for (int step = 0; step < steps; ++step) {
    stash_transposed(rb, sB, PL, tid);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t one = 0x3F80u;  // bf16 1.0
      const uint32_t e0 = code[0][j] == ck[j] ? one : 0u, e1 = code[1][j] == ck[j] ? one : 0u;
      const uint32_t e2 = code[2][j] == ck[j] ? one : 0u, e3 = code[3][j] == ck[j] ? one : 0u;
      *reinterpret_cast<uint2*>(sA + (4 * cq + j) * kLdP + 8 * mq) = make_uint2(e0 | (e1 << 16), e2 | (e3 << 16));
    }
    __syncthreads();
    if (step + 1 < steps) fetch(step + 1);
#pragma unroll
    for (int ks = 0; ks < kKC / 16; ++ks) {
      const int ko = ks * 32 + hi * 16;
      bf16x8 a[2], b[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = *reinterpret_cast<const bf16x8*>(sA + (wi * 64 + t * 32 + lj) * kLdP + ko);
#pragma unroll
        for (int p = 0; p < 3; ++p)
          b[t][p] = *reinterpret_cast<const bf16x8*>(sB + p * PL + (wj * 64 + t * 32 + lj) * kLdP + ko);
      }
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
          floatx16 c = acc[ti][tj];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti], b[tj][2], c, 0, 0, 0);  // small pieces first
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti], b[tj][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti], b[tj][0], c, 0, 0, 0);
          acc[ti][tj] = c;
        }
    }
    __syncthreads();
  }

  float* pw = partial + static_cast<int64_t>(split) * NC * D;
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const int k = k0 + wj * 64 + tj * 32 + lj;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wi * 64 + ti * 32 + acc_row(r, hi);
        if (n < NC && k < D) pw[static_cast<int64_t>(n) * D + k] = acc[ti][tj][r];
      }
    }
}

// C(m, n) = sum_k A(m, k) B(k, n) for SMALL operands with arbitrary element strides (codebook-sized products: the
// project_out table P_h = embed_h W_out_h^T and dW_out_h = S_h^T embed_h): plain fp32 FMA in a fixed k order --
// exact fp32 arithmetic, bit-reproducible.  blockIdx.z = batch.
struct SmallGemm {
  const float* a; int64_t a_m, a_k, a_b;   // element strides of A along m, k and per batch
  const float* b; int64_t b_k, b_n, b_b;
  float* c; int64_t c_m, c_n, c_b;
  int M, N, K;
};
template <int T>  // T x T output tile per block: 32 (many small blocks: codebook-sized problems are latency bound) or 64
__global__ void __launch_bounds__(kBlock) k_small_gemm(SmallGemm p) {
  constexpr int KC = 32, R = T / 16, PER = T * KC / kBlock;  // outputs per thread R x R, staged elements per thread
  __shared__ float sA[KC][T + 1], sB[KC][T + 1];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.x * T, n0 = blockIdx.y * T;
  const float* A = p.a + blockIdx.z * p.a_b;
  const float* B = p.b + blockIdx.z * p.b_b;
  float* C = p.c + blockIdx.z * p.c_b;
  float acc[R][R];
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int j = 0; j < R; ++j) acc[i][j] = 0.f;
  float ra[PER], rb[PER];
  auto fetch = [&](int k0) {  // the next chunk travels in registers while the current one is multiplied
#pragma unroll
    for (int t = 0; t < PER; ++t) {
      const int idx = t * kBlock + tid;
      const int kk = idx / T, mm = idx % T;
      const int k = k0 + kk;
      ra[t] = (k < p.K && m0 + mm < p.M) ? A[static_cast<int64_t>(m0 + mm) * p.a_m + k * p.a_k] : 0.f;
      rb[t] = (k < p.K && n0 + mm < p.N) ? B[static_cast<int64_t>(k) * p.b_k + (n0 + mm) * p.b_n] : 0.f;
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < p.K; k0 += KC) {
#pragma unroll
    for (int t = 0; t < PER; ++t) {
      const int idx = t * kBlock + tid;
      sA[idx / T][idx % T] = ra[t];
      sB[idx / T][idx % T] = rb[t];
    }
    __syncthreads();
    if (k0 + KC < p.K) fetch(k0 + KC);
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {
      float a[R], b[R];
#pragma unroll
      for (int i = 0; i < R; ++i) { a[i] = sA[kk][ty + 16 * i]; b[i] = sB[kk][tx + 16 * i]; }
#pragma unroll
      for (int i = 0; i < R; ++i)
#pragma unroll
        for (int j = 0; j < R; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int m = m0 + ty + 16 * i, n = n0 + tx + 16 * j;
      if (m < p.M && n < p.N) C[static_cast<int64_t>(m) * p.c_m + static_cast<int64_t>(n) * p.c_n] = acc[i][j];
    }
}

// out[i] = sum_s partial[s][i]: 16 float4 columns x 16 split-slices per block, fixed-order LDS
// tree over the slices (deterministic).
__global__ void __launch_bounds__(kBlock)
k_reduce_splits(const float* __restrict__ partial, int splits, int64_t n, float* __restrict__ out,
                const float* __restrict__ partial2, int64_t n2, float* __restrict__ out2, int blocks1) {
  // blocks [0, blocks1) reduce the first slab family (dW), the rest the second (db), in one launch
  __shared__ float4 red[kBlock];
  const int col = threadIdx.x & 15, slice = threadIdx.x >> 4;
  int bx = blockIdx.x;
  if (bx >= blocks1) { bx -= blocks1; partial = partial2; n = n2; out = out2; }
  const int64_t i = (static_cast<int64_t>(bx) * 16 + col) * 4;
  float4 a = zero4();
  if (i < n) {
    // four slabs in flight per thread (a one-at-a-time walk is a chain of memory latencies: 0.93 of the wave cycles
    // at a s_waitcnt), added in the same order as before
    int s = slice;
    for (; s + 48 < splits; s += 64) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = ld4(partial + static_cast<int64_t>(s + 16 * u) * n + i);
#pragma unroll
      for (int u = 0; u < 4; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    for (; s < splits; s += 16) {
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

// The same reduction for several slab families in one launch (a phase's weight and bias gradients).
constexpr int kReduceItems = 2 * kDwJobs;
struct ReduceItem {
  const float* partial;
  float* out;
  int64_t n;
  int splits, block_end;
};
struct ReduceTable {
  ReduceItem item[kReduceItems];
  int count;
};
__global__ void __launch_bounds__(kBlock) k_reduce_many(const ReduceTable tab) {
  __shared__ float4 red[kBlock];
  const int col = threadIdx.x & 15, slice = threadIdx.x >> 4;
  int j = 0, begin = 0;
  while (j + 1 < tab.count && static_cast<int>(blockIdx.x) >= tab.item[j].block_end) { begin = tab.item[j].block_end; ++j; }
  const float* __restrict__ partial = tab.item[j].partial;
  const int64_t n = tab.item[j].n;
  const int splits = tab.item[j].splits;
  const int64_t i = (static_cast<int64_t>(static_cast<int>(blockIdx.x) - begin) * 16 + col) * 4;
  float4 a = zero4();
  if (i < n) {
    // four slabs in flight per thread (a one-at-a-time walk is a chain of memory latencies: 0.93 of the wave cycles
    // at a s_waitcnt), added in the same order as before
    int s = slice;
    for (; s + 48 < splits; s += 64) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = ld4(partial + static_cast<int64_t>(s + 16 * u) * n + i);
#pragma unroll
      for (int u = 0; u < 4; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    for (; s < splits; s += 16) {
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
  if (slice == 0 && i < n) st4(tab.item[j].out + i, red[threadIdx.x]);
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

// Which matrix-core path the dense products take: 1 (default) = fp32 results from three exact bf16 pieces per operand;
// 0 = v_mfma_f32_32x32x2_f32 (kept as the cross-check of the split product); 2 = bf16 GEMMs (BASELINE config 5,
// what autocast would do to the reference's Linears) -- both operands of every product
// ROUNDED to bf16 while staged, one matrix pass, fp32 accumulation, fp32 bias / output / bias gradient; the quantiser's
// similarity / arg-max core keeps the exact form (vq.py:623,634 force fp32 there).  Mode 2 runs on the tile kernels
// only (the weight-stationary, few-row and fused-quantiser kernels are exact-form kernels and stand aside).
// STEMGNN_GEMM=f32 / bf16 in the environment starts the process in mode 0 / 2; stemgnn_linear_set_mode switches at run
// time.
std::atomic<int> g_gemm_mode{-1};
inline int gemm_mode() {
  int m = g_gemm_mode.load(std::memory_order_relaxed);
  if (m < 0) {
    const char* e = getenv("STEMGNN_GEMM");
    m = (e && e[0] == 'f') ? 0 : ((e && e[0] == 'b') ? 2 : 1);
    g_gemm_mode.store(m, std::memory_order_relaxed);
  }
  return m;
}
// The large products (D = 768 configurations) run on the big-tile core (csrc/bigtile.hip) in the exact mode and in the
// bf16 GEMM mode alike; stemgnn_linear_set_bigtile(0) keeps them on the 128-row tile kernels (tests compare the two).
inline int bigtile_pieces() { const int m = gemm_mode(); return m == 1 ? 3 : (m == 2 ? 1 : 0); }  // 0: fp32-MFMA mode, not served
inline bool bigtile(int64_t M, int64_t N, int64_t K) { return bigtile_pieces() != 0 && bt_gemm_ok(M, N, K); }
// runs a bt_* call: STEMGNN_OK = served; STEMGNN_ERR_WORKSPACE = no arena (counted; the caller goes on to the tile
// kernels); anything else is an error the caller must return
#define STEMGNN_TRY_BIGTILE(call)                                  \
  do {                                                             \
    const int rc_bt__ = (call);                                    \
    if (rc_bt__ == STEMGNN_OK) { bt_served(); return STEMGNN_OK; } \
    if (rc_bt__ != STEMGNN_ERR_WORKSPACE) return rc_bt__;          \
    bt_missed();                                                   \
  } while (0)
inline bool gemm_x3() { return gemm_mode() == 1; }        // the exact three-piece form
inline bool gemm_bf16() { return gemm_mode() == 2; }      // one rounded piece
inline bool gemm_matrix_bf16() { return gemm_mode() >= 1; }  // either: the bf16 matrix-core tile kernels

// Resident-block slots of the 128-row forward tile kernel on this device (blocks per CU x CUs).
template <bool X3>
inline int64_t fwd_slots_of() {
  static const int64_t slots = [] {
    int per_cu = 0, cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        (X3 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_linear_fwd_x3<128, false>, kBlock, 0)
            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_linear_fwd<128, false>, kBlock, 0)) !=
            hipSuccess ||
        per_cu <= 0 || cus <= 0) {
      (void)hipGetLastError();
      return static_cast<int64_t>((X3 ? 2 : 3) * 256);
    }
    return static_cast<int64_t>(per_cu) * cus;
  }();
  return slots;
}
inline int64_t fwd_slots() { return gemm_matrix_bf16() ? fwd_slots_of<true>() : fwd_slots_of<false>(); }

// Tile plan of the forward product: whole rounds of 128-row tiles, and (when the last round
// would be mostly empty) its rows as 32-row tiles.  Measured at N = 128 on MI355X: 768 tiles (one
// full round) run at 84 / 101 / 116 TFLOP/s for K = 128 / 256 / 512, 800 tiles at 66 / 85 / 95.
struct FwdPlan {
  int64_t main_tiles, tail_tiles;
};
inline FwdPlan plan_fwd(int64_t M, int64_t N) {
  const int64_t gy = (N + kBN - 1) / kBN;
  const int64_t tiles = (M + kBM - 1) / kBM;
  const int64_t slots = fwd_slots();
  const int64_t per_round = slots / gy > 0 ? slots / gy : 1;  // row tiles per resident round
  const int64_t rem = tiles % per_round;
  FwdPlan p{tiles, 0};
  if (tiles > per_round && rem > 0 && rem * 4 <= per_round) {
    p.main_tiles = tiles - rem;
    const int64_t rem_rows = M - p.main_tiles * kBM;
    p.tail_tiles = (rem_rows + 31) / 32;
  }
  return p;
}

// Row splits of the weight-gradient product.  Measured on MI355X (M = 102400): the kernel wants
// ~512 blocks in total (2 resident per CU) but the partial slabs (S * N * K floats, written and
// re-read) want S small: S = 512 / (#output tiles), capped at 256 (one tile: 58.8 us at S = 247
// vs 65 us at S = 458; four tiles: 134-149 us at S = 128 vs 176-183 us at S = 458).
inline int pick_splits(int64_t M, int64_t tiles) {
  int64_t target = 512 / (tiles < 1 ? 1 : tiles);
  const int64_t cap = 256;
  if (target > cap) target = cap;
  if (target < 1) target = 1;
  int64_t rows = (M + target - 1) / target;
  rows = (rows + kKC - 1) / kKC * kKC;
  if (rows < kKC) rows = kKC;
  int64_t s = (M + rows - 1) / rows;
  if (s < 1) s = 1;
  if (s > kMaxSplits) s = kMaxSplits;
  return static_cast<int>(s);
}

// the weight-stationary kernel wants most CUs busy: products with fewer 128-row tiles stay on the tile kernel.
// stemgnn_linear_set_ws(0) keeps every product there (the tests compare the two kernels' bits that way).
std::atomic<int> g_ws_min_tiles{128};  // 0: off
inline int64_t ws_min_tiles() { return g_ws_min_tiles.load(std::memory_order_relaxed); }
inline bool ws_enabled() { return ws_min_tiles() > 0; }
#define kWsMinTiles ws_min_tiles()

// One launch of the forward tile kernel in the process's matrix-core mode (0 fp32 MFMA, 1 exact bf16 pieces, 2 bf16
// GEMM), for fp32 or bf16-stored second operands, with or without the BatchNorm column sums.
template <int BM>
inline void launch_fwd_tile(int mode, bool bf, dim3 grid, hipStream_t st, const float* x1, const float* w1, int k1,
                            const float* x2, const float* w2, int k2, const float* bias, int64_t M, int n, float* y,
                            float* stats, int64_t row_base, int64_t stats_block0, int64_t x1r, int64_t sr) {
#define STEMGNN_FWD_ARGS x1, w1, k1, x2, w2, k2, bias, M, n, y, stats, row_base, stats_block0, x1r, sr
  if (mode == 0) {
    if (stats) k_linear_fwd<BM, true><<<grid, kBlock, 0, st>>>(STEMGNN_FWD_ARGS);
    else k_linear_fwd<BM, false><<<grid, kBlock, 0, st>>>(STEMGNN_FWD_ARGS);
  } else if (mode == 1) {
    if (stats && bf) k_linear_fwd_x3<BM, true, false, kBF16><<<grid, kBlock, 0, st>>>(STEMGNN_FWD_ARGS);
    else if (stats) k_linear_fwd_x3<BM, true><<<grid, kBlock, 0, st>>>(STEMGNN_FWD_ARGS);
    else if (bf) k_linear_fwd_x3<BM, false, false, kBF16><<<grid, kBlock, 0, st>>>(STEMGNN_FWD_ARGS);
    else k_linear_fwd_x3<BM, false><<<grid, kBlock, 0, st>>>(STEMGNN_FWD_ARGS);
  } else {
    if (stats && bf) k_linear_fwd_x3<BM, true, false, kBF16, 1><<<grid, kBlock, 0, st>>>(STEMGNN_FWD_ARGS);
    else if (stats) k_linear_fwd_x3<BM, true, false, kF32, 1><<<grid, kBlock, 0, st>>>(STEMGNN_FWD_ARGS);
    else if (bf) k_linear_fwd_x3<BM, false, false, kBF16, 1><<<grid, kBlock, 0, st>>>(STEMGNN_FWD_ARGS);
    else k_linear_fwd_x3<BM, false, false, kF32, 1><<<grid, kBlock, 0, st>>>(STEMGNN_FWD_ARGS);
  }
#undef STEMGNN_FWD_ARGS
}

inline int64_t out_tiles(int64_t N, int64_t K) { return ((N + kBN - 1) / kBN) * ((K + kBN - 1) / kBN); }

}  // namespace
}  // namespace stemgnn

namespace stemgnn {
int DwBatch::add(const float* dy, const void* x, int x_kind, int64_t M, int64_t N, int64_t K, float* dw, float* db,
                 void* workspace, size_t workspace_bytes, hipStream_t st) {
  if (x_kind != kF32 && x_kind != kBF16) return STEMGNN_ERR_INVALID_ARG;
  if (x_kind == kBF16 && !gemm_matrix_bf16()) return STEMGNN_ERR_INVALID_ARG;  // the fp32-MFMA cross-check twins read fp32 only
  if (!lin_dims_ok(M, N, K) || N % 4 != 0 || !dw) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(M)) return STEMGNN_ERR_TOO_LARGE;
  if (M == 0) {
    STEMGNN_HIP_TRY(hipMemsetAsync(dw, 0, sizeof(float) * N * K, st));
    if (db) STEMGNN_HIP_TRY(hipMemsetAsync(db, 0, sizeof(float) * N, st));
    return STEMGNN_OK;
  }
  if (!dy || !x || !workspace) return STEMGNN_ERR_INVALID_ARG;
  if (workspace_bytes < stemgnn_linear_bwd_weight_workspace_bytes(M, N, K)) return STEMGNN_ERR_WORKSPACE;
  // a large product: on the big-tile core at once (nothing to queue)
  if (bigtile(M, N, K)) STEMGNN_TRY_BIGTILE(bt_linear_bwd_weight(bigtile_pieces(), dy, x, x_kind, M, N, K, dw, db, st));
  if (count == kMax) {
    const int rc = flush(st);
    if (rc != STEMGNN_OK) return rc;
  }
  Job& j = jobs[count++];
  j.splits = pick_splits(M, out_tiles(N, K));
  j.rows = (M + j.splits - 1) / j.splits;
  j.rows = (j.rows + kKC - 1) / kKC * kKC;
  j.dy = dy; j.x = x; j.kind = x_kind; j.M = M; j.N = N; j.K = K; j.dw = dw; j.db = db;
  j.pw = reinterpret_cast<float*>(align_up(reinterpret_cast<uintptr_t>(workspace), 256));
  j.pb = j.pw + static_cast<size_t>(j.splits) * N * K;
  return STEMGNN_OK;
}

int DwBatch::flush(hipStream_t st) {
  if (count == 0) return STEMGNN_OK;
  if (gemm_matrix_bf16()) {
    const bool one = gemm_bf16();
    for (int kind = kF32; kind <= kBF16; ++kind) {
      DwTable tab;
      tab.count = 0;
      int blocks = 0;
      for (int i = 0; i < count; ++i) {
        const Job& j = jobs[i];
        if (j.kind != kind) continue;
        DwJob& t = tab.job[tab.count++];
        t.dy = j.dy; t.x = j.x; t.pw = j.pw; t.pb = j.db ? j.pb : nullptr;
        t.M = j.M; t.rows_per_split = j.rows;
        t.N = static_cast<int>(j.N); t.K = static_cast<int>(j.K); t.splits = j.splits;
        t.ktiles = static_cast<int>((j.K + kBN - 1) / kBN);
        blocks += j.splits * static_cast<int>(out_tiles(j.N, j.K));
        t.block_end = blocks;
      }
      if (tab.count == 0) continue;
      if (kind == kBF16 && one) k_linear_bwd_weight_x3<kBF16, 1><<<static_cast<unsigned>(blocks), kBlock, 0, st>>>(tab);
      else if (kind == kBF16) k_linear_bwd_weight_x3<kBF16><<<static_cast<unsigned>(blocks), kBlock, 0, st>>>(tab);
      else if (one) k_linear_bwd_weight_x3<kF32, 1><<<static_cast<unsigned>(blocks), kBlock, 0, st>>>(tab);
      else k_linear_bwd_weight_x3<kF32><<<static_cast<unsigned>(blocks), kBlock, 0, st>>>(tab);
      STEMGNN_LAUNCH_CHECK();
    }
  } else {
    for (int i = 0; i < count; ++i) {
      const Job& j = jobs[i];
      dim3 grid(static_cast<unsigned>(j.splits), static_cast<unsigned>((j.N + kBN - 1) / kBN),
                static_cast<unsigned>((j.K + kBN - 1) / kBN));
      k_linear_bwd_weight<<<grid, kBlock, 0, st>>>(j.dy, static_cast<const float*>(j.x), j.M, static_cast<int>(j.N),
                                                   static_cast<int>(j.K), j.rows, j.pw, j.db ? j.pb : nullptr);
      STEMGNN_LAUNCH_CHECK();
    }
  }
  ReduceTable red;
  red.count = 0;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    const Job& j = jobs[i];
    ReduceItem& a = red.item[red.count++];
    a.partial = j.pw; a.out = j.dw; a.n = j.N * j.K; a.splits = j.splits;
    blocks += static_cast<int>((a.n / 4 + 15) / 16);
    a.block_end = blocks;
    if (j.db) {
      ReduceItem& b = red.item[red.count++];
      b.partial = j.pb; b.out = j.db; b.n = j.N; b.splits = j.splits;
      blocks += static_cast<int>((b.n / 4 + 15) / 16);
      b.block_end = blocks;
    }
  }
  k_reduce_many<<<static_cast<unsigned>(blocks), kBlock, 0, st>>>(red);
  STEMGNN_LAUNCH_CHECK();
  count = 0;
  return STEMGNN_OK;
}
}  // namespace stemgnn

using namespace stemgnn;

extern "C" {

int stemgnn_linear_few_rows(const float* x, const float* w, const float* bias, int64_t M, int64_t N, int64_t K, float* y,
                            int32_t weight_is_kn, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (M == 0) return STEMGNN_OK;
  if (!gemm_x3() || !linear_direct_ok(M, N, K)) return STEMGNN_ERR_INVALID_ARG;
  DirectBatch b;
  const int rc = b.add(x, w, bias, M, N, K, y, weight_is_kn != 0, st);
  if (rc != STEMGNN_OK) return rc;
  return b.flush(st);
}

int stemgnn_linear_set_ws(int min_tiles) {
  const int prev = static_cast<int>(ws_min_tiles());
  if (min_tiles >= 0) g_ws_min_tiles.store(min_tiles, std::memory_order_relaxed);
  return prev;
}

int stemgnn_linear_set_mode(int mode) {
  const int prev = gemm_mode();
  if (mode == 0 || mode == 1 || mode == 2) g_gemm_mode.store(mode, std::memory_order_relaxed);
  return prev;
}

size_t stemgnn_linear_stats_partial_bytes(int64_t M, int64_t N) {
  if (M < 0 || N <= 0) return 0;
  return static_cast<size_t>((M + 31) / 32 + 2) * 2 * N * sizeof(float) + 256;  // upper bound over every tile plan
}

int64_t stemgnn_linear_stats_blocks(int64_t M, int64_t N) {
  if (M < 0 || N <= 0) return 0;
  const FwdPlan p = plan_fwd(M, N);
  return 2 * p.main_tiles + p.tail_tiles;  // a slab per 64-row half of a 128-row tile, one per 32-row tile
}

int stemgnn_linear_fwd(const float* x1, const float* w1, int64_t K1, const float* x2, const float* w2, int64_t K2,
                       const float* bias, int64_t M, int64_t N, float* y, float* stats_partial,
                       int64_t* stats_blocks_host, int64_t x1_rows, void* stream_) {
  return stemgnn_linear_fwd_rows(x1, w1, K1, x2, w2, K2, bias, M, N, y, stats_partial, stats_blocks_host, x1_rows, M,
                                 stream_);
}

int stemgnn_linear_fwd_rows(const float* x1, const float* w1, int64_t K1, const float* x2, const float* w2, int64_t K2,
                            const float* bias, int64_t M, int64_t N, float* y, float* stats_partial,
                            int64_t* stats_blocks_host, int64_t x1_rows, int64_t store_rows, void* stream_) {
  return stemgnn_linear_fwd_rows_k(x1, w1, K1, x2, kF32, w2, K2, bias, M, N, y, stats_partial, stats_blocks_host, x1_rows,
                                   store_rows, stream_);
}

int stemgnn_linear_fwd_rows_k(const float* x1, const float* w1, int64_t K1, const void* x2_, int32_t x2_kind,
                              const float* w2, int64_t K2, const float* bias, int64_t M, int64_t N, float* y,
                              float* stats_partial, int64_t* stats_blocks_host, int64_t x1_rows, int64_t store_rows,
                              void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (x2_kind != kF32 && x2_kind != kBF16) return STEMGNN_ERR_INVALID_ARG;
  const float* x2 = static_cast<const float*>(x2_);
  const bool bf = x2_kind == kBF16 && K2 > 0;
  if (bf && !gemm_matrix_bf16()) return STEMGNN_ERR_INVALID_ARG;  // the fp32-MFMA cross-check twins read fp32 only
  const int64_t sr = (store_rows < 0 || store_rows > M) ? M : store_rows;
  if (!lin_dims_ok(M, N, K1) || K2 < 0 || K2 % 4 != 0) return STEMGNN_ERR_INVALID_ARG;
  const int64_t x1r = (x1_rows < 0 || x1_rows > M) ? M : x1_rows;
  if (!fits_i32(M)) return STEMGNN_ERR_TOO_LARGE;
  if (stats_blocks_host) *stats_blocks_host = stemgnn_linear_stats_blocks(M, N);
  if (M == 0) return STEMGNN_OK;
  if (!x1 || !w1 || !y || (K2 > 0 && (!x2 || !w2))) return STEMGNN_ERR_INVALID_ARG;
  if (N % 4 != 0) return STEMGNN_ERR_INVALID_ARG;
  const int gy = static_cast<int>((N + kBN - 1) / kBN);
  const FwdPlan plan = plan_fwd(M, N);
  const int k1 = static_cast<int>(K1), k2 = static_cast<int>(K2), n = static_cast<int>(N);
  const bool x3 = gemm_x3();
  // Weight-stationary kernel (csrc/wsgemm.hip) for a product with ONE 128-column operand.  (Splitting a sampled
  // batch's layer product -- leading tiles with both operands on the tile kernel, the rest there -- was measured: the
  // small leading launch costs more than the rest gains, 54 vs 41 us per layer at C4.)
  if (x3 && !bf && ws_enabled() && N % kBN == 0) {
    const int64_t tiles = (M + kBM - 1) / kBM;
    const float *xs = nullptr, *wsrc = nullptr;
    if (K2 == 0 && K1 == 128 && x1r >= M) { xs = x1; wsrc = w1; }
    else if (K2 == 128 && x1r == 0) { xs = x2; wsrc = w2; }  // no row carries the first operand
    // pair format (csrc/wspair.hip): three matrix passes instead of six, and the sampled batch's two-operand layer product
    // too (the leading rows' aggregate is multiplied in the blocks' prologues)
    const float *xh = nullptr, *wh = nullptr;
    int64_t hr = 0;
    const bool pair = linear_pair_on();
    if (pair && !xs && K1 == 128 && K2 == 128 && x1r > 0 && x1r < M && linear_wsp_ok(M, N, 128, x1r)) {
      xs = x2; wsrc = w2; xh = x1; wh = w1; hr = x1r;
    }
    if (xs && tiles >= kWsMinTiles) {
      const int rc = pair ? linear_wsp_launch(xs, wsrc, bias, M, N, y, stats_partial, 0, sr, false, xh, wh, hr, st)
                          : linear_ws_launch(xs, wsrc, bias, M, N, 128, y, stats_partial, 0, 0, sr, false, st);
      if (rc != STEMGNN_OK) return rc;
      // what callers reduce over (stemgnn_linear_stats_blocks): the slabs past the 64-row tiles written here are zero
      const int64_t count = 2 * plan.main_tiles + plan.tail_tiles, written = (M + 63) / 64;
      if (stats_partial && count > written)
        STEMGNN_HIP_TRY(hipMemsetAsync(stats_partial + written * 2 * N, 0, sizeof(float) * (count - written) * 2 * N, st));
      return STEMGNN_OK;
    }
  }
  const int mode = gemm_mode();
  if (bigtile(M, N, K1 + K2))  // a large product: cut pass + the big-tile core
    STEMGNN_TRY_BIGTILE(bt_linear_fwd(bigtile_pieces(), x1, w1, K1, x2_, x2_kind, w2, K2, bias, M, N, y, x1r, sr,
                                      stats_partial, 2 * plan.main_tiles + plan.tail_tiles, st));
  if (plan.main_tiles > 0) {
    dim3 grid(static_cast<unsigned>(plan.main_tiles), static_cast<unsigned>(gy));
    launch_fwd_tile<128>(mode, bf, grid, st, x1, w1, k1, x2, w2, k2, bias, M, n, y, stats_partial, 0, 0, x1r, sr);
    STEMGNN_LAUNCH_CHECK();
  }
  if (plan.tail_tiles > 0) {
    // the rows of the last, mostly empty round of 128-row tiles run as 32-row tiles: 4x the blocks,
    // a quarter of the latency each, instead of a handful of full-size stragglers
    dim3 grid(static_cast<unsigned>(plan.tail_tiles), static_cast<unsigned>(gy));
    launch_fwd_tile<32>(mode, bf, grid, st, x1, w1, k1, x2, w2, k2, bias, M, n, y, stats_partial, plan.main_tiles * kBM,
                        2 * plan.main_tiles, x1r, sr);
    STEMGNN_LAUNCH_CHECK();
  }
  return STEMGNN_OK;
}

int stemgnn_linear_bwd_data(const float* dy, const float* w, int64_t M, int64_t N, int64_t K, float* dx,
                            void* stream_) {
  // dx[M, K] = dy[M, N] w[N, K]: the forward tile with the weight read as stored (contraction index slow) and
  // transposed while it is staged, instead of a separate transpose kernel and a W^T buffer per call
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (!lin_dims_ok(M, K, N) || K % 4 != 0) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(M)) return STEMGNN_ERR_TOO_LARGE;
  if (M == 0) return STEMGNN_OK;
  if (!dy || !w || !dx) return STEMGNN_ERR_INVALID_ARG;
  const int gy = static_cast<int>((K + kBN - 1) / kBN);
  const FwdPlan plan = plan_fwd(M, K);
  const int kc = static_cast<int>(N), n = static_cast<int>(K);
  const bool x3 = gemm_x3();
  if (x3 && ws_enabled() && N == 128 && K % kBN == 0 && (M + kBM - 1) / kBM >= kWsMinTiles)
    return linear_pair_on() ? linear_wsp_launch(dy, w, nullptr, M, K, dx, nullptr, 0, M, true, nullptr, nullptr, 0, st)
                            : linear_ws_launch(dy, w, nullptr, M, K, 128, dx, nullptr, 0, 0, M, true, st);
  const int mode = gemm_mode();
  if (bigtile(M, K, N)) STEMGNN_TRY_BIGTILE(bt_linear_bwd_data(bigtile_pieces(), dy, w, M, N, K, dx, st));
  if (plan.main_tiles > 0) {
    dim3 grid(static_cast<unsigned>(plan.main_tiles), static_cast<unsigned>(gy));
    if (mode == 2) k_linear_fwd_x3<128, false, true, kF32, 1><<<grid, kBlock, 0, st>>>(dy, w, kc, nullptr, nullptr, 0, nullptr, M, n, dx, nullptr, 0, 0, M, M);
    else if (mode == 1) k_linear_fwd_x3<128, false, true><<<grid, kBlock, 0, st>>>(dy, w, kc, nullptr, nullptr, 0, nullptr, M, n, dx, nullptr, 0, 0, M, M);
    else k_linear_fwd<128, false, true><<<grid, kBlock, 0, st>>>(dy, w, kc, nullptr, nullptr, 0, nullptr, M, n, dx, nullptr, 0, 0, M, M);
    STEMGNN_LAUNCH_CHECK();
  }
  if (plan.tail_tiles > 0) {
    dim3 grid(static_cast<unsigned>(plan.tail_tiles), static_cast<unsigned>(gy));
    const int64_t row_base = plan.main_tiles * kBM;
    if (mode == 2) k_linear_fwd_x3<32, false, true, kF32, 1><<<grid, kBlock, 0, st>>>(dy, w, kc, nullptr, nullptr, 0, nullptr, M, n, dx, nullptr, row_base, 0, M, M);
    else if (mode == 1) k_linear_fwd_x3<32, false, true><<<grid, kBlock, 0, st>>>(dy, w, kc, nullptr, nullptr, 0, nullptr, M, n, dx, nullptr, row_base, 0, M, M);
    else k_linear_fwd<32, false, true><<<grid, kBlock, 0, st>>>(dy, w, kc, nullptr, nullptr, 0, nullptr, M, n, dx, nullptr, row_base, 0, M, M);
    STEMGNN_LAUNCH_CHECK();
  }
  return STEMGNN_OK;
}

size_t stemgnn_linear_bwd_weight_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (!lin_dims_ok(M, N, K)) return 0;
  return static_cast<size_t>(pick_splits(M, out_tiles(N, K))) * (N * K + N) * sizeof(float) + 512;
}

int stemgnn_linear_bwd_weight(const float* dy, const float* x, int64_t M, int64_t N, int64_t K, float* dw, float* db,
                              void* workspace, size_t workspace_bytes, void* stream_) {
  return stemgnn_linear_bwd_weight_k(dy, x, kF32, M, N, K, dw, db, workspace, workspace_bytes, stream_);
}

int stemgnn_linear_bwd_weight_k(const float* dy, const void* x_, int32_t x_kind, int64_t M, int64_t N, int64_t K, float* dw,
                                float* db, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  DwBatch batch;
  const int rc = batch.add(dy, x_, x_kind, M, N, K, dw, db, workspace, workspace_bytes, st);
  if (rc != STEMGNN_OK) return rc;
  return batch.flush(st);
}

size_t stemgnn_code_segment_sums_workspace_bytes(int64_t M, int64_t H, int64_t K, int64_t D) {
  if (M < 0 || H <= 0 || K <= 0 || D <= 0 || D % 4 != 0) return 0;
  return static_cast<size_t>(pick_splits(M, out_tiles(H * K, D))) * (H * K) * D * sizeof(float) + 512;
}

int stemgnn_code_segment_sums(const int64_t* ind, int64_t H, int64_t K, const float* g, int64_t M, int64_t D,
                              float* sums, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (M < 0 || H <= 0 || K <= 0 || D <= 0 || D % 4 != 0 || H * K > (1 << 24) || !sums) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(M)) return STEMGNN_ERR_TOO_LARGE;
  const int64_t NC = H * K;
  if (M == 0) {
    STEMGNN_HIP_TRY(hipMemsetAsync(sums, 0, sizeof(float) * NC * D, st));
    return STEMGNN_OK;
  }
  if (!ind || !g || !workspace) return STEMGNN_ERR_INVALID_ARG;
  if (workspace_bytes < stemgnn_code_segment_sums_workspace_bytes(M, H, K, D)) return STEMGNN_ERR_WORKSPACE;
  const int S = pick_splits(M, out_tiles(NC, D));
  int64_t rows = (M + S - 1) / S;
  rows = (rows + kKC - 1) / kKC * kKC;
  float* pw = reinterpret_cast<float*>(align_up(reinterpret_cast<uintptr_t>(workspace), 256));
  if (D % 32 == 0 && H <= 16) {
    rows = (rows + 127) / 128 * 128;  // its steps are 128 rows
    dim3 grid(static_cast<unsigned>(S), static_cast<unsigned>(D / 32), static_cast<unsigned>((NC + 511) / 512));
    k_code_segment_sums_cols<<<grid, kBlock, 0, st>>>(ind, static_cast<int>(H), static_cast<int>(K), g, M,
                                                      static_cast<int>(D), rows, pw);
  } else {
    dim3 grid(static_cast<unsigned>(S), static_cast<unsigned>((NC + kBN - 1) / kBN), static_cast<unsigned>((D + kBN - 1) / kBN));
    k_code_segment_sums<<<grid, kBlock, 0, st>>>(ind, static_cast<int>(H), static_cast<int>(K), g, M, static_cast<int>(D),
                                                 rows, pw);
  }
  STEMGNN_LAUNCH_CHECK();
  const int64_t nk = NC * D;
  const int blocks1 = static_cast<int>((nk / 4 + 15) / 16);
  k_reduce_splits<<<static_cast<unsigned>(blocks1), kBlock, 0, st>>>(pw, S, nk, sums, nullptr, 0, nullptr, blocks1);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_small_gemm(const float* a, int64_t a_m, int64_t a_k, int64_t a_batch, const float* b, int64_t b_k,
                       int64_t b_n, int64_t b_batch, float* c, int64_t c_m, int64_t c_n, int64_t c_batch, int64_t M,
                       int64_t N, int64_t K, int64_t batches, void* stream_) {
  if (M <= 0 || N <= 0 || K <= 0 || batches <= 0 || batches > 65535 || M > (1 << 24) || N > (1 << 24) || K > (1 << 24) ||
      !a || !b || !c)
    return STEMGNN_ERR_INVALID_ARG;
  SmallGemm p{a, a_m, a_k, a_batch, b, b_k, b_n, b_batch, c, c_m, c_n, c_batch, static_cast<int>(M),
              static_cast<int>(N), static_cast<int>(K)};
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (((M + 63) / 64) * ((N + 63) / 64) * batches >= 512) {
    dim3 grid(static_cast<unsigned>((M + 63) / 64), static_cast<unsigned>((N + 63) / 64), static_cast<unsigned>(batches));
    k_small_gemm<64><<<grid, kBlock, 0, st>>>(p);
  } else {
    dim3 grid(static_cast<unsigned>((M + 31) / 32), static_cast<unsigned>((N + 31) / 32), static_cast<unsigned>(batches));
    k_small_gemm<32><<<grid, kBlock, 0, st>>>(p);
  }
  STEMGNN_LAUNCH_CHECK();
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
