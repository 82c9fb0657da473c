// RECORD of a round-3 experiment, NOT part of libstemgnn_hip.so.  It was built into the library (dispatched for products
// of >= 256 output tiles), returned the tile kernel's bits on every shape (tests/test_gpu_kernels.py at that commit) and
// was measured on C4-batch shapes (M = 102 400; us, tools/micro/sp_probe.py):
//     forward K = 128 + 128, N = 128 (+ statistics)   tile 63.9   weight-stationary  --    this 71.7
//     forward K = 128, N = 512 (project_in)           tile 115.6  weight-stationary 81.5  this 117.4
//     forward K = 128, N = 128                        tile 40.2   weight-stationary 26.8  this 36.7
//     backward-data N = 512 -> K = 128                tile 101.1                          this 113.6
// i.e. no faster than the kernels it would replace.  Ablation of THIS kernel (STEMGNN_SP_DBG bits, K = 256, N = 128,
// 63.3 us whole): no output stores 58.5, no matrix instructions 52.3, no global loads 48.8, neither loads nor matrix
// instructions 38.6, none of the three 33.9, and with the cut into bf16 pieces removed as well 17.0 (barriers, fragment
// reads, staging).  So: skeleton 17 + cut 17 + matrix 29 alone (11 on top of the rest) + loads 14 + stores 5 -- five
// comparable terms that only partly overlap even with dedicated loader waves; no single phase to remove.  What bounds
// these products is the SUM of an HBM pass and six matrix products per fp32 product, not missing overlap inside a CU.
//
// Dense products with SPECIALISED WAVES (round 3): y = x1 w1^T (+ x2 w2^T) + b on the bf16 matrix cores with the exact
// three-piece operands of csrc/linear.hip, the same 128 x 128 tile, k-step order and arithmetic (bit-identical
// results) -- but the two halves of the work no longer take turns inside one wave.
//
// Reference ops replaced: lin_l(agg) + lin_r(x) (STEM-GNN/model/encoder.py:83-87), VectorQuantize.project_in and its
// backward-data (model/vq.py:881), the decoders' Linears (model/pt_model.py:42-43,80-81).
//
// Why: in k_linear_fwd_x3 every wave loads a chunk, cuts it into bf16 pieces (~5.5 vector instructions per element),
// writes it to LDS, waits at a barrier, multiplies, waits at a barrier -- the phases ADD (tools/micro/pgemm_ablate.py,
// DESIGN.md section K3), the matrix pipe is busy a quarter of the time (768 of ~3 000 cycles per 32-wide chunk and CU),
// and only a second resident block fills gaps.  Here a 512-thread block (one per CU) is eight waves, two per SIMD:
//   waves 0-3  LOADERS    global loads -> cut -> LDS planes of chunk c + 1 (and the loads of chunk c + 2 in flight)
//   waves 4-7  CONSUMERS  LDS fragment reads + matrix instructions of chunk c
// A SIMD's matrix pipe and its vector ALU issue from different waves at the same time (MI355X_MICROARCH.md: an MFMA
// holds vector issue for 8 of its 32 cycles), so the cut runs UNDER the matrix work instead of in front of it.  Two LDS
// slots, ONE barrier per chunk: behind barrier c the loaders write slot (c + 1) & 1 -- last read for chunk c - 1, which
// every consumer finished before it arrived -- while the consumers read slot c & 1.  The block is persistent over row
// tiles, so the loaders run ahead across tile boundaries and a tile's epilogue (staging tile in its own LDS region,
// 512-byte row stores by all 512 threads) overlaps the next tile's first loads.
//
// Barrier discipline: both roles execute the SAME sequence of __syncthreads() -- one per chunk, then the epilogue's
// (two per stored 64-row half, one for the statistics), all behind block-uniform conditions.
#include "common.h"
#include <cstdlib>

namespace stemgnn {
namespace {

constexpr int kThreads = 512, kLoaders = 256;
constexpr int kBM = 128, kBN = 128, kKC = 32;
constexpr int PA = kBM * kLdP, PB = kBN * kLdP;  // bytes of one bf16 plane of a chunk
constexpr int kSlot = 3 * PA + 3 * PB;           // 61 440 B: planes h, m, l of the activation and of the weight chunk
constexpr int kLdT = kBN + 4;                    // fp32 row stride of the epilogue's staging tile
constexpr int kStage = 64 * kLdT * 4;            // 33 792 B
constexpr int kStats = 2 * 2 * kBN * 4;          // [wave row][sum | sum of squares][column]
constexpr int kLds = 2 * kSlot + kStage + kStats;  // 158 720 B of the CU's 163 840

__device__ __forceinline__ int stage_row(int idx) {  // csrc/linear.hip: rows of a group of eight in the order 0 4 1 5 2 6 3 7
  const int r = idx >> 3;
  return (r & ~7) | ((r & 1) << 2) | ((r >> 1) & 3);
}

template <bool STATS, bool BT>
__global__ void __launch_bounds__(kThreads, 2)
k_linear_sp(const float* __restrict__ x1, const float* __restrict__ w1, int K1, const float* __restrict__ x2,
            const float* __restrict__ w2, int K2, const float* __restrict__ bias, int64_t M, int N,
            float* __restrict__ y, float* __restrict__ stats_partial /*[stats_slabs][2][N]*/, int64_t x1_rows,
            int64_t store_rows, int64_t tiles, int64_t stats_slabs, int dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const stage = reinterpret_cast<float*>(smem + 2 * kSlot);
  float* const s_stats = reinterpret_cast<float*>(smem + 2 * kSlot + kStage);  // [2][2][kBN]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool loader = wave < 4;
  const int ltid = tid & (kLoaders - 1);           // index within the role
  const int cw = wave & 3, wm = cw >> 1, wn = cw & 1, hi = lane >> 5, lj = lane & 31;  // consumer tile of 64 x 64
  const int n0 = blockIdx.y * kBN;
  const int c1 = (K1 + kKC - 1) / kKC, c2 = (K2 + kKC - 1) / kKC;
  const int steps = c1 + c2;
  const int64_t stride = gridDim.x;
  // rows >= x1_rows of the first operand are zero by promise and never read: a tile past them starts at the second
  // operand's chunks (the host only takes this kernel when every tile has at least one chunk)
  auto first_of = [&](int64_t t) { return (t * kBM >= x1_rows) ? c1 : 0; };

  // ---- loader state: the chunk held in registers (fetched, not yet cut): (f_t, f_s), live while f_t < tiles
  float4 ra[4], rb[4];
  int64_t f_t = blockIdx.x;
  int f_s = f_t < tiles ? first_of(f_t) : 0;
  auto fetch = [&]() {
    if (dbg & 4) return;
    const bool second = f_s >= c1;
    const float* xs = second ? x2 : x1;
    const float* ws = second ? w2 : w1;
    const int K = second ? K2 : K1;
    const int k0 = (second ? f_s - c1 : f_s) * kKC;
    const int64_t m0 = f_t * kBM;
    const int64_t mlim = second ? M : x1_rows;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = t * kLoaders + ltid;
      const int r = stage_row(idx), k = k0 + 4 * (idx & 7);
      if (BT) {  // weight given as [K][N]: rows k0 + 4 (ltid & 7) + t, columns n0 + 4 (ltid >> 3) .. + 3
        const int kk = k0 + 4 * (ltid & 7) + t, nn = n0 + 4 * (ltid >> 3);
        rb[t] = (kk < K && nn < N) ? ld4(ws + static_cast<int64_t>(kk) * N + nn) : zero4();
      } else {
        const int n = n0 + r;
        rb[t] = (n < N && k < K) ? ld4(ws + static_cast<int64_t>(n) * K + k) : zero4();
      }
      const int64_t m = m0 + r;
      ra[t] = (m < mlim && k < K) ? ld4(xs + m * K + k) : zero4();
    }
  };
  auto advance = [&]() {
    if (++f_s == steps) {
      f_t += stride;
      f_s = f_t < tiles ? first_of(f_t) : 0;
    }
  };
  auto stash = [&](unsigned char* slot) {
    if (dbg & 8) return;
    unsigned char* const sA = slot;
    unsigned char* const sB = slot + 3 * PA;
    if (BT) stash_transposed(rb, sB, PB, ltid);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = t * kLoaders + ltid;
      const int off = stage_row(idx) * kLdP + 8 * (idx & 7);
      uint2 h, m, l;
      if (!BT) {
        split3(rb[t], h, m, l);
        *reinterpret_cast<uint2*>(sB + off) = h;
        *reinterpret_cast<uint2*>(sB + PB + off) = m;
        *reinterpret_cast<uint2*>(sB + 2 * PB + off) = l;
      }
      split3(ra[t], h, m, l);
      *reinterpret_cast<uint2*>(sA + off) = h;
      *reinterpret_cast<uint2*>(sA + PA + off) = m;
      *reinterpret_cast<uint2*>(sA + 2 * PA + off) = l;
    }
  };

  // prologue: chunk 0 into slot 0, chunk 1 into the registers
  if (loader && f_t < tiles) {
    fetch();
    stash(smem);
    advance();
    if (f_t < tiles) fetch();
  }

  float bias_v[2];
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int n = n0 + wn * 64 + tn * 32 + lj;
    bias_v[tn] = (bias != nullptr && n < N) ? bias[n] : 0.f;
  }
  const int64_t row_end = store_rows < M ? store_rows : M;  // rows past it only feed the column statistics

  int c = 0;  // chunks this block has been through: chunk c lives in slot c & 1
  for (int64_t t = blockIdx.x; t < tiles; t += stride) {
    const int64_t m0 = t * kBM;
    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    for (int step = first_of(t); step < steps; ++step, ++c) {
      __syncthreads();  // chunk c is in slot c & 1; every consumer is done with chunk c - 1
      if (loader) {
        if (f_t < tiles) {  // chunk c + 1 -> the other slot, chunk c + 2 -> registers
          stash(smem + ((c + 1) & 1) * kSlot);
          advance();
          if (f_t < tiles) fetch();
        }
      } else {
        const unsigned char* const sA = smem + (c & 1) * kSlot;
        const unsigned char* const sB = sA + 3 * PA;
#pragma unroll
        for (int ks = 0; ks < kKC / 16; ++ks) {
          const int ko = ks * 32 + hi * 16;  // bytes: lane half 0 takes k 0..7, half 1 k 8..15 of the 16-wide step
          bf16x8 a[2][3], b[2][3];
#pragma unroll
          for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              a[q][p] = *reinterpret_cast<const bf16x8*>(sA + p * PA + (wm * 64 + q * 32 + lj) * kLdP + ko);
              b[q][p] = *reinterpret_cast<const bf16x8*>(sB + p * PB + (wn * 64 + q * 32 + lj) * kLdP + ko);
            }
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) if (!(dbg & 2)) acc[tm][tn] = mfma_x3(a[tm], b[tn], acc[tm][tn]);
        }
      }
    }

    // ---- epilogue of tile t (both roles walk the same barriers): 64 rows at a time through the staging tile, out as
    // 16-byte stores of whole 512-byte rows by all 512 threads; column sums per 64-row half as in csrc/linear.hip
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    for (int half = 0; half < 2; ++half) {
      const bool stored = m0 + half * 64 < row_end;  // block-uniform
      if (!loader && wm == half) {
        const int64_t live64 = M - m0 - half * 64 - 4 * hi;
        const int live = live64 > 64 ? 64 : (live64 < 0 ? 0 : static_cast<int>(live64));
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          const int nl = wn * 64 + tn * 32 + lj;
          const bool col_ok = n0 + nl < N;
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int rs = tm * 32 + (r & 3) + 8 * (r >> 2);  // row of the sequence, without the lane half's 4 hi
              const float v = acc[tm][tn][r] + bias_v[tn];
              if (stored) stage[(rs + 4 * hi) * kLdT + nl] = v;
              if (STATS && col_ok && rs < live) { s1[tn] += v; s2[tn] += v * v; }
            }
        }
      }
      if (!stored) continue;
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rl = (tid >> 5) + 16 * i, c4 = tid & 31;
        const int64_t m = m0 + half * 64 + rl;
        if (m < row_end && n0 + 4 * c4 < N && !(dbg & 1)) st4(y + m * N + n0 + 4 * c4, ld4(stage + rl * kLdT + 4 * c4));
      }
      __syncthreads();
    }
    if (STATS) {
      if (!loader) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          const int nl = wn * 64 + tn * 32 + lj;
          float a = s1[tn], b = s2[tn];
          a += __shfl_xor(a, 32, 64);
          b += __shfl_xor(b, 32, 64);
          if (hi == 0) { s_stats[(wm * 2 + 0) * kBN + nl] = a; s_stats[(wm * 2 + 1) * kBN + nl] = b; }
        }
      }
      __syncthreads();
      if (tid < kBN && n0 + tid < N) {
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          if (t * 2 + w >= stats_slabs) continue;  // a half without live rows past the caller's slab count: all zeros
          float* p = stats_partial + (t * 2 + w) * 2 * N;
          p[n0 + tid] = s_stats[(w * 2 + 0) * kBN + tid];
          p[N + n0 + tid] = s_stats[(w * 2 + 1) * kBN + tid];
        }
      }
    }
  }
}

template <bool STATS, bool BT>
int launch_sp(const float* x1, const float* w1, int K1, const float* x2, const float* w2, int K2, const float* bias,
              int64_t M, int N, float* y, float* stats_partial, int64_t stats_slabs, int64_t x1_rows, int64_t store_rows,
              hipStream_t st) {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1)
      n = 256;
    return n;
  }();
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_linear_sp<STATS, BT>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
  if (attr != hipSuccess) return hip_fail(attr);
  const int64_t tiles = (M + kBM - 1) / kBM;
  const int gy = (N + kBN - 1) / kBN;
  int64_t gx = cus / gy;  // one block per CU
  if (gx < 1) gx = 1;
  if (gx > tiles) gx = tiles;
  dim3 grid(static_cast<unsigned>(gx), static_cast<unsigned>(gy));
  k_linear_sp<STATS, BT><<<grid, kThreads, kLds, st>>>(x1, w1, K1, x2, w2, K2, bias, M, N, y, stats_partial, x1_rows,
                                                       store_rows, tiles, stats_slabs,
                                                       getenv("STEMGNN_SP_DBG") ? atoi(getenv("STEMGNN_SP_DBG")) : 0);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

}  // namespace

// every tile needs at least one chunk: either the first operand reaches every row, or there is a second operand
bool linear_sp_ok(int64_t M, int64_t N, int64_t K1, int64_t K2, int64_t x1_rows) {
  return M > 0 && N > 0 && N % 4 == 0 && K1 > 0 && K1 % 4 == 0 && K2 >= 0 && K2 % 4 == 0 && (K2 > 0 || x1_rows >= M);
}

int linear_sp_launch(const float* x1, const float* w1, int64_t K1, const float* x2, const float* w2, int64_t K2,
                     const float* bias, int64_t M, int64_t N, float* y, float* stats_partial, int64_t stats_slabs,
                     int64_t x1_rows, int64_t store_rows, bool bt, hipStream_t st) {
  if (!linear_sp_ok(M, N, K1, K2, x1_rows) || (bt && (K2 != 0 || stats_partial))) return STEMGNN_ERR_INVALID_ARG;
  const int k1 = static_cast<int>(K1), k2 = static_cast<int>(K2), n = static_cast<int>(N);
  if (bt) return launch_sp<false, true>(x1, w1, k1, nullptr, nullptr, 0, bias, M, n, y, nullptr, 0, x1_rows, store_rows, st);
  if (stats_partial) return launch_sp<true, false>(x1, w1, k1, x2, w2, k2, bias, M, n, y, stats_partial, stats_slabs, x1_rows, store_rows, st);
  return launch_sp<false, false>(x1, w1, k1, x2, w2, k2, bias, M, n, y, nullptr, 0, x1_rows, store_rows, st);
}

}  // namespace stemgnn
