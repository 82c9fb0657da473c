// Weight-stationary form of the dense product  Y = X W^T + b  (and  dX = dY W)  for K <= 128 contraction columns:
// the form every single-operand product of the D = 128 configurations takes (reference model/encoder.py:62-70 lin_r,
// model/pt_model.py decoders, vq.py project_in / project_out).
//
// The tile kernel of csrc/linear.hip stages BOTH operands through LDS for every 128 x 128 tile: the same 64 KB of
// weights are loaded, cut into bf16 pieces and written to LDS 800 times per launch at C4's batch size, and the pipes
// a tile uses (loads, the cut on the VALU, LDS, matrix cores, stores) mostly take turns.  Here a persistent
// 256-thread block (two per CU) keeps its 128 weight columns in REGISTERS as ready-made matrix-core fragments (wave =
// 64 rows x 32 columns, 96 registers of weight pieces for K = 128, cut once per block), walks over 64-row tiles, and
// only the activation rows go through LDS.  The registers freed from the weight staging hold the NEXT tile's
// activation chunks, so a whole tile of loads per block (64 KB per CU) is in flight while the current one is
// multiplied and stored; the output leaves through an LDS staging tile as 512-byte rows.  One barrier per 32-column
// chunk, five per tile.  The waves of a block run in lockstep (measured on a 512-thread, 128-row variant: the
// phases of a tile added up, whatever the instruction order); the two blocks of a CU drift apart and fill each
// other's gaps.
//
// Arithmetic is the three-way bf16 cut of common.h (fp32-accurate); the accumulation order per output element is
// the tile kernel's, so both kernels return the same bits.
#include "common.h"

#include <cstdlib>

namespace stemgnn {
namespace {

constexpr int kThreads = 256;
constexpr int kTileM = 64, kTileN = 128, kKC = 32;
constexpr int kLdT = kTileN + 4;             // fp32 row stride of the staging tile
constexpr int kPlane = kTileM * kLdP;        // bytes of one bf16 plane of an activation chunk
constexpr int kSlot = 3 * kPlane;            // h, m, l
constexpr size_t kRingBytes = 2 * kSlot;
constexpr size_t kStageBytes = static_cast<size_t>(kTileM) * kLdT * sizeof(float);
constexpr size_t kStatsBytes = 2 * kTileN * sizeof(float);
constexpr size_t kLdsBytes = kRingBytes + kStageBytes + kStatsBytes;

// staging row of float4 slot idx (8 slots per row): rows of every group of eight in the order 0 4 1 5 2 6 3 7 (the two
// rows one ds_write_b64 group covers share no bank; csrc/linear.hip)
__device__ __forceinline__ int stage_row(int idx) {
  const int r = idx >> 3;
  return (r & ~7) | ((r & 1) << 2) | ((r >> 1) & 3);
}

__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

template <int V> struct IntTag { static constexpr int value = V; };

// KS = K / 32 chunks per tile.  BT: the weight is given as [K][N] (backward-data, dX = dY W), else as [N][K].
// N is a multiple of 128 (no column guards anywhere).
//
// Every global load and store of the steady-state loop is UNCONDITIONAL (row indices are clamped instead of
// predicated; a row past M is a copy of row M - 1 that is never stored nor counted).  vmcnt counts loads and stores
// in one in-order queue, and the compiler can only emit a counted wait -- "the chunk fetched a tile ago has landed"
// while fourteen younger loads and stores stay in flight -- when the number of younger operations is the same on
// every path; one predicated load or store between a fetch and its use turns every wait into vmcnt(0), which drains
// the prefetch and the stores once per tile (measured: no faster than the tile kernel).  Tiles that store all their
// rows, tiles that store none (rows past store_rows only feed the statistics) and the one tile on the boundary run
// through three instances of the same body, in that order.
template <int KS, bool STATS, bool BT>
__global__ void __launch_bounds__(kThreads, 2)
k_linear_ws(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, int64_t M, int N,
            float* __restrict__ y, float* __restrict__ stats_partial /*[64-row tiles][2][N]*/, int64_t row_base,
            int64_t stats_block0, int64_t store_rows) {
  constexpr int K = KS * kKC;
  static_assert(KS % 2 == 0, "the last step refills ring slot 0 while slot (KS - 1) & 1 is read");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const ring = smem;
  float* const tile = reinterpret_cast<float*>(smem + kRingBytes);
  float* const s_stats = reinterpret_cast<float*>(smem + kRingBytes + kStageBytes);  // [sum|sumsq][128]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave, hi = lane >> 5, lj = lane & 31;
  const int n0 = blockIdx.y * kTileN;
  const int64_t row_end = store_rows < M ? store_rows : M;
  // without statistics nothing past the stored rows is wanted
  const int64_t tiles = ((STATS ? M : row_end) - row_base + kTileM - 1) / kTileM;

  // ---- the block's weight columns as matrix-core fragments: lane (lj, hi) holds column n0 + 32 wn + lj,
  // contraction steps 16 ks + 8 hi .. + 7 of every 16-wide step ks, cut into the three bf16 planes
  bf16x8 bw[2 * KS][3];
  {
    const int n = n0 + 32 * wn + lj;
#pragma unroll
    for (int ks = 0; ks < 2 * KS; ++ks) {
      const int k = 16 * ks + 8 * hi;
      float4 lo, up;
      if (BT) {
        const float* p = w + static_cast<int64_t>(k) * N + n;
        const int64_t ld = N;
        lo = make_float4(p[0], p[ld], p[2 * ld], p[3 * ld]);
        up = make_float4(p[4 * ld], p[5 * ld], p[6 * ld], p[7 * ld]);
      } else {
        lo = ld4(w + static_cast<int64_t>(n) * K + k);
        up = ld4(w + static_cast<int64_t>(n) * K + k + 4);
      }
      uint4 h, m, l;
      split8(lo, up, h, m, l);
      bw[ks][0] = as_bf16x8(h);
      bw[ks][1] = as_bf16x8(m);
      bw[ks][2] = as_bf16x8(l);
    }
  }
  const float bias_v = bias != nullptr ? bias[n0 + 32 * wn + lj] : 0.f;

  // ---- activation chunks: 64 rows x 32 columns = 512 float4, two per thread; ra[s] = chunk s of a tile
  float4 ra[KS][2];
  const int r0 = stage_row(tid), r1 = stage_row(kThreads + tid), c4s = 4 * (tid & 7);
  auto fetch = [&](int64_t t, int s) {
    const int64_t m0 = row_base + (t < tiles ? t : tiles - 1) * kTileM;  // past the last tile: a harmless re-read
    const int64_t ma = m0 + r0 < M ? m0 + r0 : M - 1, mb = m0 + r1 < M ? m0 + r1 : M - 1;
    ra[s][0] = ld4(x + ma * K + s * kKC + c4s);
    ra[s][1] = ld4(x + mb * K + s * kKC + c4s);
  };
  auto stash = [&](int s, int slot) {
    unsigned char* const base = ring + slot * kSlot;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int off = (i ? r1 : r0) * kLdP + 2 * c4s;
      uint2 h, m, l;
      split3(ra[s][i], h, m, l);
      *reinterpret_cast<uint2*>(base + off) = h;
      *reinterpret_cast<uint2*>(base + kPlane + off) = m;
      *reinterpret_cast<uint2*>(base + 2 * kPlane + off) = l;
    }
  };

  int64_t t = blockIdx.x;
  if (t >= tiles) return;  // whole block
  const int64_t stride = gridDim.x;
#pragma unroll
  for (int s = 0; s < KS; ++s) fetch(t, s);
  stash(0, 0);
  fetch(t + stride, 0);
  __syncthreads();

  // CLS 0: every row of the tile is stored; 1: none is (statistics only); 2: the boundary tile (predicated stores)
  auto do_tile = [&](auto cls) {
    constexpr int CLS = decltype(cls)::value;
    const int64_t m0 = row_base + t * kTileM;
    floatx16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const unsigned char* const slot = ring + (s & 1) * kSlot;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int ko = q * 32 + hi * 16;  // bytes: lane half 0 takes k 0..7, half 1 k 8..15 of the 16-wide step
        bf16x8 a[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
            a[tm][p] = *reinterpret_cast<const bf16x8*>(slot + p * kPlane + (tm * 32 + lj) * kLdP + ko);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) acc[tm] = mfma_x3(a[tm], bw[2 * s + q], acc[tm]);
      }
      // behind the matrix instructions (they run on while the VALU cuts the next chunk): chunk s + 1 of this tile --
      // or, at the last step, chunk 0 of the next -- goes to the other ring slot (last read one barrier ago), and
      // its registers take the same chunk of the tile after
      if (s + 1 < KS) {
        stash(s + 1, (s + 1) & 1);
        fetch(t + stride, s + 1);
      } else {
        stash(0, 0);
        fetch(t + 2 * stride, 0);
      }
      // Issue order of the step (the waves of a block run its steps in lockstep, so nothing else hides one pipe
      // behind another): per 16-wide k step the six fragment reads, then the twelve matrix instructions with the
      // VALU work of the cut in their shadow (a 32x32x16 occupies the matrix pipe for 32 cycles, the two waves of a
      // SIMD alternate on it), the LDS writes and the refill loads last.
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);  // DS read
#pragma unroll
        for (int j = 0; j < 12; ++j) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // VALU
        }
      }
      __builtin_amdgcn_sched_group_barrier(0x200, 6, 0);  // DS write
      __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);  // VMEM read
      __syncthreads();
    }

    // ---- epilogue: the tile (+ bias) through the staging buffer, out as 512-byte rows
    const int nl = 32 * wn + lj;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[tm][r] += bias_v;
        if (CLS != 1) tile[(tm * 32 + 4 * hi + (r & 3) + 8 * (r >> 2)) * kLdT + nl] = acc[tm][r];
      }
    if (STATS) {
      // column sums of the 64 rows (tm, then r: the tile kernel's order per 64-row half), then the two lane halves
      float s1 = 0.f, s2 = 0.f;
      const int64_t live = M - m0 - 4 * hi;  // rows of this lane's sequence below M (whole tile: >= 64)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (tm * 32 + (r & 3) + 8 * (r >> 2) < live) { s1 += acc[tm][r]; s2 += acc[tm][r] * acc[tm][r]; }
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (hi == 0) { s_stats[nl] = s1; s_stats[kTileN + nl] = s2; }
    }
    __syncthreads();
    if (CLS != 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rl = (tid >> 5) + 8 * i, c4 = tid & 31;
        const int64_t m = m0 + rl;
        if (CLS == 0 || m < row_end) st4(y + m * N + n0 + 4 * c4, ld4(tile + rl * kLdT + 4 * c4));
      }
    }
    if (STATS)  // 256 values per tile, one per thread
      stats_partial[(stats_block0 + t) * 2 * N + static_cast<int64_t>(tid >> 7) * N + n0 + (tid & 127)] = s_stats[tid];
  };

  // The first fully stored tile is peeled: at the loop header the compiler merges the wait counters of the entry
  // path and of the back edge to the smaller count, and the prologue (no stores, fewer loads in flight) would cap
  // every wait of the loop at 7 younger operations -- which drains the 8 stores of the previous tile at every tile.
  if (t < tiles && row_base + (t + 1) * kTileM <= row_end) {
    do_tile(IntTag<0>{});
    t += stride;
  }
  for (; t < tiles && row_base + (t + 1) * kTileM <= row_end; t += stride) do_tile(IntTag<0>{});
  for (; t < tiles && row_base + t * kTileM < row_end; t += stride) do_tile(IntTag<2>{});
  if (STATS)
    for (; t < tiles; t += stride) do_tile(IntTag<1>{});
}

template <int KS, bool STATS, bool BT>
int launch_ws(const float* x, const float* w, const float* bias, int64_t M, int N, float* y, float* stats_partial,
              int64_t row_base, int64_t stats_block0, int64_t store_rows, hipStream_t st) {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1)
      n = 256;
    return n;
  }();
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_linear_ws<KS, STATS, BT>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     static_cast<int>(kLdsBytes));
  if (attr != hipSuccess) return STEMGNN_ERR_HIP;
  const int64_t row_end = store_rows < M ? store_rows : M;
  const int64_t tiles = ((STATS ? M : row_end) - row_base + kTileM - 1) / kTileM;  // as in the kernel
  if (tiles <= 0) return STEMGNN_OK;
  const int gy = (N + kTileN - 1) / kTileN;
  int64_t gx = 2 * cus / gy;  // two resident blocks per CU
  if (gx < 1) gx = 1;
  if (gx > tiles) gx = tiles;
  dim3 grid(static_cast<unsigned>(gx), static_cast<unsigned>(gy));
  k_linear_ws<KS, STATS, BT><<<grid, kThreads, kLdsBytes, st>>>(x, w, bias, M, N, y, stats_partial, row_base,
                                                               stats_block0, store_rows);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

}  // namespace

bool linear_ws_ok(int64_t M, int64_t N, int64_t K) { return K == 128 && N % 128 == 0 && M > 0; }

int linear_ws_launch(const float* x, const float* w, const float* bias, int64_t M, int64_t N, int64_t K, float* y,
                     float* stats_partial, int64_t row_base, int64_t stats_block0, int64_t store_rows, bool bt,
                     hipStream_t st) {
  if (!linear_ws_ok(M, N, K)) return STEMGNN_ERR_INVALID_ARG;
  const int n = static_cast<int>(N);
  if (bt) return launch_ws<4, false, true>(x, w, bias, M, n, y, nullptr, row_base, 0, store_rows, st);
  if (stats_partial) return launch_ws<4, true, false>(x, w, bias, M, n, y, stats_partial, row_base, stats_block0, store_rows, st);
  return launch_ws<4, false, false>(x, w, bias, M, n, y, nullptr, row_base, 0, store_rows, st);
}

}  // namespace stemgnn

// ---------------------------------------------------------------------------------------------------------------
// Products over FEW rows (the seed-row and sampled-pair heads of the pretraining step: 1 k - 10 k rows): a 128-row
// tile kernel puts 8 blocks on 256 CUs and each walks its four k-chunks one memory latency after the other (12 us
// for 8 tiles).  Here a WAVE owns a 32 x 32 output tile, reads its operand fragments straight from global memory --
// all 32 loads of a 128-wide k block in flight at once, no LDS, no barrier -- cuts them in registers and multiplies;
// several products (a table passed by value) share one launch.  Same k-step order and arithmetic as the tile kernel:
// the same bits.
// ---------------------------------------------------------------------------------------------------------------
namespace stemgnn {
namespace {

constexpr int kDirectJobs = 4;
struct DirectJob {
  const float* x;     // [M][K]
  const float* w;     // [N][K], or [K][N] when bt
  const float* bias;  // [N] or null
  float* y;           // [M][N]
  int64_t M;
  int N, K, wave_end;  // waves [wave_end of the job before, wave_end): one per 32 x 32 output tile
};
struct DirectTable {
  DirectJob job[kDirectJobs];
  int count;
};

template <bool BT>  // every job of a launch has its weight in the same layout ([K][N] when BT)
__global__ void __launch_bounds__(256) k_linear_direct(const DirectTable tab) {
  const int lane = threadIdx.x & 63, hi = lane >> 5, lj = lane & 31;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  int j = 0, begin = 0;
  while (j + 1 < tab.count && wave >= tab.job[j].wave_end) { begin = tab.job[j].wave_end; ++j; }
  const DirectJob& jb = tab.job[j];
  if (wave >= jb.wave_end) return;  // the last block's spare waves
  const int groups = jb.N / 32, local = wave - begin;
  const int64_t m0 = static_cast<int64_t>(local / groups) * 32;
  const int n0 = (local % groups) * 32;
  const int K = jb.K, N = jb.N;
  const int64_t M = jb.M;
  const int64_t mr = m0 + lj < M ? m0 + lj : M - 1;  // a row past M repeats the last one; it is never stored
  const float* __restrict__ xr = jb.x + mr * K + 8 * hi;
  const float* __restrict__ wr = BT ? jb.w + static_cast<int64_t>(8 * hi) * N + n0 + lj
                                    : jb.w + static_cast<int64_t>(n0 + lj) * K + 8 * hi;
  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int kb = 0; kb < K; kb += 128) {
    float4 xa[8][2], wb[8][2];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int k = kb + 16 * ks < K ? kb + 16 * ks : kb;  // a short last block re-reads its first step (skipped below)
      xa[ks][0] = ld4(xr + k);
      xa[ks][1] = ld4(xr + k + 4);
      if (BT) {
        const float* p = wr + static_cast<int64_t>(k) * N;
        const int64_t ld = N;
        wb[ks][0] = make_float4(p[0], p[ld], p[2 * ld], p[3 * ld]);
        wb[ks][1] = make_float4(p[4 * ld], p[5 * ld], p[6 * ld], p[7 * ld]);
      } else {
        wb[ks][0] = ld4(wr + k);
        wb[ks][1] = ld4(wr + k + 4);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (kb + 16 * ks >= K) break;
      uint4 h, m, l;
      bf16x8 a[3], b[3];
      split8(xa[ks][0], xa[ks][1], h, m, l);
      a[0] = as_bf16x8(h); a[1] = as_bf16x8(m); a[2] = as_bf16x8(l);
      split8(wb[ks][0], wb[ks][1], h, m, l);
      b[0] = as_bf16x8(h); b[1] = as_bf16x8(m); b[2] = as_bf16x8(l);
      acc = mfma_x3(a, b, acc);
    }
  }
  const float bv = jb.bias ? jb.bias[n0 + lj] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t m = m0 + 4 * hi + (r & 3) + 8 * (r >> 2);
    if (m < M) jb.y[m * N + n0 + lj] = acc[r] + bv;
  }
}

}  // namespace

bool linear_direct_ok(int64_t M, int64_t N, int64_t K) {
  return M > 0 && M <= (1 << 16) && N > 0 && N % 32 == 0 && K > 0 && K % 16 == 0 && N <= 4096 && K <= 4096;
}

int DirectBatch::add(const float* x, const float* w, const float* bias, int64_t M, int64_t N, int64_t K, float* y, bool bt,
                     hipStream_t st) {
  if (!linear_direct_ok(M, N, K) || !x || !w || !y) return STEMGNN_ERR_INVALID_ARG;
  if (count == kMax) {
    const int rc = flush(st);
    if (rc != STEMGNN_OK) return rc;
  }
  Job& j = jobs[count++];
  j.x = x; j.w = w; j.bias = bias; j.y = y; j.M = M; j.N = N; j.K = K; j.bt = bt ? 1 : 0;
  return STEMGNN_OK;
}

int DirectBatch::flush(hipStream_t st) {
  if (count == 0) return STEMGNN_OK;
  for (int bt = 0; bt < 2; ++bt) {
    DirectTable tab;
    tab.count = 0;
    int waves = 0;
    for (int i = 0; i < count; ++i) {
      const Job& j = jobs[i];
      if (j.bt != bt) continue;
      DirectJob& t = tab.job[tab.count++];
      t.x = j.x; t.w = j.w; t.bias = j.bias; t.y = j.y; t.M = j.M;
      t.N = static_cast<int>(j.N); t.K = static_cast<int>(j.K);
      waves += static_cast<int>((j.M + 31) / 32) * static_cast<int>(j.N / 32);
      t.wave_end = waves;
    }
    if (tab.count == 0) continue;
    if (bt) k_linear_direct<true><<<static_cast<unsigned>((waves + 3) / 4), 256, 0, st>>>(tab);
    else k_linear_direct<false><<<static_cast<unsigned>((waves + 3) / 4), 256, 0, st>>>(tab);
    STEMGNN_LAUNCH_CHECK();
  }
  count = 0;
  return STEMGNN_OK;
}

}  // namespace stemgnn

// ---------------------------------------------------------------------------------------------------------------
// Code assignment of the vector quantiser (reference model/vq.py:28-29,650-657: l2-normalise the head's rows, cosine
// similarity with the head's codes, arg-max; commitment term vq.py:1007-1009) on the weight-stationary skeleton above:
// the 128 codes of a head are the register-resident operand, the rows of xp stream through LDS, and the epilogue
// takes the arg-max of a 64 x 128 similarity tile instead of storing it.  Same arithmetic as k_vq_assign (csrc/vq.hip)
// -- the same six-product cut in the same k order, the same order of additions for the row norms -- so the same
// indices, norms and commitment sum.  K = Dc = 128 (BASELINE configuration 4).
// ---------------------------------------------------------------------------------------------------------------
namespace stemgnn {
namespace {

constexpr size_t kVqLdsBytes = kRingBytes + kStageBytes + 2 * kTileM * sizeof(float);

__global__ void __launch_bounds__(kThreads, 2)
k_vq_assign_ws(const float* __restrict__ xp, int64_t N, int H, const float* __restrict__ embed,
               const float* __restrict__ esq, float* __restrict__ norm_out, int64_t* __restrict__ ind_out,
               float* __restrict__ sq_partial, unsigned int* counter, double sq_scale, float* __restrict__ sq_out) {
  constexpr int KS = 4, K = 128, Dc = 128;
  constexpr float kNormEps = 1e-12f;  // F.normalize eps
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const ring = smem;
  float* const tile = reinterpret_cast<float*>(smem + kRingBytes);
  float* const s_ssq = reinterpret_cast<float*>(smem + kRingBytes + kStageBytes);  // [64] squared row norms

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave, hi = lane >> 5, lj = lane & 31;
  const int h = blockIdx.y;
  const int64_t HD = static_cast<int64_t>(H) * Dc;
  const float* const xh = xp + static_cast<int64_t>(h) * Dc;
  const float* const emb = embed + static_cast<int64_t>(h) * K * Dc;
  const int64_t tiles = (N + kTileM - 1) / kTileM;

  bf16x8 bw[2 * KS][3];  // the head's codes 32 wn + lj, contraction steps 16 ks + 8 hi .. + 7, as bf16 pieces
#pragma unroll
  for (int ks = 0; ks < 2 * KS; ++ks) {
    const int k = 16 * ks + 8 * hi;
    const float4 lo = ld4(emb + static_cast<int64_t>(32 * wn + lj) * Dc + k);
    const float4 up = ld4(emb + static_cast<int64_t>(32 * wn + lj) * Dc + k + 4);
    uint4 ph, pm, pl;
    split8(lo, up, ph, pm, pl);
    bw[ks][0] = as_bf16x8(ph);
    bw[ks][1] = as_bf16x8(pm);
    bw[ks][2] = as_bf16x8(pl);
  }

  float4 ra[KS][2];
  const int r0 = stage_row(tid), r1 = stage_row(kThreads + tid), c4s = 4 * (tid & 7);
  auto fetch = [&](int64_t t, int s) {
    const int64_t m0 = (t < tiles ? t : tiles - 1) * kTileM;
    const int64_t ma = m0 + r0 < N ? m0 + r0 : N - 1, mb = m0 + r1 < N ? m0 + r1 : N - 1;  // past N: row N - 1 again
    ra[s][0] = ld4(xh + ma * HD + s * kKC + c4s);
    ra[s][1] = ld4(xh + mb * HD + s * kKC + c4s);
  };
  // squared norms of this thread's two rows, summed chunk by chunk as the chunks are cut (k_vq_assign's order); chunk 0
  // of the NEXT tile is cut during this tile's last step, hence the second pair
  float ssq[2] = {0.f, 0.f}, ssq_next[2] = {0.f, 0.f};
  auto stash = [&](int s, int slot, float (&acc2)[2]) {
    unsigned char* const base = ring + slot * kSlot;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int off = (i ? r1 : r0) * kLdP + 2 * c4s;
      const float4 v = ra[s][i];
      acc2[i] += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      uint2 ph, pm, pl;
      split3(v, ph, pm, pl);
      *reinterpret_cast<uint2*>(base + off) = ph;
      *reinterpret_cast<uint2*>(base + kPlane + off) = pm;
      *reinterpret_cast<uint2*>(base + 2 * kPlane + off) = pl;
    }
  };

  int64_t t = blockIdx.x;
  if (t >= tiles) return;  // never: the grid has at most `tiles` blocks per head
  const int64_t stride = gridDim.x;
#pragma unroll
  for (int s = 0; s < KS; ++s) fetch(t, s);
  stash(0, 0, ssq);
  fetch(t + stride, 0);
  __syncthreads();

  float sq_acc = 0.f;  // commitment terms of the rows this thread reports
  for (; t < tiles; t += stride) {
    const int64_t m0 = t * kTileM;
    floatx16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const unsigned char* const slot = ring + (s & 1) * kSlot;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int ko = q * 32 + hi * 16;
        bf16x8 a[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
            a[tm][p] = *reinterpret_cast<const bf16x8*>(slot + p * kPlane + (tm * 32 + lj) * kLdP + ko);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) acc[tm] = mfma_x3(bw[2 * s + q], a[tm], acc[tm]);  // codes x rows, as k_vq_assign
      }
      if (s + 1 < KS) {
        stash(s + 1, (s + 1) & 1, ssq);
        fetch(t + stride, s + 1);
      } else {
        stash(0, 0, ssq_next);
        fetch(t + 2 * stride, 0);
      }
      __syncthreads();
    }
    // ---- epilogue: the arg-max straight from the accumulators.  Operands are swapped relative to the product kernel
    // (the accumulator's rows are codes): lane (lj, hi) of wave wn holds, for data row tm * 32 + lj, the 16 codes
    // 32 wn + 4 hi + (r & 3) + 8 (r >> 2) -- ascending in r.  Strict '>' keeps the lowest index, as torch.argmax does;
    // the two lane halves meet through one shuffle, the four waves through 2 KB of LDS (the similarity tile used to
    // travel through a 33 KB staging tile: 32 scattered stores and eight 16-byte loads per lane and tile).
    float* const s_best = tile;                                           // [4 waves][64 rows]
    int* const s_code = reinterpret_cast<int*>(tile + 4 * kTileM);        // [4 waves][64 rows]
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      float best = -INFINITY;
      int bi = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[tm][r];
        if (v > best) { best = v; bi = 32 * wn + 4 * hi + (r & 3) + 8 * (r >> 2); }
      }
      const float ov = __shfl_xor(best, 32, 64);
      const int oi = __shfl_xor(bi, 32, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      if (hi == 0) {
        s_best[wn * kTileM + tm * 32 + lj] = best;
        s_code[wn * kTileM + tm * 32 + lj] = bi;
      }
    }
    {
      float v0 = ssq[0], v1 = ssq[1];
      v0 += __shfl_xor(v0, 1, 64); v0 += __shfl_xor(v0, 2, 64); v0 += __shfl_xor(v0, 4, 64);
      v1 += __shfl_xor(v1, 1, 64); v1 += __shfl_xor(v1, 2, 64); v1 += __shfl_xor(v1, 4, 64);
      if ((tid & 7) == 0) { s_ssq[r0] = v0; s_ssq[r1] = v1; }
      ssq[0] = ssq_next[0]; ssq[1] = ssq_next[1];
      ssq_next[0] = ssq_next[1] = 0.f;
    }
    __syncthreads();
    if ((tid & 3) == 0) {  // one thread per row (the thread that summed this row's commitment terms before: same sums)
      const int row = tid >> 2;
      float best = s_best[row];
      int bi = s_code[row];
#pragma unroll
      for (int w = 1; w < 4; ++w) {  // waves in ascending code order: strict '>' keeps the lowest index
        const float ov = s_best[w * kTileM + row];
        if (ov > best) { best = ov; bi = s_code[w * kTileM + row]; }
      }
      const float nrm = sqrtf(s_ssq[row]);
      const float inv = 1.0f / fmaxf(nrm, kNormEps), xn2 = nrm * inv;
      const int64_t m = m0 + row < N ? m0 + row : N - 1;  // a row past N is a copy of row N - 1: the same values again
      ind_out[m * H + h] = static_cast<int64_t>(bi);
      norm_out[m * H + h] = nrm;
      if (m0 + row < N) sq_acc += esq[static_cast<int64_t>(h) * K + bi] + xn2 * xn2 - 2.0f * best * inv;
    }
  }
  // ---- the block's commitment sum; the last block to arrive adds all of them in index order (common.h: ticket_last)
  __shared__ double red[kThreads];
  __syncthreads();
  red[tid] = static_cast<double>(sq_acc);
  __syncthreads();
  for (int o = kThreads / 2; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    st_agent(sq_partial + static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x, static_cast<float>(red[0]));
    wait_stores();
  }
  if (!ticket_last(counter)) return;
  double tot = 0.0;
  const int64_t nb = static_cast<int64_t>(gridDim.x) * gridDim.y;
  for (int64_t i = tid; i < nb; i += kThreads) tot += ld_agent(sq_partial + i);
  red[tid] = tot;
  __syncthreads();
  for (int o = kThreads / 2; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) sq_out[0] = static_cast<float>(red[0] * sq_scale);
}

}  // namespace

bool vq_assign_ws_ok(int64_t N, int64_t H, int64_t Dc, int64_t K) {
  return Dc == 128 && K == 128 && H >= 1 && H <= 64 && N >= 16384;
}

int vq_assign_ws_launch(const float* xp, int64_t N, int64_t H, const float* embed, const float* esq, float* norm,
                        int64_t* ind, float* sq_partial, unsigned int* counter, double sq_scale, float* sq_out,
                        hipStream_t st) {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1)
      n = 256;
    return n;
  }();
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_vq_assign_ws),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     static_cast<int>(kVqLdsBytes));
  if (attr != hipSuccess) return STEMGNN_ERR_HIP;
  const int64_t tiles = (N + kTileM - 1) / kTileM;
  int64_t gx = 2 * cus / H;
  if (gx < 1) gx = 1;
  if (gx > tiles) gx = tiles;
  if (gx * H > 1024) gx = 1024 / H;  // the partials the caller's workspace reserves (csrc/vq.hip: kWsPartials)
  dim3 grid(static_cast<unsigned>(gx), static_cast<unsigned>(H));
  k_vq_assign_ws<<<grid, kThreads, kVqLdsBytes, st>>>(xp, N, static_cast<int>(H), embed, esq, norm, ind, sq_partial,
                                                      counter, sq_scale, sq_out);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

}  // namespace stemgnn
