// Weight-stationary dense product in the PAIR format:  Y = X W^T + b  (and  dX = dY W)  for K = 128 contraction columns,
// optionally  + X1 W1^T  on the leading rows -- the products of the D = 128 configurations (reference
// model/encoder.py:62-70 lin_l / lin_r, model/pt_model.py decoders, vq.py project_in / project_out).
//
// csrc/wsgemm.hip runs these products from three exact bf16 pieces per operand: six matrix passes, and a cut that costs
// more VALU instructions than the passes cost matrix cycles (round-4 counters: 3.5 - 9 VALU instructions per MFMA; the
// matrix pipe 36 - 59 % busy; an HBM-bound shape running at a third of the HBM rate).  The pair format of
// csrc/bigtile.hip -- hi = fp16(x s), lo = fp16(x s - hi) with s the power of two that puts the ROW's largest magnitude
// at the top of fp16's range, a b = lo_a hi_b + hi_a lo_b + hi_a hi_b, fp32-accurate -- needs the row's largest magnitude
// before its first element is cut.  With K = 128 a 64-row tile is 32 KB: a block holds the WHOLE tile in registers
// (8 float4 per thread), so the maximum is a register reduction plus three cross-lane steps, and the tile is cut in one
// go into LDS planes [64 rows][128 k] x {hi, lo}.  Per tile:
//     wait for the tile's loads -> row maxima, cut -> barrier -> issue the next tile's loads (a whole tile per block, two
//     blocks per CU: 64 KB in flight per CU while this one is multiplied and stored) -> 48 matrix instructions per wave
//     from LDS fragments against the block's 128 weight columns held in registers as ready-made fragments (cut once
//     per block, scaled per weight row) -> accumulators x (row factor x column factor) + bias through an LDS staging
//     tile -> barrier -> 512-byte rows out, column sums for BatchNorm.
// Two barriers per tile (the bf16-piece kernel: five), three passes instead of six, half the cut arithmetic.
//
// A sampled batch's layer product lin_l(agg) + lin_r(h) carries its first operand on the leading rows only (the nodes
// that receive edges: a ninth of the batch at C4).  Those tiles are at most one per block and come first in a block's
// walk: the block multiplies the aggregate's tile with lin_l's fragments in a prologue (own row factors), leaves the
// scaled result in the staging tile and adds it in that tile's epilogue; lin_l's 64 registers are dead before lin_r's
// are loaded.
//
// In this file, all on the same skeleton (DESIGN.md section 3 "K3-pair" has the measurements):
//   k_linear_wsp<STATS, BT, HEAD, EPI>   the product; EPI = 1 / 2: the quantiser's backward as its epilogue (project_out's
//                                        backward-data product never leaves LDS; 2: also the row maxima of its output)
//   k_vq_assign_wsp                      the code assignment at K = Dc = 128 (arg-max from the accumulators)
//   k_ksp_weight_frags, k_linear_ksp<4>  project_in's backward-data product, 512 -> 128: four segments of contraction,
//                                        the weight's fragments cut once per launch and reloaded per segment from L2
// Nothing here allocates, frees or synchronises; scratch comes from the calling phase's workspace.
#include "common.h"

#include <atomic>
#include <cstdlib>

namespace stemgnn {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kPT = 256;                        // threads: four waves, wave w = columns 32 w .. of the block's 128
constexpr int kPM = 64, kPN = 128, kPK = 128;   // tile rows, block columns, contraction columns
constexpr int kPRow = 2 * kPK + 16;             // bytes per plane row: 68 dwords, conflict-free b128 fragment reads
constexpr int kPPlane = kPM * kPRow;
constexpr int kPLdT = kPN + 4;                  // fp32 row stride of the staging tile
constexpr size_t kPPlanesBytes = 2 * static_cast<size_t>(kPPlane);
constexpr size_t kPStageBytes = static_cast<size_t>(kPM) * kPLdT * sizeof(float);
constexpr size_t kPStatsBytes = 2 * kPN * sizeof(float);
constexpr size_t kPInvBytes = kPM * sizeof(float);
constexpr size_t kPLdsBytes = kPPlanesBytes + kPStageBytes + kPStatsBytes + kPInvBytes;  // 69 888: two blocks per CU

// rows of every group of eight in the order 0 4 1 5 2 6 3 7: the two rows one 16-lane ds_write_b64 group covers are
// four rows apart (16 dwords of bank offset at this row stride: no overlap), csrc/linear.hip
__device__ __forceinline__ int stage_row(int idx) {
  const int r = idx >> 3;
  return (r & ~7) | ((r & 1) << 2) | ((r >> 1) & 3);
}

template <int V> struct IntTag { static constexpr int value = V; };

std::atomic<int64_t> g_wsp_calls{0};

// The block's weight columns as matrix-core fragments: lane (lj, hi) holds column n, contraction steps
// 16 ks + 8 hi .. + 7 of every 16-wide step -- the lane pair (hi = 0, 1) holds the whole weight row, whose largest
// magnitude scales it.  Returns the inverse factor.  BT: the weight is [K][N] (column n read with stride N).
template <bool BT>
__device__ __forceinline__ float weight_fragments(const float* __restrict__ w, int N, int n, int hi, f16x8 (&bw)[8][2],
                                                  float* ssq = nullptr /* the weight row's sum of squares, if wanted */) {
  float4 q[16];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const int k = 16 * ks + 8 * hi;
    if (BT) {
      const float* p = w + static_cast<int64_t>(k) * N + n;
      const int64_t ld = N;
      q[2 * ks] = make_float4(p[0], p[ld], p[2 * ld], p[3 * ld]);
      q[2 * ks + 1] = make_float4(p[4 * ld], p[5 * ld], p[6 * ld], p[7 * ld]);
    } else {
      q[2 * ks] = ld4(w + static_cast<int64_t>(n) * kPK + k);
      q[2 * ks + 1] = ld4(w + static_cast<int64_t>(n) * kPK + k + 4);
    }
  }
  float mx = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) mx = max_abs4(mx, q[i]);
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  if (ssq) {
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sq += q[i].x * q[i].x + q[i].y * q[i].y + q[i].z * q[i].z + q[i].w * q[i].w;
    *ssq = sq + __shfl_xor(sq, 32, 64);
  }
  float sc, inv;
  pair_scale(mx, sc, inv);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    uint2 h0, l0, h1, l1;
    pair_cut4(q[2 * ks], sc, h0, l0);
    pair_cut4(q[2 * ks + 1], sc, h1, l1);
    bw[ks][0] = __builtin_bit_cast(f16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
    bw[ks][1] = __builtin_bit_cast(f16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
  }
  return inv;
}

// STATS: per 64-row tile column sums / sums of squares of y (BatchNorm).  BT: weight given as [K][N].  HEAD: the leading
// x1_rows rows also carry x1 w1^T (at most one such tile per block: the host checks).  N is a multiple of 128.
// Loads and stores of the steady-state loop are unconditional (clamped rows), as in csrc/wsgemm.hip: the compiler can then
// wait for "the tile fetched a tile ago" while the previous tile's stores stay in flight.
// EPI = 1 / 2: the quantiser's backward with project_out's backward-data product inside (csrc/vq.hip: k_vq_assign_bwd_fused
// describes the arithmetic; reference vq.py:937,1041): x = g_out [M, 128], w = W_out [128][H * 128] (BT), a block column =
// one head; the product tile g_q never leaves LDS -- one 32-lane group per row turns it, the row of xp, the row's norm and
// its code into the row of g_xp = y.
struct WspVq {
  const float* xp;       // [M][H * 128]
  const float* norm;     // [M][H]
  const int64_t* ind;    // [M][H]
  const float* embed;    // [H][K][128]
  const float* g_loss;   // [1] or null
  float coef;
  int H, K;
  float* rowmax;         // EPI = 2: [M][H], largest magnitude of every (row, head) stretch of y (k_linear_ksp's row factors)
};

template <bool STATS, bool BT, bool HEAD, int EPI = 0, bool DBG = false>
__global__ void __launch_bounds__(kPT, 2)
k_linear_wsp(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, int64_t M, int N,
             float* __restrict__ y, float* __restrict__ stats_partial /*[64-row tiles][2][N]*/, int64_t stats_block0,
             int64_t store_rows, const float* __restrict__ x1, const float* __restrict__ w1, int64_t x1_rows, int dbg,
             const WspVq vq) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const planes = smem;
  float* const tile = reinterpret_cast<float*>(smem + kPPlanesBytes);
  float* const s_stats = reinterpret_cast<float*>(smem + kPPlanesBytes + kPStageBytes);  // [sum|sumsq][128]
  float* const s_inv = reinterpret_cast<float*>(smem + kPPlanesBytes + kPStageBytes + kPStatsBytes);  // [64] row factors

  constexpr bool VQB = EPI == 1 || EPI == 2;  // the quantiser's backward as the epilogue; EPI == 2: row maxima as well
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  const int hi = lane >> 5, lj = lane & 31;
  const int n0 = blockIdx.y * kPN, nl = 32 * wn + lj;
  const int64_t row_end = store_rows < M ? store_rows : M;
  const int64_t tiles = ((STATS ? M : row_end) + kPM - 1) / kPM;  // without statistics nothing past the stored rows is wanted

  int64_t t = blockIdx.x;
  if (t >= tiles) return;  // whole block
  const int64_t stride = gridDim.x;

  float4 ra[4][2];  // the tile: chunk s (32 columns), rows r0 / r1, columns 32 s + c4s .. + 3
  const int r0 = stage_row(tid), r1 = stage_row(kPT + tid), c4s = 4 * (tid & 7);
  auto fetch = [&](const float* __restrict__ src, int64_t tt, int64_t rows) {
    const int64_t m0 = (tt < tiles ? tt : tiles - 1) * kPM;  // past the last tile: a harmless re-read
    const int64_t ma = m0 + r0 < rows ? m0 + r0 : rows - 1, mb = m0 + r1 < rows ? m0 + r1 : rows - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      ra[s][0] = ld4(src + ma * kPK + 32 * s + c4s);
      ra[s][1] = ld4(src + mb * kPK + 32 * s + c4s);
    }
  };
  // row maxima (8 lanes share a row), factors, the two planes of the whole tile; rows >= live count as zero
  auto cut_tile = [&](int64_t m0, int64_t live) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = i ? r1 : r0;
      if (HEAD && m0 + r >= live) {
#pragma unroll
        for (int s = 0; s < 4; ++s) ra[s][i] = zero4();
      }
      float mx = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) mx = max_abs4(mx, ra[s][i]);
      mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
      float sc, inv;
      pair_scale(mx, sc, inv);
      if ((tid & 7) == 0) s_inv[r] = inv;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        uint2 h, l;
        pair_cut4(ra[s][i], sc, h, l);
        unsigned char* const o = planes + r * kPRow + 2 * (32 * s + c4s);
        *reinterpret_cast<uint2*>(o) = h;
        *reinterpret_cast<uint2*>(o + kPPlane) = l;
      }
    }
  };
  // fragments of k step ks + 1 are requested before the six matrix instructions of step ks are issued (an LDS round trip
  // is about as long as those take); nothing crosses a step's end, so the requests stay one step ahead
  auto multiply = [&](const f16x8 (&bf)[8][2], floatx16 (&acc)[2]) {
    f16x8 a[2][2][2];  // [step parity][tm][plane]
    auto request = [&](int ks) {
      const int ko = 32 * ks + 16 * hi;  // bytes: lane half 0 takes k 0..7, half 1 k 8..15 of the 16-wide step
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
          a[ks & 1][tm][p] = *reinterpret_cast<const f16x8*>(planes + p * kPPlane + (tm * 32 + lj) * kPRow + ko);
    };
    request(0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks + 1 < 8) request(ks + 1);
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {  // small terms first
        acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks & 1][tm][1], bf[ks][0], acc[tm], 0, 0, 0);
        acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks & 1][tm][0], bf[ks][1], acc[tm], 0, 0, 0);
        acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks & 1][tm][0], bf[ks][0], acc[tm], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // this lane's 32 row factors: rows tm * 32 + 4 hi + (r & 3) + 8 (r >> 2), four consecutive rows per read
  auto row_factors = [&](float (&f)[2][16]) {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 v = ld4(s_inv + tm * 32 + 4 * hi + 8 * j);
        f[tm][4 * j] = v.x; f[tm][4 * j + 1] = v.y; f[tm][4 * j + 2] = v.z; f[tm][4 * j + 3] = v.w;
      }
  };

  // ---- the leading tile of a sampled batch: x1 w1^T into the staging tile (each lane keeps its own 32 entries there)
  bool head_pending = false;
  if (HEAD) {
    const int64_t m0 = t * kPM;
    if (m0 < x1_rows) {  // block-uniform
      fetch(x1, t, x1_rows);  // the buffer may end at x1_rows: clamped reads, zeroed in the cut; requested BEFORE the
      __builtin_amdgcn_sched_barrier(0);  // weight's loads: one memory latency for both
      f16x8 bl[8][2];
      const float inv_l = weight_fragments<false>(w1, N, n0 + nl, hi, bl);
      cut_tile(m0, x1_rows);
      __syncthreads();
      floatx16 acc[2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
      multiply(bl, acc);
      float rf[2][16];
      row_factors(rf);
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          tile[(tm * 32 + 4 * hi + (r & 3) + 8 * (r >> 2)) * kPLdT + nl] = acc[tm][r] * (rf[tm][r] * inv_l);
      head_pending = true;
      __syncthreads();  // the planes and the row factors are free again
    }
  }

  fetch(x, t, M);  // the first tile's rows travel while the weight is loaded and cut
  __builtin_amdgcn_sched_barrier(0);
  f16x8 bw[8][2];
  const float inv_w = weight_fragments<BT>(w, N, n0 + nl, hi, bw);
  const float bias_v = bias != nullptr ? bias[n0 + nl] : 0.f;
  const float vq_s = (VQB && vq.g_loss) ? vq.g_loss[0] * vq.coef : 0.f;

  // CLS 0: every row of the tile is stored; 1: none is (statistics only); 2: the boundary tile (predicated stores)
  auto do_tile = [&](auto cls) {
    constexpr int CLS = decltype(cls)::value;
    const int64_t m0 = t * kPM;
    if (!DBG || !(dbg & 2)) cut_tile(m0, M + kPM);  // (rows past M are copies of row M - 1: never stored nor counted)
    __syncthreads();
    if (!DBG || !(dbg & 8)) fetch(x, t + stride, M);  // a whole iteration ahead of its use: pinned here (the scheduler would sink the loads
    __builtin_amdgcn_sched_barrier(0);  // behind the matrix work to save their registers)
    // EPI = 1: what the row-wise part needs from memory is requested before the matrix work -- codes, norms and the xp
    // rows of this thread's eight rows (lane group g serves rows g, g + 8, ...) --, the code rows right after it
    const int l32 = tid & 31, grp = tid >> 5;
    int code[VQB ? 8 : 1];
    float nrm[VQB ? 8 : 1];
    float4 xv[VQB ? 8 : 1], qv[VQB ? 8 : 1];
    // (addresses: a block-uniform base per row step + one 32-bit lane offset that never changes -- eight 64-bit lane
    // addresses per array cost more registers than the kernel has; only the boundary tile clamps rows per lane)
    const int xoff = grp * N + n0 + 4 * l32, ioff = grp * vq.H + static_cast<int>(blockIdx.y);
    auto row_of = [&](int i) -> int64_t {  // first row of step i for the uniform base; lanes add grp through xoff / ioff
      return m0 + 8 * i;
    };
    auto lane_row_ok = [&](int i) { return CLS == 0 || m0 + grp + 8 * i < M; };
    if (VQB) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool ok = lane_row_ok(i);
        const int64_t base = ok ? row_of(i) : M - 1 - grp;  // a row past M reads row M - 1 again (never stored)
        code[i] = static_cast<int>((vq.ind + base * vq.H)[ioff]);
        nrm[i] = (vq.norm + base * vq.H)[ioff];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    floatx16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    if (!DBG || !(dbg & 1)) multiply(bw, acc);
    if (DBG && (dbg & 4)) {  // no epilogue: keep the accumulators alive
      if (acc[0][0] + acc[1][5] == 123.456f) y[tid] = acc[0][0];
      return;
    }
    if (VQB) {
      const float* erow = vq.embed + static_cast<int64_t>(blockIdx.y) * vq.K * kPK + 4 * l32;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int c = code[i];
        if (c < 0 || c >= vq.K) c = 0;
        qv[i] = ld4(erow + static_cast<int64_t>(c) * kPK);
      }
    }

    // ---- epilogue: factors, bias (+ the leading operand's part), through the staging tile, out as 512-byte rows
    float rf[2][16];
    row_factors(rf);
    if (HEAD && head_pending) {  // block-uniform, a block's first tile only: the leading operand's part joins the
      // accumulators in THEIR scale (the factors are powers of two: the division and the later product are exact)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          acc[tm][r] += tile[(tm * 32 + 4 * hi + (r & 3) + 8 * (r >> 2)) * kPLdT + nl] / (rf[tm][r] * inv_w);
      head_pending = false;
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rs = tm * 32 + 4 * hi + (r & 3) + 8 * (r >> 2);
        const float v = acc[tm][r] * (rf[tm][r] * inv_w) + bias_v;
        if (CLS != 1) tile[rs * kPLdT + nl] = v;
        acc[tm][r] = v;
      }
    if (VQB) {  // the xp rows: requested once the accumulators have left for the staging tile (registers)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t base = lane_row_ok(i) ? row_of(i) : M - 1 - grp;
        xv[i] = ld4(vq.xp + base * N + xoff);
      }
    }
    if (STATS) {
      float s1 = 0.f, s2 = 0.f;
      if (m0 + kPM <= M) {  // block-uniform: every tile but the last
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int r = 0; r < 16; ++r) { s1 += acc[tm][r]; s2 += acc[tm][r] * acc[tm][r]; }
      } else {
        const int live = static_cast<int>(M - m0) - 4 * hi;  // rows of this lane's sequence below M
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (tm * 32 + (r & 3) + 8 * (r >> 2) < live) { s1 += acc[tm][r]; s2 += acc[tm][r] * acc[tm][r]; }
      }
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (hi == 0) { s_stats[nl] = s1; s_stats[kPN + nl] = s2; }
    }
    __syncthreads();
    if (VQB) {
      constexpr float kNormEps = 1e-12f;  // F.normalize eps (csrc/vq.hip)
      float rmx[EPI == 2 ? 8 : 1];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rl = grp + 8 * i;
        const float nr = nrm[i];
        const bool clamped = nr < kNormEps;
        const float inv = 1.0f / fmaxf(nr, kNormEps);
        const float4 gq = ld4(tile + rl * kPLdT + 4 * l32);
        const float4 xq = xv[i], q = qv[i];
        const float4 n = make_float4(xq.x * inv, xq.y * inv, xq.z * inv, xq.w * inv);
        const float4 gx = make_float4(gq.x + vq_s * (n.x - q.x), gq.y + vq_s * (n.y - q.y), gq.z + vq_s * (n.z - q.z),
                                      gq.w + vq_s * (n.w - q.w));
        float dot = gx.x * n.x + gx.y * n.y + gx.z * n.z + gx.w * n.w;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 32);
        if (clamped) dot = 0.f;  // the x / eps branch of F.normalize: plain scaling
        const float4 o4 = make_float4((gx.x - n.x * dot) * inv, (gx.y - n.y * dot) * inv, (gx.z - n.z * dot) * inv,
                                      (gx.w - n.w * dot) * inv);
        if (lane_row_ok(i)) st4(y + row_of(i) * N + xoff, o4);
        if (EPI == 2) rmx[i] = max_abs4(0.f, o4);
      }
      if (EPI == 2) {
        // the eight rows' maxima over the group's 32 lanes as ONE butterfly: a lane keeps the half of the rows its own
        // bit selects and hands the other half over -- 4 + 2 + 1 + 1 + 1 cross-lane steps instead of 8 x 5 -- and lane
        // 4 i of the group ends up with row i's maximum: one store instruction per thread
        const bool b16 = (l32 & 16) != 0, b8 = (l32 & 8) != 0, b4 = (l32 & 4) != 0;
        float m4[4], m2[2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float keep = b16 ? rmx[j + 4] : rmx[j], send = b16 ? rmx[j] : rmx[j + 4];
          m4[j] = fmaxf(keep, __shfl_xor(send, 16, 32));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float keep = b8 ? m4[j + 2] : m4[j], send = b8 ? m4[j] : m4[j + 2];
          m2[j] = fmaxf(keep, __shfl_xor(send, 8, 32));
        }
        float m1 = fmaxf(b4 ? m2[1] : m2[0], __shfl_xor(b4 ? m2[0] : m2[1], 4, 32));
        m1 = fmaxf(m1, __shfl_xor(m1, 2, 32));
        m1 = fmaxf(m1, __shfl_xor(m1, 1, 32));
        const int ri = (b4 ? 1 : 0) | (b8 ? 2 : 0) | (b16 ? 4 : 0);  // the row step this lane reports
        if ((l32 & 3) == 0 && (CLS == 0 || m0 + grp + 8 * ri < M)) (vq.rowmax + (m0 + 8 * ri) * vq.H)[ioff] = m1;
      }
    } else if (CLS != 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rl = (tid >> 5) + 8 * i, c4 = tid & 31;
        const int64_t m = m0 + rl;
        if (CLS == 0 || m < row_end) st4(y + m * N + n0 + 4 * c4, ld4(tile + rl * kPLdT + 4 * c4));
      }
    }
    if (STATS)  // 256 values per tile, one per thread
      stats_partial[(stats_block0 + t) * 2 * N + static_cast<int64_t>(tid >> 7) * N + n0 + (tid & 127)] = s_stats[tid];
  };

  // Every loop is entered through a peeled first tile of its class only: at a loop header the compiler merges the wait
  // counts of all entering paths to the smallest, and a path with fewer operations in flight (the prologue: loads, no
  // stores) would make every iteration wait for the previous tile's stores before it cuts the next one.
  auto full = [&]() { return t < tiles && (t + 1) * kPM <= row_end; };
  if (full()) {
    do_tile(IntTag<0>{});
    t += stride;
    while (full()) {
      do_tile(IntTag<0>{});
      t += stride;
    }
  }
  if (t < tiles && t * kPM < row_end) {  // the boundary tile
    do_tile(IntTag<2>{});
    t += stride;
  }
  if (STATS && t < tiles) {
    do_tile(IntTag<1>{});
    t += stride;
    while (t < tiles) {
      do_tile(IntTag<1>{});
      t += stride;
    }
  }
}

template <bool STATS, bool BT, bool HEAD, int EPI = 0, bool DBG = false>
int launch_wsp(const float* x, const float* w, const float* bias, int64_t M, int N, float* y, float* stats_partial,
               int64_t stats_block0, int64_t store_rows, const float* x1, const float* w1, int64_t x1_rows, int64_t gx,
               hipStream_t st, const WspVq& vq = WspVq{}, int dbg = 0) {
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_linear_wsp<STATS, BT, HEAD, EPI, DBG>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     static_cast<int>(kPLdsBytes));
  if (attr != hipSuccess) return STEMGNN_ERR_HIP;
  dim3 grid(static_cast<unsigned>(gx), static_cast<unsigned>(N / kPN));
  k_linear_wsp<STATS, BT, HEAD, EPI, DBG><<<grid, kPT, kPLdsBytes, st>>>(x, w, bias, M, N, y, stats_partial, stats_block0,
                                                                        store_rows, x1, w1, x1_rows, dbg, vq);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

inline int64_t wsp_blocks(int64_t N) {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1)
      n = 256;
    return n;
  }();
  const int64_t gx = 2 * cus / (N / kPN);  // two resident blocks per CU
  return gx < 1 ? 1 : gx;
}

}  // namespace

// x1_rows > 0: the leading-operand form -- its tiles must be the first of their blocks' walks
bool linear_wsp_ok(int64_t M, int64_t N, int64_t K, int64_t x1_rows) {
  if (K != kPK || N % kPN != 0 || N <= 0 || M <= 0) return false;
  if (x1_rows <= 0) return true;
  const int64_t tiles = (M + kPM - 1) / kPM;
  const int64_t gx = wsp_blocks(N) < tiles ? wsp_blocks(N) : tiles;
  return (x1_rows + kPM - 1) / kPM <= gx;
}

int linear_wsp_launch(const float* x, const float* w, const float* bias, int64_t M, int64_t N, float* y,
                      float* stats_partial, int64_t stats_block0, int64_t store_rows, bool bt, const float* x1,
                      const float* w1, int64_t x1_rows, hipStream_t st) {
  const bool head = x1_rows > 0 && x1 != nullptr && w1 != nullptr;
  if (!linear_wsp_ok(M, N, kPK, head ? x1_rows : 0) || (bt && (head || stats_partial))) return STEMGNN_ERR_INVALID_ARG;
  const int n = static_cast<int>(N);
  const int64_t row_end = store_rows < M ? store_rows : M;
  const int64_t tiles = ((stats_partial ? M : row_end) + kPM - 1) / kPM;  // as in the kernel
  if (tiles <= 0) return STEMGNN_OK;
  g_wsp_calls.fetch_add(1, std::memory_order_relaxed);
  int64_t gx = wsp_blocks(N);
  if (gx > tiles) gx = tiles;
  {  // head tiles that exist in this launch (rows past store_rows are not walked without statistics): one per block
    const int64_t head_tiles = head ? (x1_rows + kPM - 1) / kPM : 0;
    if ((head_tiles < tiles ? head_tiles : tiles) > gx) return STEMGNN_ERR_INVALID_ARG;
  }
  if (bt) {
    static const int dbg = std::getenv("STEMGNN_WSP_DBG") ? std::atoi(std::getenv("STEMGNN_WSP_DBG")) : 0;
    if (dbg)
      return launch_wsp<false, true, false, 0, true>(x, w, bias, M, n, y, nullptr, 0, store_rows, nullptr, nullptr, 0, gx, st,
                                                     WspVq{}, dbg);
    return launch_wsp<false, true, false>(x, w, bias, M, n, y, nullptr, 0, store_rows, nullptr, nullptr, 0, gx, st);
  }
  if (head) {
    if (stats_partial)
      return launch_wsp<true, false, true>(x, w, bias, M, n, y, stats_partial, stats_block0, store_rows, x1, w1, x1_rows, gx, st);
    return launch_wsp<false, false, true>(x, w, bias, M, n, y, nullptr, 0, store_rows, x1, w1, x1_rows, gx, st);
  }
  if (stats_partial)
    return launch_wsp<true, false, false>(x, w, bias, M, n, y, stats_partial, stats_block0, store_rows, nullptr, nullptr, 0, gx, st);
  return launch_wsp<false, false, false>(x, w, bias, M, n, y, nullptr, 0, store_rows, nullptr, nullptr, 0, gx, st);
}

namespace {

// ---------------------------------------------------------------------------------------------------------------
// Code assignment of the vector quantiser (reference model/vq.py:28-29,650-657: l2-normalise the head's rows, cosine
// similarity with the head's codes, arg-max; commitment term vq.py:1007-1009) at K = Dc = 128 on the same skeleton:
// the head's 128 codes are the register-resident operand (each code row scaled by its own power of two), the rows of
// xp stream through the planes, and the arg-max is taken from the accumulators -- rows of the accumulator are CODES
// (operands swapped, as in k_vq_assign_ws), so a lane compares 16 codes of one data row in registers: the candidates
// are accumulator x code factor (the row's factor is positive and common to its candidates), lowest index among
// equals; the similarity that leaves is candidate x row factor, an fp32-accurate dot product (the reference forces
// fp32 here, vq.py:623,634).  Row norms from the fp32 values as they are cut.  Outputs as k_vq_assign's lean form.
// ---------------------------------------------------------------------------------------------------------------
constexpr size_t kAqBest = 4 * kPM * sizeof(float), kAqCode = 4 * kPM * sizeof(int);
constexpr size_t kAqLdsBytes = kPPlanesBytes + kAqBest + kAqCode + 2 * 2 * kPM * sizeof(float) + 2 * kPN * sizeof(float);

__global__ void __launch_bounds__(kPT, 2)
k_vq_assign_wsp(const float* __restrict__ xp, int64_t N, int H, const float* __restrict__ embed,
                float* __restrict__ norm_out, int64_t* __restrict__ ind_out,
                float* __restrict__ sq_partial, unsigned int* counter, double sq_scale, float* __restrict__ sq_out) {
  constexpr int K = 128;
  constexpr float kNormEps = 1e-12f;  // F.normalize eps
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const planes = smem;
  float* const s_best = reinterpret_cast<float*>(smem + kPPlanesBytes);                 // [4 waves][64 rows]
  int* const s_code = reinterpret_cast<int*>(smem + kPPlanesBytes + kAqBest);           // [4 waves][64 rows]
  float* const s_row = reinterpret_cast<float*>(smem + kPPlanesBytes + kAqBest + kAqCode);  // [parity][inv | ssq][64]
  float* const s_cf = s_row + 2 * 2 * kPM;                                              // [128] code factors
  float* const s_esq = s_cf + kPN;                                                      // [128] squared code norms

  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  const int hi = lane >> 5, lj = lane & 31;
  const int h = blockIdx.y;
  const int64_t HD = static_cast<int64_t>(H) * kPK;
  const float* const xh = xp + static_cast<int64_t>(h) * kPK;
  const float* const emb = embed + static_cast<int64_t>(h) * K * kPK;
  const int64_t tiles = (N + kPM - 1) / kPM;
  int64_t t = blockIdx.x;
  if (t >= tiles) return;  // never: the grid has at most `tiles` blocks per head
  const int64_t stride = gridDim.x;

  float4 ra[4][2];
  const int r0 = stage_row(tid), r1 = stage_row(kPT + tid), c4s = 4 * (tid & 7);
  auto fetch = [&](int64_t tt) {
    const int64_t m0 = (tt < tiles ? tt : tiles - 1) * kPM;
    const int64_t ma = m0 + r0 < N ? m0 + r0 : N - 1, mb = m0 + r1 < N ? m0 + r1 : N - 1;  // past N: row N - 1 again
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      ra[s][0] = ld4(xh + ma * HD + 32 * s + c4s);
      ra[s][1] = ld4(xh + mb * HD + 32 * s + c4s);
    }
  };
  fetch(t);
  __builtin_amdgcn_sched_barrier(0);
  f16x8 bw[8][2];
  {
    float code_sq;  // |e|^2 of this lane pair's code: the block's own copy of what k_code_sqnorm computes (one launch less)
    const float inv_c = weight_fragments<false>(emb, K, 32 * wn + lj, hi, bw, &code_sq);
    if (hi == 0) { s_cf[32 * wn + lj] = inv_c; s_esq[32 * wn + lj] = code_sq; }
  }
  __syncthreads();
  float cf[16];  // factors of this lane's 16 codes 32 wn + 4 hi + (r & 3) + 8 (r >> 2)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 v = ld4(s_cf + 32 * wn + 4 * hi + 8 * j);
    cf[4 * j] = v.x; cf[4 * j + 1] = v.y; cf[4 * j + 2] = v.z; cf[4 * j + 3] = v.w;
  }

  float sq_acc = 0.f;  // commitment terms of the rows this thread reports
  int par = 0;
  // (entered through a peeled first tile, csrc/wspair.hip k_linear_wsp: counted waits in the loop)
  auto do_tile = [&]() {
    const int64_t m0 = t * kPM;
    float* const s_inv = s_row + par * 2 * kPM;
    float* const s_ssq = s_inv + kPM;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = i ? r1 : r0;
      float mx = 0.f, sq = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float4 v = ra[s][i];
        mx = max_abs4(mx, v);
        sq += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      }
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        sq += __shfl_xor(sq, o, 64);
      }
      float sc, inv;
      pair_scale(mx, sc, inv);
      if ((tid & 7) == 0) { s_inv[r] = inv; s_ssq[r] = sq; }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        uint2 ph, pl;
        pair_cut4(ra[s][i], sc, ph, pl);
        unsigned char* const o = planes + r * kPRow + 2 * (32 * s + c4s);
        *reinterpret_cast<uint2*>(o) = ph;
        *reinterpret_cast<uint2*>(o + kPPlane) = pl;
      }
    }
    __syncthreads();
    fetch(t + stride);
    __builtin_amdgcn_sched_barrier(0);
    floatx16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    {
      f16x8 a[2][2][2];  // [step parity][tm][plane]
      auto request = [&](int ks) {
        const int ko = 32 * ks + 16 * hi;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
            a[ks & 1][tm][p] = *reinterpret_cast<const f16x8*>(planes + p * kPPlane + (tm * 32 + lj) * kPRow + ko);
      };
      request(0);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks + 1 < 8) request(ks + 1);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {  // codes x rows; small terms first
          acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw[ks][1], a[ks & 1][tm][0], acc[tm], 0, 0, 0);
          acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw[ks][0], a[ks & 1][tm][1], acc[tm], 0, 0, 0);
          acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw[ks][0], a[ks & 1][tm][0], acc[tm], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- arg-max: 16 codes per lane in ascending order (strict '>' keeps the lowest), the two lane halves through one
    // shuffle, the four waves through LDS
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      float best = -INFINITY;
      int bi = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[tm][r] * cf[r];
        if (v > best) { best = v; bi = 32 * wn + 4 * hi + (r & 3) + 8 * (r >> 2); }
      }
      const float ov = __shfl_xor(best, 32, 64);
      const int oi = __shfl_xor(bi, 32, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      if (hi == 0) {
        s_best[wn * kPM + tm * 32 + lj] = best;
        s_code[wn * kPM + tm * 32 + lj] = bi;
      }
    }
    __syncthreads();
    if ((tid & 3) == 0) {  // one thread per row
      const int row = tid >> 2;
      float best = s_best[row];
      int bi = s_code[row];
#pragma unroll
      for (int w = 1; w < 4; ++w) {  // waves in ascending code order: strict '>' keeps the lowest index
        const float ov = s_best[w * kPM + row];
        if (ov > best) { best = ov; bi = s_code[w * kPM + row]; }
      }
      best *= s_inv[row];  // the row's factor: now the dot product of the row with its code
      const float nrm = sqrtf(s_ssq[row]);
      const float inv = 1.0f / fmaxf(nrm, kNormEps), xn2 = nrm * inv;
      const int64_t m = m0 + row < N ? m0 + row : N - 1;  // a row past N is a copy of row N - 1: the same values again
      ind_out[m * H + h] = static_cast<int64_t>(bi);
      norm_out[m * H + h] = nrm;
      if (m0 + row < N) sq_acc += s_esq[bi] + xn2 * xn2 - 2.0f * best * inv;
    }
    par ^= 1;  // the next tile's cut writes the other set of row factors while slow waves still read this one
  };
  do_tile();
  t += stride;
  while (t < tiles) {
    do_tile();
    t += stride;
  }

  // ---- the block's commitment sum; the last block to arrive adds all of them in index order (common.h: ticket_last)
  __shared__ double red[kPT];
  __syncthreads();
  red[tid] = static_cast<double>(sq_acc);
  __syncthreads();
  for (int o = kPT / 2; o > 0; o >>= 1) {
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
  for (int64_t i = tid; i < nb; i += kPT) tot += ld_agent(sq_partial + i);
  red[tid] = tot;
  __syncthreads();
  for (int o = kPT / 2; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) sq_out[0] = static_cast<float>(red[0] * sq_scale);
}

}  // namespace

int vq_assign_wsp_launch(const float* xp, int64_t N, int64_t H, const float* embed, const float* esq, float* norm,
                         int64_t* ind, float* sq_partial, unsigned int* counter, double sq_scale, float* sq_out,
                         hipStream_t st) {
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_vq_assign_wsp),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     static_cast<int>(kAqLdsBytes));
  if (attr != hipSuccess) return STEMGNN_ERR_HIP;
  const int64_t tiles = (N + kPM - 1) / kPM;
  int64_t gx = wsp_blocks(H * kPN);
  if (gx > tiles) gx = tiles;
  if (gx * H > 1024) gx = 1024 / H;  // the partials the caller's workspace reserves (csrc/vq.hip: kWsPartials)
  if (gx < 1) gx = 1;
  g_wsp_calls.fetch_add(1, std::memory_order_relaxed);
  dim3 grid(static_cast<unsigned>(gx), static_cast<unsigned>(H));
  (void)esq;  // the kernel takes the codes' squared norms from the rows it cuts
  k_vq_assign_wsp<<<grid, kPT, kAqLdsBytes, st>>>(xp, N, static_cast<int>(H), embed, norm, ind, sq_partial, counter, sq_scale,
                                                  sq_out);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

namespace {

// ---------------------------------------------------------------------------------------------------------------
// y [M, 128] = x [M, S * 128] w  with w given as [S * 128][128] (the backward-data product of a Linear with 128 inputs and
// S * 128 outputs: project_in's, dz = g_xp W_in) in the pair format.  The contraction is S segments of 128 columns; the
// weight does not fit the registers (S x 64 per lane), so its fragments are cut ONCE per launch by k_ksp_weight_frags into a
// buffer laid out exactly as the registers want them (one coalesced 1 KB load per fragment and wave, L2-resident: 64 KB
// per segment) and a block reloads the 64 registers per segment while it cuts the segment's rows.  One power-of-two
// factor per ROW over all S segments -- its largest magnitude comes from `rowmax` [M][S], which the producer of x writes
// (EPI = 1 above) -- and one per weight column over the whole contraction: a single accumulator runs through the segments.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kKspFragU4 = 4 * 16 * 64;  // uint4 per segment: 4 waves x (8 k steps x 2 planes) x 64 lanes

// grid (segments, 4): block (seg, wn) cuts the fragments of wave wn's 32 weight columns for segment seg
__global__ void __launch_bounds__(kPT)
k_ksp_weight_frags(const float* __restrict__ w /*[K][128]*/, int K, uint4* __restrict__ frag, float* __restrict__ inv_w) {
  __shared__ float s_part[32][32];
  __shared__ float s_scale[32];
  const int tid = threadIdx.x;
  const int seg = blockIdx.x, wn = blockIdx.y;
  {  // largest magnitude of the block's 32 weight columns over the WHOLE contraction: 8 lanes per row, 32 rows per pass,
    // sixteen rows in flight per thread
    const int c4 = 32 * wn + 4 * (tid & 7), rg = tid >> 3;
    float4 mx = zero4();
    for (int k = rg; k < K; k += 32 * 16) {
      float4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int kk = k + 32 * u < K ? k + 32 * u : rg;  // a short last batch re-reads the thread's first row
        v[u] = ld4(w + static_cast<int64_t>(kk) * kPN + c4);
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        mx = make_float4(fmaxf(mx.x, fabsf(v[u].x)), fmaxf(mx.y, fabsf(v[u].y)), fmaxf(mx.z, fabsf(v[u].z)),
                         fmaxf(mx.w, fabsf(v[u].w)));
    }
    st4(&s_part[rg][4 * (tid & 7)], mx);
    __syncthreads();
    if (tid < 32) {
      float m = 0.f;
#pragma unroll
      for (int g = 0; g < 32; ++g) m = fmaxf(m, s_part[g][tid]);
      float sc, inv;
      pair_scale(m, sc, inv);
      s_scale[tid] = sc;
      if (seg == 0) inv_w[32 * wn + tid] = inv;
    }
    __syncthreads();
  }
  if (tid >= 64) return;  // one wave cuts the 32 columns' fragments of this segment
  const int lane = tid, hi = lane >> 5, lj = lane & 31;
  const int n = 32 * wn + lj;
  const float sc = s_scale[lj];
  const float* p = w + (static_cast<int64_t>(seg) * kPK + 8 * hi) * kPN + n;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const float* q = p + static_cast<int64_t>(16 * ks) * kPN;
    uint2 h0, l0, h1, l1;
    pair_cut4(make_float4(q[0], q[kPN], q[2 * kPN], q[3 * kPN]), sc, h0, l0);
    pair_cut4(make_float4(q[4 * kPN], q[5 * kPN], q[6 * kPN], q[7 * kPN]), sc, h1, l1);
    uint4* o = frag + static_cast<int64_t>(seg) * kKspFragU4 + (wn * 16 + 2 * ks) * 64 + lane;
    o[0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
    o[64] = make_uint4(l0.x, l0.y, l1.x, l1.y);
  }
}

template <int S>
__global__ void __launch_bounds__(kPT, 2)
k_linear_ksp(const float* __restrict__ x /*[M][S * 128]*/, const float* __restrict__ rowmax /*[M][S]*/,
             const uint4* __restrict__ frag, const float* __restrict__ inv_w_all, int64_t M, float* __restrict__ y) {
  // two sets of planes: segment s + 1 is cut into the other set while slow waves still multiply segment s -- ONE barrier
  // per segment.  The staging tile of the epilogue lies over the set the tile's last segment did NOT use, and the next
  // tile starts with the other set (`par`), so nothing is overwritten while a wave still reads it.
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const s_inv = reinterpret_cast<float*>(smem + 2 * kPPlanesBytes);  // [64] row factors
  int par = 0;
  constexpr int64_t ldx = static_cast<int64_t>(S) * kPK;

  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  const int hi = lane >> 5, lj = lane & 31, nl = 32 * wn + lj;
  const int64_t tiles = (M + kPM - 1) / kPM;
  int64_t t = blockIdx.x;
  if (t >= tiles) return;
  const int64_t stride = gridDim.x;
  const float inv_w = inv_w_all[nl];

  float4 ra[4][2];
  const int r0 = stage_row(tid), r1 = stage_row(kPT + tid), c4s = 4 * (tid & 7);
  // segment `seg` of tile `tt` (seg == S: segment 0 of the block's next tile)
  auto fetch = [&](int64_t tt, int seg) {
    if (seg == S) { tt += stride; seg = 0; }
    const int64_t m0 = (tt < tiles ? tt : tiles - 1) * kPM;
    const int64_t ma = m0 + r0 < M ? m0 + r0 : M - 1, mb = m0 + r1 < M ? m0 + r1 : M - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      ra[s][0] = ld4(x + ma * ldx + seg * kPK + 32 * s + c4s);
      ra[s][1] = ld4(x + mb * ldx + seg * kPK + 32 * s + c4s);
    }
  };
  fetch(t, 0);

  auto do_tile = [&](auto cls) {
    constexpr int CLS = decltype(cls)::value;
    const int64_t m0 = t * kPM;
    // the rows' factors: largest magnitude over the S stretches
    float sc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = i ? r1 : r0;
      const int64_t m = m0 + r < M ? m0 + r : M - 1;
      float mx = 0.f;
#pragma unroll
      for (int q = 0; q < S; ++q) mx = fmaxf(mx, rowmax[m * S + q]);
      float inv;
      pair_scale(mx, sc[i], inv);
      if ((tid & 7) == 0) s_inv[r] = inv;
    }
    float* const tile = reinterpret_cast<float*>(smem + par * kPPlanesBytes);
    floatx16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
#pragma unroll 1  // (unrolled, the compiler hoists every segment's 64 fragment registers to the top: 129 spills)
    for (int seg = 0; seg < S; ++seg) {
      unsigned char* const planes = smem + ((seg + par) & 1) * kPPlanesBytes;
      // this segment's weight fragments travel while the rows are cut
      f16x8 bw[8][2];
      const uint4* fp = frag + static_cast<int64_t>(seg) * kKspFragU4 + wn * 16 * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        bw[ks][0] = __builtin_bit_cast(f16x8, fp[(2 * ks) * 64]);
        bw[ks][1] = __builtin_bit_cast(f16x8, fp[(2 * ks + 1) * 64]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = i ? r1 : r0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          uint2 ph, pl;
          pair_cut4(ra[s][i], sc[i], ph, pl);
          unsigned char* const o = planes + r * kPRow + 2 * (32 * s + c4s);
          *reinterpret_cast<uint2*>(o) = ph;
          *reinterpret_cast<uint2*>(o + kPPlane) = pl;
        }
      }
      __syncthreads();
      fetch(t, seg + 1);
      __builtin_amdgcn_sched_barrier(0);
      {
        f16x8 a[2][2][2];  // [step parity][tm][plane]
        auto request = [&](int ks) {
          const int ko = 32 * ks + 16 * hi;
#pragma unroll
          for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
              a[ks & 1][tm][p] = *reinterpret_cast<const f16x8*>(planes + p * kPPlane + (tm * 32 + lj) * kPRow + ko);
        };
        request(0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          if (ks + 1 < 8) request(ks + 1);
#pragma unroll
          for (int tm = 0; tm < 2; ++tm) {  // small terms first
            acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks & 1][tm][1], bw[ks][0], acc[tm], 0, 0, 0);
            acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks & 1][tm][0], bw[ks][1], acc[tm], 0, 0, 0);
            acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks & 1][tm][0], bw[ks][0], acc[tm], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // ---- epilogue: factors, through the staging tile, out as 512-byte rows
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 f = ld4(s_inv + tm * 32 + 4 * hi + 8 * j);
        const float ff[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
          tile[(tm * 32 + 4 * hi + q + 8 * j) * kPLdT + nl] = acc[tm][4 * j + q] * (ff[q] * inv_w);
      }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rl = (tid >> 5) + 8 * i, c4 = tid & 31;
      const int64_t m = m0 + rl;
      if (CLS == 0 || m < M) st4(y + m * kPN + 4 * c4, ld4(tile + rl * kPLdT + 4 * c4));
    }
    par ^= 1;
  };
  auto full = [&]() { return t < tiles && (t + 1) * kPM <= M; };
  if (full()) {
    do_tile(IntTag<0>{});
    t += stride;
    while (full()) {
      do_tile(IntTag<0>{});
      t += stride;
    }
  }
  if (t < tiles) do_tile(IntTag<2>{});
}

}  // namespace

bool linear_ksp_ok(int64_t M, int64_t N_in, int64_t K_out) { return K_out == kPN && N_in == 4 * kPK && M >= 8192; }
size_t linear_ksp_scratch_bytes(int64_t N_in) { return static_cast<size_t>(N_in / kPK) * kKspFragU4 * sizeof(uint4) + 1024; }

// dx [M, 128] = dy [M, 512] w  (w [512][128] as stored); rowmax [M][4] = largest magnitude of every 128-column stretch of
// dy's rows; scratch: linear_ksp_scratch_bytes, 256-byte aligned
int linear_ksp_launch(const float* dy, const float* rowmax, const float* w, int64_t M, float* dx, void* scratch, hipStream_t st) {
  constexpr int S = 4;
  static_assert(S % 2 == 0, "the staging tile lies over the plane set the last segment did not use");
  constexpr size_t kLds = 2 * kPPlanesBytes + kPInvBytes;
  static_assert(kPStageBytes <= kPPlanesBytes, "the staging tile fits one set of planes");
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_linear_ksp<S>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     static_cast<int>(kLds));
  if (attr != hipSuccess) return STEMGNN_ERR_HIP;
  float* inv_w = static_cast<float*>(scratch);
  uint4* frag = reinterpret_cast<uint4*>(static_cast<unsigned char*>(scratch) + 1024);
  k_ksp_weight_frags<<<dim3(S, 4), kPT, 0, st>>>(w, S * kPK, frag, inv_w);
  STEMGNN_LAUNCH_CHECK();
  const int64_t tiles = (M + kPM - 1) / kPM;
  int64_t gx = wsp_blocks(kPN);
  if (gx > tiles) gx = tiles;
  g_wsp_calls.fetch_add(1, std::memory_order_relaxed);
  k_linear_ksp<S><<<static_cast<unsigned>(gx), kPT, kLds, st>>>(dy, rowmax, frag, inv_w, M, dx);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

// the quantiser's fused backward (csrc/vq.hip: stemgnn_vq_assign_bwd_fused) at D = Dc = 128
bool vq_bwd_wsp_ok(int64_t N, int64_t D, int64_t H, int64_t Dc) { return D == kPK && Dc == kPK && H >= 1 && N >= 8192; }

int vq_bwd_wsp_launch(const float* g_out, const float* w_out, const float* g_loss, float coef, const float* xp,
                      const float* norm, const int64_t* ind, const float* embed, int64_t N, int64_t H, int64_t K,
                      float* g_xp, float* rowmax, hipStream_t st) {
  const int64_t HD = H * kPK;
  const int64_t tiles = (N + kPM - 1) / kPM;
  int64_t gx = wsp_blocks(HD);
  if (gx > tiles) gx = tiles;
  g_wsp_calls.fetch_add(1, std::memory_order_relaxed);
  const WspVq vq{xp, norm, ind, embed, g_loss, coef, static_cast<int>(H), static_cast<int>(K), rowmax};
  if (rowmax)
    return launch_wsp<false, true, false, 2>(g_out, w_out, nullptr, N, static_cast<int>(HD), g_xp, nullptr, 0, N, nullptr,
                                             nullptr, 0, gx, st, vq);
  return launch_wsp<false, true, false, 1>(g_out, w_out, nullptr, N, static_cast<int>(HD), g_xp, nullptr, 0, N, nullptr,
                                           nullptr, 0, gx, st, vq);
}

}  // namespace stemgnn

extern "C" int64_t stemgnn_linear_wsp_calls(void) { return stemgnn::g_wsp_calls.load(std::memory_order_relaxed); }
