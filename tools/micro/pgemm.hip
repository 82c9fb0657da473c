// NOT part of libstemgnn_hip.so any more (round 3): the plane-operand products were built, bit-identical to the staged
// kernels and measured slower (DESIGN.md, K3 "planes"; profiles/round2_pgemm_bench.log).  Kept as the record of that
// experiment; it compiled against csrc/common.h of commit b742edb and exported the stemgnn_pgemm_* entry points.
// Dense products on operands that are ALREADY cut into their three exact bf16 pieces ("planes").
//
// csrc/linear.hip cuts every fp32 operand into h + m + l while it stages it: per wave and K chunk ~440 VALU
// instructions against 48 MFMAs, and every consumer of an activation repeats the cut on every launch.  Here the
// producers of an activation (aggregation, BatchNorm/activation pass, gradient kernels) write the three planes once
// ([3][M][K] bf16, plane p at p * plane_stride elements), the weights are cut once per call by a tiny prep kernel, and
// the products only move bytes and issue matrix instructions:
//   * tiles travel global -> LDS by DMA (global_load_lds_dwordx4: no VGPR round trip, no VALU), three stages deep,
//     counted s_waitcnt vmcnt so two stages stay in flight across the barrier of the third;
//   * fragments are plain ds_read_b128 (forward / backward-data) or ds_read_b64_tr_b16 transposing reads
//     (weight gradient, where both operands are contracted over their slow dimension);
//   * six v_mfma_f32_32x32x16_bf16 per 16 k exactly as in linear.hip (common.h: mfma_x3) -- same arithmetic, same
//     fp32 accuracy, results bit-identical to the register-staged kernels.
// Reference operations: nn.Linear forward / autograd of MySAGEConv.lin_l / lin_r (STEM-GNN/model/encoder.py:83-87),
// VectorQuantize.project_in / project_out (model/vq.py:881,1041), InnerProductDecoder.lin (model/encoder.py:364).
#include "common.h"

#include <cstdlib>

namespace stemgnn {

// 4 KB of zeros: the DMA source of tile rows / columns that lie outside an operand (LDS-DMA cannot write a
// constant; a lane that is out of range reads from here instead)
__device__ __attribute__((aligned(256))) uint16_t g_zero_page[2048];

namespace {

constexpr int kBlock = 256;
constexpr int kT = 128;          // output tile edge
constexpr int kKC = 16;          // contraction elements per stage (one MFMA k step)
constexpr int kStages = 3;
constexpr int kPlaneTile = kT * kKC * 2;        // bytes of one plane of one operand in a stage: 4 KB
constexpr int kStageBytes = 6 * kPlaneTile;     // 3 A planes + 3 B planes: 24 KB
constexpr int kDmaPerWave = 6;                  // 24 one-KB pieces per stage over 4 waves

struct PlaneOp {
  const uint16_t* p;   // [3][rows_total][K] bf16 bit patterns
  int64_t ps;          // plane stride in elements
  int K;               // row length
  int64_t rows;        // rows >= this count as zero and are not read
};

// 16 bytes per lane, global -> LDS at (wave-uniform) lds_addr + 16 * lane.  M0 is written in the statement that
// uses it and restored (the compiler keeps M0 for its own LDS accesses).
__device__ __forceinline__ void dma16(const void* g, uint32_t lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(g), "s"(lds_addr)
               : "memory");
}

__device__ __forceinline__ uint32_t lds_base(const void* smem) {
  return __builtin_amdgcn_readfirstlane(
      static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) const void*)smem)));
}

// C-tile register r of lane (lj, hi) is row (r&3) + 8*(r>>2) + 4*hi, column lj.
__device__ inline int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// ---------------------------------------------------------------------------------------
// Y[M, N] = A1[M, K1] W1[N, K1]^T (+ A2[M, K2] W2[N, K2]^T) + bias, all four operands as planes.
// Backward-data is the same kernel on the transposed weight planes ([K_out][N_contract]).
// Stage image: plane tile [128 rows][16 k] = 32-byte rows, fragments = one contiguous KB per ds_read_b128.
// ---------------------------------------------------------------------------------------
template <int BM, int BN, bool STATS>
__global__ void __launch_bounds__(kBlock, (BM == 64 ? 4 : 2))
k_pgemm_fwd(PlaneOp a1, PlaneOp w1, PlaneOp a2, PlaneOp w2, const float* __restrict__ bias, int64_t M, int N,
            float* __restrict__ y, float* __restrict__ stats_partial /*[row tiles][2][N]*/, int dbg) {
  // 4 waves as 2 (m) x 2 (n); a wave owns (BM/2) x (BN/2) outputs = TM x TN MFMA tiles
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int kPA = BM * kKC * 2, kPB = BN * kKC * 2;   // bytes of one plane of an operand in a stage
  constexpr int kStage = 3 * (kPA + kPB);
  constexpr int kPieces = kStage / 1024 / 4;              // one-KB DMA pieces per wave and stage
  static_assert(kStage % 4096 == 0, "pieces must divide over the four waves");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  __shared__ float s_stats[2][2][BN];
  const uint32_t smem_addr = lds_base(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, hi = lane >> 5, lj = lane & 31;
  const int64_t m0 = static_cast<int64_t>(blockIdx.x) * BM;
  const int n0 = blockIdx.y * BN;
  const int c1 = a1.K / kKC, c2 = a2.p ? a2.K / kKC : 0;
  // rows >= a1.rows of the first operand are zero: a tile past them starts at the second operand's chunks
  const int first = (m0 >= a1.rows && c2 > 0) ? c1 : 0;
  const int steps = c1 + c2;

  // DMA pieces.  A stage is 3 (kPA + kPB) bytes = pieces of one KB (32 rows x 32 B of one plane tile); piece q of a
  // stage goes to wave q % 4.  Per piece: lane -> row 32 i + (lane >> 1), 16-byte half (lane & 1).  A lane whose row
  // lies outside its operand reads the zero page and does not advance along k.
  const uint16_t* zero = g_zero_page;
  const uint16_t* src[2][kPieces];
  int adv[2][kPieces];
  uint32_t dst[kPieces];
#pragma unroll
  for (int j = 0; j < kPieces; ++j) {
    const int q = 4 * j + wave;                         // piece index in the stage
    const bool is_b = q >= 3 * (BM / 32);
    const int qq = is_b ? q - 3 * (BM / 32) : q;
    const int per_plane = (is_b ? BN : BM) / 32;
    const int plane = qq / per_plane, i = qq % per_plane;
    const int r = 32 * i + (lane >> 1);
    const int half = (lane & 1) * 8;
    dst[j] = (is_b ? 3 * kPA + plane * kPB : plane * kPA) + i * 1024;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const PlaneOp& op = is_b ? (o ? w2 : w1) : (o ? a2 : a1);
      const int64_t row = (is_b ? n0 : m0) + r;
      const bool ok = (o == 0 || c2 > 0) && row < op.rows && (is_b ? row < N : row < M);
      adv[o][j] = ok;
      src[o][j] = ok ? op.p + plane * op.ps + row * op.K + half : zero;
    }
  }
  auto issue = [&](int step) {
    const int o = step >= c1;
    const int k0 = (o ? step - c1 : step) * kKC;
    const uint32_t base = smem_addr + (step % kStages) * kStage;
#pragma unroll
    for (int j = 0; j < kPieces; ++j) {
      const bool is_b = 4 * j + wave >= 3 * (BM / 32);
      const bool off = is_b ? (dbg & 2) : (dbg & 8);
      dma16(off ? zero : src[o][j] + (adv[o][j] ? k0 : 0), base + dst[j]);
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;

  float bias_v[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + wn * (BN / 2) + tn * 32 + lj;
    bias_v[tn] = (bias != nullptr && n < N) ? bias[n] : 0.f;
  }
  // the bias loads are the only compiler-visible vector loads: retire them before the first DMA so that no later
  // compiler-placed wait can drain the ring
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if (first < steps) issue(first);
  if (first + 1 < steps) issue(first + 1);
  for (int step = first; step < steps; ++step) {
    // this wave's pieces of `step` have landed once at most one later stage is outstanding
    if (step + 1 < steps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPieces) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // everyone's pieces are in LDS; everyone is done reading stage step - 1
    if (step + 2 < steps) issue(step + 2);  // into the buffer stage step - 1 occupied
    const unsigned char* st = smem + (step % kStages) * kStage;
    bf16x8 a[TM][3], b[TN][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int t = 0; t < TM; ++t)
        a[t][p] = *reinterpret_cast<const bf16x8*>(st + p * kPA + (wm * (BM / 2) + t * 32 + lj) * 32 + hi * 16);
#pragma unroll
      for (int t = 0; t < TN; ++t)
        b[t][p] = *reinterpret_cast<const bf16x8*>(st + 3 * kPA + p * kPB + (wn * (BN / 2) + t * 32 + lj) * 32 + hi * 16);
    }
    if (dbg & 1) {  // ablation: no matrix work (keeps the fragment reads alive)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) acc[tm][0][0] += static_cast<float>(a[tm][0][0]) + static_cast<float>(b[0][2][1]);
      continue;
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma_x3(a[tm], b[tn], acc[tm][tn]);
  }
  if ((dbg & 4) && acc[0][0][0] != 12345.678f) return;  // ablation: no stores

  // ---- epilogue: bias, store (two 128-byte row segments per store instruction), column statistics
  const bool interior = (m0 + BM <= M) && (n0 + BN <= N);
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int nl = wn * (BN / 2) + tn * 32 + lj;
    const int n = n0 + nl;
    const float bv = bias_v[tn];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int64_t mrow0 = m0 + wm * (BM / 2) + tm * 32 + 4 * hi;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int64_t m = mrow0 + (q & 3) + 8 * (q >> 2);
        const float v = acc[tm][tn][q] + bv;
        if (interior || (m < M && n < N)) {
          y[m * N + n] = v;
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
    if (tid < BN && n0 + tid < N) {
      float* ps = stats_partial + static_cast<int64_t>(blockIdx.x) * 2 * N;
      ps[n0 + tid] = s_stats[0][0][tid] + s_stats[1][0][tid];
      ps[N + n0 + tid] = s_stats[0][1][tid] + s_stats[1][1][tid];
    }
  }
}

// ---------------------------------------------------------------------------------------
// dW[N, K] partial = dY[rows of the split]^T X[rows of the split]; db partial = column sums of dY.
// Both operands are contracted over their SLOW dimension (rows m), so a stage holds [16 m][128 cols] images (256-byte
// rows, DMA'd as whole 256-byte row segments) and the fragments are transposing reads (ds_read_b64_tr_b16: a 16-lane
// group reads a 4-row x 16-column block column-major).  The 16-byte chunks of a row are XOR-swizzled (on the DMA's
// source side; the LDS image is lane-linear) so that the transposing reads are conflict-free.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int tr_sw(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* plane, int col_base, int lane) {
  // operand fragment of v_mfma_f32_32x32x16_bf16 for the 32 columns col_base .. +31 of a [16][128] image: lane l
  // gets rows 8 (l / 32) .. + 7 of column col_base + l % 32, as two 4-row transposing reads
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int col = col_base + 16 * (g & 1) + 4 * p;
  const int chunk = col >> 3, byte = (col & 7) * 2;
  union { s16x4 v[2]; bf16x8 f; } u;
#pragma unroll
  for (int part = 0; part < 2; ++part) {
    const int row = 8 * (g >> 1) + 4 * part + q;
    const unsigned char* addr = plane + row * 256 + 16 * (chunk ^ tr_sw(row)) + byte;
    u.v[part] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(reinterpret_cast<uintptr_t>(addr)));
  }
  return u.f;
}

__global__ void __launch_bounds__(kBlock, 2)
k_pgemm_dw(PlaneOp dy, PlaneOp x, int64_t M, int64_t rows_per_split, float* __restrict__ partial_dw /*[S][N][K]*/,
           float* __restrict__ partial_db /*[S][N] or null*/) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const uint32_t smem_addr = lds_base(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wj = wave & 1, hi = lane >> 5, lj = lane & 31;
  const int N = dy.K, K = x.K;
  const int split = blockIdx.x;
  const int n0 = blockIdx.y * kT, k0 = blockIdx.z * kT;
  const int64_t mbeg = split * rows_per_split;
  const int64_t mend = min(M, mbeg + rows_per_split);
  const int steps = static_cast<int>((mend - mbeg + kKC - 1) / kKC);
  const bool do_db = partial_db != nullptr && blockIdx.z == 0;

  // DMA: a plane tile = 16 rows x 256 B = 4 pieces of 4 rows; wave w moves rows 4 w .. 4 w + 3 of every plane tile;
  // lane -> row 4 w + (lane >> 4), LDS chunk lane & 15, which must hold source chunk (lane & 15) ^ sw(row)
  const int r = 4 * wave + (lane >> 4);
  const int chunk = (lane & 15) ^ tr_sw(r);
  const uint16_t* zero = g_zero_page;
  const bool col_a = n0 + 8 * chunk < N, col_b = k0 + 8 * chunk < K;
  const uint16_t* sa = dy.p + n0 + 8 * chunk;
  const uint16_t* sb = x.p + k0 + 8 * chunk;
  auto issue = [&](int step) {
    const int64_t m = mbeg + static_cast<int64_t>(step) * kKC + r;
    const bool row_ok = m < mend;
    const uint32_t base = smem_addr + (step % kStages) * kStageBytes + wave * 1024;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      dma16((row_ok && col_a) ? sa + p * dy.ps + m * N : zero, base + p * kPlaneTile);
      dma16((row_ok && col_b) ? sb + p * x.ps + m * K : zero, base + (3 + p) * kPlaneTile);
    }
  };

  floatx16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
  float colsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // dY columns 8 c .. 8 c + 7 of row tid >> 4, c below

  if (steps > 0) issue(0);
  if (steps > 1) issue(1);
  for (int step = 0; step < steps; ++step) {
    if (step + 1 < steps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDmaPerWave) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (step + 2 < steps) issue(step + 2);
    const unsigned char* st = smem + (step % kStages) * kStageBytes;
    bf16x8 a[2][3], b[2][3];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t][p] = tr_frag(st + p * kPlaneTile, wi * 64 + t * 32, lane);
        b[t][p] = tr_frag(st + (3 + p) * kPlaneTile, wj * 64 + t * 32, lane);
      }
    if (do_db) {
      // this thread's 16 bytes of each dY plane image: row tid >> 4, LDS chunk tid & 15; h + m + l is the fp32 value
      const uint4 ph = *reinterpret_cast<const uint4*>(st + tid * 16);
      const uint4 pm = *reinterpret_cast<const uint4*>(st + kPlaneTile + tid * 16);
      const uint4 pl = *reinterpret_cast<const uint4*>(st + 2 * kPlaneTile + tid * 16);
      const uint32_t hh[4] = {ph.x, ph.y, ph.z, ph.w}, mm[4] = {pm.x, pm.y, pm.z, pm.w}, ll[4] = {pl.x, pl.y, pl.z, pl.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        colsum[2 * e] += (__uint_as_float(hh[e] << 16) + __uint_as_float(mm[e] << 16)) + __uint_as_float(ll[e] << 16);
        colsum[2 * e + 1] += (__uint_as_float(hh[e] & 0xffff0000u) + __uint_as_float(mm[e] & 0xffff0000u)) +
                             __uint_as_float(ll[e] & 0xffff0000u);
      }
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = mfma_x3(a[ti], b[tj], acc[ti][tj]);
  }

  float* pw = partial_dw + static_cast<int64_t>(split) * N * K;
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const int k = k0 + wj * 64 + tj * 32 + lj;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int n = n0 + wi * 64 + ti * 32 + acc_row(q, hi);
        if (n < N && k < K) pw[static_cast<int64_t>(n) * K + k] = acc[ti][tj][q];
      }
    }
  if (do_db) {
    // rows (tid >> 4) hold different source chunks at the same LDS chunk: un-swizzle, then add the 16 rows in order
    // (through the ring's memory: every wave is past its last fragment read once it reaches this barrier)
    __syncthreads();
    float (*s_db)[kT + 1] = reinterpret_cast<float (*)[kT + 1]>(smem);
    const int row = tid >> 4, c = (tid & 15) ^ tr_sw(row);
#pragma unroll
    for (int e = 0; e < 8; ++e) s_db[row][8 * c + e] = colsum[e];
    __syncthreads();
    if (tid < kT && n0 + tid < N) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) t += s_db[q][tid];
      partial_db[static_cast<int64_t>(split) * N + n0 + tid] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------
// fp32 -> planes.  Weights: `count` matrices in one launch, each optionally transposed (backward-data reads W^T).
// ---------------------------------------------------------------------------------------
constexpr int kMaxPrep = 12;
struct PrepTable {
  const float* src[kMaxPrep];
  uint16_t* dst[kMaxPrep];
  int rows[kMaxPrep], cols[kMaxPrep], transpose[kMaxPrep];
  int64_t first_block[kMaxPrep + 1];
  int count;
};

__device__ __forceinline__ void cut3(float v, uint16_t& h, uint16_t& m, uint16_t& l) {
  const uint32_t hb = hi16(v);
  const float r1 = v - __uint_as_float(hb);
  const uint32_t mb = hi16(r1);
  h = static_cast<uint16_t>(hb >> 16);
  m = static_cast<uint16_t>(mb >> 16);
  l = static_cast<uint16_t>(__float_as_uint(r1 - __uint_as_float(mb)) >> 16);
}

__global__ void __launch_bounds__(kBlock) k_prep_planes(PrepTable t) {
  int w = 0;
  while (w + 1 < t.count && static_cast<int64_t>(blockIdx.x) >= t.first_block[w + 1]) ++w;
  const int64_t i = (static_cast<int64_t>(blockIdx.x) - t.first_block[w]) * kBlock + threadIdx.x;
  const int64_t n = static_cast<int64_t>(t.rows[w]) * t.cols[w];
  if (i >= n) return;
  uint16_t h, m, l;
  cut3(t.src[w][i], h, m, l);
  int64_t o = i;
  if (t.transpose[w]) {
    const int64_t rr = i / t.cols[w], cc = i - rr * t.cols[w];
    o = cc * t.rows[w] + rr;
  }
  t.dst[w][o] = h;
  t.dst[w][n + o] = m;
  t.dst[w][2 * n + o] = l;
}

// activation rows: 8 elements per thread, 16-byte stores per plane
__global__ void __launch_bounds__(kBlock) k_split_rows(const float* __restrict__ x, int64_t n8, int64_t plane_stride,
                                                       uint16_t* __restrict__ planes) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n8;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    uint4 h, m, l;
    split8(ld4(x + 8 * i), ld4(x + 8 * i + 4), h, m, l);
    *reinterpret_cast<uint4*>(planes + 8 * i) = h;
    *reinterpret_cast<uint4*>(planes + plane_stride + 8 * i) = m;
    *reinterpret_cast<uint4*>(planes + 2 * plane_stride + 8 * i) = l;
  }
}

inline bool plane_dims_ok(int64_t M, int64_t N, int64_t K) {
  return M >= 0 && N > 0 && K > 0 && K % kKC == 0 && N % 4 == 0 && N <= 65536 && K <= 65536;
}

}  // namespace
}  // namespace stemgnn

using namespace stemgnn;

extern "C" {

int stemgnn_planes_ok(int64_t k) { return k > 0 && k % kKC == 0 ? 1 : 0; }

int stemgnn_split_planes(const float* x, int64_t rows, int64_t cols, uint16_t* planes, int64_t plane_stride,
                         void* stream_) {
  if (rows < 0 || cols <= 0 || (rows * cols) % 8 != 0 || plane_stride < rows * cols) return STEMGNN_ERR_INVALID_ARG;
  if (rows == 0) return STEMGNN_OK;
  if (!x || !planes) return STEMGNN_ERR_INVALID_ARG;
  const int64_t n8 = rows * cols / 8;
  int64_t g = (n8 + kBlock - 1) / kBlock;
  if (g > 256 * 16) g = 256 * 16;
  k_split_rows<<<static_cast<unsigned>(g), kBlock, 0, static_cast<hipStream_t>(stream_)>>>(x, n8, plane_stride, planes);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_prep_weight_planes(const float* const* weights, const int64_t* rows, const int64_t* cols,
                               const int32_t* transpose, uint16_t* const* planes, int32_t count, void* stream_) {
  if (count <= 0 || count > kMaxPrep || !weights || !rows || !cols || !planes) return STEMGNN_ERR_INVALID_ARG;
  PrepTable t{};
  t.count = count;
  int64_t blocks = 0;
  for (int i = 0; i < count; ++i) {
    if (!weights[i] || !planes[i] || rows[i] <= 0 || cols[i] <= 0 || rows[i] > (1 << 20) || cols[i] > (1 << 20))
      return STEMGNN_ERR_INVALID_ARG;
    t.src[i] = weights[i];
    t.dst[i] = planes[i];
    t.rows[i] = static_cast<int>(rows[i]);
    t.cols[i] = static_cast<int>(cols[i]);
    t.transpose[i] = transpose ? transpose[i] : 0;
    t.first_block[i] = blocks;
    blocks += (rows[i] * cols[i] + kBlock - 1) / kBlock;
  }
  t.first_block[count] = blocks;
  k_prep_planes<<<static_cast<unsigned>(blocks), kBlock, 0, static_cast<hipStream_t>(stream_)>>>(t);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_pgemm_fwd(const uint16_t* a1, int64_t a1_stride, int64_t k1, int64_t a1_rows, const uint16_t* w1,
                      const uint16_t* a2, int64_t a2_stride, int64_t k2, const uint16_t* w2, const float* bias,
                      int64_t M, int64_t N, float* y, float* stats_partial, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (!plane_dims_ok(M, N, k1) || k2 < 0 || k2 % kKC != 0) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(M)) return STEMGNN_ERR_TOO_LARGE;
  if (M == 0) return STEMGNN_OK;
  if (!a1 || !w1 || !y || (k2 > 0 && (!a2 || !w2))) return STEMGNN_ERR_INVALID_ARG;
  const int64_t r1 = (a1_rows < 0 || a1_rows > M) ? M : a1_rows;
  PlaneOp A1{a1, a1_stride, static_cast<int>(k1), k2 > 0 ? r1 : M};
  PlaneOp W1{w1, N * k1, static_cast<int>(k1), N};
  PlaneOp A2{k2 > 0 ? a2 : nullptr, a2_stride, static_cast<int>(k2), M};
  PlaneOp W2{k2 > 0 ? w2 : nullptr, N * k2, static_cast<int>(k2), N};
  static const int dbg = [] { const char* e = getenv("STEMGNN_PGEMM_DBG"); return e ? atoi(e) : 0; }();
  static const int tile = [] { const char* e = getenv("STEMGNN_PGEMM_TILE"); return e ? atoi(e) : 128; }();
  const int Ni = static_cast<int>(N);
#define STEMGNN_PG_LAUNCH(BM, BN)                                                                                      \
  do {                                                                                                                 \
    constexpr int lds = kStages * 3 * (BM + BN) * kKC * 2;                                                             \
    static const bool attr_set = [] {                                                                                  \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pgemm_fwd<BM, BN, true>),                             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds);                                      \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pgemm_fwd<BM, BN, false>),                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds);                                      \
      return true;                                                                                                     \
    }();                                                                                                               \
    (void)attr_set;                                                                                                    \
    dim3 grid(static_cast<unsigned>((M + BM - 1) / BM), static_cast<unsigned>((N + BN - 1) / BN));                     \
    if (stats_partial) k_pgemm_fwd<BM, BN, true><<<grid, kBlock, lds, st>>>(A1, W1, A2, W2, bias, M, Ni, y, stats_partial, dbg); \
    else k_pgemm_fwd<BM, BN, false><<<grid, kBlock, lds, st>>>(A1, W1, A2, W2, bias, M, Ni, y, nullptr, dbg);           \
  } while (0)
  if (tile == 64) STEMGNN_PG_LAUNCH(64, 64);
  else STEMGNN_PG_LAUNCH(128, 128);
#undef STEMGNN_PG_LAUNCH
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int64_t stemgnn_pgemm_stats_blocks(int64_t M) {
  static const int tile = [] { const char* e = getenv("STEMGNN_PGEMM_TILE"); return e ? atoi(e) : 128; }();
  return M <= 0 ? 0 : (M + tile - 1) / tile;
}

size_t stemgnn_pgemm_dw_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M < 0 || N <= 0 || K <= 0) return 0;
  const int64_t tiles = ((N + kT - 1) / kT) * ((K + kT - 1) / kT);
  return static_cast<size_t>(plane_split_count(M, tiles)) * (N * K + N) * sizeof(float) + 512;
}

int stemgnn_pgemm_dw(const uint16_t* dy, int64_t dy_stride, const uint16_t* x, int64_t x_stride, int64_t M, int64_t N,
                     int64_t K, float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (M < 0 || N <= 0 || K <= 0 || N % 8 != 0 || K % 8 != 0 || !dw) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(M)) return STEMGNN_ERR_TOO_LARGE;
  if (M == 0) {
    STEMGNN_HIP_TRY(hipMemsetAsync(dw, 0, sizeof(float) * N * K, st));
    if (db) STEMGNN_HIP_TRY(hipMemsetAsync(db, 0, sizeof(float) * N, st));
    return STEMGNN_OK;
  }
  if (!dy || !x || !workspace) return STEMGNN_ERR_INVALID_ARG;
  if (workspace_bytes < stemgnn_pgemm_dw_workspace_bytes(M, N, K)) return STEMGNN_ERR_WORKSPACE;
  const int64_t nt = (N + kT - 1) / kT, kt = (K + kT - 1) / kT;
  const int S = plane_split_count(M, nt * kt);
  int64_t rows = (M + S - 1) / S;
  rows = (rows + 31) / 32 * 32;  // the row splits of csrc/linear.hip: same partial sums, same bits
  float* pw = reinterpret_cast<float*>(align_up(reinterpret_cast<uintptr_t>(workspace), 256));
  float* pb = pw + static_cast<size_t>(S) * N * K;
  static const bool attr_set = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pgemm_dw), hipFuncAttributeMaxDynamicSharedMemorySize,
                              kStages * kStageBytes);
    return true;
  }();
  (void)attr_set;
  PlaneOp DY{dy, dy_stride, static_cast<int>(N), M};
  PlaneOp X{x, x_stride, static_cast<int>(K), M};
  dim3 grid(static_cast<unsigned>(S), static_cast<unsigned>(nt), static_cast<unsigned>(kt));
  k_pgemm_dw<<<grid, kBlock, kStages * kStageBytes, st>>>(DY, X, M, rows, pw, db ? pb : nullptr);
  STEMGNN_LAUNCH_CHECK();
  return reduce_splits_launch(pw, S, N * K, dw, db ? pb : nullptr, N, db, st);
}

}  // extern "C"
