// The big-tile product core (round 4): one hand-written 16-bit MFMA GEMM for the LARGE dense products of the path -- the
// D = 768 configurations (BASELINE configs 3 and 5, the reference's own default width, config/pretrain.yaml:3-16) --
// used by the projections (nn.Linear forward / backward-data / weight gradient: STEM-GNN/model/encoder.py:83-87,
// model/vq.py:881,1041, model/pt_model.py:42,80,94) AND by the quantiser's code assignment (model/vq.py:650-657), where
// the arg-max over the codes is taken from the accumulators: no [N, K] similarity matrix exists.  It replaces the
// vendor-library wrapper of round 3 (csrc/blaslt.hip, deleted): nothing on the path is a library GEMM any more.
//
//   NT form:  C[i][j] = sum over stretches s, k:  A_{pa(s)}[i][k] * B_{pb(s)}[j][k]     (both operands k-contiguous)
//   TN form:  C[i][j] = sum over stretches s, m:  A_{pa(s)}[m][i] * B_{pb(s)}[m][j]     (the weight gradient: contraction
//             over the operands' rows, read with transposing LDS reads from the same straight planes)
//
// Operands are 16-bit PLANES written once by a cut pass (k_bt_cut_*), a product is a list of STRETCHES of one contraction
// (which plane of each side, which K tiles) -- nothing is concatenated in memory:
//   * exact mode (stemgnn_linear_set_mode(1)), forward / backward-data / code assignment: the PAIR format -- two fp16
//     pieces of a row scaled by a power of two, three stretches (lo hi, hi lo, hi hi), fp32-accurate (k_bt_cut_pair);
//   * exact mode, weight gradients (and everything with stemgnn_linear_set_pair(0)): the three exact bf16 pieces
//     h + m + l of common.h, the six significant piece products, small terms first;
//   * bf16 GEMM mode (mode 2): one rounded bf16 plane, one stretch.
// At D = 768 the products are matrix-bound, so the plane traffic the round-2 plane kernels lost on (D = 128: HBM-bound)
// is noise here: 0.8 GB of planes against 4.8 PFLOP of executed matrix work for project_in at C3 size.
//
// Tile: 256 x 256 x 64 per 512-thread block (8 waves as 2 (a rows) x 4 (b rows); a wave owns 128 x 64 of C as 8 x 4
// accumulators of v_mfma_f32_16x16x32_{bf16,f16}), LDS 128 KiB = 2 buffers x 4 half tiles (128 rows x 128 B each), filled
// by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip), XOR-swizzled on the SOURCE address and on the read (the DMA
// writes lane-linear), fragments by ds_read_b128 (conflict-free: chunk ^= (row >> 1) & 7).  Blocks are persistent (one
// per CU) and walk XCD-contiguous work ids.
// Schedule (after MI355X cdna_hip_programming.md section 5, "256^2 8-phase template", re-derived for this staging
// order): a K tile is four phases, one C quadrant (64 x 32 per wave, 16 MFMAs) each; every phase is
//     { ds_read this quadrant's new fragments | stage ONE half tile (2 DMAs per wave) | s_waitcnt vmcnt(8) }
//     s_barrier  { s_waitcnt lgkmcnt(0); 16 MFMAs }  s_barrier
// and waves 4-7 run ONE BARRIER BEHIND waves 0-3, so that on every SIMD one wave multiplies while its partner loads.
// Half tiles are interleaved row sets chosen so that each is read in ONE phase only (A0: phase 0, B1: phase 1, A1: phase
// 2, the NEXT tile's B0: phase 3, into the fragment registers B1 has just left); a slot is re-staged three phases after
// its last read (the guide asks for two with staggered wave groups) and read at the earliest one phase after the wait
// that retires it: phase 0 stages B1(t+1), 1: A1(t+1), 2: B0(t+2), 3: A0(t+2); with one half tile per phase, "all but the
// four youngest half tiles have landed" (vmcnt(8)) is exactly what the next phase's reads need.
#include "common.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>
#include <atomic>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace stemgnn {
namespace {

constexpr int kBtThreads = 512;
constexpr int kTile = 256;                  // C tile edge
constexpr int kBK = 64;                     // contraction elements per K tile (128-byte rows)
constexpr int kHalfBytes = 128 * 128;       // one half tile: 128 rows x 128 B
constexpr int kBufBytes = 4 * kHalfBytes;   // A0 | A1 | B0 | B1
constexpr int kBtLds = 2 * kBufBytes;       // 128 KiB
constexpr int kMaxSeg = 12;

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct BtOp {
  const uint16_t* p;  // element (plane, row, k) at p + plane * ps + row * ld + k
  int64_t ps;         // plane stride, elements
  int64_t ld;         // row stride, elements (multiple of 8: 16-byte chunks)
  int64_t rows;       // rows of the operand; tile rows past them are clamped reads the epilogue masks
  int64_t bs;         // batch stride, elements
};
// one stretch of the contraction: planes (pa, pb), K tiles [k0, k0 + kt); (k0z, ktz) is the stretch for b-row tiles
// at or past BtArgs::b_full_rows (rows of a concatenated operand whose leading part is known to be zero there)
struct BtSeg { int pa, pb, k0, kt, k0z, ktz; };

enum { kEpiStore = 0, kEpiArgmax = 1 };

struct BtArgs {
  BtOp a, b;
  BtSeg seg[kMaxSeg];
  int nseg, tiles_m, tiles_n, splits, batch, work;
  int64_t b_full_rows;
  // kEpiStore: y[bz * y_bs + split * y_ss + j * ldy + i] = C[i][j] (+ bias[i]) for j < store_rows; stats (optional):
  // per b-row tile column sums / sums of squares over the rows j < b.rows: stats[tn][2][a.rows]
  float* y;
  int64_t ldy, y_bs, y_ss, store_rows;
  const float* bias;
  float* stats;
  // kEpiArgmax: per (batch, b row, a tile) the largest C[i][j] over the tile's valid a rows and the lowest i reaching it
  float* cand_val;
  int32_t* cand_idx;
  // pair format (two fp16 pieces of a row scaled by a power of two, k_bt_cut_pair): the factors that undo the scaling,
  // one per a row / b row (exact powers of two; NULL: none): C[i][j] is multiplied by a_scale[i] * b_scale[j]
  const float* a_scale;
  const float* b_scale;
  int dbg;  // measurement only (STEMGNN_BT_DBG): 1 = no output stores, 2 = no statistics
};

__device__ __forceinline__ void bt_dma16(const uint16_t* src, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

typedef short s16x4 __attribute__((ext_vector_type(4)));

// two transposing reads (ds_read_b64_tr_b16: a 16-lane group reads a block of 4 rows x 16 columns and every lane gets one
// column of it) = the eight contraction steps of one lane's MFMA fragment, for an operand whose contraction index is its
// SLOW dimension; `p4` is the second block, four rows (4 x 256 B) further on
// (inline asm: behind the builtin form hipcc puts s_waitcnt vmcnt(0) -- it cannot tell these reads from the LDS-DMA
// writes in flight and drains the staging pipeline in front of every phase, which cost 60 % of the kernel.  The reads
// are issued before the phase's barrier and waited for by the explicit s_waitcnt lgkmcnt(0) + sched_barrier behind it;
// their results are first touched by the MFMAs after that wait.)
typedef int v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x8 bt_tr_frag(uint32_t lds_addr) {
  union { v2i v[2]; bf16x8 f; } u;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(u.v[0]) : "v"(lds_addr));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(u.v[1]) : "v"(lds_addr));
  return u.f;
}
__device__ __forceinline__ uint32_t bt_lds_addr(const unsigned char* p) {
  return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) unsigned char*)p));
}

// TN = false: C[i][j] = sum_k A[i][k] B[j][k], both operands k-contiguous (planes [rows][k]).
// TN = true:  C[i][j] = sum_m A[m][i] B[m][j], the contraction over the operands' ROWS (planes [m][features], as the
//   straight cut pass writes them -- the weight gradient dW = dY^T X reads the same planes as the backward-data product,
//   no transposed copy exists): a K tile is 64 rows m; a half tile's LDS image is [64 m][128 features] (256-byte rows,
//   one DMA piece = 4 rows), its 32-byte units XOR-swizzled by (m & 3) | ((m >> 3) & 1) << 2 so that the transposing
//   reads of a 32-lane half (8 rows x 32 B) cover all 64 banks; BtOp::rows counts FEATURES, ld is the row stride, the
//   planes hold whole K tiles of rows (zero rows behind the operand's last).
template <int EPI, bool TN, bool F16>
__global__ void __launch_bounds__(kBtThreads) k_bt_gemm(const BtArgs g) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];  // the ONLY LDS object of this kernel
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const int fr = lane & 15, fq = lane >> 4;

  // ---- persistent blocks (one per CU): a block walks work ids (batch, split, a tile, b tile).  Blocks that share an XCD
  // (bid % 8) own one contiguous stretch of the ids and take them round-robin, so the blocks resident on an XCD work on
  // NEIGHBOURING ids at any time; consecutive ids walk patches of 4 b tiles x all a tiles -> they share operand panels
  // in their L2.  A block's output stores drain while it already stages and multiplies its next tile.
  const int nb = gridDim.x, bid = blockIdx.x;
  const int nx = nb < 8 ? nb : 8;
  const int xcd = bid % nx, local = bid / nx;
  const int nbx = nb / nx + (xcd < nb % nx ? 1 : 0);
  const int wq = g.work / nx, wrm = g.work % nx;
  const int wstart = xcd * wq + (xcd < wrm ? xcd : wrm), wcount = wq + (xcd < wrm ? 1 : 0);
  for (int wi = local; wi < wcount; wi += nbx) {
  const int wg = wstart + wi;
  const int per = g.tiles_m * g.tiles_n;
  const int outer = wg / per, rem = wg - outer * per;
  const int bz = outer / g.splits, split = outer - bz * g.splits;
  constexpr int PW = 4;
  const int patch = rem / (PW * g.tiles_m);
  const int pw = min(PW, g.tiles_n - patch * PW);
  const int idx = rem - patch * PW * g.tiles_m;
  const int tn = patch * PW + idx % pw, tm = idx / pw;
  const bool zrange = static_cast<int64_t>(tn) * kTile >= g.b_full_rows;

  // ---- staging: a half tile is 16 one-KB pieces (8 rows x 128 B); wave w moves pieces w and w + 8.  Lane -> row
  // (lane >> 3) of the piece, LDS chunk (lane & 7); it FETCHES source chunk (lane & 7) ^ ((row >> 1) & 7).
  // Half-tile row rho of half h is tile row (rho >> 6) * 128 + h * 64 + (rho & 63) of a, (rho >> 5) * 64 + h * 32 + (rho & 31) of b.
  const uint16_t* const abase = g.a.p + static_cast<int64_t>(bz) * g.a.bs;
  const uint16_t* const bbase = g.b.p + static_cast<int64_t>(bz) * g.b.bs;
  uint32_t offA[2][2], offB[2][2];
  if (!TN) {
    const int r0 = wid * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((r0 >> 1) & 7);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int rho = r0 + 64 * j;
        const int64_t ra = min(static_cast<int64_t>(tm) * kTile + (rho >> 6) * 128 + h * 64 + (rho & 63), g.a.rows - 1);
        const int64_t rb = min(static_cast<int64_t>(tn) * kTile + (rho >> 5) * 64 + h * 32 + (rho & 31), g.b.rows - 1);
        offA[h][j] = static_cast<uint32_t>(ra * g.a.ld + chunk * 8);
        offB[h][j] = static_cast<uint32_t>(rb * g.b.ld + chunk * 8);
      }
  } else {
    // piece = 4 rows m x 256 B; lane -> row (lane >> 4) of the piece, LDS chunk (lane & 15), source chunk ^ swizzle(m);
    // source chunk c of half h is features (c >> 3) * 128 + h * 64 + (c & 7) * 8 of a, (c >> 2) * 64 + h * 32 + (c & 3) * 8 of b
    const int sw = (((lane >> 4) & 3) | (((wid >> 1) & 1) << 2)) << 1;
    const int c = (lane & 15) ^ sw;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int64_t m = (wid + 8 * j) * 4 + (lane >> 4);
        const int64_t fa = min(static_cast<int64_t>(tm) * kTile + (c >> 3) * 128 + h * 64 + (c & 7) * 8, g.a.rows - 8);
        const int64_t fb = min(static_cast<int64_t>(tn) * kTile + (c >> 2) * 64 + h * 32 + (c & 3) * 8, g.b.rows - 8);
        offA[h][j] = static_cast<uint32_t>(m * g.a.ld + fa);
        offB[h][j] = static_cast<uint32_t>(m * g.b.ld + fb);
      }
  }
  unsigned char* const dma0 = smem + wid * 1024;  // this wave's first piece of half tile 0 of buffer 0

  // ---- the contraction as a list of K tiles: cursor over the segments
  int T = 0;
  // a split walks K tiles [split * c, (split + 1) * c) of every segment, c = ceil(tiles / splits) (the last split is short)
  auto split_range = [&](int kt, int* k_begin) {
    const int c = (kt + g.splits - 1) / g.splits;
    *k_begin = split * c;
    const int left = kt - split * c;
    return left < 0 ? 0 : (left < c ? left : c);
  };
  for (int s = 0; s < g.nseg; ++s) { int kb_; T += split_range(zrange ? g.seg[s].ktz : g.seg[s].kt, &kb_); }
  int si = 0, kk = 0;
  int cur_kt = 0;
  int64_t cur_a = 0, cur_b = 0;  // element offsets of the cursor segment's K tile 0 (this split)
  const int64_t ka = TN ? kBK * g.a.ld : kBK, kb = TN ? kBK * g.b.ld : kBK;  // elements from one K tile to the next
  auto seg_load = [&]() {
    const BtSeg sg = g.seg[si < g.nseg ? si : g.nseg - 1];
    int kbeg = 0;
    cur_kt = split_range(zrange ? sg.ktz : sg.kt, &kbeg);
    const int64_t k0 = static_cast<int64_t>(zrange ? sg.k0z : sg.k0) + kbeg;
    cur_a = sg.pa * g.a.ps + k0 * ka;
    cur_b = sg.pb * g.b.ps + k0 * kb;
  };
  auto cursor_next = [&]() {  // to the next K tile with cur_kt > 0 (segments may be empty in the z range)
    ++kk;
    while (kk >= cur_kt && si < g.nseg) { ++si; kk = 0; seg_load(); }
  };
  seg_load();
  while (cur_kt == 0 && si < g.nseg) { ++si; seg_load(); }

  floatx4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

  if (T > 0) {
    // ---- prologue: tile 0 whole (A0 B0 B1 A1) and A0, B0 of tile 1
    int64_t oa = cur_a + kk * ka, ob = cur_b + kk * kb;
#define BT_STAGE_A(BUF, H, OA)                                                                    \
  do {                                                                                            \
    bt_dma16(abase + (OA) + offA[H][0], dma0 + (BUF) * kBufBytes + (H) * kHalfBytes);            \
    bt_dma16(abase + (OA) + offA[H][1], dma0 + (BUF) * kBufBytes + (H) * kHalfBytes + 8192);     \
  } while (0)
#define BT_STAGE_B(BUF, H, OB)                                                                    \
  do {                                                                                            \
    bt_dma16(bbase + (OB) + offB[H][0], dma0 + (BUF) * kBufBytes + (2 + (H)) * kHalfBytes);      \
    bt_dma16(bbase + (OB) + offB[H][1], dma0 + (BUF) * kBufBytes + (2 + (H)) * kHalfBytes + 8192); \
  } while (0)
    BT_STAGE_A(0, 0, oa);
    BT_STAGE_B(0, 0, ob);
    BT_STAGE_B(0, 1, ob);
    BT_STAGE_A(0, 1, oa);
    cursor_next();
    // offsets of tile t + 1 (o*1) and t + 2 (o*2) while tile t is multiplied; past the end they repeat the last tile
    // and are never used (the staging that would read them is skipped)
    int64_t oa1 = cur_a + kk * ka, ob1 = cur_b + kk * kb;
    if (T > 1) {
      BT_STAGE_B(1, 0, ob1);  // (B0 before A0: phase 2's wait must already cover B0(1), which phase 3 reads)
      BT_STAGE_A(1, 0, oa1);
      cursor_next();
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    int64_t oa2 = cur_a + kk * ka, ob2 = cur_b + kk * kb;
    __builtin_amdgcn_s_barrier();            // every wave's share of tile 0 is in LDS
    if (wr == 1) __builtin_amdgcn_s_barrier();  // waves 4-7 run one barrier behind: they load while waves 0-3 multiply

    // ---- per-lane fragment addresses.  NT: row fr of a 16-row group, 16-byte chunk (s * 4 + fq) ^ (fr >> 1).
    // TN: rows fq * 8 + (fr >> 2) (+ 4: second read; + 32: k step 1) of the [64 m][128 f] image, 8 bytes at feature
    // 16 u + 4 (fr & 3) of 32-byte unit u ^ x, x = (fr >> 2) | (fq & 1) << 2
    const int roff = fr * 128 + ((fq ^ (fr >> 1)) << 4);
    const int tx = (fr >> 2) | ((fq & 1) << 2);
    const int trow = (fq * 8 + (fr >> 2)) * 256 + (fr & 3) * 8;
    const unsigned char* const ra0 = TN ? smem + trow + 32 * ((wr * 4) ^ (tx & 4)) : smem + wr * 64 * 128 + roff;
    const unsigned char* const rb0 = TN ? smem + 2 * kHalfBytes + trow + 32 * ((wc * 2) ^ (tx & 6))
                                        : smem + 2 * kHalfBytes + wc * 32 * 128 + roff;
    bf16x8 fa[4][2], fb[2][2][2];

#define BT_READ_A(BUF, MH)                                                                                        \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                              \
    if (TN) {                                                                                                     \
      const unsigned char* p_ = ra0 + (BUF) * kBufBytes + (MH) * kHalfBytes + 32 * (i_ ^ (tx & 3));              \
      fa[i_][0] = bt_tr_frag(bt_lds_addr(p_));                                                                    \
      fa[i_][1] = bt_tr_frag(bt_lds_addr(p_ + 32 * 256));                                                         \
    } else {                                                                                                      \
      fa[i_][0] = *reinterpret_cast<const bf16x8*>(ra0 + (BUF) * kBufBytes + (MH) * kHalfBytes + i_ * 2048);     \
      fa[i_][1] = *reinterpret_cast<const bf16x8*>((ra0 + (BUF) * kBufBytes + (MH) * kHalfBytes + i_ * 2048) + (64 - 2 * ((roff) & 64))); \
    }                                                                                                             \
  }
#define BT_READ_B(BUF, NH, SET)                                                                                      \
  _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                              \
    if (TN) {                                                                                                     \
      const unsigned char* p_ = rb0 + (BUF) * kBufBytes + (NH) * kHalfBytes + 32 * (j_ ^ (tx & 1));              \
      fb[SET][j_][0] = bt_tr_frag(bt_lds_addr(p_));                                                                \
      fb[SET][j_][1] = bt_tr_frag(bt_lds_addr(p_ + 32 * 256));                                                     \
    } else {                                                                                                      \
      fb[SET][j_][0] = *reinterpret_cast<const bf16x8*>(rb0 + (BUF) * kBufBytes + (NH) * kHalfBytes + j_ * 2048); \
      fb[SET][j_][1] = *reinterpret_cast<const bf16x8*>((rb0 + (BUF) * kBufBytes + (NH) * kHalfBytes + j_ * 2048) + (64 - 2 * ((roff) & 64))); \
    }                                                                                                             \
  }
#define BT_MFMA(MH, NH, SET)                                                                                          \
  do {                                                                                                            \
    __builtin_amdgcn_s_barrier();                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    __builtin_amdgcn_s_setprio(1);                                                                                \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                              \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                            \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                          \
          acc[(MH) * 4 + i_][(NH) * 2 + j_] =                                                                    \
              F16 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa[i_][s_]),                  \
                                                           __builtin_bit_cast(f16x8, fb[SET][j_][s_]),             \
                                                           acc[(MH) * 4 + i_][(NH) * 2 + j_], 0, 0, 0)             \
                  : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i_][s_], fb[SET][j_][s_],                           \
                                                            acc[(MH) * 4 + i_][(NH) * 2 + j_], 0, 0, 0);           \
    __builtin_amdgcn_s_setprio(0);                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    __builtin_amdgcn_s_barrier();                                                                                 \
  } while (0)
#define BT_WAIT(STAGED)                                                   \
  do {                                                                    \
    if (STAGED) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 \
  } while (0)
// Fragment reads per phase 8 / 4 / 8 / 4: this tile's B0 fragments were read in the previous tile's phase 3, into the
// register set B1 had just left (the two B sets swap roles every tile); staging order B1(t+1), A1(t+1), B0(t+2), A0(t+2).
// (Measured against the first form -- 12 / 4 / 8 / 0 reads, B0's fragments kept for phases 0 and 3 -- in one process
// on one box: within 1 % either way, profiles/round4_bigtile_schedule_ab.txt: the fragment reads are not what the load
// segments wait for.)
#define BT_TILE(BUF)                                                                                   \
  do {                                                                                                 \
    const bool ok1 = t + 1 < T, ok2 = t + 2 < T;                                                       \
    /* phase 0: quadrant (0, 0) */                                                                     \
    BT_READ_A(BUF, 0);                                                                                 \
    if (ok1) BT_STAGE_B((BUF) ^ 1, 1, ob1);                                                            \
    BT_WAIT(ok1);                                                                                      \
    BT_MFMA(0, 0, BUF);                                                                                \
    /* phase 1: quadrant (0, 1) */                                                                     \
    BT_READ_B(BUF, 1, (BUF) ^ 1);                                                                      \
    if (ok1) BT_STAGE_A((BUF) ^ 1, 1, oa1);                                                            \
    BT_WAIT(ok1);                                                                                      \
    BT_MFMA(0, 1, (BUF) ^ 1);                                                                          \
    /* phase 2: quadrant (1, 1) */                                                                     \
    BT_READ_A(BUF, 1);                                                                                 \
    if (ok2) BT_STAGE_B(BUF, 0, ob2);                                                                  \
    BT_WAIT(ok2);                                                                                      \
    BT_MFMA(1, 1, (BUF) ^ 1);                                                                          \
    /* phase 3: quadrant (1, 0); the NEXT tile's B0 fragments go into the set B1 has just left */      \
    if (ok1) BT_READ_B((BUF) ^ 1, 0, (BUF) ^ 1);                                                       \
    if (ok2) BT_STAGE_A(BUF, 0, oa2);                                                                  \
    BT_WAIT(ok2);                                                                                      \
    BT_MFMA(1, 0, BUF);                                                                                \
    ++t;                                                                                               \
    oa1 = oa2; ob1 = ob2;                                                                              \
    if (ok2) cursor_next();                                                                            \
    oa2 = cur_a + kk * ka;                                                                             \
    ob2 = cur_b + kk * kb;                                                                             \
  } while (0)

    BT_READ_B(0, 0, 0);  // tile 0's B0 fragments (every wave is past the barrier behind the prologue's wait)
    int t = 0;
    while (t < T) {
      BT_TILE(0);
      if (t < T) BT_TILE(1);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // the barrier waves 4-7 took first
#undef BT_TILE
#undef BT_WAIT
#undef BT_MFMA
#undef BT_READ_A
#undef BT_READ_B
#undef BT_STAGE_A
#undef BT_STAGE_B
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // LDS is free: every wave's DMAs have landed, every wave is done reading

  // ---- epilogues.  acc[mi][nj][r] = C[i][j]: i = tm * 256 + wr * 128 + mi * 16 + fq * 4 + r, j = tn * 256 + wc * 64 + nj * 16 + fr
  const int64_t i_wave = static_cast<int64_t>(tm) * kTile + wr * 128 + fq * 4;
  const int64_t j_wave = static_cast<int64_t>(tn) * kTile + wc * 64 + fr;
  if (EPI == kEpiStore) {
    float* const y = g.y + static_cast<int64_t>(bz) * g.y_bs + static_cast<int64_t>(split) * g.y_ss;
    float s1[8][4], s2[8][4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      const int64_t i = i_wave + mi * 16;
      const bool iok = i < g.a.rows;  // a.rows is a multiple of 4 (host-checked): the four values share the verdict
      float4 bv = zero4(), sa = make_float4(1.f, 1.f, 1.f, 1.f);
      if (g.bias && iok) bv = ld4(g.bias + i);
      if (g.a_scale && iok) sa = ld4(g.a_scale + i);
#pragma unroll
      for (int r = 0; r < 4; ++r) s1[mi][r] = s2[mi][r] = 0.f;
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) {
        const int64_t j = j_wave + nj * 16;
        const float sb = (g.b_scale && j < g.b.rows) ? g.b_scale[j] : 1.f;  // (powers of two: the products are exact)
        const float4 v = make_float4(acc[mi][nj][0] * (sa.x * sb) + bv.x, acc[mi][nj][1] * (sa.y * sb) + bv.y,
                                     acc[mi][nj][2] * (sa.z * sb) + bv.z, acc[mi][nj][3] * (sa.w * sb) + bv.w);
        if (iok && j < g.store_rows && !(g.dbg & 1) && !((g.dbg & 4) && (nj & 1))) {
          if (g.dbg & 16) __builtin_nontemporal_store(floatx4{v.x, v.y, v.z, v.w}, reinterpret_cast<floatx4*>(y + j * g.ldy + i));
          else st4(y + j * g.ldy + i, v);
        }
        if (g.stats && j < g.b.rows && !(g.dbg & 2)) {
          s1[mi][0] += v.x; s1[mi][1] += v.y; s1[mi][2] += v.z; s1[mi][3] += v.w;
          s2[mi][0] += v.x * v.x; s2[mi][1] += v.y * v.y; s2[mi][2] += v.z * v.z; s2[mi][3] += v.w * v.w;
        }
      }
    }
    if (g.stats && !(g.dbg & 2)) {
      // over the 16 lanes that share fq (the b rows of a 16-column group), then over the four waves of this a half
      // through LDS, in fixed order: slab [tn][2][a.rows]
      float* const red = reinterpret_cast<float*>(smem);  // [2 stats][8 waves][128 i]
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float a = s1[mi][r], b = s2[mi][r];
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) {
            a += __shfl_xor(a, o, 64);
            b += __shfl_xor(b, o, 64);
          }
          if (fr == 0) {
            red[(0 * 8 + wid) * 128 + mi * 16 + fq * 4 + r] = a;
            red[(1 * 8 + wid) * 128 + mi * 16 + fq * 4 + r] = b;
          }
        }
      __syncthreads();
      // 512 threads: (stat, a half, 128 i)
      const int st = tid >> 8, half = (tid >> 7) & 1, il = tid & 127;
      const int64_t i = static_cast<int64_t>(tm) * kTile + half * 128 + il;
      if (i < g.a.rows) {
        const float* rr = red + (st * 8 + half * 4) * 128 + il;
        g.stats[(static_cast<int64_t>(tn) * 2 + st) * g.a.rows + i] = (rr[0] + rr[128]) + (rr[256] + rr[384]);
      }
      __syncthreads();  // the partial sums' LDS is read: the next tile may stage
    }
  } else {
    // running arg-max over this tile's a rows (codes), per b row (data row): a lane's codes ascend with (mi, r), so a
    // strict '>' keeps the lowest index; then the four fq groups, then the two a halves (lower codes win ties)
    float* const cv = reinterpret_cast<float*>(smem);                 // [2 a halves][256 j]
    int32_t* const ci = reinterpret_cast<int32_t*>(smem + 2 * 256 * 4);
#pragma unroll
    for (int nj = 0; nj < 4; ++nj) {
      float best = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = static_cast<int>(i_wave) + mi * 16 + r;
          const float v = (g.a_scale && i < g.a.rows) ? acc[mi][nj][r] * g.a_scale[static_cast<int64_t>(bz) * g.a.rows + i]
                                                      : acc[mi][nj][r];
          if (i < g.a.rows && v > best) { best = v; bi = i; }
        }
#pragma unroll
      for (int o = 16; o < 64; o <<= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      }
      if (fq == 0) {
        cv[wr * 256 + wc * 64 + nj * 16 + fr] = best;
        ci[wr * 256 + wc * 64 + nj * 16 + fr] = bi;
      }
    }
    __syncthreads();
    if (tid < 256) {
      const int64_t j = static_cast<int64_t>(tn) * kTile + tid;
      if (j < g.b.rows) {
        float best = cv[tid];
        int bi = ci[tid];
        const float ov = cv[256 + tid];
        const int oi = ci[256 + tid];
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        const int64_t o = (static_cast<int64_t>(bz) * g.b.rows + j) * g.tiles_m + tm;
        if (g.b_scale) best *= g.b_scale[static_cast<int64_t>(bz) * g.b.rows + j];  // a positive factor per data row
        g.cand_val[o] = best;
        g.cand_idx[o] = bi;
      }
    }
    __syncthreads();  // the candidates' LDS is read: the next tile may stage
  }
  }  // work ids
}


// ---------------------------------------------------------------------------------------------------------------------
// Cut passes: fp32 (or bf16-stored) operands -> bf16 planes.  NP = 3: the exact pieces h, m, l (common.h: split3);
// NP = 1: one plane rounded to nearest even (the bf16 GEMM mode).  A bf16-stored operand IS its own h piece: its m and l
// planes are never read (the callers' segment lists skip them).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kCutThreads = 256;

// out[plane][m][c0 ...] for two sources side by side: columns [0, K1) from x1 (rows >= x1_rows: zero), [K1, K1 + K2) from
// x2 (element kind x2_kind); ld = row stride of the planes.  ssq (optional): per row and per `seg_len` columns of the
// x1 part the sum of squares (the quantiser's row norms): ssq[m][K1 / seg_len].
template <int NP>
__global__ void __launch_bounds__(kCutThreads)
k_bt_cut_rows(const float* __restrict__ x1, int64_t ldx1, int K1, int64_t x1_rows, const void* __restrict__ x2, int x2_kind,
              int K2, int64_t M, uint16_t* __restrict__ out, int64_t ps, int64_t ld) {
  const int nv = (K1 + K2) / 4;
  const int64_t total = M * nv;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kCutThreads + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * kCutThreads) {
    const int64_t m = i / nv;
    const int c = static_cast<int>(i - m * nv) * 4;
    uint16_t* o = out + m * ld + c;
    if (c >= K1 && x2_kind == kBF16) {  // stored as bf16: the h plane as it is
      *reinterpret_cast<uint2*>(o) = *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(x2) + m * K2 + (c - K1));
      continue;
    }
    float4 v = zero4();
    if (c < K1) {
      if (m < x1_rows) v = ld4(x1 + m * ldx1 + c);
    } else {
      v = ld4(static_cast<const float*>(x2) + m * K2 + (c - K1));
    }
    if (NP == 1) {
      *reinterpret_cast<uint2*>(o) = pack_rne(v);
    } else {
      uint2 h, mm, l;
      split3(v, h, mm, l);
      *reinterpret_cast<uint2*>(o) = h;
      *reinterpret_cast<uint2*>(o + ps) = mm;
      *reinterpret_cast<uint2*>(o + 2 * ps) = l;
    }
  }
}

// the quantiser's rows: one wave per row of xp [N, H * Dc]: the three planes and ssq[row][h] = |xp_h|^2
__global__ void __launch_bounds__(kCutThreads)
k_bt_cut_rows_ssq(const float* __restrict__ x, int64_t N, int H, int Dc, uint16_t* __restrict__ out, int64_t ps,
                  float* __restrict__ ssq) {
  const int lane = threadIdx.x & 63;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * (kCutThreads / 64) + (threadIdx.x >> 6);
  if (row >= N) return;
  const int64_t HD = static_cast<int64_t>(H) * Dc;
  for (int h = 0; h < H; ++h) {
    float acc = 0.f;
    for (int c = 4 * lane; c < Dc; c += 256) {
      const float4 v = ld4(x + row * HD + static_cast<int64_t>(h) * Dc + c);
      acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      uint2 a, b, l;
      split3(v, a, b, l);
      uint16_t* o = out + row * HD + static_cast<int64_t>(h) * Dc + c;
      *reinterpret_cast<uint2*>(o) = a;
      *reinterpret_cast<uint2*>(o + ps) = b;
      *reinterpret_cast<uint2*>(o + 2 * ps) = l;
    }
    acc = wave_sum(acc);
    if (lane == 0) ssq[row * H + h] = acc;
  }
}

// ---- the pair format (round 4): fp32 accuracy from TWO fp16 pieces and three matrix passes instead of three bf16 pieces
// and six.  A row (or, for the quantiser's input, a head's stretch of a row) is scaled by a power of two that puts its
// largest magnitude into [2^14, 2^15) -- the top of fp16's range --, then cut into hi = fp16(x s) (round to nearest: 11
// significant bits) and lo = fp16(x s - hi) (the remainder is exact in fp32; lo keeps 11 more bits).  For an element within
// 2^-17 of the row's largest that is x to 2^-24 relative -- fp32's own resolution; a smaller element is kept to an
// ABSOLUTE 2^-25 of the scaled row, i.e. 2^-39 of the row's largest magnitude: far below what the fp32 sum of the row's
// products resolves.  a b = hi_a hi_b + hi_a lo_b + lo_a hi_b (+ lo_a lo_b < 2^-24 |a b|, dropped), accumulated in the
// fp32 MFMA accumulator, small terms first; the epilogue multiplies by the two inverse scales (powers of two: exact).
// The scale is per contraction ROW, so the format serves the products that contract along rows' columns (forward,
// backward-data, the code assignment); the weight gradient contracts over the rows themselves and keeps the bf16 pieces.
// A zero or non-finite row is not scaled; a bf16-stored stretch has lo = 0 exactly (its lo plane is never read).
// One wave per row; `seg` = scaling stretch in columns (the whole row, or code_dim); out[plane][m][ld], plane 0 = hi;
// inv_scale[(c / seg) * M + m]; ssq (optional) likewise: the stretch's sum of squares (the quantiser's row norms).
// NV = float4 chunks a lane keeps in registers per stretch (stretch <= NV * 256 columns): the row is READ ONCE -- largest
// magnitude, then the cut, from registers; NV = 0: any stretch length, two reads of the row (the second one misses the
// L2 at these sizes: one third more traffic for a pass that is HBM-bound).
template <int NV>
__global__ void __launch_bounds__(kCutThreads)
k_bt_cut_pair(const float* __restrict__ x1, int64_t ldx1, int K1, int64_t x1_rows, const void* __restrict__ x2, int x2_kind,
              int K2, int64_t M, int seg, uint16_t* __restrict__ out, int64_t ps, int64_t ld, float* __restrict__ inv_scale,
              float* __restrict__ ssq) {
  const int lane = threadIdx.x & 63;
  const int64_t m = static_cast<int64_t>(blockIdx.x) * (kCutThreads / 64) + (threadIdx.x >> 6);
  if (m >= M) return;
  const int K = K1 + K2;
  auto load4 = [&](int c) -> float4 {
    if (c < K1) return m < x1_rows ? ld4(x1 + m * ldx1 + c) : zero4();
    return ld4_kind(x2, m * static_cast<int64_t>(K2) + (c - K1), x2_kind);
  };
  auto cut4 = [&](int c, float4 v, float sc) {
    const float xs[4] = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};
    uint32_t hb[4], lb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const _Float16 h = static_cast<_Float16>(xs[e]);
      const _Float16 l = static_cast<_Float16>(xs[e] - static_cast<float>(h));
      hb[e] = static_cast<uint32_t>(__builtin_bit_cast(uint16_t, h));
      lb[e] = static_cast<uint32_t>(__builtin_bit_cast(uint16_t, l));
    }
    uint16_t* o = out + m * ld + c;
    *reinterpret_cast<uint2*>(o) = make_uint2(hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16));
    *reinterpret_cast<uint2*>(o + ps) = make_uint2(lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16));
  };
  for (int s0 = 0, si = 0; s0 < K; s0 += seg, ++si) {
    float4 keep[NV > 0 ? NV : 1];
    float mx = 0.f, sq = 0.f;
    if (NV > 0) {
#pragma unroll
      for (int t = 0; t < NV; ++t) {
        const int c = s0 + 4 * lane + 256 * t;
        keep[t] = c < s0 + seg ? load4(c) : zero4();
      }
#pragma unroll
      for (int t = 0; t < NV; ++t) {
        const float4 v = keep[t];
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        sq += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      }
    } else {
      for (int c = s0 + 4 * lane; c < s0 + seg; c += 256) {
        const float4 v = load4(c);
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        sq += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    // (fmaxf drops NaNs: a row holding one is scaled by its largest finite magnitude and keeps its NaN)
    int shift = 0;
    if (mx > 0.f && mx <= 3.0e38f) {
      shift = 14 - ilogbf(mx);
      shift = shift > 110 ? 110 : (shift < -110 ? -110 : shift);
    }
    const float sc = ldexpf(1.f, shift);
    if (lane == 0) inv_scale[static_cast<int64_t>(si) * M + m] = ldexpf(1.f, -shift);
    if (ssq) {
      sq = wave_sum(sq);
      if (lane == 0) ssq[static_cast<int64_t>(si) * M + m] = sq;
    }
    if (NV > 0) {
#pragma unroll
      for (int t = 0; t < NV; ++t) {
        const int c = s0 + 4 * lane + 256 * t;
        if (c < s0 + seg) cut4(c, keep[t], sc);
      }
    } else {
      for (int c = s0 + 4 * lane; c < s0 + seg; c += 256) cut4(c, load4(c), sc);
    }
  }
}

// launch with the smallest register-resident form that holds a stretch
static int bt_cut_pair(hipStream_t st, const float* x1, int64_t ldx1, int K1, int64_t x1_rows, const void* x2, int x2_kind, int K2,
                       int64_t M, int seg, uint16_t* out, int64_t ps, int64_t ld, float* inv_scale, float* ssq) {
  const unsigned grid = static_cast<unsigned>((M + kCutThreads / 64 - 1) / (kCutThreads / 64));
  const int nv = (seg + 255) / 256;
#define BT_CUT_GO(NV) k_bt_cut_pair<NV><<<grid, kCutThreads, 0, st>>>(x1, ldx1, K1, x1_rows, x2, x2_kind, K2, M, seg, out, ps, ld, inv_scale, ssq)
  if (nv <= 1) BT_CUT_GO(1);
  else if (nv <= 2) BT_CUT_GO(2);
  else if (nv <= 3) BT_CUT_GO(3);
  else if (nv <= 4) BT_CUT_GO(4);
  else if (nv <= 6) BT_CUT_GO(6);
  else if (nv <= 8) BT_CUT_GO(8);
  else if (nv <= 12) BT_CUT_GO(12);
  else if (nv <= 16) BT_CUT_GO(16);
  else BT_CUT_GO(0);
#undef BT_CUT_GO
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

// Straight planes of x [M, C] (element kind `kind`) written for M_out >= M rows -- the rows behind the operand's last are
// ZERO: the contraction of a weight gradient (k_bt_gemm<.., TN = true>) walks the rows in whole K tiles -- and, in the
// same pass, partial column sums (the bias gradient): colsum[blockIdx.x][C] over the block's 256 rows (finished by
// k_bt_colsum_finish in fixed order).  A block owns 256 rows x 256 columns: wave w takes rows w, w + 4, ... (a wave reads
// one contiguous KB of a row per step, four rows in flight), a lane 4 columns; the four waves' sums meet in LDS.
template <int NP>
__global__ void __launch_bounds__(kCutThreads)
k_bt_cut_rows_colsum(const void* __restrict__ x_, int kind, int64_t M, int64_t M_out, int C, uint16_t* __restrict__ out,
                     int64_t ps, float* __restrict__ colsum) {
  __shared__ float4 red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = (blockIdx.y * 64 + lane) * 4;
  const bool cok = c < C;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * 256;
  float4 cs = zero4();
  if (cok) {
    for (int it = 0; it < 64; it += 4) {
      float4 v[4];
      uint2 raw[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t r = r0 + w + 4 * (it + u);
        v[u] = zero4();
        raw[u] = make_uint2(0u, 0u);
        if (r < M) {
          if (kind == kBF16) raw[u] = *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(x_) + r * C + c);
          else v[u] = ld4(static_cast<const float*>(x_) + r * C + c);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t r = r0 + w + 4 * (it + u);
        if (r >= M_out) continue;
        uint16_t* o = out + r * C + c;
        if (kind == kBF16) {  // stored as bf16: the h plane as it is (its m and l planes are never read)
          *reinterpret_cast<uint2*>(o) = raw[u];
          cs.x += __uint_as_float(raw[u].x << 16); cs.y += __uint_as_float(raw[u].x & 0xffff0000u);
          cs.z += __uint_as_float(raw[u].y << 16); cs.w += __uint_as_float(raw[u].y & 0xffff0000u);
          continue;
        }
        cs.x += v[u].x; cs.y += v[u].y; cs.z += v[u].z; cs.w += v[u].w;
        if (NP == 1) {
          *reinterpret_cast<uint2*>(o) = pack_rne(v[u]);
        } else {
          uint2 h, mm, l;
          split3(v[u], h, mm, l);
          *reinterpret_cast<uint2*>(o) = h;
          *reinterpret_cast<uint2*>(o + ps) = mm;
          *reinterpret_cast<uint2*>(o + 2 * ps) = l;
        }
      }
    }
  }
  red[w][lane] = cs;
  __syncthreads();
  if (w == 0 && cok) {
    const float4 a0 = red[0][lane], a1 = red[1][lane], a2 = red[2][lane], a3 = red[3][lane];
    st4(colsum + static_cast<int64_t>(blockIdx.x) * C + c,
        make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                    (a0.w + a1.w) + (a2.w + a3.w)));
  }
}

// Transposed planes: x [R, C] (element kind `kind`, row stride C) -> out[plane][c][r], row stride ld >= R rounded up;
// columns r in [R, ld) are written as zeros (the contraction runs over r in whole K tiles).  A block owns 64 columns c
// and 256 rows r (four 64 x 64 sub-tiles transposed through LDS).  colsum (optional): partial column sums of x over the
// block's rows, colsum[blockIdx.x][C] (the bias gradient: finished by k_bt_colsum_finish in fixed order).
template <int NP>
__global__ void __launch_bounds__(kCutThreads)
k_bt_cut_cols(const void* __restrict__ x_, int kind, int64_t R, int C, uint16_t* __restrict__ out, int64_t ps, int64_t ld,
              float* __restrict__ colsum) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[NP][64][64 + 8];  // [plane][c][r], 144-byte rows
  __shared__ float csum[4][64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c4 = (tid & 15) * 4;   // this thread's four columns inside the 64-column strip
  const int rq = tid >> 4;         // 0..15: row inside a group of 16
  const int c0 = blockIdx.y * 64;
  float4 cs = zero4();
  for (int sub = 0; sub < 4; ++sub) {
    const int64_t r0 = (static_cast<int64_t>(blockIdx.x) * 4 + sub) * 64;
    if (r0 >= ld) break;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rl = rq + 16 * it;
      const int64_t r = r0 + rl;
      float4 v = zero4();
      if (r < R && c0 + c4 < C) v = ld4_kind(x_, r * C + c0 + c4, kind);
      cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
      const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (NP == 1) {
          tile[0][c4 + e][rl] = static_cast<uint16_t>(rne_bits(vv[e]));
        } else {
          const uint32_t hb = hi16(vv[e]);
          const float r1 = vv[e] - __uint_as_float(hb);
          const uint32_t mb = hi16(r1);
          const uint32_t lb = __float_as_uint(r1 - __uint_as_float(mb));
          tile[0][c4 + e][rl] = static_cast<uint16_t>(hb >> 16);
          tile[NP > 1 ? 1 : 0][c4 + e][rl] = static_cast<uint16_t>(mb >> 16);
          tile[NP > 2 ? 2 : 0][c4 + e][rl] = static_cast<uint16_t>(lb >> 16);
        }
      }
    }
    __syncthreads();
    // 64 columns x 8 sixteen-byte chunks per plane
#pragma unroll
    for (int q = tid; q < NP * 64 * 8; q += kCutThreads) {
      const int plane = q / 512, cc = (q >> 3) & 63, ch = q & 7;
      if (c0 + cc < C && r0 + ch * 8 < ld)
        *reinterpret_cast<uint4*>(out + plane * ps + static_cast<int64_t>(c0 + cc) * ld + r0 + ch * 8) =
            *reinterpret_cast<const uint4*>(&tile[plane][cc][ch * 8]);
    }
    __syncthreads();
  }
  if (colsum) {
    // the 16 threads that share c4: lanes c4/4 + 16 k of each wave, then the four waves
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
      cs.x += __shfl_xor(cs.x, o, 64); cs.y += __shfl_xor(cs.y, o, 64);
      cs.z += __shfl_xor(cs.z, o, 64); cs.w += __shfl_xor(cs.w, o, 64);
    }
    if (lane < 16) {
      csum[w][lane * 4 + 0] = cs.x; csum[w][lane * 4 + 1] = cs.y;
      csum[w][lane * 4 + 2] = cs.z; csum[w][lane * 4 + 3] = cs.w;
    }
    __syncthreads();
    if (tid < 64 && c0 + tid < C)
      colsum[static_cast<int64_t>(blockIdx.x) * C + c0 + tid] = (csum[0][tid] + csum[1][tid]) + (csum[2][tid] + csum[3][tid]);
  }
}

// db[c] = sum over slabs of colsum[slab][c], fixed order, fp64 accumulation: 16 columns x 16 slab slices per block
__global__ void __launch_bounds__(kCutThreads)
k_bt_colsum_finish(const float* __restrict__ partial, int slabs, int C, float* __restrict__ db) {
  __shared__ double red[kCutThreads];
  const int cl = threadIdx.x & 15, slice = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double s = 0.0;
  if (c < C)
    for (int b = slice; b < slabs; b += 16) s += partial[static_cast<int64_t>(b) * C + c];
  red[threadIdx.x] = s;
  __syncthreads();
  if (slice == 0 && c < C) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k * 16 + cl];
    db[c] = static_cast<float>(t);
  }
}

// out[i] = sum over s < splits of slab[s][i] (+ nothing): the weight gradient's split slabs, fixed order
__global__ void __launch_bounds__(kCutThreads)
k_bt_reduce_slabs(const float* __restrict__ slab, int splits, int64_t n, float* __restrict__ out) {
  const int64_t i = (static_cast<int64_t>(blockIdx.x) * kCutThreads + threadIdx.x) * 4;
  if (i >= n) return;
  float4 a = ld4(slab + i);
  for (int s = 1; s < splits; ++s) {
    const float4 v = ld4(slab + static_cast<int64_t>(s) * n + i);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  st4(out + i, a);
}

// ---- the quantiser's finish: candidates of the code tiles -> index, row norm, commitment term
// one thread per (row, head); a block's terms -> partial[block] (fp64), finished by k_bt_commit_finish
__global__ void __launch_bounds__(kCutThreads)
k_bt_argmax_finish(const float* __restrict__ cand_val, const int32_t* __restrict__ cand_idx, const float* __restrict__ ssq,
                   const float* __restrict__ esq, int64_t N, int H, int K, int tiles, float* __restrict__ norm_out,
                   int64_t* __restrict__ ind_out, double* __restrict__ partial, int ssq_by_head) {
  __shared__ double red[kCutThreads / 64];
  const int64_t item = static_cast<int64_t>(blockIdx.x) * kCutThreads + threadIdx.x;  // row * H + h
  double term = 0.0;
  if (item < N * H) {
    const int64_t row = item / H;
    const int h = static_cast<int>(item - row * H);
    const int64_t o = (static_cast<int64_t>(h) * N + row) * tiles;
    float best = cand_val[o];
    int bi = cand_idx[o];
    for (int t = 1; t < tiles; ++t) {  // tiles ascend in code index: strict '>' keeps the lowest index among equals
      const float v = cand_val[o + t];
      if (v > best) { best = v; bi = cand_idx[o + t]; }
    }
    const float nrm = sqrtf(ssq[ssq_by_head ? static_cast<int64_t>(h) * N + row : item]);
    const float inv = 1.0f / fmaxf(nrm, 1e-12f), xn2 = nrm * inv;  // F.normalize eps
    ind_out[item] = static_cast<int64_t>(bi);
    norm_out[item] = nrm;
    if (esq) term = static_cast<double>(esq[static_cast<int64_t>(h) * K + bi] + xn2 * xn2 - 2.0f * best * inv);
  }
  for (int o = 32; o > 0; o >>= 1) term += __shfl_xor(term, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = term;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// the form that hands the per-head codes on (csrc/vq.hip: k_vq_assign with `quant`): one wave per (row, head) gathers the
// winning code, writes x^ + (q - x^) (training, vq.py:937) or q, optionally x^ itself, and sums |q - x^|^2
__global__ void __launch_bounds__(kCutThreads)
k_bt_gather_commit(const float* __restrict__ xp, int64_t N, int H, int Dc, int K, const float* __restrict__ embed,
                   const int64_t* __restrict__ ind, const float* __restrict__ ssq, int training, float* __restrict__ quant,
                   float* __restrict__ xn_out, double* __restrict__ partial, int ssq_by_head) {
  __shared__ double red[kCutThreads / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t item = static_cast<int64_t>(blockIdx.x) * (kCutThreads / 64) + w;
  float sq = 0.f;
  if (item < N * H) {
    const int64_t row = item / H;
    const int h = static_cast<int>(item - row * H);
    const int64_t HD = static_cast<int64_t>(H) * Dc;
    const float inv = 1.0f / fmaxf(sqrtf(ssq[ssq_by_head ? static_cast<int64_t>(h) * N + row : item]), 1e-12f);
    const float* xr = xp + row * HD + static_cast<int64_t>(h) * Dc;
    const float* qr = embed + (static_cast<int64_t>(h) * K + ind[item]) * Dc;
    for (int c = 4 * lane; c < Dc; c += 256) {
      const float4 x = ld4(xr + c), q = ld4(qr + c);
      const float4 n = make_float4(x.x * inv, x.y * inv, x.z * inv, x.w * inv);
      const float4 d = make_float4(q.x - n.x, q.y - n.y, q.z - n.z, q.w - n.w);
      sq += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
      const float4 o = training ? make_float4(n.x + d.x, n.y + d.y, n.z + d.z, n.w + d.w) : q;
      st4(quant + row * HD + static_cast<int64_t>(h) * Dc + c, o);
      if (xn_out) st4(xn_out + row * HD + static_cast<int64_t>(h) * Dc + c, n);
    }
  }
  sq = wave_sum(sq);
  if (lane == 0) red[w] = static_cast<double>(sq);
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(kCutThreads)
k_bt_commit_finish(const double* __restrict__ partial, int64_t n, double scale, float* __restrict__ out) {
  __shared__ double red[kCutThreads];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += kCutThreads) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = kCutThreads / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = static_cast<float>(red[0] * scale);
}

// ---------------------------------------------------------------------------------------------------------------------
// Host side.  Scratch (the planes, the split slabs, the arg-max candidates) comes from the CALLER: one arena per
// (device, stream), registered with stemgnn_linear_set_scratch and sized with stemgnn_linear_scratch_bytes; nothing here
// allocates, frees or synchronises.  A product that qualifies but finds no arena (or one too small) runs on the tile
// kernels and is counted (stemgnn_linear_bigtile_fallbacks) -- a capacity miss, never an error swallowed.
// ---------------------------------------------------------------------------------------------------------------------
inline size_t a256(size_t b) { return (b + 255) / 256 * 256; }

// a set of straight planes of one operand, cut earlier in the current scope
struct PlaneSet {
  const void* src = nullptr;
  int64_t rows = 0, cols = 0, rows_padded = 0;
  int kind = 0, np = 0;
  uint16_t* p = nullptr;     // [np][rows_padded][cols]
  float* colsum = nullptr;   // [row_blocks][cols] partial column sums (the bias gradient), row_blocks = ceil(rows_padded / 256)
  float* inv_scale = nullptr;  // pair format (np == 2): [rows] factors that undo the row scaling
};
struct Arena {
  unsigned char* p = nullptr;
  size_t bytes = 0;
  size_t used = 0;   // bump pointer: reset by every product outside a scope, by the scope's end inside one
  int depth = 0;     // open scopes
  std::vector<PlaneSet> cache;
};
std::mutex g_arena_mu;
std::map<std::pair<int, hipStream_t>, Arena> g_arena;

Arena* arena_find(hipStream_t st) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  std::lock_guard<std::mutex> lock(g_arena_mu);
  auto it = g_arena.find({dev, st});
  return (it == g_arena.end() || !it->second.p) ? nullptr : &it->second;
}

// One product's (or, inside a scope, one phase's) claim on the stream's arena.  Kernels of a stream run one after the
// other, so a product may overwrite the previous product's planes -- unless a scope is open: then the planes cut so far
// stay (they are shared between the products of the phase) and new claims go behind them.
struct Lease {
  Arena* a;
  explicit Lease(hipStream_t st) : a(arena_find(st)) {
    if (a && a->depth == 0) { a->used = 0; a->cache.clear(); }
  }
  bool ok() const { return a != nullptr; }
  template <typename T> T* take(size_t bytes) {
    if (!a) return nullptr;
    const size_t start = a256(reinterpret_cast<uintptr_t>(a->p) + a->used) - reinterpret_cast<uintptr_t>(a->p);
    if (start + a256(bytes) > a->bytes) return nullptr;
    a->used = start + a256(bytes);
    return reinterpret_cast<T*>(a->p + start);
  }
};

inline unsigned cut_grid(int64_t elems4) {
  int64_t gsz = (elems4 + kCutThreads - 1) / kCutThreads;
  if (gsz > 16384) gsz = 16384;
  return static_cast<unsigned>(gsz < 1 ? 1 : gsz);
}

// Optional in-situ timing of the core's launches (bench.py's matrix-roofline leg): the launch goes through
// hipExtLaunchKernelGGL with a start / stop event pair that stamps the kernel's own begin and end, like K1's
// (csrc/sage_agg.hip); each record carries the matrix work the launch executed (2 x 256^2 x 64 per K tile and block tile).
struct BtProfile {
  std::mutex mu;
  bool enabled = false;
  struct Rec { hipEvent_t e0, e1; double flop; };
  std::vector<Rec> recs;
};
BtProfile g_bt_profile;

int bt_launch(BtArgs& a, int epi, hipStream_t st, bool tn = false, bool f16 = false) {
  a.tiles_m = static_cast<int>((a.a.rows + kTile - 1) / kTile);
  a.tiles_n = static_cast<int>((a.b.rows + kTile - 1) / kTile);
  const int64_t work = static_cast<int64_t>(a.tiles_m) * a.tiles_n * a.splits * a.batch;
  static const int cus = [] {
    int n = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
      (void)hipGetLastError();
      n = 256;
    }
    return n;
  }();
  const int64_t blocks = std::min<int64_t>(work, cus);  // one persistent block per CU (128 KiB of LDS each)
  a.work = static_cast<int>(work);
  if (work <= 0 || work >= (1ll << 31) || (!tn && (a.a.rows * a.a.ld >= (1ll << 31) || a.b.rows * a.b.ld >= (1ll << 31))))
    return STEMGNN_ERR_TOO_LARGE;
#define BT_ATTR(K) hipFuncSetAttribute(reinterpret_cast<const void*>(K), hipFuncAttributeMaxDynamicSharedMemorySize, kBtLds)
  static const bool attrs_ok = BT_ATTR((k_bt_gemm<kEpiStore, false, false>)) == hipSuccess &&
                               BT_ATTR((k_bt_gemm<kEpiArgmax, false, false>)) == hipSuccess &&
                               BT_ATTR((k_bt_gemm<kEpiStore, true, false>)) == hipSuccess &&
                               BT_ATTR((k_bt_gemm<kEpiStore, false, true>)) == hipSuccess &&
                               BT_ATTR((k_bt_gemm<kEpiArgmax, false, true>)) == hipSuccess;
#undef BT_ATTR
  if (!attrs_ok) return STEMGNN_ERR_HIP;
  static const int dbg = getenv("STEMGNN_BT_DBG") ? atoi(getenv("STEMGNN_BT_DBG")) : 0;
  a.dbg = dbg;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_bt_profile.mu);
    if (g_bt_profile.enabled) {
      STEMGNN_HIP_TRY(hipEventCreate(&ev0));
      STEMGNN_HIP_TRY(hipEventCreate(&ev1));
      // executed multiply-adds x 2 over the operands' real extents (tile padding not counted)
      double ktiles = 0.0;
      for (int sgi = 0; sgi < a.nseg; ++sgi) ktiles += a.seg[sgi].kt;
      g_bt_profile.recs.push_back({ev0, ev1, 2.0 * a.a.rows * a.b.rows * ktiles * kBK * a.batch});
    }
  }
  const dim3 grid(static_cast<unsigned>(blocks)), block(kBtThreads);
  if (tn && f16) return STEMGNN_ERR_INVALID_ARG;  // the pair format is scaled per row: no contraction over rows
  if (tn) hipExtLaunchKernelGGL((k_bt_gemm<kEpiStore, true, false>), grid, block, kBtLds, st, ev0, ev1, 0, a);
  else if (epi == kEpiStore && f16) hipExtLaunchKernelGGL((k_bt_gemm<kEpiStore, false, true>), grid, block, kBtLds, st, ev0, ev1, 0, a);
  else if (epi == kEpiStore) hipExtLaunchKernelGGL((k_bt_gemm<kEpiStore, false, false>), grid, block, kBtLds, st, ev0, ev1, 0, a);
  else if (f16) hipExtLaunchKernelGGL((k_bt_gemm<kEpiArgmax, false, true>), grid, block, kBtLds, st, ev0, ev1, 0, a);
  else hipExtLaunchKernelGGL((k_bt_gemm<kEpiArgmax, false, false>), grid, block, kBtLds, st, ev0, ev1, 0, a);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

// the segment list of a product: exact mode = the six significant piece products, small terms first (as common.h:
// mfma_x3); kt_a / kt_b: K tiles of the planes of the a / b side that exist (a bf16-stored part has an h plane only)
inline void exact_segments(BtArgs& g, int pieces, int kt_all, int kt_b_lo /*K tiles the b side's m / l planes cover*/,
                           int z_k0, int z_kt_all, int z_kt_lo) {
  static const int order[6][2] = {{2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};  // (a piece, b piece)
  g.nseg = 0;
  if (pieces == 1) {
    g.seg[g.nseg++] = BtSeg{0, 0, 0, kt_all, z_k0, z_kt_all};
    return;
  }
  for (int s = 0; s < 6; ++s) {
    const int pa = order[s][0], pb = order[s][1];
    const bool lo = pb != 0;  // the b side's m / l planes
    g.seg[g.nseg++] = BtSeg{pa, pb, 0, lo ? kt_b_lo : kt_all, z_k0, lo ? z_kt_lo : z_kt_all};
  }
}

std::atomic<int> g_bt_on{1};
// exact mode: forward / backward-data / code assignment from two fp16 pieces (0: three bf16 pieces); STEMGNN_LINEAR_PAIR=0
// in the environment starts a process with the bf16 pieces (same-box A/B runs of the bench)
std::atomic<int> g_bt_pair{[] {
  const char* e = std::getenv("STEMGNN_LINEAR_PAIR");
  return (e && e[0] == '0') ? 0 : 1;
}()};
std::atomic<int64_t> g_bt_calls{0}, g_bt_fallbacks{0};

}  // namespace

// Products the big-tile core takes: large (>= 8 192 rows, >= 1e10 flop), both feature extents at least one 256 tile and
// multiples of 64.  Below that the 128-row tile / weight-stationary kernels win (HBM-bound shapes, D = 128).
bool bt_gemm_ok(int64_t M, int64_t N, int64_t K) {
  // (operand planes are addressed with 32-bit element offsets: rows x row length below 2^31, padding included)
  return g_bt_on.load(std::memory_order_relaxed) != 0 && M >= 8192 && N >= 256 && K >= 256 && N % 64 == 0 && K % 64 == 0 &&
         2.0 * static_cast<double>(M) * N * K >= 1.0e10 && (M + 8192) * std::max(N, K) < (1ll << 31);
}
void bt_served() { g_bt_calls.fetch_add(1, std::memory_order_relaxed); }
void bt_missed() { g_bt_fallbacks.fetch_add(1, std::memory_order_relaxed); }

bool linear_pair_on() { return g_bt_pair.load(std::memory_order_relaxed) != 0; }
static inline int np_of(int pieces) { return pieces == 1 ? 1 : 3; }
static inline bool use_pair(int pieces) { return pieces == 3 && g_bt_pair.load(std::memory_order_relaxed) != 0; }
// the pair format's three products, small terms first: (lo, hi), (hi, lo), (hi, hi); plane 0 = hi, 1 = lo
static inline void pair_segments(BtArgs& g, int kt_all, int z_k0, int z_kt) {
  static const int order[3][2] = {{1, 0}, {0, 1}, {0, 0}};
  g.nseg = 0;
  for (int s = 0; s < 3; ++s) g.seg[g.nseg++] = BtSeg{order[s][0], order[s][1], 0, kt_all, z_k0, z_kt};
}

// contraction splits of a weight gradient: about one block per CU over the output tiles
static inline int bt_dw_splits(int64_t N, int64_t K) {
  const int64_t tiles = ((N + kTile - 1) / kTile) * ((K + kTile - 1) / kTile);
  int64_t s = 256 / (tiles < 1 ? 1 : tiles);
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  return static_cast<int>(s);
}
static inline int64_t bt_rows_padded(int64_t M) { return (M + kBK - 1) / kBK * kBK; }  // whole K tiles of rows
// arena bytes of the three products and of the code assignment (what the bt_* claim; the scratch queries take their maximum)
static inline size_t bt_need_planes(int np, int64_t M, int64_t C) {
  const int64_t Mp = bt_rows_padded(M);
  return a256(static_cast<size_t>(np) * Mp * C * 2) + a256(static_cast<size_t>((Mp + 255) / 256) * C * 4);
}
static inline size_t bt_need_fwd(int np, int64_t M, int64_t N, int64_t K) {
  // (+ the pair format's transposed fp32 copy of the weight and the two scale vectors)
  return bt_need_planes(np, M, K) + a256(static_cast<size_t>(np) * N * K * 2) + a256(static_cast<size_t>(N) * K * 4) +
         a256(static_cast<size_t>(M) * 4) + a256(static_cast<size_t>(std::max(N, K)) * 4) + 1024;
}
static inline size_t bt_need_bwd_weight(int np, int64_t M, int64_t N, int64_t K) {
  return bt_need_planes(np, M, N) + bt_need_planes(np, M, K) + a256(static_cast<size_t>(bt_dw_splits(N, K)) * N * K * 4);
}
static inline size_t bt_need_vq(int64_t N, int64_t H, int64_t Dc, int64_t K) {
  const int64_t tiles = (K + kTile - 1) / kTile;
  const int64_t fin_blocks = (N * H + kCutThreads - 1) / kCutThreads;
  const int64_t gat_blocks = (N * H + kCutThreads / 64 - 1) / (kCutThreads / 64);
  return a256(static_cast<size_t>(3) * H * K * Dc * 2) + a256(static_cast<size_t>(3) * N * H * Dc * 2) +
         2 * a256(static_cast<size_t>(N) * H * tiles * 4) + 2 * a256(static_cast<size_t>(N) * H * 4) +
         a256(static_cast<size_t>(H) * K * 4) + a256(static_cast<size_t>(fin_blocks + gat_blocks) * 8);
}

// The straight planes of an operand x [M, C] (rows padded with zeros to whole K tiles) and the partial column sums the
// cut pass takes along: from the scope's cache when this operand has been cut in the scope, else cut now into the lease.
static int bt_planes(Lease& lease, const void* x, int kind, int64_t M, int64_t C, int np, hipStream_t st, PlaneSet* out) {
  if (!lease.ok()) return STEMGNN_ERR_WORKSPACE;
  if (lease.a->depth > 0)
    for (const PlaneSet& e : lease.a->cache)
      if (e.src == x && e.rows == M && e.cols == C && e.kind == kind && e.np == np) { *out = e; return STEMGNN_OK; }
  PlaneSet ps;
  ps.src = x; ps.rows = M; ps.cols = C; ps.kind = kind; ps.np = np;
  ps.rows_padded = bt_rows_padded(M);
  const int row_blocks = static_cast<int>((ps.rows_padded + 255) / 256);
  ps.p = lease.take<uint16_t>(static_cast<size_t>(np) * ps.rows_padded * C * 2);
  if (np == 2) {  // pair format: two fp16 planes of the row-scaled operand + the factors that undo the scaling
    ps.inv_scale = lease.take<float>(static_cast<size_t>(M) * 4);
    if (!ps.p || !ps.inv_scale) return STEMGNN_ERR_WORKSPACE;
    { const int rc_cut = bt_cut_pair(st, kind == kF32 ? static_cast<const float*>(x) : nullptr, C, kind == kF32 ? static_cast<int>(C) : 0, M, kind == kF32 ? nullptr : x, kind, kind == kF32 ? 0 : static_cast<int>(C), M, static_cast<int>(C), ps.p, ps.rows_padded * C, C, ps.inv_scale, nullptr); if (rc_cut != STEMGNN_OK) return rc_cut; }
    if (lease.a->depth > 0) lease.a->cache.push_back(ps);
    *out = ps;
    return STEMGNN_OK;
  }
  ps.colsum = lease.take<float>(static_cast<size_t>(row_blocks) * C * 4);
  if (!ps.p || !ps.colsum) return STEMGNN_ERR_WORKSPACE;
  dim3 grid(static_cast<unsigned>(row_blocks), static_cast<unsigned>((C / 4 + 63) / 64));
  if (np == 3)
    k_bt_cut_rows_colsum<3><<<grid, kCutThreads, 0, st>>>(x, kind, M, ps.rows_padded, static_cast<int>(C), ps.p,
                                                         ps.rows_padded * C, ps.colsum);
  else
    k_bt_cut_rows_colsum<1><<<grid, kCutThreads, 0, st>>>(x, kind, M, ps.rows_padded, static_cast<int>(C), ps.p,
                                                         ps.rows_padded * C, ps.colsum);
  STEMGNN_LAUNCH_CHECK();
  if (lease.a->depth > 0) lease.a->cache.push_back(ps);
  *out = ps;
  return STEMGNN_OK;
}

// y [M, N] = [x1 | x2] [w1 | w2]^T + bias.  pieces: 3 = exact mode, 1 = bf16 GEMM mode.  Returns STEMGNN_ERR_WORKSPACE
// when the arena is missing or too small (the caller takes the tile kernels).
int bt_linear_fwd(int pieces, const float* x1, const float* w1, int64_t K1, const void* x2, int x2_kind, const float* w2,
                  int64_t K2, const float* bias, int64_t M, int64_t N, float* y, int64_t x1_rows, int64_t store_rows,
                  float* stats_partial, int64_t stats_slabs, hipStream_t st) {
  const int np = np_of(pieces);
  const int64_t K = K1 + K2;
  if (K1 % kBK != 0 || K2 % kBK != 0) return STEMGNN_ERR_WORKSPACE;
  if (use_pair(pieces)) {
    // exact mode, pair format: both operands as two fp16 planes of their row-scaled rows, three matrix passes
    Lease lease(st);
    uint16_t* xpl = lease.take<uint16_t>(static_cast<size_t>(2) * M * K * 2);
    uint16_t* wpl = lease.take<uint16_t>(static_cast<size_t>(2) * N * K * 2);
    float* xsc = lease.take<float>(static_cast<size_t>(M) * 4);
    float* wsc = lease.take<float>(static_cast<size_t>(N) * 4);
    if (!xpl || !wpl || !xsc || !wsc) return STEMGNN_ERR_WORKSPACE;
    { const int rc_cut = bt_cut_pair(st, x1, K1, static_cast<int>(K1), x1_rows, x2, x2_kind, static_cast<int>(K2), M, static_cast<int>(K), xpl, M * K, K, xsc, nullptr); if (rc_cut != STEMGNN_OK) return rc_cut; }
    { const int rc_cut = bt_cut_pair(st, w1, K1, static_cast<int>(K1), N, w2, kF32, static_cast<int>(K2), N, static_cast<int>(K), wpl, N * K, K, wsc, nullptr); if (rc_cut != STEMGNN_OK) return rc_cut; }
    BtArgs g{};
    g.a = BtOp{wpl, N * K, K, N, 0};
    g.b = BtOp{xpl, M * K, K, M, 0};
    const int kt1 = static_cast<int>(K1 / kBK), kt2 = static_cast<int>(K2 / kBK);
    pair_segments(g, kt1 + kt2, kt1, kt2);
    g.splits = 1; g.batch = 1;
    g.b_full_rows = (K2 > 0 && x1_rows < M) ? (x1_rows + kTile - 1) / kTile * kTile : (1ll << 62);
    g.y = y; g.ldy = N; g.store_rows = store_rows;
    g.bias = bias;
    g.stats = stats_partial;
    g.a_scale = wsc; g.b_scale = xsc;
    const int rc = bt_launch(g, kEpiStore, st, false, true);
    if (rc != STEMGNN_OK) return rc;
    if (stats_partial && stats_slabs > g.tiles_n)
      STEMGNN_HIP_TRY(hipMemsetAsync(stats_partial + static_cast<int64_t>(g.tiles_n) * 2 * N, 0,
                                     sizeof(float) * (stats_slabs - g.tiles_n) * 2 * N, st));
    return STEMGNN_OK;
  }
  const size_t xb = static_cast<size_t>(np) * M * K * 2, wb = static_cast<size_t>(np) * N * K * 2;
  Lease lease(st);
  uint16_t* xpl = lease.take<uint16_t>(xb);
  uint16_t* wpl = lease.take<uint16_t>(wb);
  if (!xpl || !wpl) return STEMGNN_ERR_WORKSPACE;
  if (np == 3) {
    k_bt_cut_rows<3><<<cut_grid(M * (K / 4)), kCutThreads, 0, st>>>(x1, K1, static_cast<int>(K1), x1_rows, x2, x2_kind,
                                                                   static_cast<int>(K2), M, xpl, M * K, K);
    STEMGNN_LAUNCH_CHECK();
    k_bt_cut_rows<3><<<cut_grid(N * (K / 4)), kCutThreads, 0, st>>>(w1, K1, static_cast<int>(K1), N, w2, kF32,
                                                                   static_cast<int>(K2), N, wpl, N * K, K);
  } else {
    k_bt_cut_rows<1><<<cut_grid(M * (K / 4)), kCutThreads, 0, st>>>(x1, K1, static_cast<int>(K1), x1_rows, x2, x2_kind,
                                                                   static_cast<int>(K2), M, xpl, M * K, K);
    STEMGNN_LAUNCH_CHECK();
    k_bt_cut_rows<1><<<cut_grid(N * (K / 4)), kCutThreads, 0, st>>>(w1, K1, static_cast<int>(K1), N, w2, kF32,
                                                                   static_cast<int>(K2), N, wpl, N * K, K);
  }
  STEMGNN_LAUNCH_CHECK();
  BtArgs g{};
  g.a = BtOp{wpl, N * K, K, N, 0};
  g.b = BtOp{xpl, M * K, K, M, 0};
  const int kt1 = static_cast<int>(K1 / kBK), kt2 = static_cast<int>(K2 / kBK);
  const bool bf = K2 > 0 && x2_kind == kBF16;  // the second part of the b side has an h plane only
  exact_segments(g, pieces, kt1 + kt2, bf ? kt1 : kt1 + kt2, kt1, kt2, bf ? 0 : kt2);
  g.splits = 1; g.batch = 1;
  // row tiles that start at or past the aggregate's rows contract over the second part only
  g.b_full_rows = (K2 > 0 && x1_rows < M) ? (x1_rows + kTile - 1) / kTile * kTile : (1ll << 62);
  g.y = y; g.ldy = N; g.y_bs = 0; g.y_ss = 0; g.store_rows = store_rows;
  g.bias = bias;
  g.stats = stats_partial;
  const int rc = bt_launch(g, kEpiStore, st);
  if (rc != STEMGNN_OK) return rc;
  if (stats_partial && stats_slabs > g.tiles_n)  // the slabs the consumer reduces over past the ones written here
    STEMGNN_HIP_TRY(hipMemsetAsync(stats_partial + static_cast<int64_t>(g.tiles_n) * 2 * N, 0,
                                   sizeof(float) * (stats_slabs - g.tiles_n) * 2 * N, st));
  return STEMGNN_OK;
}

// dx [M, K] = dy [M, N] w [N, K]
int bt_linear_bwd_data(int pieces, const float* dy, const float* w, int64_t M, int64_t N, int64_t K, float* dx,
                       hipStream_t st) {
  const int np = np_of(pieces);
  if (N % kBK != 0) return STEMGNN_ERR_WORKSPACE;
  Lease lease(st);
  PlaneSet gp;
  if (use_pair(pieces)) {
    // pair format: dy's rows scaled per row; the weight's COLUMNS are the contraction rows of w^T: transpose, then cut
    int rc = bt_planes(lease, dy, kF32, M, N, 2, st, &gp);
    if (rc != STEMGNN_OK) return rc;
    float* wt = lease.take<float>(static_cast<size_t>(K) * N * 4);
    uint16_t* wpl = lease.take<uint16_t>(static_cast<size_t>(2) * K * N * 2);
    float* wsc = lease.take<float>(static_cast<size_t>(K) * 4);
    if (!wt || !wpl || !wsc) return STEMGNN_ERR_WORKSPACE;
    rc = stemgnn_transpose(w, N, K, wt, st);
    if (rc != STEMGNN_OK) return rc;
    { const int rc_cut = bt_cut_pair(st, wt, N, static_cast<int>(N), K, nullptr, kF32, 0, K, static_cast<int>(N), wpl, K * N, N, wsc, nullptr); if (rc_cut != STEMGNN_OK) return rc_cut; }
    BtArgs g{};
    g.a = BtOp{wpl, K * N, N, K, 0};
    g.b = BtOp{gp.p, gp.rows_padded * N, N, M, 0};
    pair_segments(g, static_cast<int>(N / kBK), 0, static_cast<int>(N / kBK));
    g.splits = 1; g.batch = 1;
    g.b_full_rows = 1ll << 62;
    g.y = dx; g.ldy = K; g.store_rows = M;
    g.a_scale = wsc; g.b_scale = gp.inv_scale;
    return bt_launch(g, kEpiStore, st, false, true);
  }
  int rc = bt_planes(lease, dy, kF32, M, N, np, st, &gp);
  if (rc != STEMGNN_OK) return rc;
  uint16_t* wpl = lease.take<uint16_t>(static_cast<size_t>(np) * K * N * 2);  // w^T planes [K][N]
  if (!wpl) return STEMGNN_ERR_WORKSPACE;
  dim3 tg(static_cast<unsigned>((N + 255) / 256), static_cast<unsigned>((K + 63) / 64));
  if (np == 3) k_bt_cut_cols<3><<<tg, kCutThreads, 0, st>>>(w, kF32, N, static_cast<int>(K), wpl, K * N, N, nullptr);
  else k_bt_cut_cols<1><<<tg, kCutThreads, 0, st>>>(w, kF32, N, static_cast<int>(K), wpl, K * N, N, nullptr);
  STEMGNN_LAUNCH_CHECK();
  BtArgs g{};
  g.a = BtOp{wpl, K * N, N, K, 0};
  g.b = BtOp{gp.p, gp.rows_padded * N, N, M, 0};
  const int kt = static_cast<int>(N / kBK);
  exact_segments(g, pieces, kt, kt, 0, kt, kt);
  g.splits = 1; g.batch = 1;
  g.b_full_rows = 1ll << 62;
  g.y = dx; g.ldy = K; g.store_rows = M;
  return bt_launch(g, kEpiStore, st);
}

// dw [N, K] = dy [M, N]^T x [M, K]; db [N] = column sums of dy (NULL: skip).  The contraction runs over the operands'
// rows: the TN form of the core on STRAIGHT planes (the same a backward-data product reads), split over the rows into
// about one block per CU, the split slabs added in fixed order.
int bt_linear_bwd_weight(int pieces, const float* dy, const void* x, int x_kind, int64_t M, int64_t N, int64_t K, float* dw,
                         float* db, hipStream_t st) {
  const int np = np_of(pieces);
  const int S = bt_dw_splits(N, K);
  const int64_t Mp = bt_rows_padded(M);
  if (Mp * std::max(N, K) >= (1ll << 31)) return STEMGNN_ERR_WORKSPACE;  // 32-bit plane offsets: the tile kernels serve
  Lease lease(st);
  PlaneSet gp, xp;
  int rc = bt_planes(lease, dy, kF32, M, N, np, st, &gp);
  if (rc != STEMGNN_OK) return rc;
  rc = bt_planes(lease, x, x_kind, M, K, np, st, &xp);
  if (rc != STEMGNN_OK) return rc;
  float* slabs = S > 1 ? lease.take<float>(static_cast<size_t>(S) * N * K * 4) : nullptr;
  if (S > 1 && !slabs) return STEMGNN_ERR_WORKSPACE;
  const bool xbf = x_kind == kBF16;
  BtArgs g{};
  g.a = BtOp{xp.p, Mp * K, K, K, 0};   // C rows i = the features of x
  g.b = BtOp{gp.p, Mp * N, N, N, 0};   // C columns j = the features of dy
  const int kt = static_cast<int>(Mp / kBK);
  g.nseg = 0;
  if (pieces == 1) {
    g.seg[g.nseg++] = BtSeg{0, 0, 0, kt, 0, kt};
  } else {
    static const int order[6][2] = {{2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};  // (x piece, dy piece)
    for (int s = 0; s < 6; ++s) {
      if (xbf && order[s][0] != 0) continue;  // a bf16-stored x is its own h piece: its m and l planes are zero
      g.seg[g.nseg++] = BtSeg{order[s][0], order[s][1], 0, kt, 0, kt};
    }
  }
  g.splits = S; g.batch = 1;
  g.b_full_rows = 1ll << 62;
  g.y = S > 1 ? slabs : dw; g.ldy = K; g.y_ss = N * K; g.store_rows = N;
  rc = bt_launch(g, kEpiStore, st, true);
  if (rc != STEMGNN_OK) return rc;
  if (S > 1) {
    k_bt_reduce_slabs<<<static_cast<unsigned>((N * K / 4 + kCutThreads - 1) / kCutThreads), kCutThreads, 0, st>>>(
        slabs, S, N * K, dw);
    STEMGNN_LAUNCH_CHECK();
  }
  if (db) {
    k_bt_colsum_finish<<<static_cast<unsigned>((N + 15) / 16), kCutThreads, 0, st>>>(
        gp.colsum, static_cast<int>((Mp + 255) / 256), static_cast<int>(N), db);
    STEMGNN_LAUNCH_CHECK();
  }
  return STEMGNN_OK;
}

// Scopes: between bt_scope_begin and bt_scope_end (a phase entry point: encoder / quantiser / heads backward) the straight
// planes of an operand are cut ONCE and shared by every product of the phase that reads it (the pre-activation gradient
// of a layer feeds two backward-data products and two weight gradients).  The caller promises that an operand's contents
// do not change inside the scope.  Without an arena a scope is a no-op.
void bt_scope_begin(hipStream_t st) {
  Arena* a = arena_find(st);
  if (!a) return;
  if (a->depth++ == 0) { a->used = 0; a->cache.clear(); }
}
void bt_scope_end(hipStream_t st) {
  Arena* a = arena_find(st);
  if (!a || a->depth == 0) return;
  if (--a->depth == 0) { a->used = 0; a->cache.clear(); }
}

// The quantiser's code assignment at large codebooks (K >= 512 codes of Dc >= 256; BASELINE configs 3 and 5; reference
// model/vq.py:650-657): the exact six-piece similarity product of ALL heads as one launch of the core, the arg-max
// taken from the accumulators per 256-code tile (no [N, K] matrix: N x H x K / 256 candidates), then one finishing pass.
bool bt_vq_assign_ok(int64_t N, int64_t H, int64_t Dc, int64_t K) {
  return g_bt_on.load(std::memory_order_relaxed) != 0 && N >= 8192 && H >= 1 && K >= 512 && Dc >= 256 && Dc % kBK == 0 &&
         N * H * Dc < (1ll << 31) && H * K * Dc < (1ll << 31) && H <= 1024;
}
int bt_vq_assign(const float* xp, int64_t N, int64_t H, int64_t Dc, const float* embed, const float* esq, int64_t K,
                 int training, float* xn, float* norm, int64_t* ind, float* quant, float* sqerr, double sq_scale,
                 hipStream_t st) {
  if (!quant && !esq) return STEMGNN_ERR_INVALID_ARG;
  const int64_t HD = H * Dc;
  const int tiles = static_cast<int>((K + kTile - 1) / kTile);
  const int64_t fin_blocks = (N * H + kCutThreads - 1) / kCutThreads;
  const int64_t gat_blocks = (N * H + kCutThreads / 64 - 1) / (kCutThreads / 64);
  const size_t ab = static_cast<size_t>(3) * H * K * Dc * 2, bb = static_cast<size_t>(3) * N * HD * 2;
  const size_t cb = static_cast<size_t>(N) * H * tiles * 4, qb = static_cast<size_t>(N) * H * 4;
  const size_t pb = static_cast<size_t>(fin_blocks + gat_blocks) * 8;
  Lease lease(st);
  uint16_t* epl = lease.take<uint16_t>(ab);
  uint16_t* xpl = lease.take<uint16_t>(bb);
  float* cval = lease.take<float>(cb);
  int32_t* cidx = lease.take<int32_t>(cb);
  float* ssq = lease.take<float>(qb);
  double* partial = lease.take<double>(pb);
  float* esc = lease.take<float>(static_cast<size_t>(H) * K * 4);
  float* xsc = lease.take<float>(qb);
  if (!epl || !xpl || !cval || !cidx || !ssq || !partial || !esc || !xsc) return STEMGNN_ERR_WORKSPACE;
  if (g_bt_pair.load(std::memory_order_relaxed) != 0) {
    // pair format (the reference forces fp32 in this product, vq.py:623,634: the pair product IS fp32-accurate): codes
    // scaled per code, rows per (row, head); the arg-max compares C[i][j] * scale(code i) -- the row's own factor is
    // positive and common to its candidates --, the candidate that leaves is the true similarity
    { const int rc_cut = bt_cut_pair(st, embed, Dc, static_cast<int>(Dc), H * K, nullptr, kF32, 0, H * K, static_cast<int>(Dc), epl, H * K * Dc, Dc, esc, nullptr); if (rc_cut != STEMGNN_OK) return rc_cut; }
    { const int rc_cut = bt_cut_pair(st, xp, HD, static_cast<int>(HD), N, nullptr, kF32, 0, N, static_cast<int>(Dc), xpl, N * HD, HD, xsc, ssq); if (rc_cut != STEMGNN_OK) return rc_cut; }
    BtArgs g{};
    g.a = BtOp{epl, H * K * Dc, Dc, K, K * Dc};
    g.b = BtOp{xpl, N * HD, HD, N, Dc};
    pair_segments(g, static_cast<int>(Dc / kBK), 0, static_cast<int>(Dc / kBK));
    g.splits = 1; g.batch = static_cast<int>(H);
    g.b_full_rows = 1ll << 62;
    g.cand_val = cval; g.cand_idx = cidx;
    g.a_scale = esc; g.b_scale = xsc;  // [h * K + i], [h * N + j]
    int rc = bt_launch(g, kEpiArgmax, st, false, true);
    if (rc != STEMGNN_OK) return rc;
    k_bt_argmax_finish<<<static_cast<unsigned>(fin_blocks), kCutThreads, 0, st>>>(
        cval, cidx, ssq, quant ? nullptr : esq, N, static_cast<int>(H), static_cast<int>(K), tiles, norm, ind, partial, 1);
    STEMGNN_LAUNCH_CHECK();
    int64_t nparts = fin_blocks;
    if (quant) {
      k_bt_gather_commit<<<static_cast<unsigned>(gat_blocks), kCutThreads, 0, st>>>(
          xp, N, static_cast<int>(H), static_cast<int>(Dc), static_cast<int>(K), embed, ind, ssq, training, quant, xn,
          partial + fin_blocks, 1);
      STEMGNN_LAUNCH_CHECK();
      nparts += gat_blocks;
    }
    k_bt_commit_finish<<<1, kCutThreads, 0, st>>>(partial, nparts, sq_scale, sqerr);
    STEMGNN_LAUNCH_CHECK();
    return STEMGNN_OK;
  }
  k_bt_cut_rows<3><<<cut_grid(H * K * (Dc / 4)), kCutThreads, 0, st>>>(embed, Dc, static_cast<int>(Dc), H * K, nullptr, kF32,
                                                                      0, H * K, epl, H * K * Dc, Dc);
  STEMGNN_LAUNCH_CHECK();
  k_bt_cut_rows_ssq<<<static_cast<unsigned>((N + 3) / 4), kCutThreads, 0, st>>>(xp, N, static_cast<int>(H),
                                                                               static_cast<int>(Dc), xpl, N * HD, ssq);
  STEMGNN_LAUNCH_CHECK();
  BtArgs g{};
  g.a = BtOp{epl, H * K * Dc, Dc, K, K * Dc};
  g.b = BtOp{xpl, N * HD, HD, N, Dc};
  const int kt = static_cast<int>(Dc / kBK);
  exact_segments(g, 3, kt, kt, 0, kt, kt);
  g.splits = 1; g.batch = static_cast<int>(H);
  g.b_full_rows = 1ll << 62;
  g.cand_val = cval; g.cand_idx = cidx;
  int rc = bt_launch(g, kEpiArgmax, st);
  if (rc != STEMGNN_OK) return rc;
  k_bt_argmax_finish<<<static_cast<unsigned>(fin_blocks), kCutThreads, 0, st>>>(
      cval, cidx, ssq, quant ? nullptr : esq, N, static_cast<int>(H), static_cast<int>(K), tiles, norm, ind, partial, 0);
  STEMGNN_LAUNCH_CHECK();
  int64_t nparts = fin_blocks;
  if (quant) {
    k_bt_gather_commit<<<static_cast<unsigned>(gat_blocks), kCutThreads, 0, st>>>(
        xp, N, static_cast<int>(H), static_cast<int>(Dc), static_cast<int>(K), embed, ind, ssq, training, quant, xn,
        partial + fin_blocks, 0);
    STEMGNN_LAUNCH_CHECK();
    nparts += gat_blocks;
  }
  k_bt_commit_finish<<<1, kCutThreads, 0, st>>>(partial, nparts, sq_scale, sqerr);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

}  // namespace stemgnn

using namespace stemgnn;

extern "C" {

size_t stemgnn_linear_scratch_bytes(int64_t max_rows, int64_t dim_a, int64_t dim_b) {
  if (max_rows <= 0 || dim_a <= 0 || dim_b <= 0) return 0;
  size_t need = 0;
  for (int o = 0; o < 2; ++o) {
    const int64_t N = o ? dim_b : dim_a, K = o ? dim_a : dim_b;
    need = std::max(need, bt_need_fwd(3, max_rows, N, K));
    need = std::max(need, bt_need_bwd_weight(3, max_rows, N, K));
  }
  return need + 4096;
}

size_t stemgnn_vq_assign_scratch_bytes(int64_t num_rows, int64_t heads, int64_t code_dim, int64_t codebook_size) {
  if (num_rows <= 0 || heads <= 0 || code_dim <= 0 || codebook_size <= 0) return 0;
  return bt_need_vq(num_rows, heads, code_dim, codebook_size) + 1024;
}

int stemgnn_linear_set_scratch(void* scratch, size_t bytes, void* stream_) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return hip_fail(hipGetLastError());
  std::lock_guard<std::mutex> lock(g_arena_mu);
  Arena& a = g_arena[{dev, static_cast<hipStream_t>(stream_)}];
  a.p = static_cast<unsigned char*>(scratch);
  a.bytes = scratch ? bytes : 0;
  a.used = 0;
  a.cache.clear();  // planes cached in an open scope lived in the old block
  return STEMGNN_OK;
}

int stemgnn_profile_bigtile(int enable) {
  std::lock_guard<std::mutex> lock(g_bt_profile.mu);
  g_bt_profile.enabled = enable != 0;
  return STEMGNN_OK;
}

int stemgnn_profile_bigtile_collect(double* total_ms_host, double* total_flop_host, int64_t* launches_host) {
  std::vector<BtProfile::Rec> recs;
  {
    std::lock_guard<std::mutex> lock(g_bt_profile.mu);
    recs.swap(g_bt_profile.recs);
  }
  double ms_sum = 0.0, flop = 0.0;
  for (auto& r : recs) {
    STEMGNN_HIP_TRY(hipEventSynchronize(r.e1));
    float ms = 0.f;
    STEMGNN_HIP_TRY(hipEventElapsedTime(&ms, r.e0, r.e1));
    ms_sum += ms;
    flop += r.flop;
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  if (total_ms_host) *total_ms_host = ms_sum;
  if (total_flop_host) *total_flop_host = flop;
  if (launches_host) *launches_host = static_cast<int64_t>(recs.size());
  return STEMGNN_OK;
}

int stemgnn_linear_set_pair(int on) {
  const int prev = g_bt_pair.load(std::memory_order_relaxed);
  if (on == 0 || on == 1) g_bt_pair.store(on, std::memory_order_relaxed);
  return prev;
}

int stemgnn_linear_set_bigtile(int on) {
  const int prev = g_bt_on.load(std::memory_order_relaxed);
  if (on == 0 || on == 1) g_bt_on.store(on, std::memory_order_relaxed);
  return prev;
}
int64_t stemgnn_linear_bigtile_calls(void) { return g_bt_calls.load(std::memory_order_relaxed); }
int64_t stemgnn_linear_bigtile_fallbacks(void) { return g_bt_fallbacks.load(std::memory_order_relaxed); }

}  // extern "C"
