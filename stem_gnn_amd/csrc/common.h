// Shared helpers for the gfx950 STEM-GNN kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/stemgnn.h"

namespace stemgnn {

constexpr int kWave = 64;  // CDNA wavefront width

extern thread_local int g_last_hip_error;

inline int hip_fail(hipError_t e) {
  g_last_hip_error = static_cast<int>(e);
  return STEMGNN_ERR_HIP;
}

#define STEMGNN_HIP_TRY(expr)                                  \
  do {                                                         \
    hipError_t e__ = (expr);                                   \
    if (e__ != hipSuccess) return ::stemgnn::hip_fail(e__);    \
  } while (0)

#define STEMGNN_LAUNCH_CHECK() STEMGNN_HIP_TRY(hipGetLastError())

inline bool fits_i32(int64_t v) { return v >= 0 && v < 2147483647LL; }

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- Philox4x32-10 counter RNG: keep mask of dropout is a pure function of
// (seed, offset, element index) so forward, backward and the test mask kernel agree.
struct Philox {
  static __host__ __device__ inline uint32_t mulhi(uint32_t a, uint32_t b) {
    return static_cast<uint32_t>((static_cast<uint64_t>(a) * b) >> 32);
  }
  static __host__ __device__ inline void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0 = mulhi(M0, c[0]), lo0 = M0 * c[0];
    uint32_t hi1 = mulhi(M1, c[2]), lo1 = M1 * c[2];
    uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
  }
  // 4 x uint32 for counter block `blk` (= element index / 4).
  static __host__ __device__ inline void gen(uint64_t seed, uint64_t offset, uint64_t blk, uint32_t (&out)[4]) {
    uint32_t c[4] = {static_cast<uint32_t>(blk), static_cast<uint32_t>(blk >> 32),
                     static_cast<uint32_t>(offset), static_cast<uint32_t>(offset >> 32)};
    uint32_t k0 = static_cast<uint32_t>(seed), k1 = static_cast<uint32_t>(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      round(c, k0, k1);
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
  }
  static __host__ __device__ inline float to_unit(uint32_t u) { return static_cast<float>(u >> 8) * (1.0f / 16777216.0f); }
};

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---- fp32 products on the bf16 matrix cores (the default path) --------------------------------
// An fp32 value is cut EXACTLY into three bf16 pieces a = h + m + l: three 8-bit slices of the 24-bit
// significand, by truncation, so every piece carries the sign of a and each remainder is exact; a bf16
// is the upper half of the fp32 pattern.  a*b is accumulated (in the fp32 MFMA accumulator) from the
// six piece products of relative size >= 2^-16 (hh, hm, mh, mm, hl, lh); the dropped ml, lm, ll sum to
// < 2^-23 |a*b|, the size of ONE fp32 rounding of the product.  Measured against fp64 the result is no
// further from the truth than the v_mfma_f32_32x32x2_f32 kernel (tools/micro/gemm_bf16x3.hip: max error
// 1.5e-6 vs 2.0e-6 at K = 128, 6.6e-6 vs 7.2e-6 at K = 512), and six v_mfma_f32_32x32x16_bf16 take 192
// cycles per 16 k where eight v_mfma_f32_32x32x2_f32 take 512.  Non-finite inputs give NaN where plain
// fp32 would give Inf (Inf - Inf in the first remainder).
constexpr int kLdP = 2 * 32 + 16;  // bytes per LDS row of one bf16 plane: 32 k + 16 B pad (conflict-free b128 reads)

__device__ __forceinline__ uint32_t hi16(float f) { return __float_as_uint(f) & 0xffff0000u; }
__device__ __forceinline__ uint32_t pack_hi(uint32_t lo_elem, uint32_t hi_elem) {  // bf16 pair from two fp32 patterns
  return __builtin_amdgcn_perm(hi_elem, lo_elem, 0x07060302u);
}
__device__ __forceinline__ void split3(float4 a, uint2& h, uint2& m, uint2& l) {
  const float v[4] = {a.x, a.y, a.z, a.w};
  uint32_t hb[4], mb[4], lb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hb[i] = hi16(v[i]);
    const float r1 = v[i] - __uint_as_float(hb[i]);
    mb[i] = hi16(r1);
    lb[i] = __float_as_uint(r1 - __uint_as_float(mb[i]));  // at most 8 significant bits left: a bf16 value
  }
  h = make_uint2(pack_hi(hb[0], hb[1]), pack_hi(hb[2], hb[3]));
  m = make_uint2(pack_hi(mb[0], mb[1]), pack_hi(mb[2], mb[3]));
  l = make_uint2(pack_hi(lb[0], lb[1]), pack_hi(lb[2], lb[3]));
}
// 8 consecutive values -> 16 bytes of each plane
__device__ __forceinline__ void split8(float4 a, float4 b, uint4& h, uint4& m, uint4& l) {
  uint2 h0, m0, l0, h1, m1, l1;
  split3(a, h0, m0, l0);
  split3(b, h1, m1, l1);
  h = make_uint4(h0.x, h0.y, h1.x, h1.y);
  m = make_uint4(m0.x, m0.y, m1.x, m1.y);
  l = make_uint4(l0.x, l0.y, l1.x, l1.y);
}
// ---- the pair format (csrc/bigtile.hip, csrc/wspair.hip): the power of two that puts a row's largest magnitude `mx`
// into [2^14, 2^15) and its inverse; a zero or non-finite row is not scaled
__device__ __forceinline__ void pair_scale(float mx, float& sc, float& inv) {
  int shift = 0;
  if (mx > 0.f && mx <= 3.0e38f) {
    shift = 14 - ilogbf(mx);
    shift = shift > 110 ? 110 : (shift < -110 ? -110 : shift);
  }
  sc = ldexpf(1.f, shift);
  inv = ldexpf(1.f, -shift);
}
// four values times `sc` -> hi = fp16(x sc) (round to nearest), lo = fp16(x sc - hi) (the remainder is exact in fp32)
__device__ __forceinline__ void pair_cut4(float4 v, float sc, uint2& h, uint2& l) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 a = {v.x * sc, v.y * sc}, b = {v.z * sc, v.w * sc};
  const h2 ha = __builtin_convertvector(a, h2), hb = __builtin_convertvector(b, h2);
  const h2 la = __builtin_convertvector(a - __builtin_convertvector(ha, f2), h2);
  const h2 lb = __builtin_convertvector(b - __builtin_convertvector(hb, f2), h2);
  h = make_uint2(__builtin_bit_cast(uint32_t, ha), __builtin_bit_cast(uint32_t, hb));
  l = make_uint2(__builtin_bit_cast(uint32_t, la), __builtin_bit_cast(uint32_t, lb));
}
__device__ __forceinline__ float max_abs4(float m, float4 v) {
  return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}

// q[i] = columns c .. c+3 of contraction row 4 (tid & 7) + i (c = 4 (tid >> 3)): split every element and store column j
// as the 4 consecutive contraction steps of plane row c + j, i.e. the transposed image [column][32 k] the fragment
// reads want, for an operand whose contraction index is its SLOW dimension.
// One-piece form (bf16 GEMM mode, stemgnn_linear_set_mode(2)): an operand is ROUNDED to bf16 (nearest even, what torch's
// .bfloat16() does) and only the h plane is written; the product is then ONE matrix instruction per tile and k step.
__device__ __forceinline__ uint32_t rne_bits(float f) {
  const __bf16 b = static_cast<__bf16>(f);
  return static_cast<uint32_t>(__builtin_bit_cast(uint16_t, b));
}
__device__ __forceinline__ uint2 pack_rne(float4 a) {
  return make_uint2(rne_bits(a.x) | (rne_bits(a.y) << 16), rne_bits(a.z) | (rne_bits(a.w) << 16));
}
template <int PIECES = 3>
__device__ __forceinline__ void stash_transposed(const float4 (&q)[4], unsigned char* planes, int plane_bytes, int tid) {
  const int kq = tid & 7, cq = tid >> 3;
  if (PIECES == 1) {
    const float v[4][4] = {{q[0].x, q[0].y, q[0].z, q[0].w}, {q[1].x, q[1].y, q[1].z, q[1].w},
                           {q[2].x, q[2].y, q[2].z, q[2].w}, {q[3].x, q[3].y, q[3].z, q[3].w}};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<uint2*>(planes + (4 * cq + j) * kLdP + 8 * kq) =
          make_uint2(rne_bits(v[0][j]) | (rne_bits(v[1][j]) << 16), rne_bits(v[2][j]) | (rne_bits(v[3][j]) << 16));
    return;
  }
  uint32_t hb[4][4], mb[4][4], lb[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float v[4] = {q[i].x, q[i].y, q[i].z, q[i].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      hb[i][j] = hi16(v[j]);
      const float r1 = v[j] - __uint_as_float(hb[i][j]);
      mb[i][j] = hi16(r1);
      lb[i][j] = __float_as_uint(r1 - __uint_as_float(mb[i][j]));
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int off = (4 * cq + j) * kLdP + 8 * kq;
    *reinterpret_cast<uint2*>(planes + off) = make_uint2(pack_hi(hb[0][j], hb[1][j]), pack_hi(hb[2][j], hb[3][j]));
    *reinterpret_cast<uint2*>(planes + plane_bytes + off) =
        make_uint2(pack_hi(mb[0][j], mb[1][j]), pack_hi(mb[2][j], mb[3][j]));
    *reinterpret_cast<uint2*>(planes + 2 * plane_bytes + off) =
        make_uint2(pack_hi(lb[0][j], lb[1][j]), pack_hi(lb[2][j], lb[3][j]));
  }
}

// c += a * b over a 16-wide k step, pieces indexed [0] = h, [1] = m, [2] = l; small terms first
__device__ __forceinline__ floatx16 mfma_x3(const bf16x8 (&a)[3], const bf16x8 (&b)[3], floatx16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
  return c;
}

// the same when one operand IS a bf16 value (bf16 feature storage): its m and l pieces are zero, three products remain
// (the skipped ones would add exact zeros)
__device__ __forceinline__ floatx16 mfma_x3_a1(const bf16x8& ah, const bf16x8 (&b)[3], floatx16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b[0], c, 0, 0, 0);
  return c;
}
__device__ __forceinline__ floatx16 mfma_x3_b1(const bf16x8 (&a)[3], const bf16x8& bh, floatx16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bh, c, 0, 0, 0);
  return c;
}

// csrc/bigtile.hip: the big-tile bf16 MFMA core (256 x 256 x 64 per block, LDS-DMA staging, eight-phase ping-pong
// schedule) for the LARGE products -- D = 768 configurations.  `pieces` = 3: the exact mode (operands cut into their three
// bf16 pieces by a cut pass, the six significant piece products as six segments of one contraction); 1: the bf16 GEMM
// mode (one rounded plane).  bt_gemm_ok: the product is large enough (and shaped) for the core to beat the 128-row tile
// kernels.  Scratch comes from the arena the caller registered (stemgnn_linear_set_scratch): the bt_* return
// STEMGNN_ERR_WORKSPACE when there is none or it is too small -- the ONLY code on which callers take the tile kernels
// instead (counted by bt_missed); every other failure is propagated.
bool bt_gemm_ok(int64_t M, int64_t N, int64_t K);
void bt_served();
void bt_missed();
int bt_linear_fwd(int pieces, const float* x1, const float* w1, int64_t K1, const void* x2, int x2_kind, const float* w2,
                  int64_t K2, const float* bias, int64_t M, int64_t N, float* y, int64_t x1_rows, int64_t store_rows,
                  float* stats_partial, int64_t stats_slabs, hipStream_t st);
int bt_linear_bwd_data(int pieces, const float* dy, const float* w, int64_t M, int64_t N, int64_t K, float* dx,
                       hipStream_t st);
int bt_linear_bwd_weight(int pieces, const float* dy, const void* x, int x_kind, int64_t M, int64_t N, int64_t K, float* dw,
                         float* db, hipStream_t st);
// a phase entry point brackets its products with a scope: inside it the planes of an operand are cut once and shared
// (the caller promises the operand does not change); no-ops without an arena
void bt_scope_begin(hipStream_t st);
void bt_scope_end(hipStream_t st);
struct BtScope {
  hipStream_t st;
  explicit BtScope(hipStream_t s) : st(s) { bt_scope_begin(st); }
  ~BtScope() { bt_scope_end(st); }
  BtScope(const BtScope&) = delete;
  BtScope& operator=(const BtScope&) = delete;
};
// the quantiser's code assignment at large codebooks: the exact similarity product of all heads as one launch of the core,
// arg-max per 256-code tile from the accumulators (no [N, K] matrix), one finishing pass (same outputs as k_vq_assign;
// quant == NULL: its lean form)
bool bt_vq_assign_ok(int64_t N, int64_t H, int64_t Dc, int64_t K);
int bt_vq_assign(const float* xp, int64_t N, int64_t H, int64_t Dc, const float* embed, const float* esq, int64_t K,
                 int training, float* xn, float* norm, int64_t* ind, float* quant, float* sqerr, double sq_scale,
                 hipStream_t st);

// csrc/wspair.hip: the weight-stationary product in the PAIR format (two fp16 pieces per operand row scaled by a power of
// two, three matrix passes instead of six; csrc/bigtile.hip describes the format): rows row_base.. of  y = x w^T + b
// (K == 128; bt: y = x w with w given as [K][N]), optionally  + x1 w1^T  on the leading x1_rows rows (a sampled batch's
// layer product: only the leading, expanded nodes carry an aggregate).  stats_partial as linear_ws_launch.
bool linear_pair_on();  // stemgnn_linear_set_pair
bool linear_wsp_ok(int64_t M, int64_t N, int64_t K, int64_t x1_rows);
int linear_wsp_launch(const float* x, const float* w, const float* bias, int64_t M, int64_t N, float* y,
                      float* stats_partial, int64_t stats_block0, int64_t store_rows, bool bt, const float* x1,
                      const float* w1, int64_t x1_rows, hipStream_t st);

// ... and the quantiser's fused backward on the same skeleton (D = Dc = 128)
bool vq_bwd_wsp_ok(int64_t N, int64_t D, int64_t H, int64_t Dc);
// (rowmax [N][H], may be null: the largest magnitude of every (row, head) stretch of g_xp, for linear_ksp_launch)
int vq_bwd_wsp_launch(const float* g_out, const float* w_out, const float* g_loss, float coef, const float* xp,
                      const float* norm, const int64_t* ind, const float* embed, int64_t N, int64_t H, int64_t K,
                      float* g_xp, float* rowmax, hipStream_t st);
int vq_assign_bwd_fused_rowmax(const float* g_out, int64_t D, const float* w_out, const float* g_loss, float commit_weight,
                               const float* xp, const float* norm, const int64_t* ind, const float* embed, int64_t N,
                               int64_t H, int64_t Dc, int64_t K, float* g_xp, float* rowmax, bool* rowmax_written,
                               void* stream_);
// ... and the backward-data product of a 128 -> 512 Linear (project_in): dx [M, 128] = dy [M, 512] w, w [512][128] as
// stored; rowmax [M][4] as above; scratch (256-byte aligned) of linear_ksp_scratch_bytes(512)
bool linear_ksp_ok(int64_t M, int64_t N_in, int64_t K_out);
size_t linear_ksp_scratch_bytes(int64_t N_in);
int linear_ksp_launch(const float* dy, const float* rowmax, const float* w, int64_t M, float* dx, void* scratch, hipStream_t st);

bool vq_assign_takes_own_sqnorm(int64_t N, int64_t H, int64_t Dc, int64_t K);  // csrc/vq.hip
// ... and the code assignment (K = Dc = 128; eligibility and outputs as vq_assign_ws_launch; `esq` is not read)
int vq_assign_wsp_launch(const float* xp, int64_t N, int64_t H, const float* embed, const float* esq, float* norm,
                         int64_t* ind, float* sq_partial, unsigned int* counter, double sq_scale, float* sq_out,
                         hipStream_t st);

// csrc/wsgemm.hip: the weight-stationary dense product (K == 128): rows row_base.. of  y = x w^T + b  (bt: y = x w with w
// given as [K][N]); stats_partial[stats_block0 + tile][2][N] takes the column sums / sums of squares per 128-row tile
bool linear_ws_ok(int64_t M, int64_t N, int64_t K);
int linear_ws_launch(const float* x, const float* w, const float* bias, int64_t M, int64_t N, int64_t K, float* y,
                     float* stats_partial, int64_t row_base, int64_t stats_block0, int64_t store_rows, bool bt,
                     hipStream_t st);

// ---- "last block finishes": a reduction whose partials are written by many blocks and summed, in a fixed order, by
// whichever block arrives last -- the sum does not depend on who that is, so the result stays bit-reproducible, and
// the separate one-block finishing launch (5 us of launch floor for 256 additions) disappears.
// No fences: on this part a device-scope fence writes back and invalidates the XCD's whole L2 (measured: the
// arg-max kernel went from 121 to 345 us with one __threadfence() per block).  Instead the few words that cross
// blocks bypass the caches.  The hand-off is the form MI355X_MICROARCH.md lists as measured-valid on gfx950 / ROCm 7.2
// ("Workgroup dispatch, XCD placement & inter-workgroup visibility", first row of the sc1 table) -- an observed
// property of this part, NOT a guarantee of the HIP memory model, which is why the file refuses to compile for
// anything else (below) and why tests/test_gpu_kernels.py::test_ticketed_reduction_many_blocks_many_rounds pins it:
//   producer: every partial is written with st_agent (a device-scope relaxed atomic store = a write-through `sc1`
//             store); the storing thread runs wait_stores() (s_waitcnt vmcnt(0): memory has acknowledged it); a
//             workgroup barrier; then ONE lane takes the ticket with a device-scope relaxed atomic add;
//   consumer: the workgroup whose add returned total - 1; its other waves pass a workgroup barrier behind that add and
//             read every partial with ld_agent (an `sc1` load: served by L2 / memory, never by the CU's L1).
// The counter resets itself.  Counters come from a per-(device, stream) pool (csrc/loss_ops.hip: ticket_counter): kernels
// of one stream run one after the other, so they may share words; reductions enqueued on different streams never do.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "common.h: the fence-free last-block protocol (st_agent / wait_stores / ticket_last) is validated on gfx950 (and gfx942) only"
#endif
constexpr int kTicketSlots = 16;  // counter words per (device, stream); distinct reductions of ONE launch take distinct slots
// slot-th zero-initialised counter word of the calling thread's current device and the given stream (nullptr: the
// allocation failed).  First use per stream allocates (hipMalloc + hipMemset: not inside a stream capture).
unsigned int* ticket_counter(hipStream_t st, int slot = 0);
// 512 doubles of per-(device, stream) scratch for partial sums of launches that take no workspace from the caller
double* stream_partials(hipStream_t st);
__device__ __forceinline__ void st_agent(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(int32_t* p, int32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int32_t ld_agent(const int32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// s_waitcnt immediate of the gfx9 family (gfx942 / gfx950): vmcnt = bits [3:0] and [15:14], expcnt = [6:4],
// lgkmcnt = [11:8]; a field at its maximum means "do not wait on this counter"
constexpr int gfx9_waitcnt(int vmcnt, int expcnt, int lgkmcnt) {
  return (vmcnt & 0xF) | ((vmcnt >> 4) << 14) | ((expcnt & 0x7) << 4) | ((lgkmcnt & 0xF) << 8);
}
constexpr int kWaitVmcnt0 = gfx9_waitcnt(0, 7, 15);  // vmcnt(0), nothing else
static_assert(kWaitVmcnt0 == 0x0F70, "gfx9 s_waitcnt encoding");
__device__ __forceinline__ void wait_stores() {
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
  __builtin_amdgcn_s_waitcnt(kWaitVmcnt0);  // this thread's stores have been acknowledged by memory
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
}
// `total`: the number of blocks that take a ticket on this counter (a launch that serves several reductions gives
// each its own counter and block count)
__device__ __forceinline__ bool ticket_last(unsigned int* counter, unsigned int total) {
  __shared__ int s_ticket_last;
  __syncthreads();  // every writer of the block is past its wait_stores()
  if (threadIdx.x == 0) {
    const unsigned int t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = t == total - 1u;
    if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_ticket_last = last;
  }
  __syncthreads();
  return s_ticket_last != 0;
}
__device__ __forceinline__ bool ticket_last(unsigned int* counter) {
  return ticket_last(counter, gridDim.x * gridDim.y * gridDim.z);
}

// csrc/wsgemm.hip: the quantiser's code assignment (lean form: indices, row norms, commitment sum) with the head's codes
// as the register-resident operand; K = Dc = 128 and at least 16384 rows.  sq_partial: 2 * CUs floats.
bool vq_assign_ws_ok(int64_t N, int64_t H, int64_t Dc, int64_t K);
int vq_assign_ws_launch(const float* xp, int64_t N, int64_t H, const float* embed, const float* esq, float* norm,
                        int64_t* ind, float* sq_partial, unsigned int* counter, double sq_scale, float* sq_out,
                        hipStream_t st);

// csrc/loss_ops.hip: the three row / element losses of the heads phase (two mean squared errors, one mean of 1 - cos) in
// ONE launch each way; the same arithmetic and summation order as stemgnn_mse_loss_* / stemgnn_cosine_loss_*.
// Workspaces as for those calls (stemgnn_loss_workspace_bytes(256) for an mse, (rows) for the cosine term).
struct HeadLossJobs {
  const float *pred_a, *tgt_a;  // mse over n_a elements
  int64_t n_a;
  const float *pred_b, *tgt_b;  // mse over n_b elements
  int64_t n_b;
  const float *z, *h;  // mean(1 - cos(z_r, h_r)) over `rows` rows of width D
  int64_t rows, D;
  float *loss_a, *loss_b, *loss_c;  // forward: outputs
  float* cos_save;                  // [rows][3]
  void *ws_a, *ws_b, *ws_c;         // forward workspaces
  const float *g_a, *g_b, *g_c;     // backward: upstream gradients of the three losses
  float *gp_a, *gp_b, *gh;          // backward: gradients w.r.t. pred_a, pred_b, h
  void* zero_ptr;                   // backward, optional: a 16-byte aligned buffer cleared by the same launch
  int64_t zero_bytes;               //   (the scatter target of the topology head), size a multiple of 16
};
int head_losses_fwd(const HeadLossJobs& j, hipStream_t st);
int head_losses_bwd(const HeadLossJobs& j, hipStream_t st);

// csrc/loss_ops.hip: stemgnn_ortho_loss_fwd that also writes total[0] = plus[0] + loss[0] (plus / total may be null)
int ortho_loss_fwd_plus(const float* embed, const int64_t* ids, int64_t heads, int64_t codebook_size, int64_t code_dim,
                        int64_t num_ids, float scale, float* loss, const float* plus, float* total, void* workspace,
                        size_t workspace_bytes, void* stream);

// csrc/wsgemm.hip: products over few rows ( y = x w^T + b, or y = x w with w given as [K][N] ), several per launch; one
// wave per 32 x 32 output tile, operands straight from global memory.  Same bits as the tile kernel.
bool linear_direct_ok(int64_t M, int64_t N, int64_t K);
struct DirectBatch {
  static constexpr int kMax = 4;
  struct Job {
    const float *x, *w, *bias;
    float* y;
    int64_t M, N, K;
    int bt;
  };
  Job jobs[kMax];
  int count = 0;
  int add(const float* x, const float* w, const float* bias, int64_t M, int64_t N, int64_t K, float* y, bool bt,
          hipStream_t st);
  int flush(hipStream_t st);
};

// csrc/linear.hip: weight gradients  dw = dy^T x  (+ db = column sums of dy), several per launch.  add() queues a
// product (its slabs live in the caller's workspace, stemgnn_linear_bwd_weight_workspace_bytes(M, N, K), until the
// flush); flush() runs ONE split-product launch per element kind over all queued products and ONE fixed-order
// reduction of all slab families.  dy / x must stay valid and unchanged until flush().  A phase queues its layers'
// gradients as their inputs become available and flushes once: fewer, fuller launches (the products of the rows that
// carry an aggregate, and of the seed rows, are a fraction of a launch on their own).
struct DwBatch {
  static constexpr int kMax = 8;
  struct Job {
    const float* dy;
    const void* x;
    float *pw, *pb, *dw, *db;
    int64_t M, N, K, rows;
    int splits, kind;
  };
  Job jobs[kMax];
  int count = 0;
  int add(const float* dy, const void* x, int x_kind, int64_t M, int64_t N, int64_t K, float* dw, float* db,
          void* workspace, size_t workspace_bytes, hipStream_t st);
  int flush(hipStream_t st);
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// ---- bf16 feature storage (BASELINE config 5): node features and the layers' outputs may live in HBM as bf16
// (element kind 1) instead of fp32 (kind 0).  A bf16 value is the upper half of an fp32 pattern, so a load widens
// exactly; a store rounds to nearest-even (v_cvt_pk_bf16_f32, what torch's .bfloat16() does).  All arithmetic stays
// fp32: the kernels widen at the load and round at the store, nothing else changes.
enum { kF32 = 0, kBF16 = 1 };
__device__ __forceinline__ float4 ld4_bf16(const uint16_t* p) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                     __uint_as_float(v.y & 0xffff0000u));
}
// Raw (un-widened) form of four consecutive elements: gather loops keep the loaded bits in registers and widen at the
// point of use, so that no conversion sits between a load and the next one (the compiler otherwise waits per load).
template <int KIND> struct Raw4;
template <> struct Raw4<kF32> {
  float4 v;
  __device__ __forceinline__ void load(const void* base, int64_t i) { v = ld4(static_cast<const float*>(base) + i); }
  __device__ __forceinline__ void clear() { v = make_float4(0.f, 0.f, 0.f, 0.f); }
  __device__ __forceinline__ float4 widen() const { return v; }
};
template <> struct Raw4<kBF16> {
  uint2 v;
  __device__ __forceinline__ void load(const void* base, int64_t i) {
    v = *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(base) + i);
  }
  __device__ __forceinline__ void clear() { v = make_uint2(0u, 0u); }
  __device__ __forceinline__ float4 widen() const {
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                       __uint_as_float(v.y & 0xffff0000u));
  }
};
__device__ __forceinline__ uint32_t bf16_bits(float f) {
  const __bf16 b = static_cast<__bf16>(f);
  return static_cast<uint32_t>(__builtin_bit_cast(uint16_t, b));
}
__device__ __forceinline__ void st4_bf16(uint16_t* p, float4 v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(bf16_bits(v.x) | (bf16_bits(v.y) << 16), bf16_bits(v.z) | (bf16_bits(v.w) << 16));
}
// four consecutive elements starting at element index i of a row-major matrix of either kind (i a multiple of 4)
__device__ __forceinline__ float4 ld4_kind(const void* base, int64_t i, int kind) {
  return kind == kBF16 ? ld4_bf16(static_cast<const uint16_t*>(base) + i) : ld4(static_cast<const float*>(base) + i);
}
__device__ __forceinline__ void st4_kind(void* base, int64_t i, int kind, float4 v) {
  if (kind == kBF16) st4_bf16(static_cast<uint16_t*>(base) + i, v);
  else st4(static_cast<float*>(base) + i, v);
}


}  // namespace stemgnn
