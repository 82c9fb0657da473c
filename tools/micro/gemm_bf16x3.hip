// Micro-benchmark: Y[M,N] = X[M,K] W[N,K]^T in fp32 accuracy on the bf16 matrix cores.
// Every fp32 operand is split exactly into three bf16 pieces (a = h + m + l, 3 x 8 significand
// bits); the product keeps the six piece products down to 2^-16 relative (hh, hm, mh, mm, hl, lh)
// and drops ml, lm, ll (<= 2^-24 relative: the size of one fp32 rounding).  Six
// v_mfma_f32_32x32x16_bf16 replace eight v_mfma_f32_32x32x2_f32 per K=16: 192 vs 512 cycles.
// build: hipcc -O3 --offload-arch=gfx950 -o gemm_bf16x3 gemm_bf16x3.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

static uint16_t* g_planes = nullptr;
constexpr int kBlock = 256, kBM = 128, kBN = 128, kKC = 32;
constexpr int kLdB = 80;                    // bytes per LDS row of one bf16 plane (32 k + 8 pad)
constexpr int kPlane = kBM * kLdB;          // 10 KB

__device__ inline float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ inline int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// a = h + m + l exactly: three 8-bit slices of the 24-bit significand (truncation, so every piece has
// the sign of a and the remainders are exact); a bf16 is the upper half of the fp32 pattern.
__device__ inline uint32_t hi16(float f) { return __float_as_uint(f) & 0xffff0000u; }
__device__ inline uint32_t pack_hi(uint32_t lo_elem, uint32_t hi_elem) {  // bf16 pair from two masked fp32 patterns
  return __builtin_amdgcn_perm(hi_elem, lo_elem, 0x07060302u);
}
__device__ inline void split3(float4 a, uint2& h, uint2& m, uint2& l) {
  const float v[4] = {a.x, a.y, a.z, a.w};
  uint32_t hb[4], mb[4], lb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hb[i] = hi16(v[i]);
    const float r1 = v[i] - __uint_as_float(hb[i]);
    mb[i] = hi16(r1);
    const float r2 = r1 - __uint_as_float(mb[i]);
    lb[i] = __float_as_uint(r2);  // at most 8 significant bits left: already a bf16 value
  }
  h = make_uint2(pack_hi(hb[0], hb[1]), pack_hi(hb[2], hb[3]));
  m = make_uint2(pack_hi(mb[0], mb[1]), pack_hi(mb[2], mb[3]));
  l = make_uint2(pack_hi(lb[0], lb[1]), pack_hi(lb[2], lb[3]));
}

// W pre-split into three bf16 planes [3][N][K] (once per weight update, not once per block and K-chunk)
__global__ void k_split_weight(const float* __restrict__ w, int64_t n, uint16_t* __restrict__ planes) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = w[i];
  const uint32_t h = hi16(v);
  const float r1 = v - __uint_as_float(h);
  const uint32_t m = hi16(r1);
  const float r2 = r1 - __uint_as_float(m);
  planes[i] = static_cast<uint16_t>(h >> 16);
  planes[n + i] = static_cast<uint16_t>(m >> 16);
  planes[2 * n + i] = static_cast<uint16_t>(__float_as_uint(r2) >> 16);
}

// 512-thread block: 8 waves as 4 (m) x 2 (n), each 32 rows x 64 columns (1 x 2 MFMA tiles)
constexpr int kBlock8 = 512;
template <int OCC, int ABL = 0>  // ABL bit0: no global loads, bit1: no MFMA, bit2: no LDS fragment reads, bit3: no stores
__global__ void __launch_bounds__(kBlock8, OCC)
k_gemm_bf16x3(const float* __restrict__ x, const uint16_t* __restrict__ wp, int K, int64_t M, int N, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // A planes h,m,l then B planes h,m,l
  unsigned char* sA = smem;
  unsigned char* sB = smem + 3 * kPlane;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, hi = lane >> 5, lj = lane & 31;
  const int64_t m0 = static_cast<int64_t>(blockIdx.x) * kBM;
  const int n0 = blockIdx.y * kBN;
  const int steps = K / kKC;

  // per-thread source pointers, fixed for the whole K loop
  const float* pa[2];
  bool va[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int idx = t * kBlock8 + tid;
    const int64_t m = m0 + (idx >> 3);
    va[t] = m < M;
    pa[t] = x + (va[t] ? m : 0) * K + 4 * (idx & 7);
  }
  const int brow = tid >> 2, bseg = tid & 3;
  const bool vb = n0 + brow < N;
  const int64_t plane_elems = static_cast<int64_t>(N) * K;
  const uint16_t* pb = wp + static_cast<int64_t>(vb ? n0 + brow : 0) * K + 8 * bseg;
  const int offA0 = (tid >> 3) * kLdB + 8 * (tid & 7);
  const int offB = brow * kLdB + 16 * bseg;

  float4 ra[2];
  uint4 rb[3];
  auto fetch = [&](int step) {
    const int k0 = step * kKC;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (ABL & 1) ra[t] = make_float4(1.f + k0, 2.f + tid, 3.f, 4.f + step);
      else ra[t] = va[t] ? ld4(pa[t] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      if (ABL & 1) rb[p] = make_uint4(k0 + p, tid, 3, 4);
      else rb[p] = vb ? *reinterpret_cast<const uint4*>(pb + p * plane_elems + k0) : make_uint4(0, 0, 0, 0);
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int off = offA0 + t * (kBlock8 / 8) * kLdB;
      uint2 h, m, l;
      split3(ra[t], h, m, l);
      *reinterpret_cast<uint2*>(sA + off) = h;
      *reinterpret_cast<uint2*>(sA + kPlane + off) = m;
      *reinterpret_cast<uint2*>(sA + 2 * kPlane + off) = l;
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<uint4*>(sB + p * kPlane + offB) = rb[p];
  };

  floatx16 acc[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

  fetch(0);
  for (int step = 0; step < steps; ++step) {
    stash();
    __syncthreads();
    if (step + 1 < steps) fetch(step + 1);
#pragma unroll
    for (int ks = 0; ks < kKC / 16; ++ks) {
      const int ko = ks * 32 + hi * 16;
      bf16x8 a[3], b[2][3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        if (ABL & 4) {
          a[p] = bf16x8{} + (__bf16)(float)(ko + p);
          b[0][p] = a[p]; b[1][p] = a[p];
          continue;
        }
        a[p] = *reinterpret_cast<const bf16x8*>(sA + p * kPlane + (wm * 32 + lj) * kLdB + ko);
#pragma unroll
        for (int t = 0; t < 2; ++t)
          b[t][p] = *reinterpret_cast<const bf16x8*>(sB + p * kPlane + (wn * 64 + t * 32 + lj) * kLdB + ko);
      }
      if (ABL & 2) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          acc[0][p] += (float)a[p][0] + (float)b[0][p][1];
          acc[1][p] += (float)a[p][2] + (float)b[1][p][3];
        }
        continue;
      }
      if (ABL & 16) {  // product outer, tile inner: consecutive MFMAs write different accumulators
        constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t) {
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa[t]], b[tn][pb[t]], acc[tn], 0, 0, 0);
        }
        continue;
      }
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        floatx16 c = acc[tn];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[tn][0], c, 0, 0, 0);  // l h
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[tn][2], c, 0, 0, 0);  // h l
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[tn][1], c, 0, 0, 0);  // m m
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[tn][0], c, 0, 0, 0);  // m h
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[tn][1], c, 0, 0, 0);  // h m
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[tn][0], c, 0, 0, 0);  // h h
        acc[tn] = c;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int n = n0 + wn * 64 + tn * 32 + lj;
    float* yr = y + (m0 + wm * 32 + 4 * hi) * N + n;
    const int64_t mrem = M - (m0 + wm * 32 + 4 * hi);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dr = (r & 3) + 8 * (r >> 2);
      if ((ABL & 8) && acc[tn][r] != 12345.678f) continue;
      if (dr < mrem && n < N) yr[static_cast<int64_t>(dr) * N] = acc[tn][r];
    }
  }
}

// v3: LDS double-buffered, one barrier per K stage; the split + LDS writes of stage s+1 and the global loads of
// stage s+2 are in the same basic block as the MFMAs of stage s.  4 waves 2x2, 64x64 per wave.
template <int KC, int OCC, int SCHED>
__global__ void __launch_bounds__(kBlock, OCC)
k_gemm_v3(const float* __restrict__ x, const uint16_t* __restrict__ wp, int K, int64_t M, int N, float* __restrict__ y) {
  constexpr int LD = 2 * KC + 16;       // bytes per LDS row of a plane
  constexpr int PL = kBM * LD;          // one plane
  constexpr int BUF = 6 * PL;           // A h,m,l + B h,m,l
  constexpr int FA = kBM * KC / 4 / kBlock;       // float4 of A per thread per stage (KC=16: 2, KC=32: 4)
  constexpr int FB = kBN * KC * 2 / 16 / kBlock;  // 16-byte pieces of one B plane per thread (KC=16: 1, KC=32: 2)
  constexpr int SEGA = KC / 4, SEGB = KC / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, hi = lane >> 5, lj = lane & 31;
  const int64_t m0 = static_cast<int64_t>(blockIdx.x) * kBM;
  const int n0 = blockIdx.y * kBN;
  const int S = K / KC;

  const float* pa[FA];
  int offA[FA];
#pragma unroll
  for (int t = 0; t < FA; ++t) {
    const int idx = t * kBlock + tid;
    int64_t m = m0 + idx / SEGA;
    if (m >= M) m = M - 1;  // clamp: duplicate rows are computed and never stored
    pa[t] = x + m * K + 4 * (idx % SEGA);
    offA[t] = (idx / SEGA) * LD + 8 * (idx % SEGA);
  }
  const uint16_t* pb[FB];
  int offB[FB];
  const int64_t plane_elems = static_cast<int64_t>(N) * K;
#pragma unroll
  for (int t = 0; t < FB; ++t) {
    const int idx = t * kBlock + tid;
    int n = n0 + idx / SEGB;
    if (n >= N) n = N - 1;
    pb[t] = wp + static_cast<int64_t>(n) * K + 8 * (idx % SEGB);
    offB[t] = (idx / SEGB) * LD + 16 * (idx % SEGB);
  }

  float4 ra[2][FA];
  uint4 rb[2][3][FB];
  auto fetch = [&](int s, float4* qa, uint4 (*qb)[FB]) {
    const int k0 = (s < S ? s : S - 1) * KC;  // past the end: harmless re-read of the last stage
#pragma unroll
    for (int t = 0; t < FA; ++t) qa[t] = ld4(pa[t] + k0);
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int t = 0; t < FB; ++t) qb[p][t] = *reinterpret_cast<const uint4*>(pb[t] + p * plane_elems + k0);
  };
  auto stash = [&](unsigned char* buf, const float4* qa, const uint4 (*qb)[FB]) {
#pragma unroll
    for (int t = 0; t < FA; ++t) {
      uint2 h, m, l;
      split3(qa[t], h, m, l);
      *reinterpret_cast<uint2*>(buf + offA[t]) = h;
      *reinterpret_cast<uint2*>(buf + PL + offA[t]) = m;
      *reinterpret_cast<uint2*>(buf + 2 * PL + offA[t]) = l;
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int t = 0; t < FB; ++t) *reinterpret_cast<uint4*>(buf + (3 + p) * PL + offB[t]) = qb[p][t];
  };

  floatx16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  auto mma = [&](const unsigned char* cur) {
#pragma unroll
    for (int ks = 0; ks < KC / 16; ++ks) {
      const int ko = ks * 32 + hi * 16;
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          a[t][p] = *reinterpret_cast<const bf16x8*>(cur + p * PL + (wm * 64 + t * 32 + lj) * LD + ko);
          b[t][p] = *reinterpret_cast<const bf16x8*>(cur + (3 + p) * PL + (wn * 64 + t * 32 + lj) * LD + ko);
        }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          floatx16 c = acc[tm][tn];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][2], b[tn][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][1], b[tn][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][1], b[tn][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][0], c, 0, 0, 0);
          acc[tm][tn] = c;
        }
    }
  };
  // stage(s): loads of stage s+2 into register set s&1 (free: its data went to LDS during stage s-1), split+store of
  // stage s+1 from set (s+1)&1 into the other LDS buffer, MFMAs of stage s from this buffer.
  auto stage = [&](int s, int set) {
    unsigned char* cur = smem + (s & 1) * BUF;
    unsigned char* nxt = smem + ((s + 1) & 1) * BUF;
    fetch(s + 2, ra[set], rb[set]);
    __builtin_amdgcn_sched_barrier(0);
    stash(nxt, ra[set ^ 1], rb[set ^ 1]);
    if (SCHED == 0) __builtin_amdgcn_sched_barrier(0);
    mma(cur);
    if (SCHED) {
#pragma unroll
      for (int i = 0; i < 24 * (KC / 16); ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, SCHED, 0);  // VALU
        __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);      // DS
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  };

  fetch(0, ra[0], rb[0]);
  fetch(1, ra[1], rb[1]);
  stash(smem, ra[0], rb[0]);
  __syncthreads();
  for (int s = 0; s < S; s += 2) {  // S even
    stage(s, 0);
    stage(s + 1, 1);
  }
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int n = n0 + wn * 64 + tn * 32 + lj;
      float* yr = y + (m0 + wm * 64 + tm * 32 + 4 * hi) * N + n;
      const int64_t mrem = M - (m0 + wm * 64 + tm * 32 + 4 * hi);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        if (dr < mrem && n < N) yr[static_cast<int64_t>(dr) * N] = acc[tm][tn][r];
      }
    }
}

template <int KC, int OCC, int SCHED>
static void launch_v3(const float* x, const float* w, int K, int64_t M, int N, float* y) {
  constexpr int lds = 2 * 6 * kBM * (2 * KC + 16);
  dim3 grid(static_cast<unsigned>((M + kBM - 1) / kBM), (N + kBN - 1) / kBN);
  static bool once = [] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_v3<KC, OCC, SCHED>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    return true;
  }();
  (void)once;
  k_split_weight<<<(N * K + 255) / 256, 256>>>(w, static_cast<int64_t>(N) * K, g_planes);
  k_gemm_v3<KC, OCC, SCHED><<<grid, kBlock, lds>>>(x, g_planes, K, M, N, y);
}

// v6: the library tile (4 waves, one LDS buffer, in-kernel split of both operands) with TWO register sets: the loads
// of chunk s+2 are issued while chunk s is multiplied, so two chunks per block are in flight.  Unrolled by two so the
// sets are static; loads are clamped instead of predicated so every iteration issues the same eight loads.
template <int OCC, int FENCE>
__global__ void __launch_bounds__(kBlock, OCC)
k_gemm_v6(const float* __restrict__ x, const float* __restrict__ w, int K, int64_t M, int N, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  unsigned char* sB = smem + 3 * kPlane;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, hi = lane >> 5, lj = lane & 31;
  const int64_t m0 = static_cast<int64_t>(blockIdx.x) * kBM;
  const int n0 = blockIdx.y * kBN;
  const int steps = K / kKC;
  const float *pa[4], *pb[4];
  int off[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int idx = t * kBlock + tid;
    int64_t m = m0 + (idx >> 3);
    if (m >= M) m = M - 1;
    int n = n0 + (idx >> 3);
    if (n >= N) n = N - 1;
    pa[t] = x + m * K + 4 * (idx & 7);
    pb[t] = w + static_cast<int64_t>(n) * K + 4 * (idx & 7);
    off[t] = (idx >> 3) * kLdB + 8 * (idx & 7);
  }
  float4 ra0[4], rb0[4], ra1[4], rb1[4];
  auto fetch = [&](int step, float4 (&qa)[4], float4 (&qb)[4]) {
    const int k0 = (step < steps ? step : steps - 1) * kKC;
#pragma unroll
    for (int t = 0; t < 4; ++t) qa[t] = ld4(pa[t] + k0);
#pragma unroll
    for (int t = 0; t < 4; ++t) qb[t] = ld4(pb[t] + k0);
  };
  auto stash = [&](const float4 (&qa)[4], const float4 (&qb)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      uint2 h, m, l;
      split3(qa[t], h, m, l);
      *reinterpret_cast<uint2*>(sA + off[t]) = h;
      *reinterpret_cast<uint2*>(sA + kPlane + off[t]) = m;
      *reinterpret_cast<uint2*>(sA + 2 * kPlane + off[t]) = l;
      split3(qb[t], h, m, l);
      *reinterpret_cast<uint2*>(sB + off[t]) = h;
      *reinterpret_cast<uint2*>(sB + kPlane + off[t]) = m;
      *reinterpret_cast<uint2*>(sB + 2 * kPlane + off[t]) = l;
    }
  };
  floatx16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  auto mma = [&]() {
#pragma unroll
    for (int ks = 0; ks < kKC / 16; ++ks) {
      const int ko = ks * 32 + hi * 16;
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          a[t][p] = *reinterpret_cast<const bf16x8*>(sA + p * kPlane + (wm * 64 + t * 32 + lj) * kLdB + ko);
          b[t][p] = *reinterpret_cast<const bf16x8*>(sB + p * kPlane + (wn * 64 + t * 32 + lj) * kLdB + ko);
        }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          floatx16 c = acc[tm][tn];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][2], b[tn][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][1], b[tn][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][1], b[tn][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][0], c, 0, 0, 0);
          acc[tm][tn] = c;
        }
    }
  };
  fetch(0, ra0, rb0);
  fetch(1, ra1, rb1);
  for (int step = 0; step < steps; step += 2) {  // steps even
    stash(ra0, rb0);
    __syncthreads();
    fetch(step + 2, ra0, rb0);
    __builtin_amdgcn_sched_barrier(0);
    mma();
    if (FENCE) __builtin_amdgcn_sched_barrier(0);  // keeps the next stash (and its wait for set 1) below the MFMAs
    __syncthreads();
    stash(ra1, rb1);
    __syncthreads();
    fetch(step + 3, ra1, rb1);
    __builtin_amdgcn_sched_barrier(0);
    mma();
    if (FENCE) __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  }
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int n = n0 + wn * 64 + tn * 32 + lj;
      float* yr = y + (m0 + wm * 64 + tm * 32 + 4 * hi) * N + n;
      const int64_t mrem = M - (m0 + wm * 64 + tm * 32 + 4 * hi);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        if (dr < mrem && n < N) yr[static_cast<int64_t>(dr) * N] = acc[tm][tn][r];
      }
    }
}

template <int OCC, int FENCE, int STEPS>
__global__ void __launch_bounds__(kBlock, OCC)
k_gemm_v7(const float* __restrict__ x, const float* __restrict__ w, int K, int64_t M, int N, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  unsigned char* sB = smem + 3 * kPlane;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, hi = lane >> 5, lj = lane & 31;
  const int64_t m0 = static_cast<int64_t>(blockIdx.x) * kBM;
  const int n0 = blockIdx.y * kBN;
  constexpr int steps = STEPS;  // K == 32 * STEPS: the K loop is straight-line code, so every s_waitcnt is counted exactly
  const float *pa[4], *pb[4];
  int off[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int idx = t * kBlock + tid;
    int64_t m = m0 + (idx >> 3);
    if (m >= M) m = M - 1;
    int n = n0 + (idx >> 3);
    if (n >= N) n = N - 1;
    pa[t] = x + m * K + 4 * (idx & 7);
    pb[t] = w + static_cast<int64_t>(n) * K + 4 * (idx & 7);
    off[t] = (idx >> 3) * kLdB + 8 * (idx & 7);
  }
  float4 ra0[4], rb0[4], ra1[4], rb1[4];
  auto fetch = [&](int step, float4 (&qa)[4], float4 (&qb)[4]) {
    const int k0 = (step < steps ? step : steps - 1) * kKC;
#pragma unroll
    for (int t = 0; t < 4; ++t) qa[t] = ld4(pa[t] + k0);
#pragma unroll
    for (int t = 0; t < 4; ++t) qb[t] = ld4(pb[t] + k0);
  };
  auto stash = [&](const float4 (&qa)[4], const float4 (&qb)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      uint2 h, m, l;
      split3(qa[t], h, m, l);
      *reinterpret_cast<uint2*>(sA + off[t]) = h;
      *reinterpret_cast<uint2*>(sA + kPlane + off[t]) = m;
      *reinterpret_cast<uint2*>(sA + 2 * kPlane + off[t]) = l;
      split3(qb[t], h, m, l);
      *reinterpret_cast<uint2*>(sB + off[t]) = h;
      *reinterpret_cast<uint2*>(sB + kPlane + off[t]) = m;
      *reinterpret_cast<uint2*>(sB + 2 * kPlane + off[t]) = l;
    }
  };
  floatx16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  auto mma = [&]() {
#pragma unroll
    for (int ks = 0; ks < kKC / 16; ++ks) {
      const int ko = ks * 32 + hi * 16;
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          a[t][p] = *reinterpret_cast<const bf16x8*>(sA + p * kPlane + (wm * 64 + t * 32 + lj) * kLdB + ko);
          b[t][p] = *reinterpret_cast<const bf16x8*>(sB + p * kPlane + (wn * 64 + t * 32 + lj) * kLdB + ko);
        }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          floatx16 c = acc[tm][tn];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][2], b[tn][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][1], b[tn][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][1], b[tn][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][0], b[tn][0], c, 0, 0, 0);
          acc[tm][tn] = c;
        }
    }
  };
  fetch(0, ra0, rb0);
  fetch(1, ra1, rb1);
#pragma unroll
  for (int step = 0; step < steps; step += 2) {  // steps even, fully unrolled
    stash(ra0, rb0);
    __syncthreads();
    if (step + 2 < steps) fetch(step + 2, ra0, rb0);
    __builtin_amdgcn_sched_barrier(0);
    mma();
    if (FENCE) __builtin_amdgcn_sched_barrier(0);  // keeps the next stash (and its wait for set 1) below the MFMAs
    __syncthreads();
    stash(ra1, rb1);
    __syncthreads();
    if (step + 3 < steps) fetch(step + 3, ra1, rb1);
    __builtin_amdgcn_sched_barrier(0);
    mma();
    if (FENCE) __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  }
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int n = n0 + wn * 64 + tn * 32 + lj;
      float* yr = y + (m0 + wm * 64 + tm * 32 + 4 * hi) * N + n;
      const int64_t mrem = M - (m0 + wm * 64 + tm * 32 + 4 * hi);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        if (dr < mrem && n < N) yr[static_cast<int64_t>(dr) * N] = acc[tm][tn][r];
      }
    }
}

template <int OCC, int FENCE, int STEPS>
static void launch_v7(const float* x, const float* w, int K, int64_t M, int N, float* y) {
  dim3 grid(static_cast<unsigned>((M + kBM - 1) / kBM), (N + kBN - 1) / kBN);
  static bool once = [] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_v7<OCC, FENCE, STEPS>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 6 * kPlane);
    return true;
  }();
  (void)once;
  k_gemm_v7<OCC, FENCE, STEPS><<<grid, kBlock, 6 * kPlane>>>(x, w, K, M, N, y);
}

template <int OCC, int FENCE>
static void launch_v6(const float* x, const float* w, int K, int64_t M, int N, float* y) {
  dim3 grid(static_cast<unsigned>((M + kBM - 1) / kBM), (N + kBN - 1) / kBN);
  static bool once = [] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_v6<OCC, FENCE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        6 * kPlane);
    return true;
  }();
  (void)once;
  k_gemm_v6<OCC, FENCE><<<grid, kBlock, 6 * kPlane>>>(x, w, K, M, N, y);
}

// the fp32-MFMA tile of csrc/linear.hip, for the side-by-side number
constexpr int kLd = kKC + 4;
template <int OCC>
__global__ void __launch_bounds__(kBlock, OCC)
k_gemm_f32(const float* __restrict__ x, const float* __restrict__ w, int K, int64_t M, int N, float* __restrict__ y) {
  __shared__ __attribute__((aligned(16))) float sA[kBM * kLd];
  __shared__ __attribute__((aligned(16))) float sB[kBN * kLd];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, hi = lane >> 5, lj = lane & 31;
  const int64_t m0 = static_cast<int64_t>(blockIdx.x) * kBM;
  const int n0 = blockIdx.y * kBN;
  const int steps = K / kKC;
  float4 ra[4], rb[4];
  auto fetch = [&](int step) {
    const int k0 = step * kKC;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = t * kBlock + tid;
      const int r = idx >> 3, k = k0 + 4 * (idx & 7);
      const int64_t m = m0 + r;
      ra[t] = m < M ? ld4(x + m * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      rb[t] = n0 + r < N ? ld4(w + static_cast<int64_t>(n0 + r) * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  floatx16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  fetch(0);
  for (int step = 0; step < steps; ++step) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = t * kBlock + tid;
      *reinterpret_cast<float4*>(sA + (idx >> 3) * kLd + 4 * (idx & 7)) = ra[t];
      *reinterpret_cast<float4*>(sB + (idx >> 3) * kLd + 4 * (idx & 7)) = rb[t];
    }
    __syncthreads();
    if (step + 1 < steps) fetch(step + 1);
#pragma unroll
    for (int ms = 0; ms < kKC / 8; ++ms) {
      const int ko = ms * 8 + hi * 4;
      float4 a[2], b[2];
      for (int t = 0; t < 2; ++t) a[t] = ld4(sA + (wm * 64 + t * 32 + lj) * kLd + ko);
      for (int t = 0; t < 2; ++t) b[t] = ld4(sB + (wn * 64 + t * 32 + lj) * kLd + ko);
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
  for (int tm = 0; tm < 2; ++tm)
    for (int tn = 0; tn < 2; ++tn) {
      const int n = n0 + wn * 64 + tn * 32 + lj;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm * 64 + tm * 32 + acc_row(r, hi);
        if (m < M && n < N) y[m * N + n] = acc[tm][tn][r];
      }
    }
}

static double time_us(void (*launch)(const float*, const float*, int, int64_t, int, float*), const float* x,
                      const float* w, int K, int64_t M, int N, float* y) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch(x, w, K, M, N, y);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) launch(x, w, K, M, N, y);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / 20;
}

template <int OCC, int ABL = 0>
static void launch_split(const float* x, const float* w, int K, int64_t M, int N, float* y) {
  k_split_weight<<<(N * K + 255) / 256, 256>>>(w, static_cast<int64_t>(N) * K, g_planes);
  dim3 grid(static_cast<unsigned>((M + kBM - 1) / kBM), (N + kBN - 1) / kBN);
  static bool once = [] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_bf16x3<OCC, ABL>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 6 * kPlane);
    return true;
  }();
  (void)once;
  k_gemm_bf16x3<OCC, ABL><<<grid, kBlock8, 6 * kPlane>>>(x, g_planes, K, M, N, y);
}
static void launch_f32(const float* x, const float* w, int K, int64_t M, int N, float* y) {
  dim3 grid(static_cast<unsigned>((M + kBM - 1) / kBM), (N + kBN - 1) / kBN);
  k_gemm_f32<2><<<grid, kBlock>>>(x, w, K, M, N, y);
}

int main() {
  const int64_t M = 102400;
  const int KMAX = 512, NMAX = 512;
  std::vector<float> hx(M * KMAX), hw(NMAX * KMAX);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : hx) v = rnd() * (1.0f + 3.0f * rnd() * rnd());  // mixed magnitudes, both signs
  for (auto& v : hw) v = rnd() * 0.2f;
  float *x, *w, *y, *y2;
  hipMalloc(&x, hx.size() * 4); hipMalloc(&w, hw.size() * 4);
  hipMalloc(&y, M * NMAX * 4); hipMalloc(&y2, M * NMAX * 4);
  hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&g_planes, 3 * NMAX * KMAX * 2);
  std::vector<float> hy(M * NMAX), hy2(M * NMAX);
  for (auto kn : {std::pair<int, int>{128, 128}, {256, 128}, {512, 128}, {128, 512}}) {
    const int K = kn.first, N = kn.second;
    {
      double t7 = 0, t7f = 0;
      if (K == 128) { t7 = time_us(launch_v7<2, 0, 4>, x, w, K, M, N, y2); t7f = time_us(launch_v7<2, 1, 4>, x, w, K, M, N, y2); }
      if (K == 256) { t7 = time_us(launch_v7<2, 0, 8>, x, w, K, M, N, y2); t7f = time_us(launch_v7<2, 1, 8>, x, w, K, M, N, y2); }
      if (K == 512) { t7 = time_us(launch_v7<2, 0, 16>, x, w, K, M, N, y2); t7f = time_us(launch_v7<2, 1, 16>, x, w, K, M, N, y2); }
      printf("  v7 (two register sets, K loop fully unrolled): free schedule %.1f us, fenced %.1f us\n", t7, t7f);
    }
    printf("  v6 (two register sets, 4 waves): free schedule %.1f us, fenced %.1f us\n",
           time_us(launch_v6<2, 0>, x, w, K, M, N, y2), time_us(launch_v6<2, 1>, x, w, K, M, N, y2));
    {
      std::vector<float> h3(M * N);
      hipMemcpy(h3.data(), y2, M * N * 4, hipMemcpyDeviceToHost);
      double e3 = 0;
      for (int64_t m = 0; m < M; m += 499)
        for (int n = 0; n < N; ++n) {
          double r = 0;
          for (int k = 0; k < K; ++k) r += static_cast<double>(hx[m * K + k]) * hw[static_cast<int64_t>(n) * K + k];
          e3 = fmax(e3, fabs(h3[m * N + n] - r));
        }
      printf("  v6 max|err| vs fp64: %.3g\n", e3);
    }
    printf("  8-wave tile: chained accumulators %.1f us, interleaved accumulators %.1f us; MFMA only (no loads/stores): %.1f / %.1f us\n",
           time_us(launch_split<2, 0>, x, w, K, M, N, y2), time_us(launch_split<2, 16>, x, w, K, M, N, y2),
           time_us(launch_split<2, 9>, x, w, K, M, N, y2), time_us(launch_split<2, 25>, x, w, K, M, N, y2));
    const double us_s1 = time_us(launch_split<1>, x, w, K, M, N, y);
    const double us_s = time_us(launch_split<2>, x, w, K, M, N, y);
    const double us_f = time_us(launch_f32, x, w, K, M, N, y2);
    hipMemcpy(hy.data(), y, M * N * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hy2.data(), y2, M * N * 4, hipMemcpyDeviceToHost);
    // error against fp64 on a sample of rows (x, w read with row stride K)
    double e_split = 0, e_f32 = 0, ref_max = 0;
    for (int64_t m = 0; m < M; m += 997)
      for (int n = 0; n < N; ++n) {
        double r = 0;
        for (int k = 0; k < K; ++k) r += static_cast<double>(hx[m * K + k]) * hw[static_cast<int64_t>(n) * K + k];
        e_split = fmax(e_split, fabs(hy[m * N + n] - r));
        e_f32 = fmax(e_f32, fabs(hy2[m * N + n] - r));
        ref_max = fmax(ref_max, fabs(r));
      }
    const double fl = 2.0 * M * K * N, by = (M * K + M * N + N * K) * 4.0;
    printf("M=%ld K=%d N=%d: bf16x3 occ1 %7.1f us, occ2 %7.1f us (%6.1f TF/s, %5.2f TB/s)  fp32-mfma %7.1f us (%6.1f TF/s)  "
           "max|err| vs fp64: bf16x3 %.3g  fp32-mfma %.3g  (max|y| %.3g)\n",
           static_cast<long>(M), K, N, us_s1, us_s, fl / us_s / 1e6, by / us_s / 1e6, us_f, fl / us_f / 1e6, e_split, e_f32, ref_max);
  }
  return 0;
}
