// The asm-prefetch variant of the bf16x3 GEMM tile that FAULTED in round 1 (git show 3f7cf3d^:tools/micro/gemm_bf16x3.hip,
// k_gemm_v5), root-caused and fixed.
//
// What it does: global loads issued PF chunks ahead into rotating register sets through inline asm
// (`global_load_dwordx4`), because the compiler's own vmcnt bookkeeping waits for ALL outstanding loads at the loop
// head and collapses any prefetch distance > 1; the waits are inline asm too (`s_waitcnt vmcnt(N)`).
//
// Why it faulted.  An asm global load returns at once; its destination VGPRs are written LATER, when the data lands,
// and the compiler does not know.  With a plain "=v" output the value is, to the compiler, defined the moment the
// statement ends, so it is free to (1) allocate the output on top of the address operand or any input that dies at the
// statement, and (2) MOVE the value: in the round-1 kernel the register sets were carried around a `for` loop whose
// body issued the loads inside `if (st < steps)`, so at the back edge the allocator resolved the loop-carried values
// with copies (v_mov of registers the load had not written yet), after which the ORIGINAL registers were dead in its
// books and handed to other values -- among them the 64-bit row pointers pa[] / pb[].  When the load finally landed it
// overwrote whatever lived there; the next load through a clobbered pointer was the out-of-bounds access.
//
// The fix has three parts, all needed:
//   * early-clobber outputs ("=&v"): the destination never shares registers with an input of the statement;
//   * the value is threaded through the wait statement as a "+v" operand, so nothing that reads it can be scheduled
//     above the wait that makes it valid;
//   * NO control-flow merge between a load's issue and its wait: the K loop is a compile-time constant and fully
//     unrolled (straight-line code has no loop-carried copies; each register set stays where the load will write it).
// Build: hipcc -O3 --offload-arch=gfx950 -o gemm_asm_prefetch gemm_asm_prefetch.hip ; prints time and max error vs fp64.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kBlock = 256, kBM = 128, kBN = 128, kKC = 32;
constexpr int kLdB = 80;
constexpr int kPlane = kBM * kLdB;

__device__ inline uint32_t hi16(float f) { return __float_as_uint(f) & 0xffff0000u; }
__device__ inline uint32_t pack_hi(uint32_t lo_elem, uint32_t hi_elem) { return __builtin_amdgcn_perm(hi_elem, lo_elem, 0x07060302u); }
__device__ inline void split3v(floatx4 a, uint2& h, uint2& m, uint2& l) {
  uint32_t hb[4], mb[4], lb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hb[i] = hi16(a[i]);
    const float r1 = a[i] - __uint_as_float(hb[i]);
    mb[i] = hi16(r1);
    lb[i] = __float_as_uint(r1 - __uint_as_float(mb[i]));
  }
  h = make_uint2(pack_hi(hb[0], hb[1]), pack_hi(hb[2], hb[3]));
  m = make_uint2(pack_hi(mb[0], mb[1]), pack_hi(mb[2], mb[3]));
  l = make_uint2(pack_hi(lb[0], lb[1]), pack_hi(lb[2], lb[3]));
}

__device__ __forceinline__ void gload4(floatx4& d, const float* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(d) : "v"(p) : "memory");  // early clobber: see the header
}
template <int N>
__device__ __forceinline__ void wait_loads(floatx4 (&a)[4], floatx4 (&b)[4]) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
               : "n"(N) : "memory");
}

template <int STEPS, int PF>
__global__ void __launch_bounds__(kBlock, 2)
k_gemm_asm_prefetch(const float* __restrict__ x, const float* __restrict__ w, int64_t M, int N, float* __restrict__ y) {
  constexpr int K = STEPS * kKC;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  unsigned char* sB = smem + 3 * kPlane;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, hi = lane >> 5, lj = lane & 31;
  const int64_t m0 = static_cast<int64_t>(blockIdx.x) * kBM;
  const int n0 = blockIdx.y * kBN;
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
  floatx4 ra[PF][4], rb[PF][4];
  floatx16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

#pragma unroll
  for (int p = 0; p < PF; ++p) {
#pragma unroll
    for (int t = 0; t < 4; ++t) gload4(ra[p][t], pa[t] + p * kKC);
#pragma unroll
    for (int t = 0; t < 4; ++t) gload4(rb[p][t], pb[t] + p * kKC);
  }
#pragma unroll
  for (int st = 0; st < STEPS; ++st) {  // fully unrolled: st, st % PF and every branch below are compile-time
    constexpr int dummy = 0;
    (void)dummy;
    const int j = st % PF;
    // loads still allowed in flight behind this step's: the later steps that have been issued
    if (st + PF <= STEPS) wait_loads<8 * (PF - 1)>(ra[j], rb[j]);
    else if (STEPS - 1 - st == 1) wait_loads<8>(ra[j], rb[j]);
    else wait_loads<0>(ra[j], rb[j]);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      uint2 h, m, l;
      split3v(ra[j][t], h, m, l);
      *reinterpret_cast<uint2*>(sA + off[t]) = h;
      *reinterpret_cast<uint2*>(sA + kPlane + off[t]) = m;
      *reinterpret_cast<uint2*>(sA + 2 * kPlane + off[t]) = l;
      split3v(rb[j][t], h, m, l);
      *reinterpret_cast<uint2*>(sB + off[t]) = h;
      *reinterpret_cast<uint2*>(sB + kPlane + off[t]) = m;
      *reinterpret_cast<uint2*>(sB + 2 * kPlane + off[t]) = l;
    }
    __syncthreads();
    if (st + PF < STEPS) {
#pragma unroll
      for (int t = 0; t < 4; ++t) gload4(ra[j][t], pa[t] + (st + PF) * kKC);
#pragma unroll
      for (int t = 0; t < 4; ++t) gload4(rb[j][t], pb[t] + (st + PF) * kKC);
    }
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
    __syncthreads();
  }
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int n = n0 + wn * 64 + tn * 32 + lj;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm * 64 + tm * 32 + 4 * hi + (r & 3) + 8 * (r >> 2);
        if (m < M && n < N) y[m * N + n] = acc[tm][tn][r];
      }
    }
}

template <int STEPS, int PF>
double run(const float* x, const float* w, int64_t M, int N, float* y) {
  dim3 grid(static_cast<unsigned>((M + kBM - 1) / kBM), (N + kBN - 1) / kBN);
  const size_t lds = 6 * kPlane;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_asm_prefetch<STEPS, PF>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  for (int i = 0; i < 3; ++i) k_gemm_asm_prefetch<STEPS, PF><<<grid, kBlock, lds>>>(x, w, M, N, y);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) k_gemm_asm_prefetch<STEPS, PF><<<grid, kBlock, lds>>>(x, w, M, N, y);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / 20;
}

int main() {
  const int64_t M = 102400;
  const int N = 128;
  for (int K : {128, 256}) {
    std::vector<float> hx(M * K), hw(static_cast<size_t>(N) * K), hy(M * N);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hx) v = rnd() * (1.0f + 3.0f * rnd() * rnd());
    for (auto& v : hw) v = rnd() * 0.2f;
    float *x, *w, *y;
    hipMalloc(&x, hx.size() * 4); hipMalloc(&w, hw.size() * 4); hipMalloc(&y, hy.size() * 4);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    for (int pf : {1, 2, 3}) {
      double us = 0;
      if (K == 128) us = pf == 1 ? run<4, 1>(x, w, M, N, y) : pf == 2 ? run<4, 2>(x, w, M, N, y) : run<4, 3>(x, w, M, N, y);
      else us = pf == 1 ? run<8, 1>(x, w, M, N, y) : pf == 2 ? run<8, 2>(x, w, M, N, y) : run<8, 3>(x, w, M, N, y);
      if (hipDeviceSynchronize() != hipSuccess) { printf("K=%d PF=%d: device error %s\n", K, pf, hipGetErrorString(hipGetLastError())); return 1; }
      hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost);
      double err = 0, ref = 0;
      for (int64_t m = 0; m < M; m += 503)
        for (int n = 0; n < N; ++n) {
          double r = 0;
          for (int k = 0; k < K; ++k) r += static_cast<double>(hx[m * K + k]) * hw[static_cast<size_t>(n) * K + k];
          err = fmax(err, fabs(hy[m * N + n] - r));
          ref = fmax(ref, fabs(r));
        }
      printf("asm prefetch, M=%ld K=%d N=%d, %d chunk(s) ahead: %.1f us, max|err| vs fp64 %.3g (max|y| %.3g)\n",
             static_cast<long>(M), K, N, pf, us, err, ref);
    }
    hipFree(x); hipFree(w); hipFree(y);
  }
  return 0;
}
