// Prototype: 128x128x32 fp32 MFMA GEMM with direct global->LDS staging (global_load_lds_dwordx4),
// XOR-swizzled source addresses, 2 LDS buffers.  Y[M,128] = X[M,K] W[128,K]^T.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int kBlock = 256, kBM = 128, kBN = 128, kKC = 32;
__device__ inline float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

#ifdef GLDS_ASM
// issued through inline asm so that the compiler's waitcnt pass does not know the LDS is written asynchronously
// (with the builtin it puts s_waitcnt vmcnt(0) in front of the first ds_read after it, which serialises the prefetch)
__device__ inline void glds16(const float* g, float* l) {
  const uint32_t lds_addr = __builtin_amdgcn_readfirstlane(
      static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) void*)l)));
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(g) : "memory", "m0");
}
#else
__device__ inline void glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
#endif

template <int OCC>
__global__ void __launch_bounds__(kBlock, OCC)
k(const float* __restrict__ x, const float* __restrict__ w, int K, int64_t M, int N, float* __restrict__ y) {
  __shared__ __attribute__((aligned(1024))) float lds[2 * (kBM + kBN) * kKC];  // [buf][A 128x32 | B 128x32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, hi = lane >> 5, lj = lane & 31;
  const int64_t m0 = (int64_t)blockIdx.x * kBM;
  const int steps = K / kKC;
  // staging geometry: wave instruction t covers rows (t*4 + wave)*8 .. +7 of the chunk
  const int sub_r = lane >> 3, pos = lane & 7;
  auto issue = [&](int step) {
    float* base = lds + (step & 1) * (kBM + kBN) * kKC;
    const int k0 = step * kKC;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = (t * 4 + wave) * 8 + sub_r;
      const int ks = pos ^ ((r >> 1) & 7);
      glds16(x + (m0 + r) * K + k0 + 4 * ks, base + (t * 4 + wave) * 8 * kKC);
      glds16(w + (int64_t)r * K + k0 + 4 * ks, base + kBM * kKC + (t * 4 + wave) * 8 * kKC);
    }
  };
  floatx16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  issue(0);
  for (int step = 0; step < steps; ++step) {
    if (step + 1 < steps) {
      issue(step + 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const float* sA = lds + (step & 1) * (kBM + kBN) * kKC;
    const float* sB = sA + kBM * kKC;
#pragma unroll
    for (int ms = 0; ms < 4; ++ms) {
      const int slot = ms * 2 + hi;
      float4 a[2], b[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ra = wm * 64 + t * 32 + lj, rb = wn * 64 + t * 32 + lj;
        a[t] = ld4(sA + ra * kKC + 4 * (slot ^ ((ra >> 1) & 7)));
        b[t] = ld4(sB + rb * kKC + 4 * (slot ^ ((rb >> 1) & 7)));
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
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  for (int tn = 0; tn < 2; ++tn) {
    const int n = wn * 64 + tn * 32 + lj;
    for (int tm = 0; tm < 2; ++tm) {
      float* yp = y + (m0 + wm * 64 + tm * 32 + 4 * hi) * N + n;
      for (int r = 0; r < 16; ++r) yp[(int64_t)((r & 3) + 8 * (r >> 2)) * N] = acc[tm][tn][r];
    }
  }
}

template <int OCC>
void run(const char* name, int K, const float* x, const float* w, float* y, const std::vector<float>& hx,
         const std::vector<float>& hw) {
  const int64_t M = 102400; const int N = 128;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) k<OCC><<<M / 128, 256>>>(x, w, K, M, N, y);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) k<OCC><<<M / 128, 256>>>(x, w, K, M, N, y);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double us = ms * 1e3 / reps;
  std::vector<float> hy(256 * 128);
  hipMemcpy(hy.data(), y + (M - 256) * 128, hy.size() * 4, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int r = 0; r < 256; r += 37) for (int n = 0; n < 128; n += 5) {
    double ref = 0; for (int kk = 0; kk < K; ++kk) ref += (double)hx[(M - 256 + r) * K + kk] * hw[(size_t)n * K + kk];
    maxerr = fmax(maxerr, fabs(ref - hy[r * 128 + n]));
  }
  printf("%-28s K=%3d  %7.1f us  %6.1f TFLOP/s  maxerr %.2e\n", name, K, us, 2.0 * M * N * K / us / 1e6, maxerr);
}

int main() {
  const int64_t M = 102400;
  for (int K : {128, 256, 512}) {
    float *x, *w, *y;
    hipMalloc(&x, M * K * 4); hipMalloc(&w, 128 * K * 4); hipMalloc(&y, M * 128 * 4);
    std::vector<float> hx(M * K), hw(128 * K);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = ((i * 2654435761u) % 1000) * 0.001f - 0.5f;
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = ((i * 40503u) % 997) * 0.001f - 0.5f;
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    run<1>("glds 2buf occ1", K, x, w, y, hx, hw);
    run<2>("glds 2buf occ2", K, x, w, y, hx, hw);
    hipFree(x); hipFree(w); hipFree(y);
  }
  return 0;
}
