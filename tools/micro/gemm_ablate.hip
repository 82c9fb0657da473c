// Ablation of the 128x128x32 LDS-staged fp32 MFMA GEMM (linear fwd v1 structure) on M=102400, N=128.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int kBlock = 256, kBM = 128, kBN = 128, kKC = 32, kLd = 36;
__device__ inline float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ inline void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

template <bool LOAD, bool STORE, bool BARRIER2, int OCC>
__global__ void __launch_bounds__(kBlock, OCC)
k(const float* __restrict__ x, const float* __restrict__ w, int K, int64_t M, int N, float* __restrict__ y) {
  __shared__ __attribute__((aligned(16))) float sA[kBM * kLd];
  __shared__ __attribute__((aligned(16))) float sB[kBN * kLd];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, hi = lane >> 5, lj = lane & 31;
  const int64_t m0 = (int64_t)blockIdx.x * kBM;
  const int steps = K / kKC;
  float4 ra[4], rb[4];
  for (int t = 0; t < 4; ++t) { ra[t] = make_float4(1, 2, 3, 4); rb[t] = make_float4(0.5f, 0.25f, 1, 2); }
  auto fetch = [&](int step) {
    if (!LOAD) return;
    const int k0 = step * kKC;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = t * kBlock + tid;
      const int r = idx >> 3, k = k0 + 4 * (idx & 7);
      ra[t] = ld4(x + (m0 + r) * K + k);
      rb[t] = ld4(w + (int64_t)r * K + k);
    }
  };
  floatx16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  fetch(0);
  for (int step = 0; step < steps; ++step) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = t * kBlock + tid;
      st4(sA + (idx >> 3) * kLd + 4 * (idx & 7), ra[t]);
      st4(sB + (idx >> 3) * kLd + 4 * (idx & 7), rb[t]);
    }
    __syncthreads();
    if (step + 1 < steps) fetch(step + 1);
#pragma unroll
    for (int ms = 0; ms < 4; ++ms) {
      const int ko = ms * 8 + hi * 4;
      float4 a[2], b[2];
      for (int t = 0; t < 2; ++t) {
        a[t] = ld4(sA + (wm * 64 + t * 32 + lj) * kLd + ko);
        b[t] = ld4(sB + (wn * 64 + t * 32 + lj) * kLd + ko);
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
    if (BARRIER2) __syncthreads();
  }
  if (STORE) {
    for (int tn = 0; tn < 2; ++tn) {
      const int n = wn * 64 + tn * 32 + lj;
      for (int tm = 0; tm < 2; ++tm) {
        float* yp = y + (m0 + wm * 64 + tm * 32 + 4 * hi) * N + n;
        for (int r = 0; r < 16; ++r) yp[(int64_t)((r & 3) + 8 * (r >> 2)) * N] = acc[tm][tn][r];
      }
    }
  } else {
    float s = 0.f;
    for (int tn = 0; tn < 2; ++tn) for (int tm = 0; tm < 2; ++tm) for (int r = 0; r < 16; ++r) s += acc[tm][tn][r];
    if (s == 12345.678f) y[tid] = s;
  }
}

template <bool LOAD, bool STORE, bool B2, int OCC>
void run(const char* name, int K, const float* x, const float* w, float* y) {
  const int64_t M = 102400; const int N = 128;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) k<LOAD, STORE, B2, OCC><<<M / 128, 256>>>(x, w, K, M, N, y);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) k<LOAD, STORE, B2, OCC><<<M / 128, 256>>>(x, w, K, M, N, y);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double us = ms * 1e3 / reps;
  printf("%-44s K=%3d  %7.1f us  %6.1f TFLOP/s\n", name, K, us, 2.0 * M * N * K / us / 1e6);
}

int main() {
  const int64_t M = 102400;
  float *x, *w, *y;
  hipMalloc(&x, M * 512 * 4); hipMalloc(&w, 128 * 512 * 4); hipMalloc(&y, M * 128 * 4);
  std::vector<float> h(M * 64);
  for (size_t i = 0; i < h.size(); ++i) h[i] = ((i * 2654435761u) % 1000) * 0.001f - 0.5f;
  for (int j = 0; j < 8; ++j) hipMemcpy(x + j * h.size(), h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(w, h.data(), 128 * 512 * 4, hipMemcpyHostToDevice);
  for (int K : {128, 512}) {
    run<true, true, true, 2>("full (load+store, 2 barriers, occ2)", K, x, w, y);
    run<true, true, true, 3>("full occ3", K, x, w, y);
    run<false, true, true, 3>("no global loads", K, x, w, y);
    run<true, false, true, 3>("no stores", K, x, w, y);
    run<false, false, true, 3>("no loads, no stores", K, x, w, y);
    run<false, false, false, 3>("no loads/stores, 1 barrier (racy)", K, x, w, y);
  }
  return 0;
}
