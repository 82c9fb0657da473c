// Micro-benchmark: fp32 MFMA issue rate with/without LDS operand reads (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int MODE, int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters, const float* in) {
  __shared__ __attribute__((aligned(16))) float lds[128 * 36 * 2];
  for (int i = threadIdx.x; i < 128 * 36 * 2; i += 256) lds[i] = in[i % 1024];
  __syncthreads();
  floatx16 acc[NACC];
  for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int lane = threadIdx.x & 63, lj = lane & 31, hi = lane >> 5, wave = threadIdx.x >> 6;
  float a = in[threadIdx.x], b = in[threadIdx.x + 256];
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    } else {
      const int ko = (it & 3) * 8 + hi * 4;
      float4 av = *reinterpret_cast<const float4*>(lds + ((wave & 1) * 64 + lj) * 36 + ko);
      float4 bv[NACC];
#pragma unroll
      for (int t = 0; t < NACC; ++t) bv[t] = *reinterpret_cast<const float4*>(lds + 128 * 36 + (t * 32 + lj) * 36 + ko);
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv[t].x, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv[t].y, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv[t].z, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv[t].w, acc[t], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NACC>
void run(const char* name, int blocks, float* out, const float* in) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, NACC><<<blocks, 256>>>(out, 10, in);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE, NACC><<<blocks, 256>>>(out, iters, in);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flop = double(blocks) * 4 * iters * 4 * NACC * 4096.0;
  printf("%-28s blocks=%4d  %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, flop / ms / 1e9);
}

int main() {
  float *out, *in;
  hipMalloc(&out, 8192 * 256 * 4);
  hipMalloc(&in, 4096 * 4);
  std::vector<float> h(4096, 1.0f);
  for (int i = 0; i < 4096; ++i) h[i] = (i % 7) * 0.25f - 0.5f;
  hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  run<0, 1>("regs, 1 acc, 1 wave/SIMD", 256, out, in);
  run<0, 4>("regs, 4 acc, 1 wave/SIMD", 256, out, in);
  run<0, 4>("regs, 4 acc, 2 wave/SIMD", 512, out, in);
  run<0, 4>("regs, 4 acc, 3 wave/SIMD", 768, out, in);
  run<1, 2>("lds b128, 2 acc, 1 w/SIMD", 256, out, in);
  run<1, 2>("lds b128, 2 acc, 2 w/SIMD", 512, out, in);
  run<1, 4>("lds b128, 4 acc, 1 w/SIMD", 256, out, in);
  run<1, 4>("lds b128, 4 acc, 2 w/SIMD", 512, out, in);
  run<1, 4>("lds b128, 4 acc, 3 w/SIMD", 768, out, in);
  return 0;
}
