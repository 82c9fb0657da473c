// Micro-benchmark: the bf16x3 (exact three-way split) GEMM with LDS-DMA staging and a three-stage ring.
//   A (activations, fp32): global_load_lds_dwordx4 of raw fp32 chunks (128 rows x 32 k = 16 KB per stage), XOR-swizzled on
//     the source side so the fragment reads are conflict-free; every wave splits its own fragments in registers.
//   B (weights): pre-split once into three bf16 planes [3][N][K]; a stage's 3 x 8 KB go straight into LDS by DMA.
//   The DMA is issued through inline asm: with the builtin the compiler puts s_waitcnt vmcnt(0) in front of the first
//   ds_read after it, which serialises the prefetch (tools/micro/gemm_glds.hip shows both).  One barrier per stage,
//   prefetch distance two stages, waits counted by hand.
// build: hipcc -O3 --offload-arch=gfx950 -o gemm_glds_x3 gemm_glds_x3.hip      (M must be a multiple of 128, K of 32)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kBlock = 256, kBM = 128, kBN = 128, kKC = 32;
constexpr int kStages = 3;
constexpr int kABytes = kBM * kKC * 4;      // 16 KB raw fp32
constexpr int kBPlane = kBN * kKC * 2;      // 8 KB per bf16 plane
constexpr int kStageBytes = kABytes + 3 * kBPlane;  // 40 KB
constexpr int kDmaPerWave = 4 + 6;          // A: 16 wave-instructions / 4 waves, B: 24 / 4

__device__ inline uint32_t hi16(float f) { return __float_as_uint(f) & 0xffff0000u; }
__device__ inline uint32_t pack_hi(uint32_t lo_elem, uint32_t hi_elem) {
  return __builtin_amdgcn_perm(hi_elem, lo_elem, 0x07060302u);
}

// 8 consecutive k of one row (two float4) -> the three bf16x8 fragments
__device__ inline void split8(float4 a, float4 b, bf16x8& h, bf16x8& m, bf16x8& l) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  uint32_t hb[8], mb[8], lb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    hb[i] = hi16(v[i]);
    const float r1 = v[i] - __uint_as_float(hb[i]);
    mb[i] = hi16(r1);
    lb[i] = __float_as_uint(r1 - __uint_as_float(mb[i]));
  }
  uint4 ph = make_uint4(pack_hi(hb[0], hb[1]), pack_hi(hb[2], hb[3]), pack_hi(hb[4], hb[5]), pack_hi(hb[6], hb[7]));
  uint4 pm = make_uint4(pack_hi(mb[0], mb[1]), pack_hi(mb[2], mb[3]), pack_hi(mb[4], mb[5]), pack_hi(mb[6], mb[7]));
  uint4 pl = make_uint4(pack_hi(lb[0], lb[1]), pack_hi(lb[2], lb[3]), pack_hi(lb[4], lb[5]), pack_hi(lb[6], lb[7]));
  h = *reinterpret_cast<bf16x8*>(&ph);
  m = *reinterpret_cast<bf16x8*>(&pm);
  l = *reinterpret_cast<bf16x8*>(&pl);
}

__global__ void k_split_weight(const float* __restrict__ w, int64_t n, uint16_t* __restrict__ planes) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = w[i];
  const uint32_t h = hi16(v);
  const float r1 = v - __uint_as_float(h);
  const uint32_t m = hi16(r1);
  planes[i] = static_cast<uint16_t>(h >> 16);
  planes[n + i] = static_cast<uint16_t>(m >> 16);
  planes[2 * n + i] = static_cast<uint16_t>(__float_as_uint(r1 - __uint_as_float(m)) >> 16);
}

// 16 bytes per lane, global -> LDS at (wave-uniform) lds_addr + 16 * lane
__device__ inline void dma16(const void* g, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(g) : "memory", "m0");
}

__global__ void __launch_bounds__(kBlock, 1)
k_gemm(const float* __restrict__ x, const uint16_t* __restrict__ wp, int K, int64_t M, int N, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const uint32_t smem_addr = __builtin_amdgcn_readfirstlane(
      static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) void*)smem)));
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, hi = lane >> 5, lj = lane & 31;
  const int64_t m0 = static_cast<int64_t>(blockIdx.x) * kBM;
  const int n0 = blockIdx.y * kBN;
  const int steps = K / kKC;
  const int64_t plane_elems = static_cast<int64_t>(N) * K;

  // ---- DMA sources, fixed per thread (rows) -- the k offset is added per stage
  // A: wave-instruction i = 4 t + wave (t = 0..3) covers rows 8 i .. 8 i + 7; lane -> row 8 i + (lane >> 3), LDS slot lane & 7,
  //    which must hold global 16-byte slot (lane & 7) ^ ((row >> 1) & 7)
  const float* a_src[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int r = (4 * t + wave) * 8 + (lane >> 3);
    const int slot = (lane & 7) ^ ((r >> 1) & 7);
    a_src[t] = x + (m0 + r) * K + 4 * slot;
  }
  // B plane p: wave-instruction i = 4 t + wave (t = 0..1) covers rows 16 i .. 16 i + 15 of the plane; lane -> row 16 i +
  //    (lane >> 2), LDS slot lane & 3 holding global 16-byte slot (lane & 3) ^ ((row >> 2) & 3)
  const uint16_t* b_src[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int r = (4 * t + wave) * 16 + (lane >> 2);
    const int slot = (lane & 3) ^ ((r >> 2) & 3);
    b_src[t] = wp + static_cast<int64_t>(n0 + r) * K + 8 * slot;
  }
  auto issue = [&](int step) {
    const int k0 = (step < steps ? step : steps - 1) * kKC;  // past the end: a harmless re-read keeps the count uniform
    const uint32_t base = smem_addr + (step % kStages) * kStageBytes;
#pragma unroll
    for (int t = 0; t < 4; ++t) dma16(a_src[t] + k0, base + (4 * t + wave) * 1024);
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int t = 0; t < 2; ++t) dma16(b_src[t] + p * plane_elems + k0, base + kABytes + p * kBPlane + (4 * t + wave) * 1024);
  };

  floatx16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  issue(0);
  issue(1);
  for (int step = 0; step < steps; ++step) {
    // this wave's DMA of stage `step` has landed once at most one later stage (10 instructions) is outstanding
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDmaPerWave) : "memory");
    __syncthreads();  // everyone's stage `step` is in LDS, and everyone is done reading stage step - 1
    issue(step + 2);  // into the buffer stage step - 1 occupied
    const unsigned char* st = smem + (step % kStages) * kStageBytes;
    const unsigned char* sB = st + kABytes;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ra = wm * 64 + t * 32 + lj;
        const int s0 = ks * 4 + hi * 2;  // 16-byte slots of k = 16 ks + 8 hi .. + 7
        const float4 lo = *reinterpret_cast<const float4*>(st + ra * 128 + 16 * (s0 ^ ((ra >> 1) & 7)));
        const float4 up = *reinterpret_cast<const float4*>(st + ra * 128 + 16 * ((s0 + 1) ^ ((ra >> 1) & 7)));
        split8(lo, up, a[t][0], a[t][1], a[t][2]);
        const int rb = wn * 64 + t * 32 + lj;
        const int sb = (ks * 2 + hi) ^ ((rb >> 2) & 3);
#pragma unroll
        for (int p = 0; p < 3; ++p) b[t][p] = *reinterpret_cast<const bf16x8*>(sB + p * kBPlane + rb * 64 + 16 * sb);
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
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the two dummy stages issued past the end
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int n = n0 + wn * 64 + tn * 32 + lj;
      float* yr = y + (m0 + wm * 64 + tm * 32 + 4 * hi) * N + n;
#pragma unroll
      for (int r = 0; r < 16; ++r) yr[static_cast<int64_t>((r & 3) + 8 * (r >> 2)) * N] = acc[tm][tn][r];
    }
}

int main() {
  const int64_t M = 102400;
  const int KMAX = 512, NMAX = 512;
  std::vector<float> hx(M * KMAX), hw(NMAX * KMAX);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : hx) v = rnd() * (1.0f + 3.0f * rnd() * rnd());
  for (auto& v : hw) v = rnd() * 0.2f;
  float *x, *w, *y;
  uint16_t* planes;
  hipMalloc(&x, hx.size() * 4); hipMalloc(&w, hw.size() * 4); hipMalloc(&y, M * NMAX * 4);
  hipMalloc(&planes, 3 * NMAX * KMAX * 2);
  hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  const int lds = kStages * kStageBytes;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  std::vector<float> hy(M * NMAX);
  for (auto kn : {std::pair<int, int>{128, 128}, {256, 128}, {512, 128}, {128, 512}}) {
    const int K = kn.first, N = kn.second;
    dim3 grid(static_cast<unsigned>(M / kBM), N / kBN);
    auto launch = [&]() {
      k_split_weight<<<(N * K + 255) / 256, 256>>>(w, static_cast<int64_t>(N) * K, planes);
      k_gemm<<<grid, kBlock, lds>>>(x, planes, K, M, N, y);
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 20;
    hipMemcpy(hy.data(), y, M * N * 4, hipMemcpyDeviceToHost);
    double err = 0, ref_max = 0;
    for (int64_t m = 0; m < M; m += 997)
      for (int n = 0; n < N; ++n) {
        double r = 0;
        for (int k = 0; k < K; ++k) r += static_cast<double>(hx[m * K + k]) * hw[static_cast<int64_t>(n) * K + k];
        err = fmax(err, fabs(hy[m * N + n] - r));
        ref_max = fmax(ref_max, fabs(r));
      }
    const double fl = 2.0 * M * K * N, by = (M * K + M * N + N * K) * 4.0;
    printf("M=%ld K=%d N=%d: LDS-DMA bf16x3 %7.1f us (%6.1f TF/s fp32-equivalent, %5.2f TB/s)  max|err| vs fp64 %.3g (max|y| %.3g)\n",
           static_cast<long>(M), K, N, us, fl / us / 1e6, by / us / 1e6, err, ref_max);
  }
  return 0;
}
