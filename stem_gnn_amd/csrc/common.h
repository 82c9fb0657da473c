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

}  // namespace stemgnn
