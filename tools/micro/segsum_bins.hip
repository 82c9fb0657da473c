// RECORD of a round-3 experiment, NOT part of libstemgnn_hip.so: the per-code segment sums of project_out's weight
// gradient (sums[h][k][:] = sum of g[n][:] over the rows n with ind[n][h] == k) WITHOUT matrix instructions.
// It was built into csrc/linear.hip beside k_code_segment_sums_cols, passed the parity / reproducibility tests, and was
// measured on a C4 batch (N = 102 400, H = 4, K = 128, D = 128): 53 us in the step against 49 us for the one-hot matrix
// form (in isolation 42 vs 49 us + 7 us of slab reduction each) -- staging 13 us, counting sort 16 us, walk 12 us, all
// LDS round trips per 128-row chunk.  Two earlier shapes (unsorted per-owner lists with an LDS read-modify-write per
// row: 78 us; bins in LDS walked bin by bin: 73 us) were slower still.  The matrix form stays.
//
// Shape: a 512-thread block owns 32 feature columns of a row split; its 64 eight-lane groups each own eight consecutive
// code bins and keep those sums in registers.  Per 128-row chunk (double-buffered in LDS) every (row, head) entry is one
// thread; entries are counting-sorted by bin (stable: wave-ballot ranks, wave and bin offsets); a group adds the first
// row of all its bins in one pass and the rest in a loop per bin.  No atomics, fixed order of additions.
#include "../../stem_gnn_amd/csrc/common.h"
namespace stemgnn {
constexpr int kSegThreads = 512, kSegMaxBins = 512;
__global__ void __launch_bounds__(kSegThreads, 4)
k_code_segment_sums_bins(const int64_t* __restrict__ ind, int H, int K, const float* __restrict__ g, int64_t M, int D,
                         int64_t rows_per_split, int R /* rows per chunk: R * H <= 512, R <= 128 */,
                         float* __restrict__ partial /*[S][H*K][D]*/) {
  extern __shared__ __attribute__((aligned(16))) unsigned char seg_smem[];
  const int NC = H * K;
  float* const stage = reinterpret_cast<float*>(seg_smem);                     // [2][R][32]
  uint16_t* const wave_cnt = reinterpret_cast<uint16_t*>(stage + 2 * R * 32);  // [8 waves][512 bins]
  uint16_t* const b_start = wave_cnt + 8 * kSegMaxBins;                        // [512]
  uint16_t* const b_total = b_start + kSegMaxBins;                             // [512]
  int* const w_sum = reinterpret_cast<int*>(b_total + kSegMaxBins);            // [8]
  uint8_t* const entries = reinterpret_cast<uint8_t*>(w_sum + 8);              // [512]: row in chunk
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, gid = tid >> 3, l8 = tid & 7;
  const int split = blockIdx.x, d0 = blockIdx.y * 32;
  const int64_t mbeg = split * rows_per_split;
  const int64_t mend = min(M, mbeg + rows_per_split);
  const int chunks = mend > mbeg ? static_cast<int>((mend - mbeg + R - 1) / R) : 0;
  float4 acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = zero4();
  const int er = tid / H, eh = tid - er * H;
  float4 rg[2];
  int rbin = -1;
  auto fetch = [&](int chunk) {
    const int64_t m0 = mbeg + static_cast<int64_t>(chunk) * R;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int idx = q * kSegThreads + tid;
      const int64_t m = m0 + (idx >> 3);
      rg[q] = (idx < R * 8 && m < mend) ? ld4(g + m * D + d0 + 4 * (idx & 7)) : zero4();
    }
    rbin = -1;
    if (er < R && m0 + er < mend) {
      const int64_t c = ind[(m0 + er) * H + eh];
      if (c >= 0 && c < K) rbin = eh * K + static_cast<int>(c);
    }
  };
  if (chunks > 0) fetch(0);
  for (int ch = 0; ch < chunks; ++ch) {
    float* const sg = stage + (ch & 1) * R * 32;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int idx = q * kSegThreads + tid;
      if (idx < R * 8) st4(sg + idx * 4, rg[q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) reinterpret_cast<uint32_t*>(wave_cnt)[q * kSegThreads + tid] = 0u;
    const int bin = rbin;
    __syncthreads();
    if (ch + 1 < chunks) fetch(ch + 1);
    const bool valid = bin >= 0;
    uint64_t same = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 9; ++b) {
      const bool bit = (bin >> b) & 1;
      const uint64_t bal = __ballot(valid && bit);
      same &= bit ? bal : ~bal;
    }
    const int rank = __popcll(same & ((1ull << lane) - 1ull));
    if (valid && rank == 0) wave_cnt[wave * kSegMaxBins + bin] = static_cast<uint16_t>(__popcll(same));
    __syncthreads();
    int run = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const int c = wave_cnt[w * kSegMaxBins + tid];
      wave_cnt[w * kSegMaxBins + tid] = static_cast<uint16_t>(run);
      run += c;
    }
    int incl = run;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) w_sum[wave] = incl;
    __syncthreads();
    int before = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) before += w < wave ? w_sum[w] : 0;
    b_start[tid] = static_cast<uint16_t>(before + incl - run);
    b_total[tid] = static_cast<uint16_t>(run);
    __syncthreads();
    if (valid) entries[b_start[bin] + wave_cnt[wave * kSegMaxBins + bin] + rank] = static_cast<uint8_t>(er);
    __syncthreads();
    const uint4 tw = *reinterpret_cast<const uint4*>(b_total + 8 * gid);
    const int n[8] = {static_cast<int>(tw.x & 0xffffu), static_cast<int>(tw.x >> 16), static_cast<int>(tw.y & 0xffffu),
                      static_cast<int>(tw.y >> 16),     static_cast<int>(tw.z & 0xffffu), static_cast<int>(tw.z >> 16),
                      static_cast<int>(tw.w & 0xffffu), static_cast<int>(tw.w >> 16)};
    int s[8];
    s[0] = b_start[8 * gid];
#pragma unroll
    for (int k = 1; k < 8; ++k) s[k] = s[k - 1] + n[k - 1];
#pragma unroll
    for (int h4 = 0; h4 < 8; h4 += 4) {
      int r0[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) r0[k] = min(static_cast<int>(entries[n[h4 + k] > 0 ? s[h4 + k] : 0]), R - 1);
      float4 v0[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v0[k] = ld4(sg + r0[k] * 32 + 4 * l8);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (n[h4 + k] > 0) {
          acc[h4 + k].x += v0[k].x; acc[h4 + k].y += v0[k].y; acc[h4 + k].z += v0[k].z; acc[h4 + k].w += v0[k].w;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      for (int i = 1; i < n[k]; ++i) {
        const float4 v = ld4(sg + static_cast<int>(entries[s[k] + i]) * 32 + 4 * l8);
        acc[k].x += v.x; acc[k].y += v.y; acc[k].z += v.z; acc[k].w += v.w;
      }
  }
  float* const pw = partial + static_cast<int64_t>(split) * NC * D + d0 + 4 * l8;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (8 * gid + k < NC) st4(pw + static_cast<int64_t>(8 * gid + k) * D, acc[k]);
}
}  // namespace stemgnn
