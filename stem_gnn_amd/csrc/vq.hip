// K6 + K7 + K8 (+ K10): cosine-similarity codebook assignment for gfx950, fused.
//
// Reference (STEM-GNN/model/vq.py): l2norm of the per-head vectors (:891), then
// CosineSimCodebook.forward (:623-688): sim = einsum('h n d, h c d -> h n c') (:650),
// argmax (:652, gumbel_sample with stochastic=False, :78), one-hot (:79) and a second
// einsum against the codebook (:657); VectorQuantize.forward adds the straight-through
// estimator (:937) and the commitment MSE (:1007-1009).  The reference materialises
// sim [H, N, K] and a same-size one-hot in HBM and spends a second 2*H*N*K*Dc-flop GEMM on
// the one-hot product.
//
// Here one kernel per call: a 128-row x (32*CG)-code tile of sim is accumulated in MFMA
// registers (v_mfma_f32_32x32x2_f32: exact fp32, as the reference forces fp32 at
// vq.py:623,634), reduced to a running arg-max per row in registers, and the winning code
// row is gathered in the epilogue.  Nothing of size [H, N, K] ever reaches memory.
// Bound: fp32 MFMA (2*H*N*K*Dc flop at 157 TFLOP/s) for K*Dc large, else HBM
// (read xp, write quant: 2*N*H*Dc*4 B).
//
// MFMA operand mapping (32x32x2 f32): A[i][k] <- codebook rows (i = code), B[k][j] <- data
// rows (j = row), so C[i][j] keeps the data row on the lane (j = lane & 31) and the 16 code
// rows of the lane's half in registers: the arg-max over codes is a per-lane scan plus one
// cross-half exchange.  Ties resolve to the lowest code index (torch.argmax semantics).
#include "common.h"

namespace stemgnn {
namespace {

constexpr int kBlock = 256;          // 4 waves; wave w owns data rows [32w, 32w+32) of the tile
constexpr int kRowsPerBlock = 128;
constexpr int kWsPartials = 1024;  // workspace floats reserved for k_vq_assign_ws: two blocks per CU, up to 512 CUs
constexpr int kKC = 32;              // k-chunk staged in LDS per step
constexpr int kPad = 4;              // row padding (floats): 36-dword stride -> conflict-free ds_read_b128
constexpr int kLd = kKC + kPad;
constexpr float kNormEps = 1e-12f;   // F.normalize eps (vq.py:28-29)


__device__ inline float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ inline void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// X3: the similarity product on the bf16 matrix cores from exact three-way operand pieces (common.h), else
// v_mfma_f32_32x32x2_f32.  Both are fp32-accurate; they round differently, so an index may differ between the
// two only where the top-2 similarity gap is at rounding level (the same band the parity tests allow against ATen).
template <int CG, bool X3>
__global__ void __launch_bounds__(kBlock, 2)
k_vq_assign(const float* __restrict__ xp, int64_t N, int H, int Dc, const float* __restrict__ embed, int K,
            int training, float* __restrict__ xn_out, float* __restrict__ norm_out, int64_t* __restrict__ ind_out,
            float* __restrict__ quant, float* __restrict__ sq_partial, const float* __restrict__ esq,
            unsigned int* counter, double sq_scale, float* __restrict__ sq_out) {
  constexpr int PA = 32 * CG * kLdP, PB = kRowsPerBlock * kLdP;  // bytes of one bf16 plane (X3)
  constexpr int kBytesA = X3 ? 3 * PA : 32 * CG * kLd * 4;
  constexpr int kBytesB = X3 ? 3 * PB : kRowsPerBlock * kLd * 4;
  __shared__ __attribute__((aligned(16))) unsigned char sA_raw[kBytesA];
  __shared__ __attribute__((aligned(16))) unsigned char sB_raw[kBytesB];
  float* sA = reinterpret_cast<float*>(sA_raw);
  float* sB = reinterpret_cast<float*>(sB_raw);
  __shared__ float s_inv[kRowsPerBlock];
  __shared__ float s_nrm[kRowsPerBlock];
  __shared__ float s_red[kBlock / kWave];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int h = blockIdx.y;
  const int64_t row0 = static_cast<int64_t>(blockIdx.x) * kRowsPerBlock;
  const int64_t HD = static_cast<int64_t>(H) * Dc;
  const int nvec = Dc / 4;
  const float* xph = xp + h * Dc;
  const float* emb = embed + static_cast<int64_t>(h) * K * Dc;

  // ---- main loop over code groups x k-chunks, register-prefetched staging
  const int kchunks = (Dc + kKC - 1) / kKC;
  const int groups = (K + 32 * CG - 1) / (32 * CG);
  const int steps = groups * kchunks;

  // The arg-max over codes is invariant to the positive per-row scale 1/||x||, so the MFMA
  // runs on the raw rows; the squared norms are accumulated on the fly while the first code
  // group's chunks are staged (each staging thread always serves the same 4 rows) and are only
  // needed by the epilogue.
  float4 ra[CG], rb[4];
  float ssq[4] = {0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int step) {
    const int g = step / kchunks, kc = step % kchunks;
    const int k0 = kc * kKC;
    // A: 32*CG code rows x 32 floats = CG*256 float4 -> CG per thread
#pragma unroll
    for (int t = 0; t < CG; ++t) {
      const int idx = t * kBlock + tid;      // float4 index in the chunk
      const int cr = idx >> 3, cc = idx & 7; // 8 float4 per row
      const int code = g * 32 * CG + cr;
      const int k = k0 + 4 * cc;
      ra[t] = (code < K && k < Dc) ? ld4(emb + static_cast<int64_t>(code) * Dc + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // B: 128 data rows x 32 floats = 1024 float4 -> 4 per thread, scaled by 1/norm
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = t * kBlock + tid;
      const int r = idx >> 3, cc = idx & 7;
      const int64_t row = row0 + r;
      const int k = k0 + 4 * cc;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < N && k < Dc) v = ld4(xph + row * HD + k);
      rb[t] = v;
      if (g == 0) ssq[t] += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int t = 0; t < CG; ++t) {
      const int idx = t * kBlock + tid;
      if (X3) {
        const int off = (idx >> 3) * kLdP + 8 * (idx & 7);
        uint2 h, m, l;
        split3(ra[t], h, m, l);
        *reinterpret_cast<uint2*>(sA_raw + off) = h;
        *reinterpret_cast<uint2*>(sA_raw + PA + off) = m;
        *reinterpret_cast<uint2*>(sA_raw + 2 * PA + off) = l;
      } else {
        st4(sA + (idx >> 3) * kLd + 4 * (idx & 7), ra[t]);
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = t * kBlock + tid;
      if (X3) {
        const int off = (idx >> 3) * kLdP + 8 * (idx & 7);
        uint2 h, m, l;
        split3(rb[t], h, m, l);
        *reinterpret_cast<uint2*>(sB_raw + off) = h;
        *reinterpret_cast<uint2*>(sB_raw + PB + off) = m;
        *reinterpret_cast<uint2*>(sB_raw + 2 * PB + off) = l;
      } else {
        st4(sB + (idx >> 3) * kLd + 4 * (idx & 7), rb[t]);
      }
    }
  };

  float best_val = -INFINITY;
  int best_idx = 0;
  const int hi = lane >> 5, lj = lane & 31;
  floatx16 acc[CG];

  fetch(0);
  for (int step = 0; step < steps; ++step) {
    const int g = step / kchunks, kc = step % kchunks;
    if (kc == 0) {
#pragma unroll
      for (int t = 0; t < CG; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    }
    stash();
    __syncthreads();
    if (step + 1 < steps) fetch(step + 1);
    if (X3) {
#pragma unroll
      for (int ks = 0; ks < kKC / 16; ++ks) {
        const int ko = ks * 32 + hi * 16;  // bytes: lane half 0 takes k 0..7, half 1 k 8..15 of the 16-wide step
        bf16x8 b[3], a[CG][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          b[p] = *reinterpret_cast<const bf16x8*>(sB_raw + p * PB + (wave * 32 + lj) * kLdP + ko);
#pragma unroll
          for (int t = 0; t < CG; ++t)
            a[t][p] = *reinterpret_cast<const bf16x8*>(sA_raw + p * PA + (t * 32 + lj) * kLdP + ko);
        }
#pragma unroll
        for (int t = 0; t < CG; ++t) acc[t] = mfma_x3(a[t], b, acc[t]);
      }
    }
    // 4 micro-steps of 8 k each: half 0 takes k 0..3, half 1 takes k 4..7 of the micro-step
#pragma unroll
    for (int ms = 0; ms < (X3 ? 0 : kKC / 8); ++ms) {
      const int ko = ms * 8 + hi * 4;
      const float4 b = ld4(sB + (wave * 32 + lj) * kLd + ko);
      float4 a[CG];
#pragma unroll
      for (int t = 0; t < CG; ++t) a[t] = ld4(sA + (t * 32 + lj) * kLd + ko);
      // k outer, code tile inner: consecutive MFMAs hit different accumulators
#pragma unroll
      for (int t = 0; t < CG; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].x, b.x, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < CG; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, b.y, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < CG; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].z, b.z, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < CG; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].w, b.w, acc[t], 0, 0, 0);
    }
    if (kc == kchunks - 1) {
      // running arg-max: codes ascend with (t, r) inside a lane; strict '>' keeps the lowest index
#pragma unroll
      for (int t = 0; t < CG; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int code = g * 32 * CG + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const float v = acc[t][r];
          if (code < K && v > best_val) { best_val = v; best_idx = code; }
        }
      }
    }
    __syncthreads();
  }
  // row norms: the 8 staging threads of a row are 8 consecutive lanes
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    float v = ssq[t];
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    if ((tid & 7) == 0) {
      const int r = (t * kBlock + tid) >> 3;
      const int64_t row = row0 + r;
      const float nrm = sqrtf(v);
      s_inv[r] = row < N ? 1.0f / fmaxf(nrm, kNormEps) : 0.f;
      s_nrm[r] = nrm;
      if (row < N) norm_out[row * H + h] = nrm;
    }
  }
  __syncthreads();
  // combine the two lane halves (same data row, disjoint code subsets)
  {
    const float ov = __shfl_xor(best_val, 32, 64);
    const int oi = __shfl_xor(best_idx, 32, 64);
    if (ov > best_val || (ov == best_val && oi < best_idx)) { best_val = ov; best_idx = oi; }
  }
  if (hi == 0) {
    const int64_t row = row0 + wave * 32 + lj;
    if (row < N) ind_out[row * H + h] = static_cast<int64_t>(best_idx);
  }

  // ---- epilogue.  With `quant` the winning code rows are gathered and the straight-through value is written
  // (callers that consume the per-head codes).  Without it nothing of size [N, H*Dc] is read again or written: the
  // commitment term follows from the arg-max itself, |q - xn|^2 = |q|^2 + |xn|^2 - 2 <q, xn>, with <q, xn> =
  // best_val / max(|x|, eps) from the fp32-accurate similarity product and |q|^2 from the per-code table `esq`.
  float sq = 0.f;
  if (quant == nullptr) {
    if (hi == 0) {
      const int r = wave * 32 + lj;
      if (row0 + r < N) {
        const float inv = s_inv[r], xn2 = s_nrm[r] * inv;  // |xn| (1 unless the row is below the eps clamp)
        sq = esq[static_cast<int64_t>(h) * K + best_idx] + xn2 * xn2 - 2.0f * best_val * inv;
      }
    }
  } else {
  // All loads of a batch are issued before its stores: vmcnt counts loads and stores in one
  // in-order queue, so a load placed behind a store would wait for that store to retire.
  constexpr int EB = 8;
  const int iters = (32 * nvec + 63) / 64;  // wave-uniform trip count: every lane takes part in the shuffles
  for (int it0 = 0; it0 < iters; it0 += EB) {
    float4 xv[EB], qv[EB];
#pragma unroll
    for (int b = 0; b < EB; ++b) {
      const int idx = (it0 + b) * 64 + lane;
      const int r = idx / nvec, c = idx - r * nvec;
      const int code = __shfl(best_idx, r & 31, 64);
      const int64_t row = row0 + wave * 32 + r;
      xv[b] = qv[b] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (it0 + b < iters && r < 32 && row < N) {
        xv[b] = ld4(xph + row * HD + 4 * c);
        qv[b] = ld4(emb + static_cast<int64_t>(code) * Dc + 4 * c);
      }
    }
#pragma unroll
    for (int b = 0; b < EB; ++b) {
      const int idx = (it0 + b) * 64 + lane;
      const int r = idx / nvec, c = idx - r * nvec;
      const int64_t row = row0 + wave * 32 + r;
      if (it0 + b < iters && r < 32 && row < N) {
        const float s = s_inv[wave * 32 + r];
        const float4 q = qv[b];
        const float4 n = make_float4(xv[b].x * s, xv[b].y * s, xv[b].z * s, xv[b].w * s);
        const float4 d = make_float4(q.x - n.x, q.y - n.y, q.z - n.z, q.w - n.w);
        sq += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
        float4 o = q;
        if (training) o = make_float4(n.x + d.x, n.y + d.y, n.z + d.z, n.w + d.w);  // x + (q - x), vq.py:937
        st4(quant + row * HD + h * Dc + 4 * c, o);
        if (xn_out) st4(xn_out + row * HD + h * Dc + 4 * c, n);
      }
    }
  }
  }
  sq = wave_sum(sq);
  if (lane == 0) s_red[wave] = sq;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < kBlock / kWave; ++w) t += s_red[w];
    st_agent(sq_partial + static_cast<int64_t>(blockIdx.y) * gridDim.x + blockIdx.x, t);
    wait_stores();
  }
  // the block that arrives last adds the blocks' sums in index order (common.h: ticket_last): no finishing launch
  if (!ticket_last(counter)) return;
  double* red = reinterpret_cast<double*>(sA_raw);  // the staging memory is free now (>= 256 doubles)
  double tot = 0.0;
  const int64_t nb = static_cast<int64_t>(gridDim.x) * gridDim.y;
  for (int64_t i = tid; i < nb; i += kBlock) tot += ld_agent(sq_partial + i);
  red[tid] = tot;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) sq_out[0] = static_cast<float>(red[0] * sq_scale);
}

// esq[c] = |embed[c]|^2, one wave per code row (the codebook drifts off the unit sphere under AdamW).
__global__ void __launch_bounds__(kBlock) k_code_sqnorm(const float* __restrict__ embed, int64_t codes, int Dc,
                                                        float* __restrict__ esq) {
  const int lane = threadIdx.x & 63;
  const int64_t c = static_cast<int64_t>(blockIdx.x) * (kBlock / kWave) + (threadIdx.x >> 6);
  if (c >= codes) return;
  float s = 0.f;
  for (int i = lane; i < Dc / 4; i += 64) {
    const float4 v = ld4(embed + c * Dc + 4 * i);
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  s = wave_sum(s);
  if (lane == 0) esq[c] = s;
}

// out[n] = bias + sum_h table[h][ind[n][h]]: project_out of the quantised heads (model/vq.py:1041) read off a
// [H, K, D] table of projected code rows (the rows of a batch take only H*K distinct values per head).
template <int G>
__global__ void __launch_bounds__(kBlock)
k_codes_project(const float* __restrict__ table, const int64_t* __restrict__ ind, const float* __restrict__ bias,
                int64_t N, int H, int K, int D, float* __restrict__ out) {
  // a lane group takes kRows rows (H <= 4: every code and every table row of the four in flight together; the walk
  // row by row and head by head is a chain of two memory latencies per head); the sum runs in head order either way
  constexpr int kRows = 4;
  const int lane = threadIdx.x % G;
  const int64_t row0 = (static_cast<int64_t>(blockIdx.x) * (kBlock / G) + threadIdx.x / G) * kRows;
  if (row0 >= N) return;
  const int nvec = D / 4;
  for (int c = lane; c < nvec; c += G) {
    const float4 bv = bias ? ld4(bias + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 a[kRows];
#pragma unroll
    for (int r = 0; r < kRows; ++r) a[r] = bv;
    for (int h0 = 0; h0 < H; h0 += 4) {
      int64_t code[kRows][4];
      float4 v[kRows][4];
#pragma unroll
      for (int r = 0; r < kRows; ++r)
#pragma unroll
        for (int u = 0; u < 4; ++u) code[r][u] = (row0 + r < N && h0 + u < H) ? ind[(row0 + r) * H + h0 + u] : 0;
#pragma unroll
      for (int r = 0; r < kRows; ++r)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (code[r][u] < 0 || code[r][u] >= K) code[r][u] = 0;
          v[r][u] = h0 + u < H ? ld4(table + (static_cast<int64_t>(h0 + u) * K + code[r][u]) * D + 4 * c)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
      for (int r = 0; r < kRows; ++r)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (h0 + u < H) { a[r].x += v[r][u].x; a[r].y += v[r][u].y; a[r].z += v[r][u].z; a[r].w += v[r][u].w; }
    }
#pragma unroll
    for (int r = 0; r < kRows; ++r)
      if (row0 + r < N) st4(out + (row0 + r) * D + 4 * c, a[r]);
  }
}

// db[d] = sum_k sums[k][d] over the K code rows of ONE head (every row of the batch belongs to exactly one code per
// head, so any head's segment sums add up to the column sums of the gradient): project_out's bias gradient.
__global__ void __launch_bounds__(kBlock) k_segment_colsum(const float* __restrict__ sums, int K, int D,
                                                           float* __restrict__ db) {
  // 64 columns x 4 row slices per block; slices added in index order (reproducible)
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, s = threadIdx.x >> 6;
  const int d = blockIdx.x * 64 + c;
  float a = 0.f;
  if (d < D) {
    int k = s;
    for (; k + 28 < K; k += 32) {  // eight rows in flight (one at a time: a chain of K / 4 latencies), same order of sums
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = sums[static_cast<int64_t>(k + 4 * u) * D + d];
#pragma unroll
      for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; k < K; k += 4) a += sums[static_cast<int64_t>(k) * D + d];
  }
  red[s][c] = a;
  __syncthreads();
  if (s == 0 && d < D) db[d] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

__global__ void __launch_bounds__(kBlock) k_sum_partials(const float* __restrict__ partial, int64_t n, double scale,
                                                         float* __restrict__ out) {
  __shared__ double red[kBlock];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += kBlock) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = static_cast<float>(red[0] * scale);
}

// Backward: one G-lane group per (row, head).
template <int G>
__global__ void __launch_bounds__(kBlock)
k_vq_assign_bwd(const float* __restrict__ g_quant, const float* __restrict__ g_loss, float coef,
                const float* __restrict__ xp, const float* __restrict__ norm, const int64_t* __restrict__ ind,
                const float* __restrict__ embed, int64_t N, int H, int Dc, int K, float* __restrict__ g_xp) {
  constexpr int kGroups = kBlock / G;
  const int lane = threadIdx.x % G;
  const int64_t item = static_cast<int64_t>(blockIdx.x) * kGroups + threadIdx.x / G;  // row * H + h
  if (item >= N * H) return;
  const int64_t row = item / H;
  const int h = static_cast<int>(item - row * H);
  const int nvec = Dc / 4;
  const int64_t off = row * static_cast<int64_t>(H) * Dc + static_cast<int64_t>(h) * Dc;
  const float nrm = norm[item];
  const bool clamped = nrm < kNormEps;
  const float inv = 1.0f / fmaxf(nrm, kNormEps);
  const float s = g_loss ? g_loss[0] * coef : 0.f;
  int64_t code = ind[item];
  if (code < 0 || code >= K) code = 0;
  const float* q = embed + (static_cast<int64_t>(h) * K + code) * Dc;
  float dot = 0.f;
  for (int c = lane; c < nvec; c += G) {
    const float4 xv = ld4(xp + off + 4 * c), gq = ld4(g_quant + off + 4 * c), qv = ld4(q + 4 * c);
    const float4 n = make_float4(xv.x * inv, xv.y * inv, xv.z * inv, xv.w * inv);
    const float4 gx = make_float4(gq.x + s * (n.x - qv.x), gq.y + s * (n.y - qv.y), gq.z + s * (n.z - qv.z),
                                  gq.w + s * (n.w - qv.w));
    dot += gx.x * n.x + gx.y * n.y + gx.z * n.z + gx.w * n.w;
  }
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o, G);
  if (clamped) dot = 0.f;  // x / eps branch of F.normalize: plain scaling
  for (int c = lane; c < nvec; c += G) {
    const float4 xv = ld4(xp + off + 4 * c), gq = ld4(g_quant + off + 4 * c), qv = ld4(q + 4 * c);
    const float4 n = make_float4(xv.x * inv, xv.y * inv, xv.z * inv, xv.w * inv);
    const float4 gx = make_float4(gq.x + s * (n.x - qv.x), gq.y + s * (n.y - qv.y), gq.z + s * (n.z - qv.z),
                                  gq.w + s * (n.w - qv.w));
    st4(g_xp + off + 4 * c, make_float4((gx.x - n.x * dot) * inv, (gx.y - n.y * dot) * inv,
                                        (gx.z - n.z * dot) * inv, (gx.w - n.w * dot) * inv));
  }
}

// LDS of k_vq_assign_bwd_fused: six operand planes of a chunk (61 440 B) during the product, then the whole 128 x 128
// fp32 tile, rows padded by four floats (67 584 B); two blocks per CU either way
constexpr int kFusedLds = 128 * (128 + 4) * 4;

// staging row of float4 slot idx (8 slots per row): the rows of every group of eight in the order 0 4 1 5 2 6 3 7, so
// that the two rows one ds_write_b64 group covers share no LDS bank (as in csrc/linear.hip)
__device__ __forceinline__ int stage_row8(int idx) {
  const int r = idx >> 3;
  return (r & ~7) | ((r & 1) << 2) | ((r >> 1) & 3);
}

// Backward with project_out's backward-data product inside: g_q = g_out W_out[:, h-block] is formed in the matrix
// cores (the tile of csrc/linear.hip's backward-data product: exact bf16 pieces, weight read as stored and transposed
// while staged) and consumed from LDS by the row-wise arithmetic of k_vq_assign_bwd -- the [N, H*Dc] gradient of the
// quantised rows (vq.py:937,1041: straight-through + project_out) is never written to or read from HBM.
// grid (row tiles, column tiles).  A column tile is 128 columns of the [N, H*Dc] gradient: `hpt` whole heads when Dc
// divides 128 (H = 4, Dc = 32: ONE tile holds all heads, so g_out is read once and no matrix work is spent on
// padding columns), else one head per tile (hpt = 1, columns Dc .. 127 of the tile are padding).
__global__ void __launch_bounds__(kBlock, 2)
k_vq_assign_bwd_fused(const float* __restrict__ g_out, int D, const float* __restrict__ w_out /*[D][H*Dc]*/,
                      const float* __restrict__ g_loss, float coef, const float* __restrict__ xp,
                      const float* __restrict__ norm, const int64_t* __restrict__ ind, const float* __restrict__ embed,
                      int64_t N, int H, int Dc, int K, float* __restrict__ g_xp, int hpt, int col_tiles) {
  constexpr int BM = 128, BN = 128;
  constexpr int PA = BM * kLdP, PB = BN * kLdP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // kFusedLds bytes: planes, then the fp32 tile
  unsigned char* const sA = smem;
  unsigned char* const sB = smem + 3 * PA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, hi = lane >> 5, lj = lane & 31;
  // 1-D grid, XCD-aware: blocks are dealt round-robin over the 8 XCDs (b and b + 8 share one and its L2), so the column
  // tiles of ONE row tile are made blocks b, b + 8, b + 16, ... -- the row tile of g_out is fetched into that L2 once
  // instead of once per head from memory (447 MB read per launch for 262 algorithmic before)
  const int b = blockIdx.x, bgrp = b / (8 * col_tiles), r = b % (8 * col_tiles);
  const int64_t row_tile = static_cast<int64_t>(bgrp) * 8 + (r % 8);
  if (row_tile * BM >= N) return;  // the grid is rounded up to whole groups
  const int h0 = (r / 8) * hpt;  // first head of the tile
  const int64_t m0 = row_tile * BM;
  const int64_t HD = static_cast<int64_t>(H) * Dc;
  const int c0 = h0 * Dc;                                    // first column of the tile
  const int ncols = min(hpt * Dc, static_cast<int>(HD) - c0);  // live columns of the tile
  const float* wh = w_out + c0;                              // column block of the tile; row stride HD
  const int steps = (D + kKC - 1) / kKC;

  float4 ra[4], rb[4];
  auto fetch = [&](int step) {
    const int k0 = step * kKC;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = t * kBlock + tid;
      const int r = stage_row8(idx), k = k0 + 4 * (idx & 7);
      const int64_t m = m0 + r;
      ra[t] = (m < N && k < D) ? ld4(g_out + m * D + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      // weight rows k0 + 4 (tid & 7) + t (contraction index d), columns 4 (tid >> 3) .. + 3 of the head
      const int kk = k0 + 4 * (tid & 7) + t, nn = 4 * (tid >> 3);
      rb[t] = (kk < D && nn < ncols) ? ld4(wh + static_cast<int64_t>(kk) * HD + nn) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stash = [&]() {
    stash_transposed(rb, sB, PB, tid);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = t * kBlock + tid;
      const int off = stage_row8(idx) * kLdP + 8 * (idx & 7);
      uint2 ph, pm, pl;
      split3(ra[t], ph, pm, pl);
      *reinterpret_cast<uint2*>(sA + off) = ph;
      *reinterpret_cast<uint2*>(sA + PA + off) = pm;
      *reinterpret_cast<uint2*>(sA + 2 * PA + off) = pl;
    }
  };

  floatx16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  fetch(0);
  for (int step = 0; step < steps; ++step) {
    stash();
    __syncthreads();
    if (step + 1 < steps) fetch(step + 1);
#pragma unroll
    for (int ks = 0; ks < kKC / 16; ++ks) {
      const int ko = ks * 32 + hi * 16;
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          a[t][p] = *reinterpret_cast<const bf16x8*>(sA + p * PA + (wm * 64 + t * 32 + lj) * kLdP + ko);
          b[t][p] = *reinterpret_cast<const bf16x8*>(sB + p * PB + (wn * 64 + t * 32 + lj) * kLdP + ko);
        }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = mfma_x3(a[tm], b[tn], acc[tm][tn]);
    }
    __syncthreads();
  }

  // ---- epilogue: the tile through LDS, 64 rows at a time; then one 32-lane group per row does what
  // k_vq_assign_bwd does with a row of g_quant.  A group serves 16 rows of the tile (8 per half); everything it needs
  // from memory is requested up front -- codes and norms of all 16 rows at once, then per half the 8 xp rows and the 8
  // code rows together -- so a block pays a handful of memory round trips, not one per row.
  constexpr int kLdT = BN + 4;
  float* tile = reinterpret_cast<float*>(smem);
  const float s = g_loss ? g_loss[0] * coef : 0.f;
  const int l32 = tid & 31, grp = tid >> 5;  // 8 groups of 32 lanes
  const bool col_ok = 4 * l32 < ncols;
  const int hl = col_ok ? h0 + (4 * l32) / Dc : h0;  // the head of this lane's four columns
  const int cl = 4 * l32 - (hl - h0) * Dc;           // ... and their place in the head
  const int lph = hpt > 1 ? Dc / 4 : 32;             // lanes that share a head (a power of two when hpt > 1)
  if (m0 + BM <= N && ncols == BN) {
    // ---- whole tiles (all but the last row tile): every load and store unconditional, so the waits are counted
    // (one predicated operation between a load and its use makes every wait a vmcnt(0), csrc/wsgemm.hip), and
    // everything the 16 rows of a lane group need -- codes, norms, xp rows, code rows -- is in flight before the one
    // barrier: the whole 128-row tile is staged at once (the accumulators' registers go to the operands), the memory
    // being sized for it.  (SQ counters before: 0.48 of the wave cycles at a s_waitcnt.)
    int code[16];
    float nrm[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t m = m0 + (i >> 3) * 64 + (i & 7) * 8 + grp;
      code[i] = static_cast<int>(ind[m * H + hl]);
      nrm[i] = norm[m * H + hl];
    }
    float4 xv[16], qv[16];
    const float* xrow = xp + (m0 + grp) * HD + c0 + 4 * l32;
#pragma unroll
    for (int i = 0; i < 16; ++i) xv[i] = ld4(xrow + static_cast<int64_t>((i >> 3) * 64 + (i & 7) * 8) * HD);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          tile[(wm * 64 + tm * 32 + 4 * hi + (r & 3) + 8 * (r >> 2)) * kLdT + wn * 64 + tn * 32 + lj] = acc[tm][tn][r];
    const float* erow = embed + static_cast<int64_t>(hl) * K * Dc + cl;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      int c = code[i];
      if (c < 0 || c >= K) c = 0;
      qv[i] = ld4(erow + static_cast<int64_t>(c) * Dc);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int rl = (i >> 3) * 64 + (i & 7) * 8 + grp;
      const float nr = nrm[i];
      const bool clamped = nr < kNormEps;
      const float inv = 1.0f / fmaxf(nr, kNormEps);
      const float4 gq = ld4(tile + rl * kLdT + 4 * l32);
      const float4 xq = xv[i], q = qv[i];
      const float4 n = make_float4(xq.x * inv, xq.y * inv, xq.z * inv, xq.w * inv);
      const float4 gx = make_float4(gq.x + s * (n.x - q.x), gq.y + s * (n.y - q.y), gq.z + s * (n.z - q.z),
                                    gq.w + s * (n.w - q.w));
      float dot = gx.x * n.x + gx.y * n.y + gx.z * n.z + gx.w * n.w;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {
        const float other = __shfl_xor(dot, o, 32);
        if (o < lph) dot += other;
      }
      if (clamped) dot = 0.f;
      st4(g_xp + (m0 + rl) * HD + c0 + 4 * l32,
          make_float4((gx.x - n.x * dot) * inv, (gx.y - n.y * dot) * inv, (gx.z - n.z * dot) * inv,
                      (gx.w - n.w * dot) * inv));
    }
    return;
  }
  int64_t code[16];
  float nrm[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int64_t m = m0 + (i >> 3) * 64 + (i & 7) * 8 + grp;
    code[i] = 0;
    nrm[i] = 1.f;
    if (m < N) {
      code[i] = ind[m * H + hl];
      nrm[i] = norm[m * H + hl];
    }
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    float4 xv[8], qv[8];
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int64_t m = m0 + half * 64 + pass * 8 + grp;
      int64_t c = code[half * 8 + pass];
      if (c < 0 || c >= K) c = 0;
      xv[pass] = qv[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < N && col_ok) {
        xv[pass] = ld4(xp + m * HD + c0 + 4 * l32);
        qv[pass] = ld4(embed + (static_cast<int64_t>(hl) * K + c) * Dc + cl);
      }
    }
    if (wm == half) {
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            tile[(tm * 32 + 4 * hi + (r & 3) + 8 * (r >> 2)) * kLdT + wn * 64 + tn * 32 + lj] = acc[tm][tn][r];
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int rl = pass * 8 + grp;
      const int64_t m = m0 + half * 64 + rl;
      const bool live = m < N;  // uniform within the 32-lane group
      const float nr = nrm[half * 8 + pass];
      const bool clamped = nr < kNormEps;
      const float inv = 1.0f / fmaxf(nr, kNormEps);
      float4 gx = make_float4(0.f, 0.f, 0.f, 0.f), n = gx;
      float dot = 0.f;
      if (live && col_ok) {
        const float4 gq = ld4(tile + rl * kLdT + 4 * l32);
        n = make_float4(xv[pass].x * inv, xv[pass].y * inv, xv[pass].z * inv, xv[pass].w * inv);
        gx = make_float4(gq.x + s * (n.x - qv[pass].x), gq.y + s * (n.y - qv[pass].y), gq.z + s * (n.z - qv[pass].z),
                         gq.w + s * (n.w - qv[pass].w));
        dot = gx.x * n.x + gx.y * n.y + gx.z * n.z + gx.w * n.w;
      }
      // the row's dot product per head: the lph lanes of a head are an aligned power-of-two run of the group
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {
        const float other = __shfl_xor(dot, o, 32);
        if (o < lph) dot += other;
      }
      if (clamped) dot = 0.f;
      if (live && col_ok)
        st4(g_xp + m * HD + c0 + 4 * l32,
            make_float4((gx.x - n.x * dot) * inv, (gx.y - n.y * dot) * inv, (gx.z - n.z * dot) * inv,
                        (gx.w - n.w * dot) * inv));
    }
    __syncthreads();
  }
}

// ---- K10 helpers -----------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_code_keys(const int64_t* __restrict__ ind, int64_t NH, int H, int K,
                                                      int32_t* __restrict__ keys) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= NH) return;
  const int h = static_cast<int>(i % H);
  int64_t c = ind[i];
  keys[i] = (c >= 0 && c < K) ? static_cast<int32_t>(h * K + c) : static_cast<int32_t>(H) * K;  // invalid -> sentinel
}

// One half-wave per (head, code) segment: ordered sum of the member rows' normalised vectors.
__global__ void __launch_bounds__(kBlock)
k_code_sums(const float* __restrict__ xp, const float* __restrict__ norm, const int32_t* __restrict__ rowptr,
            const int32_t* __restrict__ perm, int H, int Dc, int K, float* __restrict__ bins,
            float* __restrict__ embed_sum) {
  constexpr int G = 32;
  const int lane = threadIdx.x % G;
  const int seg = blockIdx.x * (kBlock / G) + threadIdx.x / G;
  if (seg >= H * K) return;
  const int h = seg / K;
  const int nvec = Dc / 4;
  const int beg = rowptr[seg], end = rowptr[seg + 1];
  if (lane == 0) bins[seg] = static_cast<float>(end - beg);
  for (int c = lane; c < nvec; c += G) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = beg; s < end; ++s) {
      const int item = perm[s];  // row * H + h
      const int64_t row = item / H;
      const float inv = 1.0f / fmaxf(norm[item], kNormEps);
      const float4 v = ld4(xp + row * static_cast<int64_t>(H) * Dc + static_cast<int64_t>(h) * Dc + 4 * c);
      acc.x += v.x * inv; acc.y += v.y * inv; acc.z += v.z * inv; acc.w += v.w * inv;
    }
    st4(embed_sum + static_cast<int64_t>(seg) * Dc + 4 * c, acc);
  }
}

inline bool vq_dims_ok(int64_t N, int64_t H, int64_t Dc, int64_t K) {
  return N >= 0 && H > 0 && H <= 65535 && Dc > 0 && Dc % 4 == 0 && Dc <= 4096 && K > 0 && K <= 65536;
}

inline int64_t row_blocks(int64_t N) { return (N + kRowsPerBlock - 1) / kRowsPerBlock; }

}  // namespace
}  // namespace stemgnn

namespace stemgnn {
// the lean assignment will run on k_vq_assign_wsp (same predicate as in vq_assign_impl below), which needs no `esq`
bool vq_assign_takes_own_sqnorm(int64_t N, int64_t H, int64_t Dc, int64_t K) {
  return stemgnn_linear_set_mode(-1) >= 1 && stemgnn_linear_set_ws(-1) > 0 && vq_assign_ws_ok(N, H, Dc, K) && linear_pair_on();
}
}  // namespace stemgnn

using namespace stemgnn;

extern "C" {

static thread_local int g_last_assign_path = 0;  // 0 none yet, 1 k_vq_assign (tile form), 2 k_vq_assign_ws, 3 k_vq_assign_wsp (pair format), 4 the big-tile core
int stemgnn_vq_assign_last_path(void) { return g_last_assign_path; }

size_t stemgnn_vq_workspace_bytes(int64_t N, int64_t H, int64_t Dc, int64_t K) {
  if (!vq_dims_ok(N, H, Dc, K)) return 0;
  // one partial per (row block, head) of the tile form; the weight-stationary form writes at most two per CU
  return static_cast<size_t>(row_blocks(N < 1 ? 1 : N) * H + kWsPartials) * sizeof(float) + 512;
}

static int vq_assign_impl(const float* xp, int64_t N, int64_t H, int64_t Dc, const float* embed, int64_t K,
                          int training, float* xn, float* norm, int64_t* ind, float* quant, const float* esq,
                          float* sqerr, float sqerr_scale, void* workspace, size_t workspace_bytes, void* stream_);

int stemgnn_vq_assign_fwd(const float* xp, int64_t N, int64_t H, int64_t Dc, const float* embed, int64_t K,
                          int training, float* xn, float* norm, int64_t* ind, float* quant, float* sqerr,
                          float sqerr_scale, void* workspace, size_t workspace_bytes, void* stream_) {
  if (!quant && N > 0) return STEMGNN_ERR_INVALID_ARG;
  return vq_assign_impl(xp, N, H, Dc, embed, K, training, xn, norm, ind, quant, nullptr, sqerr, sqerr_scale, workspace,
                        workspace_bytes, stream_);
}

int stemgnn_vq_assign_lean(const float* xp, int64_t N, int64_t H, int64_t Dc, const float* embed, const float* esq,
                           int64_t K, float* norm, int64_t* ind, float* sqerr, float sqerr_scale, void* workspace,
                           size_t workspace_bytes, void* stream_) {
  if (!esq && N > 0) return STEMGNN_ERR_INVALID_ARG;
  return vq_assign_impl(xp, N, H, Dc, embed, K, 1, nullptr, norm, ind, nullptr, esq, sqerr, sqerr_scale, workspace,
                        workspace_bytes, stream_);
}

int stemgnn_code_sqnorm(const float* embed, int64_t codes, int64_t Dc, float* esq, void* stream_) {
  if (codes < 0 || Dc <= 0 || Dc % 4 != 0) return STEMGNN_ERR_INVALID_ARG;
  if (codes == 0) return STEMGNN_OK;
  if (!embed || !esq) return STEMGNN_ERR_INVALID_ARG;
  k_code_sqnorm<<<static_cast<unsigned>((codes + 3) / 4), kBlock, 0, static_cast<hipStream_t>(stream_)>>>(
      embed, codes, static_cast<int>(Dc), esq);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_codes_project(const float* table, const int64_t* ind, const float* bias, int64_t N, int64_t H, int64_t K,
                          int64_t D, float* out, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (N < 0 || H <= 0 || K <= 0 || D <= 0 || D % 4 != 0) return STEMGNN_ERR_INVALID_ARG;
  if (N == 0) return STEMGNN_OK;
  if (!table || !ind || !out) return STEMGNN_ERR_INVALID_ARG;
  const int Hi = static_cast<int>(H), Ki = static_cast<int>(K), Di = static_cast<int>(D);
  if (D / 4 <= 16) k_codes_project<16><<<static_cast<unsigned>((N + 63) / 64), kBlock, 0, st>>>(table, ind, bias, N, Hi, Ki, Di, out);
  else if (D / 4 <= 32) k_codes_project<32><<<static_cast<unsigned>((N + 31) / 32), kBlock, 0, st>>>(table, ind, bias, N, Hi, Ki, Di, out);
  else k_codes_project<64><<<static_cast<unsigned>((N + 15) / 16), kBlock, 0, st>>>(table, ind, bias, N, Hi, Ki, Di, out);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_segment_colsum(const float* sums, int64_t K, int64_t D, float* db, void* stream_) {
  if (K <= 0 || D <= 0 || !sums || !db) return STEMGNN_ERR_INVALID_ARG;
  k_segment_colsum<<<static_cast<unsigned>((D + 63) / 64), kBlock, 0, static_cast<hipStream_t>(stream_)>>>(
      sums, static_cast<int>(K), static_cast<int>(D), db);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

static int vq_assign_impl(const float* xp, int64_t N, int64_t H, int64_t Dc, const float* embed, int64_t K,
                          int training, float* xn, float* norm, int64_t* ind, float* quant, const float* esq,
                          float* sqerr, float sqerr_scale, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (!vq_dims_ok(N, H, Dc, K)) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(N) || !fits_i32(N * H)) return STEMGNN_ERR_TOO_LARGE;
  if (!sqerr) return STEMGNN_ERR_INVALID_ARG;
  if (N == 0) {
    STEMGNN_HIP_TRY(hipMemsetAsync(sqerr, 0, sizeof(float), st));
    return STEMGNN_OK;
  }
  if (!xp || !embed || !norm || !ind || (!quant && !esq) || !workspace) return STEMGNN_ERR_INVALID_ARG;
  if (workspace_bytes < stemgnn_vq_workspace_bytes(N, H, Dc, K)) return STEMGNN_ERR_WORKSPACE;
  float* partial = reinterpret_cast<float*>(align_up(reinterpret_cast<uintptr_t>(workspace), 256));
  const int64_t rb = row_blocks(N);
  dim3 grid(static_cast<unsigned>(rb), static_cast<unsigned>(H));
  const int Hi = static_cast<int>(H), Dci = static_cast<int>(Dc), Ki = static_cast<int>(K);
  // the similarity product takes the exact three-piece form in the bf16 GEMM mode too (vq.py:623,634: fp32 forced there)
  const bool x3 = stemgnn_linear_set_mode(-1) >= 1;
  unsigned int* counter = ticket_counter(st);
  if (!counter) return STEMGNN_ERR_HIP;
  const double sq_scale = static_cast<double>(sqerr_scale);
  // lean form at K = Dc = 128: the weight-stationary kernel (csrc/wsgemm.hip), same results
  if (x3 && !quant && esq && stemgnn_linear_set_ws(-1) > 0 && vq_assign_ws_ok(N, H, Dc, K))
  {
    if (linear_pair_on()) {  // the pair format: three matrix passes (csrc/wspair.hip)
      g_last_assign_path = 3;
      return vq_assign_wsp_launch(xp, N, H, embed, esq, norm, ind, partial, counter, sq_scale, sqerr, st);
    }
    g_last_assign_path = 2;
    return vq_assign_ws_launch(xp, N, H, embed, esq, norm, ind, partial, counter, sq_scale, sqerr, st);
  }
  // a large codebook: the exact six-piece similarity product on the big-tile core, arg-max from the accumulators
  // (csrc/bigtile.hip); without a scratch arena (the only miss that is not an error) the tile form below serves
  if (x3 && bt_vq_assign_ok(N, H, Dc, K)) {
    const int rc = bt_vq_assign(xp, N, H, Dc, embed, esq, K, training, xn, norm, ind, quant, sqerr, sq_scale, st);
    if (rc == STEMGNN_OK) {
      g_last_assign_path = 4;
      bt_served();
      return STEMGNN_OK;
    }
    if (rc != STEMGNN_ERR_WORKSPACE) return rc;
    bt_missed();
  }
  g_last_assign_path = 1;
#define STEMGNN_VQ_LAUNCH(CG)                                                                                         \
  do {                                                                                                                \
    if (x3) k_vq_assign<CG, true><<<grid, kBlock, 0, st>>>(xp, N, Hi, Dci, embed, Ki, training, xn, norm, ind, quant, \
                                                            partial, esq, counter, sq_scale, sqerr);                  \
    else k_vq_assign<CG, false><<<grid, kBlock, 0, st>>>(xp, N, Hi, Dci, embed, Ki, training, xn, norm, ind, quant,   \
                                                          partial, esq, counter, sq_scale, sqerr);                    \
  } while (0)
  if (K <= 32) STEMGNN_VQ_LAUNCH(1);
  else if (K <= 64) STEMGNN_VQ_LAUNCH(2);
  else STEMGNN_VQ_LAUNCH(4);
#undef STEMGNN_VQ_LAUNCH
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_vq_assign_bwd(const float* g_quant, const float* g_loss, float commit_weight, const float* xp,
                          const float* norm, const int64_t* ind, const float* embed, int64_t N, int64_t H,
                          int64_t Dc, int64_t K, float* g_xp, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (!vq_dims_ok(N, H, Dc, K)) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(N * H)) return STEMGNN_ERR_TOO_LARGE;
  if (N == 0) return STEMGNN_OK;
  if (!g_quant || !xp || !norm || !ind || !embed || !g_xp) return STEMGNN_ERR_INVALID_ARG;
  // d/dxn of commit_weight * mean((q - xn)^2) = commit_weight * 2 (xn - q) / (N*H*Dc)
  const float coef = commit_weight * 2.0f / static_cast<float>(static_cast<double>(N) * H * Dc);
  const int64_t items = N * H;
  if (Dc / 4 <= 32) {
    const int groups = kBlock / 32;
    k_vq_assign_bwd<32><<<static_cast<unsigned>((items + groups - 1) / groups), kBlock, 0, st>>>(
        g_quant, g_loss, coef, xp, norm, ind, embed, N, static_cast<int>(H), static_cast<int>(Dc),
        static_cast<int>(K), g_xp);
  } else {
    const int groups = kBlock / 64;
    k_vq_assign_bwd<64><<<static_cast<unsigned>((items + groups - 1) / groups), kBlock, 0, st>>>(
        g_quant, g_loss, coef, xp, norm, ind, embed, N, static_cast<int>(H), static_cast<int>(Dc),
        static_cast<int>(K), g_xp);
  }
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_vq_assign_bwd_fused(const float* g_out, int64_t D, const float* w_out, const float* g_loss,
                                float commit_weight, const float* xp, const float* norm, const int64_t* ind,
                                const float* embed, int64_t N, int64_t H, int64_t Dc, int64_t K, float* g_xp,
                                void* stream_) {
  return stemgnn::vq_assign_bwd_fused_rowmax(g_out, D, w_out, g_loss, commit_weight, xp, norm, ind, embed, N, H, Dc, K, g_xp,
                                             nullptr, nullptr, stream_);
}
}  // extern "C"

namespace stemgnn {
// *rowmax_written (may be null) tells whether `rowmax` [N][H] (may be null) received the largest magnitude of every
// (row, head) stretch of g_xp: only the pair-format kernel writes it
int vq_assign_bwd_fused_rowmax(const float* g_out, int64_t D, const float* w_out, const float* g_loss, float commit_weight,
                               const float* xp, const float* norm, const int64_t* ind, const float* embed, int64_t N,
                               int64_t H, int64_t Dc, int64_t K, float* g_xp, float* rowmax, bool* rowmax_written,
                               void* stream_) {
  if (rowmax_written) *rowmax_written = false;
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (!vq_dims_ok(N, H, Dc, K) || Dc > 128 || D <= 0 || D % 4 != 0 || D > 65536) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(N * H)) return STEMGNN_ERR_TOO_LARGE;
  if (N == 0) return STEMGNN_OK;
  if (!g_out || !w_out || !xp || !norm || !ind || !embed || !g_xp) return STEMGNN_ERR_INVALID_ARG;
  const float coef = commit_weight * 2.0f / static_cast<float>(static_cast<double>(N) * H * Dc);
  // D = Dc = 128: the product in the pair format on the weight-stationary skeleton (csrc/wspair.hip)
  if (linear_pair_on() && vq_bwd_wsp_ok(N, D, H, Dc)) {
    if (rowmax_written) *rowmax_written = rowmax != nullptr;
    return vq_bwd_wsp_launch(g_out, w_out, g_loss, coef, xp, norm, ind, embed, N, H, K, g_xp, rowmax, st);
  }
  // heads per 128-column tile: whole heads when Dc divides 128 (then Dc / 4 lanes per head is a power of two)
  const int hpt = (128 % Dc == 0) ? static_cast<int>(std::min<int64_t>(H, 128 / Dc)) : 1;
  const int col_tiles = static_cast<int>((H + hpt - 1) / hpt);
  const int64_t groups = (row_blocks(N) + 7) / 8;  // 8 row tiles (one per XCD) x col_tiles blocks each
  if (groups * 8 * col_tiles >= (1ll << 31)) return STEMGNN_ERR_TOO_LARGE;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_vq_assign_bwd_fused),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, kFusedLds);
  if (attr != hipSuccess) return STEMGNN_ERR_HIP;
  k_vq_assign_bwd_fused<<<static_cast<unsigned>(groups * 8 * col_tiles), kBlock, kFusedLds, st>>>(
      g_out, static_cast<int>(D), w_out, g_loss, coef, xp, norm, ind, embed, N, static_cast<int>(H), static_cast<int>(Dc),
      static_cast<int>(K), g_xp, hpt, col_tiles);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}
}  // namespace stemgnn

extern "C" {

size_t stemgnn_vq_ema_workspace_bytes(int64_t N, int64_t H, int64_t Dc, int64_t K) {
  if (!vq_dims_ok(N, H, Dc, K)) return 0;
  const int64_t items = (N < 1 ? 1 : N) * H;
  size_t keys = align_up(static_cast<size_t>(items) * sizeof(int32_t), 256);
  size_t rowptr = align_up(static_cast<size_t>(H * K + 2) * sizeof(int32_t), 256);
  return 2 * keys + rowptr + stemgnn_csr_workspace_bytes(H * K, items) + 512;
}

int stemgnn_vq_ema_stats(const float* xp, const float* norm, const int64_t* ind, int64_t N, int64_t H, int64_t Dc,
                         int64_t K, float* bins, float* embed_sum, void* workspace, size_t workspace_bytes,
                         void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (!vq_dims_ok(N, H, Dc, K) || !bins || !embed_sum) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(N * H) || !fits_i32(H * K)) return STEMGNN_ERR_TOO_LARGE;
  if (N == 0) {
    STEMGNN_HIP_TRY(hipMemsetAsync(bins, 0, sizeof(float) * H * K, st));
    STEMGNN_HIP_TRY(hipMemsetAsync(embed_sum, 0, sizeof(float) * H * K * Dc, st));
    return STEMGNN_OK;
  }
  if (!xp || !norm || !ind || !workspace) return STEMGNN_ERR_INVALID_ARG;
  if (workspace_bytes < stemgnn_vq_ema_workspace_bytes(N, H, Dc, K)) return STEMGNN_ERR_WORKSPACE;
  const int64_t items = N * H;
  uintptr_t base = align_up(reinterpret_cast<uintptr_t>(workspace), 256);
  size_t keys_b = align_up(static_cast<size_t>(items) * sizeof(int32_t), 256);
  size_t rowptr_b = align_up(static_cast<size_t>(H * K + 2) * sizeof(int32_t), 256);
  int32_t* keys = reinterpret_cast<int32_t*>(base);
  int32_t* perm = reinterpret_cast<int32_t*>(base + keys_b);
  int32_t* rowptr = reinterpret_cast<int32_t*>(base + 2 * keys_b);
  void* sort_ws = reinterpret_cast<void*>(base + 2 * keys_b + rowptr_b);
  size_t sort_ws_bytes = workspace_bytes - (base - reinterpret_cast<uintptr_t>(workspace)) - 2 * keys_b - rowptr_b;
  k_code_keys<<<static_cast<unsigned>((items + kBlock - 1) / kBlock), kBlock, 0, st>>>(ind, items, static_cast<int>(H),
                                                                                       static_cast<int>(K), keys);
  STEMGNN_LAUNCH_CHECK();
  int rc = stemgnn_group_by_key(keys, items, H * K, rowptr, perm, sort_ws, sort_ws_bytes, stream_);
  if (rc != STEMGNN_OK) return rc;
  const int segs = static_cast<int>(H * K);
  k_code_sums<<<(segs + 7) / 8, kBlock, 0, st>>>(xp, norm, rowptr, perm, static_cast<int>(H), static_cast<int>(Dc),
                                                 static_cast<int>(K), bins, embed_sum);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

}  // extern "C"
