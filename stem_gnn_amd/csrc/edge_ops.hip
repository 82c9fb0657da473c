// K11 / K12 feed / feature lookup / K14: edge-wise gather kernels and small elementwise ops.
//
// Reference: InnerProductDecoder.forward (STEM-GNN/model/encoder.py:364-366:
// (z[ei[0]] * z[ei[1]]).sum(1)), the cat([z[u], z[v]]) feeding topo_sem_recon_decoder
// (model/pt_model.py:80), the host-side feature lookup node_text_feat[x] (pretrain.py:33-38)
// and the teacher EMA loop (model/pt_model.py:104-106).  All HBM-bound row gathers:
// one G-lane group per edge / row, 16-byte loads.
#include "common.h"

#include <atomic>
#include <cstdlib>

namespace stemgnn {
namespace {

constexpr int kBlock = 256;

__device__ inline float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ inline void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// valid endpoint pair or (-1): out-of-range edges produce 0 / are skipped (never fault).
__device__ inline bool load_edge(const int64_t* __restrict__ ei, int64_t E, int64_t e, int64_t N, int64_t* u, int64_t* v) {
  *u = ei[e];
  *v = ei[E + e];
  return *u >= 0 && *u < N && *v >= 0 && *v < N;
}

template <int G>
__global__ void __launch_bounds__(kBlock)
k_edge_dot_fwd(const float* __restrict__ z, int64_t N, int D, const int64_t* __restrict__ ei, int64_t E,
               float* __restrict__ out) {
  const int lane = threadIdx.x % G;
  const int64_t e = static_cast<int64_t>(blockIdx.x) * (kBlock / G) + threadIdx.x / G;
  if (e >= E) return;
  int64_t u, v;
  const bool ok = load_edge(ei, E, e, N, &u, &v);
  float acc = 0.f;
  if (ok) {
    const int nvec = D / 4;
    for (int c = lane; c < nvec; c += G) {
      const float4 a = ld4(z + u * D + 4 * c), b = ld4(z + v * D + 4 * c);
      acc += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
  }
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, G);
  if (lane == 0) out[e] = acc;
}

// k_edge_dot_fwd and k_edge_bce in one launch (the heads phase): a lane group scores its edges and turns each score into
// its loss term and its gradient coefficient at once; block sums -> the block that arrives last adds them in index
// order (common.h: ticket_last).  The scores themselves are not kept (the backward needs the coefficients only).
// Round 3: at most kBceBlocks blocks, each walking its share of the edges with kBceFlight edges (index pairs first, then
// all their rows) in flight per group.  One block per eight edges (2 800 blocks on a C4 batch) made 2 800 tickets on one
// counter word, which serialises at ~88 per microsecond (MI355X_MICROARCH.md, "dequeue"): 41 us for 23 MB of gathers.
constexpr int kBceBlocks = 256, kBceFlight = 4;
template <int G>
__global__ void __launch_bounds__(kBlock)
k_edge_dot_bce(const float* __restrict__ z, int64_t N, int D, const int64_t* __restrict__ ei, int64_t kp, int64_t kn,
               float* __restrict__ loss, float* __restrict__ coef, double* __restrict__ partial /*[blocks][2]*/,
               unsigned int* counter) {
  constexpr int kGroups = kBlock / G;
  __shared__ double s_terms[2][kGroups];
  __shared__ double red[2][kBlock];
  const int lane = threadIdx.x % G, grp = threadIdx.x / G;
  const int64_t E = kp + kn;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kGroups;
  const int nvec = D / 4;
  double tp = 0.0, tn = 0.0;  // this group's terms, added in the order it meets its edges (fixed for a fixed grid)
  for (int64_t e0 = static_cast<int64_t>(blockIdx.x) * kGroups + grp; e0 < E; e0 += kBceFlight * stride) {
    int64_t u[kBceFlight], v[kBceFlight];
    bool ok[kBceFlight];
#pragma unroll
    for (int j = 0; j < kBceFlight; ++j) {
      const int64_t e = e0 + j * stride;
      u[j] = v[j] = 0;
      ok[j] = e < E && load_edge(ei, E, e, N, &u[j], &v[j]);
      if (!ok[j]) u[j] = v[j] = 0;  // row 0 is read and ignored: the loads below stay unconditional
    }
    float acc[kBceFlight];
#pragma unroll
    for (int j = 0; j < kBceFlight; ++j) acc[j] = 0.f;
    for (int c = lane; c < nvec; c += G) {
      float4 a[kBceFlight], b[kBceFlight];
#pragma unroll
      for (int j = 0; j < kBceFlight; ++j) { a[j] = ld4(z + u[j] * D + 4 * c); b[j] = ld4(z + v[j] * D + 4 * c); }
#pragma unroll
      for (int j = 0; j < kBceFlight; ++j) acc[j] += a[j].x * b[j].x + a[j].y * b[j].y + a[j].z * b[j].z + a[j].w * b[j].w;
    }
#pragma unroll
    for (int j = 0; j < kBceFlight; ++j) {
      float s = ok[j] ? acc[j] : 0.f;
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, G);
      const int64_t e = e0 + j * stride;
      if (lane == 0 && e < E) {
        const float eps = 1e-15f;
        const float sg = 1.0f / (1.0f + expf(-s));
        if (e < kp) {
          tp += static_cast<double>(-logf(sg + eps));
          coef[e] = -(sg * (1.0f - sg)) / (sg + eps) / static_cast<float>(kp);
        } else {
          const float q = 1.0f - sg + eps;
          tn += static_cast<double>(-logf(q));
          coef[e] = (sg * (1.0f - sg)) / q / static_cast<float>(kn);
        }
      }
    }
  }
  if (lane == 0) {
    s_terms[0][grp] = tp;
    s_terms[1][grp] = tn;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sp = 0.0, sn = 0.0;
    for (int i = 0; i < kGroups; ++i) { sp += s_terms[0][i]; sn += s_terms[1][i]; }
    st_agent(partial + 2 * blockIdx.x, sp);
    st_agent(partial + 2 * blockIdx.x + 1, sn);
    wait_stores();
  }
  if (!ticket_last(counter)) return;
  double sp = 0.0, sn = 0.0;
  for (int i = threadIdx.x; i < static_cast<int>(gridDim.x); i += kBlock) {
    sp += ld_agent(partial + 2 * i);
    sn += ld_agent(partial + 2 * i + 1);
  }
  red[0][threadIdx.x] = sp;
  red[1][threadIdx.x] = sn;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0)
    loss[0] = static_cast<float>(red[0][0] / static_cast<double>(kp > 0 ? kp : 1) +
                                 red[1][0] / static_cast<double>(kn > 0 ? kn : 1));
}

__device__ inline void atomic_add4(float* p, float4 v) {
  atomicAdd(p + 0, v.x);
  atomicAdd(p + 1, v.y);
  atomicAdd(p + 2, v.z);
  atomicAdd(p + 3, v.w);
}

template <int G>
__global__ void __launch_bounds__(kBlock)
k_edge_dot_bwd(const float* __restrict__ g_out, const float* __restrict__ g_scalar, const float* __restrict__ z,
               int64_t N, int D, const int64_t* __restrict__ ei, int64_t E, float* __restrict__ g_z) {
  const int lane = threadIdx.x % G;
  const int64_t e = static_cast<int64_t>(blockIdx.x) * (kBlock / G) + threadIdx.x / G;
  if (e >= E) return;
  int64_t u, v;
  if (!load_edge(ei, E, e, N, &u, &v)) return;
  const float g = g_out[e] * (g_scalar ? g_scalar[0] : 1.0f);
  // one dword per lane, consecutive lanes on consecutive addresses: each atomic wave-instruction
  // covers whole 128-byte row segments (the shape the memory-side atomic units run at full rate)
  // four column steps at a time: all eight loads first, then the eight atomics (a step-by-step walk waits for each
  // pair of loads before its atomics: 0.93 of the wave cycles at a s_waitcnt)
  for (int c0 = lane; c0 < D; c0 += 4 * G) {
    float a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = c0 + q * G;
      a[q] = c < D ? z[u * D + c] : 0.f;
      b[q] = c < D ? z[v * D + c] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = c0 + q * G;
      if (c < D) {
        atomicAdd(g_z + u * D + c, g * b[q]);
        atomicAdd(g_z + v * D + c, g * a[q]);
      }
    }
  }
}

template <int G>
__global__ void __launch_bounds__(kBlock)
k_edge_concat_fwd(const float* __restrict__ z, int64_t N, int D, const int64_t* __restrict__ ei, int64_t E,
                  float* __restrict__ out) {
  const int lane = threadIdx.x % G;
  const int64_t e = static_cast<int64_t>(blockIdx.x) * (kBlock / G) + threadIdx.x / G;
  if (e >= E) return;
  int64_t u, v;
  const bool ok = load_edge(ei, E, e, N, &u, &v);
  const int nvec = D / 4;
  float* o = out + e * 2 * D;
  for (int c = lane; c < nvec; c += G) {
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    st4(o + 4 * c, ok ? ld4(z + u * D + 4 * c) : zero);
    st4(o + D + 4 * c, ok ? ld4(z + v * D + 4 * c) : zero);
  }
}

// k_edge_concat_fwd and the gather of the pairs' target rows (table[type[e]]) in one launch (the topo-sem head)
template <int G>
__global__ void __launch_bounds__(kBlock)
k_edge_concat_gather(const float* __restrict__ z, int64_t N, int D, const int64_t* __restrict__ ei, int64_t E,
                     float* __restrict__ out, const float* __restrict__ table, int64_t T,
                     const int64_t* __restrict__ type, float* __restrict__ target) {
  const int lane = threadIdx.x % G;
  const int64_t e = static_cast<int64_t>(blockIdx.x) * (kBlock / G) + threadIdx.x / G;
  if (e >= E) return;
  int64_t u, v;
  const bool ok = load_edge(ei, E, e, N, &u, &v);
  const int64_t ty = type[e];
  const bool tok = ty >= 0 && ty < T;
  const int nvec = D / 4;
  float* o = out + e * 2 * D;
  for (int c = lane; c < nvec; c += G) {
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 a = ok ? ld4(z + u * D + 4 * c) : zero, b = ok ? ld4(z + v * D + 4 * c) : zero;
    const float4 t = tok ? ld4(table + ty * D + 4 * c) : zero;
    st4(o + 4 * c, a);
    st4(o + D + 4 * c, b);
    st4(target + e * D + 4 * c, t);
  }
}

template <int G>
__global__ void __launch_bounds__(kBlock)
k_edge_concat_bwd(const float* __restrict__ g_out, int64_t N, int D, const int64_t* __restrict__ ei, int64_t E,
                  float* __restrict__ g_z, const float* __restrict__ add_a = nullptr,
                  const float* __restrict__ add_b = nullptr, int64_t add_n = 0, int edge_blocks = 0) {
  if (add_n > 0 && static_cast<int>(blockIdx.x) >= edge_blocks) {
    // extra blocks of the same launch: g_z[i] += add_a[i] + add_b[i] for the leading add_n elements (the two seed-row
    // heads' gradients), as atomics because the scatter below may hit the same rows concurrently
    for (int64_t i = static_cast<int64_t>(blockIdx.x - edge_blocks) * kBlock + threadIdx.x; i < add_n;
         i += static_cast<int64_t>(gridDim.x - edge_blocks) * kBlock)
      atomicAdd(g_z + i, add_a[i] + add_b[i]);
    return;
  }
  const int lane = threadIdx.x % G;
  const int64_t e = static_cast<int64_t>(blockIdx.x) * (kBlock / G) + threadIdx.x / G;
  if (e >= E) return;
  int64_t u, v;
  if (!load_edge(ei, E, e, N, &u, &v)) return;
  const float* g = g_out + e * 2 * D;
  for (int c0 = lane; c0 < D; c0 += 4 * G) {
    float a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = c0 + q * G;
      a[q] = c < D ? g[c] : 0.f;
      b[q] = c < D ? g[D + c] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = c0 + q * G;
      if (c < D) {
        atomicAdd(g_z + u * D + c, a[q]);
        atomicAdd(g_z + v * D + c, b[q]);
      }
    }
  }
}

template <int G>
__global__ void __launch_bounds__(kBlock)
k_gather_rows(const float* __restrict__ table, int64_t R, int D, const int64_t* __restrict__ index, int64_t n,
              float* __restrict__ out, int32_t* __restrict__ bad_count, int kind) {
  const int lane = threadIdx.x % G;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * (kBlock / G) + threadIdx.x / G;
  if (i >= n) return;
  const int64_t r = index[i];
  const bool ok = r >= 0 && r < R;
  if (!ok && bad_count != nullptr && lane == 0) atomicAdd(bad_count, 1);  // the reference's indexing raises here
  const int nvec = D / 4;
  for (int c = lane; c < nvec; c += G)  // table and output share their element kind: bf16 rows move as they are
    st4_kind(out, i * D + 4 * c, kind, ok ? ld4_kind(table, r * D + 4 * c, kind) : make_float4(0.f, 0.f, 0.f, 0.f));
}

// topo_recon_loss terms (reference model/pt_model.py:62-65, EPS = 1e-15):
//   loss = mean_{e < kp} -log(sigmoid(d_e) + EPS) + mean_{e >= kp} -log(1 - sigmoid(d_e) + EPS)
// and coef[e] = d loss / d d_e (so the backward is one scaled edge-dot scatter).
__global__ void __launch_bounds__(256) k_edge_bce(const float* __restrict__ dots, int64_t kp, int64_t kn,
                                                  float* __restrict__ loss, float* __restrict__ coef,
                                                  double* __restrict__ partial /*[blocks][2]*/, unsigned int* counter) {
  // grid-stride over the scores; the block that arrives last adds the partial sums in index order (common.h:
  // ticket_last) -- one launch, and a few hundred thousand logs no longer run on a single block
  __shared__ double red[2][256];
  const float eps = 1e-15f;
  double sp = 0.0, sn = 0.0;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; e < kp + kn;
       e += static_cast<int64_t>(gridDim.x) * 256) {
    const float d = dots[e];
    const float sg = 1.0f / (1.0f + expf(-d));
    if (e < kp) {
      sp += static_cast<double>(-logf(sg + eps));
      coef[e] = -(sg * (1.0f - sg)) / (sg + eps) / static_cast<float>(kp);
    } else {
      const float q = 1.0f - sg + eps;
      sn += static_cast<double>(-logf(q));
      coef[e] = (sg * (1.0f - sg)) / q / static_cast<float>(kn);
    }
  }
  red[0][threadIdx.x] = sp;
  red[1][threadIdx.x] = sn;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    st_agent(partial + 2 * blockIdx.x, red[0][0]);
    st_agent(partial + 2 * blockIdx.x + 1, red[1][0]);
    wait_stores();
  }
  if (!ticket_last(counter)) return;
  sp = sn = 0.0;
  for (int i = threadIdx.x; i < static_cast<int>(gridDim.x); i += 256) {
    sp += ld_agent(partial + 2 * i);
    sn += ld_agent(partial + 2 * i + 1);
  }
  red[0][threadIdx.x] = sp;
  red[1][threadIdx.x] = sn;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0)
    loss[0] = static_cast<float>(red[0][0] / static_cast<double>(kp > 0 ? kp : 1) +
                                 red[1][0] / static_cast<double>(kn > 0 ? kn : 1));
}

__global__ void __launch_bounds__(kBlock) k_ema_lerp(float* __restrict__ t, const float* __restrict__ s, int64_t n,
                                                     float decay) {
  const float w = 1.0f - decay;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * kBlock)
    t[i] = t[i] * decay + s[i] * w;  // param_k * decay + param_q * (1 - decay), pt_model.py:106
}

inline bool dim_ok(int64_t D) { return D > 0 && D % 4 == 0 && D <= 16384; }

inline unsigned groups_grid(int64_t items, int G) {
  const int per = kBlock / G;
  return static_cast<unsigned>((items + per - 1) / per);
}

// Deterministic form of the two decoder backward scatters (the default ones above add with fp32 atomics, in whatever
// order the memory system serves them): the edge list is grouped by node twice -- by its first and by its second
// endpoint row (graph_build.hip: a stable radix sort) -- and ONE lane group per node adds its incident edges'
// contributions in that order.  CONCAT = false: g_z[n] = g_scalar * sum_e coef[e] z[other endpoint] (overwrites: no
// zero pass); CONCAT = true: g_z[n] += sum over edges with n first  g_out[e][:D]  + with n second  g_out[e][D:].
template <int G, bool CONCAT>
__global__ void __launch_bounds__(kBlock)
k_edge_bwd_det(const float* __restrict__ src_rows /*coef-scaled z, or g_out*/, const float* __restrict__ coef,
               const float* __restrict__ g_scalar, int64_t N, int D, const int32_t* __restrict__ rp0,
               const int32_t* __restrict__ other0, const int32_t* __restrict__ eid0, const int32_t* __restrict__ rp1,
               const int32_t* __restrict__ other1, const int32_t* __restrict__ eid1, float* __restrict__ g_z) {
  const int lane = threadIdx.x % G;
  const int64_t n = static_cast<int64_t>(blockIdx.x) * (kBlock / G) + threadIdx.x / G;
  if (n >= N) return;
  const int nvec = D / 4;
  const float gs = (!CONCAT && g_scalar) ? g_scalar[0] : 1.f;
  for (int c = lane; c < nvec; c += G) {
    float4 acc = CONCAT ? ld4(g_z + n * D + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      const int32_t* rp = side ? rp1 : rp0;
      const int32_t* ot = side ? other1 : other0;
      const int32_t* ei = side ? eid1 : eid0;
      for (int j = rp[n]; j < rp[n + 1]; ++j) {
        float4 v;
        float w = 1.f;
        if (CONCAT) {
          v = ld4(src_rows + static_cast<int64_t>(ei[j]) * 2 * D + side * D + 4 * c);
        } else {
          v = ld4(src_rows + static_cast<int64_t>(ot[j]) * D + 4 * c);
          w = coef[ei[j]] * gs;
        }
        acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
      }
    }
    st4(g_z + n * D + 4 * c, acc);
  }
}

#define STEMGNN_EDGE_DISPATCH(KERNEL, ITEMS, ...)                                                   \
  do {                                                                                             \
    if (D / 4 <= 16) KERNEL<16><<<groups_grid(ITEMS, 16), kBlock, 0, st>>>(__VA_ARGS__);            \
    else if (D / 4 <= 32) KERNEL<32><<<groups_grid(ITEMS, 32), kBlock, 0, st>>>(__VA_ARGS__);       \
    else KERNEL<64><<<groups_grid(ITEMS, 64), kBlock, 0, st>>>(__VA_ARGS__);                        \
    STEMGNN_LAUNCH_CHECK();                                                                        \
  } while (0)

#define STEMGNN_EDGE_DISPATCH_T(KERNEL, FLAG, ITEMS, ...)                                           \
  do {                                                                                             \
    if (D / 4 <= 16) KERNEL<16, FLAG><<<groups_grid(ITEMS, 16), kBlock, 0, st>>>(__VA_ARGS__);      \
    else if (D / 4 <= 32) KERNEL<32, FLAG><<<groups_grid(ITEMS, 32), kBlock, 0, st>>>(__VA_ARGS__); \
    else KERNEL<64, FLAG><<<groups_grid(ITEMS, 64), kBlock, 0, st>>>(__VA_ARGS__);                  \
    STEMGNN_LAUNCH_CHECK();                                                                        \
  } while (0)

}  // namespace
}  // namespace stemgnn

using namespace stemgnn;

extern "C" {

int stemgnn_edge_dot_fwd(const float* z, int64_t N, int64_t D, const int64_t* edge_index, int64_t E, float* out,
                         void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (N < 0 || E < 0 || !dim_ok(D)) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(E)) return STEMGNN_ERR_TOO_LARGE;
  if (E == 0) return STEMGNN_OK;
  if (!z || !edge_index || !out) return STEMGNN_ERR_INVALID_ARG;
  STEMGNN_EDGE_DISPATCH(k_edge_dot_fwd, E, z, N, static_cast<int>(D), edge_index, E, out);
  return STEMGNN_OK;
}

int stemgnn_edge_dot_bwd(const float* g_out, const float* z, int64_t N, int64_t D, const int64_t* edge_index,
                         int64_t E, float* g_z, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (N < 0 || E < 0 || !dim_ok(D)) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(E)) return STEMGNN_ERR_TOO_LARGE;
  if (E == 0) return STEMGNN_OK;
  if (!g_out || !z || !edge_index || !g_z) return STEMGNN_ERR_INVALID_ARG;
  STEMGNN_EDGE_DISPATCH(k_edge_dot_bwd, E, g_out, nullptr, z, N, static_cast<int>(D), edge_index, E, g_z);
  return STEMGNN_OK;
}

int stemgnn_edge_dot_bwd_scaled(const float* coef, const float* g_scalar, const float* z, int64_t N, int64_t D,
                                const int64_t* edge_index, int64_t E, float* g_z, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (N < 0 || E < 0 || !dim_ok(D)) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(E)) return STEMGNN_ERR_TOO_LARGE;
  if (E == 0) return STEMGNN_OK;
  if (!coef || !g_scalar || !z || !edge_index || !g_z) return STEMGNN_ERR_INVALID_ARG;
  STEMGNN_EDGE_DISPATCH(k_edge_dot_bwd, E, coef, g_scalar, z, N, static_cast<int>(D), edge_index, E, g_z);
  return STEMGNN_OK;
}

size_t stemgnn_edge_dot_bce_workspace_bytes(int64_t num_edges) {
  if (num_edges < 0) return 0;
  return static_cast<size_t>(num_edges / 4 + 2) * 2 * sizeof(double) + 256;  // two sums per block, >= 4 edges per block
}

int stemgnn_edge_dot_bce(const float* z, int64_t N, int64_t D, const int64_t* edge_index, int64_t kp, int64_t kn,
                         float* loss, float* coef, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  const int64_t E = kp + kn;
  if (N <= 0 || kp < 0 || kn < 0 || E <= 0 || !dim_ok(D) || !loss) return STEMGNN_ERR_INVALID_ARG;  // edges need nodes
  if (!fits_i32(E)) return STEMGNN_ERR_TOO_LARGE;
  if (!z || !edge_index || !coef || !workspace) return STEMGNN_ERR_INVALID_ARG;
  if (workspace_bytes < stemgnn_edge_dot_bce_workspace_bytes(E)) return STEMGNN_ERR_WORKSPACE;
  double* partial = reinterpret_cast<double*>(align_up(reinterpret_cast<uintptr_t>(workspace), 256));
  unsigned int* counter = ticket_counter(st);
  if (!counter) return STEMGNN_ERR_HIP;
  {
    const int Di = static_cast<int>(D);
    const int G = D / 4 <= 16 ? 16 : (D / 4 <= 32 ? 32 : 64);
    int64_t grid = (E + kBlock / G - 1) / (kBlock / G);
    if (grid > kBceBlocks) grid = kBceBlocks;
    const unsigned g = static_cast<unsigned>(grid);
    if (G == 16) k_edge_dot_bce<16><<<g, kBlock, 0, st>>>(z, N, Di, edge_index, kp, kn, loss, coef, partial, counter);
    else if (G == 32) k_edge_dot_bce<32><<<g, kBlock, 0, st>>>(z, N, Di, edge_index, kp, kn, loss, coef, partial, counter);
    else k_edge_dot_bce<64><<<g, kBlock, 0, st>>>(z, N, Di, edge_index, kp, kn, loss, coef, partial, counter);
    STEMGNN_LAUNCH_CHECK();
  }
  return STEMGNN_OK;
}

int stemgnn_edge_bce_loss(const float* dots, int64_t kp, int64_t kn, float* loss, float* coef, void* stream_) {
  if (kp < 0 || kn < 0 || !loss) return STEMGNN_ERR_INVALID_ARG;
  if (kp + kn > 0 && (!dots || !coef)) return STEMGNN_ERR_INVALID_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream_);
  double* partial = stream_partials(st);  // [256 blocks][2] of the stream's own scratch (csrc/loss_ops.hip)
  unsigned int* counter = ticket_counter(st);
  if (!partial || !counter) return STEMGNN_ERR_HIP;
  int64_t blocks = (kp + kn + 1023) / 1024;
  if (blocks < 1) blocks = 1;
  if (blocks > 256) blocks = 256;
  k_edge_bce<<<static_cast<unsigned>(blocks), 256, 0, st>>>(dots, kp, kn, loss, coef, partial, counter);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_edge_concat_fwd(const float* z, int64_t N, int64_t D, const int64_t* edge_index, int64_t E, float* out,
                            void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (N < 0 || E < 0 || !dim_ok(D)) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(E)) return STEMGNN_ERR_TOO_LARGE;
  if (E == 0) return STEMGNN_OK;
  if (!z || !edge_index || !out) return STEMGNN_ERR_INVALID_ARG;
  STEMGNN_EDGE_DISPATCH(k_edge_concat_fwd, E, z, N, static_cast<int>(D), edge_index, E, out);
  return STEMGNN_OK;
}

int stemgnn_edge_concat_gather(const float* z, int64_t N, int64_t D, const int64_t* edge_index, int64_t E, float* out,
                               const float* table, int64_t T, const int64_t* type, float* target, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (N < 0 || E < 0 || T <= 0 || !dim_ok(D)) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(E)) return STEMGNN_ERR_TOO_LARGE;
  if (E == 0) return STEMGNN_OK;
  if (!z || !edge_index || !out || !table || !type || !target) return STEMGNN_ERR_INVALID_ARG;
  STEMGNN_EDGE_DISPATCH(k_edge_concat_gather, E, z, N, static_cast<int>(D), edge_index, E, out, table, T, type, target);
  return STEMGNN_OK;
}

int stemgnn_edge_concat_bwd_add(const float* g_out, int64_t N, int64_t D, const int64_t* edge_index, int64_t E,
                                float* g_z, const float* add_a, const float* add_b, int64_t add_n, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (N < 0 || E <= 0 || !dim_ok(D) || add_n <= 0 || add_n > N * D) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(E)) return STEMGNN_ERR_TOO_LARGE;
  if (!g_out || !edge_index || !g_z || !add_a || !add_b) return STEMGNN_ERR_INVALID_ARG;
  const int G = D / 4 <= 16 ? 16 : (D / 4 <= 32 ? 32 : 64);
  const int eb = static_cast<int>(groups_grid(E, G));
  int64_t ab = (add_n + kBlock - 1) / kBlock;
  if (ab > 512) ab = 512;
  const unsigned grid = static_cast<unsigned>(eb + ab);
  const int Di = static_cast<int>(D);
  if (G == 16) k_edge_concat_bwd<16><<<grid, kBlock, 0, st>>>(g_out, N, Di, edge_index, E, g_z, add_a, add_b, add_n, eb);
  else if (G == 32) k_edge_concat_bwd<32><<<grid, kBlock, 0, st>>>(g_out, N, Di, edge_index, E, g_z, add_a, add_b, add_n, eb);
  else k_edge_concat_bwd<64><<<grid, kBlock, 0, st>>>(g_out, N, Di, edge_index, E, g_z, add_a, add_b, add_n, eb);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_edge_concat_bwd(const float* g_out, int64_t N, int64_t D, const int64_t* edge_index, int64_t E,
                            float* g_z, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (N < 0 || E < 0 || !dim_ok(D)) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(E)) return STEMGNN_ERR_TOO_LARGE;
  if (E == 0) return STEMGNN_OK;
  if (!g_out || !edge_index || !g_z) return STEMGNN_ERR_INVALID_ARG;
  STEMGNN_EDGE_DISPATCH(k_edge_concat_bwd, E, g_out, N, static_cast<int>(D), edge_index, E, g_z);
  return STEMGNN_OK;
}

int stemgnn_gather_rows(const float* table, int64_t R, int64_t D, const int64_t* index, int64_t n, float* out,
                        void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (R < 0 || n < 0 || !dim_ok(D)) return STEMGNN_ERR_INVALID_ARG;
  if (n == 0) return STEMGNN_OK;
  if (!table || !index || !out) return STEMGNN_ERR_INVALID_ARG;
  STEMGNN_EDGE_DISPATCH(k_gather_rows, n, table, R, static_cast<int>(D), index, n, out,
                        static_cast<int32_t*>(nullptr), kF32);
  return STEMGNN_OK;
}

int stemgnn_gather_rows_k(const void* table, int32_t kind, int64_t R, int64_t D, const int64_t* index, int64_t n, void* out,
                          int32_t* bad_count, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (R < 0 || n < 0 || !dim_ok(D) || (kind != kF32 && kind != kBF16)) return STEMGNN_ERR_INVALID_ARG;
  if (bad_count) STEMGNN_HIP_TRY(hipMemsetAsync(bad_count, 0, sizeof(int32_t), st));
  if (n == 0) return STEMGNN_OK;
  if (!table || !index || !out) return STEMGNN_ERR_INVALID_ARG;
  STEMGNN_EDGE_DISPATCH(k_gather_rows, n, static_cast<const float*>(table), R, static_cast<int>(D), index, n,
                        static_cast<float*>(out), bad_count, kind);
  return STEMGNN_OK;
}

int stemgnn_gather_rows_checked(const float* table, int64_t R, int64_t D, const int64_t* index, int64_t n, float* out,
                                int32_t* bad_count, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (R < 0 || n < 0 || !dim_ok(D) || !bad_count) return STEMGNN_ERR_INVALID_ARG;
  STEMGNN_HIP_TRY(hipMemsetAsync(bad_count, 0, sizeof(int32_t), st));
  if (n == 0) return STEMGNN_OK;
  if (!table || !index || !out) return STEMGNN_ERR_INVALID_ARG;
  STEMGNN_EDGE_DISPATCH(k_gather_rows, n, table, R, static_cast<int>(D), index, n, out, bad_count, kF32);
  return STEMGNN_OK;
}

int stemgnn_ema_lerp(float* teacher, const float* student, int64_t n, float decay, void* stream_) {
  if (n < 0) return STEMGNN_ERR_INVALID_ARG;
  if (n == 0) return STEMGNN_OK;
  if (!teacher || !student) return STEMGNN_ERR_INVALID_ARG;
  int64_t g = (n + kBlock - 1) / kBlock;
  if (g > 4096) g = 4096;
  k_ema_lerp<<<static_cast<unsigned>(g), kBlock, 0, static_cast<hipStream_t>(stream_)>>>(teacher, student, n, decay);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}


static std::atomic<int> g_deterministic{-1};

int stemgnn_set_deterministic(int on) {
  int prev = g_deterministic.load(std::memory_order_relaxed);
  if (prev < 0) {
    const char* e = getenv("STEMGNN_DETERMINISTIC");
    prev = (e && e[0] == '1') ? 1 : 0;
    g_deterministic.store(prev, std::memory_order_relaxed);
  }
  if (on == 0 || on == 1) g_deterministic.store(on, std::memory_order_relaxed);
  return prev;
}

size_t stemgnn_edge_det_workspace_bytes(int64_t N, int64_t E) {
  if (N < 0 || E < 0) return 0;
  const size_t n1 = static_cast<size_t>(N) + 1, e1 = static_cast<size_t>(E > 0 ? E : 1);
  return 2 * ((n1 + 2 * e1) * sizeof(int32_t) + 1024) + stemgnn_csr_workspace_bytes(N, E) + 1024;
}

// mode 0: dot (coef, g_scalar, z -> g_z overwritten); mode 1: concat (g_out -> added into g_z)
static int edge_bwd_det(int mode, const float* rows, const float* coef, const float* g_scalar, int64_t N, int64_t D,
                        const int64_t* edge_index, int64_t E, float* g_z, void* workspace, size_t workspace_bytes,
                        void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (N < 0 || E < 0 || !dim_ok(D)) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(E) || !fits_i32(N)) return STEMGNN_ERR_TOO_LARGE;
  if (N == 0) return STEMGNN_OK;
  if (!rows || !g_z || (E > 0 && !edge_index) || !workspace) return STEMGNN_ERR_INVALID_ARG;
  if (workspace_bytes < stemgnn_edge_det_workspace_bytes(N, E)) return STEMGNN_ERR_WORKSPACE;
  unsigned char* w = reinterpret_cast<unsigned char*>(align_up(reinterpret_cast<uintptr_t>(workspace), 256));
  int32_t *rp[2], *other[2], *eid[2];
  const size_t e1 = static_cast<size_t>(E > 0 ? E : 1);
  for (int k = 0; k < 2; ++k) {
    rp[k] = reinterpret_cast<int32_t*>(w);
    w += align_up((static_cast<size_t>(N) + 1) * 4, 256);
    other[k] = reinterpret_cast<int32_t*>(w);
    w += align_up(e1 * 4, 256);
    eid[k] = reinterpret_cast<int32_t*>(w);
    w += align_up(e1 * 4, 256);
  }
  int32_t* bad = reinterpret_cast<int32_t*>(w);
  w += 256;
  const size_t sort_bytes = stemgnn_csr_workspace_bytes(N, E);
  for (int k = 0; k < 2; ++k) {
    const int rc = stemgnn_csr_build(edge_index, E, N, k, rp[k], other[k], eid[k], bad, w, sort_bytes, stream_);
    if (rc != STEMGNN_OK) return rc;
  }
  const int Di = static_cast<int>(D);
  if (mode == 0)
    STEMGNN_EDGE_DISPATCH_T(k_edge_bwd_det, false, N, rows, coef, g_scalar, N, Di, rp[0], other[0], eid[0], rp[1], other[1],
                            eid[1], g_z);
  else
    STEMGNN_EDGE_DISPATCH_T(k_edge_bwd_det, true, N, rows, nullptr, nullptr, N, Di, rp[0], other[0], eid[0], rp[1],
                            other[1], eid[1], g_z);
  return STEMGNN_OK;
}

int stemgnn_edge_dot_bwd_det(const float* coef, const float* g_scalar, const float* z, int64_t N, int64_t D,
                             const int64_t* edge_index, int64_t E, float* g_z, void* workspace,
                             size_t workspace_bytes, void* stream_) {
  if (E > 0 && !coef) return STEMGNN_ERR_INVALID_ARG;
  return edge_bwd_det(0, z, coef, g_scalar, N, D, edge_index, E, g_z, workspace, workspace_bytes, stream_);
}

int stemgnn_edge_concat_bwd_det(const float* g_out, int64_t N, int64_t D, const int64_t* edge_index, int64_t E,
                                float* g_z, void* workspace, size_t workspace_bytes, void* stream_) {
  return edge_bwd_det(1, g_out, nullptr, nullptr, N, D, edge_index, E, g_z, workspace, workspace_bytes, stream_);
}

}  // extern "C"
