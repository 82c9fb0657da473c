// K1 / K2: edge-feature-aware GraphSAGE mean aggregation, forward and backward, for gfx950.
//
// Reference semantics (STEM-GNN/model/encoder.py:72-97 through PyG 2.3.0 propagate):
//   agg[i] = (1 / max(indeg(i), 1)) * sum_{e: dst(e)=i} relu(x[src(e)] + ea[e])
// The reference runs this as index_select (E x D gather) + add + relu + scatter_add_
// (atomics) + count + divide: five [E, D] HBM round trips.  Here the edges are grouped by
// destination (graph_build.hip), a group of G lanes owns one destination row and walks its
// neighbour list, gathering whole source rows with 16-byte loads (4 neighbours in flight per
// group), adding the edge term from an LDS-resident edge-type table, and reducing in
// registers: one pass, no atomics, deterministic (neighbours are summed in edge order).
//
// HBM traffic per call (algorithmic bytes):
//   E*D*4 (source rows) + A + 4E (src ids) + 4(N+1) (rowptr) + N*D*4 (output)
//   A = E*D*4 + 4E (dense edge_attr rows + edge ids)  |  4E + T*D*4 (edge-type ids + table)
#include "common.h"

#include <cstdlib>

#include <hip/hip_ext.h>
#include <mutex>
#include <utility>
#include <vector>

namespace stemgnn {
namespace {

constexpr int kBlock = 256;
constexpr int kMaxLdsTableBytes = 48 * 1024;

// Optional in-situ timing of the K1 forward launches (bench.py's roofline leg): the launch
// goes through hipExtLaunchKernelGGL with a start/stop event pair, which stamps the kernel's
// own begin/end (not marker packets around it), so the figure matches rocprofv3's.
struct K1Profile {
  std::mutex mu;
  bool enabled = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
};
K1Profile g_k1_profile;

enum EdgeMode { kNoEdge = 0, kDenseEdge = 1, kTableLds = 2, kTableGlobal = 3 };

__device__ inline float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ inline void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

__device__ inline float relu_keep_nan(float v) { return v < 0.f ? 0.f : v; }
__device__ inline float msg(float v, int relu) { return relu ? relu_keep_nan(v) : v; }


// Load the edge-type table [T, D] into LDS (all threads of the block).
__device__ inline void stage_table(float* lds, const float* __restrict__ etab, int64_t T, int64_t D) {
  const int64_t n4 = T * D / 4;
  for (int64_t i = threadIdx.x; i < n4; i += kBlock) st4(lds + 4 * i, ld4(etab + 4 * i));
  __syncthreads();
}

// Heavy rows (hubs of a skewed graph) are not walked by one group: they are cut into chunks of
// `chunk` edges that separate groups reduce into partial rows, summed per row in chunk order by a
// combine pass (deterministic: only the PLACE of a row's items in the list depends on atomics).
// The same kernels serve both passes.  ROW mode: rows with more than `skip_above` edges are
// skipped, and with `fill` set the skipping group appends the row's chunk items to the plan.
// ITEM mode (`items` set): a group's unit of work is one chunk, its output an unscaled partial row.
struct SplitArgs {
  int32_t* item_row;    // [cap_items]
  int32_t* item_beg;    // [cap_items] first CSR slot of the chunk
  int32_t* heavy_row;   // [cap_heavy]
  int32_t* heavy_span;  // [cap_heavy][2] first item, item count
  int32_t* counts;      // counts[0] = items, counts[1] = heavy rows
  float* partial;       // [cap_items][D]
  int chunk;
  int skip_above;
  int fill;
  int items;
  int cap_items;        // capacities of the plan buffers: appends past them are dropped, never written
  int cap_heavy;
};

inline SplitArgs no_split() {
  return SplitArgs{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0x7fffffff, 0, 0, 0, 0};
}

template <int G>
__device__ __forceinline__ void plan_append(const SplitArgs& sp, int row, int beg, int deg, int lane) {
  const int n = (deg + sp.chunk - 1) / sp.chunk;
  int i0 = 0, h = 0;
  if (lane == 0) {
    i0 = atomicAdd(sp.counts, n);
    h = atomicAdd(sp.counts + 1, 1);
    // The host sizes the plan so that this cannot overflow when E bounds the CSR's live edges (plan_ok); a caller
    // that under-states E gets dropped items (counts[0] / counts[1] then exceed the capacities, which the owner of
    // the plan can read back), never an out-of-bounds write.
    const bool fits = h < sp.cap_heavy && i0 + n <= sp.cap_items;
    if (h < sp.cap_heavy) {
      sp.heavy_row[h] = row;
      sp.heavy_span[2 * h] = fits ? i0 : 0;
      sp.heavy_span[2 * h + 1] = fits ? n : 0;
    }
    if (!fits) i0 = -1;
  }
  i0 = __shfl(i0, 0, G);
  if (i0 < 0) return;
  for (int c = lane; c < n; c += G) {
    sp.item_row[i0 + c] = row;
    sp.item_beg[i0 + c] = beg + c * sp.chunk;
  }
}

// ---------------------------------------------------------------------------------------
// Forward.  G lanes per destination row, each lane owns V float4 columns (col = lane + G*v).
// ---------------------------------------------------------------------------------------
template <int G, int V, int MODE, int R, int UU = 0, int XK = 0>
__global__ void __launch_bounds__(kBlock)
k_sage_agg_fwd(const float* __restrict__ x, int64_t N, int D, const int32_t* __restrict__ rowptr,
               const int32_t* __restrict__ src, const int32_t* __restrict__ aux,  // eid (dense) or etype per slot
               const float* __restrict__ edge_attr, const float* __restrict__ etab, int64_t T,
               float* __restrict__ agg, int relu, SplitArgs sp) {
  extern __shared__ __attribute__((aligned(16))) float lds_tab[];
  constexpr int kGroups = kBlock / G;
  // neighbour rows in flight per group: 4 on full graphs (8 measured slower there); UU overrides it for launches over
  // the ~1e4 active rows of a sampled batch, which are a chain of round trips and want a whole row's edges in flight
  constexpr int U = UU ? UU : (V <= 3 ? 4 : (V == 4 ? 2 : 1));
  // XK: the source rows are stored as fp32 (0) or bf16 (1, common.h: kBF16).  Compile-time, and the loaded bits stay
  // raw until the accumulate phase: a widening next to each load made the compiler wait per load (1.4x at D = 768).
  relu &= 1;
  const int lane = threadIdx.x % G;
  const int group = threadIdx.x / G;
  const int nvec = D / 4;
  // A group owns R units (rows, or chunk items of heavy rows), kGroups apart.  Prologue ordered for latency: the
  // extents of all R units and their first chunks of neighbour ids are requested BEFORE the edge-type table is
  // staged (three dependent global round trips become two), and units without edges store their zeros before the
  // block's barrier.  Every thread reaches the barrier; nobody returns above it.
  int beg[R], end[R], first_src[R], first_aux[R];
  float* out[R];
  float inv[R];
  bool live[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int64_t unit = (static_cast<int64_t>(blockIdx.x) * R + i) * kGroups + group;
    beg[i] = end[i] = 0;
    out[i] = nullptr;
    inv[i] = 1.0f;
    if (sp.items) {  // ITEM mode: one chunk of a heavy row -> unscaled partial row
      live[i] = unit < min(sp.counts[0], sp.cap_items);
      if (live[i]) {
        const int r = sp.item_row[unit];
        beg[i] = sp.item_beg[unit];
        end[i] = min(beg[i] + sp.chunk, rowptr[r + 1]);
        out[i] = sp.partial + unit * D;
      }
    } else {
      live[i] = unit < N;
      if (live[i]) {
        beg[i] = rowptr[unit];
        end[i] = rowptr[unit + 1];
        out[i] = agg + unit * D;
        const int deg = end[i] - beg[i];
        inv[i] = 1.0f / static_cast<float>(deg < 1 ? 1 : deg);
        if (deg > sp.skip_above) {  // left to the item + combine passes
          if (sp.fill) plan_append<G>(sp, static_cast<int>(unit), beg[i], deg, lane);
          live[i] = false;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < R; ++i) {
    first_src[i] = first_aux[i] = 0;
    if (live[i] && beg[i] + lane < end[i]) {
      first_src[i] = src[beg[i] + lane];
      if (MODE != kNoEdge) first_aux[i] = aux[beg[i] + lane];
    }
    if (live[i] && beg[i] == end[i]) {
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int c = lane + G * v;
        if (c < nvec) st4(out[i] + 4 * c, make_float4(0.f, 0.f, 0.f, 0.f));
      }
      live[i] = false;
    }
  }
  if (MODE == kTableLds) stage_table(lds_tab, etab, T, D);  // ends in the block's barrier

#pragma unroll
  for (int i = 0; i < R; ++i) {
    if (!live[i]) continue;
    float4 acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int base = beg[i]; base < end[i]; base += G) {
      const int cnt = min(G, end[i] - base);
      int my_src = first_src[i], my_aux = first_aux[i];
      if (base != beg[i] && lane < cnt) {
        my_src = src[base + lane];
        if (MODE != kNoEdge) my_aux = aux[base + lane];
      }
      for (int j = 0; j < cnt; j += U) {
        Raw4<XK> xv[U][V];
        float4 ev[U][V];
#pragma unroll
        for (int k = 0; k < U; ++k) {
          const int jj = (j + k < cnt) ? j + k : j;  // clamp: duplicate load, contribution masked below
          const int s = __shfl(my_src, jj, G);
          const int a = __shfl(my_aux, jj, G);
          const int64_t xr = static_cast<int64_t>(s) * D;
#pragma unroll
          for (int v = 0; v < V; ++v) {
            const int c = lane + G * v;
            if (c < nvec) {
              xv[k][v].load(x, xr + 4 * c);
              if (MODE == kDenseEdge) ev[k][v] = ld4(edge_attr + static_cast<int64_t>(a) * D + 4 * c);
              else if (MODE == kTableLds) ev[k][v] = ld4(lds_tab + static_cast<int64_t>(a) * D + 4 * c);
              else if (MODE == kTableGlobal) ev[k][v] = ld4(etab + static_cast<int64_t>(a) * D + 4 * c);
              else ev[k][v] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
          }
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
          if (j + k < cnt) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
              if (lane + G * v < nvec) {
                const float4 xw = xv[k][v].widen();
                acc[v].x += msg(xw.x + ev[k][v].x, relu);
                acc[v].y += msg(xw.y + ev[k][v].y, relu);
                acc[v].z += msg(xw.z + ev[k][v].z, relu);
                acc[v].w += msg(xw.w + ev[k][v].w, relu);
              }
            }
          }
        }
      }
    }
    const float sc = inv[i];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int c = lane + G * v;
      if (c < nvec) st4(out[i] + 4 * c, make_float4(acc[v].x * sc, acc[v].y * sc, acc[v].z * sc, acc[v].w * sc));
    }
  }
}

// out[heavy_row[h]] = scale * sum of the row's chunk partials.  One block per heavy row: kBlock/G
// sub-groups each sum a strided subset of the partials (in order), then sub-group 0 adds the
// kBlock/G sub-sums from LDS in index order -> the same bits on every run.
template <int G>
__global__ void __launch_bounds__(kBlock)
k_split_combine(const float* __restrict__ partial, const int32_t* __restrict__ heavy_row,
                const int32_t* __restrict__ heavy_span, const int32_t* __restrict__ counts,
                const int32_t* __restrict__ rowptr, int D, int mean, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds_sub[];  // [S][D]
  constexpr int S = kBlock / G;
  const int h = blockIdx.x;
  if (h >= counts[1]) return;
  const int lane = threadIdx.x % G, sub = threadIdx.x / G;
  const int row = heavy_row[h];
  const int i0 = heavy_span[2 * h], i1 = i0 + heavy_span[2 * h + 1];
  const int nvec = D / 4;
  for (int c = lane; c < nvec; c += G) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    int i = i0 + sub;
    for (; i + 3 * S < i1; i += 4 * S) {
      const float4 v0 = ld4(partial + static_cast<int64_t>(i) * D + 4 * c);
      const float4 v1 = ld4(partial + static_cast<int64_t>(i + S) * D + 4 * c);
      const float4 v2 = ld4(partial + static_cast<int64_t>(i + 2 * S) * D + 4 * c);
      const float4 v3 = ld4(partial + static_cast<int64_t>(i + 3 * S) * D + 4 * c);
      a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
      a.x += v1.x; a.y += v1.y; a.z += v1.z; a.w += v1.w;
      a.x += v2.x; a.y += v2.y; a.z += v2.z; a.w += v2.w;
      a.x += v3.x; a.y += v3.y; a.z += v3.z; a.w += v3.w;
    }
    for (; i < i1; i += S) {
      const float4 v = ld4(partial + static_cast<int64_t>(i) * D + 4 * c);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    st4(lds_sub + static_cast<int64_t>(sub) * D + 4 * c, a);
  }
  __syncthreads();
  if (sub == 0) {
    const float scale = mean ? 1.0f / static_cast<float>(rowptr[row + 1] - rowptr[row]) : 1.0f;
    for (int c = lane; c < nvec; c += G) {
      float4 a = ld4(lds_sub + 4 * c);
      for (int k = 1; k < S; ++k) {
        const float4 v = ld4(lds_sub + static_cast<int64_t>(k) * D + 4 * c);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
      st4(out + static_cast<int64_t>(row) * D + 4 * c, make_float4(a.x * scale, a.y * scale, a.z * scale, a.w * scale));
    }
  }
}

// ---------------------------------------------------------------------------------------
// Backward w.r.t. x.  Grouped by SOURCE: the group owning source row s keeps x[s] in
// registers, gathers g_agg[dst] * inv_deg[dst] per out-edge and recomputes the relu mask.
// ---------------------------------------------------------------------------------------
template <int G, int V, int MODE>
__global__ void __launch_bounds__(kBlock)
k_sage_agg_bwd(const float* __restrict__ g_agg, const float* __restrict__ x, int64_t N, int D,
               const int32_t* __restrict__ rowptr_t, const int32_t* __restrict__ dst_t,
               const int32_t* __restrict__ aux, const float* __restrict__ inv_deg,
               const float* __restrict__ edge_attr, const float* __restrict__ etab, int64_t T,
               float* __restrict__ g_x, int relu, SplitArgs sp) {
  extern __shared__ __attribute__((aligned(16))) float lds_tab[];
  constexpr int kGroups = kBlock / G;
  const int lane = threadIdx.x % G;
  const int group = threadIdx.x / G;
  const int64_t unit = static_cast<int64_t>(blockIdx.x) * kGroups + group;
  const int nvec = D / 4;
  constexpr int U = V <= 3 ? 4 : (V == 4 ? 2 : 1);
  // bit 1 of `relu`: ACCUMULATE -- g_x already holds a gradient (lin_r's backward-data product) and this pass adds
  // to it; rows without out-edges are then left alone instead of being zeroed.  Row mode only.
  const bool accumulate = (relu & 2) != 0;
  const int xk = (relu >> 2) & 1;  // bit 2: the layer input x (read for the relu mask) is stored as bf16
  relu &= 1;
  // Same latency ordering as the forward: extent, first chunk of (target, weight, type), the source row and the
  // zero rows all go out before the table is staged; nobody returns above the barrier inside stage_table.
  int64_t row = 0;
  int beg = 0, end = 0;
  float* out = nullptr;
  bool live;
  if (sp.items) {  // ITEM mode (see SplitArgs)
    live = unit < min(sp.counts[0], sp.cap_items);
    if (live) {
      row = sp.item_row[unit];
      beg = sp.item_beg[unit];
      end = min(beg + sp.chunk, rowptr_t[row + 1]);
      out = sp.partial + unit * D;
    }
  } else {
    live = unit < N;
    if (live) {
      row = unit;
      beg = rowptr_t[row];
      end = rowptr_t[row + 1];
      out = g_x + row * D;
      if (end - beg > sp.skip_above) {
        if (sp.fill) plan_append<G>(sp, static_cast<int>(row), beg, end - beg, lane);
        live = false;
      }
    }
  }
  int first_dst = 0, first_aux = 0;
  float first_w = 0.f;
  if (live && beg + lane < end) {
    first_dst = dst_t[beg + lane];
    first_w = inv_deg[first_dst];
    if (MODE != kNoEdge) first_aux = aux[beg + lane];
  }
  float4 acc[V], xs[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c = lane + G * v;
    xs[v] = (live && c < nvec && beg < end) ? ld4_kind(x, row * D + 4 * c, xk) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float4 prior[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const int c = lane + G * v;
    prior[v] = (accumulate && live && c < nvec && beg < end) ? ld4(out + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (live && beg == end) {
    if (!accumulate) {
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int c = lane + G * v;
        if (c < nvec) st4(out + 4 * c, acc[v]);
      }
    }
    live = false;
  }
  if (MODE == kTableLds) stage_table(lds_tab, etab, T, D);
  if (!live) return;

  for (int base = beg; base < end; base += G) {
    const int cnt = min(G, end - base);
    int my_dst = first_dst, my_aux = first_aux;
    float my_w = first_w;
    if (base != beg && lane < cnt) {
      my_dst = dst_t[base + lane];
      my_w = inv_deg[my_dst];
      if (MODE != kNoEdge) my_aux = aux[base + lane];
    }
    for (int j = 0; j < cnt; j += U) {
      float4 gv[U][V], ev[U][V];
      float w[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int jj = (j + k < cnt) ? j + k : j;
        const int d = __shfl(my_dst, jj, G);
        const int a = __shfl(my_aux, jj, G);
        w[k] = (j + k < cnt) ? __shfl(my_w, jj, G) : 0.f;
        const float* gr = g_agg + static_cast<int64_t>(d) * D;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const int c = lane + G * v;
          if (c < nvec) {
            gv[k][v] = ld4(gr + 4 * c);
            if (MODE == kDenseEdge) ev[k][v] = ld4(edge_attr + static_cast<int64_t>(a) * D + 4 * c);
            else if (MODE == kTableLds) ev[k][v] = ld4(lds_tab + static_cast<int64_t>(a) * D + 4 * c);
            else if (MODE == kTableGlobal) ev[k][v] = ld4(etab + static_cast<int64_t>(a) * D + 4 * c);
            else ev[k][v] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < U; ++k) {
        if (j + k < cnt) {
#pragma unroll
          for (int v = 0; v < V; ++v) {
            if (lane + G * v < nvec) {
              acc[v].x += (!relu || xs[v].x + ev[k][v].x > 0.f) ? gv[k][v].x * w[k] : 0.f;
              acc[v].y += (!relu || xs[v].y + ev[k][v].y > 0.f) ? gv[k][v].y * w[k] : 0.f;
              acc[v].z += (!relu || xs[v].z + ev[k][v].z > 0.f) ? gv[k][v].z * w[k] : 0.f;
              acc[v].w += (!relu || xs[v].w + ev[k][v].w > 0.f) ? gv[k][v].w * w[k] : 0.f;
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const int c = lane + G * v;
    if (c < nvec)
      st4(out + 4 * c, make_float4(acc[v].x + prior[v].x, acc[v].y + prior[v].y, acc[v].z + prior[v].z,
                                   acc[v].w + prior[v].w));
  }
}

struct Geometry {
  int G, V;
};

// Smallest power-of-two group (>= 8 lanes) covering D/4 float4 columns; beyond one wave,
// each lane takes V columns.
inline bool pick_geometry(int64_t D, Geometry* g) {
  if (D <= 0 || D % 4 != 0 || D > 2048) return false;
  int nvec = static_cast<int>(D / 4);
  if (nvec <= 8) { *g = {8, 1}; return true; }
  if (nvec <= 16) { *g = {16, 1}; return true; }
  if (nvec <= 32) { *g = {32, 1}; return true; }
  if (nvec <= 64) { *g = {64, 1}; return true; }
  int V = (nvec + 63) / 64;
  int Vr = V <= 2 ? 2 : V <= 3 ? 3 : V <= 4 ? 4 : 8;
  *g = {64, Vr};
  return true;
}

template <int G, int V, int MODE, int R, int UU = 0>
int launch_fwd_r(size_t lds, dim3 grid, hipStream_t st, const float* x, int64_t N, int D, const int32_t* rowptr,
                 const int32_t* src, const int32_t* aux, const float* ea, const float* etab, int64_t T, float* agg,
                 int relu, SplitArgs sp) {
  constexpr int kUnitsPerBlock = (kBlock / G) * R;
  grid.x = (grid.x + kUnitsPerBlock - 1) / kUnitsPerBlock;  // grid.x arrives as the unit count
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_k1_profile.mu);
    if (g_k1_profile.enabled && !sp.items) {
      STEMGNN_HIP_TRY(hipEventCreate(&ev0));
      STEMGNN_HIP_TRY(hipEventCreate(&ev1));
      g_k1_profile.events.emplace_back(ev0, ev1);
    }
  }
  const bool xb = (relu & 4) != 0;  // bit 2: bf16-stored source rows
  if (ev0) {
    if (xb) hipExtLaunchKernelGGL((k_sage_agg_fwd<G, V, MODE, R, UU, 1>), grid, dim3(kBlock), lds, st, ev0, ev1, 0, x, N, D,
                                  rowptr, src, aux, ea, etab, T, agg, relu, sp);
    else hipExtLaunchKernelGGL((k_sage_agg_fwd<G, V, MODE, R, UU, 0>), grid, dim3(kBlock), lds, st, ev0, ev1, 0, x, N, D,
                               rowptr, src, aux, ea, etab, T, agg, relu, sp);
  } else {
    if (xb) k_sage_agg_fwd<G, V, MODE, R, UU, 1><<<grid, kBlock, lds, st>>>(x, N, D, rowptr, src, aux, ea, etab, T, agg, relu, sp);
    else k_sage_agg_fwd<G, V, MODE, R, UU, 0><<<grid, kBlock, lds, st>>>(x, N, D, rowptr, src, aux, ea, etab, T, agg, relu, sp);
  }
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

template <int G, int V, int MODE>
int launch_fwd_one(size_t lds, dim3 units, hipStream_t st, const float* x, int64_t N, int D, const int32_t* rowptr,
                   const int32_t* src, const int32_t* aux, const float* ea, const float* etab, int64_t T,
                   float* agg, int relu, SplitArgs sp) {
  // Launch shape by row count (no split plan, D <= 256).  Measured on C4 batches (tools/k1_batch_probe.py):
  //  * launches over ALL rows of a sampled batch (1e5 rows, 89 % of them empty): two rows per group, 22.3 -> 20.1 us;
  //  * launches over the ~1.1e4 rows that can receive edges (what the encoder phase issues; every row has ~10 edges):
  //    one row per group, four source rows in flight -- 14.1 -> 12.9 us on the batch graph, 6.8 -> 5.2 us on the
  //    augmented graph (eight or twelve rows in flight and two rows per group were all slower; round 3 re-measured
  //    4 / 5 / 6 / 10 rows in flight at 12.90 / 12.81 / 13.02 / 13.96 us: the depth is not what bounds the launch);
  //  * 1e5 rows / 1e6 edges and beyond: one row per group (78.0 vs 78.5 us), also with a split plan (95 vs 110 us).
  const bool small = V == 1 && sp.counts == nullptr && N <= (1 << 18);
  if (small && N > 32768) return launch_fwd_r<G, V, MODE, (V == 1 ? 2 : 1)>(lds, units, st, x, N, D, rowptr, src, aux, ea, etab, T, agg, relu, sp);
  return launch_fwd_r<G, V, MODE, 1>(lds, units, st, x, N, D, rowptr, src, aux, ea, etab, T, agg, relu, sp);
}

template <int G, int V>
int launch_fwd_mode(int mode, size_t lds, dim3 grid, hipStream_t st, const float* x, int64_t N, int D,
                    const int32_t* rowptr, const int32_t* src, const int32_t* aux, const float* ea,
                    const float* etab, int64_t T, float* agg, int relu, SplitArgs sp) {
  switch (mode) {
    case kNoEdge: return launch_fwd_one<G, V, kNoEdge>(0, grid, st, x, N, D, rowptr, src, aux, ea, etab, T, agg, relu, sp);
    case kDenseEdge: return launch_fwd_one<G, V, kDenseEdge>(0, grid, st, x, N, D, rowptr, src, aux, ea, etab, T, agg, relu, sp);
    case kTableLds: return launch_fwd_one<G, V, kTableLds>(lds, grid, st, x, N, D, rowptr, src, aux, ea, etab, T, agg, relu, sp);
    default: return launch_fwd_one<G, V, kTableGlobal>(0, grid, st, x, N, D, rowptr, src, aux, ea, etab, T, agg, relu, sp);
  }
}

template <int G, int V>
int launch_bwd_mode(int mode, size_t lds, dim3 grid, hipStream_t st, const float* g_agg, const float* x, int64_t N,
                    int D, const int32_t* rowptr_t, const int32_t* dst_t, const int32_t* aux, const float* inv_deg,
                    const float* ea, const float* etab, int64_t T, float* g_x, int relu, SplitArgs sp) {
  switch (mode) {
    case kNoEdge: k_sage_agg_bwd<G, V, kNoEdge><<<grid, kBlock, 0, st>>>(g_agg, x, N, D, rowptr_t, dst_t, aux, inv_deg, ea, etab, T, g_x, relu, sp); break;
    case kDenseEdge: k_sage_agg_bwd<G, V, kDenseEdge><<<grid, kBlock, 0, st>>>(g_agg, x, N, D, rowptr_t, dst_t, aux, inv_deg, ea, etab, T, g_x, relu, sp); break;
    case kTableLds: k_sage_agg_bwd<G, V, kTableLds><<<grid, kBlock, lds, st>>>(g_agg, x, N, D, rowptr_t, dst_t, aux, inv_deg, ea, etab, T, g_x, relu, sp); break;
    default: k_sage_agg_bwd<G, V, kTableGlobal><<<grid, kBlock, 0, st>>>(g_agg, x, N, D, rowptr_t, dst_t, aux, inv_deg, ea, etab, T, g_x, relu, sp); break;
  }
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

#define STEMGNN_GEOM_DISPATCH(FN, ...)                                   \
  do {                                                                   \
    if (geo.G == 8) return FN<8, 1>(__VA_ARGS__);                        \
    if (geo.G == 16) return FN<16, 1>(__VA_ARGS__);                      \
    if (geo.G == 32) return FN<32, 1>(__VA_ARGS__);                      \
    if (geo.V == 1) return FN<64, 1>(__VA_ARGS__);                       \
    if (geo.V == 2) return FN<64, 2>(__VA_ARGS__);                       \
    if (geo.V == 3) return FN<64, 3>(__VA_ARGS__);                       \
    if (geo.V == 4) return FN<64, 4>(__VA_ARGS__);                       \
    return FN<64, 8>(__VA_ARGS__);                                       \
  } while (0)

inline int resolve_mode(const float* edge_attr, const float* etab, const int32_t* etype_slot, const int32_t* eid,
                        int64_t T, int64_t D, int* mode, const int32_t** aux, size_t* lds) {
  *lds = 0;
  if (edge_attr && etab) return STEMGNN_ERR_INVALID_ARG;
  if (edge_attr) {
    if (!eid) return STEMGNN_ERR_INVALID_ARG;
    *mode = kDenseEdge; *aux = eid;
  } else if (etab) {
    if (!etype_slot || T <= 0) return STEMGNN_ERR_INVALID_ARG;
    size_t bytes = static_cast<size_t>(T) * D * sizeof(float);
    *mode = bytes <= kMaxLdsTableBytes ? kTableLds : kTableGlobal;
    *lds = *mode == kTableLds ? bytes : 0;
    *aux = etype_slot;
  } else {
    *mode = kNoEdge; *aux = nullptr;
  }
  return STEMGNN_OK;
}

}  // namespace
}  // namespace stemgnn

using namespace stemgnn;

namespace {
struct SplitPlan {  // host view of the caller's split-plan buffers (see stemgnn_sage_agg_fwd_split)
  int chunk, heavy_above, build;
  int64_t cap_items, cap_heavy;
  int32_t *item_row, *item_beg, *heavy_row, *heavy_span, *counts;
  float* partial;
};

// Capacities that can never overflow: a heavy row has more than heavy_above >= chunk edges, so there
// are fewer than E/chunk of them and their chunk count is below E/chunk + (number of heavy rows).
inline bool plan_ok(const SplitPlan* p, int64_t E) {
  if (p->chunk <= 0 || p->heavy_above < p->chunk || E < 0) return false;
  if (p->cap_heavy < E / p->chunk + 1 || p->cap_items < 2 * (E / p->chunk) + 2) return false;
  if (!fits_i32(p->cap_items)) return false;
  return p->item_row && p->item_beg && p->heavy_row && p->heavy_span && p->counts && p->partial;
}

inline SplitArgs split_args(const SplitPlan* plan, int pass) {
  SplitArgs sp = no_split();
  if (!plan) return sp;
  sp.item_row = plan->item_row; sp.item_beg = plan->item_beg; sp.heavy_row = plan->heavy_row;
  sp.heavy_span = plan->heavy_span; sp.counts = plan->counts; sp.partial = plan->partial;
  sp.chunk = plan->chunk;
  sp.cap_items = static_cast<int>(plan->cap_items);
  sp.cap_heavy = static_cast<int>(plan->cap_heavy);
  if (pass == 0) { sp.skip_above = plan->heavy_above; sp.fill = plan->build; }
  else sp.items = 1;
  return sp;
}

int launch_combine(const SplitPlan* p, const int32_t* rowptr, int D, int mean, float* out, hipStream_t st) {
  const unsigned grid = static_cast<unsigned>(p->cap_heavy);
  if (D / 4 <= 32) {
    const size_t lds = static_cast<size_t>(kBlock / 32) * D * sizeof(float);
    k_split_combine<32><<<grid, kBlock, lds, st>>>(p->partial, p->heavy_row, p->heavy_span, p->counts, rowptr, D,
                                                     mean, out);
  } else {
    const size_t lds = static_cast<size_t>(kBlock / 64) * D * sizeof(float);
    k_split_combine<64><<<grid, kBlock, lds, st>>>(p->partial, p->heavy_row, p->heavy_span, p->counts, rowptr, D,
                                                     mean, out);
  }
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}
}  // namespace

extern "C" {

static int sage_agg_fwd_impl(const float* x, int64_t N, int64_t D, const int32_t* rowptr, const int32_t* src,
                             const int32_t* eid, const float* edge_attr, const float* etab,
                             const int32_t* etype_slot, int64_t T, float* agg, int relu, const SplitPlan* plan,
                             int pass, void* stream_) {
  // pass 0: rows (heavy ones skipped when a plan is given); pass 1: chunk items of the heavy rows
  hipStream_t st = static_cast<hipStream_t>(stream_);
  Geometry geo;
  if (N < 0 || !pick_geometry(D, &geo)) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(N)) return STEMGNN_ERR_TOO_LARGE;
  if (N == 0) return STEMGNN_OK;
  if (!x || !rowptr || !agg) return STEMGNN_ERR_INVALID_ARG;
  int mode; const int32_t* aux; size_t lds;
  int rc = resolve_mode(edge_attr, etab, etype_slot, eid, T, D, &mode, &aux, &lds);
  if (rc != STEMGNN_OK) return rc;
  const SplitArgs sp = split_args(plan, pass);
  const int64_t units = (plan && pass == 1) ? plan->cap_items : N;
  dim3 grid(static_cast<unsigned>(units));  // unit count; the launcher divides by its units per block
  const int Di = static_cast<int>(D);
  STEMGNN_GEOM_DISPATCH(launch_fwd_mode, mode, lds, grid, st, x, N, Di, rowptr, src, aux, edge_attr, etab, T, agg,
                        relu, sp);
}

int stemgnn_sage_agg_fwd(const float* x, int64_t N, int64_t D, const int32_t* rowptr, const int32_t* src,
                         const int32_t* eid, const float* edge_attr, const float* etab, const int32_t* etype_slot,
                         int64_t T, float* agg, void* stream_) {
  return sage_agg_fwd_impl(x, N, D, rowptr, src, eid, edge_attr, etab, etype_slot, T, agg, 1, nullptr, 0, stream_);
}

int stemgnn_mean_agg_fwd(const float* x, int64_t N, int64_t D, const int32_t* rowptr, const int32_t* src, float* agg,
                         void* stream_) {
  return sage_agg_fwd_impl(x, N, D, rowptr, src, nullptr, nullptr, nullptr, nullptr, 0, agg, 0, nullptr, 0, stream_);
}

int stemgnn_profile_k1(int enable) {
  std::lock_guard<std::mutex> lock(g_k1_profile.mu);
  g_k1_profile.enabled = enable != 0;
  return STEMGNN_OK;
}

int stemgnn_profile_k1_collect_each(float* ms_host, int64_t capacity, int64_t* launches_host) {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> evs;
  {
    std::lock_guard<std::mutex> lock(g_k1_profile.mu);
    evs.swap(g_k1_profile.events);
  }
  int64_t i = 0;
  for (auto& pr : evs) {
    STEMGNN_HIP_TRY(hipEventSynchronize(pr.second));
    float ms = 0.f;
    STEMGNN_HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
    if (ms_host && i < capacity) ms_host[i] = ms;
    ++i;
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  if (launches_host) *launches_host = i;
  return STEMGNN_OK;
}

int stemgnn_profile_k1_collect(double* total_ms_host, int64_t* launches_host) {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> evs;
  {
    std::lock_guard<std::mutex> lock(g_k1_profile.mu);
    evs.swap(g_k1_profile.events);
  }
  double total = 0.0;
  for (auto& pr : evs) {
    STEMGNN_HIP_TRY(hipEventSynchronize(pr.second));
    float ms = 0.f;
    STEMGNN_HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
    total += ms;
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  if (total_ms_host) *total_ms_host = total;
  if (launches_host) *launches_host = static_cast<int64_t>(evs.size());
  return STEMGNN_OK;
}

static int sage_agg_bwd_impl(const float* g_agg, const float* x, int64_t N, int64_t D, const int32_t* rowptr_t,
                             const int32_t* dst_t, const int32_t* eid_t, const float* inv_deg,
                             const float* edge_attr, const float* etab, const int32_t* etype_slot_t, int64_t T,
                             float* g_x, int relu, const SplitPlan* plan, int pass, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  Geometry geo;
  if (N < 0 || !pick_geometry(D, &geo)) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(N)) return STEMGNN_ERR_TOO_LARGE;
  if (N == 0) return STEMGNN_OK;
  if (!g_agg || !x || !rowptr_t || !inv_deg || !g_x) return STEMGNN_ERR_INVALID_ARG;
  int mode; const int32_t* aux; size_t lds;
  int rc = resolve_mode(edge_attr, etab, etype_slot_t, eid_t, T, D, &mode, &aux, &lds);
  if (rc != STEMGNN_OK) return rc;
  const int groups = kBlock / geo.G;
  const SplitArgs sp = split_args(plan, pass);
  const int64_t units = (plan && pass == 1) ? plan->cap_items : N;
  dim3 grid(static_cast<unsigned>((units + groups - 1) / groups));
  const int Di = static_cast<int>(D);
  STEMGNN_GEOM_DISPATCH(launch_bwd_mode, mode, lds, grid, st, g_agg, x, N, Di, rowptr_t, dst_t, aux, inv_deg,
                        edge_attr, etab, T, g_x, relu, sp);
}

int stemgnn_sage_agg_bwd(const float* g_agg, const float* x, int64_t N, int64_t D, const int32_t* rowptr_t,
                         const int32_t* dst_t, const int32_t* eid_t, const float* inv_deg, const float* edge_attr,
                         const float* etab, const int32_t* etype_slot_t, int64_t T, float* g_x, void* stream_) {
  return sage_agg_bwd_impl(g_agg, x, N, D, rowptr_t, dst_t, eid_t, inv_deg, edge_attr, etab, etype_slot_t, T, g_x, 1,
                           nullptr, 0, stream_);
}

int stemgnn_sage_agg_fwd_k(const void* x, int32_t x_kind, int64_t N, int64_t D, const int32_t* rowptr, const int32_t* src,
                           const int32_t* eid, const float* edge_attr, const float* etab, const int32_t* etype_slot,
                           int64_t T, float* agg, void* stream_) {
  if (x_kind != kF32 && x_kind != kBF16) return STEMGNN_ERR_INVALID_ARG;
  return sage_agg_fwd_impl(static_cast<const float*>(x), N, D, rowptr, src, eid, edge_attr, etab, etype_slot, T, agg,
                           1 | (x_kind << 2), nullptr, 0, stream_);
}

int stemgnn_sage_agg_bwd_acc_k(const float* g_agg, const void* x, int32_t x_kind, int64_t N, int64_t D,
                               const int32_t* rowptr_t, const int32_t* dst_t, const int32_t* eid_t, const float* inv_deg,
                               const float* edge_attr, const float* etab, const int32_t* etype_slot_t, int64_t T,
                               float* g_x, void* stream_) {
  if (x_kind != kF32 && x_kind != kBF16) return STEMGNN_ERR_INVALID_ARG;
  return sage_agg_bwd_impl(g_agg, static_cast<const float*>(x), N, D, rowptr_t, dst_t, eid_t, inv_deg, edge_attr, etab,
                           etype_slot_t, T, g_x, 1 | 2 | (x_kind << 2), nullptr, 0, stream_);
}

int stemgnn_sage_agg_bwd_acc(const float* g_agg, const float* x, int64_t N, int64_t D, const int32_t* rowptr_t,
                             const int32_t* dst_t, const int32_t* eid_t, const float* inv_deg, const float* edge_attr,
                             const float* etab, const int32_t* etype_slot_t, int64_t T, float* g_x, void* stream_) {
  return sage_agg_bwd_impl(g_agg, x, N, D, rowptr_t, dst_t, eid_t, inv_deg, edge_attr, etab, etype_slot_t, T, g_x, 1 | 2,
                           nullptr, 0, stream_);
}

int stemgnn_mean_agg_bwd(const float* g_agg, int64_t N, int64_t D, const int32_t* rowptr_t, const int32_t* dst_t,
                         const float* inv_deg, float* g_x, void* stream_) {
  // the relu mask is not evaluated: x is only dereferenced for rows with out-edges, any valid [N, D] buffer serves
  return sage_agg_bwd_impl(g_agg, g_agg, N, D, rowptr_t, dst_t, nullptr, inv_deg, nullptr, nullptr, nullptr, 0, g_x, 0,
                           nullptr, 0, stream_);
}

// ---- heavy-row splitting ---------------------------------------------------------------------
static int plan_begin(const SplitPlan* plan, int64_t E, hipStream_t st) {
  if (!plan_ok(plan, E)) return STEMGNN_ERR_INVALID_ARG;
  if (plan->build) STEMGNN_HIP_TRY(hipMemsetAsync(plan->counts, 0, 2 * sizeof(int32_t), st));
  return STEMGNN_OK;
}

int stemgnn_sage_agg_fwd_split(const float* x, int64_t N, int64_t D, int64_t E, const int32_t* rowptr,
                               const int32_t* src, const int32_t* eid, const float* edge_attr, const float* etab,
                               const int32_t* etype_slot, int64_t T, float* agg, int32_t relu, int32_t chunk,
                               int32_t heavy_above, int32_t build_plan, int64_t cap_items, int64_t cap_heavy,
                               int32_t* item_row, int32_t* item_beg, int32_t* heavy_row, int32_t* heavy_span,
                               int32_t* counts, float* partial, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  SplitPlan plan{chunk, heavy_above, build_plan, cap_items, cap_heavy, item_row, item_beg, heavy_row, heavy_span,
                 counts, partial};
  int rc = plan_begin(&plan, E, st);
  if (rc != STEMGNN_OK) return rc;
  rc = sage_agg_fwd_impl(x, N, D, rowptr, src, eid, edge_attr, etab, etype_slot, T, agg, relu, &plan, 0, stream_);
  if (rc != STEMGNN_OK || N == 0) return rc;
  rc = sage_agg_fwd_impl(x, N, D, rowptr, src, eid, edge_attr, etab, etype_slot, T, agg, relu, &plan, 1, stream_);
  if (rc != STEMGNN_OK) return rc;
  return launch_combine(&plan, rowptr, static_cast<int>(D), 1, agg, st);
}

int stemgnn_sage_agg_bwd_split(const float* g_agg, const float* x, int64_t N, int64_t D, int64_t E,
                               const int32_t* rowptr_t, const int32_t* dst_t, const int32_t* eid_t,
                               const float* inv_deg, const float* edge_attr, const float* etab,
                               const int32_t* etype_slot_t, int64_t T, float* g_x, int32_t relu, int32_t chunk,
                               int32_t heavy_above, int32_t build_plan, int64_t cap_items, int64_t cap_heavy,
                               int32_t* item_row, int32_t* item_beg, int32_t* heavy_row, int32_t* heavy_span,
                               int32_t* counts, float* partial, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  SplitPlan plan{chunk, heavy_above, build_plan, cap_items, cap_heavy, item_row, item_beg, heavy_row, heavy_span,
                 counts, partial};
  int rc = plan_begin(&plan, E, st);
  if (rc != STEMGNN_OK) return rc;
  const float* xx = relu ? x : g_agg;  // plain mean: the kernel never reads x (see stemgnn_mean_agg_bwd)
  rc = sage_agg_bwd_impl(g_agg, xx, N, D, rowptr_t, dst_t, eid_t, inv_deg, edge_attr, etab, etype_slot_t, T, g_x,
                         relu, &plan, 0, stream_);
  if (rc != STEMGNN_OK || N == 0) return rc;
  rc = sage_agg_bwd_impl(g_agg, xx, N, D, rowptr_t, dst_t, eid_t, inv_deg, edge_attr, etab, etype_slot_t, T, g_x,
                         relu, &plan, 1, stream_);
  if (rc != STEMGNN_OK) return rc;
  return launch_combine(&plan, rowptr_t, static_cast<int>(D), 0, g_x, st);
}

}  // extern "C"
