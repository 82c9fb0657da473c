// GPU-resident mini-batch neighbour sampler.
//
// Replaces torch_geometric.loader.NeighborLoader(num_neighbors=[f]*L) as the reference uses it
// (STEM-GNN/pretrain.py:151-153; pyg-lib / torch-sparse C++ sampler on the host): per hop every
// node first reached in the previous hop draws up to `fanout` of its in-neighbours uniformly
// WITHOUT replacement (Floyd's algorithm on a Philox stream), the subgraph keeps every sampled
// edge (neighbour -> node), local numbering puts the seeds first, then new nodes hop by hop in
// order of first appearance.  The full graph's by-target CSR (int32) stays resident in HBM; a
// batch costs 13 small launches (two hops); its sizes (12 bytes) are the only thing the host reads.
// Because the sampler walks targets in local order, it emits the batch's by-target CSR directly
// (no sort); edge j of the batch is slot j of that CSR.
//
// Round 3: 21 launches (of which 8 inside four rocPRIM scans over the hop's ENTRIES) -> 13 for two hops, by-source
// view included.  Per hop:
//   sample + claim (one thread per frontier node; fused with the PREVIOUS hop's edge emission, whose inputs it does
//   not touch) -> wins per frontier node -> ONE single-block scan over the frontier NODES (new-node offsets and edge
//   offsets at once: a tenth of the entries) -> assign the new local ids -> emit the hop's edges.
// The numbering is the same as before (winners in entry order).  stemgnn_sample_batch_views additionally emits the
// batch's BY-SOURCE CSR (what the augmentation and the aggregation's backward walk) without a sort: out-degrees are
// counted while the edges are emitted, one single-block scan gives the row offsets, a scatter places the edges and a
// per-node insertion sort puts the (mostly one-element) segments into edge order -- the stable order a sort by source
// returns (csrc/graph_build.hip), bit for bit.
#include "common.h"

#include <climits>
#include <cstring>

namespace stemgnn {
namespace {

constexpr int kThreads = 256;
constexpr int kMaxFanout = 32;
constexpr int32_t kUnassigned = INT_MIN;

struct BatchCounters {  // device-side running counts: nodes[h] = nodes known before hop h, edges likewise
  int32_t nodes[16];
  int32_t edges[16];
};

__global__ void __launch_bounds__(kThreads)
k_seed_init(const int64_t* __restrict__ seeds, int32_t B, int64_t N, int32_t* __restrict__ local_of,
            int32_t* __restrict__ n_id, BatchCounters* __restrict__ ctr, int32_t* __restrict__ deg_out,
            int32_t* __restrict__ cursor) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i == 0) { ctr->nodes[0] = 0; ctr->nodes[1] = B; ctr->edges[0] = 0; }
  if (i >= B) return;
  if (deg_out) { deg_out[i] = 0; cursor[i] = 0; }
  const int64_t g = seeds[i];
  if (g < 0 || g >= N) { n_id[i] = 0; return; }  // an id outside the graph stands for node 0 (callers validate seeds)
  n_id[i] = static_cast<int32_t>(g);
  // duplicate seeds keep the lowest position (atomicMax over non-negative ids would keep the highest)
  atomicMax(&local_of[g], INT_MAX - i);  // provisional: decoded by k_count_wins of hop 0
}

// Sample min(deg, fanout) distinct in-neighbour slots of global node v (Floyd's algorithm: for j = deg - f .. deg - 1
// draw t ~ U[0, j] and take t unless already taken, else j), written in CSR order to s_src / s_type [out0 ..).  Rows of
// up to 64 slots keep the chosen set as a 64-bit mask in a register: membership is a shift, the ascending walk a
// find-first-set; longer rows keep a sorted array.  Returns the count.
__device__ inline int sample_row(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src,
                                 const int32_t* __restrict__ etype, int32_t v, int fanout, uint64_t seed, uint64_t offset,
                                 int i, int32_t* __restrict__ s_src, int32_t* __restrict__ s_type, int64_t out0) {
  const int beg = rowptr[v], deg = rowptr[v + 1] - beg;
  if (fanout < 0 || deg <= fanout) {  // the whole row (fan-out -1: every in-neighbour; the caller sized the entries)
    const int c = deg;
    for (int j = 0; j < c; ++j) {
      s_src[out0 + j] = src[beg + j];
      s_type[out0 + j] = etype ? etype[beg + j] : 0;
    }
    return c;
  }
  uint32_t r[4];
  if (deg <= 64) {
    uint64_t mask = 0;
    for (int q = 0; q < fanout; ++q) {
      if ((q & 3) == 0) Philox::gen(seed, offset, static_cast<uint64_t>(i) * 8 + (q >> 2), r);
      const int j = deg - fanout + q;
      int t = static_cast<int>(r[q & 3] % static_cast<uint32_t>(j + 1));
      if ((mask >> t) & 1ull) t = j;  // j itself was never drawn before: every earlier draw is < j
      mask |= 1ull << t;
    }
    if (fanout <= 16) {
      // all loads of the row in flight before the first store (they are scattered over an 80 MB array)
      int pos[16], vs[16], vt[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        pos[j] = mask ? __ffsll(static_cast<unsigned long long>(mask)) - 1 : 0;
        mask &= mask - 1;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        vs[j] = j < fanout ? src[beg + pos[j]] : 0;
        vt[j] = (j < fanout && etype) ? etype[beg + pos[j]] : 0;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (j < fanout) { s_src[out0 + j] = vs[j]; s_type[out0 + j] = vt[j]; }
      return fanout;
    }
    for (int j = 0; j < fanout; ++j) {
      const int t = __ffsll(static_cast<unsigned long long>(mask)) - 1;
      mask &= mask - 1;
      s_src[out0 + j] = src[beg + t];
      s_type[out0 + j] = etype ? etype[beg + t] : 0;
    }
    return fanout;
  }
  int chosen[kMaxFanout];
  int c = 0;
  for (int q = 0; q < fanout; ++q) {
    if ((q & 3) == 0) Philox::gen(seed, offset, static_cast<uint64_t>(i) * 8 + (q >> 2), r);
    const int j = deg - fanout + q;
    int t = static_cast<int>(r[q & 3] % static_cast<uint32_t>(j + 1));
    bool taken = false;
    for (int a = 0; a < c; ++a) taken |= (chosen[a] == t);
    if (taken) t = j;
    int p = c++;
    while (p > 0 && chosen[p - 1] > t) { chosen[p] = chosen[p - 1]; --p; }  // keeps `chosen` ascending
    chosen[p] = t;
  }
  for (int j = 0; j < c; ++j) {
    s_src[out0 + j] = src[beg + chosen[j]];
    s_type[out0 + j] = etype ? etype[beg + chosen[j]] : 0;
  }
  return c;
}

// One hop's scratch: the sampled entries of its frontier nodes (entry p = entry0(i) + j) and the per-node offsets
__device__ __forceinline__ int64_t entry0(const int32_t* __restrict__ ent_base, int i, int fanout) {
  return ent_base ? static_cast<int64_t>(ent_base[i]) : static_cast<int64_t>(i) * fanout;
}
struct HopBuf {
  int32_t *s_src, *s_type, *s_cnt;  // [cap * fanout] global ids / edge types, [cap] counts
  const int32_t* ent_base;          // first entry of frontier node i (sized-per-hop path); NULL: i * fanout
  int32_t *wins, *new_base, *edge_base;  // [cap]: new nodes a frontier node wins, and the exclusive scans
  int32_t cap;                           // frontier capacity of the hop
  int32_t fanout, hop;
};
struct EmitOut {
  int32_t *b_rowptr, *b_src, *b_type, *b_dst;
  int64_t* b_type64;
  int32_t* deg_out;  // out-degree per local node, counted while the edges are emitted (NULL: no by-source view)
  int32_t cap_edges;
};

// Edges of hop `e` into the batch CSR (frontier node i, local id f0 + i, owns slots [edges[hop] + edge_base[i], + cnt)):
// every source of the hop has its local id by now.
__device__ inline void emit_role(const HopBuf& hb, const EmitOut& o, const BatchCounters* __restrict__ ctr,
                                 const int32_t* __restrict__ local_of, int i) {
  if (i >= hb.cap) return;
  const int f0 = ctr->nodes[hb.hop], nf = ctr->nodes[hb.hop + 1] - f0;
  if (i >= nf) return;
  const int c = hb.s_cnt[i];
  const int base = ctr->edges[hb.hop] + hb.edge_base[i];
  o.b_rowptr[f0 + i] = base;
  const int64_t p0 = entry0(hb.ent_base, i, hb.fanout);
  for (int j0 = 0; j0 < c; j0 += 4) {  // four entries at a time: their two dependent reads each overlap
    int32_t g[4], ty[4], ls[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = min(j0 + k, c - 1);
      g[k] = hb.s_src[p0 + j];
      ty[k] = hb.s_type[p0 + j];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) ls[k] = local_of[g[k]];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int slot = base + j0 + k;
      if (j0 + k >= c || slot >= o.cap_edges) break;
      o.b_src[slot] = ls[k];
      o.b_type[slot] = ty[k];
      if (o.b_type64) o.b_type64[slot] = ty[k];
      o.b_dst[slot] = f0 + i;
      if (o.deg_out) atomicAdd(&o.deg_out[ls[k]], 1);
    }
  }
}

// Sample the frontier of hop `hb.hop` and claim the unknown neighbours: the lowest entry position wins (deterministic
// numbering).  local_of < 0 = unknown (kUnassigned) or claimed in this hop; >= 0 = a local id.
__device__ inline void sample_role(const HopBuf& hb, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src,
                                   const int32_t* __restrict__ etype, const int32_t* __restrict__ n_id,
                                   const BatchCounters* __restrict__ ctr, uint64_t seed, uint64_t offset,
                                   int32_t* __restrict__ local_of, int i) {
  if (i >= hb.cap) return;
  const int f0 = ctr->nodes[hb.hop], f1 = ctr->nodes[hb.hop + 1];
  if (i >= f1 - f0) { hb.s_cnt[i] = 0; return; }
  const int64_t out0 = entry0(hb.ent_base, i, hb.fanout);
  const int c = sample_row(rowptr, src, etype, n_id[f0 + i], hb.fanout, seed, offset, i, hb.s_src, hb.s_type, out0);
  hb.s_cnt[i] = c;
  for (int j = 0; j < c; ++j) {
    const int32_t g = hb.s_src[out0 + j];
    if (local_of[g] < 0) atomicMax(&local_of[g], -(static_cast<int32_t>(out0) + j + 2));
  }
}

// blocks [0, emit_blocks): emit hop `em` (when emit_blocks > 0); the rest: sample hop `sa` (when it has blocks).  The two
// roles touch disjoint data: the emitter READS local ids of nodes assigned in earlier launches, the sampler claims
// nodes that have none yet.
__global__ void __launch_bounds__(kThreads)
k_emit_and_sample(HopBuf em, EmitOut out, int emit_blocks, HopBuf sa, const int32_t* __restrict__ rowptr,
                  const int32_t* __restrict__ src, const int32_t* __restrict__ etype, const int32_t* __restrict__ n_id,
                  const BatchCounters* __restrict__ ctr, uint64_t seed, uint64_t offset, int32_t* __restrict__ local_of) {
  const int b = blockIdx.x;
  if (b < emit_blocks) emit_role(em, out, ctr, local_of, b * kThreads + threadIdx.x);
  else sample_role(sa, rowptr, src, etype, n_id, ctr, seed, offset, local_of, (b - emit_blocks) * kThreads + threadIdx.x);
}

// wins[i] = the unknown neighbours frontier node i claimed first; hop 0 also decodes the seeds' provisional ids
__global__ void __launch_bounds__(kThreads)
k_count_wins(HopBuf hb, const BatchCounters* __restrict__ ctr, int32_t B, const int32_t* __restrict__ n_id,
             int32_t* __restrict__ local_of) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (hb.hop == 0 && i < B) {
    const int32_t g = n_id[i];
    const int32_t v = local_of[g];
    if (v > INT_MAX - B - 1) local_of[g] = INT_MAX - v;  // first position that named this node
  }
  if (i >= hb.cap) return;
  const int nf = ctr->nodes[hb.hop + 1] - ctr->nodes[hb.hop];
  int w = 0;
  if (i < nf) {
    const int c = hb.s_cnt[i];
    const int32_t p0 = static_cast<int32_t>(entry0(hb.ent_base, i, hb.fanout));
    for (int j = 0; j < c; ++j) w += local_of[hb.s_src[p0 + j]] == -(p0 + j + 2) ? 1 : 0;
  }
  hb.wins[i] = w;
}

// ONE block: exclusive scans of a[0, n) -> a_out (and of b -> b_out when given), n = *n_hi - *n_lo read on the device.
// Rounds of kScanGroups * 4096 elements: a thread takes four consecutive elements (one 16-byte load: a wave reads 1 KB
// in a piece) of each of the round's groups, all loads in flight together; the groups are scanned side by side (wave
// scans by shuffles, one exchange of the 16 x kScanGroups wave totals through LDS, two barriers per round).
// (A thread taking 16 consecutive elements -- 64-byte stride between lanes -- made every load instruction touch 64
// cache lines: 85 us for the 102k out-degrees of a C4 batch.)  tot_a[0] = base_a[0] + sum(a) (likewise b); closing:
// a_out[n] = sum(a).  b_out may be NULL (only b's total is wanted).
constexpr int kScanThreads = 1024, kScanGroups = 4, kScanGroupElems = kScanThreads * 4;
// `al`: the array starts on a 16-byte boundary (the library's own buffers do; a caller's output array need not)
__device__ __forceinline__ int4 scan_load4(const int32_t* __restrict__ p, int i, int n, bool al) {
  if (al && i + 3 < n) return *reinterpret_cast<const int4*>(p + i);
  return make_int4(i < n ? p[i] : 0, i + 1 < n ? p[i + 1] : 0, i + 2 < n ? p[i + 2] : 0, i + 3 < n ? p[i + 3] : 0);
}
__device__ __forceinline__ void scan_store4(int32_t* __restrict__ p, int i, int n, int32_t e, int4 v, bool al) {
  const int4 o = make_int4(e, e + v.x, e + v.x + v.y, e + v.x + v.y + v.z);
  if (al && i + 3 < n) { *reinterpret_cast<int4*>(p + i) = o; return; }
  if (i < n) p[i] = o.x;
  if (i + 1 < n) p[i + 1] = o.y;
  if (i + 2 < n) p[i + 2] = o.z;
  if (i + 3 < n) p[i + 3] = o.w;
}
__global__ void __launch_bounds__(kScanThreads)
k_scan_block(const int32_t* __restrict__ a, int32_t* __restrict__ a_out, const int32_t* __restrict__ b,
             int32_t* __restrict__ b_out, const int32_t* __restrict__ n_hi, const int32_t* __restrict__ n_lo,
             int32_t n_cap, int32_t* __restrict__ tot_a, const int32_t* __restrict__ base_a, int32_t* __restrict__ tot_b,
             const int32_t* __restrict__ base_b, bool closing) {
  __shared__ int32_t wa[kScanGroups][kScanThreads / 64], wb[kScanGroups][kScanThreads / 64];
  int n = n_hi[0] - n_lo[0];
  if (n > n_cap) n = n_cap;
  if (n < 0) n = 0;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool al_a = al16(a), al_b = al16(b), al_ao = al16(a_out), al_bo = al16(b_out);
  int32_t ca = 0, cb = 0;  // running totals of the rounds before this one (the same in every thread)
  for (int base = 0; base < n; base += kScanGroups * kScanGroupElems) {
    int4 va[kScanGroups], vb[kScanGroups];
#pragma unroll
    for (int g = 0; g < kScanGroups; ++g) {
      const int i = base + g * kScanGroupElems + 4 * t;
      va[g] = scan_load4(a, i, n, al_a);
      vb[g] = b ? scan_load4(b, i, n, al_b) : make_int4(0, 0, 0, 0);
    }
    int32_t sa[kScanGroups], sb[kScanGroups], ia[kScanGroups], ib[kScanGroups];
#pragma unroll
    for (int g = 0; g < kScanGroups; ++g) {
      ia[g] = sa[g] = va[g].x + va[g].y + va[g].z + va[g].w;
      ib[g] = sb[g] = vb[g].x + vb[g].y + vb[g].z + vb[g].w;
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1)
#pragma unroll
      for (int g = 0; g < kScanGroups; ++g) {  // inclusive over the wave, the groups' chains interleaved
        const int32_t x = __shfl_up(ia[g], o), y = __shfl_up(ib[g], o);
        if (lane >= o) { ia[g] += x; ib[g] += y; }
      }
    if (lane == 63)
#pragma unroll
      for (int g = 0; g < kScanGroups; ++g) { wa[g][w] = ia[g]; wb[g][w] = ib[g]; }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < kScanGroups; ++g) {
      int32_t oa = 0, ob = 0, ta = 0, tb = 0;
#pragma unroll
      for (int k = 0; k < kScanThreads / 64; ++k) {
        const int32_t x = wa[g][k], y = wb[g][k];
        if (k < w) { oa += x; ob += y; }
        ta += x; tb += y;
      }
      const int i = base + g * kScanGroupElems + 4 * t;
      scan_store4(a_out, i, n, ca + oa + ia[g] - sa[g], va[g], al_ao);
      if (b && b_out) scan_store4(b_out, i, n, cb + ob + ib[g] - sb[g], vb[g], al_bo);
      ca += ta; cb += tb;  // the next group starts behind this one
    }
    __syncthreads();  // wa / wb are rewritten by the next round
  }
  if (t == 0) {
    if (tot_a) tot_a[0] = (base_a ? base_a[0] : 0) + ca;
    if (tot_b) tot_b[0] = (base_b ? base_b[0] : 0) + cb;
    if (closing) a_out[n] = ca;
  }
}

// New local ids: the winners of frontier node i, in slot order, from nodes[hop + 1] + new_base[i]
__global__ void __launch_bounds__(kThreads)
k_assign_new(HopBuf hb, int32_t cap_nodes, const BatchCounters* __restrict__ ctr, int32_t* __restrict__ local_of,
             int32_t* __restrict__ n_id, int32_t* __restrict__ deg_out, int32_t* __restrict__ cursor,
             int32_t* __restrict__ nodes_after /* may be NULL or pinned host memory */) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i == 0 && nodes_after) nodes_after[0] = ctr->nodes[hb.hop + 2];
  if (i >= hb.cap) return;
  const int nf = ctr->nodes[hb.hop + 1] - ctr->nodes[hb.hop];
  if (i >= nf) return;
  int id = ctr->nodes[hb.hop + 1] + hb.new_base[i];
  const int c = hb.s_cnt[i];
  const int32_t p0 = static_cast<int32_t>(entry0(hb.ent_base, i, hb.fanout));
  for (int j = 0; j < c; ++j) {
    const int32_t g = hb.s_src[p0 + j];
    if (local_of[g] == -(p0 + j + 2)) {
      if (id < cap_nodes) {
        n_id[id] = g;
        local_of[g] = id;
        if (deg_out) { deg_out[id] = 0; cursor[id] = 0; }
      }
      ++id;
    }
  }
}

struct ViewOut {  // the by-source view (all NULL when not wanted) and the per-node outputs of the finish
  int32_t *rowptr_t, *dst_t, *eid_t, *type_t, *cursor;
  bool compact_coo;  // COO rows E_b apart (a contiguous [2, E_b]) instead of cap_edges apart
  float* inv_deg;
  int64_t *n_id64, *x_out;
  const int64_t* x;  // node -> feature row table (x_out[i] = x[n_id[i]]), may be NULL
};

// Last launch over max(cap_edges, cap_nodes + 1) threads.  Node part: rowptr of every node that was never a frontier
// node (no in-edges) + the closing entry, 1 / in-degree, the int64 copies, the scratch map cleared, (N_b, E_b, A_b)
// published.  Edge part: the int64 COO (its size is only known now, so a contiguous [2, E_b] can be written) and, when
// the by-source view is wanted, edge e goes to its source's row at an atomically drawn position -- k_sort_segments
// then puts every row into edge order.
__global__ void __launch_bounds__(kThreads)
k_finish(int hops, const BatchCounters* __restrict__ ctr, int32_t cap_nodes, int32_t cap_edges,
         const int32_t* __restrict__ n_id, int32_t* __restrict__ local_of, int32_t* __restrict__ b_rowptr,
         const int32_t* __restrict__ b_src, const int32_t* __restrict__ b_type, const int32_t* __restrict__ b_dst,
         int64_t* __restrict__ b_coo, ViewOut v, int32_t* __restrict__ counts /*[3]*/) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  int nb = ctr->nodes[hops + 1];
  if (nb > cap_nodes) nb = cap_nodes;
  int eb = ctr->edges[hops];
  if (eb > cap_edges) eb = cap_edges;
  const int expanded = min(ctr->nodes[hops], nb);  // nodes [0, expanded) were frontier nodes of some hop
  if (i == 0) { counts[0] = nb; counts[1] = eb; counts[2] = expanded; }
  if (i <= cap_nodes) {
    if (i >= expanded && i <= nb) b_rowptr[i] = eb;
    if (i < nb) {
      const int32_t g = n_id[i];
      local_of[g] = kUnassigned;
      if (v.n_id64) v.n_id64[i] = g;
      if (v.x_out) v.x_out[i] = v.x ? v.x[g] : g;
      if (v.inv_deg) {
        int d = 0;
        if (i < expanded) d = ((i + 1 < expanded) ? b_rowptr[i + 1] : eb) - b_rowptr[i];
        v.inv_deg[i] = 1.0f / static_cast<float>(d < 1 ? 1 : d);
      }
    }
  }
  if (i < eb) {
    const int32_t s = b_src[i], d = b_dst[i];
    b_coo[i] = s;
    b_coo[(v.compact_coo ? eb : cap_edges) + i] = d;
    if (!v.rowptr_t) return;
    const int pos = v.rowptr_t[s] + atomicAdd(&v.cursor[s], 1);
    v.dst_t[pos] = d;
    v.eid_t[pos] = static_cast<int32_t>(i);
    v.type_t[pos] = b_type[i];
  }
}

// every by-source row into edge order (insertion sort by edge id; rows of a sampled batch are mostly one element)
__global__ void __launch_bounds__(kThreads)
k_sort_segments(const int32_t* __restrict__ n_nodes, int32_t cap_nodes, const int32_t* __restrict__ rowptr_t,
                int32_t* __restrict__ dst_t, int32_t* __restrict__ eid_t, int32_t* __restrict__ type_t) {
  const int v = blockIdx.x * kThreads + threadIdx.x;
  if (v >= n_nodes[0] || v >= cap_nodes) return;
  const int beg = rowptr_t[v], end = rowptr_t[v + 1];
  for (int a = beg + 1; a < end; ++a) {
    const int32_t e = eid_t[a], d = dst_t[a], t = type_t[a];
    int b = a - 1;
    while (b >= beg && eid_t[b] > e) {
      eid_t[b + 1] = eid_t[b]; dst_t[b + 1] = dst_t[b]; type_t[b + 1] = type_t[b];
      --b;
    }
    eid_t[b + 1] = e; dst_t[b + 1] = d; type_t[b + 1] = t;
  }
}

__global__ void __launch_bounds__(kThreads) k_fill_i32(int32_t* p, int64_t n, int32_t v) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i < n) p[i] = v;
}

inline unsigned grid_for(int64_t n) { return static_cast<unsigned>((n < 1 ? 1 : n + kThreads - 1) / kThreads); }

struct Caps {
  int64_t fmax, need_nodes, need_edges, cap_frontier_max, cap_entries_max;
  bool ok;
};
inline Caps caps_of(int64_t batch_size, const int32_t* fanouts_host, int64_t hops) {
  Caps c{0, batch_size, 0, batch_size, 0, true};
  int64_t level = batch_size;
  for (int64_t h = 0; h < hops; ++h) {
    const int64_t f = fanouts_host[h];
    if (f <= 0 || f > kMaxFanout) { c.ok = false; return c; }
    if (f > c.fmax) c.fmax = f;
    level *= f;
    c.need_nodes += level;
    c.need_edges += level;
  }
  for (int64_t h = 1; h < hops; ++h) c.cap_frontier_max *= c.fmax;
  c.cap_entries_max = c.cap_frontier_max * c.fmax;
  return c;
}

int sample_impl(const int32_t* rowptr, const int32_t* src, const int32_t* etype, int64_t num_nodes, const int64_t* seeds,
                int64_t batch_size, const int32_t* fanouts_host, int64_t hops, uint64_t seed, uint64_t offset,
                int32_t* local_of, int64_t cap_nodes, int64_t cap_edges, int32_t* n_id, int32_t* b_rowptr, int32_t* b_src,
                int32_t* b_type, int64_t* b_type64, int64_t* b_coo, int32_t* counts, ViewOut view, bool want_view,
                void* workspace, size_t workspace_bytes, hipStream_t st) {
  if (batch_size <= 0 || hops <= 0 || hops > 14 || !fanouts_host || !rowptr || !src || !seeds || !local_of || !n_id ||
      !b_rowptr || !b_src || !b_type || !b_coo || !counts || !workspace)
    return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(num_nodes) || !fits_i32(cap_nodes) || !fits_i32(cap_edges)) return STEMGNN_ERR_TOO_LARGE;
  const Caps cp = caps_of(batch_size, fanouts_host, hops);
  if (!cp.ok) return STEMGNN_ERR_INVALID_ARG;
  if (cap_nodes < cp.need_nodes || cap_edges < cp.need_edges) return STEMGNN_ERR_WORKSPACE;
  if (!fits_i32(cp.cap_entries_max + 2)) return STEMGNN_ERR_TOO_LARGE;  // entry positions are 32-bit claim values
  if (workspace_bytes < stemgnn_sampler_workspace_bytes(batch_size, hops, cp.fmax)) return STEMGNN_ERR_WORKSPACE;

  uintptr_t p = align_up(reinterpret_cast<uintptr_t>(workspace), 256);
  auto carve = [&](size_t bytes) { uintptr_t q = p; p += align_up(bytes, 256); return reinterpret_cast<int32_t*>(q); };
  HopBuf buf[2];  // ping-pong: hop h + 1 is sampled while hop h is emitted
  for (int k = 0; k < 2; ++k) {
    buf[k].s_src = carve(cp.cap_entries_max * 4);
    buf[k].s_type = carve(cp.cap_entries_max * 4);
    buf[k].s_cnt = carve(cp.cap_frontier_max * 4);
    buf[k].wins = carve(cp.cap_frontier_max * 4);
    buf[k].new_base = carve(cp.cap_frontier_max * 4);
    buf[k].edge_base = carve(cp.cap_frontier_max * 4);
    buf[k].ent_base = nullptr;
  }
  // local ids never reach need_nodes (every frontier node adds at most its fan-out), whatever cap_nodes the caller holds
  int32_t* deg_out = carve(static_cast<size_t>(cp.need_nodes) * 4);
  int32_t* cursor = carve(static_cast<size_t>(cp.need_nodes) * 4);
  int32_t* b_dst = carve(static_cast<size_t>(cp.need_edges) * 4);
  BatchCounters* ctr = reinterpret_cast<BatchCounters*>(carve(sizeof(BatchCounters)));
  if (p - reinterpret_cast<uintptr_t>(workspace) > workspace_bytes) return STEMGNN_ERR_WORKSPACE;
  const int32_t B = static_cast<int32_t>(batch_size);
  k_seed_init<<<grid_for(B), kThreads, 0, st>>>(seeds, B, num_nodes, local_of, n_id, ctr, want_view ? deg_out : nullptr,
                                               want_view ? cursor : nullptr);
  STEMGNN_LAUNCH_CHECK();

  EmitOut eo{b_rowptr, b_src, b_type, b_dst, b_type64, want_view ? deg_out : nullptr, static_cast<int32_t>(cap_edges)};
  int64_t cap_frontier = batch_size;
  HopBuf none{};
  for (int hop = 0; hop <= hops; ++hop) {
    // launch `hop`: emit hop - 1 (if any) and sample hop (if any)
    HopBuf em = none, sa = none;
    int emit_blocks = 0, sample_blocks = 0;
    if (hop > 0) {
      em = buf[(hop - 1) & 1];
      emit_blocks = static_cast<int>(grid_for(em.cap));
    }
    if (hop < hops) {
      HopBuf& h = buf[hop & 1];
      h.cap = static_cast<int32_t>(cap_frontier);
      h.fanout = fanouts_host[hop];
      h.hop = hop;
      sa = h;
      sample_blocks = static_cast<int>(grid_for(h.cap));
    }
    k_emit_and_sample<<<static_cast<unsigned>(emit_blocks + sample_blocks), kThreads, 0, st>>>(
        em, eo, emit_blocks, sa, rowptr, src, etype, n_id, ctr, seed, offset + static_cast<uint64_t>(hop) * 1000003ull,
        local_of);
    STEMGNN_LAUNCH_CHECK();
    if (hop == hops) break;
    const HopBuf& h = buf[hop & 1];
    k_count_wins<<<grid_for(std::max<int64_t>(h.cap, hop == 0 ? B : 0)), kThreads, 0, st>>>(h, ctr, B, n_id, local_of);
    STEMGNN_LAUNCH_CHECK();
    k_scan_block<<<1, kScanThreads, 0, st>>>(h.wins, h.new_base, h.s_cnt, h.edge_base, ctr->nodes + hop + 1,
                                             ctr->nodes + hop, h.cap, ctr->nodes + hop + 2, ctr->nodes + hop + 1,
                                             ctr->edges + hop + 1, ctr->edges + hop, false);
    STEMGNN_LAUNCH_CHECK();
    k_assign_new<<<grid_for(h.cap), kThreads, 0, st>>>(h, static_cast<int32_t>(cap_nodes), ctr, local_of, n_id,
                                                       want_view ? deg_out : nullptr, want_view ? cursor : nullptr, nullptr);
    STEMGNN_LAUNCH_CHECK();
    cap_frontier *= fanouts_host[hop];
  }
  if (want_view) {
    // row offsets of the by-source view: one block over the batch's nodes (n = nodes[hops + 1] - nodes[0])
    k_scan_block<<<1, kScanThreads, 0, st>>>(deg_out, view.rowptr_t, nullptr, nullptr, ctr->nodes + hops + 1, ctr->nodes,
                                             static_cast<int32_t>(cp.need_nodes), nullptr, nullptr, nullptr, nullptr, true);
    STEMGNN_LAUNCH_CHECK();
  }
  view.cursor = cursor;
  if (!want_view) view.rowptr_t = view.dst_t = view.eid_t = view.type_t = nullptr;
  k_finish<<<grid_for(std::max<int64_t>(cap_edges, cap_nodes + 1)), kThreads, 0, st>>>(
      static_cast<int>(hops), ctr, static_cast<int32_t>(cap_nodes), static_cast<int32_t>(cap_edges), n_id, local_of,
      b_rowptr, b_src, b_type, b_dst, b_coo, view, counts);
  STEMGNN_LAUNCH_CHECK();
  if (want_view) {
    // the node count comes from the device-side counters: `counts` may be pinned host memory (written, never read)
    k_sort_segments<<<grid_for(cp.need_nodes), kThreads, 0, st>>>(ctr->nodes + hops + 1, static_cast<int32_t>(cap_nodes),
                                                                  view.rowptr_t, view.dst_t, view.eid_t, view.type_t);
    STEMGNN_LAUNCH_CHECK();
  }
  return STEMGNN_OK;
}

// Entries per frontier node of a hop: min(deg, fanout), or deg for fanout < 0 (every in-neighbour)
__global__ void __launch_bounds__(kThreads)
k_hop_counts(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ n_id, const BatchCounters* __restrict__ ctr,
             int hop, int fanout, int32_t cap, int32_t* __restrict__ s_cnt) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= cap) return;
  const int f0 = ctr->nodes[hop], nf = ctr->nodes[hop + 1] - f0;
  int c = 0;
  if (i < nf) {
    const int32_t v = n_id[f0 + i];
    const int deg = rowptr[v + 1] - rowptr[v];
    c = (fanout < 0 || deg <= fanout) ? deg : fanout;
  }
  s_cnt[i] = c;
}

}  // namespace
}  // namespace stemgnn

using namespace stemgnn;

extern "C" {

int stemgnn_sampler_init_map(int32_t* local_of, int64_t num_nodes, void* stream_) {
  if (num_nodes < 0 || (num_nodes > 0 && !local_of)) return STEMGNN_ERR_INVALID_ARG;
  if (num_nodes == 0) return STEMGNN_OK;
  k_fill_i32<<<grid_for(num_nodes), kThreads, 0, static_cast<hipStream_t>(stream_)>>>(local_of, num_nodes, kUnassigned);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

size_t stemgnn_sampler_workspace_bytes(int64_t batch_size, int64_t hops, int64_t fanout) {
  if (batch_size <= 0 || hops <= 0 || hops > 14 || fanout <= 0 || fanout > kMaxFanout) return 0;
  int64_t cap_frontier = batch_size, cap_nodes = batch_size, level = batch_size;
  for (int64_t h = 1; h < hops; ++h) cap_frontier *= fanout;  // last hop's frontier bound
  for (int64_t h = 0; h < hops; ++h) { level *= fanout; cap_nodes += level; }
  const int64_t cap_entries = cap_frontier * fanout;
  size_t b = 0;
  b += 2 * (2 * align_up(static_cast<size_t>(cap_entries) * 4, 256) + 4 * align_up(static_cast<size_t>(cap_frontier) * 4, 256));
  b += 2 * align_up(static_cast<size_t>(cap_nodes) * 4, 256);  // out-degrees, scatter cursors
  b += align_up(static_cast<size_t>(cap_nodes - batch_size) * 4, 256);  // edge targets
  b += align_up(sizeof(BatchCounters), 256);                   // counters
  return b + 1024;
}

int stemgnn_sample_batch(const int32_t* rowptr, const int32_t* src, const int32_t* etype, int64_t num_nodes,
                         const int64_t* seeds, int64_t batch_size, const int32_t* fanouts_host, int64_t hops,
                         uint64_t seed, uint64_t offset, int32_t* local_of, int64_t cap_nodes, int64_t cap_edges,
                         int32_t* n_id, int32_t* b_rowptr, int32_t* b_src, int32_t* b_type, int64_t* b_coo,
                         int32_t* counts, void* workspace, size_t workspace_bytes, void* stream_) {
  return sample_impl(rowptr, src, etype, num_nodes, seeds, batch_size, fanouts_host, hops, seed, offset, local_of,
                     cap_nodes, cap_edges, n_id, b_rowptr, b_src, b_type, nullptr, b_coo, counts, ViewOut{}, false,
                     workspace, workspace_bytes, static_cast<hipStream_t>(stream_));
}

int stemgnn_sample_batch_views(const int32_t* rowptr, const int32_t* src, const int32_t* etype, int64_t num_nodes,
                               const int64_t* seeds, int64_t batch_size, const int32_t* fanouts_host, int64_t hops,
                               uint64_t seed, uint64_t offset, int32_t* local_of, int64_t cap_nodes, int64_t cap_edges,
                               int32_t* n_id, int32_t* b_rowptr, int32_t* b_src, int32_t* b_type, int64_t* b_coo,
                               int32_t* counts, int32_t* rowptr_t, int32_t* dst_t, int32_t* eid_t, int32_t* type_t,
                               float* inv_deg, int64_t* n_id64, int64_t* type64, const int64_t* x, int64_t* x_out,
                               void* workspace, size_t workspace_bytes, void* stream_) {
  const bool want_view = rowptr_t || dst_t || eid_t || type_t;  // all four or none
  if (!inv_deg || (want_view && (!rowptr_t || !dst_t || !eid_t || !type_t))) return STEMGNN_ERR_INVALID_ARG;
  ViewOut v{rowptr_t, dst_t, eid_t, type_t, nullptr, true, inv_deg, n_id64, x_out, x};
  return sample_impl(rowptr, src, etype, num_nodes, seeds, batch_size, fanouts_host, hops, seed, offset, local_of,
                     cap_nodes, cap_edges, n_id, b_rowptr, b_src, b_type, type64, b_coo, counts, v, want_view, workspace,
                     workspace_bytes, static_cast<hipStream_t>(stream_));
}

// ---- fan-out -1 (every in-neighbour; reference utils/loader.py:18-25, finetune.py:237: the evaluation loaders) -------
// The entries of a hop are as many as its frontier's in-degrees add up to: no bound is known before the hop runs.  The
// batch is therefore built in steps, with one 4-byte size read by the host per hop (entries) -- an evaluation path, not
// the training loop's: begin -> per hop [sizes -> (host allocates the hop's entries) -> expand] -> (host allocates the
// outputs at their exact size) -> finish.  Same kernels, same numbering as stemgnn_sample_batch; the entries of frontier
// node i start at the exclusive scan of the counts instead of at i * fanout.  `state`: 32 int32 on the device.

int stemgnn_sampler_full_begin(const int64_t* seeds, int64_t batch_size, int64_t num_nodes, int32_t* local_of,
                               int32_t* n_id, int32_t* state, void* stream_) {
  if (batch_size <= 0 || !seeds || !local_of || !n_id || !state) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(num_nodes) || !fits_i32(batch_size)) return STEMGNN_ERR_TOO_LARGE;
  const int32_t B = static_cast<int32_t>(batch_size);
  k_seed_init<<<grid_for(B), kThreads, 0, static_cast<hipStream_t>(stream_)>>>(
      seeds, B, num_nodes, local_of, n_id, reinterpret_cast<BatchCounters*>(state), nullptr, nullptr);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_sampler_full_hop_sizes(const int32_t* rowptr, const int32_t* n_id, int32_t* state, int32_t hop,
                                   int32_t fanout, int64_t frontier_cap, int32_t* cnt, int32_t* ent_base,
                                   int32_t* total, void* stream_) {
  if (!rowptr || !n_id || !state || !cnt || !ent_base || !total || hop < 0 || hop > 13 || frontier_cap <= 0 ||
      fanout == 0 || fanout > kMaxFanout)
    return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(frontier_cap)) return STEMGNN_ERR_TOO_LARGE;
  hipStream_t st = static_cast<hipStream_t>(stream_);
  BatchCounters* ctr = reinterpret_cast<BatchCounters*>(state);
  const int32_t cap = static_cast<int32_t>(frontier_cap);
  k_hop_counts<<<grid_for(cap), kThreads, 0, st>>>(rowptr, n_id, ctr, hop, fanout, cap, cnt);
  STEMGNN_LAUNCH_CHECK();
  k_scan_block<<<1, kScanThreads, 0, st>>>(cnt, ent_base, nullptr, nullptr, ctr->nodes + hop + 1, ctr->nodes + hop, cap,
                                           total, nullptr, nullptr, nullptr, false);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_sampler_full_hop_expand(const int32_t* rowptr, const int32_t* src, const int32_t* etype, int32_t* n_id,
                                    int64_t n_cap, int32_t* state, int32_t hop, int32_t fanout,
                                    uint64_t seed, uint64_t offset, int64_t batch_size, int64_t frontier_cap,
                                    int64_t entries, int32_t* cnt, const int32_t* ent_base, int32_t* s_src,
                                    int32_t* s_type, int32_t* wins, int32_t* new_base, int32_t* local_of,
                                    int32_t* nodes_after, void* stream_) {
  if (!rowptr || !src || !n_id || !state || !cnt || !ent_base || !wins || !new_base || !local_of || hop < 0 || hop > 13 ||
      frontier_cap <= 0 || entries < 0 || fanout == 0 || fanout > kMaxFanout || (entries > 0 && (!s_src || !s_type)))
    return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(frontier_cap) || !fits_i32(entries + 2) || !fits_i32(n_cap)) return STEMGNN_ERR_TOO_LARGE;
  hipStream_t st = static_cast<hipStream_t>(stream_);
  BatchCounters* ctr = reinterpret_cast<BatchCounters*>(state);
  HopBuf h{};
  h.s_src = s_src; h.s_type = s_type; h.s_cnt = cnt; h.ent_base = ent_base;
  h.wins = wins; h.new_base = new_base; h.edge_base = const_cast<int32_t*>(ent_base);
  h.cap = static_cast<int32_t>(frontier_cap); h.fanout = fanout; h.hop = hop;
  const int32_t B = static_cast<int32_t>(batch_size);
  k_emit_and_sample<<<grid_for(h.cap), kThreads, 0, st>>>(HopBuf{}, EmitOut{}, 0, h, rowptr, src, etype, n_id, ctr, seed,
                                                         offset + static_cast<uint64_t>(hop) * 1000003ull, local_of);
  STEMGNN_LAUNCH_CHECK();
  k_count_wins<<<grid_for(std::max<int64_t>(h.cap, hop == 0 ? B : 0)), kThreads, 0, st>>>(h, ctr, B, n_id, local_of);
  STEMGNN_LAUNCH_CHECK();
  k_scan_block<<<1, kScanThreads, 0, st>>>(wins, new_base, cnt, nullptr, ctr->nodes + hop + 1, ctr->nodes + hop, h.cap,
                                           ctr->nodes + hop + 2, ctr->nodes + hop + 1, ctr->edges + hop + 1,
                                           ctr->edges + hop, false);
  STEMGNN_LAUNCH_CHECK();
  k_assign_new<<<grid_for(h.cap), kThreads, 0, st>>>(h, static_cast<int32_t>(n_cap), ctr, local_of, n_id, nullptr, nullptr,
                                                     nodes_after);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

size_t stemgnn_sampler_full_finish_workspace_bytes(int64_t num_edges) {
  return num_edges < 0 ? 0 : align_up(static_cast<size_t>(num_edges < 1 ? 1 : num_edges) * 4, 256) + 256;
}

int stemgnn_sampler_full_finish(int32_t* state, int32_t hops, const int32_t* fanouts_host, const int64_t* frontier_caps_host,
                                int32_t* const* cnt_h, int32_t* const* ent_base_h, int32_t* const* s_src_h,
                                int32_t* const* s_type_h, int32_t* local_of, const int32_t* n_id, int64_t num_batch_nodes,
                                int64_t num_batch_edges, int32_t* b_rowptr, int32_t* b_src, int32_t* b_type,
                                int64_t* b_coo, float* inv_deg, int64_t* n_id64, int64_t* type64, const int64_t* x,
                                int64_t* x_out, int32_t* counts, void* workspace, size_t workspace_bytes, void* stream_) {
  if (!state || hops <= 0 || hops > 14 || !fanouts_host || !frontier_caps_host || !cnt_h || !ent_base_h || !s_src_h ||
      !s_type_h || !local_of || !n_id || num_batch_nodes <= 0 || num_batch_edges < 0 || !b_rowptr || !inv_deg || !counts ||
      !workspace || (num_batch_edges > 0 && (!b_src || !b_type || !b_coo)))
    return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(num_batch_nodes) || !fits_i32(num_batch_edges)) return STEMGNN_ERR_TOO_LARGE;
  if (workspace_bytes < stemgnn_sampler_full_finish_workspace_bytes(num_batch_edges)) return STEMGNN_ERR_WORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream_);
  BatchCounters* ctr = reinterpret_cast<BatchCounters*>(state);
  int32_t* b_dst = reinterpret_cast<int32_t*>(align_up(reinterpret_cast<uintptr_t>(workspace), 256));
  EmitOut eo{b_rowptr, b_src, b_type, b_dst, type64, nullptr, static_cast<int32_t>(num_batch_edges)};
  for (int hop = 0; hop < hops; ++hop) {
    if (frontier_caps_host[hop] <= 0 || !cnt_h[hop] || !ent_base_h[hop]) return STEMGNN_ERR_INVALID_ARG;
    HopBuf h{};
    h.s_src = s_src_h[hop]; h.s_type = s_type_h[hop]; h.s_cnt = cnt_h[hop]; h.ent_base = ent_base_h[hop];
    h.edge_base = ent_base_h[hop];
    h.cap = static_cast<int32_t>(frontier_caps_host[hop]); h.fanout = fanouts_host[hop]; h.hop = hop;
    const int blocks = static_cast<int>(grid_for(h.cap));
    k_emit_and_sample<<<static_cast<unsigned>(blocks), kThreads, 0, st>>>(h, eo, blocks, HopBuf{}, nullptr, nullptr, nullptr,
                                                                         n_id, ctr, 0, 0, local_of);
    STEMGNN_LAUNCH_CHECK();
  }
  ViewOut v{};
  v.compact_coo = true;
  v.inv_deg = inv_deg; v.n_id64 = n_id64; v.x_out = x_out; v.x = x;
  k_finish<<<grid_for(std::max<int64_t>(num_batch_edges, num_batch_nodes + 1)), kThreads, 0, st>>>(
      hops, ctr, static_cast<int32_t>(num_batch_nodes), static_cast<int32_t>(num_batch_edges), n_id, local_of, b_rowptr,
      b_src, b_type, b_dst, b_coo, v, counts);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

}  // extern "C"
