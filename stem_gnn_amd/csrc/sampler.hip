// GPU-resident mini-batch neighbour sampler.
//
// Replaces torch_geometric.loader.NeighborLoader(num_neighbors=[f]*L) as the reference uses it
// (STEM-GNN/pretrain.py:151-153; pyg-lib / torch-sparse C++ sampler on the host): per hop every
// node first reached in the previous hop draws up to `fanout` of its in-neighbours uniformly
// WITHOUT replacement (Floyd's algorithm on a Philox stream), the subgraph keeps every sampled
// edge (neighbour -> node), local numbering puts the seeds first, then new nodes hop by hop in
// order of first appearance.  The full graph's by-target CSR (int32) stays resident in HBM; a
// batch costs a dozen small launches and ONE 8-byte device->host copy (its node / edge counts).
// Because the sampler walks targets in local order, it emits the batch's by-target CSR directly
// (no sort); edge j of the batch is slot j of that CSR.
#include "common.h"

#include <climits>
#include <cstring>
#include <rocprim/device/device_scan.hpp>

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
            int32_t* __restrict__ n_id, BatchCounters* __restrict__ ctr, int32_t* __restrict__ bad) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i == 0) { ctr->nodes[0] = 0; ctr->nodes[1] = B; ctr->edges[0] = 0; }
  if (i >= B) return;
  const int64_t g = seeds[i];
  if (g < 0 || g >= N) { atomicAdd(bad, 1); n_id[i] = 0; return; }
  n_id[i] = static_cast<int32_t>(g);
  // duplicate seeds keep the lowest position (atomicMax over non-negative ids would keep the highest)
  atomicMax(&local_of[g], INT_MAX - i);  // provisional: decoded by k_seed_fix
}

__global__ void __launch_bounds__(kThreads)
k_seed_fix(int32_t B, const int32_t* __restrict__ n_id, int32_t* __restrict__ local_of) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= B) return;
  const int32_t g = n_id[i];
  const int32_t v = local_of[g];
  if (v > INT_MAX - B - 1) local_of[g] = INT_MAX - v;  // first position that named this node
}

// One thread per frontier node: draw min(deg, fanout) distinct in-neighbour slots (Floyd's algorithm: for j = deg - f ..
// deg - 1 draw t ~ U[0, j] and take t unless already taken, else j), emitted in CSR order.  Rows of up to 64 slots
// (every row of the C4 graph) keep the chosen set as a 64-bit mask in a register: membership is a shift, the ascending
// walk a find-first-set -- the round-2 form kept a sorted array, which the compiler put in scratch memory, and took
// 41 us per hop for ten thousand rows; longer rows still take that form.
__global__ void __launch_bounds__(kThreads)
k_sample_hop(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src, const int32_t* __restrict__ etype,
             const int32_t* __restrict__ n_id, const BatchCounters* __restrict__ ctr, int hop, int fanout,
             uint64_t seed, uint64_t offset, int32_t cap_frontier, int32_t* __restrict__ s_src /*[cap*fanout] global ids*/,
             int32_t* __restrict__ s_type, int32_t* __restrict__ s_cnt /*[cap]*/) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= cap_frontier) return;
  const int f0 = ctr->nodes[hop], f1 = ctr->nodes[hop + 1];
  if (i >= f1 - f0) { s_cnt[i] = 0; return; }
  const int32_t v = n_id[f0 + i];
  const int beg = rowptr[v], deg = rowptr[v + 1] - beg;
  const int64_t out0 = static_cast<int64_t>(i) * fanout;
  if (fanout < 0 || deg <= fanout) {  // the whole row (capped at the widest fan-out the buffers hold)
    int c = deg < kMaxFanout ? deg : kMaxFanout;
    if (fanout >= 0 && c > fanout) c = fanout;
    s_cnt[i] = c;
    for (int j = 0; j < c; ++j) {
      s_src[out0 + j] = src[beg + j];
      s_type[out0 + j] = etype ? etype[beg + j] : 0;
    }
    return;
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
    s_cnt[i] = fanout;
    if (fanout <= 16) {
      // ascending slots = CSR order among the chosen edges; all loads of the row in flight before the first store (the
      // reads are scattered over an 80 MB array: one at a time they are a chain of cache and TLB misses)
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
      return;
    }
    for (int j = 0; j < fanout; ++j) {
      const int t = __ffsll(static_cast<unsigned long long>(mask)) - 1;
      mask &= mask - 1;
      s_src[out0 + j] = src[beg + t];
      s_type[out0 + j] = etype ? etype[beg + t] : 0;
    }
    return;
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
    // insertion keeps `chosen` ascending (CSR order among the chosen edges)
    int p = c++;
    while (p > 0 && chosen[p - 1] > t) { chosen[p] = chosen[p - 1]; --p; }
    chosen[p] = t;
  }
  s_cnt[i] = c;
  for (int j = 0; j < c; ++j) {
    s_src[out0 + j] = src[beg + chosen[j]];
    s_type[out0 + j] = etype ? etype[beg + chosen[j]] : 0;
  }
}

// Claim unknown neighbours: the lowest entry position wins (deterministic numbering).
__global__ void __launch_bounds__(kThreads)
k_claim(const int32_t* __restrict__ s_src, const int32_t* __restrict__ s_cnt, int fanout, int32_t cap_entries,
        int32_t* __restrict__ local_of) {
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= cap_entries) return;
  const int i = p / fanout, j = p - i * fanout;
  if (j >= s_cnt[i]) return;
  const int32_t g = s_src[p];
  if (local_of[g] < 0) atomicMax(&local_of[g], -(p + 2));
}

__global__ void __launch_bounds__(kThreads)
k_flag_new(const int32_t* __restrict__ s_src, const int32_t* __restrict__ s_cnt, int fanout, int32_t cap_entries,
           const int32_t* __restrict__ local_of, int32_t* __restrict__ is_new) {
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= cap_entries) return;
  const int i = p / fanout, j = p - i * fanout;
  is_new[p] = (j < s_cnt[i] && local_of[s_src[p]] == -(p + 2)) ? 1 : 0;
}

__global__ void __launch_bounds__(kThreads)
k_assign_new(const int32_t* __restrict__ s_src, const int32_t* __restrict__ is_new, const int32_t* __restrict__ new_pos,
             int32_t cap_entries, int hop, int32_t cap_nodes, BatchCounters* __restrict__ ctr,
             int32_t* __restrict__ local_of, int32_t* __restrict__ n_id) {
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= cap_entries) return;
  const int base = ctr->nodes[hop + 1];
  if (is_new[p]) {
    const int id = base + new_pos[p];
    if (id < cap_nodes) {
      n_id[id] = s_src[p];
      local_of[s_src[p]] = id;
    }
  }
  if (p == cap_entries - 1) ctr->nodes[hop + 2] = base + new_pos[p] + is_new[p];
}

// Write the hop's edges into the batch CSR: frontier node i (local id f0 + i) owns slots
// [e0 + cnt_pos[i], +cnt[i]).
__global__ void __launch_bounds__(kThreads)
k_emit_edges(const int32_t* __restrict__ s_src, const int32_t* __restrict__ s_type, const int32_t* __restrict__ s_cnt,
             const int32_t* __restrict__ cnt_pos, int fanout, int32_t cap_frontier, int hop,
             BatchCounters* __restrict__ ctr, const int32_t* __restrict__ local_of, int32_t cap_edges,
             int32_t* __restrict__ b_rowptr, int32_t* __restrict__ b_src, int32_t* __restrict__ b_type,
             int64_t* __restrict__ b_coo /*[2][cap_edges]*/) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= cap_frontier) return;
  const int f0 = ctr->nodes[hop], nf = ctr->nodes[hop + 1] - f0;
  const int e0 = ctr->edges[hop];
  if (i == cap_frontier - 1) ctr->edges[hop + 1] = e0 + cnt_pos[i] + s_cnt[i];
  if (i >= nf) return;
  const int c = s_cnt[i];
  const int base = e0 + cnt_pos[i];
  b_rowptr[f0 + i] = base;
  for (int j = 0; j < c; ++j) {
    const int slot = base + j;
    if (slot >= cap_edges) break;
    const int32_t ls = local_of[s_src[static_cast<int64_t>(i) * fanout + j]];
    b_src[slot] = ls;
    b_type[slot] = s_type[static_cast<int64_t>(i) * fanout + j];
    b_coo[slot] = ls;
    b_coo[cap_edges + slot] = f0 + i;
  }
}

// rowptr for every node that was never a frontier node (no in-edges) + the closing entry;
// publishes (N_b, E_b); clears the global->local scratch map.
__global__ void __launch_bounds__(kThreads)
k_finish(int hops, const BatchCounters* __restrict__ ctr, int32_t cap_nodes, const int32_t* __restrict__ n_id,
         int32_t* __restrict__ local_of, int32_t* __restrict__ b_rowptr, int32_t* __restrict__ counts /*[3]*/) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  int nb = ctr->nodes[hops + 1];
  if (nb > cap_nodes) nb = cap_nodes;
  const int eb = ctr->edges[hops];
  const int expanded = ctr->nodes[hops];  // nodes [0, expanded) were frontier nodes of some hop
  if (i == 0) { counts[0] = nb; counts[1] = eb; counts[2] = expanded < nb ? expanded : nb; }
  if (i > cap_nodes) return;
  if (i >= expanded && i <= nb) b_rowptr[i] = eb;
  if (i < nb) local_of[n_id[i]] = kUnassigned;
}

__global__ void __launch_bounds__(kThreads) k_fill_i32(int32_t* p, int64_t n, int32_t v) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i < n) p[i] = v;
}

inline unsigned grid_for(int64_t n) { return static_cast<unsigned>((n < 1 ? 1 : n + kThreads - 1) / kThreads); }

inline size_t scan_temp_bytes(size_t n) {
  size_t temp = 0;
  int32_t* p = nullptr;
  hipError_t e = rocprim::exclusive_scan(nullptr, temp, p, p, 0, n, rocprim::plus<int32_t>(), hipStream_t(0), false);
  if (e != hipSuccess || temp == 0) { (void)hipGetLastError(); temp = n * 8 + (1u << 20); }
  return temp;
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
  int64_t cap_frontier = batch_size;
  for (int64_t h = 1; h < hops; ++h) cap_frontier *= fanout;  // last hop's frontier bound
  const int64_t cap_entries = cap_frontier * fanout;
  size_t b = 0;
  b += align_up(static_cast<size_t>(cap_entries) * 4, 256) * 4;   // s_src, s_type, is_new, new_pos
  b += align_up(static_cast<size_t>(cap_frontier) * 4, 256) * 2;  // s_cnt, cnt_pos
  b += align_up(sizeof(BatchCounters), 256) + 256;                 // counters, bad flag
  b += align_up(scan_temp_bytes(static_cast<size_t>(cap_entries)), 256);
  return b + 1024;
}

/*
 * One mini-batch.  Capacities: cap_nodes >= batch_size * (1 + f + ... + f^L), cap_edges >= batch_size *
 * (f + ... + f^L) (both are checked on the host).  Outputs (device): n_id [cap_nodes] int32,
 * b_rowptr [cap_nodes + 1], b_src / b_type [cap_edges] int32, b_coo [2, cap_edges] int64 (row stride
 * cap_edges), counts [2] = (N_b, E_b).
 */
int stemgnn_sample_batch(const int32_t* rowptr, const int32_t* src, const int32_t* etype, int64_t num_nodes,
                         const int64_t* seeds, int64_t batch_size, const int32_t* fanouts_host, int64_t hops,
                         uint64_t seed, uint64_t offset, int32_t* local_of, int64_t cap_nodes, int64_t cap_edges,
                         int32_t* n_id, int32_t* b_rowptr, int32_t* b_src, int32_t* b_type, int64_t* b_coo,
                         int32_t* counts, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (batch_size <= 0 || hops <= 0 || hops > 14 || !fanouts_host || !rowptr || !src || !seeds || !local_of || !n_id ||
      !b_rowptr || !b_src || !b_type || !b_coo || !counts || !workspace)
    return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(num_nodes) || !fits_i32(cap_nodes) || !fits_i32(cap_edges)) return STEMGNN_ERR_TOO_LARGE;
  int64_t fmax = 0, need_nodes = batch_size, need_edges = 0, level = batch_size;
  for (int64_t h = 0; h < hops; ++h) {
    const int64_t f = fanouts_host[h];
    if (f <= 0 || f > kMaxFanout) return STEMGNN_ERR_INVALID_ARG;
    if (f > fmax) fmax = f;
    level *= f;
    need_nodes += level;
    need_edges += level;
  }
  if (cap_nodes < need_nodes || cap_edges < need_edges) return STEMGNN_ERR_WORKSPACE;
  if (workspace_bytes < stemgnn_sampler_workspace_bytes(batch_size, hops, fmax)) return STEMGNN_ERR_WORKSPACE;

  int64_t cap_frontier_max = batch_size;
  for (int64_t h = 1; h < hops; ++h) cap_frontier_max *= fmax;
  const int64_t cap_entries_max = cap_frontier_max * fmax;
  uintptr_t p = align_up(reinterpret_cast<uintptr_t>(workspace), 256);
  auto carve = [&](size_t bytes) { uintptr_t q = p; p += align_up(bytes, 256); return q; };
  int32_t* s_src = reinterpret_cast<int32_t*>(carve(cap_entries_max * 4));
  int32_t* s_type = reinterpret_cast<int32_t*>(carve(cap_entries_max * 4));
  int32_t* is_new = reinterpret_cast<int32_t*>(carve(cap_entries_max * 4));
  int32_t* new_pos = reinterpret_cast<int32_t*>(carve(cap_entries_max * 4));
  int32_t* s_cnt = reinterpret_cast<int32_t*>(carve(cap_frontier_max * 4));
  int32_t* cnt_pos = reinterpret_cast<int32_t*>(carve(cap_frontier_max * 4));
  BatchCounters* ctr = reinterpret_cast<BatchCounters*>(carve(sizeof(BatchCounters)));
  int32_t* bad = reinterpret_cast<int32_t*>(carve(256));
  void* temp = reinterpret_cast<void*>(p);
  const size_t temp_bytes = workspace_bytes - (p - reinterpret_cast<uintptr_t>(workspace));

  STEMGNN_HIP_TRY(hipMemsetAsync(bad, 0, 4, st));
  const int32_t B = static_cast<int32_t>(batch_size);
  k_seed_init<<<grid_for(B), kThreads, 0, st>>>(seeds, B, num_nodes, local_of, n_id, ctr, bad);
  STEMGNN_LAUNCH_CHECK();
  k_seed_fix<<<grid_for(B), kThreads, 0, st>>>(B, n_id, local_of);
  STEMGNN_LAUNCH_CHECK();

  int64_t cap_frontier = batch_size;
  for (int hop = 0; hop < hops; ++hop) {
    const int f = fanouts_host[hop];
    const int32_t cf = static_cast<int32_t>(cap_frontier);
    const int32_t ce = static_cast<int32_t>(cap_frontier * f);
    k_sample_hop<<<grid_for(cf), kThreads, 0, st>>>(rowptr, src, etype, n_id, ctr, hop, f, seed,
                                                    offset + static_cast<uint64_t>(hop) * 1000003ull, cf, s_src,
                                                    s_type, s_cnt);
    STEMGNN_LAUNCH_CHECK();
    k_claim<<<grid_for(ce), kThreads, 0, st>>>(s_src, s_cnt, f, ce, local_of);
    STEMGNN_LAUNCH_CHECK();
    k_flag_new<<<grid_for(ce), kThreads, 0, st>>>(s_src, s_cnt, f, ce, local_of, is_new);
    STEMGNN_LAUNCH_CHECK();
    size_t need = 0;
    STEMGNN_HIP_TRY(rocprim::exclusive_scan(nullptr, need, is_new, new_pos, 0, static_cast<size_t>(ce),
                                            rocprim::plus<int32_t>(), st, false));
    if (need > temp_bytes) return STEMGNN_ERR_WORKSPACE;
    STEMGNN_HIP_TRY(rocprim::exclusive_scan(temp, need, is_new, new_pos, 0, static_cast<size_t>(ce),
                                            rocprim::plus<int32_t>(), st, false));
    k_assign_new<<<grid_for(ce), kThreads, 0, st>>>(s_src, is_new, new_pos, ce, hop, static_cast<int32_t>(cap_nodes),
                                                    ctr, local_of, n_id);
    STEMGNN_LAUNCH_CHECK();
    STEMGNN_HIP_TRY(rocprim::exclusive_scan(nullptr, need, s_cnt, cnt_pos, 0, static_cast<size_t>(cf),
                                            rocprim::plus<int32_t>(), st, false));
    if (need > temp_bytes) return STEMGNN_ERR_WORKSPACE;
    STEMGNN_HIP_TRY(rocprim::exclusive_scan(temp, need, s_cnt, cnt_pos, 0, static_cast<size_t>(cf),
                                            rocprim::plus<int32_t>(), st, false));
    k_emit_edges<<<grid_for(cf), kThreads, 0, st>>>(s_src, s_type, s_cnt, cnt_pos, f, cf, hop, ctr, local_of,
                                                    static_cast<int32_t>(cap_edges), b_rowptr, b_src, b_type, b_coo);
    STEMGNN_LAUNCH_CHECK();
    cap_frontier *= f;
  }
  k_finish<<<grid_for(cap_nodes + 1), kThreads, 0, st>>>(static_cast<int>(hops), ctr, static_cast<int32_t>(cap_nodes),
                                                         n_id, local_of, b_rowptr, counts);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

}  // extern "C"
