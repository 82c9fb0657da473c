// Graph structure build: int64 COO edge_index -> int32 CSR (by target or by source),
// stable in the original edge order, on the device.
//
// Replaces the per-call index handling inside PyG's MessagePassing.propagate
// (reference STEM-GNN/model/encoder.py:82): the reference re-reads the unsorted int64 COO
// with index_select / scatter_add_ (atomics) on every layer call; here the edges are
// grouped once per (sub)graph so the aggregation kernels read neighbour lists contiguously
// and reduce without atomics.  The sort is rocPRIM's LSD radix sort (stable), restricted to
// the bits a node id can occupy.
#include "common.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace stemgnn {

thread_local int g_last_hip_error = 0;

namespace {

constexpr int kThreads = 256;

inline unsigned key_bits(int64_t num_keys_with_sentinel) {
  unsigned b = 1;
  while ((1LL << b) < num_keys_with_sentinel) ++b;
  return b;
}

inline int grid_for(int64_t n) {
  int64_t g = (n + kThreads - 1) / kThreads;
  return static_cast<int>(g < 1 ? 1 : g);
}

// key = endpoint to group by; invalid edges get the sentinel key N and sort to the tail.
__global__ void __launch_bounds__(kThreads) k_edge_keys(const int64_t* __restrict__ ei, int64_t E, int64_t N,
                                                        int key_row, uint32_t* __restrict__ keys,
                                                        uint32_t* __restrict__ vals, int32_t* __restrict__ bad) {
  int64_t e = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (e >= E) return;
  int64_t s = ei[e], d = ei[E + e];
  bool ok = s >= 0 && s < N && d >= 0 && d < N;
  keys[e] = ok ? static_cast<uint32_t>(key_row ? d : s) : static_cast<uint32_t>(N);
  vals[e] = static_cast<uint32_t>(e);
  if (!ok) atomicAdd(bad, 1);
}

__global__ void __launch_bounds__(kThreads) k_plain_keys(const int32_t* __restrict__ in, int64_t n, int64_t num_keys,
                                                         uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= n) return;
  int32_t k = in[i];
  keys[i] = (k >= 0 && k < num_keys) ? static_cast<uint32_t>(k) : static_cast<uint32_t>(num_keys);
  vals[i] = static_cast<uint32_t>(i);
}

// rowptr[r] = first slot whose key >= r (binary search in the sorted keys), r in [0, N].
__global__ void __launch_bounds__(kThreads) k_rowptr(const uint32_t* __restrict__ sorted_keys, int64_t n,
                                                     int64_t num_keys, int32_t* __restrict__ rowptr) {
  int64_t r = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (r > num_keys) return;
  int64_t lo = 0, hi = n;
  uint32_t key = static_cast<uint32_t>(r);
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (sorted_keys[mid] < key) lo = mid + 1; else hi = mid;
  }
  rowptr[r] = static_cast<int32_t>(lo);
}

__global__ void __launch_bounds__(kThreads) k_other_endpoint(const int64_t* __restrict__ ei, int64_t E, int64_t N,
                                                             int key_row, const uint32_t* __restrict__ sorted_keys,
                                                             const int32_t* __restrict__ eid,
                                                             int32_t* __restrict__ other) {
  int64_t s = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (s >= E) return;
  bool ok = sorted_keys[s] < static_cast<uint32_t>(N);
  int64_t e = eid[s];
  other[s] = ok ? static_cast<int32_t>(ei[(key_row ? 0 : E) + e]) : 0;
}

__global__ void __launch_bounds__(kThreads) k_gather_i32(const int32_t* __restrict__ table,
                                                         const int32_t* __restrict__ index, int64_t n,
                                                         int32_t* __restrict__ out) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i < n) out[i] = table[index[i]];
}

__global__ void __launch_bounds__(kThreads) k_inv_degree(const int32_t* __restrict__ rowptr, int64_t N,
                                                         float* __restrict__ inv_deg) {
  int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i >= N) return;
  int d = rowptr[i + 1] - rowptr[i];
  inv_deg[i] = 1.0f / static_cast<float>(d < 1 ? 1 : d);
}

size_t sort_temp_bytes(size_t n, unsigned bits) {
  size_t temp = 0;
  uint32_t* p = nullptr;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, temp, p, p, p, p, n, 0u, bits, hipStream_t(0), false);
  if (e != hipSuccess || temp == 0) {
    (void)hipGetLastError();
    // No device visible (build container): conservative bound — double buffers + histograms.
    temp = 2 * (2 * n * sizeof(uint32_t)) + (4u << 20);
  }
  return temp;
}

struct SortSpace {
  uint32_t *keys_in, *keys_out, *vals_in;
  void* temp;
  size_t temp_bytes;
};

size_t sort_space_bytes(int64_t n, int64_t num_keys) {
  size_t arr = align_up(static_cast<size_t>(n < 1 ? 1 : n) * sizeof(uint32_t), 256);
  return 3 * arr + align_up(sort_temp_bytes(static_cast<size_t>(n < 1 ? 1 : n), key_bits(num_keys + 1)), 256) + 256;
}

bool carve(void* ws, size_t ws_bytes, int64_t n, int64_t num_keys, SortSpace* sp) {
  size_t arr = align_up(static_cast<size_t>(n < 1 ? 1 : n) * sizeof(uint32_t), 256);
  uintptr_t base = align_up(reinterpret_cast<uintptr_t>(ws), 256);
  size_t lead = base - reinterpret_cast<uintptr_t>(ws);
  if (ws_bytes < lead + 3 * arr) return false;
  sp->keys_in = reinterpret_cast<uint32_t*>(base);
  sp->keys_out = reinterpret_cast<uint32_t*>(base + arr);
  sp->vals_in = reinterpret_cast<uint32_t*>(base + 2 * arr);
  sp->temp = reinterpret_cast<void*>(base + 3 * arr);
  sp->temp_bytes = ws_bytes - lead - 3 * arr;
  return true;
}

}  // namespace
}  // namespace stemgnn

using namespace stemgnn;

extern "C" {

int stemgnn_abi_version(void) { return STEMGNN_ABI_VERSION; }

const char* stemgnn_status_string(int status) {
  switch (status) {
    case STEMGNN_OK: return "ok";
    case STEMGNN_ERR_INVALID_ARG: return "invalid argument (null pointer, negative size or unsupported dimension)";
    case STEMGNN_ERR_TOO_LARGE: return "problem too large for int32 CSR (N or E >= 2^31-1)";
    case STEMGNN_ERR_WORKSPACE: return "workspace too small";
    case STEMGNN_ERR_HIP: return "HIP runtime error";
    default: return "unknown status";
  }
}

int stemgnn_last_hip_error(void) { return g_last_hip_error; }

size_t stemgnn_csr_workspace_bytes(int64_t num_nodes, int64_t num_edges) {
  if (num_nodes < 0 || num_edges < 0) return 0;
  return sort_space_bytes(num_edges, num_nodes);
}

int stemgnn_csr_build(const int64_t* edge_index, int64_t E, int64_t N, int key_row, int32_t* rowptr,
                      int32_t* other, int32_t* eid, int32_t* bad_count, void* workspace, size_t workspace_bytes,
                      void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (E < 0 || N < 0 || !rowptr || !bad_count || (key_row != 0 && key_row != 1)) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(E) || !fits_i32(N)) return STEMGNN_ERR_TOO_LARGE;
  STEMGNN_HIP_TRY(hipMemsetAsync(bad_count, 0, sizeof(int32_t), stream));
  if (E == 0) {
    STEMGNN_HIP_TRY(hipMemsetAsync(rowptr, 0, sizeof(int32_t) * (N + 1), stream));
    return STEMGNN_OK;
  }
  if (!edge_index || !other || !eid || !workspace) return STEMGNN_ERR_INVALID_ARG;
  SortSpace sp;
  if (!carve(workspace, workspace_bytes, E, N, &sp)) return STEMGNN_ERR_WORKSPACE;
  unsigned bits = key_bits(N + 1);
  size_t need = 0;
  {
    uint32_t* p = nullptr;
    STEMGNN_HIP_TRY(rocprim::radix_sort_pairs(nullptr, need, p, p, p, p, static_cast<size_t>(E), 0u, bits, stream, false));
  }
  if (need > sp.temp_bytes) return STEMGNN_ERR_WORKSPACE;
  k_edge_keys<<<grid_for(E), kThreads, 0, stream>>>(edge_index, E, N, key_row, sp.keys_in, sp.vals_in, bad_count);
  STEMGNN_LAUNCH_CHECK();
  STEMGNN_HIP_TRY(rocprim::radix_sort_pairs(sp.temp, need, sp.keys_in, sp.keys_out, sp.vals_in,
                                            reinterpret_cast<uint32_t*>(eid), static_cast<size_t>(E), 0u, bits,
                                            stream, false));
  k_rowptr<<<grid_for(N + 1), kThreads, 0, stream>>>(sp.keys_out, E, N, rowptr);
  STEMGNN_LAUNCH_CHECK();
  k_other_endpoint<<<grid_for(E), kThreads, 0, stream>>>(edge_index, E, N, key_row, sp.keys_out, eid, other);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_group_by_key(const int32_t* keys, int64_t n, int64_t num_keys, int32_t* rowptr, int32_t* perm,
                         void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (n < 0 || num_keys < 0 || !rowptr) return STEMGNN_ERR_INVALID_ARG;
  if (!fits_i32(n) || !fits_i32(num_keys)) return STEMGNN_ERR_TOO_LARGE;
  if (n == 0) {
    STEMGNN_HIP_TRY(hipMemsetAsync(rowptr, 0, sizeof(int32_t) * (num_keys + 1), stream));
    return STEMGNN_OK;
  }
  if (!keys || !perm || !workspace) return STEMGNN_ERR_INVALID_ARG;
  SortSpace sp;
  if (!carve(workspace, workspace_bytes, n, num_keys, &sp)) return STEMGNN_ERR_WORKSPACE;
  unsigned bits = key_bits(num_keys + 1);
  size_t need = 0;
  {
    uint32_t* p = nullptr;
    STEMGNN_HIP_TRY(rocprim::radix_sort_pairs(nullptr, need, p, p, p, p, static_cast<size_t>(n), 0u, bits, stream, false));
  }
  if (need > sp.temp_bytes) return STEMGNN_ERR_WORKSPACE;
  k_plain_keys<<<grid_for(n), kThreads, 0, stream>>>(keys, n, num_keys, sp.keys_in, sp.vals_in);
  STEMGNN_LAUNCH_CHECK();
  STEMGNN_HIP_TRY(rocprim::radix_sort_pairs(sp.temp, need, sp.keys_in, sp.keys_out, sp.vals_in,
                                            reinterpret_cast<uint32_t*>(perm), static_cast<size_t>(n), 0u, bits,
                                            stream, false));
  k_rowptr<<<grid_for(num_keys + 1), kThreads, 0, stream>>>(sp.keys_out, n, num_keys, rowptr);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_gather_i32(const int32_t* table, const int32_t* index, int64_t n, int32_t* out, void* stream_) {
  if (n < 0) return STEMGNN_ERR_INVALID_ARG;
  if (n == 0) return STEMGNN_OK;
  if (!table || !index || !out) return STEMGNN_ERR_INVALID_ARG;
  k_gather_i32<<<grid_for(n), kThreads, 0, static_cast<hipStream_t>(stream_)>>>(table, index, n, out);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_inv_degree(const int32_t* rowptr, int64_t N, float* inv_deg, void* stream_) {
  if (N < 0) return STEMGNN_ERR_INVALID_ARG;
  if (N == 0) return STEMGNN_OK;
  if (!rowptr || !inv_deg) return STEMGNN_ERR_INVALID_ARG;
  k_inv_degree<<<grid_for(N), kThreads, 0, static_cast<hipStream_t>(stream_)>>>(rowptr, N, inv_deg);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

}  // extern "C"
