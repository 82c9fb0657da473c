// bf16 GEMM mode (stemgnn_linear_set_mode(2): BASELINE config 5, what autocast does to the reference's nn.Linear calls) on
// the vendor library for the large products.
//
// In that mode a product IS a plain bf16 GEMM with fp32 accumulation: both operands rounded to bf16 (nearest even),
// one matrix pass.  The tile kernel of csrc/linear.hip rounds while it stages (no extra pass, 250-290 TFLOP/s at the
// D = 768 shapes); hipBLASLt on pre-rounded operands runs the same product at 660-1 110 TFLOP/s
// (tools/micro/bf16_mode_probe.py), so for a large product a rounding pass + the library is 2.2-2.4x faster.  Here:
//   * k_round_cat: one pass that rounds an activation operand to bf16 -- and, for the layer product
//     lin_l(agg) + lin_r(x), writes both operands side by side ([M, K1 + K2]) so that the two products are ONE GEMM;
//   * lt_matmul: hipblasLtMatmul on row-major operands (a row-major [R, C] matrix is the library's column-major
//     [C, R]); bias through the library's epilogue; heuristics cached per shape;
//   * k_col_stats: the BatchNorm column sums the tile kernel produces in its epilogue, as one pass over y.
// Scratch (the rounded copies, the library's workspace) is a grow-only allocation per (device, stream), like the ticket
// pool of common.h: products of one stream run one after the other.
// The exact-piece mode (the default) and the quantiser's similarity core never come here.
#include "common.h"

#include <hipblaslt/hipblaslt.h>

#include <map>
#include <mutex>
#include <tuple>

namespace stemgnn {
namespace {

constexpr int kThreads = 256;
constexpr size_t kLtWorkspace = 64u << 20;

struct LtState {
  hipblasLtHandle_t handle = nullptr;
  void* workspace = nullptr;
  unsigned char* scratch = nullptr;
  size_t scratch_bytes = 0;
};
std::mutex g_lt_mu;
std::map<std::pair<int, hipStream_t>, LtState> g_lt;
using AlgoKey = std::tuple<int, int, int64_t, int64_t, int64_t, int>;
std::map<AlgoKey, hipblasLtMatmulHeuristicResult_t> g_algo;

// the (device, stream)'s library handle, workspace and at least `bytes` of scratch (nullptr: the library or an allocation
// failed; the caller falls back to the tile kernels)
LtState* lt_state(hipStream_t st, size_t bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(g_lt_mu);
  LtState& s = g_lt[{dev, st}];
  if (!s.handle) {
    if (hipblasLtCreate(&s.handle) != HIPBLAS_STATUS_SUCCESS) { s.handle = nullptr; return nullptr; }
    if (hipMalloc(&s.workspace, kLtWorkspace) != hipSuccess) { (void)hipGetLastError(); s.workspace = nullptr; }
  }
  if (!s.workspace) return nullptr;
  if (bytes > s.scratch_bytes) {
    // grow-only; the old block may still be read by a product in flight on this stream: wait for the stream first
    if (hipStreamSynchronize(st) != hipSuccess) return nullptr;
    if (s.scratch) (void)hipFree(s.scratch);
    s.scratch = nullptr;
    s.scratch_bytes = 0;
    const size_t want = bytes + bytes / 8 + (1u << 20);
    if (hipMalloc(reinterpret_cast<void**>(&s.scratch), want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    s.scratch_bytes = want;
  }
  return &s;
}

// out[m][0 .. K1) = bf16(x1[m]) (zero for m >= x1_rows), out[m][K1 .. K1 + K2) = bf16(x2[m]) (x2 fp32 or already bf16)
__global__ void __launch_bounds__(kThreads)
k_round_cat(const float* __restrict__ x1, int K1, int64_t x1_rows, const void* __restrict__ x2, int x2_kind, int K2,
            int64_t M, uint16_t* __restrict__ out) {
  const int nv = (K1 + K2) / 4;
  const int64_t total = M * nv;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * kThreads) {
    const int64_t m = i / nv;
    const int c = static_cast<int>(i - m * nv) * 4;
    uint2 v;
    if (c < K1) {
      v = m < x1_rows ? pack_rne(ld4(x1 + m * K1 + c)) : make_uint2(0u, 0u);
    } else if (x2_kind == kBF16) {
      v = *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(x2) + m * K2 + (c - K1));
    } else {
      v = pack_rne(ld4(static_cast<const float*>(x2) + m * K2 + (c - K1)));
    }
    *reinterpret_cast<uint2*>(out + m * (K1 + K2) + c) = v;
  }
}

// column sums and sums of squares of y [M, N] per slab of rows: partial[slab][2][N]; slabs >= used are zeroed
__global__ void __launch_bounds__(kThreads)
k_col_stats(const float* __restrict__ y, int64_t M, int N, int rows_per_slab, int used, float* __restrict__ partial) {
  const int slab = blockIdx.x;
  const int c = (blockIdx.y * kThreads + threadIdx.x) * 4;
  if (c >= N) return;
  float4 s = zero4(), q = zero4();
  if (slab < used) {
    const int64_t r0 = static_cast<int64_t>(slab) * rows_per_slab;
    const int64_t r1 = r0 + rows_per_slab < M ? r0 + rows_per_slab : M;
    for (int64_t r = r0; r < r1; ++r) {
      const float4 v = ld4(y + r * N + c);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
    }
  }
  float* p = partial + static_cast<int64_t>(slab) * 2 * N;
  st4(p + c, s);
  st4(p + N + c, q);
}

// The weight gradient's dy operand: rounded to bf16 AND summed over rows (db[n] = sum_m dy[m][n]) in the same pass -- per
// slab of rows, then the slabs (fixed order)
__global__ void __launch_bounds__(kThreads)
k_round_colsum(const float* __restrict__ dy, int64_t M, int N, int rows_per_slab, uint16_t* __restrict__ out,
               float* __restrict__ partial) {
  const int slab = blockIdx.x;
  const int c = (blockIdx.y * kThreads + threadIdx.x) * 4;
  if (c >= N) return;
  float4 s = zero4();
  const int64_t r0 = static_cast<int64_t>(slab) * rows_per_slab;
  const int64_t r1 = r0 + rows_per_slab < M ? r0 + rows_per_slab : M;
  for (int64_t r = r0; r < r1; ++r) {
    const float4 v = ld4(dy + r * N + c);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    *reinterpret_cast<uint2*>(out + r * N + c) = pack_rne(v);
  }
  st4(partial + static_cast<int64_t>(slab) * N + c, s);
}
// 64 columns x 4 slab slices per block; the slices meet in LDS (fixed order)
__global__ void __launch_bounds__(kThreads)
k_col_sum_finish(const float* __restrict__ partial, int slabs, int N, float* __restrict__ db) {
  __shared__ double red[kThreads];
  const int cl = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double s = 0.0;
  if (c < N)
    for (int b = slice; b < slabs; b += 4) s += partial[static_cast<int64_t>(b) * N + c];
  red[threadIdx.x] = s;
  __syncthreads();
  if (slice == 0 && c < N) db[c] = static_cast<float>((red[cl] + red[64 + cl]) + (red[128 + cl] + red[192 + cl]));
}

inline unsigned round_grid(int64_t elems4) {
  int64_t g = (elems4 + kThreads - 1) / kThreads;
  if (g > 8192) g = 8192;
  return static_cast<unsigned>(g < 1 ? 1 : g);
}

// D (column-major m x n, fp32, ld = ldc) = op(A) op(B) (+ bias over D's rows), A / B bf16 column-major
int lt_matmul(LtState* s, hipblasOperation_t ta, hipblasOperation_t tb, int64_t m, int64_t n, int64_t k, const void* A,
              int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, const float* bias, hipStream_t st) {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
  hipblasLtMatmulPreference_t pref = nullptr;
  int rc = STEMGNN_ERR_HIP;
  do {
    if (hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) break;
    const int32_t opa = ta, opb = tb;
    if (hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opa, sizeof(opa)) != HIPBLAS_STATUS_SUCCESS) break;
    if (hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opb, sizeof(opb)) != HIPBLAS_STATUS_SUCCESS) break;
    if (bias) {
      const hipblasLtEpilogue_t ep = HIPBLASLT_EPILOGUE_BIAS;
      const int32_t bt = HIP_R_32F;
      if (hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof(ep)) != HIPBLAS_STATUS_SUCCESS) break;
      if (hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)) != HIPBLAS_STATUS_SUCCESS) break;
      if (hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)) != HIPBLAS_STATUS_SUCCESS) break;
    }
    const int64_t ar = ta == HIPBLAS_OP_N ? m : k, ac = ta == HIPBLAS_OP_N ? k : m;
    const int64_t br = tb == HIPBLAS_OP_N ? k : n, bc = tb == HIPBLAS_OP_N ? n : k;
    if (hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, ar, ac, lda) != HIPBLAS_STATUS_SUCCESS) break;
    if (hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, br, bc, ldb) != HIPBLAS_STATUS_SUCCESS) break;
    if (hipblasLtMatrixLayoutCreate(&lc, HIP_R_32F, m, n, ldc) != HIPBLAS_STATUS_SUCCESS) break;
    hipblasLtMatmulHeuristicResult_t res;
    const AlgoKey key{static_cast<int>(ta), static_cast<int>(tb), m, n, k, bias ? 1 : 0};
    bool have = false;
    {
      std::lock_guard<std::mutex> lock(g_lt_mu);
      auto it = g_algo.find(key);
      if (it != g_algo.end()) { res = it->second; have = true; }
    }
    if (!have) {
      if (hipblasLtMatmulPreferenceCreate(&pref) != HIPBLAS_STATUS_SUCCESS) break;
      const uint64_t wsb = kLtWorkspace;
      if (hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsb, sizeof(wsb)) !=
          HIPBLAS_STATUS_SUCCESS)
        break;
      int found = 0;
      if (hipblasLtMatmulAlgoGetHeuristic(s->handle, desc, la, lb, lc, lc, pref, 1, &res, &found) != HIPBLAS_STATUS_SUCCESS ||
          found < 1) {
        rc = STEMGNN_ERR_INVALID_ARG;  // no kernel for this shape: the caller takes the tile kernels
        break;
      }
      std::lock_guard<std::mutex> lock(g_lt_mu);
      g_algo[key] = res;
    }
    const float alpha = 1.f, beta = 0.f;
    if (hipblasLtMatmul(s->handle, desc, &alpha, A, la, B, lb, &beta, C, lc, C, lc, &res.algo, s->workspace, kLtWorkspace,
                        st) != HIPBLAS_STATUS_SUCCESS)
      break;
    rc = STEMGNN_OK;
  } while (false);
  if (pref) hipblasLtMatmulPreferenceDestroy(pref);
  if (lc) hipblasLtMatrixLayoutDestroy(lc);
  if (lb) hipblasLtMatrixLayoutDestroy(lb);
  if (la) hipblasLtMatrixLayoutDestroy(la);
  if (desc) hipblasLtMatmulDescDestroy(desc);
  return rc;
}

inline size_t a256(size_t b) { return (b + 255) / 256 * 256; }

}  // namespace

// products large enough for a rounding pass + the library to beat the tile kernel that rounds while it stages
bool lt_gemm_ok(int64_t M, int64_t N, int64_t K) {
  // the tile kernel takes ~0.1 ms for 2.5e10 flop in this mode; below that the two extra launches and the host side
  // of the library call eat the gain
  return M >= 8192 && N % 16 == 0 && K % 16 == 0 && N >= 128 && K >= 256 && 2.0 * M * N * K >= 2.5e10;
}

int lt_linear_fwd(const float* x1, const float* w1, int64_t K1, const void* x2, int x2_kind, const float* w2, int64_t K2,
                  const float* bias, int64_t M, int64_t N, float* y, int64_t x1_rows, float* stats_partial,
                  int64_t stats_slabs, hipStream_t st) {
  const int64_t K = K1 + K2;
  const size_t xb = a256(static_cast<size_t>(M) * K * 2), wb = a256(static_cast<size_t>(N) * K * 2);
  LtState* s = lt_state(st, xb + wb);
  if (!s) return STEMGNN_ERR_HIP;
  uint16_t* xr = reinterpret_cast<uint16_t*>(s->scratch);
  uint16_t* wr = reinterpret_cast<uint16_t*>(s->scratch + xb);
  k_round_cat<<<round_grid(M * (K / 4)), kThreads, 0, st>>>(x1, static_cast<int>(K1), x1_rows, x2, x2_kind,
                                                            static_cast<int>(K2), M, xr);
  STEMGNN_LAUNCH_CHECK();
  k_round_cat<<<round_grid(N * (K / 4)), kThreads, 0, st>>>(w1, static_cast<int>(K1), N, w2, kF32, static_cast<int>(K2), N, wr);
  STEMGNN_LAUNCH_CHECK();
  // y^T (column-major [N, M]) = W (column-major [K, N], transposed) x^T (column-major [K, M])
  const int rc = lt_matmul(s, HIPBLAS_OP_T, HIPBLAS_OP_N, N, M, K, wr, K, xr, K, y, N, bias, st);
  if (rc != STEMGNN_OK) return rc;
  if (stats_partial && stats_slabs > 0) {
    const int used = static_cast<int>(std::min<int64_t>(stats_slabs, 1024));
    const int rows = static_cast<int>((M + used - 1) / used);
    dim3 grid(static_cast<unsigned>(stats_slabs), static_cast<unsigned>((N / 4 + kThreads - 1) / kThreads));
    k_col_stats<<<grid, kThreads, 0, st>>>(y, M, static_cast<int>(N), rows, static_cast<int>((M + rows - 1) / rows),
                                          stats_partial);
    STEMGNN_LAUNCH_CHECK();
  }
  return STEMGNN_OK;
}

int lt_linear_bwd_data(const float* dy, const float* w, int64_t M, int64_t N, int64_t K, float* dx, hipStream_t st) {
  const size_t gb = a256(static_cast<size_t>(M) * N * 2), wb = a256(static_cast<size_t>(N) * K * 2);
  LtState* s = lt_state(st, gb + wb);
  if (!s) return STEMGNN_ERR_HIP;
  uint16_t* gr = reinterpret_cast<uint16_t*>(s->scratch);
  uint16_t* wr = reinterpret_cast<uint16_t*>(s->scratch + gb);
  k_round_cat<<<round_grid(M * (N / 4)), kThreads, 0, st>>>(dy, static_cast<int>(N), M, nullptr, kF32, 0, M, gr);
  STEMGNN_LAUNCH_CHECK();
  k_round_cat<<<round_grid(N * (K / 4)), kThreads, 0, st>>>(w, static_cast<int>(K), N, nullptr, kF32, 0, N, wr);
  STEMGNN_LAUNCH_CHECK();
  // dx^T (column-major [K, M]) = W (column-major [K, N]) dy^T (column-major [N, M])
  return lt_matmul(s, HIPBLAS_OP_N, HIPBLAS_OP_N, K, M, N, wr, K, gr, N, dx, K, nullptr, st);
}

int lt_linear_bwd_weight(const float* dy, const void* x, int x_kind, int64_t M, int64_t N, int64_t K, float* dw, float* db,
                         hipStream_t st) {
  const int slabs = static_cast<int>(std::min<int64_t>((M + 127) / 128, 256));  // enough blocks to stream dy, few enough to finish
  const size_t gb = a256(static_cast<size_t>(M) * N * 2), xb = a256(static_cast<size_t>(M) * K * 2);
  const size_t pb = db ? a256(static_cast<size_t>(slabs) * N * 4) : 0;
  LtState* s = lt_state(st, gb + xb + pb);
  if (!s) return STEMGNN_ERR_HIP;
  uint16_t* gr = reinterpret_cast<uint16_t*>(s->scratch);
  uint16_t* xr = reinterpret_cast<uint16_t*>(s->scratch + gb);
  const int rows = static_cast<int>((M + slabs - 1) / slabs), used = static_cast<int>((M + rows - 1) / rows);
  float* partial = reinterpret_cast<float*>(s->scratch + gb + xb);
  if (db) {
    dim3 grid(static_cast<unsigned>(used), static_cast<unsigned>((N / 4 + kThreads - 1) / kThreads));
    k_round_colsum<<<grid, kThreads, 0, st>>>(dy, M, static_cast<int>(N), rows, gr, partial);
  } else {
    k_round_cat<<<round_grid(M * (N / 4)), kThreads, 0, st>>>(dy, static_cast<int>(N), M, nullptr, kF32, 0, M, gr);
  }
  STEMGNN_LAUNCH_CHECK();
  k_round_cat<<<round_grid(M * (K / 4)), kThreads, 0, st>>>(nullptr, 0, 0, x, x_kind, static_cast<int>(K), M, xr);
  STEMGNN_LAUNCH_CHECK();
  // dw^T (column-major [K, N]) = x^T (column-major [K, M]) dy (column-major [N, M], transposed)
  const int rc = lt_matmul(s, HIPBLAS_OP_N, HIPBLAS_OP_T, K, N, M, xr, K, gr, N, dw, K, nullptr, st);
  if (rc != STEMGNN_OK) return rc;
  if (db) {
    k_col_sum_finish<<<static_cast<unsigned>((N + 63) / 64), kThreads, 0, st>>>(partial, used, static_cast<int>(N), db);
    STEMGNN_LAUNCH_CHECK();
  }
  return STEMGNN_OK;
}

}  // namespace stemgnn
