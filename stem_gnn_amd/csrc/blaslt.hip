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
    // batches differ in their row counts by a few per cent from step to step: a kernel is chosen per size BUCKET (the
    // leading four bits of every large extent: 6-12 % wide), not per size -- the choice is re-made only when a bucket is new
    const auto bucket = [](int64_t v) {
      int sh = 0;
      while ((v >> sh) >= 16) ++sh;
      return v < 4096 ? v : ((v >> sh) << sh);
    };
    const AlgoKey key{static_cast<int>(ta), static_cast<int>(tb), bucket(m), bucket(n), bucket(k), bias ? 1 : 0};
    bool have = false;
    {
      std::lock_guard<std::mutex> lock(g_lt_mu);
      auto it = g_algo.find(key);
      if (it != g_algo.end()) { res = it->second; have = true; }
    }
    const float alpha = 1.f, beta = 0.f;
    if (!have) {
      if (hipblasLtMatmulPreferenceCreate(&pref) != HIPBLAS_STATUS_SUCCESS) break;
      const uint64_t wsb = kLtWorkspace;
      if (hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsb, sizeof(wsb)) !=
          HIPBLAS_STATUS_SUCCESS)
        break;
      // The library's first suggestion is not always its fastest kernel for these tall shapes (measured: 64 x 256 tiles
      // picked for a weight gradient, 630 TFLOP/s for the similarity product): the first call of a shape runs up to
      // eight suggestions once each on the real operands (beta = 0: every run writes the same result) and keeps the
      // fastest.  One-time cost per shape: a few launches and stream synchronisations.
      constexpr int kTry = 8;
      hipblasLtMatmulHeuristicResult_t cand[kTry];
      int found = 0;
      if (hipblasLtMatmulAlgoGetHeuristic(s->handle, desc, la, lb, lc, lc, pref, kTry, cand, &found) != HIPBLAS_STATUS_SUCCESS ||
          found < 1) {
        rc = STEMGNN_ERR_INVALID_ARG;  // no kernel for this shape: the caller takes the tile kernels
        break;
      }
      int best = 0;
      if (found > 1) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
          float best_ms = 0.f;
          best = -1;
          for (int i = 0; i < found; ++i) {
            if (cand[i].workspaceSize > kLtWorkspace) continue;
            bool ok = true;
            for (int rep = 0; rep < 2 && ok; ++rep) {  // the first run of a kernel pays its code load
              (void)hipEventRecord(e0, st);
              ok = hipblasLtMatmul(s->handle, desc, &alpha, A, la, B, lb, &beta, C, lc, C, lc, &cand[i].algo, s->workspace,
                                   kLtWorkspace, st) == HIPBLAS_STATUS_SUCCESS;
              (void)hipEventRecord(e1, st);
              ok = ok && hipEventSynchronize(e1) == hipSuccess;
            }
            float ms = 0.f;
            if (!ok || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { (void)hipGetLastError(); continue; }
            if (best < 0 || ms < best_ms) { best = i; best_ms = ms; }
          }
          if (best < 0) best = 0;
        }
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
      }
      res = cand[best];
      std::lock_guard<std::mutex> lock(g_lt_mu);
      g_algo[key] = res;
    }
    if (hipblasLtMatmul(s->handle, desc, &alpha, A, la, B, lb, &beta, C, lc, C, lc, &res.algo, s->workspace, kLtWorkspace,
                        st) != HIPBLAS_STATUS_SUCCESS) {
      if (have) {  // the bucket's kernel does not take this exact size: forget it, the next call chooses again
        std::lock_guard<std::mutex> lock(g_lt_mu);
        g_algo.erase(key);
        rc = STEMGNN_ERR_INVALID_ARG;
      }
      break;
    }
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
  if (x_kind == kBF16) {
    xr = const_cast<uint16_t*>(static_cast<const uint16_t*>(x));  // stored as bf16 already: the operand as it is
  } else {
    k_round_cat<<<round_grid(M * (K / 4)), kThreads, 0, st>>>(nullptr, 0, 0, x, x_kind, static_cast<int>(K), M, xr);
    STEMGNN_LAUNCH_CHECK();
  }
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

// ---------------------------------------------------------------------------------------------------------------
// The quantiser's similarity product at LARGE codebooks (K >= 512 codes of Dc >= 256: BASELINE configs 3 and 5) on the
// vendor library, still exact: the six significant products of the three-piece cut (common.h) are ONE bf16 GEMM over a
// contraction of 6 Dc -- rows [h h m h m l] against codes [h m h l m h] --, then one pass takes the arg-max and the
// commitment terms off the similarity matrix.  The fused tile kernel (csrc/vq.hip: k_vq_assign) re-cuts a row tile for
// every group of 128 codes and reaches 74 TFLOP/s fp32-equivalent at K = 2 048, Dc = 768; the library runs the
// concatenated product at 1.1-1.3 PFLOP/s bf16 (190-220 fp32-equivalent), and the similarity matrix of one head
// (N x K fp32: 0.2 GB at C5) is written and read once.  Same arithmetic up to the order of the fp32 additions.
// ---------------------------------------------------------------------------------------------------------------
namespace stemgnn {
namespace {

// one wave per row: the row cut into its three bf16 pieces, written as [h h m h m l] (6 Dc values), and its squared norm
__global__ void __launch_bounds__(kThreads)
k_split6_rows(const float* __restrict__ x, int64_t row_stride, int64_t N, int Dc, uint16_t* __restrict__ out,
              float* __restrict__ ssq) {
  const int lane = threadIdx.x & 63;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * (kThreads / 64) + (threadIdx.x >> 6);
  if (row >= N) return;
  const float* xr = x + row * row_stride;
  uint16_t* o = out + row * 6 * Dc;
  float acc = 0.f;
  for (int c = 4 * lane; c < Dc; c += 256) {
    const float4 v = ld4(xr + c);
    acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    uint2 h, m, l;
    split3(v, h, m, l);
    *reinterpret_cast<uint2*>(o + c) = h;
    *reinterpret_cast<uint2*>(o + Dc + c) = h;
    *reinterpret_cast<uint2*>(o + 2 * Dc + c) = m;
    *reinterpret_cast<uint2*>(o + 3 * Dc + c) = h;
    *reinterpret_cast<uint2*>(o + 4 * Dc + c) = m;
    *reinterpret_cast<uint2*>(o + 5 * Dc + c) = l;
  }
  acc = wave_sum(acc);
  if (lane == 0) ssq[row] = acc;
}

// the codes' side of the same six products: [h m h l m h]
__global__ void __launch_bounds__(kThreads)
k_split6_codes(const float* __restrict__ e, int64_t K, int Dc, uint16_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * (kThreads / 64) + (threadIdx.x >> 6);
  if (row >= K) return;
  const float* er = e + row * Dc;
  uint16_t* o = out + row * 6 * Dc;
  for (int c = 4 * lane; c < Dc; c += 256) {
    uint2 h, m, l;
    split3(ld4(er + c), h, m, l);
    *reinterpret_cast<uint2*>(o + c) = h;
    *reinterpret_cast<uint2*>(o + Dc + c) = m;
    *reinterpret_cast<uint2*>(o + 2 * Dc + c) = h;
    *reinterpret_cast<uint2*>(o + 3 * Dc + c) = l;
    *reinterpret_cast<uint2*>(o + 4 * Dc + c) = m;
    *reinterpret_cast<uint2*>(o + 5 * Dc + c) = h;
  }
}

// one wave per row of the similarity matrix: arg-max (the lowest index among equals, as torch.argmax), the row norm,
// and the row's commitment term |e - x / |x||^2 = |e|^2 + |x / |x||^2 - 2 <e, x> / |x|; a block's terms -> partial[block]
__global__ void __launch_bounds__(kThreads)
k_argmax_commit(const float* __restrict__ sim, const float* __restrict__ ssq, const float* __restrict__ esq_h, int64_t N,
                int K, int H, int h, float* __restrict__ norm_out, int64_t* __restrict__ ind_out,
                double* __restrict__ partial) {
  __shared__ double red[kThreads / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * (kThreads / 64) + w;
  double term = 0.0;
  if (row < N) {
    const float* sr = sim + row * K;
    float best = -INFINITY;
    int bi = 0;
    for (int c = 4 * lane; c < K; c += 256) {  // a lane's codes ascend: strict '>' keeps the lowest index
      const float4 v = ld4(sr + c);
      if (v.x > best) { best = v.x; bi = c; }
      if (v.y > best) { best = v.y; bi = c + 1; }
      if (v.z > best) { best = v.z; bi = c + 2; }
      if (v.w > best) { best = v.w; bi = c + 3; }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) {
      const float nrm = sqrtf(ssq[row]);
      const float inv = 1.0f / fmaxf(nrm, 1e-12f), xn2 = nrm * inv;  // F.normalize eps
      ind_out[row * H + h] = static_cast<int64_t>(bi);
      norm_out[row * H + h] = nrm;
      if (esq_h) term = static_cast<double>(esq_h[bi] + xn2 * xn2 - 2.0f * best * inv);
    }
  }
  if (lane == 0) red[w] = term;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// The form that hands the per-head codes on (csrc/vq.hip: k_vq_assign with `quant`): one wave per row gathers the winning
// code, writes the straight-through value x^ + (q - x^) (training) or q, optionally x^ itself, and sums |q - x^|^2
__global__ void __launch_bounds__(kThreads)
k_gather_commit(const float* __restrict__ xp, int64_t N, int H, int h, int Dc, const float* __restrict__ embed_h,
                const int64_t* __restrict__ ind, const float* __restrict__ ssq, int training, float* __restrict__ quant,
                float* __restrict__ xn_out, double* __restrict__ partial) {
  __shared__ double red[kThreads / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * (kThreads / 64) + w;
  float sq = 0.f;
  if (row < N) {
    const int64_t HD = static_cast<int64_t>(H) * Dc;
    const float inv = 1.0f / fmaxf(sqrtf(ssq[row]), 1e-12f);
    const float* xr = xp + row * HD + static_cast<int64_t>(h) * Dc;
    const float* qr = embed_h + ind[row * H + h] * Dc;
    for (int c = 4 * lane; c < Dc; c += 256) {
      const float4 x = ld4(xr + c), q = ld4(qr + c);
      const float4 n = make_float4(x.x * inv, x.y * inv, x.z * inv, x.w * inv);
      const float4 d = make_float4(q.x - n.x, q.y - n.y, q.z - n.z, q.w - n.w);
      sq += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
      const float4 o = training ? make_float4(n.x + d.x, n.y + d.y, n.z + d.z, n.w + d.w) : q;  // x + (q - x), vq.py:937
      st4(quant + row * HD + static_cast<int64_t>(h) * Dc + c, o);
      if (xn_out) st4(xn_out + row * HD + static_cast<int64_t>(h) * Dc + c, n);
    }
  }
  sq = wave_sum(sq);
  if (lane == 0) red[w] = static_cast<double>(sq);
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(kThreads)
k_commit_finish(const double* __restrict__ partial, int64_t n, double scale, float* __restrict__ out) {
  __shared__ double red[kThreads];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += kThreads) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = kThreads / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = static_cast<float>(red[0] * scale);
}

}  // namespace

bool lt_vq_assign_ok(int64_t N, int64_t H, int64_t Dc, int64_t K) {
  return N >= 8192 && H >= 1 && K >= 512 && Dc >= 256 && Dc % 16 == 0 && K % 16 == 0 && 6 * Dc <= (1 << 20);
}

// quant == NULL: the lean form (commitment terms from esq); else the codes are gathered (xn optional)
int lt_vq_assign(const float* xp, int64_t N, int64_t H, int64_t Dc, const float* embed, const float* esq, int64_t K,
                 int training, float* xn, float* norm, int64_t* ind, float* quant, float* sqerr, double sq_scale,
                 hipStream_t st) {
  if (!quant && !esq) return STEMGNN_ERR_INVALID_ARG;
  const int64_t rows_blocks = (N + kThreads / 64 - 1) / (kThreads / 64);
  const size_t ab = a256(static_cast<size_t>(N) * 6 * Dc * 2), bb = a256(static_cast<size_t>(K) * 6 * Dc * 2);
  const size_t sb = a256(static_cast<size_t>(N) * K * 4), qb = a256(static_cast<size_t>(N) * 4);
  const size_t pb = a256(static_cast<size_t>(rows_blocks) * H * 8);
  LtState* s = lt_state(st, ab + bb + sb + qb + pb);
  if (!s) return STEMGNN_ERR_HIP;
  uint16_t* a6 = reinterpret_cast<uint16_t*>(s->scratch);
  uint16_t* b6 = reinterpret_cast<uint16_t*>(s->scratch + ab);
  float* sim = reinterpret_cast<float*>(s->scratch + ab + bb);
  float* ssq = reinterpret_cast<float*>(s->scratch + ab + bb + sb);
  double* partial = reinterpret_cast<double*>(s->scratch + ab + bb + sb + qb);
  const int dc = static_cast<int>(Dc);
  for (int64_t h = 0; h < H; ++h) {
    k_split6_rows<<<static_cast<unsigned>(rows_blocks), kThreads, 0, st>>>(xp + h * Dc, H * Dc, N, dc, a6, ssq);
    STEMGNN_LAUNCH_CHECK();
    k_split6_codes<<<static_cast<unsigned>((K + 3) / 4), kThreads, 0, st>>>(embed + h * K * Dc, K, dc, b6);
    STEMGNN_LAUNCH_CHECK();
    // sim^T (column-major [K, N]) = codes6 (column-major [6 Dc, K], transposed) rows6^T (column-major [6 Dc, N])
    const int rc = lt_matmul(s, HIPBLAS_OP_T, HIPBLAS_OP_N, K, N, 6 * Dc, b6, 6 * Dc, a6, 6 * Dc, sim, K, nullptr, st);
    if (rc != STEMGNN_OK) return rc;  // the library has a kernel for every head or for none: nothing half-written matters
    k_argmax_commit<<<static_cast<unsigned>(rows_blocks), kThreads, 0, st>>>(
        sim, ssq, quant ? nullptr : esq + h * K, N, static_cast<int>(K), static_cast<int>(H), static_cast<int>(h), norm, ind,
        partial + h * rows_blocks);
    STEMGNN_LAUNCH_CHECK();
    if (quant) {
      k_gather_commit<<<static_cast<unsigned>(rows_blocks), kThreads, 0, st>>>(
          xp, N, static_cast<int>(H), static_cast<int>(h), dc, embed + h * K * Dc, ind, ssq, training, quant, xn,
          partial + h * rows_blocks);
      STEMGNN_LAUNCH_CHECK();
    }
  }
  k_commit_finish<<<1, kThreads, 0, st>>>(partial, rows_blocks * H, sq_scale, sqerr);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

}  // namespace stemgnn
