// Gradient clipping by global L2 norm (reference pretrain.py:62: torch.nn.utils.clip_grad_norm_(params, 1.0)) as
// three launches over a by-value table of the gradient tensors, instead of ATen's per-call chain (_foreach_norm,
// stack, vector_norm, add, div, clamp, _foreach_mul: 8-9 launches and as many host round trips through Python).
//   total = sqrt(sum_i sum(g_i^2))  (fp64 accumulation in a fixed order: reproducible)
//   coef  = min(1, max_norm / (total + 1e-6));  g_i *= coef  (skipped when coef == 1: x * 1.0f is exact)
#include "common.h"

#include <cmath>

namespace stemgnn {
namespace {

constexpr int kBlock = 256;
constexpr int kMaxTensors = 64;
constexpr int kChunk = 2048;  // elements per block (8 per thread: the reduction is latency-bound, not bandwidth-bound)

struct TensorTable {
  float* p[kMaxTensors];
  int64_t n[kMaxTensors];
  int32_t first_block[kMaxTensors + 1];
  int32_t count;
};

__device__ inline int find_tensor(const TensorTable& t, int b) {
  int i = 0;
  while (i + 1 < t.count && b >= t.first_block[i + 1]) ++i;
  return i;
}

// ... and the block that arrives last turns the partials into (total norm, clipping factor): no finishing launch
// (common.h: ticket_last).  The sum over the partials runs in index order whoever that block is.
__global__ void __launch_bounds__(kBlock) k_sumsq_partial(TensorTable t, double* __restrict__ partial,
                                                          unsigned int* counter, float max_norm,
                                                          float* __restrict__ out /*[2]: total norm, coef*/) {
  __shared__ double red[kBlock];
  const int b = blockIdx.x, i = find_tensor(t, b);
  const int64_t beg = static_cast<int64_t>(b - t.first_block[i]) * kChunk;
  const int64_t end = beg + kChunk < t.n[i] ? beg + kChunk : t.n[i];
  const float* g = t.p[i];
  double s = 0.0;
  for (int64_t j = beg + threadIdx.x; j < end; j += kBlock) {
    const double v = g[j];
    s += v * v;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) { st_agent(partial + b, red[0]); wait_stores(); }
  if (!ticket_last(counter)) return;
  double tot = 0.0;
  const int n = gridDim.x;
  for (int i = threadIdx.x; i < n; i += kBlock) tot += ld_agent(partial + i);
  red[threadIdx.x] = tot;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float total = static_cast<float>(sqrt(red[0]));
    float coef = max_norm / (total + 1e-6f);  // clip_grad.py: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
    if (!(coef < 1.0f)) coef = 1.0f;
    out[0] = total;
    out[1] = coef;
  }
}

__global__ void __launch_bounds__(kBlock) k_norm_finish(const double* __restrict__ partial, int n, float max_norm,
                                                        float* __restrict__ out /*[2]: total norm, coef*/) {
  __shared__ double red[kBlock];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += kBlock) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float total = static_cast<float>(sqrt(red[0]));
    float coef = max_norm / (total + 1e-6f);  // clip_grad.py: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
    if (!(coef < 1.0f)) coef = 1.0f;
    out[0] = total;
    out[1] = coef;
  }
}

__global__ void __launch_bounds__(kBlock) k_scale_tensors(TensorTable t, const float* __restrict__ out) {
  const float coef = out[1];
  if (coef >= 1.0f) return;
  const int b = blockIdx.x, i = find_tensor(t, b);
  const int64_t beg = static_cast<int64_t>(b - t.first_block[i]) * kChunk;
  const int64_t end = beg + kChunk < t.n[i] ? beg + kChunk : t.n[i];
  float* g = t.p[i];
  for (int64_t j = beg + threadIdx.x; j < end; j += kBlock) g[j] *= coef;
}

inline int64_t blocks_of(int64_t n) { return (n + kChunk - 1) / kChunk; }

// AdamW (torch.optim.AdamW, amsgrad = False, maximize = False; reference pretrain.py:134-136) over a by-value table:
//   p *= 1 - lr wd;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
// with g read as grad * grad_coef[0] when grad_coef != NULL (the clipping factor of stemgnn_grad_norm_coef, so the
// clipped gradient is never written back).
struct AdamTable {
  float* p[kMaxTensors];
  const float* g[kMaxTensors];
  float* m[kMaxTensors];
  float* v[kMaxTensors];
  int64_t n[kMaxTensors];
  int32_t first_block[kMaxTensors + 1];
  int32_t count;
};

__global__ void __launch_bounds__(kBlock)
k_adamw(AdamTable t, float lr, float beta1, float beta2, float eps, float weight_decay, float step_size,
        float bc2_sqrt, const float* __restrict__ grad_coef) {
  const int b = blockIdx.x;
  int i = 0;
  while (i + 1 < t.count && b >= t.first_block[i + 1]) ++i;
  const int64_t beg = static_cast<int64_t>(b - t.first_block[i]) * kChunk;
  const int64_t end = beg + kChunk < t.n[i] ? beg + kChunk : t.n[i];
  const float coef = grad_coef ? grad_coef[0] : 1.0f;
  float* p = t.p[i];
  const float* g = t.g[i];
  float* m = t.m[i];
  float* v = t.v[i];
  for (int64_t j = beg + threadIdx.x; j < end; j += kBlock) {
    const float gj = g[j] * coef;
    float pj = p[j];
    pj -= lr * weight_decay * pj;
    const float mj = m[j] + (gj - m[j]) * (1.0f - beta1);  // exp_avg.lerp_(grad, 1 - beta1)
    const float vj = beta2 * v[j] + (1.0f - beta2) * gj * gj;
    const float denom = sqrtf(vj) / bc2_sqrt + eps;
    pj -= step_size * mj / denom;
    p[j] = pj;
    m[j] = mj;
    v[j] = vj;
  }
}

}  // namespace
}  // namespace stemgnn

using namespace stemgnn;

extern "C" {

int32_t stemgnn_clip_grad_max_tensors(void) { return kMaxTensors; }

size_t stemgnn_clip_grad_workspace_bytes(int64_t total_elements, int32_t count) {
  if (total_elements < 0 || count < 0) return 0;
  return static_cast<size_t>(total_elements / kChunk + count + 1) * sizeof(double) + 256;
}

int stemgnn_adamw_step(float* const* params, const float* const* grads, float* const* exp_avg,
                       float* const* exp_avg_sq, const int64_t* sizes, int32_t count, float lr, float beta1,
                       float beta2, float eps, float weight_decay, int64_t step, const float* grad_coef,
                       void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (count < 0 || count > kMaxTensors || step < 1 || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f))
    return STEMGNN_ERR_INVALID_ARG;
  AdamTable t;
  int64_t blocks = 0;
  t.count = 0;
  for (int i = 0; i < count; ++i) {
    if (sizes[i] < 0) return STEMGNN_ERR_INVALID_ARG;
    if (sizes[i] == 0) continue;
    if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i]) return STEMGNN_ERR_INVALID_ARG;
    t.p[t.count] = params[i]; t.g[t.count] = grads[i]; t.m[t.count] = exp_avg[i]; t.v[t.count] = exp_avg_sq[i];
    t.n[t.count] = sizes[i];
    t.first_block[t.count] = static_cast<int32_t>(blocks);
    blocks += blocks_of(sizes[i]);
    ++t.count;
    if (!fits_i32(blocks)) return STEMGNN_ERR_TOO_LARGE;
  }
  t.first_block[t.count] = static_cast<int32_t>(blocks);
  if (blocks == 0) return STEMGNN_OK;
  const double bc1 = 1.0 - pow(static_cast<double>(beta1), static_cast<double>(step));
  const double bc2 = 1.0 - pow(static_cast<double>(beta2), static_cast<double>(step));
  k_adamw<<<static_cast<unsigned>(blocks), kBlock, 0, st>>>(t, lr, beta1, beta2, eps, weight_decay,
                                                           static_cast<float>(lr / bc1),
                                                           static_cast<float>(sqrt(bc2)), grad_coef);
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_grad_norm_coef(const float* const* grads, const int64_t* sizes, int32_t count, float max_norm, float* out,
                           void* workspace, size_t workspace_bytes, void* stream_) {
  // the first two launches of stemgnn_clip_grad_norm: out[0] = total norm, out[1] = min(1, max_norm / (total + 1e-6));
  // the gradients are left untouched (stemgnn_adamw_step applies out[1] while it reads them)
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (count < 0 || count > kMaxTensors || !out || !(max_norm > 0.f)) return STEMGNN_ERR_INVALID_ARG;
  TensorTable t;
  int64_t blocks = 0, total = 0;
  t.count = 0;
  for (int i = 0; i < count; ++i) {
    if (sizes[i] < 0 || (sizes[i] > 0 && !grads[i])) return STEMGNN_ERR_INVALID_ARG;
    if (sizes[i] == 0) continue;
    t.p[t.count] = const_cast<float*>(grads[i]);
    t.n[t.count] = sizes[i];
    t.first_block[t.count] = static_cast<int32_t>(blocks);
    blocks += blocks_of(sizes[i]);
    total += sizes[i];
    ++t.count;
    if (!fits_i32(blocks)) return STEMGNN_ERR_TOO_LARGE;
  }
  t.first_block[t.count] = static_cast<int32_t>(blocks);
  if (!workspace || workspace_bytes < stemgnn_clip_grad_workspace_bytes(total, count)) return STEMGNN_ERR_WORKSPACE;
  double* partial = reinterpret_cast<double*>(align_up(reinterpret_cast<uintptr_t>(workspace), 256));
  if (blocks > 0) {
    unsigned int* counter = ticket_counter(st);
    if (!counter) return STEMGNN_ERR_HIP;
    k_sumsq_partial<<<static_cast<unsigned>(blocks), kBlock, 0, st>>>(t, partial, counter, max_norm, out);
  } else {
    k_norm_finish<<<1, kBlock, 0, st>>>(partial, 0, max_norm, out);
  }
  STEMGNN_LAUNCH_CHECK();
  return STEMGNN_OK;
}

int stemgnn_clip_grad_norm(float* const* grads, const int64_t* sizes, int32_t count, float max_norm, float* out,
                           void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t st = static_cast<hipStream_t>(stream_);
  if (count < 0 || count > kMaxTensors || !out || !(max_norm > 0.f)) return STEMGNN_ERR_INVALID_ARG;
  TensorTable t;
  int64_t blocks = 0, total = 0;
  t.count = 0;
  for (int i = 0; i < count; ++i) {
    if (sizes[i] < 0 || (sizes[i] > 0 && !grads[i])) return STEMGNN_ERR_INVALID_ARG;
    if (sizes[i] == 0) continue;
    t.p[t.count] = grads[i];
    t.n[t.count] = sizes[i];
    t.first_block[t.count] = static_cast<int32_t>(blocks);
    blocks += blocks_of(sizes[i]);
    total += sizes[i];
    ++t.count;
    if (!fits_i32(blocks)) return STEMGNN_ERR_TOO_LARGE;
  }
  t.first_block[t.count] = static_cast<int32_t>(blocks);
  if (!workspace || workspace_bytes < stemgnn_clip_grad_workspace_bytes(total, count)) return STEMGNN_ERR_WORKSPACE;
  double* partial = reinterpret_cast<double*>(align_up(reinterpret_cast<uintptr_t>(workspace), 256));
  if (blocks > 0) {
    unsigned int* counter = ticket_counter(st);
    if (!counter) return STEMGNN_ERR_HIP;
    k_sumsq_partial<<<static_cast<unsigned>(blocks), kBlock, 0, st>>>(t, partial, counter, max_norm, out);
  } else {
    k_norm_finish<<<1, kBlock, 0, st>>>(partial, 0, max_norm, out);
  }
  STEMGNN_LAUNCH_CHECK();
  if (blocks > 0) {
    k_scale_tensors<<<static_cast<unsigned>(blocks), kBlock, 0, st>>>(t, out);
    STEMGNN_LAUNCH_CHECK();
  }
  return STEMGNN_OK;
}

}  // extern "C"
