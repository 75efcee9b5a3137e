// Classifier head of the node-classification experiments: mean cross-entropy over the LABELLED rows of the logits and its gradient,
// one launch.
//
// Reference: experiments/classify_nodes.py:107-110 -- `criterion(model()[train_idx, :], train_lbl)` with nn.CrossEntropyLoss() --
// which ATen runs as a gather, log-softmax, nll-loss, and in the backward nll-loss backward, log-softmax backward and an
// index_put with accumulation (a radix sort, an arange, three index-arithmetic kernels, a zero fill): ~13 launches around 200 rows
// of 4 numbers -- and memset nodes inside a captured step, which this ROCm runtime replays wrongly (torch_rgcn/__init__.py).
//   blocks 0 .. nb-1 : one thread per node: dlogits[n, :] = (softmax(logits[n, :]) - onehot(label_n)) / n_lab for labelled nodes, 0 else
//   block  nb        : loss = -1 / n_lab * sum over the labelled rows of log softmax(logits[row])[label]   (double accumulation,
//                      fixed order: bit-reproducible)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "rgcn_hip.h"

extern "C" void rgcn_set_error(const char *fmt, ...);

namespace {
constexpr int HB = 256;
constexpr int MAXC = 64;

__global__ __launch_bounds__(HB) void ce_head_kernel(const float *__restrict__ logits, const int *__restrict__ row_label,
                                                      const int *__restrict__ lab_rows, float *__restrict__ loss,
                                                      float *__restrict__ dlogits, long long N, int C, int ld, int n_lab, int nb) {
  if ((int)blockIdx.x < nb) {
    const long long n = (long long)blockIdx.x * HB + threadIdx.x;
    if (n >= N) return;
    const int lbl = row_label[n];
    float *g = dlogits + n * ld;
    for (int c = C; c < ld; ++c) g[c] = 0.f;                 // rows of ld >= C floats (a layer's padded output): the padding's gradient is 0
    if (lbl < 0) {
      for (int c = 0; c < C; ++c) g[c] = 0.f;
      return;
    }
    const float *x = logits + n * ld;
    float m = x[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(x[c] - m);
    const float inv = 1.f / (s * (float)n_lab), il = 1.f / (float)n_lab;
    for (int c = 0; c < C; ++c) g[c] = expf(x[c] - m) * inv - (c == lbl ? il : 0.f);
    return;
  }
  __shared__ double part[HB];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_lab; i += HB) {
    const long long n = lab_rows[i];
    const int lbl = row_label[n];
    const float *x = logits + n * ld;
    float m = x[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(x[c] - m);
    acc += (double)(m + logf(s) - x[lbl]);
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int off = HB / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)(part[0] / (double)n_lab);
}
// dst[a][b][c] = (b < B && c < C) ? src[a][b][c] : 0 over dst [A][Bd][Cd]: zero-padding (Bd >= B, Cd >= C) or cropping (Bd <= B, Cd <= C)
// of the two trailing dimensions; a second, 1-D tensor (the bias) rides in the same launch
__global__ __launch_bounds__(HB) void resize3_kernel(const float *__restrict__ src, float *__restrict__ dst, long long A, int B, int C,
                                                      int Bd, int Cd, const float *__restrict__ src1, float *__restrict__ dst1, int n1,
                                                      int n1d) {
  const long long total = A * Bd * Cd;
  for (long long i = (long long)blockIdx.x * HB + threadIdx.x; i < total + n1d; i += (long long)gridDim.x * HB) {
    if (i < total) {
      const int c = (int)(i % Cd), b = (int)((i / Cd) % Bd);
      const long long a = i / ((long long)Cd * Bd);
      dst[i] = (b < B && c < C) ? src[(a * B + b) * C + c] : 0.f;
    } else {
      const int j = (int)(i - total);
      dst1[j] = j < n1 ? src1[j] : 0.f;
    }
  }
}

// Loss head of the link-prediction experiments: loss = mean over the scored triples of the binary cross-entropy with logits, and
// dscores = (sigmoid(x) - y) / T, one launch (reference experiments/predict_links.py:152-153: F.binary_cross_entropy_with_logits --
// ATen: ~13 elementwise / reduction launches around 330 k scalars).  Every block writes its pieces of dscores and a partial sum in
// double; the block that finishes LAST adds the partials in block order: one launch, bit-reproducible.
constexpr int BCE_BLOCKS = HB;       // (the last block adds one partial per thread)
__global__ __launch_bounds__(HB) void bce_head_kernel(const float *__restrict__ x, const float *__restrict__ y, float *__restrict__ loss,
                                                       float *__restrict__ dx, double *__restrict__ partial, unsigned *__restrict__ ticket,
                                                       long long T) {
  __shared__ double part[HB];
  __shared__ bool is_last;
  double acc = 0.0;
  const float inv = 1.f / (float)T;
  for (long long i = (long long)blockIdx.x * HB + threadIdx.x; i < T; i += (long long)gridDim.x * HB) {
    const float v = x[i], t = y[i];
    const float e = expf(-fabsf(v));                         // the stable form ATen uses: max(x, 0) - x y + log1p(exp(-|x|))
    acc += (double)(fmaxf(v, 0.f) - v * t + log1pf(e));
    const float sg = v >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
    dx[i] = (sg - t) * inv;
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int off = HB / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    // one 128-byte line per block, by an agent-scope atomic exchange whose return is awaited before the ticket: see table_flush_ordered (rgcn_basis.hip)
    const double seen = __hip_atomic_exchange(partial + 16 * blockIdx.x, part[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(seen) : "memory");                 // the exchange has RETURNED (its value is in a register) and nothing moves across
    // release / acquire on the ticket: the order of partial and ticket does not rest on code generation; a tiny kernel, the
    // L2 write-back of a release costs it nothing
    is_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  }
  __syncthreads();
  if (is_last) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                                            // the partials in block order, by the whole block (a lone thread's 256 dependent
                                                            // round trips took longer than the rest of the kernel)
    part[threadIdx.x] = threadIdx.x < gridDim.x ? __hip_atomic_load(partial + 16 * threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    __syncthreads();
    for (int off = HB / 2; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      loss[0] = (float)(part[0] / (double)T);
      *ticket = 0u;                                          // ready for the next launch (a captured step replays this kernel)
    }
  }
}
}  // namespace

extern "C" int rgcn_bce_head_workspace_bytes(void) { return (int)(BCE_BLOCKS * 128 + 256); }

extern "C" int rgcn_bce_head_f32(const float *scores, const float *labels, float *loss, float *dscores, void *workspace, int64_t T,
                                 void *stream) {
  if (!scores || !labels || !loss || !dscores || !workspace || T <= 0) { rgcn_set_error("bce_head: bad argument"); return RGCN_EINVAL; }
  const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((T + HB - 1) / HB, BCE_BLOCKS));
  unsigned *ticket = reinterpret_cast<unsigned *>(workspace);
  double *partial = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(workspace) + 128 + 127) & ~(uintptr_t)127);
  hipLaunchKernelGGL(bce_head_kernel, dim3(grid), dim3(HB), 0, (hipStream_t)stream, scores, labels, loss, dscores, partial, ticket, (long long)T);
  if (hipGetLastError() != hipSuccess) { rgcn_set_error("bce_head: launch failed"); return RGCN_EHIP; }
  return RGCN_OK;
}

extern "C" int rgcn_resize3_f32(const float *src, float *dst, int64_t A, int32_t B, int32_t C, int32_t Bd, int32_t Cd, const float *src1,
                                float *dst1, int32_t n1, int32_t n1d, void *stream) {
  if (!src || !dst || A <= 0 || B <= 0 || C <= 0 || Bd <= 0 || Cd <= 0 || ((src1 == nullptr) != (dst1 == nullptr)) || (src1 && (n1 <= 0 || n1d <= 0))) {
    rgcn_set_error("resize3: bad argument");
    return RGCN_EINVAL;
  }
  const long long total = (long long)A * Bd * Cd + (src1 ? n1d : 0);
  const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((total + HB - 1) / HB, 4096));
  hipLaunchKernelGGL(resize3_kernel, dim3(grid), dim3(HB), 0, (hipStream_t)stream, src, dst, (long long)A, B, C, Bd, Cd, src1, dst1, src1 ? n1 : 0,
                     src1 ? n1d : 0);
  if (hipGetLastError() != hipSuccess) { rgcn_set_error("resize3: launch failed"); return RGCN_EHIP; }
  return RGCN_OK;
}

extern "C" int rgcn_ce_head_f32(const float *logits, const int32_t *row_label, const int32_t *lab_rows, float *loss, float *dlogits,
                                int64_t N, int32_t C, int32_t ld, int32_t n_lab, void *stream) {
  if (!logits || !row_label || !lab_rows || !loss || !dlogits || N <= 0 || C <= 0 || ld < C || n_lab <= 0) { rgcn_set_error("ce_head: bad argument"); return RGCN_EINVAL; }
  if (C > MAXC) { rgcn_set_error("ce_head: %d classes (at most %d)", C, MAXC); return RGCN_EUNSUPPORTED; }
  const int nb = (int)((N + HB - 1) / HB);
  hipLaunchKernelGGL(ce_head_kernel, dim3((unsigned)(nb + 1)), dim3(HB), 0, (hipStream_t)stream, logits, row_label, lab_rows, loss, dlogits,
                     (long long)N, C, ld, n_lab, nb);
  if (hipGetLastError() != hipSuccess) { rgcn_set_error("ce_head: launch failed"); return RGCN_EHIP; }
  return RGCN_OK;
}
