// Ranking evaluator of the link-prediction experiments: every entity scored as the
// head or the tail of each test triple, known true triples masked, rank of the target counted with ties.
// Reference lines taken over: utils/misc.py:60-110 (evaluate: toscore expansion :78-83, model call :85,
// filter_scores :40-58, raw_ranks / num_ties :93-99) and torch_rgcn/layers.py:87-98 (DistMult.forward on the
// expanded [bn, N, 3] index tensor, which is what the reference scores -- three gathers of bn x N x d floats).
//
// Here the [bn, N, 3] index tensor is never built.  For a batch of Q queries
//     scores[q, n] = sum_k (nodes[fixed_q, k] * rel[p_q, k]) * nodes[n, k]  (+ biases)
// is an NT product of the Q x d query vectors with the N x d entity table: fp32 MFMA (v_mfma_f32_16x16x4_f32,
// exact fp32 products and accumulation), one 64 x 64 score tile per workgroup, operands straight from L2 into
// registers (no LDS: each lane's float4 feeds four MFMAs of each of the two tiles that share it).
// The K index is permuted -- lane group kq carries k = 16t + 4kq + c at MFMA c of step t -- which is legal
// because both operands use the same permutation and the sum over k does not care.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>

#include "rgcn_hip.h"

extern "C" void rgcn_set_error(const char *fmt, ...);

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) {                                                             \
      rgcn_set_error("%s failed: %s", #expr, hipGetErrorString(e_));                    \
      return RGCN_EHIP;                                                                 \
    }                                                                                   \
  } while (0)

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int WG = 256;

// query vectors and the query-side bias terms; one wave per query
__global__ __launch_bounds__(WG) void rank_query_kernel(
    const long long *__restrict__ batch, int Q, int head, const float *__restrict__ nodes,
    const float *__restrict__ rel, const float *__restrict__ sbias, const float *__restrict__ pbias,
    const float *__restrict__ obias, float *__restrict__ qvec, float *__restrict__ qb, int d) {
  const int q = blockIdx.x * (WG / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (q >= Q) return;
  const long long s = batch[3 * q], p = batch[3 * q + 1], o = batch[3 * q + 2];
  const long long fixed = head ? o : s;
  for (int k = lane; k < d; k += 64) qvec[(size_t)q * d + k] = nodes[(size_t)fixed * d + k] * rel[(size_t)p * d + k];
  if (lane == 0 && pbias) {
    qb[2 * q] = pbias[p];
    qb[2 * q + 1] = head ? obias[o] : sbias[s];
  }
}

// four K-adjacent operand values of one row.  The address is clamped into the row and the tail (k >= d) is zeroed by
// mask_k4 at the point of USE: a select right after the load would make hipcc wait for the load there.
template <bool VEC>
__device__ __forceinline__ f32x4 load_k4(const float *__restrict__ row, int k, int d) {
  if (VEC) return *reinterpret_cast<const f32x4 *>(row + min(k, d - 4));   // d % 4 == 0: rows are 16-byte aligned
  f32x4 v;
#pragma unroll
  for (int c = 0; c < 4; ++c) v[c] = row[min(k + c, d - 1)];
  return v;
}
template <bool VEC>
__device__ __forceinline__ f32x4 mask_k4(f32x4 v, int k, int d) {
  if (VEC) return k < d ? v : f32x4{0.f, 0.f, 0.f, 0.f};                  // k < d implies k + 3 < d
#pragma unroll
  for (int c = 0; c < 4; ++c) v[c] = (k + c < d) ? v[c] : 0.f;
  return v;
}

// scores tile: 64 queries x 64 candidates per workgroup, 32 x 32 per wave (2 x 2 MFMA tiles)
template <bool VEC>
__global__ __launch_bounds__(WG) void score_all_kernel(
    const float *__restrict__ qvec, const float *__restrict__ qb, const float *__restrict__ nodes,
    const float *__restrict__ cbias, float *__restrict__ scores, int Q, long long N, int d, int head) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = lane & 15, kq = lane >> 4;
  const int q0 = blockIdx.y * 64 + (wave >> 1) * 32;
  const long long c0 = (long long)blockIdx.x * 64 + (wave & 1) * 32;
  const float *arow[2], *brow[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    arow[a] = qvec + (size_t)min(q0 + 16 * a + i, Q - 1) * d;
    brow[a] = nodes + (size_t)min(c0 + 16 * a + i, N - 1) * d;
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int steps = (d + 15) / 16;
  f32x4 av[2], bv[2], an[2], bn[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    av[a] = mask_k4<VEC>(load_k4<VEC>(arow[a], 4 * kq, d), 4 * kq, d);
    bv[a] = mask_k4<VEC>(load_k4<VEC>(brow[a], 4 * kq, d), 4 * kq, d);
  }
  for (int t = 0; t < steps; ++t) {
    const int kn = 16 * min(t + 1, steps - 1) + 4 * kq;       // next step's operands fly during this step's MFMAs
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      an[a] = load_k4<VEC>(arow[a], kn, d);
      bn[a] = load_k4<VEC>(brow[a], kn, d);
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the loads above the MFMAs (hipcc sinks them to their first use)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a][c], bv[b][c], acc[a][b], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < 2; ++a) { av[a] = mask_k4<VEC>(an[a], kn, d); bv[a] = mask_k4<VEC>(bn[a], kn, d); }
  }
  // D: lane 16*kq + i holds query rows 4kq..4kq+3 of the tile, candidate column i
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qrow = q0 + 16 * a + 4 * kq + r;
      if (qrow >= Q) continue;
      float b1 = 0.f, b2 = 0.f;
      if (qb) { b1 = qb[2 * qrow]; b2 = qb[2 * qrow + 1]; }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const long long col = c0 + 16 * b + i;
        if (col >= N) continue;
        float sc = acc[a][b][r];
        if (qb) {   // layers.py:96: scores + (sbias[s] + pbias[p] + obias[o]), same association
          const float cb = cbias[col];
          sc += head ? ((cb + b1) + b2) : ((b2 + b1) + cb);
        }
        scores[(size_t)qrow * N + col] = sc;
      }
    }
}

__global__ void rank_filter_kernel(float *__restrict__ scores, long long N, const int *__restrict__ fq,
                                   const int *__restrict__ fn, long long F) {
  for (long long e = (long long)blockIdx.x * WG + threadIdx.x; e < F; e += (long long)gridDim.x * WG)
    scores[(size_t)fq[e] * N + fn[e]] = -INFINITY;
}

// one workgroup per query: #scores above the target's and #scores equal to it (the target included)
__global__ __launch_bounds__(WG) void rank_count_kernel(const float *__restrict__ scores,
                                                        const long long *__restrict__ batch, int head, long long N,
                                                        long long *__restrict__ greater, long long *__restrict__ ties) {
  __shared__ int red[2 * (WG / 64)];
  const int q = blockIdx.x;
  const float *row = scores + (size_t)q * N;
  const float t = row[batch[3 * q + (head ? 0 : 2)]];
  int g = 0, e = 0;
  for (long long n = threadIdx.x; n < N; n += WG) {
    const float s = row[n];
    g += s > t;
    e += s == t;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    g += __shfl_down(g, off);
    e += __shfl_down(e, off);
  }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = g; red[WG / 64 + (threadIdx.x >> 6)] = e; }
  __syncthreads();
  if (threadIdx.x == 0) {
    long long gs = 0, es = 0;
    for (int w = 0; w < WG / 64; ++w) { gs += red[w]; es += red[WG / 64 + w]; }
    greater[q] = gs;
    ties[q] = es;
  }
}

}  // namespace

extern "C" int rgcn_distmult_score_all_f32(const int64_t *batch, int64_t Q, int32_t head, const float *nodes,
                                           const float *rel, const float *sbias, const float *pbias,
                                           const float *obias, float *qvec, float *qbias, float *scores,
                                           int64_t n_nodes, int32_t n_rel, int32_t d, void *stream) {
  (void)n_rel;
  if (Q < 0 || n_nodes <= 0 || d <= 0 || Q > INT32_MAX || (Q && (!batch || !nodes || !rel || !qvec || !scores))) {
    rgcn_set_error("distmult_score_all: bad argument");
    return RGCN_EINVAL;
  }
  if ((sbias != nullptr) != (pbias != nullptr) || (sbias != nullptr) != (obias != nullptr) || (sbias && !qbias)) {
    rgcn_set_error("distmult_score_all: biases must be all set (with the qbias scratch) or all NULL");
    return RGCN_EINVAL;
  }
  if (Q == 0) return RGCN_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(rank_query_kernel, dim3((unsigned)((Q + 3) / 4)), dim3(WG), 0, st,
                     reinterpret_cast<const long long *>(batch), (int)Q, head, nodes, rel, sbias, pbias, obias, qvec,
                     qbias, d);
  const dim3 grid((unsigned)((n_nodes + 63) / 64), (unsigned)((Q + 63) / 64));
  const float *qb = sbias ? qbias : nullptr, *cb = sbias ? (head ? sbias : obias) : nullptr;
  if (d % 4 == 0)
    hipLaunchKernelGGL(score_all_kernel<true>, grid, dim3(WG), 0, st, qvec, qb, nodes, cb, scores, (int)Q,
                       (long long)n_nodes, d, head);
  else
    hipLaunchKernelGGL(score_all_kernel<false>, grid, dim3(WG), 0, st, qvec, qb, nodes, cb, scores, (int)Q,
                       (long long)n_nodes, d, head);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_rank_filter_f32(float *scores, int64_t Q, int64_t n_nodes, const int32_t *filt_q,
                                    const int32_t *filt_n, int64_t F, void *stream) {
  if (Q < 0 || n_nodes <= 0 || F < 0 || (F && (!scores || !filt_q || !filt_n))) {
    rgcn_set_error("rank_filter: bad argument");
    return RGCN_EINVAL;
  }
  if (F == 0) return RGCN_OK;
  hipLaunchKernelGGL(rank_filter_kernel, dim3((unsigned)std::min<int64_t>((F + WG - 1) / WG, 1 << 16)), dim3(WG), 0,
                     (hipStream_t)stream, scores, (long long)n_nodes, filt_q, filt_n, (long long)F);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_rank_count_f32(const float *scores, const int64_t *batch, int64_t Q, int32_t head,
                                   int64_t n_nodes, int64_t *greater, int64_t *ties, void *stream) {
  if (Q < 0 || n_nodes <= 0 || Q > INT32_MAX || (Q && (!scores || !batch || !greater || !ties))) {
    rgcn_set_error("rank_count: bad argument");
    return RGCN_EINVAL;
  }
  if (Q == 0) return RGCN_OK;
  hipLaunchKernelGGL(rank_count_kernel, dim3((unsigned)Q), dim3(WG), 0, (hipStream_t)stream, scores,
                     reinterpret_cast<const long long *>(batch), head, (long long)n_nodes,
                     reinterpret_cast<long long *>(greater), reinterpret_cast<long long *>(ties));
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}
