// Ranking evaluator of the link-prediction experiments: every entity scored as the
// head or the tail of each test triple, known true triples masked, rank of the target counted with ties.
// Reference lines taken over: utils/misc.py:60-110 (evaluate: toscore expansion :78-83, model call :85,
// filter_scores :40-58, raw_ranks / num_ties :93-99) and torch_rgcn/layers.py:87-98 (DistMult.forward on the
// expanded [bn, N, 3] index tensor, which is what the reference scores -- three gathers of bn x N x d floats).
//
// Here the [bn, N, 3] index tensor is never built.  For a batch of Q queries
//     scores[q, n] = sum_k (nodes[fixed_q, k] * rel[p_q, k]) * nodes[n, k]  (+ biases)
// is an NT product of the Q x d query vectors with the N x d entity table: fp32 MFMA (v_mfma_f32_16x16x4_f32,
// exact fp32 products and accumulation).  Default kernel: 128 x 128 scores per workgroup, operand slabs staged through
// double-buffered LDS (score_all_lds_kernel); the register-only variant (operands straight from L2, each lane's float4
// feeding four MFMAs of each tile that shares it) is kept selectable for comparison (RGCN_RANK_TILE=44).
// The K index is permuted -- lane group kq carries k = 16t + 4kq + c at MFMA c of step t -- which is legal
// because both operands use the same permutation and the sum over k does not care.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "rgcn_hip.h"
#include "rgcn_options.h"

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

// Same product with the operands staged through LDS: a workgroup owns 128 queries x 128 candidates (a 64 x 64 block
// per wave), every K step's two 128 x 16 operand slabs are fetched from L2 ONCE per workgroup (the register-only kernel
// fetches each slab twice, and 16 rows x 64 B per instruction is a poor shape for the L1), written to a double-buffered
// LDS tile (row stride 20 floats: the ds_read_b128 of 16 rows x 4 column groups is conflict-free) and read back as MFMA
// operands.  The loads of step t+1 are issued before the 64 MFMAs of step t and land in LDS after them.
constexpr int LDS_LD = 20;
template <bool VEC>
__global__ __launch_bounds__(WG) void score_all_lds_kernel(
    const float *__restrict__ qvec, const float *__restrict__ qb, const float *__restrict__ nodes,
    const float *__restrict__ cbias, float *__restrict__ scores, int Q, long long N, int d, int head, int q_blocks) {
  __shared__ __attribute__((aligned(16))) float sA[2][128 * LDS_LD], sB[2][128 * LDS_LD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 15, kq = lane >> 4;
  const int qt = (blockIdx.x % q_blocks) * 128;
  const long long ct = (long long)(blockIdx.x / q_blocks) * 128;
  const int q0 = qt + (wave >> 1) * 64;
  const long long c0 = ct + (wave & 1) * 64;
  // staging: thread -> rows (tid >> 2) and (tid >> 2) + 64 of both slabs, 16-byte column group tid & 3
  const int sr = tid >> 2, sc = 4 * (tid & 3);
  const float *ga[2], *gb[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    ga[h] = qvec + (size_t)min(qt + sr + 64 * h, Q - 1) * d;
    gb[h] = nodes + (size_t)min(ct + sr + 64 * h, N - 1) * d;
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int steps = (d + 15) / 16;
  f32x4 sa[2], sb[2];
  auto fetch = [&](int t) {
    const int k = 16 * t + sc;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      sa[h] = load_k4<VEC>(ga[h], k, d);
      sb[h] = load_k4<VEC>(gb[h], k, d);
    }
  };
  auto stash = [&](int buf, int t) {         // zero the K tail here, so the compute loop never masks
    const int k = 16 * t + sc;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *reinterpret_cast<f32x4 *>(&sA[buf][(sr + 64 * h) * LDS_LD + sc]) = mask_k4<VEC>(sa[h], k, d);
      *reinterpret_cast<f32x4 *>(&sB[buf][(sr + 64 * h) * LDS_LD + sc]) = mask_k4<VEC>(sb[h], k, d);
    }
  };
  fetch(0);
  stash(0, 0);
  __syncthreads();
  for (int t = 0; t < steps; ++t) {
    const int cur = t & 1;
    if (t + 1 < steps) fetch(t + 1);
    f32x4 av[4], bv[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      av[a] = *reinterpret_cast<const f32x4 *>(&sA[cur][((wave >> 1) * 64 + 16 * a + i) * LDS_LD + 4 * kq]);
      bv[a] = *reinterpret_cast<const f32x4 *>(&sB[cur][((wave & 1) * 64 + 16 * a + i) * LDS_LD + 4 * kq]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a][c], bv[b][c], acc[a][b], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < steps) stash(cur ^ 1, t + 1);
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qrow = q0 + 16 * a + 4 * kq + r;
      if (qrow >= Q) continue;
      float s1 = 0.f, s2 = 0.f;
      if (qb) { s1 = qb[2 * qrow]; s2 = qb[2 * qrow + 1]; }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const long long col = c0 + 16 * b + i;
        if (col >= N) continue;
        float sc_ = acc[a][b][r];
        if (qb) {
          const float cb = cbias[col];
          sc_ += head ? ((cb + s1) + s2) : ((s2 + s1) + cb);
        }
        scores[(size_t)qrow * N + col] = sc_;
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
  {     // (round 5: the register-only tile variants behind RGCN_RANK_TILE -- measured slower than this LDS-staged kernel in round 1 -- are gone)
    const int qbl = (int)((Q + 127) / 128);
    const int64_t nwg = ((n_nodes + 127) / 128) * qbl;
    if (nwg > INT32_MAX) { rgcn_set_error("distmult_score_all: too many scores in one call; split the batch"); return RGCN_EUNSUPPORTED; }
    const float *qb1 = sbias ? qbias : nullptr, *cb1 = sbias ? (head ? sbias : obias) : nullptr;
    if (d % 4 == 0)
      hipLaunchKernelGGL(score_all_lds_kernel<true>, dim3((unsigned)nwg), dim3(WG), 0, st, qvec, qb1, nodes, cb1, scores,
                         (int)Q, (long long)n_nodes, d, head, qbl);
    else
      hipLaunchKernelGGL(score_all_lds_kernel<false>, dim3((unsigned)nwg), dim3(WG), 0, st, qvec, qb1, nodes, cb1, scores,
                         (int)Q, (long long)n_nodes, d, head, qbl);
    HIP_TRY(hipGetLastError());
    return RGCN_OK;
  }
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
