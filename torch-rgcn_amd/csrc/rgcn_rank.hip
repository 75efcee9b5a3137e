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

// scores tile per workgroup: (32 TA) queries x (32 TB) candidates, a (16 TA) x (16 TB) block of MFMA tiles per wave.
// Full K steps (all of k .. k+15 inside the row) run unmasked from two ping-pong operand buffers: while the MFMAs of one
// step execute, the loads of the next are in flight and nothing but pointer bumps competes for the VALU.
template <bool VEC, int TA, int TB>
__global__ __launch_bounds__(WG) void score_all_kernel(
    const float *__restrict__ qvec, const float *__restrict__ qb, const float *__restrict__ nodes,
    const float *__restrict__ cbias, float *__restrict__ scores, int Q, long long N, int d, int head, int q_blocks, int ablate_arg) {
#ifdef RGCN_ABLATIONS
  const int ablate = ablate_arg;      // timing experiments (wrong results): ablation build only
#else
  constexpr int ablate = 0;
  (void)ablate_arg;
#endif
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = lane & 15, kq = lane >> 4;
  // query block fastest: workgroups that are resident together share one slab of the entity table (L2), and the whole
  // query matrix (Q x d) stays cached
  const int q0 = (blockIdx.x % q_blocks) * (32 * TA) + (wave >> 1) * (16 * TA);
  const long long c0 = (long long)(blockIdx.x / q_blocks) * (32 * TB) + (wave & 1) * (16 * TB);
  const float *ap[TA], *bp[TB];
#pragma unroll
  for (int a = 0; a < TA; ++a) ap[a] = qvec + (size_t)min(q0 + 16 * a + i, Q - 1) * d;
#pragma unroll
  for (int b = 0; b < TB; ++b) bp[b] = nodes + (size_t)min(c0 + 16 * b + i, N - 1) * d;
  f32x4 acc[TA][TB];
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 a0[TA], b0[TB], a1[TA], b1[TB];
  if (ablate & 4) {
    for (int a = 0; a < TA; ++a) a1[a] = f32x4{1.f, 2.f, 3.f, 4.f};
    for (int b = 0; b < TB; ++b) b1[b] = f32x4{1.f, 2.f, 3.f, 4.f};
  }
  auto fetch = [&](f32x4 (&av)[TA], f32x4 (&bv)[TB], int k) {
#pragma unroll
    for (int a = 0; a < TA; ++a) av[a] = load_k4<VEC>(ap[a], k, d);
#pragma unroll
    for (int b = 0; b < TB; ++b) bv[b] = load_k4<VEC>(bp[b], k, d);
  };
  auto multiply = [&](const f32x4 (&av)[TA], const f32x4 (&bv)[TB]) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a][c], bv[b][c], acc[a][b], 0, 0, 0);
  };
  const int full = d / 16, k0 = 4 * kq;
  if (full > 0) {
    fetch(a0, b0, k0);
    int t = 0;
    for (; t + 2 <= full; t += 2) {
      if (!(ablate & 4)) fetch(a1, b1, (ablate & 1) ? k0 : 16 * (t + 1) + k0);
      __builtin_amdgcn_sched_barrier(0);   // loads stay above the MFMAs (hipcc would sink them to their first use)
      multiply(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      if (!(ablate & 4)) fetch(a0, b0, (ablate & 1) ? k0 : 16 * min(t + 2, full - 1) + k0);
      __builtin_amdgcn_sched_barrier(0);
      multiply(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (t < full) multiply(a0, b0);        // odd number of full steps: a0/b0 hold the last one
  }
  if (d % 16) {                            // tail: k >= d zeroed
    const int k = 16 * full + k0;
    fetch(a0, b0, k);
#pragma unroll
    for (int a = 0; a < TA; ++a) a0[a] = mask_k4<VEC>(a0[a], k, d);
#pragma unroll
    for (int b = 0; b < TB; ++b) b0[b] = mask_k4<VEC>(b0[b], k, d);
    multiply(a0, b0);
  }
  // D: lane 16*kq + i holds query rows 4kq..4kq+3 of the tile, candidate column i
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qrow = q0 + 16 * a + 4 * kq + r;
      if (qrow >= Q) continue;
      float s1 = 0.f, s2 = 0.f;
      if (qb) { s1 = qb[2 * qrow]; s2 = qb[2 * qrow + 1]; }
#pragma unroll
      for (int b = 0; b < TB; ++b) {
        const long long col = c0 + 16 * b + i;
        if (col >= N) continue;
        float sc = acc[a][b][r];
        if (qb) {   // layers.py:96: scores + (sbias[s] + pbias[p] + obias[o]), same association
          const float cb = cbias[col];
          sc += head ? ((cb + s1) + s2) : ((s2 + s1) + cb);
        }
        if (!(ablate & 2)) scores[(size_t)qrow * N + col] = sc;
      }
    }
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
  #ifdef RGCN_ABLATIONS
  const int ablate = rgcn_option_value(RGCN_OPT_RANK_ABLATE);   // measurement only (ablation build)
#else
  constexpr int ablate = 0;
#endif
  const int tile_env = rgcn_option_value(RGCN_OPT_RANK_TILE);   // 1 = LDS-staged (default); TA*10 + TB = register-only variant
  const int tile = (tile_env == 22 || tile_env == 24 || tile_env == 42 || tile_env == 44) ? tile_env : 1;
  if (tile == 1) {
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
  const int TAv = tile / 10, TBv = tile % 10;
  const int q_blocks = (int)((Q + 32 * TAv - 1) / (32 * TAv));
  const int64_t n_wg = ((n_nodes + 32 * TBv - 1) / (32 * TBv)) * q_blocks;
  if (n_wg > INT32_MAX) { rgcn_set_error("distmult_score_all: %lld x %lld scores in one call is too many; split the batch", (long long)Q, (long long)n_nodes); return RGCN_EUNSUPPORTED; }
  const dim3 grid((unsigned)n_wg);
  const float *qb = sbias ? qbias : nullptr, *cb = sbias ? (head ? sbias : obias) : nullptr;
#define RGCN_SCORE(VECV, TAC, TBC)                                                                                   \
  hipLaunchKernelGGL((score_all_kernel<VECV, TAC, TBC>), grid, dim3(WG), 0, st, qvec, qb, nodes, cb, scores, (int)Q, \
                     (long long)n_nodes, d, head, q_blocks, ablate)
  if (d % 4 == 0) {
    if (tile == 22) RGCN_SCORE(true, 2, 2); else if (tile == 42) RGCN_SCORE(true, 4, 2); else if (tile == 24) RGCN_SCORE(true, 2, 4); else RGCN_SCORE(true, 4, 4);
  } else {
    if (tile == 22) RGCN_SCORE(false, 2, 2); else if (tile == 42) RGCN_SCORE(false, 4, 2); else if (tile == 24) RGCN_SCORE(false, 2, 4); else RGCN_SCORE(false, 4, 4);
  }
#undef RGCN_SCORE
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
