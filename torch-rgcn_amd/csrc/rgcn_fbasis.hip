// Featureless layer with basis decomposition, source-major (the NodeClassifier's first layer
// on MUTAG / BGS / AM; reference layers.py:241-242 materialises weights = einsum('rb,bio->rio') -- R x N x d_out
// floats, 17.8 GB on AM -- and :286-288 multiplies the stacked adjacency with it).
//
//   out[s,:] = sum_{e=(s,r,o)} val_e * sum_b comps[r,b] * bases[b,o,:]
//
// Every message needs the B x d block of its SOURCE node o.  Walking the messages by destination (or by relation)
// re-reads that block once per message (B random 4d-byte rows each: 545 M fabric requests on AM).  Here the messages
// are walked by source: one wave per source node loads the node's block ONCE into registers from a node-major copy
// of the parameter ([N, B, d]: the block is 4Bd contiguous bytes; reading the parameter's own [B, N, d] layout in
// place was measured 4x slower -- B partially used 128-byte lines per node, evicted before the neighbours use them)
// and produces
//   forward : y_e = val_e * comps[r_e,:] . block            -> Y[e,:]   (e in source-major order, sequential write)
//   backward: dblock += val_e * comps[r_e,:]^T (x) g[s_e,:]  -> dbases[:,o,:] written once per node (no atomics)
//             t_e[b] = val_e * <block[b,:], g[s_e,:]>        -> T[e,:]   (for dcomps)
// and a gather-segment-sum (rows of Y by destination, rows of T by relation, through a permutation) finishes the job.
// Long rows (hub nodes, whole relations) are cut into work units that merge with fp32 atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

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

constexpr int WG = 256, WAVES = WG / 64;
constexpr int U_SHARED = RGCN_U_SHARED, U_FIRST = RGCN_U_FIRST;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float lane_bcast(float v, int src_lane) {   // src_lane wave-uniform
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}

// ---------------------------------------------------------------- forward, pass 1
// lane = (bg, i): feature i = lane % dp, basis group bg = lane / dp; the lane keeps block[bg*NREG + k][i], k < NREG, so
// that its NREG coefficients are contiguous: the coefficient table sits in LDS as [R][Bp = NREG * ngrp] (zero padded)
// and a lane reads its coefficients of one message with NREG/4 ds_read_b128.  1024-thread persistent workgroups: the
// table is staged once per workgroup and 2 workgroups (32 waves) fit a CU next to a 51 KB table.
constexpr int FWD_WG = 1024;

template <int NREG, bool TAB_LDS>
__global__ __launch_bounds__(FWD_WG) void fbasis_fwd_kernel(
    const float *__restrict__ table, const float *__restrict__ comps, float *__restrict__ Y,
    const int *__restrict__ e_rel, const float *__restrict__ e_val, const int4 *__restrict__ units, int n_units,
    int R, int B, int d, int dp) {
  extern __shared__ __attribute__((aligned(16))) float ctab[];
  const int ngrp = 64 / dp, Bp = NREG * ngrp;
  if (TAB_LDS) {
    for (int j = threadIdx.x; j < R * Bp; j += FWD_WG) {
      const int r = j / Bp, b = j % Bp;
      ctab[j] = b < B ? comps[(size_t)r * B + b] : 0.f;
    }
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane % dp, bg = lane / dp;
  for (int u = blockIdx.x * (FWD_WG / 64) + wave; u < n_units; u += gridDim.x * (FWD_WG / 64)) {
    const int4 unit = units[u];
    const float *blk_p = table + (size_t)unit.x * B * d;
    float blk[NREG];
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
      const int b = bg * NREG + k;
      blk[k] = (b < B && i < d) ? blk_p[b * d + i] : 0.f;          // zero where the lane has no element
    }
    for (int e0 = unit.y; e0 < unit.z; e0 += 64) {
      const int n = min(64, unit.z - e0);
      const int my_r = lane < n ? e_rel[e0 + lane] : 0;
      const float my_v = lane < n ? e_val[e0 + lane] : 0.f;
      for (int j = 0; j < n; ++j) {
        const int r = __builtin_amdgcn_readlane(my_r, j);
        const float v = lane_bcast(my_v, j);
        float t = 0.f;
        if (TAB_LDS) {
          const f32x4 *c4 = reinterpret_cast<const f32x4 *>(ctab + (size_t)r * Bp + bg * NREG);
#pragma unroll
          for (int k4 = 0; k4 < NREG / 4; ++k4) {
            const f32x4 c = c4[k4];
            t += c[0] * blk[4 * k4] + c[1] * blk[4 * k4 + 1] + c[2] * blk[4 * k4 + 2] + c[3] * blk[4 * k4 + 3];
          }
        } else {
          const float *c = comps + (size_t)r * B;
#pragma unroll
          for (int k = 0; k < NREG; ++k) {
            const int b = bg * NREG + k;
            if (b < B) t += c[b] * blk[k];
          }
        }
        for (int off = dp; off < 64; off <<= 1) t += __shfl_xor(t, off, 64);
        if (bg == 0 && i < d) Y[(size_t)(e0 + j) * d + i] = v * t;
      }
    }
  }
}

// ---------------------------------------------------------------- backward, pass 1
// lane = basis b (B <= 64); registers over the d <= DP features: block[b][:] and the block's gradient.
template <int DP>
__global__ __launch_bounds__(WG) void fbasis_bwd_kernel(
    const float *__restrict__ bases, const float *__restrict__ comps, const float *__restrict__ G,
    float *__restrict__ dbases, float *__restrict__ T, const int *__restrict__ e_dst, const int *__restrict__ e_rel,
    const float *__restrict__ e_val, const int4 *__restrict__ units, int n_units, long long N, int B, int d) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int u = blockIdx.x * WAVES + wave;
  if (u >= n_units) return;
  const int4 unit = units[u];
  const long long o = unit.x;
  const bool has_b = lane < B;
  float blk[DP], dblk[DP];
#pragma unroll
  for (int i = 0; i < DP; ++i) {
    blk[i] = (has_b && i < d && T) ? bases[((size_t)o * B + lane) * d + i] : 0.f;
    dblk[i] = 0.f;
  }
  constexpr int MB = 4;                                 // messages whose row loads fly together (8 measured slower)
  for (int e0 = unit.y; e0 < unit.z; e0 += 64) {
    const int n = min(64, unit.z - e0);
    const int my_s = lane < n ? e_dst[e0 + lane] : 0, my_r = lane < n ? e_rel[e0 + lane] : 0;
    const float my_v = lane < n ? e_val[e0 + lane] : 0.f;
    for (int j0 = 0; j0 < n; j0 += MB) {
      float grow[MB], cv[MB], vv[MB];
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const int j = min(j0 + m, n - 1);
        const int s = __builtin_amdgcn_readlane(my_s, j), r = __builtin_amdgcn_readlane(my_r, j);
        vv[m] = (j0 + m < n) ? lane_bcast(my_v, j) : 0.f;
        grow[m] = (lane < d) ? G[(size_t)s * d + lane] : 0.f;          // the upstream gradient row, one float per lane
        cv[m] = (has_b) ? comps[(size_t)r * B + lane] * vv[m] : 0.f;
      }
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < DP; ++i) {
          const float gi = lane_bcast(grow[m], i);                    // lanes >= d hold 0
          dblk[i] += cv[m] * gi;
          t += blk[i] * gi;
        }
        if (T && has_b && j0 + m < n) T[(size_t)(e0 + j0 + m) * B + lane] = vv[m] * t;
      }
    }
  }
  if (dbases && has_b) {
#pragma unroll
    for (int i = 0; i < DP; ++i)
      if (i < d) {
        float *p = dbases + ((size_t)o * B + lane) * d + i;
        if (unit.w & U_SHARED) atomicAdd(p, dblk[i]); else *p = dblk[i];
      }
  }
}

// ---------------------------------------------------------------- pass 2: out[row,:] = (bias) + sum_j Y[perm[j],:]
// lane = (g, i): column i = lane % wp, message group g = lane / wp.
__global__ __launch_bounds__(WG) void gather_rows_sum_kernel(
    const float *__restrict__ Y, const int *__restrict__ perm, const int4 *__restrict__ units, int n_units,
    const float *__restrict__ bias, float *__restrict__ out, int w, int wp) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int u = blockIdx.x * WAVES + wave;
  if (u >= n_units) return;
  const int4 unit = units[u];
  const int i = lane % wp, g = lane / wp, ngrp = 64 / wp;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int j = unit.y + g;
  for (; j + 3 * ngrp < unit.z; j += 4 * ngrp) {        // four independent row reads in flight
    const int p0 = perm ? perm[j] : j, p1 = perm ? perm[j + ngrp] : j + ngrp;
    const int p2 = perm ? perm[j + 2 * ngrp] : j + 2 * ngrp, p3 = perm ? perm[j + 3 * ngrp] : j + 3 * ngrp;
    if (i < w) {
      a0 += Y[(size_t)p0 * w + i];
      a1 += Y[(size_t)p1 * w + i];
      a2 += Y[(size_t)p2 * w + i];
      a3 += Y[(size_t)p3 * w + i];
    }
  }
  for (; j < unit.z; j += ngrp) {
    const int p0 = perm ? perm[j] : j;
    if (i < w) a0 += Y[(size_t)p0 * w + i];
  }
  float a = (a0 + a1) + (a2 + a3);
  for (int off = wp; off < 64; off <<= 1) a += __shfl_xor(a, off, 64);
  if (g == 0 && i < w) {
    float *p = out + (size_t)unit.x * w + i;
    if (unit.w & U_SHARED) {
      if ((unit.w & U_FIRST) && bias) a += bias[i];
      atomicAdd(p, a);
    } else {
      *p = bias ? a + bias[i] : a;
    }
  }
}

inline int pow2_at_least(int v, int lo) {
  int p = lo;
  while (p < v && p < 64) p <<= 1;
  return p;
}

}  // namespace

extern "C" int rgcn_fbasis_fwd_f32(const float *bases, const float *comps, float *Y, const int32_t *e_rel,
                                   const float *e_val, const int32_t *units, int64_t n_units, int64_t n_nodes,
                                   int32_t R, int32_t B, int32_t d, void *stream) {
  if (n_units < 0 || n_nodes <= 0 || R <= 0 || B <= 0 || d <= 0 || (n_units && (!bases || !comps || !Y || !units))) {
    rgcn_set_error("fbasis_fwd: bad argument");
    return RGCN_EINVAL;
  }
  if (d > 64) { rgcn_set_error("fbasis_fwd: d_out > 64 unsupported"); return RGCN_EUNSUPPORTED; }
  const int dp = pow2_at_least(d, 4), ngrp = 64 / dp, nreg = 4 * ((B + 4 * ngrp - 1) / (4 * ngrp));
  if (nreg > 16) { rgcn_set_error("fbasis_fwd: %d bases at width %d exceed the register block", B, d); return RGCN_EUNSUPPORTED; }
  if (n_units == 0) return RGCN_OK;
  const size_t tab_bytes = (size_t)R * nreg * ngrp * sizeof(float);
  const bool in_lds = tab_bytes <= 52 * 1024;
  const int waves = FWD_WG / 64;
  const dim3 grid((unsigned)std::min<int64_t>((n_units + waves - 1) / waves, 256 * 2));
  const int4 *un = reinterpret_cast<const int4 *>(units);
#define RGCN_FB_FWD(NR)                                                                                                  \
  {                                                                                                                      \
    if (in_lds)                                                                                                          \
      hipLaunchKernelGGL((fbasis_fwd_kernel<NR, true>), grid, dim3(FWD_WG), tab_bytes, (hipStream_t)stream, bases, comps, Y, \
                         e_rel, e_val, un, (int)n_units, R, B, d, dp);                                                   \
    else                                                                                                                 \
      hipLaunchKernelGGL((fbasis_fwd_kernel<NR, false>), grid, dim3(FWD_WG), 0, (hipStream_t)stream, bases, comps, Y, e_rel,  \
                         e_val, un, (int)n_units, R, B, d, dp);                                                          \
  }
  if (nreg <= 4) RGCN_FB_FWD(4) else if (nreg <= 8) RGCN_FB_FWD(8) else if (nreg <= 12) RGCN_FB_FWD(12) else RGCN_FB_FWD(16)
#undef RGCN_FB_FWD
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_fbasis_bwd_f32(const float *bases, const float *comps, const float *G, float *dbases, float *T,
                                   const int32_t *e_dst, const int32_t *e_rel, const float *e_val,
                                   const int32_t *units, int64_t n_units, int64_t n_split, int64_t n_nodes, int32_t R,
                                   int32_t B, int32_t d, void *stream) {
  if (n_units < 0 || n_nodes <= 0 || R <= 0 || B <= 0 || d <= 0 || (!dbases && !T) ||
      (n_units && (!bases || !comps || !G || !units))) {
    rgcn_set_error("fbasis_bwd: bad argument");
    return RGCN_EINVAL;
  }
  if (B > 64 || d > 16) { rgcn_set_error("fbasis_bwd: needs B <= 64 and d_out <= 16"); return RGCN_EUNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  if (dbases && n_split) HIP_TRY(hipMemsetAsync(dbases, 0, (size_t)B * n_nodes * d * sizeof(float), st));
  if (n_units == 0) return RGCN_OK;
  const dim3 grid((unsigned)((n_units + WAVES - 1) / WAVES));
  const int4 *un = reinterpret_cast<const int4 *>(units);
#define RGCN_FB_BWD(DPC)                                                                                          \
  hipLaunchKernelGGL(fbasis_bwd_kernel<DPC>, grid, dim3(WG), 0, st, bases, comps, G, dbases, T, e_dst, e_rel, e_val, un, \
                     (int)n_units, (long long)n_nodes, B, d)
  if (d <= 4) RGCN_FB_BWD(4); else if (d <= 8) RGCN_FB_BWD(8); else if (d <= 12) RGCN_FB_BWD(12); else RGCN_FB_BWD(16);
#undef RGCN_FB_BWD
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_gather_rows_sum_f32(const float *Y, const int32_t *perm, const int32_t *units, int64_t n_units,
                                        int64_t n_split, const float *bias, float *out, int64_t n_rows, int32_t w,
                                        void *stream) {
  if (n_units < 0 || n_rows < 0 || w <= 0 || (n_units && (!Y || !units || !out))) {
    rgcn_set_error("gather_rows_sum: bad argument");
    return RGCN_EINVAL;
  }
  if (w > 64) { rgcn_set_error("gather_rows_sum: width > 64 unsupported"); return RGCN_EUNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  if (n_split) HIP_TRY(hipMemsetAsync(out, 0, (size_t)n_rows * w * sizeof(float), st));
  if (n_units == 0) return RGCN_OK;
  hipLaunchKernelGGL(gather_rows_sum_kernel, dim3((unsigned)((n_units + WAVES - 1) / WAVES)), dim3(WG), 0, st, Y, perm,
                     reinterpret_cast<const int4 *>(units), (int)n_units, bias, out, w, pow2_at_least(w, 4));
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}
