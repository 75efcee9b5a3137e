// Basis decomposition (layers.py:241-242, :468-469) without the R x ... weight tensors.
// Part 1 -- featureless layer with basis decomposition, source-major (the NodeClassifier's first layer
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
#include "rgcn_zero.h"
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
    int R, int B, int d, int dp, long long sn, long long sb) {      // element (node, b, i) of the table at node * sn + b * sb + i
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
    const float *blk_p = table + (size_t)unit.x * sn;
    float blk[NREG];
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
      const int b = bg * NREG + k;
      blk[k] = (b < B && i < d) ? blk_p[(size_t)b * sb + i] : 0.f;          // zero where the lane has no element
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

// ---------------------------------------------------------------- backward with dcomps summed on the chip (round 4)
// The same walk as fbasis_bwd_kernel below, persistent 1024-thread workgroups striding over the units; t_e[b] = val <block[b], G[s_e]> is
// not written to a [M][B] scratch (AM as shipped: 2.2 GB out and 2.2 GB back in through a relation-major permutation) but added to an LDS
// table dcomps[R][B] of DOUBLES (ds_add_f64: lane = basis -> the B lanes of a message add to B consecutive doubles) that every workgroup
// flushes once.
constexpr int DC_WG = 1024;
template <int DP>
__global__ __launch_bounds__(DC_WG) void fbasis_bwd_dc_kernel(
    const float *__restrict__ bases, const float *__restrict__ comps, const float *__restrict__ G,
    float *__restrict__ dbases, float *__restrict__ dC, const int *__restrict__ e_dst, const int *__restrict__ e_rel,
    const float *__restrict__ e_val, const int4 *__restrict__ units, int n_units, int R, int B, int d, long long sn,
    long long sb) {
  extern __shared__ __attribute__((aligned(16))) double dcl[];          // [R][B]
  for (int i = threadIdx.x; i < R * B; i += DC_WG) dcl[i] = 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool has_b = lane < B;
  const bool vec = DP % 4 == 0 && d == DP;
  for (int u = blockIdx.x * (DC_WG / 64) + wave; u < n_units; u += gridDim.x * (DC_WG / 64)) {
    const int4 unit = units[u];
    const long long o = unit.x;
    float blk[DP], dblk[DP];
#pragma unroll
    for (int i = 0; i < DP; ++i) {
      blk[i] = 0.f;
      dblk[i] = 0.f;
    }
    if (has_b) {
      const float *bp = bases + (size_t)o * sn + (size_t)lane * sb;
      if (vec) {
#pragma unroll
        for (int i = 0; i < DP / 4; ++i) {
          const f32x4 t4 = reinterpret_cast<const f32x4 *>(bp)[i];
          blk[4 * i] = t4[0]; blk[4 * i + 1] = t4[1]; blk[4 * i + 2] = t4[2]; blk[4 * i + 3] = t4[3];
        }
      } else {
#pragma unroll
        for (int i = 0; i < DP; ++i)
          if (i < d) blk[i] = bp[i];
      }
    }
    constexpr int MB = 4;
    for (int e0 = unit.y; e0 < unit.z; e0 += 64) {
      const int n = min(64, unit.z - e0);
      const int my_s = lane < n ? e_dst[e0 + lane] : 0, my_r = lane < n ? e_rel[e0 + lane] : 0;
      const float my_v = lane < n ? e_val[e0 + lane] : 0.f;
      for (int j0 = 0; j0 < n; j0 += MB) {
        float grow[MB], cv[MB], vv[MB];
        int rr[MB];
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          const int j = min(j0 + m, n - 1);
          const int s = __builtin_amdgcn_readlane(my_s, j);
          rr[m] = __builtin_amdgcn_readlane(my_r, j);
          vv[m] = (j0 + m < n) ? lane_bcast(my_v, j) : 0.f;
          grow[m] = (lane < d) ? G[(size_t)s * d + lane] : 0.f;
          cv[m] = (has_b) ? comps[(size_t)rr[m] * B + lane] * vv[m] : 0.f;
        }
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          float t = 0.f;
#pragma unroll
          for (int i = 0; i < DP; ++i) {
            const float gi = lane_bcast(grow[m], i);
            dblk[i] += cv[m] * gi;
            t += blk[i] * gi;
          }
          if (has_b && j0 + m < n)
            __hip_atomic_fetch_add(dcl + rr[m] * B + lane, (double)(vv[m] * t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
    if (dbases && has_b) {
      float *dp_ = dbases + (size_t)o * sn + (size_t)lane * sb;
      if (vec && !(unit.w & U_SHARED)) {
#pragma unroll
        for (int i = 0; i < DP / 4; ++i)
          reinterpret_cast<f32x4 *>(dp_)[i] = f32x4{dblk[4 * i], dblk[4 * i + 1], dblk[4 * i + 2], dblk[4 * i + 3]};
      } else {
#pragma unroll
        for (int i = 0; i < DP; ++i)
          if (i < d) {
            if (unit.w & U_SHARED) atomicAdd(dp_ + i, dblk[i]); else dp_[i] = dblk[i];
          }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < R * B; i += DC_WG) {
    const float t = (float)dcl[i];
    if (t != 0.f) atomicAdd(dC + i, t);
  }
}

// ---------------------------------------------------------------- backward, pass 1
// lane = basis b (B <= 64); registers over the d <= DP features: block[b][:] and the block's gradient.
template <int DP>
__global__ __launch_bounds__(WG) void fbasis_bwd_kernel(
    const float *__restrict__ bases, const float *__restrict__ comps, const float *__restrict__ G,
    float *__restrict__ dbases, float *__restrict__ T, const int *__restrict__ e_dst, const int *__restrict__ e_rel,
    const float *__restrict__ e_val, const int4 *__restrict__ units, int n_units, long long N, int B, int d, long long sn,
    long long sb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int u = blockIdx.x * WAVES + wave;
  if (u >= n_units) return;
  const int4 unit = units[u];
  const long long o = unit.x;
  const bool has_b = lane < B;
  float blk[DP], dblk[DP];
  const bool vec = DP % 4 == 0 && d == DP;              // a lane's d floats are one aligned run: 16-byte loads / stores
#pragma unroll
  for (int i = 0; i < DP; ++i) {
    blk[i] = 0.f;
    dblk[i] = 0.f;
  }
  if (has_b && T) {
    const float *bp = bases + (size_t)o * sn + (size_t)lane * sb;
    if (vec) {
#pragma unroll
      for (int i = 0; i < DP / 4; ++i) {
        const f32x4 t4 = reinterpret_cast<const f32x4 *>(bp)[i];
        blk[4 * i] = t4[0]; blk[4 * i + 1] = t4[1]; blk[4 * i + 2] = t4[2]; blk[4 * i + 3] = t4[3];
      }
    } else {
#pragma unroll
      for (int i = 0; i < DP; ++i)
        if (i < d) blk[i] = bp[i];
    }
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
    float *dp_ = dbases + (size_t)o * sn + (size_t)lane * sb;
    if (vec && !(unit.w & U_SHARED)) {
#pragma unroll
      for (int i = 0; i < DP / 4; ++i)
        reinterpret_cast<f32x4 *>(dp_)[i] = f32x4{dblk[4 * i], dblk[4 * i + 1], dblk[4 * i + 2], dblk[4 * i + 3]};
    } else {
#pragma unroll
      for (int i = 0; i < DP; ++i)
        if (i < d) {
          if (unit.w & U_SHARED) atomicAdd(dp_ + i, dblk[i]); else dp_[i] = dblk[i];
        }
    }
  }
}

// ---------------------------------------------------------------- pass 2: out[row,:] = (bias) + sum_j Y[perm[j],:]
// lane = (g, i): column i = lane % wp, message group g = lane / wp.
__global__ __launch_bounds__(WG) void gather_rows_sum_kernel(
    const float *__restrict__ Y, const int *__restrict__ perm, const int4 *__restrict__ units, int n_units,
    const float *__restrict__ bias, float *__restrict__ out, int w, int wp, int relu_out) {
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
      a = bias ? a + bias[i] : a;
      *p = relu_out ? fmaxf(a, 0.f) : a;
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
                                   int32_t R, int32_t B, int32_t d, int32_t basis_major, void *stream) {
  const long long sn = basis_major ? d : (long long)B * d, sb = basis_major ? (long long)n_nodes * d : d;
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
                         e_rel, e_val, un, (int)n_units, R, B, d, dp, sn, sb);                                           \
    else                                                                                                                 \
      hipLaunchKernelGGL((fbasis_fwd_kernel<NR, false>), grid, dim3(FWD_WG), 0, (hipStream_t)stream, bases, comps, Y, e_rel,  \
                         e_val, un, (int)n_units, R, B, d, dp, sn, sb);                                                  \
  }
  if (nreg <= 4) RGCN_FB_FWD(4) else if (nreg <= 8) RGCN_FB_FWD(8) else if (nreg <= 12) RGCN_FB_FWD(12) else RGCN_FB_FWD(16)
#undef RGCN_FB_FWD
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_fbasis_bwd_f32(const float *bases, const float *comps, const float *G, float *dbases, float *T,
                                   const int32_t *e_dst, const int32_t *e_rel, const float *e_val,
                                   const int32_t *units, int64_t n_units, int64_t n_split, int64_t n_nodes, int32_t R,
                                   int32_t B, int32_t d, int32_t basis_major, void *stream) {
  const long long sn = basis_major ? d : (long long)B * d, sb = basis_major ? (long long)n_nodes * d : d;
  if (n_units < 0 || n_nodes <= 0 || R <= 0 || B <= 0 || d <= 0 || (!dbases && !T) ||
      (n_units && (!bases || !comps || !G || !units))) {
    rgcn_set_error("fbasis_bwd: bad argument");
    return RGCN_EINVAL;
  }
  if (B > 64 || d > 16) { rgcn_set_error("fbasis_bwd: needs B <= 64 and d_out <= 16"); return RGCN_EUNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  if (dbases && n_split) HIP_TRY(zero_async(dbases, (size_t)B * n_nodes * d * sizeof(float), st));
  if (n_units == 0) return RGCN_OK;
  const dim3 grid((unsigned)((n_units + WAVES - 1) / WAVES));
  const int4 *un = reinterpret_cast<const int4 *>(units);
#define RGCN_FB_BWD(DPC)                                                                                          \
  hipLaunchKernelGGL(fbasis_bwd_kernel<DPC>, grid, dim3(WG), 0, st, bases, comps, G, dbases, T, e_dst, e_rel, e_val, un, \
                     (int)n_units, (long long)n_nodes, B, d, sn, sb)
  if (d <= 4) RGCN_FB_BWD(4); else if (d <= 8) RGCN_FB_BWD(8); else if (d <= 12) RGCN_FB_BWD(12); else RGCN_FB_BWD(16);
#undef RGCN_FB_BWD
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_fbasis_bwd_dc_supported(int32_t R, int32_t B, int32_t d) {
  return R > 0 && B > 0 && B <= 64 && d > 0 && d <= 16 && (size_t)R * B * sizeof(double) <= 120 * 1024;
}

extern "C" int rgcn_fbasis_bwd_dc_f32(const float *bases, const float *comps, const float *G, float *dbases, float *dcomps,
                                      const int32_t *e_dst, const int32_t *e_rel, const float *e_val, const int32_t *units,
                                      int64_t n_units, int64_t n_split, int64_t n_nodes, int32_t R, int32_t B, int32_t d,
                                      int32_t basis_major, void *stream) {
  const long long sn = basis_major ? d : (long long)B * d, sb = basis_major ? (long long)n_nodes * d : d;
  if (n_units < 0 || n_nodes <= 0 || !dcomps || (n_units && (!bases || !comps || !G || !units))) { rgcn_set_error("fbasis_bwd_dc: bad argument"); return RGCN_EINVAL; }
  if (!rgcn_fbasis_bwd_dc_supported(R, B, d)) { rgcn_set_error("fbasis_bwd_dc: needs B <= 64, d_out <= 16 and R x B doubles within 120 KiB of LDS"); return RGCN_EUNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(zero_async(dcomps, (size_t)R * B * sizeof(float), st));
  if (dbases && n_split) HIP_TRY(zero_async(dbases, (size_t)B * n_nodes * d * sizeof(float), st));
  if (n_units == 0) return RGCN_OK;
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0, v = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
    n_cu = v > 0 ? v : 256;
  }
  const size_t lds = (size_t)R * B * sizeof(double);
  const int per_cu = lds <= 72 * 1024 ? 2 : 1;
  const dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>((n_units + DC_WG / 64 - 1) / (DC_WG / 64), (int64_t)n_cu * per_cu)));
  const int4 *un = reinterpret_cast<const int4 *>(units);
#define RGCN_FB_DC(DPC)                                                                                                            \
  {                                                                                                                                \
    static bool raised = false;                                                                                                    \
    if (lds > 64 * 1024 && !raised) {                                                                                              \
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fbasis_bwd_dc_kernel<DPC>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024)); \
      raised = true;                                                                                                               \
    }                                                                                                                              \
    hipLaunchKernelGGL(fbasis_bwd_dc_kernel<DPC>, grid, dim3(DC_WG), lds, st, bases, comps, G, dbases, dcomps, e_dst, e_rel, e_val, un, \
                       (int)n_units, R, B, d, sn, sb);                                                                             \
  }
  if (d <= 4) RGCN_FB_DC(4) else if (d <= 8) RGCN_FB_DC(8) else if (d <= 12) RGCN_FB_DC(12) else RGCN_FB_DC(16)
#undef RGCN_FB_DC
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_gather_rows_sum_f32(const float *Y, const int32_t *perm, const int32_t *units, int64_t n_units,
                                        int64_t n_split, const float *bias, float *out, int64_t n_rows, int32_t w,
                                        int32_t relu, void *stream) {
  if (n_units < 0 || n_rows < 0 || w <= 0 || (n_units && (!Y || !units || !out))) {
    rgcn_set_error("gather_rows_sum: bad argument");
    return RGCN_EINVAL;
  }
  if (relu && n_split) { rgcn_set_error("gather_rows_sum: relu in the epilogue needs rows that are not cut into shared pieces"); return RGCN_EUNSUPPORTED; }
  if (w > 64) { rgcn_set_error("gather_rows_sum: width > 64 unsupported"); return RGCN_EUNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  if (n_split) HIP_TRY(zero_async(out, (size_t)n_rows * w * sizeof(float), st));
  if (n_units == 0) return RGCN_OK;
  hipLaunchKernelGGL(gather_rows_sum_kernel, dim3((unsigned)((n_units + WAVES - 1) / WAVES)), dim3(WG), 0, st, Y, perm,
                     reinterpret_cast<const int4 *>(units), (int)n_units, bias, out, w, pow2_at_least(w, 4), relu ? 1 : 0);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

// ==================================================================== destination-major kernels
// (featured layers with basis decomposition at large width, and the fallback of the featureless layer outside the
//  source-major kernels' limits)
// ------------------------------------------------------------------ basis decomposition, aggregate-then-contract
namespace {

constexpr int TB = 256;
inline unsigned blocks_for(int64_t n, int per = TB) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + per - 1) / per, 1 << 20)); }

// One wave per destination row; the wave is split into 64/lpr groups of lpr = min(64, pow2 >= d) lanes and every
// group takes every (64/lpr)-th message of the row (narrow rows keep all lanes busy), reduced across groups at the end.
// NB_IN = 1: out[row][b][:] = sum_e comps[rel_e][b] * val_e * X[src_e][:]
// NB_IN = B: out[row][:]    = sum_e sum_b comps[rel_e][b] * val_e * X[src_e][b][:]
__device__ __forceinline__ float group_sum(float a, int lpr) {   // sum over the lane groups; result valid in group 0
  for (int off = lpr; off < 64; off <<= 1) a += __shfl_xor(a, off, 64);
  return a;
}

__global__ __launch_bounds__(TB) void basis_aggregate_kernel(
    const float *__restrict__ X, const float *__restrict__ comps, float *__restrict__ out,
    const int *__restrict__ rowptr, const int *__restrict__ p_src, const int *__restrict__ p_rel,
    const float *__restrict__ p_val, long long n_rows, int B, int d, int n_b_in, int lpr) {
  const int lane = threadIdx.x & 63;
  const int sub = lane / lpr, il = lane % lpr, ngrp = 64 / lpr;
  const long long wave0 = ((long long)blockIdx.x * TB + threadIdx.x) >> 6, nw = ((long long)gridDim.x * TB) >> 6;
  for (long long row = wave0; row < n_rows; row += nw) {
    const int e0 = rowptr[row], e1 = rowptr[row + 1];
    for (int i0 = 0; i0 < d; i0 += lpr) {
      const int i = i0 + il;
      constexpr int MB = 4;   // messages whose index / row / coefficient loads fly together (the row loop is latency bound)
      if (n_b_in == 1) {
        for (int b0 = 0; b0 < B; b0 += 4) {           // up to 4 bases per pass in registers
          float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
          for (int eb = e0 + sub; eb < e1; eb += ngrp * MB) {
            int src[MB], rel[MB];
            float val[MB];
#pragma unroll
            for (int m = 0; m < MB; ++m) {
              const int e = min(eb + m * ngrp, e1 - 1);
              src[m] = p_src[e];
              rel[m] = p_rel[e];
              val[m] = (eb + m * ngrp < e1) ? p_val[e] : 0.f;
            }
            float x[MB], c[MB][4];
#pragma unroll
            for (int m = 0; m < MB; ++m) {
              x[m] = i < d ? X[(size_t)src[m] * d + i] : 0.f;
              const float *cp = comps + (size_t)rel[m] * B + b0;
              c[m][0] = cp[0];
              c[m][1] = b0 + 1 < B ? cp[1] : 0.f;
              c[m][2] = b0 + 2 < B ? cp[2] : 0.f;
              c[m][3] = b0 + 3 < B ? cp[3] : 0.f;
            }
#pragma unroll
            for (int m = 0; m < MB; ++m) {
              const float v = val[m] * x[m];
              a0 += c[m][0] * v; a1 += c[m][1] * v; a2 += c[m][2] * v; a3 += c[m][3] * v;
            }
          }
          a0 = group_sum(a0, lpr); a1 = group_sum(a1, lpr); a2 = group_sum(a2, lpr); a3 = group_sum(a3, lpr);
          if (sub == 0 && i < d) {
            float *o = out + ((size_t)row * B + b0) * d + i;
            o[0] = a0;
            if (b0 + 1 < B) o[d] = a1;
            if (b0 + 2 < B) o[2 * (size_t)d] = a2;
            if (b0 + 3 < B) o[3 * (size_t)d] = a3;
          }
        }
      } else {
        float a = 0.f;
        for (int eb = e0 + sub; eb < e1; eb += ngrp * MB) {
          int src[MB], rel[MB];
          float val[MB];
#pragma unroll
          for (int m = 0; m < MB; ++m) {
            const int e = min(eb + m * ngrp, e1 - 1);
            src[m] = p_src[e];
            rel[m] = p_rel[e];
            val[m] = (eb + m * ngrp < e1) ? p_val[e] : 0.f;
          }
          float t[MB];
#pragma unroll
          for (int m = 0; m < MB; ++m) t[m] = 0.f;
          if (i < d)
            for (int b = 0; b < B; ++b) {
#pragma unroll
              for (int m = 0; m < MB; ++m)
                t[m] += comps[(size_t)rel[m] * B + b] * X[((size_t)src[m] * B + b) * d + i];
            }
#pragma unroll
          for (int m = 0; m < MB; ++m) a += val[m] * t[m];
        }
        a = group_sum(a, lpr);
        if (sub == 0 && i < d) out[(size_t)row * d + i] = a;
      }
    }
  }
}

// The same with 16-byte loads (d % 4 == 0): a lane owns 4 consecutive columns, lpm = pow2 >= d / 4 (<= 64) lanes per message --
// a 200-wide row is ONE pass of 50 lanes instead of four passes of 64 scalar loads, and the per-row set-up (index loads, loop,
// reduction) -- which is what a 2-message row costs -- is paid once (WN18-shaped graph: 58 -> ~30 us per launch).
// NB_IN = 1 takes up to 4 bases per pass.
template <int NB_IN_ONE>
__global__ __launch_bounds__(TB) void basis_aggregate_vec4_kernel(
    const float *__restrict__ X, const float *__restrict__ comps, float *__restrict__ out,
    const int *__restrict__ rowptr, const int *__restrict__ p_src, const int *__restrict__ p_rel,
    const float *__restrict__ p_val, long long n_rows, int B, int d, int lpm) {
  const int lane = threadIdx.x & 63;
  const int sub = lane / lpm, il = lane % lpm, ngrp = 64 / lpm;
  const long long wave0 = ((long long)blockIdx.x * TB + threadIdx.x) >> 6, nw = ((long long)gridDim.x * TB) >> 6;
  constexpr int MB = 2;      // messages in flight per lane group
  for (long long row = wave0; row < n_rows; row += nw) {
    const int e0 = rowptr[row], e1 = rowptr[row + 1];
    for (int f0 = 0; f0 < d; f0 += 4 * lpm) {
      const int f = f0 + 4 * il;
      const bool on = f < d;
      if (NB_IN_ONE) {
        for (int b0 = 0; b0 < B; b0 += 4) {
          f32x4 a[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
          for (int eb = e0 + sub; eb < e1; eb += ngrp * MB) {
#pragma unroll
            for (int m = 0; m < MB; ++m) {
              const int e = min(eb + m * ngrp, e1 - 1);
              const float v = (eb + m * ngrp < e1) ? p_val[e] : 0.f;
              const f32x4 x = on ? *reinterpret_cast<const f32x4 *>(X + (size_t)p_src[e] * d + f) : f32x4{0.f, 0.f, 0.f, 0.f};
              const float *cp = comps + (size_t)p_rel[e] * B + b0;
              a[0] += x * (cp[0] * v);
              if (b0 + 1 < B) a[1] += x * (cp[1] * v);
              if (b0 + 2 < B) a[2] += x * (cp[2] * v);
              if (b0 + 3 < B) a[3] += x * (cp[3] * v);
            }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) a[q][c] = group_sum(a[q][c], lpm);
          if (sub == 0 && on) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (b0 + q < B) *reinterpret_cast<f32x4 *>(out + ((size_t)row * B + b0 + q) * d + f) = a[q];
          }
        }
      } else {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        for (int eb = e0 + sub; eb < e1; eb += ngrp * MB) {
#pragma unroll
          for (int m = 0; m < MB; ++m) {
            const int e = min(eb + m * ngrp, e1 - 1);
            const float v = (eb + m * ngrp < e1) ? p_val[e] : 0.f;
            const float *cp = comps + (size_t)p_rel[e] * B;
            const float *xr = X + (size_t)p_src[e] * B * d + f;
            if (on)
              for (int b = 0; b < B; ++b) a += *reinterpret_cast<const f32x4 *>(xr + (size_t)b * d) * (cp[b] * v);
          }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c] = group_sum(a[c], lpm);
        if (sub == 0 && on) *reinterpret_cast<f32x4 *>(out + (size_t)row * d + f) = a;
      }
    }
  }
}

// The same for rows of ONE pass (d <= 256) and B <= 4 bases, with the ROWS software-pipelined over persistent waves.  A row is a chain of three
// dependent round trips (row pointers -> the messages' indices -> the rows and coefficients) and most rows fit one trip of the message
// loop (21 messages at S2, 2 on the WN18-shaped graph): nothing inside a row overlaps them.  While row k is summed the indices of row
// k + 1 are in flight, and its gathers are issued (behind the pointers of row k + 2) BEFORE row k's result is stored -- s_waitcnt vmcnt
// counts loads and stores in order: loads issued after the store would wait for the store to complete.  Every load is unconditional
// (entries past a row's end re-read its last entry with a zero value; lanes past the row's width read feature 0 and keep nothing).
// MODE 0: X [N, B, d] (one B x d block per node), out [n_rows, d];  MODE 1: X [N, d], out [n_rows, B, d].
template <int MODE, int NB>
__global__ __launch_bounds__(TB) void basis_aggregate_pipe_kernel(
    const float *__restrict__ X, const float *__restrict__ comps, float *__restrict__ out,
    const int *__restrict__ rowptr, const int *__restrict__ p_src, const int *__restrict__ p_rel,
    const float *__restrict__ p_val, long long n_rows, int B, int d, int lpm) {
  const int lane = threadIdx.x & 63;
  const int sub = lane / lpm, il = lane % lpm, ngrp = 64 / lpm;
  const bool on = 4 * il < d;
  const int f = on ? 4 * il : 0;
  const long long wave0 = ((long long)blockIdx.x * TB + threadIdx.x) >> 6, nw = ((long long)gridDim.x * TB) >> 6;
  constexpr int MB = 2;      // messages in flight per lane group
  constexpr int NX = MODE ? 1 : NB, NA = MODE ? NB : 1;
  struct Idx { int src[MB], rel[MB]; float v[MB]; };
  struct Rows { f32x4 x[MB][NX]; float c[MB][NB]; };
  auto load_idx = [&](int e0, int e1, int base, Idx &ix) {
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const int eb = base + sub + m * ngrp;
      const int e = max(min(eb, e1 - 1), 0);
      ix.src[m] = p_src[e]; ix.rel[m] = p_rel[e];
      const float pv = p_val[e];
      ix.v[m] = eb < e1 ? pv : 0.f;
    }
  };
  auto gather = [&](const Idx &ix, Rows &g) {
#pragma unroll
    for (int m = 0; m < MB; ++m) {
#pragma unroll
      for (int b = 0; b < NB; ++b) g.c[m][b] = comps[(size_t)ix.rel[m] * B + min(b, B - 1)];
#pragma unroll
      for (int b = 0; b < NX; ++b)
        g.x[m][b] = *reinterpret_cast<const f32x4 *>(X + (MODE ? (size_t)ix.src[m] : (size_t)ix.src[m] * B + min(b, B - 1)) * d + f);
    }
  };
  long long row = wave0;
  if (row >= n_rows) return;
  if (rowptr[n_rows] == 0) {                           // no messages at all (the unconditional index loads would read entry 0 of nothing): zeros
    for (; row < n_rows; row += nw)
      if (sub == 0 && on)
        for (int q = 0; q < (MODE ? B : 1); ++q) *reinterpret_cast<f32x4 *>(out + ((size_t)row * (MODE ? B : 1) + q) * d + f) = f32x4{0.f, 0.f, 0.f, 0.f};
    return;
  }
  int e0 = rowptr[row], e1 = rowptr[row + 1];
  Idx cur;
  load_idx(e0, e1, e0, cur);
  int ne0 = rowptr[min(row + nw, n_rows - 1)], ne1 = rowptr[min(row + nw, n_rows - 1) + 1];
  Rows g;
  gather(cur, g);
  for (; row < n_rows; row += nw) {
    const long long nnrow = min(row + 2 * nw, n_rows - 1);
    Idx nxt;
    load_idx(ne0, ne1, ne0, nxt);                         // (row k + 1's indices: in flight under row k's sums)
    f32x4 a[NA];
#pragma unroll
    for (int q = 0; q < NA; ++q) a[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto sum = [&](const Idx &ix, const Rows &gr) {
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const float cv = b < B ? gr.c[m][b] * ix.v[m] : 0.f;
          a[MODE ? b : 0] += gr.x[m][MODE ? 0 : b] * cv;
        }
    };
    sum(cur, g);
    for (int base = e0 + ngrp * MB; base < e1; base += ngrp * MB) {       // rows longer than one trip: their further trips are not pipelined
      Idx t;
      Rows gt;
      load_idx(e0, e1, base, t);
      gather(t, gt);
      sum(t, gt);
    }
#pragma unroll
    for (int q = 0; q < NA; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c) a[q][c] = group_sum(a[q][c], lpm);
    // row k + 2's pointers, then row k + 1's gathers -- all issued before row k's store
    const int nne0 = rowptr[nnrow], nne1 = rowptr[nnrow + 1];
    gather(nxt, g);
    if (sub == 0 && on) {
      if (MODE) {
#pragma unroll
        for (int q = 0; q < NA; ++q)
          if (q < B) *reinterpret_cast<f32x4 *>(out + ((size_t)row * B + q) * d + f) = a[q];
      } else {
        *reinterpret_cast<f32x4 *>(out + (size_t)row * d + f) = a[0];
      }
    }
    cur = nxt; e0 = ne0; e1 = ne1; ne0 = nne0; ne1 = nne1;
  }
}

// Relation-major work items (chunk ranges of one relation, relation-major plan): per-lane partial sums over the
// whole item, ONE wave reduction and one atomic per (item piece, basis) -- not per message.  Lane groups of lpr
// lanes take alternate slots so that narrow rows keep all 64 lanes busy.
template <bool VEC>      // VEC: rows read 16 bytes per lane (d % 4 == 0): a 200-wide row is one trip of 50 lanes instead of four of 64
__global__ __launch_bounds__(TB) void basis_dcomps_kernel(
    const float *__restrict__ X, const float *__restrict__ D, float *__restrict__ dcomps,
    const int *__restrict__ p_src, const int *__restrict__ p_dst, const float *__restrict__ p_val,
    const int *__restrict__ chunk_rel, const int2 *__restrict__ items, int n_items, int B, int d, int lpr, int n_copies,
    int copy_floats) {
  const int lane = threadIdx.x & 63;
  const int sub = lane / lpr, il = lane % lpr, ngrp = 64 / lpr;
  const int item = blockIdx.x * (TB / 64) + (threadIdx.x >> 6);
  if (item >= n_items) return;
  dcomps += (size_t)(blockIdx.y % n_copies) * copy_floats;       // the pieces of an item add into different copies
  const int2 whole = items[item];
  const int r = chunk_rel[whole.x];
  // blockIdx.y = piece of the item (an item of the shared work list can hold 1024 messages: too long for one wave)
  const int per = (whole.y - whole.x + (int)gridDim.y - 1) / (int)gridDim.y;
  int2 range;
  range.x = whole.x + (int)blockIdx.y * per;
  range.y = min(whole.y, range.x + per);
  if (range.x >= range.y) return;
  for (int b0 = 0; b0 < B; b0 += 4) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const int e_end = range.y * RGCN_CHUNK;
    for (int e = range.x * RGCN_CHUNK + sub; e < e_end; e += 2 * ngrp) {       // two messages' row loads in flight together
      const int e2 = min(e + ngrp, e_end - 1);
      const float v = p_val[e], v2 = (e + ngrp < e_end) ? p_val[e2] : 0.f;
      if (v == 0.f && v2 == 0.f) continue;       // pads
      // (either index array may carry the pads' -1: the callers swap them for the featureless layer)
      const float *x = X + (size_t)max(p_src[e], 0) * d, *x2 = X + (size_t)max(p_src[e2], 0) * d;
      const float *dd = D + ((size_t)max(p_dst[e], 0) * B + b0) * d, *dd2 = D + ((size_t)max(p_dst[e2], 0) * B + b0) * d;
      if (VEC) {
        auto dot4 = [](const f32x4 &p, const f32x4 &q) { return (p[0] * q[0] + p[1] * q[1]) + (p[2] * q[2] + p[3] * q[3]); };
        for (int f = 4 * il; f < d; f += 4 * lpr) {
          const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + f) * v, xw = *reinterpret_cast<const f32x4 *>(x2 + f) * v2;
          a0 += dot4(xv, *reinterpret_cast<const f32x4 *>(dd + f)) + dot4(xw, *reinterpret_cast<const f32x4 *>(dd2 + f));
          if (b0 + 1 < B) a1 += dot4(xv, *reinterpret_cast<const f32x4 *>(dd + (size_t)d + f)) + dot4(xw, *reinterpret_cast<const f32x4 *>(dd2 + (size_t)d + f));
          if (b0 + 2 < B) a2 += dot4(xv, *reinterpret_cast<const f32x4 *>(dd + 2 * (size_t)d + f)) + dot4(xw, *reinterpret_cast<const f32x4 *>(dd2 + 2 * (size_t)d + f));
          if (b0 + 3 < B) a3 += dot4(xv, *reinterpret_cast<const f32x4 *>(dd + 3 * (size_t)d + f)) + dot4(xw, *reinterpret_cast<const f32x4 *>(dd2 + 3 * (size_t)d + f));
        }
      } else {
        for (int i = il; i < d; i += lpr) {
          const float xv = v * x[i], xw = v2 * x2[i];
          a0 += xv * dd[i] + xw * dd2[i];
          if (b0 + 1 < B) a1 += xv * dd[(size_t)d + i] + xw * dd2[(size_t)d + i];
          if (b0 + 2 < B) a2 += xv * dd[2 * (size_t)d + i] + xw * dd2[2 * (size_t)d + i];
          if (b0 + 3 < B) a3 += xv * dd[3 * (size_t)d + i] + xw * dd2[3 * (size_t)d + i];
        }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      a0 += __shfl_xor(a0, off, 64); a1 += __shfl_xor(a1, off, 64);
      a2 += __shfl_xor(a2, off, 64); a3 += __shfl_xor(a3, off, 64);
    }
    if (lane == 0) {
      atomicAdd(&dcomps[(size_t)r * B + b0], a0);
      if (b0 + 1 < B) atomicAdd(&dcomps[(size_t)r * B + b0 + 1], a1);
      if (b0 + 2 < B) atomicAdd(&dcomps[(size_t)r * B + b0 + 2], a2);
      if (b0 + 3 < B) atomicAdd(&dcomps[(size_t)r * B + b0 + 3], a3);
    }
  }
}

int lanes_per_row(int d) {
  int l = 1;
  while (l < d && l < 64) l <<= 1;
  return l;
}

}  // namespace

extern "C" int rgcn_basis_aggregate_f32(const float *X, const float *comps, float *out, const int32_t *rowptr,
                                        const int32_t *p_src, const int32_t *p_rel, const float *p_val, int64_t n_rows,
                                        int32_t R, int32_t B, int32_t d, int32_t n_b_in, void *stream) {
  (void)R;
  if (!X || !comps || !out || !rowptr || n_rows < 0 || B <= 0 || d <= 0 || (n_b_in != 1 && n_b_in != B)) { rgcn_set_error("basis_aggregate: bad argument"); return RGCN_EINVAL; }
  if (!n_rows) return RGCN_OK;
  if ((d & 3) == 0 && d >= 16 && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
    int lpm = 1;
    while (lpm < 64 && 4 * lpm < d) lpm *= 2;
    if (d <= 256 && B <= 4) {                            // one pass per row, few bases: the rows pipelined over persistent waves
      static int n_cu = 0;
      if (!n_cu) {
        int dev = 0, v = 0;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
        n_cu = v > 0 ? v : 256;
      }
      const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_rows * 64 + TB - 1) / TB, (int64_t)n_cu * 8));
#define RGCN_BAP(MODE_, NB_) hipLaunchKernelGGL((basis_aggregate_pipe_kernel<MODE_, NB_>), dim3(grid), dim3(TB), 0, (hipStream_t)stream, X, comps, \
                                                out, rowptr, p_src, p_rel, p_val, (long long)n_rows, B, d, lpm)
      if (n_b_in == 1) { if (B <= 2) RGCN_BAP(1, 2); else RGCN_BAP(1, 4); }
      else { if (B <= 2) RGCN_BAP(0, 2); else RGCN_BAP(0, 4); }
#undef RGCN_BAP
      HIP_TRY(hipGetLastError());
      return RGCN_OK;
    }
    if (n_b_in == 1)
      hipLaunchKernelGGL(basis_aggregate_vec4_kernel<1>, dim3(blocks_for(n_rows * 64)), dim3(TB), 0, (hipStream_t)stream, X, comps, out,
                         rowptr, p_src, p_rel, p_val, (long long)n_rows, B, d, lpm);
    else
      hipLaunchKernelGGL(basis_aggregate_vec4_kernel<0>, dim3(blocks_for(n_rows * 64)), dim3(TB), 0, (hipStream_t)stream, X, comps, out,
                         rowptr, p_src, p_rel, p_val, (long long)n_rows, B, d, lpm);
    HIP_TRY(hipGetLastError());
    return RGCN_OK;
  }
  hipLaunchKernelGGL(basis_aggregate_kernel, dim3(blocks_for(n_rows * 64)), dim3(TB), 0, (hipStream_t)stream, X, comps, out,
                     rowptr, p_src, p_rel, p_val, (long long)n_rows, B, d, n_b_in, lanes_per_row(d));
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

// ------------------------------------------------------------------ featureless basis layer, SMALL blocks: the fused backward
// S2 (S1's graph, featureless layer 1, B = 2, d = 16): a node's B x d block is ONE 128-byte line, and the source-major
// kernels above (lane = basis) would run 2 lanes of 64.  Round 3 took the destination-major fallback: basis_aggregate on the
// source-major CSR for dbases (gathers G[s]: 0.69 ms) and basis_dcomps on the relation-major plan for dcomps (gathers G[s] AND the
// block again: 1.13 ms).  Here ONE walk of the source-major CSR does both: the wave holds the row's block in registers (read once),
// gathers G[s] once per message, accumulates dbases[o] in registers (written once, no atomics) and adds t_e[b] = val <block[b], G[s]>
// to an LDS table dcomps[R][B] kept in DOUBLES (ds_add_f64: the LDS float atomic that is native on gfx950); every workgroup
// flushes its table once.  lpm = d / 4 lanes per message (16-byte loads), 64 / lpm messages of a row in flight per wave.
// The workgroups' LDS tables of doubles -> out[0..n) as floats, without n atomics per workgroup on the same few addresses (R x B = 74 on WN18:
// 2048 workgroups x 74 fp32 atomics serialise at the L2 -- most of the kernel's time) and in a FIXED order: every workgroup writes its table
// to its row of `scratch`, the one that finishes last adds the rows in workgroup order.  ticket: one zeroed word, left zeroed.
namespace {
constexpr int TBW = 1024;      // threads of the persistent workgroups that end with this flush
__device__ __forceinline__ void table_flush_ordered(const double *dcl, int n, double *__restrict__ scratch, unsigned *__restrict__ ticket,
                                                    float *__restrict__ out) {
  __shared__ bool last;
  // rows of whole 128-byte lines: a line shared by two workgroups' rows lives half-written in the L2s of two XCDs, and the reader's own L2
  // then serves the neighbour's half from what it holds -- LAST launch's values (found as a gradient one step stale on a 2-workgroup launch)
  const size_t ns = ((size_t)n + 15) & ~(size_t)15;
  double *mine = scratch + (size_t)blockIdx.x * ns;
  // The row goes out with agent-scope atomic EXCHANGES: read-modify-write atomics are performed where all XCDs see them (that is what makes
  // the fp32 atomics of the other kernels add up across the chip), a plain or even an "atomic" store sits in the writer's L2 until that L2 is
  // written back -- and a __threadfence() per workgroup (an L2 write-back each, 512 of them) cost more than the kernel.  The ticket is taken
  // once the exchanges have returned (they return the old value: waiting for it IS the acknowledgement).
  double seen = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) seen += __hip_atomic_exchange(mine + i, dcl[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("" ::"v"(seen) : "memory");                  // (the returned values are waited for, not used; "memory": the compiler may not move the ticket above this point)
  __syncthreads();
  if (threadIdx.x == 0) last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  // consecutive threads read consecutive entries of one workgroup's row; the rows are dealt over S slices of threads (workgroup b -> slice
  // b % S), every thread keeps eight loads in flight, the slices' partial sums meet in LDS in slice order: a fixed order, ~nb / S round trips
  // instead of nb (a lone thread per entry walking 512 rows took longer than the kernel's own work)
  __shared__ double slice_sum[TBW];
  const int nb = (int)gridDim.x;
  for (int i0 = 0; i0 < n; i0 += TBW) {
    const int cols = min(n - i0, TBW);
    const int S = max(TBW / cols, 1), sl = (int)threadIdx.x / cols, i = i0 + (int)threadIdx.x % cols;
    double t = 0.0;
    if (sl < S) {
      int b = sl;
      for (; b + 7 * S < nb; b += 8 * S) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __hip_atomic_load(scratch + (size_t)(b + u * S) * ns + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int u = 0; u < 8; ++u) t += v[u];
      }
      for (; b < nb; b += S) t += __hip_atomic_load(scratch + (size_t)b * ns + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      slice_sum[sl * cols + (i - i0)] = t;
    }
    __syncthreads();
    if ((int)threadIdx.x < cols) {
      double tot = 0.0;
      for (int q = 0; q < S; ++q) tot += slice_sum[q * cols + threadIdx.x];
      out[i0 + threadIdx.x] = (float)tot;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *ticket = 0u;
}
inline unsigned persistent_grid(int64_t n_rows, int n_cu) {
  return (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_rows * 64 + TBW - 1) / TBW, (int64_t)n_cu * 2));
}
}  // namespace

extern "C" int64_t rgcn_basis_sum_workspace_bytes(int32_t R, int32_t B) {
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
  return (int64_t)2 * v * (((int64_t)R * B + 15) & ~(int64_t)15) * (int64_t)sizeof(double) + 256;     // (rows of whole lines; ticket + alignment slack)
}

namespace {
// LPM: lanes per message when known at compile time (4: d = 16, the models' hidden width), 0 = the run-time value lpm_rt
template <int B, int LPM>
__global__ __launch_bounds__(TB) void fbasis_small_bwd_kernel(
    const float *__restrict__ G, const float *__restrict__ table, const float *__restrict__ comps, float *__restrict__ dB,
    float *__restrict__ dC, const int *__restrict__ rowptr, const int *__restrict__ p_src, const int *__restrict__ p_rel,
    const float *__restrict__ p_val, long long n_rows, int R, int d, int lpm_rt, long long bstride) {
  const int lpm = LPM ? LPM : lpm_rt;
  extern __shared__ __attribute__((aligned(16))) double dcl[];          // [R][B]
  for (int i = threadIdx.x; i < R * B; i += TB) dcl[i] = 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int sub = lane / lpm, il = lane % lpm, ngrp = 64 / lpm;
  const bool on = 4 * il < d;
  const int f = on ? 4 * il : 0;          // (lanes past the row's end load feature 0 without a branch and keep nothing)
  const long long wave0 = ((long long)blockIdx.x * TB + threadIdx.x) >> 6, nw = ((long long)gridDim.x * TB) >> 6;
  constexpr int MB = 2;      // messages in flight per lane group
  // A row is a chain of three dependent round trips (row pointers -> the messages' indices -> the upstream rows and coefficients) and the
  // average row (21 messages at S2) fits ONE trip of the message loop: nothing inside a row overlaps them.  So the rows are software-
  // pipelined -- while row k is summed, the indices of row k + 1 are in flight, and its gathers are issued (behind the row pointers of
  // row k + 2) BEFORE row k's gradient block is stored: s_waitcnt vmcnt counts loads and stores in order, loads issued after the store
  // would wait for the store to complete.  (Two rows of gathers in flight per wave, in two register sets: 107 VGPRs = 4 waves per SIMD
  // instead of 6, measured 0.645 ms at S2 against 0.618 for this form; before the pipelining 0.750.)
  struct Idx { int src[MB], rel[MB]; float v[MB]; };
  const size_t n_floats_row = (size_t)d;
  auto load_idx = [&](int e0, int e1, int base, Idx &ix) {
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const int eb = base + sub + m * ngrp;
      const int e = max(min(eb, e1 - 1), 0);
      ix.src[m] = p_src[e]; ix.rel[m] = p_rel[e];
      const float pv = p_val[e];
      ix.v[m] = eb < e1 ? pv : 0.f;
    }
  };
  auto tab_at = [&](long long row, int b) -> size_t {
    // bstride: floats between a node's consecutive bases -- d for the node-major table [N, B, d], N d for the parameter's own [B, N, d]
    return (bstride == d ? ((size_t)row * B + b) * n_floats_row : (size_t)b * bstride + (size_t)row * n_floats_row) + f;
  };
  struct Rows { f32x4 x[MB]; float c[MB][B]; };
  auto gather = [&](const Idx &ix, Rows &g) {
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      g.x[m] = *reinterpret_cast<const f32x4 *>(G + (size_t)ix.src[m] * d + f);
      if (!on) g.x[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int b = 0; b < B; ++b) g.c[m][b] = comps[(size_t)ix.rel[m] * B + b];
    }
  };
  long long row = wave0;
  if (row < n_rows && rowptr[n_rows] == 0) {           // no messages at all (the unconditional index loads would read entry 0 of nothing): zeros
    for (; row < n_rows; row += nw)
      if (sub == 0 && on)
        for (int b = 0; b < B; ++b) *reinterpret_cast<f32x4 *>(dB + tab_at(row, b)) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (row < n_rows) {
    int e0 = rowptr[row], e1 = rowptr[row + 1];
    Idx cur;
    load_idx(e0, e1, e0, cur);
    int ne0 = rowptr[min(row + nw, n_rows - 1)], ne1 = rowptr[min(row + nw, n_rows - 1) + 1];
    Rows g;
    f32x4 blk[B];
    gather(cur, g);
#pragma unroll
    for (int b = 0; b < B; ++b) blk[b] = *reinterpret_cast<const f32x4 *>(table + tab_at(row, b));     // (lanes that are not `on`: multiplied with zeros)
    for (; row < n_rows; row += nw) {
      const long long nrow = min(row + nw, n_rows - 1), nnrow = min(row + 2 * nw, n_rows - 1);
      Idx nxt;
      load_idx(ne0, ne1, ne0, nxt);                       // (row k + 1's indices: in flight under row k's sums)
      f32x4 a[B];
#pragma unroll
      for (int b = 0; b < B; ++b) a[b] = f32x4{0.f, 0.f, 0.f, 0.f};
      auto sum = [&](const Idx &ix, const Rows &gr) {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
#pragma unroll
          for (int b = 0; b < B; ++b) {
            a[b] += gr.x[m] * (gr.c[m][b] * ix.v[m]);
            float dot = blk[b][0] * gr.x[m][0] + blk[b][1] * gr.x[m][1] + blk[b][2] * gr.x[m][2] + blk[b][3] * gr.x[m][3];
            for (int off = 1; off < lpm; off <<= 1) dot += __shfl_xor(dot, off, 64);
            if (il == 0 && ix.v[m] != 0.f)
              __hip_atomic_fetch_add(dcl + ix.rel[m] * B + b, (double)(ix.v[m] * dot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      };
      sum(cur, g);
      for (int base = e0 + ngrp * MB; base < e1; base += ngrp * MB) {     // rows longer than one trip: their further trips are not pipelined
        Idx t;
        Rows gt;
        load_idx(e0, e1, base, t);
        gather(t, gt);
        sum(t, gt);
      }
#pragma unroll
      for (int b = 0; b < B; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) a[b][c] = group_sum(a[b][c], lpm);
      // row k + 2's pointers, then row k + 1's gathers -- all issued before row k's store
      const int nne0 = rowptr[nnrow], nne1 = rowptr[nnrow + 1];
      gather(nxt, g);
#pragma unroll
      for (int b = 0; b < B; ++b) blk[b] = *reinterpret_cast<const f32x4 *>(table + tab_at(nrow, b));     // (row k's blocks are spent)
      if (sub == 0 && on) {
#pragma unroll
        for (int b = 0; b < B; ++b) *reinterpret_cast<f32x4 *>(dB + tab_at(row, b)) = a[b];
      }
      cur = nxt; e0 = ne0; e1 = ne1; ne0 = nne0; ne1 = nne1;
    }
  }
  __syncthreads();
  // (atomics on the zeroed table: this kernel wants many small workgroups -- 1024-thread persistent ones with the ordered flush of the
  // CSR dcomps kernel below were 15 % slower at S2 --, and 2048 tables cannot meet in one workgroup's time)
  for (int i = threadIdx.x; i < R * B; i += TB) {
    const float t = (float)dcl[i];
    if (t != 0.f) atomicAdd(dC + i, t);
  }
}
}  // namespace

// dcomps of the aggregate-then-contract basis layer on the DESTINATION-major CSR the forward already walks (round 5; the relation-major
// form above needs the relation-major plan: a dozen launches to build per call on a per-step LP graph -- more than the kernel itself on
// the WN18-shaped step).  dcomps[r,b] = sum_e val_e <X[src_e], D[dst_e, b, :]>: one wave per destination row holds the row's B blocks of D
// in registers (read once), gathers X[src] once per message, and adds the B dot products to an LDS table [R][B] of DOUBLES (ds_add_f64);
// persistent workgroups, one flush each.  lpm lanes per message (16-byte loads), 64 / lpm messages of a row in flight.
namespace {
template <int NB>
__global__ __launch_bounds__(TBW) void basis_dcomps_csr_kernel(
    const float *__restrict__ X, const float *__restrict__ D, float *__restrict__ dC, const int *__restrict__ rowptr,
    const int *__restrict__ p_src, const int *__restrict__ p_rel, const float *__restrict__ p_val, long long n_rows, int R, int B, int d,
    int lpm, double *__restrict__ scratch, unsigned *__restrict__ ticket) {
  extern __shared__ __attribute__((aligned(16))) double dcl[];          // [R][B]
  for (int i = threadIdx.x; i < R * B; i += TBW) dcl[i] = 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int sub = lane / lpm, il = lane % lpm, ngrp = 64 / lpm;
  const long long wave0 = ((long long)blockIdx.x * TBW + threadIdx.x) >> 6, nw = ((long long)gridDim.x * TBW) >> 6;
  for (long long row = wave0; row < n_rows; row += nw) {
    const int e0 = rowptr[row], e1 = rowptr[row + 1];
    if (e0 == e1) continue;
    for (int f0 = 0; f0 < d; f0 += 4 * lpm) {
      const int f = f0 + 4 * il;
      const bool on = f < d;
      f32x4 blk[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b)
        blk[b] = (on && b < B) ? *reinterpret_cast<const f32x4 *>(D + ((size_t)row * B + b) * d + f) : f32x4{0.f, 0.f, 0.f, 0.f};
      for (int eb = e0 + sub; eb < e1 + sub; eb += ngrp) {            // (every group runs the same trips: the shuffles below are wave-wide)
        const bool live = eb < e1;
        const int e = min(eb, e1 - 1);
        const float v = live ? p_val[e] : 0.f;
        const int r = p_rel[e];
        const f32x4 x = on ? *reinterpret_cast<const f32x4 *>(X + (size_t)p_src[e] * d + f) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          float dot = blk[b][0] * x[0] + blk[b][1] * x[1] + blk[b][2] * x[2] + blk[b][3] * x[3];
          for (int off = 1; off < lpm; off <<= 1) dot += __shfl_xor(dot, off, 64);
          if (il == 0 && b < B && v != 0.f)
            __hip_atomic_fetch_add(dcl + r * B + b, (double)(v * dot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
  }
  __syncthreads();
  table_flush_ordered(dcl, R * B, scratch, ticket, dC);
}
}  // namespace

extern "C" int rgcn_basis_dcomps_csr_supported(int32_t R, int32_t B, int32_t d) {
  return B >= 1 && B <= 8 && d >= 4 && (d & 3) == 0 && d <= 1024 && (size_t)R * B * sizeof(double) <= 60 * 1024;
}

extern "C" int rgcn_basis_dcomps_csr_f32(const float *X, const float *D, float *dcomps, const int32_t *rowptr, const int32_t *p_src,
                                         const int32_t *p_rel, const float *p_val, int64_t n_rows, int32_t R, int32_t B, int32_t d,
                                         void *workspace, void *stream) {
  if (!X || !D || !dcomps || !rowptr || !workspace || n_rows < 0 || R <= 0) { rgcn_set_error("basis_dcomps_csr: bad argument"); return RGCN_EINVAL; }
  if (!rgcn_basis_dcomps_csr_supported(R, B, d)) { rgcn_set_error("basis_dcomps_csr: B = %d (1..8), d = %d (multiple of 4), R = %d", B, d, R); return RGCN_EUNSUPPORTED; }
  if (((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(D)) & 15) != 0) { rgcn_set_error("basis_dcomps_csr: X / D must be 16-byte aligned"); return RGCN_EINVAL; }
  hipStream_t st = (hipStream_t)stream;
  if (!n_rows) { HIP_TRY(zero_async(dcomps, (size_t)R * B * sizeof(float), st)); return RGCN_OK; }
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0, v = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
    n_cu = v > 0 ? v : 256;
  }
  const unsigned grid = persistent_grid(n_rows, n_cu);
  const size_t lds = (size_t)R * B * sizeof(double);
  int lpm = 1;
  while (lpm < 64 && 4 * lpm < d) lpm *= 2;
  unsigned *ticket = reinterpret_cast<unsigned *>(workspace);            // (the first bytes: zeroed by the caller once, left zeroed)
  double *scratch = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(workspace) + 128 + 127) & ~(uintptr_t)127);   // line-aligned rows
#define RGCN_BDC(NB_) hipLaunchKernelGGL(basis_dcomps_csr_kernel<NB_>, dim3(grid), dim3(TBW), lds, st, X, D, dcomps, rowptr, p_src, p_rel, p_val, \
                                         (long long)n_rows, R, B, d, lpm, scratch, ticket)
  if (B <= 2) RGCN_BDC(2); else if (B <= 4) RGCN_BDC(4); else RGCN_BDC(8);
#undef RGCN_BDC
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_fbasis_small_supported(int32_t R, int32_t B, int32_t d) {
  return B >= 1 && B <= 4 && d >= 4 && d <= 64 && (d & 3) == 0 && (d & (d - 1)) == 0 && (size_t)R * B * sizeof(double) <= 60 * 1024;
}

extern "C" int rgcn_fbasis_small_bwd_f32(const float *G, const float *table, const float *comps, float *dbases, float *dcomps,
                                         const int32_t *rowptr, const int32_t *p_src, const int32_t *p_rel, const float *p_val,
                                         int64_t n_rows, int32_t R, int32_t B, int32_t d, int32_t basis_major, void *stream) {
  if (!G || !table || !comps || !dbases || !dcomps || !rowptr || n_rows < 0 || R <= 0) { rgcn_set_error("fbasis_small_bwd: bad argument"); return RGCN_EINVAL; }
  if (!rgcn_fbasis_small_supported(R, B, d)) { rgcn_set_error("fbasis_small_bwd: B = %d (1..4), d = %d (power of two, 4..64), R = %d", B, d, R); return RGCN_EUNSUPPORTED; }
  if (((reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(dbases)) & 15) != 0) {
    rgcn_set_error("fbasis_small_bwd: G / table / dbases must be 16-byte aligned");
    return RGCN_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(zero_async(dcomps, (size_t)R * B * sizeof(float), st));
  if (!n_rows) return RGCN_OK;
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0, v = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
    n_cu = v > 0 ? v : 256;
  }
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_rows * 64 + TB - 1) / TB, (int64_t)n_cu * 8));
  const size_t lds = (size_t)R * B * sizeof(double);
  const int lpm = d / 4;
#define RGCN_FBS(BB) do { if (lpm == 4) RGCN_FBS2(BB, 4); else RGCN_FBS2(BB, 0); } while (0)
#define RGCN_FBS2(BB, LL) hipLaunchKernelGGL((fbasis_small_bwd_kernel<BB, LL>), dim3(grid), dim3(TB), lds, st, G, table, comps, dbases, dcomps, rowptr, \
                                             p_src, p_rel, p_val, (long long)n_rows, R, d, lpm, basis_major ? (long long)n_rows * d : (long long)d)
  if (B == 1) RGCN_FBS(1); else if (B == 2) RGCN_FBS(2); else if (B == 3) RGCN_FBS(3); else RGCN_FBS(4);
#undef RGCN_FBS2
#undef RGCN_FBS
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_basis_dcomps_f32(const float *X, const float *D, float *dcomps, const int32_t *p_src,
                                     const int32_t *p_dst, const float *p_val, const int32_t *chunk_rel,
                                     const int32_t *items, int64_t n_items, int32_t R, int32_t B, int32_t d,
                                     int32_t n_copies, void *stream) {
  if (!X || !D || !dcomps || n_items < 0 || R <= 0 || B <= 0 || d <= 0 || n_copies < 1) { rgcn_set_error("basis_dcomps: bad argument"); return RGCN_EINVAL; }
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(zero_async(dcomps, (size_t)n_copies * R * B * sizeof(float), st));
  if (!n_items) return RGCN_OK;
  // few items (small per-call graphs): cut every item into more pieces so that the chip is filled
  const unsigned pieces = n_items < 2048 ? 64 : 16;
  const bool vec = (d & 3) == 0 && d >= 32 && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(D)) & 15) == 0;
  int lpv = 1;                                   // lanes per row with 16-byte loads
  while (lpv < 64 && 4 * lpv < d) lpv *= 2;
  if (vec) {
    hipLaunchKernelGGL(basis_dcomps_kernel<true>, dim3((unsigned)((n_items + TB / 64 - 1) / (TB / 64)), pieces), dim3(TB), 0, st, X, D, dcomps,
                     p_src, p_dst, p_val, chunk_rel, reinterpret_cast<const int2 *>(items), (int)n_items, B, d, lpv, (int)n_copies, R * B);
  } else {
    hipLaunchKernelGGL(basis_dcomps_kernel<false>, dim3((unsigned)((n_items + TB / 64 - 1) / (TB / 64)), pieces), dim3(TB), 0, st, X, D, dcomps,
                     p_src, p_dst, p_val, chunk_rel, reinterpret_cast<const int2 *>(items), (int)n_items, B, d, lanes_per_row(d), (int)n_copies, R * B);
  }
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}
