// Fused backward of the hidden-16 relational layer: ONE random row gather per message.
//
// Autograd duals of reference torch_rgcn/layers.py:293-301:
//     dX[o]  += val * G[s] W_r^T            dW_r += val * X[o]^T G[s]            for every message (s <- o, r, val)
// Round 1 computed them in two passes (spmm on the transposed plan gathers G[s]; the tile-major weight gradient gathers
// X[o]): two random 64-byte row reads per message.  Here the transposed plan (destination tile = tile of OBJECT rows o)
// is walked once: G[s] is gathered once per message and used for both results, X[o] is tile-local (a few hundred
// bytes of L1/L2 hits per chunk).
//
// What makes this hard is where the dW partials go.  A wave that owns a destination tile sees ~2 chunks of one relation
// before the relation changes, so it holds a finished 16x16 partial of dW_r every ~2 chunks.  Measured alternatives
// (tools/atomic_probe.hip, profiles/r02_atomic_probe.txt): flushing those 789 k partials per launch with fp32 global
// atomics costs 0.6-0.8 ms on its own (a wave-wide atomic instruction occupies the CU's texture path for ~125
// cycles, one copy or one copy per XCD alike); LDS float atomics are as slow (round 1: ~160 cycles per instruction).
// So the partials are reduced ACROSS the four waves of the workgroup first and leave the CU as plain stores:
//
//   workgroup = 4 waves = 4 adjacent destination tiles; every wave streams through its tile's chunks (sorted by
//   relation) exactly like the forward kernel.  Relations are grouped in intervals of D = 4.  Inside an interval a wave
//   parks each finished partial in its own LDS staging slot [wave][relation % 4] (1 KiB, ds_write_b128); when it crosses
//   the interval boundary it waits at a workgroup barrier, wave w sums the four waves' slots of relation 4 iv + w and
//   stores the 1 KiB sum, coalesced, to partial[r][workgroup][256]; second barrier; next interval.  Loads of the chunks
//   after the boundary are already in flight while a wave waits.  Every wave passes every boundary (26 barrier pairs at
//   R = 101), whatever its tile holds.
//   A second, tiny kernel pair sums partial[r][*] in a fixed order -> dW: no atomics anywhere, bit-reproducible.
//
// Per chunk of 16 messages (one relation), lane = 16 k + m:
//   gather       lane (k, m): G[s_m][4k..4k+3]                     (one 16-byte load, the only random HBM access)
//   dX           D^T[o'][slot] = sum_f Wt_r[f][o'] (val G[s_slot][f])   4 x v_mfma_f32_16x16x4_f32, DPP segment fold,
//                one ds_read/ds_write_b128 per destination segment into the wave-owned LDS tile (as the forward kernel)
//   dW           the scaled rows go through a 1.25 KiB LDS scratch into K-over-messages operand layout
//                (B[mu][j] = val G[s_mu][j]; A[i][mu] = X[o_mu][i], the tile-local rows read as 16-byte row quarters from
//                L1/L2 -- keeping the tile's X rows in LDS instead was measured: 0.94 ms against 0.73, fewer resident
//                waves and LDS traffic cost more than the saved requests); 4 x v_mfma_f32_16x16x4_f32 accumulate dW_r in
//                4 registers per lane.
#include <stdlib.h>

#include <algorithm>

#include "rgcn_device.h"

namespace {

// BW_D = relations per barrier interval (<= 4 = waves per workgroup: wave w reduces relation BW_D iv + w), template parameter
constexpr int BW_SCR = 16 * 20;      // floats of transposition scratch per wave (row stride 20: conflict-free b128 writes)

template <int U>
struct BwdStage {
  int s[U];        // source row (of G) of slot m
  float v[U];      // adjacency value of slot m
  int d[U];        // destination row of slot m (-1: pad)
  int dl[U];       // destination row inside the tile (0xFF: pad)
  int r[U];        // relation (wave-uniform)
  float4 g[U];     // gathered G[s_m][4k..4k+3]
  float4 w[U];     // W_r^T fragment
  float4 xn[U];    // X[o_m][4k..4k+3] (tile-local row of the destination)
};

template <int U, bool ATOMIC, int BW_D, int NW = 4>      // NW = waves (= tiles) per workgroup
__global__ __launch_bounds__(64 * NW, 16 / NW) void bwd_fused_d16_kernel(
    const float *__restrict__ G, const float *__restrict__ X, const float *__restrict__ Wtp, float *__restrict__ dX,
    float *__restrict__ dWout, const int2 *__restrict__ p_pack, const int *__restrict__ chunk_rel,
    const int *__restrict__ run_ptr, int n_tiles, int n_blocks, int tile_rows, int n_dst, int R) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int t = blockIdx.x * NW + wave;
  const bool valid = t < n_tiles;
  float *tile = lds + wave * tile_rows * 16;
  float *xs = lds + NW * tile_rows * 16 + wave * BW_SCR;
  float *stage = lds + NW * tile_rows * 16 + NW * BW_SCR;      // [wave][BW_D][256], fragment order
  float *my_stage = stage + wave * (BW_D * 256);
  const int row0 = t * tile_rows;
  const int nrows = valid ? min(tile_rows, n_dst - row0) : 0;
  for (int i = lane; i < nrows * 4; i += 64) reinterpret_cast<float4 *>(tile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int q = 0; q < BW_D; ++q) reinterpret_cast<f32x4 *>(my_stage + q * 256)[lane] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int my0 = valid ? run_ptr[(size_t)t * (R + 1)] : 0, my1 = valid ? run_ptr[(size_t)t * (R + 1) + R] : 0;
  const int m = lane & 15, k = lane >> 4;
  const int n_iv = (R + BW_D - 1) / BW_D;
  int iv = 0;                 // current interval: relations [BW_D iv, BW_D iv + BW_D)
  int cur = -1;               // relation accumulating in acc_w (-1: none)
  f32x4 acc_w = {0.f, 0.f, 0.f, 0.f};

  // close the current interval: park the open partial, meet the other waves, reduce one relation, meet again
  auto close_interval = [&]() {
    if (cur >= 0) {
      reinterpret_cast<f32x4 *>(my_stage + (cur - iv * BW_D) * 256)[lane] = acc_w;
      cur = -1;
      acc_w = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    const int r = iv * BW_D + wave;
    if (wave < BW_D && r < R) {
      f32x4 sum = reinterpret_cast<const f32x4 *>(stage + (0 * BW_D + wave) * 256)[lane];
#pragma unroll
      for (int w2 = 1; w2 < NW; ++w2) sum += reinterpret_cast<const f32x4 *>(stage + (w2 * BW_D + wave) * 256)[lane];
      if (ATOMIC) {       // D: lane 16q+j holds rows 4q..4q+3 (input feature), column j (output feature)
        float *wr = dWout + (size_t)r * 256 + (4 * k) * 16 + m;
        atomicAdd(wr, sum[0]); atomicAdd(wr + 16, sum[1]); atomicAdd(wr + 32, sum[2]); atomicAdd(wr + 48, sum[3]);
      } else {
        reinterpret_cast<f32x4 *>(dWout + ((size_t)r * n_blocks + blockIdx.x) * 256)[lane] = sum;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < BW_D; ++q) reinterpret_cast<f32x4 *>(my_stage + q * 256)[lane] = f32x4{0.f, 0.f, 0.f, 0.f};
    ++iv;
  };

  if (my0 < my1) {
    const int last = my1 - 1;
    // index data (packed slots + relations) of the NEXT group of U chunks is requested while the current group's gathers
    // are in flight: a wave's iteration is then gather latency + compute, not index latency + gather latency + compute
    // (0.77 -> 0.72 ms at S1)
    int2 pk_n[U];
    int relv_n;
    auto request_idx = [&](int c) {
      relv_n = chunk_rel[min(c + (lane & (U - 1)), last)];
#pragma unroll
      for (int j = 0; j < U; ++j) pk_n[j] = p_pack[min(c + j, last) * RGCN_CHUNK + m];
    };
    request_idx(my0);
    for (int c = my0; c < my1; c += U) {
      BwdStage<U> A;
      // stage 1: unpack the slots requested one iteration ago (chunks past the range re-read the last chunk with val = 0);
      // the U relations come from ONE vector load (a scalar load per chunk makes hipcc wait for each in turn)
      const int relv = relv_n;
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int2 pk = pk_n[j];
        A.s[j] = pk.x & 0xFFFFFF;
        A.dl[j] = (int)((unsigned)pk.x >> 24);
        A.d[j] = A.dl[j] == 0xFF ? -1 : row0 + A.dl[j];
        A.v[j] = (c + j <= last) ? __builtin_bit_cast(float, pk.y) : 0.f;
      }
#pragma unroll
      for (int j = 0; j < U; ++j) A.r[j] = __builtin_amdgcn_readlane(relv, j);
#pragma unroll
      for (int j = 0; j < U; ++j) asm volatile("" : "+v"(A.d[j]), "+v"(A.v[j]));   // pin the index data here
      __builtin_amdgcn_sched_barrier(0);
      // stage 2: the random gather, the W_r^T fragment, and the tile-local X rows in operand order
#pragma unroll
      for (int j = 0; j < U; ++j) {
        // 32-bit byte offsets from the uniform base pointers (packed slots: source ids < 2^24, so row << 6 fits): the loads take
        // the scalar-base + vector-offset form instead of a 64-bit address per lane
        const unsigned og = ((unsigned)A.s[j] << 6) | ((unsigned)k << 4);
        const unsigned ox = ((unsigned)(row0 + (A.dl[j] == 0xFF ? 0 : A.dl[j])) << 6) | ((unsigned)k << 4);
        A.g[j] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(G) + og);
        A.w[j] = reinterpret_cast<const float4 *>(Wtp)[(size_t)A.r[j] * 64 + lane];
        A.xn[j] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(X) + ox);
      }
      __builtin_amdgcn_sched_barrier(0);
      request_idx(c + U);                  // behind the gathers in the (in-order) memory pipeline; used next iteration
      __builtin_amdgcn_sched_barrier(0);
      // stage 3: matrix cores
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const float v = A.v[j];
        const bool live = v != 0.f;
        const f32x4 sc = {live ? A.g[j].x * v : 0.f, live ? A.g[j].y * v : 0.f, live ? A.g[j].z * v : 0.f,
                          live ? A.g[j].w * v : 0.f};
        // ---- dX
        f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A.w[j].x, sc[0], acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A.w[j].y, sc[1], acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A.w[j].z, sc[2], acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A.w[j].w, sc[3], acc[0], 0, 0, 0);
        if (fold_segments<1>(acc, A.d[j])) {
          f32x4 *p = reinterpret_cast<f32x4 *>(tile + A.dl[j] * 16 + 4 * (k ^ ((A.dl[j] >> 2) & 3)));   // swizzled: see tile_swz
          *p += acc[0];
        }
        // ---- dW: relation / interval bookkeeping (wave-uniform), then K over the chunk's 16 messages
        const int rj = __builtin_amdgcn_readfirstlane(A.r[j]);
        while (rj >= (iv + 1) * BW_D) close_interval();
        if (rj != cur) {
          if (cur >= 0) reinterpret_cast<f32x4 *>(my_stage + (cur - iv * BW_D) * 256)[lane] = acc_w;
          acc_w = f32x4{0.f, 0.f, 0.f, 0.f};
          cur = rj;
        }
        // both operands go through the wave's LDS scratch into K-over-messages layout (written as rows of a slot, read as
        // one feature of four slots); the LDS pipeline is in order, so one scratch serves both
        float bv[4], av[4];
        asm volatile("" ::: "memory");
        *reinterpret_cast<f32x4 *>(xs + m * 20 + 4 * k) = sc;                    // xs[slot m][4k..4k+3] = val G[s_m][4k..]
        asm volatile("" ::: "memory");
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) bv[t4] = xs[(4 * t4 + k) * 20 + m];       // B[mu][j = m], mu = 4 t4 + k
        asm volatile("" ::: "memory");
        const bool pad = A.dl[j] == 0xFF;                                        // pads contribute nothing (B is 0; keep A finite)
        *reinterpret_cast<f32x4 *>(xs + m * 20 + 4 * k) = f32x4{pad ? 0.f : A.xn[j].x, pad ? 0.f : A.xn[j].y,
                                                                pad ? 0.f : A.xn[j].z, pad ? 0.f : A.xn[j].w};
        asm volatile("" ::: "memory");
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) av[t4] = xs[(4 * t4 + k) * 20 + m];       // A[i = m][mu] = X[o_mu][m]
        asm volatile("" ::: "memory");
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) acc_w = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t4], bv[t4], acc_w, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  while (iv < n_iv) close_interval();

  float4 *o4 = reinterpret_cast<float4 *>(dX + (size_t)row0 * 16);
  for (int i = lane; i < nrows * 4; i += 64) o4[i] = reinterpret_cast<const float4 *>(tile)[tile_swz(i)];
}

// ---- sparse (tile, relation) buckets (AM: 267 relations): backward of the two-pass path with ONE relation-major walk.
// Round 1: pass 1 of the feature gradient gathers G[s] per message (relation-major chunks, dense), and the weight gradient
// walks the same relation-major plan again gathering G[dst] AND X[src] -- three random row reads per message.  Here one
// wave per work item (<= 64 chunks of ONE relation) gathers G[s] and X[o] once each and produces both the transformed rows
// Y[slot] = val G[s] W_r^T (summed per destination by pass 2, rgcn_segment_gather_sum_f32) and the item's share of dW_r,
// which stays in 4 accumulator registers for the whole item (runs are long in relation-major order): one flush of 256
// atomics per item, no staging, no barriers.
template <int U>
__global__ __launch_bounds__(WG) void bwd_scatter_dw_d16_kernel(
    const float *__restrict__ G, const float *__restrict__ X, const float *__restrict__ Wtp, float *__restrict__ Y,
    float *__restrict__ dW, const int *__restrict__ p_src, const int *__restrict__ p_dst, const float *__restrict__ p_val,
    const int *__restrict__ chunk_rel, const int2 *__restrict__ items, int n_items) {
  __shared__ __attribute__((aligned(16))) float scr_all[(WG / 64) * BW_SCR];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int item = blockIdx.x * (WG / 64) + wave;
  if (item >= n_items) return;
  const int2 range = items[item];
  if (range.x >= range.y) return;
  const int r = __builtin_amdgcn_readfirstlane(chunk_rel[range.x]);
  float *xs = scr_all + wave * BW_SCR;
  const int m = lane & 15, k = lane >> 4;
  const float4 w = reinterpret_cast<const float4 *>(Wtp)[(size_t)r * 64 + lane];
  f32x4 acc_w = {0.f, 0.f, 0.f, 0.f};
  const int last = range.y - 1;
  for (int c = range.x; c < range.y; c += U) {
    int s[U], d[U];
    float v[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int e = min(c + j, last) * RGCN_CHUNK + m;
      s[j] = p_src[e];
      d[j] = p_dst[e];
      const float vv = p_val[e];
      v[j] = (c + j <= last) ? vv : 0.f;
    }
#pragma unroll
    for (int j = 0; j < U; ++j) asm volatile("" : "+v"(s[j]), "+v"(d[j]), "+v"(v[j]));
    __builtin_amdgcn_sched_barrier(0);
    float4 g[U], x[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      g[j] = *reinterpret_cast<const float4 *>(G + (size_t)s[j] * 16 + 4 * k);
      x[j] = *reinterpret_cast<const float4 *>(X + (size_t)max(d[j], 0) * 16 + 4 * k);     // pads: dst = -1, val = 0
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const float vv = v[j];
      const bool live = vv != 0.f;
      const f32x4 sc = {live ? g[j].x * vv : 0.f, live ? g[j].y * vv : 0.f, live ? g[j].z * vv : 0.f, live ? g[j].w * vv : 0.f};
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, sc[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, sc[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, sc[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, sc[3], acc, 0, 0, 0);
      if (live) *reinterpret_cast<f32x4 *>(Y + ((size_t)min(c + j, last) * RGCN_CHUNK + m) * 16 + 4 * k) = acc;
      // dW_r += (X[o])^T (val G[s]): both operands through the LDS scratch into K-over-messages layout
      float bv[4], av[4];
      asm volatile("" ::: "memory");
      *reinterpret_cast<f32x4 *>(xs + m * 20 + 4 * k) = sc;
      asm volatile("" ::: "memory");
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) bv[t4] = xs[(4 * t4 + k) * 20 + m];
      asm volatile("" ::: "memory");
      *reinterpret_cast<f32x4 *>(xs + m * 20 + 4 * k) = f32x4{live ? x[j].x : 0.f, live ? x[j].y : 0.f, live ? x[j].z : 0.f,
                                                              live ? x[j].w : 0.f};
      asm volatile("" ::: "memory");
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) av[t4] = xs[(4 * t4 + k) * 20 + m];
      asm volatile("" ::: "memory");
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) acc_w = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t4], bv[t4], acc_w, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float *wr = dW + (size_t)r * 256 + (4 * k) * 16 + m;       // D: lane 16q+j holds rows 4q..4q+3 (input feature), column j
  atomicAdd(wr, acc_w[0]); atomicAdd(wr + 16, acc_w[1]); atomicAdd(wr + 32, acc_w[2]); atomicAdd(wr + 48, acc_w[3]);
}

// partial[r][block][256] (fragment order) -> tmp[r][s][256]: slice s sums blocks s, s + S, ...
__global__ __launch_bounds__(WG) void dw_reduce_a_kernel(const float *__restrict__ partial, float *__restrict__ tmp,
                                                         int n_blocks, int S) {
  const int r = blockIdx.x, s = blockIdx.y, tau = threadIdx.x;
  const float *p = partial + (size_t)r * n_blocks * 256 + tau;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int b = s;
  for (; b + 3 * S < n_blocks; b += 4 * S) {
    a0 += p[(size_t)b * 256];
    a1 += p[(size_t)(b + S) * 256];
    a2 += p[(size_t)(b + 2 * S) * 256];
    a3 += p[(size_t)(b + 3 * S) * 256];
  }
  for (; b < n_blocks; b += S) a0 += p[(size_t)b * 256];
  tmp[((size_t)r * S + s) * 256 + tau] = (a0 + a1) + (a2 + a3);
}

// tmp[r][s][256] -> dW[r][i][j]: fragment element (lane = 16 q + j, c) is dW[4 q + c][j]
__global__ __launch_bounds__(WG) void dw_reduce_b_kernel(const float *__restrict__ tmp, float *__restrict__ dW, int S) {
  const int r = blockIdx.x, tau = threadIdx.x;
  float a = 0.f;
  for (int s = 0; s < S; ++s) a += tmp[((size_t)r * S + s) * 256 + tau];
  const int ln = tau >> 2, c = tau & 3;
  dW[(size_t)r * 256 + (4 * (ln >> 4) + c) * 16 + (ln & 15)] = a;
}

// W[r][f][o] -> fragments of W_r^T: Wp[r][lane = 16 k + f'][c] = W^T[4k + c][f'] = W[r][f'][4 k + c]
__global__ __launch_bounds__(WG) void pack_w16t_kernel(const float *__restrict__ W, float *__restrict__ Wp, int n) {
  const int i = blockIdx.x * WG + threadIdx.x;
  if (i >= n) return;
  const int c = i & 3, o = (i >> 2) & 15, kk = (i >> 6) & 3, r = i >> 8;
  Wp[i] = W[r * 256 + o * 16 + (4 * kk + c)];
}

struct BwdLaunch {
  const float *G, *X, *Wtp;
  float *dX, *dWout;
  const int2 *pk;
  const int *chunk_rel, *run_ptr;
  int n_tiles, n_blocks, tile_rows, n_dst, R;
  size_t lds;
  hipStream_t st;
};

template <int U, bool AT, int D>
void launch_bwd(const BwdLaunch &a) {
  hipLaunchKernelGGL((bwd_fused_d16_kernel<U, AT, D>), dim3((unsigned)a.n_blocks), dim3(WG), a.lds, a.st, a.G, a.X, a.Wtp,
                     a.dX, a.dWout, a.pk, a.chunk_rel, a.run_ptr, a.n_tiles, a.n_blocks, a.tile_rows, a.n_dst, a.R);
}

// 8 tiles per workgroup (512 threads, 2 workgroups per CU): half the dW flushes of the 4-tile form; needs > 64 KiB of LDS
template <bool AT>
hipError_t launch_bwd8(const BwdLaunch &a) {
  auto kern = bwd_fused_d16_kernel<4, AT, 4, 8>;
  static bool raised = false;                    // once per process (not a stream operation: keep it out of captures)
  if (a.lds > 64 * 1024 && !raised) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (e != hipSuccess) return e;
    raised = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)a.n_blocks), dim3(512), a.lds, a.st, a.G, a.X, a.Wtp, a.dX, a.dWout, a.pk, a.chunk_rel,
                     a.run_ptr, a.n_tiles, a.n_blocks, a.tile_rows, a.n_dst, a.R);
  return hipGetLastError();
}

template <int U, bool AT>
void launch_bwd_d(const BwdLaunch &a, int D) {
  if (D == 4) launch_bwd<U, AT, 4>(a);
  else if (D == 2) launch_bwd<U, AT, 2>(a);
  else launch_bwd<U, AT, 1>(a);
}

}  // namespace

extern "C" int rgcn_pack_w16t_f32(const float *W, float *Wp, int32_t R, void *stream) {
  if (!W || !Wp || R <= 0) { rgcn_set_error("pack_w16t: bad argument"); return RGCN_EINVAL; }
  const int n = R * 256;
  hipLaunchKernelGGL(pack_w16t_kernel, dim3((unsigned)((n + WG - 1) / WG)), dim3(WG), 0, (hipStream_t)stream, W, Wp, n);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int64_t rgcn_bwd_fused_scratch_floats(int64_t n_tiles, int32_t R) {
  const int64_t n_blocks = (n_tiles + 3) / 4;
  const int64_t S = std::max<int64_t>(1, std::min<int64_t>(16, n_blocks / 64));
  return (n_blocks * R + (int64_t)R * S) * 256;
}

extern "C" int rgcn_bwd_fused_f32(const float *G, const float *X, const float *Wt_packed, float *dX, float *dW,
                                  float *scratch, const int32_t *p_pack, const int32_t *chunk_rel,
                                  const int32_t *run_ptr, int64_t n_tiles, int32_t tile_rows, int64_t n_dst, int32_t R,
                                  int32_t flags, void *stream) {
  if (!G || !X || !Wt_packed || !dX || !dW || !p_pack || !chunk_rel || !run_ptr || n_tiles <= 0 || tile_rows <= 0 ||
      tile_rows > 255 || n_dst <= 0 || R <= 0) {
    rgcn_set_error("bwd_fused: bad argument");
    return RGCN_EINVAL;
  }
  const bool atomic = (flags & RGCN_F_DW_ATOMIC) != 0;
  if (!atomic && !scratch) { rgcn_set_error("bwd_fused: the deterministic reduction needs a scratch buffer"); return RGCN_EINVAL; }
  static const int DSEL = getenv("RGCN_BWD_D") ? atoi(getenv("RGCN_BWD_D")) : 4;
  const int Dv = DSEL >= 4 ? 4 : (DSEL >= 2 ? 2 : 1);
  // tiles per workgroup: 8 halve the number of dW partials -- measured at S1 (profiles/r02_bwd_fused_ablation.txt): atomic flush
  // 0.719 -> 0.740 ms (the barrier now waits for the slowest of 8 waves, which costs more than the saved atomics), plain-store
  // flush of the deterministic mode 0.798 -> 0.769 ms (half the partial bytes) -- so 8 only there
  static const int NWSEL = getenv("RGCN_BWD_WAVES") ? atoi(getenv("RGCN_BWD_WAVES")) : 0;
  const int NWwant = NWSEL ? NWSEL : (atomic ? 4 : 8);
  const int NWv = (NWwant >= 8 && Dv == 4 && tile_rows <= 64) ? 8 : 4;
  const size_t lds = ((size_t)NWv * tile_rows * 16 + NWv * BW_SCR + NWv * Dv * 256) * sizeof(float);
  if (lds > (NWv == 8 ? 80 : 64) * 1024) { rgcn_set_error("bwd_fused: tile_rows = %d needs %zu bytes of LDS per workgroup", tile_rows, lds); return RGCN_EUNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  const int n_blocks = (int)((n_tiles + NWv - 1) / NWv);
  static const int USEL = getenv("RGCN_BWD_U") ? atoi(getenv("RGCN_BWD_U")) : 4;
  const int2 *pk = reinterpret_cast<const int2 *>(p_pack);
  const BwdLaunch L{G, X, Wt_packed, dX, atomic ? dW : scratch, pk, chunk_rel, run_ptr, (int)n_tiles, n_blocks, tile_rows,
                    (int)n_dst, R, lds, st};
  if (atomic) {
    HIP_TRY(zero_async(dW, (size_t)R * 256 * sizeof(float), st));
    if (NWv == 8) HIP_TRY(launch_bwd8<true>(L));
    else if (USEL >= 4) launch_bwd_d<4, true>(L, Dv); else launch_bwd_d<2, true>(L, Dv);
  } else {
    if (NWv == 8) HIP_TRY(launch_bwd8<false>(L));
    else if (USEL >= 4) launch_bwd_d<4, false>(L, Dv); else launch_bwd_d<2, false>(L, Dv);
    const int S = (int)std::max<int64_t>(1, std::min<int64_t>(16, n_blocks / 64));
    float *tmp = scratch + (size_t)n_blocks * R * 256;
    hipLaunchKernelGGL(dw_reduce_a_kernel, dim3((unsigned)R, (unsigned)S), dim3(WG), 0, st, scratch, tmp, n_blocks, S);
    hipLaunchKernelGGL(dw_reduce_b_kernel, dim3((unsigned)R), dim3(WG), 0, st, tmp, dW, S);
  }
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_bwd_scatter_dw_f32(const float *G, const float *X, const float *Wt_packed, float *Y, float *dW,
                                       const int32_t *p_src, const int32_t *p_dst, const float *p_val,
                                       const int32_t *chunk_rel, const int32_t *items, int64_t n_items, int32_t R, int32_t d,
                                       void *stream) {
  if (!G || !X || !Wt_packed || !Y || !dW || R <= 0 || n_items < 0 || (n_items && (!p_src || !p_dst || !p_val || !chunk_rel || !items))) {
    rgcn_set_error("bwd_scatter_dw: bad argument");
    return RGCN_EINVAL;
  }
  if (d != 16) { rgcn_set_error("bwd_scatter_dw: only d = 16"); return RGCN_EUNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(zero_async(dW, (size_t)R * 256 * sizeof(float), st));
  if (!n_items) return RGCN_OK;
  hipLaunchKernelGGL(bwd_scatter_dw_d16_kernel<4>, dim3((unsigned)((n_items + WG / 64 - 1) / (WG / 64))), dim3(WG), 0, st, G, X,
                     Wt_packed, Y, dW, p_src, p_dst, p_val, chunk_rel, reinterpret_cast<const int2 *>(items), (int)n_items);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}
