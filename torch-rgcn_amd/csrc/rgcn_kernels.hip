// Device half of librgcn_hip.so: hand-written gfx950 (CDNA4) kernels for the R-GCN
// relational message-passing hot path.  Contract: include/rgcn_hip.h.
//
// Layout the kernels consume (built by rgcn_plan_fill_host): messages bucketed by
// (destination tile, relation), padded so that every chunk of 16 slots has ONE
// relation and ONE destination tile.
//
//   spmm  : one workgroup per destination tile.  The tile's output rows live in LDS.
//           A wave takes a chunk of 16 messages, gathers their 16 source rows from HBM
//           (one coalesced row segment per 16 lanes), multiplies the 16 x d_in block
//           by W_rel with v_mfma_f32_16x16x4_f32 (exact fp32), and adds the 16 result
//           rows into the LDS tile with ds_add_f32.  One coalesced write of the tile at
//           the end: no global atomics, every output row written once.
//   wgrad : a wave owns a run of chunks of one relation and keeps the d_in x d_out
//           gradient block in MFMA accumulators; the K dimension of the MFMA runs over
//           MESSAGES (A = val * X[src]^T, B = G[dst]).  One flush per run.
//
// 64-wide wavefronts throughout: lane = 16*k + m addresses MFMA operand element
// (row m, k-slot k) -- see cdna_hip_programming.md section 3.
#include <stdlib.h>

#include "rgcn_device.h"

namespace {

constexpr int SPMM_WAVES = WG / 64;

// ---- d_in = d_out = 16 (the S1 / hidden-16 fast path).  K-slot permutation f(c,k) = 4k + c, so a
// lane's four B values are ONE 16-byte load of its source row.
template <int U>
struct D16Stage {
  int s[U];      // source row of slot m
  float v[U];    // adjacency value of slot m
  int d[U];      // destination row of slot m
  int r[U];      // relation (same in every lane)
  float4 x[U];   // gathered source-row quarter (features 4k..4k+3)
  float4 w[U];   // W_rel fragment for the four MFMA steps
};

// PACKED: slots come as 8 bytes {src | dst_local << 24, val} and W_rel as pre-swizzled fragments
// (one float4 per lane), i.e. 4 VMEM instructions per chunk instead of 8.
template <int U, bool PACKED>
__global__ __launch_bounds__(WG) void spmm_d16_kernel(
    const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias,
    float *__restrict__ out, const int *__restrict__ p_src, const int *__restrict__ p_dst,
    const float *__restrict__ p_val, const int2 *__restrict__ p_pack, const int *__restrict__ chunk_rel,
    const int4 *__restrict__ units, int n_units, int tile_rows, int n_dst, int relu_out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int u = blockIdx.x * SPMM_WAVES + wave;
  if (u >= n_units) return;  // whole wave; no workgroup barrier is used anywhere below
  const int4 unit = units[u];
  const int t = unit.x;
  float *tile = lds + wave * tile_rows * 16;
  const int row0 = t * tile_rows;
  const int nrows = min(tile_rows, n_dst - row0);
  for (int i = lane; i < nrows * 4; i += 64) reinterpret_cast<float4 *>(tile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

  const int my0 = unit.y, my1 = unit.z;
  const int m = lane & 15, k = lane >> 4;
  const int woff = (4 * k) * 16 + m;

  if (my0 < my1) {
    const int last = my1 - 1;
    // stage 1: slot indices (chunks past the range re-read the last chunk with val = 0)
    auto load_idx = [&](int c, D16Stage<U> &g) {
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int cc = min(c + j, last);
        const int e = cc * RGCN_CHUNK + m;
        if (PACKED) {
          const int2 pk = p_pack[e];
          g.s[j] = pk.x & 0xFFFFFF;
          const int dl = (int)((unsigned)pk.x >> 24);
          g.d[j] = dl == 0xFF ? -1 : row0 + dl;
          g.v[j] = (c + j <= last) ? __builtin_bit_cast(float, pk.y) : 0.f;
        } else {
          g.s[j] = p_src[e];
          const float vv = p_val[e];
          g.v[j] = (c + j <= last) ? vv : 0.f;
          g.d[j] = p_dst[e];
        }
        g.r[j] = chunk_rel[cc];
      }
    };
    // stage 2: row gathers; the W_rel fragments ride along unconditionally (a branch on "relation
    // changed" makes hipcc drain vmcnt to 0 at every join)
    auto gather = [&](D16Stage<U> &g) {
#pragma unroll
      for (int j = 0; j < U; ++j) {
        if (PACKED)       // packed slots: source ids < 2^24 -> a 32-bit byte offset from the uniform base (scalar-base load form)
          g.x[j] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(X) + (((unsigned)g.s[j] << 6) | ((unsigned)k << 4)));
        else
          g.x[j] = *reinterpret_cast<const float4 *>(X + (size_t)g.s[j] * 16 + 4 * k);
        if (PACKED) {
          g.w[j] = reinterpret_cast<const float4 *>(W)[(size_t)g.r[j] * 64 + lane];
        } else {
          const float *wr = W + (size_t)g.r[j] * 256 + woff;
          g.w[j] = make_float4(wr[0], wr[16], wr[32], wr[48]);
        }
      }
    };
    // stage 3: matrix cores, fold equal destinations, one LDS update per segment
    auto compute = [&](const D16Stage<U> &g) {
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const float v = g.v[j];
        const bool live = v != 0.f;
        f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(g.w[j].x, live ? g.x[j].x * v : 0.f, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(g.w[j].y, live ? g.x[j].y * v : 0.f, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(g.w[j].z, live ? g.x[j].z * v : 0.f, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(g.w[j].w, live ? g.x[j].w * v : 0.f, acc[0], 0, 0, 0);
        const bool tail = fold_segments<1>(acc, g.d[j]);
        if (tail) {
          const int dl = g.d[j] - row0;
          f32x4 *p = reinterpret_cast<f32x4 *>(tile + dl * 16 + 4 * (k ^ ((dl >> 2) & 3)));   // swizzled: see tile_swz
          *p += acc[0];
        }
      }
    };
    for (int c = my0; c < my1; c += U) {
      D16Stage<U> A;
      load_idx(c, A);
#pragma unroll
      for (int j = 0; j < U; ++j)  // pin the index loads here: hipcc otherwise sinks them to their first use
        asm volatile("" : "+v"(A.d[j]), "+v"(A.v[j]));
      __builtin_amdgcn_sched_barrier(0);
      gather(A);
      __builtin_amdgcn_sched_barrier(0);
      compute(A);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias && (!(unit.w & RGCN_U_SHARED) || (unit.w & RGCN_U_FIRST))) bv = reinterpret_cast<const float4 *>(bias)[lane & 3];
  float4 *o4 = reinterpret_cast<float4 *>(out + (size_t)row0 * 16);
  if (unit.w & RGCN_U_SHARED) {   // piece of a hub tile: out was zeroed by the launcher, pieces are summed atomically
    for (int i = lane; i < nrows * 4; i += 64) {
      const float4 a = reinterpret_cast<const float4 *>(tile)[tile_swz(i)];
      float *o = reinterpret_cast<float *>(o4 + i);
      atomicAdd(o, a.x + bv.x); atomicAdd(o + 1, a.y + bv.y); atomicAdd(o + 2, a.z + bv.z); atomicAdd(o + 3, a.w + bv.w);
    }
    return;
  }
  for (int i = lane; i < nrows * 4; i += 64) {  // (i & 3) == (lane & 3)
    float4 a = reinterpret_cast<const float4 *>(tile)[tile_swz(i)];
    a.x += bv.x; a.y += bv.y; a.z += bv.z; a.w += bv.w;
    if (relu_out) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
    o4[i] = a;
  }
}

// W[r][f][o] -> fragment order Wp[r][lane = 16k+o][c] = W[r][4k+c][o]  (one float4 per lane and chunk)
__global__ __launch_bounds__(WG) void pack_w16_kernel(const float *__restrict__ W, float *__restrict__ Wp, int n) {
  const int i = blockIdx.x * WG + threadIdx.x;   // index into Wp
  if (i >= n) return;
  const int c = i & 3, o = (i >> 2) & 15, kk = (i >> 6) & 3, r = i >> 8;
  Wp[i] = W[r * 256 + (4 * kk + c) * 16 + o];
}

// ---- d_in = 16 NI, d_out = 16 NJ (NI, NJ <= 4): the hidden-16 scheme over blocks of 16 features.  A lane gathers NI
// quarters of its slot's source row (one float4 per 16-feature block; at d_in = 32 the row is a full 128-byte line), the
// W_rel fragments come pre-swizzled per (input block, output block), the NJ accumulators are folded together and a
// segment updates NJ float4 of its LDS row.  Packed slots only.
// Wp[r][ib][jb][lane = 16k+o][c] = W[r][16 ib + 4k + c][16 jb + o]
__global__ __launch_bounds__(WG) void pack_w_blocks_kernel(const float *__restrict__ W, float *__restrict__ Wp, int n,
                                                           int NI, int NJ) {
  const int i = blockIdx.x * WG + threadIdx.x;   // index into Wp
  if (i >= n) return;
  const int c = i & 3, o = (i >> 2) & 15, kk = (i >> 6) & 3;
  const int blk = i >> 8, jb = blk % NJ, ib = (blk / NJ) % NI, r = blk / (NI * NJ);
  Wp[i] = W[((size_t)r * 16 * NI + 16 * ib + 4 * kk + c) * (16 * NJ) + 16 * jb + o];
}

template <int NI, int NJ, int U>
__global__ __launch_bounds__(WG) void spmm_wide_kernel(
    const float *__restrict__ X, const float *__restrict__ Wp, const float *__restrict__ bias,
    float *__restrict__ out, const int2 *__restrict__ p_pack, const int *__restrict__ chunk_rel,
    const int4 *__restrict__ units, int n_units, int tile_rows, int n_dst, int relu_out) {
  constexpr int DI = 16 * NI, DO = 16 * NJ;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int u = blockIdx.x * SPMM_WAVES + wave;
  if (u >= n_units) return;
  const int4 unit = units[u];
  float *tile = lds + wave * tile_rows * DO;
  const int row0 = unit.x * tile_rows;
  const int nrows = min(tile_rows, n_dst - row0);
  for (int i = lane; i < nrows * (DO / 4); i += 64) reinterpret_cast<float4 *>(tile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int my0 = unit.y, my1 = unit.z;
  const int m = lane & 15, k = lane >> 4;
  if (my0 < my1) {
    const int last = my1 - 1;
    for (int c0 = my0; c0 < my1; c0 += U) {
      int s[U], d[U], r[U];
      float v[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int cc = min(c0 + j, last);
        const int2 pk = p_pack[cc * RGCN_CHUNK + m];
        s[j] = pk.x & 0xFFFFFF;
        const int dl = (int)((unsigned)pk.x >> 24);
        d[j] = dl == 0xFF ? -1 : row0 + dl;
        v[j] = (c0 + j <= last) ? __builtin_bit_cast(float, pk.y) : 0.f;
        r[j] = chunk_rel[cc];
      }
#pragma unroll
      for (int j = 0; j < U; ++j) asm volatile("" : "+v"(d[j]), "+v"(v[j]));   // keep the index loads up here
      __builtin_amdgcn_sched_barrier(0);
      float4 x[U][NI], w[U][NI][NJ];
#pragma unroll
      for (int j = 0; j < U; ++j) {
#pragma unroll
        for (int ib = 0; ib < NI; ++ib) x[j][ib] = *reinterpret_cast<const float4 *>(X + (size_t)s[j] * DI + 16 * ib + 4 * k);
        const float4 *wr = reinterpret_cast<const float4 *>(Wp) + (size_t)r[j] * (NI * NJ * 64) + lane;
#pragma unroll
        for (int ib = 0; ib < NI; ++ib)
#pragma unroll
          for (int jb = 0; jb < NJ; ++jb) w[j][ib][jb] = wr[(ib * NJ + jb) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const bool live = v[j] != 0.f;
        f32x4 acc[NJ];
#pragma unroll
        for (int jb = 0; jb < NJ; ++jb) acc[jb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ib = 0; ib < NI; ++ib) {
          const float bx = live ? x[j][ib].x * v[j] : 0.f, by = live ? x[j][ib].y * v[j] : 0.f;
          const float bz = live ? x[j][ib].z * v[j] : 0.f, bw = live ? x[j][ib].w * v[j] : 0.f;
#pragma unroll
          for (int jb = 0; jb < NJ; ++jb) {
            acc[jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j][ib][jb].x, bx, acc[jb], 0, 0, 0);
            acc[jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j][ib][jb].y, by, acc[jb], 0, 0, 0);
            acc[jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j][ib][jb].z, bz, acc[jb], 0, 0, 0);
            acc[jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j][ib][jb].w, bw, acc[jb], 0, 0, 0);
          }
        }
        if (fold_segments<NJ>(acc, d[j])) {
          float *row = tile + (d[j] - row0) * DO + 4 * k;
#pragma unroll
          for (int jb = 0; jb < NJ; ++jb) *reinterpret_cast<f32x4 *>(row + 16 * jb) += acc[jb];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const bool shared = (unit.w & RGCN_U_SHARED) != 0, add_bias = bias && (!shared || (unit.w & RGCN_U_FIRST));
  float4 *o4 = reinterpret_cast<float4 *>(out + (size_t)row0 * DO);
  for (int i = lane; i < nrows * (DO / 4); i += 64) {
    float4 a = reinterpret_cast<const float4 *>(tile)[i];
    if (add_bias) {
      const float4 bv = reinterpret_cast<const float4 *>(bias)[i % (DO / 4)];
      a.x += bv.x; a.y += bv.y; a.z += bv.z; a.w += bv.w;
    }
    if (shared) {   // piece of a hub tile: out was zeroed by the launcher, pieces are summed atomically
      float *o = reinterpret_cast<float *>(o4 + i);
      atomicAdd(o, a.x); atomicAdd(o + 1, a.y); atomicAdd(o + 2, a.z); atomicAdd(o + 3, a.w);
    } else {
      if (relu_out) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
      o4[i] = a;
    }
  }
}

// ---- sparse-bucket path, pass 1: relation-major chunks (dense), transformed messages scattered to their slot in
// destination-major order.  One wave per work item (<= 64 chunks of ONE relation): the W fragment is loaded once.
template <int U>
__global__ __launch_bounds__(WG) void spmm_scatter_d16_kernel(
    const float *__restrict__ X, const float *__restrict__ Wp, float *__restrict__ Y, const int *__restrict__ p_src,
    const float *__restrict__ p_val, const int *__restrict__ p_pos, const int *__restrict__ chunk_rel,
    const int2 *__restrict__ items, int n_items) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int item = blockIdx.x * (WG / 64) + wave;
  if (item >= n_items) return;
  const int2 range = items[item];
  if (range.x >= range.y) return;          // padding of a work list sized by an upper bound (rgcn_dev_plan_finish_nosync)
  const int r = __builtin_amdgcn_readfirstlane(chunk_rel[range.x]);
  const int m = lane & 15, k = lane >> 4;
  const float4 w = reinterpret_cast<const float4 *>(Wp)[(size_t)r * 64 + lane];
  const int last = range.y - 1;
  for (int c = range.x; c < range.y; c += U) {
    int s[U], pos[U];
    float v[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int cc = min(c + j, last);
      const int e = cc * RGCN_CHUNK + m;
      s[j] = p_src[e];
      pos[j] = p_pos ? p_pos[e] : e;            // no position list: rows stay in relation-major slot order (sequential writes)
      const float vv = p_val[e];
      v[j] = (c + j <= last) ? vv : 0.f;
    }
#pragma unroll
    for (int j = 0; j < U; ++j) asm volatile("" : "+v"(s[j]), "+v"(pos[j]), "+v"(v[j]));
    __builtin_amdgcn_sched_barrier(0);
    float4 x[U];
#pragma unroll
    for (int j = 0; j < U; ++j) x[j] = *reinterpret_cast<const float4 *>(X + (size_t)s[j] * 16 + 4 * k);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const float vv = v[j];
      const bool live = vv != 0.f;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, live ? x[j].x * vv : 0.f, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, live ? x[j].y * vv : 0.f, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, live ? x[j].z * vv : 0.f, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, live ? x[j].w * vv : 0.f, acc, 0, 0, 0);
      if (live) *reinterpret_cast<f32x4 *>(Y + (size_t)pos[j] * 16 + 4 * k) = acc;   // lane 16q+m: features 4q..4q+3 of slot m
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- pass 2: out[row] = bias + sum of the row's (contiguous) transformed messages.  4 lanes per row (16 B each).
__global__ __launch_bounds__(WG) void segment_sum_d16_kernel(const float *__restrict__ Y, const int *__restrict__ rowptr,
                                                             const float *__restrict__ bias, float *__restrict__ out,
                                                             long long n_rows, int relu_out) {
  const int q = threadIdx.x & 3;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) bv = reinterpret_cast<const float4 *>(bias)[q];
  for (long long row = ((long long)blockIdx.x * WG + threadIdx.x) >> 2; row < n_rows; row += ((long long)gridDim.x * WG) >> 2) {
    const int e0 = rowptr[row], e1 = rowptr[row + 1];
    float4 a = bv;
    for (int e = e0; e < e1; ++e) {
      const float4 y = *reinterpret_cast<const float4 *>(Y + (size_t)e * 16 + 4 * q);
      a.x += y.x; a.y += y.y; a.z += y.z; a.w += y.w;
    }
    if (relu_out) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
    *reinterpret_cast<float4 *>(out + (size_t)row * 16 + 4 * q) = a;
  }
}

// a, b += the rows Y[perm[e0 .. e1), 4 q .. 4 q + 4): four row reads in flight, and the NEXT four indices loaded before this trip's rows are
// used -- a trip of the loop is one round trip (the rows), not two (indices, then rows).  Entries past the end re-read the last one
// (unconditional loads) and add nothing.
__device__ __forceinline__ void gather_sum4(const float *__restrict__ Y, const int *__restrict__ perm, int e0, int e1, int q, float4 &a,
                                            float4 &b) {
  if (e1 <= e0) return;
  int p[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) p[j] = perm[min(e0 + j, e1 - 1)];
  for (int e = e0; e < e1; e += 4) {
    float4 y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = *reinterpret_cast<const float4 *>(Y + (size_t)p[j] * 16 + 4 * q);
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = perm[min(e + 4 + j, e1 - 1)];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (e + j < e1) {
        float4 &t = (j & 1) ? b : a;
        t.x += y[j].x; t.y += y[j].y; t.z += y[j].z; t.w += y[j].w;
      }
  }
}

// ---- pass 2 when pass 1 wrote its rows in slot order: the rows of a destination are gathered through `perm`
// (destination-major position -> slot).  4 lanes per row, 4 row reads in flight.
__global__ __launch_bounds__(WG) void segment_gather_sum_d16_kernel(const float *__restrict__ Y, const int *__restrict__ perm,
                                                                    const int *__restrict__ rowptr, const float *__restrict__ bias,
                                                                    float *__restrict__ out, long long n_rows, int relu_out) {
  const int q = threadIdx.x & 3;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) bv = reinterpret_cast<const float4 *>(bias)[q];
  for (long long row = ((long long)blockIdx.x * WG + threadIdx.x) >> 2; row < n_rows; row += ((long long)gridDim.x * WG) >> 2) {
    const int e0 = rowptr[row], e1 = rowptr[row + 1];
    float4 a = bv, b = make_float4(0.f, 0.f, 0.f, 0.f);
    gather_sum4(Y, perm, e0, e1, q, a, b);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    if (relu_out) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
    *reinterpret_cast<float4 *>(out + (size_t)row * 16 + 4 * q) = a;
  }
}

// The same over work UNITS {row, first entry, end entry, flags}: one unit per row, hub rows cut into pieces whose partial sums are
// added to `out` with fp32 atomics (RGCN_U_SHARED; `out` zeroed by the launcher; the RGCN_U_FIRST piece adds the bias).  Without
// the cut a row with 184 k entries (Zipf(0.9) AM-shaped graph) is one 4-lane loop: 50 ms per launch against 0.5.
__global__ __launch_bounds__(WG) void segment_gather_sum_units_d16_kernel(const float *__restrict__ Y, const int *__restrict__ perm,
                                                                          const int4 *__restrict__ units, const float *__restrict__ bias,
                                                                          float *__restrict__ out, long long n_units, int relu_out) {
  const int q = threadIdx.x & 3;
  for (long long u = ((long long)blockIdx.x * WG + threadIdx.x) >> 2; u < n_units; u += ((long long)gridDim.x * WG) >> 2) {
    const int4 unit = units[u];
    const bool shared = unit.w & RGCN_U_SHARED;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias && (!shared || (unit.w & RGCN_U_FIRST))) a = reinterpret_cast<const float4 *>(bias)[q];
    gather_sum4(Y, perm, unit.y, unit.z, q, a, b);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    float *o = out + (size_t)unit.x * 16 + 4 * q;
    if (shared) {
      atomicAdd(o, a.x); atomicAdd(o + 1, a.y); atomicAdd(o + 2, a.z); atomicAdd(o + 3, a.w);
    } else {
      if (relu_out) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
      *reinterpret_cast<float4 *>(o) = a;
    }
  }
}

// ---- any d_in / d_out.  NJT = 16-wide output column tiles kept in accumulators per pass over d_in;
// the LDS tile rows are padded to ldt = round_up(d_out, 4) floats so every update is one aligned b128.
template <int NJT>
__global__ __launch_bounds__(WG) void spmm_generic_kernel(
    const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias,
    float *__restrict__ out, const int *__restrict__ p_src, const int *__restrict__ p_dst,
    const float *__restrict__ p_val, const int *__restrict__ chunk_rel, const int4 *__restrict__ units,
    int n_units, int tile_rows, int n_dst, int d_in, int d_out, int ldt, int relu_out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int u = blockIdx.x * SPMM_WAVES + wave;
  if (u >= n_units) return;
  const int4 unit = units[u];
  const int t = unit.x;
  float *tile = lds + (size_t)wave * tile_rows * ldt;
  const int row0 = t * tile_rows;
  const int nrows = min(tile_rows, n_dst - row0);
  for (int i = lane; i < nrows * ldt; i += 64) tile[i] = 0.f;

  const int my0 = unit.y, my1 = unit.z;
  const int m = lane & 15, k = lane >> 4;
  for (int c = my0; c < my1; ++c) {
    const int r = __builtin_amdgcn_readfirstlane(chunk_rel[c]);
    const int e = c * RGCN_CHUNK + m;
    const int s = p_src[e];
    const float v = p_val[e];
    const int dst = p_dst[e];
    const bool live = v != 0.f;
    const float *xrow = X + (size_t)s * d_in;
    const float *wr = W + (size_t)r * d_in * d_out;
    for (int jg = 0; jg < d_out; jg += 16 * NJT) {
      f32x4 acc[NJT];
#pragma unroll
      for (int jt = 0; jt < NJT; ++jt) acc[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int kc = 0; kc < d_in; kc += 4) {
        const int f = kc + k;
        const bool fin = f < d_in;
        const float b = (fin && live) ? xrow[f] * v : 0.f;   // B: lane 16k+m -> feature f of slot m
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
          const int col = jg + jt * 16 + m;                   // A: lane 16k+o -> W[f][o]
          const float a = (fin && col < d_out) ? wr[(size_t)f * d_out + col] : 0.f;
          acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[jt], 0, 0, 0);
        }
      }
      const bool tail = fold_segments<NJT>(acc, dst);
      if (tail) {
        float *prow = tile + (size_t)(dst - row0) * ldt;
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
          const int col = jg + jt * 16 + 4 * k;               // D: lane 16q+m -> features 4q..4q+3 of slot m
          if (col < ldt) *reinterpret_cast<f32x4 *>(prow + col) += acc[jt];
        }
      }
    }
  }
  float *o = out + (size_t)row0 * d_out;
  const bool shared = unit.w & RGCN_U_SHARED;
  const bool add_bias = bias && (!shared || (unit.w & RGCN_U_FIRST));
  for (int i = lane; i < nrows * d_out; i += 64) {
    const int rr = i / d_out, cc = i - rr * d_out;
    float a = tile[rr * ldt + cc] + (add_bias ? bias[cc] : 0.f);
    if (shared) { atomicAdd(&o[i], a); continue; }
    if (relu_out) a = fmaxf(a, 0.f);
    o[i] = a;
  }
}

// ------------------------------------------------------------------ weight gradient, any d
// One wave per work item (a run of chunks with one relation).  grid.y enumerates
// (row-tile group, column-tile group) blocks of dW[r].
template <int NIT, int NJT>
__global__ __launch_bounds__(WG) void wgrad_generic_kernel(
    const float *__restrict__ X, const float *__restrict__ G, float *__restrict__ dW,
    const int *__restrict__ p_src, const int *__restrict__ p_dst, const float *__restrict__ p_val,
    const int *__restrict__ chunk_rel, const int2 *__restrict__ items, int n_items, int d_in, int d_out,
    int n_jgroups) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int item = blockIdx.x * (WG / 64) + wave;
  if (item >= n_items) return;
  const int2 range = items[item];
  if (range.x >= range.y) return;          // padding of a work list sized by an upper bound (rgcn_dev_plan_finish_nosync)
  const int r = __builtin_amdgcn_readfirstlane(chunk_rel[range.x]);
  const int i0 = (blockIdx.y / n_jgroups) * 16 * NIT;
  const int j0 = (blockIdx.y % n_jgroups) * 16 * NJT;
  const int m = lane & 15, k = lane >> 4;
  f32x4 acc[NIT][NJT];
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) acc[it][jt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int c = range.x; c < range.y; ++c) {
#pragma unroll
    for (int step = 0; step < 4; ++step) {  // 4 messages per MFMA K-step
      const int e = c * RGCN_CHUNK + 4 * step + k;
      const int s = p_src[e];
      const int d = p_dst[e];
      const float v = p_val[e];
      const bool live = v != 0.f;
      float a[NIT], b[NJT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int f = i0 + it * 16 + m;
        a[it] = (live && f < d_in) ? X[(size_t)s * d_in + f] * v : 0.f;
      }
#pragma unroll
      for (int jt = 0; jt < NJT; ++jt) {
        const int col = j0 + jt * 16 + m;
        b[jt] = (live && col < d_out) ? G[(size_t)d * d_out + col] : 0.f;
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt)
          acc[it][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[it], b[jt], acc[it][jt], 0, 0, 0);
    }
  }
  float *wr = dW + (size_t)r * d_in * d_out;
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) {
      const int col = j0 + jt * 16 + m;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = i0 + it * 16 + 4 * k + q;
        if (row < d_in && col < d_out) atomicAdd(&wr[(size_t)row * d_out + col], acc[it][jt][q]);
      }
    }
}

// ------------------------------------------------------------------ weight gradient, d_in = d_out = 16
// v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 outer products per instruction, one per MESSAGE:
// lane 4b+i supplies A_b[i], lane 4b+j supplies B_b[j], D_b[i][j] lands in lane 4b+j, register i.
// Lane 4b+q loads ONE 16-byte quarter of message b's source row and of its gradient row, so the
// sixteen (c,c') register pairs cover the whole 16x16 outer product:
//     acc[c][c'] (block b, reg i, lane-col j)  +=  val * X[src_b][4i+c] * G[dst_b][4j+c']
// The 16 blocks keep accumulating over all chunks of the work item and are summed once at the end.
template <int U>
__global__ __launch_bounds__(WG) void wgrad_d16_kernel(
    const float *__restrict__ X, const float *__restrict__ G, float *__restrict__ dW,
    const int *__restrict__ p_src, const int *__restrict__ p_dst, const float *__restrict__ p_val,
    const int *__restrict__ chunk_rel, const int2 *__restrict__ items, int n_items) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int item = blockIdx.x * (WG / 64) + wave;
  if (item >= n_items) return;
  const int2 range = items[item];
  if (range.x >= range.y) return;          // padding of a work list sized by an upper bound (rgcn_dev_plan_finish_nosync)
  const int r = __builtin_amdgcn_readfirstlane(chunk_rel[range.x]);
  const int b = lane >> 2, q = lane & 3;
  f32x4 acc[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) acc[c][cc] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int last = range.y - 1;
  for (int c0 = range.x; c0 < range.y; c0 += U) {
    int s[U], d[U];
    float v[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int cc = min(c0 + j, last);
      const int e = cc * RGCN_CHUNK + b;
      s[j] = p_src[e];
      d[j] = p_dst[e];
      const float vv = p_val[e];
      v[j] = (c0 + j <= last) ? vv : 0.f;
    }
#pragma unroll
    for (int j = 0; j < U; ++j) asm volatile("" : "+v"(s[j]), "+v"(d[j]), "+v"(v[j]));
    __builtin_amdgcn_sched_barrier(0);
    float4 x[U], g[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      x[j] = *reinterpret_cast<const float4 *>(X + (size_t)s[j] * 16 + 4 * q);
      g[j] = *reinterpret_cast<const float4 *>(G + (size_t)max(d[j], 0) * 16 + 4 * q);   // pads: dst = -1, val = 0
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const bool live = v[j] != 0.f;
      const float xa[4] = {live ? x[j].x * v[j] : 0.f, live ? x[j].y * v[j] : 0.f, live ? x[j].z * v[j] : 0.f,
                           live ? x[j].w * v[j] : 0.f};
      const float gb[4] = {live ? g[j].x : 0.f, live ? g[j].y : 0.f, live ? g[j].z : 0.f, live ? g[j].w : 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          acc[c][cc] = __builtin_amdgcn_mfma_f32_4x4x1f32(xa[c], gb[cc], acc[c][cc], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // sum the 16 blocks: lanes with equal (lane & 3); rows of 16 lanes by DPP rotates, then across rows
  float *wr = dW + (size_t)r * 256;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float a = acc[c][cc][i];
        a += dpp_f<0x120 + 4>(0.f, a);   // row_ror:4
        a += dpp_f<0x120 + 8>(0.f, a);   // row_ror:8
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        if (lane < 4) atomicAdd(&wr[(4 * i + c) * 16 + 4 * q + cc], a);
      }
}

// ------------------------------------------------------------------ weight gradient, tile-major, d = 16
// Walks the forward plan.  Work item = (group of RG consecutive relations, range of tiles); the chunks
// of a relation group are contiguous inside every tile.  X rows (random) are gathered in the natural
// layout (lane 16q+m: quarter q of slot m's row) and transposed through a 1 KiB LDS scratch into the
// MFMA operand layout (K over messages).  G rows are tile-local -- every row of the tile is reused by
// ~deg messages -- so they are read straight from L1/L2 in operand layout (4 rows x 64 B per load).
// dW[rel] += (val X[src])^T G[dst] accumulates in RG x 4 accumulator registers per lane.
template <int RG, int U>
__global__ __launch_bounds__(WG) void wgrad_tiled_d16_kernel(
    const float *__restrict__ X, const float *__restrict__ G, float *__restrict__ dW,
    const int *__restrict__ p_src, const int *__restrict__ p_dst, const float *__restrict__ p_val,
    const int *__restrict__ chunk_rel, const int *__restrict__ run_ptr, int n_tiles, int R, int tiles_per_item,
    int n_groups, int n_items) {
  __shared__ __attribute__((aligned(16))) float lds_all[(WG / 64) * (512 + 64)];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  // XCD-aware work mapping (workgroup b runs on XCD b % 8, each XCD has its own L2): all the relation items of one
  // tile block read the same G rows, so they go to ONE XCD -- the rows are fetched into one L2 instead of eight.
  const int xcd = blockIdx.x & 7, local = (blockIdx.x >> 3) * (WG / 64) + wave;
  const int grp = local % n_groups, tb = (local / n_groups) * 8 + xcd;
  const int t0 = tb * tiles_per_item, t1 = min(n_tiles, t0 + tiles_per_item);
  if (t0 >= n_tiles) return;
  const int r0 = grp * RG, r1 = min(R, r0 + RG);
  float *xs = lds_all + wave * (512 + 64);                          // 2 x scratch 16x16 | dst rows of U chunks
  int *dl = reinterpret_cast<int *>(xs + 512);
  const int m = lane & 15, kq = lane >> 4;
  f32x4 acc[RG];
#pragma unroll
  for (int i = 0; i < RG; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // The item's chunks live in (t1 - t0) separate runs of one or two chunks each.  Fetch all run bounds at once and
  // walk the chunks as ONE flat sequence in groups of U: per item that is 1 + 2 * ceil(chunks / U) dependent memory
  // latencies instead of 3 per tile.
  constexpr int MAXT = 8;
  int run_b[MAXT], run_n[MAXT], total = 0;
  {
    const int tl = min(lane, t1 - t0 - 1);
    const int vb = run_ptr[(size_t)(t0 + tl) * (R + 1) + r0], ve = run_ptr[(size_t)(t0 + tl) * (R + 1) + r1];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int bb = __builtin_amdgcn_readlane(vb, t), ee = __builtin_amdgcn_readlane(ve, t);
      run_b[t] = bb;
      run_n[t] = (t < t1 - t0) ? ee - bb : 0;
      total += run_n[t];
    }
  }
  auto chunk_at = [&](int q) {   // q-th chunk of the item (wave-uniform); q >= total -> the last chunk
    q = min(q, total - 1);
    int c = run_b[0];
    bool found = false;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      if (!found && q < run_n[t]) { c = run_b[t] + q; found = true; }
      if (!found) q -= run_n[t];
    }
    return c;
  };
  if (total > 0) {
    for (int q0 = 0; q0 < total; q0 += U) {
      int s[U], d[U], rr[U];
      float v[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int cc = chunk_at(q0 + j);
        const int e = cc * RGCN_CHUNK + m;
        s[j] = p_src[e];
        d[j] = p_dst[e];
        const float vv = p_val[e];
        v[j] = (q0 + j < total) ? vv : 0.f;
        rr[j] = (RG == 1) ? r0 : chunk_rel[cc];
      }
#pragma unroll
      for (int j = 0; j < U; ++j) asm volatile("" : "+v"(s[j]), "+v"(d[j]), "+v"(v[j]), "+v"(rr[j]));
      __builtin_amdgcn_sched_barrier(0);
      // destination rows into MFMA K-slot order (slot 4*t4 + kq for step t4) through the LDS scratch, for all U chunks
      asm volatile("" ::: "memory");
#pragma unroll
      for (int j = 0; j < U; ++j)
        if (kq == 0) dl[j * 16 + m] = d[j];
      asm volatile("" ::: "memory");
      int dmu[U][4];
#pragma unroll
      for (int j = 0; j < U; ++j)
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) dmu[j][t4] = dl[j * 16 + 4 * t4 + kq];
      __builtin_amdgcn_sched_barrier(0);
      // all loads of the group in flight together: U random X row gathers + 4U tile-local G operand loads (L1/L2)
      float4 x[U];
      float gl[U][4];
#pragma unroll
      for (int j = 0; j < U; ++j) x[j] = *reinterpret_cast<const float4 *>(X + (size_t)s[j] * 16 + 4 * kq);
#pragma unroll
      for (int j = 0; j < U; ++j)
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) gl[j][t4] = G[(size_t)max(dmu[j][t4], 0) * 16 + m];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const bool live = v[j] != 0.f;
        const f32x4 xv = {live ? x[j].x * v[j] : 0.f, live ? x[j].y * v[j] : 0.f, live ? x[j].z * v[j] : 0.f,
                          live ? x[j].w * v[j] : 0.f};
        float *xj = xs + (j & 1) * 256;                              // two scratch tiles: no wait on the previous chunk's reads
        asm volatile("" ::: "memory");   // LDS written as vectors, read as scalars: no compiler reordering
        *reinterpret_cast<f32x4 *>(xj + m * 16 + 4 * kq) = xv;      // xs[slot m][4kq..4kq+3]
        asm volatile("" ::: "memory");
        float av[4], bv[4];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
          const int mu = 4 * t4 + kq;                                // message carried by K-slot kq at step t4
          av[t4] = xj[mu * 16 + m];                                  // A[i = m][k] = val * X[src_mu][m]
          bv[t4] = dmu[j][t4] < 0 ? 0.f : gl[j][t4];                 // B[k][j = m] = G[dst_mu][m]; pads: dst = -1
        }
        const int rho = __builtin_amdgcn_readfirstlane(rr[j]) - r0;
#define RGCN_ACC_CASE(N)                                                                       \
  case N:                                                                                      \
    if (N < RG) {                                                                              \
      acc[N < RG ? N : 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], acc[N < RG ? N : 0], 0, 0, 0); \
      acc[N < RG ? N : 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], acc[N < RG ? N : 0], 0, 0, 0); \
      acc[N < RG ? N : 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], acc[N < RG ? N : 0], 0, 0, 0); \
      acc[N < RG ? N : 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], acc[N < RG ? N : 0], 0, 0, 0); \
    }                                                                                          \
    break;
        if (RG == 1) {
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], acc[0], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], acc[0], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], acc[0], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], acc[0], 0, 0, 0);
        } else {
          switch (rho) {
            RGCN_ACC_CASE(0) RGCN_ACC_CASE(1) RGCN_ACC_CASE(2) RGCN_ACC_CASE(3) RGCN_ACC_CASE(4) RGCN_ACC_CASE(5)
            RGCN_ACC_CASE(6) RGCN_ACC_CASE(7) RGCN_ACC_CASE(8) RGCN_ACC_CASE(9) RGCN_ACC_CASE(10) RGCN_ACC_CASE(11)
            RGCN_ACC_CASE(12) RGCN_ACC_CASE(13) RGCN_ACC_CASE(14) RGCN_ACC_CASE(15)
            default: break;
          }
        }
#undef RGCN_ACC_CASE
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // D: lane 16q+j holds rows 4q..4q+3 (input feature), column j (output feature)
#pragma unroll
  for (int i = 0; i < RG; ++i)
    if (r0 + i < R) {
      float *wr = dW + (size_t)(r0 + i) * 256 + (4 * kq) * 16 + m;
      atomicAdd(wr, acc[i][0]);
      atomicAdd(wr + 16, acc[i][1]);
      atomicAdd(wr + 32, acc[i][2]);
      atomicAdd(wr + 48, acc[i][3]);
    }
}

// ------------------------------------------------------------------ featureless layer
// Forward: same wave-owned-tile scheme as spmm, without the matrix product: lane 16q+m carries
// features jb+4q..jb+4q+3 of slot m's table row.
__global__ __launch_bounds__(WG) void featureless_fwd_kernel(
    const float *__restrict__ table, const float *__restrict__ bias, float *__restrict__ out,
    const int *__restrict__ p_src, const int *__restrict__ p_dst, const float *__restrict__ p_val,
    const int *__restrict__ chunk_rel, const int4 *__restrict__ units, int n_units, int tile_rows, int n_dst,
    long long n_src, int d, int ldt) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int u = blockIdx.x * SPMM_WAVES + wave;
  if (u >= n_units) return;
  const int4 unit = units[u];
  const int t = unit.x;
  float *tile = lds + (size_t)wave * tile_rows * ldt;
  const int row0 = t * tile_rows;
  const int nrows = min(tile_rows, n_dst - row0);
  for (int i = lane; i < nrows * ldt; i += 64) tile[i] = 0.f;
  const int my0 = unit.y, my1 = unit.z;
  const int m = lane & 15, q = lane >> 4;
  const bool vec4 = (d & 3) == 0;
  for (int c = my0; c < my1; ++c) {
    const long long r = __builtin_amdgcn_readfirstlane(chunk_rel[c]);
    const int e = c * RGCN_CHUNK + m;
    const float v = p_val[e];
    const int dst = p_dst[e];
    const float *row = table + (size_t)(r * n_src + p_src[e]) * d;
    for (int jb = 0; jb < d; jb += 16) {
      const int col = jb + 4 * q;
      f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
      if (v != 0.f) {
        if (vec4) {
          if (col < d) {
            const float4 x = *reinterpret_cast<const float4 *>(row + col);
            acc[0] = f32x4{x.x * v, x.y * v, x.z * v, x.w * v};
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (col + i < d) acc[0][i] = row[col + i] * v;
        }
      }
      const bool tail = fold_segments<1>(acc, dst);
      if (tail && col < ldt) *reinterpret_cast<f32x4 *>(tile + (size_t)(dst - row0) * ldt + col) += acc[0];
    }
  }
  float *o = out + (size_t)row0 * d;
  const bool shared = unit.w & RGCN_U_SHARED;
  const bool add_bias = bias && (!shared || (unit.w & RGCN_U_FIRST));
  for (int i = lane; i < nrows * d; i += 64) {
    const int rr = i / d, cc = i - rr * d;
    const float a = tile[rr * ldt + cc] + (add_bias ? bias[cc] : 0.f);
    if (shared) atomicAdd(&o[i], a);
    else o[i] = a;
  }
}

__global__ __launch_bounds__(WG) void featureless_wgrad_kernel(
    const float *__restrict__ G, float *__restrict__ dtable, const int *__restrict__ p_src,
    const int *__restrict__ p_dst, const float *__restrict__ p_val, const int *__restrict__ chunk_rel,
    long long n_chunks, long long n_src, int d, int lpr) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rpi = 64 / lpr;
  const int sub = lane / lpr, jj = lane % lpr;
  const long long wstride = (long long)gridDim.x * (WG / 64);
  for (long long c = (long long)blockIdx.x * (WG / 64) + wave; c < n_chunks; c += wstride) {
    const long long r = chunk_rel[c];
    for (int m0 = 0; m0 < RGCN_CHUNK; m0 += rpi) {
      const int mm = m0 + sub;
      if (mm >= RGCN_CHUNK) continue;
      const long long e = c * RGCN_CHUNK + mm;
      const float v = p_val[e];
      if (v == 0.f) continue;
      const float *g = G + (size_t)p_dst[e] * d;
      float *row = dtable + (size_t)(r * n_src + p_src[e]) * d;
      for (int j = jj; j < d; j += lpr) atomicAdd(&row[j], v * g[j]);
    }
  }
}

// Diagonal-weight layer (reference layers.py:289-292: einsum('ij,kj->kij') + torch.mm(adj, fw)): out[row, :] = bias + the sum over
// the row's messages of val * X[src, :] * w[rel, :] -- d multiplies per message instead of the d x d product of the embedded
// diagonal, and nothing materialised.  Destination-major CSR (no (tile, relation) buckets: with hundreds of relations and
// wide rows those are nearly all padding); work units = rows, long rows cut into pieces (RGCN_U_SHARED: fp32 atomics into
// the zeroed output).  `lr` lanes per unit = lr / lpm messages in flight x lpm lanes x float4; unrolled by two.
// TABLE (featureless layer on a graph with sparse (tile, relation) buckets): X is the weight table [R][n_src][d] itself, the message's
// row is table[rel][src] and there is no w -- out[row, :] = bias + the sum of val * table[rel, src, :].
template <bool TABLE>
__global__ __launch_bounds__(WG) void diag_csr_kernel(
    const float *__restrict__ X, const float *__restrict__ w, const float *__restrict__ bias, float *__restrict__ out,
    const int4 *__restrict__ units, long long n_units, const int *__restrict__ e_src, const int *__restrict__ e_rel,
    const float *__restrict__ e_val, int d, int lpm, int lr, long long n_src, int relu_out) {
  const long long u = ((long long)blockIdx.x * WG + threadIdx.x) / lr;
  const int sub = threadIdx.x % lr, g = sub / lpm, j = sub % lpm, gpr = lr / lpm;
  const bool on = u < n_units;
  const int4 unit = on ? units[u] : int4{0, 0, 0, 0};
  const int e1 = unit.z;
  const bool vec = (d & 3) == 0;      // (as a template parameter this kernel measured 6-10 % slower)
  const bool shared = unit.w & RGCN_U_SHARED;
  const bool add_bias = bias && (!shared || (unit.w & RGCN_U_FIRST));
  for (int f0 = 0; f0 < d; f0 += 4 * lpm) {
    const int f = f0 + 4 * j;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    if (f < d)
      for (int e = unit.y + g; e < e1; e += 2 * gpr) {
        const int eb = e + gpr;
        const bool hb = eb < e1;
        const float va = e_val[e], vb = hb ? e_val[eb] : 0.f;
        const long long ra = e_rel[e], rb = e_rel[hb ? eb : e];
        const float *xa = X + ((TABLE ? ra * n_src : 0) + e_src[e]) * (size_t)d + f;
        const float *xb = X + ((TABLE ? rb * n_src : 0) + e_src[hb ? eb : e]) * (size_t)d + f;
        const float *wa = TABLE ? xa : w + (size_t)ra * d + f, *wb = TABLE ? xb : w + (size_t)rb * d + f;
        if (vec) {
          const f32x4 x0 = *reinterpret_cast<const f32x4 *>(xa), x1 = *reinterpret_cast<const f32x4 *>(xb);
          if (TABLE) {
            a += x0 * va;
            b += x1 * vb;
          } else {
            const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wa), w1 = *reinterpret_cast<const f32x4 *>(wb);
            a += x0 * w0 * va;
            b += x1 * w1 * vb;
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (f + q < d) { a[q] += xa[q] * (TABLE ? 1.f : wa[q]) * va; b[q] += xb[q] * (TABLE ? 1.f : wb[q]) * vb; }
        }
      }
    a += b;
    for (int s = lpm; s < lr; s *= 2) {          // sum over the message groups of the unit (same trip count in every lane)
#pragma unroll
      for (int q = 0; q < 4; ++q) a[q] += __shfl_xor(a[q], s, 64);
    }
    if (on && g == 0 && f < d) {
      float *o = out + (size_t)unit.x * d + f;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (add_bias && f + q < d) a[q] += bias[f + q];
      if (relu_out && !shared) {                  // (the launcher refuses relu_out with hub rows cut into shared pieces)
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = fmaxf(a[q], 0.f);
      }
      if (vec && !shared) {
        *reinterpret_cast<f32x4 *>(o) = a;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (f + q < d) {
            if (shared) atomicAdd(o + q, a[q]);
            else o[q] = a[q];
          }
      }
    }
  }
}

// Featureless layer, gradient of the weight table on the destination-major CSR: dtable[rel, src, :] += val * G[row, :] for every
// message of row `row` -- `lpm` lanes per message (a float4 each), WG / lpm messages of consecutive rows' entries in flight; one
// lane group per CSR ENTRY (not per padded plan slot: on a graph with sparse (tile, relation) buckets the tile plan is 10-20x
// padding).  fp32 atomics: several messages may share (relation, source).
__global__ __launch_bounds__(WG) void featureless_csr_wgrad_kernel(
    const float *__restrict__ G, float *__restrict__ dtable, const int *__restrict__ rowptr, const int *__restrict__ e_row,
    const int *__restrict__ e_src, const int *__restrict__ e_rel, const float *__restrict__ e_val, long long n_entries, long long n_rows,
    long long n_src, int d, int lpm) {
  const long long e = ((long long)blockIdx.x * WG + threadIdx.x) / lpm;
  const int j = threadIdx.x % lpm;
  if (e >= n_entries) return;
  // the entry's row: e_row when the caller has it, else a binary search in rowptr
  long long row;
  if (e_row) {
    row = e_row[e];
  } else {
    long long lo = 0, hi = n_rows;
    while (hi - lo > 1) {
      const long long mid = (lo + hi) >> 1;
      if (rowptr[mid] <= e) lo = mid; else hi = mid;
    }
    row = lo;
  }
  const float v = e_val[e];
  if (v == 0.f) return;
  float *t = dtable + ((long long)e_rel[e] * n_src + e_src[e]) * (size_t)d;
  const float *g = G + (size_t)row * d;
  for (int f = 4 * j; f < d; f += 4 * lpm) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (f + q < d) atomicAdd(t + f + q, v * g[f + q]);
  }
}

// Diagonal-weight layer, weight gradient: dw[r, j] = sum_{slots of r} val * X[src, j] * G[dst, j].  Relation-major plan; one
// wave per work item (a chunk range of one relation).  `lpm` lanes per slot (one float4 of the row each: a slot's two rows
// are read ONCE, whole -- the first version walked 16 columns per pass and fetched every 128-byte line of a d = 32 row
// twice: 57 M fabric requests per launch on the AM-shaped graph instead of 27 M, profiles/r02_pmc_csr_kernels.json),
// 64 / lpm slots in flight, two per lane; the slot groups are summed with wave shuffles, one float4 of atomics per
// (item, 4 columns).  Rows wider than 256 floats loop over column blocks.
template <bool VEC>
__global__ __launch_bounds__(WG) void diag_wgrad_kernel(
    const float *__restrict__ X, const float *__restrict__ G, float *__restrict__ dw, const int *__restrict__ p_src,
    const int *__restrict__ p_dst, const float *__restrict__ p_val, const int *__restrict__ chunk_rel,
    const int2 *__restrict__ items, long long n_items, int d, int lpm) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long it = (long long)blockIdx.x * (WG / 64) + wave;
  if (it >= n_items) return;
  const int2 range = items[it];
  if (range.x >= range.y) return;
  const int rel = chunk_rel[range.x];
  const int g = lane / lpm, j = lane % lpm, groups = 64 / lpm;
  const int s0 = range.x * RGCN_CHUNK, s1 = range.y * RGCN_CHUNK;
  constexpr bool vec4 = VEC;
  for (int f0 = 0; f0 < d; f0 += 4 * lpm) {
    const int col = f0 + 4 * j;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (col < d) {
      // branch-free (pads: val = 0, src = 0, dst = -1 -> row 0)
      auto one = [&](int s, bool have) -> f32x4 {
        const int ss = have ? s : s0;
        const float v = have ? p_val[ss] : 0.f;
        const float *x = X + (size_t)p_src[ss] * d + col, *gr = G + (size_t)max(p_dst[ss], 0) * d + col;
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (vec4) {
          r = *reinterpret_cast<const f32x4 *>(x) * *reinterpret_cast<const f32x4 *>(gr) * v;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (col + i < d) r[i] = v * x[i] * gr[i];
        }
        return r;
      };
      for (int s = s0 + g; s < s1; s += 2 * groups) {
        const f32x4 a = one(s, true), b2 = one(s + groups, s + groups < s1);
        acc += a + b2;
      }
    }
    for (int sh = lpm; sh < 64; sh *= 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] += __shfl_xor(acc[i], sh, 64);
    }
    if (g == 0 && col < d) {
      float *o = dw + (size_t)rel * d + col;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (col + i < d && acc[i] != 0.f) atomicAdd(o + i, acc[i]);
    }
  }
}

// ------------------------------------------------------------------ column sum (bias gradient)
// Two stages, no atomics, fixed summation order (bit-reproducible): stage A -- every workgroup streams its rows (float4 per
// thread when d % 4 == 0) and leaves one partial row in `partial[block][d]`; stage B -- one workgroup sums the partial rows.
template <bool VEC4>
__global__ __launch_bounds__(WG) void colsum_a_kernel(const float *__restrict__ G, float *__restrict__ partial, long long n, int d) {
  __shared__ float part[WG * 4];
  const int tid = threadIdx.x;
  const int lanes = VEC4 ? d / 4 : d;                 // threads per row
  const int groups = max(1, WG / lanes);
  const int c = tid % lanes, grp = tid / lanes;
  for (int cb = 0; cb < lanes; cb += WG) {            // column blocks when a row needs more than 256 threads
    const int cc = cb + c;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (grp < groups && cc < lanes)
      for (long long row = (long long)blockIdx.x * groups + grp; row < n; row += (long long)gridDim.x * groups) {
        if (VEC4) {
          const float4 x = *reinterpret_cast<const float4 *>(G + (size_t)row * d + 4 * cc);
          a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
        } else {
          a.x += G[(size_t)row * d + cc];
        }
      }
    reinterpret_cast<float4 *>(part)[tid] = a;
    __syncthreads();
    if (grp == 0 && cc < lanes) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int g2 = 0; g2 < groups; ++g2) {
        const float4 x = reinterpret_cast<const float4 *>(part)[g2 * lanes + c];
        t.x += x.x; t.y += x.y; t.z += x.z; t.w += x.w;
      }
      float *o = partial + (size_t)blockIdx.x * d;
      if (VEC4) *reinterpret_cast<float4 *>(o + 4 * cc) = t;
      else o[cc] = t.x;
    }
    __syncthreads();
  }
}

// Wide rows (d >= 128, d % 4 == 0): the narrow form above gives a row to d / 4 threads and the launch few workgroups (one
// partial row each, summed serially by stage B): 330,000 x 200 took 320 us, 0.8 TB/s.  Here a workgroup owns a block of 64
// columns (16 lanes x float4) and 16 rows at a time (256-byte contiguous pieces per row), grid = column blocks x row
// partitions (~1024 workgroups), fixed summation order: rows of a partition in stride order per lane, then the 16 row lanes.
__global__ __launch_bounds__(WG) void colsum_wide_a_kernel(const float *__restrict__ G, float *__restrict__ partial, long long n, int d) {
  __shared__ f32x4 part[WG];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int col = blockIdx.x * 64 + 4 * cl;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
  if (col < d) {
    const long long step = (long long)gridDim.y * 16;
    long long row = (long long)blockIdx.y * 16 + rl;
    for (; row + step < n; row += 2 * step) {             // two independent chains: two loads in flight per thread
      a0 += *reinterpret_cast<const f32x4 *>(G + (size_t)row * d + col);
      a1 += *reinterpret_cast<const f32x4 *>(G + (size_t)(row + step) * d + col);
    }
    if (row < n) a0 += *reinterpret_cast<const f32x4 *>(G + (size_t)row * d + col);
  }
  part[threadIdx.x] = a0 + a1;
  __syncthreads();
  if (rl == 0 && col < d) {
    f32x4 t = part[cl];
    for (int r = 1; r < 16; ++r) t += part[16 * r + cl];
    *reinterpret_cast<f32x4 *>(partial + (size_t)blockIdx.y * d + col) = t;
  }
}

// partial [n_part][d] -> db: one workgroup per 64 columns, 4 row groups x 64 columns, four chains per thread
__global__ __launch_bounds__(WG) void colsum_wide_b_kernel(const float *__restrict__ partial, float *__restrict__ db, int n_part, int d) {
  __shared__ float part[WG];
  const int c = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + c;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col < d) {
    int p = grp;
    for (; p + 12 < n_part; p += 16) {
      a0 += partial[(size_t)p * d + col];
      a1 += partial[(size_t)(p + 4) * d + col];
      a2 += partial[(size_t)(p + 8) * d + col];
      a3 += partial[(size_t)(p + 12) * d + col];
    }
    for (; p < n_part; p += 4) a0 += partial[(size_t)p * d + col];
  }
  part[threadIdx.x] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (grp == 0 && col < d) db[col] = (part[c] + part[64 + c]) + (part[128 + c] + part[192 + c]);
}

__global__ __launch_bounds__(WG) void colsum_b_kernel(const float *__restrict__ partial, float *__restrict__ db, int n_part, int d) {
  __shared__ float part[WG];
  const int tid = threadIdx.x;
  const int groups = max(1, WG / d);
  const int c = tid % d, grp = tid / d;
  for (int cb = 0; cb < d; cb += WG) {
    const int cc = cb + c;
    float a = 0.f;
    if (grp < groups && cc < d) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;      // four independent chains: the loads of one thread overlap
      int p = grp;
      for (; p + 3 * groups < n_part; p += 4 * groups) {
        a0 += partial[(size_t)p * d + cc];
        a1 += partial[(size_t)(p + groups) * d + cc];
        a2 += partial[(size_t)(p + 2 * groups) * d + cc];
        a3 += partial[(size_t)(p + 3 * groups) * d + cc];
      }
      for (; p < n_part; p += groups) a0 += partial[(size_t)p * d + cc];
      a = (a0 + a1) + (a2 + a3);
    }
    part[tid] = a;
    __syncthreads();
    if (grp == 0 && cc < d) {
      float t = 0.f;
      for (int g2 = 0; g2 < groups; ++g2) t += part[g2 * d + c];
      db[cc] = t;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ DistMult
__global__ __launch_bounds__(WG) void distmult_fwd_kernel(
    const long long *__restrict__ tr, long long T, const float *__restrict__ nodes, const float *__restrict__ rel,
    const float *__restrict__ sb, const float *__restrict__ pb, const float *__restrict__ ob,
    float *__restrict__ scores, int d, long long n_nodes, int n_rel, int *__restrict__ err, int *__restrict__ counts,
    int *__restrict__ ranks, int vec) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long wstride = (long long)gridDim.x * (WG / 64);
  for (long long t = (long long)blockIdx.x * (WG / 64) + wave; t < T; t += wstride) {
    const long long s = tr[3 * t], p = tr[3 * t + 1], o = tr[3 * t + 2];
    if (s < 0 || s >= n_nodes || o < 0 || o >= n_nodes || p < 0 || p >= n_rel) {   // the reference raises IndexError here
      if (lane == 0) { if (err) atomicMax(err, 1); scores[t] = 0.f; }
      continue;
    }
    // The backward pass walks the scored triples as two CSRs (by subject, by object).  Their counting pass rides here: one returning
    // atomic per side gives the triple its rank within its row, behind the row loads of this wave instead of in two launches of their
    // own (28 + 41 us for WN18's 330 k triples, bound by the rate of atomic requests); rgcn_distmult_csr_place finishes the CSRs.
    int rk_s = 0, rk_o = 0;
    if (counts && lane == 0) { rk_s = atomicAdd(counts + 1 + s, 1); rk_o = atomicAdd(counts + n_nodes + 2 + o, 1); }
    const float *ns = nodes + (size_t)s * d, *rp = rel + (size_t)p * d, *no = nodes + (size_t)o * d;
    float a = 0.f;
    if (vec) {                                            // 16-byte pieces (d % 4 == 0, aligned tables): a 200-wide row is ONE load of 50 lanes instead of four passes of 64
      for (int j = 4 * lane; j < d; j += 256) {
        const f32x4 x = *reinterpret_cast<const f32x4 *>(ns + j), w = *reinterpret_cast<const f32x4 *>(rp + j),
                    y = *reinterpret_cast<const f32x4 *>(no + j);
        const f32x4 q = x * w * y;
        a += (q[0] + q[1]) + (q[2] + q[3]);
      }
    } else {
      for (int j = lane; j < d; j += 64) a += ns[j] * rp[j] * no[j];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if (lane == 0) {
      if (sb) a += sb[s] + pb[p] + ob[o];
      scores[t] = a;
      if (counts) *reinterpret_cast<int2 *>(ranks + 2 * t) = make_int2(rk_s, rk_o);
    }
  }
}

// Each wave walks 64 consecutive triples.  The relation gradient is accumulated in registers while the
// predicate stays the same and flushed with one atomic per (run, feature): callers that sort the triples by
// predicate (functional.py does) turn T*d contended atomics on R*d addresses into ~(T/64 + R)*d.
__global__ __launch_bounds__(WG) void distmult_bwd_kernel(
    const long long *__restrict__ tr, long long T, const float *__restrict__ nodes, const float *__restrict__ rel,
    const float *__restrict__ gs, float *__restrict__ dnodes, float *__restrict__ drel, float *__restrict__ dsb,
    float *__restrict__ dpb, float *__restrict__ dob, int d, long long n_nodes, int n_rel, int skip_nodes) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long wstride = (long long)gridDim.x * (WG / 64) * 64;
  for (long long t0 = ((long long)blockIdx.x * (WG / 64) + wave) * 64; t0 < T; t0 += wstride) {
    const long long t1 = min(T, t0 + 64);
    for (int i0 = 0; i0 < d; i0 += 256) {                 // 4 features per lane and pass
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      long long cur = -1;
      for (long long t = t0; t < t1; ++t) {
        const long long s = tr[3 * t], p = tr[3 * t + 1], o = tr[3 * t + 2];
        if (s < 0 || s >= n_nodes || o < 0 || o >= n_nodes || p < 0 || p >= n_rel) continue;   // flagged by the forward
        if (p != cur) {
          if (cur >= 0)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int j = i0 + q * 64 + lane;
              if (j < d) atomicAdd(&drel[(size_t)cur * d + j], acc[q]);
              acc[q] = 0.f;
            }
          cur = p;
        }
        const float g = gs[t];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = i0 + q * 64 + lane;
          if (j < d) {
            const float a = nodes[(size_t)s * d + j], b = rel[(size_t)p * d + j], c = nodes[(size_t)o * d + j];
            if (!skip_nodes) {
              atomicAdd(&dnodes[(size_t)s * d + j], g * b * c);
              atomicAdd(&dnodes[(size_t)o * d + j], g * b * a);
            }
            acc[q] += g * a * c;
          }
        }
        if (i0 == 0 && dsb && lane == 0) {
          atomicAdd(&dsb[s], g);
          atomicAdd(&dpb[p], g);
          atomicAdd(&dob[o], g);
        }
      }
      if (cur >= 0)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = i0 + q * 64 + lane;
          if (j < d) atomicAdd(&drel[(size_t)cur * d + j], acc[q]);
        }
    }
  }
}

// Entity gradients of DistMult without atomics: the scored triples are indexed twice (CSR by subject, CSR by object --
// the counting-sort builder of rgcn_build.hip) and one wave per entity sums its rows
//     dnodes[n] = sum_{t: s_t = n} g_t r[p_t] * nodes[o_t]  +  sum_{t: o_t = n} g_t r[p_t] * nodes[s_t]
// in registers (4 features per lane and pass, four entries' row loads in flight).  The scatter form it replaces issued
// 2 T d fp32 atomics (132 M for a WN18 batch: 0.43 of the kernel's 0.51 ms at 2 cycles per lane and CU).
// (distmult_bwd_all_kernel<3, VEC, false> below: both sides in one launch, no relation table.)
// All of DistMult's gradients from the two CSRs, no sort and no second pass over the triples: the entity walk above already
// holds, for every triple of the subject-side CSR, the row of the object; with the entity's own row it is the triple's
// contribution to the RELATION gradient, d_rel[p] += g * x_s * x_o.  Each wave keeps a private [n_rel][d] table (+ the
// predicate bias) in LDS -- plain read-modify-write, no atomics --, the four tables of a workgroup are summed at the end and
// added to d_rel with one atomic per workgroup and element (<= 512 persistent workgroups).  Needs n_rel (d + 1) <= 4096 floats
// per wave (WN18: 18 x 200); larger relation tables keep the predicate-sorted kernel above.  Round 2 before this: torch.argsort
// by predicate (rocPRIM merge sort, 0.15 ms for 330 k triples) + two index gathers + distmult_bwd_kernel (0.12 ms).
// (One table per WORKGROUP with ds_add_f32 -- 4x less LDS, twice the resident waves -- measured 0.37 ms against 0.21: LDS float
// atomics are slow on gfx950 even without address conflicts.)
// Round 5: ONE table per workgroup after all, in DOUBLES with ds_add_f64 -- the LDS float atomic gfx950 runs at full rate when the lanes
// of an instruction touch consecutive doubles.  The kernel is a chain of dependent round trips per entity (row pointers -> indices ->
// rows -> read-modify-write of the entity's row) that only resident waves hide, and four private tables per workgroup (58 KB) held it
// at 8 waves per CU: 135 us for WN18's subject side where the table-free object side takes 48.  29 KB per workgroup = 20 waves per CU.
// The table is stored feature-permuted (feature 4 l + q of a relation's row at q * ceil(d / 4) + l) so that instruction q of the 64
// lanes adds to 64 consecutive doubles.
// SIDES: 1 = subject side (with the LDS table), ADDING to the dnodes rows the object-side launch wrote; 2 = object side only (no LDS:
// launched first, with many more resident waves).
// The entries of an entity's row (other end, predicate, score gradient) are loaded by the lanes in ONE round trip -- lane i takes entry i --
// and handed to the wave four at a time with v_readlane: scalar row addresses, four entity rows and four relation rows in flight per
// step of the chain instead of two behind a per-pair index load.
// VEC: d is a multiple of 4 -- every lane loads whole float4s without a branch (lanes past the row's end from feature 0, and keep nothing).
// TABLE = false: entity gradients only (relation tables too large for the LDS: the predicate-sorted kernel above does the rest); SIDES = 3.
// Entries: int4 {other end, predicate, score gradient (bits), -} -- one 16-byte load per lane (rgcn_distmult_csr_place writes them).
template <int SIDES, bool VEC, bool TABLE>
__global__ __launch_bounds__(WG) void distmult_bwd_all_kernel(
    const int *__restrict__ rp_s, const int *__restrict__ rp_o, const int4 *__restrict__ entries, const float *__restrict__ nodes,
    const float *__restrict__ rel, float *__restrict__ dnodes, float *__restrict__ drel, float *__restrict__ dsb,
    float *__restrict__ dpb, float *__restrict__ dob, long long N, int n_rel, int d) {
  extern __shared__ __attribute__((aligned(16))) double ldsd[];
  const int lane = threadIdx.x & 63;
  const int dq = (d + 3) >> 2, dp = 4 * dq;                 // a relation's row: 4 runs of dq doubles (feature 4 l + q at q * dq + l)
  const int tab_n = n_rel * dp + n_rel;                     // [n_rel][dp] + predicate bias [n_rel]
  double *tab = ldsd, *pb = tab + (size_t)n_rel * dp;
  if (TABLE && (SIDES & 1)) {
    for (int i = threadIdx.x; i < tab_n; i += WG) tab[i] = 0.0;
    __syncthreads();
  }
  const long long wave0 = ((long long)blockIdx.x * WG + threadIdx.x) >> 6, nw = ((long long)gridDim.x * WG) >> 6;
  for (long long n = wave0; n < N; n += nw) {
    for (int f0 = 0; f0 < d; f0 += 256) {
      const int f = f0 + 4 * lane;
      const bool act = f < d;
      const int fc = act ? f : 0;
      auto load4 = [&](const float *row) -> f32x4 {
        if (VEC) return *reinterpret_cast<const f32x4 *>(row + fc);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (f + q < d) v[q] = row[f + q];
        return v;
      };
      float *o = dnodes + (size_t)n * d;
      const f32x4 xn = load4(nodes + (size_t)n * d);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (SIDES == 1) acc = load4(o);                  // the object-side launch wrote this row
#pragma unroll
      for (int side = 0; side < 2; ++side) {
        if (!((SIDES >> side) & 1)) continue;
        const int *rp = side ? rp_o : rp_s;
        const int e0 = __builtin_amdgcn_readfirstlane(rp[n]), e1 = __builtin_amdgcn_readfirstlane(rp[n + 1]);
        float gsum = 0.f;
        for (int c0 = e0; c0 < e1; c0 += 64) {
          const int cnt = min(64, e1 - c0);
          const int4 me = entries[c0 + min(lane, cnt - 1)];      // lanes past the row's end repeat its last entry, with a zero gradient
          const int mo = me.x, mr = me.y, mg = lane < cnt ? me.z : 0;
          for (int j = 0; j < cnt; j += 4) {
            f32x4 xe[4], we[4];
            float ge[4];
            int pe[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int oe = __builtin_amdgcn_readlane(mo, j + u);
              pe[u] = __builtin_amdgcn_readlane(mr, j + u);
              ge[u] = __int_as_float(__builtin_amdgcn_readlane(mg, j + u));
              xe[u] = load4(nodes + (size_t)oe * d);
              we[u] = load4(rel + (size_t)pe[u] * d);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              acc += xe[u] * we[u] * ge[u];
              gsum += ge[u];
              if (TABLE && side == 0 && act && j + u < cnt) {   // relation gradient: the workgroup's LDS table of doubles (ds_add_f64, consecutive doubles per instruction)
                double *ta = tab + (size_t)pe[u] * dp + (f >> 2);
                const f32x4 ca = xn * xe[u] * ge[u];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                  if (VEC || f + q < d) __hip_atomic_fetch_add(ta + q * dq, (double)ca[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (f0 == 0 && lane == 0 && dpb) __hip_atomic_fetch_add(pb + pe[u], (double)ge[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
            }
          }
        }
        if (f0 == 0 && lane == 0 && dsb) (side ? dob : dsb)[n] = gsum;
      }
      if (VEC) {
        if (act) *reinterpret_cast<f32x4 *>(o + f) = acc;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (f + q < d) o[f + q] = acc[q];
      }
    }
  }
  if (!TABLE || !(SIDES & 1)) return;
  __syncthreads();
  for (int i = threadIdx.x; i < n_rel * d; i += WG) {       // un-permute: drel[r][j] = tab[r][(j & 3) * dq + (j >> 2)]
    const int r = i / d, j = i - r * d;
    const float t = (float)tab[(size_t)r * dp + (j & 3) * dq + (j >> 2)];
    if (t != 0.f) atomicAdd(drel + i, t);
  }
  if (dpb)
    for (int i = threadIdx.x; i < n_rel; i += WG) {
      const float t = (float)pb[i];
      if (t != 0.f) atomicAdd(dpb + i, t);
    }
}

int pow2_lanes(int d) {
  int l = 1;
  while (l < d && l < 64) l <<= 1;
  return l;
}

}  // namespace

// =================================================================== C ABI launchers

extern "C" int rgcn_pack_w16_f32(const float *W, float *Wp, int32_t R, void *stream) {
  if (!W || !Wp || R <= 0) { rgcn_set_error("pack_w16: bad argument"); return RGCN_EINVAL; }
  const int n = R * 256;
  hipLaunchKernelGGL(pack_w16_kernel, dim3((unsigned)((n + WG - 1) / WG)), dim3(WG), 0, (hipStream_t)stream, W, Wp, n);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_pack_w_blocks_f32(const float *W, float *Wp, int32_t R, int32_t d_in, int32_t d_out, void *stream) {
  if (!W || !Wp || R <= 0 || d_in <= 0 || d_out <= 0 || d_in % 16 || d_out % 16) { rgcn_set_error("pack_w_blocks: widths must be multiples of 16"); return RGCN_EINVAL; }
  const int64_t n = (int64_t)R * d_in * d_out;
  if (n > INT32_MAX) { rgcn_set_error("pack_w_blocks: weight tensor too large"); return RGCN_EUNSUPPORTED; }
  hipLaunchKernelGGL(pack_w_blocks_kernel, dim3((unsigned)((n + WG - 1) / WG)), dim3(WG), 0, (hipStream_t)stream, W, Wp,
                     (int)n, d_in / 16, d_out / 16);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_spmm_f32(const float *X, const float *W, const float *bias, float *out, const int32_t *p_src,
                             const int32_t *p_dst, const float *p_val, const int32_t *p_pack,
                             const int32_t *chunk_rel, const int32_t *units, int64_t n_units, int64_t n_split,
                             int32_t tile_rows, int64_t n_dst, int64_t n_src, int32_t R, int32_t d_in,
                             int32_t d_out, int32_t flags, void *stream) {
  (void)n_src;
  (void)R;
  const int64_t n_tiles = n_units;   // one wave per work unit
  // n_units may cover only a slab of the tiles (callers that overlap a collective with the remaining slabs)
  if (!X || !W || !out || (n_units && !units) || d_in <= 0 || d_out <= 0 || tile_rows <= 0 || n_dst < 0 ||
      n_units < 0 || n_split < 0 || ((flags & RGCN_F_RELU) && n_split)) {
    rgcn_set_error("spmm: bad argument");
    return RGCN_EINVAL;
  }
  const int relu_out = flags & RGCN_F_RELU;
  const bool packed = (flags & RGCN_F_WPACKED) != 0;
  const bool blocks = d_in % 16 == 0 && d_out % 16 == 0 && d_in <= 64 && d_out <= 64;
  if (packed && (!blocks || !p_pack)) {
    rgcn_set_error("spmm: RGCN_F_WPACKED needs widths that are multiples of 16 (<= 64) and packed slots");
    return RGCN_EINVAL;
  }
  if (n_tiles == 0) return RGCN_OK;
  const int ldt = (d_out + 3) & ~3;
  const size_t lds = (size_t)SPMM_WAVES * tile_rows * ldt * sizeof(float);
  if (lds > LDS_TILE_BYTES) {
    rgcn_set_error("spmm: %d waves x tile_rows*d_out*4 = %zu exceeds the %d-byte LDS budget", SPMM_WAVES, lds,
                   LDS_TILE_BYTES);
    return RGCN_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  if (n_split) HIP_TRY(zero_async(out, (size_t)n_dst * d_out * sizeof(float), st));  // hub tiles are summed atomically
  dim3 grid((unsigned)((n_tiles + SPMM_WAVES - 1) / SPMM_WAVES)), block(WG);
  const int nt = (int)n_tiles;
  const int2 *pk = reinterpret_cast<const int2 *>(p_pack);
  const int4 *tile_ptr = reinterpret_cast<const int4 *>(units);
#define RGCN_LAUNCH_GENERIC(NJT)                                                                                   \
  hipLaunchKernelGGL(spmm_generic_kernel<NJT>, grid, block, lds, st, X, W, bias, out, p_src, p_dst, p_val,         \
                     chunk_rel, tile_ptr, nt, tile_rows, (int)n_dst, d_in, d_out, ldt, relu_out)
#define RGCN_LAUNCH_D16(U, P)                                                                                      \
  hipLaunchKernelGGL((spmm_d16_kernel<U, P>), grid, block, lds, st, X, W, bias, out, p_src, p_dst, p_val, pk,      \
                     chunk_rel, tile_ptr, nt, tile_rows, (int)n_dst, relu_out)
  if (d_in == 16 && d_out == 16) {
    const int U = rgcn_option_value(RGCN_OPT_SPMM_U);
    if (packed) {
      if (U >= 8) RGCN_LAUNCH_D16(8, true);
      else if (U >= 4) RGCN_LAUNCH_D16(4, true);
      else if (U >= 2) RGCN_LAUNCH_D16(2, true);
      else RGCN_LAUNCH_D16(1, true);
    } else {
      if (U >= 8) RGCN_LAUNCH_D16(8, false);
      else if (U >= 4) RGCN_LAUNCH_D16(4, false);
      else if (U >= 2) RGCN_LAUNCH_D16(2, false);
      else RGCN_LAUNCH_D16(1, false);
    }
  } else if (packed) {   // blocks of 16 features, fragments pre-swizzled by rgcn_pack_w_blocks_f32
#define RGCN_LAUNCH_WIDE(NIC, NJC, UC)                                                                              \
  hipLaunchKernelGGL((spmm_wide_kernel<NIC, NJC, UC>), grid, block, lds, st, X, W, bias, out, pk, chunk_rel, tile_ptr, \
                     nt, tile_rows, (int)n_dst, relu_out)
#define RGCN_WIDE_ROW(NIC)                                                                       \
  switch (d_out / 16) {                                                                          \
    case 1: RGCN_LAUNCH_WIDE(NIC, 1, 2); break;                                                  \
    case 2: RGCN_LAUNCH_WIDE(NIC, 2, (NIC <= 2 ? 2 : 1)); break;                                 \
    case 3: RGCN_LAUNCH_WIDE(NIC, 3, 1); break;                                                  \
    default: RGCN_LAUNCH_WIDE(NIC, 4, 1); break;                                                 \
  }
    switch (d_in / 16) {
      case 1: RGCN_WIDE_ROW(1) break;
      case 2: RGCN_WIDE_ROW(2) break;
      case 3: RGCN_WIDE_ROW(3) break;
      default: RGCN_WIDE_ROW(4) break;
    }
#undef RGCN_WIDE_ROW
#undef RGCN_LAUNCH_WIDE
  } else if (d_out <= 16) {
    RGCN_LAUNCH_GENERIC(1);
  } else if (d_out <= 32) {
    RGCN_LAUNCH_GENERIC(2);
  } else if (d_out <= 64) {
    RGCN_LAUNCH_GENERIC(4);
  } else {
    RGCN_LAUNCH_GENERIC(8);
  }
#undef RGCN_LAUNCH_GENERIC
#undef RGCN_LAUNCH_D16
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_spmm_scatter_f32(const float *X, const float *Wp, float *Y, const int32_t *p_src, const float *p_val,
                                     const int32_t *p_pos, const int32_t *chunk_rel, const int32_t *items,
                                     int64_t n_items, int32_t d, void *stream) {
  if (!X || !Wp || !Y || n_items < 0 || (n_items && (!p_src || !p_val || !chunk_rel || !items))) { rgcn_set_error("spmm_scatter: bad argument"); return RGCN_EINVAL; }
  if (d != 16) { rgcn_set_error("spmm_scatter: only d = 16"); return RGCN_EUNSUPPORTED; }
  if (!n_items) return RGCN_OK;
  hipLaunchKernelGGL(spmm_scatter_d16_kernel<4>, dim3((unsigned)((n_items + WG / 64 - 1) / (WG / 64))), dim3(WG), 0,
                     (hipStream_t)stream, X, Wp, Y, p_src, p_val, p_pos, chunk_rel, reinterpret_cast<const int2 *>(items),
                     (int)n_items);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_segment_sum_f32(const float *Y, const int32_t *rowptr, const float *bias, float *out, int64_t n_rows,
                                    int32_t d, int32_t flags, void *stream) {
  if (!Y || !rowptr || !out || n_rows < 0) { rgcn_set_error("segment_sum: bad argument"); return RGCN_EINVAL; }
  if (d != 16) { rgcn_set_error("segment_sum: only d = 16"); return RGCN_EUNSUPPORTED; }
  if (!n_rows) return RGCN_OK;
  const unsigned gx = (unsigned)std::min<int64_t>((n_rows * 4 + WG - 1) / WG, 256 * 64);
  hipLaunchKernelGGL(segment_sum_d16_kernel, dim3(gx), dim3(WG), 0, (hipStream_t)stream, Y, rowptr, bias, out,
                     (long long)n_rows, flags & RGCN_F_RELU);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_segment_gather_sum_f32(const float *Y, const int32_t *perm, const int32_t *rowptr, const float *bias,
                                           float *out, int64_t n_rows, int32_t d, int32_t flags, void *stream) {
  if (!Y || !perm || !rowptr || !out || n_rows < 0) { rgcn_set_error("segment_gather_sum: bad argument"); return RGCN_EINVAL; }
  if (d != 16) { rgcn_set_error("segment_gather_sum: only d = 16"); return RGCN_EUNSUPPORTED; }
  if (!n_rows) return RGCN_OK;
  const unsigned gx = (unsigned)std::min<int64_t>((n_rows * 4 + WG - 1) / WG, 256 * 64);
  hipLaunchKernelGGL(segment_gather_sum_d16_kernel, dim3(gx), dim3(WG), 0, (hipStream_t)stream, Y, perm, rowptr, bias, out,
                     (long long)n_rows, flags & RGCN_F_RELU);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_segment_gather_sum_units_f32(const float *Y, const int32_t *perm, const int32_t *units, int64_t n_units, int64_t n_split,
                                                 const float *bias, float *out, int64_t n_rows, int32_t d, int32_t flags, void *stream) {
  if (!Y || !perm || !units || !out || n_rows < 0 || n_units < 0) { rgcn_set_error("segment_gather_sum_units: bad argument"); return RGCN_EINVAL; }
  if (d != 16) { rgcn_set_error("segment_gather_sum_units: only d = 16"); return RGCN_EUNSUPPORTED; }
  if ((flags & RGCN_F_RELU) && n_split) { rgcn_set_error("segment_gather_sum_units: RGCN_F_RELU with shared units"); return RGCN_EINVAL; }
  if (!n_rows || !n_units) return RGCN_OK;
  hipStream_t st = (hipStream_t)stream;
  if (n_split) HIP_TRY(zero_async(out, (size_t)n_rows * 16 * sizeof(float), st));
  const unsigned gx = (unsigned)std::min<int64_t>((n_units * 4 + WG - 1) / WG, 256 * 64);
  hipLaunchKernelGGL(segment_gather_sum_units_d16_kernel, dim3(gx), dim3(WG), 0, st, Y, perm, reinterpret_cast<const int4 *>(units), bias, out,
                     (long long)n_units, flags & RGCN_F_RELU);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_wgrad_f32(const float *X, const float *G, float *dW, const int32_t *p_src, const int32_t *p_dst,
                              const float *p_val, const int32_t *chunk_rel, const int32_t *items, int64_t n_items,
                              int64_t n_dst, int64_t n_src, int32_t R, int32_t d_in, int32_t d_out, void *stream) {
  (void)n_dst;
  (void)n_src;
  if (!X || !G || !dW || R <= 0 || d_in <= 0 || d_out <= 0 || n_items < 0) {
    rgcn_set_error("wgrad: bad argument");
    return RGCN_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(zero_async(dW, (size_t)R * d_in * d_out * sizeof(float), st));
  if (n_items == 0) return RGCN_OK;
  const unsigned gx = (unsigned)((n_items + WG / 64 - 1) / (WG / 64));
  const int2 *it2 = reinterpret_cast<const int2 *>(items);
  if (d_in == 16 && d_out == 16) {
    hipLaunchKernelGGL(wgrad_d16_kernel<4>, dim3(gx), dim3(WG), 0, st, X, G, dW, p_src, p_dst, p_val, chunk_rel, it2,
                       (int)n_items);
  } else if (d_in <= 16 && d_out <= 16) {
    hipLaunchKernelGGL((wgrad_generic_kernel<1, 1>), dim3(gx, 1), dim3(WG), 0, st, X, G, dW, p_src, p_dst, p_val,
                       chunk_rel, it2, (int)n_items, d_in, d_out, 1);
  } else {
    constexpr int NIT = 2, NJT = 4;
    const int nig = (d_in + 16 * NIT - 1) / (16 * NIT), njg = (d_out + 16 * NJT - 1) / (16 * NJT);
    hipLaunchKernelGGL((wgrad_generic_kernel<NIT, NJT>), dim3(gx, (unsigned)(nig * njg)), dim3(WG), 0, st, X, G, dW,
                       p_src, p_dst, p_val, chunk_rel, it2, (int)n_items, d_in, d_out, njg);
  }
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_wgrad_tiled_f32(const float *X, const float *G, float *dW, const int32_t *p_src,
                                    const int32_t *p_dst, const float *p_val, const int32_t *chunk_rel,
                                    const int32_t *run_ptr, int64_t n_tiles, int32_t tile_rows, int64_t n_dst,
                                    int64_t n_src, int32_t R, int32_t d_in, int32_t d_out, int32_t tiles_per_item,
                                    void *stream) {
  (void)n_src;
  if (!X || !G || !dW || !run_ptr || R <= 0 || tile_rows <= 0 || tiles_per_item <= 0 || n_tiles < 0) {
    rgcn_set_error("wgrad_tiled: bad argument");
    return RGCN_EINVAL;
  }
  if (d_in != 16 || d_out != 16) { rgcn_set_error("wgrad_tiled: only d_in = d_out = 16"); return RGCN_EUNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(zero_async(dW, (size_t)R * 256 * sizeof(float), st));
  if (n_tiles == 0) return RGCN_OK;
  const int RGSEL = rgcn_option_value(RGCN_OPT_WGRAD_RG);
  const int USEL = rgcn_option_value(RGCN_OPT_WGRAD_U);
  const int RGv = RGSEL <= 1 ? 1 : RGSEL <= 2 ? 2 : RGSEL <= 4 ? 4 : RGSEL <= 8 ? 8 : 16;
  const int n_groups = (R + RGv - 1) / RGv;
  tiles_per_item = std::min(tiles_per_item, 8);   // the kernel keeps the run bounds of one item in registers
  const int64_t n_blocks = (n_tiles + tiles_per_item - 1) / tiles_per_item;
  const int64_t n_items = n_blocks * n_groups;
  // 8 interleaved work lists (one per XCD): list x holds the items of tile blocks x, x + 8, x + 16, ...
  const int64_t per_xcd = ((n_blocks + 7) / 8) * n_groups;
  const unsigned gx = (unsigned)(8 * ((per_xcd + WG / 64 - 1) / (WG / 64)));
#define RGCN_LAUNCH_WT(RGC, UC)                                                                                    \
  hipLaunchKernelGGL((wgrad_tiled_d16_kernel<RGC, UC>), dim3(gx), dim3(WG), 0, st, X, G, dW, p_src, p_dst, p_val, \
                     chunk_rel, run_ptr, (int)n_tiles, R, tiles_per_item, n_groups, (int)n_items)
#define RGCN_LAUNCH_WTU(RGC) { if (USEL >= 4) RGCN_LAUNCH_WT(RGC, 4); else if (USEL >= 2) RGCN_LAUNCH_WT(RGC, 2); else RGCN_LAUNCH_WT(RGC, 1); }
  switch (RGv) {
    case 1: RGCN_LAUNCH_WTU(1) break;
    case 2: RGCN_LAUNCH_WTU(2) break;
    case 4: RGCN_LAUNCH_WTU(4) break;
    case 8: RGCN_LAUNCH_WTU(8) break;
    default: RGCN_LAUNCH_WTU(16) break;
  }
#undef RGCN_LAUNCH_WTU
#undef RGCN_LAUNCH_WT
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_featureless_fwd_f32(const float *table, const float *bias, float *out, const int32_t *p_src,
                                        const int32_t *p_dst, const float *p_val, const int32_t *chunk_rel,
                                        const int32_t *units, int64_t n_units, int64_t n_split, int32_t tile_rows,
                                        int64_t n_dst, int64_t n_src, int32_t R, int32_t d_out, void *stream) {
  (void)R;
  const int64_t n_tiles = n_units;
  if (!table || !out || (n_units && !units) || d_out <= 0 || tile_rows <= 0 ||
      n_units < (n_dst + tile_rows - 1) / tile_rows || n_split < 0) {
    rgcn_set_error("featureless_fwd: bad argument");
    return RGCN_EINVAL;
  }
  if (n_tiles == 0) return RGCN_OK;
  const int ldt = (d_out + 3) & ~3;
  const size_t lds = (size_t)SPMM_WAVES * tile_rows * ldt * sizeof(float);
  if (lds > LDS_TILE_BYTES) { rgcn_set_error("featureless_fwd: LDS tile too large"); return RGCN_EINVAL; }
  if (n_split) HIP_TRY(zero_async(out, (size_t)n_dst * d_out * sizeof(float), (hipStream_t)stream));
  hipLaunchKernelGGL(featureless_fwd_kernel, dim3((unsigned)((n_tiles + SPMM_WAVES - 1) / SPMM_WAVES)), dim3(WG), lds,
                     (hipStream_t)stream, table, bias, out, p_src, p_dst, p_val, chunk_rel,
                     reinterpret_cast<const int4 *>(units), (int)n_tiles,
                     tile_rows, (int)n_dst, (long long)n_src, d_out, ldt);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_diag_spmm_f32(const float *X, const float *w, const float *bias, float *out, const int32_t *rowptr_units,
                                  int64_t n_units, int64_t n_split, const int32_t *e_src, const int32_t *e_rel,
                                  const float *e_val, int64_t n_rows, int32_t R, int32_t d, void *stream) {
  (void)R;
  if (!X || !w || !out || d <= 0 || n_rows < 0 || n_units < 0 || n_split < 0 ||
      (n_units && (!rowptr_units || !e_src || !e_rel || !e_val))) {
    rgcn_set_error("diag_spmm: bad argument");
    return RGCN_EINVAL;
  }
  if (n_rows == 0 || n_units == 0) return RGCN_OK;
  if (n_split) HIP_TRY(zero_async(out, (size_t)n_rows * d * sizeof(float), (hipStream_t)stream));
  int lpm = 1;                                  // lanes per message: float4 each, a power of two
  while (lpm < 64 && 4 * lpm < d) lpm *= 2;
  const int lr = std::min(64, std::max(16, 2 * lpm));   // lanes per unit: >= 2 messages of a row in flight
  const int upw = WG / lr;                      // units per workgroup
  const unsigned gx = (unsigned)((n_units + upw - 1) / upw);
  hipLaunchKernelGGL(diag_csr_kernel<false>, dim3(gx), dim3(WG), 0, (hipStream_t)stream, X, w, bias, out,
                     reinterpret_cast<const int4 *>(rowptr_units), (long long)n_units, e_src, e_rel, e_val, d, lpm, lr, 0LL, 0);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_featureless_csr_fwd_f32(const float *table, const float *bias, float *out, const int32_t *units, int64_t n_units,
                                            int64_t n_split, const int32_t *e_src, const int32_t *e_rel, const float *e_val,
                                            int64_t n_rows, int64_t n_src, int32_t R, int32_t d, int32_t relu, void *stream) {
  if (!table || !out || d <= 0 || n_rows < 0 || n_src <= 0 || R <= 0 || n_units < 0 || n_split < 0 ||
      (n_units && (!units || !e_src || !e_rel || !e_val))) {
    rgcn_set_error("featureless_csr_fwd: bad argument");
    return RGCN_EINVAL;
  }
  if (relu && n_split) { rgcn_set_error("featureless_csr_fwd: relu in the epilogue needs rows that are not cut into shared pieces"); return RGCN_EUNSUPPORTED; }
  if (n_rows == 0 || n_units == 0) return RGCN_OK;
  if (n_split) HIP_TRY(zero_async(out, (size_t)n_rows * d * sizeof(float), (hipStream_t)stream));
  int lpm = 1;                                  // lanes per message: float4 each, a power of two
  while (lpm < 64 && 4 * lpm < d) lpm *= 2;
  const int lr = std::min(64, std::max(16, 2 * lpm));
  const int upw = WG / lr;
  const unsigned gx = (unsigned)((n_units + upw - 1) / upw);
  hipLaunchKernelGGL(diag_csr_kernel<true>, dim3(gx), dim3(WG), 0, (hipStream_t)stream, table, nullptr, bias, out,
                     reinterpret_cast<const int4 *>(units), (long long)n_units, e_src, e_rel, e_val, d, lpm, lr, (long long)n_src, relu ? 1 : 0);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_featureless_csr_wgrad_f32(const float *G, float *dtable, const int32_t *rowptr, const int32_t *e_src, const int32_t *e_rel,
                                              const float *e_val, int64_t n_entries, int64_t n_rows, int64_t n_src, int32_t R, int32_t d,
                                              void *stream) {
  if (!G || !dtable || d <= 0 || n_rows < 0 || n_src <= 0 || R <= 0 || n_entries < 0 || (n_entries && (!rowptr || !e_src || !e_rel || !e_val))) {
    rgcn_set_error("featureless_csr_wgrad: bad argument");
    return RGCN_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(zero_async(dtable, (size_t)R * n_src * d * sizeof(float), st));
  if (!n_entries) return RGCN_OK;
  int lpm = 1;
  while (lpm < 64 && 4 * lpm < d) lpm *= 2;
  const long long threads = (long long)n_entries * lpm;
  hipLaunchKernelGGL(featureless_csr_wgrad_kernel, dim3((unsigned)((threads + WG - 1) / WG)), dim3(WG), 0, st, G, dtable, rowptr, nullptr,
                     e_src, e_rel, e_val, (long long)n_entries, (long long)n_rows, (long long)n_src, d, lpm);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_diag_wgrad_f32(const float *X, const float *G, float *dw, const int32_t *p_src, const int32_t *p_dst,
                                   const float *p_val, const int32_t *chunk_rel, const int32_t *items, int64_t n_items,
                                   int32_t R, int32_t d, void *stream) {
  if (!X || !G || !dw || R <= 0 || d <= 0 || n_items < 0 || (n_items && (!items || !p_src || !p_dst || !p_val || !chunk_rel))) {
    rgcn_set_error("diag_wgrad: bad argument");
    return RGCN_EINVAL;
  }
  HIP_TRY(zero_async(dw, (size_t)R * d * sizeof(float), (hipStream_t)stream));
  if (n_items == 0) return RGCN_OK;
  int lpm = 1;                                  // lanes per slot: float4 each, a power of two
  while (lpm < 64 && 4 * lpm < d) lpm *= 2;
  if ((d & 3) == 0) {
    hipLaunchKernelGGL(diag_wgrad_kernel<true>, dim3((unsigned)((n_items + WG / 64 - 1) / (WG / 64))), dim3(WG), 0, (hipStream_t)stream,
                     X, G, dw, p_src, p_dst, p_val, chunk_rel, reinterpret_cast<const int2 *>(items), (long long)n_items, d, lpm);
  } else {
    hipLaunchKernelGGL(diag_wgrad_kernel<false>, dim3((unsigned)((n_items + WG / 64 - 1) / (WG / 64))), dim3(WG), 0, (hipStream_t)stream,
                     X, G, dw, p_src, p_dst, p_val, chunk_rel, reinterpret_cast<const int2 *>(items), (long long)n_items, d, lpm);
  }
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_featureless_wgrad_f32(const float *G, float *dtable, const int32_t *p_src, const int32_t *p_dst,
                                          const float *p_val, const int32_t *chunk_rel, int64_t n_chunks,
                                          int64_t n_dst, int64_t n_src, int32_t R, int32_t d_out, void *stream) {
  (void)n_dst;
  if (!G || !dtable || R <= 0 || d_out <= 0 || n_chunks < 0) { rgcn_set_error("featureless_wgrad: bad argument"); return RGCN_EINVAL; }
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(zero_async(dtable, (size_t)R * n_src * d_out * sizeof(float), st));
  if (n_chunks == 0) return RGCN_OK;
  const unsigned gx = (unsigned)std::min<int64_t>((n_chunks + 3) / 4, 256 * 16);
  hipLaunchKernelGGL(featureless_wgrad_kernel, dim3(gx), dim3(WG), 0, st, G, dtable, p_src, p_dst, p_val, chunk_rel,
                     (long long)n_chunks, (long long)n_src, d_out, pow2_lanes(d_out));
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int64_t rgcn_colsum_scratch_floats(int64_t n, int32_t d) {
  (void)n;
  return (int64_t)1024 * d;
}

extern "C" int rgcn_colsum_f32(const float *G, float *db, float *scratch, int64_t n, int32_t d, void *stream) {
  if (!G || !db || !scratch || n < 0 || d <= 0) { rgcn_set_error("colsum: bad argument"); return RGCN_EINVAL; }
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) { HIP_TRY(zero_async(db, (size_t)d * sizeof(float), st)); return RGCN_OK; }
  const bool vec4 = d % 4 == 0;
  if (vec4 && d >= 128) {
    const int cblocks = (d + 63) / 64;
    const int parts = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)1024 / cblocks, (n + 15) / 16, (int64_t)1024}));
    hipLaunchKernelGGL(colsum_wide_a_kernel, dim3((unsigned)cblocks, (unsigned)parts), dim3(WG), 0, st, G, scratch, (long long)n, d);
    hipLaunchKernelGGL(colsum_wide_b_kernel, dim3((unsigned)cblocks), dim3(WG), 0, st, scratch, db, parts, d);
    HIP_TRY(hipGetLastError());
    return RGCN_OK;
  }
  const int lanes = vec4 ? d / 4 : d;
  const int groups = std::max(1, WG / lanes);
  // stage B is one workgroup walking the partial rows: wide rows leave it few row groups, so they get fewer partials
  const int64_t max_part = d <= 32 ? 512 : (d <= 128 ? 128 : 64);
  const unsigned gx = (unsigned)std::min<int64_t>((n + groups - 1) / groups, max_part);
  if (vec4) hipLaunchKernelGGL(colsum_a_kernel<true>, dim3(gx), dim3(WG), 0, st, G, scratch, (long long)n, d);
  else hipLaunchKernelGGL(colsum_a_kernel<false>, dim3(gx), dim3(WG), 0, st, G, scratch, (long long)n, d);
  hipLaunchKernelGGL(colsum_b_kernel, dim3(1), dim3(WG), 0, st, scratch, db, (int)gx, d);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_distmult_fwd_f32(const int64_t *triples, int64_t T, const float *nodes, const float *rel,
                                     const float *sbias, const float *pbias, const float *obias, float *scores,
                                     int64_t n_nodes, int32_t n_rel, int32_t d, int32_t *err_flag, int32_t *rank_counts,
                                     int32_t *ranks, void *stream) {
  if (T < 0 || d <= 0 || (T && (!triples || !nodes || !rel || !scores)) || (rank_counts != nullptr) != (ranks != nullptr) ||
      (rank_counts && (n_nodes <= 0 || 2 * T > INT32_MAX))) { rgcn_set_error("distmult_fwd: bad argument"); return RGCN_EINVAL; }
  if ((sbias != nullptr) != (pbias != nullptr) || (sbias != nullptr) != (obias != nullptr)) { rgcn_set_error("distmult_fwd: biases must be all set or all NULL"); return RGCN_EINVAL; }
  if (err_flag) HIP_TRY(zero_async(err_flag, sizeof(int32_t), (hipStream_t)stream));
  if (rank_counts) HIP_TRY(zero_async(rank_counts, (size_t)(2 * n_nodes + 3) * sizeof(int32_t), (hipStream_t)stream));
  if (T == 0) return RGCN_OK;
  const unsigned gx = (unsigned)std::min<int64_t>((T + 3) / 4, 256 * 32);
  hipLaunchKernelGGL(distmult_fwd_kernel, dim3(gx), dim3(WG), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long *>(triples), (long long)T, nodes, rel, sbias, pbias, obias,
                     scores, d, (long long)n_nodes, n_rel, err_flag, rank_counts, ranks,
                     (int)((d & 3) == 0 && ((reinterpret_cast<uintptr_t>(nodes) | reinterpret_cast<uintptr_t>(rel)) & 15) == 0));
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_distmult_bwd_f32(const int64_t *triples, int64_t T, const float *nodes, const float *rel,
                                     const float *gs, float *dnodes, float *drel, float *dsbias, float *dpbias,
                                     float *dobias, int64_t n_nodes, int32_t n_rel, int32_t d, void *stream) {
  if (T < 0 || d <= 0 || !drel || (T && (!triples || !nodes || !rel || !gs))) { rgcn_set_error("distmult_bwd: bad argument"); return RGCN_EINVAL; }
  hipStream_t st = (hipStream_t)stream;
  // dnodes == NULL: the entity gradients are computed by rgcn_distmult_bwd_nodes_f32 (no atomics); this call then yields the
  // relation (and bias) gradients only
  if (dnodes) HIP_TRY(zero_async(dnodes, (size_t)n_nodes * d * sizeof(float), st));
  HIP_TRY(zero_async(drel, (size_t)n_rel * d * sizeof(float), st));
  if (dsbias) {
    HIP_TRY(zero_async(dsbias, (size_t)n_nodes * sizeof(float), st));
    HIP_TRY(zero_async(dobias, (size_t)n_nodes * sizeof(float), st));
    HIP_TRY(zero_async(dpbias, (size_t)n_rel * sizeof(float), st));
  }
  if (T == 0) return RGCN_OK;
  const unsigned gx = (unsigned)std::min<int64_t>((T + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(distmult_bwd_kernel, dim3(gx), dim3(WG), 0, st, reinterpret_cast<const long long *>(triples),
                     (long long)T, nodes, rel, gs, dnodes, drel, dsbias, dpbias, dobias, d, (long long)n_nodes, n_rel, dnodes ? 0 : 1);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_distmult_bwd_all_supported(int32_t n_rel, int32_t d) { return n_rel > 0 && d > 0 && (int64_t)n_rel * (d + 1) <= 4096; }

extern "C" int rgcn_distmult_bwd_all_f32(const int32_t *rowptr_s, const int32_t *rowptr_o, const int32_t *entries, const float *nodes,
                                         const float *rel, float *dnodes, float *drel, float *dsbias, float *dpbias, float *dobias,
                                         int64_t n_nodes, int32_t n_rel, int32_t d, void *stream) {
  if (!rowptr_s || !rowptr_o || !nodes || !rel || !dnodes || !drel || n_nodes < 0 || d <= 0 || n_rel <= 0) { rgcn_set_error("distmult_bwd_all: bad argument"); return RGCN_EINVAL; }
  if ((dsbias != nullptr) != (dpbias != nullptr) || (dsbias != nullptr) != (dobias != nullptr)) { rgcn_set_error("distmult_bwd_all: bias gradients must be all set or all NULL"); return RGCN_EINVAL; }
  if (!rgcn_distmult_bwd_all_supported(n_rel, d)) { rgcn_set_error("distmult_bwd_all: n_rel (d + 1) = %lld floats do not fit the LDS table (4096)", (long long)n_rel * (d + 1)); return RGCN_EUNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(zero_async(drel, (size_t)n_rel * d * sizeof(float), st));
  if (dpbias) HIP_TRY(zero_async(dpbias, (size_t)n_rel * sizeof(float), st));
  if (!n_nodes) return RGCN_OK;
  const unsigned gx = (unsigned)std::min<int64_t>((n_nodes + 3) / 4, 1024);        // (29 KB of LDS per workgroup at WN18's size: up to 5 per CU)
  const size_t lds = ((size_t)n_rel * 4 * ((d + 3) / 4) + n_rel) * sizeof(double);
  const unsigned gx2 = (unsigned)std::min<int64_t>((n_nodes + 3) / 4, 256 * 32);
  const int4 *en = reinterpret_cast<const int4 *>(entries);
  // two launches (object side without LDS at full occupancy, then subject side + relation table): the one-launch form was measured slower
  auto go = [&](auto k2, auto k1) {
    hipLaunchKernelGGL(k2, dim3(gx2), dim3(WG), 0, st, rowptr_s, rowptr_o, en, nodes, rel, dnodes, drel, dsbias, dpbias, dobias,
                       (long long)n_nodes, n_rel, d);
    hipLaunchKernelGGL(k1, dim3(gx), dim3(WG), lds, st, rowptr_s, rowptr_o, en, nodes, rel, dnodes, drel, dsbias, dpbias, dobias,
                       (long long)n_nodes, n_rel, d);
  };
  if (d % 4 == 0) go(distmult_bwd_all_kernel<2, true, true>, distmult_bwd_all_kernel<1, true, true>);
  else go(distmult_bwd_all_kernel<2, false, true>, distmult_bwd_all_kernel<1, false, true>);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_distmult_bwd_nodes_f32(const int32_t *rowptr_s, const int32_t *rowptr_o, const int32_t *entries, const float *nodes,
                                           const float *rel, float *dnodes, int64_t n_nodes, int32_t d, void *stream) {
  if (!rowptr_s || !rowptr_o || !nodes || !rel || !dnodes || n_nodes < 0 || d <= 0) { rgcn_set_error("distmult_bwd_nodes: bad argument"); return RGCN_EINVAL; }
  if (!n_nodes) return RGCN_OK;
  const unsigned gx = (unsigned)std::min<int64_t>((n_nodes + 3) / 4, 256 * 32);
  const int4 *en = reinterpret_cast<const int4 *>(entries);
  auto go = [&](auto k) {
    hipLaunchKernelGGL(k, dim3(gx), dim3(WG), 0, (hipStream_t)stream, rowptr_s, rowptr_o, en, nodes, rel, dnodes, (float *)nullptr,
                       (float *)nullptr, (float *)nullptr, (float *)nullptr, (long long)n_nodes, 0, d);
  };
  if (d % 4 == 0) go(distmult_bwd_all_kernel<3, true, false>);
  else go(distmult_bwd_all_kernel<3, false, false>);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}
