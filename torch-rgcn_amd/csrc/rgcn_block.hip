// Block-diagonal per-relation weights (decomposition {type: block}; reference layers.py:176-183 parameters, :243-244 /
// :520-527 `block_diag(self.blocks)` expanded to dense R x d_in x d_out, then the dense message passing).
//
// Here the blocks are never expanded: a message is a row of nb segments, each multiplied by its own bi x bo block --
// 1/nb of the flops of the dense product and no R x d x d tensor.
//
//   block_csr_kernel   out[row, b, :] = bias + sum over the row's messages of val * X[src, b, :] . W[rel, b]      (forward; with
//                      the blocks read transposed and the transposed CSR: the feature gradient)
//   block_wgrad_kernel dW[rel, b] = sum over the messages of rel of val * X[src, b, :]^T G[dst, b, :]           (relation-major)
//
// Both are gather-bound (HBM): per message d_in * 4 bytes of features + 12 bytes of indices, the blocks (R nb bi bo floats:
// 68 KB for AM at width 16, 4.75 MB for FB15k-237 at width 500) come from L2.  Destination-major CSR rather than the
// (tile, relation) buckets of the dense kernels: nothing is shared between the messages of a bucket here, and with
// hundreds of relations the buckets of a small tile are nearly all padding.
//
// Lane layout (both kernels): `lpm` lanes per message, one block each (lpm = the number of blocks rounded up to a power of
// two, at most 64; wider rows loop); the remaining 64 / lpm (forward: lr / lpm within a unit) lane groups hold different
// messages and are summed with wave shuffles at the end.  gfx950 only.
#include <stdlib.h>

#include <algorithm>

#include "rgcn_device.h"

namespace {

constexpr int MAXB = 8;      // runtime-sized blocks up to 8 x 8 (larger blocks: the gather-GEMM of rgcn_gemm.hip is the better tool)

// BI_ x BO_ = the block as stored ([bi][bo] row-major); TR: multiply by the transposed block (input width bo, output bi).
// BI_ = 0: sizes at run time (bi, bo <= MAXB).
// one work unit (a row, or a piece of a hub row) on `lr` lanes; W = the block table in global memory or in LDS
template <int BI_, int BO_, bool TR>
__device__ __forceinline__ void block_unit(
    const float *__restrict__ X, const float *W, const float *__restrict__ bias, float *__restrict__ out,
    const int4 *__restrict__ units, const int *__restrict__ rowptr, long long u, long long n_units,
    const int *__restrict__ e_src, const int *__restrict__ e_rel, const float *__restrict__ e_val, int nb, int bi_rt, int bo_rt,
    int n_rel_blocks, int lpm, int lr, int relu, int rel_stride) {
  constexpr bool FIXED = BI_ > 0;
  constexpr int CAP_IN = FIXED ? (TR ? BO_ : BI_) : MAXB, CAP_OUT = FIXED ? (TR ? BI_ : BO_) : MAXB;
  const int bi = FIXED ? BI_ : bi_rt, bo = FIXED ? BO_ : bo_rt;
  const int n_in = TR ? bo : bi, n_out = TR ? bi : bo;          // floats per block on the input / output side
  const int sub = threadIdx.x % lr, g = sub / lpm, j = sub % lpm, gpr = lr / lpm;
  const bool on = u < n_units;
  int4 unit = {0, 0, 0, 0};
  if (on) {
    if (units) unit = units[u];
    else unit = int4{(int)u, rowptr[u], rowptr[u + 1], 0};
  }
  const int e1 = unit.z;
  const bool shared = unit.w & RGCN_U_SHARED;
  const bool add_bias = bias && (!shared || (unit.w & RGCN_U_FIRST));
  const size_t d_in = (size_t)nb * n_in, d_out = (size_t)nb * n_out;
  const bool vec_in = FIXED && (CAP_IN % 4 == 0), vec_w = FIXED && ((BI_ * BO_) % 4 == 0);
  for (int b0 = 0; b0 < nb; b0 += lpm) {
    const int b = b0 + j;
    float acc[CAP_OUT];
#pragma unroll
    for (int o = 0; o < CAP_OUT; ++o) acc[o] = 0.f;
    if (b < nb) {
      // one message: indices -> feature segment + block -> acc.  Two messages of the unit per lane group are in flight (their
      // index loads and row gathers are independent: a row of <= 2 gpr messages -- AM: 8 on average, gpr = 4 -- costs ONE chain of
      // index latency + gather latency instead of two; measured at AM scale: 0.455 -> see DESIGN 4.3)
      auto one = [&](int e, bool have) {
        const int ee = have ? e : unit.y;
        const int rel = e_rel[ee];
        const float v = (have && rel < n_rel_blocks) ? e_val[ee] : 0.f;        // relations past the block table (LP self loops): not ours
        const float *xr = X + (size_t)e_src[ee] * d_in + (size_t)b * n_in;
        const float *wr = W + (size_t)(rel < n_rel_blocks ? rel : 0) * rel_stride + (size_t)b * (bi * bo);
        float x[CAP_IN], w[FIXED ? BI_ * BO_ : MAXB * MAXB];
        if (vec_in) {
#pragma unroll
          for (int i = 0; i < CAP_IN / 4; ++i) {
            const f32x4 t = reinterpret_cast<const f32x4 *>(xr)[i];
            x[4 * i] = t[0] * v; x[4 * i + 1] = t[1] * v; x[4 * i + 2] = t[2] * v; x[4 * i + 3] = t[3] * v;
          }
        } else {
#pragma unroll
          for (int i = 0; i < CAP_IN; ++i) x[i] = (FIXED || i < n_in) ? xr[i] * v : 0.f;
        }
        if (vec_w) {
#pragma unroll
          for (int i = 0; i < (FIXED ? BI_ * BO_ : 4) / 4; ++i) {
            const f32x4 t = reinterpret_cast<const f32x4 *>(wr)[i];
            w[4 * i] = t[0]; w[4 * i + 1] = t[1]; w[4 * i + 2] = t[2]; w[4 * i + 3] = t[3];
          }
        } else if (FIXED) {
#pragma unroll
          for (int i = 0; i < BI_ * BO_; ++i) w[i] = wr[i];
        }
        if (FIXED) {
#pragma unroll
          for (int o = 0; o < CAP_OUT; ++o)
#pragma unroll
            for (int i = 0; i < CAP_IN; ++i) acc[o] = fmaf(x[i], TR ? w[o * BO_ + i] : w[i * BO_ + o], acc[o]);
        } else {
#pragma unroll
          for (int o = 0; o < MAXB; ++o)
            if (o < n_out) {
#pragma unroll
              for (int i = 0; i < MAXB; ++i)
                if (i < n_in) acc[o] = fmaf(x[i], TR ? wr[o * bo + i] : wr[i * bo + o], acc[o]);
            }
        }
      };
      for (int e = unit.y + g; e < e1; e += 2 * gpr) {
        one(e, true);
        one(e + gpr, e + gpr < e1);
      }
    }
    for (int s = lpm; s < lr; s *= 2) {          // sum over the message groups of the unit (same trip count in every lane)
#pragma unroll
      for (int o = 0; o < CAP_OUT; ++o) acc[o] += __shfl_xor(acc[o], s, 64);
    }
    if (on && g == 0 && b < nb) {
      float *orow = out + (size_t)unit.x * d_out + (size_t)b * n_out;
#pragma unroll
      for (int o = 0; o < CAP_OUT; ++o)
        if (FIXED || o < n_out) {
          float r = acc[o] + (add_bias ? bias[(size_t)b * n_out + o] : 0.f);
          if (shared) {
            atomicAdd(orow + o, r);
          } else {
            if (relu) r = fmaxf(r, 0.f);
            orow[o] = r;
          }
        }
    }
  }
}

template <int BI_, int BO_, bool TR>
__global__ __launch_bounds__(WG) void block_csr_kernel(
    const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias, float *__restrict__ out,
    const int4 *__restrict__ units, const int *__restrict__ rowptr, long long n_units, const int *__restrict__ e_src,
    const int *__restrict__ e_rel, const float *__restrict__ e_val, int nb, int bi_rt, int bo_rt, int n_rel_blocks, int lpm,
    int lr, int relu) {
  block_unit<BI_, BO_, TR>(X, W, bias, out, units, rowptr, ((long long)blockIdx.x * WG + threadIdx.x) / lr, n_units, e_src, e_rel,
                           e_val, nb, bi_rt, bo_rt, n_rel_blocks, lpm, lr, relu, nb * (BI_ > 0 ? BI_ * BO_ : bi_rt * bo_rt));
}

// The same with the WHOLE block table resident in LDS (tables up to LDS_TABLE_BYTES: AM at width 16 is 267 x 4 x 16 floats =
// 68 KB): the per-message block reads (4x the bytes of the feature row at 4 x 4) leave L2 alone.  Persistent 1024-thread
// workgroups (two per CU), each loads the table once and walks the units grid-stride.
constexpr int BIG_WG = 1024;
constexpr size_t LDS_TABLE_BYTES = 76 * 1024;

template <int BI_, int BO_, bool TR>
__global__ __launch_bounds__(BIG_WG) void block_csr_lds_kernel(
    const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias, float *__restrict__ out,
    const int4 *__restrict__ units, const int *__restrict__ rowptr, long long n_units, const int *__restrict__ e_src,
    const int *__restrict__ e_rel, const float *__restrict__ e_val, int nb, int bi_rt, int bo_rt, int n_rel_blocks, int lpm,
    int lr, int relu, int table_floats) {
  extern __shared__ __attribute__((aligned(16))) float wt[];
  // a relation's blocks are padded by 4 floats: unpadded, every relation starts on bank 0 (nb x 16 floats = a multiple of the
  // 64 banks at nb = 4) and the 16 messages of a ds_read_b128 collide 16 ways
  const int per_rel = nb * BI_ * BO_, rel_stride = per_rel + 4;
  for (int i = threadIdx.x; i < table_floats; i += BIG_WG) wt[(i / per_rel) * rel_stride + i % per_rel] = W[i];
  __syncthreads();
  const int upw = BIG_WG / lr;
  for (long long base = (long long)blockIdx.x * upw; base < n_units; base += (long long)gridDim.x * upw)
    block_unit<BI_, BO_, TR>(X, wt, bias, out, units, rowptr, base + threadIdx.x / lr, n_units, e_src, e_rel, e_val, nb, bi_rt,
                             bo_rt, n_rel_blocks, lpm, lr, relu, rel_stride);
}

// The common case by hand: 4 blocks of 4 x 4 (width 16), block table in LDS, work units given.  Same arithmetic as
// block_csr_lds_kernel<4, 4, TR> with the dependent load chain of a row -- unit -> (rel, src, val) -> feature segment -- software
// pipelined across the persistent loop: the unit of the row after next and the indices of the next row are requested while the
// current row's four feature segments (four messages per lane group in flight: rows of up to 16 messages in one pass) are on
// their way.  16 lanes per row: lane (g, j) = message group g (0..3), block j (0..3).
template <bool TR>
__global__ __launch_bounds__(BIG_WG) void block44_csr_kernel(
    const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias, float *__restrict__ out,
    const int4 *__restrict__ units, long long n_units, const int *__restrict__ e_src, const int *__restrict__ e_rel,
    const float *__restrict__ e_val, int n_rel_blocks, int relu, int table_floats) {
  extern __shared__ __attribute__((aligned(16))) float wt[];
  constexpr int PER_REL = 64, REL_STRIDE = 68;          // (the 4-float pad: see block_csr_lds_kernel)
  for (int i = threadIdx.x; i < table_floats; i += BIG_WG) wt[(i / PER_REL) * REL_STRIDE + i % PER_REL] = W[i];
  __syncthreads();
  constexpr int UPW = BIG_WG / 16, MF = 4;              // rows per workgroup pass, messages in flight per lane group
  const int sub = threadIdx.x & 15, g = sub >> 2, j = sub & 3;
  const long long stride = (long long)gridDim.x * UPW;
  long long u = (long long)blockIdx.x * UPW + (threadIdx.x >> 4);
  struct Idx { int rel[MF], src[MF]; float v[MF]; };
  auto load_unit = [&](long long uu) { return uu < n_units ? units[uu] : int4{0, 0, 0, 0}; };
  // the three index loads of a message are UNCONDITIONAL and their sanitising happens where they are waited for (finish_idx): with
  // `val = (relation in the table) ? e_val[e] : 0` the val load hung on the relation's arrival -- four dependent round trips per row
  // (round 5, from the ISA: s_waitcnt vmcnt(1) after every relation load; the kernel ran at ~7 us per workgroup pass of 64 rows)
  auto load_idx = [&](const int4 &un, Idx &ix) {
#pragma unroll
    for (int q = 0; q < MF; ++q) {
      const int e = un.y + g + 4 * q;
      const int ee = e < un.z ? e : 0;
      ix.rel[q] = e_rel[ee];
      ix.src[q] = e_src[ee];
      ix.v[q] = e_val[ee];
    }
  };
  auto finish_idx = [&](const int4 &un, Idx &ix) {
#pragma unroll
    for (int q = 0; q < MF; ++q) {
      const bool ours = un.y + g + 4 * q < un.z && ix.rel[q] < n_rel_blocks;      // relations past the block table (LP self loops): not ours
      ix.rel[q] = ix.rel[q] < n_rel_blocks ? ix.rel[q] : 0;
      ix.v[q] = ours ? ix.v[q] : 0.f;
    }
  };
  auto mac = [&](f32x4 &acc, const f32x4 &x, float v, int rel) {
    const f32x4 *w = reinterpret_cast<const f32x4 *>(wt + rel * REL_STRIDE + j * 16);
    const f32x4 w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];     // block rows i = 0..3: w_i[o]
    const float x0 = x[0] * v, x1 = x[1] * v, x2 = x[2] * v, x3 = x[3] * v;
    if (TR) {       // out[i] += sum_o x[o] * w[i][o]
      acc[0] += x0 * w0[0] + x1 * w0[1] + x2 * w0[2] + x3 * w0[3];
      acc[1] += x0 * w1[0] + x1 * w1[1] + x2 * w1[2] + x3 * w1[3];
      acc[2] += x0 * w2[0] + x1 * w2[1] + x2 * w2[2] + x3 * w2[3];
      acc[3] += x0 * w3[0] + x1 * w3[1] + x2 * w3[2] + x3 * w3[3];
    } else {        // out[o] += sum_i x[i] * w[i][o]
      acc += w0 * x0 + w1 * x1 + w2 * x2 + w3 * x3;
    }
  };
  int4 un_c = load_unit(u), un_n = load_unit(u + stride);
  Idx ix_c, ix_n;
  load_idx(un_c, ix_c);
  finish_idx(un_c, ix_c);
  const f32x4 bias4 = bias ? *reinterpret_cast<const f32x4 *>(bias + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};      // (once: a load ahead of every row's store would be waited for there)
  const long long u_first = (long long)blockIdx.x * UPW;      // uniform loop bound for the whole workgroup (shuffles below)
  // The row's gathers are issued at the END of the row before, AHEAD of that row's store (round 5, from the ISA): vmcnt counts in order, so a
  // wait for loads issued after a store also waits for the store to complete -- with the gathers behind the store every row paid a write
  // round trip before its first product.  Per iteration now: prefetch (unit two ahead, indices one ahead) | products of this row | wait for
  // the prefetch | NEXT row's gathers | this row's store.
  f32x4 x[MF];
#pragma unroll
  for (int q = 0; q < MF; ++q) x[q] = *reinterpret_cast<const f32x4 *>(X + (size_t)ix_c.src[q] * 16 + 4 * j);
  for (long long base = u_first; base < n_units; base += stride, u += stride) {
    const int4 un_nn = load_unit(u + 2 * stride);
    load_idx(un_n, ix_n);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < MF; ++q) mac(acc, x[q], ix_c.v[q], ix_c.rel[q]);
    for (int e = un_c.y + g + 4 * MF; e < un_c.z; e += 4) {      // rows of more than 16 messages: the rest, unpipelined
      const int r = e_rel[e];
      const float v = r < n_rel_blocks ? e_val[e] : 0.f;
      const f32x4 xx = *reinterpret_cast<const f32x4 *>(X + (size_t)e_src[e] * 16 + 4 * j);
      mac(acc, xx, v, r < n_rel_blocks ? r : 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { acc[q] += __shfl_xor(acc[q], 4, 64); }
#pragma unroll
    for (int q = 0; q < 4; ++q) { acc[q] += __shfl_xor(acc[q], 8, 64); }
    int4 un_nn_p = un_nn;
    asm volatile("" : "+v"(un_nn_p.x), "+v"(un_nn_p.y), "+v"(un_nn_p.z), "+v"(un_nn_p.w));
#pragma unroll
    for (int q = 0; q < MF; ++q) asm volatile("" : "+v"(ix_n.rel[q]), "+v"(ix_n.src[q]), "+v"(ix_n.v[q]));
    finish_idx(un_n, ix_n);
#pragma unroll
    for (int q = 0; q < MF; ++q) x[q] = *reinterpret_cast<const f32x4 *>(X + (size_t)ix_n.src[q] * 16 + 4 * j);      // (past the end: row 0, unused)
    __builtin_amdgcn_sched_barrier(0);
    if (u < n_units && g == 0) {
      const bool shared = un_c.w & RGCN_U_SHARED;
      const bool add_bias = bias && (!shared || (un_c.w & RGCN_U_FIRST));
      float *orow = out + (size_t)un_c.x * 16 + 4 * j;
      if (add_bias) acc += bias4;
      if (shared) {
#pragma unroll
        for (int q = 0; q < 4; ++q) atomicAdd(orow + q, acc[q]);
      } else {
        if (relu) { acc[0] = fmaxf(acc[0], 0.f); acc[1] = fmaxf(acc[1], 0.f); acc[2] = fmaxf(acc[2], 0.f); acc[3] = fmaxf(acc[3], 0.f); }
        *reinterpret_cast<f32x4 *>(orow) = acc;
      }
    }
    un_c = un_n; un_n = un_nn_p; ix_c = ix_n;
  }
}

// Dense 16 x 16 weights on a destination-major CSR ("chunks of messages with MIXED relations, W fetched per slot from
// an LDS-resident R x 1 KiB table"): the forward of a hidden-16 layer whose (tile, relation) buckets are sparse, in ONE pass --
// out[row, j] = bias[j] + sum over the row's messages of val * sum_i X[src, i] W[rel, i, j] -- instead of the two-pass route
// (transform in relation-major order into an M x 64 B buffer, then sum per destination).  16 lanes per row (lane j = output feature
// j), the row's messages four at a time: lane i loads X[src][i] (one 64-byte row per message), the 16 products per message are
// v_fmac with a DPP row_share operand (x_i broadcast inside the 16 lanes) against W[rel][.][j], which sits TRANSPOSED in LDS
// ([rel][j][i], row stride 20 floats: the four ds_read_b128 of a lane are conflict-free).  R * 1.25 KiB of LDS: R <= 120; the
// per-message matrix-vector product runs on the vector ALU (256 FMAs; a small graph's few hundred thousand messages are nothing),
// dense buckets stay on the MFMA tile kernel.  Persistent 1024-thread workgroups, one per CU.
constexpr int CSRW_LD = 20, CSRW_REL = 16 * CSRW_LD;      // floats per relation in LDS
__global__ __launch_bounds__(BIG_WG) void spmm_csr_d16_kernel(
    const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias, float *__restrict__ out,
    const int4 *__restrict__ units, long long n_units, const int *__restrict__ e_src, const int *__restrict__ e_rel,
    const float *__restrict__ e_val, int R, int relu) {
  extern __shared__ __attribute__((aligned(16))) float wt[];
  for (int t = threadIdx.x; t < R * 256; t += BIG_WG) {       // W[rel][i][j] -> wt[rel][j][i]
    const int rel = t >> 8, i = (t >> 4) & 15, j = t & 15;
    wt[rel * CSRW_REL + j * CSRW_LD + i] = W[t];
  }
  __syncthreads();
  constexpr int UPW = BIG_WG / 16, MF = 4;
  const int j = threadIdx.x & 15;
  const long long stride = (long long)gridDim.x * UPW;
  for (long long u = (long long)blockIdx.x * UPW + (threadIdx.x >> 4); u < n_units; u += stride) {
    const int4 un = units[u];
    float acc = 0.f;
    for (int e0 = un.y; e0 < un.z; e0 += MF) {
      float x[MF];
      int rel[MF];
#pragma unroll
      for (int q = 0; q < MF; ++q) {
        const int e = min(e0 + q, un.z - 1);
        rel[q] = e_rel[e];
        x[q] = X[(size_t)e_src[e] * 16 + j] * (e0 + q < un.z ? e_val[e] : 0.f);
      }
#pragma unroll
      for (int q = 0; q < MF; ++q) {
        const f32x4 *w4 = reinterpret_cast<const f32x4 *>(wt + rel[q] * CSRW_REL + j * CSRW_LD);
        const f32x4 w0 = w4[0], w1 = w4[1], w2 = w4[2], w3 = w4[3];
        const int xi = __builtin_bit_cast(int, x[q]);
#define RGCN_XS(I) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xi, 0x150 + (I), 0xF, 0xF, false))
        acc += RGCN_XS(0) * w0[0] + RGCN_XS(1) * w0[1] + RGCN_XS(2) * w0[2] + RGCN_XS(3) * w0[3];
        acc += RGCN_XS(4) * w1[0] + RGCN_XS(5) * w1[1] + RGCN_XS(6) * w1[2] + RGCN_XS(7) * w1[3];
        acc += RGCN_XS(8) * w2[0] + RGCN_XS(9) * w2[1] + RGCN_XS(10) * w2[2] + RGCN_XS(11) * w2[3];
        acc += RGCN_XS(12) * w3[0] + RGCN_XS(13) * w3[1] + RGCN_XS(14) * w3[2] + RGCN_XS(15) * w3[3];
#undef RGCN_XS
      }
    }
    const bool shared = un.w & RGCN_U_SHARED;
    if (bias && (!shared || (un.w & RGCN_U_FIRST))) acc += bias[j];
    float *o = out + (size_t)un.x * 16 + j;
    if (shared) atomicAdd(o, acc);
    else *o = relu ? fmaxf(acc, 0.f) : acc;
  }
}

// One wave per work item (a chunk range of ONE relation in the relation-major plan; pads carry val = 0).
template <int BI_, int BO_>
__global__ __launch_bounds__(WG) void block_wgrad_kernel(
    const float *__restrict__ X, const float *__restrict__ G, float *__restrict__ dW, const int *__restrict__ p_src,
    const int *__restrict__ p_dst, const float *__restrict__ p_val, const int *__restrict__ chunk_rel,
    const int2 *__restrict__ items, long long n_items, int nb, int bi_rt, int bo_rt, int n_rel_blocks, int lpm) {
  constexpr bool FIXED = BI_ > 0;
  constexpr int CAP_I = FIXED ? BI_ : MAXB, CAP_O = FIXED ? BO_ : MAXB;
  const int bi = FIXED ? BI_ : bi_rt, bo = FIXED ? BO_ : bo_rt;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long it = (long long)blockIdx.x * (WG / 64) + wave;
  if (it >= n_items) return;
  const int2 range = items[it];
  if (range.x >= range.y) return;
  const int rel = chunk_rel[range.x];
  if (rel >= n_rel_blocks) return;
  const int g = lane / lpm, j = lane % lpm, groups = 64 / lpm;
  const size_t d_in = (size_t)nb * bi, d_out = (size_t)nb * bo;
  const int s0 = range.x * RGCN_CHUNK, s1 = range.y * RGCN_CHUNK;
  const bool vec_i = FIXED && (BI_ % 4 == 0), vec_o = FIXED && (BO_ % 4 == 0);
  for (int b0 = 0; b0 < nb; b0 += lpm) {
    const int b = b0 + j;
    float acc[CAP_I][CAP_O];
#pragma unroll
    for (int i = 0; i < CAP_I; ++i)
#pragma unroll
      for (int o = 0; o < CAP_O; ++o) acc[i][o] = 0.f;
    if (b < nb) {
      // branch-free body (pads: val = 0, src = 0, dst = -1 -> row 0), two slots in flight per lane
      auto rows = [&](int s, bool have, float (&x)[CAP_I], float (&gg)[CAP_O]) {
        const int ss = have ? s : s0;
        const float v = have ? p_val[ss] : 0.f;
        const float *xr = X + (size_t)p_src[ss] * d_in + (size_t)b * bi;
        const float *gr = G + (size_t)max(p_dst[ss], 0) * d_out + (size_t)b * bo;
        if (vec_i) {
#pragma unroll
          for (int i = 0; i < CAP_I / 4; ++i) {
            const f32x4 t = reinterpret_cast<const f32x4 *>(xr)[i];
            x[4 * i] = t[0] * v; x[4 * i + 1] = t[1] * v; x[4 * i + 2] = t[2] * v; x[4 * i + 3] = t[3] * v;
          }
        } else {
#pragma unroll
          for (int i = 0; i < CAP_I; ++i) x[i] = (FIXED || i < bi) ? xr[i] * v : 0.f;
        }
        if (vec_o) {
#pragma unroll
          for (int o = 0; o < CAP_O / 4; ++o) {
            const f32x4 t = reinterpret_cast<const f32x4 *>(gr)[o];
            gg[4 * o] = t[0]; gg[4 * o + 1] = t[1]; gg[4 * o + 2] = t[2]; gg[4 * o + 3] = t[3];
          }
        } else {
#pragma unroll
          for (int o = 0; o < CAP_O; ++o) gg[o] = (FIXED || o < bo) ? gr[o] : 0.f;
        }
      };
      for (int s = s0 + g; s < s1; s += 2 * groups) {
        float xa[CAP_I], ga[CAP_O], xb[CAP_I], gb[CAP_O];
        rows(s, true, xa, ga);
        rows(s + groups, s + groups < s1, xb, gb);
#pragma unroll
        for (int i = 0; i < CAP_I; ++i)
#pragma unroll
          for (int o = 0; o < CAP_O; ++o) acc[i][o] = fmaf(xb[i], gb[o], fmaf(xa[i], ga[o], acc[i][o]));
      }
    }
    for (int s = lpm; s < 64; s *= 2) {
#pragma unroll
      for (int i = 0; i < CAP_I; ++i)
#pragma unroll
        for (int o = 0; o < CAP_O; ++o) acc[i][o] += __shfl_xor(acc[i][o], s, 64);
    }
    if (g == 0 && b < nb) {
      float *w = dW + ((size_t)rel * nb + b) * (size_t)(bi * bo);
#pragma unroll
      for (int i = 0; i < CAP_I; ++i)
#pragma unroll
        for (int o = 0; o < CAP_O; ++o)
          if ((FIXED || (i < bi && o < bo)) && acc[i][o] != 0.f) atomicAdd(w + i * bo + o, acc[i][o]);
    }
  }
}

int lanes_per_message(int nb) {
  int lpm = 1;
  while (lpm < 64 && lpm < nb) lpm *= 2;
  return lpm;
}

}  // namespace

extern "C" int rgcn_block_supported(int32_t bi, int32_t bo) { return bi >= 1 && bo >= 1 && bi <= MAXB && bo <= MAXB; }

extern "C" int rgcn_block_spmm_f32(const float *X, const float *blocks, const float *bias, float *out, const int32_t *units,
                                   const int32_t *rowptr, int64_t n_units, int64_t n_split, const int32_t *e_src,
                                   const int32_t *e_rel, const float *e_val, int64_t n_rows, int32_t n_rel_blocks, int32_t nb,
                                   int32_t bi, int32_t bo, int32_t flags, void *stream) {
  if (!X || !blocks || !out || n_rows < 0 || n_units < 0 || n_split < 0 || nb <= 0 || n_rel_blocks < 0 || (!units && !rowptr) ||
      (n_units && (!e_src || !e_rel || !e_val))) {
    rgcn_set_error("block_spmm: bad argument");
    return RGCN_EINVAL;
  }
  if (!rgcn_block_supported(bi, bo)) {
    rgcn_set_error("block_spmm: blocks of %d x %d (limit %d x %d)", bi, bo, MAXB, MAXB);
    return RGCN_EUNSUPPORTED;
  }
  if (!units && n_units != n_rows) { rgcn_set_error("block_spmm: without units, one unit per row"); return RGCN_EINVAL; }
  const bool tr = flags & RGCN_F_TRANSPOSE_W, relu = flags & RGCN_F_RELU;
  if (relu && n_split) { rgcn_set_error("block_spmm: RGCN_F_RELU with shared units"); return RGCN_EINVAL; }
  if (n_rows == 0 || n_units == 0) return RGCN_OK;
  hipStream_t st = (hipStream_t)stream;
  const int n_out = tr ? bi : bo;
  if (n_split) HIP_TRY(zero_async(out, (size_t)n_rows * nb * n_out * sizeof(float), st));
  const int lpm = lanes_per_message(nb);
  const int lr = std::min(64, std::max(16, 2 * lpm));
  const int upw = WG / lr;
  const dim3 grid((unsigned)((n_units + upw - 1) / upw)), block(WG);
  const int4 *un = reinterpret_cast<const int4 *>(units);
  // table in LDS: worth it when the graph is large enough to amortise 512 table loads (and the table fits)
  const size_t table_bytes = (size_t)n_rel_blocks * nb * bi * bo * sizeof(float);
  const size_t lds_bytes = table_bytes + (size_t)n_rel_blocks * 4 * sizeof(float);          // + the per-relation pad
  if (bi == 4 && bo == 4 && lds_bytes <= LDS_TABLE_BYTES && n_units >= 64 * 1024) {
    const dim3 pgrid((unsigned)std::min<int64_t>(512, (n_units * lr + BIG_WG - 1) / BIG_WG));
    auto launch = [&](auto kern, bool &raised) -> hipError_t {
      if (lds_bytes > 64 * 1024 && !raised) {     // once per process and kernel (not a stream operation: keep it out of captures)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)LDS_TABLE_BYTES);
        if (e != hipSuccess) return e;
        raised = true;
      }
      hipLaunchKernelGGL(kern, pgrid, dim3(BIG_WG), lds_bytes, st, X, blocks, bias, out, un, rowptr, (long long)n_units, e_src,
                         e_rel, e_val, nb, bi, bo, n_rel_blocks, lpm, lr, (int)relu, (int)(table_bytes / sizeof(float)));
      return hipGetLastError();
    };
    static bool raised_t = false, raised_n = false, raised_pt = false, raised_pn = false;
    if (nb == 4 && un) {          // width 16: the software-pipelined form
      const dim3 pg((unsigned)std::min<int64_t>(512, (n_units + BIG_WG / 16 - 1) / (BIG_WG / 16)));
      auto launch_p = [&](auto kern, bool &raised) -> hipError_t {
        if (lds_bytes > 64 * 1024 && !raised) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_TABLE_BYTES);
          if (e != hipSuccess) return e;
          raised = true;
        }
        hipLaunchKernelGGL(kern, pg, dim3(BIG_WG), lds_bytes, st, X, blocks, bias, out, un, (long long)n_units, e_src, e_rel, e_val,
                           n_rel_blocks, (int)relu, (int)(table_bytes / sizeof(float)));
        return hipGetLastError();
      };
      if (tr) HIP_TRY(launch_p(block44_csr_kernel<true>, raised_pt));
      else HIP_TRY(launch_p(block44_csr_kernel<false>, raised_pn));
      return RGCN_OK;
    }
    if (tr) HIP_TRY(launch(block_csr_lds_kernel<4, 4, true>, raised_t));
    else HIP_TRY(launch(block_csr_lds_kernel<4, 4, false>, raised_n));
    return RGCN_OK;
  }
#define RGCN_BLOCK_LAUNCH(BI, BO)                                                                                              \
  do {                                                                                                                         \
    if (tr) hipLaunchKernelGGL((block_csr_kernel<BI, BO, true>), grid, block, 0, st, X, blocks, bias, out, un, rowptr,         \
                               (long long)n_units, e_src, e_rel, e_val, nb, bi, bo, n_rel_blocks, lpm, lr, (int)relu);         \
    else hipLaunchKernelGGL((block_csr_kernel<BI, BO, false>), grid, block, 0, st, X, blocks, bias, out, un, rowptr,           \
                            (long long)n_units, e_src, e_rel, e_val, nb, bi, bo, n_rel_blocks, lpm, lr, (int)relu);            \
  } while (0)
  if (bi == 4 && bo == 4) RGCN_BLOCK_LAUNCH(4, 4);
  else if (bi == 5 && bo == 5) RGCN_BLOCK_LAUNCH(5, 5);
  else if (bi == 8 && bo == 8) RGCN_BLOCK_LAUNCH(8, 8);
  else if (bi == 2 && bo == 2) RGCN_BLOCK_LAUNCH(2, 2);
  else RGCN_BLOCK_LAUNCH(0, 0);
#undef RGCN_BLOCK_LAUNCH
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_spmm_csr_d16_supported(int32_t R) { return R > 0 && (size_t)R * CSRW_REL * sizeof(float) <= 150 * 1024; }

extern "C" int rgcn_spmm_csr_d16_f32(const float *X, const float *W, const float *bias, float *out, const int32_t *units, int64_t n_units,
                                     int64_t n_split, const int32_t *e_src, const int32_t *e_rel, const float *e_val, int64_t n_rows,
                                     int32_t R, int32_t flags, void *stream) {
  if (!X || !W || !out || n_rows < 0 || n_units < 0 || n_split < 0 || R <= 0 || (n_units && (!units || !e_src || !e_rel || !e_val))) {
    rgcn_set_error("spmm_csr_d16: bad argument");
    return RGCN_EINVAL;
  }
  if (!rgcn_spmm_csr_d16_supported(R)) { rgcn_set_error("spmm_csr_d16: %d relations x 1.25 KiB exceed the LDS table (150 KiB)", R); return RGCN_EUNSUPPORTED; }
  const bool relu = flags & RGCN_F_RELU;
  if (relu && n_split) { rgcn_set_error("spmm_csr_d16: RGCN_F_RELU with shared units"); return RGCN_EINVAL; }
  if (n_rows == 0 || n_units == 0) return RGCN_OK;
  hipStream_t st = (hipStream_t)stream;
  if (n_split) HIP_TRY(zero_async(out, (size_t)n_rows * 16 * sizeof(float), st));
  const size_t lds = (size_t)R * CSRW_REL * sizeof(float);
  static bool raised = false;
  if (lds > 64 * 1024 && !raised) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(spmm_csr_d16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    raised = true;
  }
  const unsigned gx = (unsigned)std::min<int64_t>(256 * (lds <= 76 * 1024 ? 2 : 1), (n_units + BIG_WG / 16 - 1) / (BIG_WG / 16));
  hipLaunchKernelGGL(spmm_csr_d16_kernel, dim3(gx), dim3(BIG_WG), lds, st, X, W, bias, out, reinterpret_cast<const int4 *>(units),
                     (long long)n_units, e_src, e_rel, e_val, R, (int)relu);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_block_wgrad_f32(const float *X, const float *G, float *dblocks, const int32_t *p_src, const int32_t *p_dst,
                                    const float *p_val, const int32_t *chunk_rel, const int32_t *items, int64_t n_items,
                                    int32_t n_rel_blocks, int32_t nb, int32_t bi, int32_t bo, void *stream) {
  if (!X || !G || !dblocks || n_items < 0 || nb <= 0 || n_rel_blocks < 0 ||
      (n_items && (!items || !p_src || !p_dst || !p_val || !chunk_rel))) {
    rgcn_set_error("block_wgrad: bad argument");
    return RGCN_EINVAL;
  }
  if (!rgcn_block_supported(bi, bo)) {
    rgcn_set_error("block_wgrad: blocks of %d x %d (limit %d x %d)", bi, bo, MAXB, MAXB);
    return RGCN_EUNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  if (n_rel_blocks) HIP_TRY(zero_async(dblocks, (size_t)n_rel_blocks * nb * bi * bo * sizeof(float), st));
  if (n_items == 0 || n_rel_blocks == 0) return RGCN_OK;
  const int lpm = lanes_per_message(nb);
  const dim3 grid((unsigned)((n_items + WG / 64 - 1) / (WG / 64))), block(WG);
  const int2 *its = reinterpret_cast<const int2 *>(items);
#define RGCN_BLOCK_LAUNCH(BI, BO)                                                                                              \
  hipLaunchKernelGGL((block_wgrad_kernel<BI, BO>), grid, block, 0, st, X, G, dblocks, p_src, p_dst, p_val, chunk_rel, its,     \
                     (long long)n_items, nb, bi, bo, n_rel_blocks, lpm)
  if (bi == 4 && bo == 4) RGCN_BLOCK_LAUNCH(4, 4);
  else if (bi == 5 && bo == 5) RGCN_BLOCK_LAUNCH(5, 5);
  else if (bi == 8 && bo == 8) RGCN_BLOCK_LAUNCH(8, 8);
  else if (bi == 2 && bo == 2) RGCN_BLOCK_LAUNCH(2, 2);
  else RGCN_BLOCK_LAUNCH(0, 0);
#undef RGCN_BLOCK_LAUNCH
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}
