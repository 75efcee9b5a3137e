// Featureless layer with basis decomposition on a table far beyond the caches, IN the parameter's own [B, N, d] layout
// (layers.py:241-242 + :286-288; nc-AM.yaml: B = 40, N = 1.67 M, d = 10 -- a 2.67 GB table).  Round 4.
//
//   out[s,:] = sum_{e=(s,r,o)} val_e * sum_b comps[r,b] * bases[b,o,:]
//
// rgcn_basis.hip walks the messages source-major with ONE WAVE PER SOURCE NODE: unit -> block -> messages are three dependent round trips
// per node and nothing overlaps them (1.8 ms forward, 3.9 ms backward on AM as shipped, both latency-bound), and it wants the table
// node-major -- a transposed 2.67 GB copy per step and a transposed gradient back (3.2 ms of elementwise kernels).  Here:
//
//   * a TILE is 16 consecutive source nodes.  In [B, N, d] their blocks are B runs of 16 d floats = 64 d bytes each, aligned to whole
//     128-byte lines when d is even: the 1024 threads of a workgroup stage a tile with 16-byte loads into LDS (forward, dcomps) or write
//     its gradient out of LDS the same way (dbases) -- the table and its gradient stream through HBM once, in place, fully coalesced;
//   * workgroups are persistent (one per CU) and software-pipelined over their tiles: row pointers three tiles ahead, message indices
//     two, gathered gradient rows one, the tile itself two (one in registers in flight, one in LDS).  Everything issued in one
//     iteration is consumed at the TOP of the next (after a full tile of compute), so the conservative s_waitcnt the compiler places
//     there never waits for a load that was just issued; the barriers are LDS-only (lds_barrier: no vmcnt drain);
//   * a tile's messages (8 per node on AM) are dealt evenly over the 16 waves as contiguous ranges, so hub nodes spread over the
//     workgroup; the node of a message comes from a ballot over the tile's 17 row pointers.  A wave keeps the block (or block gradient) of
//     its current node in registers and exchanges it with LDS when the node changes;
//   * block gradients and dcomps are summed in LDS DOUBLES with ds_add_f64 -- the one native LDS float atomic on gfx950 (ds_add_f32
//     is serialised, 3 cycles per lane: profiles/r04_lds_cas_patterns.txt) -- lane = basis b, so the lanes of a quarter wave hit
//     distinct banks.  The backward is two kernels because of LDS: dbases needs the tile gradient in doubles (2 x 51 KB) next to the
//     coefficient table (43 KB), dcomps the R x B doubles (85 KB) next to two staged tiles (2 x 26 KB).
//
// Not bit-reproducible (arrival order of the LDS adds); the deterministic route stays in rgcn_basis.hip.
#include "rgcn_device.h"
#include <string.h>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int TW = 1024, TWV = TW / 64;     // threads / waves of a workgroup
constexpr int TN = 16;                      // nodes of a tile
constexpr int PER = 32;                     // messages of one wave per chunk (a chunk = 16 x PER messages of a tile)
constexpr int LDS_MAX = 160 * 1024;

__device__ __forceinline__ float bcast(float v, int src_lane) {   // src_lane wave-uniform
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}
__device__ __forceinline__ int rlane(int v, int src_lane) { return __builtin_amdgcn_readlane(v, src_lane); }
#define FBT_ARRIVED(x) asm volatile("" : "+v"(x))          // the value must be in its register HERE (pins the s_waitcnt)
#define FBT_KEEP(x) asm volatile("" ::"v"(x))               // the registers of x are not reused before HERE
#ifdef RGCN_ABLATIONS       // timing experiments with WRONG results (tools/fbt_bench.py, ablation build only): 1 no message loop, 2 no tile loads, 4 no row gathers, 8 no LDS adds
#define FBT_ABL(bit) (abl & (bit))
// the instrumented kernels also add up, per wave, the 100 MHz ticks of every phase of the pipeline (rgcn_fbt_debug_read)
#define FBT_T() ((long long)__builtin_amdgcn_s_memtime())
#define FBT_DBG(...) __VA_ARGS__
__device__ unsigned long long rgcn_fbt_dbg[8 * 256 * 16];
#else
#define FBT_ABL(bit) false
#define FBT_T() 0ll
#define FBT_DBG(...)
#endif

// x + y after exchanging halves (W = 32: lanes 32..63 of x with lanes 0..31 of y) or rows (W = 16: the odd 16-lane rows of x with the even
// rows of y): with x, y = the partial sums of two messages, the lower (even) part of the result belongs to x's message and the upper (odd)
// part to y's -- one level of a transpose-reduce in two VALU instructions (v_permlane32_swap / v_permlane16_swap, gfx950; no LDS).
template <int W>
__device__ __forceinline__ float swap_add(float x, float y) {
  // inline asm, not __builtin_amdgcn_permlane{32,16}_swap: hipcc 7.2 folds the builtin's two results into ONE register when they are added
  // (v_permlane32_swap v71, v73; v_add_f32 v58, v71, v71 -- tools/micro/permlane_swap.hip shows the instruction itself is fine).  The
  // s_nop cover the VALU -> permlane and permlane -> VALU hazards the compiler would otherwise pad.
  if (W == 32) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
  else asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
  return x + y;
}

// the part of a tile's messages one wave walks first; rp: lane l <= 16 holds rowptr[tile's first node + l]
struct Share { int mb, me, per, a, n; };
template <int PERW = PER>
__device__ __forceinline__ Share share_of(int rp, int wave, bool valid) {
  Share s;
  s.mb = rlane(rp, 0);
  s.me = rlane(rp, TN);
  s.per = min(PERW, (s.me - s.mb + TWV - 1) / TWV);
  s.a = s.mb + wave * s.per;
  s.n = valid ? max(0, min(s.per, s.me - s.a)) : 0;
  return s;
}
// tile-local node of message position p
__device__ __forceinline__ int node_of(int rp, int lane, int p) {
  return __popcll(__ballot(lane >= 1 && lane <= TN && rp <= p));
}

// staging geometry of one thread: its 16-byte pieces of a tile (piece = 4 consecutive floats of one basis' run)
template <int KLD>
struct Geo { long long goff[KLD]; int loff[KLD], pos[KLD]; bool act[KLD]; };
template <int KLD>
__device__ __forceinline__ void geo_init(Geo<KLD> &g, int tid, int B, int d, long long N, int ts, int tn = TN, int nthreads = TW) {
  const int q4 = tn * d / 4;                                // pieces per basis (tn nodes per tile)
#pragma unroll
  for (int k = 0; k < KLD; ++k) {
    const int idx = tid + k * nthreads;
    g.act[k] = idx < B * q4;                                 // (an inactive piece aliases piece 0: its loads stay unconditional)
    const int b = g.act[k] ? idx / q4 : 0, q = g.act[k] ? idx - b * q4 : 0;
    g.goff[k] = (long long)b * N * d + 4 * q;
    g.loff[k] = b * ts + 4 * q;
    g.pos[k] = 4 * q;
  }
}
template <int KLD, bool VEC>
__device__ __forceinline__ void stage_load(f32x4 (&st)[KLD], const Geo<KLD> &g, const float *__restrict__ bases, long long base) {
#pragma unroll
  for (int k = 0; k < KLD; ++k) {                            // no branches: a conditional load costs a phi, and the phi's temporary a wait
    const float *p = bases + g.goff[k] + base;
    if (VEC) st[k] = *reinterpret_cast<const f32x4 *>(p);
    else st[k] = f32x4{p[0], p[1], p[2], p[3]};
  }
}
template <int KLD, bool V4>
__device__ __forceinline__ void stage_store(float *buf, const f32x4 (&st)[KLD], const Geo<KLD> &g) {
#pragma unroll
  for (int k = 0; k < KLD; ++k)
    if (g.act[k]) {
      if (V4) *reinterpret_cast<f32x4 *>(buf + g.loff[k]) = st[k];
      else { buf[g.loff[k]] = st[k][0]; buf[g.loff[k] + 1] = st[k][1]; buf[g.loff[k] + 2] = st[k][2]; buf[g.loff[k] + 3] = st[k][3]; }
    }
}

// ================================================================== forward
// lane = (bg, i): feature i = lane % DP, basis group bg = lane / DP; the lane keeps block[bg * NREG + k][i], k < NREG, of the wave's current
// node; coefficient table in LDS as [R][BP = NREG * (64 / DP)], zero padded.  Y rows are DP floats wide (zero padded: rgcn_gather_rows_sum4_f32
// reads them 16 bytes per lane), written by the wave one iteration later from a wave-private LDS strip (coalesced, and issued before the
// iteration's loads so no wait ever includes them).
template <int DP, int NREG, int KLD, bool VEC>
__global__ __launch_bounds__(TW) void fbt_fwd_kernel(
    const float *__restrict__ bases, const float *__restrict__ comps, float *__restrict__ Y, const int *__restrict__ rowptr,
    const int *__restrict__ e_rel, const float *__restrict__ e_val, int n_tiles, int N, int R, int B, int d, int ts, int last, int abl) {
  constexpr int NGRP = 64 / DP, BP = NREG * NGRP;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (uniform: message counts stay in SGPRs)
  const int i = lane % DP, bg = lane / DP;
  float *ctab = lds;
  float *tb = lds + ((R * BP + 3) & ~3);
  float *yb = tb + 2 * B * ts + wave * (PER * DP);
  for (int j = tid; j < R * BP; j += TW) {
    const int r = j / BP, b = j % BP;
    ctab[j] = b < B ? comps[(size_t)r * B + b] : 0.f;
  }
  Geo<KLD> g;
  geo_init(g, tid, B, d, N, ts);
  const int G = gridDim.x;
  int t = blockIdx.x;
  auto rp_of = [&](int tt) {
    tt = min(tt, n_tiles - 1);
    return rowptr[min((long long)tt * TN + min(lane, TN), (long long)N)];
  };
  auto base_of = [&](int tt) { return (long long)min(min(tt, n_tiles - 1) * TN, N - TN) * d; };     // the last tile is staged from node N - 16

  f32x4 st[KLD];
  int rp = rp_of(t), rp1 = rp_of(t + G);
  stage_load<KLD, VEC>(st, g, bases, base_of(t));
  Share s = share_of(rp, wave, true);
  int er = e_rel[min(s.a + lane, last)];
  float ev = e_val[min(s.a + lane, last)];
  stage_store<KLD, true>(tb, st, g);
  stage_load<KLD, VEC>(st, g, bases, base_of(t + G));
  FBT_DBG(long long dbg[5] = {0, 0, 0, 0, 0};)
  int pa = 0, pn = 0;                                      // the wave's results waiting in yb: rows pa .. pa + pn
  auto flush = [&]() {
    f32x4 *dst = reinterpret_cast<f32x4 *>(Y + (size_t)pa * DP);
    const f32x4 *src = reinterpret_cast<const f32x4 *>(yb);
    for (int x = lane; x < pn * (DP / 4); x += 64) dst[x] = src[x];
    pn = 0;
  };
  lds_barrier();
  for (int k = 0;; ++k) {
    const bool has1 = t + G < n_tiles;
    const float *cb = tb + (k & 1) * (B * ts);
    FBT_DBG(const long long T0 = FBT_T();)
    // everything issued one iteration ago has had a tile of compute to arrive
    if (has1) stage_store<KLD, true>(tb + ((k + 1) & 1) * (B * ts), st, g);
    flush();
    __builtin_amdgcn_sched_barrier(0);
    FBT_DBG(const long long T1 = FBT_T();)
    if (!FBT_ABL(2)) stage_load<KLD, VEC>(st, g, bases, base_of(t + 2 * G));           // (past the end: the last tile again, never stored)
    const int rp2 = rp_of(t + 2 * G);
    const Share s1 = share_of(rp1, wave, has1);
    const int er1 = e_rel[min(s1.a + lane, last)];
    const float ev1 = e_val[min(s1.a + lane, last)];
    __builtin_amdgcn_sched_barrier(0);
    FBT_DBG(const long long T2 = FBT_T();)
    // ---- tile t
    const int shift = t * TN - min(t * TN, N - TN);
    int c_a = s.a, c_n = s.n, c_er = er;
    float c_ev = ev, blk[NREG];
    for (int c0 = s.mb;;) {
      // node by node: the wave's messages of one source node are a run; the node's block comes out of LDS once per run
      for (int p = c_a, end = FBT_ABL(1) ? c_a : c_a + c_n; p < end;) {
        const int nt = node_of(rp, lane, p);
        const int run_end = min(end, rlane(rp, nt + 1));
        const int nl = nt + shift;
#pragma unroll
        for (int q = 0; q < NREG; ++q)                       // (lanes without an element re-read a neighbour's: their coefficient is 0 / their column unused)
          blk[q] = cb[min(bg * NREG + q, B - 1) * ts + nl * d + min(i, d - 1)];
        while (p < run_end) {
          const int j0 = p - c_a, nv = min(DP == 16 ? 4 : 1, run_end - p);
          if (DP == 16) {
            // four messages at once (independent chains), then a transpose-reduce over the four basis groups: group g ends up with message g's sum
            float a4[4];
            constexpr int QB = (NREG <= 12 && KLD == 2) ? 4 : 2;     // messages whose coefficient reads fly together (registers)
#pragma unroll
            for (int h = 0; h < 4; h += QB) {
              f32x4 cq[QB][NREG / 4];
#pragma unroll
              for (int m = 0; m < QB; ++m) {                // (messages past the run: entries the wave holds anyway, results dropped)
                const f32x4 *c4 = reinterpret_cast<const f32x4 *>(ctab + rlane(c_er, j0 + h + m) * BP + bg * NREG);
#pragma unroll
                for (int q4 = 0; q4 < NREG / 4; ++q4) cq[m][q4] = c4[q4];
              }
              __builtin_amdgcn_sched_barrier(0);            // all of these reads in flight before the first FMA: one LDS round trip, not QB
#pragma unroll
              for (int m = 0; m < QB; ++m) {
                f32x2 e = {0.f, 0.f};                        // two floats per lane and instruction (v_pk_fma_f32)
#pragma unroll
                for (int q4 = 0; q4 < NREG / 4; ++q4) {
                  const f32x4 c = cq[m][q4];
                  e += f32x2{c[0], c[1]} * f32x2{blk[4 * q4], blk[4 * q4 + 1]};
                  e += f32x2{c[2], c[3]} * f32x2{blk[4 * q4 + 2], blk[4 * q4 + 3]};
                }
                a4[h + m] = (e[0] + e[1]) * bcast(c_ev, j0 + h + m);
              }
            }
            const float u = swap_add<32>(a4[0], a4[2]), w = swap_add<32>(a4[1], a4[3]);
            const float y = swap_add<16>(u, w);
            if (bg < nv) yb[(j0 + bg) * DP + i] = i < d ? y : 0.f;
          } else {
            const int r = rlane(c_er, j0);
            const f32x4 *c4 = reinterpret_cast<const f32x4 *>(ctab + r * BP + bg * NREG);
            float acc = 0.f;
#pragma unroll
            for (int q4 = 0; q4 < NREG / 4; ++q4) {
              const f32x4 c = c4[q4];
              acc += c[0] * blk[4 * q4] + c[1] * blk[4 * q4 + 1] + c[2] * blk[4 * q4 + 2] + c[3] * blk[4 * q4 + 3];
            }
#pragma unroll
            for (int off = DP; off < 64; off <<= 1) acc += __shfl_xor(acc, off, 64);
            if (bg == 0) yb[j0 * DP + i] = i < d ? bcast(c_ev, j0) * acc : 0.f;
          }
          p += nv;
        }
      }
      pa = c_a;
      pn = c_n;
      c0 += TWV * s.per;
      if (c0 >= s.me) break;
      flush();                                              // tiles with more than 16 x PER messages: further chunks, loaded on demand
      c_a = c0 + wave * s.per;
      c_n = max(0, min(s.per, min(c0 + TWV * s.per, s.me) - c_a));
      c_er = e_rel[min(c_a + lane, last)];
      c_ev = e_val[min(c_a + lane, last)];
      FBT_ARRIVED(c_er); FBT_ARRIVED(c_ev);                 // (the wait belongs HERE, on the rare path: left to the compiler it lands inside the message loop)
    }
    FBT_DBG(const long long T3 = FBT_T();)
    lds_barrier();
    FBT_DBG(const long long T4 = FBT_T(); dbg[0] += T1 - T0; dbg[1] += T2 - T1; dbg[2] += T3 - T2; dbg[3] += T4 - T3;)
    if (!has1) break;
    t += G;
    rp = rp1; rp1 = rp2; s = s1; er = er1; ev = ev1;
    FBT_ARRIVED(er); FBT_ARRIVED(ev); FBT_ARRIVED(rp1);
    FBT_DBG(dbg[4] += FBT_T() - T4;)
  }
  flush();
  FBT_DBG(if (lane == 0) { unsigned long long *o = rgcn_fbt_dbg + 8 * ((blockIdx.x & 255) * 16 + wave); for (int q = 0; q < 5; ++q) o[q] += dbg[q]; o[5] += 1; })
}

// out[row, 0..w) = (bias) + sum_j Y[perm[j], :] over the units of a row; Y rows are 4 LPR floats (16-byte pieces, one per lane), LPR lanes
// per unit, two row reads in flight per lane (the layout of segment_gather_sum_units_d16_kernel for any width <= 16).
template <int LPR>
__global__ __launch_bounds__(256) void gather_rows_sum4_kernel(const float *__restrict__ Y, const int *__restrict__ perm,
                                                               const int4 *__restrict__ units, const float *__restrict__ bias,
                                                               float *__restrict__ out, long long n_units, int w, int ow, int relu_out, int vec_out) {
  const int q = threadIdx.x % LPR;
  for (long long u = ((long long)blockIdx.x * 256 + threadIdx.x) / LPR; u < n_units; u += ((long long)gridDim.x * 256) / LPR) {
    const int4 unit = units[u];
    const bool shared = unit.w & RGCN_U_SHARED;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    if (bias && (!shared || (unit.w & RGCN_U_FIRST))) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (4 * q + c < w) a[c] = bias[4 * q + c];
    }
    // Four row reads in flight, and the NEXT four indices loaded before this trip's rows are used: a trip of the loop is one round trip
    // (the rows), not two (indices, then rows).  Entries past the unit's end re-read its last entry (unconditional loads) and add nothing.
    const int e0 = unit.y, e1 = unit.z;
    if (e1 > e0) {
      int p[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) p[j] = perm[min(e0 + j, e1 - 1)];
      for (int e = e0; e < e1; e += 4) {
        f32x4 y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = *reinterpret_cast<const f32x4 *>(Y + (size_t)p[j] * (4 * LPR) + 4 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = perm[min(e + 4 + j, e1 - 1)];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (e + j < e1) { if (j & 1) b += y[j]; else a += y[j]; }
      }
    }
    a += b;
    float *o = out + (size_t)unit.x * ow + 4 * q;           // rows of ow >= w floats: columns w .. ow are written as zeros (a zero-padded
                                                            // [N, 16] row is what the width-16 kernels of the next layer read in place)
    if (!shared && vec_out) {                               // whole 16-byte pieces: one store per lane (four scalar ones: 0.34 -> 0.30 ms at AM size)
      if (4 * q < ow) {
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = 4 * q + c < w ? (relu_out ? fmaxf(a[c], 0.f) : a[c]) : 0.f;
        *reinterpret_cast<f32x4 *>(o) = v;
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (4 * q + c < w) {
          if (shared) atomicAdd(o + c, a[c]); else o[c] = relu_out ? fmaxf(a[c], 0.f) : a[c];
        } else if (4 * q + c < ow && !shared) o[c] = 0.f;
      }
    }
  }
}

// ================================================================== backward
// Both kernels: lane = basis b for the sums.  The upstream gradient rows of the wave's messages are gathered one tile ahead into registers
// (packed: register q of lane (m, c) = G[dst of message 4 q + m][c]) and, once arrived, laid down in a wave-private LDS strip
// [PERB messages][GS = 4 ceil(d / 4) floats]: the message loops then read a row with broadcast 16-byte reads (dcomps) or as the MFMA's B
// operand (dbases) -- no readlane per feature, no register indexed by the message.
constexpr int PERB = 16, GQ = PERB / 4;     // messages of one wave per chunk in the backward kernels

struct Idx { int es, er; float ev; };
// (lanes past the wave's n messages re-read entry `last` = M - 1: every load of the pipeline is unconditional -- see stage_load)
__device__ __forceinline__ Idx idx_load(const int *__restrict__ e_dst, const int *__restrict__ e_rel, const float *__restrict__ e_val,
                                        int a, int last, int lane) {
  Idx x;
  const int e = min(a + lane, last);
  x.es = e_dst[e];
  x.er = e_rel[e];
  x.ev = e_val[e];
  return x;
}
// lanes (m, c >= d) re-read column d - 1 and messages past n the wave's last one: finite duplicates that no sum ever uses
__device__ __forceinline__ void gather_rows(float (&gp)[GQ], const float *__restrict__ G, int es, int n, int d, int lane, int gstride) {
  const int m = lane >> 4, c = min(lane & 15, d - 1);
#pragma unroll
  for (int q = 0; q < GQ; ++q)
    if (4 * q < n) {                                       // wave-uniform
      const int s = __shfl(es, min(4 * q + m, n - 1), 64);
      gp[q] = G[(size_t)s * gstride + c];
    }
}
template <int GS, int NQ>
__device__ __forceinline__ void strip_store(float *gs, const float (&gp)[NQ], int n, int lane) {
  const int m = lane >> 4, c = lane & 15;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
    if (4 * q < n && c < GS) gs[(4 * q + m) * GS + c] = gp[q];          // (rows n .. 4 ceil(n / 4) hold duplicates of row n - 1)
}

// ---- dcomps[r,b] += val_e <bases[b,o_e,:], G[s_e,:]>: tiles staged like the forward's; dcomps summed per workgroup in an LDS table of doubles
template <int DPB, int KLD, bool VEC>
__global__ __launch_bounds__(TW) void fbt_dcomps_kernel(
    const float *__restrict__ bases, const float *__restrict__ G, float *__restrict__ dC, const int *__restrict__ rowptr,
    const int *__restrict__ e_dst, const int *__restrict__ e_rel, const float *__restrict__ e_val, int n_tiles, int N, int R, int B,
    int d, int ts, int last, int gstride, int abl) {
  constexpr int GS = DPB;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  double *dcl = reinterpret_cast<double *>(lds);            // [R][B]
  const int strips0 = (2 * R * B + 3) & ~3;                 // (16-byte aligned: the rows are read 16 bytes at a time)
  float *gs = lds + strips0 + wave * (PERB * GS);           // the wave's strip of gradient rows
  float *tb = lds + strips0 + TWV * PERB * GS;              // 2 x [B][ts], ts odd: lane b reads tb[b * ts + ...] without bank conflicts
  for (int j = tid; j < R * B; j += TW) dcl[j] = 0.0;
  Geo<KLD> g;
  geo_init(g, tid, B, d, N, ts);
  const int Gd = gridDim.x;
  int t = blockIdx.x;
  auto rp_of = [&](int tt) {
    tt = min(tt, n_tiles - 1);
    return rowptr[min((long long)tt * TN + min(lane, TN), (long long)N)];
  };
  auto base_of = [&](int tt) { return (long long)min(min(tt, n_tiles - 1) * TN, N - TN) * d; };
  const int bl = min(lane, B - 1);
  const bool has_b = lane < B;

  f32x4 st[KLD];
  int rp = rp_of(t), rp1 = rp_of(t + Gd), rp2 = rp_of(t + 2 * Gd);
  stage_load<KLD, VEC>(st, g, bases, base_of(t));
  Share s = share_of<PERB>(rp, wave, true);
  Idx x = idx_load(e_dst, e_rel, e_val, s.a, last, lane);
  Share s1 = share_of<PERB>(rp1, wave, t + Gd < n_tiles);
  Idx x1 = idx_load(e_dst, e_rel, e_val, s1.a, last, lane);
  float gp1[GQ] = {};
  gather_rows(gp1, G, x.es, s.n, d, lane, gstride);
  strip_store<GS>(gs, gp1, s.n, lane);
  stage_store<KLD, false>(tb, st, g);
  stage_load<KLD, VEC>(st, g, bases, base_of(t + Gd));
  lds_barrier();
  for (int k = 0;; ++k) {
    const bool has1 = t + Gd < n_tiles, has2 = t + 2 * Gd < n_tiles;
    const float *cb = tb + (k & 1) * (B * ts);
    if (has1) stage_store<KLD, false>(tb + ((k + 1) & 1) * (B * ts), st, g);
    __builtin_amdgcn_sched_barrier(0);
    // issue: the tile two ahead, row pointers three, indices two, gradient rows one -- all arrive before the rotation at the bottom
    if (!FBT_ABL(2)) stage_load<KLD, VEC>(st, g, bases, base_of(t + 2 * Gd));
    const int rp3 = rp_of(t + 3 * Gd);
    const Share s2 = share_of<PERB>(rp2, wave, has2);
    const Idx x2 = idx_load(e_dst, e_rel, e_val, s2.a, last, lane);
    if (!FBT_ABL(4)) gather_rows(gp1, G, x1.es, s1.n, d, lane, gstride);
    __builtin_amdgcn_sched_barrier(0);
    // ---- tile t
    const int shift = t * TN - min(t * TN, N - TN);
    int c_a = s.a, c_n = s.n;
    Idx c_x = x;
    for (int c0 = s.mb;;) {
      for (int p = c_a, end = FBT_ABL(1) ? c_a : c_a + c_n; p < end;) {
        const int nt = node_of(rp, lane, p);
        const int run_end = min(end, rlane(rp, nt + 1));
        const int nl = nt + shift;
        float blk[DPB];
#pragma unroll
        for (int i = 0; i < DPB; ++i) {
          const float v = cb[bl * ts + nl * d + min(i, d - 1)];
          blk[i] = i < d ? v : 0.f;
        }
        auto one = [&](int j) {
          const f32x4 *g4 = reinterpret_cast<const f32x4 *>(gs + j * GS);      // the same address in every lane: a broadcast read
          f32x2 e = {0.f, 0.f};
#pragma unroll
          for (int i4 = 0; i4 < DPB / 4; ++i4) {
            const f32x4 gv = g4[i4];
            e += f32x2{blk[4 * i4], blk[4 * i4 + 1]} * f32x2{gv[0], gv[1]};
            e += f32x2{blk[4 * i4 + 2], blk[4 * i4 + 3]} * f32x2{gv[2], gv[3]};
          }
          const int r = rlane(c_x.er, j);
          const float v = bcast(c_x.ev, j);
          if (has_b && !FBT_ABL(8)) __hip_atomic_fetch_add(dcl + r * B + lane, (double)(v * (e[0] + e[1])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        for (; p + 4 <= run_end; p += 4) {                  // four messages: all row reads in flight before the first FMA
          constexpr int QB = (DPB <= 12 && KLD == 2) ? 4 : 2;
#pragma unroll
          for (int h = 0; h < 4; h += QB) {
            const int j0 = p - c_a + h;
            f32x4 gq[QB][DPB / 4];
#pragma unroll
            for (int m = 0; m < QB; ++m)
#pragma unroll
              for (int i4 = 0; i4 < DPB / 4; ++i4) gq[m][i4] = reinterpret_cast<const f32x4 *>(gs + (j0 + m) * GS)[i4];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < QB; ++m) {
              f32x2 e = {0.f, 0.f};
#pragma unroll
              for (int i4 = 0; i4 < DPB / 4; ++i4) {
                e += f32x2{blk[4 * i4], blk[4 * i4 + 1]} * f32x2{gq[m][i4][0], gq[m][i4][1]};
                e += f32x2{blk[4 * i4 + 2], blk[4 * i4 + 3]} * f32x2{gq[m][i4][2], gq[m][i4][3]};
              }
              const int r = rlane(c_x.er, j0 + m);
              const float v = bcast(c_x.ev, j0 + m);
              if (has_b && !FBT_ABL(8)) __hip_atomic_fetch_add(dcl + r * B + lane, (double)(v * (e[0] + e[1])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
        }
        for (; p < run_end; ++p) one(p - c_a);
      }
      c0 += TWV * s.per;
      if (c0 >= s.me) break;
      c_a = c0 + wave * s.per;                              // tiles with more than 16 x PERB messages: further chunks, loaded on demand
      c_n = max(0, min(s.per, min(c0 + TWV * s.per, s.me) - c_a));
      c_x = idx_load(e_dst, e_rel, e_val, c_a, last, lane);
      float gq[GQ] = {};
      gather_rows(gq, G, c_x.es, c_n, d, lane, gstride);
      FBT_ARRIVED(c_x.er); FBT_ARRIVED(c_x.ev);
      strip_store<GS>(gs, gq, c_n, lane);
    }
    lds_barrier();
    if (!has1) break;
    t += Gd;
    rp = rp1; rp1 = rp2; rp2 = rp3; s = s1; s1 = s2; x = x1; x1 = x2;
    strip_store<GS>(gs, gp1, s.n, lane);                    // (wave-private: written after the wave's own reads of the tile before)
  }
  for (int j = tid; j < R * B; j += TW) {
    const float v = (float)dcl[j];
    if (v != 0.f) atomicAdd(dC + j, v);
  }
}

// ---- dbases[b,o,:] = sum_e val_e comps[r_e,b] G[s_e,:]: per run of a node's messages D[b][i] += sum_m A[b][m] B[m][i] on the matrix cores
// (v_mfma_f32_16x16x4_f32, four messages per step: A = val_m comps[r_m, 16 t + row] out of the LDS coefficient table, B = the strip's rows);
// the tile's gradient is summed in LDS doubles (two tiles: one being written out while the next is summed) and written once, in place
template <int DPB, int KLD, bool VEC>
__global__ __launch_bounds__(TW) void fbt_dbases_kernel(
    const float *__restrict__ comps, const float *__restrict__ G, float *__restrict__ dbases, const int *__restrict__ rowptr,
    const int *__restrict__ e_dst, const int *__restrict__ e_rel, const float *__restrict__ e_val, int n_tiles, int N, int R, int B,
    int d, int ts, int last, int gstride, int abl) {
  constexpr int GS = DPB, NT = 4;                           // NT 16-row tiles of bases (B <= 64)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  double *dt = reinterpret_cast<double *>(lds);             // 2 x [B][ts] doubles, ts odd
  float *gs = lds + 4 * B * ts + wave * (PERB * GS);
  float *ctab = lds + 4 * B * ts + TWV * PERB * GS;         // [R][B]
  for (int j = tid; j < R * B; j += TW) ctab[j] = comps[j];
  for (int j = tid; j < 2 * B * ts; j += TW) dt[j] = 0.0;
  Geo<KLD> g;
  geo_init(g, tid, B, d, N, ts);
  const int Gd = gridDim.x;
  int t = blockIdx.x;
  auto rp_of = [&](int tt) {
    tt = min(tt, n_tiles - 1);
    return rowptr[min((long long)tt * TN + min(lane, TN), (long long)N)];
  };
  const int nbt = (B + 15) >> 4;                            // 16-row tiles in use
  const int lm = lane >> 4, lc = lane & 15;

  int rp = rp_of(t), rp1 = rp_of(t + Gd), rp2 = rp_of(t + 2 * Gd);
  Share s = share_of<PERB>(rp, wave, true);
  Idx x = idx_load(e_dst, e_rel, e_val, s.a, last, lane);
  Share s1 = share_of<PERB>(rp1, wave, t + Gd < n_tiles);
  Idx x1 = idx_load(e_dst, e_rel, e_val, s1.a, last, lane);
  float gp1[GQ] = {};
  gather_rows(gp1, G, x.es, s.n, d, lane, gstride);
  strip_store<GS>(gs, gp1, s.n, lane);
  int t_out = -1;                                           // the tile whose gradient waits in dt[(k - 1) & 1]
  auto write_out = [&](double *src) {
    const int n0 = t_out * TN, n0s = min(n0, N - TN);
    const long long base = (long long)n0s * d;
    const int lo = (n0 - n0s) * d;                          // the last tile: positions below `lo` belong to the tile before
#pragma unroll
    for (int k2 = 0; k2 < KLD; ++k2)
      if (g.act[k2]) {
        double *p = src + g.loff[k2];
        const f32x4 v = {(float)p[0], (float)p[1], (float)p[2], (float)p[3]};
        p[0] = 0.0; p[1] = 0.0; p[2] = 0.0; p[3] = 0.0;
        float *o = dbases + g.goff[k2] + base;
        const int pos = g.pos[k2];
        if (VEC && pos >= lo) *reinterpret_cast<f32x4 *>(o) = v;
        else {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (pos + c >= lo) o[c] = v[c];
        }
      }
  };
  lds_barrier();
  for (int k = 0;; ++k) {
    const bool has1 = t + Gd < n_tiles, has2 = t + 2 * Gd < n_tiles;
    double *dtc = dt + (k & 1) * (B * ts);
    if (t_out >= 0) write_out(dt + ((k + 1) & 1) * (B * ts));
    __builtin_amdgcn_sched_barrier(0);
    const int rp3 = rp_of(t + 3 * Gd);
    const Share s2 = share_of<PERB>(rp2, wave, has2);
    const Idx x2 = idx_load(e_dst, e_rel, e_val, s2.a, last, lane);
    if (!FBT_ABL(4)) gather_rows(gp1, G, x1.es, s1.n, d, lane, gstride);
    __builtin_amdgcn_sched_barrier(0);
    // ---- tile t (never the overlapped part of the last tile: local node = tile-relative + shift)
    const int shift = t * TN - min(t * TN, N - TN);
    int c_a = s.a, c_n = s.n;
    Idx c_x = x;
    for (int c0 = s.mb;;) {
      for (int p = c_a, end = FBT_ABL(1) ? c_a : c_a + c_n; p < end;) {
        const int nt = node_of(rp, lane, p);
        const int run_end = min(end, rlane(rp, nt + 1));
        const int nl = nt + shift;
        f32x4 acc[NT];
#pragma unroll
        for (int tb = 0; tb < NT; ++tb) acc[tb] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int p4 = p; p4 < run_end; p4 += 4) {
          const int j0 = p4 - c_a, jm = min(j0 + lm, c_n - 1);                    // this lane group's message (past the run: weight 0)
          const int r = __shfl(c_x.er, jm, 64);
          const float vs = __shfl(c_x.ev, jm, 64);                                // (unconditional: a shuffle inside `cond ? ... : 0` runs under the
          const float v = (p4 + lm < run_end) ? vs : 0.f;                        //  condition's EXEC mask, and a masked-off SOURCE lane delivers 0)
          const float bv = gs[jm * GS + min(lc, GS - 1)];                        // B operand: lane (m, c) = G row of message m, column c
#pragma unroll
          for (int tb = 0; tb < NT; ++tb)
            if (tb < nbt) {                                                      // uniform
              const float av = ctab[r * B + min(16 * tb + lc, B - 1)] * v;       // A operand: lane (m, row) = val_m comps[r_m][16 tb + row]
              acc[tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[tb], 0, 0, 0);
            }
        }
        p = run_end;
        // D: lane (q, column i) holds rows 4 q .. 4 q + 3 of every 16-row tile: the lanes of a quarter wave add to consecutive doubles
        if (lc < d && !FBT_ABL(8)) {
#pragma unroll
          for (int tb = 0; tb < NT; ++tb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int b = 16 * tb + 4 * lm + e;
              if (tb < nbt && b < B)
                __hip_atomic_fetch_add(dtc + b * ts + nl * d + lc, (double)acc[tb][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
      }
      c0 += TWV * s.per;
      if (c0 >= s.me) break;
      c_a = c0 + wave * s.per;
      c_n = max(0, min(s.per, min(c0 + TWV * s.per, s.me) - c_a));
      c_x = idx_load(e_dst, e_rel, e_val, c_a, last, lane);
      float gq[GQ] = {};
      gather_rows(gq, G, c_x.es, c_n, d, lane, gstride);
      FBT_ARRIVED(c_x.er); FBT_ARRIVED(c_x.ev);
      strip_store<GS>(gs, gq, c_n, lane);
    }
    t_out = t;
    lds_barrier();
    if (!has1) {
      write_out(dtc);
      break;
    }
    t += Gd;
    rp = rp1; rp1 = rp2; rp2 = rp3; s = s1; s1 = s2; x = x1; x1 = x2;
    strip_store<GS>(gs, gp1, s.n, lane);
  }
}

// ================================================================== one wave per node, on the matrix cores (round 4, second form)
// The kernels above deal a tile's MESSAGES evenly over the waves: balanced on any degree distribution, but the per-message instruction
// streams (VALU dot products, a transpose-reduce per four messages, one LDS atomic flush per node run) and the partial quads at node
// boundaries bound them, and the waves reach the tile barrier unevenly.  When no node is a hub (rgcn_fbasis_tile_*: mode 1, chosen by
// the caller from the largest source degree) wave w of the workgroup takes NODE w of the tile and runs its messages through
// v_mfma_f32_16x16x4_f32, sixteen at a time -- the cost of a node is nearly the same for 1 or 16 messages, so the waves stay in step:
//   forward  D[m][i] = sum_b (val_m comps[r_m, b]) bases[b, o, i]      A = scaled coefficient rows out of the LDS table, B = the node's block
//   dcomps   D[m][b] = sum_i G[s_m, i] bases[b, o, i]                   A = the gathered rows (LDS strip), B = the block transposed;
//                                                                       val_m D[m][b] is added to the R x B doubles (16 consecutive b per quarter wave)
//   dbases   D[b][i] = sum_m (val_m comps[r_m, b]) G[s_m, i]            four messages per step; the wave OWNS its node's gradient: plain
//                                                                       stores into an fp32 tile in LDS, no atomics
// Lane = (k, c) = (lane >> 4, lane & 15) throughout: A operand lane 16 k + row, B operand lane 16 k + column, D lane 16 q + column holds
// rows 4 q .. 4 q + 3.  The pipeline (tile two ahead, row pointers three, indices two, gradient rows one) is the one above.
struct NodeRange { int a, n; };
__device__ __forceinline__ NodeRange node_range(int rp, int wave, bool valid) {
  NodeRange r;
  r.a = rlane(rp, wave);
  r.n = valid ? rlane(rp, wave + 1) - r.a : 0;
  return r;
}
// rows of up to 16 messages starting at entry `off` of the wave's 64 prefetched indices -> packed registers (see gather_rows)
template <int NQ>
__device__ __forceinline__ void gather_rows_at(float (&gp)[NQ], const float *__restrict__ G, int es, int off, int n, int d, int lane, int gstride) {
  const int m = lane >> 4, c = min(lane & 15, d - 1);
#pragma unroll
  for (int q = 0; q < NQ; ++q)
    if (4 * q < n) {
      const int s = __shfl(es, off + min(4 * q + m, n - 1), 64);
      gp[q] = G[(size_t)s * gstride + c];
    }
}

// the first rows of the wave's run, ALWAYS four loads (a run shorter than 16 re-reads its last row, an empty one entry 0 of the indices):
// a fixed number of loads per iteration lets the compiler count them (s_waitcnt vmcnt(n)) instead of draining everything
template <int NQ>
__device__ __forceinline__ void gather_rows_all(float (&gp)[NQ], const float *__restrict__ G, int es, int n, int d, int lane, int gstride) {
  const int m = lane >> 4, c = min(lane & 15, d - 1);
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int s = __shfl(es, min(4 * q + m, max(n - 1, 0)), 64);
    gp[q] = G[(size_t)s * gstride + c];
  }
}

template <int NKSM, int KLD, bool VEC>
__global__ __launch_bounds__(TW) void fbn_fwd_kernel(
    const float *__restrict__ bases, const float *__restrict__ comps, float *__restrict__ Y, const int *__restrict__ rowptr,
    const int *__restrict__ e_rel, const float *__restrict__ e_val, int n_tiles, int N, int R, int B, int d, int ts, int last, int ys,
    int abl) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = lane >> 4, c = lane & 15;
  // row stride of the coefficient table: ODD.  The A operand is ctab[r_m * BP + 4 ks + k] for the 16 relations r_m of a chunk: with
  // BP = 40 (B = 40) the rows start at banks 8 r mod 32 -- four bank groups for sixteen rows, SQ_LDS_BANK_CONFLICT 0.62 of the LDS cycles
  // (profiles/r05_amshipped_pmc.json); an odd stride spreads the rows over all 32 banks
  const int nks = (B + 3) >> 2, BP = 4 * nks + 1;
  float *ctab = lds;                                        // [R][BP], zero padded
  float *tb = lds + ((R * BP + 3) & ~3);
  float *yb = tb + 2 * B * ts + wave * (16 * 16);
  for (int j = tid; j < R * BP; j += TW) {
    const int r = j / BP, b = j % BP;
    ctab[j] = b < B ? comps[(size_t)r * B + b] : 0.f;
  }
  Geo<KLD> g;
  geo_init(g, tid, B, d, N, ts);
  const int G = gridDim.x;
  int t = blockIdx.x;
  auto rp_of = [&](int tt) {
    tt = min(tt, n_tiles - 1);
    return rowptr[min((long long)tt * TN + min(lane, TN), (long long)N)];
  };
  auto base_of = [&](int tt) { return (long long)min(min(tt, n_tiles - 1) * TN, N - TN) * d; };

  f32x4 st[KLD];
  int rp = rp_of(t), rp1 = rp_of(t + G);
  stage_load<KLD, VEC>(st, g, bases, base_of(t));
  NodeRange s = node_range(rp, wave, true);
  int er = e_rel[min(s.a + lane, last)];
  float ev = e_val[min(s.a + lane, last)];
  stage_store<KLD, true>(tb, st, g);
  stage_load<KLD, VEC>(st, g, bases, base_of(t + G));
  int pa = 0, pn = 0;                                      // rows pa .. pa + pn of Y wait in the wave's strip
  // -> the stored registers: the caller keeps them alive (FBT_KEEP) across the message loops.  hipcc guards a store's DATA registers with
  // s_waitcnt vmcnt(0) before their next write; reused right away -- by the address arithmetic of the loads that follow -- every
  // iteration waited for its Y store to COMPLETE before it issued a single load
  auto flush = [&]() {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (lane < pn * (ys >> 2)) {
      v = reinterpret_cast<const f32x4 *>(yb)[lane];
      reinterpret_cast<f32x4 *>(Y + (size_t)pa * ys)[lane] = v;
    }
    pn = 0;
    return v;
  };
  lds_barrier();
  for (int it = 0;; ++it) {
    const bool has1 = t + G < n_tiles;
    const float *cb = tb + (it & 1) * (B * ts);
    // The staged pieces are waited for HERE, by every thread, before anything is stored (round 5, found in the ISA): a thread without a second
    // piece skips that piece's LDS write and with it the wait for its load -- at the join the compiler then has to assume the load still in
    // flight and, before the next loads may overwrite its registers, waits with vmcnt(0) AFTER the Y flush was issued: every iteration paid the
    // round trip of its own store before it requested the next tile
#pragma unroll
    for (int k2 = 0; k2 < KLD; ++k2) asm volatile("" : "+v"(st[k2]));
    if (has1) stage_store<KLD, true>(tb + ((it + 1) & 1) * (B * ts), st, g);
    f32x4 kept = flush();
    __builtin_amdgcn_sched_barrier(0);
    // the small loads FIRST, the tile two ahead LAST: vmcnt counts in order, so the wait for the indices at the bottom of the iteration then
    // leaves the tile's loads in flight (issued the other way round it drained them: the tile was never more than one iteration ahead)
    const int rp2 = rp_of(t + 2 * G);
    const NodeRange s1 = node_range(rp1, wave, has1);
    const int er1 = e_rel[min(s1.a + lane, last)];
    const float ev1 = e_val[min(s1.a + lane, last)];
    __builtin_amdgcn_sched_barrier(0);
    if (!FBT_ABL(2)) stage_load<KLD, VEC>(st, g, bases, base_of(t + 2 * G));
    __builtin_amdgcn_sched_barrier(0);
    // ---- node `wave` of tile t
    const int nl = wave + t * TN - min(t * TN, N - TN);
    if (s.n > 0 && !FBT_ABL(1)) {
      float bb[NKSM];
#pragma unroll
      for (int ks = 0; ks < NKSM; ++ks)
        if (ks < nks) bb[ks] = cb[min(4 * ks + k, B - 1) * ts + nl * d + min(c, d - 1)];      // (rows past B meet zero coefficients)
      int c_er = er;
      float c_ev = ev;
      for (int g0 = 0; g0 < s.n; g0 += 16) {
        if (g0) {
          flush();
          if ((g0 & 63) == 0) {                             // more than 64 messages: the next 64 indices, on demand
            c_er = e_rel[min(s.a + g0 + lane, last)];
            c_ev = e_val[min(s.a + g0 + lane, last)];
            FBT_ARRIVED(c_er); FBT_ARRIVED(c_ev);
          }
        }
        const int src = (g0 & 63) + c;
        const int r = __shfl(c_er, src, 64);
        const float vs = __shfl(c_ev, src, 64);            // (never inside the conditional: a masked-off source lane delivers 0)
        const float v = (g0 + c < s.n) ? vs : 0.f;
        const float *crow = ctab + r * BP + k;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKSM; ++ks)
          if (ks < nks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(crow[4 * ks] * v, bb[ks], acc, 0, 0, 0);
        if (c < ys) {
#pragma unroll
          for (int e = 0; e < 4; ++e) yb[(4 * k + e) * ys + c] = c < d ? acc[e] : 0.f;
        }
        pa = s.a + g0;
        pn = min(16, s.n - g0);
      }
    }
    FBT_KEEP(kept);
    lds_barrier();
    if (!has1) break;
    t += G;
    rp = rp1; rp1 = rp2; s = s1; er = er1; ev = ev1;
    FBT_ARRIVED(er); FBT_ARRIVED(ev); FBT_ARRIVED(rp1);
  }
  flush();
}

template <int NKD, int KLD, bool VEC>
__global__ __launch_bounds__(TW) void fbn_dcomps_kernel(
    const float *__restrict__ bases, const float *__restrict__ G, float *__restrict__ dC, const int *__restrict__ rowptr,
    const int *__restrict__ e_dst, const int *__restrict__ e_rel, const float *__restrict__ e_val, int n_tiles, int N, int R, int B,
    int d, int ts, int last, int gstride, int abl) {
  constexpr int GS = 4 * NKD, NBTM = 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = lane >> 4, c = lane & 15;
  const int nbt = (B + 15) >> 4;
  double *dcl = reinterpret_cast<double *>(lds);            // [R][B]
  const int strips0 = (2 * R * B + 3) & ~3;
  float *gs = lds + strips0 + wave * (PERB * GS);
  float *tb = lds + strips0 + TWV * PERB * GS;              // 2 x [B][ts]
  for (int j = tid; j < R * B; j += TW) dcl[j] = 0.0;
  Geo<KLD> g;
  geo_init(g, tid, B, d, N, ts);
  const int Gd = gridDim.x;
  int t = blockIdx.x;
  auto rp_of = [&](int tt) {
    tt = min(tt, n_tiles - 1);
    return rowptr[min((long long)tt * TN + min(lane, TN), (long long)N)];
  };
  auto base_of = [&](int tt) { return (long long)min(min(tt, n_tiles - 1) * TN, N - TN) * d; };

  f32x4 st[KLD];
  int rp = rp_of(t), rp1 = rp_of(t + Gd), rp2 = rp_of(t + 2 * Gd);
  stage_load<KLD, VEC>(st, g, bases, base_of(t));
  NodeRange s = node_range(rp, wave, true);
  Idx x = idx_load(e_dst, e_rel, e_val, s.a, last, lane);
  NodeRange s1 = node_range(rp1, wave, t + Gd < n_tiles);
  Idx x1 = idx_load(e_dst, e_rel, e_val, s1.a, last, lane);
  float gp1[GQ] = {};
  gather_rows_at(gp1, G, x.es, 0, min(16, s.n), d, lane, gstride);
  strip_store<GS>(gs, gp1, min(16, s.n), lane);
  stage_store<KLD, false>(tb, st, g);
  stage_load<KLD, VEC>(st, g, bases, base_of(t + Gd));
  lds_barrier();
  for (int it = 0;; ++it) {
    const bool has1 = t + Gd < n_tiles, has2 = t + 2 * Gd < n_tiles;
    const float *cb = tb + (it & 1) * (B * ts);
#pragma unroll
    for (int k2 = 0; k2 < KLD; ++k2) asm volatile("" : "+v"(st[k2]));     // (every thread waits for its staged pieces here: see fbn_fwd_kernel)
    if (has1) stage_store<KLD, false>(tb + ((it + 1) & 1) * (B * ts), st, g);
    __builtin_amdgcn_sched_barrier(0);
    const int rp3 = rp_of(t + 3 * Gd);
    const NodeRange s2 = node_range(rp2, wave, has2);
    const Idx x2 = idx_load(e_dst, e_rel, e_val, s2.a, last, lane);
    if (!FBT_ABL(4)) gather_rows_at(gp1, G, x1.es, 0, min(16, s1.n), d, lane, gstride);
    __builtin_amdgcn_sched_barrier(0);
    if (!FBT_ABL(2)) stage_load<KLD, VEC>(st, g, bases, base_of(t + 2 * Gd));      // (the tile LAST: the waits for the small loads leave it in flight)
    __builtin_amdgcn_sched_barrier(0);
    // ---- node `wave` of tile t
    const int nl = wave + t * TN - min(t * TN, N - TN);
    if (s.n > 0 && !FBT_ABL(1)) {
      float bt[NBTM][NKD];                                  // the block transposed: B operand lane (k, c) = bases[16 tb + c][4 ks + k]
#pragma unroll
      for (int tb2 = 0; tb2 < NBTM; ++tb2)
#pragma unroll
        for (int ks = 0; ks < NKD; ++ks) {
          bt[tb2][ks] = 0.f;
          if (tb2 < nbt) {
            const float v = cb[min(16 * tb2 + c, B - 1) * ts + nl * d + min(4 * ks + k, d - 1)];
            bt[tb2][ks] = (16 * tb2 + c < B && 4 * ks + k < d) ? v : 0.f;
          }
        }
      Idx c_x = x;
      for (int g0 = 0; g0 < s.n; g0 += 16) {
        const int n16 = min(16, s.n - g0);
        if (g0) {                                           // more than 16 messages: indices per 64, rows per 16, on demand
          if ((g0 & 63) == 0) {
            c_x = idx_load(e_dst, e_rel, e_val, s.a + g0, last, lane);
            FBT_ARRIVED(c_x.es); FBT_ARRIVED(c_x.er); FBT_ARRIVED(c_x.ev);
          }
          float gq[GQ] = {};
          gather_rows_at(gq, G, c_x.es, g0 & 63, n16, d, lane, gstride);
          strip_store<GS>(gs, gq, n16, lane);
        }
        float av[NKD];
        // A operand lane (k, row c) = G row of message pm(c) = 4 (c & 3) + (c >> 2), feature 4 ks + k: the rows in 4 x 4-transposed message order, so that
        // D's row 4 k + e is message 4 e + k and the e-th table update below is skipped whole for a node of at most 4 e messages (fbn_bwd_kernel)
        const int pm = min(4 * (c & 3) + (c >> 2), n16 - 1);
#pragma unroll
        for (int ks = 0; ks < NKD; ++ks) av[ks] = gs[pm * GS + 4 * ks + k];
        int rq[4];
        float vq[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {                       // D's rows of this lane: row 4 k + e = message 4 e + k
          const int src = (g0 & 63) + min(4 * e + k, n16 - 1);
          rq[e] = __shfl(c_x.er, src, 64);
          const float vs = __shfl(c_x.ev, src, 64);
          vq[e] = (4 * e + k < n16) ? vs : 0.f;
        }
#pragma unroll
        for (int tb2 = 0; tb2 < NBTM; ++tb2)
          if (tb2 < nbt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKD; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bt[tb2][ks], acc, 0, 0, 0);
            if (16 * tb2 + c < B && !FBT_ABL(8)) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (4 * e + k < n16)
                  __hip_atomic_fetch_add(dcl + rq[e] * B + 16 * tb2 + c, (double)(vq[e] * acc[e]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
      }
    }
    lds_barrier();
    if (!has1) break;
    t += Gd;
    rp = rp1; rp1 = rp2; rp2 = rp3; s = s1; s1 = s2; x = x1; x1 = x2;
    strip_store<GS>(gs, gp1, min(16, s.n), lane);
  }
  for (int j = tid; j < R * B; j += TW) {
    const float v = (float)dcl[j];
    if (v != 0.f) atomicAdd(dC + j, v);
  }
}

// NPW nodes per wave (tiles of 16 NPW nodes): with 32-node tiles the per-tile costs (write-out, issue, barrier, the wait for the loads) are paid
// half as often and a tile's message loops are as long as the memory latency they have to cover
// NW waves per workgroup: 16 (one workgroup per CU, the tile gradient double-buffered) or 8 (512 threads, ONE tile buffer, 75 KB of LDS: two
// workgroups per CU -- one's stores and loads fly under the other's message loops; a single workgroup's waves are all in the same phase)
template <int NKD, int KLD, bool VEC, int NPW, int NW>
__global__ __launch_bounds__(64 * NW) void fbn_dbases_kernel(
    const float *__restrict__ comps, const float *__restrict__ G, float *__restrict__ dbases, const int *__restrict__ rowptr,
    const int *__restrict__ e_dst, const int *__restrict__ e_rel, const float *__restrict__ e_val, int n_tiles, int N, int R, int B,
    int d, int ts, int last, int gstride, int abl) {
  constexpr int GS = 4 * NKD, NBTM = 4, TNV = NW * NPW, NTH = 64 * NW;
  constexpr bool TWO = NW == 16;                           // two tile buffers
  constexpr int SR = 16 * NPW, NQ = SR / 4;                // rows of the wave's strip = gradient rows gathered ahead: the wave's nodes send ~8 NPW messages
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = lane >> 4, c = lane & 15;
  const int nbt = (B + 15) >> 4;
  float *dt = lds;                                          // (2 x) [B][ts] floats: every entry is stored by the wave that owns its node
  float *gs = lds + (TWO ? 2 : 1) * B * ts + wave * (SR * GS);
  float *ctab = lds + (TWO ? 2 : 1) * B * ts + NW * SR * GS;        // [R][B]
  for (int j = tid; j < R * B; j += NTH) ctab[j] = comps[j];
  Geo<KLD> g;
  geo_init(g, tid, B, d, N, ts, TNV, NTH);
  const int Gd = gridDim.x;
  int t = blockIdx.x;
  // tile tt = nodes first_of(tt) .. + TNV; the LAST tile is nodes N - TNV .. N whatever N: where it overlaps the tile before, both workgroups
  // compute and store the same gradients (identical values) -- no partial tile, no guarded stores, every tile's write-out is the same
  // KLD 16-byte stores per thread
  auto first_of = [&](int tt) { return min(min(tt, n_tiles - 1) * TNV, N - TNV); };
  auto rp_of = [&](int tt) { return rowptr[first_of(tt) + min(lane, TNV)]; };      // lane l <= TNV: rowptr[tile's first node + l]
  auto range_of = [&](int rpv, bool valid) {                // the wave's NPW nodes send one contiguous run of messages
    NodeRange r;
    r.a = rlane(rpv, wave * NPW);
    r.n = valid ? rlane(rpv, wave * NPW + NPW) - r.a : 0;
    return r;
  };

  int rp = rp_of(t), rp1 = rp_of(t + Gd), rp2 = rp_of(t + 2 * Gd);
  NodeRange s = range_of(rp, true);
  Idx x = idx_load(e_dst, e_rel, e_val, s.a, last, lane);
  NodeRange s1 = range_of(rp1, t + Gd < n_tiles);
  Idx x1 = idx_load(e_dst, e_rel, e_val, s1.a, last, lane);
  float gp1[NQ] = {};
  gather_rows_at(gp1, G, x.es, 0, min(SR, s.n), d, lane, gstride);
  strip_store<GS>(gs, gp1, min(SR, s.n), lane);
  // the tile whose gradient waits in the other buffer.  Before the first tile: the workgroup's own first tile -- whatever the LDS holds goes out
  // and is overwritten one iteration later by the same threads: every iteration then issues the same number of stores, which lets the
  // compiler wait for the iteration's LOADS with an exact count and leave the stores in flight
  int t_out = t;
  f32x4 kept[KLD];                                          // the stored pieces, kept alive over the message loops (see fbn_fwd_kernel's flush)
  auto write_out = [&](const float *src, int part) {        // part < NPW: that share of the thread's pieces; part < 0: all of them
    const long long base = (long long)first_of(t_out) * d;
#pragma unroll
    for (int k2 = 0; k2 < KLD; ++k2)
      if (!FBT_ABL(32) && (part < 0 || k2 * NPW / KLD == part)) {      // (a thread without a k2-th piece stores piece 0 again: same bytes, no predicate)
        kept[k2] = *reinterpret_cast<const f32x4 *>(src + g.loff[k2]);
        float *o = dbases + g.goff[k2] + base;
        if (VEC) *reinterpret_cast<f32x4 *>(o) = kept[k2];  // (non-temporal stores measured the same)
        else { o[0] = kept[k2][0]; o[1] = kept[k2][1]; o[2] = kept[k2][2]; o[3] = kept[k2][3]; }
      }
  };
  FBT_DBG(long long dbg[6] = {0, 0, 0, 0, 0, 0};)
  lds_barrier();
  for (int it = 0;; ++it) {
    const bool has1 = t + Gd < n_tiles, has2 = t + 2 * Gd < n_tiles;
    float *dtc = dt + (TWO ? (it & 1) * (B * ts) : 0);
    FBT_DBG(const long long T0 = FBT_T();)
    const int rp3 = rp_of(t + 3 * Gd);
    const NodeRange s2 = range_of(rp2, has2);
    const Idx x2 = idx_load(e_dst, e_rel, e_val, s2.a, last, lane);
    if (!FBT_ABL(4)) gather_rows_all(gp1, G, x1.es, min(SR, s1.n), d, lane, gstride);
    __builtin_amdgcn_sched_barrier(0);
    // the stores of the tile before go out AFTER this iteration's loads (the wait for those loads at the bottom -- vmcnt counts in order -- then
    // leaves the stores a whole iteration to drain) and in NPW shares, one ahead of each node's message loops: 51 KB per tile and CU take
    // ~3 us to leave the CU, and a store blocks at issue once the write queue is full
    __builtin_amdgcn_sched_barrier(0);
    FBT_DBG(const long long T1 = FBT_T();)
    // ---- nodes wave * NPW .. of tile t: the wave owns their B x d gradients (zero for a node without messages)
    Idx c_x = x;
    int cb0 = 0, sb0 = 0;                                   // windows over the wave's messages: 64 indices from cb0, SR strip rows from sb0
    for (int i2 = 0; i2 < NPW; ++i2) {
      const int nl = wave * NPW + i2;                       // (every tile has all its TNV nodes)
      if (TWO) write_out(dt + ((it + 1) & 1) * (B * ts), i2);
      __builtin_amdgcn_sched_barrier(0);
      const int off0 = rlane(rp, wave * NPW + i2) - s.a, off1 = rlane(rp, wave * NPW + i2 + 1) - s.a;
      f32x4 acc[NBTM];
#pragma unroll
      for (int tb2 = 0; tb2 < NBTM; ++tb2) acc[tb2] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int g0 = off0; g0 < off1 && !FBT_ABL(1); g0 += 16) {
        const int n16 = min(16, off1 - g0);
        if (g0 + n16 > cb0 + 64) {                          // past the 64 prefetched indices: the next 64, on demand
          cb0 = g0;
          c_x = idx_load(e_dst, e_rel, e_val, s.a + cb0, last, lane);
          FBT_ARRIVED(c_x.es); FBT_ARRIVED(c_x.er); FBT_ARRIVED(c_x.ev);
        }
        if (g0 + n16 > sb0 + SR) {                          // past the prefetched rows
          sb0 = g0;
          const int nrow = min(SR, min(s.n, cb0 + 64) - g0);  // a whole window (the next node's group may start inside it)
          float gq[NQ] = {};
          gather_rows_at(gq, G, c_x.es, g0 - cb0, nrow, d, lane, gstride);
          strip_store<GS>(gs, gq, nrow, lane);
        }
        for (int j0 = 0; j0 < n16; j0 += 4) {               // four messages per step: k = the message
          const int jm = min(j0 + k, n16 - 1), src = g0 - cb0 + jm;
          const int r = __shfl(c_x.er, src, 64);
          const float vs = __shfl(c_x.ev, src, 64);
          const float v = (j0 + k < n16) ? vs : 0.f;
          const float bv = gs[(g0 - sb0 + jm) * GS + min(c, GS - 1)];
#pragma unroll
          for (int tb2 = 0; tb2 < NBTM; ++tb2)
            if (tb2 < nbt) acc[tb2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ctab[r * B + min(16 * tb2 + c, B - 1)] * v, bv, acc[tb2], 0, 0, 0);
        }
      }
      if (c < d) {                                          // D: lane (q, column i): rows 4 q .. 4 q + 3 of every 16-row tile of bases
#pragma unroll
        for (int tb2 = 0; tb2 < NBTM; ++tb2)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int b = 16 * tb2 + 4 * k + e;
            if (tb2 < nbt && b < B) dtc[b * ts + nl * d + c] = acc[tb2][e];
          }
      }
    }
    t_out = t;
    if (TWO) {
#pragma unroll
      for (int k2 = 0; k2 < KLD; ++k2) FBT_KEEP(kept[k2]);
    }
    FBT_DBG(const long long T2 = FBT_T();)
    lds_barrier();
    FBT_DBG(const long long T3 = FBT_T(); dbg[0] += T1 - T0; dbg[1] += T2 - T1; dbg[2] += T3 - T2;)
    if (!TWO) {                                             // one buffer: out it goes, and nobody sums the next tile into it before everybody has read it
      write_out(dtc, -1);
      FBT_DBG(const long long T4 = FBT_T();)
      lds_barrier();
      FBT_DBG(dbg[3] += T4 - T3; dbg[4] += FBT_T() - T4;)
      if (!has1) break;
    } else if (!has1) {
      write_out(dtc, -1);
      break;
    }
    t += Gd;
    FBT_DBG(const long long T5 = FBT_T();)
    rp = rp1; rp1 = rp2; rp2 = rp3; s = s1; s1 = s2; x = x1; x1 = x2;
    strip_store<GS>(gs, gp1, min(SR, s.n), lane);
    FBT_DBG(dbg[5] += FBT_T() - T5;)
  }
  FBT_DBG(if (lane == 0) { unsigned long long *o = rgcn_fbt_dbg + 8 * ((blockIdx.x & 255) * 16 + wave); for (int q = 0; q < 6; ++q) o[q] += dbg[q]; o[6] += 1; })
}

// ---- both gradients from ONE walk (round 5).  The two kernels above each stream a [B, N, d] array once (dcomps READS the table, dbases WRITES
// its gradient) and each gathers the upstream rows of all messages: 5.3 + 4.9 GB of fabric traffic per backward on AM as shipped (PMC,
// profiles/r05_*), both at ~4.1 TB/s -- at the rate of their access pattern, so the only thing left to remove was traffic.  Here wave w OWNS
// node w of the tile for both sums, which makes the tile's LDS image private to the wave slot by slot: ONE buffer serves as the staged
// table tile on the way in and as the gradient tile on the way out (the wave takes its node's block into registers -- the B operand of the
// dcomps product -- and later lays the node's gradient over it), the rows of G are gathered once, the indices read once:
//   dcomps  D[m][b] = sum_i G[s_m, i] bases[b, o, i]     -> val_m D[m][b] added to the workgroup's R x B doubles (ds_add_f64)
//   dbases  D[b][i] = sum_m (val_m comps[r_m, b]) G[s_m, i]   four messages per MFMA step, accumulated in registers over the node's messages
// LDS: R x B doubles + R x B floats (coefficients) + ONE tile (AM: 85.4 + 42.7 + 26.2 KB of 160; the two kernels otherwise).  No strips of
// gradient rows: the wave transposes them through its own slot of the tile (B >= 16: sixteen runs of d floats).
// Per iteration two LDS-only barriers: gradients complete -> [every thread: its pieces of the gradient out to HBM, the next tile's pieces (requested
// one hand-over earlier: a whole iteration in flight) into the buffer, the tile after that requested] -> staged.
// NK consecutive floats at any 4-byte alignment (gfx950 takes unaligned LDS accesses: the compiler may use one wide ds_read / ds_write)
template <int NK>
struct __attribute__((packed, aligned(4))) FK { float f[NK]; };
struct __attribute__((packed, aligned(4))) F2U { float f[2]; };

template <int NKD, int KLD, bool VEC, int NBTM>
__global__ __launch_bounds__(TW) void fbn_bwd_kernel(
    const float *__restrict__ bases, const float *__restrict__ comps, const float *__restrict__ G, float *__restrict__ dbases,
    float *__restrict__ dC, const int *__restrict__ rowptr, const int *__restrict__ e_dst, const int *__restrict__ e_rel,
    const float *__restrict__ e_val, int n_tiles, int N, int R, int B, int d, int ts, int last, int gstride, int gn, int abl) {
  // (NBTM = ceil(B / 16): the 16-row tiles of bases, a template parameter -- registers)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = lane >> 4, c = lane & 15;
  constexpr int nbt = NBTM;
  double *dcl = reinterpret_cast<double *>(lds);            // [R][B]
  float *ctab = lds + ((2 * R * B + 3) & ~3);               // [R][B]
  float *xt = ctab + ((R * B + 3) & ~3);                    // [B][ts]: the staged tile, then its gradient
  for (int j = tid; j < R * B; j += TW) { dcl[j] = 0.0; ctab[j] = comps[j]; }
  // the thread's 16-byte pieces of a tile, packed (b << 16 | q: registers are what this kernel is short of); piece = floats 4 q .. 4 q + 3 of
  // basis b's run; a thread without a k-th piece aliases piece 0 for its loads (they stay unconditional) and skips the LDS / store side
  const int q4 = TN * d / 4;
  int bq[KLD];
#pragma unroll
  for (int k2 = 0; k2 < KLD; ++k2) {
    const int idx = tid + k2 * TW;
    const bool act = idx < B * q4;
    const int b = act ? idx / q4 : 0;
    bq[k2] = (b << 16) | (act ? idx - b * q4 : 0);
  }
  const long long Nd = (long long)N * d;
  auto piece_act = [&](int k2) { return tid + k2 * TW < B * q4; };
  auto tile_load = [&](f32x4 (&st)[KLD], long long base) {
#pragma unroll
    for (int k2 = 0; k2 < KLD; ++k2) {
      const float *p = bases + (bq[k2] >> 16) * Nd + 4 * (bq[k2] & 0xffff) + base;
      if (VEC) st[k2] = *reinterpret_cast<const f32x4 *>(p);
      else st[k2] = f32x4{p[0], p[1], p[2], p[3]};
    }
  };
  const int Gd = gridDim.x;
  int t = blockIdx.x;
  auto rp_of = [&](int tt) {
    tt = min(tt, n_tiles - 1);
    return rowptr[min((long long)tt * TN + min(lane, TN), (long long)N)];
  };
  auto base_of = [&](int tt) { return (long long)min(min(tt, n_tiles - 1) * TN, N - TN) * d; };
  auto rows_first = [&](float (&gp)[GQ], int es, int n) {    // the first rows of the wave's run: always GQ loads (see gather_rows_all)
    const int m = lane >> 4, cc = min(lane & 15, d - 1);
#pragma unroll
    for (int q = 0; q < GQ; ++q) {
      const int sidx = __shfl(es, min(4 * q + m, max(n - 1, 0)), 64);
      gp[q] = G[(size_t)sidx * gstride + cc];
    }
  };
  auto rows_at = [&](float (&gp)[GQ], int es, int off, int n) {
    const int m = lane >> 4, cc = min(lane & 15, d - 1);
#pragma unroll
    for (int q = 0; q < GQ; ++q)
      if (4 * q < n) {
        const int sidx = __shfl(es, off + min(4 * q + m, n - 1), 64);
        gp[q] = G[(size_t)sidx * gstride + cc];
      }
  };

  f32x4 st[KLD], kept[KLD];
  int rp = rp_of(t), rp1 = rp_of(t + Gd), rp2 = rp_of(t + 2 * Gd);
  tile_load(st, base_of(t));
  NodeRange s = node_range(rp, wave, true);
  Idx x = idx_load(e_dst, e_rel, e_val, s.a, last, lane);
  NodeRange s1 = node_range(rp1, wave, t + Gd < n_tiles);
  Idx x1 = idx_load(e_dst, e_rel, e_val, s1.a, last, lane);
  float gp[GQ], gp1[GQ];                                    // rows of the current tile's messages, of the next tile's
  rows_first(gp, x.es, min(gn, s.n));
#pragma unroll
  for (int k2 = 0; k2 < KLD; ++k2)
    if (piece_act(k2)) *reinterpret_cast<f32x4 *>(xt + (bq[k2] >> 16) * ts + 4 * (bq[k2] & 0xffff)) = st[k2];
  tile_load(st, base_of(t + Gd));
#pragma unroll
  for (int k2 = 0; k2 < KLD; ++k2) kept[k2] = f32x4{0.f, 0.f, 0.f, 0.f};
  lds_barrier();

  // one tile per iteration; st = tile t + Gd, requested one hand-over ago and laid down at this iteration's hand-over
  FBT_DBG(long long dbg[7] = {0, 0, 0, 0, 0, 0, 0};)
  for (;;) {
    const bool has1 = t + Gd < n_tiles, has2 = t + 2 * Gd < n_tiles;
    FBT_DBG(const long long T0 = FBT_T();)
    int rp3 = rp_of(t + 3 * Gd);
    const NodeRange s2 = node_range(rp2, wave, has2);
    const Idx x2 = idx_load(e_dst, e_rel, e_val, s2.a, last, lane);
    if (!FBT_ABL(4)) rows_first(gp1, x1.es, min(gn, s1.n));
    __builtin_amdgcn_sched_barrier(0);
    FBT_DBG(const long long T1 = FBT_T();)
    // ---- node `wave` of tile t: slot nl of the staged tile (the last tile is staged from node N - 16: its first `shift` slots belong to the tile before)
    const int shift = t * TN - min(t * TN, N - TN);
    const int nl = wave + shift;
    f32x4 accD[NBTM];
#pragma unroll
    for (int tb2 = 0; tb2 < NBTM; ++tb2) accD[tb2] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (s.n > 0 && !FBT_ABL(1)) {
      // the node's block transposed: B operand lane (k, c) = bases[16 tb + c][o][feature of K index (ks, k)].  The contraction runs over the
      // features in ANY order as long as both operands agree: K index (ks, k) <-> feature NKD k + ks, so that a lane's NKD features are
      // consecutive floats -- ONE LDS read per 16-row tile (and per message row below) instead of NKD: the message loops are bound by the
      // CU's LDS instruction rate (~55 wave-level LDS instructions per node before, phase timers of the ablation build)
      float bt[NBTM][NKD];
#pragma unroll
      for (int tb2 = 0; tb2 < NBTM; ++tb2) {
        FK<NKD> v = *reinterpret_cast<const FK<NKD> *>(xt + min(16 * tb2 + c, B - 1) * ts + nl * d + min(NKD * k, d - 1));
#pragma unroll
        for (int ks = 0; ks < NKD; ++ks) bt[tb2][ks] = (16 * tb2 + c < B && NKD * k + ks < d) ? v.f[ks] : 0.f;
      }
      Idx c_x = x;
      int cb0 = 0;                                          // c_x holds the 64 indices from message cb0 of the node
      for (int g0 = 0; g0 < s.n; g0 += gn) {
        const int n16 = min(gn, s.n - g0);
        if (g0) {                                           // more than gn messages (rare): this pass' indices and rows on demand -- the
          cb0 = g0;                                         // prefetched destination ids are not kept for it (registers)
          c_x = idx_load(e_dst, e_rel, e_val, s.a + cb0, last, lane);
          FBT_ARRIVED(c_x.es); FBT_ARRIVED(c_x.er); FBT_ARRIVED(c_x.ev);
          rows_at(gp, c_x.es, g0 - cb0, n16);
#pragma unroll
          for (int q = 0; q < GQ; ++q) FBT_ARRIVED(gp[q]);   // (the wait belongs HERE: left to the compiler, every later use of gp -- and of the registers
        }                                                   //  it shares -- waits for ALL outstanding loads, the next tile's included)
        const int w0 = g0 - cb0;
        // dcomps: A operand lane (k, m) = G row of message m, feature 4 ks + k
        // The A operand wants the same rows transposed (lane (k, m) = feature 4 ks + k of message m).  The transposition strip is the wave's
        // OWN slot of the tile: its block sits in registers by now (bt) and its gradient is laid down only after the last pass, so in between
        // the slot's first 16 runs of d floats are 16 rows of scratch -- the LDS has no room for strips next to the tables (same wave, LDS
        // operations in order: no barrier)
#pragma unroll
        for (int q = 0; q < GQ; ++q)
          if (4 * q < n16 && c < d) xt[(4 * q + k) * ts + nl * d + c] = gp[q];
        float av[NKD];
        {
          // A row c holds message pm = 4 (c & 3) + (c >> 2) -- the 4 x 4 transpose of the row index -- so that D's row 4 k + e is message 4 e + k:
          // the e-th table update below then covers messages 4 e .. 4 e + 3 and is SKIPPED whole (no lane active) when the node has no more
          // than 4 e messages.  With rows in message order every one of the four updates ran for any node of 4 or more messages, half empty:
          // 12 ds_add_f64 per node and 16-row tile of bases at AM's 8 messages per node instead of 6.
          const int pm = min(4 * (c & 3) + (c >> 2), n16 - 1);
          const float vm = __shfl(c_x.ev, w0 + pm, 64);       // val of the row's message rides on the A operand: D[m][b] comes out scaled
          const FK<NKD> v = *reinterpret_cast<const FK<NKD> *>(xt + pm * ts + nl * d + min(NKD * k, d - 1));
#pragma unroll
          for (int ks = 0; ks < NKD; ++ks) av[ks] = NKD * k + ks < d ? vm * v.f[ks] : 0.f;
          // (K positions past d are zeroed HERE: they are read from whatever follows the row in LDS -- the neighbour slot, or padding nobody
          // ever wrote -- and "they meet zeros of bt" only holds while that garbage is finite: NaN x 0 is NaN.  Found as comps.grad full of
          // NaN in 1 of 25 runs of the full-size AM test, and in every launch of a process whose LDS padding happened to hold NaN patterns)
        }
        int rq[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) rq[e] = __shfl(c_x.er, w0 + min(4 * e + k, n16 - 1), 64);       // D's rows of this lane: row 4 k + e = message 4 e + k
#pragma unroll
        for (int tb2 = 0; tb2 < NBTM; ++tb2)
          if (tb2 < nbt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKD; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bt[tb2][ks], acc, 0, 0, 0);
            if (16 * tb2 + c < B && !FBT_ABL(8)) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (4 * e + k < n16)
                  __hip_atomic_fetch_add(dcl + rq[e] * B + 16 * tb2 + c, (double)acc[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
        // dbases: four messages per step, k = the message of the step
#pragma unroll
        for (int q = 0; q < GQ; ++q)
          if (4 * q < n16) {                                  // (uniform) step q: messages 4 q .. 4 q + 3, B operand = the gathered rows as they are
            const int jm = min(4 * q + k, n16 - 1);
            const int r = __shfl(c_x.er, w0 + jm, 64);
            const float vs = __shfl(c_x.ev, w0 + jm, 64);
            const float v = (4 * q + k < n16) ? vs : 0.f;
#pragma unroll
            for (int tb2 = 0; tb2 < NBTM; ++tb2)
              if (tb2 < nbt) accD[tb2] = __builtin_amdgcn_mfma_f32_16x16x4f32(gp[q], ctab[r * B + min(16 * tb2 + c, B - 1)] * v, accD[tb2], 0, 0, 0);
          }
      }
    }
    // the node's gradient over the node's block (the wave is the only reader and the only writer of slot nl).  The product is taken
    // transposed (A = the gathered rows, B = the scaled coefficients): D lane (q, column b) holds features 4 q .. 4 q + 3 of basis row
    // 16 tb + b -- consecutive floats, two 8-byte LDS writes per tile of bases instead of four scattered ones; a node without messages writes zeros
    if (nl < TN) {
#pragma unroll
      for (int tb2 = 0; tb2 < NBTM; ++tb2)
        if (16 * tb2 + c < B) {
          float *o = xt + (16 * tb2 + c) * ts + nl * d + 4 * k;
          if (4 * k + 1 < d) *reinterpret_cast<F2U *>(o) = F2U{{accD[tb2][0], accD[tb2][1]}};
          else if (4 * k < d) o[0] = accD[tb2][0];
          if (4 * k + 3 < d) *reinterpret_cast<F2U *>(o + 2) = F2U{{accD[tb2][2], accD[tb2][3]}};
          else if (4 * k + 2 < d) o[2] = accD[tb2][2];
        }
    }
#pragma unroll
    for (int k2 = 0; k2 < KLD; ++k2) FBT_KEEP(kept[k2]);      // (the registers of the tile before's stores stay untouched until here)
    // Rotate everything that was LOADED (next tile's indices and rows) BEFORE this tile's stores are issued: vmcnt counts in order, so a
    // register move of a loaded value placed after the stores waits for the stores to COMPLETE -- every iteration then paid its own
    // write latency before it could start the next tile (round 5, found in the ISA: s_waitcnt vmcnt(2) ahead of these moves)
    FBT_DBG(const long long T2 = FBT_T();)
    const NodeRange s_next = s1;
    x = x1; x1 = x2;
#pragma unroll
    for (int q = 0; q < GQ; ++q) gp[q] = gp1[q];
    FBT_ARRIVED(x.er); FBT_ARRIVED(x.ev); FBT_ARRIVED(x1.es); FBT_ARRIVED(x1.er); FBT_ARRIVED(x1.ev);
#pragma unroll
    for (int q = 0; q < GQ; ++q) FBT_ARRIVED(gp[q]);
    FBT_ARRIVED(rp3);
    FBT_DBG(const long long T3 = FBT_T();)
    lds_barrier();
    FBT_DBG(const long long T4 = FBT_T();)
    // hand-over: every thread takes its pieces of the gradient out and puts the next tile's pieces in
    {
      const long long base = base_of(t);
      const int lo = shift * d;                             // the last tile: positions below `lo` are the tile before's
#pragma unroll
      for (int k2 = 0; k2 < KLD; ++k2)
        if (piece_act(k2)) {
          const int pos = 4 * (bq[k2] & 0xffff);
          float *p = xt + (bq[k2] >> 16) * ts + pos;
          kept[k2] = *reinterpret_cast<const f32x4 *>(p);
          float *o = dbases + (bq[k2] >> 16) * Nd + pos + base;
          if (!FBT_ABL(32)) {
            if (VEC && pos >= lo) *reinterpret_cast<f32x4 *>(o) = kept[k2];
            else {
#pragma unroll
              for (int cc = 0; cc < 4; ++cc)
                if (pos + cc >= lo) o[cc] = kept[k2][cc];
            }
          }
          if (has1) *reinterpret_cast<f32x4 *>(p) = st[k2];
        }
      if (!FBT_ABL(2)) tile_load(st, base_of(t + 2 * Gd));     // (past the end: the last tile again, never laid down)
    }
    FBT_DBG(const long long T5 = FBT_T();)
    lds_barrier();
    FBT_DBG(dbg[0] += T1 - T0; dbg[1] += T2 - T1; dbg[2] += T3 - T2; dbg[3] += T4 - T3; dbg[4] += T5 - T4; dbg[5] += FBT_T() - T5; dbg[6] += 1;)
    if (!has1) break;
    t += Gd;
    rp = rp1; rp1 = rp2; rp2 = rp3; s = s_next; s1 = s2;
  }
  for (int j = tid; j < R * B; j += TW) {
    const float v = (float)dcl[j];
    if (v != 0.f) atomicAdd(dC + j, v);
  }
  FBT_DBG(if (lane == 0) { unsigned long long *o = rgcn_fbt_dbg + 8 * ((blockIdx.x & 255) * 16 + wave); for (int q = 0; q < 7; ++q) o[q] += dbg[q]; })
}

inline int pow2_at_least(int v, int lo) {
  int p = lo;
  while (p < v) p <<= 1;
  return p;
}
struct TileShape { int dp, nreg, bp, kld, ts_f, ts_b, dpb, nks, kld2, kld8; size_t lds_fwd, lds_dc, lds_db, lds_fwd_n, lds_db_n, lds_db_n2, lds_db_n8; };
inline bool tile_shape(int R, int B, int d, long long N, TileShape &s) {
  if (R <= 0 || B < 1 || B > 64 || d < 1 || d > 16 || N < TN) return false;
  s.dp = pow2_at_least(d, 4);
  const int ngrp = 64 / s.dp;
  s.nreg = 4 * ((B + 4 * ngrp - 1) / (4 * ngrp));
  if (s.nreg > 16) return false;
  s.bp = s.nreg * ngrp;
  const int pieces = B * 4 * d;
  if (pieces > 4 * TW) return false;
  s.kld = pieces <= 2 * TW ? 2 : 4;
  s.ts_f = TN * d + 4;                                      // forward: 16-byte pieces stay aligned, basis groups land on different banks
  s.ts_b = TN * d + 1;                                      // backward: lane = basis reads / adds at stride ts -> odd
  s.dpb = 4 * ((d + 3) / 4);
  s.lds_fwd = ((size_t)((R * s.bp + 3) & ~3) + 2 * (size_t)B * s.ts_f + (size_t)TWV * PER * s.dp) * 4;
  const size_t strips = (size_t)TWV * PERB * s.dpb * 4;    // the waves' strips of gradient rows
  s.lds_dc = (size_t)R * B * 8 + 16 + strips + 2 * (size_t)B * s.ts_b * 4;
  s.lds_db = 2 * (size_t)B * s.ts_b * 8 + strips + (size_t)R * B * 4;
  s.nks = (B + 3) / 4;                                      // one wave per node (mode 1): K steps of the forward's MFMAs
  s.lds_fwd_n = ((size_t)((R * (4 * s.nks + 1) + 3) & ~3) + 2 * (size_t)B * s.ts_f + (size_t)TWV * 16 * 16) * 4;
  s.lds_db_n = 2 * (size_t)B * s.ts_f * 4 + strips + (size_t)R * B * 4;
  // dbases with 32-node tiles (two nodes per wave): row stride 32 d + 4, B x 8 d pieces
  s.lds_db_n2 = 2 * (size_t)B * (2 * TN * d + 4) * 4 + 2 * strips + (size_t)R * B * 4;       // (strips of 32 rows)
  s.kld2 = 2 * pieces <= 2 * TW ? 2 : 4;
  if (2 * pieces > 4 * TW || N < 2 * TN) s.lds_db_n2 = (size_t)LDS_MAX + 1;
  // dbases in 512-thread workgroups, two per CU: 16-node tiles, two nodes per wave, one tile buffer
  s.lds_db_n8 = (size_t)B * s.ts_f * 4 + strips + (size_t)R * B * 4;                        // (8 waves, strips of 32 rows)
  s.kld8 = pieces <= 2 * 512 ? 2 : 4;
  if (pieces > 4 * 512) s.lds_db_n8 = (size_t)LDS_MAX + 1;
  return true;
}
// LDS image of the one-walk backward (fbn_bwd_kernel): R x B doubles, R x B floats, one tile
inline size_t fused_lds(const TileShape &s, int R, int B, int /*gn*/) {
  return ((size_t)((2 * R * B + 3) & ~3) + (size_t)((R * B + 3) & ~3) + (size_t)B * s.ts_f) * 4;
}
int n_cus() {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n_cu = v;
    else n_cu = 256;
  }
  return n_cu;
}
template <typename K>
hipError_t raise_lds(K kernel, size_t bytes) {
  return bytes > 64 * 1024 ? hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX) : hipSuccess;
}

}  // namespace

#ifdef RGCN_ABLATIONS
/* ablation library only: 100 MHz ticks summed over the waves of all forward tile launches since the last reset:
 * {arrival + tile store + Y flush, issue of the next loads, message loops, barrier, rotation, waves} */
extern "C" __attribute__((visibility("default"))) int rgcn_fbt_debug_read(unsigned long long *out8, int reset) {
  static unsigned long long h[8 * 256 * 16];
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(rgcn_fbt_dbg), sizeof(h)));
  for (int i = 0; i < 8; ++i) out8[i] = 0;
  for (int w = 0; w < 256 * 16; ++w)
    for (int i = 0; i < 8; ++i) out8[i] += h[8 * w + i];
  if (reset) { memset(h, 0, sizeof(h)); HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(rgcn_fbt_dbg), h, sizeof(h))); }
  return RGCN_OK;
}
#endif

extern "C" int rgcn_fbasis_tile_supported(int32_t R, int32_t B, int32_t d, int64_t n_nodes) {
  TileShape s;
  if (!tile_shape(R, B, d, n_nodes, s)) return 0;
  return (s.lds_fwd <= (size_t)LDS_MAX ? 1 : 0) | ((s.lds_dc <= (size_t)LDS_MAX && s.lds_db <= (size_t)LDS_MAX) ? 2 : 0) |
         (s.lds_fwd_n <= (size_t)LDS_MAX ? 4 : 0) | ((s.lds_dc <= (size_t)LDS_MAX && s.lds_db_n <= (size_t)LDS_MAX) ? 8 : 0);
}

extern "C" int rgcn_fbasis_tile_ystride(int32_t d) { return d >= 1 && d <= 16 ? pow2_at_least(d, 4) : 0; }

extern "C" int rgcn_fbasis_tile_fwd_f32(const float *bases, const float *comps, float *Y, const int32_t *rowptr, const int32_t *e_rel,
                                        const float *e_val, int64_t n_messages, int64_t n_nodes, int32_t R, int32_t B, int32_t d, int32_t mode,
                                        void *stream) {
  TileShape s;
  if (n_messages == 0) return RGCN_OK;
  if (!bases || !comps || !Y || !rowptr || !e_rel || !e_val || n_messages < 0 || n_messages > INT32_MAX) { rgcn_set_error("fbasis_tile_fwd: bad argument"); return RGCN_EINVAL; }
  const int last = (int)(n_messages - 1);
  const int abl = rgcn_option_value(RGCN_OPT_BWD_ABL);      // 0 in the shipped library (rgcn_set_option refuses it)
  if (!tile_shape(R, B, d, n_nodes, s) || (mode ? s.lds_fwd_n : s.lds_fwd) > (size_t)LDS_MAX || (mode != 0 && mode != 1)) { rgcn_set_error("fbasis_tile_fwd: shape outside the tile kernel (rgcn_fbasis_tile_supported)"); return RGCN_EUNSUPPORTED; }
  const int n_tiles = (int)((n_nodes + TN - 1) / TN);
  const bool vec = ((n_nodes * d) % 4 == 0) && (reinterpret_cast<uintptr_t>(bases) % 16 == 0);
  const dim3 grid((unsigned)std::min<int64_t>(n_tiles, n_cus()));
  hipStream_t st = (hipStream_t)stream;
  if (mode == 1) {                                          // one wave per node, MFMA
    const int ys = pow2_at_least(d, 4);
#define FBN_FWD3(NK_, KL_, VE_)                                                                                                 \
  {                                                                                                                             \
    HIP_TRY(raise_lds(fbn_fwd_kernel<NK_, KL_, VE_>, s.lds_fwd_n));                                                             \
    hipLaunchKernelGGL((fbn_fwd_kernel<NK_, KL_, VE_>), grid, dim3(TW), s.lds_fwd_n, st, bases, comps, Y, rowptr, e_rel, e_val, n_tiles, \
                       (int)n_nodes, R, B, d, s.ts_f, last, ys, abl);                                                            \
  }
#define FBN_FWD2(NK_, KL_) { if (vec) FBN_FWD3(NK_, KL_, true) else FBN_FWD3(NK_, KL_, false) }
#define FBN_FWD1(NK_) { if (s.kld == 2) FBN_FWD2(NK_, 2) else FBN_FWD2(NK_, 4) }
    if (s.nks <= 4) FBN_FWD1(4) else if (s.nks <= 8) FBN_FWD1(8) else if (s.nks <= 12) FBN_FWD1(12) else FBN_FWD1(16)
#undef FBN_FWD1
#undef FBN_FWD2
#undef FBN_FWD3
    HIP_TRY(hipGetLastError());
    return RGCN_OK;
  }
#define FBT_FWD3(DP_, NR_, KL_)                                                                                                   \
  {                                                                                                                               \
    if (vec) {                                                                                                                    \
      HIP_TRY(raise_lds(fbt_fwd_kernel<DP_, NR_, KL_, true>, s.lds_fwd));                                                         \
      hipLaunchKernelGGL((fbt_fwd_kernel<DP_, NR_, KL_, true>), grid, dim3(TW), s.lds_fwd, st, bases, comps, Y, rowptr, e_rel, e_val, n_tiles, \
                         (int)n_nodes, R, B, d, s.ts_f, last, abl);                                                                          \
    } else {                                                                                                                      \
      HIP_TRY(raise_lds(fbt_fwd_kernel<DP_, NR_, KL_, false>, s.lds_fwd));                                                        \
      hipLaunchKernelGGL((fbt_fwd_kernel<DP_, NR_, KL_, false>), grid, dim3(TW), s.lds_fwd, st, bases, comps, Y, rowptr, e_rel, e_val, n_tiles, \
                         (int)n_nodes, R, B, d, s.ts_f, last, abl);                                                                          \
    }                                                                                                                             \
  }
#define FBT_FWD2(DP_, NR_) { if (s.kld == 2) FBT_FWD3(DP_, NR_, 2) else FBT_FWD3(DP_, NR_, 4) }
#define FBT_FWD1(DP_) { if (s.nreg <= 4) FBT_FWD2(DP_, 4) else if (s.nreg <= 8) FBT_FWD2(DP_, 8) else if (s.nreg <= 12) FBT_FWD2(DP_, 12) else FBT_FWD2(DP_, 16) }
  if (s.dp == 4) FBT_FWD1(4) else if (s.dp == 8) FBT_FWD1(8) else FBT_FWD1(16)
#undef FBT_FWD1
#undef FBT_FWD2
#undef FBT_FWD3
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

namespace {
__global__ __launch_bounds__(1024) void poison_lds_kernel(int n_words) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < n_words; i += 1024) lds[i] = __int_as_float(0x7fc00000 | (i & 0xffff));
  __syncthreads();
  if (lds[(threadIdx.x * 37) % n_words] == 0.f) asm volatile("s_nop 0");     // (keeps the stores: nothing else reads them)
}
}  // namespace

extern "C" int rgcn_poison_lds(void *stream) {
  const int bytes = 160 * 1024;
  static bool raised = false;
  if (!raised) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(poison_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    raised = true;
  }
  // one workgroup per CU holds the whole LDS, so 4 x the CU count of them visit every CU at least once
  hipLaunchKernelGGL(poison_lds_kernel, dim3((unsigned)(4 * n_cus())), dim3(1024), bytes, (hipStream_t)stream, bytes / 4);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_gather_rows_sum4_f32(const float *Y, int32_t ys, const int32_t *perm, const int32_t *units, int64_t n_units,
                                         int64_t n_split, const float *bias, float *out, int64_t n_rows, int32_t w, int32_t out_stride,
                                         int32_t relu, void *stream) {
  const int ow = out_stride;
  if (n_units < 0 || n_rows < 0 || w <= 0 || w > ys || ow < w || ow > ys || (ys != 4 && ys != 8 && ys != 16) || (n_units && (!Y || !perm || !units || !out))) {
    rgcn_set_error("gather_rows_sum4: bad argument");
    return RGCN_EINVAL;
  }
  if (relu && n_split) { rgcn_set_error("gather_rows_sum4: relu in the epilogue needs rows that are not cut into shared pieces"); return RGCN_EUNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  if (n_split) HIP_TRY(zero_async(out, (size_t)n_rows * ow * sizeof(float), st));
  if (n_units == 0) return RGCN_OK;
  const int lpr = ys / 4;
  const dim3 grid((unsigned)std::min<int64_t>((n_units * lpr + 255) / 256, (int64_t)n_cus() * 64));
  const int4 *un = reinterpret_cast<const int4 *>(units);
  const int vec_out = (ow & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  if (lpr == 1) hipLaunchKernelGGL(gather_rows_sum4_kernel<1>, grid, dim3(256), 0, st, Y, perm, un, bias, out, (long long)n_units, w, ow, relu ? 1 : 0, vec_out);
  else if (lpr == 2) hipLaunchKernelGGL(gather_rows_sum4_kernel<2>, grid, dim3(256), 0, st, Y, perm, un, bias, out, (long long)n_units, w, ow, relu ? 1 : 0, vec_out);
  else hipLaunchKernelGGL(gather_rows_sum4_kernel<4>, grid, dim3(256), 0, st, Y, perm, un, bias, out, (long long)n_units, w, ow, relu ? 1 : 0, vec_out);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_fbasis_tile_bwd_fused_gn(int32_t R, int32_t B, int32_t d, int64_t n_nodes) {
  TileShape s;
  if (!tile_shape(R, B, d, n_nodes, s)) return 0;
  // the instantiations that keep their state in 128 registers (1024 threads): two pieces per thread, and not four 16-row tiles of bases next
  // to three or four K steps -- the others spill into scratch inside the pipeline and stay on the two kernels
  const int nbt = (B + 15) / 16, nkd = s.dpb / 4;
  if (s.kld != 2 || nbt * nkd > 9 || (nbt == 4 && nkd >= 2) || B < 16) return 0;     // (B >= 16: the slot doubles as a 16-row strip)
  return fused_lds(s, R, B, 16) <= (size_t)LDS_MAX ? 16 : 0;      // messages of a node per pass
}

extern "C" int rgcn_fbasis_tile_bwd_f32(const float *bases, const float *comps, const float *G, int32_t g_stride, float *dbases, float *dcomps,
                                        const int32_t *rowptr, const int32_t *e_dst, const int32_t *e_rel, const float *e_val,
                                        int64_t n_messages, int64_t n_nodes, int32_t R, int32_t B, int32_t d, int32_t mode, void *stream) {
  TileShape s;
  const int gstride = g_stride;
  if (g_stride < d) { rgcn_set_error("fbasis_tile_bwd: row stride of the upstream gradient below its width"); return RGCN_EINVAL; }
  if (!bases || !comps || !G || !rowptr || !e_dst || !e_rel || !e_val || (!dbases && !dcomps) || n_messages < 0 || n_messages > INT32_MAX) { rgcn_set_error("fbasis_tile_bwd: bad argument"); return RGCN_EINVAL; }
  if (n_messages == 0) {                                    // no messages: both gradients are zero
    if (dbases) HIP_TRY(zero_async(dbases, (size_t)B * n_nodes * d * sizeof(float), (hipStream_t)stream));
    if (dcomps) HIP_TRY(zero_async(dcomps, (size_t)R * B * sizeof(float), (hipStream_t)stream));
    return RGCN_OK;
  }
  const int last = (int)(n_messages - 1);
  const int abl = rgcn_option_value(RGCN_OPT_BWD_ABL);      // 0 in the shipped library (rgcn_set_option refuses it)
  if (!tile_shape(R, B, d, n_nodes, s) || s.lds_dc > (size_t)LDS_MAX || (mode ? s.lds_db_n : s.lds_db) > (size_t)LDS_MAX || (mode != 0 && mode != 1 && mode != 3)) { rgcn_set_error("fbasis_tile_bwd: shape outside the tile kernels (rgcn_fbasis_tile_supported)"); return RGCN_EUNSUPPORTED; }
  const int n_tiles = (int)((n_nodes + TN - 1) / TN);
  const bool vec = ((n_nodes * d) % 4 == 0) && (reinterpret_cast<uintptr_t>(bases) % 16 == 0) && (!dbases || reinterpret_cast<uintptr_t>(dbases) % 16 == 0);
  const dim3 grid((unsigned)std::min<int64_t>(n_tiles, n_cus()));
  hipStream_t st = (hipStream_t)stream;
#define FBT_BWD3(KERNEL, LDSB, DPB_, KL_, ...)                                                                            \
  {                                                                                                                       \
    if (vec) {                                                                                                            \
      HIP_TRY(raise_lds(KERNEL<DPB_, KL_, true>, LDSB));                                                                  \
      hipLaunchKernelGGL((KERNEL<DPB_, KL_, true>), grid, dim3(TW), LDSB, st, __VA_ARGS__);                               \
    } else {                                                                                                              \
      HIP_TRY(raise_lds(KERNEL<DPB_, KL_, false>, LDSB));                                                                 \
      hipLaunchKernelGGL((KERNEL<DPB_, KL_, false>), grid, dim3(TW), LDSB, st, __VA_ARGS__);                              \
    }                                                                                                                     \
  }
#define FBT_BWD2(KERNEL, LDSB, DPB_, ...) { if (s.kld == 2) FBT_BWD3(KERNEL, LDSB, DPB_, 2, __VA_ARGS__) else FBT_BWD3(KERNEL, LDSB, DPB_, 4, __VA_ARGS__) }
#define FBT_BWD1(KERNEL, LDSB, ...)                                                                                        \
  { if (s.dpb == 4) FBT_BWD2(KERNEL, LDSB, 4, __VA_ARGS__) else if (s.dpb == 8) FBT_BWD2(KERNEL, LDSB, 8, __VA_ARGS__)       \
    else if (s.dpb == 12) FBT_BWD2(KERNEL, LDSB, 12, __VA_ARGS__) else FBT_BWD2(KERNEL, LDSB, 16, __VA_ARGS__) }
  // mode 1 and both gradients wanted: ONE walk (fbn_bwd_kernel) when its LDS image fits; mode 3 = mode 1 on the two kernels (tests, comparisons)
  const int gn_fused = (mode == 1 && dbases && dcomps) ? rgcn_fbasis_tile_bwd_fused_gn(R, B, d, n_nodes) : 0;
  if (gn_fused) {
    HIP_TRY(zero_async(dcomps, (size_t)R * B * sizeof(float), st));
    const size_t lds_f = fused_lds(s, R, B, gn_fused);
#define FBN_F4(NK_, KL_, VE_, NB_)                                                                                                  \
  {                                                                                                                                 \
    HIP_TRY(raise_lds(fbn_bwd_kernel<NK_, KL_, VE_, NB_>, lds_f));                                                                  \
    hipLaunchKernelGGL((fbn_bwd_kernel<NK_, KL_, VE_, NB_>), grid, dim3(TW), lds_f, st, bases, comps, G, dbases, dcomps, rowptr, e_dst, e_rel, e_val, \
                       n_tiles, (int)n_nodes, R, B, d, s.ts_f, last, gstride, gn_fused, abl);                                        \
  }
#define FBN_F3(NK_, KL_, VE_) { if (B <= 16) FBN_F4(NK_, KL_, VE_, 1) else if (B <= 32) FBN_F4(NK_, KL_, VE_, 2) else if (B <= 48) FBN_F4(NK_, KL_, VE_, 3) else FBN_F4(NK_, KL_, VE_, 4) }
#define FBN_F2(NK_, KL_) { if (vec) FBN_F3(NK_, KL_, true) else FBN_F3(NK_, KL_, false) }
#define FBN_F1(NK_) { FBN_F2(NK_, 2) }
    if (s.dpb == 4) FBN_F1(1) else if (s.dpb == 8) FBN_F1(2) else if (s.dpb == 12) FBN_F1(3) else FBN_F1(4)
#undef FBN_F1
#undef FBN_F2
#undef FBN_F3
#undef FBN_F4
    HIP_TRY(hipGetLastError());
    return RGCN_OK;
  }
  if (mode != 0) {                                          // one wave per node, MFMA: template on the K steps over the features (d / 4)
#define FBN_BWD2(KERNEL, LDSB, NK_, ...) { if (s.kld == 2) FBT_BWD3(KERNEL, LDSB, NK_, 2, __VA_ARGS__) else FBT_BWD3(KERNEL, LDSB, NK_, 4, __VA_ARGS__) }
#define FBN_BWD1(KERNEL, LDSB, ...)                                                                                         \
  { if (s.dpb == 4) FBN_BWD2(KERNEL, LDSB, 1, __VA_ARGS__) else if (s.dpb == 8) FBN_BWD2(KERNEL, LDSB, 2, __VA_ARGS__)        \
    else if (s.dpb == 12) FBN_BWD2(KERNEL, LDSB, 3, __VA_ARGS__) else FBN_BWD2(KERNEL, LDSB, 4, __VA_ARGS__) }
    if (dbases) {
      // form: two 512-thread workgroups per CU (16-node tiles, one buffer) when two of them fit the LDS; else 32-node tiles, one workgroup
      // per CU; else 16-node tiles.  (Ablation build: bwd_abl bit 64 skips the first form, bit 128 the second too.)
      const bool pair = 2 * s.lds_db_n8 <= (size_t)LDS_MAX && !(abl & 64);
      const bool two = !pair && s.lds_db_n2 <= (size_t)LDS_MAX && !(abl & 128);
      const int tiles_db = two ? (int)((n_nodes + 2 * TN - 1) / (2 * TN)) : n_tiles;
      const int ts_db = two ? 2 * TN * d + 4 : s.ts_f;
      const size_t lds_db = pair ? s.lds_db_n8 : (two ? s.lds_db_n2 : s.lds_db_n);
      const int kld_db = pair ? s.kld8 : (two ? s.kld2 : s.kld);
      const dim3 grid_db((unsigned)std::min<int64_t>(tiles_db, (int64_t)n_cus() * (pair ? 2 : 1)));
#define FBN_DB3(NK_, KL_, VE_, NP_, NW_)                                                                                          \
  {                                                                                                                               \
    HIP_TRY(raise_lds(fbn_dbases_kernel<NK_, KL_, VE_, NP_, NW_>, lds_db));                                                        \
    hipLaunchKernelGGL((fbn_dbases_kernel<NK_, KL_, VE_, NP_, NW_>), grid_db, dim3(64 * NW_), lds_db, st, comps, G, dbases, rowptr, e_dst, e_rel, \
                       e_val, tiles_db, (int)n_nodes, R, B, d, ts_db, last, gstride, abl);                                                  \
  }
#define FBN_DB2(NK_, KL_, VE_) { if (pair) FBN_DB3(NK_, KL_, VE_, 2, 8) else if (two) FBN_DB3(NK_, KL_, VE_, 2, 16) else FBN_DB3(NK_, KL_, VE_, 1, 16) }
#define FBN_DB1(NK_, KL_) { if (vec) FBN_DB2(NK_, KL_, true) else FBN_DB2(NK_, KL_, false) }
#define FBN_DB0(NK_) { if (kld_db == 2) FBN_DB1(NK_, 2) else FBN_DB1(NK_, 4) }
      if (s.dpb == 4) FBN_DB0(1) else if (s.dpb == 8) FBN_DB0(2) else if (s.dpb == 12) FBN_DB0(3) else FBN_DB0(4)
#undef FBN_DB0
#undef FBN_DB1
#undef FBN_DB2
#undef FBN_DB3
    }
    if (dcomps) {
      HIP_TRY(zero_async(dcomps, (size_t)R * B * sizeof(float), st));
      FBN_BWD1(fbn_dcomps_kernel, s.lds_dc, bases, G, dcomps, rowptr, e_dst, e_rel, e_val, n_tiles, (int)n_nodes, R, B, d, s.ts_b, last, gstride, abl)
    }
#undef FBN_BWD1
#undef FBN_BWD2
  } else {
  if (dbases)
    FBT_BWD1(fbt_dbases_kernel, s.lds_db, comps, G, dbases, rowptr, e_dst, e_rel, e_val, n_tiles, (int)n_nodes, R, B, d, s.ts_b, last, gstride, abl)
  if (dcomps) {
    HIP_TRY(zero_async(dcomps, (size_t)R * B * sizeof(float), st));
    FBT_BWD1(fbt_dcomps_kernel, s.lds_dc, bases, G, dcomps, rowptr, e_dst, e_rel, e_val, n_tiles, (int)n_nodes, R, B, d, s.ts_b, last, gstride, abl)
  }
  }
#undef FBT_BWD1
#undef FBT_BWD2
#undef FBT_BWD3
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}
