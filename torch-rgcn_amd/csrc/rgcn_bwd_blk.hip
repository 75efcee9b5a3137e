// Block-tile backward of the hidden-16 relational layer (default on large graphs): the WORKGROUP owns the destination tile.
//
// Autograd duals of reference torch_rgcn/layers.py:293-301, one random row gather per message:
//     dX[o]  += val * G[s] W_r^T            dW_r += val * X[o]^T G[s]            for every message (s <- o, r, val)
//
// One tile of object rows per workgroup (16 waves, persistent: one workgroup per CU striding over the tiles); the tile's chunks
// (16 slots of ONE relation) are dealt to the waves four at a time from an LDS counter.  Shared by the waves, all in LDS:
//   X tile   [rows][16] fp32       the tile's input rows (A operand of the dW products)
//   dX tile  [rows][16] fp64       the tile's feature gradient
//   dW       [R][256]   fp32       every relation's weight gradient for the workgroup's whole life (MFMA fragment order; flushed ONCE,
//                                  dirty relations only: 26 MB of global atomics per launch at S1)
// Round 3 kept the dX tile in fp32 and added to it with 64-bit compare-and-swap loops, lane (k, m) = features 4k..4k+3 of slot m:
// 16 DIFFERENT rows inside every group of 16 lanes.  gfx950's LDS takes an atomic at full rate only when the 16 lanes of a quarter
// wave do not collide on banks (tools/micro/lds_cas_patterns.hip, profiles/r04_lds_cas_patterns.txt: any pattern whose quarter
// waves touch 64 CONTIGUOUS bytes runs in 9 cycles whatever the four quarters' rows are; rows that differ inside a quarter: one
// lane per cycle) -- that update cost 128 of the kernel's ~170 LDS cycles per chunk (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
// 0.63) and kept it at 0.59 ms.  Round 4:
//   * the dX product is computed the other way round, D[slot][o] = sum_f (val G[s_slot][f]) Wt_r[f][o] (the MFMA operands swapped):
//     lane (k, m) now holds output feature m of slots 4k .. 4k+3, so the four updates of a chunk are "column" instructions -- the
//     16 lanes of a quarter wave add to the 16 features of ONE row;
//   * the tile is kept in DOUBLES and updated with ds_add_f64, the one LDS float atomic that is native on gfx950 (9 cycles per
//     wave instruction; ds_add_f32: 193): no read, no compare-and-swap, no retry, no wait for a returned value, no fold of equal
//     destinations (slots of one row simply add twice), pads add 0.0 to a real row.  36 LDS cycles per chunk instead of 136, and
//     the sums are exact to fp64 before the one rounding to fp32 in the epilogue;
//   * per chunk ONE 176-byte record (rgcn_bwd_blk_prepare_f32, made once per plan): 16 x {source row << 6, val}, the 16 tile
//     rows (x 64) as 16-bit numbers (lane (k, m) reads the four of its quarter k with one 8-byte load), the relation.
// LDS: rows x 192 bytes + 16 KiB transposition scratch + R KiB (R / 4 KiB with RGCN_F_DIAG4): S1 (R = 101) -> tiles of up to
// 227 rows (218 used: 18 tiles per CU); AM with 4 x 4 diagonal blocks (R = 267) -> up to 406 rows.
//
// Per chunk, lane = 16 k + m:
//   gather       lane (k, m): G[s_m][4k..4k+3]                     (one 16-byte load, the only random HBM access)
//   dX           4 x v_mfma_f32_16x16x4_f32 (A = scaled rows, B = Wt_r fragment), 4 x ds_add_f64
//   dW           the scaled rows go through the wave's 1 KiB LDS scratch into K-over-messages layout (K index (k, t) = slot 4k + t);
//                A = X rows of those slots (ds_read_b32 from the X tile), 4 x v_mfma_f32_16x16x4_f32; consecutive chunks of one
//                relation accumulate in registers, a relation change adds the partial to the LDS table (lds_cas_add4 on the
//                lane-contiguous fragment: conflict-free)
// Between two tiles two barriers bracket the epilogue (dX rows -> fp32, ReLU mask of the consumer, store; re-zero; install the
// next tile's X rows, requested before the first barrier together with the wave's next records and one float4 of G for the bias
// gradient, which is summed on the side).  Hub tiles arrive in pieces (work units), their rows are added in memory.
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "rgcn_device.h"

namespace {

constexpr int BLK_NW = 16;
// chunks per loop trip of the forward kernel (rgcn_spmm_blk_f32).  FEWER is faster on the soft-window plan (S1, bench.py's per-launch average,
// one box: 6 chunks 0.343 ms, 4: 0.325, 3: 0.325, 2: 0.317, 1: 0.314): what sets the gather's time is the span of sources the chip reads at one
// time, and the chunks a wave has in flight widen it; AM's layer-2 forward (destination order, sparse buckets) does not care (0.292 / 0.293 / 0.295)
#ifndef RGCN_BLK_FWD_U
#define RGCN_BLK_FWD_U 1
#endif
constexpr int BLK_FWD_U = RGCN_BLK_FWD_U;
#ifndef RGCN_BLK_FWD_NW
#define RGCN_BLK_FWD_NW 16
#endif
constexpr int BLK_FWD_NW = RGCN_BLK_FWD_NW;    // waves per workgroup of the forward kernel (16, 8 or 4).  S1, waves x chunks per trip: 16 x 1 0.316 ms,
                                               // 8 x 2 0.364, 8 x 1 0.499, 4 x 4 0.477, 4 x 2 0.623: sixteen chunks in flight per CU, on as many waves as fit
constexpr int BLK_FWD_TQS = 16 / BLK_FWD_NW;   // a thread's share of the tile at the hand-over grows accordingly
// chunks per loop trip of the block-tile backward (rgcn_bwd_blk_f32).  Rounds 3-5 ran 4 -- on the 128-VGPR cliff, 6 registers spilled outside the
// loop; round 6 measured 3 (S1 on this kernel 0.483 -> 0.461 ms per launch; AM-shaped block-diagonal layer, DIAG4 on 509-row tiles, 0.515 -> 0.504;
// AIFB 0.032, MUTAG 0.037 -> 0.035) and 2 (AM 0.520).  (A first measurement of 3 showed 0.435 for AM: the pair loop of the dW part indexed a fourth
// chunk that was not there and the compiler dropped the work -- the AM tenth-scale parity test (tests/test_gpu_configs.py) caught it.)
#ifndef RGCN_BLK_BWD_U
#define RGCN_BLK_BWD_U 3
#endif
constexpr int BLK_BWD_U = RGCN_BLK_BWD_U;
constexpr int BLK_REC = 176;          // bytes per chunk record: 16 x {source row << 6, val} | 16 x u16 tile row | relation | 12 spare
constexpr int BLK_REC_ROWS = 128;
constexpr int BLK_REC_HDR = 160;
constexpr int BLK_LDS_MAX = 160 * 1024;

// Timing experiments live in a SEPARATE library (make abl -> librgcn_hip_abl.so, -DRGCN_ABLATIONS; tools/kbench.py loads it through
// RGCN_HIP_LIB): the shipped library has no switch that changes results.  ABL bits (wrong results): 2 no dW part, 8 no dX tile
// update, 16 loads only; the instrumented kernels also add up, per wave, the 100 MHz ticks spent waiting at the tile-end barrier
// and in the epilogue (rgcn_blk_debug_read).
#ifdef RGCN_ABLATIONS
constexpr int BLK_DBG_WAVES = 256 * BLK_NW;
__device__ unsigned long long rgcn_blk_dbg[4 * BLK_DBG_WAVES];     // per wave (no same-address atomics: they would cost 0.1 ms)
#define BLK_ABL_PARAM , int ABL
#define BLK_ABL_ARG(a) , a
#define BLK_T() ((long long)__builtin_amdgcn_s_memtime())
#define BLK_DBG(stmt) stmt
#else
#define BLK_ABL_PARAM
#define BLK_ABL_ARG(a)
constexpr int ABL = 0;
#define BLK_T() 0ll
#define BLK_DBG(stmt)
#endif

size_t bwd_blk_lds(int R, bool diag4, int rows) {
  return (size_t)rows * 192 + (size_t)BLK_NW * BW_SCR2 * 4 + (size_t)R * (diag4 ? 64 : 256) * 4 + (4 + (size_t)R) * 4;
}

// the transposed plan (packed 8-byte slots, or the unpacked arrays when p_pack == nullptr: tiles taller than 255 rows) -> chunk
// records; one lane per slot.  Pads (val 0) copy the source and the row of their chunk's first slot: what they add is 0.0 * a row
// some real message of the bucket reads anyway.
__global__ __launch_bounds__(WG) void bwd_blk_prep_kernel(const int2 *__restrict__ p_pack, const int *__restrict__ p_src,
                                                          const int *__restrict__ p_dst, const float *__restrict__ p_val, int tile_rows,
                                                          const int *__restrict__ chunk_rel, char *__restrict__ rec, long long n_chunks) {
  const long long e = (long long)blockIdx.x * WG + threadIdx.x;          // slot index
  const long long c = e >> 4;
  const int s = (int)(e & 15);
  const bool in = c < n_chunks;
  int src = 0, dl = -1;
  float val = 0.f;
  if (in) {
    if (p_pack) {
      const int2 pk = p_pack[e];
      src = pk.x & 0xFFFFFF;
      dl = (int)((unsigned)pk.x >> 24);
      if (dl == 0xFF) dl = -1;
      val = __builtin_bit_cast(float, pk.y);
    } else {
      const int gd = p_dst[e];
      src = p_src[e];
      dl = gd < 0 ? -1 : gd % tile_rows;
      val = p_val[e];
    }
  }
  const int lane0 = (threadIdx.x & 63) & ~15;
  const int src0 = __shfl(src, lane0), dl0 = __shfl(dl, lane0);
  if (dl < 0) { val = 0.f; src = dl0 < 0 ? 0 : src0; dl = dl0 < 0 ? 0 : dl0; }
  if (in) {
    char *r = rec + (size_t)c * BLK_REC;
    *reinterpret_cast<uint2 *>(r + s * 8) = uint2{(unsigned)src << 6, __builtin_bit_cast(unsigned, val)};
    *reinterpret_cast<unsigned short *>(r + BLK_REC_ROWS + s * 2) = (unsigned short)(dl << 6);     // tile row x 64: the LDS offset of its X row
    if (s == 0) *reinterpret_cast<int4 *>(r + BLK_REC_HDR) = int4{chunk_rel[c], 0, 0, 0};
  }
}

// Register budget: 1024 threads per workgroup = 128 VGPRs per lane, and the loop below sits exactly on it.  ANY value that has to
// be reloaded from scratch inside the loop is fatal here, not just slow: scratch reloads are VMEM operations, they return in
// order behind the record prefetch of the next quad, so the wave stalls for a fabric round trip in the middle of its compute
// phase (measured: 0.52 -> 0.60 .. 0.65 ms with 4 .. 6 reloads per quad).  Tried on that cliff in round 4 and NOT kept
// (profiles/r04_blk_ablation.txt): software-pipelined pairs with the gathers of two pairs in flight across tile switches (no spills,
// but 66 VALU + 54 SALU per chunk against 42 + 27: 0.548 against 0.527 ms); gathers of the next tile's first quad issued before
// the tile-end barrier and a second copy of the loop body for it (70 .. 100 dwords of scratch); the same as one-dword "touch" loads
// (20 dwords of scratch); the tile update before the dW products (the dX MFMA chains then have nothing to overlap with).
// DIAG4: W is block-diagonal with 4 x 4 blocks (decomposition {type: block}, width 16): only the four diagonal blocks of dW_r are
// wanted (64 floats per relation instead of 256: hundreds of relations fit, AM has 267)
// TQ: float4 of X a thread carries from one tile to the next = ceil(tile_rows / 256)
template <bool RELU, bool DIAG4, int TQ BLK_ABL_PARAM>
__global__ __launch_bounds__(64 * BLK_NW) void bwd_blk_d16_kernel(
    const float *__restrict__ G, const float *__restrict__ X, const float *__restrict__ Wtp, float *__restrict__ dX,
    float *__restrict__ dWout, const char *__restrict__ rec, const int *__restrict__ run_ptr, int n_tiles, int tile_rows, int n_dst,
    int R, float *__restrict__ dbias, int n_src,
    const int4 *__restrict__ units, int n_units) {      // units: {tile, first chunk, end chunk, flags}: a hub tile arrives in pieces (RGCN_U_SHARED:
                                                        // their dX rows are ADDED to a zeroed dX); NULL: one unit per tile (n_units = n_tiles)
  constexpr int U = BLK_BWD_U, NW = BLK_NW, NT = 64 * BLK_NW;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const unsigned xt_off = (unsigned)tile_rows * 128u;             // bytes: the dX tile [rows][16] doubles comes first
  const unsigned xs_off = (unsigned)tile_rows * 192u;
  float4 *dxz = reinterpret_cast<float4 *>(lds);                  // the dX tile as 16-byte units (zeroing)
  const double2 *dxd2 = reinterpret_cast<const double2 *>(lds);
  float4 *xt4 = reinterpret_cast<float4 *>(lds + xt_off);         // X tile [rows][16] floats
  float *xs = reinterpret_cast<float *>(lds + xs_off) + wave * BW_SCR2;          // transposition scratch of this wave
  constexpr int DWR = DIAG4 ? 64 : 256;                           // floats of dW kept per relation
  float *dwl = reinterpret_cast<float *>(lds + xs_off) + NW * BW_SCR2;           // dW [R][64 lanes][4] (fragment order, a lane's four elements adjacent); DIAG4: [R][16 lanes][4]
  int *ctl = reinterpret_cast<int *>(dwl + (size_t)R * DWR);      // [0]: next quad of the tile; [4 + r]: relation r has data
  int *dirty = ctl + 4;

  auto unit_of = [&](int u) {
    if (units) return units[u];
    return int4{u, run_ptr[(size_t)u * (R + 1)], run_ptr[(size_t)u * (R + 1) + R], 0};
  };
  int un = blockIdx.x;
  int4 unit = unit_of(un);
  int t = __builtin_amdgcn_readfirstlane(unit.x);
  int row0 = t * tile_rows;
  int nrows = min(tile_rows, n_dst - row0);
  int c0 = __builtin_amdgcn_readfirstlane(unit.y);
  int c1 = __builtin_amdgcn_readfirstlane(unit.z);
  int shared = __builtin_amdgcn_readfirstlane(unit.w) & RGCN_U_SHARED;
  int nq = (c1 - c0 + U - 1) / U;
  // dealing: a wave's first ns quads are consecutive (wave * ns ...: chunks of one relation that straddle quads stay in its registers,
  // fewer adds to the shared dW table), the last quarter of the tile is dealt from the LDS counter (balance)
  auto static_quads = [](int n) { return max(1, min(4, (3 * n) / (4 * BLK_NW))); };      // (4 of ~5.2 quads per wave fixed: measured slower, 0.54 against 0.525 ms)
  int ns = static_quads(nq);
  // bias gradient (column sums of G) on the side: every tile switch a thread adds one float4 of G's rows, the workgroups striding
  // through G together (S1: 18 stripes of 16 KiB per workgroup = its 18 tiles); what is left after the last tile is read at the end
  float4 gs = make_float4(0.f, 0.f, 0.f, 0.f);
  const long long g_n4 = dbias ? (long long)n_src * 4 : 0, g_step = (long long)gridDim.x * NT;
  long long g_i = (long long)blockIdx.x * NT + tid;
  {
#pragma unroll
    for (int q = 0; q < TQ; ++q) {
      const int idx = tid + q * NT;
      if (idx < tile_rows * 4) {
        float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < nrows * 4) x0 = reinterpret_cast<const float4 *>(X + (size_t)row0 * 16)[idx];
        xt4[idx] = x0;
      }
    }
    for (int i = tid; i < tile_rows * 8; i += NT) dxz[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < R * (DWR / 4); i += NT) reinterpret_cast<float4 *>(dwl)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < R + 4; i += NT) ctl[i] = i == 0 ? NW * (ns + 1) : 0;
  }
  __syncthreads();

  const int m = lane & 15, k = lane >> 4;
  const unsigned kofs = (unsigned)k << 4;
  const unsigned dx_lane = (unsigned)m * 8u;                            // LDS byte address of dX tile [0][m]
  const unsigned xrd = xt_off + (unsigned)m * 4u;                       // LDS byte address of X tile [0][m]
  // scratch: the float4 column (features 4c .. 4c+3) of slot s is stored at column (c + (s >> 2)) & 3 -- b128 writes of 8
  // consecutive slots and the b32 reads of one slot's 16 features by a quarter wave are conflict-free
  float *xs_wr = xs + m * 16 + 4 * ((k + (m >> 2)) & 3);                // this lane's float4: features 4k .. 4k+3 of slot m
  const float *xs_rd = xs + (4 * k) * 16 + 4 * (((m >> 2) + k) & 3) + (m & 3);   // feature m of slot 4k + t: + 16 t
  const unsigned slot_lane = (unsigned)m * 8u;
  const unsigned rows_lane = (unsigned)BLK_REC_ROWS + (unsigned)k * 8u;
  const unsigned w_lane = (unsigned)lane * 16u;

  f32x4 hold = f32x4{0.f, 0.f, 0.f, 0.f};    // dW partial of relation cur_r (this wave's consecutive chunks)
  int cur_r = -1;
  auto flush_hold = [&]() {
    if (cur_r >= 0 && (ABL & 4)) {           // a plain store instead of the compare-and-swap add (racy: timing only)
      *reinterpret_cast<f32x4 *>(dwl + (size_t)cur_r * 256 + lane * 4) = hold;
    } else if (cur_r >= 0) {
      if (DIAG4) {       // D: lane 16 k + m holds rows 4k .. 4k + 3, column m: the diagonal block k lives in the lanes with m >> 2 == k
        if ((m >> 2) == k) lds_cas_add4(dwl + (size_t)cur_r * 64 + (4 * k + (m & 3)) * 4, hold);
      } else {
        lds_cas_add4(dwl + (size_t)cur_r * 256 + lane * 4, hold);
      }
      if (lane == 0) lds_st(dirty + cur_r, 1);
    }
  };

  uint2 sl_n[U], rw_n[U];
  int hd_n[U];
  auto request_idx = [&](int c, int last) {
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int cc = min(c + j, last);                          // scalar: chunks past the tile re-read its last chunk (val forced to 0)
      const char *r = rec + (size_t)cc * BLK_REC;
      sl_n[j] = *reinterpret_cast<const uint2 *>(r + slot_lane);
      rw_n[j] = *reinterpret_cast<const uint2 *>(r + rows_lane);
      hd_n[j] = *reinterpret_cast<const int *>(r + BLK_REC_HDR);
    }
  };
  // the wave's p-th quad of a tile: p < ns: wave * ns + p; p == ns: NW * ns + wave (fixed as well: no counter round trip); then the counter
  int q_cur = wave * ns, q_nxt = 1 < ns ? wave * ns + 1 : NW * ns + wave, q_pos = 2;
  if (q_cur < nq) request_idx(c0 + q_cur * U, c1 - 1);
  BLK_DBG(const long long dbg_t0 = BLK_T(); long long dbg_wait = 0; long long dbg_epi = 0; long long dbg_ta = 0; long long dbg_tb = 0;)

  for (;;) {
    const int last = c1 - 1;
    while (q_cur < nq) {
      const int c = c0 + q_cur * U;
      unsigned w0_[U];
      uint2 rw_[U];
      float v_[U];
      int hd_[U];
      float4 g_[U], w_[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        w0_[j] = sl_n[j].x;
        rw_[j] = rw_n[j];
        v_[j] = (c + j <= last) ? __builtin_bit_cast(float, sl_n[j].y) : 0.f;
        hd_[j] = __builtin_amdgcn_readfirstlane(hd_n[j]);
      }
#pragma unroll
      for (int j = 0; j < U; ++j) asm volatile("" : "+v"(w0_[j]), "+v"(v_[j]));   // pin the index data here
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const unsigned og = w0_[j] | kofs;
        g_[j] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(G) + og);
        w_[j] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(Wtp) + (size_t)hd_[j] * 1024 + w_lane);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (q_nxt < nq) request_idx(c0 + q_nxt * U, last);
      int q_nn = q_pos < ns ? wave * ns + q_pos : NW * ns + wave;
      if (q_pos > ns && lane == 0) q_nn = __hip_atomic_fetch_add(ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      ++q_pos;
      __builtin_amdgcn_sched_barrier(0);
      if (ABL & 16) {
#pragma unroll
        for (int j = 0; j < U; ++j) asm volatile("" :: "v"(g_[j].x), "v"(g_[j].y), "v"(g_[j].z), "v"(g_[j].w), "v"(w_[j].x), "v"(w_[j].w), "v"(rw_[j].x), "v"(rw_[j].y), "v"(v_[j]));
        q_cur = q_nxt;
        q_nxt = __builtin_amdgcn_readfirstlane(q_nn);
        continue;
      }
      // ---- phase 1: scaled rows, dX products (four independent MFMA chains): D[slot][o], lane (k, m) <- slots 4k .. 4k+3, feature m
      f32x4 sc[U], acc[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        sc[j] = f32x4{g_[j].x * v_[j], g_[j].y * v_[j], g_[j].z * v_[j], g_[j].w * v_[j]};
        acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < U; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[j][0], w_[j].x, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < U; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[j][1], w_[j].y, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < U; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[j][2], w_[j].z, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < U; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[j][3], w_[j].w, acc[j], 0, 0, 0);
      // tile rows of slots 4k .. 4k+3 (the records hold them x 64: the byte offset of the X row)
      unsigned ro[U][4];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        ro[j][0] = rw_[j].x & 0xFFFFu;
        ro[j][1] = rw_[j].x >> 16;
        ro[j][2] = rw_[j].y & 0xFFFFu;
        ro[j][3] = rw_[j].y >> 16;
      }
      // ---- phase 2: dW products, two chunks at a time
      if (!(ABL & 2)) {
        f32x4 aw[U];
#pragma unroll
        for (int h = 0; h < U; h += 2) {
          float bv[2][4], av[2][4];
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const int j = h + jj;
            if (j >= U) break;    // (odd U: the last pair is a single chunk)
            if (ABL & 32) {       // no LDS round trips: the operands are whatever the lane holds
#pragma unroll
              for (int t4 = 0; t4 < 4; ++t4) { bv[jj][t4] = sc[j][t4]; av[jj][t4] = __builtin_bit_cast(float, ro[j][t4]); }
            } else {
            asm volatile("" ::: "memory");
            *reinterpret_cast<f32x4 *>(xs_wr) = sc[j];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) bv[jj][t4] = xs_rd[16 * t4];
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) av[jj][t4] = *reinterpret_cast<const float *>(lds + (xrd + ro[j][t4]));
            asm volatile("" ::: "memory");
            }
          }
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
            if (h + jj < U) aw[h + jj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
              if (h + jj < U) aw[h + jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[jj][t4], bv[jj][t4], aw[h + jj], 0, 0, 0);
        }
        // relation bookkeeping (wave-uniform): consecutive chunks of one relation accumulate in registers
#pragma unroll
        for (int j = 0; j < U; ++j) {
          if (c + j > last) break;
          const int rj = hd_[j];
          if (rj != cur_r) {
            flush_hold();
            cur_r = rj;
            hold = aw[j];
          } else {
            hold += aw[j];
          }
        }
      }
      // ---- phase 3: the tile update, one ds_add_f64 per slot quarter: the 16 lanes of a quarter wave add to the 16 features of one row
      if (ABL & 8) {
#pragma unroll
        for (int j = 0; j < U; ++j) asm volatile("" :: "v"(acc[j][0]), "v"(acc[j][1]), "v"(acc[j][2]), "v"(acc[j][3]));
      } else {
#pragma unroll
        for (int j = 0; j < U; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            __hip_atomic_fetch_add(static_cast<double *>(__builtin_assume_aligned(lds + (dx_lane + 2u * ro[j][e]), 8)), (double)acc[j][e], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      __builtin_amdgcn_sched_barrier(0);
      q_cur = q_nxt;
      q_nxt = __builtin_amdgcn_readfirstlane(q_nn);
    }
    flush_hold();
    cur_r = -1;
    // the next tile of this workgroup: its X rows and this wave's first chunks are requested before the barrier
    const int unn = un + (int)gridDim.x;
    float4 xn[TQ];
#pragma unroll
    for (int q = 0; q < TQ; ++q) xn[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    int c0n = 0, c1n = 0, nqn = 0, row0n = 0, nrn = 0, tn = 0, sharedn = 0, nsn = 1;
    if (unn < n_units) {
      const int4 unx = unit_of(unn);
      tn = __builtin_amdgcn_readfirstlane(unx.x);
      row0n = tn * tile_rows;
      nrn = min(tile_rows, n_dst - row0n);
      c0n = __builtin_amdgcn_readfirstlane(unx.y);
      c1n = __builtin_amdgcn_readfirstlane(unx.z);
      sharedn = __builtin_amdgcn_readfirstlane(unx.w) & RGCN_U_SHARED;
      nqn = (c1n - c0n + U - 1) / U;
      nsn = static_quads(nqn);
#pragma unroll
      for (int q = 0; q < TQ; ++q)
        if (tid + q * NT < nrn * 4) xn[q] = reinterpret_cast<const float4 *>(X + (size_t)row0n * 16)[tid + q * NT];
      if (wave * nsn < nqn) request_idx(c0n + wave * nsn * U, c1n - 1);
    }
    float4 gn = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g_i < g_n4) gn = reinterpret_cast<const float4 *>(G)[g_i];
    g_i += g_step;
    BLK_DBG(dbg_ta = BLK_T();)
    lds_barrier();                                                 // every wave has finished adding to the dX tile (LDS only: the prefetches stay in flight)
    BLK_DBG(dbg_tb = BLK_T(); dbg_wait += dbg_tb - dbg_ta;)
    gs.x += gn.x; gs.y += gn.y; gs.z += gn.z; gs.w += gn.w;
    // The next tile's X rows (requested before the barrier) are waited for HERE, before the tile's rows are stored (round 5, from the ISA):
    // they are installed after the stores, vmcnt counts in order and the number of stores differs from thread to thread, so the compiler's
    // wait at the install was vmcnt(0) -- it also waited for the stores to COMPLETE: one write round trip per tile with nothing in flight.
    // With the arrival pinned here the stores drain under the re-zeroing, the barrier and the next tile's first gathers.
#pragma unroll
    for (int q = 0; q < TQ; ++q) asm volatile("" : "+v"(xn[q].x), "+v"(xn[q].y), "+v"(xn[q].z), "+v"(xn[q].w));
#pragma unroll
    for (int q = 0; q < TQ; ++q) {
      const int idx = tid + q * NT;
      if (idx < nrows * 4) {
        const double2 d0 = dxd2[2 * idx], d1 = dxd2[2 * idx + 1];
        float4 a = make_float4((float)d0.x, (float)d0.y, (float)d1.x, (float)d1.y);
        if (RELU) {
          const float4 x = xt4[idx];
          a.x = x.x > 0.f ? a.x : 0.f; a.y = x.y > 0.f ? a.y : 0.f; a.z = x.z > 0.f ? a.z : 0.f; a.w = x.w > 0.f ? a.w : 0.f;
        }
        float4 *o = reinterpret_cast<float4 *>(dX + (size_t)row0 * 16) + idx;
        if (shared) {       // a piece of a hub tile: the pieces' rows are summed in memory (dX was zeroed)
          atomicAdd(&o->x, a.x); atomicAdd(&o->y, a.y); atomicAdd(&o->z, a.z); atomicAdd(&o->w, a.w);
        } else {
          *o = a;
        }
      }
    }
    if (unn >= n_units) break;
#pragma unroll
    for (int q = 0; q < TQ; ++q) {
      const int idx = tid + q * NT;
      if (idx < tile_rows * 4) {
        dxz[2 * idx] = make_float4(0.f, 0.f, 0.f, 0.f);
        dxz[2 * idx + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
        xt4[idx] = xn[q];
      }
    }
    if (tid == 0) ctl[0] = NW * (nsn + 1);
    lds_barrier();                                                 // the next tile is installed (nobody waits for the dX stores)
    BLK_DBG(dbg_epi += BLK_T() - dbg_tb;)
    un = unn; t = tn; row0 = row0n; nrows = nrn; c0 = c0n; c1 = c1n; nq = nqn; shared = sharedn;
    ns = nsn;
    q_cur = wave * ns; q_nxt = 1 < ns ? wave * ns + 1 : NW * ns + wave; q_pos = 2;
  }
  BLK_DBG(if (lane == 0 && blockIdx.x < 256) { unsigned long long *d = rgcn_blk_dbg + 4 * (blockIdx.x * BLK_NW + wave); d[0] += (unsigned long long)dbg_wait;
                                               d[1] += (unsigned long long)dbg_epi; d[2] += (unsigned long long)(BLK_T() - dbg_t0); d[3] += 1ull; })
  if (dbias) {
    for (; g_i < g_n4; g_i += g_step) {
      const float4 gn = reinterpret_cast<const float4 *>(G)[g_i];
      gs.x += gn.x; gs.y += gn.y; gs.z += gn.z; gs.w += gn.w;
    }
    // thread tid holds features 4 (tid & 3) .. + 3: fold the 16 lanes of a wave that share (lane & 3), then the 16 waves through
    // the (now idle) transposition scratch
#pragma unroll
    for (int sft = 4; sft < 64; sft <<= 1) {
      gs.x += __shfl_xor(gs.x, sft); gs.y += __shfl_xor(gs.y, sft); gs.z += __shfl_xor(gs.z, sft); gs.w += __shfl_xor(gs.w, sft);
    }
    if (lane < 4) *reinterpret_cast<float4 *>(xs + 4 * lane) = gs;
    __syncthreads();
    if (tid < 16) {
      float a = 0.f;
      const float *all = reinterpret_cast<const float *>(lds + xs_off);
      for (int i = 0; i < NW; ++i) a += all[i * BW_SCR2 + tid];
      atomicAdd(dbias + tid, a);
    }
  }
  // one flush of the workgroup's dW: dirty relations only.  D fragment: lane 16k + m, element e = row 4k + e (input feature), column m
  for (int i = tid; i < R * DWR; i += NT) {
    const int r = i / DWR, e = i & 3;
    const int ln = DIAG4 ? 16 * ((i >> 4) & 3) + 4 * ((i >> 4) & 3) + ((i >> 2) & 3) : (i >> 2) & 63;     // DIAG4: k = (i >> 4) & 3, m = 4 k + ((i >> 2) & 3)
    if (lds_ld(dirty + r)) atomicAdd(dWout + (size_t)r * 256 + (4 * (ln >> 4) + e) * 16 + (ln & 15), dwl[i]);
  }
}


// ------------------------------------------------------------------ the same walk as a FORWARD kernel (round 5)
// out[dst] = bias + sum val X[src] W_r on the forward plan cut into tall tiles: the workgroup owns the destination tile [rows][16] in
// doubles (ds_add_f64), its chunks are dealt to the 16 waves as above, a chunk is one gather of 16 source rows, one 1 KiB weight fragment out
// of L2 and four MFMAs.  No X tile, no dW table: 128 bytes of LDS per row, tiles of up to 1023 rows (the records keep a slot's tile row
// x 64 in 16 bits).  For whom: hidden-16 layers whose (tile, relation) buckets on the wave-owned 32 .. 64-row tiles are mostly padding AND
// whose relations do not fit the one-pass CSR kernel's LDS (AM as shipped, layer 2: R = 267 -- 31 messages per bucket on 930-row tiles
// instead of ~2 on 64-row ones); until round 5 that layer's forward was two passes over a [M, 16] intermediate (0.40 + 0.22 ms).
// TQ = ceil(tile_rows / 256): float4 of the tile a thread converts and stores per tile.
template <bool RELU, int TQ>
__global__ __launch_bounds__(64 * BLK_FWD_NW) void spmm_blk_d16_kernel(
    const float *__restrict__ X, const float *__restrict__ Wp, const float *__restrict__ bias, float *__restrict__ out,
    const char *__restrict__ rec, const int *__restrict__ run_ptr, int n_tiles, int tile_rows, int n_dst, int R,
    const int4 *__restrict__ units, int n_units) {
  constexpr int U = BLK_FWD_U, NW = BLK_FWD_NW, NT = 64 * BLK_FWD_NW;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  float4 *tz = reinterpret_cast<float4 *>(lds);                   // the tile [rows][16] doubles as 16-byte units (zeroing)
  const double2 *td2 = reinterpret_cast<const double2 *>(lds);
  int *ctl = reinterpret_cast<int *>(lds + (size_t)tile_rows * 128u);      // [0]: next quad of the tile

  auto unit_of = [&](int u) {
    if (units) return units[u];
    return int4{u, run_ptr[(size_t)u * (R + 1)], run_ptr[(size_t)u * (R + 1) + R], 0};
  };
  auto static_quads = [](int n) { return max(1, min(4, (3 * n) / (4 * BLK_FWD_NW))); };
  int un = blockIdx.x;
  int4 unit = unit_of(un);
  int row0 = __builtin_amdgcn_readfirstlane(unit.x) * tile_rows;
  int nrows = min(tile_rows, n_dst - row0);
  int c0 = __builtin_amdgcn_readfirstlane(unit.y), c1 = __builtin_amdgcn_readfirstlane(unit.z);
  int uflags = __builtin_amdgcn_readfirstlane(unit.w);
  int nq = (c1 - c0 + U - 1) / U;
  int ns = static_quads(nq);
  for (int i = tid; i < tile_rows * 8; i += NT) tz[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid == 0) ctl[0] = NW * (ns + 1);
  __syncthreads();

  const int m = lane & 15, k = lane >> 4;
  const unsigned kofs = (unsigned)k << 4;
  const unsigned t_lane = (unsigned)m * 8u;                             // LDS byte address of tile [0][m]
  const unsigned slot_lane = (unsigned)m * 8u;
  const unsigned rows_lane = (unsigned)BLK_REC_ROWS + (unsigned)k * 8u;
  const unsigned w_lane = (unsigned)lane * 16u;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) bv = reinterpret_cast<const float4 *>(bias)[tid & 3];       // (a thread's float4 of a tile row: columns 4 (tid & 3) ..)

  uint2 sl_n[U], rw_n[U];
  int hd_n[U];
  auto request_idx = [&](int c, int last) {
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int cc = min(c + j, last);                          // scalar: chunks past the tile re-read its last chunk (val forced to 0)
      const char *r = rec + (size_t)cc * BLK_REC;
      sl_n[j] = *reinterpret_cast<const uint2 *>(r + slot_lane);
      rw_n[j] = *reinterpret_cast<const uint2 *>(r + rows_lane);
      hd_n[j] = *reinterpret_cast<const int *>(r + BLK_REC_HDR);
    }
  };
  int q_cur = wave * ns, q_nxt = 1 < ns ? wave * ns + 1 : NW * ns + wave, q_pos = 2;
  if (q_cur < nq) request_idx(c0 + q_cur * U, c1 - 1);

  for (;;) {
    const int last = c1 - 1;
    while (q_cur < nq) {
      const int c = c0 + q_cur * U;
      unsigned w0_[U];
      uint2 rw_[U];
      float v_[U];
      int hd_[U];
      float4 g_[U], w_[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        w0_[j] = sl_n[j].x;
        rw_[j] = rw_n[j];
        v_[j] = (c + j <= last) ? __builtin_bit_cast(float, sl_n[j].y) : 0.f;
        hd_[j] = __builtin_amdgcn_readfirstlane(hd_n[j]);
      }
#pragma unroll
      for (int j = 0; j < U; ++j) asm volatile("" : "+v"(w0_[j]), "+v"(v_[j]));   // pin the index data here
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const unsigned og = w0_[j] | kofs;
        g_[j] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(X) + og);
        w_[j] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(Wp) + (size_t)hd_[j] * 1024 + w_lane);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (q_nxt < nq) request_idx(c0 + q_nxt * U, last);
      int q_nn = q_pos < ns ? wave * ns + q_pos : NW * ns + wave;
      if (q_pos > ns && lane == 0) q_nn = __hip_atomic_fetch_add(ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      ++q_pos;
      __builtin_amdgcn_sched_barrier(0);
      // D[slot][o] = sum_f (val X[src_slot][f]) W_r[f][o]: lane (k, m) <- slots 4k .. 4k+3, output feature m
      f32x4 sc[U], acc[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        sc[j] = f32x4{g_[j].x * v_[j], g_[j].y * v_[j], g_[j].z * v_[j], g_[j].w * v_[j]};
        acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < U; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[j][0], w_[j].x, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < U; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[j][1], w_[j].y, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < U; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[j][2], w_[j].z, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < U; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[j][3], w_[j].w, acc[j], 0, 0, 0);
      // the tile update, one ds_add_f64 per slot quarter (the records hold a slot's tile row x 64; a tile row is 128 bytes of doubles)
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const unsigned ro[4] = {rw_[j].x & 0xFFFFu, rw_[j].x >> 16, rw_[j].y & 0xFFFFu, rw_[j].y >> 16};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          __hip_atomic_fetch_add(static_cast<double *>(__builtin_assume_aligned(lds + (t_lane + 2u * ro[e]), 8)), (double)acc[j][e], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      __builtin_amdgcn_sched_barrier(0);
      q_cur = q_nxt;
      q_nxt = __builtin_amdgcn_readfirstlane(q_nn);
    }
    // the next unit of this workgroup: this wave's first chunks are requested before the barrier
    const int unn = un + (int)gridDim.x;
    int c0n = 0, c1n = 0, nqn = 0, row0n = 0, nrn = 0, flagsn = 0, nsn = 1;
    if (unn < n_units) {
      const int4 unx = unit_of(unn);
      row0n = __builtin_amdgcn_readfirstlane(unx.x) * tile_rows;
      nrn = min(tile_rows, n_dst - row0n);
      c0n = __builtin_amdgcn_readfirstlane(unx.y);
      c1n = __builtin_amdgcn_readfirstlane(unx.z);
      flagsn = __builtin_amdgcn_readfirstlane(unx.w);
      nqn = (c1n - c0n + U - 1) / U;
      nsn = static_quads(nqn);
      if (wave * nsn < nqn) request_idx(c0n + wave * nsn * U, c1n - 1);
    }
    // (the records just requested are pinned as arrived BEFORE the tile's stores: vmcnt counts in order, and the number of stores differs
    // from thread to thread -- left to the compiler, their first use would wait for the stores to complete)
    lds_barrier();                                                 // every wave has finished adding to the tile
#pragma unroll
    for (int j = 0; j < U; ++j) asm volatile("" : "+v"(sl_n[j].x), "+v"(sl_n[j].y), "+v"(rw_n[j].x), "+v"(rw_n[j].y), "+v"(hd_n[j]));
    const bool shared = uflags & RGCN_U_SHARED;
    const bool with_bias = !shared || (uflags & RGCN_U_FIRST);
#pragma unroll
    for (int q = 0; q < TQ; ++q) {
      const int idx = tid + q * NT;
      if (idx < nrows * 4) {
        const double2 d0 = td2[2 * idx], d1 = td2[2 * idx + 1];
        float4 a = make_float4((float)d0.x, (float)d0.y, (float)d1.x, (float)d1.y);
        if (with_bias) { a.x += bv.x; a.y += bv.y; a.z += bv.z; a.w += bv.w; }
        float4 *o = reinterpret_cast<float4 *>(out + (size_t)row0 * 16) + idx;
        if (shared) {       // a piece of a hub tile: the pieces' rows are summed in memory (out was zeroed; no ReLU on such plans)
          atomicAdd(&o->x, a.x); atomicAdd(&o->y, a.y); atomicAdd(&o->z, a.z); atomicAdd(&o->w, a.w);
        } else {
          if (RELU) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
          *o = a;
        }
      }
    }
    if (unn >= n_units) break;
#pragma unroll
    for (int q = 0; q < TQ; ++q) {
      const int idx = tid + q * NT;
      if (idx < tile_rows * 4) {
        tz[2 * idx] = make_float4(0.f, 0.f, 0.f, 0.f);
        tz[2 * idx + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if (tid == 0) ctl[0] = NW * (nsn + 1);
    lds_barrier();                                                 // the tile is clean again (nobody waits for the stores)
    un = unn; row0 = row0n; nrows = nrn; c0 = c0n; c1 = c1n; nq = nqn; uflags = flagsn;
    ns = nsn;
    q_cur = wave * ns; q_nxt = 1 < ns ? wave * ns + 1 : NW * ns + wave; q_pos = 2;
  }
}


}  // namespace

extern "C" int64_t rgcn_bwd_blk_rec_bytes(int64_t n_chunks) { return n_chunks * (int64_t)BLK_REC; }

extern "C" int rgcn_bwd_blk_prepare_f32(const int32_t *p_pack, const int32_t *p_src, const int32_t *p_dst, const float *p_val,
                                        int32_t tile_rows, const int32_t *chunk_rel, int64_t n_chunks, void *rec, void *stream) {
  if (n_chunks < 0 || tile_rows <= 0 || tile_rows > 1023 || (!p_pack && n_chunks && (!p_src || !p_dst || !p_val)) ||
      (p_pack && tile_rows > 255) || (n_chunks && (!chunk_rel || !rec))) {
    rgcn_set_error("bwd_blk_prepare: bad argument");
    return RGCN_EINVAL;
  }
  if (!n_chunks) return RGCN_OK;
  const long long n = n_chunks * RGCN_CHUNK;
  hipLaunchKernelGGL(bwd_blk_prep_kernel, dim3((unsigned)((n + WG - 1) / WG)), dim3(WG), 0, (hipStream_t)stream,
                     reinterpret_cast<const int2 *>(p_pack), p_src, p_dst, p_val, tile_rows, chunk_rel, static_cast<char *>(rec),
                     (long long)n_chunks);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int32_t rgcn_bwd_blk_max_rows(int32_t R, int32_t flags) {
  if (R <= 0 || R >= 0xFFFF) return 0;
  const long long fixed = (long long)bwd_blk_lds(R, (flags & RGCN_F_DIAG4) != 0, 0);
  const long long rows = (BLK_LDS_MAX - fixed) / 192;
  return (int32_t)std::max<long long>(0, std::min<long long>(512, rows));
}

extern "C" int rgcn_bwd_blk_supported(int32_t tile_rows, int32_t R, int32_t flags) {
  return tile_rows > 0 && tile_rows <= rgcn_bwd_blk_max_rows(R, flags);
}

extern "C" int rgcn_bwd_blk_f32(const float *G, const float *X, const float *Wt_packed, float *dX, float *dW, const void *rec,
                                const int32_t *run_ptr, int64_t n_tiles, int32_t tile_rows, int64_t n_dst, int32_t R, int32_t flags,
                                float *dbias, int64_t n_src, const int32_t *units, int64_t n_units, int64_t n_split, void *stream) {
  if (!G || !X || !Wt_packed || !dX || !dW || !rec || !run_ptr || n_tiles <= 0 || tile_rows <= 0 || n_dst <= 0 || R <= 0 ||
      (units && ((n_units < n_tiles && !(flags & (RGCN_F_ACCUMULATE | RGCN_F_PARTIAL))) || n_split < 0 || n_units > INT32_MAX || n_units <= 0))) {
    rgcn_set_error("bwd_blk: bad argument");
    return RGCN_EINVAL;
  }
  if ((flags & (RGCN_F_ACCUMULATE | RGCN_F_PARTIAL)) && (!units || n_split)) { rgcn_set_error("bwd_blk: a slab (RGCN_F_PARTIAL / RGCN_F_ACCUMULATE) is an explicit list of whole-tile work units"); return RGCN_EINVAL; }
  if (!units) { n_units = n_tiles; n_split = 0; }
  if (!rgcn_bwd_blk_supported(tile_rows, R, flags)) {
    rgcn_set_error("bwd_blk: tile_rows = %d / R = %d: 192 bytes per row + 16 KiB + R KiB (R / 4 KiB with RGCN_F_DIAG4) of LDS do not fit (at most %d rows)",
                   tile_rows, R, rgcn_bwd_blk_max_rows(R, flags));
    return RGCN_EUNSUPPORTED;
  }
  const bool relu = (flags & RGCN_F_RELU) != 0, diag4 = (flags & RGCN_F_DIAG4) != 0;
  const int tq = tile_rows > 256 ? 2 : 1;
  const size_t lds = bwd_blk_lds(R, diag4, tile_rows);
  hipStream_t st = (hipStream_t)stream;
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0, v = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
    n_cu = v > 0 ? v : 256;
  }
  if (dbias && (n_src <= 0 || n_src >= (int64_t(1) << 29))) { rgcn_set_error("bwd_blk: dbias needs 0 < n_src < 2^29"); return RGCN_EINVAL; }
  if (flags & RGCN_F_ACCUMULATE) {
    // a later slab of the same backward (relation-sharded layers: the slab before is being all-reduced while this one runs): dW / dbias keep adding
  } else if (dbias == dW + (size_t)R * 256) {     // one fill for both when the caller laid them out back to back
    HIP_TRY(zero_async(dW, ((size_t)R * 256 + 16) * sizeof(float), st));
  } else {
    HIP_TRY(zero_async(dW, (size_t)R * 256 * sizeof(float), st));
    if (dbias) HIP_TRY(zero_async(dbias, 16 * sizeof(float), st));
  }
  if (n_split) HIP_TRY(zero_async(dX, (size_t)n_dst * 16 * sizeof(float), st));        // pieces of hub tiles add their rows
  const unsigned n_blocks = (unsigned)std::min<int64_t>(n_units, n_cu);
  auto launch = [&](auto kern, bool &raised) -> hipError_t {
    if (lds > 64 * 1024 && !raised) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, BLK_LDS_MAX);
      if (e != hipSuccess) return e;
      raised = true;
    }
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(64 * BLK_NW), lds, st, G, X, Wt_packed, dX, dW, static_cast<const char *>(rec), run_ptr,
                       (int)n_tiles, tile_rows, (int)n_dst, R, dbias, (int)n_src, reinterpret_cast<const int4 *>(units), (int)n_units);
    return hipGetLastError();
  };
#ifdef RGCN_ABLATIONS
  {
    const int ABLV = rgcn_option_value(RGCN_OPT_BWD_ABL);     // timing experiments (wrong results): this library only
    static bool a0 = false, a1 = false, a2 = false, a3 = false, a4 = false, a5 = false, a6 = false, a7 = false;
    if (tq != 1 || diag4) { rgcn_set_error("bwd_blk (ablation library): only dense weights on tiles of up to 256 rows"); return RGCN_EUNSUPPORTED; }
    if (ABLV == 2) HIP_TRY(launch(bwd_blk_d16_kernel<false, false, 1, 2>, a0));
    else if (ABLV == 8) HIP_TRY(launch(bwd_blk_d16_kernel<false, false, 1, 8>, a1));
    else if (ABLV == 10) HIP_TRY(launch(bwd_blk_d16_kernel<false, false, 1, 10>, a2));
    else if (ABLV == 16) HIP_TRY(launch(bwd_blk_d16_kernel<false, false, 1, 16>, a3));
    else if (ABLV == 4) HIP_TRY(launch(bwd_blk_d16_kernel<false, false, 1, 4>, a4));
    else if (ABLV == 32) HIP_TRY(launch(bwd_blk_d16_kernel<false, false, 1, 32>, a5));
    else if (ABLV == 36) HIP_TRY(launch(bwd_blk_d16_kernel<false, false, 1, 36>, a6));
    else HIP_TRY(launch(bwd_blk_d16_kernel<false, false, 1, 0>, a7));
    return RGCN_OK;
  }
#else
  static bool r0 = false, r1 = false, r2 = false, r3 = false, r4 = false, r5 = false, r6 = false, r7 = false;
  if (tq == 2 && diag4 && relu) HIP_TRY(launch(bwd_blk_d16_kernel<true, true, 2>, r0));
  else if (tq == 2 && diag4) HIP_TRY(launch(bwd_blk_d16_kernel<false, true, 2>, r1));
  else if (tq == 2 && relu) HIP_TRY(launch(bwd_blk_d16_kernel<true, false, 2>, r2));
  else if (tq == 2) HIP_TRY(launch(bwd_blk_d16_kernel<false, false, 2>, r3));
  else if (diag4 && relu) HIP_TRY(launch(bwd_blk_d16_kernel<true, true, 1>, r4));
  else if (diag4) HIP_TRY(launch(bwd_blk_d16_kernel<false, true, 1>, r5));
  else if (relu) HIP_TRY(launch(bwd_blk_d16_kernel<true, false, 1>, r6));
  else HIP_TRY(launch(bwd_blk_d16_kernel<false, false, 1>, r7));
#endif
  return RGCN_OK;
}

extern "C" int32_t rgcn_spmm_blk_max_rows(void) { return 1023; }

extern "C" int rgcn_spmm_blk_f32(const float *X, const float *W_packed, const float *bias, float *out, const void *rec,
                                 const int32_t *run_ptr, int64_t n_tiles, int32_t tile_rows, int64_t n_dst, int32_t R, int32_t flags,
                                 const int32_t *units, int64_t n_units, int64_t n_split, void *stream) {
  if (!X || !W_packed || !out || !rec || !run_ptr || n_tiles <= 0 || tile_rows <= 0 || n_dst <= 0 || R <= 0 || n_dst > INT32_MAX ||
      (units && (n_units < n_tiles || n_split < 0 || n_units > INT32_MAX))) {
    rgcn_set_error("spmm_blk: bad argument");
    return RGCN_EINVAL;
  }
  if (tile_rows > rgcn_spmm_blk_max_rows()) { rgcn_set_error("spmm_blk: tiles of at most %d rows (got %d)", rgcn_spmm_blk_max_rows(), tile_rows); return RGCN_EUNSUPPORTED; }
  if (!units) { n_units = n_tiles; n_split = 0; }
  const bool relu = (flags & RGCN_F_RELU) != 0;
  if (relu && n_split) { rgcn_set_error("spmm_blk: relu in the epilogue needs tiles that are not cut into shared pieces"); return RGCN_EUNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0, v = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
    n_cu = v > 0 ? v : 256;
  }
  if (n_split) HIP_TRY(zero_async(out, (size_t)n_dst * 16 * sizeof(float), st));          // pieces of hub tiles add their rows
  const size_t lds = (size_t)tile_rows * 128 + 64;
  const unsigned n_blocks = (unsigned)std::min<int64_t>(n_units, n_cu);
  auto launch = [&](auto kern, bool &raised) -> hipError_t {
    if (lds > 64 * 1024 && !raised) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, BLK_LDS_MAX);
      if (e != hipSuccess) return e;
      raised = true;
    }
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(64 * BLK_FWD_NW), lds, st, X, W_packed, bias, out, static_cast<const char *>(rec), run_ptr, (int)n_tiles,
                       tile_rows, (int)n_dst, R, reinterpret_cast<const int4 *>(units), (int)n_units);
    return hipGetLastError();
  };
  static bool r[8] = {false, false, false, false, false, false, false, false};
  const int tq = (tile_rows + 255) / 256;
  if (relu) {
    if (tq == 1) HIP_TRY(launch(spmm_blk_d16_kernel<true, 1 * BLK_FWD_TQS>, r[0]));
    else if (tq == 2) HIP_TRY(launch(spmm_blk_d16_kernel<true, 2 * BLK_FWD_TQS>, r[1]));
    else if (tq == 3) HIP_TRY(launch(spmm_blk_d16_kernel<true, 3 * BLK_FWD_TQS>, r[2]));
    else HIP_TRY(launch(spmm_blk_d16_kernel<true, 4 * BLK_FWD_TQS>, r[3]));
  } else {
    if (tq == 1) HIP_TRY(launch(spmm_blk_d16_kernel<false, 1 * BLK_FWD_TQS>, r[4]));
    else if (tq == 2) HIP_TRY(launch(spmm_blk_d16_kernel<false, 2 * BLK_FWD_TQS>, r[5]));
    else if (tq == 3) HIP_TRY(launch(spmm_blk_d16_kernel<false, 3 * BLK_FWD_TQS>, r[6]));
    else HIP_TRY(launch(spmm_blk_d16_kernel<false, 4 * BLK_FWD_TQS>, r[7]));
  }
  return RGCN_OK;
}

#ifdef RGCN_ABLATIONS
/* ablation library only: {ticks waiting at the tile-end barrier, ticks in the epilogue, ticks in the kernel, waves} summed over all
 * waves of all block-tile launches since the last reset (100 MHz ticks) */
extern "C" __attribute__((visibility("default"))) int rgcn_blk_debug_read(unsigned long long *out4, int reset) {
  static unsigned long long h[4 * BLK_DBG_WAVES];
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(rgcn_blk_dbg), sizeof(h)));
  for (int i = 0; i < 4; ++i) out4[i] = 0;
  for (int w = 0; w < BLK_DBG_WAVES; ++w)
    for (int i = 0; i < 4; ++i) out4[i] += h[4 * w + i];
  if (reset) { memset(h, 0, sizeof(h)); HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(rgcn_blk_dbg), h, sizeof(h))); }
  return RGCN_OK;
}
#endif
