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
#include <string.h>

#include <algorithm>

#include "rgcn_device.h"

namespace {

constexpr int BW_SCR = 16 * 20;      // floats of transposition scratch per wave (row stride 20: conflict-free b128 writes)

// (Round 2's staging kernel described in the header -- bwd_fused_d16_kernel, rgcn_bwd_fused_f32: four wave-owned tiles per workgroup, the dW
// partials parked in LDS staging slots and reduced across the waves at workgroup barriers -- was the fallback of rounds 3-4 for wave-owned
// tiles of 65..160 rows and RGCN_BWD_KERNEL=stage; nothing selected it by default and round 5 removed it: such plans take the two-pass
// backward.  Its measurements stay in profiles/r02_*; the byte ledger below is what led from it to the kernels that follow.)

// ---- round 3: the same walk with everything tile-local ON the CU and no workgroup barrier in the loop.
//
// Byte ledger of round 2's staging kernel at S1 (profiles/r02_pmc_kernels.json: 4.18 GB of fabric traffic per launch for 1.64 GB
// of algorithmic bytes, the kernel sits at the ~6.3 TB/s fabric ceiling): G gathers 2.69 GB + packed slots 0.24 GB are
// inherent; the rest is (a) the tile's X rows re-fetched through the fabric (the random gathers evict them from L1 / L2
// between a wave's chunks) and (b) the dW flush -- 3,906 workgroups x 101 relations x 1 KiB of fp32 atomics = 394 MB.
// Here:
//   (a) the tile's X rows are copied to LDS once (4 KiB per wave, 64 MB per launch, coalesced) and the K-over-messages A
//       operand of the dW MFMA is read straight from that copy: lane (k, m) reads X_lds[dl(4 t + k)][m] with ds_read_b32
//       (one row = 16 consecutive banks).  The destination row of slot 4 t + k reaches every lane of DPP row k through
//       three row rotations (row k rotated by k) and one row_share per MFMA step -- no second global load, no second trip
//       through the transposition scratch.
//   (b) a workgroup is NW = 16 (or 8) waves = 16 adjacent tiles, and the dW partials of a relation are summed across ALL
//       of them before they leave the CU: 977 x 101 KiB = 98 MB instead of 394 MB.  With the X copy there is no room for
//       NW x D staging slots (16 waves x (4 + 4 + 1) KiB = 144 KiB), so the waves share ONE window of slots (1 KiB per relation)
//       and serialise on a slot group with a state word in LDS (group << 16 | dirty slots << 8 | contributions << 1 | lock):
//         wait until the slot group serves my relation group and is free; take it (ds_cmpst); v = (dirty ? slot : 0) + my partial;
//         not the last wave: slot = v, state = (g, c + 1); last wave: state = (g + NG, 0), then add v to dW.
//       Every wave passes every relation group in order (an empty one contributes nothing but its count), so the slowest wave
//       never waits and nobody runs more than NG groups ahead: a sliding window instead of 26 workgroup barriers.
//       ORDERED (RGCN_DETERMINISTIC=1): waves take the slot group in wave order (no lock), the partial sums go to scratch with
//       plain stores and the two reduce kernels sum them in a fixed order -- bit-reproducible.
//       (The first form of this kernel -- one slot and one lock per relation, chunk after chunk -- is in the history of round 3;
//       its measurements are in profiles/r03_bwd_ablation.txt as "win1".)
// RELU: dX is the gradient before the ReLU that produced X (X = relu(pre)): rows are masked with X > 0 in the epilogue
// from the LDS copy (the caller's aten.threshold_backward launch, functional.py, disappears).
// ABL (timing experiments only, tools/kbench.py; results are wrong): 1 gathers hit the tile's own rows (no random HBM access),
// 2 no dW part, 4 no window protocol (partials dropped), 8 no fold / dX tile update, 16 no compute at all (loads only)
// ---- window kernel, second form: the U = 4 chunks of an iteration go through each phase TOGETHER, and the window is taken per
// GROUP of 4 relations.  Measured on the first form (profiles/r03_bwd_ablation.txt): with the random gathers replaced by
// tile-local rows it runs as long as with them (0.69 ms) -- the kernel is bound by a wave's chain of dependent latencies (4
// waves per SIMD): per chunk 4 dependent MFMAs, the fold, an LDS read-modify-write, the LDS transposition, 4 more MFMAs, and
// one lock / read-modify-write / unlock of the window per relation (0.13 ms by itself, 101 critical sections per tile).  Here
//   * the dX MFMAs of the four chunks are four independent chains, the four folds run in step (shared wave-uniform exits),
//     the transposition writes / reads of the four chunks are issued back to back (the LDS pipeline is in order) and waited
//     for once, each chunk's dW product starts from a zero accumulator (four independent chains) and is added to the held
//     partial of its relation afterwards;
//   * a wave holds the finished partials of a group of 4 consecutive relations in 16 registers and enters the window once per
//     group: 26 critical sections per tile, each moving up to four 1 KiB slots; state = group << 16 | dirty (4 bits) << 8 |
//     contributions << 1 | lock; NG groups in the window.
constexpr int WIN_GS = 4;       // relations per window group


// (The kernel of this second form, bwd_win2_d16_kernel, was measured at 0.69 ms and is superseded by the lean form below, which
// keeps its phases, its window protocol and its LDS layout; it lives in the history of round 3.)

// ---- window kernel, third form ("lean"): the same algorithm with the per-chunk instruction count cut to what the data flow needs.
// What bounds the second form (profiles/r03_bwd_ablation.txt, r03_pmc_sq.json): with 4 waves per SIMD the kernel's time follows
// the SIMD's issue cycles -- 8 MFMAs (32 cycles each) + ~90 VALU + ~70 SALU + 15 LDS instructions per chunk; removing the fold
// and the tile update (25 VALU) takes 13 % off, removing the window hand-over (53 SALU, 20 VALU) 17 %, removing the random
// gathers nothing.  So the chunk's bookkeeping moves into a one-off reformatting of the plan (rgcn_bwd_lean_prepare_f32):
//   slot (12 bytes)   word 0 = source row << 6 | same destination as the previous slot (bit 0) | last slot of its destination
//                     (bit 1): the gather offset is ONE v_and_or; word 1 = val; word 2 = destination row inside the tile << 6 (pads: 0):
//                     the LDS addresses of the tile update and of the X rows are ONE v_and_or / v_or each
//   chunk header      relation | some slot repeats its predecessor's destination (bit 16) | some run is 3 slots or longer (bit 17):
//                     the fold's wave-uniform exits come from a scalar register; the flags replace the DPP compare chains
//   slots, headers and W fragments are addressed as scalar base (advanced per chunk on the scalar unit) + a constant lane offset.
// The dX tile is no longer swizzled (the update address is base | row << 6 | k << 4; measured in round 2: the swizzle moved the LDS
// conflict counter, not the time).
struct LeanSlot { unsigned w0; float val; unsigned w2; };

// p_pack / chunk_rel (the packed transposed plan) -> lean slots + chunk headers; one lane per slot, one wave per 4 chunks
// p_pack == nullptr: the plan's unpacked arrays (tiles taller than 255 rows have no packed slots): source row, GLOBAL destination row
// (< 0: pad) and val per slot
__global__ __launch_bounds__(WG) void bwd_lean_prep_kernel(const int2 *__restrict__ p_pack, const int *__restrict__ p_src,
                                                           const int *__restrict__ p_dst, const float *__restrict__ p_val, int tile_rows,
                                                           const int *__restrict__ chunk_rel,
                                                           LeanSlot *__restrict__ slots, int *__restrict__ hdr, long long n_chunks) {
  const long long e = (long long)blockIdx.x * WG + threadIdx.x;          // slot index
  const long long c = e >> 4;
  const bool in = c < n_chunks;
  int2 pk = make_int2(0, 0);
  int dl = 0xFFFF;
  if (in) {
    if (p_pack) {
      pk = p_pack[e];
      dl = (int)((unsigned)pk.x >> 24);
      if (dl == 0xFF) dl = 0xFFFF;
    } else {
      const int gd = p_dst[e];
      pk = make_int2(p_src[e] & 0xFFFFFF, __builtin_bit_cast(int, p_val[e]));
      dl = gd < 0 ? 0xFFFF : gd % tile_rows;
    }
  }
  const bool pad = dl == 0xFFFF;
  const int key = pad ? -1 : dl;
  const int prev1 = dpp_i<ROW_SHR + 1>(-2, key), prev2 = dpp_i<ROW_SHR + 2>(-2, key), next1 = dpp_i<ROW_SHL + 1>(-3, key);
  const bool dup1 = !pad && prev1 == key, dup2 = !pad && prev2 == key, tail = !pad && next1 != key;
  const unsigned long long b1 = __builtin_amdgcn_ballot_w64(dup1), b2 = __builtin_amdgcn_ballot_w64(dup2);
  if (in) {
    LeanSlot o;
    o.w0 = ((unsigned)(pk.x & 0xFFFFFF) << 6) | (dup1 ? 1u : 0u) | (tail ? 2u : 0u);
    o.val = pad ? 0.f : __builtin_bit_cast(float, pk.y);
    o.w2 = pad ? 0u : ((unsigned)dl << 6);
    slots[e] = o;
    if ((threadIdx.x & 15) == 0) {
      const int row = (threadIdx.x & 63) >> 4;                           // this chunk's 16 lanes inside the wave
      const unsigned m1 = (unsigned)(b1 >> (16 * row)) & 0xFFFFu, m2 = (unsigned)(b2 >> (16 * row)) & 0xFFFFu;
      hdr[c] = chunk_rel[c] | (m1 ? 1 << 16 : 0) | (m2 ? 1 << 17 : 0);
    }
  }
}

template <int NW, int NG, bool ATOMIC, bool RELU, int ABL = 0>
__global__ __launch_bounds__(64 * NW, 4) void bwd_lean_d16_kernel(
    const float *__restrict__ G, const float *__restrict__ X, const float *__restrict__ Wtp, float *__restrict__ dX,
    float *__restrict__ dWout, const LeanSlot *__restrict__ slots, const int *__restrict__ hdr,
    const int *__restrict__ run_ptr, int n_tiles, int n_blocks, int tile_rows, int n_dst, int R) {
  constexpr int U = 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int t = blockIdx.x * NW + wave;
  const bool valid = t < n_tiles;
  const int nwv = min(NW, n_tiles - (int)blockIdx.x * NW);       // waves of this workgroup that own a tile
  float *tile = lds + wave * tile_rows * 16;                     // dX tile (row-major)
  float *xt = lds + (NW + wave) * tile_rows * 16;                // X tile (row-major)
  float *xs = lds + 2 * NW * tile_rows * 16 + wave * BW_SCR2;    // transposition scratch
  float *win = lds + 2 * NW * tile_rows * 16 + NW * BW_SCR2;     // [NG][WIN_GS][256] fragment order
  int *state = reinterpret_cast<int *>(win + NG * WIN_GS * 256); // [NG]
  const int row0 = t * tile_rows;
  const int nrows = valid ? min(tile_rows, n_dst - row0) : 0;
  for (int i = lane; i < nrows * 4; i += 64) {
    reinterpret_cast<float4 *>(tile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    reinterpret_cast<float4 *>(xt)[i] = reinterpret_cast<const float4 *>(X + (size_t)row0 * 16)[i];
  }
  if (tid < NG) state[tid] = tid << 16;                           // slot group q serves relation group q first
  __syncthreads();                                                // the only workgroup barrier
  if (!valid) return;

  const int my0 = __builtin_amdgcn_readfirstlane(run_ptr[(size_t)t * (R + 1)]);
  const int my1 = __builtin_amdgcn_readfirstlane(run_ptr[(size_t)t * (R + 1) + R]);
  const int m = lane & 15, k = lane >> 4;
  const int n_groups = (R + WIN_GS - 1) / WIN_GS;
  int doneg = 0;                // relation groups [0, doneg) have been contributed by this wave
  int curg = -1;                // group whose partials are held (-1: none)
  int hasmask = 0;              // which of the group's relations have data in hold[]
  f32x4 hold[WIN_GS];
#pragma unroll
  for (int q = 0; q < WIN_GS; ++q) hold[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned kofs = (unsigned)k << 4;
  const unsigned tile_k = (unsigned)((tile - lds) * 4) + kofs;       // LDS byte address of tile[0][4k]
  const unsigned xrd = (unsigned)((xt - lds) * 4) + (unsigned)m * 4;   // LDS byte address of X_lds[0][m]
  float *xs_wr = xs + m * 16 + 4 * ((k + (m >> 1)) & 3);                                  // this lane's float4 of slot m
  const float *xs_rd0 = xs + k * 16 + 4 * (((m >> 2) + (k >> 1)) & 3) + (m & 3);          // feature m of slot 4 t + k, t even: + 128 t
  const float *xs_rd1 = xs + (4 + k) * 16 + 4 * (((m >> 2) + 2 + (k >> 1)) & 3) + (m & 3); // t odd: + 128 (t - 1)
  const unsigned slot_lane = (unsigned)m * 12u;                       // this lane's slot inside a chunk (bytes)
  const unsigned w_lane = (unsigned)lane * 16u;                       // this lane's float4 inside a W fragment (bytes)

  // hand this wave's partials of relation group g to the shared window (mask: which relations carry data)
  auto contribute = [&](int g, const f32x4 (&part)[WIN_GS], int mask) {
    const int gq = (int)((unsigned)g % (unsigned)NG);
    int *st = state + gq;
    f32x4 *slot = reinterpret_cast<f32x4 *>(win + gq * (WIN_GS * 256)) + lane;
    int s;
    if (ATOMIC) {
      for (;;) {
        s = __builtin_amdgcn_readfirstlane(__hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if ((s >> 16) == g && !(s & 1)) {
          int ok = 0;
          if (lane == 0) {
            int e = s;
            ok = __hip_atomic_compare_exchange_strong(st, &e, s | 1, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          if (__builtin_amdgcn_readfirstlane(ok)) break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    } else {
      for (;;) {      // fixed order: wave 0, 1, ... (bit-reproducible sums)
        s = __builtin_amdgcn_readfirstlane(__hip_atomic_load(st, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
        if ((s >> 16) == g && ((s >> 1) & 31) == wave) break;
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int c = (s >> 1) & 31;
    const int dirty = (s >> 8) & 15;                             // slots somebody added data to
    const bool last = c + 1 == nwv;
    const int touch = last ? (dirty | mask) : mask;              // slots this wave reads / writes
    f32x4 v[WIN_GS];
#pragma unroll
    for (int q = 0; q < WIN_GS; ++q) {
      v[q] = part[q];
      if ((touch & dirty) >> q & 1) v[q] = slot[q * 64];
    }
#pragma unroll
    for (int q = 0; q < WIN_GS; ++q)
      if (((touch & dirty) >> q & 1) && (mask >> q & 1)) v[q] += part[q];
    if (!last) {
#pragma unroll
      for (int q = 0; q < WIN_GS; ++q)
        if (mask >> q & 1) slot[q * 64] = v[q];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0)
        __hip_atomic_store(st, (g << 16) | ((dirty | mask) << 8) | ((c + 1) << 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_store(st, (g + NG) << 16, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
      for (int q = 0; q < WIN_GS; ++q) {
        const int r = g * WIN_GS + q;
        if (r >= R) break;
        if (ATOMIC) {
          if (touch >> q & 1) {  // D: lane 16q+j holds rows 4q..4q+3 (input feature), column j (output feature)
            float *wr = dWout + (size_t)r * 256 + (4 * k) * 16 + m;
            atomicAdd(wr, v[q][0]); atomicAdd(wr + 16, v[q][1]); atomicAdd(wr + 32, v[q][2]); atomicAdd(wr + 48, v[q][3]);
          }
        } else {
          reinterpret_cast<f32x4 *>(dWout + ((size_t)r * n_blocks + blockIdx.x) * 256)[lane] = (touch >> q & 1) ? v[q] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
  };
  auto flush_group = [&]() {
    if (ABL & 4) return;
    const f32x4 zero[WIN_GS] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (; doneg < curg; ++doneg) contribute(doneg, zero, 0);
    contribute(curg, hold, hasmask);
    doneg = curg + 1;
  };

  if (my0 < my1) {
    const int last = my1 - 1;
    const char *sl_base = reinterpret_cast<const char *>(slots);
    LeanSlot sl_n[U];
    int hd_n[U];
    auto request_idx = [&](int c) {
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int cc = min(c + j, last);                          // scalar: chunks past the range re-read the last chunk (val forced to 0)
        sl_n[j] = *reinterpret_cast<const LeanSlot *>(sl_base + (size_t)cc * (RGCN_CHUNK * 12) + slot_lane);
        hd_n[j] = hdr[cc];
      }
    };
    request_idx(my0);
    for (int c = my0; c < my1; c += U) {
      unsigned w0_[U], w2_[U];
      float v_[U];
      int hd_[U];
      float4 g_[U], w_[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        w0_[j] = sl_n[j].w0;
        w2_[j] = sl_n[j].w2;
        v_[j] = (c + j <= last) ? sl_n[j].val : 0.f;
        hd_[j] = __builtin_amdgcn_readfirstlane(hd_n[j]);
      }
#pragma unroll
      for (int j = 0; j < U; ++j) asm volatile("" : "+v"(w0_[j]), "+v"(v_[j]));   // pin the index data here
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const unsigned og = (ABL & 1) ? ((((unsigned)row0 << 6) + (w2_[j])) | kofs) : ((w0_[j] & ~63u) | kofs);
        g_[j] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(G) + og);
        w_[j] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(Wtp) + (size_t)(hd_[j] & 0xFFFF) * 1024 + w_lane);
      }
      __builtin_amdgcn_sched_barrier(0);
      request_idx(c + U);
      __builtin_amdgcn_sched_barrier(0);
      if (ABL & 16) {
#pragma unroll
        for (int j = 0; j < U; ++j) asm volatile("" :: "v"(g_[j].x), "v"(g_[j].y), "v"(g_[j].z), "v"(g_[j].w), "v"(w_[j].x), "v"(w_[j].w), "v"(w2_[j]), "v"(v_[j]));
        continue;
      }
      // ---- phase 1: scaled rows, dX products (four independent MFMA chains)
      f32x4 sc[U], acc[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        sc[j] = f32x4{g_[j].x * v_[j], g_[j].y * v_[j], g_[j].z * v_[j], g_[j].w * v_[j]};
        acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < U; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_[j].x, sc[j][0], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < U; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_[j].y, sc[j][1], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < U; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_[j].z, sc[j][2], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < U; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_[j].w, sc[j][3], acc[j], 0, 0, 0);
      // ---- phase 2: dW products, two chunks at a time
      if (!(ABL & 2)) {
        f32x4 aw[U];
#pragma unroll
        for (int h = 0; h < U; h += 2) {
          float bv[2][4], av[2][4];
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const int j = h + jj;
            int rot = (int)w2_[j];                                  // lane (k, m) <- (row << 6) of slot (m + k) & 15
            rot = __builtin_amdgcn_update_dpp(rot, rot, 0x120 + 15, 0x2, 0xF, false);
            rot = __builtin_amdgcn_update_dpp(rot, rot, 0x120 + 14, 0x4, 0xF, false);
            rot = __builtin_amdgcn_update_dpp(rot, rot, 0x120 + 13, 0x8, 0xF, false);
            int rowk[4];                                            // row_share: slot 4 t4 + k
            rowk[0] = __builtin_amdgcn_update_dpp(0, rot, 0x150 + 0, 0xF, 0xF, false);
            rowk[1] = __builtin_amdgcn_update_dpp(0, rot, 0x150 + 4, 0xF, 0xF, false);
            rowk[2] = __builtin_amdgcn_update_dpp(0, rot, 0x150 + 8, 0xF, 0xF, false);
            rowk[3] = __builtin_amdgcn_update_dpp(0, rot, 0x150 + 12, 0xF, 0xF, false);
            asm volatile("" ::: "memory");
            *reinterpret_cast<f32x4 *>(xs_wr) = sc[j];
            asm volatile("" ::: "memory");
            bv[jj][0] = xs_rd0[0]; bv[jj][1] = xs_rd1[0]; bv[jj][2] = xs_rd0[128]; bv[jj][3] = xs_rd1[128];
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4)
              av[jj][t4] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(lds) + (xrd + (unsigned)rowk[t4]));
            asm volatile("" ::: "memory");
          }
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) aw[h + jj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
              aw[h + jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[jj][t4], bv[jj][t4], aw[h + jj], 0, 0, 0);
        }
        // relation bookkeeping (wave-uniform): add each chunk's product to the held partial of its relation
#pragma unroll
        for (int j = 0; j < U; ++j) {
          if (c + j > last) break;
          const int rj = hd_[j] & 0xFFFF;
          const int gj = rj >> 2, qj = rj & 3;
          if (gj != curg) {
            if (curg >= 0) {
              flush_group();
#pragma unroll
              for (int q = 0; q < WIN_GS; ++q) hold[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            curg = gj;
            hasmask = 0;
          }
          hasmask |= 1 << qj;
          if (qj == 0) hold[0] += aw[j];
          else if (qj == 1) hold[1] += aw[j];
          else if (qj == 2) hold[2] += aw[j];
          else hold[3] += aw[j];
        }
      }
      // ---- phase 3: fold equal destinations (flags from the plan), one LDS update per segment, chunk after chunk
      if (ABL & 8) {
#pragma unroll
        for (int j = 0; j < U; ++j) asm volatile("" :: "v"(acc[j][0]), "v"(acc[j][1]), "v"(acc[j][2]), "v"(acc[j][3]));
      } else {
        const int any1 = (hd_[0] | hd_[1] | hd_[2] | hd_[3]) & (1 << 16), any2 = (hd_[0] | hd_[1] | hd_[2] | hd_[3]) & (1 << 17);
        if (any1) {
#pragma unroll
          for (int j = 0; j < U; ++j) {
            const float sf = (w0_[j] & 1u) ? 1.f : 0.f;
            acc[j][0] = fmaf(dpp_shr0<1>(acc[j][0]), sf, acc[j][0]);
            acc[j][1] = fmaf(dpp_shr0<1>(acc[j][1]), sf, acc[j][1]);
            acc[j][2] = fmaf(dpp_shr0<1>(acc[j][2]), sf, acc[j][2]);
            acc[j][3] = fmaf(dpp_shr0<1>(acc[j][3]), sf, acc[j][3]);
          }
          if (any2) {     // runs of 3 and more (rare): the general fold on the destination rows
#pragma unroll
            for (int j = 0; j < U; ++j) {
              const int dst = (v_[j] != 0.f || (w0_[j] & 3u)) ? (int)(w2_[j] >> 6) : -1;     // pads: no flag, val 0
              const bool s2 = dpp_i<ROW_SHR + 2>(-1, dst) == dst && dst >= 0;
              const bool s4 = dpp_i<ROW_SHR + 4>(-1, dst) == dst && dst >= 0;
              const bool s8 = dpp_i<ROW_SHR + 8>(-1, dst) == dst && dst >= 0;
              const float f2 = s2 ? 1.f : 0.f, f4 = s4 ? 1.f : 0.f, f8 = s8 ? 1.f : 0.f;
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[j][e] = fmaf(dpp_shr0<2>(acc[j][e]), f2, acc[j][e]);
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[j][e] = fmaf(dpp_shr0<4>(acc[j][e]), f4, acc[j][e]);
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[j][e] = fmaf(dpp_shr0<8>(acc[j][e]), f8, acc[j][e]);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
          if (w0_[j] & 2u) {
            f32x4 *p = reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(lds) + (tile_k + w2_[j]));
            *p += acc[j];
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (!(ABL & (2 | 4 | 16))) {
    if (curg >= 0) flush_group();
    const f32x4 zero[WIN_GS] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (; doneg < n_groups; ++doneg) contribute(doneg, zero, 0);
  } else {
#pragma unroll
    for (int q = 0; q < WIN_GS; ++q) asm volatile("" :: "v"(hold[q][0]), "v"(hold[q][1]), "v"(hold[q][2]), "v"(hold[q][3]));
  }

  float4 *o4 = reinterpret_cast<float4 *>(dX + (size_t)row0 * 16);
  for (int i = lane; i < nrows * 4; i += 64) {
    float4 a = reinterpret_cast<const float4 *>(tile)[i];
    if (RELU) {
      const float4 x = reinterpret_cast<const float4 *>(xt)[i];
      a.x = x.x > 0.f ? a.x : 0.f; a.y = x.y > 0.f ? a.y : 0.f; a.z = x.z > 0.f ? a.z : 0.f; a.w = x.w > 0.f ? a.w : 0.f;
    }
    o4[i] = a;
  }
}

// ---- fifth form ("block tile", the default on large graphs): the workgroup, not the wave, owns the destination tile -- its own
// translation unit, csrc/rgcn_bwd_blk.hip (round 4: the dX tile is accumulated in DOUBLES with ds_add_f64, the only LDS float
// atomic gfx950 runs at full rate).

// ---- fourth form (measured, not kept): producers and consumers -- 12 waves of a workgroup own a tile each (gather, dX) and post
// their dW operands to LDS rings, 4 consumer waves own the relations (r mod 4) and accumulate a relation's dW in registers: no merge
// of partial sums, no locks.  Correct, 1.20 ms at S1: a consumer wave's chain of dependent LDS reads + MFMAs bounds the CU.

// ---- sparse (tile, relation) buckets (AM: 267 relations): backward of the two-pass path with ONE relation-major walk.
// Round 1: pass 1 of the feature gradient gathers G[s] per message (relation-major chunks, dense), and the weight gradient
// walks the same relation-major plan again gathering G[dst] AND X[src] -- three random row reads per message.  Here one
// wave per work item (<= 64 chunks of ONE relation) gathers G[s] and X[o] once each and produces both the transformed rows
// Y[slot] = val G[s] W_r^T (summed per destination by pass 2, rgcn_segment_gather_sum_f32) and the item's share of dW_r,
// which stays in 4 accumulator registers for the whole item (runs are long in relation-major order): one flush of 256
// atomics per item, no staging, no barriers.
// RELU: the feature gradient leaves masked with X > 0 (X = relu(...) of the producing layer, whose backward then skips its own masking launch):
// the lane that writes features 4k .. 4k+3 of a slot's transformed row holds the same features of X[p_dst] already.
template <int U, bool RELU>
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
      if (RELU) {
        acc[0] = x[j].x > 0.f ? acc[0] : 0.f; acc[1] = x[j].y > 0.f ? acc[1] : 0.f;
        acc[2] = x[j].z > 0.f ? acc[2] : 0.f; acc[3] = x[j].w > 0.f ? acc[3] : 0.f;
      }
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

// both fragment orders of W in one launch (forward: Wp[r][lane = 16k+o][c] = W[r][4k+c][o]; backward: Wtp[r][lane = 16k+f][c] = W[r][f][4k+c])
__global__ __launch_bounds__(WG) void pack_w16_pair_kernel(const float *__restrict__ W, float *__restrict__ Wp, float *__restrict__ Wtp, int n) {
  const int i = blockIdx.x * WG + threadIdx.x;
  if (i >= n) return;
  const int c = i & 3, o = (i >> 2) & 15, kk = (i >> 6) & 3, r = i >> 8;
  Wp[i] = W[r * 256 + (4 * kk + c) * 16 + o];
  Wtp[i] = W[r * 256 + o * 16 + (4 * kk + c)];
}

template <int NW, int NG>
size_t bwd_lean_lds(int tile_rows) {
  return ((size_t)2 * NW * tile_rows * 16 + NW * BW_SCR2 + NG * WIN_GS * 256) * sizeof(float) + 4 * NG * sizeof(int);
}
struct LeanLaunch {
  const float *G, *X, *Wtp;
  float *dX, *dWout;
  const LeanSlot *slots;
  const int *hdr, *run_ptr;
  int n_tiles, n_blocks, tile_rows, n_dst, R;
  size_t lds;
  hipStream_t st;
};
template <int NW, int NG, bool AT, bool RELU, int ABL = 0>
hipError_t launch_bwd_lean(const LeanLaunch &a) {
  auto kern = bwd_lean_d16_kernel<NW, NG, AT, RELU, ABL>;
  static bool raised = false;                    // once per process and instantiation (not a stream operation)
  if (a.lds > 64 * 1024 && !raised) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    raised = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)a.n_blocks), dim3(64 * NW), a.lds, a.st, a.G, a.X, a.Wtp, a.dX, a.dWout, a.slots, a.hdr,
                     a.run_ptr, a.n_tiles, a.n_blocks, a.tile_rows, a.n_dst, a.R);
  return hipGetLastError();
}
template <int NW, int NG, bool AT>
hipError_t launch_bwd_lean_f(const LeanLaunch &a, bool relu) {
#ifdef RGCN_ABLATIONS
  const int ABL = rgcn_option_value(RGCN_OPT_BWD_ABL);     // timing experiments (wrong results): ablation build only
  if (NW == 16 && AT && ABL) {
    switch (ABL) {
      case 1: return launch_bwd_lean<16, 3, true, false, 1>(a);
      case 2: return launch_bwd_lean<16, 3, true, false, 2>(a);
      case 4: return launch_bwd_lean<16, 3, true, false, 4>(a);
      case 5: return launch_bwd_lean<16, 3, true, false, 5>(a);
      case 8: return launch_bwd_lean<16, 3, true, false, 8>(a);
      case 16: return launch_bwd_lean<16, 3, true, false, 16>(a);
      default: break;
    }
  }
#endif
  return relu ? launch_bwd_lean<NW, NG, AT, true>(a) : launch_bwd_lean<NW, NG, AT, false>(a);
}

}  // namespace

extern "C" int rgcn_pack_w16t_f32(const float *W, float *Wp, int32_t R, void *stream) {
  if (!W || !Wp || R <= 0) { rgcn_set_error("pack_w16t: bad argument"); return RGCN_EINVAL; }
  const int n = R * 256;
  hipLaunchKernelGGL(pack_w16t_kernel, dim3((unsigned)((n + WG - 1) / WG)), dim3(WG), 0, (hipStream_t)stream, W, Wp, n);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_pack_w16_pair_f32(const float *W, float *Wp, float *Wtp, int32_t R, void *stream) {
  if (!W || !Wp || !Wtp || R <= 0) { rgcn_set_error("pack_w16_pair: bad argument"); return RGCN_EINVAL; }
  const int n = R * 256;
  hipLaunchKernelGGL(pack_w16_pair_kernel, dim3((unsigned)((n + WG - 1) / WG)), dim3(WG), 0, (hipStream_t)stream, W, Wp, Wtp, n);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int64_t rgcn_bwd_fused_scratch_floats(int64_t n_tiles, int32_t R) {
  const int64_t n_blocks = (n_tiles + 3) / 4;
  const int64_t S = std::max<int64_t>(1, std::min<int64_t>(16, n_blocks / 64));
  return (n_blocks * R + (int64_t)R * S) * 256;
}

extern "C" int64_t rgcn_bwd_lean_slot_bytes(int64_t n_chunks) { return n_chunks * RGCN_CHUNK * (int64_t)sizeof(LeanSlot); }

extern "C" int rgcn_bwd_lean_prepare_f32(const int32_t *p_pack, const int32_t *chunk_rel, int64_t n_chunks, void *slots, int32_t *hdr,
                                         void *stream) {
  if (n_chunks < 0 || (n_chunks && (!p_pack || !chunk_rel || !slots || !hdr))) { rgcn_set_error("bwd_lean_prepare: bad argument"); return RGCN_EINVAL; }
  if (!n_chunks) return RGCN_OK;
  const long long n = n_chunks * RGCN_CHUNK;
  hipLaunchKernelGGL(bwd_lean_prep_kernel, dim3((unsigned)((n + WG - 1) / WG)), dim3(WG), 0, (hipStream_t)stream,
                     reinterpret_cast<const int2 *>(p_pack), nullptr, nullptr, nullptr, 0, chunk_rel, reinterpret_cast<LeanSlot *>(slots), hdr,
                     (long long)n_chunks);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_bwd_lean_prepare_unpacked_f32(const int32_t *p_src, const int32_t *p_dst, const float *p_val, int32_t tile_rows,
                                                  const int32_t *chunk_rel, int64_t n_chunks, void *slots, int32_t *hdr, void *stream) {
  if (n_chunks < 0 || tile_rows <= 0 || tile_rows > 512 || (n_chunks && (!p_src || !p_dst || !p_val || !chunk_rel || !slots || !hdr))) {
    rgcn_set_error("bwd_lean_prepare_unpacked: bad argument");
    return RGCN_EINVAL;
  }
  if (!n_chunks) return RGCN_OK;
  const long long n = n_chunks * RGCN_CHUNK;
  hipLaunchKernelGGL(bwd_lean_prep_kernel, dim3((unsigned)((n + WG - 1) / WG)), dim3(WG), 0, (hipStream_t)stream,
                     nullptr, p_src, p_dst, p_val, tile_rows, chunk_rel, reinterpret_cast<LeanSlot *>(slots), hdr, (long long)n_chunks);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_bwd_lean_supported(int32_t tile_rows) {
  return bwd_lean_lds<16, 3>(tile_rows) <= 160 * 1024 || bwd_lean_lds<8, 1>(tile_rows) <= 160 * 1024;
}

extern "C" int rgcn_bwd_lean_f32(const float *G, const float *X, const float *Wt_packed, float *dX, float *dW, float *scratch,
                                 const void *slots, const int32_t *hdr, const int32_t *run_ptr, int64_t n_tiles, int32_t tile_rows,
                                 int64_t n_dst, int32_t R, int32_t flags, void *stream) {
  if (!G || !X || !Wt_packed || !dX || !dW || !slots || !hdr || !run_ptr || n_tiles <= 0 || tile_rows <= 0 || n_dst <= 0 || R <= 0 ||
      R > 0xFFFF) {
    rgcn_set_error("bwd_lean: bad argument");
    return RGCN_EINVAL;
  }
  const bool atomic = (flags & RGCN_F_DW_ATOMIC) != 0, relu = (flags & RGCN_F_RELU) != 0;
  if (!atomic && !scratch) { rgcn_set_error("bwd_lean: the deterministic reduction needs a scratch buffer"); return RGCN_EINVAL; }
  const int WIN_NW = rgcn_option_value(RGCN_OPT_BWD_NW);
  const bool nw16 = WIN_NW >= 16 && bwd_lean_lds<16, 3>(tile_rows) <= 160 * 1024;
  const bool nw8 = !nw16 && bwd_lean_lds<8, 1>(tile_rows) <= 160 * 1024;
  if (!nw16 && !nw8) { rgcn_set_error("bwd_lean: tile_rows = %d does not fit the LDS of a CU", tile_rows); return RGCN_EUNSUPPORTED; }
  const int NWv = nw16 ? 16 : 8;
  const int n_blocks = (int)((n_tiles + NWv - 1) / NWv);
  hipStream_t st = (hipStream_t)stream;
  const LeanLaunch L{G, X, Wt_packed, dX, atomic ? dW : scratch, reinterpret_cast<const LeanSlot *>(slots), hdr, run_ptr, (int)n_tiles, n_blocks,
                     tile_rows, (int)n_dst, R, nw16 ? bwd_lean_lds<16, 3>(tile_rows) : bwd_lean_lds<8, 1>(tile_rows), st};
  if (atomic) {
    HIP_TRY(zero_async(dW, (size_t)R * 256 * sizeof(float), st));
    if (nw16) HIP_TRY((launch_bwd_lean_f<16, 3, true>(L, relu)));
    else HIP_TRY((launch_bwd_lean_f<8, 1, true>(L, relu)));
  } else {
    if (nw16) HIP_TRY((launch_bwd_lean_f<16, 3, false>(L, relu)));
    else HIP_TRY((launch_bwd_lean_f<8, 1, false>(L, relu)));
    const int S = (int)std::max<int64_t>(1, std::min<int64_t>(16, n_blocks / 64));
    float *tmp = scratch + (size_t)n_blocks * R * 256;
    hipLaunchKernelGGL(dw_reduce_a_kernel, dim3((unsigned)R, (unsigned)S), dim3(WG), 0, st, scratch, tmp, n_blocks, S);
    hipLaunchKernelGGL(dw_reduce_b_kernel, dim3((unsigned)R), dim3(WG), 0, st, tmp, dW, S);
    HIP_TRY(hipGetLastError());
  }
  return RGCN_OK;
}

extern "C" int rgcn_bwd_scatter_dw_f32(const float *G, const float *X, const float *Wt_packed, float *Y, float *dW,
                                       const int32_t *p_src, const int32_t *p_dst, const float *p_val,
                                       const int32_t *chunk_rel, const int32_t *items, int64_t n_items, int32_t R, int32_t d,
                                       int32_t flags, void *stream) {
  if (!G || !X || !Wt_packed || !Y || !dW || R <= 0 || n_items < 0 || (n_items && (!p_src || !p_dst || !p_val || !chunk_rel || !items))) {
    rgcn_set_error("bwd_scatter_dw: bad argument");
    return RGCN_EINVAL;
  }
  if (d != 16) { rgcn_set_error("bwd_scatter_dw: only d = 16"); return RGCN_EUNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(zero_async(dW, (size_t)R * 256 * sizeof(float), st));
  if (!n_items) return RGCN_OK;
  const dim3 grid((unsigned)((n_items + WG / 64 - 1) / (WG / 64)));
  if (flags & RGCN_F_RELU)
    hipLaunchKernelGGL((bwd_scatter_dw_d16_kernel<4, true>), grid, dim3(WG), 0, st, G, X, Wt_packed, Y, dW, p_src, p_dst, p_val, chunk_rel,
                       reinterpret_cast<const int2 *>(items), (int)n_items);
  else
    hipLaunchKernelGGL((bwd_scatter_dw_d16_kernel<4, false>), grid, dim3(WG), 0, st, G, X, Wt_packed, Y, dW, p_src, p_dst, p_val, chunk_rel,
                       reinterpret_cast<const int2 *>(items), (int)n_items);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}
