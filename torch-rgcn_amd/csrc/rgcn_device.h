// Device-side helpers shared by the message-passing kernels (rgcn_kernels.hip, rgcn_bwd.hip): DPP primitives and the
// segmented fold over the 16 slots of a chunk.  gfx950 only, 64-wide wavefronts: lane = 16*k + m.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "rgcn_hip.h"
#include "rgcn_zero.h"

extern "C" void rgcn_set_error(const char *fmt, ...);

typedef float f32x4 __attribute__((ext_vector_type(4)));

#include "rgcn_options.h"


#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) {                                                             \
      rgcn_set_error("%s failed: %s", #expr, hipGetErrorString(e_));                    \
      return RGCN_EHIP;                                                                 \
    }                                                                                   \
  } while (0)

namespace {

constexpr int WG = 256;            // 4 wavefronts
constexpr int LDS_TILE_BYTES = 64 * 1024;

// ------------------------------------------------------------------ spmm
// One WAVE per destination tile (a workgroup = 4 independent waves = 4 tiles).  The wave is the
// only writer of its tile's rows, so the LDS accumulation is a plain read-modify-write
// (ds_read_b128 / ds_write_b128): LDS float atomics measured ~160 cycles per wave-instruction on
// gfx950 and were 80% of the first version of this kernel; the RMW form is free and deterministic.
//
// Per chunk of 16 messages (one relation):   D^T[o][slot] = sum_f W_rel[f][o] * (val * X[src_slot][f])
//   A operand = W fragment  (lane 16k+o, step c : W[f(c,k)][o])
//   B operand = gathered rows (lane 16k+m, step c : val_m * X[src_m][f(c,k)])
//   D         : lane 16q+m holds output features 4q..4q+3 of slot m  -> ONE 16-byte LDS update per lane
// Slots are sorted by destination, so messages that share a destination sit in adjacent lanes of a
// 16-lane DPP row: a 4-step segmented scan (v_*_dpp row_shr) folds them and only the last lane of
// each segment touches LDS -- no two lanes of one instruction ever update the same address.

template <int CTRL>
__device__ __forceinline__ float dpp_f(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                              CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xF, 0xF, false);
}
// The wave-owned LDS tile of the hidden-16 kernels is XOR-swizzled: float4 column q of row r is stored at column
// q ^ ((r >> 2) & 3).  With the plain layout (row stride 16 floats) rows r and r + 4 share their banks and the 8 lanes of a
// ds_read/write_b128 group that carry the same column collide 2-3 ways (round 1: SQ_LDS_BANK_CONFLICT = 70 % of
// SQ_LDS_IDX_ACTIVE in spmm_d16_kernel); swizzled, 16 consecutive rows cover all 64 banks.  i = float4 index (4 row + q).
__device__ __forceinline__ int tile_swz(int i) { return i ^ ((i >> 4) & 3); }

constexpr int ROW_SHR = 0x110;  // + n : lane m reads lane m-n of its 16-lane row
constexpr int ROW_SHL = 0x100;  // + n : lane m reads lane m+n

// In-place inclusive segmented sum over the slots (lanes m = 0..15 of a DPP row) of NV accumulators.
// Branch-free: lane m adds lane m-N's value times a 0/1 mask (same destination), N = 1, 2, 4, 8; the
// DPP shift is foldable into the multiply-add.  Returns true in the last lane of every run of equal `dst`.
template <int N>
__device__ __forceinline__ float dpp_shr0(float src) {   // lane m <- lane m-N of the 16-lane row, 0 when m < N
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), ROW_SHR + N, 0xF, 0xF, true));
}

template <int NV>
__device__ __forceinline__ bool fold_segments(f32x4 (&acc)[NV], int dst) {
#define RGCN_FOLD_STEP(N, SAME)                                                \
  {                                                                            \
    const float sf = (SAME) ? 1.f : 0.f;                                       \
    _Pragma("unroll") for (int v = 0; v < NV; ++v) {                           \
      acc[v][0] = fmaf(dpp_shr0<N>(acc[v][0]), sf, acc[v][0]);                 \
      acc[v][1] = fmaf(dpp_shr0<N>(acc[v][1]), sf, acc[v][1]);                 \
      acc[v][2] = fmaf(dpp_shr0<N>(acc[v][2]), sf, acc[v][2]);                 \
      acc[v][3] = fmaf(dpp_shr0<N>(acc[v][3]), sf, acc[v][3]);                 \
    }                                                                          \
  }
  // runs of equal destinations are contiguous (slots are sorted): a step of distance N is needed only if some
  // run is longer than N -- most chunks stop after step 1 or need no step at all (wave-uniform early exits)
  // pads carry dst = -1 and never join a run
  const bool s1 = dpp_i<ROW_SHR + 1>(-1, dst) == dst && dst >= 0;
  if (__builtin_amdgcn_ballot_w64(s1)) {
    RGCN_FOLD_STEP(1, s1)
    const bool s2 = dpp_i<ROW_SHR + 2>(-1, dst) == dst && dst >= 0;
    if (__builtin_amdgcn_ballot_w64(s2)) {
      RGCN_FOLD_STEP(2, s2)
      const bool s4 = dpp_i<ROW_SHR + 4>(-1, dst) == dst && dst >= 0;
      if (__builtin_amdgcn_ballot_w64(s4)) {
        RGCN_FOLD_STEP(4, s4)
        const bool s8 = dpp_i<ROW_SHR + 8>(-1, dst) == dst && dst >= 0;
        RGCN_FOLD_STEP(8, s8)
      }
    }
  }
#undef RGCN_FOLD_STEP
  return dpp_i<ROW_SHL + 1>(-2, dst) != dst && dst >= 0;   // last lane of a run of real slots
}


// The same fold for NC chunks at once (each with its own destinations): the chunks' steps are interleaved and share the
// wave-uniform early exits -- a step is taken by all chunks when any of them needs it (independent dependency chains for the
// scheduler instead of NC chains one after the other).  tail[c] = last lane of every run of real slots of chunk c.
template <int NC>
__device__ __forceinline__ void fold_segments_multi(f32x4 (&acc)[NC], const int (&dst)[NC], bool (&tail)[NC]) {
#define RGCN_FOLDM_STEP(N, SAME)                                                       \
  _Pragma("unroll") for (int c = 0; c < NC; ++c) {                                     \
    const float sf = (SAME)[c] ? 1.f : 0.f;                                            \
    acc[c][0] = fmaf(dpp_shr0<N>(acc[c][0]), sf, acc[c][0]);                           \
    acc[c][1] = fmaf(dpp_shr0<N>(acc[c][1]), sf, acc[c][1]);                           \
    acc[c][2] = fmaf(dpp_shr0<N>(acc[c][2]), sf, acc[c][2]);                           \
    acc[c][3] = fmaf(dpp_shr0<N>(acc[c][3]), sf, acc[c][3]);                           \
  }
  bool s1[NC], any = false;
#pragma unroll
  for (int c = 0; c < NC; ++c) { s1[c] = dpp_i<ROW_SHR + 1>(-1, dst[c]) == dst[c] && dst[c] >= 0; any |= s1[c]; }
  if (__builtin_amdgcn_ballot_w64(any)) {
    RGCN_FOLDM_STEP(1, s1)
    bool s2[NC]; any = false;
#pragma unroll
    for (int c = 0; c < NC; ++c) { s2[c] = dpp_i<ROW_SHR + 2>(-1, dst[c]) == dst[c] && dst[c] >= 0; any |= s2[c]; }
    if (__builtin_amdgcn_ballot_w64(any)) {
      RGCN_FOLDM_STEP(2, s2)
      bool s4[NC]; any = false;
#pragma unroll
      for (int c = 0; c < NC; ++c) { s4[c] = dpp_i<ROW_SHR + 4>(-1, dst[c]) == dst[c] && dst[c] >= 0; any |= s4[c]; }
      if (__builtin_amdgcn_ballot_w64(any)) {
        RGCN_FOLDM_STEP(4, s4)
        bool s8[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) s8[c] = dpp_i<ROW_SHR + 8>(-1, dst[c]) == dst[c] && dst[c] >= 0;
        RGCN_FOLDM_STEP(8, s8)
      }
    }
  }
#undef RGCN_FOLDM_STEP
#pragma unroll
  for (int c = 0; c < NC; ++c) tail[c] = dpp_i<ROW_SHL + 1>(-2, dst[c]) != dst[c] && dst[c] >= 0;
}


constexpr int BW_SCR2 = 16 * 16;  // floats of transposition scratch per wave: [16 slots][16 features], the float4 column of features 4k..4k+3 of
                                  // slot m is stored at column (k + (m >> 1)) & 3 (b128 writes of 8 consecutive slots and b32 reads of two
                                  // consecutive slots are conflict-free without the 4-float row padding of the first form: 1 KiB, not 1.25)

// LDS words other waves write: relaxed workgroup-scope atomics (ds_read_b32 / ds_write_b32).  NOT `volatile`: a volatile access through
// a generic pointer compiles to flat_load / flat_store sc0 sc1 and a wait for vmcnt(0) -- every poll then drains the wave's global loads.
__device__ __forceinline__ int lds_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt -- every wave would wait for its own outstanding
// global loads (prefetches) and for the acknowledgement of its stores at each barrier.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// p[0..3] += v, atomically against the other waves of the workgroup (p: LDS, 16-byte aligned).  NOT ds_add_f32: measured on gfx950
// (tools/micro/lds_atomic_rate.hip) one wave-level ds_add_f32 occupies the CU's LDS for ~190 cycles (3 per lane, serialised) against
// 4.4 for ds_add_u32 / ds_write_b32 -- float atomics in LDS are 40x slower than integer ones.  So: optimistic read, add in registers,
// two 64-bit compare-and-swaps (integer rate), each half retried on interference (rare: 16 waves on a 16 KiB tile).
__device__ __forceinline__ void lds_cas_add4(float *p, const f32x4 &v) {
  unsigned long long *q = reinterpret_cast<unsigned long long *>(p);
  const f32x4 o = *reinterpret_cast<const f32x4 *>(p);
  asm volatile("" ::: "memory");
  unsigned long long e0 = ((unsigned long long)__float_as_uint(o[1]) << 32) | __float_as_uint(o[0]);
  unsigned long long e1 = ((unsigned long long)__float_as_uint(o[3]) << 32) | __float_as_uint(o[2]);
  unsigned long long n0 = ((unsigned long long)__float_as_uint(o[1] + v[1]) << 32) | __float_as_uint(o[0] + v[0]);
  unsigned long long n1 = ((unsigned long long)__float_as_uint(o[3] + v[3]) << 32) | __float_as_uint(o[2] + v[2]);
  bool ok0 = __hip_atomic_compare_exchange_strong(q, &e0, n0, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  bool ok1 = __hip_atomic_compare_exchange_strong(q + 1, &e1, n1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (!ok0) {
    n0 = ((unsigned long long)__float_as_uint(__uint_as_float((unsigned)(e0 >> 32)) + v[1]) << 32) | __float_as_uint(__uint_as_float((unsigned)e0) + v[0]);
    ok0 = __hip_atomic_compare_exchange_strong(q, &e0, n0, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  while (!ok1) {
    n1 = ((unsigned long long)__float_as_uint(__uint_as_float((unsigned)(e1 >> 32)) + v[3]) << 32) | __float_as_uint(__uint_as_float((unsigned)e1) + v[2]);
    ok1 = __hip_atomic_compare_exchange_strong(q + 1, &e1, n1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

}  // namespace
