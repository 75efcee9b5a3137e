// Relation-owner backward of the hidden-16 relational layer on TALL tiles in soft-window order (round 6).
//
// Autograd duals of reference torch_rgcn/layers.py:293-301, one random row gather per message:
//     dX[o]  += val * G[s] W_r^T            dW_r += val * X[o]^T G[s]            for every message (s <- o, r, val)
//
// Why another form.  The gather of S1's 21 M rows takes 0.375 ms in random order and 0.19-0.22 ms when the whole chip reads from a few MB
// of the table at any time (tools/micro/gather_window.hip).  The plan gives that order for free when a (tile, relation) bucket holds many
// chunks (_native.build_softwin_plan: buckets sorted by source, a tile's chunks ordered by first source) -- which wants tiles of ~1000
// rows, and the block-tile kernel (rgcn_bwd_blk.hip) holds 218: its LDS keeps every relation's dW (R KiB) next to the X and dX tiles.
// Here dW sits in REGISTERS.  Every relation -- every part of a large one -- belongs to one of the workgroup's NW waves (LPT over the
// message counts, made with the plan); a wave walks only the chunks of its units and keeps their K accumulators (the MFMA's 16 x 16 D
// fragment: 4 registers each) in a <32 x float> register vector indexed by the chunk's local number (wave-uniform: s_set_gpr_idx_on +
// v_mov), the dW products accumulate in them directly.  No LDS table, no compare-and-swap adds, no dirty flags; the registers leave the
// CU once, at the end of the kernel (the same R KiB of global atomics per workgroup as the block-tile kernel's one flush).
// LDS = the dX tile in doubles (128 bytes per row, ds_add_f64 as in the block-tile kernel) + the X tile (64 bytes per row: A operand of
// the dW products, ReLU mask of the epilogue) + 1 KiB of transposition scratch per wave: tiles of up to 767 rows (S1: 652 rows, 1534
// tiles, 6 per CU; a bucket holds 135 messages = 8-9 chunks of ~8 MB source span).
// Shape: 16 waves x 8 units x 2 chunks per loop trip.  Measured at S1 (tools/r6_own_variants.sh, bench.py's per-launch
// average): 12 waves x 9 units x 4 chunks (152 VGPRs, the first shipped form) 0.460 ms; 16 x 7 x 4 (128 VGPRs, 11 spilled) 0.454;
// 16 x 8 x 3 0.437 -- the loads in flight per CU are the same 48 chunks, the fourth wave per SIMD hides the LDS and MFMA phases;
// 16 x 8 x 2 **0.432**, 16 x 8 x 1 0.447 (fewer chunks in flight narrow the span of sources the chip reads at one time).
// Measured and dropped on the way (tools/r6_own_abl.sh, profiles/r06_own_ablation.txt): the X rows read from global memory instead of an
// LDS tile (tiles of 977 rows fit then): 16 more loads per four chunks, +0.09 ms; accumulators updated by indexed adds: +0.03 ms.
//
// The chunk records are the block-tile kernel's (rgcn_bwd_blk_prepare_f32, 176 bytes); the relation word carries the local relation
// number in its high half: rel | li << 16.  own_ptr[tile * NW + wave] = first chunk of the wave in the tile (n_tiles * NW + 1 entries).
// No hub pieces: plans with hub tiles keep the block-tile kernel.
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "rgcn_device.h"

namespace {

constexpr int OWN_REC = 176;
constexpr int OWN_REC_ROWS = 128;
constexpr int OWN_REC_HDR = 160;
constexpr int OWN_LDS_MAX = 160 * 1024;
#ifndef RGCN_OWN_NW
#define RGCN_OWN_NW 16
#define RGCN_OWN_K 8
#define RGCN_OWN_U 2
#endif
constexpr int OWN_NW = RGCN_OWN_NW;  // waves per workgroup
constexpr int OWN_K = RGCN_OWN_K;    // relations per wave (accumulators in registers)
constexpr int OWN_U = RGCN_OWN_U;    // chunks per loop trip
constexpr int OWN_MAX_ROWS = (OWN_LDS_MAX - OWN_NW * BW_SCR2 * 4 - 64) / 192;      // 789: dX tile (doubles) + X tile + the waves' scratch

// the first 8 accumulators of a wave: ONE register vector indexed by the chunk's local relation number (a <32 x float> is the largest
// the compiler keeps in registers under a dynamic index: s_set_gpr_idx_on + v_mov; <36 x float> goes to scratch); the ninth is a
// vector of its own, fed through a wave-uniform select
typedef float own_acc32 __attribute__((ext_vector_type(32)));

size_t bwd_own_lds(int rows) { return (size_t)rows * 192 + (size_t)OWN_NW * BW_SCR2 * 4 + 64; }

// Timing experiments (ablation library only, make abl; wrong results): ABL bits 2 no dW part, 4 the accumulator index is always 0,
// 8 no dX tile update, 16 loads only
template <bool RELU, int NW, int K, int ABL = 0>
__global__ __launch_bounds__(64 * NW) void bwd_own_d16_kernel(
    const float *__restrict__ G, const float *__restrict__ X, const float *__restrict__ Wtp, float *__restrict__ dX,
    float *__restrict__ dWout, const char *__restrict__ rec, const int *__restrict__ own_ptr, int n_tiles, int tile_rows, int n_dst,
    float *__restrict__ dbias, int n_src, const int *__restrict__ unit_rel) {
  constexpr int U = OWN_U, NT = 64 * NW;
  constexpr int TQ = (OWN_MAX_ROWS * 4 + NT - 1) / NT;            // float4 of a tile a thread carries / converts per tile, at most
  static_assert(K <= 9, "8 indexed accumulators + 1");
  constexpr bool EXT = K == 9;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const unsigned xt_off = (unsigned)tile_rows * 128u;             // bytes: the dX tile [rows][16] doubles comes first
  const unsigned xs_off = (unsigned)tile_rows * 192u;
  float4 *dxz = reinterpret_cast<float4 *>(lds);
  const double2 *dxd2 = reinterpret_cast<const double2 *>(lds);
  float4 *xt4 = reinterpret_cast<float4 *>(lds + xt_off);         // X tile [rows][16] floats
  float *xs = reinterpret_cast<float *>(lds + xs_off) + wave * BW_SCR2;

  int t = blockIdx.x;
  int row0 = t * tile_rows;
  int nrows = min(tile_rows, n_dst - row0);
  int c0 = own_ptr[(size_t)t * NW + wave], c1 = own_ptr[(size_t)t * NW + wave + 1];
  c0 = __builtin_amdgcn_readfirstlane(c0);
  c1 = __builtin_amdgcn_readfirstlane(c1);
  float4 gs = make_float4(0.f, 0.f, 0.f, 0.f);
  const long long g_n4 = dbias ? (long long)n_src * 4 : 0, g_step = (long long)gridDim.x * NT;
  long long g_i = (long long)blockIdx.x * NT + tid;
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
  __syncthreads();

  const int m = lane & 15, k = lane >> 4;
  const unsigned kofs = (unsigned)k << 4;
  const unsigned dx_lane = (unsigned)m * 8u;                            // LDS byte address of dX tile [0][m]
  const unsigned xrd = xt_off + (unsigned)m * 4u;                       // LDS byte address of X tile [0][m]
  // scratch (as in the block-tile kernel): the float4 column (features 4c .. 4c+3) of slot s is stored at column (c + (s >> 2)) & 3
  float *xs_wr = xs + m * 16 + 4 * ((k + (m >> 2)) & 3);                // this lane's float4: features 4k .. 4k+3 of slot m
  const float *xs_rd = xs + (4 * k) * 16 + 4 * (((m >> 2) + k) & 3) + (m & 3);   // feature m of slot 4k + t: + 16 t
  const unsigned slot_lane = (unsigned)m * 8u;
  const unsigned rows_lane = (unsigned)OWN_REC_ROWS + (unsigned)k * 8u;
  const unsigned w_lane = (unsigned)lane * 16u;

  own_acc32 accs;
#pragma unroll
  for (int i = 0; i < 32; ++i) accs[i] = 0.f;
  f32x4 acc8 = f32x4{0.f, 0.f, 0.f, 0.f};

  uint2 sl_n[U], rw_n[U];
  int hd_n[U];
  auto request_idx = [&](int c, int last) {
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int cc = min(c + j, last);                          // scalar: chunks past the wave's range re-read its last chunk (val forced to 0)
      const char *r = rec + (size_t)cc * OWN_REC;
      sl_n[j] = *reinterpret_cast<const uint2 *>(r + slot_lane);
      rw_n[j] = *reinterpret_cast<const uint2 *>(r + rows_lane);
      hd_n[j] = *reinterpret_cast<const int *>(r + OWN_REC_HDR);
    }
  };
  if (c0 < c1) request_idx(c0, c1 - 1);

  for (;;) {
    const int last = c1 - 1;
    for (int c = c0; c < c1; c += U) {
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
        w_[j] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(Wtp) + (size_t)(hd_[j] & 0xFFFF) * 1024 + w_lane);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (c + U < c1) request_idx(c + U, last);
      __builtin_amdgcn_sched_barrier(0);
      if (ABL & 16) {
#pragma unroll
        for (int j = 0; j < U; ++j) asm volatile("" :: "v"(g_[j].x), "v"(g_[j].y), "v"(g_[j].z), "v"(g_[j].w), "v"(w_[j].x), "v"(w_[j].w), "v"(rw_[j].x), "v"(rw_[j].y), "v"(v_[j]));
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
      // ---- phase 2: dW products, two chunks at a time: the scaled rows go through the wave's scratch into K-over-messages layout, the
      // X rows of the slots come from the X tile; the products accumulate IN the wave's register accumulator of the chunk's relation
#pragma unroll
      for (int h = 0; h < U && !(ABL & 2); h += 2) {
        constexpr int P = 2;
        float bv[P][4], av[P][4];
#pragma unroll
        for (int jj = 0; jj < P; ++jj) {
          const int j = h + jj;
          if (j >= U) break;
          asm volatile("" ::: "memory");
          *reinterpret_cast<f32x4 *>(xs_wr) = sc[j];
          asm volatile("" ::: "memory");
#pragma unroll
          for (int t4 = 0; t4 < 4; ++t4) bv[jj][t4] = xs_rd[16 * t4];
#pragma unroll
          for (int t4 = 0; t4 < 4; ++t4) av[jj][t4] = *reinterpret_cast<const float *>(lds + (xrd + ro[j][t4]));
          asm volatile("" ::: "memory");
        }
        // (two chunks of one pair may share their relation: the second reads what the first wrote)
#pragma unroll
        for (int jj = 0; jj < P; ++jj) {
          if (h + jj >= U) break;
          const int lr = (ABL & 4) ? 0 : (int)((unsigned)hd_[h + jj] >> 16);
          const bool ext = EXT && lr >= 8;           // (wave-uniform selects, not multiplications by 0 / 1: an Inf stays in ITS relation)
          const int li = 4 * (lr & 7);
          const f32x4 in0 = f32x4{accs[li], accs[li + 1], accs[li + 2], accs[li + 3]};
          f32x4 aw = EXT ? f32x4{ext ? acc8[0] : in0[0], ext ? acc8[1] : in0[1], ext ? acc8[2] : in0[2], ext ? acc8[3] : in0[3]} : in0;
#pragma unroll
          for (int t4 = 0; t4 < 4; ++t4) aw = __builtin_amdgcn_mfma_f32_16x16x4f32(av[jj][t4], bv[jj][t4], aw, 0, 0, 0);
          accs[li] = ext ? in0[0] : aw[0];
          accs[li + 1] = ext ? in0[1] : aw[1];
          accs[li + 2] = ext ? in0[2] : aw[2];
          accs[li + 3] = ext ? in0[3] : aw[3];
          if (EXT) acc8 = f32x4{ext ? aw[0] : acc8[0], ext ? aw[1] : acc8[1], ext ? aw[2] : acc8[2], ext ? aw[3] : acc8[3]};
        }
      }
      // ---- phase 3: the tile update, one ds_add_f64 per slot quarter: the 16 lanes of a quarter wave add to the 16 features of one row
#pragma unroll
      for (int j = 0; j < U && !(ABL & 8); ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          __hip_atomic_fetch_add(static_cast<double *>(__builtin_assume_aligned(lds + (dx_lane + 2u * ro[j][e]), 8)), (double)acc[j][e], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_WORKGROUP);
      __builtin_amdgcn_sched_barrier(0);
    }
    // the next tile of this workgroup: its X rows, this wave's first chunks and one float4 of G (bias gradient) are requested before the barrier
    const int tn = t + (int)gridDim.x;
    int c0n = 0, c1n = 0, row0n = 0, nrn = 0;
    float4 xn[TQ];
#pragma unroll
    for (int q = 0; q < TQ; ++q) xn[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tn < n_tiles) {
      row0n = tn * tile_rows;
      nrn = min(tile_rows, n_dst - row0n);
      c0n = __builtin_amdgcn_readfirstlane(own_ptr[(size_t)tn * NW + wave]);
      c1n = __builtin_amdgcn_readfirstlane(own_ptr[(size_t)tn * NW + wave + 1]);
#pragma unroll
      for (int q = 0; q < TQ; ++q)
        if (tid + q * NT < nrn * 4) xn[q] = reinterpret_cast<const float4 *>(X + (size_t)row0n * 16)[tid + q * NT];
      if (c0n < c1n) request_idx(c0n, c1n - 1);
    }
    float4 gn = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g_i < g_n4) gn = reinterpret_cast<const float4 *>(G)[g_i];
    g_i += g_step;
    lds_barrier();                                                 // every wave has finished adding to the dX tile
    gs.x += gn.x; gs.y += gn.y; gs.z += gn.z; gs.w += gn.w;
    // (what was requested before the barrier is pinned as arrived BEFORE the tile's stores: vmcnt counts in order)
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
        reinterpret_cast<float4 *>(dX + (size_t)row0 * 16)[idx] = a;
      }
    }
    if (tn >= n_tiles) break;
#pragma unroll
    for (int q = 0; q < TQ; ++q) {
      const int idx = tid + q * NT;
      if (idx < tile_rows * 4) {
        dxz[2 * idx] = make_float4(0.f, 0.f, 0.f, 0.f);
        dxz[2 * idx + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
        xt4[idx] = xn[q];
      }
    }
    lds_barrier();                                                 // the next tile is installed (nobody waits for the dX stores)
    t = tn; row0 = row0n; nrows = nrn; c0 = c0n; c1 = c1n;
  }
  if (dbias) {
    for (; g_i < g_n4; g_i += g_step) {
      const float4 gn = reinterpret_cast<const float4 *>(G)[g_i];
      gs.x += gn.x; gs.y += gn.y; gs.z += gn.z; gs.w += gn.w;
    }
    // thread tid holds features 4 (tid & 3) .. + 3: fold the 16 lanes of a wave that share (lane & 3), then the waves through the scratch
#pragma unroll
    for (int sft = 4; sft < 64; sft <<= 1) {
      gs.x += __shfl_xor(gs.x, sft); gs.y += __shfl_xor(gs.y, sft); gs.z += __shfl_xor(gs.z, sft); gs.w += __shfl_xor(gs.w, sft);
    }
    __syncthreads();                                               // (every wave is done with its scratch)
    if (lane < 4) *reinterpret_cast<float4 *>(xs + 4 * lane) = gs;
    __syncthreads();
    if (tid < 16) {
      float a = 0.f;
      const float *all = reinterpret_cast<const float *>(lds + xs_off);
      for (int i = 0; i < NW; ++i) a += all[i * BW_SCR2 + tid];
      atomicAdd(dbias + tid, a);
    }
  }
  // the wave's accumulators leave the CU once.  D fragment: lane 16k + m, element e = row 4k + e (input feature), column m
#pragma unroll
  for (int li = 0; li < K; ++li) {
    const int r = __builtin_amdgcn_readfirstlane(unit_rel[wave * K + li]);
    if (r >= 0) {
      float *o = dWout + (size_t)r * 256 + (4 * k) * 16 + m;
      atomicAdd(o, li < 8 ? accs[(4 * li) & 31] : acc8[0]);
      atomicAdd(o + 16, li < 8 ? accs[(4 * li + 1) & 31] : acc8[1]);
      atomicAdd(o + 32, li < 8 ? accs[(4 * li + 2) & 31] : acc8[2]);
      atomicAdd(o + 48, li < 8 ? accs[(4 * li + 3) & 31] : acc8[3]);
    }
  }
}

}  // namespace

extern "C" int32_t rgcn_bwd_own_waves(void) { return OWN_NW; }
extern "C" int32_t rgcn_bwd_own_units(void) { return OWN_NW * OWN_K; }
extern "C" int32_t rgcn_bwd_own_max_rows(void) { return OWN_MAX_ROWS; }

extern "C" int rgcn_bwd_own_f32(const float *G, const float *X, const float *Wt_packed, float *dX, float *dW, const void *rec,
                                const int32_t *own_ptr, const int32_t *unit_rel, int64_t n_tiles, int32_t tile_rows, int64_t n_dst,
                                int32_t R, int32_t flags, float *dbias, int64_t n_src, void *stream) {
  if (!G || !X || !Wt_packed || !dX || !dW || !rec || !own_ptr || !unit_rel || n_tiles <= 0 || tile_rows <= 0 || n_dst <= 0 || R <= 0 ||
      R > 0xFFFF || n_dst > INT32_MAX || n_tiles * (int64_t)OWN_NW >= INT32_MAX) {
    rgcn_set_error("bwd_own: bad argument");
    return RGCN_EINVAL;
  }
  if (tile_rows > rgcn_bwd_own_max_rows()) {
    rgcn_set_error("bwd_own: tiles of at most %d rows (got %d)", rgcn_bwd_own_max_rows(), tile_rows);
    return RGCN_EUNSUPPORTED;
  }
  if (dbias && (n_src <= 0 || n_src >= (int64_t(1) << 29))) { rgcn_set_error("bwd_own: dbias needs 0 < n_src < 2^29"); return RGCN_EINVAL; }
  hipStream_t st = (hipStream_t)stream;
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0, v = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
    n_cu = v > 0 ? v : 256;
  }
  if (dbias == dW + (size_t)R * 256) {     // one fill for both when the caller laid them out back to back
    HIP_TRY(zero_async(dW, ((size_t)R * 256 + 16) * sizeof(float), st));
  } else {
    HIP_TRY(zero_async(dW, (size_t)R * 256 * sizeof(float), st));
    if (dbias) HIP_TRY(zero_async(dbias, 16 * sizeof(float), st));
  }
  const size_t lds = bwd_own_lds(tile_rows);
  const unsigned n_blocks = (unsigned)std::min<int64_t>(n_tiles, n_cu);
  auto launch = [&](auto kern, bool &raised) -> hipError_t {
    if (lds > 64 * 1024 && !raised) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, OWN_LDS_MAX);
      if (e != hipSuccess) return e;
      raised = true;
    }
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(64 * OWN_NW), lds, st, G, X, Wt_packed, dX, dW, static_cast<const char *>(rec), own_ptr, (int)n_tiles,
                       tile_rows, (int)n_dst, dbias, (int)n_src, unit_rel);
    return hipGetLastError();
  };
#ifdef RGCN_ABLATIONS
  {
    const int ABLV = rgcn_option_value(RGCN_OPT_BWD_ABL);
    static bool a[8] = {false, false, false, false, false, false, false, false};
    if (ABLV == 2) HIP_TRY(launch(bwd_own_d16_kernel<false, OWN_NW, OWN_K, 2>, a[1]));
    else if (ABLV == 4) HIP_TRY(launch(bwd_own_d16_kernel<false, OWN_NW, OWN_K, 4>, a[3]));
    else if (ABLV == 8) HIP_TRY(launch(bwd_own_d16_kernel<false, OWN_NW, OWN_K, 8>, a[4]));
    else if (ABLV == 16) HIP_TRY(launch(bwd_own_d16_kernel<false, OWN_NW, OWN_K, 16>, a[5]));
    else if (ABLV == 6) HIP_TRY(launch(bwd_own_d16_kernel<false, OWN_NW, OWN_K, 6>, a[6]));
    else HIP_TRY(launch(bwd_own_d16_kernel<false, OWN_NW, OWN_K, 0>, a[7]));
    return RGCN_OK;
  }
#endif
  static bool r0 = false, r1 = false;
  if (flags & RGCN_F_RELU) HIP_TRY(launch(bwd_own_d16_kernel<true, OWN_NW, OWN_K>, r0));
  else HIP_TRY(launch(bwd_own_d16_kernel<false, OWN_NW, OWN_K>, r1));
  return RGCN_OK;
}
