// Dense contractions of the basis-decomposed layer on the matrix cores, hand-written (no rocBLAS on the path).
//
// Reference lines: torch_rgcn/layers.py:241-242, :468-469 (einsum('rb,bio->rio')) followed by the per-relation products of
// layers.py:293-301 / :524-551.  With W_r = sum_b comps[r,b] bases[b] the layer is
//     out[s] = sum_b ( sum_e comps[r_e,b] val_e X[o_e] ) bases[b] + bias  =  ag[s, :] @ flat(bases) + bias,
// ag[N, B d_in] the per-basis aggregation and flat(bases) [B d_in, d_out]: the "(B V) H" contraction north_star assigns to
// MFMA.  Here:
//   gemm_kernel              LDS-tiled fp32 MFMA GEMM, 128 x 128 x 16 tiles, double-buffered, either operand K-inner or
//                            K-outer, optional split-K with a fixed-order reduction: the backward's two products
//                            d_ag = g flat^T  and  dbases = ag^T g  (and the wide-block / generic fallbacks).
// fp32 in, fp32 accumulate: the MFMA f32 forms are exact fp32 FMA chains (1e-4 bound of north_star needs no care).
#include <stdlib.h>

#include <algorithm>

#include "rgcn_device.h"

namespace {

constexpr int GT = 128;          // C tile per workgroup (GT x GT), 64 x 64 per wave
constexpr int GK = 16;           // K per step
constexpr int LD_IN = 20;        // LDS row stride of a K-inner slab [128][16]: conflict-free ds_read_b128 (round 1, rank kernel)
constexpr int LD_OUT = 132;      // LDS row stride of a K-outer slab [16][128]: (4 kq + c) * 132 + i covers 32 banks

// ------------------------------------------------------------------ general GEMM  C[M,N] = op(A) op(B) (+ bias[n])
// TA: A is stored [K][M] (M contiguous) instead of [M][K];  TB: B is stored [N][K] (K contiguous) instead of [K][N].
// VEC: both operands are read with 16-byte loads (aligned bases, leading dimensions and contiguous extents multiples of 4) -- a
// TEMPLATE parameter: decided at run time inside the kernel, the two load forms become branches, and hipcc then waits for the
// outstanding loads between the four loads of a slab (s_waitcnt vmcnt(2) at every join) -- the global latency of each slab was
// paid before its MFMAs started instead of under them (ablation: loads 12 us + MFMAs 22 us + rest 19 us = the whole 53 us of a
// one-workgroup-per-CU launch, nothing overlapped).
// BM: rows of the C tile (128 x 128 or 64 x 128 per workgroup).  64-row tiles halve the work of a tile: launches whose
// 128-row tile count is an awkward multiple of the CU count (WN18: 640 tiles over 256 CUs = 3 rounds on some CUs, 2 on others)
// balance better with twice as many half tiles (rgcn_gemm_f32 picks per launch).
template <bool TA, bool TB, bool VEC, int BM>
__global__ __launch_bounds__(WG) void gemm_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                  const float *__restrict__ bias, float *__restrict__ C, int M, int N, int K,
                                                  long long lda, long long ldb, long long ldc, int tiles_m, int k_per_split,
                                                  long long split_stride) {
  constexpr int LDA_OUT = BM + 4;          // K-outer A slab [16][BM]
  constexpr int NA = BM / 32;              // 16 x 16 row tiles per wave (the wave's quadrant is BM / 2 x 64)
  __shared__ __attribute__((aligned(16))) float sA[2][TA ? GK * LDA_OUT : BM * LD_IN];
  __shared__ __attribute__((aligned(16))) float sB[2][TB ? GT * LD_IN : GK * LD_OUT];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 15, kq = lane >> 4;
  const int m0 = (blockIdx.x % tiles_m) * BM, n0 = (blockIdx.x / tiles_m) * GT;
  const int kb = blockIdx.y * k_per_split, ke = min(K, kb + k_per_split);
  const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * 64;

  f32x4 acc[NA][4];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging: K-inner slab [ROWS][16]: thread -> rows (tid >> 2) (+ 64), k group 4 (tid & 3)
  //          K-outer slab [16][COLS]: thread -> k rows tid / (COLS / 4) (+ 256 / (COLS / 4)), column group 4 (tid % (COLS / 4))
  f32x4 ra[2], rb[2];
  constexpr bool vec_a = VEC, vec_b = VEC;
  auto fetch_inner = [&](const float *P, long long ld, int r0, int rows, int k0, bool vec, f32x4 (&reg)[2], int passes) {
    const int kk = k0 + 4 * (tid & 3);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h >= passes) continue;
      const float *row = P + (size_t)min(r0 + (tid >> 2) + 64 * h, rows - 1) * ld;
      if (vec) {
        reg[h] = *reinterpret_cast<const f32x4 *>(row + min(kk, K - 4));
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) reg[h][c] = row[min(kk + c, ke - 1)];
      }
    }
  };
  auto stash_inner = [&](float *S, int k0, const f32x4 (&reg)[2], int passes) {
    const int kk = k0 + 4 * (tid & 3);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h >= passes) continue;
      f32x4 v = reg[h];
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = (kk + c < ke) ? v[c] : 0.f;
      *reinterpret_cast<f32x4 *>(S + ((tid >> 2) + 64 * h) * LD_IN + 4 * (tid & 3)) = v;
    }
  };
  // cg = column groups of 4 per slab row (32 for 128 columns, 16 for 64), 256 / cg k rows per pass, 16 / (256 / cg) passes
  auto fetch_outer = [&](const float *P, long long ld, int c0, int cols, int k0, bool vec, f32x4 (&reg)[2], int cg) {
    const int cc = c0 + 4 * (tid % cg), rp = WG / cg;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h * rp >= GK) continue;
      const float *row = P + (size_t)min(k0 + tid / cg + rp * h, ke - 1) * ld;
      if (vec) {          // columns past the matrix edge belong to rows / columns of C that are never stored
        reg[h] = *reinterpret_cast<const f32x4 *>(row + min(cc, cols - 4));
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) reg[h][c] = row[min(cc + c, cols - 1)];
      }
    }
  };
  auto stash_outer = [&](float *S, int k0, const f32x4 (&reg)[2], int cg, int ld_s) {
    const int rp = WG / cg;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h * rp >= GK) continue;
      const bool live = k0 + tid / cg + rp * h < ke;
      *reinterpret_cast<f32x4 *>(S + (tid / cg + rp * h) * ld_s + 4 * (tid % cg)) = live ? reg[h] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto fetch = [&](int k0) {
    if (TA) fetch_outer(A, lda, m0, M, k0, vec_a, ra, BM / 4); else fetch_inner(A, lda, m0, M, k0, vec_a, ra, BM / 64);
    if (TB) fetch_inner(B, ldb, n0, N, k0, vec_b, rb, 2); else fetch_outer(B, ldb, n0, N, k0, vec_b, rb, GT / 4);
  };
  auto stash = [&](int buf, int k0) {
    if (TA) stash_outer(sA[buf], k0, ra, BM / 4, LDA_OUT); else stash_inner(sA[buf], k0, ra, BM / 64);
    if (TB) stash_inner(sB[buf], k0, rb, 2); else stash_outer(sB[buf], k0, rb, GT / 4, LD_OUT);
  };

  const int steps = (ke - kb + GK - 1) / GK;
  if (steps > 0) {
    fetch(kb);
    stash(0, kb);
    __syncthreads();
    for (int t = 0; t < steps; ++t) {
      const int cur = t & 1;
      if (t + 1 < steps) fetch(kb + GK * (t + 1));
      f32x4 av[NA], bv[4];
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        if (TA) {
#pragma unroll
          for (int c = 0; c < 4; ++c) av[a][c] = sA[cur][(4 * kq + c) * LDA_OUT + wm + 16 * a + i];
        } else {
          av[a] = *reinterpret_cast<const f32x4 *>(&sA[cur][(wm + 16 * a + i) * LD_IN + 4 * kq]);
        }
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (TB) {
          bv[b] = *reinterpret_cast<const f32x4 *>(&sB[cur][(wn + 16 * b + i) * LD_IN + 4 * kq]);
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) bv[b][c] = sB[cur][(4 * kq + c) * LD_OUT + wn + 16 * b + i];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a][c], bv[b][c], acc[a][b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < steps) stash(cur ^ 1, kb + GK * (t + 1));
      __syncthreads();
    }
  }
  // D: lane 16 q + j holds rows 4 q .. 4 q + 3 (M index), column j (N index) of every 16 x 16 tile
  float *Cs = C + (size_t)blockIdx.y * split_stride;
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + wm + 16 * a + 4 * kq + r;
      if (row >= M) continue;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int col = n0 + wn + 16 * b + i;
        if (col >= N) continue;
        Cs[(size_t)row * ldc + col] = acc[a][b][r] + ((bias && gridDim.y == 1) ? bias[col] : 0.f);
      }
    }
}

// ------------------------------------------------------------------ row-panel GEMM: (a slice of) N in ONE tile
// The 128-wide tiles above pay for N rounded up to 128: the WN18 layer's products have N = 200 and 400 (256 / 512 computed: 22 % of the
// MFMAs wasted), and 640 tiles of 128 x 128 over 256 CUs leave half the chip with 3 tiles and half with 2.  Here a workgroup computes 64 rows
// x 16 NB columns (NB <= 13: 208 columns = 200 rounded up to MFMA tiles), the four waves side by side along M (16 rows each, NB accumulator
// tiles per wave): N = 200 is one panel, 400 two of 208, and the 640 row panels are all resident at once.  Slabs keep their operand's
// orientation in LDS as in gemm_kernel (16-byte stores; K-inner slabs are read with 16-byte LDS reads, K-outer ones with four scalar
// reads per operand: transposing a K-outer slab in the stash put its 4 scalar writes 16 ways into the same banks -- 0.130 ms per product
// against gemm_kernel's 0.111).  16-byte global loads only (the launcher falls back to gemm_kernel otherwise).
template <bool TA, bool TB, int NB>
__global__ __launch_bounds__(WG) void gemm_panel_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                        const float *__restrict__ bias, float *__restrict__ C, int M, int N, int K,
                                                        long long lda, long long ldb, long long ldc, int tiles_m, int k_per_split,
                                                        long long split_stride) {
  constexpr int BM = 64, BN = 16 * NB;
  constexpr int NLB = (BN * 4 + WG - 1) / WG;          // float4 loads of the B slab per thread (16 x BN floats)
  // K-outer slabs [16][LD]: LD = 16 mod 32, so that the four k rows a wave's scalar operand read touches (16 consecutive floats each) fall
  // into four disjoint groups of 16 banks
  constexpr int LDA_O = 80, LDB_O = (BN % 32 == 16) ? BN : BN + 16;
  __shared__ __attribute__((aligned(16))) float sA[2][TA ? GK * LDA_O : BM * LD_IN];
  __shared__ __attribute__((aligned(16))) float sB[2][TB ? BN * LD_IN : GK * LDB_O];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 15, kq = lane >> 4;
  const int m0 = (blockIdx.x % tiles_m) * BM, n0 = (blockIdx.x / tiles_m) * BN;
  const int kb = blockIdx.y * k_per_split, ke = min(K, kb + k_per_split);

  f32x4 acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 ra, rb[NLB];
  auto fetch = [&](int k0) {
    if (TA) {          // A stored [K][M]: k row tid >> 4, columns m0 + 4 (tid & 15) ..
      ra = *reinterpret_cast<const f32x4 *>(A + (size_t)min(k0 + (tid >> 4), ke - 1) * lda + min(m0 + 4 * (tid & 15), M - 4));
    } else {           // A stored [M][K]: row tid >> 2, k group 4 (tid & 3)
      ra = *reinterpret_cast<const f32x4 *>(A + (size_t)min(m0 + (tid >> 2), M - 1) * lda + min(k0 + 4 * (tid & 3), K - 4));
    }
#pragma unroll
    for (int h = 0; h < NLB; ++h) {
      const int idx = min(tid + WG * h, BN * 4 - 1);
      if (TB) {        // B stored [N][K]: row (a column of C) idx >> 2, k group 4 (idx & 3)
        rb[h] = *reinterpret_cast<const f32x4 *>(B + (size_t)min(n0 + (idx >> 2), N - 1) * ldb + min(k0 + 4 * (idx & 3), K - 4));
      } else {         // B stored [K][N]: k row idx / (BN / 4), columns n0 + 4 (idx % (BN / 4)) ..  (columns past N: duplicates, never stored)
        rb[h] = *reinterpret_cast<const f32x4 *>(B + (size_t)min(k0 + idx / (BN / 4), ke - 1) * ldb + min(n0 + 4 * (idx % (BN / 4)), N - 4));
      }
    }
  };
  auto stash = [&](int buf, int k0) {
    if (TA) {
      const bool live = k0 + (tid >> 4) < ke;
      *reinterpret_cast<f32x4 *>(&sA[buf][(tid >> 4) * LDA_O + 4 * (tid & 15)]) = live ? ra : f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
      const int kk = k0 + 4 * (tid & 3);
      f32x4 v = ra;
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = (kk + c < ke) ? v[c] : 0.f;
      *reinterpret_cast<f32x4 *>(&sA[buf][(tid >> 2) * LD_IN + 4 * (tid & 3)]) = v;
    }
#pragma unroll
    for (int h = 0; h < NLB; ++h) {
      const int idx = tid + WG * h;
      if (idx >= BN * 4) continue;
      if (TB) {
        const int kk = k0 + 4 * (idx & 3);
        f32x4 v = rb[h];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = (kk + c < ke) ? v[c] : 0.f;
        *reinterpret_cast<f32x4 *>(&sB[buf][(idx >> 2) * LD_IN + 4 * (idx & 3)]) = v;
      } else {
        const int kr = idx / (BN / 4), cg = idx % (BN / 4);
        const bool live = k0 + kr < ke;
        *reinterpret_cast<f32x4 *>(&sB[buf][kr * LDB_O + 4 * cg]) = live ? rb[h] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  const int steps = (ke - kb + GK - 1) / GK;
  if (steps > 0) {
    fetch(kb);
    stash(0, kb);
    __syncthreads();
    for (int t = 0; t < steps; ++t) {
      const int cur = t & 1;
      if (t + 1 < steps) fetch(kb + GK * (t + 1));
      f32x4 av;
      if (TA) {
#pragma unroll
        for (int c = 0; c < 4; ++c) av[c] = sA[cur][(4 * kq + c) * LDA_O + 16 * wave + i];
      } else {
        av = *reinterpret_cast<const f32x4 *>(&sA[cur][(16 * wave + i) * LD_IN + 4 * kq]);
      }
      // The B operands are read from LDS one phase AHEAD of the MFMAs that use them, the phases pinned with scheduling barriers.  Left to
      // itself hipcc kept ONE register pair for all of them: read, s_waitcnt lgkmcnt(0), two MFMAs, read ... -- an LDS round trip exposed
      // between every two MFMAs of a wave (found in the ISA; the kernel sat at 0.42 of the MFMA peak whatever its tiles and prefetches).
      if (TB) {          // K-inner slab: a 16-byte read is one column tile's four K values -> phases of four column tiles
        constexpr int G = 4, NG = (NB + G - 1) / G;
        f32x4 bg[2][G];
#pragma unroll
        for (int j = 0; j < G; ++j) bg[0][j] = *reinterpret_cast<const f32x4 *>(&sB[cur][(16 * min(j, NB - 1) + i) * LD_IN + 4 * kq]);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (g + 1 < NG) {
#pragma unroll
            for (int j = 0; j < G; ++j)
              bg[(g + 1) & 1][j] = *reinterpret_cast<const f32x4 *>(&sB[cur][(16 * min((g + 1) * G + j, NB - 1) + i) * LD_IN + 4 * kq]);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < G; ++j)
              if (g * G + j < NB) acc[g * G + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], bg[g & 1][j][c], acc[g * G + j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {           // K-outer slab: a phase is one K value of every column tile
        float bc[2][NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) bc[0][b] = sB[cur][(4 * kq) * LDB_O + 16 * b + i];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (c < 3) {
#pragma unroll
            for (int b = 0; b < NB; ++b) bc[(c + 1) & 1][b] = sB[cur][(4 * kq + c + 1) * LDB_O + 16 * b + i];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], bc[c & 1][b], acc[b], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (t + 1 < steps) stash(cur ^ 1, kb + GK * (t + 1));
      __syncthreads();
    }
  }
  // D: lane 16 q + j holds rows 4 q .. 4 q + 3 (M index), column j (N index) of every 16 x 16 tile
  float *Cs = C + (size_t)blockIdx.y * split_stride;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = m0 + 16 * wave + 4 * kq + r;
    if (row >= M) continue;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int col = n0 + 16 * b + i;
      if (col >= N) continue;
      Cs[(size_t)row * ldc + col] = acc[b][r] + ((bias && gridDim.y == 1) ? bias[col] : 0.f);
    }
  }
}

// C[i] = bias[i % N] + sum_s partial[s][i]   (fixed order)
__global__ __launch_bounds__(WG) void sum_slices_kernel(const float *__restrict__ partial, const float *__restrict__ bias,
                                                        float *__restrict__ C, long long n, int N, int S, long long ldc, int vec) {
  if (vec) {           // n, N, ldc multiples of 4, aligned: 16 bytes per lane and four slices in flight (64 slices of a 400 x 200 product: 17 -> 7 us);
                       // the slices are added in the same order as below
    const long long n4 = n >> 2;
    for (long long i4 = (long long)blockIdx.x * WG + threadIdx.x; i4 < n4; i4 += (long long)gridDim.x * WG) {
      const long long idx = 4 * i4;
      f32x4 a = bias ? *reinterpret_cast<const f32x4 *>(bias + idx % N) : f32x4{0.f, 0.f, 0.f, 0.f};
      int s = 0;
      for (; s + 3 < S; s += 4) {
        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(partial + (size_t)s * n + idx), v1 = *reinterpret_cast<const f32x4 *>(partial + (size_t)(s + 1) * n + idx);
        const f32x4 v2 = *reinterpret_cast<const f32x4 *>(partial + (size_t)(s + 2) * n + idx), v3 = *reinterpret_cast<const f32x4 *>(partial + (size_t)(s + 3) * n + idx);
        a += v0; a += v1; a += v2; a += v3;
      }
      for (; s < S; ++s) a += *reinterpret_cast<const f32x4 *>(partial + (size_t)s * n + idx);
      *reinterpret_cast<f32x4 *>(C + (idx / N) * ldc + idx % N) = a;
    }
    return;
  }
  for (long long idx = (long long)blockIdx.x * WG + threadIdx.x; idx < n; idx += (long long)gridDim.x * WG) {
    float a = bias ? bias[idx % N] : 0.f;
    for (int s = 0; s < S; ++s) a += partial[(size_t)s * n + idx];
    C[(idx / N) * ldc + idx % N] = a;
  }
}

// ------------------------------------------------------------------ wide undecomposed layers: relation-grouped gather-GEMM
// Widths above 64 without a decomposition (W [R, d_in, d_out]; layers.py:293-301 with e.g. d = 100, 200).  Round 1 cut W into
// 64-wide block pairs and launched the hidden-16-style block kernel once per pair, re-gathering X for every column block.
// Here the RELATION-major plan (chunks of 16 slots of one relation, contiguous per relation) is read as row blocks of a
// GEMM: a work item (<= 8 chunks = 128 message slots of ONE relation r) x a 128-column block of the output is one
// workgroup tile of the LDS-tiled MFMA GEMM above, with the A rows GATHERED (val[slot] * Xs[src[slot], :]) and B = W_r:
//     Y[slot, :] = val[slot] * Xs[src[slot], :] @ W_r                       (rel_rows_kernel; forward and, with W^T and G, dX)
// followed by the per-destination sum of the rows (segment_gather_sum_wide_kernel) -- X is gathered once per message,
// every W_r streams through LDS once per 128 messages.  The weight gradient is the same tiling transposed:
//     dW_r[m, n] += sum_{slots of r} Xs[src[slot], m] * (val[slot] * G[dst[slot], n])      (rel_wgrad_kernel, K = slots)
// one workgroup per (item, 128 x 128 block of dW_r), fp32 atomics across the items of a relation.
template <bool VEC>      // 16-byte loads of both operands (K and N multiples of 4): compile-time, see gemm_kernel
__global__ __launch_bounds__(WG) void rel_rows_kernel(
    const float *__restrict__ Xs, const float *__restrict__ W /* [R][K][N] */, float *__restrict__ Y /* [slots][N] */,
    const int *__restrict__ p_src, const float *__restrict__ p_val, const int *__restrict__ chunk_rel,
    const int2 *__restrict__ items, int K, int N) {
  __shared__ __attribute__((aligned(16))) float sA[2][GT * LD_IN];
  __shared__ __attribute__((aligned(16))) float sB[2][GK * LD_OUT];
  const int2 range = items[blockIdx.x];
  if (range.x >= range.y) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 15, kq = lane >> 4;
  const int rel = chunk_rel[range.x];
  const int slot0 = range.x * RGCN_CHUNK, M = (range.y - range.x) * RGCN_CHUNK;        // <= 128 rows
  const int n0 = blockIdx.y * GT;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const float *Wr = W + (size_t)rel * K * N;
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // A staging: thread -> rows (tid >> 2) and (tid >> 2) + 64 (gathered source rows, scaled), k group 4 (tid & 3)
  const float *arow[2];
  float ascale[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int rr = (tid >> 2) + 64 * h;
    const bool on = rr < M;
    const int slot = slot0 + (on ? rr : 0);
    arow[h] = Xs + (size_t)p_src[slot] * K;
    ascale[h] = on ? p_val[slot] : 0.f;            // pads carry val = 0
  }
  constexpr bool vec_a = VEC, vec_b = VEC;
  f32x4 ra[2], rb[2];
  auto fetch = [&](int k0) {
    const int kk = k0 + 4 * (tid & 3);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (vec_a) {
        ra[h] = *reinterpret_cast<const f32x4 *>(arow[h] + min(kk, K - 4));
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) ra[h][c] = arow[h][min(kk + c, K - 1)];
      }
      const float *brow = Wr + (size_t)min(k0 + (tid >> 5) + 8 * h, K - 1) * N;
      const int cc = n0 + 4 * (tid & 31);
      if (vec_b) {
        rb[h] = *reinterpret_cast<const f32x4 *>(brow + min(cc, N - 4));
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) rb[h][c] = brow[min(cc + c, N - 1)];
      }
    }
  };
  auto stash = [&](int buf, int k0) {
    const int kk = k0 + 4 * (tid & 3);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 v = ra[h] * ascale[h];
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = (kk + c < K && ascale[h] != 0.f) ? v[c] : 0.f;
      *reinterpret_cast<f32x4 *>(&sA[buf][((tid >> 2) + 64 * h) * LD_IN + 4 * (tid & 3)]) = v;
      const bool live = k0 + (tid >> 5) + 8 * h < K;
      *reinterpret_cast<f32x4 *>(&sB[buf][((tid >> 5) + 8 * h) * LD_OUT + 4 * (tid & 31)]) = live ? rb[h] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  const int steps = (K + GK - 1) / GK;
  fetch(0);
  stash(0, 0);
  __syncthreads();
  for (int t = 0; t < steps; ++t) {
    const int cur = t & 1;
    if (t + 1 < steps) fetch(GK * (t + 1));
    f32x4 av[4], bv[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      av[a] = *reinterpret_cast<const f32x4 *>(&sA[cur][(wm + 16 * a + i) * LD_IN + 4 * kq]);
#pragma unroll
      for (int c = 0; c < 4; ++c) bv[a][c] = sB[cur][(4 * kq + c) * LD_OUT + wn + 16 * a + i];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a][c], bv[b][c], acc[a][b], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < steps) stash(cur ^ 1, GK * (t + 1));
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wm + 16 * a + 4 * kq + r;
      if (row >= M) continue;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int col = n0 + wn + 16 * b + i;
        if (col < N) Y[(size_t)(slot0 + row) * N + col] = acc[a][b][r];
      }
    }
}

// dW[rel][m0.., n0..] += sum over the item's slots of Xs[src[slot], m] * val[slot] * G[dst[slot], n]
template <bool VEC>      // 16-byte loads of both gathered rows (both widths multiples of 4): compile-time, see gemm_kernel
__global__ __launch_bounds__(WG) void rel_wgrad_kernel(
    const float *__restrict__ Xs, const float *__restrict__ G, float *__restrict__ dW /* [R][Mw][Nw] */,
    const int *__restrict__ p_src, const int *__restrict__ p_dst, const float *__restrict__ p_val,
    const int *__restrict__ chunk_rel, const int2 *__restrict__ items, int Mw, int Nw, int tiles_m) {
  __shared__ __attribute__((aligned(16))) float sA[2][GK * LD_OUT];
  __shared__ __attribute__((aligned(16))) float sB[2][GK * LD_OUT];
  const int2 range = items[blockIdx.x];
  if (range.x >= range.y) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i = lane & 15, kq = lane >> 4;
  const int rel = chunk_rel[range.x];
  const int slot0 = range.x * RGCN_CHUNK, Ks = (range.y - range.x) * RGCN_CHUNK;        // K = the item's slots
  const int m0 = (blockIdx.y % tiles_m) * GT, n0 = (blockIdx.y / tiles_m) * GT;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr bool vec_a = VEC, vec_b = VEC;
  f32x4 ra[2], rb[2];
  float sc[2];
  auto fetch = [&](int k0) {     // thread -> slots k0 + (tid >> 5) (+ 8), column group 4 (tid & 31) of both gathered rows
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int kk = k0 + (tid >> 5) + 8 * h;
      const bool on = kk < Ks;
      const int slot = slot0 + (on ? kk : 0);
      const float v = on ? p_val[slot] : 0.f;
      sc[h] = v;
      const float *xr = Xs + (size_t)p_src[slot] * Mw, *gr = G + (size_t)max(p_dst[slot], 0) * Nw;   // pads: dst = -1, val = 0
      const int ca = m0 + 4 * (tid & 31), cb = n0 + 4 * (tid & 31);
      if (vec_a) {
        ra[h] = *reinterpret_cast<const f32x4 *>(xr + min(ca, Mw - 4));
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) ra[h][c] = xr[min(ca + c, Mw - 1)];
      }
      if (vec_b) {
        rb[h] = *reinterpret_cast<const f32x4 *>(gr + min(cb, Nw - 4));
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) rb[h][c] = gr[min(cb + c, Nw - 1)];
      }
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool live = sc[h] != 0.f;
      *reinterpret_cast<f32x4 *>(&sA[buf][((tid >> 5) + 8 * h) * LD_OUT + 4 * (tid & 31)]) = live ? ra[h] : f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4 *>(&sB[buf][((tid >> 5) + 8 * h) * LD_OUT + 4 * (tid & 31)]) = live ? rb[h] * sc[h] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  const int steps = (Ks + GK - 1) / GK;
  fetch(0);
  stash(0);
  __syncthreads();
  for (int t = 0; t < steps; ++t) {
    const int cur = t & 1;
    if (t + 1 < steps) fetch(GK * (t + 1));
    f32x4 av[4], bv[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        av[a][c] = sA[cur][(4 * kq + c) * LD_OUT + wm + 16 * a + i];
        bv[a][c] = sB[cur][(4 * kq + c) * LD_OUT + wn + 16 * a + i];
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a][c], bv[b][c], acc[a][b], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < steps) stash(cur ^ 1);
    __syncthreads();
  }
  float *Wr = dW + (size_t)rel * Mw * Nw;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + wm + 16 * a + 4 * kq + r;
      if (row >= Mw) continue;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int col = n0 + wn + 16 * b + i;
        if (col < Nw) atomicAdd(&Wr[(size_t)row * Nw + col], acc[a][b][r]);
      }
    }
}

// out[row, :] = bias + sum_{j in rowptr[row] .. rowptr[row+1]} Y[perm[j], :]   (rows of any width; one wave per row)
// The sums are carried in DOUBLES and rounded once: a destination with thousands of messages (a hub, or one edge repeated 9,000 times)
// is a chain of that many additions per lane, and in fp32 a chain of EQUAL terms drifts by the same rounding residue at every step
// (1.5e-4 relative at 18,001 terms, found by the randomised sweep).  The kernel waits for memory either way (v_add_f64 is full rate).
__global__ __launch_bounds__(WG) void segment_gather_sum_wide_kernel(const float *__restrict__ Y, const int *__restrict__ perm,
                                                                     const int *__restrict__ rowptr, const float *__restrict__ bias,
                                                                     float *__restrict__ out, long long n_rows, int d, int relu_out) {
  const int lane = threadIdx.x & 63;
  const long long wave0 = ((long long)blockIdx.x * WG + threadIdx.x) >> 6, nw = ((long long)gridDim.x * WG) >> 6;
  const bool vec = (d & 3) == 0;
  for (long long row = wave0; row < n_rows; row += nw) {
    const int e0 = rowptr[row], e1 = rowptr[row + 1];
    for (int f0 = 0; f0 < d; f0 += 256) {
      const int f = f0 + 4 * lane;
      double sa[4], sb[4] = {0., 0., 0., 0.};
      if (f < d) {
#pragma unroll
        for (int q = 0; q < 4; ++q) sa[q] = (bias && f + q < d) ? (double)bias[f + q] : 0.;
        for (int e = e0; e < e1; e += 2) {
          const float *ya = Y + (size_t)perm[e] * d, *yb = Y + (size_t)perm[min(e + 1, e1 - 1)] * d;
          const bool two = e + 1 < e1;
          if (vec) {
            const f32x4 va = *reinterpret_cast<const f32x4 *>(ya + f), vb = *reinterpret_cast<const f32x4 *>(yb + f);
#pragma unroll
            for (int q = 0; q < 4; ++q) { sa[q] += (double)va[q]; sb[q] += two ? (double)vb[q] : 0.; }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (f + q < d) { sa[q] += (double)ya[f + q]; sb[q] += two ? (double)yb[f + q] : 0.; }
          }
        }
        f32x4 a;
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = (float)(sa[q] + sb[q]);
        if (relu_out) {
#pragma unroll
          for (int q = 0; q < 4; ++q) a[q] = fmaxf(a[q], 0.f);
        }
        float *o = out + (size_t)row * d + f;
        if (vec) {
          *reinterpret_cast<f32x4 *>(o) = a;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (f + q < d) o[q] = a[q];
        }
      }
    }
  }
}

}  // namespace

extern "C" int64_t rgcn_gemm_scratch_floats(int64_t M, int64_t N, int64_t K, int32_t split_k) {
  (void)K;
  return split_k > 1 ? (int64_t)split_k * M * N : 0;
}

extern "C" int rgcn_gemm_f32(const float *A, const float *B, const float *bias, float *C, float *scratch, int64_t M,
                             int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int32_t flags, int32_t split_k,
                             void *stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K < 0 || M > INT32_MAX || N > INT32_MAX || K > INT32_MAX || split_k < 1 ||
      (split_k > 1 && !scratch)) {
    rgcn_set_error("gemm: bad argument");
    return RGCN_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  int tiles_m = (int)((M + GT - 1) / GT);
  const int tiles_n = (int)((N + GT - 1) / GT);
  int S = (int)std::min<int64_t>(split_k, std::max<int64_t>(1, (K + GK - 1) / GK));
  const int kps = (int)(((K + S - 1) / S + GK - 1) / GK * GK);
  S = K > 0 ? (int)((K + kps - 1) / kps) : 1;
  float *out = S > 1 ? scratch : C;
  const int sum_vec = (N & 3) == 0 && (ldc & 3) == 0 &&
                      ((reinterpret_cast<uintptr_t>(scratch) | reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0;
  const long long ldo = S > 1 ? N : ldc, sstride = S > 1 ? (long long)M * N : 0;
  // 64-row tiles when they spread better over the 256 CUs: rounds of 128-row tile work on the busiest CU
  const int bm_env = rgcn_option_value(RGCN_OPT_GEMM_BM);
  const int64_t t128 = (int64_t)tiles_m * tiles_n * S, t64 = ((M + 63) / 64) * tiles_n * S;
  const bool half = bm_env ? bm_env == 64 : ((t64 + 255) / 256 < 2 * ((t128 + 255) / 256) && t128 > 256);
  if (half) tiles_m = (int)((M + 63) / 64);
  const bool ta = flags & RGCN_G_TRANS_A, tb = flags & RGCN_G_TRANS_B;
  // 16-byte loads when rows / columns are 16-byte aligned (the contiguous extent and the leading dimension multiples of 4): the
  // K tail is then whole groups of 4 and is zeroed by the stash; otherwise element loads with clamped indices
  const bool vec = ((lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && (((ta ? M : K) & 3) == 0) &&
                   ((ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && (((tb ? K : N) & 3) == 0);
  // row panels (64 rows x up to 208 columns) when the 128-wide tiles would compute noticeably more columns than the panels: N = 200 is
  // 208 instead of 256, 400 is 416 instead of 512, a hidden width of 16 is 64 instead of 128
  if (vec && !bm_env && M >= 4 && N >= 4 && K >= 4) {
    const int nt = (int)((N + 15) / 16), n_panels = (nt + 12) / 13, need = (nt + n_panels - 1) / n_panels;
    const int nb = need <= 4 ? 4 : need <= 7 ? 7 : need <= 10 ? 10 : 13;
    if ((int64_t)n_panels * nb * 16 * 100 < (int64_t)tiles_n * GT * 95) {
      const int pm = (int)((M + 63) / 64);
      dim3 pgrid((unsigned)(pm * n_panels), (unsigned)S);
#define RGCN_PANEL_ARGS pgrid, dim3(WG), 0, st, A, B, bias, out, (int)M, (int)N, (int)K, (long long)lda, (long long)ldb, ldo, pm, kps > 0 ? kps : GK, sstride
#define RGCN_PANEL_NB(TAc, TBc)                                                                          \
  do {                                                                                                   \
    if (nb == 4) hipLaunchKernelGGL((gemm_panel_kernel<TAc, TBc, 4>), RGCN_PANEL_ARGS);                  \
    else if (nb == 7) hipLaunchKernelGGL((gemm_panel_kernel<TAc, TBc, 7>), RGCN_PANEL_ARGS);             \
    else if (nb == 10) hipLaunchKernelGGL((gemm_panel_kernel<TAc, TBc, 10>), RGCN_PANEL_ARGS);           \
    else hipLaunchKernelGGL((gemm_panel_kernel<TAc, TBc, 13>), RGCN_PANEL_ARGS);                         \
  } while (0)
      if (ta && tb) RGCN_PANEL_NB(true, true);
      else if (ta) RGCN_PANEL_NB(true, false);
      else if (tb) RGCN_PANEL_NB(false, true);
      else RGCN_PANEL_NB(false, false);
#undef RGCN_PANEL_NB
#undef RGCN_PANEL_ARGS
      if (S > 1) {
        const long long n = (long long)M * N;
        hipLaunchKernelGGL(sum_slices_kernel, dim3((unsigned)std::min<long long>((n + WG - 1) / WG, 4096)), dim3(WG), 0, st, scratch,
                           bias, C, n, (int)N, S, (long long)ldc, sum_vec);
      }
      HIP_TRY(hipGetLastError());
      return RGCN_OK;
    }
  }
  dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)S), block(WG);
#define RGCN_GEMM_ARGS grid, block, 0, st, A, B, bias, out, (int)M, (int)N, (int)K, (long long)lda, (long long)ldb, ldo, tiles_m, kps > 0 ? kps : GK, sstride
#define RGCN_GEMM_LAUNCH(TAc, TBc)                                                                  \
  do {                                                                                               \
    if (vec && half) hipLaunchKernelGGL((gemm_kernel<TAc, TBc, true, 64>), RGCN_GEMM_ARGS);          \
    else if (vec) hipLaunchKernelGGL((gemm_kernel<TAc, TBc, true, 128>), RGCN_GEMM_ARGS);            \
    else if (half) hipLaunchKernelGGL((gemm_kernel<TAc, TBc, false, 64>), RGCN_GEMM_ARGS);           \
    else hipLaunchKernelGGL((gemm_kernel<TAc, TBc, false, 128>), RGCN_GEMM_ARGS);                    \
  } while (0)
  if (ta && tb) RGCN_GEMM_LAUNCH(true, true);
  else if (ta) RGCN_GEMM_LAUNCH(true, false);
  else if (tb) RGCN_GEMM_LAUNCH(false, true);
  else RGCN_GEMM_LAUNCH(false, false);
#undef RGCN_GEMM_LAUNCH
#undef RGCN_GEMM_ARGS
  if (S > 1) {
    const long long n = (long long)M * N;
    hipLaunchKernelGGL(sum_slices_kernel, dim3((unsigned)std::min<long long>((n + WG - 1) / WG, 4096)), dim3(WG), 0, st, scratch,
                       bias, C, n, (int)N, S, (long long)ldc, sum_vec);
  }
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_rel_rows_f32(const float *Xs, const float *W, float *Y, const int32_t *p_src, const float *p_val,
                                 const int32_t *chunk_rel, const int32_t *items, int64_t n_items, int32_t R, int32_t d_in,
                                 int32_t d_out, void *stream) {
  (void)R;
  if (!Xs || !W || !Y || n_items < 0 || d_in <= 0 || d_out <= 0 || (n_items && (!p_src || !p_val || !chunk_rel || !items))) {
    rgcn_set_error("rel_rows: bad argument");
    return RGCN_EINVAL;
  }
  if (!n_items) return RGCN_OK;
  if (((d_in | d_out) & 3) == 0) {
    hipLaunchKernelGGL(rel_rows_kernel<true>, dim3((unsigned)n_items, (unsigned)((d_out + GT - 1) / GT)), dim3(WG), 0, (hipStream_t)stream,
                     Xs, W, Y, p_src, p_val, chunk_rel, reinterpret_cast<const int2 *>(items), d_in, d_out);
  } else {
    hipLaunchKernelGGL(rel_rows_kernel<false>, dim3((unsigned)n_items, (unsigned)((d_out + GT - 1) / GT)), dim3(WG), 0, (hipStream_t)stream,
                     Xs, W, Y, p_src, p_val, chunk_rel, reinterpret_cast<const int2 *>(items), d_in, d_out);
  }
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_rel_wgrad_f32(const float *Xs, const float *G, float *dW, const int32_t *p_src, const int32_t *p_dst,
                                  const float *p_val, const int32_t *chunk_rel, const int32_t *items, int64_t n_items,
                                  int32_t R, int32_t d_in, int32_t d_out, void *stream) {
  if (!Xs || !G || !dW || n_items < 0 || R <= 0 || d_in <= 0 || d_out <= 0 || (n_items && (!p_src || !p_dst || !p_val || !chunk_rel || !items))) {
    rgcn_set_error("rel_wgrad: bad argument");
    return RGCN_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(zero_async(dW, (size_t)R * d_in * d_out * sizeof(float), st));
  if (!n_items) return RGCN_OK;
  const int tiles_m = (d_in + GT - 1) / GT, tiles_n = (d_out + GT - 1) / GT;
  if (((d_in | d_out) & 3) == 0) {
    hipLaunchKernelGGL(rel_wgrad_kernel<true>, dim3((unsigned)n_items, (unsigned)(tiles_m * tiles_n)), dim3(WG), 0, st, Xs, G, dW, p_src,
                     p_dst, p_val, chunk_rel, reinterpret_cast<const int2 *>(items), d_in, d_out, tiles_m);
  } else {
    hipLaunchKernelGGL(rel_wgrad_kernel<false>, dim3((unsigned)n_items, (unsigned)(tiles_m * tiles_n)), dim3(WG), 0, st, Xs, G, dW, p_src,
                     p_dst, p_val, chunk_rel, reinterpret_cast<const int2 *>(items), d_in, d_out, tiles_m);
  }
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_segment_gather_sum_wide_f32(const float *Y, const int32_t *perm, const int32_t *rowptr, const float *bias,
                                                float *out, int64_t n_rows, int32_t d, int32_t flags, void *stream) {
  if (!Y || !perm || !rowptr || !out || n_rows < 0 || d <= 0) { rgcn_set_error("segment_gather_sum_wide: bad argument"); return RGCN_EINVAL; }
  if (!n_rows) return RGCN_OK;
  const unsigned gx = (unsigned)std::min<int64_t>((n_rows + 3) / 4, 256 * 32);
  hipLaunchKernelGGL(segment_gather_sum_wide_kernel, dim3(gx), dim3(WG), 0, (hipStream_t)stream, Y, perm, rowptr, bias, out,
                     (long long)n_rows, d, flags & RGCN_F_RELU);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}
