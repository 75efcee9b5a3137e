// Device-side graph preparation: the per-call work of the link-prediction
// layer -- augmentation, literal normalisation, relation-tile plan -- without leaving the GPU.
// Reference lines taken over: utils.py:100-124 / layers.py:481-487 (augmentation), utils.py:143-166 +
// :71-97 + layers.py:498-510 (stack, degree count, swap, 1/c), layers.py:513-516 (sparse ctor).
//
// Method: dense counting.  With 288 GB of HBM a table with one int per (tile, relation, row) cell
// (= N_pad x R ints: 1.5 M for WN18, 101 M for S1, 445 M for AM) is affordable, so both the degree count
// and the bucket sort are ONE pass of fp-free integer atomics over the messages plus scans over the table:
//     cells[cell(e)]++                       cell = ((dst / T) * R + rel) * T + dst % T
//     per bucket: size, exclusive offsets of its T cells, size rounded up to 16
//     exclusive scan over the buckets -> first slot of every bucket (-> tile_ptr, run_ptr, chunk_rel)
//     slot(e) = bucket_base + atomicAdd(&cells[cell(e)], 1)   (cells now hold running offsets)
// The order of messages inside one (relation, destination) cell is arbitrary (they are summed anyway).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

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

constexpr int TB = 256;
inline unsigned blocks_for(int64_t n, int per = TB) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + per - 1) / per, 1 << 20)); }

// ------------------------------------------------------------------ message lists
__global__ void split_triples_kernel(const long long *__restrict__ tp, long long M, long long N, int R,
                                     int *__restrict__ s, int *__restrict__ p, int *__restrict__ o,
                                     int *__restrict__ err) {
  for (long long e = (long long)blockIdx.x * TB + threadIdx.x; e < M; e += (long long)gridDim.x * TB) {
    long long a = tp[3 * e], b = tp[3 * e + 1], c = tp[3 * e + 2];
    // an id out of range raises the flag (the caller turns it into the reference's AssertionError) AND is clamped: with deferred
    // checks the flag is looked at after the step, and nothing downstream (degree tables, plan counters) may index with it
    if (a < 0 || a >= N || c < 0 || c >= N || b < 0 || b >= R) { atomicMax(err, 1); a = b = c = 0; }
    s[e] = (int)a; p[e] = (int)b; o[e] = (int)c;
  }
}

// [T | inverse(T) | T | self loops]; dropped self loops stay in the list but are marked dead
__global__ void lp_expand_kernel(const long long *__restrict__ t, long long E, long long N, int R0,
                                 const unsigned char *__restrict__ keep, int *__restrict__ s, int *__restrict__ p,
                                 int *__restrict__ o, unsigned char *__restrict__ alive, int *__restrict__ err) {
  const long long M = 3 * E + N;
  for (long long e = (long long)blockIdx.x * TB + threadIdx.x; e < M; e += (long long)gridDim.x * TB) {
    if (e < 3 * E) {
      const long long j = e % E, blk = e / E;
      long long a = t[3 * j], b = t[3 * j + 1], c = t[3 * j + 2];
      // a bad triple raises the flag, is clamped to (0, 0, 0) and marked DEAD: in deferred-check mode (RGCN_DEFERRED_CHECKS=1, hipGraph
      // replays) the flag is only read after the step, so the message must not reach the degree tables / plan counters as it is
      const bool bad = a < 0 || a >= N || c < 0 || c >= N || b < 0 || b >= R0;
      if (bad) { atomicMax(err, 1); a = b = c = 0; }
      if (blk == 1) { s[e] = (int)c; p[e] = (int)b + R0; o[e] = (int)a; }
      else { s[e] = (int)a; p[e] = (int)b; o[e] = (int)c; }
      alive[e] = bad ? 0 : 1;
    } else {
      const long long n = e - 3 * E;
      s[e] = (int)n; p[e] = 2 * R0; o[e] = (int)n;
      alive[e] = keep ? (keep[n] != 0) : 1;
    }
  }
}

// ------------------------------------------------------------------ literal normalisation
__global__ void norm_count_kernel(const int *__restrict__ s, const int *__restrict__ p, const int *__restrict__ o,
                                  const unsigned char *__restrict__ alive, long long M, long long N, int vertical,
                                  int *__restrict__ table) {
  for (long long e = (long long)blockIdx.x * TB + threadIdx.x; e < M; e += (long long)gridDim.x * TB) {
    if (alive && !alive[e]) continue;
    atomicAdd(&table[(long long)p[e] * N + (vertical ? s[e] : o[e])], 1);
  }
}

// vertical: val = 1 / count(p, s).  horizontal: the count of the PARTNER position after the block swap
// c = [k[n:2n] | k[0:n] | k[2n:]] (third block maps to itself because 2n + i = length of the list).
__global__ void norm_val_kernel(const int *__restrict__ s, const int *__restrict__ p, const int *__restrict__ o,
                                const unsigned char *__restrict__ alive, long long M, long long N, int vertical,
                                long long n_swap, const int *__restrict__ table, float *__restrict__ val) {
  for (long long e = (long long)blockIdx.x * TB + threadIdx.x; e < M; e += (long long)gridDim.x * TB) {
    if (alive && !alive[e]) { val[e] = 0.f; continue; }
    long long q = e;
    if (!vertical) q = e < n_swap ? e + n_swap : (e < 2 * n_swap ? e - n_swap : e);
    const int c = table[(long long)p[q] * N + (vertical ? s[q] : o[q])];
    val[e] = 1.0f / (float)c;
  }
}

// ------------------------------------------------------------------ plan: counting sort on a dense cell table
__device__ __forceinline__ long long cell_of(int d, int r, int R, int T) {
  return ((long long)(d / T) * R + r) * T + d % T;
}

__global__ void cell_count_kernel(const int *__restrict__ dst, const int *__restrict__ rel,
                                  const unsigned char *__restrict__ alive, long long M, int R, int T,
                                  int *__restrict__ cells) {
  for (long long e = (long long)blockIdx.x * TB + threadIdx.x; e < M; e += (long long)gridDim.x * TB) {
    if (alive && !alive[e]) continue;
    atomicAdd(&cells[cell_of(dst[e], rel ? rel[e] : 0, R, T)], 1);
  }
}

// one wave per bucket: bucket size, cells -> exclusive offsets inside the bucket, padded size
__global__ void bucket_scan_kernel(int *__restrict__ cells, long long n_buckets, int T, int *__restrict__ bucket_cnt,
                                   int *__restrict__ bucket_pad) {
  const int lane = threadIdx.x & 63;
  const long long wave0 = ((long long)blockIdx.x * TB + threadIdx.x) >> 6, nw = ((long long)gridDim.x * TB) >> 6;
  for (long long b = wave0; b < n_buckets; b += nw) {
    int *c = cells + b * T;
    int run = 0;
    for (int base = 0; base < T; base += 64) {
      const int i = base + lane;
      const int v = i < T ? c[i] : 0;
      int incl = v;                                  // inclusive scan across the wave
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
      }
      if (i < T) c[i] = run + incl - v;
      run += __shfl(incl, 63, 64);
    }
    if (lane == 0) {
      bucket_cnt[b] = run;
      bucket_pad[b] = (run + RGCN_CHUNK - 1) / RGCN_CHUNK * RGCN_CHUNK;
    }
  }
}

// Few, very tall buckets (relation-major plan: T = N; CSR: one bucket): a wave per bucket would walk millions of
// cells serially, so scan ALL cells at once (G) and take the bucket-local offsets as differences.
// (one launch: cells -> bucket-local offsets, and the bucket sizes / padded sizes)
__global__ void from_global_scan_kernel(const int *__restrict__ G, const int *__restrict__ total, int *__restrict__ cells,
                                        long long n_cells, long long n_buckets, int T, int *__restrict__ bucket_cnt,
                                        int *__restrict__ bucket_pad) {
  const long long i0 = (long long)blockIdx.x * TB + threadIdx.x, stride = (long long)gridDim.x * TB;
  for (long long i = i0; i < n_cells; i += stride) cells[i] = G[i] - G[(i / T) * T];
  for (long long b = i0; b < n_buckets; b += stride) {
    const int end = b + 1 < n_buckets ? G[(b + 1) * T] : *total;
    const int cnt = end - G[b * T];
    bucket_cnt[b] = cnt;
    bucket_pad[b] = (cnt + RGCN_CHUNK - 1) / RGCN_CHUNK * RGCN_CHUNK;
  }
}

// chunk -> relation of its bucket (binary search over the bucket bases)
__global__ void chunk_rel_kernel(const int *__restrict__ bucket_base, long long n_buckets, int R, long long n_chunks,
                                 int *__restrict__ chunk_rel) {
  for (long long c = (long long)blockIdx.x * TB + threadIdx.x; c < n_chunks; c += (long long)gridDim.x * TB) {
    const int slot = (int)(c * RGCN_CHUNK);
    long long lo = 0, hi = n_buckets;            // last bucket with base <= slot
    while (hi - lo > 1) {
      const long long mid = (lo + hi) >> 1;
      if (bucket_base[mid] <= slot) lo = mid; else hi = mid;
    }
    // empty buckets share a base with the next non-empty one: move to the last bucket having this base
    while (lo + 1 < n_buckets && bucket_base[lo + 1] <= slot) ++lo;
    chunk_rel[c] = (int)(lo % R);
  }
}

// exclusive scan of n ints, three launches: per-block (1024 items) scan + totals, scan of totals, add back
__global__ __launch_bounds__(TB) void scan_blocks_kernel(const int *in, int *out, int *__restrict__ totals, long long n) {  // in may alias out
  __shared__ int wsum[TB / 64];
  const long long base = (long long)blockIdx.x * (TB * 4);
  int v[4], sum = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long i = base + (long long)threadIdx.x * 4 + j;
    v[j] = i < n ? in[i] : 0;
    sum += v[j];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < wave; ++w) woff += wsum[w];
  int excl = woff + incl - sum;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long i = base + (long long)threadIdx.x * 4 + j;
    if (i < n) out[i] = excl;
    excl += v[j];
  }
  if (threadIdx.x == TB - 1) totals[blockIdx.x] = woff + incl;
}

__global__ __launch_bounds__(TB) void scan_totals_kernel(int *__restrict__ totals, long long nb, int *__restrict__ grand) {
  // single block, sequential over chunks of TB
  __shared__ int wsum[TB / 64];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (long long base = 0; base < nb; base += TB) {
    const long long i = base + threadIdx.x;
    const int v = i < nb ? totals[i] : 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = carry_s;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    if (i < nb) totals[i] = woff + incl - v;
    __syncthreads();
    if (threadIdx.x == TB - 1) carry_s = woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *grand = carry_s;
}

__global__ void scan_add_kernel(int *__restrict__ out, const int *__restrict__ totals, long long n) {
  for (long long i = (long long)blockIdx.x * TB + threadIdx.x; i < n; i += (long long)gridDim.x * TB)
    out[i] += totals[i / (TB * 4)];
}

__global__ void plan_scatter_kernel(const int *__restrict__ dst, const int *__restrict__ src, const int *__restrict__ rel,
                                    const float *__restrict__ val, const unsigned char *__restrict__ alive, long long M,
                                    int R, int T, int *__restrict__ cells, const int *__restrict__ bucket_base,
                                    int *__restrict__ p_src, int *__restrict__ p_dst, float *__restrict__ p_val,
                                    int2 *__restrict__ p_pack, const int *__restrict__ aux, int *__restrict__ p_aux,
                                    int *__restrict__ msg_slot) {
  for (long long e = (long long)blockIdx.x * TB + threadIdx.x; e < M; e += (long long)gridDim.x * TB) {
    if (alive && !alive[e]) continue;
    const int d = dst[e], r = rel ? rel[e] : 0;           // rel == NULL: one relation (CSR builds)
    const long long bucket = (long long)(d / T) * R + r;
    const int pos = bucket_base[bucket] + atomicAdd(&cells[bucket * T + d % T], 1);
    p_src[pos] = src[e];
    p_dst[pos] = d;
    p_val[pos] = val[e];
    if (p_pack) p_pack[pos] = make_int2((int)((unsigned)src[e] | ((unsigned)(d % T) << 24)), __builtin_bit_cast(int, val[e]));
    if (p_aux) p_aux[pos] = aux[e];
    if (msg_slot) msg_slot[e] = pos;
  }
}

// pads, chunk relations, tile / run pointers: one thread per bucket
__global__ void plan_finish_kernel(long long n_buckets, int R, const int *__restrict__ bucket_cnt,
                                   const int *__restrict__ bucket_base, int *__restrict__ p_src, int *__restrict__ p_dst,
                                   float *__restrict__ p_val, int2 *__restrict__ p_pack, int *__restrict__ chunk_rel,
                                   int *__restrict__ tile_ptr, int *__restrict__ run_ptr) {
  for (long long b = (long long)blockIdx.x * TB + threadIdx.x; b <= n_buckets; b += (long long)gridDim.x * TB) {
    const int base = bucket_base[b];
    const long long t = b / R;
    const int r = (int)(b % R);
    if (r == 0) tile_ptr[t] = base / RGCN_CHUNK;          // also writes tile_ptr[n_tiles] for b == n_buckets
    if (b == n_buckets) {
      if (run_ptr) run_ptr[(t - 1) * (R + 1) + R] = base / RGCN_CHUNK;
      break;
    }
    if (run_ptr) {
      run_ptr[t * (R + 1) + r] = base / RGCN_CHUNK;
      if (r == R - 1) run_ptr[t * (R + 1) + R] = bucket_base[b + 1] / RGCN_CHUNK;
    }
    const int cnt = bucket_cnt[b], end = bucket_base[b + 1];
    (void)chunk_rel;
    if (!cnt) continue;
    const int last_src = p_src[base + cnt - 1];
    for (int q = base + cnt; q < end; ++q) {
      p_src[q] = last_src;
      p_dst[q] = -1;
      p_val[q] = 0.f;
      if (p_pack) p_pack[q] = make_int2((int)((unsigned)last_src | (0xFFu << 24)), 0);
    }
  }
}

// ---- sync-free completion of a plan whose arrays were sized by an UPPER BOUND (the per-call graphs of the link-prediction
// layer: nothing may be read back).  The real padded size lives in bucket_base[n_buckets] on the device.
// slots past the real end become pads (val = 0, dst = -1): every consumer kernel may walk them
__global__ void plan_tail_kernel(const int *__restrict__ bucket_base, long long n_buckets, long long m_pad_ub,
                                 int *__restrict__ p_src, int *__restrict__ p_dst, float *__restrict__ p_val,
                                 int2 *__restrict__ p_pack, int *__restrict__ p_aux) {
  const long long real = bucket_base[n_buckets];
  for (long long q = real + (long long)blockIdx.x * TB + threadIdx.x; q < m_pad_ub; q += (long long)gridDim.x * TB) {
    p_src[q] = 0;
    p_dst[q] = -1;
    p_val[q] = 0.f;
    if (p_pack) p_pack[q] = make_int2((int)(0xFFu << 24), 0);
    if (p_aux) p_aux[q] = 0;
  }
}

// one work unit per destination tile (no hub splitting: the tile sizes are not known on the host)
__global__ void plan_units_kernel(const int *__restrict__ tile_ptr, long long n_tiles, int4 *__restrict__ units) {
  for (long long t = (long long)blockIdx.x * TB + threadIdx.x; t < n_tiles; t += (long long)gridDim.x * TB)
    units[t] = make_int4((int)t, tile_ptr[t], tile_ptr[t + 1], 0);
}

// relation-major plan (one tile): work items = chunk ranges of ONE relation, at most max_item_chunks long; the list is
// padded with empty items {0, 0} up to n_items_ub (consumers return at once on an empty item).  Single workgroup.
__global__ __launch_bounds__(TB) void plan_items_kernel(const int *__restrict__ bucket_base, int R, int max_item_chunks,
                                                        int2 *__restrict__ items, long long n_items_ub) {
  __shared__ int offs[TB + 1];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int r0 = 0; r0 < R; r0 += TB) {
    const int r = r0 + threadIdx.x;
    const int c0 = r < R ? bucket_base[r] / RGCN_CHUNK : 0, c1 = r < R ? bucket_base[r + 1] / RGCN_CHUNK : 0;
    const int pieces = (c1 - c0 + max_item_chunks - 1) / max_item_chunks;
    offs[threadIdx.x + 1] = pieces;
    if (threadIdx.x == 0) offs[0] = 0;
    __syncthreads();
    if (threadIdx.x == 0)
      for (int q = 1; q <= TB; ++q) offs[q] += offs[q - 1];      // 256 adds: the plan is built once per call
    __syncthreads();
    const int first = carry + offs[threadIdx.x];
    for (int q = 0; q < pieces; ++q)
      if (first + q < n_items_ub) items[first + q] = make_int2(c0 + q * max_item_chunks, min(c1, c0 + (q + 1) * max_item_chunks));
    __syncthreads();
    if (threadIdx.x == 0) carry += offs[TB];
    __syncthreads();
  }
  for (long long q = carry + threadIdx.x; q < n_items_ub; q += TB) items[q] = make_int2(0, 0);
}

// the last two steps of the scan in one launch when there are few blocks: workgroup b sums the totals of the blocks before it
// (<= 1024 values) instead of reading a scanned copy, adds the offset to its 1024 items; the last one writes the grand total
__global__ __launch_bounds__(TB) void scan_add_small_kernel(int *__restrict__ out, const int *__restrict__ totals, long long n,
                                                            int nb, int *__restrict__ grand) {
  __shared__ int wsum[TB / 64];
  const int b = blockIdx.x;
  int part = 0;
  for (int j = threadIdx.x; j < b; j += TB) part += totals[j];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = part;
  __syncthreads();
  int off0 = 0;
  for (int w = 0; w < TB / 64; ++w) off0 += wsum[w];
  const long long base = (long long)b * (TB * 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long i = base + (long long)threadIdx.x * 4 + j;
    if (i < n) out[i] += off0;
  }
  if (b == nb - 1 && threadIdx.x == 0) *grand = off0 + totals[b];
}

// exclusive scan of up to 1024 ints in place, total to v[n]: one workgroup, one launch (bucket bases of single-tile plans: a
// CSR has one bucket, a relation-major plan R)
__global__ __launch_bounds__(1024) void scan_small_kernel(int *__restrict__ v, int n) {
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x = tid < n ? v[tid] : 0;
  int incl = x;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int woff = 0, total = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wave) woff += wsum[w];
    total += wsum[w];
  }
  if (tid < n) v[tid] = woff + incl - x;
  if (tid == 0) v[n] = total;
}

int exclusive_scan(const int *in, int *out, int *totals, long long n, int *grand, hipStream_t st) {
  const long long nb = (n + TB * 4 - 1) / (TB * 4);
  hipLaunchKernelGGL(scan_blocks_kernel, dim3((unsigned)nb), dim3(TB), 0, st, in, out, totals, n);
  if (nb <= 1024) {        // (the per-call graphs of the LP layer: a step builds five plans, every launch counts)
    hipLaunchKernelGGL(scan_add_small_kernel, dim3((unsigned)nb), dim3(TB), 0, st, out, totals, n, (int)nb, grand);
  } else {
    hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(TB), 0, st, totals, nb, grand);
    hipLaunchKernelGGL(scan_add_kernel, dim3(blocks_for(n)), dim3(TB), 0, st, out, totals, n);
  }
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

// ---- a PAIR of CSRs of one message list (rows = a with entries b, and rows = b with entries a), nothing else: no buckets,
// pads, tiles or chunk relations.  The basis / diagonal / block kernels and the DistMult backward walk exactly these two
// structures; through the general plan builder they cost 16 launches per pair, here 5: zero, count both keys, scan (2), scatter.
// Layout: one row buffer R of 2 N + 2 ints = [0 | rows of the first CSR | rows of the second CSR shifted by one dummy row]; the
// count pass leaves the row sizes in R[1..], the scan turns them into offsets, the scatter advances every counter to the end of
// its row = the start of the next: afterwards R[0 .. N] is the first row pointer and R[N + 1 .. 2 N + 1] the second (its entries
// live behind the first CSR's in the SAME entry arrays, so its offsets need no rebasing).
__global__ void csr2_count_kernel(const int *__restrict__ a, const int *__restrict__ b, const unsigned char *__restrict__ alive,
                                  long long M, long long N, int *__restrict__ R) {
  for (long long e = (long long)blockIdx.x * TB + threadIdx.x; e < M; e += (long long)gridDim.x * TB) {
    if (alive && !alive[e]) continue;
    atomicAdd(&R[1 + a[e]], 1);
    atomicAdd(&R[1 + N + 1 + b[e]], 1);
  }
}

__global__ void csr2_scatter_kernel(const int *__restrict__ a, const int *__restrict__ b, const int *__restrict__ rel,
                                    const float *__restrict__ val, const unsigned char *__restrict__ alive, long long M,
                                    long long N, int *__restrict__ R, int *__restrict__ e_other, int *__restrict__ e_rel,
                                    float *__restrict__ e_val) {
  for (long long e = (long long)blockIdx.x * TB + threadIdx.x; e < M; e += (long long)gridDim.x * TB) {
    if (alive && !alive[e]) continue;
    const int ka = a[e], kb = b[e], r = rel ? rel[e] : 0;
    const float v = val[e];
    const int pa = atomicAdd(&R[1 + ka], 1), pb = atomicAdd(&R[1 + N + 1 + kb], 1);
    e_other[pa] = kb; e_rel[pa] = r; e_val[pa] = v;
    e_other[pb] = ka; e_rel[pb] = r; e_val[pb] = v;
  }
}

// The two CSRs of a batch of SCORED triples once the scoring kernel has left every triple's rank within its subject row and its object
// row (distmult_fwd_kernel) and the row sizes: no atomics here, one pass.
__global__ void triple_csr_place_kernel(const long long *__restrict__ tr, long long T, long long N, int n_rel,
                                        const int *__restrict__ ranks, const int *__restrict__ C, const float *__restrict__ gs,
                                        int4 *__restrict__ entries) {
  for (long long t = (long long)blockIdx.x * TB + threadIdx.x; t < T; t += (long long)gridDim.x * TB) {
    const long long s = tr[3 * t], p = tr[3 * t + 1], o = tr[3 * t + 2];
    if (s < 0 || s >= N || o < 0 || o >= N || p < 0 || p >= n_rel) continue;       // the scoring kernel skipped (and flagged) these
    const int2 rk = *reinterpret_cast<const int2 *>(ranks + 2 * t);
    const float g = gs[t];
    const int ps = C[1 + s] + rk.x, po = C[N + 2 + o] + rk.y;
    entries[ps] = make_int4((int)o, (int)p, __float_as_int(g), 0);      // one 16-byte store per entry, one 16-byte load per lane later
    entries[po] = make_int4((int)s, (int)p, __float_as_int(g), 0);
  }
}

}  // namespace

extern "C" int rgcn_distmult_csr_place(const int64_t *triples, int64_t T, int64_t N, int32_t n_rel, const int32_t *ranks,
                                       int32_t *rank_counts, int32_t *scan_tmp, const float *gs, int32_t *entries, void *stream) {
  if (T < 0 || N <= 0 || n_rel <= 0 || !rank_counts || (T && (!triples || !ranks || !gs || !entries))) {
    rgcn_set_error("distmult_csr_place: bad argument");
    return RGCN_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  const long long n_cells = 2 * (long long)N + 2;                 // C[1 ..]: N subject rows, one dummy, N object rows, the end
  if (scan_tmp) {                                                 // (NULL: rank_counts was scanned by an earlier call -- a second backward pass)
    int rc = exclusive_scan(rank_counts + 1, rank_counts + 1, scan_tmp, n_cells, scan_tmp + (n_cells / (TB * 4) + 2), st);
    if (rc) return rc;
  }
  if (T) hipLaunchKernelGGL(triple_csr_place_kernel, dim3(blocks_for(T)), dim3(TB), 0, st, (const long long *)triples, (long long)T,
                            (long long)N, n_rel, ranks, rank_counts, gs, reinterpret_cast<int4 *>(entries));
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_dev_split_triples(const int64_t *triples_plus, int64_t M, int64_t N, int32_t R, int32_t *s, int32_t *p,
                                      int32_t *o, int32_t *err_flag, void *stream) {
  if (M < 0 || N <= 0 || R <= 0 || !err_flag || (M && (!triples_plus || !s || !p || !o))) { rgcn_set_error("dev_split_triples: bad argument"); return RGCN_EINVAL; }
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(zero_async(err_flag, sizeof(int), st));
  if (!M) return RGCN_OK;
  hipLaunchKernelGGL(split_triples_kernel, dim3(blocks_for(M)), dim3(TB), 0, st, (const long long *)triples_plus,
                     (long long)M, (long long)N, R, s, p, o, err_flag);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_dev_lp_expand(const int64_t *triples, int64_t E, int64_t N, int32_t R0, const uint8_t *keep, int32_t *s,
                                  int32_t *p, int32_t *o, uint8_t *alive, int32_t *err_flag, void *stream) {
  if (E < 0 || N <= 0 || R0 <= 0 || !s || !p || !o || !alive || !err_flag || (E && !triples)) { rgcn_set_error("dev_lp_expand: bad argument"); return RGCN_EINVAL; }
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(zero_async(err_flag, sizeof(int), st));
  hipLaunchKernelGGL(lp_expand_kernel, dim3(blocks_for(3 * E + N)), dim3(TB), 0, st, (const long long *)triples,
                     (long long)E, (long long)N, R0, keep, s, p, o, alive, err_flag);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_dev_edge_norm(const int32_t *s, const int32_t *p, const int32_t *o, const uint8_t *alive, int64_t M,
                                  int64_t N, int32_t R, int vertical, int64_t n_swap, int32_t *table, float *val,
                                  void *stream) {
  if (M < 0 || N <= 0 || R <= 0 || !table || (M && (!s || !p || !o || !val)) || n_swap < 0 || 2 * n_swap > M) { rgcn_set_error("dev_edge_norm: bad argument"); return RGCN_EINVAL; }
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(zero_async(table, (size_t)R * N * sizeof(int), st));
  if (!M) return RGCN_OK;
  hipLaunchKernelGGL(norm_count_kernel, dim3(blocks_for(M)), dim3(TB), 0, st, s, p, o, alive, (long long)M, (long long)N,
                     vertical, table);
  hipLaunchKernelGGL(norm_val_kernel, dim3(blocks_for(M)), dim3(TB), 0, st, s, p, o, alive, (long long)M, (long long)N,
                     vertical, (long long)n_swap, table, val);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_dev_plan_count(const int32_t *dst, const int32_t *rel, const uint8_t *alive, int64_t M, int64_t n_dst,
                                   int32_t R, int32_t tile_rows, int32_t *cells, int32_t *bucket_cnt,
                                   int32_t *bucket_base, int32_t *scan_tmp, int32_t *cells_tmp, void *stream) {
  if (M < 0 || n_dst <= 0 || R <= 0 || tile_rows <= 0 || !cells || !bucket_cnt || !bucket_base || !scan_tmp || (M && (!dst || (!rel && R != 1)))) { rgcn_set_error("dev_plan_count: bad argument"); return RGCN_EINVAL; }
  const int64_t n_tiles = (n_dst + tile_rows - 1) / tile_rows, nbk = n_tiles * R;
  if (nbk * tile_rows >= (int64_t(1) << 40)) { rgcn_set_error("dev_plan_count: cell table too large"); return RGCN_EUNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(zero_async(cells, (size_t)nbk * tile_rows * sizeof(int), st));
  if (M) hipLaunchKernelGGL(cell_count_kernel, dim3(blocks_for(M)), dim3(TB), 0, st, dst, rel, alive, (long long)M, R, tile_rows, cells);
  // bucket_base doubles as the padded-size array before the scan
  if (cells_tmp && tile_rows > 1024) {
    // scan_tmp must then hold (n_cells / 1024 + 2) + 1 ints: block totals and the grand total
    const long long n_cells = (long long)nbk * tile_rows;
    int *total = scan_tmp + (n_cells / (TB * 4) + 2);
    int rc = exclusive_scan(cells, cells_tmp, scan_tmp, n_cells, total, st);
    if (rc) return rc;
    hipLaunchKernelGGL(from_global_scan_kernel, dim3(blocks_for(n_cells)), dim3(TB), 0, st, cells_tmp, total, cells, n_cells,
                       (long long)nbk, tile_rows, bucket_cnt, bucket_base);
  } else {
    hipLaunchKernelGGL(bucket_scan_kernel, dim3(blocks_for(nbk * 64)), dim3(TB), 0, st, cells, (long long)nbk, tile_rows,
                       bucket_cnt, bucket_base);
  }
  HIP_TRY(hipGetLastError());
  if (nbk <= 1024) {
    hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(1024), 0, st, bucket_base, (int)nbk);
    HIP_TRY(hipGetLastError());
    return RGCN_OK;
  }
  return exclusive_scan(bucket_base, bucket_base, scan_tmp, nbk, bucket_base + nbk, st);
}

extern "C" int rgcn_dev_plan_fill(const int32_t *dst, const int32_t *src, const int32_t *rel, const float *val,
                                  const uint8_t *alive, int64_t M, int64_t n_dst, int64_t n_src, int32_t R,
                                  int32_t tile_rows, int32_t *cells, const int32_t *bucket_cnt, const int32_t *bucket_base,
                                  int32_t *p_src, int32_t *p_dst, float *p_val, int32_t *p_pack, int32_t *chunk_rel,
                                  int32_t *tile_ptr, int32_t *run_ptr, const int32_t *aux, int32_t *p_aux,
                                  int32_t *msg_slot, int64_t n_chunks, void *stream) {
  if (M < 0 || n_dst <= 0 || R <= 0 || tile_rows <= 0 || !cells || !bucket_cnt || !bucket_base || !tile_ptr ||
      (M && (!dst || !src || (!rel && R != 1) || !val || !p_src || !p_dst || !p_val || (!chunk_rel && R != 1)))) { rgcn_set_error("dev_plan_fill: bad argument"); return RGCN_EINVAL; }
  if (p_pack && (n_src >= (int64_t(1) << 24) || tile_rows > 255)) { rgcn_set_error("dev_plan_fill: packed slots need n_src < 2^24 and tile_rows <= 255"); return RGCN_EUNSUPPORTED; }
  const int64_t n_tiles = (n_dst + tile_rows - 1) / tile_rows, nbk = n_tiles * R;
  hipStream_t st = (hipStream_t)stream;
  if (M) hipLaunchKernelGGL(plan_scatter_kernel, dim3(blocks_for(M)), dim3(TB), 0, st, dst, src, rel, val, alive, (long long)M, R,
                            tile_rows, cells, bucket_base, p_src, p_dst, p_val, reinterpret_cast<int2 *>(p_pack), aux, p_aux,
                            msg_slot);
  hipLaunchKernelGGL(plan_finish_kernel, dim3(blocks_for(nbk + 1)), dim3(TB), 0, st, (long long)nbk, R, bucket_cnt, bucket_base,
                     p_src, p_dst, p_val, reinterpret_cast<int2 *>(p_pack), chunk_rel, tile_ptr, run_ptr);
  if (n_chunks > 0 && chunk_rel)
    hipLaunchKernelGGL(chunk_rel_kernel, dim3(blocks_for(n_chunks)), dim3(TB), 0, st, bucket_base, (long long)nbk, R,
                       (long long)n_chunks, chunk_rel);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_dev_csr_pair(const int32_t *a, const int32_t *b, const int32_t *rel, const float *val, const uint8_t *alive,
                                 int64_t M, int64_t N, int32_t *rowbuf, int32_t *scan_tmp, int32_t *e_other, int32_t *e_rel,
                                 float *e_val, void *stream) {
  if (M < 0 || N <= 0 || !rowbuf || !scan_tmp || (M && (!a || !b || !val || !e_other || !e_rel || !e_val))) {
    rgcn_set_error("dev_csr_pair: bad argument");
    return RGCN_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  const long long n_cells = 2 * (long long)N + 1;                 // R[1 ..]: N rows, one dummy, N rows
  HIP_TRY(zero_async(rowbuf, (size_t)(n_cells + 1) * sizeof(int), st));
  if (M) hipLaunchKernelGGL(csr2_count_kernel, dim3(blocks_for(M)), dim3(TB), 0, st, a, b, alive, (long long)M, (long long)N, rowbuf);
  // scan_tmp: n_cells / 1024 + 3 ints (block totals + the grand total, which nobody reads)
  int rc = exclusive_scan(rowbuf + 1, rowbuf + 1, scan_tmp, n_cells, scan_tmp + (n_cells / (TB * 4) + 2), st);
  if (rc) return rc;
  if (M) hipLaunchKernelGGL(csr2_scatter_kernel, dim3(blocks_for(M)), dim3(TB), 0, st, a, b, rel, val, alive, (long long)M,
                            (long long)N, rowbuf, e_other, e_rel, e_val);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_dev_plan_finish_nosync(const int32_t *bucket_base, int64_t n_tiles, int32_t R, int64_t m_pad_ub,
                                           int32_t *p_src, int32_t *p_dst, float *p_val, int32_t *p_pack, int32_t *p_aux,
                                           const int32_t *tile_ptr, int32_t *units, int32_t *items, int64_t n_items_ub,
                                           int32_t max_item_chunks, void *stream) {
  if (!bucket_base || n_tiles <= 0 || R <= 0 || m_pad_ub < 0 || !p_src || !p_dst || !p_val || !tile_ptr ||
      (items && (n_tiles != 1 || max_item_chunks <= 0 || n_items_ub <= 0))) {
    rgcn_set_error("dev_plan_finish_nosync: bad argument");
    return RGCN_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  const long long nbk = (long long)n_tiles * R;
  if (m_pad_ub)
    hipLaunchKernelGGL(plan_tail_kernel, dim3(blocks_for(m_pad_ub)), dim3(TB), 0, st, bucket_base, nbk, (long long)m_pad_ub, p_src,
                       p_dst, p_val, reinterpret_cast<int2 *>(p_pack), p_aux);
  if (units)
    hipLaunchKernelGGL(plan_units_kernel, dim3(blocks_for(n_tiles)), dim3(TB), 0, st, tile_ptr, (long long)n_tiles,
                       reinterpret_cast<int4 *>(units));
  if (items)
    hipLaunchKernelGGL(plan_items_kernel, dim3(1), dim3(TB), 0, st, bucket_base, R, max_item_chunks,
                       reinterpret_cast<int2 *>(items), (long long)n_items_ub);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}
