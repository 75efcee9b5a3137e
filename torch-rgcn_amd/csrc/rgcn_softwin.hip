// Soft-window plans on the device (round 6): the graph-build side of rgcn_spmm_blk_f32 / rgcn_bwd_own_f32 on large static graphs.
//
// What replaces the reference's per-forward stack_matrices -> sum_sparse -> sparse COO pipeline (torch_rgcn/utils.py:143-166, :71-97;
// layers.py:255-279) for these kernels is a relation-tile plan in a particular ORDER:
//   * messages bucketed by (destination tile, relation), every bucket padded to a multiple of 16 slots, the slots of a bucket sorted by
//     SOURCE row -- the 16 sources of a chunk span 1 / (chunks per bucket) of the feature table;
//   * the chunks of a tile ordered by their first source -- for the relation-owner backward: grouped by the wave that owns the chunk's
//     relation (or the part of it the chunk belongs to) first.
// Every workgroup then sweeps the source table once per tile, all workgroups together: the whole chip gathers from a few MB at a time, which is
// what the gather is fast on (tools/micro/gather_window.hip).  Two radix sorts (rocPRIM: the one place the library does not hand-roll its own --
// a one-off build step, not the path), a histogram, three scans and four small kernels; the host reads ONE number back (m_pad).
// torch_rgcn._native.build_softwin_plan is the same procedure in torch ops (CPU tests of the invariants run on it).
#include <cstring>

#include "rgcn_device.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace {

typedef unsigned long long u64;
constexpr int SB = 256;
inline unsigned sw_blocks(long long n) { return (unsigned)std::max<long long>(1, (n + SB - 1) / SB); }

// key of a live message: (tile * R + rel) * n_src + src; dropped messages sort behind everything
__global__ __launch_bounds__(SB) void sw_keys_kernel(const int *__restrict__ dst, const int *__restrict__ src, const int *__restrict__ rel,
                                                     const unsigned char *__restrict__ alive, long long M, long long n_src, int R, int tile_rows,
                                                     u64 dropped, u64 *__restrict__ keys, int *__restrict__ order, int *__restrict__ bucket_cnt) {
  const long long m = (long long)blockIdx.x * SB + threadIdx.x;
  if (m >= M) return;
  const bool live = !alive || alive[m];
  u64 key = dropped;
  if (live) {
    const long long b = (long long)(dst[m] / tile_rows) * R + (rel ? rel[m] : 0);
    key = (u64)b * (u64)n_src + (u64)src[m];
    atomicAdd(bucket_cnt + b, 1);
  }
  keys[m] = key;
  order[m] = (int)m;
}

struct Pad16 {
  __host__ __device__ int operator()(int c) const { return (c + 15) & ~15; }
};

__global__ __launch_bounds__(SB) void sw_pads_kernel(int *__restrict__ p_src, int *__restrict__ p_dst, float *__restrict__ p_val, long long n) {
  const long long i = (long long)blockIdx.x * SB + threadIdx.x;
  if (i < n) { p_src[i] = 0; p_dst[i] = -1; p_val[i] = 0.f; }
}

// sorted live position i -> its slot in the bucket-major layout
__global__ __launch_bounds__(SB) void sw_place_kernel(const u64 *__restrict__ keys, const int *__restrict__ order, long long n_live, long long n_src,
                                                      const int *__restrict__ dst, const int *__restrict__ src, const float *__restrict__ val,
                                                      const int *__restrict__ bucket_base, const int *__restrict__ bucket_first,
                                                      int *__restrict__ p_src, int *__restrict__ p_dst, float *__restrict__ p_val) {
  const long long i = (long long)blockIdx.x * SB + threadIdx.x;
  if (i >= n_live) return;
  const long long b = (long long)(keys[i] / (u64)n_src);
  const int m = order[i];
  const long long slot = (long long)bucket_base[b] + (i - bucket_first[b]);
  p_src[slot] = src[m];
  p_dst[slot] = dst[m];
  p_val[slot] = val[m];
}

// chunk c of the bucket-major layout: its bucket (the last bucket whose first slot is <= 16 c: empty buckets share their successor's base),
// its group (tile, or tile and owner wave), its sort key (group, first source), its relation word
__global__ __launch_bounds__(SB) void sw_chunk_kernel(const int *__restrict__ p_src, const int *__restrict__ bucket_base, long long nbk, int R, long long n_src,
                                                      long long n_chunks, const int *__restrict__ parts, const int *__restrict__ unit_base,
                                                      const int *__restrict__ unit_owner, const int *__restrict__ unit_local, int own_waves,
                                                      u64 *__restrict__ ckeys, int *__restrict__ cidx, int *__restrict__ crel, int *__restrict__ group_cnt) {
  const long long c = (long long)blockIdx.x * SB + threadIdx.x;
  if (c >= n_chunks) return;
  const int slot = (int)(16 * c);
  long long lo = 0, hi = nbk;                       // upper bound of slot in bucket_base[0 .. nbk)
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (bucket_base[mid] <= slot) lo = mid + 1; else hi = mid;
  }
  const long long b = lo - 1;
  const int r = (int)(b % R);
  const long long tile = b / R;
  long long group = tile;
  int word = r;
  if (own_waves) {
    const int j = (int)(c - bucket_base[b] / 16);
    const int unit = unit_base[r] + j % parts[r];
    group = tile * own_waves + unit_owner[unit];
    word = r | (unit_local[unit] << 16);
  }
  ckeys[c] = (u64)group * (u64)n_src + (u64)p_src[slot];
  cidx[c] = (int)c;
  crel[c] = word;
  atomicAdd(group_cnt + group, 1);
}

__global__ __launch_bounds__(SB) void sw_permute_kernel(const int *__restrict__ cidx, long long n_chunks, const int *__restrict__ s_src, const int *__restrict__ s_dst,
                                                        const float *__restrict__ s_val, const int *__restrict__ crel, int *__restrict__ p_src,
                                                        int *__restrict__ p_dst, float *__restrict__ p_val, int *__restrict__ chunk_rel) {
  const long long t = (long long)blockIdx.x * SB + threadIdx.x;
  const long long c = t >> 4;
  if (c >= n_chunks) return;
  const int s = (int)(t & 15);
  const long long from = (long long)cidx[c] * 16 + s;
  p_src[t] = s_src[from];
  p_dst[t] = s_dst[from];
  p_val[t] = s_val[from];
  if (s == 0) chunk_rel[c] = crel[cidx[c]];
}

int key_bits(u64 max_key) {
  int b = 1;
  while (b < 64 && (max_key >> b)) ++b;
  return b;
}

}  // namespace

/* device temp bytes for the sorts and scans of up to n elements */
extern "C" int64_t rgcn_softwin_tmp_bytes(int64_t n) {
  if (n < 1) n = 1;
  size_t a = 0, b = 0;
  (void)rocprim::radix_sort_pairs(nullptr, a, (const u64 *)nullptr, (u64 *)nullptr, (const int *)nullptr, (int *)nullptr, (size_t)n, 0, 64, (hipStream_t)0);
  (void)rocprim::exclusive_scan(nullptr, b, (const int *)nullptr, (int *)nullptr, 0, (size_t)n + 1, rocprim::plus<int>(), (hipStream_t)0);
  return (int64_t)std::max(a, b) + 256;
}

extern "C" int rgcn_softwin_order(const int32_t *dst, const int32_t *src, const int32_t *rel, const uint8_t *alive, int64_t M, int64_t n_dst,
                                  int64_t n_src, int32_t R, int32_t tile_rows, uint64_t *keys, uint64_t *keys_sorted, int32_t *order,
                                  int32_t *order_sorted, int32_t *bucket_cnt, int32_t *bucket_base, int32_t *bucket_first, void *tmp,
                                  int64_t tmp_bytes, void *stream) {
  if (M < 0 || n_dst <= 0 || n_src <= 0 || R <= 0 || tile_rows <= 0 || !bucket_cnt || !bucket_base || !bucket_first || !tmp ||
      (M && (!dst || !src || (!rel && R != 1) || !keys || !keys_sorted || !order || !order_sorted))) {
    rgcn_set_error("softwin_order: bad argument");
    return RGCN_EINVAL;
  }
  const int64_t n_tiles = (n_dst + tile_rows - 1) / tile_rows, nbk = n_tiles * R;
  if ((long double)(nbk + 1) * (long double)n_src >= 1.8e19L || M >= INT32_MAX || nbk >= INT32_MAX) { rgcn_set_error("softwin_order: graph too large for 64-bit keys"); return RGCN_EUNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(zero_async(bucket_cnt, (size_t)(nbk + 1) * sizeof(int), st));
  size_t bytes = (size_t)tmp_bytes;
  if (M) {
    const u64 dropped = (u64)nbk * (u64)n_src;                  // past every live key: dropped messages end up behind the live ones
    hipLaunchKernelGGL(sw_keys_kernel, dim3(sw_blocks(M)), dim3(SB), 0, st, dst, src, rel, alive, (long long)M, (long long)n_src, R, tile_rows,
                       dropped, reinterpret_cast<u64 *>(keys), order, bucket_cnt);
    HIP_TRY(hipGetLastError());
    HIP_TRY(rocprim::radix_sort_pairs(tmp, bytes, reinterpret_cast<const u64 *>(keys), reinterpret_cast<u64 *>(keys_sorted), (const int *)order,
                                      order_sorted, (size_t)M, 0, key_bits(dropped), st));
  }
  // first slot of every bucket (counts padded to 16) and first live message of every bucket; entry nbk = the totals
  bytes = (size_t)tmp_bytes;
  HIP_TRY(rocprim::exclusive_scan(tmp, bytes, rocprim::make_transform_iterator((const int *)bucket_cnt, Pad16()), bucket_base, 0, (size_t)nbk + 1,
                                  rocprim::plus<int>(), st));
  bytes = (size_t)tmp_bytes;
  HIP_TRY(rocprim::exclusive_scan(tmp, bytes, (const int *)bucket_cnt, bucket_first, 0, (size_t)nbk + 1, rocprim::plus<int>(), st));
  return RGCN_OK;
}

extern "C" int rgcn_softwin_fill(const int32_t *dst, const int32_t *src, const float *val, const uint64_t *keys_sorted, const int32_t *order_sorted,
                                 int64_t n_live, int64_t n_dst, int64_t n_src, int32_t R, int32_t tile_rows, const int32_t *bucket_base,
                                 const int32_t *bucket_first, int64_t m_pad, const int32_t *parts, const int32_t *unit_base,
                                 const int32_t *unit_owner, const int32_t *unit_local, int32_t own_waves, int32_t *s_src, int32_t *s_dst, float *s_val,
                                 uint64_t *ckeys, uint64_t *ckeys_sorted, int32_t *cidx, int32_t *cidx_sorted, int32_t *crel, int32_t *group_cnt,
                                 int32_t *p_src, int32_t *p_dst, float *p_val, int32_t *chunk_rel, int32_t *group_ptr, void *tmp, int64_t tmp_bytes,
                                 void *stream) {
  if (n_live < 0 || m_pad < 0 || (m_pad & 15) || n_dst <= 0 || n_src <= 0 || R <= 0 || tile_rows <= 0 || own_waves < 0 || !bucket_base || !bucket_first ||
      !group_cnt || !group_ptr || !tmp || (own_waves && (!parts || !unit_base || !unit_owner || !unit_local)) ||
      (m_pad && (!s_src || !s_dst || !s_val || !ckeys || !ckeys_sorted || !cidx || !cidx_sorted || !crel || !p_src || !p_dst || !p_val || !chunk_rel)) ||
      (n_live && (!dst || !src || !val || !keys_sorted || !order_sorted))) {
    rgcn_set_error("softwin_fill: bad argument");
    return RGCN_EINVAL;
  }
  const int64_t n_tiles = (n_dst + tile_rows - 1) / tile_rows, nbk = n_tiles * R, n_chunks = m_pad / 16;
  const int64_t n_groups = n_tiles * (own_waves ? own_waves : 1);
  if ((long double)(n_groups + 1) * (long double)n_src >= 1.8e19L || n_groups >= INT32_MAX || m_pad >= INT32_MAX) { rgcn_set_error("softwin_fill: graph too large"); return RGCN_EUNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(zero_async(group_cnt, (size_t)(n_groups + 1) * sizeof(int), st));
  size_t bytes = (size_t)tmp_bytes;
  if (m_pad) {
    hipLaunchKernelGGL(sw_pads_kernel, dim3(sw_blocks(m_pad)), dim3(SB), 0, st, s_src, s_dst, s_val, (long long)m_pad);
    if (n_live)
      hipLaunchKernelGGL(sw_place_kernel, dim3(sw_blocks(n_live)), dim3(SB), 0, st, reinterpret_cast<const u64 *>(keys_sorted), order_sorted, (long long)n_live,
                         (long long)n_src, dst, src, val, bucket_base, bucket_first, s_src, s_dst, s_val);
    hipLaunchKernelGGL(sw_chunk_kernel, dim3(sw_blocks(n_chunks)), dim3(SB), 0, st, (const int *)s_src, bucket_base, (long long)nbk, R, (long long)n_src,
                       (long long)n_chunks, parts, unit_base, unit_owner, unit_local, own_waves, reinterpret_cast<u64 *>(ckeys), cidx, crel, group_cnt);
    HIP_TRY(hipGetLastError());
    const int bits = key_bits((u64)(n_groups + 1) * (u64)n_src);
    HIP_TRY(rocprim::radix_sort_pairs(tmp, bytes, reinterpret_cast<const u64 *>(ckeys), reinterpret_cast<u64 *>(ckeys_sorted), (const int *)cidx, cidx_sorted,
                                      (size_t)n_chunks, 0, bits, st));
    hipLaunchKernelGGL(sw_permute_kernel, dim3(sw_blocks(m_pad)), dim3(SB), 0, st, (const int *)cidx_sorted, (long long)n_chunks, (const int *)s_src,
                       (const int *)s_dst, (const float *)s_val, (const int *)crel, p_src, p_dst, p_val, chunk_rel);
    HIP_TRY(hipGetLastError());
  }
  bytes = (size_t)tmp_bytes;
  HIP_TRY(rocprim::exclusive_scan(tmp, bytes, (const int *)group_cnt, group_ptr, 0, (size_t)n_groups + 1, rocprim::plus<int>(), st));
  return RGCN_OK;
}
