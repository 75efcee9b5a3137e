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
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rgcn_hip.h"

extern "C" void rgcn_set_error(const char *fmt, ...);

typedef float f32x4 __attribute__((ext_vector_type(4)));

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

__device__ __forceinline__ void lds_add(float *p, float v) { atomicAdd(p, v); }

// ------------------------------------------------------------------ spmm, d_in = d_out = 16
// The S1 / hidden-16 fast path.  K-slot permutation: MFMA step c, k-slot k carries feature
// 4k + c, so each lane's four A values are ONE 16-byte load of its source row.
__global__ __launch_bounds__(WG) void spmm_d16_kernel(
    const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias,
    float *__restrict__ out, const int *__restrict__ p_src, const int *__restrict__ p_dst,
    const float *__restrict__ p_val, const int *__restrict__ chunk_rel, const int *__restrict__ tile_ptr,
    int tile_rows, int n_dst, int relu_out) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const int t = blockIdx.x;
  const int row0 = t * tile_rows;
  const int nrows = min(tile_rows, n_dst - row0);
  const int tid = threadIdx.x;
  for (int i = tid; i < nrows * 4; i += WG) reinterpret_cast<float4 *>(tile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();

  const int c0 = tile_ptr[t], c1 = tile_ptr[t + 1];
  const int wave = tid >> 6, lane = tid & 63;
  const int m = lane & 15, k = lane >> 4;
  const int per = (c1 - c0 + 3) >> 2;
  const int my0 = c0 + wave * per;
  const int my1 = min(c1, my0 + per);
  int cur = -1;
  float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
  for (int c = my0; c < my1; ++c) {
    const int r = __builtin_amdgcn_readfirstlane(chunk_rel[c]);
    if (r != cur) {
      const float *wr = W + (size_t)r * 256 + (4 * k) * 16 + m;
      b0 = wr[0];
      b1 = wr[16];
      b2 = wr[32];
      b3 = wr[48];
      cur = r;
    }
    const int e = c * RGCN_CHUNK + m;
    const int s = p_src[e];
    const float v = p_val[e];
    const int4 dd = *reinterpret_cast<const int4 *>(p_dst + c * RGCN_CHUNK + 4 * k);
    const float4 x = *reinterpret_cast<const float4 *>(X + (size_t)s * 16 + 4 * k);
    const bool live = v != 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(live ? x.x * v : 0.f, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(live ? x.y * v : 0.f, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(live ? x.z * v : 0.f, b2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(live ? x.w * v : 0.f, b3, acc, 0, 0, 0);
    // D: lane (k, m) holds rows 4k..4k+3, column m
    lds_add(&tile[(dd.x - row0) * 16 + m], acc[0]);
    lds_add(&tile[(dd.y - row0) * 16 + m], acc[1]);
    lds_add(&tile[(dd.z - row0) * 16 + m], acc[2]);
    lds_add(&tile[(dd.w - row0) * 16 + m], acc[3]);
  }
  __syncthreads();
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) bv = reinterpret_cast<const float4 *>(bias)[tid & 3];
  float4 *o4 = reinterpret_cast<float4 *>(out + (size_t)row0 * 16);
  for (int i = tid; i < nrows * 4; i += WG) {  // (i & 3) == (tid & 3) because WG % 4 == 0
    float4 a = reinterpret_cast<const float4 *>(tile)[i];
    a.x += bv.x; a.y += bv.y; a.z += bv.z; a.w += bv.w;
    if (relu_out) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
    o4[i] = a;
  }
}

// ------------------------------------------------------------------ spmm, any d_in / d_out
// NJT = output column tiles (16 wide) kept in accumulators per pass over d_in.
template <int NJT>
__global__ __launch_bounds__(WG) void spmm_generic_kernel(
    const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ bias,
    float *__restrict__ out, const int *__restrict__ p_src, const int *__restrict__ p_dst,
    const float *__restrict__ p_val, const int *__restrict__ chunk_rel, const int *__restrict__ tile_ptr,
    int tile_rows, int n_dst, int d_in, int d_out, int relu_out) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const int t = blockIdx.x;
  const int row0 = t * tile_rows;
  const int nrows = min(tile_rows, n_dst - row0);
  const int tid = threadIdx.x;
  for (int i = tid; i < nrows * d_out; i += WG) tile[i] = 0.f;
  __syncthreads();

  const int c0 = tile_ptr[t], c1 = tile_ptr[t + 1];
  const int wave = tid >> 6, lane = tid & 63;
  const int m = lane & 15, k = lane >> 4;
  const int per = (c1 - c0 + 3) >> 2;
  const int my0 = c0 + wave * per;
  const int my1 = min(c1, my0 + per);
  for (int c = my0; c < my1; ++c) {
    const int r = __builtin_amdgcn_readfirstlane(chunk_rel[c]);
    const int e = c * RGCN_CHUNK + m;
    const int s = p_src[e];
    const float v = p_val[e];
    const bool live = v != 0.f;
    const int4 dd = *reinterpret_cast<const int4 *>(p_dst + c * RGCN_CHUNK + 4 * k);
    const float *xrow = X + (size_t)s * d_in;
    const float *wr = W + (size_t)r * d_in * d_out;
    for (int jg = 0; jg < d_out; jg += 16 * NJT) {
      f32x4 acc[NJT];
#pragma unroll
      for (int jt = 0; jt < NJT; ++jt) acc[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int kc = 0; kc < d_in; kc += 4) {
        const int f = kc + k;
        const bool fin = f < d_in;
        float a = (fin && live) ? xrow[f] * v : 0.f;
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
          const int col = jg + jt * 16 + m;
          const float b = (fin && col < d_out) ? wr[(size_t)f * d_out + col] : 0.f;
          acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[jt], 0, 0, 0);
        }
      }
#pragma unroll
      for (int jt = 0; jt < NJT; ++jt) {
        const int col = jg + jt * 16 + m;
        if (col < d_out) {
          lds_add(&tile[(dd.x - row0) * d_out + col], acc[jt][0]);
          lds_add(&tile[(dd.y - row0) * d_out + col], acc[jt][1]);
          lds_add(&tile[(dd.z - row0) * d_out + col], acc[jt][2]);
          lds_add(&tile[(dd.w - row0) * d_out + col], acc[jt][3]);
        }
      }
    }
  }
  __syncthreads();
  float *o = out + (size_t)row0 * d_out;
  for (int i = tid; i < nrows * d_out; i += WG) {
    float a = tile[i] + (bias ? bias[i % d_out] : 0.f);
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

// ------------------------------------------------------------------ featureless layer
// lpr = lanes per table row (power of two <= 64, >= min(d,64) rounded up)
__global__ __launch_bounds__(WG) void featureless_fwd_kernel(
    const float *__restrict__ table, const float *__restrict__ bias, float *__restrict__ out,
    const int *__restrict__ p_src, const int *__restrict__ p_dst, const float *__restrict__ p_val,
    const int *__restrict__ chunk_rel, const int *__restrict__ tile_ptr, int tile_rows, int n_dst,
    long long n_src, int d, int lpr) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const int t = blockIdx.x;
  const int row0 = t * tile_rows;
  const int nrows = min(tile_rows, n_dst - row0);
  const int tid = threadIdx.x;
  for (int i = tid; i < nrows * d; i += WG) tile[i] = 0.f;
  __syncthreads();
  const int c0 = tile_ptr[t], c1 = tile_ptr[t + 1];
  const int wave = tid >> 6, lane = tid & 63;
  const int rpi = 64 / lpr;  // rows per iteration
  const int sub = lane / lpr, jj = lane % lpr;
  for (int c = c0 + wave; c < c1; c += WG / 64) {
    const long long r = chunk_rel[c];
    for (int m0 = 0; m0 < RGCN_CHUNK; m0 += rpi) {
      const int mm = m0 + sub;
      if (mm >= RGCN_CHUNK) continue;
      const int e = c * RGCN_CHUNK + mm;
      const float v = p_val[e];
      if (v == 0.f) continue;
      const float *row = table + (size_t)(r * n_src + p_src[e]) * d;
      float *dstp = tile + (size_t)(p_dst[e] - row0) * d;
      for (int j = jj; j < d; j += lpr) lds_add(&dstp[j], v * row[j]);
    }
  }
  __syncthreads();
  float *o = out + (size_t)row0 * d;
  for (int i = tid; i < nrows * d; i += WG) o[i] = tile[i] + (bias ? bias[i % d] : 0.f);
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

// ------------------------------------------------------------------ column sum (bias gradient)
__global__ __launch_bounds__(WG) void colsum_kernel(const float *__restrict__ G, float *__restrict__ db,
                                                    long long n, int d) {
  // thread -> column (tid % d) when d <= 256; rows strided by WG / d groups
  __shared__ float part[WG];
  const int tid = threadIdx.x;
  const int groups = max(1, WG / d);
  const int col = tid % d, grp = tid / d;
  for (int cb = 0; cb < d; cb += WG) {  // column blocks when d > 256
    const int cc = cb + col;
    float a = 0.f;
    if (grp < groups && cc < d)
      for (long long row = (long long)blockIdx.x * groups + grp; row < n; row += (long long)gridDim.x * groups)
        a += G[(size_t)row * d + cc];
    part[tid] = a;
    __syncthreads();
    if (grp == 0 && cc < d) {
      float s = 0.f;
      for (int g2 = 0; g2 < groups; ++g2) s += part[g2 * d + col];
      atomicAdd(&db[cc], s);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ DistMult
__global__ __launch_bounds__(WG) void distmult_fwd_kernel(
    const long long *__restrict__ tr, long long T, const float *__restrict__ nodes, const float *__restrict__ rel,
    const float *__restrict__ sb, const float *__restrict__ pb, const float *__restrict__ ob,
    float *__restrict__ scores, int d) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long wstride = (long long)gridDim.x * (WG / 64);
  for (long long t = (long long)blockIdx.x * (WG / 64) + wave; t < T; t += wstride) {
    const long long s = tr[3 * t], p = tr[3 * t + 1], o = tr[3 * t + 2];
    const float *ns = nodes + (size_t)s * d, *rp = rel + (size_t)p * d, *no = nodes + (size_t)o * d;
    float a = 0.f;
    for (int j = lane; j < d; j += 64) a += ns[j] * rp[j] * no[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if (lane == 0) {
      if (sb) a += sb[s] + pb[p] + ob[o];
      scores[t] = a;
    }
  }
}

__global__ __launch_bounds__(WG) void distmult_bwd_kernel(
    const long long *__restrict__ tr, long long T, const float *__restrict__ nodes, const float *__restrict__ rel,
    const float *__restrict__ gs, float *__restrict__ dnodes, float *__restrict__ drel, float *__restrict__ dsb,
    float *__restrict__ dpb, float *__restrict__ dob, int d) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long wstride = (long long)gridDim.x * (WG / 64);
  for (long long t = (long long)blockIdx.x * (WG / 64) + wave; t < T; t += wstride) {
    const long long s = tr[3 * t], p = tr[3 * t + 1], o = tr[3 * t + 2];
    const float g = gs[t];
    const float *ns = nodes + (size_t)s * d, *rp = rel + (size_t)p * d, *no = nodes + (size_t)o * d;
    for (int j = lane; j < d; j += 64) {
      const float a = ns[j], b = rp[j], c = no[j];
      atomicAdd(&dnodes[(size_t)s * d + j], g * b * c);
      atomicAdd(&dnodes[(size_t)o * d + j], g * b * a);
      atomicAdd(&drel[(size_t)p * d + j], g * a * c);
    }
    if (dsb && lane == 0) {
      atomicAdd(&dsb[s], g);
      atomicAdd(&dpb[p], g);
      atomicAdd(&dob[o], g);
    }
  }
}

int pow2_lanes(int d) {
  int l = 1;
  while (l < d && l < 64) l <<= 1;
  return l;
}

}  // namespace

// =================================================================== C ABI launchers

extern "C" int rgcn_spmm_f32(const float *X, const float *W, const float *bias, float *out, const int32_t *p_src,
                             const int32_t *p_dst, const float *p_val, const int32_t *chunk_rel,
                             const int32_t *tile_ptr, int64_t n_tiles, int32_t tile_rows, int64_t n_dst,
                             int64_t n_src, int32_t R, int32_t d_in, int32_t d_out, int32_t relu_out,
                             void *stream) {
  (void)n_src;
  (void)R;
  if (!X || !W || !out || !tile_ptr || d_in <= 0 || d_out <= 0 || tile_rows <= 0 || n_dst < 0 ||
      n_tiles != (n_dst + tile_rows - 1) / tile_rows) {
    rgcn_set_error("spmm: bad argument");
    return RGCN_EINVAL;
  }
  if (n_tiles == 0) return RGCN_OK;
  const size_t lds = (size_t)tile_rows * d_out * sizeof(float);
  if (lds > LDS_TILE_BYTES) {
    rgcn_set_error("spmm: tile_rows*d_out*4 = %zu exceeds the %d-byte LDS tile budget", lds, LDS_TILE_BYTES);
    return RGCN_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)n_tiles), block(WG);
  if (d_in == 16 && d_out == 16) {
    hipLaunchKernelGGL(spmm_d16_kernel, grid, block, lds, st, X, W, bias, out, p_src, p_dst, p_val, chunk_rel,
                       tile_ptr, tile_rows, (int)n_dst, relu_out);
  } else if (d_out <= 16) {
    hipLaunchKernelGGL(spmm_generic_kernel<1>, grid, block, lds, st, X, W, bias, out, p_src, p_dst, p_val,
                       chunk_rel, tile_ptr, tile_rows, (int)n_dst, d_in, d_out, relu_out);
  } else if (d_out <= 32) {
    hipLaunchKernelGGL(spmm_generic_kernel<2>, grid, block, lds, st, X, W, bias, out, p_src, p_dst, p_val,
                       chunk_rel, tile_ptr, tile_rows, (int)n_dst, d_in, d_out, relu_out);
  } else if (d_out <= 64) {
    hipLaunchKernelGGL(spmm_generic_kernel<4>, grid, block, lds, st, X, W, bias, out, p_src, p_dst, p_val,
                       chunk_rel, tile_ptr, tile_rows, (int)n_dst, d_in, d_out, relu_out);
  } else {
    hipLaunchKernelGGL(spmm_generic_kernel<8>, grid, block, lds, st, X, W, bias, out, p_src, p_dst, p_val,
                       chunk_rel, tile_ptr, tile_rows, (int)n_dst, d_in, d_out, relu_out);
  }
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
  HIP_TRY(hipMemsetAsync(dW, 0, (size_t)R * d_in * d_out * sizeof(float), st));
  if (n_items == 0) return RGCN_OK;
  const unsigned gx = (unsigned)((n_items + WG / 64 - 1) / (WG / 64));
  const int2 *it2 = reinterpret_cast<const int2 *>(items);
  if (d_in <= 16 && d_out <= 16) {
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

extern "C" int rgcn_featureless_fwd_f32(const float *table, const float *bias, float *out, const int32_t *p_src,
                                        const int32_t *p_dst, const float *p_val, const int32_t *chunk_rel,
                                        const int32_t *tile_ptr, int64_t n_tiles, int32_t tile_rows, int64_t n_dst,
                                        int64_t n_src, int32_t R, int32_t d_out, void *stream) {
  (void)R;
  if (!table || !out || !tile_ptr || d_out <= 0 || tile_rows <= 0 ||
      n_tiles != (n_dst + tile_rows - 1) / tile_rows) {
    rgcn_set_error("featureless_fwd: bad argument");
    return RGCN_EINVAL;
  }
  if (n_tiles == 0) return RGCN_OK;
  const size_t lds = (size_t)tile_rows * d_out * sizeof(float);
  if (lds > LDS_TILE_BYTES) { rgcn_set_error("featureless_fwd: LDS tile too large"); return RGCN_EINVAL; }
  hipLaunchKernelGGL(featureless_fwd_kernel, dim3((unsigned)n_tiles), dim3(WG), lds, (hipStream_t)stream, table,
                     bias, out, p_src, p_dst, p_val, chunk_rel, tile_ptr, tile_rows, (int)n_dst, (long long)n_src,
                     d_out, pow2_lanes(d_out));
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_featureless_wgrad_f32(const float *G, float *dtable, const int32_t *p_src, const int32_t *p_dst,
                                          const float *p_val, const int32_t *chunk_rel, int64_t n_chunks,
                                          int64_t n_dst, int64_t n_src, int32_t R, int32_t d_out, void *stream) {
  (void)n_dst;
  if (!G || !dtable || R <= 0 || d_out <= 0 || n_chunks < 0) { rgcn_set_error("featureless_wgrad: bad argument"); return RGCN_EINVAL; }
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(dtable, 0, (size_t)R * n_src * d_out * sizeof(float), st));
  if (n_chunks == 0) return RGCN_OK;
  const unsigned gx = (unsigned)std::min<int64_t>((n_chunks + 3) / 4, 256 * 16);
  hipLaunchKernelGGL(featureless_wgrad_kernel, dim3(gx), dim3(WG), 0, st, G, dtable, p_src, p_dst, p_val, chunk_rel,
                     (long long)n_chunks, (long long)n_src, d_out, pow2_lanes(d_out));
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_colsum_f32(const float *G, float *db, int64_t n, int32_t d, void *stream) {
  if (!G || !db || n < 0 || d <= 0) { rgcn_set_error("colsum: bad argument"); return RGCN_EINVAL; }
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(db, 0, (size_t)d * sizeof(float), st));
  if (n == 0) return RGCN_OK;
  const int groups = std::max(1, WG / d);
  const unsigned gx = (unsigned)std::min<int64_t>((n + groups - 1) / groups, 1024);
  hipLaunchKernelGGL(colsum_kernel, dim3(gx), dim3(WG), 0, st, G, db, (long long)n, d);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_distmult_fwd_f32(const int64_t *triples, int64_t T, const float *nodes, const float *rel,
                                     const float *sbias, const float *pbias, const float *obias, float *scores,
                                     int64_t n_nodes, int32_t n_rel, int32_t d, void *stream) {
  (void)n_nodes;
  (void)n_rel;
  if (T < 0 || d <= 0 || (T && (!triples || !nodes || !rel || !scores))) { rgcn_set_error("distmult_fwd: bad argument"); return RGCN_EINVAL; }
  if ((sbias != nullptr) != (pbias != nullptr) || (sbias != nullptr) != (obias != nullptr)) { rgcn_set_error("distmult_fwd: biases must be all set or all NULL"); return RGCN_EINVAL; }
  if (T == 0) return RGCN_OK;
  const unsigned gx = (unsigned)std::min<int64_t>((T + 3) / 4, 256 * 32);
  hipLaunchKernelGGL(distmult_fwd_kernel, dim3(gx), dim3(WG), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long *>(triples), (long long)T, nodes, rel, sbias, pbias, obias,
                     scores, d);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

extern "C" int rgcn_distmult_bwd_f32(const int64_t *triples, int64_t T, const float *nodes, const float *rel,
                                     const float *gs, float *dnodes, float *drel, float *dsbias, float *dpbias,
                                     float *dobias, int64_t n_nodes, int32_t n_rel, int32_t d, void *stream) {
  if (T < 0 || d <= 0 || !dnodes || !drel || (T && (!triples || !nodes || !rel || !gs))) { rgcn_set_error("distmult_bwd: bad argument"); return RGCN_EINVAL; }
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(dnodes, 0, (size_t)n_nodes * d * sizeof(float), st));
  HIP_TRY(hipMemsetAsync(drel, 0, (size_t)n_rel * d * sizeof(float), st));
  if (dsbias) {
    HIP_TRY(hipMemsetAsync(dsbias, 0, (size_t)n_nodes * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(dobias, 0, (size_t)n_nodes * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(dpbias, 0, (size_t)n_rel * sizeof(float), st));
  }
  if (T == 0) return RGCN_OK;
  const unsigned gx = (unsigned)std::min<int64_t>((T + 3) / 4, 256 * 32);
  hipLaunchKernelGGL(distmult_bwd_kernel, dim3(gx), dim3(WG), 0, st, reinterpret_cast<const long long *>(triples),
                     (long long)T, nodes, rel, gs, dnodes, drel, dsbias, dpbias, dobias, d);
  HIP_TRY(hipGetLastError());
  return RGCN_OK;
}
