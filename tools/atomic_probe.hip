// Micro-probe: what does flushing a 16x16 fp32 block (256 floats = 4 wave-wide global_atomic_add_f32) per (tile, relation)
// run cost on MI355X, alone and next to the random row gathers of the message-passing kernels?  Decides how the fused
// backward kernel (dX tile + dW partial per run) gets its dW partials out: one device-wide copy, one copy per XCD (indexed by
// the hardware XCC id), or many copies.
// Build: make -C tools atomic_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15; }   // HW_REG_XCC_ID[3:0]

// MODE 0: one copy; 1: copy per XCC id; 2: copy = blockIdx % n_copies; GATHER: also do `chunks` random 16-row gathers per run
// SCOPE 0: atomicAdd (agent scope); 1: workgroup scope; 2: wavefront scope (executed in the XCD's own L2 -- only sound for
// per-XCC copies, where every updater of a copy sits behind the same L2)
template <int SCOPE>
__device__ __forceinline__ void add_f32(float* p, float v) {
  if (SCOPE == 0) atomicAdd(p, v);
  else if (SCOPE == 1) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

template <int MODE, bool GATHER, bool FLUSH, int SCOPE = 0>
__global__ __launch_bounds__(256) void probe(float* __restrict__ dW, int n_copies, int R, int n_tiles, const float4* __restrict__ tab,
                                             const int* __restrict__ idx, int chunks, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= n_tiles) return;
  int copy = 0;
  if (MODE == 1) copy = xcc_id() % n_copies;
  if (MODE == 2) copy = blockIdx.x % n_copies;
  float* base = dW + (size_t)copy * R * 256;
  const int m = lane & 15, k = lane >> 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = 0; r < R; ++r) {
    if (GATHER) {
      for (int c = 0; c < chunks; ++c) {
        const int s = idx[((size_t)(tile * R + r) * chunks + c) * 16 + m];
        const float4 x = tab[(size_t)s * 4 + k];
        acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
      }
    } else {
      acc.x += 1.f; acc.y += 2.f; acc.z += 1.f; acc.w += 0.5f;
    }
    if (FLUSH) {
      float* p = base + (size_t)r * 256 + lane;
      add_f32<SCOPE>(p, acc.x); add_f32<SCOPE>(p + 64, acc.y); add_f32<SCOPE>(p + 128, acc.z); add_f32<SCOPE>(p + 192, acc.w);
    }
  }
  if (acc.x == 12345.678f) out[0] = acc.x;
}

template <int MODE, bool GATHER, bool FLUSH, int SCOPE = 0>
float run(float* dW, int n_copies, int R, int n_tiles, const float4* tab, const int* idx, int chunks, float* out) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int grid = (n_tiles + 3) / 4;
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe<MODE, GATHER, FLUSH, SCOPE>), dim3(grid), dim3(256), 0, 0, dW, n_copies, R, n_tiles, tab, idx, chunks, out);
  CK(hipEventRecord(a));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((probe<MODE, GATHER, FLUSH, SCOPE>), dim3(grid), dim3(256), 0, 0, dW, n_copies, R, n_tiles, tab, idx, chunks, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 5;
}

int main() {
  const int R = 101, n_tiles = 7813, chunks = 2;
  const long rows = 1L << 20, n_idx = (long)n_tiles * R * chunks * 16;
  float4* tab; int* idx; float* out; float* dW;
  CK(hipMalloc(&tab, rows * 64)); CK(hipMalloc(&idx, n_idx * 4)); CK(hipMalloc(&out, 4)); CK(hipMalloc(&dW, (size_t)2048 * R * 256 * 4));
  CK(hipMemset(tab, 0, rows * 64)); CK(hipMemset(dW, 0, (size_t)2048 * R * 256 * 4));
  std::vector<int> h(n_idx);
  unsigned long long x = 88172645463325252ull;
  for (long i = 0; i < n_idx; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (int)(x % (unsigned long long)rows); }
  CK(hipMemcpy(idx, h.data(), n_idx * 4, hipMemcpyHostToDevice));
  printf("flushes per launch: %d (R=%d x tiles=%d), 256 floats each\n", R * n_tiles, R, n_tiles);
  printf("flush only : one copy %.3f ms | per-XCC-id x8 %.3f | blockIdx%%8 %.3f | blockIdx%%64 %.3f | blockIdx%%512 %.3f | blockIdx%%2048 %.3f\n",
         run<0, false, true>(dW, 1, R, n_tiles, tab, idx, chunks, out), run<1, false, true>(dW, 8, R, n_tiles, tab, idx, chunks, out),
         run<2, false, true>(dW, 8, R, n_tiles, tab, idx, chunks, out), run<2, false, true>(dW, 64, R, n_tiles, tab, idx, chunks, out),
         run<2, false, true>(dW, 512, R, n_tiles, tab, idx, chunks, out), run<2, false, true>(dW, 2048, R, n_tiles, tab, idx, chunks, out));
  printf("gather only (%d chunks of 16 random 64-B rows per run): %.3f ms\n", chunks, run<0, true, false>(dW, 1, R, n_tiles, tab, idx, chunks, out));
  printf("gather+flush: one copy %.3f ms | per-XCC-id x8 %.3f | blockIdx%%8 %.3f | blockIdx%%64 %.3f | blockIdx%%512 %.3f | blockIdx%%2048 %.3f\n",
         run<0, true, true>(dW, 1, R, n_tiles, tab, idx, chunks, out), run<1, true, true>(dW, 8, R, n_tiles, tab, idx, chunks, out),
         run<2, true, true>(dW, 8, R, n_tiles, tab, idx, chunks, out), run<2, true, true>(dW, 64, R, n_tiles, tab, idx, chunks, out),
         run<2, true, true>(dW, 512, R, n_tiles, tab, idx, chunks, out), run<2, true, true>(dW, 2048, R, n_tiles, tab, idx, chunks, out));
  printf("scopes, per-XCC-id x8 copies: flush only  agent %.3f | workgroup %.3f | wavefront %.3f ms\n",
         run<1, false, true, 0>(dW, 8, R, n_tiles, tab, idx, chunks, out), run<1, false, true, 1>(dW, 8, R, n_tiles, tab, idx, chunks, out),
         run<1, false, true, 2>(dW, 8, R, n_tiles, tab, idx, chunks, out));
  printf("scopes, per-XCC-id x8 copies: gather+flush agent %.3f | workgroup %.3f | wavefront %.3f ms\n",
         run<1, true, true, 0>(dW, 8, R, n_tiles, tab, idx, chunks, out), run<1, true, true, 1>(dW, 8, R, n_tiles, tab, idx, chunks, out),
         run<1, true, true, 2>(dW, 8, R, n_tiles, tab, idx, chunks, out));
  for (int scope = 1; scope <= 2; ++scope) {      // do the narrow scopes still add up?
    CK(hipMemset(dW, 0, (size_t)8 * R * 256 * 4));
    if (scope == 1) hipLaunchKernelGGL((probe<1, false, true, 1>), dim3((n_tiles + 3) / 4), dim3(256), 0, 0, dW, 8, R, n_tiles, tab, idx, chunks, out);
    else hipLaunchKernelGGL((probe<1, false, true, 2>), dim3((n_tiles + 3) / 4), dim3(256), 0, 0, dW, 8, R, n_tiles, tab, idx, chunks, out);
    CK(hipDeviceSynchronize());
    std::vector<float> w2((size_t)8 * R * 256);
    CK(hipMemcpy(w2.data(), dW, w2.size() * 4, hipMemcpyDeviceToHost));
    double tot2 = 0;
    for (int c = 0; c < 8; ++c) for (int i = 0; i < 64; ++i) tot2 += w2[(size_t)c * R * 256 + i];
    printf("scope %d: sum over copies of dW[0][0..63] = %.0f (expected %d)\n", scope, tot2, 64 * n_tiles);
  }
  // correctness of the per-XCC copies: the sum over copies must equal the number of flushes times the flushed value
  CK(hipMemset(dW, 0, (size_t)8 * R * 256 * 4));
  hipLaunchKernelGGL((probe<1, false, true>), dim3((n_tiles + 3) / 4), dim3(256), 0, 0, dW, 8, R, n_tiles, tab, idx, chunks, out);
  CK(hipDeviceSynchronize());
  std::vector<float> w((size_t)8 * R * 256);
  CK(hipMemcpy(w.data(), dW, w.size() * 4, hipMemcpyDeviceToHost));
  double tot = 0; int used = 0;
  for (int c = 0; c < 8; ++c) { double s = 0; for (int i = 0; i < 64; ++i) s += w[(size_t)c * R * 256 + i]; tot += s; used += s != 0; }
  printf("per-XCC copies in use: %d of 8; sum over copies of dW[0][0..63] = %.0f (expected %d)\n", used, tot, 64 * n_tiles);
  return 0;
}
