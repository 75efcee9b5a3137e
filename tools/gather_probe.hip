// Micro-probe: what does MI355X deliver for random 64-byte row gathers (the R-GCN access pattern)?
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o gpurun_out/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// each 16-lane group reads one 64-byte row per step; U independent loads in flight per lane
template <int U>
__global__ __launch_bounds__(256) void gather(const float4* __restrict__ tab, const int* __restrict__ idx, float* out, long n_rows_to_read) {
  const int lane = threadIdx.x & 63, m = lane & 15, k = lane >> 4;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * 4;
  float4 acc = make_float4(0, 0, 0, 0);
  // each wave-step covers U chunks of 16 rows
  for (long c = wave * U; c * 16 < n_rows_to_read; c += nw * U) {
    int s[U];
#pragma unroll
    for (int j = 0; j < U; ++j) s[j] = idx[(c + j) * 16 + m];
    float4 x[U];
#pragma unroll
    for (int j = 0; j < U; ++j) x[j] = tab[(size_t)s[j] * 4 + k];
#pragma unroll
    for (int j = 0; j < U; ++j) { acc.x += x[j].x; acc.y += x[j].y; acc.z += x[j].z; acc.w += x[j].w; }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

__global__ __launch_bounds__(256) void stream(const float4* __restrict__ tab, float* out, long n4) {
  float4 acc = make_float4(0, 0, 0, 0);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 x = tab[i]; acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <int U>
float run_gather(const float4* tab, const int* idx, float* out, long n, int grid) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(gather<U>, dim3(grid), dim3(256), 0, 0, tab, idx, out, n);
  CK(hipEventRecord(a));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(gather<U>, dim3(grid), dim3(256), 0, 0, tab, idx, out, n);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 5;
}

int main() {
  const long n_reads = 24L << 20;  // 24M row reads (~S1 messages)
  for (long rows : {1L << 20, 16L << 20}) {   // 64 MB (fits Infinity Cache) and 1 GB tables
    float4* tab; int* idx; float* out;
    CK(hipMalloc(&tab, rows * 64)); CK(hipMalloc(&idx, n_reads * 4)); CK(hipMalloc(&out, 4));
    CK(hipMemset(tab, 0, rows * 64));
    std::vector<int> h(n_reads);
    unsigned long long x = 88172645463325252ull;
    for (long i = 0; i < n_reads; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (int)(x % (unsigned long long)rows); }
    CK(hipMemcpy(idx, h.data(), n_reads * 4, hipMemcpyHostToDevice));
    for (int grid : {1024, 2048, 4096}) {
      float t1 = run_gather<1>(tab, idx, out, n_reads, grid), t2 = run_gather<2>(tab, idx, out, n_reads, grid);
      float t4 = run_gather<4>(tab, idx, out, n_reads, grid), t8 = run_gather<8>(tab, idx, out, n_reads, grid);
      printf("table %5ld MB grid %4d : random 64B rows  U1 %.3f ms (%.0f GB/s)  U2 %.3f (%.0f)  U4 %.3f (%.0f)  U8 %.3f (%.0f)\n", rows * 64 >> 20, grid,
             t1, n_reads * 64 / t1 / 1e6, t2, n_reads * 64 / t2 / 1e6, t4, n_reads * 64 / t4 / 1e6, t8, n_reads * 64 / t8 / 1e6);
    }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(stream, dim3(4096), dim3(256), 0, 0, tab, out, rows * 4);
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(stream, dim3(4096), dim3(256), 0, 0, tab, out, rows * 4);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    printf("table %5ld MB sequential float4 stream %.3f ms (%.0f GB/s)\n", rows * 64 >> 20, ms, rows * 64 / ms / 1e6);
    CK(hipFree(tab)); CK(hipFree(idx)); CK(hipFree(out));
  }
  return 0;
}
