// Random gather of 64-byte rows (S1's X / G gather: 21 M rows out of a 64 MB table): does any load flavour make the L2 ask the fabric for
// 64 bytes instead of a whole 128-byte line?  Variants: 0 plain global_load_dwordx4, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc0, 5 sc1 nt, 6 sc0 sc1 nt.
// Prints ms per pass; run under rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum for the request sizes.
// build: hipcc -O2 --offload-arch=gfx950 -o tools/micro/gather64.bin tools/micro/gather64.hip   (run on the GPU box: gpurun -- tools/micro/gather64.bin)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V>
__device__ __forceinline__ f32x4 ld(const f32x4 *p) {
  f32x4 r;
  if (V == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
  if (V == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(r) : "v"(p) : "memory");
  if (V == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r) : "v"(p) : "memory");
  if (V == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(r) : "v"(p) : "memory");
  if (V == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(r) : "v"(p) : "memory");
  if (V == 5) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(r) : "v"(p) : "memory");
  if (V == 6) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(r) : "v"(p) : "memory");
  return r;
}

template <int V>
__global__ __launch_bounds__(256) void gather_kernel(const float *__restrict__ X, const int *__restrict__ idx, float *__restrict__ out, long long m) {
  const int q = threadIdx.x & 3;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const long long stride = (long long)gridDim.x * 64;
  for (long long e = ((long long)blockIdx.x * 256 + threadIdx.x) >> 2; e < m; e += 4 * stride) {      // four rows in flight per lane group
    f32x4 r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long ee = e + u * stride;
      const int row = idx[ee < m ? ee : e];
      r[u] = ld<V>(reinterpret_cast<const f32x4 *>(X + (size_t)row * 16 + 4 * q));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += r[u];
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main(int argc, char **argv) {
  // gather64.bin [variant | -1] [N rows of the table] [M rows gathered]: S1's shape by default; AM's is 1666764 13643406
  const long long N = argc > 2 ? atoll(argv[2]) : 1000000, M = argc > 3 ? atoll(argv[3]) : 21000000;
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  std::vector<int> h(M);
  unsigned long long s = 88172645463325252ull;
  for (long long i = 0; i < M; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int)(s % N); }
  float *X, *out; int *idx;
  hipMalloc(&X, N * 64); hipMalloc(&idx, M * 4); hipMalloc(&out, 4096 * 256 * 4);
  hipMemset(X, 0, N * 64);
  hipMemcpy(idx, h.data(), M * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const char *names[7] = {"plain", "nt", "sc1", "sc0 sc1", "sc0", "sc1 nt", "sc0 sc1 nt"};
#define RUN(V)                                                                                          \
  if (only < 0 || only == V) {                                                                          \
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(gather_kernel<V>, dim3(4096), dim3(256), 0, 0, X, idx, out, M); \
    hipEventRecord(a);                                                                                  \
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(gather_kernel<V>, dim3(4096), dim3(256), 0, 0, X, idx, out, M); \
    hipEventRecord(b); hipEventSynchronize(b);                                                          \
    float ms; hipEventElapsedTime(&ms, a, b);                                                           \
    printf("variant %d (%s): %.3f ms per pass of %.1f M rows out of %.2f M (%.0f MB)\n", V, names[V], ms / 5, M / 1e6, N / 1e6, N * 64 / 1e6);                    \
  }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
  return 0;
}
