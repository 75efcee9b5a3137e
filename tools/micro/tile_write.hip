// How fast can a [B][N][d] table (AM as shipped: 40 x 1,666,764 x 10 floats = 2.67 GB) be WRITTEN tile by tile -- TN consecutive nodes per
// tile = B runs of TN d floats at a stride of N d floats, 16 bytes per thread -- and read the same way?  The floor of the tile kernels' stores / loads.
// build: hipcc -O2 --offload-arch=gfx950 -o tools/micro/tile_write.bin tools/micro/tile_write.hip   (run on the GPU box: gpurun -- tools/micro/tile_write.bin)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool WRITE>
__global__ void tile_kernel(float *__restrict__ T, float *__restrict__ sink, long long N, int B, int d, int tn, long long n_tiles) {
  const int pieces = B * tn * d / 4, q4 = tn * d / 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const long long n0 = t * tn < N - tn ? t * tn : N - tn;
    for (int p = threadIdx.x; p < pieces; p += blockDim.x) {
      const int b = p / q4, q = p - b * q4;
      f32x4 *a = reinterpret_cast<f32x4 *>(T + (long long)b * N * d + n0 * d + 4 * q);
      if (WRITE) *a = f32x4{1.f, 2.f, 3.f, (float)t};
      else acc += *a;
    }
  }
  if (!WRITE) sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
  const long long N = 1666764; const int B = 40, d = 10;
  float *T, *sink;
  hipMalloc(&T, (size_t)B * N * d * 4); hipMalloc(&sink, 1024 * 1024 * 4);
  hipMemset(T, 0, (size_t)B * N * d * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int cfg[4][3] = {{16, 512, 512}, {16, 256, 1024}, {32, 256, 1024}, {64, 256, 1024}};      // nodes per tile, workgroups, threads
  for (int w = 0; w < 2; ++w)
    for (auto &c : cfg) {
      const long long nt = (N + c[0] - 1) / c[0];
      float ms;
      for (int r = 0; r < 2; ++r) {
        hipEventRecord(a);
        for (int i = 0; i < 3; ++i) {
          if (w) hipLaunchKernelGGL(tile_kernel<true>, dim3(c[1]), dim3(c[2]), 0, 0, T, sink, N, B, d, c[0], nt);
          else hipLaunchKernelGGL(tile_kernel<false>, dim3(c[1]), dim3(c[2]), 0, 0, T, sink, N, B, d, c[0], nt);
        }
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
      }
      printf("%s %2d nodes per tile, %d workgroups x %d threads: %.3f ms per pass = %.2f TB/s\n", w ? "write" : "read ", c[0], c[1], c[2], ms / 3, (double)B * N * d * 4 / (ms / 3 * 1e-3) / 1e12);
    }
  return 0;
}
