// What does it cost to ADD a message's row straight into the destination table instead of writing it to a per-message buffer and
// gathering it back (the tile forward of the featureless basis layer: Y [M, 16] written + gather_rows_sum4)?  M random rows of W floats
// (W = 10: AM as shipped) added into a [N, 16] table with global_atomic_add_f32 (no return), 16 lanes per row, next to the same rows
// written sequentially (the Y write) and gathered back (the row sum's reads).
// build: hipcc -O2 --offload-arch=gfx950 -munsafe-fp-atomics -o tools/micro/row_atomic.bin tools/micro/row_atomic.hip
// run:   tools/micro/row_atomic.bin [N] [M] [W]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void atomic_rows(float *__restrict__ out, const int *__restrict__ idx, long long m, int w) {
  const int c = threadIdx.x & 15;
  const long long stride = (long long)gridDim.x * 16;
  for (long long e = ((long long)blockIdx.x * 256 + threadIdx.x) >> 4; e < m; e += stride) {
    const int row = idx[e];
    if (c < w) __hip_atomic_fetch_add(out + (size_t)row * 16 + c, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// four 16-byte pieces per row with packed adds? (global_atomic_pk_add has no f32 form: one dword per lane is all there is)
__global__ __launch_bounds__(256) void write_rows(float *__restrict__ Y, long long m) {
  const int q = threadIdx.x & 3;
  const long long stride = (long long)gridDim.x * 64;
  for (long long e = ((long long)blockIdx.x * 256 + threadIdx.x) >> 2; e < m; e += stride)
    *reinterpret_cast<f32x4 *>(Y + (size_t)e * 16 + 4 * q) = f32x4{1.f, 2.f, 3.f, 4.f};
}
__global__ __launch_bounds__(256) void scatter_rows(float *__restrict__ Y, const int *__restrict__ perm, long long m) {
  const int q = threadIdx.x & 3;
  const long long stride = (long long)gridDim.x * 64;
  for (long long e = ((long long)blockIdx.x * 256 + threadIdx.x) >> 2; e < m; e += stride)
    *reinterpret_cast<f32x4 *>(Y + (size_t)perm[e] * 16 + 4 * q) = f32x4{1.f, 2.f, 3.f, 4.f};
}
__global__ __launch_bounds__(256) void gather_rows(const float *__restrict__ Y, const int *__restrict__ perm, float *__restrict__ out, long long m) {
  const int q = threadIdx.x & 3;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const long long stride = (long long)gridDim.x * 64;
  for (long long e = ((long long)blockIdx.x * 256 + threadIdx.x) >> 2; e < m; e += 2 * stride) {
    const long long e2 = e + stride < m ? e + stride : e;
    const int p0 = perm[e], p1 = perm[e2];
    acc += *reinterpret_cast<const f32x4 *>(Y + (size_t)p0 * 16 + 4 * q);
    acc += *reinterpret_cast<const f32x4 *>(Y + (size_t)p1 * 16 + 4 * q);
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main(int argc, char **argv) {
  const long long N = argc > 1 ? atoll(argv[1]) : 1666764, M = argc > 2 ? atoll(argv[2]) : 13643406;
  const int W = argc > 3 ? atoi(argv[3]) : 10;
  std::vector<int> h(M), hp(M);
  unsigned long long s = 88172645463325252ull;
  for (long long i = 0; i < M; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int)(s % N); }
  for (long long i = 0; i < M; ++i) hp[i] = (int)i;
  for (long long i = M - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; const long long j = s % (i + 1); std::swap(hp[i], hp[j]); }
  float *out, *Y, *sink; int *idx, *perm;
  hipMalloc(&out, N * 64); hipMalloc(&Y, M * 64); hipMalloc(&idx, M * 4); hipMalloc(&perm, M * 4); hipMalloc(&sink, 4096 * 256 * 4);
  hipMemset(out, 0, N * 64);
  hipMemcpy(idx, h.data(), M * 4, hipMemcpyHostToDevice);
  hipMemcpy(perm, hp.data(), M * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto time = [&](const char *name, auto launch) {
    for (int w = 0; w < 2; ++w) launch();
    hipEventRecord(a);
    for (int w = 0; w < 5; ++w) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-60s %.3f ms per pass (%.1f M rows, table of %.2f M rows)\n", name, ms / 5, M / 1e6, N / 1e6);
  };
  time("atomic add of W floats per random row (16 lanes per row)", [&] { hipLaunchKernelGGL(atomic_rows, dim3(4096), dim3(256), 0, 0, out, idx, M, W); });
  time("sequential write of 64-byte rows (the Y write)", [&] { hipLaunchKernelGGL(write_rows, dim3(4096), dim3(256), 0, 0, Y, M); });
  time("scattered write of 64-byte rows through a permutation", [&] { hipLaunchKernelGGL(scatter_rows, dim3(4096), dim3(256), 0, 0, Y, perm, M); });
  time("gather of 64-byte rows through a permutation (the row sum)", [&] { hipLaunchKernelGGL(gather_rows, dim3(4096), dim3(256), 0, 0, Y, perm, sink, M); });
  float chk[16]; hipMemcpy(chk, out, 64, hipMemcpyDeviceToHost);
  printf("out[0][0] = %.0f\n", chk[0]);
  return 0;
}
