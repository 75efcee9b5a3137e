// Windowed gather of 64-byte rows: the SAME 21 M row reads as tools/micro/gather64.hip (S1's X / G gather), but the order of the reads is
// window-major -- every workgroup walks the table's windows 0, 1, 2, ... in the same order and reads, inside window w, only rows of
// [w W, (w + 1) W).  Workgroup b runs on XCD b % 8 and every XCD has its own 4 MiB L2: while the workgroups of an XCD are in the same window,
// a row fetched by one of them is an L2 hit for the others (each row is read M / N = 21 times per pass, 2.6 times per XCD).  No synchronisation
// between workgroups: they drift as they would in a real kernel.  Question: how far below the 0.373 ms of the
// uniformly random gather does the windowed floor sit, per window size?
//   gather_window.bin [W rows | 0 = sweep] [U rows in flight per lane group: 4 | 8] [workgroups per CU] [N] [M]
// Prints ms per pass; run one window size under rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum / TCC_EA0_RDREQ_sum for the hit rate and the requests.
// build: make -C tools micro/gather_window.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(256) void gather_kernel(const float *__restrict__ X, const int *__restrict__ idx, float *__restrict__ out, long long per_wg) {
  const int q = threadIdx.x & 3, g = threadIdx.x >> 2;       // 4 lanes per row, 64 rows per workgroup instruction
  const int *my = idx + (long long)blockIdx.x * per_wg;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long long e = g; e < per_wg; e += 64 * U) {
    int row[U];
#pragma unroll
    for (int u = 0; u < U; ++u) row[u] = my[e + 64 * u < per_wg ? e + 64 * u : e];
    f32x4 r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = *reinterpret_cast<const f32x4 *>(X + (size_t)row[u] * 16 + 4 * q);
#pragma unroll
    for (int u = 0; u < U; ++u) acc += r[u];
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

static unsigned long long rng_state = 88172645463325252ull;
static inline unsigned long long rng() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main(int argc, char **argv) {
  const long long Wonly = argc > 1 ? atoll(argv[1]) : 0;
  const int U = argc > 2 ? atoi(argv[2]) : 4;
  const int per_cu = argc > 3 ? atoi(argv[3]) : 8;
  const long long N = argc > 4 ? atoll(argv[4]) : 1000000, M = argc > 5 ? atoll(argv[5]) : 21000000;
  const int wgs = 256 * per_cu;
  const long long per_wg = M / wgs;                // (the tail of M % wgs rows is dropped: < 0.01 %)
  float *X, *out; int *idx;
  hipMalloc(&X, N * 64); hipMalloc(&idx, per_wg * wgs * 4); hipMalloc(&out, (size_t)wgs * 256 * 4);
  hipMemset(X, 0, N * 64);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  std::vector<int> h(per_wg * wgs);
  const long long sweep[] = {N, 262144, 131072, 65536, 32768, 16384, 8192, 4096};
  for (long long W : sweep) {
    if (Wonly && W != Wonly && !(Wonly >= N && W == N)) continue;
    const long long nw = (N + W - 1) / W;
    // workgroup b, window w: its share of the window's reads, uniformly random rows of the window
    for (int bq = 0; bq < wgs; ++bq)
      for (long long i = 0; i < per_wg; ++i) {
        const long long w = i * nw / per_wg, lo = w * W, hi = (lo + W < N ? lo + W : N);
        h[bq * per_wg + i] = (int)(lo + rng() % (hi - lo));
      }
    hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
      if (U == 8) hipLaunchKernelGGL(gather_kernel<8>, dim3(wgs), dim3(256), 0, 0, X, idx, out, per_wg);
      else hipLaunchKernelGGL(gather_kernel<4>, dim3(wgs), dim3(256), 0, 0, X, idx, out, per_wg);
    }
    hipEventRecord(a);
    for (int rep = 0; rep < 5; ++rep) {
      if (U == 8) hipLaunchKernelGGL(gather_kernel<8>, dim3(wgs), dim3(256), 0, 0, X, idx, out, per_wg);
      else hipLaunchKernelGGL(gather_kernel<4>, dim3(wgs), dim3(256), 0, 0, X, idx, out, per_wg);
    }
    hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("window %8lld rows (%7.2f MB, %4lld windows), U=%d, %d workgroups/CU: %.3f ms per pass of %.1f M rows out of %.2f M (%s)\n", W, W * 64 / 1e6, nw, U,
           per_cu, ms / 5, per_wg * wgs / 1e6, N / 1e6, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
