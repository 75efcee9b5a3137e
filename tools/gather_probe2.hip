// Probe: can a random 64-byte row gather be served with 64-byte (not 128-byte) fabric requests?  Cache-policy variants
// of the same 16-bytes-per-lane gather (plain, non-temporal, buffer loads with sc0 / sc1 / nt bits).
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_probe2.hip -o tools/gather_probe2.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void gather(const f4* __restrict__ tab, const int* __restrict__ idx, float* out, long n_rows_to_read, long table_bytes) {
  const int lane = threadIdx.x & 63, m = lane & 15, k = lane >> 4;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * 4;
  constexpr int U = 4;
  f4 acc = {0, 0, 0, 0};
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)tab, 0, (int)table_bytes, 0x00020000);
  for (long c = wave * U; c * 16 < n_rows_to_read; c += nw * U) {
    int s[U];
#pragma unroll
    for (int j = 0; j < U; ++j) s[j] = idx[(c + j) * 16 + m];
    f4 x[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      if (MODE == 0) x[j] = tab[(size_t)s[j] * 4 + k];
      else if (MODE == 1) x[j] = __builtin_nontemporal_load(&tab[(size_t)s[j] * 4 + k]);
      else {
        constexpr int AUX = MODE - 100;
        i4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (s[j] * 4 + k) * 16, 0, AUX);
        x[j] = __builtin_bit_cast(f4, r);
      }
    }
#pragma unroll
    for (int j = 0; j < U; ++j) acc += x[j];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

template <int MODE>
float run(const f4* tab, const int* idx, float* out, long n, long bytes) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(gather<MODE>, dim3(2048), dim3(256), 0, 0, tab, idx, out, n, bytes);
  CK(hipEventRecord(a));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(gather<MODE>, dim3(2048), dim3(256), 0, 0, tab, idx, out, n, bytes);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 5;
}

int main() {
  const long n_reads = 24L << 20, rows = 1L << 20;
  f4* tab; int* idx; float* out;
  CK(hipMalloc(&tab, rows * 64)); CK(hipMalloc(&idx, n_reads * 4)); CK(hipMalloc(&out, 4));
  CK(hipMemset(tab, 0, rows * 64));
  std::vector<int> h(n_reads);
  unsigned long long x = 88172645463325252ull;
  for (long i = 0; i < n_reads; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (int)(x % (unsigned long long)rows); }
  CK(hipMemcpy(idx, h.data(), n_reads * 4, hipMemcpyHostToDevice));
#define RUN(M, name) { float t = run<M>(tab, idx, out, n_reads, rows * 64); printf("%-28s %.3f ms  %.0f GB/s useful\n", name, t, n_reads * 64 / t / 1e6); }
  RUN(0, "plain global_load");
  RUN(1, "nontemporal");
  RUN(100, "buffer aux=0");
  RUN(101, "buffer aux=1 (sc0)");
  RUN(102, "buffer aux=2 (nt)");
  RUN(103, "buffer aux=3 (sc0 nt)");
  RUN(116, "buffer aux=16 (sc1)");
  RUN(117, "buffer aux=17 (sc0 sc1)");
  RUN(118, "buffer aux=18 (sc1 nt)");
  RUN(119, "buffer aux=19 (sc0 sc1 nt)");
  return 0;
}
