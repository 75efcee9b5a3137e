// What does the chip deliver for v_mfma_f32_16x16x4_f32?  Every wave issues ITER x 4 x CH MFMAs on CH independent
// accumulators from registers only; prints TFLOP/s, shader cycles per MFMA per SIMD (s_memtime) and the implied clock.
//   hipcc -O3 --offload-arch=gfx950 -o tools/mfma_probe.bin tools/mfma_probe.hip && ./tools/mfma_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ __launch_bounds__(256) void probe(float *out, unsigned long long *cyc, int iters) {
  f32x4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CH>
void run(int waves_per_simd) {
  const int blocks = 256 * waves_per_simd, iters = 4000;     // 4 waves per block -> one per SIMD
  float *out; unsigned long long *cyc;
  hipMalloc(&out, blocks * 256 * sizeof(float));
  hipMalloc(&cyc, blocks * sizeof(unsigned long long));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<CH><<<blocks, 256>>>(out, cyc, 10);
  hipEventRecord(e0);
  probe<CH><<<blocks, 256>>>(out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += double(v); mean /= blocks;
  const double n_mfma = double(blocks) * 4 * iters * 4 * CH, flops = n_mfma * 2048.0;
  printf("chains=%d waves/SIMD=%d: %.3f ms  %.1f TFLOP/s  wave-loop %.0f ticks (%.1f ticks per MFMA of one wave)\n", CH,
         waves_per_simd, ms, flops / ms / 1e9, mean, mean / (iters * 4.0 * CH));
  hipFree(out); hipFree(cyc);
}

int main() {
  run<1>(1); run<2>(1); run<4>(1); run<8>(1);
  run<4>(2); run<4>(4); run<8>(2); run<1>(8); run<2>(4);
  return 0;
}
