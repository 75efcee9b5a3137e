// LDS atomic throughput on gfx950: cycles per wave-level instruction for ds_add_f32 / ds_add_u32 / ds_write_b32 / read+add+write,
// 16 waves per CU, every lane its own word (conflict-free), per-wave private or all waves on the same 64 words.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o /tmp/lds_atomic_rate tools/micro/lds_atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, bool SHARED>
__global__ __launch_bounds__(1024) void k(float *out, long long *cyc, int iters) {
  __shared__ float buf[16 * 64 * 4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 16 * 64 * 4; i += 1024) buf[i] = 0.f;
  __syncthreads();
  float *p = buf + (SHARED ? 0 : wave * 256) + lane;
  float v = 1.0f + lane;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (MODE == 0) __hip_atomic_fetch_add(p + 64 * e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (MODE == 1) __hip_atomic_fetch_add(reinterpret_cast<unsigned *>(p + 64 * e), (unsigned)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (MODE == 2) { __hip_atomic_store(p + 64 * e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
      else if (MODE == 4 || MODE == 5 || MODE == 7) {   // 64-bit compare-and-swap: lane-contiguous (4), dX-tile pattern row * 64 + k * 16 bytes (5: rows 5 m; 7: rows 4 m)
        unsigned long long *q = MODE == 4 ? reinterpret_cast<unsigned long long *>(buf + (SHARED ? 0 : wave * 256)) + lane + 64 * (e & 1)
                                          : reinterpret_cast<unsigned long long *>(buf + ((lane & 15) * (MODE == 5 ? 5 : 4)) * 16 + (lane >> 4) * 4) + (e & 1);
        unsigned long long ex = 0, de = (unsigned long long)i;
        __hip_atomic_compare_exchange_strong(q, &ex, de, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        v += (float)ex;
      } else if (MODE == 8 || MODE == 9) {   // double-precision LDS atomic add: contiguous (8), dX-tile-like pattern (9: rows 5 m, 4 lanes per row)
        double *q = MODE == 8 ? reinterpret_cast<double *>(buf + (SHARED ? 0 : wave * 256)) + lane + 64 * (e & 1)
                              : reinterpret_cast<double *>(buf) + ((lane & 15) * 5) * 8 + (lane >> 4) * 2 + (e & 1);
        __hip_atomic_fetch_add(q, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (MODE == 10) {               // 16 active lanes only, conflicting CAS (does the conflict cost scale with active lanes?)
        if ((lane & 3) == 0) {
          unsigned long long *q = reinterpret_cast<unsigned long long *>(buf + ((lane & 15) * 4) * 16 + (lane >> 4) * 4) + (e & 1);
          unsigned long long ex = 0, de = (unsigned long long)i;
          __hip_atomic_compare_exchange_strong(q, &ex, de, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          v += (float)ex;
        }
      } else if (MODE == 11 || MODE == 12) {   // conflict-FREE compare-and-swap with few active lanes: 16 (11), 32 (12)
        if ((MODE == 11 && (lane & 3) == 0) || (MODE == 12 && (lane & 1) == 0)) {
          unsigned long long *q = reinterpret_cast<unsigned long long *>(buf) + (MODE == 11 ? (lane >> 2) * 2 : (lane >> 1)) + 32 * (e & 1);
          unsigned long long ex = 0, de = (unsigned long long)i;
          __hip_atomic_compare_exchange_strong(q, &ex, de, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          v += (float)ex;
        }
      } else if (MODE == 6) {
        unsigned *q = reinterpret_cast<unsigned *>(p + 64 * e);
        unsigned ex = 0, de = (unsigned)i;
        __hip_atomic_compare_exchange_strong(q, &ex, de, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        v += (float)ex;
      }
      else { float x = __hip_atomic_load(p + 64 * e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_store(p + 64 * e, x + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    }
  }
  __syncthreads();
  long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 1024 + tid] = buf[tid] + v;
}
template <int MODE, bool SHARED> void run(const char *name) {
  float *out; long long *cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 2000;
  hipLaunchKernelGGL((k<MODE, SHARED>), dim3(256), dim3(1024), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<MODE, SHARED>), dim3(256), dim3(1024), 0, 0, out, cyc, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  // per CU: 16 waves x iters x 4 instructions; memtime ticks at 100 MHz -> use wall time and 2.4 GHz
  const double instr = 16.0 * iters * 4;
  printf("%-28s %8.3f ms  -> %.1f ns per wave instruction per CU (%.1f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
}
int main() {
  run<0, false>("ds_add_f32 private");
  run<0, true>("ds_add_f32 shared");
  run<1, false>("ds_add_u32 private");
  run<1, true>("ds_add_u32 shared");
  run<2, false>("ds_write_b32 private");
  run<3, false>("read+add+write private");
  run<4, false>("cas_b64 contiguous private");
  run<4, true>("cas_b64 contiguous shared");
  run<5, true>("cas_b64 rows 5m (tile pattern)");
  run<7, true>("cas_b64 rows 4m (16-way)");
  run<6, false>("cas_b32 contiguous private");
  run<8, false>("ds_add_f64 contiguous private");
  run<9, true>("ds_add_f64 tile pattern");
  run<10, true>("cas_b64 conflicting, 16 active lanes");
  run<11, true>("cas_b64 conflict-free, 16 active lanes");
  run<12, true>("cas_b64 conflict-free, 32 active lanes");
  return 0;
}
