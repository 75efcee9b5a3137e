// Which address patterns does the gfx950 LDS take at full rate for an atomic?  (round 4: the dX tile update of bwd_blk_d16_kernel)
// Every pattern = 4 tables of 64 byte offsets (one per instruction of the unrolled body); 16 waves per CU, 256 workgroups.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_cas_patterns tools/micro/lds_cas_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <functional>

enum { OP_CAS32 = 0, OP_CAS64 = 1, OP_ADDF64 = 2, OP_READ32 = 3, OP_ADDF32 = 4, OP_ADDU32 = 5, OP_RMW32 = 6 };

template <int OP>
__global__ __launch_bounds__(1024) void k(float *out, const int *offs, int iters) {
  extern __shared__ __attribute__((aligned(16))) char buf[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 65536 / 4; i += 1024) reinterpret_cast<float *>(buf)[i] = 0.f;
  __syncthreads();
  int o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = offs[e * 64 + lane];
  float v = 1.0f + lane;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (o[e] < 0) continue;                                  // inactive lane (pad slot)
      char *p = buf + o[e];
      if (OP == OP_CAS32) {
        unsigned ex = 0, de = (unsigned)i;
        __hip_atomic_compare_exchange_strong(reinterpret_cast<unsigned *>(p), &ex, de, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        v += (float)ex;
      } else if (OP == OP_CAS64) {
        unsigned long long ex = 0, de = (unsigned long long)i;
        __hip_atomic_compare_exchange_strong(reinterpret_cast<unsigned long long *>(p), &ex, de, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        v += (float)ex;
      } else if (OP == OP_ADDF64) {
        __hip_atomic_fetch_add(reinterpret_cast<double *>(p), (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (OP == OP_READ32) {
        v += __hip_atomic_load(reinterpret_cast<float *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (OP == OP_ADDF32) {
        __hip_atomic_fetch_add(reinterpret_cast<float *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (OP == OP_ADDU32) {
        __hip_atomic_fetch_add(reinterpret_cast<unsigned *>(p), (unsigned)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        float x = __hip_atomic_load(reinterpret_cast<float *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(reinterpret_cast<float *>(p), x + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  __syncthreads();
  out[blockIdx.x * 1024 + tid] = reinterpret_cast<float *>(buf)[tid] + v;
}

static float *g_out;
static int *g_offs;

template <int OP>
void run(const char *name, const std::function<int(int e, int lane)> &f) {
  std::vector<int> h(256);
  for (int e = 0; e < 4; ++e)
    for (int l = 0; l < 64; ++l) h[e * 64 + l] = f(e, l);
  hipMemcpy(g_offs, h.data(), 256 * 4, hipMemcpyHostToDevice);
  const int iters = 2000;
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL((k<OP>), dim3(256), dim3(1024), 65536, 0, g_out, g_offs, iters);
  hipDeviceSynchronize();
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<OP>), dim3(256), dim3(1024), 65536, 0, g_out, g_offs, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double instr = 16.0 * iters * 4;
  printf("%-74s %8.3f ms -> %6.1f cycles per wave instruction per CU\n", name, ms, ms * 1e6 / instr * 2.4);
}

int main() {
  hipMalloc(&g_out, 256 * 1024 * 4);
  hipMalloc(&g_offs, 256 * 4);
  srand(1);
  // lane = 16 q + o.  "column" pattern: lane (q, o) touches element o of row rows[q] (row = 64 bytes of floats)
  auto col32 = [](const int (*rows)[4]) {
    return [rows](int e, int l) { const int r = rows[e][l >> 4]; return r < 0 ? -1 : r * 64 + (l & 15) * 4; };
  };
  static const int distinct[4][4] = {{0, 5, 10, 15}, {17, 22, 27, 28}, {32, 37, 42, 47}, {51, 52, 57, 62}};
  static const int two_same[4][4] = {{0, 4, 10, 15}, {17, 21, 27, 28}, {32, 36, 42, 47}, {51, 55, 57, 62}};
  static const int all_same[4][4] = {{0, 4, 8, 12}, {17, 21, 25, 29}, {32, 36, 40, 44}, {51, 55, 59, 63}};
  static const int dup_row[4][4] = {{0, 0, 10, 15}, {17, 17, 27, 28}, {32, 32, 42, 47}, {51, 51, 57, 62}};
  static const int one_pad[4][4] = {{0, 5, -1, 15}, {17, -1, 27, 28}, {-1, 37, 42, 47}, {51, 52, 57, -1}};
  static const int pad_same[4][4] = {{0, 4, -1, 15}, {17, -1, 27, 31}, {-1, 37, 42, 46}, {51, 55, 57, -1}};
  run<OP_CAS32>("cas_b32 column, 4 rows of distinct class (row % 4)", col32(distinct));
  run<OP_CAS32>("cas_b32 column, two rows share a class", col32(two_same));
  run<OP_CAS32>("cas_b32 column, all rows share a class", col32(all_same));
  run<OP_CAS32>("cas_b32 column, two lane groups hit the SAME row", col32(dup_row));
  run<OP_CAS32>("cas_b32 column, distinct classes, one lane group inactive", col32(one_pad));
  run<OP_CAS32>("cas_b32 column, one group inactive, two of the others share a class", col32(pad_same));
  run<OP_READ32>("read_b32 column, distinct classes", col32(distinct));
  run<OP_READ32>("read_b32 column, two rows share a class", col32(two_same));
  run<OP_READ32>("read_b32 column, all rows share a class", col32(all_same));
  run<OP_RMW32>("read+add+write b32 column, distinct classes", col32(distinct));
  run<OP_ADDU32>("ds_add_u32 column, distinct classes", col32(distinct));
  run<OP_ADDU32>("ds_add_u32 column, two rows share a class", col32(two_same));
  run<OP_ADDF32>("ds_add_f32 column, distinct classes", col32(distinct));
  // ds_add_f32 with few active lanes: is it 3 cycles per ACTIVE lane?
  run<OP_ADDF32>("ds_add_f32 contiguous, 16 active lanes", [](int e, int l) { return (l & 3) ? -1 : e * 256 + l * 4; });
  run<OP_ADDF32>("ds_add_f32 contiguous, 64 active lanes", [](int e, int l) { return e * 256 + l * 4; });
  // double-precision tile: row = 128 bytes; lane (q, o) adds element o of row rows[q]
  auto col64 = [](const int (*rows)[4]) {
    return [rows](int e, int l) { const int r = rows[e][l >> 4]; return r < 0 ? -1 : r * 128 + (l & 15) * 8; };
  };
  static const int d_alt[4][4] = {{0, 5, 10, 15}, {16, 21, 26, 31}, {32, 37, 42, 47}, {48, 53, 58, 63}};       // parities 0 1 0 1
  static const int d_pair[4][4] = {{0, 4, 9, 15}, {16, 22, 27, 31}, {32, 36, 43, 47}, {48, 52, 59, 63}};       // parities 0 0 1 1
  static const int d_same[4][4] = {{0, 4, 8, 12}, {16, 20, 24, 28}, {32, 36, 40, 44}, {48, 52, 56, 60}};       // all even
  run<OP_ADDF64>("ds_add_f64 column, row parities 0 1 0 1", col64(d_alt));
  run<OP_ADDF64>("ds_add_f64 column, row parities 0 0 1 1", col64(d_pair));
  run<OP_ADDF64>("ds_add_f64 column, all rows even", col64(d_same));
  run<OP_ADDF64>("ds_add_f64 contiguous", [](int e, int l) { return e * 512 + l * 8; });
  run<OP_ADDF64>("ds_add_f64 contiguous, 16 active lanes", [](int e, int l) { return (l & 3) ? -1 : e * 512 + l * 8; });
  run<OP_ADDF64>("ds_add_f64 16-way conflict, 16 active lanes", [](int e, int l) { return (l & 3) ? -1 : e * 8 + (l >> 2) * 256; });
  run<OP_ADDF64>("ds_add_f64 16-way conflict, 64 active lanes", [](int e, int l) { return e * 8 + (l & 15) * 256 + (l >> 4) * 1024; });
  // 64-bit CAS: is the conflict rule per half wave (lanes 0-31 / 32-63)?  halves conflict-free inside, same banks across the halves
  run<OP_CAS64>("cas_b64 each half-wave conflict-free, halves share banks", [](int e, int l) { return e * 1024 + (l & 31) * 8 + (l >> 5) * 256; });
  run<OP_CAS64>("cas_b64 contiguous", [](int e, int l) { return e * 512 + l * 8; });
  run<OP_CAS64>("cas_b64 lanes l and l+16 share banks (2-way inside a half)", [](int e, int l) { return e * 1024 + (l & 15) * 8 + ((l >> 4) & 1) * 256 + (l >> 5) * 128; });
  // 64-bit column pattern of a float tile after a lane swap: lane (q, o): row rows[q >> 1 ...]: 8 slots x 4 k per half -- skip
  // random rows (what an unarranged chunk looks like): 4 rows uniform in 0..255
  static int rnd[8][4][4];
  for (int t = 0; t < 8; ++t)
    for (int e = 0; e < 4; ++e)
      for (int q = 0; q < 4; ++q) rnd[t][e][q] = rand() % 256;
  for (int t = 0; t < 4; ++t) {
    char nm[128];
    int nconf = 0;
    for (int e = 0; e < 4; ++e) {
      int seen = 0, c = 0;
      for (int q = 0; q < 4; ++q) { const int g = rnd[t][e][q] & 3; if (seen & (1 << g)) c = 1; seen |= 1 << g; }
      nconf += c;
    }
    snprintf(nm, sizeof nm, "cas_b32 column, random rows, draw %d (%d of 4 instructions have a class clash)", t, nconf);
    run<OP_CAS32>(nm, col32(rnd[t]));
  }
  return 0;
}
