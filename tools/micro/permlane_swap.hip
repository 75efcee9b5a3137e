// what v_permlane32_swap / v_permlane16_swap do on gfx950 (x = lane, y = 100 + lane): prints both results per lane
// build: hipcc -O2 --offload-arch=gfx950 -o tools/micro/permlane_swap.bin tools/micro/permlane_swap.hip   (run on the GPU box: gpurun -- tools/micro/permlane_swap.bin)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o) {
  const unsigned x = threadIdx.x, y = 100 + threadIdx.x;
  const auto a = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  o[threadIdx.x] = a[0]; o[64 + threadIdx.x] = a[1]; o[128 + threadIdx.x] = b[0]; o[192 + threadIdx.x] = b[1];
}
int main() {
  unsigned *d, h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char *names[4] = {"swap32 r0", "swap32 r1", "swap16 r0", "swap16 r1"};
  for (int r = 0; r < 4; ++r) {
    printf("%s:", names[r]);
    for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[64 * r + l]);
    printf("\n");
  }
  return 0;
}
