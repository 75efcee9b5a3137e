// Zero-fill helper shared by every translation unit of librgcn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

namespace {

// Zero-fill as a KERNEL, never hipMemsetAsync: inside a captured hipGraph a memset becomes a memset node, and the HIP runtime
// PyTorch 2.10 bundles (ROCm 7.0.51831, AQL packet capture of graph nodes on by default) replays memset nodes with stale
// arguments once eager kernels have run between two replays -- the "memset" then writes garbage to another address
// (tools/hipgraph_repro/memset_node.py reproduces it without this library; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 hides it).
// Kernel nodes replay correctly.  `bytes` is a multiple of 4 everywhere in this library; any 4-byte alignment.
__global__ __launch_bounds__(256) void zero_fill_kernel(uint32_t *__restrict__ p, size_t head, size_t body, size_t tail) {
  const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  uint4 *v = reinterpret_cast<uint4 *>(p + head);
  for (size_t i = i0; i < body; i += stride) v[i] = uint4{0u, 0u, 0u, 0u};
  if (i0 < head) p[i0] = 0u;
  if (i0 < tail) p[head + 4 * body + i0] = 0u;
}

inline hipError_t zero_async(void *ptr, size_t bytes, hipStream_t st) {
  if (bytes == 0) return hipSuccess;
  const size_t words = bytes / 4;
  const size_t head = std::min(words, (size_t)(((16 - (reinterpret_cast<uintptr_t>(ptr) & 15)) & 15) / 4));
  const size_t body = (words - head) / 4, tail = words - head - 4 * body;
  const unsigned grid = (unsigned)std::min<size_t>(std::max<size_t>((body + 255) / 256, 1), 256 * 16);
  hipLaunchKernelGGL(zero_fill_kernel, dim3(grid), dim3(256), 0, st, static_cast<uint32_t *>(ptr), head, body, tail);
  return hipGetLastError();
}

}  // namespace
