// Micro-benchmark (tools/): what a pure streaming READ (and copy) sustains on this box — the ceiling the HBM-paced
// kernels (contour conv2: 465 MB in ~105 us = 4.4 TB/s) are measured against.  512 MiB buffers (beyond the 256 MB
// Infinity Cache), 16 bytes per lane, W waves per SIMD, grid-stride; wall clock from HIP events.
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void rd(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n, int unroll_dummy) {
  uint4 acc = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += 4 * stride) {
    uint4 v0 = in[i], v1 = i + stride < n ? in[i + stride] : acc, v2 = i + 2 * stride < n ? in[i + 2 * stride] : acc,
          v3 = i + 3 * stride < n ? in[i + 3 * stride] : acc;
    acc.x ^= v0.x ^ v1.x ^ v2.x ^ v3.x;
    acc.y ^= v0.y ^ v1.y ^ v2.y ^ v3.y;
    acc.z ^= v0.z ^ v1.z ^ v2.z ^ v3.z;
    acc.w ^= v0.w ^ v1.w ^ v2.w ^ v3.w;
  }
  if (acc.x == 0x12345678u && unroll_dummy) out[threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void cp(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = in[i];
}

int main() {
  const size_t bytes = (size_t)512 << 20, n = bytes / 16;
  uint4 *a, *b;
  (void)hipMalloc(&a, bytes);
  (void)hipMalloc(&b, bytes);
  (void)hipMemset(a, 1, bytes);
  (void)hipMemset(b, 2, bytes);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int wps : {2, 4, 8}) {
    const int grid = 256 * wps;
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, a, b, n, 0);
      (void)hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, i & 1 ? a : b, b, n, 0);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      printf("read  512 MiB, %d waves/SIMD: %7.3f ms per pass  %6.2f TB/s\n", wps, ms / 10, bytes * 10.0 / ms / 1e9);
    }
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(cp, dim3(grid), dim3(256), 0, 0, a, b, n);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("copy  512 MiB, %d waves/SIMD: %7.3f ms per pass  %6.2f TB/s (read + write)\n", wps, ms / 10, 2.0 * bytes * 10.0 / ms / 1e9);
  }
  return 0;
}
