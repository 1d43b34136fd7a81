// Micro-benchmark (tools/, not part of the library): ds_read_b128 throughput per CU with and without MFMAs
// in flight, one wave per SIMD (256 threads, 1 block per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int READS, int MFMAS, int STRIDE16>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) uint4 lds[8192];  // 128 KB
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = uint4{(unsigned)i, 1u, 2u, 3u};
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int base = w * 1024 + lane * STRIDE16;
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  uint4 sum{0, 0, 0, 0};
  f16x8 af = __builtin_bit_cast(f16x8, lds[lane]);
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    uint4 v[READS > 0 ? READS : 1];
#pragma unroll
    for (int r = 0; r < READS; ++r) v[r] = lds[(base + r * 64 + (it & 7) * 16) & 8191];
#pragma unroll
    for (int m = 0; m < MFMAS; ++m) {
      f16x8 bf = READS > 0 ? __builtin_bit_cast(f16x8, v[m % (READS > 0 ? READS : 1)]) : af;
      acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[m & 3], 0, 0, 0);
    }
    if (MFMAS == 0) {
#pragma unroll
      for (int r = 0; r < READS; ++r) {
        sum.x += v[r].x;
        sum.y ^= v[r].y;
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = (float)sum.x + (float)sum.y;
  for (int a = 0; a < 4; ++a) s += acc[a][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int READS, int MFMAS, int STRIDE16>
void run(const char* name) {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4);
  hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipLaunchKernelGGL((k<READS, MFMAS, STRIDE16>), dim3(256), dim3(256), 0, 0, out, iters, cyc);
  hipLaunchKernelGGL((k<READS, MFMAS, STRIDE16>), dim3(256), dim3(256), 0, 0, out, iters, cyc);
  unsigned long long h = 0;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  double per_it = (double)h / iters;
  printf("%-28s reads/iter/wave %d mfma/iter/wave %d : %.1f cycles/iter", name, READS, MFMAS, per_it);
  if (READS) printf("  -> %.2f cycles per wave-read at CU level (4 waves)", per_it / (4.0 * READS));
  if (MFMAS) printf("  mfma pipe %.0f %%", 100.0 * MFMAS * 32 / per_it);
  printf("\n");
  hipFree(out);
  hipFree(cyc);
}

int main() {
  run<8, 0, 1>("reads only, contiguous");
  run<8, 0, 2>("reads only, stride 32B");
  run<0, 8, 1>("mfma only");
  run<2, 8, 1>("2 reads + 8 mfma");
  run<4, 8, 1>("4 reads + 8 mfma");
  run<6, 6, 1>("6 reads + 6 mfma");
  run<8, 8, 1>("8 reads + 8 mfma");
  run<4, 4, 1>("4 reads + 4 mfma");
  return 0;
}
