// Micro-benchmark (tools/): VALU-only throughput of v_fma_f32 vs v_pk_fma_f32 (with VGPR and with SGPR multiplicands),
// whole chip, W waves per SIMD, wall clock.  Question behind it: is contour conv2 (100 packed FMAs per row and 64 bins,
// no matrix work) paced by its FMA stream?
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x2 = __attribute__((ext_vector_type(2))) float;

template <int KIND>
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters) {
  f32x2 acc[8];
  float a1[16];
  for (int i = 0; i < 8; ++i) acc[i] = f32x2{in[threadIdx.x & 63], in[(threadIdx.x + i) & 63]};
  for (int i = 0; i < 16; ++i) a1[i] = in[(threadIdx.x + 3 * i) & 63];
  f32x2 x = {in[1], in[2]};
  const float s0 = in[blockIdx.x & 7], s1 = in[(blockIdx.x + 1) & 7];  // wave-uniform: SGPRs
  const f32x2 sx = {s0, s1};
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(acc[(i + 1) & 7]));
        if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "s"(sx), "v"(acc[(i + 1) & 7]));
        if (KIND == 2) {
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a1[2 * i]) : "v"(x.x), "v"(a1[(2 * i + 2) & 15]));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a1[2 * i + 1]) : "v"(x.y), "v"(a1[(2 * i + 3) & 15]));
        }
        if (KIND == 3) {
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a1[2 * i]) : "s"(s0), "v"(a1[(2 * i + 2) & 15]));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a1[2 * i + 1]) : "s"(s1), "v"(a1[(2 * i + 3) & 15]));
        }
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
  for (int i = 0; i < 16; ++i) s += a1[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, int wg_per_cu) {
  float *in, *out;
  (void)hipMalloc(&in, 256);
  (void)hipMemset(in, 0, 256);
  const int grid = 256 * wg_per_cu;
  (void)hipMalloc(&out, grid * 256 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, in, out, 2000);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, in, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double fma_pairs = (double)grid * 4 * iters * 64;  // per wave: 64 packed FMAs (= 128 plain) per iteration
  printf("%-46s waves/SIMD %d: %8.3f ms  %6.2f ns per packed-FMA-equivalent per SIMD  %6.1f TFLOP/s\n", name, wg_per_cu, ms,
         ms * 1e6 / (fma_pairs / 1024.0), fma_pairs * 64 * 4 / ms / 1e9);
}

int main() {
  for (int w : {1, 2, 5}) {
    run<0>("v_pk_fma_f32 (VGPR x VGPR)", w);
    run<1>("v_pk_fma_f32 (SGPR pair x VGPR)", w);
    run<2>("2 x v_fma_f32 (VGPR x VGPR)", w);
    run<3>("2 x v_fma_f32 (SGPR x VGPR)", w);
  }
  return 0;
}
