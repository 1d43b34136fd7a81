// Micro-benchmark (tools/): what a v_mfma_f32_16x16x16_f16 costs beside v_mfma_f32_16x16x32_f16 on gfx950 — if the K = 16 form
// took half the time, the half-empty last k-step of the marches (conv1: K = 176 = 5.5 x 32; onset: 200 = 6.25 x 32; rim: 432 =
// 13.5 x 32) could be issued in it.  Four independent accumulator chains per wave, operands in registers, 4 and 8 waves per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int KIND>  // 0: 16x16x32, 1: 16x16x16, 2: five of the first and one of the second per group (a 5.5-step row)
__global__ __launch_bounds__(512) void k(const uint4* in, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  const f16x8 a = __builtin_bit_cast(f16x8, in[lane]), b = __builtin_bit_cast(f16x8, in[64 + lane]);
  const f16x4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
  f32x4 acc[4];
  for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 6; ++s)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (KIND == 0 || (KIND == 2 && s < 5))
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[c], 0, 0, 0);
        else
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[c], 0, 0, 0);
      }
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

template <int KIND>
void run(int threads, const char* name) {
  uint4* in;
  float* out;
  (void)hipMalloc(&in, 128 * 16);
  (void)hipMalloc(&out, 256 * 512 * 4);
  unsigned short h[128 * 8];
  for (int i = 0; i < 128 * 8; ++i) h[i] = (unsigned short)(((rand() & 1) << 15) | ((12 + (rand() & 3)) << 10) | (rand() & 1023));
  (void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(threads), 0, 0, in, out, 4000);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(threads), 0, 0, in, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double per = ms * 1e6 / iters / 24 / (threads / 256.0);  // ns per matrix instruction per SIMD
  printf("%-34s %d waves/CU: %7.3f ms, %5.2f ns per matrix instruction and SIMD\n", name, threads / 64, ms, per);
}

int main() {
  for (int threads : {256, 512}) {
    run<0>(threads, "16x16x32 only");
    run<1>(threads, "16x16x16 only");
    run<2>(threads, "5 x 16x16x32 + 1 x 16x16x16 per row");
  }
  return 0;
}
