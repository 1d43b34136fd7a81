// Micro-benchmark (tools/): sustained MFMA rate when the B operands stream from LDS (random f16 data), whole chip,
// wall clock.  Design points of contour conv1:
//   A: 32x32x16, 8 waves/CU, per k-step 4 ds_read_b128 (A hi, A lo, B hi, B lo) -> 3 MFMAs
//   B: 16x16x32, 4 waves/CU, A in registers, per k-step 8 ds_read_b128 (4 tiles x B hi, B lo) -> 12 MFMAs
//   C: 32x32x16, 4 waves/CU, A in registers, per k-step 4 ds_read_b128 (2 tiles x B hi, B lo) -> 6 MFMAs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int MODE>
__global__ __launch_bounds__(MODE == 0 ? 512 : 256, MODE == 0 ? 2 : 1) void k(const uint4* in, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) uint4 lds[8192];
  constexpr int NT = MODE == 0 ? 512 : 256;
  for (int i = threadIdx.x; i < 8192; i += NT) lds[i] = in[i & 1023];
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  f16x8 areg[8];
  for (int i = 0; i < 8; ++i) areg[i] = __builtin_bit_cast(f16x8, in[i * 64 + lane]);
  f32x16 acc[4];
  f32x4 acc4[8];
  for (int x = 0; x < 4; ++x)
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
  for (int x = 0; x < 8; ++x)
    for (int r = 0; r < 4; ++r) acc4[x][r] = 0.f;
  int base = (w * 640 + lane) & 8191;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int o = (base + s * 64 + (it & 15) * 32) & 8191;
      if (MODE == 0) {
        f16x8 ah = __builtin_bit_cast(f16x8, lds[o]);
        f16x8 al = __builtin_bit_cast(f16x8, lds[(o + 2048) & 8191]);
        f16x8 bh = __builtin_bit_cast(f16x8, lds[(o + 4096) & 8191]);
        f16x8 bl = __builtin_bit_cast(f16x8, lds[(o + 6144) & 8191]);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[0], 0, 0, 0);
      } else if (MODE == 2) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          f16x8 bh = __builtin_bit_cast(f16x8, lds[(o + t * 1024) & 8191]);
          f16x8 bl = __builtin_bit_cast(f16x8, lds[(o + t * 1024 + 4096) & 8191]);
          acc[2 * t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[1], bh, acc[2 * t], 0, 0, 0);
          acc[2 * t + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[0], bh, acc[2 * t + 1], 0, 0, 0);
          acc[2 * t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[0], bl, acc[2 * t], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          f16x8 bh = __builtin_bit_cast(f16x8, lds[(o + t * 1024) & 8191]);
          f16x8 bl = __builtin_bit_cast(f16x8, lds[(o + t * 1024 + 4096) & 8191]);
          acc4[2 * t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(areg[(2 * s + 1) & 7], bh, acc4[2 * t], 0, 0, 0);
          acc4[2 * t + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(areg[(2 * s) & 7], bh, acc4[2 * t + 1], 0, 0, 0);
          acc4[2 * t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(areg[(2 * s) & 7], bl, acc4[2 * t], 0, 0, 0);
        }
      }
    }
  }
  float sum = 0;
  for (int x = 0; x < 4; ++x) sum += acc[x][0] + acc[x][7];
  for (int x = 0; x < 8; ++x) sum += acc4[x][0];
  out[blockIdx.x * NT + threadIdx.x] = sum;
}

template <int MODE>
void run(const char* name) {
  uint4* in;
  float* out;
  (void)hipMalloc(&in, 1024 * 16);
  (void)hipMalloc(&out, 256 * 512 * 4);
  unsigned short h[1024 * 8];
  for (int i = 0; i < 1024 * 8; ++i)
    h[i] = (unsigned short)(((rand() & 1) << 15) | ((10 + (rand() & 3)) << 10) | (rand() & 1023));
  (void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  constexpr int NT = MODE == 0 ? 512 : 256;
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(NT), 0, 0, in, out, 4000);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(NT), 0, 0, in, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double mf = MODE == 0 ? 3 * 32768.0 : MODE == 2 ? 6 * 32768.0 : 12 * 16384.0;
  const double flop = 256.0 * (NT / 64) * iters * 8 * mf;
  printf("%-64s %.3f ms  %.0f TFLOP/s\n", name, ms, flop / ms / 1e9);
}

int main() {
  run<0>("A: 32x32x16, 8 waves, 4 reads (A+B from LDS) per 3 MFMA");
  run<2>("C: 32x32x16, 4 waves, A in regs, 4 reads per 6 MFMA");
  run<1>("B: 16x16x32, 4 waves, A in regs, 8 reads per 12 MFMA");
  run<0>("A again");
  return 0;
}
