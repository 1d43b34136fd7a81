// Micro-benchmark (tools/): variant D of the round-3 review — v_mfma_f32_16x16x32_f16 with the WEIGHTS (A, hi + lo,
// two 16-row M blocks, all k-steps) resident in registers and only the activations (B hi / B lo) streaming from LDS,
// one ds_read_b128 per B fragment feeding 6 matrix instructions (48 KFLOP per read, the LDS load of mfma_lds.hip's
// variant C).  The shape is the onset conv1 (M = 32 channels, K = 200 -> 7 k-steps of 32): a wave owns NT 16-pixel
// column tiles and per k-step issues 2 NT reads and 6 NT matrix instructions on 4 NT accumulators (hi.hi chains and
// correction chains apart, the kernels' form).  Same operand data in every mode, whole chip, wall clock.
//   mode 0  A32 : 32x32x16, A and B from LDS, 4 reads per 3 MFMA (variant A of mfma_lds.hip; 8 waves per CU)
//   mode 1  R32 : 32x32x16 from registers only (what mfma_power.hip measures), 3 MFMA per k-step, 13 k-steps
//   mode 2  R16 : 16x16x32 from registers only, 12 MFMA per k-step, 7 k-steps
//   mode 3  C32 : 32x32x16, A in registers (13 k-steps x hi, lo = 104 VGPRs), B from LDS: 2 reads per 3 MFMA — the
//                 onset march as it is today (24 KFLOP... 48 KFLOP per read)
//   mode 4  D   : 16x16x32, A in registers (7 x 2 x 2 = 112 VGPRs), NT = 2: 4 reads per 12 MFMA
//   mode 5  D1  : 16x16x32, A in registers, NT = 1: 2 reads per 6 MFMA (half the accumulators, more waves possible)
// WAVES per CU: 4 (one per SIMD) and 8 (two per SIMD, what the kernels run).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k(const uint4* in, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) uint4 lds[8192];
  constexpr int NT = 64 * WAVES;
  for (int i = threadIdx.x; i < 8192; i += NT) lds[i] = in[i & 1023];
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  f16x8 areg[28];
#pragma unroll
  for (int i = 0; i < 28; ++i) areg[i] = __builtin_bit_cast(f16x8, in[(i * 37 + lane) & 1023]);
  f32x16 acc[2];
  f32x4 acc4[8];
  for (int x = 0; x < 2; ++x)
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
  for (int x = 0; x < 8; ++x)
    for (int r = 0; r < 4; ++r) acc4[x][r] = 0.f;
  const int base = (w * 640 + lane) & 8191;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const int ob = base + (it & 15) * 32;
    if (MODE == 0) {
#pragma unroll
      for (int s = 0; s < 13; ++s) {
        const int o = (ob + s * 64) & 8191;
        f16x8 ah = __builtin_bit_cast(f16x8, lds[o]);
        f16x8 al = __builtin_bit_cast(f16x8, lds[(o + 2048) & 8191]);
        f16x8 bh = __builtin_bit_cast(f16x8, lds[(o + 4096) & 8191]);
        f16x8 bl = __builtin_bit_cast(f16x8, lds[(o + 6144) & 8191]);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[0], 0, 0, 0);
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int s = 0; s < 13; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[2 * s + 1], areg[(s + 3) % 26], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[2 * s], areg[(s + 3) % 26], acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[2 * s], areg[(s + 7) % 26], acc[0], 0, 0, 0);
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int s = 0; s < 7; ++s) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const f16x8 bh = areg[(4 * s + 9 + t) % 28], bl = areg[(4 * s + 14 + t) % 28];
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const int x = 2 * t + m;
            acc4[4 + x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(areg[4 * s + 2 * m + 1], bh, acc4[4 + x], 0, 0, 0);
            acc4[x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(areg[4 * s + 2 * m], bh, acc4[x], 0, 0, 0);
            acc4[4 + x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(areg[4 * s + 2 * m], bl, acc4[4 + x], 0, 0, 0);
          }
        }
      }
    } else if (MODE == 3) {
#pragma unroll
      for (int s = 0; s < 13; ++s) {
        const int o = (ob + s * 64) & 8191;
        f16x8 bh = __builtin_bit_cast(f16x8, lds[(o + 4096) & 8191]);
        f16x8 bl = __builtin_bit_cast(f16x8, lds[(o + 6144) & 8191]);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[2 * s + 1], bh, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[2 * s], bh, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[2 * s], bl, acc[0], 0, 0, 0);
      }
    } else {
      constexpr int TILES = MODE == 4 ? 2 : 1;
#pragma unroll
      for (int s = 0; s < 7; ++s) {
        const int o = (ob + s * 64) & 8191;
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
          const f16x8 bh = __builtin_bit_cast(f16x8, lds[(o + t * 1024) & 8191]);
          const f16x8 bl = __builtin_bit_cast(f16x8, lds[(o + t * 1024 + 4096) & 8191]);
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const int x = 2 * t + m;
            acc4[4 + x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(areg[4 * s + 2 * m + 1], bh, acc4[4 + x], 0, 0, 0);
            acc4[x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(areg[4 * s + 2 * m], bh, acc4[x], 0, 0, 0);
            acc4[4 + x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(areg[4 * s + 2 * m], bl, acc4[4 + x], 0, 0, 0);
          }
        }
      }
    }
  }
  float sum = 0;
  for (int x = 0; x < 2; ++x) sum += acc[x][0] + acc[x][7];
  for (int x = 0; x < 8; ++x) sum += acc4[x][0] + acc4[x][3];
  out[blockIdx.x * NT + threadIdx.x] = sum;
}

template <int MODE, int WAVES>
void run(const char* name) {
  uint4* in;
  float* out;
  (void)hipMalloc(&in, 1024 * 16);
  (void)hipMalloc(&out, 256 * 64 * WAVES * 4);
  static unsigned short h[1024 * 8];
  srand(1);
  for (int i = 0; i < 1024 * 8; ++i)
    h[i] = (unsigned short)(((rand() & 1) << 15) | ((10 + (rand() & 3)) << 10) | (rand() & 1023));
  (void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  const int iters = 12000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, in, out, 3000);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, in, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_iter = (MODE == 0 || MODE == 1 || MODE == 3) ? 13 * 3 * 32768.0
                          : (MODE == 5)                          ? 7 * 6 * 16384.0
                                                                 : 7 * 12 * 16384.0;
  const double flop = 256.0 * WAVES * iters * per_iter;
  printf("%-72s %8.3f ms  %5.0f TFLOP/s\n", name, ms, flop / ms / 1e9);
  (void)hipFree(in);
  (void)hipFree(out);
}

int main() {
  run<1, 4>("R32: 32x32x16 from registers, 4 waves/CU");
  run<2, 4>("R16: 16x16x32 from registers, 4 waves/CU");
  run<0, 8>("A32: 32x32x16, A+B from LDS, 4 reads per 3 MFMA, 8 waves/CU");
  run<3, 4>("C32: 32x32x16, A in regs, 2 reads per 3 MFMA, 4 waves/CU");
  run<3, 8>("C32: 32x32x16, A in regs, 2 reads per 3 MFMA, 8 waves/CU");
  run<4, 4>("D  : 16x16x32, A in regs, 4 reads per 12 MFMA, 4 waves/CU");
  run<4, 8>("D  : 16x16x32, A in regs, 4 reads per 12 MFMA, 8 waves/CU");
  run<5, 8>("D1 : 16x16x32, A in regs, 2 reads per 6 MFMA, 8 waves/CU");
  run<5, 12>("D1 : 16x16x32, A in regs, 2 reads per 6 MFMA, 12 waves/CU");
  run<1, 4>("R32 again");
  run<2, 8>("R16: 16x16x32 from registers, 8 waves/CU");
  return 0;
}
