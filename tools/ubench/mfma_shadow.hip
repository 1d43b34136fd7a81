// Micro-benchmark (tools/): do VALU instructions hide in the shadow of v_mfma_f32_16x16x32_f16 / 32x32x16 on gfx950?
// One wave-loop = 7 k-steps x 6 MFMAs (the onset conv1 tile of onset_march16.hip: weights in registers, B hi / B lo of one
// 16-pixel tile from LDS, two accumulator chains per weight block) with K independent VALU instructions placed in program
// order right behind every matrix instruction.  Whole chip, W waves per CU, wall clock; reported: ns per matrix
// instruction per SIMD-slot and the implied cycles at the measured rate.  K = 0 is the matrix-only floor; if the VALU
// hide, the time stays flat until K reaches the free issue slots of one instruction (3 for 16 cycles, 7 for 32).
//   kind 0: v_fma_f32   1: v_pk_fma_f32   2: v_cvt_pkrtz_f16_f32   3: v_mov_b32 dpp row_shr:1   4: v_max_f32
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

template <int KIND>
__device__ __forceinline__ void valu(float (&x)[8], f32x2 (&y)[4], int i) {
  if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i & 7]) : "v"(x[(i + 3) & 7]));
  if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(y[i & 3]) : "v"(y[(i + 1) & 3]));
  if (KIND == 2) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(x[i & 7]) : "v"(x[(i + 3) & 7]));
  if (KIND == 3) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i & 7]) : "v"(x[(i + 3) & 7]));
  if (KIND == 4) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i & 7]) : "v"(x[(i + 3) & 7]));
}

template <int WAVES, int K, int KIND, bool BIG>
__global__ __launch_bounds__(64 * WAVES, WAVES == 4 ? 2 : 1) void k(const uint4* in, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) uint4 lds[4096];  // 256 registers per wave in either shape: no AGPR shuffling
  constexpr int NT = 64 * WAVES;
  for (int i = threadIdx.x; i < 4096; i += NT) lds[i] = in[i & 1023];
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  f16x8 areg[28];
#pragma unroll
  for (int i = 0; i < 28; ++i) areg[i] = __builtin_bit_cast(f16x8, in[(i * 37 + lane) & 1023]);
  f32x4 acc4[4];
  f32x16 acc16[2];
  for (int x = 0; x < 4; ++x)
    for (int r = 0; r < 4; ++r) acc4[x][r] = 0.f;
  for (int x = 0; x < 2; ++x)
    for (int r = 0; r < 16; ++r) acc16[x][r] = 0.f;
  float x8[8];
  f32x2 y4[4];
  for (int i = 0; i < 8; ++i) x8[i] = 0.001f * (lane + i);
  for (int i = 0; i < 4; ++i) y4[i] = f32x2{0.001f * lane, 0.002f * i};
  const int base = (w * 320 + lane) & 4095;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const int ob = base + (it & 15) * 32;
    f16x8 bhn = __builtin_bit_cast(f16x8, lds[ob & 4095]), bln = __builtin_bit_cast(f16x8, lds[(ob + 2048) & 4095]);
#pragma unroll
    for (int s = 0; s < 7; ++s) {
      const f16x8 bh = bhn, bl = bln;  // fragments are read one k-step ahead
      if (s < 6) {
        const int o = (ob + (s + 1) * 64) & 4095;
        bhn = __builtin_bit_cast(f16x8, lds[o]);
        bln = __builtin_bit_cast(f16x8, lds[(o + 2048) & 4095]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!BIG) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int m = j & 1, c = j >> 1;  // c: 0 = lo.hi, 1 = hi.hi, 2 = hi.lo
          const int ai = 4 * s + 2 * m + (c == 0 ? 1 : 0);
          const int x = c == 1 ? m : 2 + m;
          acc4[x] = __builtin_amdgcn_mfma_f32_16x16x32_f16(areg[ai], c == 2 ? bl : bh, acc4[x], 0, 0, 0);
#pragma unroll
          for (int v = 0; v < K; ++v) valu<KIND>(x8, y4, j * K + v);
          __builtin_amdgcn_sched_barrier(0);  // pin the program order: the compiler otherwise sinks the matrix instructions
        }
      } else {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          acc16[j == 1 ? 0 : 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[4 * s + (j == 0)], j == 2 ? bl : bh, acc16[j == 1 ? 0 : 1], 0, 0, 0);
#pragma unroll
          for (int v = 0; v < K; ++v) valu<KIND>(x8, y4, j * K + v);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  float sum = 0;
  for (int x = 0; x < 4; ++x) sum += acc4[x][0] + acc4[x][3];
  for (int x = 0; x < 2; ++x) sum += acc16[x][0] + acc16[x][9];
  for (int i = 0; i < 8; ++i) sum += x8[i];
  for (int i = 0; i < 4; ++i) sum += y4[i].x + y4[i].y;
  out[blockIdx.x * NT + threadIdx.x] = sum;
}

template <int WAVES, int K, int KIND, bool BIG>
void run(const char* kind) {
  uint4* in;
  float* out;
  (void)hipMalloc(&in, 1024 * 16);
  (void)hipMalloc(&out, 256 * 64 * WAVES * 4);
  static unsigned short h[1024 * 8];
  srand(1);
  for (int i = 0; i < 1024 * 8; ++i)
    h[i] = (unsigned short)(((rand() & 1) << 15) | ((10 + (rand() & 3)) << 10) | (rand() & 1023));
  (void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  const int iters = 6000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<WAVES, K, KIND, BIG>), dim3(256), dim3(64 * WAVES), 0, 0, in, out, 1500);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<WAVES, K, KIND, BIG>), dim3(256), dim3(64 * WAVES), 0, 0, in, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double n_mfma = (double)iters * 7 * (BIG ? 3 : 6) * (WAVES / 4.0);  // per SIMD
  const double flop = 256.0 * WAVES * iters * 7 * 6 * 16384.0;
  printf("%-10s %s waves/CU %2d  VALU per MFMA %d : %8.3f ms  %6.1f ns per MFMA per SIMD  %5.0f TFLOP/s\n", kind,
         BIG ? "32x32x16" : "16x16x32", WAVES, K, ms, ms * 1e6 / n_mfma, flop / ms / 1e9);
  (void)hipFree(in);
  (void)hipFree(out);
}

template <int KIND>
void sweep(const char* kind) {
  run<4, 0, KIND, false>(kind);
  run<4, 1, KIND, false>(kind);
  run<4, 2, KIND, false>(kind);
  run<4, 3, KIND, false>(kind);
  run<4, 4, KIND, false>(kind);
  run<4, 6, KIND, false>(kind);
  run<8, 0, KIND, false>(kind);
  run<8, 2, KIND, false>(kind);
  run<8, 3, KIND, false>(kind);
  run<8, 4, KIND, false>(kind);
  run<8, 6, KIND, false>(kind);
  run<4, 0, KIND, true>(kind);
  run<4, 4, KIND, true>(kind);
  run<4, 6, KIND, true>(kind);
  run<4, 8, KIND, true>(kind);
  run<8, 0, KIND, true>(kind);
  run<8, 6, KIND, true>(kind);
  run<8, 8, KIND, true>(kind);
}

int main() {
  sweep<0>("v_fma");
  sweep<1>("v_pk_fma");
  sweep<2>("v_cvt_pk");
  sweep<3>("dpp_mov");
  return 0;
}
