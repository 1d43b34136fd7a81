// Micro-benchmark (tools/): dense f16 MFMA rate with trivial vs random operand data (wall clock, whole chip).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

// MODE 0: 32x32x16 f16, 1: 16x16x32 f16, 2: 32x32x16 bf16
template <int WAVES, int MODE>
__global__ __launch_bounds__(64 * WAVES, 1) void k2(const uint4* in, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  uint4 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = in[(i * 64 + lane)];
    b[i] = in[(256 + i * 64 + lane)];
  }
  f32x16 acc[4];
  f32x4 acc4[8];
  for (int x = 0; x < 4; ++x)
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
  for (int x = 0; x < 8; ++x)
    for (int r = 0; r < 4; ++r) acc4[x][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (MODE == 0)
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[m & 3]), __builtin_bit_cast(f16x8, b[(m >> 2) & 3]), acc[m & 3], 0, 0, 0);
      else if (MODE == 2)
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[m & 3]), __builtin_bit_cast(bf16x8, b[(m >> 2) & 3]), acc[m & 3], 0, 0, 0);
      else {
        acc4[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[m & 3]), __builtin_bit_cast(f16x8, b[(m >> 2) & 3]), acc4[m & 7], 0, 0, 0);
        acc4[(m + 4) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[(m + 1) & 3]), __builtin_bit_cast(f16x8, b[(m >> 2) & 3]), acc4[(m + 4) & 7], 0, 0, 0);
      }
    }
  }
  float s = 0;
  for (int x = 0; x < 4; ++x) s += acc[x][0] + acc[x][7];
  for (int x = 0; x < 8; ++x) s += acc4[x][0];
  out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
}

template <int WAVES, int MODE>
void run2(const char* name, int kind) {
  uint4* in;
  float* out;
  (void)hipMalloc(&in, 512 * 16);
  (void)hipMalloc(&out, 1024 * 64 * WAVES * 4);
  unsigned short h[512 * 8];
  for (int i = 0; i < 512 * 8; ++i) {
    unsigned short v = (unsigned short)(((rand() & 1) << 15) | ((12 + (rand() & 3)) << 10) | (rand() & 1023));
    if (kind == 0) v = 0;
    if (kind == 2 && (i / 8) % 2) v = 0;          // every other lane's operand zero
    if (kind == 3) v = (unsigned short)(v & 0xFC00); // random sign/exponent, zero mantissa
    if (kind == 4 && i >= 256 * 8) v = (unsigned short)((v & 0x83FF) | (2 << 10));  // B tiny (like scaled residuals ~ 2^-13)
    h[i] = v;
  }
  (void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k2<WAVES, MODE>), dim3(1024), dim3(64 * WAVES), 0, 0, in, out, 4000);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k2<WAVES, MODE>), dim3(1024), dim3(64 * WAVES), 0, 0, in, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  double flop = 1024.0 * WAVES * iters * 16 * 32768.0;
  printf("%-44s %.3f ms  %.0f TFLOP/s\n", name, ms, flop / ms / 1e9);
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k(const uint4* in, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = __builtin_bit_cast(f16x8, in[(i * 64 + lane)]);
    b[i] = __builtin_bit_cast(f16x8, in[(256 + i * 64 + lane)]);
  }
  f32x16 acc[4];
  for (int x = 0; x < 4; ++x)
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m & 3], b[(m >> 2) & 3], acc[m & 3], 0, 0, 0);
  }
  float s = 0;
  for (int x = 0; x < 4; ++x) s += acc[x][0] + acc[x][7];
  out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
}

template <int WAVES>
void run(const char* name, bool random) {
  uint4* in;
  float* out;
  (void)hipMalloc(&in, 512 * 16);
  (void)hipMalloc(&out, 1024 * 64 * WAVES * 4);
  unsigned short h[512 * 8];
  for (int i = 0; i < 512 * 8; ++i) {
    // f16 in [-2, 2): sign | exponent 12..15 | random mantissa
    h[i] = random ? (unsigned short)(((rand() & 1) << 15) | ((12 + (rand() & 3)) << 10) | (rand() & 1023)) : 0;
  }
  (void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<WAVES>), dim3(1024), dim3(64 * WAVES), 0, 0, in, out, 2000);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<WAVES>), dim3(1024), dim3(64 * WAVES), 0, 0, in, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  double flop = 1024.0 * WAVES * iters * 16 * 32768.0;
  printf("%-28s %d waves/block: %.3f ms  %.0f TFLOP/s\n", name, WAVES, ms, flop / ms / 1e9);
}

int main2() {
  run2<4, 0>("32x32x16 f16 zeros", 0);
  run2<4, 0>("32x32x16 f16 random", 1);
  run2<4, 0>("32x32x16 f16 random, half the lanes zero", 2);
  run2<4, 0>("32x32x16 f16 random exponent, zero mantissa", 3);
  run2<4, 0>("32x32x16 f16 random A, small B", 4);
  run2<4, 1>("16x16x32 f16 random", 1);
  run2<4, 2>("32x32x16 bf16 random", 1);
  run2<4, 0>("32x32x16 f16 random (again)", 1);
  return 0;
}
int main_old() {
  run<4>("zeros", false);
  run<4>("random f16", true);
  run<8>("zeros", false);
  run<8>("random f16", true);
  return 0;
}

int main() { return main2(); }
