// Micro-benchmark (tools/, round 5): is the sustained f16 MFMA rate of this pool's MI355X a CLOCK limit?
//
// Whole chip, two waves per SIMD, back-to-back matrix instructions from registers on four (32x32x16) or
// eight (16x16x32) independent accumulators, ~2 s per configuration as a train of ~40 ms launches.  Each wave brackets its
// loop with s_memtime (the shader-clock counter) and s_memrealtime (100 MHz): the kernel itself reports the clock it ran at
// and the matrix-pipe cycles per instruction, so "TFLOP/s = issue rate x clock" can be read off without trusting a
// sampler.  Run under tools/clock_log.py for the hwmon view (sclk, socket power) of the same seconds.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/mfma_sustain tools/ubench/mfma_sustain.hip
//   python tools/clock_log.py --tag mfma_sustain --interval 0.01 -- tools/ubench/bin/mfma_sustain [seconds per config]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

struct Stamp {
  unsigned long long cyc, rt;
};

// MODE 0: 32x32x16 f16 (16 per iteration on 4 accumulators), 1: 16x16x32 f16 (32 per iteration on 8 accumulators)
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const uint4* in, float* out, Stamp* st, int iters) {
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
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (MODE == 0) {
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[m & 3]), __builtin_bit_cast(f16x8, b[(m >> 2) & 3]), acc[m & 3], 0, 0, 0);
      } else {
        acc4[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[m & 3]), __builtin_bit_cast(f16x8, b[(m >> 2) & 3]), acc4[m & 7], 0, 0, 0);
        acc4[(m + 4) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[(m + 1) & 3]), __builtin_bit_cast(f16x8, b[(m >> 2) & 3]), acc4[(m + 4) & 7], 0, 0, 0);
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = wall_clock64();
  float s = 0;
  for (int x = 0; x < 4; ++x) s += acc[x][0] + acc[x][7];
  for (int x = 0; x < 8; ++x) s += acc4[x][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{c1 - c0, r1 - r0};
}

template <int MODE>
void run(const char* name, int kind, double seconds) {
  uint4* in;
  float* out;
  Stamp* st;
  const int blocks = 512;  // 2 blocks of 4 waves per CU: exactly two waves on every SIMD, all resident at once
  (void)hipMalloc(&in, 512 * 16);
  (void)hipMalloc(&out, blocks * 256 * 4);
  (void)hipMalloc(&st, blocks * 4 * sizeof(Stamp));
  unsigned short h[512 * 8];
  srand(1);
  for (int i = 0; i < 512 * 8; ++i) {
    unsigned short v = (unsigned short)(((rand() & 1) << 15) | ((12 + (rand() & 3)) << 10) | (rand() & 1023));
    if (kind == 0) v = 0;
    h[i] = v;
  }
  (void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  const int iters = 20000;
  const double flop_per_launch = (double)blocks * 4 * iters * (MODE == 0 ? 16 * 32768.0 : 32 * 16384.0);
  const double mfma_per_wave = (double)iters * (MODE == 0 ? 16 : 32);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, in, out, st, 2000);
  (void)hipDeviceSynchronize();
  double total_ms = 0;
  int launches = 0;
  float first_ms = 0, last_ms = 0;
  Stamp* hs = (Stamp*)malloc(blocks * 4 * sizeof(Stamp));
  double mhz_sum = 0, cpm_sum = 0;
  while (total_ms < seconds * 1e3) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, in, out, st, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (!launches) first_ms = ms;
    last_ms = ms;
    total_ms += ms;
    ++launches;
    (void)hipMemcpy(hs, st, blocks * 4 * sizeof(Stamp), hipMemcpyDeviceToHost);
    double mhz = 0, cpm = 0;
    for (int i = 0; i < blocks * 4; ++i) {
      mhz += (double)hs[i].cyc / ((double)hs[i].rt / 100.0);  // cycles per microsecond
      cpm += (double)hs[i].cyc / mfma_per_wave;
    }
    mhz_sum += mhz / (blocks * 4), cpm_sum += cpm / (blocks * 4);
  }
  // two waves per SIMD share its matrix pipe: pipe cycles per instruction = ticks per instruction and wave / 2
  printf("%-22s %-7s %3d launches: first %.2f ms, last %.2f ms, mean %.0f TFLOP/s | in-kernel: s_memtime %.0f ticks/us, %.1f ticks per instruction and wave\n",
         name, kind ? "random" : "zeros", launches, first_ms, last_ms, flop_per_launch * launches / total_ms / 1e9, mhz_sum / launches,
         cpm_sum / launches);
  fflush(stdout);
  free(hs);
  (void)hipFree(in), (void)hipFree(out), (void)hipFree(st);
}

int main(int argc, char** argv) {
  const double sec = argc > 1 ? atof(argv[1]) : 2.0;
  run<0>("32x32x16 f16", 1, sec);
  run<0>("32x32x16 f16", 0, sec);
  run<1>("16x16x32 f16", 1, sec);
  run<1>("16x16x32 f16", 0, sec);
  run<0>("32x32x16 f16 (again)", 1, sec);
  return 0;
}
