// Micro-benchmark (tools/): how fast ONE accumulator chain of [f16 32x32x16, f16 32x32x16, fp8 32x32x64 block-scaled]
// runs on a SIMD with 1 and 2 waves resident, operands from registers — the matrix-side bound of the folded conv1 block
// (conv_contour_fold_mx.hip), against the same instructions on independent accumulators.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using i32x8 = __attribute__((ext_vector_type(8))) int;

template <int CHAINS>
__global__ __launch_bounds__(512) void chain(const uint4* in, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f16x8 a = __builtin_bit_cast(f16x8, in[lane]), b = __builtin_bit_cast(f16x8, in[64 + lane]);
  const uint4 x = in[128 + lane], y = in[192 + lane];
  i32x8 ma = {(int)x.x, (int)x.y, (int)x.z, (int)x.w, (int)y.x, (int)y.y, (int)y.z, (int)y.w};
  i32x8 mb = {(int)y.w, (int)y.z, (int)x.y, (int)x.x, (int)y.y, (int)x.w, (int)x.z, (int)y.x};
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
      acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ma, mb, acc[c], 0, 0, 0, 0x7f, 0, 0x7f);
  }
  float s = 0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][9];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int CHAINS>
void run(int threads) {
  uint4* in;
  float* out;
  (void)hipMalloc(&in, 256 * 16);
  (void)hipMalloc(&out, 256 * 512 * 4);
  unsigned short h[256 * 8];
  for (int i = 0; i < 128 * 8; ++i) h[i] = (unsigned short)(((rand() & 1) << 15) | ((12 + (rand() & 3)) << 10) | (rand() & 1023));
  unsigned char* h8 = reinterpret_cast<unsigned char*>(h + 128 * 8);
  for (int i = 0; i < 128 * 16; ++i) h8[i] = (unsigned char)(((rand() & 1) << 7) | ((5 + (rand() & 3)) << 3) | (rand() & 7));
  (void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  const int iters = 40000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((chain<CHAINS>), dim3(256), dim3(threads), 0, 0, in, out, 8000);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((chain<CHAINS>), dim3(256), dim3(threads), 0, 0, in, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const int waves_per_simd = threads / 256;
  const double ns_group = ms * 1e6 / iters / CHAINS;  // per [f16, f16, fp8] group of one chain of one wave
  printf("%d chain(s) per wave, %d wave(s) per SIMD: %7.1f ns per group per chain; pipe time per group at 2.4 GHz = 53.3 ns "
         "-> pipe busy %.0f %% (nominal clock)\n",
         CHAINS, waves_per_simd, ns_group, 100.0 * 53.33 * waves_per_simd / ns_group);
}

int main() {
  run<1>(256);
  run<1>(512);
  run<2>(256);
  run<2>(512);
  run<4>(256);
  return 0;
}
