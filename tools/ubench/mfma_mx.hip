// Micro-benchmark (tools/): the block-scaled fp8 matrix instruction v_mfma_scale_f32_32x32x64_f8f6f4 next to the f16
// one, whole chip, wall clock, random operands — (1) its sustained rate, (2) the rate of the instruction mix a
// "f16 main product + fp8 correction products" split-precision kernel would issue per 64 taps (4 f16 + 2 MX) against
// today's 12 f16, (3) a layout / scale check: C = A x B for a known small problem.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using i32x8 = __attribute__((ext_vector_type(8))) int;

// MODE 0: 12 f16 per iteration; 1: 4 f16 + 2 MX; 2: 6 MX (same K as 24 f16); 3: 4 f16 only
template <int MODE>
__global__ __launch_bounds__(256, 1) void rate(const uint4* in, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f16x8 a[4], b[4];
  i32x8 ma[2], mb[2];
  for (int i = 0; i < 4; ++i) {
    a[i] = __builtin_bit_cast(f16x8, in[i * 64 + lane]);
    b[i] = __builtin_bit_cast(f16x8, in[256 + i * 64 + lane]);
  }
  for (int i = 0; i < 2; ++i) {
    const uint4 x = in[512 + i * 128 + lane], y = in[512 + i * 128 + 64 + lane];
    ma[i] = i32x8{(int)x.x, (int)x.y, (int)x.z, (int)x.w, (int)y.x, (int)y.y, (int)y.z, (int)y.w};
    mb[i] = i32x8{(int)y.w, (int)y.z, (int)x.y, (int)x.x, (int)y.y, (int)x.w, (int)x.z, (int)y.x};
  }
  const int sc = 0x7f7f7f7f;  // E8M0 127 = 2^0
  f32x16 acc[4];
  for (int x = 0; x < 4; ++x)
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int m = 0; m < 12; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m & 3], b[(m >> 2) & 3], acc[m & 3], 0, 0, 0);
    } else if (MODE == 1) {
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m & 3], b[m & 3], acc[m & 3], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ma[m], mb[m], acc[m], 0, 0, 0, sc, 0, sc);
    } else if (MODE == 2) {
#pragma unroll
      for (int m = 0; m < 6; ++m) acc[m & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ma[m & 1], mb[(m >> 1) & 1], acc[m & 3], 0, 0, 0, sc, 0, sc);
    } else {
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m & 3], b[m & 3], acc[m & 3], 0, 0, 0);
    }
  }
  float s = 0;
  for (int x = 0; x < 4; ++x) s += acc[x][0] + acc[x][7];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, double taps_per_iter) {
  uint4* in;
  float* out;
  (void)hipMalloc(&in, 1024 * 16);
  (void)hipMalloc(&out, 1024 * 256 * 4);
  unsigned short h[1024 * 8];
  for (int i = 0; i < 512 * 8; ++i) h[i] = (unsigned short)(((rand() & 1) << 15) | ((12 + (rand() & 3)) << 10) | (rand() & 1023));
  unsigned char* h8 = reinterpret_cast<unsigned char*>(h + 512 * 8);
  for (int i = 0; i < 512 * 16; ++i) h8[i] = (unsigned char)(((rand() & 1) << 7) | ((5 + (rand() & 3)) << 3) | (rand() & 7));  // e4m3 ~ [0.25, 4)
  (void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  const int iters = 30000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((rate<MODE>), dim3(1024), dim3(256), 0, 0, in, out, 6000);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((rate<MODE>), dim3(1024), dim3(256), 0, 0, in, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  // cycles per iteration per SIMD at a nominal 2.4 GHz: 1024 blocks x 4 waves over 1024 SIMDs -> 4 waves per SIMD
  printf("%-52s %8.3f ms  %7.1f ns per iteration per wave-slot  (%.0f taps-equivalent per iteration)\n", name, ms,
         ms * 1e6 / iters / 4.0, taps_per_iter);
}

// layout / scale check: A[m][k], B[k][n] with small integers exactly representable in e4m3.  Result on gfx950 (ROCm
// 7.2): byte j of lane (i = lane & 31, kh = lane >> 5) is K = 32 (j >> 4) + 16 kh + (j & 15), and the E8M0 scale a lane
// supplies applies to K block kh (K = 32 kh .. 32 kh + 31) — i.e. to the FIRST 16 bytes of both lane halves for kh = 0
// and to the SECOND 16 bytes for kh = 1, not to the lane's own 32 bytes.  The reference below encodes that.
__global__ void layout(float* out) {
  const int lane = threadIdx.x, i = lane & 31, kh = lane >> 5;
  auto enc = [](int v) -> unsigned {  // e4m3 encoding of v in {0, 1, 2, 3, 4}: 0 -> 0x00, 1 -> 0x38, 2 -> 0x40, 3 -> 0x44, 4 -> 0x48
    const unsigned t[5] = {0x00, 0x38, 0x40, 0x44, 0x48};
    return t[v];
  };
  i32x8 a, b;
  for (int d = 0; d < 8; ++d) {
    unsigned wa = 0, wb = 0;
    for (int j = 0; j < 4; ++j) {
      const int jb = 4 * d + j, k = 32 * (jb >> 4) + 16 * kh + (jb & 15);
      wa |= enc((i + k) % 5) << (8 * j);       // A[i][k] = (i + k) mod 5
      wb |= enc((2 * i + 3 * k) % 5) << (8 * j);  // B[k][i] = (2 i + 3 k) mod 5
    }
    a[d] = (int)wa;
    b[d] = (int)wb;
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  // scales: A blocks x 2^1 (E8M0 128) for kh = 0 and x 2^-2 (125) for kh = 1; B all 2^0
  const int sa = kh ? 125 : 128, sb = kh ? 129 : 126;  // B: x 2^-1 for K block 0, x 2^2 for K block 1
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
  for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
}

int main() {
  run<0>("12 x f16 32x32x16 (today: 3 products, 64 taps)", 64);
  run<1>("4 x f16 + 2 x MX fp8 32x32x64 (f16 main + fp8 corrections)", 64);
  run<3>("4 x f16 32x32x16 (one product, 64 taps)", 64);
  run<2>("6 x MX fp8 32x32x64", 384);
  run<0>("12 x f16 32x32x16 (again)", 64);
  float* out;
  (void)hipMalloc(&out, 64 * 16 * 4);
  hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, out);
  float h[64 * 16];
  (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int r = 0; r < 16; ++r) {
      const int n = lane & 31, m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      double ref = 0;
      for (int k = 0; k < 64; ++k) ref += (double)((m + k) % 5) * ((2 * n + 3 * k) % 5) * (k < 32 ? 2.0 * 0.5 : 0.25 * 4.0);
      if (fabs(ref - h[lane * 16 + r]) > 1e-3) {
        if (bad < 5) printf("mismatch C[%d][%d]: got %g want %g\n", m, n, h[lane * 16 + r], ref);
        ++bad;
      }
    }
  printf("layout / scale check: %d mismatches of 1024\n", bad);
  return 0;
}
