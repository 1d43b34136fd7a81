// Micro-benchmark (tools/): peak LDS read rate per CU for ds_read_b32 / b64 / b128 with contiguous lane addresses,
// 16 reads in flight per wave, W waves per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <typename T, int W>
__global__ __launch_bounds__(64 * W, 1) void k(float* out, int iters, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[128 * 1024];
  for (int i = threadIdx.x; i < 32 * 1024; i += 64 * W) reinterpret_cast<unsigned*>(lds)[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int boff = (w & 3) * (int)(16384 / sizeof(T)) + lane;
  unsigned acc = 0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    T v[16];
    const int off = (it & 3) * 16;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int idx = boff + r * 64 + off;
      asm volatile("" : "+v"(idx));  // distinct, opaque addresses: no CSE, no narrowing
      v[r] = reinterpret_cast<const T*>(lds)[idx];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
      for (int e = 0; e < (int)(sizeof(T) / 4); ++e) acc ^= reinterpret_cast<const unsigned*>(&v[r])[e];
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 * W + threadIdx.x] = (float)acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <typename T, int W>
void run(const char* name) {
  float* out;
  unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 64 * W * 4);
  (void)hipMalloc(&cyc, 8);
  const int iters = 4000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<T, W>), dim3(256), dim3(64 * W), 0, 0, out, iters, cyc);
  unsigned long long h = 0;
  (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  double per_read = (double)h / iters / 16.0 / W;  // CU-level cycles per wave-instruction
  printf("%-10s %d waves/CU: %.2f cycles per wave-instruction, %.0f B/clk/CU\n", name, W, per_read,
         64.0 * sizeof(T) / per_read);
  (void)hipFree(out);
  (void)hipFree(cyc);
}

int main() {
  run<unsigned, 4>("b32");
  run<unsigned, 8>("b32");
  run<uint2, 4>("b64");
  run<uint2, 8>("b64");
  run<uint4, 4>("b128");
  run<uint4, 8>("b128");
  run<uint4, 16>("b128");
  return 0;
}
