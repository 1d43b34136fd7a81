// LDS-DMA semantics check (gfx950): where does lane l's 16 bytes of global_load_lds_dwordx4 land, and is the data
// there after s_waitcnt vmcnt(0) + barrier?  Prints the number of mismatches against "LDS base + 16 * lane".
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void k(const float* __restrict__ src, float* __restrict__ out, int n_units) {
  __shared__ __attribute__((aligned(16))) float raw[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) raw[i] = -1.0f;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // unit u (16 bytes) <- src unit perm(u); each wave moves 64 units
  const int u = wave * 64 + lane;
  if (u < n_units) {
    const int su = (u * 37) % n_units;
    __builtin_amdgcn_global_load_lds(src + 4 * su, raw + 256 * wave, 16, 0, 0);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __syncthreads();
  float* rp = raw;
  asm volatile("" : "+v"(rp)::"memory");
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) out[i] = rp[i];
}

int main() {
  const int n_units = 200;  // 3 full waves + a partial one
  std::vector<float> h(4 * n_units);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
  float *d, *o;
  hipMalloc(&d, h.size() * 4);
  hipMalloc(&o, 4096 * 4);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, o, n_units);
  std::vector<float> r(4096);
  hipMemcpy(r.data(), o, 4096 * 4, hipMemcpyDeviceToHost);
  int bad = 0, untouched_bad = 0;
  for (int u = 0; u < 1024; ++u) {
    for (int e = 0; e < 4; ++e) {
      const float got = r[4 * u + e];
      if (u < n_units) {
        const float want = (float)(4 * ((u * 37) % n_units) + e);
        if (got != want) { if (bad < 5) printf("unit %d elt %d: got %g want %g\n", u, e, got, want); ++bad; }
      } else if (got != -1.0f) ++untouched_bad;
    }
  }
  printf("lds_dma dwordx4: %d mismatches, %d stray writes (expect 0, 0)\n", bad, untouched_bad);
  return 0;
}
