// The two frequency-stride-3 convolutions that reduce 264 contour bins to 88 note bins:
//   onset branch  Conv2D 8->32, 5x5, strides (1,3), "same", folded BN, ReLU on the harmonic stack
//   note  branch  Conv2D 1->32, 7x7, strides (1,3), "same", ReLU on the sigmoid contour map
//
// Reference behaviour (spotify/basic-pitch v0.4.0):
//   basic_pitch/models.py:295-304  onset: Conv2D(32,(5,5),padding="same",strides=(1,3)) + BN + ReLU
//                                  (TF "same" with stride 3: pads time 2/2, freq 1/1 — ONNX pads [2,1,2,1])
//   basic_pitch/models.py:266-278  note:  Conv2D(32,(7,7),padding="same",strides=(1,3)) + ReLU
//                                  (pads time 3/3, freq 2/2 — ONNX pads [3,2,3,2])
//   basic_pitch/nn.py:69-88, signal.py:177-183, models.py:187-189 for the onset input (as contour1)
//
// MI355X mapping: implicit GEMM, exact-f32 MFMA 32x32x2 with N = 32 output channels, M = 32
// consecutive (frame, note-bin) pixels, K = 200 (onset) / 49 (+1 zero, note).  All B fragments stay
// in VGPRs (100 / 25 per lane); A is one ds_read_b32 per MFMA from the LDS copy of the input slab —
// lanes step 3 floats, row stride = 8 (mod 32), so reads are conflict free.  Each wave owns whole
// tiles (no K split); the 32x32 result is transposed through a private LDS scratch so the planar
// [channel][frame][bin] stores are 128-byte segments.
//
// Roofline: bound = f32 MFMA.  Algorithmic work 193.7 MFLOP (onset) + 47.5 MFLOP (note) per window;
// bytes: 212,592 (lp) resp. 181,632 (contour) read, 1,937,408 written each.
#include "bp_common.h"

namespace bp {

constexpr int kS3Threads = 256;
constexpr int kS3Slab = 16;
constexpr int kS3Slabs = (kFrames + kS3Slab - 1) / kS3Slab;  // 11
constexpr int kScrRow = 33;
constexpr int kScrTile = 32 * kScrRow;

// acc (+bias, ReLU) -> planar out[(b*32 + ch) * kPlaneN + pix0 + 0..31]
__device__ __forceinline__ void s3_store_tile(const f32x16& acc, float bias_n, float* __restrict__ scr,
                                              float* __restrict__ out_b, int pix0, int n_valid, int lane) {
  const int li = lane & 31, kodd = lane >> 5;
  // C layout 32x32: col n = lane & 31 (channel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (pixel)
#pragma unroll
  for (int r = 0; r < 16; ++r)
    scr[li * kScrRow + (r & 3) + 8 * (r >> 2) + 4 * kodd] = fmaxf(acc[r] + bias_n, 0.0f);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int ch = 2 * it + kodd;
    const float v = scr[ch * kScrRow + li];
    if (li < n_valid) out_b[(int64_t)ch * kPlaneN + pix0 + li] = v;
  }
  __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------
// onset conv1
constexpr int kO1Rows = kS3Slab + 4;
constexpr int kO1Ts = 424;   // >= 403, = 8 (mod 32)
constexpr int kO1Goff = 37;  // zl index of bin 0 (lowest bin read: 3*0 - 1 - 36 = -37)
constexpr int kO1Steps = 100;

__global__ __launch_bounds__(kS3Threads, 2) void onset1_kernel(
    const float* __restrict__ lp, const int* __restrict__ mm, const float* __restrict__ bfrag,
    const float* __restrict__ bias, float* __restrict__ o1, int n_windows, LogConsts kc) {
  __shared__ float zl[kO1Rows * kO1Ts];
  __shared__ float scr_all[4 * kScrTile];
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int kodd = lane >> 5, li = lane & 31;
  float* scr = scr_all + wave * kScrTile;

  float breg[kO1Steps];
#pragma unroll
  for (int j = 0; j < kO1Steps; ++j) breg[j] = bfrag[j * 64 + lane];
  const float bias_n = bias[li];

  const int n_items = n_windows * kS3Slabs;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int b = item / kS3Slabs;
    const int t0 = (item - b * kS3Slabs) * kS3Slab;
    const int rows = (kFrames - t0) < kS3Slab ? (kFrames - t0) : kS3Slab;
    const int n_pix = rows * kFreqN;
    const int n_tiles = (n_pix + 31) >> 5;

    __syncthreads();
    for (int i = threadIdx.x; i < kO1Rows * kO1Ts; i += kS3Threads) zl[i] = 0.0f;
    __syncthreads();
    {
      const float mn = ord2f(mm[2 * b]);
      const float range = ord2f(mm[2 * b + 1]) - mn;
      const float* lpb = lp + (int64_t)b * kFrames * kBins;
      for (int r = 0; r < kO1Rows; ++r) {
        const int t = t0 - 2 + r;
        if (t < 0 || t >= kFrames) continue;
        for (int g = threadIdx.x; g < kBins; g += kS3Threads)
          zl[r * kO1Ts + g + kO1Goff] = norm_bn(lpb[t * kBins + g], mn, range, kc);
      }
    }
    __syncthreads();

    float* out_b = o1 + (int64_t)b * 32 * kPlaneN + (int64_t)t0 * kFreqN;
    for (int tile = wave; tile < n_tiles; tile += 4) {
      int m = tile * 32 + li;
      m = m < n_pix ? m : n_pix - 1;
      const int tr = m / kFreqN;
      const int w = m - tr * kFreqN;
      const float* base = zl + tr * kO1Ts + 3 * w + kO1Goff - 1;
      // "same" padding acts on the cropped 264-bin stack (nn.py:87): stack bins -1 (w = 0, dw = 0) and
      // 264 (w = 87, dw = 4) are zero even where the shifted CQT row has data there.
      const bool edge_lo = (w == 0), edge_hi = (w == kFreqN - 1);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
      for (int cp = 0; cp < 4; ++cp) {
        // lanes 0-31 read channel 2cp, lanes 32-63 channel 2cp+1 (bin shift difference in the base)
        const float* bc = base + harm_shift(2 * cp) + kodd * (harm_shift(2 * cp + 1) - harm_shift(2 * cp));
#pragma unroll
        for (int dt = 0; dt < 5; ++dt) {
#pragma unroll
          for (int dw = 0; dw < 5; ++dw) {
            float a = bc[dt * kO1Ts + dw];
            if (dw == 0) a = edge_lo ? 0.0f : a;
            if (dw == 4) a = edge_hi ? 0.0f : a;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, breg[(cp * 5 + dt) * 5 + dw], acc, 0, 0, 0);
          }
        }
      }
      const int rem = n_pix - tile * 32;
      s3_store_tile(acc, bias_n, scr, out_b, tile * 32, rem < 32 ? rem : 32, lane);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// note conv1
constexpr int kN1Rows = kS3Slab + 6;
constexpr int kN1Ts = 296;   // >= 264 + 4 (+1 pad tap), = 8 (mod 32)
constexpr int kN1Goff = 2;
constexpr int kN1Steps = 25;

__global__ __launch_bounds__(kS3Threads, 2) void note1_kernel(const float* __restrict__ contour,
                                                              const float* __restrict__ bfrag,
                                                              const float* __restrict__ bias,
                                                              float* __restrict__ n1, int n_windows) {
  __shared__ float zl[kN1Rows * kN1Ts];
  __shared__ float scr_all[4 * kScrTile];
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int kodd = lane >> 5, li = lane & 31;
  float* scr = scr_all + wave * kScrTile;

  float breg[kN1Steps];
#pragma unroll
  for (int j = 0; j < kN1Steps; ++j) breg[j] = bfrag[j * 64 + lane];
  const float bias_n = bias[li];

  const int n_items = n_windows * kS3Slabs;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int b = item / kS3Slabs;
    const int t0 = (item - b * kS3Slabs) * kS3Slab;
    const int rows = (kFrames - t0) < kS3Slab ? (kFrames - t0) : kS3Slab;
    const int n_pix = rows * kFreqN;
    const int n_tiles = (n_pix + 31) >> 5;

    __syncthreads();
    for (int i = threadIdx.x; i < kN1Rows * kN1Ts; i += kS3Threads) zl[i] = 0.0f;
    __syncthreads();
    {
      const float* cb = contour + (int64_t)b * kPlaneC;
      for (int r = 0; r < kN1Rows; ++r) {
        const int t = t0 - 3 + r;
        if (t < 0 || t >= kFrames) continue;
        for (int g = threadIdx.x; g < kFreqC; g += kS3Threads)
          zl[r * kN1Ts + g + kN1Goff] = cb[t * kFreqC + g];
      }
    }
    __syncthreads();

    float* out_b = n1 + (int64_t)b * 32 * kPlaneN + (int64_t)t0 * kFreqN;
    for (int tile = wave; tile < n_tiles; tile += 4) {
      int m = tile * 32 + li;
      m = m < n_pix ? m : n_pix - 1;
      const int tr = m / kFreqN;
      const int w = m - tr * kFreqN;
      // tap k = dt*7 + dw reads bin 3w + dw - 2, i.e. zl column 3w + dw; lanes 32-63 read tap k+1
      const float* base = zl + tr * kN1Ts + 3 * w;
      const float* baseLo = base + kodd;                 // next tap in the same row
      const float* baseHi = base + kodd * (kN1Ts - 6);   // (dt, 6) -> (dt+1, 0)
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
      for (int s = 0; s < kN1Steps; ++s) {
        const int k0 = 2 * s;
        const int dt = k0 / 7, dw = k0 - 7 * dt;
        const bool wrap = (dw == 6) && (k0 + 1 < 49);  // the 50th tap is a zero-weight pad: stay in-row
        const float a = wrap ? baseHi[dt * kN1Ts + dw] : baseLo[dt * kN1Ts + dw];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, breg[s], acc, 0, 0, 0);
      }
      const int rem = n_pix - tile * 32;
      s3_store_tile(acc, bias_n, scr, out_b, tile * 32, rem < 32 ? rem : 32, lane);
    }
  }
}

void launch_onset1(const float* lp, const int* mm, const float* bfrag, const float* bias, float* o1,
                   int n_windows, LogConsts kc, int n_cu, hipStream_t stream) {
  const int items = n_windows * kS3Slabs;
  const int grid = items < 2 * n_cu ? items : 2 * n_cu;
  hipLaunchKernelGGL(onset1_kernel, dim3(grid), dim3(kS3Threads), 0, stream, lp, mm, bfrag, bias, o1,
                     n_windows, kc);
}

void launch_note1(const float* contour, const float* bfrag, const float* bias, float* n1,
                  int n_windows, int n_cu, hipStream_t stream) {
  const int items = n_windows * kS3Slabs;
  const int grid = items < 2 * n_cu ? items : 2 * n_cu;
  hipLaunchKernelGGL(note1_kernel, dim3(grid), dim3(kS3Threads), 0, stream, contour, bfrag, bias, n1,
                     n_windows);
}

}  // namespace bp
