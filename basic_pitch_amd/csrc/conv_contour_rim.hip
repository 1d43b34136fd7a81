// Contour conv1, the rim of the harmonic stack as a dense GEMM (default path since round 2).
//
//   basic_pitch/nn.py:69-88 (HarmonicStacking: shift, zero-pad, CROP to 264 bins) + basic_pitch/models.py:241-250
//   (Conv2D 8->8, 3 x 39, "same", folded BN, ReLU) for the output bins f < 20 and f >= 244.
//
// Away from the rim the 8 shifted stack channels fold into ONE 176-tap kernel per output channel
// (contour_conv1_folded_kernel).  At the rim "crop, then zero-pad" removes a different set of taps for every output
// bin, so the folded kernel becomes position dependent: K_f[o][dt][j] = sum over (c, df) with stack bin f + df - 19
// inside [0, 264) and z bin j = f + df - 19 + shift_c.  Round 1 kept the rim on the exact 8-channel form (63 k-steps of
// 2 taps x 8 channels, 10.3 k matrix instructions per window).  But "a different kernel per output bin" is just a
// dense matrix: per rim side
//     C[(f, o) : 160 rows][frame : columns] = K[(f, o)][(dt, j) : 3 x 144] x Z[(dt, j)][frame],
// with Z[(dt, j)][t] = z[t + dt - 1][j0 + j] straight from the normalised CQT (no stack image at all): 27 k-steps per
// 32 x 32 tile, 4.9 k matrix instructions per window — and the B operand is a plain row image of z.
//   * low rim: f in [0, 20), z bins [0, 144) (141 used); high rim: f in [244, 264), z bins [184, 328) (125 used; bins
//     >= 309 are zp's zero padding).
//   * a work item is (window, side, 64 frames); a workgroup is 5 waves, wave m owns the 32 rows (4 bins x 8 channels)
//     of M block m for both 32-frame column tiles; the A fragments (K, split hi | lo, 54 KB per M block) stream from L2,
//     fetched three k-steps ahead; B = 16 bytes of an f16 row image in LDS (hi plane, lo plane; row stride 37 units:
//     conflict-free for the ds_read_b128 lane groups, one frame per lane).
//   * epilogue from the accumulator layout: a lane holds 4 consecutive channels of 4 bins of its frame -> bias, ReLU,
//     four 16-byte stores into c1 (the same [172][268][8] tensor the folded kernel fills the interior of).
// Split-precision products hi*hi + (lo*hi + hi*lo) * 2^-11 with fp32 accumulation as everywhere (bp_common.h).
// Roofline: f16 MFMA issue.  Algorithmic work 103 MFLOP per window (the reference's 8-channel products for 40 of 264
// bins); executed 4860 x 3 MFMAs; bytes per window: 2 x 172 x 576 B of zp read, 220 KB of c1 written, 1.6 MB of A
// fragments from L2.
#include "bp_common.h"

namespace bp {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int kRimWaves = 5;              // M blocks of 32 rows: 20 bins x 8 channels per side
constexpr int kRimBins = 144;             // z bins per frame a side reads (multiple of 16)
constexpr int kRimStepsDt = kRimBins / 16;  // 9 k-steps per frame tap
constexpr int kRimSteps = 3 * kRimStepsDt;  // 27
#ifndef BP_RIM_FRAMES
#define BP_RIM_FRAMES 64
#endif
constexpr int kRimFrames = BP_RIM_FRAMES;  // frames per work item: kRimNT 32-column tiles
constexpr int kRimNT = kRimFrames / 32;
constexpr int kRimTiles = (kFrames + kRimFrames - 1) / kRimFrames;  // 3
constexpr int kRimRowU = 37;              // LDS row: 18 units hi | 18 units lo | 1 pad (odd: conflict-free)
constexpr int kRimRows = kRimFrames + 2;
constexpr int kRimPf = 6;                 // k-steps of A prefetch (L2 latency under load ~ 4 k-steps of 6 MFMAs)
__host__ __device__ constexpr int rim_j0(int side) { return side ? 184 : 0; }
__host__ __device__ constexpr int rim_f0(int side) { return side ? 244 : 0; }

struct RimParams {
  const uint32_t* zp;   // [n][kZRowsP][kZRow]
  const uint4* afrag;   // [side 2][mb 5][step 27][hi|lo][64 lanes] x (8 x f16)   (bp_api.hip pack_contour_rim)
  const float* bias;    // [8]
  float* c1;            // [n][172][kC1Row][8]
  int n_items;          // n_windows * 2 * kRimTiles
};

template <bool WLO>
__global__ __launch_bounds__(64 * kRimWaves) void contour_conv1_rim_kernel(RimParams p) {
  __shared__ __attribute__((aligned(16))) uint4 img[kRimRows * kRimRowU];

  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int kh = lane >> 5, n = lane & 31;

  const int item = blockIdx.x;
  const int b = item / (2 * kRimTiles);
  const int rem = item - b * (2 * kRimTiles);
  const int side = rem / kRimTiles, tile = rem - side * kRimTiles;
  const int t0 = tile * kRimFrames;

  // ---- stage frames t0 - 1 .. t0 + 64 of the side's 144 z bins as f16 planes: unit (row, u) = bins j0 + 8u .. + 7
  const uint32_t* zwin = p.zp + (int64_t)b * kZWin + kZPadL + rim_j0(side);
  for (int e = threadIdx.x; e < kRimRows * (kRimBins / 8); e += 64 * kRimWaves) {
    const int row = e / (kRimBins / 8), u = e - row * (kRimBins / 8);
    const int t = t0 - 1 + row;  // frame of this image row; zp rows -1 and 172 are zero, beyond them nothing exists
    uint4 vh{0u, 0u, 0u, 0u}, vl{0u, 0u, 0u, 0u};
    if (t >= -1 && t <= kFrames) {
      const uint4* src = reinterpret_cast<const uint4*>(zwin + (int64_t)(t + 1) * kZRow + 8 * u);
      const uint4 w0 = src[0], w1 = src[1];
      vh.x = (w0.x & 0xffffu) | (w0.y << 16);
      vh.y = (w0.z & 0xffffu) | (w0.w << 16);
      vh.z = (w1.x & 0xffffu) | (w1.y << 16);
      vh.w = (w1.z & 0xffffu) | (w1.w << 16);
      vl.x = (w0.x >> 16) | (w0.y & 0xffff0000u);
      vl.y = (w0.z >> 16) | (w0.w & 0xffff0000u);
      vl.z = (w1.x >> 16) | (w1.y & 0xffff0000u);
      vl.w = (w1.z >> 16) | (w1.w & 0xffff0000u);
    }
    img[row * kRimRowU + u] = vh;
    img[row * kRimRowU + 18 + u] = vl;
  }

  // ---- A fragments of this wave's M block, fetched kRimPf k-steps ahead (global, L2 resident)
  const uint4* afr = p.afrag + ((int64_t)(side * kRimWaves + wave) * kRimSteps) * 2 * 64 + lane;
  uint4 ah[kRimPf], al[kRimPf];
#pragma unroll
  for (int s = 0; s < kRimPf; ++s) {
    ah[s] = afr[(s * 2 + 0) * 64];
    al[s] = afr[(s * 2 + 1) * 64];
  }
  __syncthreads();

  f32x16 acc[kRimNT], accc[kRimNT];
#pragma unroll
  for (int j = 0; j < kRimNT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = accc[j][r] = 0.0f;

  // lane (n, kh) of column tile j reads image row (32 j + n + dt), unit 2 e + kh of the hi / lo plane
  const int lane_u = n * kRimRowU + kh;
#pragma unroll
  for (int s = 0; s < kRimSteps; ++s) {
    const int dt = s / kRimStepsDt, e = s - dt * kRimStepsDt;
    const uint4 a_hi = ah[s % kRimPf], a_lo = al[s % kRimPf];
    if (s + kRimPf < kRimSteps) {
      ah[s % kRimPf] = afr[((s + kRimPf) * 2 + 0) * 64];
      al[s % kRimPf] = afr[((s + kRimPf) * 2 + 1) * 64];
    }
    const f16x8 fah = __builtin_bit_cast(f16x8, a_hi);
    const f16x8 fal = __builtin_bit_cast(f16x8, a_lo);
    f16x8 bh[kRimNT], bl[kRimNT];
#pragma unroll
    for (int j = 0; j < kRimNT; ++j) {
      const int at = lane_u + (32 * j + dt) * kRimRowU + 2 * e;
      bh[j] = __builtin_bit_cast(f16x8, img[at]);
      bl[j] = __builtin_bit_cast(f16x8, img[at + 18]);
    }
    // the column tiles alternate so that no MFMA waits for the one just issued
#pragma unroll
    for (int j = 0; j < kRimNT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, bh[j], acc[j], 0, 0, 0);
    if (WLO) {
#pragma unroll
      for (int j = 0; j < kRimNT; ++j) accc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal, bh[j], accc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < kRimNT; ++j) accc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, bl[j], accc[j], 0, 0, 0);
  }

  // ---- epilogue: C row i = (r & 3) + 8 (r >> 2) + 4 kh = 8 (bin of the block) + channel -> channels 4 kh .. 4 kh + 3
  // of bin r >> 2; column n = frame
  float bias4[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) bias4[c] = p.bias[4 * kh + c];
#pragma unroll
  for (int j = 0; j < kRimNT; ++j) {
    const int t = t0 + 32 * j + n;
    if (t < kFrames) {
      float* row = p.c1 + (((int64_t)b * kFrames + t) * kC1Row + kC1Pad + rim_f0(side) + 4 * wave) * 8 + 4 * kh;
#pragma unroll
      for (int fb = 0; fb < 4; ++fb) {
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          v[c] = fmaxf(__builtin_fmaf(accc[j][4 * fb + c], kLoUnscale, acc[j][4 * fb + c]) + bias4[c], 0.0f);
        *reinterpret_cast<f32x4*>(row + fb * 8) = v;
      }
    }
  }
}

void launch_contour_conv1_rim(const uint32_t* zp, const void* afrag, const float* bias, float* c1, int n_windows,
                              bool weights_have_lo, hipStream_t stream) {
  RimParams p{zp, static_cast<const uint4*>(afrag), bias, c1, n_windows * 2 * kRimTiles};
  if (p.n_items <= 0) return;
  if (weights_have_lo)
    hipLaunchKernelGGL(contour_conv1_rim_kernel<true>, dim3(p.n_items), dim3(64 * kRimWaves), 0, stream, p);
  else
    hipLaunchKernelGGL(contour_conv1_rim_kernel<false>, dim3(p.n_items), dim3(64 * kRimWaves), 0, stream, p);
}

}  // namespace bp
