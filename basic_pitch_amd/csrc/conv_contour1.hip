// Contour branch, first convolution: NormalizedLog tail + BatchNorm + harmonic stacking (virtual) +
// Conv2D 8->8, kernel (3 time x 39 freq), "same", folded BN, ReLU.  65 % of the path's FLOPs.
//
// Reference behaviour (spotify/basic-pitch v0.4.0):
//   basic_pitch/layers/signal.py:177-183   (lp - min) / max(lp - min), divide_no_nan
//   basic_pitch/models.py:187-189          BatchNormalization on the 1-channel CQT (folded affine)
//   basic_pitch/nn.py:69-88                HarmonicStacking: channel c = z shifted by s_c bins, zero outside
//   basic_pitch/models.py:241-250          Conv2D(8, (3, 39), padding="same") + BN + ReLU
//
// MI355X mapping: implicit GEMM on the exact-f32 matrix cores (v_mfma_f32_32x32x2_f32).
//   * N = 32 = 8 output channels x 4 adjacent output bins (a Toeplitz expansion of the 39-tap
//     frequency kernel to 42 taps: 7.7 % extra MACs instead of the 75 % an N = 8 tile would idle).
//   * M = 32 consecutive (frame, 4-bin group) positions of a 16-frame slab (16 x 66 = 33 tiles exactly).
//   * K = 8 ch x 3 x 42 = 1008 is split across the 4 waves of a workgroup by input-channel pair; each
//     wave keeps ITS 126 B fragments (the Toeplitz weights) in VGPRs for the whole kernel, so the
//     only per-MFMA operand traffic is one ds_read_b32 of A.
//   * A comes from ONE normalised copy of z in LDS (the 8-channel stack is never materialised):
//     channel c only changes the column offset.  z is stored as 4 phase planes (bin mod 4) so the
//     stride-4 reads of the 32 lanes are consecutive dwords — bank-conflict free, row stride
//     450 = 2 (mod 32) keeps that true across the frame wrap inside a tile.
//   * Taps that fall outside the cropped 264-bin stack are zeroed per lane (one compare + select
//     per MFMA; the conv's "same" padding is defined on the stack, not on the 309-bin CQT row).
//   * The four K-partials are summed in a fixed order through LDS, + bias, ReLU, 16-byte stores.
//
// Roofline (this kernel): bound = f32 MFMA (157.3 TFLOP/s).  Algorithmic work 8*8*3*39*172*264*2 =
// 680,030,208 FLOP / window (executed on MFMA: 732 MFLOP incl. Toeplitz padding).  Algorithmic
// bytes: 212,592 read (lp) + 1,453,056 written (c1).
#include "bp_common.h"

namespace bp {

constexpr int kC1Threads = 256;
constexpr int kC1Slab = 16;                 // output frames per work item
constexpr int kC1Rows = kC1Slab + 2;        // + 1-frame halo each side
constexpr int kC1Plane = 112;               // floats per phase plane
constexpr int kC1Ts = 4 * kC1Plane + 2;     // 450: row stride, = 2 (mod 32)
constexpr int kC1Qoff = 14;                 // plane index of bin group 0: bin g = 4*(q - 14) + r
constexpr int kC1Groups = kFreqC / 4;       // 66 4-bin groups per frame
constexpr int kC1Slabs = (kFrames + kC1Slab - 1) / kC1Slab;  // 11
constexpr int kC1Steps = 126;               // MFMAs per tile per wave: 2 ch x 3 dt x 21 tap pairs
constexpr int kRedRow = 36;                 // padded row (floats) of a 32x32 partial tile
constexpr int kRedTile = 32 * kRedRow;

// channel pair of each wave (both members of a pair share a code path through their bin phase)
__host__ __device__ constexpr int c1_wave_chan(int wave, int slot) {
  constexpr int t[4][2] = {{0, 1}, {2, 4}, {5, 3}, {6, 7}};
  return t[wave][slot];
}
// s_c = 4*a + rho with rho in {0, 1}
__host__ __device__ constexpr int c1_rho(int c) { return ((harm_shift(c) % 4) + 4) % 4; }
__host__ __device__ constexpr int c1_a(int c) { return (harm_shift(c) - c1_rho(c)) / 4; }
static_assert(c1_rho(0) == 0 && c1_rho(3) == 1 && c1_rho(6) == 1 && c1_rho(7) == 1 && c1_rho(5) == 0, "phases");

// One channel slot: 3 rows x 21 tap pairs.  Tap e = 2*ep + kodd reads bin 4*m_f + e - 19 + s_c.
template <int RHO, int SLOT>
__device__ __forceinline__ void c1_channel(const float* __restrict__ baseLo,
                                           const float* __restrict__ baseHi, int fbase,
                                           const float (&breg)[kC1Steps], f32x16& acc) {
#pragma unroll
  for (int dt = 0; dt < 3; ++dt) {
#pragma unroll
    for (int ep = 0; ep < 21; ++ep) {
      const int d0 = 2 * ep - 19 + RHO;           // kodd = 0 tap, relative bin
      const int r0 = ((d0 % 4) + 4) % 4;
      const int q0 = (d0 - r0) / 4;               // floor(d0 / 4) >= -5
      const int imm = dt * kC1Ts + r0 * kC1Plane + q0 + 5;
      float a = (r0 < 3) ? baseLo[imm] : baseHi[imm];
      // "same" padding acts on the cropped 264-bin stack (nn.py:87): taps outside it are zero even
      // where the shifted CQT row still has data.  fbase = 4*m_f + kodd - 19 -> stack bin of this tap.
      a = ((unsigned)(fbase + 2 * ep) < (unsigned)kFreqC) ? a : 0.0f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, breg[SLOT * 63 + dt * 21 + ep], acc, 0, 0, 0);
    }
  }
}

template <int RHO_A, int RHO_B>
__device__ __forceinline__ void c1_tile(const float* __restrict__ zl, int lane_base, int kodd, int fbase,
                                        int aA, int aB, const float (&breg)[kC1Steps], f32x16& acc) {
  // kodd = 1 lanes read the next tap: +1 phase plane, or (phase 3 -> 0) next group
  const float* bA = zl + lane_base + aA;
  const float* bB = zl + lane_base + aB;
  const int dLo = kodd * kC1Plane, dHi = kodd * (1 - 3 * kC1Plane);
  c1_channel<RHO_A, 0>(bA + dLo, bA + dHi, fbase, breg, acc);
  c1_channel<RHO_B, 1>(bB + dLo, bB + dHi, fbase, breg, acc);
}

__global__ __launch_bounds__(kC1Threads, 2) void contour1_kernel(
    const float* __restrict__ lp, const int* __restrict__ mm, const float* __restrict__ bfrag,
    const float* __restrict__ bias, float* __restrict__ c1, int n_windows, LogConsts kc) {
  __shared__ __attribute__((aligned(16))) float zl[kC1Rows * kC1Ts];
  __shared__ __attribute__((aligned(16))) float red[2 * 4 * kRedTile];

  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int kodd = lane >> 5;
  const int li = lane & 31;

  float breg[kC1Steps];
  {
    const float* bp_ = bfrag + (size_t)wave * kC1Steps * 64 + lane;
#pragma unroll
    for (int j = 0; j < kC1Steps; ++j) breg[j] = bp_[j * 64];
  }
  const int chA = (wave == 0) ? 0 : (wave == 1) ? 2 : (wave == 2) ? 5 : 6;
  const int chB = (wave == 0) ? 1 : (wave == 1) ? 4 : (wave == 2) ? 3 : 7;
  const int shA = (wave == 0) ? -36 : (wave == 1) ? 36 : (wave == 2) ? 84 : 93;
  const int shB = (wave == 0) ? 0 : (wave == 1) ? 72 : (wave == 2) ? 57 : 101;
  (void)chA;
  (void)chB;
  const int aA = (shA - (((shA % 4) + 4) % 4)) / 4;
  const int aB = (shB - (((shB % 4) + 4) % 4)) / 4;
  // epilogue role: this lane finalises output channel oc for tile row li
  const int oc = 2 * wave + kodd;
  const float obias = bias[oc];

  const int n_items = n_windows * kC1Slabs;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int b = item / kC1Slabs;
    const int t0 = (item - b * kC1Slabs) * kC1Slab;
    const int rows = (kFrames - t0) < kC1Slab ? (kFrames - t0) : kC1Slab;
    const int n_pos = rows * kC1Groups;
    const int n_tiles = (n_pos + 31) >> 5;

    // ---- stage z = BN(normalised log-power) for frames t0-1 .. t0+16, zero outside the image
    __syncthreads();  // previous item's readers of zl / red are done
    for (int i = threadIdx.x; i < kC1Rows * kC1Ts; i += kC1Threads) zl[i] = 0.0f;
    __syncthreads();
    {
      const float mn = ord2f(mm[2 * b]);
      const float range = ord2f(mm[2 * b + 1]) - mn;
      const float* lpb = lp + (int64_t)b * kFrames * kBins;
      for (int r = 0; r < kC1Rows; ++r) {
        const int t = t0 - 1 + r;
        if (t < 0 || t >= kFrames) continue;
        for (int g = threadIdx.x; g < kBins; g += kC1Threads) {
          const float z = norm_bn(lpb[t * kBins + g], mn, range, kc);
          zl[r * kC1Ts + (g & 3) * kC1Plane + (g >> 2) + kC1Qoff] = z;
        }
      }
    }
    __syncthreads();

    for (int tile = 0; tile < n_tiles; ++tile) {
      int m = tile * 32 + li;
      const bool mvalid = m < n_pos;
      m = mvalid ? m : n_pos - 1;
      const int tr = m / kC1Groups;
      const int mf = m - tr * kC1Groups;
      const int lane_base = tr * kC1Ts + mf + kC1Qoff - 5;
      const int fbase = 4 * mf + kodd - 19;

      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
      switch (wave) {
        case 0:
        case 1: c1_tile<0, 0>(zl, lane_base, kodd, fbase, aA, aB, breg, acc); break;
        case 2: c1_tile<0, 1>(zl, lane_base, kodd, fbase, aA, aB, breg, acc); break;
        default: c1_tile<1, 1>(zl, lane_base, kodd, fbase, aA, aB, breg, acc); break;
      }

      // K-partials -> LDS.  C layout 32x32: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
      float* rbuf = red + (tile & 1) * 4 * kRedTile;
      {
        float* rp = rbuf + wave * kRedTile + (4 * kodd) * kRedRow + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) rp[((r & 3) + 8 * (r >> 2)) * kRedRow] = acc[r];
      }
      __syncthreads();
      {
        // lane -> (tile row li, output channel oc): 4 adjacent bins, summed over the 4 K-partials
        const float4* p = reinterpret_cast<const float4*>(rbuf + li * kRedRow + 4 * oc);
        float4 v0 = p[0], v1 = p[kRedTile / 4], v2 = p[2 * kRedTile / 4], v3 = p[3 * kRedTile / 4];
        float4 o;
        o.x = fmaxf(((v0.x + v1.x) + (v2.x + v3.x)) + obias, 0.0f);
        o.y = fmaxf(((v0.y + v1.y) + (v2.y + v3.y)) + obias, 0.0f);
        o.z = fmaxf(((v0.z + v1.z) + (v2.z + v3.z)) + obias, 0.0f);
        o.w = fmaxf(((v0.w + v1.w) + (v2.w + v3.w)) + obias, 0.0f);
        if (mvalid) {
          float* dst = c1 + (((int64_t)b * 8 + oc) * kFrames + (t0 + tr)) * kFreqC + 4 * mf;
          *reinterpret_cast<float4*>(dst) = o;
        }
      }
      // red is double-buffered: the barrier of tile+1 separates these reads from tile+2's writes
    }
  }
}

void launch_contour1(const float* lp, const int* mm, const float* bfrag, const float* bias, float* c1,
                     int n_windows, LogConsts kc, int n_cu, hipStream_t stream) {
  const int items = n_windows * kC1Slabs;
  const int grid = items < 2 * n_cu ? items : 2 * n_cu;
  hipLaunchKernelGGL(contour1_kernel, dim3(grid), dim3(kC1Threads), 0, stream, lp, mm, bfrag, bias,
                     c1, n_windows, kc);
}

}  // namespace bp
