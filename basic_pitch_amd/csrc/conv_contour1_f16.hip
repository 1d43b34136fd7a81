// Contour branch, first convolution — split-precision matrix-core version (default path).
//
// Same operator as conv_contour1.hip (NormalizedLog tail + BatchNorm + virtual harmonic stack +
// Conv2D 8->8 (3 x 39) "same" + folded BN + ReLU; reference: basic_pitch/layers/signal.py:177-183,
// models.py:187-189, nn.py:69-88, models.py:241-250), but the contraction runs on the f16 matrix
// cores with BOTH operands split into an f16 "hi" and an f16 "lo" part:
//        x = hi(x) + lo(x),   hi = rn_f16(x),  lo = rn_f16(x - hi)        (22 significand bits kept)
//        a*b ~= hi(a)hi(b) + lo(a)hi(b) + hi(a)lo(b)                      (dropped term <= 2^-22 |ab|)
// Every f16 x f16 product is exact in the fp32 accumulator of v_mfma_f32_32x32x16_f16, so the result
// differs from exact arithmetic by ~2^-22 per product plus fp32 accumulation — measured on the
// oracle: 1.6e-6 max abs on c1 (values up to 4.6), i.e. not worse than a plain f32 convolution
// (5.5e-6).  Three f16 MFMAs (16 k-slots each, 32 cycles) replace eight f32 MFMAs (2 k-slots, 64
// cycles): 5.3x fewer matrix-core cycles at fp32-class accuracy.  The exact-f32 kernel stays
// available (BP_FLAG_F32_MFMA) as the A/B reference.
//
// Mapping (per 4-frame slab of one window; a workgroup = 4 waves):
//   * LDS holds the harmonic stack of the slab ONCE, channel-last and already split:
//     S_hi / S_lo [6 frames][4 phase planes][76 groups] x (8 channels x f16 = 16 B).  The stack's crop to
//     264 bins, the conv's zero padding and the out-of-image frames are all just zeros in this image,
//     so the main loop has no masking.  Phase planes (bin mod 4) make the 32 lanes of an A read
//     consecutive 16-byte slots: conflict-free ds_read_b128.
//   * M = 32 consecutive (frame, 4-bin group) positions, N = 32 = 8 out channels x 4 adjacent bins
//     (Toeplitz, 42/39 extra taps), one MFMA k-step = 2 adjacent taps x 8 input channels.
//   * K = 3 frames x 21 tap pairs = 63 k-steps, split over the 4 waves (16/16/16/15); each wave keeps
//     ITS hi and lo B fragments in VGPRs (128) for the whole kernel; per k-step 2 ds_read_b128 + 3 MFMA.
//   * fixed-order 4-way reduction of the K partials through LDS, + bias, ReLU, 16-byte stores.
//
// Roofline: bound = f16 MFMA issue (3 MFMA per 16 k-slots); algorithmic work is still the operator's
// 680,030,208 FLOP / window.  Bytes: 212,592 read (lp) + 1,453,056 written (c1).
#include "bp_common.h"

namespace bp {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int kH1Threads = 256;
constexpr int kH1Slab = 4;                      // output frames per work item (172 = 43 * 4)
constexpr int kH1Rows = kH1Slab + 2;
constexpr int kH1Q = 76;                        // 16-byte slots per phase plane
constexpr int kH1Rs = 4 * kH1Q + 2;             // 306 slots per frame, = 2 (mod 16)
constexpr int kH1Slabs = kFrames / kH1Slab;     // 43
constexpr int kH1Groups = kFreqC / 4;           // 66
constexpr int kH1Tiles = (kH1Slab * kH1Groups + 31) / 32;  // 9
constexpr int kH1StepsTotal = 63;               // 3 frames x 21 tap pairs
constexpr int kH1StepsWave = 16;                // k-steps per wave (wave 3: 15 + one zero step)
constexpr int kH1RedRow = 36;
constexpr int kH1RedTile = 32 * kH1RedRow;
constexpr int kH1ZRow = 312;                    // fp32 scratch row (309 bins)
static_assert(kFrames % kH1Slab == 0, "slabs tile the window");
static_assert(kH1Rows * kH1ZRow <= 4 * kH1RedTile, "z scratch aliases the reduction buffer");

// One k-step: 2 adjacent taps (e = 2*ep + h) x 8 channels; stack bin of tap e for group m_f is
// 4*m_f + e - 19, stored at slot plane r = (e + 1) & 3, q = m_f + ((e + 1) >> 2).
template <int WAVE>
__device__ __forceinline__ void h1_tile(const uint4* __restrict__ s_hi, const uint4* __restrict__ s_lo,
                                        int base_lo, int base_hi, const uint4 (&bh)[kH1StepsWave],
                                        const uint4 (&bl)[kH1StepsWave], f32x16& acc_main,
                                        f32x16& acc_corr) {
#pragma unroll
  for (int s = 0; s < kH1StepsWave; ++s) {
    const int step = WAVE * kH1StepsWave + s;
    if (step >= kH1StepsTotal) continue;
    const int dt = step / 21, ep = step - 21 * dt;
    const int r0 = (2 * ep + 1) & 3, q0 = (2 * ep + 1) >> 2;
    const int imm = dt * kH1Rs + r0 * kH1Q + q0;
    const int slot = ((r0 == 1) ? base_lo : base_hi) + imm;
    const uint4 ah_u = s_hi[slot];
    const uint4 al_u = s_lo[slot];
    const f16x8 ah = __builtin_bit_cast(f16x8, ah_u);
    const f16x8 al = __builtin_bit_cast(f16x8, al_u);
    const f16x8 bhv = __builtin_bit_cast(f16x8, bh[s]);
    const f16x8 blv = __builtin_bit_cast(f16x8, bl[s]);
    acc_main = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bhv, acc_main, 0, 0, 0);
    acc_corr = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bhv, acc_corr, 0, 0, 0);
    acc_corr = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, blv, acc_corr, 0, 0, 0);
  }
}

__global__ __launch_bounds__(kH1Threads, 2) void contour1_f16_kernel(
    const float* __restrict__ lp, const int* __restrict__ mm, const uint4* __restrict__ bfrag,
    const float* __restrict__ bias, float* __restrict__ c1, int n_windows, LogConsts kc) {
  __shared__ __attribute__((aligned(16))) uint4 s_hi[kH1Rows * kH1Rs];
  __shared__ __attribute__((aligned(16))) uint4 s_lo[kH1Rows * kH1Rs];
  __shared__ __attribute__((aligned(16))) float red[4 * kH1RedTile];
  float* zrow = red;  // staging scratch, dead before the first partial is written

  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int h = lane >> 5;
  const int li = lane & 31;

  // B fragments: [wave][step][hi|lo][lane] x 16 B, resident for the whole kernel
  uint4 bh[kH1StepsWave], bl[kH1StepsWave];
  {
    const uint4* bp_ = bfrag + (size_t)wave * kH1StepsWave * 2 * 64 + lane;
#pragma unroll
    for (int s = 0; s < kH1StepsWave; ++s) {
      bh[s] = bp_[(2 * s) * 64];
      bl[s] = bp_[(2 * s + 1) * 64];
    }
  }
  const int oc = 2 * wave + h;  // epilogue: this lane finalises output channel oc of tile row li
  const float obias = bias[oc];

  const int n_items = n_windows * kH1Slabs;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int b = item / kH1Slabs;
    const int t0 = (item - b * kH1Slabs) * kH1Slab;

    // ---- stage A: z = BN(normalised log-power) of frames t0-1 .. t0+4 as fp32 rows
    __syncthreads();  // previous item's epilogue reads of red / MFMA reads of s_* are done
    {
      const float mn = ord2f(mm[2 * b]);
      const float range = ord2f(mm[2 * b + 1]) - mn;
      const float* lpb = lp + (int64_t)b * kFrames * kBins;
      for (int i = threadIdx.x; i < kH1Rows * kH1ZRow; i += kH1Threads) {
        const int r = i / kH1ZRow, g = i - r * kH1ZRow;
        const int t = t0 - 1 + r;
        float z = 0.0f;
        if (g < kBins && t >= 0 && t < kFrames) z = norm_bn(lpb[t * kBins + g], mn, range, kc);
        zrow[i] = z;
      }
    }
    __syncthreads();
    // ---- stage B: channel-last, hi/lo-split, cropped + zero-padded stack image
    for (int i = threadIdx.x; i < kH1Rows * 4 * kH1Q; i += kH1Threads) {
      const int r = i / (4 * kH1Q);
      const int slot = i - r * (4 * kH1Q);
      const int pl = slot / kH1Q, q = slot - pl * kH1Q;
      const int f = 4 * q + pl - 20;  // stack bin held by this slot
      const int t = t0 - 1 + r;
      const bool inside = (f >= 0) && (f < kFreqC) && (t >= 0) && (t < kFrames);
      f16x8 vh, vl;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int g = f + harm_shift(c);
        float v = 0.0f;
        if (inside && g >= 0 && g < kBins) v = zrow[r * kH1ZRow + g];
        const _Float16 hi = (_Float16)v;
        vh[c] = hi;
        vl[c] = (_Float16)(v - (float)hi);
      }
      s_hi[r * kH1Rs + slot] = __builtin_bit_cast(uint4, vh);
      s_lo[r * kH1Rs + slot] = __builtin_bit_cast(uint4, vl);
    }
    __syncthreads();

    for (int tile = 0; tile < kH1Tiles; ++tile) {
      int m = tile * 32 + li;
      const bool mvalid = m < kH1Slab * kH1Groups;
      m = mvalid ? m : kH1Slab * kH1Groups - 1;
      const int tr = m / kH1Groups;
      const int mf = m - tr * kH1Groups;
      const int lane_slot = tr * kH1Rs + mf;
      const int base_lo = lane_slot + h * kH1Q;            // tap plane 1 -> 2 (same group)
      const int base_hi = lane_slot + h * (1 - 3 * kH1Q);  // tap plane 3 -> 0 of the next group

      f32x16 acc_main, acc_corr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc_main[r] = 0.0f;
        acc_corr[r] = 0.0f;
      }
      switch (wave) {
        case 0: h1_tile<0>(s_hi, s_lo, base_lo, base_hi, bh, bl, acc_main, acc_corr); break;
        case 1: h1_tile<1>(s_hi, s_lo, base_lo, base_hi, bh, bl, acc_main, acc_corr); break;
        case 2: h1_tile<2>(s_hi, s_lo, base_lo, base_hi, bh, bl, acc_main, acc_corr); break;
        default: h1_tile<3>(s_hi, s_lo, base_lo, base_hi, bh, bl, acc_main, acc_corr); break;
      }

      // K-partials -> LDS (single buffer: barrier before the writes and before the reads)
      __syncthreads();
      {
        float* rp = red + wave * kH1RedTile + (4 * h) * kH1RedRow + li;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          rp[((r & 3) + 8 * (r >> 2)) * kH1RedRow] = acc_main[r] + acc_corr[r];
      }
      __syncthreads();
      {
        const float4* p = reinterpret_cast<const float4*>(red + li * kH1RedRow + 4 * oc);
        const float4 v0 = p[0], v1 = p[kH1RedTile / 4], v2 = p[2 * kH1RedTile / 4], v3 = p[3 * kH1RedTile / 4];
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
    }
  }
}

void launch_contour1_f16(const float* lp, const int* mm, const void* bfrag, const float* bias, float* c1,
                         int n_windows, LogConsts kc, int n_cu, hipStream_t stream) {
  const int items = n_windows * kH1Slabs;
  const int grid = items < 2 * n_cu ? items : 2 * n_cu;
  hipLaunchKernelGGL(contour1_f16_kernel, dim3(grid), dim3(kH1Threads), 0, stream, lp, mm,
                     reinterpret_cast<const uint4*>(bfrag), bias, c1, n_windows, kc);
}

}  // namespace bp
