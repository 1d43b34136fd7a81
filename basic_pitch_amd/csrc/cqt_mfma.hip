// CQT front end on the f16 matrix cores with split operands (default path): the 8 x decimate-by-2
// pyramid and the 9-level complex filterbank.  Same operators as cqt_pyramid.hip / cqt_filterbank.hip
// (the exact-f32 A/B reference kernels), same reference lines:
//   basic_pitch/layers/nnaudio.py:259-284, 636-638   downsampling_by_n: zero-pad 127, 256-tap FIR, stride 2
//   basic_pitch/layers/nnaudio.py:216-256, 640-661   get_cqt_complex per level, * sqrt(lengths), magnitude
//   basic_pitch/layers/signal.py:171-178             power, 10*log10(power + 1e-10), per-example min / max
//
// Every operand x is carried as x = hi + lo (two f16, 22 significand bits) and every product as
// hi*hi + lo*hi + hi*lo on v_mfma_f32_16x16x32_f16 with fp32 accumulation: fp32-class accuracy
// (dropped term <= 2^-22 |ab|) at 16x the f32 matrix rate.  The signal is split ONCE while it is staged
// into LDS; the filters are split on the host.
//
// Decimator.  y[n] = sum_j h[j] xz[2n + j - 127] has ONE filter, so the GEMM is built from the band
// structure instead: a tile is 16 row-blocks x 16 outputs, A[m][i] = xp[32 m + i] (Hankel, 32-sample row
// stride, 16-byte aligned reads), B[i][u] = h[i - 2u] (Toeplitz band, 256 of the 288 k-slots non-zero:
// 89 % dense), 9 k-steps.  The 9 B fragments are the same for all 8 levels and stay in VGPRs.
//
// Filterbank.  Per (window, level, 16-frame tile): [16 frames x K] x [K x 72 filter columns] with the
// Hankel A[t][i] = xp[t*hop + i].  K is clipped to the kernels' support (61 % of 256 taps) in 32-tap
// steps, the four waves own (re 0-15 | im 0-15 | re 16-31 + half of {re,im} 32-35 | im 16-31 + other half):
// 7 k-steps each, B fragments resident.  Aligned 16-byte A reads need t*hop = 0 (mod 8): levels with
// hop < 8 stage 8/hop shifted copies of their (short) signal.  All 9 levels run in ONE launch (the
// filters are level-independent), items = (window, level, tile).
//
// LDS rows are skewed so every ds_read_b128 of a fragment is bank-conflict free (row stride = 2 or 6
// sixteen-byte units mod 16).
//
// Roofline: f16 MFMA issue.  Algorithmic work per window: pyramid 22.4 MFLOP, filterbank 57.1 MFLOP
// (SURVEY.md §8a row a8).  Bytes per window: pyramid 175,376 + 174,764 read, 174,764 written;
// filterbank 350,140 read + 212,592 written.
#include "bp_common.h"

namespace bp {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

// f16 has 5 exponent bits: the residual of a sample (<= 2^-12 |x|) and the small taps of the filters
// (1e-3 .. 1e-8) would land in its subnormal range and lose their low bits — measured 8.7e-6 instead of
// 2.4e-6 on the CQT magnitudes.  So residuals are stored multiplied by 2^11 and the taps pre-scaled by a
// power of two; the three product classes have their own accumulators and are recombined with exact
// power-of-two factors:  result = (hh + (lh + hl) * 2^-11) * 2^-tapshift.
constexpr float kDmTapUnscale = 1.0f / 1024.0f;   // decimator taps are packed * 2^10 (bp_api.hip)
constexpr float kFmTapUnscale = 1.0f / 4096.0f;   // CQT kernels are packed * 2^12

__device__ __forceinline__ void split8(const float (&v)[8], uint4& hi, uint4& lo) {
  split_f16x2_rn(f32x2{v[0], v[1]}, hi.x, lo.x);
  split_f16x2_rn(f32x2{v[2], v[3]}, hi.y, lo.y);
  split_f16x2_rn(f32x2{v[4], v[5]}, hi.z, lo.z);
  split_f16x2_rn(f32x2{v[6], v[7]}, hi.w, lo.w);
}

#define BP_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0)

// ================================================================================================
// decimate by 2
constexpr int kDmThreads = 256;
constexpr int kDmTiles = 8;                           // MFMA tiles (256 outputs each) per workgroup
constexpr int kDmOutPerWg = 256 * kDmTiles;           // 2048
constexpr int kDmPos = 2 * kDmOutPerWg + 256;         // 4352 staged input positions
constexpr int kDmUnits = (kDmPos / 32) * 6;           // 16-byte units, 32-sample rows skewed to 6 units
constexpr int kDmSteps = 9;                           // 288 k-slots (286 used)

// one workgroup's share of a level: outputs o0 .. o0 + kDmOutPerWg of window row x -> y
// (x may point into LDS: the tail kernel keeps the deep levels there; y_lds, if not null, receives a copy of y)
__device__ __forceinline__ void dm_block(const float* x, int len_in, float* __restrict__ y, int len_out, int o0,
                                         const uint4 (&hh)[kDmSteps], const uint4 (&hl)[kDmSteps],
                                         uint4* __restrict__ s_hi, uint4* __restrict__ s_lo, float* y_lds = nullptr) {
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int in0 = 2 * o0 - 127;  // xp[p] = xz[in0 + p]
  // only the 32-sample rows this block's outputs live in (a short level or a level's last block needs a fraction of the
  // 4352 positions).  Whole rows, because a row's out-of-band taps are zero WEIGHTS times whatever the slot holds: it
  // must be finite.  Rows beyond them feed outputs that are not stored.
  const int n_here = len_out - o0 < kDmOutPerWg ? len_out - o0 : kDmOutPerWg;
  const int q_rows = 4 * ((n_here + 15) / 16) + 32;
  const int q_end = q_rows < kDmPos / 8 ? q_rows : kDmPos / 8;
  for (int q = threadIdx.x; q < q_end; q += kDmThreads) {
    float v[8];
    const int g0 = in0 + 8 * q;
    if (g0 >= 0 && g0 + 8 <= len_in) {
      // interior: two 16-byte loads (dword aligned is enough for global / LDS vector loads) instead of eight 4-byte
      // ones at a 32-byte lane stride
      const float4 a = *reinterpret_cast<const float4*>(x + g0);
      const float4 c = *reinterpret_cast<const float4*>(x + g0 + 4);
      v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = c.x, v[5] = c.y, v[6] = c.z, v[7] = c.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int g = g0 + e;
        const bool ok = g >= 0 && g < len_in;
        v[e] = ok ? x[ok ? g : 0] : 0.0f;
      }
    }
    uint4 hi, lo;
    split8(v, hi, lo);
    const int unit = q + 2 * (q >> 2);
    s_hi[unit] = hi;
    s_lo[unit] = lo;
  }
  lds_barrier();

  const int m = lane & 15, kg = lane >> 4;
  for (int tile = wave; tile < kDmTiles; tile += kDmThreads / 64) {
    const int mb = tile * 16;
    if (o0 + 16 * mb >= len_out) break;
    f32x4 a_hh = {0.f, 0.f, 0.f, 0.f}, a_lh = a_hh, a_hl = a_hh;
    const int base = 6 * (mb + m) + kg;
#pragma unroll
    for (int s = 0; s < kDmSteps; ++s) {
      const uint4 ah = s_hi[base + 6 * s];
      const uint4 al = s_lo[base + 6 * s];
      a_hh = BP_MFMA16(ah, hh[s], a_hh);
      a_lh = BP_MFMA16(al, hh[s], a_lh);
      a_hl = BP_MFMA16(ah, hl[s], a_hl);
    }
    // C: col u = lane & 15, row (row-block) = 4*kg + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = o0 + 16 * (mb + 4 * kg + r) + m;
      if (n < len_out) {
        const float v = (a_hh[r] + (a_lh[r] + a_hl[r]) * kLoUnscale) * kDmTapUnscale;
        y[n] = v;
        if (y_lds) y_lds[n] = v;
      }
    }
  }
}

// Persistent: a workgroup loads the 18 filter fragments once (73 KB per workgroup from L2 — four times the 17 KB of
// signal a block stages) and walks (window, block) items.
__global__ __launch_bounds__(kDmThreads) void decimate2_mfma_kernel(const float* __restrict__ src,
                                                                    int64_t src_stride, int len_in,
                                                                    float* __restrict__ dst,
                                                                    int64_t dst_stride, int len_out,
                                                                    const uint4* __restrict__ hfrag, int n_windows) {
  __shared__ __attribute__((aligned(16))) uint4 s_hi[kDmUnits];
  __shared__ __attribute__((aligned(16))) uint4 s_lo[kDmUnits];
  const int lane = threadIdx.x & 63;
  uint4 hh[kDmSteps], hl[kDmSteps];
#pragma unroll
  for (int s = 0; s < kDmSteps; ++s) {
    hh[s] = hfrag[s * 64 + lane];
    hl[s] = hfrag[(kDmSteps + s) * 64 + lane];
  }
  const int blocks = (len_out + kDmOutPerWg - 1) / kDmOutPerWg;
  const int n_items = n_windows * blocks;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int b = item / blocks, blk = item - b * blocks;
    dm_block(src + (int64_t)b * src_stride, len_in, dst + (int64_t)b * dst_stride, len_out, blk * kDmOutPerWg, hh, hl,
             s_hi, s_lo);
    lds_barrier();  // the block's MFMAs are done with s_hi / s_lo
  }
}

// The deep levels (<= 2740 samples per window) are one or two blocks each: as separate launches every level pays ~12 us
// of dispatch, load latency and ramp.  Here ONE workgroup per window runs levels `first` .. `last` back to back with the
// same block routine; every level is written to HBM as usual AND kept in LDS, from where the next level is staged (a
// round trip through HBM between levels costs as much as the separate launches did).
constexpr int kDmTailMax = 2 * kDmOutPerWg;  // longest level the tail takes (samples)
struct DmTail {
  int first, last;
  int len[10];
  int off[10];
};

__global__ __launch_bounds__(kDmThreads) void decimate2_tail_kernel(float* __restrict__ pyr, int64_t pyr_stride,
                                                                    const uint4* __restrict__ hfrag, DmTail t) {
  __shared__ __attribute__((aligned(16))) uint4 s_hi[kDmUnits];
  __shared__ __attribute__((aligned(16))) uint4 s_lo[kDmUnits];
  __shared__ float sig[2][kDmTailMax];
  const int lane = threadIdx.x & 63;
  uint4 hh[kDmSteps], hl[kDmSteps];
#pragma unroll
  for (int s = 0; s < kDmSteps; ++s) {
    hh[s] = hfrag[s * 64 + lane];
    hl[s] = hfrag[(kDmSteps + s) * 64 + lane];
  }
  float* pw = pyr + (int64_t)blockIdx.x * pyr_stride;
  const float* x = pw + t.off[t.first - 1];  // the first level still comes from HBM
  int cur = 0;
  for (int k = t.first; k <= t.last; ++k) {
    const int len_in = t.len[k - 1], len_out = t.len[k];
    for (int o0 = 0; o0 < len_out; o0 += kDmOutPerWg) {
      dm_block(x, len_in, pw + t.off[k], len_out, o0, hh, hl, s_hi, s_lo, sig[cur]);
      lds_barrier();  // the block's MFMAs are done with s_hi / s_lo, its outputs are in sig[cur]
    }
    x = sig[cur];
    cur ^= 1;
  }
}

// ================================================================================================
// filterbank
constexpr int kFmThreads = 256;
constexpr int kFmTileFrames = 16;
constexpr int kFmTilesPerLevel = (kFrames + kFmTileFrames - 1) / kFmTileFrames;  // 11
constexpr int kFmSteps = 7;                       // k-steps (32 taps) per wave
constexpr int kFmMaxUnits = (15 * 512 + 256 + 16 * 16) / 8;  // hop 512 (extended 44.1 kHz CQT): 7936 samples + 16 skews
// exchange between the role waves and the epilogue: layer 0 = re / im planes [frame][36 filters] (row 37), layer 1 =
// the second K half of filters 32..35 [frame][4] plus a column that stays zero (row 5), so that the epilogue reads
// re = L0re[fr][k] + L1re[fr][k >= 32 ? k - 32 : 4] without a branch
constexpr int kFmL0Row = 37, kFmL0 = kFmTileFrames * kFmL0Row;
constexpr int kFmL1Row = 5, kFmL1 = kFmTileFrames * kFmL1Row;
constexpr int kFmExch = 2 * kFmL0 + 2 * kFmL1;  // floats

__host__ __device__ constexpr int fm_copies(int hop) { return hop >= 8 ? 1 : 8 / hop; }
__host__ __device__ constexpr int fm_copy_units(int hop) { return hop == 4 ? 40 : 36; }  // per shifted copy

// uint4 index of the 8 samples xp[t*HOP + off .. +7] of the tile (off a multiple of 8)
template <int HOP>
__device__ __forceinline__ int fm_unit(int t, int off) {
  if constexpr (HOP >= 32) {
    return (t * (HOP + 16) + off + 16 * (off / HOP)) >> 3;
  } else if constexpr (HOP >= 8) {
    return (t * HOP + off) >> 3;
  } else {
    constexpr int C = fm_copies(HOP);
    return (t % C) * fm_copy_units(HOP) + (t / C) + (off >> 3);
  }
}

template <int HOP>
__device__ __forceinline__ void fm_stage(const float* __restrict__ x, int L, int t0, uint4* __restrict__ s_hi,
                                         uint4* __restrict__ s_lo) {
  // xp[p] = reflect(x)[t0*HOP + p - 128]    (nnaudio.py:229, 300-301)
  auto sample = [&](int p) {
    int g = t0 * HOP + p - 128;
    g = g < 0 ? -g : g;
    g = g >= L ? 2 * (L - 1) - g : g;
    g = g < 0 ? 0 : (g >= L ? L - 1 : g);  // only reachable for the padding frames 172..175
    return x[g];
  };
  if constexpr (HOP >= 8) {
    constexpr int NBLK = (15 * HOP + 256) / 8;
    for (int q = threadIdx.x; q < NBLK; q += kFmThreads) {
      float v[8];
      const int g0 = t0 * HOP + 8 * q - 128;
      if (g0 >= 0 && g0 + 8 <= L) {  // interior (no reflection): two 16-byte loads instead of eight 4-byte ones
        const float4 a = *reinterpret_cast<const float4*>(x + g0);
        const float4 c = *reinterpret_cast<const float4*>(x + g0 + 4);
        v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = c.x, v[5] = c.y, v[6] = c.z, v[7] = c.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = sample(8 * q + e);
      }
      uint4 hi, lo;
      split8(v, hi, lo);
      const int unit = (HOP >= 32) ? q + 2 * ((8 * q) / HOP) : q;
      s_hi[unit] = hi;
      s_lo[unit] = lo;
    }
  } else {
    constexpr int C = fm_copies(HOP);
    constexpr int NBLK = 16 / C + 32;  // 8-sample blocks a copy must hold
    for (int i = threadIdx.x; i < C * NBLK; i += kFmThreads) {
      const int c = i / NBLK, q = i - c * NBLK;
      float v[8];
      const int g0 = t0 * HOP + 8 * q + c * HOP - 128;
      if (g0 >= 0 && g0 + 8 <= L) {  // interior: two (dword-aligned) 16-byte loads
        const float4 a = *reinterpret_cast<const float4*>(x + g0);
        const float4 d = *reinterpret_cast<const float4*>(x + g0 + 4);
        v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = d.x, v[5] = d.y, v[6] = d.z, v[7] = d.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = sample(8 * q + c * HOP + e);
      }
      uint4 hi, lo;
      split8(v, hi, lo);
      s_hi[c * fm_copy_units(HOP) + q] = hi;
      s_lo[c * fm_copy_units(HOP) + q] = lo;
    }
  }
}

// segment A: NA steps from tap BASE_A: 16 filters (COL_A ..) of plane PLANE (0 re, 1 im) of layer 0; segment B (roles
// 2, 3): 2 steps of the {re,im} 32..35 columns starting at tap BASE_B, into layer LAYER_B
template <int ROLE>
struct FmRole;
template <> struct FmRole<0> { static constexpr int NA = 7, BASE_A = 16, NB = 0, BASE_B = 0, PLANE = 0, COL_A = 0, LAYER_B = 0; };
template <> struct FmRole<1> { static constexpr int NA = 7, BASE_A = 16, NB = 0, BASE_B = 0, PLANE = 1, COL_A = 0, LAYER_B = 0; };
template <> struct FmRole<2> { static constexpr int NA = 5, BASE_A = 48, NB = 2, BASE_B = 64, PLANE = 0, COL_A = 16, LAYER_B = 0; };
template <> struct FmRole<3> { static constexpr int NA = 5, BASE_A = 48, NB = 2, BASE_B = 128, PLANE = 1, COL_A = 16, LAYER_B = 1; };

// The unit of lane (t, kg) at tap offset B + 32 s + 8 kg as "a per-lane base + a compile-time constant": for HOP >= 32
// the row skew adds 2 units per HOP samples crossed, and the 8 kg part crosses a row boundary only when (B + 32 s) mod
// HOP = HOP - 16 and kg >= 2; so two bases (ua, and ub = ua + 2 for kg >= 2) and an immediate per step replace ~4 VALU
// operations per fragment read.  B and 32 s are multiples of 16.
template <int HOP>
struct FmAddr {
  int ua, ub;
  __device__ __forceinline__ FmAddr(int t, int kg) {
    ua = fm_unit<HOP>(t, 8 * kg);
    ub = ua + (HOP >= 32 ? 2 * (kg >> 1) : 0);
  }
  // constant part of fm_unit<HOP>(t, off + 8 kg) - fm_unit<HOP>(t, 8 kg) for kg < 2 (off a multiple of 16)
  static __device__ constexpr int delta(int off) {
    if (HOP >= 32) return (off >> 3) + 2 * (off / HOP);
    return off >> 3;
  }
  static __device__ constexpr bool cross(int off) { return HOP >= 32 && (off % HOP) == HOP - 16; }
  __device__ __forceinline__ int unit(int off) const { return (cross(off) ? ub : ua) + delta(off); }
};

template <int ROLE, int HOP>
__device__ __forceinline__ void fm_role_compute(const uint4* __restrict__ s_hi, const uint4* __restrict__ s_lo,
                                                const uint4 (&bh)[kFmSteps], const uint4 (&bl)[kFmSteps],
                                                float* __restrict__ exch, int lane) {
  using R = FmRole<ROLE>;
  const int t = lane & 15, kg = lane >> 4;
  const FmAddr<HOP> addr(t, kg);
  f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 a_hh = z4, a_lh = z4, a_hl = z4;
#pragma unroll
  for (int s = 0; s < R::NA; ++s) {
    const int u = addr.unit(R::BASE_A + 32 * s);
    const uint4 ah = s_hi[u], al = s_lo[u];
    a_hh = BP_MFMA16(ah, bh[s], a_hh);
    a_lh = BP_MFMA16(al, bh[s], a_lh);
    a_hl = BP_MFMA16(ah, bl[s], a_hl);
  }
  // C: col = lane & 15 (filter), row = 4*kg + r (frame)
  float* ea = exch + R::PLANE * kFmL0 + (kg * 4) * kFmL0Row + R::COL_A + t;
#pragma unroll
  for (int r = 0; r < 4; ++r) ea[r * kFmL0Row] = (a_hh[r] + (a_lh[r] + a_hl[r]) * kLoUnscale) * kFmTapUnscale;
  if constexpr (R::NB > 0) {
    f32x4 b_hh = z4, b_lh = z4, b_hl = z4;
#pragma unroll
    for (int s = 0; s < R::NB; ++s) {
      const int u = addr.unit(R::BASE_B + 32 * s);
      const uint4 ah = s_hi[u], al = s_lo[u];
      b_hh = BP_MFMA16(ah, bh[R::NA + s], b_hh);
      b_lh = BP_MFMA16(al, bh[R::NA + s], b_lh);
      b_hl = BP_MFMA16(ah, bl[R::NA + s], b_hl);
    }
    // columns 0..3 = re of filters 32..35, 4..7 = im; 8..15 carry zero weights
    if (t < 8) {
      const int plane = t >> 2, c = t & 3;
      float* eb = R::LAYER_B == 0 ? exch + plane * kFmL0 + (kg * 4) * kFmL0Row + 32 + c
                                  : exch + 2 * kFmL0 + plane * kFmL1 + (kg * 4) * kFmL1Row + c;
      constexpr int row = R::LAYER_B == 0 ? kFmL0Row : kFmL1Row;
#pragma unroll
      for (int r = 0; r < 4; ++r) eb[r * row] = (b_hh[r] + (b_lh[r] + b_hl[r]) * kLoUnscale) * kFmTapUnscale;
    }
  }
}

// one (window, level, tile) item; x = that window's signal of the level, L its length, `level` its index in the
// pyramid of `n_levels` levels (hop = hop0 >> level), n_bins the CQT width (levels * 36 - 15)
template <int HOP>
__device__ __forceinline__ void fm_item(const float* __restrict__ x, int L, int b, int level, int n_levels,
                                        int n_bins, int tile, const uint4 (&bh)[kFmSteps], const uint4 (&bl)[kFmSteps],
                                        const float* __restrict__ sqrt_len, float* __restrict__ lp,
                                        float2* __restrict__ mmp, const LogConsts& kc, uint4* s_hi, uint4* s_lo,
                                        float* exch, int role, int lane) {
  const int t0 = tile * kFmTileFrames;
  // keep the per-level LDS address arithmetic inside the item (hoisting all 36 level x role variants
  // out of the persistent loop costs > 100 VGPRs)
  asm volatile("" : "+v"(lane));

  lds_barrier();  // previous item's epilogue is done with exch, its MFMAs with s_hi / s_lo
  fm_stage<HOP>(x, L, t0, s_hi, s_lo);
  lds_barrier();
  switch (role) {
    case 0: fm_role_compute<0, HOP>(s_hi, s_lo, bh, bl, exch, lane); break;
    case 1: fm_role_compute<1, HOP>(s_hi, s_lo, bh, bl, exch, lane); break;
    case 2: fm_role_compute<2, HOP>(s_hi, s_lo, bh, bl, exch, lane); break;
    default: fm_role_compute<3, HOP>(s_hi, s_lo, bh, bl, exch, lane); break;
  }
  lds_barrier();

  // epilogue: 16 frames x 36 filters -> * sqrt(len), magnitude, log-power, tile extrema.  576 outputs = 2.25 passes of
  // the workgroup (the third pass is wave 0 alone), branch-free inside a pass.
  float vmin = __int_as_float(0x7f800000), vmax = -__int_as_float(0x7f800000);
  const float kln2 = 0.69314718055994531f * kc.s0 * kc.s1;  // log2 -> 10 log10
  const int bin0 = (n_levels - 1 - level) * kBpo - 15;      // nnaudio.py:640-642
  float* lp_tile = lp + ((int64_t)b * kFrames + t0) * n_bins + bin0;
  const float* sl_tile = sqrt_len + bin0;
  const float* l0re = exch;
  const float* l0im = exch + kFmL0;
  const float* l1re = exch + 2 * kFmL0;
  const float* l1im = l1re + kFmL1;
#pragma unroll
  for (int pass = 0; pass < 3; ++pass) {
    if (pass == 2 && role != 0) break;  // wave-uniform
    const int idx = threadIdx.x + pass * kFmThreads;  // < 576 in every pass that runs
    const int fr = idx / kBpo, k = idx - fr * kBpo;
    const bool ok = t0 + fr < kFrames && bin0 + k >= 0;
    const int c1 = k >= 32 ? k - 32 : 4;
    float re = l0re[fr * kFmL0Row + k] + l1re[fr * kFmL1Row + c1];
    float im = l0im[fr * kFmL0Row + k] + l1im[fr * kFmL1Row + c1];
    const float sl = sl_tile[ok ? k : 15];
    re = __fmul_rn(re, sl);  // nnaudio.py:650: scale before squaring
    im = __fmul_rn(im, sl);
    // nnaudio.py:661 magnitude, signal.py:174-175 power and 10 log10: the hardware's 1-ulp sqrt and log2 (the precise
    // library forms are ~35 VALU instructions per bin of a kernel whose pace the VALU sets; 1e-7 relative on lp)
    const float mag = __builtin_amdgcn_sqrtf(__fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im)));
    const float pw = __fmul_rn(mag, mag);
    const float v = __fmul_rn(__builtin_amdgcn_logf(__fadd_rn(pw, kc.eps)), kln2);
    if (ok) lp_tile[fr * n_bins + k] = v;
    vmin = fminf(vmin, ok ? v : vmin);
    vmax = fmaxf(vmax, ok ? v : vmax);
  }
  vmin = wave_min_lane63(vmin);
  vmax = wave_max_lane63(vmax);
  if (lane == 63)
    mmp[(((int64_t)b * n_levels + level) * kFmTilesPerLevel + tile) * 4 + role] = make_float2(vmin, vmax);
}

// Pyramid geometry of a launch: the reference's 22.05 kHz model (9 levels, hop 256, 309 bins, 43,844 samples) or the
// extended 44.1 kHz range of BASELINE.json configs[4] (10 levels, hop 512, 345 bins, 87,688 samples: SURVEY.md App. A.6).
struct FmGeo {
  int n_levels, hop0, n_bins;
  int64_t audio_stride, pyr_stride;
  int len[10];  // samples of level k
  int off[10];  // offset of level k inside a window's pyramid row (level 0 is the audio itself)
};

__global__ __launch_bounds__(kFmThreads, 4) void cqt_filterbank_mfma_kernel(
    const float* __restrict__ audio, const float* __restrict__ pyr, const uint4* __restrict__ bfrag,
    const float* __restrict__ sqrt_len, float* __restrict__ lp, float2* __restrict__ mmp, int n_windows,
    LogConsts kc, FmGeo geo) {
  __shared__ __attribute__((aligned(16))) uint4 s_hi[kFmMaxUnits];
  __shared__ __attribute__((aligned(16))) uint4 s_lo[kFmMaxUnits];
  __shared__ float exch[kFmExch];
  const int lane = threadIdx.x & 63;
  const int role = wave_id();

  uint4 bh[kFmSteps], bl[kFmSteps];
  {
    const uint4* bp_ = bfrag + (size_t)role * kFmSteps * 2 * 64 + lane;
#pragma unroll
    for (int s = 0; s < kFmSteps; ++s) {
      bh[s] = bp_[(2 * s) * 64];
      bl[s] = bp_[(2 * s + 1) * 64];
    }
  }
  for (int i = threadIdx.x; i < kFmExch; i += kFmThreads) exch[i] = 0.0f;  // layer 1's zero column stays zero
  const int per_window = geo.n_levels * kFmTilesPerLevel;
  const int n_items = n_windows * per_window;
  // (window, item inside the window) advance by the grid stride without a division by the runtime per_window
  int b = blockIdx.x / per_window, rem = blockIdx.x - b * per_window;
  const int db = gridDim.x / per_window, drem = gridDim.x - db * per_window;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x, b += db, rem += drem) {
    if (rem >= per_window) {
      rem -= per_window;
      ++b;
    }
    const int level = rem / kFmTilesPerLevel;
    const int tile = rem - level * kFmTilesPerLevel;
    const float* x = (level == 0) ? audio + (int64_t)b * geo.audio_stride
                                  : pyr + (int64_t)b * geo.pyr_stride + geo.off[level];
    const int L = geo.len[level];
#define BP_FM_CASE(HOP)                                                                                         \
  case HOP:                                                                                                     \
    fm_item<HOP>(x, L, b, level, geo.n_levels, geo.n_bins, tile, bh, bl, sqrt_len, lp, mmp, kc, s_hi, s_lo,      \
                 exch, role, lane);                                                                             \
    break;
    switch (geo.hop0 >> level) {
      BP_FM_CASE(512)
      BP_FM_CASE(256)
      BP_FM_CASE(128)
      BP_FM_CASE(64)
      BP_FM_CASE(32)
      BP_FM_CASE(16)
      BP_FM_CASE(8)
      BP_FM_CASE(4)
      BP_FM_CASE(2)
      default:
        fm_item<1>(x, L, b, level, geo.n_levels, geo.n_bins, tile, bh, bl, sqrt_len, lp, mmp, kc, s_hi, s_lo, exch,
                   role, lane);
        break;
    }
#undef BP_FM_CASE
  }
}

void launch_mm_reduce(const float* scratch, int* mm, int n_windows, int n_partials, hipStream_t stream);

FmGeo make_fm_geo(bool ext) {
  FmGeo g{};
  if (!ext) {
    g.n_levels = kOctaves;
    g.hop0 = 256;
    g.n_bins = kBins;
    g.audio_stride = kAudioN;
    g.pyr_stride = kPyrStride;
    for (int k = 0; k < kOctaves; ++k) {
      g.len[k] = level_len(k);
      g.off[k] = k ? pyr_off(k) : 0;
    }
  } else {  // level k >= 1 of the 44.1 kHz pyramid has the length of level k - 1 of the 22.05 kHz one
    g.n_levels = kOctavesExt;
    g.hop0 = 512;
    g.n_bins = kBinsExt;
    g.audio_stride = kAudioNExt;
    g.pyr_stride = kPyrStrideExt;
    g.len[0] = kAudioNExt;
    g.off[0] = 0;
    for (int k = 1; k < kOctavesExt; ++k) {
      g.len[k] = level_len(k - 1);
      g.off[k] = k == 1 ? 0 : kAudioN + pyr_off(k - 1);
    }
  }
  return g;
}

// workgroups of the decimator resident on the device at once (occupancy x CUs), queried once
static int dm_resident_workgroups() {
  static const int n = [] {
    int dev = 0, cus = 256, per_cu = 4;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decimate2_mfma_kernel, kDmThreads, 0);
    return cus * (per_cu > 0 ? per_cu : 1);
  }();
  return n;
}

void launch_pyramid_mfma(const float* audio, float* pyr, const void* hfrag, int n_windows, bool ext,
                         hipStream_t stream) {
  const FmGeo g = make_fm_geo(ext);
  // levels whose input already lives in the pyramid row and that are at most two blocks long go to the tail kernel
  int first_tail = g.n_levels;
  for (int k = g.n_levels - 1; k >= 2 && g.len[k] <= kDmTailMax; --k) first_tail = k;
  for (int k = 1; k < first_tail; ++k) {
    const float* src = (k == 1) ? audio : pyr + g.off[k - 1];
    const int64_t sstride = (k == 1) ? g.audio_stride : g.pyr_stride;
    const int lin = g.len[k - 1], lout = g.len[k];
    const int items = ((lout + kDmOutPerWg - 1) / kDmOutPerWg) * n_windows;
    const int grid = items < dm_resident_workgroups() ? items : dm_resident_workgroups();
    hipLaunchKernelGGL(decimate2_mfma_kernel, dim3(grid), dim3(kDmThreads), 0, stream, src, sstride, lin,
                       pyr + g.off[k], g.pyr_stride, lout, static_cast<const uint4*>(hfrag), n_windows);
  }
  if (first_tail < g.n_levels) {
    DmTail t{};
    t.first = first_tail;
    t.last = g.n_levels - 1;
    for (int k = 0; k < g.n_levels; ++k) {
      t.len[k] = g.len[k];
      t.off[k] = g.off[k];
    }
    hipLaunchKernelGGL(decimate2_tail_kernel, dim3(n_windows), dim3(kDmThreads), 0, stream, pyr, g.pyr_stride,
                       static_cast<const uint4*>(hfrag), t);
  }
}

void launch_filterbank_mfma(const float* audio, const float* pyr, const void* bfrag, const float* sqrt_len,
                            float* lp, int* mm, float* scratch, int n_windows, LogConsts kc, int n_cu, bool ext,
                            hipStream_t stream) {
  const FmGeo g = make_fm_geo(ext);
  float2* mmp = reinterpret_cast<float2*>(scratch);
  const int items = n_windows * g.n_levels * kFmTilesPerLevel;
  const int grid = items < 4 * n_cu ? items : 4 * n_cu;
  hipLaunchKernelGGL(cqt_filterbank_mfma_kernel, dim3(grid), dim3(kFmThreads), 0, stream, audio, pyr,
                     static_cast<const uint4*>(bfrag), sqrt_len, lp, mmp, n_windows, kc, g);
  // mm == null: the caller folds the partial extrema itself (launch_zpack_partials)
  if (mm) launch_mm_reduce(scratch, mm, n_windows, g.n_levels * kFmTilesPerLevel * 4, stream);
}

int filterbank_mfma_partials(bool ext) { return make_fm_geo(ext).n_levels * kFmTilesPerLevel * 4; }

}  // namespace bp
