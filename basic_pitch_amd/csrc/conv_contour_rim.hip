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
//   * a work item is (window, side, 64 frames) = 5 M blocks of 32 rows (4 bins x 8 channels) x 2 column tiles; a
//     workgroup is FOUR waves: wave w owns M block w for all 27 k-steps and a quarter of the k-steps of M block 4, whose
//     four partial sums meet in LDS (the image's space, read out by then).  Four, not five, because a workgroup's waves
//     go to the SIMDs round-robin: five waves put two on one SIMD, which then paces the CU with 2/5 of the work (measured:
//     62 % of the balanced rate), and at 3 - 4 wave slots per SIMD only one or two such workgroups fit at all.
//   * the A fragments (K, split hi | lo, 54 KB per M block) stream from L2, three k-steps ahead; B = 16 bytes of an f16
//     row image in LDS (hi plane, lo plane; row stride 37 units: conflict-free for the ds_read_b128 lane groups, one frame
//     per lane), read ahead of the matrix instructions and refilled in place.
//   * epilogue from the accumulator layout: a lane holds 4 consecutive channels of 4 bins of its frame -> bias, ReLU,
//     four 16-byte stores into c1 (the same [172][268][8] tensor the folded kernel fills the interior of).
// Split-precision products hi*hi + (lo*hi + hi*lo) * 2^-11 with fp32 accumulation as everywhere (bp_common.h).
// Roofline: f16 MFMA issue.  Algorithmic work 103 MFLOP per window (the reference's 8-channel products for 40 of 264
// bins); executed 4860 x 3 MFMAs; bytes per window: 2 x 172 x 576 B of zp read, 220 KB of c1 written, 1.6 MB of A
// fragments from L2.
#include "bp_common.h"

namespace bp {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int kRimBlocks = 5;             // M blocks of 32 rows: 20 bins x 8 channels per side
constexpr int kRimWaves = 4;              // wave w: block w, and k-steps [ks(w), ks(w + 1)) of block 4
#ifndef BP_RIM_FRAMES
#define BP_RIM_FRAMES 64
#endif
constexpr int kRimFrames = BP_RIM_FRAMES;  // frames per work item: kRimNT 32-column tiles
constexpr int kRimNT = kRimFrames / 32;
constexpr int kRimTiles = (kFrames + kRimFrames - 1) / kRimFrames;  // 3
constexpr int kRimRows = kRimFrames + 2;
constexpr int kRimPf = 3;                 // k-steps of A prefetch (L2 latency under load ~ 4 k-steps of 6 MFMAs)
__host__ __device__ constexpr int rim_f0(int side) { return side ? 244 : 0; }
// z bins per frame a side reads (a multiple of 16).  144 for the model's 309-bin CQT: low rim bins [0, 144) (141 used), high
// rim [184, 328) (125 used; bins >= 309 are zp's zero padding).  160 for the extended 345-bin CQT of the 44.1 kHz mode
// (round 4): there bins 309 .. 344 carry data and the high rim (stack bins up to 263 + shift 101) reads z bins up to 343.
template <int BINS>
struct RimGeo {
  static_assert(BINS == 144 || BINS == 160, "rim window");
  static constexpr int kBins = BINS;
  static constexpr int kUnitsRow = BINS / 8;        // 16-byte units per plane and row
  static constexpr int kStepsDt = BINS / 16;        // k-steps per frame tap: 9 / 10
  static constexpr int kSteps = 3 * kStepsDt;       // 27 / 30
  static constexpr int kRowU = 2 * kUnitsRow + 1;   // LDS row: hi units | lo units | 1 pad (odd: conflict-free): 37 / 41
  // block 4's k-steps dealt to the four waves: 7 + 7 + 7 + 6 / 8 + 8 + 7 + 7
  // first z bin of a side's window: the high rim reads z bins 189 .. min(364, n_bins - 1): [184, 328) holds them for 309 bins,
  // [188, 348) for 345
  static __host__ __device__ constexpr int j0(int side) { return side ? (BINS == 144 ? 184 : 188) : 0; }
  static __host__ __device__ constexpr int ks(int w) { return w >= 4 ? kSteps : (BINS == 144 ? 7 * w : (w <= 2 ? 8 * w : 23)); }
};

struct RimParams {
  const uint32_t* zp;   // [n][kZRowsP][kZRow]
  const uint4* afrag;   // [side 2][mb 5][step 27 | 30][hi|lo][64 lanes] x (8 x f16)   (bp_api.hip pack_contour_rim)
  const float* bias;    // [8]
  float* c1;            // [n][172][kC1Row][8]
  int n_items;          // n_windows * 2 * kRimTiles
};

// k-steps [S0, S1) of one M block into acc / accc.  The first kRimPf A fragments are already in the ring (rim_a_prefetch).
// B fragments are read ahead of the matrix instructions that use them (the compiler's own order reads them right in front
// of their use and waits: the k loop then took 24 k cycles for 5 k of matrix work) and refilled IN PLACE: a step issues
// [A_hi x B_hi], [A_lo x B_hi], then B_hi's registers take the next step's reads while [A_hi x B_lo] runs, then B_lo's.
// One register set (16) instead of two keeps the kernel under 128 VGPRs = 4 waves per SIMD.
template <int S0>
__device__ __forceinline__ void rim_a_prefetch(const uint4* afr, uint4 (&ah)[kRimPf], uint4 (&al)[kRimPf]) {
#pragma unroll
  for (int s = S0; s < S0 + kRimPf; ++s) {
    ah[s % kRimPf] = afr[(s * 2 + 0) * 64];
    al[s % kRimPf] = afr[(s * 2 + 1) * 64];
  }
}

template <class Geo, bool WLO, int S0, int S1>
__device__ __forceinline__ void rim_ksteps(const uint4* afr, const uint4* img, int lane_u, uint4 (&ah)[kRimPf],
                                           uint4 (&al)[kRimPf], f32x16 (&acc)[kRimNT], f32x16 (&accc)[kRimNT]) {
  static_assert(S1 - S0 >= kRimPf, "the ring is full at entry");
  f16x8 bh[kRimNT], bl[kRimNT];
  auto read_bh = [&](int s) {
    const int dt = s / Geo::kStepsDt, e = s - dt * Geo::kStepsDt;
#pragma unroll
    for (int j = 0; j < kRimNT; ++j) bh[j] = __builtin_bit_cast(f16x8, img[lane_u + (32 * j + dt) * Geo::kRowU + 2 * e]);
  };
  auto read_bl = [&](int s) {
    const int dt = s / Geo::kStepsDt, e = s - dt * Geo::kStepsDt;
#pragma unroll
    for (int j = 0; j < kRimNT; ++j)
      bl[j] = __builtin_bit_cast(f16x8, img[lane_u + (32 * j + dt) * Geo::kRowU + 2 * e + Geo::kUnitsRow]);
  };
  read_bh(S0);
  read_bl(S0);
#pragma unroll
  for (int s = S0; s < S1; ++s) {
    const uint4 a_hi = ah[s % kRimPf], a_lo = al[s % kRimPf];
    const f16x8 fah = __builtin_bit_cast(f16x8, a_hi);
    const f16x8 fal = __builtin_bit_cast(f16x8, a_lo);
    __builtin_amdgcn_sched_barrier(0);
    // the column tiles alternate so that no MFMA waits for the one just issued
#pragma unroll
    for (int j = 0; j < kRimNT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, bh[j], acc[j], 0, 0, 0);
    if (WLO) {
#pragma unroll
      for (int j = 0; j < kRimNT; ++j) accc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal, bh[j], accc[j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 < S1) read_bh(s + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < kRimNT; ++j) accc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, bl[j], accc[j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 < S1) read_bl(s + 1);
    if (s + kRimPf < S1) {
      ah[s % kRimPf] = afr[((s + kRimPf) * 2 + 0) * 64];
      al[s % kRimPf] = afr[((s + kRimPf) * 2 + 1) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <class Geo, bool WLO>
__global__ __launch_bounds__(64 * kRimWaves, kRimNT <= 2 ? 4 : 2) void contour_conv1_rim_kernel(RimParams p) {
  constexpr int kRimBins = Geo::kBins, kRimRowU = Geo::kRowU, kRimSteps = Geo::kSteps;
  __shared__ __attribute__((aligned(16))) uint4 img[kRimRows * kRimRowU];
  static_assert(sizeof(uint4) * kRimRows * kRimRowU >= sizeof(float) * kRimWaves * 16 * kRimNT * 64, "the partial sums fit the image");

  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int kh = lane >> 5, n = lane & 31;

#if defined(RIM_PROF)
  unsigned long long pt[6], pc = __builtin_amdgcn_s_memtime();
  const unsigned long long r_start = __builtin_amdgcn_s_memrealtime();
#define RIM_STAMP(k)                                             \
  {                                                              \
    const unsigned long long now = __builtin_amdgcn_s_memtime(); \
    pt[k] = now - pc;                                            \
    pc = now;                                                    \
  }
#else
#define RIM_STAMP(k)
#endif
  const int item = blockIdx.x;
  const int b = item / (2 * kRimTiles);
  const int rem = item - b * (2 * kRimTiles);
  const int side = rem / kRimTiles, tile = rem - side * kRimTiles;
  const int t0 = tile * kRimFrames;


  // ---- stage frames t0 - 1 .. t0 + 64 of the side's 144 z bins as f16 planes: unit (row, u) = bins j0 + 8u .. + 7.
  // All loads of a thread are in flight before the first is used (one global round trip per item, not five).
  {
    const uint32_t* zwin = p.zp + (int64_t)b * kZWin + kZPadL + Geo::j0(side);
    constexpr int kUnits = kRimRows * (kRimBins / 8), kPerThread = (kUnits + 64 * kRimWaves - 1) / (64 * kRimWaves);
    uint4 w0[kPerThread], w1[kPerThread];
#pragma unroll
    for (int i = 0; i < kPerThread; ++i) {
      const int e = threadIdx.x + i * 64 * kRimWaves;
      const int row = e / (kRimBins / 8), u = e - row * (kRimBins / 8);
      const int t = t0 - 1 + row;  // zp rows -1 and 172 are zero; frames beyond them read the all-zero row -1:
      const bool ok = e < kUnits && t >= -1 && t <= kFrames;  // no select behind the load
      const uint4* src = reinterpret_cast<const uint4*>(zwin + (int64_t)((ok ? t : -1) + 1) * kZRow + 8 * (ok ? u : 0));
      w0[i] = src[0], w1[i] = src[1];
    }
#pragma unroll
    for (int i = 0; i < kPerThread; ++i) {
      const int e = threadIdx.x + i * 64 * kRimWaves;
      if (e >= kUnits) break;
      const int row = e / (kRimBins / 8), u = e - row * (kRimBins / 8);
      uint4 vh, vl;
      vh.x = (w0[i].x & 0xffffu) | (w0[i].y << 16);
      vh.y = (w0[i].z & 0xffffu) | (w0[i].w << 16);
      vh.z = (w1[i].x & 0xffffu) | (w1[i].y << 16);
      vh.w = (w1[i].z & 0xffffu) | (w1[i].w << 16);
      vl.x = (w0[i].x >> 16) | (w0[i].y & 0xffff0000u);
      vl.y = (w0[i].z >> 16) | (w0[i].w & 0xffff0000u);
      vl.z = (w1[i].x >> 16) | (w1[i].y & 0xffff0000u);
      vl.w = (w1[i].z >> 16) | (w1[i].w & 0xffff0000u);
      img[row * kRimRowU + u] = vh;
      img[row * kRimRowU + Geo::kUnitsRow + u] = vl;
    }
  }
  // ---- A fragments of this wave's own M block: on their way (L2) across the barrier
  const uint4* afr = p.afrag + ((int64_t)(side * kRimBlocks + wave) * kRimSteps) * 2 * 64 + lane;
  uint4 ah[kRimPf], al[kRimPf];
  rim_a_prefetch<0>(afr, ah, al);
  RIM_STAMP(0);
  lds_barrier();
  RIM_STAMP(1);

  f32x16 acc[kRimNT], accc[kRimNT];
  auto clear = [&]() {
#pragma unroll
    for (int j = 0; j < kRimNT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = accc[j][r] = 0.0f;
  };
  // lane (n, kh) of column tile j reads image row (32 j + n + dt), unit 2 e + kh of the hi / lo plane
  const int lane_u = n * kRimRowU + kh;

  // ---- phase 1: the wave's own M block, all 27 k-steps
  clear();
  rim_ksteps<Geo, WLO, 0, kRimSteps>(afr, img, lane_u, ah, al, acc, accc);

#if defined(RIM_PROF)
  asm volatile("" : "+v"(acc[0][0]), "+v"(accc[kRimNT - 1][15]));
#endif
  RIM_STAMP(2);
  // ---- epilogue of a full block: C row i = (r & 3) + 8 (r >> 2) + 4 kh = 8 (bin of the block) + channel -> channels
  // 4 kh .. 4 kh + 3 of bin r >> 2; column n = frame.  (Sending the block's 32 frames x 128 contiguous bytes through a
  // wave-private LDS tile so that they leave as whole rows — 8 frames per store instruction instead of 32 partial lines —
  // measured the same 51 us: the stores are not what paces this kernel.)
  const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias + 4 * kh);
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

  // ---- phase 2: a quarter of M block 4's k-steps; the four partial sums meet in LDS
  RIM_STAMP(3);
  const uint4* afr4 = p.afrag + ((int64_t)(side * kRimBlocks + 4) * kRimSteps) * 2 * 64 + lane;
  clear();
#define RIM_QUARTER(w)                                                                              \
  case w:                                                                                           \
    rim_a_prefetch<Geo::ks(w)>(afr4, ah, al);                                                       \
    rim_ksteps<Geo, WLO, Geo::ks(w), Geo::ks(w + 1)>(afr4, img, lane_u, ah, al, acc, accc);         \
    break;
  switch (wave) {
    RIM_QUARTER(0)
    RIM_QUARTER(1)
    RIM_QUARTER(2)
    RIM_QUARTER(3)
  }
#undef RIM_QUARTER
#if defined(RIM_PROF)
  asm volatile("" : "+v"(acc[0][0]), "+v"(accc[kRimNT - 1][15]));
#endif
  RIM_STAMP(4);
  lds_barrier();  // every wave is done with the image: its space takes the partial sums [wave][16 NT values][64 lanes]
  float* part = reinterpret_cast<float*>(img);
  constexpr int kVals = 16 * kRimNT;
#pragma unroll
  for (int j = 0; j < kRimNT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      part[(wave * kVals + 16 * j + r) * 64 + lane] = __builtin_fmaf(accc[j][r], kLoUnscale, acc[j][r]);
  lds_barrier();
  // the 4 NT (column tile, bin) pairs of the block are dealt to the waves: wave w finishes pairs w NT .. w NT + NT - 1, each
  // the four channels 4 kh .. 4 kh + 3 of one bin = one 16-byte store
#pragma unroll
  for (int i = 0; i < kRimNT; ++i) {
    const int sidx = wave * kRimNT + i, j = sidx >> 2, fb = sidx & 3;
    const int t = t0 + 32 * j + n;
    f32x4 v;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float sum = 0.0f;
#pragma unroll
      for (int w = 0; w < kRimWaves; ++w) sum += part[(w * kVals + 16 * j + 4 * fb + c) * 64 + lane];
      v[c] = fmaxf(sum + bias4[c], 0.0f);
    }
    if (t < kFrames)
      *reinterpret_cast<f32x4*>(p.c1 + (((int64_t)b * kFrames + t) * kC1Row + kC1Pad + rim_f0(side) + 16 + fb) * 8 + 4 * kh) = v;
  }
  RIM_STAMP(5);
#if defined(RIM_PROF)
  if (lane == 0 && (blockIdx.x % 61 == 0 || blockIdx.x < 4) && p.n_items > 1000)
    printf("RIMQ wg %d wave %d real %llu stage %llu bar %llu k1 %llu epi %llu k2 %llu red %llu\n", (int)blockIdx.x, wave, r_start,
           pt[0], pt[1], pt[2], pt[3], pt[4], pt[5]);
#endif
#undef RIM_STAMP
}

void launch_contour_conv1_rim(const uint32_t* zp, const void* afrag, const float* bias, float* c1, int n_windows, int n_cu,
                              bool weights_have_lo, bool ext, hipStream_t stream) {
  (void)n_cu;
  RimParams p{zp, static_cast<const uint4*>(afrag), bias, c1, n_windows * 2 * kRimTiles};
  if (p.n_items <= 0) return;
  const dim3 grid(p.n_items), block(64 * kRimWaves);
  if (ext) {  // the 345-bin CQT of the 44.1 kHz mode: 160 z bins per side (afrag packed for them)
    if (weights_have_lo)
      hipLaunchKernelGGL((contour_conv1_rim_kernel<RimGeo<160>, true>), grid, block, 0, stream, p);
    else
      hipLaunchKernelGGL((contour_conv1_rim_kernel<RimGeo<160>, false>), grid, block, 0, stream, p);
  } else {
    if (weights_have_lo)
      hipLaunchKernelGGL((contour_conv1_rim_kernel<RimGeo<144>, true>), grid, block, 0, stream, p);
    else
      hipLaunchKernelGGL((contour_conv1_rim_kernel<RimGeo<144>, false>), grid, block, 0, stream, p);
  }
}

}  // namespace bp
