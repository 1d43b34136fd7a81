// A/B kernels of the contour conv1 — compiled only into builds with -DBP_AB_KERNELS (basic_pitch_amd/build.py
// build_library(ab=True): the comparison tests and tools); the product library does not carry them.  The default path is
// conv_contour_march.hip (interior) + conv_contour_rim.hip (rim) + conv_contour2.hip (conv2).  Here: the exact
// 8-channel conv1 (BP_RIM=exact for the rim, BP_CONV1=full for every group) and the round-2 folded conv1 (BP_CONV1=rounds).
//
//   contour_conv1_kernel   Conv2D 8->8, (3 frames x 39 bins), "same", folded BN, ReLU on the harmonic stack
//                          (basic_pitch/models.py:241-250, nn.py:69-88)                          zp -> c1
//
// Why not one fused kernel (conv_contour.hip): 65 % of the whole path's FLOPs are conv1, and the fused
// kernel's matrix pipe idles two thirds of the time — the register-resident weights force a K split over 4
// waves, so every 32-position tile pays a cross-wave reduction, two barriers and a serial conv2 epilogue
// (measured: 2.2 k cycles of MFMA phase + 4.5 k cycles of latency-bound epilogue per tile; neither more
// prefetch nor software pipelining across tiles nor de-phasing the two resident workgroups moved it, see
// DESIGN.md §7).  Here conv1 is a pure matrix kernel and the tiny conv2 runs at the HBM rate behind it.
//
// conv1 mapping (v_mfma_f32_32x32x16_f16, split-precision operands, bp_common.h):
//   C[(bin offset j, out channel o) (32 rows)][position (32 cols)] = Wt[(j,o)][k] x S[k][position]
//   * position = (frame, group of 4 adjacent bins); rows carry a 4-bin Toeplitz expansion of the 39-tap
//     kernel; one k-step = 2 adjacent taps x 8 stack channels; K = 3 frames x 21 tap pairs = 63 k-steps.
//   * B (the stack image) comes from an LDS ring of image rows exactly as in the fused kernel: channel-last
//     16-byte slots, f16 hi | scaled lo, 4 phase planes so that the 32 lanes of a read are consecutive slots.
//   * A (the weights) ALSO comes from LDS, un-expanded: slot (dt, tap + 3, o) holds the 8 input channels of
//     W1[o][:][dt][tap]; the Toeplitz expansion is pure addressing (lane (j, o, half) reads tap 2 ep + half - j),
//     zero taps are materialised.  34.6 KB instead of 126 KB of fragments, conflict-free, and NO K split:
//     every wave owns complete sums of its positions — no reduction, no barrier per tile.
//   * one workgroup = 4 waves = one wave per SIMD with the whole register file: each wave walks 2 tiles
//     (64 positions) through all 63 k-steps with operand reads issued 2 k-steps ahead (6 ds_read_b128 feed
//     6 MFMAs: LDS 50 % busy at full matrix rate), then adds bias, applies ReLU and stores c1 straight from
//     the accumulator layout (a lane holds 4 consecutive channels of a pixel = one 16-byte store).
//   * a round (256 positions = 3.9 image rows) ends with the only barrier; the image rows of the NEXT round
//     are gathered from zp (8 harmonic shifts, no masks: zp carries its own zero padding) and written to the
//     ring between the MFMAs of the current round.
//
// Roofline: conv1 f16 MFMA issue — 680.0 MFLOP per window algorithmic, 3 f16 MFMAs per product (hi*hi,
// lo*hi, hi*lo) and 42/39 Toeplitz padding executed; bytes per window: 311,808 (zp) read, 1,475,072 (c1)
// written.  conv2 HBM: 1.48 MB read, 181,632 B written per window, 18.2 MFLOP on the f32 VALU.
#include <stdlib.h>
#include <string.h>

#include "bp_common.h"

namespace bp {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int kD1WTap = 45;                        // tap + 3 in [0, 45): 3 zero taps below, 3 above
constexpr int kD1WHalf = 3 * kD1WTap * 8;          // 1080 slots of hi weights, then 1080 of lo
constexpr int kD1Steps = 63;
constexpr int kD1Pf = 3;                           // k-steps of operand prefetch (12 reads in flight)
constexpr int kD1GroupsRow = kFreqC / 4;           // 66 four-bin groups per frame
constexpr int kD1EdgeGroups = 5;                   // groups 0..4 and 61..65 see the crop of the stack (bins < 20, >= 244)

// Geometry of the exact 8-channel kernel.
//   FullGeo: every group of a frame (A/B reference of the folded form, BP_CONV1=full).
//   EdgeGeo: only the 2 x 5 groups at the rim of the 264-bin stack, where "crop, then pad" (nn.py:87) makes the
//            folded kernel position dependent; the image keeps two 60-bin windows per row instead of 304 bins.
struct FullGeo {
  static constexpr int kWaves = 8, kGroups = 66, kQ = 76, kRing = 12, kStage = 3, kStageEvery = 20, kStageLag = 15;
  static constexpr int kRowU4 = 4 * kQ;                                            // row stride of the image, 16-byte units
  static __device__ __forceinline__ int group_of(int gi) { return gi; }
  static __device__ __forceinline__ int q_of(int q) { return q; }                 // plane index of bin slot q
  static __device__ __forceinline__ int n_bins() { return kFreqC; }               // bins staged per row
  static __device__ __forceinline__ int bin_of(int i) { return i; }
  // lane li of wave w in round k -> (frame offset, group index inside the frame): 32 consecutive (frame, group) pairs
  static __device__ __forceinline__ bool locate(int k, int w, int li, int n_frames, int& frame, int& gi) {
    const int pos = kWaves * 32 * k + 32 * w + li, npos = n_frames * kGroups;
    const int posc = pos < npos ? pos : npos - 1;
    frame = posc / kGroups;
    gi = posc - frame * kGroups;
    return pos < npos;
  }
};
// The rim kernel's image: per row 4 phase planes of kQ = 31 units — the low rim's bin slots at 0..14, the high rim's at
// 16..30 — and a row stride of 133 units.  Its lane mapping and these numbers come from enumerating the bank columns
// (unit index mod 16) of every ds_read_b128 service group ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32): lanes
// 0..15 of a tile take 16 consecutive (frame, group) pairs of the LOW rim, lanes 16..31 the same pairs of the HIGH rim;
// with 5 groups per frame and stride = 5 mod 16 the low lanes' columns are their lane numbers, the high lanes' are
// lane + 16: 0.4 extra LDS cycles per group instead of 6.4 for "32 consecutive pairs of a 10-group frame, stride 120".
struct EdgeGeo {
  static constexpr int kWaves = 4, kGroups = 10, kQ = 31, kRing = 30, kStage = 5, kStageEvery = 12, kStageLag = 9;
  static constexpr int kRowU4 = 133;
  static __device__ __forceinline__ int group_of(int gi) { return gi < kD1EdgeGroups ? gi : gi + 56; }
  static __device__ __forceinline__ int q_of(int q) { return q < 15 ? q : q - 45; }  // [0,15) U [61,76) -> [0,15) U [16,31)
  static __device__ __forceinline__ int n_bins() { return 80; }                   // bins [0,40) and [224,264)
  static __device__ __forceinline__ int bin_of(int i) { return i < 40 ? i : i + 184; }
  static __device__ __forceinline__ bool locate(int k, int w, int li, int n_frames, int& frame, int& gi) {
    const int side = li >> 4;
    const int sp = 16 * kWaves * k + 16 * w + (li & 15), nside = n_frames * kD1EdgeGroups;
    const int spc = sp < nside ? sp : nside - 1;
    frame = spc / kD1EdgeGroups;
    gi = spc - frame * kD1EdgeGroups + kD1EdgeGroups * side;
    return sp < nside;
  }
};

struct Conv1Params {
  const uint32_t* zp;   // [n][kZRowsP][kZRow] padded pre-split z (zpack_kernel)
  const uint4* wlds;    // exact: [hi|lo][3][45][8] x (8 x f16) weight image; folded: [3][12][hi|lo][64] fragments
  const float* bias;    // [8]
  float* c1;            // [n][172][kC1Row][8] relu(conv1), 2 zero bins of padding either side of a row
  int n_windows;
  int chunks;           // row chunks per window (work items = n_windows * chunks)
};

__device__ __forceinline__ void d1_split_words(const uint32_t (&u)[8], uint4& vh, uint4& vl) {
  vh.x = (u[0] & 0xffffu) | (u[1] << 16);
  vh.y = (u[2] & 0xffffu) | (u[3] << 16);
  vh.z = (u[4] & 0xffffu) | (u[5] << 16);
  vh.w = (u[6] & 0xffffu) | (u[7] << 16);
  vl.x = (u[0] >> 16) | (u[1] & 0xffff0000u);
  vl.y = (u[2] >> 16) | (u[3] & 0xffff0000u);
  vl.z = (u[4] >> 16) | (u[5] & 0xffff0000u);
  vl.w = (u[6] >> 16) | (u[7] & 0xffff0000u);
}

// gather the 8 harmonic-stack channels of bin f of image row `row` (zp is zero outside the CQT: no masks)
__device__ __forceinline__ void d1_gather(const uint32_t* __restrict__ zorigin, int row, int f,
                                          uint32_t (&u)[8]) {
  const uint32_t* zr = zorigin + row * kZRow + f;
#pragma unroll
  for (int c = 0; c < 8; ++c) u[c] = zr[harm_shift(c)];
}

// bias + ReLU and the c1 store: register r of a lane is (bin offset j = r >> 2, channel o = 4 kh + (r & 3))
__device__ __forceinline__ void d1_store(float* __restrict__ dst, const f32x16& hh, const f32x16& xx,
                                         const float (&bias4)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float4 v;
    v.x = fmaxf((hh[4 * j + 0] + xx[4 * j + 0] * kLoUnscale) + bias4[0], 0.0f);
    v.y = fmaxf((hh[4 * j + 1] + xx[4 * j + 1] * kLoUnscale) + bias4[1], 0.0f);
    v.z = fmaxf((hh[4 * j + 2] + xx[4 * j + 2] * kLoUnscale) + bias4[2], 0.0f);
    v.w = fmaxf((hh[4 * j + 3] + xx[4 * j + 3] * kLoUnscale) + bias4[3], 0.0f);
    *reinterpret_cast<float4*>(dst + j * 8) = v;
  }
}

// WLO = false: the weights have no lo part (BP_FLAG_BF16_WEIGHTS): no lo fragment reads, 2 MFMAs per k-step
template <class Geo, bool WLO>
__global__ __launch_bounds__(Geo::kWaves * 64, Geo::kWaves / 4) void contour_conv1_kernel(Conv1Params p) {
  constexpr int kThreads = Geo::kWaves * 64;
  constexpr int kQ = Geo::kQ, kSlots = Geo::kRowU4, kRing = Geo::kRing;
  static_assert(kSlots >= 4 * kQ, "a row holds the four phase planes");
  constexpr int kLoOff = kRing * kSlots;          // img[] = hi image, then lo image (uint4 units)
  constexpr int kRound = Geo::kWaves * 32;        // positions per round
  constexpr int kGroups = Geo::kGroups;
  static_assert(2 * kLoOff * 16 + 2 * kD1WHalf * 16 <= 160 * 1024, "LDS budget");
  static_assert(kLoOff * 16 + (3 * kQ + 12) * 16 < 65536, "ds_read immediate offset of the lo image");
  __shared__ __attribute__((aligned(16))) uint4 img[2 * kLoOff];
  __shared__ __attribute__((aligned(16))) uint4 wl[2 * kD1WHalf];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = wave_id();
  const int kh = lane >> 5, li = lane & 31;

  for (int i = tid; i < 2 * kD1WHalf; i += kThreads) wl[i] = p.wlds[i];
  // every slot starts as zero; the slots of bins outside the cropped stack (nn.py:87: crop to 264 bins, then "same"
  // padding) are never written again
  for (int i = tid; i < 2 * kLoOff; i += kThreads) img[i] = uint4{0u, 0u, 0u, 0u};
  float bias4[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bias4[q] = p.bias[4 * kh + q];
  // A operand: lane (row i = 8 j + o, half kh) reads weight slot (dt, 2 ep + kh - j + 3, o)
  const int aidx = (kh - (li >> 3) + 3) * 8 + (li & 7);
  // image slot of stack bin f: plane (f + 20) & 3, index q_of((f + 20) >> 2)
  auto slot_of_bin = [](int f) { return ((f + 20) & 3) * kQ + Geo::q_of((f + 20) >> 2); };

  const int rows_per = (kFrames + p.chunks - 1) / p.chunks;
  const int n_items = p.n_windows * p.chunks;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int b = item / p.chunks;
    const int t0 = (item - b * p.chunks) * rows_per;
    const int t1 = t0 + rows_per < kFrames ? t0 + rows_per : kFrames;
    const int npos = (t1 - t0) * kGroups;
    const int nrounds = (npos + kRound - 1) / kRound;
    const uint32_t* zorigin = p.zp + (int64_t)b * kZWin + kZRow + kZPadL;  // (frame 0, bin 0)
    float* c1b = p.c1 + (int64_t)b * kC1Win;
    const int nb = Geo::n_bins();

    lds_barrier();  // the previous item is done with the ring
    // image rows t0 - 1 .. staged_hi of round 0
    int staged_hi = t0 + (kRound - 1) / kGroups + 1;
    staged_hi = staged_hi < t1 ? staged_hi : t1;
    for (int e = tid; e < (staged_hi - t0 + 2) * nb; e += kThreads) {
      const int ri = e / nb, f = Geo::bin_of(e - ri * nb);
      const int row = t0 - 1 + ri;
      uint32_t u[8];
      d1_gather(zorigin, row, f, u);
      uint4 vh, vl;
      d1_split_words(u, vh, vl);
      const int idx = ((row + kRing) % kRing) * kSlots + slot_of_bin(f);
      img[idx] = vh;
      img[idx + kLoOff] = vl;
    }
    lds_barrier();

    for (int k = 0; k < nrounds; ++k) {
      // ---- image rows to bring in during this round: (staged_hi, need_hi]
      int need_hi = t0 + (kRound * (k + 1) + kRound - 1) / kGroups + 1;
      need_hi = need_hi < t1 ? need_hi : t1;
      const int n_new = (k + 1 < nrounds) ? need_hi - staged_hi : 0;
      const int first_new = staged_hi + 1;
      uint32_t pf[8];
      int put_idx = 0;
      bool put_ok = false;
      auto stage_issue = [&](int i) {
        const int e = i * kThreads + tid;
        put_ok = e < n_new * nb;
        if (put_ok) {
          const int ri = e / nb, f = Geo::bin_of(e - ri * nb);
          const int row = first_new + ri;
          d1_gather(zorigin, row, f, pf);
          put_idx = ((row + kRing) % kRing) * kSlots + slot_of_bin(f);
        }
      };
      auto stage_put = [&]() {
        if (put_ok) {
          uint4 vh, vl;
          d1_split_words(pf, vh, vl);
          img[put_idx] = vh;
          img[put_idx + kLoOff] = vl;
        }
      };

      // ---- this wave's tile: 32 consecutive positions of the item's (frame, group) list
      int prr, pgi;
      const bool pvalid = Geo::locate(k, w, li, t1 - t0, prr, pgi);
      const int pgrp = Geo::group_of(pgi);                    // four-bin group of the frame
      const int pmf = Geo::q_of(pgrp);                        // its index inside a phase plane
      const int prow = t0 + prr;
      int lo_base[3], hi_base[3];
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) {
        const int rowslot = ((prow - 1 + dt + kRing) % kRing) * kSlots;
        lo_base[dt] = rowslot + pmf + kh * kQ;              // tap plane 1 -> 2 (same group)
        hi_base[dt] = rowslot + pmf + kh * (1 - 3 * kQ);    // tap plane 3 -> 0 of the next group
      }
      f32x16 hh, xx;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        hh[r] = 0.0f;
        xx[r] = 0.0f;
      }
      f16x8 ah[kD1Steps], al[kD1Steps], bh[kD1Steps], bl[kD1Steps];
      auto issue = [&](int s) {
        const int dt = s / 21, ep = s - 21 * dt;
        const int r0 = (2 * ep + 1) & 3, q0 = (2 * ep + 1) >> 2;
        const int widx = aidx + (dt * kD1WTap + 2 * ep) * 8;
        const int sb = ((r0 == 1) ? lo_base[dt] : hi_base[dt]) + r0 * kQ + q0;
        if (WLO) al[s] = __builtin_bit_cast(f16x8, wl[widx + kD1WHalf]);
        bh[s] = __builtin_bit_cast(f16x8, img[sb]);
        ah[s] = __builtin_bit_cast(f16x8, wl[widx]);
        bl[s] = __builtin_bit_cast(f16x8, img[sb + kLoOff]);
      };
#pragma unroll
      for (int s = 0; s < kD1Pf; ++s) issue(s);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < kD1Steps; ++s) {
        if (s + kD1Pf < kD1Steps) issue(s + kD1Pf);
        // staging of the next round's rows, spread over the k-steps
        if (s % Geo::kStageEvery == 1 && s / Geo::kStageEvery < Geo::kStage) stage_issue(s / Geo::kStageEvery);
        if (s % Geo::kStageEvery == 1 + Geo::kStageLag && s / Geo::kStageEvery < Geo::kStage) stage_put();
        __builtin_amdgcn_sched_barrier(0);
        if (WLO) xx = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh[s], xx, 0, 0, 0);
        hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[s], hh, 0, 0, 0);
        xx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl[s], xx, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      staged_hi += n_new;
      if (pvalid) d1_store(c1b + ((int64_t)prow * kC1Row + kC1Pad + 4 * pgrp) * 8 + 4 * kh, hh, xx, bias4);
      lds_barrier();  // the round's reads are done; the rows written for the next round are visible
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Folded conv1 for the interior of the stack (groups 5..60 = bins 20..243).
//
// The 8 stack channels are shifted copies of ONE image (z at bin f + s_c, s = -36, 0, 36, 57, 72, 84, 93, 101:
// nn.py:51-54,73-85), so away from the crop the 8 x 39-tap kernels of an output channel collapse into a single
// kernel over z:   K[o][dt][g] = sum_c W1[o][c][dt][g - s_c + 19],   g in [-55, 120]   (176 taps: the eight 39-tap
// intervals overlap or abut).  K = 3 x 176 = 528 instead of 3 x 312 = 936 products per output: 36 k-steps of 16
// taps (179 with the 4-bin Toeplitz expansion, 192 padded) instead of 63.  The same MFMA mapping otherwise:
//   C[(j, o)][position] = Kt[(j, o)][tap'] x Z[tap'][position],   Z[tap'][m] = z[4 m + tap' - 56]
//   * B: 8 consecutive taps = 8 consecutive z bins = 16 bytes of the f16 image row; the lane stride is 4 bins = 8
//     bytes, so each row is kept twice (the second copy shifted by 4 bins) and odd groups read the shifted copy:
//     every read is an aligned ds_read_b128;
//   * A: Toeplitz-expanded fragments straight from LDS ([dt][k-step][hi|lo][lane], 73.7 KB), packed on the host.
// At the rim (bins < 20 or >= 244) "crop to 264 bins, then zero-pad" removes a different set of taps for every
// output bin: those 2 x 5 groups per frame stay with the exact kernel (EdgeGeo) above.
constexpr int kF1Threads = 512;
constexpr int kF1Steps = 36;                       // 3 frames x 12 k-steps of 16 taps
constexpr int kF1Groups = 56;                      // groups 5..60
// LDS layout of a z row: hi copy 0, hi copy 1 (shifted 4 bins), lo copy 0, lo copy 1, kF1Copy 16-byte units each.
// A ds_read_b128 is serviced in four groups of 16 lanes — {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same
// + 32 (MI355X_MICROARCH.md, LDS) — and a group is conflict-free when its 16 units fall into 16 distinct 16-byte bank
// columns (unit index mod 16).  A wave's 32 lanes are consecutive (frame, group) positions, even groups reading copy 0
// at unit m / 2 and odd groups copy 1 at kF1Copy + (m - 1) / 2, wrapping to the next frame (+ row stride - 28 units)
// after group 60.  Enumerating every tile start: copies 73 units apart and a row stride of 4 x 73 = 292 give ZERO
// conflicts (the first layout, 72 / 288, cost 4.3 extra LDS cycles per group: SQ_LDS_BANK_CONFLICT 3.2e7 per launch);
// the ring is 16 rows so that its own wrap (16 strides) keeps the columns too.
constexpr int kF1Ring = 16;                        // z rows resident
constexpr int kF1Copy = 73;                        // uint4 per row copy (464 f16 used)
constexpr int kF1RowU4 = 4 * kF1Copy;
constexpr int kF1Round = 256;
constexpr int kF1Pf = 3;
static_assert(kF1Ring * kF1RowU4 * 16 + 3 * 12 * 2 * 64 * 16 <= 160 * 1024, "LDS budget");

template <bool WLO>
__global__ __launch_bounds__(kF1Threads, 2) void contour_conv1_folded_kernel(Conv1Params p) {
  __shared__ __attribute__((aligned(16))) uint4 zimg[kF1Ring * kF1RowU4];
  __shared__ __attribute__((aligned(16))) uint4 afr[kF1Steps * 2 * 64];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = wave_id();
  const int kh = lane >> 5, li = lane & 31;

  for (int i = tid; i < kF1Steps * 2 * 64; i += kF1Threads) afr[i] = p.wlds[i];
  for (int i = tid; i < kF1Ring * kF1RowU4; i += kF1Threads) zimg[i] = uint4{0u, 0u, 0u, 0u};
  float bias4[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bias4[q] = p.bias[4 * kh + q];

  // one staging task = 4 consecutive zp words (bins 4 u - 56 .. 4 u - 53) of one image row -> 8 B of hi and of lo,
  // written to copy 0 at element 4 u and to copy 1 (shifted by 4 bins) at element 4 u - 4
  constexpr int kTasksRow = kZRow / 4;  // 112
  auto stage_row_task = [&](const uint32_t* __restrict__ zwin, int row, int u) {
    const uint4 wv = *reinterpret_cast<const uint4*>(zwin + (int64_t)(row + 1) * kZRow + 4 * u);
    uint2 h2, l2;
    h2.x = (wv.x & 0xffffu) | (wv.y << 16);
    h2.y = (wv.z & 0xffffu) | (wv.w << 16);
    l2.x = (wv.x >> 16) | (wv.y & 0xffff0000u);
    l2.y = (wv.z >> 16) | (wv.w & 0xffff0000u);
    uint2* rowp = reinterpret_cast<uint2*>(zimg + ((row + kF1Ring) % kF1Ring) * kF1RowU4);
    rowp[u] = h2;                                   // hi copy 0
    rowp[2 * 2 * kF1Copy + u] = l2;                 // lo copy 0
    if (u > 0) {
      rowp[2 * kF1Copy + u - 1] = h2;               // hi copy 1
      rowp[2 * 3 * kF1Copy + u - 1] = l2;           // lo copy 1
    }
  };

  const int rows_per = (kFrames + p.chunks - 1) / p.chunks;
  const int n_items = p.n_windows * p.chunks;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int b = item / p.chunks;
    const int t0 = (item - b * p.chunks) * rows_per;
    const int t1 = t0 + rows_per < kFrames ? t0 + rows_per : kFrames;
    const int npos = (t1 - t0) * kF1Groups;
    const int nrounds = (npos + kF1Round - 1) / kF1Round;
    const uint32_t* zwin = p.zp + (int64_t)b * kZWin;  // padded window: frame -1 is row 0, bin -56 is word 0
    float* c1b = p.c1 + (int64_t)b * kC1Win;

    lds_barrier();
    int staged_hi = t0 + (kF1Round - 1) / kF1Groups + 1;
    staged_hi = staged_hi < t1 ? staged_hi : t1;
    for (int e = tid; e < (staged_hi - t0 + 2) * kTasksRow; e += kF1Threads) {
      const int ri = e / kTasksRow;
      stage_row_task(zwin, t0 - 1 + ri, e - ri * kTasksRow);
    }
    lds_barrier();

    for (int k = 0; k < nrounds; ++k) {
      int need_hi = t0 + (kF1Round * (k + 1) + kF1Round - 1) / kF1Groups + 1;
      need_hi = need_hi < t1 ? need_hi : t1;
      const int n_new = (k + 1 < nrounds) ? need_hi - staged_hi : 0;
      const int first_new = staged_hi + 1;

      const int pos = kF1Round * k + 32 * w + li;
      const bool pvalid = pos < npos;
      const int posc = pvalid ? pos : npos - 1;
      const int prr = posc / kF1Groups;
      const int pgrp = kD1EdgeGroups + posc - prr * kF1Groups;  // group 5..60
      const int prow = t0 + prr;
      // B operand: z elements 4 m + 16 s + 8 kh .. + 7 of copy (m & 1); uint4 index (m - copy) / 2 + kh + 2 s
      const int cpy = pgrp & 1;
      const int boff = cpy * kF1Copy + ((pgrp - cpy) >> 1) + kh;
      int rowb[3];
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) rowb[dt] = ((prow - 1 + dt + kF1Ring) % kF1Ring) * kF1RowU4 + boff;

      f32x16 hh, xx;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        hh[r] = 0.0f;
        xx[r] = 0.0f;
      }
      f16x8 ah[kF1Steps], al[kF1Steps], bh[kF1Steps], bl[kF1Steps];
      auto issue = [&](int s) {
        const int dt = s / 12, e = s - 12 * dt;
        if (WLO) al[s] = __builtin_bit_cast(f16x8, afr[(2 * s + 1) * 64 + lane]);
        bh[s] = __builtin_bit_cast(f16x8, zimg[rowb[dt] + 2 * e]);
        ah[s] = __builtin_bit_cast(f16x8, afr[(2 * s) * 64 + lane]);
        bl[s] = __builtin_bit_cast(f16x8, zimg[rowb[dt] + 2 * e + 2 * kF1Copy]);
      };
#pragma unroll
      for (int s = 0; s < kF1Pf; ++s) issue(s);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < kF1Steps; ++s) {
        if (s + kF1Pf < kF1Steps) issue(s + kF1Pf);
        // the z rows of the next round: at most 6 rows x 112 tasks, two per thread, early in the round
        if (s == 2 || s == 14) {
          const int e = (s == 2 ? 0 : kF1Threads) + tid;
          if (e < n_new * kTasksRow) {
            const int ri = e / kTasksRow;
            stage_row_task(zwin, first_new + ri, e - ri * kTasksRow);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (WLO) xx = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh[s], xx, 0, 0, 0);
        hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[s], hh, 0, 0, 0);
        xx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl[s], xx, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      staged_hi += n_new;
      if (pvalid) d1_store(c1b + ((int64_t)prow * kC1Row + kC1Pad + 4 * pgrp) * 8 + 4 * kh, hh, xx, bias4);
      lds_barrier();
    }
  }
}

template <class Geo>
static void launch_exact(const Conv1Params& p, int n_cu, bool wlo, hipStream_t stream) {
  const int items = p.n_windows * p.chunks;
  const int grid = items < n_cu ? items : n_cu;
  if (wlo)
    hipLaunchKernelGGL((contour_conv1_kernel<Geo, true>), dim3(grid), dim3(Geo::kWaves * 64), 0, stream, p);
  else
    hipLaunchKernelGGL((contour_conv1_kernel<Geo, false>), dim3(grid), dim3(Geo::kWaves * 64), 0, stream, p);
}

static int conv1_chunks(int n_windows, int n_cu) {
  // one workgroup per CU; split windows into row chunks when there are fewer windows than CUs
  int chunks = 1;
  while (chunks < 4 && n_windows * chunks < n_cu) chunks *= 2;
  return chunks;
}

// BP_CONV1=full: the exact kernel over every group (A/B reference of the folded form)
bool contour_conv1_full() {
  static const bool full = [] {
    const char* e = ab_env("BP_CONV1");
    return e && strcmp(e, "full") == 0;
  }();
  return full;
}

// exact 8-channel kernel: the rim groups of every frame (or every group with BP_CONV1=full)
void launch_contour_conv1_exact(const uint32_t* zp, const void* wlds, const float* bias, float* c1, int n_windows,
                                int n_cu, bool weights_have_lo, hipStream_t stream) {
  Conv1Params p{zp, static_cast<const uint4*>(wlds), bias, c1, n_windows, conv1_chunks(n_windows, n_cu)};
  if (contour_conv1_full())
    launch_exact<FullGeo>(p, n_cu, weights_have_lo, stream);
  else
    launch_exact<EdgeGeo>(p, n_cu, weights_have_lo, stream);
}

// folded kernel: the interior groups (nothing to do with BP_CONV1=full)
void launch_contour_conv1_folded(const uint32_t* zp, const void* wfold, const float* bias, float* c1, int n_windows,
                                 int n_cu, bool weights_have_lo, hipStream_t stream) {
  if (contour_conv1_full()) return;
  Conv1Params p{zp, static_cast<const uint4*>(wfold), bias, c1, n_windows, conv1_chunks(n_windows, n_cu)};
  const int items = p.n_windows * p.chunks;
  const int grid = items < n_cu ? items : n_cu;
  if (weights_have_lo)
    hipLaunchKernelGGL(contour_conv1_folded_kernel<true>, dim3(grid), dim3(kF1Threads), 0, stream, p);
  else
    hipLaunchKernelGGL(contour_conv1_folded_kernel<false>, dim3(grid), dim3(kF1Threads), 0, stream, p);
}

}  // namespace bp
