// Fused contour branch — split-precision matrix-core kernel (default path).
//
//   Conv2D 8->8, (3 frames x 39 bins), "same", folded BN, ReLU on the harmonic stack   (models.py:241-250,
//   nn.py:69-88), then Conv2D 8->1, 5x5, "same", sigmoid (models.py:254-263), FlattenFreqCh (nn.py:105-119)
//                                                                                       -> contour
// 65 % of the whole path's FLOPs are the first convolution.  Mapping:
//
//   * conv1 is an implicit GEMM in transposed form on v_mfma_f32_32x32x16_f16:
//         C1[(out channel o, bin offset j) (32 rows)][position (32 cols)] = Wt[(o,j)][k] x S[k][position]
//     position = (frame, group of 4 adjacent bins); the rows carry a 4-bin Toeplitz expansion of the 39-tap
//     kernel (42/39 extra taps); one k-step = 2 adjacent taps x 8 stack channels = one ds_read_b128 of the
//     LDS image of the harmonic stack (channel-last, pre-split f16 hi | scaled lo, 4 phase planes so the
//     32 lanes of a read are consecutive 16-byte slots).  K = 3 frames x 21 tap pairs = 63 k-steps.
//   * the 63 A fragments (hi + lo = 126 x 16 B per lane) exceed one wave's registers, so K is split over
//     the 4 waves (16 steps = 128 VGPRs each, resident for the whole kernel).  The partial sums are
//     combined by a REDUCE-SCATTER through LDS: wave g ends up with the complete sums of bin offset j = g
//     (12 values out, 12 in per lane instead of 16 + 16 for a gather to one owner).
//   * what wave g then holds per lane — 4 channels of one position/bin — is exactly the B operand of the
//     tap projection of the second convolution:  P[tap (25)][position] = W2t[tap][c] x relu(C1)[c][position]
//     (K = 8 channels; hi and scaled lo share the 16 k-slots, 2 MFMAs).  conv2's 5x5 spatial sum is then
//     25 adds per pixel: P goes through an LDS scratch and is accumulated into a ring of output rows.
//   * a workgroup walks a time chunk of one window linearly, 32 positions at a time, with a 5-row ring of
//     the stack image (next row prefetched from HBM while the MFMAs of the current tile run): no halo
//     restaging, no c1 / stack tensor in HBM.  2 chunks per window, 2 workgroups per CU.
//
// Numerics: operands are x = hi + lo, lo stored * 2^11 (f16 exponent range, see cqt_mfma.hip); products
// hi*hi + (lo*hi + hi*lo) * 2^-11 accumulate in fp32.  Deterministic: fixed reduction orders everywhere.
//
// Roofline: f16 MFMA issue.  Algorithmic work 680.0 + 18.2 MFLOP per window;
// bytes per window: 214,656 (zp) read, 181,632 written.
#include <stdio.h>
#include <stdlib.h>

#include "bp_common.h"

namespace bp {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int kCbThreads = 256;
constexpr int kCbChunks = 2;
constexpr int kCbChunkFrames = kFrames / kCbChunks;   // 86 output frames per chunk
constexpr int kCbRows = kCbChunkFrames + 4;            // conv1 rows a chunk needs (conv2 pads 2 + 2)
constexpr int kCbGroups = kFreqC / 4;                  // 66 four-bin groups per row
constexpr int kCbPos = kCbRows * kCbGroups;            // 5940 positions
constexpr int kCbTiles = (kCbPos + 31) / 32;           // 186
constexpr int kCbQ = 76;                               // slots per phase plane
constexpr int kCbSlots = 4 * kCbQ;                     // 304 slots per image row
constexpr int kCbRing = 5;                             // image rows resident
constexpr int kCbORing = 6;                            // output rows accumulating
constexpr int kCbScrT = 136;                           // scratch floats per tap (132 pixels + skew)
constexpr int kCbStepsTotal = 63, kCbStepsWave = 16;
static_assert(kFrames % kCbChunks == 0, "chunks tile the window");
static_assert(kCbTiles * 32 - kCbPos >= 1, "the 2 deferred pixels of the last tile must be padding");

struct ContourParams {
  const uint32_t* zp;   // [n][172][kZRow] pre-split z (zpack_kernel)
  const uint4* wfrag;   // [4 waves][16 steps][hi|lo][64] conv1 A fragments, then [2][64] conv2 A fragments
  const float* wf32;    // bias1[8], bias2
  float* contour;       // [n][172][264]
  int n_windows;
  int dbg;                   // timing experiments only (tools/): 1 = skip the epilogue, 2 = no LDS reads in the MFMA phase
  int dephase;               // s_sleep units (64 clk) the upper half of the grid waits before its first tile
  unsigned long long* prof;  // optional [4 waves][8] cycle totals of block 0 (tools/ only; null in production)
};

// Image slot `slot` of a row holds the 8 stack channels of bin f = 4 q + pl - 20 (pl = slot / 76, q = slot % 76).
__device__ __forceinline__ int cb_slot_bin(int slot) {
  const int pl = slot / kCbQ, q = slot - pl * kCbQ;
  return 4 * q + pl - 20;
}
// issue the 8 loads of one slot (addresses clamped into the row: no masking yet, so no dependent ALU and the
// loads stay in flight together)
__device__ __forceinline__ void cb_issue(const uint32_t* __restrict__ zrow, int f, uint32_t (&u)[8]) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    int g = f + harm_shift(c);
    g = g < 0 ? 0 : (g > kZRow - 1 ? kZRow - 1 : g);
    u[c] = zrow[g];
  }
}
// zero what lies outside the cropped stack (nn.py:87: crop to 264 bins, then "same" padding) or the window
__device__ __forceinline__ void cb_mask(int f, bool row_ok, uint32_t (&u)[8]) {
  const bool inside = row_ok && f >= 0 && f < kFreqC;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int g = f + harm_shift(c);
    u[c] = (inside && g >= 0 && g < kBins) ? u[c] : 0u;
  }
}
__device__ __forceinline__ void cb_gather(const uint32_t* __restrict__ zpb, int row, int slot, uint32_t (&u)[8]) {
  const bool row_ok = row >= 0 && row < kFrames;
  const int f = cb_slot_bin(slot);
  cb_issue(zpb + (int64_t)(row_ok ? row : 0) * kZRow, f, u);
  cb_mask(f, row_ok, u);
}

__device__ __forceinline__ void cb_put(const uint32_t (&u)[8], uint4* __restrict__ img_hi,
                                       uint4* __restrict__ img_lo, int idx) {
  uint4 vh, vl;
  vh.x = (u[0] & 0xffffu) | (u[1] << 16);
  vh.y = (u[2] & 0xffffu) | (u[3] << 16);
  vh.z = (u[4] & 0xffffu) | (u[5] << 16);
  vh.w = (u[6] & 0xffffu) | (u[7] << 16);
  vl.x = (u[0] >> 16) | (u[1] & 0xffff0000u);
  vl.y = (u[2] >> 16) | (u[3] & 0xffff0000u);
  vl.z = (u[4] >> 16) | (u[5] & 0xffff0000u);
  vl.w = (u[6] >> 16) | (u[7] & 0xffff0000u);
  img_hi[idx] = vh;
  img_lo[idx] = vl;
}

// conv1 partial sums of one wave's K slice
template <int WAVE>
__device__ __forceinline__ void cb_mfma(const uint4* __restrict__ img_hi, const uint4* __restrict__ img_lo,
                                        const int (&rowslot)[3], int lo_off, int hi_off,
                                        const uint4 (&wh)[kCbStepsWave], const uint4 (&wl)[kCbStepsWave],
                                        f32x16& a_hh, f32x16& a_lh, f32x16& a_hl) {
#pragma unroll
  for (int s = 0; s < kCbStepsWave; ++s) {
    const int step = WAVE * kCbStepsWave + s;
    if (step >= kCbStepsTotal) continue;
    const int dt = step / 21, ep = step - 21 * dt;
    const int r0 = (2 * ep + 1) & 3, q0 = (2 * ep + 1) >> 2;
    const int slot = rowslot[dt] + ((r0 == 1) ? lo_off : hi_off) + r0 * kCbQ + q0;
    const f16x8 bh = __builtin_bit_cast(f16x8, img_hi[slot]);
    const f16x8 bl = __builtin_bit_cast(f16x8, img_lo[slot]);
    const f16x8 ah = __builtin_bit_cast(f16x8, wh[s]);
    const f16x8 al = __builtin_bit_cast(f16x8, wl[s]);
    // three independent accumulation chains: no back-to-back dependent MFMAs
    a_hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, a_hh, 0, 0, 0);
    a_lh = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, a_lh, 0, 0, 0);
    a_hl = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, a_hl, 0, 0, 0);
  }
}

__global__ __launch_bounds__(kCbThreads, 2) void contour_branch_kernel(ContourParams p) {
  __shared__ __attribute__((aligned(16))) uint4 img_hi[kCbRing * kCbSlots];
  __shared__ __attribute__((aligned(16))) uint4 img_lo[kCbRing * kCbSlots];
  __shared__ float xbuf[4 * 3 * 4 * 64];      // [dst wave][src slot][q][lane]
  __shared__ float scr[25 * kCbScrT];         // P[tap][pixel + 4 (+ skew)] of the current tile
  __shared__ float oring[kCbORing * kFreqC];  // output rows being accumulated
  __shared__ float tailb[2 * 100];            // P of the last position of the previous tile, [tap][j]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int g = wave_id();
  const int h = lane >> 5, li = lane & 31;

  // resident conv1 A fragments of this wave's K slice, conv2 A fragments, biases
  uint4 wh[kCbStepsWave], wl[kCbStepsWave];
  {
    const uint4* wp = p.wfrag + (size_t)g * kCbStepsWave * 2 * 64 + lane;
#pragma unroll
    for (int s = 0; s < kCbStepsWave; ++s) {
      wh[s] = wp[(2 * s) * 64];
      wl[s] = wp[(2 * s + 1) * 64];
    }
  }
  const uint4 a2m = p.wfrag[(size_t)4 * kCbStepsWave * 2 * 64 + lane];
  const uint4 a2x = p.wfrag[(size_t)4 * kCbStepsWave * 2 * 64 + 64 + lane];
  float bias1[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    bias1[q] = p.wf32[2 * q + h];
    asm volatile("" : "+v"(bias1[q]));  // pin in a register: a reload inside the loop would serialise on vmcnt
  }
  float bias2 = p.wf32[8];
  asm volatile("" : "+v"(bias2));

  const int n_items = p.n_windows * kCbChunks;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int b = item / kCbChunks;
    const int T0 = (item - b * kCbChunks) * kCbChunkFrames;
    const int T1 = T0 + kCbChunkFrames;
    const int R0 = T0 - 2;  // first conv1 row of the chunk
    const uint32_t* zpb = p.zp + (int64_t)b * kFrames * kZRow;
    float* outb = p.contour + (int64_t)b * kPlaneC;

    __syncthreads();  // previous item is done with LDS
    for (int i = tid; i < kCbORing * kFreqC; i += kCbThreads) oring[i] = 0.0f;
    for (int i = tid; i < kCbRing * kCbSlots; i += kCbThreads) {
      const int rr = i / kCbSlots, slot = i - rr * kCbSlots;
      const int row = R0 - 1 + rr;
      uint32_t u[8];
      cb_gather(zpb, row, slot, u);
      cb_put(u, img_hi, img_lo, ((row + 5 * 8) % kCbRing) * kCbSlots + slot);
    }
    int next_emit = T0;
    int pending_row = -1000;
    __syncthreads();

    for (int n = 0; n <= kCbTiles; ++n) {
      const bool compute = n < kCbTiles;  // iteration kCbTiles only flushes the last rows

      // ---- image row that becomes visible 3 rows ahead when the walk enters a new row: slots 0..255 during
      // this tile, slots 256..303 during the next one (the row is first read two tiles later at the earliest)
      uint32_t pf[8];
      int stage_row = -1000, stage_slot = 0;
      if (pending_row != -1000) {
        stage_row = pending_row;
        stage_slot = tid + kCbThreads;
        pending_row = -1000;
      } else if (compute && n > 0) {
        const int rl = (32 * n) / kCbGroups;
        if (rl != (32 * (n - 1)) / kCbGroups && rl + 3 <= kCbRows) {
          stage_row = R0 + rl + 3;
          stage_slot = tid;
          pending_row = stage_row;
        }
      }
      const bool staging = stage_row != -1000 && stage_slot < kCbSlots;
      const bool stage_row_ok = stage_row >= 0 && stage_row < kFrames;
      const int stage_f = cb_slot_bin(stage_slot);

      float own[4] = {0.f, 0.f, 0.f, 0.f};  // this wave's partial sums for bin offset j = g
      bool cvalid = false;
      if (compute) {
        // ---- conv1: this wave's K slice of tile n
        const int pos = 32 * n + li;
        const int posc = pos < kCbPos ? pos : kCbPos - 1;
        const int rr = posc / kCbGroups;
        const int mf = posc - rr * kCbGroups;
        const int row = R0 + rr;
        cvalid = pos < kCbPos && row >= 0 && row < kFrames;  // conv2 zero-pads outside the window
        int rowslot[3];
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) rowslot[dt] = ((row - 1 + dt + 5 * 8) % kCbRing) * kCbSlots;
        const int lo_off = mf + h * kCbQ;                  // tap plane 1 -> 2 (same group)
        const int hi_off = mf + h * (1 - 3 * kCbQ);        // tap plane 3 -> 0 of the next group
        f32x16 a_hh, a_lh, a_hl;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          a_hh[r] = 0.0f;
          a_lh[r] = 0.0f;
          a_hl[r] = 0.0f;
        }
        switch (g) {
          case 0: cb_mfma<0>(img_hi, img_lo, rowslot, lo_off, hi_off, wh, wl, a_hh, a_lh, a_hl); break;
          case 1: cb_mfma<1>(img_hi, img_lo, rowslot, lo_off, hi_off, wh, wl, a_hh, a_lh, a_hl); break;
          case 2: cb_mfma<2>(img_hi, img_lo, rowslot, lo_off, hi_off, wh, wl, a_hh, a_lh, a_hl); break;
          default: cb_mfma<3>(img_hi, img_lo, rowslot, lo_off, hi_off, wh, wl, a_hh, a_lh, a_hl); break;
        }
        // reduce-scatter: register r holds (o = 2(r>>2) + h, j = r & 3); bin offset j goes to wave j
        // (g is wave-uniform: the own-partial selects are scalar-condition moves, no register indexing)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float pv = a_hh[4 * q + j] + (a_lh[4 * q + j] + a_hl[4 * q + j]) * kLoUnscale;
            if (j == g) {
              own[q] = pv;
            } else {
              const int sidx = g < j ? g : g - 1;
              xbuf[((j * 3 + sidx) * 4 + q) * 64 + lane] = pv;
            }
          }
        }
      }
      __syncthreads();  // B1: partials exchanged; previous tile's accumulation into oring is complete

      // ---- emit the output rows completed by the previous tile (conv1 rows <= done are fully accumulated)
      {
        const int done = n == 0 ? -1 : (128 * (n - 1) + 126) / kFreqC - 1;  // chunk-relative conv1 row
        int last = R0 + done - 2;
        last = last < T1 - 1 ? last : T1 - 1;
        for (; next_emit <= last; ++next_emit) {
          float* orow = oring + (next_emit % kCbORing) * kFreqC;
          for (int f = tid; f < kFreqC; f += kCbThreads) {
            outb[next_emit * kFreqC + f] = sigmoidf_exact(orow[f] + bias2);
            orow[f] = 0.0f;
          }
        }
      }
      if (!compute) break;
      // loads of the image row to stage: issued here (the conv1 accumulators are dead, registers are free), in
      // flight under the reduction, projection, barrier B2 and the spatial sum; written to LDS at the end
      if (staging) cb_issue(zpb + (int64_t)(stage_row_ok ? stage_row : 0) * kZRow, stage_f, pf);

      // ---- finish conv1 for bin offset j = g, ReLU, and project onto the 25 taps of conv2
      {
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float* xp = xbuf + ((g * 3) * 4 + q) * 64 + lane;
          const float e0 = xp[0], e1 = xp[4 * 64], e2 = xp[8 * 64];
          // fixed summation order: source waves 0, 1, 2, 3 with the own partial in its place
          const float x0 = g == 0 ? own[q] : e0;
          const float x1 = g == 1 ? own[q] : (g < 1 ? e0 : e1);
          const float x2 = g == 2 ? own[q] : (g < 2 ? e1 : e2);
          const float x3 = g == 3 ? own[q] : e2;
          const float s = fmaxf((((x0 + x1) + x2) + x3) + bias1[q], 0.0f);
          v[q] = cvalid ? s : 0.0f;
        }
        f16x8 b2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const _Float16 hi = (_Float16)v[q];
          b2[q] = hi;
          b2[4 + q] = (_Float16)((v[q] - (float)hi) * kLoScale);
        }
        f32x16 pm, px;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pm[r] = 0.0f;
          px[r] = 0.0f;
        }
        pm = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a2m), b2, pm, 0, 0, 0);
        px = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a2x), b2, px, 0, 0, 0);
        // P[tap][position li, bin g] -> scratch (pixel y = 4 li + g at index y + 4 + skew), tail copy
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int t0 = (r & 3) + 8 * (r >> 2);  // tap of half h = 0; h = 1 adds 4
          if (t0 >= 25) continue;
          const float pv = pm[r] + px[r] * kLoUnscale;
          const int tap = t0 + 4 * h;
          if (t0 + 4 < 25 || h == 0) {
            scr[tap * kCbScrT + 4 * (li + 1) + g + ((li + 1) >> 3)] = pv;
            if (li == 31) tailb[(n & 1) * 100 + tap * 4 + g] = pv;
          }
        }
        if (lane < 25) scr[lane * kCbScrT + g] = tailb[((n + 1) & 1) * 100 + lane * 4 + g];
      }
      __syncthreads();  // B2: P of the tile (and the previous tile's last position) is in scr

      // ---- conv2 spatial sum: pixel x of the tile, x in [-2, 126) (the last 2 wait for their neighbours)
      {
        int tv = tid;
        asm volatile("" : "+v"(tv));  // keep this block's address arithmetic out of the loop-invariant set
        const int x = (tv & 127) - 2;
        const int part2 = tv >> 7;  // 0: frame taps 0..2, 1: frame taps 3..4
        const int G = 128 * n + x;
        if (G >= 0 && G < 4 * kCbPos) {
          const int rr = G / kFreqC;
          const int f = G - rr * kFreqC;
          const int row = R0 + rr;
          // physical scratch index of pixel x + dw - 2 and whether it is inside the row (zero padding)
          int phys[5];
          bool okw[5];
#pragma unroll
          for (int dw = 0; dw < 5; ++dw) {
            const int y4 = x + dw + 2;  // (x + dw - 2) + 4, always inside the scratch row
            phys[dw] = y4 + (y4 >> 5);
            okw[dw] = (unsigned)(f + dw - 2) < (unsigned)kFreqC;
          }
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            const int dt = part2 ? 3 + d : d;
            if (d == 2 && part2) continue;
            const int t = row - dt + 2;
            const float* sp = scr + dt * 5 * kCbScrT;
            float s = 0.0f;
#pragma unroll
            for (int dw = 0; dw < 5; ++dw) {
              const float pv = sp[dw * kCbScrT + phys[dw]];
              s += okw[dw] ? pv : 0.0f;
            }
            if (t >= T0 && t < T1) oring[(t % kCbORing) * kFreqC + f] += s;
          }
        }
      }
      if (staging) {
        cb_mask(stage_f, stage_row_ok, pf);
        cb_put(pf, img_hi, img_lo, ((stage_row + 5 * 8) % kCbRing) * kCbSlots + stage_slot);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// v2 of the same kernel: identical mapping and LDS layout, different instruction schedule.
//   * the whole chunk walk is specialised on the wave index G (one dispatch per kernel): the K-slice offsets
//     of the image reads are immediates, the reduce-scatter has no scalar selects or branches;
//   * image fragments are prefetched kCbPf k-steps ahead of the MFMAs that consume them (the register budget
//     is freed by accumulating the two cross terms lo*hi and hi*lo, which share the 2^-11 scale, in ONE chain);
//   * the reduce-scatter moves float4 (ds_write_b128 / ds_read_b128).
constexpr int kCbPf = 2;
constexpr int kCbLoOff = kCbRing * kCbSlots;  // img[] = hi image, then lo image

template <int G, bool NOLDS>
__device__ __forceinline__ void cb2_mfma(const uint4* __restrict__ img, const int (&rowslot)[3], int lo_off,
                                         int hi_off, const uint4 (&wh)[kCbStepsWave],
                                         const uint4 (&wl)[kCbStepsWave], f32x16& a_hh, f32x16& a_x) {
  constexpr int NS = (G == 3) ? kCbStepsTotal - 3 * kCbStepsWave : kCbStepsWave;
  f16x8 bh[NS], bl[NS];
  auto issue = [&](int s) {
    const int step = G * kCbStepsWave + s;
    const int dt = step / 21, ep = step - 21 * dt;
    const int r0 = (2 * ep + 1) & 3, q0 = (2 * ep + 1) >> 2;
    const int slot = rowslot[dt] + ((r0 == 1) ? lo_off : hi_off) + r0 * kCbQ + q0;
    if (NOLDS) {
      uint4 t{(unsigned)slot, (unsigned)s, 1u, 2u};
      asm volatile("" : "+v"(t.x), "+v"(t.y), "+v"(t.z), "+v"(t.w));
      bh[s] = __builtin_bit_cast(f16x8, t);
      bl[s] = __builtin_bit_cast(f16x8, t);
    } else {
      bh[s] = __builtin_bit_cast(f16x8, img[slot]);
      bl[s] = __builtin_bit_cast(f16x8, img[slot + kCbLoOff]);
    }
  };
#pragma unroll
  for (int s = 0; s < kCbPf; ++s) issue(s);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if (s + kCbPf < NS) issue(s + kCbPf);
    __builtin_amdgcn_sched_barrier(0);
    const f16x8 ah = __builtin_bit_cast(f16x8, wh[s]);
    const f16x8 al = __builtin_bit_cast(f16x8, wl[s]);
    a_x = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[s], a_x, 0, 0, 0);
    a_hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[s], a_hh, 0, 0, 0);
    a_x = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[s], a_x, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// phase k = time from the previous stamp to stamp k, summed over the tiles of a chunk
#define CB_STAMP(k)                                              \
  if (PROF) {                                                    \
    const unsigned long long t_now = __builtin_readcyclecounter(); \
    acc_t[k] += t_now - t_prev;                                  \
    t_prev = t_now;                                              \
  }
template <int G, bool PROF>
__device__ __forceinline__ void cb2_run(const ContourParams& p, uint4* __restrict__ img, float4* __restrict__ xbuf4,
                                        float* __restrict__ scr, float* __restrict__ oring,
                                        float* __restrict__ tailb) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int h = lane >> 5, li = lane & 31;
  uint4* img_hi = img;
  uint4* img_lo = img + kCbLoOff;

  uint4 wh[kCbStepsWave], wl[kCbStepsWave];
  {
    const uint4* wp = p.wfrag + (size_t)G * kCbStepsWave * 2 * 64 + lane;
#pragma unroll
    for (int s = 0; s < kCbStepsWave; ++s) {
      wh[s] = wp[(2 * s) * 64];
      wl[s] = wp[(2 * s + 1) * 64];
    }
  }
  const uint4 a2m = p.wfrag[(size_t)4 * kCbStepsWave * 2 * 64 + lane];
  const uint4 a2x = p.wfrag[(size_t)4 * kCbStepsWave * 2 * 64 + 64 + lane];
  float bias1[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    bias1[q] = p.wf32[2 * q + h];
    asm volatile("" : "+v"(bias1[q]));
  }
  float bias2 = p.wf32[8];
  asm volatile("" : "+v"(bias2));

  unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_prev = PROF ? __builtin_readcyclecounter() : 0;
  const int n_items = p.n_windows * kCbChunks;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int b = item / kCbChunks;
    const int T0 = (item - b * kCbChunks) * kCbChunkFrames;
    const int T1 = T0 + kCbChunkFrames;
    const int R0 = T0 - 2;
    const uint32_t* zpb = p.zp + (int64_t)b * kFrames * kZRow;
    float* outb = p.contour + (int64_t)b * kPlaneC;

    __syncthreads();
    for (int i = tid; i < kCbORing * kFreqC; i += kCbThreads) oring[i] = 0.0f;
    for (int i = tid; i < kCbRing * kCbSlots; i += kCbThreads) {
      const int rr = i / kCbSlots, slot = i - rr * kCbSlots;
      const int row = R0 - 1 + rr;
      uint32_t u[8];
      cb_gather(zpb, row, slot, u);
      cb_put(u, img_hi, img_lo, ((row + 5 * 8) % kCbRing) * kCbSlots + slot);
    }
    int next_emit = T0;
    int pending_row = -1000;
    __syncthreads();

    for (int n = 0; n <= kCbTiles; ++n) {
      const bool compute = n < kCbTiles;
      uint32_t pf[8];
      int stage_row = -1000, stage_slot = 0;
      if (pending_row != -1000) {
        stage_row = pending_row;
        stage_slot = tid + kCbThreads;
        pending_row = -1000;
      } else if (compute && n > 0) {
        const int rl = (32 * n) / kCbGroups;
        if (rl != (32 * (n - 1)) / kCbGroups && rl + 3 <= kCbRows) {
          stage_row = R0 + rl + 3;
          stage_slot = tid;
          pending_row = stage_row;
        }
      }
      const bool staging = stage_row != -1000 && stage_slot < kCbSlots;
      const bool stage_row_ok = stage_row >= 0 && stage_row < kFrames;
      const int stage_f = cb_slot_bin(stage_slot);

      float own[4] = {0.f, 0.f, 0.f, 0.f};
      bool cvalid = false;
      CB_STAMP(0);
      if (compute) {
        const int pos = 32 * n + li;
        const int posc = pos < kCbPos ? pos : kCbPos - 1;
        const int rr = posc / kCbGroups;
        const int mf = posc - rr * kCbGroups;
        const int row = R0 + rr;
        cvalid = pos < kCbPos && row >= 0 && row < kFrames;
        int rowslot[3];
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) rowslot[dt] = ((row - 1 + dt + 5 * 8) % kCbRing) * kCbSlots;
        const int lo_off = mf + h * kCbQ;
        const int hi_off = mf + h * (1 - 3 * kCbQ);
        f32x16 a_hh, a_x;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          a_hh[r] = 0.0f;
          a_x[r] = 0.0f;
        }
        if (p.dbg & 2)
          cb2_mfma<G, true>(img, rowslot, lo_off, hi_off, wh, wl, a_hh, a_x);
        else
          cb2_mfma<G, false>(img, rowslot, lo_off, hi_off, wh, wl, a_hh, a_x);
        // reduce-scatter: register r holds (o = 2(r>>2) + h, j = r & 3); bin offset j goes to wave j
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float4 v4;
          v4.x = a_hh[j] + a_x[j] * kLoUnscale;
          v4.y = a_hh[4 + j] + a_x[4 + j] * kLoUnscale;
          v4.z = a_hh[8 + j] + a_x[8 + j] * kLoUnscale;
          v4.w = a_hh[12 + j] + a_x[12 + j] * kLoUnscale;
          if (j == G) {
            own[0] = v4.x;
            own[1] = v4.y;
            own[2] = v4.z;
            own[3] = v4.w;
          } else {
            constexpr int dummy = 0;
            (void)dummy;
            const int sidx = G < j ? G : G - 1;
            xbuf4[(j * 3 + sidx) * 64 + lane] = v4;
          }
        }
      }
      CB_STAMP(1);
      __syncthreads();  // B1
      CB_STAMP(2);

      {
        const int done = n == 0 ? -1 : (128 * (n - 1) + 126) / kFreqC - 1;
        int last = R0 + done - 2;
        last = last < T1 - 1 ? last : T1 - 1;
        for (; next_emit <= last; ++next_emit) {
          float* orow = oring + (next_emit % kCbORing) * kFreqC;
          for (int f = tid; f < kFreqC; f += kCbThreads) {
            outb[next_emit * kFreqC + f] = sigmoidf_exact(orow[f] + bias2);
            orow[f] = 0.0f;
          }
        }
      }
      if (!compute) break;
      if (p.dbg & 1) {
        __syncthreads();
        continue;
      }
      if (staging) cb_issue(zpb + (int64_t)(stage_row_ok ? stage_row : 0) * kZRow, stage_f, pf);
      CB_STAMP(3);

      {
        // complete sums of bin offset j = G: source waves in the fixed order 0, 1, 2, 3 (own partial in place)
        float4 e[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) e[s] = xbuf4[(G * 3 + s) * 64 + lane];
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float e0 = q == 0 ? e[0].x : q == 1 ? e[0].y : q == 2 ? e[0].z : e[0].w;
          const float e1 = q == 0 ? e[1].x : q == 1 ? e[1].y : q == 2 ? e[1].z : e[1].w;
          const float e2 = q == 0 ? e[2].x : q == 1 ? e[2].y : q == 2 ? e[2].z : e[2].w;
          const float x0 = G == 0 ? own[q] : e0;
          const float x1 = G == 1 ? own[q] : (G < 1 ? e0 : e1);
          const float x2 = G == 2 ? own[q] : (G < 2 ? e1 : e2);
          const float x3 = G == 3 ? own[q] : e2;
          const float s = fmaxf((((x0 + x1) + x2) + x3) + bias1[q], 0.0f);
          v[q] = cvalid ? s : 0.0f;
        }
        f16x8 b2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const _Float16 hi = (_Float16)v[q];
          b2[q] = hi;
          b2[4 + q] = (_Float16)((v[q] - (float)hi) * kLoScale);
        }
        f32x16 pm, px;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pm[r] = 0.0f;
          px[r] = 0.0f;
        }
        pm = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a2m), b2, pm, 0, 0, 0);
        px = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a2x), b2, px, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int t0 = (r & 3) + 8 * (r >> 2);
          if (t0 >= 25) continue;
          const float pv = pm[r] + px[r] * kLoUnscale;
          const int tap = t0 + 4 * h;
          if (t0 + 4 < 25 || h == 0) {
            scr[tap * kCbScrT + 4 * (li + 1) + G + ((li + 1) >> 3)] = pv;
            if (li == 31) tailb[(n & 1) * 100 + tap * 4 + G] = pv;
          }
        }
        if (lane < 25) scr[lane * kCbScrT + G] = tailb[((n + 1) & 1) * 100 + lane * 4 + G];
      }
      CB_STAMP(4);
      __syncthreads();  // B2
      CB_STAMP(5);

      {
        int tv = tid;
        asm volatile("" : "+v"(tv));
        const int x = (tv & 127) - 2;
        const int part2 = tv >> 7;
        const int Gp = 128 * n + x;
        if (Gp >= 0 && Gp < 4 * kCbPos) {
          const int rr = Gp / kFreqC;
          const int f = Gp - rr * kFreqC;
          const int row = R0 + rr;
          int phys[5];
          bool okw[5];
#pragma unroll
          for (int dw = 0; dw < 5; ++dw) {
            const int y4 = x + dw + 2;
            phys[dw] = y4 + (y4 >> 5);
            okw[dw] = (unsigned)(f + dw - 2) < (unsigned)kFreqC;
          }
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            const int dt = part2 ? 3 + d : d;
            if (d == 2 && part2) continue;
            const int t = row - dt + 2;
            const float* sp = scr + dt * 5 * kCbScrT;
            float s = 0.0f;
#pragma unroll
            for (int dw = 0; dw < 5; ++dw) {
              const float pv = sp[dw * kCbScrT + phys[dw]];
              s += okw[dw] ? pv : 0.0f;
            }
            if (t >= T0 && t < T1) oring[(t % kCbORing) * kFreqC + f] += s;
          }
        }
      }
      CB_STAMP(6);
      if (staging) {
        cb_mask(stage_f, stage_row_ok, pf);
        cb_put(pf, img_hi, img_lo, ((stage_row + 5 * 8) % kCbRing) * kCbSlots + stage_slot);
      }
      CB_STAMP(7);
    }
  }
  if (PROF && blockIdx.x == 0 && lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) p.prof[G * 8 + k] = acc_t[k];
  }
}
#undef CB_STAMP

template <bool PROF>
__global__ __launch_bounds__(kCbThreads, 2) void contour_branch2_kernel(ContourParams p) {
  __shared__ __attribute__((aligned(16))) uint4 img[2 * kCbRing * kCbSlots];
  __shared__ __attribute__((aligned(16))) float4 xbuf4[4 * 3 * 64];
  __shared__ float scr[25 * kCbScrT];
  __shared__ float oring[kCbORing * kFreqC];
  __shared__ float tailb[2 * 100];
  {
    // dephase < 0: delay odd blocks; > 0: delay the upper half of the grid
    const bool late = p.dephase < 0 ? (blockIdx.x & 1) : (blockIdx.x >= gridDim.x / 2);
    const int nsl = p.dephase < 0 ? -p.dephase : p.dephase;
    if (late)
      for (int i = 0; i < nsl; ++i) __builtin_amdgcn_s_sleep(1);
  }
  switch (wave_id()) {
    case 0: cb2_run<0, PROF>(p, img, xbuf4, scr, oring, tailb); break;
    case 1: cb2_run<1, PROF>(p, img, xbuf4, scr, oring, tailb); break;
    case 2: cb2_run<2, PROF>(p, img, xbuf4, scr, oring, tailb); break;
    default: cb2_run<3, PROF>(p, img, xbuf4, scr, oring, tailb); break;
  }
}


// ---------------------------------------------------------------------------------------------------------
// v3: v2 + software pipelining across tiles.  The work of a tile that does not feed its own matrix products —
// the conv2 spatial sum of the PREVIOUS tile, the LDS write of the image row fetched during the previous tile
// and the sigmoid / store of a finished output row — is cut into slices that are issued between the MFMAs of
// the current tile, so one wave keeps the matrix pipe and the VALU / LDS busy at the same time; the serial
// part of a tile shrinks to: MFMA phase -> partials -> barrier -> finish + projection -> barrier.
constexpr int kCbORing3 = 7;  // emit now lags the accumulation by one more tile

#define CB_STAMP(k)                                                \
  if (PROF) {                                                      \
    const unsigned long long t_now = __builtin_readcyclecounter(); \
    acc_t[k] += t_now - t_prev;                                    \
    t_prev = t_now;                                                \
  }

template <int G, bool PROF>
__device__ __forceinline__ void cb3_run(const ContourParams& p, uint4* __restrict__ img,
                                        float4* __restrict__ xbuf4, float* __restrict__ scr,
                                        float* __restrict__ oring) {
  constexpr int NS = (G == 3) ? kCbStepsTotal - 3 * kCbStepsWave : kCbStepsWave;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int h = lane >> 5, li = lane & 31;
  uint4* img_hi = img;
  uint4* img_lo = img + kCbLoOff;

  uint4 wh[kCbStepsWave], wl[kCbStepsWave];
  {
    const uint4* wp = p.wfrag + (size_t)G * kCbStepsWave * 2 * 64 + lane;
#pragma unroll
    for (int s = 0; s < kCbStepsWave; ++s) {
      wh[s] = wp[(2 * s) * 64];
      wl[s] = wp[(2 * s + 1) * 64];
    }
  }
  const uint4 a2m = p.wfrag[(size_t)4 * kCbStepsWave * 2 * 64 + lane];
  const uint4 a2x = p.wfrag[(size_t)4 * kCbStepsWave * 2 * 64 + 64 + lane];
  // biases live in scalar registers (9 SGPRs): a vector copy per lane half would cost 5 VGPRs of a full budget
#define CB_SB(name, idx)                                                                                      \
  float name = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.wf32[idx]))); \
  asm volatile("" : "+s"(name));
  CB_SB(sb0, 0) CB_SB(sb1, 1) CB_SB(sb2, 2) CB_SB(sb3, 3) CB_SB(sb4, 4) CB_SB(sb5, 5) CB_SB(sb6, 6) CB_SB(sb7, 7)
  CB_SB(bias2, 8)
#undef CB_SB

  // spatial-sum role of this thread: pixel column x of a tile and a group of frame taps (waves 0,1: dt 0..2,
  // waves 2,3: dt 3..4)
  const int sx = (tid & 127) - 2;
  constexpr int kDt0 = (G < 2) ? 0 : 3;
  constexpr int kNdt = (G < 2) ? 3 : 2;
  // P[tap][pixel] scratch address of this lane's pixel (tap 4h, + t0 rows by immediate)
  float* const sbase = scr + (4 * h) * kCbScrT + 4 * (li + 1) + G + ((li + 1) >> 3);

  unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_prev = PROF ? __builtin_readcyclecounter() : 0;
  const int n_items = p.n_windows * kCbChunks;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int b = item / kCbChunks;
    const int T0 = (item - b * kCbChunks) * kCbChunkFrames;
    const int T1 = T0 + kCbChunkFrames;
    const int R0 = T0 - 2;
    const uint32_t* zpb = p.zp + (int64_t)b * kFrames * kZRow;
    float* outb = p.contour + (int64_t)b * kPlaneC;

    __syncthreads();
    for (int i = tid; i < kCbORing3 * kFreqC; i += kCbThreads) oring[i] = 0.0f;
    for (int i = tid; i < kCbRing * kCbSlots; i += kCbThreads) {
      const int rr = i / kCbSlots, slot = i - rr * kCbSlots;
      const int row = R0 - 1 + rr;
      uint32_t u[8];
      cb_gather(zpb, row, slot, u);
      cb_put(u, img_hi, img_lo, ((row + 5 * 8) % kCbRing) * kCbSlots + slot);
    }
    int next_emit = T0;
    int pending_row = -1000;
    // image slot fetched during the previous iteration, written to LDS during this one
    uint32_t pf[8];
    bool put_active = false, put_row_ok = false;
    int put_idx = 0, put_f = 0;
    __syncthreads();

    for (int n = 0; n <= kCbTiles + 1; ++n) {
      const bool compute = n < kCbTiles;
      // ---- which image slot this thread fetches during this iteration (as v1: a new row becomes visible 3
      // rows ahead when the walk enters a row; slots 0..255 now, slots 256..303 in the next iteration)
      int stage_row = -1000, stage_slot = 0;
      if (pending_row != -1000) {
        stage_row = pending_row;
        stage_slot = tid + kCbThreads;
        pending_row = -1000;
      } else if (compute && n > 0) {
        const int rl = (32 * n) / kCbGroups;
        if (rl != (32 * (n - 1)) / kCbGroups && rl + 3 <= kCbRows) {
          stage_row = R0 + rl + 3;
          stage_slot = tid;
          pending_row = stage_row;
        }
      }
      const bool staging = stage_row != -1000 && stage_slot < kCbSlots;

      // ---- side work of this iteration, in slices
      // (a) LDS write of the slot fetched last iteration
      auto side_put = [&]() {
        if (put_active) {
          cb_mask(put_f, put_row_ok, pf);
          cb_put(pf, img_hi, img_lo, put_idx);
        }
      };
      // (b) conv2 spatial sum of tile m = n - 1: pixel sx, frame taps kDt0 .. kDt0 + kNdt - 1
      const int m = n - 1;
      int sxl = sx;  // laundered per tile: the five skewed scratch addresses are recomputed in the MFMA shadow
      asm volatile("" : "+v"(sxl));  // instead of living in five registers across the whole loop
      const int Gp = 128 * m + sxl;
      const bool sp_ok = m >= 0 && m < kCbTiles && Gp >= 0 && Gp < 4 * kCbPos;
      const int Gc = sp_ok ? Gp : 0;
      const int srr = Gc / kFreqC;
      const int sf = Gc - srr * kFreqC;
      const int srow = R0 + srr;
      float sv[kNdt][5];
      auto side_sp_load = [&](int d) {
        const float* sp = scr + (kDt0 + d) * 5 * kCbScrT;
#pragma unroll
        for (int dw = 0; dw < 5; ++dw) {
          const int y4 = sxl + dw + 2;
          sv[d][dw] = sp[dw * kCbScrT + y4 + (y4 >> 5)];
        }
      };
      auto side_sp_sum = [&](int d) {
        float sacc = 0.0f;
#pragma unroll
        for (int dw = 0; dw < 5; ++dw) {
          const bool okw = (unsigned)(sf + dw - 2) < (unsigned)kFreqC;
          sacc += okw ? sv[d][dw] : 0.0f;
        }
        const int t = srow - (kDt0 + d) + 2;
        const bool ok = sp_ok && t >= T0 && t < T1;
        const int tc = ok ? t : T0;
        // one add per address and phase (phases are separated by barriers): deterministic
        __hip_atomic_fetch_add(&oring[(tc % kCbORing3) * kFreqC + sf], ok ? sacc : 0.0f, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
      };
      // (c) output row completed by the tiles <= n - 2
      const int done2 = n < 2 ? -1 : (128 * (n - 2) + 126) / kFreqC - 1;
      int last = R0 + done2 - 2;
      last = last < T1 - 1 ? last : T1 - 1;
      const bool do_emit = next_emit <= last;
      float* const orow = oring + (next_emit % kCbORing3) * kFreqC;
      float* const gout = outb + next_emit * kFreqC;
      float ev = 0.0f, ev2 = 0.0f;
      auto side_emit_load = [&]() {
        if (do_emit) {
          ev = orow[tid];
          orow[tid] = 0.0f;
          if (G == 3 && lane < kFreqC - kCbThreads) {
            ev2 = orow[kCbThreads + lane];
            orow[kCbThreads + lane] = 0.0f;
          }
        }
      };
      auto side_emit_store = [&]() {
        if (do_emit) {
          gout[tid] = sigmoidf_exact(ev + bias2);
          if (G == 3 && lane < kFreqC - kCbThreads) gout[kCbThreads + lane] = sigmoidf_exact(ev2 + bias2);
        }
      };

      float own[4] = {0.f, 0.f, 0.f, 0.f};
      bool cvalid = false;
      CB_STAMP(0);
      if (compute) {
        const int pos = 32 * n + li;
        const int posc = pos < kCbPos ? pos : kCbPos - 1;
        const int rr = posc / kCbGroups;
        const int mf = posc - rr * kCbGroups;
        const int row = R0 + rr;
        cvalid = pos < kCbPos && row >= 0 && row < kFrames;
        int rowslot[3];
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) rowslot[dt] = ((row - 1 + dt + 5 * 8) % kCbRing) * kCbSlots;
        const int lo_off = mf + h * kCbQ;
        const int hi_off = mf + h * (1 - 3 * kCbQ);
        f32x16 a_hh, a_x;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          a_hh[r] = 0.0f;
          a_x[r] = 0.0f;
        }
        f16x8 bh[NS], bl[NS];
        auto issue = [&](int s) {
          const int step = G * kCbStepsWave + s;
          const int dt = step / 21, ep = step - 21 * dt;
          const int r0 = (2 * ep + 1) & 3, q0 = (2 * ep + 1) >> 2;
          const int slot = rowslot[dt] + ((r0 == 1) ? lo_off : hi_off) + r0 * kCbQ + q0;
          bh[s] = __builtin_bit_cast(f16x8, img[slot]);
          bl[s] = __builtin_bit_cast(f16x8, img[slot + kCbLoOff]);
        };
        auto mfma_step = [&](int s) {
          if (s + kCbPf < NS) issue(s + kCbPf);
          __builtin_amdgcn_sched_barrier(0);
          const f16x8 ah = __builtin_bit_cast(f16x8, wh[s]);
          const f16x8 al = __builtin_bit_cast(f16x8, wl[s]);
          a_x = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[s], a_x, 0, 0, 0);
          a_hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[s], a_hh, 0, 0, 0);
          a_x = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[s], a_x, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        };
        side_put();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < kCbPf; ++s) issue(s);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(0);
        side_sp_load(0);
        mfma_step(1);
        mfma_step(2);
        side_sp_sum(0);
        side_sp_load(1);
        mfma_step(3);
        mfma_step(4);
        side_sp_sum(1);
        if (kNdt > 2) side_sp_load(2);
        mfma_step(5);
        side_emit_load();
        mfma_step(6);
        if (kNdt > 2) side_sp_sum(2);
        mfma_step(7);
        mfma_step(8);
        side_emit_store();
#pragma unroll
        for (int s = 9; s < NS; ++s) mfma_step(s);
        // reduce-scatter: register r holds (o = 2(r>>2) + h, j = r & 3); bin offset j goes to wave j
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float4 v4;
          v4.x = a_hh[j] + a_x[j] * kLoUnscale;
          v4.y = a_hh[4 + j] + a_x[4 + j] * kLoUnscale;
          v4.z = a_hh[8 + j] + a_x[8 + j] * kLoUnscale;
          v4.w = a_hh[12 + j] + a_x[12 + j] * kLoUnscale;
          if (j == G) {
            own[0] = v4.x;
            own[1] = v4.y;
            own[2] = v4.z;
            own[3] = v4.w;
          } else {
            const int sidx = G < j ? G : G - 1;
            xbuf4[(j * 3 + sidx) * 64 + lane] = v4;
          }
        }
      } else {
        side_put();
#pragma unroll
        for (int d = 0; d < kNdt; ++d) side_sp_load(d);
#pragma unroll
        for (int d = 0; d < kNdt; ++d) side_sp_sum(d);
        side_emit_load();
        side_emit_store();
      }
      if (do_emit) ++next_emit;
      put_active = false;
      CB_STAMP(1);
      __syncthreads();  // B1: partials exchanged; spatial sum of tile n - 1 complete; image row slice visible
      CB_STAMP(2);
      if (!compute) {
        if (n > kCbTiles) break;
        __syncthreads();  // keep the barrier pattern of a computing iteration
        continue;
      }
      // ---- fetch of the image slot to stage (written to LDS during the next iteration)
      if (staging) {
        put_active = true;
        put_row_ok = stage_row >= 0 && stage_row < kFrames;
        put_f = cb_slot_bin(stage_slot);
        put_idx = ((stage_row + 5 * 8) % kCbRing) * kCbSlots + stage_slot;
        cb_issue(zpb + (int64_t)(put_row_ok ? stage_row : 0) * kZRow, put_f, pf);
      }
      CB_STAMP(3);
      {
        // complete sums of bin offset j = G: source waves in the fixed order 0, 1, 2, 3 (own partial in place)
        float4 e[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) e[s] = xbuf4[(G * 3 + s) * 64 + lane];
        // P of the last position of the previous tile (this wave's column) moves to the head of the scratch
        float tl = 0.0f;
        if (lane < 25) tl = scr[lane * kCbScrT + 4 * 32 + G + 4];
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float e0 = q == 0 ? e[0].x : q == 1 ? e[0].y : q == 2 ? e[0].z : e[0].w;
          const float e1 = q == 0 ? e[1].x : q == 1 ? e[1].y : q == 2 ? e[1].z : e[1].w;
          const float e2 = q == 0 ? e[2].x : q == 1 ? e[2].y : q == 2 ? e[2].z : e[2].w;
          const float x0 = G == 0 ? own[q] : e0;
          const float x1 = G == 1 ? own[q] : (G < 1 ? e0 : e1);
          const float x2 = G == 2 ? own[q] : (G < 2 ? e1 : e2);
          const float x3 = G == 3 ? own[q] : e2;
          const float be = q == 0 ? sb0 : q == 1 ? sb2 : q == 2 ? sb4 : sb6;
          const float bo = q == 0 ? sb1 : q == 1 ? sb3 : q == 2 ? sb5 : sb7;
          const float s = fmaxf((((x0 + x1) + x2) + x3) + (h ? bo : be), 0.0f);
          v[q] = cvalid ? s : 0.0f;
        }
        f16x8 b2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const _Float16 hi = (_Float16)v[q];
          b2[q] = hi;
          b2[4 + q] = (_Float16)((v[q] - (float)hi) * kLoScale);
        }
        f32x16 pm, px;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pm[r] = 0.0f;
          px[r] = 0.0f;
        }
        pm = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a2m), b2, pm, 0, 0, 0);
        px = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a2x), b2, px, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int t0 = (r & 3) + 8 * (r >> 2);  // tap of half h = 0; h = 1 adds 4
          if (t0 >= 25) continue;
          const float pv = pm[r] + px[r] * kLoUnscale;
          if (t0 + 4 < 25) {
            sbase[t0 * kCbScrT] = pv;
          } else if (h == 0) {
            sbase[t0 * kCbScrT] = pv;
          }
        }
        if (lane < 25) scr[lane * kCbScrT + G] = tl;
      }
      CB_STAMP(4);
      __syncthreads();  // B2: P of the tile (and the previous tile's last position) is in scr
      CB_STAMP(5);
    }
  }
  if (PROF && blockIdx.x == 0 && lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) p.prof[G * 8 + k] = acc_t[k];
  }
}
#undef CB_STAMP

template <bool PROF>
__global__ __launch_bounds__(kCbThreads, 2) void contour_branch3_kernel(ContourParams p) {
  __shared__ __attribute__((aligned(16))) uint4 img[2 * kCbRing * kCbSlots];
  __shared__ __attribute__((aligned(16))) float4 xbuf4[4 * 3 * 64];
  __shared__ float scr[25 * kCbScrT];
  __shared__ float oring[kCbORing3 * kFreqC];
  {
    // dephase < 0: delay odd blocks; > 0: delay the upper half of the grid
    const bool late = p.dephase < 0 ? (blockIdx.x & 1) : (blockIdx.x >= gridDim.x / 2);
    const int nsl = p.dephase < 0 ? -p.dephase : p.dephase;
    if (late)
      for (int i = 0; i < nsl; ++i) __builtin_amdgcn_s_sleep(1);
  }
  switch (wave_id()) {
    case 0: cb3_run<0, PROF>(p, img, xbuf4, scr, oring); break;
    case 1: cb3_run<1, PROF>(p, img, xbuf4, scr, oring); break;
    case 2: cb3_run<2, PROF>(p, img, xbuf4, scr, oring); break;
    default: cb3_run<3, PROF>(p, img, xbuf4, scr, oring); break;
  }
}


void launch_contour_branch(const uint32_t* zp, const void* wfrag, const float* wf32, float* contour,
                           int n_windows, int n_cu, hipStream_t stream) {
  static const int dephase = [] {
    const char* e = getenv("BP_CONTOUR_DEPHASE");
    return e ? atoi(e) : 0;
  }();
  static const int dbg = [] {
    const char* e = getenv("BP_CONTOUR_DBG");
    return e ? atoi(e) : 0;
  }();
  ContourParams p{zp, static_cast<const uint4*>(wfrag), wf32, contour, n_windows, dbg, dephase, nullptr};
  const int items = n_windows * kCbChunks;
  static const int solo = [] {
    const char* e = getenv("BP_CONTOUR_SOLO");
    return e ? atoi(e) : 0;
  }();
  const int wg_per_cu = solo ? 1 : 2;
  const size_t dyn = solo ? 4096 : 0;  // timing experiment: extra LDS keeps a second workgroup off the CU
  const int grid = items < wg_per_cu * n_cu ? items : wg_per_cu * n_cu;
  static const int variant = [] {
    const char* e = getenv("BP_CONTOUR_VARIANT");
    return e ? atoi(e) : 2;
  }();
  if (variant == 1)
    hipLaunchKernelGGL(contour_branch_kernel, dim3(grid), dim3(kCbThreads), dyn, stream, p);
  else if (variant == 2)
    hipLaunchKernelGGL(contour_branch2_kernel<false>, dim3(grid), dim3(kCbThreads), dyn, stream, p);
  else if (variant == 4)
    hipLaunchKernelGGL(contour_branch3_kernel<false>, dim3(grid), dim3(kCbThreads), dyn, stream, p);
  else {  // 3: profiling run (tools/): per-phase cycle totals of block 0, printed to stderr
    unsigned long long* d = nullptr;
    unsigned long long hbuf[32];
    if (hipMalloc(&d, sizeof(hbuf)) != hipSuccess) return;
    (void)hipMemsetAsync(d, 0, sizeof(hbuf), stream);
    p.prof = d;
    if (variant == 3)
      hipLaunchKernelGGL(contour_branch2_kernel<true>, dim3(grid), dim3(kCbThreads), dyn, stream, p);
    else
      hipLaunchKernelGGL(contour_branch3_kernel<true>, dim3(grid), dim3(kCbThreads), dyn, stream, p);
    (void)hipMemcpyAsync(hbuf, d, sizeof(hbuf), hipMemcpyDeviceToHost, stream);
    (void)hipStreamSynchronize(stream);
    (void)hipFree(d);
    for (int w = 0; w < 4; ++w) {
      fprintf(stderr, "cbprof wave %d:", w);
      for (int k = 0; k < 8; ++k) fprintf(stderr, " %llu", hbuf[w * 8 + k] / (unsigned long long)kCbTiles);
      fprintf(stderr, "\n");
    }
  }
}

}  // namespace bp
