// Note branch, wave-private march on v_mfma_f32_16x16x32_f16 (round 6; the default.  note_march.hip keeps the 32x32x16
// form behind BP_NOTE=march32 in the A/B library).
//
//   basic_pitch/models.py:266-290: Conv2D 1->32, 7x7, strides (1,3), "same" (pads time 3/3, freq 2/2), ReLU on the sigmoid
//   contour map, then Conv2D 32->1, (7,3), "same", sigmoid -> note
//
// The 32x32x16 march spent 11.6 vector instructions per matrix instruction (ReLU + operand split of 16 C values per lane
// and tile, two accumulator sets to recombine, an im2col staging that split every contour value 8/3 times two rows at a
// time) and 28 % of its rows on chunk halo.  What changes:
//   * conv1: M = 2 blocks of 16 channels, N = 16 pixels, K = 2 k-steps of (4 frame taps x 8 bins, 7 used): lane
//     (n = lane & 15, g = lane >> 4) supplies frame tap 4 s + g of pixel n — one ds_read_b128 per plane, tile and k-step
//     from the wave's own ring of im2col rows (slot = the 8 bins 3 w - 2 .. 3 w + 5 of a pixel, f16 hi plane | lo plane).
//   * ONE accumulator per (tile, block) at scale 2^11: the three products of the split-precision form are
//     (hi_w 2^11) hi_a + (lo_w 2^11) hi_a + hi_w (lo_a 2^11) — the weights' hi parts are kept twice (scaled and not; f16 has
//     the range: |w| < 32 is checked at bp_create), every term is exact as before, and nothing is recombined on the VALU.
//   * ReLU + split: v = max(acc, 0); hi = rtz_f16(v 2^-11); the residual lo = rn_f16(v - 2^11 hi) comes from the MATRIX
//     pipe: D = v + Sel x hi with a constant selection fragment (-2^11 where the C layout's row meets the B layout's k) —
//     exact (f16 x f16 products, one f32 subtraction of a truncation from its source).  The lane's 8 channels of its
//     pixel ARE its B fragment of the tap projection (the order pack_note16 uses), as in onset_march16.hip.
//     3 vector operations per value instead of 6.
//   * projection (conv2 as 21 taps x 32 channels): 2 blocks of 16 rows, row 4 g + i of block mb = tap (dt = 4 mb + i,
//     dw = g): a lane holds the seven frame taps of ONE dw for its pixel, so conv2's vertical 7-tap sum is seven in-lane
//     additions into eight rotating accumulators (the row loop is written out eight times: every index is static, no
//     moves), and only the finished output row crosses lanes: two ds_bpermute bring dw = 0 of pixel w - 1 and dw = 2 of
//     w + 1 to the dw = 1 lanes.  out[t] = sum_dw (((((q0 + q1) + q2) + q3) + q4) + q5) + q6 — vertical first; the old
//     kernel summed horizontally first (same terms, another fp32 order).
//   * staging straight from the fp32 contour map: lane (slot = lane >> 1, half = lane & 1) fetches 4 consecutive bins
//     of one image row with ONE 16-byte load whose start is clamped into the row, splits them and puts the halves in
//     place with the v_perm_b32 that packs them (its selector carries the shift of a clamped start and the zeros of the
//     "same" padding), writes 8 + 8 bytes; one row per step, loaded two steps ahead of its first use.
//   * the frames of all (window, strip) pairs are laid end to end and cut into equal contiguous shares (the onset
//     march's cut): 8 % halo rows instead of 28 %.
// Roofline: f16 MFMA issue; 20 16x16x32 per 16 pixels (12 conv1 + 2 residual + 6 projection); HBM 182 KB read (through L2:
// the three strips of a window overlap by 2 pixels, shares by 6 rows) and 61 KB written per window.
#include <stdio.h>
#include <stdlib.h>

#include "bp_common.h"

namespace bp {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int kN16Waves = 4;    // independent waves per workgroup
constexpr int kN16Strips = 3;   // 32-pixel strips of a row, 30 inner pixels each
constexpr int kN16Ring = 8;     // image rows a wave keeps: r - 3 .. r + 3 in use, r + 4 being written
constexpr int kN16Row = 64;     // ring row stride in 16-byte units: hi plane (32 pixels), lo plane at kN16Lo
constexpr int kN16Lo = 32;
constexpr int kN16Frags = 18;   // pack_note16
#ifndef N16_OCC
#define N16_OCC 2
#endif
constexpr int kN16Occ = N16_OCC;  // resident workgroups per CU (x 4 waves: waves per SIMD)
static_assert(kN16Strips * 30 >= kFreqN, "strips cover a row");
static_assert(kN16Row % 16 == 0, "conflict-free ds_read_b128: rows differ by a multiple of 16 units");

struct Note16Params {
  const uint4* wfrag;    // pack_note16: 18 fragments x 64 lanes x (8 x f16)
  const float* wf32;     // bias1[32], ..., bias2 at [41]
  const float* contour;  // [n][172][264]
  float* out;            // [n][172][88]
  int n_ws;              // n_windows * kN16Strips (window, strip) pairs of 172 frames each
};

template <int V>
struct N16Phase {
  static constexpr int v = V;
};

template <bool WLO>
__global__ __launch_bounds__(64 * kN16Waves, kN16Occ) void note_march16_kernel(Note16Params p) {
  __shared__ __attribute__((aligned(16))) uint4 lds[kN16Waves][kN16Ring * kN16Row];  // [wave][row slot][hi | lo][pixel]

  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int g = lane >> 4, n = lane & 15;
  uint4* ring = lds[wave];
  uint2* ring2 = reinterpret_cast<uint2*>(ring);
#ifdef N16_PROF  // tools only: 100 MHz stamps of a few waves (entry, set-up done, per share: prologue done, march done)
  unsigned long long pt[8];
  int npt = 0;
  pt[npt++] = __builtin_amdgcn_s_memrealtime();
#endif

  // resident A operands (pack_note16): conv1 [kind H = hi 2^11, h = hi, L = lo 2^11][k-step][block], conv2 [kind][block]
  uint4 a1H[2][2], a1h[2][2], a1L[2][2], a2H[2], a2h[2], a2L[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      a1H[s][mb] = p.wfrag[(0 + 2 * s + mb) * 64 + lane];
      a1h[s][mb] = p.wfrag[(4 + 2 * s + mb) * 64 + lane];
      a1L[s][mb] = WLO ? p.wfrag[(8 + 2 * s + mb) * 64 + lane] : uint4{0u, 0u, 0u, 0u};
    }
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    a2H[mb] = p.wfrag[(12 + mb) * 64 + lane];
    a2h[mb] = p.wfrag[(14 + mb) * 64 + lane];
    a2L[mb] = WLO ? p.wfrag[(16 + mb) * 64 + lane] : uint4{0u, 0u, 0u, 0u};
  }
  f32x4 bias1s[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int r = 0; r < 4; ++r) bias1s[mb][r] = p.wf32[16 * mb + 4 * g + r] * kLoScale;
  const float bias2s = p.wf32[41] * kLoScale;
  // selection fragments of the residual: C row 4 G + i of block mb is the channel that B fragment element 4 mb + i of lane
  // group G carries, so A[m][k] = -2^11 at k = 8 (m >> 2) + 4 mb + (m & 3): lane (m = n, g) holds it iff g == m >> 2
  uint4 sel[2];
  {
    const uint32_t word = g == (n >> 2) ? (0xE800u << (16 * (n & 1))) : 0u;  // -2048 as f16
    const bool second = (n & 2) != 0;
    sel[0] = uint4{second ? 0u : word, second ? word : 0u, 0u, 0u};
    sel[1] = uint4{0u, 0u, second ? 0u : word, second ? word : 0u};
  }

  // the ring starts finite: the dummy frame tap multiplies whatever lies there by zero weights
  for (int i = lane; i < kN16Ring * kN16Row; i += 64) ring[i] = uint4{0u, 0u, 0u, 0u};

  // ring units this lane reads when the step's oldest row sits in slot k: row slot (k + g) & 7, pixel n (tile 1: + 16)
  int rdu[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) rdu[k] = ((k + g) & 7) * kN16Row + n;
  // staging: lane -> (slot = pixel of the strip, half = bins 0..3 / 4..7 of its slot)
  const int st_slot = lane >> 1, st_half = lane & 1;
  const int src_l4 = ((n + 15) & 15) * 4;         // ds_bpermute: pixel n - 1 of lane group 0
  const int src_r4 = (32 + ((n + 1) & 15)) * 4;   // pixel n + 1 of lane group 2

#ifdef N16_PROF
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  pt[npt++] = __builtin_amdgcn_s_memrealtime();
#endif
  // work distribution: onset_march16.hip's (XCD-aware order; equal contiguous shares of the frames of all (window, strip)
  // pairs; at exactly 8 waves per window the three strips of a frame range are marched by neighbouring waves)
  const int total_waves = gridDim.x * kN16Waves;
  const int half_n = (int)gridDim.x / 2, pq = (int)blockIdx.x % (half_n > 0 ? half_n : 1);
  const int lblock = (gridDim.x % 16 == 0) ? ((int)blockIdx.x / half_n) * half_n + (pq % 8) * (half_n / 8) + pq / 8 : (int)blockIdx.x;
  const int gw = lblock * kN16Waves + wave;
  const bool aligned = total_waves == 8 * (p.n_ws / kN16Strips);  // wave-uniform
  // cuts by COST, not by frames: a share costs its frames + 3 rows of halo per cut end (rows outside the window are skipped) and
  // ~2.3 rows per prologue: 68 + 3 | 65 + 6 | 39 + 3 + (18 + 6) + prologue | (21 + 3) + (39 + 3) + prologue = 71, 71, 68.3, 68.3
  // (profiles/r06_note_phases.md: with the onset march's cuts 64 | 65 | 43 the two-share waves ran 75 rows against 67)
  constexpr int kCut1 = 68, kCut2 = 133, kCut3 = 151;
  const int b8 = gw >> 3, j8 = gw & 7;
  const int64_t total = (int64_t)p.n_ws * kFrames;
  int64_t F0 = total * gw / total_waves;
  const int64_t F1 = total * (gw + 1) / total_waves;
#pragma unroll 1
  for (int pi = 0;; ++pi) {  // wave-uniform; no barriers
    int ws, T0, T1;
    if (aligned) {
      if (pi >= (j8 < 6 ? 1 : 2)) break;
      if (j8 < 6) {
        ws = kN16Strips * b8 + (j8 < 3 ? j8 : j8 - 3), T0 = j8 < 3 ? 0 : kCut1, T1 = j8 < 3 ? kCut1 : kCut2;
      } else if (pi == 0) {
        ws = kN16Strips * b8 + (j8 - 6), T0 = j8 == 6 ? kCut2 : kCut3, T1 = kFrames;
      } else {
        ws = kN16Strips * b8 + (j8 - 5), T0 = kCut2, T1 = j8 == 6 ? kCut3 : kFrames;
      }
    } else {
      if (F0 >= F1) break;
      ws = (int)(F0 / kFrames);
      T0 = (int)(F0 - (int64_t)ws * kFrames);
      T1 = F1 - (int64_t)ws * kFrames < kFrames ? (int)(F1 - (int64_t)ws * kFrames) : kFrames;
      F0 = (int64_t)(ws + 1) * kFrames;
    }
    const int b = ws / kN16Strips, strip = ws - b * kN16Strips;

    // this lane's two pixels of the strip (tile nt: strip pixel 16 nt + n)
    int w[2];
    bool wvalid[2], store_lane[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int pz = 16 * nt + n;
      w[nt] = strip * 30 - 1 + pz;
      wvalid[nt] = w[nt] >= 0 && w[nt] < kFreqN;
      store_lane[nt] = g == 1 && pz >= 1 && pz <= 30 && w[nt] < kFreqN;
    }
    float* owin = p.out + (int64_t)b * kPlaneN;
    // Staging source: this lane's 4 consecutive bins of an image row, fetched with ONE 16-byte load from a start clamped
    // into the row (c = clamp(bin0, 0, 260): every address lies inside the window's map); the halves are put in place —
    // shifted by c - bin0 where the clamp moved the start, zero where the bin is "same" padding — by the v_perm_b32 that
    // packs them (selector bytes 0x0c = zero), so the padding costs nothing.
    const float* cwin = p.contour + (int64_t)b * kPlaneC;
    const int st_bin0 = 3 * (strip * 30 - 1 + st_slot) - 2 + 4 * st_half;  // first of this lane's 4 bins (ONNX pads [3,2,3,2])
    const int st_c = st_bin0 < 0 ? 0 : (st_bin0 > kFreqC - 4 ? kFreqC - 4 : st_bin0);
    uint32_t st_s01 = 0, st_s23 = 0;  // out half j <- loaded half st_bin0 + j - st_c (bytes 2 h, 2 h + 1 of {h01, h23})
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int bin = st_bin0 + j, hsrc = bin - st_c;
      const uint32_t two = (bin >= 0 && bin < kFreqC) ? (uint32_t)((2 * hsrc) | ((2 * hsrc + 1) << 8)) : 0x0c0cu;
      if (j < 2)
        st_s01 |= two << (16 * j);
      else
        st_s23 |= two << (16 * (j - 2));
    }
    auto stage_issue = [&](int rho) {  // rows outside the window load a row of it and commit zeros
      const int rc = rho < 0 ? 0 : (rho > kFrames - 1 ? kFrames - 1 : rho);
      u32x4 v;  // 4-byte aligned: one global_load_dwordx4
      __builtin_memcpy(&v, cwin + rc * kFreqC + st_c, 16);
      return v;
    };
    auto stage_commit = [&](int slot, u32x4 v, int rho) {  // slot: ring slot of the row (static); rho: wave-uniform
      uint2 h, l;
      if (rho >= 0 && rho < kFrames) {
        uint32_t h01, l01, h23, l23;
        // (elements by index: __builtin_bit_cast of an ext-vector's .y / .z / .w reads element 0 with hipcc 7.2)
        split_f16x2(f32x2{__uint_as_float(v[0]), __uint_as_float(v[1])}, h01, l01);
        split_f16x2(f32x2{__uint_as_float(v[2]), __uint_as_float(v[3])}, h23, l23);
        h = uint2{__builtin_amdgcn_perm(h23, h01, st_s01), __builtin_amdgcn_perm(h23, h01, st_s23)};
        l = uint2{__builtin_amdgcn_perm(l23, l01, st_s01), __builtin_amdgcn_perm(l23, l01, st_s23)};
      } else {  // conv1's zero padding
        h = l = uint2{0u, 0u};
      }
      ring2[(slot * kN16Row + st_slot) * 2 + st_half] = h;
      ring2[(slot * kN16Row + kN16Lo + st_slot) * 2 + st_half] = l;
    };
    auto ring_fence = [] {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    // ---- the march: conv1 rows r = T0 - 3 .. T1 + 2; step i = r - r_first keeps image row r - 3 + j in slot (i + j) & 7
    const int r_first = T0 - 3, r_last = T1 + 2;
    u32x4 pend;  // the image row loaded one step ago, committed at the end of this step
    {
      u32x4 v[7];
#pragma unroll
      for (int j = 0; j < 7; ++j) v[j] = stage_issue(r_first - 3 + j);
      pend = stage_issue(r_first + 4);
#pragma unroll
      for (int j = 0; j < 7; ++j) stage_commit(j, v[j], r_first - 3 + j);
      ring_fence();
    }
    float V[2][8];  // V[nt][t & 7 relative]: the open output rows of this lane's (pixel, dw)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int k = 0; k < 8; ++k) V[nt][k] = 0.0f;

    // the fragments of k-step 0 are read one step ahead (at the end of the previous step, under its epilogue)
    f16x8 bh0[2], bl0[2];
    auto read_k = [&](int k, f16x8 (&bh)[2], f16x8 (&bl)[2]) {  // k: static slot index of the fragment's first row
      const int u = rdu[k & 7];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        bh[nt] = __builtin_bit_cast(f16x8, ring[u + 16 * nt]);
        bl[nt] = __builtin_bit_cast(f16x8, ring[u + 16 * nt + kN16Lo]);
      }
    };
    read_k(0, bh0, bl0);
#ifdef N16_PROF
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (npt < 7) pt[npt++] = __builtin_amdgcn_s_memrealtime();
#endif
    const float sig_k = -1.44269504088896341f * kLoUnscale;  // sigmoid((y + b) 2^-11) = 1 / (1 + exp2((y + b) sig_k))
    const int st_w0 = w[0], st_w1 = w[1];

    // one conv1 row; PH = step & 7 (static: every ring slot and accumulator index below is a constant).  Order of a step:
    // fence | k-step 1's reads | k-step 0's matrix work | commit of the staged row + next row's load (vector work under the
    // matrix pipe) | k-step 1's matrix work | the NEXT step's k-step 0 reads | epilogue — every LDS read has a k-step of
    // matrix work or the epilogue between its issue and its first use (two waves per SIMD cover little else).
    auto step = [&](auto PH, int r) {
      constexpr int ph = decltype(PH)::v;
      const bool live = r >= 0 && r < kFrames;  // wave-uniform; a conv1 row outside the window is conv2's zero padding
      f32x4 acc[2][2];
      f16x8 bh1[2], bl1[2];
      auto conv1_kstep = [&](int s, const f16x8 (&bh)[2], const f16x8 (&bl)[2]) {
        // passes over the four (tile, block) accumulators: dependent instructions sit 4 apart
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
            acc[nt][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a1H[s][mb]), bh[nt], acc[nt][mb], 0, 0, 0);
        if (WLO) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
              acc[nt][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a1L[s][mb]), bh[nt], acc[nt][mb], 0, 0, 0);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
            acc[nt][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a1h[s][mb]), bl[nt], acc[nt][mb], 0, 0, 0);
      };
      ring_fence();  // k-step 1 touches the row committed in the middle of the last step
      if (!live) {  // at most six steps per share (the window's first and last rows)
        stage_commit((ph + 7) & 7, pend, r + 4);
        pend = stage_issue(r + 5);
        read_k(ph + 1, bh0, bl0);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) V[nt][(ph + 3) & 7] = 0.0f;
      } else {
        read_k(ph + 4, bh1, bl1);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) acc[nt][mb] = bias1s[mb];
        conv1_kstep(0, bh0, bl0);
        stage_commit((ph + 7) & 7, pend, r + 4);  // image row r + 4: first read by the next step's k-step 1
        pend = stage_issue(r + 5);
        conv1_kstep(1, bh1, bl1);
        read_k(ph + 1, bh0, bl0);  // rows r - 2 .. r + 1: the next step's k-step 0
        // ReLU, split, tap projection — written tile-interleaved, stage by stage: the hi parts of both tiles, their residual
        // matrix instructions, the projection's two hi products (which need no lo part) while the residuals complete, the
        // lo parts, the third product
        f32x4 P[2][2];  // [tile][block]: row i = frame tap 4 mb + i of dw = g, at scale 2^11
        f32x4 v[2][2];
        f16x8 b2h[2], b2l[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          uint32_t hw[4];
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[nt][mb][i] = relu_f32(acc[nt][mb][i]);
            hw[2 * mb] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(v[nt][mb][0] * kLoUnscale, v[nt][mb][1] * kLoUnscale));
            hw[2 * mb + 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(v[nt][mb][2] * kLoUnscale, v[nt][mb][3] * kLoUnscale));
          }
          b2h[nt] = __builtin_bit_cast(f16x8, uint4{hw[0], hw[1], hw[2], hw[3]});
        }
#ifndef N16_RESID_VALU
        f32x4 d[2][2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
            d[nt][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, sel[mb]), b2h[nt], v[nt][mb], 0, 0, 0);
#endif
        const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
            P[nt][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a2H[mb]), b2h[nt], zero4, 0, 0, 0);
        if (WLO) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
              P[nt][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a2L[mb]), b2h[nt], P[nt][mb], 0, 0, 0);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          uint32_t lw[4];
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) {
#ifdef N16_RESID_VALU
            // the residual on the vector pipe (A/B): v - 2^11 float(hi), exact
            const f16x8 hh = b2h[nt];
            f32x4 dd;
#pragma unroll
            for (int i = 0; i < 4; ++i) dd[i] = __builtin_fmaf((float)hh[4 * mb + i], -kLoScale, v[nt][mb][i]);
            const f16x2 d01 = {(_Float16)dd[0], (_Float16)dd[1]}, d23 = {(_Float16)dd[2], (_Float16)dd[3]};
#else
            const f16x2 d01 = {(_Float16)d[nt][mb][0], (_Float16)d[nt][mb][1]}, d23 = {(_Float16)d[nt][mb][2], (_Float16)d[nt][mb][3]};
#endif
            lw[2 * mb] = __builtin_bit_cast(uint32_t, d01);
            lw[2 * mb + 1] = __builtin_bit_cast(uint32_t, d23);
          }
          b2l[nt] = __builtin_bit_cast(f16x8, uint4{lw[0], lw[1], lw[2], lw[3]});
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
            P[nt][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a2h[mb]), b2l[nt], P[nt][mb], 0, 0, 0);
        // conv2's vertical sum: frame tap dt of conv1 row r belongs to output row r + 3 - dt
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          V[nt][(ph + 3) & 7] = P[nt][0][0];
#pragma unroll
          for (int dt = 1; dt < 7; ++dt) V[nt][(ph + 3 - dt) & 7] += P[nt][dt >> 2][dt & 3];
        }
      }
      const int t = r - 3;  // the output row that has all seven frame taps now
      if (t >= T0 && t < T1) {  // wave-uniform
        // pixels outside the row are conv2's zero padding; out[w] = (S_dw0[w - 1] + S_dw1[w]) + S_dw2[w + 1]
        const float s0 = wvalid[0] ? V[0][(ph + 5) & 7] : 0.0f, s1 = wvalid[1] ? V[1][(ph + 5) & 7] : 0.0f;
        const float dl1 = n == 15 ? s0 : s1, dr0 = n == 0 ? s1 : s0;  // the two tiles hand over their edge pixels
        const float l0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_l4, __builtin_bit_cast(int, s0)));
        const float l1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_l4, __builtin_bit_cast(int, dl1)));
        const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_r4, __builtin_bit_cast(int, dr0)));
        const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_r4, __builtin_bit_cast(int, s1)));
        const float y0 = ((l0 + s0) + r0) + bias2s, y1 = ((l1 + s1) + r1) + bias2s;
        const float o0 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y0 * sig_k));
        const float o1 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y1 * sig_k));
        float* orow = owin + t * kFreqN;  // wave-uniform base; 32-bit lane offsets
        if (store_lane[0]) orow[st_w0] = o0;
        if (store_lane[1]) orow[st_w1] = o1;
      }
    };

    int r = r_first;
#pragma unroll 1
    for (;;) {
      step(N16Phase<0>{}, r);
      if (++r > r_last) break;
      step(N16Phase<1>{}, r);
      if (++r > r_last) break;
      step(N16Phase<2>{}, r);
      if (++r > r_last) break;
      step(N16Phase<3>{}, r);
      if (++r > r_last) break;
      step(N16Phase<4>{}, r);
      if (++r > r_last) break;
      step(N16Phase<5>{}, r);
      if (++r > r_last) break;
      step(N16Phase<6>{}, r);
      if (++r > r_last) break;
      step(N16Phase<7>{}, r);
      if (++r > r_last) break;
    }
    ring_fence();  // the next piece's prologue overwrites the ring
#ifdef N16_PROF
    if (npt < 7) pt[npt++] = __builtin_amdgcn_s_memrealtime();
#endif
  }
#ifdef N16_PROF
  pt[npt++] = __builtin_amdgcn_s_memrealtime();
  if (lane == 0 && p.n_ws >= 768 && (gw < 8 || gw == 1003 || gw == 1006 || gw == 2040 || gw == 2047))
    printf("N16P gw %d block %d t0 %llu setup %llu | %llu %llu %llu %llu | end %llu\n", gw, (int)blockIdx.x, pt[0] % 100000000ull, pt[1] - pt[0],
           npt > 3 ? pt[2] - pt[0] : 0ull, npt > 3 ? pt[3] - pt[0] : 0ull, npt > 5 ? pt[4] - pt[0] : 0ull, npt > 5 ? pt[5] - pt[0] : 0ull, pt[npt - 1] - pt[0]);
#endif
}

void launch_note_march16(const float* contour, const void* wfrag, const float* wf32, float* note, int n_windows, int n_cu,
                         bool weights_have_lo, hipStream_t stream) {
  // two resident workgroups per CU, persistent; small batches: fewer waves, at least kMinFrames frames each
  Note16Params p{static_cast<const uint4*>(wfrag), wf32, contour, note, n_windows * kN16Strips};
  if (p.n_ws <= 0) return;
  constexpr int kMinFrames = 12;
  const int64_t waves = ((int64_t)p.n_ws * kFrames + kMinFrames - 1) / kMinFrames;
  int grid = (int)((waves + kN16Waves - 1) / kN16Waves);
  if (grid > kN16Occ * n_cu) grid = kN16Occ * n_cu;
  if (weights_have_lo)
    hipLaunchKernelGGL(note_march16_kernel<true>, dim3(grid), dim3(64 * kN16Waves), 0, stream, p);
  else
    hipLaunchKernelGGL(note_march16_kernel<false>, dim3(grid), dim3(64 * kN16Waves), 0, stream, p);
}

}  // namespace bp
