// Onset branch, wave-private march on v_mfma_f32_16x16x32_f16 (round 4; the default.  onset_march.hip keeps the
// 32x32x16 form behind BP_ONSET=march32, conv_branch.hip the workgroup kernel behind BP_ONSET=ring / the fp8 mode).
//
//   basic_pitch/models.py:295-318: Conv2D 8->32, 5x5, strides (1,3), "same", folded BN, ReLU on the harmonic stack
//   (nn.py:69-88), Concatenate([note, features]) (305), Conv2D 33->1, 3x3, "same", sigmoid -> onset
//
// Why another instruction shape: this chip sustains ~1.46 PFLOP/s on 32x32x16 with the weights in registers and the
// activations from LDS (the round-3 march) and ~1.70 PFLOP/s on 16x16x32 in the same arrangement when every activation
// fragment read from LDS feeds TWO 16-row weight blocks (profiles/r04_ubench_rega.md) — the FLOPs per tile row are the
// same (45 x 32x32x16 = 90 x 16x16x32: conv1 pads its 25 taps to 28 instead of 26, the tap projection is one k-step of 32
// channels instead of two of 16).
//
// Decomposition as in onset_march.hip: a work item is (window, time chunk, 32-pixel strip), belongs to ONE wave, no
// workgroup barrier; the wave keeps a 6-row ring of its strip's stack image in LDS (98 slots = stack bins x 8 harmonic
// channels, f16 hi plane | lo plane) gathered straight from zp a row ahead; conv2's vertical 3-tap sum stays in registers.
// What changes with the instruction:
//   * conv1: M = 2 blocks of 16 channels, N = 2 tiles of 16 pixels, K = 7 k-steps of (4 taps x 8 channels).  Lane
//     (n = lane & 15, g = lane >> 4) supplies tap TAP[s][g] of pixel 16 nt + n: ONE ds_read_b128 per plane and tile (slot
//     3 p + dw of image row dt), 4 reads -> 12 matrix instructions per k-step.  The weights (7 x 2 x {hi, lo} fragments =
//     112 VGPRs) never leave the registers.  Tap order: k-steps 0..4 = image row dt = s with dw = {0, 3, 1, 4}[g], k-step
//     5 = dw 2 of rows g, k-step 6 = (4, 2) + three zero-weight dummies.  With the ring's row stride a multiple of 16
//     slots (256 B = all 64 banks) a lane pair (g even, g odd) of one ds_read_b128 service group is conflict-free iff
//     dw_odd - dw_even is 0 or 3 (enumerated against the hardware's lane groups {0-3, 12-15, 20-27}, ...): every k-step here.
//   * the C layout hands lane (n, g) channels 4 g + r of block 0 and 16 + 4 g + r of block 1 for ITS pixel: after ReLU
//     and the hi / lo split those 8 values ARE the lane's B fragment of the tap projection (K index 8 g + j <-> channel
//     4 g + j or 16 + 4 g + (j - 4); the packed conv2 weights use the same order) — no exchange.
//   * projection: C row 4 dt + dw = tap (dt, dw): lane group g holds the three dw taps of frame tap dt = g (group 3:
//     zero rows).  Horizontal sum = row shifts by DPP (the strip's two tiles exchange their edge pixels by a row rotate),
//     vertical sum = one 16-lane ds_bpermute per tile and row: C <- (g == 0 ? q : X + q), group 2 ends with
//     (q0(r - 2) + q1(r - 1)) + q2(r) — the summation order of the other onset kernels.
// Roofline: f16 MFMA issue; 90 16x16x32 per 30 output pixels; + 9 % rows of chunk halo.
#include <stdlib.h>

#include "bp_common.h"

namespace bp {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int kO16Waves = 4;  // independent waves per workgroup
constexpr int kO16Strips = 3;                  // 32-pixel strips of a row, 30 inner pixels each
constexpr int kO16Ring = 6;                    // image rows a wave keeps: r - 2 .. r + 2 in use, r + 3 being written
constexpr int kO16Slots = 98;                  // stack bins a strip's 32 pixels read: 3 * 31 + 5
constexpr int kO16Row = 208;                   // ring row stride in 16-byte slots: hi plane, lo plane at kO16Lo; 13 x 16
constexpr int kO16Lo = 104;
constexpr int kO16KS = kOnset16KSteps;         // 7 (bp_common.h: the tap table is shared with the host packer)
#ifndef BP_ONSET16_PF
#define BP_ONSET16_PF 1
#endif
constexpr int kO16Pf = BP_ONSET16_PF;          // k-steps of image fragments read ahead of the matrix instructions
static_assert(kO16Strips * 30 >= kFreqN, "strips cover a row");
static_assert(kO16Lo >= kO16Slots && 2 * kO16Lo <= kO16Row && kO16Row % 16 == 0, "ring geometry");

struct Onset16Params {
  const uint4* wfrag;   // pack_onset16: [A1 hi: 14 x 64][A1 lo: 14 x 64][A2 hi: 64][A2 lo: 64] x (8 x f16)
  const float* wf32;    // bias1[32], the note channel's 3x3 taps at [32 + 3 dt + dw], bias2 at [41]
  const uint32_t* zp;   // [n][kZRowsP][kZRow] packed (hi | lo << 16) words, zero padded (bp_common.h)
  const float* note;    // [n][172][88]
  float* out;           // [n][172][88]
  int n_ws;             // n_windows * kO16Strips (window, strip) pairs of 172 frames each
};

template <bool WLO>
__global__ __launch_bounds__(64 * kO16Waves, 2) void onset_march16_kernel(Onset16Params p) {
  __shared__ __attribute__((aligned(16))) uint4 lds[kO16Waves][kO16Ring * kO16Row];  // [wave][row slot][hi | lo][slot]

  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int g = lane >> 4, n = lane & 15;
  uint4* ring = lds[wave];

  // resident A operands and constants
  uint4 a1h[kO16KS][2], a1l[WLO ? kO16KS : 1][2], a2h, a2l;
#pragma unroll
  for (int s = 0; s < kO16KS; ++s)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      a1h[s][mb] = p.wfrag[(2 * s + mb) * 64 + lane];
      if (WLO) a1l[WLO ? s : 0][mb] = p.wfrag[(2 * kO16KS + 2 * s + mb) * 64 + lane];
    }
  a2h = p.wfrag[4 * kO16KS * 64 + lane];
  a2l = p.wfrag[(4 * kO16KS + 1) * 64 + lane];
  f32x4 bias1[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int r = 0; r < 4; ++r) bias1[mb][r] = p.wf32[16 * mb + 4 * g + r];
  // the 3 x 3 taps of the note channel (concat channel 0) for the frame tap this lane group owns: dt = g (group 3: none)
  float extra[3];
#pragma unroll
  for (int dw = 0; dw < 3; ++dw) extra[dw] = g < 3 ? p.wf32[32 + 3 * g + dw] : 0.0f;
  const float bias2 = p.wf32[41];

  // the ring starts finite: the dummy taps and the halo pixels multiply whatever lies there by zero weights
  for (int i = lane; i < kO16Ring * kO16Row; i += 64) ring[i] = uint4{0u, 0u, 0u, 0u};

  // lane shifts inside a 16-lane row (one helper call per value: update_dpp on elements of an ext-vector inside an
  // unrolled loop is emitted once and reused by hipcc 7.2)
  auto row_shr1 = [](float old, float v) {  // value of lane - 1; lane 0 of a row keeps `old`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                  0x111, 0xf, 0xf, false));
  };
  auto row_shl1 = [](float old, float v) {  // value of lane + 1; lane 15 of a row keeps `old`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                  0x101, 0xf, 0xf, false));
  };
  auto row_ror1 = [](float v) {  // lane 0 of a row gets the row's lane 15
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
  };
  auto row_ror15 = [](float v) {  // lane 15 of a row gets the row's lane 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x12F, 0xf, 0xf, false));
  };
  const int src_lane4 = ((lane - 16) & 63) * 4;  // ds_bpermute address: lane group g reads group g - 1

  // image slot this lane reads at k-steps 0..4 relative to the row base: pixel n of tile 0, dw = {0, 3, 1, 4}[g]
  const int at04 = 3 * n + onset16_dw(0, g);
  const int at56 = 3 * n + 2;

  const int total_waves = gridDim.x * kO16Waves;
  // XCD-aware order: workgroups go to the 8 XCDs round-robin (blockIdx % 8) and each XCD has its own L2; consecutive
  // pieces of work — the strips of ONE window, which read overlapping parts of the same zp rows — are given to
  // workgroups of the same XCD, so a row is fetched from HBM by one L2 instead of by up to eight
  const int half_n = (int)gridDim.x / 2, pq = (int)blockIdx.x % (half_n > 0 ? half_n : 1);
  const int lblock = (gridDim.x % 16 == 0) ? ((int)blockIdx.x / half_n) * half_n + (pq % 8) * (half_n / 8) + pq / 8 : (int)blockIdx.x;
  // Work = the frames of all (window, strip) pairs; every wave takes an equal share of them as at most two marches
  // (round 4: three tasks of an eighth of a window-strip each — three prologues and 3 x 2 warm-up rows per wave).
  //  * exactly 8 waves per window (full batches: 2048 waves, 256 windows; 3 strips x 172 frames = 8 x 64.5): waves 0-2
  //    of a window march frames 0..63 of strips 0, 1, 2, waves 3-5 frames 64..128, wave 6 the rest of strip 0 and half the
  //    rest of strip 1, wave 7 the other half and the rest of strip 2 — 1.25 pieces per wave, and the three strips of a
  //    frame range, which gather from the SAME zp rows, are marched at the same time by neighbouring waves of one
  //    workgroup (laid end to end instead, the strips of a window were marched at different times: 183 MB of HBM reads
  //    per launch where this order needs 121 — the time is the same);
  //  * any other wave count: the pairs laid end to end, wave g of G takes the g-th G-th.
  const int gw = lblock * kO16Waves + wave;
  const bool aligned = total_waves == 8 * (p.n_ws / kO16Strips);  // wave-uniform
  constexpr int kCut1 = 64, kCut2 = 129, kCut3 = 150;             // 64 | 65 | 43 = 21 + 22
  const int b8 = gw >> 3, j8 = gw & 7;
  const int64_t total = (int64_t)p.n_ws * kFrames;
  int64_t F0 = total * gw / total_waves;  // the end-to-end order's share (a share may span several pairs)
  const int64_t F1 = total * (gw + 1) / total_waves;
#pragma unroll 1
  for (int pi = 0;; ++pi) {  // wave-uniform; no barriers
    int ws, T0, T1;
    if (aligned) {
      if (pi >= (j8 < 6 ? 1 : 2)) break;
      if (j8 < 6) {
        ws = kO16Strips * b8 + (j8 < 3 ? j8 : j8 - 3), T0 = j8 < 3 ? 0 : kCut1, T1 = j8 < 3 ? kCut1 : kCut2;
      } else if (pi == 0) {
        ws = kO16Strips * b8 + (j8 - 6), T0 = j8 == 6 ? kCut2 : kCut3, T1 = kFrames;
      } else {
        ws = kO16Strips * b8 + (j8 - 5), T0 = kCut2, T1 = j8 == 6 ? kCut3 : kFrames;
      }
    } else {
      if (F0 >= F1) break;
      ws = (int)(F0 / kFrames);
      T0 = (int)(F0 - (int64_t)ws * kFrames);
      T1 = F1 - (int64_t)ws * kFrames < kFrames ? (int)(F1 - (int64_t)ws * kFrames) : kFrames;
      F0 = (int64_t)(ws + 1) * kFrames;
    }
    const int b = ws / kO16Strips, strip = ws - b * kO16Strips;

    // this lane's two pixels of the strip (tile nt: strip pixel 16 nt + n), and the stack bin of image slot 0
    int w[2], wc[2];
    bool wvalid[2], store_lane[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int pz = 16 * nt + n;
      w[nt] = strip * 30 - 1 + pz;
      wvalid[nt] = w[nt] >= 0 && w[nt] < kFreqN;
      wc[nt] = w[nt] < 0 ? 0 : (w[nt] >= kFreqN ? kFreqN - 1 : w[nt]);
      store_lane[nt] = g == 2 && pz >= 1 && pz <= 30 && w[nt] < kFreqN;
    }
    const int f0 = 3 * (strip * 30 - 1) - 1;  // pixel w reads stack bins 3 w - 1 .. 3 w + 3 (ONNX pads [2,1,2,1])
    const uint32_t* zwin = p.zp + (int64_t)b * kZWin + kZPadL;
    const float* nwin = p.note + (int64_t)b * kPlaneN;
    float* owin = p.out + (int64_t)b * kPlaneN;
    // slots of this lane: q = lane and q = 64 + lane (lanes >= 34 re-read slot 97 and write nothing)
    const int qb = 64 + lane < kO16Slots ? 64 + lane : kO16Slots - 1;
    const int fa = f0 + lane, fb = f0 + qb;
    const bool fa_ok = fa >= 0 && fa < kFreqC, fb_ok = fb >= 0 && fb < kFreqC;  // outside: "same" padding of the stack

    // ---- staging of image row t: issue 16 loads; commit = pack hi / lo, zero the padding, four 16-byte LDS stores
    auto stage_issue = [&](int t, uint32_t (&u)[16]) {
      // zp rows -1 and 172 are zero (bp_common.h); frames further outside read the zero row -1
      const uint32_t* src = zwin + (int64_t)(((t >= -1 && t <= kFrames) ? t : -1) + 1) * kZRow;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        u[c] = src[fa + harm_shift(c)];
        u[8 + c] = src[fb + harm_shift(c)];
      }
    };
    // pack: word c of a slot is (hi | lo << 16) of channel c; the image wants 8 hi halves and 8 lo halves.  One v_perm_b32
    // per output word (bytes {u0.b0, u0.b1, u1.b0, u1.b1} / {u0.b2, u0.b3, u1.b2, u1.b3}); a lane whose stack bin is
    // "same" padding carries the selector 0x0c0c0c0c = four zero bytes, so the mask costs nothing (was and / shift / or +
    // 16 selects per row)
    const uint32_t sel_ha = fa_ok ? 0x05040100u : 0x0c0c0c0cu, sel_la = fa_ok ? 0x07060302u : 0x0c0c0c0cu;
    const uint32_t sel_hb = fb_ok ? 0x05040100u : 0x0c0c0c0cu, sel_lb = fb_ok ? 0x07060302u : 0x0c0c0c0cu;
    auto pack = [](const uint32_t* u, uint32_t sel_h, uint32_t sel_l, uint4& vh, uint4& vl) {
      vh.x = __builtin_amdgcn_perm(u[1], u[0], sel_h);
      vh.y = __builtin_amdgcn_perm(u[3], u[2], sel_h);
      vh.z = __builtin_amdgcn_perm(u[5], u[4], sel_h);
      vh.w = __builtin_amdgcn_perm(u[7], u[6], sel_h);
      vl.x = __builtin_amdgcn_perm(u[1], u[0], sel_l);
      vl.y = __builtin_amdgcn_perm(u[3], u[2], sel_l);
      vl.z = __builtin_amdgcn_perm(u[5], u[4], sel_l);
      vl.w = __builtin_amdgcn_perm(u[7], u[6], sel_l);
    };
    auto stage_commit = [&](int slot, const uint32_t (&u)[16]) {  // slot: ring slot of the row (scalar)
      uint4 vh, vl;
      pack(u, sel_ha, sel_la, vh, vl);
      ring[slot * kO16Row + lane] = vh;
      ring[slot * kO16Row + kO16Lo + lane] = vl;
      pack(u + 8, sel_hb, sel_lb, vh, vl);
      if (64 + lane < kO16Slots) {
        ring[slot * kO16Row + 64 + lane] = vh;
        ring[slot * kO16Row + kO16Lo + 64 + lane] = vl;
      }
    };
    // The ring is written lane-private and read across lanes: a wavefront fence orders the two.  Not right behind the commit
    // (there it exposes the LDS write latency once per row): the row committed at the end of step r is image row (r + 1) + 2
    // of the next step and first read by that step's k-step 4 — the fence sits in front of that read, three k-steps of matrix
    // work behind the writes.
    auto ring_fence = [] {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto note_at = [&](int row, int nt) {  // unconditional load from a clamped address, masked where it is used
      const int rc = row < 0 ? 0 : (row > kFrames - 1 ? kFrames - 1 : row);
      return nwin[rc * kFreqN + wc[nt]];
    };

    // ---- one conv1 row r: q[nt] = horizontal-summed projection of frame tap dt = g for the lane's pixel of tile nt, the
    // note channel's taps included.  slot_m2 = ring slot of image row r - 2
    auto tile = [&](int slot_m2, const float (&note_c)[2], float (&q)[2]) {
      int rb[5];  // ring offsets (in uint4 units) of the hi plane of image rows r - 2 + d
#pragma unroll
      for (int d = 0, s = slot_m2; d < 5; ++d) {
        rb[d] = s * kO16Row;
        s = s + 1 == kO16Ring ? 0 : s + 1;
      }
      const int rbg = g == 0 ? rb[0] : (g == 1 ? rb[1] : (g == 2 ? rb[2] : rb[3]));  // k-step 5: image row dt = g
      f32x4 acc[2][2], accc[2][2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          acc[nt][mb] = bias1[mb];
          accc[nt][mb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
      f16x8 bhf[kO16KS][2], blf[kO16KS][2];
      auto issue = [&](int s) {
        const int at = s < 5 ? rb[s < 5 ? s : 0] + at04 : (s == 5 ? rbg + at56 : rb[4] + at56);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          bhf[s][nt] = __builtin_bit_cast(f16x8, ring[at + 48 * nt]);
          blf[s][nt] = __builtin_bit_cast(f16x8, ring[at + 48 * nt + kO16Lo]);
        }
      };
#pragma unroll
      for (int s = 0; s < kO16Pf; ++s) issue(s);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < kO16KS; ++s) {
        if (s + kO16Pf == 4) ring_fence();  // the next read touches the row committed at the end of the last step
        if (s + kO16Pf < kO16KS) issue(s + kO16Pf);
        __builtin_amdgcn_sched_barrier(0);
        // three passes over the four (tile, block) accumulator pairs: dependent instructions sit 4 and 8 apart
        if (WLO) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
              accc[nt][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a1l[WLO ? s : 0][mb]), bhf[s][nt],
                                                                    accc[nt][mb], 0, 0, 0);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
            acc[nt][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a1h[s][mb]), bhf[s][nt], acc[nt][mb],
                                                                 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
            accc[nt][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a1h[s][mb]), blf[s][nt],
                                                                  accc[nt][mb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ReLU, split (the lane's 8 channels of its pixel = its B fragment of the projection), tap projection
      float p3[2][3];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        uint32_t b2hw[4], b2lw[4];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            f32x2 v = {__builtin_fmaf(accc[nt][mb][r], kLoUnscale, acc[nt][mb][r]), __builtin_fmaf(accc[nt][mb][r + 1], kLoUnscale, acc[nt][mb][r + 1])};  // plain, not v_pk_fma_f32 (3.5 x the cost beside MFMAs)
            v.x = fmaxf(v.x, 0.0f);
            v.y = fmaxf(v.y, 0.0f);
            split_f16x2(v, b2hw[2 * mb + (r >> 1)], b2lw[2 * mb + (r >> 1)]);
          }
        const f16x8 b2h = __builtin_bit_cast(f16x8, uint4{b2hw[0], b2hw[1], b2hw[2], b2hw[3]});
        const f16x8 b2l = __builtin_bit_cast(f16x8, uint4{b2lw[0], b2lw[1], b2lw[2], b2lw[3]});
        const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
        const f32x4 pp = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a2h), b2h, zero4, 0, 0, 0);
        f32x4 ppc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a2l), b2h, zero4, 0, 0, 0);
        ppc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a2h), b2l, ppc, 0, 0, 0);
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) p3[nt][dw] = pp[dw] + ppc[dw] * kLoUnscale;
      }
      // Q[dt][w] = (P[dt,0][w-1] + P[dt,1][w]) + P[dt,2][w+1]: row shifts; the two tiles exchange their edge pixels; pixels
      // outside the row are conv2's zero padding; the note channel's taps on the VALU
      const float p0a = wvalid[0] ? p3[0][0] : 0.0f, p0b = wvalid[1] ? p3[1][0] : 0.0f;
      const float p2a = wvalid[0] ? p3[0][2] : 0.0f, p2b = wvalid[1] ? p3[1][2] : 0.0f;
      const float na = wvalid[0] ? note_c[0] : 0.0f, nb = wvalid[1] ? note_c[1] : 0.0f;
      const float l0 = row_shr1(0.0f, p0a), l1 = row_shr1(row_ror1(p0a), p0b);
      const float r0 = row_shl1(row_ror15(p2b), p2a), r1 = row_shl1(0.0f, p2b);
      const float nl0 = row_shr1(0.0f, na), nl1 = row_shr1(row_ror1(na), nb);
      const float nr0 = row_shl1(row_ror15(nb), na), nr1 = row_shl1(0.0f, nb);
      float qa = (l0 + p3[0][1]) + r0;
      qa += (nl0 * extra[0] + na * extra[1]) + nr0 * extra[2];
      float qb2 = (l1 + p3[1][1]) + r1;
      qb2 += (nl1 * extra[0] + nb * extra[1]) + nr1 * extra[2];
      q[0] = qa;
      q[1] = qb2;
    };

    // ---- the march: conv1 rows r = T0 - 1 .. T1
    const int r_first = T0 - 1;
    {  // prologue: image rows r_first - 2 .. r_first + 2 into ring slots 0 .. 4
      uint32_t ua[16], ub[16], uc[16];
      stage_issue(r_first - 2, ua);
      stage_issue(r_first - 1, ub);
      stage_issue(r_first, uc);
      stage_commit(0, ua);
      stage_commit(1, ub);
      stage_commit(2, uc);
      stage_issue(r_first + 1, ua);
      stage_issue(r_first + 2, ub);
      stage_commit(3, ua);
      stage_commit(4, ub);
      ring_fence();
    }
    float note_nx[2] = {note_at(r_first, 0), note_at(r_first, 1)};
    float C[2] = {0.0f, 0.0f};  // the vertical sum in flight: group 0: q0(r), group 1: q0(r - 1) + q1(r)
    int slot_m2 = 0;            // ring slot of image row r - 2
#pragma unroll 1
    for (int r = r_first; r <= T1; ++r) {
      uint32_t st[16];
      int slot_p3 = slot_m2 + 5;  // row r + 3 takes the slot of row r - 3
      slot_p3 = slot_p3 >= kO16Ring ? slot_p3 - kO16Ring : slot_p3;
      stage_issue(r + 3, st);
      const float note_c[2] = {note_nx[0], note_nx[1]};
      note_nx[0] = note_at(r + 1, 0);
      note_nx[1] = note_at(r + 1, 1);
      // the carries of the previous row move one lane group up while this row's matrix work runs
      float X[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        X[nt] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane4, __builtin_bit_cast(int, C[nt])));
      float q[2] = {0.0f, 0.0f};
      // a conv1 row outside the window is conv2's zero padding (the note channel's row too)
      if (r >= 0 && r < kFrames) {
        tile(slot_m2, note_c, q);
      }
      const int t = r - 1;  // the output row group 2 finishes now: (q0(r - 2) + q1(r - 1)) + q2(r)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const float y = X[nt] + q[nt];
        C[nt] = g == 0 ? q[nt] : y;
        if (store_lane[nt] && t >= T0 && t < T1) owin[t * kFreqN + w[nt]] = sigmoidf_fast(y + bias2);
      }
      stage_commit(slot_p3, st);
      slot_m2 = slot_m2 + 1 == kO16Ring ? 0 : slot_m2 + 1;
    }
  }
}

void launch_onset_march16(const uint32_t* zp, const float* note, const void* wfrag, const float* wf32, float* onset,
                          int n_windows, int n_cu, bool weights_have_lo, hipStream_t stream) {
  // two resident workgroups per CU (LDS), persistent; small batches: fewer waves, at least kMinFrames frames each
  Onset16Params p{static_cast<const uint4*>(wfrag), wf32, zp, note, onset, n_windows * kO16Strips};
  if (p.n_ws <= 0) return;
  constexpr int kMinFrames = 6;
  const int64_t waves = ((int64_t)p.n_ws * kFrames + kMinFrames - 1) / kMinFrames;
  int grid = (int)((waves + kO16Waves - 1) / kO16Waves);
  if (grid > 2 * n_cu) grid = 2 * n_cu;
  if (weights_have_lo)
    hipLaunchKernelGGL(onset_march16_kernel<true>, dim3(grid), dim3(64 * kO16Waves), 0, stream, p);
  else
    hipLaunchKernelGGL(onset_march16_kernel<false>, dim3(grid), dim3(64 * kO16Waves), 0, stream, p);
}

}  // namespace bp
