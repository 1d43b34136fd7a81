// Onset branch, wave-private march on 32x32x16 (round 3; since round 4 behind BP_ONSET=march32: the default is its
// 16x16x32 form, onset_march16.hip.  The workgroup kernel of conv_branch.hip stays for the fp8-correction mode and as the
// A/B reference, BP_ONSET=ring).
//
//   basic_pitch/models.py:295-318: Conv2D 8->32, 5x5, strides (1,3), "same", folded BN, ReLU on the harmonic stack
//   (nn.py:69-88), Concatenate([note, features]) (305), Conv2D 33->1, 3x3, "same", sigmoid -> onset
//
// Same arithmetic and the same packed weight fragments as branch_kernel<OnsetBr> (conv1 as a transposed implicit GEMM on
// v_mfma_f32_32x32x16_f16 with hi/lo-split operands, ReLU + split in registers, conv2 as a 9-tap projection MFMA, the
// horizontal tap sum as two whole-wave DPP shifts, concat channel 0 on the VALU), the decomposition of note_march.hip:
//   * a work item is (window, time chunk, 32-pixel strip) and belongs to ONE wave; the four waves of a workgroup are
//     unrelated tasks and there is no workgroup barrier.  The workgroup kernel spent 37 % of its time in phases without
//     matrix work (LDS-DMA issue, the LDS -> LDS gather of the 8 harmonic shifts, barriers, an output phase) that its two
//     resident workgroups did not cover for each other; here a wave in its staging or epilogue leaves the matrix pipe to
//     the other wave of its SIMD;
//   * the wave keeps its own 6-row ring of the strip's stack image in LDS: 98 slots (stack bins 3 w0 - 1 .. 3 w0 + 96,
//     8 harmonic channels each, f16 hi | lo) x 6 rows = 18.4 KiB per wave, 8 waves per CU.  A row is gathered STRAIGHT
//     from zp (global, L2-resident): a lane fetches the 8 shifted words of its slot — consecutive lanes, consecutive
//     words: coalesced — one row ahead, issued before the row's matrix work and committed (pack + two 16-byte LDS writes)
//     after it.  No raw-row staging, no LDS -> LDS pass;
//   * conv2's vertical 3-tap sum stays in registers: lane half 0 carries q0(r - 1) and (q0(r - 2) + q1(r - 1)), hands the
//     latter to half 1 (one ds_bpermute per row), which adds q2(r) and stores output row r - 1:
//     out[t] = ((q0 + q1) + q2) + bias — the workgroup kernel's summation order;
//   * persistent waves walk the tasks with a fixed stride: 8 chunks x 3 strips x 256 windows = 3 tasks per wave slot.
// Roofline: f16 MFMA issue; 45 MFMAs (39 conv1 + 6 projection) per 30 output pixels; + 9 % rows of chunk halo.
#include <stdlib.h>

#include "bp_common.h"

namespace bp {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int kOmWaves = 4;  // independent waves per workgroup
#ifndef BP_ONSET_MARCH_CHUNKS
#define BP_ONSET_MARCH_CHUNKS 8
#endif
constexpr int kOmChunks = BP_ONSET_MARCH_CHUNKS;  // time chunks per window
constexpr int kOmStrips = 3;                      // 32-pixel strips of a row, 30 inner pixels each
constexpr int kOmRing = 6;                        // image rows a wave keeps: r - 2 .. r + 2 in use, r + 3 being written
constexpr int kOmSlots = 98;                      // stack bins a strip's 32 pixels read: 3 * 31 + 5
constexpr int kOmKS1 = 13;                        // conv1 k-steps: (tap pair of the 5x5 window) x 8 channels
constexpr int kOmPf = 3;                          // k-steps of image fragments read ahead of the matrix instructions
static_assert(kOmStrips * 30 >= kFreqN, "strips cover a row");

struct OnsetMarchParams {
  const uint4* wfrag;   // pack_branch: [A1 hi: 13*64][A1 lo: 13*64][A2 hi: 2*64][A2 lo: 2*64] x (8 x f16)
  const float* wf32;    // bias1[32], the note channel's 3x3 taps at [32 + 3 dt + dw], bias2 at [41]
  const uint32_t* zp;   // [n][kZRowsP][kZRow] packed (hi | lo << 16) words, zero padded (bp_common.h)
  const float* note;    // [n][172][88]
  float* out;           // [n][172][88]
  int n_tasks;          // n_windows * kOmChunks * kOmStrips
};

// tap pair of k-step s: lane half h takes tap 2 s + h of the 5 x 5 window (tap 25 is a zero-weight dummy)
__device__ constexpr int om_dt(int s, int h) { return (2 * s + h) / 5 > 4 ? 4 : (2 * s + h) / 5; }
__device__ constexpr int om_dw(int s, int h) { return (2 * s + h) / 5 > 4 ? 0 : (2 * s + h) % 5; }

template <bool WLO>
__global__ __launch_bounds__(64 * kOmWaves, 2) void onset_march_kernel(OnsetMarchParams p) {
  __shared__ __attribute__((aligned(16))) uint4 lds[kOmWaves][kOmRing * 2 * kOmSlots];  // [wave][row slot][hi | lo][slot]

  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int h = lane >> 5, li = lane & 31;
  uint4* ring = lds[wave];

  // resident A operands and constants
  uint4 a1h[kOmKS1], a1l[WLO ? kOmKS1 : 1], a2h[2], a2l[2];
#pragma unroll
  for (int s = 0; s < kOmKS1; ++s) {
    a1h[s] = p.wfrag[s * 64 + lane];
    if (WLO) a1l[s] = p.wfrag[(kOmKS1 + s) * 64 + lane];
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    a2h[s] = p.wfrag[(2 * kOmKS1 + s) * 64 + lane];
    a2l[s] = p.wfrag[(2 * kOmKS1 + 2 + s) * 64 + lane];
  }
  f32x16 bias1;
#pragma unroll
  for (int r = 0; r < 16; ++r) bias1[r] = p.wf32[(r & 3) + 8 * (r >> 2) + 4 * h];
  // the 3 x 3 taps of the note channel (concat channel 0) for the frame taps this lane half owns: dt = 2 h + i
  float extra[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) extra[i][dw] = (2 * h + i < 3) ? p.wf32[32 + (2 * h + i) * 3 + dw] : 0.0f;
  const float bias2 = p.wf32[41];

  // the ring starts finite: the dummy tap and the halo pixels multiply whatever lies there by zero weights
  for (int i = lane; i < kOmRing * 2 * kOmSlots; i += 64) ring[i] = uint4{0u, 0u, 0u, 0u};

  auto from_left = [](float v) {  // value of lane - 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
  };
  auto from_right = [](float v) {  // value of lane + 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
  };
  const int src_lane4 = (lane & 31) * 4;  // ds_bpermute address: half 1 reads its partner in half 0

  const int total_waves = gridDim.x * kOmWaves;
#pragma unroll 1
  for (int task = blockIdx.x * kOmWaves + wave; task < p.n_tasks; task += total_waves) {  // wave-uniform; no barriers
    const int b = task / (kOmChunks * kOmStrips);
    const int rem = task - b * (kOmChunks * kOmStrips);
    const int ci = rem / kOmStrips, strip = rem - ci * kOmStrips;
    const int T0 = (ci * kFrames) / kOmChunks, T1 = ((ci + 1) * kFrames) / kOmChunks;

    // this lane's pixel of the strip, and the stack bin of image slot 0
    const int w = strip * 30 - 1 + li;
    const bool wvalid = w >= 0 && w < kFreqN;
    const int wc = w < 0 ? 0 : (w >= kFreqN ? kFreqN - 1 : w);
    const bool store_lane = h == 1 && li >= 1 && li <= 30 && w < kFreqN;
    const int f0 = 3 * (strip * 30 - 1) - 1;  // pixel w reads stack bins 3 w - 1 .. 3 w + 3 (ONNX pads [2,1,2,1])
    const uint32_t* zwin = p.zp + (int64_t)b * kZWin + kZPadL;
    const float* nwin = p.note + (int64_t)b * kPlaneN;
    float* owin = p.out + (int64_t)b * kPlaneN;
    // slots of this lane: q = lane and q = 64 + lane (lanes >= 34 re-read slot 97 and write nothing)
    const int qb = 64 + lane < kOmSlots ? 64 + lane : kOmSlots - 1;
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
    auto pack = [](const uint32_t* u, bool ok, uint4& vh, uint4& vl) {
      vh.x = (u[0] & 0xffffu) | (u[1] << 16);
      vh.y = (u[2] & 0xffffu) | (u[3] << 16);
      vh.z = (u[4] & 0xffffu) | (u[5] << 16);
      vh.w = (u[6] & 0xffffu) | (u[7] << 16);
      vl.x = (u[0] >> 16) | (u[1] & 0xffff0000u);
      vl.y = (u[2] >> 16) | (u[3] & 0xffff0000u);
      vl.z = (u[4] >> 16) | (u[5] & 0xffff0000u);
      vl.w = (u[6] >> 16) | (u[7] & 0xffff0000u);
      if (!ok) vh = vl = uint4{0u, 0u, 0u, 0u};
    };
    auto stage_commit = [&](int slot, const uint32_t (&u)[16]) {  // slot: ring slot of the row (scalar)
      uint4 vh, vl;
      pack(u, fa_ok, vh, vl);
      ring[(slot * 2 + 0) * kOmSlots + lane] = vh;
      ring[(slot * 2 + 1) * kOmSlots + lane] = vl;
      pack(u + 8, fb_ok, vh, vl);
      if (64 + lane < kOmSlots) {
        ring[(slot * 2 + 0) * kOmSlots + 64 + lane] = vh;
        ring[(slot * 2 + 1) * kOmSlots + 64 + lane] = vl;
      }
      // the ring is written lane-private and read across lanes: order the wave's LDS writes before the reads that follow
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto note_at = [&](int row) {  // unconditional load from a clamped address, masked where it is used
      const int rc = row < 0 ? 0 : (row > kFrames - 1 ? kFrames - 1 : row);
      return nwin[rc * kFreqN + wc];
    };

    // ---- one conv1 row r: q[i] = horizontal-summed projection of frame tap dt = 2 h + i (half 1: i = 0 only), the
    // note channel's taps included.  slot_m2 = ring slot of image row r - 2
    auto tile = [&](int slot_m2, float note_c, float (&q)[2]) {
      int rb[5];  // ring offsets (in uint4 units) of the hi plane of image rows r - 2 + d
#pragma unroll
      for (int d = 0, s = slot_m2; d < 5; ++d) {
        rb[d] = s * 2 * kOmSlots;
        s = s + 1 == kOmRing ? 0 : s + 1;
      }
      f32x16 acc = bias1, accc;
#pragma unroll
      for (int r = 0; r < 16; ++r) accc[r] = 0.0f;
      f16x8 bhf[kOmKS1], blf[kOmKS1];
      auto issue = [&](int s) {
        const int o0 = rb[om_dt(s, 0)] + om_dw(s, 0);
        const int o1 = rb[om_dt(s, 1)] + om_dw(s, 1);
        const int at = 3 * li + (h ? o1 : o0);
        bhf[s] = __builtin_bit_cast(f16x8, ring[at]);
        blf[s] = __builtin_bit_cast(f16x8, ring[at + kOmSlots]);
      };
#pragma unroll
      for (int s = 0; s < kOmPf; ++s) issue(s);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < kOmKS1; ++s) {
        if (s + kOmPf < kOmKS1) issue(s + kOmPf);
        __builtin_amdgcn_sched_barrier(0);
        const f16x8 ah = __builtin_bit_cast(f16x8, a1h[s]);
        if (WLO) accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1l[WLO ? s : 0]), bhf[s], accc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bhf[s], acc, 0, 0, 0);
        accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, blf[s], accc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ReLU, split, tap projection
      uint32_t b2hw[8], b2lw[8];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        f32x2 v = {__builtin_fmaf(accc[r], kLoUnscale, acc[r]), __builtin_fmaf(accc[r + 1], kLoUnscale, acc[r + 1])};  // plain, not v_pk_fma_f32 (3.5 x the cost beside MFMAs)
        v.x = fmaxf(v.x, 0.0f);
        v.y = fmaxf(v.y, 0.0f);
        split_f16x2(v, b2hw[r >> 1], b2lw[r >> 1]);
      }
      f32x16 pp, ppc;
#pragma unroll
      for (int r = 0; r < 16; ++r) pp[r] = ppc[r] = 0.0f;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const f16x8 b2h = __builtin_bit_cast(f16x8, uint4{b2hw[4 * s], b2hw[4 * s + 1], b2hw[4 * s + 2], b2hw[4 * s + 3]});
        const f16x8 b2l = __builtin_bit_cast(f16x8, uint4{b2lw[4 * s], b2lw[4 * s + 1], b2lw[4 * s + 2], b2lw[4 * s + 3]});
        const f16x8 ah = __builtin_bit_cast(f16x8, a2h[s]);
        const f16x8 al = __builtin_bit_cast(f16x8, a2l[s]);
        pp = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b2h, pp, 0, 0, 0);
        ppc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b2h, ppc, 0, 0, 0);
        ppc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b2l, ppc, 0, 0, 0);
      }
      // packed conv2 weights: C row r = 3 i + dw of lane half h holds tap (dt = 2 h + i, dw) (bp_api.hip pack_branch);
      // Q[dt][w] = (P[dt,0][w-1] + P[dt,1][w]) + P[dt,2][w+1] by two whole-wave lane shifts; pixels outside the row are
      // conv2's zero padding; the note channel's taps on the VALU
      const float n_c = wvalid ? note_c : 0.0f;
      const float n_l = from_left(n_c), n_r = from_right(n_c);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float p0 = pp[3 * i] + ppc[3 * i] * kLoUnscale;
        const float p1 = pp[3 * i + 1] + ppc[3 * i + 1] * kLoUnscale;
        float p2 = pp[3 * i + 2] + ppc[3 * i + 2] * kLoUnscale;
        p0 = wvalid ? p0 : 0.0f;
        p2 = wvalid ? p2 : 0.0f;
        float qq = (from_left(p0) + p1) + from_right(p2);
        qq += (n_l * extra[i][0] + n_c * extra[i][1]) + n_r * extra[i][2];
        q[i] = qq;
      }
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
    }
    float note_nx = note_at(r_first);
    float A = 0.0f, S = 0.0f;
    int slot_m2 = 0;  // ring slot of image row r - 2
#pragma unroll 1
    for (int r = r_first; r <= T1; ++r) {
      uint32_t st[16];
      int slot_p3 = slot_m2 + 5;  // row r + 3 takes the slot of row r - 3
      slot_p3 = slot_p3 >= kOmRing ? slot_p3 - kOmRing : slot_p3;
      stage_issue(r + 3, st);
      const float note_c = note_nx;
      note_nx = note_at(r + 1);
      float q[2] = {0.0f, 0.0f};
      // a conv1 row outside the window is conv2's zero padding (the note channel's row too)
      if (r >= 0 && r < kFrames) {
        tile(slot_m2, note_c, q);
      }
      // half 0: q[0] = q0(r), q[1] = q1(r); half 1: q[0] = q2(r)
      const float X = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane4, __builtin_bit_cast(int, S)));
      S = A + q[1];
      A = q[0];
      const int t = r - 1;  // the output row half 1 finishes now: (q0(r - 2) + q1(r - 1)) + q2(r)
      if (store_lane && t >= T0 && t < T1) owin[t * kFreqN + w] = sigmoidf_fast((X + q[0]) + bias2);
      stage_commit(slot_p3, st);
      slot_m2 = slot_m2 + 1 == kOmRing ? 0 : slot_m2 + 1;
    }
  }
}

void launch_onset_march(const uint32_t* zp, const float* note, const void* wfrag, const float* wf32, float* onset,
                        int n_windows, int n_cu, bool weights_have_lo, hipStream_t stream) {
  OnsetMarchParams p{static_cast<const uint4*>(wfrag), wf32, zp, note, onset, n_windows * kOmChunks * kOmStrips};
  if (p.n_tasks <= 0) return;
  int grid = (p.n_tasks + kOmWaves - 1) / kOmWaves;
  if (grid > 2 * n_cu) grid = 2 * n_cu;  // two resident workgroups per CU (LDS), persistent: the waves walk the tasks
  if (weights_have_lo)
    hipLaunchKernelGGL(onset_march_kernel<true>, dim3(grid), dim3(64 * kOmWaves), 0, stream, p);
  else
    hipLaunchKernelGGL(onset_march_kernel<false>, dim3(grid), dim3(64 * kOmWaves), 0, stream, p);
}

}  // namespace bp
