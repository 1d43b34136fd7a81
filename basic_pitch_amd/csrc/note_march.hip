// Note branch, wave-private march (default path since round 2).
//
//   basic_pitch/models.py:266-290: Conv2D 1->32, 7x7, strides (1,3), "same", ReLU on the sigmoid contour map, then
//   Conv2D 32->1, (7,3), "same", sigmoid -> note
//
// Same arithmetic as branch_kernel<NoteBr> (conv_branch.hip: conv1 as a transposed implicit GEMM on
// v_mfma_f32_32x32x16_f16 with hi/lo-split operands, ReLU + split in registers, conv2 as a 21-tap projection MFMA, the
// horizontal tap sum as two whole-wave DPP shifts, the same packed weight fragments), but a different decomposition.
// That kernel spends 40 % of its time outside the matrix phase (LDS-DMA staging + LDS->LDS im2col, a ring of
// projections in LDS, an output phase, two workgroup barriers per 4 rows) and is limited to 2 waves per SIMD by its
// LDS rings, where its dependent chain MFMA -> VALU -> MFMA -> DPP -> LDS cannot be covered.  Here:
//   * a work item is (window, time chunk, 32-pixel strip) and belongs to ONE wave: the four waves of a workgroup are
//     unrelated tasks; there is no __syncthreads() in the kernel;
//   * the wave keeps its own 10-row ring of the strip's im2col image in LDS (32 slots x (hi | lo) 16 bytes per row
//     = 1 KiB per row, 10 KiB per wave -> 12 waves per CU), staged two rows at a time straight from global memory
//     (lanes 0-31 one row, lanes 32-63 the next; the loads are issued before the pair's matrix work and committed
//     after it);
//   * conv2's vertical 7-tap sum never leaves the registers: marching down the frames, lane half 0 carries the partial
//     sums of frame taps 0..3 of the three output rows still open below it, hands the finished half-sum to half 1
//     (one ds_bpermute per row), which adds taps 4..6 over the next three rows and stores the finished output row:
//     out[t] = (((((q0 + q1) + q2) + q3) + q4) + q5) + q6 + bias — the old kernel's summation order.
// Roofline: f16 MFMA issue; 18 MFMAs (12 conv1 + 6 projection) per 30 output pixels; HBM 182 KB read (+28 % chunk
// halo) and 61 KB written per window.
#include <stdio.h>
#include <stdlib.h>

#include "bp_common.h"

namespace bp {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int kNmWaves = 4;    // independent waves per workgroup
#ifndef BP_NOTE_CHUNKS
#define BP_NOTE_CHUNKS 4
#endif
constexpr int kNmChunks = BP_NOTE_CHUNKS;   // time chunks per window
constexpr int kNmStrips = 3;   // 32-pixel strips of a row, 30 inner pixels each
constexpr int kNmRing = 10;    // image rows a wave keeps: r-3 .. r+4 in use, r+5 / r+6 being written
constexpr int kNmKS1 = 4, kNmPH1 = 3, kNmPH2 = 3;
static_assert(kNmStrips * 30 >= kFreqN, "strips cover a row");

struct NoteMarchParams {
  const uint4* wfrag;    // pack_branch(4, ...): [A1 hi: 4*64][A1 lo: 4*64][A2 hi: 2*64][A2 lo: 2*64] x (8 x f16)
  const float* wf32;     // bias1[32], ..., bias2 at [41]
  const float* contour;  // [n][172][264]
  float* out;            // [n][172][88]
  int n_tasks;           // n_windows * chunks * kNmStrips
  int chunks;            // time chunks per window: kNmChunks at full batches, more when few windows must fill the chip
};

template <bool WLO>
__global__ __launch_bounds__(64 * kNmWaves, 3) void note_march_kernel(NoteMarchParams p) {
  __shared__ __attribute__((aligned(16))) uint4 lds[kNmWaves][2][kNmRing * 32];  // [wave][hi | lo][row slot][pixel]

  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int h = lane >> 5, li = lane & 31;
  const int task = blockIdx.x * kNmWaves + wave;
  if (task >= p.n_tasks) return;  // wave-uniform; no barriers below
  const int b = task / (p.chunks * kNmStrips);
  const int rem = task - b * (p.chunks * kNmStrips);
  const int ci = rem / kNmStrips, strip = rem - ci * kNmStrips;
  const int T0 = (ci * kFrames) / p.chunks, T1 = ((ci + 1) * kFrames) / p.chunks;

  uint4* img_hi = lds[wave][0];
  uint4* img_lo = lds[wave][1];

  // resident A operands and constants
  uint4 a1h[kNmKS1], a1l[kNmKS1], a2h[2], a2l[2];
#pragma unroll
  for (int s = 0; s < kNmKS1; ++s) {
    a1h[s] = p.wfrag[s * 64 + lane];
    a1l[s] = p.wfrag[(kNmKS1 + s) * 64 + lane];
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    a2h[s] = p.wfrag[(2 * kNmKS1 + s) * 64 + lane];
    a2l[s] = p.wfrag[(2 * kNmKS1 + 2 + s) * 64 + lane];
  }
  f32x16 bias1;
#pragma unroll
  for (int r = 0; r < 16; ++r) bias1[r] = p.wf32[(r & 3) + 8 * (r >> 2) + 4 * h];
  const float bias2 = p.wf32[41];

  // this lane's pixel of the strip
  const int w = strip * 30 - 1 + li;
  const bool wvalid = w >= 0 && w < kFreqN;
  const int wc = w < 0 ? 0 : (w >= kFreqN ? kFreqN - 1 : w);
  const bool store_lane = h == 1 && li >= 1 && li <= 30 && w < kFreqN;
  const float* cwin = p.contour + (int64_t)b * kPlaneC;
  float* owin = p.out + (int64_t)b * kPlaneN;
  int bin[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int bb = 3 * wc + i - 2;
    bin[i] = bb < 0 ? 0 : (bb > kFreqC - 1 ? kFreqC - 1 : bb);
  }

  // ---- staging of two image rows (row_a + h): issue 8 loads, commit = zero the padding, split, two 16-byte stores
  auto stage_issue = [&](int row_a, float (&v)[8]) {
    const int row = row_a + h;
    const int rc = row < 0 ? 0 : (row > kFrames - 1 ? kFrames - 1 : row);
    const float* src = cwin + rc * kFreqC;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = src[bin[i]];
  };
  auto stage_commit = [&](int row_a, int slot_a, const float (&vin)[8]) {  // slot_a: ring slot of row_a (scalar)
    const int row = row_a + h;
    int slot = slot_a + h;
    slot = slot >= kNmRing ? slot - kNmRing : slot;
    uint4 vh{0u, 0u, 0u, 0u}, vl{0u, 0u, 0u, 0u};
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = vin[i];
    const bool rvalid = row >= 0 && row < kFrames;
    if (wc == 0) v[0] = v[1] = 0.0f;                       // bins -2, -1: "same" padding
    if (wc == kFreqN - 1) v[5] = v[6] = v[7] = 0.0f;       // bins 264, 265 and the zero-weight dummy tap
    split_f16x2(f32x2{v[0], v[1]}, vh.x, vl.x);
    split_f16x2(f32x2{v[2], v[3]}, vh.y, vl.y);
    split_f16x2(f32x2{v[4], v[5]}, vh.z, vl.z);
    split_f16x2(f32x2{v[6], v[7]}, vh.w, vl.w);
    if (!rvalid) vh = vl = uint4{0u, 0u, 0u, 0u};          // rows outside the window: conv1's zero padding
    img_hi[slot * 32 + li] = vh;
    img_lo[slot * 32 + li] = vl;
  };

  auto from_left = [](float v) {  // value of lane - 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
  };
  auto from_right = [](float v) {  // value of lane + 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
  };

  // ---- one conv1 row r: q[i] = horizontal-summed projection of frame tap dt = 4 h + i (i < 4; half 1: i < 3)
  // slot_m3 = ring slot of image row r - 3
  auto tile = [&](int slot_m3, float (&q)[4]) {
    int rb[8];  // ring offsets (in uint4 units) of image rows r - 3 + d
#pragma unroll
    for (int d = 0, s = slot_m3; d < 8; ++d) {
      rb[d] = s * 32;
      s = s + 1 == kNmRing ? 0 : s + 1;
    }
    f16x8 bhf[kNmKS1], blf[kNmKS1];
#pragma unroll
    for (int s = 0; s < kNmKS1; ++s) {
      // frame tap 7 (s = 3, half 1) is a zero-weight dummy: it re-reads row r + 3 (finite data) instead of row r + 4,
      // which for the second row of a pair is not staged yet
      const int at = (h ? rb[2 * s + 1 < 7 ? 2 * s + 1 : 6] : rb[2 * s]) + li;
      bhf[s] = __builtin_bit_cast(f16x8, img_hi[at]);
      blf[s] = __builtin_bit_cast(f16x8, img_lo[at]);
    }
    // all eight fragment reads in flight before the first MFMA (left alone the compiler waits for each k-step's pair
    // right before using it: four exposed LDS latencies per row)
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc = bias1, accc;
#pragma unroll
    for (int r = 0; r < 16; ++r) accc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < kNmKS1; ++s) {
      const f16x8 ah = __builtin_bit_cast(f16x8, a1h[s]);
      const f16x8 al = __builtin_bit_cast(f16x8, a1l[s]);
      if (WLO) accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bhf[s], accc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bhf[s], acc, 0, 0, 0);
      accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, blf[s], accc, 0, 0, 0);
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
    // packed conv2 weights: C row r = 3 i + dw of lane half h holds tap (dt = 4 h + i, dw) (bp_api.hip pack_branch)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float p0 = pp[3 * i] + ppc[3 * i] * kLoUnscale;
      const float p1 = pp[3 * i + 1] + ppc[3 * i + 1] * kLoUnscale;
      float p2 = pp[3 * i + 2] + ppc[3 * i + 2] * kLoUnscale;
      p0 = wvalid ? p0 : 0.0f;  // pixels outside the row are conv2's zero padding
      p2 = wvalid ? p2 : 0.0f;
      q[i] = (from_left(p0) + p1) + from_right(p2);
    }
  };

  // ---- the march.  Rows r = T0 - 3 .. T1 + 2 in pairs; U[] = open partial sums of this lane half, X = the finished
  // half-sum handed over from half 0 one row ago.
  const int r_first = T0 - kNmPH2;
  const int n_pairs = (T1 - T0 + 2 * kNmPH2 + 1) / 2;
  {  // prologue: image rows r_first - 3 .. r_first + 4 into slots 0 .. 7
    float va[8], vb[8];
    stage_issue(r_first - 3, va);
    stage_issue(r_first - 1, vb);
    stage_commit(r_first - 3, 0, va);
    stage_commit(r_first - 1, 2, vb);
    stage_issue(r_first + 1, va);
    stage_issue(r_first + 3, vb);
    stage_commit(r_first + 1, 4, va);
    stage_commit(r_first + 3, 6, vb);
  }
  float U0 = 0.0f, U1 = 0.0f, U2 = 0.0f, X = 0.0f;
  int slot_m3 = 0;  // ring slot of image row r - 3
  const int src_lane4 = ((lane & 31)) * 4;  // ds_bpermute address: half 1 reads its partner in half 0
#pragma unroll 1
  for (int pi = 0; pi < n_pairs; ++pi) {
    const int r = r_first + 2 * pi;
    float st[8];
    int slot_p5 = slot_m3 + 8;  // rows r + 5, r + 6 go to the slots of rows r - 5 + 10 ...
    slot_p5 = slot_p5 >= kNmRing ? slot_p5 - kNmRing : slot_p5;
    stage_issue(r + 5, st);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int row = r + k;
      float q[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      int sm3 = slot_m3 + k;
      sm3 = sm3 >= kNmRing ? sm3 - kNmRing : sm3;
      // a conv1 row outside the window is conv2's zero padding.  (Both tiles of the pair in one basic block, computed
      // unconditionally, were measured 9 % slower: the chunk's edge rows are wasted work and nothing overlaps anyway
      // at this register budget.)
      if (row >= 0 && row < kFrames) tile(sm3, q);
      // half 0: taps 0..3 of rows row+3 .. row; half 1: X + taps 4..6 of rows row-1 .. row-3
      const float v0 = (h ? X : 0.0f) + q[0];
      const float v1 = U0 + q[1];
      const float v2 = U1 + q[2];
      const float v3 = U2 + q[3];
      U0 = v0, U1 = v1, U2 = v2;
      X = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane4, __builtin_bit_cast(int, v3)));
      const int t = row - 3;  // the output row half 1 has just finished
      if (store_lane && t >= T0 && t < T1) owin[t * kFreqN + w] = sigmoidf_fast(v2 + bias2);
    }
    stage_commit(r + 5, slot_p5, st);
    slot_m3 = slot_m3 + 2 >= kNmRing ? slot_m3 + 2 - kNmRing : slot_m3 + 2;
  }
}

void launch_note_march(const float* contour, const void* wfrag, const float* wf32, float* note, int n_windows,
                       bool weights_have_lo, hipStream_t stream) {
  int chunks = kNmChunks;  // small batches: shorter chunks (a task's 6 rows of halo weigh more, its serial march less)
  while (chunks < 16 && (int64_t)n_windows * chunks * kNmStrips < 3072) chunks *= 2;
  NoteMarchParams p{static_cast<const uint4*>(wfrag), wf32, contour, note, n_windows * chunks * kNmStrips, chunks};
  if (p.n_tasks <= 0) return;
  const int grid = (p.n_tasks + kNmWaves - 1) / kNmWaves;
  static const bool prof = ab_env("BP_BRANCH_PROF") != nullptr;
  if (prof) {  // tools only
    int resident = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, note_march_kernel<true>, 64 * kNmWaves, 0);
    fprintf(stderr, "brprof note_march: %d workgroups resident per CU, grid %d\n", resident, grid);
  }
  if (weights_have_lo)
    hipLaunchKernelGGL(note_march_kernel<true>, dim3(grid), dim3(64 * kNmWaves), 0, stream, p);
  else
    hipLaunchKernelGGL(note_march_kernel<false>, dim3(grid), dim3(64 * kNmWaves), 0, stream, p);
}

}  // namespace bp
