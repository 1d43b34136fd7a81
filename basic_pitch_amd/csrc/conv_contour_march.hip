// Folded contour conv1 for the interior of the stack as a wave-private VERTICAL march on v_mfma_f32_16x16x32_f16
// (round 4; the default.  conv_contour_direct.hip keeps the round-2 form — 256-position rounds, A and B from LDS,
// 32x32x16 — behind BP_CONV1=rounds).
//
//   basic_pitch/nn.py:69-88 (harmonic stack), basic_pitch/models.py:241-250: Conv2D 8 -> 8, 3 x 39, "same", BN, ReLU
//
// Same operator as contour_conv1_folded_kernel: away from the crop of the stack the 8 x 39-tap kernels of an output channel
// fold into one 176-tap kernel over z, K[o][dt][g] = sum_c W1[o][c][dt][g - s_c + 19]; bins 20 .. 243 here, the rim stays
// with conv_contour_rim.hip.  What changes is who holds what.  The round form streams BOTH operands from LDS (4
// ds_read_b128 per 3 32x32x16) because its 288 VGPRs of weight fragments do not fit beside the accumulators; this chip
// sustains ~1.3 PFLOP/s that way and ~1.7 PFLOP/s on 16x16x32 with the weights in registers and one B read feeding several
// matrix instructions (profiles/r04_ubench_rega.md).  Here:
//   * M = 16 rows = (2-bin offset j, 8 out channels): ONE block of weights, 3 dt x 6 k-steps x {hi, lo} fragments = 144
//     VGPRs, resident for the whole kernel.  A position is a PAIR of bins; N = 16 positions = 32 bins = a wave's strip;
//   * the wave marches down the z rows: a z row's B fragment (8 consecutive bins of one position and k-step) is read ONCE
//     and feeds all three frame taps — output frame zr + 1 through dt = 0, zr through dt = 1, zr - 1 through dt = 2 — so a
//     pair of reads (hi, lo) feeds 9 matrix instructions (72 KFLOP per ds_read_b128; the round form: 24).  Three
//     accumulator pairs are open at a time; the row loop is unrolled by 3 so that their rotation is a renaming;
//   * a work item is (window, 32-bin strip, chunk of frames) and belongs to ONE wave: no workgroup barrier, no loader, 8
//     independent waves per CU.  The wave stages its own z rows — one 16-byte load per lane covers the strip's window of a
//     row (230 words), split to hi / lo by v_perm and written to a double-buffered image in LDS one row ahead;
//   * a position's 8 bins start at an arbitrary EVEN bin, 4 bytes of f16: the image keeps four copies of a row, shifted by
//     0 / 2 / 4 / 6 bins, so that every fragment is an aligned ds_read_b128; copy offsets {0, 40, 85, 125} units were
//     enumerated against the hardware's ds_read_b128 lane groups: conflict-free (uniform copy strides are not);
//   * the C layout hands a lane 4 consecutive channels of one bin: bias + ReLU + one 16-byte store; a wave's 64 lanes
//     write 1 KB of c1 contiguously per frame.
// Roofline: f16 MFMA issue; 54 16x16x32 per 32 bins and frame (= the round form's 108 32x32x16 per 128); + 2 rows of chunk
// warm-up per 21.5 frames.  Bytes per window: 312 KB of zp read (3.5 x through L2: overlapping strip windows), 1.25 MB of
// c1 written.
#include <stdlib.h>
#include <string.h>

#include "bp_common.h"

namespace bp {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int kCmWaves = 4;          // independent waves per workgroup
constexpr int kCmStrips = 7;         // 32-bin strips: bins 20 .. 243
#ifndef BP_CM_CHUNKS
#define BP_CM_CHUNKS 4
#endif
constexpr int kCmChunks = BP_CM_CHUNKS;  // frame chunks per window (4: 0.211 ms at B = 256; 8: 0.225 — twice the warm-up rows; 1: 0.227 —
                                         // 7 of a CU's 8 wave slots busy; same-box sweep at the end of round 4: 2 / 3 / 4 / 5 / 6
                                         // chunks = 0.225 / 0.227 / 0.201 / 0.210 / 0.215 ms)
constexpr int kCmKS = 6;             // k-steps of 32 taps per frame tap: 192 >= 176 + 1 + 1
constexpr int kCmCopyU = 30;         // 16-byte units per row copy: one unit of front slack (the staging lanes whose words lie
                                     // in front of a shifted copy write there instead of branching), 28 used, one behind
// unit offsets of the four shifted copies of a row (hi plane; the lo plane kCmLoU units behind): residues 0 / 8 / 5 / 13
// mod 16 make every ds_read_b128 of the kernel conflict-free (tools: the enumeration in DESIGN.md §7)
__device__ constexpr int cm_copy_off(int c) { return c == 0 ? 0 : (c == 1 ? 40 : (c == 2 ? 85 : 125)); }
constexpr int kCmLoU = 160;
constexpr int kCmRowU = 2 * kCmLoU;  // one z row image: 320 units = 5 KB
#ifndef BP_CM_PF
#define BP_CM_PF 1
#endif
constexpr int kCmPf = BP_CM_PF;      // k-steps of B fragments read ahead of the matrix instructions
static_assert(cm_copy_off(3) + kCmCopyU <= kCmLoU, "copies fit a plane");

struct ContourMarchParams {
  const uint4* wfrag;  // pack_contour_march: [3 dt][6 k-steps][hi | lo][64 lanes] x (8 x f16)
  const float* bias;   // [8]
  const uint32_t* zp;  // [n][kZRowsP][kZRow] packed (hi | lo << 16) words, zero padded (bp_common.h)
  float* c1;           // [n][172][kC1Row][8]
  int n_tasks;         // n_windows * chunks * kCmStrips
  int chunks;          // frame chunks per window: kCmChunks at full batches, more when few windows must fill the chip
};

template <bool WLO>
__global__ __launch_bounds__(64 * kCmWaves, 2) void contour_conv1_march_kernel(ContourMarchParams p) {
  __shared__ __attribute__((aligned(16))) uint4 lds[kCmWaves][2 * kCmRowU];  // [wave][row buffer][hi | lo][copy][unit]

  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int gq = lane >> 4, n = lane & 15;
  uint4* img = lds[wave];

  // resident weights
  uint4 ah[3][kCmKS], al[WLO ? 3 : 1][kCmKS];
#pragma unroll
  for (int dt = 0; dt < 3; ++dt)
#pragma unroll
    for (int s = 0; s < kCmKS; ++s) {
      ah[dt][s] = p.wfrag[((dt * kCmKS + s) * 2 + 0) * 64 + lane];
      if (WLO) al[WLO ? dt : 0][s] = p.wfrag[((dt * kCmKS + s) * 2 + 1) * 64 + lane];
    }
  float bias4[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bias4[r] = p.bias[4 * (gq & 1) + r];

  // the image starts finite (slack units are read and multiplied by zero weights)
  for (int i = lane; i < 2 * kCmRowU; i += 64) img[i] = uint4{0u, 0u, 0u, 0u};

  const int total_waves = gridDim.x * kCmWaves;
  // XCD-aware task order: workgroups go to the 8 XCDs round-robin (blockIdx % 8) and each XCD has its own L2; consecutive
  // tasks — the strips and chunks of ONE window, which read overlapping parts of the same zp rows — are given to
  // workgroups of the same XCD, so a row is fetched from HBM by one L2 instead of by up to eight
  // (inside each half of the grid: a CU hosts workgroups p and p + gridDim.x / 2, and when the tasks per wave are not a whole
  // number the first half of the LOGICAL blocks carries the extra task — the pair of a CU must stay (first, second half))
  // (Round 5 tried equal shares of the frames per WAVE instead — wave g of 2048 marching the g-th 2048th of all frames,
  // 154 rows each instead of 4 x 45 for half the waves and 3 x 45 for the others: 0.228 ms against 0.2155.  The 3.5 tasks
  // per wave are no imbalance — every SIMD hosts one wave of each kind and the matrix pipe is what they share — while
  // consecutive tasks, the strips of one (window, chunk), read the same zp rows at the same time from the same CU.)
  const int half_n = (int)gridDim.x / 2, pq = (int)blockIdx.x % (half_n > 0 ? half_n : 1);
  const int lblock = (gridDim.x % 16 == 0) ? ((int)blockIdx.x / half_n) * half_n + (pq % 8) * (half_n / 8) + pq / 8 : (int)blockIdx.x;
#pragma unroll 1
  for (int task = lblock * kCmWaves + wave; task < p.n_tasks; task += total_waves) {  // wave-uniform; no barriers
    const int b = task / (p.chunks * kCmStrips);
    const int rem = task - b * (p.chunks * kCmStrips);
    const int ci = rem / kCmStrips, strip = rem - ci * kCmStrips;
    const int T0 = (ci * kFrames) / p.chunks, T1 = ((ci + 1) * kFrames) / p.chunks;
    const int p0 = 10 + 16 * strip;                  // first position (bin pair) of the strip: bins 2 p0 = 20 + 32 strip
    const int wb = 2 * p0 - 4;                       // first zp word of the strip's window (a multiple of 4)
    const uint32_t* zwin = p.zp + (int64_t)b * kZWin + wb;
    float* c1b = p.c1 + (int64_t)b * kC1Win;

    // B fragment of this lane's position p = p0 + n: copy c = p & 3 (p0 = 2 mod 4), unit (p - c - (p0 - 2)) / 4 + 4 s + gq
    const int cpy = (n + 2) & 3;
    const int boff = (cpy == 0 ? cm_copy_off(0) : cpy == 1 ? cm_copy_off(1) : cpy == 2 ? cm_copy_off(2) : cm_copy_off(3)) +
                     1 + ((n + 2 - cpy) >> 2) + gq;  // + 1: the copy's slack unit
    // this lane's store: bin 2 p + (gq >> 1), channels 4 (gq & 1) .. + 3
    const int64_t st_off = ((int64_t)kC1Pad + 2 * (p0 + n) + (gq >> 1)) * 8 + 4 * (gq & 1);

    // ---- staging: lane l < 58 fetches words wb + 4 l .. + 3 of a z row; commit = split to hi / lo pairs (v_perm) and write
    // them into the four shifted copies: copy c holds word wb + 2 c + e at element e, i.e. this lane's 4 words at byte
    // 8 l - 4 c of the copy (behind its slack unit)
    auto stage_issue = [&](int zr) -> uint4 {  // zr in [-1, 172]: zp row zr + 1 (rows -1 and 172 are zero)
      const uint4* src = reinterpret_cast<const uint4*>(zwin + (int64_t)(zr + 1) * kZRow);
      return src[lane < 58 ? lane : 57];
    };
    auto stage_commit = [&](int buf, const uint4& wv) {
      const uint32_t h0 = __builtin_amdgcn_perm(wv.y, wv.x, 0x05040100u), h1 = __builtin_amdgcn_perm(wv.w, wv.z, 0x05040100u);
      const uint32_t l0 = __builtin_amdgcn_perm(wv.y, wv.x, 0x07060302u), l1 = __builtin_amdgcn_perm(wv.w, wv.z, 0x07060302u);
      uint32_t* base = reinterpret_cast<uint32_t*>(img + buf * kCmRowU);  // dword view of the row image
      if (lane < 58) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int dw = 4 + 2 * lane - c;  // dword index inside copy c (>= 1: the slack unit absorbs the words in front)
          uint32_t* ch = base + 4 * cm_copy_off(c);
          uint32_t* cl = ch + 4 * kCmLoU;
          ch[dw] = h0;
          ch[dw + 1] = h1;
          cl[dw] = l0;
          cl[dw + 1] = l1;
        }
      }
    };
    // lane-private writes, cross-lane reads: ordered inside the wave by this fence — placed at the END of a row step, a
    // whole row of matrix work behind the writes it waits for (right behind them it exposed the LDS write latency per row)
    auto image_fence = [] {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    // ---- one z row: its fragments feed the three open output frames.  a0 <- dt = 0 (first contribution: starts from
    // zero), a1 <- dt = 1, a2 <- dt = 2 (complete afterwards)
    auto row_step = [&](int buf, f32x4 (&h0)[1], f32x4 (&x0)[1], f32x4 (&h1)[1], f32x4 (&x1)[1], f32x4 (&h2)[1],
                        f32x4 (&x2)[1]) {
      const uint4* rowp = img + buf * kCmRowU + boff;
      f16x8 bh[kCmKS], bl[kCmKS];
      auto issue = [&](int s) {
        bh[s] = __builtin_bit_cast(f16x8, rowp[4 * s]);
        bl[s] = __builtin_bit_cast(f16x8, rowp[4 * s + kCmLoU]);
      };
#pragma unroll
      for (int s = 0; s < kCmPf; ++s) issue(s);
      __builtin_amdgcn_sched_barrier(0);
      const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
      const f32x4 bias_v = {bias4[0], bias4[1], bias4[2], bias4[3]};
#pragma unroll
      for (int s = 0; s < kCmKS; ++s) {
        if (s + kCmPf < kCmKS) issue(s + kCmPf);
        __builtin_amdgcn_sched_barrier(0);
#define BP_CM_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), b, c, 0, 0, 0)
        if (WLO) {
          x0[0] = BP_CM_MFMA(al[0][s], bh[s], s == 0 ? zero4 : x0[0]);
          x1[0] = BP_CM_MFMA(al[WLO ? 1 : 0][s], bh[s], x1[0]);
          x2[0] = BP_CM_MFMA(al[WLO ? 2 : 0][s], bh[s], x2[0]);
        }
        h0[0] = BP_CM_MFMA(ah[0][s], bh[s], s == 0 ? bias_v : h0[0]);  // the frame's sum starts at the bias
        h1[0] = BP_CM_MFMA(ah[1][s], bh[s], h1[0]);
        h2[0] = BP_CM_MFMA(ah[2][s], bh[s], h2[0]);
        x0[0] = BP_CM_MFMA(ah[0][s], bl[s], (s == 0 && !WLO) ? zero4 : x0[0]);
        x1[0] = BP_CM_MFMA(ah[1][s], bl[s], x1[0]);
        x2[0] = BP_CM_MFMA(ah[2][s], bl[s], x2[0]);
#undef BP_CM_MFMA
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    auto finish = [&](int t, const f32x4& hh, const f32x4& xx) {  // frame t is complete: bias, ReLU, one 16-byte store
      if (t < T0 || t >= T1) return;                              // wave-uniform: warm-up / run-out rows of the chunk
      float4 v;
      v.x = relu_f32(__builtin_fmaf(xx[0], kLoUnscale, hh[0]));
      v.y = relu_f32(__builtin_fmaf(xx[1], kLoUnscale, hh[1]));
      v.z = relu_f32(__builtin_fmaf(xx[2], kLoUnscale, hh[2]));
      v.w = relu_f32(__builtin_fmaf(xx[3], kLoUnscale, hh[3]));
      *reinterpret_cast<float4*>(c1b + (int64_t)t * kC1Row * 8 + st_off) = v;
    };

    // ---- the march over z rows zr = T0 - 1 .. T1: row zr completes frame zr - 1
    f32x4 hA[1], xA[1], hB[1], xB[1], hC[1], xC[1];
    hA[0] = xA[0] = hB[0] = xB[0] = hC[0] = xC[0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // rows T0 - 1, T0, T0 + 1 requested up front; in the march two rows are in flight: the one committed in a step was
    // requested two steps ago (one row of matrix work is less than an L2 / HBM round trip under load: with one row in flight
    // the waves spent 28 % of their cycles in s_waitcnt)
    const auto row_c = [](int r) { return r <= kFrames ? r : kFrames; };
    uint4 ld0 = stage_issue(T0 - 1);
    uint4 ld1 = stage_issue(T0);
    uint4 ld2 = stage_issue(row_c(T0 + 1));
    stage_commit(0, ld0);
    image_fence();
    ld0 = ld1, ld1 = ld2;
    int buf = 0;
    int zr = T0 - 1;
    // one step: commit the next row's words (requested two rows ago) into the other buffer, request the row after the next,
    // run this row's matrix work, finish the frame it completes
#define BP_CM_STEP(N0, N1, N2)                                          \
  do {                                                                  \
    stage_commit(buf ^ 1, ld0);                                         \
    ld0 = ld1;                                                          \
    ld1 = stage_issue(row_c(zr + 3));                                   \
    row_step(buf, h##N0, x##N0, h##N1, x##N1, h##N2, x##N2);            \
    image_fence();                                                      \
    finish(zr - 1, h##N2[0], x##N2[0]);                                 \
    buf ^= 1;                                                           \
    ++zr;                                                               \
  } while (0)
    // accumulators rotate: the set that took dt = 0 takes dt = 1 on the next row and dt = 2 on the one after
#pragma unroll 1
    while (zr <= T1) {
      BP_CM_STEP(A, C, B);
      if (zr > T1) break;
      BP_CM_STEP(B, A, C);
      if (zr > T1) break;
      BP_CM_STEP(C, B, A);
    }
#undef BP_CM_STEP
  }
}

// folded conv1, interior bins: the wave-private march (default) or the round-2 kernel (BP_CONV1=rounds)
bool contour_conv1_use_march() {
  static const bool rounds = [] {
    const char* e = ab_env("BP_CONV1");
    return e && strcmp(e, "rounds") == 0;
  }();
  return !rounds;
}

void launch_contour_conv1_march(const uint32_t* zp, const void* wfrag, const float* bias, float* c1, int n_windows, int n_cu,
                                bool weights_have_lo, hipStream_t stream) {
  // small batches: more, shorter chunks until there is a task per resident wave (a task is a serial march of rows: at
  // one window, 4 chunks leave 28 waves marching 45 rows each while 2020 wave slots idle)
  int chunks = kCmChunks;
  while (chunks < 32 && (int64_t)n_windows * chunks * kCmStrips < (int64_t)2 * n_cu * kCmWaves) chunks *= 2;
  ContourMarchParams p{static_cast<const uint4*>(wfrag), bias, zp, c1, n_windows * chunks * kCmStrips, chunks};
  if (p.n_tasks <= 0) return;
  int grid = (p.n_tasks + kCmWaves - 1) / kCmWaves;
  if (grid > 2 * n_cu) grid = 2 * n_cu;  // two workgroups of four waves per CU (registers), persistent
  if (weights_have_lo)
    hipLaunchKernelGGL(contour_conv1_march_kernel<true>, dim3(grid), dim3(64 * kCmWaves), 0, stream, p);
  else
    hipLaunchKernelGGL(contour_conv1_march_kernel<false>, dim3(grid), dim3(64 * kCmWaves), 0, stream, p);
}

}  // namespace bp
