// Contour conv1, the rim of the harmonic stack with the weights RESIDENT IN REGISTERS (round 5; default for the 309-bin CQT).
//
//   basic_pitch/nn.py:69-88 (HarmonicStacking: shift, zero-pad, CROP to 264 bins) + basic_pitch/models.py:241-250
//   (Conv2D 8->8, 3 x 39, "same", folded BN, ReLU) for the output bins f < 20 and f >= 244.
//
// The operator is the dense per-side matrix of conv_contour_rim.hip,
//     C[(f, o) : 160 rows][frame] = K[(f, o)][(dt, j) : 3 x 144] x Z[(dt, j)][frame],   Z[(dt, j)][t] = z[t + dt - 1][j0 + j],
// and that kernel's problem is its A operand: every workgroup of (window, side, 64 frames) streams the side's 276 KB of K
// from L2 again — 425 MB per 256 windows, its waves in s_waitcnt 42 % of their cycles, the matrix pipes busy 44 %
// (profiles/r05_c_stalls.md).  Here K never moves:
//   * M = 16 rows = (2 bins x 8 channels) on v_mfma_f32_16x16x32_f16, K = 432 -> 14 k-steps of 32 (the last half zero):
//     a wave owns ONE 16-row block of one side and keeps its A fragments — 14 steps x {hi, lo} x 4 = 112 VGPRs — for
//     the whole kernel (the round-4 plan: "ten 16-row weight blocks of 120 VGPRs per rim side, frames as N");
//   * a workgroup is the TEN waves of a side and is persistent: it walks (window, 16-frame tile) items; the B operand of
//     an item — z rows t0 - 1 .. t0 + 16 of the side's 144 bins, 18 x 288 B per plane — is staged once per item into a
//     double-buffered f16 hi / lo row image in LDS and read by all ten waves: one ds_read_b128 per k-step and plane, row
//     pitch = the natural 18 units, conflict-free for the hardware's ds_read_b128 lane groups at every k-step (enumerated);
//   * the next item's zp words are requested before the item's 42 matrix instructions and de-interleaved into the other
//     buffer behind them; one barrier per item;
//   * three accumulator chains (hi hi, lo hi, hi lo) of 14 dependent matrix instructions each; bias + ReLU + one 16-byte
//     store per lane from the accumulator layout (a lane holds 4 consecutive channels of one bin of its frame).
// The frames of a tile are the N columns: the three frame taps of a z row are lane-shifted copies of each other, so there
// is no B reuse across taps (the interior march's trick) — the gain is the A stream alone.
// Roofline: f16 MFMA issue.  Algorithmic work 103 MFLOP per window; executed 2 x 10 x 11 x 42 matrix instructions of 16 KFLOP;
// bytes per window: 2 x 11 x 18 x 576 B of zp read (through L2), 220 KB of c1 written; A: 573 KB per WORKGROUP lifetime.
#include <stdio.h>

#include "bp_common.h"

namespace bp {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int kRmBlocks = 10;                       // 16-row blocks per side: 20 bins x 8 channels
constexpr int kRmThreads = 64 * kRmBlocks;          // one wave per block
constexpr int kRmBins = 144;                        // z bins per frame a side reads (conv_contour_rim.hip RimGeo<144>)
constexpr int kRmUnits = kRmBins / 8;               // 16-byte units per plane and row: 18 (also the LDS row pitch)
constexpr int kRmK = 3 * kRmBins;                   // 432
constexpr int kRmSteps = (kRmK + 31) / 32;          // 14
constexpr int kRmTileFrames = 16;
constexpr int kRmTiles = (kFrames + kRmTileFrames - 1) / kRmTileFrames;  // 11
constexpr int kRmRows = kRmTileFrames + 2;          // z rows of a tile
constexpr int kRmPlaneU = kRmRows * kRmUnits;       // units per plane: 324
constexpr int kRmBufU = 2 * kRmPlaneU;              // hi plane, lo plane
__host__ __device__ constexpr int rm_f0(int side) { return side ? 244 : 0; }
__host__ __device__ constexpr int rm_j0(int side) { return side ? 184 : 0; }

struct RimMarchParams {
  const uint32_t* zp;   // [n][kZRowsP][kZRow]
  const uint4* afrag;   // [side 2][block 10][step 14][hi | lo][64 lanes] x (8 x f16)   (bp_api.hip pack_contour_rim_march)
  const float* bias;    // [8]
  float* c1;            // [n][172][kC1Row][8]
  int n_items;          // n_windows * kRmTiles (per side)
};

// 16 bytes per lane, global -> LDS at `lds_wave_base` + 16 * lane (wave-uniform base), asynchronous (vmcnt).  Inline assembly
// on purpose (conv_branch.hip lds_dma16: through the builtin the compiler puts s_waitcnt vmcnt(0) in front of the next LDS
// read); the kernel orders the DMA itself.
__device__ __forceinline__ void rm_dma16(const void* gsrc, uint4* lds_wave_base) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(lds_wave_base));
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(base) : "memory");
#endif
}

#ifndef BP_RM_PF
#define BP_RM_PF 2
#endif
#ifndef BP_RM_DEPTH
#define BP_RM_DEPTH 4
#endif
constexpr int kRmDepth = BP_RM_DEPTH;        // items whose zp words are in flight (LDS-DMA)
constexpr int kRmRing = kRmDepth + 1;        // raw buffers
static_assert(kRmDepth >= 2, "the counted s_waitcnt vmcnt(1 + 3 (kRmDepth - 2)) of the item loop assumes at least two items in flight");
constexpr int kRmRawU = 2 * kRmPlaneU + 2;   // 648 pieces of 16 bytes + 2 dummies (every wave issues two DMA instructions)
static_assert(2 * kRmPlaneU == kRmBlocks * 64 + 8, "piece q = 64 wave + lane, and one more piece for waves 0..7");
static_assert((kRmRing * kRmRawU + 2 * kRmBufU) * 16 <= 160 * 1024, "LDS budget");

template <bool WLO>
__global__ __launch_bounds__(kRmThreads) void contour_conv1_rim_march_kernel(RimMarchParams p) {
  __shared__ __attribute__((aligned(16))) uint4 raw_ring[kRmRing * kRmRawU];  // zp words as they come: [row 18][36 pieces]
  __shared__ __attribute__((aligned(16))) uint4 img[2 * kRmBufU];             // [buffer][hi | lo][row][unit]

  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int n = lane & 15, g = lane >> 4;
  const int side = blockIdx.x & 1;
  const int wg = blockIdx.x >> 1, n_wg = gridDim.x >> 1;  // workgroups of this side
  const int n_my = wg < p.n_items ? (p.n_items - wg + n_wg - 1) / n_wg : 0;  // items wg, wg + n_wg, ...

  // ---- resident A fragments of this wave's block
  uint4 ah[kRmSteps], al[WLO ? kRmSteps : 1];
  {
    const uint4* afr = p.afrag + ((int64_t)(side * kRmBlocks + wave) * kRmSteps) * 2 * 64 + lane;
#pragma unroll
    for (int s = 0; s < kRmSteps; ++s) {
      ah[s] = afr[(2 * s) * 64];
      if (WLO) al[WLO ? s : 0] = afr[(2 * s + 1) * 64];
    }
  }
  const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias + 4 * (g & 1));

  // ---- the z rows of an item come in by LDS-DMA, kRmDepth items ahead (one item of matrix work, ~0.8 us, is a third of
  // the round trip under load: with the words of ONE item ahead in registers the kernel took 2.4 us per item).  Piece q of
  // an item = 16 bytes = words 4 c .. 4 c + 3 of row q / 36; wave w brings pieces 64 w + lane, and lane 0 piece 640 + w
  // (waves 8, 9: piece 647 once more, into a dummy unit — every wave issues the same two instructions per item, which is
  // what the vmcnt arithmetic below counts on).
  const int q0 = 64 * wave + lane, q1 = wave < 8 ? 640 + wave : 647;
  const int q0_row = q0 / 36, q0_col = q0 - 36 * q0_row, q1_row = q1 / 36, q1_col = q1 - 36 * q1_row;
  auto dma_issue = [&](int k) {  // this workgroup's k-th item
    const int item = wg + k * n_wg;
    const int b = item / kRmTiles, t0 = (item - b * kRmTiles) * kRmTileFrames;
    const uint32_t* zwin = p.zp + (int64_t)b * kZWin + kZPadL + rm_j0(side);
    auto row_ptr = [&](int row, int col) {
      const int t = t0 - 1 + row;  // zp rows -1 and 172 are zero; frames beyond them read the all-zero row -1
      return zwin + (int64_t)((t <= kFrames ? t : -1) + 1) * kZRow + 4 * col;
    };
    uint4* dst = raw_ring + (k % kRmRing) * kRmRawU;
    rm_dma16(row_ptr(q0_row, q0_col), dst + 64 * wave);
    if (lane == 0) rm_dma16(row_ptr(q1_row, q1_col), dst + (wave < 8 ? 640 + wave : 640 + wave));  // 648, 649: dummies
  };
  // raw -> hi / lo image: thread e < 324 owns unit (row, u) = raw pieces 2 e, 2 e + 1 (8 packed words)
  const int su = (int)threadIdx.x;
  auto convert = [&](int k) {
    if (su >= kRmPlaneU) return;
    // the DMA's LDS writes are invisible to the optimiser: a laundered INDEX (a laundered pointer loses its address space:
    // the reads became flat loads, whose s_waitcnt vmcnt(0) drained every DMA in flight once per item)
    int ri = (k % kRmRing) * kRmRawU + 2 * su;
    asm volatile("" : "+v"(ri));
    const uint4 w0 = raw_ring[ri], w1 = raw_ring[ri + 1];
    uint4 vh, vl;
    vh.x = __builtin_amdgcn_perm(w0.y, w0.x, 0x05040100u), vl.x = __builtin_amdgcn_perm(w0.y, w0.x, 0x07060302u);
    vh.y = __builtin_amdgcn_perm(w0.w, w0.z, 0x05040100u), vl.y = __builtin_amdgcn_perm(w0.w, w0.z, 0x07060302u);
    vh.z = __builtin_amdgcn_perm(w1.y, w1.x, 0x05040100u), vl.z = __builtin_amdgcn_perm(w1.y, w1.x, 0x07060302u);
    vh.w = __builtin_amdgcn_perm(w1.w, w1.z, 0x05040100u), vl.w = __builtin_amdgcn_perm(w1.w, w1.z, 0x07060302u);
    uint4* dst = img + (k & 1) * kRmBufU;
    dst[su] = vh;
    dst[kRmPlaneU + su] = vl;
  };

  // ---- this lane's B fragment at k-step s: k = 32 s + 8 g -> (dt, 8-bin unit) of row n + dt; k >= 432 carries zero
  // weights: those lanes re-read the units of k - 16 (finite data, and the read stays conflict-free)
  // (row n + dt, unit (k - 144 dt) / 8 of a plane whose row pitch is the 18 units of a row: the frame taps cancel, the unit
  // is 18 n + k / 8 = (18 n + g) + 4 s — one lane base and an immediate per k-step)
  static_assert(kRmBins % 8 == 0 && kRmK % 8 == 0, "units do not straddle frame taps");
  const int ub = kRmUnits * n + g;
  const int ub_last = ub - (32 * (kRmSteps - 1) + 8 * g >= kRmK ? 2 : 0);  // the last k-step's base
  auto b_unit = [&](int s) { return (s == kRmSteps - 1 ? ub_last : ub) + 4 * s; };

  if (n_my <= 0) return;  // workgroup-uniform
#pragma unroll
  for (int k = 0; k < kRmDepth; ++k)
    if (k < n_my) dma_issue(k);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // The compiler must KNOW that the weights have landed: it waits for a load at its first use, which is inside the item loop
  // — s_waitcnt vmcnt(26) .. vmcnt(4) between the matrix instructions, no-ops from the second item on by its own count,
  // but with DMA instructions it does not see in flight each of them waits until all but N operations have completed, i.e.
  // for the DMA of items ahead: the loop ran at the round trip's pace, 2.2 us per item.  A use in front of the loop puts
  // the compiler's wait here.
#pragma unroll
  for (int s = 0; s < kRmSteps; ++s) {
    asm volatile("" ::"v"(ah[s].x), "v"(ah[s].y), "v"(ah[s].z), "v"(ah[s].w));
    if (WLO) asm volatile("" ::"v"(al[WLO ? s : 0].x), "v"(al[WLO ? s : 0].y), "v"(al[WLO ? s : 0].z), "v"(al[WLO ? s : 0].w));
  }
  asm volatile("" ::"v"(bias4[0]), "v"(bias4[1]), "v"(bias4[2]), "v"(bias4[3]));
  __syncthreads();  // every wave's pieces of items 0 .. kRmDepth - 1 are in LDS
  convert(0);
  lds_barrier();
#pragma unroll 1
  for (int k = 0; k < n_my; ++k) {
    const bool more = k + kRmDepth < n_my;  // workgroup-uniform
#ifdef RM_PROF
    unsigned long long ps[7];
    ps[0] = __builtin_amdgcn_s_memtime();
#endif
    if (more) dma_issue(k + kRmDepth);      // its ring slot held item k - 1, converted during item k - 2
#ifdef RM_PROF
    ps[1] = __builtin_amdgcn_s_memtime();
#endif

    const uint4* bh_p = img + (k & 1) * kRmBufU;
    f32x4 hh = {0.f, 0.f, 0.f, 0.f}, xa = hh, xb = hh;
    constexpr int kPf = BP_RM_PF;  // k-steps of B fragments read ahead
    uint4 bh[kPf + 1], bl[kPf + 1];
    auto rd = [&](int s) {
      const int u = b_unit(s);
      bh[s % (kPf + 1)] = bh_p[u];
      bl[s % (kPf + 1)] = bh_p[kRmPlaneU + u];
    };
#pragma unroll
    for (int s = 0; s < kPf; ++s) rd(s);
#pragma unroll
    for (int s = 0; s < kRmSteps; ++s) {
      if (s + kPf < kRmSteps) rd(s + kPf);
      __builtin_amdgcn_sched_barrier(0);
#define BP_RM_MFMA(a, b, c) \
  __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0)
      hh = BP_RM_MFMA(ah[s], bh[s % (kPf + 1)], hh);
      if (WLO) xa = BP_RM_MFMA(al[WLO ? s : 0], bh[s % (kPf + 1)], xa);
      xb = BP_RM_MFMA(ah[s], bl[s % (kPf + 1)], xb);
#undef BP_RM_MFMA
      __builtin_amdgcn_sched_barrier(0);
    }

#ifdef RM_PROF
    asm volatile("" : "+v"(hh), "+v"(xa), "+v"(xb));
    ps[2] = __builtin_amdgcn_s_memtime();
#endif
    // epilogue: D row i = 4 g + r = 8 (bin of the block) + channel -> channels 4 (g & 1) .. + 3 of bin g >> 1; column n = frame
    {
      const int item = wg + k * n_wg;
      const int b = item / kRmTiles, t = (item - b * kRmTiles) * kRmTileFrames + n;
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = relu_f32(__builtin_fmaf(xa[r] + xb[r], kLoUnscale, hh[r]) + bias4[r]);
      // (the one compiler-visible VMEM operation of an item: the vmcnt below counts it)
      if (t < kFrames)
        *reinterpret_cast<f32x4*>(p.c1 + (((int64_t)b * kFrames + t) * kC1Row + kC1Pad + rm_f0(side) + 2 * wave + (g >> 1)) * 8 +
                                  4 * (g & 1)) = v;
    }
#ifdef RM_PROF
    ps[3] = __builtin_amdgcn_s_memtime();
#endif
    if (k + 1 < n_my) convert(k + 1);  // raw item k + 1 is visible since the barrier that closed item k - 1
#ifdef RM_PROF
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ps[4] = __builtin_amdgcn_s_memtime();
#endif
    // this wave's pieces of item k + 2 must have landed before that barrier closes item k: they were issued at the start of
    // item k + 2 - kRmDepth; behind them this wave issued that item's store and, per later item, two DMA instructions and a
    // store: 1 + 3 (kRmDepth - 2) operations may stay in flight.  In the run-out (no DMA issued this item) simply drain.
    if (more)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 + 3 * (kRmDepth - 2)) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef RM_PROF
    ps[5] = __builtin_amdgcn_s_memtime();
#endif
    lds_barrier();  // item k + 1's image is complete, item k's is free, item k + 2's raw words are visible
#ifdef RM_PROF
    ps[6] = __builtin_amdgcn_s_memtime();
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 101) && (k == 10 || k == 11) && p.n_items > 1000)
      printf("RMQ wg %d wave %d k %d dma %llu kloop %llu epi %llu conv %llu vmwait %llu bar %llu total %llu\n", (int)blockIdx.x, wave, k,
             ps[1] - ps[0], ps[2] - ps[1], ps[3] - ps[2], ps[4] - ps[3], ps[5] - ps[4], ps[6] - ps[5], ps[6] - ps[0]);
#endif
  }
}

void launch_contour_conv1_rim_march(const uint32_t* zp, const void* afrag, const float* bias, float* c1, int n_windows, int n_cu,
                                    bool weights_have_lo, hipStream_t stream) {
  RimMarchParams p{zp, static_cast<const uint4*>(afrag), bias, c1, n_windows * kRmTiles};
  if (p.n_items <= 0) return;
  // one workgroup of ten waves per CU (168 VGPRs: three waves per SIMD), half of them per side; few windows: a workgroup
  // per item and side
  int per_side = n_cu / 2 > 0 ? n_cu / 2 : 1;
  if (per_side > p.n_items) per_side = p.n_items;
  const dim3 grid(2 * per_side), block(kRmThreads);
  if (weights_have_lo)
    hipLaunchKernelGGL(contour_conv1_rim_march_kernel<true>, grid, block, 0, stream, p);
  else
    hipLaunchKernelGGL(contour_conv1_rim_march_kernel<false>, grid, block, 0, stream, p);
}

}  // namespace bp
