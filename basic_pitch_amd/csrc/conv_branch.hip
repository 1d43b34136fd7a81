// Fused onset branch, workgroup form — the kernel of the fp8-correction mode (BP_FLAG_FP8_CORRECTIONS) and the A/B reference
// of the default wave-private march (onset_march.hip, round 3; BP_ONSET=ring selects this one); also home of zpack_kernel.
// (The note branch, which shared this skeleton in round 1, lives in note_march.hip since round 2; its workgroup kernel was
// retired in round 3.)
//
//   onset branch (basic_pitch/models.py:295-318): Conv2D 8->32, 5x5, strides (1,3), "same", folded BN,
//                ReLU on the harmonic stack (nn.py:69-88), Concatenate([note, features]) (305),
//                Conv2D 33->1, 3x3, "same", sigmoid                                      -> onset
//
// The branch is "conv (many taps) -> 32 channels -> conv (few taps) -> 1 channel".  The 32-channel
// intermediate (1.9 MB / window each way in the unfused kernels, conv_stride3.hip + conv_heads.hip) never
// leaves the CU here:
//
//   1. conv1 runs as an implicit GEMM in TRANSPOSED form on v_mfma_f32_32x32x16_f16:
//          C1[channel (32 rows)][pixel (32 cols)] = W1[channel][k] x patch[k][pixel]
//      A = weights (resident in VGPRs for the whole kernel), B = one ds_read_b128 per lane from an LDS
//      "image" whose 16-byte slots hold the 8 k-values a lane needs (note: 8 adjacent contour bins of one
//      frame; onset: the 8 harmonic-stack channels of one bin).  Operands are split x = hi + lo (two
//      f16, lo stored * 2^11 — bp_common.h), products hi*hi + (lo*hi + hi*lo) * 2^-11 accumulate in fp32:
//      fp32-class accuracy at the f16 matrix rate.
//   2. The C layout of a 32x32 MFMA (col = lane & 31, row = (r&3) + 8(r>>2) + 4(lane>>5)) is, up to a
//      permutation of K that is folded into the packed conv2 weights, exactly the B-operand layout of the
//      next MFMA.  So bias + ReLU + hi/lo split happen in registers and feed the "tap projection"
//          P[tap (rows)][pixel] = W2[tap][channel] x relu(C1)[channel][pixel]          (K = 32 channels)
//      with no data movement at all.
//   3. conv2's spatial part is then only a shifted sum of P: per pixel KH2*3 adds instead of
//      32*KH2*3 FMAs.  The packed conv2 weights order the taps so that a lane holds the three dw projections of
//      its pixel; Q[row][dt][w] = sum_dw P[row][w+dw-1][dt,dw] is two whole-wave DPP lane shifts in registers,
//      kept in a ring of rows, and out[t][w] = sigmoid(b + sum_dt Q[t+dt-PH2][dt][w]).
//   4. A workgroup walks a time chunk of one window 4 conv1-rows at a time with ring buffers for the
//      image and for Q: no halo recompute inside a chunk (two chunks per window -> 512 work items at
//      B = 256; 2 workgroups per CU overlap one's staging with the other's MFMAs).
//
// Roofline: bound = f16 MFMA issue.  Algorithmic work per window: note 47.5 + 20.3 MFLOP, onset
// 193.7 + 9.0 MFLOP.  HBM bytes per window: note 181,632 read + 60,544
// written; onset 214,656 (zp) + 60,544 (note) read + 60,544 written.
#include <stdio.h>
#include <stdlib.h>

#include "bp_common.h"

namespace bp {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

__device__ __forceinline__ void split_f16(float v, _Float16& hi, _Float16& lo) {
  hi = (_Float16)v;
  lo = (_Float16)((v - (float)hi) * kLoScale);
}

#ifdef BP_AB_KERNELS  // the workgroup branch kernel (BP_ONSET=ring) and its fp8-correction variant (BP_FLAG_FP8_CORRECTIONS): A/B library only since round 6
constexpr int kBrThreads = 256;
constexpr int kBrRows = 4;                               // conv1 rows per phase
constexpr int kBrTilesPerRow = 3;                        // 32-pixel tiles, 30 inner pixels each
static_assert(kBrTilesPerRow * 30 >= kFreqN, "tiles cover a row");

struct BranchParams {
  const uint4* wfrag;  // [A1 hi: KS1*64][A1 lo: KS1*64][A2 hi: 2*64][A2 lo: 2*64] x (8 x f16)
  const float* wf32;   // bias1[32], extra[9] (onset: taps of the note channel), bias2
  const void* src;     // note: contour f32 [n][172][264]; onset: zp u32 [n][kZRowsP][kZRow] (padded, bp_common.h)
  const float* note;   // onset only: note posteriorgram [n][172][88]
  float* out;          // [n][172][88]
  int n_windows;
  unsigned long long* prof;  // tools only: per-phase reference-clock totals of block 0 (null in production)
  const uint4* wmx;    // MX kernels: [kMxSteps][64 lanes][2] x 16 bytes of fp8 conv1 corrections, then [64] E8M0 scales
};

// ---- branch descriptions ----------------------------------------------------------------------
struct OnsetBr {
  static constexpr bool kOnset = true;
  static constexpr int KS1 = 13;            // conv1 k-steps: (tap pair of the 5x5 window) x 8 channels
  static constexpr int PH1 = 2;             // ONNX pads [2,1,2,1]
  static constexpr int ND = 6;              // tap 25 (dt = 5, dw = 0) is a zero-weight dummy
  static constexpr int KH2 = 3, PH2 = 1;
  static constexpr int SLOTS = kFreqC + 2;  // slot (row, s) = stack bin s-1, 8 channels; bins -1 and 264 zero
  static constexpr int RING = kBrRows + 2 * PH1;
  static constexpr int QRING = kBrRows + 2 * PH2;
  static constexpr int PIECE = 2;           // rows per staging call (3 tasks of 8 loads per thread)
  static constexpr int DT0 = 2;
  static constexpr int RAW_W0 = 20;         // first zp word of a row brought in by LDS-DMA: bin -36 (= kZPadL - 36)
  static constexpr int RAW_ROW = 101;       // 16-byte units: words 20 .. 423 (bins -36 .. 263 + 101 and the tail)
  static constexpr int RAW_PAD = 0;
  static constexpr int RAW_UNITS = RAW_PAD + kBrRows * RAW_ROW;
#ifndef BP_ONSET_CHUNKS
#define BP_ONSET_CHUNKS 2
#endif
  static constexpr int CHUNKS = BP_ONSET_CHUNKS;  // time chunks per window (work items = windows x CHUNKS)
  static constexpr int WGS = 2;
  static __device__ constexpr int d_of(int s, int h) { return (2 * s + h) / 5; }
  static __device__ constexpr int x_of(int s, int h) { return (2 * s + h) % 5; }
  static __device__ __forceinline__ int lane_slot(int wc) { return 3 * wc; }  // bin 3w+dw-1 -> slot 3w+dw
};


// ---- fp8 planes of the MX variant (the correction products lo_w a + hi_w lo_a of conv1 on
// v_mfma_scale_f32_32x32x64_f8f6f4, see conv_contour_fold_mx.hip): a slot of the second image then holds
// [fp8(a 2^6) x 8 channels | fp8(lo_a 2^6) x 8 channels] instead of 8 f16 lo parts — the same 16 bytes.
using s16x2 = __attribute__((ext_vector_type(2))) short;
using h16x2 = __attribute__((ext_vector_type(2))) _Float16;
using i32x8 = __attribute__((ext_vector_type(8))) int;
constexpr int kMxSA = 6;      // |z| <= 1.61 (BN of a [0, 1] map, checked in bp_create): a 2^6 <= 103 < 448
constexpr int kMxSteps = 7;   // onset conv1: 28 taps (25 + 3 zero) in steps of 4 taps x 8 channels
__device__ __forceinline__ uint32_t fp8x4_of_f16x4(uint32_t pair01, uint32_t pair23) {
  s16x2 r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, __builtin_bit_cast(h16x2, pair01), 1.0f / (float)(1 << kMxSA), false);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, __builtin_bit_cast(h16x2, pair23), 1.0f / (float)(1 << kMxSA), true);
  return __builtin_bit_cast(uint32_t, r);
}
// (hi f16 x 8, lo f16 x 8) of a slot -> its fp8 form
__device__ __forceinline__ uint4 fp8_slot(const uint4 vh, const uint4 vl) {
  return uint4{fp8x4_of_f16x4(vh.x, vh.y), fp8x4_of_f16x4(vh.z, vh.w), fp8x4_of_f16x4(vl.x, vl.y),
               fp8x4_of_f16x4(vl.z, vl.w)};
}

// ---- image staging: `nrows` rows starting at `row_first` (absolute frame index, may be outside the window).
// Tasks (row, slot) are dealt round-robin to the 256 threads; a thread first ISSUES the loads of all its tasks
// (NT x 8 in flight), then splits / packs and writes them: the global-load latency is paid once per call, not once
// per task.  Slots that are zero for every row (onset: the two padding bins of "same") are written by init_rows.
template <class Br, int NROWS, bool MX = false>
__device__ __forceinline__ void stage_rows(const BranchParams& p, int b, int row_first,
                                           uint4* __restrict__ img_hi, uint4* __restrict__ img_lo, int tid) {
  constexpr int PER_ROW = Br::kOnset ? kFreqC : Br::SLOTS;       // tasks per row
  constexpr int NT = (NROWS * PER_ROW + kBrThreads - 1) / kBrThreads;
  constexpr int ntask = NROWS * PER_ROW;
  static_assert(Br::kOnset, "the onset branch is the only tenant");
  {
    // stack bin f -> slot f + 1; zp is zero outside the CQT and in its pad frames -1 / 172 (bp_common.h): only rows
    // further outside the window need a guard
    uint32_t u[NT][8];
    int dst[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      const int e = tid + k * kBrThreads;
      dst[k] = -1;
      if (e < ntask) {
        const int rr = e / PER_ROW, f = e - rr * PER_ROW;
        const int row = row_first + rr;
        const bool rvalid = row >= -1 && row <= kFrames;
        const uint32_t* src = static_cast<const uint32_t*>(p.src) + (int64_t)b * kZWin +
                              (int64_t)((rvalid ? row : -1) + 1) * kZRow + kZPadL + f;
        dst[k] = ((row + 64 * Br::RING) % Br::RING) * Br::SLOTS + f + 1;
#pragma unroll
        for (int c = 0; c < 8; ++c) u[k][c] = src[harm_shift(c)];  // row -1 is all zero: also serves rows < -1
      }
    }
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      if (dst[k] < 0) continue;
      uint4 vh, vl;
      vh.x = (u[k][0] & 0xffffu) | (u[k][1] << 16);
      vh.y = (u[k][2] & 0xffffu) | (u[k][3] << 16);
      vh.z = (u[k][4] & 0xffffu) | (u[k][5] << 16);
      vh.w = (u[k][6] & 0xffffu) | (u[k][7] << 16);
      vl.x = (u[k][0] >> 16) | (u[k][1] & 0xffff0000u);
      vl.y = (u[k][2] >> 16) | (u[k][3] & 0xffff0000u);
      vl.z = (u[k][4] >> 16) | (u[k][5] & 0xffff0000u);
      vl.w = (u[k][6] >> 16) | (u[k][7] & 0xffff0000u);
      img_hi[dst[k]] = vh;
      img_lo[dst[k]] = MX ? fp8_slot(vh, vl) : vl;
    }
  }
}

// `NROWS` rows in pieces of Br::PIECE rows: bounds the registers a staging call holds in flight
template <class Br, int NROWS, bool MX = false>
__device__ __forceinline__ void stage_block(const BranchParams& p, int b, int row_first, uint4* __restrict__ img_hi,
                                            uint4* __restrict__ img_lo, int tid) {
  constexpr int P = Br::PIECE;
#pragma unroll
  for (int r = 0; r + P <= NROWS; r += P) stage_rows<Br, P, MX>(p, b, row_first + r, img_hi, img_lo, tid);
  if constexpr (NROWS % P != 0)
    stage_rows<Br, NROWS % P, MX>(p, b, row_first + NROWS - NROWS % P, img_hi, img_lo, tid);
}

// ---- steady-state staging: the kBrRows source rows of the NEXT phase come in by LDS-DMA (global_load_lds_dwordx4:
// lane l's 16 bytes land at the wave's LDS base + 16 l, checked in tools/ubench/lds_dma.hip) while the tiles of this
// phase run, and are split / gathered LDS -> LDS after the barrier: no global-load latency on the phase's critical path.
// 16 bytes per lane, global -> LDS at `lds_wave_base` + 16 * lane (wave-uniform base), asynchronous (vmcnt).
// Issued as inline assembly on purpose: through the builtin the compiler's wait-count pass knows an LDS-writing VMEM
// operation is in flight and, unable to prove that `raw` does not alias the image, puts s_waitcnt vmcnt(0) in front of
// the next LDS read — the first tile of every phase then waited for the whole DMA (the latency this staging exists to
// hide).  The kernel orders the DMA itself: s_waitcnt vmcnt(0) + barrier before raw_convert reads `raw`.  (Compiler
// generated vmcnt(N) waits for its own loads stay correct: VMEM returns in issue order, an extra outstanding operation
// only makes them wait longer.)
__device__ __forceinline__ void lds_dma16(const void* gsrc, uint4* lds_wave_base) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(lds_wave_base));
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(base) : "memory");
#endif
}

template <class Br>
__device__ __forceinline__ void raw_dma_issue(const BranchParams& p, int b, int row_first, uint4* raw, int wave,
                                              int lane) {
  // one source row per wave (kBrRows = 4 waves): RAW_ROW units in ceil(RAW_ROW / 64) instructions, no per-lane
  // division
  static_assert(kBrRows == kBrThreads / 64, "one row per wave");
  const int row = row_first + wave;  // wave-uniform
  const float* src;
  if constexpr (Br::kOnset) {
    // rows outside [-1, 172] read the all-zero pad row -1 of the padded window (bp_common.h)
    const bool rvalid = row >= -1 && row <= kFrames;
    src = reinterpret_cast<const float*>(static_cast<const uint32_t*>(p.src) + (int64_t)b * kZWin +
                                         (int64_t)((rvalid ? row : -1) + 1) * kZRow + Br::RAW_W0);
  }
  uint4* dst = raw + Br::RAW_PAD + wave * Br::RAW_ROW;
#pragma unroll
  for (int u0 = 0; u0 < Br::RAW_ROW; u0 += 64) {
    if (u0 + 64 <= Br::RAW_ROW || u0 + lane < Br::RAW_ROW) lds_dma16(src + 4 * (u0 + lane), dst + u0);
  }
}

template <class Br, bool MX = false>
__device__ __forceinline__ void raw_convert(int row_first, const uint4* raw_, uint4* __restrict__ img_hi,
                                            uint4* __restrict__ img_lo, int tid) {
  // the DMA's LDS writes are invisible to the optimiser: read behind a memory clobber, at a laundered OFFSET (laundering the
  // pointer itself loses its address space: the reads became flat_load_dword — round 5, found in conv_contour_rim_march.hip)
  int zoff = 0;
  asm volatile("" : "+v"(zoff)::"memory");
  const uint4* raw = raw_ + zoff;
  constexpr int PER_ROW = Br::kOnset ? kFreqC : kFreqN;
  constexpr int ntask = kBrRows * PER_ROW;
#pragma unroll 1
  for (int e = tid; e < ntask; e += kBrThreads) {
    const int rr = e / PER_ROW, f = e - rr * PER_ROW;
    const int row = row_first + rr;
    uint4 vh{0u, 0u, 0u, 0u}, vl{0u, 0u, 0u, 0u};
    if constexpr (Br::kOnset) {
      const uint32_t* words = reinterpret_cast<const uint32_t*>(raw + Br::RAW_PAD + rr * Br::RAW_ROW) +
                              (kZPadL - Br::RAW_W0) + f;
      uint32_t u[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) u[c] = words[harm_shift(c)];
      vh.x = (u[0] & 0xffffu) | (u[1] << 16);
      vh.y = (u[2] & 0xffffu) | (u[3] << 16);
      vh.z = (u[4] & 0xffffu) | (u[5] << 16);
      vh.w = (u[6] & 0xffffu) | (u[7] << 16);
      vl.x = (u[0] >> 16) | (u[1] & 0xffff0000u);
      vl.y = (u[2] >> 16) | (u[3] & 0xffff0000u);
      vl.z = (u[4] >> 16) | (u[5] & 0xffff0000u);
      vl.w = (u[6] >> 16) | (u[7] & 0xffff0000u);
      const int dst = ((row + 64 * Br::RING) % Br::RING) * Br::SLOTS + f + 1;
      img_hi[dst] = vh;
      img_lo[dst] = MX ? fp8_slot(vh, vl) : vl;
    }
  }
}

// WLO = false: conv1 weights without a lo part (BP_FLAG_BF16_WEIGHTS): 2 MFMAs per k-step
// MX = true (onset, WLO): conv1's two correction products on the block-scaled fp8 instruction, everything in ONE
// accumulator: per tile 13 f16 + 7 fp8 matrix instructions (864 pipe cycles) instead of 39 f16 ones (1248)
template <class Br, bool WLO, bool PROF = false, bool MX = false>
__global__ __launch_bounds__(kBrThreads, Br::WGS) void branch_kernel(BranchParams p) {
  static_assert(!MX || (Br::kOnset && WLO), "the fp8-correction variant exists for the onset branch");
  unsigned long long acc_t[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long t_prev = PROF ? __builtin_readcyclecounter() : 0;
#define BR_STAMP(k)                                                \
  if (PROF) {                                                      \
    const unsigned long long t_now = __builtin_readcyclecounter(); \
    acc_t[k] += t_now - t_prev;                                    \
    t_prev = t_now;                                                \
  }
  constexpr int KS1 = Br::KS1, KH2 = Br::KH2, PH1 = Br::PH1, PH2 = Br::PH2;
  __shared__ __attribute__((aligned(16))) uint4 img_hi[Br::RING * Br::SLOTS];
  __shared__ __attribute__((aligned(16))) uint4 img_lo[Br::RING * Br::SLOTS];
  __shared__ float qring[Br::QRING * KH2 * kFreqN + 64];  // + a scratch slot per lane for the stores that must not land
  __shared__ __attribute__((aligned(16))) uint4 raw[Br::RAW_UNITS];
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int h = lane >> 5, li = lane & 31;

  // resident A operands (weights) and biases
  uint4 a1h[KS1], a1l[MX ? 1 : KS1], a2h[2], a2l[2];
  i32x8 amx[MX ? kMxSteps : 1];
  int amx_scale = 127;
#pragma unroll
  for (int s = 0; s < KS1; ++s) {
    a1h[s] = p.wfrag[s * 64 + lane];
    if (!MX) a1l[s] = p.wfrag[(KS1 + s) * 64 + lane];
  }
  if constexpr (MX) {
#pragma unroll
    for (int S = 0; S < kMxSteps; ++S) {
      const uint4 m0 = p.wmx[(S * 64 + lane) * 2], m1 = p.wmx[(S * 64 + lane) * 2 + 1];
      amx[S] = i32x8{(int)m0.x, (int)m0.y, (int)m0.z, (int)m0.w, (int)m1.x, (int)m1.y, (int)m1.z, (int)m1.w};
    }
    amx_scale = reinterpret_cast<const int*>(p.wmx + kMxSteps * 64 * 2)[lane];
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    a2h[s] = p.wfrag[(2 * KS1 + s) * 64 + lane];
    a2l[s] = p.wfrag[(2 * KS1 + 2 + s) * 64 + lane];
  }
  float bias1[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bias1[r] = p.wf32[(r & 3) + 8 * (r >> 2) + 4 * h];
  // onset: the 3x3 taps of the note channel (concat channel 0) for the frame taps this lane half owns
  constexpr int DT0 = Br::DT0;
  float extra[DT0][3];
#pragma unroll
  for (int i = 0; i < DT0; ++i)
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) {
      const int dt = DT0 * h + i;
      extra[i][dw] = (Br::kOnset && dt < KH2) ? p.wf32[32 + dt * 3 + dw] : 0.0f;
    }
  const float bias2 = p.wf32[41];
  // every load above has landed before the loops: with loads pending from the preheader the wait-count pass puts a
  // conservative s_waitcnt vmcnt(1) in front of each tile's first MFMA, which also waits for the (untracked) LDS-DMA
  __builtin_amdgcn_s_waitcnt(0x0F70);

  // slots no staging call writes (onset: the two zero bins either side of a row) are zero from here on
  for (int i = threadIdx.x; i < Br::RING * Br::SLOTS; i += kBrThreads) {
    img_hi[i] = uint4{0u, 0u, 0u, 0u};
    img_lo[i] = uint4{0u, 0u, 0u, 0u};
  }

  const int n_items = p.n_windows * Br::CHUNKS;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int b = item / Br::CHUNKS;
    const int ci = item - b * Br::CHUNKS;
    const int T0 = (ci * kFrames) / Br::CHUNKS;
    const int T1 = ((ci + 1) * kFrames) / Br::CHUNKS;
    const int n_phase = (T1 - T0 + 2 * PH2 + kBrRows - 1) / kBrRows;

    __syncthreads();  // previous item finished with the rings
    stage_block<Br, Br::RING, MX>(p, b, T0 - PH2 - PH1, img_hi, img_lo, threadIdx.x);
    // onset: the note values (concat channel 0, models.py:305) of this lane's pixels in the phase's three tiles are
    // fetched one phase ahead into registers, right in front of a vmcnt(0) wait that exists anyway: no global load is
    // left inside the tile loop (one there makes the compiler wait on vmcnt in front of every tile's MFMAs, and with
    // it on the LDS-DMA in flight).  Unconditional loads from clamped addresses, masked where they are used.
    float note_nx[kBrTilesPerRow] = {0.0f, 0.0f, 0.0f};
    auto fetch_notes = [&](int r_first) {
      if constexpr (Br::kOnset) {
#pragma unroll
        for (int j = 0; j < kBrTilesPerRow; ++j) {
          const int tile = wave + 4 * j;
          int row = r_first + tile / kBrTilesPerRow;
          row = row < 0 ? 0 : (row > kFrames - 1 ? kFrames - 1 : row);
          int w = (tile % kBrTilesPerRow) * 30 - 1 + li;
          w = w < 0 ? 0 : (w > kFreqN - 1 ? kFreqN - 1 : w);
          note_nx[j] = p.note[((int64_t)b * kFrames + row) * kFreqN + w];
        }
      }
    };
    fetch_notes(T0 - PH2);
    __syncthreads();

    for (int ph = 0; ph < n_phase; ++ph) {
      const int r0 = T0 - PH2 + kBrRows * ph;  // first conv1 row of this phase
      BR_STAMP(0);
      // the next phase's source rows start their way into LDS now (raw was consumed before the last barrier)
      if (ph + 1 < n_phase) raw_dma_issue<Br>(p, b, r0 + kBrRows + PH1, raw, wave, lane);
      float note_cur[kBrTilesPerRow];
#pragma unroll
      for (int j = 0; j < kBrTilesPerRow; ++j) note_cur[j] = note_nx[j];

      // ---- conv1 + projection, 12 tiles: 4 rows x 3 overlapping 32-pixel tiles (30 inner pixels each)
#pragma unroll 1
      for (int j = 0; j < kBrTilesPerRow; ++j) {
        const int tile = wave + 4 * j;
        const int row = r0 + tile / kBrTilesPerRow;
        const int wbase = (tile % kBrTilesPerRow) * 30 - 1;
        const int w = wbase + li;
        const bool wvalid = w >= 0 && w < kFreqN;
        const int wc = w < 0 ? 0 : (w >= kFreqN ? kFreqN - 1 : w);
        const bool rvalid = row >= 0 && row < kFrames;  // wave-uniform
        const int qoff = ((row + 64 * Br::QRING) % Br::QRING) * (KH2 * kFreqN);
        float* qrow = qring + qoff;

        if (rvalid) {
          int rb[Br::ND];
#pragma unroll
          for (int d = 0, r = (row - PH1 + 64 * Br::RING) % Br::RING; d < Br::ND; ++d) {  // one modulo, then wrap
            rb[d] = r * Br::SLOTS;
            r = r + 1 == Br::RING ? 0 : r + 1;
          }
          const int lane_off = Br::lane_slot(wc);
          float note_c = 0.0f;
          if constexpr (Br::kOnset) note_c = j == 0 ? note_cur[0] : (j == 1 ? note_cur[1] : note_cur[2]);

          f32x16 acc, accc;  // the hi x hi chain starts from the bias
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            acc[r] = bias1[r];
            accc[r] = 0.0f;
          }
          uint32_t b2hw[8], b2lw[8];
          if constexpr (!MX) {
            // image fragments are read kBrPf k-steps ahead of the MFMAs that consume them (the compiler's own
            // schedule waits for every read right after issuing it)
            constexpr int kBrPf = KS1 < 3 ? KS1 : 3;
            f16x8 bhf[KS1], blf[KS1];
            auto issue = [&](int s) {
              const int o0 = rb[Br::d_of(s, 0)] + Br::x_of(s, 0);
              const int o1 = rb[Br::d_of(s, 1)] + Br::x_of(s, 1);
              const int slot = lane_off + (h ? o1 : o0);
              bhf[s] = __builtin_bit_cast(f16x8, img_hi[slot]);
              blf[s] = __builtin_bit_cast(f16x8, img_lo[slot]);
            };
#pragma unroll
            for (int s = 0; s < kBrPf; ++s) issue(s);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < KS1; ++s) {
              if (s + kBrPf < KS1) issue(s + kBrPf);
              __builtin_amdgcn_sched_barrier(0);
              const f16x8 ah = __builtin_bit_cast(f16x8, a1h[s]);
              const f16x8 al = __builtin_bit_cast(f16x8, a1l[s]);
              if (WLO) accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bhf[s], accc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bhf[s], acc, 0, 0, 0);
              accc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, blf[s], accc, 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
            }
            // ReLU, split, and the tap projection (two values per VALU operation where the packed-f32 pipe has one)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              f32x2 v = __builtin_elementwise_fma(f32x2{accc[r], accc[r + 1]}, f32x2{kLoUnscale, kLoUnscale},
                                                  f32x2{acc[r], acc[r + 1]});
              v.x = fmaxf(v.x, 0.0f);
              v.y = fmaxf(v.y, 0.0f);
              split_f16x2(v, b2hw[r >> 1], b2lw[r >> 1]);
            }
          } else {
            // block S = k-steps 2 S, 2 S + 1 on the f16 instruction (hi x hi) + one block-scaled fp8 instruction for the
            // corrections of the same four taps: lane half h holds taps 4 S + h (bytes 0..15: fp8(a) | fp8(lo_a) of its 8
            // channels) and 4 S + 2 + h (bytes 16..31) — the slots its two hi reads use.  One E8M0 scale for both
            // correction kinds (bp_api.hip pack_onset_mx), so a 32-tap K block may mix them.  Operands one block ahead.
            constexpr int kBlocks = kMxSteps;
            constexpr int kPf = 1, kBuf = kPf + 1;  // two blocks ahead: 8 spills at the 256-register budget for 1-2 %
            f16x8 bhf[kBuf][2];
            uint4 bmf[kBuf][2];
            // half i of block S: the f16 hi operand of k-step 2 S + i and the fp8 pair of the same slot.  A wave issues
            // about one LDS read per 14 cycles in order with everything else, so the reads of block S + 1 go BETWEEN the
            // (dependent) matrix instructions of block S instead of in front of them
            auto issue = [&](int S, int i) {
              const int buf = S % kBuf;
              const int s = 2 * S + i < KS1 ? 2 * S + i : KS1 - 1;  // block 6 has one k-step: its second half is zero weights
              const int o0 = rb[Br::d_of(s, 0)] + Br::x_of(s, 0);
              const int o1 = rb[Br::d_of(s, 1)] + Br::x_of(s, 1);
              const int slot = lane_off + (h ? o1 : o0);
              if (2 * S + i < KS1) bhf[buf][i] = __builtin_bit_cast(f16x8, img_hi[slot]);
              bmf[buf][i] = img_lo[slot];
            };
            issue(0, 0);
            issue(0, 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int S = 0; S < kBlocks; ++S) {
              const int buf = S % kBuf;
              const bool more = S + 1 < kBlocks;
              acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1h[2 * S]), bhf[buf][0], acc, 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              if (more) issue(S + 1, 0);
              __builtin_amdgcn_sched_barrier(0);
              if (2 * S + 1 < KS1)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1h[2 * S + 1 < KS1 ? 2 * S + 1 : 0]),
                                                             bhf[buf][1], acc, 0, 0, 0);
              const i32x8 bm = {(int)bmf[buf][0].x, (int)bmf[buf][0].y, (int)bmf[buf][0].z, (int)bmf[buf][0].w,
                                (int)bmf[buf][1].x, (int)bmf[buf][1].y, (int)bmf[buf][1].z, (int)bmf[buf][1].w};
              acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(amx[S], bm, acc, 0, 0, 0, amx_scale, 0, 127 - kMxSA);
              __builtin_amdgcn_sched_barrier(0);
              if (more) issue(S + 1, 1);
              __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              f32x2 v{relu_f32(acc[r]), relu_f32(acc[r + 1])};
              split_f16x2(v, b2hw[r >> 1], b2lw[r >> 1]);
            }
          }
          f16x8 b2h[2], b2l[2];
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            b2h[s] = __builtin_bit_cast(f16x8, uint4{b2hw[4 * s], b2hw[4 * s + 1], b2hw[4 * s + 2], b2hw[4 * s + 3]});
            b2l[s] = __builtin_bit_cast(f16x8, uint4{b2lw[4 * s], b2lw[4 * s + 1], b2lw[4 * s + 2], b2lw[4 * s + 3]});
          }
          f32x16 pp, ppc;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            pp[r] = 0.0f;
            ppc[r] = 0.0f;
          }
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const f16x8 ah = __builtin_bit_cast(f16x8, a2h[s]);
            const f16x8 al = __builtin_bit_cast(f16x8, a2l[s]);
            pp = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b2h[s], pp, 0, 0, 0);
            ppc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b2h[s], ppc, 0, 0, 0);
            ppc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b2l[s], ppc, 0, 0, 0);
          }
          // The packed conv2 weights put tap (dt, dw) in C row r = 3 (dt - DT0 h) + dw of lane half h = (dt >= DT0)
          // (bp_api.hip pack_branch), so a lane holds all three dw projections of its pixel for its frame taps, and
          //   Q[row][dt][w] = (P[dt,0][w-1] + P[dt,1][w]) + P[dt,2][w+1]
          // is two whole-wave lane shifts (DPP wave_shr / wave_shl; the tile's edge pixels li = 0, 31 are halo and take
          // garbage from the other half) - no LDS round trip.  The onset head adds the note channel of the concat.
          auto from_left = [](float v) {  // value of lane - 1
            return __builtin_bit_cast(float,
                                      __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
          };
          auto from_right = [](float v) {  // value of lane + 1
            return __builtin_bit_cast(float,
                                      __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
          };
          float n_c = 0.0f, n_l = 0.0f, n_r = 0.0f;
          if constexpr (Br::kOnset) {
            n_c = wvalid ? note_c : 0.0f;
            n_l = from_left(n_c);
            n_r = from_right(n_c);
          }
          const bool store_ok = li >= 1 && li <= 30 && w < kFreqN;
#pragma unroll
          for (int i = 0; i < DT0; ++i) {
            // pixels outside the row are conv2's zero padding: they only matter as the neighbours of an edge pixel
            float p0 = pp[3 * i] + ppc[3 * i] * kLoUnscale;
            const float p1 = pp[3 * i + 1] + ppc[3 * i + 1] * kLoUnscale;
            float p2 = pp[3 * i + 2] + ppc[3 * i + 2] * kLoUnscale;
            p0 = wvalid ? p0 : 0.0f;
            p2 = wvalid ? p2 : 0.0f;
            float q = (from_left(p0) + p1) + from_right(p2);
            if constexpr (Br::kOnset) q += (n_l * extra[i][0] + n_c * extra[i][1]) + n_r * extra[i][2];
            const int dt = DT0 * h + i;
            // unconditional store: lanes that must not write (halo pixels, the frame taps half 1 does not own) aim at a
            // scratch slot — an exec-mask branch per store costs more issue slots than the select
            const int at = (store_ok && dt < KH2) ? qoff + dt * kFreqN + w : Br::QRING * KH2 * kFreqN + lane;
            qring[at] = q;
          }
        } else {
          // conv1 row outside the window: conv2 sees zeros there
          for (int idx = lane; idx < 30 * KH2; idx += 64) {
            const int dt = idx / 30;
            const int wq = wbase + 1 + idx - 30 * dt;
            if (wq < kFreqN) qrow[dt * kFreqN + wq] = 0.0f;
          }
        }
      }
      BR_STAMP(1);
      if (ph + 1 < n_phase) fetch_notes(r0 + kBrRows);
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's share of the DMA (and the note values) has landed
      lds_barrier();
      BR_STAMP(2);

      // ---- output rows r0-PH2 .. r0-PH2+3 (one per wave), then the next 4 image rows
      {
        const int t = r0 - PH2 + wave;
        if (t >= T0 && t < T1) {
          int qb[KH2];
          int qr = (t - PH2 + 64 * Br::QRING) % Br::QRING;
#pragma unroll
          for (int dt = 0; dt < KH2; ++dt)
            qb[dt] = (qr * KH2 + dt) * kFreqN, qr = qr + 1 == Br::QRING ? 0 : qr + 1;
          float* orow = p.out + ((int64_t)b * kFrames + t) * kFreqN;
          for (int w = lane; w < kFreqN; w += 64) {
            float s = 0.0f;
#pragma unroll
            for (int dt = 0; dt < KH2; ++dt) s += qring[qb[dt] + w];
            orow[w] = sigmoidf_fast(s + bias2);
          }
        }
      }
      BR_STAMP(3);
      if (ph + 1 < n_phase) raw_convert<Br, MX>(r0 + kBrRows + PH1, raw, img_hi, img_lo, threadIdx.x);
      BR_STAMP(4);
      lds_barrier();
      BR_STAMP(5);
    }
  }
  if (PROF && blockIdx.x == 0 && lane == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) p.prof[wave * 6 + k] = acc_t[k];
  }
#undef BR_STAMP
}
#endif  // BP_AB_KERNELS

// ---- z pack: NormalizedLog tail + BatchNorm affine, stored pre-split as (f16 hi | f16 lo << 16) ----
// (signal.py:177-183, models.py:187-189).  One pass over lp; consumers gather these words straight
// into MFMA operand slots with no further arithmetic.
// The window's extrema come either from mm (the FILTERBANK stage's output, test hook) or — the production path — from
// the filterbank's per-tile partial extrema, folded here by every workgroup for itself (n_partials float2 per window,
// 3 KB from L2): a separate reduction kernel costs a launch boundary (~5 us) for 5 us of work.
__global__ __launch_bounds__(256) void zpack_kernel(const float* __restrict__ lp, const int* __restrict__ mm,
                                                    const float2* __restrict__ mmp, int n_partials,
                                                    uint32_t* __restrict__ zp, LogConsts kc, int n_bins) {
  const int b = blockIdx.y;
  float mn, mx;
  if (mmp) {
    __shared__ float2 red[4];
    float vmin = __int_as_float(0x7f800000), vmax = -__int_as_float(0x7f800000);
    for (int i = threadIdx.x; i < n_partials; i += 256) {
      const float2 pr = mmp[(int64_t)b * n_partials + i];
      vmin = fminf(vmin, pr.x);
      vmax = fmaxf(vmax, pr.y);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      vmin = fminf(vmin, __shfl_xor(vmin, o));
      vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = make_float2(vmin, vmax);
    __syncthreads();
    mn = fminf(fminf(red[0].x, red[1].x), fminf(red[2].x, red[3].x));
    mx = fmaxf(fmaxf(red[0].y, red[1].y), fmaxf(red[2].y, red[3].y));
  } else {
    mn = ord2f(mm[2 * b]);
    mx = ord2f(mm[2 * b + 1]);
  }
  const float nk = norm_scale(mn, mx, kc);
  const float* lpb = lp + (int64_t)b * kFrames * n_bins;
  uint32_t* zb = zp + (int64_t)b * kZWin;
  // the whole padded window is written every time (pad frames and pad words are zero: the zero padding of the
  // harmonic stack, nn.py:73-85, and of the convolutions' frame halo); four words per thread, one 16-byte store
  static_assert(kZRow % 4 == 0 && kZPadL % 4 == 0, "a thread's four words stay in one row");
  for (int i4 = blockIdx.x * 256 + threadIdx.x; i4 < kZWin / 4; i4 += gridDim.x * 256) {
    const int tp = i4 / (kZRow / 4), g0 = 4 * (i4 - tp * (kZRow / 4)) - kZPadL;
    const int t = tp - 1;
    uint32_t u[4] = {0u, 0u, 0u, 0u};
    if (t >= 0 && t < kFrames && g0 + 3 >= 0 && g0 < n_bins) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int g = g0 + e;
        if (g >= 0 && g < n_bins) {
          const float z = norm_bn_k(lpb[t * n_bins + g], mn, nk, kc.bn_b);
          _Float16 hi, lo;
          split_f16(z, hi, lo);
          u[e] = (uint32_t)__builtin_bit_cast(unsigned short, hi) | ((uint32_t)__builtin_bit_cast(unsigned short, lo) << 16);
        }
      }
    }
    *reinterpret_cast<uint4*>(zb + 4 * i4) = uint4{u[0], u[1], u[2], u[3]};
  }
}

void launch_zpack(const float* lp, const int* mm, uint32_t* zp, int n_windows, LogConsts kc, int n_bins,
                  hipStream_t stream) {
  hipLaunchKernelGGL(zpack_kernel, dim3(16, n_windows), dim3(256), 0, stream, lp, mm, nullptr, 0, zp, kc, n_bins);
}

// production path: extrema folded from the filterbank's partials (`scratch` of launch_filterbank_mfma(..., mm = null))
void launch_zpack_partials(const float* lp, const float* scratch, int n_partials, uint32_t* zp, int n_windows,
                           LogConsts kc, int n_bins, hipStream_t stream) {
  hipLaunchKernelGGL(zpack_kernel, dim3(16, n_windows), dim3(256), 0, stream, lp, nullptr,
                     reinterpret_cast<const float2*>(scratch), n_partials, zp, kc, n_bins);
}

#ifdef BP_AB_KERNELS
template <class Br>
static void launch_branch(const BranchParams& p, int n_cu, bool weights_have_lo, hipStream_t stream) {
  const int items = p.n_windows * Br::CHUNKS;
  const int grid = items < Br::WGS * n_cu ? items : Br::WGS * n_cu;
#ifdef BP_AB_KERNELS  // tools only (A/B builds): phase profile of block 0 to stderr
  static const bool prof = ab_env("BP_BRANCH_PROF") != nullptr;
  if (prof) {
    BranchParams q = p;
    unsigned long long hbuf[24];
    int resident = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, branch_kernel<Br, true>, kBrThreads, 0);
    fprintf(stderr, "brprof %s: %d workgroups resident per CU\n", Br::kOnset ? "onset" : "note", resident);
    if (hipMalloc(&q.prof, sizeof hbuf) != hipSuccess) return;
    (void)hipMemsetAsync(q.prof, 0, sizeof hbuf, stream);
    if (p.wmx)
      hipLaunchKernelGGL((branch_kernel<Br, true, true, true>), dim3(grid), dim3(kBrThreads), 0, stream, q);
    else
      hipLaunchKernelGGL((branch_kernel<Br, true, true>), dim3(grid), dim3(kBrThreads), 0, stream, q);
    (void)hipMemcpyAsync(hbuf, q.prof, sizeof hbuf, hipMemcpyDeviceToHost, stream);
    (void)hipStreamSynchronize(stream);
    (void)hipFree(q.prof);
    for (int w = 0; w < 4; ++w) {
      fprintf(stderr, "brprof %s wave %d:", Br::kOnset ? "onset" : "note", w);
      for (int k = 0; k < 6; ++k) fprintf(stderr, " %llu", hbuf[w * 6 + k]);
      fprintf(stderr, "\n");
    }
    return;
  }
#endif
  if constexpr (Br::kOnset) {
    if (weights_have_lo && p.wmx) {
      hipLaunchKernelGGL((branch_kernel<Br, true, false, true>), dim3(grid), dim3(kBrThreads), 0, stream, p);
      return;
    }
  }
  if (weights_have_lo)
    hipLaunchKernelGGL((branch_kernel<Br, true>), dim3(grid), dim3(kBrThreads), 0, stream, p);
  else
    hipLaunchKernelGGL((branch_kernel<Br, false>), dim3(grid), dim3(kBrThreads), 0, stream, p);
}

// wmx: the fp8 correction fragments (pack_onset_mx) or null for the three-product f16 kernel
void launch_onset_branch(const uint32_t* zp, const float* note, const void* wfrag, const float* wf32, const void* wmx,
                         float* onset, int n_windows, int n_cu, bool weights_have_lo, hipStream_t stream) {
  BranchParams p{static_cast<const uint4*>(wfrag), wf32, zp, note, onset, n_windows, nullptr,
                 static_cast<const uint4*>(wmx)};
  launch_branch<OnsetBr>(p, n_cu, weights_have_lo, stream);
}
#endif  // BP_AB_KERNELS

}  // namespace bp
