// Contour branch, default path: two kernels with the 8-channel intermediate in HBM.
//
//   contour_conv1_kernel   Conv2D 8->8, (3 frames x 39 bins), "same", folded BN, ReLU on the harmonic stack
//                          (basic_pitch/models.py:241-250, nn.py:69-88)                          zp -> c1
//   contour_conv2_kernel   Conv2D 8->1, 5x5, "same", sigmoid, FlattenFreqCh (models.py:254-263,
//                          nn.py:105-119)                                                         c1 -> contour
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

#include "bp_common.h"

namespace bp {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int kD1Threads = 512;                    // 8 waves: two per SIMD
constexpr int kD1Ring = 12;                        // image rows resident (11 needed: 7 live + 4 incoming)
constexpr int kD1Q = 76;                           // slots per phase plane
constexpr int kD1Slots = 4 * kD1Q;                 // 304 slots per image row
constexpr int kD1LoOff = kD1Ring * kD1Slots;       // img[] = hi image, then lo image (uint4 units)
constexpr int kD1WTap = 45;                        // tap + 3 in [0, 45): 3 zero taps below, 3 above
constexpr int kD1WHalf = 3 * kD1WTap * 8;          // 1080 slots of hi weights, then 1080 of lo
constexpr int kD1Steps = 63;
constexpr int kD1Pf = 3;                           // k-steps of operand prefetch (12 reads in flight)
constexpr int kD1Groups = kFreqC / 4;              // 66 four-bin groups per frame
constexpr int kD1Round = 256;                      // positions per round (8 waves x 32)
constexpr int kD1Stage = 3;                        // image slots a thread may gather per round
static_assert(2 * kD1LoOff * 16 + 2 * kD1WHalf * 16 <= 160 * 1024, "LDS budget");
static_assert(kD1LoOff * 16 + (3 * kD1Q + 10) * 16 < 65536, "ds_read immediate offset of the lo image");
static_assert(kD1Stage * kD1Threads >= 4 * kFreqC, "a round brings in at most 4 image rows");

struct Conv1Params {
  const uint32_t* zp;   // [n][kZRowsP][kZRow] padded pre-split z (zpack_kernel)
  const uint4* wlds;    // [hi|lo][3][45][8] x (8 x f16): the LDS weight image
  const float* bias;    // [8]
  float* c1;            // [n][172][kC1Row][8] relu(conv1), 2 zero bins of padding either side of a row
  int n_windows;
  int chunks;           // row chunks per window (work items = n_windows * chunks)
};

// slot of stack bin f in [0, 264): plane (f + 20) & 3, index (f + 20) >> 2   (slot s holds bin 4 q + pl - 20)
__device__ __forceinline__ int d1_slot_of_bin(int f) { return ((f + 20) & 3) * kD1Q + ((f + 20) >> 2); }

__device__ __forceinline__ void d1_pack_put(const uint32_t (&u)[8], uint4* __restrict__ img, int idx) {
  uint4 vh, vl;
  vh.x = (u[0] & 0xffffu) | (u[1] << 16);
  vh.y = (u[2] & 0xffffu) | (u[3] << 16);
  vh.z = (u[4] & 0xffffu) | (u[5] << 16);
  vh.w = (u[6] & 0xffffu) | (u[7] << 16);
  vl.x = (u[0] >> 16) | (u[1] & 0xffff0000u);
  vl.y = (u[2] >> 16) | (u[3] & 0xffff0000u);
  vl.z = (u[4] >> 16) | (u[5] & 0xffff0000u);
  vl.w = (u[6] >> 16) | (u[7] & 0xffff0000u);
  img[idx] = vh;
  img[idx + kD1LoOff] = vl;
}

// gather the 8 harmonic-stack channels of bin f of image row `row` (zp is zero outside the CQT: no masks)
__device__ __forceinline__ void d1_gather(const uint32_t* __restrict__ zorigin, int row, int f,
                                          uint32_t (&u)[8]) {
  const uint32_t* zr = zorigin + row * kZRow + f;
#pragma unroll
  for (int c = 0; c < 8; ++c) u[c] = zr[harm_shift(c)];
}

// WLO = false: the weights have no lo part (BP_FLAG_BF16_WEIGHTS): no lo fragment reads, 2 MFMAs per k-step
template <bool WLO>
__global__ __launch_bounds__(kD1Threads, 2) void contour_conv1_kernel(Conv1Params p) {
  __shared__ __attribute__((aligned(16))) uint4 img[2 * kD1LoOff];
  __shared__ __attribute__((aligned(16))) uint4 wl[2 * kD1WHalf];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = wave_id();
  const int kh = lane >> 5, li = lane & 31;

  for (int i = tid; i < 2 * kD1WHalf; i += kD1Threads) wl[i] = p.wlds[i];
  // slots of bins outside the cropped stack (nn.py:87: crop to 264 bins, then "same" padding) stay zero forever
  for (int i = tid; i < kD1Ring * kD1Slots; i += kD1Threads) {
    const int slot = i % kD1Slots;
    const int pl = slot / kD1Q, q = slot - pl * kD1Q;
    const int f = 4 * q + pl - 20;
    if (f < 0 || f >= kFreqC) {
      img[i] = uint4{0u, 0u, 0u, 0u};
      img[i + kD1LoOff] = uint4{0u, 0u, 0u, 0u};
    }
  }
  float bias4[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bias4[q] = p.bias[4 * kh + q];
  // A operand: lane (row i = 8 j + o, half kh) reads weight slot (dt, 2 ep + kh - j + 3, o)
  const int aidx = (kh - (li >> 3) + 3) * 8 + (li & 7);

  const int rows_per = (kFrames + p.chunks - 1) / p.chunks;
  const int n_items = p.n_windows * p.chunks;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int b = item / p.chunks;
    const int t0 = (item - b * p.chunks) * rows_per;
    const int t1 = t0 + rows_per < kFrames ? t0 + rows_per : kFrames;
    const int npos = (t1 - t0) * kD1Groups;
    const int nrounds = (npos + kD1Round - 1) / kD1Round;
    const uint32_t* zorigin = p.zp + (int64_t)b * kZWin + kZRow + kZPadL;  // (frame 0, bin 0)
    float* c1b = p.c1 + (int64_t)b * kC1Win;

    __syncthreads();  // the previous item is done with the ring
    // image rows t0 - 1 .. staged_hi of round 0
    int staged_hi = t0 + (kD1Round - 1) / kD1Groups + 1;
    staged_hi = staged_hi < t1 ? staged_hi : t1;
    for (int e = tid; e < (staged_hi - t0 + 2) * kFreqC; e += kD1Threads) {
      const int ri = e / kFreqC, f = e - ri * kFreqC;
      const int row = t0 - 1 + ri;
      uint32_t u[8];
      d1_gather(zorigin, row, f, u);
      d1_pack_put(u, img, ((row + kD1Ring) % kD1Ring) * kD1Slots + d1_slot_of_bin(f));
    }
    __syncthreads();

    for (int k = 0; k < nrounds; ++k) {
      // ---- image rows to bring in during this round: (staged_hi, need_hi]
      int need_hi = t0 + (kD1Round * (k + 1) + kD1Round - 1) / kD1Groups + 1;
      need_hi = need_hi < t1 ? need_hi : t1;
      const int n_new = (k + 1 < nrounds) ? need_hi - staged_hi : 0;
      const int first_new = staged_hi + 1;
      uint32_t pf[8];
      int put_idx = 0;
      bool put_ok = false;
      auto stage_issue = [&](int i) {
        const int e = i * kD1Threads + tid;
        put_ok = e < n_new * kFreqC;
        if (put_ok) {
          const int ri = e / kFreqC, f = e - ri * kFreqC;
          const int row = first_new + ri;
          d1_gather(zorigin, row, f, pf);
          put_idx = ((row + kD1Ring) % kD1Ring) * kD1Slots + d1_slot_of_bin(f);
        }
      };
      auto stage_put = [&]() {
        if (put_ok) d1_pack_put(pf, img, put_idx);
      };

      // ---- this wave's tile: 32 consecutive positions
      const int pos = kD1Round * k + 32 * w + li;
      const bool pvalid = pos < npos;
      const int posc = pvalid ? pos : npos - 1;
      const int prr = posc / kD1Groups;
      const int pmf = posc - prr * kD1Groups;
      const int prow = t0 + prr;
      int lo_base[3], hi_base[3];
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) {
        const int rowslot = ((prow - 1 + dt + kD1Ring) % kD1Ring) * kD1Slots;
        lo_base[dt] = rowslot + pmf + kh * kD1Q;              // tap plane 1 -> 2 (same group)
        hi_base[dt] = rowslot + pmf + kh * (1 - 3 * kD1Q);    // tap plane 3 -> 0 of the next group
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
        const int sb = ((r0 == 1) ? lo_base[dt] : hi_base[dt]) + r0 * kD1Q + q0;
        if (WLO) al[s] = __builtin_bit_cast(f16x8, wl[widx + kD1WHalf]);
        bh[s] = __builtin_bit_cast(f16x8, img[sb]);
        ah[s] = __builtin_bit_cast(f16x8, wl[widx]);
        bl[s] = __builtin_bit_cast(f16x8, img[sb + kD1LoOff]);
      };
#pragma unroll
      for (int s = 0; s < kD1Pf; ++s) issue(s);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < kD1Steps; ++s) {
        if (s + kD1Pf < kD1Steps) issue(s + kD1Pf);
        // staging of the next round's rows, spread over the k-steps: gather at s = 1 + 20 i, LDS write 15 steps later
        if (s % 20 == 1 && s / 20 < kD1Stage) stage_issue(s / 20);
        if (s % 20 == 16 && s / 20 < kD1Stage) stage_put();
        __builtin_amdgcn_sched_barrier(0);
        if (WLO) xx = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh[s], xx, 0, 0, 0);
        hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh[s], hh, 0, 0, 0);
        xx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl[s], xx, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      staged_hi += n_new;

      // ---- bias + ReLU, c1 store: register r of a lane is (bin offset j = r >> 2, channel o = 4 kh + (r & 3))
      if (pvalid) {
        float* dst = c1b + ((int64_t)prow * kC1Row + kC1Pad + 4 * pmf) * 8 + 4 * kh;
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
      __syncthreads();  // the round's reads are done; the rows written for the next round are visible
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// conv2: a thread owns a strip of 4 adjacent bins and marches down a slab of frames.  Each input row it loads
// (8 bins x 8 channels = 256 contiguous bytes of the zero-padded c1 row, prefetched one row ahead) feeds the 5
// output rows it touches: 800 FMAs per 16 sixteen-byte loads, accumulators of the 5 open output rows in
// registers, weights wave-uniform (scalar loads).  No LDS, no barriers; threads are a flat index over
// (window, slab, strip), so any batch size fills whole waves.
constexpr int kD2Strips = kFreqC / 4;  // 66

struct Conv2Params {
  const float* c1;   // [n][172][kC1Row][8]
  const float* w2;   // [5 dt][5 dw][8 c]
  float bias;
  float* out;        // [n][172][264]
  int n_windows;
  int slab_rows;     // frames per slab
  int n_slabs;       // slabs per window
};

using v2f = __attribute__((ext_vector_type(2))) float;

__device__ __forceinline__ void d2_load(const float* __restrict__ src, float4 (&x)[16]) {
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = reinterpret_cast<const float4*>(src)[i];
}

// acc[d][bin] += sum_{dw, c} W2[4 - d][dw][c] * x[bin + dw][c]   for the open output rows d with live[d].
// Even and odd channels accumulate in the two halves of a float2: weight pairs (scalar registers) and channel pairs
// (adjacent registers of the 16-byte loads) are both naturally packed -> v_pk_fma_f32 without operand shuffles.
__device__ __forceinline__ void d2_fma(const float4* __restrict__ w2, const float4 (&x)[16], v2f (&acc)[5][4],
                                       const bool (&live)[5]) {
#pragma unroll
  for (int d = 0; d < 5; ++d) {
    if (!live[d]) continue;
    int wofs = (4 - d) * 10;  // LDS, same address in every lane: broadcast reads; opaque offset so the 50 reads
    asm volatile("" : "+v"(wofs));  // stay inside the loop instead of being hoisted into 200 registers
    const float4* wd = w2 + wofs;
#pragma unroll
    for (int dw = 0; dw < 5; ++dw) {
#pragma unroll
      for (int c4 = 0; c4 < 2; ++c4) {
        const float4 wv = wd[dw * 2 + c4];
        const v2f wa = {wv.x, wv.y};
        const v2f wb = {wv.z, wv.w};
#pragma unroll
        for (int bin = 0; bin < 4; ++bin) {
          const float4 xv = x[(bin + dw) * 2 + c4];
          acc[d][bin] = __builtin_elementwise_fma(wa, v2f{xv.x, xv.y}, acc[d][bin]);
          acc[d][bin] = __builtin_elementwise_fma(wb, v2f{xv.z, xv.w}, acc[d][bin]);
        }
      }
    }
  }
}

__device__ __forceinline__ void d2_emit(float* __restrict__ dst, const v2f (&a)[4], float bias) {
  float4 o;
  o.x = sigmoidf_exact((a[0].x + a[0].y) + bias);
  o.y = sigmoidf_exact((a[1].x + a[1].y) + bias);
  o.z = sigmoidf_exact((a[2].x + a[2].y) + bias);
  o.w = sigmoidf_exact((a[3].x + a[3].y) + bias);
  *reinterpret_cast<float4*>(dst) = o;
}

__device__ __forceinline__ void d2_shift(v2f (&acc)[5][4]) {
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[d][i] = acc[d + 1][i];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[4][i] = v2f{0.0f, 0.0f};
}

__global__ __launch_bounds__(256) void contour_conv2_kernel(Conv2Params p) {
  // the 200 taps live in LDS: their reads count on lgkmcnt, so they never wait for the row prefetch (vmcnt)
  __shared__ __attribute__((aligned(16))) float4 w2s[50];
  if (threadIdx.x < 50) w2s[threadIdx.x] = reinterpret_cast<const float4*>(p.w2)[threadIdx.x];
  __syncthreads();
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)p.n_windows * p.n_slabs * kD2Strips;
  if (gid >= total) return;
  const int strip = (int)(gid % kD2Strips);
  const int slab = (int)((gid / kD2Strips) % p.n_slabs);
  const int64_t b = gid / (kD2Strips * p.n_slabs);
  const int ta = slab * p.slab_rows;
  const int tb = ta + p.slab_rows < kFrames ? ta + p.slab_rows : kFrames;
  const float* src = p.c1 + b * kC1Win + (int64_t)(4 * strip) * 8;  // padded bin 4 s = stack bin 4 s - 2
  float* dst = p.out + b * kPlaneC + 4 * strip;

  v2f acc[5][4];  // acc[d] = output row r - 2 + d while input row r is being added
#pragma unroll
  for (int d = 0; d < 5; ++d)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[d][i] = v2f{0.0f, 0.0f};

  // rows outside the window are zero: the march covers input rows r_first .. r_last only
  const int r_first = ta - 2 > 0 ? ta - 2 : 0;
  const int r_last = tb + 1 < kFrames - 1 ? tb + 1 : kFrames - 1;
  float4 xa[16], xb[16];
  d2_load(src + (int64_t)r_first * kC1Row * 8, xa);
  for (int r = r_first; r <= r_last; r += 2) {
    // ---- row r from xa (row r + 1 prefetched into xb)
    if (r + 1 <= r_last) d2_load(src + (int64_t)(r + 1) * kC1Row * 8, xb);
    {
      bool live[5];
#pragma unroll
      for (int d = 0; d < 5; ++d) live[d] = (r - 2 + d) >= ta && (r - 2 + d) < tb;
      d2_fma(w2s, xa, acc, live);
      if (live[0]) d2_emit(dst + (int64_t)(r - 2) * kFreqC, acc[0], p.bias);
      d2_shift(acc);
    }
    if (r + 1 > r_last) break;
    // ---- row r + 1 from xb (row r + 2 prefetched into xa)
    if (r + 2 <= r_last) d2_load(src + (int64_t)(r + 2) * kC1Row * 8, xa);
    {
      bool live[5];
#pragma unroll
      for (int d = 0; d < 5; ++d) live[d] = (r - 1 + d) >= ta && (r - 1 + d) < tb;
      d2_fma(w2s, xb, acc, live);
      if (live[0]) d2_emit(dst + (int64_t)(r - 1) * kFreqC, acc[0], p.bias);
      d2_shift(acc);
    }
  }
  // output rows whose last input rows lie below the window (zero rows): flush what is still open
  for (int t = r_last - 1; t < tb; ++t) {
    if (t >= ta) d2_emit(dst + (int64_t)t * kFreqC, acc[0], p.bias);
    d2_shift(acc);
  }
}

void launch_contour_conv1(const uint32_t* zp, const void* wlds, const float* bias, float* c1, int n_windows,
                          int n_cu, bool weights_have_lo, hipStream_t stream) {
  // one workgroup per CU; split windows into row chunks when there are fewer windows than CUs
  int chunks = 1;
  while (chunks < 4 && n_windows * chunks < n_cu) chunks *= 2;
  Conv1Params p{zp, static_cast<const uint4*>(wlds), bias, c1, n_windows, chunks};
  const int items = n_windows * chunks;
  const int grid = items < n_cu ? items : n_cu;
  if (weights_have_lo)
    hipLaunchKernelGGL(contour_conv1_kernel<true>, dim3(grid), dim3(kD1Threads), 0, stream, p);
  else
    hipLaunchKernelGGL(contour_conv1_kernel<false>, dim3(grid), dim3(kD1Threads), 0, stream, p);
}

void launch_contour_conv2(const float* c1, const float* w2, float bias, float* contour, int n_windows, int n_cu,
                          hipStream_t stream) {
  // Every wave of a launch should be resident at once (2 waves per SIMD at ~220 VGPRs): a second, partially filled
  // round of waves doubles a launch's duration.  So: sub-batches of at most `per_launch` windows, each cut into as
  // many frame slabs as fill the chip (>= 7 slabs keeps the 4-row halo re-read under 16 %).
  const int64_t slots = (int64_t)n_cu * 4 * 2 * 64;  // resident threads
  int per_launch = (int)(slots / (7 * kD2Strips));
  per_launch = per_launch < 1 ? 1 : per_launch;
  for (int w0 = 0; w0 < n_windows; w0 += per_launch) {
    const int n = n_windows - w0 < per_launch ? n_windows - w0 : per_launch;
    int n_slabs = (int)(slots / ((int64_t)n * kD2Strips));
    n_slabs = n_slabs < 1 ? 1 : (n_slabs > 16 ? 16 : n_slabs);
    const int slab_rows = (kFrames + n_slabs - 1) / n_slabs;
    n_slabs = (kFrames + slab_rows - 1) / slab_rows;
    Conv2Params p{c1 + (int64_t)w0 * kC1Win, w2, bias, contour + (int64_t)w0 * kPlaneC, n, slab_rows, n_slabs};
    const int64_t total = (int64_t)n * n_slabs * kD2Strips;
    hipLaunchKernelGGL(contour_conv2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p);
  }
}

}  // namespace bp
