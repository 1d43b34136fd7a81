// Fused contour branch — split-precision matrix-core kernel.  NOT the default any more: bp_create selects it
// with BP_CONTOUR_PATH=fused for A/B runs against the two-kernel form (conv_contour_direct.hip), which reaches
// the same end-to-end rate with a matrix pipe that is busy instead of waiting.
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
//   * the chunk walk is specialised on the wave index (one dispatch per kernel): the K-slice offsets of the image
//     reads are immediates, the reduce-scatter has no scalar selects; image fragments are prefetched kCbPf k-steps
//     ahead; the two cross terms lo*hi and hi*lo share one accumulator chain.
//
// Numerics: operands are x = hi + lo, lo stored * 2^11 (f16 exponent range, see cqt_mfma.hip); products
// hi*hi + (lo*hi + hi*lo) * 2^-11 accumulate in fp32.  Deterministic: fixed reduction orders everywhere.
//
// Roofline: f16 MFMA issue.  Algorithmic work 680.0 + 18.2 MFLOP per window;
// bytes per window: 214,656 (zp) read, 181,632 written.
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
    g = g < 0 ? 0 : (g > kBins + 2 ? kBins + 2 : g);
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
// ---------------------------------------------------------------------------------------------------------
// v2 of the same kernel: identical mapping and LDS layout, different instruction schedule.
//   * the whole chunk walk is specialised on the wave index G (one dispatch per kernel): the K-slice offsets
//     of the image reads are immediates, the reduce-scatter has no scalar selects or branches;
//   * image fragments are prefetched kCbPf k-steps ahead of the MFMAs that consume them (the register budget
//     is freed by accumulating the two cross terms lo*hi and hi*lo, which share the 2^-11 scale, in ONE chain);
//   * the reduce-scatter moves float4 (ds_write_b128 / ds_read_b128).
constexpr int kCbPf = 2;
constexpr int kCbLoOff = kCbRing * kCbSlots;  // img[] = hi image, then lo image

template <int G>
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
    bh[s] = __builtin_bit_cast(f16x8, img[slot]);
    bl[s] = __builtin_bit_cast(f16x8, img[slot + kCbLoOff]);
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

template <int G>
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

  const int n_items = p.n_windows * kCbChunks;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int b = item / kCbChunks;
    const int T0 = (item - b * kCbChunks) * kCbChunkFrames;
    const int T1 = T0 + kCbChunkFrames;
    const int R0 = T0 - 2;
    const uint32_t* zpb = p.zp + (int64_t)b * kZWin + kZRow + kZPadL;
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
        cb2_mfma<G>(img, rowslot, lo_off, hi_off, wh, wl, a_hh, a_x);
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
      }
      __syncthreads();  // B1

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
      if (staging) cb_issue(zpb + (int64_t)(stage_row_ok ? stage_row : 0) * kZRow, stage_f, pf);

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
      __syncthreads();  // B2

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
      if (staging) {
        cb_mask(stage_f, stage_row_ok, pf);
        cb_put(pf, img_hi, img_lo, ((stage_row + 5 * 8) % kCbRing) * kCbSlots + stage_slot);
      }
    }
  }
}

__global__ __launch_bounds__(kCbThreads, 2) void contour_branch_kernel(ContourParams p) {
  __shared__ __attribute__((aligned(16))) uint4 img[2 * kCbRing * kCbSlots];
  __shared__ __attribute__((aligned(16))) float4 xbuf4[4 * 3 * 64];
  __shared__ float scr[25 * kCbScrT];
  __shared__ float oring[kCbORing * kFreqC];
  __shared__ float tailb[2 * 100];
  switch (wave_id()) {
    case 0: cb2_run<0>(p, img, xbuf4, scr, oring, tailb); break;
    case 1: cb2_run<1>(p, img, xbuf4, scr, oring, tailb); break;
    case 2: cb2_run<2>(p, img, xbuf4, scr, oring, tailb); break;
    default: cb2_run<3>(p, img, xbuf4, scr, oring, tailb); break;
  }
}


void launch_contour_branch(const uint32_t* zp, const void* wfrag, const float* wf32, float* contour,
                           int n_windows, int n_cu, hipStream_t stream) {
  ContourParams p{zp, static_cast<const uint4*>(wfrag), wf32, contour, n_windows};
  const int items = n_windows * kCbChunks;
  const int grid = items < 2 * n_cu ? items : 2 * n_cu;
  hipLaunchKernelGGL(contour_branch_kernel, dim3(grid), dim3(kCbThreads), 0, stream, p);
}

}  // namespace bp
