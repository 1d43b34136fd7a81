// The three single-output-channel "head" convolutions with their sigmoids:
//   contour:  Conv2D 8->1,  5x5, "same", sigmoid      (basic_pitch/models.py:254-263)
//   note:     Conv2D 32->1, 7x3, "same", sigmoid      (basic_pitch/models.py:282-290)
//   onset:    Concatenate([note_sigmoid, onset_features]) -> Conv2D 33->1, 3x3, "same", sigmoid
//                                                      (basic_pitch/models.py:305-318)
// FlattenFreqCh (nn.py:105-119) is a no-op for one channel: outputs are [frame][bin] directly.
//
// N = 1 contractions: a Toeplitz expansion onto MFMA tiles would idle 75-83 % of the array, so these
// run on the f32 VALU, organised so the VALU (not LDS or L1) is the limit:
//   * a workgroup owns 16 output frames x the full bin width; input channels stream through LDS in
//     chunks (planar rows, zero halo materialised, row origin shifted by the left pad so every
//     thread's 8-bin window is two aligned ds_read_b128);
//   * every thread produces a 2-frame x 4-bin block, so one pair of 16-byte reads feeds
//     2*KW*4 FMAs (>= 10 FMA per LDS read: above the 8:1 VALU:LDS issue ratio of a CU);
//   * weights are wave-uniform -> scalar loads, used as SGPR operands of v_fma_f32.
// Roofline: f32 VALU; algorithmic work 18.2 / 20.3 / 9.0 MFLOP per window.
#include "bp_common.h"

namespace bp {

constexpr int kHeadThreads = 192;
constexpr int kHeadSlab = 16;
constexpr int kHeadSlabs = (kFrames + kHeadSlab - 1) / kHeadSlab;  // 11

template <int CIN, int C0, int KH, int KW, int W, int CHUNK>
struct HeadCfg {
  static constexpr int PH = KH / 2, PW = KW / 2;
  static constexpr int G = W / 4;                       // 4-bin groups per frame
  static constexpr int ROWS = kHeadSlab + KH - 1;       // staged frames
  static constexpr int WP = ((W + 2 * PW + 3) / 4) * 4 + 4;  // padded row (floats), multiple of 4
  static constexpr int ITEMS = (kHeadSlab / 2) * G;     // 2x4 output blocks per slab
  static constexpr int IPT = (ITEMS + kHeadThreads - 1) / kHeadThreads;  // blocks per thread
  static constexpr int NCHUNK = (CIN + CHUNK - 1) / CHUNK;
  static constexpr int LDS_FLOATS = CHUNK * ROWS * WP;
  static_assert(W % 4 == 0 && PW <= 2 && kHeadSlab % 2 == 0, "geometry");
};

template <int CIN, int C0, int KH, int KW, int W, int CHUNK>
__global__ __launch_bounds__(kHeadThreads) void head_conv_kernel(const float* __restrict__ src0,
                                                                 int64_t src0_bstride,
                                                                 const float* __restrict__ src1,
                                                                 int64_t src1_bstride,
                                                                 const float* __restrict__ wgt,
                                                                 float bias, float* __restrict__ out) {
  using Cfg = HeadCfg<CIN, C0, KH, KW, W, CHUNK>;
  constexpr int PH = Cfg::PH, PW = Cfg::PW, G = Cfg::G, ROWS = Cfg::ROWS, WP = Cfg::WP;
  constexpr int PLANE = kFrames * W;
  __shared__ __attribute__((aligned(16))) float tile[Cfg::LDS_FLOATS];

  const int b = blockIdx.y;
  const int t0 = blockIdx.x * kHeadSlab;

  // this thread's output blocks: item -> (frame pair rp, bin group g)
  int rp[Cfg::IPT], gg[Cfg::IPT];
  bool live[Cfg::IPT];
#pragma unroll
  for (int i = 0; i < Cfg::IPT; ++i) {
    const int item = threadIdx.x + i * kHeadThreads;
    live[i] = item < Cfg::ITEMS;
    const int it = live[i] ? item : 0;
    rp[i] = it / G;
    gg[i] = it - rp[i] * G;
  }
  float acc[Cfg::IPT][2][4];
#pragma unroll
  for (int i = 0; i < Cfg::IPT; ++i)
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][o][j] = 0.0f;

  // zero once: halo columns and out-of-image rows stay zero for every chunk (same geometry)
  for (int i = threadIdx.x; i < Cfg::LDS_FLOATS; i += kHeadThreads) tile[i] = 0.0f;

#pragma unroll 1
  for (int ch0 = 0; ch0 < CIN; ch0 += CHUNK) {
    __syncthreads();  // previous chunk fully consumed (and the zero fill done)
    // ---- stage CHUNK channels x ROWS frames x W bins (float4, coalesced)
    for (int i = threadIdx.x; i < CHUNK * ROWS * (W / 4); i += kHeadThreads) {
      const int cc = i / (ROWS * (W / 4));
      const int rem = i - cc * (ROWS * (W / 4));
      const int r = rem / (W / 4);
      const int q = rem - r * (W / 4);
      const int c = ch0 + cc;
      const int t = t0 - PH + r;
      if (c < CIN && t >= 0 && t < kFrames) {
        const float* plane = (c < C0) ? src0 + (int64_t)b * src0_bstride + (int64_t)c * PLANE
                                      : src1 + (int64_t)b * src1_bstride + (int64_t)(c - C0) * PLANE;
        const float4 v = *reinterpret_cast<const float4*>(plane + t * W + 4 * q);
        float* d = tile + (cc * ROWS + r) * WP + 4 * q + PW;  // column j holds bin j - PW
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      } else if (c >= CIN) {
        float* d = tile + (cc * ROWS + r) * WP + 4 * q + PW;  // ragged last chunk: clear stale data
        d[0] = 0.f; d[1] = 0.f; d[2] = 0.f; d[3] = 0.f;
      }
    }
    __syncthreads();

    // ---- accumulate
#pragma unroll 1
    for (int cc = 0; cc < CHUNK; ++cc) {
      const int c = ch0 + cc;
      if (c >= CIN) break;
      const float* wc = wgt + c * KH * KW;
#pragma unroll
      for (int i = 0; i < Cfg::IPT; ++i) {
        const float* base = tile + (cc * ROWS + 2 * rp[i]) * WP + 4 * gg[i];
#pragma unroll
        for (int r = 0; r <= KH; ++r) {  // input frame 2*rp + r feeds output frame o with dt = r - o
          const float4 lo = *reinterpret_cast<const float4*>(base + r * WP);
          const float4 hi = *reinterpret_cast<const float4*>(base + r * WP + 4);
          const float in[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
          for (int o = 0; o < 2; ++o) {
            const int dt = r - o;
            if (dt < 0 || dt >= KH) continue;
#pragma unroll
            for (int dw = 0; dw < KW; ++dw) {
              const float wv = wc[dt * KW + dw];
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[i][o][j] = fmaf(wv, in[j + dw], acc[i][o][j]);
            }
          }
        }
      }
    }
  }

#pragma unroll
  for (int i = 0; i < Cfg::IPT; ++i) {
    if (!live[i]) continue;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int t = t0 + 2 * rp[i] + o;
      if (t >= kFrames) continue;
      float4 v;
      v.x = sigmoidf_exact(acc[i][o][0] + bias);
      v.y = sigmoidf_exact(acc[i][o][1] + bias);
      v.z = sigmoidf_exact(acc[i][o][2] + bias);
      v.w = sigmoidf_exact(acc[i][o][3] + bias);
      *reinterpret_cast<float4*>(out + (int64_t)b * PLANE + t * W + 4 * gg[i]) = v;
    }
  }
}

void launch_contour2(const float* c1, const float* wgt, float bias, float* contour, int n_windows,
                     hipStream_t stream) {
  hipLaunchKernelGGL((head_conv_kernel<8, 8, 5, 5, kFreqC, 2>), dim3(kHeadSlabs, n_windows),
                     dim3(kHeadThreads), 0, stream, c1, (int64_t)8 * kPlaneC, (const float*)nullptr,
                     (int64_t)0, wgt, bias, contour);
}

void launch_note2(const float* n1, const float* wgt, float bias, float* note, int n_windows,
                  hipStream_t stream) {
  hipLaunchKernelGGL((head_conv_kernel<32, 32, 7, 3, kFreqN, 4>), dim3(kHeadSlabs, n_windows),
                     dim3(kHeadThreads), 0, stream, n1, (int64_t)32 * kPlaneN, (const float*)nullptr,
                     (int64_t)0, wgt, bias, note);
}

void launch_onset2(const float* note, const float* o1, const float* wgt, float bias, float* onset,
                   int n_windows, hipStream_t stream) {
  hipLaunchKernelGGL((head_conv_kernel<33, 1, 3, 3, kFreqN, 11>), dim3(kHeadSlabs, n_windows),
                     dim3(kHeadThreads), 0, stream, note, (int64_t)kPlaneN, o1, (int64_t)32 * kPlaneN,
                     wgt, bias, onset);
}

}  // namespace bp
