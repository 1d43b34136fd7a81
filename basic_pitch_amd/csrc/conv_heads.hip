// The three single-output-channel "head" convolutions with their sigmoids:
//   contour:  Conv2D 8->1,  5x5, "same", sigmoid      (basic_pitch/models.py:254-263)
//   note:     Conv2D 32->1, 7x3, "same", sigmoid      (basic_pitch/models.py:282-290)
//   onset:    Concatenate([note_sigmoid, onset_features]) -> Conv2D 33->1, 3x3, "same", sigmoid
//                                                      (basic_pitch/models.py:305-318)
// FlattenFreqCh (nn.py:105-119) is a no-op for one channel: outputs are [frame][bin] directly.
//
// N = 1 contractions do not map onto MFMA tiles without idling >75 % of the array, and together
// they are 4.5 % of the path's FLOPs, so these run on the f32 VALU: every thread produces 4 adjacent
// bins from 16-byte (float4) loads of the planar input rows (L1/L2 resident: produced by the
// previous kernel), weights are wave-uniform and travel through the scalar cache.
// Roofline: VALU/L1 bound; algorithmic work 18.2 / 20.3 / 9.0 MFLOP per window.
#include "bp_common.h"

namespace bp {

// Input channel c comes from src0 for c < C0 and from src1 otherwise (the onset head concatenates
// the 1-channel note map with the 32 onset feature planes).
template <int CIN, int C0, int KH, int KW, int W>
__global__ __launch_bounds__(256) void head_conv_kernel(const float* __restrict__ src0,
                                                        int64_t src0_bstride,
                                                        const float* __restrict__ src1,
                                                        int64_t src1_bstride,
                                                        const float* __restrict__ wgt, float bias,
                                                        float* __restrict__ out) {
  constexpr int G = W / 4;            // 4-bin groups per frame
  constexpr int PH = KH / 2, PW = KW / 2;
  constexpr int PLANE = kFrames * W;
  static_assert(W % 4 == 0 && PW <= 4, "geometry");
  const int b = blockIdx.y;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= kFrames * G) return;
  const int t = idx / G;
  const int w0 = (idx - t * G) * 4;

  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int c = 0; c < CIN; ++c) {
    const float* plane = (c < C0) ? src0 + (int64_t)b * src0_bstride + (int64_t)c * PLANE
                                  : src1 + (int64_t)b * src1_bstride + (int64_t)(c - C0) * PLANE;
    const float* wc = wgt + c * KH * KW;
#pragma unroll
    for (int dt = 0; dt < KH; ++dt) {
      const int tt = t + dt - PH;
      if (tt < 0 || tt >= kFrames) continue;
      const float* row = plane + tt * W;
      // window of 12 inputs: bins w0-4 .. w0+7 (zero outside the image)
      float in[12];
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 lo = (w0 >= 4) ? *reinterpret_cast<const float4*>(row + w0 - 4) : z4;
      const float4 mid = *reinterpret_cast<const float4*>(row + w0);
      const float4 hi = (w0 + 4 < W) ? *reinterpret_cast<const float4*>(row + w0 + 4) : z4;
      in[0] = lo.x; in[1] = lo.y; in[2] = lo.z; in[3] = lo.w;
      in[4] = mid.x; in[5] = mid.y; in[6] = mid.z; in[7] = mid.w;
      in[8] = hi.x; in[9] = hi.y; in[10] = hi.z; in[11] = hi.w;
#pragma unroll
      for (int dw = 0; dw < KW; ++dw) {
        const float wv = wc[dt * KW + dw];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(wv, in[4 + j + dw - PW], acc[j]);
      }
    }
  }
  float4 o;
  o.x = sigmoidf_exact(acc[0] + bias);
  o.y = sigmoidf_exact(acc[1] + bias);
  o.z = sigmoidf_exact(acc[2] + bias);
  o.w = sigmoidf_exact(acc[3] + bias);
  *reinterpret_cast<float4*>(out + (int64_t)b * PLANE + t * W + w0) = o;
}

void launch_contour2(const float* c1, const float* wgt, float bias, float* contour, int n_windows,
                     hipStream_t stream) {
  constexpr int G = kFreqC / 4;
  dim3 grid((kFrames * G + 255) / 256, n_windows);
  hipLaunchKernelGGL((head_conv_kernel<8, 8, 5, 5, kFreqC>), grid, dim3(256), 0, stream, c1,
                     (int64_t)8 * kPlaneC, (const float*)nullptr, (int64_t)0, wgt, bias, contour);
}

void launch_note2(const float* n1, const float* wgt, float bias, float* note, int n_windows,
                  hipStream_t stream) {
  constexpr int G = kFreqN / 4;
  dim3 grid((kFrames * G + 255) / 256, n_windows);
  hipLaunchKernelGGL((head_conv_kernel<32, 32, 7, 3, kFreqN>), grid, dim3(256), 0, stream, n1,
                     (int64_t)32 * kPlaneN, (const float*)nullptr, (int64_t)0, wgt, bias, note);
}

void launch_onset2(const float* note, const float* o1, const float* wgt, float bias, float* onset,
                   int n_windows, hipStream_t stream) {
  constexpr int G = kFreqN / 4;
  dim3 grid((kFrames * G + 255) / 256, n_windows);
  hipLaunchKernelGGL((head_conv_kernel<33, 1, 3, 3, kFreqN>), grid, dim3(256), 0, stream, note,
                     (int64_t)kPlaneN, o1, (int64_t)32 * kPlaneN, wgt, bias, onset);
}

}  // namespace bp
