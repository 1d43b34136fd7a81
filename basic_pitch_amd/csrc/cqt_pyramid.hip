// CQT pyramid: 8 x decimate-by-2 with the 256-tap half-band FIR, plus the track windowing /
// un-overlapping copies.
//
// Reference behaviour (spotify/basic-pitch v0.4.0):
//   basic_pitch/layers/nnaudio.py:259-284 downsampling_by_n(match_torch_exactly=True):
//       zero-pad 127 samples each side, conv1d with the firwin2 kernel, stride 2, VALID
//   basic_pitch/layers/nnaudio.py:636-638     applied 8 times, level k+1 from level k
//   basic_pitch/inference.py:194-244          window_audio_file / get_audio_input
//   basic_pitch/inference.py:247-279          unwrap_output
//
// Roofline: 11.18 M MAC / window on the f32 VALU (2 % of the path's FLOPs); the signal tile is
// staged once in LDS and every thread produces two adjacent outputs from 65 ds_read_b128, so the
// kernel is VALU-issue bound, not LDS- or HBM-bound.  Algorithmic bytes per window: 175,376 B read
// (level 0) + 174,764 B of pyramid written.
#include "bp_common.h"

namespace bp {

constexpr int kDecThreads = 256;
constexpr int kDecOutPerBlock = 2 * kDecThreads;        // 512 outputs
constexpr int kDecTileIn = 2 * kDecOutPerBlock + 256;   // 1280 staged inputs (1278 used)

// y[n] = sum_{j=0}^{255} h[j] * xz[2n + j - 127],  xz = x zero-extended     (nnaudio.py:269-279)
__global__ __launch_bounds__(kDecThreads) void decimate2_kernel(const float* __restrict__ src,
                                                                int64_t src_stride, int len_in,
                                                                float* __restrict__ dst,
                                                                int64_t dst_stride, int len_out,
                                                                const float* __restrict__ h) {
  __shared__ __attribute__((aligned(16))) float s[kDecTileIn];
  const int b = blockIdx.y;
  const int o0 = blockIdx.x * kDecOutPerBlock;
  const float* x = src + (int64_t)b * src_stride;
  const int in0 = 2 * o0 - 127;
  for (int i = threadIdx.x; i < kDecTileIn; i += kDecThreads) {
    const int g = in0 + i;
    s[i] = (g >= 0 && g < len_in) ? x[g] : 0.0f;
  }
  __syncthreads();

  // outputs n = o0 + 2*tid (A) and n+1 (B).  A reads s[4*tid + j], B reads s[4*tid + 2 + j].
  // Tap quad k (taps 4k..4k+3) multiplies s[4tid+4k .. +3] for A and s[4tid+4k+2 .. +5] for B, so
  // both outputs consume the same four (wave-uniform, scalar-loaded) coefficients per step.
  const float4* s4 = reinterpret_cast<const float4*>(s + 4 * threadIdx.x);
  float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;  // two partial sums per output (even / odd quads)
  float4 cur = s4[0];
#pragma unroll 4
  for (int k = 0; k < 64; k += 2) {
    const float4 mid = s4[k + 1];
    const float4 nxt = s4[k + 2];
    const float h0 = h[4 * k + 0], h1 = h[4 * k + 1], h2 = h[4 * k + 2], h3 = h[4 * k + 3];
    const float h4 = h[4 * k + 4], h5 = h[4 * k + 5], h6 = h[4 * k + 6], h7 = h[4 * k + 7];
    a0 = fmaf(h0, cur.x, a0);
    a0 = fmaf(h1, cur.y, a0);
    a0 = fmaf(h2, cur.z, a0);
    a0 = fmaf(h3, cur.w, a0);
    b0 = fmaf(h0, cur.z, b0);
    b0 = fmaf(h1, cur.w, b0);
    b0 = fmaf(h2, mid.x, b0);
    b0 = fmaf(h3, mid.y, b0);
    a1 = fmaf(h4, mid.x, a1);
    a1 = fmaf(h5, mid.y, a1);
    a1 = fmaf(h6, mid.z, a1);
    a1 = fmaf(h7, mid.w, a1);
    b1 = fmaf(h4, mid.z, b1);
    b1 = fmaf(h5, mid.w, b1);
    b1 = fmaf(h6, nxt.x, b1);
    b1 = fmaf(h7, nxt.y, b1);
    cur = nxt;
  }
  const int n = o0 + 2 * threadIdx.x;
  float* y = dst + (int64_t)b * dst_stride;
  if (n + 1 < len_out) {
    *reinterpret_cast<float2*>(y + n) = make_float2(a0 + a1, b0 + b1);
  } else if (n < len_out) {
    y[n] = a0 + a1;
  }
}

void launch_pyramid(const float* audio, float* pyr, const float* lowpass, int n_windows,
                    hipStream_t stream) {
  for (int k = 1; k < kOctaves; ++k) {
    const float* src = (k == 1) ? audio : pyr + pyr_off(k - 1);
    const int64_t sstride = (k == 1) ? kAudioN : kPyrStride;
    const int lin = level_len(k - 1), lout = level_len(k);
    dim3 grid((lout + kDecOutPerBlock - 1) / kDecOutPerBlock, n_windows);
    hipLaunchKernelGGL(decimate2_kernel, grid, dim3(kDecThreads), 0, stream, src, sstride, lin,
                       pyr + pyr_off(k), (int64_t)kPyrStride, lout, lowpass);
  }
}

// ---- track windowing (inference.py:242 zero lead-in of 3840, 207-213 hop 36164 + tail pad) ----
// win_len / hop / lead: 43844 / 36164 / 3840 samples at 22.05 kHz, doubled for the extended 44.1 kHz geometry
__global__ __launch_bounds__(256) void window_track_kernel(const float* __restrict__ samples,
                                                           int64_t n_samples, int64_t first_window,
                                                           float* __restrict__ audio, int win_len, int hop,
                                                           int lead) {
  const int64_t w = first_window + blockIdx.y;
  const int64_t start = w * hop - lead;  // index into the un-padded track
  float* dst = audio + (int64_t)blockIdx.y * win_len;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < win_len; i += gridDim.x * 256) {
    const int64_t g = start + i;
    dst[i] = (g >= 0 && g < n_samples) ? samples[g] : 0.0f;
  }
}

void launch_window_track(const float* samples, int64_t n_samples, int64_t first_window,
                         int n_windows, float* audio, int win_len, int hop, int lead, hipStream_t stream) {
  hipLaunchKernelGGL(window_track_kernel, dim3(43, n_windows), dim3(256), 0, stream, samples,
                     n_samples, first_window, audio, win_len, hop, lead);
}

// all pieces of a chunk in one launch: blockIdx.y = window slot of the chunk
__global__ __launch_bounds__(256) void window_tracks_kernel(TrackSegs ts, float* __restrict__ audio, int win_len, int hop,
                                                            int lead) {
  const int slot = blockIdx.y;
  int k = 0;
  while (k + 1 < ts.n && slot >= ts.seg[k + 1].at) ++k;  // block-uniform
  const TrackSeg& g = ts.seg[k];
  const int64_t start = (g.first_window + (slot - g.at)) * hop - lead;
  float* dst = audio + (int64_t)slot * win_len;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < win_len; i += gridDim.x * 256) {
    const int64_t x = start + i;
    dst[i] = (x >= 0 && x < g.n_samples) ? g.samples[x] : 0.0f;
  }
}

void launch_window_tracks(const TrackSegs& ts, int n_slots, float* audio, int win_len, int hop, int lead,
                          hipStream_t stream) {
  hipLaunchKernelGGL(window_tracks_kernel, dim3(43, n_slots), dim3(256), 0, stream, ts, audio, win_len, hop, lead);
}

// ---- unwrap_output (inference.py:267-279): keep frames 15..156 of every window, trim to T rows ----
__global__ __launch_bounds__(256) void unwrap_kernel(const float* __restrict__ win_out, int n_freq,
                                                     int64_t first_window, int64_t total_rows,
                                                     float* __restrict__ out) {
  const int lw = blockIdx.y;            // local window in this chunk
  const int64_t row0 = (first_window + lw) * 142;
  const float* src = win_out + ((int64_t)lw * kFrames + 15) * n_freq;
  const int n = 142 * n_freq;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int64_t row = row0 + i / n_freq;
    if (row < total_rows) out[row0 * n_freq + i] = src[i];
  }
}

// the three maps of a chunk in ONE launch (blockIdx.z = map): three launches of this small kernel cost a track ~30 us
__global__ __launch_bounds__(256) void unwrap3_kernel(const float* __restrict__ note, const float* __restrict__ onset,
                                                      const float* __restrict__ contour, int64_t first_window,
                                                      int64_t total_rows, float* __restrict__ o_note,
                                                      float* __restrict__ o_onset, float* __restrict__ o_contour) {
  const int m = blockIdx.z;
  const int n_freq = m == 2 ? kFreqC : kFreqN;
  const float* win_out = m == 0 ? note : (m == 1 ? onset : contour);
  float* out = m == 0 ? o_note : (m == 1 ? o_onset : o_contour);
  const int lw = blockIdx.y;
  const int64_t row0 = (first_window + lw) * 142;
  const float* src = win_out + ((int64_t)lw * kFrames + 15) * n_freq;
  const int n = 142 * n_freq;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int64_t row = row0 + i / n_freq;
    if (row < total_rows) out[row0 * n_freq + i] = src[i];
  }
}
void launch_unwrap3(const float* note, const float* onset, const float* contour, int64_t first_window, int n_windows,
                    int64_t total_rows, float* o_note, float* o_onset, float* o_contour, hipStream_t stream) {
  hipLaunchKernelGGL(unwrap3_kernel, dim3(16, n_windows, 3), dim3(256), 0, stream, note, onset, contour, first_window,
                     total_rows, o_note, o_onset, o_contour);
}

void launch_unwrap(const float* win_out, int n_freq, int64_t first_window, int n_windows,
                   int64_t total_rows, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(unwrap_kernel, dim3(16, n_windows), dim3(256), 0, stream, win_out, n_freq,
                     first_window, total_rows, out);
}

// all pieces of a chunk and all three maps in one launch: blockIdx.y = window slot, blockIdx.z = map
__global__ __launch_bounds__(256) void unwrap_tracks_kernel(TrackSegs ts, const float* __restrict__ note,
                                                            const float* __restrict__ onset,
                                                            const float* __restrict__ contour) {
  const int slot = blockIdx.y, map = blockIdx.z;
  int k = 0;
  while (k + 1 < ts.n && slot >= ts.seg[k + 1].at) ++k;
  const TrackSeg& g = ts.seg[k];
  if (g.total_rows <= 0) return;
  const int n_freq = map == 2 ? 264 : 88;
  const float* win_out = map == 0 ? note : (map == 1 ? onset : contour);
  float* out = g.out[map];
  const int64_t row0 = (g.first_window + (slot - g.at)) * 142;
  const float* src = win_out + ((int64_t)slot * kFrames + 15) * n_freq;
  const int n = 142 * n_freq;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int64_t row = row0 + i / n_freq;
    if (row < g.total_rows) out[row0 * n_freq + i] = src[i];
  }
}

void launch_unwrap_tracks(const TrackSegs& ts, int n_slots, const float* note, const float* onset, const float* contour,
                          hipStream_t stream) {
  hipLaunchKernelGGL(unwrap_tracks_kernel, dim3(16, n_slots, 3), dim3(256), 0, stream, ts, note, onset, contour);
}

}  // namespace bp
