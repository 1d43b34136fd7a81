// Audio ingest on the device: channel-mean downmix + polyphase FIR resampling to 22.05 kHz.
//
// Replaces the `librosa.load(path, sr=22050, mono=True)` step of basic_pitch/inference.py:239 for decoded PCM
// (SURVEY.md §8f rank 2): the caller hands over interleaved float PCM at the file's rate, the 22.05 kHz mono signal
// never exists on the host, and the windowing that follows (inference.py:194-244) reads it in place.
//   * downmix = mean over channels (librosa.to_mono);
//   * resampling = the rational polyphase FIR of scipy.signal.resample_poly(x, up, down) with its default design
//     (firwin, 2 * 10 * max(up, down) + 1 taps, cutoff 1 / max(up, down), Kaiser beta = 5, DC gain up): the taps are
//     generated here in float64 with the same formulas, products accumulate in float64, so the result equals scipy's
//     float64 output rounded to float32 up to summation order.  This is the same resampler basic_pitch_amd/audio.py
//     uses on the host; librosa's soxr_hq is not reproducible without libsoxr.
// Roofline: HBM — 4 B x channels read + 4 B x 22050 / rate written per input frame, ~44 MACs per output sample.
#include <cmath>
#include <vector>

#include "bp_common.h"

namespace bp {

static double bessel_i0(double x) {
  double sum = 1.0, term = 1.0;
  const double q = x * x / 4.0;
  for (int k = 1; k < 200; ++k) {
    term *= q / ((double)k * (double)k);
    sum += term;
    if (term < 1e-18 * sum) break;
  }
  return sum;
}

static int gcd_int(int a, int b) {
  while (b) {
    const int t = a % b;
    a = b;
    b = t;
  }
  return a;
}

// scipy.signal.resample_poly's default filter and alignment for target / source rates
ResamplePlan make_resample_plan(int source_rate, int target_rate, std::vector<double>& taps) {
  ResamplePlan pl{1, 1, 0, 0, 0};
  const int g = gcd_int(source_rate, target_rate);
  pl.up = target_rate / g;
  pl.down = source_rate / g;
  const int max_rate = pl.up > pl.down ? pl.up : pl.down;
  const int half_len = 10 * max_rate;
  pl.n_taps = 2 * half_len + 1;
  taps.assign(pl.n_taps, 0.0);
  const double fc = 1.0 / (double)max_rate;  // cutoff as a fraction of Nyquist
  const double beta = 5.0, i0b = bessel_i0(beta);
  const double pi = 3.14159265358979323846;
  double sum = 0.0;
  for (int n = 0; n < pl.n_taps; ++n) {
    const double m = (double)(n - half_len);
    const double x = fc * m;
    const double sinc = (m == 0.0) ? 1.0 : std::sin(pi * x) / (pi * x);
    const double r = 2.0 * (double)n / (double)(pl.n_taps - 1) - 1.0;
    const double w = bessel_i0(beta * std::sqrt(1.0 - r * r > 0.0 ? 1.0 - r * r : 0.0)) / i0b;
    taps[n] = fc * sinc * w;
    sum += taps[n];
  }
  for (double& t : taps) t = t / sum * (double)pl.up;
  pl.n_pre_pad = pl.down - half_len % pl.down;
  pl.n_pre_remove = (half_len + pl.n_pre_pad) / pl.down;
  return pl;
}

__global__ __launch_bounds__(256) void downmix_kernel(const float* __restrict__ pcm, int64_t n_frames, int channels,
                                                      float* __restrict__ mono) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_frames) return;
  const float* p = pcm + i * channels;
  float s = 0.0f;
  for (int c = 0; c < channels; ++c) s += p[c];  // numpy mean(axis=1, dtype=float32): pairwise for > 8, plain here
  mono[i] = s / (float)channels;
}

__global__ __launch_bounds__(256) void resample_poly_kernel(const float* __restrict__ x, int64_t n_in,
                                                            const double* __restrict__ taps, ResamplePlan pl,
                                                            float* __restrict__ y, int64_t n_out) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n_out) return;
  // y[k] = sum_j x[j] * h[(k + n_pre_remove) * down - n_pre_pad - j * up]
  const int64_t base = (k + pl.n_pre_remove) * (int64_t)pl.down - pl.n_pre_pad;
  int64_t j_hi = base >= 0 ? base / pl.up : -1;
  if (j_hi > n_in - 1) j_hi = n_in - 1;
  int64_t lo_num = base - (pl.n_taps - 1);
  int64_t j_lo = lo_num <= 0 ? 0 : (lo_num + pl.up - 1) / pl.up;
  double acc = 0.0;
  for (int64_t j = j_lo; j <= j_hi; ++j) acc += (double)x[j] * taps[base - j * pl.up];
  y[k] = (float)acc;
}

void launch_downmix(const float* pcm, int64_t n_frames, int channels, float* mono, hipStream_t stream) {
  if (n_frames <= 0) return;
  hipLaunchKernelGGL(downmix_kernel, dim3((unsigned)((n_frames + 255) / 256)), dim3(256), 0, stream, pcm, n_frames,
                     channels, mono);
}

void launch_resample(const float* x, int64_t n_in, const double* taps, const ResamplePlan& pl, float* y,
                     int64_t n_out, hipStream_t stream) {
  if (n_out <= 0) return;
  hipLaunchKernelGGL(resample_poly_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, stream, x, n_in, taps,
                     pl, y, n_out);
}

}  // namespace bp
