// Audio ingest on the device: channel-mean downmix + zero-phase polyphase FIR resampling to 22.05 kHz.
//
// Replaces the `librosa.load(path, sr=22050, mono=True)` step of basic_pitch/inference.py:239 for decoded PCM
// (SURVEY.md §8f rank 2): the caller hands over interleaved float PCM at the file's rate, the 22.05 kHz mono signal
// never exists on the host, and the windowing that follows (inference.py:194-244) reads it in place.
//   * downmix = mean over channels (librosa.to_mono);
//   * resampling = the response of librosa's default `res_type="soxr_hq"` (libsoxr 0.1.3 at SOXR_HQ; a third-party
//     dependency of the reference, restated from its published design: soxr.c soxr_quality_spec, filter.c
//     lsx_design_lpf / lsx_kaiser_beta / lsx_make_lpf): linear phase, pass-band to 0.9136 of the lower Nyquist,
//     stop-band from that Nyquist at 126.4 dB, Kaiser-windowed sinc with libsoxr's beta fit (13.04) and length
//     formula — 389 taps at the input rate for 2 : 1, which is libsoxr's whole pipeline for that ratio (one dft_stage).
//     Output sample k sits at input time k * down / up, the signal is zero outside the file, length
//     ceil(n * 22050 / rate) (librosa.resample).  Taps in float64, float64 accumulation, one rounding to fp32.
//     With it the reference's golden posteriorgrams of its 44.1 kHz clip are met at its own atol 1e-4 end to end
//     (tests/test_gpu_parity.py); other ratios are ONE polyphase stage of the same specification where libsoxr
//     cascades several.  Ratios whose tap table would pass 2^22 entries (e.g. 44101 Hz) evaluate the taps on the
//     fly from a tabulated window instead of failing.
// Roofline: HBM — 4 B x channels read + 4 B x 22050 / rate written per input frame; 389 / 2 fp64 MACs per input sample
// at 2 : 1 (0.16 ms for a 3-minute 44.1 kHz track at the chip's 78 TFLOP/s fp64 vector rate: negligible).
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

// libsoxr lsx_kaiser_beta (att >= 60 dB): cubic fits of beta over the attenuation, one row per octave of tr_bw / .0005
static double soxr_kaiser_beta(double att, double tr_bw) {
  static const double fit[10][4] = {
      {-6.784957e-10, 1.02856e-05, 0.1087556, -0.8988365 + .001}, {-6.897885e-10, 1.027433e-05, 0.10876, -0.8994658 + .002},
      {-1.000683e-09, 1.030092e-05, 0.1087677, -0.9007898 + .003}, {-3.654474e-10, 1.040631e-05, 0.1087085, -0.8977766 + .006},
      {8.106988e-09, 6.983091e-06, 0.1091387, -0.9172048 + .015},  {9.519571e-09, 7.272678e-06, 0.1090068, -0.9140768 + .025},
      {-5.626821e-09, 1.342186e-05, 0.1083999, -0.9065452 + .05},  {-9.965946e-08, 5.073548e-05, 0.1040967, -0.7672778 + .085},
      {1.604808e-07, -5.856462e-05, 0.1185998, -1.34824 + .1},     {-1.511964e-07, 6.363034e-05, 0.1064627, -0.9876665 + .18}};
  const double realm = std::log(tr_bw / .0005) / std::log(2.0);
  int r0 = (int)realm, r1 = r0 + 1;
  r0 = r0 < 0 ? 0 : r0 > 9 ? 9 : r0;
  r1 = r1 < 0 ? 0 : r1 > 9 ? 9 : r1;
  const double b0 = ((fit[r0][0] * att + fit[r0][1]) * att + fit[r0][2]) * att + fit[r0][3];
  const double b1 = ((fit[r1][0] * att + fit[r1][1]) * att + fit[r1][2]) * att + fit[r1][3];
  return b0 + (b1 - b0) * (realm - (int)realm);
}

// SOXR_HQ design for source_rate -> target_rate as one polyphase stage at the rate source_rate * up.
// Table mode (n_taps <= kMaxTableTaps): `table` = the taps, DC gain up.  Direct mode: `table` = the Kaiser window
// sampled at kWindowTable + 3 points of |t| / (centre + .5) in [0, 1] (cubic interpolation in the kernel).
ResamplePlan make_resample_plan(int source_rate, int target_rate, std::vector<double>& table) {
  ResamplePlan pl{};
  const int g = gcd_int(source_rate, target_rate);
  pl.up = target_rate / g;
  pl.down = source_rate / g;
  const double pi = 3.14159265358979323846;
  const double db = 20.0 * std::log10(2.0), rej = 20.0 * db;            // soxr_quality_spec(SOXR_HQ): 20 bit
  const double fp = 1.0 - .05 / ((1.6e-6 * rej - 7.5e-4) * rej + .646);  // pass-band end, of the lower Nyquist
  const double att = 21.0 * db;                                         // cr.c: (bits + 1) * 6.02 dB
  const double fn = (double)(pl.up > pl.down ? pl.up : pl.down);        // the filter rate's Nyquist / the lower Nyquist
  const double tr_bw = .5 * (1.0 - fp) / fn;
  pl.fc = 1.0 / fn - tr_bw;
  pl.beta = soxr_kaiser_beta(att, tr_bw * .5 / pl.fc);
  const double len = ((.0007528358 - 1.577737e-05 * pl.beta) * pl.beta + .6248022) * pl.beta + .06186902;
  int64_t n = (int64_t)std::ceil(len / tr_bw + 1.0);
  n = (n + 2) / 4 * 4 + 1;  // lsx_design_lpf with k = -4: 1 (mod 4)
  pl.n_taps = n;
  pl.centre = (n - 1) / 2;
  pl.inv_half = 1.0 / ((double)pl.centre + .5);  // lsx_make_lpf, rho = .5: the window ends half a tap outside the filter
  pl.gain = (double)pl.up;
  const double i0b = bessel_i0(pl.beta);
  pl.direct = n > kMaxTableTaps;
  if (pl.direct) {
    table.assign(kWindowTable + 3, 0.0);
    for (int i = 0; i < kWindowTable + 3; ++i) {  // entry i holds W((i - 1) / kWindowTable); W is even, W(u > 1) := 0
      const double u = std::fabs((double)(i - 1) / (double)kWindowTable);
      table[i] = u <= 1.0 ? bessel_i0(pl.beta * std::sqrt(1.0 - u * u)) / i0b : 0.0;
    }
    return pl;
  }
  table.assign((size_t)n, 0.0);
  for (int64_t i = 0; i <= pl.centre; ++i) {
    const double z = (double)(i - pl.centre), y = z * pl.inv_half;
    const double sinc = (z == 0.0) ? pl.fc : std::sin(pl.fc * pi * z) / (pi * z);
    table[i] = table[n - 1 - i] = sinc * bessel_i0(pl.beta * std::sqrt(1.0 - y * y)) / i0b * pl.gain;
  }
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

// y[k] = sum_j x[j] h[k * down + centre - j * up], the signal zero outside [0, n_in)
__global__ __launch_bounds__(256) void resample_poly_kernel(const float* __restrict__ x, int64_t n_in,
                                                            const double* __restrict__ taps, ResamplePlan pl,
                                                            float* __restrict__ y, int64_t n_out) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n_out) return;
  const int64_t base = k * (int64_t)pl.down + pl.centre;
  int64_t j_hi = base / pl.up;
  if (j_hi > n_in - 1) j_hi = n_in - 1;
  const int64_t lo_num = base - (pl.n_taps - 1);
  const int64_t j_lo = lo_num <= 0 ? 0 : (lo_num + pl.up - 1) / pl.up;
  double acc = 0.0;
  for (int64_t j = j_lo; j <= j_hi; ++j) acc += (double)x[j] * taps[base - j * pl.up];
  y[k] = (float)acc;
}

// the same sum with the taps evaluated in place: sinc in closed form, the Kaiser window by cubic (4-point Lagrange)
// interpolation in a table of kWindowTable intervals (interpolation error ~1e-15 of the window's peak)
__global__ __launch_bounds__(256) void resample_direct_kernel(const float* __restrict__ x, int64_t n_in,
                                                              const double* __restrict__ win, ResamplePlan pl,
                                                              float* __restrict__ y, int64_t n_out) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n_out) return;
  const double pi = 3.14159265358979323846;
  const int64_t base = k * (int64_t)pl.down;  // time of output k in filter-rate ticks
  int64_t j_hi = (base + pl.centre) / pl.up;
  if (j_hi > n_in - 1) j_hi = n_in - 1;
  const int64_t lo_num = base - pl.centre;
  const int64_t j_lo = lo_num <= 0 ? 0 : (lo_num + pl.up - 1) / pl.up;
  double acc = 0.0;
  for (int64_t j = j_lo; j <= j_hi; ++j) {
    const double t = (double)(base - j * pl.up);
    const double sinc = (t == 0.0) ? pl.fc : sin(pl.fc * pi * t) / (pi * t);
    const double u = fabs(t) * pl.inv_half * (double)kWindowTable;
    const int i = (int)u;
    const double f = u - (double)i;
    const double* w = win + i;  // w[0..3] = W at i - 1, i, i + 1, i + 2
    const double wv = w[0] * (-f * (f - 1.0) * (f - 2.0) / 6.0) + w[1] * ((f + 1.0) * (f - 1.0) * (f - 2.0) / 2.0) +
                      w[2] * (-(f + 1.0) * f * (f - 2.0) / 2.0) + w[3] * ((f + 1.0) * f * (f - 1.0) / 6.0);
    acc += (double)x[j] * sinc * wv;
  }
  y[k] = (float)(acc * pl.gain);
}

void launch_downmix(const float* pcm, int64_t n_frames, int channels, float* mono, hipStream_t stream) {
  if (n_frames <= 0) return;
  hipLaunchKernelGGL(downmix_kernel, dim3((unsigned)((n_frames + 255) / 256)), dim3(256), 0, stream, pcm, n_frames,
                     channels, mono);
}

void launch_resample(const float* x, int64_t n_in, const double* taps, const ResamplePlan& pl, float* y,
                     int64_t n_out, hipStream_t stream) {
  if (n_out <= 0) return;
  if (pl.direct)
    hipLaunchKernelGGL(resample_direct_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, stream, x, n_in,
                       taps, pl, y, n_out);
  else
    hipLaunchKernelGGL(resample_poly_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, stream, x, n_in,
                       taps, pl, y, n_out);
}

}  // namespace bp
