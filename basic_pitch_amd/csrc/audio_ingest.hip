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

#include "../../include/basic_pitch_amd.h"
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

// The same downmix straight from the file's sample format (bp_pcm_format): 16-bit stereo is half the bytes of its float
// form over PCIe and the host never converts it.  Scales as in basic_pitch_amd/audio.py read_wav / the WAV reader of
// file_pipeline.cpp (powers of two: exact), so the mono signal is bit-identical to the float path's.
template <int FMT>
__device__ __forceinline__ float pcm_sample(const uint8_t* __restrict__ raw, int64_t i) {
  if (FMT == BP_PCM_S16) return (float)reinterpret_cast<const int16_t*>(raw)[i] * (1.0f / 32768.0f);
  if (FMT == BP_PCM_U8) return ((float)raw[i] - 128.0f) * (1.0f / 128.0f);
  if (FMT == BP_PCM_S24) {
    const uint8_t* p = raw + 3 * i;
    int32_t v = (int32_t)p[0] | ((int32_t)p[1] << 8) | ((int32_t)p[2] << 16);
    if (v >= 1 << 23) v -= 1 << 24;
    return (float)v * (1.0f / 8388608.0f);
  }
  if (FMT == BP_PCM_S32) return (float)((double)reinterpret_cast<const int32_t*>(raw)[i] * (1.0 / 2147483648.0));
  if (FMT == BP_PCM_F64) return (float)reinterpret_cast<const double*>(raw)[i];
  return reinterpret_cast<const float*>(raw)[i];
}

// Four frames per thread (round 5: one frame per thread moved 63 MB in 34 us, a quarter of what the memory system streams;
// in a file job this kernel runs once per file).  16-bit stereo — what a CD-quality WAV holds — takes them as one 16-byte
// load and one 16-byte store; every format and channel count computes a frame exactly as before.
template <int FMT>
__global__ __launch_bounds__(256) void downmix_raw_kernel(const uint8_t* __restrict__ raw, int64_t n_frames, int channels,
                                                          float* __restrict__ mono) {
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n_frames) return;
  if (FMT == BP_PCM_S16 && channels == 2 && i4 + 4 <= n_frames &&
      ((reinterpret_cast<uintptr_t>(raw) | reinterpret_cast<uintptr_t>(mono)) & 15) == 0) {
    const uint4 v = *reinterpret_cast<const uint4*>(raw + i4 * 4);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float s = 0.0f;
      s += (float)(int16_t)(w[k] & 0xffffu) * (1.0f / 32768.0f);
      s += (float)(int16_t)(w[k] >> 16) * (1.0f / 32768.0f);
      o[k] = s / 2.0f;
    }
    *reinterpret_cast<float4*>(mono + i4) = float4{o[0], o[1], o[2], o[3]};
    return;
  }
  for (int64_t i = i4; i < i4 + 4 && i < n_frames; ++i) {
    if (channels == 1) {
      mono[i] = pcm_sample<FMT>(raw, i);
      continue;
    }
    float s = 0.0f;
    for (int c = 0; c < channels; ++c) s += pcm_sample<FMT>(raw, i * channels + c);
    mono[i] = s / (float)channels;
  }
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

// The same sum, term for term in the same order, for a block of 256 outputs whose input span fits LDS: the block's slice of
// x is staged once (each input sample is used by n_taps / down outputs — 195 at 2 : 1, each of which reads 389 —, where the kernel above fetched every
// one of them from memory again: 1.55 ms for a 3-minute track, more than the CQT + CNN of its 110 windows) and, when the
// table is small (pure decimation: 389 taps at 2 : 1), the taps too.
constexpr int kResTileX = 4096;     // floats of x per block
constexpr int kResTileTaps = 1024;  // float64 taps held in LDS
template <bool TAPS_LDS>
__global__ __launch_bounds__(256) void resample_tiled_kernel(const float* __restrict__ x, int64_t n_in,
                                                             const double* __restrict__ taps, ResamplePlan pl,
                                                             float* __restrict__ y, int64_t n_out) {
  __shared__ float xs[kResTileX];
  __shared__ double hs[TAPS_LDS ? kResTileTaps : 1];
  auto j_lo_of = [&](int64_t base) {
    const int64_t lo_num = base - (pl.n_taps - 1);
    return lo_num <= 0 ? (int64_t)0 : (lo_num + pl.up - 1) / pl.up;
  };
  auto j_hi_of = [&](int64_t base) {
    const int64_t j = base / pl.up;
    return j > n_in - 1 ? n_in - 1 : j;
  };
  const int64_t k0 = (int64_t)blockIdx.x * 256;
  const int64_t k_last = k0 + 255 < n_out - 1 ? k0 + 255 : n_out - 1;
  const int64_t j0 = j_lo_of(k0 * (int64_t)pl.down + pl.centre);
  const int64_t j1 = j_hi_of(k_last * (int64_t)pl.down + pl.centre);
  for (int64_t j = j0 + threadIdx.x; j <= j1; j += 256) xs[j - j0] = x[j];
  if (TAPS_LDS)
    for (int i = threadIdx.x; i < (int)pl.n_taps; i += 256) hs[i] = taps[i];
  __syncthreads();
  const int64_t k = k0 + threadIdx.x;
  if (k >= n_out) return;
  const int64_t base = k * (int64_t)pl.down + pl.centre;
  const int64_t j_lo = j_lo_of(base), j_hi = j_hi_of(base);
  double acc = 0.0;
  for (int64_t j = j_lo; j <= j_hi; ++j) {
    const int64_t t = base - j * pl.up;
    acc += (double)xs[j - j0] * (TAPS_LDS ? hs[t] : taps[t]);
  }
  y[k] = (float)acc;
}

// 2 : 1 (44.1 kHz files, the common case): the same sum once more, each thread OUT consecutive outputs.  Output k at
// step s (tap n_taps - 1 - s, i.e. j ascending as above) reads x[X0 + W t + s + 2 i] for k = K0 + OUT t + i, W = 2 OUT,
// so the block's slice of x is stored de-interleaved by W (sub[e % W][e / W]): a thread holds one value of each of the
// W sub-sequences in registers, step s multiplies OUT of them with ONE tap and replaces one — per OUT multiply-adds
// one conflict-free ds_read_b32, one conversion and one broadcast tap read, where the tiled kernel paid two reads and a
// conversion for each.  Samples outside the signal and the steps that pad the tap count to a multiple of W contribute
// x * 0 or 0 * tap = 0, which leaves a float64 accumulator as it is: the result is the kernels' above, bit for bit.
// 3-minute track (1.54 G multiply-adds = 39 us at the fp64 vector rate): 303 us (tiled) -> 83 us with OUT = 4 -> 70 us with OUT = 8.
// Round 5: OUT = 16 and the taps through the scalar cache (the tap index is the same for every lane: an s_load, not a
// broadcast LDS read) — per 16 multiply-adds one ds_read_b32 and one conversion.
#ifndef BP_HALF_OUT
#define BP_HALF_OUT 16
#endif
constexpr int kHalfOut = BP_HALF_OUT, kHalfW = 2 * kHalfOut;
// row length of a sub-sequence in LDS: 256 threads + 1024 / W tap blocks + 1, and = 1 (mod 32) (W = 32) or 2 (mod 32)
// (W = 16), so that the 32 lanes of a fill pass hit 32 different banks
constexpr int kHalfSub = kHalfW == 32 ? 289 : 322;
static_assert((kHalfW == 16 && kHalfSub % 32 == 2) || (kHalfW == 32 && kHalfSub % 32 == 1), "fill pattern");
static_assert(kHalfSub >= 256 + kResTileTaps / kHalfW + 1, "a block's slice fits");
__global__ __launch_bounds__(256) void resample_half_kernel(const float* __restrict__ x, int64_t n_in,
                                                            const double* __restrict__ taps, ResamplePlan pl,
                                                            float* __restrict__ y, int64_t n_out) {
  __shared__ float sub[kHalfW][kHalfSub];
  const int M = (int)pl.n_taps, nb = (M + kHalfW - 1) / kHalfW, t = threadIdx.x;
  const int64_t K0 = (int64_t)blockIdx.x * (256 * kHalfOut);
  const int64_t X0 = 2 * K0 + pl.centre - (M - 1);  // x index behind sub[0][0]
  const int n_tile = kHalfW * (256 + nb + 1);
  for (int e = t; e < n_tile; e += 256) {
    const int64_t X = X0 + e;
    sub[e % kHalfW][e / kHalfW] = (X >= 0 && X < n_in) ? x[X] : 0.0f;
  }
  __syncthreads();
  const double* __restrict__ hr = taps + pl.rev_off;  // taps[M - 1 - s] at s, zeros from M to the end of its block of 32
  double r[kHalfW], acc[kHalfOut];
#pragma unroll
  for (int i = 0; i < kHalfOut; ++i) acc[i] = 0.0;
#pragma unroll
  for (int q = 0; q < kHalfW; ++q) r[q] = (double)sub[q][t];
  for (int b = 0; b < nb; ++b) {
#pragma unroll
    for (int q = 0; q < kHalfW; ++q) {
      const double h = hr[kHalfW * b + q];  // wave-uniform address: scalar loads, a block of taps at a time
#pragma unroll
      for (int i = 0; i < kHalfOut; ++i) acc[i] = __builtin_fma(r[(q + 2 * i) % kHalfW], h, acc[i]);
      r[q] = (double)sub[q][t + b + 1];
    }
  }
  const int64_t k = K0 + kHalfOut * t;
  if (k + kHalfOut - 1 < n_out) {
#pragma unroll
    for (int i = 0; i < kHalfOut; i += 4)
      *reinterpret_cast<float4*>(y + k + i) =
          make_float4((float)acc[i], (float)acc[i + 1], (float)acc[i + 2], (float)acc[i + 3]);
  } else {
#pragma unroll
    for (int i = 0; i < kHalfOut; ++i)
      if (k + i < n_out) y[k + i] = (float)acc[i];
  }
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

void launch_downmix_raw(const void* raw, int format, int64_t n_frames, int channels, float* mono, hipStream_t stream) {
  if (n_frames <= 0) return;
  const dim3 grid((unsigned)((n_frames + 1023) / 1024));  // four frames per thread
  const uint8_t* p = static_cast<const uint8_t*>(raw);
  switch (format) {
    case BP_PCM_S16: hipLaunchKernelGGL(downmix_raw_kernel<BP_PCM_S16>, grid, dim3(256), 0, stream, p, n_frames, channels, mono); break;
    case BP_PCM_S24: hipLaunchKernelGGL(downmix_raw_kernel<BP_PCM_S24>, grid, dim3(256), 0, stream, p, n_frames, channels, mono); break;
    case BP_PCM_S32: hipLaunchKernelGGL(downmix_raw_kernel<BP_PCM_S32>, grid, dim3(256), 0, stream, p, n_frames, channels, mono); break;
    case BP_PCM_U8: hipLaunchKernelGGL(downmix_raw_kernel<BP_PCM_U8>, grid, dim3(256), 0, stream, p, n_frames, channels, mono); break;
    case BP_PCM_F64: hipLaunchKernelGGL(downmix_raw_kernel<BP_PCM_F64>, grid, dim3(256), 0, stream, p, n_frames, channels, mono); break;
    default: hipLaunchKernelGGL(downmix_raw_kernel<BP_PCM_F32>, grid, dim3(256), 0, stream, p, n_frames, channels, mono); break;
  }
}

// mode 0: the fastest kernel that fits; 1: the one-thread-per-output kernel; 2: at most the tiled one (A/B runs, BP_RESAMPLE)
void launch_resample(const float* x, int64_t n_in, const double* taps, const ResamplePlan& pl, float* y,
                     int64_t n_out, int mode, hipStream_t stream) {
  if (n_out <= 0) return;
  // inputs a block of 256 outputs reaches (upper bound)
  const int64_t span = (255 * (int64_t)pl.down + pl.n_taps - 1) / pl.up + 2;
  const dim3 per_output((unsigned)((n_out + 255) / 256));
  if (pl.direct)
    hipLaunchKernelGGL(resample_direct_kernel, per_output, dim3(256), 0, stream, x, n_in, taps, pl, y, n_out);
  else if (mode == 0 && pl.up == 1 && pl.down == 2 && pl.rev_off > 0 && pl.n_taps <= kResTileTaps && (reinterpret_cast<uintptr_t>(y) & 15) == 0)
    hipLaunchKernelGGL(resample_half_kernel, dim3((unsigned)((n_out + 256 * kHalfOut - 1) / (256 * kHalfOut))), dim3(256), 0, stream, x, n_in, taps,
                       pl, y, n_out);
  else if (mode != 1 && span <= kResTileX && pl.n_taps <= kResTileTaps)
    hipLaunchKernelGGL(resample_tiled_kernel<true>, per_output, dim3(256), 0, stream, x, n_in, taps, pl, y, n_out);
  else if (mode != 1 && span <= kResTileX)
    hipLaunchKernelGGL(resample_tiled_kernel<false>, per_output, dim3(256), 0, stream, x, n_in, taps, pl, y, n_out);
  else
    hipLaunchKernelGGL(resample_poly_kernel, per_output, dim3(256), 0, stream, x, n_in, taps, pl, y, n_out);
}

}  // namespace bp
