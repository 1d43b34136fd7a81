// The dense half of note decoding on the device (round 5; SURVEY.md §8f rank 1: "HIP for the dense parts: onset inference,
// peak picking, thresholding").  The posteriorgrams of a track are already in HBM when the CNN is done; what the host's
// note tracker (csrc/note_decode.cpp, the sequential half) needs of them is
//   * the note map (T x 88 float32) — it follows energies along a pitch and averages amplitudes,
//   * WHERE the onset peaks are: a bitmap (T x 88 bits), not the onset map,
//   * the pitch bend of bin f at frame t: T x 88 int8, not the 264-bin contour map,
// 7.1 MB per 3-minute track instead of 27.6 MB over PCIe, and the three dense scans leave the host cores.
//
// Reference lines (spotify/basic-pitch v0.4.0, basic_pitch/note_creation.py):
//   constrain_frequency   314-343   bins outside [min, max] zeroed in the note and onset maps
//   get_infered_onsets    289-311   onsets = max(onsets, max(onsets) * fd / max(fd)), fd = max(0, min_n (frames[t] - frames[t - n])), n = 1, 2
//   output_to_notes_polyphonic 394-402   scipy.signal.argrelmax along time (strictly above both neighbours), >= onset_thresh
//   get_pitch_bends       182-219   argmax over the 51-bin Gaussian-weighted window of the contour row, minus the centre
// The arithmetic is the host decoder's, operation for operation (float64 differences and quotient, IEEE division; float64
// products for the bend's argmax with the host's Gaussian table): the same bits, the same events —
// tests/test_gpu_parity.py::test_device_note_candidates_give_the_host_decoders_events.
#include "bp_common.h"

namespace bp {

constexpr int kNdF = 88, kNdFC = 264;

struct NdStats {
  int max_on_ord;                  // f2ord(max onset)
  int nan;                         // a NaN in the note or onset map: the host decides (numpy's propagation rules)
  unsigned long long max_fd_bits;  // bits of max fd (a non-negative double: its bits order like the value)
};

__global__ __launch_bounds__(256) void nd_constrain_kernel(float* __restrict__ note, float* __restrict__ onset, int64_t n_cells,
                                                           int lo, int hi) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_cells) return;
  const int f = (int)(i % kNdF);
  if (f < lo || f >= hi) note[i] = onset[i] = 0.0f;
}

__global__ __launch_bounds__(256) void nd_stats_init_kernel(NdStats* st) {
  st->max_on_ord = f2ord(-__int_as_float(0x7f800000));
  st->nan = 0;
  st->max_fd_bits = 0ull;
}

// one wave per frame: lanes take bins lane and lane + 64
__global__ __launch_bounds__(256) void nd_stats_kernel(const float* __restrict__ note, const float* __restrict__ onset, int64_t T,
                                                       int infer, NdStats* __restrict__ st) {
  const int lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  float mo = -__int_as_float(0x7f800000);
  double mfd = 0.0;
  int nan = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int f = lane + 64 * h;
    if (f >= kNdF) break;
    const float o = onset[t * kNdF + f], n0 = note[t * kNdF + f];
    nan |= (o != o) | (n0 != n0);
    mo = o > mo ? o : mo;
    if (infer && t >= 2) {
      const double d1 = (double)n0 - (double)note[(t - 1) * kNdF + f], d2 = (double)n0 - (double)note[(t - 2) * kNdF + f];
      const double d = d1 < d2 ? d1 : d2;
      mfd = d > mfd ? d : mfd;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float a = __shfl_xor(mo, o);
    mo = a > mo ? a : mo;
    const double b = __shfl_xor(mfd, o);
    mfd = b > mfd ? b : mfd;
    nan |= __shfl_xor(nan, o);
  }
  if (lane == 0) {
    atomicMax(&st->max_on_ord, f2ord(mo));
    if (mfd > 0.0) atomicMax(&st->max_fd_bits, (unsigned long long)__double_as_longlong(mfd));
    if (nan) atomicOr(&st->nan, 1);
  }
}

// np.maximum: NaN if either operand is NaN
__device__ __forceinline__ double nd_np_maximum(double a, double b) {
  if (a != a || b != b) return __longlong_as_double(0x7ff8000000000000ll);
  return a > b ? a : b;
}

__global__ __launch_bounds__(256) void nd_candidates_kernel(const float* __restrict__ note, const float* __restrict__ onset,
                                                            int64_t T, int infer, double onset_thresh,
                                                            const NdStats* __restrict__ st, uint8_t* __restrict__ bits) {
  const int lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const double max_on_d = (double)ord2f(st->max_on_ord);
  const double max_fd = __longlong_as_double((long long)st->max_fd_bits);
  auto fd_at = [&](int64_t tt, int f) -> double {
    if (tt < 2) return 0.0;
    const double n0 = (double)note[tt * kNdF + f];
    const double d1 = n0 - (double)note[(tt - 1) * kNdF + f], d2 = n0 - (double)note[(tt - 2) * kNdF + f];
    const double d = d1 < d2 ? d1 : d2;
    return d < 0 ? 0.0 : d;
  };
  auto on_at = [&](int64_t tt, int f) -> double {
    const double o = (double)onset[tt * kNdF + f];
    if (!infer) return o;
    const double scaled = (max_on_d * fd_at(tt, f)) / max_fd;  // 0 / 0 -> NaN when nothing rises, like numpy
    return nd_np_maximum(o, scaled);
  };
  bool c[2] = {false, false};
  if (t >= 1 && t <= T - 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int f = lane + 64 * h;
      if (f >= kNdF) break;
      const double v = on_at(t, f);
      c[h] = v > on_at(t - 1, f) && v > on_at(t + 1, f) && v >= onset_thresh;
    }
  }
  const unsigned long long b0 = __ballot(c[0]), b1 = __ballot(c[1]);
  if (lane < 11) bits[t * 11 + lane] = (uint8_t)(lane < 8 ? (b0 >> (8 * lane)) : (b1 >> (8 * (lane - 8))));
}

__global__ __launch_bounds__(256) void nd_bend_kernel(const float* __restrict__ contour, int64_t T, const int4* __restrict__ tab,
                                                      const double* __restrict__ gauss, int8_t* __restrict__ bend) {
  const int lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const float* row = contour + t * kNdFC;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int b = lane + 64 * h;
    if (b >= kNdF) break;
    const int4 w = tab[b];  // f0, n, g0, shift
    int best = 0;
    double bestv = (double)row[w.x] * gauss[w.z];
    for (int j = 1; j < w.y; ++j) {
      const double v = (double)row[w.x + j] * gauss[w.z + j];
      if (v > bestv || (v != v && bestv == bestv)) {  // np.argmax: first maximum, NaN wins
        bestv = v;
        best = j;
      }
    }
    bend[t * kNdF + b] = (int8_t)(best - w.w);
  }
}

// note / onset / contour: device maps of T frames.  Leaves the bitmap, the bend map (when `bend` != null) and the stats
// on the device; `lo`, `hi`: the bins constrain_frequency keeps (0, 88: none to zero).
void launch_note_candidates(float* note, float* onset, const float* contour, int64_t T, int lo, int hi, int infer,
                            double onset_thresh, const void* tab, const double* gauss, void* stats, uint8_t* bits,
                            int8_t* bend, hipStream_t s) {
  if (T <= 0) return;
  NdStats* st = static_cast<NdStats*>(stats);
  const unsigned frames4 = (unsigned)((T + 3) / 4);
  if (lo > 0 || hi < kNdF)
    hipLaunchKernelGGL(nd_constrain_kernel, dim3((unsigned)((T * kNdF + 255) / 256)), dim3(256), 0, s, note, onset, T * kNdF, lo, hi);
  hipLaunchKernelGGL(nd_stats_init_kernel, dim3(1), dim3(1), 0, s, st);
  hipLaunchKernelGGL(nd_stats_kernel, dim3(frames4), dim3(256), 0, s, note, onset, T, infer, st);
  hipLaunchKernelGGL(nd_candidates_kernel, dim3(frames4), dim3(256), 0, s, note, onset, T, infer, onset_thresh, st, bits);
  if (bend)
    hipLaunchKernelGGL(nd_bend_kernel, dim3(frames4), dim3(256), 0, s, contour, T, static_cast<const int4*>(tab), gauss, bend);
}

}  // namespace bp
