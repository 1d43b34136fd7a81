// The dense half of note decoding on the device (round 5; SURVEY.md §8f rank 1: "HIP for the dense parts: onset inference,
// peak picking, thresholding").  The posteriorgrams of a track are already in HBM when the CNN is done; what the host's
// note tracker (csrc/note_decode.cpp, the sequential half) needs of them is
//   * the note map (T x 88 float32) — it follows energies along a pitch and averages amplitudes,
//   * WHERE the onset peaks are: a bitmap (T x 12 bytes: 88 bits per frame), not the onset map,
//   * the pitch bend of bin f at frame t: T x 88 int8, not the 264-bin contour map,
// 7.0 MB per 3-minute track instead of 27.6 MB over PCIe — written into page-locked host memory by the kernels
// themselves when the caller's buffers are (no copy engine: the engine is busy bringing the next file in) —, and the
// three dense scans leave the host cores.
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
constexpr int kNdBitsRow = 12;  // bytes per frame of the onset-peak bitmap: 88 bits + 8 zero bits (rows of whole dwords)

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

__global__ __launch_bounds__(64) void nd_stats_init_kernel(NdStats* st) {
  if (threadIdx.x == 0) {
    st->max_on_ord = f2ord(-__int_as_float(0x7f800000));
    st->nan = 0;
    st->max_fd_bits = 0ull;
  }
}

// Extrema of the two maps.  A wave walks frames (lanes take bins lane and lane + 64), a workgroup folds its four waves in
// LDS and publishes ONE atomic per quantity (as first written, every frame's wave published its own: 15 k serialised
// atomics per 3-minute track on three addresses, 0.36 ms — more than the CQT of the track).
__global__ __launch_bounds__(256) void nd_stats_kernel(const float* __restrict__ note, const float* __restrict__ onset, int64_t T,
                                                       int infer, NdStats* __restrict__ st) {
  __shared__ float s_mo[4];
  __shared__ double s_fd[4];
  __shared__ int s_nan[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float mo = -__int_as_float(0x7f800000);
  double mfd = 0.0;
  int nan = 0;
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < T; t += (int64_t)gridDim.x * 4) {
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
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float a = __shfl_xor(mo, o);
    mo = a > mo ? a : mo;
    const double b = __shfl_xor(mfd, o);
    mfd = b > mfd ? b : mfd;
    nan |= __shfl_xor(nan, o);
  }
  if (lane == 0) s_mo[wave] = mo, s_fd[wave] = mfd, s_nan[wave] = nan;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      mo = s_mo[w] > mo ? s_mo[w] : mo;
      mfd = s_fd[w] > mfd ? s_fd[w] : mfd;
      nan |= s_nan[w];
    }
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
                                                            const NdStats* __restrict__ st, uint32_t* __restrict__ bits) {
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
  if (lane < 3) bits[t * 3 + lane] = lane == 0 ? (uint32_t)b0 : (lane == 1 ? (uint32_t)(b0 >> 32) : (uint32_t)b1);
}

// Pitch bends.  A workgroup takes kNdBendFrames consecutive frames: their contour rows go to LDS as float64 once (264
// conversions per frame instead of 88 x 51), each between two margins of -inf, so that every bin's window is the full 51
// taps — a tap outside the row yields -inf x gauss = -inf, which never exceeds the running maximum and never comes
// first (the running index starts at the first tap inside the row, where the host's loop starts).  An item is (frame,
// bin): 16 x 88 = 1408 items over 256 threads, consecutive lanes consecutive bins (3 doubles apart: conflict-free
// ds_read_b64), the Gaussian in registers, four vector operations per tap (multiply, compare, maximum, select) where the
// wave-per-frame form with table-driven loop bounds issued twelve at 69 % lane use: 72 -> ~20 us per 3-minute track.
// The products and comparisons are the host loop's float64 operations in the host loop's order: the same argmax.  A NaN
// anywhere in the block's rows (np.argmax: the first NaN wins) sends the block through the comparison that handles it.
constexpr int kNdBendFrames = 16, kNdBendPad = 26, kNdBendPitch = kNdFC + 2 * kNdBendPad;  // rows 16-byte aligned
static_assert(kNdBendPad >= 25 && (kNdBendPad * 8) % 16 == 0 && (kNdBendPitch * 8) % 16 == 0, "aligned rows");

__device__ __forceinline__ int nd_bend_argmax(const double* __restrict__ win, const double (&g)[26], int first) {
  int best = first;
  double bestv = -__longlong_as_double(0x7ff0000000000000ll);
#pragma unroll
  for (int j = 0; j < 51; ++j) {
    const double v = win[j] * g[j <= 25 ? j : 50 - j];
    best = v > bestv ? j : best;
    bestv = __builtin_fmax(bestv, v);
  }
  return best;
}

// np.argmax over the taps [first, 50] with a NaN somewhere in the block: the first maximum, the first NaN wins (the margins
// behind the row are -inf and never win).  A rolled loop with the Gaussian from LDS: this path is for broken inputs.
__device__ __forceinline__ int nd_bend_argmax_nan(const double* __restrict__ win, const double* __restrict__ g, int first) {
  int best = first;
  double bestv = win[first] * g[first];
#pragma unroll 1
  for (int j = first + 1; j < 51; ++j) {
    const double v = win[j] * g[j];
    const bool take = (v > bestv) | ((v != v) & (bestv == bestv));
    bestv = take ? v : bestv;
    best = take ? j : best;
  }
  return best;
}

__global__ __launch_bounds__(256, 3) void nd_bend_kernel(const float* __restrict__ contour, int64_t T, const int4* __restrict__ tab,
                                                      const double* __restrict__ gauss, int8_t* __restrict__ bend) {
  __shared__ __attribute__((aligned(16))) double s_row[kNdBendFrames * kNdBendPitch];
  __shared__ int s_start[kNdF], s_first[kNdF];
  __shared__ double s_g[51];
  __shared__ int s_nan;
  const int tid = threadIdx.x;
  const int64_t t0 = (int64_t)blockIdx.x * kNdBendFrames;
  const double ninf = -__longlong_as_double(0x7ff0000000000000ll);
  if (tid == 0) s_nan = 0;
  if (tid >= 128 && tid < 128 + 51) s_g[tid - 128] = gauss[tid - 128];
  if (tid < kNdF) {
    const int4 w = tab[tid];  // f0, n, g0, shift: tap j of the 51 reads row bin f0 - g0 + j; the first inside the row is g0
    s_start[tid] = kNdBendPad + w.x - w.z;
    s_first[tid] = w.z;
  }
  for (int i = tid; i < kNdBendFrames * 2 * kNdBendPad; i += 256) {
    const int r = i / (2 * kNdBendPad), c = i - r * (2 * kNdBendPad);
    s_row[r * kNdBendPitch + (c < kNdBendPad ? c : kNdFC + c)] = ninf;
  }
  // the Gaussian exp(-(j - 25)^2 / 50) is symmetric bit for bit (the host squares j - 25): 26 values, in vector registers
  // (left to itself the compiler keeps the words in scalar registers it does not have)
  double g[26];
#pragma unroll
  for (int j = 0; j < 26; ++j) {
    g[j] = gauss[j];
    asm volatile("" : "+v"(g[j]));
  }
  __syncthreads();
  // the block's rows are contiguous in memory: 16 x 264 floats as float4s (264 = 4 x 66: no float4 straddles two rows)
  int nan = 0;
  const int64_t n_rows = T - t0 < kNdBendFrames ? T - t0 : kNdBendFrames;
  const float4* src = reinterpret_cast<const float4*>(contour + t0 * kNdFC);
  for (int e = tid; e < kNdBendFrames * (kNdFC / 4); e += 256) {
    const int r = e / (kNdFC / 4), c4 = e - r * (kNdFC / 4);
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (r < n_rows) v = src[e];
    nan |= (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
    double* d = &s_row[r * kNdBendPitch + kNdBendPad + 4 * c4];
    *reinterpret_cast<double2*>(d) = double2{(double)v.x, (double)v.y};
    *reinterpret_cast<double2*>(d + 2) = double2{(double)v.z, (double)v.w};
  }
  if (nan) s_nan = 1;
  __syncthreads();
  if (s_nan == 0) {  // block-uniform
#pragma unroll 1
    for (int item = tid; item < kNdBendFrames * kNdF; item += 256) {
      const int r = item / kNdF, b = item - r * kNdF;
      if (r >= n_rows) break;
      const int best = nd_bend_argmax(&s_row[r * kNdBendPitch + s_start[b]], g, s_first[b]);
      bend[t0 * kNdF + item] = (int8_t)(best - 25);  // = (best - g0) - shift, shift = 25 - g0
    }
  } else {
#pragma unroll 1
    for (int item = tid; item < kNdBendFrames * kNdF; item += 256) {
      const int r = item / kNdF, b = item - r * kNdF;
      if (r >= n_rows) break;
      const int best = nd_bend_argmax_nan(&s_row[r * kNdBendPitch + s_start[b]], s_g, s_first[b]);
      bend[t0 * kNdF + item] = (int8_t)(best - 25);
    }
  }
}

// device -> page-locked host memory, by the compute queue: the note map, the bitmap and the bend map of a track in one
// launch (4-byte words, a wave writes 256 contiguous bytes; the copy engine stays free for the next file's samples), the
// stats record with them — and the device record back to its initial values for the next track (one launch and one
// 16-byte copy fewer per track).
__global__ __launch_bounds__(256) void nd_export_kernel(const uint32_t* __restrict__ s0, uint32_t* __restrict__ d0, int64_t n0,
                                                        const uint32_t* __restrict__ s1, uint32_t* __restrict__ d1, int64_t n1,
                                                        const uint32_t* __restrict__ s2, uint32_t* __restrict__ d2, int64_t n2,
                                                        NdStats* __restrict__ st, NdStats* __restrict__ st_dst) {
  const int64_t step = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n0; i += step) d0[i] = s0[i];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n1; i += step) d1[i] = s1[i];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += step) d2[i] = s2[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *st_dst = *st;
    st->max_on_ord = f2ord(-__int_as_float(0x7f800000));
    st->nan = 0;
    st->max_fd_bits = 0ull;
  }
}

// note / onset / contour: device maps of T frames.  Leaves the bitmap ([T][12] bytes), the bend map ([T][88] bytes, when
// `bend` != null) and the stats on the device (the stats record must hold its initial values: launch_note_stats_init); `lo`, `hi`: the bins constrain_frequency keeps (0, 88: none to zero).
void launch_note_candidates(float* note, float* onset, const float* contour, int64_t T, int lo, int hi, int infer,
                            double onset_thresh, const void* tab, const double* gauss, void* stats, uint8_t* bits,
                            int8_t* bend, hipStream_t s) {
  if (T <= 0) return;
  NdStats* st = static_cast<NdStats*>(stats);
  const unsigned frames4 = (unsigned)((T + 3) / 4);
  if (lo > 0 || hi < kNdF)
    hipLaunchKernelGGL(nd_constrain_kernel, dim3((unsigned)((T * kNdF + 255) / 256)), dim3(256), 0, s, note, onset, T * kNdF, lo, hi);
  hipLaunchKernelGGL(nd_stats_kernel, dim3(frames4 < 512u ? frames4 : 512u), dim3(256), 0, s, note, onset, T, infer, st);
  hipLaunchKernelGGL(nd_candidates_kernel, dim3(frames4), dim3(256), 0, s, note, onset, T, infer, onset_thresh, st,
                     reinterpret_cast<uint32_t*>(bits));
  if (bend)
    hipLaunchKernelGGL(nd_bend_kernel, dim3((unsigned)((T + kNdBendFrames - 1) / kNdBendFrames)), dim3(256), 0, s, contour, T,
                       static_cast<const int4*>(tab), gauss, bend);
}

void launch_note_stats_init(void* stats, hipStream_t s) {
  hipLaunchKernelGGL(nd_stats_init_kernel, dim3(1), dim3(64), 0, s, static_cast<NdStats*>(stats));
}

// the three results to device-visible host pointers (sizes in bytes, multiples of 4; a null destination is skipped)
void launch_note_export(const void* note, void* note_dst, int64_t note_bytes, const void* bits, void* bits_dst,
                        int64_t bits_bytes, const void* bend, void* bend_dst, int64_t bend_bytes, void* stats,
                        void* stats_dst, hipStream_t s) {
  hipLaunchKernelGGL(nd_export_kernel, dim3(64), dim3(256), 0, s, static_cast<const uint32_t*>(note),
                     static_cast<uint32_t*>(note_dst), note_dst ? note_bytes / 4 : 0, static_cast<const uint32_t*>(bits),
                     static_cast<uint32_t*>(bits_dst), bits_dst ? bits_bytes / 4 : 0, static_cast<const uint32_t*>(bend),
                     static_cast<uint32_t*>(bend_dst), bend_dst ? bend_bytes / 4 : 0, static_cast<NdStats*>(stats),
                     static_cast<NdStats*>(stats_dst));
}

}  // namespace bp
