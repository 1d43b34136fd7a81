// CQT filterbank: per pyramid level, 172 frames x 36 complex Hann-windowed kernels, then
// magnitude * sqrt(length), log-power, and the per-window min / max of the log-power (partial
// extrema per tile, folded by a one-wave-per-window reduction: no atomics).
//
// Reference behaviour (spotify/basic-pitch v0.4.0):
//   basic_pitch/layers/nnaudio.py:216-256  get_cqt_complex: reflect-pad 128, conv1d(real), -conv1d(imag),
//                                           stride hop_k = 256 / 2^k on level k
//   basic_pitch/layers/nnaudio.py:640-661  lower octaves prepended, bottom 15 bins dropped,
//                                           * sqrt(lengths), magnitude
//   basic_pitch/layers/signal.py:171-178   power, 10*log10(power + 1e-10), per-example min (and max)
//
// MI355X mapping.  Per (window, level, 16-frame tile) the filterbank is a [16 x K] x [K x 72] product
// with a Hankel A (A[t][i] = xp[t*hop + i]) — exact-f32 MFMA v_mfma_f32_16x16x4_f32.  The 72 filter
// columns are split into five 16-wide tiles whose K range is clipped to the kernels' non-zero
// support (61 % of the 256 taps), and the four waves of a workgroup each own one share of those
// tiles with the B fragments (the filters) held in VGPRs for the whole persistent kernel:
//     wave 0: re 0..15   taps 20..235           wave 1: im 0..15   taps 20..235
//     wave 2: re 16..31  taps 48..207  +  {re,im} 32..35 taps 68..127
//     wave 3: im 16..31  taps 48..207  +  {re,im} 32..35 taps 128..187
// (54/54/55/55 MFMAs per tile).  The signal tile is staged once in LDS with reflection applied
// and a +2-float skew per hop so the strided Hankel reads are bank-conflict free; partial tiles
// are exchanged through LDS for the magnitude/log epilogue.
//
// Roofline (this kernel): bound = f32 MFMA; algorithmic work 9 * 172 * 72 * 256 MAC = 57.06 MFLOP
// per window (dense; 34.8 MFLOP on non-zero taps), algorithmic bytes 175,376 (audio) + 174,764
// (pyramid) read + 212,592 written.
#include "bp_common.h"

namespace bp {

constexpr int kFbThreads = 256;
constexpr int kFbTileFrames = 16;
constexpr int kFbTilesPerLevel = (kFrames + kFbTileFrames - 1) / kFbTileFrames;  // 11
constexpr int kFbMaxSteps = 55;  // B fragments per lane

// Role tables: segment A -> accumulators accA0/accA1 (alternating), segment B -> accB.
template <int ROLE> struct FbRole;
template <> struct FbRole<0> { static constexpr int A0 = 5, A1 = 59, B0 = 0, B1 = 0; };
template <> struct FbRole<1> { static constexpr int A0 = 5, A1 = 59, B0 = 0, B1 = 0; };
template <> struct FbRole<2> { static constexpr int A0 = 12, A1 = 52, B0 = 17, B1 = 32; };
template <> struct FbRole<3> { static constexpr int A0 = 12, A1 = 52, B0 = 32, B1 = 47; };

__host__ __device__ constexpr int fb_pad(int hop) { return hop >= 4 ? 2 : 0; }
__host__ __device__ constexpr int fb_span(int hop) { return 15 * hop + 256; }
__host__ __device__ constexpr int fb_lds_floats(int hop) {
  return fb_span(hop) + fb_pad(hop) * (fb_span(hop) / hop + 1);
}

constexpr int kExRow = 17;                         // padded row of an exchanged 16x16 tile
constexpr int kExTile = kFbTileFrames * kExRow;    // 272 floats

template <int ROLE, int HOP>
__device__ __forceinline__ void fb_role_compute(const float* __restrict__ sig_lds,
                                                const float (&breg)[kFbMaxSteps],
                                                float* __restrict__ exch, int lane) {
  using R = FbRole<ROLE>;
  constexpr int PADH = fb_pad(HOP);
  constexpr int NA = R::A1 - R::A0;
  constexpr int S0 = (R::B1 > R::B0 && R::B0 < R::A0) ? R::B0 : R::A0;
  constexpr int S1 = (R::B1 > R::A1) ? R::B1 : R::A1;
  const int fr = lane & 15, kk = lane >> 4;
  const float* ap = sig_lds + fr * (HOP + PADH) + kk;
  f32x4 accA0 = {0.f, 0.f, 0.f, 0.f}, accA1 = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = S0; s < S1; ++s) {
    const bool inA = (s >= R::A0 && s < R::A1);
    const bool inB = (s >= R::B0 && s < R::B1);
    if (!inA && !inB) continue;
    const float a = ap[4 * s + PADH * ((4 * s) / HOP)];
    if (inA) {
      if ((s - R::A0) & 1)
        accA1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, breg[s - R::A0], accA1, 0, 0, 0);
      else
        accA0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, breg[s - R::A0], accA0, 0, 0, 0);
    }
    if (inB) accB = __builtin_amdgcn_mfma_f32_16x16x4f32(a, breg[NA + s - R::B0], accB, 0, 0, 0);
  }
  // C layout of 16x16x4: col = lane & 15 (filter), row = (lane >> 4) * 4 + reg (frame)
  constexpr int slotA = (ROLE == 0) ? 0 : (ROLE == 1) ? 1 : (ROLE == 2) ? 2 : 3;
  float* ea = exch + slotA * kExTile + (kk * 4) * kExRow + fr;
#pragma unroll
  for (int r = 0; r < 4; ++r) ea[r * kExRow] = accA0[r] + accA1[r];
  if (R::B1 > R::B0) {
    float* eb = exch + ((ROLE == 2) ? 4 : 5) * kExTile + (kk * 4) * kExRow + fr;
#pragma unroll
    for (int r = 0; r < 4; ++r) eb[r * kExRow] = accB[r];
  }
}

template <int LEVEL>
__global__ __launch_bounds__(kFbThreads) void cqt_filterbank_kernel(
    const float* __restrict__ sig, int64_t sig_stride, const float* __restrict__ bfrag,
    const float* __restrict__ sqrt_len, float* __restrict__ lp, float2* __restrict__ mmp, int n_windows,
    LogConsts kc) {
  constexpr int HOP = 256 >> LEVEL;
  constexpr int PADH = fb_pad(HOP);
  constexpr int SPAN = fb_span(HOP);
  constexpr int L = level_len(LEVEL);
  __shared__ float sig_lds[fb_lds_floats(HOP)];
  __shared__ float exch[6 * kExTile];

  const int lane = threadIdx.x & 63;
  const int role = wave_id();

  // B fragments (filter taps) stay in registers for the whole kernel.
  float breg[kFbMaxSteps];
  {
    const float* bp_ = bfrag + (size_t)role * kFbMaxSteps * 64 + lane;
#pragma unroll
    for (int j = 0; j < kFbMaxSteps; ++j) breg[j] = bp_[j * 64];
  }

  const int n_items = n_windows * kFbTilesPerLevel;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int b = item / kFbTilesPerLevel;
    const int t0 = (item - b * kFbTilesPerLevel) * kFbTileFrames;
    const float* x = sig + (int64_t)b * sig_stride;

    // stage xp[t0*HOP + p] = reflect(x)[t0*HOP + p - 128], p in [0, SPAN)   (nnaudio.py:229,300-301)
    for (int p = threadIdx.x; p < SPAN; p += kFbThreads) {
      int g = t0 * HOP + p - 128;
      g = g < 0 ? -g : g;
      g = g >= L ? 2 * (L - 1) - g : g;
      g = g < 0 ? 0 : (g >= L ? L - 1 : g);  // only reachable for the padding frames 172..175
      sig_lds[p + PADH * (p / HOP)] = x[g];
    }
    __syncthreads();

    switch (role) {
      case 0: fb_role_compute<0, HOP>(sig_lds, breg, exch, lane); break;
      case 1: fb_role_compute<1, HOP>(sig_lds, breg, exch, lane); break;
      case 2: fb_role_compute<2, HOP>(sig_lds, breg, exch, lane); break;
      default: fb_role_compute<3, HOP>(sig_lds, breg, exch, lane); break;
    }
    __syncthreads();

    // epilogue: 16 frames x 36 filters -> log-power + min/max
    float vmin = __int_as_float(0x7f800000), vmax = -__int_as_float(0x7f800000);
    for (int idx = threadIdx.x; idx < kFbTileFrames * kBpo; idx += kFbThreads) {
      const int fr = idx / kBpo, k = idx - fr * kBpo;
      const int t = t0 + fr;
      const int bin = (kOctaves - 1 - LEVEL) * kBpo + k - 15;  // nnaudio.py:640-642
      if (t >= kFrames || bin < 0) continue;
      float re, im;
      const float* e = exch + fr * kExRow;
      if (k < 16) {
        re = e[0 * kExTile + k];
        im = e[1 * kExTile + k];
      } else if (k < 32) {
        re = e[2 * kExTile + k - 16];
        im = e[3 * kExTile + k - 16];
      } else {
        re = e[4 * kExTile + k - 32] + e[5 * kExTile + k - 32];
        im = e[4 * kExTile + k - 28] + e[5 * kExTile + k - 28];
      }
      const float sl = sqrt_len[bin];
      re = __fmul_rn(re, sl);  // nnaudio.py:650: scale before squaring
      im = __fmul_rn(im, sl);
      const float mag = sqrtf(__fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im)));  // nnaudio.py:661
      const float pw = __fmul_rn(mag, mag);                                      // signal.py:174
      const float v = __fmul_rn(__fmul_rn(logf(__fadd_rn(pw, kc.eps)), kc.s0), kc.s1);
      lp[((int64_t)b * kFrames + t) * kBins + bin] = v;
      vmin = fminf(vmin, v);
      vmax = fmaxf(vmax, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      vmin = fminf(vmin, __shfl_xor(vmin, o));
      vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    }
    // per-(window, level, tile, wave) partial extrema; mm_reduce_kernel folds the 396 of a window.
    // (No atomics: 22 k same-line L2 atomics per level serialise; min/max are order-free anyway.)
    if (lane == 0)
      mmp[(((int64_t)b * kOctaves + LEVEL) * kFbTilesPerLevel + t0 / kFbTileFrames) * 4 + role] =
          make_float2(vmin, vmax);
    // the next iteration's staging barrier also orders exch reuse
  }
}

constexpr int kMmPartials = kOctaves * kFbTilesPerLevel * 4;  // 396 per window

// one wave per window: fold the partial extrema into mm[b] = (ord(min), ord(max))
__global__ __launch_bounds__(64) void mm_reduce_kernel(const float2* __restrict__ mmp, int n_partials,
                                                       int* __restrict__ mm) {
  const int b = blockIdx.x;
  float vmin = __int_as_float(0x7f800000), vmax = -__int_as_float(0x7f800000);
  for (int i = threadIdx.x; i < n_partials; i += 64) {
    const float2 p = mmp[(int64_t)b * n_partials + i];
    vmin = fminf(vmin, p.x);
    vmax = fmaxf(vmax, p.y);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    vmin = fminf(vmin, __shfl_xor(vmin, o));
    vmax = fmaxf(vmax, __shfl_xor(vmax, o));
  }
  if (threadIdx.x == 0) {
    mm[2 * b] = f2ord(vmin);
    mm[2 * b + 1] = f2ord(vmax);
  }
}

template <int LEVEL>
static void launch_fb_level(const float* audio, const float* pyr, const float* bfrag,
                            const float* sqrt_len, float* lp, float2* mmp, int n_windows, LogConsts kc,
                            int grid, hipStream_t stream) {
  const float* sig = (LEVEL == 0) ? audio : pyr + pyr_off(LEVEL);
  const int64_t stride = (LEVEL == 0) ? kAudioN : kPyrStride;
  const int items = n_windows * kFbTilesPerLevel;
  const int g = items < grid ? items : grid;
  hipLaunchKernelGGL(cqt_filterbank_kernel<LEVEL>, dim3(g), dim3(kFbThreads), 0, stream, sig, stride,
                     bfrag, sqrt_len, lp, mmp, n_windows, kc);
}

void launch_mm_reduce(const float* scratch, int* mm, int n_windows, int n_partials, hipStream_t stream) {
  hipLaunchKernelGGL(mm_reduce_kernel, dim3(n_windows), dim3(64), 0, stream,
                     reinterpret_cast<const float2*>(scratch), n_partials, mm);
}

// sized for the extended 10-level pyramid as well
size_t filterbank_scratch_floats(int n_windows) { return (size_t)n_windows * (kMmPartials + kFbTilesPerLevel * 4) * 2; }

void launch_filterbank(const float* audio, const float* pyr, const float* bfrag,
                       const float* sqrt_len, float* lp, int* mm, float* scratch, int n_windows,
                       LogConsts kc, int n_cu, hipStream_t stream) {
  float2* mmp = reinterpret_cast<float2*>(scratch);
  const int grid = n_cu * 4;
  launch_fb_level<0>(audio, pyr, bfrag, sqrt_len, lp, mmp, n_windows, kc, grid, stream);
  launch_fb_level<1>(audio, pyr, bfrag, sqrt_len, lp, mmp, n_windows, kc, grid, stream);
  launch_fb_level<2>(audio, pyr, bfrag, sqrt_len, lp, mmp, n_windows, kc, grid, stream);
  launch_fb_level<3>(audio, pyr, bfrag, sqrt_len, lp, mmp, n_windows, kc, grid, stream);
  launch_fb_level<4>(audio, pyr, bfrag, sqrt_len, lp, mmp, n_windows, kc, grid, stream);
  launch_fb_level<5>(audio, pyr, bfrag, sqrt_len, lp, mmp, n_windows, kc, grid, stream);
  launch_fb_level<6>(audio, pyr, bfrag, sqrt_len, lp, mmp, n_windows, kc, grid, stream);
  launch_fb_level<7>(audio, pyr, bfrag, sqrt_len, lp, mmp, n_windows, kc, grid, stream);
  launch_fb_level<8>(audio, pyr, bfrag, sqrt_len, lp, mmp, n_windows, kc, grid, stream);
  hipLaunchKernelGGL(mm_reduce_kernel, dim3(n_windows), dim3(64), 0, stream, mmp, kMmPartials, mm);
}

}  // namespace bp
