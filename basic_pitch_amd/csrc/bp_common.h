// Shared constants and device helpers for the gfx950 Basic Pitch kernels.
// Geometry follows the reference's frozen graph (basic_pitch/constants.py:25-47).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

namespace bp {

// A/B switches.  The product library reads no behaviour switch from the environment: the env-selected variants of a
// kernel (BP_CONV1, BP_ONSET, BP_RIM, BP_RESAMPLE, BP_CONTOUR_PARTS, BP_BRANCH_PROF, BP_STAGE_NOSYNC) and the kernels only
// they can reach exist in builds with -DBP_AB_KERNELS (basic_pitch_amd/build.py: build_library(ab=True), the library
// the comparison tests and tools load through BASIC_PITCH_AMD_LIB); in the default build this returns null.
inline const char* ab_env(const char* name) {
#ifdef BP_AB_KERNELS
  return std::getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

constexpr int kAudioN = 43844;   // constants.py:47
constexpr int kFrames = 172;     // constants.py:44
constexpr int kBins = 309;       // models.py:172-177
constexpr int kFreqC = 264;      // constants.py:36
constexpr int kFreqN = 88;       // constants.py:35
constexpr int kOctaves = 9;      // nnaudio.py:543
constexpr int kBpo = 36;         // bins per octave
constexpr int kTaps = 256;       // n_fft of the top-octave kernels / lowpass length
constexpr int kPlaneC = kFrames * kFreqC;  // 45408
constexpr int kPlaneN = kFrames * kFreqN;  // 15136
constexpr int kPyrStride = 43712;
// Pre-split z tensor `zp` (normalised + BatchNorm-ed CQT as f16 hi | f16 lo << 16), zero padded so that the
// harmonic-stack gathers (bin f - 20 .. f + 283 shifted by -36 .. +101) and the frame halo need no masks:
// [kZRowsP = 1 + 172 + 1 frames][kZRow = 56 + 309 + 83 words]; frame t, bin g lives at (t + 1) * kZRow + kZPadL + g.
constexpr int kZRow = 448;
constexpr int kZPadL = 56;
constexpr int kZRowsP = kFrames + 2;
constexpr int kZWin = kZRowsP * kZRow;   // words per window
// relu(conv1) of the contour branch: [172][kC1Row = 2 + 264 + 2 bins][8 channels] fp32, pad bins zero
constexpr int kC1Pad = 2;
constexpr int kC1Row = kFreqC + 2 * kC1Pad;
constexpr int kC1Win = kFrames * kC1Row * 8;  // floats per window

// pyramid level k (1..8) lives at kPyrOff[k] inside a window's pyr row; level 0 is the audio itself
__host__ __device__ constexpr int level_len(int k) {
  int l = kAudioN;
  for (int i = 0; i < k; ++i) l = (l - 2) / 2 + 1;  // nnaudio.py:269-279 (pad 127, 256 taps, stride 2)
  return l;
}
__host__ __device__ constexpr int pyr_off(int k) {
  int off = 0;
  for (int i = 1; i < k; ++i) off += (level_len(i) + 3) & ~3;
  return off;
}
static_assert(level_len(1) == 21922 && level_len(8) == 171, "pyramid geometry");

// Extended range for 44.1 kHz input (BASELINE.json configs[4]; SURVEY.md App. A.6 — not a reference behaviour): the same
// 36 kernels and low-pass, hop 512, one more octave on top: 10 levels, 345 bins, windows of 87,688 samples.  Level
// k >= 1 of this pyramid has exactly the length of level k - 1 of the 22.05 kHz one.
constexpr int kAudioNExt = 2 * kAudioN;          // 87688
constexpr int kOctavesExt = kOctaves + 1;         // 10
constexpr int kBinsExt = kBins + kBpo;            // 345
constexpr int kPyrStrideExt = kAudioN + kPyrStride;  // level 1 (43844 samples), then levels 2..9 at the std offsets
static_assert((kAudioNExt - 2) / 2 + 1 == kAudioN, "level 1 of the extended pyramid");
static_assert(pyr_off(8) + level_len(8) <= kPyrStride, "pyr stride");

// harmonic shifts round(36*log2(h)), h = 0.5,1,2..7   (nn.py:51-54, models.py:213-218)
__host__ __device__ constexpr int harm_shift(int c) {
  constexpr int s[8] = {-36, 0, 36, 57, 72, 84, 93, 101};
  return s[c];
}

// Tap order of the onset conv1 on v_mfma_f32_16x16x32_f16 (onset_march16.hip; shared with the host packer in bp_api.hip):
// k-step s, lane group g = lane >> 4 takes tap (dt, dw) of the 5 x 5 window (models.py:295-304) for its 8 channels.
// k-steps 0..4: image row dt = s, dw = {0, 3, 1, 4}[g]; k-step 5: dw = 2 of rows dt = g; k-step 6: (4, 2) and three
// zero-weight dummies that read the same slot.  Lane pairs (g even, g odd) differ by dw 0 or +3: conflict-free reads.
constexpr int kOnset16KSteps = 7;
__host__ __device__ constexpr int onset16_dt(int s, int g) { return s < 5 ? s : (s == 5 ? g : 4); }
__host__ __device__ constexpr int onset16_dw(int s, int g) { return s < 5 ? (g == 0 ? 0 : (g == 1 ? 3 : (g == 2 ? 1 : 4))) : 2; }
__host__ __device__ constexpr bool onset16_live(int s, int g) { return s < 6 || g == 0; }

// Split-precision operands: x = hi + lo / kLoScale with hi = rn_f16(x), lo = rn_f16((x - hi) * kLoScale).
// The scale keeps the residual (<= 2^-12 |x|) inside f16's normal exponent range; products are
// hi*hi + (lo*hi + hi*lo) * kLoUnscale, accumulated in fp32 (separate accumulators per scale).
constexpr float kLoScale = 2048.0f, kLoUnscale = 1.0f / 2048.0f;

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;

// Whole-wave min / max by DPP (row scans + row_bcast15 / row_bcast31, GFX9 controls): the result is in lane 63.
// 6 VALU operations instead of 6 x (address + ds_bpermute + op).
#define BP_DPP_STEP(op, v, ctrl, rmask)                                                                                \
  v = op(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), \
                                                                   ctrl, rmask, 0xf, false)))
__device__ __forceinline__ float wave_min_lane63(float v) {
  BP_DPP_STEP(fminf, v, 0x111, 0xf);  // row_shr:1
  BP_DPP_STEP(fminf, v, 0x112, 0xf);  // row_shr:2
  BP_DPP_STEP(fminf, v, 0x114, 0xf);  // row_shr:4
  BP_DPP_STEP(fminf, v, 0x118, 0xf);  // row_shr:8   -> lane 15 of every row holds the row's result
  BP_DPP_STEP(fminf, v, 0x142, 0xa);  // row_bcast:15 into rows 1, 3
  BP_DPP_STEP(fminf, v, 0x143, 0xc);  // row_bcast:31 into rows 2, 3
  return v;
}
__device__ __forceinline__ float wave_max_lane63(float v) {
  BP_DPP_STEP(fmaxf, v, 0x111, 0xf);
  BP_DPP_STEP(fmaxf, v, 0x112, 0xf);
  BP_DPP_STEP(fmaxf, v, 0x114, 0xf);
  BP_DPP_STEP(fmaxf, v, 0x118, 0xf);
  BP_DPP_STEP(fmaxf, v, 0x142, 0xa);
  BP_DPP_STEP(fmaxf, v, 0x143, 0xc);
  return v;
}
#undef BP_DPP_STEP

// lo halves of a split pair: rn_f16((v - float(hi)) * 2^11), on the packed-f32 pipe.
// (Round 3 tried ONE mixed-precision FMA per value instead — v_fma_mixlo_f16 / v_fma_mixhi_f16 through inline assembly,
// reading the f16 hi straight from its half of the packed register: two VALU operations per value fewer, bit-identical in
// the note, onset and CQT kernels, 1 % faster.  In a new kernel the same helper produced lo halves that were off (the
// onset map moved by 7e-5, deterministically).  Cause: a VALU write of ONE 16-bit half of a register needs a wait state
// before the next instruction that touches that register (gfx940's destination-select forwarding hazard) — the compiler
// inserts it for its own instructions and cannot see into inline assembly; with `s_nop 1` behind each of the two
// instructions the kernel was bit-identical again.  With the nops the form saves nothing measurable: removed.)
__device__ __forceinline__ uint32_t split_lo(uint32_t hi2, f32x2 v) {
  uint32_t l = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  // plain f32 operations on purpose: beside matrix instructions a v_pk_add_f32 / v_pk_mul_f32 costs ~3.5 x a plain VALU
  // operation (tools/ubench/mfma_shadow.hip, profiles/r04_ubench_shadow.md); the files are built with -fno-slp-vectorize
  const f16x2 h = __builtin_bit_cast(f16x2, hi2);
  // ((v * 2^11) - hi * 2^11 as one v_fma_mixlo/hi_f16 per value — conversions of hi and of the result folded in, 2.5
  // instead of 4 operations per value — measured the same: onset + 1.5 us, note - 1.4 us at 24 fewer VALU per row; the
  // VOP3P mixed-precision operations cost beside matrix instructions what the packed ones do.)
  const float d0 = (v.x - (float)h.x) * 2048.0f;
  const float d1 = (v.y - (float)h.y) * 2048.0f;
  const f16x2 lh = {(_Float16)d0, (_Float16)d1};
  l = __builtin_bit_cast(uint32_t, lh);
#endif
  return l;
}

// The same with hi rounded to nearest (the CQT operands: half the representation error of the truncating form, which
// the log of weak bins amplifies)
__device__ __forceinline__ void split_f16x2_rn(f32x2 v, uint32_t& hi2, uint32_t& lo2) {
#if defined(__HIP_DEVICE_COMPILE__)
  const f16x2 h = {(_Float16)v.x, (_Float16)v.y};  // one v_cvt_pk_f16_f32
  hi2 = __builtin_bit_cast(uint32_t, h);
  lo2 = split_lo(hi2, v);
#endif
}

// Split of two values at once for in-kernel operands (|v| < 65504): hi = v truncated to f16 (one v_cvt_pkrtz_f16_f32
// for the pair; any hi within an f16 ulp of v serves, the residual carries the rest exactly), lo = rn_f16((v - hi) *
// 2^11) on the packed-f32 pipe.  ~3 VALU operations per value instead of 5.
__device__ __forceinline__ void split_f16x2(f32x2 v, uint32_t& hi2, uint32_t& lo2) {
#if defined(__HIP_DEVICE_COMPILE__)
  hi2 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(v.x, v.y));
  lo2 = split_lo(hi2, v);
#endif
}

// ReLU as ONE integer max on the bit pattern (negative floats, -0 included, are negative integers).  fmaxf(v, 0) on a
// value that comes straight out of a matrix instruction costs two v_max_f32: the compiler has to quiet a possible
// signalling NaN first.
__device__ __forceinline__ float relu_f32(float v) {
  const int b = __float_as_int(v);
  return __int_as_float(b > 0 ? b : 0);
}

// The pieces of tracks that make up one chunk of windows (bp_infer_tracks packs the windows of consecutive tracks into
// full chunks): one launch windows them all, one launch un-overlaps all three posteriorgrams of all of them — per-piece
// launches of these small kernels cost 5-9 us each, ~10 % of a whole-tracks job.
constexpr int kMaxTrackSegs = 16;
struct TrackSeg {
  const float* samples;  // the track (device)
  float* out[3];         // its un-overlapped note / onset / contour maps (device), T rows each
  int64_t n_samples;
  int64_t first_window;  // first window of this piece within the track
  int64_t total_rows;    // T of the track
  int n_windows;         // windows of this piece
  int at;                // its first window's slot in the chunk
};
struct TrackSegs {
  TrackSeg seg[kMaxTrackSegs];
  int n;
};

// order-preserving float <-> int map so per-window min/max can use integer atomics
__device__ __forceinline__ int f2ord(float f) {
  int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

struct LogConsts {
  float eps;     // 1e-10                 (signal.py:175)
  float s0, s1;  // 1/ln(10), 10          (math.py:21-32 as frozen: ONNX nodes 193-194)
  float bn_a;    // folded BatchNorm scale (models.py:188-189; ONNX node 211)
  float bn_b;    // folded BatchNorm shift (ONNX node 212)
};

// NormalizedLog tail + BN affine for one element (signal.py:177-183, divide_no_nan)
__device__ __forceinline__ float norm_bn(float lp, float mn, float range, const LogConsts& k) {
  float off = lp - mn;
  float nrm = (range == 0.0f) ? 0.0f : off / range;
  return __fadd_rn(__fmul_rn(nrm, k.bn_a), k.bn_b);  // Mul then Add in the frozen graph: no FMA contraction
}

// The same map as the default path evaluates it (round 5): the affine part folded into one scale per window,
// z = (lp - min) * (bn_a / range) + bn_b — one subtraction and one FMA per element instead of sub, IEEE division, mul, add
// (the division alone is ~10 instructions).  <= 2 ulp of the product nrm x bn_a (<= 4.8e-7 absolute; z itself passes
// through 0) from the sequence above — tests/test_gpu_parity.py::test_zpack_folded_affine_is_within_two_ulp_of_the_graph_order; every kernel of the default path
// (the fused filterbank's normalise phase, zpack_kernel) uses THESE two functions, so a window's words do not depend on
// which of them produced it.  range == 0 (a silent window): scale 0, z = bn_b, as divide_no_nan gives.
__device__ __forceinline__ float norm_scale(float mn, float mx, const LogConsts& k) {
  const float range = mx - mn;
  return range == 0.0f ? 0.0f : __fdiv_rn(k.bn_a, range);
}
__device__ __forceinline__ float norm_bn_k(float lp, float mn, float nk, float bn_b) {
  float z = __fmaf_rn(__fsub_rn(lp, mn), nk, bn_b);
#if defined(__HIP_DEVICE_COMPILE__)
  // z is an fp32 value in a register before anything converts it: without this the compiler may fold the f16 conversion
  // of the split into the FMA (v_fma_mixlo_f16) in one kernel and not in another
  asm volatile("" : "+v"(z));
#endif
  return z;
}

// zero-phase rational polyphase resampler (audio_ingest.hip), libsoxr SOXR_HQ design at the rate source * up
struct ResamplePlan {
  int up, down;
  int direct;       // 0: `table` holds the n_taps taps; 1: taps evaluated in the kernel, `table` holds the window
  int64_t n_taps;   // odd, 1 (mod 4)
  int64_t centre;   // (n_taps - 1) / 2: output k sits at filter index k * down + centre
  double fc;        // 6 dB point as a fraction of the filter rate's Nyquist
  double beta;      // Kaiser beta
  double inv_half;  // 1 / (centre + .5): window argument per tap
  double gain;      // up
  int64_t rev_off;  // 2 : 1 plans: offset (in doubles) of the taps once more in reverse order, zero-padded to whole
                    // blocks of 32 (resample_half_kernel reads them through the scalar cache); 0: absent
};
constexpr int64_t kMaxTableTaps = (int64_t)1 << 22;  // 32 MB of float64 taps; beyond it the direct kernel
constexpr int kWindowTable = 1 << 16;

__device__ __forceinline__ float sigmoidf_exact(float x) { return 1.0f / (1.0f + expf(-x)); }
// the same on the hardware's 1-ulp exp2 / rcp (4 VALU operations instead of ~35; <= 2e-7 off on a value in [0, 1]):
// the default path's kernels are paced by their instruction count
__device__ __forceinline__ float sigmoidf_fast(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}

// Workgroup barrier for LDS hand-offs only.  __syncthreads() is a workgroup-scope fence + s_barrier, and on gfx9 that
// fence drains vmcnt: every global load still in flight is waited for at every barrier, which defeats fetching the
// next step's data ahead of a barrier.  This one waits for the wave's own LDS operations (lgkmcnt) and meets the other
// waves; global loads stay in flight, the compiler still waits for them before their first use.  Not a fence for
// global memory: do not use it to publish global stores to other waves.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);  // vmcnt = 63, expcnt = 7, lgkmcnt = 0
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

}  // namespace bp
