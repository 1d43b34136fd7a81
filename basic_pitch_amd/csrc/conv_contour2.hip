// Contour branch, second layer: Conv2D 8->1, 5x5, "same", sigmoid, FlattenFreqCh (basic_pitch/models.py:254-263,
// nn.py:105-119), c1 -> contour.  conv1 leaves relu(conv1) channel-last in HBM (conv_contour_march.hip + conv_contour_rim.hip);
// this kernel is HBM-paced: 1.48 MB read, 181,632 B written per window, 18.2 MFLOP on the f32 VALU.
#include "bp_common.h"

namespace bp {

// ---------------------------------------------------------------------------------------------------------
// conv2: Conv2D 8->1 5x5 + sigmoid as a march down the frames with ONE OUTPUT BIN PER LANE.
//   * a wave owns 64 consecutive bins of a flat (window, PADDED bin) index (16 windows x 268 columns of c1 = 67 waves
//     exactly; the 4 lanes per window on pad columns compute nothing that is stored) and a slab of frames; every row it loads its own pixel (8 channels = two 16-byte loads at a 32-byte lane stride: two
//     instructions cover 2 KB contiguous — the 4-bin-strip version before it made every load instruction touch 64
//     cache lines and stalled at 2.9 TB/s with the memory side alone taking 0.15 ms);
//   * the 4 neighbours (bins -2 .. +2) come from the other lanes through a per-wave LDS row (two channel-half
//     planes, consecutive lanes = consecutive 16-byte slots), lanes 0..3 also fetch the 2 + 2 halo pixels; the zero
//     padding of "same" is c1's own pad columns, which sit between the windows in the flat index: no masks;
//   * everything about rows is wave-uniform, so the 200 taps are scalar loads and operands of v_pk_fma_f32 (even /
//     odd channels in the two halves of a float2); the 5 open output rows live in 5 float2 accumulators;
//   * the next row's pixel is prefetched before the current row's 100 packed FMAs.
constexpr int kD2Group = 16;                              // windows per flat index group
constexpr int kD2Waves = kD2Group * kC1Row / 64;          // 67 waves per (group, slab)
static_assert(kD2Group * kC1Row % 64 == 0, "a group of windows fills whole waves");
static_assert(kC1Pad == 2, "the pad columns of c1 are the zero padding of the 5-tap rows");

struct Conv2Params {
  const float* c1;   // [n][172][kC1Row][8]
  const float* w2;   // [5 dt][5 dw][8 c]
  float bias;
  float* out;        // [n][172][264]
  int n_windows;
  int slab_rows;     // frames per slab
  int n_slabs;       // slabs per window
};

using v2f = __attribute__((ext_vector_type(2))) float;

// separate __restrict__ arguments (not a struct): the taps must be provably unclobbered to become scalar loads
__global__ __launch_bounds__(256) void contour_conv2_kernel(const float* __restrict__ c1_, const float* __restrict__ w2_,
                                                            float bias_, float* __restrict__ out_, int n_windows_,
                                                            int slab_rows_, int n_slabs_) {
  const Conv2Params p{c1_, w2_, bias_, out_, n_windows_, slab_rows_, n_slabs_};
  __shared__ __attribute__((aligned(16))) float4 xch[4][2][68];  // [wave][channel half][2 halo + 64 lanes + 2 halo]
  const int lane = threadIdx.x & 63;
  const int wv = wave_id();
  const int64_t wg = (int64_t)blockIdx.x * 4 + wv;                 // wave-uniform work item
  const int n_groups = (p.n_windows + kD2Group - 1) / kD2Group;
  if (wg >= (int64_t)n_groups * p.n_slabs * kD2Waves) return;      // whole waves only: no barrier below
  const int seg = (int)(wg % kD2Waves);
  const int slab = (int)((wg / kD2Waves) % p.n_slabs);
  const int group = (int)(wg / ((int64_t)kD2Waves * p.n_slabs));
  const int ta = slab * p.slab_rows;
  const int tb = ta + p.slab_rows < kFrames ? ta + p.slab_rows : kFrames;

  const int lin = seg * 64 + lane;
  const int wl = lin / kC1Row;                 // window inside the group
  const int pb = lin - wl * kC1Row;            // padded bin: column of c1 (real bins are 2 .. 265)
  const int win = group * kD2Group + wl;
  const bool wvalid = win < p.n_windows && pb >= kC1Pad && pb < kC1Pad + kFreqC;
  const float* src = p.c1 + (int64_t)(win < p.n_windows ? win : 0) * kC1Win + (int64_t)pb * 8;
  float* dst = p.out + (int64_t)(win < p.n_windows ? win : 0) * kPlaneC + (pb - kC1Pad);
  // lanes 0..3 also fetch the halo pixels of the wave: lane 0 / 1 -> columns -2 / -1 of lane 0's pixel, lane 2 / 3 ->
  // columns +1 / +2 of lane 63's pixel.  A real bin's neighbours are always columns of its own window's row (pads
  // included); where the flat index would leave the row, the lane at the wave's edge is a pad lane whose result is not
  // stored, and the column is clamped.
  const int lin_h = seg * 64 + (lane < 2 ? 0 : 63);
  const int wl_h = lin_h / kC1Row;
  int pb_h = lin_h - wl_h * kC1Row + (lane < 2 ? lane - 2 : lane - 1);
  pb_h = pb_h < 0 ? 0 : (pb_h > kC1Row - 1 ? kC1Row - 1 : pb_h);
  const int win_h = group * kD2Group + wl_h;
  const float* src_h = p.c1 + (int64_t)(win_h < p.n_windows ? win_h : 0) * kC1Win + (int64_t)pb_h * 8;
  const int slot_h = lane < 2 ? lane : 64 + lane;  // 0, 1, 66, 67

  float4 (*xw)[68] = xch[wv];
  v2f acc[5];  // acc[d] = output row r - 2 + d while input row r is being added
#pragma unroll
  for (int d = 0; d < 5; ++d) acc[d] = v2f{0.0f, 0.0f};

  const int r_first = ta - 2 > 0 ? ta - 2 : 0;
  const int r_last = tb + 1 < kFrames - 1 ? tb + 1 : kFrames - 1;
  const int64_t rs = (int64_t)kC1Row * 8;
  float4 own0 = *reinterpret_cast<const float4*>(src + r_first * rs);
  float4 own1 = *reinterpret_cast<const float4*>(src + r_first * rs + 4);
  float4 hal0 = own0, hal1 = own1;
  if (lane < 4) {
    hal0 = *reinterpret_cast<const float4*>(src_h + r_first * rs);
    hal1 = *reinterpret_cast<const float4*>(src_h + r_first * rs + 4);
  }
  for (int r = r_first; r <= r_last; ++r) {
    // publish this row's pixels to the wave, fetch the next row's
    xw[0][2 + lane] = own0;
    xw[1][2 + lane] = own1;
    if (lane < 4) {
      xw[0][slot_h] = hal0;
      xw[1][slot_h] = hal1;
    }
    if (r < r_last) {
      own0 = *reinterpret_cast<const float4*>(src + (r + 1) * rs);
      own1 = *reinterpret_cast<const float4*>(src + (r + 1) * rs + 4);
      if (lane < 4) {
        hal0 = *reinterpret_cast<const float4*>(src_h + (r + 1) * rs);
        hal1 = *reinterpret_cast<const float4*>(src_h + (r + 1) * rs + 4);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    v2f x[5][4];  // [dw][channel pair]
#pragma unroll
    for (int dw = 0; dw < 5; ++dw) {
      const float4 a = xw[0][lane + dw];
      const float4 b4 = xw[1][lane + dw];
      x[dw][0] = v2f{a.x, a.y};
      x[dw][1] = v2f{a.z, a.w};
      x[dw][2] = v2f{b4.x, b4.y};
      x[dw][3] = v2f{b4.z, b4.w};
    }
    __builtin_amdgcn_wave_barrier();  // every lane has read the row before the next one overwrites it
    // one open row after the other (the 5 accumulators side by side was measured slower: 0.130 vs 0.115 ms — the tap
    // stream, not the dependent chain, sets the pace)
#pragma unroll
    for (int d = 0; d < 5; ++d) {
      const int t = r - 2 + d;
      if (t < ta || t >= tb) continue;  // wave-uniform
      const float* __restrict__ wd = w2_ + (4 - d) * 40;
#pragma unroll
      for (int dw = 0; dw < 5; ++dw) {
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2) {
          // one chain of 20 dependent v_pk_fma_f32 (each behind a wait state): two interleaved chains per open row
          // measured the same (round 5: 0.112 vs 0.112 ms) — other waves fill the slots; 200 plain v_fmac_f32 instead
          // of the 100 packed ones: 0.132 ms; two rows of loads in flight instead of one: 0.118; six resident waves per
          // SIMD instead of five (5 slabs of 35 frames): 0.103 - 0.107 against 0.106 - 0.109, seven or eight: 0.123
          // (profiles/r05_c_stalls.md: 58 % of the wave cycles sit in s_waitcnt, 11 % issue vector instructions.  TWO bins
          // per lane — six pixel columns serve two outputs, every tap feeds two packed FMAs, 108 VGPRs, four waves per SIMD —
          // measured 0.127 / 0.114 / 0.115 ms with 4 / 5 / 7 slabs against 0.112: what the shared taps save the lost
          // occupancy costs.)
          const v2f wv2 = {wd[dw * 8 + 2 * c2], wd[dw * 8 + 2 * c2 + 1]};
          acc[d] = __builtin_elementwise_fma(wv2, x[dw][c2], acc[d]);
        }
      }
    }
    const int t_out = r - 2;
    if (t_out >= ta && t_out < tb && wvalid) dst[(int64_t)t_out * kFreqC] = sigmoidf_fast((acc[0].x + acc[0].y) + p.bias);
#pragma unroll
    for (int d = 0; d < 4; ++d) acc[d] = acc[d + 1];
    acc[4] = v2f{0.0f, 0.0f};
  }
  // output rows whose last input rows lie below the window (zero rows): flush what is still open
  for (int t = r_last - 1; t < tb; ++t) {
    if (t >= ta && wvalid) dst[(int64_t)t * kFreqC] = sigmoidf_fast((acc[0].x + acc[0].y) + p.bias);
#pragma unroll
    for (int d = 0; d < 4; ++d) acc[d] = acc[d + 1];
    acc[4] = v2f{0.0f, 0.0f};
  }
}

void launch_contour_conv2(const float* c1, const float* w2, float bias, float* contour, int n_windows, int n_cu,
                          hipStream_t stream) {
  // <= 96 VGPRs: 5 waves per SIMD resident.  Sub-batches of 256 windows, each cut into as many frame slabs as keep one
  // launch within the resident waves (4 slabs = 9 % halo re-read at 256 windows on 256 CUs).
  const int64_t slots = (int64_t)n_cu * 4 * 5;  // resident waves (<= 96 VGPRs)
  const int per_launch = 256;
  for (int w0 = 0; w0 < n_windows; w0 += per_launch) {
    const int n = n_windows - w0 < per_launch ? n_windows - w0 : per_launch;
    const int n_groups = (n + kD2Group - 1) / kD2Group;
    int n_slabs = (int)(slots / ((int64_t)n_groups * kD2Waves));
    n_slabs = n_slabs < 1 ? 1 : (n_slabs > 16 ? 16 : n_slabs);
    const int slab_rows = (kFrames + n_slabs - 1) / n_slabs;
    n_slabs = (kFrames + slab_rows - 1) / slab_rows;
    Conv2Params p{c1 + (int64_t)w0 * kC1Win, w2, bias, contour + (int64_t)w0 * kPlaneC, n, slab_rows, n_slabs};
    const int64_t waves = (int64_t)n_groups * n_slabs * kD2Waves;
    hipLaunchKernelGGL(contour_conv2_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, p.c1, p.w2, p.bias,
                       p.out, p.n_windows, p.slab_rows, p.n_slabs);
  }
}

}  // namespace bp
