// Contour branch, second layer: Conv2D 8->1, 5x5, "same", sigmoid, FlattenFreqCh (basic_pitch/models.py:254-263,
// nn.py:105-119), c1 -> contour.  conv1 leaves relu(conv1) channel-last in HBM (conv_contour_march.hip + conv_contour_rim.hip);
// the layer is HBM-paced: 1.48 MB read, 181,632 B written per window, 18.2 MFLOP.
//
// Round 6: `contour_conv2_proj_kernel` (the default) puts the channel contraction on the matrix cores; the round-2 vector
// kernel `contour_conv2_kernel` (200 FMAs per output behind scalar tap loads: 58 % of its wave cycles in s_waitcnt, neither at
// the HBM rate nor at the vector rate) stays in the A/B library behind BP_CONV2=valu.
#include <stdlib.h>

#include "bp_common.h"

namespace bp {

// ---------------------------------------------------------------------------------------------------------
// conv2 as a TAP PROJECTION on the matrix cores (the decomposition of the note / onset heads): for every INPUT pixel
//   P[tap (dt, df)][pixel] = sum_c W[dt][df][c] c1[pixel][c]        M = 25 taps (32 rows), K = 8 channels, N = 32 pixels
// — K is the 8 channels: no Toeplitz padding, no im2col, no LDS staging.  out[t][f] = sum_taps P[dt, df][t + dt - 2][f + df - 2] is then 25 additions per output on the vector pipe
// (the round-2 kernel: 200 FMAs).
//   * B operand: lane (n = lane & 31, h = lane >> 5) holds channels 4 h .. 4 h + 3 of pixel n = the 16 bytes one
//     global_load_dwordx4 brings (a wave-row of 32 pixels is 1 KB contiguous), split to f16 hi / lo in registers — every c1
//     value is loaded and split exactly once per strip; A operand: the taps as two resident fragments (8 VGPRs), all three
//     products into ONE accumulator at scale 2^11 (note_march16.hip) — packed along K, see the kernel.
//   * C layout: lane half h = 0 holds frame taps dt = 0, 1, 2 (rows 5 dt + df), half 1 dt = 3, 4: marching down the frames a
//     lane adds its taps into a chain of open output rows — X0 = P[dt0] + IN, X1 = X0' + P[dt1], X2 = X1' + P[dt2] —, half 0's
//     finished three-tap sum crosses to half 1 (ds_bpermute lane ^ 32) as the IN of its two-tap chain, and the finished row
//     leaves half 1 two frames later; the five frequency taps are four more ds_bpermute (lanes n - 2 .. n + 2; the two tiles of
//     a strip hand over their edge pixels through the source lanes' select).
//   * a work item is (16 windows, frame slab, strip of 64 columns of the flat (window, padded bin) index, 60 of them stored):
//     the zero padding of "same" is c1's own pad columns, which sit between the windows in the flat index — no masks; rows
//     outside the window contribute P = 0.
// Roofline: HBM (1.48 MB in + 7 % strip overlap + 9 % slab halo, 182 KB out per window); matrix 4 instructions per 64 pixels.
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int kP2Group = 16;                         // windows per flat index group
constexpr int kP2Cols = kP2Group * kC1Row;           // 4288 flat columns per group
constexpr int kP2Strip = 64, kP2Stride = 60;         // columns a wave reads / stores per row
constexpr int kP2Strips = (kP2Cols + kP2Stride - 1) / kP2Stride;  // 72
static_assert(kC1Pad == 2, "the pad columns of c1 are the zero padding of the 5-tap rows");

struct Conv2ProjParams {
  const float* c1;      // [n][172][kC1Row][8]
  const uint4* wfrag;   // pack_conv2_proj: [A1 = hi 2^11 | hi][A2 = lo 2^11 | 0] x 64 lanes x (8 x f16)
  float bias;
  float* out;           // [n][172][264]
  int n_windows;
  int slab_rows;        // frames per slab
  int n_slabs;          // slabs per window
};

#ifndef P2_OCC
#define P2_OCC 4
#endif
constexpr int kP2Occ = P2_OCC;  // resident waves per SIMD the launch is cut for (and the register budget allows)

// The instruction is v_mfma_f32_32x32x16_f16 with the three split-precision products packed along K: a lane's B operand is
// [hi(c1[pixel][4 h .. 4 h + 3]) | lo 2^11 of the same four channels] — the 8 halves its own split leaves —, A1 = [w_hi 2^11 |
// w_hi] gives hi_w hi_a + hi_w lo_a in ONE instruction, A2 = [w_lo 2^11 | 0] the third product against the same B: two
// instructions of 32 cycles per 32 pixels (three v_mfma_f32_32x32x8_f16, the shape whose K is the 8 channels, cost 96).
template <int V>
struct P2Phase {
  static constexpr int v = V;
};

template <bool WLO>
__global__ __launch_bounds__(256, kP2Occ) void contour_conv2_proj_kernel(Conv2ProjParams p) {
  const int lane = threadIdx.x & 63;
  const int n = lane & 31, h = lane >> 5;
  // XCD-aware order: workgroups go to the 8 XCDs round-robin (blockIdx % 8) and each XCD has its own L2.  Neighbouring
  // strips read 4 columns in common and neighbouring slabs 4 rows: a contiguous run of work items (whole windows) goes to ONE
  // XCD, so what two items share is fetched from HBM once, by one L2, instead of once per XCD that touches it.
#ifdef P2_NO_XCD
  const int lblock = (int)blockIdx.x;
#else
  const int lblock = (gridDim.x % 8 == 0) ? ((int)blockIdx.x % 8) * ((int)gridDim.x / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;
#endif
  const int64_t wg = (int64_t)lblock * 4 + wave_id();  // wave-uniform work item; whole waves only, no barrier
  const int n_groups = (p.n_windows + kP2Group - 1) / kP2Group;
  if (wg >= (int64_t)n_groups * p.n_slabs * kP2Strips) return;
  const int strip = (int)(wg % kP2Strips);
  const int slab = (int)((wg / kP2Strips) % p.n_slabs);
#ifdef P2_FORWARD
  const int group = (int)(wg / ((int64_t)kP2Strips * p.n_slabs));
#else
  // LAST windows first: conv1 wrote c1 in window order, so the windows it wrote last are the ones still in the 256 MB
  // Infinity Cache when this launch starts — reading them first takes them from there before this launch's own traffic
  // evicts them
  const int group = n_groups - 1 - (int)(wg / ((int64_t)kP2Strips * p.n_slabs));
#endif
  const int ta = slab * p.slab_rows;
  const int tb = ta + p.slab_rows < kFrames ? ta + p.slab_rows : kFrames;

  const f16x8 a1 = __builtin_bit_cast(f16x8, p.wfrag[lane]);
  const f16x8 a2 = __builtin_bit_cast(f16x8, WLO ? p.wfrag[64 + lane] : uint4{0u, 0u, 0u, 0u});

  // this lane's pixel column of the two tiles: flat column 60 strip - 2 + 32 tile + n, clamped into the group (a clamped
  // column is never stored and only ever feeds columns that are not stored either).  32-bit offsets from uniform bases.
  uint32_t src_off[2], dst_off[2];
  bool store[2];
#pragma unroll
  for (int tl = 0; tl < 2; ++tl) {
    const int j = 32 * tl + n;
    const int fc = kP2Stride * strip - 2 + j;
    const int fcc = fc < 0 ? 0 : (fc > kP2Cols - 1 ? kP2Cols - 1 : fc);
    const int wl = fcc / kC1Row, pb = fcc - wl * kC1Row;
    const int win = group * kP2Group + wl;
    const int winc = win < p.n_windows ? win : p.n_windows - 1;
    src_off[tl] = (uint32_t)winc * (uint32_t)kC1Win + (uint32_t)(pb * 8 + 4 * h);  // floats: < 2^30 for <= 256 windows a launch
    const bool ok = h == 1 && j >= 2 && j < 2 + kP2Stride && fc == fcc && win < p.n_windows && pb >= kC1Pad && pb < kC1Pad + kFreqC;
    store[tl] = ok;
    dst_off[tl] = ok ? (uint32_t)winc * (uint32_t)kPlaneC + (uint32_t)(pb - kC1Pad) : 0u;
  }
  const int xhalf4 = (lane ^ 32) * 4;
  int sh4[4];  // ds_bpermute addresses of the frequency taps df = 0, 1, 3, 4: lane n + df - 2 of half 1
#pragma unroll
  for (int k = 0; k < 4; ++k) sh4[k] = (32 + ((n + (k < 2 ? k - 2 : k - 1)) & 31)) * 4;
  const float bias_s = p.bias * kLoScale;
  const float sig_k = -1.44269504088896341f * kLoUnscale;
  const float hmask = h ? 1.0f : 0.0f;  // half 0's chain starts at 0, half 1's at half 0's finished sum

  float X0[2][5], X1[2][5], IN[2][5];
#pragma unroll
  for (int tl = 0; tl < 2; ++tl)
#pragma unroll
    for (int df = 0; df < 5; ++df) X0[tl][df] = X1[tl][df] = IN[tl][df] = 0.0f;

  constexpr int kRowFloats = kC1Row * 8;
  auto load = [&](int rho, int tl) {  // rows outside the window are never used: a row of it
    const int rc = rho < 0 ? 0 : (rho > kFrames - 1 ? kFrames - 1 : rho);
    const float* row = p.c1 + (int64_t)rc * kRowFloats;  // wave-uniform base
    return *reinterpret_cast<const float4*>(row + src_off[tl]);
  };
  const int r_first = ta - 2, r_last = tb + 1;
  // Three rows in flight ahead of the one in work, in three register sets whose roles rotate with the row (the loop is
  // written out three times: a rotation by copies would make the copy wait for the load it moves — the first version ran
  // with ONE row in flight per wave and 3,400 cycles per row, 3.8 TB/s).
  float4 buf[3][2];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) buf[k][tl] = load(r_first + k, tl);

  auto step = [&](auto PH, int rho) {
    constexpr int ph = decltype(PH)::v;
    const int t = rho - 2;                // the output row half 1 completes now
    const bool emit = t >= ta && t < tb;  // wave-uniform
    float V[2][5];
#ifdef P2_STREAM  // tools only: the launch's loads and stores without its arithmetic (the access pattern's own pace)
    {
#pragma unroll
      for (int tl = 0; tl < 2; ++tl) {
        const float v = (buf[ph][tl].x + buf[ph][tl].y) + (buf[ph][tl].z + buf[ph][tl].w);
        buf[ph][tl] = load(rho + 3, tl);
#pragma unroll
        for (int df = 0; df < 5; ++df) V[tl][df] = v;
      }
      if (emit) {
        float* orow = p.out + (int64_t)t * kFreqC;
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
          if (store[tl]) orow[dst_off[tl]] = V[tl][2];
      }
      return;
    }
#endif
    if (rho >= 0 && rho < kFrames) {  // wave-uniform
      f16x8 b[2];
#pragma unroll
      for (int tl = 0; tl < 2; ++tl) {
        uint32_t h01, l01, h23, l23;
        split_f16x2(f32x2{buf[ph][tl].x, buf[ph][tl].y}, h01, l01);
        split_f16x2(f32x2{buf[ph][tl].z, buf[ph][tl].w}, h23, l23);
        b[tl] = __builtin_bit_cast(f16x8, uint4{h01, h23, l01, l23});
        buf[ph][tl] = load(rho + 3, tl);  // this set's next row
      }
#pragma unroll
      for (int tl = 0; tl < 2; ++tl) {
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 P = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[tl], zero16, 0, 0, 0);
        if (WLO) P = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b[tl], P, 0, 0, 0);
        // the chain of open output rows (half 0: dt = 0, 1, 2; half 1: dt = 3, 4 behind half 0's finished sum)
#pragma unroll
        for (int df = 0; df < 5; ++df) {
          const float x2 = X1[tl][df] + P[10 + df];
          const float x1 = X0[tl][df] + P[5 + df];
          X0[tl][df] = __builtin_fmaf(IN[tl][df], hmask, P[df]);
          X1[tl][df] = x1;
          V[tl][df] = x1;  // half 1: all five frame taps of output row rho - 2
          IN[tl][df] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(xhalf4, __builtin_bit_cast(int, x2)));
        }
      }
    } else {  // a row outside the window is "same" padding: P = 0
#pragma unroll
      for (int tl = 0; tl < 2; ++tl) {
        buf[ph][tl] = load(rho + 3, tl);
#pragma unroll
        for (int df = 0; df < 5; ++df) {
          const float x2 = X1[tl][df];
          V[tl][df] = X1[tl][df] = X0[tl][df];
          X0[tl][df] = IN[tl][df] * hmask;
          IN[tl][df] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(xhalf4, __builtin_bit_cast(int, x2)));
        }
      }
    }
    if (emit) {
      // out[n] = sum_df V[df][n + df - 2]; the tiles hand over their edge pixels: a source lane offers the other tile's value
      // to the readers that wrap around
      float y[2] = {V[0][2], V[1][2]};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int df = k < 2 ? k : k + 1, s = df - 2;
        const float d0 = s > 0 ? (n < s ? V[1][df] : V[0][df]) : V[0][df];
        const float d1 = s < 0 ? (n >= 32 + s ? V[0][df] : V[1][df]) : V[1][df];
        y[0] += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sh4[k], __builtin_bit_cast(int, d0)));
        y[1] += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sh4[k], __builtin_bit_cast(int, d1)));
      }
      float* orow = p.out + (int64_t)t * kFreqC;  // wave-uniform base
#pragma unroll
      for (int tl = 0; tl < 2; ++tl)
        if (store[tl]) orow[dst_off[tl]] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((y[tl] + bias_s) * sig_k));
    }
  };
  int rho = r_first;
#pragma unroll 1
  for (;;) {
    step(P2Phase<0>{}, rho);
    if (++rho > r_last) break;
    step(P2Phase<1>{}, rho);
    if (++rho > r_last) break;
    step(P2Phase<2>{}, rho);
    if (++rho > r_last) break;
  }
}

void launch_contour_conv2_proj(const float* c1, const void* wfrag, float bias, float* contour, int n_windows, int n_cu,
                               bool weights_have_lo, hipStream_t stream) {
  // sub-batches of 256 windows, each cut into as many frame slabs as keep one launch within the resident waves
  const int64_t slots = (int64_t)n_cu * 4 * kP2Occ;
  const int per_launch = 256;
  for (int w0 = 0; w0 < n_windows; w0 += per_launch) {
    const int n = n_windows - w0 < per_launch ? n_windows - w0 : per_launch;
    const int n_groups = (n + kP2Group - 1) / kP2Group;
    int n_slabs = (int)(slots / ((int64_t)n_groups * kP2Strips));
    n_slabs = n_slabs < 1 ? 1 : (n_slabs > 16 ? 16 : n_slabs);
#ifdef P2_SLABS  // tools only
    n_slabs = P2_SLABS;
#endif
    const int slab_rows = (kFrames + n_slabs - 1) / n_slabs;
    n_slabs = (kFrames + slab_rows - 1) / slab_rows;
    Conv2ProjParams p{c1 + (int64_t)w0 * kC1Win, static_cast<const uint4*>(wfrag), bias, contour + (int64_t)w0 * kPlaneC, n,
                      slab_rows, n_slabs};
    const int64_t waves = (int64_t)n_groups * n_slabs * kP2Strips;
    if (weights_have_lo)
      hipLaunchKernelGGL(contour_conv2_proj_kernel<true>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, p);
    else
      hipLaunchKernelGGL(contour_conv2_proj_kernel<false>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, p);
  }
}

#ifdef BP_AB_KERNELS  // the round-2 vector kernel (BP_CONV2=valu)

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

#endif  // BP_AB_KERNELS

}  // namespace bp
