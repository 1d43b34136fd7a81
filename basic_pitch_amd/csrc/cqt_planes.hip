// CQT front end, round 3: the pyramid lives in HBM as PRE-SPLIT, REFLECT-PADDED f16 planes and the matrix operands of
// both kernels come straight from those planes — no LDS staging, no workgroup barriers, every wave an independent
// worker.  Same operators and reference lines as cqt_mfma.hip (which this file supersedes on the default path):
//   basic_pitch/layers/nnaudio.py:259-284, 636-638   downsampling_by_n: zero-pad 127, 256-tap FIR, stride 2
//   basic_pitch/layers/nnaudio.py:216-256, 640-661   get_cqt_complex per level, * sqrt(lengths), magnitude
//   basic_pitch/layers/nnaudio.py:300-301            ReflectionPad1D(128)
//   basic_pitch/layers/signal.py:171-178             power, 10*log10(power + 1e-10), per-example min / max
//
// Why.  The staged kernels were paced by their instruction count: per (window, level, 16-frame tile) the
// four role waves of a workgroup spent ~1570 wave-instructions around 84 matrix instructions — every sample split into
// f16 hi + lo again in front of every use (2.3 times on average: once for the decimator, ~1.25 times for the
// filterbank's overlapping tiles), an exchange of the re / im planes through LDS, three workgroup barriers.  Here
//   * a sample is split ONCE, where it is produced (level 0: pl_split_kernel; level k >= 1: the decimator's epilogue),
//     and stored as two f16 planes (hi, lo * 2^11) — the same 4 bytes per sample as fp32;
//   * a level's region carries its own reflect padding (128 samples either side, nnaudio.py:300-301), written by the
//     tile that computes the mirrored samples, so a filterbank A fragment — 8 consecutive samples of a frame's 256-tap
//     window — is ONE aligned 16-byte global load per lane (L1 / L2 absorb the Hankel overlap), for every frame;
//   * one wave owns a whole (window, level, tile): all five 16-column groups of the 72 filter columns, re and im of a
//     filter in the SAME lane, so the magnitude / log epilogue runs in registers: no exchange, no barrier.  The filter
//     fragments (58 KB) are the only LDS tenants (read-only, one copy per CU);
//   * the decimator runs transposed (filter = A operand, signal = B operand): a lane ends up with 4 CONSECUTIVE
//     outputs, i.e. one 8-byte store per plane.  The reference zero-pads where the filterbank reflects: the two edge
//     tiles of a level mask their fragments, all others run unmasked.
//
// Arithmetic is unchanged: x = hi + lo 2^-11 (rn), products hi*hi + (lo*hi + hi*lo) 2^-11 on v_mfma_f32_16x16x32_f16,
// fp32 accumulation, taps pre-scaled by 2^10 (decimator) / 2^12 (CQT kernels) — see cqt_mfma.hip's header.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "bp_common.h"

namespace bp {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr float kPlDmTapUnscale = 1.0f / 1024.0f;
constexpr float kPlFmTapUnscale = 1.0f / 4096.0f;
constexpr int kPlPad = 128;        // reflect padding in front of a level's samples (a multiple of 8: units stay aligned)
constexpr int kPlTileOut = 256;    // decimator outputs per tile (16 row-blocks x 16)
constexpr int kPlDmSteps = 9;
constexpr int kPlTilesPerLevel = (kFrames + 15) / 16;  // 11 filterbank tiles of 16 frames

// Geometry of a window's planes.  Element = one f16; a window owns 2 * stride elements: hi plane, then lo plane.  Level
// k's samples live at [off[k] + kPlPad, off[k] + kPlPad + len[k]); regions are multiples of 64 elements (128 bytes).
struct PlGeo {
  int n_levels, hop0, n_bins;
  int len[10];
  int off[10];
  int rlen[10];
  int64_t stride;
};

PlGeo make_pl_geo(bool ext) {
  PlGeo g{};
  g.n_levels = ext ? kOctavesExt : kOctaves;
  g.hop0 = ext ? 512 : 256;
  g.n_bins = ext ? kBinsExt : kBins;
  int64_t off = 0;
  for (int k = 0; k < g.n_levels; ++k) {
    g.len[k] = ext ? (k == 0 ? kAudioNExt : level_len(k - 1)) : level_len(k);
    // readers: the next level's decimator up to len + 767 past the region start + pad; the filterbank's padding frames
    // (172..175 of the 11th tile) up to 176 hop + 256
    const int hop = g.hop0 >> k;
    int need = kPlPad + g.len[k] + 776;
    if (need < 176 * hop + 256) need = 176 * hop + 256;
    g.rlen[k] = (need + 63) & ~63;
    g.off[k] = (int)off;
    off += g.rlen[k];
  }
  g.stride = off;
  return g;
}

int64_t planes_elements_per_window(bool ext) { return 2 * make_pl_geo(ext).stride; }

// Level 0's "edge rows" (fp32, where the level-0 planes used to be: element 0 of the window's hi plane).  Frame f's
// 256-sample window reads samples f hop - 128 .. f hop + 127 with reflection at both ends (nnaudio.py:229, 300-301).  The
// filterbank takes a frame straight from the audio when its taps 16..239 lie inside the signal; row 0 holds frame 0's
// window, rows 1.. those of the frames from pl_edge_frame() to 175 (the 11th tile's padding frames included: finite
// values nobody's result depends on), all with the reflection applied.
__host__ __device__ inline int pl_edge_frame(int L0, int hop0) { return (L0 - 111 + hop0 - 1) / hop0; }  // first f with f hop + 111 >= L0
constexpr int kPlEdgeRowsMax = 1 + 8;

__device__ __forceinline__ void pl_write_edge_rows(const float* __restrict__ x, int L, float* __restrict__ rows, int hop0,
                                                   bool head, int lane) {
  const int f_edge = pl_edge_frame(L, hop0);
  const int r0 = head ? 0 : 1, r1 = head ? 1 : 1 + kPlTilesPerLevel * 16 - f_edge;
  for (int r = r0; r < r1; ++r) {
    const int f = r == 0 ? 0 : f_edge + r - 1;
    for (int j = lane; j < 256; j += 64) {
      int i = f * hop0 + j - kPlPad;
      i = i < 0 ? -i : i;                     // reflection without repeating the edge sample
      i = i >= L ? 2 * (L - 1) - i : i;
      const bool ok = i >= 0 && i < L;        // beyond one reflection: the padding frames' slack
      rows[r * 256 + j] = ok ? x[ok ? i : 0] : 0.0f;
    }
  }
}

// the same as a launch of its own (the per-stage test hook, whose planes come from pl_split_kernel)
__global__ __launch_bounds__(64) void pl_edge_rows_kernel(const float* __restrict__ audio, int64_t audio_stride, int L,
                                                          uint16_t* __restrict__ pl, int64_t stride, int off0, int hop0) {
  const float* x = audio + (int64_t)blockIdx.x * audio_stride;
  float* rows = reinterpret_cast<float*>(pl + (int64_t)blockIdx.x * 2 * stride + off0);
  pl_write_edge_rows(x, L, rows, hop0, true, threadIdx.x);
  pl_write_edge_rows(x, L, rows, hop0, false, threadIdx.x);
}

#define BP_PL_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0)

// value of lane + 4 of the same 16-lane row (DPP row_shl:4; lanes 12..15 of a row read 0)
__device__ __forceinline__ float pl_from_lane_plus4(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x104, 0xf, 0xf, true));
}

__device__ __forceinline__ uint4 pl_load16(const uint16_t* p) {
  uint4 v;
  __builtin_memcpy(&v, p, 16);  // alignment as the pointer has it (2 bytes for the hop-1 level): the compiler picks
  return v;
}

// ================================================================================================
// fp32 signal -> planes of one level (level 0 in production; any level for the per-stage test hook)
__global__ __launch_bounds__(256) void pl_split_kernel(const float* __restrict__ src, int64_t src_stride, int L,
                                                       uint16_t* __restrict__ pl, int64_t stride, int off, int rlen) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (8 * q >= rlen) return;
  const float* x = src + (int64_t)blockIdx.y * src_stride;
  uint16_t* hi = pl + (int64_t)blockIdx.y * 2 * stride + off + 8 * q;
  const int g0 = 8 * q - kPlPad;
  float v[8];
  if (g0 >= 0 && g0 + 8 <= L) {
    const float4 a = *reinterpret_cast<const float4*>(x + g0);  // dword alignment is enough for global vector loads
    const float4 c = *reinterpret_cast<const float4*>(x + g0 + 4);
    v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = c.x, v[5] = c.y, v[6] = c.z, v[7] = c.w;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int g = g0 + e;
      g = g < 0 ? -g : g;                    // nnaudio.py:300-301: reflection without repeating the edge sample
      g = g >= L ? 2 * (L - 1) - g : g;
      const bool ok = g >= 0 && g < L;       // beyond one reflection: slack nobody's result depends on
      v[e] = ok ? x[ok ? g : 0] : 0.0f;
    }
  }
  uint4 h, l;
  split_f16x2_rn(f32x2{v[0], v[1]}, h.x, l.x);
  split_f16x2_rn(f32x2{v[2], v[3]}, h.y, l.y);
  split_f16x2_rn(f32x2{v[4], v[5]}, h.z, l.z);
  split_f16x2_rn(f32x2{v[6], v[7]}, h.w, l.w);
  *reinterpret_cast<uint4*>(hi) = h;
  *reinterpret_cast<uint4*>(hi + stride) = l;
}

// planes -> fp32 (test hook: the pyramid stage's levels as the oracle lays them out)
__global__ __launch_bounds__(256) void pl_unsplit_kernel(const uint16_t* __restrict__ pl, int64_t stride, int off, int L,
                                                         float* __restrict__ dst, int64_t dst_stride) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= L) return;
  const uint16_t* hi = pl + (int64_t)blockIdx.y * 2 * stride + off + kPlPad + i;
  const float h = (float)__builtin_bit_cast(_Float16, hi[0]);
  const float l = (float)__builtin_bit_cast(_Float16, hi[stride]);
  dst[(int64_t)blockIdx.y * dst_stride + i] = h + l * kLoUnscale;
}

// ================================================================================================
// decimate by 2: one wave per tile of 256 outputs
//   D[u][m] = sum_i T[u][i] X[i][m],  T[u][i] = h[i - 2u - 1] (A operand, packed on the host),  X[i][m] = element
//   2 o0 + 32 m + i of the input region (B operand): output n = o0 + 16 m + u reads samples 2n + j - 127, i.e. region
//   elements kPlPad + 2n + j - 127 = 2 o0 + 32 m + (2u + j + 1).
// Row rho of a tile = the 32 elements from 2 o0 + 32 rho; the fragment of lane (m, kg) at k-step s is row m + s, unit kg:
// 24 rows serve all 9 steps.  Re-loading the fragment every step would pull the tile through the texture path 9 times
// (18 KB per tile, and a chain of dependent latencies); instead the 24 rows are fetched ONCE — lane (m, kg) row m, lanes
// m >= 8 also row m + 8 —, parked in a wave-private LDS image, and the fragments are ds_read_b128 from there.
constexpr int kPlRowU = 5;                      // 16-byte units per row in LDS (4 + 1 of skew)
constexpr int kPlRowsU = 2 * 24 * kPlRowU;      // hi rows, then lo rows: 240 units = 3840 bytes per wave

__device__ __forceinline__ uint4 pl_zero_outside(uint4 v, int idx, int L_in) {
  // keep elements whose sample index idx + e - kPlPad lies in [0, L_in): the reference zero-pads the decimator's input
  int nv = kPlPad + L_in - idx;  // elements [0, nv) of this unit are real samples ...
  nv = idx < kPlPad ? 0 : nv;    // ... unless the whole unit lies in front of the signal (kPlPad is a multiple of 8)
  v.x &= (nv > 0 ? 0xffffu : 0u) | (nv > 1 ? 0xffff0000u : 0u);
  v.y &= (nv > 2 ? 0xffffu : 0u) | (nv > 3 ? 0xffff0000u : 0u);
  v.z &= (nv > 4 ? 0xffffu : 0u) | (nv > 5 ? 0xffff0000u : 0u);
  v.w &= (nv > 6 ? 0xffffu : 0u) | (nv > 7 ? 0xffff0000u : 0u);
  return v;
}

// One value split like split_f16x2_rn (hi and lo rounded to nearest), for the few scalar pad writes.
__device__ __forceinline__ void pl_split1(float v, uint16_t& h, uint16_t& l) {
  uint32_t h2, l2;
  split_f16x2_rn(f32x2{v, 0.0f}, h2, l2);
  h = (uint16_t)(h2 & 0xffffu);
  l = (uint16_t)(l2 & 0xffffu);
}

__device__ __forceinline__ bool pl_tile_is_edge(int tile, int L_in) {
  return tile == 0 || 2 * kPlTileOut * tile + 32 * 23 + 32 > kPlPad + L_in;  // some fragment reaches outside the signal
}

// What a lane fetches for a tile: its row m, and (lanes m >= 8) row m + 8.  F32IN: 8 fp32 samples per row (the level-0
// signal itself, zero outside [0, L): nnaudio.py:269-279); else 8 f16 hi + 8 f16 lo from the input level's planes.
template <bool F32IN>
struct PlRaw;
template <>
struct PlRaw<true> {
  float4 a0, a1, b0, b1;
};
template <>
struct PlRaw<false> {
  uint4 ah, al, bh, bl;
};

template <bool F32IN>
__device__ __forceinline__ PlRaw<F32IN> pl_fetch_rows(const float* __restrict__ x, const uint16_t* __restrict__ in_hi,
                                                       int64_t stride, int L_in, int tile, int lane) {
  const int m = lane & 15, kg = lane >> 4;
  const int base = 2 * kPlTileOut * tile + 32 * m + 8 * kg;
  PlRaw<F32IN> r;
#if defined(PL_ABLATE) && (PL_ABLATE & 4)  // tools only: no input loads (timing)
  if constexpr (F32IN) {
    r.a0 = r.a1 = r.b0 = r.b1 = float4{1.f * tile, 2.f, 3.f * lane, 4.f};
    return r;
  }
#endif
  if constexpr (F32IN) {
    auto row = [&](int g0, float4& lo4, float4& hi4) {
      if (!pl_tile_is_edge(tile, L_in)) {
        lo4 = *reinterpret_cast<const float4*>(x + g0);  // dword alignment is enough for global vector loads
        hi4 = *reinterpret_cast<const float4*>(x + g0 + 4);
      } else {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int g = g0 + e;
          const bool ok = g >= 0 && g < L_in;
          v[e] = ok ? x[ok ? g : 0] : 0.0f;
        }
        lo4 = float4{v[0], v[1], v[2], v[3]};
        hi4 = float4{v[4], v[5], v[6], v[7]};
      }
    };
    row(base - kPlPad, r.a0, r.a1);
    r.b0 = r.b1 = float4{0.f, 0.f, 0.f, 0.f};
    if (m >= 8) row(base - kPlPad + 256, r.b0, r.b1);
  } else {
    r.ah = *reinterpret_cast<const uint4*>(in_hi + base);
    r.al = *reinterpret_cast<const uint4*>(in_hi + stride + base);
    r.bh = r.bl = uint4{0u, 0u, 0u, 0u};
    if (m >= 8) {
      r.bh = *reinterpret_cast<const uint4*>(in_hi + base + 256);
      r.bl = *reinterpret_cast<const uint4*>(in_hi + stride + base + 256);
    }
  }
  return r;
}

__device__ __forceinline__ void pl_split8(const float4& a, const float4& c, uint4& h, uint4& l) {
  split_f16x2_rn(f32x2{a.x, a.y}, h.x, l.x);
  split_f16x2_rn(f32x2{a.z, a.w}, h.y, l.y);
  split_f16x2_rn(f32x2{c.x, c.y}, h.z, l.z);
  split_f16x2_rn(f32x2{c.z, c.w}, h.w, l.w);
}

// F32IN = false: the input level is a plane region (`in_hi`, lo plane `stride` elements behind it).
// F32IN = true:  the input is the fp32 signal `x` (level 0): rows are split in registers, and the tile also WRITES the
//                level-0 planes (`in_hi` is then the level-0 region to fill): its sixteen own rows are exactly the 512
//                elements [2 o0, 2 o0 + 512); tile 0 / the last tile add the reflect padding of level 0.
// MIRROR: the outputs also go to a second image of the output level (`mir_hi` = its sample 0, lo plane `mir_stride`
//         elements behind; no padding there): the per-window kernel keeps the levels it reads again in LDS.
// PF:     k-steps of LDS fragment reads kept ahead of the matrix instructions (0 = the compiler's own order, which reads
//         each fragment right in front of its use and waits: fine where four waves per SIMD and a prefetched next item
//         cover it, 1.5 k cycles per tile where a tile's latency is the critical path — the per-window kernel).
template <bool F32IN, bool MIRROR = false, int PF = 0>
__device__ __forceinline__ void pl_dec_tile(const PlRaw<F32IN>& raw, const float* __restrict__ x,
                                            uint16_t* __restrict__ in_hi, int64_t stride, int L_in,
                                            uint16_t* __restrict__ out_hi, int L_out, int tile, int n_tiles,
                                            const uint4 (&th)[kPlDmSteps], const uint4* __restrict__ tlo,
                                            uint4* __restrict__ rows, int lane, uint16_t* mir_hi = nullptr,
                                            int mir_stride = 0, int hop0 = 0) {
  // keep the lo fragments in LDS: without an opaque offset the compiler hoists the 9 item-invariant reads into registers
  asm volatile("" : "+v"(lane));
  const int m = lane & 15, kg = lane >> 4;
  const int o0 = kPlTileOut * tile;
  const int base = 2 * o0 + 32 * m + 8 * kg;  // region element of this lane's own row
  uint4 ah, al, bh, bl;
  if constexpr (F32IN) {
    pl_split8(raw.a0, raw.a1, ah, al);
    pl_split8(raw.b0, raw.b1, bh, bl);
  } else {
    ah = raw.ah, al = raw.al, bh = raw.bh, bl = raw.bl;
    if (pl_tile_is_edge(tile, L_in)) {  // the planes carry reflect padding where the decimator wants zeros
      ah = pl_zero_outside(ah, base, L_in);
      al = pl_zero_outside(al, base, L_in);
      bh = pl_zero_outside(bh, base + 256, L_in);
      bl = pl_zero_outside(bl, base + 256, L_in);
    }
  }
  uint4* rh = rows + m * kPlRowU + kg;
  uint4* rl = rh + 24 * kPlRowU;
  rh[0] = ah;
  rl[0] = al;
  if (m >= 8) {
    rh[8 * kPlRowU] = bh;
    rl[8 * kPlRowU] = bl;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  if constexpr (F32IN) {
    // level 0 has no planes since round 4 (the filterbank splits its level-0 samples itself, straight from the fp32
    // audio); what it cannot read from the audio are the windows that are mirrored at the ends of the signal: the first
    // tile of a window leaves frame 0's reflect-padded window, the last tile those of the frames from `edge_frame` on, as
    // fp32 rows where the level-0 planes used to be
    if (tile == 0 || tile == n_tiles - 1) pl_write_edge_rows(x, L_in, reinterpret_cast<float*>(in_hi), hop0, tile == 0, lane);
  }

  f32x4 hh = {0.f, 0.f, 0.f, 0.f}, xx = hh;
  const uint4* tl = tlo + lane;
#if defined(PL_ABLATE) && (PL_ABLATE & 1)  // tools only: no matrix work (timing; garbage results)
  hh[0] = __builtin_bit_cast(float, rh[0].x);
#else
  if constexpr (PF == 0) {
#pragma unroll
    for (int s = 0; s < kPlDmSteps; ++s) {
      const uint4 xh = rh[s * kPlRowU], xl = rl[s * kPlRowU];
      const uint4 tls = tl[s * 64];
      hh = BP_PL_MFMA16(th[s], xh, hh);
      xx = BP_PL_MFMA16(tls, xh, xx);
      xx = BP_PL_MFMA16(th[s], xl, xx);
    }
  } else {
    uint4 xh[PF + 1], xl[PF + 1], tls[PF + 1];
    auto rd = [&](int s) {
      xh[s % (PF + 1)] = rh[s * kPlRowU];
      xl[s % (PF + 1)] = rl[s * kPlRowU];
      tls[s % (PF + 1)] = tl[s * 64];
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) rd(s);
#pragma unroll
    for (int s = 0; s < kPlDmSteps; ++s) {
      if (s + PF < kPlDmSteps) rd(s + PF);
      __builtin_amdgcn_sched_barrier(0);
      hh = BP_PL_MFMA16(th[s], xh[s % (PF + 1)], hh);
      xx = BP_PL_MFMA16(tls[s % (PF + 1)], xh[s % (PF + 1)], xx);
      xx = BP_PL_MFMA16(th[s], xl[s % (PF + 1)], xx);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#endif
  // the rows are read: the next tile of this wave may overwrite them
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // D: column m = lane & 15, rows u = 4 kg + r: four consecutive outputs
  const int n0 = o0 + 16 * m + 4 * kg;
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = (hh[r] + xx[r] * kLoUnscale) * kPlDmTapUnscale;
  uint2 h2, l2;
  split_f16x2_rn(f32x2{v[0], v[1]}, h2.x, l2.x);
  split_f16x2_rn(f32x2{v[2], v[3]}, h2.y, l2.y);
  uint16_t* oh = out_hi + kPlPad + n0;
#if defined(PL_ABLATE) && (PL_ABLATE & 2)  // tools only: no output stores unless impossible (timing)
  if (n0 + 3 < L_out && h2.x == 0x12345678u) {
#else
  if (n0 + 3 < L_out) {
#endif
    *reinterpret_cast<uint2*>(oh) = h2;
    *reinterpret_cast<uint2*>(oh + stride) = l2;
  }
  if constexpr (MIRROR) {
    if (mir_hi != nullptr) {  // wave-uniform
      uint16_t* mh = mir_hi + n0;
      if (n0 + 3 < L_out) {
        *reinterpret_cast<uint2*>(mh) = h2;
        *reinterpret_cast<uint2*>(mh + mir_stride) = l2;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n0 + r < L_out) {
            mh[r] = (uint16_t)((r < 2 ? h2.x : h2.y) >> (16 * (r & 1)));
            mh[mir_stride + r] = (uint16_t)((r < 2 ? l2.x : l2.y) >> (16 * (r & 1)));
          }
      }
    }
  }
  // the level's last (partial) group of four, and the reflect padding: the tiles that hold samples 1..128 and
  // L-129..L-2 write their mirror images (nnaudio.py:300-301).  Wave-uniform conditions, 16-bit stores.
  const bool tail = o0 + kPlTileOut > L_out - 130 || tile == 0;
  if (tail) {
    const uint16_t eh[4] = {(uint16_t)(h2.x & 0xffffu), (uint16_t)(h2.x >> 16), (uint16_t)(h2.y & 0xffffu), (uint16_t)(h2.y >> 16)};
    const uint16_t el[4] = {(uint16_t)(l2.x & 0xffffu), (uint16_t)(l2.x >> 16), (uint16_t)(l2.y & 0xffffu), (uint16_t)(l2.y >> 16)};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + r;
      if (n >= L_out) continue;
      if (n0 + 3 >= L_out) {
        oh[r] = eh[r];
        oh[stride + r] = el[r];
      }
      if (n >= 1 && n <= kPlPad) {
        out_hi[kPlPad - n] = eh[r];
        out_hi[stride + kPlPad - n] = el[r];
      }
      if (n <= L_out - 2 && n >= L_out - 1 - kPlPad) {
        const int q = kPlPad + 2 * (L_out - 1) - n;
        out_hi[q] = eh[r];
        out_hi[stride + q] = el[r];
      }
    }
  }
}

// The filter's hi fragments live in registers (36 VGPRs), its lo fragments in LDS (9 KB, one ds_read_b128 per k-step):
// with all 18 in registers the kernels would hold 3 waves per SIMD instead of 4.
__device__ __forceinline__ void pl_load_tfrag(const uint4* __restrict__ tfrag, uint4 (&th)[kPlDmSteps], uint4* tlo) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int s = 0; s < kPlDmSteps; ++s) th[s] = tfrag[s * 64 + lane];
  for (int i = threadIdx.x; i < kPlDmSteps * 64; i += blockDim.x) tlo[i] = tfrag[kPlDmSteps * 64 + i];
  __syncthreads();
}

// One level of every window.  F32IN: level 0 -> 1 from the fp32 audio (splits the signal once, writes the level-0 planes
// with their reflect padding and the level-1 planes); else planes -> planes.  Waves walk (window, tile) items, the rows
// of the next item are fetched before the matrix work of the current one; no workgroup barrier after the fragments are in.
template <bool F32IN>
__global__ __launch_bounds__(256, 4) void pl_decimate_kernel(const float* __restrict__ audio, int64_t audio_stride,
                                                             uint16_t* __restrict__ pl, int64_t stride, int off_in, int L_in,
                                                             int off_out, int L_out, int tiles,
                                                             const uint4* __restrict__ tfrag, int n_windows,
                                                             int hop0) {
  __shared__ __attribute__((aligned(16))) uint4 tlo[kPlDmSteps * 64];
  __shared__ __attribute__((aligned(16))) uint4 rows_all[4 * kPlRowsU];
  uint4 th[kPlDmSteps];
  pl_load_tfrag(tfrag, th, tlo);
  const int lane = threadIdx.x & 63;
  uint4* rows = rows_all + wave_id() * kPlRowsU;
  const int n_items = n_windows * tiles;
  // items of this workgroup: blockIdx.x + gridDim.x * j, drawn from a counter in LDS (the older waves of a workgroup win
  // the issue arbitration and would otherwise run out of work long before the younger ones: see the filterbank below)
  __shared__ int s_next;
  if (threadIdx.x == 0) s_next = 0;
  __syncthreads();
  auto grab = [&]() -> int {
    int j = 0;
    if ((threadIdx.x & 63) == 0) j = atomicAdd(&s_next, 1);
    const int it = blockIdx.x + gridDim.x * __builtin_amdgcn_readfirstlane(j);
    return it < n_items ? it : -1;
  };
  // rows of TWO items ahead in flight (round 4: with one, a wave had ~3 KB outstanding and the launch sat at the
  // latency-bandwidth product of 16 waves per CU, not at the HBM rate — dropping the level-0 plane writes did not move it)
  auto fetch = [&](int it) {
    const int fb = it / tiles, ft = it - fb * tiles;
    return pl_fetch_rows<F32IN>(audio + (int64_t)fb * audio_stride, pl + (int64_t)fb * 2 * stride + off_in, stride, L_in, ft,
                                lane);
  };
  int item = grab();
  if (item < 0) return;
  int item1 = grab();
  int item2 = item1 >= 0 ? grab() : -1;
  PlRaw<F32IN> raw = fetch(item);
  PlRaw<F32IN> raw1 = fetch(item1 >= 0 ? item1 : item);
  for (;;) {
    const int item3 = item2 >= 0 ? grab() : -1;
    const PlRaw<F32IN> raw2 = fetch(item2 >= 0 ? item2 : item);
    const int b = item / tiles, tile = item - b * tiles;
    uint16_t* w = pl + (int64_t)b * 2 * stride;
    pl_dec_tile<F32IN>(raw, audio + (int64_t)b * audio_stride, w + off_in, stride, L_in, w + off_out, L_out, tile, tiles, th,
                       tlo, rows, lane, nullptr, 0, hop0);
    if (item1 < 0) break;
    raw = raw1, raw1 = raw2, item = item1, item1 = item2, item2 = item3;
  }
}

// The deeper levels of one window, one workgroup of 16 waves: a level is a few dozen tiles at most and depends on the one
// above it, so as separate launches each would pay a launch boundary and a ramp for a few microseconds of work.  The
// levels this workgroup reads again stay in LDS (hi and lo plane of levels `lds_first` .. last - 1, samples only: the
// decimator zero-pads, the two edge tiles of a level mask whatever lies beside the samples), so the seven dependent
// steps pay an LDS round trip and a barrier each, not a store-acknowledge + L2 read (28 us for 3 us of matrix work);
// every level is also written to its planes in HBM for the filterbank.
struct PlTail {
  int first, last, lds_first;
};
constexpr int kPlTailThreads = 1024;
constexpr int kPlTailGuard = kPlPad;     // elements in front of the first resident level: tile 0 reads (and masks) them
constexpr int kPlTailLdsElems = 21760;   // per plane: guard + levels 2..7 of the 22.05 kHz pyramid (each rounded up to 8)

__host__ __device__ inline int pl_tail_lds_need(const PlGeo& g, int lds_first, int last) {
  int n = kPlTailGuard;
  for (int k = lds_first; k < last; ++k) n += (g.len[k] + 7) & ~7;
  return n;
}

__global__ __launch_bounds__(kPlTailThreads) void pl_decimate_tail_kernel(const float* __restrict__ audio, int64_t audio_stride,
                                                                       uint16_t* __restrict__ pl, PlGeo g, PlTail t,
                                                                       const uint4* __restrict__ tfrag) {
  __shared__ __attribute__((aligned(16))) uint4 tlo[kPlDmSteps * 64];
  __shared__ __attribute__((aligned(16))) uint4 rows_all[(kPlTailThreads / 64) * kPlRowsU];
  __shared__ __attribute__((aligned(16))) uint16_t s_pl[2 * kPlTailLdsElems];  // hi plane, lo plane
  uint4 th[kPlDmSteps];
  pl_load_tfrag(tfrag, th, tlo);
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  uint4* rows = rows_all + wave * kPlRowsU;
  uint16_t* w = pl + (int64_t)blockIdx.x * 2 * g.stride;
  int loff_in = 0, loff_out = kPlTailGuard;  // LDS element of sample 0 of the input / output level (when resident)
  for (int k = t.first; k <= t.last; ++k) {
    const int tiles = (g.len[k] + kPlTileOut - 1) / kPlTileOut;
    const bool in_lds = k - 1 >= t.lds_first, out_lds = k >= t.lds_first && k < t.last;  // workgroup-uniform
    uint16_t* mir = out_lds ? s_pl + loff_out : nullptr;
    if (k == 1) {
      // level 0 -> 1 straight from the fp32 audio (round 4: the wide launch that did this spent 18 of its 35 us on per-item
      // bookkeeping — queue draws, 64-bit addresses, edge predicates — with nothing to do: ablation in DESIGN.md §7).  Here a
      // wave walks tiles wave, wave + 16, ... of its workgroup's window, the next tile's rows in flight during the current
      // one's matrix work; level 1 goes to its planes in HBM (level 2 reads it back through L2 after the barrier below).
      const float* x = audio + (int64_t)blockIdx.x * audio_stride;
      int tile = wave;
      if (tile < tiles) {
        PlRaw<true> raw = pl_fetch_rows<true>(x, nullptr, 0, g.len[0], tile, lane);
        for (;;) {
          const int ntile = tile + kPlTailThreads / 64;
          const PlRaw<true> nraw = pl_fetch_rows<true>(x, nullptr, 0, g.len[0], ntile < tiles ? ntile : tile, lane);
          pl_dec_tile<true, false, 1>(raw, x, w + g.off[0], g.stride, g.len[0], w + g.off[1], g.len[1], tile, tiles, th, tlo,
                                      rows, lane, nullptr, 0, g.hop0);
          if (ntile >= tiles) break;
          raw = nraw, tile = ntile;
        }
      }
    } else
    for (int tile = wave; tile < tiles; tile += kPlTailThreads / 64) {
      PlRaw<false> raw;
      if (in_lds)
        raw = pl_fetch_rows<false>(nullptr, s_pl + loff_in - kPlPad, kPlTailLdsElems, g.len[k - 1], tile, lane);
      else
        raw = pl_fetch_rows<false>(nullptr, w + g.off[k - 1], g.stride, g.len[k - 1], tile, lane);
      pl_dec_tile<false, true, 3>(raw, nullptr, w + g.off[k - 1], g.stride, g.len[k - 1], w + g.off[k], g.len[k], tile, tiles,
                               th, tlo, rows, lane, mir, kPlTailLdsElems);
    }
    __syncthreads();  // level k is complete: in LDS for this workgroup (and in L2: same CU, same L1, workgroup scope)
    if (out_lds) {
      loff_in = loff_out;
      loff_out += (g.len[k] + 7) & ~7;
    }
  }
}

// ================================================================================================
// filterbank: one wave per (window, level, 16-frame tile), all 72 filter columns, epilogue in registers
//   column groups (16 columns each): 0 = re of filters 0..15, 1 = im 0..15 (taps 16..239: k-steps 0..6),
//   2 = re 16..31, 3 = im 16..31, 4 = {re 32..35 | im 32..35 | 8 zero columns} (taps 48..207: k-steps 1..5)
constexpr int kPlFbFrags = 7 + 7 + 5 + 5 + 5;  // step-fragments, hi and lo each
__host__ __device__ constexpr int pl_fb_frag0(int g) { return g == 0 ? 0 : g == 1 ? 7 : g == 2 ? 14 : g == 3 ? 19 : 24; }
__host__ __device__ constexpr int pl_fb_step0(int g) { return g < 2 ? 0 : 1; }
__host__ __device__ constexpr int pl_fb_steps(int g) { return g < 2 ? 7 : 5; }
// the 29 (k-step, group) products of a task in issue order: k-step major, so an A fragment is finished with after its step
struct PlFbItem {
  int s, q, f;  // k-step, column group, index of the group's step-fragment in LDS
};
__host__ __device__ constexpr PlFbItem pl_fb_item(int i) {
  int n = 0;
  for (int s = 0; s < 7; ++s)
    for (int q = 0; q < 5; ++q) {
      if (s < pl_fb_step0(q) || s >= pl_fb_step0(q) + pl_fb_steps(q)) continue;
      if (n == i) return PlFbItem{s, q, pl_fb_frag0(q) + s - pl_fb_step0(q)};
      ++n;
    }
  return PlFbItem{-1, -1, -1};
}
static_assert(pl_fb_item(kPlFbFrags - 1).s == 6 && pl_fb_item(kPlFbFrags).s == -1, "29 products per task");

// THREADS / APF: 1024 threads = 4 waves per SIMD (128 VGPRs each) keep three k-steps of A fragments ahead (the default:
// 54 us at B = 256); 768 / 704 threads = 3 waves per SIMD with up to 168 VGPRs hold the whole next task's fragments in
// flight (56 registers) — measured the same 55 us: once the waves draw their tasks from a queue the kernel is paced by
// instruction issue (matrix pipe 41 % busy, VALU most of the rest), not by memory latency.
// FUSED: a workgroup owns whole windows (its waves draw the window's 99 tasks), keeps the tiles' extrema in LDS and, when
// the window's last task is done, normalises its log-power map itself and writes the pre-split, BatchNorm-ed `zp` words
// (signal.py:177-183, models.py:187-189: what zpack_kernel does in a launch of its own) — the map was written by this
// CU a few microseconds ago and comes back from L2, the extrema never leave the CU.  !FUSED: tasks strided over all
// workgroups, extrema partials to `mmp` (launches with fewer windows than CUs, and the per-stage test hook).
template <int THREADS, int APF, bool FUSED>
__global__ __launch_bounds__(THREADS) void cqt_filterbank_planes_kernel(
    const uint16_t* __restrict__ pl, const float* __restrict__ audio, int64_t audio_stride,
    const uint4* __restrict__ bfrag, const float* __restrict__ sqrt_len, float* __restrict__ lp, float2* __restrict__ mmp,
    uint32_t* __restrict__ zp, int n_windows, LogConsts kc, PlGeo g, unsigned per_window_magic) {
  __shared__ __attribute__((aligned(16))) uint4 bfr[kPlFbFrags * 2 * 64];
  // sqrt(lengths) and the level offsets from LDS, not from global / constant memory: a wave's memory counters are in
  // order, so a global load in the epilogue would wait for every A fragment prefetched for the next task before it
  __shared__ float s_sqrt_len[kBinsExt];
  __shared__ int s_off[10];
  __shared__ int s_next;
  __shared__ float2 s_mm[FUSED ? 10 * kPlTilesPerLevel : 1];
  // FUSED: the log-power values of the four top levels (144 bins x 172 frames = 97 KB: what is left of the CU's LDS) wait
  // for the normalise phase here instead of making the round trip through L2 / HBM
  constexpr int kLdsBins = 4 * kBpo;
  __shared__ float s_lp[FUSED ? kFrames * kLdsBins : 1];
#if defined(PL_FB_PROF)
  const unsigned long long pentry = __builtin_amdgcn_s_memtime();
#endif
  if (threadIdx.x == 0) s_next = 0;
  for (int i = threadIdx.x; i < kPlFbFrags * 2 * 64; i += THREADS) bfr[i] = bfrag[i];
  for (int i = threadIdx.x; i < g.n_bins; i += THREADS) s_sqrt_len[i] = sqrt_len[i];
  if (threadIdx.x < 10) s_off[threadIdx.x] = g.off[threadIdx.x];
  __syncthreads();
  int lane = threadIdx.x & 63;
  const int per_window = g.n_levels * kPlTilesPerLevel;
  const int n_tasks = n_windows * per_window;
  const float kln2 = 0.69314718055994531f * kc.s0 * kc.s1;  // log2 -> 10 log10
  // Tasks of this workgroup: blockIdx.x + gridDim.x * j, j = 0, 1, ...; its waves DRAW j from a counter in LDS instead of
  // owning a fixed share: the SIMD's issue arbitration favours the older waves of a workgroup (phase clocks: wave 0
  // finishes a task in 7.5 k cycles, wave 10 in 19.5 k), so with fixed shares the old waves ran out of work at 40 % of
  // the kernel's duration and the young ones finished it alone.
  struct Pos {
    int b, rem;
  };
  int win = blockIdx.x;  // FUSED: the window this workgroup is working on
  auto grab = [&]() -> int {
    int j = 0;
    if ((threadIdx.x & 63) == 0) j = atomicAdd(&s_next, 1);
    j = __builtin_amdgcn_readfirstlane(j);
    if constexpr (FUSED) return j < per_window ? win * per_window + j : -1;
    const int task_ = blockIdx.x + gridDim.x * j;
    return task_ < n_tasks ? task_ : -1;
  };
  auto pos_of = [&](int task_) {  // task / per_window by multiply-shift (exact below 2^32 / 95 for 99, 2^32 / 4 for 110)
    if constexpr (FUSED) return Pos{win, task_ - win * per_window};
    const int b_ = (int)__umulhi((unsigned)task_, per_window_magic);
    return Pos{b_, task_ - b_ * per_window};
  };
  // Where a task's A fragments come from.  Planes: 16 bytes of the hi plane per k-step (32 elements apart), the lo
  // plane g.stride elements behind it.  Level 0 (round 4): the fp32 audio itself — 8 samples = two 16-byte loads per
  // k-step, split to hi / lo in registers when the k-step is consumed (every level-0 sample feeds at most one frame: hop
  // >= window, so nothing is split twice); level 0 has no planes any more: -45 MB written and -45 MB read per 256 windows.
  struct Src {
    const char* p;     // this lane's first fragment
    int step;          // bytes between k-steps
    int64_t second;    // bytes from the first to the second 16-byte half (lo plane / samples 4..7)
    bool raw;          // fp32 samples: split at consumption
  };
  const int f_edge = pl_edge_frame(g.len[0], g.hop0);
  auto src_of = [&](Pos p) -> Src {
    const int level_ = (p.rem * 745) >> 13;  // rem / 11 for rem < 2700
    const int tile_ = p.rem - level_ * kPlTilesPerLevel;
    if (audio != nullptr && level_ == 0) {  // wave-uniform
      const int f = 16 * tile_ + (lane & 15);
      const float* a = audio + (int64_t)p.b * audio_stride + f * g.hop0 - kPlPad;
      if (f == 0 || f >= f_edge)  // a mirrored window: the edge rows the decimator left where level 0's planes were
        a = reinterpret_cast<const float*>(pl + (int64_t)p.b * 2 * g.stride + s_off[0]) + 256 * (f == 0 ? 0 : 1 + f - f_edge);
      return Src{reinterpret_cast<const char*>(a + 16 + 8 * (lane >> 4)), 128, 16, true};
    }
    const uint16_t* q = pl + (int64_t)p.b * 2 * g.stride + s_off[level_] + (16 * tile_ + (lane & 15)) * (g.hop0 >> level_) + 16 +
                        8 * (lane >> 4);
    return Src{reinterpret_cast<const char*>(q), 64, 2 * g.stride, false};
  };
  auto load16 = [](const char* p) {
    uint4 v;
    __builtin_memcpy(&v, __builtin_assume_aligned(p, 2), 16);  // 2-byte alignment at the hop-1 level; dword at least elsewhere
    return v;
  };
  static_assert(kPlTilesPerLevel == 11, "the multiply-shift above divides by 11");
  for (; win < (FUSED ? n_windows : blockIdx.x + 1); win += gridDim.x) {
  int task = grab();
  int ntask = task >= 0 ? grab() : -1;
  if (task >= 0) {
  Pos pos = pos_of(task);
  // A fragments: a ring of 7 k-steps.  When a task starts, its first APF steps are in the ring (fetched during the task
  // before); step s + APF is fetched when step s has been consumed — for s + APF >= 7 that is step s + APF - 7 of the NEXT
  // task.  APF = 7: every load has a whole task's matrix work to land (a level-0 / level-1 task streams from HBM).
  uint4 ah[7], al[7];
  Src src = src_of(pos);
  {
#pragma unroll
    for (int s = 0; s < APF; ++s) {
      ah[s] = load16(src.p + src.step * s);
      al[s] = load16(src.p + src.second + src.step * s);
    }
  }
#if defined(PL_FB_PROF)
  unsigned long long pk = 0, pe = 0, pn_ = 0, pstart = __builtin_amdgcn_s_memtime();
#endif
  for (;;) {
#if defined(PL_FB_PROF)
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
#endif
    // keep the filter fragments in LDS: without an opaque offset the compiler hoists all 58 loop-invariant reads
    asm volatile("" : "+v"(lane));
    const int nntask = ntask >= 0 ? grab() : -1;  // drawn a task ahead: its LDS round trip is nobody's critical path
    const int b = pos.b, rem = pos.rem;
    const int level = (rem * 745) >> 13, tile = rem - level * kPlTilesPerLevel;
    const int t = lane & 15, kg = lane >> 4;
    const uint4* bl = bfr + lane;
    const bool more = ntask >= 0;
    const Pos npos = more ? pos_of(ntask) : pos;
    const Src nsrc = more ? src_of(npos) : src;

    f32x4 hh[5], xx[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) hh[q] = xx[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    // filter fragments two products ahead of the matrix instructions that use them: issued right behind a product's
    // instructions, an LDS read's ~100 cycles would be exposed 29 times per task (they were: the compiler's own order)
    constexpr int kBPf = 2;
    uint4 bh[kPlFbFrags], bw[kPlFbFrags];
#pragma unroll
    for (int i = 0; i < kBPf; ++i) {
      bh[i] = bl[(2 * pl_fb_item(i).f) * 64];
      bw[i] = bl[(2 * pl_fb_item(i).f + 1) * 64];
    }
#pragma unroll
    for (int i = 0; i < kPlFbFrags; ++i) {
      constexpr auto item = [](int j) { return pl_fb_item(j); };
      const int s = item(i).s, q = item(i).q;
      if (i + kBPf < kPlFbFrags) {
        bh[i + kBPf] = bl[(2 * item(i + kBPf).f) * 64];
        bw[i + kBPf] = bl[(2 * item(i + kBPf).f + 1) * 64];
      }
      if (src.raw && (i == 0 || item(i - 1).s != s)) {  // first product of k-step s of a level-0 task (wave-uniform): the
        const float4 a = __builtin_bit_cast(float4, ah[s]), c = __builtin_bit_cast(float4, al[s]);  // slot holds 8 samples
        pl_split8(a, c, ah[s], al[s]);
      }
      __builtin_amdgcn_sched_barrier(0);
      hh[q] = BP_PL_MFMA16(ah[s], bh[i], hh[q]);
      xx[q] = BP_PL_MFMA16(al[s], bh[i], xx[q]);
      xx[q] = BP_PL_MFMA16(ah[s], bw[i], xx[q]);
      if (i + 1 == kPlFbFrags || item(i + 1).s != s) {  // last product of k-step s: its ring slot takes step s + APF
        const int sn = s + APF;
        if (sn < 7) {
          ah[sn] = load16(src.p + src.step * sn);
          al[sn] = load16(src.p + src.second + src.step * sn);
        } else {
          ah[sn - 7] = load16(nsrc.p + nsrc.step * (sn - 7));
          al[sn - 7] = load16(nsrc.p + nsrc.second + nsrc.step * (sn - 7));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }

#if defined(PL_FB_PROF)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(hh[0]), "+v"(xx[4]));
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
#endif
    // epilogue: D row (frame) = 4 kg + r, column (filter of the group) = lane & 15
    const int bin0 = (g.n_levels - 1 - level) * kBpo - 15;  // nnaudio.py:640-642
    float* lp_t = lp + ((int64_t)b * kFrames + 16 * tile + 4 * kg) * g.n_bins + bin0;
    const int fr0 = 16 * tile + 4 * kg;
    float vmin = __int_as_float(0x7f800000), vmax = -__int_as_float(0x7f800000);
    // `masked` (wave-uniform): the tile has padding frames (the 11th tile) or the level has bins below the CQT's first
    // (the deepest level): 8 of 10 tasks have neither, and then only group 4's unused columns need a predicate
    const bool masked = tile == kPlTilesPerLevel - 1 || bin0 < 0;
    const bool to_lds = FUSED && level < 4;  // wave-uniform
    float* lds_t = s_lp + (16 * tile + 4 * kg) * kLdsBins + (3 - level) * kBpo;
    auto finish = [&](auto masked_c, auto lds_c, const f32x4& hr, const f32x4& xr, const f32x4& hi_, const f32x4& xi, int k,
                      bool col_ok) {
      constexpr bool kMasked = decltype(masked_c)::value, kLds = decltype(lds_c)::value;
      const bool bin_ok = col_ok && (!kMasked || bin0 + k >= 0);
      // * sqrt(lengths) (nnaudio.py:650, before squaring) and the taps' 2^-12 in one factor: a power of two commutes
      // with the rounding of the product
      const float slk = s_sqrt_len[bin_ok ? bin0 + k : 0] * kPlFmTapUnscale;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float re = __fmul_rn(hr[r] + xr[r] * kLoUnscale, slk);
        const float im = __fmul_rn(hi_[r] + xi[r] * kLoUnscale, slk);
        // nnaudio.py:661 magnitude, signal.py:174-175 power and 10 log10 on the hardware's 1-ulp sqrt / log2
        const float mag = __builtin_amdgcn_sqrtf(__fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im)));
        const float pw = __fmul_rn(mag, mag);
        v[r] = __fmul_rn(__builtin_amdgcn_logf(__fadd_rn(pw, kc.eps)), kln2);
      }
      auto put = [&](int r) {
        if constexpr (kLds)
          lds_t[r * kLdsBins + k] = v[r];
        else
          lp_t[r * g.n_bins + k] = v[r];
      };
      if constexpr (kMasked) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (bin_ok && fr0 + r < kFrames) {
            put(r);
            vmin = fminf(vmin, v[r]);
            vmax = fmaxf(vmax, v[r]);
          }
      } else if (bin_ok) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          put(r);
          vmin = fminf(vmin, v[r]);
          vmax = fmaxf(vmax, v[r]);
        }
      }
    };
    // group 4: re of filter 32 + c in column c, im in column 4 + c: bring the im values over (row_shl:4)
    // (one call per element: written as a loop over r, hipcc 7.2 emits the DPP move for r = 0 only and reuses it)
    const f32x4 hi4 = {pl_from_lane_plus4(hh[4][0]), pl_from_lane_plus4(hh[4][1]), pl_from_lane_plus4(hh[4][2]),
                       pl_from_lane_plus4(hh[4][3])};
    const f32x4 xi4 = {pl_from_lane_plus4(xx[4][0]), pl_from_lane_plus4(xx[4][1]), pl_from_lane_plus4(xx[4][2]),
                       pl_from_lane_plus4(xx[4][3])};
    auto finish_all = [&](auto mc, auto lc) {
      finish(mc, lc, hh[0], xx[0], hh[1], xx[1], t, true);
      finish(mc, lc, hh[2], xx[2], hh[3], xx[3], 16 + t, true);
      finish(mc, lc, hh[4], xx[4], hi4, xi4, 32 + (t & 3), t < 4);
    };
    if (to_lds) {
      if (masked)
        finish_all(std::true_type{}, std::true_type{});
      else
        finish_all(std::false_type{}, std::true_type{});
    } else {
      if (masked)
        finish_all(std::true_type{}, std::false_type{});
      else
        finish_all(std::false_type{}, std::false_type{});
    }
    vmin = wave_min_lane63(vmin);
    vmax = wave_max_lane63(vmax);
    if ((threadIdx.x & 63) == 63) {
      if constexpr (FUSED)
        s_mm[rem] = make_float2(vmin, vmax);
      else
        mmp[(int64_t)b * per_window + rem] = make_float2(vmin, vmax);
    }
#if defined(PL_FB_PROF)
    {
      const unsigned long long c2 = __builtin_amdgcn_s_memtime();
      pk += c1 - c0, pe += c2 - c1, ++pn_;
    }
#endif
    if (!more) break;
    pos = npos, src = nsrc, task = ntask, ntask = nntask;
  }
  }  // if (task >= 0)
  if constexpr (!FUSED) break;
  if constexpr (FUSED) {
    // ---- the window is complete: normalise + BatchNorm + split, as zpack_kernel (conv_branch.hip) ----
    __syncthreads();  // every tile's log-power values (global stores of this workgroup) and extrema (LDS) are visible
    float vmin = __int_as_float(0x7f800000), vmax = -__int_as_float(0x7f800000);
    for (int i = threadIdx.x & 63; i < per_window; i += 64) {
      vmin = fminf(vmin, s_mm[i].x);
      vmax = fmaxf(vmax, s_mm[i].y);
    }
    const float mn = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_min_lane63(vmin)), 63));
    const float mx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_max_lane63(vmax)), 63));
    const float range = mx - mn;
    const float* lpb = lp + (int64_t)win * kFrames * g.n_bins;
    uint32_t* zb = zp + (int64_t)win * kZWin;
    // only the words that carry bins: `zp`'s pad frames / pad words are zero since bp_create and nobody writes them
    const int row_u4 = (g.n_bins + 3) / 4;  // uint4 per frame that hold bins (kZPadL is a multiple of 4)
    // loads of kZb items in flight per thread (the values come back from L2): issued one by one, every item would pay
    // the round trip on its own.  The last item of a row reads up to 3 floats of the next row (masked below; `lp` is
    // allocated with that much slack behind its last row).
    constexpr int kZb = 7;
    const int n_items = kFrames * row_u4;
    for (int i0 = threadIdx.x; i0 < n_items; i0 += kZb * THREADS) {
      const int lds_bin0 = g.n_bins - kLdsBins;  // bins from here on wait in LDS
      float4 v[kZb];
#pragma unroll
      for (int k = 0; k < kZb; ++k) {
        const int i = i0 + k * THREADS;
        const int ic = i < n_items ? i : n_items - 1;
        const int t_ = ic / row_u4, g0 = 4 * (ic - t_ * row_u4);
        v[k] = float4{0.f, 0.f, 0.f, 0.f};
        if (g0 < lds_bin0) v[k] = *reinterpret_cast<const float4*>(lpb + t_ * g.n_bins + g0);  // dword alignment is enough
      }
#pragma unroll
      for (int k = 0; k < kZb; ++k) {
        const int i = i0 + k * THREADS;
        if (i >= n_items) break;
        const int t_ = i / row_u4, g0 = 4 * (i - t_ * row_u4);
        float x4[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
        if (g0 + 3 >= lds_bin0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int c = g0 + e - lds_bin0;
            if (c >= 0 && c < kLdsBins) x4[e] = s_lp[t_ * kLdsBins + c];
          }
        }
        uint32_t u[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float z = norm_bn(x4[e], mn, range, kc);
          const _Float16 hi = (_Float16)z;
          const _Float16 lo = (_Float16)((z - (float)hi) * kLoScale);
          const uint32_t w = (uint32_t)__builtin_bit_cast(unsigned short, hi) | ((uint32_t)__builtin_bit_cast(unsigned short, lo) << 16);
          u[e] = g0 + e < g.n_bins ? w : 0u;
        }
        *reinterpret_cast<uint4*>(zb + (t_ + 1) * kZRow + kZPadL + g0) = uint4{u[0], u[1], u[2], u[3]};
      }
    }
    __syncthreads();  // s_mm and the task counter are free for the next window
    if (threadIdx.x == 0) s_next = 0;
    __syncthreads();
  }
  }  // windows
#if defined(PL_FB_PROF)
  (void)pentry;
#endif
}

// ================================================================================================
// host side
static int pl_resident_waves(int n_cu) { return n_cu * 16; }

void launch_planes_split(const float* src, int64_t src_stride, int level, uint16_t* pl, int n_windows, bool ext,
                         hipStream_t stream) {
  const PlGeo g = make_pl_geo(ext);
  const int units = g.rlen[level] / 8;
  hipLaunchKernelGGL(pl_split_kernel, dim3((units + 255) / 256, n_windows), dim3(256), 0, stream, src, src_stride,
                     g.len[level], pl, g.stride, g.off[level], g.rlen[level]);
}

void launch_planes_edge_rows(const float* audio, int64_t audio_stride, uint16_t* pl, int n_windows, bool ext,
                             hipStream_t stream) {
  const PlGeo g = make_pl_geo(ext);
  hipLaunchKernelGGL(pl_edge_rows_kernel, dim3(n_windows), dim3(64), 0, stream, audio, audio_stride, g.len[0], pl, g.stride,
                     g.off[0], g.hop0);
}

void launch_planes_unsplit(const uint16_t* pl, int level, float* dst, int64_t dst_stride, int n_windows, bool ext,
                           hipStream_t stream) {
  const PlGeo g = make_pl_geo(ext);
  hipLaunchKernelGGL(pl_unsplit_kernel, dim3((g.len[level] + 255) / 256, n_windows), dim3(256), 0, stream, pl, g.stride,
                     g.off[level], g.len[level], dst, dst_stride);
}

// levels 0 (planes) and 1 .. n-1 from the fp32 audio: one wide launch for level 0 -> 1, then the rest per window
void launch_pyramid_planes(const float* audio, int64_t audio_stride, uint16_t* pl, const void* tfrag, int n_windows, int n_cu,
                           bool ext, hipStream_t stream) {
  const PlGeo g = make_pl_geo(ext);
  const uint4* tf = static_cast<const uint4*>(tfrag);
  // with at least half a window per CU the whole pyramid is ONE launch (a workgroup per window, level 1 from the audio
  // included); BP_PYR=wide keeps the wide level-0 -> 1 launch in front (A/B runs)
  static const bool wide = [] {
    const char* e = getenv("BP_PYR");
    return e && strcmp(e, "wide") == 0;
  }();
  const bool one_launch = !wide && n_windows >= n_cu / 2;
  if (!one_launch) {
    const int tiles = (g.len[1] + kPlTileOut - 1) / kPlTileOut;
    const int items = tiles * n_windows;
    int grid = (items + 3) / 4;
    if (grid > pl_resident_waves(n_cu) / 4) grid = pl_resident_waves(n_cu) / 4;
    hipLaunchKernelGGL(pl_decimate_kernel<true>, dim3(grid), dim3(256), 0, stream, audio, audio_stride, pl, g.stride, g.off[0],
                       g.len[0], g.off[1], g.len[1], tiles, tf, n_windows, g.hop0);
  }
  // with fewer windows than CUs the per-window kernel would leave most of the chip idle on the long levels: those run wide
  int first_tail = one_launch ? 1 : 2;
  if (n_windows < n_cu / 2)
    for (; first_tail < g.n_levels && g.len[first_tail] > 16 * kPlTileOut; ++first_tail) {
      const int k = first_tail;
      const int tiles = (g.len[k] + kPlTileOut - 1) / kPlTileOut;
      const int items = tiles * n_windows;
      int grid = (items + 3) / 4;
      if (grid > pl_resident_waves(n_cu) / 4) grid = pl_resident_waves(n_cu) / 4;
      hipLaunchKernelGGL(pl_decimate_kernel<false>, dim3(grid), dim3(256), 0, stream, (const float*)nullptr, (int64_t)0, pl,
                         g.stride, g.off[k - 1], g.len[k - 1], g.off[k], g.len[k], tiles, tf, n_windows, 0);
    }
  if (first_tail < g.n_levels) {
    // levels kept in LDS: as many of the deepest ones as fit (22.05 kHz: all from level 2; extended pyramid: from level 3)
    int lds_first = first_tail < 2 ? 2 : first_tail;
    while (pl_tail_lds_need(g, lds_first, g.n_levels - 1) > kPlTailLdsElems) ++lds_first;
    int last = g.n_levels - 1;
#ifdef BP_PLANES_DEBUG_HOOKS  // tools only (tools/build_variant.sh ... -DBP_PLANES_DEBUG_HOOKS): garbage results
    if (const char* e = getenv("BP_TAIL_LAST")) last = atoi(e);  // timing of the first levels
#endif
    hipLaunchKernelGGL(pl_decimate_tail_kernel, dim3(n_windows), dim3(kPlTailThreads), 0, stream, audio, audio_stride, pl, g,
                       PlTail{first_tail, last, lds_first}, tf);
  }
}

int filterbank_planes_partials(bool ext) { return make_pl_geo(ext).n_levels * kPlTilesPerLevel; }

// zp != null and enough windows to give every CU its own: the fused kernel (filterbank + normalise / BatchNorm / split of
// whole windows per workgroup) — returns true, `zp` is complete; otherwise tasks strided over the chip, extrema partials in
// `scratch` (fold them with launch_zpack_partials or launch_mm_reduce) — returns false.
// `audio` (may be null: every level from the planes): the fp32 signal the level-0 planes were made from; the interior tiles
// of level 0 then read it directly and the level-0 planes need to hold the two edge tiles only.
bool launch_filterbank_planes(const uint16_t* pl, const float* audio, int64_t audio_stride, const void* bfrag,
                              const float* sqrt_len, float* lp, float* scratch, uint32_t* zp, int n_windows, LogConsts kc,
                              int n_cu, bool ext, hipStream_t stream) {
  PlGeo g = make_pl_geo(ext);
#ifdef BP_PLANES_DEBUG_HOOKS  // tools only: timing of one level's tasks; results are garbage
  if (const char* e = getenv("BP_FB_ONLY_LEVEL")) {
    const int k = atoi(e);
    g.hop0 >>= k, g.off[0] = g.off[k], g.len[0] = g.len[k], g.n_levels = 1;
    zp = nullptr;
    if (k > 0) audio = nullptr;
  }
#endif
  const int tasks = n_windows * g.n_levels * kPlTilesPerLevel;
  static const int variant = [] {  // BP_FB_WAVES=16 / 12 / 11: waves per workgroup (A/B runs); default 16
    const char* e = getenv("BP_FB_WAVES");
    return e ? atoi(e) : 16;
  }();
  static const bool no_fuse = getenv("BP_FB_NOFUSE") != nullptr;  // A/B runs: the separate zpack launch
  const uint4* bf = static_cast<const uint4*>(bfrag);
  float2* mm = reinterpret_cast<float2*>(scratch);
  const unsigned per_window = (unsigned)(g.n_levels * kPlTilesPerLevel);
  const unsigned magic = (unsigned)((0x100000000ull + per_window - 1) / per_window);
  const bool fused = zp != nullptr && !no_fuse && 2 * n_windows >= n_cu;
  auto grid_for = [&](int waves) {
    if (fused) return n_windows < n_cu ? n_windows : n_cu;
    const int grid = (tasks + waves - 1) / waves;
    return grid > n_cu ? n_cu : grid;
  };
#define BP_PL_FB_LAUNCH(T, A, W)                                                                                         \
  do {                                                                                                                   \
    if (fused)                                                                                                           \
      hipLaunchKernelGGL((cqt_filterbank_planes_kernel<T, A, true>), dim3(grid_for(W)), dim3(T), 0, stream, pl, audio,   \
                         audio_stride, bf, sqrt_len, lp, mm, zp, n_windows, kc, g, magic);                               \
    else                                                                                                                 \
      hipLaunchKernelGGL((cqt_filterbank_planes_kernel<T, A, false>), dim3(grid_for(W)), dim3(T), 0, stream, pl, audio,  \
                         audio_stride, bf, sqrt_len, lp, mm, zp, n_windows, kc, g, magic);                               \
  } while (0)
  if (variant == 11)
    BP_PL_FB_LAUNCH(704, 7, 11);
  else if (variant == 12)
    BP_PL_FB_LAUNCH(768, 7, 12);
  else
    BP_PL_FB_LAUNCH(1024, 3, 16);
#undef BP_PL_FB_LAUNCH
  return fused;
}

}  // namespace bp
