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
#include <cmath>
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
// filterbank takes the frames of its interior tiles (frames 16 .. 159) straight from the audio; the two tiles that touch
// the ends of the signal — tile 0 (frame 0 is mirrored at the front) and tile 10 (frames 171 .. 175 run past the end; the
// 11th tile's padding frames included: finite values nobody's result depends on) — read these 32 rows instead: rows
// 0 .. 15 = the windows of frames 0 .. 15, rows 16 .. 31 = those of frames 160 .. 175, reflection applied, row pitch =
// hop0 floats (so that a lane's byte offset into a row block equals its offset into the audio: round 5 — every A fragment
// of a task is then `uniform base + one per-lane 32-bit offset`, the addressing form that costs no vector arithmetic).
__host__ __device__ inline int pl_edge_frame(int L0, int hop0) { return (L0 - 111 + hop0 - 1) / hop0; }  // first f with f hop + 111 >= L0
constexpr int kPlEdgeRows = 32;

__device__ __forceinline__ float4 pl_edge_row_load(const float* __restrict__ x, int L, int f, int hop0, int lane) {
  const int i0 = f * hop0 - kPlPad + 4 * lane;  // this lane's four samples of frame f's window
  if (i0 >= 0 && i0 + 3 < L) return *reinterpret_cast<const float4*>(x + i0);  // dword alignment is enough for global vector loads
  float e[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int i = i0 + k;
    i = i < 0 ? -i : i;                     // reflection without repeating the edge sample
    i = i >= L ? 2 * (L - 1) - i : i;
    const bool ok = i >= 0 && i < L;        // beyond one reflection: the padding frames' slack
    e[k] = ok ? x[ok ? i : 0] : 0.0f;
  }
  return float4{e[0], e[1], e[2], e[3]};
}

// rows r and 16 + r (r = 0 .. 15) by one wave: both loads in flight, then both stores.  (A wave that wrote all 16 rows of
// a block one after the other chained 16 memory round trips in front of its own tiles: measured + 4 us on the pyramid.)
__device__ __forceinline__ void pl_write_edge_row_pair(const float* __restrict__ x, int L, float* __restrict__ rows, int hop0,
                                                       int r, int lane) {
  const float4 a = pl_edge_row_load(x, L, r, hop0, lane);
  const float4 b = pl_edge_row_load(x, L, (kPlTilesPerLevel - 1) * 16 + r, hop0, lane);
  *reinterpret_cast<float4*>(rows + r * hop0 + 4 * lane) = a;
  *reinterpret_cast<float4*>(rows + (16 + r) * hop0 + 4 * lane) = b;
}

// one block of 16 rows (head: frames 0 .. 15, else frames 160 .. 175), four rows in flight at a time
__device__ __forceinline__ void pl_write_edge_rows(const float* __restrict__ x, int L, float* __restrict__ rows, int hop0,
                                                   bool head, int lane) {
  const int r0 = head ? 0 : 16, f0 = head ? 0 : (kPlTilesPerLevel - 1) * 16;
#pragma unroll 1
  for (int r = 0; r < 16; r += 4) {
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = pl_edge_row_load(x, L, f0 + r + k, hop0, lane);
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(rows + (r0 + r + k) * hop0 + 4 * lane) = v[k];
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

// tools only (tools/build_variant.sh prof cqt_planes.hip -DPL_PROF; tools/experiments/cqt_prof.py): phase stamps of two
// workgroups of the per-window kernels, [kernel 0 = pyramid, 1 = filterbank][workgroup slot][wave][stamp]
#ifdef PL_PROF
__device__ unsigned long long g_pl_prof[2][2][16][16];
#define PL_STAMP(kern, i)                                                                                   \
  do {                                                                                                      \
    if ((blockIdx.x == 0 || blockIdx.x == 131) && (threadIdx.x & 63) == 0)                                  \
      g_pl_prof[kern][blockIdx.x ? 1 : 0][threadIdx.x >> 6][i] = __builtin_amdgcn_s_memtime();             \
  } while (0)
#define PL_STAMP_RT(kern, i)                                                                                \
  do {                                                                                                      \
    if ((blockIdx.x == 0 || blockIdx.x == 131) && (threadIdx.x & 63) == 0)                                  \
      g_pl_prof[kern][blockIdx.x ? 1 : 0][threadIdx.x >> 6][i] = wall_clock64();                           \
  } while (0)
#else
#define PL_STAMP(kern, i) ((void)0)
#define PL_STAMP_RT(kern, i) ((void)0)
#endif

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
// LDS images whose 16-byte units are read as B fragments — the row images here, the resident planes of the per-window
// kernel — are XOR-swizzled: unit u of an image (4 units = one row of 32 elements) sits at u ^ 2 when bit 4 of u is set,
// i.e. the unit pairs (0, 2) and (1, 3) of rows 4 .. 7 (mod 8) are swapped.  ds_read_b128 is served in four groups of 16
// lanes ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md): at the natural 64-byte row pitch rows rho and rho + 4 (+ 12) of
// a group share banks — 8 LDS cycles per read instead of 4, with a padded pitch of 5 units (rounds 3 - 4: "4 + 1 of
// skew") just the same, with 6 none but no LDS left for it; enumerated (round 5) for every k-step, tile offset and 16-byte
// aligned base: 8 -> 4.  The decimator tiles' 18 signal-fragment reads were what the LDS pipe of a CU spent its time on
// (a round of 16 tiles: 2,880 LDS cycles against 1,400 matrix-pipe cycles per SIMD).  A lane's unit at k-step s is row
// m + s, unit kg: it reads unit kg ^ 2 p(s), p(s) = ((m + s) >> 2) & 1 — one select per k-step between two lane bases.
constexpr int kPlRowU = 4;                      // 16-byte units per row in LDS
constexpr int kPlRowsU = 2 * 24 * kPlRowU;      // hi rows, then lo rows: 192 units = 3072 bytes per wave
__device__ __forceinline__ int pl_swz_units(int u) { return u ^ ((u >> 3) & 2); }     // 16-byte units
__device__ __forceinline__ int pl_swz_elems(int e) { return e ^ ((e >> 3) & 16); }    // f16 elements (8 per unit)

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
  if constexpr (F32IN) {
    // zero outside [0, L_in) (nnaudio.py:269-279).  A row's two halves start at multiples of 4 and both window lengths are
    // multiples of 4: a half is either all signal or all padding, so the edge tiles need two predicated 16-byte loads
    // per row, not eight predicated samples.
    static_assert(kAudioN % 4 == 0 && kAudioNExt % 4 == 0 && kPlPad % 4 == 0, "no half straddles an end of the signal");
    auto row = [&](int g0, float4& lo4, float4& hi4) {
      if (!pl_tile_is_edge(tile, L_in)) {
        lo4 = *reinterpret_cast<const float4*>(x + g0);  // dword alignment is enough for global vector loads
        hi4 = *reinterpret_cast<const float4*>(x + g0 + 4);
      } else {
        lo4 = hi4 = float4{0.f, 0.f, 0.f, 0.f};
        if (g0 >= 0 && g0 + 3 < L_in) lo4 = *reinterpret_cast<const float4*>(x + g0);
        if (g0 + 4 >= 0 && g0 + 7 < L_in) hi4 = *reinterpret_cast<const float4*>(x + g0 + 4);
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

// The four consecutive outputs a lane holds after a tile's matrix work (D: column m = lane & 15, rows u = 4 kg + r):
// tap scale off, split, one 8-byte store per plane into the output level's planes (HBM, for the filterbank), optionally a
// second copy into an LDS image of the level (`mir_hi` = its sample 0, lo plane `mir_stride` elements behind; no padding
// there; null = none), and — the tiles that hold samples 1..128 and L-129..L-2 — the level's reflect padding
// (nnaudio.py:300-301).  Shared by every decimator kernel: a level's bits do not depend on which kernel made it.
// 8 / 16 bytes to global memory that only a LATER launch reads (planes, zp).  -DPL_STORE_SC1 (tools: A/B): write-through
// stores that do not leave the line in the XCD's L2 (MI355X_MICROARCH.md, "stores of each flavour").
__device__ __forceinline__ void pl_store8(void* p, uint2 v) {
#ifdef PL_STORE_SC1
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v.x | ((unsigned long long)v.y << 32),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  *reinterpret_cast<uint2*>(p) = v;
#endif
}
__device__ __forceinline__ void pl_store16(void* p, uint4 v) {
#ifdef PL_STORE_SC1
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
  const u32x4 d = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(d) : "memory");
#else
  *reinterpret_cast<uint4*>(p) = v;
#endif
}

// TAIL = false: the level's reflect padding is written by somebody else (pl_reflect_pad_from_lds); the level's last,
// partial group of four is stored here either way.
// SWZ: the LDS image is a swizzled resident region (per-window kernel): `mir_hi` = the REGION's element 0 (sample 0 sits
// kPlPad elements behind it), elements at pl_swz_elems(region index).
template <bool TAIL = true, bool SWZ = false>
__device__ __forceinline__ void pl_tile_store(const f32x4& hh, const f32x4& xx, uint16_t* __restrict__ out_hi, int64_t stride,
                                              int L_out, int tile, int lane, uint16_t* mir_hi, int mir_stride) {
  const int m = lane & 15, kg = lane >> 4;
  const int o0 = kPlTileOut * tile;
  const int n0 = o0 + 16 * m + 4 * kg;
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = (hh[r] + xx[r] * kLoUnscale) * kPlDmTapUnscale;
  uint2 h2, l2;
  split_f16x2_rn(f32x2{v[0], v[1]}, h2.x, l2.x);
  split_f16x2_rn(f32x2{v[2], v[3]}, h2.y, l2.y);
  uint16_t* oh = out_hi + kPlPad + n0;
  if (n0 + 3 < L_out) {
    pl_store8(oh, h2);
    pl_store8(oh + stride, l2);
  }
  if (mir_hi != nullptr) {  // wave-uniform
    uint16_t* mh = SWZ ? mir_hi + pl_swz_elems(kPlPad + n0) : mir_hi + n0;  // four elements: inside one half-row either way
    if (n0 + 3 < L_out) {
      *reinterpret_cast<uint2*>(mh) = h2;
      *reinterpret_cast<uint2*>(mh + mir_stride) = l2;
    } else if (n0 < L_out) {  // the level's last, partial group of four: zeros behind the last sample (images are sized in fours)
      const int nv = L_out - n0;  // 1 .. 3 valid
      const uint32_t m0 = nv > 1 ? 0xffffffffu : 0xffffu, m1 = nv > 2 ? 0xffffu : 0u;
      *reinterpret_cast<uint2*>(mh) = uint2{h2.x & m0, h2.y & m1};
      *reinterpret_cast<uint2*>(mh + mir_stride) = uint2{l2.x & m0, l2.y & m1};
    }
  }
  // the level's last (partial) group of four, and the reflect padding: the tiles that hold samples 1..128 and
  // L-129..L-2 write their mirror images (nnaudio.py:300-301).  Wave-uniform conditions, 16-bit stores.
  const bool tail = TAIL ? (o0 + kPlTileOut > L_out - 130 || tile == 0) : o0 + kPlTileOut > L_out;
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
      if constexpr (TAIL) {
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
}

// A level's reflect padding in its HBM planes (nnaudio.py:300-301: 128 mirrored samples in front and behind, the edge
// sample not repeated) from the level's LDS image (`img` = its sample 0, lo plane `img_stride` elements behind), by ONE
// wave: out[kPlPad - n] = s[n], n = 1 .. 128, and out[kPlPad + L + j] = s[L - 2 - j], j = 0 .. 127 — two elements per
// lane, end and plane.  The bits pl_tile_store<true> writes; off the tiles' critical path (below level 4 every tile of a
// level holds padding samples, and the scattered 16-bit stores were a third of a one-tile level's time).
// `region` = the swizzled resident region of the level (sample n at pl_swz_elems(kPlPad + n)).
__device__ __forceinline__ void pl_reflect_pad_from_lds(const uint16_t* region, int img_stride, int L, uint16_t* __restrict__ out_hi,
                                                        int64_t stride, int lane) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int n = 2 * lane + 1 + e;  // 1 .. 128
    const int a = pl_swz_elems(kPlPad + n);
    out_hi[kPlPad - n] = region[a];
    out_hi[stride + kPlPad - n] = region[img_stride + a];
    const int j = 2 * lane + e;      // 0 .. 127
    const int b = pl_swz_elems(kPlPad + L - 2 - j);
    out_hi[kPlPad + L + j] = region[b];
    out_hi[stride + kPlPad + L + j] = region[img_stride + b];
  }
}

// One tile through a wave-private row image in LDS (the input comes from global memory: the fp32 signal or a level's planes).
// F32IN = false: the input level is a plane region (`raw` holds the lane's units of its hi / lo planes).
// F32IN = true:  the input is the fp32 signal (level 0): rows are split in registers.  (Level 0 has no planes: what the
//                filterbank cannot read from the audio — the windows mirrored at the ends of the signal — are the edge
//                rows, written by the kernels around this routine.)
// MIRROR: see pl_tile_store.
// PF:     k-steps of LDS fragment reads kept ahead of the matrix instructions (0 = the compiler's own order, which reads
//         each fragment right in front of its use and waits: fine where four waves per SIMD and a prefetched next item
//         cover it, 1.5 k cycles per tile where a tile's latency is the critical path — the per-window kernel).
// SWZ_MIR: the mirror image is a swizzled resident region (see pl_tile_store).
template <bool F32IN, bool MIRROR = false, int PF = 0, bool SWZ_MIR = false>
__device__ __forceinline__ void pl_dec_tile(const PlRaw<F32IN>& raw, int64_t stride, int L_in, uint16_t* __restrict__ out_hi,
                                            int L_out, int tile, const uint4 (&th)[kPlDmSteps], const uint4* __restrict__ tlo,
                                            uint4* __restrict__ rows, int lane, uint16_t* mir_hi = nullptr, int mir_stride = 0) {
  // keep the lo fragments in LDS: without an opaque offset the compiler hoists the 9 item-invariant reads into registers
  asm volatile("" : "+v"(lane));
  const int m = lane & 15, kg = lane >> 4;
  const int base = 2 * kPlTileOut * tile + 32 * m + 8 * kg;  // region element of this lane's own row
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
  // the swizzled row image: row rho's unit kg at 4 rho + (kg ^ 2 p), p = (rho >> 2) & 1
  uint4* r0 = rows + m * kPlRowU + kg;
  uint4* r1 = rows + m * kPlRowU + (kg ^ 2);
  {
    uint4* wr = (m & 4) ? r1 : r0;  // rows m and m + 8 have the same p
    wr[0] = ah;
    wr[24 * kPlRowU] = al;
    if (m >= 8) {
      wr[8 * kPlRowU] = bh;
      wr[(24 + 8) * kPlRowU] = bl;
    }
  }
  auto row_at = [&](int s) { return ((m + s) & 4) ? r1 : r0; };  // the lane's unit of row m + s
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // three independent accumulator chains (hi hi, lo hi, hi lo: nine dependent matrix instructions each) — round 5: with
  // the two correction products on ONE accumulator the 18-deep dependent chain was the latency of a deep level's tile
  f32x4 hh = {0.f, 0.f, 0.f, 0.f}, xa = hh, xb = hh;
  const uint4* tl = tlo + lane;
  if constexpr (PF == 0) {
#pragma unroll
    for (int s = 0; s < kPlDmSteps; ++s) {
      const uint4* rs = row_at(s);
      const uint4 xh = rs[s * kPlRowU], xl = rs[(24 + s) * kPlRowU];
      const uint4 tls = tl[s * 64];
      hh = BP_PL_MFMA16(th[s], xh, hh);
      xa = BP_PL_MFMA16(tls, xh, xa);
      xb = BP_PL_MFMA16(th[s], xl, xb);
    }
  } else {
    uint4 xh[PF + 1], xl[PF + 1], tls[PF + 1];
    auto rd = [&](int s) {
      const uint4* rs = row_at(s);
      xh[s % (PF + 1)] = rs[s * kPlRowU];
      xl[s % (PF + 1)] = rs[(24 + s) * kPlRowU];
      tls[s % (PF + 1)] = tl[s * 64];
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) rd(s);
#pragma unroll
    for (int s = 0; s < kPlDmSteps; ++s) {
      if (s + PF < kPlDmSteps) rd(s + PF);
      __builtin_amdgcn_sched_barrier(0);
      hh = BP_PL_MFMA16(th[s], xh[s % (PF + 1)], hh);
      xa = BP_PL_MFMA16(tls[s % (PF + 1)], xh[s % (PF + 1)], xa);
      xb = BP_PL_MFMA16(th[s], xl[s % (PF + 1)], xb);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // the rows are read: the next tile of this wave may overwrite them
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  pl_tile_store<true, SWZ_MIR>(hh, xa + xb, out_hi, stride, L_out, tile, lane, MIRROR ? mir_hi : nullptr, mir_stride);
}

// One tile whose input level is resident in LDS as contiguous hi / lo planes (round 5): the B fragment of lane (m, kg) at
// k-step s — the 8 elements from 2 o0 + 32 (m + s) + 8 kg of the region that starts kPlPad elements in front of the level's
// sample 0 — is ONE ds_read_b128 straight from the plane: no row image, no copy, no wavefront fence (the level was complete
// at the last workgroup barrier).  `in_hi` = LDS address of that region's element 0, lo plane `in_stride` elements behind.
// The region is ZERO beside the samples — kPlPad elements in front, kPwTail behind (the reference zero-pads the decimator's
// input, nnaudio.py:269-279) — so no tile masks anything: with the masks in registers (18 fragments x ~26 operations in
// an edge tile, and below level 4 every tile is one) a one-tile level took 2.2 us, half of it the masks.
// At the natural 64-byte row pitch lanes m and m + 4 share banks (2-way: 8 instead of 4 LDS cycles per read, ~70 cycles a
// tile).  The same matrix instructions in the same order on the same operands as pl_dec_tile: the same bits.
// `in_tile`: the tile's index inside the region `in_hi` starts (>= 0: a region that holds only part of the level — the
// chunks of level 0 — starts at a multiple of 16 tiles).
template <int PF, bool TAIL>
__device__ __forceinline__ void pl_dec_tile_lds(const uint16_t* in_hi, int in_stride, uint16_t* __restrict__ out_hi,
                                                int64_t stride, int L_out, int tile, const uint4 (&th)[kPlDmSteps],
                                                const uint4* __restrict__ tlo, int lane, uint16_t* mir_hi, int mir_stride,
                                                int in_tile = -1) {
  asm volatile("" : "+v"(lane));  // keep the filter's lo fragments in LDS (see pl_dec_tile)
  const int m = lane & 15, kg = lane >> 4;
  // the region is swizzled (see kPlRowU): the lane's unit of row 16 tile + m + s is kg ^ 2 p(s), p(s) = ((m + s) >> 2) & 1
  const int base = 2 * kPlTileOut * (in_tile >= 0 ? in_tile : tile) + 32 * m;
  const uint4* p0 = reinterpret_cast<const uint4*>(in_hi + base + 8 * kg);
  const uint4* p1 = reinterpret_cast<const uint4*>(in_hi + base + 8 * (kg ^ 2));
  static_assert((2 * kPlTileOut) % 256 == 0, "a tile starts on a swizzle period");
  const uint4* tl = tlo + lane;
  f32x4 hh = {0.f, 0.f, 0.f, 0.f}, xa = hh, xb = hh;
  uint4 xh[PF + 1], xl[PF + 1], tls[PF + 1];
  auto rd = [&](int s) {
    const uint4* ps = ((m + s) & 4) ? p1 : p0;
    xh[s % (PF + 1)] = ps[4 * s];
    xl[s % (PF + 1)] = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(ps + 4 * s) + in_stride);
    tls[s % (PF + 1)] = tl[s * 64];
  };
#pragma unroll
  for (int s = 0; s < PF; ++s) rd(s);
#pragma unroll
  for (int s = 0; s < kPlDmSteps; ++s) {
    if (s + PF < kPlDmSteps) rd(s + PF);
    __builtin_amdgcn_sched_barrier(0);
    hh = BP_PL_MFMA16(th[s], xh[s % (PF + 1)], hh);
    xa = BP_PL_MFMA16(tls[s % (PF + 1)], xh[s % (PF + 1)], xa);
    xb = BP_PL_MFMA16(th[s], xl[s % (PF + 1)], xb);
    __builtin_amdgcn_sched_barrier(0);
  }
  pl_tile_store<TAIL, true>(hh, xa + xb, out_hi, stride, L_out, tile, lane, mir_hi, mir_stride);
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

// One level of every window.  F32IN: level 0 -> 1 from the fp32 audio (the tiles at the ends of a window also leave level
// 0's edge rows); else planes -> planes.  Waves walk (window, tile) items, the rows of the next item are fetched before the
// matrix work of the current one; no workgroup barrier after the fragments are in.  (Launches with fewer windows than half
// the CUs: the per-window kernels below would leave most of the chip idle.)
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
    if constexpr (F32IN) {
      if (tile == 0 || tile == tiles - 1)  // wave-uniform
        pl_write_edge_rows(audio + (int64_t)b * audio_stride, L_in, reinterpret_cast<float*>(w + off_in), hop0, tile == 0, lane);
    }
    pl_dec_tile<F32IN>(raw, stride, L_in, w + off_out, L_out, tile, th, tlo, rows, lane);
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
// (Since round 5 the 22.05 kHz pyramid of a launch with at least half a window per CU runs in pl_pyramid_window_kernel
// below; this one serves the extended 44.1 kHz pyramid, whose level 1 does not fit the CU's LDS, and the deep levels of
// small launches.)
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
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  uint16_t* w = pl + (int64_t)blockIdx.x * 2 * g.stride;
  // level 0's 32 edge rows, two per wave, before anything else is live in registers
  if (t.first == 1)
    pl_write_edge_row_pair(audio + (int64_t)blockIdx.x * audio_stride, g.len[0], reinterpret_cast<float*>(w + g.off[0]), g.hop0,
                           wave, lane);
  uint4 th[kPlDmSteps];
  pl_load_tfrag(tfrag, th, tlo);
  uint4* rows = rows_all + wave * kPlRowsU;
  int loff_in = 0, loff_out = kPlTailGuard;  // LDS element of sample 0 of the input / output level (when resident)
  for (int k = t.first; k <= t.last; ++k) {
    const int tiles = (g.len[k] + kPlTileOut - 1) / kPlTileOut;
    const bool in_lds = k - 1 >= t.lds_first, out_lds = k >= t.lds_first && k < t.last;  // workgroup-uniform
    uint16_t* mir = out_lds ? s_pl + loff_out : nullptr;
    if (k == 1) {
      // level 0 -> 1 straight from the fp32 audio (round 4: the wide launch that did this spent 18 of its 35 us on per-item
      // bookkeeping — queue draws, 64-bit addresses, edge predicates — with nothing to do: ablation in DESIGN_LOG.md).  Here a
      // wave walks tiles wave, wave + 16, ... of its workgroup's window, the next tile's rows in flight during the current
      // one's matrix work; level 1 goes to its planes in HBM (level 2 reads it back through L2 after the barrier below).
      const float* x = audio + (int64_t)blockIdx.x * audio_stride;
      int tile = wave;
      if (tile < tiles) {
        PlRaw<true> raw = pl_fetch_rows<true>(x, nullptr, 0, g.len[0], tile, lane);
        for (;;) {
          const int ntile = tile + kPlTailThreads / 64;
          const PlRaw<true> nraw = pl_fetch_rows<true>(x, nullptr, 0, g.len[0], ntile < tiles ? ntile : tile, lane);
          pl_dec_tile<true, false, 1>(raw, g.stride, g.len[0], w + g.off[1], g.len[1], tile, th, tlo, rows, lane);
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
      pl_dec_tile<false, true, 3>(raw, g.stride, g.len[k - 1], w + g.off[k], g.len[k], tile, th, tlo, rows, lane, mir,
                                  kPlTailLdsElems);
    }
    __syncthreads();  // level k is complete: in LDS for this workgroup (and in L2: same CU, same L1, workgroup scope)
    if (out_lds) {
      loff_in = loff_out;
      loff_out += (g.len[k] + 7) & ~7;
    }
  }
}

// The whole 22.05 kHz pyramid of one window, one workgroup of 16 waves, every level a wave reads again resident in LDS
// (round 5).  What the round-4 kernel above spent per tile — ~350 instructions around 27 matrix instructions, for level
// 1 as for the last, and a round of 16 tiles is paced by instruction issue (four waves per SIMD in lockstep behind the
// level's barrier) — was mostly staging: every input row fetched, parked in a wave-private row image, fenced and read
// back.  Here
//   * level 0 -> 1 still goes through the row images (the input is the fp32 signal in HBM: fetched a tile ahead, split in
//     registers), but level 1 ALSO stays in LDS (region A, 88 KB: hi and lo plane, samples only);
//   * levels 2 .. 8 read their B fragments straight from the resident planes (pl_dec_tile_lds: 27 ds_read_b128 of the
//     signal + 9 of the filter's lo fragments, 27 matrix instructions, the store): ~125 instructions a tile;
//   * the LDS is time-shared: the 16 row images (60 KB, region B) are dead once level 1 is complete and take level 2;
//     level 1's region is dead once level 2 is complete and takes levels 3 .. 7 back to back;
//   * the barrier between levels waits for LDS traffic only (lds_barrier): the plane stores to HBM, which only the
//     filterbank launch reads, stay in flight.
// Every level is written to its planes in HBM as before; the bits are those of the kernels above (same matrix
// instructions in the same order, same split, same masks).
constexpr int kPwThreads = 1024;
constexpr int kPwGuard = kPlPad;   // zeros in front of a resident level (the decimator's zero padding; tile 0 reads them)
constexpr int kPwTail = 768;       // zeros behind it (the last tile's fragments reach at most 767 elements past the samples)
// (whole swizzle periods of 256 elements: a region's swizzle is then that of its plane, whatever its offset in it)
__host__ __device__ constexpr int pw_level_elems(int k) { return (kPwGuard + ((level_len(k) + 7) & ~7) + kPwTail + 255) & ~255; }
constexpr int kPwPlaneA = pw_level_elems(1);  // 22824 elements per plane: level 1; later levels 3 .. 7 back to back
constexpr int kPwPlaneB = pw_level_elems(2);  // 11864: level 2, in the space of the row images
static_assert(2 * kPwPlaneB * 2 <= (kPwThreads / 64) * kPlRowsU * 16, "level 2 fits the row images' space");
static_assert(pw_level_elems(3) + pw_level_elems(4) + pw_level_elems(5) + pw_level_elems(6) + pw_level_elems(7) <= kPwPlaneA,
              "levels 3..7 fit level 1's space");
static_assert(kPlDmSteps * 64 * 16 + (kPwThreads / 64) * kPlRowsU * 16 + 2 * kPwPlaneA * 2 <= 160 * 1024, "LDS budget");

// zero elements [from, to) of both planes of a resident region (from, to multiples of 4), all threads of the workgroup
__device__ __forceinline__ void pw_zero(uint16_t* hi, int stride, int from, int to) {
  for (int i = from + 4 * (int)threadIdx.x; i < to; i += 4 * kPwThreads) {
    const int a = pl_swz_elems(i);  // region element i (a group of four stays inside its half-row)
    *reinterpret_cast<uint2*>(hi + a) = uint2{0u, 0u};
    *reinterpret_cast<uint2*>(hi + stride + a) = uint2{0u, 0u};
  }
}

// levels 2 .. 8 with compile-time geometry (the kernel serves the 22.05 kHz pyramid only): no scalar loads of the level's
// lengths and offsets between the barriers of a 1-microsecond level
template <int K>
struct PwLevel {
  static constexpr int kLen = level_len(K), kLenIn = level_len(K - 1);
  static constexpr int kTiles = (kLen + kPlTileOut - 1) / kPlTileOut;
  static constexpr bool kKeep = K < kOctaves - 1;  // somebody reads this level again: it gets an LDS image
  // element offset of the level's region (its zero guard) inside its LDS plane: level 1 and 3 open region A, 2 opens B
  static constexpr int region_off() {
    int off = 0;
    for (int j = 3; j < K; ++j) off += pw_level_elems(j);
    return K <= 3 ? 0 : off;
  }
};

template <int K>
__device__ __forceinline__ void pw_level(uint16_t* reg_a, uint16_t* b16, uint16_t* w, const int (&off)[10], int64_t stride,
                                         const uint4 (&th)[kPlDmSteps], const uint4* tlo, int wave, int lane) {
  using G = PwLevel<K>;
  using GI = PwLevel<K - 1>;
  constexpr int kWaves = kPwThreads / 64;
  // input: level K - 1 (region A for K = 2 and K >= 4, region B for K = 3); output image: B for K = 2, A otherwise
  const uint16_t* in_hi = (K == 3 ? b16 : reg_a) + GI::region_off();
  constexpr int in_stride = K == 3 ? kPwPlaneB : kPwPlaneA;
  uint16_t* img = (K == 2 ? b16 : reg_a) + G::region_off();  // level K's (swizzled) region: sample 0 kPwGuard elements in
  constexpr int img_stride = K == 2 ? kPwPlaneB : kPwPlaneA;
  if constexpr (K == 2) {  // level 2's zero surroundings in region B (the row images' space)
    pw_zero(b16, kPwPlaneB, 0, kPwGuard);
    pw_zero(b16, kPwPlaneB, kPwGuard + ((G::kLen + 3) & ~3), kPwPlaneB);
  }
  if constexpr (K == 3) {  // those of levels 3 .. 7 in region A (level 1 is dead): one contiguous stretch behind each level
    pw_zero(reg_a, kPwPlaneA, 0, kPwGuard);
    pw_zero(reg_a, kPwPlaneA, PwLevel<3>::region_off() + kPwGuard + ((level_len(3) + 3) & ~3), PwLevel<4>::region_off() + kPwGuard);
    pw_zero(reg_a, kPwPlaneA, PwLevel<4>::region_off() + kPwGuard + ((level_len(4) + 3) & ~3), PwLevel<5>::region_off() + kPwGuard);
    pw_zero(reg_a, kPwPlaneA, PwLevel<5>::region_off() + kPwGuard + ((level_len(5) + 3) & ~3), PwLevel<6>::region_off() + kPwGuard);
    pw_zero(reg_a, kPwPlaneA, PwLevel<6>::region_off() + kPwGuard + ((level_len(6) + 3) & ~3), PwLevel<7>::region_off() + kPwGuard);
    pw_zero(reg_a, kPwPlaneA, PwLevel<7>::region_off() + kPwGuard + ((level_len(7) + 3) & ~3), PwLevel<7>::region_off() + pw_level_elems(7));
  }
  // the reflect padding of level K - 1's planes, from its image, by the youngest wave (it has the fewest tiles): levels
  // 2 .. 7; level 1 and level 8 (no image) write theirs from the tiles
  if constexpr (K >= 3)
    if (wave == kWaves - 1)
      pl_reflect_pad_from_lds(in_hi, in_stride, GI::kLen, w + off[K - 1], stride, lane);
  for (int tile = wave; tile < G::kTiles; tile += kWaves)
    pl_dec_tile_lds<3, !G::kKeep>(in_hi, in_stride, w + off[K], stride, G::kLen, tile, th, tlo, lane, G::kKeep ? img : nullptr,
                                  img_stride);
  lds_barrier();  // level K is complete
  PL_STAMP(0, 3 + K);
}

__global__ __launch_bounds__(kPwThreads) void pl_pyramid_window_kernel(const float* __restrict__ audio, int64_t audio_stride,
                                                                      uint16_t* __restrict__ pl, PlGeo g,
                                                                      const uint4* __restrict__ tfrag) {
  __shared__ __attribute__((aligned(16))) uint4 tlo[kPlDmSteps * 64];
  __shared__ __attribute__((aligned(16))) uint4 reg_b[(kPwThreads / 64) * kPlRowsU];
  __shared__ __attribute__((aligned(16))) uint16_t reg_a[2 * kPwPlaneA];
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  constexpr int kWaves = kPwThreads / 64;
  static_assert(kWaves == 16, "two of level 0's 32 edge rows per wave");
  uint16_t* w = pl + (int64_t)blockIdx.x * 2 * g.stride;
  uint16_t* const b16 = reinterpret_cast<uint16_t*>(reg_b);
  const float* x = audio + (int64_t)blockIdx.x * audio_stride;
  constexpr int kL0 = kAudioN, kL1 = level_len(1);
  constexpr int tiles1 = (kL1 + kPlTileOut - 1) / kPlTileOut;
  PL_STAMP(0, 0);
  PL_STAMP_RT(0, 14);
  // The kernel's first memory round trips ALL AT ONCE (issued one behind the other they were 5.6 us of a 36 us kernel):
  // the first chunk of the signal, the filter fragments (hi: registers, lo: one 16-byte piece per thread for LDS), the
  // samples of this wave's two edge rows.
  //
  // Level 0 -> 1 through resident CHUNKS (round 5, second half): 16 tiles' worth of the fp32 signal (8,192 samples + the
  // 768 the last tile's fragments reach past them, zeros outside [0, L): nnaudio.py:269-279) are split to f16 hi + lo ONCE
  // per sample by the whole workgroup — coalesced float4 loads, the next chunk's in flight during this chunk's tiles — and
  // parked as swizzled planes in region B; the tiles then read their fragments straight from the planes like levels 2 .. 8
  // do (~150 instructions a tile).  The wave-private row images before it fetched, split and parked every sample 1.5 times
  // in per-lane code: ~350 instructions a tile, and instruction issue is what paces a round of 16 tiles.
  constexpr int kChunkTiles = 16;
  constexpr int kChunkSamples = 2 * kPlTileOut * kChunkTiles;        // 8192
  constexpr int kChunkElems = kChunkSamples + kPwTail;               // 8960 region elements per plane
  constexpr int kChunkGroups = kChunkElems / 4;                      // float4 groups: 2240
  constexpr int kChunkPer = (kChunkGroups + kPwThreads - 1) / kPwThreads;
  constexpr int kChunks = (tiles1 + kChunkTiles - 1) / kChunkTiles;
  static_assert(kChunkElems % 256 == 0 && 2 * kChunkElems * 2 <= (kPwThreads / 64) * kPlRowsU * 16, "a chunk fits region B");
  static_assert(kL0 % 4 == 0 && kPwGuard % 4 == 0, "a group of four samples lies inside the signal or outside it");
  float4 cv[kChunkPer];
  auto chunk_load = [&](int c) {
#pragma unroll
    for (int k = 0; k < kChunkPer; ++k) {
      const int gq = (int)threadIdx.x + k * kPwThreads;
      const int n = kChunkSamples * c + 4 * gq - kPwGuard;  // sample index of the group's first element
      cv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gq < kChunkGroups && n >= 0 && n < kL0) cv[k] = *reinterpret_cast<const float4*>(x + n);
    }
  };
  auto chunk_store = [&]() {
#pragma unroll
    for (int k = 0; k < kChunkPer; ++k) {
      const int gq = (int)threadIdx.x + k * kPwThreads;
      if (gq < kChunkGroups) {
        uint2 h2, l2;
        split_f16x2_rn(f32x2{cv[k].x, cv[k].y}, h2.x, l2.x);
        split_f16x2_rn(f32x2{cv[k].z, cv[k].w}, h2.y, l2.y);
        const int a = pl_swz_elems(4 * gq);
        *reinterpret_cast<uint2*>(b16 + a) = h2;
        *reinterpret_cast<uint2*>(b16 + kChunkElems + a) = l2;
      }
    }
  };
  chunk_load(0);
  uint4 th[kPlDmSteps];
#pragma unroll
  for (int s = 0; s < kPlDmSteps; ++s) th[s] = tfrag[s * 64 + lane];
  static_assert(kPlDmSteps * 64 <= kPwThreads, "one lo-fragment piece per thread");
  uint4 tlo_piece = uint4{0u, 0u, 0u, 0u};
  if (threadIdx.x < kPlDmSteps * 64) tlo_piece = tfrag[kPlDmSteps * 64 + threadIdx.x];
  float* edge_rows = reinterpret_cast<float*>(w + g.off[0]);
  const float4 er_a = pl_edge_row_load(x, kL0, wave, g.hop0, lane);
  const float4 er_b = pl_edge_row_load(x, kL0, (kPlTilesPerLevel - 1) * 16 + wave, g.hop0, lane);
  // (nothing above waits) level 1's zero surroundings in region A — its samples come from the tiles below
  pw_zero(reg_a, kPwPlaneA, 0, kPwGuard);
  pw_zero(reg_a, kPwPlaneA, kPwGuard + ((kL1 + 3) & ~3), kPwPlaneA);
  if (threadIdx.x < kPlDmSteps * 64) tlo[threadIdx.x] = tlo_piece;
  *reinterpret_cast<float4*>(edge_rows + wave * g.hop0 + 4 * lane) = er_a;
  *reinterpret_cast<float4*>(edge_rows + (16 + wave) * g.hop0 + 4 * lane) = er_b;
  PL_STAMP(0, 1);
  int off[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) off[k] = g.off[k];
#pragma unroll 1
  for (int c = 0; c < kChunks; ++c) {
    chunk_store();                      // chunk c: split, into region B
    if (c + 1 < kChunks) chunk_load(c + 1);
    lds_barrier();                      // the chunk (and, the first time, the lo fragments) are in LDS
    if (c == 0) PL_STAMP(0, 2);
    const int tile = kChunkTiles * c + wave;
    if (tile < tiles1)
      pl_dec_tile_lds<3, true>(b16, kChunkElems, w + off[1], g.stride, kL1, tile, th, tlo, lane, reg_a, kPwPlaneA, wave);
    lds_barrier();                      // every wave has read the chunk
  }
  PL_STAMP(0, 3);
  PL_STAMP(0, 4);  // (level 1 is complete in region A, the chunk buffer is dead: the loop's last barrier)
  pw_level<2>(reg_a, b16, w, off, g.stride, th, tlo, wave, lane);
  pw_level<3>(reg_a, b16, w, off, g.stride, th, tlo, wave, lane);
  pw_level<4>(reg_a, b16, w, off, g.stride, th, tlo, wave, lane);
  pw_level<5>(reg_a, b16, w, off, g.stride, th, tlo, wave, lane);
  pw_level<6>(reg_a, b16, w, off, g.stride, th, tlo, wave, lane);
  pw_level<7>(reg_a, b16, w, off, g.stride, th, tlo, wave, lane);
  pw_level<8>(reg_a, b16, w, off, g.stride, th, tlo, wave, lane);
  PL_STAMP_RT(0, 15);
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

// Normalise + BatchNorm + split of four consecutive bins into `zp` words (signal.py:177-183, models.py:187-189):
// z = (lp - min) * (bn_a / range) + bn_b, hi = rn_f16(z), lo = rn_f16((z - hi) 2^11), word = hi | lo << 16.
__device__ __forceinline__ uint4 pl_zp_pack4(const float (&x)[4], float mn, float nk, float bn_b) {
  uint32_t u[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float z = norm_bn_k(x[e], mn, nk, bn_b);
    const _Float16 hi = (_Float16)z;
    const _Float16 lo = (_Float16)((z - (float)hi) * kLoScale);
    u[e] = (uint32_t)__builtin_bit_cast(unsigned short, hi) | ((uint32_t)__builtin_bit_cast(unsigned short, lo) << 16);
  }
  return uint4{u[0], u[1], u[2], u[3]};
}

// THREADS / APF: 1024 threads = 4 waves per SIMD (128 VGPRs each) keep three k-steps of A fragments ahead; 768 / 704
// threads = 3 waves per SIMD with up to 168 VGPRs hold the whole next task's fragments in flight.
// FUSED: a workgroup owns whole windows (its waves draw the window's 99 tasks), keeps the tiles' extrema in LDS and, when
// the window's last task is done, normalises its log-power map itself and writes the pre-split, BatchNorm-ed `zp` words
// (signal.py:177-183, models.py:187-189: what zpack_kernel does in a launch of its own) — the map was written by this
// CU a few microseconds ago and comes back from L2, the extrema never leave the CU.  !FUSED: tasks strided over all
// workgroups, extrema partials to `mmp` (launches with fewer windows than CUs, and the per-stage test hook).
//
// Round 5 (the kernel was paced by its instruction count: 9 vector instructions per matrix instruction, of which the
// normalise phase issued 45 %):
//  * every A fragment is `uniform base (SGPR pair) + one 32-bit lane offset`: no 64-bit vector address arithmetic.  The
//    two level-0 tiles that touch the ends of the signal read the 32 edge rows (above), whose row pitch makes the lane
//    offset the same as into the audio;
//  * the epilogue works on the accumulators as they are: with s = sqrt(len_b) 2^-12 (nnaudio.py:649-650 and the taps'
//    scale) the reference's 10 log10((s re)^2 + (s im)^2 + eps) is kln2 [log2(re^2 + im^2 + eps / s^2) + log2(s^2)]; the
//    two per-bin constants come from an LDS table (built at the kernel's start): 7 instead of 13 operations per value
//    and no sqrt (the reference takes the root for the magnitude and squares it again, nnaudio.py:661, signal.py:174);
//  * FUSED: a lane keeps running extrema over all its tasks of a window; one DPP reduction per wave and window;
//  * the normalise phase runs over three index spaces (bins that come back from L2, bins in LDS, the one mixed group
//    of four) with compile-time divisors, the affine map folded to (lp - min) * (bn_a / range) + bn_b.
template <int THREADS, int APF, bool FUSED, bool EXT>
__global__ __launch_bounds__(THREADS) void cqt_filterbank_planes_kernel(
    const uint16_t* __restrict__ pl, const float* __restrict__ audio, int64_t audio_stride,
    const uint4* __restrict__ bfrag, const float* __restrict__ bin_eps, float* __restrict__ lp, float2* __restrict__ mmp,
    uint32_t* __restrict__ zp, int n_windows, LogConsts kc, PlGeo g, unsigned per_window_magic) {
  constexpr int NB = EXT ? kBinsExt : kBins;
  constexpr int NL = EXT ? kOctavesExt : kOctaves;
  constexpr int HOP0 = EXT ? 512 : 256;
  constexpr int kLog2Hop0 = EXT ? 9 : 8;
  constexpr int kPerWindow = NL * kPlTilesPerLevel;
  constexpr int kWaves = THREADS / 64;
  static_assert(kPlTilesPerLevel == 11, "the multiply-shift below divides by 11");
  __shared__ __attribute__((aligned(16))) uint4 bfr[kPlFbFrags * 2 * 64];
  // the per-bin constants and the level offsets from LDS, not from global / constant memory: a wave's memory counters are
  // in order, so a global load in the epilogue would wait for every A fragment prefetched for the next task before it
  __shared__ float2 s_bin[NB];
  __shared__ int s_off[10];
  __shared__ int s_next;
  __shared__ float2 s_mm[FUSED ? kWaves : 1];
  // FUSED: the log-power values of the four top levels (144 bins x 172 frames = 97 KB: what is left of the CU's LDS) wait
  // for the normalise phase here instead of making the round trip through L2 / HBM
  constexpr int kLdsBins = 4 * kBpo;
  constexpr int kLdsBin0 = NB - kLdsBins;  // bins from here on wait in LDS
  __shared__ float s_lp[FUSED ? kFrames * kLdsBins : 1];
  PL_STAMP(1, 0);
  PL_STAMP_RT(1, 14);
  // log2 -> 10 log10 (uniform: kept in a scalar register)
  const float kln2 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(0.69314718055994531f * kc.s0 * kc.s1)));
  if (threadIdx.x == 0) s_next = 0;
  for (int i = threadIdx.x; i < kPlFbFrags * 2 * 64; i += THREADS) bfr[i] = bfrag[i];
  {
    // The per-bin pair {eps_b = eps / s^2, c_b = kln2 log2(s^2)}.  c_b is derived HERE, with the very instructions the
    // epilogue uses, as c_b = rn(v0 - log2(eps_b) kln2), v0 = the log-power of a silent bin (10 log10(eps), the value
    // every bin had before round 5): a bin whose power vanishes beside eps_b (digital silence) then evaluates to
    // rn(log2(eps_b) kln2 + c_b) = v0 EXACTLY, whatever its bin (|c_b| < 64 rounds to 2^-19, half an ulp of v0 ~ -100 is
    // 2^-18) — the window's range is exactly 0 and divide_no_nan yields the reference's constant map (signal.py:179-183).
    // With a table rounded on the host the silent window's extrema differed in the last bit and the normalisation blew
    // that up to a full-scale pattern.
    const float v0 = __fmul_rn(__builtin_amdgcn_logf(kc.eps), kln2);
    for (int i = threadIdx.x; i < NB; i += THREADS) {
      const float e = bin_eps[i];
      s_bin[i] = make_float2(e, __fmaf_rn(-__builtin_amdgcn_logf(e), kln2, v0));
    }
  }
  if (threadIdx.x < 10) s_off[threadIdx.x] = g.off[threadIdx.x];
  __syncthreads();
  PL_STAMP(1, 1);
  int lane = threadIdx.x & 63;
  const int n_tasks = n_windows * kPerWindow;
  const float kInf = __int_as_float(0x7f800000);
  // D row (frame of the tile) = 4 kg + r, column (filter of the group) = t.  Every per-lane offset below is re-derived
  // from the (opaque) lane index inside the task loop — a handful of operations per task — instead of living in a dozen
  // registers through it: at 128 registers per lane the kernel would spill them.
  int t = lane & 15, kg = lane >> 4;
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
    if constexpr (FUSED) return j < kPerWindow ? win * kPerWindow + j : -1;
    const int task_ = blockIdx.x + gridDim.x * j;
    return task_ < n_tasks ? task_ : -1;
  };
  auto pos_of = [&](int task_) {  // task / per_window by multiply-shift (exact below 2^32 / 95 for 99, 2^32 / 4 for 110)
    if constexpr (FUSED) return Pos{win, task_ - win * kPerWindow};
    const int b_ = (int)__umulhi((unsigned)task_, per_window_magic);
    return Pos{b_, task_ - b_ * kPerWindow};
  };
  // Where a task's A fragments come from: two uniform bases (first / second 16-byte half of a k-step's fragment) + this
  // lane's byte offset + step * s.  Planes: 16 bytes of the hi plane per k-step (32 elements apart), the lo plane g.stride
  // elements behind it.  Level 0: the fp32 audio itself — 8 samples = two 16-byte loads per k-step, split to hi / lo in
  // registers when the k-step is consumed (every level-0 sample feeds at most one frame: hop >= window, so nothing is
  // split twice) — or, for the two tiles at the ends of the signal, the edge rows.
  struct Src {
    const char* p0;
    const char* p1;
    uint32_t voff;
    int step;  // bytes between k-steps
    int raw;   // fp32 samples: split at consumption (an int: a bool's padding bytes made the struct copies go through scratch)
  };
  auto src_of = [&](Pos p) -> Src {
    const int level_ = (p.rem * 745) >> 13;  // rem / 11 for rem < 2700
    const int tile_ = p.rem - level_ * kPlTilesPerLevel;
    const uint16_t* wpl = pl + (int64_t)p.b * 2 * g.stride;
    if (level_ == 0) {  // wave-uniform
      const float* a = (tile_ == 0 || tile_ == kPlTilesPerLevel - 1)
                           ? reinterpret_cast<const float*>(wpl + __builtin_amdgcn_readfirstlane(s_off[0])) + (tile_ ? 16 * HOP0 : 0)
                           : audio + (int64_t)p.b * audio_stride + (16 * tile_ * HOP0 - kPlPad);
      const char* c = reinterpret_cast<const char*>(a);
      return Src{c, c + 16, (uint32_t)(t * HOP0 + 16 + 8 * kg) * 4u, 128, 1};  // fp32 samples, frame pitch = hop0
    }
    const int sh = kLog2Hop0 - level_;  // log2(hop of the level)
    const char* c = reinterpret_cast<const char*>(wpl + __builtin_amdgcn_readfirstlane(s_off[level_]) + ((16 * tile_) << sh));
    return Src{c, c + 2 * g.stride, (((uint32_t)t << sh) << 1) + 32u + 16u * (uint32_t)kg, 64, 0};  // f16 elements
  };
  auto load16 = [](const char* base, uint32_t off) {
    uint4 v;
    __builtin_memcpy(&v, __builtin_assume_aligned(base + off, 2), 16);  // 2-byte alignment at the hop-1 level; dword at least elsewhere
    return v;
  };
  for (; win < (FUSED ? n_windows : blockIdx.x + 1); win += gridDim.x) {
  float rmin = kInf, rmax = -kInf;  // FUSED: this lane's extrema over its tasks of the window
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
      ah[s] = load16(src.p0, src.voff + src.step * s);
      al[s] = load16(src.p1, src.voff + src.step * s);
    }
  }
  for (;;) {
    // keep the filter fragments in LDS: without an opaque offset the compiler hoists all 58 loop-invariant reads
    asm volatile("" : "+v"(lane));
    t = lane & 15, kg = lane >> 4;
    const int nntask = ntask >= 0 ? grab() : -1;  // drawn a task ahead: its LDS round trip is nobody's critical path
    const int b = pos.b, rem = pos.rem;
    const int level = (rem * 745) >> 13, tile = rem - level * kPlTilesPerLevel;
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
#ifdef PL_FB_PAIRED
      // tools (A/B): the second correction product of item i - 1 behind the first two of item i, so that the two updates of
      // an xx accumulator are three matrix instructions apart instead of back to back (same order per accumulator: same bits)
      hh[q] = BP_PL_MFMA16(ah[s], bh[i], hh[q]);
      xx[q] = BP_PL_MFMA16(al[s], bh[i], xx[q]);
      if (i > 0) xx[item(i - 1).q] = BP_PL_MFMA16(ah[item(i - 1).s], bw[i - 1], xx[item(i - 1).q]);
      if (i + 1 == kPlFbFrags) xx[q] = BP_PL_MFMA16(ah[s], bw[i], xx[q]);
#else
      hh[q] = BP_PL_MFMA16(ah[s], bh[i], hh[q]);
      xx[q] = BP_PL_MFMA16(al[s], bh[i], xx[q]);
      xx[q] = BP_PL_MFMA16(ah[s], bw[i], xx[q]);
#endif
      if (i + 1 == kPlFbFrags || item(i + 1).s != s) {  // last product of k-step s: its ring slot takes step s + APF
        const int sn = s + APF;
        if (sn < 7) {
          ah[sn] = load16(src.p0, src.voff + src.step * sn);
          al[sn] = load16(src.p1, src.voff + src.step * sn);
        } else {
          ah[sn - 7] = load16(nsrc.p0, nsrc.voff + nsrc.step * (sn - 7));
          al[sn - 7] = load16(nsrc.p1, nsrc.voff + nsrc.step * (sn - 7));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    // epilogue: D row (frame) = 4 kg + r, column (filter of the group) = lane & 15
    const int bin0 = (NL - 1 - level) * kBpo - 15;  // nnaudio.py:640-642
    char* lp_t = reinterpret_cast<char*>(lp + ((int64_t)b * kFrames + 16 * tile) * NB + bin0);
    const int fr0 = 16 * tile + 4 * kg;
    const uint32_t so_kg = __umul24((unsigned)kg, 16u * NB);             // lp stores: + (r NB + 16 group) floats
    const uint32_t so_hbm = so_kg + 4u * (unsigned)t, so_hbm4 = so_kg + 4u * (32u + ((unsigned)t & 3u));
    const int so_lds = 4 * kg * kLdsBins + t, so_lds4 = 4 * kg * kLdsBins + 32 + (t & 3);
    float vmin = FUSED ? rmin : kInf, vmax = FUSED ? rmax : -kInf;
    // `masked` (wave-uniform): the tile has padding frames (the 11th tile) or the level has bins below the CQT's first
    // (the deepest level): 8 of 10 tasks have neither, and then only group 4's unused columns need a predicate
    const bool masked = tile == kPlTilesPerLevel - 1 || bin0 < 0;
    const bool to_lds = FUSED && level < 4;  // wave-uniform
    float* lds_t = s_lp + 16 * tile * kLdsBins + (3 - level) * kBpo;
    auto finish = [&](auto masked_c, auto lds_c, const f32x4& hr, const f32x4& xr, const f32x4& hi_, const f32x4& xi, int k,
                      int grp, bool col_ok) {
      constexpr bool kMasked = decltype(masked_c)::value, kLds = decltype(lds_c)::value;
      const bool bin_ok = col_ok && (!kMasked || bin0 + k >= 0);
      const float2 bc = s_bin[kMasked ? (bin0 + k >= 0 ? bin0 + k : 0) : bin0 + k];
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float re = __fmaf_rn(xr[r], kLoUnscale, hr[r]);
        const float im = __fmaf_rn(xi[r], kLoUnscale, hi_[r]);
        const float pw = __fmaf_rn(im, im, __fmul_rn(re, re));
        // nnaudio.py:649-661 and signal.py:174-175 in accumulator units (see the header), hardware 1-ulp log2
        v[r] = __fmaf_rn(__builtin_amdgcn_logf(__fadd_rn(pw, bc.x)), kln2, bc.y);
      }
      auto put = [&](int r) {
        if constexpr (kLds)
          lds_t[(grp == 2 ? so_lds4 : so_lds + 16 * grp) + r * kLdsBins] = v[r];
        else
          *reinterpret_cast<float*>(lp_t + (grp == 2 ? so_hbm4 : so_hbm) + (grp == 2 ? 0 : 64 * grp) + r * NB * 4) = v[r];
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
      finish(mc, lc, hh[0], xx[0], hh[1], xx[1], t, 0, true);
      finish(mc, lc, hh[2], xx[2], hh[3], xx[3], 16 + t, 1, true);
      finish(mc, lc, hh[4], xx[4], hi4, xi4, 32 + (t & 3), 2, t < 4);
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
    if constexpr (FUSED) {
      rmin = vmin, rmax = vmax;
    } else {
      vmin = wave_min_lane63(vmin);
      vmax = wave_max_lane63(vmax);
      if ((threadIdx.x & 63) == 63) mmp[(int64_t)b * kPerWindow + rem] = make_float2(vmin, vmax);
    }
    if (!more) break;
    pos = npos, src = nsrc, task = ntask, ntask = nntask;
  }
  }  // if (task >= 0)
  if constexpr (!FUSED) break;
  if constexpr (FUSED) {
    // ---- the window is complete: normalise + BatchNorm + split, as zpack_kernel (conv_branch.hip) ----
    PL_STAMP(1, 2);
    rmin = wave_min_lane63(rmin);
    rmax = wave_max_lane63(rmax);
    if ((threadIdx.x & 63) == 63) s_mm[threadIdx.x >> 6] = make_float2(rmin, rmax);
    __syncthreads();  // every tile's log-power values (global stores of this workgroup / LDS) and the waves' extrema are visible
    PL_STAMP(1, 3);
    float vmin = kInf, vmax = -kInf;
    if ((threadIdx.x & 63) < kWaves) {
      const float2 e = s_mm[threadIdx.x & 63];
      vmin = e.x, vmax = e.y;
    }
    const float mn = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_min_lane63(vmin)), 63));
    const float mx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_max_lane63(vmax)), 63));
    const float nk = norm_scale(mn, mx, kc);
    const float* lpb = lp + (int64_t)win * kFrames * NB;
    // only the words that carry bins: `zp`'s pad frames / pad words are zero since bp_create and nobody writes them
    uint32_t* zb = zp + (int64_t)win * kZWin + kZRow + kZPadL;  // frame 0, bin 0 (kZPadL is a multiple of 4)
    constexpr int kJ = (NB + 3) / 4;     // groups of four bins per frame
    constexpr int kJH = kLdsBin0 / 4;    // groups whose four bins all come back from L2
    static_assert(kLdsBin0 % 4 == 1 && NB % 4 == 1, "group kJH is {1 bin from L2, 3 from LDS}; the last group holds one bin");
    constexpr int kJL = kJ - kJH - 1;    // groups whose bins all wait in LDS: columns 4 j' + 3 .. 4 j' + 6
    // (the thread index through an opaque copy: the index arithmetic below does not depend on the window, and hoisted out
    // of the window loop it would sit in ~40 registers through the task loop — the compiler spilled them to scratch)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const char* lpc = reinterpret_cast<const char*>(lpb);
    char* zc = reinterpret_cast<char*>(zb);
    // i / d and i % d for the three small index spaces on the full-rate 24-bit multiplier (the generic 32-bit forms are
    // quarter rate): q = (i M) >> 20 with M = ceil(2^20 / d), exact while i (M d - 2^20) < 2^20
    auto divmod = [](int i, auto d_c, auto n_c) {
      constexpr unsigned d = decltype(d_c)::value, n = decltype(n_c)::value, M = ((1u << 20) + d - 1) / d;
      static_assert((unsigned long long)n * (M * d - (1u << 20)) < (1u << 20) && (unsigned long long)n * M < (1ull << 32), "exact");
      const unsigned q = __umul24((unsigned)i, M) >> 20;
      return uint2{q, (unsigned)i - __umul24(q, d)};
    };
    // (A) from L2: all loads of a thread in flight at once (issued one by one, every item would pay the round trip; the
    // memory clobber keeps the compiler from sinking each load into the block that uses it)
    {
      constexpr int nA = kFrames * kJH;
      constexpr int kZb = (nA + THREADS - 1) / THREADS;
      float4 v[kZb];
#pragma unroll
      for (int k = 0; k < kZb; ++k) {
        const int i = tid + k * THREADS;
        const uint2 tj = divmod(i < nA ? i : nA - 1, std::integral_constant<unsigned, kJH>{}, std::integral_constant<unsigned, nA>{});
        v[k] = *reinterpret_cast<const float4*>(lpc + (__umul24(tj.x, NB * 4u) + 16u * tj.y));  // dword alignment is enough
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int k = 0; k < kZb; ++k) {
        const int i = tid + k * THREADS;
        if (i < nA) {
          const uint2 tj = divmod(i, std::integral_constant<unsigned, kJH>{}, std::integral_constant<unsigned, nA>{});
          const float x4[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
          pl_store16(zc + (__umul24(tj.x, kZRow * 4u) + 16u * tj.y), pl_zp_pack4(x4, mn, nk, kc.bn_b));
        }
      }
    }
    PL_STAMP(1, 4);
    // (B) from LDS
    {
      constexpr int nB = kFrames * kJL;
      for (int i = tid; i < nB; i += THREADS) {
        const uint2 tj = divmod(i, std::integral_constant<unsigned, kJL>{}, std::integral_constant<unsigned, nB>{});
        const float* row = s_lp + __umul24(tj.x, (unsigned)kLdsBins) + 4u * tj.y + 3u;
        const bool last = tj.y == kJL - 1;  // bins NB - 1 .. NB + 2: one bin, three pad words
        const float x4[4] = {row[0], last ? 0.f : row[1], last ? 0.f : row[2], last ? 0.f : row[3]};
        uint4 w = pl_zp_pack4(x4, mn, nk, kc.bn_b);
        if (last) w.y = w.z = w.w = 0u;
        pl_store16(zc + (__umul24(tj.x, kZRow * 4u) + 16u * (tj.y + kJH + 1)), w);
      }
    }
    // (C) the mixed group: bin 4 kJH from L2, the next three from LDS
    if (tid < kFrames) {
      const float x4[4] = {lpb[tid * NB + 4 * kJH], s_lp[tid * kLdsBins], s_lp[tid * kLdsBins + 1], s_lp[tid * kLdsBins + 2]};
      pl_store16(zc + (__umul24((unsigned)tid, kZRow * 4u) + 16u * kJH), pl_zp_pack4(x4, mn, nk, kc.bn_b));
    }
    PL_STAMP(1, 5);
    __syncthreads();  // s_lp, s_mm and the task counter are free for the next window
    PL_STAMP(1, 6);
    PL_STAMP_RT(1, 15);
    if (threadIdx.x == 0) s_next = 0;
    __syncthreads();
  }
  }  // windows
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

// levels 1 .. n-1 (planes) and level 0's edge rows from the fp32 audio
void launch_pyramid_planes(const float* audio, int64_t audio_stride, uint16_t* pl, const void* tfrag, int n_windows, int n_cu,
                           bool ext, hipStream_t stream) {
  const PlGeo g = make_pl_geo(ext);
  const uint4* tf = static_cast<const uint4*>(tfrag);
  // with at least half a window per CU the whole pyramid is ONE launch, a workgroup per window (the 22.05 kHz kernel from
  // a third: a 3-minute track is 110 windows, and the four wide launches it took below half the CUs cost 55 us where
  // 110 workgroups of the per-window kernel take 40)
  const bool one_launch = ext ? n_windows >= n_cu / 2 : 3 * n_windows >= n_cu;
  if (one_launch && !ext) {
    hipLaunchKernelGGL(pl_pyramid_window_kernel, dim3(n_windows), dim3(kPwThreads), 0, stream, audio, audio_stride, pl, g, tf);
    return;
  }
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
  if (!one_launch)
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
    hipLaunchKernelGGL(pl_decimate_tail_kernel, dim3(n_windows), dim3(kPlTailThreads), 0, stream, audio, audio_stride, pl, g,
                       PlTail{first_tail, g.n_levels - 1, lds_first}, tf);
  }
}

int filterbank_planes_partials(bool ext) { return make_pl_geo(ext).n_levels * kPlTilesPerLevel; }

// The per-bin constant of the filterbank's epilogue (see the kernel's header): eps / s^2 with s = sqrt(len_b) 2^-12,
// evaluated in float64 and rounded once.  (Its partner kln2 log2(s^2) is derived on the device: see the kernel.)
void filterbank_planes_bin_consts(const float* sqrt_len, int n_bins, LogConsts kc, float* out) {
  for (int b = 0; b < n_bins; ++b) {
    const double s = (double)sqrt_len[b] * (double)kPlFmTapUnscale;
    out[b] = (float)((double)kc.eps / (s * s));
  }
}

// zp != null and enough windows to give every CU its own: the fused kernel (filterbank + normalise / BatchNorm / split of
// whole windows per workgroup) — returns true, `zp` is complete; otherwise tasks strided over the chip, extrema partials in
// `scratch` (fold them with launch_zpack_partials or launch_mm_reduce) — returns false.
// `audio`: the fp32 signal (level 0 has no planes: the interior tiles of level 0 read it directly, the two tiles at the
// ends of the signal read the edge rows the pyramid kernel / launch_planes_edge_rows left where level 0's planes were).
bool launch_filterbank_planes(const uint16_t* pl, const float* audio, int64_t audio_stride, const void* bfrag,
                              const float* bin_consts, float* lp, float* scratch, uint32_t* zp, int n_windows, LogConsts kc,
                              int n_cu, bool ext, hipStream_t stream) {
  const PlGeo g = make_pl_geo(ext);
  const int tasks = n_windows * g.n_levels * kPlTilesPerLevel;
  const uint4* bf = static_cast<const uint4*>(bfrag);
  const float* bk = bin_consts;
  float2* mm = reinterpret_cast<float2*>(scratch);
  const unsigned per_window = (unsigned)(g.n_levels * kPlTilesPerLevel);
  const unsigned magic = (unsigned)((0x100000000ull + per_window - 1) / per_window);
  // one window per workgroup pays from half a window per CU on.  (A file job's 110-window tracks on three lanes, round 5:
  // with the fused form from 32 windows on, 1,258 files/s against 1,460 — a third of the CUs for 60 us is worse than all
  // of them for 29 + 15 us even when other lanes' kernels could fill the rest.)
  const bool fused = zp != nullptr && 2 * n_windows >= n_cu;
  // (768 threads with 7 / 5 k-steps of A fragments ahead, 158 VGPRs: 0.070 - 0.074 / 0.069 ms against 0.069 - 0.070, round 5)
  constexpr int kThreads = 1024, kApf = 3;
  int grid;
  if (fused) {
    grid = n_windows < n_cu ? n_windows : n_cu;
  } else {
    grid = (tasks + kThreads / 64 - 1) / (kThreads / 64);
    if (grid > n_cu) grid = n_cu;
  }
#define BP_PL_FB_LAUNCH(F, E)                                                                                           \
  hipLaunchKernelGGL((cqt_filterbank_planes_kernel<kThreads, kApf, F, E>), dim3(grid), dim3(kThreads), 0, stream, pl,   \
                     audio, audio_stride, bf, bk, lp, mm, zp, n_windows, kc, g, magic)
  if (fused) {
    if (ext) BP_PL_FB_LAUNCH(true, true); else BP_PL_FB_LAUNCH(true, false);
  } else {
    if (ext) BP_PL_FB_LAUNCH(false, true); else BP_PL_FB_LAUNCH(false, false);
  }
#undef BP_PL_FB_LAUNCH
  return fused;
}

#ifdef PL_PROF
extern "C" int bp_debug_pl_prof(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pl_prof), sizeof(g_pl_prof));
}
#endif

}  // namespace bp
