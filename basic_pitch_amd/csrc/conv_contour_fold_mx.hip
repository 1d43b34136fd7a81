// Folded contour conv1 with the correction products on the block-scaled fp8 matrix instruction (the default folded
// kernel; BP_CONV1=f16 selects contour_conv1_folded_kernel of conv_contour_direct.hip, all three products in f16).
//
//   basic_pitch/nn.py:69-88 + basic_pitch/models.py:241-250 for the interior of the stack (bins 20..243): the same
//   operator, mapping and z-row image as contour_conv1_folded_kernel —
//       C[(j, o)][position] = Kt[(j, o)][tap'] x Z[tap'][position],  Z[tap'][m] = z[4 m + tap' - 56],
//   36 k-steps of 16 taps — but the split-precision product  w a = hi_w hi_a + lo_w a + hi_w lo_a  is issued as
//       hi_w hi_a          on v_mfma_f32_32x32x16_f16 (36 per tile, as before), and
//       lo_w a + hi_w lo_a on v_mfma_scale_f32_32x32x64_f8f6f4 (18 per tile instead of 72 f16 ones):
//   the two corrections need ~4 significant bits (they are <= 2^-11 of the product), and one block-scaled instruction
//   takes both for 32 taps — K block 0 = fp8(lo_w) x fp8(a), K block 1 = fp8(hi_w) x fp8(lo_a), E8M0 block scales put
//   them on the main product's scale, everything accumulates into ONE fp32 accumulator: 2304 instead of 3456 matrix
//   pipe cycles per tile.  Numerics: the contour map moves by 1.1e-5 against the fp64 oracle (stage test, 2e-5 bound).
//   * operand layout of the instruction (established with tools/ubench/mfma_mx.hip): byte j of lane (i, kh) is
//     K = 32 (j >> 4) + 16 kh + (j & 15); a lane's E8M0 scale applies to K block kh.  One scale per accumulator row for
//     all steps (e4m3 is a floating format; a row's taps span far fewer than its 15 binades of normals).
//   * LDS (115 KB, one workgroup per CU): z rows as two f16 hi copies (aligned ds_read_b128 at a lane stride of 4
//     bins) and, in their own array, four fp8 planes (fp8(a) and fp8(lo_a), each twice, the second copy shifted by 4
//     bins): a lane reads 16 consecutive bins of a plane as two ds_read_b64 — kept apart with opaque offsets, the merged
//     ds_read2_b64 runs at half the rate on 32 banks.  Strides enumerated against the service groups: 0 conflict cycles.
//     The f16 A fragments are read from LDS (36.9 KB); the 18 fp8 A fragments (32 bytes per lane each = 144 registers)
//     stay in registers for the whole kernel.  The LDS reads of block q + 1 are issued BETWEEN the three (dependent) matrix
//     instructions of block q: a wave issues about one LDS read per 14 cycles, in order with everything else, and reads
//     in front of a block only overlap its last instruction (233 VGPRs, 0 spills).
//   * a lane converts z to fp8 with v_cvt_scalef32_pk_fp8_f16 straight from the packed f16 pairs of `zp`.
// Roofline: matrix issue.  Per 32-position tile 36 f16 + 18 block-scaled instructions; bytes as the folded kernel.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bp_common.h"

namespace bp {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using i32x8 = __attribute__((ext_vector_type(8))) int;
using s16x2 = __attribute__((ext_vector_type(2))) short;
using h16x2 = __attribute__((ext_vector_type(2))) _Float16;

constexpr int kFxThreads = 512;
constexpr int kFxSteps = 36;             // f16 k-steps of 16 taps
constexpr int kFxMx = 18;                // block-scaled steps of 32 taps
constexpr int kFxPf = 1;                 // operand blocks read ahead (2, with a third of the fp8 A fragments in LDS to
                                         // pay for the buffers: no change — the kernel is not waiting for these reads)
constexpr int kFxGroups = 56;            // groups 5..60
constexpr int kFxFirstGroup = 5;
constexpr int kFxRing = 16;
constexpr int kFxCopy = 73;              // units per f16 hi copy (as the folded kernel)
constexpr int kFxRowU4 = 148;            // f16 row stride in 16-byte units (= 4 mod 16 like the folded kernel's 292)
// fp8 rows live in their own array: ds_read_b64 serves lanes 0-31 / 32-63 in one cycle each over 32 8-byte slots.  A tile
// starts at an odd group, its 16 even-group lanes take slots k + 1 .. k + 16, so the odd-group copy must sit 17 slots
// further (648 = 8 (17 + 64)), and a tile that wraps into the next row stays conflict-free for row strides = 96 mod 128
// (enumerated over all tile starts, like the f16 strides; 640 / 4928 cost one extra LDS cycle on every read)
constexpr int kFx8Row = 2272;            // fp8 row stride in bytes
constexpr int kFxF8Copy = 648;           // bytes between the two copies of an fp8 plane
constexpr int kFxF8Plane = 1104;         // bytes between the fp8(a) and the fp8(lo_a) planes
constexpr int kFxRound = 256;
constexpr int kFxPut0 = 6, kFxPut1 = 9;  // blocks after which the two staged tasks of a thread are converted and written
constexpr int kFxSA = 6;                 // fp8(a) = a * 2^6, fp8(lo_a) = (a - hi) * 2^11 * 2^6
static_assert(2 * kFxCopy <= kFxRowU4 && kFxF8Copy + 448 <= kFxF8Plane && kFxF8Plane + kFxF8Copy + 448 <= kFx8Row,
              "copies and planes do not overlap");
static_assert(kFxRing * (kFxRowU4 * 16 + kFx8Row) + kFxSteps * 64 * 16 <= 150 * 1024, "LDS budget");

#ifdef FX_PROF
__device__ unsigned long long fx_prof[8][8];  // tools only (build_variant.sh -DFX_PROF): phase clocks of workgroup 0
#define FX_T(k)                                                    \
  {                                                                \
    const unsigned long long t_now = __builtin_readcyclecounter(); \
    pt[k] += t_now - t_prev;                                       \
    t_prev = t_now;                                                \
  }
#else
#define FX_T(k)
#endif

struct FoldMxParams {
  const uint32_t* zp;    // [n][kZRowsP][kZRow]
  const uint4* a16;      // [36 steps][64 lanes] x (8 x f16): hi part of the Toeplitz-expanded folded kernel
  const uint4* amx;      // [18 steps][64 lanes][2] x 16 bytes: fp8(lo_w) | fp8(hi_w)
  const int* ascale;     // [64 lanes]: the E8M0 scale of the lane's K block (one per accumulator row for all steps)
  const float* bias;     // [8]
  float* c1;             // [n][172][kC1Row][8]
  int n_windows;
  int chunks;
};

__global__ __launch_bounds__(kFxThreads, 2) void contour_conv1_fold_mx_kernel(FoldMxParams p) {
  __shared__ __attribute__((aligned(16))) uint4 zimg[kFxRing * kFxRowU4];
  __shared__ __attribute__((aligned(16))) uint2 z8[kFxRing * kFx8Row / 8];
  __shared__ __attribute__((aligned(16))) uint4 afr[kFxSteps * 64];
  __shared__ __attribute__((aligned(16))) float bias_l[8];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = wave_id();
  const int kh = lane >> 5, li = lane & 31;

  for (int i = tid; i < kFxSteps * 64; i += kFxThreads) afr[i] = p.a16[i];
  for (int i = tid; i < kFxRing * kFxRowU4; i += kFxThreads) zimg[i] = uint4{0u, 0u, 0u, 0u};
  for (int i = tid; i < kFxRing * kFx8Row / 8; i += kFxThreads) z8[i] = uint2{0u, 0u};
  // resident fp8 A fragments (all 18 steps: 144 registers) and scales
  i32x8 amx[kFxMx];
#pragma unroll
  for (int S = 0; S < kFxMx; ++S) {
    const uint4 a0 = p.amx[(S * 64 + lane) * 2], a1 = p.amx[(S * 64 + lane) * 2 + 1];
    amx[S] = i32x8{(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
  }
  const int asc = p.ascale[lane];
  if (tid < 8) bias_l[tid] = p.bias[tid];
  const int sb = kh ? 127 - kFxSA - 11 : 127 - kFxSA;

  // one staging task = 4 consecutive zp words (row elements 4 u .. 4 u + 3): 8 bytes of f16 hi and 4 bytes of each fp8
  // plane, written to copy 0 at element 4 u and to copy 1 (shifted by 4 elements) at element 4 u - 4
  constexpr int kTasksRow = kZRow / 4;  // 112
  auto stage_load = [&](const uint32_t* __restrict__ zwin, int row, int u) {
    // 32-bit unsigned offset from the (uniform) window base: an SGPR base + one VGPR instead of a 64-bit lane address
    return *reinterpret_cast<const uint4*>(zwin + (unsigned)((row + 1) * kZRow + 4 * u));
  };
  auto stage_put = [&](const uint4 wv, int row, int u) {
    uint2 h2;
    h2.x = (wv.x & 0xffffu) | (wv.y << 16);
    h2.y = (wv.z & 0xffffu) | (wv.w << 16);
    // fp8(a 2^6), fp8(lo_a 2^6) straight from the packed f16 pairs (v_cvt_scalef32_pk_fp8_f16 divides by its scale)
    const uint32_t l2x = (wv.x >> 16) | (wv.y & 0xffff0000u), l2y = (wv.z >> 16) | (wv.w & 0xffff0000u);
    auto cvt2 = [&](uint32_t lo_pair, uint32_t hi_pair) -> uint32_t {
      s16x2 r = {0, 0};
      r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, __builtin_bit_cast(h16x2, lo_pair), (1.0f / (float)(1 << kFxSA)), false);
      r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, __builtin_bit_cast(h16x2, hi_pair), (1.0f / (float)(1 << kFxSA)), true);
      return __builtin_bit_cast(uint32_t, r);
    };
    const uint32_t a8 = cvt2(h2.x, h2.y), l8 = cvt2(l2x, l2y);
    uint4* rowu = zimg + ((row + kFxRing) % kFxRing) * kFxRowU4;
    uint2* rowp = reinterpret_cast<uint2*>(rowu);
    uint32_t* row8 = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(z8) + ((row + kFxRing) % kFxRing) * kFx8Row);
    rowp[u] = h2;  // hi copy 0
    row8[u] = a8;
    row8[kFxF8Plane / 4 + u] = l8;
    if (u > 0) {
      rowp[2 * kFxCopy + u - 1] = h2;  // hi copy 1
      row8[kFxF8Copy / 4 + u - 1] = a8;
      row8[(kFxF8Plane + kFxF8Copy) / 4 + u - 1] = l8;
    }
  };

#ifdef FX_PROF
  unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_prev = __builtin_readcyclecounter();
#endif
  const int rows_per = (kFrames + p.chunks - 1) / p.chunks;
  const int n_items = p.n_windows * p.chunks;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int b = item / p.chunks;
    const int t0 = (item - b * p.chunks) * rows_per;
    const int t1 = t0 + rows_per < kFrames ? t0 + rows_per : kFrames;
    const int npos = (t1 - t0) * kFxGroups;
    const int nrounds = (npos + kFxRound - 1) / kFxRound;
    const uint32_t* zwin = p.zp + (int64_t)b * kZWin;
    float* c1b = p.c1 + (int64_t)b * kC1Win;

    lds_barrier();
    int staged_hi = t0 + (kFxRound - 1) / kFxGroups + 1;
    staged_hi = staged_hi < t1 ? staged_hi : t1;
    for (int e = tid; e < (staged_hi - t0 + 2) * kTasksRow; e += kFxThreads) {
      const int ri = e / kTasksRow;
      stage_put(stage_load(zwin, t0 - 1 + ri, e - ri * kTasksRow), t0 - 1 + ri, e - ri * kTasksRow);
    }
    lds_barrier();

    FX_T(0);  // item prologue: first rows staged
    for (int k = 0; k < nrounds; ++k) {
      int need_hi = t0 + (kFxRound * (k + 1) + kFxRound - 1) / kFxGroups + 1;
      need_hi = need_hi < t1 ? need_hi : t1;
      const int n_new = (k + 1 < nrounds) ? need_hi - staged_hi : 0;
      const int first_new = staged_hi + 1;

      const int pos = kFxRound * k + 32 * w + li;
      const bool pvalid = pos < npos;
      const int posc = pvalid ? pos : npos - 1;
      const int prr = posc / kFxGroups;
      const int pgrp = kFxFirstGroup + posc - prr * kFxGroups;
      const int prow = t0 + prr;
      const int cpy = pgrp & 1;
      // f16 B: z elements 4 m + 16 s + 8 kh .. + 7 of hi copy (m & 1): uint4 index (m - copy) / 2 + kh + 2 s
      const int boff = cpy * kFxCopy + ((pgrp - cpy) >> 1) + kh;
      // fp8 B: z elements 4 m + 32 e + 16 kh .. + 15 of copy (m & 1): byte 4 (m - copy) + 32 e + 16 kh
      const int f8off = cpy * kFxF8Copy + 4 * (pgrp - cpy) + 16 * kh;
      int rowb[3], row8[3];  // byte offsets of the lane's first f16 / fp8 B operand in the three rows
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) {
        const int r = (prow - 1 + dt + kFxRing) % kFxRing;
        rowb[dt] = (r * kFxRowU4 + boff) * 16;
        row8[dt] = r * kFx8Row + f8off;
      }
      const char* zbytes = reinterpret_cast<const char*>(zimg);
      const char* z8bytes = reinterpret_cast<const char*>(z8);

      // the accumulator starts at the bias (a lane holds channels 4 kh .. 4 kh + 3 of four pixels)
      f32x16 acc;
      {
        const float4 bv = *reinterpret_cast<const float4*>(bias_l + 4 * kh);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[4 * j + 0] = bv.x, acc[4 * j + 1] = bv.y, acc[4 * j + 2] = bv.z, acc[4 * j + 3] = bv.w;
      }

      // operands of block q = (dt, e): two f16 k-steps + one block-scaled step, read one block ahead
      constexpr int kBuf = kFxPf + 1;
      f16x8 ah[kBuf][2], bh[kBuf][2];
      uint2 b8[kBuf][4];
      uint4 st_w[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
      // A wave issues about one LDS read per 14 cycles, in order with everything else: the 8 reads of a block in front of
      // its three (dependent) matrix instructions cost 112 cycles of which only the last instruction's 64 overlap.  So the
      // reads of block q + 1 go BETWEEN the matrix instructions of block q, in three pieces.
      auto issue = [&](int q, int piece) {
        const int buf = q % kBuf;
        const int dt = q / 6, e = q - 6 * dt;
        if (piece < 2) {
          const int i = piece;
          bh[buf][i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(zbytes + rowb[dt] + 32 * (2 * e + i)));
          ah[buf][i] = __builtin_bit_cast(f16x8, afr[(dt * 12 + 2 * e + i) * 64 + lane]);
          return;
        }
        // four separate ds_read_b64 (2 LDS cycles each, 64 banks): merged into ds_read2_b64 they take 8 cycles a pair on
        // 32 banks, where the two copies collide — the opaque offsets keep the merge pass off them
        const int o0 = row8[dt];
        int o[4] = {o0 + 32 * e, o0 + 32 * e + 8, o0 + 32 * e + kFxF8Plane, o0 + 32 * e + kFxF8Plane + 8};
#pragma unroll
        for (int i = 1; i < 4; ++i) asm volatile("" : "+v"(o[i]));
#pragma unroll
        for (int i = 0; i < 4; ++i) b8[buf][i] = *reinterpret_cast<const uint2*>(z8bytes + o[i]);
      };
      FX_T(1);  // round prologue: positions, addresses, bias
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) issue(0, pc);
#pragma unroll
      for (int q = 0; q < kFxMx; ++q) {
        const int buf = q % kBuf;
        const bool more = q + 1 < kFxMx;
        // the z rows of the next round: at most 6 rows x 112 tasks, two per thread.  Both loads leave at the start of the
        // tile and are converted and written later: zp comes from HBM / the Infinity Cache, and (gfx9 vmcnt counts
        // stores too) the wait also covers the previous tile's c1 stores
        if (q == 0) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int e = i * kFxThreads + tid;
            if (e < n_new * kTasksRow) st_w[i] = stage_load(zwin, first_new + e / kTasksRow, e % kTasksRow);
          }
        }
        if (q == kFxPut0 || q == kFxPut1) {
          FX_T(2);  // matrix blocks
          const int i = q == kFxPut0 ? 0 : 1;
          const int e = i * kFxThreads + tid;
          if (e < n_new * kTasksRow) stage_put(st_w[i], first_new + e / kTasksRow, e % kTasksRow);
          FX_T(3);  // staging puts (incl. the wait for their loads)
        }
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[buf][0], bh[buf][0], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) issue(q + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[buf][1], bh[buf][1], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) issue(q + 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        const i32x8 bm = {(int)b8[buf][0].x, (int)b8[buf][0].y, (int)b8[buf][1].x, (int)b8[buf][1].y,
                          (int)b8[buf][2].x, (int)b8[buf][2].y, (int)b8[buf][3].x, (int)b8[buf][3].y};
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(amx[q], bm, acc, 0, 0, 0, asc, 0, sb);
        __builtin_amdgcn_sched_barrier(0);
        if (more) issue(q + 1, 2);
        __builtin_amdgcn_sched_barrier(0);
      }
      staged_hi += n_new;
      FX_T(2);
      // the accumulator is pinned here: otherwise the compiler sinks the last matrix instruction into the divergent
      // store branch below, TOGETHER with the scratch reload of its (lane-private) A fragment — the lanes without a
      // valid position then skip the reload and feed stale rows of A into every lane's result
      asm volatile("" : "+v"(acc));
      if (pvalid) {
        float* dst = c1b + (unsigned)((prow * kC1Row + kC1Pad + 4 * pgrp) * 8 + 4 * kh);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float4 v;
          v.x = relu_f32(acc[4 * j + 0]);
          v.y = relu_f32(acc[4 * j + 1]);
          v.z = relu_f32(acc[4 * j + 2]);
          v.w = relu_f32(acc[4 * j + 3]);
          *reinterpret_cast<float4*>(dst + j * 8) = v;
        }
      }
      FX_T(4);  // last MFMA's latency + ReLU + stores
      lds_barrier();
      FX_T(5);  // barrier wait
    }
  }
#ifdef FX_PROF
  if (blockIdx.x == 0 && lane == 0)
    for (int k = 0; k < 8; ++k) fx_prof[w][k] = pt[k];
#endif
}

void launch_contour_conv1_fold_mx(const uint32_t* zp, const void* a16, const void* amx, const void* ascale,
                                  const float* bias, float* c1, int n_windows, int n_cu, hipStream_t stream) {
  int chunks = 1;
  while (chunks < 4 && n_windows * chunks < n_cu) chunks *= 2;
  FoldMxParams p{zp, static_cast<const uint4*>(a16), static_cast<const uint4*>(amx), static_cast<const int*>(ascale),
                 bias, c1, n_windows, chunks};
  const int items = n_windows * chunks;
  if (items <= 0) return;
  const int grid = items < n_cu ? items : n_cu;
  hipLaunchKernelGGL(contour_conv1_fold_mx_kernel, dim3(grid), dim3(kFxThreads), 0, stream, p);
#ifdef FX_PROF
  static int calls = 0;
  if (++calls == 20) {
    unsigned long long h[8][8];
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(fx_prof), sizeof h);
    for (int w = 0; w < 8; ++w) {
      fprintf(stderr, "fxprof wave %d:", w);
      for (int k = 0; k < 6; ++k) fprintf(stderr, " %llu", h[w][k]);
      fprintf(stderr, "\n");
    }
  }
#endif
}

}  // namespace bp
