// FLAC decode on the device (round 6; SURVEY.md 8(f) rank 2: "WAV/FLAC decode").  The container half of
// `librosa.load(path, sr=22050, mono=True)` (basic_pitch/inference.py:239; README.md:182-189 lists .flac) for the file job:
// the host decoder (flac_decode.cpp) sustains ~80 M samples per second and core — five three-minute stereo files per second
// and core against the ~1,400 a GPU transcribes — and FLAC halves the bytes a file costs on the storage device, in host
// DRAM and on PCIe, which are what an 8-GPU file job runs out of first.
//
// The format (RFC 9639) is serial inside a frame — a Rice code's position is the sum of the lengths of all codes before
// it, a subframe starts where the previous one ends, linear prediction is a recurrence — and independent from frame to
// frame, each frame beginning on a byte boundary with a sync code and a CRC-8-protected header that carries its own
// position in the stream.  So the FILE's bytes go to the device as they are, and three launches decode them:
//   1. flac_scan_kernel     every byte position that looks like a frame header (sync code, no reserved value, sample size
//                           and channel count of STREAMINFO, CRC-8 right) becomes a candidate (offset, coded number,
//                           block size), kept in file order (a workgroup owns 64 KB of the file and a slice of the list);
//   2. flac_chain_kernel    one workgroup compacts the candidates in file order and keeps those that continue their
//                           predecessor or are continued by their successor (coded number + 1 / sample number + block
//                           size): a sync pattern inside compressed data passes the CRC-8 once in ~10^7 bytes and the number
//                           check practically never — and if it did, the chain as a whole or the frame's CRC-16 fails and
//                           the call reports the file as not decodable here;
//   3. flac_decode_kernel   A LANE TRIO PER FRAME (three waves per 64 frames): the parser — subframe headers, Rice / escaped
//                           residuals from a 32-bit funnel-shift window over a ring of the lane's stream in LDS (the lanes
//                           of a wave load TOGETHER, at a service every 16 codes: see FdBits) —, the restorer — prediction
//                           (constant, verbatim, fixed order 0..4, LPC order 1..12 as exact float64 FMAs on a register
//                           history, 13..32 with 64-bit sums on an LDS ring), wasted bits, the channels as coded to scratch
//                           rows —, fed through a mailbox in LDS, and the checker — the frame's CRC-16 (eight bytes per
//                           step, tables in LDS);
//   4. flac_finalize_kernel the parallel tail, a thread per sample: stereo decorrelation and the interleaved 16- or 32-bit
//                           PCM the ingest kernels read (audio_ingest.hip downmix_raw_kernel) — bit for bit what the host
//                           decoder produces.  A three-minute stereo file is ~1,940 frames = 31 waves; the serial decode of
//                           a frame sets the latency of the call (~1 - 2 ms), not the throughput of the job, whose lanes
//                           keep several files in flight.
// Not decoded here (status BP_FLACDEV_UNSUPPORTED, the caller uses the host decoder): streams without a sample count or
// block sizes in STREAMINFO, more than 24 bits per sample, more than 8 channels.  The MD5 of STREAMINFO is not checked on
// the device (one serial pass over the whole stream); every frame's CRC-16 and the stream's sample count are.
#include <stdint.h>
#include <stdio.h>

#include "bp_common.h"

namespace bp {

enum : int {
  kFdOk = 0,
  kFdUnsupported = 1,   // a feature the device decoder leaves to the host
  kFdChain = 2,         // frames missing / out of order / sample count differs from STREAMINFO
  kFdCrc16 = 4,         // a frame's CRC-16 does not match
  kFdParse = 8,         // reserved value, overrun or inconsistent subframe
  kFdOverflow = 16,     // more candidates in a 64 KB chunk than the list holds
};

constexpr int kFdChunk = 65536;      // bytes of the file a scan workgroup owns
constexpr int kFdChunkCands = 512;   // candidates a chunk may hold (a frame is >= ~14 bytes; real streams: a handful)

struct FdCand {
  uint32_t offset;     // of the sync code
  uint32_t blocksize;
  uint64_t number;     // coded frame number (fixed block size) or sample number (variable)
  uint32_t hdr_bytes;  // header length including the CRC-8
  uint32_t flags;      // bit 0: variable block size; bits 4..7: channel assignment code
};

struct FdFrame {
  uint32_t offset, end;  // the frame's bytes: [offset, end) (end = the next frame's offset or the file's end)
  uint32_t blocksize, hdr_bytes;
  int64_t first_sample;
  uint32_t ch_code, pad;
};

struct FdStream {
  int channels, bits, min_block, max_block;
  int64_t total;       // samples per channel
  uint32_t audio_start, nbytes;
};

__device__ __forceinline__ uint8_t fd_crc8(const uint8_t* d, int n) {
  uint32_t c = 0;
  for (int i = 0; i < n; ++i) {
    c ^= d[i];
    for (int b = 0; b < 8; ++b) c = (c & 0x80) ? ((c << 1) ^ 0x07) & 0xff : (c << 1) & 0xff;
  }
  return (uint8_t)c;
}

// A frame header at d[0..] (at least 16 readable bytes)?  Fills the candidate; RFC 9639 section 9.1.
__device__ bool fd_parse_header(const uint8_t* d, const FdStream& st, FdCand& c) {
  if (d[0] != 0xff || (d[1] & 0xfe) != 0xf8) return false;
  const int variable = d[1] & 1;
  const int bs_code = d[2] >> 4, sr_code = d[2] & 15, ch_code = d[3] >> 4, sz_code = (d[3] >> 1) & 7;
  if ((d[3] & 1) || bs_code == 0 || sr_code == 15 || ch_code > 10 || sz_code == 3) return false;
  int p = 4;
  const int lead = d[p++];
  uint64_t number = 0;
  if (lead & 0x80) {
    int extra = 0;
    while (extra < 7 && (lead & (0x40 >> extra))) ++extra;
    if (extra == 0 || extra > 6) return false;
    number = lead & (0x3f >> extra);
    for (int i = 0; i < extra; ++i) {
      if ((d[p] & 0xc0) != 0x80) return false;
      number = (number << 6) | (d[p++] & 0x3f);
    }
  } else {
    number = (uint64_t)lead;
  }
  int blocksize;
  if (bs_code == 1) blocksize = 192;
  else if (bs_code <= 5) blocksize = 576 << (bs_code - 2);
  else if (bs_code == 6) blocksize = d[p++] + 1;
  else if (bs_code == 7) { blocksize = ((d[p] << 8) | d[p + 1]) + 1; p += 2; }
  else blocksize = 256 << (bs_code - 8);
  if (sr_code == 12) p += 1;
  else if (sr_code == 13 || sr_code == 14) p += 2;
  const int sz_table[8] = {0, 8, 12, 0, 16, 20, 24, 32};
  const int bits = sz_code ? sz_table[sz_code] : st.bits;
  const int n_ch = ch_code < 8 ? ch_code + 1 : 2;
  if (bits != st.bits || n_ch != st.channels) return false;
  if (blocksize > st.max_block) return false;
  if (fd_crc8(d, p) != d[p]) return false;
  c.blocksize = (uint32_t)blocksize;
  c.number = number;
  c.hdr_bytes = (uint32_t)(p + 1);
  c.flags = (uint32_t)variable | ((uint32_t)ch_code << 4);
  return true;
}

// ---- 1. candidates, in file order ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void flac_scan_kernel(const uint8_t* __restrict__ file, FdStream st, FdCand* __restrict__ cands,
                                                        uint32_t* __restrict__ counts, int* __restrict__ status) {
  __shared__ FdCand found[kFdChunkCands];
  __shared__ uint32_t n_found;
  if (threadIdx.x == 0) n_found = 0;
  __syncthreads();
  const uint32_t chunk0 = st.audio_start + blockIdx.x * (uint32_t)kFdChunk;
  // 16 bytes per lane and trip, a wave's lanes on consecutive pieces (the first version walked 256 bytes per thread with byte
  // loads: 87 us for a 22 MB file); a byte 0xff is found in the registers (the zero-byte test on the complement, each hit
  // checked), and only there is a header parsed from memory.  The file's buffer is padded with zeros: a piece or a header
  // read may run up to 16 bytes past the end.
  constexpr int kTrips = kFdChunk / (256 * 16);
  typedef uint32_t Piece __attribute__((ext_vector_type(4)));
  Piece pc[kTrips];  // all of the thread's pieces asked for at once: one memory round trip per workgroup
#pragma unroll
  for (int i = 0; i < kTrips; ++i) {
    const uint32_t piece = chunk0 + (uint32_t)(i * 256 + threadIdx.x) * 16;
    const uint32_t from = piece + 2 <= st.nbytes ? piece : 0u;  // (outside the file: any bytes of it; the piece is skipped)
    __builtin_memcpy(&pc[i], file + from, 16);
  }
#pragma unroll
  for (int i = 0; i < kTrips; ++i) {
    const uint32_t piece = chunk0 + (uint32_t)(i * 256 + threadIdx.x) * 16;
    if (piece + 2 > st.nbytes) continue;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t wd = pc[i][d];
      const uint32_t x = ~wd;
      uint32_t z = (x - 0x01010101u) & wd & 0x80808080u;  // candidates for bytes of wd that are 0xff
      while (z) {
        const uint32_t b = (uint32_t)__builtin_ctz(z) >> 3;
        z &= z - 1;
        const uint32_t pos = piece + 4 * d + b;
        if (((wd >> (8 * b)) & 0xffu) != 0xffu || pos + 2 > st.nbytes) continue;
        // the sync code's second byte (1111100x) where it is in the registers too: one 0xff in 128 gets to the header parse,
        // whose dependent byte loads are what this kernel's time is made of
        if (b < 3 || d < 3) {
          const uint32_t nb = b < 3 ? wd >> (8 * (b + 1)) : pc[i][d < 3 ? d + 1 : 3];
          if ((nb & 0xfeu) != 0xf8u) continue;
        }
        FdCand c;
        if (!fd_parse_header(file + pos, st, c)) continue;
        c.offset = pos;
        const uint32_t slot = atomicAdd(&n_found, 1u);
        if (slot < kFdChunkCands) found[slot] = c;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t n = n_found;
    if (n > kFdChunkCands) {
      atomicOr(status, kFdOverflow);
      n = kFdChunkCands;
    }
    for (uint32_t i = 1; i < n; ++i) {  // a handful: insertion sort by offset
      const FdCand c = found[i];
      uint32_t j = i;
      for (; j > 0 && found[j - 1].offset > c.offset; --j) found[j] = found[j - 1];
      found[j] = c;
    }
    counts[blockIdx.x] = n;
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n_found && i < kFdChunkCands; i += 256) cands[(size_t)blockIdx.x * kFdChunkCands + i] = found[i];
}

// ---- 2. the chain of real frames ---------------------------------------------------------------------------------------------
// One workgroup, everything parallel (the first version — one lane walking the candidates through global memory — took 1.2 ms
// of a 3-minute file's 4 ms): offsets of the chunks' slices by a block scan, the candidates compacted in file order, a
// candidate kept iff it continues its predecessor or is continued by its successor (coded number + 1, or sample number +
// block size: a false sync code passes the CRC-8 once in ~10^7 bytes and then carries an arbitrary number), the kept ones
// compacted into frames, and the chain checked as a whole: frame k starts at sample k x block size (or where frame k - 1
// ended), the first at 0, the last reaches STREAMINFO's count.  Anything else is kFdChain: the host decoder takes the file.
constexpr int kFdChainThreads = 1024;

__device__ __forceinline__ uint32_t fd_block_scan(uint32_t v, uint32_t* lds, uint32_t* total) {  // exclusive, 1024 threads
  const int t = threadIdx.x;
  lds[t] = v;
  __syncthreads();
  for (int d = 1; d < kFdChainThreads; d <<= 1) {
    const uint32_t x = t >= d ? lds[t - d] : 0u;
    __syncthreads();
    lds[t] += x;
    __syncthreads();
  }
  const uint32_t incl = lds[t];
  *total = lds[kFdChainThreads - 1];
  __syncthreads();
  return incl - v;
}

__global__ __launch_bounds__(kFdChainThreads) void flac_chain_kernel(const FdCand* __restrict__ cands, const uint32_t* __restrict__ counts,
                                                                     int n_chunks, FdStream st, FdCand* __restrict__ packed,
                                                                     uint32_t* __restrict__ offs, FdFrame* __restrict__ frames,
                                                                     int max_frames, int* __restrict__ n_frames,
                                                                     int* __restrict__ status) {
  __shared__ uint32_t lds[kFdChainThreads];
  __shared__ uint32_t carry_s;
  const int t = threadIdx.x;
  // (a) offsets of the chunks' slices
  uint32_t carry = 0;
  for (int base = 0; base < n_chunks; base += kFdChainThreads) {
    const int ch = base + t;
    const uint32_t c = ch < n_chunks ? counts[ch] : 0u;
    uint32_t tot;
    const uint32_t ex = fd_block_scan(c, lds, &tot);
    if (ch < n_chunks) offs[ch] = carry + ex;
    carry += tot;
  }
  const uint32_t n_cand = carry;
  // (b) candidates in file order
  for (int ch = t / 32; ch < n_chunks; ch += kFdChainThreads / 32) {
    const uint32_t c = counts[ch], o = offs[ch];
    for (uint32_t i = t & 31; i < c; i += 32) packed[o + i] = cands[(size_t)ch * kFdChunkCands + i];
  }
  __threadfence_block();
  __syncthreads();
  const uint32_t variable = n_cand ? (packed[0].flags & 1u) : 0u;
  auto follows = [&](const FdCand& a, const FdCand& b) {  // b is the frame right behind a
    if ((a.flags & 1u) != variable || (b.flags & 1u) != variable) return false;
    return variable ? b.number == a.number + a.blocksize : b.number == a.number + 1;
  };
  // (c) + (d) kept candidates -> frames
  carry = 0;
  for (uint32_t base = 0; base < n_cand; base += kFdChainThreads) {
    const uint32_t j = base + t;
    bool good = false;
    FdCand c{};
    if (j < n_cand) {
      c = packed[j];
      good = (j > 0 && follows(packed[j - 1], c)) || (j + 1 < n_cand && follows(c, packed[j + 1])) || n_cand == 1;
    }
    uint32_t tot;
    const uint32_t idx = carry + fd_block_scan(good ? 1u : 0u, lds, &tot);
    if (good && (int)idx < max_frames) {
      FdFrame f;
      f.offset = c.offset, f.end = st.nbytes, f.blocksize = c.blocksize, f.hdr_bytes = c.hdr_bytes;
      f.first_sample = variable ? (int64_t)c.number : (int64_t)c.number * (int64_t)packed[0].blocksize;
      f.ch_code = c.flags >> 4, f.pad = 0;
      frames[idx] = f;
    }
    carry += tot;
  }
  const uint32_t n = carry;
  __threadfence_block();
  __syncthreads();
  if (t == 0) {
    carry_s = 0;
    *n_frames = (int)(n < (uint32_t)max_frames ? n : (uint32_t)max_frames);
  }
  __syncthreads();
  // (e) a frame ends where the next begins; the chain as a whole
  bool bad = n == 0 || (int)n > max_frames;
  for (uint32_t k = t; k < n && (int)k < max_frames; k += kFdChainThreads) {
    const FdFrame f = frames[k];
    if (k + 1 < n) {
      const FdFrame g = frames[k + 1];
      frames[k].end = g.offset;
      if (f.first_sample + (int64_t)f.blocksize != g.first_sample) bad = true;
    } else if (f.first_sample + (int64_t)f.blocksize < st.total) {
      bad = true;
    }
    if (k == 0 && f.first_sample != 0) bad = true;
  }
  if (bad) atomicOr(status, kFdChain);
}

// ---- 3. one lane per frame -----------------------------------------------------------------------------------------------------
// The stream as 32-bit words: `hi` and `lo` hold the next 64 bits, the window starts `s` bits above the bottom of `hi`, `nx`
// and `n2` are the words behind them, read from the lane's ring in LDS two refills ahead of their use.  A peek is one funnel
// shift, a skip a subtraction and — every 32 bits — a rotation of the words; no 64-bit shifts on the serial path.
#ifndef FD_LANES  // tools: 32 or 16 frames per wave are SLOWER (1.42 / 1.38 ms against 1.26: the vector pipe does not skip the
#define FD_LANES 64  // passes of inactive lanes, and the waves crowd fewer CUs)
#endif
constexpr int kFdLanes = FD_LANES;  // lanes (frames) per workgroup: one wave
constexpr int kFdRing = 128;     // words of its stream a lane holds in LDS
constexpr int kFdRingRow = 132;  // row stride in words: 16-byte rows for ds_write_b128, the lanes' equal indices on four banks
constexpr int kFdBurst = 16;     // codes between two services
constexpr int kFdSlotRow = 20;   // kFdBurst values + padding (16-byte rows)
constexpr int kFdSlots = 8;      // bursts a lane's parser may be ahead of its restorer

// A frame is decoded by TWO lanes of the same number in two waves of a workgroup: the PARSER walks the bit stream (subframe
// headers, Rice codes -> residuals), the RESTORER runs the prediction and stores the samples.  The two chains — the bit
// position, the prediction history — share nothing, and a wave that is alone on its SIMD issues one instruction per ~8
// cycles on a dependent chain: side by side in one wave they add up (the compiler's schedule does not interleave them,
// profiles/r06_flac_device.md), in two waves they overlap.  They talk through a mailbox in LDS, per lane: kFdSlots slots of a
// burst (<= 16 values + a descriptor word), a produced and a consumed counter (release / acquire at workgroup scope).
enum : uint32_t {
  kFdMsgResidual = 0,  // n residuals of the current subframe
  kFdMsgSamples = 1,   // n samples as coded (verbatim subframe)
  kFdMsgSubframe = 2,  // a subframe begins: order / shift / wasted bits / channel in the lane's descriptor, coefficients and
                       // warm-up samples in `coefs` / `hist`
  kFdMsgConstant = 3,  // slot[0] = the value of a constant subframe
  kFdMsgEnd = 4,       // the frame is parsed (or given up)
  kFdMsgCodes = 5,     // a full burst of Rice codes as the parser saw them: the 16 windows of 32 bits they start in, the parameter
                       // in bits 16.. of the descriptor — the restorer finds each code's length again and cuts the value out:
                       // the parser's chain is the bit position alone (811 us with the values composed by the parser, 795 so)
};
constexpr uint32_t kFdSpinCap = 1u << 24;  // reads of a counter before a wave gives its partner up (a bug, not a stream)

// The parser's view of its frame: the bit window and the words behind it.
//
// What bounds it is neither arithmetic nor bandwidth but the latency of the lanes' own memory operations, and the fact that
// a wave has ONE counter for them (vmcnt).  A lane needs its next word every ~5 codes, but SOME lane of the 64 needs one at
// nearly every code: with a load per refill the wave sat out an L2 / HBM round trip per code (measured: 490 core cycles per
// Rice code with prediction and stores compiled out, SQ_WAIT_ANY 60 - 68 % of the wave's cycles, 1.2 load instructions per
// code and wave).  So the lanes touch global memory together, at a SERVICE every kFdBurst codes:
//   * the stream lives in a ring of kFdRing words per lane in LDS; a refill of the window is a ds_read (its own counter,
//     ~64 cycles, asked for two refills ahead);
//   * a service commits the <= 4 blocks of 16 bytes it asked for at the PREVIOUS service (the one wait: everything in
//     flight is a burst old) and asks for the next 4.  Invariant: a lane that consumes <= 16 words per burst (a code of
//     the fast path is <= 32 bits) has >= 32 words committed after every service (c' = c - u + 16 while c < 112; >= 109 -
//     16 above), the header fields of a subframe (<= 34 + 16 words) sit between two double services (refuel()), a code
//     longer than the window (unary runs of hundreds of zeros: the test-side encoder writes them) serves itself every four
//     words and is followed by a refuel.  Should a lane run dry anyway it reads stale words: memory-safe, the frame's
//     CRC-16 fails, the call reports the stream as not decodable here.
// The parser never stores to global memory (the restorer does): its one wait is for loads alone.
struct FdBits {
  const uint8_t* org;  // the byte the stream's word 0 starts on (frame offset + header length)
  uint32_t off0;       // its offset from the file's start
  uint32_t hi, lo;
  uint32_t nx, n2;     // the two words behind `lo` as read (little-endian): swapped when they move up, and read two refills
                       // ahead, so that a refill never waits for its own read
  int s;         // the window starts s bits above the bottom of `hi`: 0..31 (0 = all of `hi` consumed, the window is `lo`) —
                 // v_alignbit's own shift operand, so a peek is that one instruction whatever the position
  uint32_t* ring;      // LDS, this lane's kFdRing words
  uint32_t rd, wr;     // words read from / committed to the ring (stream word numbers): `lo` is word rd - 3
  uint32_t wmax;       // the last word a block may start on (inside the buffer's padding)
  typedef uint32_t Block __attribute__((ext_vector_type(4)));
  Block pb0, pb1, pb2, pb3;  // blocks in flight (named, not an array: they live in registers)
  int np;

  __device__ __forceinline__ Block get(uint32_t word) const {
    // never behind the 64 zero bytes that follow the file in its buffer: a lane that has lost a corrupt stream (a burst of
    // maximal unary runs is 16 KB) reads the padding again and again, its frame fails the position check or the CRC-16
    const uint32_t w = word < wmax ? word : wmax;
    Block b;
    __builtin_memcpy(&b, org + 4 * (size_t)w, 16);
    return b;
  }
  __device__ __forceinline__ void put(uint32_t word, Block b) { __builtin_memcpy(ring + (word & (kFdRing - 1)), &b, 16); }
  __device__ __forceinline__ void init(const uint8_t* file, uint32_t off, uint32_t nbytes, uint32_t* ring_row) {
    wmax = (nbytes + 48 - off) >> 2;  // off < nbytes: a frame starts inside the file
    org = file + off, off0 = off, s = 0, hi = 0, ring = ring_row, wr = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // 32 words committed before the first bit is read
      pb0 = get(wr), pb1 = get(wr + 4), pb2 = get(wr + 8), pb3 = get(wr + 12);
      put(wr, pb0), put(wr + 4, pb1), put(wr + 8, pb2), put(wr + 12, pb3);
      wr += 16;
    }
    pb0 = get(wr), pb1 = get(wr + 4), pb2 = get(wr + 8), pb3 = get(wr + 12);
    np = 4;
    lo = __builtin_bswap32(ring[0]), nx = ring[1], n2 = ring[2], rd = 3;
  }
  __device__ __forceinline__ void service() {
    if (np > 0) put(wr, pb0);
    if (np > 1) put(wr + 4, pb1);
    if (np > 2) put(wr + 8, pb2);
    if (np > 3) put(wr + 12, pb3);
    wr += 4 * (uint32_t)np;
    const int room = (kFdRing - 1 - (int)(wr - rd)) >> 2;  // word rd - 1 (n2) stays: skip_select() reads it again
    np = room < 4 ? room : 4;
    // all four asked for whatever the room: a load under a lane mask would make the compiler guard its target registers
    // with a wait of its own — behind the load issued just before; the blocks without room are asked for again next time
    pb0 = get(wr), pb1 = get(wr + 4), pb2 = get(wr + 8), pb3 = get(wr + 12);
  }
  __device__ __forceinline__ void refuel() {  // up to 32 more words committed at once (before a subframe's header fields)
    service();
    service();
  }
  __device__ __forceinline__ uint32_t peek() const { return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)s); }  // the next 32 bits
  __device__ __forceinline__ void skip(int n) {  // n <= 32
    s -= n;
    if (s < 0) {
      hi = lo, lo = __builtin_bswap32(nx), nx = n2;
      n2 = ring[rd & (kFdRing - 1)];
      ++rd, s += 32;
    }
  }
  // the same without a branch (n <= 32 for a valid step): the burst of sixteen codes is one basic block, its instructions
  // interleave; n2 is read again at every step (the same word until a refill moves rd)
  __device__ __forceinline__ void skip_select(int n) {
    s -= n;
    const bool need = s < 0;
    hi = need ? lo : hi;
    lo = need ? __builtin_bswap32(nx) : lo;
    nx = need ? n2 : nx;
    rd += need ? 1u : 0u;
    s &= 31;  // + 32 where it went below zero (>= -32 for a valid step)
#ifdef FD_NO_RING_READ  // tools only (wrong samples): what the ring read of every step costs the parser
    n2 ^= rd;
#else
    n2 = ring[(rd - 1) & (kFdRing - 1)];
#endif
  }
  __device__ __forceinline__ uint32_t at() const { return off0 + 4 * (rd - 3); }  // byte offset of the start of `lo`
  __device__ __forceinline__ uint32_t bits(int k) {  // k <= 32
    if (k == 0) return 0;
    const uint32_t v = peek() >> (32 - k);
    skip(k);
    return v;
  }
  __device__ __forceinline__ int32_t sbits(int k) {  // k <= 32
    if (k == 0) return 0;
    const int32_t v = (int32_t)peek() >> (32 - k);  // arithmetic: sign-extends
    skip(k);
    return v;
  }
  __device__ __forceinline__ uint32_t unary() {  // zeros before the next one
    uint32_t q = 0;
    for (;;) {
      const uint32_t w = peek();
      if (w) {
        const int lz = __builtin_clz(w);
        skip(lz + 1);
        return q + (uint32_t)lz;
      }
      skip(32);
      q += 32;
      if ((q & 127) == 0) service();  // a long run outlives the ring: four words at most between two services
      if (q > (1u << 13)) return q;  // a run no encoder writes (the zero padding behind the file, a misread stream): the caller's
                                     // position check ends the frame
    }
  }
  // one Rice code: the whole code inside the 32-bit window (every code of ordinary audio) is one peek; a longer one (the
  // ring's invariant counts 32 bits per code) refuels behind itself
  __device__ __forceinline__ int32_t rice(int k) {
    const uint32_t w = peek();
    const int n = (w ? __builtin_clz(w) : 32) + 1 + k;  // the code's length
    uint32_t v;
    if (__builtin_expect(n <= 32, 1)) {
      v = ((uint32_t)(n - 1 - k) << k) | __builtin_amdgcn_ubfe(w, (uint32_t)(32 - n), (uint32_t)k);
      skip(n);
    } else {
      const uint32_t q = unary();
      v = (q << k) | bits(k);
      refuel();
    }
    return (int32_t)(v >> 1) ^ -(int32_t)(v & 1);
  }
  // the same for a burst that checks afterwards: no branch for the long code, *nmax collects the lengths; a burst with one
  // beyond the window is decoded again from its start by rice() (what this stepped over then was garbage).  Returns the
  // window the code starts in: the value is cut out by the restorer (fd_rice_value).
  __device__ __forceinline__ uint32_t rice_window(int k, int* nmax) {
    const uint32_t w = peek();
    const int n = (w ? __builtin_clz(w) : 32) + 1 + k;
    *nmax = n > *nmax ? n : *nmax;
    skip_select(n);
    return w;
  }
  __device__ __forceinline__ uint32_t byte_pos() const { return at() - (uint32_t)((s + 7) >> 3); }  // of the next unread bit
  __device__ __forceinline__ uint32_t bytes_consumed_aligned() {  // after dropping the bits up to the next byte boundary
    if (s & 7) skip(s & 7);
    return byte_pos();
  }
};

// the residual of the Rice code with parameter k that starts at the top of window w and ends inside it (w != 0)
__device__ __forceinline__ int32_t fd_rice_value(uint32_t w, int k) {
  const int lz = __builtin_clz(w);
  const uint32_t v = ((uint32_t)lz << k) | __builtin_amdgcn_ubfe(w, (uint32_t)(31 - lz - k), (uint32_t)k);
  return (int32_t)(v >> 1) ^ -(int32_t)(v & 1);
}

// Linear prediction with the history in registers, as float64: a restored sample is an int32, a coefficient has <= 15 bits, a
// sum of <= 12 products stays below 2^51 — every operation is exact.  (The vector pipe runs v_fma_f64 at a fraction of the
// cost of the 64-bit integer multiply-adds it replaces, and the history moves down by register copies, no addressing.)  One
// set of 12 coefficients and 12 samples serves the orders 1..12 in three classes (4, 8, 12 products per sample, the unused
// coefficients zero); the orders above 12 take the generic path in the kernel (LDS ring, 64-bit integers).
struct FdPred {
  double c[12], h[12];  // h[0] = s[i - 1]
  int shift;            // 0..15 (a 5-bit signed field, negative refused)
  template <int ORD>
  __device__ __forceinline__ int32_t step(int32_t res) {
    // the sum rides on 1.5 * 2^52: |sum| < 2^51 is an integer, so the double's low 51 bits ARE its two's complement and
    // the arithmetic shift is a funnel shift of the two words (no multiply, floor or conversion)
    double a0 = 6755399441055744.0, a1 = 0.0;
    // oldest samples first: only the last product waits for the sample the previous step has just restored
#pragma unroll
    for (int j = ORD - 1; j >= 1; j -= 2) {
      a1 = __builtin_fma(c[j], h[j], a1);
      a0 = __builtin_fma(c[j - 1], h[j - 1], a0);
    }
    const uint64_t sb = __builtin_bit_cast(uint64_t, a0 + a1);
    const int32_t pred = (int32_t)__builtin_amdgcn_alignbit((uint32_t)(sb >> 32), (uint32_t)sb, (uint32_t)shift);
    const int32_t v = (int32_t)((uint32_t)res + (uint32_t)pred);
#pragma unroll
    for (int j = ORD - 1; j > 0; --j) h[j] = h[j - 1];
    h[0] = (double)v;
    return v;
  }
  // a full burst: sixteen residuals in registers become samples (wasted bits restored); written out sixteen times, the
  // history's moves are register names
  template <int ORD>
  __device__ __forceinline__ void burst(int32_t (&r)[kFdBurst], int wasted) {
#pragma unroll
    for (int t = 0; t < kFdBurst; ++t) r[t] = (int32_t)((uint32_t)step<ORD>(r[t]) << wasted);
  }
  template <int ORD>
  __device__ __forceinline__ void some(int32_t* v, int n, int wasted) {
    for (int t = 0; t < n; ++t) v[t] = (int32_t)((uint32_t)step<ORD>(v[t]) << wasted);
  }
};

struct FdDecodeParams {
  const uint8_t* file;
  const FdFrame* frames;
  const int* n_frames;
  FdStream st;
  int32_t* scratch;   // [max_frames][channels][max_block] int32: every channel of a frame as coded (wasted bits restored)
  void* pcm;          // interleaved output: int16 (bits <= 16) or int32 (left-justified) samples
  int out_shift;      // sample << out_shift fills the output word
  int out_wide;       // 0: int16, 1: int32
  int* status;
  const uint16_t* crc_tab;  // [8][256]
};

// sixteen bytes to a row of scratch as a streaming store: a lane's 64 bytes per burst open a fresh cache line that nothing
// reads before the finalize kernel (as ordinary stores they cost the single-wave form of this kernel 12 %)
__device__ __forceinline__ void fd_store4(int32_t* dst, int32_t a, int32_t b, int32_t c, int32_t d) {
#ifndef FD_NO_STORE  // tools only: what the stores cost
  typedef int32_t I4 __attribute__((ext_vector_type(4)));
  typedef I4 I4u __attribute__((aligned(4)));
  const I4 v = {a, b, c, d};
  __builtin_nontemporal_store(v, reinterpret_cast<I4u*>(dst));
#endif
}

// a frame's CRC-16 (poly 0x8005, no reflection, initial value 0: flac_decode.cpp crc16), eight bytes per table step (tables in
// LDS), 64 bytes per trip with the next 64 asked for before this trip's steps: the wave waits for memory once per 64 bytes
// (with 8 bytes per load it waited per load: ~170 us of a 1 ms kernel)
__device__ __forceinline__ uint32_t fd_crc16(const uint16_t (*crc)[256], const uint8_t* d, uint32_t n) {
  uint32_t cc = 0, i = 0;
  auto step8 = [&](uint32_t w0, uint32_t w1) __attribute__((always_inline)) {
    cc = crc[7][((cc >> 8) ^ w0) & 0xff] ^ crc[6][((cc & 0xff) ^ (w0 >> 8)) & 0xff] ^ crc[5][(w0 >> 16) & 0xff] ^ crc[4][w0 >> 24] ^
         crc[3][w1 & 0xff] ^ crc[2][(w1 >> 8) & 0xff] ^ crc[1][(w1 >> 16) & 0xff] ^ crc[0][w1 >> 24];
  };
  if (n >= 64) {
    uint32_t q[16], qn[16];
#pragma unroll
    for (int b = 0; b < 4; ++b) __builtin_memcpy(q + 4 * b, d + 16 * b, 16);
    for (; i + 64 <= n; i += 64) {
      const uint32_t nxt = i + 128 <= n ? i + 64 : i;  // (the last trip loads its own bytes again: no read past the frame)
#pragma unroll
      for (int b = 0; b < 4; ++b) __builtin_memcpy(qn + 4 * b, d + nxt + 16 * b, 16);
#pragma unroll
      for (int b = 0; b < 8; ++b) step8(q[2 * b], q[2 * b + 1]);
#pragma unroll
      for (int b = 0; b < 16; ++b) q[b] = qn[b];
    }
  }
  for (; i + 8 <= n; i += 8) {
    uint32_t w0, w1;
    __builtin_memcpy(&w0, d + i, 4);
    __builtin_memcpy(&w1, d + i + 4, 4);
    step8(w0, w1);
  }
  for (; i < n; ++i) cc = ((cc << 8) & 0xffff) ^ crc[0][((cc >> 8) ^ d[i]) & 0xff];
  return cc & 0xffff;
}

__global__ __launch_bounds__(3 * kFdLanes) void flac_decode_kernel(FdDecodeParams p) {
  __shared__ int32_t hist[32][kFdLanes];   // a subframe's warm-up samples; the last 32 restored samples for the orders above 12
  __shared__ int32_t coefs[32][kFdLanes];
  __shared__ uint16_t crc[8][256];
  __shared__ __attribute__((aligned(16))) uint32_t ring[kFdLanes][kFdRingRow];
  __shared__ __attribute__((aligned(16))) int32_t mb_val[kFdSlots][kFdLanes][kFdSlotRow];
  __shared__ uint32_t mb_desc[kFdSlots][kFdLanes];  // count | kind << 8
  __shared__ uint32_t mb_prod[kFdLanes], mb_cons[kFdLanes];
  __shared__ int sf_order[kFdLanes], sf_shift[kFdLanes], sf_wasted[kFdLanes], sf_chan[kFdLanes];
  for (int i = threadIdx.x; i < 8 * 256; i += 3 * kFdLanes) crc[i >> 8][i & 255] = p.crc_tab[i];
  if (threadIdx.x < kFdLanes) mb_prod[threadIdx.x] = 0, mb_cons[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & (kFdLanes - 1);
  const int role = threadIdx.x / kFdLanes;  // wave 0 parses, wave 1 restores, wave 2 checks the CRC-16
  const int f = blockIdx.x * kFdLanes + lane;
  if (f >= *p.n_frames) return;
#ifdef FD_CLOCK  // tools only: the shader clock this kernel runs at (core cycles against the 100 MHz wall clock)
  const long long fd_c0 = clock64(), fd_w0 = wall_clock64();
#endif
  const FdFrame fr = p.frames[f];
  const int bs = (int)fr.blocksize, n_ch = p.st.channels;
  int32_t* const scr = p.scratch + (size_t)f * p.st.max_block * n_ch;
  int err = 0;

  if (role == 2) {
    // ================================================ the checker ===============================================================
    // The chain kernel has fixed where every frame but the last one ends (the next frame's header: number + 1, CRC-8 right):
    // the CRC-16 over [offset, end - 2) needs nothing from the parser, which only confirms that ITS end is that end.
    if (f + 1 < *p.n_frames && fr.end >= fr.offset + 2 + fr.hdr_bytes) {
      const uint32_t cc = fd_crc16(crc, p.file + fr.offset, fr.end - 2 - fr.offset);
      const uint32_t want = ((uint32_t)p.file[fr.end - 2] << 8) | p.file[fr.end - 1];
      if (cc != want) atomicOr(p.status, (int)kFdCrc16);
    }
    return;
  }
  if (role == 1) {
    // ================================================ the restorer ==============================================================
    uint32_t cons = 0, idle = 0;
    int order = 0, shift = 0, wasted = 0, cls = 0, hat = 0;
    int32_t* out = scr;
    FdPred lpc;
#pragma unroll
    for (int j = 0; j < 12; ++j) lpc.c[j] = lpc.h[j] = 0.0;
    lpc.shift = 0;
    for (bool done = false; !done;) {
      const uint32_t prod = __hip_atomic_load(&mb_prod[lane], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (prod == cons) {
        if (++idle > kFdSpinCap) err |= kFdParse, done = true;
#ifndef FD_SLEEP
#define FD_SLEEP 1
#endif
        __builtin_amdgcn_s_sleep(FD_SLEEP);
        continue;
      }
      idle = 0;
      const int slot = (int)(cons & (kFdSlots - 1));
      const uint32_t desc = mb_desc[slot][lane];
      const int n = (int)(desc & 0xff);
      const uint32_t kind = (desc >> 8) & 0xff;
      int32_t* const v = mb_val[slot][lane];
      if ((kind == kFdMsgResidual && n == kFdBurst) || kind == kFdMsgCodes) {
        int32_t r[kFdBurst];
#pragma unroll
        for (int t = 0; t < kFdBurst; t += 4) __builtin_memcpy(r + t, v + t, 16);
        if (kind == kFdMsgCodes) {
          const int k = (int)(desc >> 16);
#pragma unroll
          for (int t = 0; t < kFdBurst; ++t) r[t] = fd_rice_value((uint32_t)r[t], k);
        }
#ifndef FD_NO_LPC  // tools only: what the prediction costs
        if (cls >= 1 && cls <= 3) {
          // the widest class among the lanes here serves them all (the coefficients beyond a lane's order are zero): one pass
          // of the prediction per burst instead of one per class present in the wave
          const int wcls = __builtin_amdgcn_ballot_w64(cls == 3) ? 3 : __builtin_amdgcn_ballot_w64(cls == 2) ? 2 : 1;
          if (wcls == 1) {
            lpc.burst<4>(r, wasted);
          } else if (wcls == 2) {
            lpc.burst<8>(r, wasted);
          } else {
            lpc.burst<12>(r, wasted);
          }
        } else if (cls == 0) {
#pragma unroll
          for (int t = 0; t < kFdBurst; ++t) r[t] = (int32_t)((uint32_t)r[t] << wasted);
        } else {
#pragma unroll
          for (int t = 0; t < kFdBurst; t += 4) __builtin_memcpy(v + t, r + t, 16);  // (the values, if codes came)
          for (int t = 0; t < kFdBurst; ++t) {
            // s[i] = res + (sum_j coef[j] s[i - 1 - j]) >> shift with 64-bit wrapping sums (flac_decode.cpp lpc_restore_n)
            uint64_t acc = 0;
            for (int j2 = 0; j2 < order; ++j2)
              acc += (uint64_t)((int64_t)coefs[j2][lane] * (int64_t)hist[(hat - 1 - j2) & 31][lane]);
            const int32_t sv = (int32_t)((uint32_t)v[t] + (uint32_t)((int64_t)acc >> shift));
            hist[hat & 31][lane] = sv;
            ++hat;
            v[t] = (int32_t)((uint32_t)sv << wasted);
          }
#pragma unroll
          for (int t = 0; t < kFdBurst; t += 4) __builtin_memcpy(r + t, v + t, 16);
        }
#endif
#pragma unroll
        for (int t = 0; t < kFdBurst; t += 4) fd_store4(out + t, r[t], r[t + 1], r[t + 2], r[t + 3]);
        out += kFdBurst;
      } else if (kind == kFdMsgResidual) {  // a partition's tail, an escaped partition: in place in the slot
#ifndef FD_NO_LPC
        if (cls == 0) {
          for (int t = 0; t < n; ++t) v[t] = (int32_t)((uint32_t)v[t] << wasted);
        } else if (cls == 1) {
          lpc.some<4>(v, n, wasted);
        } else if (cls == 2) {
          lpc.some<8>(v, n, wasted);
        } else if (cls == 3) {
          lpc.some<12>(v, n, wasted);
        } else {
          for (int t = 0; t < n; ++t) {
            uint64_t acc = 0;
            for (int j2 = 0; j2 < order; ++j2)
              acc += (uint64_t)((int64_t)coefs[j2][lane] * (int64_t)hist[(hat - 1 - j2) & 31][lane]);
            const int32_t sv = (int32_t)((uint32_t)v[t] + (uint32_t)((int64_t)acc >> shift));
            hist[hat & 31][lane] = sv;
            ++hat;
            v[t] = (int32_t)((uint32_t)sv << wasted);
          }
        }
#endif
#ifndef FD_NO_STORE
        for (int t = 0; t < n; ++t) out[t] = v[t];
#endif
        out += n;
      } else if (kind == kFdMsgSamples) {
#ifndef FD_NO_STORE
        for (int t = 0; t < n; ++t) out[t] = (int32_t)((uint32_t)v[t] << wasted);
#endif
        out += n;
      } else if (kind == kFdMsgSubframe) {
        order = sf_order[lane], shift = sf_shift[lane], wasted = sf_wasted[lane];
        out = scr + (size_t)sf_chan[lane] * p.st.max_block;
        cls = order == 0 ? 0 : order <= 4 ? 1 : order <= 8 ? 2 : order <= 12 ? 3 : 4;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
          const bool live = cls >= 1 && cls <= 3 && j < order;
          lpc.c[j] = live ? (double)coefs[j][lane] : 0.0;
          lpc.h[j] = live ? (double)hist[(order - 1 - j) & 31][lane] : 0.0;
        }
        lpc.shift = shift;
        hat = order;
        for (int i = 0; i < order; ++i) out[i] = (int32_t)((uint32_t)hist[i & 31][lane] << wasted);
        out += order;
      } else if (kind == kFdMsgConstant) {
        const int32_t cv = (int32_t)((uint32_t)v[0] << wasted);
        for (int i = 0; i < bs; ++i) out[i] = cv;
      } else {
        done = true;
      }
      ++cons;
      __hip_atomic_store(&mb_cons[lane], cons, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (err) atomicOr(p.status, err);
    return;
  }

  // ==================================================== the parser ================================================================
  FdBits br;
  br.init(p.file, fr.offset + fr.hdr_bytes, p.st.nbytes, ring[lane]);
  const uint32_t guard = fr.end + 16;  // a lane that reads past this has lost the stream
  uint32_t prod = 0, cons_seen = 0;
  // a free slot of this lane's mailbox.  The consumed counter is read again only when the last value seen leaves no slot:
  // the restorer is the faster of the two, so that is one LDS round trip per kFdSlots bursts, and it rarely has to wait.
  auto acquire = [&]() __attribute__((always_inline)) -> int32_t* {
    for (uint32_t spins = 0; prod - cons_seen >= (uint32_t)kFdSlots; ++spins) {
      cons_seen = __hip_atomic_load(&mb_cons[lane], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (spins > kFdSpinCap) {
        err |= kFdParse;
        break;
      }
    }
    return mb_val[prod & (kFdSlots - 1)][lane];
  };
  auto publish = [&](uint32_t kind, int n, int extra = 0) __attribute__((always_inline)) {
    mb_desc[prod & (kFdSlots - 1)][lane] = (uint32_t)n | (kind << 8) | ((uint32_t)extra << 16);
    ++prod;
    __hip_atomic_store(&mb_prod[lane], prod, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  // the restorer has taken everything sent so far (before `hist` / `coefs` / the descriptor of the next subframe are written)
  auto drained = [&]() __attribute__((always_inline)) {
    for (uint32_t spins = 0; spins <= kFdSpinCap; ++spins)
      if (__hip_atomic_load(&mb_cons[lane], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == prod) return;
    err |= kFdParse;
  };

  for (int c = 0; c < n_ch && !err; ++c) {
    const bool side = (fr.ch_code == 8 && c == 1) || (fr.ch_code == 9 && c == 0) || (fr.ch_code == 10 && c == 1);
    int bps = p.st.bits + (side ? 1 : 0);
    br.refuel();
    if (br.bits(1)) err |= kFdParse;
    const int type = (int)br.bits(6);
    int wasted = 0;
    if (br.bits(1)) wasted = (int)br.unary() + 1;
    bps -= wasted;
    if (bps <= 0 || bps > 32) {
      err |= kFdParse;
      break;
    }
    // predictor of this subframe: order, shift, coefficients in LDS (fixed predictors are LPC with binomial coefficients)
    int order = 0, shift = 0;
    if (type != 0 && type != 1) {
      if (type >= 8 && type <= 12) {
        order = type - 8;
      } else if (type >= 32) {
        order = type - 31;
      } else {
        err |= kFdParse;
        break;
      }
      if (order > bs) {
        err |= kFdParse;
        break;
      }
    }
    drained();
    if (type >= 8 && type <= 12) {
      const int fx[5][4] = {{0, 0, 0, 0}, {1, 0, 0, 0}, {2, -1, 0, 0}, {3, -3, 1, 0}, {4, -6, 4, -1}};
      for (int j = 0; j < order; ++j) coefs[j][lane] = fx[order][j];
    }
    for (int i = 0; i < order; ++i) hist[i & 31][lane] = br.sbits(bps);
    br.refuel();
    if (type >= 32) {
      const int prec = (int)br.bits(4) + 1;
      shift = br.sbits(5);
      if (prec == 16 || shift < 0) {
        err |= kFdParse;
        break;
      }
      for (int j = 0; j < order; ++j) coefs[j][lane] = br.sbits(prec);
    }
    sf_order[lane] = order, sf_shift[lane] = shift, sf_wasted[lane] = wasted, sf_chan[lane] = c;
    (void)acquire();
    publish(kFdMsgSubframe, 0);
    if (type == 0) {  // constant
      int32_t* v = acquire();
      v[0] = br.sbits(bps);
      publish(kFdMsgConstant, 1);
      continue;
    }
    if (type == 1) {  // verbatim
      for (int i = 0; i < bs && br.at() <= guard;) {
        const int n = bs - i < kFdBurst ? bs - i : kFdBurst;
        br.service();
        int32_t* v = acquire();
        for (int t = 0; t < n; ++t) v[t] = br.sbits(bps);
        publish(kFdMsgSamples, n);
        i += n;
      }
      if (br.at() > guard) err |= kFdParse;
      continue;
    }
    // residual (RFC 9639 section 9.2.7): partitions of Rice codes or escaped raw values
    const int method = (int)br.bits(2);
    if (method > 1) {
      err |= kFdParse;
      break;
    }
    const int pbits = method ? 5 : 4, esc = method ? 31 : 15;
    const int porder = (int)br.bits(4);
    const int parts = 1 << porder;
    if ((bs & (parts - 1)) || (bs >> porder) < order) {
      err |= kFdParse;
      break;
    }
    // ONE loop over bursts for the whole subframe, whatever its partitions: the lanes of a wave are frames whose partition
    // orders differ (a libFLAC stream: 0..6 from subframe to subframe), and a loop nest over (partition, burst) would hold
    // every lane at each partition's end until the lane with the longest partition got there.  Here a partition's header is
    // a short branch at the top and every trip is a burst.
    const int psize = bs >> porder;
    int left = 0, part = 0, k = 0, raw = 0;
    for (int i = order; i < bs && br.at() <= guard && !err;) {
      if (left == 0) {  // a partition begins (the first one may hold no residual at all)
        k = (int)br.bits(pbits);
        raw = k == esc ? (int)br.bits(5) : 0;
        left = psize - (part == 0 ? order : 0);
        ++part;
        if (left == 0) continue;
      }
      const bool escaped = k == esc;
      const int cap = escaped ? 8 : kFdBurst;
      const int n = left < cap ? left : cap;
      br.service();
      int32_t* v = acquire();
      if (!escaped && n == kFdBurst) {
        // sixteen codes = one basic block; the residuals cross to the restorer 16 bytes at a time
        uint32_t w[kFdBurst];
        const uint32_t hi0 = br.hi, lo0 = br.lo, nx0 = br.nx, n20 = br.n2, rd0 = br.rd;
        const int s0 = br.s;
        int nmax = 0;
#pragma unroll
        for (int t = 0; t < kFdBurst; ++t) w[t] = br.rice_window(k, &nmax);
        if (__builtin_expect(nmax > 32, 0)) {  // a code beyond the window somewhere: the burst again, code by code
          br.hi = hi0, br.lo = lo0, br.nx = nx0, br.n2 = n20, br.rd = rd0, br.s = s0;
          for (int t = 0; t < kFdBurst; ++t) v[t] = br.rice(k);
        } else {
#pragma unroll
          for (int t = 0; t < kFdBurst; t += 4) __builtin_memcpy(v + t, w + t, 16);
          publish(kFdMsgCodes, kFdBurst, k);
          left -= n;
          i += n;
          continue;
        }
      } else if (escaped) {
        for (int t = 0; t < n; ++t) v[t] = br.sbits(raw);
      } else {
        for (int t = 0; t < n; ++t) v[t] = br.rice(k);
      }
      publish(kFdMsgResidual, n);
      left -= n;
      i += n;
    }
    if (br.at() > guard) {  // ran off the frame (corrupt): never read far behind the file's buffer
      err |= kFdParse;
      break;
    }
  }
  (void)acquire();
  publish(kFdMsgEnd, 0);
  if (!err) {
    const uint32_t body_end = br.bytes_consumed_aligned();
    if (body_end + 2 > fr.end || body_end <= fr.offset) {
      err |= kFdParse;
    } else if (f + 1 < *p.n_frames) {
      // the next frame must begin right behind the CRC-16, which the checker wave has then computed over the right bytes
      if (body_end + 2 != fr.end) err |= kFdChain;
    } else {  // the last frame may be followed by padding / tags: its end is known only now
      const uint32_t cc = fd_crc16(crc, p.file + fr.offset, body_end - fr.offset);
      const uint32_t want = ((uint32_t)p.file[body_end] << 8) | p.file[body_end + 1];
      if ((cc & 0xffff) != want) err |= kFdCrc16;
    }
  }
#ifdef FD_CLOCK
  if (f == 0 || f == 700) {
    const long long dc = clock64() - fd_c0, dw = wall_clock64() - fd_w0;
    printf("FDCLK frame %d: %lld core cycles, %lld wall ticks (100 MHz) = %.1f us, %.0f MHz\n", f, dc, dw, dw / 100.0, dc * 100.0 / dw);
  }
#endif
#ifdef FD_DEBUG
  if (err || f < 2) printf("FDDBG frame %d off %u end %u bs %d err %d at %u\n", f, fr.offset, fr.end, bs, err, br.at());
#endif
  if (err) atomicOr(p.status, err);
}

// ---- 4. the parallel tail: stereo decorrelation (RFC 9639 section 4.2) and the interleaved output words ------------------------
__global__ __launch_bounds__(256) void flac_finalize_kernel(FdDecodeParams p) {
  const int f = blockIdx.y;
  if (f >= *p.n_frames) return;
  const FdFrame fr = p.frames[f];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int64_t keep = fr.first_sample + fr.blocksize <= p.st.total ? (int64_t)fr.blocksize : (p.st.total - fr.first_sample);
  if (i >= keep) return;
  const int n_ch = p.st.channels;
  const int32_t* scr = p.scratch + (size_t)f * p.st.max_block * n_ch;
  const int64_t o = (fr.first_sample + i) * n_ch;
  if (n_ch == 2) {
    int32_t a = scr[i], b = scr[p.st.max_block + i];  // channel 0, channel 1 as coded
    if (fr.ch_code == 8) {         // left / side
      b = (int32_t)((uint32_t)a - (uint32_t)b);
    } else if (fr.ch_code == 9) {  // side / right
      a = (int32_t)((uint32_t)a + (uint32_t)b);
    } else if (fr.ch_code == 10) {  // mid / side
      const int64_t side_v = b, mid = ((int64_t)a << 1) + (side_v & 1);
      a = (int32_t)((mid + side_v) >> 1);
      b = (int32_t)((mid - side_v) >> 1);
    }
    if (p.out_wide) {
      static_cast<int32_t*>(p.pcm)[o] = (int32_t)((uint32_t)a << p.out_shift);
      static_cast<int32_t*>(p.pcm)[o + 1] = (int32_t)((uint32_t)b << p.out_shift);
    } else {
      static_cast<int16_t*>(p.pcm)[o] = (int16_t)((uint32_t)a << p.out_shift);
      static_cast<int16_t*>(p.pcm)[o + 1] = (int16_t)((uint32_t)b << p.out_shift);
    }
  } else {
    for (int c = 0; c < n_ch; ++c) {
      const int32_t v = scr[(size_t)c * p.st.max_block + i];
      if (p.out_wide) static_cast<int32_t*>(p.pcm)[o + c] = (int32_t)((uint32_t)v << p.out_shift);
      else static_cast<int16_t*>(p.pcm)[o + c] = (int16_t)((uint32_t)v << p.out_shift);
    }
  }
}

// ---- host side --------------------------------------------------------------------------------------------------------------------
struct FlacDeviceBuffers {  // (the same layout is declared in bp_api.hip)
  uint8_t* file = nullptr;      // the file's bytes + 64 zero bytes
  size_t file_cap = 0;
  void* cands = nullptr;        // FdCand [chunks][kFdChunkCands]
  uint32_t* counts = nullptr;
  size_t cands_cap = 0, counts_cap = 0;
  void* packed = nullptr;       // FdCand, in file order
  uint32_t* offs = nullptr;
  size_t packed_cap = 0, offs_cap = 0;
  void* frames = nullptr;       // FdFrame
  int32_t* scratch = nullptr;
  size_t frames_cap = 0, scratch_cap = 0;
  int* meta = nullptr;          // [0] status, [1] n_frames
  uint16_t* crc_tab = nullptr;
};

static int fd_grow(void** p, size_t* cap, size_t want, size_t elem) {
  if (want <= *cap) return 0;
  if (*p) (void)hipFree(*p);
  *p = nullptr, *cap = 0;
  const size_t n = want + want / 4 + 64;
  if (hipMalloc(p, n * elem) != hipSuccess) return -1;
  *cap = n;
  return 0;
}

// Decode the FLAC stream at `d_file` (already on the device, padded with >= 64 zero bytes) into interleaved PCM at `d_pcm`
// (int16 when bits <= 16, else int32 left-justified).  Asynchronous on `stream`; *status (device) receives the error bits.
int flac_device_decode(FlacDeviceBuffers& b, const FdStream& st, void* d_pcm, hipStream_t stream) {
  const int n_chunks = (int)((st.nbytes - st.audio_start + kFdChunk - 1) / kFdChunk);
  const int64_t max_frames = (st.total + st.min_block - 1) / st.min_block + 1;
  if (fd_grow(&b.cands, &b.cands_cap, (size_t)n_chunks * kFdChunkCands, sizeof(FdCand))) return -1;
  if (fd_grow(reinterpret_cast<void**>(&b.counts), &b.counts_cap, (size_t)n_chunks, sizeof(uint32_t))) return -1;
  if (fd_grow(&b.packed, &b.packed_cap, (size_t)n_chunks * kFdChunkCands, sizeof(FdCand))) return -1;
  if (fd_grow(reinterpret_cast<void**>(&b.offs), &b.offs_cap, (size_t)n_chunks, sizeof(uint32_t))) return -1;
  if (fd_grow(&b.frames, &b.frames_cap, (size_t)max_frames, sizeof(FdFrame))) return -1;
  const size_t rows = (size_t)max_frames * st.max_block * st.channels;
  if (fd_grow(reinterpret_cast<void**>(&b.scratch), &b.scratch_cap, rows, sizeof(int32_t))) return -1;
  if (!b.meta && hipMalloc(&b.meta, 2 * sizeof(int)) != hipSuccess) return -1;
  if (!b.crc_tab) {
    uint16_t tab[8][256];
    for (int i = 0; i < 256; ++i) {
      uint16_t c = (uint16_t)(i << 8);
      for (int k = 0; k < 8; ++k) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : c << 1);
      tab[0][i] = c;
    }
    for (int k = 1; k < 8; ++k)
      for (int i = 0; i < 256; ++i) tab[k][i] = (uint16_t)((tab[k - 1][i] << 8) ^ tab[0][tab[k - 1][i] >> 8]);
    if (hipMalloc(&b.crc_tab, sizeof tab) != hipSuccess) return -1;
    if (hipMemcpy(b.crc_tab, tab, sizeof tab, hipMemcpyHostToDevice) != hipSuccess) return -1;
  }
  if (hipMemsetAsync(b.meta, 0, 2 * sizeof(int), stream) != hipSuccess) return -1;
  hipLaunchKernelGGL(flac_scan_kernel, dim3(n_chunks), dim3(256), 0, stream, b.file, st, static_cast<FdCand*>(b.cands), b.counts, b.meta);
  hipLaunchKernelGGL(flac_chain_kernel, dim3(1), dim3(kFdChainThreads), 0, stream, static_cast<const FdCand*>(b.cands), b.counts, n_chunks,
                     st, static_cast<FdCand*>(b.packed), b.offs, static_cast<FdFrame*>(b.frames), (int)max_frames, b.meta + 1, b.meta);
  FdDecodeParams p{b.file, static_cast<const FdFrame*>(b.frames), b.meta + 1, st, b.scratch, d_pcm, st.bits <= 16 ? 16 - st.bits : 32 - st.bits,
                   st.bits <= 16 ? 0 : 1, b.meta, b.crc_tab};
  hipLaunchKernelGGL(flac_decode_kernel, dim3((unsigned)((max_frames + kFdLanes - 1) / kFdLanes)), dim3(3 * kFdLanes), 0, stream, p);
  hipLaunchKernelGGL(flac_finalize_kernel, dim3((unsigned)((st.max_block + 255) / 256), (unsigned)max_frames), dim3(256), 0, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

void flac_device_free(FlacDeviceBuffers& b) {
  void* ptrs[] = {b.file, b.cands, b.counts, b.packed, b.offs, b.frames, b.scratch, b.meta, b.crc_tab};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  b = FlacDeviceBuffers();
}

}  // namespace bp
