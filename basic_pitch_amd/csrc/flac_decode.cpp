// FLAC decode on the host — the container-parsing half of `librosa.load(path, sr=22050, mono=True)`
// (basic_pitch/inference.py:239; the reference README lists .flac among the supported inputs, README.md:182-189).
// librosa reads FLAC through soundfile/libsndfile, neither of which is available here, so this is a decoder written
// from the FLAC format specification (RFC 9639): STREAMINFO, frame headers with CRC-8, the four subframe types
// (constant, verbatim, fixed order 0..4, LPC order 1..32) with wasted bits, Rice / Rice2 residuals with escaped
// partitions, the three stereo decorrelations, CRC-16 per frame and the MD5 of the decoded samples against
// STREAMINFO's.  Samples leave as interleaved float32 in [-1, 1): value / 2^(bits - 1), libsndfile's float conversion.
// Downmix and resampling happen on the device afterwards (audio_ingest.hip).
// Speed (round 4: Rice partitions decoded from a register-resident bit buffer, 64-bit windows elsewhere instead of
// byte-wise reads, the LPC sum unrolled per order, CRC-16 eight bytes at a time, MD5 written out): 76 M samples per second
// and core on 16-bit material (was 27) — a 3-minute stereo file in 0.21 s.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/basic_pitch_amd.h"

namespace {

thread_local std::string g_err;

struct BitReader {
  const uint8_t* p;
  size_t n, pos = 0;  // pos in bits
  bool fail = false;
  BitReader(const uint8_t* d, size_t len) : p(d), n(len) {}
  // the next 57+ bits, left-aligned in a 64-bit word (only where eight whole bytes are left)
  bool can_peek() const { return (pos >> 3) + 8 <= n; }
  uint64_t peek() const {
    uint64_t w;
    std::memcpy(&w, p + (pos >> 3), 8);
    return __builtin_bswap64(w) << (pos & 7);
  }
  uint64_t bits_slow(int k) {
    uint64_t v = 0;
    while (k > 0) {
      const size_t byte = pos >> 3;
      if (byte >= n) {
        fail = true;
        return 0;
      }
      const int avail = 8 - (int)(pos & 7);
      const int take = k < avail ? k : avail;
      v = (v << take) | ((p[byte] >> (avail - take)) & ((1u << take) - 1u));
      pos += take;
      k -= take;
    }
    return v;
  }
  uint64_t bits(int k) {  // k <= 57
    if (k > 0 && can_peek()) {
      const uint64_t v = peek() >> (64 - k);
      pos += k;
      return v;
    }
    return bits_slow(k);
  }
  int64_t sbits(int k) {
    if (k == 0) return 0;
    uint64_t v = k > 32 ? ((bits(k - 32) << 32) | bits(32)) : bits(k);
    const uint64_t sign = 1ull << (k - 1);
    return (int64_t)((v ^ sign) - sign);
  }
  uint32_t unary() {  // number of 0 bits before the next 1
    uint32_t q = 0;
    for (;;) {
      const size_t byte = pos >> 3;
      if (byte >= n) {
        fail = true;
        return 0;
      }
      const int off = (int)(pos & 7);
      const uint8_t rest = (uint8_t)(p[byte] << off);
      if (rest) {
        const int lz = __builtin_clz((uint32_t)rest) - 24;
        pos += lz + 1;
        return q + lz;
      }
      q += 8 - off;
      pos += 8 - off;
    }
  }
  // one Rice-coded residual with parameter k (unary quotient, k remainder bits, zig-zag sign): one 64-bit window when the
  // whole code fits it, which is every code of ordinary audio
  int64_t rice(int k) {
    uint64_t v;
    if (can_peek()) {
      const uint64_t w = peek();
      const int lz = w ? __builtin_clzll(w) : 64;
      if (lz + 1 + k <= 57) {
        const uint64_t r = k ? (w << (lz + 1)) >> (64 - k) : 0;
        pos += (size_t)(lz + 1 + k);
        v = ((uint64_t)lz << k) | r;
        return (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
      }
    }
    const uint64_t q = unary();
    v = (q << k) | (k ? bits(k) : 0);
    return (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
  }
  // `count` Rice codes of one partition.  The bits live in a register (left-aligned, `cnt` of them valid, refilled eight
  // bytes at a time) so that the serial chain per code is count-leading-zeros -> shift, not position -> load -> swap ->
  // shift -> count; anything unusual (a unary run past the register, the stream's last bytes) goes through rice().
  void rice_run(int k, int count, int64_t* out) {
    int j = 0;
    if (k <= 24) {
      const uint8_t* bp = p + (pos >> 3);
      const uint8_t* const safe_end = p + n - 8;  // a refill reads eight bytes
      uint64_t buf = 0;
      int cnt = 0;
      if (bp <= safe_end) {
        uint64_t w;
        std::memcpy(&w, bp, 8);
        buf = __builtin_bswap64(w) << (pos & 7);
        cnt = 64 - (int)(pos & 7);
        bp += 8;
        for (; j < count; ++j) {
          if (cnt < 32) {  // top up to > 56 valid bits: whole bytes only
            if (bp > safe_end) break;
            std::memcpy(&w, bp, 8);
            const int take = (64 - cnt) >> 3;  // bytes that fit
            buf |= (__builtin_bswap64(w) >> cnt) & ~((take * 8 + cnt) >= 64 ? 0ull : (~0ull >> (take * 8 + cnt)));
            bp += take;
            cnt += take * 8;
          }
          if (buf == 0) break;
          const int lz = __builtin_clzll(buf);
          const int need = lz + 1 + k;
          if (need > cnt) break;
          const uint64_t r = k ? (buf << (lz + 1)) >> (64 - k) : 0;
          // a code may fill the register exactly (need == cnt == 64: k = 0 and 63 zeros, after a byte-aligned load or a
          // refill from a multiple of 8): `buf <<= 64` is undefined — on x86 a no-op that would leave the stop bit in the
          // register, to be OR-ed onto the next refill — so the shift goes in two steps (need >= 1 always)
          buf = (buf << (need - 1)) << 1;
          cnt -= need;
          const uint64_t v = ((uint64_t)lz << k) | r;
          out[j] = (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
        }
        pos = (size_t)(bp - p) * 8 - (size_t)cnt;
      }
    }
    for (; j < count; ++j) out[j] = rice(k);
  }
  void align() { pos = (pos + 7) & ~(size_t)7; }
};

uint8_t crc8(const uint8_t* d, size_t n) {
  uint8_t c = 0;
  for (size_t i = 0; i < n; ++i) {
    c ^= d[i];
    for (int b = 0; b < 8; ++b) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : c << 1);
  }
  return c;
}

// CRC-16 (polynomial 0x8005, MSB first) eight bytes at a time: v[k][x] = the CRC of byte x followed by k zero bytes, so
// the state after eight more bytes is the XOR of eight independent look-ups (the byte-serial form, one dependent look-up
// per byte, was 2.5 ns of a decoded sample's 16)
struct Crc16Table {
  uint16_t v[8][256];
  Crc16Table() {
    for (int i = 0; i < 256; ++i) {
      uint16_t c = (uint16_t)(i << 8);
      for (int b = 0; b < 8; ++b) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : c << 1);
      v[0][i] = c;
    }
    for (int k = 1; k < 8; ++k)
      for (int i = 0; i < 256; ++i) v[k][i] = (uint16_t)((v[k - 1][i] << 8) ^ v[0][v[k - 1][i] >> 8]);
  }
};

uint16_t crc16(const uint8_t* d, size_t n) {
  // a C++11 magic static: initialised once, thread-safely (files are decoded concurrently from a host thread pool)
  static const Crc16Table table;
  uint16_t c = 0;
  size_t i = 0;
  for (; i + 8 <= n; i += 8)
    c = (uint16_t)(table.v[7][(c >> 8) ^ d[i]] ^ table.v[6][(c & 0xff) ^ d[i + 1]] ^ table.v[5][d[i + 2]] ^ table.v[4][d[i + 3]] ^
                   table.v[3][d[i + 4]] ^ table.v[2][d[i + 5]] ^ table.v[1][d[i + 6]] ^ table.v[0][d[i + 7]]);
  for (; i < n; ++i) c = (uint16_t)((c << 8) ^ table.v[0][(c >> 8) ^ d[i]]);
  return c;
}

// MD5 (RFC 1321) of the decoded samples, little-endian, interleaved, ceil(bits / 8) bytes each
struct Md5 {
  uint32_t a = 0x67452301, b = 0xefcdab89, c = 0x98badcfe, d = 0x10325476;
  uint64_t len = 0;
  uint8_t buf[64];
  size_t fill = 0;
  static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
  void block(const uint8_t* m) {
    static const uint32_t K[64] = {
        0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8,
        0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340,
        0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87,
        0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c,
        0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039,
        0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92,
        0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb,
        0xeb86d391};
    static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,
                              14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                              4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
    uint32_t w[16];
    std::memcpy(w, m, 64);  // little-endian host (x86-64): the words as they lie
    uint32_t A = a, B = b, C = c, D = d;
    // the 64 steps written out (round function, message index and rotation are compile-time constants per step)
#define BP_MD5_STEP(F, A_, B_, C_, D_, I)                                   \
  A_ = B_ + rol(A_ + F(B_, C_, D_) + K[I] + w[G(I)], S[I]);
#define BP_MD5_F1(x, y, z) ((z) ^ ((x) & ((y) ^ (z))))
#define BP_MD5_F2(x, y, z) ((y) ^ ((z) & ((x) ^ (y))))
#define BP_MD5_F3(x, y, z) ((x) ^ (y) ^ (z))
#define BP_MD5_F4(x, y, z) ((y) ^ ((x) | ~(z)))
#define BP_MD5_4(F, I) BP_MD5_STEP(F, A, B, C, D, I) BP_MD5_STEP(F, D, A, B, C, I + 1) BP_MD5_STEP(F, C, D, A, B, I + 2) BP_MD5_STEP(F, B, C, D, A, I + 3)
#define G(i) (i)
    BP_MD5_4(BP_MD5_F1, 0) BP_MD5_4(BP_MD5_F1, 4) BP_MD5_4(BP_MD5_F1, 8) BP_MD5_4(BP_MD5_F1, 12)
#undef G
#define G(i) ((5 * (i) + 1) & 15)
    BP_MD5_4(BP_MD5_F2, 16) BP_MD5_4(BP_MD5_F2, 20) BP_MD5_4(BP_MD5_F2, 24) BP_MD5_4(BP_MD5_F2, 28)
#undef G
#define G(i) ((3 * (i) + 5) & 15)
    BP_MD5_4(BP_MD5_F3, 32) BP_MD5_4(BP_MD5_F3, 36) BP_MD5_4(BP_MD5_F3, 40) BP_MD5_4(BP_MD5_F3, 44)
#undef G
#define G(i) ((7 * (i)) & 15)
    BP_MD5_4(BP_MD5_F4, 48) BP_MD5_4(BP_MD5_F4, 52) BP_MD5_4(BP_MD5_F4, 56) BP_MD5_4(BP_MD5_F4, 60)
#undef G
#undef BP_MD5_4
#undef BP_MD5_F1
#undef BP_MD5_F2
#undef BP_MD5_F3
#undef BP_MD5_F4
#undef BP_MD5_STEP
    a += A, b += B, c += C, d += D;
  }
  void update(const uint8_t* p, size_t n) {
    len += n;
    while (n) {
      if (fill == 0 && n >= 64) {  // whole blocks straight from the input
        block(p);
        p += 64, n -= 64;
        continue;
      }
      const size_t take = n < 64 - fill ? n : 64 - fill;
      std::memcpy(buf + fill, p, take);
      fill += take, p += take, n -= take;
      if (fill == 64) block(buf), fill = 0;
    }
  }
  void finish(uint8_t out[16]) {
    const uint64_t bitlen = len * 8;
    const uint8_t one = 0x80, zero = 0;
    update(&one, 1);
    while (fill != 56) update(&zero, 1);
    uint8_t l[8];
    for (int i = 0; i < 8; ++i) l[i] = (uint8_t)(bitlen >> (8 * i));
    update(l, 8);
    const uint32_t r[4] = {a, b, c, d};
    for (int i = 0; i < 16; ++i) out[i] = (uint8_t)(r[i / 4] >> (8 * (i % 4)));
  }
};

struct StreamInfo {
  int sample_rate = 0, channels = 0, bits = 0;
  int min_block = 0, max_block = 0;
  int64_t total = 0;
  uint8_t md5[16] = {0};
  size_t audio_start = 0;
};

bool parse_header(const uint8_t* d, size_t n, StreamInfo& si) {
  size_t pos = 0;
  if (n >= 10 && !std::memcmp(d, "ID3", 3))  // ID3v2 tag in front of the stream
    pos = 10 + (((size_t)d[6] & 0x7f) << 21 | ((size_t)d[7] & 0x7f) << 14 | ((size_t)d[8] & 0x7f) << 7 | ((size_t)d[9] & 0x7f));
  if (pos + 4 > n || std::memcmp(d + pos, "fLaC", 4)) {
    g_err = "not a FLAC stream (no fLaC marker)";
    return false;
  }
  pos += 4;
  bool have = false;
  for (;;) {
    if (pos + 4 > n) {
      g_err = "truncated FLAC metadata";
      return false;
    }
    const bool last = d[pos] & 0x80;
    const int type = d[pos] & 0x7f;
    const size_t len = ((size_t)d[pos + 1] << 16) | ((size_t)d[pos + 2] << 8) | d[pos + 3];
    pos += 4;
    if (pos + len > n) {
      g_err = "truncated FLAC metadata block";
      return false;
    }
    if (type == 0) {
      if (len < 34) {
        g_err = "short STREAMINFO block";
        return false;
      }
      const uint8_t* s = d + pos;
      si.min_block = (s[0] << 8) | s[1];
      si.max_block = (s[2] << 8) | s[3];
      si.sample_rate = (s[10] << 12) | (s[11] << 4) | (s[12] >> 4);
      si.channels = ((s[12] >> 1) & 7) + 1;
      si.bits = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
      si.total = ((int64_t)(s[13] & 0x0f) << 32) | ((int64_t)s[14] << 24) | (s[15] << 16) | (s[16] << 8) | s[17];
      std::memcpy(si.md5, s + 18, 16);
      have = true;
    }
    pos += len;
    if (last) break;
  }
  if (!have || si.sample_rate == 0) {
    g_err = "FLAC stream without a valid STREAMINFO block";
    return false;
  }
  si.audio_start = pos;
  return true;
}

bool read_residual(BitReader& br, int order, int blocksize, std::vector<int64_t>& s) {
  const int method = (int)br.bits(2);
  if (method > 1) {
    g_err = "reserved residual coding method";
    return false;
  }
  const int pbits = method ? 5 : 4, esc = method ? 31 : 15;
  const int porder = (int)br.bits(4);
  const int parts = 1 << porder;
  if ((blocksize & (parts - 1)) || (blocksize >> porder) < order) {
    g_err = "invalid Rice partition order";
    return false;
  }
  int i = order;
  for (int p = 0; p < parts; ++p) {
    const int count = (blocksize >> porder) - (p == 0 ? order : 0);
    const int k = (int)br.bits(pbits);
    if (k == esc) {
      const int raw = (int)br.bits(5);
      for (int j = 0; j < count; ++j) s[i++] = br.sbits(raw);
    } else {
      br.rice_run(k, count, s.data() + i);
      i += count;
    }
    if (br.fail) break;
  }
  return !br.fail;
}

// s[i] += (sum_j coef[j] s[i - 1 - j]) >> shift for i >= order: the inner sum unrolled for the order at hand (the loop
// over a run-time order costs a branch per tap in the one serial chain of the decoder)
template <int ORDER>
void lpc_restore_n(int64_t* s, const int64_t* coef, int shift, int blocksize) {
  int64_t c[ORDER];
  for (int j = 0; j < ORDER; ++j) c[j] = coef[j];
  // unsigned arithmetic: a corrupted stream (caught by the frame CRC afterwards) may drive the sums past 63 bits, and
  // wrapping is defined only there
  for (int i = ORDER; i < blocksize; ++i) {
    uint64_t acc = 0;
#pragma GCC unroll 32
    for (int j = 0; j < ORDER; ++j) acc += (uint64_t)c[j] * (uint64_t)s[i - 1 - j];
    s[i] = (int64_t)((uint64_t)s[i] + (uint64_t)((int64_t)acc >> shift));
  }
}

void lpc_restore(int64_t* s, const int64_t* coef, int order, int shift, int blocksize) {
  switch (order) {
#define BP_LPC_CASE(N) case N: lpc_restore_n<N>(s, coef, shift, blocksize); return;
    BP_LPC_CASE(1) BP_LPC_CASE(2) BP_LPC_CASE(3) BP_LPC_CASE(4) BP_LPC_CASE(5) BP_LPC_CASE(6) BP_LPC_CASE(7) BP_LPC_CASE(8)
    BP_LPC_CASE(9) BP_LPC_CASE(10) BP_LPC_CASE(11) BP_LPC_CASE(12) BP_LPC_CASE(13) BP_LPC_CASE(14) BP_LPC_CASE(15) BP_LPC_CASE(16)
    BP_LPC_CASE(17) BP_LPC_CASE(18) BP_LPC_CASE(19) BP_LPC_CASE(20) BP_LPC_CASE(21) BP_LPC_CASE(22) BP_LPC_CASE(23) BP_LPC_CASE(24)
    BP_LPC_CASE(25) BP_LPC_CASE(26) BP_LPC_CASE(27) BP_LPC_CASE(28) BP_LPC_CASE(29) BP_LPC_CASE(30) BP_LPC_CASE(31) BP_LPC_CASE(32)
#undef BP_LPC_CASE
    default: return;
  }
}

bool read_subframe(BitReader& br, int bps, int blocksize, std::vector<int64_t>& s) {
  if (br.bits(1)) {
    g_err = "subframe padding bit set";
    return false;
  }
  const int type = (int)br.bits(6);
  int wasted = 0;
  if (br.bits(1)) wasted = (int)br.unary() + 1;
  bps -= wasted;
  if (bps <= 0) {
    g_err = "wasted bits exceed the sample size";
    return false;
  }
  s.assign(blocksize, 0);
  if (type == 0) {  // constant
    const int64_t v = br.sbits(bps);
    for (int i = 0; i < blocksize; ++i) s[i] = v;
  } else if (type == 1) {  // verbatim
    for (int i = 0; i < blocksize; ++i) s[i] = br.sbits(bps);
  } else if (type >= 8 && type <= 12) {  // fixed predictor
    const int order = type - 8;
    if (order > blocksize) {
      g_err = "fixed predictor order exceeds the block size";
      return false;
    }
    for (int i = 0; i < order; ++i) s[i] = br.sbits(bps);
    if (!read_residual(br, order, blocksize, s)) return false;
    for (int i = order; i < blocksize; ++i) {
      const auto u = [&](int k) { return (uint64_t)s[i - k]; };  // wrapping sums (see lpc_restore_n)
      switch (order) {
        case 1: s[i] = (int64_t)((uint64_t)s[i] + u(1)); break;
        case 2: s[i] = (int64_t)((uint64_t)s[i] + 2 * u(1) - u(2)); break;
        case 3: s[i] = (int64_t)((uint64_t)s[i] + 3 * u(1) - 3 * u(2) + u(3)); break;
        case 4: s[i] = (int64_t)((uint64_t)s[i] + 4 * u(1) - 6 * u(2) + 4 * u(3) - u(4)); break;
        default: break;
      }
    }
  } else if (type >= 32) {  // LPC
    const int order = type - 31;
    if (order > blocksize) {
      g_err = "LPC order exceeds the block size";
      return false;
    }
    for (int i = 0; i < order; ++i) s[i] = br.sbits(bps);
    const int prec = (int)br.bits(4) + 1;
    if (prec == 16) {
      g_err = "invalid LPC coefficient precision";
      return false;
    }
    const int shift = (int)br.sbits(5);
    if (shift < 0) {
      g_err = "negative LPC shift";
      return false;
    }
    int64_t coef[32];
    for (int j = 0; j < order; ++j) coef[j] = br.sbits(prec);
    if (!read_residual(br, order, blocksize, s)) return false;
    lpc_restore(s.data(), coef, order, shift, blocksize);
  } else {
    g_err = "reserved subframe type";
    return false;
  }
  if (wasted)
    for (int i = 0; i < blocksize; ++i) s[i] = (int64_t)((uint64_t)s[i] << wasted);
  return !br.fail;
}

// decode everything; pcm == nullptr: headers only
int decode(const uint8_t* d, size_t n, float* pcm, int64_t capacity, StreamInfo& si, int64_t* n_frames) {
  if (!parse_header(d, n, si)) return BP_ERR_BAD_AUDIO;
  if (!pcm) {
    *n_frames = si.total;
    if (si.total > 0) return BP_OK;
  }
  const double scale = 1.0 / (double)((int64_t)1 << (si.bits - 1));
  const int bytes_per = (si.bits + 7) / 8;
  Md5 md5;
  std::vector<std::vector<int64_t>> ch(si.channels);
  std::vector<uint8_t> raw;
  int64_t done = 0;
  size_t pos = si.audio_start;
  while (pos + 2 <= n) {
    if (d[pos] != 0xff || (d[pos + 1] & 0xfe) != 0xf8) {  // trailing bytes (ID3v1, padding) after the last frame
      if (si.total > 0 && done >= si.total) break;
      g_err = "lost FLAC frame sync";
      return BP_ERR_BAD_AUDIO;
    }
    BitReader br(d + pos, n - pos);
    br.bits(15);
    br.bits(1);  // blocking strategy: only changes the meaning of the coded number
    const int bs_code = (int)br.bits(4), sr_code = (int)br.bits(4);
    const int ch_code = (int)br.bits(4), sz_code = (int)br.bits(3);
    if (br.bits(1) || bs_code == 0 || sr_code == 15 || ch_code > 10 || sz_code == 3) {
      g_err = "reserved value in a FLAC frame header";
      return BP_ERR_BAD_AUDIO;
    }
    int lead = (int)br.bits(8);  // UTF-8-like coded frame / sample number: skip
    if (lead & 0x80) {
      int extra = 0;
      while (lead & (0x40 >> extra)) ++extra;
      for (int i = 0; i < extra; ++i) br.bits(8);
    }
    int blocksize;
    if (bs_code == 1) blocksize = 192;
    else if (bs_code <= 5) blocksize = 576 << (bs_code - 2);
    else if (bs_code == 6) blocksize = (int)br.bits(8) + 1;
    else if (bs_code == 7) blocksize = (int)br.bits(16) + 1;
    else blocksize = 256 << (bs_code - 8);
    if (sr_code == 12) br.bits(8);
    else if (sr_code == 13 || sr_code == 14) br.bits(16);
    static const int sz_table[8] = {0, 8, 12, 0, 16, 20, 24, 32};
    const int bits = sz_code ? sz_table[sz_code] : si.bits;
    if (bits != si.bits) {
      g_err = "FLAC frame sample size differs from STREAMINFO";
      return BP_ERR_BAD_AUDIO;
    }
    const size_t hdr_bytes = br.pos >> 3;
    const uint8_t want8 = (uint8_t)br.bits(8);
    if (br.fail || crc8(d + pos, hdr_bytes) != want8) {
      g_err = "FLAC frame header CRC-8 mismatch";
      return BP_ERR_BAD_AUDIO;
    }
    const int n_ch = ch_code < 8 ? ch_code + 1 : 2;
    if (n_ch != si.channels) {
      g_err = "FLAC frame channel count differs from STREAMINFO";
      return BP_ERR_BAD_AUDIO;
    }
    for (int c = 0; c < n_ch; ++c) {
      const bool side = (ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1);
      if (!read_subframe(br, bits + (side ? 1 : 0), blocksize, ch[c])) {
        if (br.fail) g_err = "truncated FLAC frame";
        return BP_ERR_BAD_AUDIO;
      }
    }
    br.align();
    const size_t body = br.pos >> 3;
    const uint16_t want16 = (uint16_t)br.bits(16);
    if (br.fail || crc16(d + pos, body) != want16) {
      g_err = "FLAC frame CRC-16 mismatch";
      return BP_ERR_BAD_AUDIO;
    }
    pos += body + 2;
    if (ch_code == 8) {
      for (int i = 0; i < blocksize; ++i) ch[1][i] = (int64_t)((uint64_t)ch[0][i] - (uint64_t)ch[1][i]);
    } else if (ch_code == 9) {
      for (int i = 0; i < blocksize; ++i) ch[0][i] = (int64_t)((uint64_t)ch[0][i] + (uint64_t)ch[1][i]);
    } else if (ch_code == 10) {
      for (int i = 0; i < blocksize; ++i) {
        const uint64_t side = (uint64_t)ch[1][i], mid = ((uint64_t)ch[0][i] << 1) + (side & 1);
        ch[0][i] = (int64_t)(mid + side) >> 1;
        ch[1][i] = (int64_t)(mid - side) >> 1;
      }
    }
    int64_t keep = blocksize;
    if (si.total > 0 && done + keep > si.total) keep = si.total - done;
    if (pcm) {
      if (done + keep > capacity) {
        g_err = "FLAC stream holds more frames than the output buffer";
        return BP_ERR_INVALID_ARG;
      }
      raw.resize((size_t)keep * n_ch * bytes_per);
      float* out = pcm + done * n_ch;
      if (bytes_per == 2) {  // the common case: the MD5's little-endian 16-bit samples written as such
        int16_t* r16 = reinterpret_cast<int16_t*>(raw.data());
        for (int c = 0; c < n_ch; ++c) {
          const int64_t* src = ch[c].data();
          for (int64_t i = 0; i < keep; ++i) {
            out[i * n_ch + c] = (float)((double)src[i] * scale);
            r16[i * n_ch + c] = (int16_t)src[i];
          }
        }
      } else {
        size_t r = 0;
        for (int64_t i = 0; i < keep; ++i)
          for (int c = 0; c < n_ch; ++c) {
            const int64_t v = ch[c][i];
            out[i * n_ch + c] = (float)((double)v * scale);
            for (int b = 0; b < bytes_per; ++b) raw[r++] = (uint8_t)((uint64_t)v >> (8 * b));
          }
      }
      md5.update(raw.data(), raw.size());
    }
    done += keep;
  }
  if (si.total > 0 && done != si.total) {
    g_err = "FLAC stream ends before STREAMINFO's sample count";
    return BP_ERR_BAD_AUDIO;
  }
  *n_frames = done;
  if (pcm) {
    static const uint8_t none[16] = {0};
    if (std::memcmp(si.md5, none, 16)) {
      uint8_t got[16];
      md5.finish(got);
      if (std::memcmp(got, si.md5, 16)) {
        g_err = "decoded audio does not match the MD5 in STREAMINFO";
        return BP_ERR_BAD_AUDIO;
      }
    }
  }
  return BP_OK;
}

}  // namespace

extern "C" {

int bp_flac_info(const void* file, size_t nbytes, int* channels, int* sample_rate, int* bits_per_sample,
                 int64_t* n_frames) {
  if (!file || !channels || !sample_rate || !bits_per_sample || !n_frames) {
    g_err = "bp_flac_info: null argument";
    return BP_ERR_INVALID_ARG;
  }
  StreamInfo si;
  int64_t n = 0;
  const int rc = decode(static_cast<const uint8_t*>(file), nbytes, nullptr, 0, si, &n);
  if (rc != BP_OK) return rc;
  *channels = si.channels, *sample_rate = si.sample_rate, *bits_per_sample = si.bits, *n_frames = n;
  return BP_OK;
}

int bp_flac_decode(const void* file, size_t nbytes, float* pcm, int64_t capacity_frames, int64_t* n_frames) {
  if (!file || !pcm || !n_frames || capacity_frames < 0) {
    g_err = "bp_flac_decode: null argument";
    return BP_ERR_INVALID_ARG;
  }
  StreamInfo si;
  return decode(static_cast<const uint8_t*>(file), nbytes, pcm, capacity_frames, si, n_frames);
}

int bp_flac_layout(const void* file, size_t nbytes, bp_flac_stream_layout* out) {
  if (!file || !out) {
    g_err = "bp_flac_layout: null argument";
    return BP_ERR_INVALID_ARG;
  }
  StreamInfo si;
  if (!parse_header(static_cast<const uint8_t*>(file), nbytes, si)) return BP_ERR_BAD_AUDIO;
  out->channels = si.channels, out->sample_rate = si.sample_rate, out->bits_per_sample = si.bits;
  out->min_block = si.min_block, out->max_block = si.max_block;
  out->n_frames = si.total;
  out->audio_start = (int64_t)si.audio_start;
  return BP_OK;
}

const char* bp_audio_last_error(void) { return g_err.c_str(); }

}  // extern "C"
