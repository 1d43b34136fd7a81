/* A small FLAC *encoder* for benchmark corpora (tools only; the product has no encoder): 16-bit PCM in, a stream like the
 * ones real encoders write out — fixed block size 4096, per subframe an LPC predictor of order 8 (autocorrelation +
 * Levinson-Durbin, 14-bit coefficients) or, where that is not better, the fixed order-2 predictor, Rice-coded residuals in 8
 * partitions with the parameter chosen per partition, independent stereo, CRC-8 / CRC-16 per frame, the sample count and
 * block sizes in STREAMINFO (MD5 left zero = "not computed", RFC 9639 section 8.2).  Written from RFC 9639; verified by the
 * product's host decoder, which checks every CRC (tests/test_flac_device.py::test_bench_corpus_encoder).
 *
 *   gcc -O2 -o flac_synth tools/flac_synth.c -lm
 *   flac_synth in.wav out.flac        (16-bit PCM WAV, canonical 44-byte header)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  uint8_t* p;
  size_t n, cap;
  uint64_t acc;
  int nb;
} BW;
static void bw_byte(BW* w, uint8_t b) {
  if (w->n == w->cap) {
    w->cap = w->cap ? w->cap * 2 : 1 << 20;
    w->p = (uint8_t*)realloc(w->p, w->cap);
  }
  w->p[w->n++] = b;
}
static void bw_bits(BW* w, uint64_t v, int k) { /* k <= 32 */
  if (!k) return;
  w->acc = (w->acc << k) | (v & ((k == 64 ? 0 : (1ull << k)) - 1));
  w->nb += k;
  while (w->nb >= 8) {
    w->nb -= 8;
    bw_byte(w, (uint8_t)(w->acc >> w->nb));
  }
}
static void bw_unary(BW* w, uint32_t q) {
  while (q >= 32) {
    bw_bits(w, 0, 32);
    q -= 32;
  }
  bw_bits(w, 1, (int)q + 1);
}
static void bw_align(BW* w) {
  if (w->nb) bw_bits(w, 0, 8 - w->nb);
}
static uint8_t crc8(const uint8_t* d, size_t n) {
  uint8_t c = 0;
  for (size_t i = 0; i < n; ++i) {
    c ^= d[i];
    for (int b = 0; b < 8; ++b) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : c << 1);
  }
  return c;
}
static uint16_t crc16(const uint8_t* d, size_t n) {
  static uint16_t t[256];
  static int init = 0;
  if (!init) {
    for (int i = 0; i < 256; ++i) {
      uint16_t c = (uint16_t)(i << 8);
      for (int b = 0; b < 8; ++b) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : c << 1);
      t[i] = c;
    }
    init = 1;
  }
  uint16_t c = 0;
  for (size_t i = 0; i < n; ++i) c = (uint16_t)((c << 8) ^ t[(c >> 8) ^ d[i]]);
  return c;
}
static void utf8(BW* w, uint64_t v) {
  if (v < 0x80) {
    bw_bits(w, v, 8);
    return;
  }
  int extra = v < 0x800 ? 1 : v < 0x10000 ? 2 : v < 0x200000 ? 3 : v < 0x4000000 ? 4 : 5;
  bw_bits(w, (0xff << (7 - extra)) & 0xff | (v >> (6 * extra)), 8);
  for (int i = extra - 1; i >= 0; --i) bw_bits(w, 0x80 | ((v >> (6 * i)) & 0x3f), 8);
}

static void residual(BW* w, const int32_t* res, int order, int bs) {
  const int porder = (bs % 8 == 0 && (bs >> 3) >= order) ? 3 : 0, parts = 1 << porder;
  bw_bits(w, 0, 2); /* Rice, 4-bit parameters */
  bw_bits(w, porder, 4);
  int i = order;
  for (int p = 0; p < parts; ++p) {
    const int count = (bs >> porder) - (p == 0 ? order : 0);
    double sum = 0;
    for (int j = 0; j < count; ++j) sum += fabs((double)res[i + j]);
    int k = 0;
    const double mean = count ? sum / count : 0;
    while (k < 14 && (double)(1 << (k + 1)) < mean * 1.4 + 1) ++k;
    bw_bits(w, k, 4);
    for (int j = 0; j < count; ++j, ++i) {
      const int32_t r = res[i];
      const uint32_t v = r >= 0 ? (uint32_t)r << 1 : (((uint32_t)(-(r + 1))) << 1) | 1;
      bw_unary(w, v >> k);
      bw_bits(w, v & ((1u << k) - 1), k);
    }
  }
}

static void subframe(BW* w, const int32_t* s, int bs) {
  /* LPC order 8: autocorrelation of the Hann-windowed block, Levinson-Durbin, 14-bit coefficients */
  enum { ORD = 8, PREC = 14 };
  double ac[ORD + 1] = {0}, lpc[ORD + 1] = {0}, tmp[ORD + 1];
  static double win[65536], ws[65536];
  static int win_bs = 0;
  if (win_bs != bs) {
    for (int i = 0; i < bs; ++i) win[i] = 0.5 - 0.5 * cos(2 * M_PI * (i + 0.5) / bs);
    win_bs = bs;
  }
  for (int i = 0; i < bs; ++i) ws[i] = s[i] * win[i];
  for (int lag = 0; lag <= ORD; ++lag) {
    double a = 0;
    for (int i = lag; i < bs; ++i) a += ws[i] * ws[i - lag];
    ac[lag] = a;
  }
  int use_lpc = ac[0] > 0;
  int32_t q[ORD] = {0};
  int shift = 0;
  if (use_lpc) {
    double err = ac[0];
    for (int i = 1; i <= ORD; ++i) {
      double r = -ac[i];
      for (int j = 1; j < i; ++j) r -= lpc[j] * ac[i - j];
      r /= err;
      memcpy(tmp, lpc, sizeof tmp);
      lpc[i] = r;
      for (int j = 1; j < i; ++j) lpc[j] = tmp[j] + r * tmp[i - j];
      err *= 1 - r * r;
      if (err <= 0) {
        use_lpc = 0;
        break;
      }
    }
  }
  if (use_lpc) {
    double mx = 0;
    for (int j = 1; j <= ORD; ++j) mx = fmax(mx, fabs(lpc[j]));
    shift = PREC - 2 - (int)floor(log2(mx > 1e-9 ? mx : 1e-9));
    if (shift > 15) shift = 15;
    if (shift < 0) use_lpc = 0;
    for (int j = 0; j < ORD && use_lpc; ++j) {
      double v = -lpc[j + 1] * (double)(1 << shift);
      long r = lround(v);
      if (r >= (1 << (PREC - 1))) r = (1 << (PREC - 1)) - 1;
      if (r < -(1 << (PREC - 1))) r = -(1 << (PREC - 1));
      q[j] = (int32_t)r;
    }
  }
  int32_t* res = (int32_t*)malloc(sizeof(int32_t) * (size_t)bs);
  bw_bits(w, 0, 1);
  if (use_lpc) {
    for (int i = ORD; i < bs; ++i) {
      int64_t acc = 0;
      for (int j = 0; j < ORD; ++j) acc += (int64_t)q[j] * s[i - 1 - j];
      res[i] = s[i] - (int32_t)(acc >> shift);
    }
    bw_bits(w, 32 + ORD - 1, 6);
    bw_bits(w, 0, 1);
    for (int i = 0; i < ORD; ++i) bw_bits(w, (uint32_t)s[i], 16);
    bw_bits(w, PREC - 1, 4);
    bw_bits(w, (uint32_t)shift, 5);
    for (int j = 0; j < ORD; ++j) bw_bits(w, (uint32_t)q[j], PREC);
    residual(w, res, ORD, bs);
  } else { /* fixed order 2 */
    for (int i = 2; i < bs; ++i) res[i] = s[i] - 2 * s[i - 1] + s[i - 2];
    bw_bits(w, 8 + 2, 6);
    bw_bits(w, 0, 1);
    for (int i = 0; i < 2; ++i) bw_bits(w, (uint32_t)s[i], 16);
    residual(w, res, 2, bs);
  }
  free(res);
}

int main(int argc, char** argv) {
  if (argc != 3) {
    fprintf(stderr, "usage: flac_synth in.wav out.flac\n");
    return 2;
  }
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  uint8_t hdr[44];
  if (fread(hdr, 1, 44, f) != 44 || memcmp(hdr, "RIFF", 4) || memcmp(hdr + 36, "data", 4) || hdr[34] != 16) {
    fprintf(stderr, "flac_synth: expects a canonical 16-bit PCM WAV\n");
    return 1;
  }
  const int ch = hdr[22], sr = hdr[24] | (hdr[25] << 8) | (hdr[26] << 16) | (hdr[27] << 24);
  const uint32_t bytes = hdr[40] | (hdr[41] << 8) | (hdr[42] << 16) | ((uint32_t)hdr[43] << 24);
  const int64_t n = bytes / 2 / ch;
  int16_t* pcm = (int16_t*)malloc(bytes);
  if (fread(pcm, 1, bytes, f) != bytes) return 1;
  fclose(f);
  const int BS = 4096;
  BW w = {0};
  bw_bits(&w, 0x664c6143, 32);
  bw_bits(&w, 0x80, 8);
  bw_bits(&w, 34, 24);
  bw_bits(&w, BS, 16);
  bw_bits(&w, BS, 16);
  bw_bits(&w, 0, 24);
  bw_bits(&w, 0, 24);
  bw_bits(&w, (uint32_t)sr, 20);
  bw_bits(&w, (uint32_t)(ch - 1), 3);
  bw_bits(&w, 15, 5);
  bw_bits(&w, (uint32_t)(n >> 32), 4);
  bw_bits(&w, (uint32_t)n, 32);
  for (int i = 0; i < 16; ++i) bw_bits(&w, 0, 8);
  int32_t* s = (int32_t*)malloc(sizeof(int32_t) * BS);
  for (int64_t at = 0, fn = 0; at < n; at += BS, ++fn) {
    const int bs = (int)(n - at < BS ? n - at : BS);
    const size_t start = w.n;
    bw_bits(&w, 0xfff8, 16);
    bw_bits(&w, bs == BS ? 12 : 7, 4); /* 4096, or a 16-bit block size */
    bw_bits(&w, 0, 4);                 /* sample rate from STREAMINFO */
    bw_bits(&w, (uint32_t)(ch - 1), 4);
    bw_bits(&w, 4, 3); /* 16 bits */
    bw_bits(&w, 0, 1);
    utf8(&w, (uint64_t)fn);
    if (bs != BS) bw_bits(&w, (uint32_t)(bs - 1), 16);
    bw_bits(&w, crc8(w.p + start, w.n - start), 8);
    for (int c = 0; c < ch; ++c) {
      for (int i = 0; i < bs; ++i) s[i] = pcm[(at + i) * ch + c];
      if (bs > 16) {
        subframe(&w, s, bs);
      } else { /* verbatim */
        bw_bits(&w, 0, 1);
        bw_bits(&w, 1, 6);
        bw_bits(&w, 0, 1);
        for (int i = 0; i < bs; ++i) bw_bits(&w, (uint32_t)s[i], 16);
      }
    }
    bw_align(&w);
    const uint16_t c16 = crc16(w.p + start, w.n - start);
    bw_bits(&w, c16, 16);
  }
  f = fopen(argv[2], "wb");
  if (!f || fwrite(w.p, 1, w.n, f) != w.n) return 1;
  fclose(f);
  fprintf(stderr, "flac_synth: %lld frames x %d ch @ %d Hz: %u -> %zu bytes\n", (long long)n, ch, sr, bytes, w.n);
  return 0;
}
