/*
 * ORACLE (C restatement) — TEST INFRASTRUCTURE ONLY.  Never linked into or called by basic_pitch_amd/.
 *
 * A plain C (gcc, OpenMP over windows) restatement of the reference's frozen Basic Pitch graph in fp32, used
 *   (1) as a second, independent check of oracle/bp_oracle.py (tests/test_oracle_golden.py), and
 *   (2) as the CPU baseline bench.py times on the GPU box's host cores ("port": the reference's own runtimes —
 *       TensorFlow / onnxruntime / TFLite / CoreML — are not installable here).
 * Reference lines restated (spotify/basic-pitch v0.4.0):
 *   basic_pitch/layers/nnaudio.py:259-284, 636-638   downsampling_by_n (zero-pad 127, 256-tap FIR, stride 2)
 *   basic_pitch/layers/nnaudio.py:216-256, 623-661   get_cqt_complex per octave, bin assembly, sqrt(len), magnitude
 *   basic_pitch/layers/signal.py:171-185, math.py:21-32  NormalizedLog
 *   basic_pitch/models.py:187-189                     BatchNorm (folded affine of the frozen graph)
 *   basic_pitch/nn.py:69-88                           HarmonicStacking (shifts -36,0,36,57,72,84,93,101; crop 264)
 *   basic_pitch/models.py:241-318                     the six Conv2D layers (BN folded), ReLU / sigmoid, concat
 * Weights: the "BPAMDW01" blob of include/basic_pitch_amd.h (tools/extract_weights.py from nmp.onnx).
 * Pinning: agrees with oracle/bp_oracle.py (fp32) to summation-order noise on the golden clip and synthetic
 * windows; that oracle is pinned to the reference's golden posteriorgrams.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define N_AUDIO 43844
#define N_FRAMES 172
#define N_BINS 309
#define N_FC 264
#define N_FN 88
#define N_OCT 9
#define BPO 36

typedef struct {
  const float *k_re, *k_im, *lowpass, *sqrt_len, *log_eps, *log_scale, *bn;
  const float *c1w, *c1b, *c2w, *c2b, *n1w, *n1b, *n2w, *n2b, *o1w, *o1b, *o2w, *o2b;
} weights_t;

static const float* find_tensor(const uint8_t* blob, size_t nbytes, const char* name, uint32_t* count) {
  uint32_t n;
  if (nbytes < 16 || memcmp(blob, "BPAMDW01", 8) != 0) return NULL;
  memcpy(&n, blob + 12, 4);
  const size_t data0 = 16 + (size_t)52 * n;
  for (uint32_t i = 0; i < n; ++i) {
    const uint8_t* e = blob + 16 + (size_t)52 * i;
    char nm[25] = {0};
    memcpy(nm, e, 24);
    if (strcmp(nm, name) == 0) {
      uint32_t off, cnt;
      memcpy(&off, e + 44, 4);
      memcpy(&cnt, e + 48, 4);
      if (count) *count = cnt;
      return (const float*)(blob + data0 + 4 * (size_t)off);
    }
  }
  return NULL;
}

static int load_weights(const uint8_t* blob, size_t nbytes, weights_t* w) {
#define GET(field, name) if (!(w->field = find_tensor(blob, nbytes, name, NULL))) return -1
  GET(k_re, "cqt_kernel_re"); GET(k_im, "cqt_kernel_im"); GET(lowpass, "cqt_lowpass"); GET(sqrt_len, "cqt_sqrt_len");
  GET(log_eps, "log_eps"); GET(log_scale, "log_scale"); GET(bn, "bn_affine");
  GET(c1w, "contour1_w"); GET(c1b, "contour1_b"); GET(c2w, "contour2_w"); GET(c2b, "contour2_b");
  GET(n1w, "note1_w"); GET(n1b, "note1_b"); GET(n2w, "note2_w"); GET(n2b, "note2_b");
  GET(o1w, "onset1_w"); GET(o1b, "onset1_b"); GET(o2w, "onset2_w"); GET(o2b, "onset2_b");
#undef GET
  return 0;
}

/* Conv2D, NCHW, cross-correlation, zero padding (pt, pl given; bottom/right implied by the output size),
 * stride sw along the width.  act: 0 none, 1 ReLU, 2 sigmoid.  in [cin][h][win] -> out [cout][h][wout]. */
static void conv2d(const float* in, int cin, int h, int win, const float* wgt, const float* bias, int cout, int kh,
                   int kw, int pt, int pl, int sw, int wout, int act, float* out, float* pad_buf) {
  const int hp = h + kh - 1, wp = (wout - 1) * sw + kw;  /* padded input extent actually read */
  memset(pad_buf, 0, sizeof(float) * (size_t)cin * hp * wp);
  for (int c = 0; c < cin; ++c)
    for (int t = 0; t < h; ++t) {
      const int n = win < wp - pl ? win : wp - pl;
      memcpy(pad_buf + ((size_t)c * hp + t + pt) * wp + pl, in + ((size_t)c * h + t) * win, sizeof(float) * n);
    }
  enum { VB = 16 };
  for (int o0 = 0; o0 < cout; o0 += 8) {
    const int on = cout - o0 < 8 ? cout - o0 : 8;
    for (int t = 0; t < h; ++t)
      for (int f0 = 0; f0 < wout; f0 += VB) {
        const int fn = wout - f0 < VB ? wout - f0 : VB;
        float acc[8][VB];
        for (int o = 0; o < 8; ++o)
          for (int v = 0; v < VB; ++v) acc[o][v] = 0.0f;
        for (int c = 0; c < cin; ++c)
          for (int dt = 0; dt < kh; ++dt) {
            const float* row = pad_buf + ((size_t)c * hp + t + dt) * wp + (size_t)f0 * sw;
            for (int df = 0; df < kw; ++df) {
              float x[VB];
              if (fn == VB) {
                for (int v = 0; v < VB; ++v) x[v] = row[v * sw + df];
              } else {
                for (int v = 0; v < VB; ++v) x[v] = v < fn ? row[v * sw + df] : 0.0f;
              }
              for (int o = 0; o < on; ++o) {
                const float wv = wgt[(((size_t)(o0 + o) * cin + c) * kh + dt) * kw + df];
                for (int v = 0; v < VB; ++v) acc[o][v] += wv * x[v];
              }
            }
          }
        for (int o = 0; o < on; ++o)
          for (int v = 0; v < fn; ++v) {
            float y = acc[o][v] + bias[o0 + o];
            if (act == 1) y = y > 0.0f ? y : 0.0f;
            if (act == 2) y = 1.0f / (1.0f + expf(-y));
            out[((size_t)(o0 + o) * h + t) * wout + f0 + v] = y;
          }
      }
  }
}

static void forward_one(const weights_t* w, const float* audio, float* note, float* onset, float* contour,
                        float* scratch) {
  /* ---- pyramid (nnaudio.py:269-279) */
  int len[N_OCT];
  float* lvl[N_OCT];
  float* p = scratch;
  len[0] = N_AUDIO;
  lvl[0] = (float*)audio;
  for (int k = 1; k < N_OCT; ++k) {
    len[k] = (len[k - 1] - 2) / 2 + 1;
    lvl[k] = p;
    p += len[k];
    for (int n = 0; n < len[k]; ++n) {
      float s = 0.0f;
      const int base = 2 * n - 127;
      const int j0 = base < 0 ? -base : 0;
      const int j1 = base + 256 > len[k - 1] ? len[k - 1] - base : 256;
      for (int j = j0; j < j1; ++j) s += w->lowpass[j] * lvl[k - 1][base + j];
      lvl[k][n] = s;
    }
  }
  /* ---- filterbank + magnitude -> lp (nnaudio.py:229-254, 640-661; signal.py:174-175) */
  float* lp = p;
  p += N_FRAMES * N_BINS;
  float mn = INFINITY, mx = -INFINITY;
  for (int k = 0; k < N_OCT; ++k) {
    const int hop = 256 >> k, L = len[k];
    for (int t = 0; t < N_FRAMES; ++t) {
      float xp[256];
      for (int i = 0; i < 256; ++i) {
        int g = t * hop + i - 128;
        g = g < 0 ? -g : g;
        g = g >= L ? 2 * (L - 1) - g : g;
        xp[i] = lvl[k][g];
      }
      for (int f = 0; f < BPO; ++f) {
        const int bin = (N_OCT - 1 - k) * BPO + f - 15;
        if (bin < 0) continue;
        float re = 0.0f, im = 0.0f;
        for (int i = 0; i < 256; ++i) {
          re += w->k_re[f * 256 + i] * xp[i];
          im += w->k_im[f * 256 + i] * xp[i];
        }
        re *= w->sqrt_len[bin];
        im = -im * w->sqrt_len[bin];
        const float mag = sqrtf(re * re + im * im);
        const float v = logf(mag * mag + w->log_eps[0]) * w->log_scale[0] * w->log_scale[1];
        lp[t * N_BINS + bin] = v;
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
      }
    }
  }
  /* ---- normalise, BN, harmonic stack (signal.py:177-183, models.py:187-189, nn.py:69-88) */
  static const int shift[8] = {-36, 0, 36, 57, 72, 84, 93, 101};
  const float range = mx - mn;
  float* stack = p;
  p += 8 * N_FRAMES * N_FC;
  for (int c = 0; c < 8; ++c)
    for (int t = 0; t < N_FRAMES; ++t)
      for (int f = 0; f < N_FC; ++f) {
        const int g = f + shift[c];
        float z = 0.0f;
        if (g >= 0 && g < N_BINS) {
          const float off = lp[t * N_BINS + g] - mn;
          z = (range == 0.0f ? 0.0f : off / range) * w->bn[0] + w->bn[1];
        }
        stack[((size_t)c * N_FRAMES + t) * N_FC + f] = z;
      }
  /* ---- CNN (models.py:241-318) */
  float* c1 = p;   p += 8 * N_FRAMES * N_FC;
  float* n1 = p;   p += 32 * N_FRAMES * N_FN;
  float* cat = p;  p += 33 * N_FRAMES * N_FN;
  float* pad = p;  /* largest padded input: 33 x 174 x 90 or 8 x 174 x 302 */
  conv2d(stack, 8, N_FRAMES, N_FC, w->c1w, w->c1b, 8, 3, 39, 1, 19, 1, N_FC, 1, c1, pad);
  conv2d(c1, 8, N_FRAMES, N_FC, w->c2w, w->c2b, 1, 5, 5, 2, 2, 1, N_FC, 2, contour, pad);
  conv2d(contour, 1, N_FRAMES, N_FC, w->n1w, w->n1b, 32, 7, 7, 3, 2, 3, N_FN, 1, n1, pad);
  conv2d(n1, 32, N_FRAMES, N_FN, w->n2w, w->n2b, 1, 7, 3, 3, 1, 1, N_FN, 2, note, pad);
  conv2d(stack, 8, N_FRAMES, N_FC, w->o1w, w->o1b, 32, 5, 5, 2, 1, 3, N_FN, 1, cat + N_FRAMES * N_FN, pad);
  memcpy(cat, note, sizeof(float) * N_FRAMES * N_FN); /* concat channel 0 = post-sigmoid note map (305) */
  conv2d(cat, 33, N_FRAMES, N_FN, w->o2w, w->o2b, 1, 3, 3, 1, 1, 1, N_FN, 2, onset, pad);
}

/* pyramid + lp + (stack, c1) + (n1: 32, cat: 33 channels) + the largest zero-padded conv input (33 x 174 x 90) */
#define SCRATCH_FLOATS (43712 + N_FRAMES * N_BINS + 2 * 8 * N_FRAMES * N_FC + 65 * N_FRAMES * N_FN + 560000)

/* audio [n][43844] -> note/onset [n][172][88], contour [n][172][264]; windows in parallel (OpenMP).
 * Returns 0, -1 for a bad weights blob, -2 for out of memory. */
int bpo_forward(const void* weights_blob, size_t nbytes, const float* audio, int n, float* note, float* onset,
                float* contour, int n_threads) {
  weights_t w;
  if (load_weights((const uint8_t*)weights_blob, nbytes, &w)) return -1;
  int failed = 0;
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
  {
    float* scratch = (float*)malloc(sizeof(float) * (size_t)SCRATCH_FLOATS);
    if (!scratch) {
#pragma omp atomic write
      failed = 1;
    }
#pragma omp for schedule(dynamic, 1)
    for (int b = 0; b < n; ++b)
      if (scratch)
        forward_one(&w, audio + (size_t)b * N_AUDIO, note + (size_t)b * N_FRAMES * N_FN,
                    onset + (size_t)b * N_FRAMES * N_FN, contour + (size_t)b * N_FRAMES * N_FC, scratch);
    free(scratch);
  }
  return failed ? -2 : 0;
}
