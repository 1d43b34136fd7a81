// C ABI of libbasicpitch_amd.so (include/basic_pitch_amd.h): context, weights blob parsing, operand
// packing for the MFMA kernels, HBM workspace and stage orchestration.
//
// Replaces, for the hot path only, what the reference delegates to TensorFlow / onnxruntime /
// TFLite / CoreML behind basic_pitch/inference.py:71-182 (Model) and the window loop of
// run_inference (inference.py:282-315).
#include "../../include/basic_pitch_amd.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "bp_common.h"

namespace bp {
// kernels (one translation unit each)
void launch_pyramid(const float* audio, float* pyr, const float* lowpass, int n_windows, hipStream_t s);
void launch_window_track(const float* samples, int64_t n_samples, int64_t first_window, int n_windows,
                         float* audio, int win_len, int hop, int lead, hipStream_t stream);
void launch_window_tracks(const TrackSegs& ts, int n_slots, float* audio, int win_len, int hop, int lead,
                          hipStream_t stream);
void launch_unwrap_tracks(const TrackSegs& ts, int n_slots, const float* note, const float* onset, const float* contour,
                          hipStream_t stream);
void launch_unwrap3(const float* note, const float* onset, const float* contour, int64_t first_window, int n_windows,
                    int64_t total_rows, float* o_note, float* o_onset, float* o_contour, hipStream_t stream);
void launch_unwrap(const float* win_out, int n_freq, int64_t first_window, int n_windows,
                   int64_t total_rows, float* out, hipStream_t s);
size_t filterbank_scratch_floats(int n_windows);
void launch_filterbank(const float* audio, const float* pyr, const float* bfrag, const float* sqrt_len,
                       float* lp, int* mm, float* scratch, int n_windows, LogConsts kc, int n_cu,
                       hipStream_t s);
void launch_contour1(const float* lp, const int* mm, const float* bfrag, const float* bias, float* c1,
                     int n_windows, LogConsts kc, int n_cu, hipStream_t s);
void launch_onset1(const float* lp, const int* mm, const float* bfrag, const float* bias, float* o1,
                   int n_windows, LogConsts kc, int n_cu, hipStream_t s);
void launch_note1(const float* contour, const float* bfrag, const float* bias, float* n1, int n_windows,
                  int n_cu, hipStream_t s);
void launch_contour2(const float* c1, const float* wgt, float bias, float* contour, int n_windows,
                     hipStream_t s);
void launch_note2(const float* n1, const float* wgt, float bias, float* note, int n_windows,
                  hipStream_t s);
void launch_onset2(const float* note, const float* o1, const float* wgt, float bias, float* onset,
                   int n_windows, hipStream_t s);
void launch_zpack_partials(const float* lp, const float* scratch, int n_partials, uint32_t* zp, int n_windows,
                           LogConsts kc, int n_bins, hipStream_t stream);
// cqt_planes.hip: the pyramid as pre-split, reflect-padded f16 planes; operands straight from HBM / L2
int64_t planes_elements_per_window(bool ext);
void launch_planes_split(const float* src, int64_t src_stride, int level, uint16_t* pl, int n_windows, bool ext,
                         hipStream_t stream);
void launch_planes_unsplit(const uint16_t* pl, int level, float* dst, int64_t dst_stride, int n_windows, bool ext,
                           hipStream_t stream);
void launch_planes_edge_rows(const float* audio, int64_t audio_stride, uint16_t* pl, int n_windows, bool ext,
                             hipStream_t stream);
void launch_pyramid_planes(const float* audio, int64_t audio_stride, uint16_t* pl, const void* tfrag, int n_windows,
                           int n_cu, bool ext, hipStream_t stream);
int filterbank_planes_partials(bool ext);
void launch_mm_reduce(const float* scratch, int* mm, int n_windows, int n_partials, hipStream_t stream);
bool launch_filterbank_planes(const uint16_t* pl, const float* audio, int64_t audio_stride, const void* bfrag,
                              const float* bin_consts, float* lp, float* scratch,
                              uint32_t* zp, int n_windows, LogConsts kc, int n_cu, bool ext, hipStream_t stream);
void filterbank_planes_bin_consts(const float* sqrt_len, int n_bins, LogConsts kc, float* out);
void launch_zpack(const float* lp, const int* mm, uint32_t* zp, int n_windows, LogConsts kc, int n_bins,
                  hipStream_t s);
ResamplePlan make_resample_plan(int source_rate, int target_rate, std::vector<double>& taps);
void launch_downmix(const float* pcm, int64_t n_frames, int channels, float* mono, hipStream_t stream);
void launch_downmix_raw(const void* raw, int format, int64_t n_frames, int channels, float* mono, hipStream_t stream);
void launch_resample(const float* x, int64_t n_in, const double* taps, const ResamplePlan& pl, float* y,
                     int64_t n_out, int mode, hipStream_t stream);
#ifdef BP_AB_KERNELS  // conv_contour_direct.hip: the exact 8-channel and the round-2 folded conv1 (A/B builds only)
void launch_contour_conv1_exact(const uint32_t* zp, const void* wlds, const float* bias, float* c1, int n_windows,
                                int n_cu, bool weights_have_lo, hipStream_t stream);
void launch_contour_conv1_folded(const uint32_t* zp, const void* wfold, const float* bias, float* c1, int n_windows,
                                 int n_cu, bool weights_have_lo, hipStream_t stream);
bool contour_conv1_full();
#else
static inline bool contour_conv1_full() { return false; }
#endif
bool contour_conv1_use_march();
void launch_contour_conv1_march(const uint32_t* zp, const void* wfrag, const float* bias, float* c1, int n_windows, int n_cu,
                                bool weights_have_lo, hipStream_t stream);
void launch_contour_conv1_rim(const uint32_t* zp, const void* afrag, const float* bias, float* c1, int n_windows, int n_cu,
                              bool weights_have_lo, bool ext, hipStream_t stream);
void launch_contour_conv1_rim_march(const uint32_t* zp, const void* afrag, const float* bias, float* c1, int n_windows, int n_cu,
                                    bool weights_have_lo, hipStream_t stream);
#ifdef BP_AB_KERNELS  // onset_march.hip: the 32x32x16 form of the onset march (A/B builds only)
void launch_onset_march(const uint32_t* zp, const float* note, const void* wfrag, const float* wf32, float* onset,
                        int n_windows, int n_cu, bool weights_have_lo, hipStream_t stream);
#endif
void launch_onset_march16(const uint32_t* zp, const float* note, const void* wfrag, const float* wf32, float* onset,
                          int n_windows, int n_cu, bool weights_have_lo, hipStream_t stream);
#ifdef BP_AB_KERNELS  // conv_contour_fold_mx.hip: the fp8-correction mode's contour conv1 (A/B builds only since round 6)
void launch_contour_conv1_fold_mx(const uint32_t* zp, const void* a16, const void* amx, const void* ascale,
                                  const float* bias, float* c1, int n_windows, int n_cu, hipStream_t stream);
#endif
#ifdef BP_AB_KERNELS  // conv_contour2.hip: the round-2 vector kernel (BP_CONV2=valu; A/B builds only)
void launch_contour_conv2(const float* c1, const float* w2, float bias, float* contour, int n_windows, int n_cu,
                          hipStream_t stream);
#endif
void launch_contour_conv2_proj(const float* c1, const void* wfrag, float bias, float* contour, int n_windows, int n_cu,
                               bool weights_have_lo, hipStream_t stream);
// flac_device.hip
struct FdStream {
  int channels, bits, min_block, max_block;
  int64_t total;
  uint32_t audio_start, nbytes;
};
struct FlacDeviceBuffers {
  uint8_t* file = nullptr;
  size_t file_cap = 0;
  void* cands = nullptr;
  uint32_t* counts = nullptr;
  size_t cands_cap = 0, counts_cap = 0;
  void* packed = nullptr;
  uint32_t* offs = nullptr;
  size_t packed_cap = 0, offs_cap = 0;
  void* frames = nullptr;
  int32_t* scratch = nullptr;
  size_t frames_cap = 0, scratch_cap = 0;
  int* meta = nullptr;
  uint16_t* crc_tab = nullptr;
};
int flac_device_decode(FlacDeviceBuffers& b, const FdStream& st, void* d_pcm, hipStream_t stream);
void flac_device_free(FlacDeviceBuffers& b);
#ifdef BP_AB_KERNELS  // note_march.hip: the 32x32x16 form of the note march (A/B builds only)
void launch_note_march(const float* contour, const void* wfrag, const float* wf32, float* note, int n_windows,
                       bool weights_have_lo, hipStream_t stream);
#endif
void launch_note_march16(const float* contour, const void* wfrag, const float* wf32, float* note, int n_windows, int n_cu,
                         bool weights_have_lo, hipStream_t stream);
void launch_note_candidates(float* note, float* onset, const float* contour, int64_t T, int lo, int hi, int infer,
                            double onset_thresh, const void* tab, const double* gauss, void* stats, uint8_t* bits,
                            int8_t* bend, hipStream_t s);
void launch_note_export(const void* note, void* note_dst, int64_t note_bytes, const void* bits, void* bits_dst,
                        int64_t bits_bytes, const void* bend, void* bend_dst, int64_t bend_bytes, void* stats,
                        void* stats_dst, hipStream_t s);
void launch_note_stats_init(void* stats, hipStream_t s);
#ifdef BP_AB_KERNELS
void launch_onset_branch(const uint32_t* zp, const float* note, const void* wfrag, const float* wf32, const void* wmx,
                         float* onset, int n_windows, int n_cu, bool weights_have_lo, hipStream_t stream);
#endif
// the onset branch: the wave-private march on 16x16x32.  A/B builds only: the workgroup kernel for the fp8-correction
// mode (it carries the block-scaled products), BP_ONSET=march32 selects the 32x32x16 form of the march, BP_ONSET=ring the
// workgroup kernel without fp8.
static void launch_onset(const uint32_t* zp, const float* note, const void* wfrag, const float* wf32, const void* wmx,
                         const void* w16, float* onset, int n_windows, int n_cu, bool weights_have_lo, hipStream_t stream) {
  int kind = 0;
#ifdef BP_AB_KERNELS
  static const int env_kind = [] {
    const char* e = ab_env("BP_ONSET");
    return e && std::strcmp(e, "ring") == 0 ? 2 : (e && std::strcmp(e, "march32") == 0 ? 1 : 0);
  }();
  kind = env_kind;
  if (kind == 1 && !wmx) {
    launch_onset_march(zp, note, wfrag, wf32, onset, n_windows, n_cu, weights_have_lo, stream);
    return;
  }
  if (wmx || kind == 2) {
    launch_onset_branch(zp, note, wfrag, wf32, wmx, onset, n_windows, n_cu, weights_have_lo, stream);
    return;
  }
#endif
  (void)wfrag, (void)wmx, (void)kind;
  launch_onset_march16(zp, note, w16, wf32, onset, n_windows, n_cu, weights_have_lo, stream);
}
// contour conv2: the tap projection on the matrix cores (round 6).  A/B builds only: BP_CONV2=valu selects the round-2 kernel.
static void launch_conv2(const float* c1, const float* w2, const void* wproj, float bias, float* contour, int n_windows,
                         int n_cu, bool weights_have_lo, hipStream_t stream) {
#ifdef BP_AB_KERNELS
  static const bool valu = [] {
    const char* e = ab_env("BP_CONV2");
    return e && std::strcmp(e, "valu") == 0;
  }();
  if (valu) {
    launch_contour_conv2(c1, w2, bias, contour, n_windows, n_cu, stream);
    return;
  }
#endif
  (void)w2;
  launch_contour_conv2_proj(c1, wproj, bias, contour, n_windows, n_cu, weights_have_lo, stream);
}
// the note branch: the wave-private march on 16x16x32 (round 6).  A/B builds only: BP_NOTE=march32 selects the 32x32x16 form.
static void launch_note(const float* contour, const void* wfrag, const void* w16, const float* wf32, float* note, int n_windows,
                        int n_cu, bool weights_have_lo, hipStream_t stream) {
#ifdef BP_AB_KERNELS
  static const bool march32 = [] {
    const char* e = ab_env("BP_NOTE");
    return e && std::strcmp(e, "march32") == 0;
  }();
  if (march32) {
    launch_note_march(contour, wfrag, wf32, note, n_windows, weights_have_lo, stream);
    return;
  }
#endif
  (void)wfrag;
  launch_note_march16(contour, w16, wf32, note, n_windows, n_cu, weights_have_lo, stream);
}
}  // namespace bp

using namespace bp;

namespace {

thread_local std::string g_create_error;

struct Tensor {
  const float* data = nullptr;
  uint32_t ndim = 0, dims[4] = {1, 1, 1, 1}, count = 0;
};

struct Blob {
  std::vector<std::pair<std::string, Tensor>> t;
  const Tensor* find(const char* name) const {
    for (auto& kv : t)
      if (kv.first == name) return &kv.second;
    return nullptr;
  }
};

bool parse_blob(const void* weights, size_t nbytes, Blob& out, std::string& err) {
  const uint8_t* p = static_cast<const uint8_t*>(weights);
  if (!p || nbytes < 16 || std::memcmp(p, "BPAMDW01", 8) != 0) {
    err = "weights blob: bad magic (expected BPAMDW01)";
    return false;
  }
  uint32_t version, n;
  std::memcpy(&version, p + 8, 4);
  std::memcpy(&n, p + 12, 4);
  if (version != 1 || n > 1024 || nbytes < 16 + (size_t)52 * n) {
    err = "weights blob: bad version or truncated directory";
    return false;
  }
  const size_t data0 = 16 + (size_t)52 * n;
  for (uint32_t i = 0; i < n; ++i) {
    const uint8_t* e = p + 16 + (size_t)52 * i;
    char name[25] = {0};
    std::memcpy(name, e, 24);
    Tensor t;
    std::memcpy(&t.ndim, e + 24, 4);
    std::memcpy(t.dims, e + 28, 16);
    uint32_t off;
    std::memcpy(&off, e + 44, 4);
    std::memcpy(&t.count, e + 48, 4);
    if (t.ndim > 4 || data0 + 4 * ((size_t)off + t.count) > nbytes) {
      err = std::string("weights blob: tensor out of bounds: ") + name;
      return false;
    }
    t.data = reinterpret_cast<const float*>(p + data0 + 4 * (size_t)off);
    out.t.emplace_back(name, t);
  }
  return true;
}

bool expect(const Blob& b, const char* name, std::initializer_list<uint32_t> shape, const Tensor*& t,
            std::string& err) {
  t = b.find(name);
  if (!t) {
    err = std::string("weights blob: missing tensor ") + name;
    return false;
  }
  uint32_t cnt = 1;
  uint32_t i = 0;
  for (uint32_t d : shape) {
    if (i >= t->ndim || t->dims[i] != d) {
      err = std::string("weights blob: wrong shape for ") + name;
      return false;
    }
    cnt *= d;
    ++i;
  }
  if (i != t->ndim || cnt != t->count) {
    err = std::string("weights blob: wrong rank/count for ") + name;
    return false;
  }
  return true;
}

}  // namespace

struct bp_context {
  int device = 0;
  unsigned flags = 0;
  int n_cu = 256;
  char arch[32] = {0};
  hipStream_t own_stream = nullptr, stream = nullptr;
  hipEvent_t done = nullptr;  // BP_FLAG_BLOCKING_WAIT: the event a waiting host thread sleeps on
  int64_t cap = 0;
  int64_t workspace_bytes = 0;
  std::string err;

  // window geometry: the reference's 22.05 kHz model, or the extended 44.1 kHz range (BP_FLAG_EXT_CQT_44K)
  bool ext = false;
  int win_len = kAudioN, hop = BP_HOP_SIZE, lead = BP_OVERLAP_LEN / 2, n_bins = kBins, rate = BP_AUDIO_SAMPLE_RATE;
  int64_t pyr_stride = kPyrStride;

  LogConsts kc{};
  float b_contour2 = 0, b_note2 = 0, b_onset2 = 0;
  // device constants
  float *d_lowpass = nullptr, *d_sqrt_len = nullptr, *d_fb_bfrag = nullptr;
  float* d_pl_bin_k = nullptr;  // cqt_planes.hip filterbank: per-bin eps / s^2, s = sqrt(len) 2^-12
  // fused branches (conv_branch.hip): f16 hi/lo A fragments (raw bytes) + {bias1[32], extra[9], bias2}
  float *d_note_wfrag = nullptr, *d_note_w16 = nullptr, *d_note_wf32 = nullptr, *d_onset_wfrag = nullptr, *d_onset_wf32 = nullptr,
        *d_onset_wmx = nullptr, *d_onset_w16 = nullptr;
  float* zp = nullptr;  // uint32 [cap][kZRowsP][kZRow] pre-split z, zero padded (bp_common.h)
  // contour branch, two-kernel form (conv_contour_direct.hip): LDS weight image, bias[8], conv2 taps [5][5][8]
  float *d_d1_wlds = nullptr, *d_d1_wfold = nullptr, *d_d1_wmarch = nullptr, *d_d1_wrim = nullptr, *d_d1_wrimm = nullptr, *d_d1_bias = nullptr,
        *d_d2_w = nullptr, *d_d2_wproj = nullptr;
  bool rim_exact = false, fold_mx = false;
  int resample_mode = 0;  // BP_RESAMPLE=plain|tiled: 1 | 2 (A/B runs of the resampling kernels)
  int contour_parts = 0;  // BP_CONTOUR_PARTS (0: automatic)
  float* d_d1_wfold_mx = nullptr;
  float* c1s = nullptr;  // [cap][172][kC1Row][8] relu(conv1); pad bins zeroed once at allocation
  // cqt_planes.hip: decimator / filterbank fragments (raw bytes of f16 hi / lo), the planes of a chunk [cap][2][stride] f16
  float *d_pl_tfrag = nullptr, *d_pl_bfrag = nullptr, *planes = nullptr;
  float *d_c1_bfrag = nullptr, *d_c1_bias = nullptr, *d_o1_bfrag = nullptr, *d_o1_bias = nullptr;
  float *d_n1_bfrag = nullptr, *d_n1_bias = nullptr, *d_w_contour2 = nullptr, *d_w_note2 = nullptr,
        *d_w_onset2 = nullptr;
  // workspace (per chunk of `cap` windows)
  float *audio = nullptr, *pyr = nullptr, *lp = nullptr, *c1 = nullptr, *contour = nullptr, *n1 = nullptr,
        *note = nullptr, *o1 = nullptr, *onset = nullptr;
  int* mm = nullptr;
  float* fb_scratch = nullptr;  // filterbank partial extrema (grow-only; >= cap windows)
  int64_t fb_scratch_windows = 0;
  // track path staging (grow-only)
  float* track = nullptr;
  int64_t track_cap = 0;
  // audio ingest (audio_ingest.hip): staging for PCM / mono / 22.05 kHz signal (grow-only), cached filter
  float *pcm_dev = nullptr, *mono_dev = nullptr, *res_dev = nullptr;
  int64_t pcm_cap = 0, mono_cap = 0, res_cap = 0;
  double* taps_dev = nullptr;
  int taps_rate = 0;
  ResamplePlan plan{};
  float* track_out = nullptr;  // [T, 88+88+264] staging when outputs are host pointers
  int64_t track_out_cap = 0;
  int64_t maps_rows = 0;       // rows of the maps a *_candidates call left in track_out (bp_track_maps); 0: none
  // device-side note candidates (note_device.hip): bitmap [T][11] + bend map [T][88] (bytes), stats, the bend tables
  float* nd_buf = nullptr;
  int64_t nd_cap = 0;          // floats
  float* nd_tables = nullptr;  // [88] int4 windows, [51] double Gaussian, then the stats record
  FlacDeviceBuffers fd;            // flac_device.hip: the file's bytes, the frame lists, the scratch rows
  int* fd_status_host = nullptr;   // page-locked: the device decoder's error bits of the last call
  float* nd_stats_host = nullptr;  // page-locked copy of the stats record
  void* nd_stats_host_dev = nullptr;  // the same buffer as the device sees it
  bool nd_stats_ready = false;     // the device record holds its initial values (the export kernel leaves it so)

  // stage timing: a ring of event sets, one per chunk, averaged by bp_get_stage_ms
  static constexpr int kTimedRing = 128;
  static constexpr int kDomEvery = 4;
  static constexpr int kMaxMarks = 32;  // a stage may be launched in parts (the contour branch): its intervals are summed
  hipEvent_t ev[kTimedRing][kMaxMarks + 1] = {};
  // per ring slot (the mark sequence depends on the chunk: zpack only below half a window per CU, contour parts):
  // stage id of the interval between ev[c][i] and ev[c][i+1]; -1: not a stage (skipped)
  int seq[kTimedRing][kMaxMarks] = {};
  int n_seq[kTimedRing] = {};
  bool ev_valid = false;
  int64_t timed_chunks = 0;  // chunks recorded since the last bp_get_stage_ms
  int64_t dom_chunks = 0;    // chunks seen in BP_FLAG_TIME_DOMINANT mode (every kDomEvery-th is recorded)
};

#define BP_HIP(call)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      char buf_[512];                                                                      \
      std::snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                    __FILE__, __LINE__);                                                   \
      h->err = buf_;                                                                       \
      return (e_ == hipErrorOutOfMemory) ? BP_ERR_OUT_OF_MEMORY : BP_ERR_HIP;              \
    }                                                                                      \
  } while (0)

namespace {

int upload(bp_handle h, const std::vector<float>& host, float** dev) {
  BP_HIP(hipMalloc(dev, host.size() * sizeof(float)));
  BP_HIP(hipMemcpy(*dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  h->workspace_bytes += host.size() * sizeof(float);
  return BP_OK;
}

int alloc(bp_handle h, float** p, int64_t floats) {
  BP_HIP(hipMalloc(p, floats * sizeof(float)));
  h->workspace_bytes += floats * sizeof(float);
  return BP_OK;
}

// ---- operand packing -----------------------------------------------------------------------
// Filterbank B fragments [4 roles][55 steps][64 lanes] (cqt_filterbank.hip roles; 16x16x4: lane ->
// B[k = lane >> 4][n = lane & 15]).
bool pack_filterbank(const Tensor* re, const Tensor* im, std::vector<float>& out, std::string& err) {
  // verify the clipped K ranges cover every non-zero tap
  for (int f = 0; f < 36; ++f) {
    const int lo = f < 16 ? 20 : f < 32 ? 48 : 68, hi = f < 16 ? 236 : f < 32 ? 208 : 188;
    for (int i = 0; i < 256; ++i) {
      if ((i < lo || i >= hi) && (re->data[f * 256 + i] != 0.f || im->data[f * 256 + i] != 0.f)) {
        err = "CQT kernel support exceeds the tap ranges this build is specialised for";
        return false;
      }
    }
  }
  out.assign(4 * 55 * 64, 0.f);
  for (int role = 0; role < 4; ++role) {
    for (int j = 0; j < 55; ++j) {
      for (int lane = 0; lane < 64; ++lane) {
        const int kk = lane >> 4, n = lane & 15;
        float v = 0.f;
        if (role < 2) {
          if (j < 54) {
            const int tap = 4 * (5 + j) + kk;
            v = (role == 0 ? re : im)->data[n * 256 + tap];
          }
        } else {
          const Tensor* main = (role == 2) ? re : im;
          if (j < 40) {
            const int tap = 4 * (12 + j) + kk;
            v = main->data[(16 + n) * 256 + tap];
          } else {
            const int s = (role == 2 ? 17 : 32) + (j - 40);
            const int tap = 4 * s + kk;
            if (n < 4)
              v = re->data[(32 + n) * 256 + tap];
            else if (n < 8)
              v = im->data[(32 + n - 4) * 256 + tap];
          }
        }
        out[((size_t)role * 55 + j) * 64 + lane] = v;
      }
    }
  }
  return true;
}

// IEEE binary16 <-> binary32 on the host (round to nearest even; inputs here are |x| < 8, no inf/nan)
uint16_t f32_to_f16(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  const int32_t exp = (int32_t)((x >> 23) & 0xff) - 127 + 15;
  uint32_t man = x & 0x7fffffu;
  if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (man ? 0x200u : 0));
  if (exp >= 31) return (uint16_t)(sign | 0x7c00u);
  if (exp <= 0) {
    if (exp < -10) return (uint16_t)sign;
    man |= 0x800000u;
    const int shift = 14 - exp;  // 14..24
    uint32_t half = man >> shift;
    const uint32_t rem = man & ((1u << shift) - 1), mid = 1u << (shift - 1);
    if (rem > mid || (rem == mid && (half & 1))) ++half;
    return (uint16_t)(sign | half);
  }
  uint32_t half = ((uint32_t)exp << 10) | (man >> 13);
  const uint32_t rem = man & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (half & 1))) ++half;  // may carry into the exponent: correct
  return (uint16_t)(sign | half);
}

float f16_to_f32(uint16_t hv) {
  const uint32_t sign = (uint32_t)(hv & 0x8000u) << 16;
  uint32_t exp = (hv >> 10) & 0x1f, man = hv & 0x3ffu, x;
  if (exp == 0) {
    if (man == 0) {
      x = sign;
    } else {
      int e = -1;
      do {
        ++e;
        man <<= 1;
      } while (!(man & 0x400u));
      x = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
    }
  } else if (exp == 31) {
    x = sign | 0x7f800000u | (man << 13);
  } else {
    x = sign | ((exp - 15 + 127) << 23) | (man << 13);
  }
  float f;
  std::memcpy(&f, &x, 4);
  return f;
}

// x = hi + lo / lo_scale: lo_scale > 1 keeps the residual inside f16's normal range (cqt_mfma.hip)
void put_split(std::vector<uint16_t>& out, size_t hi_base, size_t lo_base, size_t idx, float v,
               float lo_scale = 1.0f) {
  const uint16_t hi = f32_to_f16(v);
  out[hi_base + idx] = hi;
  out[lo_base + idx] = f32_to_f16((v - f16_to_f32(hi)) * lo_scale);
}

// Two-kernel contour branch (conv_contour_direct.hip): LDS weight image [hi | lo][3 dt][45 taps][8 o] x (8 c) f16,
// tap slot = df + 3 (three zero taps either side: the Toeplitz expansion is done by addressing).
void pack_contour_direct(const Tensor* w1, std::vector<uint16_t>& out) {
  const size_t half = (size_t)3 * 45 * 8 * 8;
  out.assign(2 * half, 0);
  for (int dt = 0; dt < 3; ++dt)
    for (int df = 0; df < 39; ++df)
      for (int o = 0; o < 8; ++o)
        for (int c = 0; c < 8; ++c) {
          const float v = w1->data[((o * 8 + c) * 3 + dt) * 39 + df];
          put_split(out, 0, half, (((size_t)dt * 45 + df + 3) * 8 + o) * 8 + c, v, 2048.0f);
        }
}

// Folded contour conv1 (conv_contour_direct.hip, interior groups): the 8 stack channels are shifted copies of one
// image, so K[o][dt][g] = sum_c W1[o][c][dt][g - s_c + 19], g in [-55, 120].  A fragments [3 dt][12 k-steps][hi|lo]
// [64 lanes] x (8 x f16): lane (row i = 8 j + o, half kh), element el -> tap' = 16 e + 8 kh + el, g = tap' - j - 56.
void pack_contour_folded(const Tensor* w1, std::vector<uint16_t>& out) {
  static const int shifts[8] = {-36, 0, 36, 57, 72, 84, 93, 101};  // nn.py:51-54 (bp_common.h harm_shift)
  std::vector<double> keff((size_t)8 * 3 * 176, 0.0);               // [o][dt][g + 55]
  for (int o = 0; o < 8; ++o)
    for (int c = 0; c < 8; ++c)
      for (int dt = 0; dt < 3; ++dt)
        for (int df = 0; df < 39; ++df)
          keff[((size_t)o * 3 + dt) * 176 + (df - 19 + shifts[c] + 55)] += (double)w1->data[((o * 8 + c) * 3 + dt) * 39 + df];
  out.assign((size_t)36 * 2 * 64 * 8, 0);
  for (int dt = 0; dt < 3; ++dt)
    for (int e = 0; e < 12; ++e)
      for (int lane = 0; lane < 64; ++lane) {
        const int kh = lane >> 5, i = lane & 31, j = i >> 3, o = i & 7;
        const size_t base_hi = (((size_t)(dt * 12 + e) * 2 + 0) * 64 + lane) * 8;
        const size_t base_lo = (((size_t)(dt * 12 + e) * 2 + 1) * 64 + lane) * 8;
        for (int el = 0; el < 8; ++el) {
          const int g = 16 * e + 8 * kh + el - j - 56;
          const float v = (g >= -55 && g <= 120) ? (float)keff[((size_t)o * 3 + dt) * 176 + g + 55] : 0.0f;
          put_split(out, base_hi, base_lo, el, v, 2048.0f);
        }
      }
}

// The same folded kernel for the vertical march (conv_contour_march.hip): M = 16 rows = (2-bin offset j, out channel o),
// a position is a pair of bins, K = 6 k-steps of 32 taps per frame tap.  A fragments [3 dt][6 k-steps][hi|lo][64 lanes] x
// (8 x f16): lane (row i = 8 j + o = lane & 15, gq = lane >> 4), element el -> tap' = 32 s + 8 gq + el, g = tap' - j - 56.
void pack_contour_march(const Tensor* w1, std::vector<uint16_t>& out) {
  static const int shifts[8] = {-36, 0, 36, 57, 72, 84, 93, 101};  // nn.py:51-54 (bp_common.h harm_shift)
  std::vector<double> keff((size_t)8 * 3 * 176, 0.0);               // [o][dt][g + 55]
  for (int o = 0; o < 8; ++o)
    for (int c = 0; c < 8; ++c)
      for (int dt = 0; dt < 3; ++dt)
        for (int df = 0; df < 39; ++df)
          keff[((size_t)o * 3 + dt) * 176 + (df - 19 + shifts[c] + 55)] += (double)w1->data[((o * 8 + c) * 3 + dt) * 39 + df];
  out.assign((size_t)18 * 2 * 64 * 8, 0);
  for (int dt = 0; dt < 3; ++dt)
    for (int s = 0; s < 6; ++s)
      for (int lane = 0; lane < 64; ++lane) {
        const int gq = lane >> 4, i = lane & 15, j = i >> 3, o = i & 7;
        const size_t base_hi = (((size_t)(dt * 6 + s) * 2 + 0) * 64 + lane) * 8;
        const size_t base_lo = (((size_t)(dt * 6 + s) * 2 + 1) * 64 + lane) * 8;
        for (int el = 0; el < 8; ++el) {
          const int g = 32 * s + 8 * gq + el - j - 56;
          const float v = (g >= -55 && g <= 120) ? (float)keff[((size_t)o * 3 + dt) * 176 + g + 55] : 0.0f;
          put_split(out, base_hi, base_lo, el, v, 2048.0f);
        }
      }
}

// Rim of the contour conv1 as a dense GEMM (conv_contour_rim.hip): per side (low rim f in [0, 20), high rim
// f in [244, 264)) the position-dependent folded kernel K[(f, o)][dt][j] over the z bins j0 + [0, 144):
//   K = sum over (c, df) with stack bin f + df - 19 inside [0, 264) (nn.py:87 crops the stack to 264 bins, the
//   convolution zero-pads THAT) and z bin f + df - 19 + shift_c == j0 + j of W1[o][c][dt][df].
// A fragments [side][M block 5][k-step 27 = dt * 9 + e][hi|lo][64 lanes][8]: lane (i = lane & 31 = 8 (f % 4) + o,
// kh = lane >> 5), element el: j = 16 e + 8 kh + el.
// `n_bins`: bins of the CQT (309; 345 for the extended 44.1 kHz mode, whose bins 309..344 reach the high rim); `kJ`: z bins a
// side's window holds (144; 160 for the extended mode: the kernel's RimGeo<160>).
void pack_contour_rim(const Tensor* w1, std::vector<uint16_t>& out, int n_bins = 309, int kJ = 144) {
  static const int shifts[8] = {-36, 0, 36, 57, 72, 84, 93, 101};
  const int kStepsDt = kJ / 16;
  out.assign((size_t)2 * 5 * 3 * kStepsDt * 2 * 64 * 8, 0);
  for (int side = 0; side < 2; ++side) {
    const int f0 = side ? 244 : 0, j0 = side ? (kJ == 144 ? 184 : 188) : 0;  // conv_contour_rim.hip RimGeo::j0
    std::vector<double> k((size_t)20 * 8 * 3 * kJ, 0.0);  // [f_local][o][dt][j]
    for (int fl = 0; fl < 20; ++fl)
      for (int o = 0; o < 8; ++o)
        for (int c = 0; c < 8; ++c)
          for (int dt = 0; dt < 3; ++dt)
            for (int df = 0; df < 39; ++df) {
              const int sb = f0 + fl + df - 19;  // stack bin this tap reads
              if (sb < 0 || sb >= 264) continue;
              const int j = sb + shifts[c] - j0;  // z bin (zero outside [0, 309): nothing to add there)
              const int zb = sb + shifts[c];
              if (zb < 0 || zb >= n_bins) continue;
              if (j < 0 || j >= kJ) {  // cannot happen with the windows above
                std::fprintf(stderr, "pack_contour_rim: z bin %d outside the side's window\n", zb);
                std::abort();
              }
              k[(((size_t)fl * 8 + o) * 3 + dt) * kJ + j] += (double)w1->data[((o * 8 + c) * 3 + dt) * 39 + df];
            }
    for (int mb = 0; mb < 5; ++mb)
      for (int dt = 0; dt < 3; ++dt)
        for (int e = 0; e < kStepsDt; ++e)
          for (int lane = 0; lane < 64; ++lane) {
            const int kh = lane >> 5, i = lane & 31, fl = 4 * mb + (i >> 3), o = i & 7;
            const size_t step = ((size_t)(side * 5 + mb) * 3 * kStepsDt + dt * kStepsDt + e);
            const size_t base_hi = ((step * 2 + 0) * 64 + lane) * 8, base_lo = ((step * 2 + 1) * 64 + lane) * 8;
            for (int el = 0; el < 8; ++el) {
              const int j = 16 * e + 8 * kh + el;
              put_split(out, base_hi, base_lo, el, (float)k[(((size_t)fl * 8 + o) * 3 + dt) * kJ + j], 2048.0f);
            }
          }
  }
}

// The same dense per-side matrix for the register-resident rim kernel (conv_contour_rim_march.hip, 309-bin CQT): M blocks of
// 16 rows = (2 bins x 8 channels), K = (dt, j) flattened = 432 -> 14 k-steps of 32 (zeros behind 432).  A fragments
// [side][block 10][k-step 14][hi|lo][64 lanes][8]: lane (row i = lane & 15 = 8 (f % 2) + o, g = lane >> 4), element el:
// k = 32 s + 8 g + el.
void pack_contour_rim_march(const Tensor* w1, std::vector<uint16_t>& out) {
  static const int shifts[8] = {-36, 0, 36, 57, 72, 84, 93, 101};
  constexpr int kJ = 144, kSteps = (3 * kJ + 31) / 32, n_bins = 309;
  out.assign((size_t)2 * 10 * kSteps * 2 * 64 * 8, 0);
  for (int side = 0; side < 2; ++side) {
    const int f0 = side ? 244 : 0, j0 = side ? 184 : 0;
    std::vector<double> k((size_t)20 * 8 * 3 * kJ, 0.0);  // [f_local][o][dt][j], as pack_contour_rim
    for (int fl = 0; fl < 20; ++fl)
      for (int o = 0; o < 8; ++o)
        for (int c = 0; c < 8; ++c)
          for (int dt = 0; dt < 3; ++dt)
            for (int df = 0; df < 39; ++df) {
              const int sb = f0 + fl + df - 19;
              if (sb < 0 || sb >= 264) continue;
              const int zb = sb + shifts[c], j = zb - j0;
              if (zb < 0 || zb >= n_bins) continue;
              if (j < 0 || j >= kJ) {
                std::fprintf(stderr, "pack_contour_rim_march: z bin %d outside the side's window\n", zb);
                std::abort();
              }
              k[(((size_t)fl * 8 + o) * 3 + dt) * kJ + j] += (double)w1->data[((o * 8 + c) * 3 + dt) * 39 + df];
            }
    for (int mb = 0; mb < 10; ++mb)
      for (int s = 0; s < kSteps; ++s)
        for (int lane = 0; lane < 64; ++lane) {
          const int g = lane >> 4, i = lane & 15, fl = 2 * mb + (i >> 3), o = i & 7;
          const size_t step = (size_t)(side * 10 + mb) * kSteps + s;
          const size_t base_hi = ((step * 2 + 0) * 64 + lane) * 8, base_lo = ((step * 2 + 1) * 64 + lane) * 8;
          for (int el = 0; el < 8; ++el) {
            const int kk = 32 * s + 8 * g + el;
            const float v = kk < 3 * kJ ? (float)k[(((size_t)fl * 8 + o) * 3 + kk / kJ) * kJ + kk % kJ] : 0.0f;
            put_split(out, base_hi, base_lo, el, v, 2048.0f);
          }
        }
  }
}

// ---- block-scaled fp8 (OCP e4m3fn, as gfx950's v_mfma_scale_f32_*_f8f6f4 reads it) for correction products ----
// encode v / 2^e to e4m3fn, round to nearest even, saturating at +-448 (no infinities in the format)
static uint8_t f32_to_e4m3(double v) {
  const uint8_t sign = v < 0 ? 0x80 : 0;
  double a = std::fabs(v);
  if (!(a > 0)) return sign;
  if (a >= 448.0) return sign | 0x7E;
  int ex;
  (void)std::frexp(a, &ex);  // a = m * 2^ex, m in [0.5, 1)
  int e = ex - 1;             // a = 1.x * 2^e
  if (e < -6) e = -6;         // subnormal range shares the exponent of the smallest normal
  const double q = std::nearbyint(a / std::ldexp(1.0, e - 3));  // units of 2^(e-3): 8..15 normal, 0..7 subnormal
  int m = (int)q;
  if (m >= 16) {
    m = 8;
    ++e;
  }
  if (e > 8) return sign | 0x7E;
  if (m < 8) return sign | (uint8_t)m;  // subnormal (e == -6)
  return sign | (uint8_t)(((e + 7) << 3) | (m - 8));
}

// Folded conv1 with fp8 corrections (conv_contour_fold_mx.hip, the default).  The Toeplitz-expanded folded kernel of
// pack_contour_folded — row i = (j = i >> 3, o = i & 7), tap' = 16 e + 8 kh + el  ->  keff[o][dt][tap' - j - 1] — as
//   a16     [36 steps][64 lanes][8] f16: the hi part;
//   mx      [18 steps][64 lanes][32 B]: for the 16 taps tap' = 32 e + 16 kh .. + 15 of step S = 6 dt + e:
//           bytes 0..15 fp8(lo_w) (K block 0 of the instruction), bytes 16..31 fp8(hi_w) (K block 1); lo_w = w - f16(w);
//   scales  [64 lanes]: the E8M0 exponent of the block that lane half supplies (kh = 0: lo_w, 1: hi_w), one per row.
void pack_contour_folded_mx(const Tensor* w1, std::vector<uint16_t>& a16, std::vector<uint8_t>& mx,
                            std::vector<int32_t>& scales) {
  static const int shifts[8] = {-36, 0, 36, 57, 72, 84, 93, 101};
  std::vector<double> keff((size_t)8 * 3 * 176, 0.0);
  for (int o = 0; o < 8; ++o)
    for (int c = 0; c < 8; ++c)
      for (int dt = 0; dt < 3; ++dt)
        for (int df = 0; df < 39; ++df)
          keff[((size_t)o * 3 + dt) * 176 + (df - 19 + shifts[c] + 55)] += (double)w1->data[((o * 8 + c) * 3 + dt) * 39 + df];
  auto tap = [&](int i, int dt, int t) -> float {  // A[i][tap' = t] of frame dt
    const int j = i >> 3, o = i & 7, g = t - j - 1;
    return (g >= 0 && g < 176) ? (float)keff[((size_t)o * 3 + dt) * 176 + g] : 0.0f;
  };
  a16.assign((size_t)36 * 64 * 8, 0);
  for (int dt = 0; dt < 3; ++dt)
    for (int e = 0; e < 12; ++e)
      for (int lane = 0; lane < 64; ++lane)
        for (int el = 0; el < 8; ++el)
          a16[((size_t)(dt * 12 + e) * 64 + lane) * 8 + el] = f32_to_f16(tap(lane & 31, dt, 16 * e + 8 * (lane >> 5) + el));
  mx.assign((size_t)18 * 64 * 32, 0);
  // ONE E8M0 scale per accumulator row and K block for all 18 steps (a register instead of 18 in the kernel): e4m3 is a
  // floating format with 15 binades of normals, taps 2^-15 below the row maximum are noise at the corrections' scale
  auto block_exp = [](double m) {
    const int e2 = m > 0 ? (int)std::ceil(std::log2(m / 448.0)) : -126;
    return e2 < -126 ? -126 : e2;
  };
  auto split = [&](int i, int dt, int t, double& h, double& l) {
    const double v = (double)tap(i, dt, t);
    h = (double)f16_to_f32(f32_to_f16((float)v)), l = v - h;
  };
  std::vector<int32_t> sc(64, 127);
  for (int i = 0; i < 32; ++i) {
    double mlo = 0, mhi = 0, h, l;
    for (int dt = 0; dt < 3; ++dt)
      for (int t = 0; t < 192; ++t) split(i, dt, t, h, l), mlo = std::fmax(mlo, std::fabs(l)), mhi = std::fmax(mhi, std::fabs(h));
    const int elo = block_exp(mlo), ehi = block_exp(mhi);
    sc[i] = 127 + elo;       // lane (i, kh = 0): K block 0 = lo_w
    sc[32 + i] = 127 + ehi;  // lane (i, kh = 1): K block 1 = hi_w
    for (int S = 0; S < 18; ++S)
      for (int kh = 0; kh < 2; ++kh)
        for (int el = 0; el < 16; ++el) {
          split(i, S / 6, 32 * (S % 6) + 16 * kh + el, h, l);
          uint8_t* dst = &mx[((size_t)S * 64 + 32 * kh + i) * 32];
          dst[el] = f32_to_e4m3(std::ldexp(l, -elo));
          dst[16 + el] = f32_to_e4m3(std::ldexp(h, -ehi));
        }
  }
  scales = sc;
}

// contour conv1 Toeplitz B fragments [4 waves][126][64] (conv_contour1.hip).
void pack_contour1(const Tensor* w, std::vector<float>& out) {
  static const int chan[4][2] = {{0, 1}, {2, 4}, {5, 3}, {6, 7}};
  out.assign(4 * 126 * 64, 0.f);
  for (int wave = 0; wave < 4; ++wave)
    for (int slot = 0; slot < 2; ++slot)
      for (int dt = 0; dt < 3; ++dt)
        for (int ep = 0; ep < 21; ++ep)
          for (int lane = 0; lane < 64; ++lane) {
            const int kodd = lane >> 5, n = lane & 31, o = n >> 2, jj = n & 3;
            const int c = chan[wave][slot];
            const int df = 2 * ep + kodd - jj;
            float v = 0.f;
            if (df >= 0 && df < 39) v = w->data[((o * 8 + c) * 3 + dt) * 39 + df];
            out[((size_t)wave * 126 + slot * 63 + dt * 21 + ep) * 64 + lane] = v;
          }
}

void pack_onset1(const Tensor* w, std::vector<float>& out) {
  out.assign(100 * 64, 0.f);
  for (int cp = 0; cp < 4; ++cp)
    for (int dt = 0; dt < 5; ++dt)
      for (int dw = 0; dw < 5; ++dw)
        for (int lane = 0; lane < 64; ++lane) {
          const int c = 2 * cp + (lane >> 5), o = lane & 31;
          out[((size_t)(cp * 5 + dt) * 5 + dw) * 64 + lane] = w->data[((o * 8 + c) * 5 + dt) * 5 + dw];
        }
}

void pack_note1(const Tensor* w, std::vector<float>& out) {
  out.assign(25 * 64, 0.f);
  for (int s = 0; s < 25; ++s)
    for (int lane = 0; lane < 64; ++lane) {
      const int k = 2 * s + (lane >> 5), o = lane & 31;
      out[(size_t)s * 64 + lane] = (k < 49) ? w->data[o * 49 + k] : 0.f;
    }
}



// Fused branch A fragments (conv_branch.hip): [A1 hi: KS1*64][A1 lo: KS1*64][A2 hi: 2*64][A2 lo: 2*64] x 8 f16.
// A1 lane (i = out channel = lane & 31, h = lane >> 5), element e: conv1 weight of k = 8h + e of step s.
// A2 lane (i = projection row, h), element e of step s2: conv2 weight of the channel that C-register
// 8*s2 + e of half h holds: (e & 3) + 16*s2 + 8*(e >> 2) + 4h.
// Onset conv1 correction products on the block-scaled fp8 instruction (conv_branch.hip, MX variant):
//   mx [7 steps][64 lanes][32 B] then [64] E8M0 scales.  Lane (i = out channel, kh) of step S: bytes 0..15 = tap
//   4 S + kh of the 5x5 window, bytes 16..31 = tap 4 S + 2 + kh (taps >= 25: zero); per tap [fp8(lo_w) x 8 channels |
//   fp8(hi_w) x 8 channels], meeting the image slot's [fp8(a) | fp8(lo_a)].  ONE scale per out channel for both kinds:
//   lo_w is stored 2^11 larger than hi_w (|lo_w| <= 2^-11 |w|), lo_a arrives 2^11 larger than a, so both products carry
//   2^(e - 17) and a 32-tap K block may mix them.
void pack_onset_mx(const Tensor* w1, std::vector<uint8_t>& mx, std::vector<int32_t>& scales) {
  mx.assign((size_t)7 * 64 * 32, 0);
  scales.assign(64, 127);
  for (int i = 0; i < 32; ++i) {
    double mhi = 0;
    for (int k = 0; k < 8 * 25; ++k) mhi = std::fmax(mhi, std::fabs((double)f16_to_f32(f32_to_f16(w1->data[i * 200 + k]))));
    int e = mhi > 0 ? (int)std::ceil(std::log2(mhi / 448.0)) : -100;
    e = e < -100 ? -100 : e;
    // products: (A0 2^(e-11)) (a8 2^-6) and (A1 2^e) (lo8 2^-6 2^-11): scale_a = 2^(e-11), scale_b = 2^-6 (kMxSA)
    scales[i] = scales[32 + i] = 127 + e - 11;
    for (int S = 0; S < 7; ++S)
      for (int kh = 0; kh < 2; ++kh)
        for (int part = 0; part < 2; ++part) {
          const int q = 4 * S + 2 * part + kh;
          uint8_t* dst = &mx[((size_t)S * 64 + 32 * kh + i) * 32 + 16 * part];
          for (int c = 0; c < 8; ++c) {
            const double v = q < 25 ? (double)w1->data[((i * 8 + c) * 5 + q / 5) * 5 + q % 5] : 0.0;
            const double hi = (double)f16_to_f32(f32_to_f16((float)v));
            dst[c] = f32_to_e4m3(std::ldexp(v - hi, -(e - 11)));
            dst[8 + c] = f32_to_e4m3(std::ldexp(hi, -e));
          }
        }
  }
}

void pack_branch(int ks1, const Tensor* w1, const Tensor* w2, bool onset, std::vector<uint16_t>& out) {
  const size_t a1h = 0, a1l = (size_t)ks1 * 64 * 8, a2h = 2 * a1l, a2l = a2h + 2 * 64 * 8;
  out.assign(a2l + 2 * 64 * 8, 0);
  const int kh2 = onset ? 3 : 7;
  for (int s = 0; s < ks1; ++s)
    for (int lane = 0; lane < 64; ++lane)
      for (int e = 0; e < 8; ++e) {
        const int i = lane & 31, hh = lane >> 5;
        float v = 0.f;
        if (onset) {  // k-step = tap pair of the 5x5 window x 8 stack channels (models.py:295-304)
          const int q = 2 * s + hh;
          if (q < 25) v = w1->data[((i * 8 + e) * 5 + q / 5) * 5 + q % 5];
        } else {  // k-step = frame-tap pair x 8 adjacent bins, 7 used (models.py:270-278)
          const int dt = 2 * s + hh;
          if (dt < 7 && e < 7) v = w1->data[(i * 7 + dt) * 7 + e];
        }
        put_split(out, a1h, a1l, ((size_t)s * 64 + lane) * 8 + e, v, 2048.0f);
      }
  // A2 row rho is C row rho of the projection: register r = (rho & 3) + 4 (rho >> 3) of lane half (rho >> 2) & 1.
  // Half 0 takes frame taps 0 .. DT0-1, half 1 the rest, three dw taps in consecutive registers: the kernel's
  // horizontal sum is then two lane shifts (conv_branch.hip, NoteBr::DT0 / OnsetBr::DT0).
  const int dt0 = onset ? 2 : 4;
  for (int s2 = 0; s2 < 2; ++s2)
    for (int lane = 0; lane < 64; ++lane)
      for (int e = 0; e < 8; ++e) {
        const int rho = lane & 31, hh = lane >> 5;
        const int ch = (e & 3) + 16 * s2 + 8 * (e >> 2) + 4 * hh;
        const int r = (rho & 3) + 4 * (rho >> 3), half = (rho >> 2) & 1;
        const int dt = dt0 * half + r / 3, dw = r % 3;
        float v = 0.f;
        if (r < 3 * dt0 && dt < kh2) {
          v = onset ? w2->data[((1 + ch) * 3 + dt) * 3 + dw]   // channel 0 of the concat is the note map
                    : w2->data[(ch * 7 + dt) * 3 + dw];
        }
        put_split(out, a2h, a2l, ((size_t)s2 * 64 + lane) * 8 + e, v, 2048.0f);
      }
}

// onset_march16.hip: conv1 (8 -> 32, 5 x 5, models.py:295-304) and the 3 x 3 head's feature channels (305-318) as
// v_mfma_f32_16x16x32_f16 A fragments: [A1 hi: (2 s + mb) x 64 lanes][A1 lo: 14 + ...][A2 hi][A2 lo] x 8 f16.
// A1: lane (m = lane & 15, g = lane >> 4), element e: out channel 16 mb + m, stack channel e, tap onset16_{dt,dw}(s, g).
// A2: row rho = lane & 15 = 4 dt + dw (dt, dw < 3), K index 8 g + e <-> conv1 channel 4 g + e (e < 4) or 16 + 4 g + e - 4:
// the order in which conv1's C layout leaves a pixel's channels in a lane.
void pack_onset16(const Tensor* w1, const Tensor* w2, std::vector<uint16_t>& out) {
  const size_t frag = 64 * 8, a1h = 0, a1l = 2 * kOnset16KSteps * frag, a2h = 2 * a1l, a2l = a2h + frag;
  out.assign(a2l + frag, 0);
  for (int s = 0; s < kOnset16KSteps; ++s)
    for (int mb = 0; mb < 2; ++mb)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const int m = lane & 15, g = lane >> 4;
          const int q = onset16_dt(s, g) * 5 + onset16_dw(s, g);
          const float v = onset16_live(s, g) ? w1->data[((16 * mb + m) * 8 + e) * 25 + q] : 0.f;
          put_split(out, a1h, a1l, ((size_t)(2 * s + mb) * 64 + lane) * 8 + e, v, 2048.0f);
        }
  for (int lane = 0; lane < 64; ++lane)
    for (int e = 0; e < 8; ++e) {
      const int rho = lane & 15, g = lane >> 4;
      const int dt = rho >> 2, dw = rho & 3;
      const int ch = e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4);
      const float v = (dt < 3 && dw < 3) ? w2->data[((1 + ch) * 3 + dt) * 3 + dw] : 0.f;  // channel 0 of the concat is the note map
      put_split(out, a2h, a2l, (size_t)lane * 8 + e, v, 2048.0f);
    }
}

// conv_contour2.hip contour_conv2_proj_kernel: Conv2D 8 -> 1, 5 x 5 (models.py:254-263; w2 is OIHW (1, 8, 5, 5)) as the A operand
// of v_mfma_f32_32x32x16_f16 with the three split-precision products packed along K: two fragments x 64 lanes x (8 x f16),
// A1 = [hi 2^11 | hi], A2 = [lo 2^11 | 0] against the B operand [hi(c1) | lo(c1) 2^11] of four channels.  Lane (m = lane & 31,
// hk = lane >> 5), elements j < 4 / j >= 4: channel 4 hk + (j & 3); row m <-> C register r = (m & 3) + 4 (m >> 3) of lane half
// (m >> 2) & 1; half 0 holds frame taps dt = 0, 1, 2 (r = 5 dt + df, r = 15 unused), half 1 dt = 3, 4 (r = 5 (dt - 3) + df,
// r >= 10 unused).
bool pack_conv2_proj(const Tensor* w2, std::vector<uint16_t>& out) {
  const size_t frag = 64 * 8;
  out.assign(2 * frag, 0);
  bool ok = true;
  for (int lane = 0; lane < 64; ++lane)
    for (int j = 0; j < 4; ++j) {
      const int m = lane & 31, hk = lane >> 5;
      const int r = (m & 3) + 4 * (m >> 3), half = (m >> 2) & 1;
      const int dt = (half ? 3 : 0) + r / 5, df = r % 5, ch = 4 * hk + j;
      const bool used = half ? r < 10 : r < 15;
      const float v = used ? w2->data[(ch * 5 + dt) * 5 + df] : 0.f;
      const uint16_t hi = f32_to_f16(v);
      const float hif = f16_to_f32(hi);
      if (!(std::fabs(hif) * 2048.0f < 65504.0f)) ok = false;
      const size_t idx = (size_t)lane * 8 + j;
      out[idx] = f32_to_f16(hif * 2048.0f);                 // x hi(c1)
      out[idx + 4] = hi;                                    // x lo(c1) 2^11
      out[frag + idx] = f32_to_f16((v - hif) * 2048.0f);    // x hi(c1); elements 4..7 stay zero
    }
  return ok;
}

// note_march16.hip: conv1 (1 -> 32, 7 x 7, stride (1, 3), models.py:270-278) and the (7, 3) head (282-289) as
// v_mfma_f32_16x16x32_f16 A fragments, 18 x 64 lanes x (8 x f16): conv1 [kind][k-step s][block mb] at (4 kind + 2 s + mb),
// conv2 [kind][block mb] at 12 + 2 kind + mb; kind 0 = hi 2^11, 1 = hi, 2 = lo 2^11 (the kernel adds all three products
// into one accumulator at scale 2^11).  conv1: lane (m = lane & 15, g = lane >> 4), element e: out channel 16 mb + m, frame
// tap dt = 4 s + g (7: zero), bin offset e (7: zero).  conv2: row rho = lane & 15 = 4 dw + i <-> tap (dt = 4 mb + i, dw)
// (dw = 3, dt = 7: zero rows); K index 8 g + e <-> conv1 channel 4 g + e (e < 4) or 16 + 4 g + e - 4 — the order in which
// conv1's C layout leaves a pixel's channels in a lane.  Returns false if a weight's hi part does not survive the 2^11.
bool pack_note16(const Tensor* w1, const Tensor* w2, std::vector<uint16_t>& out) {
  const size_t frag = 64 * 8;
  out.assign(18 * frag, 0);
  bool ok = true;
  auto put3 = [&](size_t f_hi_scaled, size_t f_hi, size_t f_lo, size_t idx, float v) {
    const uint16_t hi = f32_to_f16(v);
    const float hif = f16_to_f32(hi);
    if (!(std::fabs(hif) * 2048.0f < 65504.0f)) ok = false;
    out[f_hi_scaled * frag + idx] = f32_to_f16(hif * 2048.0f);
    out[f_hi * frag + idx] = hi;
    out[f_lo * frag + idx] = f32_to_f16((v - hif) * 2048.0f);
  };
  for (int s = 0; s < 2; ++s)
    for (int mb = 0; mb < 2; ++mb)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const int m = lane & 15, g = lane >> 4, dt = 4 * s + g;
          const float v = (dt < 7 && e < 7) ? w1->data[((16 * mb + m) * 7 + dt) * 7 + e] : 0.f;
          put3(0 + 2 * s + mb, 4 + 2 * s + mb, 8 + 2 * s + mb, (size_t)lane * 8 + e, v);
        }
  for (int mb = 0; mb < 2; ++mb)
    for (int lane = 0; lane < 64; ++lane)
      for (int e = 0; e < 8; ++e) {
        const int rho = lane & 15, g = lane >> 4;
        const int dw = rho >> 2, dt = 4 * mb + (rho & 3);
        const int ch = e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4);
        const float v = (dw < 3 && dt < 7) ? w2->data[(ch * 7 + dt) * 3 + dw] : 0.f;
        put3(12 + mb, 14 + mb, 16 + mb, (size_t)lane * 8 + e, v);
      }
  return ok;
}

// cqt_planes.hip decimator (transposed: the filter is the A operand): T[u][i] = h[i - 2u - 1] — the input window starts one
// sample before the reference's (an 8-sample aligned element of the padded plane) — as [hi: 9 steps][lo: 9 steps] x 64
// lanes x 8 f16; lane (u = lane & 15, kg = lane >> 4), element e: i = 32 s + 8 kg + e.
void pack_decimator_f16(const Tensor* lowp, std::vector<uint16_t>& out, int shift = 1) {
  const size_t lo_base = (size_t)9 * 64 * 8;
  out.assign(2 * lo_base, 0);
  for (int s = 0; s < 9; ++s)
    for (int lane = 0; lane < 64; ++lane)
      for (int e = 0; e < 8; ++e) {
        const int u = lane & 15, kg = lane >> 4;
        const int j = 32 * s + 8 * kg + e - 2 * u - shift;
        // taps pre-scaled by 2^10, residuals by a further 2^11 (kDmTapScale / kLoScale in cqt_mfma.hip)
        put_split(out, 0, lo_base, ((size_t)s * 64 + lane) * 8 + e,
                  (j >= 0 && j < 256) ? lowp->data[j] * 1024.0f : 0.f, 2048.0f);
      }
}

// cqt_planes.hip filterbank: [29 step-fragments][hi|lo][64 lanes][8] f16.  Column groups of 16: 0 = re of filters 0..15,
// 1 = im 0..15 (7 steps from tap 16), 2 = re 16..31, 3 = im 16..31, 4 = {re 32..35, im 32..35, 8 zero columns} (5 steps
// from tap 48); lane (n = lane & 15, kg), element e: tap = 16 + 32 s + 8 kg + e of k-step s.
void pack_filterbank_planes(const Tensor* re, const Tensor* im, std::vector<uint16_t>& out) {
  out.assign((size_t)29 * 2 * 64 * 8, 0);
  static const int frag0[5] = {0, 7, 14, 19, 24}, step0[5] = {0, 0, 1, 1, 1}, steps[5] = {7, 7, 5, 5, 5};
  for (int g = 0; g < 5; ++g)
    for (int s = 0; s < steps[g]; ++s)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const int n = lane & 15, kg = lane >> 4;
          const int tap = 16 + 32 * (step0[g] + s) + 8 * kg + e;
          float v = 0.f;
          if (g < 4) {
            v = ((g & 1) ? im : re)->data[((g >> 1) * 16 + n) * 256 + tap];
          } else if (n < 8) {
            v = (n < 4 ? re : im)->data[(32 + (n & 3)) * 256 + tap];
          }
          const size_t base = ((size_t)(frag0[g] + s) * 2) * 64 * 8;
          put_split(out, base, base + 64 * 8, (size_t)lane * 8 + e, v * 4096.0f, 2048.0f);
        }
}

int free_all(bp_handle h) {
  float* ptrs[] = {h->d_pl_tfrag, h->d_pl_bfrag, h->d_pl_bin_k, h->planes, h->d_note_wfrag, h->d_note_w16, h->d_note_wf32, h->d_onset_wfrag, h->d_onset_wf32, h->d_onset_wmx, h->d_onset_w16, h->zp, h->c1s, h->d_d1_wlds, h->d_d1_wfold, h->d_d1_wmarch, h->d_d1_wrim, h->d_d1_wrimm, h->d_d1_wfold_mx, h->d_d1_bias, h->d_d2_w, h->d_d2_wproj, h->d_lowpass, h->d_sqrt_len, h->d_fb_bfrag, h->d_c1_bfrag, h->d_c1_bias, h->d_o1_bfrag,
                   h->d_o1_bias, h->d_n1_bfrag, h->d_n1_bias, h->d_w_contour2, h->d_w_note2, h->d_w_onset2,
                   h->audio, h->pyr, h->lp, h->c1, h->contour, h->n1, h->note, h->o1, h->onset, h->track,
                   h->track_out, h->nd_buf, h->nd_tables, h->fb_scratch, h->pcm_dev, h->mono_dev, h->res_dev, reinterpret_cast<float*>(h->taps_dev)};
  for (float* p : ptrs)
    if (p) (void)hipFree(p);
  if (h->mm) (void)hipFree(h->mm);
  if (h->ev_valid)
    for (auto& row : h->ev)
      for (auto& e : row) (void)hipEventDestroy(e);
  if (h->done) (void)hipEventDestroy(h->done);
  if (h->nd_stats_host) (void)hipHostFree(h->nd_stats_host);
  if (h->fd_status_host) (void)hipHostFree(h->fd_status_host);
  flac_device_free(h->fd);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  return BP_OK;
}

int ensure_fb_scratch(bp_handle h, int64_t n) {
  if (n <= h->fb_scratch_windows) return BP_OK;
  if (h->fb_scratch) BP_HIP(hipFree(h->fb_scratch));
  h->fb_scratch = nullptr;
  h->fb_scratch_windows = 0;
  BP_HIP(hipMalloc(&h->fb_scratch, filterbank_scratch_floats((int)n) * sizeof(float)));
  h->fb_scratch_windows = n;
  return BP_OK;
}

// The rim of contour conv1: the register-resident march (309-bin CQT), the round-3 GEMM for the extended 44.1 kHz mode
// (its 160 z bins per side would need 120 VGPRs of weights) and, in the A/B library, on BP_RIM=gemm.
static void launch_rim(bp_handle h, const uint32_t* zp, float* c1, int n, bool wlo, hipStream_t s) {
  static const bool gemm = [] {
    const char* e = ab_env("BP_RIM");
    return e && std::strcmp(e, "gemm") == 0;
  }();
  if (h->ext || gemm || !h->d_d1_wrimm)
    launch_contour_conv1_rim(zp, h->d_d1_wrim, h->d_d1_bias, c1, n, h->n_cu, wlo, h->ext, s);
  else
    launch_contour_conv1_rim_march(zp, h->d_d1_wrimm, h->d_d1_bias, c1, n, h->n_cu, wlo, s);
}

// One chunk (n <= cap) of windows already resident at `audio_dev`; outputs to device pointers.
int run_chunk(bp_handle h, const float* audio_dev, int n, float* note_dev, float* onset_dev,
              float* contour_dev) {
  hipStream_t s = h->stream;
  const bool timing = (h->flags & BP_FLAG_STAGE_TIMING) != 0;
  // BP_FLAG_TIME_DOMINANT: events only around the dominant kernel, on every fourth chunk (an event record between two
  // kernels keeps the second from starting under the first's tail: a pair per chunk cost 0.75 against 0.71 ms per step
  // at B = 256, round 5; sampled, the launches that are measured are the same and the rest run undisturbed)
  const bool dom = !timing && (h->flags & BP_FLAG_TIME_DOMINANT) && !(h->flags & BP_FLAG_F32_MFMA) &&
                   (h->dom_chunks++ % bp_context::kDomEvery) == 0;
  const bool wlo = !(h->flags & BP_FLAG_BF16_WEIGHTS);  // conv weights carry an f16 lo part
  int e = dom ? -1 : 0;  // index of the last event recorded
  const int ring_slot = (int)(h->timed_chunks % bp_context::kTimedRing);
  hipEvent_t* ev = h->ev[ring_slot];
  int* seq = h->seq[ring_slot];
  if (timing) BP_HIP(hipEventRecord(ev[0], s));
  // dominant-kernel timing: one (begin, end) pair per launch of the kernel; the interval between two pairs is no stage
#define BP_DOM_BEGIN()                                \
  do {                                                \
    if (dom) {                                        \
      if (e >= 0) seq[e] = -1;                        \
      BP_HIP(hipEventRecord(ev[e + 1], s));           \
      ++e;                                            \
    }                                                 \
  } while (0)
#define BP_DOM_END(id)                                \
  do {                                                \
    if (dom) {                                        \
      seq[e] = (id);                                  \
      BP_HIP(hipEventRecord(ev[e + 1], s));           \
      ++e;                                            \
    }                                                 \
  } while (0)
  // closes the interval of stage `id` (the kernels launched since the previous mark)
#define BP_MARK(id)                                \
  do {                                             \
    if (timing) {                                  \
      seq[e] = (id);                               \
      BP_HIP(hipEventRecord(ev[++e], s));          \
    }                                              \
  } while (0)
  bool zp_done = false;
  if (h->flags & BP_FLAG_F32_MFMA) {
    launch_pyramid(audio_dev, h->pyr, h->d_lowpass, n, s);
    BP_MARK(BP_STAGE_PYRAMID);
    launch_filterbank(audio_dev, h->pyr, h->d_fb_bfrag, h->d_sqrt_len, h->lp, h->mm, h->fb_scratch, n, h->kc,
                      h->n_cu, s);
    BP_MARK(BP_STAGE_FILTERBANK);
  } else {
    uint16_t* pl = reinterpret_cast<uint16_t*>(h->planes);
    launch_pyramid_planes(audio_dev, h->win_len, pl, h->d_pl_tfrag, n, h->n_cu, h->ext, s);
    BP_MARK(BP_STAGE_PYRAMID);
    // with at least half a window per CU the kernel also normalises / BatchNorms / splits its windows (`zp` complete)
    zp_done = launch_filterbank_planes(pl, audio_dev, h->win_len, h->d_pl_bfrag, h->d_pl_bin_k, h->lp, h->fb_scratch,
                                       reinterpret_cast<uint32_t*>(h->zp), n, h->kc, h->n_cu, h->ext, s);
    BP_MARK(BP_STAGE_FILTERBANK);
  }
  if (h->flags & BP_FLAG_F32_MFMA) {
    launch_contour1(h->lp, h->mm, h->d_c1_bfrag, h->d_c1_bias, h->c1, n, h->kc, h->n_cu, s);
    BP_MARK(BP_STAGE_CONTOUR1);
    launch_contour2(h->c1, h->d_w_contour2, h->b_contour2, contour_dev, n, s);
    BP_MARK(BP_STAGE_CONTOUR2);
    launch_note1(contour_dev, h->d_n1_bfrag, h->d_n1_bias, h->n1, n, h->n_cu, s);
    BP_MARK(BP_STAGE_NOTE1);
    launch_note2(h->n1, h->d_w_note2, h->b_note2, note_dev, n, s);
    BP_MARK(BP_STAGE_NOTE2);
    launch_onset1(h->lp, h->mm, h->d_o1_bfrag, h->d_o1_bias, h->o1, n, h->kc, h->n_cu, s);
    BP_MARK(BP_STAGE_ONSET1);
    launch_onset2(note_dev, h->o1, h->d_w_onset2, h->b_onset2, onset_dev, n, s);
    BP_MARK(BP_STAGE_ONSET2);
  } else {
    if (!zp_done) {
      // fewer windows than CUs: the filterbank left per-tile extrema, folded here by every workgroup for itself
      launch_zpack_partials(h->lp, h->fb_scratch, filterbank_planes_partials(h->ext), reinterpret_cast<uint32_t*>(h->zp), n,
                            h->kc, h->n_bins, s);
      BP_MARK(BP_STAGE_ZPACK);
    }
    // The contour branch can run in parts of a chunk (BP_CONTOUR_PARTS=2: rim, folded conv1, conv2 of windows 0..127, then
    // of 128..255), so that conv1's 1.48 MB of c1 per window stay inside the 256 MB Infinity Cache until conv2 reads them.
    // Measured in round 3 at B = 256: conv2 0.115 -> 0.107 ms, but folded conv1 0.228 -> 0.239 and rim 0.072 -> 0.081 (each
    // part pays the kernels' prologue again): 0.852 vs 0.844 ms per step.  Kept as a tool; one part is the default.
    const int parts = h->contour_parts > 0 ? h->contour_parts : 1;
    for (int part = 0; part < parts; ++part) {
      const int w0 = (int)((int64_t)n * part / parts), nw = (int)((int64_t)n * (part + 1) / parts) - w0;
      if (nw <= 0) continue;
      const uint32_t* zpp = reinterpret_cast<const uint32_t*>(h->zp) + (int64_t)w0 * kZWin;
      float* c1p = h->c1s + (int64_t)w0 * kC1Win;
      if (contour_conv1_full()) BP_DOM_BEGIN();
#ifdef BP_AB_KERNELS
      if (contour_conv1_full() || h->rim_exact)
        launch_contour_conv1_exact(zpp, h->d_d1_wlds, h->d_d1_bias, c1p, nw, h->n_cu, wlo, s);
      else
#endif
        launch_rim(h, zpp, c1p, nw, wlo, s);
      if (!contour_conv1_full()) {
        BP_MARK(BP_STAGE_CONTOUR_CONV1_EDGE);
        BP_DOM_BEGIN();
#ifdef BP_AB_KERNELS
        if (h->fold_mx && wlo) {
          const char* base = reinterpret_cast<const char*>(h->d_d1_wfold_mx);
          launch_contour_conv1_fold_mx(zpp, base, base + 36 * 64 * 16, base + 36 * 64 * 16 + 18 * 64 * 32, h->d_d1_bias, c1p,
                                       nw, h->n_cu, s);
        } else if (!contour_conv1_use_march()) {
          launch_contour_conv1_folded(zpp, h->d_d1_wfold, h->d_d1_bias, c1p, nw, h->n_cu, wlo, s);
        } else
#endif
          launch_contour_conv1_march(zpp, h->d_d1_wmarch, h->d_d1_bias, c1p, nw, h->n_cu, wlo, s);
      }
      BP_DOM_END(BP_STAGE_CONTOUR_CONV1);
      BP_MARK(BP_STAGE_CONTOUR_CONV1);
      launch_conv2(c1p, h->d_d2_w, h->d_d2_wproj, h->b_contour2, contour_dev + (int64_t)w0 * kPlaneC, nw, h->n_cu, wlo, s);
      BP_MARK(BP_STAGE_CONTOUR_CONV2);
    }
    launch_note(contour_dev, h->d_note_wfrag, h->d_note_w16, h->d_note_wf32, note_dev, n, h->n_cu, wlo, s);
    BP_MARK(BP_STAGE_NOTE);
    launch_onset(reinterpret_cast<const uint32_t*>(h->zp), note_dev, h->d_onset_wfrag, h->d_onset_wf32, h->d_onset_wmx, h->d_onset_w16, onset_dev,
                 n, h->n_cu, wlo, s);
    BP_MARK(BP_STAGE_ONSET);
  }
#undef BP_MARK
#undef BP_DOM_BEGIN
#undef BP_DOM_END
  if (timing || dom) {
    h->n_seq[ring_slot] = e < 0 ? 0 : e;
    h->timed_chunks++;
  }
  BP_HIP(hipGetLastError());
  return BP_OK;
}

}  // namespace

extern "C" {

const char* bp_version(void) { return "basic_pitch_amd 0.1.0 (gfx950)"; }

// Number of HIP devices this process sees (0 without a GPU or a HIP runtime).
int bp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

const char* bp_last_error(bp_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int bp_create(const void* weights, size_t nbytes, int device_ordinal, unsigned flags,
              int64_t max_windows_hint, bp_handle* out) {
  if (!out) {
    g_create_error = "bp_create: out is NULL";
    return BP_ERR_INVALID_ARG;
  }
  *out = nullptr;
  // windows per chunk: the per-window kernels put the window index in gridDim.y (<= 65535) and several launch helpers
  // count items in 32-bit; 16384 windows (3 GB of workspace) is far beyond where a larger chunk still helps
  if (max_windows_hint > BP_MAX_WINDOWS_PER_CHUNK) {
    g_create_error = "bp_create: max_windows_hint exceeds BP_MAX_WINDOWS_PER_CHUNK (16384); larger batches are chunked "
                     "inside bp_infer, pass 0 for the default of 256";
    return BP_ERR_INVALID_ARG;
  }
  Blob blob;
  std::string err;
  if (!parse_blob(weights, nbytes, blob, err)) {
    g_create_error = err;
    return BP_ERR_BAD_WEIGHTS;
  }
  const Tensor *re, *im, *lowp, *sq, *eps, *lsc, *bn, *c1w, *c1b, *c2w, *c2b, *n1w, *n1b, *n2w, *n2b, *o1w,
      *o1b, *o2w, *o2b;
  if (!expect(blob, "cqt_kernel_re", {36, 256}, re, err) || !expect(blob, "cqt_kernel_im", {36, 256}, im, err) ||
      !expect(blob, "cqt_lowpass", {256}, lowp, err) || !expect(blob, "cqt_sqrt_len", {309}, sq, err) ||
      !expect(blob, "log_eps", {1}, eps, err) || !expect(blob, "log_scale", {2}, lsc, err) ||
      !expect(blob, "bn_affine", {2}, bn, err) || !expect(blob, "contour1_w", {8, 8, 3, 39}, c1w, err) ||
      !expect(blob, "contour1_b", {8}, c1b, err) || !expect(blob, "contour2_w", {1, 8, 5, 5}, c2w, err) ||
      !expect(blob, "contour2_b", {1}, c2b, err) || !expect(blob, "note1_w", {32, 1, 7, 7}, n1w, err) ||
      !expect(blob, "note1_b", {32}, n1b, err) || !expect(blob, "note2_w", {1, 32, 7, 3}, n2w, err) ||
      !expect(blob, "note2_b", {1}, n2b, err) || !expect(blob, "onset1_w", {32, 8, 5, 5}, o1w, err) ||
      !expect(blob, "onset1_b", {32}, o1b, err) || !expect(blob, "onset2_w", {1, 33, 3, 3}, o2w, err) ||
      !expect(blob, "onset2_b", {1}, o2b, err)) {
    g_create_error = err;
    return BP_ERR_BAD_WEIGHTS;
  }
  // BP_FLAG_BF16_WEIGHTS: the six Conv2D weight tensors rounded to bf16 (round to nearest even); everything
  // downstream (packing, the exact same kernels) sees ordinary fp32 numbers with 8 significant bits
  std::vector<std::vector<float>> rounded;
  std::vector<Tensor> rounded_t;
  rounded.reserve(6);
  rounded_t.reserve(6);
  if ((flags & BP_FLAG_BF16_WEIGHTS) && !(flags & BP_FLAG_F32_MFMA)) {
    auto to_bf16 = [&](const Tensor*& t) {
      rounded.emplace_back(t->data, t->data + t->count);
      for (float& v : rounded.back()) {
        uint32_t u;
        std::memcpy(&u, &v, 4);
        u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
        std::memcpy(&v, &u, 4);
      }
      Tensor c = *t;
      c.data = rounded.back().data();
      rounded_t.push_back(c);
      t = &rounded_t.back();
    };
    to_bf16(c1w), to_bf16(c2w), to_bf16(n1w), to_bf16(n2w), to_bf16(o1w), to_bf16(o2w);
  }
  std::vector<float> fb;
  if (!pack_filterbank(re, im, fb, err)) {
    g_create_error = err;
    return BP_ERR_UNSUPPORTED;
  }

  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
    g_create_error = "bp_create: no HIP device visible (this library has no CPU path)";
    return BP_ERR_NO_DEVICE;
  }
  if (device_ordinal < 0 || device_ordinal >= n_dev) {
    g_create_error = "bp_create: device_ordinal out of range";
    return BP_ERR_INVALID_ARG;
  }
  hipDeviceProp_t prop;
  if (hipSetDevice(device_ordinal) != hipSuccess || hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) {
    g_create_error = "bp_create: hipSetDevice / hipGetDeviceProperties failed";
    return BP_ERR_NO_DEVICE;
  }
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    g_create_error = std::string("bp_create: device is ") + prop.gcnArchName +
                     ", this library only carries gfx950 (MI355X) code objects";
    return BP_ERR_NO_DEVICE;
  }

  bp_handle h = new bp_context();
  h->device = device_ordinal;
  h->flags = (flags & BP_FLAG_F32_MFMA) ? (flags & ~BP_FLAG_BF16_WEIGHTS) : flags;
  h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  std::snprintf(h->arch, sizeof h->arch, "%s", prop.gcnArchName);
  h->cap = max_windows_hint > 0 ? max_windows_hint : 256;
  if (flags & BP_FLAG_EXT_CQT_44K) {
    if (flags & BP_FLAG_F32_MFMA) {
      g_create_error = "bp_create: BP_FLAG_EXT_CQT_44K is not available on the exact-f32 path (BP_FLAG_F32_MFMA)";
      delete h;
      return BP_ERR_UNSUPPORTED;
    }
    h->ext = true;
    h->win_len = kAudioNExt;
    h->hop = 2 * BP_HOP_SIZE;
    h->lead = BP_OVERLAP_LEN;
    h->n_bins = kBinsExt;
    h->rate = 2 * BP_AUDIO_SAMPLE_RATE;
    h->pyr_stride = kPyrStrideExt;
  }
  h->kc.eps = eps->data[0];
  h->kc.s0 = lsc->data[0];
  h->kc.s1 = lsc->data[1];
  h->kc.bn_a = bn->data[0];
  h->kc.bn_b = bn->data[1];
  h->b_contour2 = c2b->data[0];
  h->b_note2 = n2b->data[0];
  h->b_onset2 = o2b->data[0];

  auto fail = [&](int code) {
    g_create_error = h->err;
    free_all(h);
    delete h;
    return code;
  };
  auto vec = [](const Tensor* t) { return std::vector<float>(t->data, t->data + t->count); };
  int rc;
  {
    hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      h->err = std::string("hipStreamCreate failed: ") + hipGetErrorString(e);
      return fail(BP_ERR_HIP);
    }
    h->stream = h->own_stream;
  }
  std::vector<float> c1f, o1f, n1f;
  {  // split-precision path: f16 hi | scaled-lo operand fragments (raw bytes) + fp32 side tables
    auto raw_of = [](const std::vector<uint16_t>& f) {
      std::vector<float> raw(f.size() / 2);
      std::memcpy(raw.data(), f.data(), f.size() * 2);
      return raw;
    };
    std::vector<uint16_t> frag;
    pack_decimator_f16(lowp, frag);
    if ((rc = upload(h, raw_of(frag), &h->d_pl_tfrag))) return fail(rc);
    pack_filterbank_planes(re, im, frag);
    if ((rc = upload(h, raw_of(frag), &h->d_pl_bfrag))) return fail(rc);
    std::vector<float> w2t(200);
    for (int dt = 0; dt < 5; ++dt)
      for (int dw = 0; dw < 5; ++dw)
        for (int c = 0; c < 8; ++c) w2t[(dt * 5 + dw) * 8 + c] = c2w->data[(c * 5 + dt) * 5 + dw];
    if ((rc = upload(h, vec(c1b), &h->d_d1_bias)) || (rc = upload(h, w2t, &h->d_d2_w))) return fail(rc);
    if (!pack_conv2_proj(c2w, frag)) {
      h->err = "bp_create: a contour conv2 weight is too large for the scaled f16 operand (|w| >= 31.98)";
      return fail(BP_ERR_BAD_WEIGHTS);
    }
    if ((rc = upload(h, raw_of(frag), &h->d_d2_wproj))) return fail(rc);
#ifdef BP_AB_KERNELS  // operand tables of the A/B conv1 kernels (conv_contour_direct.hip)
    pack_contour_direct(c1w, frag);
    if ((rc = upload(h, raw_of(frag), &h->d_d1_wlds))) return fail(rc);
    pack_contour_folded(c1w, frag);
    if ((rc = upload(h, raw_of(frag), &h->d_d1_wfold))) return fail(rc);
#endif
    pack_contour_march(c1w, frag);
    if ((rc = upload(h, raw_of(frag), &h->d_d1_wmarch))) return fail(rc);
    if (flags & BP_FLAG_EXT_CQT_44K)
      pack_contour_rim(c1w, frag, kBinsExt, 160);  // the 345-bin CQT: 160 z bins per rim side (conv_contour_rim.hip RimGeo<160>)
    else
      pack_contour_rim(c1w, frag);
    if ((rc = upload(h, raw_of(frag), &h->d_d1_wrim))) return fail(rc);
    if (!(flags & BP_FLAG_EXT_CQT_44K)) {  // the register-resident rim kernel serves the 309-bin CQT
      pack_contour_rim_march(c1w, frag);
      if ((rc = upload(h, raw_of(frag), &h->d_d1_wrimm))) return fail(rc);
    }
#ifdef BP_AB_KERNELS
    // folded conv1: all three split-precision products on f16 by default (fp32-class); BP_FLAG_FP8_CORRECTIONS opts into
    // the block-scaled fp8 corrections (conv_contour_fold_mx.hip; ~1e-5 on the contour map), BP_CONV1=f16 then keeps this
    // one layer on the three-product f16 kernel (A/B runs).
    // the fp8 planes hold z 2^6 with z = bn_a x + bn_b, x in [0, 1] (NormalizedLog): they must stay below e4m3's 448
    const bool fp8_ok = std::fmax(std::fabs(h->kc.bn_b), std::fabs(h->kc.bn_a + h->kc.bn_b)) * 64.0f <= 440.0f &&
                        (flags & BP_FLAG_FP8_CORRECTIONS) && !(flags & BP_FLAG_F16_CORRECTIONS);
    if (const char* ec = ab_env("BP_CONV1");
        !(ec && std::strcmp(ec, "f16") == 0) && fp8_ok && !(flags & BP_FLAG_BF16_WEIGHTS)) {
      std::vector<uint16_t> a16;
      std::vector<uint8_t> mxf;
      std::vector<int32_t> mxs;
      pack_contour_folded_mx(c1w, a16, mxf, mxs);
      std::vector<float> raw(a16.size() / 2 + mxf.size() / 4 + mxs.size());
      std::memcpy(raw.data(), a16.data(), a16.size() * 2);
      std::memcpy(raw.data() + a16.size() / 2, mxf.data(), mxf.size());
      std::memcpy(raw.data() + a16.size() / 2 + mxf.size() / 4, mxs.data(), mxs.size() * 4);
      if ((rc = upload(h, raw, &h->d_d1_wfold_mx))) return fail(rc);
      h->fold_mx = true;
    }
#else
    // the reduced-precision fp8-corrections mode left the product library in round 6 (not faster than the default any more,
    // narrower than the config's fp32): its kernels are built into the A/B library only
    const bool fp8_ok = false;
    if ((flags & BP_FLAG_FP8_CORRECTIONS) && !(flags & BP_FLAG_F16_CORRECTIONS)) {
      h->err = "bp_create: BP_FLAG_FP8_CORRECTIONS is built into the A/B library only (basic_pitch_amd.build.build_library(ab=True), "
               "BASIC_PITCH_AMD_LIB); the product library computes all three split-precision products on f16";
      return fail(BP_ERR_INVALID_ARG);
    }
#endif
    if (const char* ep = ab_env("BP_CONTOUR_PARTS")) h->contour_parts = std::atoi(ep) > 8 ? 8 : std::atoi(ep);
    {
      const char* er = ab_env("BP_RIM");  // "exact": the round-1 rim kernel on the 8-channel form (A/B runs)
      // (the extended 345-bin CQT has its own GEMM table since round 4: 160 z bins per side)
      h->rim_exact = er && std::strcmp(er, "exact") == 0;
    }
    if (const char* es = ab_env("BP_RESAMPLE"))  // A/B runs: the resampler's simpler kernels (bit-identical results)
      h->resample_mode = std::strcmp(es, "plain") == 0 ? 1 : std::strcmp(es, "tiled") == 0 ? 2 : 0;
    for (int br = 0; br < 2; ++br) {
      std::vector<float> f32(42, 0.f);
      const Tensor* b1 = br ? o1b : n1b;
      for (int i = 0; i < 32; ++i) f32[i] = b1->data[i];
      if (br)
        for (int i = 0; i < 9; ++i) f32[32 + i] = o2w->data[i];  // onset2 taps of concat channel 0 (the note map)
      f32[41] = br ? o2b->data[0] : n2b->data[0];
#ifdef BP_AB_KERNELS  // the 32x32x16 kernels' fragments (note_march.hip, onset_march.hip, conv_branch.hip)
      pack_branch(br ? 13 : 4, br ? o1w : n1w, br ? o2w : n2w, br == 1, frag);
      if ((rc = upload(h, raw_of(frag), br ? &h->d_onset_wfrag : &h->d_note_wfrag))) return fail(rc);
#endif
      if ((rc = upload(h, f32, br ? &h->d_onset_wf32 : &h->d_note_wf32))) return fail(rc);
    }
    {  // the note march on 16x16x32 (the default): its own fragment order, the hi parts also at scale 2^11
      if (!pack_note16(n1w, n2w, frag)) {
        h->err = "bp_create: a note-branch weight is too large for the scaled f16 operand (|w| >= 31.98)";
        return fail(BP_ERR_BAD_WEIGHTS);
      }
      if ((rc = upload(h, raw_of(frag), &h->d_note_w16))) return fail(rc);
    }
    {  // the onset march on 16x16x32 (the default): its own fragment order
      pack_onset16(o1w, o2w, frag);
      if ((rc = upload(h, raw_of(frag), &h->d_onset_w16))) return fail(rc);
    }
#ifdef BP_AB_KERNELS
    // onset conv1: fp8 corrections under BP_FLAG_FP8_CORRECTIONS like the folded contour conv1 (BP_ONSET=f16: not this layer)
    if (const char* eo = ab_env("BP_ONSET"); !(eo && std::strcmp(eo, "f16") == 0) && fp8_ok && !(flags & BP_FLAG_BF16_WEIGHTS)) {
      std::vector<uint8_t> mxf;
      std::vector<int32_t> mxs;
      pack_onset_mx(o1w, mxf, mxs);
      std::vector<float> raw(mxf.size() / 4 + mxs.size());
      std::memcpy(raw.data(), mxf.data(), mxf.size());
      std::memcpy(raw.data() + mxf.size() / 4, mxs.data(), mxs.size() * 4);
      if ((rc = upload(h, raw, &h->d_onset_wmx))) return fail(rc);
    }
#else
    (void)fp8_ok;
#endif
  }
  pack_contour1(c1w, c1f);
  pack_onset1(o1w, o1f);
  pack_note1(n1w, n1f);
  std::vector<float> sqrt_len = vec(sq);
  if (h->ext) {
    // lengths = ceil(Q * sr / f_b), f_b = 27.5 * 2^(b / 36), Q = 1 / (2^(1/36) - 1) at sr = 44100 (nnaudio.py:532,
    // 590-593); bin b + 36 of this table must reproduce bin b of the 22.05 kHz artifact
    const double Q = 1.0 / (std::pow(2.0, 1.0 / 36.0) - 1.0);
    std::vector<float> ext(kBinsExt);
    for (int bn = 0; bn < kBinsExt; ++bn)
      ext[bn] = (float)std::sqrt(std::ceil(Q * 44100.0 / (27.5 * std::pow(2.0, bn / 36.0))));
    for (int bn = 0; bn < kBins; ++bn)
      if (std::fabs(ext[bn + 36] - sqrt_len[bn]) > 1e-6f * sqrt_len[bn]) {
        h->err = "bp_create: the extended sqrt(lengths) table does not continue the model's table";
        return fail(BP_ERR_BAD_WEIGHTS);
      }
    sqrt_len = ext;
  }
  {
    std::vector<float> bin_k(sqrt_len.size());
    filterbank_planes_bin_consts(sqrt_len.data(), (int)sqrt_len.size(), h->kc, bin_k.data());
    if ((rc = upload(h, bin_k, &h->d_pl_bin_k))) return fail(rc);
  }
  if ((rc = upload(h, vec(lowp), &h->d_lowpass)) || (rc = upload(h, sqrt_len, &h->d_sqrt_len)) ||
      (rc = upload(h, fb, &h->d_fb_bfrag)) || (rc = upload(h, c1f, &h->d_c1_bfrag)) ||
      (rc = upload(h, vec(c1b), &h->d_c1_bias)) || (rc = upload(h, o1f, &h->d_o1_bfrag)) ||
      (rc = upload(h, vec(o1b), &h->d_o1_bias)) || (rc = upload(h, n1f, &h->d_n1_bfrag)) ||
      (rc = upload(h, vec(n1b), &h->d_n1_bias)) || (rc = upload(h, vec(c2w), &h->d_w_contour2)) ||
      (rc = upload(h, vec(n2w), &h->d_w_note2)) || (rc = upload(h, vec(o2w), &h->d_w_onset2)))
    return fail(rc);

  const int64_t cap = h->cap;
  if ((rc = alloc(h, &h->audio, cap * (int64_t)h->win_len)) || (rc = alloc(h, &h->pyr, cap * h->pyr_stride)) ||
      (rc = alloc(h, &h->lp, cap * kFrames * (int64_t)h->n_bins + 4)) || (rc = alloc(h, &h->c1, cap * 8 * kPlaneC)) ||
      (rc = alloc(h, &h->contour, cap * kPlaneC)) || (rc = alloc(h, &h->n1, cap * 32 * kPlaneN)) ||
      (rc = alloc(h, &h->note, cap * kPlaneN)) || (rc = alloc(h, &h->o1, cap * 32 * kPlaneN)) ||
      (rc = alloc(h, &h->onset, cap * kPlaneN)) || (rc = alloc(h, &h->zp, cap * (int64_t)kZWin)) ||
      (rc = alloc(h, &h->c1s, cap * (int64_t)kC1Win)))
    return fail(rc);
  {
    // pad frames / pad words of zp are zero for good: the fused filterbank writes only the words that carry bins
    hipError_t e = hipMemset(h->zp, 0, (size_t)cap * kZWin * sizeof(float));
    if (e == hipSuccess) e = hipMemset(h->c1s, 0, (size_t)cap * kC1Win * sizeof(float));  // pad bins stay zero
    if (e != hipSuccess) {
      h->err = std::string("hipMemset(c1s) failed: ") + hipGetErrorString(e);
      return fail(BP_ERR_HIP);
    }
  }
  if (!(h->flags & BP_FLAG_F32_MFMA)) {
    // f16 planes of a chunk; zeroed once: the slack behind a level's reflect padding is read (and discarded or masked)
    // but never written, and has to stay finite
    const int64_t pl_floats = (cap * planes_elements_per_window(h->ext) + 1) / 2;
    if ((rc = alloc(h, &h->planes, pl_floats))) return fail(rc);
    hipError_t e = hipMemset(h->planes, 0, (size_t)pl_floats * sizeof(float));
    if (e != hipSuccess) {
      h->err = std::string("hipMemset(planes) failed: ") + hipGetErrorString(e);
      return fail(BP_ERR_HIP);
    }
  }
  {
    hipError_t e = hipMalloc(&h->mm, cap * 2 * sizeof(int));
    if (e != hipSuccess) {
      h->err = std::string("hipMalloc(mm) failed: ") + hipGetErrorString(e);
      return fail(BP_ERR_OUT_OF_MEMORY);
    }
  }
  if ((rc = ensure_fb_scratch(h, cap))) return fail(rc);
  if (flags & (BP_FLAG_STAGE_TIMING | BP_FLAG_TIME_DOMINANT)) {
    for (auto& row : h->ev)
      for (auto& e : row)
        if (hipEventCreate(&e) != hipSuccess) {
          h->err = "hipEventCreate failed";
          return fail(BP_ERR_HIP);
        }
    h->ev_valid = true;
  }
  if ((flags & BP_FLAG_BLOCKING_WAIT) &&
      hipEventCreateWithFlags(&h->done, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) {
    h->err = "hipEventCreateWithFlags failed";
    return fail(BP_ERR_HIP);
  }
  *out = h;
  return BP_OK;
}

void bp_destroy(bp_handle h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  free_all(h);
  delete h;
}

int bp_set_stream(bp_handle h, void* hip_stream) {
  if (!h) return BP_ERR_INVALID_ARG;
  hipStream_t next = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
  if (next == h->stream) return BP_OK;
  // One workspace per handle (pyr, lp, zp, c1s, staging buffers): work still queued on the previous stream must
  // finish before work on the new stream may touch it.  Order the two streams with an event instead of a host sync.
  BP_HIP(hipSetDevice(h->device));
  hipEvent_t ev;
  BP_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ev, h->stream);
  if (e == hipSuccess) e = hipStreamWaitEvent(next, ev, 0);
  (void)hipEventDestroy(ev);  // released once the recorded work completes
  BP_HIP(e);
  h->stream = next;
  return BP_OK;
}

int bp_synchronize(bp_handle h) {
  if (!h) return BP_ERR_INVALID_ARG;
  BP_HIP(hipSetDevice(h->device));
  BP_HIP(hipStreamSynchronize(h->stream));
  return BP_OK;
}

int bp_get_info(bp_handle h, bp_info* out) {
  if (!h || !out) return BP_ERR_INVALID_ARG;
  out->device_ordinal = h->device;
  out->compute_units = h->n_cu;
  out->max_windows = h->cap;
  out->workspace_bytes = h->workspace_bytes;
  std::memset(out->arch, 0, sizeof out->arch);
  std::snprintf(out->arch, sizeof out->arch, "%s", h->arch);
  return BP_OK;
}

int bp_infer_async(bp_handle h, const float* audio_dev, int64_t n_windows, float* note_dev,
                   float* onset_dev, float* contour_dev) {
  if (!h) return BP_ERR_INVALID_ARG;
  if (n_windows < 0 || (n_windows > 0 && (!audio_dev || !note_dev || !onset_dev || !contour_dev))) {
    h->err = "bp_infer: null pointer or negative window count";
    return BP_ERR_INVALID_ARG;
  }
  BP_HIP(hipSetDevice(h->device));
  for (int64_t w0 = 0; w0 < n_windows; w0 += h->cap) {
    const int n = (int)((n_windows - w0) < h->cap ? (n_windows - w0) : h->cap);
    int rc = run_chunk(h, audio_dev + w0 * h->win_len, n, note_dev + w0 * kPlaneN, onset_dev + w0 * kPlaneN,
                       contour_dev + w0 * kPlaneC);
    if (rc) return rc;
  }
  return BP_OK;
}

int bp_infer(bp_handle h, const float* audio, int64_t n_windows, float* note, float* onset,
             float* contour, int mem_kind) {
  if (!h) return BP_ERR_INVALID_ARG;
  if (mem_kind == BP_MEM_DEVICE) {
    int rc = bp_infer_async(h, audio, n_windows, note, onset, contour);
    if (rc) return rc;
    BP_HIP(hipStreamSynchronize(h->stream));
    return BP_OK;
  }
  if (mem_kind != BP_MEM_HOST) {
    h->err = "bp_infer: unknown mem_kind";
    return BP_ERR_INVALID_ARG;
  }
  if (n_windows < 0 || (n_windows > 0 && (!audio || !note || !onset || !contour))) {
    h->err = "bp_infer: null pointer or negative window count";
    return BP_ERR_INVALID_ARG;
  }
  BP_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  for (int64_t w0 = 0; w0 < n_windows; w0 += h->cap) {
    const int n = (int)((n_windows - w0) < h->cap ? (n_windows - w0) : h->cap);
    BP_HIP(hipMemcpyAsync(h->audio, audio + w0 * h->win_len, (size_t)n * h->win_len * 4, hipMemcpyHostToDevice, s));
    int rc = run_chunk(h, h->audio, n, h->note, h->onset, h->contour);
    if (rc) return rc;
    BP_HIP(hipMemcpyAsync(note + w0 * kPlaneN, h->note, (size_t)n * kPlaneN * 4, hipMemcpyDeviceToHost, s));
    BP_HIP(hipMemcpyAsync(onset + w0 * kPlaneN, h->onset, (size_t)n * kPlaneN * 4, hipMemcpyDeviceToHost, s));
    BP_HIP(hipMemcpyAsync(contour + w0 * kPlaneC, h->contour, (size_t)n * kPlaneC * 4, hipMemcpyDeviceToHost, s));
    BP_HIP(hipStreamSynchronize(s));
  }
  return BP_OK;
}

// the same counts for a handle's geometry (hop 36164 / lead-in 3840 at 22.05 kHz, doubled for the extended range)
static int64_t h_track_n_windows(bp_handle h, int64_t n_samples) {
  if (n_samples <= 0) return 0;
  return (n_samples + h->lead + h->hop - 1) / h->hop;
}
static int64_t h_track_n_frames(bp_handle h, int64_t n_samples) {
  if (n_samples <= 0) return 0;
  const double n_expected_windows = (double)n_samples / (double)h->hop;
  const int64_t rows = (int64_t)(n_expected_windows * (double)BP_FRAMES_PER_WINDOW);
  const int64_t avail = h_track_n_windows(h, n_samples) * BP_FRAMES_PER_WINDOW;
  return rows < avail ? rows : avail;
}

int64_t bp_handle_track_n_windows(bp_handle h, int64_t n_samples) { return h ? h_track_n_windows(h, n_samples) : 0; }
int64_t bp_handle_track_n_frames(bp_handle h, int64_t n_samples) { return h ? h_track_n_frames(h, n_samples) : 0; }
int64_t bp_handle_window_samples(bp_handle h) { return h ? h->win_len : 0; }
int bp_handle_sample_rate(bp_handle h) { return h ? h->rate : 0; }
int64_t bp_handle_resampled_length(bp_handle h, int64_t n_frames, int sample_rate) {
  if (!h || n_frames <= 0 || sample_rate <= 0) return 0;
  return (n_frames * (int64_t)h->rate + sample_rate - 1) / sample_rate;
}

int64_t bp_track_n_windows(int64_t n_samples) {
  if (n_samples <= 0) return 0;
  return (n_samples + BP_OVERLAP_LEN / 2 + BP_HOP_SIZE - 1) / BP_HOP_SIZE;
}

int64_t bp_track_n_frames(int64_t n_samples) {
  if (n_samples <= 0) return 0;
  // int(n_samples / hop_size * 142) evaluated like the reference: float64 division, then product
  const double n_expected_windows = (double)n_samples / (double)BP_HOP_SIZE;
  int64_t rows = (int64_t)(n_expected_windows * (double)BP_FRAMES_PER_WINDOW);
  const int64_t avail = bp_track_n_windows(n_samples) * BP_FRAMES_PER_WINDOW;
  return rows < avail ? rows : avail;
}

// windows of a device-resident 22.05 kHz signal -> un-overlapped posteriorgrams (host or device outputs)
// end of a host-blocking call: spin on the stream (lowest latency) or, with BP_FLAG_BLOCKING_WAIT, give the core to another
// worker thread while the device works.  hipEventSynchronize on a hipEventBlockingSync event does not do that here: measured
// on the MI355X box its user time equals its wall time (tools/experiments/host_cpu.py — the runtime polls the signal), so
// the wait is a query every 20..160 us with the thread asleep in between (a call of a few ms ends ~0.1 ms late).
static int wait_stream(bp_handle h) {
  if (h->done) {
    BP_HIP(hipEventRecord(h->done, h->stream));
    for (long ns = 20000;;) {
      const hipError_t e = hipEventQuery(h->done);
      if (e == hipSuccess) break;
      if (e != hipErrorNotReady) BP_HIP(e);
      const timespec ts{0, ns};
      nanosleep(&ts, nullptr);
      if (ns < 80000) ns *= 2;
    }
  } else {
    BP_HIP(hipStreamSynchronize(h->stream));
  }
  return BP_OK;
}

// out_kind kTrackOutInternal: the un-overlapped maps stay in h->track_out ([T][88] note, [T][88] onset, [T][264] contour) and
// the call returns with the work queued on the handle's stream (the caller goes on with device work on them)
constexpr int kTrackOutInternal = 100;
static int track_core(bp_handle h, const float* d_samples, int64_t n_samples, float* note, float* onset,
                      float* contour, int out_kind) {
  hipStream_t s = h->stream;
  const int64_t n_win = h_track_n_windows(h, n_samples);
  const int64_t T = h_track_n_frames(h, n_samples);
  float *d_note = note, *d_onset = onset, *d_contour = contour;
  h->maps_rows = 0;
  if (out_kind == BP_MEM_HOST || out_kind == kTrackOutInternal) {
    const int64_t need = T * (88 + 88 + 264);
    if (need > h->track_out_cap) {
      if (h->track_out) BP_HIP(hipFree(h->track_out));
      h->track_out = nullptr;
      h->track_out_cap = 0;
      BP_HIP(hipMalloc(&h->track_out, (size_t)(need > 0 ? need : 1) * 4));
      h->track_out_cap = need;
    }
    d_note = h->track_out;
    d_onset = d_note + T * 88;
    d_contour = d_onset + T * 88;
  }
  for (int64_t w0 = 0; w0 < n_win; w0 += h->cap) {
    const int n = (int)((n_win - w0) < h->cap ? (n_win - w0) : h->cap);
    launch_window_track(d_samples, n_samples, w0, n, h->audio, h->win_len, h->hop, h->lead, s);
    int rc = run_chunk(h, h->audio, n, h->note, h->onset, h->contour);
    if (rc) return rc;
    if (T > 0) launch_unwrap3(h->note, h->onset, h->contour, w0, n, T, d_note, d_onset, d_contour, s);
  }
  BP_HIP(hipGetLastError());
  if (out_kind == kTrackOutInternal) {
    h->maps_rows = T;
    return BP_OK;
  }
  if (out_kind == BP_MEM_HOST && T > 0) {
    BP_HIP(hipMemcpyAsync(note, d_note, (size_t)T * 88 * 4, hipMemcpyDeviceToHost, s));
    BP_HIP(hipMemcpyAsync(onset, d_onset, (size_t)T * 88 * 4, hipMemcpyDeviceToHost, s));
    BP_HIP(hipMemcpyAsync(contour, d_contour, (size_t)T * 264 * 4, hipMemcpyDeviceToHost, s));
  }
  return wait_stream(h);
}

static int grow(bp_handle h, float** buf, int64_t* cap, int64_t need) {
  if (need <= *cap) return BP_OK;
  // hipFree waits for the whole device, so work of an earlier call that still reads the old buffer has finished
  if (*buf) BP_HIP(hipFree(*buf));
  *buf = nullptr;
  *cap = 0;
  BP_HIP(hipMalloc(buf, (size_t)(need > 0 ? need : 1) * 4));
  *cap = need;
  return BP_OK;
}

int bp_infer_track(bp_handle h, const float* samples, int64_t n_samples, float* note, float* onset,
                   float* contour, int mem_kind) {
  if (!h) return BP_ERR_INVALID_ARG;
  if (n_samples < 0 || (mem_kind != BP_MEM_HOST && mem_kind != BP_MEM_DEVICE)) {
    h->err = "bp_infer_track: bad argument";
    return BP_ERR_INVALID_ARG;
  }
  const int64_t n_win = h_track_n_windows(h, n_samples);
  const int64_t T = h_track_n_frames(h, n_samples);
  if (n_win == 0) return BP_OK;
  if (!samples || (T > 0 && (!note || !onset || !contour))) {
    h->err = "bp_infer_track: null pointer";
    return BP_ERR_INVALID_ARG;
  }
  BP_HIP(hipSetDevice(h->device));
  const float* d_samples = samples;
  if (mem_kind == BP_MEM_HOST) {
    int rc = grow(h, &h->track, &h->track_cap, n_samples);
    if (rc) return rc;
    BP_HIP(hipMemcpyAsync(h->track, samples, (size_t)n_samples * 4, hipMemcpyHostToDevice, h->stream));
    d_samples = h->track;
  }
  return track_core(h, d_samples, n_samples, note, onset, contour, mem_kind);
}

int bp_infer_tracks(bp_handle h, int64_t n_tracks, const float* const* samples, const int64_t* n_samples,
                    float* const* note, float* const* onset, float* const* contour, int mem_kind) {
  if (!h) return BP_ERR_INVALID_ARG;
  if (n_tracks < 0 || (mem_kind != BP_MEM_HOST && mem_kind != BP_MEM_DEVICE) ||
      (n_tracks > 0 && (!samples || !n_samples || !note || !onset || !contour))) {
    h->err = "bp_infer_tracks: bad argument";
    return BP_ERR_INVALID_ARG;
  }
  int64_t total_samples = 0, total_rows = 0;
  for (int64_t t = 0; t < n_tracks; ++t) {
    if (n_samples[t] < 0 || (n_samples[t] > 0 && !samples[t])) {
      h->err = "bp_infer_tracks: negative length or null samples";
      return BP_ERR_INVALID_ARG;
    }
    const int64_t T = h_track_n_frames(h, n_samples[t]);
    if (T > 0 && (!note[t] || !onset[t] || !contour[t])) {
      h->err = "bp_infer_tracks: null output pointer";
      return BP_ERR_INVALID_ARG;
    }
    total_samples += n_samples[t];
    total_rows += T;
  }
  if (total_samples == 0) return BP_OK;
  BP_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  // device views of every track's input and outputs
  std::vector<const float*> d_in(n_tracks);
  std::vector<float*> d_note(n_tracks), d_onset(n_tracks), d_contour(n_tracks);
  if (mem_kind == BP_MEM_HOST) {
    int rc = grow(h, &h->track, &h->track_cap, total_samples);
    if (rc) return rc;
    rc = grow(h, &h->track_out, &h->track_out_cap, total_rows * (88 + 88 + 264));
    if (rc) return rc;
    int64_t so = 0, ro = 0;
    for (int64_t t = 0; t < n_tracks; ++t) {
      const int64_t T = h_track_n_frames(h, n_samples[t]);
      if (n_samples[t] > 0)
        BP_HIP(hipMemcpyAsync(h->track + so, samples[t], (size_t)n_samples[t] * 4, hipMemcpyHostToDevice, s));
      d_in[t] = h->track + so;
      d_note[t] = h->track_out + ro * 440;
      d_onset[t] = d_note[t] + T * 88;
      d_contour[t] = d_onset[t] + T * 88;
      so += n_samples[t];
      ro += T;
    }
  } else {
    for (int64_t t = 0; t < n_tracks; ++t) {
      d_in[t] = samples[t];
      d_note[t] = note[t];
      d_onset[t] = onset[t];
      d_contour[t] = contour[t];
    }
  }
  // the windows of consecutive tracks are packed into full chunks; the pieces of a chunk are windowed by ONE launch and
  // un-overlapped by one launch (bp_common.h TrackSegs; more than kMaxTrackSegs pieces per chunk: several launches)
  std::vector<TrackSeg> segs;
  int cur = 0;
  auto for_groups = [&](auto&& fn) {
    for (size_t g0 = 0; g0 < segs.size(); g0 += kMaxTrackSegs) {
      TrackSegs ts{};
      ts.n = (int)std::min<size_t>(kMaxTrackSegs, segs.size() - g0);
      for (int k = 0; k < ts.n; ++k) ts.seg[k] = segs[g0 + k];
      const int first = ts.seg[0].at, slots = ts.seg[ts.n - 1].at + ts.seg[ts.n - 1].n_windows - first;
      for (int k = 0; k < ts.n; ++k) ts.seg[k].at -= first;  // slots relative to the group's first window
      fn(ts, first, slots);
    }
  };
  auto flush = [&]() -> int {
    if (cur == 0) return BP_OK;
    for_groups([&](const TrackSegs& ts, int first, int slots) {
      launch_window_tracks(ts, slots, h->audio + (int64_t)first * h->win_len, h->win_len, h->hop, h->lead, s);
    });
    int rc = run_chunk(h, h->audio, cur, h->note, h->onset, h->contour);
    if (rc) return rc;
    for_groups([&](const TrackSegs& ts, int first, int slots) {
      launch_unwrap_tracks(ts, slots, h->note + (int64_t)first * kPlaneN, h->onset + (int64_t)first * kPlaneN,
                           h->contour + (int64_t)first * kPlaneC, s);
    });
    segs.clear();
    cur = 0;
    return BP_OK;
  };
  for (int64_t t = 0; t < n_tracks; ++t) {
    const int64_t n_win = h_track_n_windows(h, n_samples[t]);
    const int64_t T = h_track_n_frames(h, n_samples[t]);
    for (int64_t w0 = 0; w0 < n_win;) {
      const int64_t room = h->cap - cur;
      const int n = (int)((n_win - w0) < room ? (n_win - w0) : room);
      segs.push_back(TrackSeg{d_in[t], {d_note[t], d_onset[t], d_contour[t]}, n_samples[t], w0, T, n, cur});
      cur += n;
      w0 += n;
      if (cur == h->cap) {
        int rc = flush();
        if (rc) return rc;
      }
    }
  }
  {
    int rc = flush();
    if (rc) return rc;
  }
  BP_HIP(hipGetLastError());
  if (mem_kind == BP_MEM_HOST) {
    for (int64_t t = 0; t < n_tracks; ++t) {
      const int64_t T = h_track_n_frames(h, n_samples[t]);
      if (T <= 0) continue;
      BP_HIP(hipMemcpyAsync(note[t], d_note[t], (size_t)T * 88 * 4, hipMemcpyDeviceToHost, s));
      BP_HIP(hipMemcpyAsync(onset[t], d_onset[t], (size_t)T * 88 * 4, hipMemcpyDeviceToHost, s));
      BP_HIP(hipMemcpyAsync(contour[t], d_contour[t], (size_t)T * 264 * 4, hipMemcpyDeviceToHost, s));
    }
  }
  return wait_stream(h);
}

int64_t bp_resampled_length(int64_t n_frames, int sample_rate) {
  if (n_frames <= 0 || sample_rate <= 0) return 0;
  return (n_frames * (int64_t)BP_AUDIO_SAMPLE_RATE + sample_rate - 1) / sample_rate;
}

// downmix + resample into h->res_dev (or straight through when already mono 22.05 kHz on the device);
// *out = device pointer of the 22.05 kHz signal, *n_out = its length
static int pcm_width(int format) {
  switch (format) {
    case BP_PCM_F32: return 4;
    case BP_PCM_S16: return 2;
    case BP_PCM_S24: return 3;
    case BP_PCM_S32: return 4;
    case BP_PCM_U8: return 1;
    case BP_PCM_F64: return 8;
    default: return 0;
  }
}

static int ingest(bp_handle h, const void* pcm, int format, int64_t n_frames, int channels, int sample_rate, int mem_kind,
                  const float** out, int64_t* n_out) {
  const int width = pcm_width(format);
  if (n_frames < 0 || channels < 1 || channels > 64 || sample_rate < 1000 || sample_rate > 768000 || width == 0 ||
      (mem_kind != BP_MEM_HOST && mem_kind != BP_MEM_DEVICE) || (n_frames > 0 && !pcm)) {
    h->err = "audio ingest: bad argument (n_frames, channels, sample_rate, format, mem_kind or null pcm)";
    return BP_ERR_INVALID_ARG;
  }
  BP_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  *n_out = bp_handle_resampled_length(h, n_frames, sample_rate);
  *out = nullptr;
  if (n_frames == 0) return BP_OK;
  const void* d_pcm = pcm;
  if (mem_kind == BP_MEM_HOST) {
    const int64_t bytes = n_frames * channels * width;
    int rc = grow(h, &h->pcm_dev, &h->pcm_cap, (bytes + 3) / 4);
    if (rc) return rc;
    BP_HIP(hipMemcpyAsync(h->pcm_dev, pcm, (size_t)bytes, hipMemcpyHostToDevice, s));
    d_pcm = h->pcm_dev;
  }
  const float* d_mono = static_cast<const float*>(d_pcm);
  if (channels > 1 || format != BP_PCM_F32) {
    int rc = grow(h, &h->mono_dev, &h->mono_cap, n_frames);
    if (rc) return rc;
    if (format == BP_PCM_F32)
      launch_downmix(static_cast<const float*>(d_pcm), n_frames, channels, h->mono_dev, s);
    else
      launch_downmix_raw(d_pcm, format, n_frames, channels, h->mono_dev, s);
    d_mono = h->mono_dev;
  }
  if (sample_rate == h->rate) {
    *out = d_mono;
    return BP_OK;
  }
  if (h->taps_rate != sample_rate) {
    std::vector<double> taps;
    ResamplePlan pl = make_resample_plan(sample_rate, h->rate, taps);
    pl.rev_off = 0;
    if (!pl.direct && pl.up == 1 && pl.down == 2) {
      // the 2 : 1 kernel walks the taps backwards, a block of 32 per scalar load: a reversed copy behind the table, padded
      // with zeros to whole blocks (a zero tap adds x * 0 = 0 to a float64 sum)
      const size_t M = taps.size(), base = (M + 31) / 32 * 32, padded = (M + 31) / 32 * 32;
      taps.resize(base + padded, 0.0);
      for (size_t i = 0; i < M; ++i) taps[base + i] = taps[M - 1 - i];
      pl.rev_off = (int64_t)base;
    }
    if (h->taps_dev) BP_HIP(hipFree(h->taps_dev));
    h->taps_dev = nullptr;
    h->taps_rate = 0;
    BP_HIP(hipMalloc(&h->taps_dev, taps.size() * sizeof(double)));
    BP_HIP(hipMemcpy(h->taps_dev, taps.data(), taps.size() * sizeof(double), hipMemcpyHostToDevice));
    h->plan = pl;
    h->taps_rate = sample_rate;
  }
  int rc = grow(h, &h->res_dev, &h->res_cap, *n_out);
  if (rc) return rc;
  launch_resample(d_mono, n_frames, h->taps_dev, h->plan, h->res_dev, *n_out, h->resample_mode, s);
  BP_HIP(hipGetLastError());
  *out = h->res_dev;
  return BP_OK;
}

int bp_resample(bp_handle h, const float* pcm, int64_t n_frames, int channels, int sample_rate, float* out22k,
                int mem_kind) {
  if (!h) return BP_ERR_INVALID_ARG;
  const float* d = nullptr;
  int64_t n_out = 0;
  int rc = ingest(h, pcm, BP_PCM_F32, n_frames, channels, sample_rate, mem_kind, &d, &n_out);
  if (rc) return rc;
  if (n_out == 0) return BP_OK;
  if (!out22k) {
    h->err = "bp_resample: out22k is NULL";
    return BP_ERR_INVALID_ARG;
  }
  BP_HIP(hipMemcpyAsync(out22k, d, (size_t)n_out * 4,
                        mem_kind == BP_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, h->stream));
  BP_HIP(hipStreamSynchronize(h->stream));
  return BP_OK;
}

int bp_infer_pcm_raw(bp_handle h, const void* pcm, int format, int64_t n_frames, int channels, int sample_rate, float* note,
                     float* onset, float* contour, int mem_kind) {
  if (!h) return BP_ERR_INVALID_ARG;
  const float* d = nullptr;
  int64_t n = 0;
  int rc = ingest(h, pcm, format, n_frames, channels, sample_rate, mem_kind, &d, &n);
  if (rc) return rc;
  if (h_track_n_windows(h, n) == 0) return BP_OK;
  if (h_track_n_frames(h, n) > 0 && (!note || !onset || !contour)) {
    h->err = "bp_infer_pcm: null output pointer";
    return BP_ERR_INVALID_ARG;
  }
  return track_core(h, d, n, note, onset, contour, mem_kind);
}

int bp_infer_pcm(bp_handle h, const float* pcm, int64_t n_frames, int channels, int sample_rate, float* note,
                 float* onset, float* contour, int mem_kind) {
  return bp_infer_pcm_raw(h, pcm, BP_PCM_F32, n_frames, channels, sample_rate, note, onset, contour, mem_kind);
}

// ---- device-side note candidates (note_device.hip): what note decoding needs of a track's posteriorgrams
extern "C" void bp_internal_bend_tables(int32_t* tab, double* gauss);
extern "C" void bp_internal_freq_limits(const bp_note_params* prm, int* lo, int* hi);

// d_note / d_onset / d_contour: device maps of T frames (note / onset are modified when the parameters set a frequency
// range, like constrain_frequency does).  Outputs: host buffers.
static int candidates_core(bp_handle h, float* d_note, float* d_onset, const float* d_contour, int64_t T,
                           const bp_note_params* prm, float* note_out, uint8_t* cand_out, int8_t* bend_out, int* status) {
  hipStream_t s = h->stream;
  *status = 0;
  if (T <= 0) return wait_stream(h);
  constexpr size_t kTabBytes = 88 * 16, kGaussBytes = 51 * 8, kStatsBytes = 16;
  if (!h->nd_tables) {
    std::vector<float> raw((kTabBytes + kGaussBytes + kStatsBytes) / 4, 0.f);
    bp_internal_bend_tables(reinterpret_cast<int32_t*>(raw.data()), reinterpret_cast<double*>(raw.data() + kTabBytes / 4));
    int rc = upload(h, raw, &h->nd_tables);
    if (rc) return rc;
    BP_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->nd_stats_host), kStatsBytes, hipHostMallocPortable));
    BP_HIP(hipHostGetDevicePointer(&h->nd_stats_host_dev, h->nd_stats_host, 0));
  }
  const int64_t bits_bytes = T * BP_NOTE_CAND_ROW_BYTES, bend_bytes = T * 88;  // multiples of 4
  int rc = grow(h, &h->nd_buf, &h->nd_cap, (((bits_bytes + 15) & ~(int64_t)15) + bend_bytes + 3) / 4);
  if (rc) return rc;
  uint8_t* d_bits = reinterpret_cast<uint8_t*>(h->nd_buf);
  int8_t* d_bend = reinterpret_cast<int8_t*>(d_bits + ((bits_bytes + 15) & ~(int64_t)15));
  char* tables = reinterpret_cast<char*>(h->nd_tables);
  void* d_stats = tables + kTabBytes + kGaussBytes;
  int lo = 0, hi = 88;
  bp_internal_freq_limits(prm, &lo, &hi);
  const bool want_bends = prm->include_pitch_bends != 0 && bend_out != nullptr;
  if (!h->nd_stats_ready) launch_note_stats_init(d_stats, s);
  h->nd_stats_ready = false;
  launch_note_candidates(d_note, d_onset, d_contour, T, lo, hi, prm->infer_onsets != 0, prm->onset_threshold, tables,
                         reinterpret_cast<const double*>(tables + kTabBytes), d_stats, d_bits, want_bends ? d_bend : nullptr, s);
  BP_HIP(hipGetLastError());
  // The results go home.  Into page-locked buffers (bp_host_alloc) a kernel of this stream writes them over PCIe itself:
  // the copy engine serialises the copies of all lanes in both directions (measured: a lane's 27 MB of posteriorgrams
  // going out kept the next file's samples from coming in), and it is busy with the inbound samples.  Pageable
  // destinations take ordinary copies.
  // (the attributes describe the START of a buffer only — a pointer into a page-locked block that ends before
  // `bytes` would send the kernel's writes past the registration.  The whole extent must lie inside the allocation the
  // pointer belongs to: hipMemGetAddressRange gives its base and size for the device view of a page-locked block; where that
  // cannot be established the copies take over.)
  auto device_view = [](void* p, size_t bytes) -> void* {
    if (!p) return nullptr;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    if (at.type != hipMemoryTypeHost || !at.devicePointer) return nullptr;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, at.devicePointer) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    const uintptr_t b0 = reinterpret_cast<uintptr_t>(base), q = reinterpret_cast<uintptr_t>(at.devicePointer);
    return (q >= b0 && q + bytes <= b0 + size) ? at.devicePointer : nullptr;
  };
  bool exported_by_kernel = false;
  void *v_note = device_view(note_out, (size_t)T * 88 * 4), *v_bits = device_view(cand_out, (size_t)bits_bytes),
       *v_bend = want_bends ? device_view(bend_out, (size_t)bend_bytes) : nullptr;
  const bool aligned = !((reinterpret_cast<uintptr_t>(v_note) | reinterpret_cast<uintptr_t>(v_bits) | reinterpret_cast<uintptr_t>(v_bend)) & 3);
  if (v_note && v_bits && (v_bend || !want_bends) && aligned) {
    // ... the stats record with them; the same kernel leaves the device record initialised for the next track
    launch_note_export(d_note, v_note, T * 88 * 4, d_bits, v_bits, bits_bytes, d_bend, v_bend, bend_bytes, d_stats,
                       h->nd_stats_host_dev, s);
    BP_HIP(hipGetLastError());
    exported_by_kernel = true;
  } else {
    BP_HIP(hipMemcpyAsync(h->nd_stats_host, d_stats, kStatsBytes, hipMemcpyDeviceToHost, s));
    BP_HIP(hipMemcpyAsync(note_out, d_note, (size_t)T * 88 * 4, hipMemcpyDeviceToHost, s));
    BP_HIP(hipMemcpyAsync(cand_out, d_bits, (size_t)bits_bytes, hipMemcpyDeviceToHost, s));
    if (want_bends) BP_HIP(hipMemcpyAsync(bend_out, d_bend, (size_t)bend_bytes, hipMemcpyDeviceToHost, s));
  }
  rc = wait_stream(h);
  if (rc) return rc;
  if (exported_by_kernel) h->nd_stats_ready = true;  // only now: the export kernel, which re-initialises the record, has run
  const int nan_flag = reinterpret_cast<const int*>(h->nd_stats_host)[1];
  // numpy's rules for NaN cells, and an onset threshold <= 0 (every cell that is not a peak qualifies), need the maps
  // themselves: the host decoder takes over (bp_infer_* + bp_notes_decode)
  if (nan_flag || !(prm->onset_threshold > 0.0)) *status = 1;
  return BP_OK;
}

int bp_note_candidates(bp_handle h, const float* note, const float* onset, const float* contour, int64_t n_frames,
                       const bp_note_params* params, int mem_kind, float* note_out, uint8_t* cand_bits, int8_t* bend_map,
                       int* status) {
  if (!h) return BP_ERR_INVALID_ARG;
  if (!params || !status || n_frames < 0 || (mem_kind != BP_MEM_HOST && mem_kind != BP_MEM_DEVICE) ||
      (n_frames > 0 && (!note || !onset || !contour || !note_out || !cand_bits))) {
    h->err = "bp_note_candidates: null pointer, negative frame count or bad mem_kind";
    return BP_ERR_INVALID_ARG;
  }
  BP_HIP(hipSetDevice(h->device));
  const int64_t T = n_frames;
  // a private copy on the device: the frequency limits are applied in place
  int rc = grow(h, &h->track_out, &h->track_out_cap, T * (88 + 88 + 264));
  if (rc) return rc;
  float *d_note = h->track_out, *d_onset = d_note + T * 88, *d_contour = d_onset + T * 88;
  const hipMemcpyKind kind = mem_kind == BP_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  if (T > 0) {
    BP_HIP(hipMemcpyAsync(d_note, note, (size_t)T * 88 * 4, kind, h->stream));
    BP_HIP(hipMemcpyAsync(d_onset, onset, (size_t)T * 88 * 4, kind, h->stream));
    BP_HIP(hipMemcpyAsync(d_contour, contour, (size_t)T * 264 * 4, kind, h->stream));
  }
  return candidates_core(h, d_note, d_onset, d_contour, T, params, note_out, cand_bits, bend_map, status);
}

int bp_infer_pcm_raw_candidates(bp_handle h, const void* pcm, int format, int64_t n_frames, int channels, int sample_rate,
                                const bp_note_params* params, float* note_out, uint8_t* cand_bits, int8_t* bend_map,
                                int* status) {
  if (!h) return BP_ERR_INVALID_ARG;
  if (!params || !status) {
    h->err = "bp_infer_pcm_raw_candidates: null params / status";
    return BP_ERR_INVALID_ARG;
  }
  const float* d = nullptr;
  int64_t n = 0;
  int rc = ingest(h, pcm, format, n_frames, channels, sample_rate, BP_MEM_HOST, &d, &n);
  if (rc) return rc;
  *status = 0;
  const int64_t T = h_track_n_frames(h, n);
  if (h_track_n_windows(h, n) == 0 || T == 0) return wait_stream(h);
  if (!note_out || !cand_bits) {
    h->err = "bp_infer_pcm_raw_candidates: null output pointer";
    return BP_ERR_INVALID_ARG;
  }
  rc = track_core(h, d, n, nullptr, nullptr, nullptr, kTrackOutInternal);
  if (rc) return rc;
  float *d_note = h->track_out, *d_onset = d_note + T * 88, *d_contour = d_onset + T * 88;
  return candidates_core(h, d_note, d_onset, d_contour, T, params, note_out, cand_bits, bend_map, status);
}

int bp_track_maps(bp_handle h, int64_t n_frames, float* note, float* onset, float* contour, int mem_kind) {
  if (!h) return BP_ERR_INVALID_ARG;
  if (n_frames <= 0 || n_frames != h->maps_rows || !note || !onset || !contour ||
      (mem_kind != BP_MEM_HOST && mem_kind != BP_MEM_DEVICE)) {
    h->err = "bp_track_maps: no maps of that many rows are left on the device (call it right after a *_candidates call of "
             "this handle, with that call's row count) or a null / unknown destination";
    return BP_ERR_INVALID_ARG;
  }
  BP_HIP(hipSetDevice(h->device));
  const int64_t T = n_frames;
  const float *d_note = h->track_out, *d_onset = d_note + T * 88, *d_contour = d_onset + T * 88;
  const hipMemcpyKind kind = mem_kind == BP_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  BP_HIP(hipMemcpyAsync(note, d_note, (size_t)T * 88 * 4, kind, h->stream));
  BP_HIP(hipMemcpyAsync(onset, d_onset, (size_t)T * 88 * 4, kind, h->stream));
  BP_HIP(hipMemcpyAsync(contour, d_contour, (size_t)T * 264 * 4, kind, h->stream));
  return wait_stream(h);
}

// ---- FLAC decoded on the device (flac_device.hip) ---------------------------------------------------------------------------
// The file's bytes to the device, the three decode launches queued on the handle's stream, the error bits on their way to a
// page-locked word; *fmt / *lay describe the PCM now (being) written to h->pcm_dev.
static int flac_to_device_pcm(bp_handle h, const void* file, size_t nbytes, bp_flac_stream_layout* lay, int* fmt) {
  if (!file || nbytes < 42) {
    h->err = "FLAC on the device: null or too short";
    return BP_ERR_INVALID_ARG;
  }
  if (bp_flac_layout(file, nbytes, lay) != BP_OK) {
    h->err = std::string("FLAC on the device: ") + bp_audio_last_error();
    return BP_ERR_BAD_AUDIO;
  }
  if (lay->n_frames <= 0 || lay->min_block < 16 || lay->max_block < lay->min_block || lay->bits_per_sample > 24 ||
      lay->bits_per_sample < 4 || lay->channels > 8 || nbytes >= ((size_t)1 << 31) ||
      lay->n_frames * lay->channels >= ((int64_t)1 << 33)) {
    h->err = "FLAC on the device: a stream the device decoder leaves to the host (no sample count / block sizes in STREAMINFO, "
             "more than 24 bits or 8 channels, or 2 GB and more)";
    return BP_ERR_UNSUPPORTED;
  }
  BP_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  if (!h->fd_status_host) BP_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->fd_status_host), 2 * sizeof(int), hipHostMallocPortable));
  if (nbytes + 64 > h->fd.file_cap) {
    if (h->fd.file) BP_HIP(hipFree(h->fd.file));
    h->fd.file = nullptr, h->fd.file_cap = 0;
    const size_t cap = nbytes + nbytes / 4 + 4096;
    BP_HIP(hipMalloc(&h->fd.file, cap));
    h->fd.file_cap = cap;
  }
  BP_HIP(hipMemcpyAsync(h->fd.file, file, nbytes, hipMemcpyHostToDevice, s));
  BP_HIP(hipMemsetAsync(h->fd.file + nbytes, 0, 64, s));
  const int wide = lay->bits_per_sample > 16;
  *fmt = wide ? BP_PCM_S32 : BP_PCM_S16;
  const int64_t bytes = lay->n_frames * lay->channels * (wide ? 4 : 2);
  int rc = grow(h, &h->pcm_dev, &h->pcm_cap, (bytes + 3) / 4);
  if (rc) return rc;
  FdStream st{lay->channels, lay->bits_per_sample, lay->min_block, lay->max_block, lay->n_frames, (uint32_t)lay->audio_start,
              (uint32_t)nbytes};
  if (flac_device_decode(h->fd, st, h->pcm_dev, s) != 0) {
    h->err = "FLAC on the device: allocation or launch failed";
    (void)hipGetLastError();
    return BP_ERR_HIP;
  }
  BP_HIP(hipMemcpyAsync(h->fd_status_host, h->fd.meta, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
  return BP_OK;
}

// after the stream has been waited for: what the decode kernels reported
static int flac_device_verdict(bp_handle h) {
  const int st = h->fd_status_host ? h->fd_status_host[0] : 0;
  if (st == 0) return BP_OK;
  h->err = std::string("FLAC on the device: the stream could not be decoded (") + ((st & 2) ? "frame chain " : "") +
           ((st & 4) ? "CRC-16 " : "") + ((st & 8) ? "reserved value / overrun " : "") + ((st & 16) ? "candidate overflow " : "") +
           "); the host decoder (bp_flac_decode) reports the cause";
  return BP_ERR_BAD_AUDIO;
}

int bp_flac_decode_device(bp_handle h, const void* file, size_t nbytes, int32_t* pcm, int64_t capacity_frames, int64_t* n_frames) {
  if (!h || !n_frames) return BP_ERR_INVALID_ARG;
  bp_flac_stream_layout lay;
  int fmt = 0;
  int rc = flac_to_device_pcm(h, file, nbytes, &lay, &fmt);
  if (rc) return rc;
  rc = wait_stream(h);
  if (rc) return rc;
  rc = flac_device_verdict(h);
  if (rc) return rc;
  *n_frames = lay.n_frames;
  if (!pcm) return BP_OK;
  if (capacity_frames < lay.n_frames) {
    h->err = "bp_flac_decode_device: the output buffer is too small";
    return BP_ERR_INVALID_ARG;
  }
  const int64_t n = lay.n_frames * lay.channels;
  if (fmt == BP_PCM_S32) {
    BP_HIP(hipMemcpy(pcm, h->pcm_dev, (size_t)n * 4, hipMemcpyDeviceToHost));
    const int sh = 32 - lay.bits_per_sample;
    for (int64_t i = 0; i < n; ++i) pcm[i] >>= sh;  // left-justified on the device
  } else {
    std::vector<int16_t> tmp((size_t)n);
    BP_HIP(hipMemcpy(tmp.data(), h->pcm_dev, (size_t)n * 2, hipMemcpyDeviceToHost));
    const int sh = 16 - lay.bits_per_sample;
    for (int64_t i = 0; i < n; ++i) pcm[i] = (int32_t)tmp[(size_t)i] >> sh;
  }
  return BP_OK;
}

int bp_infer_flac(bp_handle h, const void* file, size_t nbytes, float* note, float* onset, float* contour, int mem_kind) {
  if (!h) return BP_ERR_INVALID_ARG;
  bp_flac_stream_layout lay;
  int fmt = 0;
  int rc = flac_to_device_pcm(h, file, nbytes, &lay, &fmt);
  if (rc) return rc;
  const float* d = nullptr;
  int64_t n = 0;
  rc = ingest(h, h->pcm_dev, fmt, lay.n_frames, lay.channels, lay.sample_rate, BP_MEM_DEVICE, &d, &n);
  if (rc) return rc;
  if (h_track_n_windows(h, n) == 0) {
    rc = wait_stream(h);
    return rc ? rc : flac_device_verdict(h);
  }
  if (h_track_n_frames(h, n) > 0 && (!note || !onset || !contour)) {
    h->err = "bp_infer_flac: null output pointer";
    return BP_ERR_INVALID_ARG;
  }
  rc = track_core(h, d, n, note, onset, contour, mem_kind);
  if (rc) return rc;
  if (mem_kind != BP_MEM_HOST) {
    rc = wait_stream(h);
    if (rc) return rc;
  }
  return flac_device_verdict(h);
}

int bp_infer_flac_candidates(bp_handle h, const void* file, size_t nbytes, const bp_note_params* params, float* note_out,
                             uint8_t* cand_bits, int8_t* bend_map, int* status) {
  if (!h) return BP_ERR_INVALID_ARG;
  if (!params || !status) {
    h->err = "bp_infer_flac_candidates: null params / status";
    return BP_ERR_INVALID_ARG;
  }
  bp_flac_stream_layout lay;
  int fmt = 0;
  int rc = flac_to_device_pcm(h, file, nbytes, &lay, &fmt);
  if (rc) return rc;
  const float* d = nullptr;
  int64_t n = 0;
  rc = ingest(h, h->pcm_dev, fmt, lay.n_frames, lay.channels, lay.sample_rate, BP_MEM_DEVICE, &d, &n);
  if (rc) return rc;
  *status = 0;
  const int64_t T = h_track_n_frames(h, n);
  if (h_track_n_windows(h, n) == 0 || T == 0) {
    rc = wait_stream(h);
    return rc ? rc : flac_device_verdict(h);
  }
  if (!note_out || !cand_bits) {
    h->err = "bp_infer_flac_candidates: null output pointer";
    return BP_ERR_INVALID_ARG;
  }
  rc = track_core(h, d, n, nullptr, nullptr, nullptr, kTrackOutInternal);
  if (rc) return rc;
  float *d_note = h->track_out, *d_onset = d_note + T * 88, *d_contour = d_onset + T * 88;
  rc = candidates_core(h, d_note, d_onset, d_contour, T, params, note_out, cand_bits, bend_map, status);
  if (rc) return rc;
  return flac_device_verdict(h);
}

void* bp_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}

void bp_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

int bp_get_stage_ms(bp_handle h, float* ms, int n) {
  if (!h || !ms || n < BP_N_STAGES) return BP_ERR_INVALID_ARG;
  if (!(h->flags & (BP_FLAG_STAGE_TIMING | BP_FLAG_TIME_DOMINANT)) || h->timed_chunks == 0) {
    h->err = "bp_get_stage_ms: handle was not created with BP_FLAG_STAGE_TIMING / BP_FLAG_TIME_DOMINANT or nothing ran yet";
    return BP_ERR_UNSUPPORTED;
  }
  BP_HIP(hipStreamSynchronize(h->stream));
  const int64_t cnt = h->timed_chunks < bp_context::kTimedRing ? h->timed_chunks : bp_context::kTimedRing;
  double acc[BP_N_STAGES] = {0};
  for (int64_t c = 0; c < cnt; ++c) {
    for (int i = 0; i < h->n_seq[c]; ++i) {
      float t = 0.f;
      if (h->seq[c][i] < 0) continue;
      BP_HIP(hipEventElapsedTime(&t, h->ev[c][i], h->ev[c][i + 1]));
      acc[h->seq[c][i]] += t;
    }
  }
  for (int i = 0; i < BP_N_STAGES; ++i) ms[i] = (float)(acc[i] / (double)cnt);
  h->timed_chunks = 0;
  h->dom_chunks = 0;  // the first chunk after a read-out is a sampled one
  return BP_OK;
}

int bp_pyramid_layout(int level, int64_t* offset, int64_t* length) {
  if (level < 1 || level > 8 || !offset || !length) return BP_ERR_INVALID_ARG;
  *offset = pyr_off(level);
  *length = level_len(level);
  return BP_OK;
}

int bp_run_stage(bp_handle h, int stage, const bp_stage_buffers* bf, int64_t n_windows) {
  if (!h || !bf || n_windows <= 0 || n_windows > (1 << 20)) return BP_ERR_INVALID_ARG;
  BP_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const int n = (int)n_windows;
  auto need = [&](const void* p) { return p != nullptr; };
  const bool wlo = !(h->flags & BP_FLAG_BF16_WEIGHTS);
  bool ok = true;
  switch (stage) {
    case BP_STAGE_PYRAMID:
      if ((ok = need(bf->audio) && need(bf->pyr))) {
        if (h->flags & BP_FLAG_F32_MFMA) {
          launch_pyramid(bf->audio, bf->pyr, h->d_lowpass, n, s);
        } else {  // the planes pyramid, its levels converted to the fp32 rows the test compares
          if (n > h->cap) {
            h->err = "bp_run_stage: pyramid needs n_windows <= max_windows (internal planes buffer)";
            return BP_ERR_INVALID_ARG;
          }
          uint16_t* pl = reinterpret_cast<uint16_t*>(h->planes);
          launch_pyramid_planes(bf->audio, h->win_len, pl, h->d_pl_tfrag, n, h->n_cu, h->ext, s);
          const int n_lev = h->ext ? kOctavesExt : kOctaves;
          for (int k = 1; k < n_lev; ++k) {
            const int64_t off = h->ext ? ((k == 1) ? 0 : kAudioN + pyr_off(k - 1)) : pyr_off(k);
            launch_planes_unsplit(pl, k, bf->pyr + off, h->pyr_stride, n, h->ext, s);
          }
        }
      }
      break;
    case BP_STAGE_FILTERBANK:
      if ((ok = need(bf->audio) && need(bf->pyr) && need(bf->lp) && need(bf->mm))) {
        int rc = ensure_fb_scratch(h, n);
        if (rc) return rc;
        if (h->flags & BP_FLAG_F32_MFMA)
          launch_filterbank(bf->audio, bf->pyr, h->d_fb_bfrag, h->d_sqrt_len, bf->lp, bf->mm, h->fb_scratch, n,
                            h->kc, h->n_cu, s);
        else {  // the given fp32 levels split into planes (test hook), then the planes filterbank
          if (n > h->cap) {
            h->err = "bp_run_stage: filterbank needs n_windows <= max_windows (internal planes buffer)";
            return BP_ERR_INVALID_ARG;
          }
          uint16_t* pl = reinterpret_cast<uint16_t*>(h->planes);
          launch_planes_edge_rows(bf->audio, h->win_len, pl, n, h->ext, s);  // level 0: fp32, straight from the audio
          const int n_lev = h->ext ? kOctavesExt : kOctaves;
          for (int k = 1; k < n_lev; ++k) {
            const int64_t off = h->ext ? ((k == 1) ? 0 : kAudioN + pyr_off(k - 1)) : pyr_off(k);
            launch_planes_split(bf->pyr + off, h->pyr_stride, k, pl, n, h->ext, s);
          }
          (void)launch_filterbank_planes(pl, bf->audio, h->win_len, h->d_pl_bfrag, h->d_pl_bin_k, bf->lp, h->fb_scratch, nullptr, n, h->kc, h->n_cu,
                                         h->ext, s);
          launch_mm_reduce(h->fb_scratch, bf->mm, n, filterbank_planes_partials(h->ext), s);
        }
      }
      break;
    case BP_STAGE_CONTOUR1:
      if ((ok = need(bf->lp) && need(bf->mm) && need(bf->c1)))
        launch_contour1(bf->lp, bf->mm, h->d_c1_bfrag, h->d_c1_bias, bf->c1, n, h->kc, h->n_cu, s);
      break;
    case BP_STAGE_CONTOUR2:
      if ((ok = need(bf->c1) && need(bf->contour)))
        launch_contour2(bf->c1, h->d_w_contour2, h->b_contour2, bf->contour, n, s);
      break;
    case BP_STAGE_NOTE1:
      if ((ok = need(bf->contour) && need(bf->n1)))
        launch_note1(bf->contour, h->d_n1_bfrag, h->d_n1_bias, bf->n1, n, h->n_cu, s);
      break;
    case BP_STAGE_NOTE2:
      if ((ok = need(bf->n1) && need(bf->note))) launch_note2(bf->n1, h->d_w_note2, h->b_note2, bf->note, n, s);
      break;
    case BP_STAGE_ONSET1:
      if ((ok = need(bf->lp) && need(bf->mm) && need(bf->o1)))
        launch_onset1(bf->lp, bf->mm, h->d_o1_bfrag, h->d_o1_bias, bf->o1, n, h->kc, h->n_cu, s);
      break;
    case BP_STAGE_ONSET2:
      if ((ok = need(bf->note) && need(bf->o1) && need(bf->onset)))
        launch_onset2(bf->note, bf->o1, h->d_w_onset2, h->b_onset2, bf->onset, n, s);
      break;
    case BP_STAGE_ZPACK:
      if ((ok = need(bf->lp) && need(bf->mm) && need(bf->zp))) launch_zpack(bf->lp, bf->mm, bf->zp, n, h->kc, h->n_bins, s);
      break;
    case BP_STAGE_CONTOUR:
      if ((ok = need(bf->zp) && need(bf->contour))) {
        if (n > h->cap) {
          h->err = "bp_run_stage: contour needs n_windows <= max_windows (internal c1 buffer)";
          return BP_ERR_INVALID_ARG;
        } else {
#ifdef BP_AB_KERNELS
          if (contour_conv1_full() || h->rim_exact)
            launch_contour_conv1_exact(bf->zp, h->d_d1_wlds, h->d_d1_bias, h->c1s, n, h->n_cu, wlo, s);
          else
#endif
            launch_rim(h, bf->zp, h->c1s, n, wlo, s);
#ifdef BP_AB_KERNELS
          if (h->fold_mx && wlo) {
            const char* base = reinterpret_cast<const char*>(h->d_d1_wfold_mx);
            launch_contour_conv1_fold_mx(bf->zp, base, base + 36 * 64 * 16, base + 36 * 64 * 16 + 18 * 64 * 32, h->d_d1_bias,
                                         h->c1s, n, h->n_cu, s);
          } else if (!contour_conv1_use_march()) {
            launch_contour_conv1_folded(bf->zp, h->d_d1_wfold, h->d_d1_bias, h->c1s, n, h->n_cu, wlo, s);
          } else
#endif
            launch_contour_conv1_march(bf->zp, h->d_d1_wmarch, h->d_d1_bias, h->c1s, n, h->n_cu, wlo, s);
          launch_conv2(h->c1s, h->d_d2_w, h->d_d2_wproj, h->b_contour2, bf->contour, n, h->n_cu, wlo, s);
        }
      }
      break;
    case BP_STAGE_NOTE:
      if ((ok = need(bf->contour) && need(bf->note)))
        launch_note(bf->contour, h->d_note_wfrag, h->d_note_w16, h->d_note_wf32, bf->note, n, h->n_cu, wlo, s);
      break;
    case BP_STAGE_ONSET:
      if ((ok = need(bf->zp) && need(bf->note) && need(bf->onset)))
        launch_onset(bf->zp, bf->note, h->d_onset_wfrag, h->d_onset_wf32, h->d_onset_wmx, h->d_onset_w16, bf->onset, n, h->n_cu, wlo, s);
      break;
    default:
      h->err = "bp_run_stage: unknown stage";
      return BP_ERR_INVALID_ARG;
  }
  if (!ok) {
    h->err = "bp_run_stage: a buffer this stage needs is NULL";
    return BP_ERR_INVALID_ARG;
  }
  BP_HIP(hipGetLastError());
  static const bool nosync = ab_env("BP_STAGE_NOSYNC") != nullptr;  // tools only: overlap experiments
  if (!nosync) BP_HIP(hipStreamSynchronize(s));
  return BP_OK;
}

}  // extern "C"
