// Note decoding: posteriorgrams -> note events (host C++).
//
// Replaces, behind the C ABI entry point bp_notes_decode (include/basic_pitch_amd.h), the Python loops of
// basic_pitch/note_creation.py (spotify/basic-pitch v0.4.0):
//   constrain_frequency            note_creation.py:314-343
//   get_infered_onsets             note_creation.py:289-311
//   output_to_notes_polyphonic     note_creation.py:360-511   (melodia trick: 452-509)
//   get_pitch_bends                note_creation.py:182-219
//   model_frames_to_time           note_creation.py:346-357
//   model_output_to_notes          note_creation.py:52-116    (without the PrettyMIDI object)
//
// The reference is single-threaded numpy/Python: 9.5 ms for the 9-second test clip, 2.07 s for a 3-minute
// track and super-linear, because every melodia iteration rescans the whole matrix for its maximum
// (np.max + np.argmax, note_creation.py:452-453).  Here the maximum comes from a tournament tree with the
// same tie-break (lowest flat index), so an iteration costs O(log n) per cell it zeroes.  Round 4: the float64 onset
// map, the frame-difference map and the float64 copy of the note map (33 MB per 3-minute track) are not built — an
// inferred onset is three loads from the float32 maps, evaluated exactly only in frames whose row maxima can reach the
// threshold: 26 -> 7 ms per 3-minute track and core, same events.  NaN cells and thresholds <= 0 follow numpy's
// propagation rules (tests/test_note_decode.py::test_decoder_fuzz_against_restatement).
//
// Bit-exactness contract (tests/test_note_decode.py): same events, in the same order, as the numpy
// restatement oracle/note_oracle.py — which reproduces the reference's golden note events — including the
// float32 pairwise summation numpy uses for np.mean (amplitudes), float64 everywhere the reference is
// float64, round-half-even where it calls np.round.
#include "../../include/basic_pitch_amd.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace {

constexpr int kF = BP_N_FREQ_NOTE;      // 88
constexpr int kFC = BP_N_FREQ_CONTOUR;  // 264
constexpr int kMidiOffset = 21;         // note_creation.py:40
constexpr int kMaxFreqIdx = 87;         // note_creation.py:43

// numpy's pairwise summation for float32 (numpy/_core/src/umath/loops_utils.h.src, *_pairwise_sum):
// the reduction np.mean / np.add.reduce run on a (strided) 1-D float32 view.
float pairwise_sum_f32(const float* a, int64_t n, int64_t stride) {
  if (n < 8) {
    float res = 0.0f;
    for (int64_t i = 0; i < n; ++i) res += a[i * stride];
    return res;
  }
  if (n <= 128) {
    float r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j * stride];
    int64_t i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[(i + j) * stride];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i * stride];
    return res;
  }
  int64_t n2 = n / 2;
  n2 -= n2 % 8;
  return pairwise_sum_f32(a, n2, stride) + pairwise_sum_f32(a + n2 * stride, n - n2, stride);
}

// np.mean of frames[a:b, f] for a float32 matrix: float32 accumulator, float32 division
float mean_f32(const float* frames, int64_t a, int64_t b, int f) {
  const int64_t n = b - a;
  const float s = 0.0f + pairwise_sum_f32(frames + a * kF + f, n, kF);
  return s / (float)n;
}

// np.round (round half to even) of a double, as an int
int64_t round_half_even(double v) { return (int64_t)std::nearbyint(v); }

// np.maximum: NaN if either operand is NaN
inline double np_maximum(double a, double b) {
  if (std::isnan(a) || std::isnan(b)) return std::nan("");
  return a > b ? a : b;
}

// Tournament tree for the melodia trick: index of the maximum of `val`, ties -> lowest index (np.argmax), cells can only
// be zeroed.  The loop that uses it stops at the first maximum that is not above the frame threshold, so only cells above
// the threshold can ever be returned: the tree is built over THOSE (in index order, a few per cent of a 3-minute map after
// the peak-picking pass) instead of all T x 88 cells — building the full 2^21-leaf tree was 40 % of a decode.  A cell's leaf
// is found by binary search in the sorted candidate list.
class MaxTree {
 public:
  MaxTree(std::vector<float>& v, double thresh, std::vector<int32_t>& cand, std::vector<int32_t>& node)
      : val_(v), cand_(cand), node_(node) {
    cand_.clear();
    const int64_t n = (int64_t)v.size();
    // (double)x > thresh  <=>  x > the largest float that is <= thresh; whole frames without such a cell are skipped
    float tf = (float)thresh;
    if ((double)tf > thresh) tf = std::nextafterf(tf, -INFINITY);
    const float* p = v.data();
    int64_t i = 0;
    for (; i + 88 <= n; i += 88) {
      int any = 0, nan = 0;
      for (int j = 0; j < 88; ++j) {
        any |= p[i + j] > tf;
        nan |= p[i + j] != p[i + j];
      }
      has_nan_ |= nan != 0;
      if (!any) continue;
      for (int j = 0; j < 88; ++j)
        if (p[i + j] > tf) cand_.push_back((int32_t)(i + j));
    }
    for (; i < n; ++i) {
      if (p[i] > tf) cand_.push_back((int32_t)i);
      has_nan_ |= p[i] != p[i];
    }
    nc_ = (int64_t)cand_.size();
    size_ = 1;
    while (size_ < nc_) size_ <<= 1;
    node_.assign((size_t)(2 * size_), -1);
    for (int64_t i = 0; i < nc_; ++i) node_[size_ + i] = cand_[(size_t)i];
    for (int64_t i = size_ - 1; i >= 1; --i) node_[i] = better(node_[2 * i], node_[2 * i + 1]);
  }
  // the maximum cell, or -1 when no cell above the threshold is left
  int32_t argmax() const { return nc_ ? node_[1] : -1; }
  // np.max of a map that holds a NaN is NaN
  bool has_nan() const { return has_nan_; }
  void set_zero(int64_t idx) {
    if (val_[idx] == 0.0f) return;
    val_[idx] = 0.0f;
    const auto it = std::lower_bound(cand_.begin(), cand_.end(), (int32_t)idx);
    if (it == cand_.end() || *it != (int32_t)idx) return;  // never a candidate: not in the tree
    for (int64_t i = (size_ + (it - cand_.begin())) >> 1; i >= 1; i >>= 1) node_[i] = better(node_[2 * i], node_[2 * i + 1]);
  }

 private:
  int32_t better(int32_t a, int32_t b) const {
    if (a < 0) return b;
    if (b < 0) return a;
    return (val_[b] > val_[a]) ? b : a;  // a < b always (left child first): ties keep the lower index
  }
  std::vector<float>& val_;
  std::vector<int32_t>& cand_;
  std::vector<int32_t>& node_;
  int64_t nc_ = 0, size_ = 1;
  bool has_nan_ = false;
};

// per-thread scratch: a 3-minute track needs 11 MB per T x 88 double map; a worker thread of the file pipeline decodes
// hundreds of tracks and would otherwise fault those pages in again for every one of them
struct Scratch {
  std::vector<float> energy, row_on;
  std::vector<double> row_fd;
  std::vector<int32_t> cand, node;
};
thread_local Scratch g_scratch;

thread_local std::string g_notes_error;

}  // namespace

extern "C" {

const char* bp_notes_last_error(void) { return g_notes_error.c_str(); }

void bp_note_params_default(bp_note_params* p) {
  if (!p) return;
  p->onset_threshold = 0.5;                                         // inference.py:434
  p->frame_threshold = 0.3;                                         // inference.py:435
  p->min_note_len = 11;                                             // note_creation.py:45 DEFAULT_MIN_NOTE_LEN
  p->infer_onsets = 1;
  p->melodia_trick = 1;
  p->include_pitch_bends = 1;
  p->energy_tol = 11;                                               // note_creation.py:46
  p->min_freq_hz = 0.0;
  p->max_freq_hz = 0.0;
}

// The decoder.  Two sources for the onset peaks and the pitch bends:
//   cand_bits == null: the onset and contour maps (bp_notes_decode: everything on the host);
//   cand_bits != null: the device extracted them (bp_note_candidates, csrc/note_device.hip): bit f of byte row t
//     ([T][12] bytes, BP_NOTE_CAND_ROW_BYTES) marks an onset peak that reaches the threshold, bend_map [T][88] holds the pitch bend of bin f at
//     frame t; `note` is already frequency-constrained, `onset` / `contour` are not looked at.
static int decode_core(float* note, float* onset, const float* contour, const uint8_t* cand_bits, const int8_t* bend_map,
                       int64_t n_frames, const bp_note_params* prm, bp_note_event* events, int64_t max_events,
                       int32_t* bends, int64_t max_bends, int64_t* n_events_out, int64_t* n_bends_out) {
  const bool cand = cand_bits != nullptr;
  if (!prm || !n_events_out || !n_bends_out || n_frames < 0 ||
      (n_frames > 0 && (!note || (!cand && (!onset || !contour)) || (cand && prm->include_pitch_bends && !bend_map)))) {
    g_notes_error = "bp_notes_decode: null pointer or negative frame count";
    return BP_ERR_INVALID_ARG;
  }
  if (n_frames > (int64_t)1 << 24) {
    g_notes_error = "bp_notes_decode: more than 2^24 frames";
    return BP_ERR_INVALID_ARG;
  }
  if (prm->melodia_trick && prm->frame_threshold < 0.0) {
    // `while np.max(remaining_energy) > frame_thresh` (note_creation.py:452) never ends below zero: the cells it zeroes
    // stay above the threshold.  The reference hangs; this reports.
    g_notes_error = "bp_notes_decode: a negative frame threshold with the melodia trick never terminates (note_creation.py:452)";
    return BP_ERR_INVALID_ARG;
  }
  *n_events_out = 0;
  *n_bends_out = 0;
  const int64_t T = n_frames;
  if (T == 0) return BP_OK;

  if (cand && !(prm->onset_threshold > 0.0)) {
    g_notes_error = "bp_notes_decode_candidates: an onset threshold <= 0 needs the onset map (every cell that is not a peak qualifies)";
    return BP_ERR_INVALID_ARG;
  }
  // ---- constrain_frequency (in place, note_creation.py:338-341)
  int64_t min_idx = 0, max_idx = kF;
  if (prm->min_freq_hz > 0.0)
    min_idx = round_half_even(12.0 * (std::log2(prm->min_freq_hz) - std::log2(440.0)) + 69.0 - kMidiOffset);
  if (prm->max_freq_hz > 0.0)
    max_idx = round_half_even(12.0 * (std::log2(prm->max_freq_hz) - std::log2(440.0)) + 69.0 - kMidiOffset);
  {
    // numpy slice semantics: [:a] and [b:] with negative / overshooting bounds
    auto norm = [](int64_t i) { return i < 0 ? (i + kF < 0 ? 0 : i + kF) : (i > kF ? kF : i); };
    const int64_t lo = norm(min_idx), hi = norm(max_idx);
    for (int64_t t = 0; t < T && !cand; ++t) {
      for (int64_t f = 0; f < lo; ++f) note[t * kF + f] = onset[t * kF + f] = 0.0f;
      for (int64_t f = hi; f < kF; ++f) note[t * kF + f] = onset[t * kF + f] = 0.0f;
    }
  }

  // ---- onsets (float64 when inferred, note_creation.py:289-311).  The float64 onset map is never stored: a cell of it is
  // max(onset, max_on * fd / max_fd) with fd = max(0, min(note[t] - note[t-1], note[t] - note[t-2])), three loads away
  // from the float32 maps, and the peak picking below needs it exactly only where it can reach the onset threshold.
  const bool infer = prm->infer_onsets != 0;
  double max_on_d = 0.0, max_fd = 0.0;
  // per frame: the largest onset and the largest rise (the peak picking skips the frames that cannot reach the threshold)
  std::vector<float>& row_on = g_scratch.row_on;
  std::vector<double>& row_fd = g_scratch.row_fd;
  row_on.resize((size_t)T), row_fd.assign((size_t)T, 0.0);
  bool on_nan = false;
  for (int64_t t = 0; t < T && !cand; ++t) {
    const float* o = onset + t * kF;
    float m[8];
    int nan = 0;
    for (int j = 0; j < 8; ++j) m[j] = o[j];
    for (int j = 0; j < 8; ++j) nan |= o[j] != o[j];
    for (int f = 8; f < kF; f += 8)
      for (int j = 0; j < 8; ++j) {
        m[j] = o[f + j] > m[j] ? o[f + j] : m[j];
        nan |= o[f + j] != o[f + j];
      }
    float r = m[0];
    for (int j = 1; j < 8; ++j) r = m[j] > r ? m[j] : r;
    row_on[(size_t)t] = r;
    on_nan |= nan != 0;
  }
  if (infer && !cand) {
    // np.max of the onset map: NaN if it holds one
    float max_on = row_on[0];
    for (int64_t t = 1; t < T; ++t) max_on = row_on[(size_t)t] > max_on ? row_on[(size_t)t] : max_on;
    max_on_d = on_nan ? std::nan("") : (double)max_on;
    int note_nan = 0;
    for (int64_t t = 2; t < T; ++t) {  // rows 0, 1 are zero, every entry is >= 0
      const float *n0 = note + t * kF, *n1 = n0 - kF, *n2 = n1 - kF;
      double m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int f = 0; f < kF; f += 8)
        for (int j = 0; j < 8; ++j) {
          const double d1 = (double)n0[f + j] - (double)n1[f + j], d2 = (double)n0[f + j] - (double)n2[f + j];
          const double d = d1 < d2 ? d1 : d2;
          m[j] = d > m[j] ? d : m[j];
          note_nan |= (d1 != d1) | (d2 != d2);
        }
      double r = m[0];
      for (int j = 1; j < 8; ++j) r = m[j] > r ? m[j] : r;
      row_fd[(size_t)t] = r;
      if (r > max_fd) max_fd = r;
    }
    // np.min / np.max propagate NaN: one NaN difference makes the scale, and with it every inferred onset, NaN
    if (note_nan) max_fd = std::nan("");
  }
  auto fd_at = [&](int64_t t, int f) -> double {
    if (t < 2) return 0.0;
    const double d1 = (double)note[t * kF + f] - (double)note[(t - 1) * kF + f];
    const double d2 = (double)note[t * kF + f] - (double)note[(t - 2) * kF + f];
    const double d = d1 < d2 ? d1 : d2;  // a NaN difference only matters through max_fd, which is NaN then
    return d < 0 ? 0.0 : d;
  };
  auto on_at = [&](int64_t t, int f) -> double {
    if (!infer) return (double)onset[t * kF + f];
    const double scaled = (max_on_d * fd_at(t, f)) / max_fd;  // 0/0 -> NaN when nothing rises, like numpy
    return np_maximum((double)onset[t * kF + f], scaled);
  };

  struct Raw {
    int32_t start, end, pitch;
    float amp;
  };
  std::vector<Raw> notes;

  // ---- peak picking (scipy.signal.argrelmax, axis 0) + threshold, visited backwards in time
  // `energy` = the note map with the cells of found notes zeroed: float32 like its source (every comparison below is
  // between values that convert to float64 exactly)
  std::vector<float>& energy = g_scratch.energy;
  energy.resize((size_t)T * kF);
  std::memcpy(energy.data(), note, (size_t)T * kF * sizeof(float));
  const int energy_tol = prm->energy_tol;
  const double frame_thresh = prm->frame_threshold, onset_thresh = prm->onset_threshold;
  // The reference keeps the onset value at the peaks and 0 elsewhere, then takes every cell >= onset_thresh
  // (note_creation.py:398-402): with a positive threshold those are peaks that reach it — onset >= thresh, or
  // max_on * fd / max_fd >= thresh, i.e. max_on * fd >= thresh * max_fd up to the rounding of the division (the margin
  // below is 10^7 ulps) — and frames whose maxima cannot are skipped; with a threshold <= 0 every cell that is not a peak
  // qualifies with its 0, first frame included.  Either way the exact value decides.
  const bool filter = onset_thresh > 0.0 && !on_nan && (!infer || max_fd > 0.0);
  const double lim = onset_thresh * max_fd * (1.0 - 1e-9);
  // a note from the onset peak (t, f): follow the energy forward (note_creation.py:404-446)
  auto track_from = [&](int64_t t, int f) {
    const int64_t start = t;
    if (start >= T - 1) return;
    int64_t i = start + 1;
    int k = 0;
    while (i < T - 1 && k < energy_tol) {
      if ((double)energy[i * kF + f] < frame_thresh)
        ++k;
      else
        k = 0;
      ++i;
    }
    i -= k;
    if (i - start <= prm->min_note_len) return;
    for (int64_t r = start; r < i; ++r) {
      energy[r * kF + f] = 0;
      if (f < kMaxFreqIdx) energy[r * kF + f + 1] = 0;
      if (f > 0) energy[r * kF + f - 1] = 0;
    }
    notes.push_back({(int32_t)start, (int32_t)i, f + kMidiOffset, mean_f32(note, start, i, f)});
  };
  if (cand) {
    // the device's peaks, visited like the loop below: backwards in time, downwards in frequency
    for (int64_t t = T - 2; t >= 1; --t) {
      const uint8_t* row = cand_bits + t * BP_NOTE_CAND_ROW_BYTES;
      uint64_t lo8;
      std::memcpy(&lo8, row, 8);
      const uint32_t hi3 = (uint32_t)row[8] | ((uint32_t)row[9] << 8) | ((uint32_t)row[10] << 16);
      if (!(lo8 | hi3)) continue;
      for (int f = kF - 1; f >= 0; --f)
        if (f < 64 ? (lo8 >> f) & 1 : (hi3 >> (f - 64)) & 1) track_from(t, f);
    }
  }
  for (int64_t t = cand ? -1 : T - 2; t >= 0; --t) {
    if (t == 0 && onset_thresh > 0.0) break;  // never a peak: its 0 is below the threshold
    const float *o0 = onset + t * kF, *n0 = note + t * kF, *n1 = t >= 1 ? n0 - kF : n0, *n2 = t >= 2 ? n1 - kF : n1;
    uint8_t flag[kF];
    int any = 0;
    if (filter && !((double)row_on[(size_t)t] >= onset_thresh) && !(infer && max_on_d * row_fd[(size_t)t] >= lim)) continue;
    if (filter && infer && t >= 2) {
      for (int f = 0; f < kF; ++f) {
        const double d1 = (double)n0[f] - (double)n1[f], d2 = (double)n0[f] - (double)n2[f];
        const double d = d1 < d2 ? d1 : d2;
        const int c = ((double)o0[f] >= onset_thresh) | (max_on_d * d >= lim);
        flag[f] = (uint8_t)c;
        any |= c;
      }
    } else if (filter) {
      for (int f = 0; f < kF; ++f) {
        const int c = (double)o0[f] >= onset_thresh;
        flag[f] = (uint8_t)c;
        any |= c;
      }
    } else {
      std::memset(flag, 1, sizeof flag);
      any = 1;
    }
    if (!any) continue;
    for (int f = kF - 1; f >= 0; --f) {
      if (!flag[f]) continue;
      double v = on_at(t, f);
      if (!(t >= 1 && v > on_at(t - 1, f) && v > on_at(t + 1, f))) v = 0.0;  // scipy.signal.argrelmax along time
      if (!(v >= onset_thresh)) continue;
      track_from(t, f);
    }
  }

  // ---- melodia trick (note_creation.py:449-509)
  if (prm->melodia_trick) {
    MaxTree tree(energy, frame_thresh, g_scratch.cand, g_scratch.node);
    while (!tree.has_nan()) {  // `while np.max(remaining_energy) > frame_thresh`: never true with a NaN in the map
      const int32_t am = tree.argmax();
      if (am < 0 || !((double)energy[am] > frame_thresh)) break;
      const int64_t i_mid = am / kF;
      const int f = am % kF;
      tree.set_zero(am);
      auto wipe = [&](int64_t r) {
        tree.set_zero(r * kF + f);
        if (f < kMaxFreqIdx) tree.set_zero(r * kF + f + 1);
        if (f > 0) tree.set_zero(r * kF + f - 1);
      };
      int64_t i = i_mid + 1;
      int k = 0;
      while (i < T - 1 && k < energy_tol) {
        if ((double)energy[i * kF + f] < frame_thresh)
          ++k;
        else
          k = 0;
        wipe(i);
        ++i;
      }
      const int64_t i_end = i - 1 - k;
      i = i_mid - 1;
      k = 0;
      while (i > 0 && k < energy_tol) {
        if ((double)energy[i * kF + f] < frame_thresh)
          ++k;
        else
          k = 0;
        wipe(i);
        --i;
      }
      const int64_t i_start = i + 1 + k;
      if (i_end - i_start <= prm->min_note_len) continue;
      notes.push_back({(int32_t)i_start, (int32_t)i_end, f + kMidiOffset, mean_f32(note, i_start, i_end, f)});
    }
  }

  // ---- pitch bends (note_creation.py:182-219) and frame times (346-357)
  int64_t total_bends = 0;
  if (prm->include_pitch_bends)
    for (const Raw& r : notes) total_bends += r.end - r.start;
  *n_events_out = (int64_t)notes.size();
  *n_bends_out = total_bends;
  if ((int64_t)notes.size() > max_events || total_bends > max_bends || (notes.size() && !events) ||
      (total_bends && !bends)) {
    g_notes_error = "bp_notes_decode: output buffers too small (required sizes returned)";
    return BP_ERR_INVALID_ARG;
  }
  double gauss[51];
  for (int i = 0; i < 51; ++i) {
    const double n = (double)i - 25.0;
    gauss[i] = std::exp(-(n * n) / (2.0 * 5.0 * 5.0));  // scipy.signal.windows.gaussian(51, std=5)
  }
  const double window_offset = (256.0 / 22050.0) * (172.0 - (43844.0 / 256.0)) + 0.0018;
  auto frame_time = [&](int64_t fr) {
    const double original = (double)(fr * 256) / 22050.0;
    const double window_number = std::floor((double)fr / 172.0);
    return original - (window_offset * window_number);
  };
  int64_t bo = 0;
  for (size_t e = 0; e < notes.size(); ++e) {
    const Raw& r = notes[e];
    bp_note_event& ev = events[e];
    ev.start_frame = r.start;
    ev.end_frame = r.end;
    ev.start_s = frame_time(r.start);
    ev.end_s = frame_time(r.end);
    ev.pitch_midi = r.pitch;
    ev.amplitude = r.amp;
    ev.bend_offset = bo;
    ev.n_bends = 0;
    if (!prm->include_pitch_bends) continue;
    if (cand) {  // get_pitch_bends evaluated on the device for every (frame, bin): note_device.hip nd_bend_kernel
      for (int64_t t = r.start; t < r.end; ++t) bends[bo++] = (int32_t)bend_map[t * kF + (r.pitch - kMidiOffset)];
      ev.n_bends = (int32_t)(r.end - r.start);
      continue;
    }
    const double pitch_hz = 440.0 * std::pow(2.0, ((double)r.pitch - 69.0) / 12.0);
    const int64_t freq_idx = round_half_even(12.0 * 3.0 * std::log2(pitch_hz / 27.5));
    const int64_t tol = 25;
    const int64_t f0 = freq_idx - tol > 0 ? freq_idx - tol : 0;
    const int64_t f1 = freq_idx + tol + 1 < kFC ? freq_idx + tol + 1 : kFC;
    const int64_t g0 = tol - freq_idx > 0 ? tol - freq_idx : 0;
    const int64_t pb_shift = tol - g0;
    for (int64_t t = r.start; t < r.end; ++t) {
      int64_t best = 0;
      double bestv = (double)contour[t * kFC + f0] * gauss[g0];
      for (int64_t j = 1; j < f1 - f0; ++j) {
        const double v = (double)contour[t * kFC + f0 + j] * gauss[g0 + j];
        if (v > bestv || (std::isnan(v) && !std::isnan(bestv))) {  // np.argmax: first maximum, NaN wins
          bestv = v;
          best = j;
        }
      }
      bends[bo++] = (int32_t)(best - pb_shift);
    }
    ev.n_bends = (int32_t)(r.end - r.start);
  }
  return BP_OK;
}

int bp_notes_decode(float* note, float* onset, const float* contour, int64_t n_frames,
                    const bp_note_params* prm, bp_note_event* events, int64_t max_events, int32_t* bends,
                    int64_t max_bends, int64_t* n_events_out, int64_t* n_bends_out) {
  return decode_core(note, onset, contour, nullptr, nullptr, n_frames, prm, events, max_events, bends, max_bends, n_events_out,
                     n_bends_out);
}

int bp_notes_decode_candidates(const float* note, const uint8_t* cand_bits, const int8_t* bend_map, int64_t n_frames,
                               const bp_note_params* prm, bp_note_event* events, int64_t max_events, int32_t* bends,
                               int64_t max_bends, int64_t* n_events_out, int64_t* n_bends_out) {
  if (!cand_bits) {
    g_notes_error = "bp_notes_decode_candidates: null candidate bitmap";
    return BP_ERR_INVALID_ARG;
  }
  // the note map is only read in this mode (frequency limits were applied where the candidates were made)
  return decode_core(const_cast<float*>(note), nullptr, nullptr, cand_bits, bend_map, n_frames, prm, events, max_events, bends,
                     max_bends, n_events_out, n_bends_out);
}

// The per-pitch windows of get_pitch_bends (note_creation.py:182-219) for note bins 0 .. 87, and its Gaussian: the tables
// the device kernel uses, computed by the expressions decode_core uses for a single note (so that both pick the same bins)
void bp_internal_bend_tables(int32_t* tab /* [88][4]: f0, n, g0, shift */, double* gauss /* [51] */) {
  for (int i = 0; i < 51; ++i) {
    const double n = (double)i - 25.0;
    gauss[i] = std::exp(-(n * n) / (2.0 * 5.0 * 5.0));
  }
  for (int b = 0; b < kF; ++b) {
    const int pitch = b + kMidiOffset;
    const double pitch_hz = 440.0 * std::pow(2.0, ((double)pitch - 69.0) / 12.0);
    const int64_t freq_idx = round_half_even(12.0 * 3.0 * std::log2(pitch_hz / 27.5));
    const int64_t tol = 25;
    const int64_t f0 = freq_idx - tol > 0 ? freq_idx - tol : 0;
    const int64_t f1 = freq_idx + tol + 1 < kFC ? freq_idx + tol + 1 : kFC;
    const int64_t g0 = tol - freq_idx > 0 ? tol - freq_idx : 0;
    tab[4 * b] = (int32_t)f0, tab[4 * b + 1] = (int32_t)(f1 - f0), tab[4 * b + 2] = (int32_t)g0, tab[4 * b + 3] = (int32_t)(tol - g0);
  }
}

// the [lo, hi) bins constrain_frequency keeps (note_creation.py:314-343), as decode_core computes them
void bp_internal_freq_limits(const bp_note_params* prm, int* lo_out, int* hi_out) {
  int64_t min_idx = 0, max_idx = kF;
  if (prm->min_freq_hz > 0.0)
    min_idx = round_half_even(12.0 * (std::log2(prm->min_freq_hz) - std::log2(440.0)) + 69.0 - kMidiOffset);
  if (prm->max_freq_hz > 0.0)
    max_idx = round_half_even(12.0 * (std::log2(prm->max_freq_hz) - std::log2(440.0)) + 69.0 - kMidiOffset);
  auto norm = [](int64_t i) { return i < 0 ? (i + kF < 0 ? 0 : i + kF) : (i > kF ? kF : i); };
  *lo_out = (int)norm(min_idx), *hi_out = (int)norm(max_idx);
}

}  // extern "C"
