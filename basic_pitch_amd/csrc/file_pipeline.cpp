// Whole files, natively (round 4): decode -> device (downmix, resampling, CQT + CNN) -> note events -> .mid / .csv on C++
// worker threads, no Python in the loop.  The batch job of the reference is a Python loop over files
//   basic_pitch/inference.py:509-604 predict_and_save: predict() per file, then pretty_midi write (586) / csv writer (409-428)
//   basic_pitch/note_creation.py:52-116 model_output_to_notes, 222-267 note_events_to_midi, 270-286 drop_overlapping_pitch_bends
// and this package's Python pipeline (predict_and_save_many) was bound by the interpreter: 78 three-minute files per
// second against 2,600 per second of device capacity.  Here a worker thread owns a file from its bytes to
// its outputs; the GPU is a shared resource the workers queue for (one lane per handle), everything else runs in parallel.
//
// The writers restate, byte for byte, what the Python side of this package writes (basic_pitch_amd/midi.py — itself
// pinned to pretty_midi + mido's layout by tests/golden/midi — and inference.save_note_events): tests/test_file_pipeline.py
// compares them on the reference-generated note fixtures without a GPU.
#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <cerrno>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/basic_pitch_amd.h"

namespace {

thread_local std::string g_file_error;

// ---------------------------------------------------------------------------------------------------------------------
// RIFF / WAVE (basic_pitch_amd/audio.py read_wav: PCM 8 / 16 / 24 / 32 and IEEE float 32 / 64, WAVE_FORMAT_EXTENSIBLE)
struct WavInfo {
  int tag = 0, channels = 0, sample_rate = 0, bits = 0;
  const uint8_t* pcm = nullptr;
  size_t pcm_bytes = 0;
  int64_t n_frames = 0;
};

uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

bool wav_parse(const uint8_t* d, size_t n, WavInfo& w) {
  if (n < 12 || std::memcmp(d, "RIFF", 4) != 0 || std::memcmp(d + 8, "WAVE", 4) != 0) {
    g_file_error = "not a RIFF/WAVE file";
    return false;
  }
  size_t pos = 12;
  bool have_fmt = false, have_data = false;
  while (pos + 8 <= n) {
    const uint32_t size = rd32(d + pos + 4);
    const uint8_t* body = d + pos + 8;
    const size_t avail = n - (pos + 8) < size ? n - (pos + 8) : size;  // a truncated last chunk: what is there
    if (std::memcmp(d + pos, "fmt ", 4) == 0 && avail >= 16) {
      w.tag = rd16(body), w.channels = rd16(body + 2), w.sample_rate = (int)rd32(body + 4), w.bits = rd16(body + 14);
      if (w.tag == 0xFFFE && avail >= 26) w.tag = rd16(body + 24);  // the real tag sits in the GUID
      have_fmt = true;
    } else if (std::memcmp(d + pos, "data", 4) == 0) {
      w.pcm = body, w.pcm_bytes = avail;
      have_data = true;
    }
    pos += 8 + (size_t)size + (size & 1);
  }
  if (!have_fmt || !have_data) {
    g_file_error = "missing fmt or data chunk";
    return false;
  }
  int width = 0;
  if (w.tag == 1 && (w.bits == 8 || w.bits == 16 || w.bits == 24 || w.bits == 32)) width = w.bits / 8;
  if (w.tag == 3 && (w.bits == 32 || w.bits == 64)) width = w.bits / 8;
  if (!width) {
    g_file_error = w.tag == 1 ? "unsupported PCM bit depth " + std::to_string(w.bits)
                              : "unsupported WAV format tag " + std::to_string(w.tag);
    return false;
  }
  if (w.channels < 1) {
    g_file_error = "zero channels";
    return false;
  }
  w.n_frames = (int64_t)(w.pcm_bytes / (size_t)width) / w.channels;
  return true;
}

// samples -> float32 exactly as read_wav does (power-of-two scales: multiplying by the reciprocal is exact)
void wav_to_float(const WavInfo& w, float* out) {
  const int64_t n = w.n_frames * w.channels;
  const uint8_t* p = w.pcm;
  if (w.tag == 1 && w.bits == 16) {
    for (int64_t i = 0; i < n; ++i) out[i] = (float)(int16_t)rd16(p + 2 * i) * (1.0f / 32768.0f);
  } else if (w.tag == 1 && w.bits == 8) {
    for (int64_t i = 0; i < n; ++i) out[i] = ((float)p[i] - 128.0f) * (1.0f / 128.0f);
  } else if (w.tag == 1 && w.bits == 24) {
    for (int64_t i = 0; i < n; ++i) {
      int32_t v = (int32_t)p[3 * i] | ((int32_t)p[3 * i + 1] << 8) | ((int32_t)p[3 * i + 2] << 16);
      if (v >= 1 << 23) v -= 1 << 24;
      out[i] = (float)v / 8388608.0f;
    }
  } else if (w.tag == 1 && w.bits == 32) {
    for (int64_t i = 0; i < n; ++i) out[i] = (float)((double)(int32_t)rd32(p + 4 * i) / 2147483648.0);
  } else if (w.bits == 32) {
    std::memcpy(out, p, (size_t)n * 4);
  } else {
    for (int64_t i = 0; i < n; ++i) {
      double v;
      std::memcpy(&v, p + 8 * i, 8);
      out[i] = (float)v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// repr(float) of CPython (float_repr_style "short"): the shortest digit string that round-trips, fixed notation while
// -4 < decimal point position <= 16, else d.ddde+XX with at least two exponent digits; ".0" behind integral values
void py_float_repr(double v, std::string& out) {
  if (std::isnan(v)) {
    out += "nan";
    return;
  }
  if (std::isinf(v)) {
    out += v < 0 ? "-inf" : "inf";
    return;
  }
  char buf[40];
  auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);  // shortest round-trip digits
  std::string s(buf, r.ptr);
  const bool neg = s[0] == '-';
  if (neg) s.erase(0, 1);
  const size_t epos = s.find('e');
  std::string digits = s.substr(0, epos);
  const int exp10 = std::stoi(s.substr(epos + 1));
  digits.erase(std::remove(digits.begin(), digits.end(), '.'), digits.end());
  const int decpt = exp10 + 1;  // value = 0.DIGITS x 10^decpt
  if (neg) out += '-';
  const int nd = (int)digits.size();
  if (decpt > -4 && decpt <= 16) {
    if (decpt <= 0) {
      out += "0.";
      out.append((size_t)-decpt, '0');
      out += digits;
    } else if (decpt >= nd) {
      out += digits;
      out.append((size_t)(decpt - nd), '0');
      out += ".0";
    } else {
      out.append(digits, 0, (size_t)decpt);
      out += '.';
      out.append(digits, (size_t)decpt, std::string::npos);
    }
  } else {
    out += digits[0];
    if (nd > 1) {
      out += '.';
      out.append(digits, 1, std::string::npos);
    }
    const int e = decpt - 1;
    out += 'e';
    out += e < 0 ? '-' : '+';
    const int ae = e < 0 ? -e : e;
    if (ae < 10) out += '0';
    out += std::to_string(ae);
  }
}

// velocity = int(np.round(127 * amplitude)) with amplitude a float32 scalar: float32 product, round half to even
int velocity_of(float amplitude) { return (int)std::nearbyintf(127.0f * amplitude); }

// inference.save_note_events: csv.writer rows start, end, pitch, velocity, *bends with "\r\n" line ends
void notes_csv(const bp_note_event* ev, int64_t n, const int32_t* bends, std::string& out) {
  out += "start_time_s,end_time_s,pitch_midi,velocity,pitch_bend\r\n";
  for (int64_t i = 0; i < n; ++i) {
    py_float_repr(ev[i].start_s, out);
    out += ',';
    py_float_repr(ev[i].end_s, out);
    out += ',';
    out += std::to_string(ev[i].pitch_midi);
    out += ',';
    out += std::to_string(velocity_of(ev[i].amplitude));
    for (int32_t k = 0; bends && k < ev[i].n_bends; ++k) {  // bends may be NULL (no pitch bends), as in notes_midi
      out += ',';
      out += std::to_string(bends[ev[i].bend_offset + k]);
    }
    out += "\r\n";
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// note events -> Standard MIDI File bytes: note_creation.note_events_to_midi + midi.PrettyMIDI.to_bytes
void put_vlq(std::vector<uint8_t>& d, int64_t v) {
  uint8_t tmp[5];
  int k = 0;
  tmp[k++] = (uint8_t)(v & 0x7F);
  v >>= 7;
  while (v) {
    tmp[k++] = (uint8_t)((v & 0x7F) | 0x80);
    v >>= 7;
  }
  while (k) d.push_back(tmp[--k]);
}
void put_chunk(std::vector<uint8_t>& out, const char* tag, const std::vector<uint8_t>& data) {
  out.insert(out.end(), tag, tag + 4);
  const uint32_t n = (uint32_t)data.size();
  out.push_back((uint8_t)(n >> 24)), out.push_back((uint8_t)(n >> 16)), out.push_back((uint8_t)(n >> 8)), out.push_back((uint8_t)n);
  out.insert(out.end(), data.begin(), data.end());
}

struct MidiNote {
  double start, end;
  int pitch, vel;
  int64_t first_bend, n_bends;  // into the flat tick / time arrays
};

bool notes_midi(const bp_note_event* ev, int64_t n, const int32_t* bends, bool multiple_pitch_bends, double tempo,
                std::vector<uint8_t>& out) {
  const int resolution = 220;
  const double tick_scale = 60.0 / (tempo * resolution);
  auto ticks_of = [&](double t) -> int64_t { return t > 0 ? (int64_t)std::nearbyint(t / tick_scale) : 0; };

  // drop_overlapping_pitch_bends: sorted(events) (tuple order: start, end, pitch, amplitude), bends dropped from any two
  // notes that overlap in time
  std::vector<int64_t> order((size_t)n);
  for (int64_t i = 0; i < n; ++i) order[(size_t)i] = i;
  std::vector<char> keep_bends((size_t)n, 1);
  if (!multiple_pitch_bends) {
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
      if (ev[a].start_s != ev[b].start_s) return ev[a].start_s < ev[b].start_s;
      if (ev[a].end_s != ev[b].end_s) return ev[a].end_s < ev[b].end_s;
      if (ev[a].pitch_midi != ev[b].pitch_midi) return ev[a].pitch_midi < ev[b].pitch_midi;
      return ev[a].amplitude < ev[b].amplitude;
    });
    for (int64_t i = 0; i + 1 < n; ++i)
      for (int64_t j = i + 1; j < n; ++j) {
        if (ev[order[(size_t)j]].start_s >= ev[order[(size_t)i]].end_s) break;
        keep_bends[(size_t)order[(size_t)i]] = keep_bends[(size_t)order[(size_t)j]] = 0;
      }
  }
  // pitch bends of all notes: ticks = round(b * 4096 / 3) clipped to [-8192, 8191], times = linspace(start, end, n)
  std::vector<int64_t> bend_tick;
  std::vector<double> bend_time;
  std::vector<MidiNote> notes((size_t)n);
  for (int64_t q = 0; q < n; ++q) {
    const bp_note_event& e = ev[order[(size_t)q]];
    MidiNote& m = notes[(size_t)q];
    m.start = e.start_s, m.end = e.end_s, m.pitch = e.pitch_midi, m.vel = velocity_of(e.amplitude);
    m.first_bend = (int64_t)bend_tick.size();
    m.n_bends = (bends && keep_bends[(size_t)order[(size_t)q]]) ? e.n_bends : 0;
    const double step = m.n_bends > 1 ? (e.end_s - e.start_s) / (double)(m.n_bends - 1) : 0.0;
    for (int64_t k = 0; k < m.n_bends; ++k) {
      int64_t t = (int64_t)std::nearbyint((double)((int64_t)bends[e.bend_offset + k] * 4096) / 3.0);
      t = t > 8191 ? 8191 : (t < -8192 ? -8192 : t);
      bend_tick.push_back(t);
      bend_time.push_back((m.n_bends > 1 && k == m.n_bends - 1) ? e.end_s : (double)k * step + e.start_s);
    }
  }
  // instruments: one, or (multiple_pitch_bends) one per note number in order of first appearance
  std::vector<std::vector<int64_t>> members;
  if (n > 0) {
    if (multiple_pitch_bends) {
      std::map<int, size_t> slot;
      for (int64_t q = 0; q < n; ++q) {
        auto it = slot.find(notes[(size_t)q].pitch);
        if (it == slot.end()) {
          slot[notes[(size_t)q].pitch] = members.size();
          members.emplace_back();
          it = slot.find(notes[(size_t)q].pitch);
        }
        members[it->second].push_back(q);
      }
    } else {
      members.emplace_back();
      for (int64_t q = 0; q < n; ++q) members[0].push_back(q);
    }
  }

  std::vector<std::vector<uint8_t>> tracks;
  {  // timing track: set_tempo, time_signature 4/4 at tick 0, end_of_track one tick later
    const int tempo_us = (int)(6e7 / (60.0 / (tick_scale * resolution)));
    std::vector<uint8_t> t = {0x00, 0xFF, 0x51, 0x03, (uint8_t)(tempo_us >> 16), (uint8_t)(tempo_us >> 8), (uint8_t)tempo_us,
                              0x00, 0xFF, 0x58, 0x04, 0x04, 0x02, 0x18, 0x08, 0x01, 0xFF, 0x2F, 0x00};
    tracks.push_back(std::move(t));
  }
  struct Ev {
    int64_t tick, key;
    int note, velo;  // -1: not a note event
    uint8_t status, d1;
    int d2;          // -1: one data byte
  };
  for (size_t inst = 0; inst < members.size(); ++inst) {
    static const int channels[15] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13, 14, 15};
    const int ch = channels[inst % 15];
    std::vector<Ev> e;
    e.push_back(Ev{0, (int64_t)6 << 16, -1, -1, (uint8_t)(0xC0 | ch), 4 /* Electric Piano 1 */, -1});
    for (int64_t q : members[inst]) {
      const MidiNote& m = notes[(size_t)q];
      e.push_back(Ev{ticks_of(m.start), ((int64_t)10 << 16) + m.pitch * 256 + m.vel, m.pitch, m.vel, (uint8_t)(0x90 | ch),
                     (uint8_t)m.pitch, m.vel});
      e.push_back(Ev{ticks_of(m.end), ((int64_t)10 << 16) + m.pitch * 256, m.pitch, 0, (uint8_t)(0x90 | ch), (uint8_t)m.pitch, 0});
    }
    for (int64_t q : members[inst]) {
      const MidiNote& m = notes[(size_t)q];
      for (int64_t k = 0; k < m.n_bends; ++k) {
        const int64_t b = bend_tick[(size_t)(m.first_bend + k)], v14 = b + 8192;
        e.push_back(Ev{ticks_of(bend_time[(size_t)(m.first_bend + k)]), ((int64_t)7 << 16) + b, -1, -1, (uint8_t)(0xE0 | ch),
                       (uint8_t)(v14 & 0x7F), (int)(v14 >> 7)});
      }
    }
    std::stable_sort(e.begin(), e.end(), [](const Ev& a, const Ev& b) { return a.tick != b.tick ? a.tick < b.tick : a.key < b.key; });
    // pretty_midi's fix-up: a note-on directly followed by the same pitch's note-off at the same tick is swapped (decided
    // on the sorted order as it was; the swaps cannot overlap)
    std::vector<size_t> sw;
    for (size_t k = 0; k + 1 < e.size(); ++k)
      if (e[k].tick == e[k + 1].tick && e[k].note >= 0 && e[k].note == e[k + 1].note && e[k].velo != 0 && e[k + 1].velo == 0)
        sw.push_back(k);
    for (size_t k : sw) std::swap(e[k], e[k + 1]);
    std::vector<uint8_t> d;
    d.reserve(e.size() * 4 + 8);
    int64_t last = 0;
    int running = -1;
    for (const Ev& x : e) {
      if (x.tick - last < 0 || x.tick - last >= ((int64_t)1 << 28)) {
        g_file_error = "MIDI delta time out of range";
        return false;
      }
      put_vlq(d, x.tick - last);
      last = x.tick;
      if (x.status != running) d.push_back(x.status);
      running = x.status;
      d.push_back(x.d1);
      if (x.d2 >= 0) d.push_back((uint8_t)x.d2);
    }
    d.push_back(0x01), d.push_back(0xFF), d.push_back(0x2F), d.push_back(0x00);  // end of track, one tick later
    tracks.push_back(std::move(d));
  }
  out.clear();
  std::vector<uint8_t> hd = {0x00, 0x01, (uint8_t)(tracks.size() >> 8), (uint8_t)tracks.size(), (uint8_t)(resolution >> 8),
                             (uint8_t)resolution};
  put_chunk(out, "MThd", hd);
  for (const auto& t : tracks) put_chunk(out, "MTrk", t);
  return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// page-locked, grow-only: what a worker hands to / receives from the device goes by DMA without the runtime's staging copy.
// Page-locking tens of megabytes costs milliseconds, so a worker's buffers go back to a process-wide pool when it ends and
// the next bp_transcribe_files call starts from them (at most kPoolMax buffers are kept; the rest are released).
struct PinnedPool {
  static constexpr size_t kPoolMax = 192;              // buffers
  static constexpr size_t kPoolBytes = (size_t)2 << 30;  // and bytes kept between calls (16 workers of 3-minute files: ~1 GB)
  std::mutex mu;
  std::vector<std::pair<void*, size_t>> free_list;
  size_t pooled_bytes = 0;
  // the smallest pooled buffer of at least n bytes, else a new one
  std::pair<void*, size_t> take(size_t n) {
    {
      std::lock_guard<std::mutex> lk(mu);
      size_t best = free_list.size();
      for (size_t i = 0; i < free_list.size(); ++i)
        if (free_list[i].second >= n && (best == free_list.size() || free_list[i].second < free_list[best].second)) best = i;
      if (best != free_list.size()) {
        const auto b = free_list[best];
        free_list.erase(free_list.begin() + (long)best);
        pooled_bytes -= b.second;
        return b;
      }
    }
    const size_t want = n + n / 8 + 4096;
    return {bp_host_alloc(want), want};
  }
  void give(void* p, size_t cap) {
    if (!p) return;
    {
      std::lock_guard<std::mutex> lk(mu);
      if (free_list.size() < kPoolMax && pooled_bytes + cap <= kPoolBytes) {
        free_list.emplace_back(p, cap);
        pooled_bytes += cap;
        return;
      }
    }
    bp_host_free(p);
  }
};
PinnedPool g_pinned;

struct Pinned {
  void* p = nullptr;
  size_t cap = 0;
  bool ensure(size_t n) {
    if (n <= cap) return true;
    g_pinned.give(p, cap);
    p = nullptr, cap = 0;
    const auto b = g_pinned.take(n);
    if (!b.first) {
      g_file_error = "page-locked host allocation of " + std::to_string(b.second) + " bytes failed";
      return false;
    }
    p = b.first, cap = b.second;
    return true;
  }
  ~Pinned() { g_pinned.give(p, cap); }
  Pinned() = default;
  Pinned(const Pinned&) = delete;
  Pinned& operator=(const Pinned&) = delete;
};

std::atomic<int64_t> g_direct_reads{0};  // files whose bytes came in by O_DIRECT since the library was loaded

// A file's bytes into a page-locked buffer.  `direct`: O_DIRECT — the storage device's DMA writes straight into the
// page-locked buffer the GPU's copy engine reads from; the page cache is bypassed, so a file crosses host DRAM twice (device
// write, PCIe read) instead of four times (page-cache fill, the copy out of it: read + write, PCIe read) and no core copies
// it (host DRAM bandwidth is the first resource an 8-GPU file job runs out of).  O_DIRECT wants the buffer, the
// offset and the length aligned to the logical block size: the pooled buffers are page-aligned, the length is rounded up
// to 4 KiB (the read stops at the end of the file).  A file system that refuses O_DIRECT (tmpfs, some overlays: EINVAL at
// open or at the first read) is read through the page cache as before.
template <class Ensure>  // ensure(bytes) -> page-aligned buffer of at least that many bytes, or null (g_file_error set)
bool read_file_into(const std::string& path, Ensure&& ensure, size_t& n, bool direct, bool* was_direct = nullptr) {
  n = 0;
  int fd = -1;
#ifdef O_DIRECT
  if (direct) fd = open(path.c_str(), O_RDONLY | O_CLOEXEC | O_DIRECT);
#endif
  bool is_direct = fd >= 0;
  if (fd < 0) fd = open(path.c_str(), O_RDONLY | O_CLOEXEC);
  if (fd < 0) {
    g_file_error = path + " is not a file path.";
    return false;
  }
  struct stat st;
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
    g_file_error = path + " is not a file path.";
    close(fd);
    return false;
  }
  const size_t size = (size_t)st.st_size;
  constexpr size_t kBlock = 4096;
  const size_t want = is_direct ? ((size + kBlock - 1) / kBlock) * kBlock : size;
  uint8_t* dst = static_cast<uint8_t*>(ensure(want ? want : 1));
  if (!dst) {
    close(fd);
    return false;
  }
  while (n < size) {
    // direct: n stays a multiple of the block size until the file's end (a short read ends the loop or the file)
    const ssize_t got = read(fd, dst + n, want - n);
    if (got < 0 && errno == EINTR) continue;
    if (got < 0 && is_direct && errno == EINVAL && n == 0) {  // the file system takes the flag at open and refuses the read
      close(fd);
      fd = open(path.c_str(), O_RDONLY | O_CLOEXEC);
      is_direct = false;
      if (fd < 0) {
        g_file_error = path + " is not a file path.";
        return false;
      }
      continue;
    }
    if (got < 0) {
      g_file_error = path + ": " + std::strerror(errno);
      close(fd);
      return false;
    }
    if (got == 0) break;  // shorter than fstat said: what is there
    n += (size_t)got;
    if (is_direct && (n % kBlock) != 0) break;  // the file's tail
  }
  if (n > size) n = size;
  close(fd);
  if (is_direct) g_direct_reads.fetch_add(1, std::memory_order_relaxed);
  if (was_direct) *was_direct = is_direct;
  return true;
}

bool read_file_pinned(const std::string& path, Pinned& buf, size_t& n, bool direct = false) {
  return read_file_into(path, [&](size_t bytes) -> void* { return buf.ensure(bytes) ? buf.p : nullptr; }, n, direct);
}

int wav_pcm_format(const WavInfo& w) {
  if (w.tag == 1) return w.bits == 8 ? BP_PCM_U8 : w.bits == 16 ? BP_PCM_S16 : w.bits == 24 ? BP_PCM_S24 : BP_PCM_S32;
  return w.bits == 32 ? BP_PCM_F32 : BP_PCM_F64;
}

// inference.py:401-404: never overwrite.  O_EXCL makes the existence check and the creation one step (two processes
// writing into the same directory cannot both win), and a short or failed write removes the partial file, so that a retry
// does not find a corpse and report "already exists".
bool write_new_file(const std::string& path, const void* data, size_t n) {
  const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_CLOEXEC, 0666);
  if (fd < 0) {
    g_file_error = errno == EEXIST ? path + " already exists and would be overwritten." : "cannot write " + path + ": " + std::strerror(errno);
    return false;
  }
  const uint8_t* p = static_cast<const uint8_t*>(data);
  size_t done = 0;
  while (done < n) {
    const ssize_t w = write(fd, p + done, n - done);
    if (w < 0 && errno == EINTR) continue;
    if (w <= 0) {
      g_file_error = "cannot write " + path + ": " + std::strerror(errno);
      close(fd);
      unlink(path.c_str());
      return false;
    }
    done += (size_t)w;
  }
  if (close(fd) != 0) {
    g_file_error = "cannot write " + path + ": " + std::strerror(errno);
    unlink(path.c_str());
    return false;
  }
  return true;
}

std::string stem_of(const std::string& path) {
  const size_t slash = path.find_last_of('/');
  std::string base = slash == std::string::npos ? path : path.substr(slash + 1);
  const size_t dot = base.find_last_of('.');
  if (dot != std::string::npos && dot != 0) base.erase(dot);  // os.path.splitext
  return base;
}

// one worker per core this process may really use: a cgroup CPU quota (containers: 16 of a host's 256 hardware threads)
// counts, not the host's thread count — oversubscribed workers measured 25 % slower (the quota throttles every thread of
// the group, the ones feeding the GPU included)
int default_threads() {
  int n = (int)std::thread::hardware_concurrency();
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char quota[32];
    long period = 0;
    if (std::fscanf(f, "%31s %ld", quota, &period) == 2 && std::strcmp(quota, "max") != 0 && period > 0) {
      const long q = std::atol(quota) / period;
      if (q >= 1 && q < n) n = (int)q;
    }
    std::fclose(f);
  }
  return n < 1 ? 1 : n;
}

void set_report(bp_file_report* r, int status, const std::string& msg) {
  r->status = status;
  std::snprintf(r->message, sizeof r->message, "%s", msg.c_str());
}

}  // namespace

extern "C" {

const char* bp_files_last_error(void) { return g_file_error.c_str(); }

int64_t bp_files_direct_reads(void) { return g_direct_reads.load(std::memory_order_relaxed); }

// The pipeline's file reader on its own, into ordinary page-aligned host memory (no device needed): the file's length, a
// 64-bit FNV-1a of its bytes and whether O_DIRECT was really used — what the CPU tests compare with Python's read.
int64_t bp_files_read_probe(const char* path, int direct_io, uint64_t* fnv1a, int* used_direct) {
  if (!path) {
    g_file_error = "bp_files_read_probe: null path";
    return BP_ERR_INVALID_ARG;
  }
  void* mem = nullptr;
  size_t n = 0;
  bool was = false;
  const bool ok = read_file_into(std::string(path), [&](size_t bytes) -> void* {
    if (posix_memalign(&mem, 4096, bytes) != 0) {
      mem = nullptr;
      g_file_error = "bp_files_read_probe: out of memory";
    }
    return mem;
  }, n, direct_io != 0, &was);
  if (!ok) {
    std::free(mem);
    return BP_ERR_BAD_AUDIO;
  }
  uint64_t h = 1469598103934665603ull;
  const uint8_t* p = static_cast<const uint8_t*>(mem);
  for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 1099511628211ull;
  std::free(mem);
  if (fnv1a) *fnv1a = h;
  if (used_direct) *used_direct = was ? 1 : 0;
  return (int64_t)n;
}

void bp_files_release_buffers(void) {
  std::vector<std::pair<void*, size_t>> all;
  {
    std::lock_guard<std::mutex> lk(g_pinned.mu);
    all.swap(g_pinned.free_list);
    g_pinned.pooled_bytes = 0;
  }
  for (auto& b : all) bp_host_free(b.first);
}

int bp_wav_info(const void* file, size_t nbytes, int* channels, int* sample_rate, int* bits_per_sample, int64_t* n_frames) {
  WavInfo w;
  if (!file || !wav_parse(static_cast<const uint8_t*>(file), nbytes, w)) {
    if (!file) g_file_error = "bp_wav_info: null pointer";
    return file ? BP_ERR_BAD_AUDIO : BP_ERR_INVALID_ARG;
  }
  if (channels) *channels = w.channels;
  if (sample_rate) *sample_rate = w.sample_rate;
  if (bits_per_sample) *bits_per_sample = w.bits;
  if (n_frames) *n_frames = w.n_frames;
  return BP_OK;
}

int bp_wav_decode(const void* file, size_t nbytes, float* pcm, int64_t max_frames, int64_t* n_frames) {
  WavInfo w;
  if (!file || !pcm) {
    g_file_error = "bp_wav_decode: null pointer";
    return BP_ERR_INVALID_ARG;
  }
  if (!wav_parse(static_cast<const uint8_t*>(file), nbytes, w)) return BP_ERR_BAD_AUDIO;
  if (n_frames) *n_frames = w.n_frames;
  if (w.n_frames > max_frames) {
    g_file_error = "bp_wav_decode: buffer too small";
    return BP_ERR_INVALID_ARG;
  }
  wav_to_float(w, pcm);
  return BP_OK;
}

int64_t bp_notes_to_midi(const bp_note_event* events, int64_t n_events, const int32_t* bends, int multiple_pitch_bends,
                         double midi_tempo, uint8_t* out, int64_t capacity) {
  if (n_events < 0 || (n_events && !events) || !(midi_tempo > 0)) {
    g_file_error = "bp_notes_to_midi: bad arguments";
    return BP_ERR_INVALID_ARG;
  }
  std::vector<uint8_t> bytes;
  if (!notes_midi(events, n_events, bends, multiple_pitch_bends != 0, midi_tempo, bytes)) return BP_ERR_INVALID_ARG;
  if (out && (int64_t)bytes.size() <= capacity) std::memcpy(out, bytes.data(), bytes.size());
  return (int64_t)bytes.size();
}

int64_t bp_notes_to_csv(const bp_note_event* events, int64_t n_events, const int32_t* bends, char* out, int64_t capacity) {
  if (n_events < 0 || (n_events && !events)) {
    g_file_error = "bp_notes_to_csv: bad arguments";
    return BP_ERR_INVALID_ARG;
  }
  std::string s;
  notes_csv(events, n_events, bends, s);
  if (out && (int64_t)s.size() <= capacity) std::memcpy(out, s.data(), s.size());
  return (int64_t)s.size();
}

void bp_transcribe_params_default(bp_transcribe_params* p) {
  if (!p) return;
  std::memset(p, 0, sizeof *p);
  bp_note_params_default(&p->notes);
  p->midi_tempo = 120.0;
  p->save_midi = 1;
  p->save_notes = 1;
}

int bp_transcribe_files(bp_handle* handles, int n_handles, const char* const* paths, int64_t n_files, const char* out_dir,
                        const bp_transcribe_params* params, bp_file_report* reports) {
  if (!handles || n_handles < 1 || n_files < 0 || (n_files && !paths) || !out_dir || !params || (n_files && !reports)) {
    g_file_error = "bp_transcribe_files: null pointer or bad count";
    return BP_ERR_INVALID_ARG;
  }
  for (int i = 0; i < n_handles; ++i)
    if (!handles[i]) {
      g_file_error = "bp_transcribe_files: null handle";
      return BP_ERR_INVALID_ARG;
    }
  // every lane must be of the same mode: a file's output buffers are sized from handles[0] and the device call runs on
  // whichever lane is free (a default and an EXT_CQT_44K handle would disagree on rate and frame count)
  for (int i = 1; i < n_handles; ++i) {
    const int64_t probe = 1 << 20;
    if (bp_handle_sample_rate(handles[i]) != bp_handle_sample_rate(handles[0]) ||
        bp_handle_track_n_frames(handles[i], probe) != bp_handle_track_n_frames(handles[0], probe) ||
        bp_handle_resampled_length(handles[i], probe, 44100) != bp_handle_resampled_length(handles[0], probe, 44100)) {
      g_file_error = "bp_transcribe_files: the handles are of different modes (sample rate / frame geometry)";
      return BP_ERR_INVALID_ARG;
    }
  }
  struct stat st;
  if (stat(out_dir, &st) != 0 || !S_ISDIR(st.st_mode)) {
    g_file_error = std::string(out_dir) + " is not a directory.";
    return BP_ERR_INVALID_ARG;
  }
  const bp_transcribe_params prm = *params;
  // two inputs with the same stem would write the same files: the first in input order wins (inference.py:401-404)
  std::vector<char> dup((size_t)n_files, 0);
  {
    std::map<std::string, int64_t> seen;
    for (int64_t i = 0; i < n_files; ++i) {
      const std::string stem = stem_of(paths[i]);
      if (seen.count(stem)) {
        dup[(size_t)i] = 1;
        set_report(&reports[i], BP_ERR_INVALID_ARG,
                   std::string("the outputs of ") + paths[i] + " would overwrite those of " + paths[seen[stem]] + " (same file stem)");
        reports[i].n_note_events = 0, reports[i].n_frames = 0;
        reports[i].ms_read = reports[i].ms_lane_wait = reports[i].ms_device = reports[i].ms_notes = reports[i].ms_write = 0.0f;
      } else {
        seen[stem] = i;
      }
    }
  }
  // GPU lanes: a worker takes any free handle for the duration of one bp_infer_pcm
  std::mutex lane_mu;
  std::condition_variable lane_cv;
  std::vector<int> free_lanes;
  for (int i = 0; i < n_handles; ++i) free_lanes.push_back(i);
  auto acquire = [&]() {
    std::unique_lock<std::mutex> lk(lane_mu);
    lane_cv.wait(lk, [&] { return !free_lanes.empty(); });
    const int l = free_lanes.back();
    free_lanes.pop_back();
    return l;
  };
  auto release = [&](int l) {
    {
      std::lock_guard<std::mutex> lk(lane_mu);
      free_lanes.push_back(l);
    }
    lane_cv.notify_one();
  };

  std::atomic<int64_t> next{0};
  auto worker = [&]() {
    Pinned file, decoded, maps;  // the file's bytes; float PCM of a FLAC file; the three posteriorgrams
    std::vector<bp_note_event> events;
    std::vector<int32_t> bends;
    std::vector<uint8_t> midi;
    std::string csv;
    for (;;) {
      const int64_t i = next.fetch_add(1);
      if (i >= n_files) break;
      if (dup[(size_t)i]) continue;
      bp_file_report* rep = &reports[i];
      rep->n_note_events = 0, rep->n_frames = 0;
      rep->ms_read = rep->ms_lane_wait = rep->ms_device = rep->ms_notes = rep->ms_write = 0.0f;
      auto clock = std::chrono::steady_clock::now();
      auto lap = [&clock]() {  // milliseconds since the previous lap
        const auto t = std::chrono::steady_clock::now();
        const float ms = std::chrono::duration<float, std::milli>(t - clock).count();
        clock = t;
        return ms;
      };
      const std::string path = paths[i];
      const std::string base = std::string(out_dir) + "/" + stem_of(path) + "_basic_pitch.";
      // refuse before doing the work, like build_output_path
      if ((prm.save_midi && stat((base + "mid").c_str(), &st) == 0) || (prm.save_notes && stat((base + "csv").c_str(), &st) == 0)) {
        set_report(rep, BP_ERR_INVALID_ARG, base + "* already exists and would be overwritten. Skipping output files for " + path + ".");
        continue;
      }
      size_t n_bytes = 0;
      if (!read_file_pinned(path, file, n_bytes, prm.direct_io != 0)) {
        set_report(rep, BP_ERR_BAD_AUDIO, g_file_error);
        continue;
      }
      const uint8_t* fb = static_cast<const uint8_t*>(file.p);
      int channels = 0, sr = 0, format = BP_PCM_F32;
      int64_t n_frames = 0;
      const void* pcm = nullptr;
      bool flac_on_device = false;
      auto host_flac_decode = [&]() -> bool {  // the host decoder (flac_decode.cpp): float32 samples in `decoded`
        int bits = 0;
        if (bp_flac_info(fb, n_bytes, &channels, &sr, &bits, &n_frames) != BP_OK) {
          set_report(rep, BP_ERR_BAD_AUDIO, path + ": " + bp_audio_last_error());
          return false;
        }
        if (!decoded.ensure((size_t)(n_frames * channels) * sizeof(float) + 4)) {
          set_report(rep, BP_ERR_OUT_OF_MEMORY, path + ": " + g_file_error);
          return false;
        }
        int64_t got = 0;
        if (bp_flac_decode(fb, n_bytes, static_cast<float*>(decoded.p), n_frames, &got) != BP_OK || got != n_frames) {
          set_report(rep, BP_ERR_BAD_AUDIO, path + ": " + bp_audio_last_error());
          return false;
        }
        pcm = decoded.p;
        format = BP_PCM_F32;
        return true;
      };
      if (n_bytes >= 12 && std::memcmp(fb, "RIFF", 4) == 0 && std::memcmp(fb + 8, "WAVE", 4) == 0) {
        WavInfo w;
        if (!wav_parse(fb, n_bytes, w)) {
          set_report(rep, BP_ERR_BAD_AUDIO, path + ": " + g_file_error);
          continue;
        }
        // the samples go to the device as the file stores them (bp_infer_pcm_raw converts there)
        channels = w.channels, sr = w.sample_rate, n_frames = w.n_frames, pcm = w.pcm, format = wav_pcm_format(w);
      } else if (n_bytes >= 4 && (std::memcmp(fb, "fLaC", 4) == 0 || std::memcmp(fb, "ID3", 3) == 0)) {
        // FLAC: the file's BYTES go to the device and are decoded there (flac_device.hip) — no core-time per sample here,
        // half the PCIe bytes of the PCM — unless the stream is one the device decoder leaves to the host (no sample count
        // or block sizes in STREAMINFO, more than 24 bits / 8 channels) or params.host_flac asks for the host decoder
        bp_flac_stream_layout lay;
        if (!prm.host_flac && bp_flac_layout(fb, n_bytes, &lay) == BP_OK && lay.n_frames > 0 && lay.min_block >= 16 &&
            lay.max_block >= lay.min_block && lay.bits_per_sample <= 24 && lay.bits_per_sample >= 4 && lay.channels <= 8) {
          flac_on_device = true;
          channels = lay.channels, sr = lay.sample_rate, n_frames = lay.n_frames;
        } else if (!host_flac_decode()) {
          continue;
        }
      } else {
        set_report(rep, BP_ERR_BAD_AUDIO, path + ": not a WAV or FLAC file (the native pipeline reads RIFF/WAVE and FLAC)");
        continue;
      }

      const int64_t T = bp_handle_track_n_frames(handles[0], bp_handle_resampled_length(handles[0], n_frames, sr));
      if (T > 0 && !maps.ensure((size_t)T * (88 + 88 + 264) * sizeof(float))) {
        set_report(rep, BP_ERR_OUT_OF_MEMORY, path + ": " + g_file_error);
        continue;
      }
      float* note = static_cast<float*>(maps.p);
      float* onset = note + T * 88;
      float* contour = onset + T * 88;
      // device-side candidates (the default): the same page-locked buffer holds the note map, the onset-peak bitmap
      // (T x 12 bytes) and, from a 16-byte boundary, the pitch-bend map (T x 88 bytes) instead of the onset / contour maps
      uint8_t* cand_bits = reinterpret_cast<uint8_t*>(onset);
      int8_t* bend_map = reinterpret_cast<int8_t*>(cand_bits + ((T * BP_NOTE_CAND_ROW_BYTES + 15) & ~(int64_t)15));
      bool use_cand = !prm.host_decode && prm.notes.onset_threshold > 0.0;
      rep->ms_read = lap();
      int rc = BP_OK;
      std::string err;
      bool reported = false;
      for (int attempt = 0; attempt < 2; ++attempt) {
        const int lane = acquire();
        rep->ms_lane_wait += lap();
        bp_handle h = handles[lane];
        rc = BP_OK;
        if (T > 0) {
          if (use_cand) {
            int status = 0;
            rc = flac_on_device ? bp_infer_flac_candidates(h, fb, n_bytes, &prm.notes, note, cand_bits,
                                                           prm.notes.include_pitch_bends ? bend_map : nullptr, &status)
                                : bp_infer_pcm_raw_candidates(h, pcm, format, n_frames, channels, sr, &prm.notes, note, cand_bits,
                                                              prm.notes.include_pitch_bends ? bend_map : nullptr, &status);
            if (rc == BP_OK && status != 0) {  // a NaN in the maps: numpy's rules need the maps themselves — they are
              use_cand = false;                // still on the device, the lane is still ours
              rc = bp_track_maps(h, T, note, onset, contour, BP_MEM_HOST);
            }
          } else {
            rc = flac_on_device ? bp_infer_flac(h, fb, n_bytes, note, onset, contour, BP_MEM_HOST)
                                : bp_infer_pcm_raw(h, pcm, format, n_frames, channels, sr, note, onset, contour, BP_MEM_HOST);
          }
          if (rc != BP_OK) err = bp_last_error(h);
        }
        release(lane);
        rep->ms_device += lap();
        if (!(flac_on_device && (rc == BP_ERR_BAD_AUDIO || rc == BP_ERR_UNSUPPORTED))) break;
        // the device decoder could not follow the stream: the host decoder either decodes it or names the fault
        flac_on_device = false;
        if (!host_flac_decode()) {
          reported = true;
          break;
        }
        rep->ms_read += lap();
      }
      if (reported) continue;
      if (rc != BP_OK) {
        set_report(rep, rc, path + ": " + err);
        continue;
      }
      rep->n_frames = T;

      int64_t n_ev = 0, n_b = 0;
      size_t cap_ev = (size_t)std::max<int64_t>(256, T / 4), cap_b = (size_t)std::max<int64_t>(4096, 4 * T);
      for (int attempt = 0; attempt < 2; ++attempt) {
        events.resize(cap_ev), bends.resize(cap_b);
        rc = T <= 0 ? BP_OK
             : use_cand ? bp_notes_decode_candidates(note, cand_bits, prm.notes.include_pitch_bends ? bend_map : nullptr, T,
                                                     &prm.notes, events.data(), (int64_t)cap_ev, bends.data(), (int64_t)cap_b,
                                                     &n_ev, &n_b)
                        : bp_notes_decode(note, onset, contour, T, &prm.notes, events.data(), (int64_t)cap_ev, bends.data(),
                                          (int64_t)cap_b, &n_ev, &n_b);
        if (rc == BP_OK || !((size_t)n_ev > cap_ev || (size_t)n_b > cap_b)) break;
        cap_ev = std::max(cap_ev, (size_t)n_ev), cap_b = std::max(cap_b, (size_t)n_b);
      }
      if (rc != BP_OK) {
        set_report(rep, rc, path + ": " + bp_notes_last_error());
        continue;
      }
      rep->n_note_events = (int32_t)n_ev;
      rep->ms_notes = lap();
      const int32_t* bp = prm.notes.include_pitch_bends ? bends.data() : nullptr;
      bool ok = true;
      if (prm.save_midi) {
        ok = notes_midi(events.data(), n_ev, bp, prm.multiple_pitch_bends != 0, prm.midi_tempo, midi) &&
             write_new_file(base + "mid", midi.data(), midi.size());
      }
      if (ok && prm.save_notes) {
        csv.clear();
        notes_csv(events.data(), n_ev, bp, csv);
        ok = write_new_file(base + "csv", csv.data(), csv.size());
      }
      if (!ok) {
        set_report(rep, BP_ERR_INVALID_ARG, g_file_error);
        continue;
      }
      rep->ms_write = lap();
      set_report(rep, BP_OK, "");
    }
  };
  int n_threads = prm.threads > 0 ? prm.threads : default_threads();
  n_threads = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
  if (n_threads > n_files) n_threads = (int)(n_files > 0 ? n_files : 1);
  std::vector<std::thread> pool;
  for (int t = 1; t < n_threads; ++t) pool.emplace_back(worker);
  worker();
  for (auto& t : pool) t.join();
  return BP_OK;
}

}  // extern "C"
