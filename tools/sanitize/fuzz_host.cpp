#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <cstdint>
#include "../../include/basic_pitch_amd.h"
// stubs for the device entry points file_pipeline.cpp links against
extern "C" {
void* bp_host_alloc(size_t n) { return malloc(n); }
void bp_host_free(void* p) { free(p); }
int bp_infer_pcm_raw(bp_handle, const void*, int, int64_t, int, int, float*, float*, float*, int) { return -1; }
const char* bp_last_error(bp_handle) { return ""; }
int64_t bp_handle_track_n_frames(bp_handle, int64_t) { return 0; }
int64_t bp_handle_resampled_length(bp_handle, int64_t, int) { return 0; }
int bp_handle_sample_rate(bp_handle) { return 22050; }
int bp_infer_pcm_raw_candidates(bp_handle, const void*, int, int64_t, int, int, const bp_note_params*, float*, uint8_t*, int8_t*,
                                int*) { return -1; }
}
static uint64_t s = 88172645463325252ull;
static uint32_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); }
static float frnd() { return (float)(rnd() & 0xffffff) / 16777216.0f; }
int main(int argc, char** argv) {
  // ---- WAV parser
  FILE* f = fopen(argv[1], "rb");
  std::vector<uint8_t> base(30000);
  base.resize(fread(base.data(), 1, base.size(), f));
  fclose(f);
  long ok = 0, bad = 0;
  for (int it = 0; it < 6000; ++it) {
    std::vector<uint8_t> d = base;
    int mode = it % 4;
    if (mode == 0) { for (int k = 0; k < 1 + (int)(rnd() % 5); ++k) d[rnd() % 64] = (uint8_t)rnd(); }
    else if (mode == 1) { d.resize(rnd() % d.size()); }
    else if (mode == 2) { for (int k = 0; k < 1 + (int)(rnd() % 9); ++k) d[rnd() % d.size()] ^= (uint8_t)(1u << (rnd() % 8)); }
    else { size_t p = rnd() % 128; for (int k = 0; k < 1 + (int)(rnd() % 63) && p + k < d.size(); ++k) d[p + k] = (uint8_t)rnd(); }
    uint8_t* h = (uint8_t*)malloc(d.size() ? d.size() : 1);
    memcpy(h, d.data(), d.size());
    int c, r, b; int64_t nf;
    int rc = bp_wav_info(h, d.size(), &c, &r, &b, &nf);
    if (rc == 0 && nf > 0 && nf * c < 5000000) {
      std::vector<float> out((size_t)(nf * c));
      int64_t got = 0;
      rc = bp_wav_decode(h, d.size(), out.data(), nf, &got);
    }
    free(h);
    (rc == 0 ? ok : bad)++;
  }
  printf("wav ok %ld bad %ld\n", ok, bad); fflush(stdout);
  // ---- note decoder + writers on random maps
  long ev_total = 0, cand_total = 0;
  for (int it = 0; it < 300; ++it) {
    const int64_t T = 1 + rnd() % 300;
    std::vector<float> note((size_t)T * 88), onset((size_t)T * 88), contour((size_t)T * 264);
    for (auto& v : note) v = powf(frnd(), 3.f);
    for (auto& v : onset) v = powf(frnd(), 6.f);
    for (auto& v : contour) v = frnd();
    for (int k = 0; k < 10; ++k) { int ff = rnd() % 88; int64_t a = rnd() % T; int64_t n = 5 + rnd() % 60; for (int64_t t = a; t < a + n && t < T; ++t) note[t * 88 + ff] = 0.4f + 0.5f * frnd(); onset[a * 88 + ff] = 0.5f + 0.5f * frnd(); }
    if (it % 7 == 0) note[(rnd() % T) * 88 + rnd() % 88] = NAN;
    if (it % 11 == 0) onset[(rnd() % T) * 88 + rnd() % 88] = NAN;
    bp_note_params prm; bp_note_params_default(&prm);
    const double th[6] = {-0.1, 0.0, 0.2, 0.5, 0.9, 1.5};
    prm.onset_threshold = th[rnd() % 6]; prm.frame_threshold = th[1 + rnd() % 4];
    prm.infer_onsets = rnd() & 1; prm.melodia_trick = rnd() & 1; prm.min_note_len = (int)(rnd() % 12);
    if (it % 5 == 0) { prm.min_freq_hz = 60.0 + rnd() % 200; prm.max_freq_hz = 500.0 + rnd() % 3000; }
    std::vector<bp_note_event> ev(256); std::vector<int32_t> bends(4096);
    int64_t ne = 0, nb = 0;
    int rc = bp_notes_decode(note.data(), onset.data(), contour.data(), T, &prm, ev.data(), (int64_t)ev.size(), bends.data(), (int64_t)bends.size(), &ne, &nb);
    if (rc != 0) { ev.resize((size_t)ne + 1); bends.resize((size_t)nb + 1);
      // decode again from fresh copies is not needed for a memory check: sizes now fit
      rc = bp_notes_decode(note.data(), onset.data(), contour.data(), T, &prm, ev.data(), (int64_t)ev.size(), bends.data(), (int64_t)bends.size(), &ne, &nb); }
    if (rc == 0) {
      bool nan_amp = false; for (int64_t i = 0; i < ne; ++i) nan_amp |= std::isnan(ev[i].amplitude);
      if (!nan_amp && ne < 1500) {
        int64_t nm = bp_notes_to_midi(ev.data(), ne, bends.data(), it & 1, 120.0, nullptr, 0);
        if (nm > 0) { std::vector<uint8_t> m((size_t)nm); bp_notes_to_midi(ev.data(), ne, bends.data(), it & 1, 120.0, m.data(), nm); }
        int64_t nc = bp_notes_to_csv(ev.data(), ne, bends.data(), nullptr, 0);
        if (nc > 0) { std::vector<char> cs((size_t)nc); bp_notes_to_csv(ev.data(), ne, bends.data(), cs.data(), nc); }
      }
      ev_total += ne;
    }
    // the candidate-mode tracker (round 5) on an arbitrary peak bitmap and bend map: whatever the device could hand over
    if (prm.onset_threshold > 0.0) {
      std::vector<uint8_t> bits((size_t)T * BP_NOTE_CAND_ROW_BYTES);
      std::vector<int8_t> bend((size_t)T * 88);
      for (auto& b : bits) b = (rnd() % 6 == 0) ? (uint8_t)rnd() : 0;
      for (auto& b : bend) b = (int8_t)((int)(rnd() % 51) - 25);
      for (auto& v : note) if (std::isnan(v)) v = 0.5f;  // the device reports NaN maps back instead (status 1)
      std::vector<bp_note_event> ev2(256); std::vector<int32_t> bends2(4096);
      int64_t ne2 = 0, nb2 = 0;
      int rc2 = bp_notes_decode_candidates(note.data(), bits.data(), (it & 1) ? bend.data() : nullptr, T, &prm, ev2.data(),
                                           (int64_t)ev2.size(), bends2.data(), (int64_t)bends2.size(), &ne2, &nb2);
      if (rc2 != 0 && (ne2 > (int64_t)ev2.size() || nb2 > (int64_t)bends2.size())) {
        ev2.resize((size_t)ne2 + 1); bends2.resize((size_t)nb2 + 1);
        rc2 = bp_notes_decode_candidates(note.data(), bits.data(), (it & 1) ? bend.data() : nullptr, T, &prm, ev2.data(),
                                         (int64_t)ev2.size(), bends2.data(), (int64_t)bends2.size(), &ne2, &nb2);
      }
      if (rc2 == 0) cand_total += ne2;
    }
  }
  printf("note decode events %ld, candidate-mode events %ld\n", ev_total, cand_total);
  return 0;
}
