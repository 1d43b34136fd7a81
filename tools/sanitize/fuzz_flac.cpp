#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cstdint>
#include "../../include/basic_pitch_amd.h"
static uint64_t s = 88172645463325252ull;
static uint32_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); }
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb");
  std::vector<uint8_t> base;
  uint8_t buf[65536]; size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) base.insert(base.end(), buf, buf + n);
  fclose(f);
  long ok = 0, bad = 0;
  for (int it = 0; it < 20000; ++it) {
    std::vector<uint8_t> d = base;
    int mode = it % 4;
    if (mode == 0) { for (int k = 0; k < 1 + (int)(rnd() % 5); ++k) d[rnd() % (d.size() < 200 ? d.size() : 200)] = (uint8_t)rnd(); }
    else if (mode == 1) { d.resize(rnd() % d.size()); }
    else if (mode == 2) { for (int k = 0; k < 1 + (int)(rnd() % 9); ++k) d[rnd() % d.size()] ^= (uint8_t)(1u << (rnd() % 8)); }
    else { size_t p = rnd() % d.size(); for (int k = 0; k < 1 + (int)(rnd() % 63) && p + k < d.size(); ++k) d[p + k] = (uint8_t)rnd(); }
    // exact-size heap copy so that ASan sees any read past the end
    uint8_t* h = (uint8_t*)malloc(d.size() ? d.size() : 1);
    memcpy(h, d.data(), d.size());
    int c, r, b; int64_t nf;
    int rc = bp_flac_info(h, d.size(), &c, &r, &b, &nf);
    if (rc == 0 && nf > 0 && nf * c < 5000000) {
      std::vector<float> out((size_t)(nf * c));
      int64_t got = 0;
      rc = bp_flac_decode(h, d.size(), out.data(), nf, &got);
    }
    free(h);
    (rc == 0 ? ok : bad)++;
  }
  printf("ok %ld bad %ld\n", ok, bad);
  return 0;
}
