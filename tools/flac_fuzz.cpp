// CPU sanitizer + mutation fuzz driver of the host-side FLAC decoder (csrc/flac.cpp, which stands in for torchaudio.load on
// /root/reference/dataset.py:31,104 and therefore parses UNTRUSTED files). Built by tests/test_flac_fuzz.py (and `make -C tools flac_fuzz`)
// with  g++ -fsanitize=address,undefined -fno-sanitize-recover=all  together with flac.cpp, twice: as shipped (CRCs enforced) and with
// -DWCA_FLAC_FUZZ_SKIP_CRC (mutations reach the subframe / residual decoders). Every call must RETURN a status code: a crash, a sanitizer
// report (non-recoverable: the process aborts) or a status outside {OK, INVALID, NOMEM} fails the run.
//   flac_fuzz <seed dir> <cases> <rng seed>
#include <dirent.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../include/wca.h"

static uint64_t g_s;
static inline uint64_t rnd() {  // splitmix64
  uint64_t z = (g_s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
static inline size_t rnd_below(size_t n) { return n ? (size_t)(rnd() % n) : 0; }

static std::vector<uint8_t> read_file(const std::string& p) {
  std::vector<uint8_t> b;
  FILE* f = fopen(p.c_str(), "rb");
  if (!f) return b;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  b.resize((size_t)n);
  if (n > 0 && fread(b.data(), 1, (size_t)n, f) != (size_t)n) b.clear();
  fclose(f);
  return b;
}

static const int64_t CAP = 1 << 18;   // samples per channel the output buffer holds (seeds are far shorter; a mutated header may claim more)
static std::vector<float> g_out;

// one decode through the C ABI, both with an output buffer and in counting mode; returns the status of the buffered call
static int run_one(const uint8_t* p, size_t n, long counts[4]) {
  int32_t sr = 0, ch = 0, bps = 0;
  int64_t total = 0, got = 0, got2 = 0;
  // the decoder is handed an exact-size heap copy so that any read past the end is an ASan report, not a read of the seed's slack
  std::vector<uint8_t> copy(p, p + n);
  const uint8_t* q = copy.empty() ? (const uint8_t*)"" : copy.data();
  const int ri = wca_flac_info(q, (int64_t)n, &sr, &ch, &bps, &total);
  const int rd = wca_flac_decode(q, (int64_t)n, g_out.data(), CAP, &got);
  const int rc = wca_flac_decode(q, (int64_t)n, nullptr, 0, &got2);
  for (int r : {ri, rd, rc})
    if (r != WCA_OK && r != WCA_ERR_INVALID && r != WCA_ERR_NOMEM) {
      fprintf(stderr, "unexpected status %d\n", r);
      exit(3);
    }
  if (rd == WCA_OK && (ri != WCA_OK || ch < 1 || ch > 8 || got < 0 || got > CAP)) {
    fprintf(stderr, "decode OK with info %d, channels %d, %lld samples\n", ri, ch, (long long)got);
    exit(4);
  }
  if (rd == WCA_OK && rc == WCA_OK && got != got2) {
    fprintf(stderr, "buffered and counting decode disagree: %lld vs %lld samples\n", (long long)got, (long long)got2);
    exit(5);
  }
  counts[rd == WCA_OK ? 0 : rd == WCA_ERR_INVALID ? 1 : 2]++;
  return rd;
}

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: flac_fuzz <seed dir> <cases> <rng seed>\n");
    return 2;
  }
  const long cases = atol(argv[2]);
  g_s = (uint64_t)atoll(argv[3]) * 0x2545F4914F6CDD1Dull + 1;
  g_out.assign((size_t)CAP * 8, 0.f);
  std::vector<std::vector<uint8_t>> seeds;
  if (DIR* d = opendir(argv[1])) {
    std::vector<std::string> names;
    while (dirent* e = readdir(d)) {
      std::string n = e->d_name;
      if (n.size() > 5 && n.substr(n.size() - 5) == ".flac") names.push_back(n);
    }
    closedir(d);
    std::sort(names.begin(), names.end());
    for (auto& n : names) seeds.push_back(read_file(std::string(argv[1]) + "/" + n));
  }
  if (seeds.empty()) {
    fprintf(stderr, "no seeds\n");
    return 2;
  }
  long counts[4] = {0, 0, 0, 0};
  // 1. every seed decodes
  for (auto& s : seeds)
    if (run_one(s.data(), s.size(), counts) != WCA_OK) {
      fprintf(stderr, "a seed stream does not decode\n");
      return 6;
    }
  const long seeds_ok = counts[0];
  // 2. every prefix of the first 96 bytes of every seed, and the empty stream
  long prefix[4] = {0, 0, 0, 0};
  for (auto& s : seeds)
    for (size_t n = 0; n <= 96 && n <= s.size(); ++n) run_one(s.data(), n, prefix);
  // 3. seeded mutations: 1-3 per case out of bit flips, byte overwrites, truncation, deletion, duplication, insertion, header tampering
  long mut[4] = {0, 0, 0, 0};
  for (long c = 0; c < cases; ++c) {
    std::vector<uint8_t> b = seeds[rnd_below(seeds.size())];
    const int nm = 1 + (int)rnd_below(3);
    for (int m = 0; m < nm && !b.empty(); ++m) {
      switch (rnd_below(8)) {
        case 0: {  // 1-8 bit flips anywhere
          const int k = 1 + (int)rnd_below(8);
          for (int i = 0; i < k; ++i) b[rnd_below(b.size())] ^= (uint8_t)(1u << rnd_below(8));
          break;
        }
        case 1: b[rnd_below(b.size())] = (uint8_t)rnd(); break;
        case 2: b.resize(rnd_below(b.size() + 1)); break;   // truncation (possibly to nothing)
        case 3: {  // delete a chunk
          const size_t a = rnd_below(b.size()), l = 1 + rnd_below(64);
          b.erase(b.begin() + a, b.begin() + (a + l < b.size() ? a + l : b.size()));
          break;
        }
        case 4: {  // duplicate a chunk
          const size_t a = rnd_below(b.size()), l = 1 + rnd_below(256);
          std::vector<uint8_t> chunk(b.begin() + a, b.begin() + (a + l < b.size() ? a + l : b.size()));
          b.insert(b.begin() + rnd_below(b.size()), chunk.begin(), chunk.end());
          break;
        }
        case 5: {  // insert random bytes
          const size_t l = 1 + rnd_below(16);
          std::vector<uint8_t> r(l);
          for (auto& x : r) x = (uint8_t)rnd();
          b.insert(b.begin() + rnd_below(b.size()), r.begin(), r.end());
          break;
        }
        case 6: b[rnd_below(b.size() < 64 ? b.size() : 64)] = (uint8_t)rnd(); break;   // STREAMINFO / first frame header
        case 7: {  // extreme values in a header-ish position: 0x00 / 0xff runs
          const size_t a = rnd_below(b.size()), l = 1 + rnd_below(8);
          for (size_t i = a; i < a + l && i < b.size(); ++i) b[i] = (rnd() & 1) ? 0xff : 0x00;
          break;
        }
      }
    }
    run_one(b.data(), b.size(), mut);
  }
  printf("flac_fuzz: seeds %zu decoded %ld | prefixes ok %ld invalid %ld nomem %ld | mutated cases %ld: ok %ld invalid %ld nomem %ld\n", seeds.size(), seeds_ok,
         prefix[0], prefix[1], prefix[2], cases, mut[0], mut[1], mut[2]);
  return 0;
}
