// Mutation fuzz of rt_image_decode - the call that picks the JPEG, PNG, GIF or BMP decoder by a file's first bytes -
// under AddressSanitizer / UBSan (host code only, no GPU):
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined tools/fuzz/image_fuzz.cpp raytracing_amd/csrc/rt_jpeg.cpp \
//     raytracing_amd/csrc/rt_png.cpp raytracing_amd/csrc/rt_gifbmp.cpp -o /tmp/image_fuzz
//   ASAN_OPTIONS=detect_leaks=0:allocator_may_return_null=1 /tmp/image_fuzz <iterations> <seed> <file> [more files]
// The blind mutations of jpeg_fuzz.cpp (truncate, overwrite anywhere or in the first kilobyte, plant 0xFF bytes,
// splice two files, cut pieces out): GIF and BMP carry no checksums, so these reach every part of their decoders;
// PNG has png_fuzz.cpp, which puts the chunk CRCs right again.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "../../include/rtmi.h"
namespace rt { int set_error(int code, const char*) { return code; } }
static std::vector<uint8_t> slurp(const char* p) {
  FILE* f = fopen(p, "rb"); if (!f) { perror(p); exit(2); }
  std::vector<uint8_t> v; uint8_t buf[65536]; size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
  fclose(f); return v;
}
int main(int argc, char** argv) {
  int iters = atoi(argv[1]);
  std::mt19937_64 rng(atoi(argv[2]));
  std::vector<std::vector<uint8_t>> bases;
  for (int i = 3; i < argc; ++i) bases.push_back(slurp(argv[i]));
  long ok = 0, refused = 0;
  std::vector<uint8_t> out;
  for (int it = 0; it < iters; ++it) {
    std::vector<uint8_t> d = bases[rng() % bases.size()];
    int mode = rng() % 6;
    if (mode == 0) { d.resize(rng() % (d.size() + 1)); }
    else if (mode == 1) { int k = 1 + rng() % 8; for (int j = 0; j < k; ++j) d[rng() % d.size()] = (uint8_t)rng(); }
    else if (mode == 2) { // mutate the header region (markers, tables, frame, scan headers)
      size_t lim = std::min<size_t>(d.size(), 1024); int k = 1 + rng() % 4; for (int j = 0; j < k; ++j) d[rng() % lim] = (uint8_t)rng(); }
    else if (mode == 3) { // insert a marker somewhere
      size_t p = rng() % d.size(); d[p] = 0xFF; if (p + 1 < d.size()) d[p + 1] = (uint8_t)(0xC0 + rng() % 0x40); }
    else if (mode == 4) { // splice two files
      const auto& o = bases[rng() % bases.size()]; size_t a = rng() % d.size(), b = rng() % o.size(); d.resize(a); d.insert(d.end(), o.begin() + b, o.end()); }
    else { size_t p = rng() % d.size(); size_t n = std::min<size_t>(d.size() - p, 1 + rng() % 64); d.erase(d.begin() + p, d.begin() + p + n); }
    // exact-size heap copy so that ASan sees any read past the end
    uint8_t* heap = (uint8_t*)malloc(d.size() ? d.size() : 1); memcpy(heap, d.data(), d.size());
    int32_t w = -1, h = -1;
    int rc = rt_image_decode(heap, d.size(), &w, &h, nullptr, 0);
    if (rc == 0 && w > 0 && h > 0 && (uint64_t)w * h <= (1u << 24)) {
      out.assign((size_t)w * h * 3, 0);
      uint8_t* oh = (uint8_t*)malloc(out.size());
      rc = rt_image_decode(heap, d.size(), &w, &h, oh, out.size());
      free(oh);
    }
    free(heap);
    if (rc == 0) ++ok; else ++refused;
  }
  printf("iters %d: decoded %ld, refused %ld\n", iters, ok, refused);
  return 0;
}
