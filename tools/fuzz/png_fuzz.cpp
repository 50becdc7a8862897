// Mutation fuzz of rt_image_decode_png (and the sniffing rt_image_decode) under AddressSanitizer / UBSan (host code only, no GPU):
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined tools/fuzz/png_fuzz.cpp raytracing_amd/csrc/rt_png.cpp raytracing_amd/csrc/rt_jpeg.cpp raytracing_amd/csrc/rt_gifbmp.cpp -o /tmp/png_fuzz
//   ASAN_OPTIONS=detect_leaks=0:allocator_may_return_null=1 /tmp/png_fuzz <iterations> <seed> <file.png> [more files]
// A PNG's chunks carry CRCs, so a blind byte flip stops at the first gate. Half of the mutations therefore edit the
// BODY of a chunk (header, palette or image data: bytes overwritten, pieces cut out or doubled) and write the chunk's
// CRC anew; the others truncate, overwrite anywhere, splice files and cut pieces out. Every result is decoded from an
// exact-size heap copy, so that a read past the end of the file is reported where it happens.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../../include/rtmi.h"
namespace rt { int set_error(int code, const char*) { return code; } }
static std::vector<uint8_t> slurp(const char* p) {
  FILE* f = fopen(p, "rb"); if (!f) { perror(p); exit(2); }
  std::vector<uint8_t> v; uint8_t buf[65536]; size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
  fclose(f); return v;
}
static uint32_t crc(const uint8_t* p, size_t n) {
  uint32_t c = 0xFFFFFFFFu;
  for (size_t i = 0; i < n; ++i) { c ^= p[i]; for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; }
  return ~c;
}
static void put32(std::vector<uint8_t>& v, size_t at, uint32_t x) { v[at] = x >> 24; v[at + 1] = x >> 16; v[at + 2] = x >> 8; v[at + 3] = x; }
struct Chunk { size_t at; uint32_t len; };
static std::vector<Chunk> chunks(const std::vector<uint8_t>& d) {
  std::vector<Chunk> out; size_t pos = 8;
  while (pos + 12 <= d.size()) {
    uint32_t len = ((uint32_t)d[pos] << 24) | (d[pos + 1] << 16) | (d[pos + 2] << 8) | d[pos + 3];
    if (len > d.size() - pos - 12) break;
    out.push_back({pos, len}); pos += 12 + (size_t)len;
  }
  return out;
}
int main(int argc, char** argv) {
  int iters = atoi(argv[1]);
  std::mt19937_64 rng(atoi(argv[2]));
  std::vector<std::vector<uint8_t>> bases;
  for (int i = 3; i < argc; ++i) bases.push_back(slurp(argv[i]));
  long ok = 0, refused = 0;
  for (int it = 0; it < iters; ++it) {
    std::vector<uint8_t> d = bases[rng() % bases.size()];
    int mode = rng() % 8;
    if (mode == 0) { d.resize(rng() % (d.size() + 1)); }
    else if (mode == 1) { int k = 1 + rng() % 8; for (int j = 0; j < k; ++j) d[rng() % d.size()] = (uint8_t)rng(); }
    else if (mode == 2) { const auto& o = bases[rng() % bases.size()]; size_t a = rng() % d.size(), b = rng() % o.size(); d.resize(a); d.insert(d.end(), o.begin() + b, o.end()); }
    else if (mode == 3) { size_t p = rng() % d.size(); size_t n = std::min<size_t>(d.size() - p, 1 + rng() % 64); d.erase(d.begin() + p, d.begin() + p + n); }
    else {  // inside a chunk, CRC made right again
      auto cs = chunks(d);
      if (!cs.empty()) {
        Chunk c = cs[rng() % cs.size()];
        if (mode == 4 && !cs.empty()) c = cs[0];  // the header
        std::vector<uint8_t> body(d.begin() + c.at + 8, d.begin() + c.at + 8 + c.len);
        int what = rng() % 4;
        if (body.empty()) what = 3;
        if (what == 0) { int k = 1 + rng() % 4; for (int j = 0; j < k; ++j) body[rng() % body.size()] = (uint8_t)rng(); }
        else if (what == 1) { size_t p = rng() % body.size(); size_t n = std::min<size_t>(body.size() - p, 1 + rng() % 32); body.erase(body.begin() + p, body.begin() + p + n); }
        else if (what == 2) { size_t p = rng() % body.size(); body[p] ^= (uint8_t)(1u << (rng() % 8)); }
        else { size_t n = 1 + rng() % 40; for (size_t j = 0; j < n; ++j) body.push_back((uint8_t)rng()); }
        std::vector<uint8_t> nd(d.begin(), d.begin() + c.at);
        nd.resize(nd.size() + 4); put32(nd, nd.size() - 4, (uint32_t)body.size());
        nd.insert(nd.end(), d.begin() + c.at + 4, d.begin() + c.at + 8);
        nd.insert(nd.end(), body.begin(), body.end());
        uint32_t k = crc(nd.data() + c.at + 4, 4 + body.size());
        nd.resize(nd.size() + 4); put32(nd, nd.size() - 4, k);
        nd.insert(nd.end(), d.begin() + c.at + 12 + c.len, d.end());
        d.swap(nd);
      }
    }
    uint8_t* heap = (uint8_t*)malloc(d.size() ? d.size() : 1); memcpy(heap, d.data(), d.size());
    int32_t w = -1, h = -1;
    int rc = (it & 1) ? rt_image_decode_png(heap, d.size(), &w, &h, nullptr, 0) : rt_image_decode(heap, d.size(), &w, &h, nullptr, 0);
    if (rc == 0 && w > 0 && h > 0 && (uint64_t)w * h <= (1u << 24)) {
      size_t n = (size_t)w * h * 3;
      uint8_t* oh = (uint8_t*)malloc(n);
      rc = rt_image_decode_png(heap, d.size(), &w, &h, oh, n);
      free(oh);
    }
    free(heap);
    if (rc == 0) ++ok; else ++refused;
  }
  printf("iters %d: decoded %ld, refused %ld\n", iters, ok, refused);
  return 0;
}
