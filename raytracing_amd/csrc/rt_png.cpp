// rt_png.cpp — a PNG decoder behind the C ABI, and the call that tells the image formats apart
// (host code, no dependency).
//
// The reference's (image-texture url) hands its URL to javax.imageio.ImageIO/read (texture.clj:72), which
// picks the reader by the file's first bytes, not its name (that is why resources/map2.png, a JPEG, loads).
// rt_image_decode does the same for the formats this library reads (JPEG, PNG, and GIF and BMP from
// rt_gifbmp.cpp); rt_image_decode_png is the PNG side: ISO/IEC 15948 with the inflate of RFC 1950 / 1951 written out here. Every colour type and bit
// depth, Adam7 interlacing, all five row filters; CRC-32 of every chunk and the Adler-32 of the image data
// are checked. PNG is lossless, so there is no choice of reconstruction to match: an 8-bit RGB file gives
// exactly the bytes the reference's raster read (texture.clj:73-77, a TYPE_3BYTE_BGR raster reversed to
// R G B) gives. The other colour types are where the reference itself goes wrong — its read assumes three
// bytes per pixel, so a grey, palette or RGBA raster comes out scrambled and a 16-bit one throws — and
// here they become what the file shows: grey replicated to R = G = B (1, 2 and 4 bits scaled to 0..255),
// palette indices looked up, 16-bit samples rounded to 8, alpha (and tRNS) dropped without compositing.
// Ancillary chunks (gAMA, sRGB, iCCP, ...) are skipped, as ImageIO's raster ignores them.
#include <stdint.h>
#include <string.h>

#include <new>
#include <string>
#include <vector>

#include "../../include/rtmi.h"

namespace rt {
int fail(int code, const char* fmt, ...);  // rt_jpeg.cpp
}

namespace {

struct PErr {
  int code;
  std::string msg;
};
[[noreturn]] void bad(const char* what) { throw PErr{RT_ERR_ARG, std::string("PNG: ") + what}; }

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

struct CrcTable {  // ISO 3309, as the PNG specification's sample code defines it
  uint32_t t[256];
  CrcTable() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      t[i] = c;
    }
  }
};
uint32_t crc32(const uint8_t* p, size_t n) {
  static const CrcTable table;
  uint32_t c = 0xFFFFFFFFu;
  for (size_t i = 0; i < n; ++i) c = table.t[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

// ---- inflate (RFC 1951) over a zlib stream (RFC 1950) ----

struct Bits {
  const uint8_t* p;
  size_t n, pos = 0;
  uint64_t hold = 0;  // bits not yet used, the next one lowest; beyond the end of the data the buffer reads as zeros
  int have = 0;       // how many of them are real
  void refill() {
    while (have <= 56 && pos < n) {
      hold |= (uint64_t)p[pos++] << have;
      have += 8;
    }
  }
  uint32_t peek(int k) const { return (uint32_t)(hold & ((1ull << k) - 1ull)); }
  void drop(int k) {
    if (k > have) bad("the compressed image data ends early");
    hold >>= k;
    have -= k;
  }
  uint32_t take(int k) {  // k <= 16 bits, least significant first
    refill();
    const uint32_t v = peek(k);
    drop(k);
    return v;
  }
  void to_byte() {  // the rest of the current byte is dropped, whole bytes read ahead are handed back
    pos -= (size_t)(have >> 3);
    hold = 0;
    have = 0;
  }
};

constexpr int FAST = 10;

struct Code {  // a canonical Huffman code: how many codes of each length, symbols in code order
  uint16_t count[16];
  uint16_t symbol[288];
  // codes of up to FAST bits resolved by one lookup on the next FAST bits of the stream: symbol << 4 | length; 0 = a
  // longer code, or none
  uint16_t fast[1 << FAST];
  // false: more codes of some length than a prefix code has room for
  bool build(const uint8_t* lengths, int n) {
    memset(count, 0, sizeof count);
    for (int i = 0; i < n; ++i) count[lengths[i]]++;
    int left = 1;
    for (int l = 1; l < 16; ++l) {
      left <<= 1;
      left -= count[l];
      if (left < 0) return false;
    }
    uint16_t offs[16], next[16];
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
    for (int i = 0; i < n; ++i)
      if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
    // the codes themselves (RFC 1951, 3.2.2), bit-reversed because the stream delivers a code's first bit lowest
    memset(fast, 0, sizeof fast);
    uint32_t code = 0;
    for (int l = 1; l < 16; ++l) {
      code = (code + (l > 1 ? count[l - 1] : 0u)) << 1;
      next[l] = (uint16_t)code;
    }
    for (int i = 0; i < n; ++i) {
      const int l = lengths[i];
      if (l == 0 || l > FAST) continue;  // (longer codes are found bit by bit from the counts)
      uint32_t c = next[l]++, rev = 0;
      for (int k = 0; k < l; ++k) rev |= ((c >> k) & 1u) << (l - 1 - k);
      for (uint32_t hi = 0; hi < (1u << (FAST - l)); ++hi) fast[rev | (hi << l)] = (uint16_t)((i << 4) | l);
    }
    return true;
  }
  int decode(Bits& b) const {
    b.refill();
    const uint16_t e = fast[b.peek(FAST)];
    if (e) {
      b.drop(e & 15);
      return e >> 4;
    }
    int code = 0, first = 0, index = 0;
    for (int l = 1; l < 16; ++l) {
      code |= (int)b.peek(1);
      b.drop(1);
      const int c = count[l];
      if (code - c < first) return symbol[index + (code - first)];
      index += c;
      first += c;
      first <<= 1;
      code <<= 1;
    }
    bad("a Huffman code that its table does not hold");
  }
};

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

// Inflates the zlib stream z[0, n) into exactly `want` bytes: fewer is a truncated image, more a damaged one.
void inflate_exact(const uint8_t* z, size_t n, std::vector<uint8_t>& out, size_t want) {
  if (n < 6) bad("no compressed image data");
  if ((z[0] & 0x0F) != 8 || (z[0] >> 4) > 7 || (((unsigned)z[0] << 8) | z[1]) % 31 != 0) bad("the image data is not a zlib stream");
  if (z[1] & 0x20) bad("a zlib stream with a preset dictionary");
  Bits b{z + 2, n - 2};
  out.assign(want, 0);
  uint8_t* const o = out.data();
  size_t at = 0;  // bytes written
  Code lit, dist;
  for (bool last = false; !last;) {
    last = b.take(1) != 0;
    const uint32_t type = b.take(2);
    if (type == 0) {
      b.to_byte();
      if (b.n - b.pos < 4) bad("the compressed image data ends early");
      const uint32_t len = b.p[b.pos] | ((uint32_t)b.p[b.pos + 1] << 8), nlen = b.p[b.pos + 2] | ((uint32_t)b.p[b.pos + 3] << 8);
      b.pos += 4;
      if ((len ^ nlen) != 0xFFFFu) bad("a stored block whose length check fails");
      if (b.n - b.pos < len) bad("the compressed image data ends early");
      if (want - at < len) bad("more image data than the header announces");
      memcpy(o + at, b.p + b.pos, len);
      at += len;
      b.pos += len;
      continue;
    }
    if (type == 3) bad("a compressed block of the reserved type");
    uint8_t lengths[320];
    if (type == 1) {
      for (int i = 0; i < 144; ++i) lengths[i] = 8;
      for (int i = 144; i < 256; ++i) lengths[i] = 9;
      for (int i = 256; i < 280; ++i) lengths[i] = 7;
      for (int i = 280; i < 288; ++i) lengths[i] = 8;
      lit.build(lengths, 288);
      for (int i = 0; i < 30; ++i) lengths[i] = 5;
      dist.build(lengths, 30);
    } else {
      const int nlen = (int)b.take(5) + 257, ndist = (int)b.take(5) + 1, ncode = (int)b.take(4) + 4;
      if (nlen > 286 || ndist > 30) bad("a compressed block with too many codes");
      static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      memset(lengths, 0, 19);
      for (int i = 0; i < ncode; ++i) lengths[order[i]] = (uint8_t)b.take(3);
      Code cl;
      if (!cl.build(lengths, 19)) bad("code lengths that are not a prefix code");
      int i = 0;
      while (i < nlen + ndist) {
        const int s = cl.decode(b);
        if (s < 16) { lengths[i++] = (uint8_t)s; continue; }
        int prev = 0, rep;
        if (s == 16) {
          if (i == 0) bad("a repeat with no code length before it");
          prev = lengths[i - 1];
          rep = 3 + (int)b.take(2);
        } else if (s == 17) {
          rep = 3 + (int)b.take(3);
        } else {
          rep = 11 + (int)b.take(7);
        }
        if (i + rep > nlen + ndist) bad("more code lengths than the block announces");
        while (rep--) lengths[i++] = (uint8_t)prev;
      }
      if (lengths[256] == 0) bad("a compressed block without an end code");
      if (!lit.build(lengths, nlen)) bad("code lengths that are not a prefix code");
      if (!dist.build(lengths + nlen, ndist)) bad("code lengths that are not a prefix code");
    }
    for (;;) {
      const int s = lit.decode(b);
      if (s < 256) {
        if (at >= want) bad("more image data than the header announces");
        o[at++] = (uint8_t)s;
        continue;
      }
      if (s == 256) break;
      if (s > 285) bad("a length code that does not exist");
      const size_t len = LEN_BASE[s - 257] + b.take(LEN_EXTRA[s - 257]);
      const int ds = dist.decode(b);
      if (ds > 29) bad("a distance code that does not exist");
      const size_t d = DIST_BASE[ds] + b.take(DIST_EXTRA[ds]);
      if (d > at) bad("a distance that reaches before the start of the data");
      if (want - at < len) bad("more image data than the header announces");
      const uint8_t* from = o + at - d;
      for (size_t k = 0; k < len; ++k) o[at + k] = from[k];  // may overlap what it writes
      at += len;
    }
  }
  if (at != want) bad("less image data than the header announces");
  b.to_byte();
  if (b.n - b.pos < 4) bad("the compressed image data ends early");
  uint32_t s1 = 1, s2 = 0;  // Adler-32 of what came out
  for (size_t i = 0; i < out.size();) {
    const size_t stop = i + 5552 < out.size() ? i + 5552 : out.size();
    for (; i < stop; ++i) { s1 += out[i]; s2 += s1; }
    s1 %= 65521u;
    s2 %= 65521u;
  }
  if (be32(b.p + b.pos) != ((s2 << 16) | s1)) bad("the checksum of the image data does not match");
}

// ---- the image ----

struct Png {
  int32_t width = 0, height = 0;
  int depth = 0, ctype = 0, interlace = 0, channels = 0;
  std::vector<uint8_t> palette;  // R G B per entry
  std::vector<uint8_t> raw;      // every pass: rows of one filter byte + packed samples, unfiltered in place

  static size_t row_bytes(uint64_t w, int bits_per_pixel) { return (size_t)((w * (uint64_t)bits_per_pixel + 7) / 8); }

  struct Pass { int x0, y0, dx, dy; };
  static const Pass* passes(int interlace, int* n) {
    static const Pass whole[1] = {{0, 0, 1, 1}};
    static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    *n = interlace ? 7 : 1;
    return interlace ? adam7 : whole;
  }

  void parse(const uint8_t* p, size_t n) {
    static const uint8_t magic[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (n < 8 || memcmp(p, magic, 8) != 0) bad("not a PNG file");
    size_t pos = 8;
    bool have_header = false, have_palette = false, ended = false;
    std::vector<uint8_t> idat;
    while (!ended) {
      if (n - pos < 12) bad("the file ends inside a chunk");
      const uint32_t len = be32(p + pos);
      if (len > 0x7FFFFFFFu || n - pos - 12 < len) bad("the file ends inside a chunk");
      const uint8_t* type = p + pos + 4;
      const uint8_t* data = p + pos + 8;
      if (crc32(type, (size_t)len + 4) != be32(data + len)) bad("a chunk whose checksum does not match");
      pos += 12 + (size_t)len;
      auto named = [&](const char* t) { return memcmp(type, t, 4) == 0; };
      if (!have_header && !named("IHDR")) bad("the first chunk is not the header");
      if (named("IHDR")) {
        if (have_header || len != 13) bad("a damaged header");
        have_header = true;
        const uint32_t w = be32(data), h = be32(data + 4);
        depth = data[8];
        ctype = data[9];
        interlace = data[12];
        if (w == 0 || h == 0 || w > 0x7FFFFFFFu || h > 0x7FFFFFFFu) bad("an image without size");
        if ((uint64_t)w * h > (1ull << 28)) throw PErr{RT_ERR_UNSUPPORTED, "PNG: more than 2^28 pixels"};
        width = (int32_t)w;
        height = (int32_t)h;
        const bool d8_16 = depth == 8 || depth == 16, low = depth == 1 || depth == 2 || depth == 4;
        switch (ctype) {
          case 0: channels = 1; if (!d8_16 && !low) bad("a bit depth its colour type does not allow"); break;
          case 2: channels = 3; if (!d8_16) bad("a bit depth its colour type does not allow"); break;
          case 3: channels = 1; if (depth != 8 && !low) bad("a bit depth its colour type does not allow"); break;
          case 4: channels = 2; if (!d8_16) bad("a bit depth its colour type does not allow"); break;
          case 6: channels = 4; if (!d8_16) bad("a bit depth its colour type does not allow"); break;
          default: bad("a colour type that does not exist");
        }
        if (data[10] != 0 || data[11] != 0) bad("a compression or filter method that does not exist");
        if (interlace > 1) bad("an interlace method that does not exist");
      } else if (named("PLTE")) {
        if (have_palette || len == 0 || len % 3 != 0 || len > 768) bad("a damaged palette");
        have_palette = true;
        palette.assign(data, data + len);
      } else if (named("IDAT")) {
        idat.insert(idat.end(), data, data + len);
      } else if (named("IEND")) {
        ended = true;
      } else if (!(type[0] & 0x20)) {
        bad("a critical chunk this decoder does not know");
      }
    }
    if (ctype == 3 && !have_palette) bad("a palette image without its palette");
    const int bpp_bits = channels * depth;
    int np;
    const Pass* ps = passes(interlace, &np);
    uint64_t need = 0;
    for (int k = 0; k < np; ++k) {
      const uint64_t pw = ((uint64_t)width - ps[k].x0 + ps[k].dx - 1) / ps[k].dx, ph = ((uint64_t)height - ps[k].y0 + ps[k].dy - 1) / ps[k].dy;
      if ((uint64_t)width <= (uint64_t)ps[k].x0 || (uint64_t)height <= (uint64_t)ps[k].y0) continue;
      need += (1 + (uint64_t)row_bytes(pw, bpp_bits)) * ph;
    }
    // (deflate expands at most 1032 : 1, so a header that announces more than its data could hold is refused
    // before anything of that size is allocated)
    if (need > (uint64_t)idat.size() * 1032 + 1024) bad("less image data than the header announces");
    inflate_exact(idat.data(), idat.size(), raw, (size_t)need);
    unfilter();
  }

  void unfilter() {
    const int bpp_bits = channels * depth;
    const size_t bpp = bpp_bits >= 8 ? (size_t)bpp_bits / 8 : 1;
    int np;
    const Pass* ps = passes(interlace, &np);
    size_t at = 0;
    for (int k = 0; k < np; ++k) {
      if (width <= ps[k].x0 || height <= ps[k].y0) continue;
      const size_t pw = ((size_t)width - ps[k].x0 + ps[k].dx - 1) / ps[k].dx, ph = ((size_t)height - ps[k].y0 + ps[k].dy - 1) / ps[k].dy;
      const size_t rb = row_bytes(pw, bpp_bits);
      const uint8_t* prev = nullptr;
      for (size_t y = 0; y < ph; ++y) {
        const int f = raw[at];
        uint8_t* cur = raw.data() + at + 1;
        if (f > 4) bad("a row filter that does not exist");
        for (size_t i = 0; i < rb; ++i) {
          const int a = i >= bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0, c = (prev && i >= bpp) ? prev[i - bpp] : 0;
          int pred = 0;
          if (f == 1) pred = a;
          else if (f == 2) pred = b;
          else if (f == 3) pred = (a + b) >> 1;
          else if (f == 4) {
            const int pa = b > c ? b - c : c - b, pb = a > c ? a - c : c - a, q = a + b - 2 * c, pc = q < 0 ? -q : q;
            pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          }
          cur[i] = (uint8_t)(cur[i] + pred);
        }
        prev = cur;
        at += 1 + rb;
      }
    }
  }

  void to_rgb(uint8_t* rgb) const {
    const int bpp_bits = channels * depth;
    const size_t n_palette = palette.size() / 3;
    int np;
    const Pass* ps = passes(interlace, &np);
    size_t at = 0;
    for (int k = 0; k < np; ++k) {
      if (width <= ps[k].x0 || height <= ps[k].y0) continue;
      const size_t pw = ((size_t)width - ps[k].x0 + ps[k].dx - 1) / ps[k].dx, ph = ((size_t)height - ps[k].y0 + ps[k].dy - 1) / ps[k].dy;
      const size_t rb = row_bytes(pw, bpp_bits);
      for (size_t y = 0; y < ph; ++y) {
        const uint8_t* row = raw.data() + at + 1;
        at += 1 + rb;
        for (size_t x = 0; x < pw; ++x) {
          uint8_t* o = rgb + (((size_t)ps[k].y0 + y * ps[k].dy) * (size_t)width + (size_t)ps[k].x0 + x * ps[k].dx) * 3;
          uint32_t s[4] = {0, 0, 0, 0};  // the pixel's samples, as 8-bit values (palette: the index)
          if (depth == 16) {
            for (int c = 0; c < channels; ++c) {
              const uint32_t v = ((uint32_t)row[(x * channels + c) * 2] << 8) | row[(x * channels + c) * 2 + 1];
              s[c] = (v * 255u + 32767u) / 65535u;
            }
          } else if (depth == 8) {
            for (int c = 0; c < channels; ++c) s[c] = row[x * channels + c];
          } else {  // 1, 2 or 4 bits, one channel, leftmost pixel in the high bits
            const size_t bit = x * (size_t)depth;
            const uint32_t v = (row[bit >> 3] >> (8 - depth - (int)(bit & 7))) & ((1u << depth) - 1u);
            s[0] = ctype == 3 ? v : v * (255u / ((1u << depth) - 1u));
          }
          if (ctype == 3) {
            if (s[0] >= n_palette) bad("a pixel whose palette entry does not exist");
            o[0] = palette[s[0] * 3]; o[1] = palette[s[0] * 3 + 1]; o[2] = palette[s[0] * 3 + 2];
          } else if (channels <= 2) {
            o[0] = o[1] = o[2] = (uint8_t)s[0];
          } else {
            o[0] = (uint8_t)s[0]; o[1] = (uint8_t)s[1]; o[2] = (uint8_t)s[2];
          }
        }
      }
    }
  }
};

}  // namespace

extern "C" int rt_image_decode_png(const uint8_t* bytes, uint64_t n_bytes, int32_t* width, int32_t* height, uint8_t* rgb8,
                                   uint64_t capacity) {
  if (!bytes || !width || !height) return rt::fail(RT_ERR_ARG, "rt_image_decode_png: bytes, width and height must not be NULL");
  *width = *height = 0;
  try {
    Png d;
    d.parse(bytes, (size_t)n_bytes);
    *width = d.width;
    *height = d.height;
    if (!rgb8) return RT_OK;  // the size alone (the file is checked through to its end all the same)
    if (capacity < (uint64_t)d.width * (uint64_t)d.height * 3)
      return rt::fail(RT_ERR_ARG, "rt_image_decode_png: the image is %d x %d, %llu bytes, the buffer holds %llu", d.width, d.height,
                      (unsigned long long)d.width * d.height * 3, (unsigned long long)capacity);
    d.to_rgb(rgb8);
    return RT_OK;
  } catch (const PErr& e) {
    return rt::fail(e.code, "%s", e.msg.c_str());
  } catch (const std::bad_alloc&) {
    return rt::fail(RT_ERR_OOM, "out of host memory decoding a PNG file");
  } catch (...) {
    return rt::fail(RT_ERR_ARG, "PNG: decoding failed");
  }
}

extern "C" int rt_image_decode(const uint8_t* bytes, uint64_t n_bytes, int32_t* width, int32_t* height, uint8_t* rgb8,
                               uint64_t capacity) {
  if (!bytes || !width || !height) return rt::fail(RT_ERR_ARG, "rt_image_decode: bytes, width and height must not be NULL");
  if (n_bytes >= 2 && bytes[0] == 0xFF && bytes[1] == 0xD8) return rt_image_decode_jpeg(bytes, n_bytes, width, height, rgb8, capacity);
  if (n_bytes >= 4 && bytes[0] == 0x89 && bytes[1] == 'P' && bytes[2] == 'N' && bytes[3] == 'G')
    return rt_image_decode_png(bytes, n_bytes, width, height, rgb8, capacity);
  if (n_bytes >= 4 && memcmp(bytes, "GIF8", 4) == 0) return rt_image_decode_gif(bytes, n_bytes, width, height, rgb8, capacity);
  if (n_bytes >= 2 && bytes[0] == 'B' && bytes[1] == 'M') return rt_image_decode_bmp(bytes, n_bytes, width, height, rgb8, capacity);
  *width = *height = 0;
  return rt::fail(RT_ERR_UNSUPPORTED, "rt_image_decode: not a JPEG, PNG, GIF or BMP file by its first bytes (WBMP and TIFF, which ImageIO also reads, are not covered)");
}
