// rt_gifbmp.cpp — GIF and BMP decoders behind the C ABI (host code, no dependency): the other two picture
// formats javax.imageio.ImageIO/read, which the reference's (image-texture url) calls (texture.clj:72),
// reads out of the box besides JPEG and PNG.
//
// Of these only a 24-bit BMP reaches the reference as the three-bytes-per-pixel raster its read assumes
// (texture.clj:73-77), and for that file this decoder returns the same bytes. A GIF (always a palette
// raster) and the other BMP depths come out of the reference scrambled or throw; here they become the
// picture the file holds. GIF: the first image of the file, as ImageIO/read returns image 0 — its own
// width and height, not the logical screen's — through its local or the global colour table, interlaced
// or not, transparency ignored. BMP: 1-, 4- and 8-bit palette, 16-bit (5-5-5, or bit fields), 24- and
// 32-bit, bottom-up or top-down, uncompressed or bit fields; run-length coded files are refused.
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

struct IErr {
  int code;
  std::string msg;
};
[[noreturn]] void bad(const char* what) { throw IErr{RT_ERR_ARG, what}; }
[[noreturn]] void unsupported(const char* what) { throw IErr{RT_ERR_UNSUPPORTED, what}; }

struct In {  // bounds-checked little-endian reads
  const uint8_t* p;
  size_t n, pos = 0;
  const char* fmt;
  void need(size_t k) const {
    if (pos > n || n - pos < k) throw IErr{RT_ERR_ARG, std::string(fmt) + ": the file ends early"};
  }
  uint8_t u8() { need(1); return p[pos++]; }
  uint32_t u16() { need(2); const uint32_t v = p[pos] | ((uint32_t)p[pos + 1] << 8); pos += 2; return v; }
  uint32_t u32() { need(4); const uint32_t v = p[pos] | ((uint32_t)p[pos + 1] << 8) | ((uint32_t)p[pos + 2] << 16) | ((uint32_t)p[pos + 3] << 24); pos += 4; return v; }
  void skip(size_t k) { need(k); pos += k; }
};

constexpr uint64_t MAX_PIXELS = 1ull << 28;

// ---- GIF (87a / 89a) ----

struct Gif {
  int32_t width = 0, height = 0;
  std::vector<uint8_t> index;    // width * height palette indices, rows in display order
  uint8_t table[768] = {0};      // the colour table the image uses
  int table_size = 0;

  void parse(const uint8_t* p, size_t n) {
    In in{p, n, 0, "GIF"};
    in.need(6);
    if (memcmp(p, "GIF87a", 6) != 0 && memcmp(p, "GIF89a", 6) != 0) bad("GIF: not a GIF file");
    in.pos = 6;
    in.skip(4);  // logical screen size: ImageIO/read returns the first image at its own size
    const uint8_t flags = in.u8();
    in.skip(2);
    uint8_t global[768] = {0};
    int global_size = 0;
    if (flags & 0x80) {
      global_size = 2 << (flags & 7);
      in.need((size_t)global_size * 3);
      memcpy(global, p + in.pos, (size_t)global_size * 3);
      in.pos += (size_t)global_size * 3;
    }
    for (;;) {
      const uint8_t block = in.u8();
      if (block == 0x3B) bad("GIF: the file holds no image");
      if (block == 0x21) {  // an extension: a label, then sub-blocks up to an empty one
        in.skip(1);
        for (uint8_t k = in.u8(); k; k = in.u8()) in.skip(k);
        continue;
      }
      if (block != 0x2C) bad("GIF: a block of an unknown kind");
      in.skip(4);  // position on the logical screen
      const uint32_t w = in.u16(), h = in.u16();
      const uint8_t iflags = in.u8();
      if (w == 0 || h == 0) bad("GIF: an image without size");
      if ((uint64_t)w * h > MAX_PIXELS) unsupported("GIF: more than 2^28 pixels");
      width = (int32_t)w;
      height = (int32_t)h;
      if (iflags & 0x80) {
        table_size = 2 << (iflags & 7);
        in.need((size_t)table_size * 3);
        memcpy(table, p + in.pos, (size_t)table_size * 3);
        in.pos += (size_t)table_size * 3;
      } else {
        table_size = global_size;
        memcpy(table, global, sizeof table);
      }
      if (table_size == 0) bad("GIF: an image without a colour table");
      const int min_bits = in.u8();
      if (min_bits < 2 || min_bits > 8) bad("GIF: a code size that does not exist");  // (2 also serves two-colour images)
      std::vector<uint8_t> data;  // the sub-blocks strung together
      for (uint8_t k = in.u8(); k; k = in.u8()) {
        in.need(k);
        data.insert(data.end(), p + in.pos, p + in.pos + k);
        in.pos += k;
      }
      // (a code is at least three bits and stands for at most 4096 pixels: a picture announced larger than its data
      // could fill is refused before it is allocated)
      if ((uint64_t)w * h > ((uint64_t)data.size() + 1) * 8 / 3 * 4096) bad("GIF: less picture data than the image announces");
      std::vector<uint8_t> px((size_t)w * h, 0);
      lzw(data, min_bits, px);
      if (iflags & 0x40) {  // interlaced: every 8th row from 0, every 8th from 4, every 4th from 2, every 2nd from 1
        index.assign(px.size(), 0);
        static const int start[4] = {0, 4, 2, 1}, step[4] = {8, 8, 4, 2};
        size_t src = 0;
        for (int pass = 0; pass < 4; ++pass)
          for (uint32_t y = (uint32_t)start[pass]; y < h; y += (uint32_t)step[pass], ++src)
            memcpy(index.data() + (size_t)y * w, px.data() + src * w, w);
      } else {
        index.swap(px);
      }
      return;
    }
  }

  // Variable-width LZW as GIF packs it: codes from the low bits up, a clear code and an end code above the
  // colours, widths from min_bits + 1 to 12, no new strings once 4096 exist. Pixels an early end code leaves
  // out stay index 0; codes past the picture's last pixel are ignored (files that carry some exist).
  static void lzw(const std::vector<uint8_t>& data, int min_bits, std::vector<uint8_t>& out) {
    const int clear = 1 << min_bits, stop = clear + 1;
    std::vector<uint16_t> prefix(4096, 0), length(4096, 0);
    std::vector<uint8_t> suffix(4096, 0), first(4096, 0);
    for (int i = 0; i < clear; ++i) {
      suffix[(size_t)i] = first[(size_t)i] = (uint8_t)i;
      length[(size_t)i] = 1;
    }
    int width = min_bits + 1, next = stop + 1, prev = -1;
    uint32_t hold = 0;
    int have = 0;
    size_t pos = 0, at = 0;
    while (at < out.size()) {
      while (have < width) {
        if (pos >= data.size()) return;
        hold |= (uint32_t)data[pos++] << have;
        have += 8;
      }
      const int code = (int)(hold & ((1u << width) - 1u));
      hold >>= width;
      have -= width;
      if (code == clear) {
        width = min_bits + 1;
        next = stop + 1;
        prev = -1;
        continue;
      }
      if (code == stop) return;
      // a code is a colour, a string built so far, or - once something has been written - the very string this
      // step is about to build (the previous string plus its own first pixel)
      if (code > next || (prev < 0 && code >= clear)) bad("GIF: a code the table does not hold yet");
      if (prev >= 0 && next < 4096) {
        prefix[(size_t)next] = (uint16_t)prev;
        suffix[(size_t)next] = first[(size_t)(code == next ? prev : code)];
        first[(size_t)next] = first[(size_t)prev];
        length[(size_t)next] = (uint16_t)(length[(size_t)prev] + 1);
        ++next;
        if (next == (1 << width) && width < 12) ++width;
      } else if (code >= next) {
        bad("GIF: a code the table does not hold yet");
      }
      const size_t len = length[(size_t)code], room = out.size() - at;
      size_t k = len;
      for (int c = code; k > 0; c = prefix[(size_t)c]) {  // the string comes out of the table backwards
        --k;
        if (k < room) out[at + k] = suffix[(size_t)c];
      }
      at += len < room ? len : room;
      prev = code;
    }
  }

  void to_rgb(uint8_t* rgb) const {
    for (size_t i = 0; i < index.size(); ++i) {
      const bool known = index[i] < table_size;  // (an index past a short table is black: ImageIO pads the table so)
      const uint8_t* c = table + (size_t)(known ? index[i] : 0) * 3;
      rgb[i * 3] = known ? c[0] : 0;
      rgb[i * 3 + 1] = known ? c[1] : 0;
      rgb[i * 3 + 2] = known ? c[2] : 0;
    }
  }
};

// ---- BMP ----

struct Bmp {
  int32_t width = 0, height = 0;
  std::vector<uint8_t> rgb;  // decoded at once: height * width * 3, top row first

  static uint8_t scale(uint32_t v, uint32_t mask) {  // the field `mask` cuts out of v, widened to 8 bits
    if (!mask) return 0;
    int shift = 0, bits = 0;
    while (!((mask >> shift) & 1u)) ++shift;
    while (shift + bits < 32 && ((mask >> (shift + bits)) & 1u)) ++bits;
    const uint32_t x = (v & mask) >> shift;
    if (bits >= 8) return (uint8_t)(x >> (bits - 8));
    return (uint8_t)((x * 255u + ((1u << bits) - 1u) / 2u) / ((1u << bits) - 1u));
  }

  void parse(const uint8_t* p, size_t n) {
    In in{p, n, 0, "BMP"};
    in.need(2);
    if (p[0] != 'B' || p[1] != 'M') bad("BMP: not a BMP file");
    in.pos = 10;
    const uint32_t data_at = in.u32();
    const uint32_t hsize = in.u32();
    uint32_t w, planes, bpp, compression = 0, colours = 0;
    int64_t h;
    if (hsize == 12) {  // the OS/2 header: 16-bit sizes, three-byte palette entries
      w = in.u16(); h = (int64_t)in.u16(); planes = in.u16(); bpp = in.u16();
    } else if (hsize >= 40 && hsize <= 124) {
      w = in.u32(); h = (int64_t)(int32_t)in.u32(); planes = in.u16(); bpp = in.u16();
      compression = in.u32();
      in.skip(12);  // image size, resolution
      colours = in.u32();
      in.skip(4);
    } else {
      bad("BMP: a header of an unknown size");
    }
    (void)planes;
    const bool top_down = h < 0;
    if (h < 0) h = -h;
    if (w == 0 || h == 0 || w > 0x7FFFFFFFu || h > 0x7FFFFFFF) bad("BMP: an image without size");
    if ((uint64_t)w * (uint64_t)h > MAX_PIXELS) unsupported("BMP: more than 2^28 pixels");
    if (bpp != 1 && bpp != 4 && bpp != 8 && bpp != 16 && bpp != 24 && bpp != 32) bad("BMP: a bit depth that does not exist");
    if (compression == 1 || compression == 2) unsupported("BMP: run-length coded files are not read");
    if (compression != 0 && compression != 3) unsupported("BMP: a compression this decoder does not read");
    if (compression == 3 && bpp != 16 && bpp != 32) bad("BMP: bit fields at a depth that has none");
    uint32_t mask[3] = {0, 0, 0};
    if (bpp == 16) { mask[0] = 0x7C00; mask[1] = 0x03E0; mask[2] = 0x001F; }
    if (bpp == 32) { mask[0] = 0x00FF0000; mask[1] = 0x0000FF00; mask[2] = 0x000000FF; }
    if (compression == 3) {
      // the masks follow a 40-byte header, or are fields of a longer one: the same place in the file either way
      in.pos = 14 + 40;
      for (int k = 0; k < 3; ++k) mask[k] = in.u32();
      if (!mask[0] || !mask[1] || !mask[2]) bad("BMP: an empty bit field");
    }
    std::vector<uint8_t> palette;
    if (bpp <= 8) {
      const uint32_t entry = hsize == 12 ? 3 : 4;
      uint32_t count = colours ? colours : (1u << bpp);
      if (count > 256) bad("BMP: a palette of more than 256 colours");
      in.pos = 14 + (size_t)hsize;
      if (in.pos > n) bad("BMP: the file ends early");
      if (hsize == 12 && data_at > in.pos && (data_at - in.pos) / 3 < count) count = (uint32_t)((data_at - in.pos) / 3);
      in.need((size_t)count * entry);
      palette.assign(768, 0);
      for (uint32_t k = 0; k < count; ++k) {  // stored B G R (and a pad byte)
        palette[k * 3] = p[in.pos + (size_t)k * entry + 2];
        palette[k * 3 + 1] = p[in.pos + (size_t)k * entry + 1];
        palette[k * 3 + 2] = p[in.pos + (size_t)k * entry];
      }
    }
    const size_t row = (((size_t)w * bpp + 31) / 32) * 4;  // rows are padded to four bytes
    if (data_at > n || (n - data_at) / row < (size_t)h) bad("BMP: the file ends early");
    width = (int32_t)w;
    height = (int32_t)h;
    rgb.assign((size_t)w * (size_t)h * 3, 0);
    for (size_t y = 0; y < (size_t)h; ++y) {
      const uint8_t* src = p + data_at + (top_down ? y : (size_t)h - 1 - y) * row;
      uint8_t* o = rgb.data() + y * (size_t)w * 3;
      for (size_t x = 0; x < w; ++x, o += 3) {
        if (bpp == 24) {
          o[0] = src[x * 3 + 2]; o[1] = src[x * 3 + 1]; o[2] = src[x * 3];
        } else if (bpp <= 8) {
          const size_t bit = x * bpp;
          const uint32_t v = (src[bit >> 3] >> (8 - bpp - (bit & 7))) & ((1u << bpp) - 1u);
          o[0] = palette[v * 3]; o[1] = palette[v * 3 + 1]; o[2] = palette[v * 3 + 2];
        } else {
          const uint32_t v = bpp == 16 ? (uint32_t)(src[x * 2] | (src[x * 2 + 1] << 8))
                                       : (uint32_t)src[x * 4] | ((uint32_t)src[x * 4 + 1] << 8) | ((uint32_t)src[x * 4 + 2] << 16) | ((uint32_t)src[x * 4 + 3] << 24);
          o[0] = scale(v, mask[0]); o[1] = scale(v, mask[1]); o[2] = scale(v, mask[2]);
        }
      }
    }
  }
};

template <class D, class F>
int decode_with(const char* name, const uint8_t* bytes, uint64_t n_bytes, int32_t* width, int32_t* height, uint8_t* rgb8, uint64_t capacity, F&& emit) {
  if (!bytes || !width || !height) return rt::fail(RT_ERR_ARG, "%s: bytes, width and height must not be NULL", name);
  *width = *height = 0;
  try {
    D d;
    d.parse(bytes, (size_t)n_bytes);
    *width = d.width;
    *height = d.height;
    if (!rgb8) return RT_OK;
    if (capacity < (uint64_t)d.width * (uint64_t)d.height * 3)
      return rt::fail(RT_ERR_ARG, "%s: the image is %d x %d, %llu bytes, the buffer holds %llu", name, d.width, d.height,
                      (unsigned long long)d.width * d.height * 3, (unsigned long long)capacity);
    emit(d, rgb8);
    return RT_OK;
  } catch (const IErr& e) {
    return rt::fail(e.code, "%s", e.msg.c_str());
  } catch (const std::bad_alloc&) {
    return rt::fail(RT_ERR_OOM, "%s: out of host memory", name);
  } catch (...) {
    return rt::fail(RT_ERR_ARG, "%s: decoding failed", name);
  }
}

}  // namespace

extern "C" int rt_image_decode_gif(const uint8_t* bytes, uint64_t n_bytes, int32_t* width, int32_t* height, uint8_t* rgb8,
                                   uint64_t capacity) {
  return decode_with<Gif>("rt_image_decode_gif", bytes, n_bytes, width, height, rgb8, capacity, [](const Gif& d, uint8_t* o) { d.to_rgb(o); });
}

extern "C" int rt_image_decode_bmp(const uint8_t* bytes, uint64_t n_bytes, int32_t* width, int32_t* height, uint8_t* rgb8,
                                   uint64_t capacity) {
  return decode_with<Bmp>("rt_image_decode_bmp", bytes, n_bytes, width, height, rgb8, capacity,
                          [](const Bmp& d, uint8_t* o) { memcpy(o, d.rgb.data(), d.rgb.size()); });
}
