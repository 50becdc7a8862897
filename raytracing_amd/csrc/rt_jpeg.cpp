// rt_jpeg.cpp — a JPEG decoder behind the C ABI (host code, no dependency).
//
// The reference's (image-texture url) reads its file with javax.imageio (texture.clj:68-81), whose JPEG
// reader is the IJG decoder with its default settings. rt_image_decode_jpeg stands in for that call, so a
// caller hands the library the bytes of earthmap.jpg as they are. Covered: ITU-T T.81 Huffman-coded
// 8-bit JPEG, sequential (SOF0, SOF1) and progressive (SOF2, spectral selection and successive
// approximation), one or three components, any integral sampling factors, restart intervals.
// Refused with a message: arithmetic coding, lossless and hierarchical processes, 12-bit samples,
// four-component (CMYK / YCCK) files.
//
// Reconstruction follows the IJG decoder's defaults, step for step, so that the texels are the ones
// that decoder produces and not merely close to them: the "slow but accurate" integer inverse DCT
// (Loeffler, Ligtenberg and Moschytz, 13-bit constants, two passes with the rounding of each), chroma
// brought to full size with the triangle filter ("fancy upsampling": 3/4 of the nearer sample and
// 1/4 of the further one, rounding alternately up and down) for 2:1 horizontal and 2:1 x 2:1, by
// replication for other ratios, and YCbCr -> RGB in 16-bit fixed point with its rounding constants.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <string>
#include <vector>

#include "../../include/rtmi.h"

namespace rt {
int set_error(int code, const char* message);  // rt_api.hip: what rt_last_error() returns on this thread
int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  return set_error(code, buf);
}
}

namespace {

struct JErr {
  int code;
  std::string msg;
};

struct Component {
  int id = 0, h = 1, v = 1, tq = 0;
  int td = 0, ta = 0;          // Huffman tables of the current scan
  int blocks_w = 0, blocks_h = 0;  // in the coefficient array (whole MCUs)
  int width = 0, height = 0;       // samples that carry image (ceil of the subsampled size)
  int pred = 0;                    // DC predictor
  std::vector<int16_t> coef;       // blocks_w * blocks_h * 64, natural (row-major) order inside a block
  std::vector<uint8_t> plane;      // blocks_w * 8 by blocks_h * 8 samples after the inverse DCT
};

struct Huff {
  bool defined = false;
  uint8_t bits[17] = {0};
  uint8_t vals[256] = {0};
  // canonical decoding tables (T.81 F.2.2.3): per length the smallest code, its index, the largest code
  int32_t mincode[18], maxcode[18], valptr[18];
  // fast path: codes of up to 9 bits resolved by one lookup (value | length << 8), 0 = longer code
  uint16_t fast[512];
  // false: the code lengths do not describe a prefix code (more codes of some length than that length has left)
  bool build() {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
      valptr[l] = k;
      mincode[l] = code;
      code += bits[l];
      k += bits[l];
      if (code > (1 << l)) return false;
      maxcode[l] = bits[l] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    memset(fast, 0, sizeof fast);
    code = 0; k = 0;
    for (int l = 1; l <= 9; ++l) {
      for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
        const int first = code << (9 - l), n = 1 << (9 - l);
        for (int j = 0; j < n; ++j) fast[first + j] = (uint16_t)(vals[k] | (l << 8));
      }
      code <<= 1;
    }
    defined = true;
    return true;
  }
};

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Decoder {
  const uint8_t* p;
  size_t n, pos = 0;
  int width = 0, height = 0, ncomp = 0;
  bool progressive = false, have_frame = false;
  int hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;
  Component comp[3];
  uint16_t qt[4][64];
  bool qt_defined[4] = {false, false, false, false};
  Huff dc[4], ac[4];
  int restart_interval = 0;
  bool adobe = false;
  int adobe_transform = -1;
  // entropy-coded segment reader
  uint32_t bitbuf = 0;
  int bitcnt = 0;
  bool hit_marker = false;
  int eobrun = 0;

  [[noreturn]] void bad(const char* what) { throw JErr{RT_ERR_ARG, std::string("JPEG: ") + what}; }
  [[noreturn]] void unsupported(const char* what) { throw JErr{RT_ERR_UNSUPPORTED, std::string("JPEG: ") + what}; }

  int u8() {
    if (pos >= n) bad("file ends inside a marker segment");
    return p[pos++];
  }
  int u16() {
    const int a = u8();
    return (a << 8) | u8();
  }

  // ---- bits of an entropy-coded segment (T.81 F.2.2.5): a 0xFF data byte is followed by a stuffed 0x00;
  // 0xFF followed by anything else is a marker, which ends the segment (zero bits are fed from there on)
  void fill() {
    while (bitcnt <= 24) {
      int b = 0;
      if (!hit_marker && pos < n) {
        b = p[pos];
        if (b == 0xFF) {
          const int b2 = pos + 1 < n ? p[pos + 1] : 0xD9;
          if (b2 == 0) {
            pos += 2;
          } else {
            hit_marker = true;
            b = 0;
          }
        } else {
          ++pos;
        }
      } else {
        hit_marker = true;
      }
      bitbuf |= (uint32_t)b << (24 - bitcnt);
      bitcnt += 8;
    }
  }
  int getbits(int k) {
    if (k == 0) return 0;
    if (bitcnt < k) fill();
    const int v = (int)(bitbuf >> (32 - k));
    bitbuf <<= k;
    bitcnt -= k;
    return v;
  }
  int getbit() { return getbits(1); }
  int decode(const Huff& h) {
    if (bitcnt < 16) fill();
    const uint16_t f = h.fast[bitbuf >> 23];
    if (f) {
      const int l = f >> 8;
      bitbuf <<= l;
      bitcnt -= l;
      return f & 255;
    }
    int code = (int)(bitbuf >> 22), l = 10;  // the first ten bits
    for (; l <= 16; ++l) {
      if (code <= h.maxcode[l] && h.maxcode[l] >= 0) break;
      code = (int)(bitbuf >> (32 - (l + 1)));
    }
    if (l > 16) bad("a Huffman code that its table does not hold");
    bitbuf <<= l;
    bitcnt -= l;
    return h.vals[h.valptr[l] + code - h.mincode[l]];
  }
  // T.81 F.2.2.1 EXTEND: an s-bit magnitude whose top bit is 0 stands for a negative number
  static int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }
  int receive_extend(int s) { return s ? extend(getbits(s), s) : 0; }

  void reset_entropy() {
    bitbuf = 0;
    bitcnt = 0;
    hit_marker = false;
    eobrun = 0;
    for (int c = 0; c < ncomp; ++c) comp[c].pred = 0;
  }
  // at a restart boundary: the segment must be followed by RSTn
  void take_restart() {
    // discard what is left of the byte in hand, then expect the marker
    bitbuf = 0;
    bitcnt = 0;
    while (pos < n && p[pos] != 0xFF) ++pos;  // (a well-formed file has the marker right here)
    while (pos + 1 < n && p[pos] == 0xFF && p[pos + 1] == 0xFF) ++pos;
    if (pos + 1 < n && p[pos] == 0xFF && p[pos + 1] >= 0xD0 && p[pos + 1] <= 0xD7) pos += 2;
    reset_entropy();
  }

  // ---- marker segments
  void read_dqt() {
    int len = u16() - 2;
    while (len > 0) {
      const int pq_tq = u8();
      const int pq = pq_tq >> 4, tq = pq_tq & 15;
      if (tq > 3 || pq > 1) bad("quantisation table header");
      for (int i = 0; i < 64; ++i) qt[tq][kZigzag[i]] = (uint16_t)(pq ? u16() : u8());
      qt_defined[tq] = true;
      len -= 1 + 64 * (pq + 1);
    }
    if (len != 0) bad("quantisation table segment length");
  }
  void read_dht() {
    int len = u16() - 2;
    while (len > 0) {
      const int tc_th = u8();
      const int tc = tc_th >> 4, th = tc_th & 15;
      if (tc > 1 || th > 3) bad("Huffman table header");
      Huff& h = tc ? ac[th] : dc[th];
      int total = 0;
      for (int l = 1; l <= 16; ++l) {
        h.bits[l] = (uint8_t)u8();
        total += h.bits[l];
      }
      if (total > 256) bad("Huffman table with more than 256 codes");
      for (int i = 0; i < total; ++i) h.vals[i] = (uint8_t)u8();
      if (!h.build()) bad("Huffman table whose code lengths are not a prefix code");
      len -= 17 + total;
    }
    if (len != 0) bad("Huffman table segment length");
  }
  void read_sof(int marker) {
    if (have_frame) bad("more than one frame header");
    const int len = u16();
    const int precision = u8();
    height = u16();
    width = u16();
    ncomp = u8();
    if (precision != 8) unsupported("only 8-bit samples are supported");
    if (width <= 0 || height <= 0) unsupported("image size not in the frame header (DNL) or zero");
    if (ncomp == 4) unsupported("four-component (CMYK / YCCK) files are not supported");
    if (ncomp != 1 && ncomp != 3) bad("component count");
    if (len != 8 + 3 * ncomp) bad("frame header length");
    progressive = marker == 0xC2;
    for (int c = 0; c < ncomp; ++c) {
      comp[c].id = u8();
      const int hv = u8();
      comp[c].h = hv >> 4;
      comp[c].v = hv & 15;
      comp[c].tq = u8();
      if (comp[c].h < 1 || comp[c].h > 4 || comp[c].v < 1 || comp[c].v > 4 || comp[c].tq > 3) bad("component sampling factors");
      hmax = comp[c].h > hmax ? comp[c].h : hmax;
      vmax = comp[c].v > vmax ? comp[c].v : vmax;
    }
    if (ncomp == 1) {  // a single component is never interleaved: its blocks are the MCUs
      comp[0].h = comp[0].v = 1;
      hmax = vmax = 1;
    }
    for (int c = 0; c < ncomp; ++c)
      if (hmax % comp[c].h || vmax % comp[c].v) unsupported("sampling factors that do not divide the largest one");
    mcus_x = (width + 8 * hmax - 1) / (8 * hmax);
    mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
    if ((uint64_t)mcus_x * mcus_y * hmax * vmax * 64 * 3 > (1ull << 31)) unsupported("image too large");
    for (int c = 0; c < ncomp; ++c) {
      Component& k = comp[c];
      k.blocks_w = mcus_x * k.h;
      k.blocks_h = mcus_y * k.v;
      k.width = (width * k.h + hmax - 1) / hmax;
      k.height = (height * k.v + vmax - 1) / vmax;
      k.coef.assign((size_t)k.blocks_w * k.blocks_h * 64, 0);
    }
    have_frame = true;
  }

  // ---- one block of a sequential scan (T.81 F.2.2)
  void block_sequential(Component& k, int16_t* b) {
    const int t = decode(dc[k.td]);
    if (t > 11) bad("DC magnitude category");
    k.pred += receive_extend(t);
    b[0] = (int16_t)k.pred;
    for (int i = 1; i < 64;) {
      const int rs = decode(ac[k.ta]);
      const int r = rs >> 4, s = rs & 15;
      if (s == 0) {
        if (r != 15) break;  // end of block
        i += 16;
        continue;
      }
      i += r;
      if (i > 63) bad("AC run past the end of a block");
      b[kZigzag[i++]] = (int16_t)receive_extend(s);
    }
  }
  // ---- progressive scans (T.81 G.1.2)
  void block_dc_first(Component& k, int16_t* b, int al) {
    const int t = decode(dc[k.td]);
    if (t > 11) bad("DC magnitude category");
    k.pred += receive_extend(t);
    b[0] = (int16_t)(k.pred * (1 << al));
  }
  void block_dc_refine(int16_t* b, int al) {
    if (getbit()) b[0] = (int16_t)(b[0] | (1 << al));
  }
  void block_ac_first(Component& k, int16_t* b, int ss, int se, int al) {
    if (eobrun > 0) {
      --eobrun;
      return;
    }
    for (int i = ss; i <= se;) {
      const int rs = decode(ac[k.ta]);
      const int r = rs >> 4, s = rs & 15;
      if (s == 0) {
        if (r < 15) {  // EOBr: this block and 2^r - 1 + extra more end here
          eobrun = (1 << r) - 1;
          if (r) eobrun += getbits(r);
          break;
        }
        i += 16;
        continue;
      }
      i += r;
      if (i > se) bad("AC run past the end of a band");
      b[kZigzag[i++]] = (int16_t)(receive_extend(s) * (1 << al));
    }
  }
  void block_ac_refine(Component& k, int16_t* b, int ss, int se, int al) {
    const int p1 = 1 << al, m1 = -(1 << al);
    int i = ss;
    if (eobrun <= 0) {
      for (; i <= se;) {
        const int rs = decode(ac[k.ta]);
        int r = rs >> 4;
        const int s = rs & 15;
        int value = 0;
        if (s == 0) {
          if (r < 15) {
            eobrun = (1 << r);
            if (r) eobrun += getbits(r);
            break;  // the rest of this block is refined below as part of the run
          }
          // r == 15: skip sixteen zero-history coefficients
        } else {
          if (s != 1) bad("refinement scan with a coefficient magnitude other than 1");
          value = getbit() ? p1 : m1;
        }
        // pass over r coefficients whose history is zero, refining the non-zero ones on the way
        for (; i <= se; ++i) {
          int16_t& c = b[kZigzag[i]];
          if (c != 0) {
            if (getbit() && (c & p1) == 0) c = (int16_t)(c >= 0 ? c + p1 : c + m1);
          } else {
            if (r == 0) break;
            --r;
          }
        }
        if (value && i <= se) b[kZigzag[i]] = (int16_t)value;
        ++i;
      }
    }
    if (eobrun > 0) {
      for (; i <= se; ++i) {
        int16_t& c = b[kZigzag[i]];
        if (c != 0 && getbit() && (c & p1) == 0) c = (int16_t)(c >= 0 ? c + p1 : c + m1);
      }
      --eobrun;
    }
  }

  void read_sos() {
    if (!have_frame) bad("scan before the frame header");
    const int len = u16();
    const int ns = u8();
    if (ns < 1 || ns > ncomp || len != 6 + 2 * ns) bad("scan header");
    int order[3];
    for (int i = 0; i < ns; ++i) {
      const int id = u8(), tt = u8();
      int c = 0;
      while (c < ncomp && comp[c].id != id) ++c;
      if (c == ncomp) bad("scan names a component the frame does not have");
      comp[c].td = tt >> 4;
      comp[c].ta = tt & 15;
      if (comp[c].td > 3 || comp[c].ta > 3) bad("scan table selector");
      order[i] = c;
    }
    const int ss = u8(), se = u8(), ahal = u8();
    const int ah = ahal >> 4, al = ahal & 15;
    if (progressive) {
      if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || al > 13) bad("progressive scan parameters");
    } else if (ss != 0 || se != 63 || ah != 0 || al != 0) {
      bad("sequential scan parameters");
    }
    for (int i = 0; i < ns; ++i) {
      const Component& k = comp[order[i]];
      const bool need_dc = !progressive || ss == 0, need_ac = !progressive || ss > 0;
      if (need_dc && !(progressive && ah) && !dc[k.td].defined) bad("scan uses a DC Huffman table that was not defined");
      if (need_ac && !ac[k.ta].defined) bad("scan uses an AC Huffman table that was not defined");
    }
    reset_entropy();
    auto do_block = [&](Component& k, int bx, int by) {
      int16_t* b = k.coef.data() + ((size_t)by * k.blocks_w + bx) * 64;
      if (!progressive) block_sequential(k, b);
      else if (ss == 0) { if (ah == 0) block_dc_first(k, b, al); else block_dc_refine(b, al); }
      else if (ah == 0) block_ac_first(k, b, ss, se, al);
      else block_ac_refine(k, b, ss, se, al);
    };
    int countdown = restart_interval;
    auto restart_tick = [&]() {
      if (restart_interval && --countdown == 0) {
        take_restart();
        countdown = restart_interval;
      }
    };
    if (ns == 1) {
      // a scan of one component is not interleaved: its MCU is one block and the scan covers only the
      // blocks that carry image (T.81 A.2.2)
      Component& k = comp[order[0]];
      const int bw = (k.width + 7) / 8, bh = (k.height + 7) / 8;
      const int total = bw * bh;
      int done = 0;
      for (int by = 0; by < bh; ++by)
        for (int bx = 0; bx < bw; ++bx) {
          do_block(k, bx, by);
          if (++done < total) restart_tick();
        }
    } else {
      const int total = mcus_x * mcus_y;
      int done = 0;
      for (int my = 0; my < mcus_y; ++my)
        for (int mx = 0; mx < mcus_x; ++mx) {
          for (int i = 0; i < ns; ++i) {
            Component& k = comp[order[i]];
            for (int y = 0; y < k.v; ++y)
              for (int x = 0; x < k.h; ++x) do_block(k, mx * k.h + x, my * k.v + y);
          }
          if (++done < total) restart_tick();
        }
    }
    // what follows the entropy-coded data is the next marker; bytes the bit reader took ahead of it
    // were only ever data bytes, so the marker is at or after `pos`
  }

  // ---- inverse DCT: the accurate integer method (LL&M), constants scaled by 2^13, a column pass that
  // keeps two extra bits and a row pass that removes them with the 2^3 of the transform's scaling
  static int descale(int64_t x, int nbits) { return (int)((x + ((int64_t)1 << (nbits - 1))) >> nbits); }
  static uint8_t idct_limit(int v) {  // v + 128 limited to 0..255 (indices wrap at 1024 as in a table lookup)
    const int i = v & 1023;
    if (i < 512) return (uint8_t)(i + 128 > 255 ? 255 : i + 128);
    const int neg = i - 1024 + 128;
    return (uint8_t)(neg < 0 ? 0 : neg);
  }
  static void idct_1d(const int in[8], int out[8], int shift) {
    const int F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299,
              F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
    int64_t z2 = in[2], z3 = in[6];
    int64_t z1 = (z2 + z3) * F0_541;
    int64_t tmp2 = z1 + z3 * (-F1_847);
    int64_t tmp3 = z1 + z2 * F0_765;
    z2 = in[0];
    z3 = in[4];
    int64_t tmp0 = (z2 + z3) * (1 << 13);
    int64_t tmp1 = (z2 - z3) * (1 << 13);
    const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7];
    tmp1 = in[5];
    tmp2 = in[3];
    tmp3 = in[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    int64_t z4 = tmp1 + tmp3;
    const int64_t z5 = (z3 + z4) * F1_175;
    tmp0 *= F0_298;
    tmp1 *= F2_053;
    tmp2 *= F3_072;
    tmp3 *= F1_501;
    z1 *= -F0_899;
    z2 *= -F2_562;
    z3 *= -F1_961;
    z4 *= -F0_390;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    out[0] = descale(tmp10 + tmp3, shift);
    out[7] = descale(tmp10 - tmp3, shift);
    out[1] = descale(tmp11 + tmp2, shift);
    out[6] = descale(tmp11 - tmp2, shift);
    out[2] = descale(tmp12 + tmp1, shift);
    out[5] = descale(tmp12 - tmp1, shift);
    out[3] = descale(tmp13 + tmp0, shift);
    out[4] = descale(tmp13 - tmp0, shift);
  }
  void idct_block(const int16_t* b, const uint16_t* q, uint8_t* out, int stride) {
    int ws[64];
    for (int x = 0; x < 8; ++x) {
      int in[8], o[8];
      for (int y = 0; y < 8; ++y) in[y] = (int)b[y * 8 + x] * (int)q[y * 8 + x];
      idct_1d(in, o, 13 - 2);
      for (int y = 0; y < 8; ++y) ws[y * 8 + x] = o[y];
    }
    for (int y = 0; y < 8; ++y) {
      int o[8];
      idct_1d(ws + y * 8, o, 13 + 2 + 3);
      for (int x = 0; x < 8; ++x) out[y * stride + x] = idct_limit(o[x]);
    }
  }
  void reconstruct_planes() {
    for (int c = 0; c < ncomp; ++c) {
      Component& k = comp[c];
      if (!qt_defined[k.tq]) bad("a component names a quantisation table that was not defined");
      const int stride = k.blocks_w * 8;
      k.plane.assign((size_t)stride * k.blocks_h * 8, 0);
      for (int by = 0; by < k.blocks_h; ++by)
        for (int bx = 0; bx < k.blocks_w; ++bx)
          idct_block(k.coef.data() + ((size_t)by * k.blocks_w + bx) * 64, qt[k.tq], k.plane.data() + (size_t)by * 8 * stride + bx * 8, stride);
    }
  }

  // ---- a component at full size. Rows and columns beyond the component's own size do not exist for the
  // filter: the edge sample is repeated, as the decoder's context rows repeat the first and the last row.
  void upsample(const Component& k, std::vector<uint8_t>& full) {
    const int fx = hmax / k.h, fy = vmax / k.v;
    const int W = k.width, H = k.height, stride = k.blocks_w * 8;
    const int outw = W * fx, outh = H * fy;
    full.assign((size_t)outw * outh, 0);
    auto row = [&](int y) { return k.plane.data() + (size_t)(y < 0 ? 0 : (y >= H ? H - 1 : y)) * stride; };
    if (fx == 1 && fy == 1) {
      for (int y = 0; y < H; ++y) memcpy(full.data() + (size_t)y * outw, row(y), (size_t)W);
    } else if (fx == 2 && fy == 1 && W > 2) {  // (components of one or two columns are replicated, below)
      for (int y = 0; y < H; ++y) {
        const uint8_t* in = row(y);
        uint8_t* out = full.data() + (size_t)y * outw;
        out[0] = in[0];
        out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
        for (int i = 1; i < W - 1; ++i) {
          const int v = in[i] * 3;
          out[2 * i] = (uint8_t)((v + in[i - 1] + 1) >> 2);
          out[2 * i + 1] = (uint8_t)((v + in[i + 1] + 2) >> 2);
        }
        out[2 * (W - 1)] = (uint8_t)((in[W - 1] * 3 + in[W - 2] + 1) >> 2);
        out[2 * (W - 1) + 1] = in[W - 1];
      }
    } else if (fx == 2 && fy == 2 && W > 2) {
      for (int y = 0; y < H; ++y)
        for (int v = 0; v < 2; ++v) {
          const uint8_t* in0 = row(y);
          const uint8_t* in1 = row(v == 0 ? y - 1 : y + 1);
          uint8_t* out = full.data() + (size_t)(2 * y + v) * outw;
          int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
          out[0] = (uint8_t)((thiscol * 4 + 8) >> 4);
          out[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
          lastcol = thiscol;
          thiscol = nextcol;
          for (int i = 1; i < W - 1; ++i) {
            nextcol = in0[i + 1] * 3 + in1[i + 1];
            out[2 * i] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
            out[2 * i + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
            lastcol = thiscol;
            thiscol = nextcol;
          }
          out[2 * (W - 1)] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
          out[2 * (W - 1) + 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
        }
    } else if (fx == 1 && fy == 2) {
      for (int y = 0; y < H; ++y)
        for (int v = 0; v < 2; ++v) {
          const uint8_t* in0 = row(y);
          const uint8_t* in1 = row(v == 0 ? y - 1 : y + 1);
          uint8_t* out = full.data() + (size_t)(2 * y + v) * outw;
          const int bias = v == 0 ? 1 : 2;
          for (int i = 0; i < W; ++i) out[i] = (uint8_t)((in0[i] * 3 + in1[i] + bias) >> 2);
        }
    } else {
      for (int y = 0; y < outh; ++y) {
        const uint8_t* in = row(y / fy);
        uint8_t* out = full.data() + (size_t)y * outw;
        for (int x = 0; x < outw; ++x) out[x] = in[x / fx];
      }
    }
  }

  void parse() {
    if (n < 4 || p[0] != 0xFF || p[1] != 0xD8) bad("not a JPEG file (no SOI marker)");
    pos = 2;
    bool eoi = false, any_scan = false;
    while (!eoi) {
      // next marker: 0xFF, any number of 0xFF fill bytes, then the code
      if (pos >= n) {
        if (any_scan) break;  // a file cut off after its scans still has its picture
        bad("file ends before any scan");
      }
      if (p[pos] != 0xFF) {
        ++pos;  // (bytes the last scan did not consume)
        continue;
      }
      while (pos < n && p[pos] == 0xFF) ++pos;
      if (pos >= n) break;
      const int m = p[pos++];
      if (m == 0 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
      switch (m) {
        case 0xD8: break;
        case 0xD9: eoi = true; break;
        case 0xDB: read_dqt(); break;
        case 0xC4: read_dht(); break;
        case 0xC0: case 0xC1: case 0xC2: read_sof(m); break;
        case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
          unsupported("lossless and hierarchical processes are not supported");
        case 0xC9: case 0xCA: case 0xCC:
          unsupported("arithmetic coding is not supported");
        case 0xDA: read_sos(); any_scan = true; break;
        case 0xDD: {
          if (u16() != 4) bad("restart interval segment length");
          restart_interval = u16();
          break;
        }
        case 0xEE: {  // Adobe: the colour transform flag
          const int len = u16();
          if (len >= 14 && pos + 12 <= n && memcmp(p + pos, "Adobe", 5) == 0) {
            adobe = true;
            adobe_transform = p[pos + 11];
          }
          if (len < 2 || pos + (size_t)(len - 2) > n) bad("segment longer than the file");
          pos += (size_t)(len - 2);
          break;
        }
        default: {  // APPn, COM and whatever else carries a length
          const int len = u16();
          if (len < 2 || pos + (size_t)(len - 2) > n) bad("segment longer than the file");
          pos += (size_t)(len - 2);
        }
      }
    }
    if (!have_frame || !any_scan) bad("no image data");
  }

  void to_rgb(uint8_t* rgb) {
    reconstruct_planes();
    std::vector<uint8_t> full[3];
    for (int c = 0; c < ncomp; ++c) upsample(comp[c], full[c]);
    if (ncomp == 1) {
      const int ow = comp[0].width;
      for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
          const uint8_t v = full[0][(size_t)y * ow + x];
          uint8_t* o = rgb + ((size_t)y * width + x) * 3;
          o[0] = o[1] = o[2] = v;
        }
      return;
    }
    // three components are Y Cb Cr unless an Adobe marker says its transform is 0, or, with no JFIF or Adobe
    // marker to go by, the component ids spell R G B
    bool ycc = true;
    if (adobe) ycc = adobe_transform != 0;
    else if (comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B') ycc = false;
    int ow[3];
    for (int c = 0; c < 3; ++c) ow[c] = comp[c].width * (hmax / comp[c].h);
    // 16-bit fixed point: FIX(x) = x * 65536 rounded, half added before each shift (the green terms share one)
    const int F1_402 = 91881, F1_772 = 116130, F0_714 = 46802, F0_344 = 22554, HALF = 1 << 15;
    for (int y = 0; y < height; ++y)
      for (int x = 0; x < width; ++x) {
        const int Y = full[0][(size_t)y * ow[0] + x], cb = full[1][(size_t)y * ow[1] + x] - 128, cr = full[2][(size_t)y * ow[2] + x] - 128;
        uint8_t* o = rgb + ((size_t)y * width + x) * 3;
        if (!ycc) {
          o[0] = (uint8_t)Y; o[1] = (uint8_t)(cb + 128); o[2] = (uint8_t)(cr + 128);
          continue;
        }
        const int r = Y + ((F1_402 * cr + HALF) >> 16);
        const int g = Y + ((-F0_344 * cb + HALF - F0_714 * cr) >> 16);
        const int b = Y + ((F1_772 * cb + HALF) >> 16);
        o[0] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        o[1] = (uint8_t)(g < 0 ? 0 : (g > 255 ? 255 : g));
        o[2] = (uint8_t)(b < 0 ? 0 : (b > 255 ? 255 : b));
      }
  }
};

}  // namespace

extern "C" int rt_image_decode_jpeg(const uint8_t* bytes, uint64_t n_bytes, int32_t* width, int32_t* height, uint8_t* rgb8,
                                    uint64_t capacity) {
  if (!bytes || !width || !height) return rt::fail(RT_ERR_ARG, "rt_image_decode_jpeg: bytes, width and height must not be NULL");
  *width = *height = 0;
  try {
    Decoder d;
    d.p = bytes;
    d.n = (size_t)n_bytes;
    d.parse();
    *width = d.width;
    *height = d.height;
    if (!rgb8) return RT_OK;  // the size alone (the file is checked through to its end all the same)
    if (capacity < (uint64_t)d.width * (uint64_t)d.height * 3)
      return rt::fail(RT_ERR_ARG, "rt_image_decode_jpeg: the image is %d x %d, %llu bytes, the buffer holds %llu", d.width, d.height,
                      (unsigned long long)d.width * d.height * 3, (unsigned long long)capacity);
    d.to_rgb(rgb8);
    return RT_OK;
  } catch (const JErr& e) {
    return rt::fail(e.code, "%s", e.msg.c_str());
  } catch (const std::bad_alloc&) {
    return rt::fail(RT_ERR_OOM, "out of host memory decoding a JPEG file");
  } catch (...) {
    return rt::fail(RT_ERR_ARG, "JPEG: decoding failed");
  }
}
