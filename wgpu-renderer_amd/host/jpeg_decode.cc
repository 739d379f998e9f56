// Baseline and progressive Huffman JPEG decoder (8-bit, 1 or 3 components) for the host-side
// texture loader: the counterpart of `createImageBitmap` in the reference's loadImageBitmap
// (src/utils.ts:13-18).  The arithmetic follows the integer pipeline that libjpeg-family decoders
// (the ones browsers and Pillow embed) run by default, so that the decoded texels are the same
// bytes: 13-bit "slow integer" inverse DCT, triangle-filter ("fancy") chroma upsampling, and the
// 16-bit fixed-point YCbCr -> RGB tables.  Not reproduced: arithmetic coding, 12-bit samples,
// CMYK/YCCK, inter-block smoothing of progressive files whose scans stop early, EXIF orientation.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

struct DecodeError { char msg[160]; };

[[noreturn]] void fail(const char* fmt, ...) {
  DecodeError e;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(e.msg, sizeof(e.msg), fmt, ap);
  va_end(ap);
  throw e;
}

// zig-zag position -> natural (row-major) position; 16 extra entries keep a corrupt run in bounds
const uint8_t kNatural[64 + 16] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct HuffTable {
  bool defined = false;
  uint8_t counts[17] = {0};
  uint8_t symbols[256] = {0};
  int32_t max_code[18];     // largest code of each length, -1 if none
  int32_t val_offset[17];   // symbols index of the first code of a length minus that code
  uint16_t fast[512];       // 9-bit lookahead: (length << 8) | symbol, 0 if the code is longer

  void build() {
    int32_t code = 0, k = 0;
    for (int len = 1; len <= 16; ++len) {
      val_offset[len] = k - code;
      k += counts[len];
      code += counts[len];
      max_code[len] = counts[len] ? code - 1 : -1;
      code <<= 1;
    }
    max_code[17] = 0x7fffffff;
    std::memset(fast, 0, sizeof(fast));
    code = 0; k = 0;
    for (int len = 1; len <= 9; ++len) {
      for (int i = 0; i < counts[len]; ++i, ++k, ++code) {
        const int first = code << (9 - len), n = 1 << (9 - len);
        for (int j = 0; j < n; ++j) fast[first + j] = (uint16_t)((len << 8) | symbols[k]);
      }
      code <<= 1;
    }
    defined = true;
  }
};

struct BitReader {
  const uint8_t* p = nullptr;
  const uint8_t* end = nullptr;
  uint64_t acc = 0;   // bits are consumed from the top
  int count = 0;
  bool at_marker = false;

  void reset(const uint8_t* from) { p = from; acc = 0; count = 0; at_marker = false; }
  void fill() {
    while (count <= 56) {
      uint64_t b = 0;
      if (!at_marker && p < end) {
        b = *p;
        if (b == 0xFF) {
          if (p + 1 < end && p[1] == 0x00) p += 2;          // stuffed zero
          else { at_marker = true; b = 0; }                  // a marker ends the entropy data; zeros follow
        } else {
          ++p;
        }
      }
      acc |= b << (56 - count);
      count += 8;
    }
  }
  inline int peek(int n) { if (count < n) fill(); return (int)(acc >> (64 - n)); }
  inline void drop(int n) { acc <<= n; count -= n; }
  inline int bits(int n) { if (n == 0) return 0; const int v = peek(n); drop(n); return v; }
  inline int bit() { return bits(1); }
};

inline int extend(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }

inline int decode_symbol(BitReader& br, const HuffTable& t) {
  const int look = br.peek(16);
  const uint16_t f = t.fast[look >> 7];
  if (f) { br.drop(f >> 8); return f & 255; }
  int len = 10;
  int32_t code = look >> 6;
  while (len <= 16 && code > t.max_code[len]) { ++len; code = look >> (16 - len); }
  if (len > 16) { br.drop(16); return 0; }   // corrupt data: libjpeg warns and returns 0
  br.drop(len);
  return t.symbols[(code + t.val_offset[len]) & 255];
}

struct Component {
  int id = 0, h = 1, v = 1, tq = 0;
  int blocks_w = 0, blocks_h = 0;      // allocated blocks (whole MCUs)
  int real_bw = 0, real_bh = 0;        // blocks that hold image samples (non-interleaved scans walk these)
  int down_w = 0, down_h = 0;          // samples that hold image data
  int dc_pred = 0;
  int td = 0, ta = 0;
  std::vector<int16_t> coef;           // blocks_w * blocks_h * 64, natural order
  std::vector<uint8_t> plane;          // (blocks_w * 8) x (blocks_h * 8)
};

struct Decoder {
  const uint8_t* data; size_t len;
  int width = 0, height = 0, ncomp = 0, hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;
  bool progressive = false, saw_jfif = false, saw_adobe = false;
  int adobe_transform = 0;
  int restart_interval = 0;
  uint16_t quant[4][64];
  bool quant_defined[4] = {false, false, false, false};
  HuffTable dc_tab[4], ac_tab[4];
  Component comp[3];
  int eobrun = 0;

  static int ceil_div(int a, int b) { return (a + b - 1) / b; }
  uint16_t be16(size_t at) const { if (at + 2 > len) fail("Truncated JPEG file."); return (uint16_t)((data[at] << 8) | data[at + 1]); }

  void read_dqt(size_t at, size_t seg_end) {
    while (at < seg_end) {
      const int pq = data[at] >> 4, tq = data[at] & 15;
      ++at;
      if (tq > 3) fail("Bad quantization table index %d.", tq);
      for (int i = 0; i < 64; ++i) {
        if (at + (pq ? 2 : 1) > seg_end) fail("Truncated quantization table.");
        const int v = pq ? (data[at] << 8) | data[at + 1] : data[at];
        at += pq ? 2 : 1;
        quant[tq][kNatural[i]] = (uint16_t)v;
      }
      quant_defined[tq] = true;
    }
  }

  void read_dht(size_t at, size_t seg_end) {
    while (at < seg_end) {
      const int tc = data[at] >> 4, th = data[at] & 15;
      ++at;
      if (tc > 1 || th > 3) fail("Bad Huffman table index.");
      HuffTable& t = tc ? ac_tab[th] : dc_tab[th];
      if (at + 16 > seg_end) fail("Truncated Huffman table.");
      int total = 0;
      t.counts[0] = 0;
      for (int i = 1; i <= 16; ++i) { t.counts[i] = data[at + i - 1]; total += t.counts[i]; }
      at += 16;
      if (total > 256 || at + total > seg_end) fail("Bad Huffman table.");
      std::memcpy(t.symbols, data + at, total);
      at += total;
      t.build();
    }
  }

  void read_sof(size_t at, size_t seg_end, int marker) {
    if (ncomp) fail("Multiple frames in one JPEG file are not supported.");
    if (marker != 0xC0 && marker != 0xC1 && marker != 0xC2) fail("Unsupported JPEG process (SOF%d).", marker - 0xC0);
    progressive = marker == 0xC2;
    if (at + 6 > seg_end) fail("Truncated frame header.");
    if (data[at] != 8) fail("Unsupported JPEG sample precision %d.", data[at]);
    height = be16(at + 1); width = be16(at + 3); ncomp = data[at + 5];
    if (width <= 0 || height <= 0) fail("Empty JPEG image.");
    if (ncomp != 1 && ncomp != 3) fail("Unsupported JPEG component count %d.", ncomp);
    if (at + 6 + 3 * (size_t)ncomp > seg_end) fail("Truncated frame header.");
    for (int i = 0; i < ncomp; ++i) {
      Component& c = comp[i];
      c.id = data[at + 6 + 3 * i];
      c.h = data[at + 7 + 3 * i] >> 4; c.v = data[at + 7 + 3 * i] & 15;
      c.tq = data[at + 8 + 3 * i];
      if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) fail("Bad component description.");
      if (c.h > hmax) hmax = c.h;
      if (c.v > vmax) vmax = c.v;
    }
    mcus_x = ceil_div(width, 8 * hmax); mcus_y = ceil_div(height, 8 * vmax);
    for (int i = 0; i < ncomp; ++i) {
      Component& c = comp[i];
      c.down_w = ceil_div(width * c.h, hmax); c.down_h = ceil_div(height * c.v, vmax);
      c.real_bw = ceil_div(c.down_w, 8); c.real_bh = ceil_div(c.down_h, 8);
      c.blocks_w = mcus_x * c.h; c.blocks_h = mcus_y * c.v;
      c.coef.assign((size_t)c.blocks_w * c.blocks_h * 64, 0);
    }
  }

  // ---- entropy-coded block decoders (ITU T.81 F.2.2, G.1.2) ----
  void block_sequential(BitReader& br, Component& c, int16_t* blk) {
    const HuffTable& dc = dc_tab[c.td];
    const HuffTable& ac = ac_tab[c.ta];
    const int t = decode_symbol(br, dc);
    const int diff = t ? extend(br.bits(t), t) : 0;
    c.dc_pred += diff;
    blk[0] = (int16_t)c.dc_pred;
    int k = 1;
    while (k < 64) {
      const int rs = decode_symbol(br, ac), s = rs & 15, r = rs >> 4;
      if (s == 0) {
        if (r != 15) break;
        k += 16;
      } else {
        k += r;
        blk[kNatural[k]] = (int16_t)extend(br.bits(s), s);
        ++k;
      }
    }
  }

  void block_dc_progressive(BitReader& br, Component& c, int16_t* blk, int ah, int al) {
    if (ah == 0) {
      const int t = decode_symbol(br, dc_tab[c.td]);
      const int diff = t ? extend(br.bits(t), t) : 0;
      c.dc_pred += diff;
      blk[0] = (int16_t)(c.dc_pred * (1 << al));
    } else if (br.bit()) {
      blk[0] = (int16_t)(blk[0] | (1 << al));
    }
  }

  void block_ac_progressive(BitReader& br, Component& c, int16_t* blk, int ss, int se, int ah, int al) {
    const HuffTable& ac = ac_tab[c.ta];
    if (ah == 0) {
      if (eobrun) { --eobrun; return; }
      int k = ss;
      while (k <= se) {
        const int rs = decode_symbol(br, ac), s = rs & 15, r = rs >> 4;
        if (s == 0) {
          if (r < 15) {
            eobrun = (1 << r) - 1;
            if (r) eobrun += br.bits(r);
            break;
          }
          k += 16;
        } else {
          k += r;
          blk[kNatural[k]] = (int16_t)(extend(br.bits(s), s) * (1 << al));
          ++k;
        }
      }
      return;
    }
    const int plus = 1 << al, minus = -plus;
    auto refine = [&](int16_t* p) {
      if (br.bit() && (*p & plus) == 0) *p = (int16_t)(*p >= 0 ? *p + plus : *p + minus);
    };
    int k = ss;
    if (eobrun == 0) {
      while (k <= se) {
        const int rs = decode_symbol(br, ac);
        int s = rs & 15, r = rs >> 4;
        if (s == 0) {
          if (r < 15) {
            eobrun = 1 << r;
            if (r) eobrun += br.bits(r);
            break;                      // the rest of this block is handled below as part of the run
          }
        } else {
          s = br.bit() ? plus : minus;  // a newly non-zero coefficient has magnitude 1 at this bit plane
        }
        while (k <= se) {
          int16_t* p = &blk[kNatural[k]];
          ++k;
          if (*p != 0) {
            refine(p);
          } else {
            if (r == 0) { if (s) *p = (int16_t)s; break; }
            --r;
          }
        }
      }
    }
    if (eobrun > 0) {
      for (; k <= se; ++k) {
        int16_t* p = &blk[kNatural[k]];
        if (*p != 0) refine(p);
      }
      --eobrun;
    }
  }

  // One scan: header at `at` (after the length), entropy data follows; returns the offset of the next marker.
  size_t read_scan(size_t at, size_t seg_end) {
    if (!ncomp) fail("Scan before frame header.");
    const int ns = data[at];
    if (ns < 1 || ns > ncomp || at + 1 + 2 * (size_t)ns + 3 > seg_end) fail("Bad scan header.");
    Component* sc[3];
    for (int i = 0; i < ns; ++i) {
      const int id = data[at + 1 + 2 * i], tabs = data[at + 2 + 2 * i];
      sc[i] = nullptr;
      for (int j = 0; j < ncomp; ++j) if (comp[j].id == id) sc[i] = &comp[j];
      if (!sc[i]) fail("Scan names an unknown component.");
      sc[i]->td = tabs >> 4; sc[i]->ta = tabs & 15;
      if (sc[i]->td > 3 || sc[i]->ta > 3) fail("Bad Huffman table selector.");
    }
    int ss = data[at + 1 + 2 * ns], se = data[at + 2 + 2 * ns];
    const int ah = data[at + 3 + 2 * ns] >> 4, al = data[at + 3 + 2 * ns] & 15;
    if (progressive) {
      if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || al > 13) fail("Bad progression parameters.");
    } else {
      ss = 0; se = 63;
    }
    const bool need_dc = !progressive || ss == 0, need_ac = !progressive || ss > 0;
    for (int i = 0; i < ns; ++i) {
      if (need_dc && !(progressive && ah) && !dc_tab[sc[i]->td].defined) fail("Huffman DC table %d was not defined.", sc[i]->td);
      if (need_ac && !ac_tab[sc[i]->ta].defined) fail("Huffman AC table %d was not defined.", sc[i]->ta);
    }

    BitReader br;
    br.end = data + len;
    br.reset(data + seg_end);
    auto restart_state = [&]() { for (int i = 0; i < ncomp; ++i) comp[i].dc_pred = 0; eobrun = 0; };
    restart_state();

    auto one_block = [&](Component& c, int bx, int by) {
      int16_t* blk = &c.coef[((size_t)by * c.blocks_w + bx) * 64];
      if (!progressive) block_sequential(br, c, blk);
      else if (ss == 0) block_dc_progressive(br, c, blk, ah, al);
      else block_ac_progressive(br, c, blk, ss, se, ah, al);
    };

    const bool interleaved = ns > 1;
    const int units_x = interleaved ? mcus_x : sc[0]->real_bw, units_y = interleaved ? mcus_y : sc[0]->real_bh;
    const long total = (long)units_x * units_y;
    int until_restart = restart_interval;
    for (long unit = 0; unit < total; ++unit) {
      const int ux = (int)(unit % units_x), uy = (int)(unit / units_x);
      if (interleaved) {
        for (int i = 0; i < ns; ++i)
          for (int v = 0; v < sc[i]->v; ++v)
            for (int h = 0; h < sc[i]->h; ++h) one_block(*sc[i], ux * sc[i]->h + h, uy * sc[i]->v + v);
      } else {
        one_block(*sc[0], ux, uy);
      }
      if (restart_interval && --until_restart == 0 && unit + 1 < total) {
        // byte-align, then expect RSTn
        // the reader never moves past a marker, so the restart marker is at or after its position;
        // anything in between belongs to the interval just finished (unread padding)
        const uint8_t* q = br.p;
        while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) {
          if (q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF) break;   // some other marker: stop looking
          ++q;
        }
        if (q + 1 < br.end && q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7) q += 2;
        br.reset(q);
        restart_state();
        until_restart = restart_interval;
      }
    }
    // locate the next marker
    const uint8_t* q = br.p;
    while (q + 1 < br.end && !(q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF && !(q[1] >= 0xD0 && q[1] <= 0xD7))) ++q;
    return (size_t)(q - data);
  }

  // ---- inverse DCT: 8x8, 13-bit fixed-point constants, two passes with 2 extra bits in between ----
  static inline uint8_t range_limit(int64_t x) {
    const int v = (int)(x & 1023);   // the 10-bit wrap of the classic range-limit table
    if (v < 128) return (uint8_t)(v + 128);
    if (v < 512) return 255;
    if (v < 896) return 0;
    return (uint8_t)(v - 896);
  }

  static void idct_block(const int16_t* in, const uint16_t* q, uint8_t* out, int stride) {
    constexpr int64_t F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137,
                      F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
    int64_t ws[64];
    auto pass = [&](const int64_t v[8], int64_t o[8], int shift) {
      int64_t z2 = v[2], z3 = v[6];
      int64_t z1 = (z2 + z3) * F0541;
      int64_t tmp2 = z1 + z3 * (-F1847);
      int64_t tmp3 = z1 + z2 * F0765;
      z2 = v[0]; z3 = v[4];
      int64_t tmp0 = (z2 + z3) * 8192;
      int64_t tmp1 = (z2 - z3) * 8192;
      const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
      tmp0 = v[7]; tmp1 = v[5]; tmp2 = v[3]; tmp3 = v[1];
      z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
      int64_t z4 = tmp1 + tmp3;
      const int64_t z5 = (z3 + z4) * F1175;
      tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
      z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
      z3 += z5; z4 += z5;
      tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
      const int64_t half = (int64_t)1 << (shift - 1);
      o[0] = (tmp10 + tmp3 + half) >> shift; o[7] = (tmp10 - tmp3 + half) >> shift;
      o[1] = (tmp11 + tmp2 + half) >> shift; o[6] = (tmp11 - tmp2 + half) >> shift;
      o[2] = (tmp12 + tmp1 + half) >> shift; o[5] = (tmp12 - tmp1 + half) >> shift;
      o[3] = (tmp13 + tmp0 + half) >> shift; o[4] = (tmp13 - tmp0 + half) >> shift;
    };
    for (int col = 0; col < 8; ++col) {
      int64_t v[8], o[8];
      for (int r = 0; r < 8; ++r) v[r] = (int64_t)in[r * 8 + col] * q[r * 8 + col];
      pass(v, o, 13 - 2);
      for (int r = 0; r < 8; ++r) ws[r * 8 + col] = o[r];
    }
    for (int row = 0; row < 8; ++row) {
      int64_t o[8];
      pass(&ws[row * 8], o, 13 + 2 + 3);
      for (int c = 0; c < 8; ++c) out[row * stride + c] = range_limit(o[c]);
    }
  }

  void inverse_transform() {
    for (int i = 0; i < ncomp; ++i) {
      Component& c = comp[i];
      if (!quant_defined[c.tq]) fail("Quantization table %d was not defined.", c.tq);
      const int stride = c.blocks_w * 8;
      c.plane.assign((size_t)stride * c.blocks_h * 8, 0);
      for (int by = 0; by < c.blocks_h; ++by)
        for (int bx = 0; bx < c.blocks_w; ++bx)
          idct_block(&c.coef[((size_t)by * c.blocks_w + bx) * 64], quant[c.tq], &c.plane[(size_t)by * 8 * stride + bx * 8], stride);
      std::vector<int16_t>().swap(c.coef);
    }
  }

  // ---- chroma upsampling to full resolution (triangle filter, edge samples replicated) ----
  std::vector<uint8_t> upsample(const Component& c) const {
    const int hs = hmax / c.h, vs = vmax / c.v;
    if (hmax % c.h || vmax % c.v) fail("Fractional sampling ratios are not supported.");
    const int stride = c.blocks_w * 8, dw = c.down_w, dh = c.down_h;
    const int ow = dw * hs, oh = dh * vs;
    std::vector<uint8_t> out((size_t)ow * oh);
    auto src = [&](int y) { return &c.plane[(size_t)(y < 0 ? 0 : (y >= dh ? dh - 1 : y)) * stride]; };
    if (hs == 1 && vs == 1) {
      for (int y = 0; y < oh; ++y) std::memcpy(&out[(size_t)y * ow], src(y), ow);
    } else if (hs == 2 && vs == 1 && dw > 2) {
      for (int y = 0; y < oh; ++y) {
        const uint8_t* in = src(y);
        uint8_t* o = &out[(size_t)y * ow];
        o[0] = in[0];
        o[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
        for (int x = 1; x < dw - 1; ++x) {
          const int v = in[x] * 3;
          o[2 * x] = (uint8_t)((v + in[x - 1] + 1) >> 2);
          o[2 * x + 1] = (uint8_t)((v + in[x + 1] + 2) >> 2);
        }
        o[2 * dw - 2] = (uint8_t)((in[dw - 1] * 3 + in[dw - 2] + 1) >> 2);
        o[2 * dw - 1] = in[dw - 1];
      }
    } else if (hs == 1 && vs == 2) {
      for (int y = 0; y < dh; ++y)
        for (int half = 0; half < 2; ++half) {
          const uint8_t* in0 = src(y);
          const uint8_t* in1 = src(half == 0 ? y - 1 : y + 1);
          uint8_t* o = &out[(size_t)(2 * y + half) * ow];
          const int bias = half == 0 ? 1 : 2;
          for (int x = 0; x < dw; ++x) o[x] = (uint8_t)((in0[x] * 3 + in1[x] + bias) >> 2);
        }
    } else if (hs == 2 && vs == 2 && dw > 2) {
      for (int y = 0; y < dh; ++y)
        for (int half = 0; half < 2; ++half) {
          const uint8_t* in0 = src(y);
          const uint8_t* in1 = src(half == 0 ? y - 1 : y + 1);
          uint8_t* o = &out[(size_t)(2 * y + half) * ow];
          int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
          o[0] = (uint8_t)((thiscol * 4 + 8) >> 4);
          o[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
          lastcol = thiscol; thiscol = nextcol;
          for (int x = 1; x < dw - 1; ++x) {
            nextcol = in0[x + 1] * 3 + in1[x + 1];
            o[2 * x] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
            o[2 * x + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
            lastcol = thiscol; thiscol = nextcol;
          }
          o[2 * dw - 2] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
          o[2 * dw - 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
        }
    } else {
      // integral replication for every other ratio (also what narrow planes get)
      for (int y = 0; y < oh; ++y) {
        const uint8_t* in = src(y / vs);
        uint8_t* o = &out[(size_t)y * ow];
        for (int x = 0; x < ow; ++x) o[x] = in[x / hs];
      }
    }
    // keep (ow, oh) implicit: callers index with ow = down_w * hs
    return out;
  }

  static inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

  void to_rgba(uint8_t* rgba) {
    if (ncomp == 1) {
      const Component& c = comp[0];
      const int stride = c.blocks_w * 8;
      for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
          const uint8_t g = c.plane[(size_t)y * stride + x];
          uint8_t* o = rgba + ((size_t)y * width + x) * 4;
          o[0] = g; o[1] = g; o[2] = g; o[3] = 255;
        }
      return;
    }
    std::vector<uint8_t> full[3];
    int fw[3];
    for (int i = 0; i < 3; ++i) { full[i] = upsample(comp[i]); fw[i] = comp[i].down_w * (hmax / comp[i].h); }
    // colour space of a 3-component file: JFIF means YCbCr; else the Adobe transform flag; else the component ids
    bool ycc = true;
    if (!saw_jfif) {
      if (saw_adobe) ycc = adobe_transform != 0;
      else if (comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B') ycc = false;
    }
    int cr_r[256], cb_b[256], cr_g[256], cb_g[256];
    for (int i = 0; i < 256; ++i) {
      const int x = i - 128;
      cr_r[i] = (int)((91881LL * x + 32768) >> 16);    // 1.40200
      cb_b[i] = (int)((116130LL * x + 32768) >> 16);   // 1.77200
      cr_g[i] = -46802 * x;                            // 0.71414
      cb_g[i] = -22554 * x + 32768;                    // 0.34414, carries the rounding term
    }
    for (int y = 0; y < height; ++y)
      for (int x = 0; x < width; ++x) {
        const int a = full[0][(size_t)y * fw[0] + x], b = full[1][(size_t)y * fw[1] + x], c = full[2][(size_t)y * fw[2] + x];
        uint8_t* o = rgba + ((size_t)y * width + x) * 4;
        if (ycc) {
          o[0] = clamp8(a + cr_r[c]);
          o[1] = clamp8(a + ((cb_g[b] + cr_g[c]) >> 16));
          o[2] = clamp8(a + cb_b[b]);
        } else {
          o[0] = (uint8_t)a; o[1] = (uint8_t)b; o[2] = (uint8_t)c;
        }
        o[3] = 255;
      }
  }

  void run(uint8_t** out, int32_t* w, int32_t* h) {
    if (len < 4 || data[0] != 0xFF || data[1] != 0xD8) fail("Not a JPEG file.");
    size_t at = 2;
    bool done = false, any_scan = false;
    while (!done) {
      while (at < len && data[at] != 0xFF) ++at;               // garbage before a marker is skipped
      while (at < len && data[at] == 0xFF) ++at;               // fill bytes
      if (at >= len) break;
      const int marker = data[at++];
      if (marker == 0xD9) { done = true; break; }
      if (marker == 0x01 || (marker >= 0xD0 && marker <= 0xD7)) continue;
      const size_t seg_len = be16(at);
      if (seg_len < 2 || at + seg_len > len) fail("Truncated JPEG segment.");
      const size_t body = at + 2, seg_end = at + seg_len;
      switch (marker) {
        case 0xDB: read_dqt(body, seg_end); break;
        case 0xC4: read_dht(body, seg_end); break;
        case 0xDD: if (seg_len >= 4) restart_interval = be16(body); break;
        case 0xE0: if (seg_len >= 7 && std::memcmp(data + body, "JFIF", 5) == 0) saw_jfif = true; break;
        case 0xEE:
          if (seg_len >= 14 && std::memcmp(data + body, "Adobe", 5) == 0) { saw_adobe = true; adobe_transform = data[body + 11]; }
          break;
        case 0xDA: at = read_scan(body, seg_end); any_scan = true; continue;
        default:
          if ((marker >= 0xC0 && marker <= 0xCF) && marker != 0xC4 && marker != 0xC8 && marker != 0xCC) read_sof(body, seg_end, marker);
          break;
      }
      at = seg_end;
    }
    if (!ncomp || !any_scan) fail("JPEG file has no image data.");
    inverse_transform();
    uint8_t* rgba = static_cast<uint8_t*>(std::malloc((size_t)width * height * 4));
    if (!rgba) fail("Out of memory decoding a %dx%d image.", width, height);
    try {
      to_rgba(rgba);
    } catch (...) {
      std::free(rgba);
      throw;
    }
    *out = rgba; *w = width; *h = height;
  }
};

}  // namespace

extern "C" {

// Decodes a JPEG byte stream to RGBA8 (alpha 255), top row first.  On success returns 0 and a
// buffer to release with pbr_host_free; on failure returns -1 with a message in err.
int pbr_host_jpeg_decode(const uint8_t* data, size_t len, uint8_t** rgba_out, int32_t* w, int32_t* h, char* err, size_t err_len) {
  if (err && err_len) err[0] = 0;
  if (!data || !rgba_out || !w || !h) { if (err && err_len) snprintf(err, err_len, "null argument"); return -1; }
  try {
    Decoder d;
    d.data = data; d.len = len;
    std::memset(d.quant, 0, sizeof(d.quant));
    d.run(rgba_out, w, h);
    return 0;
  } catch (const DecodeError& e) {
    if (err && err_len) snprintf(err, err_len, "%s", e.msg);
  } catch (const std::exception& e) {
    if (err && err_len) snprintf(err, err_len, "%s", e.what());
  }
  return -1;
}

void pbr_host_free(void* p) { std::free(p); }

}  // extern "C"
