// yt_jpeg.h — JPEG textures for ythip_load_scene (host code).
//
// What the reference gets from load_texture for a .jpg / .jpeg (yocto_sceneio.cpp:1832-1844): stb_image's
// stbi_load_from_memory(..., 4), i.e. RGBA8.  A JPEG decoder's output is not defined by the file alone — the inverse DCT,
// the chroma upsampling filter and the YCbCr -> RGB arithmetic are the decoder's — so this reader restates stb_image's
// choices (stb_image.h of the reference's vendored copy; its SSE2 kernels are built to match its scalar ones bit for bit,
// and the scalar forms are what is restated here):
//
//   * baseline and progressive Huffman JPEG, 8 bit, 1 / 3 / 4 components, restart intervals, fill bytes (:2052-2066,
//     :2182-2400, :2915-3062, :3065-3400);
//   * the integer "islow" inverse DCT with 12-bit constants, two extra bits kept between the passes (:2408-2500);
//   * chroma upsampling: 2x horizontally / vertically / both with the 3:1 tent filter centred as JFIF sites the samples,
//     any other ratio by repetition (:3402-3600);
//   * YCbCr -> RGB in 20-bit fixed point with constants rounded to 12 bits, the Cb term of green truncated to its high
//     16 bits (:3604-3632); Adobe APP14 transform 0 = RGB / CMYK as they are, 2 = YCCK; component ids 'R','G','B' = RGB;
//   * a stream that runs into a marker or the end of the file decodes zeros from there on; a file without its EOI marker
//     is refused, as stb_image refuses it (:3355-3397).  Blocks a scan never reaches are blocks of zero coefficients here;
//     stb_image leaves their samples as malloc returned them.
//
// tests/test_sceneio.py compares every pixel with the reference's loader on the reference's own JPEG files and on files
// written by PIL in every sampling / progressive / restart / colour-space combination it offers.
#pragma once

#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace ytjpeg {

struct Info {
  int width = 0, height = 0, components = 0;
};

namespace detail {
constexpr int FAST_BITS = 9;

struct Reader {
  const uint8_t *p, *end;
  int  get8() { return p < end ? *p++ : 0; }
  int  get16() {
    const int hi = get8();
    return (hi << 8) | get8();
  }
  bool eof() const { return p >= end; }
  void skip(int n) {
    if (n < 0 || (size_t)(end - p) < (size_t)n) p = end;
    else p += n;
  }
};

struct Huffman {
  uint8_t  fast[1 << FAST_BITS];
  uint16_t code[256];
  uint8_t  values[256];
  uint8_t  size[257];
  uint32_t maxcode[18];
  int      delta[17];
};

// zig-zag position -> row-major position; 15 more entries so that a corrupt run cannot leave the block
static const uint8_t dezigzag[64 + 15] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20,
    13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60,
    61, 54, 47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

inline bool build_huffman(Huffman& h, const int* count) {
  int k = 0;
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < count[i]; j++) {
      if (k >= 256) return false;
      h.size[k++] = (uint8_t)(i + 1);
    }
  h.size[k] = 0;
  unsigned code = 0;
  k             = 0;
  int j;
  for (j = 1; j <= 16; j++) {
    h.delta[j] = k - (int)code;
    if (h.size[k] == j) {
      while (h.size[k] == j) h.code[k++] = (uint16_t)(code++);
      if (code - 1 >= (1u << j)) return false;
    }
    h.maxcode[j] = code << (16 - j);
    code <<= 1;
  }
  h.maxcode[j] = 0xffffffffu;
  std::memset(h.fast, 255, sizeof(h.fast));
  for (int i = 0; i < k; i++) {
    const int s = h.size[i];
    if (s <= FAST_BITS) {
      const int c = h.code[i] << (FAST_BITS - s), m = 1 << (FAST_BITS - s);
      for (int q = 0; q < m; q++) h.fast[c + q] = (uint8_t)i;
    }
  }
  return true;
}
// run, size and VALUE of a small AC coefficient in one lookup
inline void build_fast_ac(int16_t* fast_ac, const Huffman& h) {
  for (int i = 0; i < (1 << FAST_BITS); i++) {
    const uint8_t fast = h.fast[i];
    fast_ac[i]         = 0;
    if (fast == 255) continue;
    const int rs = h.values[fast], run = (rs >> 4) & 15, magbits = rs & 15, len = h.size[fast];
    if (magbits && len + magbits <= FAST_BITS) {
      int       k = ((i << len) & ((1 << FAST_BITS) - 1)) >> (FAST_BITS - magbits);
      const int m = 1 << (magbits - 1);
      if (k < m) k += (int)((~0u << magbits) + 1);
      if (k >= -128 && k <= 127) fast_ac[i] = (int16_t)((k * 256) + (run * 16) + (len + magbits));
    }
  }
}

struct Component {
  int id = 0, h = 0, v = 0, tq = 0, hd = 0, ha = 0, dc_pred = 0;
  int x = 0, y = 0, w2 = 0, h2 = 0, coeff_w = 0;
  std::vector<uint8_t> data;
  std::vector<int16_t> coeff;  // progressive only
};

inline uint8_t clamp8(int x) { return (unsigned)x > 255 ? (x < 0 ? 0 : 255) : (uint8_t)x; }

// one pass of the integer inverse DCT (constants scaled by 4096).  64-bit intermediates: the coefficients of a valid stream
// never leave 32 bits (the results are stb_image's), those of a damaged one cannot overflow here.
#define YTJ_F2F(x) ((int64_t)(((x)*4096 + 0.5)))
#define YTJ_IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                                            \
  int64_t t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;                                  \
  p2 = s2, p3 = s6;                                                                            \
  p1 = (p2 + p3) * YTJ_F2F(0.5411961f);                                                        \
  t2 = p1 + p3 * YTJ_F2F(-1.847759065f);                                                       \
  t3 = p1 + p2 * YTJ_F2F(0.765366865f);                                                        \
  p2 = s0, p3 = s4;                                                                            \
  t0 = (p2 + p3) * 4096, t1 = (p2 - p3) * 4096;                                                \
  x0 = t0 + t3, x3 = t0 - t3, x1 = t1 + t2, x2 = t1 - t2;                                      \
  t0 = s7, t1 = s5, t2 = s3, t3 = s1;                                                          \
  p3 = t0 + t2, p4 = t1 + t3, p1 = t0 + t3, p2 = t1 + t2;                                      \
  p5 = (p3 + p4) * YTJ_F2F(1.175875602f);                                                      \
  t0 = t0 * YTJ_F2F(0.298631336f), t1 = t1 * YTJ_F2F(2.053119869f);                            \
  t2 = t2 * YTJ_F2F(3.072711026f), t3 = t3 * YTJ_F2F(1.501321110f);                            \
  p1 = p5 + p1 * YTJ_F2F(-0.899976223f), p2 = p5 + p2 * YTJ_F2F(-2.562915447f);                \
  p3 = p3 * YTJ_F2F(-1.961570560f), p4 = p4 * YTJ_F2F(-0.390180644f);                          \
  t3 += p1 + p4, t2 += p2 + p3, t1 += p2 + p4, t0 += p1 + p3;

inline void idct_block(uint8_t* out, int stride, const int16_t* d) {
  int64_t val[64], *v = val;
  for (int i = 0; i < 8; i++, d++, v++) {  // columns
    if (d[8] == 0 && d[16] == 0 && d[24] == 0 && d[32] == 0 && d[40] == 0 && d[48] == 0 && d[56] == 0) {
      v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = (int64_t)d[0] * 4;
    } else {
      YTJ_IDCT_1D(d[0], d[8], d[16], d[24], d[32], d[40], d[48], d[56])
      x0 += 512, x1 += 512, x2 += 512, x3 += 512;  // 12 bits of constants down to 2 bits of extra precision
      v[0] = (x0 + t3) >> 10, v[56] = (x0 - t3) >> 10, v[8] = (x1 + t2) >> 10, v[48] = (x1 - t2) >> 10;
      v[16] = (x2 + t1) >> 10, v[40] = (x2 - t1) >> 10, v[24] = (x3 + t0) >> 10, v[32] = (x3 - t0) >> 10;
    }
  }
  v = val;
  for (int i = 0; i < 8; i++, v += 8, out += stride) {  // rows: 12 + 2 + 3 bits to remove, + 128 to leave the signed range
    YTJ_IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
    x0 += 65536 + (128 << 17), x1 += 65536 + (128 << 17), x2 += 65536 + (128 << 17), x3 += 65536 + (128 << 17);
    auto c8 = [](int64_t x) { return x < 0 ? (uint8_t)0 : x > 255 ? (uint8_t)255 : (uint8_t)x; };
    out[0] = c8((x0 + t3) >> 17), out[7] = c8((x0 - t3) >> 17), out[1] = c8((x1 + t2) >> 17), out[6] = c8((x1 - t2) >> 17);
    out[2] = c8((x2 + t1) >> 17), out[5] = c8((x2 - t1) >> 17), out[3] = c8((x3 + t0) >> 17), out[4] = c8((x3 - t0) >> 17);
  }
}
#undef YTJ_IDCT_1D
#undef YTJ_F2F

struct Decoder {
  Reader    s{nullptr, nullptr};
  Huffman   huff_dc[4], huff_ac[4];
  uint16_t  dequant[4][64];
  int16_t   fast_ac[4][1 << FAST_BITS];
  Component comp[4];
  int       img_x = 0, img_y = 0, img_n = 0;
  int       h_max = 1, v_max = 1, mcu_x = 0, mcu_y = 0;
  uint32_t  code_buffer = 0;
  int       code_bits = 0, nomore = 0;
  uint8_t   marker = 0xff;  // 0xff = none cached
  int       progressive = 0, spec_start = 0, spec_end = 0, succ_high = 0, succ_low = 0, eob_run = 0;
  int       jfif = 0, app14 = -1, rgb = 0;
  int       scan_n = 0, order[4] = {0, 0, 0, 0};
  int       restart_interval = 0, todo = 0;
  std::string why;

  bool fail(const char* m) {
    if (why.empty()) why = m;
    return false;
  }

  // ---- entropy-coded bits: a 32-bit window, refilled a byte at a time; a marker (or the end) turns the rest into zeros
  void grow() {
    do {
      const unsigned b = nomore ? 0 : (unsigned)s.get8();
      if (b == 0xff) {
        int c = s.get8();
        while (c == 0xff) c = s.get8();
        if (c != 0) {
          marker = (uint8_t)c, nomore = 1;
          return;
        }
      }
      code_buffer |= b << (24 - code_bits);
      code_bits += 8;
    } while (code_bits <= 24);
  }
  int huff_decode(const Huffman& h) {
    if (code_bits < 16) grow();
    int c = (code_buffer >> (32 - FAST_BITS)) & ((1 << FAST_BITS) - 1), k = h.fast[c];
    if (k < 255) {
      const int sz = h.size[k];
      if (sz > code_bits) return -1;
      code_buffer <<= sz, code_bits -= sz;
      return h.values[k];
    }
    const unsigned temp = code_buffer >> 16;
    for (k = FAST_BITS + 1;; k++)
      if (temp < h.maxcode[k]) break;
    if (k == 17) {
      code_bits -= 16;
      return -1;
    }
    if (k > code_bits) return -1;
    c = (int)((code_buffer >> (32 - k)) & ((1u << k) - 1)) + h.delta[k];
    if (c < 0 || c > 255) return -1;
    code_bits -= k, code_buffer <<= k;
    return h.values[c];
  }
  static uint32_t rotl(uint32_t x, int n) { return n ? (x << n) | (x >> (32 - n)) : x; }
  int extend_receive(int n) {  // n bits, sign-extended as JPEG's EXTEND does
    if (code_bits < n) grow();
    const int      sgn = (int)(code_buffer >> 31);
    const uint32_t m   = (1u << n) - 1;
    uint32_t       k   = rotl(code_buffer, n);
    code_buffer        = k & ~m;
    k &= m;
    code_bits -= n;
    static const int bias[16] = {0, -1, -3, -7, -15, -31, -63, -127, -255, -511, -1023, -2047, -4095, -8191, -16383, -32767};
    return (int)k + (bias[n] & (sgn - 1));
  }
  int get_bits(int n) {
    if (code_bits < n) grow();
    const uint32_t m = (1u << n) - 1;
    uint32_t       k = rotl(code_buffer, n);
    code_buffer      = k & ~m;
    k &= m;
    code_bits -= n;
    return (int)k;
  }
  int get_bit() {
    if (code_bits < 1) grow();
    const uint32_t k = code_buffer;
    code_buffer <<= 1, --code_bits;
    return (int)(k & 0x80000000u);
  }

  // ---- blocks ---------------------------------------------------------------------------------------------------------
  bool decode_block(int16_t* data, const Huffman& hdc, const Huffman& hac, const int16_t* fac, int b, const uint16_t* dq) {
    if (code_bits < 16) grow();
    const int t = huff_decode(hdc);
    if (t < 0 || t > 15) return fail("bad huffman code");
    std::memset(data, 0, 64 * sizeof(int16_t));
    const int diff = t ? extend_receive(t) : 0, dc = (int)((unsigned)comp[b].dc_pred + (unsigned)diff);
    comp[b].dc_pred = dc;
    data[0]         = (int16_t)((unsigned)dc * (unsigned)dq[0]);
    int k           = 1;
    do {
      if (code_bits < 16) grow();
      const int c = (code_buffer >> (32 - FAST_BITS)) & ((1 << FAST_BITS) - 1);
      int       r = fac[c], sz;
      if (r) {
        k += (r >> 4) & 15, sz = r & 15;
        code_buffer <<= sz, code_bits -= sz;
        const unsigned zig = dezigzag[k++];
        data[zig]          = (int16_t)((r >> 8) * dq[zig]);
      } else {
        const int rs = huff_decode(hac);
        if (rs < 0) return fail("bad huffman code");
        sz = rs & 15, r = rs >> 4;
        if (sz == 0) {
          if (rs != 0xf0) break;
          k += 16;
        } else {
          k += r;
          const unsigned zig = dezigzag[k++];
          data[zig]          = (int16_t)(extend_receive(sz) * dq[zig]);
        }
      }
    } while (k < 64);
    return true;
  }
  bool decode_block_prog_dc(int16_t* data, const Huffman& hdc, int b) {
    if (spec_end != 0) return fail("can't merge dc and ac");
    if (code_bits < 16) grow();
    if (succ_high == 0) {
      std::memset(data, 0, 64 * sizeof(int16_t));
      const int t = huff_decode(hdc);
      if (t < 0 || t > 15) return fail("can't merge dc and ac");
      const int diff = t ? extend_receive(t) : 0, dc = (int)((unsigned)comp[b].dc_pred + (unsigned)diff);
      comp[b].dc_pred = dc;
      data[0]         = (int16_t)((unsigned)dc << succ_low);
    } else if (get_bit()) data[0] = (int16_t)(data[0] + (int16_t)(1 << succ_low));
    return true;
  }
  bool decode_block_prog_ac(int16_t* data, const Huffman& hac, const int16_t* fac) {
    if (spec_start == 0) return fail("can't merge dc and ac");
    if (succ_high == 0) {
      const int shift = succ_low;
      if (eob_run) return --eob_run, true;
      int k = spec_start;
      do {
        if (code_bits < 16) grow();
        const int c = (code_buffer >> (32 - FAST_BITS)) & ((1 << FAST_BITS) - 1);
        int       r = fac[c], sz;
        if (r) {
          k += (r >> 4) & 15, sz = r & 15;
          code_buffer <<= sz, code_bits -= sz;
          const unsigned zig = dezigzag[k++];
          data[zig]          = (int16_t)((r >> 8) * (1 << shift));
        } else {
          const int rs = huff_decode(hac);
          if (rs < 0) return fail("bad huffman code");
          sz = rs & 15, r = rs >> 4;
          if (sz == 0) {
            if (r < 15) {
              eob_run = 1 << r;
              if (r) eob_run += get_bits(r);
              --eob_run;
              break;
            }
            k += 16;
          } else {
            k += r;
            const unsigned zig = dezigzag[k++];
            data[zig]          = (int16_t)(extend_receive(sz) * (1 << shift));
          }
        }
      } while (k <= spec_end);
    } else {  // refinement: one more bit for the coefficients already non-zero, new ones arrive as +-1 << succ_low
      const int16_t bit = (int16_t)(1 << succ_low);
      auto refine = [&](int16_t* p) {
        if (get_bit())
          if ((*p & bit) == 0) *p = (int16_t)(*p > 0 ? *p + bit : *p - bit);
      };
      if (eob_run) {
        --eob_run;
        for (int k = spec_start; k <= spec_end; k++) {
          int16_t* p = &data[dezigzag[k]];
          if (*p != 0) refine(p);
        }
      } else {
        int k = spec_start;
        do {
          const int rs = huff_decode(hac);
          if (rs < 0) return fail("bad huffman code");
          int sz = rs & 15, r = rs >> 4;
          if (sz == 0) {
            if (r < 15) {
              eob_run = (1 << r) - 1;
              if (r) eob_run += get_bits(r);
              r = 64;  // to the end of the band
            }
          } else {
            if (sz != 1) return fail("bad huffman code");
            sz = get_bit() ? bit : -bit;
          }
          while (k <= spec_end) {
            int16_t* p = &data[dezigzag[k++]];
            if (*p != 0) refine(p);
            else {
              if (r == 0) {
                *p = (int16_t)sz;
                break;
              }
              --r;
            }
          }
        } while (k <= spec_end);
      }
    }
    return true;
  }

  // ---- markers --------------------------------------------------------------------------------------------------------
  uint8_t get_marker() {
    if (marker != 0xff) {
      const uint8_t x = marker;
      marker          = 0xff;
      return x;
    }
    int x = s.get8();
    if (x != 0xff) return 0xff;
    while (x == 0xff) x = s.get8();
    return (uint8_t)x;
  }
  void reset() {
    code_bits = 0, code_buffer = 0, nomore = 0;
    for (auto& c : comp) c.dc_pred = 0;
    marker  = 0xff;
    todo    = restart_interval ? restart_interval : 0x7fffffff;
    eob_run = 0;
  }
  // after every MCU: false = the scan ends here (no restart marker where one was due: the rest stays as it is)
  bool count_down() {
    if (--todo <= 0) {
      if (code_bits < 24) grow();
      if (!(marker >= 0xd0 && marker <= 0xd7)) return false;
      reset();
    }
    return true;
  }
  bool parse_entropy_coded_data() {
    reset();
    int16_t block[64];
    if (scan_n == 1) {  // one component: its blocks in raster order, however the frame interleaves
      const int n = order[0], w = (comp[n].x + 7) >> 3, h = (comp[n].y + 7) >> 3;
      for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++) {
          if (!progressive) {
            const int ha = comp[n].ha;
            if (!decode_block(block, huff_dc[comp[n].hd], huff_ac[ha], fast_ac[ha], n, dequant[comp[n].tq])) return false;
            idct_block(comp[n].data.data() + (size_t)comp[n].w2 * j * 8 + i * 8, comp[n].w2, block);
          } else {
            int16_t* data = comp[n].coeff.data() + 64 * ((size_t)i + (size_t)j * comp[n].coeff_w);
            if (spec_start == 0 ? !decode_block_prog_dc(data, huff_dc[comp[n].hd], n)
                                : !decode_block_prog_ac(data, huff_ac[comp[n].ha], fast_ac[comp[n].ha]))
              return false;
          }
          if (!count_down()) return true;
        }
      return true;
    }
    for (int j = 0; j < mcu_y; j++)
      for (int i = 0; i < mcu_x; i++) {
        for (int k = 0; k < scan_n; k++) {
          const int n = order[k];
          for (int y = 0; y < comp[n].v; y++)
            for (int x = 0; x < comp[n].h; x++) {
              const int bx = i * comp[n].h + x, by = j * comp[n].v + y;
              if (!progressive) {
                const int ha = comp[n].ha;
                if (!decode_block(block, huff_dc[comp[n].hd], huff_ac[ha], fast_ac[ha], n, dequant[comp[n].tq])) return false;
                idct_block(comp[n].data.data() + (size_t)comp[n].w2 * by * 8 + bx * 8, comp[n].w2, block);
              } else {
                int16_t* data = comp[n].coeff.data() + 64 * ((size_t)bx + (size_t)by * comp[n].coeff_w);
                if (!decode_block_prog_dc(data, huff_dc[comp[n].hd], n)) return false;
              }
            }
        }
        if (!count_down()) return true;
      }
    return true;
  }
  bool process_marker(int m) {
    int L;
    switch (m) {
      case 0xff: return fail("expected marker");
      case 0xDD:
        if (s.get16() != 4) return fail("bad DRI len");
        restart_interval = s.get16();
        return true;
      case 0xDB:
        L = s.get16() - 2;
        while (L > 0) {
          const int q = s.get8(), p = q >> 4, t = q & 15;
          if (p != 0 && p != 1) return fail("bad DQT type");
          if (t > 3) return fail("bad DQT table");
          for (int i = 0; i < 64; i++) dequant[t][dezigzag[i]] = (uint16_t)(p ? s.get16() : s.get8());
          L -= p ? 129 : 65;
        }
        return L == 0 ? true : fail("bad DQT len");
      case 0xC4:
        L = s.get16() - 2;
        while (L > 0) {
          int       sizes[16], n = 0;
          const int q = s.get8(), tc = q >> 4, th = q & 15;
          if (tc > 1 || th > 3) return fail("bad DHT header");
          for (int i = 0; i < 16; i++) sizes[i] = s.get8(), n += sizes[i];
          if (n > 256) return fail("bad DHT header");
          L -= 17;
          Huffman& h = tc == 0 ? huff_dc[th] : huff_ac[th];
          if (!build_huffman(h, sizes)) return fail("bad code lengths");
          for (int i = 0; i < n; i++) h.values[i] = (uint8_t)s.get8();
          if (tc != 0) build_fast_ac(fast_ac[th], h);
          L -= n;
        }
        return L == 0 ? true : fail("bad DHT len");
      default: break;
    }
    if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) {
      L = s.get16();
      if (L < 2) return fail(m == 0xFE ? "bad COM len" : "bad APP len");
      L -= 2;
      if (m == 0xE0 && L >= 5) {  // JFIF
        static const uint8_t tag[5] = {'J', 'F', 'I', 'F', 0};
        bool                 ok     = true;
        for (int i = 0; i < 5; i++) ok = (s.get8() == tag[i]) && ok;
        L -= 5;
        if (ok) jfif = 1;
      } else if (m == 0xEE && L >= 12) {  // Adobe: the colour transform
        static const uint8_t tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
        bool                 ok     = true;
        for (int i = 0; i < 6; i++) ok = (s.get8() == tag[i]) && ok;
        L -= 6;
        if (ok) {
          s.get8(), s.get16(), s.get16();
          app14 = s.get8();
          L -= 6;
        }
      }
      s.skip(L);
      return true;
    }
    return fail("unknown marker");
  }
  bool process_scan_header() {
    const int Ls = s.get16();
    scan_n       = s.get8();
    if (scan_n < 1 || scan_n > 4 || scan_n > img_n) return fail("bad SOS component count");
    if (Ls != 6 + 2 * scan_n) return fail("bad SOS len");
    for (int i = 0; i < scan_n; i++) {
      const int id = s.get8(), q = s.get8();
      int       which = 0;
      for (; which < img_n; which++)
        if (comp[which].id == id) break;
      if (which == img_n) return fail("bad SOS component");
      comp[which].hd = q >> 4, comp[which].ha = q & 15;
      if (comp[which].hd > 3) return fail("bad DC huff");
      if (comp[which].ha > 3) return fail("bad AC huff");
      order[i] = which;
    }
    spec_start   = s.get8();
    spec_end     = s.get8();
    const int aa = s.get8();
    succ_high = aa >> 4, succ_low = aa & 15;
    if (progressive) {
      if (spec_start > 63 || spec_end > 63 || spec_start > spec_end || succ_high > 13 || succ_low > 13) return fail("bad SOS");
    } else {
      if (spec_start != 0 || succ_high != 0 || succ_low != 0) return fail("bad SOS");
      spec_end = 63;
    }
    return true;
  }
  bool process_frame_header(bool load) {
    const int Lf = s.get16();
    if (Lf < 11) return fail("bad SOF len");
    if (s.get8() != 8) return fail("only 8-bit JPEG is read (as by the reference)");
    img_y = s.get16();
    if (img_y == 0) return fail("no header height (delayed height is not supported, as in the reference)");
    img_x = s.get16();
    if (img_x == 0) return fail("0 width");
    const int c = s.get8();
    if (c != 3 && c != 1 && c != 4) return fail("bad component count");
    img_n = c;
    if (Lf != 8 + 3 * img_n) return fail("bad SOF len");
    rgb = 0;
    for (int i = 0; i < img_n; i++) {
      static const uint8_t names[3] = {'R', 'G', 'B'};
      comp[i].id                    = s.get8();
      if (img_n == 3 && comp[i].id == names[i]) ++rgb;
      const int q = s.get8();
      comp[i].h = q >> 4, comp[i].v = q & 15;
      if (!comp[i].h || comp[i].h > 4) return fail("bad H");
      if (!comp[i].v || comp[i].v > 4) return fail("bad V");
      comp[i].tq = s.get8();
      if (comp[i].tq > 3) return fail("bad TQ");
    }
    if ((double)img_x * img_y * 4.0 > 2147483647.0) return fail("image too large to decode");
    h_max = v_max = 1;
    for (int i = 0; i < img_n; i++) h_max = comp[i].h > h_max ? comp[i].h : h_max, v_max = comp[i].v > v_max ? comp[i].v : v_max;
    for (int i = 0; i < img_n; i++) {
      if (h_max % comp[i].h != 0) return fail("bad H");
      if (v_max % comp[i].v != 0) return fail("bad V");
    }
    if (!load) return true;
    mcu_x = (img_x + h_max * 8 - 1) / (h_max * 8), mcu_y = (img_y + v_max * 8 - 1) / (v_max * 8);
    for (int i = 0; i < img_n; i++) {
      comp[i].x  = (img_x * comp[i].h + h_max - 1) / h_max;
      comp[i].y  = (img_y * comp[i].v + v_max - 1) / v_max;
      comp[i].w2 = mcu_x * comp[i].h * 8, comp[i].h2 = mcu_y * comp[i].v * 8;
      comp[i].data.assign((size_t)comp[i].w2 * comp[i].h2, 0);
      if (progressive) {
        comp[i].coeff_w = comp[i].w2 / 8;
        comp[i].coeff.assign((size_t)comp[i].w2 * comp[i].h2, 0);
      }
    }
    return true;
  }
  bool decode_header(bool load) {
    jfif = 0, app14 = -1, marker = 0xff;
    if (get_marker() != 0xd8) return fail("no SOI");
    int m = get_marker();
    while (!(m == 0xc0 || m == 0xc1 || m == 0xc2)) {
      if (!process_marker(m)) return false;
      m = get_marker();
      while (m == 0xff) {  // padding after a segment
        if (s.eof()) return fail("no SOF");
        m = get_marker();
      }
    }
    progressive = m == 0xc2;
    return process_frame_header(load);
  }
  bool decode_image() {
    restart_interval = 0;
    if (!decode_header(true)) return false;
    int m = get_marker();
    while (m != 0xd9) {
      if (m == 0xda) {
        if (!process_scan_header()) return false;
        if (!parse_entropy_coded_data()) return false;
        if (marker == 0xff) {  // stray bytes after the scan: on to the next 0xff
          while (!s.eof()) {
            if (s.get8() == 255) {
              marker = (uint8_t)s.get8();
              break;
            }
          }
        }
      } else if (m == 0xdc) {
        const int Ld = s.get16(), NL = s.get16();
        if (Ld != 4) return fail("bad DNL len");
        if (NL != img_y) return fail("bad DNL height");
      } else if (!process_marker(m)) return false;
      m = get_marker();
    }
    if (progressive)
      for (int n = 0; n < img_n; n++) {
        const int w = (comp[n].x + 7) >> 3, h = (comp[n].y + 7) >> 3;
        for (int j = 0; j < h; j++)
          for (int i = 0; i < w; i++) {
            int16_t* data = comp[n].coeff.data() + 64 * ((size_t)i + (size_t)j * comp[n].coeff_w);
            for (int q = 0; q < 64; q++) data[q] = (int16_t)((unsigned)(int)data[q] * (unsigned)dequant[comp[n].tq][q]);
            idct_block(comp[n].data.data() + (size_t)comp[n].w2 * j * 8 + i * 8, comp[n].w2, data);
          }
      }
    return true;
  }
};

// ---- upsampling: one output row from the two nearest input rows ----------------------------------------------------------
inline uint8_t div4(int x) { return (uint8_t)(x >> 2); }
inline uint8_t div16(int x) { return (uint8_t)(x >> 4); }
inline const uint8_t* resample_v2(uint8_t* out, const uint8_t* nearr, const uint8_t* farr, int w) {
  for (int i = 0; i < w; i++) out[i] = div4(3 * nearr[i] + farr[i] + 2);
  return out;
}
inline const uint8_t* resample_h2(uint8_t* out, const uint8_t* in, int w) {
  if (w == 1) return out[0] = out[1] = in[0], out;
  out[0] = in[0];
  out[1] = div4(in[0] * 3 + in[1] + 2);
  int i;
  for (i = 1; i < w - 1; i++) {
    const int n    = 3 * in[i] + 2;
    out[i * 2 + 0] = div4(n + in[i - 1]);
    out[i * 2 + 1] = div4(n + in[i + 1]);
  }
  out[i * 2 + 0] = div4(in[w - 2] * 3 + in[w - 1] + 2);
  out[i * 2 + 1] = in[w - 1];
  return out;
}
inline const uint8_t* resample_hv2(uint8_t* out, const uint8_t* nearr, const uint8_t* farr, int w) {
  if (w == 1) return out[0] = out[1] = div4(3 * nearr[0] + farr[0] + 2), out;
  int t1 = 3 * nearr[0] + farr[0], t0;
  out[0] = div4(t1 + 2);
  for (int i = 1; i < w; i++) {
    t0 = t1, t1 = 3 * nearr[i] + farr[i];
    out[i * 2 - 1] = div16(3 * t0 + t1 + 8);
    out[i * 2]     = div16(3 * t1 + t0 + 8);
  }
  out[w * 2 - 1] = div4(t1 + 2);
  return out;
}
inline const uint8_t* resample_repeat(uint8_t* out, const uint8_t* in, int w, int hs) {
  for (int i = 0; i < w; i++)
    for (int j = 0; j < hs; j++) out[i * hs + j] = in[i];
  return out;
}

inline void ycbcr_to_rgba(uint8_t* out, const uint8_t* y, const uint8_t* pcb, const uint8_t* pcr, int count) {
  const int c_r = ((int)(1.40200f * 4096.0f + 0.5f)) << 8, c_gr = ((int)(0.71414f * 4096.0f + 0.5f)) << 8,
            c_gb = ((int)(0.34414f * 4096.0f + 0.5f)) << 8, c_b = ((int)(1.77200f * 4096.0f + 0.5f)) << 8;
  for (int i = 0; i < count; i++, out += 4) {
    const int y_fixed = (y[i] << 20) + (1 << 19), cr = pcr[i] - 128, cb = pcb[i] - 128;
    int       r = y_fixed + cr * c_r;
    int       g = (int)((uint32_t)(y_fixed + cr * -c_gr) + ((uint32_t)(cb * -c_gb) & 0xffff0000u));
    int       b = y_fixed + cb * c_b;
    r >>= 20, g >>= 20, b >>= 20;
    out[0] = clamp8(r), out[1] = clamp8(g), out[2] = clamp8(b), out[3] = 255;
  }
}
inline uint8_t blinn_8x8(uint8_t x, uint8_t y) {
  const unsigned t = (unsigned)x * y + 128;
  return (uint8_t)((t + (t >> 8)) >> 8);
}
}  // namespace detail

// size and component count, everything the frame header can be refused for
inline bool header(const uint8_t* data, size_t size, Info& info, std::string& why) {
  auto d = std::make_unique<detail::Decoder>();
  d->s   = {data, data + size};
  if (!d->decode_header(false)) return why = "corrupt JPEG: " + d->why, false;
  info = {d->img_x, d->img_y, d->img_n};
  return true;
}

// RGBA8, width * height * 4 bytes
inline bool decode(const uint8_t* data, size_t size, uint8_t* out, std::string& why) {
  using namespace detail;
  auto zp = std::make_unique<Decoder>();
  auto& z = *zp;
  z.s     = {data, data + size};
  if (!z.decode_image()) return why = "corrupt JPEG: " + z.why, false;
  const bool is_rgb = z.img_n == 3 && (z.rgb == 3 || (z.app14 == 0 && !z.jfif));
  struct Resample {
    int            hs, vs, w_lores, ystep, ypos;
    const uint8_t *line0, *line1;
    std::vector<uint8_t> linebuf;
  } res[4];
  for (int k = 0; k < z.img_n; k++) {
    auto& r = res[k];
    r.hs = z.h_max / z.comp[k].h, r.vs = z.v_max / z.comp[k].v;
    r.ystep = r.vs >> 1, r.w_lores = (z.img_x + r.hs - 1) / r.hs, r.ypos = 0;
    r.line0 = r.line1 = z.comp[k].data.data();
    r.linebuf.assign((size_t)z.img_x + 3 + 8, 0);
  }
  const uint8_t* co[4] = {nullptr, nullptr, nullptr, nullptr};
  for (int j = 0; j < z.img_y; j++) {
    uint8_t* o = out + (size_t)4 * z.img_x * j;
    for (int k = 0; k < z.img_n; k++) {
      auto&          r     = res[k];
      const bool     y_bot = r.ystep >= (r.vs >> 1);
      const uint8_t *nearr = y_bot ? r.line1 : r.line0, *farr = y_bot ? r.line0 : r.line1;
      if (r.hs == 1 && r.vs == 1) co[k] = nearr;
      else if (r.hs == 1 && r.vs == 2) co[k] = resample_v2(r.linebuf.data(), nearr, farr, r.w_lores);
      else if (r.hs == 2 && r.vs == 1) co[k] = resample_h2(r.linebuf.data(), nearr, r.w_lores);
      else if (r.hs == 2 && r.vs == 2) co[k] = resample_hv2(r.linebuf.data(), nearr, farr, r.w_lores);
      else co[k] = resample_repeat(r.linebuf.data(), nearr, r.w_lores, r.hs);
      if (++r.ystep >= r.vs) {
        r.ystep = 0, r.line0 = r.line1;
        if (++r.ypos < z.comp[k].y) r.line1 += z.comp[k].w2;
      }
    }
    if (z.img_n == 3) {
      if (is_rgb)
        for (int i = 0; i < z.img_x; i++) o[4 * i] = co[0][i], o[4 * i + 1] = co[1][i], o[4 * i + 2] = co[2][i], o[4 * i + 3] = 255;
      else ycbcr_to_rgba(o, co[0], co[1], co[2], z.img_x);
    } else if (z.img_n == 4) {
      if (z.app14 == 0) {  // CMYK
        for (int i = 0; i < z.img_x; i++) {
          const uint8_t m = co[3][i];
          o[4 * i] = blinn_8x8(co[0][i], m), o[4 * i + 1] = blinn_8x8(co[1][i], m), o[4 * i + 2] = blinn_8x8(co[2][i], m), o[4 * i + 3] = 255;
        }
      } else if (z.app14 == 2) {  // YCCK
        ycbcr_to_rgba(o, co[0], co[1], co[2], z.img_x);
        for (int i = 0; i < z.img_x; i++) {
          const uint8_t m = co[3][i];
          o[4 * i] = blinn_8x8(255 - o[4 * i], m), o[4 * i + 1] = blinn_8x8(255 - o[4 * i + 1], m), o[4 * i + 2] = blinn_8x8(255 - o[4 * i + 2], m);
        }
      } else ycbcr_to_rgba(o, co[0], co[1], co[2], z.img_x);  // (a fourth channel of unknown meaning is ignored)
    } else {
      for (int i = 0; i < z.img_x; i++) o[4 * i] = o[4 * i + 1] = o[4 * i + 2] = co[0][i], o[4 * i + 3] = 255;
    }
  }
  return true;
}

}  // namespace ytjpeg
