// yt_bmptga.h — BMP and TGA textures for ythip_load_scene (host code).
//
// What the reference gets from load_texture for a .bmp / .tga (yocto_sceneio.cpp:1846-1873): stb_image's
// stbi_load_from_memory(..., 4), RGBA8.  Restated from what stb_image does with such files (stb_image.h of the reference's
// vendored copy: stbi__bmp_parse_header / stbi__bmp_load :5384-5655, stbi__tga_test / stbi__tga_load :5742-5995,
// stbi__convert_format :1760-1810), so that the same files give the same bytes:
//
//   * BMP: core (12-byte), V3 (40), 56-byte, V4 (108) and V5 (124) headers; 1 / 4 / 8-bit palettised, 16-bit (555 or
//     BI_BITFIELDS masks), 24-bit, 32-bit (masks, or the alpha byte — replaced by 255 everywhere when it is 0 everywhere);
//     bottom-up and top-down rows; RLE and embedded JPEG / PNG refused, as stb_image refuses them;
//   * TGA: types 1 / 2 / 3 and their run-length forms 9 / 10 / 11; 8-bit grey, 16-bit grey + alpha, 15 / 16-bit colour (five
//     bits per channel widened as (c * 255) / 31, no alpha), 24 / 32-bit; colour maps of 15 / 16 / 24 / 32 bits with 8- or
//     16-bit indices (an index past the map reads entry 0); either row order;
//   * bytes past the end of the file read as 0, as stb_image's reader returns them.
#pragma once

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace ytimg {

struct Bytes {  // stb_image's memory reader: past the end everything is 0
  const uint8_t *begin, *p, *end;
  Bytes(const uint8_t* d, size_t n) : begin(d), p(d), end(d + n) {}
  int      get8() { return p < end ? *p++ : 0; }
  int      get16le() {
    const int lo = get8();
    return lo | (get8() << 8);
  }
  uint32_t get32le() {
    const uint32_t lo = (uint32_t)get16le();
    return lo | ((uint32_t)get16le() << 16);
  }
  void skip(int n) {
    if (n == 0) return;
    if (n < 0 || (size_t)(end - p) < (size_t)n) p = end;
    else p += n;
  }
  bool getn(uint8_t* out, int n) {
    if (n < 0 || (size_t)(end - p) < (size_t)n) return false;
    std::memcpy(out, p, (size_t)n), p += n;
    return true;
  }
  size_t at() const { return (size_t)(p - begin); }
};

struct Size {
  int width = 0, height = 0;
};
constexpr int MAX_DIM = 1 << 24;
inline bool   fits(int64_t w, int64_t h, int64_t comp) { return w >= 0 && h >= 0 && w * h * comp <= 2147483647ll; }

// n-component rows -> RGBA (stbi__convert_format to 4 components)
inline void to_rgba(const uint8_t* src, int comp, size_t pixels, uint8_t* out) {
  for (size_t k = 0; k < pixels; k++, src += comp, out += 4) {
    switch (comp) {
      case 1: out[0] = out[1] = out[2] = src[0], out[3] = 255; break;
      case 2: out[0] = out[1] = out[2] = src[0], out[3] = src[1]; break;
      case 3: out[0] = src[0], out[1] = src[1], out[2] = src[2], out[3] = 255; break;
      default: out[0] = src[0], out[1] = src[1], out[2] = src[2], out[3] = src[3]; break;
    }
  }
}

// ---- BMP ----------------------------------------------------------------------------------------------------------------
namespace bmp {
struct Header {
  int      bpp = 0, offset = 0, hsz = 0, extra_read = 14, width = 0, height_signed = 0;
  uint32_t mr = 0, mg = 0, mb = 0, ma = 0, all_a = 255;
};
inline bool test(const uint8_t* d, size_t n) {
  Bytes s(d, n);
  if (s.get8() != 'B' || s.get8() != 'M') return false;
  s.get32le(), s.get16le(), s.get16le(), s.get32le();
  const int sz = (int)s.get32le();
  return sz == 12 || sz == 40 || sz == 56 || sz == 108 || sz == 124;
}
inline bool mask_defaults(Header& h, int compress) {
  if (compress == 3) return true;
  if (compress != 0) return false;
  if (h.bpp == 16) h.mr = 31u << 10, h.mg = 31u << 5, h.mb = 31u;
  else if (h.bpp == 32) h.mr = 0xffu << 16, h.mg = 0xffu << 8, h.mb = 0xffu, h.ma = 0xffu << 24, h.all_a = 0;
  else h.mr = h.mg = h.mb = h.ma = 0;
  return true;
}
inline bool parse(Bytes& s, Header& h, std::string& why) {
  if (s.get8() != 'B' || s.get8() != 'M') return why = "not BMP", false;
  s.get32le(), s.get16le(), s.get16le();
  h.offset = (int)s.get32le();
  h.hsz    = (int)s.get32le();
  if (h.offset < 0) return why = "bad BMP", false;
  const int hsz = h.hsz;
  if (hsz != 12 && hsz != 40 && hsz != 56 && hsz != 108 && hsz != 124) return why = "BMP type not supported: unknown header", false;
  if (hsz == 12) h.width = s.get16le(), h.height_signed = s.get16le();
  else h.width = (int)s.get32le(), h.height_signed = (int)s.get32le();
  if (s.get16le() != 1) return why = "bad BMP", false;
  h.bpp = s.get16le();
  if (hsz != 12) {
    const int compress = (int)s.get32le();
    if (compress == 1 || compress == 2) return why = "BMP type not supported: RLE", false;
    if (compress >= 4) return why = "BMP type not supported: unsupported compression", false;
    if (compress == 3 && h.bpp != 16 && h.bpp != 32) return why = "bad BMP", false;
    for (int k = 0; k < 5; k++) s.get32le();  // image size, resolutions, colours used / important
    if (hsz == 40 || hsz == 56) {
      if (hsz == 56)
        for (int k = 0; k < 4; k++) s.get32le();
      if (h.bpp == 16 || h.bpp == 32) {
        if (compress == 0) mask_defaults(h, compress);
        else if (compress == 3) {
          h.mr = s.get32le(), h.mg = s.get32le(), h.mb = s.get32le();
          h.extra_read += 12;
          if (h.mr == h.mg && h.mg == h.mb) return why = "bad BMP", false;
        } else return why = "bad BMP", false;
      }
    } else {  // V4 / V5
      h.mr = s.get32le(), h.mg = s.get32le(), h.mb = s.get32le(), h.ma = s.get32le();
      if (compress != 3) mask_defaults(h, compress);
      s.get32le();
      for (int k = 0; k < 12; k++) s.get32le();  // colour space
      if (hsz == 124)
        for (int k = 0; k < 4; k++) s.get32le();
    }
  }
  return true;
}
inline bool header(const uint8_t* d, size_t n, Size& size, std::string& why) {
  Bytes  s(d, n);
  Header h;
  if (!parse(s, h, why)) return why = "corrupt BMP: " + why, false;
  const int64_t height = h.height_signed < 0 ? -(int64_t)h.height_signed : h.height_signed;
  if (height > MAX_DIM || h.width > MAX_DIM || h.width < 0) return why = "corrupt BMP: very large image", false;
  if (!fits(h.width, height, 4)) return why = "corrupt BMP: too large", false;
  size = {h.width, (int)height};
  return true;
}
inline int high_bit(uint32_t z) {
  if (z == 0) return -1;
  int n = 0;
  if (z >= 0x10000) n += 16, z >>= 16;
  if (z >= 0x00100) n += 8, z >>= 8;
  if (z >= 0x00010) n += 4, z >>= 4;
  if (z >= 0x00004) n += 2, z >>= 2;
  if (z >= 0x00002) n += 1;
  return n;
}
inline int bitcount(uint32_t a) {
  int n = 0;
  for (; a; a &= a - 1) n++;
  return n;
}
// a masked field of `bits` bits whose top bit sits `shift` above bit 7 -> 8 bits, the value repeated into the low bits
inline int shiftsigned(uint32_t v, int shift, int bits) {
  static const unsigned mul[9] = {0, 0xff, 0x55, 0x49, 0x11, 0x21, 0x41, 0x81, 0x01}, sh[9] = {0, 0, 0, 1, 0, 2, 4, 6, 0};
  if (shift < 0) v <<= -shift;
  else v >>= shift;
  v >>= (8 - bits);
  return (int)((unsigned)v * mul[bits]) >> sh[bits];
}
inline bool decode(const uint8_t* d, size_t n, uint8_t* out, std::string& why) {
  Bytes  s(d, n);
  Header h;
  if (!parse(s, h, why)) return why = "corrupt BMP: " + why, false;
  const bool flip = h.height_signed > 0;
  const int  W = h.width, H = h.height_signed < 0 ? -h.height_signed : h.height_signed;
  if (H > MAX_DIM || W > MAX_DIM || W < 0 || !fits(W, H, 4)) return why = "corrupt BMP: too large", false;
  uint32_t mr = h.mr, mg = h.mg, mb = h.mb, ma = h.ma, all_a = h.all_a;
  int      psize = 0;
  if (h.hsz == 12) {
    if (h.bpp < 24) psize = (h.offset - h.extra_read - 24) / 3;
  } else if (h.bpp < 16) psize = (h.offset - h.extra_read - h.hsz) >> 2;
  if (psize == 0 && (size_t)h.offset != s.at()) return why = "corrupt BMP: bad offset", false;
  size_t z = 0;
  if (h.bpp < 16) {
    if (psize == 0 || psize > 256) return why = "corrupt BMP: invalid palette", false;
    uint8_t pal[256][4];
    for (int i = 0; i < psize; i++) {
      pal[i][2] = (uint8_t)s.get8(), pal[i][1] = (uint8_t)s.get8(), pal[i][0] = (uint8_t)s.get8();
      if (h.hsz != 12) s.get8();
      pal[i][3] = 255;
    }
    for (int i = psize; i < 256; i++) pal[i][0] = pal[i][1] = pal[i][2] = 0, pal[i][3] = 255;  // (stb_image leaves these unset)
    s.skip(h.offset - h.extra_read - h.hsz - psize * (h.hsz == 12 ? 3 : 4));
    int width;
    if (h.bpp == 1) width = (W + 7) >> 3;
    else if (h.bpp == 4) width = (W + 1) >> 1;
    else if (h.bpp == 8) width = W;
    else return why = "corrupt BMP: bad bpp", false;
    const int pad = (-width) & 3;
    auto      put = [&](int c) { out[z++] = pal[c][0], out[z++] = pal[c][1], out[z++] = pal[c][2], out[z++] = 255; };
    if (h.bpp == 1) {
      for (int j = 0; j < H; j++) {
        int bit = 7, v = s.get8();
        for (int i = 0; i < W; i++) {
          put((v >> bit) & 1);
          if (i + 1 == W) break;
          if (--bit < 0) bit = 7, v = s.get8();
        }
        s.skip(pad);
      }
    } else {
      for (int j = 0; j < H; j++) {
        for (int i = 0; i < W; i += 2) {
          int v = s.get8(), v2 = 0;
          if (h.bpp == 4) v2 = v & 15, v >>= 4;
          put(v);
          if (i + 1 == W) break;
          put(h.bpp == 8 ? s.get8() : v2);
        }
        s.skip(pad);
      }
    }
  } else {
    int rshift = 0, gshift = 0, bshift = 0, ashift = 0, rcount = 0, gcount = 0, bcount = 0, acount = 0, easy = 0;
    s.skip(h.offset - h.extra_read - h.hsz);
    const int width = h.bpp == 24 ? 3 * W : h.bpp == 16 ? 2 * W : 0, pad = (-width) & 3;
    if (h.bpp == 24) easy = 1;
    else if (h.bpp == 32 && mb == 0xff && mg == 0xff00 && mr == 0x00ff0000 && ma == 0xff000000) easy = 2;
    if (!easy) {
      if (!mr || !mg || !mb) return why = "corrupt BMP: bad masks", false;
      rshift = high_bit(mr) - 7, rcount = bitcount(mr), gshift = high_bit(mg) - 7, gcount = bitcount(mg);
      bshift = high_bit(mb) - 7, bcount = bitcount(mb), ashift = high_bit(ma) - 7, acount = bitcount(ma);
      if (rcount > 8 || gcount > 8 || bcount > 8 || acount > 8) return why = "corrupt BMP: bad masks", false;
    }
    for (int j = 0; j < H; j++) {
      if (easy) {
        for (int i = 0; i < W; i++) {
          out[z + 2] = (uint8_t)s.get8(), out[z + 1] = (uint8_t)s.get8(), out[z] = (uint8_t)s.get8();
          z += 3;
          const uint8_t a = easy == 2 ? (uint8_t)s.get8() : 255;
          all_a |= a, out[z++] = a;
        }
      } else {
        for (int i = 0; i < W; i++) {
          const uint32_t v = h.bpp == 16 ? (uint32_t)s.get16le() : s.get32le();
          out[z++]         = (uint8_t)(shiftsigned(v & mr, rshift, rcount) & 255);
          out[z++]         = (uint8_t)(shiftsigned(v & mg, gshift, gcount) & 255);
          out[z++]         = (uint8_t)(shiftsigned(v & mb, bshift, bcount) & 255);
          const unsigned a = ma ? (unsigned)shiftsigned(v & ma, ashift, acount) : 255u;
          all_a |= a, out[z++] = (uint8_t)(a & 255);
        }
      }
      s.skip(pad);
    }
  }
  if (all_a == 0)  // an alpha channel of zeros was no alpha channel
    for (size_t i = 3; i < (size_t)4 * W * H; i += 4) out[i] = 255;
  if (flip)
    for (int j = 0; j < H >> 1; j++) {
      uint8_t *p1 = out + (size_t)j * W * 4, *p2 = out + (size_t)(H - 1 - j) * W * 4;
      for (int i = 0; i < W * 4; i++) std::swap(p1[i], p2[i]);
    }
  return true;
}
}  // namespace bmp

// ---- TGA ----------------------------------------------------------------------------------------------------------------
namespace tga {
inline int components(int bits, bool grey, bool& rgb16) {
  rgb16 = false;
  switch (bits) {
    case 8: return 1;
    case 16:
      if (grey) return 2;
      rgb16 = true;
      return 3;
    case 15: rgb16 = true; return 3;
    case 24:
    case 32: return bits / 8;
    default: return 0;
  }
}
inline bool test(const uint8_t* d, size_t n) {
  Bytes s(d, n);
  s.get8();
  const int color_type = s.get8();
  if (color_type > 1) return false;
  int sz = s.get8();
  if (color_type == 1) {
    if (sz != 1 && sz != 9) return false;
    s.skip(4);
    sz = s.get8();
    if (sz != 8 && sz != 15 && sz != 16 && sz != 24 && sz != 32) return false;
    s.skip(4);
  } else {
    if (sz != 2 && sz != 3 && sz != 10 && sz != 11) return false;
    s.skip(9);
  }
  if (s.get16le() < 1 || s.get16le() < 1) return false;
  sz = s.get8();
  if (color_type == 1 && sz != 8 && sz != 16) return false;
  return sz == 8 || sz == 15 || sz == 16 || sz == 24 || sz == 32;
}
inline void rgb16(Bytes& s, uint8_t* out) {
  const unsigned px = (unsigned)s.get16le();
  out[0] = (uint8_t)((((px >> 10) & 31) * 255) / 31), out[1] = (uint8_t)((((px >> 5) & 31) * 255) / 31), out[2] = (uint8_t)(((px & 31) * 255) / 31);
}
struct Header {
  int  offset, indexed, type, pal_start, pal_len, pal_bits, width, height, bpp, inverted, comp;
  bool rle, is16;
};
inline bool parse(Bytes& s, Header& h, std::string& why) {
  h.offset = s.get8(), h.indexed = s.get8(), h.type = s.get8();
  h.pal_start = s.get16le(), h.pal_len = s.get16le(), h.pal_bits = s.get8();
  s.get16le(), s.get16le();  // origin
  h.width = s.get16le(), h.height = s.get16le(), h.bpp = s.get8();
  const int descriptor = s.get8();
  if (h.height > MAX_DIM || h.width > MAX_DIM) return why = "very large image", false;
  h.rle = h.type >= 8;
  if (h.rle) h.type -= 8;
  h.inverted = 1 - ((descriptor >> 5) & 1);
  h.comp     = h.indexed ? components(h.pal_bits, false, h.is16) : components(h.bpp, h.type == 3, h.is16);
  if (!h.comp) return why = "can't find out TGA pixel format", false;
  if (!fits(h.width, h.height, h.comp)) return why = "too large", false;
  return true;
}
inline bool header(const uint8_t* d, size_t n, Size& size, std::string& why) {
  Bytes  s(d, n);
  Header h;
  if (!parse(s, h, why)) return why = "corrupt TGA: " + why, false;
  if (!fits(h.width, h.height, 4)) return why = "corrupt TGA: too large", false;
  size = {h.width, h.height};
  return true;
}
inline bool decode(const uint8_t* d, size_t n, uint8_t* out, std::string& why) {
  Bytes  s(d, n);
  Header h;
  if (!parse(s, h, why)) return why = "corrupt TGA: " + why, false;
  const int            W = h.width, H = h.height, C = h.comp;
  std::vector<uint8_t> data((size_t)W * H * C + 4, 0), palette;
  s.skip(h.offset);
  if (!h.indexed && !h.rle && !h.is16) {
    for (int i = 0; i < H; i++) {
      const int row = h.inverted ? H - i - 1 : i;
      uint8_t*  r   = data.data() + (size_t)row * W * C;
      if (!s.getn(r, W * C)) {  // (a short file: stb_image's getn copies what is there and leaves the rest as malloc returned it)
        const size_t have = (size_t)(s.end - s.p);
        std::memcpy(r, s.p, have), s.p = s.end;
      }
    }
  } else {
    if (h.indexed) {
      if (h.pal_len == 0) return why = "corrupt TGA: bad palette", false;
      s.skip(h.pal_start);
      palette.assign((size_t)h.pal_len * C, 0);
      if (h.is16)
        for (int i = 0; i < h.pal_len; i++) rgb16(s, palette.data() + (size_t)i * C);
      else if (!s.getn(palette.data(), h.pal_len * C)) return why = "corrupt TGA: bad palette", false;
    }
    uint8_t raw[4]    = {0, 0, 0, 0};
    int     rle_count = 0, repeating = 0;
    bool    read_next = true;
    for (size_t i = 0; i < (size_t)W * H; i++) {
      if (h.rle) {
        if (rle_count == 0) {
          const int cmd = s.get8();
          rle_count = 1 + (cmd & 127), repeating = cmd >> 7, read_next = true;
        } else if (!repeating) read_next = true;
      } else read_next = true;
      if (read_next) {
        if (h.indexed) {
          int idx = h.bpp == 8 ? s.get8() : s.get16le();
          if (idx >= h.pal_len) idx = 0;
          for (int j = 0; j < C; j++) raw[j] = palette[(size_t)idx * C + j];
        } else if (h.is16) rgb16(s, raw);
        else
          for (int j = 0; j < C; j++) raw[j] = (uint8_t)s.get8();
        read_next = false;
      }
      for (int j = 0; j < C; j++) data[i * C + j] = raw[j];
      --rle_count;
    }
    if (h.inverted)
      for (int j = 0; j * 2 < H; j++) {
        uint8_t *a = data.data() + (size_t)j * W * C, *b = data.data() + (size_t)(H - 1 - j) * W * C;
        for (int i = 0; i < W * C; i++) std::swap(a[i], b[i]);
      }
  }
  if (C >= 3 && !h.is16)  // stored blue first
    for (size_t i = 0; i < (size_t)W * H; i++) std::swap(data[i * C], data[i * C + 2]);
  to_rgba(data.data(), C, (size_t)W * H, out);
  return true;
}
}  // namespace tga

}  // namespace ytimg
