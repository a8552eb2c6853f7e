// yt_exr.h — OpenEXR textures for ythip_load_scene (host code).
//
// What the reference gets from load_texture for an .exr (yocto_sceneio.cpp:1803-1812): tinyexr's LoadEXR, i.e. an RGBA
// float image.  This is a reader of the same files with the same results, restated from the file format and from what
// tinyexr (exts/tinyexr/tinyexr.h of the reference: LoadEXRWithLayer :11613-11860, ParseEXRHeader :10549-10810,
// DecodeEXRImage :11351-11500, DecodeChunk :10940-11300, DecodePixelData :9811-10430) does with them:
//
//   * single-part scan-line and tiled files (level 0 of a tiled file, as tinyexr), compression NONE / RLE / ZIPS / ZIP / PIZ; channels HALF (widened to float), FLOAT,
//     UINT (its BITS end up in the float, as in tinyexr, which reads the uint plane through a float pointer);
//   * the eight attributes tinyexr insists on must be there (channels, compression, dataWindow, displayWindow, lineOrder,
//     pixelAspectRatio, screenWindowCenter, screenWindowWidth);
//   * RGBA from the channel list as tinyexr picks it: one channel -> replicated into all four; otherwise the channels
//     named R, G, B (A) AMONG THE FIRST FOUR of the list (names taken after the last '.'), A absent -> 1;
//   * a DECREASING_Y file lands flipped, as in tinyexr (row = height - 1 - (line - dataWindow.min.y));
//   * chunk by chunk through the offset table (a zero entry: the table is rebuilt by walking the chunks).
//
// Refused by name (tinyexr misreads them): sub-sampled channels.  Rows no chunk covers are zero here (tinyexr leaves them as
// malloc returned them; that includes the short edge tiles of a DECREASING_Y tiled file, which it flips inside a full-size tile).
#pragma once

#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace ytexr {

struct Channel {
  std::string name;
  int         type;  // 0 UINT, 1 HALF, 2 FLOAT
  size_t      offset;  // byte offset of the channel's run inside one scan line, per pixel of width
};
struct Info {
  int                  width = 0, height = 0;
  int                  compression = 0, line_order = 0, min_x = 0, min_y = 0, max_y = 0;
  int                  chunk_count = 0;
  bool                 tiled = false;
  int                  tile_x = 0, tile_y = 0;  // tiled: the tile size
  size_t               header_end  = 0;  // offset of the chunk offset table
  size_t               pixel_bytes = 0;  // bytes of one pixel over all channels
  std::vector<Channel> channels;
  int                  idx[4] = {-1, -1, -1, -1};  // planes of R, G, B, A (all the same plane for a one-channel file)
};

inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
inline uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | (uint64_t)rd32(p + 4) << 32; }

inline float half_bits_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 1023u;
  uint32_t       u;
  if (e == 31) u = 0x7f800000u | (m << 13);  // inf / nan (payload kept)
  else if (e) u = ((e + 112u) << 23) | (m << 13);
  else if (m) {
    int      s  = 0;
    uint32_t mm = m;
    while (!(mm & 1024u)) mm <<= 1, s++;
    u = ((uint32_t)(113 - s) << 23) | ((mm & 1023u) << 13);
  } else u = 0;
  u |= sign;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

// header: everything LoadEXR checks before it touches pixel data
inline bool header(const uint8_t* data, size_t size, Info& info, std::string& why) {
  if (size < 8 || rd32(data) != 20000630u) return why = "not an OpenEXR file", false;
  if (data[4] != 2) return why = "unsupported EXR version", false;
  const bool tiled = data[5] & 0x2, long_names = data[5] & 0x4, non_image = data[5] & 0x8, multipart = data[5] & 0x10;
  (void)long_names;
  if (multipart || non_image) return why = "multipart or deep EXR files are not supported (nor by the reference's LoadEXR)", false;
  info.tiled = tiled;
  size_t at = 8;
  bool   has[8] = {false, false, false, false, false, false, false, false};
  int    max_x = 0;
  for (int nattr = 0;; nattr++) {
    if (nattr >= 1024 || at >= size) return why = "corrupt EXR: header does not end", false;
    if (data[at] == 0) {
      at++;
      break;
    }
    auto cstring = [&](std::string& s) {
      const void* z = std::memchr(data + at, 0, size - at);
      if (!z) return false;
      s.assign((const char*)data + at, (const char*)z);
      at = (size_t)((const uint8_t*)z - data) + 1;
      return true;
    };
    std::string name, type;
    if (!cstring(name) || !cstring(type) || size - at < 4) return why = "corrupt EXR: failed to read attribute", false;
    const uint32_t len = rd32(data + at);
    at += 4;
    if (len == 0 && type != "string") return why = "corrupt EXR: failed to read attribute", false;
    if (size - at < len) return why = "corrupt EXR: failed to read attribute", false;
    const uint8_t* v = data + at;
    at += len;
    if (name == "compression") {
      if (len < 1) return why = "corrupt EXR: failed to read attribute", false;  // (an empty "string"-typed value got here and v[0] was read past it)
      if (v[0] > 4) return why = v[0] == 128 ? "ZFP compression is not supported" : "unknown EXR compression type " + std::to_string((int)v[0]) + " (NONE, RLE, ZIPS, ZIP and PIZ are read)", false;
      info.compression = v[0], has[0] = true;
    } else if (name == "channels") {
      size_t p = 0;
      info.channels.clear();
      while (true) {
        if (p >= len) return why = "corrupt EXR: failed to parse channel info", false;
        if (v[p] == 0) break;
        const void* z = std::memchr(v + p, 0, len - p);
        if (!z) return why = "corrupt EXR: failed to parse channel info", false;
        Channel c;
        c.name.assign((const char*)v + p, (const char*)z);
        p = (size_t)((const uint8_t*)z - v) + 1;
        if (p + 16 >= len) return why = "corrupt EXR: failed to parse channel info", false;
        const uint32_t type_ = rd32(v + p), xs = rd32(v + p + 8), ys = rd32(v + p + 12);
        p += 16;
        if (type_ > 2) return why = "corrupt EXR: unknown channel type", false;
        if (xs != 1 || ys != 1) return why = "sub-sampled EXR channels are not read here", false;
        c.type = (int)type_;
        info.channels.push_back(c);
      }
      if (info.channels.empty()) return why = "corrupt EXR: # of channels is zero", false;
      has[1] = true;
    } else if (name == "dataWindow" && len >= 16) {
      info.min_x = (int)rd32(v), info.min_y = (int)rd32(v + 4), max_x = (int)rd32(v + 8), info.max_y = (int)rd32(v + 12), has[2] = true;
    } else if (name == "displayWindow" && len >= 16) has[3] = true;
    else if (name == "lineOrder" && len >= 1) info.line_order = v[0], has[4] = true;
    else if (name == "pixelAspectRatio" && len >= 4) has[5] = true;
    else if (name == "screenWindowCenter" && len >= 8) has[6] = true;
    else if (name == "screenWindowWidth" && len >= 4) has[7] = true;
    else if (name == "chunkCount" && len >= 4) info.chunk_count = (int)rd32(v);
    else if (name == "tiles" && tiled) {
      if (len != 9 || rd32(v) > 0x7fffffffu || rd32(v + 4) > 0x7fffffffu) return why = "corrupt EXR: tile sizes are invalid", false;
      info.tile_x = (int)rd32(v), info.tile_y = (int)rd32(v + 4);
    }
  }
  static const char* names[8] = {"compression", "channels", "dataWindow", "displayWindow", "lineOrder", "pixelAspectRatio",
      "screenWindowCenter", "screenWindowWidth"};
  for (int k = 0; k < 8; k++)
    if (!has[k]) return why = std::string("corrupt EXR: \"") + names[k] + "\" attribute not found in the header", false;
  info.header_end = at;
  const int64_t w = (int64_t)max_x - info.min_x + 1, h = (int64_t)info.max_y - info.min_y + 1;
  if (w < 1 || h < 1 || w > 1024 * 8192 || h > 1024 * 8192) return why = "corrupt EXR: invalid data window", false;
  if ((double)w * (double)h * 16.0 > 2147483647.0 * 4.0) return why = "EXR image is too large", false;
  info.width = (int)w, info.height = (int)h;
  if (tiled && (info.tile_x < 1 || info.tile_y < 1 || info.tile_x > info.width || info.tile_y > info.height))
    return why = "corrupt EXR: tile sizes are invalid", false;
  size_t off = 0;
  for (auto& c : info.channels) c.offset = off, off += c.type == 1 ? 2 : 4;
  info.pixel_bytes = off;
  // RGBA as LoadEXRWithLayer picks it (layer ""): names after the last '.', R / G / B / A among the first four channels
  if (info.channels.size() == 1) {
    info.idx[0] = info.idx[1] = info.idx[2] = info.idx[3] = 0;
  } else {
    for (size_t c = 0; c < info.channels.size() && c < 4; c++) {
      std::string n   = info.channels[c].name;
      const auto  dot = n.find_last_of('.');
      if (dot != std::string::npos) n = n.substr(dot + 1);
      if (n == "R") info.idx[0] = (int)c;
      else if (n == "G") info.idx[1] = (int)c;
      else if (n == "B") info.idx[2] = (int)c;
      else if (n == "A") info.idx[3] = (int)c;
    }
    static const char* rgb = "RGB";
    for (int k = 0; k < 3; k++)
      if (info.idx[k] < 0) return why = std::string(1, rgb[k]) + " channel not found", false;
  }
  return true;
}

// ---- decompressors ---------------------------------------------------------------------------------------------------
inline void unpredict_and_interleave(std::vector<uint8_t>& tmp, uint8_t* dst) {
  const size_t n = tmp.size();
  for (size_t k = 1; k < n; k++) tmp[k] = (uint8_t)((int)tmp[k - 1] + (int)tmp[k] - 128);
  const uint8_t *t1 = tmp.data(), *t2 = tmp.data() + (n + 1) / 2;
  for (size_t k = 0; k < n; k++) dst[k] = (k & 1) ? *t2++ : *t1++;
}
inline bool unzip(uint8_t* dst, size_t want, const uint8_t* src, size_t src_size) {
  if (want == src_size) return std::memcpy(dst, src, src_size), true;  // stored as it was (tinyexr issue 40)
  std::vector<uint8_t> tmp(want);
  uLongf               got = (uLongf)want;
  if (uncompress(tmp.data(), &got, src, (uLong)src_size) != Z_OK) return false;
  tmp.resize(got);  // (a shorter stream leaves the tail of dst untouched, as tinyexr does)
  unpredict_and_interleave(tmp, dst);
  return true;
}
inline bool unrle(uint8_t* dst, size_t want, const uint8_t* src, size_t src_size) {
  if (want == src_size) return std::memcpy(dst, src, src_size), true;
  if (src_size <= 2) return false;
  std::vector<uint8_t> tmp(want);
  size_t               o = 0, i = 0;
  while (i < src_size) {
    const int c = (int8_t)src[i++];
    if (c < 0) {  // -c literal bytes
      const size_t n = (size_t)-c;
      if (o + n > want || i + n > src_size) return false;
      std::memcpy(tmp.data() + o, src + i, n), o += n, i += n;
    } else {  // the next byte c + 1 times
      const size_t n = (size_t)c + 1;
      if (o + n > want || i >= src_size) return false;
      std::memset(tmp.data() + o, src[i++], n), o += n;
    }
  }
  if (o != want) return false;
  unpredict_and_interleave(tmp, dst);
  return true;
}

// ---- PIZ: range-compacted 16-bit words, Haar-like wavelet, canonical Huffman code (OpenEXR's ImfPizCompressor / ImfHuf /
// ImfWav as tinyexr carries them, :8000-9570) ---------------------------------------------------------------------------
namespace piz {
constexpr int HUF_ENCBITS = 16, HUF_ENCSIZE = (1 << HUF_ENCBITS) + 1;
constexpr int SHORT_ZEROCODE_RUN = 59, LONG_ZEROCODE_RUN = 63, SHORTEST_LONG_RUN = 2 + LONG_ZEROCODE_RUN - SHORT_ZEROCODE_RUN;

struct BitReader {
  const uint8_t *p, *end;
  uint64_t       c  = 0;
  int            lc = 0;
  bool           overrun = false;
  uint32_t       get(int n) {
    while (lc < n) {
      if (p >= end) {
        overrun = true;
        c <<= 8, lc += 8;
        continue;
      }
      c = (c << 8) | *p++, lc += 8;
    }
    lc -= n;
    return (uint32_t)((c >> lc) & ((1ull << n) - 1));
  }
};

// code lengths (6 bits each, zero runs packed) for symbols [im, iM] -> canonical codes: code | length in one word
inline bool unpack_table(const uint8_t*& p, size_t ni, int im, int iM, std::vector<uint64_t>& hcode) {
  BitReader r{p, p + ni};
  for (; im <= iM; im++) {
    if (r.p > r.end) return false;
    const uint64_t l = hcode[(size_t)im] = r.get(6);
    if (r.overrun) return false;
    if (l == (uint64_t)LONG_ZEROCODE_RUN) {
      int zerun = (int)r.get(8) + SHORTEST_LONG_RUN;
      if (r.overrun || im + zerun > iM + 1) return false;
      while (zerun--) hcode[(size_t)im++] = 0;
      im--;
    } else if (l >= (uint64_t)SHORT_ZEROCODE_RUN) {
      int zerun = (int)l - SHORT_ZEROCODE_RUN + 2;
      if (im + zerun > iM + 1) return false;
      while (zerun--) hcode[(size_t)im++] = 0;
      im--;
    }
  }
  p = r.p;
  // canonical codes: shorter codes have the numerically LARGER prefixes (hufCanonicalCodeTable)
  uint64_t n[59];
  for (auto& x : n) x = 0;
  for (auto h : hcode) n[h] += 1;
  uint64_t c = 0;
  for (int i = 58; i > 0; --i) {
    const uint64_t nc = (c + n[i]) >> 1;
    n[i]              = c;
    c                 = nc;
  }
  for (auto& h : hcode) {
    const int l = (int)h;
    if (l > 0) h = (uint64_t)l | (n[l]++ << 6);
  }
  return true;
}

// decodes nBits of code into no 16-bit words; rlc = the run-length symbol (iM): "repeat the previous word <8 bits> times"
inline bool decode(const std::vector<uint64_t>& hcode, const uint8_t* in, int nBits, int rlc, int no, uint16_t* out) {
  // (length, code) -> symbol.  A plain map by length: first code and first index of each length in a sorted list.
  struct Entry {
    uint64_t code;
    int      len, sym;
  };
  std::vector<Entry> sorted;
  for (size_t s = 0; s < hcode.size(); s++)
    if (hcode[s] & 63) sorted.push_back({hcode[s] >> 6, (int)(hcode[s] & 63), (int)s});
  if (sorted.empty()) return no == 0;
  std::vector<std::vector<std::pair<uint64_t, int>>> by_len(59);
  for (auto& e : sorted) by_len[(size_t)e.len].push_back({e.code, e.sym});
  for (auto& v : by_len) std::sort(v.begin(), v.end());
  // a 14-bit first-level table for the short codes, as the format's designers intended; longer codes by search
  constexpr int              FAST = 14;
  std::vector<int32_t>       fast((size_t)1 << FAST, -1);  // sym << 6 | len
  for (auto& e : sorted)
    if (e.len <= FAST) {
      const uint64_t base = e.code << (FAST - e.len);
      for (uint64_t k = 0; k < (1ull << (FAST - e.len)); k++) fast[(size_t)(base + k)] = (int32_t)((e.sym << 6) | e.len);
    }
  const uint8_t* end = in + (nBits + 7) / 8;
  // (the bit window is 128 bits wide: codes are up to 58 bits long and the window is refilled by whole bytes, so with 57 bits
  //  in it the next byte would push the oldest valid bit out of a 64-bit word — ADVICE r5)
  unsigned __int128 c  = 0;
  int               lc = 0;
  int64_t           bits_left = nBits;
  int            o   = 0;
  auto           fill = [&](int n) {
    while (lc < n) {
      c = (c << 8) | (in < end ? *in++ : 0), lc += 8;
    }
  };
  auto emit = [&](int sym) -> bool {
    if (sym == rlc) {
      if (bits_left < 8) return false;
      fill(8);
      lc -= 8, bits_left -= 8;
      int n = (int)((c >> lc) & 255);
      if (o == 0 || o + n > no) return false;
      const uint16_t s = out[o - 1];
      while (n-- > 0) out[o++] = s;
    } else {
      if (o >= no) return false;
      out[o++] = (uint16_t)sym;
    }
    return true;
  };
  while (bits_left > 0) {
    fill(FAST);
    const uint32_t peek = (uint32_t)((c >> (lc - FAST)) & ((1u << FAST) - 1));
    const int32_t  f    = fast[peek];
    if (f >= 0 && (f & 63) <= bits_left) {
      lc -= f & 63, bits_left -= f & 63;
      if (!emit(f >> 6)) return false;
      continue;
    }
    // a long code (or the tail of the stream: fewer than FAST real bits left)
    bool found = false;
    for (int l = 1; l <= 58 && l <= bits_left; l++) {
      if (by_len[(size_t)l].empty()) continue;
      fill(l);
      const uint64_t code = (uint64_t)(c >> (lc - l)) & ((1ull << l) - 1);
      auto&          v    = by_len[(size_t)l];
      auto           it   = std::lower_bound(v.begin(), v.end(), std::make_pair(code, -1));
      if (it != v.end() && it->first == code) {
        lc -= l, bits_left -= l;
        if (!emit(it->second)) return false;
        found = true;
        break;
      }
    }
    if (!found) {
      // trailing padding bits that match no code end the stream (OpenEXR's decoder shifts them out the same way)
      break;
    }
  }
  return o == no;
}

inline bool huf_uncompress(const uint8_t* src, size_t n, uint16_t* raw, int nRaw) {
  if (n == 0) return nRaw == 0;
  if (n < 20) return false;
  const int im = (int)rd32(src), iM = (int)rd32(src + 4), nBits = (int)rd32(src + 12);
  if (im < 0 || im >= HUF_ENCSIZE || iM < 0 || iM >= HUF_ENCSIZE || nBits < 0) return false;
  const uint8_t*        p = src + 20;
  std::vector<uint64_t> hcode((size_t)HUF_ENCSIZE, 0);
  if (!unpack_table(p, n - 20, im, iM, hcode)) return false;
  if ((size_t)(p - src) > n || (int64_t)nBits > 8 * (int64_t)(n - (size_t)(p - src))) return false;
  return decode(hcode, p, nBits, iM, nRaw, raw);
}

// inverse wavelet: 14-bit (wdec14) when every value < 1 << 14, else the 16-bit modulo form (wdec16)
inline void wdec14(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
  const int16_t ls = (int16_t)l, hs = (int16_t)h;
  const int     hi = hs, ai = ls + (hi & 1) + (hi >> 1);
  a = (uint16_t)(int16_t)ai, b = (uint16_t)(int16_t)(ai - hi);
}
inline void wdec16(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
  constexpr int A_OFFSET = 1 << 15, MOD_MASK = (1 << 16) - 1;
  const int     m = l, d = h, bb = (m - (d >> 1)) & MOD_MASK, aa = (d + bb - A_OFFSET) & MOD_MASK;
  b = (uint16_t)bb, a = (uint16_t)aa;
}
inline void wav2_decode(uint16_t* in, int nx, int ox, int ny, int oy, uint16_t mx) {
  const bool w14 = mx < (1 << 14);
  const int  n   = nx > ny ? ny : nx;
  int        p   = 1, p2;
  while (p <= n) p <<= 1;
  p >>= 1, p2 = p, p >>= 1;
  auto dec = [&](uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) { w14 ? wdec14(l, h, a, b) : wdec16(l, h, a, b); };
  while (p >= 1) {
    uint16_t*       py  = in;
    uint16_t* const ey  = in + oy * (ny - p2);
    const int       oy1 = oy * p, oy2 = oy * p2, ox1 = ox * p, ox2 = ox * p2;
    uint16_t        i00, i01, i10, i11;
    for (; py <= ey; py += oy2) {
      uint16_t*       px = py;
      uint16_t* const ex = py + ox * (nx - p2);
      for (; px <= ex; px += ox2) {
        uint16_t *p01 = px + ox1, *p10 = px + oy1, *p11 = p10 + ox1;
        dec(*px, *p10, i00, i10);
        dec(*p01, *p11, i01, i11);
        dec(i00, i01, *px, *p01);
        dec(i10, i11, *p10, *p11);
      }
      if (nx & p) {
        uint16_t* p10 = px + oy1;
        dec(*px, *p10, i00, *p10);
        *px = i00;
      }
    }
    if (ny & p) {
      uint16_t*       px = py;
      uint16_t* const ex = py + ox * (nx - p2);
      for (; px <= ex; px += ox2) {
        uint16_t* p01 = px + ox1;
        dec(*px, *p01, i00, *p01);
        *px = i00;
      }
    }
    p2 = p, p >>= 1;
  }
}

// one chunk: `lines` scan lines of `width` pixels, channel by channel inside each line
inline bool uncompress_chunk(uint8_t* dst, size_t want, const uint8_t* src, size_t src_size, const Info& info, int width, int lines) {
  if (want == src_size) return std::memcpy(dst, src, src_size), true;  // stored as it was
  constexpr int        BITMAP_SIZE = 8192;
  std::vector<uint8_t> bitmap(BITMAP_SIZE, 0);
  if (src_size < 4) return false;
  const int minNonZero = (int)(src[0] | src[1] << 8), maxNonZero = (int)(src[2] | src[3] << 8);
  size_t    at         = 4;
  if (maxNonZero >= BITMAP_SIZE) return false;
  if (minNonZero <= maxNonZero) {
    const size_t n = (size_t)(maxNonZero - minNonZero + 1);
    if (src_size - at < n) return false;
    std::memcpy(bitmap.data() + minNonZero, src + at, n), at += n;
  }
  std::vector<uint16_t> lut(65536, 0);
  int                   k = 0;
  for (int i = 0; i < 65536; i++)
    if (i == 0 || (bitmap[(size_t)i >> 3] & (1 << (i & 7)))) lut[(size_t)k++] = (uint16_t)i;
  const uint16_t maxValue = (uint16_t)(k - 1);
  if (src_size - at < 4) return false;
  const int length = (int)rd32(src + at);
  at += 4;
  if (length < 0 || (size_t)length > src_size - at) return false;
  std::vector<uint16_t> tmp(want / 2 + 1);
  if (!huf_uncompress(src + at, (size_t)length, tmp.data(), (int)(want / 2))) return false;
  // the chunk's words are stored channel after channel (all lines of a channel together); wavelet-decode each plane
  struct Plane {
    uint16_t* start;
    int       nx, ny, size;
  };
  std::vector<Plane> planes;
  uint16_t*          t = tmp.data();
  for (auto& c : info.channels) {
    const int sz = c.type == 1 ? 1 : 2;
    planes.push_back({t, width, lines, sz});
    t += (size_t)width * lines * sz;
  }
  for (auto& pl : planes)
    for (int j = 0; j < pl.size; j++) wav2_decode(pl.start + j, pl.nx, pl.size, pl.ny, pl.nx * pl.size, maxValue);
  for (size_t i = 0; i < want / 2; i++) tmp[i] = lut[tmp[i]];
  // back to line-interleaved order (native-endian words; the file is little-endian and so are we)
  uint8_t*               out = dst;
  std::vector<uint16_t*> cur;
  for (auto& pl : planes) cur.push_back(pl.start);
  for (int y = 0; y < lines; y++)
    for (size_t c = 0; c < planes.size(); c++) {
      const size_t n = (size_t)planes[c].nx * planes[c].size;
      std::memcpy(out, cur[c], n * 2), out += n * 2, cur[c] += n;
    }
  return true;
}
}  // namespace piz

// one chunk's pixel data -> rows of the RGBA image: `lines` rows of `width` pixels land at (x0, row0 + v) — or, line_order 1,
// at (x0, flip - (row0 + v)): tinyexr's reading of DECREASING_Y
inline bool decode_block(const Info& info, const uint8_t* src, size_t len, int width, int lines, int x0, int64_t row0, int64_t flip,
    float* out, std::vector<uint8_t>& raw) {
  const size_t   want = (size_t)width * (size_t)lines * info.pixel_bytes;
  const uint8_t* px   = src;
  if (info.compression == 0) {
    if (len < want) return false;
  } else {
    raw.assign(want, 0);
    const bool ok = info.compression == 1   ? unrle(raw.data(), want, src, len)
                    : info.compression == 4 ? piz::uncompress_chunk(raw.data(), want, src, len, info, width, lines)
                                            : unzip(raw.data(), want, src, len);
    if (!ok) return false;
    px = raw.data();
  }
  for (int v = 0; v < lines; v++) {
    const int64_t row = info.line_order == 0 ? row0 + v : flip - (row0 + v);
    if (row < 0 || row >= info.height) continue;
    float*         o = out + ((size_t)row * (size_t)info.width + (size_t)x0) * 4;
    const uint8_t* l = px + (size_t)v * info.pixel_bytes * (size_t)width;
    for (int comp = 0; comp < 4; comp++) {
      if (info.idx[comp] < 0) continue;
      const Channel& c = info.channels[(size_t)info.idx[comp]];
      const uint8_t* s = l + c.offset * (size_t)width;
      if (c.type == 1) {
        for (int u = 0; u < width; u++) o[4 * u + comp] = half_bits_to_float((uint16_t)(s[2 * u] | s[2 * u + 1] << 8));
      } else {  // FLOAT, or UINT whose bits tinyexr hands over as they are
        for (int u = 0; u < width; u++) std::memcpy(&o[4 * u + comp], s + 4 * (size_t)u, 4);
      }
    }
  }
  return true;
}

// pixels: RGBA floats, width * height * 4
inline bool decode(const uint8_t* data, size_t size, const Info& info, float* out, std::string& why) {
  const int    block   = info.compression == 3 ? 16 : info.compression == 4 ? 32 : 1;
  const size_t tiles_x = info.tiled ? ((size_t)info.width + info.tile_x - 1) / info.tile_x : 0;
  const size_t tiles_y = info.tiled ? ((size_t)info.height + info.tile_y - 1) / info.tile_y : 0;
  const size_t natural = info.tiled ? tiles_x * tiles_y : ((size_t)info.height + block - 1) / block;
  const size_t blocks  = info.chunk_count > 0 ? (size_t)info.chunk_count : natural;
  if (blocks > natural + 16 || size - info.header_end < blocks * 8) return why = "corrupt EXR: insufficient data size in offset table", false;
  std::vector<uint64_t> offsets(blocks);
  bool                  rebuild = false;
  for (size_t k = 0; k < blocks; k++) {
    offsets[k] = rd64(data + info.header_end + 8 * k);
    if (offsets[k] >= size) return why = "corrupt EXR: invalid offset value", false;
    rebuild = rebuild || offsets[k] == 0;
  }
  if (rebuild) {  // an unfinished file: the chunks follow the table back to back (tinyexr walks them as scan-line chunks)
    size_t at = info.header_end + 8 * blocks;
    for (size_t k = 0; k < blocks; k++) {
      if (at + 8 >= size) return why = "corrupt EXR: cannot reconstruct the line offset table", false;
      const uint32_t len = rd32(data + at + 4);
      if (len >= size) return why = "corrupt EXR: cannot reconstruct the line offset table", false;
      offsets[k] = at;
      at += 8 + (size_t)len;
    }
  }
  const size_t npix = (size_t)info.width * info.height;
  for (size_t k = 0; k < npix * 4; k++) out[k] = 0;
  if (info.idx[3] < 0)
    for (size_t k = 0; k < npix; k++) out[4 * k + 3] = 1.0f;
  std::vector<uint8_t> raw;
  const char*          bad = "corrupt EXR: invalid data found when decoding pixels";
  for (size_t k = 0; k < blocks; k++) {
    const uint64_t off = offsets[k];
    if (info.tiled) {
      if (off + 20 > size) return why = bad, false;
      const int32_t tx = (int32_t)rd32(data + off), ty = (int32_t)rd32(data + off + 4), lx = (int32_t)rd32(data + off + 8),
                    ly = (int32_t)rd32(data + off + 12), len = (int32_t)rd32(data + off + 16);
      if (lx != 0 || ly != 0) return why = "mip / rip levels among the first tiles of an EXR file are not supported (nor by the reference)", false;
      if (len < 4 || (uint64_t)len > size - (off + 20)) return why = bad, false;
      if (tx < 0 || ty < 0 || (int64_t)tx * info.tile_x > info.width || (int64_t)ty * info.tile_y > info.height) return why = bad, false;
      const int64_t x0 = (int64_t)tx * info.tile_x, y0 = (int64_t)ty * info.tile_y;
      const int     tw = (int)std::min<int64_t>(info.tile_x, info.width - x0), th = (int)std::min<int64_t>(info.tile_y, info.height - y0);
      if (tw <= 0 || th <= 0) continue;  // (a tile that starts on the image's edge: nothing of it is copied)
      // DECREASING_Y: tinyexr flips the rows inside the full-size tile buffer
      if (!decode_block(info, data + off + 20, (size_t)len, tw, th, (int)x0, y0, 2 * y0 + info.tile_y - 1, out, raw)) return why = bad, false;
      continue;
    }
    if (off + 8 > size) return why = bad, false;
    int64_t       line = (int32_t)rd32(data + off);
    const int32_t len  = (int32_t)rd32(data + off + 4);
    if (len <= 0 || (uint64_t)len > size - (off + 8) || line > (2 << 20) || line < -(2 << 20)) return why = bad, false;
    const int64_t end   = std::min<int64_t>(line + block, (int64_t)info.max_y + 1);
    const int     lines = (int)(end - line);
    line -= info.min_y;
    if (lines <= 0 || line < 0 || line + lines > info.height) return why = bad, false;
    if (!decode_block(info, data + off + 8, (size_t)len, info.width, lines, 0, line, (int64_t)info.height - 1, out, raw)) return why = bad, false;
  }
  return true;
}

}  // namespace ytexr
