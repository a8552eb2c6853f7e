// yt_sceneio.hip — a scene FILE into the flat pools, host code.
//
// The reference's load_scene for its builtin format (load_json_scene, yocto_sceneio.cpp:3618-3857)
// parses scene.json into a json tree, fills scene_data's vectors of cameras / textures / materials /
// shapes / instances / environments, then loads every shape (load_shape -> load_ply) and every
// texture (load_texture -> stb_image) into per-object vectors, and the application flattens those
// for the device: every byte is copied three or four times before it reaches HBM.  Here the same
// file set is opened once — scene.json parsed, every PLY mapped and its header scanned, every
// texture's header read — to get the COUNTS; the caller (or ythip_load_scene) sizes the pinned
// staging pools of ythip_scene_staging from them; ythip_scene_read then converts shapes and decodes
// textures straight into the pools at their offsets, on a thread pool (the reference's
// parallel_for over shapes and textures, :3822-3843), and ythip_upload_scene_staged sends them.
// The pools are, byte for byte, what the reference's loader + the flatten step produce
// (tests/test_sceneio.py compares them with the live reference on its own test corpus).
//
// What is read: the builtin JSON format, versions 4.2 / 5.0 (what save_scene writes) and 4.0 (files
// without asset.version: named elements that refer to each other by name), shapes in PLY, textures
// in Radiance HDR (stbi_loadf's reader, stb_image.h:7080-7197 of the reference's vendored copy: RLE
// and flat scanlines), PNG (stbi_load's results: every colour type, bit depth, tRNS, Adam7;
// inflate through zlib), JPEG (yt_jpeg.h: stb_image's baseline / progressive decoder restated, its
// inverse DCT, upsampling and colour arithmetic bit for bit) and OpenEXR (yt_exr.h: tinyexr's LoadEXR
// on scan-line and tiled files, NONE / RLE / ZIPS / ZIP / PIZ), BMP and TGA (yt_bmptga.h: stb_image's readers).  Anything else — subdivs (tesselate_subdivs is scene processing, re-used
// from the reference: SURVEY.md §2), format 4.1, PLY instance files, OBJ / glTF / PBRT scenes,
// GIF / PSD / PIC / PNM content, .ypreset textures — fails loudly by name: those stay with the
// reference's loader, whose scene_data goes through ythip_upload_scene as before.
//
// No device code here; the file is a .hip unit only so that the one build rule covers it.
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <filesystem>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/ythip.h"
#include "yt_bmptga.h"
#include "yt_exr.h"
#include "yt_jpeg.h"

namespace ytio {
int fail(int code, const std::string& msg);  // yt_io.hip: sets ythip_io_last_error() of this thread
}

namespace {

using ytio::fail;

// ---------------------------------------------------------------------------------------------
// a mapped file
// ---------------------------------------------------------------------------------------------
struct FileBytes {
  std::vector<uint8_t> data;
  bool                 load(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (n < 0) {
      std::fclose(f);
      return false;
    }
    data.resize((size_t)n);
    bool ok = n == 0 || std::fread(data.data(), 1, (size_t)n, f) == (size_t)n;
    std::fclose(f);
    return ok;
  }
};

// ---------------------------------------------------------------------------------------------
// JSON tree.  Conversions follow nlohmann::json's (the reference's json_value): numbers keep
// their lexical class (integer / unsigned / float) and are static_cast to the target; arithmetic
// targets other than bool also accept a boolean; a bool target accepts only a boolean; a string
// target only a string; fixed arrays take the first N items and fail when there are fewer.
// ---------------------------------------------------------------------------------------------
struct Json {
  enum Kind { Null, Bool, Int, Uint, Float, String, Array, Object } kind = Null;
  bool                                      b = false;
  long long                                 i = 0;
  unsigned long long                        u = 0;
  double                                    d = 0;
  std::string                               s;
  std::vector<Json>                         items;
  std::vector<std::pair<std::string, Json>> members;

  const Json* find(const char* key) const {
    if (kind != Object) return nullptr;
    const Json* hit = nullptr;
    for (auto& m : members)
      if (m.first == key) hit = &m.second;  // (a repeated key: the last value stays)
    return hit;
  }
};

struct JsonParser {
  const char* p;
  const char* end;
  int         depth = 0;
  void        ws() {
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
  }
  static void utf8(std::string& out, unsigned cp) {
    if (cp < 0x80) out += (char)cp;
    else if (cp < 0x800) out += (char)(0xC0 | (cp >> 6)), out += (char)(0x80 | (cp & 0x3F));
    else if (cp < 0x10000) out += (char)(0xE0 | (cp >> 12)), out += (char)(0x80 | ((cp >> 6) & 0x3F)), out += (char)(0x80 | (cp & 0x3F));
    else
      out += (char)(0xF0 | (cp >> 18)), out += (char)(0x80 | ((cp >> 12) & 0x3F)), out += (char)(0x80 | ((cp >> 6) & 0x3F)),
          out += (char)(0x80 | (cp & 0x3F));
  }
  bool hex4(unsigned& v) {
    if (end - p < 4) return false;
    v = 0;
    for (int k = 0; k < 4; k++, p++) {
      char c = *p;
      v <<= 4;
      if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
      else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
      else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
      else return false;
    }
    return true;
  }
  bool string(std::string& out) {
    if (p >= end || *p != '"') return false;
    p++;
    out.clear();
    while (p < end && *p != '"') {
      if ((unsigned char)*p < 0x20) return false;
      if (*p != '\\') {
        out += *p++;
        continue;
      }
      if (++p >= end) return false;
      char c = *p++;
      switch (c) {
        case '"': out += '"'; break;
        case '\\': out += '\\'; break;
        case '/': out += '/'; break;
        case 'b': out += '\b'; break;
        case 'f': out += '\f'; break;
        case 'n': out += '\n'; break;
        case 'r': out += '\r'; break;
        case 't': out += '\t'; break;
        case 'u': {
          unsigned cp = 0;
          if (!hex4(cp)) return false;
          if (cp >= 0xD800 && cp < 0xDC00) {  // surrogate pair
            unsigned lo = 0;
            if (end - p < 6 || p[0] != '\\' || p[1] != 'u') return false;
            p += 2;
            if (!hex4(lo) || lo < 0xDC00 || lo > 0xDFFF) return false;
            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
          } else if (cp >= 0xDC00 && cp < 0xE000) {
            return false;
          }
          utf8(out, cp);
          break;
        }
        default: return false;
      }
    }
    if (p >= end) return false;
    p++;
    return true;
  }
  bool number(Json& v) {
    const char* s = p;
    if (p < end && *p == '-') p++;
    if (p >= end || *p < '0' || *p > '9') return false;
    if (*p == '0') p++;
    else
      while (p < end && *p >= '0' && *p <= '9') p++;
    bool integral = true;
    if (p < end && *p == '.') {
      integral = false;
      p++;
      if (p >= end || *p < '0' || *p > '9') return false;
      while (p < end && *p >= '0' && *p <= '9') p++;
    }
    if (p < end && (*p == 'e' || *p == 'E')) {
      integral = false;
      p++;
      if (p < end && (*p == '+' || *p == '-')) p++;
      if (p >= end || *p < '0' || *p > '9') return false;
      while (p < end && *p >= '0' && *p <= '9') p++;
    }
    std::string tok(s, (size_t)(p - s));
    if (integral) {  // integers that do not fit 64 bits become floats, as in the reference's lexer
      errno = 0;
      if (tok[0] == '-') {
        long long x = std::strtoll(tok.c_str(), nullptr, 10);
        if (errno == 0) {
          v.kind = Json::Int, v.i = x;
          return true;
        }
      } else {
        unsigned long long x = std::strtoull(tok.c_str(), nullptr, 10);
        if (errno == 0) {
          v.kind = Json::Uint, v.u = x;
          return true;
        }
      }
    }
    v.kind = Json::Float;
    v.d    = std::strtod(tok.c_str(), nullptr);
    return std::isfinite(v.d);  // ("number overflow" is a parse error there)
  }
  bool lit(const char* s) {
    size_t n = std::strlen(s);
    if ((size_t)(end - p) < n || std::memcmp(p, s, n) != 0) return false;
    p += n;
    return true;
  }
  bool value(Json& v) {
    ws();
    if (p >= end || ++depth > 256) return false;
    bool ok = false;
    if (*p == '{') {
      p++;
      v.kind = Json::Object;
      ws();
      if (p < end && *p == '}') {
        p++;
        ok = true;
      } else {
        while (true) {
          ws();
          std::string key;
          if (!string(key)) break;
          ws();
          if (p >= end || *p != ':') break;
          p++;
          v.members.emplace_back(std::move(key), Json{});
          if (!value(v.members.back().second)) break;
          ws();
          if (p < end && *p == ',') {
            p++;
            continue;
          }
          if (p < end && *p == '}') {
            p++;
            ok = true;
          }
          break;
        }
      }
    } else if (*p == '[') {
      p++;
      v.kind = Json::Array;
      ws();
      if (p < end && *p == ']') {
        p++;
        ok = true;
      } else {
        while (true) {
          v.items.emplace_back();
          if (!value(v.items.back())) break;
          ws();
          if (p < end && *p == ',') {
            p++;
            continue;
          }
          if (p < end && *p == ']') {
            p++;
            ok = true;
          }
          break;
        }
      }
    } else if (*p == '"') {
      v.kind = Json::String;
      ok     = string(v.s);
    } else if (lit("true")) {
      v.kind = Json::Bool, v.b = true, ok = true;
    } else if (lit("false")) {
      v.kind = Json::Bool, v.b = false, ok = true;
    } else if (lit("null")) {
      v.kind = Json::Null, ok = true;
    } else {
      ok = number(v);
    }
    depth--;
    return ok;
  }
  bool document(Json& v) {
    if (end - p >= 3 && (unsigned char)p[0] == 0xEF && (unsigned char)p[1] == 0xBB && (unsigned char)p[2] == 0xBF) p += 3;  // UTF-8 BOM
    if (!value(v)) return false;
    ws();
    return p == end;
  }
};

struct BadValue {};  // a conversion the reference's json would throw on: "cannot parse <file>"

float to_float(const Json& j) {
  switch (j.kind) {
    case Json::Int: return (float)j.i;
    case Json::Uint: return (float)j.u;
    case Json::Float: return (float)j.d;
    case Json::Bool: return j.b ? 1.0f : 0.0f;
    default: throw BadValue{};
  }
}
int32_t to_int(const Json& j) {
  switch (j.kind) {
    case Json::Int: return (int32_t)j.i;
    case Json::Uint: return (int32_t)j.u;
    case Json::Float: return (int32_t)j.d;
    case Json::Bool: return j.b ? 1 : 0;
    default: throw BadValue{};
  }
}
void get_opt(const Json& e, const char* key, float& v) {
  if (auto j = e.find(key)) v = to_float(*j);
}
void get_opt(const Json& e, const char* key, int32_t& v) {
  if (auto j = e.find(key)) v = to_int(*j);
}
void get_flag(const Json& e, const char* key, int32_t& v) {  // a bool field widened to int32 in the flat records
  if (auto j = e.find(key)) {
    if (j->kind != Json::Bool) throw BadValue{};
    v = j->b ? 1 : 0;
  }
}
void get_opt(const Json& e, const char* key, std::string& v) {
  if (auto j = e.find(key)) {
    if (j->kind != Json::String) throw BadValue{};
    v = j->s;
  }
}
template <size_t N>
void get_floats(const Json& e, const char* key, float (&v)[N]) {
  if (auto j = e.find(key)) {
    if (j->kind != Json::Array || j->items.size() < N) throw BadValue{};
    float t[N];  // (the reference converts into a temporary: nothing is written when an item is bad)
    for (size_t k = 0; k < N; k++) t[k] = to_float(j->items[k]);
    std::memcpy(v, t, sizeof(t));
  }
}
void get_frame(const Json& e, const char* key, ythip_frame& f) {
  float t[12];
  std::memcpy(t, &f, sizeof(t));
  get_floats(e, key, t);
  std::memcpy(&f, t, sizeof(t));
}

// the little vector algebra the loader's fix-ups need, in the reference's association order
// (yocto_math.h:1290-1320, :2263; this unit is built with -ffp-contract=off like the rest)
struct vec3f {
  float x, y, z;
};
vec3f operator-(vec3f a) { return {-a.x, -a.y, -a.z}; }
vec3f operator+(vec3f a, vec3f b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
vec3f operator-(vec3f a, vec3f b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
vec3f operator*(vec3f a, float b) { return {a.x * b, a.y * b, a.z * b}; }
vec3f operator/(vec3f a, float b) { return {a.x / b, a.y / b, a.z / b}; }
float dot(vec3f a, vec3f b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
vec3f cross(vec3f a, vec3f b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
float length(vec3f a) { return std::sqrt(dot(a, a)); }
vec3f normalize(vec3f a) {
  auto l = length(a);
  return (l != 0) ? a / l : a;
}
vec3f vmin(vec3f a, vec3f b) { return {a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z}; }  // min(a, b) = (a < b) ? a : b
vec3f vmax(vec3f a, vec3f b) { return {a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z}; }
vec3f v3(const float* p) { return {p[0], p[1], p[2]}; }
void  put(float* p, vec3f v) { p[0] = v.x, p[1] = v.y, p[2] = v.z; }
vec3f transform_point(const ythip_frame& f, vec3f b) {  // a.x * b.x + a.y * b.y + a.z * b.z + a.o
  return v3(f.x) * b.x + v3(f.y) * b.y + v3(f.z) * b.z + v3(f.o);
}

// lookat_frame — yocto_math.h:2348-2358
ythip_frame lookat_frame(vec3f eye, vec3f center, vec3f up, bool inv_xz) {
  auto w = normalize(eye - center);
  auto u = normalize(cross(up, w));
  auto v = normalize(cross(w, u));
  if (inv_xz) w = -w, u = -u;
  ythip_frame f;
  put(f.x, u), put(f.y, v), put(f.z, w), put(f.o, eye);
  return f;
}
// "lookat": nine floats over the frame's x, y, z = eye, center, up (yocto_sceneio.cpp:3668-3673)
bool get_lookat(const Json& e, ythip_frame& f) {
  if (!e.find("lookat")) return false;
  float t[9];
  std::memcpy(t, &f, sizeof(t));
  get_floats(e, "lookat", t);
  std::memcpy(&f, t, sizeof(t));
  return true;
}

const char* const kMaterialTypes[] = {"matte", "glossy", "reflective", "transparent", "refractive", "subsurface", "volumetric",
    "gltfpbr"};  // material_type_names, yocto_scene.h:114-116

// ---------------------------------------------------------------------------------------------
// Radiance HDR — the results of stbi_loadf_from_memory(..., 4) (stb_image.h:7031-7197)
// ---------------------------------------------------------------------------------------------
struct ByteReader {  // stb's reader: past the end every byte is 0
  const uint8_t* p;
  const uint8_t* end;
  int            get8() { return p < end ? *p++ : 0; }
  bool           eof() const { return p >= end; }
};
std::string hdr_token(ByteReader& r) {  // stbi__hdr_gettoken: one line, at most 1023 characters kept
  std::string t;
  char        c = (char)r.get8();
  while (!r.eof() && c != '\n') {
    t += c;
    if (t.size() == 1023) {
      while (!r.eof() && r.get8() != '\n') {
      }
      break;
    }
    c = (char)r.get8();
  }
  return t;
}
struct HdrInfo {
  int            width = 0, height = 0;
  const uint8_t* pixels = nullptr;  // where the scanlines start
};
bool hdr_header(const uint8_t* data, size_t size, HdrInfo& info, std::string& why) {
  static const char* sigs[] = {"#?RADIANCE\n", "#?RGBE\n"};  // stbi__hdr_test
  bool               is_hdr = false;
  for (auto sig : sigs) is_hdr = is_hdr || (size >= std::strlen(sig) && std::memcmp(data, sig, std::strlen(sig)) == 0);
  if (!is_hdr) return why = "not a Radiance HDR file", false;
  ByteReader r{data, data + size};
  hdr_token(r);
  bool valid = false;
  while (true) {
    auto t = hdr_token(r);
    if (t.empty()) break;
    if (t == "FORMAT=32-bit_rle_rgbe") valid = true;
  }
  if (!valid) return why = "unsupported HDR format", false;
  auto        t = hdr_token(r);
  const char* s = t.c_str();
  if (std::strncmp(s, "-Y ", 3)) return why = "unsupported HDR data layout", false;
  char* e     = nullptr;
  info.height = (int)std::strtol(s + 3, &e, 10);
  while (*e == ' ') e++;
  if (std::strncmp(e, "+X ", 3)) return why = "unsupported HDR data layout", false;
  info.width = (int)std::strtol(e + 3, nullptr, 10);
  if (info.width <= 0 || info.height <= 0 || info.width > (1 << 24) || info.height > (1 << 24)) return why = "bad HDR size", false;
  // stb's own limit (stbi__mad4sizes_valid: width * height * 4 * sizeof(float) fits an int), and what the file can
  // hold at all: a run-length byte pair covers at most 127 samples
  if ((double)info.width * info.height * 16.0 > 2147483647.0) return why = "HDR image is too large", false;
  if ((double)info.width * info.height * 4.0 > 64.0 * (double)size + 1024.0) return why = "corrupt HDR: truncated", false;
  info.pixels = r.p;
  return true;
}
inline void hdr_texel(float* out, const uint8_t* rgbe) {  // stbi__hdr_convert, req_comp 4
  if (rgbe[3] != 0) {
    float f1 = (float)std::ldexp(1.0f, rgbe[3] - (int)(128 + 8));
    out[0] = rgbe[0] * f1, out[1] = rgbe[1] * f1, out[2] = rgbe[2] * f1;
  } else {
    out[0] = out[1] = out[2] = 0;
  }
  out[3] = 1;
}
bool hdr_decode(const uint8_t* data, size_t size, const HdrInfo& info, float* out, std::string& why) {
  ByteReader r{info.pixels, data + size};
  int        width = info.width, height = info.height;
  auto       flat  = [&](size_t first) {  // the rest of the image as plain RGBE quadruples
    for (size_t k = first; k < (size_t)width * height; k++) {
      uint8_t rgbe[4];
      for (auto& b : rgbe) b = (uint8_t)r.get8();
      hdr_texel(out + 4 * k, rgbe);
    }
  };
  if (width < 8 || width >= 32768) {
    flat(0);
    return true;
  }
  std::vector<uint8_t> scanline((size_t)width * 4);
  for (int j = 0; j < height; j++) {
    int c1 = r.get8(), c2 = r.get8(), len = r.get8();
    if (c1 != 2 || c2 != 2 || (len & 0x80)) {
      // not run-length encoded: stb takes these four bytes as the image's FIRST texel and reads
      // everything after it flat, whichever row it was on
      uint8_t rgbe[4] = {(uint8_t)c1, (uint8_t)c2, (uint8_t)len, (uint8_t)r.get8()};
      hdr_texel(out, rgbe);
      flat(1);
      return true;
    }
    len = (len << 8) | r.get8();
    if (len != width) return why = "corrupt HDR: invalid decoded scanline length", false;
    for (int k = 0; k < 4; k++) {
      int i = 0, nleft;
      while ((nleft = width - i) > 0) {
        int count = r.get8();
        if (count > 128) {
          int value = r.get8();
          count -= 128;
          if (count > nleft) return why = "corrupt HDR: bad RLE data", false;
          for (int z = 0; z < count; z++) scanline[(size_t)(i++) * 4 + k] = (uint8_t)value;
        } else {
          if (count > nleft) return why = "corrupt HDR: bad RLE data", false;
          if (count == 0 && r.eof()) return why = "corrupt HDR: truncated", false;  // (stb would spin on zeros here)
          for (int z = 0; z < count; z++) scanline[(size_t)(i++) * 4 + k] = (uint8_t)r.get8();
        }
      }
    }
    for (int i = 0; i < width; i++) hdr_texel(out + ((size_t)j * width + i) * 4, scanline.data() + (size_t)i * 4);
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// PNG — the results of stbi_load_from_memory(..., 4): RGBA8 whatever the file holds
// (stb_image.h: stbi__parse_png_file, stbi__create_png_image[_raw], stbi__compute_transparency[16],
// stbi__expand_png_palette, stbi__convert_format[16], stbi__convert_16_to_8).  A standard decoder
// plus stb's conversions: grey of depth < 8 is scaled to 0..255, 16-bit samples keep their high
// byte, a tRNS colour key compares the (scaled) sample and gives alpha 0 / 255, palettes take
// their alpha from tRNS, missing alpha is 255.  Chunk CRCs are not checked (stb does not).
// ---------------------------------------------------------------------------------------------
struct PngInfo {
  int                  width = 0, height = 0, depth = 0, color = 0, interlace = 0;
  int                  channels = 0;
  std::vector<uint8_t> idat;
  uint8_t              palette[256][4];
  int                  palette_len = 0;
  bool                 has_key     = false;
  uint16_t             key[3]      = {0, 0, 0};
};
uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

// header only (`with_data` false) or all chunks
bool png_parse(const uint8_t* data, size_t size, PngInfo& png, bool with_data, std::string& why) {
  static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
  if (size < 8 || std::memcmp(data, sig, 8) != 0) return why = "not a PNG file", false;
  size_t at = 8;
  bool   first = true, seen_idat = false;
  for (auto& e : png.palette) e[0] = e[1] = e[2] = 0, e[3] = 255;
  while (true) {
    if (size - at < 8) return why = "corrupt PNG: truncated", false;
    uint32_t       len = be32(data + at), type = be32(data + at + 4);
    const uint8_t* body = data + at + 8;
    if (len > (1u << 30) || size - at - 8 < (size_t)len + 4) return why = "corrupt PNG: bad chunk length", false;
    at += 8 + (size_t)len + 4;
    auto is = [&](const char* t) { return type == be32((const uint8_t*)t); };
    if (first && !is("IHDR")) return why = "corrupt PNG: first chunk is not IHDR", false;
    if (is("IHDR")) {
      if (!first || len != 13) return why = "corrupt PNG: bad IHDR", false;
      first      = false;
      png.width  = (int)be32(body), png.height = (int)be32(body + 4);
      png.depth  = body[8], png.color = body[9];
      png.interlace = body[12];
      if (png.width <= 0 || png.height <= 0 || png.width > (1 << 24) || png.height > (1 << 24)) return why = "bad PNG size", false;
      if (png.depth != 1 && png.depth != 2 && png.depth != 4 && png.depth != 8 && png.depth != 16)
        return why = "corrupt PNG: bad bit depth", false;
      if (png.color > 6 || png.color == 1 || png.color == 5) return why = "corrupt PNG: bad colour type", false;
      if (png.color == 3 && png.depth == 16) return why = "corrupt PNG: bad colour type", false;
      if ((png.color == 2 || png.color == 4 || png.color == 6) && png.depth < 8) return why = "corrupt PNG: bad bit depth", false;
      if (body[10] || body[11] || png.interlace > 1) return why = "corrupt PNG: bad compression / filter / interlace method", false;
      png.channels = png.color == 3 ? 1 : (png.color & 2 ? 3 : 1) + (png.color & 4 ? 1 : 0);
      // stb's limit (the RGBA8 result fits an int) and what a deflate stream of this file's size can hold (1032 : 1)
      if ((double)png.width * png.height * 4.0 > 2147483647.0) return why = "PNG image is too large", false;
      if ((double)png.width * png.height * png.channels * png.depth / 8.0 > 1032.0 * (double)size + 65536.0) return why = "corrupt PNG: not enough pixels", false;
      if (!with_data) return true;
    } else if (is("PLTE")) {
      if (len > 256 * 3 || len % 3) return why = "corrupt PNG: bad PLTE", false;
      png.palette_len = (int)len / 3;
      for (int k = 0; k < png.palette_len; k++) png.palette[k][0] = body[3 * k], png.palette[k][1] = body[3 * k + 1], png.palette[k][2] = body[3 * k + 2];
    } else if (is("tRNS")) {
      if (seen_idat) return why = "corrupt PNG: tRNS after IDAT", false;
      if (png.color == 3) {
        if (png.palette_len == 0 || (int)len > png.palette_len) return why = "corrupt PNG: bad tRNS", false;
        for (uint32_t k = 0; k < len; k++) png.palette[k][3] = body[k];
      } else {
        if (png.color & 4) return why = "corrupt PNG: tRNS with alpha", false;
        int n = png.color & 2 ? 3 : 1;
        if (len != (uint32_t)n * 2) return why = "corrupt PNG: bad tRNS", false;
        png.has_key = true;
        for (int k = 0; k < n; k++) png.key[k] = (uint16_t)(body[2 * k] << 8 | body[2 * k + 1]);
      }
    } else if (is("IDAT")) {
      if (png.color == 3 && png.palette_len == 0) return why = "corrupt PNG: no PLTE", false;
      seen_idat = true;
      png.idat.insert(png.idat.end(), body, body + len);
    } else if (is("IEND")) {
      if (!seen_idat) return why = "corrupt PNG: no IDAT", false;
      return true;
    } else if (!(type & (1u << 29))) {  // an unknown critical chunk
      return why = "PNG chunk not known", false;
    }
  }
}

inline int paeth(int a, int b, int c) {
  int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  if (pa <= pb && pa <= pc) return a;
  return pb <= pc ? b : c;
}

bool png_decode(PngInfo& png, uint8_t* out, std::string& why) {
  const int w = png.width, h = png.height, depth = png.depth, ch = png.channels;
  // geometry of the (up to seven) passes
  struct Pass {
    int x0, y0, dx, dy, pw, ph;
  } passes[7];
  int npass = 0;
  if (!png.interlace) {
    passes[npass++] = {0, 0, 1, 1, w, h};
  } else {
    static const int xo[7] = {0, 4, 0, 2, 0, 1, 0}, yo[7] = {0, 0, 4, 0, 2, 0, 1}, xs[7] = {8, 8, 4, 4, 2, 2, 1}, ys[7] = {8, 8, 8, 4, 4, 2, 2};
    for (int k = 0; k < 7; k++) {
      int pw = (w - xo[k] + xs[k] - 1) / xs[k], ph = (h - yo[k] + ys[k] - 1) / ys[k];
      if (pw > 0 && ph > 0) passes[npass++] = {xo[k], yo[k], xs[k], ys[k], pw, ph};
    }
  }
  auto   row_bytes = [&](int pw) { return ((size_t)pw * ch * depth + 7) / 8; };
  size_t need      = 0;
  for (int k = 0; k < npass; k++) need += (row_bytes(passes[k].pw) + 1) * (size_t)passes[k].ph;
  // inflate
  std::vector<uint8_t> raw(need);
  {
    z_stream zs{};
    if (inflateInit(&zs) != Z_OK) return why = "zlib failed to start", false;
    zs.next_in = png.idat.data(), zs.avail_in = (uInt)png.idat.size();
    zs.next_out = raw.data(), zs.avail_out = (uInt)raw.size();
    if (png.idat.size() > 0xffffffffull || raw.size() > 0xffffffffull) {
      inflateEnd(&zs);
      return why = "PNG too large", false;
    }
    int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    // Z_BUF_ERROR with no room left = the stream holds more than the image needs: fine, as for stb
    if (!(rc == Z_STREAM_END || ((rc == Z_BUF_ERROR || rc == Z_OK) && zs.avail_out == 0))) return why = "corrupt PNG: bad zlib stream", false;
    if (zs.avail_out != 0) return why = "corrupt PNG: not enough pixels", false;
  }
  png.idat.clear();
  png.idat.shrink_to_fit();
  const int  bpp   = std::max(1, ch * depth / 8);  // filter unit
  const int  scale = png.color == 0 ? (depth == 1 ? 255 : depth == 2 ? 85 : depth == 4 ? 17 : 1) : 1;
  const bool key   = png.has_key;
  // the colour key as stb compares it: 8-bit paths use the low byte times the depth scale, 16-bit the full value
  uint8_t key8[3];
  for (int k = 0; k < 3; k++) key8[k] = (uint8_t)((png.key[k] & 255) * (depth < 8 ? scale : 1));
  const uint8_t* src = raw.data();
  std::vector<uint8_t> prior;
  for (int pi = 0; pi < npass; pi++) {
    const Pass& ps = passes[pi];
    size_t      rb = row_bytes(ps.pw);
    prior.assign(rb, 0);
    for (int y = 0; y < ps.ph; y++) {
      int      filter = *src++;
      uint8_t* cur    = const_cast<uint8_t*>(src);
      if (filter > 4) return why = "corrupt PNG: invalid filter", false;
      const uint8_t* up = prior.data();
      const size_t   bp = (size_t)bpp, head = std::min(bp, rb);
      switch (filter) {  // (left of the first pixel and above the first row: zeros)
        case 1:
          for (size_t i = bp; i < rb; i++) cur[i] = (uint8_t)(cur[i] + cur[i - bp]);
          break;
        case 2:
          for (size_t i = 0; i < rb; i++) cur[i] = (uint8_t)(cur[i] + up[i]);
          break;
        case 3:
          for (size_t i = 0; i < head; i++) cur[i] = (uint8_t)(cur[i] + (up[i] >> 1));
          for (size_t i = bp; i < rb; i++) cur[i] = (uint8_t)(cur[i] + ((cur[i - bp] + up[i]) >> 1));
          break;
        case 4:
          for (size_t i = 0; i < head; i++) cur[i] = (uint8_t)(cur[i] + up[i]);  // paeth(0, b, 0) = b
          for (size_t i = bp; i < rb; i++) cur[i] = (uint8_t)(cur[i] + paeth(cur[i - bp], up[i], up[i - bp]));
          break;
        default: break;
      }
      std::memcpy(prior.data(), cur, rb);
      src += rb;
      // samples of this row -> RGBA8
      auto sample = [&](int index) -> int {  // depth <= 8: the sample; 16: its 16-bit value
        if (depth == 8) return cur[index];
        if (depth == 16) return cur[2 * index] << 8 | cur[2 * index + 1];
        int per = 8 / depth, byte = index / per, shift = (per - 1 - index % per) * depth;
        return (cur[byte] >> shift) & ((1 << depth) - 1);
      };
      auto to8 = [&](int s) -> uint8_t { return depth == 16 ? (uint8_t)(s >> 8) : (uint8_t)(s * scale); };
      for (int x = 0; x < ps.pw; x++) {
        uint8_t* px = out + ((size_t)(ps.y0 + y * ps.dy) * w + (size_t)(ps.x0 + x * ps.dx)) * 4;
        switch (png.color) {
          case 0: {
            int     s = sample(x);
            uint8_t g = to8(s);
            px[0] = px[1] = px[2] = g;
            px[3] = key && (depth == 16 ? s == png.key[0] : g == key8[0]) ? 0 : 255;
          } break;
          case 2: {
            int s0 = sample(3 * x), s1 = sample(3 * x + 1), s2 = sample(3 * x + 2);
            px[0] = to8(s0), px[1] = to8(s1), px[2] = to8(s2);
            bool hit = key && (depth == 16 ? (s0 == png.key[0] && s1 == png.key[1] && s2 == png.key[2])
                                           : (px[0] == key8[0] && px[1] == key8[1] && px[2] == key8[2]));
            px[3] = hit ? 0 : 255;
          } break;
          case 3: {
            int s = sample(x);
            if (s >= png.palette_len) return why = "corrupt PNG: palette index out of range", false;
            std::memcpy(px, png.palette[s], 4);
          } break;
          case 4: px[0] = px[1] = px[2] = to8(sample(2 * x)), px[3] = to8(sample(2 * x + 1)); break;
          default: px[0] = to8(sample(4 * x)), px[1] = to8(sample(4 * x + 1)), px[2] = to8(sample(4 * x + 2)), px[3] = to8(sample(4 * x + 3));
        }
      }
    }
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// the scene file
// ---------------------------------------------------------------------------------------------
std::string lower_ext(const std::string& path) {
  auto ext = std::filesystem::u8path(path).extension().generic_u8string();
  return ext;
}
std::string join(const std::string& dir, const std::string& uri) {  // path_join, yocto_sceneio.cpp:188-190
  return (std::filesystem::u8path(dir) / std::filesystem::u8path(uri)).generic_u8string();
}

enum TexKind { TEX_HDR, TEX_PNG, TEX_JPG, TEX_EXR, TEX_BMP, TEX_TGA };
struct TextureFile {
  std::string path;
  TexKind     kind;
  FileBytes   bytes;
  HdrInfo     hdr;
  PngInfo     png;
  ytjpeg::Info jpg;
  ytexr::Info  exr;
  int          width = 0, height = 0;  // whatever the format
  bool         is_float() const { return kind == TEX_HDR || kind == TEX_EXR; }
};

template <typename Fn>
int for_each_parallel(size_t count, int threads, Fn&& fn, std::string& error) {  // parallel_for, yocto_sceneio.cpp:60-90
  if (count == 0) return YTHIP_OK;
  int n = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
  n     = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(n, 1), count));
  std::atomic<size_t> next{0};
  std::atomic<int>    failed{0};
  // Items are claimed in order and a claimed item always runs, so every item before a failing one has run too: the
  // failure with the smallest index is the first one in file order whatever the threads did — the message does not
  // depend on timing (the reference's parallel loader reports whichever failing file a thread reached first).
  std::vector<std::pair<size_t, std::string>> errors((size_t)n, {count, std::string()});
  auto                worker = [&](int t) {
    while (!failed.load(std::memory_order_relaxed)) {
      size_t k = next.fetch_add(1);
      if (k >= count) break;
      std::string why;
      bool        ok = false;
      try {
        ok = fn(k, why);
      } catch (const std::exception& e) {  // (an allocation that fails on a damaged file must not end the process)
        why = std::string("out of resources (") + e.what() + ")";
      }
      if (!ok) {
        errors[(size_t)t] = {k, why};
        failed.store(1);
        break;
      }
    }
  };
  if (n == 1) {
    worker(0);
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < n; t++) pool.emplace_back(worker, t);
    for (auto& t : pool) t.join();
  }
  if (failed.load()) {
    size_t first = count;
    for (auto& e : errors)
      if (e.first < first) first = e.first, error = e.second;
    return YTHIP_ERR_INVALID;
  }
  return YTHIP_OK;
}

}  // namespace

struct ythip_scene_file {
  std::string                    path;
  std::vector<ythip_camera>      cameras;
  std::vector<ythip_instance>    instances;
  std::vector<ythip_environment> environments;
  std::vector<ythip_material>    materials;
  std::vector<ythip_texture>     textures;
  std::vector<ythip_shape>       shapes;  // counts from the files, offsets assigned, missing radius included
  std::vector<std::string>       camera_names, instance_names, environment_names, material_names, texture_names, shape_names;
  std::vector<ythip_ply*>        plys;
  std::vector<char>              default_radius;  // add_missing_radius applies to this shape
  std::vector<TextureFile>       texture_files;
  bool                           missing_camera = false;
  std::string                    copyright;
  ythip_scene                    counts{};
  ~ythip_scene_file() {
    for (auto p : plys) ythip_ply_close(p);
  }
};

namespace {

// the JSON part of load_json_scene (yocto_sceneio.cpp:3648-3786), same keys, same defaults
// (camera_data / material_data / ... {} of yocto_scene.h:83-157)
void parse_scene(const Json& json, ythip_scene_file& f, std::vector<std::string>& shape_uris, std::vector<std::string>& texture_uris) {
  auto identity = [](ythip_frame& fr) {
    fr = ythip_frame{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}};
  };
  // `for (auto& element : json.at(name))`: the items of an array, the values of an object in file order
  // (the reference's json_value is nlohmann::ordered_json), nothing for null; anything else is a conversion error, and
  // so is an element that is not an object (its first get_opt throws)
  std::vector<std::vector<const Json*>> groups;
  auto group = [&](const char* name) -> const std::vector<const Json*>* {
    auto g = json.find(name);
    if (!g) return nullptr;
    groups.emplace_back();
    auto& out = groups.back();
    if (g->kind == Json::Array) {
      for (auto& e : g->items) out.push_back(&e);
    } else if (g->kind == Json::Object) {
      std::vector<std::pair<std::string, const Json*>> ordered;  // a repeated key keeps its first place and its last value
      for (auto& m : g->members) {
        bool seen = false;
        for (auto& have : ordered)
          if (have.first == m.first) have.second = &m.second, seen = true;
        if (!seen) ordered.emplace_back(m.first, &m.second);
      }
      for (auto& m : ordered) out.push_back(m.second);
    } else if (g->kind != Json::Null) {
      throw BadValue{};
    }
    for (auto e : out)
      if (e->kind != Json::Object) throw BadValue{};
    return &out;
  };
  if (auto items = group("cameras"))
    for (auto ep : *items) {
      const Json& e = *ep;
      ythip_camera c{};
      identity(c.frame);
      c.orthographic = 0, c.lens = 0.050f, c.film = 0.036f, c.aspect = 1.500f, c.focus = 10000, c.aperture = 0;
      f.camera_names.emplace_back();
      get_opt(e, "name", f.camera_names.back());
      get_frame(e, "frame", c.frame);
      get_flag(e, "orthographic", c.orthographic);
      get_opt(e, "lens", c.lens);
      get_opt(e, "aspect", c.aspect);
      get_opt(e, "film", c.film);
      get_opt(e, "focus", c.focus);
      get_opt(e, "aperture", c.aperture);
      if (get_lookat(e, c.frame)) {
        c.focus = length(v3(c.frame.x) - v3(c.frame.y));
        c.frame = lookat_frame(v3(c.frame.x), v3(c.frame.y), v3(c.frame.z), false);
      }
      f.cameras.push_back(c);
    }
  if (auto items = group("textures"))
    for (auto ep : *items) {
      const Json& e = *ep;
      ythip_texture t{};
      f.texture_names.emplace_back();
      texture_uris.emplace_back();
      get_opt(e, "name", f.texture_names.back());
      get_opt(e, "uri", texture_uris.back());
      get_flag(e, "linear", t.linear);
      get_flag(e, "nearest", t.nearest);
      get_flag(e, "clamp", t.clamp);
      f.textures.push_back(t);
    }
  if (auto items = group("materials"))
    for (auto ep : *items) {
      const Json& e = *ep;
      ythip_material m{};
      m.type      = YTHIP_MATTE;
      m.color[0] = m.color[1] = m.color[2] = 0;
      m.roughness = 0, m.metallic = 0, m.ior = 1.5f, m.scanisotropy = 0, m.trdepth = 0.01f, m.opacity = 1;
      m.emission_tex = m.color_tex = m.roughness_tex = m.scattering_tex = m.normal_tex = YTHIP_INVALIDID;
      f.material_names.emplace_back();
      get_opt(e, "name", f.material_names.back());
      if (auto j = e.find("type")) {  // an enum by label; anything unknown is the first label
        m.type = YTHIP_MATTE;
        if (j->kind == Json::String)
          for (int k = 0; k < 8; k++)
            if (j->s == kMaterialTypes[k]) m.type = k;
      }
      get_floats(e, "emission", m.emission);
      get_floats(e, "color", m.color);
      get_opt(e, "metallic", m.metallic);
      get_opt(e, "roughness", m.roughness);
      get_opt(e, "ior", m.ior);
      get_opt(e, "trdepth", m.trdepth);
      get_floats(e, "scattering", m.scattering);
      get_opt(e, "scanisotropy", m.scanisotropy);
      get_opt(e, "opacity", m.opacity);
      get_opt(e, "emission_tex", m.emission_tex);
      get_opt(e, "color_tex", m.color_tex);
      get_opt(e, "roughness_tex", m.roughness_tex);
      get_opt(e, "scattering_tex", m.scattering_tex);
      get_opt(e, "normal_tex", m.normal_tex);
      f.materials.push_back(m);
    }
  if (auto items = group("shapes"))
    for (auto ep : *items) {
      const Json& e = *ep;
      f.shape_names.emplace_back();
      shape_uris.emplace_back();
      get_opt(e, "name", f.shape_names.back());
      get_opt(e, "uri", shape_uris.back());
    }
  if (auto items = group("instances"))
    for (auto ep : *items) {
      const Json& e = *ep;
      ythip_instance i{};
      identity(i.frame);
      i.shape = i.material = YTHIP_INVALIDID;
      f.instance_names.emplace_back();
      get_opt(e, "name", f.instance_names.back());
      get_frame(e, "frame", i.frame);
      get_opt(e, "shape", i.shape);
      get_opt(e, "material", i.material);
      if (get_lookat(e, i.frame)) i.frame = lookat_frame(v3(i.frame.x), v3(i.frame.y), v3(i.frame.z), true);
      f.instances.push_back(i);
    }
  if (auto items = group("environments"))
    for (auto ep : *items) {
      const Json& e = *ep;
      ythip_environment env{};
      identity(env.frame);
      env.emission_tex = YTHIP_INVALIDID;
      f.environment_names.emplace_back();
      get_opt(e, "name", f.environment_names.back());
      get_frame(e, "frame", env.frame);
      get_floats(e, "emission", env.emission);
      get_opt(e, "emission_tex", env.emission_tex);
      if (get_lookat(e, env.frame)) env.frame = lookat_frame(v3(env.frame.x), v3(env.frame.y), v3(env.frame.z), true);
      f.environments.push_back(env);
    }
}


// Format 4.0 (load_json_scene_version40, yocto_sceneio.cpp:3025-3373; files without asset.version): every group is an object of NAMED
// elements; instances and materials refer to shapes, materials and textures by name; shapes and textures have no
// entry of their own — they exist because something names them, in the order they are first named (environments,
// then materials, then instances / objects), and their files are <dir>/shapes/<name>.ply, <dir>/textures/<name>.hdr |
// .png (find_path, :3263-3270: the first extension that exists, else the first of the list).  Material types carry
// their old labels ("metallic" = reflective, "volume" = volumetric); lookat frames of instances and environments are
// NOT mirrored (inv_xz false, unlike 4.2).  "objects" with a PLY instance file and subdivs are refused by name.
struct Refused {
  std::string why;
};
void parse_scene40(const Json& json, ythip_scene_file& f, const std::string& dirname, std::vector<std::string>& shape_uris,
    std::vector<std::string>& texture_uris) {
  auto identity = [](ythip_frame& fr) {
    fr = ythip_frame{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}};
  };
  // `for (auto& [key, element] : json.at(name).items())`: members in file order (a repeated key keeps its first place and
  // its last value), array items under their index; an element that is not an object fails at its first value()
  std::vector<std::vector<std::pair<std::string, const Json*>>> groups;
  auto group = [&](const char* name) -> const std::vector<std::pair<std::string, const Json*>>* {
    auto g = json.find(name);
    if (!g) return nullptr;
    groups.emplace_back();
    auto& out = groups.back();
    if (g->kind == Json::Object) {
      for (auto& m : g->members) {
        bool seen = false;
        for (auto& have : out)
          if (have.first == m.first) have.second = &m.second, seen = true;
        if (!seen) out.emplace_back(m.first, &m.second);
      }
    } else if (g->kind == Json::Array) {
      for (size_t k = 0; k < g->items.size(); k++) out.emplace_back(std::to_string(k), &g->items[k]);
    } else if (g->kind != Json::Null) {
      throw BadValue{};  // (items() of a scalar yields the scalar, whose value() throws)
    }
    for (auto& e : out)
      if (e.second->kind != Json::Object) throw BadValue{};
    return &out;
  };
  auto name_of = [](const Json& e, const char* key) -> std::string {
    std::string name;
    get_opt(e, key, name);
    return name;
  };
  auto find_path = [&](const std::string& name, const char* dir, std::initializer_list<const char*> extensions) {
    for (auto ext : extensions) {
      auto rel = (std::filesystem::u8path(dir) / std::filesystem::u8path(name + ext)).generic_u8string();
      std::error_code ec;
      if (std::filesystem::exists(std::filesystem::u8path(join(dirname, rel)), ec)) return rel;
    }
    return (std::filesystem::u8path(dir) / std::filesystem::u8path(name + *extensions.begin())).generic_u8string();
  };
  std::vector<std::string> texture_names, shape_names;  // in order of first mention
  auto get_tex = [&](const Json& e, const char* key, int32_t& value) {
    auto name = name_of(e, key);
    if (name.empty()) return;
    for (size_t k = 0; k < texture_names.size(); k++)
      if (texture_names[k] == name) {
        value = (int32_t)k;
        return;
      }
    texture_names.push_back(name);
    f.textures.push_back(ythip_texture{});
    value = (int32_t)texture_names.size() - 1;
  };
  auto get_shp = [&](const Json& e, const char* key, int32_t& value) {
    auto name = name_of(e, key);
    if (name.empty()) return;
    for (size_t k = 0; k < shape_names.size(); k++)
      if (shape_names[k] == name) {
        value = (int32_t)k;
        return;
      }
    shape_names.push_back(name);
    value = (int32_t)shape_names.size() - 1;
  };
  auto get_mat = [&](const Json& e, const char* key, int32_t& value) {
    auto name = name_of(e, key);
    if (name.empty()) return;
    for (size_t k = 0; k < f.material_names.size(); k++)
      if (f.material_names[k] == name) value = (int32_t)k;  // (a repeated name: the last one, as the map's assignment leaves it)
    bool found = false;
    for (auto& n : f.material_names) found = found || n == name;
    if (!found) throw BadValue{};  // "missing key"
  };
  if (auto items = group("cameras"))
    for (auto& [key, ep] : *items) {
      const Json&  e = *ep;
      ythip_camera c{};
      identity(c.frame);
      c.orthographic = 0, c.lens = 0.050f, c.film = 0.036f, c.aspect = 1.500f, c.focus = 10000, c.aperture = 0;
      f.camera_names.push_back(key);
      get_frame(e, "frame", c.frame);
      get_flag(e, "orthographic", c.orthographic);
      get_flag(e, "ortho", c.orthographic);
      get_opt(e, "lens", c.lens);
      get_opt(e, "aspect", c.aspect);
      get_opt(e, "film", c.film);
      get_opt(e, "focus", c.focus);
      get_opt(e, "aperture", c.aperture);
      if (get_lookat(e, c.frame)) {
        c.focus = length(v3(c.frame.x) - v3(c.frame.y));
        c.frame = lookat_frame(v3(c.frame.x), v3(c.frame.y), v3(c.frame.z), false);
      }
      f.cameras.push_back(c);
    }
  if (auto items = group("environments"))
    for (auto& [key, ep] : *items) {
      const Json&       e = *ep;
      ythip_environment env{};
      identity(env.frame);
      env.emission_tex = YTHIP_INVALIDID;
      f.environment_names.push_back(key);
      get_frame(e, "frame", env.frame);
      get_floats(e, "emission", env.emission);
      get_tex(e, "emission_tex", env.emission_tex);
      if (get_lookat(e, env.frame)) env.frame = lookat_frame(v3(env.frame.x), v3(env.frame.y), v3(env.frame.z), false);
      f.environments.push_back(env);
    }
  static const char* const kTypes40[] = {"matte", "glossy", "metallic", "transparent", "refractive", "subsurface", "volume", "gltfpbr"};
  if (auto items = group("materials"))
    for (auto& [key, ep] : *items) {
      const Json&    e = *ep;
      ythip_material m{};
      m.type = YTHIP_MATTE;
      m.ior = 1.5f, m.trdepth = 0.01f, m.opacity = 1;
      m.emission_tex = m.color_tex = m.roughness_tex = m.scattering_tex = m.normal_tex = YTHIP_INVALIDID;
      f.material_names.push_back(key);
      if (auto j = e.find("type")) {
        m.type = YTHIP_MATTE;
        if (j->kind == Json::String)
          for (int k = 0; k < 8; k++)
            if (j->s == kTypes40[k]) m.type = k;
      }
      get_floats(e, "emission", m.emission);
      get_floats(e, "color", m.color);
      get_opt(e, "metallic", m.metallic);
      get_opt(e, "roughness", m.roughness);
      get_opt(e, "ior", m.ior);
      get_opt(e, "trdepth", m.trdepth);
      get_floats(e, "scattering", m.scattering);
      get_opt(e, "scanisotropy", m.scanisotropy);
      get_opt(e, "opacity", m.opacity);
      get_tex(e, "emission_tex", m.emission_tex);
      get_tex(e, "color_tex", m.color_tex);
      get_tex(e, "roughness_tex", m.roughness_tex);
      get_tex(e, "scattering_tex", m.scattering_tex);
      get_tex(e, "normal_tex", m.normal_tex);
      f.materials.push_back(m);
    }
  for (const char* which : {"instances", "objects"})
    if (auto items = group(which))
      for (auto& [key, ep] : *items) {
        const Json&    e = *ep;
        ythip_instance i{};
        identity(i.frame);
        i.shape = i.material = YTHIP_INVALIDID;
        f.instance_names.push_back(key);
        get_frame(e, "frame", i.frame);
        get_shp(e, "shape", i.shape);
        get_mat(e, "material", i.material);
        if (get_lookat(e, i.frame)) i.frame = lookat_frame(v3(i.frame.x), v3(i.frame.y), v3(i.frame.z), false);
        if (which[0] == 'o' && e.find("instance") && !name_of(e, "instance").empty())
          throw Refused{"object \"" + key + "\" is instanced from a PLY file of frames (not read here)"};
        f.instances.push_back(i);
      }
  if (auto items = group("subdivs"))
    if (!items->empty()) throw Refused{"the scene has subdivs (tesselation is not on this path)"};
  f.texture_names = texture_names, f.shape_names = shape_names;
  for (auto& n : shape_names) shape_uris.push_back(find_path(n, "shapes", {".ply", ".obj"}));
  for (auto& n : texture_names) texture_uris.push_back(find_path(n, "textures", {".hdr", ".exr", ".png", ".jpg"}));
}

}  // namespace

extern "C" {

// load_json_scene up to (not including) the bulk data: scene.json parsed, every shape and texture
// file opened and measured.  `counts` gets every num_* of ythip_scene (pointers null) — what to
// size the pools with.
static int scene_open_impl(const char* path, ythip_scene_file** out, ythip_scene* counts) {
  if (!path || !out || !counts) return fail(YTHIP_ERR_INVALID, "null argument");
  *out = nullptr;
  std::string filename = path;
  auto        ext      = lower_ext(filename);
  if (ext != ".json" && ext != ".JSON")
    return fail(YTHIP_ERR_INVALID, "unsupported format " + filename + " (only the builtin JSON scene format is read here)");
  FileBytes text;
  if (!text.load(filename)) return fail(YTHIP_ERR_INVALID, "cannot open " + filename);
  Json       json;
  JsonParser parser{(const char*)text.data.data(), (const char*)text.data.data() + text.data.size()};
  if (!parser.document(json)) return fail(YTHIP_ERR_INVALID, "cannot parse " + filename);
  // version gate (yocto_sceneio.cpp:3625-3654): no asset.version = format 4.0 (read below); "4.1", a third layout that
  // no file of the reference's corpus uses, is refused by name
  auto        asset   = json.find("asset");
  const Json* version = asset ? asset->find("version") : nullptr;
  const bool  old     = !version;
  if (version && version->kind == Json::String && version->s == "4.1")
    return fail(YTHIP_ERR_INVALID, "cannot load " + filename + ": format 4.1 is not read here");
  if (!old && (version->kind != Json::String || (version->s != "4.2" && version->s != "5.0")))
    return fail(YTHIP_ERR_INVALID, "cannot parse " + filename);
  if (!old)
    if (auto s = json.find("subdivs"); s && (s->kind == Json::Array ? !s->items.empty() : s->kind == Json::Object && !s->members.empty()))
      return fail(YTHIP_ERR_INVALID, "cannot load " + filename + ": the scene has subdivs (tesselation is not on this path)");
  auto f  = std::make_unique<ythip_scene_file>();
  f->path = filename;
  auto dirname = std::filesystem::u8path(filename).parent_path().generic_u8string();
  std::vector<std::string> shape_uris, texture_uris;
  try {
    if (asset) get_opt(*asset, "copyright", f->copyright);
    if (old) parse_scene40(json, *f, dirname, shape_uris, texture_uris);
    else parse_scene(json, *f, shape_uris, texture_uris);
  } catch (const BadValue&) {
    return fail(YTHIP_ERR_INVALID, "cannot parse " + filename);
  } catch (const Refused& r) {
    return fail(YTHIP_ERR_INVALID, "cannot load " + filename + ": " + r.why);
  }
  auto dependent = [&](const std::string& why) { return fail(YTHIP_ERR_INVALID, "cannot load " + filename + " since " + why); };

  // shapes: map and measure (load_shape's format switch, yocto_sceneio.cpp:1008-1016: PLY only here)
  size_t nshapes = shape_uris.size();
  f->plys.assign(nshapes, nullptr);
  f->shapes.assign(nshapes, ythip_shape{});
  f->default_radius.assign(nshapes, 0);
  std::string error;
  int rc = for_each_parallel(nshapes, 0, [&](size_t k, std::string& why) {
    auto file = join(dirname, shape_uris[k]);
    auto e    = lower_ext(file);
    if (e != ".ply" && e != ".PLY") {
      why = "unsupported format " + file + " (shapes are read from PLY here)";
      return false;
    }
    if (ythip_ply_open(file.c_str(), &f->plys[k], &f->shapes[k]) != YTHIP_OK) {
      why = ythip_io_last_error();
      return false;
    }
    return true;
  }, error);
  if (rc) return dependent(error);
  // textures: read and measure
  size_t ntex = texture_uris.size();
  f->texture_files.resize(ntex);
  rc = for_each_parallel(ntex, 0, [&](size_t k, std::string& why) {
    auto& tf = f->texture_files[k];
    tf.path  = join(dirname, texture_uris[k]);
    auto e   = lower_ext(tf.path);
    if (e == ".hdr" || e == ".HDR") tf.kind = TEX_HDR;
    else if (e == ".png" || e == ".PNG") tf.kind = TEX_PNG;
    else if (e == ".jpg" || e == ".JPG" || e == ".jpeg" || e == ".JPEG") tf.kind = TEX_JPG;
    else if (e == ".exr" || e == ".EXR") tf.kind = TEX_EXR;
    else if (e == ".bmp" || e == ".BMP") tf.kind = TEX_BMP;
    else if (e == ".tga" || e == ".TGA") tf.kind = TEX_TGA;
    else {
      why = "unsupported format " + tf.path + " (textures are read from Radiance HDR, OpenEXR, PNG, JPEG, BMP and TGA here)";
      return false;
    }
    if (!tf.bytes.load(tf.path)) {
      why = "cannot open " + tf.path;
      return false;
    }
    std::string    detail;
    const uint8_t* bytes = tf.bytes.data.data();
    const size_t   count = tf.bytes.data.size();
    // stbi_load goes by what the file IS, not by what it is called (load_texture only picks 8-bit or float by the
    // extension): a PNG called .jpg is read as a PNG, and so on — in stb_image's order of tests (png, bmp, gif, psd, pic,
    // jpeg, pnm, hdr, tga: stb_image.h:1117-1180)
    if (tf.kind != TEX_HDR && tf.kind != TEX_EXR) {
      static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
      auto                 starts     = [&](const char* t) { return count >= std::strlen(t) && std::memcmp(bytes, t, std::strlen(t)) == 0; };
      // stb_image's jpeg test reads the SOI through its get_marker, which skips 0xFF fill bytes in front of the marker code:
      // FF FF D8 is a JPEG to the reference
      auto jpeg_soi = [](const uint8_t* b, size_t n) {
        if (n < 2 || b[0] != 0xff) return false;
        size_t i = 1;
        while (i < n && b[i] == 0xff) i++;
        return i < n && b[i] == 0xd8;
      };
      const char*          other      = nullptr;
      if (count >= 8 && std::memcmp(bytes, png_sig, 8) == 0) tf.kind = TEX_PNG;
      else if (ytimg::bmp::test(bytes, count)) tf.kind = TEX_BMP;
      else if (starts("GIF87a") || starts("GIF89a")) other = "GIF";
      else if (starts("8BPS")) other = "PSD";
      else if (count >= 4 && bytes[0] == 0x53 && bytes[1] == 0x80 && bytes[2] == 0xf6 && bytes[3] == 0x34) other = "Softimage PIC";
      else if (jpeg_soi(bytes, count)) tf.kind = TEX_JPG;
      else if (starts("P5") || starts("P6")) other = "PNM";
      else if (starts("#?RADIANCE\n") || starts("#?RGBE\n")) other = "Radiance HDR (as an 8-bit texture)";
      else if (ytimg::tga::test(bytes, count)) tf.kind = TEX_TGA;
      else {
        why = "cannot raed " + tf.path + " (unknown image type)";
        return false;
      }
      if (other) {
        why = "unsupported format " + tf.path + " (" + other + " content is not read here)";
        return false;
      }
    }
    bool ok = false;
    switch (tf.kind) {
      case TEX_HDR: ok = hdr_header(bytes, count, tf.hdr, detail), tf.width = tf.hdr.width, tf.height = tf.hdr.height; break;
      case TEX_PNG: ok = png_parse(bytes, count, tf.png, false, detail), tf.width = tf.png.width, tf.height = tf.png.height; break;
      case TEX_JPG: ok = ytjpeg::header(bytes, count, tf.jpg, detail), tf.width = tf.jpg.width, tf.height = tf.jpg.height; break;
      case TEX_EXR: ok = ytexr::header(bytes, count, tf.exr, detail), tf.width = tf.exr.width, tf.height = tf.exr.height; break;
      case TEX_BMP: {
        ytimg::Size sz;
        ok = ytimg::bmp::header(bytes, count, sz, detail), tf.width = sz.width, tf.height = sz.height;
      } break;
      case TEX_TGA: {
        ytimg::Size sz;
        ok = ytimg::tga::header(bytes, count, sz, detail), tf.width = sz.width, tf.height = sz.height;
      } break;
    }
    if (!ok) {
      why = "cannot raed " + tf.path + " (" + detail + ")";  // (the reference's spelling, load_texture's read_error)
      return false;
    }
    return true;
  }, error);
  if (rc) return dependent(error);

  // offsets and totals, in the order the flatten step appends (shape by shape, texture by texture)
  ythip_scene c{};
  for (size_t k = 0; k < nshapes; k++) {
    auto& s = f->shapes[k];
    if ((s.num_points || s.num_lines) && !s.num_radius) {  // add_missing_radius, yocto_sceneio.cpp:2142-2148
      f->default_radius[k] = 1;
      s.num_radius         = s.num_positions;
    }
    auto take = [](int32_t n, int64_t& total) {
      int64_t off = n ? total : -1;
      total += n;
      return off;
    };
    s.points_offset    = take(s.num_points, c.num_points);
    s.lines_offset     = take(s.num_lines, c.num_lines);
    s.triangles_offset = take(s.num_triangles, c.num_triangles);
    s.quads_offset     = take(s.num_quads, c.num_quads);
    s.positions_offset = take(s.num_positions, c.num_positions);
    s.normals_offset   = take(s.num_normals, c.num_normals);
    s.texcoords_offset = take(s.num_texcoords, c.num_texcoords);
    s.colors_offset    = take(s.num_colors, c.num_colors);
    s.radius_offset    = take(s.num_radius, c.num_radius);
  }
  for (size_t k = 0; k < ntex; k++) {
    auto& t  = f->textures[k];
    auto& tf = f->texture_files[k];
    bool  hdr = tf.is_float();
    t.width = tf.width, t.height = tf.height;
    t.linear   = hdr ? 1 : 0;  // load_texture overwrites what the json said (yocto_sceneio.cpp:1819, :1830)
    t.is_float = hdr ? 1 : 0;
    int64_t& total = hdr ? c.num_pixelsf : c.num_pixelsb;
    t.offset       = total;
    total += (int64_t)t.width * t.height;
  }
  f->missing_camera  = f->cameras.empty();  // add_missing_camera, :2119-2139: needs the positions, done in read
  c.num_cameras      = (int32_t)f->cameras.size() + (f->missing_camera ? 1 : 0);
  c.num_instances    = (int32_t)f->instances.size();
  c.num_environments = (int32_t)f->environments.size();
  c.num_shapes       = (int32_t)nshapes;
  c.num_textures     = (int32_t)ntex;
  c.num_materials    = (int32_t)f->materials.size();
  f->counts          = c;
  *counts            = c;
  *out               = f.release();
  return YTHIP_OK;
}

// Fills the pools `dst` points to (every pointer of ythip_scene, writable, sized by the counts of
// ythip_scene_open; a pool whose count is 0 may be null): records, converted shapes, decoded
// textures.  `threads` <= 0: one per hardware thread.
static int scene_read_impl(ythip_scene_file* f, const ythip_scene* dst, int threads) {
  if (!f || !dst) return fail(YTHIP_ERR_INVALID, "null argument");
  auto& c       = f->counts;
  auto  missing = [&](const void* p, int64_t n) { return n > 0 && !p; };
  if (missing(dst->cameras, c.num_cameras) || missing(dst->instances, c.num_instances) || missing(dst->environments, c.num_environments) ||
      missing(dst->shapes, c.num_shapes) || missing(dst->textures, c.num_textures) || missing(dst->materials, c.num_materials) ||
      missing(dst->points, c.num_points) || missing(dst->lines, c.num_lines) || missing(dst->triangles, c.num_triangles) ||
      missing(dst->quads, c.num_quads) || missing(dst->positions, c.num_positions) || missing(dst->normals, c.num_normals) ||
      missing(dst->texcoords, c.num_texcoords) || missing(dst->colors, c.num_colors) || missing(dst->radius, c.num_radius) ||
      missing(dst->pixelsf, c.num_pixelsf) || missing(dst->pixelsb, c.num_pixelsb))
    return fail(YTHIP_ERR_INVALID, "ythip_scene_read: a pool with a non-zero count is null");
  auto dependent = [&](const std::string& why) { return fail(YTHIP_ERR_INVALID, "cannot load " + f->path + " since " + why); };
  auto copy      = [](const void* to, const auto& from) {
    if (!from.empty()) std::memcpy(const_cast<void*>(to), from.data(), from.size() * sizeof(from[0]));
  };
  copy(dst->instances, f->instances);
  copy(dst->environments, f->environments);
  copy(dst->materials, f->materials);
  copy(dst->textures, f->textures);
  copy(dst->shapes, f->shapes);
  copy(dst->cameras, f->cameras);

  auto W = [](const auto* p) { return const_cast<std::remove_const_t<std::remove_pointer_t<decltype(p)>>*>(p); };
  std::string error;
  int rc = for_each_parallel(f->shapes.size(), threads, [&](size_t k, std::string& why) {
    auto& s  = f->shapes[k];
    auto  at = [](auto* pool, int64_t offset, int width) { return offset < 0 ? nullptr : pool + offset * width; };
    bool  fill_radius = f->default_radius[k] != 0;
    if (ythip_ply_read(f->plys[k], 1, at(W(dst->positions), s.positions_offset, 3), at(W(dst->normals), s.normals_offset, 3),
            at(W(dst->texcoords), s.texcoords_offset, 2), at(W(dst->colors), s.colors_offset, 4),
            fill_radius ? nullptr : at(W(dst->radius), s.radius_offset, 1), at(W(dst->points), s.points_offset, 1),
            at(W(dst->lines), s.lines_offset, 2), at(W(dst->triangles), s.triangles_offset, 3), at(W(dst->quads), s.quads_offset, 4)) != YTHIP_OK) {
      why = ythip_io_last_error();
      return false;
    }
    if (fill_radius) {
      float* r = W(dst->radius) + s.radius_offset;
      for (int32_t i = 0; i < s.num_radius; i++) r[i] = 0.001f;
    }
    return true;
  }, error);
  if (rc) return dependent(error);
  rc = for_each_parallel(f->texture_files.size(), threads, [&](size_t k, std::string& why) {
    auto&       tf = f->texture_files[k];
    auto&       t  = f->textures[k];
    std::string detail;
    bool        ok;
    if (tf.kind == TEX_HDR) {
      ok = hdr_decode(tf.bytes.data.data(), tf.bytes.data.size(), tf.hdr, W(dst->pixelsf) + t.offset * 4, detail);
    } else if (tf.kind == TEX_EXR) {
      ok = ytexr::decode(tf.bytes.data.data(), tf.bytes.data.size(), tf.exr, W(dst->pixelsf) + t.offset * 4, detail);
    } else if (tf.kind == TEX_JPG) {
      ok = ytjpeg::decode(tf.bytes.data.data(), tf.bytes.data.size(), W(dst->pixelsb) + t.offset * 4, detail);
    } else if (tf.kind == TEX_BMP) {
      ok = ytimg::bmp::decode(tf.bytes.data.data(), tf.bytes.data.size(), W(dst->pixelsb) + t.offset * 4, detail);
    } else if (tf.kind == TEX_TGA) {
      ok = ytimg::tga::decode(tf.bytes.data.data(), tf.bytes.data.size(), W(dst->pixelsb) + t.offset * 4, detail);
    } else {
      PngInfo png;  // (a second read of the same handle decodes again: the parsed chunks are not kept)
      ok = png_parse(tf.bytes.data.data(), tf.bytes.data.size(), png, true, detail) && png_decode(png, W(dst->pixelsb) + t.offset * 4, detail);
    }
    if (!ok) why = "cannot raed " + tf.path + " (" + detail + ")";
    return ok;
  }, error);
  if (rc) return dependent(error);

  if (f->missing_camera) {  // add_missing_camera (yocto_sceneio.cpp:2119-2139) with compute_bounds (yocto_scene.cpp:718-730)
    const float inf = std::numeric_limits<float>::max();
    std::vector<vec3f> lo(f->shapes.size(), vec3f{inf, inf, inf}), hi(f->shapes.size(), vec3f{-inf, -inf, -inf});
    for (size_t k = 0; k < f->shapes.size(); k++) {
      auto&        s = f->shapes[k];
      const float* p = dst->positions + (s.positions_offset < 0 ? 0 : s.positions_offset * 3);
      for (int32_t i = 0; i < s.num_positions; i++) lo[k] = vmin(lo[k], v3(p + 3 * i)), hi[k] = vmax(hi[k], v3(p + 3 * i));
    }
    vec3f bmin{inf, inf, inf}, bmax{-inf, -inf, -inf};
    for (auto& inst : f->instances) {
      if (inst.shape < 0 || (size_t)inst.shape >= f->shapes.size()) return dependent("an instance without a shape and no camera to frame the scene with");
      vec3f a = lo[(size_t)inst.shape], b = hi[(size_t)inst.shape];
      vec3f corners[8] = {{a.x, a.y, a.z}, {a.x, a.y, b.z}, {a.x, b.y, a.z}, {a.x, b.y, b.z}, {b.x, a.y, a.z}, {b.x, a.y, b.z}, {b.x, b.y, a.z},
          {b.x, b.y, b.z}};
      vec3f tmin{inf, inf, inf}, tmax{-inf, -inf, -inf};
      for (auto& corner : corners) {
        auto q = transform_point(inst.frame, corner);
        tmin = vmin(tmin, q), tmax = vmax(tmax, q);
      }
      bmin = vmin(bmin, tmin), bmax = vmax(bmax, tmax);
    }
    ythip_camera cam{};
    cam.orthographic = 0, cam.film = 0.036f, cam.aspect = (float)16 / (float)9, cam.aperture = 0, cam.lens = 0.050f;
    auto  center      = (bmax + bmin) / 2;
    auto  bbox_radius = length(bmax - bmin) / 2;
    vec3f camera_dir{0, 0, 1};
    auto  camera_dist = bbox_radius * cam.lens / (cam.film / cam.aspect);
    camera_dist *= 2.0f;
    auto from = camera_dir * camera_dist + center;
    cam.frame = lookat_frame(from, center, vec3f{0, 1, 0}, false);
    cam.focus = length(from - center);
    std::memcpy(W(dst->cameras), &cam, sizeof(cam));
  }
  return YTHIP_OK;
}

// (nothing thrown inside — std::filesystem, an allocation sized by a damaged file — may cross the C boundary)
int ythip_scene_open(const char* path, ythip_scene_file** out, ythip_scene* counts) {
  try {
    return scene_open_impl(path, out, counts);
  } catch (const std::exception& e) {
    if (out) *out = nullptr;
    return fail(YTHIP_ERR_INVALID, std::string("cannot load ") + (path ? path : "") + " (" + e.what() + ")");
  }
}
int ythip_scene_read(ythip_scene_file* f, const ythip_scene* dst, int threads) {
  try {
    return scene_read_impl(f, dst, threads);
  } catch (const std::exception& e) {
    return fail(YTHIP_ERR_INVALID, std::string("cannot load ") + (f ? f->path : std::string()) + " (" + e.what() + ")");
  }
}

int32_t ythip_scene_find_camera(const ythip_scene_file* f, const char* name) {  // find_camera, yocto_scene.cpp:656-675
  if (!f) return YTHIP_INVALIDID;
  size_t n = f->missing_camera ? 1 : f->cameras.size();
  if (n == 0) return YTHIP_INVALIDID;
  if (f->missing_camera) return 0;
  if (name && *name)
    for (size_t k = 0; k < n; k++)
      if (f->camera_names[k] == name) return (int32_t)k;
  for (const char* fallback : {"default", "camera", "camera0", "camera1"})
    for (size_t k = 0; k < n; k++)
      if (f->camera_names[k] == fallback) return (int32_t)k;
  return 0;
}

const char* ythip_scene_name(const ythip_scene_file* f, int what, int32_t index) {
  if (!f || index < 0) return nullptr;
  const std::vector<std::string>* names[] = {&f->camera_names, &f->instance_names, &f->environment_names, &f->shape_names, &f->texture_names,
      &f->material_names};
  if (what < 0 || what > 5) return nullptr;
  if (what == 0 && f->missing_camera) return index == 0 ? "camera" : nullptr;
  return (size_t)index < names[what]->size() ? (*names[what])[(size_t)index].c_str() : nullptr;
}

void ythip_scene_close(ythip_scene_file* f) { delete f; }

// The whole of load_scene for a device context: files -> pinned staging pools -> HBM.  After it the
// scene is resident exactly as after ythip_upload_scene of the reference loader's flattened
// scene_data; `staged` (optional) receives the pools for make_trace_bvh / make_trace_lights.
int ythip_load_scene(ythip_ctx* ctx, const char* path, int threads, ythip_scene* staged) {
  if (!ctx) return fail(YTHIP_ERR_INVALID, "null argument");
  ythip_scene_file* f = nullptr;
  ythip_scene       counts{}, pools{};
  int               rc = ythip_scene_open(path, &f, &counts);
  if (rc) return rc;
  rc = ythip_scene_staging(ctx, &counts, &pools);
  if (rc == YTHIP_OK) {
    rc = ythip_scene_read(f, &pools, threads);
    if (rc) {  // half-filled pools must not be uploadable: the staging is replaced by an empty one (the pinned pools are kept)
      std::string why = ythip_io_last_error();
      ythip_scene none{}, unused{};
      (void)ythip_scene_staging(ctx, &none, &unused);
      fail(rc, why);
    }
  } else {
    fail(rc, std::string("ythip_scene_staging: ") + ythip_last_error(ctx));
  }
  ythip_scene_close(f);
  if (rc) return rc;
  rc = ythip_upload_scene_staged(ctx);
  if (rc) return fail(rc, std::string("ythip_upload_scene_staged: ") + ythip_last_error(ctx));
  if (staged) *staged = pools;
  return YTHIP_OK;
}

}  // extern "C"
