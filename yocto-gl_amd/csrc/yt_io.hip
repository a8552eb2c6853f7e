// yt_io.hip — the two file formats either side of the hot path, host code:
//
//   * trace_params <-> JSON, the reference's parameter files (yocto_sceneio.cpp:5815-5852,
//     load / save / update_trace_params :5933-5945): same keys, same enum labels, "update"
//     semantics on load (a key that is absent keeps the value already in the struct).
//   * PLY -> flat pools.  The reference's loader (load_ply, yocto_modelio.cpp:487-740, then
//     load_shape's getters, yocto_sceneio.cpp:1008-1033 with yocto_modelio.h:548-800) builds a
//     ply_model of per-property vectors, then per-shape vectors, then the application flattens
//     them: three generations of copies.  Here a file is opened (mapped), its header and — for
//     face / line lists — its list sizes are scanned once to get the COUNTS, the caller sizes the
//     pinned staging pools of ythip_scene_staging from the counts of all its shapes, and
//     ythip_ply_read converts the properties straight into the pools at the shape's offsets: the
//     bytes the reference's load_shape would have produced, tested against it
//     (tests/test_io.py).
//
// No device code here; the file is a .hip unit only so that the one build rule covers it.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <string>
#include <vector>

#include "../../include/ythip.h"

namespace {

thread_local std::string g_io_error;
int io_fail(int code, const std::string& msg) {
  g_io_error = msg;
  return code;
}

// ---------------------------------------------------------------------------------------------
// JSON: one flat object of scalars (what a trace_params file is).  Nested values are skipped.
// ---------------------------------------------------------------------------------------------
struct JsonCursor {
  const char* p;
  const char* end;
  void ws() {
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
  }
  bool lit(const char* s) {
    size_t n = std::strlen(s);
    if ((size_t)(end - p) >= n && std::memcmp(p, s, n) == 0) {
      p += n;
      return true;
    }
    return false;
  }
  bool string(std::string& out) {
    ws();
    if (p >= end || *p != '"') return false;
    p++;
    out.clear();
    while (p < end && *p != '"') {
      if (*p == '\\') {
        if (++p >= end) return false;
        switch (*p) {
          case 'n': out += '\n'; break;
          case 't': out += '\t'; break;
          case 'r': out += '\r'; break;
          case 'b': out += '\b'; break;
          case 'f': out += '\f'; break;
          case 'u':  // (labels and keys are ASCII: a \uXXXX escape is kept as '?')
            if (end - p < 5) return false;
            p += 4;
            out += '?';
            break;
          default: out += *p;
        }
        p++;
      } else {
        out += *p++;
      }
    }
    if (p >= end) return false;
    p++;
    return true;
  }
  bool skip_value() {  // any JSON value
    ws();
    if (p >= end) return false;
    if (*p == '"') {
      std::string s;
      return string(s);
    }
    if (*p == '{' || *p == '[') {
      char open = *p, close = open == '{' ? '}' : ']';
      int  depth = 0;
      while (p < end) {
        if (*p == '"') {
          std::string s;
          if (!string(s)) return false;
          continue;
        }
        if (*p == open) depth++;
        if (*p == close && --depth == 0) {
          p++;
          return true;
        }
        p++;
      }
      return false;
    }
    while (p < end && *p != ',' && *p != '}' && *p != ']') p++;
    return true;
  }
};

enum { J_NUMBER, J_BOOL, J_STRING };
struct JsonScalar {
  int         kind;
  double      number = 0;
  bool        integral = false;
  long long   inumber = 0;
  unsigned long long unumber = 0;
  bool        boolean = false;
  std::string text;
};
bool parse_scalar(JsonCursor& c, JsonScalar& v, bool& is_scalar) {
  c.ws();
  is_scalar = true;
  if (c.p >= c.end) return false;
  if (*c.p == '"') {
    v.kind = J_STRING;
    return c.string(v.text);
  }
  if (c.lit("true")) {
    v.kind = J_BOOL, v.boolean = true;
    return true;
  }
  if (c.lit("false")) {
    v.kind = J_BOOL, v.boolean = false;
    return true;
  }
  if (*c.p == '-' || (*c.p >= '0' && *c.p <= '9')) {
    const char* s = c.p;
    char*       e = nullptr;
    v.kind        = J_NUMBER;
    v.number      = std::strtod(s, &e);
    if (e == s) return false;
    std::string tok(s, (size_t)(e - s));
    v.integral = tok.find_first_of(".eE") == std::string::npos;
    if (v.integral) {
      v.inumber = std::strtoll(tok.c_str(), nullptr, 10);
      v.unumber = tok[0] == '-' ? (unsigned long long)v.inumber : std::strtoull(tok.c_str(), nullptr, 10);
    }
    c.p = e;
    return true;
  }
  is_scalar = false;  // null, object, array: not something a trace_params key holds
  return c.skip_value();
}

const char* const kSamplers[]    = {"path", "pathdirect", "pathmis", "pathtest", "naive", "eyelight", "diagram", "furnace",
       "falsecolor"};  // trace_sampler_labels, yocto_trace.h:245-254
const char* const kFalsecolors[] = {"position", "normal", "frontfacing", "gnormal", "gfrontfacing", "texcoord", "mtype", "color",
    "emission", "roughness", "opacity", "metallic", "delta", "instance", "shape", "material", "element",
    "highlight"};  // trace_falsecolor_labels, yocto_trace.h:257-275

template <size_t N>
bool label_to_enum(const char* const (&labels)[N], const std::string& s, int32_t& out) {
  for (size_t k = 0; k < N; k++)
    if (s == labels[k]) {
      out = (int32_t)k;
      return true;
    }
  return false;
}

}  // namespace

namespace ytio {
int fail(int code, const std::string& msg) { return io_fail(code, msg); }  // for yt_sceneio.hip
}

extern "C" {

const char* ythip_io_last_error(void) { return g_io_error.c_str(); }

// trace_params{} — yocto_trace.h:95-113
void ythip_params_default(ythip_params* p) {
  if (!p) return;
  *p            = {};
  p->camera     = 0;
  p->resolution = 1280;
  p->sampler    = YTHIP_SAMPLER_PATH;
  p->falsecolor = YTHIP_FC_COLOR;
  p->samples    = 512;
  p->bounces    = 8;
  p->clamp      = 10;
  p->seed       = 961748941ull;  // trace_default_seed
  p->pratio     = 8;
  p->batch      = 1;
}

// from_json(json, trace_params&) — yocto_sceneio.cpp:5834-5852: every key is optional
// (json.value(key, current)); enums travel as their labels; unknown keys are ignored.
int ythip_params_from_json(const char* text, int64_t length, ythip_params* p) {
  if (!text || !p) return io_fail(YTHIP_ERR_INVALID, "null argument");
  if (length < 0) length = (int64_t)std::strlen(text);
  JsonCursor c{text, text + length};
  c.ws();
  if (c.p >= c.end || *c.p != '{') return io_fail(YTHIP_ERR_INVALID, "trace_params json: an object is expected");
  c.p++;
  c.ws();
  if (c.p < c.end && *c.p == '}') return YTHIP_OK;
  while (true) {
    std::string key;
    if (!c.string(key)) return io_fail(YTHIP_ERR_INVALID, "trace_params json: a key is expected");
    c.ws();
    if (c.p >= c.end || *c.p != ':') return io_fail(YTHIP_ERR_INVALID, "trace_params json: ':' expected after \"" + key + "\"");
    c.p++;
    JsonScalar v;
    bool       scalar = false;
    if (!parse_scalar(c, v, scalar)) return io_fail(YTHIP_ERR_INVALID, "trace_params json: bad value of \"" + key + "\"");
    auto bad = [&](const char* want) { return io_fail(YTHIP_ERR_INVALID, "trace_params json: \"" + key + "\" must be " + want); };
    auto as_int = [&](int32_t& out) {
      if (!scalar || v.kind != J_NUMBER) return false;
      out = v.integral ? (int32_t)v.inumber : (int32_t)v.number;
      return true;
    };
    auto as_bool = [&](int32_t& out) {
      if (!scalar || v.kind != J_BOOL) return false;
      out = v.boolean ? 1 : 0;
      return true;
    };
    if (key == "camera") { if (!as_int(p->camera)) return bad("a number"); }
    else if (key == "resolution") { if (!as_int(p->resolution)) return bad("a number"); }
    else if (key == "samples") { if (!as_int(p->samples)) return bad("a number"); }
    else if (key == "bounces") { if (!as_int(p->bounces)) return bad("a number"); }
    else if (key == "pratio") { if (!as_int(p->pratio)) return bad("a number"); }
    else if (key == "batch") { if (!as_int(p->batch)) return bad("a number"); }
    else if (key == "clamp") {
      if (!scalar || v.kind != J_NUMBER) return bad("a number");
      p->clamp = (float)v.number;
    } else if (key == "seed") {
      if (!scalar || v.kind != J_NUMBER) return bad("a number");
      p->seed = v.integral ? (uint64_t)v.unumber : (uint64_t)v.number;
    } else if (key == "sampler") {
      if (!scalar || v.kind != J_STRING || !label_to_enum(kSamplers, v.text, p->sampler)) return bad("a sampler label");
    } else if (key == "falsecolor") {
      if (!scalar || v.kind != J_STRING || !label_to_enum(kFalsecolors, v.text, p->falsecolor)) return bad("a falsecolor label");
    } else if (key == "nocaustics") { if (!as_bool(p->nocaustics)) return bad("a boolean"); }
    else if (key == "envhidden") { if (!as_bool(p->envhidden)) return bad("a boolean"); }
    else if (key == "tentfilter") { if (!as_bool(p->tentfilter)) return bad("a boolean"); }
    else if (key == "embreebvh") { if (!as_bool(p->embreebvh)) return bad("a boolean"); }
    else if (key == "highqualitybvh") { if (!as_bool(p->highqualitybvh)) return bad("a boolean"); }
    else if (key == "noparallel") { if (!as_bool(p->noparallel)) return bad("a boolean"); }
    else if (key == "denoise") { if (!as_bool(p->denoise)) return bad("a boolean"); }
    else if (key == "fastmath") { if (!as_bool(p->fastmath)) return bad("a boolean"); }  // (this library's key; the reference ignores it)
    c.ws();
    if (c.p < c.end && *c.p == ',') {
      c.p++;
      continue;
    }
    if (c.p < c.end && *c.p == '}') return YTHIP_OK;
    return io_fail(YTHIP_ERR_INVALID, "trace_params json: ',' or '}' expected");
  }
}

// to_json(json, trace_params) — yocto_sceneio.cpp:5815-5833: the same keys in the same order.
// Returns the length of the text (without the terminator); writes at most `capacity` bytes.
int64_t ythip_params_to_json(const ythip_params* p, char* buffer, int64_t capacity) {
  if (!p) return -1;
  if (p->sampler < 0 || p->sampler > YTHIP_SAMPLER_FALSECOLOR || p->falsecolor < 0 || p->falsecolor > YTHIP_FC_HIGHLIGHT) {
    io_fail(YTHIP_ERR_INVALID, "bad enum value");  // (the reference throws std::invalid_argument)
    return -1;
  }
  char clamp[64];
  std::snprintf(clamp, sizeof(clamp), "%.9g", (double)p->clamp);  // round-trips a float
  if (!std::strpbrk(clamp, ".eEn")) std::strcat(clamp, ".0");
  auto        b = [](int32_t v) { return v ? "true" : "false"; };
  std::string s = "{\n";
  s += "  \"camera\": " + std::to_string(p->camera) + ",\n";
  s += "  \"resolution\": " + std::to_string(p->resolution) + ",\n";
  s += std::string("  \"sampler\": \"") + kSamplers[p->sampler] + "\",\n";
  s += std::string("  \"falsecolor\": \"") + kFalsecolors[p->falsecolor] + "\",\n";
  s += "  \"samples\": " + std::to_string(p->samples) + ",\n";
  s += "  \"bounces\": " + std::to_string(p->bounces) + ",\n";
  s += std::string("  \"clamp\": ") + clamp + ",\n";
  s += std::string("  \"nocaustics\": ") + b(p->nocaustics) + ",\n";
  s += std::string("  \"envhidden\": ") + b(p->envhidden) + ",\n";
  s += std::string("  \"tentfilter\": ") + b(p->tentfilter) + ",\n";
  s += "  \"seed\": " + std::to_string((unsigned long long)p->seed) + ",\n";
  s += std::string("  \"embreebvh\": ") + b(p->embreebvh) + ",\n";
  s += std::string("  \"highqualitybvh\": ") + b(p->highqualitybvh) + ",\n";
  s += std::string("  \"noparallel\": ") + b(p->noparallel) + ",\n";
  s += "  \"pratio\": " + std::to_string(p->pratio) + ",\n";
  s += std::string("  \"denoise\": ") + b(p->denoise) + ",\n";
  // (the tolerance switch is this library's: written only when it is on, so that a default file is the reference's)
  s += "  \"batch\": " + std::to_string(p->batch) + (p->fastmath ? std::string(",\n  \"fastmath\": true") : std::string()) + "\n}\n";
  if (buffer && capacity > 0) {
    int64_t n = std::min<int64_t>((int64_t)s.size(), capacity - 1);
    std::memcpy(buffer, s.data(), (size_t)n);
    buffer[n] = 0;
  }
  return (int64_t)s.size();
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// PLY
// ---------------------------------------------------------------------------------------------
namespace {

enum PlyType { T_I8, T_I16, T_I32, T_I64, T_U8, T_U16, T_U32, T_U64, T_F32, T_F64, T_BAD };
int type_size(int t) {
  static const int sz[] = {1, 2, 4, 8, 1, 2, 4, 8, 4, 8, 0};
  return sz[t];
}
int type_of(const std::string& s) {  // yocto_modelio.cpp:489-500
  static const char* names[][2] = {{"char", "int8"}, {"short", "int16"}, {"int", "int32"}, {"long", "int64"}, {"uchar", "uint8"},
      {"ushort", "uint16"}, {"uint", "uint32"}, {"ulong", "uint64"}, {"float", "float32"}, {"double", "float64"}};
  for (int k = 0; k < 10; k++)
    if (s == names[k][0] || s == names[k][1]) return k;
  return T_BAD;
}

struct PlyProp {
  std::string name;
  int         type = T_BAD;
  bool        is_list = false;
  // scalars: one value per element; lists: sizes + the concatenated values
  std::vector<double>  ascii_values;  // ascii files only: parsed numbers (scalars, or list values)
  std::vector<uint8_t> sizes;         // lists: ldata_u8
  std::vector<size_t>  offsets;       // binary: byte offset of each element's value / of each list's first value
};
struct PlyElem {
  std::string          name;
  size_t               count = 0;
  std::vector<PlyProp> props;
  PlyProp*             find(const char* n) {
    for (auto& p : props)
      if (p.name == n) return &p;
    return nullptr;
  }
};

}  // namespace

struct ythip_ply {
  std::string          path;
  const uint8_t*       data = nullptr;
  size_t               size = 0;
  int                  format = 0;  // 0 ascii, 1 binary little endian, 2 binary big endian
  std::vector<PlyElem> elems;
  PlyElem*             find(const char* n) {
    for (auto& e : elems)
      if (e.name == n) return &e;
    return nullptr;
  }
};

namespace {

// a stored value as a double / as T the way the reference's get_value<T> casts it (yocto_modelio.h:519-533)
template <typename T, typename S>
T cast_from(const uint8_t* p, bool swap) {
  S    v;
  auto b = (uint8_t*)&v;
  if (!swap)
    std::memcpy(b, p, sizeof(S));
  else
    for (size_t k = 0; k < sizeof(S); k++) b[k] = p[sizeof(S) - 1 - k];
  return (T)v;
}
template <typename T>
T load_as(const ythip_ply& ply, const PlyProp& pr, size_t index) {
  if (ply.format == 0) {
    // ascii: the reference parses every token with the property's own type (parse_value into
    // int8 ... double) and casts to T on access; integers and floats below survive a double
    double d = pr.ascii_values[index];
    switch (pr.type) {
      case T_I8: return (T)(int8_t)d;
      case T_I16: return (T)(int16_t)d;
      case T_I32: return (T)(int32_t)d;
      case T_I64: return (T)(int64_t)d;
      case T_U8: return (T)(uint8_t)d;
      case T_U16: return (T)(uint16_t)d;
      case T_U32: return (T)(uint32_t)d;
      case T_U64: return (T)(uint64_t)d;
      case T_F32: return (T)(float)d;
      default: return (T)d;
    }
  }
  return T{};  // (binary files are read in place: value_at)
}
template <typename T>
T value_at(const ythip_ply& ply, const PlyProp& pr, const uint8_t* p) {
  const bool swap = ply.format == 2;
  switch (pr.type) {
    case T_I8: return cast_from<T, int8_t>(p, swap);
    case T_I16: return cast_from<T, int16_t>(p, swap);
    case T_I32: return cast_from<T, int32_t>(p, swap);
    case T_I64: return cast_from<T, int64_t>(p, swap);
    case T_U8: return cast_from<T, uint8_t>(p, swap);
    case T_U16: return cast_from<T, uint16_t>(p, swap);
    case T_U32: return cast_from<T, uint32_t>(p, swap);
    case T_U64: return cast_from<T, uint64_t>(p, swap);
    case T_F32: return cast_from<T, float>(p, swap);
    default: return cast_from<T, double>(p, swap);
  }
}
// scalar property value `index` as T
template <typename T>
T scalar(const ythip_ply& ply, const PlyProp& pr, size_t index) {
  if (ply.format == 0) return load_as<T>(ply, pr, index);
  return value_at<T>(ply, pr, ply.data + pr.offsets[index]);
}
// list property: value `item` of list `list` as T
template <typename T>
T list_value(const ythip_ply& ply, const PlyProp& pr, size_t list, size_t first, size_t item) {
  if (ply.format == 0) return load_as<T>(ply, pr, first + item);
  return value_at<T>(ply, pr, ply.data + pr.offsets[list] + item * (size_t)type_size(pr.type));
}

bool next_line(const char*& p, const char* end, std::string& line) {
  if (p >= end) return false;
  const char* s = p;
  while (p < end && *p != '\n') p++;
  line.assign(s, p);
  if (p < end) p++;
  if (!line.empty() && line.back() == '\r') line.pop_back();
  return true;
}
std::vector<std::string> tokens(const std::string& s) {
  std::vector<std::string> t;
  size_t                   i = 0;
  while (i < s.size()) {
    while (i < s.size() && (s[i] == ' ' || s[i] == '\t')) i++;
    size_t j = i;
    while (j < s.size() && s[j] != ' ' && s[j] != '\t') j++;
    if (j > i) t.emplace_back(s, i, j - i);
    i = j;
  }
  return t;
}

// header + the positions of every value (binary) or every parsed number (ascii)
int parse_ply(ythip_ply& ply) {
  const char* p   = (const char*)ply.data;
  const char* end = p + ply.size;
  std::string line;
  bool        first = true, done = false;
  while (!done && next_line(p, end, line)) {
    auto t = tokens(line);
    if (t.empty()) continue;
    if (first) {
      if (t[0] != "ply") return io_fail(YTHIP_ERR_INVALID, "cannot parse " + ply.path);
      first = false;
      continue;
    }
    if (t[0] == "format") {
      if (t.size() < 2) return io_fail(YTHIP_ERR_INVALID, "cannot parse " + ply.path);
      if (t[1] == "ascii") ply.format = 0;
      else if (t[1] == "binary_little_endian") ply.format = 1;
      else if (t[1] == "binary_big_endian") ply.format = 2;
      else return io_fail(YTHIP_ERR_INVALID, "cannot parse " + ply.path);
    } else if (t[0] == "comment" || t[0] == "obj_info") {
    } else if (t[0] == "element") {
      if (t.size() < 3) return io_fail(YTHIP_ERR_INVALID, "cannot parse " + ply.path);
      ply.elems.emplace_back();
      ply.elems.back().name  = t[1];
      ply.elems.back().count = (size_t)std::strtoull(t[2].c_str(), nullptr, 10);
    } else if (t[0] == "property") {
      if (ply.elems.empty() || t.size() < 3) return io_fail(YTHIP_ERR_INVALID, "cannot parse " + ply.path);
      PlyProp pr;
      if (t[1] == "list") {
        if (t.size() < 5) return io_fail(YTHIP_ERR_INVALID, "cannot parse " + ply.path);
        if (type_of(t[2]) != T_U8) return io_fail(YTHIP_ERR_INVALID, "cannot parse " + ply.path);  // (the reference supports uchar counts only)
        pr.is_list = true, pr.type = type_of(t[3]), pr.name = t[4];
      } else {
        pr.type = type_of(t[1]), pr.name = t[2];
      }
      if (pr.type == T_BAD) return io_fail(YTHIP_ERR_INVALID, "cannot parse " + ply.path);
      ply.elems.back().props.push_back(std::move(pr));
    } else if (t[0] == "end_header") {
      done = true;
    } else {
      return io_fail(YTHIP_ERR_INVALID, "cannot parse " + ply.path);
    }
  }
  if (!done) return io_fail(YTHIP_ERR_INVALID, "cannot parse " + ply.path);
  // an element occupies at least one byte of the file (a line, a record): a count the file cannot
  // hold is a damaged header, refused before anything is sized by it; counts travel as int32
  for (auto& e : ply.elems)
    if (e.count > ply.size || e.count > 0x7fffffffull || e.props.size() > 4096) return io_fail(YTHIP_ERR_INVALID, "cannot read " + ply.path);
  if (ply.format == 0) {
    for (auto& e : ply.elems) {
      for (auto& pr : e.props) pr.ascii_values.reserve(e.count);
      for (size_t k = 0; k < e.count; k++) {
        if (!next_line(p, end, line)) return io_fail(YTHIP_ERR_INVALID, "cannot read " + ply.path);
        const char* s = line.c_str();
        char*       q = nullptr;
        for (auto& pr : e.props) {
          if (pr.is_list) {
            double n = std::strtod(s, &q);
            if (q == s) return io_fail(YTHIP_ERR_INVALID, "cannot parse " + ply.path);
            s = q;
            pr.sizes.push_back((uint8_t)n);
            for (int i = 0; i < (int)(uint8_t)n; i++) {
              double v = pr.type == T_F32 ? (double)std::strtof(s, &q) : std::strtod(s, &q);
              if (q == s) return io_fail(YTHIP_ERR_INVALID, "cannot parse " + ply.path);
              s = q;
              pr.ascii_values.push_back(v);
            }
          } else {
            double v = pr.type == T_F32 ? (double)std::strtof(s, &q) : std::strtod(s, &q);
            if (q == s) return io_fail(YTHIP_ERR_INVALID, "cannot parse " + ply.path);
            s = q;
            pr.ascii_values.push_back(v);
          }
        }
      }
    }
  } else {
    const uint8_t* b = (const uint8_t*)p;
    const uint8_t* e_ = ply.data + ply.size;
    for (auto& e : ply.elems) {
      bool   fixed  = true;
      size_t stride = 0;
      for (auto& pr : e.props) fixed &= !pr.is_list, stride += (size_t)type_size(pr.type);
      if (fixed) {  // a table of fixed-size records: no per-value bookkeeping, offsets are arithmetic
        if ((size_t)(e_ - b) < stride * e.count) return io_fail(YTHIP_ERR_INVALID, "cannot read " + ply.path);
        size_t off = 0;
        for (auto& pr : e.props) {
          pr.offsets.resize(e.count);
          for (size_t k = 0; k < e.count; k++) pr.offsets[k] = (size_t)(b - ply.data) + k * stride + off;
          off += (size_t)type_size(pr.type);
        }
        b += stride * e.count;
        continue;
      }
      for (auto& pr : e.props) pr.offsets.reserve(e.count), pr.sizes.reserve(pr.is_list ? e.count : 0);
      for (size_t k = 0; k < e.count; k++)
        for (auto& pr : e.props) {
          if (pr.is_list) {
            if (b >= e_) return io_fail(YTHIP_ERR_INVALID, "cannot read " + ply.path);
            uint8_t n = *b++;
            pr.sizes.push_back(n);
            pr.offsets.push_back((size_t)(b - ply.data));
            b += (size_t)n * (size_t)type_size(pr.type);
          } else {
            pr.offsets.push_back((size_t)(b - ply.data));
            b += (size_t)type_size(pr.type);
          }
          if (b > e_) return io_fail(YTHIP_ERR_INVALID, "cannot read " + ply.path);
        }
    }
  }
  return YTHIP_OK;
}

// how many triangles / quads / lines a list property turns into (yocto_modelio.h:618-737)
bool list_has_quads(const PlyProp& pr) {
  for (auto n : pr.sizes)
    if (n == 4) return true;
  return false;
}
size_t fan_count(const PlyProp& pr, int arity) {  // arity 3: triangles, 4: quads
  size_t n = 0;
  for (auto s : pr.sizes) n += s <= (arity == 3 ? 3 : 4) ? 1 : (size_t)s - 2;
  return n;
}
size_t line_count(const PlyProp& pr) {
  size_t n = 0;
  for (auto s : pr.sizes) n += s <= 2 ? 1 : (size_t)s - 1;
  return n;
}
size_t list_total(const PlyProp& pr) {
  size_t n = 0;
  for (auto s : pr.sizes) n += s;
  return n;
}
// get_values(ply, "vertex", {a, b, ...}): every named property must exist and be a scalar
bool have_all(PlyElem* e, std::initializer_list<const char*> names) {
  if (!e) return false;
  for (auto n : names) {
    auto pr = e->find(n);
    if (!pr || pr->is_list) return false;
  }
  return true;
}

}  // namespace

extern "C" {

// Opens (maps) a PLY file and reports what load_shape (yocto_sceneio.cpp:1017-1033) would produce
// from it as the num_* fields of `counts` (the *_offset fields are left at -1 / for the caller).
static int ply_open_impl(const char* path, ythip_ply** out, ythip_shape* counts) {
  if (!path || !out || !counts) return io_fail(YTHIP_ERR_INVALID, "null argument");
  *out   = nullptr;
  int fd = ::open(path, O_RDONLY);
  if (fd < 0) return io_fail(YTHIP_ERR_INVALID, std::string("cannot open ") + path);
  struct stat st;
  if (::fstat(fd, &st) != 0 || st.st_size <= 0) {
    ::close(fd);
    return io_fail(YTHIP_ERR_INVALID, std::string("cannot read ") + path);
  }
  void* m = ::mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
  ::close(fd);
  if (m == MAP_FAILED) return io_fail(YTHIP_ERR_INVALID, std::string("cannot read ") + path);
  std::unique_ptr<ythip_ply, void (*)(ythip_ply*)> ply(new ythip_ply{}, ythip_ply_close);  // (unmaps, whatever happens below)
  ply->path = path, ply->data = (const uint8_t*)m, ply->size = (size_t)st.st_size;
  int rc = parse_ply(*ply);
  if (rc) return rc;
  ythip_shape c = {};
  c.points_offset = c.lines_offset = c.triangles_offset = c.quads_offset = -1;
  c.positions_offset = c.normals_offset = c.texcoords_offset = c.colors_offset = c.radius_offset = -1;
  auto v = ply->find("vertex");
  if (have_all(v, {"x", "y", "z"})) c.num_positions = (int)v->count;
  if (have_all(v, {"nx", "ny", "nz"})) c.num_normals = (int)v->count;
  if (v && v->find("u") ? have_all(v, {"u", "v"}) : have_all(v, {"s", "t"})) c.num_texcoords = (int)v->count;
  if (v && v->find("alpha") ? have_all(v, {"red", "green", "blue", "alpha"}) : have_all(v, {"red", "green", "blue"}))
    c.num_colors = (int)v->count;
  if (have_all(v, {"radius"})) c.num_radius = (int)v->count;
  if (auto f = ply->find("face"))
    if (auto pr = f->find("vertex_indices"); pr && pr->is_list) {
      size_t n = fan_count(*pr, list_has_quads(*pr) ? 4 : 3);
      if (n > 0x7fffffffull) return io_fail(YTHIP_ERR_INVALID, std::string("cannot read ") + path);
      if (list_has_quads(*pr)) c.num_quads = (int)n;
      else c.num_triangles = (int)n;
    }
  if (auto l = ply->find("line"))
    if (auto pr = l->find("vertex_indices"); pr && pr->is_list) {
      size_t n = line_count(*pr);
      if (n > 0x7fffffffull) return io_fail(YTHIP_ERR_INVALID, std::string("cannot read ") + path);
      c.num_lines = (int)n;
    }
  if (auto pt = ply->find("point"))
    if (auto pr = pt->find("vertex_indices"); pr && pr->is_list) {
      size_t n = list_total(*pr);
      if (n > 0x7fffffffull) return io_fail(YTHIP_ERR_INVALID, std::string("cannot read ") + path);
      c.num_points = (int)n;
    }
  if (!c.num_points && !c.num_lines && !c.num_triangles && !c.num_quads)
    return io_fail(YTHIP_ERR_INVALID, std::string("empty shape ") + path);  // load_shape's shape_error()
  *counts = c;
  *out    = ply.release();
  return YTHIP_OK;
}

// Converts the file's properties into the caller's arrays (pool + shape offset; null = skip), sized
// by the counts ythip_ply_open reported: positions / normals [n][3], texcoords [n][2] (v flipped to
// 1 - v when flip_texcoord), colors [n][4] (alpha 1 when the file has none), radius [n], points [n],
// lines [n][2], triangles [n][3], quads [n][4] — the values of load_shape's shape_data.
static int ply_read_impl(ythip_ply* ply, int flip_texcoord, float* positions, float* normals, float* texcoords, float* colors,
    float* radius, int32_t* points, int32_t* lines, int32_t* triangles, int32_t* quads) {
  if (!ply) return io_fail(YTHIP_ERR_INVALID, "null argument");
  auto v      = ply->find("vertex");
  auto column = [&](float* dst, int stride, int item, const char* name) {
    auto&  pr = *v->find(name);
    size_t n  = v->count;
    if (ply->format == 1 && pr.type == T_F32)  // the common case: a little-endian float column
      for (size_t k = 0; k < n; k++) std::memcpy(dst + k * stride + item, ply->data + pr.offsets[k], 4);
    else
      for (size_t k = 0; k < n; k++) dst[k * stride + item] = scalar<float>(*ply, pr, k);
  };
  if (positions && have_all(v, {"x", "y", "z"})) column(positions, 3, 0, "x"), column(positions, 3, 1, "y"), column(positions, 3, 2, "z");
  if (normals && have_all(v, {"nx", "ny", "nz"})) column(normals, 3, 0, "nx"), column(normals, 3, 1, "ny"), column(normals, 3, 2, "nz");
  if (texcoords) {
    bool uv = v && v->find("u");
    if (uv ? have_all(v, {"u", "v"}) : have_all(v, {"s", "t"})) {
      column(texcoords, 2, 0, uv ? "u" : "s"), column(texcoords, 2, 1, uv ? "v" : "t");
      if (flip_texcoord)
        for (size_t k = 0; k < v->count; k++) texcoords[2 * k + 1] = 1 - texcoords[2 * k + 1];
    }
  }
  if (colors) {
    bool alpha = v && v->find("alpha");
    if (alpha ? have_all(v, {"red", "green", "blue", "alpha"}) : have_all(v, {"red", "green", "blue"})) {
      column(colors, 4, 0, "red"), column(colors, 4, 1, "green"), column(colors, 4, 2, "blue");
      if (alpha) column(colors, 4, 3, "alpha");
      else
        for (size_t k = 0; k < v->count; k++) colors[4 * k + 3] = 1;
    }
  }
  if (radius && have_all(v, {"radius"})) column(radius, 1, 0, "radius");
  // faces: quads if any face has four corners, else triangles; polygons are fanned (yocto_modelio.h:618-707)
  if (auto f = ply->find("face"))
    if (auto prp = f->find("vertex_indices"); prp && prp->is_list) {
      auto&  pr    = *prp;
      bool   asq   = list_has_quads(pr);
      int*   out   = asq ? quads : triangles;
      size_t first = 0, o = 0;
      if (out)
        for (size_t l = 0; l < pr.sizes.size(); l++) {
          size_t n  = pr.sizes[l];
          auto   at = [&](size_t i) { return list_value<int32_t>(*ply, pr, l, first, i); };
          if (!asq) {
            if (n <= 3) {
              out[3 * o] = n > 0 ? at(0) : -1, out[3 * o + 1] = n > 1 ? at(1) : -1, out[3 * o + 2] = n > 2 ? at(2) : -1;
              o++;
            } else {
              for (size_t i = 2; i < n; i++, o++) out[3 * o] = at(0), out[3 * o + 1] = at(i - 1), out[3 * o + 2] = at(i);
            }
          } else {
            if (n <= 4) {
              out[4 * o] = n > 0 ? at(0) : -1, out[4 * o + 1] = n > 1 ? at(1) : -1;
              out[4 * o + 2] = n > 2 ? at(2) : -1, out[4 * o + 3] = n > 3 ? at(3) : (n == 3 ? at(2) : -1);
              o++;
            } else {
              for (size_t i = 2; i < n; i++, o++)
                out[4 * o] = at(0), out[4 * o + 1] = at(i - 1), out[4 * o + 2] = at(i), out[4 * o + 3] = at(i);
            }
          }
          first += n;
        }
    }
  if (lines)
    if (auto le = ply->find("line"))
      if (auto prp = le->find("vertex_indices"); prp && prp->is_list) {
        auto&  pr    = *prp;
        size_t first = 0, o = 0;
        for (size_t l = 0; l < pr.sizes.size(); l++) {
          size_t n  = pr.sizes[l];
          auto   at = [&](size_t i) { return list_value<int32_t>(*ply, pr, l, first, i); };
          if (n <= 2) {
            lines[2 * o] = n > 0 ? at(0) : -1, lines[2 * o + 1] = n > 1 ? at(1) : -1;
            o++;
          } else {
            for (size_t i = 1; i < n; i++, o++) lines[2 * o] = at(i - 1), lines[2 * o + 1] = at(i);
          }
          first += n;
        }
      }
  if (points)
    if (auto pe = ply->find("point"))
      if (auto prp = pe->find("vertex_indices"); prp && prp->is_list) {
        auto&  pr    = *prp;
        size_t first = 0, o = 0;
        for (size_t l = 0; l < pr.sizes.size(); l++) {
          for (size_t i = 0; i < pr.sizes[l]; i++) points[o++] = list_value<int32_t>(*ply, pr, l, first, i);
          first += pr.sizes[l];
        }
      }
  return YTHIP_OK;
}

// (nothing thrown inside — an allocation that fails on a damaged file, say — may cross the C boundary)
int ythip_ply_open(const char* path, ythip_ply** out, ythip_shape* counts) {
  try {
    return ply_open_impl(path, out, counts);
  } catch (const std::exception& e) {
    if (out) *out = nullptr;
    return io_fail(YTHIP_ERR_INVALID, std::string("cannot read ") + (path ? path : "") + " (" + e.what() + ")");
  }
}
int ythip_ply_read(ythip_ply* ply, int flip_texcoord, float* positions, float* normals, float* texcoords, float* colors,
    float* radius, int32_t* points, int32_t* lines, int32_t* triangles, int32_t* quads) {
  try {
    return ply_read_impl(ply, flip_texcoord, positions, normals, texcoords, colors, radius, points, lines, triangles, quads);
  } catch (const std::exception& e) {
    return io_fail(YTHIP_ERR_INVALID, std::string("cannot read ") + (ply ? ply->path : std::string()) + " (" + e.what() + ")");
  }
}

void ythip_ply_close(ythip_ply* ply) {
  if (!ply) return;
  if (ply->data) ::munmap((void*)ply->data, ply->size);
  delete ply;
}

}  // extern "C"
