//
// yocto_hiptrace.cpp — host side of the drop-in: turns the reference's value
// types into the flat POD views of include/ythip.h, keeps the device mirrors
// resident between calls, and converts C-ABI error codes into the reference's
// exceptions.  Plain C++17, no HIP headers: everything device-side is behind
// libythip.so's C ABI.
//
// Built against the reference's own headers (-I<yocto-gl>/libs), the way a
// maintainer would build it inside the yocto-gl tree.
//
#include "yocto_hiptrace.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/ythip.h"
#include "yt_stamp.h"

namespace yocto::hip {

namespace {

// ---------------------------------------------------------------------------
// flat views (what ythip_upload_* read); storage lives here while uploading
// ---------------------------------------------------------------------------
struct flat_scene {
  std::vector<ythip_camera>      cameras;
  std::vector<ythip_instance>    instances;
  std::vector<ythip_environment> environments;
  std::vector<ythip_shape>       shapes;
  std::vector<ythip_texture>     textures;
  std::vector<ythip_material>    materials;
  std::vector<int32_t>           points, lines, triangles, quads;
  std::vector<float>             positions, normals, texcoords, colors, radius, pixelsf;
  std::vector<uint8_t>           pixelsb;
  ythip_scene                    view = {};
};

ythip_frame flat(const frame3f& f) {
  return {{f.x.x, f.x.y, f.x.z}, {f.y.x, f.y.y, f.y.z}, {f.z.x, f.z.y, f.z.z}, {f.o.x, f.o.y, f.o.z}};
}
ythip_camera flat(const camera_data& c) {
  return {flat(c.frame), c.orthographic ? 1 : 0, c.lens, c.film, c.aspect, c.focus, c.aperture};
}

// appends `src` to a pool of scalars; returns the offset in elements of U, -1 if empty
template <typename T, typename U>
int64_t append(std::vector<T>& pool, const std::vector<U>& src) {
  static_assert(sizeof(U) % sizeof(T) == 0, "pool element must divide the source element");
  if (src.empty()) return -1;
  constexpr size_t n   = sizeof(U) / sizeof(T);
  auto             off = (int64_t)(pool.size() / n);
  auto             ptr = reinterpret_cast<const T*>(src.data());
  pool.insert(pool.end(), ptr, ptr + src.size() * n);
  return off;
}

void flatten(const scene_data& s, flat_scene& f) {
  f = {};
  for (auto& c : s.cameras) f.cameras.push_back(flat(c));
  for (auto& i : s.instances) f.instances.push_back({flat(i.frame), i.shape, i.material});
  for (auto& e : s.environments)
    f.environments.push_back({flat(e.frame), {e.emission.x, e.emission.y, e.emission.z}, e.emission_tex});
  for (auto& m : s.materials) {
    // material_data (yocto_scene.h:122-141) and ythip_material share their layout
    static_assert(sizeof(ythip_material) == sizeof(material_data), "material layout drifted");
    ythip_material fm;
    std::memcpy(&fm, &m, sizeof(fm));
    f.materials.push_back(fm);
  }
  for (auto& t : s.textures) {
    ythip_texture ft = {t.width, t.height, t.linear ? 1 : 0, t.nearest ? 1 : 0, t.clamp ? 1 : 0,
        t.pixelsf.empty() ? 0 : 1, 0};
    auto off  = t.pixelsf.empty() ? append(f.pixelsb, t.pixelsb) : append(f.pixelsf, t.pixelsf);
    ft.offset = off < 0 ? 0 : off;
    f.textures.push_back(ft);
  }
  for (auto& sh : s.shapes) {
    ythip_shape fs      = {};
    fs.points_offset    = append(f.points, sh.points);
    fs.lines_offset     = append(f.lines, sh.lines);
    fs.triangles_offset = append(f.triangles, sh.triangles);
    fs.quads_offset     = append(f.quads, sh.quads);
    fs.positions_offset = append(f.positions, sh.positions);
    fs.normals_offset   = append(f.normals, sh.normals);
    fs.texcoords_offset = append(f.texcoords, sh.texcoords);
    fs.colors_offset    = append(f.colors, sh.colors);
    fs.radius_offset    = append(f.radius, sh.radius);
    fs.num_points       = (int)sh.points.size();
    fs.num_lines        = (int)sh.lines.size();
    fs.num_triangles    = (int)sh.triangles.size();
    fs.num_quads        = (int)sh.quads.size();
    fs.num_positions    = (int)sh.positions.size();
    fs.num_normals      = (int)sh.normals.size();
    fs.num_texcoords    = (int)sh.texcoords.size();
    fs.num_colors       = (int)sh.colors.size();
    fs.num_radius       = (int)sh.radius.size();
    f.shapes.push_back(fs);
  }
  auto& v            = f.view;
  v                  = {};
  v.num_cameras      = (int)f.cameras.size();
  v.num_instances    = (int)f.instances.size();
  v.num_environments = (int)f.environments.size();
  v.num_shapes       = (int)f.shapes.size();
  v.num_textures     = (int)f.textures.size();
  v.num_materials    = (int)f.materials.size();
  v.cameras          = f.cameras.data();
  v.instances        = f.instances.data();
  v.environments     = f.environments.data();
  v.shapes           = f.shapes.data();
  v.textures         = f.textures.data();
  v.materials        = f.materials.data();
  v.num_points       = (int64_t)f.points.size();
  v.num_lines        = (int64_t)f.lines.size() / 2;
  v.num_triangles    = (int64_t)f.triangles.size() / 3;
  v.num_quads        = (int64_t)f.quads.size() / 4;
  v.points           = f.points.data();
  v.lines            = f.lines.data();
  v.triangles        = f.triangles.data();
  v.quads            = f.quads.data();
  v.num_positions    = (int64_t)f.positions.size() / 3;
  v.num_normals      = (int64_t)f.normals.size() / 3;
  v.num_texcoords    = (int64_t)f.texcoords.size() / 2;
  v.num_colors       = (int64_t)f.colors.size() / 4;
  v.num_radius       = (int64_t)f.radius.size();
  v.positions        = f.positions.data();
  v.normals          = f.normals.data();
  v.texcoords        = f.texcoords.data();
  v.colors           = f.colors.data();
  v.radius           = f.radius.data();
  v.num_pixelsf      = (int64_t)f.pixelsf.size() / 4;
  v.num_pixelsb      = (int64_t)f.pixelsb.size() / 4;
  v.pixelsf          = f.pixelsf.data();
  v.pixelsb          = f.pixelsb.data();
}

// scene_bvh (yocto_bvh.h:70-79) → ythip_bvh: trees 0..S-1 are the shapes', tree S
// the instances'.  bvh_node (yocto_shape.h:474-480) and ythip_bvh_node share
// their 32-byte layout.
struct flat_bvh {
  std::vector<int64_t>        node_offset, prim_offset;
  std::vector<ythip_bvh_node> nodes;
  std::vector<int32_t>        prims;
  ythip_bvh                   view = {};
};
void flatten(const scene_bvh& b, flat_bvh& f) {
  static_assert(sizeof(ythip_bvh_node) == sizeof(bvh_node), "bvh_node layout drifted");
  f = {};
  auto add = [&](const bvh_tree& t) {
    f.node_offset.push_back((int64_t)f.nodes.size());
    f.prim_offset.push_back((int64_t)f.prims.size());
    auto n = f.nodes.size();
    f.nodes.resize(n + t.nodes.size());
    if (!t.nodes.empty()) std::memcpy(f.nodes.data() + n, t.nodes.data(), t.nodes.size() * sizeof(bvh_node));
    f.prims.insert(f.prims.end(), t.primitives.begin(), t.primitives.end());
  };
  for (auto& s : b.shapes) add(s.bvh);
  add(b.bvh);
  f.node_offset.push_back((int64_t)f.nodes.size());
  f.prim_offset.push_back((int64_t)f.prims.size());
  f.view = {(int)b.shapes.size() + 1, f.node_offset.data(), f.prim_offset.data(), f.nodes.data(), f.prims.data()};
}

struct flat_lights {
  std::vector<ythip_light> lights;
  std::vector<float>       cdf;
  ythip_lights             view = {};
};
void flatten(const trace_lights& l, flat_lights& f) {
  f = {};
  for (auto& light : l.lights) {
    f.lights.push_back({light.instance, light.environment, (int64_t)f.cdf.size(), (int)light.elements_cdf.size(), 0});
    f.cdf.insert(f.cdf.end(), light.elements_cdf.begin(), light.elements_cdf.end());
  }
  f.view = {(int)f.lights.size(), f.lights.data(), (int64_t)f.cdf.size(), f.cdf.data()};
}

// the tolerance mode (ythip_params::fastmath) is not a field of the reference's trace_params: a process-wide switch
// (0 bit-exact, 1 the tolerance mode, 2 the own-tree mode: include/ythip.h, ythip_params::fastmath)
std::atomic<int>& fast_math() {
  static std::atomic<int> level{[] {
    auto e = std::getenv("YOCTO_HIP_FASTMATH");
    return e ? std::min(std::max(std::atoi(e), 0), 2) : 0;
  }()};
  return level;
}

ythip_params flat(const trace_params& p) {
  ythip_params q   = {};
  q.fastmath       = fast_math().load();
  q.camera         = p.camera;
  q.resolution     = p.resolution;
  q.sampler        = (int)p.sampler;
  q.falsecolor     = (int)p.falsecolor;
  q.samples        = p.samples;
  q.bounces        = p.bounces;
  q.clamp          = p.clamp;
  q.nocaustics     = p.nocaustics;
  q.envhidden      = p.envhidden;
  q.tentfilter     = p.tentfilter;
  q.seed           = p.seed;
  q.embreebvh      = 0;
  q.highqualitybvh = p.highqualitybvh;
  q.noparallel     = p.noparallel;
  q.pratio         = p.pratio;
  q.denoise        = 0;
  q.batch          = p.batch;
  return q;
}

// ---------------------------------------------------------------------------
// residency cache
// ---------------------------------------------------------------------------
// The reference reads scene / bvh / lights fresh on every trace_samples call
// (yocto_trace.cpp:1595-1619: they are arguments); the device mirrors must notice edits.  A
// stamp = the object's identity + a CONTENT hash:
//   * small pools (cameras, materials, environments, instances): hashed in full on every
//     call — a material edited in place by the GUI is seen, and only that pool is re-sent;
//   * large arrays (vertices, elements, texture pixels, bvh nodes, light cdfs): their size,
//     their storage address and — by default — a 64-bit hash of EVERY byte, computed in 1-MiB
//     pieces on a pool of host threads (memory-bound: ~1 ms per 100 MB on a server host).  An
//     in-place edit of one vertex or one texel is therefore seen like any other, as in the
//     reference.  (Until round 3 the large arrays were hashed through a 256-element strided
//     sample: an edit that missed the sample rendered the old data, silently.)
//     hip::set_residency_check(residency_sampled) / YOCTO_HIP_RESIDENCY=sampled brings the
//     sample back for loops that call trace_samples thousands of times a second on a large
//     scene and promise to announce such edits with hip::invalidate().
using namespace stamp;  // yt_stamp.h: hash_t, fnv, array_hasher, residency_mode

struct scene_stamp {
  const void* who = nullptr;
  hash_t      layout = 0;     // counts + large arrays (sampled): a change re-uploads the scene
  hash_t      cameras = 0, materials = 0, environments = 0;  // small pools: re-sent on their own
  bool        valid   = false;
};
scene_stamp stamp_of(const scene_data& s) {
  scene_stamp st;
  st.who   = &s;
  st.valid = true;
  hash_t h = 1469598103934665603ull;
  size_t counts[] = {s.cameras.size(), s.instances.size(), s.environments.size(), s.shapes.size(), s.textures.size(),
      s.materials.size()};
  h = fnv(counts, sizeof(counts), h);
  // instances: frames feed the instance tree and the traversal records, so they belong to the layout
  static_assert(sizeof(instance_data) == 56, "instance_data layout");
  h = fnv_all(s.instances, h);
  array_hasher big;
  for (auto& sh : s.shapes) {
    big.add(sh.points), big.add(sh.lines), big.add(sh.triangles), big.add(sh.quads), big.add(sh.positions);
    big.add(sh.normals), big.add(sh.texcoords), big.add(sh.colors), big.add(sh.radius);
  }
  for (auto& t : s.textures) {
    int f[] = {t.width, t.height, t.linear, t.nearest, t.clamp};
    h       = fnv(f, sizeof(f), h);
    big.add(t.pixelsf), big.add(t.pixelsb);
  }
  auto hb   = big.finish();
  st.layout = fnv(&hb, sizeof(hb), h);
  hash_t c  = 1469598103934665603ull;
  for (auto& cam : s.cameras) {
    auto f = flat(cam);
    c      = fnv(&f, sizeof(f), c);
  }
  st.cameras   = c;
  st.materials = fnv_all(s.materials, 1469598103934665603ull);
  hash_t e     = 1469598103934665603ull;
  for (auto& env : s.environments) {
    ythip_environment f = {flat(env.frame), {env.emission.x, env.emission.y, env.emission.z}, env.emission_tex};
    e                   = fnv(&f, sizeof(f), e);
  }
  st.environments = e;
  return st;
}
// (a trace_bvh is identified by its node storage, not by the object's address: it is
// returned by value from make_trace_bvh and moved into the caller's variable)
hash_t stamp_of(const trace_bvh& b) {
  array_hasher big;
  big.add(b.bvh.bvh.nodes), big.add(b.bvh.bvh.primitives);
  for (auto& s : b.bvh.shapes) big.add(s.bvh.nodes), big.add(s.bvh.primitives);
  hash_t h = big.finish();
  return h ? h : 1;
}
hash_t stamp_of(const trace_lights& l) {
  auto         a = (uintptr_t)&l;
  hash_t       h = fnv(&a, sizeof(a));
  array_hasher big;
  for (auto& x : l.lights) {
    int ids[] = {x.instance, x.environment};
    h         = fnv(ids, sizeof(ids), h);
    big.add(x.elements_cdf);
  }
  auto hb = big.finish();
  h       = fnv(&hb, sizeof(hb), h);
  return h ? h : 1;
}

// A device state that ran ahead of its host vectors (trace_samples_resident) and then had to
// make room for another trace_state: kept here, NOT written through the old object's
// address (which may be gone), until that state comes back or asks for its data.
struct shadow_state {
  const void*           who = nullptr;
  int                   width = 0, height = 0, samples = 0;
  std::vector<float>    image, albedo, normal;
  std::vector<int>      hits;
  std::vector<uint64_t> rngs;
};

struct residency {
  std::mutex   mutex;
  ythip_multi* multi = nullptr;
  int          ranks = 0;
  scene_stamp  scene;
  hash_t       bvh = 0, lights = 0;
  const void*  state       = nullptr;  // which trace_state the device slices mirror
  int          width = 0, height = 0;
  int          dev_samples = -1;       // state.samples the device arrays correspond to
  bool         host_stale  = false;    // device ahead of the host vectors (trace_samples_resident)
  std::vector<shadow_state> shadows;
  ythip_scene  staged      = {};       // rank 0's pinned staging pools (the flat scene, filled in place)
  bool         have_staged = false;
  ythip_ctx*   ctx(int r = 0) { return ythip_multi_ctx(multi, r); }
};
// trace_cancel's relay words, one per trace_context (one global word made a cancel of context A end a batch
// of context B too, and trace_start's reset could erase a cancel meant for a batch still in flight).  trace_start makes
// a fresh word for its context and its worker hands that word to the library as the batch's `stop` flag (libythip
// polls it while the batch runs and cancels the batch by its number); trace_cancel raises the word of ITS context and
// nothing else.  A word lives as long as somebody holds it (the worker, the table).
using cancel_word = std::atomic<int32_t>;
static_assert(sizeof(cancel_word) == sizeof(int32_t) && cancel_word::is_always_lock_free,
    "the relay word is read by libythip as a plain volatile int32_t");
struct cancel_table {
  std::mutex                                                             mutex;
  std::unordered_map<const trace_context*, std::shared_ptr<cancel_word>> words;
};
cancel_table& cancel_words() {
  static cancel_table t;
  return t;
}
residency& cache() {
  static residency r;
  return r;
}
// params.denoise hand-off: false = the reference's default build (a copy), true = libythip's filter
std::atomic<bool>& device_denoiser() {
  static std::atomic<bool> on{false};
  return on;
}

[[noreturn]] void raise(const std::string& msg, int code) {
  if (code == YTHIP_ERR_SAMPLER) throw std::runtime_error("sampler unknown");  // yocto_trace.cpp:1437
  if (code == YTHIP_ERR_INVALID) throw std::invalid_argument(msg);
  throw std::runtime_error("ythip: " + msg);
}
void check(ythip_ctx* ctx, int code) {
  if (code != YTHIP_OK) raise(ythip_last_error(ctx), code);
}
void mcheck(residency& r, int code) {
  if (code != YTHIP_OK) raise(ythip_multi_last_error(r.multi), code);
}
// a replicated call: the same on every rank's context
template <typename F>
void on_all(residency& r, F f) {
  for (int k = 0; k < r.ranks; k++) check(r.ctx(k), f(r.ctx(k)));
}

// YOCTO_HIP_DEVICES=0,1,...,7 (one process drives them all: pixels sharded by tile columns,
// one RCCL framebuffer gather) or YOCTO_HIP_DEVICE=k; default device 0.
void ensure_context(residency& r) {
  if (r.multi) return;
  std::vector<int> devices;
  if (auto env = std::getenv("YOCTO_HIP_DEVICES")) {
    for (auto p = env; *p;) {
      char* end = nullptr;
      auto  v   = std::strtol(p, &end, 10);
      if (end == p) break;
      devices.push_back((int)v);
      p = *end ? end + 1 : end;
    }
  } else if (auto env1 = std::getenv("YOCTO_HIP_DEVICE")) {
    devices.push_back(std::atoi(env1));
  }
  if (devices.empty()) devices.push_back(0);
  auto rc = ythip_create_multi(devices.data(), (int)devices.size(), &r.multi);
  if (rc != YTHIP_OK) raise(ythip_multi_last_error(nullptr), rc);
  r.ranks = ythip_multi_size(r.multi);
  // parity is pinned to one libm (csrc/yt_libm.h): say so once if this host's is another
  static std::once_flag libm_once;
  std::call_once(libm_once, [&] {
    if (ythip_host_libm_matches(r.ctx(0)) == 0)
      std::fprintf(stderr, "yocto::hip: %s\n", ythip_last_error(r.ctx(0)));
  });
}

// Scene ingest: scene_data's vector-of-vectors go straight into
// libythip's pinned staging pools — one pass over the source, no intermediate flat copy, the
// upload is a DMA from the pools and the library keeps them as its host copies (flatten()
// above + ythip_upload_scene made three copies of the geometry; this makes one).  The layout
// written is flatten()'s, byte for byte (ingest_selfcheck()).
void count_scene(const scene_data& s, ythip_scene& v) {
  v                  = {};
  v.num_cameras      = (int)s.cameras.size();
  v.num_instances    = (int)s.instances.size();
  v.num_environments = (int)s.environments.size();
  v.num_shapes       = (int)s.shapes.size();
  v.num_textures     = (int)s.textures.size();
  v.num_materials    = (int)s.materials.size();
  for (auto& sh : s.shapes) {
    v.num_points += (int64_t)sh.points.size(), v.num_lines += (int64_t)sh.lines.size();
    v.num_triangles += (int64_t)sh.triangles.size(), v.num_quads += (int64_t)sh.quads.size();
    v.num_positions += (int64_t)sh.positions.size(), v.num_normals += (int64_t)sh.normals.size();
    v.num_texcoords += (int64_t)sh.texcoords.size(), v.num_colors += (int64_t)sh.colors.size();
    v.num_radius += (int64_t)sh.radius.size();
  }
  for (auto& t : s.textures) {
    if (t.pixelsf.empty())
      v.num_pixelsb += (int64_t)t.pixelsb.size();
    else
      v.num_pixelsf += (int64_t)t.pixelsf.size();
  }
}
// fills pools sized by count_scene (the pointers of `v` are writable staging memory)
void fill_scene(const scene_data& s, const ythip_scene& v) {
  auto cameras = (ythip_camera*)v.cameras;
  for (size_t k = 0; k < s.cameras.size(); k++) cameras[k] = flat(s.cameras[k]);
  auto instances = (ythip_instance*)v.instances;
  for (size_t k = 0; k < s.instances.size(); k++)
    instances[k] = {flat(s.instances[k].frame), s.instances[k].shape, s.instances[k].material};
  auto environments = (ythip_environment*)v.environments;
  for (size_t k = 0; k < s.environments.size(); k++) {
    auto& e         = s.environments[k];
    environments[k] = {flat(e.frame), {e.emission.x, e.emission.y, e.emission.z}, e.emission_tex};
  }
  static_assert(sizeof(ythip_material) == sizeof(material_data), "material layout drifted");
  if (!s.materials.empty())
    std::memcpy((void*)v.materials, (const void*)s.materials.data(), s.materials.size() * sizeof(ythip_material));
  // a cursor per pool, in elements; put() copies one source vector and returns its offset (-1: empty)
  auto put = [](auto* pool, int64_t& cursor, const auto& src) -> int64_t {
    if (src.empty()) return -1;
    auto off = cursor;
    std::memcpy((void*)((char*)pool + (size_t)cursor * sizeof(src[0])), (const void*)src.data(), src.size() * sizeof(src[0]));
    cursor += (int64_t)src.size();
    return off;
  };
  int64_t cf = 0, cb = 0;
  auto    textures = (ythip_texture*)v.textures;
  for (size_t k = 0; k < s.textures.size(); k++) {
    auto&         t  = s.textures[k];
    ythip_texture ft = {t.width, t.height, t.linear ? 1 : 0, t.nearest ? 1 : 0, t.clamp ? 1 : 0,
        t.pixelsf.empty() ? 0 : 1, 0};
    auto off  = t.pixelsf.empty() ? put(v.pixelsb, cb, t.pixelsb) : put(v.pixelsf, cf, t.pixelsf);
    ft.offset = off < 0 ? 0 : off;
    textures[k] = ft;
  }
  int64_t c_points = 0, c_lines = 0, c_triangles = 0, c_quads = 0, c_positions = 0, c_normals = 0, c_texcoords = 0,
          c_colors = 0, c_radius = 0;
  auto shapes = (ythip_shape*)v.shapes;
  for (size_t k = 0; k < s.shapes.size(); k++) {
    auto&       sh      = s.shapes[k];
    ythip_shape fs      = {};
    fs.points_offset    = put(v.points, c_points, sh.points);
    fs.lines_offset     = put(v.lines, c_lines, sh.lines);
    fs.triangles_offset = put(v.triangles, c_triangles, sh.triangles);
    fs.quads_offset     = put(v.quads, c_quads, sh.quads);
    fs.positions_offset = put(v.positions, c_positions, sh.positions);
    fs.normals_offset   = put(v.normals, c_normals, sh.normals);
    fs.texcoords_offset = put(v.texcoords, c_texcoords, sh.texcoords);
    fs.colors_offset    = put(v.colors, c_colors, sh.colors);
    fs.radius_offset    = put(v.radius, c_radius, sh.radius);
    fs.num_points       = (int)sh.points.size();
    fs.num_lines        = (int)sh.lines.size();
    fs.num_triangles    = (int)sh.triangles.size();
    fs.num_quads        = (int)sh.quads.size();
    fs.num_positions    = (int)sh.positions.size();
    fs.num_normals      = (int)sh.normals.size();
    fs.num_texcoords    = (int)sh.texcoords.size();
    fs.num_colors       = (int)sh.colors.size();
    fs.num_radius       = (int)sh.radius.size();
    shapes[k]           = fs;
  }
}
void ingest(residency& r, const scene_data& scene) {
  ythip_scene counts;
  count_scene(scene, counts);
  // ythip_scene_staging drops the resident scene before anything can fail: from here on nothing
  // is resident as far as the stamps are concerned (a failed ingest is retried by the next call)
  r.scene       = {};
  r.bvh = r.lights = 0;
  r.have_staged = false;
  check(r.ctx(0), ythip_scene_staging(r.ctx(0), &counts, &r.staged));
  fill_scene(scene, r.staged);
  check(r.ctx(0), ythip_upload_scene_staged(r.ctx(0)));
  r.have_staged = true;
  // the other ranks read rank 0's pinned pools (replicated scene: SURVEY.md §8e)
  for (int k = 1; k < r.ranks; k++) check(r.ctx(k), ythip_upload_scene(r.ctx(k), &r.staged));
}

// cameras / materials / environments: re-sent whenever their content hash differs from what is
// resident (the GUI edits them in place between batches)
void sync_small_pools(residency& r, const scene_data& scene, const scene_stamp& ss) {
  if (ss.cameras != r.scene.cameras) {
    std::vector<ythip_camera> cams;
    for (auto& c : scene.cameras) cams.push_back(flat(c));
    on_all(r, [&](ythip_ctx* c) { return ythip_update_cameras(c, cams.data(), (int)cams.size()); });
  }
  if (ss.materials != r.scene.materials) {
    static_assert(sizeof(ythip_material) == sizeof(material_data), "material layout drifted");
    on_all(r, [&](ythip_ctx* c) {
      return ythip_update_materials(c, (const ythip_material*)scene.materials.data(), (int)scene.materials.size());
    });
  }
  if (ss.environments != r.scene.environments) {
    std::vector<ythip_environment> envs;
    for (auto& e : scene.environments)
      envs.push_back({flat(e.frame), {e.emission.x, e.emission.y, e.emission.z}, e.emission_tex});
    on_all(r, [&](ythip_ctx* c) { return ythip_update_environments(c, envs.data(), (int)envs.size()); });
  }
  r.scene.cameras = ss.cameras, r.scene.materials = ss.materials, r.scene.environments = ss.environments;
}

// the scene's device mirrors (uploads what the stamp says changed); with `need_view` the
// flat view r.staged is valid on return (make_trace_bvh reads the geometry through it)
void ensure_scene(residency& r, const scene_data& scene, bool need_view = false) {
  ensure_context(r);
  auto ss    = stamp_of(scene);
  bool whole = !r.scene.valid || ss.who != r.scene.who || ss.layout != r.scene.layout;
  if (whole || (need_view && !r.have_staged)) {
    ingest(r, scene);
    r.scene = ss;
    r.bvh = r.lights = 0;  // a scene upload drops the device trees and lights
    return;
  }
  sync_small_pools(r, scene, ss);
  r.scene = ss;
}

void ensure_resident(residency& r, const scene_data& scene, const trace_bvh& bvh, const trace_lights& lights) {
  const bool own = fast_math().load() == 2;
  ensure_scene(r, scene);
  auto bs = stamp_of(bvh);
  if (bs != r.bvh) {
    flat_bvh f;
    flatten(bvh.bvh, f);
    on_all(r, [&](ythip_ctx* c) { return ythip_upload_bvh(c, &f.view); });
    r.bvh = bs;
  }
  if (own)  // (the library drops its own tree whenever geometry or the reference tree changes: ask it)
    // ... and let it build from ITS resident host copies (null scene): those follow ythip_update_instance_frames /
    // ythip_update_shape_vertices, which the staging view of the last ingest does not (a moved instance was
    // culled by a TLAS built from its pre-edit frame)
    on_all(r, [&](ythip_ctx* c) {
      return ythip_own_bvh_info(c, nullptr, nullptr, nullptr) == YTHIP_OK ? YTHIP_OK : ythip_build_own_bvh(c, nullptr);
    });
  auto ls = stamp_of(lights);
  if (ls != r.lights) {
    flat_lights f;
    flatten(lights, f);
    on_all(r, [&](ythip_ctx* c) { return ythip_upload_lights(c, &f.view); });
    r.lights = ls;
  }
}

void pull_state(residency& r, trace_state& state) {
  int samples = 0;
  mcheck(r, ythip_multi_state_download(r.multi, (float*)state.image.data(), (float*)state.albedo.data(),
                (float*)state.normal.data(), state.hits.data(), (uint64_t*)state.rngs.data(), &samples));
  state.samples = samples;
  r.host_stale  = false;
}

shadow_state* find_shadow(residency& r, const trace_state& state) {
  for (auto& s : r.shadows)
    if (s.who == &state && s.width == state.width && s.height == state.height && s.samples == state.samples) return &s;
  return nullptr;
}
void drop_shadows(residency& r, const void* who) {
  for (size_t k = 0; k < r.shadows.size();)
    if (r.shadows[k].who == who)
      r.shadows.erase(r.shadows.begin() + (long)k);
    else
      k++;
}
// the device is about to hold another trace_state: if it is the only up-to-date copy of
// the one it mirrors now, keep that copy
void stash_device_state(residency& r) {
  if (!r.multi || !r.state || !r.host_stale) return;
  drop_shadows(r, r.state);
  shadow_state s;
  s.who = r.state, s.width = r.width, s.height = r.height;
  auto n = (size_t)r.width * (size_t)r.height;
  s.image.resize(n * 4), s.albedo.resize(n * 3), s.normal.resize(n * 3), s.hits.resize(n), s.rngs.resize(n * 2);
  mcheck(r, ythip_multi_state_download(r.multi, s.image.data(), s.albedo.data(), s.normal.data(), s.hits.data(),
                s.rngs.data(), &s.samples));
  if (r.shadows.size() >= 4) r.shadows.erase(r.shadows.begin());
  r.shadows.push_back(std::move(s));
  r.host_stale = false;
}

// true when the device slices are the up-to-date copy of `state`
bool device_has(residency& r, const trace_state& state) {
  return r.multi && r.state == &state && r.width == state.width && r.height == state.height &&
         r.dev_samples == state.samples;
}

// scene / bvh / lights / state resident and current on the device (caller holds the lock)
void ensure_state_only(residency& r, const trace_state& state) {
  static_assert(sizeof(rng_state) == 16 && sizeof(vec4f) == 16 && sizeof(vec3f) == 12, "trace_state layout");
  auto npix = (size_t)state.width * (size_t)state.height;
  if (state.image.size() != npix || state.albedo.size() != npix || state.normal.size() != npix ||
      state.hits.size() != npix || state.rngs.size() != npix)
    throw std::invalid_argument("yocto::hip::trace_samples: trace_state arrays do not match width x height");
  // the device slices mirror `state` iff it is the same object, same size, and nobody
  // advanced it on the host since (samples match)
  if (device_has(r, state)) return;
  stash_device_state(r);
  mcheck(r, ythip_multi_state_create(r.multi, state.width, state.height));
  if (auto sh = find_shadow(r, state)) {  // its newest copy is the one that was stashed
    mcheck(r, ythip_multi_state_upload(r.multi, sh->image.data(), sh->albedo.data(), sh->normal.data(),
                  sh->hits.data(), sh->rngs.data(), state.samples));
    drop_shadows(r, &state);
    r.host_stale = true;
  } else {
    drop_shadows(r, &state);  // (a stale shadow of another incarnation at this address)
    mcheck(r, ythip_multi_state_upload(r.multi, (const float*)state.image.data(), (const float*)state.albedo.data(),
                  (const float*)state.normal.data(), state.hits.data(), (const uint64_t*)state.rngs.data(),
                  state.samples));
    r.host_stale = false;
  }
  r.state = &state, r.width = state.width, r.height = state.height;
  r.dev_samples = state.samples;
}
void ensure_state(residency& r, trace_state& state, const scene_data& scene, const trace_bvh& bvh,
    const trace_lights& lights) {
  ensure_resident(r, scene, bvh, lights);
  ensure_state_only(r, state);
}

void trace_impl(trace_state& state, const scene_data& scene, const trace_bvh& bvh, const trace_lights& lights,
    const trace_params& params, bool download, const std::atomic<bool>* stop = nullptr, const cancel_word* relay = nullptr) {
  if (state.samples >= params.samples) return;  // yocto_trace.cpp:1598
  if (params.embreebvh) throw std::invalid_argument("yocto::hip::trace_samples: embreebvh has no device mirror");
  auto& r = cache();
  auto  lock = std::lock_guard{r.mutex};
  ensure_state(r, state, scene, bvh, lights);
  if (stop && stop->load()) return;  // cancelled during the uploads: nothing is launched
  auto p  = flat(params);
  auto rc = ythip_multi_trace_samples(r.multi, &p, relay ? reinterpret_cast<const volatile int32_t*>(relay) : nullptr);
  if (rc != YTHIP_OK && rc != YTHIP_ERR_CANCELLED) mcheck(r, rc);
  if (rc == YTHIP_ERR_CANCELLED || (stop && stop->load())) {
    // cancelled while the batch ran (trace_cancel raised the device flags): as in the
    // reference (yocto_trace.cpp:1636-1641) the pixels have taken different numbers of the
    // batch's samples and state.samples does not advance
    for (int k = 0; k < r.ranks; k++) check(r.ctx(k), ythip_state_set_samples(r.ctx(k), state.samples));
    r.host_stale = true;
    return;
  }
  state.samples += params.batch;  // yocto_trace.cpp:1614
  r.dev_samples = state.samples;
  r.host_stale  = true;
  if (download) pull_state(r, state);
  if (params.denoise && !state.denoised.empty()) {  // yocto_trace.cpp:1615-1618
    if (device_denoiser().load() && r.ranks == 1) {
      // the filter runs on the resident image + guides; only the result (16 B/pixel) comes back
      check(r.ctx(), ythip_denoise_state(r.ctx(), nullptr, (float*)state.denoised.data()));
      return;
    }
    if (r.host_stale) pull_state(r, state);
    if (device_denoiser().load())
      check(r.ctx(), ythip_denoise_image(r.ctx(), nullptr, state.width, state.height, (const float*)state.image.data(),
                         (const float*)state.albedo.data(), (const float*)state.normal.data(),
                         (float*)state.denoised.data()));
    else
      yocto::denoise_image(state.denoised, state.width, state.height, state.image, state.albedo, state.normal);
  }
}

}  // namespace

bool hip_supported() {
  try {
    auto& r    = cache();
    auto  lock = std::lock_guard{r.mutex};
    ensure_context(r);
    return true;
  } catch (...) {
    return false;
  }
}

int hip_device_count() {
  auto& r    = cache();
  auto  lock = std::lock_guard{r.mutex};
  ensure_context(r);
  return r.ranks;
}

// Test hook: ingests `scene` into staging pools and compares every pool, byte for byte, with
// the copy-based flatten() the first round shipped.  Returns "" when identical.
std::string ingest_selfcheck(const scene_data& scene, double* ms_staged, double* ms_copy) {
  auto& r    = cache();
  auto  lock = std::lock_guard{r.mutex};
  ensure_context(r);
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms  = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  flat_scene f;
  auto       t0 = now();
  flatten(scene, f);
  on_all(r, [&](ythip_ctx* c) { return ythip_upload_scene(c, &f.view); });
  auto t1 = now();
  ingest(r, scene);
  auto t2 = now();
  if (ms_copy) *ms_copy = ms(t0, t1);
  if (ms_staged) *ms_staged = ms(t1, t2);
  r.scene = stamp_of(scene);
  r.bvh = r.lights = 0;
  auto& a = f.view;
  auto& b = r.staged;
  if (a.num_cameras != b.num_cameras || a.num_instances != b.num_instances || a.num_environments != b.num_environments ||
      a.num_shapes != b.num_shapes || a.num_textures != b.num_textures || a.num_materials != b.num_materials ||
      a.num_points != b.num_points || a.num_lines != b.num_lines || a.num_triangles != b.num_triangles ||
      a.num_quads != b.num_quads || a.num_positions != b.num_positions || a.num_normals != b.num_normals ||
      a.num_texcoords != b.num_texcoords || a.num_colors != b.num_colors || a.num_radius != b.num_radius ||
      a.num_pixelsf != b.num_pixelsf || a.num_pixelsb != b.num_pixelsb)
    return "counts";
  auto same = [](const void* x, const void* y, size_t bytes) { return bytes == 0 || std::memcmp(x, y, bytes) == 0; };
#define YT_SAME(field, bytes) \
  if (!same(a.field, b.field, (size_t)(bytes))) return #field;
  YT_SAME(cameras, a.num_cameras * sizeof(ythip_camera));
  YT_SAME(instances, a.num_instances * sizeof(ythip_instance));
  YT_SAME(environments, a.num_environments * sizeof(ythip_environment));
  YT_SAME(shapes, a.num_shapes * sizeof(ythip_shape));
  YT_SAME(textures, a.num_textures * sizeof(ythip_texture));
  YT_SAME(materials, a.num_materials * sizeof(ythip_material));
  YT_SAME(points, a.num_points * 4);
  YT_SAME(lines, a.num_lines * 8);
  YT_SAME(triangles, a.num_triangles * 12);
  YT_SAME(quads, a.num_quads * 16);
  YT_SAME(positions, a.num_positions * 12);
  YT_SAME(normals, a.num_normals * 12);
  YT_SAME(texcoords, a.num_texcoords * 8);
  YT_SAME(colors, a.num_colors * 16);
  YT_SAME(radius, a.num_radius * 4);
  YT_SAME(pixelsf, a.num_pixelsf * 16);
  YT_SAME(pixelsb, a.num_pixelsb * 4);
#undef YT_SAME
  return "";
}

void invalidate() {
  auto& r    = cache();
  auto  lock = std::lock_guard{r.mutex};
  r.scene    = {};
  r.bvh = r.lights = 0;
}
void set_fast_math(bool on) { fast_math().store(on ? 1 : 0); }
bool get_fast_math() { return fast_math().load() != 0; }
void set_fast_math_level(int level) { fast_math().store(std::min(std::max(level, 0), 2)); }
int  get_fast_math_level() { return fast_math().load(); }
void set_residency_check(residency_check mode) {
  auto& r    = cache();
  auto  lock = std::lock_guard{r.mutex};
  residency_mode().store(mode == residency_sampled ? 1 : 0);
  r.scene = {};  // stamps taken one way do not compare with stamps taken the other way
  r.bvh = r.lights = 0;
}
residency_check get_residency_check() { return residency_mode().load() == 1 ? residency_sampled : residency_full; }

// make_trace_state — yocto_trace.cpp:1495-1520, through libythip's own builders (the image-size
// rule ythip_state_size, the serial master stream ythip_make_rngs): the same code the
// multi-GPU launcher and the tests use.
trace_state make_trace_state(const scene_data& scene, const trace_params& params) {
  if (params.camera < 0 || params.camera >= (int)scene.cameras.size()) return yocto::make_trace_state(scene, params);
  auto cam = flat(scene.cameras[params.camera]);
  auto state = trace_state{};
  if (auto rc = ythip_state_size(&cam, params.resolution, &state.width, &state.height); rc != YTHIP_OK)
    raise(ythip_last_error(nullptr), rc);
  state.samples = 0;
  auto n = (size_t)state.width * (size_t)state.height;
  state.image.assign(n, {0, 0, 0, 0});
  state.albedo.assign(n, {0, 0, 0});
  state.normal.assign(n, {0, 0, 0});
  state.hits.assign(n, 0);
  state.rngs.assign(n, {});
  if (auto rc = ythip_make_rngs(params.seed, (int64_t)n, (uint64_t*)state.rngs.data()); rc != YTHIP_OK)
    raise(ythip_last_error(nullptr), rc);
  if (params.denoise) state.denoised.assign(n, {0, 0, 0, 0});
  return state;
}
// make_trace_lights — yocto_trace.cpp:1528-1581, through libythip's host builder
// (ythip_host_lights_build: area CDFs and environment texel CDFs, tested equal to the
// reference's bytes).
trace_lights make_trace_lights(const scene_data& scene, const trace_params& params) {
  (void)params;
  flat_scene f;
  flatten(scene, f);
  ythip_hostlights* hl = nullptr;
  if (auto rc = ythip_host_lights_build(&f.view, &hl); rc != YTHIP_OK) raise(ythip_last_error(nullptr), rc);
  ythip_lights view = {};
  if (auto rc = ythip_host_lights_view(hl, &view); rc != YTHIP_OK) {
    ythip_host_lights_free(hl);
    raise(ythip_last_error(nullptr), rc);
  }
  auto lights = trace_lights{};
  for (int k = 0; k < view.num_lights; k++) {
    auto& l     = view.lights[k];
    auto& light = lights.lights.emplace_back();
    light.instance    = l.instance;
    light.environment = l.environment;
    light.elements_cdf.assign(view.cdf + l.cdf_offset, view.cdf + l.cdf_offset + l.cdf_count);
  }
  ythip_host_lights_free(hl);
  return lights;
}
// make_trace_bvh — yocto_trace.cpp:88-96.  Large shapes are built on the device
// (ythip_build_bvh, yt_gpubuild.hip: the reference's tree node for node), the
// result stays resident for trace_samples AND comes back as the reference's own
// value type, so either back-end can use it.  params.highqualitybvh selects the
// reference's binned-SAH split (split_sah), built on the device as well.
trace_bvh make_trace_bvh(const scene_data& scene, const trace_params& params) {
  if (params.embreebvh) throw std::invalid_argument("yocto::hip::make_trace_bvh: embreebvh has no device mirror");
  auto&      r    = cache();
  auto       lock = std::lock_guard{r.mutex};
  ensure_scene(r, scene, true);
  on_all(r, [&](ythip_ctx* c) { return ythip_build_bvh(c, &r.staged, params.highqualitybvh ? 1 : 0); });  // deterministic: the same tree on every rank
  int32_t ntrees = 0;
  int64_t nnodes = 0, nprims = 0;
  check(r.ctx(), ythip_bvh_sizes(r.ctx(), &ntrees, &nnodes, &nprims));
  auto node_offset = std::vector<int64_t>(ntrees + 1), prim_offset = std::vector<int64_t>(ntrees + 1);
  auto nodes       = std::vector<ythip_bvh_node>((size_t)nnodes);
  auto prims       = std::vector<int32_t>((size_t)nprims);
  check(r.ctx(), ythip_bvh_download(r.ctx(), node_offset.data(), prim_offset.data(), nodes.data(), prims.data()));
  static_assert(sizeof(ythip_bvh_node) == sizeof(bvh_node), "bvh_node layout drifted");
  auto out  = trace_bvh{};
  auto fill = [&](bvh_tree& t, int k) {
    t.nodes.resize((size_t)(node_offset[k + 1] - node_offset[k]));
    if (!t.nodes.empty()) std::memcpy((void*)t.nodes.data(), nodes.data() + node_offset[k], t.nodes.size() * sizeof(bvh_node));
    t.primitives.assign(prims.begin() + prim_offset[k], prims.begin() + prim_offset[k + 1]);
  };
  out.bvh.shapes.resize((size_t)ntrees - 1);
  for (auto k = 0; k < ntrees - 1; k++) fill(out.bvh.shapes[k].bvh, k);
  fill(out.bvh.bvh, ntrees - 1);
  r.bvh = stamp_of(out);  // already resident: trace_samples will not upload it again
  return out;
}

void update_trace_bvh(trace_bvh& bvh, const scene_data& scene, const vector<int>& updated_instances,
    const vector<int>& updated_shapes) {
  auto& r    = cache();
  auto  lock = std::lock_guard{r.mutex};
  for (auto s : updated_shapes)
    if (s < 0 || s >= (int)scene.shapes.size()) throw std::out_of_range("update_trace_bvh: shape index");
  for (auto i : updated_instances)
    if (i < 0 || i >= (int)scene.instances.size()) throw std::out_of_range("update_trace_bvh: instance index");
  ensure_context(r);
  // the scene as the device has it differs from `scene` exactly by the announced edits when
  // the same object with the same element lists is resident: then they go up on their own
  auto fresh_scene = !r.scene.valid || r.scene.who != &scene;
  if (fresh_scene) ensure_scene(r, scene);  // a scene that was not resident goes up whole, already edited
  if (auto bs = stamp_of(bvh); bs != r.bvh) {
    flat_bvh f;
    flatten(bvh.bvh, f);
    on_all(r, [&](ythip_ctx* c) { return ythip_upload_bvh(c, &f.view); });
    r.bvh = bs;
  }
  if (!fresh_scene) {
    // an edit of a camera / material / environment in the same frame (an animated camera plus a
    // moving object) goes up too: the stamp taken below would otherwise call it resident
    sync_small_pools(r, scene, stamp_of(scene));
    for (auto s : updated_shapes) {
      auto& sh = scene.shapes[s];
      on_all(r, [&](ythip_ctx* c) {
        return ythip_update_shape_vertices(c, s, sh.positions.empty() ? nullptr : &sh.positions[0].x,
            (int64_t)sh.positions.size(), sh.normals.empty() ? nullptr : &sh.normals[0].x, (int64_t)sh.normals.size(),
            sh.radius.empty() ? nullptr : sh.radius.data(), (int64_t)sh.radius.size());
      });
    }
    auto frames = std::vector<ythip_frame>{};
    for (auto i : updated_instances) frames.push_back(flat(scene.instances[i].frame));
    on_all(r, [&](ythip_ctx* c) {
      return ythip_update_instance_frames(c, updated_instances.data(), (int)updated_instances.size(), frames.data());
    });
  }
  on_all(r, [&](ythip_ctx* c) {
    return ythip_update_bvh(c, updated_instances.data(), (int)updated_instances.size(), updated_shapes.data(),
        (int)updated_shapes.size());
  });
  // the refitted boxes back into the caller's trace_bvh (listed shapes + the instance tree)
  int32_t ntrees = 0;
  int64_t nnodes = 0, nprims = 0;
  check(r.ctx(), ythip_bvh_sizes(r.ctx(), &ntrees, &nnodes, &nprims));
  if (ntrees != (int)bvh.bvh.shapes.size() + 1) throw std::invalid_argument("update_trace_bvh: bvh / scene mismatch");
  auto node_offset = std::vector<int64_t>(ntrees + 1), prim_offset = std::vector<int64_t>(ntrees + 1);
  auto nodes       = std::vector<ythip_bvh_node>((size_t)nnodes);
  auto prims       = std::vector<int32_t>((size_t)nprims);
  check(r.ctx(), ythip_bvh_download(r.ctx(), node_offset.data(), prim_offset.data(), nodes.data(), prims.data()));
  auto fill = [&](bvh_tree& t, int k) {
    auto n = (size_t)(node_offset[k + 1] - node_offset[k]);
    if (n != t.nodes.size()) throw std::invalid_argument("update_trace_bvh: tree sizes changed");
    if (n) std::memcpy((void*)t.nodes.data(), nodes.data() + node_offset[k], n * sizeof(bvh_node));
  };
  for (auto s : updated_shapes) fill(bvh.bvh.shapes[s].bvh, s);
  fill(bvh.bvh.bvh, ntrees - 1);
  r.bvh   = stamp_of(bvh);
  r.scene = stamp_of(scene);  // the edited vertices / frames are what is resident now
}

void trace_samples(trace_state& state, const scene_data& scene, const trace_bvh& bvh, const trace_lights& lights,
    const trace_params& params) {
  trace_impl(state, scene, bvh, lights, params, true);
}
void trace_samples_resident(trace_state& state, const scene_data& scene, const trace_bvh& bvh,
    const trace_lights& lights, const trace_params& params) {
  trace_impl(state, scene, bvh, lights, params, false);
}
void trace_sample(trace_state& state, const scene_data& scene, const trace_bvh& bvh, const trace_lights& lights,
    int i, int j, int sample, const trace_params& params) {
  if (params.embreebvh) throw std::invalid_argument("yocto::hip::trace_sample: embreebvh has no device mirror");
  auto& r    = cache();
  auto  lock = std::lock_guard{r.mutex};
  // a per-pixel caller may have edited the host arrays without touching state.samples:
  // unless the device copy is known to be ahead, the host copy goes up again
  if (!r.host_stale) r.state = nullptr;
  ensure_state(r, state, scene, bvh, lights);
  auto p    = flat(params);
  auto rank = i >= 0 ? (i / 16) % r.ranks : 0;  // the rank that owns the pixel's tile column
  check(r.ctx(rank), ythip_trace_sample(r.ctx(rank), &p, i, j, sample));
  auto samples = state.samples;
  pull_state(r, state);  // the pixel's five entries changed; state.samples did not
  state.samples = samples;
}
void download_state(trace_state& state) {
  auto& r    = cache();
  auto  lock = std::lock_guard{r.mutex};
  if (device_has(r, state) && r.host_stale) return pull_state(r, state);
  if (auto sh = find_shadow(r, state)) {  // its newest copy was stashed when another state took the device
    std::memcpy((void*)state.image.data(), sh->image.data(), sh->image.size() * sizeof(float));
    std::memcpy((void*)state.albedo.data(), sh->albedo.data(), sh->albedo.size() * sizeof(float));
    std::memcpy((void*)state.normal.data(), sh->normal.data(), sh->normal.size() * sizeof(float));
    std::memcpy((void*)state.hits.data(), sh->hits.data(), sh->hits.size() * sizeof(int));
    std::memcpy((void*)state.rngs.data(), sh->rngs.data(), sh->rngs.size() * sizeof(uint64_t));
    drop_shadows(r, &state);
  }
}

image_data trace_image(const scene_data& scene, const trace_params& params) {
  auto bvh    = hip::make_trace_bvh(scene, params);
  auto lights = hip::make_trace_lights(scene, params);
  auto state  = hip::make_trace_state(scene, params);
  for (auto sample = 0; sample < params.samples; sample++)  // yocto_trace.cpp:1588-1590
    trace_samples_resident(state, scene, bvh, lights, params);
  download_state(state);
  auto image = yocto::get_image(state);
  release();  // bvh / lights / state die with this frame: their stamps must not outlive them
  return image;
}

// ---------------------------------------------------------------------------
// the interactive loop
// ---------------------------------------------------------------------------
namespace {
void check_linear_image(const image_data& image, const trace_state& state) {  // check_image, yocto_trace.cpp:1679-1686
  if (image.width != state.width || image.height != state.height)
    throw std::invalid_argument{"image should have the same size"};
  if (!image.linear) throw std::invalid_argument{"expected linear image"};
}
// where the newest copy of `state` is: 0 the host vectors, 1 the device, 2 a stashed shadow
int newest_copy(residency& r, const trace_state& state) {
  if (device_has(r, state) && r.host_stale) return 1;
  if (find_shadow(r, state)) return 2;
  return 0;
}
// a guide buffer (vec3f per pixel) of the resident / stashed state as the {xyz, 1} image
// get_albedo_image / get_normal_image return (yocto_trace.cpp:1769-1791)
void guide_image(residency& r, const trace_state& state, image_data& image, bool albedo) {
  auto n = (size_t)state.width * (size_t)state.height;
  if (r.ranks == 1 && device_has(r, state) && r.host_stale) {  // expanded on the device
    check(r.ctx(), albedo ? ythip_get_albedo_image(r.ctx(), (float*)image.pixels.data())
                          : ythip_get_normal_image(r.ctx(), (float*)image.pixels.data()));
    return;
  }
  std::vector<float> g(n * 3);
  if (auto sh = find_shadow(r, state); sh && !(device_has(r, state) && r.host_stale)) {
    g = albedo ? sh->albedo : sh->normal;
  } else {
    mcheck(r, ythip_multi_state_download(r.multi, nullptr, albedo ? g.data() : nullptr, albedo ? nullptr : g.data(),
                  nullptr, nullptr, nullptr));
  }
  for (size_t k = 0; k < n; k++) image.pixels[k] = {g[3 * k], g[3 * k + 1], g[3 * k + 2], 1};
}
}  // namespace

void get_image(image_data& image, const trace_state& state) {
  auto& r     = cache();
  auto  lock  = std::lock_guard{r.mutex};
  auto  where = newest_copy(r, state);
  if (where == 0) return yocto::get_image(image, state);  // host copy is current
  check_linear_image(image, state);
  if (where == 1) {  // the framebuffer gather: 16 B/pixel, not the whole 60 B/pixel trace_state
    mcheck(r, ythip_multi_get_image(r.multi, (float*)image.pixels.data()));
  } else {
    auto sh = find_shadow(r, state);
    std::memcpy((void*)image.pixels.data(), sh->image.data(), sh->image.size() * sizeof(float));
  }
}
image_data get_image(const trace_state& state) {
  auto image = make_image(state.width, state.height, true);
  hip::get_image(image, state);
  return image;
}

// the render and the denoiser's guide buffers (yocto_trace.h:183-190)
void get_rendered_image(image_data& image, const trace_state& state) {
  {
    auto& r    = cache();
    auto  lock = std::lock_guard{r.mutex};
    if (newest_copy(r, state) == 0) return yocto::get_rendered_image(image, state);
  }
  hip::get_image(image, state);
}
void get_albedo_image(image_data& image, const trace_state& state) {
  auto& r    = cache();
  auto  lock = std::lock_guard{r.mutex};
  if (newest_copy(r, state) == 0) return yocto::get_albedo_image(image, state);
  check_linear_image(image, state);
  guide_image(r, state, image, true);
}
void get_normal_image(image_data& image, const trace_state& state) {
  auto& r    = cache();
  auto  lock = std::lock_guard{r.mutex};
  if (newest_copy(r, state) == 0) return yocto::get_normal_image(image, state);
  check_linear_image(image, state);
  guide_image(r, state, image, false);
}
void get_denoised_image(image_data& image, const trace_state& state) {
  hip::get_rendered_image(image, state);  // yocto_trace.cpp:1763-1765 (no OIDN in this build)
}
image_data get_rendered_image(const trace_state& state) {
  auto image = make_image(state.width, state.height, true);
  hip::get_rendered_image(image, state);
  return image;
}
image_data get_albedo_image(const trace_state& state) {
  auto image = make_image(state.width, state.height, true);
  hip::get_albedo_image(image, state);
  return image;
}
image_data get_normal_image(const trace_state& state) {
  auto image = make_image(state.width, state.height, true);
  hip::get_normal_image(image, state);
  return image;
}
image_data get_denoised_image(const trace_state& state) {
  auto image = make_image(state.width, state.height, true);
  hip::get_denoised_image(image, state);
  return image;
}

// denoise_image — yocto_trace.cpp:1794-1872, on the device (libythip's guide-driven à-trous
// filter in the slot the reference gives to OIDN).  Same argument checks as the reference.
void set_device_denoiser(bool on) { device_denoiser().store(on); }
void denoise_image(vector<vec4f>& denoised, int width, int height, const vector<vec4f>& render,
    const vector<vec3f>& albedo, const vector<vec3f>& normal) {
  auto n = (size_t)width * (size_t)height;
  if (width <= 0 || height <= 0 || denoised.size() != n || render.size() != n || albedo.size() != n || normal.size() != n)
    throw std::invalid_argument{"image should have the same size"};  // check_image, yocto_trace.cpp:1843-1846
  auto& r    = cache();
  auto  lock = std::lock_guard{r.mutex};
  ensure_context(r);
  check(r.ctx(), ythip_denoise_image(r.ctx(), nullptr, width, height, (const float*)render.data(),
                     (const float*)albedo.data(), (const float*)normal.data(), (float*)denoised.data()));
}
void denoise_image(image_data& denoised, const image_data& render, const image_data& albedo, const image_data& normal) {
  auto n = (size_t)render.width * (size_t)render.height;
  if (denoised.width != render.width || denoised.height != render.height || albedo.width != render.width ||
      albedo.height != render.height || normal.width != render.width || normal.height != render.height ||
      denoised.pixels.size() != n || albedo.pixels.size() != n || normal.pixels.size() != n)
    throw std::invalid_argument{"image should have the same size"};  // yocto_trace.cpp:1802-1804
  auto a3 = vector<vec3f>(n), n3 = vector<vec3f>(n);
  for (size_t k = 0; k < n; k++) a3[k] = xyz(albedo.pixels[k]), n3[k] = xyz(normal.pixels[k]);
  hip::denoise_image(denoised.pixels, render.width, render.height, render.pixels, a3, n3);
}
image_data denoise_image(const image_data& render, const image_data& albedo, const image_data& normal) {
  auto denoised = make_image(render.width, render.height, render.linear);
  hip::denoise_image(denoised, render, albedo, normal);
  return denoised;
}

namespace {
void tonemap_device(residency& r, const trace_state& state, float exposure, bool filmic, float* ldr, uint8_t* ldrb) {
  ensure_context(r);
  if (r.ranks > 1) {
    // several devices: gather the frame (RCCL), tonemap with the reference's own function
    auto hdr = make_image(state.width, state.height, true);
    if (newest_copy(r, state) == 1)
      mcheck(r, ythip_multi_get_image(r.multi, (float*)hdr.pixels.data()));
    else if (auto sh = find_shadow(r, state))
      std::memcpy((void*)hdr.pixels.data(), sh->image.data(), sh->image.size() * sizeof(float));
    else
      yocto::get_image(hdr, state);
    auto out = yocto::tonemap_image(hdr, exposure, filmic);
    auto n   = (size_t)state.width * (size_t)state.height;
    if (ldr) std::memcpy(ldr, out.pixels.data(), n * sizeof(vec4f));
    if (ldrb)
      for (size_t k = 0; k < n; k++) {
        auto b = float_to_byte(out.pixels[k]);
        std::memcpy(ldrb + 4 * k, &b, 4);
      }
    return;
  }
  if (!device_has(r, state)) ensure_state_only(r, state);  // bring the newest copy over (host vectors or its shadow)
  check(r.ctx(), ythip_tonemap_image(r.ctx(), exposure, filmic ? 1 : 0, 1, ldr, ldrb));
}
}  // namespace

image_data tonemap_image(const trace_state& state, float exposure, bool filmic) {
  auto& r      = cache();
  auto  lock   = std::lock_guard{r.mutex};
  auto  result = make_image(state.width, state.height, false);  // yocto_image.cpp:185-192
  tonemap_device(r, state, exposure, filmic, (float*)result.pixels.data(), nullptr);
  return result;
}
vector<vec4b> tonemap_image_bytes(const trace_state& state, float exposure, bool filmic) {
  auto& r      = cache();
  auto  lock   = std::lock_guard{r.mutex};
  auto  result = vector<vec4b>((size_t)state.width * state.height);
  tonemap_device(r, state, exposure, filmic, nullptr, (uint8_t*)result.data());
  return result;
}

// trace_start — yocto_trace.cpp:1627-1649
void trace_start(trace_context& context, trace_state& state, const scene_data& scene, const trace_bvh& bvh,
    const trace_lights& lights, const trace_params& params) {
  if (state.samples >= params.samples) return;
  context.stop   = false;
  context.done   = false;
  auto word      = std::make_shared<cancel_word>(0);
  {
    auto& t    = cancel_words();
    auto  lock = std::lock_guard{t.mutex};
    t.words[&context] = word;  // (replaces the word of an earlier batch of this context: that batch has been joined)
  }
  context.worker = std::async(std::launch::async, [&, word]() {
    // the table entry leaves with the batch, however the batch ends (a context destroyed after a batch that ran to its
    // end must not leave its address behind: the next context allocated there would inherit a dead word) — only if it
    // still is THIS batch's word
    struct forget {
      const trace_context*                ctx;
      const std::shared_ptr<cancel_word>& word;
      ~forget() {
        auto& t    = cancel_words();
        auto  lock = std::lock_guard{t.mutex};
        if (auto it = t.words.find(ctx); it != t.words.end() && it->second == word) t.words.erase(it);
      }
    } forget_word{&context, word};
    if (context.stop) return;
    trace_impl(state, scene, bvh, lights, params, false, &context.stop, word.get());  // includes the denoise hand-off
    if (context.stop) return;
    context.done = true;
  });
}
// trace_cancel — yocto_trace.cpp:1652-1655.  The reference's workers test context.stop
// before every sample; here the flag is relayed to the devices (ythip_cancel), whose
// kernels test it at every sample boundary, so the batch in flight winds down within
// about one sample's time instead of running to its end.
void trace_cancel(trace_context& context) {
  context.stop = true;
  std::shared_ptr<cancel_word> word;
  {
    auto& t    = cancel_words();
    auto  lock = std::lock_guard{t.mutex};
    if (auto it = t.words.find(&context); it != t.words.end()) word = it->second, t.words.erase(it);
  }
  if (word) word->store(1);  // the library relays it to this context's batch in flight, and to no other
  if (context.worker.valid()) context.worker.get();
}
size_t pending_cancel_words() {
  auto& t    = cancel_words();
  auto  lock = std::lock_guard{t.mutex};
  return t.words.size();
}
// trace_preview — yocto_trace.cpp:1660-1676
void trace_preview(color_image& image, trace_context& context, trace_state& state, const scene_data& scene,
    const trace_bvh& bvh, const trace_lights& lights, const trace_params& params) {
  auto pparams = params;
  pparams.resolution /= params.pratio;
  pparams.samples = 1;
  auto pstate     = hip::make_trace_state(scene, pparams);
  hip::trace_samples(pstate, scene, bvh, lights, pparams);
  {  // the preview state dies here: forget its device mirror
    auto& r    = cache();
    auto  lock = std::lock_guard{r.mutex};
    if (r.state == &pstate) r.state = nullptr, r.dev_samples = -1, r.host_stale = false;
    drop_shadows(r, &pstate);
  }
  auto preview = yocto::get_image(pstate);
  for (auto idx = 0; idx < state.width * state.height; idx++) {
    auto i = idx % image.width, j = idx / image.width;
    auto pi = clamp(i / params.pratio, 0, preview.width - 1), pj = clamp(j / params.pratio, 0, preview.height - 1);
    image.pixels[idx] = preview.pixels[pj * preview.width + pi];
  }
}

void release() {
  auto& r    = cache();
  auto  lock = std::lock_guard{r.mutex};
  if (r.multi) ythip_destroy_multi(r.multi);
  r.multi = nullptr;
  r.ranks = 0;
  r.scene = {};
  r.bvh = r.lights = 0;
  r.state          = nullptr;
  r.dev_samples    = -1;
  r.host_stale     = false;
  r.shadows.clear();
  r.staged      = {};
  r.have_staged = false;
}

}  // namespace yocto::hip
