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

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <future>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ythip.h"

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

ythip_params flat(const trace_params& p) {
  ythip_params q   = {};
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
// A stamp identifies "the same content" cheaply: the object's address plus the
// sizes of everything that would force a re-flatten.  Cameras are excluded on
// purpose: they are re-sent on every call (72 B each).
struct stamp {
  const void*         who = nullptr;
  std::vector<size_t> sizes;
  bool operator==(const stamp& o) const { return who == o.who && sizes == o.sizes; }
  bool operator!=(const stamp& o) const { return !(*this == o); }
};
stamp stamp_of(const scene_data& s) {
  stamp st{&s, {s.cameras.size(), s.instances.size(), s.environments.size(), s.shapes.size(), s.textures.size(),
                   s.materials.size()}};
  for (auto& sh : s.shapes)
    st.sizes.insert(st.sizes.end(), {sh.points.size(), sh.lines.size(), sh.triangles.size(), sh.quads.size(),
                                        sh.positions.size(), (size_t)(uintptr_t)sh.positions.data()});
  return st;
}
// (identified by its node storage, not by the object's address: a trace_bvh is
// returned by value from make_trace_bvh and moved into the caller's variable)
stamp stamp_of(const trace_bvh& b) {
  stamp st{b.bvh.bvh.nodes.data(), {b.bvh.bvh.nodes.size(), b.bvh.shapes.size()}};
  for (auto& s : b.bvh.shapes) st.sizes.insert(st.sizes.end(), {s.bvh.nodes.size(), (size_t)(uintptr_t)s.bvh.nodes.data()});
  return st;
}
stamp stamp_of(const trace_lights& l) {
  stamp st{&l, {l.lights.size()}};
  for (auto& x : l.lights) st.sizes.push_back(x.elements_cdf.size());
  return st;
}

struct residency {
  std::mutex  mutex;
  ythip_ctx*  ctx = nullptr;
  stamp       scene, bvh, lights;
  const void* state       = nullptr;  // which trace_state the device slice mirrors
  int         width = 0, height = 0;
  int         dev_samples = -1;       // state.samples the device arrays correspond to
  bool        host_stale  = false;    // device ahead of the host vectors (trace_samples_resident)
};
residency& cache() {
  static residency r;
  return r;
}

[[noreturn]] void raise(ythip_ctx* ctx, int code) {
  std::string msg = ythip_last_error(ctx);
  if (code == YTHIP_ERR_SAMPLER) throw std::runtime_error("sampler unknown");  // yocto_trace.cpp:1437
  if (code == YTHIP_ERR_INVALID) throw std::invalid_argument(msg);
  throw std::runtime_error("ythip: " + msg);
}
void check(ythip_ctx* ctx, int code) {
  if (code != YTHIP_OK) raise(ctx, code);
}

void ensure_context(residency& r) {
  if (r.ctx) return;
  auto device = 0;
  if (auto env = std::getenv("YOCTO_HIP_DEVICE")) device = std::atoi(env);
  check(nullptr, ythip_create(device, &r.ctx));
}

// the scene's device mirror (uploads when the stamp changed)
void ensure_scene(residency& r, const scene_data& scene, flat_scene* keep = nullptr) {
  ensure_context(r);
  auto ss = stamp_of(scene);
  if (ss != r.scene || keep) {
    flat_scene  local;
    flat_scene& f = keep ? *keep : local;
    flatten(scene, f);
    if (ss != r.scene) {
      check(r.ctx, ythip_upload_scene(r.ctx, &f.view));
      r.scene = ss;
      r.bvh = r.lights = {};
    }
  }
}

void ensure_resident(residency& r, const scene_data& scene, const trace_bvh& bvh, const trace_lights& lights) {
  ensure_context(r);
  auto ss = stamp_of(scene);
  if (ss != r.scene) {
    ensure_scene(r, scene);
  } else {
    std::vector<ythip_camera> cams;
    for (auto& c : scene.cameras) cams.push_back(flat(c));
    check(r.ctx, ythip_update_cameras(r.ctx, cams.data(), (int)cams.size()));
  }
  auto bs = stamp_of(bvh);
  if (bs != r.bvh) {
    flat_bvh f;
    flatten(bvh.bvh, f);
    check(r.ctx, ythip_upload_bvh(r.ctx, &f.view));
    r.bvh = bs;
  }
  auto ls = stamp_of(lights);
  if (ls != r.lights) {
    flat_lights f;
    flatten(lights, f);
    check(r.ctx, ythip_upload_lights(r.ctx, &f.view));
    r.lights = ls;
  }
}

void pull_state(residency& r, trace_state& state) {
  int samples = 0;
  check(r.ctx, ythip_state_download(r.ctx, (float*)state.image.data(), (float*)state.albedo.data(),
                   (float*)state.normal.data(), state.hits.data(), (uint64_t*)state.rngs.data(), &samples));
  state.samples = samples;
  r.host_stale  = false;
}

// scene / bvh / lights / state resident and current on the device (caller holds the lock)
void ensure_state(residency& r, trace_state& state, const scene_data& scene, const trace_bvh& bvh,
    const trace_lights& lights) {
  static_assert(sizeof(rng_state) == 16 && sizeof(vec4f) == 16 && sizeof(vec3f) == 12, "trace_state layout");
  ensure_resident(r, scene, bvh, lights);
  auto npix = (size_t)state.width * (size_t)state.height;
  if (state.image.size() != npix || state.albedo.size() != npix || state.normal.size() != npix ||
      state.hits.size() != npix || state.rngs.size() != npix)
    throw std::invalid_argument("yocto::hip::trace_samples: trace_state arrays do not match width x height");
  // the device slice mirrors `state` iff it is the same object, same size, and
  // nobody advanced it on the host since (samples match)
  bool current = r.state == &state && r.width == state.width && r.height == state.height &&
                 r.dev_samples == state.samples;
  if (!current) {
    check(r.ctx, ythip_state_create(r.ctx, state.width, state.height, 0, state.height));
    check(r.ctx, ythip_state_upload(r.ctx, (const float*)state.image.data(), (const float*)state.albedo.data(),
                     (const float*)state.normal.data(), state.hits.data(), (const uint64_t*)state.rngs.data(),
                     state.samples));
    r.state = &state, r.width = state.width, r.height = state.height;
    r.dev_samples = state.samples;
  }
}

void trace_impl(trace_state& state, const scene_data& scene, const trace_bvh& bvh, const trace_lights& lights,
    const trace_params& params, bool download) {
  if (state.samples >= params.samples) return;  // yocto_trace.cpp:1598
  if (params.embreebvh) throw std::invalid_argument("yocto::hip::trace_samples: embreebvh has no device mirror");
  auto& r = cache();
  auto  lock = std::lock_guard{r.mutex};
  ensure_state(r, state, scene, bvh, lights);
  auto p = flat(params);
  check(r.ctx, ythip_trace_samples(r.ctx, &p, nullptr));
  state.samples += params.batch;  // yocto_trace.cpp:1614
  r.dev_samples = state.samples;
  r.host_stale  = true;
  if (download) pull_state(r, state);
  if (params.denoise && !state.denoised.empty()) {  // yocto_trace.cpp:1615-1618
    if (r.host_stale) pull_state(r, state);
    denoise_image(state.denoised, state.width, state.height, state.image, state.albedo, state.normal);
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

trace_state make_trace_state(const scene_data& scene, const trace_params& params) {
  return yocto::make_trace_state(scene, params);
}
trace_lights make_trace_lights(const scene_data& scene, const trace_params& params) {
  return yocto::make_trace_lights(scene, params);
}
// make_trace_bvh — yocto_trace.cpp:88-96.  Large shapes are built on the device
// (ythip_build_bvh, yt_gpubuild.hip: the reference's tree node for node), the
// result stays resident for trace_samples AND comes back as the reference's own
// value type, so either back-end can use it.  SAH builds stay on the host.
trace_bvh make_trace_bvh(const scene_data& scene, const trace_params& params) {
  if (params.embreebvh) throw std::invalid_argument("yocto::hip::make_trace_bvh: embreebvh has no device mirror");
  if (params.highqualitybvh) return yocto::make_trace_bvh(scene, params);
  auto&      r    = cache();
  auto       lock = std::lock_guard{r.mutex};
  flat_scene f;
  ensure_scene(r, scene, &f);
  check(r.ctx, ythip_build_bvh(r.ctx, &f.view, 0));
  int32_t ntrees = 0;
  int64_t nnodes = 0, nprims = 0;
  check(r.ctx, ythip_bvh_sizes(r.ctx, &ntrees, &nnodes, &nprims));
  auto node_offset = std::vector<int64_t>(ntrees + 1), prim_offset = std::vector<int64_t>(ntrees + 1);
  auto nodes       = std::vector<ythip_bvh_node>((size_t)nnodes);
  auto prims       = std::vector<int32_t>((size_t)nprims);
  check(r.ctx, ythip_bvh_download(r.ctx, node_offset.data(), prim_offset.data(), nodes.data(), prims.data()));
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
  auto fresh_scene = stamp_of(scene) != r.scene;
  ensure_scene(r, scene);  // a scene that was not resident goes up whole, already edited
  if (auto bs = stamp_of(bvh); bs != r.bvh) {
    flat_bvh f;
    flatten(bvh.bvh, f);
    check(r.ctx, ythip_upload_bvh(r.ctx, &f.view));
    r.bvh = bs;
  }
  if (!fresh_scene) {
    for (auto s : updated_shapes) {
      auto& sh = scene.shapes[s];
      check(r.ctx, ythip_update_shape_vertices(r.ctx, s, sh.positions.empty() ? nullptr : &sh.positions[0].x,
                       (int64_t)sh.positions.size(), sh.normals.empty() ? nullptr : &sh.normals[0].x,
                       (int64_t)sh.normals.size(), sh.radius.empty() ? nullptr : sh.radius.data(),
                       (int64_t)sh.radius.size()));
    }
    auto frames = std::vector<ythip_frame>{};
    for (auto i : updated_instances) frames.push_back(flat(scene.instances[i].frame));
    check(r.ctx, ythip_update_instance_frames(r.ctx, updated_instances.data(), (int)updated_instances.size(),
                     frames.data()));
  }
  check(r.ctx, ythip_update_bvh(r.ctx, updated_instances.data(), (int)updated_instances.size(),
                   updated_shapes.data(), (int)updated_shapes.size()));
  // the refitted boxes back into the caller's trace_bvh (listed shapes + the instance tree)
  int32_t ntrees = 0;
  int64_t nnodes = 0, nprims = 0;
  check(r.ctx, ythip_bvh_sizes(r.ctx, &ntrees, &nnodes, &nprims));
  if (ntrees != (int)bvh.bvh.shapes.size() + 1) throw std::invalid_argument("update_trace_bvh: bvh / scene mismatch");
  auto node_offset = std::vector<int64_t>(ntrees + 1), prim_offset = std::vector<int64_t>(ntrees + 1);
  auto nodes       = std::vector<ythip_bvh_node>((size_t)nnodes);
  auto prims       = std::vector<int32_t>((size_t)nprims);
  check(r.ctx, ythip_bvh_download(r.ctx, node_offset.data(), prim_offset.data(), nodes.data(), prims.data()));
  auto fill = [&](bvh_tree& t, int k) {
    auto n = (size_t)(node_offset[k + 1] - node_offset[k]);
    if (n != t.nodes.size()) throw std::invalid_argument("update_trace_bvh: tree sizes changed");
    if (n) std::memcpy((void*)t.nodes.data(), nodes.data() + node_offset[k], n * sizeof(bvh_node));
  };
  for (auto s : updated_shapes) fill(bvh.bvh.shapes[s].bvh, s);
  fill(bvh.bvh.bvh, ntrees - 1);
  r.bvh = stamp_of(bvh);
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
  auto p = flat(params);
  check(r.ctx, ythip_trace_sample(r.ctx, &p, i, j, sample));
  auto samples = state.samples;
  pull_state(r, state);  // the pixel's five entries changed; state.samples did not
  state.samples = samples;
}
void download_state(trace_state& state) {
  auto& r    = cache();
  auto  lock = std::lock_guard{r.mutex};
  if (r.ctx && r.state == &state && r.host_stale) pull_state(r, state);
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
// true when the device slice is the up-to-date copy of `state`
bool device_has(residency& r, const trace_state& state) {
  return r.ctx && r.state == &state && r.width == state.width && r.height == state.height &&
         r.dev_samples == state.samples;
}
}  // namespace

void get_image(image_data& image, const trace_state& state) {
  auto& r    = cache();
  auto  lock = std::lock_guard{r.mutex};
  if (!device_has(r, state) || !r.host_stale) return yocto::get_image(image, state);  // host copy is current
  if (image.width != state.width || image.height != state.height)
    throw std::invalid_argument{"image should have the same size"};  // check_image, yocto_trace.cpp:1679-1686
  if (!image.linear) throw std::invalid_argument{"expected linear image"};
  check(r.ctx, ythip_get_image(r.ctx, (float*)image.pixels.data()));
}
image_data get_image(const trace_state& state) {
  auto image = make_image(state.width, state.height, true);
  hip::get_image(image, state);
  return image;
}

// the render and the denoiser's guide buffers (yocto_trace.h:183-190)
namespace {
void check_linear_image(const image_data& image, const trace_state& state) {  // check_image, yocto_trace.cpp:1679-1686
  if (image.width != state.width || image.height != state.height)
    throw std::invalid_argument{"image should have the same size"};
  if (!image.linear) throw std::invalid_argument{"expected linear image"};
}
}  // namespace
void get_rendered_image(image_data& image, const trace_state& state) {
  auto& r    = cache();
  auto  lock = std::lock_guard{r.mutex};
  if (!device_has(r, state) || !r.host_stale) return yocto::get_rendered_image(image, state);
  check_linear_image(image, state);
  check(r.ctx, ythip_get_image(r.ctx, (float*)image.pixels.data()));
}
void get_albedo_image(image_data& image, const trace_state& state) {
  auto& r    = cache();
  auto  lock = std::lock_guard{r.mutex};
  if (!device_has(r, state) || !r.host_stale) return yocto::get_albedo_image(image, state);
  check_linear_image(image, state);
  check(r.ctx, ythip_get_albedo_image(r.ctx, (float*)image.pixels.data()));
}
void get_normal_image(image_data& image, const trace_state& state) {
  auto& r    = cache();
  auto  lock = std::lock_guard{r.mutex};
  if (!device_has(r, state) || !r.host_stale) return yocto::get_normal_image(image, state);
  check_linear_image(image, state);
  check(r.ctx, ythip_get_normal_image(r.ctx, (float*)image.pixels.data()));
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

namespace {
void tonemap_device(residency& r, const trace_state& state, float exposure, bool filmic, float* ldr, uint8_t* ldrb) {
  if (!device_has(r, state)) {  // bring the host copy over (image only is enough for this)
    ensure_context(r);
    check(r.ctx, ythip_state_create(r.ctx, state.width, state.height, 0, state.height));
    check(r.ctx, ythip_state_upload(r.ctx, (const float*)state.image.data(), (const float*)state.albedo.data(),
                     (const float*)state.normal.data(), state.hits.data(), (const uint64_t*)state.rngs.data(),
                     state.samples));
    r.state = &state, r.width = state.width, r.height = state.height, r.dev_samples = state.samples;
    r.host_stale = false;
  }
  check(r.ctx, ythip_tonemap_image(r.ctx, exposure, filmic ? 1 : 0, 1, ldr, ldrb));
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
  context.worker = std::async(std::launch::async, [&]() {
    if (context.stop) return;
    hip::trace_samples_resident(state, scene, bvh, lights, params);  // includes the denoise hand-off
    if (context.stop) return;
    context.done = true;
  });
}
// trace_cancel — yocto_trace.cpp:1652-1655
void trace_cancel(trace_context& context) {
  context.stop = true;
  if (context.worker.valid()) context.worker.get();
}
// trace_preview — yocto_trace.cpp:1660-1676
void trace_preview(color_image& image, trace_context& context, trace_state& state, const scene_data& scene,
    const trace_bvh& bvh, const trace_lights& lights, const trace_params& params) {
  auto pparams = params;
  pparams.resolution /= params.pratio;
  pparams.samples = 1;
  auto pstate     = yocto::make_trace_state(scene, pparams);
  hip::trace_samples(pstate, scene, bvh, lights, pparams);
  {  // the preview state dies here: forget its device mirror
    auto& r    = cache();
    auto  lock = std::lock_guard{r.mutex};
    if (r.state == &pstate) r.state = nullptr, r.dev_samples = -1, r.host_stale = false;
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
  if (r.ctx) ythip_destroy(r.ctx);
  r.ctx   = nullptr;
  r.scene = r.bvh = r.lights = {};
  r.state                    = nullptr;
  r.dev_samples              = -1;
  r.host_stale               = false;
}

}  // namespace yocto::hip
