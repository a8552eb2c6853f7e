// ref_harness.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// C-ABI wrapper around the UNMODIFIED reference (Yocto/GL, compiled in place
// from /root/reference/libs/yocto by oracle/Makefile with g++) so that tests and
// bench.py's cpu_baseline leg can drive the reference's own trace_samples /
// intersect_scene_bvh / make_trace_* on flat (ythip.h) scenes.  Output goes to
// oracle/_ref/libyocto_ref.so (git-ignored).  Nothing under yocto-gl_amd/ may
// link, import or call this.
//
// Only reference *headers* are included here; no reference source is copied.

#include <yocto/yocto_bvh.h>
#include <yocto/yocto_geometry.h>
#include <yocto/yocto_math.h>
#include <yocto/yocto_sampling.h>
#include <yocto/yocto_scene.h>
#include <yocto/yocto_sceneio.h>
#include <yocto/yocto_shading.h>
#include <yocto/yocto_shape.h>
#include <yocto/yocto_trace.h>

#include <chrono>
#include <cstring>
#include <memory>
#include <vector>

#include "../include/ythip.h"

using namespace yocto;

namespace {

struct flat_store {
  std::vector<ythip_camera>      cameras;
  std::vector<ythip_instance>    instances;
  std::vector<ythip_environment> environments;
  std::vector<ythip_shape>       shapes;
  std::vector<ythip_texture>     textures;
  std::vector<ythip_material>    materials;
  std::vector<int32_t>           points, lines, triangles, quads;
  std::vector<float>   positions, normals, texcoords, colors, radius, pixelsf;
  std::vector<uint8_t> pixelsb;
};

struct ref_scene {
  scene_data scene;
  flat_store flat;
};

struct ref_bvh {
  trace_bvh                   bvh;
  std::vector<int64_t>        node_offset, prim_offset;
  std::vector<ythip_bvh_node> nodes;
  std::vector<int32_t>        prims;
};

struct ref_lights {
  trace_lights             lights;
  std::vector<ythip_light> flat;
  std::vector<float>       cdf;
};

struct ref_state {
  trace_state state;
};

frame3f to_frame(const ythip_frame& f) {
  return frame3f{{f.x[0], f.x[1], f.x[2]}, {f.y[0], f.y[1], f.y[2]},
      {f.z[0], f.z[1], f.z[2]}, {f.o[0], f.o[1], f.o[2]}};
}
ythip_frame from_frame(const frame3f& f) {
  return ythip_frame{{f.x.x, f.x.y, f.x.z}, {f.y.x, f.y.y, f.y.z},
      {f.z.x, f.z.y, f.z.z}, {f.o.x, f.o.y, f.o.z}};
}

trace_params to_params(const ythip_params& p) {
  auto params           = trace_params{};
  params.camera         = p.camera;
  params.resolution     = p.resolution;
  params.sampler        = (trace_sampler_type)p.sampler;
  params.falsecolor     = (trace_falsecolor_type)p.falsecolor;
  params.samples        = p.samples;
  params.bounces        = p.bounces;
  params.clamp          = p.clamp;
  params.nocaustics     = p.nocaustics != 0;
  params.envhidden      = p.envhidden != 0;
  params.tentfilter     = p.tentfilter != 0;
  params.seed           = p.seed;
  params.embreebvh      = false;
  params.highqualitybvh = p.highqualitybvh != 0;
  params.noparallel     = p.noparallel != 0;
  params.pratio         = p.pratio;
  params.denoise        = false;
  params.batch          = p.batch;
  return params;
}

template <typename T, typename U>
int64_t append(std::vector<T>& pool, const std::vector<U>& src) {
  static_assert(sizeof(U) % sizeof(T) == 0);
  if (src.empty()) return -1;
  auto n   = sizeof(U) / sizeof(T);
  auto off = (int64_t)(pool.size() / n);
  auto ptr = (const T*)src.data();
  pool.insert(pool.end(), ptr, ptr + src.size() * n);
  return off;
}

void flatten(ref_scene& rs) {
  auto& s = rs.scene;
  auto& f = rs.flat;
  f       = {};
  for (auto& c : s.cameras) {
    f.cameras.push_back({from_frame(c.frame), c.orthographic ? 1 : 0, c.lens,
        c.film, c.aspect, c.focus, c.aperture});
  }
  for (auto& i : s.instances)
    f.instances.push_back({from_frame(i.frame), i.shape, i.material});
  for (auto& e : s.environments)
    f.environments.push_back({from_frame(e.frame),
        {e.emission.x, e.emission.y, e.emission.z}, e.emission_tex});
  for (auto& m : s.materials) {
    auto fm = ythip_material{};
    static_assert(sizeof(ythip_material) == sizeof(material_data));
    std::memcpy(&fm, &m, sizeof(fm));
    f.materials.push_back(fm);
  }
  for (auto& t : s.textures) {
    auto ft = ythip_texture{t.width, t.height, t.linear ? 1 : 0,
        t.nearest ? 1 : 0, t.clamp ? 1 : 0, t.pixelsf.empty() ? 0 : 1, 0};
    ft.offset = t.pixelsf.empty() ? append(f.pixelsb, t.pixelsb)
                                  : append(f.pixelsf, t.pixelsf);
    if (ft.offset < 0) ft.offset = 0;
    f.textures.push_back(ft);
  }
  for (auto& sh : s.shapes) {
    auto fs             = ythip_shape{};
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
}

void fill_flat(const ref_scene& rs, ythip_scene* out) {
  auto& f               = rs.flat;
  *out                  = ythip_scene{};
  out->num_cameras      = (int)f.cameras.size();
  out->num_instances    = (int)f.instances.size();
  out->num_environments = (int)f.environments.size();
  out->num_shapes       = (int)f.shapes.size();
  out->num_textures     = (int)f.textures.size();
  out->num_materials    = (int)f.materials.size();
  out->cameras          = f.cameras.data();
  out->instances        = f.instances.data();
  out->environments     = f.environments.data();
  out->shapes           = f.shapes.data();
  out->textures         = f.textures.data();
  out->materials        = f.materials.data();
  out->num_points       = (int64_t)f.points.size();
  out->num_lines        = (int64_t)f.lines.size() / 2;
  out->num_triangles    = (int64_t)f.triangles.size() / 3;
  out->num_quads        = (int64_t)f.quads.size() / 4;
  out->points           = f.points.data();
  out->lines            = f.lines.data();
  out->triangles        = f.triangles.data();
  out->quads            = f.quads.data();
  out->num_positions    = (int64_t)f.positions.size() / 3;
  out->num_normals      = (int64_t)f.normals.size() / 3;
  out->num_texcoords    = (int64_t)f.texcoords.size() / 2;
  out->num_colors       = (int64_t)f.colors.size() / 4;
  out->num_radius       = (int64_t)f.radius.size();
  out->positions        = f.positions.data();
  out->normals          = f.normals.data();
  out->texcoords        = f.texcoords.data();
  out->colors           = f.colors.data();
  out->radius           = f.radius.data();
  out->num_pixelsf      = (int64_t)f.pixelsf.size() / 4;
  out->num_pixelsb      = (int64_t)f.pixelsb.size() / 4;
  out->pixelsf          = f.pixelsf.data();
  out->pixelsb          = f.pixelsb.data();
}

template <typename U, typename T>
std::vector<U> slice(const T* pool, int64_t offset, int count) {
  auto out = std::vector<U>{};
  if (offset < 0 || count <= 0 || pool == nullptr) return out;
  auto n = sizeof(U) / sizeof(T);
  out.resize(count);
  std::memcpy(out.data(), pool + offset * n, sizeof(U) * count);
  return out;
}

ray3f to_ray(const ythip_ray& r) {
  return ray3f{{r.o[0], r.o[1], r.o[2]}, {r.d[0], r.d[1], r.d[2]}, r.tmin,
      r.tmax};
}

}  // namespace

extern "C" {

// ---------------------------------------------------------------------------
// scenes
// ---------------------------------------------------------------------------
ref_scene* ref_scene_new() { return new ref_scene{}; }
void       ref_scene_free(ref_scene* s) { delete s; }

// scene_data from flat arrays (Python-authored scenes)
ref_scene* ref_scene_from_flat(const ythip_scene* in) {
  auto  rs = new ref_scene{};
  auto& s  = rs->scene;
  for (auto k = 0; k < in->num_cameras; k++) {
    auto& c            = in->cameras[k];
    auto& camera       = s.cameras.emplace_back();
    camera.frame       = to_frame(c.frame);
    camera.orthographic = c.orthographic != 0;
    camera.lens        = c.lens;
    camera.film        = c.film;
    camera.aspect      = c.aspect;
    camera.focus       = c.focus;
    camera.aperture    = c.aperture;
  }
  for (auto k = 0; k < in->num_instances; k++) {
    auto& i = in->instances[k];
    s.instances.push_back({to_frame(i.frame), i.shape, i.material});
  }
  for (auto k = 0; k < in->num_environments; k++) {
    auto& e = in->environments[k];
    s.environments.push_back({to_frame(e.frame),
        {e.emission[0], e.emission[1], e.emission[2]}, e.emission_tex});
  }
  for (auto k = 0; k < in->num_materials; k++) {
    auto m = material_data{};
    std::memcpy(&m, &in->materials[k], sizeof(m));
    s.materials.push_back(m);
  }
  for (auto k = 0; k < in->num_textures; k++) {
    auto& t       = in->textures[k];
    auto& texture = s.textures.emplace_back();
    texture.width   = t.width;
    texture.height  = t.height;
    texture.linear  = t.linear != 0;
    texture.nearest = t.nearest != 0;
    texture.clamp   = t.clamp != 0;
    if (t.is_float)
      texture.pixelsf = slice<vec4f>(in->pixelsf, t.offset, t.width * t.height);
    else
      texture.pixelsb = slice<vec4b>(in->pixelsb, t.offset, t.width * t.height);
  }
  for (auto k = 0; k < in->num_shapes; k++) {
    auto& f      = in->shapes[k];
    auto& shape  = s.shapes.emplace_back();
    shape.points = slice<int>(in->points, f.points_offset, f.num_points);
    shape.lines  = slice<vec2i>(in->lines, f.lines_offset, f.num_lines);
    shape.triangles = slice<vec3i>(
        in->triangles, f.triangles_offset, f.num_triangles);
    shape.quads     = slice<vec4i>(in->quads, f.quads_offset, f.num_quads);
    shape.positions = slice<vec3f>(
        in->positions, f.positions_offset, f.num_positions);
    shape.normals   = slice<vec3f>(in->normals, f.normals_offset, f.num_normals);
    shape.texcoords = slice<vec2f>(
        in->texcoords, f.texcoords_offset, f.num_texcoords);
    shape.colors = slice<vec4f>(in->colors, f.colors_offset, f.num_colors);
    shape.radius = slice<float>(in->radius, f.radius_offset, f.num_radius);
  }
  flatten(*rs);
  return rs;
}

// reference generators --------------------------------------------------------
// make_cornellbox (yocto_scene.cpp:970-1075)
// load_scene (yocto_sceneio.h:201) + what apps/ytrace.cpp:103-120 does before rendering
// (tesselate_subdivs); returns null and keeps the message on failure
static thread_local std::string g_load_error;
const char* ref_load_error() { return g_load_error.c_str(); }
ref_scene*  ref_scene_load(const char* filename) {
  auto rs = new ref_scene{};
  try {
    rs->scene = load_scene(filename);
    if (!rs->scene.subdivs.empty()) tesselate_subdivs(rs->scene);
  } catch (const std::exception& e) {
    g_load_error = e.what();
    delete rs;
    return nullptr;
  }
  flatten(*rs);
  return rs;
}

// save_scene (yocto_sceneio.h:204) + make_scene_directories: writes the scene as the
// reference's own JSON + shapes/*.ply + textures, so that the reference's own apps can
// load it (tests: apps/ytrace.cpp on both back-ends).  0 on success.
int ref_scene_save(const ref_scene* rs, const char* filename) {
  try {
    make_scene_directories(filename, rs->scene);
    save_scene(filename, rs->scene);
  } catch (const std::exception& e) {
    g_load_error = e.what();
    return 1;
  }
  return 0;
}

// load_image (yocto_sceneio.h): reads back what the reference's apps saved.  Call with
// rgba == nullptr for the size, then with a width*height*4 float buffer.  0 on success.
int ref_image_load(const char* filename, int* width, int* height, float* rgba) {
  try {
    auto img = load_image(filename);
    *width = img.width, *height = img.height;
    if (rgba) std::memcpy(rgba, img.pixels.data(), img.pixels.size() * sizeof(vec4f));
  } catch (const std::exception& e) {
    g_load_error = e.what();
    return 1;
  }
  return 0;
}

ref_scene* ref_scene_cornellbox() {
  auto rs   = new ref_scene{};
  rs->scene = make_cornellbox();
  flatten(*rs);
  return rs;
}

// Shape generators; each appends one shape and returns its index.
// make_recty (+ optional quads_to_triangles) — yocto_shape.cpp:620-627,2535-2543
int ref_add_recty(ref_scene* rs, int stepx, int stepy, float scalex,
    float scaley, float uvx, float uvy, int triangulate) {
  auto shape = make_recty({stepx, stepy}, {scalex, scaley}, {uvx, uvy});
  if (triangulate) {
    shape.triangles = quads_to_triangles(shape.quads);
    shape.quads     = {};
  }
  rs->scene.shapes.push_back(shape);
  return (int)rs->scene.shapes.size() - 1;
}
// make_rect — yocto_shape.cpp:599-602
int ref_add_rect(ref_scene* rs, int stepx, int stepy, float scalex,
    float scaley, float uvx, float uvy, int triangulate) {
  auto shape = make_rect({stepx, stepy}, {scalex, scaley}, {uvx, uvy});
  if (triangulate) {
    shape.triangles = quads_to_triangles(shape.quads);
    shape.quads     = {};
  }
  rs->scene.shapes.push_back(shape);
  return (int)rs->scene.shapes.size() - 1;
}
// make_uvsphere — yocto_shape.cpp:783-796
int ref_add_uvsphere(ref_scene* rs, int stepx, int stepy, float scale,
    int triangulate) {
  auto shape = make_uvsphere({stepx, stepy}, scale, {1, 1});
  if (triangulate) {
    shape.triangles = quads_to_triangles(shape.quads);
    shape.quads     = {};
  }
  rs->scene.shapes.push_back(shape);
  return (int)rs->scene.shapes.size() - 1;
}
// make_sphere (cube-sphere, quads) — yocto_shape.cpp
int ref_add_sphere(ref_scene* rs, int steps, float scale, int triangulate) {
  auto shape = make_sphere(steps, scale, 1);
  if (triangulate) {
    shape.triangles = quads_to_triangles(shape.quads);
    shape.quads     = {};
  }
  rs->scene.shapes.push_back(shape);
  return (int)rs->scene.shapes.size() - 1;
}
// make_hair over an existing shape — yocto_shape.cpp:1264-1334
int ref_add_hair(ref_scene* rs, int base_shape, int stepx, int stepy,
    float len0, float len1, float rad0, float rad1) {
  auto shape = make_hair(rs->scene.shapes[base_shape], {stepx, stepy},
      {len0, len1}, {rad0, rad1});
  rs->scene.shapes.push_back(shape);
  return (int)rs->scene.shapes.size() - 1;
}
// make_points / make_random_points — yocto_shape.cpp
int ref_add_random_points(ref_scene* rs, int num, float sx, float sy, float sz,
    float radius) {
  auto shape = make_random_points(num, {sx, sy, sz}, 1, radius, 17);
  rs->scene.shapes.push_back(shape);
  return (int)rs->scene.shapes.size() - 1;
}
int ref_add_material(ref_scene* rs, const ythip_material* m) {
  auto material = material_data{};
  std::memcpy(&material, m, sizeof(material));
  rs->scene.materials.push_back(material);
  return (int)rs->scene.materials.size() - 1;
}
int ref_add_instance(ref_scene* rs, const ythip_frame* frame, int shape,
    int material) {
  rs->scene.instances.push_back({to_frame(*frame), shape, material});
  return (int)rs->scene.instances.size() - 1;
}
int ref_add_environment(ref_scene* rs, const ythip_frame* frame, float er,
    float eg, float eb, int emission_tex) {
  rs->scene.environments.push_back(
      {to_frame(*frame), {er, eg, eb}, emission_tex});
  return (int)rs->scene.environments.size() - 1;
}
// camera with lookat_frame (yocto_math.h:2348-2358)
int ref_add_camera_lookat(ref_scene* rs, float fx, float fy, float fz, float tx,
    float ty, float tz, float lens, float film, float aspect, float aperture,
    int orthographic) {
  auto& camera       = rs->scene.cameras.emplace_back();
  camera.frame       = lookat_frame({fx, fy, fz}, {tx, ty, tz}, {0, 1, 0});
  camera.lens        = lens;
  camera.film        = film;
  camera.aspect      = aspect;
  camera.aperture    = aperture;
  camera.focus       = length(vec3f{fx, fy, fz} - vec3f{tx, ty, tz});
  camera.orthographic = orthographic != 0;
  return (int)rs->scene.cameras.size() - 1;
}
// float texture from caller pixels
int ref_add_texture(ref_scene* rs, int width, int height, int linear,
    int nearest, int clamp, int is_float, const void* pixels) {
  auto& t   = rs->scene.textures.emplace_back();
  t.width   = width;
  t.height  = height;
  t.linear  = linear != 0;
  t.nearest = nearest != 0;
  t.clamp   = clamp != 0;
  if (is_float) {
    t.pixelsf.resize((size_t)width * height);
    std::memcpy(t.pixelsf.data(), pixels, sizeof(vec4f) * t.pixelsf.size());
  } else {
    t.pixelsb.resize((size_t)width * height);
    std::memcpy(t.pixelsb.data(), pixels, sizeof(vec4b) * t.pixelsb.size());
  }
  return (int)rs->scene.textures.size() - 1;
}
// make_sunsky environment texture (yocto_image.cpp) as a float texture
int ref_add_sunsky_texture(ref_scene* rs, int width, int height, float sun_angle) {
  auto img = make_sunsky(width, height, sun_angle, 3, false, 1, 1,
      vec3f{0.7f, 0.7f, 0.7f});
  auto& t   = rs->scene.textures.emplace_back();
  t.width   = img.width;
  t.height  = img.height;
  t.linear  = true;
  t.pixelsf = img.pixels;
  return (int)rs->scene.textures.size() - 1;
}
void ref_scene_commit(ref_scene* rs) { flatten(*rs); }
void ref_scene_flat(const ref_scene* rs, ythip_scene* out) {
  fill_flat(*rs, out);
}

// ---------------------------------------------------------------------------
// make_trace_bvh / make_trace_lights / make_trace_state
// ---------------------------------------------------------------------------
static void reflatten(ref_bvh* rb) {
  rb->node_offset.clear(), rb->prim_offset.clear(), rb->nodes.clear(), rb->prims.clear();
  auto& sb   = rb->bvh.bvh;
  auto  push = [&](const bvh_tree& tree) {
    rb->node_offset.push_back((int64_t)rb->nodes.size());
    rb->prim_offset.push_back((int64_t)rb->prims.size());
    static_assert(sizeof(bvh_node) == sizeof(ythip_bvh_node));
    auto n0 = rb->nodes.size();
    rb->nodes.resize(n0 + tree.nodes.size());
    if (!tree.nodes.empty())
      std::memcpy(rb->nodes.data() + n0, tree.nodes.data(),
          sizeof(bvh_node) * tree.nodes.size());
    rb->prims.insert(
        rb->prims.end(), tree.primitives.begin(), tree.primitives.end());
  };
  for (auto& shape : sb.shapes) push(shape.bvh);
  push(sb.bvh);
  rb->node_offset.push_back((int64_t)rb->nodes.size());
  rb->prim_offset.push_back((int64_t)rb->prims.size());
}
ref_bvh* ref_bvh_build(const ref_scene* rs, int highquality) {
  auto rb    = new ref_bvh{};
  auto params = trace_params{};
  params.highqualitybvh = highquality != 0;
  rb->bvh    = make_trace_bvh(rs->scene, params);
  reflatten(rb);
  return rb;
}
// scene edits that keep the element lists + update_scene_bvh (yocto_bvh.cpp:434-451)
int ref_scene_set_vertices(ref_scene* rs, int shape, const float* positions,
    int64_t num_positions, const float* normals, int64_t num_normals,
    const float* radius, int64_t num_radius) {
  if (shape < 0 || shape >= (int)rs->scene.shapes.size()) return -1;
  auto& sh = rs->scene.shapes[shape];
  if (positions) {
    if (num_positions != (int64_t)sh.positions.size()) return -1;
    std::memcpy((void*)sh.positions.data(), positions, sizeof(vec3f) * sh.positions.size());
  }
  if (normals) {
    if (num_normals != (int64_t)sh.normals.size()) return -1;
    std::memcpy((void*)sh.normals.data(), normals, sizeof(vec3f) * sh.normals.size());
  }
  if (radius) {
    if (num_radius != (int64_t)sh.radius.size()) return -1;
    std::memcpy(sh.radius.data(), radius, sizeof(float) * sh.radius.size());
  }
  return 0;
}
int ref_scene_set_instance_frame(ref_scene* rs, int instance, const ythip_frame* frame) {
  if (instance < 0 || instance >= (int)rs->scene.instances.size()) return -1;
  rs->scene.instances[instance].frame = to_frame(*frame);
  return 0;
}
double ref_bvh_update(ref_bvh* rb, const ref_scene* rs, const int* instances,
    int num_instances, const int* shapes, int num_shapes) {
  auto vi = std::vector<int>(instances, instances + num_instances);
  auto vs = std::vector<int>(shapes, shapes + num_shapes);
  auto t0 = std::chrono::steady_clock::now();
  update_scene_bvh(rb->bvh.bvh, rs->scene, vi, vs);
  auto dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  reflatten(rb);
  return dt;
}
void ref_bvh_free(ref_bvh* b) { delete b; }
void ref_bvh_flat(const ref_bvh* rb, ythip_bvh* out) {
  out->num_trees   = (int)rb->node_offset.size() - 1;
  out->node_offset = rb->node_offset.data();
  out->prim_offset = rb->prim_offset.data();
  out->nodes       = rb->nodes.data();
  out->primitives  = rb->prims.data();
}

ref_lights* ref_lights_build(const ref_scene* rs) {
  auto rl    = new ref_lights{};
  rl->lights = make_trace_lights(rs->scene, trace_params{});
  for (auto& light : rl->lights.lights) {
    rl->flat.push_back({light.instance, light.environment,
        (int64_t)rl->cdf.size(), (int)light.elements_cdf.size(), 0});
    rl->cdf.insert(
        rl->cdf.end(), light.elements_cdf.begin(), light.elements_cdf.end());
  }
  return rl;
}
void ref_lights_free(ref_lights* l) { delete l; }
void ref_lights_flat(const ref_lights* rl, ythip_lights* out) {
  out->num_lights = (int)rl->flat.size();
  out->lights     = rl->flat.data();
  out->num_cdf    = (int64_t)rl->cdf.size();
  out->cdf        = rl->cdf.data();
}

ref_state* ref_state_make(const ref_scene* rs, const ythip_params* p) {
  auto st   = new ref_state{};
  st->state = make_trace_state(rs->scene, to_params(*p));
  return st;
}
void ref_state_free(ref_state* s) { delete s; }
void ref_state_info(const ref_state* st, int* width, int* height, int* samples) {
  *width   = st->state.width;
  *height  = st->state.height;
  *samples = st->state.samples;
}
// copy out (any pointer may be null)
void ref_state_get(const ref_state* st, float* image, float* albedo,
    float* normal, int32_t* hits, uint64_t* rngs) {
  auto& s = st->state;
  auto  n = (size_t)s.width * s.height;
  if (image) std::memcpy(image, s.image.data(), n * sizeof(vec4f));
  if (albedo) std::memcpy(albedo, s.albedo.data(), n * sizeof(vec3f));
  if (normal) std::memcpy(normal, s.normal.data(), n * sizeof(vec3f));
  if (hits) std::memcpy(hits, s.hits.data(), n * sizeof(int));
  if (rngs) std::memcpy(rngs, s.rngs.data(), n * sizeof(rng_state));
}
void ref_state_set(ref_state* st, const float* image, const float* albedo,
    const float* normal, const int32_t* hits, const uint64_t* rngs,
    int samples) {
  auto& s = st->state;
  auto  n = (size_t)s.width * s.height;
  if (image) std::memcpy(s.image.data(), image, n * sizeof(vec4f));
  if (albedo) std::memcpy(s.albedo.data(), albedo, n * sizeof(vec3f));
  if (normal) std::memcpy(s.normal.data(), normal, n * sizeof(vec3f));
  if (hits) std::memcpy(s.hits.data(), hits, n * sizeof(int));
  if (rngs) std::memcpy(s.rngs.data(), rngs, n * sizeof(rng_state));
  s.samples = samples;
}

// ---------------------------------------------------------------------------
// the hot path, reference implementation
// ---------------------------------------------------------------------------
// trace_samples (yocto_trace.cpp:1595-1619); returns wall seconds
double ref_trace_samples(ref_state* st, const ref_scene* rs, const ref_bvh* rb,
    const ref_lights* rl, const ythip_params* p) {
  auto params = to_params(*p);
  auto t0     = std::chrono::steady_clock::now();
  trace_samples(st->state, rs->scene, rb->bvh, rl->lights, params);
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double>(t1 - t0).count();
}

// trace_sample (yocto_trace.cpp:1461-1492): one sample of one pixel
void ref_trace_sample(ref_state* st, const ref_scene* rs, const ref_bvh* rb,
    const ref_lights* rl, const ythip_params* p, int i, int j, int sample) {
  trace_sample(st->state, rs->scene, rb->bvh, rl->lights, i, j, sample, to_params(*p));
}
// get_albedo_image / get_normal_image (yocto_trace.cpp:1769-1791): which = 0 / 1
void ref_guide_image(const ref_state* st, int which, float* out) {
  auto img = which == 0 ? get_albedo_image(st->state) : get_normal_image(st->state);
  std::memcpy(out, img.pixels.data(), img.pixels.size() * sizeof(vec4f));
}

// intersect_scene_bvh (yocto_bvh.cpp:554-617) over a ray batch
void ref_intersect_batch(const ref_bvh* rb, const ref_scene* rs,
    const ythip_ray* rays, int64_t n, int find_any, ythip_hit* hits) {
  for (auto k = (int64_t)0; k < n; k++) {
    auto isec = intersect_scene_bvh(
        rb->bvh.bvh, rs->scene, to_ray(rays[k]), find_any != 0);
    hits[k] = {isec.instance, isec.element, isec.uv.x, isec.uv.y,
        isec.distance, isec.hit ? 1 : 0};
  }
}
// intersect_instance_bvh (yocto_bvh.cpp:619-628)
void ref_intersect_instance_batch(const ref_bvh* rb, const ref_scene* rs,
    const int32_t* instances, const ythip_ray* rays, int64_t n, int find_any,
    ythip_hit* hits) {
  for (auto k = (int64_t)0; k < n; k++) {
    auto isec = intersect_instance_bvh(
        rb->bvh.bvh, rs->scene, instances[k], to_ray(rays[k]), find_any != 0);
    hits[k] = {isec.instance, isec.element, isec.uv.x, isec.uv.y,
        isec.distance, isec.hit ? 1 : 0};
  }
}

// Primary rays for the NEXT sample of every pixel, using the reference's public
// eval_camera/sample_disk/rand2f and the g++ draw order of trace_sample
// (yocto_trace.cpp:1467-1468: luv is drawn before puv).  The state's rngs are
// NOT advanced.  sample_camera itself is file-static in the reference; the
// falsecolor=position render through ref_trace_samples pins this convention.
void ref_camera_rays(const ref_state* st, const ref_scene* rs,
    const ythip_params* p, ythip_ray* rays) {
  auto& state  = st->state;
  auto& camera = rs->scene.cameras[p->camera];
  for (auto j = 0; j < state.height; j++) {
    for (auto i = 0; i < state.width; i++) {
      auto idx = state.width * j + i;
      auto rng = state.rngs[idx];
      auto luv = rand2f(rng);
      auto puv = rand2f(rng);
      if (p->tentfilter) {
        auto tent = [](float u) {
          return u < 0.5f ? sqrt(2 * u) - 1 : 1 - sqrt(2 - 2 * u);
        };
        puv = 2.0f * vec2f{tent(puv.x), tent(puv.y)} + 0.5f;
      }
      auto uv  = vec2f{(i + puv.x) / state.width, (j + puv.y) / state.height};
      auto ray = eval_camera(camera, uv, sample_disk(luv));
      rays[idx] = {{ray.o.x, ray.o.y, ray.o.z}, {ray.d.x, ray.d.y, ray.d.z},
          ray.tmin, ray.tmax};
    }
  }
}

// Stage-level oracles built from the reference's public functions -----------
// eval_shading_position / eval_shading_normal / eval_material at hit points
// (yocto_scene.cpp:469-581).  out: 3 pos + 3 normal + 19 material floats
// {type, emission3, color3, opacity, roughness, metallic, ior, density3,
//  scattering3, scanisotropy, trdepth}
void ref_eval_shading(const ref_scene* rs, const ythip_hit* hits,
    const float* outgoing, int64_t n, float* out) {
  for (auto k = (int64_t)0; k < n; k++) {
    auto& h  = hits[k];
    auto  o  = out + k * 25;
    auto  wo = vec3f{outgoing[k * 3], outgoing[k * 3 + 1], outgoing[k * 3 + 2]};
    if (!h.hit) {
      for (auto c = 0; c < 25; c++) o[c] = 0;
      continue;
    }
    auto& instance = rs->scene.instances[h.instance];
    auto  uv       = vec2f{h.u, h.v};
    auto  pos = eval_shading_position(rs->scene, instance, h.element, uv, wo);
    auto  nrm = eval_shading_normal(rs->scene, instance, h.element, uv, wo);
    auto  m   = eval_material(rs->scene, instance, h.element, uv);
    o[0] = pos.x, o[1] = pos.y, o[2] = pos.z;
    o[3] = nrm.x, o[4] = nrm.y, o[5] = nrm.z;
    o[6] = (float)(int)m.type;
    o[7] = m.emission.x, o[8] = m.emission.y, o[9] = m.emission.z;
    o[10] = m.color.x, o[11] = m.color.y, o[12] = m.color.z;
    o[13] = m.opacity, o[14] = m.roughness, o[15] = m.metallic, o[16] = m.ior;
    o[17] = m.density.x, o[18] = m.density.y, o[19] = m.density.z;
    o[20] = m.scattering.x, o[21] = m.scattering.y, o[22] = m.scattering.z;
    o[23] = m.scanisotropy, o[24] = m.trdepth;
  }
}

// eval_environment (yocto_scene.cpp:596-613)
void ref_eval_environment(const ref_scene* rs, const float* dirs, int64_t n,
    float* out) {
  for (auto k = (int64_t)0; k < n; k++) {
    auto e = eval_environment(
        rs->scene, vec3f{dirs[k * 3], dirs[k * 3 + 1], dirs[k * 3 + 2]});
    out[k * 3] = e.x, out[k * 3 + 1] = e.y, out[k * 3 + 2] = e.z;
  }
}

// PCG32 known answers (yocto_sampling.h:187-232)
// tonemap_image (yocto_image.cpp:911-922) of n vec4f pixels: float and byte results
void ref_tonemap(const float* hdr, int64_t n, float exposure, int filmic, int srgb, float* ldr, uint8_t* ldrb) {
  auto h = std::vector<vec4f>((const vec4f*)hdr, (const vec4f*)hdr + n);
  auto f = std::vector<vec4f>{};
  auto b = std::vector<vec4b>{};
  tonemap_image(f, h, exposure, filmic != 0, srgb != 0);
  tonemap_image(b, h, exposure, filmic != 0, srgb != 0);
  std::memcpy(ldr, f.data(), (size_t)n * sizeof(vec4f));
  std::memcpy(ldrb, b.data(), (size_t)n * sizeof(vec4b));
}

void ref_make_rng(uint64_t seed, uint64_t seq, uint64_t* out) {
  auto rng = make_rng(seed, seq);
  out[0]   = rng.state;
  out[1]   = rng.inc;
}
void ref_rand1f(uint64_t* state, int n, float* out) {
  auto rng = rng_state{state[0], state[1]};
  for (auto k = 0; k < n; k++) out[k] = rand1f(rng);
  state[0] = rng.state;
  state[1] = rng.inc;
}

// save_trace_params / load_trace_params (yocto_sceneio.cpp:5933-5945): the reference's own
// parameter files, for tests/test_io.py (ythip_params_to_json / _from_json must interoperate)
int ref_params_save(const ythip_params* p, const char* filename) {
  try {
    auto params      = to_params(*p);
    params.embreebvh = p->embreebvh != 0;
    params.denoise   = p->denoise != 0;
    save_trace_params(filename, params);
  } catch (const std::exception& e) {
    g_load_error = e.what();
    return 1;
  }
  return 0;
}
int ref_params_load(const char* filename, ythip_params* p) {
  try {
    auto params = to_params(*p);  // (update semantics: absent keys keep the incoming values)
    params.embreebvh = p->embreebvh != 0;
    params.denoise   = p->denoise != 0;
    update_trace_params(filename, params);
    p->camera = params.camera, p->resolution = params.resolution, p->sampler = (int)params.sampler;
    p->falsecolor = (int)params.falsecolor, p->samples = params.samples, p->bounces = params.bounces;
    p->clamp = params.clamp, p->nocaustics = params.nocaustics, p->envhidden = params.envhidden;
    p->tentfilter = params.tentfilter, p->seed = params.seed, p->embreebvh = params.embreebvh;
    p->highqualitybvh = params.highqualitybvh, p->noparallel = params.noparallel, p->pratio = params.pratio;
    p->denoise = params.denoise, p->batch = params.batch;
  } catch (const std::exception& e) {
    g_load_error = e.what();
    return 1;
  }
  return 0;
}

int ref_hardware_concurrency() {
  return (int)std::thread::hardware_concurrency();
}

}  // extern "C"
