// yt_oracle.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A single-threaded, plain-CPU RESTATEMENT of the reference's hot path, written
// from the reference sources function by function (each function cites the
// file:line under /root/reference/libs/yocto it follows).  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline may load it; the product
// (libythip.so) never does.
//
// Scope (what SURVEY.md §8a puts on the path, restated for the BASELINE
// workloads):
//   * PCG32 + make_trace_state seeding + image size rule
//   * sample_camera / eval_camera
//   * intersect_bbox, intersect_triangle/quad/line/point,
//     intersect_shape_bvh, intersect_scene_bvh, intersect_instance_bvh
//   * eval_position / element_normal / normal / texcoord / color,
//     eval_shading_position / eval_shading_normal (no normal map)
//   * eval_material (no textures), eval_environment (no texture), is_delta
//   * matte lobe (eval / sample / pdf), eval_emission
//   * sample_lights / sample_lights_pdf (area lights + constant environments)
//   * trace_path, trace_naive, trace_eyelight, trace_sample, trace_samples
// Materials other than `matte`, textures and environment maps are NOT restated:
// yto_trace_samples refuses such scenes (error, never a silent approximation);
// for those the checker is the compiled reference itself (oracle/_ref).
//
// Pinning: tests/test_oracle.py checks this file against the reference's own
// known answers (SURVEY.md §8c KATs), against the golden fixtures generated
// from the compiled reference (tests/golden/), and — where oracle/_ref is
// present — bit for bit against the live reference on the same inputs.
//
// Bit-exactness rules: compiled with g++ (never clang), no FMA contraction, no
// fast-math; every expression keeps the reference's association order; the rng
// draw order at multi-argument call sites is g++'s right-to-left order, which
// is what the g++-built reference executes — written
// here as explicit statements so the restatement does not depend on it.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../include/ythip.h"

namespace {

thread_local std::string g_err;
int fail(const std::string& msg) {
  g_err = msg;
  return 1;
}

constexpr float pif     = 3.14159265358979323846f;  // yocto_math.h:57
constexpr float flt_max = 3.402823466e+38f;          // yocto_math.h:66
constexpr float ray_eps = 1e-4f;                     // yocto_geometry.h:125

// --- yocto_math.h scalar helpers :1045-1050 ---------------------------------
inline float abs_(float a) { return a < 0 ? -a : a; }
inline float min_(float a, float b) { return (a < b) ? a : b; }
inline float max_(float a, float b) { return (a > b) ? a : b; }
inline float clamp_(float a, float lo, float hi) { return min_(max_(a, lo), hi); }
inline int   clampi(int a, int lo, int hi) { return std::min(std::max(a, lo), hi); }

// --- vec3f, yocto_math.h:1256-1394 ------------------------------------------
struct vec2f {
  float x, y;
};
struct vec3f {
  float x, y, z;
};
struct vec4f {
  float x, y, z, w;
};
inline vec3f operator-(vec3f a) { return {-a.x, -a.y, -a.z}; }
inline vec3f operator+(vec3f a, vec3f b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3f operator-(vec3f a, vec3f b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3f operator*(vec3f a, vec3f b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3f operator*(vec3f a, float b) { return {a.x * b, a.y * b, a.z * b}; }
inline vec3f operator*(float a, vec3f b) { return {a * b.x, a * b.y, a * b.z}; }
inline vec3f operator/(vec3f a, vec3f b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline vec3f operator/(vec3f a, float b) { return {a.x / b, a.y / b, a.z / b}; }
inline bool  operator==(vec3f a, vec3f b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline vec3f& operator+=(vec3f& a, vec3f b) { return a = a + b; }
inline vec3f& operator*=(vec3f& a, vec3f b) { return a = a * b; }
inline vec3f& operator*=(vec3f& a, float b) { return a = a * b; }
inline float dot(vec3f a, vec3f b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // :1305
inline vec3f cross(vec3f a, vec3f b) {                                            // :1308
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float length(vec3f a) { return std::sqrt(dot(a, a)); }  // :1312
inline vec3f normalize(vec3f a) {                              // :1314
  auto l = length(a);
  return (l != 0) ? a / l : a;
}
inline float distance_squared(vec3f a, vec3f b) { return dot(a - b, a - b); }   // :1319
inline vec3f orthonormalize(vec3f a, vec3f b) { return normalize(a - b * dot(a, b)); }  // :1331
inline vec3f lerp(vec3f a, vec3f b, float u) { return a * (1 - u) + b * u; }    // :1362
inline float max3(vec3f a) { return max_(max_(a.x, a.y), a.z); }                // :1369
inline float min3(vec3f a) { return min_(min_(a.x, a.y), a.z); }                // :1370
inline vec3f vmin(vec3f a, vec3f b) { return {min_(a.x, b.x), min_(a.y, b.y), min_(a.z, b.z)}; }
inline vec3f vmax(vec3f a, vec3f b) { return {max_(a.x, b.x), max_(a.y, b.y), max_(a.z, b.z)}; }
inline bool  isfinite3(vec3f a) { return std::isfinite(a.x) && std::isfinite(a.y) && std::isfinite(a.z); }
inline vec4f operator*(vec4f a, float b) { return {a.x * b, a.y * b, a.z * b, a.w * b}; }
inline vec4f operator+(vec4f a, vec4f b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline vec4f lerp(vec4f a, vec4f b, float u) { return a * (1 - u) + b * u; }    // :1512
inline vec2f operator*(vec2f a, float b) { return {a.x * b, a.y * b}; }
inline vec2f operator+(vec2f a, vec2f b) { return {a.x + b.x, a.y + b.y}; }
inline vec4f operator*(vec4f a, vec4f b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }

// --- mat3f / frame3f, yocto_math.h:1960-2125, 2235-2275 ---------------------
struct mat3f {
  vec3f x, y, z;
};
struct frame3f {
  vec3f x, y, z, o;
};
inline frame3f ldframe(const ythip_frame& f) {
  return {{f.x[0], f.x[1], f.x[2]}, {f.y[0], f.y[1], f.y[2]}, {f.z[0], f.z[1], f.z[2]}, {f.o[0], f.o[1], f.o[2]}};
}
inline vec3f operator*(const mat3f& a, vec3f b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline mat3f operator*(const mat3f& a, float b) { return {a.x * b, a.y * b, a.z * b}; }
inline mat3f transpose(const mat3f& a) {
  return {{a.x.x, a.y.x, a.z.x}, {a.x.y, a.y.y, a.z.y}, {a.x.z, a.y.z, a.z.z}};
}
inline float determinant(const mat3f& a) { return dot(a.x, cross(a.y, a.z)); }  // :1967
inline mat3f adjoint(const mat3f& a) {                                          // :1968
  return transpose(mat3f{cross(a.y, a.z), cross(a.z, a.x), cross(a.x, a.y)});
}
inline mat3f inverse(const mat3f& a) { return adjoint(a) * (1 / determinant(a)); }  // :1971
inline mat3f rotation(const frame3f& a) { return {a.x, a.y, a.z}; }
inline frame3f inverse(const frame3f& a, bool non_rigid) {  // :2114
  if (non_rigid) {
    auto minv = inverse(rotation(a));
    auto t    = minv * a.o;
    return {minv.x, minv.y, minv.z, -t};
  } else {
    auto minv = transpose(rotation(a));
    auto t    = minv * a.o;
    return {minv.x, minv.y, minv.z, -t};
  }
}
inline mat3f basis_fromz(vec3f v) {  // :1977 (Pixar ONB)
  auto z    = normalize(v);
  auto sign = copysignf(1.0f, z.z);
  auto a    = -1.0f / (sign + z.z);
  auto b    = z.x * z.y * a;
  auto x    = vec3f{1.0f + sign * z.x * z.x * a, sign * b, -sign * z.x};
  auto y    = vec3f{b, sign + z.y * z.y * a, -z.y};
  return {x, y, z};
}
inline vec3f transform_direction(const mat3f& a, vec3f b) { return normalize(a * b); }  // :2236
inline vec3f transform_point(const frame3f& a, vec3f b) {   // :2262
  return a.x * b.x + a.y * b.y + a.z * b.z + a.o;
}
inline vec3f transform_vector(const frame3f& a, vec3f b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // :2265
inline vec3f transform_direction(const frame3f& a, vec3f b) { return normalize(transform_vector(a, b)); }
// transform_normal(frame, n, non_rigid = false): the path always takes the default
inline vec3f transform_normal(const frame3f& a, vec3f b) { return normalize(transform_vector(a, b)); }  // :2271

// --- ray3f, yocto_geometry.h:135-140, transform_ray :441-443 -----------------
struct ray3f {
  vec3f o, d;
  float tmin = ray_eps, tmax = flt_max;
};
inline ray3f transform_ray(const frame3f& a, const ray3f& b) {
  return {transform_point(a, b.o), transform_vector(a, b.d), b.tmin, b.tmax};
}

// --- PCG32, yocto_sampling.h:183-232 ------------------------------------------
struct rng_state {
  uint64_t state, inc;
};
inline uint32_t advance_rng(rng_state& rng) {  // :187
  uint64_t oldstate = rng.state;
  rng.state         = oldstate * 6364136223846793005ULL + rng.inc;
  auto xorshifted   = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
  auto rot          = (uint32_t)(oldstate >> 59u);
  return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}
inline rng_state make_rng(uint64_t seed, uint64_t seq = 1) {  // :197
  rng_state rng;
  rng.state = 0U;
  rng.inc   = (seq << 1u) | 1u;
  advance_rng(rng);
  rng.state += seed;
  advance_rng(rng);
  return rng;
}
inline int   rand1i(rng_state& rng, int n) { return advance_rng(rng) % n; }  // :208
inline float rand1f(rng_state& rng) {                                        // :209
  union {
    uint32_t u;
    float    f;
  } x;
  x.u = (advance_rng(rng) >> 9) | 0x3f800000u;
  return x.f - 1.0f;
}
inline vec2f rand2f(rng_state& rng) {  // :220
  auto x = rand1f(rng);
  auto y = rand1f(rng);
  return {x, y};
}

// --- sampling, yocto_sampling.h:274-398 ---------------------------------------
inline vec3f sample_sphere(vec2f ruv) {  // :276
  auto z   = 2 * ruv.y - 1;
  auto r   = std::sqrt(clamp_(1 - z * z, 0.0f, 1.0f));
  auto phi = 2 * pif * ruv.x;
  return {r * std::cos(phi), r * std::sin(phi), z};
}
inline vec3f sample_hemisphere_cos(vec3f normal, vec2f ruv) {  // :296
  auto z               = std::sqrt(ruv.y);
  auto r               = std::sqrt(1 - z * z);
  auto phi             = 2 * pif * ruv.x;
  auto local_direction = vec3f{r * std::cos(phi), r * std::sin(phi), z};
  return transform_direction(basis_fromz(normal), local_direction);
}
inline float sample_hemisphere_cos_pdf(vec3f normal, vec3f direction) {  // :303
  auto cosw = dot(normal, direction);
  return (cosw <= 0) ? 0 : cosw / pif;
}
inline vec2f sample_disk(vec2f ruv) {  // :339
  auto r   = std::sqrt(ruv.y);
  auto phi = 2 * pif * ruv.x;
  return {std::cos(phi) * r, std::sin(phi) * r};
}
inline vec2f sample_triangle(vec2f ruv) { return {1 - std::sqrt(ruv.x), ruv.y * std::sqrt(ruv.x)}; }  // :354
inline int   sample_uniform(int size, float r) { return clampi((int)(r * size), 0, size - 1); }        // :371
inline float sample_uniform_pdf(int size) { return (float)1 / (float)size; }                           // :374
inline int   sample_discrete(const float* cdf, int n, float r) {                                       // :388
  r        = clamp_(r * cdf[n - 1], (float)0, cdf[n - 1] - (float)0.00001);
  auto idx = (int)(std::upper_bound(cdf, cdf + n, r) - cdf);
  return clampi(idx, 0, n - 1);
}

// --- geometry, yocto_geometry.h:505-556 ---------------------------------------
inline vec3f line_tangent(vec3f p0, vec3f p1) { return normalize(p1 - p0); }
inline vec3f triangle_normal(vec3f p0, vec3f p1, vec3f p2) { return normalize(cross(p1 - p0, p2 - p0)); }
inline vec3f quad_normal(vec3f p0, vec3f p1, vec3f p2, vec3f p3) {
  return normalize(triangle_normal(p0, p1, p3) + triangle_normal(p2, p3, p1));
}
template <typename T>
inline T interpolate_line(T p0, T p1, float u) {
  return p0 * (1 - u) + p1 * u;
}
template <typename T>
inline T interpolate_triangle(T p0, T p1, T p2, vec2f uv) {
  return p0 * (1 - uv.x - uv.y) + p1 * uv.x + p2 * uv.y;
}
template <typename T>
inline T interpolate_quad(T p0, T p1, T p2, T p3, vec2f uv) {
  if (uv.x + uv.y <= 1) {
    return interpolate_triangle(p0, p1, p3, uv);
  } else {
    return interpolate_triangle(p2, p3, p1, vec2f{1 - uv.x, 1 - uv.y});
  }
}

// --- intersectors, yocto_geometry.h:697-864 -----------------------------------
struct prim_intersection {
  vec2f uv       = {0, 0};
  float distance = flt_max;
  bool  hit      = false;
};
inline prim_intersection intersect_point(const ray3f& ray, vec3f p, float r) {  // :697
  auto w = p - ray.o;
  auto t = dot(w, ray.d) / dot(ray.d, ray.d);
  if (t < ray.tmin || t > ray.tmax) return {};
  auto rp  = ray.o + ray.d * t;
  auto prp = p - rp;
  if (dot(prp, prp) > r * r) return {};
  return {{0, 0}, t, true};
}
inline prim_intersection intersect_line(const ray3f& ray, vec3f p0, vec3f p1, float r0, float r1) {  // :716
  auto u   = ray.d;
  auto v   = p1 - p0;
  auto w   = ray.o - p0;
  auto a   = dot(u, u);
  auto b   = dot(u, v);
  auto c   = dot(v, v);
  auto d   = dot(u, w);
  auto e   = dot(v, w);
  auto det = a * c - b * b;
  if (det == 0) return {};
  auto t = (b * e - c * d) / det;
  auto s = (a * e - b * d) / det;
  if (t < ray.tmin || t > ray.tmax) return {};
  s        = clamp_(s, (float)0, (float)1);
  auto pr  = ray.o + ray.d * t;
  auto pl  = p0 + (p1 - p0) * s;
  auto prl = pr - pl;
  auto d2  = dot(prl, prl);
  auto r   = r0 * (1 - s) + r1 * s;
  if (d2 > r * r) return {};
  return {{s, std::sqrt(d2) / r}, t, true};
}
inline prim_intersection intersect_triangle(const ray3f& ray, vec3f p0, vec3f p1, vec3f p2) {  // :794
  auto edge1 = p1 - p0;
  auto edge2 = p2 - p0;
  auto pvec  = cross(ray.d, edge2);
  auto det   = dot(edge1, pvec);
  if (det == 0) return {};
  auto inv_det = 1.0f / det;
  auto tvec    = ray.o - p0;
  auto u       = dot(tvec, pvec) * inv_det;
  if (u < 0 || u > 1) return {};
  auto qvec = cross(tvec, edge1);
  auto v    = dot(ray.d, qvec) * inv_det;
  if (v < 0 || u + v > 1) return {};
  auto t = dot(edge2, qvec) * inv_det;
  if (t < ray.tmin || t > ray.tmax) return {};
  return {{u, v}, t, true};
}
inline prim_intersection intersect_quad(const ray3f& ray, vec3f p0, vec3f p1, vec3f p2, vec3f p3) {  // :828
  if (p2 == p3) return intersect_triangle(ray, p0, p1, p3);
  auto isec1 = intersect_triangle(ray, p0, p1, p3);
  auto isec2 = intersect_triangle(ray, p2, p3, p1);
  if (isec2.hit) isec2.uv = {1 - isec2.uv.x, 1 - isec2.uv.y};
  return isec1.distance < isec2.distance ? isec1 : isec2;
}
inline bool intersect_bbox(const ray3f& ray, vec3f ray_dinv, vec3f bmin, vec3f bmax) {  // :854
  auto it_min = (bmin - ray.o) * ray_dinv;
  auto it_max = (bmax - ray.o) * ray_dinv;
  auto tmin   = vmin(it_min, it_max);
  auto tmax   = vmax(it_min, it_max);
  auto t0     = max_(max3(tmin), ray.tmin);
  auto t1     = min_(min3(tmax), ray.tmax);
  t1 *= 1.00000024f;
  return t0 <= t1;
}

// --- scene access over the flat POD scene (ythip_scene mirrors scene_data) ----
struct Scene {
  const ythip_scene*  sc;
  const ythip_bvh*    bvh;
  const ythip_lights* lights;
};
inline vec3f pos(const ythip_scene& sc, const ythip_shape& sh, int v) {
  const float* p = sc.positions + 3 * (sh.positions_offset + v);
  return {p[0], p[1], p[2]};
}
inline vec3f nrm(const ythip_scene& sc, const ythip_shape& sh, int v) {
  const float* p = sc.normals + 3 * (sh.normals_offset + v);
  return {p[0], p[1], p[2]};
}
inline vec2f tex(const ythip_scene& sc, const ythip_shape& sh, int v) {
  const float* p = sc.texcoords + 2 * (sh.texcoords_offset + v);
  return {p[0], p[1]};
}
inline vec4f col(const ythip_scene& sc, const ythip_shape& sh, int v) {
  const float* p = sc.colors + 4 * (sh.colors_offset + v);
  return {p[0], p[1], p[2], p[3]};
}
inline float rad(const ythip_scene& sc, const ythip_shape& sh, int v) { return sc.radius[sh.radius_offset + v]; }
inline const int* tri(const ythip_scene& sc, const ythip_shape& sh, int e) { return sc.triangles + 3 * (sh.triangles_offset + e); }
inline const int* quad(const ythip_scene& sc, const ythip_shape& sh, int e) { return sc.quads + 4 * (sh.quads_offset + e); }
inline const int* line(const ythip_scene& sc, const ythip_shape& sh, int e) { return sc.lines + 2 * (sh.lines_offset + e); }
inline int        point(const ythip_scene& sc, const ythip_shape& sh, int e) { return sc.points[sh.points_offset + e]; }

// --- BVH walkers, yocto_bvh.cpp:460-628 ---------------------------------------
struct shape_intersection {
  int   element  = -1;
  vec2f uv       = {0, 0};
  float distance = 0;
  bool  hit      = false;
};
struct scene_intersection {
  int   instance = -1;
  int   element  = -1;
  vec2f uv       = {0, 0};
  float distance = 0;
  bool  hit      = false;
};

// intersect_shape_bvh — yocto_bvh.cpp:460-552
shape_intersection intersect_shape_bvh(const Scene& S, int shape_id, const ray3f& ray_, bool find_any) {
  const auto& sc     = *S.sc;
  const auto& shape  = sc.shapes[shape_id];
  const auto* nodes  = S.bvh->nodes + S.bvh->node_offset[shape_id];
  const auto* prims  = S.bvh->primitives + S.bvh->prim_offset[shape_id];
  auto        nnodes = S.bvh->node_offset[shape_id + 1] - S.bvh->node_offset[shape_id];
  if (nnodes == 0) return {};  // :466

  int node_stack[128];
  int node_cur           = 0;
  node_stack[node_cur++] = 0;
  auto intersection      = shape_intersection{};
  auto ray               = ray_;
  auto ray_dinv          = vec3f{1 / ray.d.x, 1 / ray.d.y, 1 / ray.d.z};
  int  ray_dsign[3]      = {(ray_dinv.x < 0) ? 1 : 0, (ray_dinv.y < 0) ? 1 : 0, (ray_dinv.z < 0) ? 1 : 0};

  while (node_cur != 0) {
    const auto& node = nodes[node_stack[--node_cur]];
    if (!intersect_bbox(ray, ray_dinv, {node.bbox_min[0], node.bbox_min[1], node.bbox_min[2]},
            {node.bbox_max[0], node.bbox_max[1], node.bbox_max[2]}))
      continue;
    if (node.internal) {
      if (ray_dsign[node.axis] != 0) {
        node_stack[node_cur++] = node.start + 0;
        node_stack[node_cur++] = node.start + 1;
      } else {
        node_stack[node_cur++] = node.start + 1;
        node_stack[node_cur++] = node.start + 0;
      }
    } else if (shape.num_points) {  // dispatch order points → lines → triangles → quads, :505-545
      for (auto idx = node.start; idx < node.start + node.num; idx++) {
        auto p             = point(sc, shape, prims[idx]);
        auto pintersection = intersect_point(ray, pos(sc, shape, p), rad(sc, shape, p));
        if (!pintersection.hit) continue;
        intersection = {prims[idx], pintersection.uv, pintersection.distance, true};
        ray.tmax     = pintersection.distance;
      }
    } else if (shape.num_lines) {
      for (auto idx = node.start; idx < node.start + node.num; idx++) {
        auto l             = line(sc, shape, prims[idx]);
        auto pintersection = intersect_line(
            ray, pos(sc, shape, l[0]), pos(sc, shape, l[1]), rad(sc, shape, l[0]), rad(sc, shape, l[1]));
        if (!pintersection.hit) continue;
        intersection = {prims[idx], pintersection.uv, pintersection.distance, true};
        ray.tmax     = pintersection.distance;
      }
    } else if (shape.num_triangles) {
      for (auto idx = node.start; idx < node.start + node.num; idx++) {
        auto t             = tri(sc, shape, prims[idx]);
        auto pintersection = intersect_triangle(ray, pos(sc, shape, t[0]), pos(sc, shape, t[1]), pos(sc, shape, t[2]));
        if (!pintersection.hit) continue;
        intersection = {prims[idx], pintersection.uv, pintersection.distance, true};
        ray.tmax     = pintersection.distance;
      }
    } else if (shape.num_quads) {
      for (auto idx = node.start; idx < node.start + node.num; idx++) {
        auto q             = quad(sc, shape, prims[idx]);
        auto pintersection = intersect_quad(
            ray, pos(sc, shape, q[0]), pos(sc, shape, q[1]), pos(sc, shape, q[2]), pos(sc, shape, q[3]));
        if (!pintersection.hit) continue;
        intersection = {prims[idx], pintersection.uv, pintersection.distance, true};
        ray.tmax     = pintersection.distance;
      }
    }
    if (find_any && intersection.hit) return intersection;  // :548
  }
  return intersection;
}

// intersect_scene_bvh — yocto_bvh.cpp:554-617
scene_intersection intersect_scene_bvh(const Scene& S, const ray3f& ray_, bool find_any) {
  const auto& sc     = *S.sc;
  int         tree   = sc.num_shapes;  // the instance tree is the last one
  const auto* nodes  = S.bvh->nodes + S.bvh->node_offset[tree];
  const auto* prims  = S.bvh->primitives + S.bvh->prim_offset[tree];
  auto        nnodes = S.bvh->node_offset[tree + 1] - S.bvh->node_offset[tree];
  if (nnodes == 0) return {};  // :560

  int node_stack[128];
  int node_cur           = 0;
  node_stack[node_cur++] = 0;
  auto intersection      = scene_intersection{};
  auto ray               = ray_;
  auto ray_dinv          = vec3f{1 / ray.d.x, 1 / ray.d.y, 1 / ray.d.z};
  int  ray_dsign[3]      = {(ray_dinv.x < 0) ? 1 : 0, (ray_dinv.y < 0) ? 1 : 0, (ray_dinv.z < 0) ? 1 : 0};

  while (node_cur != 0) {
    const auto& node = nodes[node_stack[--node_cur]];
    if (!intersect_bbox(ray, ray_dinv, {node.bbox_min[0], node.bbox_min[1], node.bbox_min[2]},
            {node.bbox_max[0], node.bbox_max[1], node.bbox_max[2]}))
      continue;
    if (node.internal) {
      if (ray_dsign[node.axis] != 0) {
        node_stack[node_cur++] = node.start + 0;
        node_stack[node_cur++] = node.start + 1;
      } else {
        node_stack[node_cur++] = node.start + 1;
        node_stack[node_cur++] = node.start + 0;
      }
    } else {
      for (auto idx = node.start; idx < node.start + node.num; idx++) {
        const auto& instance_     = sc.instances[prims[idx]];
        auto        inv_ray       = transform_ray(inverse(ldframe(instance_.frame), true), ray);
        auto        sintersection = intersect_shape_bvh(S, instance_.shape, inv_ray, find_any);
        if (!sintersection.hit) continue;
        intersection = {prims[idx], sintersection.element, sintersection.uv, sintersection.distance, true};
        ray.tmax     = sintersection.distance;
      }
    }
    if (find_any && intersection.hit) return intersection;  // :613
  }
  return intersection;
}

// intersect_instance_bvh — yocto_bvh.cpp:619-628
scene_intersection intersect_instance_bvh(const Scene& S, int instance_, const ray3f& ray, bool find_any) {
  const auto& instance     = S.sc->instances[instance_];
  auto        inv_ray      = transform_ray(inverse(ldframe(instance.frame), true), ray);
  auto        intersection = intersect_shape_bvh(S, instance.shape, inv_ray, find_any);
  if (!intersection.hit) return {};
  return {instance_, intersection.element, intersection.uv, intersection.distance, true};
}

// --- camera, yocto_scene.cpp:66-101, yocto_trace.cpp:338-358 ------------------
ray3f eval_camera(const ythip_camera& camera, vec2f image_uv, vec2f lens_uv) {
  auto film  = camera.aspect >= 1 ? vec2f{camera.film, camera.film / camera.aspect}
                                 : vec2f{camera.film * camera.aspect, camera.film};
  auto frame = ldframe(camera.frame);
  if (!camera.orthographic) {
    auto q  = vec3f{film.x * (0.5f - image_uv.x), film.y * (image_uv.y - 0.5f), camera.lens};
    auto dc = -normalize(q);
    auto e  = vec3f{lens_uv.x * camera.aperture / 2, lens_uv.y * camera.aperture / 2, 0};
    auto p  = dc * camera.focus / abs_(dc.z);
    auto d  = normalize(p - e);
    return ray3f{transform_point(frame, e), transform_direction(frame, d)};
  } else {
    auto scale = 1 / camera.lens;
    auto q     = vec3f{film.x * (0.5f - image_uv.x) * scale, film.y * (image_uv.y - 0.5f) * scale, camera.lens};
    auto e     = vec3f{-q.x, -q.y, 0} + vec3f{lens_uv.x * camera.aperture / 2, lens_uv.y * camera.aperture / 2, 0};
    auto p     = vec3f{-q.x, -q.y, -camera.focus};
    auto d     = normalize(p - e);
    return ray3f{transform_point(frame, e), transform_direction(frame, d)};
  }
}
ray3f sample_camera(const ythip_camera& camera, int i, int j, int width, int height, vec2f puv, vec2f luv,
    bool tent) {
  if (!tent) {
    auto uv = vec2f{(i + puv.x) / width, (j + puv.y) / height};
    return eval_camera(camera, uv, sample_disk(luv));
  } else {
    const auto width_ = 2.0f;
    const auto offset = 0.5f;
    auto       fuv    = vec2f{
                 puv.x < 0.5f ? std::sqrt(2 * puv.x) - 1 : 1 - std::sqrt(2 - 2 * puv.x),
                 puv.y < 0.5f ? std::sqrt(2 * puv.y) - 1 : 1 - std::sqrt(2 - 2 * puv.y),
             } * width_ +
               vec2f{offset, offset};
    auto uv = vec2f{(i + fuv.x) / width, (j + fuv.y) / height};
    return eval_camera(camera, uv, sample_disk(luv));
  }
}

// --- shading-point evaluation, yocto_scene.cpp:288-528 ------------------------
vec3f eval_position(const ythip_scene& sc, const ythip_instance& instance, int element, vec2f uv) {  // :288
  const auto& shape = sc.shapes[instance.shape];
  auto        frame = ldframe(instance.frame);
  if (shape.num_triangles) {
    auto t = tri(sc, shape, element);
    return transform_point(frame, interpolate_triangle(pos(sc, shape, t[0]), pos(sc, shape, t[1]), pos(sc, shape, t[2]), uv));
  } else if (shape.num_quads) {
    auto q = quad(sc, shape, element);
    return transform_point(frame,
        interpolate_quad(pos(sc, shape, q[0]), pos(sc, shape, q[1]), pos(sc, shape, q[2]), pos(sc, shape, q[3]), uv));
  } else if (shape.num_lines) {
    auto l = line(sc, shape, element);
    return transform_point(frame, interpolate_line(pos(sc, shape, l[0]), pos(sc, shape, l[1]), uv.x));
  } else if (shape.num_points) {
    return transform_point(frame, pos(sc, shape, point(sc, shape, element)));
  } else {
    return {0, 0, 0};
  }
}
vec3f eval_element_normal(const ythip_scene& sc, const ythip_instance& instance, int element) {  // :314
  const auto& shape = sc.shapes[instance.shape];
  auto        frame = ldframe(instance.frame);
  if (shape.num_triangles) {
    auto t = tri(sc, shape, element);
    return transform_normal(frame, triangle_normal(pos(sc, shape, t[0]), pos(sc, shape, t[1]), pos(sc, shape, t[2])));
  } else if (shape.num_quads) {
    auto q = quad(sc, shape, element);
    return transform_normal(
        frame, quad_normal(pos(sc, shape, q[0]), pos(sc, shape, q[1]), pos(sc, shape, q[2]), pos(sc, shape, q[3])));
  } else if (shape.num_lines) {
    auto l = line(sc, shape, element);
    return transform_normal(frame, line_tangent(pos(sc, shape, l[0]), pos(sc, shape, l[1])));
  } else if (shape.num_points) {
    return {0, 0, 1};
  } else {
    return {0, 0, 0};
  }
}
vec3f eval_normal(const ythip_scene& sc, const ythip_instance& instance, int element, vec2f uv) {  // :339
  const auto& shape = sc.shapes[instance.shape];
  auto        frame = ldframe(instance.frame);
  if (shape.num_normals == 0) return eval_element_normal(sc, instance, element);
  if (shape.num_triangles) {
    auto t = tri(sc, shape, element);
    return transform_normal(
        frame, normalize(interpolate_triangle(nrm(sc, shape, t[0]), nrm(sc, shape, t[1]), nrm(sc, shape, t[2]), uv)));
  } else if (shape.num_quads) {
    auto q = quad(sc, shape, element);
    return transform_normal(frame,
        normalize(interpolate_quad(nrm(sc, shape, q[0]), nrm(sc, shape, q[1]), nrm(sc, shape, q[2]), nrm(sc, shape, q[3]), uv)));
  } else if (shape.num_lines) {
    auto l = line(sc, shape, element);
    return transform_normal(frame, normalize(interpolate_line(nrm(sc, shape, l[0]), nrm(sc, shape, l[1]), uv.x)));
  } else if (shape.num_points) {
    return transform_normal(frame, normalize(nrm(sc, shape, point(sc, shape, element))));
  } else {
    return {0, 0, 0};
  }
}
vec4f eval_color(const ythip_scene& sc, const ythip_instance& instance, int element, vec2f uv) {  // :508
  const auto& shape = sc.shapes[instance.shape];
  if (shape.num_colors == 0) return {1, 1, 1, 1};
  if (shape.num_triangles) {
    auto t = tri(sc, shape, element);
    return interpolate_triangle(col(sc, shape, t[0]), col(sc, shape, t[1]), col(sc, shape, t[2]), uv);
  } else if (shape.num_quads) {
    auto q = quad(sc, shape, element);
    return interpolate_quad(col(sc, shape, q[0]), col(sc, shape, q[1]), col(sc, shape, q[2]), col(sc, shape, q[3]), uv);
  } else if (shape.num_lines) {
    auto l = line(sc, shape, element);
    return interpolate_line(col(sc, shape, l[0]), col(sc, shape, l[1]), uv.x);
  } else if (shape.num_points) {
    return col(sc, shape, point(sc, shape, element));
  } else {
    return {0, 0, 0, 0};
  }
}
vec3f eval_shading_position(const ythip_scene& sc, const scene_intersection& isec, vec3f outgoing) {  // :468
  const auto& instance = sc.instances[isec.instance];
  const auto& shape    = sc.shapes[instance.shape];
  if (shape.num_triangles || shape.num_quads) {
    return eval_position(sc, instance, isec.element, isec.uv);
  } else if (shape.num_lines) {
    return eval_position(sc, instance, isec.element, isec.uv);
  } else if (shape.num_points) {
    return pos(sc, shape, point(sc, shape, isec.element));  // eval_position(shape, ...): NO instance transform (:477)
  } else {
    return {0, 0, 0};
  }
}
vec3f eval_shading_normal(const ythip_scene& sc, const scene_intersection& isec, vec3f outgoing) {  // :485
  const auto& instance = sc.instances[isec.instance];
  const auto& shape    = sc.shapes[instance.shape];
  const auto& material = sc.materials[instance.material];
  if (shape.num_triangles || shape.num_quads) {
    auto normal = eval_normal(sc, instance, isec.element, isec.uv);
    // normal maps are outside this restatement's scope (checked by supported())
    if (material.type == YTHIP_REFRACTIVE) return normal;
    return dot(normal, outgoing) >= 0 ? normal : -normal;
  } else if (shape.num_lines) {
    auto normal = eval_normal(sc, instance, isec.element, isec.uv);
    return orthonormalize(outgoing, normal);
  } else if (shape.num_points) {
    return outgoing;
  } else {
    return {0, 0, 0};
  }
}

// material_point, yocto_scene.h:258-270 (the fields the matte path reads)
struct material_point {
  int   type      = YTHIP_MATTE;
  vec3f emission  = {0, 0, 0};
  vec3f color     = {0, 0, 0};
  float opacity   = 1;
  float roughness = 0;
  float metallic  = 0;
  float ior       = 1;
  vec3f density   = {0, 0, 0};
};
constexpr float min_roughness = 0.03f * 0.03f;  // yocto_scene.cpp:200
// eval_material — yocto_scene.cpp:531-581 with every eval_texture(...) == {1,1,1,1}
// (texture id == invalidid, yocto_scene.cpp:169)
material_point eval_material(const ythip_scene& sc, const scene_intersection& isec) {
  const auto& instance       = sc.instances[isec.instance];
  const auto& material       = sc.materials[instance.material];
  auto        emission_tex   = vec4f{1, 1, 1, 1};
  auto        color_shp      = eval_color(sc, instance, isec.element, isec.uv);
  auto        color_tex      = vec4f{1, 1, 1, 1};
  auto        roughness_tex  = vec4f{1, 1, 1, 1};
  auto        point          = material_point{};
  point.type                 = material.type;
  point.emission = vec3f{material.emission[0], material.emission[1], material.emission[2]} *
                   vec3f{emission_tex.x, emission_tex.y, emission_tex.z} * vec3f{color_shp.x, color_shp.y, color_shp.z};
  point.color = vec3f{material.color[0], material.color[1], material.color[2]} *
                vec3f{color_tex.x, color_tex.y, color_tex.z} * vec3f{color_shp.x, color_shp.y, color_shp.z};
  point.opacity   = material.opacity * color_tex.w * color_shp.w;
  point.metallic  = material.metallic * roughness_tex.z;
  point.roughness = material.roughness * roughness_tex.y;
  point.roughness = point.roughness * point.roughness;
  point.ior       = material.ior;
  point.density   = {0, 0, 0};  // matte: not refractive / volumetric / subsurface
  // fix roughness: matte / gltfpbr / glossy branch
  point.roughness = clamp_(point.roughness, min_roughness, 1.0f);
  return point;
}
inline bool is_delta(const material_point&) { return false; }  // matte, yocto_scene.cpp:265-273

// eval_environment — yocto_scene.cpp:596-613 with emission_tex == invalidid.
// The lat-long texcoord is computed and dropped in the reference (the texture
// lookup returns {1,1,1,1}); it has no side effects, so it is omitted here.
vec3f eval_environment(const ythip_scene& sc, vec3f direction) {
  auto emission = vec3f{0, 0, 0};
  for (int k = 0; k < sc.num_environments; k++) {
    const auto& env = sc.environments[k];
    emission += vec3f{env.emission[0], env.emission[1], env.emission[2]} * vec3f{1, 1, 1};
  }
  return emission;
}

// --- matte lobe + emission, yocto_shading.h:554-573, yocto_trace.cpp:166-169 --
inline vec3f eval_emission(const material_point& material, vec3f normal, vec3f outgoing) {
  return dot(normal, outgoing) >= 0 ? material.emission : vec3f{0, 0, 0};
}
inline vec3f eval_matte(vec3f color, vec3f normal, vec3f outgoing, vec3f incoming) {
  if (dot(normal, incoming) * dot(normal, outgoing) <= 0) return {0, 0, 0};
  return color / pif * abs_(dot(normal, incoming));
}
inline vec3f sample_matte(vec3f color, vec3f normal, vec3f outgoing, vec2f rn) {
  auto up_normal = dot(normal, outgoing) <= 0 ? -normal : normal;
  return sample_hemisphere_cos(up_normal, rn);
}
inline float sample_matte_pdf(vec3f color, vec3f normal, vec3f outgoing, vec3f incoming) {
  if (dot(normal, incoming) * dot(normal, outgoing) <= 0) return 0;
  auto up_normal = dot(normal, outgoing) <= 0 ? -normal : normal;
  return sample_hemisphere_cos_pdf(up_normal, incoming);
}
// yocto_trace.cpp:172-300, matte branch only
inline vec3f eval_bsdfcos(const material_point& m, vec3f normal, vec3f outgoing, vec3f incoming) {
  if (m.roughness == 0) return {0, 0, 0};
  return eval_matte(m.color, normal, outgoing, incoming);
}
inline vec3f sample_bsdfcos(const material_point& m, vec3f normal, vec3f outgoing, float rnl, vec2f rn) {
  if (m.roughness == 0) return {0, 0, 0};
  return sample_matte(m.color, normal, outgoing, rn);
}
inline float sample_bsdfcos_pdf(const material_point& m, vec3f normal, vec3f outgoing, vec3f incoming) {
  if (m.roughness == 0) return 0;
  return sample_matte_pdf(m.color, normal, outgoing, incoming);
}

// --- lights, yocto_trace.cpp:361-443 -------------------------------------------
vec3f sample_lights(const Scene& S, vec3f position, float rl, float rel, vec2f ruv) {
  const auto& sc       = *S.sc;
  auto        light_id = sample_uniform(S.lights->num_lights, rl);
  const auto& light    = S.lights->lights[light_id];
  if (light.instance != YTHIP_INVALIDID) {
    const auto& instance  = sc.instances[light.instance];
    const auto& shape     = sc.shapes[instance.shape];
    auto        element   = sample_discrete(S.lights->cdf + light.cdf_offset, light.cdf_count, rel);
    auto        uv        = shape.num_triangles ? sample_triangle(ruv) : ruv;
    auto        lposition = eval_position(sc, instance, element, uv);
    return normalize(lposition - position);
  } else if (light.environment != YTHIP_INVALIDID) {
    return sample_sphere(ruv);  // emission_tex == invalidid
  } else {
    return {0, 0, 0};
  }
}
float sample_lights_pdf(const Scene& S, vec3f position, vec3f direction) {
  const auto& sc  = *S.sc;
  auto        pdf = 0.0f;
  for (int l = 0; l < S.lights->num_lights; l++) {
    const auto& light = S.lights->lights[l];
    if (light.instance != YTHIP_INVALIDID) {
      const auto& instance      = sc.instances[light.instance];
      auto        lpdf          = 0.0f;
      auto        next_position = position;
      for (auto bounce = 0; bounce < 100; bounce++) {
        auto intersection = intersect_instance_bvh(S, light.instance, ray3f{next_position, direction}, false);
        if (!intersection.hit) break;
        auto lposition = eval_position(sc, instance, intersection.element, intersection.uv);
        auto lnormal   = eval_element_normal(sc, instance, intersection.element);
        auto area      = S.lights->cdf[light.cdf_offset + light.cdf_count - 1];
        lpdf += distance_squared(lposition, position) / (abs_(dot(lnormal, direction)) * area);
        next_position = lposition + direction * 1e-3f;
      }
      pdf += lpdf;
    } else if (light.environment != YTHIP_INVALIDID) {
      pdf += 1 / (4 * pif);
    }
  }
  pdf *= sample_uniform_pdf(S.lights->num_lights);
  return pdf;
}

// --- integrators ----------------------------------------------------------------
struct trace_result {
  vec3f radiance = {0, 0, 0};
  bool  hit      = false;
  vec3f albedo   = {0, 0, 0};
  vec3f normal   = {0, 0, 0};
};

// trace_path — yocto_trace.cpp:453-596.  The volume stack stays empty for
// matte scenes (is_volumetric false), so the volume branch is not restated.
trace_result trace_path(const Scene& S, const ray3f& ray_, rng_state& rng, const ythip_params& params) {
  const auto& sc            = *S.sc;
  auto        radiance      = vec3f{0, 0, 0};
  auto        weight        = vec3f{1, 1, 1};
  auto        ray           = ray_;
  auto        max_roughness = 0.0f;
  auto        hit           = false;
  auto        hit_albedo    = vec3f{0, 0, 0};
  auto        hit_normal    = vec3f{0, 0, 0};
  auto        opbounce      = 0;
  for (auto bounce = 0; bounce < params.bounces; bounce++) {
    auto intersection = intersect_scene_bvh(S, ray, false);
    if (!intersection.hit) {
      if (bounce > 0 || !params.envhidden) radiance += weight * eval_environment(sc, ray.d);
      break;
    }
    auto outgoing = -ray.d;
    auto position = eval_shading_position(sc, intersection, outgoing);
    auto normal   = eval_shading_normal(sc, intersection, outgoing);
    auto material = eval_material(sc, intersection);
    if (params.nocaustics) {
      max_roughness      = max_(material.roughness, max_roughness);
      material.roughness = max_roughness;
    }
    if (material.opacity < 1 && rand1f(rng) >= material.opacity) {
      if (opbounce++ > 128) break;
      ray = {position + ray.d * 1e-2f, ray.d};
      bounce -= 1;
      continue;
    }
    if (bounce == 0) {
      hit        = true;
      hit_albedo = material.color;
      hit_normal = normal;
    }
    radiance += weight * eval_emission(material, normal, outgoing);
    auto incoming = vec3f{0, 0, 0};
    if (!is_delta(material)) {
      if (rand1f(rng) < 0.5f) {
        // sample_bsdfcos(material, normal, outgoing, rand1f(rng), rand2f(rng)): g++ draws rn, then rnl
        auto rn  = rand2f(rng);
        auto rnl = rand1f(rng);
        incoming = sample_bsdfcos(material, normal, outgoing, rnl, rn);
      } else {
        // sample_lights(..., rand1f, rand1f, rand2f): g++ draws ruv, rel, rl
        auto ruv = rand2f(rng);
        auto rel = rand1f(rng);
        auto rl  = rand1f(rng);
        incoming = sample_lights(S, position, rl, rel, ruv);
      }
      if (incoming == vec3f{0, 0, 0}) break;
      weight *= eval_bsdfcos(material, normal, outgoing, incoming) /
                (0.5f * sample_bsdfcos_pdf(material, normal, outgoing, incoming) +
                    0.5f * sample_lights_pdf(S, position, incoming));
    }
    ray = {position, incoming};
    if (weight == vec3f{0, 0, 0} || !isfinite3(weight)) break;
    if (bounce > 3) {
      auto rr_prob = min_((float)0.99, max3(weight));
      if (rand1f(rng) >= rr_prob) break;
      weight *= 1 / rr_prob;
    }
  }
  return {radiance, hit, hit_albedo, hit_normal};
}

// trace_naive — yocto_trace.cpp:1032-1108
trace_result trace_naive(const Scene& S, const ray3f& ray_, rng_state& rng, const ythip_params& params) {
  const auto& sc         = *S.sc;
  auto        radiance   = vec3f{0, 0, 0};
  auto        weight     = vec3f{1, 1, 1};
  auto        ray        = ray_;
  auto        hit        = false;
  auto        hit_albedo = vec3f{0, 0, 0};
  auto        hit_normal = vec3f{0, 0, 0};
  auto        opbounce   = 0;
  for (auto bounce = 0; bounce < params.bounces; bounce++) {
    auto intersection = intersect_scene_bvh(S, ray, false);
    if (!intersection.hit) {
      if (bounce > 0 || !params.envhidden) radiance += weight * eval_environment(sc, ray.d);
      break;
    }
    auto outgoing = -ray.d;
    auto position = eval_shading_position(sc, intersection, outgoing);
    auto normal   = eval_shading_normal(sc, intersection, outgoing);
    auto material = eval_material(sc, intersection);
    if (material.opacity < 1 && rand1f(rng) >= material.opacity) {
      if (opbounce++ > 128) break;
      ray = {position + ray.d * 1e-2f, ray.d};
      bounce -= 1;
      continue;
    }
    if (bounce == 0) {
      hit        = true;
      hit_albedo = material.color;
      hit_normal = normal;
    }
    radiance += weight * eval_emission(material, normal, outgoing);
    auto incoming = vec3f{0, 0, 0};
    if (material.roughness != 0) {
      auto rn  = rand2f(rng);
      auto rnl = rand1f(rng);
      incoming = sample_bsdfcos(material, normal, outgoing, rnl, rn);
      if (incoming == vec3f{0, 0, 0}) break;
      weight *= eval_bsdfcos(material, normal, outgoing, incoming) /
                sample_bsdfcos_pdf(material, normal, outgoing, incoming);
    } else {
      break;  // sample_delta of a matte material returns {0,0,0} (yocto_trace.cpp:246) after one draw
    }
    if (weight == vec3f{0, 0, 0} || !isfinite3(weight)) break;
    if (bounce > 3) {
      auto rr_prob = min_((float)0.99, max3(weight));
      if (rand1f(rng) >= rr_prob) break;
      weight *= 1 / rr_prob;
    }
    ray = {position, incoming};
  }
  return {radiance, hit, hit_albedo, hit_normal};
}

// trace_eyelight — yocto_trace.cpp:1111-1176
trace_result trace_eyelight(const Scene& S, const ray3f& ray_, rng_state& rng, const ythip_params& params) {
  const auto& sc         = *S.sc;
  auto        radiance   = vec3f{0, 0, 0};
  auto        weight     = vec3f{1, 1, 1};
  auto        ray        = ray_;
  auto        hit        = false;
  auto        hit_albedo = vec3f{0, 0, 0};
  auto        hit_normal = vec3f{0, 0, 0};
  auto        opbounce   = 0;
  for (auto bounce = 0; bounce < std::max(params.bounces, 4); bounce++) {
    auto intersection = intersect_scene_bvh(S, ray, false);
    if (!intersection.hit) {
      if (bounce > 0 || !params.envhidden) radiance += weight * eval_environment(sc, ray.d);
      break;
    }
    auto outgoing = -ray.d;
    auto position = eval_shading_position(sc, intersection, outgoing);
    auto normal   = eval_shading_normal(sc, intersection, outgoing);
    auto material = eval_material(sc, intersection);
    if (material.opacity < 1 && rand1f(rng) >= material.opacity) {
      if (opbounce++ > 128) break;
      ray = {position + ray.d * 1e-2f, ray.d};
      bounce -= 1;
      continue;
    }
    if (bounce == 0) {
      hit        = true;
      hit_albedo = material.color;
      hit_normal = normal;
    }
    auto incoming = outgoing;
    radiance += weight * eval_emission(material, normal, outgoing);
    radiance += weight * pif * eval_bsdfcos(material, normal, outgoing, incoming);
    if (!is_delta(material)) break;
  }
  return {radiance, hit, hit_albedo, hit_normal};
}

// what this restatement covers; anything else must go to the compiled reference
const char* unsupported(const ythip_scene& sc, const ythip_lights* lights, const ythip_params& p) {
  if (p.sampler != YTHIP_SAMPLER_PATH && p.sampler != YTHIP_SAMPLER_NAIVE && p.sampler != YTHIP_SAMPLER_EYELIGHT)
    return "sampler not restated (path, naive, eyelight only)";
  if (p.camera < 0 || p.camera >= sc.num_cameras) return "camera out of range";
  for (int k = 0; k < sc.num_materials; k++) {
    const auto& m = sc.materials[k];
    if (m.type != YTHIP_MATTE) return "material type not restated (matte only)";
    if (m.emission_tex != YTHIP_INVALIDID || m.color_tex != YTHIP_INVALIDID || m.roughness_tex != YTHIP_INVALIDID ||
        m.scattering_tex != YTHIP_INVALIDID || m.normal_tex != YTHIP_INVALIDID)
      return "textures not restated";
  }
  for (int k = 0; k < sc.num_environments; k++)
    if (sc.environments[k].emission_tex != YTHIP_INVALIDID) return "environment maps not restated";
  if (p.sampler == YTHIP_SAMPLER_PATH && (!lights || lights->num_lights <= 0))
    return "trace_path needs at least one light (the reference reads lights[0] out of bounds)";
  return nullptr;
}

// trace_sample — yocto_trace.cpp:1461-1492
void trace_sample(const Scene& S, const ythip_params& params, int width, int height, int i, int j, int sample,
    vec4f* image, vec3f* albedo_, vec3f* normal_, int32_t* hits, rng_state* rngs) {
  const auto& sc     = *S.sc;
  const auto& camera = sc.cameras[params.camera];
  auto        idx    = width * j + i;
  // sample_camera(camera, ij, size, rand2f(rng), rand2f(rng), tent): g++ evaluates
  // the LAST argument first → luv takes draws 1-2, puv draws 3-4
  auto luv = rand2f(rngs[idx]);
  auto puv = rand2f(rngs[idx]);
  auto ray = sample_camera(camera, i, j, width, height, puv, luv, params.tentfilter != 0);
  trace_result r;
  switch (params.sampler) {
    case YTHIP_SAMPLER_PATH: r = trace_path(S, ray, rngs[idx], params); break;
    case YTHIP_SAMPLER_NAIVE: r = trace_naive(S, ray, rngs[idx], params); break;
    default: r = trace_eyelight(S, ray, rngs[idx], params); break;
  }
  auto radiance = r.radiance;
  if (!isfinite3(radiance)) radiance = {0, 0, 0};
  if (max3(radiance) > params.clamp) radiance = radiance * (params.clamp / max3(radiance));
  auto weight = 1.0f / (sample + 1);
  if (r.hit) {
    image[idx]   = lerp(image[idx], vec4f{radiance.x, radiance.y, radiance.z, 1}, weight);
    albedo_[idx] = lerp(albedo_[idx], r.albedo, weight);
    normal_[idx] = lerp(normal_[idx], r.normal, weight);
    hits[idx] += 1;
  } else if (!params.envhidden && sc.num_environments != 0) {
    image[idx]   = lerp(image[idx], vec4f{radiance.x, radiance.y, radiance.z, 1}, weight);
    albedo_[idx] = lerp(albedo_[idx], vec3f{1, 1, 1}, weight);
    normal_[idx] = lerp(normal_[idx], -ray.d, weight);
    hits[idx] += 1;
  } else {
    image[idx]   = lerp(image[idx], vec4f{0, 0, 0, 0}, weight);
    albedo_[idx] = lerp(albedo_[idx], vec3f{0, 0, 0}, weight);
    normal_[idx] = lerp(normal_[idx], -ray.d, weight);
  }
}

}  // namespace

// ===========================================================================
// C ABI (ctypes: oracle/ytoracle.py)
// ===========================================================================
extern "C" {

const char* yto_last_error() { return g_err.c_str(); }

// make_trace_state seeding — yocto_trace.cpp:1512-1515
int yto_make_rngs(uint64_t seed, int64_t n, uint64_t* out) {
  auto rng_ = make_rng(1301081);
  for (int64_t k = 0; k < n; k++) {
    auto rng       = make_rng(seed, rand1i(rng_, 1 << 31) / 2 + 1);
    out[2 * k]     = rng.state;
    out[2 * k + 1] = rng.inc;
  }
  return 0;
}
// make_rng + n rand1f draws (known-answer tests)
int yto_rand1f(uint64_t seed, uint64_t seq, int n, uint64_t* state_out, float* out) {
  auto rng     = make_rng(seed, seq);
  state_out[0] = rng.state, state_out[1] = rng.inc;
  for (int k = 0; k < n; k++) out[k] = rand1f(rng);
  return 0;
}
// make_trace_state size rule — yocto_trace.cpp:1499-1505
int yto_state_size(const ythip_camera* camera, int resolution, int* width, int* height) {
  if (camera->aspect >= 1) {
    *width  = resolution;
    *height = (int)std::round(resolution / camera->aspect);
  } else {
    *height = resolution;
    *width  = (int)std::round(resolution * camera->aspect);
  }
  return 0;
}
// The camera ray trace_sample would generate next for every pixel (does not
// advance the caller's rngs).
int yto_camera_rays(const ythip_scene* sc, const ythip_params* params, int width, int height, const uint64_t* rngs,
    ythip_ray* rays) {
  if (params->camera < 0 || params->camera >= sc->num_cameras) return fail("camera out of range");
  for (int j = 0; j < height; j++)
    for (int i = 0; i < width; i++) {
      auto      idx = width * j + i;
      rng_state rng = {rngs[2 * idx], rngs[2 * idx + 1]};
      auto      luv = rand2f(rng);
      auto      puv = rand2f(rng);
      auto ray = sample_camera(sc->cameras[params->camera], i, j, width, height, puv, luv, params->tentfilter != 0);
      rays[idx] = {{ray.o.x, ray.o.y, ray.o.z}, {ray.d.x, ray.d.y, ray.d.z}, ray.tmin, ray.tmax};
    }
  return 0;
}
static ythip_hit to_hit(const scene_intersection& h) {
  return {h.instance, h.element, h.uv.x, h.uv.y, h.distance, h.hit ? 1 : 0};
}
// intersect_scene_bvh on a ray batch
int yto_intersect_batch(const ythip_scene* sc, const ythip_bvh* bvh, const ythip_ray* rays, int64_t n, int find_any,
    ythip_hit* hits) {
  if (bvh->num_trees != sc->num_shapes + 1) return fail("bvh / scene tree count mismatch");
  Scene S = {sc, bvh, nullptr};
  for (int64_t k = 0; k < n; k++) {
    const auto& r = rays[k];
    ray3f ray     = {{r.o[0], r.o[1], r.o[2]}, {r.d[0], r.d[1], r.d[2]}, r.tmin, r.tmax};
    hits[k]       = to_hit(intersect_scene_bvh(S, ray, find_any != 0));
  }
  return 0;
}
// intersect_instance_bvh on a ray batch
int yto_intersect_instance_batch(const ythip_scene* sc, const ythip_bvh* bvh, const int32_t* instances,
    const ythip_ray* rays, int64_t n, int find_any, ythip_hit* hits) {
  if (bvh->num_trees != sc->num_shapes + 1) return fail("bvh / scene tree count mismatch");
  Scene S = {sc, bvh, nullptr};
  for (int64_t k = 0; k < n; k++) {
    if (instances[k] < 0 || instances[k] >= sc->num_instances) return fail("instance out of range");
    const auto& r = rays[k];
    ray3f ray     = {{r.o[0], r.o[1], r.o[2]}, {r.d[0], r.d[1], r.d[2]}, r.tmin, r.tmax};
    hits[k]       = to_hit(intersect_instance_bvh(S, instances[k], ray, find_any != 0));
  }
  return 0;
}
// Returns 0 when yto_trace_samples restates this scene/params combination, else 1
// (reason in yto_last_error).
int yto_supported(const ythip_scene* sc, const ythip_lights* lights, const ythip_params* params) {
  auto why = unsupported(*sc, lights, *params);
  return why ? fail(why) : 0;
}
// trace_samples — yocto_trace.cpp:1595-1619 (the noparallel branch: results do
// not depend on the thread count).  State arrays are the caller's
// (width*height pixels), `samples` is state.samples in and out.
int yto_trace_samples(const ythip_scene* sc, const ythip_bvh* bvh, const ythip_lights* lights,
    const ythip_params* params, int width, int height, float* image, float* albedo, float* normal, int32_t* hits,
    uint64_t* rngs, int* samples) {
  if (auto why = unsupported(*sc, lights, *params)) return fail(why);
  if (bvh->num_trees != sc->num_shapes + 1) return fail("bvh / scene tree count mismatch");
  if (*samples >= params->samples) return 0;  // :1598
  Scene S = {sc, bvh, lights};
  for (int j = 0; j < height; j++)
    for (int i = 0; i < width; i++)
      for (int sample = *samples; sample < *samples + params->batch; sample++)
        trace_sample(S, *params, width, height, i, j, sample, (vec4f*)image, (vec3f*)albedo, (vec3f*)normal, hits,
            (rng_state*)rngs);
  *samples += params->batch;
  return 0;
}

}  // extern "C"
