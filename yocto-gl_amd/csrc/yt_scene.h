// yt_scene.h — device-resident scene view and the eval_* subset of
// libs/yocto/yocto_scene.cpp used by trace_samples (camera, texture, material,
// position/normal/texcoord/color interpolation, environment).
#pragma once

#include "../../include/ythip.h"
#include "yt_material.h"
#include "yt_math.h"

#ifdef YT_FAST  // tolerance mode: multiply-adds written in this file may fuse (the traversal's, in yt_bvh.h, never do)
#pragma clang fp contract(fast)
#endif
namespace yt {

// element kinds, in the reference's two dispatch orders
enum { KIND_NONE = 0, KIND_POINTS = 1, KIND_LINES = 2, KIND_TRIANGLES = 3, KIND_QUADS = 4 };

// Per-shape record (device).  Offsets index the shared pools.
struct DShape {
  int kind_bvh;   // points→lines→triangles→quads  (yocto_bvh.cpp:505-545)
  int kind_eval;  // triangles→quads→lines→points  (yocto_scene.cpp:288-311)
  int elem_bvh;   // element pool offset for kind_bvh
  int elem_eval;  // element pool offset for kind_eval
  int positions, normals, texcoords, colors, radius;  // vertex pool offsets, -1 absent
  int pad_[3];
};

// Per-instance record for traversal (96 B, one fetch per TLAS leaf entry):
// inverse(frame, non_rigid=true) computed on the host with the reference's
// operation order (yocto_math.h:2114-2118), the bbox and ref of the shape's BLAS
// root (so entering an instance needs no dependent node fetch), the shape's
// element kind and where its leaf data lives.
struct DInstanceT {
  float inv[12];
  float root_bmin[3], root_bmax[3];
  int   root_ref;   // ref of the BLAS root (yt_bvh.h), REF_NONE if the tree is empty
  int   kind;       // kind_bvh of the shape
  int   leaf_bias;  // float4 index: leafdata of global primitive p starts at leaf_bias + p * stride
  int   shape;
  int   instance;   // tinst_leaf only: the instance this copy belongs to
  int   pad_;
};
static_assert(sizeof(DInstanceT) == 96, "DInstanceT is fetched as 6 float4");

struct DLight {
  int instance, environment, cdf_offset, cdf_count;
};

struct DScene {
  // scene_data
  const ythip_camera*      cameras;
  const ythip_instance*    instances;
  const ythip_environment* environments;
  const float*             env_inv;  // 12 floats per environment: inverse(frame) rigid
  const ythip_material*    materials;
  const ythip_texture*     textures;
  const DShape*            shapes;
  int num_instances, num_environments, num_shapes, num_materials, num_textures;
  // pools
  const int*   points;
  const int*   lines;
  const int*   triangles;
  const int*   quads;
  const float* positions;
  const float* normals;
  const float* texcoords;
  const float* colors;
  const float* radius;
  const float*   pixelsf;
  const uint8_t* pixelsb;
  const float*   srgb_lut;  // srgb_to_rgb(b / 255.0f) for b = 0 ... 255, filled on the device by the same function (ythip.hip)
  // bvh
  const float4*     pairs;      // sibling-pair records (4 float4 each), all trees
  const float4*     wide;       // grandchildren ("quad") records (8 float4 each), same ids as `pairs`
  const float4*     leafdata;   // pre-gathered leaf primitives in leaf order
  const int*        tlas_prims; // instance ids in TLAS leaf order
  const DInstanceT* tinst;      // per instance
  const DInstanceT* tinst_leaf; // the same records gathered in TLAS-leaf order (tinst[tlas_prims[k]], with .instance set)
  int               tlas_ref;   // ref of the TLAS root, REF_NONE if empty
  vec3f             tlas_bmin, tlas_bmax;
  const uint4*      own;        // fastmath = 2 only: compressed 64-B nodes of the own tree (yt_own.h), same ids as `wide`; null otherwise
  // lights
  const DLight* lights;
  const float*  cdf;
  int           num_lights;
};

// Scalar-cache reads of records every lane asks for alike (round 4; the rationale is at wave_uniform in yt_bvh.h).
constexpr bool SCALAR_LOADS = true;
// a POD record through the scalar cache, dword by dword (the compiler merges the loads): `p` wave-uniform, immutable data
template <typename T>
YT_FN T ldc_record(const T* p) {
  static_assert(sizeof(T) % 4 == 0, "dword records");
  typedef __attribute__((address_space(4))) const int cint;
  int   w[sizeof(T) / 4];
  cint* q = (cint*)p;
#pragma unroll
  for (int k = 0; k < (int)(sizeof(T) / 4); k++) w[k] = q[k];
  T v;
  __builtin_memcpy(&v, w, sizeof(T));
  return v;
}
// wave_uniform(x, u): u = x of the first active lane; true when all active lanes agree (the rationale is at ldc4 in yt_bvh.h)
YT_FN bool wave_uniform(int x, int& u) {
  u = __builtin_amdgcn_readfirstlane(x);
  return __ballot(x != u) == 0ull;
}
// -DYT_RECORDS_BY_VALUE (measured lead for round 5; off: the shipped code is unchanged).  A small POD record read
// through a reference is fetched field by field where each field is first used — hipcc narrows the loads to the words used and
// sinks them behind the branches — so a record whose fields are tested one after the other (a texture's width, height, clamp,
// nearest, is_float; a light's instance, then its cdf range) costs one dependent L1 round trip PER FIELD.  ld_record: the record by
// value, every word fetched at once (the empty asm keeps the loads whole and where they are); load_record: through the scalar
// cache instead when the wavefront agrees on the index.
template <typename T>
YT_FN T ld_record(const T* p) {
  static_assert(sizeof(T) % 4 == 0, "dword records");
  constexpr int N = (int)(sizeof(T) / 4);
  int           w[N];
  const int*    q = reinterpret_cast<const int*>(p);
#pragma unroll
  for (int k = 0; k < N; k++) w[k] = q[k];
#pragma unroll
  for (int k = 0; k < N; k++) asm volatile("" : "+v"(w[k]));
  T v;
  __builtin_memcpy(&v, w, sizeof(T));
  return v;
}
template <typename T>
YT_FN T load_record(const T* base, int idx) {
  if (int u; SCALAR_LOADS && wave_uniform(idx, u)) return ldc_record(base + u);
  return ld_record(base + idx);
}

YT_FN vec3f ld3(const float* p, int i) { return {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }
YT_FN vec2f ld2(const float* p, int i) { return {p[2 * i], p[2 * i + 1]}; }
YT_FN vec4f ld4(const float* p, int i) {
  auto v = reinterpret_cast<const float4*>(p)[i];
  return {v.x, v.y, v.z, v.w};
}
YT_FN frame3f ldframe(const float* f) {
  return {{f[0], f[1], f[2]}, {f[3], f[4], f[5]}, {f[6], f[7], f[8]}, {f[9], f[10], f[11]}};
}
YT_FN frame3f ldframe(const ythip_frame& f) { return ldframe(f.x); }

struct ray3f {
  vec3f o, d;
  float tmin, tmax;
};
YT_FN ray3f make_ray(vec3f o, vec3f d) { return {o, d, ray_eps, flt_max}; }

// ---------------------------------------------------------------------------
// eval_camera — yocto_scene.cpp:66-101
// ---------------------------------------------------------------------------
YT_FN ray3f eval_camera(const ythip_camera& camera, vec2f image_uv, vec2f lens_uv) {
  auto film = camera.aspect >= 1 ? vec2f{camera.film, div_(camera.film, camera.aspect)}
                                 : vec2f{camera.film * camera.aspect, camera.film};
  auto frame = ldframe(camera.frame);
  if (!camera.orthographic) {
    auto q  = vec3f{film.x * (0.5f - image_uv.x), film.y * (image_uv.y - 0.5f), camera.lens};
    auto dc = -normalize(q);
    auto e  = vec3f{div_(lens_uv.x * camera.aperture, 2), div_(lens_uv.y * camera.aperture, 2), 0};
    auto p  = dc * camera.focus / fabs_(dc.z);
    auto d  = normalize(p - e);
    return make_ray(transform_point(frame, e), transform_direction(frame, d));
  } else {
    auto scale = rcp_(camera.lens);
    auto q     = vec3f{film.x * (0.5f - image_uv.x) * scale,
        film.y * (image_uv.y - 0.5f) * scale, camera.lens};
    auto e     = vec3f{-q.x, -q.y, 0} +
             vec3f{div_(lens_uv.x * camera.aperture, 2), div_(lens_uv.y * camera.aperture, 2), 0};
    auto p = vec3f{-q.x, -q.y, -camera.focus};
    auto d = normalize(p - e);
    return make_ray(transform_point(frame, e), transform_direction(frame, d));
  }
}

// sample_camera — yocto_trace.cpp:338-358
YT_FN ray3f sample_camera(const ythip_camera& camera, int i, int j, int width, int height,
    vec2f puv, vec2f luv, bool tent) {
  if (!tent) {
    auto uv = vec2f{div_(i + puv.x, (float)width), div_(j + puv.y, (float)height)};
    return eval_camera(camera, uv, sample_disk(luv));
  } else {
    const auto width_ = 2.0f;
    const auto offset = 0.5f;
    auto       fuv    = vec2f{puv.x < 0.5f ? sqrt_(2 * puv.x) - 1 : 1 - sqrt_(2 - 2 * puv.x),
                     puv.y < 0.5f ? sqrt_(2 * puv.y) - 1 : 1 - sqrt_(2 - 2 * puv.y)};
    fuv = {width_ * fuv.x + offset, width_ * fuv.y + offset};
    auto uv = vec2f{div_(i + fuv.x, (float)width), div_(j + fuv.y, (float)height)};
    return eval_camera(camera, uv, sample_disk(luv));
  }
}

// ---------------------------------------------------------------------------
// textures — yocto_scene.cpp:111-178, yocto_color.h:223-249
// ---------------------------------------------------------------------------
YT_FN float srgb_to_rgb(float srgb) {  // yocto_color.h:235 (double-typed threshold)
  return ((double)srgb <= 0.04045) ? div_(srgb, 12.92f)
                                   : ytm::powf(div_(srgb + 0.055f, 1.0f + 0.055f), 2.4f);
}
YT_FN vec3f srgb_to_rgb(vec3f c) { return {srgb_to_rgb(c.x), srgb_to_rgb(c.y), srgb_to_rgb(c.z)}; }
// -DYT_SRGB_LUT, -DYT_TEXELS_TOGETHER (leads for round 5; off: unchanged).
//  * A byte texel has 256 possible values per channel, and srgb_to_rgb of each is a double-precision powf (two table loads, ~150
//    instructions): twelve of them per bilinear lookup of an sRGB texture.  With the LUT the decoded value comes from a 1-KB table
//    that a device kernel fills with this very function at scene upload — the same floats by construction.
//  * The four taps of a bilinear lookup are fetched one after the other (hipcc -S: load, s_waitcnt vmcnt(0), convert, next address):
//    four dependent round trips.  TEXELS_TOGETHER fetches the four raw texels first and converts afterwards; the weighted sum
//    keeps the reference's order.
struct RawTexel {
  float4   f;
  unsigned b;
};
YT_FN RawTexel fetch_texel(const DScene& sc, const ythip_texture& t, int i, int j) {
  RawTexel r   = {{0, 0, 0, 0}, 0};
  auto     idx = t.offset + (int64_t)j * t.width + i;
  if (t.is_float) r.f = reinterpret_cast<const float4*>(sc.pixelsf)[idx];
  else r.b = reinterpret_cast<const unsigned*>(sc.pixelsb)[idx];
  return r;
}
YT_FN vec4f decode_texel(const DScene& sc, const ythip_texture& t, const RawTexel& r, bool as_linear) {
  vec4f color;
  if (t.is_float) {
    color = {r.f.x, r.f.y, r.f.z, r.f.w};
  } else {
    const unsigned bx = r.b & 255u, by = (r.b >> 8) & 255u, bz = (r.b >> 16) & 255u, bw = r.b >> 24;
    if (as_linear && !t.linear) return {sc.srgb_lut[bx], sc.srgb_lut[by], sc.srgb_lut[bz], div_((float)bw, 255.0f)};
    color = {div_((float)bx, 255.0f), div_((float)by, 255.0f), div_((float)bz, 255.0f), div_((float)bw, 255.0f)};
  }
  if (as_linear && !t.linear) {
    return {srgb_to_rgb(color.x), srgb_to_rgb(color.y), srgb_to_rgb(color.z), color.w};
  }
  return color;
}
YT_FN vec4f lookup_texture(const DScene& sc, const ythip_texture& t, int i, int j, bool as_linear) {
  return decode_texel(sc, t, fetch_texel(sc, t, i, j), as_linear);
}
YT_FN vec4f eval_texture(const DScene& sc, int texture, vec2f uv, bool as_linear) {
  if (texture == YTHIP_INVALIDID) return {1, 1, 1, 1};
  const ythip_texture t = load_record(sc.textures, texture);
  if (t.width == 0 || t.height == 0) return {0, 0, 0, 0};
  auto sx = t.width, sy = t.height;
  auto s = 0.0f, tt = 0.0f;
  if (t.clamp) {
    s  = clamp_(uv.x, 0.0f, 1.0f) * sx;
    tt = clamp_(uv.y, 0.0f, 1.0f) * sy;
  } else {
    s = fmodf(uv.x, 1.0f) * sx;
    if (s < 0) s += sx;
    tt = fmodf(uv.y, 1.0f) * sy;
    if (tt < 0) tt += sy;
  }
  auto i = clamp_((int)s, 0, sx - 1), j = clamp_((int)tt, 0, sy - 1);
  auto ii = (i + 1) % sx, jj = (j + 1) % sy;
  auto u = s - i, v = tt - j;
  if (t.nearest) {
    return lookup_texture(sc, t, i, j, as_linear);
  } else {
    // (one branch on the texel type, then four independent fetches in a row; the empty asm keeps them ahead of the first conversion)
    const int64_t o00 = t.offset + (int64_t)j * t.width + i, o01 = t.offset + (int64_t)jj * t.width + i,
                  o10 = t.offset + (int64_t)j * t.width + ii, o11 = t.offset + (int64_t)jj * t.width + ii;
    RawTexel r00 = {{0, 0, 0, 0}, 0}, r01 = r00, r10 = r00, r11 = r00;
    if (t.is_float) {
      const float4* px = reinterpret_cast<const float4*>(sc.pixelsf);
      r00.f = px[o00], r01.f = px[o01], r10.f = px[o10], r11.f = px[o11];
      asm volatile("" : "+v"(r00.f.x), "+v"(r00.f.y), "+v"(r00.f.z), "+v"(r00.f.w), "+v"(r01.f.x), "+v"(r01.f.y), "+v"(r01.f.z), "+v"(r01.f.w),
                   "+v"(r10.f.x), "+v"(r10.f.y), "+v"(r10.f.z), "+v"(r10.f.w), "+v"(r11.f.x), "+v"(r11.f.y), "+v"(r11.f.z), "+v"(r11.f.w));
    } else {
      const unsigned* px = reinterpret_cast<const unsigned*>(sc.pixelsb);
      r00.b = px[o00], r01.b = px[o01], r10.b = px[o10], r11.b = px[o11];
      asm volatile("" : "+v"(r00.b), "+v"(r01.b), "+v"(r10.b), "+v"(r11.b));
    }
    return decode_texel(sc, t, r00, as_linear) * (1 - u) * (1 - v) + decode_texel(sc, t, r01, as_linear) * (1 - u) * v +
           decode_texel(sc, t, r10, as_linear) * u * (1 - v) + decode_texel(sc, t, r11, as_linear) * u * v;
  }
}

// ---------------------------------------------------------------------------
// interpolation — yocto_geometry.h:535-556
// ---------------------------------------------------------------------------
template <typename T>
YT_FN T interpolate_line(T p0, T p1, float u) {
  return p0 * (1 - u) + p1 * u;
}
template <typename T>
YT_FN T interpolate_triangle(T p0, T p1, T p2, vec2f uv) {
  return p0 * (1 - uv.x - uv.y) + p1 * uv.x + p2 * uv.y;
}
template <typename T>
YT_FN T interpolate_quad(T p0, T p1, T p2, T p3, vec2f uv) {
  if (uv.x + uv.y <= 1) {
    return interpolate_triangle(p0, p1, p3, uv);
  } else {
    return interpolate_triangle(p2, p3, p1, 1 - uv);
  }
}
YT_FN vec3f triangle_normal(vec3f p0, vec3f p1, vec3f p2) {  // yocto_geometry.h:512
  return normalize(cross(p1 - p0, p2 - p0));
}
YT_FN vec3f quad_normal(vec3f p0, vec3f p1, vec3f p2, vec3f p3) {  // :522
  return normalize(triangle_normal(p0, p1, p3) + triangle_normal(p2, p3, p1));
}

// ---------------------------------------------------------------------------
// instance properties — yocto_scene.cpp:288-528
// ---------------------------------------------------------------------------
struct elem4 {
  int x, y, z, w;
};
YT_FN elem4 load_element(const DScene& sc, const DShape& sh, int element) {
  switch (sh.kind_eval) {
    case KIND_TRIANGLES: {
      auto p = sc.triangles + 3 * (sh.elem_eval + (int64_t)element);
      return {p[0], p[1], p[2], 0};
    }
    case KIND_QUADS: {
      auto p = sc.quads + 4 * (sh.elem_eval + (int64_t)element);
      return {p[0], p[1], p[2], p[3]};
    }
    case KIND_LINES: {
      auto p = sc.lines + 2 * (sh.elem_eval + (int64_t)element);
      return {p[0], p[1], 0, 0};
    }
    case KIND_POINTS: return {sc.points[sh.elem_eval + (int64_t)element], 0, 0, 0};
    default: return {0, 0, 0, 0};
  }
}

// The vertices of a hit triangle as the walk read them: the leaf record of the hit (Hit::leaf — DScene::leafdata holds every
// primitive's vertices pre-gathered in leaf order, bit-identical copies of `positions`).  The shading point's position and
// geometric normal then come from three contiguous 16-B loads that depend on nothing but the hit, instead of the element's index
// triple and three gathers behind it (yocto_scene.cpp:288-336 read shape.triangles[element] -> shape.positions[]); a triangle
// mesh without vertex normals or colours needs no index fetch at all (round 6).  Null: read through the indices.
struct TriPos {
  vec3f p0, p1, p2;
};
#ifndef YT_LEAF_SHADE
#define YT_LEAF_SHADE 1
#endif
YT_FN TriPos load_tripos(const DScene& sc, int leaf) {
  const float4* L = sc.leafdata + leaf;
  const float4  a = L[0], b = L[1], c = L[2];
  return {{a.x, a.y, a.z}, {a.w, b.x, b.y}, {b.z, b.w, c.x}};
}
// shape-local position (yocto_shape.cpp:63-82 for points; interpolation otherwise)
YT_FN vec3f eval_position_local(const DScene& sc, const DShape& sh, elem4 e, vec2f uv, const TriPos* tp = nullptr) {
  auto P = sc.positions + 3 * (int64_t)sh.positions;
  switch (sh.kind_eval) {
    case KIND_TRIANGLES:
      if (tp) return interpolate_triangle(tp->p0, tp->p1, tp->p2, uv);
      return interpolate_triangle(ld3(P, e.x), ld3(P, e.y), ld3(P, e.z), uv);
    case KIND_QUADS:
      return interpolate_quad(ld3(P, e.x), ld3(P, e.y), ld3(P, e.z), ld3(P, e.w), uv);
    case KIND_LINES: return interpolate_line(ld3(P, e.x), ld3(P, e.y), uv.x);
    case KIND_POINTS: return ld3(P, e.x);
    default: return {0, 0, 0};
  }
}
// eval_position — yocto_scene.cpp:288-311
YT_FN vec3f eval_position(const DScene& sc, const frame3f& frame, const DShape& sh, elem4 e, vec2f uv, const TriPos* tp = nullptr) {
  if (sh.kind_eval == KIND_NONE) return {0, 0, 0};
  return transform_point(frame, eval_position_local(sc, sh, e, uv, tp));
}
// eval_element_normal — yocto_scene.cpp:314-336
YT_FN vec3f eval_element_normal(const DScene& sc, const frame3f& frame, const DShape& sh, elem4 e, const TriPos* tp = nullptr) {
  auto P = sc.positions + 3 * (int64_t)sh.positions;
  switch (sh.kind_eval) {
    case KIND_TRIANGLES:
      if (tp) return transform_normal(frame, triangle_normal(tp->p0, tp->p1, tp->p2));
      return transform_normal(frame, triangle_normal(ld3(P, e.x), ld3(P, e.y), ld3(P, e.z)));
    case KIND_QUADS:
      return transform_normal(frame, quad_normal(ld3(P, e.x), ld3(P, e.y), ld3(P, e.z), ld3(P, e.w)));
    case KIND_LINES: return transform_normal(frame, normalize(ld3(P, e.y) - ld3(P, e.x)));
    case KIND_POINTS: return {0, 0, 1};
    default: return {0, 0, 0};
  }
}
// eval_normal — yocto_scene.cpp:339-366
YT_FN vec3f eval_normal(const DScene& sc, const frame3f& frame, const DShape& sh, elem4 e, vec2f uv, const TriPos* tp = nullptr) {
  if (sh.normals < 0) return eval_element_normal(sc, frame, sh, e, tp);
  auto N = sc.normals + 3 * (int64_t)sh.normals;
  switch (sh.kind_eval) {
    case KIND_TRIANGLES:
      return transform_normal(
          frame, normalize(interpolate_triangle(ld3(N, e.x), ld3(N, e.y), ld3(N, e.z), uv)));
    case KIND_QUADS:
      return transform_normal(frame,
          normalize(interpolate_quad(ld3(N, e.x), ld3(N, e.y), ld3(N, e.z), ld3(N, e.w), uv)));
    case KIND_LINES:
      return transform_normal(frame, normalize(interpolate_line(ld3(N, e.x), ld3(N, e.y), uv.x)));
    case KIND_POINTS: return transform_normal(frame, normalize(ld3(N, e.x)));
    default: return {0, 0, 0};
  }
}
// eval_texcoord — yocto_scene.cpp:369-389
YT_FN vec2f eval_texcoord(const DScene& sc, const DShape& sh, elem4 e, vec2f uv) {
  if (sh.texcoords < 0) return uv;
  auto T = sc.texcoords + 2 * (int64_t)sh.texcoords;
  switch (sh.kind_eval) {
    case KIND_TRIANGLES: return interpolate_triangle(ld2(T, e.x), ld2(T, e.y), ld2(T, e.z), uv);
    case KIND_QUADS: return interpolate_quad(ld2(T, e.x), ld2(T, e.y), ld2(T, e.z), ld2(T, e.w), uv);
    case KIND_LINES: return interpolate_line(ld2(T, e.x), ld2(T, e.y), uv.x);
    case KIND_POINTS: return ld2(T, e.x);
    default: return {0, 0};
  }
}
// eval_color — yocto_scene.cpp:508-528
YT_FN vec4f eval_color(const DScene& sc, const DShape& sh, elem4 e, vec2f uv) {
  if (sh.colors < 0) return {1, 1, 1, 1};
  auto Cc = sc.colors + 4 * (int64_t)sh.colors;
  switch (sh.kind_eval) {
    case KIND_TRIANGLES: return interpolate_triangle(ld4(Cc, e.x), ld4(Cc, e.y), ld4(Cc, e.z), uv);
    case KIND_QUADS:
      return interpolate_quad(ld4(Cc, e.x), ld4(Cc, e.y), ld4(Cc, e.z), ld4(Cc, e.w), uv);
    case KIND_LINES: return interpolate_line(ld4(Cc, e.x), ld4(Cc, e.y), uv.x);
    case KIND_POINTS: return ld4(Cc, e.x);
    default: return {0, 0, 0, 0};
  }
}

// triangle_tangents_fromuv — yocto_geometry.h:610-627
YT_FN void triangle_tangents_fromuv(vec3f p0, vec3f p1, vec3f p2, vec2f uv0, vec2f uv1, vec2f uv2,
    vec3f& tu, vec3f& tv) {
  auto p = p1 - p0, q = p2 - p0;
  auto s = vec2f{uv1.x - uv0.x, uv2.x - uv0.x};
  auto t = vec2f{uv1.y - uv0.y, uv2.y - uv0.y};
  auto div = s.x * t.y - s.y * t.x;
  if (div != 0) {
    tu = vec3f{t.y * p.x - t.x * q.x, t.y * p.y - t.x * q.y, t.y * p.z - t.x * q.z} / div;
    tv = vec3f{s.x * q.x - s.y * p.x, s.x * q.y - s.y * p.y, s.x * q.z - s.y * p.z} / div;
  } else {
    tu = {1, 0, 0};
    tv = {0, 1, 0};
  }
}

// eval_normalmap — yocto_scene.cpp:446-466 (+ eval_element_tangents :424-444)
YT_FN vec3f eval_normalmap(const DScene& sc, const frame3f& frame, const DShape& sh,
    const ythip_material& material, elem4 e, vec2f uv) {
  auto normal   = eval_normal(sc, frame, sh, e, uv);
  auto texcoord = eval_texcoord(sc, sh, e, uv);
  if (material.normal_tex != YTHIP_INVALIDID &&
      (sh.kind_eval == KIND_TRIANGLES || sh.kind_eval == KIND_QUADS)) {
    const auto& ntex = sc.textures[material.normal_tex];
    // eval_texture(normal_tex, texcoord, false) → texture's own nearest/clamp
    auto normalmap = -1 + 2 * xyz(eval_texture(sc, material.normal_tex, texcoord, false));
    (void)ntex;
    vec3f tu = {0, 0, 0}, tv = {0, 0, 0};
    if (sh.texcoords >= 0) {
      auto P = sc.positions + 3 * (int64_t)sh.positions;
      auto T = sc.texcoords + 2 * (int64_t)sh.texcoords;
      vec3f ltu, ltv;
      if (sh.kind_eval == KIND_TRIANGLES) {
        triangle_tangents_fromuv(ld3(P, e.x), ld3(P, e.y), ld3(P, e.z), ld2(T, e.x), ld2(T, e.y),
            ld2(T, e.z), ltu, ltv);
      } else {
        // quad_tangents_fromuv(..., current_uv = {0,0}) → first triangle (p0,p1,p3)
        triangle_tangents_fromuv(ld3(P, e.x), ld3(P, e.y), ld3(P, e.w), ld2(T, e.x), ld2(T, e.y),
            ld2(T, e.w), ltu, ltv);
      }
      tu = transform_direction(frame, ltu);
      tv = transform_direction(frame, ltv);
    }
    auto fx     = orthonormalize(tu, normal);
    auto fy     = normalize(cross(normal, fx));
    auto flip_v = dot(fy, tv) < 0;
    normalmap.y *= flip_v ? 1 : -1;
    // transform_normal(frame{fx,fy,normal,0}, normalmap) rigid → normalize(transform_vector)
    normal = normalize(fx * normalmap.x + fy * normalmap.y + normal * normalmap.z);
  }
  return normal;
}

// eval_shading_position — yocto_scene.cpp:469-482 (points: shape-local, a
// reference quirk kept for parity)
YT_FN vec3f eval_shading_position(const DScene& sc, const frame3f& frame, const DShape& sh, elem4 e,
    vec2f uv, const TriPos* tp = nullptr) {
  if (sh.kind_eval == KIND_TRIANGLES || sh.kind_eval == KIND_QUADS || sh.kind_eval == KIND_LINES) {
    return eval_position(sc, frame, sh, e, uv, tp);
  } else if (sh.kind_eval == KIND_POINTS) {
    return eval_position_local(sc, sh, e, uv);
  }
  return {0, 0, 0};
}
// eval_shading_normal — yocto_scene.cpp:485-505
template <bool NOTEX = false>
YT_FN vec3f eval_shading_normal(const DScene& sc, const frame3f& frame, const DShape& sh,
    const ythip_material& material, elem4 e, vec2f uv, vec3f outgoing, const TriPos* tp = nullptr) {
  if (sh.kind_eval == KIND_TRIANGLES || sh.kind_eval == KIND_QUADS) {
    auto normal = eval_normal(sc, frame, sh, e, uv, tp);
    if (!NOTEX && material.normal_tex != YTHIP_INVALIDID) normal = eval_normalmap(sc, frame, sh, material, e, uv);
    if (material.type == YTHIP_REFRACTIVE) return normal;
    return dot(normal, outgoing) >= 0 ? normal : -normal;
  } else if (sh.kind_eval == KIND_LINES) {
    auto normal = eval_normal(sc, frame, sh, e, uv);
    return orthonormalize(outgoing, normal);
  } else if (sh.kind_eval == KIND_POINTS) {
    return outgoing;
  }
  return {0, 0, 0};
}

// ---------------------------------------------------------------------------
// material_point / eval_material — yocto_scene.h:258-270, yocto_scene.cpp:531-581
// ---------------------------------------------------------------------------
// (struct material_point: yt_material.h)

// What the caller knows about the resident scene (checked at upload; same arithmetic on the live path):
// NOTEX: no material references a texture.  OPAQUE: the "opaque textured" class — every material is matte, glossy or
// reflective (no transmission: no density) and the only texture slots in use are color_tex and normal_tex.
template <bool NOTEX = false, bool OPAQUE = false>
YT_FN material_point eval_material(const DScene& sc, const DShape& sh, const ythip_material& material_,
    elem4 e, vec2f uv) {
  ythip_material material = material_;
  if (NOTEX) material.emission_tex = material.color_tex = material.roughness_tex = material.scattering_tex = -1;
  if (OPAQUE) {
    material.emission_tex = material.roughness_tex = material.scattering_tex = -1;
    if ((unsigned)material.type > (unsigned)YTHIP_REFLECTIVE) material.type = YTHIP_MATTE;  // (never: tells the compiler)
  }
  // texcoords are only read by texture lookups: skip the three gathers for untextured materials
  const bool textured = (material.emission_tex & material.color_tex & material.roughness_tex &
                            material.scattering_tex) != YTHIP_INVALIDID;
  auto texcoord       = textured ? eval_texcoord(sc, sh, e, uv) : vec2f{0, 0};
  auto emission_tex   = eval_texture(sc, material.emission_tex, texcoord, true);
  auto color_shp      = eval_color(sc, sh, e, uv);
  auto color_tex      = eval_texture(sc, material.color_tex, texcoord, true);
  auto roughness_tex  = eval_texture(sc, material.roughness_tex, texcoord, false);
  auto scattering_tex = eval_texture(sc, material.scattering_tex, texcoord, true);

  material_point point;
  point.type     = material.type;
  auto memission = vec3f{material.emission[0], material.emission[1], material.emission[2]};
  auto mcolor    = vec3f{material.color[0], material.color[1], material.color[2]};
  auto mscatter  = vec3f{material.scattering[0], material.scattering[1], material.scattering[2]};
  point.emission = memission * xyz(emission_tex) * xyz(color_shp);
  point.color    = mcolor * xyz(color_tex) * xyz(color_shp);
  point.opacity  = material.opacity * color_tex.w * color_shp.w;
  point.metallic = material.metallic * roughness_tex.z;
  point.roughness    = material.roughness * roughness_tex.y;
  point.roughness    = point.roughness * point.roughness;
  point.ior          = material.ior;
  point.scattering   = mscatter * xyz(scattering_tex);
  point.scanisotropy = material.scanisotropy;
  point.trdepth      = material.trdepth;

  if (!OPAQUE && (material.type == YTHIP_REFRACTIVE || material.type == YTHIP_VOLUMETRIC ||
                     material.type == YTHIP_SUBSURFACE)) {
    point.density = -log_(clamp_(point.color, 0.0001f, 1.0f)) / point.trdepth;
  } else {
    point.density = {0, 0, 0};
  }

  if (point.type == YTHIP_MATTE || point.type == YTHIP_GLTFPBR || point.type == YTHIP_GLOSSY) {
    point.roughness = clamp_(point.roughness, min_roughness, 1.0f);
  } else if (!OPAQUE && material.type == YTHIP_VOLUMETRIC) {
    point.roughness = 0;
  } else {
    if (point.roughness < min_roughness) point.roughness = 0;
  }
  return point;
}
YT_FN bool is_delta(const material_point& m) {  // yocto_scene.cpp:265-273
  return (m.type == YTHIP_REFLECTIVE && m.roughness == 0) ||
         (m.type == YTHIP_REFRACTIVE && m.roughness == 0) ||
         (m.type == YTHIP_TRANSPARENT && m.roughness == 0) || (m.type == YTHIP_VOLUMETRIC);
}
YT_FN bool is_volumetric(const ythip_material& m) {  // yocto_scene.cpp:258-262
  return m.type == YTHIP_REFRACTIVE || m.type == YTHIP_VOLUMETRIC || m.type == YTHIP_SUBSURFACE;
}

// ---------------------------------------------------------------------------
// eval_environment — yocto_scene.cpp:596-613
// ---------------------------------------------------------------------------
YT_FN vec3f eval_environment(const DScene& sc, int env, vec3f direction) {
  const auto environment = SCALAR_LOADS ? ldc_record(sc.environments + env) : sc.environments[env];  // (env is wave-uniform)
  auto       emission = vec3f{environment.emission[0], environment.emission[1], environment.emission[2]};
  if (environment.emission_tex == YTHIP_INVALIDID) {
    // eval_texture(invalidid) = {1,1,1,1}: the lat-long lookup does not influence the result
    return emission * vec3f{1, 1, 1};
  }
  auto inv      = ldframe(sc.env_inv + 12 * env);
  auto wl       = transform_direction(inv, direction);
  auto texcoord = vec2f{div_(ytm::atan2f(wl.z, wl.x), 2 * pif), div_(ytm::acosf(clamp_(wl.y, -1.0f, 1.0f)), pif)};
  if (texcoord.x < 0) texcoord.x += 1;
  return emission * xyz(eval_texture(sc, environment.emission_tex, texcoord, false));
}
YT_FN vec3f eval_environment(const DScene& sc, vec3f direction) {
  auto emission = vec3f{0, 0, 0};
  for (auto env = 0; env < sc.num_environments; env++) emission += eval_environment(sc, env, direction);
  return emission;
}

}  // namespace yt
#ifdef YT_FAST
#pragma clang fp contract(off)
#endif

