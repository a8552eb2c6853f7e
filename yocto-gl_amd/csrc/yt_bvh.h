// yt_bvh.h — device BVH traversal: intersect_scene_bvh / intersect_shape_bvh /
// intersect_instance_bvh of libs/yocto/yocto_bvh.cpp:460-628 as ONE two-level
// state machine with a single per-lane stack (LDS-resident, scratch overflow),
// plus the ray-primitive intersectors of libs/yocto/yocto_geometry.h:697-864.
//
// Bit-exact contract: identical tree, identical
// push order (ray_dsign[node.axis]), bbox test at POP time, `t > tmax` reject
// (a later equal-t primitive replaces an earlier one), ternary min/max, no FMA.
#pragma once

#include "yt_scene.h"

namespace yt {

constexpr int YT_BLOCK      = 256;  // threads per workgroup (4 waves)
constexpr int YT_LDS_DEPTH  = 32;   // stack entries per lane kept in LDS
constexpr int YT_SPILL      = 96;   // further entries in scratch (total 128 = reference)

// Per-lane traversal stack.  LDS layout [level][thread]: lane l at any level
// hits bank l%32, i.e. conflict-free for ds_read/write_b32.
constexpr int ENTRY_DROPPED = -1;  // == ENTRY_EXIT: unwinds safely
struct Stack {
  int* lds;  // &s_stack[0][threadIdx.x]
  int  sp;
  int  spill[YT_SPILL];
  YT_FN void push(int v) {
    if (sp < YT_LDS_DEPTH)
      lds[sp * YT_BLOCK] = v;
    else if (sp < YT_LDS_DEPTH + YT_SPILL)
      spill[sp - YT_LDS_DEPTH] = v;
    sp++;  // entries beyond 128 are dropped (the reference's array<int,128> would overflow)
  }
  YT_FN int pop() {
    sp--;
    if (sp < YT_LDS_DEPTH) return lds[sp * YT_BLOCK];
    return (sp < YT_LDS_DEPTH + YT_SPILL) ? spill[sp - YT_LDS_DEPTH] : ENTRY_DROPPED;
  }
};

struct Hit {
  int   instance, element;
  float u, v, distance;
  bool  hit;
};

struct Counters {
  unsigned nodes, triangles, quads, lines, points, instances, rays;
};

struct PrimHit {
  float u, v, t;
  bool  hit;
};

// intersect_triangle — yocto_geometry.h:794-825 (Möller–Trumbore, no epsilon)
YT_FN PrimHit intersect_triangle(vec3f o, vec3f d, float tmin, float tmax, vec3f p0, vec3f p1, vec3f p2) {
  auto edge1 = p1 - p0;
  auto edge2 = p2 - p0;
  auto pvec  = cross(d, edge2);
  auto det   = dot(edge1, pvec);
  if (det == 0) return {0, 0, flt_max, false};
  auto inv_det = 1.0f / det;
  auto tvec    = o - p0;
  auto u       = dot(tvec, pvec) * inv_det;
  if (u < 0 || u > 1) return {0, 0, flt_max, false};
  auto qvec = cross(tvec, edge1);
  auto v    = dot(d, qvec) * inv_det;
  if (v < 0 || u + v > 1) return {0, 0, flt_max, false};
  auto t = dot(edge2, qvec) * inv_det;
  if (t < tmin || t > tmax) return {0, 0, flt_max, false};
  return {u, v, t, true};
}
// intersect_quad — yocto_geometry.h:828-835
YT_FN PrimHit intersect_quad(vec3f o, vec3f d, float tmin, float tmax, vec3f p0, vec3f p1, vec3f p2,
    vec3f p3) {
  if (p2 == p3) return intersect_triangle(o, d, tmin, tmax, p0, p1, p3);
  auto isec1 = intersect_triangle(o, d, tmin, tmax, p0, p1, p3);
  auto isec2 = intersect_triangle(o, d, tmin, tmax, p2, p3, p1);
  if (isec2.hit) {
    isec2.u = 1 - isec2.u;
    isec2.v = 1 - isec2.v;
  }
  return isec1.t < isec2.t ? isec1 : isec2;
}
// intersect_line — yocto_geometry.h:716-757
YT_FN PrimHit intersect_line(vec3f o, vec3f dd, float tmin, float tmax, vec3f p0, vec3f p1, float r0,
    float r1) {
  auto u   = dd;
  auto v   = p1 - p0;
  auto w   = o - p0;
  auto a   = dot(u, u);
  auto b   = dot(u, v);
  auto c   = dot(v, v);
  auto d   = dot(u, w);
  auto e   = dot(v, w);
  auto det = a * c - b * b;
  if (det == 0) return {0, 0, flt_max, false};
  auto t = (b * e - c * d) / det;
  auto s = (a * e - b * d) / det;
  if (t < tmin || t > tmax) return {0, 0, flt_max, false};
  s        = clamp_(s, (float)0, (float)1);
  auto pr  = o + dd * t;
  auto pl  = p0 + (p1 - p0) * s;
  auto prl = pr - pl;
  auto d2  = dot(prl, prl);
  auto r   = r0 * (1 - s) + r1 * s;
  if (d2 > r * r) return {0, 0, flt_max, false};
  return {s, sqrt_(d2) / r, t, true};
}
// intersect_point — yocto_geometry.h:697-713
YT_FN PrimHit intersect_point(vec3f o, vec3f d, float tmin, float tmax, vec3f p, float r) {
  auto w = p - o;
  auto t = dot(w, d) / dot(d, d);
  if (t < tmin || t > tmax) return {0, 0, flt_max, false};
  auto rp  = o + d * t;
  auto prp = p - rp;
  if (dot(prp, prp) > r * r) return {0, 0, flt_max, false};
  return {0, 0, t, true};
}
// intersect_bbox(ray, ray_dinv, bbox) — yocto_geometry.h:854-864
YT_FN bool intersect_bbox(vec3f o, vec3f dinv, float tmin, float tmax, vec3f bmin, vec3f bmax) {
  auto it_min = (bmin - o) * dinv;
  auto it_max = (bmax - o) * dinv;
  auto tmn    = min3_(it_min, it_max);
  auto tmx    = max3_(it_min, it_max);
  auto t0     = max_(max_(tmn), tmin);
  auto t1     = min_(min_(tmx), tmax);
  t1 *= 1.00000024f;
  return t0 <= t1;
}

constexpr int ENTRY_EXIT = -1;  // end of the current instance's BLAS entries

// Leaf-data strides in float4 units, by kind_bvh.
YT_FN int leaf_stride(int kind) {
  return kind == KIND_TRIANGLES ? 3 : (kind == KIND_QUADS ? 4 : (kind == KIND_LINES ? 3 : 2));
}

// The traversal.  `only_instance` < 0: intersect_scene_bvh (yocto_bvh.cpp:554-617);
// otherwise intersect_instance_bvh of that instance (yocto_bvh.cpp:619-628).
template <bool COUNT>
YT_FN Hit traverse(const DScene& sc, const ray3f& wray, int only_instance, bool find_any, Stack& st,
    Counters& cnt) {
  Hit best = {-1, -1, 0, 0, 0, false};

  // world-level ray + the ray of the level being walked
  const vec3f wo = wray.o, wd = wray.d;
  const float tmin  = wray.tmin;
  float       tmax  = wray.tmax;
  const vec3f wdinv = {1 / wd.x, 1 / wd.y, 1 / wd.z};
  const int   wsign = ((wdinv.x < 0) ? 1 : 0) | ((wdinv.y < 0) ? 2 : 0) | ((wdinv.z < 0) ? 4 : 0);
  vec3f o = wo, d = wd, dinv = wdinv;
  int   sign     = wsign;
  int   cur_inst = -1;   // instance whose BLAS is being walked, -1 at TLAS level
  int   kind     = KIND_NONE;
  bool  cur_last = false, blas_hit = false;

  if (COUNT && only_instance < 0) cnt.rays++;  // intersect_scene_bvh call (yocto_bvh.cpp:554)
  const float4* nodes4 = reinterpret_cast<const float4*>(sc.nodes);
  st.sp               = 0;

  auto enter = [&](int inst) -> bool {
    const float4* ti  = reinterpret_cast<const float4*>(sc.tinst + inst);
    float4        m0 = ti[0], m1 = ti[1], m2 = ti[2];
    int4          m3 = reinterpret_cast<const int4*>(ti)[3];
    if (m3.x < 0) return false;  // empty shape BVH → miss (yocto_bvh.cpp:466)
    frame3f inv = {{m0.x, m0.y, m0.z}, {m0.w, m1.x, m1.y}, {m1.z, m1.w, m2.x}, {m2.y, m2.z, m2.w}};
    // transform_ray(inverse(frame, true), ray) — yocto_geometry.h:441-443
    o        = transform_point(inv, wo);
    d        = transform_vector(inv, wd);
    dinv     = {1 / d.x, 1 / d.y, 1 / d.z};
    sign     = ((dinv.x < 0) ? 1 : 0) | ((dinv.y < 0) ? 2 : 0) | ((dinv.z < 0) ? 4 : 0);
    cur_inst = inst;
    kind     = m3.y;
    blas_hit = false;
    st.push(ENTRY_EXIT);
    st.push(m3.x);
    return true;
  };

  // back to the TLAS level: restore the world ray.  Returns true when the
  // find_any early-out of intersect_scene_bvh fires (yocto_bvh.cpp:613: checked
  // after the whole TLAS leaf has been processed).
  auto exit_instance = [&]() -> bool {
    o = wo, d = wd, dinv = wdinv, sign = wsign;
    cur_inst = -1;
    return find_any && cur_last && best.hit;
  };

  if (only_instance >= 0) {
    cur_last = true;
    if (!enter(only_instance)) return best;
  } else {
    if (sc.tlas_root < 0) return best;
    st.push(sc.tlas_root);
  }

  while (st.sp > 0) {
    int e = st.pop();
    if (e >= 0) {
      float4 na = nodes4[2 * (int64_t)e], nb = nodes4[2 * (int64_t)e + 1];
      if (COUNT) cnt.nodes++;
      if (!intersect_bbox(o, dinv, tmin, tmax, {na.x, na.y, na.z}, {na.w, nb.x, nb.y})) continue;
      int      start    = __float_as_int(nb.z);
      unsigned packed   = (unsigned)__float_as_int(nb.w);
      int      num      = (int)(short)(packed & 0xffffu);
      int      axis     = (int)((packed >> 16) & 0xffu);
      bool     internal = (packed >> 24) != 0;
      if (internal) {
        // near-first along the split axis — yocto_bvh.cpp:498-504
        if ((sign >> axis) & 1) {
          st.push(start + 0);
          st.push(start + 1);
        } else {
          st.push(start + 1);
          st.push(start + 0);
        }
      } else if (cur_inst < 0) {
        // TLAS leaf: instances must be walked in order, each to completion
        // (yocto_bvh.cpp:600-609) → push continuation entries in reverse.
        for (int k = num - 1; k >= 0; k--) st.push(-2 - (((start + k) << 1) | (k == num - 1 ? 1 : 0)));
      } else {
        // BLAS leaf — yocto_bvh.cpp:505-545
        const float4* L = sc.leafdata + start;
        if (kind == KIND_TRIANGLES) {
          for (int k = 0; k < num; k++) {
            float4 a = L[3 * k], b = L[3 * k + 1], c = L[3 * k + 2];
            if (COUNT) cnt.triangles++;
            auto   h = intersect_triangle(o, d, tmin, tmax, {a.x, a.y, a.z}, {a.w, b.x, b.y}, {b.z, b.w, c.x});
            if (!h.hit) continue;
            best     = {cur_inst, __float_as_int(c.y), h.u, h.v, h.t, true};
            tmax     = h.t;
            blas_hit = true;
          }
        } else if (kind == KIND_QUADS) {
          for (int k = 0; k < num; k++) {
            float4 a = L[4 * k], b = L[4 * k + 1], c = L[4 * k + 2], e4 = L[4 * k + 3];
            if (COUNT) cnt.quads++;
            auto   h = intersect_quad(o, d, tmin, tmax, {a.x, a.y, a.z}, {a.w, b.x, b.y}, {b.z, b.w, c.x},
                  {c.y, c.z, c.w});
            if (!h.hit) continue;
            best     = {cur_inst, __float_as_int(e4.x), h.u, h.v, h.t, true};
            tmax     = h.t;
            blas_hit = true;
          }
        } else if (kind == KIND_LINES) {
          for (int k = 0; k < num; k++) {
            float4 a = L[3 * k], b = L[3 * k + 1], c = L[3 * k + 2];
            if (COUNT) cnt.lines++;
            auto   h = intersect_line(o, d, tmin, tmax, {a.x, a.y, a.z}, {a.w, b.x, b.y}, b.z, b.w);
            if (!h.hit) continue;
            best     = {cur_inst, __float_as_int(c.x), h.u, h.v, h.t, true};
            tmax     = h.t;
            blas_hit = true;
          }
        } else if (kind == KIND_POINTS) {
          for (int k = 0; k < num; k++) {
            float4 a = L[2 * k], b = L[2 * k + 1];
            if (COUNT) cnt.points++;
            auto   h = intersect_point(o, d, tmin, tmax, {a.x, a.y, a.z}, a.w);
            if (!h.hit) continue;
            best     = {cur_inst, __float_as_int(b.x), h.u, h.v, h.t, true};
            tmax     = h.t;
            blas_hit = true;
          }
        }
        // find_any early-out of intersect_shape_bvh — yocto_bvh.cpp:548
        if (find_any && blas_hit) {
          while (st.sp > 0 && st.pop() != ENTRY_EXIT) {
          }
          if (exit_instance()) return best;
        }
      }
    } else if (e == ENTRY_EXIT) {
      if (exit_instance()) return best;
    } else {
      int code = -2 - e;
      cur_last = (code & 1) != 0;
      if (COUNT) cnt.instances++;  // TLAS leaf entry (yocto_bvh.cpp:600-604)
      if (!enter(sc.tlas_prims[code >> 1])) {
        if (find_any && cur_last && best.hit) return best;
      }
    }
  }
  return best;
}

}  // namespace yt
