// yt_own.h — the OWN-TREE walk (ythip_params::fastmath = 2).
//
// The exact and the tolerance mode walk the REFERENCE's trees in the reference's order (yt_bvh.h): that is what makes
// their hit records the reference's, and it is also what binds them — eight 16-B loads per two tree levels through an
// address path that is the co-bound of every incoherent workload.  This mode gives that up, on request:
// radiance within the stated tolerance (tests/test_gpu_own_tree.py gates it against oracle/_ref), hit records that
// agree with the reference's except at exact ties and box-edge grazes (measured and asserted there).
// ythip_intersect_batch never comes here: hit indices for a ray batch stay bit-exact.
//
//   * the tree: the binary SAH tree of the device builder (yt_gpubuild.hip, the reference's split_sah — built whatever
//     trace_params::highqualitybvh says), collapsed two levels per node like the quad records, and each node
//     COMPRESSED into 64 B (k_own_compress, yt_bake.hip): a frame {origin, one power-of-two scale per axis} + the four
//     children's boxes as 8-bit grid coordinates, rounded outwards (conservative), + their refs.  One dependent fetch
//     of FOUR 16-B loads advances two levels (the quad walk: eight).  The closed-room kernels run it with the
//     majority-phase schedule of yt_bvh.h (+10-12 % there, a loss elsewhere: as for the exact kernels);
//   * the slab test in the node's frame: per node A = scale / d, B = (origin - o) / d, then one fused multiply-add per
//     plane, t = q * A + B (q straight from a byte: v_cvt_f32_ubyteN).  The near / far planes are chosen per AXIS by the
//     ray's direction sign, a word select, not per plane by min / max;
//   * no parity obligations: reciprocals instead of IEEE divisions, zero direction components nudged to 1e-20 (no
//     inf * 0), no "tame ray" test and no binary redo, fused multiply-adds everywhere, Möller–Trumbore with one v_rcp, line
//     and point tests on one reciprocal each (own_line / own_point: hair +4.6 %, lines_points +8.5 %).
//     The visit order (near child first by the split axes' signs), the pop-time cull against the current tmax, the
//     TLAS-leaf pretest and the direct enter are the exact walk's — they are what made it fast, not what made it exact.
//
// Same ref encoding, leaf data and instance records as yt_bvh.h (of the SAH tree: DScene of this mode is a copy whose
// bvh pointers are the own tree's — yt_ctx.h: BvhView).
#pragma once

#include "yt_bvh.h"

#pragma clang fp contract(fast)
namespace yt {

// node layout (4 x uint4)
//   q0  origin.x  origin.y  origin.z  scale.z | axes     (a scale is a power of two: its float has an empty mantissa, the axes ride in it)
//   q1  lo.x[4]   lo.y[4]   lo.z[4]   hi.x[4]            (byte s of a word = slot s; slots as in the quad record:
//   q2  hi.y[4]   hi.z[4]   ref[0]    ref[1]              0, 1 = children of child 0 (or child 0 itself + empty), 2, 3 of child 1)
//   q3  ref[2]    ref[3]    scale.x   scale.y
//   axes = node axis | child 0's axis << 2 | child 1's axis << 4   (the quad record's)
// No slack factor on the far side (the reference's 1.00000024): the grid boxes are rounded outwards by up to a cell, and the
// measured hit agreement is the same with and without it (profiles/r05_own_tree.txt).
constexpr float OWN_TINY = 1e-20f;  // |d| below this counts as this (keeps 1 / d finite: no inf * 0 in the plane equations)

YT_FN float own_rcp(float x) {
  const float a = __builtin_fabsf(x) < OWN_TINY ? __builtin_copysignf(OWN_TINY, x) : x;
  return __builtin_amdgcn_rcpf(a);
}
template <int S>
YT_FN float own_byte(unsigned w) {  // v_cvt_f32_ubyteS
  return (float)((w >> (8 * S)) & 0xffu);
}
YT_FN float own_max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
YT_FN float own_min3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }

// an uncompressed box (instance roots, the TLAS root): the interval's near end in t0, false when the ray misses it
// whatever tmax is
YT_FN bool own_box(vec3f o, vec3f idir, float tmin, vec3f bmin, vec3f bmax, float& t0) {
  const float ax = (bmin.x - o.x) * idir.x, bx = (bmax.x - o.x) * idir.x;
  const float ay = (bmin.y - o.y) * idir.y, by = (bmax.y - o.y) * idir.y;
  const float az = (bmin.z - o.z) * idir.z, bz = (bmax.z - o.z) * idir.z;
  t0              = __builtin_fmaxf(own_max3(__builtin_fminf(ax, bx), __builtin_fminf(ay, by), __builtin_fminf(az, bz)), tmin);
  const float far = own_min3(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by), __builtin_fmaxf(az, bz));
  return t0 <= far * BBOX_K;
}

// Möller–Trumbore, one reciprocal, fused multiply-adds (yocto_geometry.h:794-825 is the exact walk's).
// Edge rule (round 6): the barycentric tests take a tolerance of OWN_EDGE_EPS.  Two triangles that share an edge evaluate it from
// different vertices and edge vectors, and with a reciprocal and fused products both can put a ray that passes ON the edge a few
// 1e-8 OUTSIDE: the ray would slip between them.  With the tolerance both neighbours take such a ray — the nearer hit wins as
// always — and the returned coordinates are clamped into the triangle.  (What it did NOT change: the 59 rays of 2.27 M that name
// "another surface" than the reference on the 10 k-sphere scene — those are silhouette grazes of the instance-space ray, as many
// in the reference's favour as in this walk's: tests/test_gpu_own_tree.py.)
constexpr float OWN_EDGE_EPS = 9.5367431640625e-7f;  // 2^-20
YT_FN PrimHit own_triangle(vec3f o, vec3f d, float tmin, float tmax, vec3f p0, vec3f p1, vec3f p2) {
  const vec3f e1 = {p1.x - p0.x, p1.y - p0.y, p1.z - p0.z}, e2 = {p2.x - p0.x, p2.y - p0.y, p2.z - p0.z};
  const vec3f pv = {d.y * e2.z - d.z * e2.y, d.z * e2.x - d.x * e2.z, d.x * e2.y - d.y * e2.x};
  const float det = e1.x * pv.x + e1.y * pv.y + e1.z * pv.z;
  if (det == 0) return {0, 0, flt_max, false};
  const float inv = __builtin_amdgcn_rcpf(det);
  const vec3f tv  = {o.x - p0.x, o.y - p0.y, o.z - p0.z};
  const float u   = (tv.x * pv.x + tv.y * pv.y + tv.z * pv.z) * inv;
  if (!(u >= -OWN_EDGE_EPS && u <= 1 + OWN_EDGE_EPS)) return {0, 0, flt_max, false};
  const vec3f qv = {tv.y * e1.z - tv.z * e1.y, tv.z * e1.x - tv.x * e1.z, tv.x * e1.y - tv.y * e1.x};
  const float v  = (d.x * qv.x + d.y * qv.y + d.z * qv.z) * inv;
  if (!(v >= -OWN_EDGE_EPS && u + v <= 1 + OWN_EDGE_EPS)) return {0, 0, flt_max, false};
  const float t = (e2.x * qv.x + e2.y * qv.y + e2.z * qv.z) * inv;
  if (!(t >= tmin && t <= tmax)) return {0, 0, flt_max, false};
  return {__builtin_fmaxf(u, 0.0f), __builtin_fmaxf(v, 0.0f), t, true};
}

// intersect_line / intersect_point (yocto_geometry.h:716-757, :697-713) on reciprocals and fused multiply-adds; `dd2` = |d|^2
// of the level's ray (kept with the ray: it does not change from segment to segment)
YT_FN PrimHit own_line(vec3f o, vec3f dd, float dd2, float tmin, float tmax, vec3f p0, vec3f p1, float r0, float r1) {
  const vec3f v = {p1.x - p0.x, p1.y - p0.y, p1.z - p0.z}, w = {o.x - p0.x, o.y - p0.y, o.z - p0.z};
  const float a = dd2, b = dd.x * v.x + dd.y * v.y + dd.z * v.z, c = v.x * v.x + v.y * v.y + v.z * v.z;
  const float d = dd.x * w.x + dd.y * w.y + dd.z * w.z, e = v.x * w.x + v.y * w.y + v.z * w.z;
  const float det = a * c - b * b;
  if (det == 0) return {0, 0, flt_max, false};
  const float inv = __builtin_amdgcn_rcpf(det);
  const float t   = (b * e - c * d) * inv;
  if (!(t >= tmin && t <= tmax)) return {0, 0, flt_max, false};
  const float s   = __builtin_fminf(__builtin_fmaxf((a * e - b * d) * inv, 0.0f), 1.0f);
  const vec3f prl = {w.x + dd.x * t - v.x * s, w.y + dd.y * t - v.y * s, w.z + dd.z * t - v.z * s};
  const float d2  = prl.x * prl.x + prl.y * prl.y + prl.z * prl.z;
  const float r   = r0 + (r1 - r0) * s;
  if (d2 > r * r) return {0, 0, flt_max, false};
  return {s, __builtin_sqrtf(d2) * __builtin_amdgcn_rcpf(r), t, true};
}
YT_FN PrimHit own_point(vec3f o, vec3f d, float dd2, float tmin, float tmax, vec3f p, float r) {
  const vec3f w = {p.x - o.x, p.y - o.y, p.z - o.z};
  const float t = (w.x * d.x + w.y * d.y + w.z * d.z) * __builtin_amdgcn_rcpf(dd2);
  if (!(t >= tmin && t <= tmax)) return {0, 0, flt_max, false};
  const vec3f q = {w.x - d.x * t, w.y - d.y * t, w.z - d.z * t};
  if (q.x * q.x + q.y * q.y + q.z * q.z > r * r) return {0, 0, flt_max, false};
  return {0, 0, t, true};
}

typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) const v4u_ cv4u;
YT_FN uint4 ldcu4(const void* p, int k) {  // uint4 #k at the (uniform) address p, by scalar load (yt_bvh.h: ldc4)
  const v4u_ v = ((cv4u*)p)[k];
  return uint4{v.x, v.y, v.z, v.w};
}

// intersect_scene_bvh (only_instance < 0) / intersect_instance_bvh on the own tree.  TRI as in traverse().
template <int TRI, bool PHASED>
YT_FN Hit traverse_own(const DScene& sc, const ray3f& wray, int only_instance, Stack& st, Counters& cnt) {
  constexpr int LDS_LEVELS = YT_LDS_DEPTH, SPILL_LEVELS = 128 - YT_LDS_DEPTH;
  Hit           best       = {-1, -1, 0, 0, 0, false};
  const vec3f   wo = wray.o, wd = wray.d;
  const float   tmin  = wray.tmin;
  float         tmax  = wray.tmax;
  const vec3f   widir = {own_rcp(wd.x), own_rcp(wd.y), own_rcp(wd.z)};
  auto sign_of = [](vec3f i) { return ((i.x < 0) ? 1 : 0) | ((i.y < 0) ? 2 : 0) | ((i.z < 0) ? 4 : 0); };
  const int     wsign = sign_of(widir);
  vec3f         o = wo, d = wd, idir = widir;
  int           sign = wsign, cur_inst = -1, kind = KIND_NONE, leafbias = 0;
  float         dd2 = 0;  // |d|^2 of the level's ray: lines / points only (set where the level's ray is)

  lds_entry* const lds = st.lds;
  int              sp  = 0;
  StackEntry       spill[SPILL_LEVELS];
  YT_STACK_OPS(LDS_LEVELS, SPILL_LEVELS)

  // down into an instance whose record is at hand: the level's ray, the exit marker, the BLAS root
  auto descend = [&](vec3f io, vec3f id, vec3f iidir, int inst, int root, int k, int bias) -> int {
    o = io, d = id, idir = iidir;
    if (TRI == 0) dd2 = id.x * id.x + id.y * id.y + id.z * id.z;
    sign     = sign_of(idir);
    cur_inst = inst;
    kind     = TRI == 1 ? KIND_TRIANGLES : k;
    if (TRI == 2 && kind != KIND_TRIANGLES) kind = KIND_QUADS;
    leafbias = bias;
    push(REF_EXIT, 0);
    return root;
  };
  // intersect_instance_bvh's prologue for one record (`tested`: its root box passed when the TLAS leaf was expanded)
  auto enter = [&](const DInstanceT* base, int idx, int inst, bool tested) -> int {
    float4 m0, m1, m2, m3, m4;
    int4   m5;
    load_instance_record(base, idx, m0, m1, m2, m3, m4, m5);
    const int root = __float_as_int(m4.z);
    if (inst < 0) inst = m5.z;
    if (root == REF_NONE) return REF_NONE;
    const frame3f inv   = {{m0.x, m0.y, m0.z}, {m0.w, m1.x, m1.y}, {m1.z, m1.w, m2.x}, {m2.y, m2.z, m2.w}};
    const vec3f   io    = transform_point(inv, wo);
    const vec3f   id    = transform_vector(inv, wd);
    const vec3f   iidir = {own_rcp(id.x), own_rcp(id.y), own_rcp(id.z)};
    if (!tested) {
      float t0;
      if (!(own_box(io, iidir, tmin, {m3.x, m3.y, m3.z}, {m3.w, m4.x, m4.y}, t0) && t0 <= tmax)) return REF_NONE;
    }
    return descend(io, id, iidir, inst, root, __float_as_int(m4.w), m5.x);
  };

  int cur = REF_NONE;  // node to process next, REF_NONE = pop one
  if (only_instance >= 0) {
    cur = enter(sc.tinst, only_instance, only_instance, false);
    if (cur_inst < 0) return best;
  } else {
    if (sc.tlas_ref == REF_NONE) return best;
    float t0;
    if (!(own_box(o, idir, tmin, sc.tlas_bmin, sc.tlas_bmax, t0) && t0 <= tmax)) return best;
    cur = sc.tlas_ref;
  }
  auto accept = [&](int element, const PrimHit& h, int leaf = 0) {
    best = {cur_inst, element, h.u, h.v, h.t, true, leaf};
    tmax = h.t;
  };

  // ---- the four step kinds, each on this lane's `cur` ---------------------------------------------------------------
  // (1) an internal node: the record by value (the wavefront-uniform form hands it scalar registers)
  auto step = [&](const uint4 n0, const uint4 n1, const uint4 n2, const uint4 n3) __attribute__((always_inline)) {
    const unsigned e = n0.w;
    const float sx = __uint_as_float(n3.z), sy = __uint_as_float(n3.w), sz = __uint_as_float(e & 0x7f800000u);
    const float ax = sx * idir.x, ay = sy * idir.y, az = sz * idir.z;
    const float bx = (__uint_as_float(n0.x) - o.x) * idir.x, by = (__uint_as_float(n0.y) - o.y) * idir.y,
                bz = (__uint_as_float(n0.z) - o.z) * idir.z;
    // the near / far plane words of each axis, by the direction's sign
    const bool     nx = (sign & 1) != 0, ny = (sign & 2) != 0, nz = (sign & 4) != 0;
    const unsigned wnx = nx ? n1.w : n1.x, wfx = nx ? n1.x : n1.w;
    const unsigned wny = ny ? n2.x : n1.y, wfy = ny ? n1.y : n2.x;
    const unsigned wnz = nz ? n2.y : n1.z, wfz = nz ? n1.z : n2.y;
#define YT_OWN_SLOT(S, REF, T0, R)                                                                                      \
  float T0;                                                                                                             \
  int   R;                                                                                                              \
  {                                                                                                                     \
const float near_ = own_max3(own_byte<S>(wnx) * ax + bx, own_byte<S>(wny) * ay + by, own_byte<S>(wnz) * az + bz);   \
const float far_  = own_min3(own_byte<S>(wfx) * ax + bx, own_byte<S>(wfy) * ay + by, own_byte<S>(wfz) * az + bz);   \
T0                = __builtin_fmaxf(near_, tmin);                                                                   \
R                 = (T0 <= __builtin_fminf(far_, tmax)) ? (int)(REF) : REF_NONE;                                    \
  }
    YT_OWN_SLOT(0, n2.z, ta, ra)
    YT_OWN_SLOT(1, n2.w, tb, rb)
    YT_OWN_SLOT(2, n3.x, tc, rc)
    YT_OWN_SLOT(3, n3.y, td, rd)
#undef YT_OWN_SLOT
    // near child first along each split axis (the exact walk's order: yocto_bvh.cpp:498-504 applied twice)
    const int  axes = (int)(e & 63u);
    const bool hs = ((sign >> (axes & 3)) & 1) != 0, ls = ((sign >> ((axes >> 2) & 3)) & 1) != 0,
               rs = ((sign >> ((axes >> 4) & 3)) & 1) != 0;
    const int   l0r = ls ? rb : ra, l1r = ls ? ra : rb, r0r = rs ? rd : rc, r1r = rs ? rc : rd;
    const float l0t = ls ? tb : ta, l1t = ls ? ta : tb, r0t = rs ? td : tc, r1t = rs ? tc : td;
    const int   v0r = hs ? r0r : l0r, v1r = hs ? r1r : l1r, v2r = hs ? l0r : r0r, v3r = hs ? l1r : r1r;
    const float v0t = hs ? r0t : l0t, v1t = hs ? r1t : l1t, v2t = hs ? l0t : r0t, v3t = hs ? l1t : r1t;
    // last to first: whatever passed is pushed, the nearest one becomes `cur`
    int   pr = REF_NONE;
    float pt = 0;
    if (v3r != REF_NONE) pr = v3r, pt = v3t;
    if (v2r != REF_NONE) {
      if (pr != REF_NONE) push(pr, pt);
      pr = v2r, pt = v2t;
    }
    if (v1r != REF_NONE) {
      if (pr != REF_NONE) push(pr, pt);
      pr = v1r, pt = v1t;
    }
    if (v0r != REF_NONE) {
      if (pr != REF_NONE) push(pr, pt);
      pr = v0r, pt = v0t;
    }
    cur = pr;
  };
  auto node_step = [&]() __attribute__((always_inline)) {
    cnt.steps++;
    if (int ucur; SCALAR_LOADS && wave_uniform(cur, ucur)) {  // every stepping lane at the same node: one scalar fetch
      const uint4* Ns = sc.own + 4 * (int64_t)ucur;
      step(ldcu4(Ns, 0), ldcu4(Ns, 1), ldcu4(Ns, 2), ldcu4(Ns, 3));
      return;
    }
    const uint4* Np = sc.own + 4 * (int64_t)cur;
    step(Np[0], Np[1], Np[2], Np[3]);
  };
  // (2) REF_EXIT (back to the TLAS level: the world ray again) or a pushed instance of a TLAS leaf
  auto entry_step = [&]() __attribute__((always_inline)) {
    if (cur == REF_EXIT) {
      cur = REF_NONE;
      o = wo, d = wd, idir = widir, sign = wsign, cur_inst = -1;
      return;
    }
    const int code = cur - REF_INST;  // entry k << 1 | "root box tested"
    cur            = enter(sc.tinst_leaf, code >> 1, -1, (code & 1) != 0);
  };
  // (3) a TLAS leaf: the root boxes of its (<= 4) instances tested at once — independent fetches, one round trip; what
  // the ray can enter is pushed far end first with its t0, the FIRST survivor entered right here with the ray its test
  // transformed (yt_bvh.h: PRETEST / direct enter)
  auto tlas_leaf_step = [&]() __attribute__((always_inline)) {
    const int first = cur & 0x0fffffff, num = (cur >> 28) & 7;
    if (num == 1) {
      cur = REF_INST + (first << 1);
      return;
    }
    for (int k = num - 1; k >= 4; k--) push(REF_INST + ((first + k) << 1), 0);  // (never: leaves hold <= 4) untested
    bool  have = false;
    vec3f co = {0, 0, 0}, cd = {0, 0, 0}, cidir = {0, 0, 0};
    float ct0 = 0;
    int   ck = 0, croot = REF_NONE, ckind = KIND_NONE;
#pragma unroll
    for (int k = 3; k >= 0; k--) {
      if (k >= num) continue;
      float4 m0, m1, m2, m3, m4;
      int4   m5;
      load_instance_record(sc.tinst_leaf, first + k, m0, m1, m2, m3, m4, m5);
      if (__float_as_int(m4.z) == REF_NONE) continue;
      const frame3f inv   = {{m0.x, m0.y, m0.z}, {m0.w, m1.x, m1.y}, {m1.z, m1.w, m2.x}, {m2.y, m2.z, m2.w}};
      const vec3f   io    = transform_point(inv, wo);
      const vec3f   id    = transform_vector(inv, wd);
      const vec3f   iidir = {own_rcp(id.x), own_rcp(id.y), own_rcp(id.z)};
      float         t0    = 0;
      if (!own_box(io, iidir, tmin, {m3.x, m3.y, m3.z}, {m3.w, m4.x, m4.y}, t0)) continue;
      if (have) push(REF_INST + (((first + ck) << 1) | 1), ct0);
      have = true, co = io, cd = id, cidir = iidir, ct0 = t0, ck = k;
      croot = __float_as_int(m4.z), ckind = __float_as_int(m4.w);
    }
    cur = REF_NONE;
    if (have && ct0 <= tmax) {
      const int4 m5 = reinterpret_cast<const int4*>(sc.tinst_leaf + (first + ck))[5];
      cur           = descend(co, cd, cidir, m5.z, croot, ckind, m5.x);
    }
  };
  // (4) a BLAS leaf: its (<= 4) primitives, any order
  auto leaf_step = [&]() __attribute__((always_inline)) {
    const int first = cur & 0x0fffffff, num = (cur >> 28) & 7;
    cur = REF_NONE;
    cnt.steps++;
    if (TRI == 1 || kind == KIND_TRIANGLES) {
      const float4* L = sc.leafdata + (leafbias + first * 3);
      for (int k0 = 0; k0 < num; k0 += 2) {  // two triangles per round trip (the pool is padded, over-reads are ignored)
        float4 a0, b0, c0, a1, b1, c1;
        if (int uoff; SCALAR_LOADS && only_instance >= 0 && wave_uniform(leafbias + first * 3, uoff)) {
          const float4* Lu = sc.leafdata + (uoff + 3 * k0);
          a0 = ldc4(Lu, 0), b0 = ldc4(Lu, 1), c0 = ldc4(Lu, 2), a1 = ldc4(Lu, 3), b1 = ldc4(Lu, 4), c1 = ldc4(Lu, 5);
        } else {
          a0 = L[3 * k0], b0 = L[3 * k0 + 1], c0 = L[3 * k0 + 2];
          a1 = L[3 * k0 + 3], b1 = L[3 * k0 + 4], c1 = L[3 * k0 + 5];
        }
        auto h = own_triangle(o, d, tmin, tmax, {a0.x, a0.y, a0.z}, {a0.w, b0.x, b0.y}, {b0.z, b0.w, c0.x});
        if (h.hit) accept(__float_as_int(c0.y), h, leafbias + first * 3 + 3 * k0);
        if (k0 + 1 < num) {
          h = own_triangle(o, d, tmin, tmax, {a1.x, a1.y, a1.z}, {a1.w, b1.x, b1.y}, {b1.z, b1.w, c1.x});
          if (h.hit) accept(__float_as_int(c1.y), h, leafbias + first * 3 + 3 * k0 + 3);
        }
      }
    } else if (TRI != 1 && kind == KIND_QUADS) {
      const float4* L = sc.leafdata + (leafbias + first * 4);
      for (int k = 0; k < num; k++) {
        const float4 a = L[4 * k], b = L[4 * k + 1], c = L[4 * k + 2], e4 = L[4 * k + 3];
        const vec3f  p0 = {a.x, a.y, a.z}, p1 = {a.w, b.x, b.y}, p2 = {b.z, b.w, c.x}, p3 = {c.y, c.z, c.w};
        auto h = own_triangle(o, d, tmin, tmax, p0, p1, p3);  // (intersect_quad, yocto_geometry.h:828-835, on the fast triangle)
        if (!(p2 == p3)) {
          auto h2 = own_triangle(o, d, tmin, tmax, p2, p3, p1);
          if (h2.hit) h2.u = 1 - h2.u, h2.v = 1 - h2.v;
          if (!(h.t < h2.t)) h = h2;
        }
        if (h.hit) accept(__float_as_int(e4.x), h);
      }
    } else if (TRI == 0 && kind == KIND_LINES) {
      const float4* L = sc.leafdata + (leafbias + first * 3);
      for (int k = 0; k < num; k++) {
        const float4 a = L[3 * k], b = L[3 * k + 1], c = L[3 * k + 2];
        auto h = own_line(o, d, dd2, tmin, tmax, {a.x, a.y, a.z}, {a.w, b.x, b.y}, b.z, b.w);
        if (h.hit) accept(__float_as_int(c.x), h);
      }
    } else if (TRI == 0 && kind == KIND_POINTS) {
      const float4* L = sc.leafdata + (leafbias + first * 2);
      for (int k = 0; k < num; k++) {
        const float4 a = L[2 * k], b = L[2 * k + 1];
        auto h = own_point(o, d, dd2, tmin, tmax, {a.x, a.y, a.z}, a.w);
        if (h.hit) accept(__float_as_int(b.x), h);
      }
    }
  };
  // pop until this lane holds something to do; false when its stack is empty
  auto next = [&]() __attribute__((always_inline)) -> bool {
    while (cur == REF_NONE) {
      if (sp == 0) return false;
      const StackEntry e = pop();
      cur                = e.ref;
      if (e.ref != REF_EXIT && !(__int_as_float(e.t0) <= tmax)) cur = REF_NONE;  // culled at pop time
    }
    return true;
  };

  if constexpr (!PHASED) {
    // while-while: descend until this lane holds a leaf / an entry, then everybody's leaves and entries
    while (true) {
      bool alive = true;
      while ((alive = next()) && (unsigned)cur < (unsigned)REF_INST) node_step();
      if (!alive) break;
      if (cur >= REF_INST) entry_step();
      else if (cur_inst < 0) tlas_leaf_step();
      else leaf_step();
    }
  } else {
    // majority phase (yt_bvh.h: traverse_phased): every lane does its private bookkeeping (pops, culls, exits), then the
    // wavefront runs the ONE step kind most of its lanes wait for
    bool alive = true;
    while (true) {
      if (alive) {
        while ((alive = next()) && cur == REF_EXIT) entry_step();
      }
      const bool wantW = alive && (unsigned)cur < (unsigned)REF_INST;
      const bool wantL = alive && cur < 0 && cur_inst >= 0;
      const bool wantE = alive && !wantW && !wantL;  // an instance entry or a TLAS leaf
      const int  nW = __popcll(__ballot(wantW)), nL = __popcll(__ballot(wantL)), nE = __popcll(__ballot(wantE));
      if (nW + nL + nE == 0) break;
      if (nW > 0 && nW * YT_PHASE_W >= nW + nL + nE) {
        if (wantW) node_step();
      } else if (nL > 0 && nL * YT_PHASE_L >= nL + nE) {
        if (wantL) leaf_step();
      } else if (wantE) {
        if (cur >= REF_INST) entry_step();
        else tlas_leaf_step();
      }
    }
  }
  return best;
}

}  // namespace yt
#pragma clang fp contract(off)
