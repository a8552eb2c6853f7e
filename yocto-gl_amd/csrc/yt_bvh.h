// yt_bvh.h — device BVH traversal: intersect_scene_bvh / intersect_shape_bvh /
// intersect_instance_bvh of libs/yocto/yocto_bvh.cpp:460-628 as ONE two-level
// state machine with a single per-lane stack (LDS-resident, scratch overflow),
// plus the ray-primitive intersectors of libs/yocto/yocto_geometry.h:697-864.
//
// Bit-exact contract: identical tree, identical
// visit order (ray_dsign[node.axis]), the slab test of a node decided with the
// ray.tmax current when the reference would POP it, `t > tmax` reject (a later
// equal-t primitive replaces an earlier one), ternary min/max, no FMA.
//
// What is different from the reference is only WHEN the memory is touched:
//   * the tree is baked into 64-B "sibling pair" records (both children of an
//     internal node: 2 x {bbox, ref}), so one dependent fetch serves the two
//     nodes the reference pops one after the other;
//   * a child's slab interval [t0, far] is computed when its parent is
//     processed; the child is pushed as (ref, t0) and the tmax-dependent half
//     of the reference's test, t0 <= tmax * 1.00000024f, is re-evaluated when
//     it is popped.  Because  min(far, tmax) * k == min(far * k, tmax * k)
//     (rounding is monotonic, k > 0), this is the reference's
//     `t0 <= min(far, tmax) * k` exactly (NaN cases spelled out at slab());
//   * "while-while" control flow: a lane that reaches a leaf waits for the
//     rest of its wavefront to reach one, so primitive tests run convergent.
//   * WIDE mode (the production path; the binary walk stays for work counting,
//     find_any and irregular rays): every internal node also has a 128-B "quad"
//     record holding its (up to) four GRANDCHILDREN {bbox, ref} + the three split
//     axes, so one dependent fetch advances two tree levels — the dependent-fetch
//     chain of a ray, which is what bounds an iteration of k_trace, halves.  The
//     grandchildren are visited in exactly the order the reference's binary walk
//     reaches them (near child of the near child first, ...), each with the same
//     pop-time test, and the two skipped box tests are implied: a box contains its
//     children's boxes and float rounding is monotonic, so for a ray whose slab
//     products are all ordinary numbers "grandchild passes" ⇒ "its parent passes"
//     with the same tmax.  Same leaves in the same order with the same tmax
//     ⇒ the same hit record, bit for bit (tests/test_gpu_parity.py).  A ray that
//     is not "tame" at world level or at the level of an instance it enters makes
//     the wide walk give up (HIT_ABORT) and is redone by the binary walk
//     (traverse_any).
#pragma once

#include "yt_scene.h"

namespace yt {

// Threads per workgroup.  ONE wavefront: the waves of a larger workgroup wait for
// each other at every iteration of k_trace (measured: 30-40 % of the wave time on
// the Cornell-type and instanced scenes, 8 % on the plane); 256 / 128 / 64 threads:
// plane 7.98 / 7.73 / 7.53 ms, Cornell box 1M 40.5 / 39.4 / 35.8 ms, 10k instances
// 44.5 / 41.4 / 38.8 ms per step.
#ifndef YT_BLOCK_SIZE
#define YT_BLOCK_SIZE 64
#endif
constexpr int YT_BLOCK      = YT_BLOCK_SIZE;
#ifndef YT_LDS_LEVELS
#define YT_LDS_LEVELS 8
#endif
// stack entries (8 B) per lane kept in LDS: 4 KB per wave.  6 / 8 / 10 levels measured:
// 10k instances 35.2 / 34.2 / 36.7 ms, hair 40.7 / 38.9 / 41.0 ms (10 levels cost occupancy)
constexpr int YT_LDS_DEPTH  = YT_LDS_LEVELS;
constexpr int YT_SPILL      = 128 - YT_LDS_DEPTH;  // further entries in scratch (total 128 = reference)

// Node references of the baked tree
constexpr int REF_INST = 0x40000000;  // [REF_INST, REF_NONE): TLAS-leaf continuation | (tlas_prim << 1 | last)
constexpr int REF_NONE = 0x7ffffffe;  // nothing / empty tree
constexpr int REF_EXIT = 0x7fffffff;  // end of the current instance's BLAS entries
// ref <  0          leaf:  bit 31 | num << 28 | first primitive (leaf order, global)
// ref in [0, 2^30)  internal: index of its children's sibling-pair record

// Per-lane traversal stack of (ref, t0) entries.  LDS layout [level][thread] of
// 8-B entries: ds_read/write_b64, lane l at any level hits banks 2l, 2l+1 mod 64
// → conflict-free within each 32-lane group.  The pointer is typed address
// space 3 so the accesses are ds_* instructions (a generic pointer would make
// them flat_*), and `sp` / the scratch overflow array are plain locals of
// traverse() so `sp` lives in a register (inside one struct with the array the
// whole struct was demoted to scratch).
struct alignas(8) StackEntry {
  int ref, t0;
};
typedef __attribute__((address_space(3))) StackEntry lds_entry;
struct Stack {
  lds_entry* lds;  // &s_stack[0][threadIdx.x]
};
#define YT_STACK_INIT(stack, s_stack) (stack).lds = (lds_entry*)&(s_stack)[0][threadIdx.x]

// (An LDS copy of the top four tree levels — north star: "BVH nodes staged through LDS" — was built and
// measured 15-18 % slower on every BASELINE scene: the top of the tree is L1-resident anyway and the
// staging costs a prologue per workgroup plus a branch per step.  tools/experiments/README.md.)
typedef __attribute__((address_space(3))) float4 lds_float4;

// push / pop of the per-lane stack (locals `lds`, `sp`, `spill` of the enclosing walk): entries [0, L) in the lane's
// LDS column, [L, L + S) in scratch.  (A ring that keeps the TOP L entries in LDS was measured -4 ... +1 %: pushes land
// beyond the 8 LDS levels in 0-3.6 % of the cases.  tools/experiments/r05_removed_macros.patch.)
#define YT_STACK_OPS(L, S)                                                                        \
  auto push = [&](int ref, float t0) {                                                            \
    StackEntry v = {ref, __float_as_int(t0)};                                                     \
    if (sp < (L))                                                                                 \
      lds[sp * YT_BLOCK].ref = v.ref, lds[sp * YT_BLOCK].t0 = v.t0;                               \
    else if (sp < (L) + (S))                                                                      \
      spill[sp - (L)] = v;                                                                        \
    sp++; /* entries beyond 128 are dropped (the reference's array<int,128> would overflow) */    \
  };                                                                                              \
  auto pop = [&]() -> StackEntry {                                                                \
    sp--;                                                                                         \
    if (sp < (L)) {                                                                               \
      StackEntry v;                                                                               \
      v.ref = lds[sp * YT_BLOCK].ref, v.t0 = lds[sp * YT_BLOCK].t0;                               \
      return v;                                                                                   \
    }                                                                                             \
    return (sp < (L) + (S)) ? spill[sp - (L)] : StackEntry{REF_EXIT, 0};                          \
  };

struct Hit {
  int   instance, element;
  float u, v, distance;
  bool  hit;
  int   leaf;  // float4 index of the hit primitive's record in DScene::leafdata (its vertices, as the walk read them: yt_scene.h TriPos)
};

struct Counters {
  unsigned nodes, triangles, quads, lines, points, instances, rays;
  unsigned steps;  // always on: sibling-pair fetches + leaf visits of this lane (k_trace's scheduling signal)
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
  return {s, sqrt_ieee_(d2) / r, t, true};
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

// Leaf-data strides in float4 units, by kind_bvh.
YT_FN int leaf_stride(int kind) {
  return kind == KIND_TRIANGLES ? 3 : (kind == KIND_QUADS ? 4 : (kind == KIND_LINES ? 3 : 2));
}

constexpr float BBOX_K = 1.00000024f;  // yocto_geometry.h:862

// Slab interval of one box for the ray (o, dinv): the tmax-independent part of
// intersect_bbox (yocto_geometry.h:854-864).  The reference's verdict is
//     t0 <= min_(far, tmax) * k          with min_(a, b) = (a < b) ? a : b
//   far NaN            → min_ = tmax            → t0 <= tmax*k
//   far < tmax         → t0 <= far*k  (and then t0 <= far*k <= tmax*k)
//   otherwise (incl. tmax NaN) → t0 <= tmax*k (and far*k >= tmax*k, or false)
// i.e. exactly  farok && t0 <= tmax*k  with farok = isnan(far) || t0 <= far*k.
//
// TAME = the ray cannot produce a NaN slab product (see ray_is_tame): then the
// ternary min_/max_ chains equal v_min_f32 / v_max3_f32 up to the sign of a
// zero, which no comparison below can see — 9 instructions instead of 24
// compare+select pairs per box.
template <bool TAME>
YT_FN bool slab(vec3f o, vec3f dinv, float tmin, vec3f bmin, vec3f bmax, float& t0) {
  auto it_min = (bmin - o) * dinv;
  auto it_max = (bmax - o) * dinv;
  if constexpr (TAME) {
    // the hardware min/max directly: no operand is a NaN here, so the canonicalising
    // v_max x, x the compiler puts in front of fminf/fmaxf (6 per box) buys nothing
    auto vmin = [](float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
    auto vmax = [](float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
    auto vmin3 = [](float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; };
    auto vmax3 = [](float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; };
    float nx = vmin(it_min.x, it_max.x), ny = vmin(it_min.y, it_max.y), nz = vmin(it_min.z, it_max.z);
    float fx = vmax(it_min.x, it_max.x), fy = vmax(it_min.y, it_max.y), fz = vmax(it_min.z, it_max.z);
    t0       = vmax(vmax3(nx, ny, nz), tmin);
    auto far = vmin3(fx, fy, fz);
    return t0 <= far * BBOX_K;
  } else {
    auto tmn = min3_(it_min, it_max);
    auto tmx = max3_(it_min, it_max);
    t0       = max_(max_(tmn), tmin);
    auto far = min_(tmx);
    return (far != far) || (t0 <= far * BBOX_K);
  }
}
// No slab product (b - o) * dinv can be NaN for finite boxes: o finite, every
// 1/d finite and non-zero (a zero direction component gives inf and 0 * inf on
// a box face through the origin), tmin a number.
YT_FN bool ray_is_tame(vec3f o, vec3f dinv, float tmin) {
  auto ok = [](float x) { return __builtin_isfinite(x); };
  return ok(o.x) && ok(o.y) && ok(o.z) && ok(dinv.x) && ok(dinv.y) && ok(dinv.z) && dinv.x != 0 && dinv.y != 0 &&
         dinv.z != 0 && tmin == tmin;
}

// slab<true> on a baked box record {min.x, min.y, max.x, max.y} {min.z, max.z, ..}: the same six
// subtractions and six multiplications, issued as v_pk_add_f32 / v_pk_mul_f32 pairs (CDNA3+ packed
// fp32: two IEEE operations per lane and instruction, same rounding as the scalar forms).
typedef float v2f __attribute__((ext_vector_type(2)));
YT_FN bool slab_rec(vec3f o, vec3f dinv, float tmin, float4 r0, float4 r1, float& t0) {
  return slab<true>(o, dinv, tmin, {r0.x, r0.y, r1.x}, {r0.z, r0.w, r1.y}, t0);
}

// The grandchildren ("quad") records of the wide walk, as SEVEN 16-B loads (round 6; until round 5: eight — per slot
// {min.x, min.y, max.x, max.y} {min.z, max.z, ref, axes}, the axes word in slot a only, three words unused).  The record keeps its
// 128-B stride and alignment; its first 112 B hold everything a step reads:
//     q0 = a {min.x, min.y, max.x, max.y}    q1 = {a.min.z, a.max.z, b.min.z, b.max.z}    q2 = b {min.x, min.y, max.x, max.y}
//     q3 = c {min.x, ...}                    q4 = {c.min.z, c.max.z, d.min.z, d.max.z}    q5 = d {min.x, ...}
//     q6 = {ref a, ref b, ref c, ref d}
// and the three split axes travel in bits 26-27 of three of the refs: the node's own axis in ref a, child 0's in ref b, child 1's
// in ref d (slot b / d is empty exactly when child 0 / 1 is a leaf, whose axis nobody asks for; an empty slot's ref is REF_NONE
// as before).  A ref keeps those two bits wherever it travels (`cur`, the stack): they are masked off where a ref is USED — the
// record address of an internal node (WIDE_REF_MASK), the first primitive of a leaf (LEAF_FIRST_MASK) — so the bake caps internal
// nodes and primitives at 2^26 each (bake_bvh: larger scenes are walked binary).  One wave-level load less per node step on a
// path that is bound by the vector-memory address unit.  k_quads_repack (yt_bake.hip) makes this layout from the
// plain one the bake kernels write.
#ifndef YT_WIDE7
#define YT_WIDE7 1
#endif
constexpr int WIDE_AXIS_SHIFT = 26;
constexpr int WIDE_REF_MASK   = YT_WIDE7 ? 0x03ffffff : 0x3fffffff;  // record index of an internal ref
constexpr int LEAF_FIRST_MASK = YT_WIDE7 ? 0x03ffffff : 0x0fffffff;  // first primitive of a leaf ref
YT_FN bool slab_rec7(vec3f o, vec3f dinv, float tmin, float4 xy, float zmin, float zmax, float& t0) {
  return slab<true>(o, dinv, tmin, {xy.x, xy.y, zmin}, {xy.z, xy.w, zmax}, t0);
}

// Wavefront-uniform reads through the SCALAR cache (round 4).  The vector-memory address path is this kernel's co-bound
// (profiles/r04_traversal.txt §5: a wave-level 16-B-per-lane load keeps the CU's texture addresser busy ~16-20 cycles
// however many lanes are active).  When every lane that executes a load asks for the SAME record — camera rays of a tile
// in the upper levels of their walks, every ray entering the one instance of a scene, every light-pdf walk of one light —
// the record can come through s_load instead: the baked traversal data never changes during a launch, so it may be read
// through the constant address space.  wave_uniform(x, u) (yt_scene.h): u = x of the first active lane; true when all active lanes agree.
typedef float v4f_ __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) const v4f_ cv4f;
YT_FN float4 ldc4(const void* p, int k) {  // float4 #k at the (uniform) address p, by scalar load
  const v4f_ v = ((cv4f*)p)[k];
  return float4{v.x, v.y, v.z, v.w};
}
// the 96-B traversal record #idx of `base` (sc.tinst or sc.tinst_leaf).  (Round 6 measured the per-shape half of the record —
// root box, root ref, kind, leaf bias — as launch constants for scenes whose instances all refer to ONE shape, half the vector
// loads of a TLAS-leaf entry on BASELINE configs[3]: -0.8 %, tools/experiments/r06_one_shape_records.patch.)
YT_FN void load_instance_record(const DInstanceT* base, int idx, float4& m0, float4& m1, float4& m2, float4& m3, float4& m4, int4& m5) {
  int u;
  if (SCALAR_LOADS && wave_uniform(idx, u)) {
    const DInstanceT* r = base + u;
    m0 = ldc4(r, 0), m1 = ldc4(r, 1), m2 = ldc4(r, 2), m3 = ldc4(r, 3), m4 = ldc4(r, 4);
    const float4 t = ldc4(r, 5);
    m5 = {__float_as_int(t.x), __float_as_int(t.y), __float_as_int(t.z), __float_as_int(t.w)};
    return;
  }
  const float4* ti = reinterpret_cast<const float4*>(base + idx);
  m0 = ti[0], m1 = ti[1], m2 = ti[2], m3 = ti[3], m4 = ti[4];
  m5 = reinterpret_cast<const int4*>(ti)[5];
}

// The traversal.  `only_instance` < 0: intersect_scene_bvh (yocto_bvh.cpp:554-617);
// otherwise intersect_instance_bvh of that instance (yocto_bvh.cpp:619-628).
constexpr int HIT_ABORT = -2;  // Hit::instance of a wide walk that met an irregular ray: redo it binary

// LDSD: stack entries per lane kept in the LDS column of `st` (the rest, up to the
// reference's 128, in scratch).  0 = a walk that leaves the LDS stack alone.
// TRI: what is known about the scene's shapes — 0 nothing, 1 every shape is a triangle mesh, 2 triangle and quad meshes only
// (checked at upload; tells the compiler which intersectors can be reached, the arithmetic of the live ones is the same)
template <bool COUNT, bool WIDE = false, int TRI = 0, int LDSD = YT_LDS_DEPTH>
YT_FN Hit traverse(const DScene& sc, const ray3f& wray, int only_instance, bool find_any, Stack& st,
    Counters& cnt) {
  constexpr int LDS_LEVELS = LDSD, SPILL_LEVELS = 128 - LDSD;
  static_assert(!(COUNT && WIDE), "work counters follow the reference's binary walk");
  Hit best = {-1, -1, 0, 0, 0, false};

  // world-level ray + the ray of the level being walked
  const vec3f wo = wray.o, wd = wray.d;
  const float tmin  = wray.tmin;
  float       tmax  = wray.tmax;
  float       tmaxk = tmax * BBOX_K;
  // tmax only shrinks while hit distances are numbers; a NaN distance (only
  // possible with non-finite rays or vertices) is accepted by the reference's
  // `t < tmin || t > tmax` test and breaks that, so push-time culling against
  // tmax is switched off from then on (pop-time tests stay exact).
  bool        weird = tmax != tmax;
  const vec3f wdinv = {1 / wd.x, 1 / wd.y, 1 / wd.z};
  const int   wsign = ((wdinv.x < 0) ? 1 : 0) | ((wdinv.y < 0) ? 2 : 0) | ((wdinv.z < 0) ? 4 : 0);
  const bool  wtame = ray_is_tame(wo, wdinv, tmin);
  bool abort = false;
  if constexpr (WIDE) {
    if (!wtame || weird || find_any) return Hit{HIT_ABORT, -1, 0, 0, 0, false};
  }
  vec3f o = wo, d = wd, dinv = wdinv;
  int   sign     = wsign;
  bool  tame     = wtame;
  int   cur_inst = -1;   // instance whose BLAS is being walked, -1 at TLAS level
  int   kind     = KIND_NONE;
  int   leafbias = 0;    // float4 index of the shape's leaf data minus first_prim * stride
  bool  cur_last = false, blas_hit = false;

  if (COUNT && only_instance < 0) cnt.rays++;  // intersect_scene_bvh call (yocto_bvh.cpp:554)
  lds_entry* const lds = st.lds;
  int             sp  = 0;
  StackEntry      spill[SPILL_LEVELS];
  YT_STACK_OPS(LDS_LEVELS, SPILL_LEVELS)

  // intersect_shape_bvh prologue for instance `inst`: transform_ray(inverse(frame,
  // true), ray) (yocto_geometry.h:441-443) and the pop + slab test of the BLAS
  // root, whose bbox travels in the instance record.  Returns the root's ref
  // when the walk goes on, REF_NONE (level state untouched) when the shape BVH
  // is empty (yocto_bvh.cpp:466) or the root is culled.
  // `rec`: the instance's traversal record — sc.tinst + inst, or the copy of it in TLAS-leaf order (sc.tinst_leaf + k,
  // whose pad word carries the instance id: one dependent fetch less per TLAS-leaf entry than tlas_prims -> tinst).
  // `tested`: the root-box test has been made when the TLAS leaf was expanded (pretest below).
  auto enter = [&](const DInstanceT* base, int idx, int inst, bool tested = false) -> int {
    float4 m0, m1, m2, m3, m4;
    int4   m5;
    load_instance_record(base, idx, m0, m1, m2, m3, m4, m5);
    int root = __float_as_int(m4.z);
    if (inst < 0) inst = m5.z;
    if (root == REF_NONE) return REF_NONE;
    frame3f inv = {{m0.x, m0.y, m0.z}, {m0.w, m1.x, m1.y}, {m1.z, m1.w, m2.x}, {m2.y, m2.z, m2.w}};
    vec3f   io   = transform_point(inv, wo);
    vec3f   id   = transform_vector(inv, wd);
    vec3f   idin = {1 / id.x, 1 / id.y, 1 / id.z};
    if constexpr (WIDE) {
      if (!ray_is_tame(io, idin, tmin)) {  // irregular at this instance's level: the caller redoes the ray binary
        abort = true;
        return REF_NONE;
      }
    }
    if (COUNT) cnt.nodes++;
    if (!tested) {
      float t0;
      bool  ok = slab<false>(io, idin, tmin, {m3.x, m3.y, m3.z}, {m3.w, m4.x, m4.y}, t0) && t0 <= tmaxk;
      if (!ok) return REF_NONE;
    }
    o = io, d = id, dinv = idin;
    tame = ray_is_tame(o, dinv, tmin);
    sign     = ((dinv.x < 0) ? 1 : 0) | ((dinv.y < 0) ? 2 : 0) | ((dinv.z < 0) ? 4 : 0);
    cur_inst = inst;
    kind     = TRI == 1 ? KIND_TRIANGLES : __float_as_int(m4.w);  // TRI 1: every shape of the scene is a triangle mesh
    if (TRI == 2 && kind != KIND_TRIANGLES) kind = KIND_QUADS;       // TRI 2: ... or a quad mesh
    leafbias = m5.x;
    blas_hit = false;
    push(REF_EXIT, 0);
    return root;
  };
  // entry `k` of the TLAS-leaf order (a continuation entry's code >> 1)
  auto enter_leaf_entry = [&](int k, bool tested) -> int {
    return enter(sc.tinst_leaf, k, -1, tested);
  };
  // PRETEST (the wide walk; round 4, +4 ... +10 % on scenes with instances, profiles/r04_traversal.txt): the tmax-INDEPENDENT half of an instance's root-box test — transform_ray + intersect_bbox's
  // interval (yocto_bvh.cpp:619-628 with :470-477) — made for all (up to 4) instances of a TLAS leaf when the leaf is
  // expanded: their records are independent fetches (one round trip instead of one per instance), and an instance whose
  // root box the ray misses whatever tmax is never becomes an entry at all.  What passes is pushed with its t0 and gets the
  // tmax-dependent half, t0 <= tmax * k, when it is popped — in the reference's order, after the earlier instances of the
  // leaf have shrunk tmax: the same decision on the same floats as testing at entry time (header, and slab()).
  // Returns false for "cannot enter whatever tmax is"; an irregular instance-level ray passes (enter() then aborts).
  auto pretest = [&](int k, float& t0) -> bool {
    float4 m0, m1, m2, m3, m4;
    int4   m5;
    load_instance_record(sc.tinst_leaf, k, m0, m1, m2, m3, m4, m5);
    t0 = 0;
    if (__float_as_int(m4.z) == REF_NONE) return false;
    frame3f inv = {{m0.x, m0.y, m0.z}, {m0.w, m1.x, m1.y}, {m1.z, m1.w, m2.x}, {m2.y, m2.z, m2.w}};
    vec3f   io   = transform_point(inv, wo);
    vec3f   id   = transform_vector(inv, wd);
    vec3f   idin = {1 / id.x, 1 / id.y, 1 / id.z};
    if (!ray_is_tame(io, idin, tmin)) return true;
    return slab<false>(io, idin, tmin, {m3.x, m3.y, m3.z}, {m3.w, m4.x, m4.y}, t0);
  };
  constexpr bool PRETEST = WIDE;

  // back to the TLAS level: restore the world ray.  Returns true when the
  // find_any early-out of intersect_scene_bvh fires (yocto_bvh.cpp:613: checked
  // after the whole TLAS leaf has been processed).
  auto exit_instance = [&]() -> bool {
    o = wo, d = wd, dinv = wdinv, sign = wsign, tame = wtame;
    cur_inst = -1;
    return find_any && cur_last && best.hit;
  };

  int cur = REF_NONE;  // node to process next, REF_NONE = pop one
  if (only_instance >= 0) {
    cur_last = true;
    cur      = enter(sc.tinst, only_instance, only_instance);
    if (WIDE && abort) return Hit{HIT_ABORT, -1, 0, 0, 0, false};
    if (cur_inst < 0) return best;
  } else {
    if (sc.tlas_ref == REF_NONE) return best;
    if (COUNT) cnt.nodes++;
    float t0;
    if (!(slab<false>(o, dinv, tmin, sc.tlas_bmin, sc.tlas_bmax, t0) && t0 <= tmaxk)) return best;
    cur = sc.tlas_ref;
  }

  auto accept = [&](int element, const PrimHit& h, int leaf = 0) {
    best     = {cur_inst, element, h.u, h.v, h.t, true, leaf};
    tmax     = h.t;
    tmaxk    = h.t * BBOX_K;
    weird    = weird || (h.t != h.t);
    blas_hit = true;
  };

  const float4* pairs = sc.pairs;
  bool          done  = false;
  while (!done) {
    // ---- (1) descend: until this lane holds a leaf / instance entry ----------
    while (true) {
      if (cur == REF_NONE) {
        if (sp == 0) {
          done = true;
          break;
        }
        StackEntry e = pop();
        cur          = e.ref;
        // culled at pop time (with PRETEST the instance entries carry their root box's t0 too)
        if ((PRETEST ? e.ref != REF_EXIT : e.ref < REF_INST) && !(__int_as_float(e.t0) <= tmaxk)) cur = REF_NONE;
        if (cur == REF_NONE) continue;
      }
      if ((unsigned)cur >= (unsigned)REF_INST) {
        break;  // BLAS leaf or instance entry → phase 2
      }
      if constexpr (WIDE) {
        // internal node, two levels at once: its grandchildren in the order the
        // reference's walk reaches them, each pushed with its own pop-time test
        cnt.steps++;
        // (the step on a record given by value: the wavefront-uniform form below hands it scalar registers)
#if YT_WIDE7
        auto wide_step = [&](const float4 q0, const float4 q1, const float4 q2, const float4 q3, const float4 q4, const float4 q5,
                             const float4 q6) __attribute__((always_inline)) {
          float ta, tb, tc, td;
          bool  fa = slab_rec7(o, dinv, tmin, q0, q1.x, q1.y, ta);
          bool  fb = slab_rec7(o, dinv, tmin, q2, q1.z, q1.w, tb);
          bool  fc = slab_rec7(o, dinv, tmin, q3, q4.x, q4.y, tc);
          bool  fd = slab_rec7(o, dinv, tmin, q5, q4.z, q4.w, td);
          const int wa = __float_as_int(q6.x), wb = __float_as_int(q6.y), wc = __float_as_int(q6.z), wd = __float_as_int(q6.w);
          // slots: a, b = children of child 0 (or child 0 itself, then b is empty); c, d likewise for child 1
          int ra = (fa && ta <= tmaxk) ? wa : REF_NONE;
          int rb = (fb && tb <= tmaxk) ? wb : REF_NONE;
          int rc = (fc && tc <= tmaxk) ? wc : REF_NONE;
          int rd = (fd && td <= tmaxk) ? wd : REF_NONE;
          // node axis in ref a, child 0's axis in ref b, child 1's in ref d (bits 26-27)
          const bool hs = ((sign >> ((wa >> WIDE_AXIS_SHIFT) & 3)) & 1) != 0, ls = ((sign >> ((wb >> WIDE_AXIS_SHIFT) & 3)) & 1) != 0,
                     rs = ((sign >> ((wd >> WIDE_AXIS_SHIFT) & 3)) & 1) != 0;
#else
        auto wide_step = [&](const float4 a0, const float4 a1, const float4 b0, const float4 b1, const float4 c0, const float4 c1,
                             const float4 d0, const float4 d1) __attribute__((always_inline)) {
          float ta, tb, tc, td;
          // per slot {min.x, min.y, max.x, max.y} {min.z, max.z, ref, axes}, like the pair records
          bool  fa = slab_rec(o, dinv, tmin, a0, a1, ta);
          bool  fb = slab_rec(o, dinv, tmin, b0, b1, tb);
          bool  fc = slab_rec(o, dinv, tmin, c0, c1, tc);
          bool  fd = slab_rec(o, dinv, tmin, d0, d1, td);
          // slots: a, b = children of child 0 (or child 0 itself, then b is empty); c, d likewise for child 1
          int ra = (fa && ta <= tmaxk) ? __float_as_int(a1.z) : REF_NONE;
          int rb = (fb && tb <= tmaxk) ? __float_as_int(b1.z) : REF_NONE;
          int rc = (fc && tc <= tmaxk) ? __float_as_int(c1.z) : REF_NONE;
          int rd = (fd && td <= tmaxk) ? __float_as_int(d1.z) : REF_NONE;
          const int  axes = __float_as_int(a1.w);  // node axis | child 0's axis << 2 | child 1's axis << 4
          const bool hs = ((sign >> (axes & 3)) & 1) != 0, ls = ((sign >> ((axes >> 2) & 3)) & 1) != 0,
                     rs = ((sign >> ((axes >> 4) & 3)) & 1) != 0;
#endif
          // within each half: ray_dsign[child.axis] → its child 1 first (yocto_bvh.cpp:498-504)
          int   l0r = ls ? rb : ra, l1r = ls ? ra : rb, r0r = rs ? rd : rc, r1r = rs ? rc : rd;
          float l0t = ls ? tb : ta, l1t = ls ? ta : tb, r0t = rs ? td : tc, r1t = rs ? tc : td;
          // the halves: ray_dsign[node.axis] → child 1's half first
          int   v0r = hs ? r0r : l0r, v1r = hs ? r1r : l1r, v2r = hs ? l0r : r0r, v3r = hs ? l1r : r1r;
          float v0t = hs ? r0t : l0t, v1t = hs ? r1t : l1t, v2t = hs ? l0t : r0t, v3t = hs ? l1t : r1t;
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
        // Wavefront-uniform steps through the scalar cache (wave_uniform / ldc4 above): when every lane that takes a node
        // step now is at the SAME node the record is fetched once, into scalar registers — no work for the vector-memory
        // address path — and the slab tests take the box as scalar operands.  Same arithmetic: configs[1] +8.4 %, the
        // rest +0 ... +3 %, bit-identical (profiles/r04_traversal.txt §7).
        if (int ucur; SCALAR_LOADS && wave_uniform(cur, ucur)) {
          const float4* Qs = sc.wide + 8 * (int64_t)(ucur & WIDE_REF_MASK);
#if YT_WIDE7
          wide_step(ldc4(Qs, 0), ldc4(Qs, 1), ldc4(Qs, 2), ldc4(Qs, 3), ldc4(Qs, 4), ldc4(Qs, 5), ldc4(Qs, 6));
#else
          wide_step(ldc4(Qs, 0), ldc4(Qs, 1), ldc4(Qs, 2), ldc4(Qs, 3), ldc4(Qs, 4), ldc4(Qs, 5), ldc4(Qs, 6), ldc4(Qs, 7));
#endif
          continue;
        }
        const float4* Qp = sc.wide + 8 * (int64_t)(cur & WIDE_REF_MASK);
        {
#if YT_WIDE7
          float4 q0 = Qp[0], q1 = Qp[1], q2 = Qp[2], q3 = Qp[3], q4 = Qp[4], q5 = Qp[5], q6 = Qp[6];
          asm volatile("" : "+v"(q6.x));  // (the loads whole and where they are: hipcc otherwise narrows them and sinks a ref behind the tests)
          wide_step(q0, q1, q2, q3, q4, q5, q6);
#else
          float4 q0 = Qp[0], q1 = Qp[1], q2 = Qp[2], q3 = Qp[3], q4 = Qp[4], q5 = Qp[5], q6 = Qp[6], q7 = Qp[7];
          asm volatile("" : "+v"(q1.z));
          wide_step(q0, q1, q2, q3, q4, q5, q6, q7);
#endif
          continue;
        }
      }
      // internal node: its two children in the reference's visit order
      // (near-first along the split axis — yocto_bvh.cpp:498-504, 592-598)
      float4 q0, q1, q2, q3;
      if (int ucur; SCALAR_LOADS && wave_uniform(cur, ucur)) {  // (as the wide step: one scalar fetch for the whole wavefront)
        const float4* P = pairs + 4 * (int64_t)ucur;
        q0 = ldc4(P, 0), q1 = ldc4(P, 1), q2 = ldc4(P, 2), q3 = ldc4(P, 3);
      } else {
        const float4* P = pairs + 4 * (int64_t)cur;
        q0 = P[0], q1 = P[1], q2 = P[2], q3 = P[3];
      }
      if (COUNT) cnt.nodes += 2;
      cnt.steps++;
      float t0a, t0b;
      bool  fa, fb;
      // record: {min.x, min.y, max.x, max.y} {min.z, max.z, ref, axis} per child — the x/y
      // and the z pairs sit in adjacent registers for the packed subtract / multiply
      if (tame) {
        fa = slab<true>(o, dinv, tmin, {q0.x, q0.y, q1.x}, {q0.z, q0.w, q1.y}, t0a);
        fb = slab<true>(o, dinv, tmin, {q2.x, q2.y, q3.x}, {q2.z, q2.w, q3.y}, t0b);
      } else {
        fa = slab<false>(o, dinv, tmin, {q0.x, q0.y, q1.x}, {q0.z, q0.w, q1.y}, t0a);
        fb = slab<false>(o, dinv, tmin, {q2.x, q2.y, q3.x}, {q2.z, q2.w, q3.y}, t0b);
      }
      int   axis = __float_as_int(q1.w);
      bool  swp  = ((sign >> axis) & 1) != 0;  // ray_dsign[axis]: child 1 is popped first
      int   r1 = swp ? __float_as_int(q3.z) : __float_as_int(q1.z);
      int   r2 = swp ? __float_as_int(q1.z) : __float_as_int(q3.z);
      float t1 = swp ? t0b : t0a, t2 = swp ? t0a : t0b;
      bool  f1 = swp ? fb : fa, f2 = swp ? fa : fb;
      bool  n1 = f1 && t1 <= tmaxk, n2 = f2 && t2 <= tmaxk;
      if (n1) {
        cur = r1;
        if (weird ? f2 : n2) push(r2, t2);
      } else {
        cur = n2 ? r2 : REF_NONE;
      }
    }
    if (done) break;

    // ---- (2) leaves, instance entries ----------------------------------------
    if (cur >= REF_INST) {
      if (cur == REF_EXIT) {
        cur = REF_NONE;
        if (exit_instance()) done = true;
        continue;
      }
      int code = cur - REF_INST;
      cur_last = (code & 1) != 0;
      if (COUNT) cnt.instances++;  // TLAS leaf entry (yocto_bvh.cpp:600-604)
      // (PRETEST: bit 0 of the code says "root box already tested" — the `last` flag it carries otherwise only matters to
      //  find_any queries, which the wide walk never serves)
      cur = enter_leaf_entry(code >> 1, PRETEST && cur_last);
      if (WIDE && abort) {
        best = Hit{HIT_ABORT, -1, 0, 0, 0, false};
        done = true;
      }
      if (cur_inst < 0 && find_any && cur_last && best.hit) done = true;
      continue;
    }
    const int first = cur & LEAF_FIRST_MASK, num = (cur >> 28) & 7;
    if (cur_inst < 0) {
      // TLAS leaf: instances are walked in order, each to completion
      // (yocto_bvh.cpp:600-609) → continuation entries in reverse, first one now.
      if constexpr (PRETEST) {
        // ... after the tmax-independent half of each instance's root-box test (pretest above): the survivors in
        // reverse, each with its t0 and the "tested" bit; the next pop applies the tmax-dependent half to the first of
        // them.  A leaf of ONE instance gains nothing from testing ahead (one dependent fetch either way): it is
        // entered as before, untested.
        if (num == 1) {
          cur = REF_INST + (first << 1);
          continue;
        }
        // ... and the FIRST survivor in the reference's order — the one the next pop would bring back — is entered right
        // here with the transformed ray its pretest computed: no second fetch of its record, no second transform_ray, no
        // three more divisions.  The instances are tested last to first; the latest survivor is held back and pushed only
        // when an earlier one turns up.  Its tmax-dependent half, t0 <= tmax * k, is taken now: nothing happens between
        // here and the pop that would have taken it.
        for (int k = num - 1; k >= 4; k--) push(REF_INST + ((first + k) << 1), 0);  // (never: leaves hold <= 4) untested
        bool  have = false, cirr = false;
        vec3f co = {0, 0, 0}, cd = {0, 0, 0}, cdinv = {0, 0, 0};
        float ct0 = 0;
        int   ck = 0, croot = REF_NONE, ckind = KIND_NONE;
#pragma unroll
        for (int k = 3; k >= 0; k--) {
          if (k >= num) continue;
          float4 m0, m1, m2, m3, m4;
          int4   m5;
          load_instance_record(sc.tinst_leaf, first + k, m0, m1, m2, m3, m4, m5);
          if (__float_as_int(m4.z) == REF_NONE) continue;
          frame3f inv = {{m0.x, m0.y, m0.z}, {m0.w, m1.x, m1.y}, {m1.z, m1.w, m2.x}, {m2.y, m2.z, m2.w}};
          vec3f   io   = transform_point(inv, wo);
          vec3f   id   = transform_vector(inv, wd);
          vec3f   idin = {1 / id.x, 1 / id.y, 1 / id.z};
          float   t0   = 0;
          const bool irr  = !ray_is_tame(io, idin, tmin);
          const bool pass = irr || slab<false>(io, idin, tmin, {m3.x, m3.y, m3.z}, {m3.w, m4.x, m4.y}, t0);
          if (!pass) continue;
          if (have) push(REF_INST + (((first + ck) << 1) | 1), ct0);
          have = true, cirr = irr, co = io, cd = id, cdinv = idin, ct0 = irr ? 0.0f : t0, ck = k;
          croot = __float_as_int(m4.z), ckind = __float_as_int(m4.w);
        }
        cur = REF_NONE;
        if (have && ct0 <= tmaxk) {
          if (cirr) {  // irregular at this instance's level: the caller redoes the ray binary (what enter() does)
            best = Hit{HIT_ABORT, -1, 0, 0, 0, false};
            done = true;
            continue;
          }
          const int4 m5 = reinterpret_cast<const int4*>(sc.tinst_leaf + (first + ck))[5];
          cur_inst      = m5.z;
          o = co, d = cd, dinv = cdinv;
          tame     = true;
          sign     = ((dinv.x < 0) ? 1 : 0) | ((dinv.y < 0) ? 2 : 0) | ((dinv.z < 0) ? 4 : 0);
          kind     = TRI == 1 ? KIND_TRIANGLES : ckind;
          if (TRI == 2 && kind != KIND_TRIANGLES) kind = KIND_QUADS;
          leafbias = m5.x;
          blas_hit = false;
          cur_last = true;
          push(REF_EXIT, 0);
          cur = croot;
        }
        continue;
      }
      for (int k = num - 1; k >= 1; k--) push(REF_INST + (((first + k) << 1) | (k == num - 1 ? 1 : 0)), 0);
      cur = num > 0 ? REF_INST + ((first << 1) | (num == 1 ? 1 : 0)) : REF_NONE;
      continue;
    }
    cur = REF_NONE;
    cnt.steps++;
    // BLAS leaf — yocto_bvh.cpp:505-545
    if (TRI == 1 || kind == KIND_TRIANGLES) {
      const float4* L = sc.leafdata + (leafbias + first * 3);
      // two triangles per round trip (the pool is padded, over-reads are ignored)
      for (int k0 = 0; k0 < num; k0 += 2) {
        float4 a0, b0, c0, a1, b1, c1;
        // (a walk of ONE instance — sample_lights_pdf's, every lane at the same light — usually has every lane in the same leaf)
        if (int uoff; SCALAR_LOADS && only_instance >= 0 && wave_uniform(leafbias + first * 3, uoff)) {
          const float4* Lu = sc.leafdata + (uoff + 3 * k0);
          a0 = ldc4(Lu, 0), b0 = ldc4(Lu, 1), c0 = ldc4(Lu, 2), a1 = ldc4(Lu, 3), b1 = ldc4(Lu, 4), c1 = ldc4(Lu, 5);
        } else {
          a0 = L[3 * k0], b0 = L[3 * k0 + 1], c0 = L[3 * k0 + 2];
          a1 = L[3 * k0 + 3], b1 = L[3 * k0 + 4], c1 = L[3 * k0 + 5];
        }
        if (COUNT) cnt.triangles++;
        auto h = intersect_triangle(o, d, tmin, tmax, {a0.x, a0.y, a0.z}, {a0.w, b0.x, b0.y}, {b0.z, b0.w, c0.x});
        if (h.hit) accept(__float_as_int(c0.y), h, leafbias + first * 3 + 3 * k0);
        if (k0 + 1 < num) {
          if (COUNT) cnt.triangles++;
          h = intersect_triangle(o, d, tmin, tmax, {a1.x, a1.y, a1.z}, {a1.w, b1.x, b1.y}, {b1.z, b1.w, c1.x});
          if (h.hit) accept(__float_as_int(c1.y), h, leafbias + first * 3 + 3 * k0 + 3);
        }
      }
    } else if (TRI != 1 && kind == KIND_QUADS) {
      const float4* L = sc.leafdata + (leafbias + first * 4);
      for (int k = 0; k < num; k++) {
        float4 a = L[4 * k], b = L[4 * k + 1], c = L[4 * k + 2], e4 = L[4 * k + 3];
        if (COUNT) cnt.quads++;
        auto h = intersect_quad(o, d, tmin, tmax, {a.x, a.y, a.z}, {a.w, b.x, b.y}, {b.z, b.w, c.x}, {c.y, c.z, c.w});
        if (h.hit) accept(__float_as_int(e4.x), h);
      }
    } else if (TRI == 0 && kind == KIND_LINES) {
      const float4* L = sc.leafdata + (leafbias + first * 3);
      for (int k = 0; k < num; k++) {
        float4 a = L[3 * k], b = L[3 * k + 1], c = L[3 * k + 2];
        if (COUNT) cnt.lines++;
        auto h = intersect_line(o, d, tmin, tmax, {a.x, a.y, a.z}, {a.w, b.x, b.y}, b.z, b.w);
        if (h.hit) accept(__float_as_int(c.x), h);
      }
    } else if (TRI == 0 && kind == KIND_POINTS) {
      const float4* L = sc.leafdata + (leafbias + first * 2);
      for (int k = 0; k < num; k++) {
        float4 a = L[2 * k], b = L[2 * k + 1];
        if (COUNT) cnt.points++;
        auto h = intersect_point(o, d, tmin, tmax, {a.x, a.y, a.z}, a.w);
        if (h.hit) accept(__float_as_int(b.x), h);
      }
    }
    // find_any early-out of intersect_shape_bvh — yocto_bvh.cpp:548
    if (find_any && blas_hit) {
      while (sp > 0 && pop().ref != REF_EXIT) {
      }
      if (exit_instance()) done = true;
    }
  }
  return best;
}

// The wide walk with MAJORITY-PHASE scheduling instead of while-while.  A lane's walk
// alternates node steps (W), leaf steps (L) and instance entries (E); with incoherent rays
// the lanes of a wavefront want different things at any moment and "descend until
// everybody holds a leaf" leaves most of them waiting.  Here every iteration first lets each
// lane do its cheap bookkeeping (pops, pop-time culling, instance exits, TLAS-leaf
// expansion) in a short private loop and then runs the ONE step kind most lanes are waiting
// for.  Same nodes, same order, same tests per ray as traverse<false, true, TRI> — only the
// interleaving between lanes differs — so the hit records are identical (tested).
template <int TRI>
YT_FN Hit traverse_phased(const DScene& sc, const ray3f& wray, int only_instance, Stack& st, Counters& cnt) {
  Hit best = {-1, -1, 0, 0, 0, false};
  const vec3f wo = wray.o, wd = wray.d;
  const float tmin  = wray.tmin;
  float       tmax  = wray.tmax;
  float       tmaxk = tmax * BBOX_K;
  const vec3f wdinv = {1 / wd.x, 1 / wd.y, 1 / wd.z};
  const int   wsign = ((wdinv.x < 0) ? 1 : 0) | ((wdinv.y < 0) ? 2 : 0) | ((wdinv.z < 0) ? 4 : 0);
  if (!ray_is_tame(wo, wdinv, tmin) || tmax != tmax) return Hit{HIT_ABORT, -1, 0, 0, 0, false};
  vec3f o = wo, d = wd, dinv = wdinv;
  int   sign = wsign, cur_inst = -1, kind = KIND_NONE, leafbias = 0;
  bool  done = false;
  lds_entry* const lds = st.lds;
  int             sp  = 0;
  StackEntry      spill[YT_SPILL];
  YT_STACK_OPS(YT_LDS_DEPTH, YT_SPILL)
  auto enter = [&](const DInstanceT* base, int idx, int inst) -> int {  // (as in traverse())
    float4 m0, m1, m2, m3, m4;
    int4   m5;
    load_instance_record(base, idx, m0, m1, m2, m3, m4, m5);
    int root = __float_as_int(m4.z);
    if (inst < 0) inst = m5.z;
    if (root == REF_NONE) return REF_NONE;
    frame3f inv = {{m0.x, m0.y, m0.z}, {m0.w, m1.x, m1.y}, {m1.z, m1.w, m2.x}, {m2.y, m2.z, m2.w}};
    vec3f   io   = transform_point(inv, wo);
    vec3f   id   = transform_vector(inv, wd);
    vec3f   idin = {1 / id.x, 1 / id.y, 1 / id.z};
    if (!ray_is_tame(io, idin, tmin)) {  // irregular at this instance's level: the caller redoes the ray binary
      best = Hit{HIT_ABORT, -1, 0, 0, 0, false};
      done = true;
      return REF_NONE;
    }
    float t0;
    bool  ok = slab<false>(io, idin, tmin, {m3.x, m3.y, m3.z}, {m3.w, m4.x, m4.y}, t0) && t0 <= tmaxk;
    if (!ok) return REF_NONE;
    o = io, d = id, dinv = idin;
    sign     = ((dinv.x < 0) ? 1 : 0) | ((dinv.y < 0) ? 2 : 0) | ((dinv.z < 0) ? 4 : 0);
    cur_inst = inst;
    kind     = TRI == 1 ? KIND_TRIANGLES : __float_as_int(m4.w);
    if (TRI == 2 && kind != KIND_TRIANGLES) kind = KIND_QUADS;
    leafbias = m5.x;
    push(REF_EXIT, 0);
    return root;
  };
  auto accept = [&](int element, const PrimHit& h, int leaf = 0) {
    best  = {cur_inst, element, h.u, h.v, h.t, true, leaf};
    tmax  = h.t;
    tmaxk = h.t * BBOX_K;
  };
  int cur = REF_NONE;
  if (only_instance >= 0) {
    cur = enter(sc.tinst, only_instance, only_instance);
    if (best.instance == HIT_ABORT) return best;
    if (cur_inst < 0) return best;
  } else {
    if (sc.tlas_ref == REF_NONE) return best;
    float t0;
    if (!(slab<false>(o, dinv, tmin, sc.tlas_bmin, sc.tlas_bmax, t0) && t0 <= tmaxk)) return best;
    cur = sc.tlas_ref;
  }
  while (true) {
    if (!done) {
      while (true) {
        if (cur == REF_NONE) {
          if (sp == 0) {
            done = true;
            break;
          }
          StackEntry e = pop();
          cur          = e.ref;
          if (e.ref < REF_INST && !(__int_as_float(e.t0) <= tmaxk)) cur = REF_NONE;  // culled at pop time
          continue;
        }
        if (cur == REF_EXIT) {  // back to the TLAS level: the world ray again
          cur = REF_NONE;
          o = wo, d = wd, dinv = wdinv, sign = wsign, cur_inst = -1;
          if (only_instance >= 0) {  // intersect_instance_bvh: the one instance has been walked
            done = true;
            break;
          }
          continue;
        }
        if (cur < 0 && cur_inst < 0) {
          // TLAS leaf: its instances in order, each to completion (yocto_bvh.cpp:600-609)
          const int first = cur & LEAF_FIRST_MASK, num = (cur >> 28) & 7;
          for (int k = num - 1; k >= 1; k--) push(REF_INST + (((first + k) << 1) | (k == num - 1 ? 1 : 0)), 0);
          cur = num > 0 ? REF_INST + ((first << 1) | (num == 1 ? 1 : 0)) : REF_NONE;
          continue;
        }
        break;
      }
    }
    const bool wantW = !done && (unsigned)cur < (unsigned)REF_INST;
    const bool wantL = !done && cur < 0;
    const bool wantE = !done && cur >= REF_INST;
    const int  nW = __popcll(__ballot(wantW)), nL = __popcll(__ballot(wantL)), nE = __popcll(__ballot(wantE));
    if (nW + nL + nE == 0) break;
#ifndef YT_PHASE_W  // node steps run while at least 1 / YT_PHASE_W of the waiting lanes want one
#define YT_PHASE_W 2
#endif
#ifndef YT_PHASE_L
#define YT_PHASE_L 2
#endif
    if (nW > 0 && nW * YT_PHASE_W >= nW + nL + nE) {
      if (wantW) {
        const float4* Qp = sc.wide + 8 * (int64_t)(cur & WIDE_REF_MASK);
        cnt.steps++;
        float ta, tb, tc, td;
#if YT_WIDE7
        float4 q0 = Qp[0], q1 = Qp[1], q2 = Qp[2], q3 = Qp[3], q4 = Qp[4], q5 = Qp[5], q6 = Qp[6];
        bool  fa = slab_rec7(o, dinv, tmin, q0, q1.x, q1.y, ta);
        bool  fb = slab_rec7(o, dinv, tmin, q2, q1.z, q1.w, tb);
        bool  fc = slab_rec7(o, dinv, tmin, q3, q4.x, q4.y, tc);
        bool  fd = slab_rec7(o, dinv, tmin, q5, q4.z, q4.w, td);
        const int wa = __float_as_int(q6.x), wb = __float_as_int(q6.y), wc = __float_as_int(q6.z), wd = __float_as_int(q6.w);
        int   ra = (fa && ta <= tmaxk) ? wa : REF_NONE;
        int   rb = (fb && tb <= tmaxk) ? wb : REF_NONE;
        int   rc = (fc && tc <= tmaxk) ? wc : REF_NONE;
        int   rd = (fd && td <= tmaxk) ? wd : REF_NONE;
        const bool hs = ((sign >> ((wa >> WIDE_AXIS_SHIFT) & 3)) & 1) != 0, ls = ((sign >> ((wb >> WIDE_AXIS_SHIFT) & 3)) & 1) != 0,
                   rs = ((sign >> ((wd >> WIDE_AXIS_SHIFT) & 3)) & 1) != 0;
#else
        float4 a0 = Qp[0], a1 = Qp[1], b0 = Qp[2], b1 = Qp[3], c0 = Qp[4], c1 = Qp[5], d0 = Qp[6], d1 = Qp[7];
        bool  fa = slab_rec(o, dinv, tmin, a0, a1, ta);
        bool  fb = slab_rec(o, dinv, tmin, b0, b1, tb);
        bool  fc = slab_rec(o, dinv, tmin, c0, c1, tc);
        bool  fd = slab_rec(o, dinv, tmin, d0, d1, td);
        int   ra = (fa && ta <= tmaxk) ? __float_as_int(a1.z) : REF_NONE;
        int   rb = (fb && tb <= tmaxk) ? __float_as_int(b1.z) : REF_NONE;
        int   rc = (fc && tc <= tmaxk) ? __float_as_int(c1.z) : REF_NONE;
        int   rd = (fd && td <= tmaxk) ? __float_as_int(d1.z) : REF_NONE;
        const int  axes = __float_as_int(a1.w);
        const bool hs = ((sign >> (axes & 3)) & 1) != 0, ls = ((sign >> ((axes >> 2) & 3)) & 1) != 0,
                   rs = ((sign >> ((axes >> 4) & 3)) & 1) != 0;
#endif
        int   l0r = ls ? rb : ra, l1r = ls ? ra : rb, r0r = rs ? rd : rc, r1r = rs ? rc : rd;
        float l0t = ls ? tb : ta, l1t = ls ? ta : tb, r0t = rs ? td : tc, r1t = rs ? tc : td;
        int   v0r = hs ? r0r : l0r, v1r = hs ? r1r : l1r, v2r = hs ? l0r : r0r, v3r = hs ? l1r : r1r;
        float v0t = hs ? r0t : l0t, v1t = hs ? r1t : l1t, v2t = hs ? l0t : r0t, v3t = hs ? l1t : r1t;
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
      }
    } else if (nL > 0 && nL * YT_PHASE_L >= nL + nE) {
      if (wantL) {
        const int first = cur & LEAF_FIRST_MASK, num = (cur >> 28) & 7;
        cur = REF_NONE;
        cnt.steps++;
        if (TRI == 1 || kind == KIND_TRIANGLES) {
          const float4* L = sc.leafdata + (leafbias + first * 3);
          for (int k0 = 0; k0 < num; k0 += 2) {
            float4 a0 = L[3 * k0], b0 = L[3 * k0 + 1], c0 = L[3 * k0 + 2];
            float4 a1 = L[3 * k0 + 3], b1 = L[3 * k0 + 4], c1 = L[3 * k0 + 5];
            auto h = intersect_triangle(o, d, tmin, tmax, {a0.x, a0.y, a0.z}, {a0.w, b0.x, b0.y}, {b0.z, b0.w, c0.x});
            if (h.hit) accept(__float_as_int(c0.y), h, leafbias + first * 3 + 3 * k0);
            if (k0 + 1 < num) {
              h = intersect_triangle(o, d, tmin, tmax, {a1.x, a1.y, a1.z}, {a1.w, b1.x, b1.y}, {b1.z, b1.w, c1.x});
              if (h.hit) accept(__float_as_int(c1.y), h, leafbias + first * 3 + 3 * k0 + 3);
            }
          }
        } else if (TRI != 1 && kind == KIND_QUADS) {
          const float4* L = sc.leafdata + (leafbias + first * 4);
          for (int k = 0; k < num; k++) {
            float4 a = L[4 * k], b = L[4 * k + 1], c = L[4 * k + 2], e4 = L[4 * k + 3];
            auto h = intersect_quad(o, d, tmin, tmax, {a.x, a.y, a.z}, {a.w, b.x, b.y}, {b.z, b.w, c.x}, {c.y, c.z, c.w});
            if (h.hit) accept(__float_as_int(e4.x), h);
          }
        } else if (TRI == 0 && kind == KIND_LINES) {
          const float4* L = sc.leafdata + (leafbias + first * 3);
          for (int k = 0; k < num; k++) {
            float4 a = L[3 * k], b = L[3 * k + 1], c = L[3 * k + 2];
            auto h = intersect_line(o, d, tmin, tmax, {a.x, a.y, a.z}, {a.w, b.x, b.y}, b.z, b.w);
            if (h.hit) accept(__float_as_int(c.x), h);
          }
        } else if (TRI == 0 && kind == KIND_POINTS) {
          const float4* L = sc.leafdata + (leafbias + first * 2);
          for (int k = 0; k < num; k++) {
            float4 a = L[2 * k], b = L[2 * k + 1];
            auto h = intersect_point(o, d, tmin, tmax, {a.x, a.y, a.z}, a.w);
            if (h.hit) accept(__float_as_int(b.x), h);
          }
        }
      }
    } else {
      if (wantE) {
        cur = enter(sc.tinst_leaf, (cur - REF_INST) >> 1, -1);
      }
    }
  }
  return best;
}

// The production entry: the wide walk, and the binary walk for the rays it declines
// (irregular at world or instance level, find_any) — the same hit record either way.
#ifdef YT_OWN_TREE  // the own-tree unit (yt_owntree.hip, fastmath = 2): every production walk is yt_own.h's
template <int TRI, bool PHASED = false>
YT_FN Hit traverse_own(const DScene& sc, const ray3f& wray, int only_instance, Stack& st, Counters& cnt);
#endif
template <bool COUNT, bool WIDE, int TRI = 0, bool PHASED = false>
YT_FN Hit traverse_any(const DScene& sc, const ray3f& wray, int only_instance, bool find_any, Stack& st,
    Counters& cnt) {
#ifdef YT_OWN_TREE
  if constexpr (WIDE && !COUNT) {
    if (!find_any) return traverse_own<TRI, PHASED>(sc, wray, only_instance, st, cnt);
  }
#endif
  if constexpr (WIDE && !COUNT) {
    if constexpr (PHASED) {
      if (!find_any) {
        Hit h = traverse_phased<TRI>(sc, wray, only_instance, st, cnt);
        if (h.instance != HIT_ABORT) return h;
      }
    } else {
      Hit h = traverse<false, true, TRI>(sc, wray, only_instance, find_any, st, cnt);
      if (h.instance != HIT_ABORT) return h;
    }
  }
  return traverse<COUNT, false, TRI>(sc, wray, only_instance, find_any, st, cnt);
}

}  // namespace yt
