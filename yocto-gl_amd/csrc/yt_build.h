// yt_build.h — host-side preparation that the reference also does on the host:
//   * make_scene_bvh / make_shape_bvh / make_bvh (+ split_middle, split_sah)
//       libs/yocto/yocto_bvh.cpp:108-164, 202-302, 321-396
//   * make_trace_lights                     libs/yocto/yocto_trace.cpp:1528-1581
//   * per-pixel rng seeding, image size rule  libs/yocto/yocto_trace.cpp:1495-1515
// The BVH builder reproduces the reference's node order and `primitives`
// permutation bit-for-bit (same DFS with an explicit stack, children allocated
// adjacently, right child processed first, std::partition from the same
// libstdc++), so hit indices stay identical.  Compiled with -ffp-contract=off.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <utility>
#include <vector>

#include "../../include/ythip.h"

namespace ythost {

constexpr float flt_max = std::numeric_limits<float>::max();
constexpr float flt_min = std::numeric_limits<float>::lowest();
constexpr float pif     = (float)3.14159265358979323846;

struct v3 {
  float x, y, z;
};
inline float fmin_(float a, float b) { return (a < b) ? a : b; }  // yocto_math.h:1046
inline float fmax_(float a, float b) { return (a > b) ? a : b; }  // yocto_math.h:1047
inline v3    operator+(v3 a, v3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline v3    operator-(v3 a, v3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline v3    operator*(v3 a, float b) { return {a.x * b, a.y * b, a.z * b}; }
inline v3    operator/(v3 a, float b) { return {a.x / b, a.y / b, a.z / b}; }
inline v3    operator-(v3 a, float b) { return {a.x - b, a.y - b, a.z - b}; }
inline v3    operator+(v3 a, float b) { return {a.x + b, a.y + b, a.z + b}; }
inline v3    vmin(v3 a, v3 b) { return {fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)}; }
inline v3    vmax(v3 a, v3 b) { return {fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)}; }
inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline v3    cross(v3 a, v3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float length(v3 a) { return std::sqrt(dot(a, a)); }
inline float at(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

struct bbox {
  v3 min = {flt_max, flt_max, flt_max};
  v3 max = {flt_min, flt_min, flt_min};
};
inline bbox merge(const bbox& a, v3 b) { return {vmin(a.min, b), vmax(a.max, b)}; }
inline bbox merge(const bbox& a, const bbox& b) { return {vmin(a.min, b.min), vmax(a.max, b.max)}; }
inline v3   center(const bbox& a) { return (a.min + a.max) / 2; }

struct tree {
  std::vector<ythip_bvh_node> nodes;
  std::vector<int32_t>        prims;
};

// split_middle — yocto_bvh.cpp:202-232
inline std::pair<int, int> split_middle(std::vector<int32_t>& primitives, const std::vector<v3>& centers,
    int start, int end) {
  auto cbbox = bbox{};
  for (auto i = start; i < end; i++) cbbox = merge(cbbox, centers[primitives[i]]);
  auto csize = cbbox.max - cbbox.min;
  if (csize.x == 0 && csize.y == 0 && csize.z == 0) return {(start + end) / 2, 0};
  auto axis = 0;
  if (csize.x >= csize.y && csize.x >= csize.z) axis = 0;
  if (csize.y >= csize.x && csize.y >= csize.z) axis = 1;
  if (csize.z >= csize.x && csize.z >= csize.y) axis = 2;
  auto split  = at(center(cbbox), axis);
  auto middle = (int)(std::partition(primitives.data() + start, primitives.data() + end,
                          [axis, split, &centers](int primitive) { return at(centers[primitive], axis) < split; }) -
                      primitives.data());
  if (middle == start || middle == end) return {(start + end) / 2, axis};
  return {middle, axis};
}

// split_sah — yocto_bvh.cpp:108-164
inline std::pair<int, int> split_sah(std::vector<int32_t>& primitives, const std::vector<bbox>& bboxes,
    const std::vector<v3>& centers, int start, int end) {
  auto cbbox = bbox{};
  for (auto i = start; i < end; i++) cbbox = merge(cbbox, centers[primitives[i]]);
  auto csize = cbbox.max - cbbox.min;
  if (csize.x == 0 && csize.y == 0 && csize.z == 0) return {(start + end) / 2, 0};
  auto      axis      = 0;
  const int nbins     = 16;
  auto      split     = 0.0f;
  auto      min_cost  = flt_max;
  auto      bbox_area = [](const bbox& b) {
    auto size = b.max - b.min;
    return 1e-12f + 2 * size.x * size.y + 2 * size.x * size.z + 2 * size.y * size.z;
  };
  for (auto saxis = 0; saxis < 3; saxis++) {
    for (auto b = 1; b < nbins; b++) {
      auto bsplit    = at(cbbox.min, saxis) + b * at(csize, saxis) / nbins;
      auto left_bbox = bbox{}, right_bbox = bbox{};
      auto left_nprims = 0, right_nprims = 0;
      for (auto i = start; i < end; i++) {
        if (at(centers[primitives[i]], saxis) < bsplit) {
          left_bbox = merge(left_bbox, bboxes[primitives[i]]);
          left_nprims += 1;
        } else {
          right_bbox = merge(right_bbox, bboxes[primitives[i]]);
          right_nprims += 1;
        }
      }
      auto cost = 1 + left_nprims * bbox_area(left_bbox) / bbox_area(cbbox) +
                  right_nprims * bbox_area(right_bbox) / bbox_area(cbbox);
      if (cost < min_cost) {
        min_cost = cost;
        split    = bsplit;
        axis     = saxis;
      }
    }
  }
  auto middle = (int)(std::partition(primitives.data() + start, primitives.data() + end,
                          [axis, split, &centers](int primitive) { return at(centers[primitive], axis) < split; }) -
                      primitives.data());
  if (middle == start || middle == end) return {(start + end) / 2, axis};
  return {middle, axis};
}

// make_bvh — yocto_bvh.cpp:238-302
inline tree make_bvh(const std::vector<bbox>& bboxes, bool highquality) {
  const int bvh_max_prims = 4;
  tree      bvh;
  bvh.nodes.reserve(bboxes.size() * 2);
  bvh.prims.resize(bboxes.size());
  for (size_t idx = 0; idx < bboxes.size(); idx++) bvh.prims[idx] = (int)idx;
  auto centers = std::vector<v3>(bboxes.size());
  for (size_t idx = 0; idx < bboxes.size(); idx++) centers[idx] = center(bboxes[idx]);

  struct item {
    int nodeid, start, end;
  };
  auto stack = std::vector<item>{{0, 0, (int)bboxes.size()}};
  bvh.nodes.emplace_back();
  std::memset(&bvh.nodes.back(), 0, sizeof(ythip_bvh_node));
  while (!stack.empty()) {
    auto [nodeid, start, end] = stack.back();
    stack.pop_back();
    auto nb = bbox{};
    for (auto i = start; i < end; i++) nb = merge(nb, bboxes[bvh.prims[i]]);
    ythip_bvh_node node;
    std::memset(&node, 0, sizeof(node));
    node.bbox_min[0] = nb.min.x, node.bbox_min[1] = nb.min.y, node.bbox_min[2] = nb.min.z;
    node.bbox_max[0] = nb.max.x, node.bbox_max[1] = nb.max.y, node.bbox_max[2] = nb.max.z;
    if (end - start > bvh_max_prims) {
      auto [mid, axis] = highquality ? split_sah(bvh.prims, bboxes, centers, start, end)
                                     : split_middle(bvh.prims, centers, start, end);
      node.internal = 1;
      node.axis     = (int8_t)axis;
      node.num      = 2;
      node.start    = (int)bvh.nodes.size();
      bvh.nodes.emplace_back();
      bvh.nodes.emplace_back();
      std::memset(&bvh.nodes[bvh.nodes.size() - 2], 0, 2 * sizeof(ythip_bvh_node));
      stack.push_back({node.start + 0, start, mid});
      stack.push_back({node.start + 1, mid, end});
    } else {
      node.internal = 0;
      node.num      = (int16_t)(end - start);
      node.start    = start;
    }
    bvh.nodes[nodeid] = node;
  }
  return bvh;
}

inline v3 ld3(const float* p, int64_t i) { return {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }

// element kind in the BVH's dispatch order points→lines→triangles→quads
// (yocto_bvh.cpp:327-355, 505-545)
inline int kind_bvh(const ythip_shape& s) {
  if (s.num_points) return 1;
  if (s.num_lines) return 2;
  if (s.num_triangles) return 3;
  if (s.num_quads) return 4;
  return 0;
}
// element kind in eval_position's order triangles→quads→lines→points
// (yocto_scene.cpp:288-311)
inline int kind_eval(const ythip_shape& s) {
  if (s.num_triangles) return 3;
  if (s.num_quads) return 4;
  if (s.num_lines) return 2;
  if (s.num_points) return 1;
  return 0;
}

// the primitive bounds of make_shape_bvh / update_shape_bvh — yocto_bvh.cpp:321-357, 398-428
// (yocto_geometry.h:475-498)
inline std::vector<bbox> shape_prim_bboxes(const ythip_scene& sc, const ythip_shape& s) {
  auto        bboxes = std::vector<bbox>{};
  const auto* P      = sc.positions + 3 * s.positions_offset;
  const auto* R      = sc.radius ? sc.radius + s.radius_offset : nullptr;
  if (s.num_points) {
    bboxes.resize(s.num_points);
    const auto* E = sc.points + s.points_offset;
    for (auto i = 0; i < s.num_points; i++) {
      auto p = ld3(P, E[i]);
      auto r = R[E[i]];
      bboxes[i] = {vmin(p - r, p + r), vmax(p - r, p + r)};
    }
  } else if (s.num_lines) {
    bboxes.resize(s.num_lines);
    const auto* E = sc.lines + 2 * s.lines_offset;
    for (auto i = 0; i < s.num_lines; i++) {
      auto p0 = ld3(P, E[2 * i]), p1 = ld3(P, E[2 * i + 1]);
      auto r0 = R[E[2 * i]], r1 = R[E[2 * i + 1]];
      bboxes[i] = {vmin(p0 - r0, p1 - r1), vmax(p0 + r0, p1 + r1)};
    }
  } else if (s.num_triangles) {
    bboxes.resize(s.num_triangles);
    const auto* E = sc.triangles + 3 * s.triangles_offset;
    for (auto i = 0; i < s.num_triangles; i++) {
      auto p0 = ld3(P, E[3 * i]), p1 = ld3(P, E[3 * i + 1]), p2 = ld3(P, E[3 * i + 2]);
      bboxes[i] = {vmin(p0, vmin(p1, p2)), vmax(p0, vmax(p1, p2))};
    }
  } else if (s.num_quads) {
    bboxes.resize(s.num_quads);
    const auto* E = sc.quads + 4 * s.quads_offset;
    for (auto i = 0; i < s.num_quads; i++) {
      auto p0 = ld3(P, E[4 * i]), p1 = ld3(P, E[4 * i + 1]), p2 = ld3(P, E[4 * i + 2]),
           p3 = ld3(P, E[4 * i + 3]);
      bboxes[i] = {vmin(p0, vmin(p1, vmin(p2, p3))), vmax(p0, vmax(p1, vmax(p2, p3)))};
    }
  }
  return bboxes;
}

// make_shape_bvh — yocto_bvh.cpp:321-362
inline tree make_shape_bvh(const ythip_scene& sc, const ythip_shape& s, bool highquality) {
  // NB: an element-less shape still gets a one-node tree (empty leaf, invalid bbox), as in the reference
  return make_bvh(shape_prim_bboxes(sc, s), highquality);
}

// refit_bvh — yocto_bvh.cpp:305-319: boxes bottom-up (children sit behind their parent
// in the node array), topology and `primitives` untouched; the merge order is the
// reference's (it decides the sign of a zero face).
inline void refit_bvh(ythip_bvh_node* nodes, int64_t num_nodes, const int32_t* prims, const std::vector<bbox>& bboxes) {
  auto box_of = [](const ythip_bvh_node& n) {
    return bbox{{n.bbox_min[0], n.bbox_min[1], n.bbox_min[2]}, {n.bbox_max[0], n.bbox_max[1], n.bbox_max[2]}};
  };
  for (auto nodeid = num_nodes - 1; nodeid >= 0; nodeid--) {
    auto& node = nodes[nodeid];
    auto  box  = bbox{};
    if (node.internal) {
      for (auto idx = 0; idx < 2; idx++) box = merge(box, box_of(nodes[node.start + idx]));
    } else {
      for (auto idx = 0; idx < node.num; idx++) box = merge(box, bboxes[prims[node.start + idx]]);
    }
    node.bbox_min[0] = box.min.x, node.bbox_min[1] = box.min.y, node.bbox_min[2] = box.min.z;
    node.bbox_max[0] = box.max.x, node.bbox_max[1] = box.max.y, node.bbox_max[2] = box.max.z;
  }
}

// transform_point(frame, p) — yocto_math.h:2263
inline v3 transform_point(const ythip_frame& f, v3 b) {
  v3 fx = {f.x[0], f.x[1], f.x[2]}, fy = {f.y[0], f.y[1], f.y[2]}, fz = {f.z[0], f.z[1], f.z[2]},
     fo = {f.o[0], f.o[1], f.o[2]};
  return fx * b.x + fy * b.y + fz * b.z + fo;
}
// transform_bbox(frame, bbox) — yocto_geometry.h:453-465
inline bbox transform_bbox(const ythip_frame& a, const bbox& b) {
  v3 corners[8] = {{b.min.x, b.min.y, b.min.z}, {b.min.x, b.min.y, b.max.z}, {b.min.x, b.max.y, b.min.z},
      {b.min.x, b.max.y, b.max.z}, {b.max.x, b.min.y, b.min.z}, {b.max.x, b.min.y, b.max.z},
      {b.max.x, b.max.y, b.min.z}, {b.max.x, b.max.y, b.max.z}};
  auto xformed  = bbox{};
  for (auto& corner : corners) xformed = merge(xformed, transform_point(a, corner));
  return xformed;
}

// resize() without value-initialisation: the slices of device-built trees are only
// written when somebody downloads the tree (20 MB for 1M triangles)
template <typename T>
struct noinit_allocator : std::allocator<T> {
  template <typename U>
  struct rebind {
    using other = noinit_allocator<U>;
  };
  template <typename U>
  void construct(U* p) noexcept {
    ::new ((void*)p) U;
  }
  template <typename U, typename... Args>
  void construct(U* p, Args&&... args) {
    ::new ((void*)p) U(std::forward<Args>(args)...);
  }
};
struct flat_bvh {
  std::vector<int64_t>                                          node_offset, prim_offset;
  std::vector<ythip_bvh_node, noinit_allocator<ythip_bvh_node>> nodes;
  std::vector<int32_t, noinit_allocator<int32_t>>               prims;
};

// make_scene_bvh — yocto_bvh.cpp:364-396
inline flat_bvh make_scene_bvh(const ythip_scene& sc, bool highquality) {
  flat_bvh out;
  auto     roots = std::vector<bbox>(sc.num_shapes);
  auto     empty = std::vector<char>(sc.num_shapes, 1);
  auto     push  = [&](const tree& t) {
    out.node_offset.push_back((int64_t)out.nodes.size());
    out.prim_offset.push_back((int64_t)out.prims.size());
    out.nodes.insert(out.nodes.end(), t.nodes.begin(), t.nodes.end());
    out.prims.insert(out.prims.end(), t.prims.begin(), t.prims.end());
  };
  for (auto k = 0; k < sc.num_shapes; k++) {
    auto t = make_shape_bvh(sc, sc.shapes[k], highquality);
    if (!t.nodes.empty()) {
      empty[k]     = 0;
      auto& n      = t.nodes[0];
      roots[k].min = {n.bbox_min[0], n.bbox_min[1], n.bbox_min[2]};
      roots[k].max = {n.bbox_max[0], n.bbox_max[1], n.bbox_max[2]};
    }
    push(t);
  }
  auto bboxes = std::vector<bbox>(sc.num_instances);
  for (auto k = 0; k < sc.num_instances; k++) {
    auto& inst = sc.instances[k];
    bboxes[k]  = empty[inst.shape] ? bbox{} : transform_bbox(inst.frame, roots[inst.shape]);
  }
  // make_bvh on an empty list yields a single empty leaf node in the reference
  // (nodes.emplace_back() before the loop), keep that.
  push(make_bvh(bboxes, highquality));
  out.node_offset.push_back((int64_t)out.nodes.size());
  out.prim_offset.push_back((int64_t)out.prims.size());
  return out;
}

// update_scene_bvh — yocto_bvh.cpp:434-451 on the flat layout: refit the listed
// shapes' trees from the scene's current vertices, then the instance tree from every
// instance's current frame.
inline void update_scene_bvh(flat_bvh& b, const ythip_scene& sc, const int32_t* updated_shapes, int num_shapes) {
  for (auto k = 0; k < num_shapes; k++) {
    auto s = updated_shapes[k];
    refit_bvh(b.nodes.data() + b.node_offset[s], b.node_offset[s + 1] - b.node_offset[s],
        b.prims.data() + b.prim_offset[s], shape_prim_bboxes(sc, sc.shapes[s]));
  }
  auto bboxes = std::vector<bbox>(sc.num_instances);
  for (auto k = 0; k < sc.num_instances; k++) {
    auto& inst = sc.instances[k];
    auto  s    = inst.shape;
    if (b.node_offset[s + 1] == b.node_offset[s]) continue;
    auto& n   = b.nodes[b.node_offset[s]];
    bboxes[k] = transform_bbox(inst.frame, bbox{{n.bbox_min[0], n.bbox_min[1], n.bbox_min[2]},
                                               {n.bbox_max[0], n.bbox_max[1], n.bbox_max[2]}});
  }
  auto t = sc.num_shapes;
  refit_bvh(b.nodes.data() + b.node_offset[t], b.node_offset[t + 1] - b.node_offset[t],
      b.prims.data() + b.prim_offset[t], bboxes);
}

// inverse(frame, non_rigid = true) — yocto_math.h:2114-2118, 1967-1974
inline void inverse_frame_nonrigid(const ythip_frame& f, float out[12]) {
  v3   ax = {f.x[0], f.x[1], f.x[2]}, ay = {f.y[0], f.y[1], f.y[2]}, az = {f.z[0], f.z[1], f.z[2]},
     ao   = {f.o[0], f.o[1], f.o[2]};
  auto det = dot(ax, cross(ay, az));
  v3   c0 = cross(ay, az), c1 = cross(az, ax), c2 = cross(ax, ay);
  // adjoint = transpose({c0, c1, c2}); columns:
  v3   tx = {c0.x, c1.x, c2.x}, ty = {c0.y, c1.y, c2.y}, tz = {c0.z, c1.z, c2.z};
  auto s  = 1 / det;
  v3   mx = tx * s, my = ty * s, mz = tz * s;
  // -(minv * o) with mat*vec = x*b.x + y*b.y + z*b.z
  v3 mo  = mx * ao.x + my * ao.y + mz * ao.z;
  out[0] = mx.x, out[1] = mx.y, out[2] = mx.z;
  out[3] = my.x, out[4] = my.y, out[5] = my.z;
  out[6] = mz.x, out[7] = mz.y, out[8] = mz.z;
  out[9] = -mo.x, out[10] = -mo.y, out[11] = -mo.z;
}
// inverse(frame) rigid — transpose(rotation), -(minv * o)
inline void inverse_frame_rigid(const ythip_frame& f, float out[12]) {
  v3 mx = {f.x[0], f.y[0], f.z[0]}, my = {f.x[1], f.y[1], f.z[1]}, mz = {f.x[2], f.y[2], f.z[2]};
  v3 ao = {f.o[0], f.o[1], f.o[2]};
  v3 mo = mx * ao.x + my * ao.y + mz * ao.z;
  out[0] = mx.x, out[1] = mx.y, out[2] = mx.z;
  out[3] = my.x, out[4] = my.y, out[5] = my.z;
  out[6] = mz.x, out[7] = mz.y, out[8] = mz.z;
  out[9] = -mo.x, out[10] = -mo.y, out[11] = -mo.z;
}

// make_trace_lights — yocto_trace.cpp:1528-1581
struct flat_lights {
  std::vector<ythip_light> lights;
  std::vector<float>       cdf;
};
inline float triangle_area(v3 p0, v3 p1, v3 p2) { return length(cross(p1 - p0, p2 - p0)) / 2; }
inline flat_lights make_trace_lights(const ythip_scene& sc) {
  flat_lights out;
  for (auto handle = 0; handle < sc.num_instances; handle++) {
    auto& instance = sc.instances[handle];
    auto& material = sc.materials[instance.material];
    if (material.emission[0] == 0 && material.emission[1] == 0 && material.emission[2] == 0) continue;
    auto& shape = sc.shapes[instance.shape];
    if (shape.num_triangles == 0 && shape.num_quads == 0) continue;
    ythip_light light = {handle, YTHIP_INVALIDID, (int64_t)out.cdf.size(), 0, 0};
    const auto* P     = sc.positions + 3 * shape.positions_offset;
    auto        cdf   = std::vector<float>{};
    if (shape.num_triangles) {
      cdf.resize(shape.num_triangles);
      const auto* E = sc.triangles + 3 * shape.triangles_offset;
      for (auto idx = 0; idx < shape.num_triangles; idx++) {
        cdf[idx] = triangle_area(ld3(P, E[3 * idx]), ld3(P, E[3 * idx + 1]), ld3(P, E[3 * idx + 2]));
        if (idx != 0) cdf[idx] += cdf[idx - 1];
      }
    }
    if (shape.num_quads) {
      cdf.assign(shape.num_quads, 0.0f);
      const auto* E = sc.quads + 4 * shape.quads_offset;
      for (auto idx = 0; idx < shape.num_quads; idx++) {
        auto p0 = ld3(P, E[4 * idx]), p1 = ld3(P, E[4 * idx + 1]), p2 = ld3(P, E[4 * idx + 2]),
             p3 = ld3(P, E[4 * idx + 3]);
        cdf[idx] = triangle_area(p0, p1, p3) + triangle_area(p2, p3, p1);
        if (idx != 0) cdf[idx] += cdf[idx - 1];
      }
    }
    light.cdf_count = (int)cdf.size();
    out.cdf.insert(out.cdf.end(), cdf.begin(), cdf.end());
    out.lights.push_back(light);
  }
  for (auto handle = 0; handle < sc.num_environments; handle++) {
    auto& environment = sc.environments[handle];
    if (environment.emission[0] == 0 && environment.emission[1] == 0 && environment.emission[2] == 0) continue;
    ythip_light light = {YTHIP_INVALIDID, handle, (int64_t)out.cdf.size(), 0, 0};
    if (environment.emission_tex != YTHIP_INVALIDID) {
      auto& texture = sc.textures[environment.emission_tex];
      auto  n       = (size_t)texture.width * texture.height;
      auto  cdf     = std::vector<float>(n);
      for (size_t idx = 0; idx < n; idx++) {
        auto  i = (int)idx % texture.width, j = (int)idx / texture.width;
        auto  th = (j + 0.5f) * pif / texture.height;
        float value[4];
        // lookup_texture(texture, i, j) with as_linear = false — yocto_scene.cpp:111-124
        if (texture.is_float) {
          std::memcpy(value, sc.pixelsf + 4 * (texture.offset + (int64_t)j * texture.width + i), 16);
        } else {
          auto b = sc.pixelsb + 4 * (texture.offset + (int64_t)j * texture.width + i);
          for (auto c = 0; c < 4; c++) value[c] = b[c] / 255.0f;
        }
        // max(vec4f) = max(max(max(x,y),z),w) — yocto_math.h
        auto mx  = fmax_(fmax_(fmax_(value[0], value[1]), value[2]), value[3]);
        cdf[idx] = mx * std::sin(th);
        if (idx != 0) cdf[idx] += cdf[idx - 1];
      }
      light.cdf_count = (int)cdf.size();
      out.cdf.insert(out.cdf.end(), cdf.begin(), cdf.end());
    }
    out.lights.push_back(light);
  }
  return out;
}

// PCG32 (host) — yocto_sampling.h:187-208
struct rng_state {
  uint64_t state, inc;
};
inline uint32_t advance_rng(rng_state& rng) {
  uint64_t oldstate   = rng.state;
  rng.state           = oldstate * 6364136223846793005ULL + rng.inc;
  uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
  uint32_t rot        = (uint32_t)(oldstate >> 59u);
  return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}
inline rng_state make_rng(uint64_t seed, uint64_t seq = 1) {
  rng_state rng;
  rng.state = 0U;
  rng.inc   = (seq << 1u) | 1u;
  advance_rng(rng);
  rng.state += seed;
  advance_rng(rng);
  return rng;
}
// make_trace_state seeding — yocto_trace.cpp:1512-1515
inline void make_rngs(uint64_t seed, int64_t n, uint64_t* out) {
  auto rng_ = make_rng(1301081);
  for (int64_t k = 0; k < n; k++) {
    // rand1i(rng_, 1 << 31): `_advance_rng(rng) % n` with int n = INT_MIN →
    // converted to uint32 2147483648 for the modulo; result converted to int
    auto r   = (int)(advance_rng(rng_) % (uint32_t)(1u << 31));
    auto rng = make_rng(seed, (uint64_t)(int64_t)(r / 2 + 1));
    out[2 * k]     = rng.state;
    out[2 * k + 1] = rng.inc;
  }
}

}  // namespace ythost
