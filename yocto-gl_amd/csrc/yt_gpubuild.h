// yt_gpubuild.h — on-device make_bvh: interface between
// ythip.hip (the C ABI) and yt_gpubuild.hip (the builder / baker kernels).
//
// The builder reproduces the reference's make_bvh + split_middle
// (libs/yocto/yocto_bvh.cpp:202-302) node for node and bit for bit — same node
// order, same `primitives` permutation, same boxes — so that every hit record
// stays identical to the host-built tree's.  See yt_gpubuild.hip for how the
// serial algorithm (explicit-stack DFS + std::partition) is restated as a
// level-synchronous parallel one.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/ythip.h"

namespace ytgpu {

// A shape's bvh_tree (yocto_shape.h:486-489) in the reference's layout, resident
// on the device.
struct DeviceTree {
  ythip_bvh_node* nodes      = nullptr;  // num_nodes records, reference node order
  int32_t*        prims      = nullptr;  // num_prims element ids (bvh_tree::primitives)
  int64_t         num_nodes  = 0;
  int64_t         num_prims  = 0;
  int             depth      = 0;        // levels of the tree
  float           build_ms   = 0;        // device time of the build (hipEvents)
};

enum { BUILD_OK = 0, BUILD_FALLBACK = 1, BUILD_ERROR = 2 };

// make_shape_bvh (yocto_bvh.cpp:321-362) for one shape: split_middle, or split_sah
// (yocto_bvh.cpp:108-164, trace_params::highqualitybvh) with `highquality`.
// `elems` / `positions` / `radius` are DEVICE pointers already offset to the
// shape's first element / vertex; kind is the BVH dispatch kind (1 points,
// 2 lines, 3 triangles, 4 quads).  Returns BUILD_FALLBACK (nothing allocated)
// when the tree cannot be reproduced bit for bit on the device (signed-zero
// ties between box faces — see the .hip) and the caller must use the host
// builder, BUILD_ERROR on a HIP failure (message in *err).
// kind 0: the primitives are boxes — `positions` holds {min.xyz, max.xyz} per box, `elems` and
// `radius` are unused: make_bvh over instance bounds, the instance tree (yocto_bvh.cpp:381-393).
int build_shape_tree(hipStream_t stream, int kind, const int32_t* elems, const float* positions,
    const float* radius, int64_t num_prims, bool highquality, DeviceTree* out, std::string* err);
void free_tree(DeviceTree* tree);

// Device bake of one BLAS into the traversal layout of yt_bvh.h (what
// bake_bvh() in ythip.hip does on the host for host-built trees):
//   pairs    + 4 * pair_base    sibling-pair records of this tree's internal nodes
//   quads    + 8 * pair_base    grandchildren records (same ids)
//   leafdata + leaf_base        pre-gathered primitives in leaf order
// prim_base = global index of the tree's first primitive (leaf refs are global).
// Writes the root's {bbox, ref} to root_out (host pointer, 7 floats: bmin, bmax, ref bits).
int bake_shape_tree(hipStream_t stream, const DeviceTree& tree, int kind, const int32_t* elems,
    const float* positions, const float* radius, int64_t pair_base, int64_t prim_base, int64_t leaf_base,
    float4* pairs, float4* quads, float4* leafdata, float* root_out, std::string* err);

// Batched device bake of HOST-resident trees (built by yt_build.h or uploaded by the caller): all of
// them in one set of launches, whatever their number (a scene of thousands of small shapes bakes in
// the time of one large one).  The caller uploads their nodes and primitives as compact arrays
// (tree after tree) and a table that says where each tree starts and where its records go.
struct HostTreeDesc {
  long long    node_off;       // first node of the tree in the compact node array
  long long    prim_off;       // first primitive of the tree in the compact primitive array
  long long    pair_base;      // id of the tree's first sibling-pair / quad record
  long long    ref_prim_base;  // added to a leaf's `start` in its ref: the tree's global primitive offset (0: instance tree)
  long long    leaf_base;      // float4 index of the tree's leaf data
  int          kind;           // 0 instance tree (no leaf data), 1 points, 2 lines, 3 triangles, 4 quads
  int          pad_;
  const int*   elems;          // the shape's elements / positions / radii on the device
  const float* positions;
  const float* radius;
};
// d_table: `ntrees` descriptors on the device, ascending node_off / prim_off.
int bake_host_trees(hipStream_t stream, const ythip_bvh_node* d_nodes, long long num_nodes, const int32_t* d_prims,
    long long num_prims, const HostTreeDesc* d_table, int ntrees, float4* pairs, float4* quads, float4* leafdata,
    std::string* err);

// refit_bvh (yocto_bvh.cpp:305-319) of a resident tree after the shape's vertices moved
// (update_shape_bvh, yocto_bvh.cpp:398-431): every box recomputed bottom-up, topology
// and `prims` untouched, the same boxes bit for bit as the host sweep.  tree.build_ms
// = device time of the refit.
int refit_shape_tree(hipStream_t stream, DeviceTree& tree, int kind, const int32_t* elems, const float* positions,
    const float* radius, std::string* err);

}  // namespace ytgpu
