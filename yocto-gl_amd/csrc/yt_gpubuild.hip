// yt_gpubuild.hip — make_bvh on the device.
//
// Restates libs/yocto/yocto_bvh.cpp:202-302 (make_bvh + split_middle), :108-164
// (split_sah, the `highqualitybvh` build: 16 bins x 3 axes per node, see k_bin) and
// :321-362 (make_shape_bvh's primitive bounds) so that the result is the SAME
// tree the reference builds — node order, `primitives` permutation, boxes —
// which is what keeps hit records bit-identical.
//
// The reference is serial: an explicit-stack DFS (right child first, children
// allocated adjacently when their parent is processed) whose nodes each run
// std::partition over their primitive range.  Everything it computes is a pure
// function of the tree topology and of per-range reductions, so it is restated
// level-synchronously (one pass per tree depth over all primitives):
//
//   * node box / centroid box of a range = min/max over the range — order only
//     matters for the sign of a zero (the reference's merge keeps the LAST of two
//     equal values, yocto_math.h:1046-1047).  Ranges reduce through wave-level
//     segmented scans + integer atomics on sortable float keys; a range whose
//     box face is 0 with both +0 and -0 contributors is flagged and the tree
//     falls back to the host builder (never seen on the BASELINE scenes).
//   * std::partition (libstdc++ bidirectional): the k-th `false` element from the
//     left of the final left part swaps with the k-th `true` element from the
//     right of the final right part.  Ranks come from ONE exclusive prefix sum of
//     the predicate over the whole primitive array per level (ranges are
//     contiguous), partners meet through two index arrays.
//   * node ids: the k-th internal node the reference processes (right-first
//     preorder) gets children 1+2k and 2+2k.  k(root)=0, k(right)=k+1,
//     k(left)=k+1+internal(right subtree): one bottom-up and one top-down sweep
//     over the levels.
//
// Compiled with -ffp-contract=off like the rest of the library.

#include "yt_gpubuild.h"

#include <cstring>
#include <vector>

namespace ytgpu {
namespace {

constexpr int BLK       = 256;
constexpr int MAX_DEPTH = 1024;  // beyond this the host builder takes over (degenerate inputs)
constexpr int MAX_PRIMS = 4;     // bvh_max_prims, yocto_bvh.cpp:235

__device__ __forceinline__ float fmin_(float a, float b) { return (a < b) ? a : b; }  // yocto_math.h:1046
__device__ __forceinline__ float fmax_(float a, float b) { return (a > b) ? a : b; }  // yocto_math.h:1047

// order-preserving float <-> uint keys (−0 < +0)
__device__ __forceinline__ unsigned fkey(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// (not inlined: ROCm 7.2's instruction selection crashes — constrainRegClass —
// when it folds this decode into k_decide's comparison chains; it only runs once
// per node)
__device__ __noinline__ float fkey_inv(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// A node of the tree under construction, in allocation ("BFS") order.
struct BNode {
  int   start, end;  // primitive range
  int   left;        // BFS index of the left child (right = left + 1), -1 for a leaf
  int   axis, mid, K, needpart;
  float split;
  float bmin[3], bmax[3];
  int   icount;      // internal nodes in the subtree
  int   rank, id;    // processing rank among internal nodes / reference node index
  int   sah;         // split_sah: index of the node's bin set in this level's Bins array, -1 = none
  float cmin[3], csize[3];  // centroid box (split_sah's cbbox.min, csize)
};
// split_sah's 16 bins per axis (yocto_bvh.cpp:120-148): primitive boxes merged per bin +
// counts.  Bin k of an axis holds the primitives whose centre is >= k of the 15 candidate
// planes; a candidate's left / right box is the merge of the bins below / from it.
constexpr int NBINS = 16, BIN_WORDS = 7, BINSET = 3 * NBINS * BIN_WORDS;  // 336 words per node
struct Bins {
  unsigned v[BINSET];  // [axis][bin]{min.x, min.y, min.z, ~max.x, ~max.y, ~max.z (sortable keys), count}
};
// Range accumulators: 12 keys reduced with min (max values are stored negated)
// + the signed-zero flags.  [0..2] box min, [3..5] ~box max, [6..8] centroid min,
// [9..11] ~centroid max, [12] flags, [13..15] pad
struct Acc {
  unsigned v[16];
};

struct float3x {
  float x, y, z;
};

// primitive bounds — yocto_bvh.cpp:327-355 with yocto_geometry.h:475-498
__device__ __forceinline__ void prim_bounds(int kind, const int* elems, const float* P, const float* R, int i,
    float3x& lo, float3x& hi) {
  auto ld  = [&](int v) { return float3x{P[3 * v], P[3 * v + 1], P[3 * v + 2]}; };
  auto mn2 = [](float3x a, float3x b) { return float3x{fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)}; };
  auto mx2 = [](float3x a, float3x b) { return float3x{fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)}; };
  auto add = [](float3x a, float r) { return float3x{a.x + r, a.y + r, a.z + r}; };
  auto sub = [](float3x a, float r) { return float3x{a.x - r, a.y - r, a.z - r}; };
  if (kind == 0) {  // the boxes ARE the primitives (the instance tree: yocto_bvh.cpp:381-393): P holds {min, max} per box
    lo = float3x{P[6 * i], P[6 * i + 1], P[6 * i + 2]}, hi = float3x{P[6 * i + 3], P[6 * i + 4], P[6 * i + 5]};
  } else if (kind == 1) {  // point_bounds(p, r) = {min(p - r, p + r), max(p - r, p + r)}
    int  v = elems[i];
    auto p = ld(v);
    auto r = R[v];
    lo = mn2(sub(p, r), add(p, r)), hi = mx2(sub(p, r), add(p, r));
  } else if (kind == 2) {  // line_bounds = {min(p0 - r0, p1 - r1), max(p0 + r0, p1 + r1)}
    int  a = elems[2 * i], b = elems[2 * i + 1];
    auto p0 = ld(a), p1 = ld(b);
    auto r0 = R[a], r1 = R[b];
    lo = mn2(sub(p0, r0), sub(p1, r1)), hi = mx2(add(p0, r0), add(p1, r1));
  } else if (kind == 3) {  // triangle_bounds = {min(p0, min(p1, p2)), max(p0, max(p1, p2))}
    auto p0 = ld(elems[3 * i]), p1 = ld(elems[3 * i + 1]), p2 = ld(elems[3 * i + 2]);
    lo = mn2(p0, mn2(p1, p2)), hi = mx2(p0, mx2(p1, p2));
  } else {  // quad_bounds = {min(p0, min(p1, min(p2, p3))), max(...)}
    auto p0 = ld(elems[4 * i]), p1 = ld(elems[4 * i + 1]), p2 = ld(elems[4 * i + 2]), p3 = ld(elems[4 * i + 3]);
    lo = mn2(p0, mn2(p1, mn2(p2, p3))), hi = mx2(p0, mx2(p1, mx2(p2, p3)));
  }
}

__global__ void k_init(int kind, const int* elems, const float* P, const float* R, int n, float4* bbmin,
    float4* bbmax, int* prim, int* node_of) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  float3x lo, hi;
  prim_bounds(kind, elems, P, R, i, lo, hi);
  bbmin[i]   = {lo.x, lo.y, lo.z, 0};
  bbmax[i]   = {hi.x, hi.y, hi.z, 0};
  prim[i]    = i;
  node_of[i] = 0;
}

__global__ void k_init_root(BNode* nodes, Acc* acc, int n, int* counter, int* ambiguous) {
  BNode r = {};
  r.start = 0, r.end = n, r.left = -1;
  nodes[0] = r;
  for (int k = 0; k < 12; k++) acc[0].v[k] = 0xffffffffu;
  for (int k = 12; k < 16; k++) acc[0].v[k] = 0;
  *counter   = 1;
  *ambiguous = 0;
}

// Per-range reductions of the level: node box (merge of the primitives' boxes,
// yocto_bvh.cpp:263-266) and centroid box (split_middle, :207-209).  A lane folds
// RED_ITEMS consecutive primitives (flushing with atomics where the range changes
// inside its stretch), then the wavefront combines equal-range neighbours with a
// segmented scan: one set of atomics per (wavefront, range) — at the top of the
// tree, where every primitive belongs to the same few ranges, that is what keeps
// the same-address atomics (~88 per microsecond on this part) off the critical path.
constexpr int RED_ITEMS = 8;
__global__ void k_reduce(int n, const int* prim, const float4* bbmin, const float4* bbmax, const int* node_of,
    Acc* acc) {
  const int lane  = threadIdx.x & 63;
  const int first = (blockIdx.x * BLK + threadIdx.x) * RED_ITEMS;
  unsigned  v[12], fl = 0;
  int       node = -1;
#pragma unroll
  for (int k = 0; k < 12; k++) v[k] = 0xffffffffu;
  auto flush = [&](int nd) {
    unsigned* a = acc[nd].v;
#pragma unroll
    for (int k = 0; k < 12; k++) atomicMin(&a[k], v[k]);
    if (fl) atomicOr(&a[12], fl);
  };
  for (int it = 0; it < RED_ITEMS; it++) {
    int i = first + it;
    int nd = (i < n) ? node_of[i] : -1;
    if (nd < 0) continue;
    if (nd != node) {
      if (node >= 0) flush(node);  // the range changed inside this lane's stretch
      node = nd, fl = 0;
#pragma unroll
      for (int k = 0; k < 12; k++) v[k] = 0xffffffffu;
    }
    int    p  = prim[i];
    float4 mn = bbmin[p], mx = bbmax[p];
    float  cx = (mn.x + mx.x) / 2, cy = (mn.y + mx.y) / 2, cz = (mn.z + mx.z) / 2;  // center(bbox)
    unsigned w[12] = {fkey(mn.x), fkey(mn.y), fkey(mn.z), ~fkey(mx.x), ~fkey(mx.y), ~fkey(mx.z),
        fkey(cx), fkey(cy), fkey(cz), ~fkey(cx), ~fkey(cy), ~fkey(cz)};
#pragma unroll
    for (int k = 0; k < 12; k++) v[k] = w[k] < v[k] ? w[k] : v[k];
    float f[6] = {mn.x, mn.y, mn.z, mx.x, mx.y, mx.z};
#pragma unroll
    for (int c = 0; c < 6; c++)
      if (f[c] == 0) fl |= (__float_as_uint(f[c]) >> 31 ? 2u : 1u) << (2 * c);
  }
  // `node` / v / fl now describe the LAST range of the lane's stretch; combine lanes
  // whose last range is the same (ranges are contiguous, so equal neighbours form runs)
  const bool act = node >= 0;
  int        seg = act ? node : (-2 - lane);
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int      oseg = __shfl_up(seg, d);
    bool     take = lane >= d && oseg == seg;
    unsigned ofl  = __shfl_up(fl, d);
    if (take) fl |= ofl;
#pragma unroll
    for (int k = 0; k < 12; k++) {
      unsigned o = __shfl_up(v[k], d);
      if (take && o < v[k]) v[k] = o;
    }
  }
  int  nseg = __shfl_down(seg, 1);
  bool last = lane == 63 || nseg != seg;
  if (act && last) flush(node);
}

// Leaf / internal decision and split plane of every node of the level
// (yocto_bvh.cpp:269-291, split_middle :202-221).
__global__ void k_decide(BNode* nodes, const Acc* acc, int lb, int le, int* ambiguous, int highquality, Bins* bins,
    int* bin_counter) {
  int x = lb + blockIdx.x * BLK + threadIdx.x;
  if (x >= le) return;
  const unsigned* a  = acc[x].v;
  BNode*          nd = &nodes[x];
  const unsigned  fl = a[12];
  float bmin0 = fkey_inv(a[0]), bmin1 = fkey_inv(a[1]), bmin2 = fkey_inv(a[2]);
  float bmax0 = fkey_inv(~a[3]), bmax1 = fkey_inv(~a[4]), bmax2 = fkey_inv(~a[5]);
  // a zero face fed by both +0 and -0: the reference keeps the sign of the last
  // contributor in range order, which the atomics do not know
  bool amb = (bmin0 == 0 && ((fl >> 0) & 3u) == 3u) || (bmin1 == 0 && ((fl >> 2) & 3u) == 3u) ||
             (bmin2 == 0 && ((fl >> 4) & 3u) == 3u) || (bmax0 == 0 && ((fl >> 6) & 3u) == 3u) ||
             (bmax1 == 0 && ((fl >> 8) & 3u) == 3u) || (bmax2 == 0 && ((fl >> 10) & 3u) == 3u);
  if (amb) *ambiguous = 1;
  const int start = nd->start, end = nd->end, size = end - start;
  if (size == 0) {  // invalidb3f of an empty range
    bmin0 = bmin1 = bmin2 = 3.402823466e+38f;
    bmax0 = bmax1 = bmax2 = -3.402823466e+38f;
  }
  nd->bmin[0] = bmin0, nd->bmin[1] = bmin1, nd->bmin[2] = bmin2;
  nd->bmax[0] = bmax0, nd->bmax[1] = bmax1, nd->bmax[2] = bmax2;
  int   left = -1, needpart = 0, axis = 0, mid = start, sah = -1;
  float split = 0;
  if (size > MAX_PRIMS) {
    float cmin0 = fkey_inv(a[6]), cmin1 = fkey_inv(a[7]), cmin2 = fkey_inv(a[8]);
    float cmax0 = fkey_inv(~a[9]), cmax1 = fkey_inv(~a[10]), cmax2 = fkey_inv(~a[11]);
    float cs0 = cmax0 - cmin0, cs1 = cmax1 - cmin1, cs2 = cmax2 - cmin2;
    left = -2;  // internal, children allocated by k_mid
    if (cs0 == 0 && cs1 == 0 && cs2 == 0) {
      mid = (start + end) / 2;
    } else if (highquality) {  // split_sah: plane chosen by k_sah_decide once the bins are filled
      sah = atomicAdd(bin_counter, 1);
      unsigned* b = bins[sah].v;
      for (int k = 0; k < BINSET; k++) b[k] = (k % BIN_WORDS == BIN_WORDS - 1) ? 0u : 0xffffffffu;
      nd->cmin[0] = cmin0, nd->cmin[1] = cmin1, nd->cmin[2] = cmin2;
      nd->csize[0] = cs0, nd->csize[1] = cs1, nd->csize[2] = cs2;
      needpart = 1;
    } else {
      if (cs0 >= cs1 && cs0 >= cs2) axis = 0;
      if (cs1 >= cs0 && cs1 >= cs2) axis = 1;
      if (cs2 >= cs0 && cs2 >= cs1) axis = 2;
      float lo = axis == 0 ? cmin0 : (axis == 1 ? cmin1 : cmin2);
      float hi = axis == 0 ? cmax0 : (axis == 1 ? cmax1 : cmax2);
      split    = (lo + hi) / 2;  // center(cbbox)[axis]
      needpart = 1;
    }
  }
  nd->left = left, nd->K = 0, nd->needpart = needpart, nd->axis = axis, nd->mid = mid, nd->split = split;
  nd->sah  = sah;
}

// ---- split_sah (yocto_bvh.cpp:108-164) ------------------------------------------------
// candidate plane b of an axis: cbbox.min[axis] + b * csize[axis] / nbins, as the reference writes it
__device__ __forceinline__ float sah_plane(float cmin, float csize, int b) { return cmin + (float)b * csize / (float)NBINS; }

// Binning pass of the level: every primitive of a split_sah node adds its box and one count to
// one bin per axis.  The planes are non-decreasing in b (a rounded monotonic function), so
// "centre < plane b" holds exactly for b > k: the bin index k is the number of planes the
// centre is not below.  A block covers BIN_ITEMS * BLK consecutive primitives; the first
// BIN_SLOTS ranges it meets are accumulated in LDS and flushed with one set of atomics per
// block (at the top of the tree all primitives share a few nodes), the rest go straight to
// global atomics (deep levels: few primitives per node, no contention).
constexpr int BIN_ITEMS = 8, BIN_SLOTS = 4;
__global__ void __launch_bounds__(BLK) k_bin(int n, const int* prim, const float4* bbmin, const float4* bbmax,
    const int* node_of, const BNode* nodes, Bins* bins) {
  __shared__ unsigned s_bins[BIN_SLOTS][BINSET];
  __shared__ int      s_set[BIN_SLOTS];
  __shared__ int      s_wave[BLK / 64];
  const int tid = threadIdx.x, base = blockIdx.x * (BLK * BIN_ITEMS), first = base + tid * BIN_ITEMS;
  for (int k = tid; k < BIN_SLOTS * BINSET; k += BLK)
    (&s_bins[0][0])[k] = (k % BIN_WORDS == BIN_WORDS - 1) ? 0u : 0xffffffffu;
  if (tid < BIN_SLOTS) s_set[tid] = -1;
  // slot of an element = number of range changes between the block's first element and it
  int nd[BIN_ITEMS], changes = 0;
  int prev = (first > base && first - 1 < n) ? node_of[first - 1] : -3;
#pragma unroll
  for (int it = 0; it < BIN_ITEMS; it++) {
    int i  = first + it;
    nd[it] = i < n ? node_of[i] : -2;
    if (i < n && i > base && nd[it] != prev) changes++;
    prev = nd[it];
  }
  const int lane = tid & 63, wave = tid >> 6;
  int       incl = changes;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int slot = incl - changes;
  for (int w = 0; w < wave; w++) slot += s_wave[w];
  prev = (first > base && first - 1 < n) ? node_of[first - 1] : -3;
#pragma unroll
  for (int it = 0; it < BIN_ITEMS; it++) {
    int i = first + it;
    if (i >= n) break;
    if (i > base && nd[it] != prev) slot++;
    prev = nd[it];
    if (nd[it] < 0) continue;
    const BNode& node = nodes[nd[it]];
    const int    set  = node.sah;
    if (set < 0) continue;
    int    p  = prim[i];
    float4 mn = bbmin[p], mx = bbmax[p];
    float  c[3] = {(mn.x + mx.x) / 2, (mn.y + mx.y) / 2, (mn.z + mx.z) / 2};  // centers[] = center(bbox), :247
    unsigned key[6] = {fkey(mn.x), fkey(mn.y), fkey(mn.z), ~fkey(mx.x), ~fkey(mx.y), ~fkey(mx.z)};
    unsigned* dst;
    const bool in_lds = slot < BIN_SLOTS;
    if (in_lds) {
      s_set[slot] = set;
      dst         = s_bins[slot];
    } else {
      dst = bins[set].v;
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
      int k = 0;
      for (int b = 1; b < NBINS; b++) k += (c[a] < sah_plane(node.cmin[a], node.csize[a], b)) ? 0 : 1;
      unsigned* bin = dst + (a * NBINS + k) * BIN_WORDS;
#pragma unroll
      for (int w = 0; w < 6; w++) atomicMin(&bin[w], key[w]);
      atomicAdd(&bin[6], 1u);
    }
  }
  __syncthreads();
  for (int sl = 0; sl < BIN_SLOTS; sl++) {
    const int set = s_set[sl];
    if (set < 0) continue;
    for (int k = tid; k < BINSET; k += BLK) {
      unsigned v = s_bins[sl][k];
      if (k % BIN_WORDS == BIN_WORDS - 1) {
        if (v) atomicAdd(&bins[set].v[k], v);
      } else if (v != 0xffffffffu) {
        atomicMin(&bins[set].v[k], v);
      }
    }
  }
}

// The 45 candidates of split_sah in its own order (axis outer, plane inner), the cost in its
// own float expression, the first strict minimum wins (yocto_bvh.cpp:119-148).
__global__ void k_sah_decide(BNode* nodes, int lb, int le, const Bins* bins) {
  int x = lb + blockIdx.x * BLK + threadIdx.x;
  if (x >= le) return;
  BNode* nd = &nodes[x];
  if (nd->sah < 0) return;
  const unsigned* B = bins[nd->sah].v;
  struct Box {
    float lo[3], hi[3];
  };
  const float flt_max = 3.402823466e+38f;
  auto invalid = [&]() { return Box{{flt_max, flt_max, flt_max}, {-flt_max, -flt_max, -flt_max}}; };
  auto area = [](const Box& b) {  // bbox_area, :124-128
    float sx = b.hi[0] - b.lo[0], sy = b.hi[1] - b.lo[1], sz = b.hi[2] - b.lo[2];
    return 1e-12f + 2 * sx * sy + 2 * sx * sz + 2 * sy * sz;
  };
  auto merge_bin = [&](Box& b, int& count, int a, int k) {
    const unsigned* bin = B + (a * NBINS + k) * BIN_WORDS;
    if (bin[6] == 0) return;
    count += (int)bin[6];
    for (int c = 0; c < 3; c++) {
      b.lo[c] = fmin_(b.lo[c], fkey_inv(bin[c]));
      b.hi[c] = fmax_(b.hi[c], fkey_inv(~bin[3 + c]));
    }
  };
  // bbox_area(cbbox): size = cbbox.max - cbbox.min = csize
  const float carea = 1e-12f + 2 * nd->csize[0] * nd->csize[1] + 2 * nd->csize[0] * nd->csize[2] +
                      2 * nd->csize[1] * nd->csize[2];
  int   axis = 0;
  float split = 0.0f, min_cost = flt_max;
  for (int a = 0; a < 3; a++) {
    for (int b = 1; b < NBINS; b++) {
      Box left = invalid(), right = invalid();
      int nl = 0, nr = 0;
      for (int k = 0; k < b; k++) merge_bin(left, nl, a, k);
      for (int k = b; k < NBINS; k++) merge_bin(right, nr, a, k);
      float cost = 1 + nl * area(left) / carea + nr * area(right) / carea;
      if (cost < min_cost) {
        min_cost = cost;
        split    = sah_plane(nd->cmin[a], nd->csize[a], b);
        axis     = a;
      }
    }
  }
  nd->axis = axis, nd->split = split;
}

// predicate of std::partition, negated: 1 where centers[primitive][axis] < split is FALSE
__global__ void k_flags(int n, const int* prim, const float4* bbmin, const float4* bbmax, const int* node_of,
    const BNode* nodes, int* flag) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  int node = node_of[i], f = 0;
  if (node >= 0) {
    const BNode& nd = nodes[node];
    if (nd.needpart) {
      int    p  = prim[i];
      float4 mn = bbmin[p], mx = bbmax[p];
      float  lo = nd.axis == 0 ? mn.x : (nd.axis == 1 ? mn.y : mn.z);
      float  hi = nd.axis == 0 ? mx.x : (nd.axis == 1 ? mx.y : mx.z);
      float  c  = (lo + hi) / 2;
      f         = (c < nd.split) ? 0 : 1;
    }
  }
  flag[i] = f;
}

// ---- exclusive prefix sum: out[0..n] (out[n] = total) -------------------------------
constexpr int SCAN_ITEMS = 4, SCAN_TILE = BLK * SCAN_ITEMS;
__global__ void k_scan_tiles(const int* in, int* out, int* tile_sums, int n) {
  __shared__ int s_wave[BLK / 64];
  int base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS], sum = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    v[k] = (base + k < n) ? in[base + k] : 0;
    sum += v[k];
  }
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int woff = 0, total = 0;
#pragma unroll
  for (int w = 0; w < BLK / 64; w++) {
    if (w < wave) woff += s_wave[w];
    total += s_wave[w];
  }
  int run = woff + incl - sum;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    if (base + k < n) out[base + k] = run;
    run += v[k];
  }
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}
__global__ void k_scan_sums(int* tile_sums, int ntiles) {  // one block, in place → exclusive
  __shared__ int s_wave[BLK / 64];
  __shared__ int s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int b = 0; b < ntiles; b += BLK) {
    int i = b + threadIdx.x, v = (i < ntiles) ? tile_sums[i] : 0;
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      int o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < BLK / 64; w++) {
      if (w < wave) woff += s_wave[w];
      total += s_wave[w];
    }
    int carry = s_carry;
    if (i < ntiles) tile_sums[i] = carry + woff + incl - v;
    __syncthreads();
    if (threadIdx.x == 0) s_carry = carry + total;
    __syncthreads();
  }
  if (threadIdx.x == 0) tile_sums[ntiles] = s_carry;  // grand total
}
__global__ void k_scan_add(int* out, const int* tile_sums, int n, int ntiles) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i < n) out[i] += tile_sums[i / SCAN_TILE];
  if (i == 0) out[n] = tile_sums[ntiles];
}
void exclusive_scan(hipStream_t s, const int* in, int* out, int* tile_sums, int n) {
  int ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(BLK), 0, s, in, out, tile_sums, n);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(BLK), 0, s, tile_sums, ntiles);
  hipLaunchKernelGGL(k_scan_add, dim3((n + BLK - 1) / BLK), dim3(BLK), 0, s, out, tile_sums, n, ntiles);
}

// Split position of every internal node of the level (split_middle :222-231) and
// allocation of its two children.
__global__ void k_mid(BNode* nodes, Acc* acc, int lb, int le, const int* G, int* counter) {
  int x = lb + blockIdx.x * BLK + threadIdx.x;
  if (x >= le) return;
  BNode nd = nodes[x];
  if (nd.left != -2) return;
  int size = nd.end - nd.start;
  if (nd.needpart) {
    int nfalse = G[nd.end] - G[nd.start];
    int m      = size - nfalse;
    if (m == 0 || m == size) {  // "if we were not able to split, just break the primitives in half"
      nd.mid = (nd.start + nd.end) / 2;
      nd.K   = 0;
    } else {
      nd.mid = nd.start + m;
      nd.K   = G[nd.mid] - G[nd.start];  // `false` elements inside the final left part = swaps
    }
  }
  int c   = atomicAdd(counter, 2);
  nd.left = c;
  BNode l = {}, r = {};
  l.start = nd.start, l.end = nd.mid, l.left = -1;
  r.start = nd.mid, r.end = nd.end, r.left = -1;
  nodes[c] = l, nodes[c + 1] = r;
  for (int k = 0; k < 16; k++) {
    unsigned init = k < 12 ? 0xffffffffu : 0u;
    acc[c].v[k] = init, acc[c + 1].v[k] = init;
  }
  nodes[x] = nd;
}

// std::partition, step 1: the k-th false of the left part and the k-th true
// (from the right) of the right part publish their positions.
__global__ void k_scatter(int n, const int* node_of, const BNode* nodes, const int* G, const int* flag, int* A,
    int* B) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  int node = node_of[i];
  if (node < 0) return;
  const BNode& nd = nodes[node];
  if (nd.K <= 0) return;
  int rf = G[i] - G[nd.start];  // falses before i in the range
  int f  = flag[i];
  if (i < nd.mid) {
    if (f) A[nd.start + rf] = i;
  } else if (!f) {
    int m          = nd.mid - nd.start;
    int trues_incl = (i - nd.start + 1) - rf;
    B[nd.start + (m - trues_incl)] = i;
  }
}
// step 2: the swaps; then every primitive moves to its child range (or retires
// with its leaf).
__global__ void k_swap_assign(int n, int* node_of, const BNode* nodes, const int* A, const int* B, int* prim) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  int node = node_of[i];
  if (node < 0) return;
  const BNode& nd = nodes[node];
  if (nd.left < 0) {
    node_of[i] = -1;
    return;
  }
  if (i - nd.start < nd.K) {
    int a = A[i], b = B[i];
    int t = prim[a];
    prim[a] = prim[b];
    prim[b] = t;
  }
  node_of[i] = nd.left + (i >= nd.mid ? 1 : 0);
}

// node numbering ------------------------------------------------------------------------
__global__ void k_icount(BNode* nodes, int lb, int le) {
  int x = lb + blockIdx.x * BLK + threadIdx.x;
  if (x >= le) return;
  int l           = nodes[x].left;
  nodes[x].icount = l >= 0 ? 1 + nodes[l].icount + nodes[l + 1].icount : 0;
}
__global__ void k_number(BNode* nodes, int lb, int le) {
  int x = lb + blockIdx.x * BLK + threadIdx.x;
  if (x >= le) return;
  if (x == 0) nodes[0].rank = 0, nodes[0].id = 0;
  int l = nodes[x].left;
  if (l < 0) return;
  int rank          = nodes[x].rank;
  nodes[l].id       = 1 + 2 * rank;
  nodes[l + 1].id   = 2 + 2 * rank;
  nodes[l + 1].rank = rank + 1;                         // the right child is processed next
  nodes[l].rank     = rank + 1 + nodes[l + 1].icount;  // the left one after the whole right subtree
}
__global__ void k_emit(const BNode* nodes, int count, ythip_bvh_node* out) {
  int x = blockIdx.x * BLK + threadIdx.x;
  if (x >= count) return;
  const BNode&   nd = nodes[x];
  ythip_bvh_node o;
  for (int c = 0; c < 3; c++) o.bbox_min[c] = nd.bmin[c], o.bbox_max[c] = nd.bmax[c];
  if (nd.left >= 0) {
    o.start = 1 + 2 * nd.rank, o.num = 2, o.axis = (int8_t)nd.axis, o.internal = 1;
  } else {
    o.start = nd.start, o.num = (int16_t)(nd.end - nd.start), o.axis = 0, o.internal = 0;
  }
  out[nd.id] = o;
}

// ---- bake (device version of bake_bvh in ythip.hip) -----------------------------------
__global__ void k_internal_flags(const ythip_bvh_node* nodes, int n, int* flag) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i < n) flag[i] = nodes[i].internal ? 1 : 0;
}
__device__ __forceinline__ int ref_of(const ythip_bvh_node& ch, int node, const int* pid, long long pair_base,
    long long prim_base) {
  if (ch.internal) return (int)(pair_base + pid[node]);
  return (int)(0x80000000u | ((unsigned)(ch.num & 7) << 28) | (unsigned)(prim_base + ch.start));
}
__global__ void k_bake_pairs(const ythip_bvh_node* nodes, int n, const int* pid, long long pair_base,
    long long prim_base, float4* pairs) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  ythip_bvh_node nd = nodes[i];
  if (!nd.internal) return;
  float4* P = pairs + 4 * (pair_base + pid[i]);
  for (int c = 0; c < 2; c++) {
    int            cn = nd.start + c;
    ythip_bvh_node ch = nodes[cn];
    P[2 * c]          = {ch.bbox_min[0], ch.bbox_min[1], ch.bbox_max[0], ch.bbox_max[1]};
    P[2 * c + 1]      = {ch.bbox_min[2], ch.bbox_max[2], __int_as_float(ref_of(ch, cn, pid, pair_base, prim_base)),
             __int_as_float((int)nd.axis)};
  }
}
// grandchildren ("quad") records, same ids as the pairs (yt_bvh.h WIDE walk)
__global__ void k_bake_quads(const ythip_bvh_node* nodes, int n, const int* pid, long long pair_base,
    long long prim_base, float4* quads) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  ythip_bvh_node nd = nodes[i];
  if (!nd.internal) return;
  float4* Q    = quads + 8 * (pair_base + pid[i]);
  int     axes = nd.axis & 3;
  for (int h = 0; h < 2; h++) {
    int            cn = nd.start + h;
    ythip_bvh_node ch = nodes[cn];
    int            s0 = cn, s1 = -1;
    if (ch.internal) {
      s0 = ch.start, s1 = ch.start + 1;
      axes |= (ch.axis & 3) << (2 + 2 * h);
    }
    for (int k = 0; k < 2; k++) {
      int     sn = k == 0 ? s0 : s1;
      float4* S  = Q + 2 * (2 * h + k);
      if (sn < 0) {
        S[0] = {0, 0, 0, 0};
        S[1] = {0, 0, __int_as_float(0x7ffffffe), 0};  // REF_NONE
        continue;
      }
      ythip_bvh_node g = nodes[sn];
      S[0]             = {g.bbox_min[0], g.bbox_min[1], g.bbox_max[0], g.bbox_max[1]};
      S[1]             = {g.bbox_min[2], g.bbox_max[2], __int_as_float(ref_of(g, sn, pid, pair_base, prim_base)), 0};
    }
  }
  Q[1].w = __int_as_float(axes);
}
__global__ void k_bake_leaf(int kind, const int* elems, const float* P, const float* R, const int* prims, int n,
    float4* leaf) {
  int k = blockIdx.x * BLK + threadIdx.x;
  if (k >= n) return;
  int  id  = prims[k];
  auto pos = [&](int v) { return float3x{P[3 * v], P[3 * v + 1], P[3 * v + 2]}; };
  if (kind == 3) {
    auto    p0 = pos(elems[3 * id]), p1 = pos(elems[3 * id + 1]), p2 = pos(elems[3 * id + 2]);
    float4* L  = leaf + 3 * (long long)k;
    L[0] = {p0.x, p0.y, p0.z, p1.x};
    L[1] = {p1.y, p1.z, p2.x, p2.y};
    L[2] = {p2.z, __int_as_float(id), 0, 0};
  } else if (kind == 4) {
    auto    p0 = pos(elems[4 * id]), p1 = pos(elems[4 * id + 1]), p2 = pos(elems[4 * id + 2]),
         p3    = pos(elems[4 * id + 3]);
    float4* L  = leaf + 4 * (long long)k;
    L[0] = {p0.x, p0.y, p0.z, p1.x};
    L[1] = {p1.y, p1.z, p2.x, p2.y};
    L[2] = {p2.z, p3.x, p3.y, p3.z};
    L[3] = {__int_as_float(id), 0, 0, 0};
  } else if (kind == 2) {
    int     a = elems[2 * id], b = elems[2 * id + 1];
    auto    p0 = pos(a), p1 = pos(b);
    float4* L  = leaf + 3 * (long long)k;
    L[0] = {p0.x, p0.y, p0.z, p1.x};
    L[1] = {p1.y, p1.z, R ? R[a] : 0.0f, R ? R[b] : 0.0f};
    L[2] = {__int_as_float(id), 0, 0, 0};
  } else {
    int     v = elems[id];
    auto    p = pos(v);
    float4* L = leaf + 2 * (long long)k;
    L[0] = {p.x, p.y, p.z, R ? R[v] : 0.0f};
    L[1] = {__int_as_float(id), 0, 0, 0};
  }
}

// ---- refit_bvh (yocto_bvh.cpp:305-319) on a resident tree ---------------------------------
// The reference sweeps the node array backwards (children sit behind their parent).
// Here every leaf recomputes its box from the moved vertices and climbs: the second of
// two siblings to arrive at the parent (an atomic counter per node decides) merges the
// children — always child 0 then child 1 into the invalid box, the reference's order,
// which fixes the sign of a zero face — and goes on.  One launch, no level lists.
__global__ void k_refit_parents(const ythip_bvh_node* nodes, int n, int* parent, int* arrived) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  arrived[i] = 0;
  if (i == 0) parent[0] = -1;
  if (nodes[i].internal) parent[nodes[i].start] = i, parent[nodes[i].start + 1] = i;
}

__device__ __forceinline__ void store_box(ythip_bvh_node* node, float3x lo, float3x hi) {
  // agent-scope stores / loads: the sibling that merges may sit on another XCD (own L2)
  __hip_atomic_store(&node->bbox_min[0], lo.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&node->bbox_min[1], lo.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&node->bbox_min[2], lo.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&node->bbox_max[0], hi.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&node->bbox_max[1], hi.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&node->bbox_max[2], hi.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void load_box(const ythip_bvh_node* node, float3x& lo, float3x& hi) {
  lo.x = __hip_atomic_load(&node->bbox_min[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  lo.y = __hip_atomic_load(&node->bbox_min[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  lo.z = __hip_atomic_load(&node->bbox_min[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  hi.x = __hip_atomic_load(&node->bbox_max[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  hi.y = __hip_atomic_load(&node->bbox_max[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  hi.z = __hip_atomic_load(&node->bbox_max[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void k_refit(ythip_bvh_node* nodes, int n, const int* prims, int kind, const int* elems, const float* P,
    const float* R, const int* parent, int* arrived) {
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  ythip_bvh_node node = nodes[i];
  if (node.internal) return;
  constexpr float FMAX = 3.402823466e+38f;
  float3x lo = {FMAX, FMAX, FMAX}, hi = {-FMAX, -FMAX, -FMAX};  // invalidb3f
  for (int k = 0; k < node.num; k++) {  // merge(node.bbox, bboxes[primitives[start + k]])
    float3x a, b;
    prim_bounds(kind, elems, P, R, prims[node.start + k], a, b);
    lo = {fmin_(lo.x, a.x), fmin_(lo.y, a.y), fmin_(lo.z, a.z)};
    hi = {fmax_(hi.x, b.x), fmax_(hi.y, b.y), fmax_(hi.z, b.z)};
  }
  store_box(&nodes[i], lo, hi);
  int cur = i;
  while (true) {
    int p = parent[cur];
    if (p < 0) break;
    // acq_rel at agent scope: my box is visible before my arrival, the sibling's box after its arrival
    if (__hip_atomic_fetch_add(&arrived[p], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == 0)
      break;  // the sibling is still on its way: it will do the parent
    int     c0 = nodes[p].start;
    float3x l0, h0, l1, h1;
    load_box(&nodes[c0], l0, h0);
    load_box(&nodes[c0 + 1], l1, h1);
    lo = {fmin_(fmin_(FMAX, l0.x), l1.x), fmin_(fmin_(FMAX, l0.y), l1.y), fmin_(fmin_(FMAX, l0.z), l1.z)};
    hi = {fmax_(fmax_(-FMAX, h0.x), h1.x), fmax_(fmax_(-FMAX, h0.y), h1.y), fmax_(fmax_(-FMAX, h0.z), h1.z)};
    store_box(&nodes[p], lo, hi);
    cur = p;
  }
}

int grid(long long n) { return (int)((n + BLK - 1) / BLK); }

#define GCHECK(call)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (call);                                                                            \
    if (e_ != hipSuccess) {                                                                            \
      if (err) *err = std::string(#call) + " failed: " + hipGetErrorString(e_);                        \
      cleanup();                                                                                       \
      return BUILD_ERROR;                                                                              \
    }                                                                                                  \
  } while (0)

}  // namespace

void free_tree(DeviceTree* t) {
  if (!t) return;
  if (t->nodes) (void)hipFree(t->nodes);
  if (t->prims) (void)hipFree(t->prims);
  *t = DeviceTree{};
}

int build_shape_tree(hipStream_t s, int kind, const int32_t* elems, const float* positions, const float* radius,
    int64_t num_prims, bool highquality, DeviceTree* out, std::string* err) {
  *out = DeviceTree{};
  if (num_prims <= MAX_PRIMS || num_prims > (1ll << 28) || kind < 0 || kind > 4) return BUILD_FALLBACK;
  if ((kind == 1 || kind == 2) && !radius) return BUILD_FALLBACK;
  const int n = (int)num_prims;

  // scratch: one allocation, carved up
  const size_t max_nodes = 2 * (size_t)n + 2;
  const int    ntiles    = (n + SCAN_TILE - 1) / SCAN_TILE;
  size_t       off       = 0;
  auto         carve     = [&](size_t bytes) {
    size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  };
  size_t o_bbmin = carve(n * sizeof(float4)), o_bbmax = carve(n * sizeof(float4));
  size_t o_nodeof = carve(n * sizeof(int)), o_flag = carve(n * sizeof(int)), o_G = carve((n + 1) * sizeof(int));
  size_t o_A = carve(n * sizeof(int)), o_B = carve(n * sizeof(int)), o_tiles = carve((ntiles + 1) * sizeof(int));
  size_t o_nodes = carve(max_nodes * sizeof(BNode)), o_acc = carve(max_nodes * sizeof(Acc));
  size_t o_ctr = carve(4 * sizeof(int));
  // split_sah: one bin set per internal node of a level; internal nodes hold > 4 primitives each
  const size_t max_binsets = highquality ? (size_t)n / (MAX_PRIMS + 1) + 2 : 0;
  size_t       o_bins      = carve(max_binsets * sizeof(Bins));
  char*  scratch = nullptr;
  int*   prim    = nullptr;
  ythip_bvh_node* out_nodes = nullptr;
  hipEvent_t      ev0 = nullptr, ev1 = nullptr;
  auto cleanup = [&]() {
    if (scratch) (void)hipFree(scratch);
    if (prim) (void)hipFree(prim);
    if (out_nodes) (void)hipFree(out_nodes);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    scratch = nullptr, prim = nullptr, out_nodes = nullptr, ev0 = ev1 = nullptr;
  };
  GCHECK(hipMalloc((void**)&scratch, off));
  GCHECK(hipMalloc((void**)&prim, (size_t)n * sizeof(int)));
  GCHECK(hipEventCreate(&ev0));
  GCHECK(hipEventCreate(&ev1));
  auto* bbmin   = (float4*)(scratch + o_bbmin);
  auto* bbmax   = (float4*)(scratch + o_bbmax);
  auto* node_of = (int*)(scratch + o_nodeof);
  auto* flag    = (int*)(scratch + o_flag);
  auto* G       = (int*)(scratch + o_G);
  auto* A       = (int*)(scratch + o_A);
  auto* B       = (int*)(scratch + o_B);
  auto* tiles   = (int*)(scratch + o_tiles);
  auto* nodes   = (BNode*)(scratch + o_nodes);
  auto* acc     = (Acc*)(scratch + o_acc);
  auto* counter = (int*)(scratch + o_ctr);
  auto* ambig   = counter + 1;
  auto* binctr  = counter + 2;
  auto* bins    = (Bins*)(scratch + o_bins);

  GCHECK(hipEventRecord(ev0, s));
  hipLaunchKernelGGL(k_init, dim3(grid(n)), dim3(BLK), 0, s, kind, elems, positions, radius, n, bbmin, bbmax, prim,
      node_of);
  hipLaunchKernelGGL(k_init_root, dim3(1), dim3(1), 0, s, nodes, acc, n, counter, ambig);

  std::vector<int> level_base = {0};
  int              le         = 1;  // nodes allocated so far
  for (int level = 0;; level++) {
    if (level >= MAX_DEPTH) {
      cleanup();
      return BUILD_FALLBACK;
    }
    int lb = level_base[level];
    hipLaunchKernelGGL(k_reduce, dim3(grid((n + RED_ITEMS - 1) / RED_ITEMS)), dim3(BLK), 0, s, n, prim, bbmin, bbmax,
        node_of, acc);
    if (highquality) GCHECK(hipMemsetAsync(binctr, 0, sizeof(int), s));
    hipLaunchKernelGGL(k_decide, dim3(grid(le - lb)), dim3(BLK), 0, s, nodes, acc, lb, le, ambig, highquality ? 1 : 0,
        bins, binctr);
    if (highquality) {
      hipLaunchKernelGGL(k_bin, dim3(grid((n + BIN_ITEMS - 1) / BIN_ITEMS)), dim3(BLK), 0, s, n, prim, bbmin, bbmax,
          node_of, nodes, bins);
      hipLaunchKernelGGL(k_sah_decide, dim3(grid(le - lb)), dim3(BLK), 0, s, nodes, lb, le, bins);
    }
    hipLaunchKernelGGL(k_flags, dim3(grid(n)), dim3(BLK), 0, s, n, prim, bbmin, bbmax, node_of, nodes, flag);
    exclusive_scan(s, flag, G, tiles, n);
    hipLaunchKernelGGL(k_mid, dim3(grid(le - lb)), dim3(BLK), 0, s, nodes, acc, lb, le, G, counter);
    hipLaunchKernelGGL(k_scatter, dim3(grid(n)), dim3(BLK), 0, s, n, node_of, nodes, G, flag, A, B);
    hipLaunchKernelGGL(k_swap_assign, dim3(grid(n)), dim3(BLK), 0, s, n, node_of, nodes, A, B, prim);
    int host[2];
    GCHECK(hipMemcpyAsync(host, counter, sizeof(host), hipMemcpyDeviceToHost, s));
    GCHECK(hipStreamSynchronize(s));
    if (host[1]) {  // signed-zero tie: only the serial order knows the answer
      cleanup();
      return BUILD_FALLBACK;
    }
    if (host[0] == le) break;  // no internal node on this level
    level_base.push_back(le);
    le = host[0];
  }
  level_base.push_back(le);  // level_base[l] .. level_base[l+1]
  const int nlevels = (int)level_base.size() - 1;
  for (int l = nlevels - 1; l >= 0; l--)
    hipLaunchKernelGGL(k_icount, dim3(grid(level_base[l + 1] - level_base[l])), dim3(BLK), 0, s, nodes,
        level_base[l], level_base[l + 1]);
  for (int l = 0; l < nlevels; l++)
    hipLaunchKernelGGL(k_number, dim3(grid(level_base[l + 1] - level_base[l])), dim3(BLK), 0, s, nodes,
        level_base[l], level_base[l + 1]);
  GCHECK(hipMalloc((void**)&out_nodes, (size_t)le * sizeof(ythip_bvh_node)));
  hipLaunchKernelGGL(k_emit, dim3(grid(le)), dim3(BLK), 0, s, nodes, le, out_nodes);
  GCHECK(hipEventRecord(ev1, s));
  GCHECK(hipStreamSynchronize(s));
  GCHECK(hipGetLastError());
  float ms = 0;
  (void)hipEventElapsedTime(&ms, ev0, ev1);
  out->nodes = out_nodes, out->prims = prim, out->num_nodes = le, out->num_prims = n;
  out->depth = nlevels, out->build_ms = ms;
  out_nodes = nullptr, prim = nullptr;  // ownership moved
  cleanup();
  return BUILD_OK;
}

int bake_shape_tree(hipStream_t s, const DeviceTree& tree, int kind, const int32_t* elems, const float* positions,
    const float* radius, int64_t pair_base, int64_t prim_base, int64_t leaf_base, float4* pairs, float4* quads,
    float4* leafdata, float* root_out, std::string* err) {
  const int n = (int)tree.num_nodes, np = (int)tree.num_prims;
  const int ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  int *     flag = nullptr, *pid = nullptr, *tiles = nullptr;
  auto      cleanup = [&]() {
    if (flag) (void)hipFree(flag);
    if (pid) (void)hipFree(pid);
    if (tiles) (void)hipFree(tiles);
    flag = pid = tiles = nullptr;
  };
  GCHECK(hipMalloc((void**)&flag, (size_t)n * sizeof(int)));
  GCHECK(hipMalloc((void**)&pid, ((size_t)n + 1) * sizeof(int)));
  GCHECK(hipMalloc((void**)&tiles, ((size_t)ntiles + 1) * sizeof(int)));
  hipLaunchKernelGGL(k_internal_flags, dim3(grid(n)), dim3(BLK), 0, s, tree.nodes, n, flag);
  exclusive_scan(s, flag, pid, tiles, n);
  hipLaunchKernelGGL(k_bake_pairs, dim3(grid(n)), dim3(BLK), 0, s, tree.nodes, n, pid, (long long)pair_base,
      (long long)prim_base, pairs);
  hipLaunchKernelGGL(k_bake_quads, dim3(grid(n)), dim3(BLK), 0, s, tree.nodes, n, pid, (long long)pair_base,
      (long long)prim_base, quads);
  if (kind != 0)  // (the instance tree has no leaf data: its leaves index tlas_prims)
    hipLaunchKernelGGL(k_bake_leaf, dim3(grid(np)), dim3(BLK), 0, s, kind, elems, positions, radius, tree.prims, np,
        leafdata + leaf_base);
  ythip_bvh_node root;
  GCHECK(hipMemcpyAsync(&root, tree.nodes, sizeof(root), hipMemcpyDeviceToHost, s));
  GCHECK(hipStreamSynchronize(s));
  GCHECK(hipGetLastError());
  for (int c = 0; c < 3; c++) root_out[c] = root.bbox_min[c], root_out[3 + c] = root.bbox_max[c];
  int ref = root.internal ? (int)pair_base
                          : (int)(0x80000000u | ((unsigned)(root.num & 7) << 28) | (unsigned)(prim_base + root.start));
  std::memcpy(&root_out[6], &ref, 4);
  cleanup();
  return BUILD_OK;
}

// ---- batched bake of host-resident trees ---------------------------------------------------
namespace {
__device__ __forceinline__ int tree_of_node(const HostTreeDesc* T, int nt, long long i) {
  int lo = 0, hi = nt - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (T[mid].node_off <= i) lo = mid; else hi = mid - 1;
  }
  return lo;
}
__device__ __forceinline__ int tree_of_prim(const HostTreeDesc* T, int nt, long long k) {
  int lo = 0, hi = nt - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (T[mid].prim_off <= k) lo = mid; else hi = mid - 1;
  }
  return lo;
}
// pair id of compact node `cn` (an internal node of tree t): the tree's base + its rank among the tree's internal nodes
__device__ __forceinline__ int ref_of_all(const ythip_bvh_node& ch, long long cn, const int* scan, const HostTreeDesc& t) {
  if (ch.internal) return (int)(t.pair_base + (scan[cn] - scan[t.node_off]));
  return (int)(0x80000000u | ((unsigned)(ch.num & 7) << 28) | (unsigned)(t.ref_prim_base + ch.start));
}
__global__ void k_bake_all_nodes(const ythip_bvh_node* nodes, long long n, const int* scan, const HostTreeDesc* T, int nt,
    float4* pairs, float4* quads) {
  long long i = (long long)blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  ythip_bvh_node nd = nodes[i];
  if (!nd.internal) return;
  const HostTreeDesc t  = T[tree_of_node(T, nt, i)];
  const long long    id = t.pair_base + (scan[i] - scan[t.node_off]);
  float4*            P  = pairs + 4 * id;
  float4*            Q  = quads + 8 * id;
  int                axes = nd.axis & 3;
  for (int h = 0; h < 2; h++) {
    long long      cn = t.node_off + nd.start + h;
    ythip_bvh_node ch = nodes[cn];
    P[2 * h]          = {ch.bbox_min[0], ch.bbox_min[1], ch.bbox_max[0], ch.bbox_max[1]};
    P[2 * h + 1]      = {ch.bbox_min[2], ch.bbox_max[2], __int_as_float(ref_of_all(ch, cn, scan, t)), __int_as_float((int)nd.axis)};
    long long s0 = cn, s1 = -1;
    if (ch.internal) {
      s0 = t.node_off + ch.start, s1 = s0 + 1;
      axes |= (ch.axis & 3) << (2 + 2 * h);
    }
    for (int k = 0; k < 2; k++) {
      long long sn = k == 0 ? s0 : s1;
      float4*   S  = Q + 2 * (2 * h + k);
      if (sn < 0) {
        S[0] = {0, 0, 0, 0};
        S[1] = {0, 0, __int_as_float(0x7ffffffe), 0};  // REF_NONE
        continue;
      }
      ythip_bvh_node g = nodes[sn];
      S[0]             = {g.bbox_min[0], g.bbox_min[1], g.bbox_max[0], g.bbox_max[1]};
      S[1]             = {g.bbox_min[2], g.bbox_max[2], __int_as_float(ref_of_all(g, sn, scan, t)), 0};
    }
  }
  Q[1].w = __int_as_float(axes);
}
__global__ void k_bake_all_leaves(const int* prims, long long n, const HostTreeDesc* T, int nt, float4* leaf) {
  long long k = (long long)blockIdx.x * BLK + threadIdx.x;
  if (k >= n) return;
  const HostTreeDesc t = T[tree_of_prim(T, nt, k)];
  if (t.kind == 0) return;  // the instance tree's primitives are instance ids (tlas_prims)
  const int       id = prims[k];
  const long long kl = k - t.prim_off;
  const float*    P  = t.positions;
  const float*    R  = t.radius;
  auto pos = [&](int v) { return float3x{P[3 * v], P[3 * v + 1], P[3 * v + 2]}; };
  if (t.kind == 3) {
    auto    p0 = pos(t.elems[3 * id]), p1 = pos(t.elems[3 * id + 1]), p2 = pos(t.elems[3 * id + 2]);
    float4* L  = leaf + t.leaf_base + 3 * kl;
    L[0] = {p0.x, p0.y, p0.z, p1.x};
    L[1] = {p1.y, p1.z, p2.x, p2.y};
    L[2] = {p2.z, __int_as_float(id), 0, 0};
  } else if (t.kind == 4) {
    auto    p0 = pos(t.elems[4 * id]), p1 = pos(t.elems[4 * id + 1]), p2 = pos(t.elems[4 * id + 2]), p3 = pos(t.elems[4 * id + 3]);
    float4* L  = leaf + t.leaf_base + 4 * kl;
    L[0] = {p0.x, p0.y, p0.z, p1.x};
    L[1] = {p1.y, p1.z, p2.x, p2.y};
    L[2] = {p2.z, p3.x, p3.y, p3.z};
    L[3] = {__int_as_float(id), 0, 0, 0};
  } else if (t.kind == 2) {
    int     a = t.elems[2 * id], b = t.elems[2 * id + 1];
    auto    p0 = pos(a), p1 = pos(b);
    float4* L  = leaf + t.leaf_base + 3 * kl;
    L[0] = {p0.x, p0.y, p0.z, p1.x};
    L[1] = {p1.y, p1.z, R ? R[a] : 0.0f, R ? R[b] : 0.0f};
    L[2] = {__int_as_float(id), 0, 0, 0};
  } else {
    int     v = t.elems[id];
    auto    p = pos(v);
    float4* L = leaf + t.leaf_base + 2 * kl;
    L[0] = {p.x, p.y, p.z, R ? R[v] : 0.0f};
    L[1] = {__int_as_float(id), 0, 0, 0};
  }
}
}  // namespace

int bake_host_trees(hipStream_t s, const ythip_bvh_node* d_nodes, long long num_nodes, const int32_t* d_prims,
    long long num_prims, const HostTreeDesc* d_table, int ntrees, float4* pairs, float4* quads, float4* leafdata,
    std::string* err) {
  if (num_nodes <= 0 || ntrees <= 0) return BUILD_OK;
  if (num_nodes > 0x7fffffffll || num_prims > 0x7fffffffll) {
    if (err) *err = "too many nodes for the batched bake";
    return BUILD_ERROR;
  }
  const int n      = (int)num_nodes;
  const int ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  int *     flag = nullptr, *scan = nullptr, *tiles = nullptr;
  auto      cleanup = [&]() {
    if (flag) (void)hipFree(flag);
    if (scan) (void)hipFree(scan);
    if (tiles) (void)hipFree(tiles);
    flag = scan = tiles = nullptr;
  };
  GCHECK(hipMalloc((void**)&flag, (size_t)n * sizeof(int)));
  GCHECK(hipMalloc((void**)&scan, ((size_t)n + 1) * sizeof(int)));
  GCHECK(hipMalloc((void**)&tiles, ((size_t)ntiles + 1) * sizeof(int)));
  hipLaunchKernelGGL(k_internal_flags, dim3(grid(n)), dim3(BLK), 0, s, d_nodes, n, flag);
  exclusive_scan(s, flag, scan, tiles, n);
  hipLaunchKernelGGL(k_bake_all_nodes, dim3(grid(n)), dim3(BLK), 0, s, d_nodes, (long long)n, scan, d_table, ntrees, pairs,
      quads);
  if (num_prims > 0)
    hipLaunchKernelGGL(k_bake_all_leaves, dim3(grid((int)num_prims)), dim3(BLK), 0, s, d_prims, num_prims, d_table, ntrees,
        leafdata);
  GCHECK(hipStreamSynchronize(s));
  GCHECK(hipGetLastError());
  cleanup();
  return BUILD_OK;
}

int refit_shape_tree(hipStream_t s, DeviceTree& tree, int kind, const int32_t* elems, const float* positions,
    const float* radius, std::string* err) {
  const int n = (int)tree.num_nodes;
  if (n <= 0) return BUILD_OK;
  if ((kind == 1 || kind == 2) && !radius) return BUILD_FALLBACK;
  int* scratch = nullptr;
  auto cleanup = [&]() {
    if (scratch) (void)hipFree(scratch);
    scratch = nullptr;
  };
  GCHECK(hipMalloc((void**)&scratch, 2 * (size_t)n * sizeof(int)));
  int *parent = scratch, *arrived = scratch + n;
  hipEvent_t e0, e1;
  GCHECK(hipEventCreate(&e0));
  GCHECK(hipEventCreate(&e1));
  GCHECK(hipEventRecord(e0, s));
  hipLaunchKernelGGL(k_refit_parents, dim3(grid(n)), dim3(BLK), 0, s, tree.nodes, n, parent, arrived);
  hipLaunchKernelGGL(k_refit, dim3(grid(n)), dim3(BLK), 0, s, tree.nodes, n, tree.prims, kind, elems, positions,
      radius, parent, arrived);
  GCHECK(hipEventRecord(e1, s));
  GCHECK(hipStreamSynchronize(s));
  GCHECK(hipGetLastError());
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  tree.build_ms = ms;
  cleanup();
  return BUILD_OK;
}

}  // namespace ytgpu
