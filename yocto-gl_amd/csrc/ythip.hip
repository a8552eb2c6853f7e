// ythip.hip — libythip.so: the C ABI of include/ythip.h over the gfx950 kernels
// of yt_kernels.h.  Host side mirrors the reference's cutrace split
// (context / scene / bvh / lights / state — libs/yocto/yocto_cutrace.cpp:385-996)
// but is written for HIP directly.
//
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -ffp-contract=off -fPIC -shared
// (-ffp-contract=off is REQUIRED: bit parity with the g++-built reference, which has no FMA
// contraction on baseline x86-64; -fno-slp-vectorize is REQUIRED too: with ROCm 7.2's SLP
// vectorizer one k_trace instantiation is miscompiled and all of them are 5-40 % slower —
// __graft_entry__.py, profiles/r03_slp_vectorizer.txt).

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <limits>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/ythip.h"
#include "yt_build.h"
#include "yt_xfer.h"
#include "yt_gpubuild.h"
#include "yt_kernels.h"
#include "yt_denoise.h"
#include "yt_order.h"

using namespace yt;

namespace {

thread_local std::string g_error;

struct DevBuf {
  void*  p = nullptr;
  size_t n = 0;
};

}  // namespace

// A host-side pool of the flat scene layout: either a copy the context owns, or a view of
// the pinned staging pool the loader filled directly (ythip_scene_staging).
template <typename T>
struct HostPool {
  std::vector<T> own;
  T*             ext = nullptr;
  size_t         n   = 0;
  T*             data() { return ext ? ext : own.data(); }
  const T*       data() const { return ext ? ext : own.data(); }
  size_t         size() const { return ext ? n : own.size(); }
  bool           empty() const { return size() == 0; }
  T&             operator[](size_t i) { return data()[i]; }
  const T&       operator[](size_t i) const { return data()[i]; }
  void           assign(const T* a, const T* b) {
    ext = nullptr, n = 0;
    own.assign(a, b);
  }
  void adopt(T* p, size_t count) {
    std::vector<T>().swap(own);
    ext = p, n = count;
  }
};

struct ythip_ctx {
  int         device     = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream     = nullptr;
  std::string err;

  std::vector<void*> scene_allocs, bvh_allocs, light_allocs, state_allocs;

  // host copies kept for BVH baking / light building
  std::vector<ythip_shape>    h_shapes;
  std::vector<ythip_instance> h_instances;
  HostPool<int32_t>           h_points, h_lines, h_triangles, h_quads;
  HostPool<float>             h_positions, h_radius;
  // scene ingest straight into the flat layout: pinned pools the
  // loader fills in place; they become the context's host copies and the DMA source
  // on-device denoiser (yt_denoise.h): working images for a w x h frame, result in dn_out
  std::vector<void*>          denoise_allocs;
  float4 *                    dn_a = nullptr, *dn_b = nullptr, *dn_gn = nullptr, *dn_ga = nullptr, *dn_out = nullptr;
  size_t                      dn_pixels    = 0;
  bool                        have_denoised = false;  // dn_out holds the filtered image of the resident state
  // longest-tile-first launch order (yt_order.hip): 0 off, 1 on (YTHIP_LPT)
  int                         lpt = 1;
  unsigned*                   d_tile_cost = nullptr;
  int*                        d_tile_perm = nullptr;
  void*                       d_sort_temp = nullptr;
  size_t                      sort_temp_bytes = 0;
  bool                        have_tile_costs = false;  // d_tile_cost holds the previous whole-slice launch's costs
  int                         lpt_age = 0;              // launches since the order was last computed
  bool                        lpt_probe = true;         // YTHIP_LPT_PROBE=0: do not split the first batch of a tile grid (see enqueue_batch)
  // pixel pool (yt_kernels.h, DState::pool_next): a launch of fewer workgroups than tiles whose lanes
  // take the next pixel of a queue when their own has had its batch.  Fills the wavefronts of scenes whose pixels
  // cost very differently (hair: +15 %) and costs a few per cent where they do not (an even scene), so the library
  // measures: once the tile costs are known, one full-size batch is timed plain, the next as a pool launch, and
  // whichever took less time per sample is kept for this state.  Results are bit-identical either way.
  int*                        d_pool_next = nullptr;     // the queue's head
  int                         pixel_pool  = 1;           // YTHIP_PIXEL_POOL: 0 never, 1 (default) measured choice, 2 always
  int                         pool_blocks = 0;           // workgroups of a pool launch (YTHIP_POOL_BLOCKS; default 16 per CU)
  int                         pool_tune   = 0;           // 0 time a plain batch next, 1 time a pool batch next, 2 waiting for both, 3 decided
  bool                        pool_on     = false;       // the decision
  hipEvent_t                  pool_ev[4]  = {nullptr, nullptr, nullptr, nullptr};  // plain begin / end, pool begin / end
  double                      pool_samples[2] = {0, 0};  // samples per pixel of the two timed launches
  float                       pool_ms[2]  = {0, 0};      // (kept for ythip_pool_info)
  int launch_blocks() const { return st.pool_next ? std::min(st.nblocks, pool_blocks) : st.nblocks; }
  std::vector<void*>          order_allocs;             // the three buffers above: they outlive a state with the same tile grid
  int                         order_tiles_x = 0, order_tiles_y = 0;
  bool                        denoise_simple = false; // YTHIP_DENOISE_SIMPLE=1: the untiled kernel for every level (cross-check)
  std::vector<void*>          staging_allocs;             // the 17 pools of ythip_scene_staging, in its order
  std::vector<size_t>         staging_caps;               // their capacities in bytes (pools are reused when they fit)
  ythip_scene                 staged      = {};
  bool                        have_staged = false;
  bool                        may_retry = false;  // opacity < 1 possible → bounce loop may exceed `bounces`
  bool                        has_volumes = false;
  bool                        all_matte   = false;  // "simple scene": matte untextured materials, triangle meshes only
  bool                        no_textures = false;  // no material references a texture
  bool                        opaque_textured = false;  // matte / glossy / reflective only, color + normal textures only, triangles + quads only
  int                         specialize  = 1;
  int                         num_cameras = 0;

  ythost::flat_bvh    h_bvh;     // as uploaded/built (reference layout) for download
  // shapes whose tree was built on the device (yt_gpubuild.hip); their slice of
  // h_bvh is downloaded on demand (ensure_host_bvh)
  std::vector<ytgpu::DeviceTree> d_trees;
  std::vector<char>              d_tree_on_host;
  int64_t                        device_build_min_prims = 16384;
  int                            bvh_builder            = 1;  // 0 host only, 1 device for large shapes
  int                            hold_policy            = 1;
  int                            peek_policy            = 1;
  // which walk k_trace / the test entries use: 0 binary, 1 wide, 2 (default) by the
  // work at hand — see use_wide()
  int     traversal_mode = 2;
  int64_t largest_tree   = 0;  // primitives of the largest tree of the resident BVH
  bool    wide_stack_ok  = true;  // the wide walk's worst-case stack depth fits the 128 entries (bake_bvh)
  bool    use_wide() const;
  ythip_build_info               build_info             = {};
  int64_t                        num_pairs = 0, num_leaf4 = 0;
  ythost::flat_lights h_lights;

  DScene ds = {};
  DState st = {};
  bool   have_scene = false, have_bvh = false, have_lights = false, have_state = false;
  bool   state_bound = false;
  int    samples     = 0;

  // measurement
  int                                          prof_mode = 0;
  unsigned long long*                          d_counters = nullptr;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  std::vector<std::pair<int, int>>             ev_used;  // (pool index, kind 0 extend / 1 shade)
  size_t                                       ev_next = 0;
  ythip_stats                                  stats   = {};
  float4 *nee_a = nullptr, *nee_b = nullptr, *nee_c = nullptr, *nee_d = nullptr, *nee_e = nullptr;  // deferred NEE (per slot)
  float4*                                      nhit_a   = nullptr;
  int*                                         nhit_e   = nullptr;

  int*               d_stop        = nullptr;  // device-visible cancel word polled by the kernels
  hipEvent_t         done_event    = nullptr;
  // Cancellation by generation: every batch gets a number, the kernels stop when the
  // word at d_stop EQUALS their batch's number, ythip_cancel writes the number of the batch in
  // flight.  Nothing ever has to lower the flag, so a cancel that races with the next enqueue can
  // neither be lost into it nor leak into it (the boolean of round 2 could end up raised with
  // nobody left to lower it: every later tile then exited at once while `samples` kept advancing).
  std::atomic<int>   stop_gen{0};
  int*               stop_host     = nullptr;  // pinned host word ythip_cancel stores the batch number into ...
  const int*         stop_host_dev = nullptr;  // ... and its device address (the kernels relay it into d_stop)
  ytx::Bounce        xfer;  // every host <-> device byte goes through pinned memory the library owns (yt_xfer.h)
  bool               last_launch_fast = false;  // the last k_trace launch ran the tolerance-mode kernels (yt_fast.hip)
};

// yt_fast.hip: the tolerance-mode kernels (same source, -DYT_FAST, own namespace); 0 = launched, 1 = no such kernel
extern "C" int ythip_fast_launch(void* stream, int blocks, const void* ds, const void* st, const void* kp, int lp, int cls);

static void drop_staging_views(ythip_ctx* ctx) {
  // host pools that view the staging memory go with it: the scene they belong to is no longer
  // resident as far as the host-side builders are concerned (a new upload must follow)
  bool viewed = false;
  for (auto* pool : {&ctx->h_points, &ctx->h_lines, &ctx->h_triangles, &ctx->h_quads})
    if (pool->ext) pool->ext = nullptr, pool->n = 0, viewed = true;
  for (auto* pool : {&ctx->h_positions, &ctx->h_radius})
    if (pool->ext) pool->ext = nullptr, pool->n = 0, viewed = true;
  if (viewed) ctx->have_scene = ctx->have_bvh = ctx->have_lights = false;
  ctx->have_staged = false;
  ctx->staged      = {};
}
static void free_staging(ythip_ctx* ctx) {
  drop_staging_views(ctx);
  for (auto p : ctx->staging_allocs)
    if (p) (void)hipHostFree(p);
  ctx->staging_allocs.clear();
  ctx->staging_caps.clear();
}

// The wide walk halves a ray's chain of dependent fetches and costs a little more
// arithmetic per level.  It pays when the waves have the machine to themselves
// (small slices: one GPU of eight, previews) and on large trees; on scenes made of
// tiny trees the 4-slot records are mostly empty.  Measured in DESIGN.md §6.
bool ythip_ctx::use_wide() const {
  if (!wide_stack_ok) return false;  // trees too deep for the wide walk's pushes (bake_bvh): the binary walk
  if (traversal_mode != 2) return traversal_mode == 1;
  return largest_tree >= 64;
}

namespace {

int fail(ythip_ctx* ctx, int code, const char* fmt, ...) {
  char    buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  g_error = buf;
  return code;
}

#define HIPCHECK(ctx, call)                                                                        \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(ctx, YTHIP_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
          __LINE__);                                                                               \
  } while (0)

void free_all(std::vector<void*>& v) {
  for (auto p : v)
    if (p) (void)hipFree(p);
  v.clear();
}

template <typename T>
int dalloc(ythip_ctx* ctx, std::vector<void*>& pool, T** out, size_t count) {
  *out = nullptr;
  if (count == 0) count = 1;  // keep pointers valid
  void* p = nullptr;
  HIPCHECK(ctx, hipMalloc(&p, count * sizeof(T)));
  pool.push_back(p);
  *out = (T*)p;
  return YTHIP_OK;
}
template <typename T>
int dupload(ythip_ctx* ctx, std::vector<void*>& pool, const T** out, const T* src, size_t count) {
  T*  d  = nullptr;
  int rc = dalloc(ctx, pool, &d, count);
  if (rc) return rc;
  if (count && src) HIPCHECK(ctx, ctx->xfer.h2d(ctx->stream, d, src, count * sizeof(T)));
  *out = d;
  return YTHIP_OK;
}

int grid_for(long long n) { return (int)((n + YT_BLOCK - 1) / YT_BLOCK); }

KParams to_kparams(const ythip_ctx* ctx, const ythip_params* p) {
  KParams k;
  k.camera     = p->camera;
  k.sampler    = p->sampler;
  k.falsecolor = p->falsecolor;
  k.bounces    = p->bounces;
  k.clamp      = p->clamp;
  k.nocaustics = p->nocaustics;
  k.envhidden  = p->envhidden;
  k.tentfilter = p->tentfilter;
  k.has_env    = ctx->ds.num_environments > 0;
  k.hold       = ctx->hold_policy;
  k.peek       = ctx->peek_policy;
  return k;
}

// Bake the reference-layout trees into the device layout of yt_bvh.h:
//   * pairs:    one 64-B record per internal node holding BOTH children
//               {bbox, ref} (+ the parent's split axis) — the two nodes the
//               reference pops one after the other arrive in one fetch
//   * leafdata: primitives pre-gathered in leaf order (ids come from
//               `primitives[]`, so hit indices are unaffected)
//   * tinst:    per-instance inverse frame + BLAS root {bbox, ref}
// Trees live either on the host (ctx->h_bvh: uploaded, or built by yt_build.h)
// or on the device (ctx->d_trees[s], built by yt_gpubuild.hip; their slice of
// h_bvh is filled lazily by ensure_host_bvh()).  Host trees are baked here and
// uploaded slice by slice, device trees are baked by kernels; both produce the
// same bytes (tests/test_gpu_build.py).
__global__ void __launch_bounds__(YT_BLOCK) k_gather_tinst(const DInstanceT* tinst, const int* tlas_prims, int n, int ninst,
    DInstanceT* out) {
  const int k = (int)(blockIdx.x * YT_BLOCK + threadIdx.x);
  if (k >= n) return;
  const int inst = tlas_prims[k];
  if (inst < 0 || inst >= ninst) {  // (an uploaded tree with a bad instance id: an empty record, never entered)
    DInstanceT e = {};
    e.root_ref = REF_NONE, e.instance = -1;
    out[k]     = e;
    return;
  }
  DInstanceT r = tinst[inst];
  r.instance   = inst;
  out[k]       = r;
}

int bake_bvh(ythip_ctx* ctx) {
  auto& b        = ctx->h_bvh;
  int   nshapes  = (int)ctx->h_shapes.size();
  int   ntrees   = (int)b.node_offset.size() - 1;
  if (ntrees != nshapes + 1)
    return fail(ctx, YTHIP_ERR_INVALID, "bvh has %d trees, scene has %d shapes (+1 expected)", ntrees, nshapes);
  free_all(ctx->bvh_allocs);
  auto on_device = [&](int t) { return t < (int)ctx->d_trees.size() && ctx->d_trees[t].nodes != nullptr; };

  const auto&          nodes = b.nodes;
  std::vector<int64_t> leaf_base(nshapes, 0);
  std::vector<int>     strides(nshapes, 0);
  int64_t              nleaf4 = 0;
  for (int s = 0; s < nshapes; s++) {
    leaf_base[s] = nleaf4;
    int kind     = ythost::kind_bvh(ctx->h_shapes[s]);
    strides[s]   = kind == KIND_TRIANGLES ? 3 : (kind == KIND_QUADS ? 4 : (kind == KIND_LINES ? 3 : 2));
    nleaf4 += (b.prim_offset[s + 1] - b.prim_offset[s]) * strides[s];
  }
  if (nleaf4 > 0x7fffff00ll || (int64_t)nodes.size() > 0x7fffffffll || b.prim_offset[ntrees] > 0x0fffffffll)
    return fail(ctx, YTHIP_ERR_INVALID, "bvh too large for 32-bit device references");
  // sibling-pair ids: one per internal node, in node order, all trees
  std::vector<int64_t> pair_base(ntrees + 1, 0);
  for (int t = 0; t < ntrees; t++) {
    int64_t n = 0;
    if (on_device(t)) {
      n = (ctx->d_trees[t].num_nodes - 1) / 2;  // strictly binary tree
    } else {
      for (int64_t k = b.node_offset[t]; k < b.node_offset[t + 1]; k++) n += nodes[k].internal ? 1 : 0;
    }
    pair_base[t + 1] = pair_base[t] + n;
  }
  const int64_t npairs = pair_base[ntrees];
  if (npairs >= (int64_t)REF_INST) return fail(ctx, YTHIP_ERR_INVALID, "bvh too large for 32-bit device references");

  const int LEAF_PAD = 8;  // the triangle loop fetches two primitives per round trip
  float4 *  d_pairs = nullptr, *d_leaf = nullptr, *d_quads = nullptr;
  int       rc;
  if ((rc = dalloc(ctx, ctx->bvh_allocs, &d_pairs, (size_t)npairs * 4 + 4))) return rc;
  if ((rc = dalloc(ctx, ctx->bvh_allocs, &d_quads, (size_t)npairs * 8 + 8))) return rc;
  if ((rc = dalloc(ctx, ctx->bvh_allocs, &d_leaf, (size_t)nleaf4 + LEAF_PAD))) return rc;
  HIPCHECK(ctx, hipMemsetAsync(d_pairs + 4 * npairs, 0, 4 * sizeof(float4), ctx->stream));
  HIPCHECK(ctx, hipMemsetAsync(d_leaf + nleaf4, 0, LEAF_PAD * sizeof(float4), ctx->stream));

  struct Root {
    float bmin[3], bmax[3];
    int   ref;
  };
  std::vector<Root>                roots(ntrees, Root{{0, 0, 0}, {0, 0, 0}, REF_NONE});
  std::vector<std::vector<float4>> staging;  // host-baked slices (YTHIP_HOST_BAKE=1)
  bool                             bad_leaf = false;
  // host-resident trees are baked by the device, all in one go (yt_gpubuild.hip: bake_host_trees)
  const bool host_bake = [] { const char* e = std::getenv("YTHIP_HOST_BAKE"); return e && std::atoi(e) != 0; }();
  struct Run {
    int64_t node_begin, node_end, prim_begin, prim_end, cnode, cprim;
  };
  std::vector<ytgpu::HostTreeDesc> table;
  std::vector<Run>                 runs;
  int64_t                          compact_nodes = 0, compact_prims = 0;

  for (int t = 0; t < ntrees; t++) {
    const bool    blas  = t < nshapes;
    const int64_t nn    = b.node_offset[t + 1] - b.node_offset[t];
    const int64_t np    = b.prim_offset[t + 1] - b.prim_offset[t];
    if (on_device(t) && !blas) {  // the instance tree, built on the device: pairs + quads, no leaf data
      float       root7[7];
      std::string err;
      if (ytgpu::bake_shape_tree(ctx->stream, ctx->d_trees[t], 0, nullptr, nullptr, nullptr, pair_base[t], 0, 0, d_pairs,
              d_quads, d_leaf, root7, &err) != ytgpu::BUILD_OK)
        return fail(ctx, YTHIP_ERR_HIP, "device instance-tree bake failed: %s", err.c_str());
      for (int c = 0; c < 3; c++) roots[t].bmin[c] = root7[c], roots[t].bmax[c] = root7[3 + c];
      std::memcpy(&roots[t].ref, &root7[6], 4);
      continue;
    }
    if (on_device(t)) {
      const auto& sh   = ctx->h_shapes[t];
      int         kind = ythost::kind_bvh(sh);
      const int*  el   = kind == KIND_TRIANGLES ? ctx->ds.triangles + 3 * sh.triangles_offset
                         : kind == KIND_QUADS   ? ctx->ds.quads + 4 * sh.quads_offset
                         : kind == KIND_LINES   ? ctx->ds.lines + 2 * sh.lines_offset
                                                : ctx->ds.points + sh.points_offset;
      float       root7[7];
      std::string err;
      if (ytgpu::bake_shape_tree(ctx->stream, ctx->d_trees[t], kind, el, ctx->ds.positions + 3 * sh.positions_offset,
              sh.radius_offset >= 0 && sh.num_radius ? ctx->ds.radius + sh.radius_offset : nullptr, pair_base[t],
              b.prim_offset[t], leaf_base[t], d_pairs, d_quads, d_leaf, root7, &err) != ytgpu::BUILD_OK)
        return fail(ctx, YTHIP_ERR_HIP, "device bvh bake failed: %s", err.c_str());
      for (int c = 0; c < 3; c++) roots[t].bmin[c] = root7[c], roots[t].bmax[c] = root7[3 + c];
      std::memcpy(&roots[t].ref, &root7[6], 4);
      continue;
    }
    // ---- a host-resident tree (yt_build.h, or uploaded by the caller) ---------------------
    if (!host_bake) {
      // baked on the device together with every other host tree (ytgpu::bake_host_trees): here only
      // its descriptor, its place in the compact upload and its root
      ytgpu::HostTreeDesc d = {};
      d.node_off      = compact_nodes;
      d.prim_off      = compact_prims;
      d.pair_base     = pair_base[t];
      d.ref_prim_base = blas ? b.prim_offset[t] : 0;
      d.leaf_base     = blas ? leaf_base[t] : 0;
      d.kind          = 0;
      if (blas) {
        const auto& sh = ctx->h_shapes[t];
        d.kind         = ythost::kind_bvh(sh);
        d.elems        = d.kind == KIND_TRIANGLES ? ctx->ds.triangles + 3 * sh.triangles_offset
                         : d.kind == KIND_QUADS   ? ctx->ds.quads + 4 * sh.quads_offset
                         : d.kind == KIND_LINES   ? ctx->ds.lines + 2 * sh.lines_offset
                         : d.kind == KIND_POINTS  ? ctx->ds.points + sh.points_offset
                                                  : nullptr;
        d.positions    = ctx->ds.positions + 3 * sh.positions_offset;
        d.radius       = sh.radius_offset >= 0 && sh.num_radius ? ctx->ds.radius + sh.radius_offset : nullptr;
        if (d.kind == KIND_NONE) d.kind = 0;  // (an element-less shape: a one-node tree with an empty leaf)
      }
      table.push_back(d);
      if (!runs.empty() && runs.back().node_end == b.node_offset[t] && runs.back().prim_end == b.prim_offset[t])
        runs.back().node_end = b.node_offset[t + 1], runs.back().prim_end = b.prim_offset[t + 1];
      else
        runs.push_back({b.node_offset[t], b.node_offset[t + 1], b.prim_offset[t], b.prim_offset[t + 1], compact_nodes, compact_prims});
      compact_nodes += nn, compact_prims += np;
      for (int64_t k = b.node_offset[t]; k < b.node_offset[t + 1]; k++)
        if (!nodes[k].internal && (nodes[k].num < 0 || nodes[k].num > 7)) bad_leaf = true;
      if (nn > 0) {
        const auto& root = nodes[b.node_offset[t]];
        roots[t].ref     = root.internal ? (int32_t)pair_base[t]
                                         : (int32_t)(0x80000000u | ((uint32_t)(root.num & 7) << 28) |
                                                     (uint32_t)((blas ? b.prim_offset[t] : 0) + root.start));
        for (int c = 0; c < 3; c++) roots[t].bmin[c] = root.bbox_min[c], roots[t].bmax[c] = root.bbox_max[c];
      }
      continue;
    }
    // ---- YTHIP_HOST_BAKE=1: the same records assembled on the host (the cross-check of the kernels) ----
    if (blas && np > 0) {
      const auto&  sh     = ctx->h_shapes[t];
      int          kind   = ythost::kind_bvh(sh);
      int          stride = strides[t];
      const float* P      = ctx->h_positions.data() + 3 * sh.positions_offset;
      const float* R      = sh.radius_offset >= 0 ? ctx->h_radius.data() + sh.radius_offset : nullptr;
      auto         pos    = [&](int v) { return float3{P[3 * v], P[3 * v + 1], P[3 * v + 2]}; };
      staging.emplace_back((size_t)np * stride, float4{0, 0, 0, 0});
      auto& leaf = staging.back();
      for (int64_t k = 0; k < np; k++) {
        int     id = b.prims[b.prim_offset[t] + k];
        float4* L  = leaf.data() + k * stride;
        if (kind == KIND_TRIANGLES) {
          const int* tr = ctx->h_triangles.data() + 3 * (sh.triangles_offset + id);
          auto       p0 = pos(tr[0]), p1 = pos(tr[1]), p2 = pos(tr[2]);
          L[0] = {p0.x, p0.y, p0.z, p1.x};
          L[1] = {p1.y, p1.z, p2.x, p2.y};
          L[2] = {p2.z, __builtin_bit_cast(float, id), 0, 0};
        } else if (kind == KIND_QUADS) {
          const int* q  = ctx->h_quads.data() + 4 * (sh.quads_offset + id);
          auto       p0 = pos(q[0]), p1 = pos(q[1]), p2 = pos(q[2]), p3 = pos(q[3]);
          L[0] = {p0.x, p0.y, p0.z, p1.x};
          L[1] = {p1.y, p1.z, p2.x, p2.y};
          L[2] = {p2.z, p3.x, p3.y, p3.z};
          L[3] = {__builtin_bit_cast(float, id), 0, 0, 0};
        } else if (kind == KIND_LINES) {
          const int* l  = ctx->h_lines.data() + 2 * (sh.lines_offset + id);
          auto       p0 = pos(l[0]), p1 = pos(l[1]);
          L[0] = {p0.x, p0.y, p0.z, p1.x};
          L[1] = {p1.y, p1.z, R ? R[l[0]] : 0.0f, R ? R[l[1]] : 0.0f};
          L[2] = {__builtin_bit_cast(float, id), 0, 0, 0};
        } else if (kind == KIND_POINTS) {
          int  v = ctx->h_points[sh.points_offset + id];
          auto p = pos(v);
          L[0]   = {p.x, p.y, p.z, R ? R[v] : 0.0f};
          L[1]   = {__builtin_bit_cast(float, id), 0, 0, 0};
        }
      }
      HIPCHECK(ctx, ctx->xfer.h2d(ctx->stream, d_leaf + leaf_base[t], leaf.data(), leaf.size() * sizeof(float4)));
    }
    // pair ids of this tree, in node order
    std::vector<int32_t> pair_id((size_t)nn, -1);
    int64_t              next = pair_base[t];
    for (int64_t n = 0; n < nn; n++)
      if (nodes[b.node_offset[t] + n].internal) pair_id[n] = (int32_t)next++;
    auto ref_of = [&](int64_t ln) -> int32_t {  // ln: tree-local node index
      const auto& node = nodes[b.node_offset[t] + ln];
      if (node.internal) return pair_id[ln];
      if (node.num < 0 || node.num > 7) bad_leaf = true;
      // BLAS leaves address leaf data by global primitive index, TLAS leaves index tlas_prims
      int64_t first = (blas ? b.prim_offset[t] : 0) + node.start;
      return (int32_t)(0x80000000u | ((uint32_t)(node.num & 7) << 28) | (uint32_t)first);
    };
    const int64_t np_t = pair_base[t + 1] - pair_base[t];
    if (np_t > 0) {
      staging.emplace_back((size_t)np_t * 4, float4{0, 0, 0, 0});
      auto& pairs = staging.back();
      for (int64_t n = 0; n < nn; n++) {
        const auto& node = nodes[b.node_offset[t] + n];
        if (!node.internal) continue;
        float4* P = pairs.data() + 4 * (size_t)(pair_id[n] - pair_base[t]);
        for (int c = 0; c < 2; c++) {
          int64_t     lc = node.start + c;
          const auto& ch = nodes[b.node_offset[t] + lc];
          P[2 * c]       = {ch.bbox_min[0], ch.bbox_min[1], ch.bbox_max[0], ch.bbox_max[1]};
          P[2 * c + 1]   = {ch.bbox_min[2], ch.bbox_max[2], __builtin_bit_cast(float, ref_of(lc)),
                __builtin_bit_cast(float, (int32_t)node.axis)};
        }
      }
      HIPCHECK(ctx, ctx->xfer.h2d(ctx->stream, d_pairs + 4 * pair_base[t], pairs.data(), pairs.size() * sizeof(float4)));
      // grandchildren ("quad") records of the wide walk, same ids: slots 0,1 = children
      // of child 0 (or child 0 itself when it is a leaf, slot 1 empty), slots 2,3
      // likewise for child 1
      staging.emplace_back((size_t)np_t * 8, float4{0, 0, 0, 0});
      auto& quads = staging.back();
      for (int64_t n = 0; n < nn; n++) {
        const auto& node = nodes[b.node_offset[t] + n];
        if (!node.internal) continue;
        float4* Qr   = quads.data() + 8 * (size_t)(pair_id[n] - pair_base[t]);
        int     axes = node.axis & 3;
        for (int h = 0; h < 2; h++) {
          int64_t     lc = node.start + h;
          const auto& ch = nodes[b.node_offset[t] + lc];
          int64_t     slot_node[2] = {lc, -1};
          if (ch.internal) {
            slot_node[0] = ch.start, slot_node[1] = ch.start + 1;
            axes |= (ch.axis & 3) << (2 + 2 * h);
          }
          for (int k = 0; k < 2; k++) {
            float4* S = Qr + 2 * (2 * h + k);
            if (slot_node[k] < 0) {
              S[0] = {0, 0, 0, 0};
              S[1] = {0, 0, __builtin_bit_cast(float, (int32_t)REF_NONE), 0};
              continue;
            }
            const auto& g = nodes[b.node_offset[t] + slot_node[k]];
            S[0]          = {g.bbox_min[0], g.bbox_min[1], g.bbox_max[0], g.bbox_max[1]};
            S[1]          = {g.bbox_min[2], g.bbox_max[2], __builtin_bit_cast(float, ref_of(slot_node[k])), 0};
          }
        }
        Qr[1].w = __builtin_bit_cast(float, (int32_t)axes);
      }
      HIPCHECK(ctx, ctx->xfer.h2d(ctx->stream, d_quads + 8 * pair_base[t], quads.data(), quads.size() * sizeof(float4)));
    }
    if (nn > 0) {
      const auto& root = nodes[b.node_offset[t]];
      roots[t].ref     = ref_of(0);
      for (int c = 0; c < 3; c++) roots[t].bmin[c] = root.bbox_min[c], roots[t].bmax[c] = root.bbox_max[c];
    }
  }
  if (!table.empty()) {
    // one compact upload of the host trees' nodes and primitives (run by run: host trees that sit next
    // to each other in h_bvh travel together), the descriptor table, one set of launches for all of them
    ythip_bvh_node*      d_nodes_c = nullptr;
    int32_t*             d_prims_c = nullptr;
    ytgpu::HostTreeDesc* d_table   = nullptr;
    std::vector<void*>   tmp;
    if ((rc = dalloc(ctx, tmp, &d_nodes_c, (size_t)compact_nodes)) || (rc = dalloc(ctx, tmp, &d_prims_c, (size_t)compact_prims)) ||
        (rc = dalloc(ctx, tmp, &d_table, table.size()))) {
      free_all(tmp);
      return rc;
    }
    hipError_t e = ctx->xfer.h2d(ctx->stream, d_table, table.data(), table.size() * sizeof(ytgpu::HostTreeDesc));
    for (auto& r : runs) {
      if (e == hipSuccess && r.node_end > r.node_begin)
        e = ctx->xfer.h2d(ctx->stream, d_nodes_c + r.cnode, b.nodes.data() + r.node_begin,
            (size_t)(r.node_end - r.node_begin) * sizeof(ythip_bvh_node));
      if (e == hipSuccess && r.prim_end > r.prim_begin)
        e = ctx->xfer.h2d(ctx->stream, d_prims_c + r.cprim, b.prims.data() + r.prim_begin,
            (size_t)(r.prim_end - r.prim_begin) * sizeof(int32_t));
    }
    std::string err;
    int brc = e == hipSuccess ? ytgpu::bake_host_trees(ctx->stream, d_nodes_c, compact_nodes, d_prims_c, compact_prims, d_table,
                                    (int)table.size(), d_pairs, d_quads, d_leaf, &err)
                              : ytgpu::BUILD_ERROR;
    free_all(tmp);
    if (brc != ytgpu::BUILD_OK)
      return fail(ctx, YTHIP_ERR_HIP, "bvh bake failed: %s", e != hipSuccess ? hipGetErrorString(e) : err.c_str());
  }
  // One 128-entry stack serves the TLAS walk, the TLAS-leaf continuation entries, the exit
  // marker and the BLAS walk (yt_bvh.h), where the reference has 128 entries PER LEVEL
  // (yocto_bvh.cpp:470, 560): refuse trees so deep that the shared stack could overflow
  // where the reference's would not (a DFS holds at most one pending sibling per level;
  // + 3 continuation entries of a 4-instance TLAS leaf + the exit marker).
  {
    auto depth_of = [&](int t) -> int {
      if (on_device(t)) return ctx->d_trees[t].depth;
      const int64_t nn = b.node_offset[t + 1] - b.node_offset[t];
      if (nn <= 0) return 0;
      int                                  best = 0;
      std::vector<std::pair<int64_t, int>> todo = {{0, 1}};
      while (!todo.empty()) {
        auto [n, dpt] = todo.back();
        todo.pop_back();
        best = std::max(best, dpt);
        const auto& node = nodes[b.node_offset[t] + n];
        if (node.internal) todo.push_back({node.start, dpt + 1}), todo.push_back({node.start + 1, dpt + 1});
      }
      return best;
    };
    int deepest_blas = 0;
    for (int t = 0; t < nshapes; t++) deepest_blas = std::max(deepest_blas, depth_of(t));
    const int tlas_depth = depth_of(nshapes);
    if (tlas_depth + deepest_blas + 5 > 128)
      return fail(ctx, YTHIP_ERR_INVALID,
          "bvh too deep for the shared traversal stack: instance tree %d levels + deepest shape tree %d levels + 5 > 128",
          tlas_depth, deepest_blas);
    // The wide walk advances two levels per step and can leave up to THREE pending siblings per
    // step: 3 * ceil(depth / 2) entries per tree.  Trees between that bound and the
    // binary one are walked binary — the reference renders them, so they are not refused.
    auto wide_need      = [](int depth) { return 3 * ((depth + 1) / 2); };
    ctx->wide_stack_ok = wide_need(tlas_depth) + wide_need(deepest_blas) + 5 <= 128;
  }
  // per-instance traversal records
  std::vector<DInstanceT> tinst(ctx->h_instances.size());
  for (size_t k = 0; k < tinst.size(); k++) {
    const auto& inst = ctx->h_instances[k];
    auto&       ti   = tinst[k];
    ti               = DInstanceT{};
    ythost::inverse_frame_nonrigid(inst.frame, ti.inv);
    int s       = inst.shape;
    ti.root_ref = roots[s].ref;
    for (int c = 0; c < 3; c++) ti.root_bmin[c] = roots[s].bmin[c], ti.root_bmax[c] = roots[s].bmax[c];
    ti.kind      = ythost::kind_bvh(ctx->h_shapes[s]);
    ti.leaf_bias = (int)(leaf_base[s] - b.prim_offset[s] * strides[s]);
    ti.shape     = s;
  }
  ctx->ds.tlas_ref  = roots[nshapes].ref;
  ctx->ds.tlas_bmin = {roots[nshapes].bmin[0], roots[nshapes].bmin[1], roots[nshapes].bmin[2]};
  ctx->ds.tlas_bmax = {roots[nshapes].bmax[0], roots[nshapes].bmax[1], roots[nshapes].bmax[2]};
  if (bad_leaf) return fail(ctx, YTHIP_ERR_INVALID, "bvh leaf with more than 7 primitives (reference builds <= 4)");
  ctx->ds.pairs    = d_pairs;
  ctx->ds.wide     = d_quads;
  ctx->ds.leafdata = d_leaf;
  ctx->largest_tree = 0;
  for (int t = 0; t < ntrees; t++) {
    const int64_t np = b.prim_offset[t + 1] - b.prim_offset[t];
    if (np > ctx->largest_tree) ctx->largest_tree = np;
  }
  ctx->num_pairs   = npairs;
  ctx->num_leaf4   = nleaf4;
  if (on_device(nshapes)) {
    ctx->ds.tlas_prims = ctx->d_trees[nshapes].prims;  // (owned by the device tree, which outlives the bake)
  } else if ((rc = dupload(ctx, ctx->bvh_allocs, &ctx->ds.tlas_prims, b.prims.data() + b.prim_offset[nshapes],
                  (size_t)(b.prim_offset[nshapes + 1] - b.prim_offset[nshapes]))))
    return rc;
  if ((rc = dupload(ctx, ctx->bvh_allocs, &ctx->ds.tinst, tinst.data(), tinst.size()))) return rc;
  {  // the records once more, in TLAS-leaf order: entering the k-th instance of a leaf is then ONE dependent fetch
    const int64_t ntl = b.prim_offset[nshapes + 1] - b.prim_offset[nshapes];
    DInstanceT*   d_tl = nullptr;
    if ((rc = dalloc(ctx, ctx->bvh_allocs, &d_tl, (size_t)ntl))) return rc;
    if (ntl > 0)
      hipLaunchKernelGGL(k_gather_tinst, dim3(grid_for(ntl)), dim3(YT_BLOCK), 0, ctx->stream, ctx->ds.tinst, ctx->ds.tlas_prims,
          (int)ntl, (int)tinst.size(), d_tl);
    HIPCHECK(ctx, hipGetLastError());
    ctx->ds.tinst_leaf = d_tl;
  }
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));  // host staging vectors die here
  ctx->have_bvh = true;
  return YTHIP_OK;
}

void free_device_trees(ythip_ctx* ctx) {
  for (auto& t : ctx->d_trees) ytgpu::free_tree(&t);
  ctx->d_trees.clear();
}

// Fill the slices of ctx->h_bvh that belong to device-built trees (download).
int ensure_host_bvh(ythip_ctx* ctx) {
  auto& b = ctx->h_bvh;
  for (size_t t = 0; t < ctx->d_trees.size(); t++) {
    auto& dt = ctx->d_trees[t];
    if (!dt.nodes || ctx->d_tree_on_host[t]) continue;
    HIPCHECK(ctx, ctx->xfer.d2h(ctx->stream, b.nodes.data() + b.node_offset[t], dt.nodes, (size_t)dt.num_nodes * sizeof(ythip_bvh_node)));
    HIPCHECK(ctx, ctx->xfer.d2h(ctx->stream, b.prims.data() + b.prim_offset[t], dt.prims, (size_t)dt.num_prims * sizeof(int32_t)));
    ctx->d_tree_on_host[t] = 1;
  }
  return YTHIP_OK;
}

// make_scene_bvh (yocto_bvh.cpp:364-396) with the large shapes built on the
// device (yt_gpubuild.hip) and everything else — small shapes, the instance
// tree — by the host builder of yt_build.h.  Same trees either way.
int build_bvh_mixed(ythip_ctx* ctx, const ythip_scene& sc, bool highquality, bool use_device) {
  auto t_start = std::chrono::steady_clock::now();
  free_device_trees(ctx);
  auto& out = ctx->h_bvh;
  out       = ythost::flat_bvh{};
  ctx->build_info = {};
  int  nshapes = sc.num_shapes;
  ctx->d_trees.assign(nshapes, ytgpu::DeviceTree{});
  ctx->d_tree_on_host.assign(nshapes, 0);
  auto roots = std::vector<ythost::bbox>(nshapes);
  auto empty = std::vector<char>(nshapes, 1);
  auto prims_of = [&](const ythip_shape& sh) -> int64_t {
    int kind = ythost::kind_bvh(sh);
    return kind == KIND_POINTS ? sh.num_points : kind == KIND_LINES ? sh.num_lines
           : kind == KIND_TRIANGLES ? sh.num_triangles : kind == KIND_QUADS ? sh.num_quads : 0;
  };
  // Shapes below the device threshold are built by a pool of host threads, as the reference does
  // (make_scene_bvh's parallel_for over the shapes: yocto_bvh.cpp:369-378), WHILE this thread
  // drives the device builds of the large ones.  Every tree is independent; they are concatenated
  // in shape order afterwards, so the flat layout does not depend on who finished first.
  std::vector<ythost::tree> host_trees(nshapes);
  std::vector<char>         on_host(nshapes, 0);
  for (int k = 0; k < nshapes; k++) on_host[k] = !(use_device && prims_of(sc.shapes[k]) >= ctx->device_build_min_prims);
  // what the worker threads may touch: the shapes that were host shapes BEFORE the threads started.  (they
  // used to test on_host[], which this thread edits when a device build falls back — a worker could then build and
  // assign the same host_trees[k] concurrently.)  Fallbacks belong to this thread alone.
  const std::vector<char> worker_shapes = on_host;
  std::atomic<int>         next_shape{0};
  std::vector<std::thread> workers;
  {
    int64_t host_count = 0, host_prims = 0;
    for (int k = 0; k < nshapes; k++)
      if (on_host[k]) host_count++, host_prims += prims_of(sc.shapes[k]);
    unsigned want = host_prims > 50000 ? std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), (unsigned)host_count) : 0;
    if (const char* e = std::getenv("YTHIP_BUILD_THREADS")) want = (unsigned)std::max(0, std::atoi(e));
    auto work = [&]() {
      for (int k; (k = next_shape.fetch_add(1)) < nshapes;)
        if (worker_shapes[k]) host_trees[k] = ythost::make_shape_bvh(sc, sc.shapes[k], highquality);
    };
    for (unsigned t = 0; t < want; t++) workers.emplace_back(work);
    ctx->build_info.host_threads = (int)want;
    // (no pool: the host shapes are built below, on this thread, after the device ones)
  }
  auto join_workers = [&]() {
    for (auto& w : workers) w.join();
    workers.clear();
  };
  // the device builds (they synchronise once per tree level: the host threads run meanwhile)
  for (int k = 0; k < nshapes; k++) {
    if (on_host[k]) continue;
    const auto& sh    = sc.shapes[k];
    int         kind  = ythost::kind_bvh(sh);
    int64_t     nprim = prims_of(sh);
    const int*  el    = kind == KIND_TRIANGLES ? ctx->ds.triangles + 3 * sh.triangles_offset
                        : kind == KIND_QUADS   ? ctx->ds.quads + 4 * sh.quads_offset
                        : kind == KIND_LINES   ? ctx->ds.lines + 2 * sh.lines_offset
                                               : ctx->ds.points + sh.points_offset;
    std::string err;
    int rc = ytgpu::build_shape_tree(ctx->stream, kind, el, ctx->ds.positions + 3 * sh.positions_offset,
        sh.radius_offset >= 0 && sh.num_radius ? ctx->ds.radius + sh.radius_offset : nullptr, nprim, highquality,
        &ctx->d_trees[k], &err);
    if (rc == ytgpu::BUILD_ERROR) {
      join_workers();
      return fail(ctx, YTHIP_ERR_HIP, "device bvh build failed: %s", err.c_str());
    }
    if (rc == ytgpu::BUILD_OK) {
      auto& dt = ctx->d_trees[k];
      ythip_bvh_node root;
      if (auto e = ctx->xfer.d2h(ctx->stream, &root, dt.nodes, sizeof(root)); e != hipSuccess) {
        join_workers();
        return fail(ctx, YTHIP_ERR_HIP, "device bvh root readback failed: %s", hipGetErrorString(e));
      }
      roots[k].min = {root.bbox_min[0], root.bbox_min[1], root.bbox_min[2]};
      roots[k].max = {root.bbox_max[0], root.bbox_max[1], root.bbox_max[2]};
      empty[k]     = 0;
      ctx->build_info.device_trees += 1;
      ctx->build_info.device_prims += nprim;
      ctx->build_info.device_ms += dt.build_ms;
      ctx->build_info.max_depth = std::max(ctx->build_info.max_depth, dt.depth);
    } else {  // (signed-zero tie: only the serial builder knows the answer)
      ctx->build_info.fallbacks += 1;
      host_trees[k] = ythost::make_shape_bvh(sc, sh, highquality);
      on_host[k]    = 2;
    }
  }
  if (workers.empty()) {
    for (int k = 0; k < nshapes; k++)
      if (on_host[k] == 1) host_trees[k] = ythost::make_shape_bvh(sc, sc.shapes[k], highquality);
  }
  join_workers();
  // concatenate in shape order
  for (int k = 0; k < nshapes; k++) {
    out.node_offset.push_back((int64_t)out.nodes.size());
    out.prim_offset.push_back((int64_t)out.prims.size());
    if (!on_host[k]) {  // device tree: its slice is filled by ensure_host_bvh() on demand
      auto& dt = ctx->d_trees[k];
      out.nodes.resize(out.nodes.size() + (size_t)dt.num_nodes);
      out.prims.resize(out.prims.size() + (size_t)dt.num_prims);
      continue;
    }
    auto& t = host_trees[k];
    if (!t.nodes.empty()) {
      empty[k]     = 0;
      auto& n      = t.nodes[0];
      roots[k].min = {n.bbox_min[0], n.bbox_min[1], n.bbox_min[2]};
      roots[k].max = {n.bbox_max[0], n.bbox_max[1], n.bbox_max[2]};
    }
    out.nodes.insert(out.nodes.end(), t.nodes.begin(), t.nodes.end());
    out.prims.insert(out.prims.end(), t.prims.begin(), t.prims.end());
    ythost::tree().nodes.swap(t.nodes);
    ctx->build_info.host_trees += 1;
  }
  // the instance tree — yocto_bvh.cpp:381-393: make_bvh over the instances' world bounds.  With
  // many instances it is built on the device like a large shape (kind 0: the boxes are the
  // primitives); instances of empty shapes carry the invalid box, which stays with the host builder.
  auto bboxes    = std::vector<ythost::bbox>(sc.num_instances);
  bool any_empty = false;
  for (auto k = 0; k < sc.num_instances; k++) {
    auto& inst = sc.instances[k];
    any_empty |= empty[inst.shape] != 0;
    bboxes[k] = empty[inst.shape] ? ythost::bbox{} : ythost::transform_bbox(inst.frame, roots[inst.shape]);
  }
  out.node_offset.push_back((int64_t)out.nodes.size());
  out.prim_offset.push_back((int64_t)out.prims.size());
  bool tlas_on_device = false;
  if (use_device && !any_empty && sc.num_instances >= ctx->device_build_min_prims) {
    static_assert(sizeof(ythost::bbox) == 6 * sizeof(float), "bbox is {min, max}");
    float* d_boxes = nullptr;
    HIPCHECK(ctx, hipMalloc((void**)&d_boxes, bboxes.size() * sizeof(ythost::bbox)));
    auto e = ctx->xfer.h2d(ctx->stream, d_boxes, bboxes.data(), bboxes.size() * sizeof(ythost::bbox));
    std::string err;
    ctx->d_trees.push_back(ytgpu::DeviceTree{});
    ctx->d_tree_on_host.push_back(0);
    int rc = e == hipSuccess ? ytgpu::build_shape_tree(ctx->stream, 0, nullptr, d_boxes, nullptr, sc.num_instances,
                                   highquality, &ctx->d_trees[nshapes], &err)
                             : ytgpu::BUILD_ERROR;
    (void)hipFree(d_boxes);
    if (rc == ytgpu::BUILD_ERROR) return fail(ctx, YTHIP_ERR_HIP, "device instance-tree build failed: %s", err.c_str());
    if (rc == ytgpu::BUILD_OK) {
      auto& dt = ctx->d_trees[nshapes];
      out.nodes.resize(out.nodes.size() + (size_t)dt.num_nodes);  // filled by ensure_host_bvh()
      out.prims.resize(out.prims.size() + (size_t)dt.num_prims);
      ctx->build_info.device_ms += dt.build_ms;
      ctx->build_info.device_tlas = 1;
      ctx->build_info.max_depth   = std::max(ctx->build_info.max_depth, dt.depth);
      tlas_on_device              = true;
    } else {
      ctx->build_info.fallbacks += 1;
      ctx->d_trees.pop_back();
      ctx->d_tree_on_host.pop_back();
    }
  }
  if (!tlas_on_device) {
    auto tlas = ythost::make_bvh(bboxes, highquality);
    out.nodes.insert(out.nodes.end(), tlas.nodes.begin(), tlas.nodes.end());
    out.prims.insert(out.prims.end(), tlas.prims.begin(), tlas.prims.end());
  }
  out.node_offset.push_back((int64_t)out.nodes.size());
  out.prim_offset.push_back((int64_t)out.prims.size());
  ctx->build_info.build_ms =
      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
  auto t_bake = std::chrono::steady_clock::now();
  int  rc     = bake_bvh(ctx);
  ctx->build_info.bake_ms =
      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_bake).count();
  return rc;
}

int upload_lights_impl(ythip_ctx* ctx) {
  free_all(ctx->light_allocs);
  auto&               L = ctx->h_lights;
  std::vector<DLight> dl(L.lights.size());
  for (size_t k = 0; k < dl.size(); k++)
    dl[k] = {L.lights[k].instance, L.lights[k].environment, (int)L.lights[k].cdf_offset, L.lights[k].cdf_count};
  int rc;
  if ((rc = dupload(ctx, ctx->light_allocs, &ctx->ds.lights, dl.data(), dl.size()))) return rc;
  if ((rc = dupload(ctx, ctx->light_allocs, &ctx->ds.cdf, L.cdf.data(), L.cdf.size()))) return rc;
  ctx->ds.num_lights = (int)dl.size();
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->have_lights = true;
  return YTHIP_OK;
}

template <int S, int LP>
void launch_trace(ythip_ctx* ctx, const KParams& kp, bool count) {
  dim3 grid(ctx->launch_blocks()), block(YT_BLOCK);  // one persistent one-wave workgroup per 16x4 tile
  if (count)  // the counting launch walks binary: its counts are the reference's
    hipLaunchKernelGGL((k_trace<S, LP, true, false>), grid, block, 0, ctx->stream, ctx->ds, ctx->st, kp);
  else if (ctx->use_wide())
    hipLaunchKernelGGL((k_trace<S, LP, false, true>), grid, block, 0, ctx->stream, ctx->ds, ctx->st, kp);
  else
    hipLaunchKernelGGL((k_trace<S, LP, false, false>), grid, block, 0, ctx->stream, ctx->ds, ctx->st, kp);
}

// lp: LP_NONE / LP_DEFER for path & pathtest (area lights absent / present);
// pathdirect & pathmis always trace inline; the rest never need a light pdf.
// fast: ythip_params::fastmath — the tolerance-mode kernels of yt_fast.hip where they exist (wide walk, real samplers).
int launch_trace_any(ythip_ctx* ctx, const KParams& kp, int lp, bool count, bool fast = false) {
  // (the tolerance-mode unit has the wide-walk kernels only: they serve every tree the wide walk's stack bound admits —
  //  use_wide()'s preference for the binary walk on scenes of tiny trees is a matter of speed, not of results)
  if (fast && !count && ctx->wide_stack_ok && ctx->traversal_mode != 0) {
    const int cls = kp.sampler == YTHIP_SAMPLER_PATH && ctx->specialize
                        ? (ctx->all_matte ? 1 : ctx->no_textures ? 2 : ctx->opaque_textured ? 3 : 0)
                        : 0;
    if (ythip_fast_launch(ctx->stream, ctx->launch_blocks(), &ctx->ds, &ctx->st, &kp, lp, cls) == 0) {
      ctx->last_launch_fast = true;
      return YTHIP_OK;
    }
  }
  ctx->last_launch_fast = false;
  switch (kp.sampler) {
    case YTHIP_SAMPLER_PATH:
      if (!count && ctx->all_matte && ctx->specialize && ctx->use_wide()) {
        // the default sampler on an all-matte scene: the variant compiled without the
        // other material lobes and the volume code (same results, fewer registers)
        dim3 grid(ctx->launch_blocks()), block(YT_BLOCK);
        if (lp == LP_DEFER)
          hipLaunchKernelGGL((k_trace<YTHIP_SAMPLER_PATH, LP_DEFER, false, true, 1>), grid, block, 0, ctx->stream,
              ctx->ds, ctx->st, kp);
        else
          hipLaunchKernelGGL((k_trace<YTHIP_SAMPLER_PATH, LP_NONE, false, true, 1>), grid, block, 0, ctx->stream,
              ctx->ds, ctx->st, kp);
      } else if (!count && ctx->no_textures && ctx->specialize && ctx->use_wide()) {
        // no material references a texture (any material types, any primitive kinds): the
        // variant compiled without the texture lookups and the normal-map code
        dim3 grid(ctx->launch_blocks()), block(YT_BLOCK);
        if (lp == LP_DEFER)
          hipLaunchKernelGGL((k_trace<YTHIP_SAMPLER_PATH, LP_DEFER, false, true, 2>), grid, block, 0, ctx->stream, ctx->ds,
              ctx->st, kp);
        else
          hipLaunchKernelGGL((k_trace<YTHIP_SAMPLER_PATH, LP_NONE, false, true, 2>), grid, block, 0, ctx->stream, ctx->ds,
              ctx->st, kp);
      } else if (!count && ctx->opaque_textured && ctx->specialize && ctx->use_wide()) {
        // the "opaque textured" class (matte / glossy / reflective materials, textures in the color and normal slots only,
        // triangle and quad meshes — the scenes of the reference's own corpus): no transmission lobes, no volume code,
        // two texture evaluators instead of five, no line / point intersectors
        dim3 grid(ctx->launch_blocks()), block(YT_BLOCK);
        if (lp == LP_DEFER)
          hipLaunchKernelGGL((k_trace<YTHIP_SAMPLER_PATH, LP_DEFER, false, true, 3>), grid, block, 0, ctx->stream, ctx->ds,
              ctx->st, kp);
        else
          hipLaunchKernelGGL((k_trace<YTHIP_SAMPLER_PATH, LP_NONE, false, true, 3>), grid, block, 0, ctx->stream, ctx->ds,
              ctx->st, kp);
      } else if (lp == LP_DEFER)
        launch_trace<YTHIP_SAMPLER_PATH, LP_DEFER>(ctx, kp, count);
      else
        launch_trace<YTHIP_SAMPLER_PATH, LP_NONE>(ctx, kp, count);
      break;
    case YTHIP_SAMPLER_PATHTEST:
      if (lp == LP_DEFER)
        launch_trace<YTHIP_SAMPLER_PATHTEST, LP_DEFER>(ctx, kp, count);
      else
        launch_trace<YTHIP_SAMPLER_PATHTEST, LP_NONE>(ctx, kp, count);
      break;
#if !defined(YT_DEV_ONLY_PATH) || defined(YT_DEV_NEE)  // development builds: compile the path / pathtest / naive kernels only (10x faster; -DYT_DEV_NEE adds these two)
    case YTHIP_SAMPLER_PATHDIRECT:
      launch_trace<YTHIP_SAMPLER_PATHDIRECT, LP_DEFER>(ctx, kp, count);
      break;
    case YTHIP_SAMPLER_PATHMIS:
      launch_trace<YTHIP_SAMPLER_PATHMIS, LP_DEFER>(ctx, kp, count);
      break;
#endif
    case YTHIP_SAMPLER_NAIVE: launch_trace<YTHIP_SAMPLER_NAIVE, LP_NONE>(ctx, kp, count); break;
#ifndef YT_DEV_ONLY_PATH
    case YTHIP_SAMPLER_EYELIGHT: launch_trace<YTHIP_SAMPLER_EYELIGHT, LP_NONE>(ctx, kp, count); break;
    case YTHIP_SAMPLER_DIAGRAM: launch_trace<YTHIP_SAMPLER_DIAGRAM, LP_NONE>(ctx, kp, count); break;
    case YTHIP_SAMPLER_FURNACE: launch_trace<YTHIP_SAMPLER_FURNACE, LP_NONE>(ctx, kp, count); break;
    case YTHIP_SAMPLER_FALSECOLOR: launch_trace<YTHIP_SAMPLER_FALSECOLOR, LP_NONE>(ctx, kp, count); break;
#endif
    default: return fail(ctx, YTHIP_ERR_SAMPLER, "sampler unknown");
  }
  return YTHIP_OK;
}

// hipEvent bracketing of one launch (profiling mode bit 0)
struct EvScope {
  ythip_ctx* ctx;
  int        idx = -1;
  EvScope(ythip_ctx* c, int kind) : ctx(c) {
    if (!(c->prof_mode & 1)) return;
    if (c->ev_next == c->ev_pool.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
      c->ev_pool.push_back({a, b});
    }
    idx = (int)c->ev_next++;
    c->ev_used.push_back({idx, kind});
    (void)hipEventRecord(c->ev_pool[idx].first, c->stream);
  }
  ~EvScope() {
    if (idx >= 0) (void)hipEventRecord(ctx->ev_pool[idx].second, ctx->stream);
  }
};

void harvest_events(ythip_ctx* ctx) {
  for (auto [idx, kind] : ctx->ev_used) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, ctx->ev_pool[idx].first, ctx->ev_pool[idx].second) != hipSuccess) continue;
    (void)kind;
    ctx->stats.trace_launches++;
    ctx->stats.trace_ms += ms;
  }
  ctx->ev_used.clear();
  ctx->ev_next = 0;
}

// only_pix >= 0: trace_sample() — one sample, numbered `sample`, of that local pixel
int enqueue_samples(ythip_ctx* ctx, const ythip_params* params, const volatile int32_t* stop, int only_pix = -1,
    int sample = 0) {
  if (!ctx->have_scene || !ctx->have_bvh || !ctx->have_lights || !ctx->have_state)
    return fail(ctx, YTHIP_ERR_STATE, "trace_samples needs scene, bvh, lights and state resident");
  if (params->sampler < 0 || params->sampler > YTHIP_SAMPLER_FALSECOLOR)
    return fail(ctx, YTHIP_ERR_SAMPLER, "sampler unknown");
  if (params->camera < 0 || params->camera >= ctx->num_cameras)
    return fail(ctx, YTHIP_ERR_INVALID, "camera index %d out of range [0,%d)", params->camera, ctx->num_cameras);
  if (params->batch < 1) return fail(ctx, YTHIP_ERR_INVALID, "batch must be >= 1");
  ctx->have_denoised = false;
  if (only_pix < 0 && ctx->samples >= params->samples) return YTHIP_OK;  // yocto_trace.cpp:1598
  if (stop && *stop) return fail(ctx, YTHIP_ERR_CANCELLED, "cancelled");

  auto kp          = to_kparams(ctx, params);
  bool count       = (ctx->prof_mode & 2) != 0;
  ctx->st.counters = count ? ctx->d_counters : nullptr;
#if defined(YT_TIMING) || defined(YT_STACK_STATS)
  ctx->st.counters = ctx->d_counters;
#endif
  int  npix        = ctx->st.nslots;  // path-state arrays are per slot
  bool mis         = params->sampler == YTHIP_SAMPLER_PATHMIS;
  if (mis && !ctx->nhit_a) {
    int rc;
    if ((rc = dalloc(ctx, ctx->state_allocs, &ctx->nhit_a, (size_t)npix))) return rc;
    if ((rc = dalloc(ctx, ctx->state_allocs, &ctx->nhit_e, (size_t)npix))) return rc;
  }
  if ((params->sampler == YTHIP_SAMPLER_PATHDIRECT || mis) && !ctx->nee_a) {
    int rc;
    if ((rc = dalloc(ctx, ctx->state_allocs, &ctx->nee_a, (size_t)npix))) return rc;
    if ((rc = dalloc(ctx, ctx->state_allocs, &ctx->nee_b, (size_t)npix))) return rc;
  }
  if (mis && !ctx->nee_c) {
    int rc;
    if ((rc = dalloc(ctx, ctx->state_allocs, &ctx->nee_c, (size_t)npix))) return rc;
    if ((rc = dalloc(ctx, ctx->state_allocs, &ctx->nee_d, (size_t)npix))) return rc;
    if ((rc = dalloc(ctx, ctx->state_allocs, &ctx->nee_e, (size_t)npix))) return rc;
  }
  ctx->st.nee_a = ctx->nee_a, ctx->st.nee_b = ctx->nee_b, ctx->st.nee_c = ctx->nee_c, ctx->st.nee_d = ctx->nee_d;
  ctx->st.nee_e = ctx->nee_e;
  ctx->st.nhit_a      = mis ? ctx->nhit_a : nullptr;
  ctx->st.nhit_e      = mis ? ctx->nhit_e : nullptr;
  ctx->st.stop        = ctx->d_stop;
  ctx->st.stop_host   = ctx->stop_host_dev;
  ctx->st.stop_gen    = ctx->stop_gen.load();  // (begin_batch() numbered this batch)
  ctx->st.sample_base = only_pix < 0 ? ctx->samples : sample;
  ctx->st.batch       = only_pix < 0 ? params->batch : 1;
  ctx->st.only_pix    = only_pix;
  // trace_sample(): one workgroup, on the pixel's tile (logical_block)
  struct GridScope {
    DState& st;
    int     saved;
    GridScope(DState& s, bool one) : st(s), saved(s.nblocks) {
      if (one) st.nblocks = 1;
    }
    ~GridScope() { st.nblocks = saved, st.only_pix = -1; }
  } grid_scope(ctx->st, only_pix >= 0);

  // how sample_lights_pdf's instance walks run
  int lp = LP_NONE;
  if (params->sampler == YTHIP_SAMPLER_PATH || params->sampler == YTHIP_SAMPLER_PATHTEST) {
    for (auto& l : ctx->h_lights.lights)
      if (l.instance != YTHIP_INVALIDID) lp = LP_DEFER;
  }

  // one launch renders the whole batch: every workgroup loops over its tile
  // until its pixels have taken `batch` samples (k_trace)
  ctx->st.tile_perm = nullptr, ctx->st.tile_cost = nullptr;
  ctx->st.pool_next = nullptr, ctx->st.pool_total = 0;
  // pixel pool: only where there are more tiles than resident workgroups, and once the tile costs of a plain launch
  // order the queue (the first batch / the probe launch runs plain and records them)
  // (bounces <= 0: k_trace finishes such a batch in its prologue, tile by tile, without ever reaching the queue)
  const bool pool_ok = ctx->pixel_pool && only_pix < 0 && !count && ctx->st.nblocks > ctx->pool_blocks && params->bounces > 0 &&
                       (ctx->have_tile_costs || !ctx->d_tile_cost || ctx->pixel_pool >= 2);
  bool pool = false;
  int  timed = -1;  // 0: this launch is the timed plain batch, 1: the timed pool batch
  if (pool_ok && ctx->pixel_pool >= 2) {
    pool = true;
  } else if (pool_ok) {
    if (ctx->pool_tune == 2 && hipEventQuery(ctx->pool_ev[1]) == hipSuccess && hipEventQuery(ctx->pool_ev[3]) == hipSuccess) {
      if (hipEventElapsedTime(&ctx->pool_ms[0], ctx->pool_ev[0], ctx->pool_ev[1]) == hipSuccess &&
          hipEventElapsedTime(&ctx->pool_ms[1], ctx->pool_ev[2], ctx->pool_ev[3]) == hipSuccess)
        ctx->pool_on = ctx->pool_ms[1] / ctx->pool_samples[1] < 0.97 * ctx->pool_ms[0] / ctx->pool_samples[0];
      ctx->pool_tune = 3;
    }
    if (ctx->pool_tune == 3) pool = ctx->pool_on;
    else if (params->batch >= 8 && ctx->pool_tune < 2) {  // (a batch long enough for its time to mean something)
      for (auto& e : ctx->pool_ev)
        if (!e) HIPCHECK(ctx, hipEventCreate(&e));
      timed = ctx->pool_tune, pool = timed == 1;
    }
  }
  if (pool) {
    if (!ctx->d_pool_next) HIPCHECK(ctx, hipMalloc((void**)&ctx->d_pool_next, 64));
    ctx->st.pool_next = ctx->d_pool_next, ctx->st.pool_total = ctx->st.nblocks * YT_BLOCK;
    // the queue's head starts behind the statically assigned first tiles (workgroup b owns entries [64 b, 64 b + 64): k_trace)
    HIPCHECK(ctx, hipMemsetD32Async((hipDeviceptr_t)ctx->d_pool_next, ctx->launch_blocks() * YT_BLOCK, 16, ctx->stream));
  }
  const bool lpt = ctx->d_tile_cost && only_pix < 0 && !count;
  if (lpt) {
    if (ctx->have_tile_costs) {  // the previous batch's costs order this one (same pixels, same work)
      // costs are stable from batch to batch: the order is refreshed every 16th launch only
      if (ctx->lpt_age % 16 == 0)
        HIPCHECK(ctx, ytorder::order_by_cost(ctx->stream, ctx->d_tile_cost, ctx->st.nblocks, ctx->d_tile_perm, ctx->d_sort_temp,
                          ctx->sort_temp_bytes));
      ctx->lpt_age++;
      ctx->st.tile_perm = ctx->d_tile_perm;
    }
    ctx->st.tile_cost = pool ? nullptr : ctx->d_tile_cost;  // (a pool launch has no per-tile time: the last plain launch's costs stay)
  }
  {
    EvScope ev(ctx, 0);
    if (timed >= 0) HIPCHECK(ctx, hipEventRecord(ctx->pool_ev[2 * timed], ctx->stream));
    int rc = launch_trace_any(ctx, kp, lp, count, params->fastmath != 0);
    if (rc) return rc;
    if (timed >= 0) {
      HIPCHECK(ctx, hipEventRecord(ctx->pool_ev[2 * timed + 1], ctx->stream));
      ctx->pool_samples[timed] = params->batch, ctx->pool_tune = timed + 1;
    }
  }
  HIPCHECK(ctx, hipGetLastError());
  if (lpt && !pool) ctx->have_tile_costs = true;
  if (only_pix < 0) ctx->samples += params->batch;  // yocto_trace.cpp:1614
  return YTHIP_OK;
}

}  // namespace

namespace {
// what the kernel specialisation and the bounce-loop bound depend on (from the resident
// shapes + the given materials)
void classify_scene(ythip_ctx* ctx, const ythip_material* materials, int num_materials) {
  ctx->all_matte = num_materials > 0;
  for (int k = 0; k < num_materials; k++) {
    const auto& m = materials[k];
    if (m.type != YTHIP_MATTE || (m.emission_tex & m.color_tex & m.roughness_tex & m.scattering_tex & m.normal_tex) != YTHIP_INVALIDID)
      ctx->all_matte = false;
  }
  for (auto& sh : ctx->h_shapes)  // ... and every shape a triangle mesh
    if (sh.num_points || sh.num_lines || sh.num_quads) ctx->all_matte = false;
  ctx->no_textures = true;
  for (int k = 0; k < num_materials; k++) {
    const auto& m = materials[k];
    if ((m.emission_tex & m.color_tex & m.roughness_tex & m.scattering_tex & m.normal_tex) != YTHIP_INVALIDID) ctx->no_textures = false;
  }
  ctx->opaque_textured = num_materials > 0;
  for (int k = 0; k < num_materials; k++) {
    const auto& m = materials[k];
    if ((m.type != YTHIP_MATTE && m.type != YTHIP_GLOSSY && m.type != YTHIP_REFLECTIVE) ||
        (m.emission_tex & m.roughness_tex & m.scattering_tex) != YTHIP_INVALIDID)
      ctx->opaque_textured = false;
  }
  for (auto& sh : ctx->h_shapes)
    if (sh.num_points || sh.num_lines) ctx->opaque_textured = false;
  ctx->may_retry = false;
  for (int k = 0; k < num_materials; k++)
    if (materials[k].opacity < 1 || materials[k].color_tex != YTHIP_INVALIDID) ctx->may_retry = true;
  for (auto& sh : ctx->h_shapes)
    if (sh.num_colors) ctx->may_retry = true;
}
}  // namespace

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

// Cancellation plumbing (measured: tools/cancel_latency.py, profiles/r03_cancel_latency.txt).
//   * Every workgroup polls a DEVICE word once per iteration (one scalar load that hits the L2).
//   * The host never touches that word.  It stores the batch number into a word of PINNED HOST memory
//     — a plain store: no stream, no command processor, no copy engine, nothing that could queue
//     behind the batch — and the kernel relays it: every 64th iteration (phase-shifted by tile) a
//     workgroup also reads the host word over the fabric, and the first one that sees the batch's
//     number writes it into the device word, which every other workgroup sees at its next iteration.
//     With thousands of resident workgroups somebody looks within microseconds.
//   Round 2 raised the device word with hipStreamWriteValue32 on a side stream: 1-40 ms on an idle
//   queue, but 100-200 ms as soon as a batch was several launches (the longest-tile-first probe:
//   1 + order kernels + (batch - 1)) — the write waited for queue arbitration.  Polling the host word
//   directly from every workgroup every iteration is no alternative: configs[1] 5.3 -> 54 ms
//   (100 M small reads per second over PCIe).
static hipError_t alloc_stop_word(ythip_ctx* ctx) {
  void* h = nullptr;
  hipError_t e = hipHostMalloc(&h, 64, hipHostMallocCoherent | hipHostMallocMapped);
  if (e != hipSuccess) return e;
  std::memset(h, 0, 64);
  void* d = nullptr;
  if ((e = hipHostGetDevicePointer(&d, h, 0)) != hipSuccess) {
    (void)hipHostFree(h);
    return e;
  }
  ctx->stop_host = (int*)h, ctx->stop_host_dev = (const int*)d;
  return hipMalloc((void**)&ctx->d_stop, 64);
}

int ythip_create(int device, ythip_ctx** out) {
  if (!out) return fail(nullptr, YTHIP_ERR_INVALID, "out is null");
  *out       = nullptr;
  int ndev   = 0;
  auto e     = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(nullptr, YTHIP_ERR_HIP, "no HIP device available (%s): libythip has no CPU fallback",
        hipGetErrorString(e));
  if (device < 0 || device >= ndev) return fail(nullptr, YTHIP_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
  auto ctx    = new ythip_ctx{};
  ctx->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&ctx->own_stream) != hipSuccess) {
    delete ctx;
    return fail(nullptr, YTHIP_ERR_HIP, "hipSetDevice/hipStreamCreate failed");
  }
  ctx->stream = ctx->own_stream;
  if (const char* e = std::getenv("YTHIP_HOLD")) ctx->hold_policy = std::atoi(e);
  if (const char* e = std::getenv("YTHIP_PEEK")) ctx->peek_policy = std::atoi(e);
  if (const char* e = std::getenv("YTHIP_LPT")) ctx->lpt = std::atoi(e);
  if (const char* e = std::getenv("YTHIP_LPT_PROBE")) ctx->lpt_probe = std::atoi(e) != 0;
  if (const char* e = std::getenv("YTHIP_PIXEL_POOL")) ctx->pixel_pool = std::atoi(e);
  {
    hipDeviceProp_t prop;
    ctx->pool_blocks = hipGetDeviceProperties(&prop, device) == hipSuccess ? prop.multiProcessorCount * 16 : 4096;
    if (const char* e = std::getenv("YTHIP_POOL_BLOCKS")) ctx->pool_blocks = std::max(1, std::atoi(e));
  }
  if (const char* e = std::getenv("YTHIP_DENOISE_SIMPLE")) ctx->denoise_simple = std::atoi(e) != 0;
  if (hipMalloc((void**)&ctx->d_counters, CNT_BANKS * CNT_STRIDE * sizeof(unsigned long long)) != hipSuccess ||
      hipMemset(ctx->d_counters, 0, CNT_BANKS * CNT_STRIDE * sizeof(unsigned long long)) != hipSuccess ||
      alloc_stop_word(ctx) != hipSuccess || hipMemset(ctx->d_stop, 0, 64) != hipSuccess) {
    delete ctx;
    return fail(nullptr, YTHIP_ERR_HIP, "context allocation failed");
  }
  *out = ctx;
  return YTHIP_OK;
}

void ythip_destroy(ythip_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  free_all(ctx->scene_allocs);
  free_all(ctx->bvh_allocs);
  free_all(ctx->light_allocs);
  free_all(ctx->state_allocs);
  free_device_trees(ctx);
  for (auto& ev : ctx->ev_pool) {
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  if (ctx->d_counters) (void)hipFree(ctx->d_counters);
  if (ctx->stop_host) (void)hipHostFree(ctx->stop_host);
  if (ctx->d_stop) (void)hipFree(ctx->d_stop);
  if (ctx->d_pool_next) (void)hipFree(ctx->d_pool_next);
  for (auto e : ctx->pool_ev)
    if (e) (void)hipEventDestroy(e);
  free_staging(ctx);
  free_all(ctx->denoise_allocs);
  free_all(ctx->order_allocs);
  if (ctx->done_event) (void)hipEventDestroy(ctx->done_event);
  ctx->xfer.destroy();
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

const char* ythip_last_error(const ythip_ctx* ctx) { return ctx ? ctx->err.c_str() : g_error.c_str(); }

int ythip_set_stream(ythip_ctx* ctx, void* hip_stream) {
  if (!ctx) return YTHIP_ERR_INVALID;
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return YTHIP_OK;
}

int ythip_sync(ythip_ctx* ctx) {
  if (!ctx) return YTHIP_ERR_INVALID;
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
  harvest_events(ctx);
  return YTHIP_OK;
}

namespace {
int upload_scene_impl(ythip_ctx* ctx, const ythip_scene* sc, bool from_staging);
}  // namespace

int ythip_upload_scene(ythip_ctx* ctx, const ythip_scene* sc) {
  if (!ctx || !sc) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  int rc = upload_scene_impl(ctx, sc, false);
  if (rc == YTHIP_OK) free_staging(ctx);  // (a staged scene that was never uploaded is dropped)
  return rc;
}

// Pinned host pools sized by `counts` (its num_* fields; pointers ignored): the loader writes
// cameras, instances, ..., the concatenated vertex / element / texel pools and the per-shape
// descriptors straight into them — no intermediate copy —, then ythip_upload_scene_staged()
// sends them (asynchronous DMA from pinned memory) and keeps them as the host copies the BVH
// and light builders read.
int ythip_scene_staging(ythip_ctx* ctx, const ythip_scene* counts, ythip_scene* staged) {
  if (!ctx || !counts || !staged) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  // the pools of the previous staging are handed out again where they are large enough (a reload, the
  // next frame of an animation: pinning 2 GB costs 0.2 s) — once nothing reads them any more
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
  drop_staging_views(ctx);
  ctx->staging_allocs.resize(17, nullptr);
  ctx->staging_caps.resize(17, 0);
  ythip_scene v = *counts;
  int  slot = 0;
  auto pin = [&](auto*& field, size_t count) -> int {
    using T = std::remove_const_t<std::remove_pointer_t<std::remove_reference_t<decltype(field)>>>;
    const size_t need = std::max<size_t>(count, 1) * sizeof(T);
    const int    k    = slot++;
    if (ctx->staging_caps[k] < need) {
      if (ctx->staging_allocs[k]) (void)hipHostFree(ctx->staging_allocs[k]);
      ctx->staging_allocs[k] = nullptr, ctx->staging_caps[k] = 0;
      void* p = nullptr;
      if (hipHostMalloc(&p, need, hipHostMallocDefault) != hipSuccess)
        return fail(ctx, YTHIP_ERR_HIP, "pinned staging allocation of %zu bytes failed", need);
      ctx->staging_allocs[k] = p, ctx->staging_caps[k] = need;
    }
    field = (T*)ctx->staging_allocs[k];
    return YTHIP_OK;
  };
  int rc;
#define PIN(field, count) \
  if ((rc = pin(v.field, (size_t)(count)))) { free_staging(ctx); return rc; }
  PIN(cameras, v.num_cameras);
  PIN(instances, v.num_instances);
  PIN(environments, v.num_environments);
  PIN(shapes, v.num_shapes);
  PIN(textures, v.num_textures);
  PIN(materials, v.num_materials);
  PIN(points, v.num_points);
  PIN(lines, v.num_lines * 2);
  PIN(triangles, v.num_triangles * 3);
  PIN(quads, v.num_quads * 4);
  PIN(positions, v.num_positions * 3);
  PIN(normals, v.num_normals * 3);
  PIN(texcoords, v.num_texcoords * 2);
  PIN(colors, v.num_colors * 4);
  PIN(radius, v.num_radius);
  PIN(pixelsf, v.num_pixelsf * 4);
  PIN(pixelsb, v.num_pixelsb * 4);
#undef PIN
  ctx->staged      = v;
  ctx->have_staged = true;
  *staged          = v;
  return YTHIP_OK;
}

int ythip_upload_scene_staged(ythip_ctx* ctx) {
  if (!ctx) return YTHIP_ERR_INVALID;
  if (!ctx->have_staged) return fail(ctx, YTHIP_ERR_STATE, "ythip_scene_staging first");
  return upload_scene_impl(ctx, &ctx->staged, true);
}

namespace {
int upload_scene_impl(ythip_ctx* ctx, const ythip_scene* sc, bool from_staging) {
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  // validation (the reference would index out of bounds)
  for (int k = 0; k < sc->num_instances; k++) {
    auto& i = sc->instances[k];
    if (i.shape < 0 || i.shape >= sc->num_shapes || i.material < 0 || i.material >= sc->num_materials)
      return fail(ctx, YTHIP_ERR_INVALID, "instance %d references shape %d / material %d out of range", k, i.shape, i.material);
  }
  if (sc->num_positions > 0x7fffffffll / 4 || sc->num_triangles > 0x7fffffffll / 4 ||
      sc->num_quads > 0x7fffffffll / 4 || sc->num_lines > 0x7fffffffll / 4)
    return fail(ctx, YTHIP_ERR_INVALID, "scene pools exceed 32-bit device indexing");
  free_all(ctx->scene_allocs);
  ctx->have_scene = ctx->have_bvh = ctx->have_lights = false;
  auto& ds = ctx->ds;
  ds       = DScene{};
  int rc;
#define UP(field, src, count) \
  if ((rc = dupload(ctx, ctx->scene_allocs, &ds.field, src, (size_t)(count)))) return rc;
  UP(cameras, sc->cameras, sc->num_cameras);
  UP(instances, sc->instances, sc->num_instances);
  UP(environments, sc->environments, sc->num_environments);
  UP(materials, sc->materials, sc->num_materials);
  UP(textures, sc->textures, sc->num_textures);
  UP(points, sc->points, sc->num_points);
  UP(lines, sc->lines, sc->num_lines * 2);
  UP(triangles, sc->triangles, sc->num_triangles * 3);
  UP(quads, sc->quads, sc->num_quads * 4);
  UP(positions, sc->positions, sc->num_positions * 3);
  UP(normals, sc->normals, sc->num_normals * 3);
  UP(texcoords, sc->texcoords, sc->num_texcoords * 2);
  UP(colors, sc->colors, sc->num_colors * 4);
  UP(radius, sc->radius, sc->num_radius);
  UP(pixelsf, sc->pixelsf, sc->num_pixelsf * 4);
  UP(pixelsb, sc->pixelsb, sc->num_pixelsb * 4);
  std::vector<DShape> shapes(sc->num_shapes);
  for (int k = 0; k < sc->num_shapes; k++) {
    auto& s  = sc->shapes[k];
    auto& d  = shapes[k];
    d        = DShape{};
    d.kind_bvh  = ythost::kind_bvh(s);
    d.kind_eval = ythost::kind_eval(s);
    auto off    = [&](int kind) -> int {
      switch (kind) {
        case KIND_POINTS: return (int)s.points_offset;
        case KIND_LINES: return (int)s.lines_offset;
        case KIND_TRIANGLES: return (int)s.triangles_offset;
        case KIND_QUADS: return (int)s.quads_offset;
        default: return 0;
      }
    };
    d.elem_bvh  = off(d.kind_bvh);
    d.elem_eval = off(d.kind_eval);
    d.positions = (int)s.positions_offset;
    d.normals   = s.num_normals ? (int)s.normals_offset : -1;
    d.texcoords = s.num_texcoords ? (int)s.texcoords_offset : -1;
    d.colors    = s.num_colors ? (int)s.colors_offset : -1;
    d.radius    = s.num_radius ? (int)s.radius_offset : -1;
  }
  UP(shapes, shapes.data(), shapes.size());
  std::vector<float> env_inv((size_t)sc->num_environments * 12);
  for (int k = 0; k < sc->num_environments; k++)
    ythost::inverse_frame_rigid(sc->environments[k].frame, env_inv.data() + 12 * k);
  UP(env_inv, env_inv.data(), env_inv.size());
#undef UP
  ctx->num_cameras    = sc->num_cameras;
  ds.num_instances    = sc->num_instances;
  ds.num_environments = sc->num_environments;
  ds.num_shapes       = sc->num_shapes;
  ds.num_materials    = sc->num_materials;
  ds.num_textures     = sc->num_textures;
  // host copies for BVH baking
  ctx->h_shapes.assign(sc->shapes, sc->shapes + sc->num_shapes);
  ctx->h_instances.assign(sc->instances, sc->instances + sc->num_instances);
  if (from_staging) {  // the pinned pools ARE the host copies
    ctx->h_points.adopt((int32_t*)sc->points, (size_t)sc->num_points);
    ctx->h_lines.adopt((int32_t*)sc->lines, (size_t)sc->num_lines * 2);
    ctx->h_triangles.adopt((int32_t*)sc->triangles, (size_t)sc->num_triangles * 3);
    ctx->h_quads.adopt((int32_t*)sc->quads, (size_t)sc->num_quads * 4);
    ctx->h_positions.adopt((float*)sc->positions, (size_t)sc->num_positions * 3);
    ctx->h_radius.adopt((float*)sc->radius, (size_t)sc->num_radius);
  } else {
    ctx->h_points.assign(sc->points, sc->points + sc->num_points);
    ctx->h_lines.assign(sc->lines, sc->lines + sc->num_lines * 2);
    ctx->h_triangles.assign(sc->triangles, sc->triangles + sc->num_triangles * 3);
    ctx->h_quads.assign(sc->quads, sc->quads + sc->num_quads * 4);
    ctx->h_positions.assign(sc->positions, sc->positions + sc->num_positions * 3);
    ctx->h_radius.assign(sc->radius, sc->radius + sc->num_radius);
  }
  classify_scene(ctx, sc->materials, sc->num_materials);
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->have_scene = true;
  return YTHIP_OK;
}
}  // namespace

// In-place edits of the small pools (what the reference's GUI does between batches: it
// reads the scene fresh on every trace_samples call).  Counts must match the resident scene.
int ythip_update_materials(ythip_ctx* ctx, const ythip_material* materials, int num) {
  if (!ctx || (!materials && num > 0)) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  if (!ctx->have_scene) return fail(ctx, YTHIP_ERR_STATE, "upload_scene first");
  if (num != ctx->ds.num_materials)
    return fail(ctx, YTHIP_ERR_INVALID, "scene has %d materials resident, %d given", ctx->ds.num_materials, num);
  for (int k = 0; k < num; k++) {
    const auto& m   = materials[k];
    const int   t[] = {m.emission_tex, m.color_tex, m.roughness_tex, m.scattering_tex, m.normal_tex};
    for (int x : t)
      if (x != YTHIP_INVALIDID && (x < 0 || x >= ctx->ds.num_textures))
        return fail(ctx, YTHIP_ERR_INVALID, "material %d references texture %d out of range", k, x);
  }
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  HIPCHECK(ctx, ctx->xfer.h2d(ctx->stream, (void*)ctx->ds.materials, materials, (size_t)num * sizeof(ythip_material)));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
  classify_scene(ctx, materials, num);  // the kernel specialisation follows the materials
  return YTHIP_OK;
}

int ythip_update_environments(ythip_ctx* ctx, const ythip_environment* environments, int num) {
  if (!ctx || (!environments && num > 0)) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  if (!ctx->have_scene) return fail(ctx, YTHIP_ERR_STATE, "upload_scene first");
  if (num != ctx->ds.num_environments)
    return fail(ctx, YTHIP_ERR_INVALID, "scene has %d environments resident, %d given", ctx->ds.num_environments, num);
  for (int k = 0; k < num; k++)
    if (environments[k].emission_tex != YTHIP_INVALIDID &&
        (environments[k].emission_tex < 0 || environments[k].emission_tex >= ctx->ds.num_textures))
      return fail(ctx, YTHIP_ERR_INVALID, "environment %d references texture %d out of range", k, environments[k].emission_tex);
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  std::vector<float> env_inv((size_t)num * 12);
  for (int k = 0; k < num; k++) ythost::inverse_frame_rigid(environments[k].frame, env_inv.data() + 12 * k);
  HIPCHECK(ctx, ctx->xfer.h2d(ctx->stream, (void*)ctx->ds.environments, environments, (size_t)num * sizeof(ythip_environment)));
  HIPCHECK(ctx, ctx->xfer.h2d(ctx->stream, (void*)ctx->ds.env_inv, env_inv.data(), env_inv.size() * sizeof(float)));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
  return YTHIP_OK;
}

int ythip_update_cameras(ythip_ctx* ctx, const ythip_camera* cameras, int num) {
  if (!ctx || !cameras) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  if (!ctx->have_scene) return fail(ctx, YTHIP_ERR_STATE, "upload_scene first");
  if (num != ctx->num_cameras)
    return fail(ctx, YTHIP_ERR_INVALID, "scene has %d cameras resident, %d given", ctx->num_cameras, num);
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  HIPCHECK(ctx, ctx->xfer.h2d(ctx->stream, (void*)ctx->ds.cameras, cameras, (size_t)num * sizeof(ythip_camera)));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
  return YTHIP_OK;
}

int ythip_build_bvh(ythip_ctx* ctx, const ythip_scene* sc, int highquality) {
  if (!ctx || !sc) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  if (!ctx->have_scene) return fail(ctx, YTHIP_ERR_STATE, "upload_scene first");
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  return build_bvh_mixed(ctx, *sc, highquality != 0, ctx->bvh_builder != 0);
}

// ---- update_scene_bvh (yocto_bvh.cpp:434-451) -------------------------------------------
int ythip_update_shape_vertices(ythip_ctx* ctx, int32_t shape, const float* positions, int64_t num_positions,
    const float* normals, int64_t num_normals, const float* radius, int64_t num_radius) {
  if (!ctx) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  if (!ctx->have_scene) return fail(ctx, YTHIP_ERR_STATE, "upload_scene first");
  if (shape < 0 || shape >= (int)ctx->h_shapes.size())
    return fail(ctx, YTHIP_ERR_INVALID, "shape %d out of range [0,%d)", shape, (int)ctx->h_shapes.size());
  const auto& sh = ctx->h_shapes[shape];
  if (positions && num_positions != sh.num_positions)
    return fail(ctx, YTHIP_ERR_INVALID, "shape %d has %lld positions resident, %lld given", shape,
        (long long)sh.num_positions, (long long)num_positions);
  if (normals && num_normals != sh.num_normals)
    return fail(ctx, YTHIP_ERR_INVALID, "shape %d has %lld normals resident, %lld given", shape,
        (long long)sh.num_normals, (long long)num_normals);
  if (radius && num_radius != sh.num_radius)
    return fail(ctx, YTHIP_ERR_INVALID, "shape %d has %lld radii resident, %lld given", shape, (long long)sh.num_radius,
        (long long)num_radius);
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  if (normals && num_normals > 0)  // shading data only: no host copy is kept
    HIPCHECK(ctx, ctx->xfer.h2d(ctx->stream, (void*)(ctx->ds.normals + 3 * sh.normals_offset), normals, (size_t)num_normals * 12));
  if (positions && num_positions > 0) {
    std::memcpy(ctx->h_positions.data() + 3 * sh.positions_offset, positions, (size_t)num_positions * 12);
    HIPCHECK(ctx, ctx->xfer.h2d(ctx->stream, (void*)(ctx->ds.positions + 3 * sh.positions_offset), positions, (size_t)num_positions * 12));
  }
  if (radius && num_radius > 0) {
    std::memcpy(ctx->h_radius.data() + sh.radius_offset, radius, (size_t)num_radius * 4);
    HIPCHECK(ctx, ctx->xfer.h2d(ctx->stream, (void*)(ctx->ds.radius + sh.radius_offset), radius, (size_t)num_radius * 4));
  }
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));  // the caller's buffers are free again
  return YTHIP_OK;
}

namespace {
// many edited frames travel as two compact arrays and are put in place by the device (one copy per 48-byte
// frame was one bounce slot, one event and one DMA descriptor each)
__global__ void __launch_bounds__(64) k_scatter_frames(ythip_instance* instances, const int32_t* ids, const ythip_frame* frames, int n) {
  const int k = (int)(blockIdx.x * 64 + threadIdx.x);
  if (k < n) instances[ids[k]].frame = frames[k];
}
}  // namespace

int ythip_update_instance_frames(ythip_ctx* ctx, const int32_t* instances, int32_t num, const ythip_frame* frames) {
  if (!ctx || (num > 0 && (!instances || !frames))) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  if (!ctx->have_scene) return fail(ctx, YTHIP_ERR_STATE, "upload_scene first");
  for (int k = 0; k < num; k++)
    if (instances[k] < 0 || instances[k] >= (int)ctx->h_instances.size())
      return fail(ctx, YTHIP_ERR_INVALID, "instance %d out of range [0,%d)", instances[k], (int)ctx->h_instances.size());
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  bool scatter = num >= 64;
  if (scatter) {  // (an instance named twice keeps its LAST frame, as a loop of assignments would: those go one by one)
    std::vector<char> seen(ctx->h_instances.size(), 0);
    for (int k = 0; k < num && scatter; k++) scatter = !seen[instances[k]], seen[instances[k]] = 1;
  }
  for (int k = 0; k < num; k++) ctx->h_instances[instances[k]].frame = frames[k];
  if (scatter) {
    std::vector<void*> tmp;
    const int32_t*     d_ids    = nullptr;
    const ythip_frame* d_frames = nullptr;
    int                rc;
    if ((rc = dupload(ctx, tmp, &d_ids, instances, (size_t)num)) || (rc = dupload(ctx, tmp, &d_frames, frames, (size_t)num))) {
      free_all(tmp);
      return rc;
    }
    hipLaunchKernelGGL(k_scatter_frames, dim3((num + 63) / 64), dim3(64), 0, ctx->stream, (ythip_instance*)ctx->ds.instances,
        d_ids, d_frames, (int)num);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    free_all(tmp);
    HIPCHECK(ctx, e);
    return YTHIP_OK;
  }
  for (int k = 0; k < num; k++)
    HIPCHECK(ctx, ctx->xfer.h2d(ctx->stream, (void*)&ctx->ds.instances[instances[k]].frame, &frames[k], sizeof(ythip_frame)));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
  return YTHIP_OK;
}

int ythip_update_bvh(ythip_ctx* ctx, const int32_t* updated_instances, int32_t num_instances,
    const int32_t* updated_shapes, int32_t num_shapes) {
  (void)updated_instances, (void)num_instances;  // every instance box is recomputed — yocto_bvh.cpp:441-447
  if (!ctx || (num_shapes > 0 && !updated_shapes)) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  if (!ctx->have_scene || !ctx->have_bvh) return fail(ctx, YTHIP_ERR_STATE, "update_bvh needs scene and bvh resident");
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  auto  t_start = std::chrono::steady_clock::now();
  auto& b       = ctx->h_bvh;
  int   nshapes = (int)ctx->h_shapes.size();
  for (int k = 0; k < num_shapes; k++)
    if (updated_shapes[k] < 0 || updated_shapes[k] >= nshapes)
      return fail(ctx, YTHIP_ERR_INVALID, "shape %d out of range [0,%d)", updated_shapes[k], nshapes);
  ctx->build_info = {};
  auto on_device  = [&](int t) { return t < (int)ctx->d_trees.size() && ctx->d_trees[t].nodes != nullptr; };
  // host-side view of the element pools (what make_shape_bvh read at build time)
  ythip_scene view = {};
  view.positions = ctx->h_positions.data(), view.radius = ctx->h_radius.empty() ? nullptr : ctx->h_radius.data();
  view.points = ctx->h_points.data(), view.lines = ctx->h_lines.data();
  view.triangles = ctx->h_triangles.data(), view.quads = ctx->h_quads.data();
  // update_shape_bvh — yocto_bvh.cpp:398-431
  for (int k = 0; k < num_shapes; k++) {
    int         s  = updated_shapes[k];
    const auto& sh = ctx->h_shapes[s];
    if (on_device(s)) {
      int        kind = ythost::kind_bvh(sh);
      const int* el   = kind == KIND_TRIANGLES ? ctx->ds.triangles + 3 * sh.triangles_offset
                        : kind == KIND_QUADS   ? ctx->ds.quads + 4 * sh.quads_offset
                        : kind == KIND_LINES   ? ctx->ds.lines + 2 * sh.lines_offset
                                               : ctx->ds.points + sh.points_offset;
      std::string err;
      if (ytgpu::refit_shape_tree(ctx->stream, ctx->d_trees[s], kind, el, ctx->ds.positions + 3 * sh.positions_offset,
              sh.radius_offset >= 0 && sh.num_radius ? ctx->ds.radius + sh.radius_offset : nullptr, &err) !=
          ytgpu::BUILD_OK)
        return fail(ctx, YTHIP_ERR_HIP, "device bvh refit failed: %s", err.c_str());
      ctx->d_tree_on_host[s] = 0;  // the host copy of this tree is stale now
      ctx->build_info.device_trees += 1;
      ctx->build_info.device_prims += ctx->d_trees[s].num_prims;
      ctx->build_info.device_ms += ctx->d_trees[s].build_ms;
    } else {
      auto bboxes = ythost::shape_prim_bboxes(view, sh);
      ythost::refit_bvh(b.nodes.data() + b.node_offset[s], b.node_offset[s + 1] - b.node_offset[s],
          b.prims.data() + b.prim_offset[s], bboxes);
      ctx->build_info.host_trees += 1;
    }
  }
  // the instance tree — yocto_bvh.cpp:441-450
  std::vector<ythost::bbox> roots(nshapes);
  std::vector<char>         empty(nshapes, 1);
  for (int s = 0; s < nshapes; s++) {
    ythip_bvh_node root;
    if (on_device(s) && !ctx->d_tree_on_host[s]) {
      HIPCHECK(ctx, ctx->xfer.d2h(ctx->stream, &root, ctx->d_trees[s].nodes, sizeof(root)));
    } else {
      if (b.node_offset[s + 1] == b.node_offset[s]) continue;
      root = b.nodes[b.node_offset[s]];
    }
    empty[s]     = 0;
    roots[s].min = {root.bbox_min[0], root.bbox_min[1], root.bbox_min[2]};
    roots[s].max = {root.bbox_max[0], root.bbox_max[1], root.bbox_max[2]};
  }
  std::vector<ythost::bbox> bboxes(ctx->h_instances.size());
  for (size_t k = 0; k < bboxes.size(); k++) {
    const auto& inst = ctx->h_instances[k];
    bboxes[k]        = empty[inst.shape] ? ythost::bbox{} : ythost::transform_bbox(inst.frame, roots[inst.shape]);
  }
  if (on_device(nshapes)) {
    // an instance tree that was built on the device is refitted there too (kind 0: the instances' boxes are the
    // primitives) and stays there.  (it used to be brought home through ensure_host_bvh — which downloads EVERY
    // device-resident shape tree — refitted on the host and dropped from the device for good.)
    static_assert(sizeof(ythost::bbox) == 6 * sizeof(float), "bbox is {min, max}");
    float* d_boxes = nullptr;
    HIPCHECK(ctx, hipMalloc((void**)&d_boxes, bboxes.size() * sizeof(ythost::bbox)));
    hipError_t  e = ctx->xfer.h2d(ctx->stream, d_boxes, bboxes.data(), bboxes.size() * sizeof(ythost::bbox));
    std::string err;
    int brc = e == hipSuccess ? ytgpu::refit_shape_tree(ctx->stream, ctx->d_trees[nshapes], 0, nullptr, d_boxes, nullptr, &err)
                              : ytgpu::BUILD_ERROR;
    (void)hipFree(d_boxes);
    if (brc != ytgpu::BUILD_OK)
      return fail(ctx, YTHIP_ERR_HIP, "device instance-tree refit failed: %s", e != hipSuccess ? hipGetErrorString(e) : err.c_str());
    ctx->d_tree_on_host[nshapes] = 0;  // the host copy of the instance tree is stale now
    ctx->build_info.device_tlas  = 1;
    ctx->build_info.device_ms += ctx->d_trees[nshapes].build_ms;
  } else {
    ythost::refit_bvh(b.nodes.data() + b.node_offset[nshapes], b.node_offset[nshapes + 1] - b.node_offset[nshapes],
        b.prims.data() + b.prim_offset[nshapes], bboxes);
  }
  ctx->build_info.build_ms =
      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
  auto t_bake = std::chrono::steady_clock::now();
  int  rc     = bake_bvh(ctx);
  ctx->build_info.bake_ms =
      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_bake).count();
  return rc;
}

int ythip_set_bvh_builder(ythip_ctx* ctx, int mode, int64_t min_prims) {
  if (!ctx || mode < 0 || mode > 1) return fail(ctx, YTHIP_ERR_INVALID, "bvh builder mode must be 0 (host) or 1 (device)");
  ctx->bvh_builder = mode;
  if (min_prims > 0) ctx->device_build_min_prims = min_prims;
  return YTHIP_OK;
}

int ythip_bvh_build_info(ythip_ctx* ctx, ythip_build_info* info) {
  if (!ctx || !info) return YTHIP_ERR_INVALID;
  *info = ctx->build_info;
  return YTHIP_OK;
}

int ythip_bvh_baked_sizes(ythip_ctx* ctx, int64_t* num_pairs, int64_t* num_leaf4) {
  if (!ctx || !ctx->have_bvh) return fail(ctx, YTHIP_ERR_STATE, "no bvh resident");
  *num_pairs = ctx->num_pairs, *num_leaf4 = ctx->num_leaf4;
  return YTHIP_OK;
}

int ythip_bvh_baked_download(ythip_ctx* ctx, float* pairs, float* leafdata, float* quads) {
  if (!ctx || !ctx->have_bvh) return fail(ctx, YTHIP_ERR_STATE, "no bvh resident");
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  if (quads && ctx->num_pairs)
    HIPCHECK(ctx, ctx->xfer.d2h(ctx->stream, quads, ctx->ds.wide, (size_t)ctx->num_pairs * 128));
  if (pairs && ctx->num_pairs)
    HIPCHECK(ctx, ctx->xfer.d2h(ctx->stream, pairs, ctx->ds.pairs, (size_t)ctx->num_pairs * 64));
  if (leafdata && ctx->num_leaf4)
    HIPCHECK(ctx, ctx->xfer.d2h(ctx->stream, leafdata, ctx->ds.leafdata, (size_t)ctx->num_leaf4 * 16));
  return YTHIP_OK;
}

int ythip_upload_bvh(ythip_ctx* ctx, const ythip_bvh* bvh) {
  if (!ctx || !bvh) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  if (!ctx->have_scene) return fail(ctx, YTHIP_ERR_STATE, "upload_scene first");
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  free_device_trees(ctx);
  ctx->build_info = {};
  auto& b = ctx->h_bvh;
  int   n = bvh->num_trees;
  b.node_offset.assign(bvh->node_offset, bvh->node_offset + n + 1);
  b.prim_offset.assign(bvh->prim_offset, bvh->prim_offset + n + 1);
  b.nodes.assign(bvh->nodes, bvh->nodes + b.node_offset[n]);
  b.prims.assign(bvh->primitives, bvh->primitives + b.prim_offset[n]);
  return bake_bvh(ctx);
}

int ythip_bvh_sizes(ythip_ctx* ctx, int32_t* num_trees, int64_t* num_nodes, int64_t* num_prims) {
  if (!ctx || !ctx->have_bvh) return fail(ctx, YTHIP_ERR_STATE, "no bvh resident");
  *num_trees = (int32_t)ctx->h_bvh.node_offset.size() - 1;
  *num_nodes = (int64_t)ctx->h_bvh.nodes.size();
  *num_prims = (int64_t)ctx->h_bvh.prims.size();
  return YTHIP_OK;
}

int ythip_bvh_download(ythip_ctx* ctx, int64_t* node_offset, int64_t* prim_offset, ythip_bvh_node* nodes,
    int32_t* primitives) {
  if (!ctx || !ctx->have_bvh) return fail(ctx, YTHIP_ERR_STATE, "no bvh resident");
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  if (int rc = ensure_host_bvh(ctx)) return rc;
  auto& b = ctx->h_bvh;
  std::memcpy(node_offset, b.node_offset.data(), b.node_offset.size() * sizeof(int64_t));
  std::memcpy(prim_offset, b.prim_offset.data(), b.prim_offset.size() * sizeof(int64_t));
  if (nodes && !b.nodes.empty()) std::memcpy(nodes, b.nodes.data(), b.nodes.size() * sizeof(ythip_bvh_node));
  if (primitives && !b.prims.empty()) std::memcpy(primitives, b.prims.data(), b.prims.size() * sizeof(int32_t));
  return YTHIP_OK;
}

int ythip_host_bvh_build(const ythip_scene* sc, int highquality, ythip_hostbvh** out) {
  if (!sc || !out) return fail(nullptr, YTHIP_ERR_INVALID, "null argument");
  auto b = new ythost::flat_bvh(ythost::make_scene_bvh(*sc, highquality != 0));
  *out   = reinterpret_cast<ythip_hostbvh*>(b);
  return YTHIP_OK;
}
int ythip_host_bvh_view(const ythip_hostbvh* bvh, ythip_bvh* view) {
  if (!bvh || !view) return YTHIP_ERR_INVALID;
  auto b            = reinterpret_cast<const ythost::flat_bvh*>(bvh);
  view->num_trees   = (int)b->node_offset.size() - 1;
  view->node_offset = b->node_offset.data();
  view->prim_offset = b->prim_offset.data();
  view->nodes       = b->nodes.data();
  view->primitives  = b->prims.data();
  return YTHIP_OK;
}
int ythip_host_bvh_refit(ythip_hostbvh* bvh, const ythip_scene* sc, const int32_t* updated_shapes, int32_t num_shapes) {
  if (!bvh || !sc || (num_shapes > 0 && !updated_shapes)) return fail(nullptr, YTHIP_ERR_INVALID, "null argument");
  auto& b = *reinterpret_cast<ythost::flat_bvh*>(bvh);
  if ((int)b.node_offset.size() != sc->num_shapes + 2)
    return fail(nullptr, YTHIP_ERR_INVALID, "bvh was built for %d shapes, scene has %d", (int)b.node_offset.size() - 2,
        sc->num_shapes);
  for (int k = 0; k < num_shapes; k++)
    if (updated_shapes[k] < 0 || updated_shapes[k] >= sc->num_shapes)
      return fail(nullptr, YTHIP_ERR_INVALID, "shape %d out of range [0,%d)", updated_shapes[k], sc->num_shapes);
  ythost::update_scene_bvh(b, *sc, updated_shapes, num_shapes);
  return YTHIP_OK;
}
void ythip_host_bvh_free(ythip_hostbvh* bvh) { delete reinterpret_cast<ythost::flat_bvh*>(bvh); }

int ythip_host_lights_build(const ythip_scene* sc, ythip_hostlights** out) {
  if (!sc || !out) return fail(nullptr, YTHIP_ERR_INVALID, "null argument");
  auto l = new ythost::flat_lights(ythost::make_trace_lights(*sc));
  *out   = reinterpret_cast<ythip_hostlights*>(l);
  return YTHIP_OK;
}
int ythip_host_lights_view(const ythip_hostlights* lights, ythip_lights* view) {
  if (!lights || !view) return YTHIP_ERR_INVALID;
  auto l           = reinterpret_cast<const ythost::flat_lights*>(lights);
  view->num_lights = (int)l->lights.size();
  view->lights     = l->lights.data();
  view->num_cdf    = (int64_t)l->cdf.size();
  view->cdf        = l->cdf.data();
  return YTHIP_OK;
}
void ythip_host_lights_free(ythip_hostlights* lights) { delete reinterpret_cast<ythost::flat_lights*>(lights); }

int ythip_build_lights(ythip_ctx* ctx, const ythip_scene* sc) {
  if (!ctx || !sc) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  if (!ctx->have_scene) return fail(ctx, YTHIP_ERR_STATE, "upload_scene first");
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  ctx->h_lights = ythost::make_trace_lights(*sc);
  return upload_lights_impl(ctx);
}

int ythip_upload_lights(ythip_ctx* ctx, const ythip_lights* lights) {
  if (!ctx || !lights) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  if (!ctx->have_scene) return fail(ctx, YTHIP_ERR_STATE, "upload_scene first");
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  ctx->h_lights.lights.assign(lights->lights, lights->lights + lights->num_lights);
  ctx->h_lights.cdf.assign(lights->cdf, lights->cdf + lights->num_cdf);
  return upload_lights_impl(ctx);
}

int ythip_lights_sizes(ythip_ctx* ctx, int32_t* num_lights, int64_t* num_cdf) {
  if (!ctx || !ctx->have_lights) return fail(ctx, YTHIP_ERR_STATE, "no lights resident");
  *num_lights = (int32_t)ctx->h_lights.lights.size();
  *num_cdf    = (int64_t)ctx->h_lights.cdf.size();
  return YTHIP_OK;
}

int ythip_lights_download(ythip_ctx* ctx, ythip_light* lights, float* cdf) {
  if (!ctx || !ctx->have_lights) return fail(ctx, YTHIP_ERR_STATE, "no lights resident");
  auto& L = ctx->h_lights;
  if (lights && !L.lights.empty()) std::memcpy(lights, L.lights.data(), L.lights.size() * sizeof(ythip_light));
  if (cdf && !L.cdf.empty()) std::memcpy(cdf, L.cdf.data(), L.cdf.size() * sizeof(float));
  return YTHIP_OK;
}

// make_trace_state size rule — yocto_trace.cpp:1499-1505
int ythip_state_size(const ythip_camera* camera, int resolution, int* width, int* height) {
  if (!camera || !width || !height) return YTHIP_ERR_INVALID;
  if (camera->aspect >= 1) {
    *width  = resolution;
    *height = (int)std::round(resolution / camera->aspect);
  } else {
    *height = resolution;
    *width  = (int)std::round(resolution * camera->aspect);
  }
  return YTHIP_OK;
}

int ythip_make_rngs(uint64_t seed, int64_t n, uint64_t* rngs) {
  if (!rngs || n < 0) return YTHIP_ERR_INVALID;
  ythost::make_rngs(seed, n, rngs);
  return YTHIP_OK;
}

int ythip_state_create(ythip_ctx* ctx, int width, int height, int row_begin, int row_end) {
  return ythip_state_create_striped(ctx, width, height, row_begin, row_end, 0, 1);
}

int ythip_state_local_width(int width, int col_first, int col_stride) {
  if (width <= 0 || col_first < 0 || col_stride < 1) return -1;
  int lw = 0;
  for (int c = col_first; c * YT_TILE < width; c += col_stride) lw += std::min(YT_TILE, width - c * YT_TILE);
  return lw;
}

int ythip_state_create_striped(ythip_ctx* ctx, int width, int height, int row_begin, int row_end, int col_first,
    int col_stride) {
  if (!ctx) return YTHIP_ERR_INVALID;
  if (width <= 0 || height <= 0 || row_begin < 0 || row_end > height || row_begin >= row_end)
    return fail(ctx, YTHIP_ERR_INVALID, "bad state geometry %dx%d rows [%d,%d)", width, height, row_begin, row_end);
  int lwidth = ythip_state_local_width(width, col_first, col_stride);
  if (lwidth <= 0)
    return fail(ctx, YTHIP_ERR_INVALID, "bad column striping first %d stride %d for width %d (no pixel owned)",
        col_first, col_stride, width);
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  free_all(ctx->state_allocs);
  ctx->have_state    = false;
  ctx->have_denoised = false;
  ctx->state_bound   = false;
  auto& st         = ctx->st;
  st               = DState{};
  st.width         = width;
  st.height        = height;
  st.row_begin     = row_begin;
  st.rows          = row_end - row_begin;
  st.lwidth        = lwidth;
  st.col_first     = col_first;
  st.col_stride    = col_stride;
  long long npix   = (long long)lwidth * st.rows;
  if (npix > 0x7fffffffll / 4) return fail(ctx, YTHIP_ERR_INVALID, "state too large");
  st.npix      = (int)npix;
  st.tiles_x   = (lwidth + YT_TILE - 1) / YT_TILE;
  st.tiles_y   = (st.rows + YT_TILE_H - 1) / YT_TILE_H;
  st.nblocks   = st.tiles_x * st.tiles_y;
  st.nslots    = st.nblocks * YT_BLOCK;
  size_t n = (size_t)npix, ns = (size_t)st.nslots;
  int    rc;
#define AL(field, count) \
  if ((rc = dalloc(ctx, ctx->state_allocs, &st.field, (size_t)(count)))) return rc;
  AL(image, n);
  AL(albedo, 3 * n);
  AL(normal, 3 * n);
  AL(hits, n);
  AL(rngs, n);
  AL(vol_a, ns);
  AL(vol_b, ns);
  AL(pend, ns);
#undef AL
  HIPCHECK(ctx, hipMemsetAsync(st.image, 0, n * sizeof(float4), ctx->stream));
  HIPCHECK(ctx, hipMemsetAsync(st.albedo, 0, 3 * n * sizeof(float), ctx->stream));
  HIPCHECK(ctx, hipMemsetAsync(st.normal, 0, 3 * n * sizeof(float), ctx->stream));
  HIPCHECK(ctx, hipMemsetAsync(st.hits, 0, n * sizeof(int), ctx->stream));
  HIPCHECK(ctx, hipMemsetAsync(st.rngs, 0, n * sizeof(ulonglong2), ctx->stream));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->nhit_a     = nullptr;
  ctx->nhit_e     = nullptr;
  ctx->nee_a = ctx->nee_b = ctx->nee_c = ctx->nee_d = ctx->nee_e = nullptr;
  ctx->samples    = 0;
  ctx->have_state = true;
  // A new state with the tile grid of the previous one (a viewer re-creates the state on every
  // camera edit, apps/ytrace.cpp:189-204) keeps the recorded tile costs: its first batch is
  // already launched most expensive tile first (the view changed a little, the costs did too).
  const bool same_grid = ctx->d_tile_cost && ctx->order_tiles_x == st.tiles_x && ctx->order_tiles_y == st.tiles_y;
  if (!same_grid) {
    free_all(ctx->order_allocs);
    ctx->d_tile_cost = nullptr, ctx->d_tile_perm = nullptr, ctx->d_sort_temp = nullptr;
    ctx->have_tile_costs = false;
    if (ctx->lpt > 0 && st.nblocks > 1) {
      const size_t nb = (size_t)st.nblocks;
      if ((rc = dalloc(ctx, ctx->order_allocs, &ctx->d_tile_cost, nb))) return rc;
      if ((rc = dalloc(ctx, ctx->order_allocs, &ctx->d_tile_perm, nb))) return rc;
      ctx->sort_temp_bytes = ytorder::temp_bytes(st.nblocks);
      unsigned char* tmp   = nullptr;
      if ((rc = dalloc(ctx, ctx->order_allocs, &tmp, std::max<size_t>(ctx->sort_temp_bytes, 16)))) return rc;
      ctx->d_sort_temp = tmp;
      ctx->order_tiles_x = st.tiles_x, ctx->order_tiles_y = st.tiles_y;
    }
  }
  ctx->lpt_age = 0;  // (the order is recomputed from the kept costs at the first launch)
  ctx->pool_tune = 0, ctx->pool_on = false;  // (a new state: the pixel pool is measured again)
  return YTHIP_OK;
}

int ythip_state_upload(ythip_ctx* ctx, const float* image, const float* albedo, const float* normal,
    const int32_t* hits, const uint64_t* rngs, int samples) {
  if (!ctx || !ctx->have_state) return fail(ctx, YTHIP_ERR_STATE, "state_create first");
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  ctx->have_denoised = false;
  size_t n = (size_t)ctx->st.npix;
  if (image) HIPCHECK(ctx, ctx->xfer.h2d(ctx->stream, ctx->st.image, image, n * 16));
  if (albedo) HIPCHECK(ctx, ctx->xfer.h2d(ctx->stream, ctx->st.albedo, albedo, n * 12));
  if (normal) HIPCHECK(ctx, ctx->xfer.h2d(ctx->stream, ctx->st.normal, normal, n * 12));
  if (hits) HIPCHECK(ctx, ctx->xfer.h2d(ctx->stream, ctx->st.hits, hits, n * 4));
  if (rngs) HIPCHECK(ctx, ctx->xfer.h2d(ctx->stream, ctx->st.rngs, rngs, n * 16));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->samples = samples;
  return YTHIP_OK;
}

int ythip_state_download(ythip_ctx* ctx, float* image, float* albedo, float* normal, int32_t* hits,
    uint64_t* rngs, int* samples) {
  if (!ctx || !ctx->have_state) return fail(ctx, YTHIP_ERR_STATE, "state_create first");
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  size_t n = (size_t)ctx->st.npix;
  if (image) HIPCHECK(ctx, ctx->xfer.d2h(ctx->stream, image, ctx->st.image, n * 16));
  if (albedo) HIPCHECK(ctx, ctx->xfer.d2h(ctx->stream, albedo, ctx->st.albedo, n * 12));
  if (normal) HIPCHECK(ctx, ctx->xfer.d2h(ctx->stream, normal, ctx->st.normal, n * 12));
  if (hits) HIPCHECK(ctx, ctx->xfer.d2h(ctx->stream, hits, ctx->st.hits, n * 4));
  if (rngs) HIPCHECK(ctx, ctx->xfer.d2h(ctx->stream, rngs, ctx->st.rngs, n * 16));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (samples) *samples = ctx->samples;
  return YTHIP_OK;
}

int ythip_get_image(ythip_ctx* ctx, float* image) {
  if (!ctx || !ctx->have_state) return fail(ctx, YTHIP_ERR_STATE, "state_create first");
  if (!image) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  HIPCHECK(ctx, ctx->xfer.d2h(ctx->stream, image, ctx->st.image, (size_t)ctx->st.npix * 16));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
  return YTHIP_OK;
}

namespace {
int guide_image(ythip_ctx* ctx, const float* rgb, float* image) {
  if (!ctx || !ctx->have_state) return fail(ctx, YTHIP_ERR_STATE, "state_create first");
  if (!image) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  const size_t       n = (size_t)ctx->st.npix;
  std::vector<void*> tmp;
  float4*            d = nullptr;
  int                rc;
  if ((rc = dalloc(ctx, tmp, &d, n))) return rc;
  hipLaunchKernelGGL(k_guide_image, dim3(grid_for((long long)n)), dim3(YT_BLOCK), 0, ctx->stream, rgb, (int)n, d);
  auto e1 = ctx->xfer.d2h(ctx->stream, image, d, n * 16);
  auto e2 = hipStreamSynchronize(ctx->stream);
  free_all(tmp);
  if (e1 != hipSuccess || e2 != hipSuccess) return fail(ctx, YTHIP_ERR_HIP, "guide image download failed");
  return YTHIP_OK;
}
}  // namespace

int ythip_get_albedo_image(ythip_ctx* ctx, float* image) {
  return guide_image(ctx, ctx && ctx->have_state ? ctx->st.albedo : nullptr, image);
}
int ythip_get_normal_image(ythip_ctx* ctx, float* image) {
  return guide_image(ctx, ctx && ctx->have_state ? ctx->st.normal : nullptr, image);
}

int ythip_tonemap_image(ythip_ctx* ctx, float exposure, int filmic, int srgb, float* ldr, uint8_t* ldr_bytes) {
  if (!ctx || !ctx->have_state) return fail(ctx, YTHIP_ERR_STATE, "state_create first");
  if (!ldr && !ldr_bytes) return fail(ctx, YTHIP_ERR_INVALID, "no output buffer");
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  const size_t       n = (size_t)ctx->st.npix;
  std::vector<void*> tmp;
  float4*            d_f = nullptr;
  uchar4*            d_b = nullptr;
  int                rc  = 0;
  if (ldr && (rc = dalloc(ctx, tmp, &d_f, n))) return rc;
  if (ldr_bytes && (rc = dalloc(ctx, tmp, &d_b, n))) {
    free_all(tmp);
    return rc;
  }
  hipLaunchKernelGGL(k_tonemap, dim3(grid_for((long long)n)), dim3(YT_BLOCK), 0, ctx->stream, ctx->st.image, (int)n,
      exposure, filmic, srgb, d_f, d_b);
  auto e1 = ldr ? ctx->xfer.d2h(ctx->stream, ldr, d_f, n * 16) : hipSuccess;
  auto e2 = ldr_bytes ? ctx->xfer.d2h(ctx->stream, ldr_bytes, d_b, n * 4) : hipSuccess;
  auto e3 = hipStreamSynchronize(ctx->stream);
  free_all(tmp);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return fail(ctx, YTHIP_ERR_HIP, "tonemap_image failed");
  return YTHIP_OK;
}

// ---- denoiser (yt_denoise.h; the slot of denoise_image, yocto_trace.cpp:1794-1872) ---------
void ythip_denoise_default_params(ythip_denoise_params* p) {
  if (!p) return;
  p->levels       = 5;
  p->sigma_color  = 4.0f;
  p->sigma_normal = 0.35f;
  p->sigma_albedo = 0.1f;
}
namespace {
int denoise_buffers(ythip_ctx* ctx, size_t n) {
  if (ctx->dn_pixels == n && ctx->dn_a) return YTHIP_OK;
  free_all(ctx->denoise_allocs);
  ctx->dn_a = ctx->dn_b = ctx->dn_gn = ctx->dn_ga = ctx->dn_out = nullptr;
  ctx->dn_pixels = 0;
  int rc;
  if ((rc = dalloc(ctx, ctx->denoise_allocs, &ctx->dn_a, n))) return rc;
  if ((rc = dalloc(ctx, ctx->denoise_allocs, &ctx->dn_b, n))) return rc;
  if ((rc = dalloc(ctx, ctx->denoise_allocs, &ctx->dn_gn, n))) return rc;
  if ((rc = dalloc(ctx, ctx->denoise_allocs, &ctx->dn_ga, n))) return rc;
  if ((rc = dalloc(ctx, ctx->denoise_allocs, &ctx->dn_out, n))) return rc;
  ctx->dn_pixels = n;
  return YTHIP_OK;
}
// image / albedo / normal on the device (trace_state layout) → ctx->dn_out
int denoise_run(ythip_ctx* ctx, const ythip_denoise_params* up, int width, int height, const float4* image,
    const float* albedo, const float* normal) {
  ythip_denoise_params dp;
  ythip_denoise_default_params(&dp);
  if (up) dp = *up;
  if (dp.levels < 0 || dp.levels > 16) return fail(ctx, YTHIP_ERR_INVALID, "denoise levels %d outside [0,16]", dp.levels);
  if (!(dp.sigma_color > 0) || !(dp.sigma_normal > 0) || !(dp.sigma_albedo > 0))
    return fail(ctx, YTHIP_ERR_INVALID, "denoise sigmas must be positive");
  const size_t n = (size_t)width * height;
  int          rc;
  if ((rc = denoise_buffers(ctx, n))) return rc;
  ytdn::Params p = {width, height, dp.levels, 1.0f / (dp.sigma_normal * dp.sigma_normal),
      1.0f / (dp.sigma_albedo * dp.sigma_albedo), 1.0f / (dp.sigma_color * dp.sigma_color)};
  const int  threads = ytdn::BX * ytdn::BY;
  const dim3 flat((unsigned)((n + threads - 1) / threads));
  const dim3 grid((unsigned)((width + ytdn::BX - 1) / ytdn::BX), (unsigned)((height + ytdn::BY - 1) / ytdn::BY));
  hipLaunchKernelGGL(ytdn::k_prep, flat, dim3(threads), 0, ctx->stream, image, albedo, normal, (int)n, ctx->dn_a,
      ctx->dn_gn, ctx->dn_ga);
  float4 *src = ctx->dn_a, *dst = ctx->dn_b;
  float   scale = 1;  // 4^l: the colour tolerance halves per level (Dammertz et al. 2010, §3)
  for (int l = 0; l < dp.levels; l++) {
    const int step = 1 << l;
    if (step <= 16 && !ctx->denoise_simple) {
      // one workgroup per tile of one residue class's sub-image (class 0 has the most tiles)
      const int sw = (width + step - 1) / step, sh = (height + step - 1) / step;
      const dim3 g((unsigned)((sw + ytdn::TX - 1) / ytdn::TX), (unsigned)((sh + ytdn::TY - 1) / ytdn::TY),
          (unsigned)(step * step));
      hipLaunchKernelGGL(ytdn::k_atrous_lds, g, dim3(ytdn::TX * ytdn::TY), 0, ctx->stream, src, ctx->dn_gn, ctx->dn_ga, dst,
          p, step, p.inv_sc2 * scale);
    } else
      hipLaunchKernelGGL(ytdn::k_atrous, grid, dim3(threads), 0, ctx->stream, src, ctx->dn_gn, ctx->dn_ga, dst, p, step,
          p.inv_sc2 * scale);
    std::swap(src, dst);
    scale *= 4;
  }
  hipLaunchKernelGGL(ytdn::k_finish, flat, dim3(threads), 0, ctx->stream, src, ctx->dn_ga, (int)n, ctx->dn_out);
  HIPCHECK(ctx, hipGetLastError());
  return YTHIP_OK;
}
}  // namespace

int ythip_denoise_image(ythip_ctx* ctx, const ythip_denoise_params* params, int32_t width, int32_t height,
    const float* render, const float* albedo, const float* normal, float* denoised) {
  if (!ctx || !render || !albedo || !normal || !denoised) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  if (width <= 0 || height <= 0) return fail(ctx, YTHIP_ERR_INVALID, "bad image size %d x %d", width, height);
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  const size_t       n = (size_t)width * height;
  std::vector<void*> tmp;
  float4*            d_img = nullptr;
  float *            d_alb = nullptr, *d_nrm = nullptr;
  int                rc;
  if ((rc = dalloc(ctx, tmp, &d_img, n)) || (rc = dalloc(ctx, tmp, &d_alb, 3 * n)) || (rc = dalloc(ctx, tmp, &d_nrm, 3 * n))) {
    free_all(tmp);
    return rc;
  }
  auto e1 = ctx->xfer.h2d(ctx->stream, d_img, render, n * 16);
  auto e2 = ctx->xfer.h2d(ctx->stream, d_alb, albedo, n * 12);
  auto e3 = ctx->xfer.h2d(ctx->stream, d_nrm, normal, n * 12);
  rc      = (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) ? fail(ctx, YTHIP_ERR_HIP, "denoise upload failed")
                                                                        : denoise_run(ctx, params, width, height, d_img, d_alb, d_nrm);
  ctx->have_denoised = false;  // dn_out no longer belongs to the resident state
  if (rc == YTHIP_OK) {
    auto e4 = ctx->xfer.d2h(ctx->stream, denoised, ctx->dn_out, n * 16);
    auto e5 = hipStreamSynchronize(ctx->stream);
    if (e4 != hipSuccess || e5 != hipSuccess) rc = fail(ctx, YTHIP_ERR_HIP, "denoise download failed");
  } else {
    (void)hipStreamSynchronize(ctx->stream);
  }
  free_all(tmp);
  return rc;
}

int ythip_denoise_state(ythip_ctx* ctx, const ythip_denoise_params* params, float* denoised) {
  if (!ctx || !ctx->have_state) return fail(ctx, YTHIP_ERR_STATE, "state_create first");
  const auto& st = ctx->st;
  if (st.lwidth != st.width || st.col_stride != 1 || st.row_begin != 0 || st.rows != st.height)
    return fail(ctx, YTHIP_ERR_STATE, "denoise needs the whole frame on one device (this state is a slice: gather, then "
                                      "ythip_denoise_image)");
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  int rc = denoise_run(ctx, params, st.width, st.height, st.image, st.albedo, st.normal);
  if (rc != YTHIP_OK) return rc;
  ctx->have_denoised = true;
  if (denoised) HIPCHECK(ctx, ctx->xfer.d2h(ctx->stream, denoised, ctx->dn_out, (size_t)st.npix * 16));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
  return YTHIP_OK;
}

int ythip_state_device_denoised(ythip_ctx* ctx, void** image) {
  if (!ctx || !image || !ctx->have_state) return fail(ctx, YTHIP_ERR_STATE, "state_create first");
  if (!ctx->have_denoised) return fail(ctx, YTHIP_ERR_STATE, "ythip_denoise_state first");
  *image = ctx->dn_out;
  return YTHIP_OK;
}

int ythip_state_bind_device(ythip_ctx* ctx, void* image, void* albedo, void* normal, void* hits, void* rngs) {
  if (!ctx || !ctx->have_state) return fail(ctx, YTHIP_ERR_STATE, "state_create first");
  if (!image || !albedo || !normal || !hits || !rngs) return fail(ctx, YTHIP_ERR_INVALID, "null device pointer");
  ctx->st.image    = (float4*)image;
  ctx->st.albedo   = (float*)albedo;
  ctx->st.normal   = (float*)normal;
  ctx->st.hits     = (int*)hits;
  ctx->st.rngs     = (ulonglong2*)rngs;
  ctx->state_bound = true;
  return YTHIP_OK;
}

int ythip_state_get_samples(ythip_ctx* ctx, int* samples) {
  if (!ctx || !samples || !ctx->have_state) return fail(ctx, YTHIP_ERR_STATE, "state_create first");
  *samples = ctx->samples;
  return YTHIP_OK;
}

int ythip_state_device_image(ythip_ctx* ctx, void** image) {
  if (!ctx || !image || !ctx->have_state) return fail(ctx, YTHIP_ERR_STATE, "state_create first");
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
  *image = ctx->st.image;
  return YTHIP_OK;
}

int ythip_poll(ythip_ctx* ctx) {
  if (!ctx) return -YTHIP_ERR_INVALID;
  if (hipSetDevice(ctx->device) != hipSuccess) return -YTHIP_ERR_HIP;
  auto e = hipStreamQuery(ctx->stream);
  if (e == hipSuccess) return 1;
  if (e == hipErrorNotReady) return 0;
  fail(ctx, YTHIP_ERR_HIP, "hipStreamQuery failed: %s", hipGetErrorString(e));
  return -YTHIP_ERR_HIP;
}

int ythip_state_set_samples(ythip_ctx* ctx, int samples) {
  if (!ctx || !ctx->have_state) return fail(ctx, YTHIP_ERR_STATE, "state_create first");
  ctx->samples = samples;
  return YTHIP_OK;
}

namespace {
// trace_cancel for the batch in flight (the reference's stop flag, yocto_trace.cpp:1636-1637).
int raise_stop(ythip_ctx* ctx) {
  // the number of the batch in flight (or of the last one: then nobody is listening) into the pinned
  // host word; the kernels relay it (alloc_stop_word).  Callable from any thread, no HIP call.
  __atomic_store_n(ctx->stop_host, ctx->stop_gen.load(), __ATOMIC_RELEASE);
  return YTHIP_OK;
}
// every batch (and every trace_sample call) is numbered before its launches are enqueued
void begin_batch(ythip_ctx* ctx) {
  int g = ctx->stop_gen.load() + 1;
  if (g <= 0) g = 1;  // (0 is what the word holds before the first cancel)
  ctx->stop_gen.store(g);
}
}  // namespace

namespace {
// On by default since round 3 (YTHIP_LPT_PROBE=0 switches it off; +12 % on a single 64-spp batch of
// Cornell-1M).  Round 2 left it off because the in-batch cancellation test then missed its 50 ms
// bound.  That was not the split: the cancel word lived in ordinary device memory and was written by
// a memset on a side stream — an XCD's L2 keeps such a line until it happens to be evicted, and a
// stream-ordered write queues behind the batch's launches: 45-200 ms (profiles/r03_cancel_latency.txt),
// the probe's two extra launches only pushed a marginal latency over the bound.  Now the host stores
// the batch number into a pinned host word, every 64th loop iteration of a workgroup reads that word
// over the fabric and relays it into a device word all workgroups poll (yt_kernels.h, relay_stop):
// ~1 ms, no HIP call; both launches of a split batch carry the same batch number (begin_batch), so
// one cancel stops both.
// A batch whose tile costs are not known yet (first batch of a tile grid) and that is long
// enough to care is launched as 1 + (batch - 1) samples: the first launch records what every
// tile costs, the second is handed out most expensive tile first (yt_order.hip).  Two launches
// of a progressive render: the same samples in the same order, bit-identical (tested).
int enqueue_batch(ythip_ctx* ctx, const ythip_params* params, const volatile int32_t* stop) {
  const bool probe = ctx->lpt_probe && ctx->lpt > 0 && ctx->d_tile_cost && !ctx->have_tile_costs &&
                     (ctx->prof_mode & 2) == 0 && params->batch >= 8 && ctx->st.nblocks > 4096 &&
                     ctx->samples < params->samples;
  if (!probe) return enqueue_samples(ctx, params, stop);
  ythip_params first = *params, rest = *params;
  first.batch        = 1;
  rest.batch         = params->batch - 1;
  rest.samples       = std::numeric_limits<int>::max();  // (the whole batch runs, as in the reference: yocto_trace.cpp:1598 tests once)
  int rc = enqueue_samples(ctx, &first, stop);
  return rc ? rc : enqueue_samples(ctx, &rest, stop);
}
}  // namespace

int ythip_trace_samples(ythip_ctx* ctx, const ythip_params* params, const volatile int32_t* stop) {
  if (!ctx || !params) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  const int samples_before = ctx->samples;
  begin_batch(ctx);
  int       rc             = enqueue_batch(ctx, params, stop);
  if (rc) return rc;
  bool cancelled = false;
  if (stop) {
    // the reference checks context.stop before every sample of every pixel; here the host
    // watches the caller's flag while the batch runs and relays it to the device
    if (!ctx->done_event) HIPCHECK(ctx, hipEventCreateWithFlags(&ctx->done_event, hipEventDisableTiming));
    HIPCHECK(ctx, hipEventRecord(ctx->done_event, ctx->stream));
    while (hipEventQuery(ctx->done_event) == hipErrorNotReady) {
      if (*stop) {
        if ((rc = raise_stop(ctx))) return rc;
        cancelled = true;
        break;
      }
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  }
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
  harvest_events(ctx);
  if (cancelled) {
    // as in the reference, some pixels have taken more samples of the batch than others and
    // state.samples does not advance (yocto_trace.cpp:1636-1641)
    ctx->samples = samples_before;
    return fail(ctx, YTHIP_ERR_CANCELLED, "cancelled");
  }
  return YTHIP_OK;
}

int ythip_set_pixel_pool(ythip_ctx* ctx, int mode, int workgroups) {
  if (!ctx || mode < 0 || mode > 2) return fail(ctx, YTHIP_ERR_INVALID, "pixel pool mode must be 0, 1 or 2");
  ctx->pixel_pool = mode, ctx->pool_tune = 0, ctx->pool_on = false;
  if (workgroups > 0) ctx->pool_blocks = workgroups;
  return YTHIP_OK;
}
int ythip_get_pixel_pool(ythip_ctx* ctx, ythip_pool_info* info) {
  if (!ctx || !info) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  *info            = {};
  info->mode       = ctx->pixel_pool;
  info->workgroups = ctx->pool_blocks;
  info->decided    = ctx->pixel_pool >= 2 || ctx->pool_tune == 3;
  info->on         = ctx->pixel_pool >= 2 || (ctx->pool_tune == 3 && ctx->pool_on);
  if (ctx->pool_tune == 3 && ctx->pool_samples[0] > 0 && ctx->pool_samples[1] > 0) {
    info->plain_ms_per_sample = (float)(ctx->pool_ms[0] / ctx->pool_samples[0]);
    info->pool_ms_per_sample  = (float)(ctx->pool_ms[1] / ctx->pool_samples[1]);
  }
  return YTHIP_OK;
}

int ythip_last_launch_fastmath(ythip_ctx* ctx) { return ctx && ctx->last_launch_fast ? 1 : 0; }

int ythip_cancel(ythip_ctx* ctx) {
  if (!ctx) return YTHIP_ERR_INVALID;
  return raise_stop(ctx);
}

int ythip_trace_sample(ythip_ctx* ctx, const ythip_params* params, int i, int j, int sample) {
  if (!ctx || !params) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  if (!ctx->have_state) return fail(ctx, YTHIP_ERR_STATE, "state_create first");
  auto& st = ctx->st;
  if (sample < 0) return fail(ctx, YTHIP_ERR_INVALID, "sample must be >= 0");
  ctx->have_denoised = false;
  int tc = i / YT_TILE, dc = tc - st.col_first;
  if (i < 0 || i >= st.width || j < st.row_begin || j >= st.row_begin + st.rows || dc < 0 || dc % st.col_stride)
    return fail(ctx, YTHIP_ERR_INVALID, "pixel (%d,%d) is not in this slice of the %dx%d frame", i, j, st.width,
        st.height);
  int pix = (j - st.row_begin) * st.lwidth + (dc / st.col_stride) * YT_TILE + i % YT_TILE;
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  begin_batch(ctx);
  int rc = enqueue_samples(ctx, params, nullptr, pix, sample);
  if (rc) return rc;
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
  harvest_events(ctx);
  return YTHIP_OK;
}

int ythip_trace_samples_async(ythip_ctx* ctx, const ythip_params* params) {
  if (!ctx || !params) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  begin_batch(ctx);
  return enqueue_batch(ctx, params, nullptr);
}

static int intersect_impl(ythip_ctx* ctx, const int32_t* instances, const ythip_ray* rays, int64_t n,
    int find_any, ythip_hit* hits) {
  if (!ctx || (n > 0 && (!rays || !hits))) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  if (!ctx->have_scene || !ctx->have_bvh) return fail(ctx, YTHIP_ERR_STATE, "scene and bvh must be resident");
  if (n == 0) return YTHIP_OK;
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  if (instances)
    for (int64_t k = 0; k < n; k++)
      if (instances[k] < 0 || instances[k] >= ctx->ds.num_instances)
        return fail(ctx, YTHIP_ERR_INVALID, "instance id %d out of range", instances[k]);
  ythip_ray* d_rays = nullptr;
  ythip_hit* d_hits = nullptr;
  int*       d_inst = nullptr;
  std::vector<void*> tmp;
  int                rc;
  if ((rc = dalloc(ctx, tmp, &d_rays, (size_t)n)) || (rc = dalloc(ctx, tmp, &d_hits, (size_t)n)) ||
      (instances && (rc = dalloc(ctx, tmp, &d_inst, (size_t)n)))) {
    free_all(tmp);
    return rc;
  }
  auto cleanup = [&](int code) {
    free_all(tmp);
    return code;
  };
  if (ctx->xfer.h2d(ctx->stream, d_rays, rays, n * sizeof(ythip_ray)) != hipSuccess)
    return cleanup(fail(ctx, YTHIP_ERR_HIP, "ray upload failed"));
  if (instances &&
      ctx->xfer.h2d(ctx->stream, d_inst, instances, n * sizeof(int)) != hipSuccess)
    return cleanup(fail(ctx, YTHIP_ERR_HIP, "instance upload failed"));
  bool count = (ctx->prof_mode & 2) != 0;
  {
  EvScope ev(ctx, 0);  // profiling bit 0: the kernel alone (ythip_get_stats: trace_ms / trace_launches)
  if (count)
    hipLaunchKernelGGL((k_intersect_batch<true, false>), dim3(grid_for(n)), dim3(YT_BLOCK), 0, ctx->stream, ctx->ds,
        d_rays, d_inst, (long long)n, find_any, d_hits, ctx->d_counters);
  else if (ctx->use_wide())
    hipLaunchKernelGGL((k_intersect_batch<false, true>), dim3(grid_for(n)), dim3(YT_BLOCK), 0, ctx->stream, ctx->ds,
        d_rays, d_inst, (long long)n, find_any, d_hits, (unsigned long long*)nullptr);
  else
    hipLaunchKernelGGL((k_intersect_batch<false, false>), dim3(grid_for(n)), dim3(YT_BLOCK), 0, ctx->stream, ctx->ds,
        d_rays, d_inst, (long long)n, find_any, d_hits, (unsigned long long*)nullptr);
  }
  if (ctx->xfer.d2h(ctx->stream, hits, d_hits, n * sizeof(ythip_hit)) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess)
    return cleanup(fail(ctx, YTHIP_ERR_HIP, "intersect batch failed: %s", hipGetErrorString(hipGetLastError())));
  return cleanup(YTHIP_OK);
}

int ythip_intersect_batch(ythip_ctx* ctx, const ythip_ray* rays, int64_t n, int find_any, ythip_hit* hits) {
  return intersect_impl(ctx, nullptr, rays, n, find_any, hits);
}
int ythip_intersect_instance_batch(ythip_ctx* ctx, const int32_t* instances, const ythip_ray* rays, int64_t n,
    int find_any, ythip_hit* hits) {
  if (!instances && n > 0) return fail(ctx, YTHIP_ERR_INVALID, "null instances");
  return intersect_impl(ctx, instances, rays, n, find_any, hits);
}

namespace {
__global__ void __launch_bounds__(256) k_test_libm(int fn, const float* x, const float* y, long long n, float* out) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float a = x[i], b = y ? y[i] : 0.0f, r = 0;
  switch (fn) {
    case 0: r = ytm::sinf(a); break;
    case 1: r = ytm::cosf(a); break;
    case 2: r = ytm::expf(a); break;
    case 3: r = ytm::exp2f(a); break;
    case 4: r = ytm::logf(a); break;
    case 5: r = ytm::atanf(a); break;
    case 6: r = ytm::acosf(a); break;
    case 7: r = ytm::atan2f(a, b); break;
    case 8: r = ytm::powf(a, b); break;
    case 9: r = fmodf(a, b); break;
    case 10: r = sqrt_(a); break;
    case 11: r = a / b; break;
  }
  out[i] = r;
}
}  // namespace

int ythip_test_libm(ythip_ctx* ctx, int fn, const float* x, const float* y, int64_t n, float* out) {
  if (!ctx || !x || !out || n < 0 || fn < 0 || fn > 11) return fail(ctx, YTHIP_ERR_INVALID, "bad argument");
  if (n == 0) return YTHIP_OK;
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  std::vector<void*> tmp;
  float *            dx = nullptr, *dy = nullptr, *dout = nullptr;
  int                rc;
  if ((rc = dalloc(ctx, tmp, &dx, (size_t)n)) || (rc = dalloc(ctx, tmp, &dout, (size_t)n)) ||
      (y && (rc = dalloc(ctx, tmp, &dy, (size_t)n)))) {
    free_all(tmp);
    return rc;
  }
  auto e = ctx->xfer.h2d(ctx->stream, dx, x, (size_t)n * 4);
  if (y && e == hipSuccess) e = ctx->xfer.h2d(ctx->stream, dy, y, (size_t)n * 4);
  hipLaunchKernelGGL(k_test_libm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, fn, dx, dy, (long long)n, dout);
  if (e == hipSuccess) e = ctx->xfer.d2h(ctx->stream, out, dout, (size_t)n * 4);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  free_all(tmp);
  if (e != hipSuccess) return fail(ctx, YTHIP_ERR_HIP, "test_libm failed: %s", hipGetErrorString(e));
  return YTHIP_OK;
}

// Does the HOST's libm — the one a reference built on this machine renders with — agree with
// what the device evaluates (yt_libm.h: glibc 2.35's x86-64 `_fma` variants)?  A host whose glibc
// dispatches to other variants (no FMA, another release) produces other last bits in sinf / powf ...,
// the reference on it renders other images, and the drop-in's bit parity with THAT reference is
// gone — silently, unless somebody looks: 1 = the probes agree, 0 = they do not (message in
// ythip_last_error), < 0 = error.  2,304 probe arguments (256 per function) over the ranges the path
// uses; the exhaustive comparison is tests/cpp/libm_check.cpp.
int ythip_host_libm_matches(ythip_ctx* ctx) {
  if (!ctx) return -YTHIP_ERR_INVALID;
  const int          N = 256;
  std::vector<float> x(N), y(N), dev(N);
  unsigned long long lcg = 0x9e3779b97f4a7c15ull;
  auto               uni = [&]() {
    lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
    return (float)((lcg >> 40) * (1.0 / 16777216.0));
  };
  static const char* names[] = {"sinf", "cosf", "expf", "exp2f", "logf", "atanf", "acosf", "atan2f", "powf"};
  for (int fn = 0; fn < 9; fn++) {
    for (int k = 0; k < N; k++) {
      float u = uni(), v = uni();
      switch (fn) {
        case 0: case 1: x[k] = (u - 0.5f) * 4 * 6.2831853f; break;    // a few periods around 0
        case 2: case 3: x[k] = (u - 0.5f) * 40; break;
        case 4: x[k] = u * 100 + 1e-6f; break;
        case 5: x[k] = (u - 0.5f) * 50; break;
        case 6: x[k] = u * 2 - 1; break;
        case 7: x[k] = (u - 0.5f) * 4, y[k] = (v - 0.5f) * 4; break;
        default: x[k] = u * 4 + 1e-4f, y[k] = (v - 0.5f) * 12; break;
      }
    }
    int rc = ythip_test_libm(ctx, fn, x.data(), fn >= 7 ? y.data() : nullptr, N, dev.data());
    if (rc) return -rc;
    for (int k = 0; k < N; k++) {
      float h = 0;
      switch (fn) {
        case 0: h = ::sinf(x[k]); break;
        case 1: h = ::cosf(x[k]); break;
        case 2: h = ::expf(x[k]); break;
        case 3: h = ::exp2f(x[k]); break;
        case 4: h = ::logf(x[k]); break;
        case 5: h = ::atanf(x[k]); break;
        case 6: h = ::acosf(x[k]); break;
        case 7: h = ::atan2f(x[k], y[k]); break;
        default: h = ::powf(x[k], y[k]); break;
      }
      if (std::memcmp(&h, &dev[k], 4) != 0 && !(h != h && dev[k] != dev[k])) {
        fail(ctx, YTHIP_OK, "host %s(%a%s) = %a, the device's restatement of glibc 2.35 gives %a: a reference built on this host "
                            "will not be matched bit for bit", names[fn], (double)x[k], fn >= 7 ? ", ..." : "", (double)h, (double)dev[k]);
        return 0;
      }
    }
  }
  return 1;
}

int ythip_camera_rays(ythip_ctx* ctx, const ythip_params* params, ythip_ray* rays) {
  if (!ctx || !params || !rays) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  if (!ctx->have_scene || !ctx->have_state) return fail(ctx, YTHIP_ERR_STATE, "scene and state must be resident");
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  ythip_ray*         d_rays = nullptr;
  std::vector<void*> tmp;
  int                rc = dalloc(ctx, tmp, &d_rays, (size_t)ctx->st.npix);
  if (rc) return rc;
  auto kp = to_kparams(ctx, params);
  hipLaunchKernelGGL(k_camera_rays, dim3(grid_for(ctx->st.npix)), dim3(YT_BLOCK), 0, ctx->stream, ctx->ds, ctx->st,
      kp, d_rays);
  auto e1 = ctx->xfer.d2h(ctx->stream, rays, d_rays, (size_t)ctx->st.npix * sizeof(ythip_ray));
  auto e2 = hipStreamSynchronize(ctx->stream);
  free_all(tmp);
  if (e1 != hipSuccess || e2 != hipSuccess) return fail(ctx, YTHIP_ERR_HIP, "camera_rays failed");
  return YTHIP_OK;
}

int ythip_set_scheduling(ythip_ctx* ctx, int adaptive_wait) {
  if (!ctx) return YTHIP_ERR_INVALID;
  ctx->hold_policy = adaptive_wait ? 1 : 0;
  return YTHIP_OK;
}

int ythip_set_early_miss(ythip_ctx* ctx, int enable) {
  if (!ctx) return YTHIP_ERR_INVALID;
  ctx->peek_policy = enable ? 1 : 0;
  return YTHIP_OK;
}

int ythip_set_specialization(ythip_ctx* ctx, int enable) {
  if (!ctx) return YTHIP_ERR_INVALID;
  ctx->specialize = enable ? 1 : 0;
  return YTHIP_OK;
}

int ythip_set_traversal(ythip_ctx* ctx, int mode) {
  if (!ctx || mode < 0 || mode > 2) return fail(ctx, YTHIP_ERR_INVALID, "traversal mode must be 0, 1 or 2");
  ctx->traversal_mode = mode;
  return YTHIP_OK;
}

int ythip_set_profiling(ythip_ctx* ctx, int mode) {
  if (!ctx) return YTHIP_ERR_INVALID;
  ctx->prof_mode = mode;
  return YTHIP_OK;
}

int ythip_reset_stats(ythip_ctx* ctx) {
  if (!ctx) return YTHIP_ERR_INVALID;
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
  harvest_events(ctx);
  ctx->stats = ythip_stats{};
  HIPCHECK(ctx, hipMemset(ctx->d_counters, 0, CNT_BANKS * CNT_STRIDE * sizeof(unsigned long long)));
  return YTHIP_OK;
}

int ythip_get_stats(ythip_ctx* ctx, ythip_stats* stats) {
  if (!ctx || !stats) return YTHIP_ERR_INVALID;
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  HIPCHECK(ctx, hipStreamSynchronize(ctx->stream));
  harvest_events(ctx);
  unsigned long long banks[CNT_BANKS * CNT_STRIDE], c[CNT_NUM] = {};
  HIPCHECK(ctx, ctx->xfer.d2h(ctx->stream, banks, ctx->d_counters, sizeof(banks)));
  for (int b = 0; b < CNT_BANKS; b++)
    for (int k = 0; k < CNT_NUM; k++) c[k] += banks[b * CNT_STRIDE + k];
#ifdef YT_TIMING
  {
    unsigned long long t[16] = {};
    for (int b = 0; b < CNT_BANKS; b++)
      for (int k = 6; k < 16; k++) t[k] += banks[b * CNT_STRIDE + k];
    double tot = (double)(t[9] + t[10] + t[11] + t[12]);
    std::fprintf(stderr, "[timing] wave-iterations %llu | extend %.1f%% shade %.1f%% partition+wait %.1f%% "
                         "idle-wave %.1f%% | cycles/wave-iteration %.0f\n",
        t[13], 100 * t[9] / tot, 100 * t[10] / tot, 100 * t[11] / tot, 100 * t[12] / tot, t[13] ? tot / t[13] : 0.0);
#ifdef YT_TIMING_OCC
    std::fprintf(stderr, "[timing] slots holding a ray in the walk (weighted by the longest lane): %.1f of 64 | lane steps per walking slot / longest lane: %.1f%%\n",
        t[7] ? 64.0 * t[14] / t[7] : 0.0, t[14] ? 100.0 * t[6] / t[14] : 0.0);
    std::fprintf(stderr, "[timing] slots running in an iteration (weighted by extend + shade time): %.1f of 64\n", t[8] ? 64.0 * t[15] / t[8] : 0.0);
#else
    std::fprintf(stderr, "[timing] of all wave time: hit shading point %.1f%% | bsdf+sampling after it %.1f%% | "
                         "finish+regenerate (resolve_step) %.1f%%\n",
        100 * t[14] / tot, 100 * t[15] / tot, 100 * t[8] / tot);
#endif
    std::fprintf(stderr, "[timing] traversal lane utilisation (sum of lane steps / 64 x longest lane): %.1f%%\n",
        t[7] ? 100.0 * t[6] / t[7] : 0.0);
  }
#endif
#ifdef YT_WALK_PROFILE
  {
    unsigned long long w[16] = {};
    HIPCHECK(ctx, hipMemcpyFromSymbol(w, HIP_SYMBOL(yt::g_walkprof), sizeof(w)));
    unsigned long long zero[16] = {};
    HIPCHECK(ctx, hipMemcpyToSymbol(HIP_SYMBOL(yt::g_walkprof), zero, sizeof(zero)));
    if (w[3])
      std::fprintf(stderr,
          "[walk] descend phase %.1f%% of the walk cycles, %.1f of 64 lanes per inner iteration | leaf/entry phase %.1f%%: "
          "%.1f lanes with a BLAS leaf + %.1f with an entry per round, %.1f lanes per primitive round (%.2f rounds per leaf round)\n",
          100.0 * w[0] / w[3], w[2] ? (double)w[1] / w[2] : 0.0, 100.0 * (w[3] - w[0]) / w[3], w[5] ? (double)w[4] / w[5] : 0.0,
          w[5] ? (double)w[6] / w[5] : 0.0, w[8] ? (double)w[7] / w[8] : 0.0, w[5] ? (double)w[8] / w[5] : 0.0);
  }
#endif
  *stats           = ctx->stats;
  stats->rays      = (int64_t)c[CNT_RAYS];
  stats->nodes     = (int64_t)c[CNT_NODES];
  stats->triangles = (int64_t)c[CNT_TRIS];
  stats->quads     = (int64_t)c[CNT_QUADS];
  stats->lines     = (int64_t)c[CNT_LINES];
  stats->points    = (int64_t)c[CNT_POINTS];
  stats->instances = (int64_t)c[CNT_INST];
  stats->shades    = (int64_t)c[CNT_SHADES];
  stats->samples   = (int64_t)c[CNT_SAMPLES];
  return YTHIP_OK;
}

}  // extern "C"
