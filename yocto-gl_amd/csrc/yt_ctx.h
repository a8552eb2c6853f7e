// yt_ctx.h — what the translation units of libythip share: the context behind the opaque `ythip_ctx` of include/ythip.h,
// the error / allocation helpers, and the functions that live in another unit than their callers.
//   ythip.hip         the C ABI: context, scene / lights / state uploads, display path, denoiser slot, enqueue of a batch
//   yt_bake.hip       make_trace_bvh: host / device builds of the trees and the bake into the traversal layout of yt_bvh.h
//   yt_trace_*.hip    the k_trace instantiations, by sampler family (yt_launch.h) — the units that take minutes to compile
//   yt_fast.hip       the same kernels as the tolerance mode
// (yt_gpubuild / yt_io / yt_sceneio / yt_multi / yt_order do not need the context's inside.)
#pragma once

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
#include "yt_stream_launch.h"

using namespace yt;

// the calling thread's last error text (ythip_last_error(nullptr)); defined in ythip.hip
std::string& ythip_thread_error();

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

// The bvh part of a DScene: the resident reference tree's, or the own tree's (fastmath = 2: the kernels get a copy of
// `ds` with these fields swapped in at launch time, so lights / materials / cameras updated since are seen by both).
struct BvhView {
  const float4*     pairs = nullptr, *wide = nullptr, *leafdata = nullptr;
  const int*        tlas_prims = nullptr;
  const DInstanceT *tinst = nullptr, *tinst_leaf = nullptr;
  int               tlas_ref = 0x7ffffffe;  // REF_NONE
  vec3f             tlas_bmin = {0, 0, 0}, tlas_bmax = {0, 0, 0};
  const uint4*      own = nullptr;
  static BvhView    of(const DScene& d) {
    BvhView v;
    v.pairs = d.pairs, v.wide = d.wide, v.leafdata = d.leafdata, v.tlas_prims = d.tlas_prims;
    v.tinst = d.tinst, v.tinst_leaf = d.tinst_leaf, v.tlas_ref = d.tlas_ref, v.tlas_bmin = d.tlas_bmin, v.tlas_bmax = d.tlas_bmax;
    v.own = d.own;
    return v;
  }
  void apply(DScene& d) const {
    d.pairs = pairs, d.wide = wide, d.leafdata = leafdata, d.tlas_prims = tlas_prims;
    d.tinst = tinst, d.tinst_leaf = tinst_leaf, d.tlas_ref = tlas_ref, d.tlas_bmin = tlas_bmin, d.tlas_bmax = tlas_bmax;
    d.own = own;
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
  int                         pool_mode   = 0;           // ... taken for this ythip_params::fastmath (another mode: measured again)
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
  bool    wide_stack_ok  = true;  // the wide walk serves the resident trees: its worst-case stack depth fits the 128 entries, and nodes / primitives fit the 26-bit refs of its records (bake_bvh)
  bool    bake_plain_quads = false;  // bake_bvh leaves the quad records in the plain layout (build_own_bvh: k_own_compress reads it)
  bool    use_wide() const;
  ythip_build_info               build_info             = {};
  int64_t                        num_pairs = 0, num_leaf4 = 0;
  ythost::flat_lights h_lights;
  // the own tree (ythip_build_own_bvh, ythip_params::fastmath = 2; yt_own.h): a second set of traversal records + the
  // compressed nodes; dropped whenever the scene's geometry or the reference tree changes
  BvhView                        own;
  std::vector<void*>             own_allocs;
  std::vector<ytgpu::DeviceTree> own_trees;
  bool                           have_own = false, own_stack_ok = false;
  int64_t                        own_nodes = 0, own_leaf4 = 0;
  ythip_build_info               own_info = {};

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
  // the streaming scheduler (yt_stream.h; ythip_set_scheduler): generations of extend / shade launches over SoA path state in HBM
  int                scheduler        = 2;        // 0 the fused persistent kernel (k_trace), 1 streaming generations, 2 (default) measured choice between the two
  // scheduler 2: as for the pixel pool — once the fused path has settled (tile costs known, pool decided), one batch is timed fused,
  // the next two run streamed (the second timed), and whichever took less time per sample serves this state / sampler / mode from then on (same bytes either way)
  bool               sync_call        = false;    // the batch being enqueued is ythip_trace_samples' (the caller waits for it): the measured choice may stream it
  int                sched_tune       = 0;        // 0 time a fused batch next, 1 a streamed batch next (untimed: buffers, first launches), 2 time a streamed batch next, 3 waiting for both, 4 decided
  bool               sched_on         = false;    // the decision: streamed
  long long          sched_key        = -1;       // what it was taken for (sampler, mode, bounces, batch)
  hipEvent_t         sched_ev[4]      = {nullptr, nullptr, nullptr, nullptr};  // fused begin / end, streamed begin / end
  double             sched_samples[2] = {0, 0};
  float              sched_ms[2]      = {0, 0};
  DStream            ss               = {};       // its arrays live in state_allocs (they go with the state)
  int                stream_slots     = 0;        // the slot count they were sized for (0: none)
  int                stream_cell_bits = 3, stream_order = 1, stream_phased = -1, stream_min_batch = 4;  // YTHIP_STREAM_CELLS / _ORDER / _PHASED / _MIN_BATCH (cell major on 8^3 cells: = octant major on 16^3 on the closed boxes, +2 ... +4 % on the hair, the instances and the own tree)
  int*               stream_counts_host = nullptr;  // pinned: {next queue length, generations run} per group
  int                stream_groups = 2;           // chains of generations side by side (YTHIP_STREAM_GROUPS; 2 measured best)
  int                stream_log_gen = -1;         // profiling: the generation whose per-ray walk lengths ks_extend logs (ythip_get_stream_walk_steps)
  int                stream_chunk_shift = 4;      // the groups' tiles are dealt round-robin in chunks of 2^this many tiles (YTHIP_STREAM_CHUNK; 0: tile by tile, 30: contiguous runs — round 6's first form: one chain got the sky, the other the scene)
  int                stream_finish = 250;         // a group leaves the generations for ks_finish once its queue is this many thousandths of its path slots (YTHIP_STREAM_FINISH; 0: never)
  int                stream_min_slots = 262144;   // a group holds at least a quarter of this many path slots (a chain of generations wants a few thousand wavefronts per launch; YTHIP_STREAM_MIN_SLOTS: tests)
  int                stream_bins_cap = 0;         // bins the hist / offs arrays hold per group
  hipStream_t        stream_side[YT_STREAM_MAX_GROUPS] = {};  // the further groups' streams ([0] unused: group 0 runs on `stream`)
  hipEvent_t         stream_ev[YT_STREAM_MAX_GROUPS] = {};
  bool               stream_cancelled = false;    // the last streamed batch was cut short by the caller's stop flag
  bool               last_launch_stream = false;  // the last batch ran on the streaming scheduler
  ythip_stream_info  stream_info      = {};
  bool               last_launch_fast = false;  // the last k_trace launch ran the tolerance-mode kernels (yt_fast.hip)
  int                last_launch_mode = 0;      // ... which mode it ran: 0 bit-exact, 1 tolerance, 2 own tree (yt_owntree.hip)
};

// yt_fast.hip: the tolerance-mode kernels (same source, -DYT_FAST, own namespace); 0 = launched, 1 = no such kernel
extern "C" int ythip_fast_launch(void* stream, int blocks, const void* ds, const void* st, const void* kp, int lp, int cls);
// yt_owntree.hip: the same kernels once more over the own tree (-DYT_FAST -DYT_OWN_TREE, own namespace; yt_own.h)
extern "C" int ythip_own_launch(void* stream, int blocks, const void* ds, const void* st, const void* kp, int lp, int cls);
// the two units' builds of the streaming scheduler's launches (yt_stream_unit.h); `launch`: a ytl::StreamLaunch
extern "C" void ythip_fast_stream_begin(const void* launch);
extern "C" void ythip_fast_stream_generation(const void* launch);
extern "C" void ythip_fast_stream_finish(const void* launch);
extern "C" void ythip_own_stream_begin(const void* launch);
extern "C" void ythip_own_stream_generation(const void* launch);
extern "C" void ythip_own_stream_finish(const void* launch);
extern "C" int ythip_own_intersect(void* stream, const void* ds, const void* rays, const int* instances, long long n, void* hits);

inline void drop_staging_views(ythip_ctx* ctx) {
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
inline void free_staging(ythip_ctx* ctx) {
  drop_staging_views(ctx);
  for (auto p : ctx->staging_allocs)
    if (p) (void)hipHostFree(p);
  ctx->staging_allocs.clear();
  ctx->staging_caps.clear();
}

// The wide walk halves a ray's chain of dependent fetches and costs a little more
// arithmetic per level.  It pays when the waves have the machine to themselves
// (small slices: one GPU of eight, previews) and on large trees; on scenes made of
// tiny trees the 4-slot records are mostly empty.  Measured in docs/HISTORY.md.
inline bool ythip_ctx::use_wide() const {
  if (!wide_stack_ok) return false;  // trees too deep for the wide walk's pushes (bake_bvh): the binary walk
  if (traversal_mode != 2) return traversal_mode == 1;
  return largest_tree >= 64;
}


inline int fail(ythip_ctx* ctx, int code, const char* fmt, ...) {
  char    buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  ythip_thread_error() = buf;
  return code;
}

#define HIPCHECK(ctx, call)                                                                        \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(ctx, YTHIP_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
          __LINE__);                                                                               \
  } while (0)

inline void free_all(std::vector<void*>& v) {
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

inline int grid_for(long long n) { return (int)((n + YT_BLOCK - 1) / YT_BLOCK); }

inline KParams to_kparams(const ythip_ctx* ctx, const ythip_params* p) {
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

// ---- functions of one unit that another calls ---------------------------------------------------------------------
// yt_bake.hip
int  bake_bvh(ythip_ctx* ctx);
void free_device_trees(ythip_ctx* ctx);
int  ensure_host_bvh(ythip_ctx* ctx);
int  build_bvh_mixed(ythip_ctx* ctx, const ythip_scene& sc, bool highquality, bool use_device);
int  build_own_bvh(ythip_ctx* ctx, const ythip_scene& sc);
void drop_own_bvh(ythip_ctx* ctx);
