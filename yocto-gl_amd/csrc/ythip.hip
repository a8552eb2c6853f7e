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
#define YT_MISC_KERNELS 1  // k_tonemap / k_guide_image / k_camera_rays of yt_kernels.h are compiled in this unit only
#include "yt_ctx.h"
#include "yt_denoise.h"  // (plain kernels: this unit only)
#include "yt_launch.h"
#include "yt_order.h"

namespace {
thread_local std::string g_error;
}  // namespace
std::string& ythip_thread_error() { return g_error; }

namespace {

bool stop_raised(ythip_ctx* ctx) {  // ythip_cancel (any thread) has asked this batch to stop
  return ctx->stop_host && __atomic_load_n(ctx->stop_host, __ATOMIC_ACQUIRE) == ctx->stop_gen.load();
}
void raise_stop_word(ythip_ctx* ctx) {  // (what ythip_cancel does: the batch's number into the pinned word the kernels relay)
  __atomic_store_n(ctx->stop_host, ctx->stop_gen.load(), __ATOMIC_RELEASE);
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
// lp: LP_NONE / LP_DEFER for path & pathtest (area lights absent / present);
// pathdirect & pathmis always trace inline; the rest never need a light pdf.
// fast: ythip_params::fastmath — the tolerance-mode kernels of yt_fast.hip where they exist (wide walk, real samplers).
// The kernels themselves are compiled in the yt_trace_*.hip units (yt_launch.h), by sampler family.
int launch_trace_any(ythip_ctx* ctx, const KParams& kp, int lp, bool count, int mode = 0) {
  // the scene class of the default sampler (yt_kernels.h: step_path's CLS)
  const bool classed = kp.sampler == YTHIP_SAMPLER_PATH || kp.sampler == YTHIP_SAMPLER_PATHDIRECT || kp.sampler == YTHIP_SAMPLER_PATHMIS;
  const int  cls     = classed && ctx->specialize ? (ctx->all_matte ? 1 : ctx->no_textures ? 2 : ctx->opaque_textured ? 3 : 0) : 0;
  // mode 2 (ythip_params::fastmath = 2): the own-tree kernels of yt_owntree.hip over the tree ythip_build_own_bvh made.
  // Asked for without such a tree it fails loudly; the debug views (diagram / falsecolor) have no such kernel and render exact.
  if (mode == 2 && !count) {
    if (!ctx->have_own) return fail(ctx, YTHIP_ERR_STATE, "fastmath = 2 needs the own tree: call ythip_build_own_bvh after the bvh is resident");
    if (!ctx->own_stack_ok) return fail(ctx, YTHIP_ERR_INVALID, "own tree too deep for the traversal stack");
    DScene d = ctx->ds;  // (lights / materials / cameras as resident NOW; only the bvh part is the own tree's)
    ctx->own.apply(d);
    if (ythip_own_launch(ctx->stream, ctx->launch_blocks(), &d, &ctx->st, &kp, lp, cls) == 0) {
      ctx->last_launch_fast = true, ctx->last_launch_mode = 2;
      return YTHIP_OK;
    }
  }
  // mode 1: the tolerance-mode unit has the wide-walk kernels only: they serve every tree the wide walk's stack bound
  // admits, whatever use_wide() would prefer (tiny trees: a matter of speed, not of results).  A context forced to the
  // binary walk (ythip_set_traversal 0) or a tree too deep for the wide walk's stack renders with the exact kernels —
  // ythip_last_launch_fastmath says which ran.
  if (mode == 1 && !count && ctx->wide_stack_ok && ctx->traversal_mode != 0) {
    if (ythip_fast_launch(ctx->stream, ctx->launch_blocks(), &ctx->ds, &ctx->st, &kp, lp, cls) == 0) {
      ctx->last_launch_fast = true, ctx->last_launch_mode = 1;
      return YTHIP_OK;
    }
  }
  ctx->last_launch_fast = false, ctx->last_launch_mode = 0;
  ytl::Launch l = {ctx->stream, ctx->launch_blocks(), &ctx->ds, &ctx->st, &kp, count, ctx->use_wide(), lp, cls};
  int rc = 1;
  switch (kp.sampler) {
    case YTHIP_SAMPLER_PATH:
    case YTHIP_SAMPLER_PATHTEST: rc = ytl::launch_path(l); break;
    case YTHIP_SAMPLER_PATHDIRECT:
    case YTHIP_SAMPLER_PATHMIS:
      rc = ytl::launch_nee_class(l);  // (the scene classes 1-3 of the wide walk: yt_trace_nee_cls.hip)
      if (rc) rc = ytl::launch_nee(l);
      break;
    case YTHIP_SAMPLER_NAIVE:
    case YTHIP_SAMPLER_EYELIGHT:
    case YTHIP_SAMPLER_DIAGRAM:
    case YTHIP_SAMPLER_FURNACE:
    case YTHIP_SAMPLER_FALSECOLOR: rc = ytl::launch_misc(l); break;
    default: break;
  }
  if (rc) return fail(ctx, YTHIP_ERR_SAMPLER, "sampler unknown");
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

// The streaming scheduler's batch (yt_stream.h): ks_init, then generations of (scatter, extend, shade, scan) until a
// read-back of the next queue's length says zero.  The first `batch` generations are enqueued blind (a sample is at least
// one generation), the rest in chunks; a chunk's surplus generations return at once on the device.  Blocks until the batch
// is done (the queue length has to come home), watching the caller's stop flag meanwhile.
//   * groups (YTHIP_STREAM_GROUPS, default 2): the path slots in runs, each a chain of generations of its own on its own
//     stream — one group's shade / sort launches fill the machine while another's extend launch drains (2 groups: +8 ... +25 %
//     over one; 4 and more lose again: profiles/r06_stream_ab_groups.txt).
//   * mode (ythip_params::fastmath): 0 = the bit-exact kernels (yt_stream.hip), 1 = the tolerance mode's (yt_fast.hip), 2 = the own
//     tree's (yt_owntree.hip; `ds` is then the scene with the own tree's bvh fields swapped in).  The same schedule of the same
//     per-pixel operations as the mode's fused kernel: a streamed batch equals the fused batch of its mode byte for byte.
int enqueue_stream(ythip_ctx* ctx, const ythip_params* params, const KParams& kp, int lp, int cls, const volatile int32_t* stop, int mode,
    const DScene& ds) {
  auto begin      = [&](const ytl::StreamLaunch& l) { mode == 2 ? ythip_own_stream_begin(&l) : mode == 1 ? ythip_fast_stream_begin(&l) : ytl::stream_begin(l); };
  auto generation = [&](const ytl::StreamLaunch& l) {
    mode == 2 ? ythip_own_stream_generation(&l) : mode == 1 ? ythip_fast_stream_generation(&l) : ytl::stream_generation(l);
  };
  auto finish = [&](const ytl::StreamLaunch& l) { mode == 2 ? ythip_own_stream_finish(&l) : mode == 1 ? ythip_fast_stream_finish(&l) : ytl::stream_finish(l); };
  auto& st = ctx->st;
  auto& S  = ctx->ss;
  int   rc;
  constexpr int MAX_GROUPS = YT_STREAM_MAX_GROUPS;
  if (ctx->stream_slots != st.nslots) {
    const size_t ns = (size_t)st.nslots;
    S               = DStream{};
    S.prim_shift    = 8;  // camera rays: bins of 4 neighbouring tiles
    while (((st.nslots >> S.prim_shift) > 16384)) S.prim_shift++;
    S.nprim_bins = std::max(1, (st.nslots + (1 << S.prim_shift) - 1) >> S.prim_shift);
    const size_t nb = (((size_t)(8 << (3 * 5)) + (size_t)S.nprim_bins + 4095) / 4096 + 1) * 4096;  // (room for the finest cell grid, padded to ks_scan's tiles)
#define AL(field, count) \
  if ((rc = dalloc(ctx, ctx->state_allocs, &S.field, (size_t)(count)))) return rc;
    AL(ray_a, ns) AL(ray_b, ns) AL(wgt, ns) AL(rad, ns) AL(rng, ns) AL(hit_a, ns) AL(hit_e, ns) AL(key, ns) AL(rank, ns)
    AL(queue, ns) AL(hist, MAX_GROUPS * nb) AL(offs, MAX_GROUPS * nb) AL(counts, MAX_GROUPS * 16) AL(stats, MAX_GROUPS * 8 * 64)
    AL(gen_rays, MAX_GROUPS * YT_STREAM_GEN_LOG) AL(step_log, ns)
#undef AL
    HIPCHECK(ctx, hipMemsetAsync(S.hist, 0, MAX_GROUPS * nb * sizeof(unsigned), ctx->stream));
    ctx->stream_bins_cap = (int)nb;
    ctx->stream_slots    = st.nslots;
  }
  if (!ctx->stream_counts_host) HIPCHECK(ctx, hipHostMalloc((void**)&ctx->stream_counts_host, MAX_GROUPS * 16 * sizeof(int), hipHostMallocDefault));
  S.order     = ctx->stream_order;
  S.cell_bits = std::min(std::max(ctx->stream_cell_bits, 1), 5);
  S.nbins     = 8 << (3 * S.cell_bits);
  // the cell grid of the sort keys: the scene's root box
  const vec3f lo = ds.tlas_bmin, hi = ds.tlas_bmax;
  const float cells = (float)(1 << S.cell_bits);
  auto        scale = [&](float a, float b) { return (b > a && std::isfinite(b - a)) ? cells / (b - a) : 0.0f; };
  S.cell_lo    = lo;
  S.cell_scale = {scale(lo.x, hi.x), scale(lo.y, hi.y), scale(lo.z, hi.z)};
  const int pslots = st.nslots;  // every pixel in flight (fewer slots fed by a pixel queue: measured, -17 ... -40 %, docs/HISTORY.md)
  int groups = std::min(std::max(ctx->stream_groups, 1), MAX_GROUPS);
  while (groups > 1 && pslots / groups < ctx->stream_min_slots / 4) groups--;  // (a chain of generations wants a few thousand wavefronts per launch)
  st.tile_perm = nullptr, st.tile_cost = nullptr, st.pool_next = nullptr, st.pool_total = 0;
  const bool prof = (ctx->prof_mode & 1) != 0;
  if (prof) HIPCHECK(ctx, hipMemsetAsync(S.stats, 0, MAX_GROUPS * 8 * 64 * sizeof(unsigned long long), ctx->stream));
  HIPCHECK(ctx, hipMemsetAsync(S.counts, 0, MAX_GROUPS * 16 * sizeof(int), ctx->stream));
  // (the majority-phase walk — the fused kernel's choice on matte scenes with area lights, where a wavefront's lanes want different step
  //  kinds at any moment — is OFF by default here: ks_extend's wavefronts are sorted rays that mostly want the same; cfg2b 62.2 -> 53.6 ms,
  //  the 9M box 19.2 -> 18.3 ms, profiles/r06_stream_ab_eviction.txt.
  //  The own tree's walk keeps the fused kernel's rule: cfg2b 46.3 against 46.9 ms, the 9M box 14.1 against 14.8.)
  const bool phased = ctx->stream_phased > 0 || (ctx->stream_phased < 0 && mode == 2 && cls == 1 && lp == LP_DEFER);
  DStream           G[MAX_GROUPS];
  ytl::StreamLaunch L[MAX_GROUPS];
  hipStream_t       streams[MAX_GROUPS];
  streams[0] = ctx->stream;
  for (int g = 1; g < groups; g++) {
    if (!ctx->stream_side[g]) HIPCHECK(ctx, hipStreamCreateWithFlags(&ctx->stream_side[g], hipStreamNonBlocking));
    if (!ctx->stream_ev[g]) HIPCHECK(ctx, hipEventCreateWithFlags(&ctx->stream_ev[g], hipEventDisableTiming));
    streams[g] = ctx->stream_side[g];
  }
  if (groups > 1 && !ctx->stream_ev[0]) HIPCHECK(ctx, hipEventCreateWithFlags(&ctx->stream_ev[0], hipEventDisableTiming));
  // the path slots in `groups` sets of whole wavefronts (tiles), interleaved tile by tile (yt_stream.h: stream_slot)
  const int tiles = pslots / 64;
  int gshift = std::min(std::max(ctx->stream_chunk_shift, 0), 30);
  while (gshift > 0 && (long long)tiles < ((long long)groups << gshift)) gshift--;  // (every group gets tiles)
  const int chunk = 1 << gshift;
  for (int g = 0, off = 0; g < groups; g++) {
    G[g]          = S;
    G[g].gstride  = groups, G[g].goff = g, G[g].gshift = gshift;
    G[g].slot0    = off;
    const long long cycle = (long long)chunk * groups, rem = tiles % cycle;  // whole rounds of the deal + what the last one leaves this group
    G[g].nslots   = (int)((tiles / cycle) * chunk + std::min<long long>(std::max<long long>(rem - (long long)g * chunk, 0), chunk)) * 64;
    off += G[g].nslots;
    G[g].hist     = S.hist + (size_t)g * ctx->stream_bins_cap;
    G[g].offs     = S.offs + (size_t)g * ctx->stream_bins_cap;
    G[g].queue    = S.queue + G[g].slot0;
    G[g].counts   = S.counts + 16 * g;
    G[g].stats    = prof ? S.stats + 8 * 64 * g : nullptr;
    G[g].gen_rays = prof ? S.gen_rays + YT_STREAM_GEN_LOG * g : nullptr;
    G[g].step_log = prof && ctx->stream_log_gen >= 0 ? S.step_log + G[g].slot0 : nullptr;
    G[g].log_gen  = ctx->stream_log_gen;
    L[g]          = {streams[g], &ds, &ctx->st, &kp, &G[g], lp, cls, phased};
  }
  ctx->stream_cancelled = false;
  int     launched    = 0;
  int64_t finish_rays = 0;
  {
    EvScope ev(ctx, 0);
    if (groups > 1) {  // the side streams start behind everything the main stream has been given so far
      HIPCHECK(ctx, hipEventRecord(ctx->stream_ev[0], ctx->stream));
      for (int g = 1; g < groups; g++) HIPCHECK(ctx, hipStreamWaitEvent(streams[g], ctx->stream_ev[0], 0));
    }
    for (int g = 0; g < groups; g++) begin(L[g]);
    // A group whose queue has shrunk to `finish` thousandths of its path slots leaves the generations: ks_finish carries what is
    // queued to the end of the batch in one launch (yt_stream.h).  1000: from the first ray on (tests: the whole batch in ks_finish).
    const int permille = std::min(std::max(ctx->stream_finish, 0), 1000);
    bool      fin[MAX_GROUPS] = {}, idle[MAX_GROUPS] = {};
    if (permille >= 1000)
      for (int g = 0; g < groups; g++) finish(L[g]), fin[g] = true, finish_rays += G[g].nslots;
    // the first `batch` generations need no look at the queue (a sample is at least one generation) — but what is enqueued cannot be
    // taken back: a cancelled batch still pays for every launch queued behind it (2.5 us each, returning at once).  64 at a time keeps
    // a cancel under 2 ms of empty launches whatever the batch (a 4096-sample batch enqueued blind: 90 ms).
    int chunk = std::min(std::max(1, params->batch), 64), blind = std::max(1, params->batch);
    if (!ctx->done_event) HIPCHECK(ctx, hipEventCreateWithFlags(&ctx->done_event, hipEventDisableTiming));
    while (true) {
      bool work = false;
      for (int g = 0; g < groups; g++) work = work || !(fin[g] || idle[g]);
      if (work) {
        for (int k = 0; k < chunk; k++)
          for (int g = 0; g < groups; g++)
            if (!(fin[g] || idle[g])) generation(L[g]);
        launched += chunk;
      }
      for (int g = 1; g < groups; g++) {  // the main stream (and with it the read-back, and whatever the caller enqueues next) waits for the side streams
        HIPCHECK(ctx, hipEventRecord(ctx->stream_ev[g], streams[g]));
        HIPCHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->stream_ev[g], 0));
      }
      HIPCHECK(ctx, hipMemcpyAsync(ctx->stream_counts_host, S.counts, MAX_GROUPS * 16 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
      HIPCHECK(ctx, hipEventRecord(ctx->done_event, ctx->stream));
      // (and the side streams' next chunk behind the read-back: the chains stay in step chunk by chunk)
      for (int g = 1; g < groups; g++) HIPCHECK(ctx, hipStreamWaitEvent(streams[g], ctx->done_event, 0));
      if (stop) {
        while (hipEventQuery(ctx->done_event) == hipErrorNotReady) {
          if (*stop && !ctx->stream_cancelled) {
            raise_stop_word(ctx);
            ctx->stream_cancelled = true;
          }
          std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
      }
      HIPCHECK(ctx, hipEventSynchronize(ctx->done_event));
      if (!work) break;  // (this round only waited for the ks_finish launches of the last one)
      bool any = false, finishing = false;
      for (int g = 0; g < groups; g++) {
        if (fin[g] || idle[g]) continue;
        const int n = ctx->stream_counts_host[16 * g];
        if (n == 0) idle[g] = true;
        else if ((int64_t)n * 1000 <= (int64_t)permille * G[g].nslots) finish(L[g]), fin[g] = finishing = true, finish_rays += n;
        else any = true;
      }
      if (!any && !finishing) break;
      // inside the first `batch` generations: 64 at a time; behind them what is left are the live paths' remaining bounces — a few
      // generations at a time (each surplus one costs four empty launches)
      blind -= chunk;
      chunk = blind > 0 ? std::min(blind, 64) : 16;
      if (ctx->stream_cancelled || stop_raised(ctx)) chunk = 4;
    }
  }
  HIPCHECK(ctx, hipGetLastError());
  ctx->stream_info             = {};
  ctx->stream_info.ran         = 1;
  for (int g = 0; g < groups; g++) ctx->stream_info.generations = std::max(ctx->stream_info.generations, ctx->stream_counts_host[16 * g + 1]);
  ctx->stream_info.launched    = launched;
  ctx->stream_info.bins        = S.nbins + S.nprim_bins;
  ctx->stream_info.groups      = groups;
  ctx->stream_info.path_slots  = pslots;
  ctx->stream_info.finish_rays = finish_rays;
  if (prof) {
    std::vector<unsigned long long> h(MAX_GROUPS * 8 * 64);
    HIPCHECK(ctx, hipMemcpy(h.data(), S.stats, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int b = 0; b < MAX_GROUPS * 64; b++) {
      ctx->stream_info.lane_steps += (int64_t)h[8 * b + 0];
      ctx->stream_info.wave_steps += (int64_t)h[8 * b + 1];
      ctx->stream_info.rays += (int64_t)h[8 * b + 3];
    }
  }
  return YTHIP_OK;
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
  if (params->fastmath < 0 || params->fastmath > 2)  // (any other value used to render exact without a word)
    return fail(ctx, YTHIP_ERR_INVALID, "fastmath must be 0 (bit-exact), 1 (tolerance) or 2 (own tree), not %d", params->fastmath);
  ctx->have_denoised = false;
  if (only_pix < 0 && ctx->samples >= params->samples) return YTHIP_OK;  // yocto_trace.cpp:1598
  if (stop && *stop) return fail(ctx, YTHIP_ERR_CANCELLED, "cancelled");

  auto kp          = to_kparams(ctx, params);
  bool count       = (ctx->prof_mode & 2) != 0;
  ctx->st.counters = count ? ctx->d_counters : nullptr;
#ifdef YT_TIMING
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
  const int stream_lp = params->sampler == YTHIP_SAMPLER_PATHDIRECT ? LP_DEFER : lp;  // (the NEE samplers always run their walks deferred)

  // the streaming scheduler (ythip_set_scheduler 1): `path`, scenes the mode's wide walk serves, real batches.  In the mode the
  // caller asked for (round 6: the tolerance and own-tree units carry their own build of the kernels); a mode whose fused kernel
  // would not run either (no own tree: an error below; a tree the tolerance unit's walk cannot serve) is left to the fused path.
  ctx->last_launch_stream = false;
  ctx->stream_info.ran    = 0;
  // pixel pool: only where there are more tiles than resident workgroups, and once the tile costs of a plain launch
  // order the queue (the first batch / the probe launch runs plain and records them)
  // (bounces <= 0: k_trace finishes such a batch in its prologue, tile by tile, without ever reaching the queue)
  const bool pool_ok = ctx->pixel_pool && only_pix < 0 && !count && ctx->st.nblocks > ctx->pool_blocks && params->bounces > 0 &&
                       (ctx->have_tile_costs || !ctx->d_tile_cost || ctx->pixel_pool >= 2);
  int sched_timed = -1;  // scheduler 2: 0 = this batch is the timed fused one, 1 = the timed streamed one
  // (the measured choice only ever streams a batch whose caller waits for it anyway — ythip_trace_samples —: ythip_trace_samples_async keeps
  //  returning at once under the default; mode 1 streams whatever it serves, and its async call returns when the batch is done)
  if (ctx->scheduler != 0 && (ctx->scheduler == 1 || ctx->sync_call) && only_pix < 0 && !count && params->batch >= ctx->stream_min_batch) {
    const int  mode = params->fastmath;
    const bool served = mode == 0 ? ctx->use_wide() : mode == 1 ? (ctx->wide_stack_ok && ctx->traversal_mode != 0) : (ctx->have_own && ctx->own_stack_ok);
    // (`naive` and `pathtest` have the general-class kernels only, as their fused kernels)
    const bool plain = params->sampler == YTHIP_SAMPLER_NAIVE || params->sampler == YTHIP_SAMPLER_PATHTEST;
    const int  cls  = (ctx->specialize && !plain) ? (ctx->all_matte ? 1 : ctx->no_textures ? 2 : ctx->opaque_textured ? 3 : 0) : 0;
    ytl::StreamLaunch probe = {ctx->stream, &ctx->ds, &ctx->st, &kp, &ctx->ss, stream_lp, cls, false};
    bool stream = served && ytl::stream_supported(probe);
    if (stream && ctx->scheduler == 2) {  // the measured choice
      const long long key = (long long)params->sampler | ((long long)mode << 8) | ((long long)(params->bounces & 0xffff) << 16) | ((long long)params->batch << 32);
      if (key != ctx->sched_key) ctx->sched_key = key, ctx->sched_tune = 0, ctx->sched_on = false;
      if (ctx->sched_tune == 3 && hipEventQuery(ctx->sched_ev[1]) == hipSuccess && hipEventQuery(ctx->sched_ev[3]) == hipSuccess) {
        if (hipEventElapsedTime(&ctx->sched_ms[0], ctx->sched_ev[0], ctx->sched_ev[1]) == hipSuccess &&
            hipEventElapsedTime(&ctx->sched_ms[1], ctx->sched_ev[2], ctx->sched_ev[3]) == hipSuccess)
          ctx->sched_on = ctx->sched_ms[1] / ctx->sched_samples[1] < 0.97 * ctx->sched_ms[0] / ctx->sched_samples[0];
        ctx->sched_tune = 4;
      }
      if (ctx->sched_tune == 4) stream = ctx->sched_on;
      else if (params->batch < 8 || ctx->sched_tune == 3) stream = false;  // (too short a batch for its time to mean something: fused, undecided)
      else {
        // the fused path first has to be the one the streamed batch competes with: tile costs known, the pixel pool decided
        const bool pool_pending = ctx->pixel_pool == 1 && ctx->pool_tune < 3 && ctx->st.nblocks > ctx->pool_blocks && params->bounces > 0;
        const bool settled      = (ctx->have_tile_costs || !ctx->d_tile_cost) && !pool_pending;
        for (auto& e : ctx->sched_ev)
          if (!e) HIPCHECK(ctx, hipEventCreate(&e));
        if (ctx->sched_tune == 0) stream = false, sched_timed = settled ? 0 : -1;
        else if (ctx->sched_tune == 1) ctx->sched_tune = 2;  // this batch streamed, untimed: the scheduler's buffers, streams and first launches
        else sched_timed = 1;                                // (tune 2: this batch streamed, timed)
      }
    }
    if (stream) {
      ctx->st.tile_perm = nullptr, ctx->st.tile_cost = nullptr, ctx->st.pool_next = nullptr, ctx->st.pool_total = 0;
      DScene d = ctx->ds;  // (mode 2: only the bvh part is the own tree's — launch_trace_any)
      if (mode == 2) ctx->own.apply(d);
      if (sched_timed == 1) HIPCHECK(ctx, hipEventRecord(ctx->sched_ev[2], ctx->stream));
      int rc = enqueue_stream(ctx, params, kp, stream_lp, cls, stop, mode, d);
      if (rc) return rc;
      if (sched_timed == 1) {
        HIPCHECK(ctx, hipEventRecord(ctx->sched_ev[3], ctx->stream));
        ctx->sched_samples[1] = params->batch, ctx->sched_tune = 3;
      }
      ctx->last_launch_fast = mode != 0, ctx->last_launch_mode = mode, ctx->last_launch_stream = true;
      ctx->samples += params->batch;
      return YTHIP_OK;
    }
  }
  // one launch renders the whole batch: every workgroup loops over its tile
  // until its pixels have taken `batch` samples (k_trace)
  ctx->st.tile_perm = nullptr, ctx->st.tile_cost = nullptr;
  ctx->st.pool_next = nullptr, ctx->st.pool_total = 0;
  bool pool = false;
  int  timed = -1;  // 0: this launch is the timed plain batch, 1: the timed pool batch
  if (params->fastmath != ctx->pool_mode) {  // (the plain / pool choice was measured on another kernel family)
    ctx->pool_mode = params->fastmath;
    if (ctx->pixel_pool < 2) ctx->pool_tune = 0, ctx->pool_on = false;
  }
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
    if (sched_timed == 0) HIPCHECK(ctx, hipEventRecord(ctx->sched_ev[0], ctx->stream));
    int rc = launch_trace_any(ctx, kp, lp, count, params->fastmath);
    if (rc) return rc;
    if (sched_timed == 0) {
      HIPCHECK(ctx, hipEventRecord(ctx->sched_ev[1], ctx->stream));
      ctx->sched_samples[0] = params->batch, ctx->sched_tune = 1;
    }
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

// the 256 values srgb_to_rgb can take on a byte texel, by the function the kernels would call (yt_scene.h: decode_texel)
__global__ void __launch_bounds__(256) k_srgb_lut(float* lut) { lut[threadIdx.x] = srgb_to_rgb(div_((float)threadIdx.x, 255.0f)); }

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
  if (const char* e = std::getenv("YTHIP_SCHEDULER")) ctx->scheduler = std::min(std::max(std::atoi(e), 0), 2);
  if (const char* e = std::getenv("YTHIP_TRAVERSAL")) ctx->traversal_mode = std::min(std::max(std::atoi(e), 0), 2);  // (ythip_set_traversal)
  if (const char* e = std::getenv("YTHIP_STREAM_CELLS")) ctx->stream_cell_bits = std::atoi(e);
  if (const char* e = std::getenv("YTHIP_STREAM_ORDER")) ctx->stream_order = std::atoi(e);
  if (const char* e = std::getenv("YTHIP_STREAM_PHASED")) ctx->stream_phased = std::atoi(e);
  if (const char* e = std::getenv("YTHIP_STREAM_MIN_BATCH")) ctx->stream_min_batch = std::atoi(e);
  if (const char* e = std::getenv("YTHIP_STREAM_GROUPS")) ctx->stream_groups = std::atoi(e);
  if (const char* e = std::getenv("YTHIP_STREAM_FINISH")) ctx->stream_finish = std::atoi(e);
  if (const char* e = std::getenv("YTHIP_STREAM_CHUNK")) ctx->stream_chunk_shift = std::atoi(e);
  if (const char* e = std::getenv("YTHIP_STREAM_MIN_SLOTS")) ctx->stream_min_slots = std::max(128, std::atoi(e));
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

int ythip_abi_version(void) { return YTHIP_ABI_VERSION; }
int ythip_params_size(void) { return (int)sizeof(ythip_params); }

void ythip_destroy(ythip_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  free_all(ctx->scene_allocs);
  free_all(ctx->bvh_allocs);
  drop_own_bvh(ctx);
  free_all(ctx->light_allocs);
  free_all(ctx->state_allocs);
  free_device_trees(ctx);
  for (auto& ev : ctx->ev_pool) {
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  if (ctx->d_counters) (void)hipFree(ctx->d_counters);
  if (ctx->stop_host) (void)hipHostFree(ctx->stop_host);
  if (ctx->stream_counts_host) (void)hipHostFree(ctx->stream_counts_host);
  for (auto q : ctx->stream_side)
    if (q) (void)hipStreamDestroy(q);
  for (auto e : ctx->stream_ev)
    if (e) (void)hipEventDestroy(e);
  if (ctx->d_stop) (void)hipFree(ctx->d_stop);
  if (ctx->d_pool_next) (void)hipFree(ctx->d_pool_next);
  for (auto e : ctx->pool_ev)
    if (e) (void)hipEventDestroy(e);
  for (auto e : ctx->sched_ev)
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
  // (the shared chunk of the small uploads remembers the stream its copies went to: it is closed — its event recorded — while
  //  that stream is still the context's, so that the ring never holds a stream the caller may destroy after switching away)
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  HIPCHECK(ctx, ctx->xfer.close_small());
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
  drop_own_bvh(ctx);  // (geometry / the reference tree changes: the own tree of fastmath = 2 is stale)
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
  {
    float* lut = nullptr;
    if ((rc = dalloc(ctx, ctx->scene_allocs, &lut, 256))) return rc;
    hipLaunchKernelGGL(k_srgb_lut, dim3(1), dim3(256), 0, ctx->stream, lut);
    HIPCHECK(ctx, hipGetLastError());
    ds.srgb_lut = lut;
  }
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
  drop_own_bvh(ctx);
  return build_bvh_mixed(ctx, *sc, highquality != 0, ctx->bvh_builder != 0);
}

// The own tree of ythip_params::fastmath = 2 (yt_own.h): built next to the resident reference tree, which stays what
// every other mode and ythip_intersect_batch use.  Dropped by anything that changes geometry or the reference tree
// (upload_scene, build / upload / update_bvh, update_shape_vertices, update_instance_frames): build it again afterwards.
int ythip_build_own_bvh(ythip_ctx* ctx, const ythip_scene* sc) {
  if (!ctx) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  if (!ctx->have_scene) return fail(ctx, YTHIP_ERR_STATE, "upload_scene first");
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  if (sc) return build_own_bvh(ctx, *sc);
  // no scene given: the geometry as RESIDENT — the context's host copies of shapes, instances, element lists and
  // positions, which follow ythip_update_shape_vertices / ythip_update_instance_frames (the caller's arrays may not)
  ythip_scene view   = {};
  view.shapes        = ctx->h_shapes.data(), view.num_shapes = (int)ctx->h_shapes.size();
  view.instances     = ctx->h_instances.data(), view.num_instances = (int)ctx->h_instances.size();
  view.positions     = ctx->h_positions.data(), view.radius = ctx->h_radius.empty() ? nullptr : ctx->h_radius.data();
  view.points        = ctx->h_points.data(), view.lines = ctx->h_lines.data();
  view.triangles     = ctx->h_triangles.data(), view.quads = ctx->h_quads.data();
  return build_own_bvh(ctx, view);
}
int ythip_own_bvh_info(ythip_ctx* ctx, int64_t* num_nodes, int64_t* num_leaf4, ythip_build_info* info) {
  if (!ctx) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  if (num_nodes) *num_nodes = ctx->have_own ? ctx->own_nodes : 0;
  if (num_leaf4) *num_leaf4 = ctx->have_own ? ctx->own_leaf4 : 0;
  if (info) *info = ctx->have_own ? ctx->own_info : ythip_build_info{};
  return ctx->have_own ? YTHIP_OK : fail(ctx, YTHIP_ERR_STATE, "no own tree resident");
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
  drop_own_bvh(ctx);  // (geometry / the reference tree changes: the own tree of fastmath = 2 is stale)
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
  drop_own_bvh(ctx);  // (geometry / the reference tree changes: the own tree of fastmath = 2 is stale)
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
  drop_own_bvh(ctx);  // (geometry / the reference tree changes: the own tree of fastmath = 2 is stale)
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
  drop_own_bvh(ctx);  // (geometry / the reference tree changes: the own tree of fastmath = 2 is stale)
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
  ctx->stream_slots  = 0;  // (the streaming scheduler's arrays went with the state)
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
  ctx->sched_tune = 0, ctx->sched_on = false;  // (... and so is the scheduler, where it is a measured choice)
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
                     ctx->samples < params->samples &&
                     ctx->pixel_pool < 2;  // (a forced pool launch records no tile costs: every batch would be probed — found in round 5)
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
  ctx->sync_call           = true;
  int       rc             = enqueue_batch(ctx, params, stop);
  ctx->sync_call           = false;
  if (rc) return rc;
  bool cancelled = ctx->last_launch_stream && ctx->stream_cancelled;  // (a streamed batch watched the flag itself)
  if (stop && !cancelled) {
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
  ctx->sched_tune = 0, ctx->sched_on = false;
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

int ythip_set_scheduler(ythip_ctx* ctx, int mode) {
  if (!ctx || mode < 0 || mode > 2) return fail(ctx, YTHIP_ERR_INVALID, "scheduler must be 0 (fused kernel), 1 (streaming) or 2 (measured choice)");
  ctx->scheduler = mode, ctx->sched_tune = 0, ctx->sched_on = false;
  return YTHIP_OK;
}
int ythip_get_scheduler(ythip_ctx* ctx) { return ctx ? ctx->scheduler : 0; }
// May the next batch with these parameters run on the streaming scheduler (its enqueue call then returns only when the batch is done)?
// Conservative: 1 whenever the scheduler might take it — under the measured choice also while the two times are still awaited; 0 for
// what the fused kernel is sure to run (not served, the choice's fused probe, a decision for the fused kernel).
int ythip_may_stream(ythip_ctx* ctx, const ythip_params* params) {
  if (!ctx || !params || ctx->scheduler == 0) return 0;
  if (params->batch < ctx->stream_min_batch || params->bounces <= 0) return 0;
  if (params->sampler != YTHIP_SAMPLER_PATH && params->sampler != YTHIP_SAMPLER_PATHDIRECT && params->sampler != YTHIP_SAMPLER_PATHTEST &&
      params->sampler != YTHIP_SAMPLER_NAIVE)
    return 0;
  if (ctx->scheduler == 2) {
    if (params->batch < 8) return 0;
    const long long key = (long long)params->sampler | ((long long)params->fastmath << 8) | ((long long)(params->bounces & 0xffff) << 16) | ((long long)params->batch << 32);
    if (key == ctx->sched_key && ctx->sched_tune == 4 && !ctx->sched_on) return 0;
    if (key != ctx->sched_key || ctx->sched_tune == 0) return 0;  // (the next batch is a fused one: timed, or waiting for the fused path to settle)
  }
  return 1;
}
int ythip_set_stream_options(ythip_ctx* ctx, int order, int cell_bits, int phased) {
  if (!ctx || order > 2 || cell_bits == 0 || cell_bits > 5 || phased > 1) return fail(ctx, YTHIP_ERR_INVALID, "stream options: order 0..2, cell_bits 1..5, phased 0..1");
  if (order >= 0) ctx->stream_order = order;
  if (cell_bits > 0) ctx->stream_cell_bits = cell_bits;
  if (phased >= 0) ctx->stream_phased = phased;
  return YTHIP_OK;
}
int ythip_set_stream_groups(ythip_ctx* ctx, int groups) {
  if (!ctx || groups < 1 || groups > YT_STREAM_MAX_GROUPS) return fail(ctx, YTHIP_ERR_INVALID, "stream groups: 1..%d", YT_STREAM_MAX_GROUPS);
  ctx->stream_groups = groups;
  return YTHIP_OK;
}
int ythip_get_stream_walk_steps(ythip_ctx* ctx, int generation, int32_t* steps, int32_t capacity) {
  if (!ctx) return YTHIP_ERR_INVALID;
  ctx->stream_log_gen = generation;
  if (!steps) return YTHIP_OK;  // (only chooses the generation the next profiled batch logs)
  if (!ctx->stream_slots || capacity < ctx->stream_slots) return fail(ctx, YTHIP_ERR_STATE, "no streamed batch to report / capacity below the path slots");
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  HIPCHECK(ctx, hipMemcpy(steps, ctx->ss.step_log, (size_t)ctx->stream_slots * sizeof(int), hipMemcpyDeviceToHost));
  return YTHIP_OK;
}
int ythip_set_stream_finish(ythip_ctx* ctx, int permille) {
  if (!ctx || permille < 0 || permille > 1000) return fail(ctx, YTHIP_ERR_INVALID, "stream finish: 0..1000 thousandths of the path slots");
  ctx->stream_finish = permille;
  return YTHIP_OK;
}
int ythip_get_stream_generations(ythip_ctx* ctx, int32_t* rays, int32_t capacity, int32_t* written) {
  if (!ctx || !rays || capacity < 0 || !written) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  *written = 0;
  if (!ctx->stream_slots || !ctx->stream_info.ran) return fail(ctx, YTHIP_ERR_STATE, "no streamed batch to report");
  const int n = std::min({capacity, (int)ctx->stream_info.generations, (int)YT_STREAM_GEN_LOG});
  HIPCHECK(ctx, hipSetDevice(ctx->device));
  if (n > 0) HIPCHECK(ctx, hipMemcpy(rays, ctx->ss.gen_rays, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
  *written = n;
  return YTHIP_OK;
}
int ythip_get_stream_info(ythip_ctx* ctx, ythip_stream_info* info) {
  if (!ctx || !info) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  *info = ctx->stream_info;
  info->choice_state = ctx->scheduler == 2 ? ctx->sched_tune : 0;
  info->choice_streamed = ctx->scheduler == 2 && ctx->sched_tune == 4 && ctx->sched_on;
  if (ctx->scheduler == 2 && ctx->sched_tune == 4 && ctx->sched_samples[0] > 0 && ctx->sched_samples[1] > 0) {
    info->fused_ms_per_sample  = (float)(ctx->sched_ms[0] / ctx->sched_samples[0]);
    info->stream_ms_per_sample = (float)(ctx->sched_ms[1] / ctx->sched_samples[1]);
  }
  return YTHIP_OK;
}

int ythip_last_launch_fastmath(ythip_ctx* ctx) { return ctx ? ctx->last_launch_mode : 0; }

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
    int find_any, ythip_hit* hits, bool own = false) {
  if (!ctx || (n > 0 && (!rays || !hits))) return fail(ctx, YTHIP_ERR_INVALID, "null argument");
  if (!ctx->have_scene || !ctx->have_bvh) return fail(ctx, YTHIP_ERR_STATE, "scene and bvh must be resident");
  if (own && !(ctx->have_own && ctx->own_stack_ok)) return fail(ctx, YTHIP_ERR_STATE, "no own tree resident (ythip_build_own_bvh)");
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
  if (own) {
    DScene d = ctx->ds;
    ctx->own.apply(d);
    ythip_own_intersect(ctx->stream, &d, d_rays, d_inst, (long long)n, d_hits);
  } else if (count)
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
// the same batch through the OWN tree's walk (fastmath = 2's traversal; instances may be null = intersect_scene): a
// measuring entry — how often do its hit records differ from ythip_intersect_batch's, i.e. the reference's?
int ythip_intersect_batch_own(ythip_ctx* ctx, const int32_t* instances, const ythip_ray* rays, int64_t n, ythip_hit* hits) {
  return intersect_impl(ctx, instances, rays, n, 0, hits, true);
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
