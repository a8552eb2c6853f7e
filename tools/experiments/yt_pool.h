// yt_pool.h — k_pool: trace_samples as a persistent wavefront with a POOL of path
// slots and dynamic ray fetch inside the BVH walk.
//
// k_trace (yt_kernels.h) gives every wavefront 64 pixels and lets the wavefront
// run "extend all 64 rays → shade all 64 paths" rounds: a round lasts as long as
// its longest ray.  On incoherent rays (interiors, instanced scenes, hair) the
// lanes' walk lengths differ so much that only 34-52 % of the issued traversal
// lanes do useful work (profiles/r01).  k_pool keeps the SAME per-path arithmetic
// — the walk of yt_bvh.h, the loop bodies of yt_kernels.h — and changes only WHO
// runs WHAT WHEN:
//
//   * a wavefront owns POOL_T (128 / 256) path slots = pixels, not 64: the hot ray
//     (24 B) of every slot sits in LDS, the rest of the path state (weight,
//     radiance, rng, hit record: 80 B) in a per-wavefront SoA block in global
//     memory (L2-resident, touched once per bounce, coalesced);
//   * the walk loop is the OUTER loop.  A lane that finishes its ray does not wait
//     for the wavefront's longest ray: once `refill_min` lanes are idle they hand
//     their hit records over and take the next queued rays (Aila-Laine style
//     dynamic fetch, wave ballots + LDS rings, no atomics);
//   * shading runs in passes of 64 finished paths whenever 64 are pending (or the
//     walk has run dry), always with full lanes; continuing and regenerated rays
//     go back to the ray ring;
//   * pixels come from a global tile counter (one atomic per 64-pixel tile): a
//     wavefront whose pixels have all taken their samples grabs the next tile, so
//     the frame is load-balanced dynamically.
//
// Pixels are independent and a pixel's samples stay strictly sequential (its PCG
// stream and running means live in its slot), so trace_state is bit-identical to
// k_trace's and the reference's whatever the schedule — tested on every scene
// (tests/test_gpu_pool.py).  Restates the same reference code as yt_kernels.h
// (libs/yocto/yocto_trace.cpp:453-596, 1461-1492; yocto_bvh.cpp:460-617).
#pragma once

#include "yt_kernels.h"

namespace yt {

#ifndef YT_POOL_T
#define YT_POOL_T 256
#endif
constexpr int POOL_T    = YT_POOL_T;  // path slots per wavefront
constexpr int POOL_MASK = POOL_T - 1;
static_assert((POOL_T & POOL_MASK) == 0 && POOL_T >= 64 && POOL_T <= 256, "slot ids are bytes, rings are masked");

// Global storage of the pool kernel: one block of POOL_T slots per wavefront.
struct DPool {
  float4*     wgt;   // weight.xyz, max_roughness
  float4*     rad;   // radiance.xyz, sample index within the batch
  ulonglong2* rng;   // the pixel's PCG stream while it is resident
  int4*       misc;  // bounce | flags << 16 | opbounce << 24, i | j << 16, pixel, hit element
  float4*     hit;   // u, v, distance, instance
  int4*       park;  // a suspended walk: current node, stack pointer, instance being walked
  float4*     vol_a; // volume / deferred-pdf records, as DState's (indexed by pool slot)
  float4*     vol_b;
  float4*     pend;
  unsigned*   tile_counter;  // next tile to hand out
  const int*  stop;          // device-visible cancel flag (may be null)
  unsigned long long* dbg;   // watchdog / statistics (16 words per wavefront), may be null
  int      nwaves, ntiles;
  int      target;      // slots a wavefront tries to keep occupied (<= POOL_T, multiple of 64)
  int      refill_min;  // idle lanes that trigger a hand-over + refill
  int      shade_min;   // pending paths that allow a partial shade pass when no ray is queued
  int      max_iters;   // watchdog: main-loop iterations per wavefront
  unsigned tile_mul;    // tile order: tile = (k * tile_mul) % ntiles (1 = scanline order)
  int      rounds;      // 1: heavy passes only when no lane walks (lock-step rounds, nothing is ever parked)
  int      phase_min;   // takers a step kind (node / leaf / instance) needs to run in an iteration (1: always)
  int      heavy_min;   // pending paths that trigger the heavy passes while lanes still walk (64..POOL_T)
};

enum { POOL_DBG_ITERS = 0, POOL_DBG_WATCHDOG, POOL_DBG_ROUNDS, POOL_DBG_ACTIVE, POOL_DBG_SHADES, POOL_DBG_SHADED,
  POOL_DBG_STEPS, POOL_DBG_WSTEPS, POOL_DBG_REFILLS, POOL_DBG_TILES, POOL_DBG_CYCLES, POOL_DBG_STRIDE = 16 };

// pixel (row-major index in the slice) -> frame coordinates (inverse of slot_pixel)
YT_FN void pixel_coords(const DState& st, int pix, int& i, int& j) {
  int jl = pix / st.lwidth, il = pix - jl * st.lwidth;
  int tx = il / YT_TILE;
  i      = (st.col_first + tx * st.col_stride) * YT_TILE + (il - tx * YT_TILE);
  j      = st.row_begin + jl;
}

// Head of trace_sample (yocto_trace.cpp:1464-1468) for pixel (i, j)
YT_FN void start_sample_at(const DScene& sc, const DState& st, const KParams& kp, int i, int j, Path& P) {
  auto luv = rand2f(P.rng);  // g++ order: luv first
  auto puv = rand2f(P.rng);
  auto ray = sample_camera(sc.cameras[kp.camera], i, j, st.width, st.height, puv, luv, kp.tentfilter != 0);
  P.o = ray.o, P.d = ray.d;
  P.weight        = {1, 1, 1};
  P.radiance      = {0, 0, 0};
  P.max_roughness = 0;
  P.bounce = 0, P.opbounce = 0, P.flags = 0;
}

// resolve_step of yt_kernels.h for a pool slot: `ij` carries the pixel's frame
// coordinates; `stopped` (cancellation) ends the pixel at its sample boundary.
template <bool PEEK>
YT_FN int pool_resolve(const DScene& sc, const DState& st, const KParams& kp, Path& P, int ij, int step,
    int max_bounces, bool stopped) {
  if (step == STEP_DEFER) return OUT_DEFER;
  bool alive = false;
  if (step == STEP_NEXT) {
    P.bounce += 1;
    alive = P.bounce < max_bounces;
  } else if (step == STEP_RETRY) {
    alive = true;
  }
  if constexpr (PEEK) {
    if (alive && kp.peek && misses_scene_root(sc, P.o, P.d)) {
      if (P.bounce > 0 || !kp.envhidden) P.radiance += P.weight * eval_environment(sc, P.d);
      alive = false;
    }
  }
  if (alive) return OUT_BOUNCE;
  finish_sample(st, kp, 0, P);
  P.sidx += 1;
  if (P.sidx < st.batch && !stopped) {
    start_sample_at(sc, st, kp, ij & 0xffff, (ij >> 16) & 0xffff, P);
    return OUT_PRIMARY;
  }
  st.rngs[P.pix] = {P.rng.state, P.rng.inc};
  return OUT_DEAD;
}

YT_FN int pool_load_stop(const int* stop) {
  return stop ? __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
}

// Everything the launch needs, passed BY VALUE as the one kernel argument: the heavy
// passes (a separate function, below) read it straight from the kernarg segment.
struct PoolLaunch {
  DScene  sc;
  DState  st;  // its per-slot arrays (vol_a / vol_b / pend) point at the pool's, indexed by pool slot
  KParams kp;
  DPool   pl;
};

// LDS of one wavefront
struct PoolLds {
  StackEntry    stack[YT_LDS_DEPTH][64];
  float4        ra[POOL_T];  // ray of a slot: o.xyz, d.x
  float2        rb[POOL_T];  //                d.y, d.z
  unsigned char rq[POOL_T];  // ring: slots whose ray waits to be walked
  unsigned char pq[POOL_T];  // ring: slots whose walk is done, waiting to be shaded
  unsigned char lq[POOL_T];  // stack: slots waiting for their light-pdf walks (LP_DEFER)
  unsigned char xq[POOL_T];  // stack: slots whose ray the wide walk declined (binary redo)
  unsigned char fl[POOL_T];  // stack: free slots
};
typedef __attribute__((address_space(3))) PoolLds* PoolLdsP;

// Queue state of a wavefront (uniform)
struct PoolQ {
  int rq_h, rq_n, pq_h, pq_n, lq_n, xq_n, fl_n, live;
  int tiles_left, stopped;
};
YT_FN PoolQ uniform(PoolQ q) {  // values known to be wavefront-uniform → SGPRs
  auto u = [](int x) { return __builtin_amdgcn_readfirstlane(x); };
  return {u(q.rq_h), u(q.rq_n), u(q.pq_h), u(q.pq_n), u(q.lq_n), u(q.xq_n), u(q.fl_n), u(q.live), u(q.tiles_left), u(q.stopped)};
}
YT_FN bool pool_want_pass(const PoolQ& q, int n, int nactive, const DPool& pl) {
  if (pl.rounds) return n > 0 && nactive == 0 && q.rq_n == 0;
  return n >= 64 || (n > 0 && (nactive == 0 || (q.rq_n == 0 && n >= pl.shade_min)));
}
YT_FN bool pool_want_tile(const PoolQ& q, int nactive, const DPool& pl) {
  if (pl.rounds && nactive != 0) return false;
  return q.tiles_left && q.fl_n >= 64 && q.live - q.pq_n - q.lq_n - q.xq_n < pl.target;
}
// The kernel's test before it parks its walks and calls pool_heavy: as the tests inside,
// except that while lanes still walk a pass has to be worth the call (`heavy_min` pending
// paths instead of 64; the passes inside then run down to fewer than 64).
template <int LP>
YT_FN bool pool_want_heavy(const PoolQ& q, int nactive, const DPool& pl) {
  auto pass = [&](int n) {
    if (!pool_want_pass(q, n, nactive, pl)) return false;
    return nactive == 0 || q.rq_n == 0 || n >= pl.heavy_min;
  };
  return pass(q.pq_n) || (LP == LP_DEFER && pass(q.lq_n)) || pool_want_tile(q, nactive, pl) || (nactive == 0 && q.xq_n > 0);
}

// ---------------------------------------------------------------------------
// The heavy passes — shade, light pdf, binary redo, new tiles — as ONE out-of-line
// function: its code needs every register, and inlined into the walk loop it made the
// register allocator spill and reload the walk state around every visit of the
// management code (measured on the instanced scene: 5x the L2 requests of k_trace, L2 hit
// rate 99 % -> 70 %).  Out of line it has its own allocation; the caller PARKS the walks
// that are in progress (yt_pool.h::k_pool) so nothing of theirs is live across the call.
// Reads the launch block from the kernarg segment, whose address the kernel hands over
// (uniform → scalar loads).
// ---------------------------------------------------------------------------
template <int SAMPLER, int LP, bool MATTE>
__device__ __noinline__ PoolQ pool_heavy(PoolQ qin, int nactive_in, unsigned lds_addr, unsigned long long kernarg) {
  constexpr bool TRI  = MATTE;
  constexpr bool PEEK = true;
  // (generic pointers cast from the constant / LDS address spaces: the address-space
  // inference turns their uses into scalar loads / ds_* operations)
  typedef const __attribute__((address_space(4))) PoolLaunch* LaunchP;
  const unsigned long long ka = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(kernarg >> 32)) << 32) |
                                (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)kernarg);
  const PoolLaunch* K  = (const PoolLaunch*)(LaunchP)ka;
  const DScene      sc = K->sc;
  const DState      st = K->st;
  const KParams     kp = K->kp;
  const DPool       pl = K->pl;
  PoolLds* const    S  = (PoolLds*)(PoolLdsP)(size_t)(unsigned)(__builtin_amdgcn_readfirstlane((int)lds_addr));
  PoolQ           q  = uniform(qin);
  const int       nactive = __builtin_amdgcn_readfirstlane(nactive_in);
  const int                lane  = (int)threadIdx.x;
  const unsigned long long below = (1ull << lane) - 1ull;
  const int                gbase = (int)blockIdx.x * POOL_T;
  const int                max_bounces = max_bounces_of<SAMPLER>(kp);
  Stack stack;
  stack.lds = (lds_entry*)&S->stack[0][threadIdx.x];

  // routing of a shaded path's outcome (wavefront-wide; `valid` lanes carry a slot)
  auto route = [&](bool valid, int s, int cls) {
    const unsigned long long mr = __ballot(valid && (cls == OUT_PRIMARY || cls == OUT_BOUNCE));
    const unsigned long long md = __ballot(valid && cls == OUT_DEFER);
    const unsigned long long mf = __ballot(valid && cls == OUT_DEAD);
    if (valid && (cls == OUT_PRIMARY || cls == OUT_BOUNCE)) S->rq[(q.rq_h + q.rq_n + __popcll(mr & below)) & POOL_MASK] = (unsigned char)s;
    if (valid && cls == OUT_DEFER) S->lq[q.lq_n + __popcll(md & below)] = (unsigned char)s;
    if (valid && cls == OUT_DEAD) S->fl[q.fl_n + __popcll(mf & below)] = (unsigned char)s;
    q.rq_n += __popcll(mr);
    q.lq_n += __popcll(md);
    q.fl_n += __popcll(mf);
    q.live -= __popcll(mf);
  };
  auto store_slot = [&](int s, const Path& P, int ij) {
    const int gi = gbase + s;
    S->ra[s]     = {P.o.x, P.o.y, P.o.z, P.d.x};
    S->rb[s]     = {P.d.y, P.d.z};
    pl.wgt[gi]   = {P.weight.x, P.weight.y, P.weight.z, P.max_roughness};
    pl.rad[gi]   = {P.radiance.x, P.radiance.y, P.radiance.z, __int_as_float(P.sidx)};
    pl.rng[gi]   = {P.rng.state, P.rng.inc};
    pl.misc[gi]  = {P.bounce | (P.flags << 16) | (P.opbounce << 24), ij, P.pix, -1};
  };
  auto load_slot = [&](int s, Path& P, int& ij) {
    const int gi = gbase + s;
    float4    ra = S->ra[s];
    float2    rb = S->rb[s];
    float4    w = pl.wgt[gi], r = pl.rad[gi], h = pl.hit[gi];
    auto      g  = pl.rng[gi];
    int4      mi = pl.misc[gi];
    P.o = {ra.x, ra.y, ra.z};
    P.d = {ra.w, rb.x, rb.y};
    int inst        = __float_as_int(h.w);
    P.isec          = {inst, mi.w, h.x, h.y, h.z, inst >= 0};
    P.bounce        = mi.x & 0xffff;
    P.flags         = (mi.x >> 16) & 0xff;
    P.opbounce      = (mi.x >> 24) & 0xff;
    ij              = mi.y;
    P.pix           = mi.z;
    P.weight        = {w.x, w.y, w.z};
    P.max_roughness = w.w;
    P.radiance      = {r.x, r.y, r.z};
    P.sidx          = __float_as_int(r.w);
    P.rng           = {g.x, g.y};
  };
#ifdef YT_POOL_STATS
  unsigned long long stat_shades = 0, stat_shaded = 0, stat_tiles = 0;
#endif

  while (true) {
    const bool want_shade = pool_want_pass(q, q.pq_n, nactive, pl);
    const bool want_defer = !want_shade && LP == LP_DEFER && pool_want_pass(q, q.lq_n, nactive, pl);
    const bool want_redo  = !want_shade && !want_defer && nactive == 0 && q.xq_n > 0;
    const bool want_grab  = !want_shade && !want_defer && !want_redo && pool_want_tile(q, nactive, pl);
    if (!want_shade && !want_defer && !want_redo && !want_grab) break;
    __syncthreads();  // hit records / slot state written by other lanes are complete
    // ---- shade: one pass of up to 64 finished paths ---------------------------------
    if (want_shade) {
      const int m = q.pq_n < 64 ? q.pq_n : 64;
      if (pool_load_stop(pl.stop)) q.stopped = 1, q.tiles_left = 0;
      int s = -1, cls = OUT_DEAD;
      if (lane < m) {
        s = S->pq[(q.pq_h + lane) & POOL_MASK];
        Path P;
        int  ij;
        load_slot(s, P, ij);
        ShadeEnv E    = {sc, st, kp, nullptr, nullptr, gbase + s};
        int      step = step_path<SAMPLER, LP, MATTE ? 1 : 0>(E, P);
        cls           = pool_resolve<PEEK>(sc, st, kp, P, ij, step, max_bounces, q.stopped != 0);
        if (cls != OUT_DEAD) store_slot(s, P, ij);
      }
      q.pq_h = (q.pq_h + m) & POOL_MASK, q.pq_n -= m;
      route(lane < m, s, cls);
#ifdef YT_POOL_STATS
      stat_shades++, stat_shaded += m;
#endif
    }
    // ---- sample_lights_pdf's instance walks + the rest of the loop body -----------------
    // (their stack lives in scratch: the LDS columns belong to the parked scene walks)
    if constexpr (LP == LP_DEFER) {
      if (want_defer) {
        const int m = q.lq_n < 64 ? q.lq_n : 64;
        int       s = -1, cls = OUT_DEAD;
        if (lane < m) {
          s = S->lq[q.lq_n - 1 - lane];
          Path P;
          int  ij;
          load_slot(s, P, ij);
          float4   pd   = st.pend[gbase + s];
          Counters cnt  = {0, 0, 0, 0, 0, 0, 0, 0};
          auto     lpdf = sample_lights_pdf<3>(sc, P.o, P.d, &stack, &cnt);
          P.weight *= vec3f{pd.x, pd.y, pd.z} / (0.5f * pd.w + 0.5f * lpdf);
          int step = step_tail(P);
          cls      = pool_resolve<PEEK>(sc, st, kp, P, ij, step, max_bounces, q.stopped != 0);
          if (cls != OUT_DEAD) store_slot(s, P, ij);
        }
        q.lq_n -= m;
        route(lane < m, s, cls);
      }
    }
    // ---- rays the wide walk declined: the binary walk (no walk is parked: it may use the LDS stack)
    if (want_redo) {
      const int m = q.xq_n < 64 ? q.xq_n : 64;
      if (lane < m) {
        const int s   = S->xq[q.xq_n - 1 - lane];
        float4    ra  = S->ra[s];
        float2    rb  = S->rb[s];
        ray3f     ray = make_ray({ra.x, ra.y, ra.z}, {ra.w, rb.x, rb.y});
        Counters  cnt = {0, 0, 0, 0, 0, 0, 0, 0};
        Hit       h   = traverse<false, false, TRI>(sc, ray, -1, false, stack, cnt);
        const int gi  = gbase + s;
        pl.hit[gi]    = {h.u, h.v, h.distance, __int_as_float(h.hit ? h.instance : -1)};
        pl.misc[gi].w = h.element;
        S->pq[(q.pq_h + q.pq_n + lane) & POOL_MASK] = (unsigned char)s;
      }
      q.xq_n -= m, q.pq_n += m;
    }
    // ---- pixels: take another tile while too few rays are in flight -----------------------
    // in flight = queued or being walked; the rest of the live slots wait for a shade or
    // light-pdf pass.  Demand-driven, so the frame's tiles drain at the pace the
    // wavefronts can take them (dynamic load balance), and a wavefront whose pixels are
    // cheap (sky) simply turns more tiles over.
    if (want_grab) {
      if (pool_load_stop(pl.stop)) q.stopped = 1, q.tiles_left = 0;
      unsigned k = 0;
      if (q.tiles_left) {
        if (lane == 0) k = atomicAdd(pl.tile_counter, 1u);
        k = (unsigned)__builtin_amdgcn_readfirstlane((int)k);
        if (k >= (unsigned)pl.ntiles) q.tiles_left = 0;
      }
      if (q.tiles_left) {
        const int tile = (int)(((unsigned long long)k * pl.tile_mul) % (unsigned)pl.ntiles);
        int       i, j;
        const int pix = slot_pixel(st, tile * YT_BLOCK + lane, i, j);
        const unsigned long long mv = __ballot(pix >= 0);
        const int                nv = __popcll(mv);
        if (pix >= 0) {
          const int s = S->fl[q.fl_n - 1 - __popcll(mv & below)];
          Path      P;
          auto      r = st.rngs[pix];
          P.rng  = {r.x, r.y};
          P.sidx = 0;
          P.pix  = pix;
          P.isec = {-1, -1, 0, 0, 0, false};
          start_sample_at(sc, st, kp, i, j, P);
          store_slot(s, P, i | (j << 16));
          S->rq[(q.rq_h + q.rq_n + __popcll(mv & below)) & POOL_MASK] = (unsigned char)s;
        }
        q.fl_n -= nv, q.live += nv, q.rq_n += nv;
#ifdef YT_POOL_STATS
        stat_tiles++;
#endif
      }
    }
  }
  __syncthreads();
#ifdef YT_POOL_STATS
  if (pl.dbg && lane == 0) {
    unsigned long long* g = pl.dbg + blockIdx.x * POOL_DBG_STRIDE;
    g[POOL_DBG_SHADES] += stat_shades, g[POOL_DBG_SHADED] += stat_shaded, g[POOL_DBG_TILES] += stat_tiles;
  }
#endif
  return q;
}

template <int SAMPLER, int LP, bool MATTE>
__global__ void __launch_bounds__(64, YT_WAVES_PER_EU) k_pool(PoolLaunch launch) {
  static_assert(SAMPLER == YTHIP_SAMPLER_PATH || SAMPLER == YTHIP_SAMPLER_PATHTEST, "pool kernel: path samplers");
  static_assert(LP == LP_NONE || LP == LP_DEFER, "pool kernel: no inline NEE");
  constexpr bool TRI = MATTE;  // the "simple scene" class: triangle meshes only
  const DScene&  sc  = launch.sc;
  const DPool&   pl  = launch.pl;
  __shared__ PoolLds lds_block;
  PoolLds* const     S = &lds_block;
#ifdef YT_POOL_STATS
  __shared__ unsigned s_wsteps;
  if (threadIdx.x == 0) s_wsteps = 0;
  unsigned long long stat_rounds = 0, stat_active = 0, stat_refills = 0;
  unsigned           lsteps = 0;
#endif
  const int                lane  = (int)threadIdx.x;
  const unsigned long long below = (1ull << lane) - 1ull;
  const int                gbase = (int)blockIdx.x * POOL_T;

  for (int k = lane; k < POOL_T; k += 64) S->fl[k] = (unsigned char)k;
  __syncthreads();

  // queue state (wavefront-uniform)
  PoolQ q = {0, 0, 0, 0, 0, 0, POOL_T, 0, 1, 0};

  // walk state of this lane (yt_bvh.h::traverse<false, true, TRI>, only_instance < 0, !find_any)
  int   slot = -1;    // slot whose ray this lane walks / has walked, -1 none
  bool  done = true;  // no walk in progress
  vec3f wo = {0, 0, 0}, wd = {0, 0, 0}, wdinv = {0, 0, 0}, o = {0, 0, 0}, d = {0, 0, 0}, dinv = {0, 0, 0};
  int   wsign = 0, sign = 0, cur = REF_NONE, sp = 0, kind = KIND_NONE, leafbias = 0, cur_inst = -1;
  float tmax = 0, tmaxk = 0;
  Hit   best = {-1, -1, 0, 0, 0, false};
  lds_entry* const lds = (lds_entry*)&S->stack[0][threadIdx.x];
  StackEntry       spill[YT_SPILL];
  auto push = [&](int ref, float t0) {
    StackEntry v = {ref, __float_as_int(t0)};
    if (sp < YT_LDS_DEPTH)
      lds[sp * YT_BLOCK].ref = v.ref, lds[sp * YT_BLOCK].t0 = v.t0;
    else if (sp < YT_LDS_DEPTH + YT_SPILL)
      spill[sp - YT_LDS_DEPTH] = v;
    sp++;
  };
  auto pop = [&]() -> StackEntry {
    sp--;
    if (sp < YT_LDS_DEPTH) {
      StackEntry v;
      v.ref = lds[sp * YT_BLOCK].ref, v.t0 = lds[sp * YT_BLOCK].t0;
      return v;
    }
    return (sp < YT_LDS_DEPTH + YT_SPILL) ? spill[sp - YT_LDS_DEPTH] : StackEntry{REF_EXIT, 0};
  };
  // head of intersect_scene_bvh for the ray of `slot` (make_ray: tmin 1e-4, tmax flt_max)
  auto begin_walk = [&](float4 ra, float2 rb) {
    wo    = {ra.x, ra.y, ra.z};
    wd    = {ra.w, rb.x, rb.y};
    tmax  = flt_max;
    tmaxk = tmax * BBOX_K;
    wdinv = {1 / wd.x, 1 / wd.y, 1 / wd.z};
    wsign = ((wdinv.x < 0) ? 1 : 0) | ((wdinv.y < 0) ? 2 : 0) | ((wdinv.z < 0) ? 4 : 0);
    best  = {-1, -1, 0, 0, 0, false};
    done  = true;
    sp = 0, cur = REF_NONE, cur_inst = -1, kind = KIND_NONE, leafbias = 0;
    if (!ray_is_tame(wo, wdinv, ray_eps)) {  // the wide walk declines: redone by the binary walk
      best.instance = HIT_ABORT;
      return;
    }
    o = wo, d = wd, dinv = wdinv, sign = wsign;
    if (sc.tlas_ref == REF_NONE) return;
    float t0;
    if (!(slab<false>(o, dinv, ray_eps, sc.tlas_bmin, sc.tlas_bmax, t0) && t0 <= tmaxk)) return;
    cur  = sc.tlas_ref;
    done = false;
  };
  auto enter = [&](int inst) -> int {
    const float4* ti = reinterpret_cast<const float4*>(sc.tinst + inst);
    float4        m0 = ti[0], m1 = ti[1], m2 = ti[2], m3 = ti[3], m4 = ti[4];
    int4          m5 = reinterpret_cast<const int4*>(ti)[5];
    int           root = __float_as_int(m4.z);
    if (root == REF_NONE) return REF_NONE;
    frame3f inv = {{m0.x, m0.y, m0.z}, {m0.w, m1.x, m1.y}, {m1.z, m1.w, m2.x}, {m2.y, m2.z, m2.w}};
    vec3f   io   = transform_point(inv, wo);
    vec3f   id   = transform_vector(inv, wd);
    vec3f   idin = {1 / id.x, 1 / id.y, 1 / id.z};
    if (!ray_is_tame(io, idin, ray_eps)) {  // irregular at this instance's level: binary redo
      best = Hit{HIT_ABORT, -1, 0, 0, 0, false};
      done = true;
      return REF_NONE;
    }
    float t0;
    bool  ok = slab<false>(io, idin, ray_eps, {m3.x, m3.y, m3.z}, {m3.w, m4.x, m4.y}, t0) && t0 <= tmaxk;
    if (!ok) return REF_NONE;
    o = io, d = id, dinv = idin;
    sign     = ((dinv.x < 0) ? 1 : 0) | ((dinv.y < 0) ? 2 : 0) | ((dinv.z < 0) ? 4 : 0);
    cur_inst = inst;
    kind     = TRI ? KIND_TRIANGLES : __float_as_int(m4.w);
    leafbias = m5.x;
    push(REF_EXIT, 0);
    return root;
  };
  auto accept = [&](int element, const PrimHit& h) {
    best  = {cur_inst, element, h.u, h.v, h.t, true};
    tmax  = h.t;
    tmaxk = h.t * BBOX_K;
  };

  int iters = 0;
#ifdef YT_POOL_STATS
  const long long clk0 = __builtin_readcyclecounter();
#endif
  while (true) {
    if (++iters > pl.max_iters) {  // watchdog: never hang the device
      if (pl.dbg && lane == 0) pl.dbg[blockIdx.x * POOL_DBG_STRIDE + POOL_DBG_WATCHDOG] = 1ull | ((unsigned long long)q.live << 8) | ((unsigned long long)q.rq_n << 20) | ((unsigned long long)q.pq_n << 32) | ((unsigned long long)q.lq_n << 44);
      break;
    }
    // ---- (1) finished walks hand their hit records over ---------------------------------
    {
      const bool               fin  = done && slot >= 0;
      const unsigned long long mfin = __ballot(fin);
      if (mfin) {
        const bool ab = fin && best.instance == HIT_ABORT;
        if (fin && !ab) {
          const int gi  = gbase + slot;
          pl.hit[gi]    = {best.u, best.v, best.distance, __int_as_float(best.hit ? best.instance : -1)};
          pl.misc[gi].w = best.element;
        }
        const unsigned long long mok = __ballot(fin && !ab), mab = __ballot(ab);
        if (fin && !ab) S->pq[(q.pq_h + q.pq_n + __popcll(mok & below)) & POOL_MASK] = (unsigned char)slot;
        if (ab) S->xq[q.xq_n + __popcll(mab & below)] = (unsigned char)slot;
        q.pq_n += __popcll(mok), q.xq_n += __popcll(mab);
        if (fin) slot = -1;
      }
    }
    int nactive = 64 - __popcll(__ballot(done));

    // ---- (2) the heavy passes (pool_heavy) -------------------------------------------------
    // The lanes that are still walking PARK their walk — 36 B per lane in the slot's global
    // records: current node, stack pointer, instance being walked, tmax and the best hit
    // so far — and rebuild everything else afterwards (the world ray from LDS, the
    // instance-level ray by redoing the transform: same operations on the same
    // operands), so no walk state is live across the call.
    if (pool_want_heavy<LP>(q, nactive, pl)) {
      if (!done) {
        const int gi  = gbase + slot;
        pl.hit[gi]    = {best.u, best.v, tmax, __int_as_float(best.hit ? best.instance : -1)};
        pl.misc[gi].w = best.element;
        pl.park[gi]   = {cur, sp, cur_inst, 0};
      }
      q = uniform(pool_heavy<SAMPLER, LP, MATTE>(q, nactive, (unsigned)(size_t)(PoolLdsP)S,
          (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr()));
      if (!done) {
        const int gi = gbase + slot;
        float4    h  = pl.hit[gi];
        int4      pk = pl.park[gi];
        const int el = pl.misc[gi].w;
        float4    ra = S->ra[slot];
        float2    rb = S->rb[slot];
        wo    = {ra.x, ra.y, ra.z};
        wd    = {ra.w, rb.x, rb.y};
        wdinv = {1 / wd.x, 1 / wd.y, 1 / wd.z};
        wsign = ((wdinv.x < 0) ? 1 : 0) | ((wdinv.y < 0) ? 2 : 0) | ((wdinv.z < 0) ? 4 : 0);
        tmax  = h.z;
        tmaxk = tmax * BBOX_K;
        const int bi = __float_as_int(h.w);
        best     = {bi, el, h.x, h.y, bi >= 0 ? h.z : 0.0f, bi >= 0};
        cur      = pk.x;
        sp       = pk.y;
        cur_inst = pk.z;
        if (cur_inst >= 0) {  // the instance-level ray again (as in enter())
          const float4* ti = reinterpret_cast<const float4*>(sc.tinst + cur_inst);
          float4        m0 = ti[0], m1 = ti[1], m2 = ti[2], m4 = ti[4];
          int4          m5 = reinterpret_cast<const int4*>(ti)[5];
          frame3f inv = {{m0.x, m0.y, m0.z}, {m0.w, m1.x, m1.y}, {m1.z, m1.w, m2.x}, {m2.y, m2.z, m2.w}};
          o        = transform_point(inv, wo);
          d        = transform_vector(inv, wd);
          dinv     = {1 / d.x, 1 / d.y, 1 / d.z};
          sign     = ((dinv.x < 0) ? 1 : 0) | ((dinv.y < 0) ? 2 : 0) | ((dinv.z < 0) ? 4 : 0);
          kind     = TRI ? KIND_TRIANGLES : __float_as_int(m4.w);
          leafbias = m5.x;
        } else {
          o = wo, d = wd, dinv = wdinv, sign = wsign, kind = KIND_NONE, leafbias = 0;
        }
      } else {
        wo = wd = wdinv = o = d = dinv = {0, 0, 0};
        wsign = sign = 0, tmax = tmaxk = 0, cur = REF_NONE, sp = 0, cur_inst = -1, kind = KIND_NONE, leafbias = 0;
        best = {-1, -1, 0, 0, 0, false};
      }
    }

    // ---- (1b) idle lanes take the next queued rays -------------------------------------
    if (q.rq_n > 0) {
      const unsigned long long idle = __ballot(done);
      if (idle) {
        const int nidle = __popcll(idle), rank = __popcll(idle & below);
        if (done && rank < q.rq_n) {
          slot = S->rq[(q.rq_h + rank) & POOL_MASK];
          begin_walk(S->ra[slot], S->rb[slot]);
        }
        const int ntake = nidle < q.rq_n ? nidle : q.rq_n;
        q.rq_h = (q.rq_h + ntake) & POOL_MASK, q.rq_n -= ntake;
#ifdef YT_POOL_STATS
        stat_refills++;
#endif
      }
    }
    nactive = 64 - __popcll(__ballot(done));

    // ---- (3) nothing walks --------------------------------------------------------
    if (nactive == 0) {
      if (__ballot(done && slot >= 0) != 0) continue;  // walks that ended at the root: hand them over first
      if (q.rq_n > 0 || q.pq_n > 0 || q.lq_n > 0 || q.xq_n > 0) continue;  // the next iteration starts them
      if (q.live == 0 && !q.tiles_left) break;
      if (q.live != 0 && !q.tiles_left) {  // cannot happen: every live slot is queued somewhere
        if (pl.dbg && lane == 0) pl.dbg[blockIdx.x * POOL_DBG_STRIDE + POOL_DBG_WATCHDOG] = 2ull | ((unsigned long long)q.live << 8);
        break;
      }
      continue;
    }

    // ---- (4) the walk ------------------------------------------------------------------
    // A lane's walk alternates node steps (W: a quad record = two tree levels), leaf steps
    // (L: up to two primitives of a BLAS leaf) and instance entries (E).  With 64 unrelated
    // rays the lanes want different things at any moment: "descend until everybody holds a
    // leaf" (yt_bvh.h's while-while, fine for the coherent rays of one tile) leaves most
    // lanes waiting, and what bounds this loop is not instruction issue but the chain of
    // dependent fetches.  So an iteration (a) lets every lane do its cheap bookkeeping — pops,
    // pop-time culling, instance exits, TLAS-leaf expansion —, (b) lets every lane FETCH the
    // one 96-128-B record its next step needs, whatever kind it is, all in flight together
    // (one memory round trip per iteration for the whole wavefront), and (c) runs the steps.
    // A kind with fewer than `phase_min` takers (and not the most wanted one) sits the
    // iteration out.  The loop holds nothing but the walk state in registers; it returns to
    // the management code above when `refill_min` walks have finished (or none is left).
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
            continue;
          }
          if (cur < 0 && cur_inst < 0) {
            // TLAS leaf: its instances in order, each to completion (yocto_bvh.cpp:600-609)
            const int first = cur & 0x0fffffff, num = (cur >> 28) & 7;
            for (int k = num - 1; k >= 1; k--) push(REF_INST + (((first + k) << 1) | (k == num - 1 ? 1 : 0)), 0);
            cur = num > 0 ? REF_INST + ((first << 1) | (num == 1 ? 1 : 0)) : REF_NONE;
            continue;
          }
          break;
        }
      }
      bool wantW = !done && (unsigned)cur < (unsigned)REF_INST;
      bool wantL = !done && cur < 0;
      bool wantE = !done && cur >= REF_INST;
      {
        const int nW = __popcll(__ballot(wantW)), nL = __popcll(__ballot(wantL)), nE = __popcll(__ballot(wantE));
        const int mx = nW >= nL && nW >= nE ? nW : (nL >= nE ? nL : nE);
        if (nW < pl.phase_min && nW != mx) wantW = false;
        if (nL < pl.phase_min && nL != mx) wantL = false;
        if (nE < pl.phase_min && nE != mx) wantE = false;
#ifdef YT_POOL_STATS
        stat_rounds++, stat_active += nW + nL + nE;
        if (nW + nL + nE > 0) {
          lsteps += (lane == 0) ? (wantW ? nW : 0) + ((nL >= pl.phase_min || nL == mx) ? nL : 0) + ((nE >= pl.phase_min || nE == mx) ? nE : 0) : 0;
          if (lane == 0) atomicAdd(&s_wsteps, 1u);
        }
#endif
      }
      // (b) the record of this lane's step
      float4 r0, r1, r2, r3, r4, r5, r6, r7;
      r0 = r1 = r2 = r3 = r4 = r5 = r6 = r7 = float4{0, 0, 0, 0};
      int lfirst = 0, lnum = 0, einst = -1;
      if (wantW) {
        const float4* Qp = sc.wide + 8 * (int64_t)cur;
        r0 = Qp[0], r1 = Qp[1], r2 = Qp[2], r3 = Qp[3], r4 = Qp[4], r5 = Qp[5], r6 = Qp[6], r7 = Qp[7];
      } else if (wantL) {
        lfirst = cur & 0x0fffffff, lnum = (cur >> 28) & 7;
        if (TRI || kind == KIND_TRIANGLES) {  // two triangles (the pool is padded, over-reads are ignored)
          const float4* L = sc.leafdata + (leafbias + lfirst * 3);
          r0 = L[0], r1 = L[1], r2 = L[2], r3 = L[3], r4 = L[4], r5 = L[5];
        } else if (kind == KIND_QUADS) {  // two quads
          const float4* L = sc.leafdata + (leafbias + lfirst * 4);
          r0 = L[0], r1 = L[1], r2 = L[2], r3 = L[3], r4 = L[4], r5 = L[5], r6 = L[6], r7 = L[7];
        } else if (kind == KIND_LINES) {  // two lines
          const float4* L = sc.leafdata + (leafbias + lfirst * 3);
          r0 = L[0], r1 = L[1], r2 = L[2], r3 = L[3], r4 = L[4], r5 = L[5];
        } else {  // four points
          const float4* L = sc.leafdata + (leafbias + lfirst * 2);
          r0 = L[0], r1 = L[1], r2 = L[2], r3 = L[3], r4 = L[4], r5 = L[5], r6 = L[6], r7 = L[7];
        }
      } else if (wantE) {
        einst            = sc.tlas_prims[(cur - REF_INST) >> 1];
        const float4* ti = reinterpret_cast<const float4*>(sc.tinst + einst);
        r0 = ti[0], r1 = ti[1], r2 = ti[2], r3 = ti[3], r4 = ti[4], r5 = ti[5];
      }
      // (c) the steps
      if (wantW) {
        // internal node, two levels at once (yt_bvh.h WIDE)
        float  ta, tb, tc, td;
        bool   fa = slab<true>(o, dinv, ray_eps, {r0.x, r0.y, r1.x}, {r0.z, r0.w, r1.y}, ta);
        bool   fb = slab<true>(o, dinv, ray_eps, {r2.x, r2.y, r3.x}, {r2.z, r2.w, r3.y}, tb);
        bool   fc = slab<true>(o, dinv, ray_eps, {r4.x, r4.y, r5.x}, {r4.z, r4.w, r5.y}, tc);
        bool   fd = slab<true>(o, dinv, ray_eps, {r6.x, r6.y, r7.x}, {r6.z, r6.w, r7.y}, td);
        int    ra = (fa && ta <= tmaxk) ? __float_as_int(r1.z) : REF_NONE;
        int    rb = (fb && tb <= tmaxk) ? __float_as_int(r3.z) : REF_NONE;
        int    rc = (fc && tc <= tmaxk) ? __float_as_int(r5.z) : REF_NONE;
        int    rd = (fd && td <= tmaxk) ? __float_as_int(r7.z) : REF_NONE;
        const int  axes = __float_as_int(r1.w);
        const bool hs = ((sign >> (axes & 3)) & 1) != 0, ls = ((sign >> ((axes >> 2) & 3)) & 1) != 0,
                   rs = ((sign >> ((axes >> 4) & 3)) & 1) != 0;
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
      if (wantL) {
        // BLAS leaf — yocto_bvh.cpp:505-545: its primitives in order, two (points: four) per
        // iteration; what is left stays in `cur` for the next one
        int ndone = 2;
        if (TRI || kind == KIND_TRIANGLES) {
          auto h = intersect_triangle(o, d, ray_eps, tmax, {r0.x, r0.y, r0.z}, {r0.w, r1.x, r1.y}, {r1.z, r1.w, r2.x});
          if (h.hit) accept(__float_as_int(r2.y), h);
          if (lnum > 1) {
            h = intersect_triangle(o, d, ray_eps, tmax, {r3.x, r3.y, r3.z}, {r3.w, r4.x, r4.y}, {r4.z, r4.w, r5.x});
            if (h.hit) accept(__float_as_int(r5.y), h);
          }
        } else if (kind == KIND_QUADS) {
          auto h = intersect_quad(o, d, ray_eps, tmax, {r0.x, r0.y, r0.z}, {r0.w, r1.x, r1.y}, {r1.z, r1.w, r2.x}, {r2.y, r2.z, r2.w});
          if (h.hit) accept(__float_as_int(r3.x), h);
          if (lnum > 1) {
            h = intersect_quad(o, d, ray_eps, tmax, {r4.x, r4.y, r4.z}, {r4.w, r5.x, r5.y}, {r5.z, r5.w, r6.x}, {r6.y, r6.z, r6.w});
            if (h.hit) accept(__float_as_int(r7.x), h);
          }
        } else if (kind == KIND_LINES) {
          auto h = intersect_line(o, d, ray_eps, tmax, {r0.x, r0.y, r0.z}, {r0.w, r1.x, r1.y}, r1.z, r1.w);
          if (h.hit) accept(__float_as_int(r2.x), h);
          if (lnum > 1) {
            h = intersect_line(o, d, ray_eps, tmax, {r3.x, r3.y, r3.z}, {r3.w, r4.x, r4.y}, r4.z, r4.w);
            if (h.hit) accept(__float_as_int(r5.x), h);
          }
        } else if (kind == KIND_POINTS) {
          ndone  = 4;
          auto h = intersect_point(o, d, ray_eps, tmax, {r0.x, r0.y, r0.z}, r0.w);
          if (h.hit) accept(__float_as_int(r1.x), h);
          if (lnum > 1) {
            h = intersect_point(o, d, ray_eps, tmax, {r2.x, r2.y, r2.z}, r2.w);
            if (h.hit) accept(__float_as_int(r3.x), h);
          }
          if (lnum > 2) {
            h = intersect_point(o, d, ray_eps, tmax, {r4.x, r4.y, r4.z}, r4.w);
            if (h.hit) accept(__float_as_int(r5.x), h);
          }
          if (lnum > 3) {
            h = intersect_point(o, d, ray_eps, tmax, {r6.x, r6.y, r6.z}, r6.w);
            if (h.hit) accept(__float_as_int(r7.x), h);
          }
        }
        cur = lnum > ndone ? (int)(0x80000000u | ((unsigned)(lnum - ndone) << 28) | (unsigned)(lfirst + ndone)) : REF_NONE;
      }
      if (wantE) {
        // intersect_shape_bvh's prologue for the instance (as enter() of yt_bvh.h): transform_ray,
        // the pop + slab test of the BLAS root, whose bbox travels in the instance record
        const int root = __float_as_int(r4.z);
        cur            = REF_NONE;
        if (root != REF_NONE) {
          frame3f inv  = {{r0.x, r0.y, r0.z}, {r0.w, r1.x, r1.y}, {r1.z, r1.w, r2.x}, {r2.y, r2.z, r2.w}};
          vec3f   io   = transform_point(inv, wo);
          vec3f   id   = transform_vector(inv, wd);
          vec3f   idin = {1 / id.x, 1 / id.y, 1 / id.z};
          if (!ray_is_tame(io, idin, ray_eps)) {  // irregular at this instance's level: binary redo
            best = Hit{HIT_ABORT, -1, 0, 0, 0, false};
            done = true;
          } else {
            float t0;
            if (slab<false>(io, idin, ray_eps, {r3.x, r3.y, r3.z}, {r3.w, r4.x, r4.y}, t0) && t0 <= tmaxk) {
              o = io, d = id, dinv = idin;
              sign     = ((dinv.x < 0) ? 1 : 0) | ((dinv.y < 0) ? 2 : 0) | ((dinv.z < 0) ? 4 : 0);
              cur_inst = einst;
              kind     = TRI ? KIND_TRIANGLES : __float_as_int(r4.w);
              leafbias = __float_as_int(r5.x);
              push(REF_EXIT, 0);
              cur = root;
            }
          }
        }
      }
      const unsigned long long mdone = __ballot(done);
      if (__popcll(__ballot(done && slot >= 0)) >= pl.refill_min || mdone == ~0ull) break;
    }
  }
  if (pl.dbg) {
    if (lane == 0) pl.dbg[blockIdx.x * POOL_DBG_STRIDE + POOL_DBG_ITERS] = (unsigned long long)iters;
#ifdef YT_POOL_STATS
    unsigned ls = lsteps;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ls += __shfl_xor(ls, off);
    if (lane == 0) {
      unsigned long long* g = pl.dbg + blockIdx.x * POOL_DBG_STRIDE;
      g[POOL_DBG_ROUNDS] = stat_rounds, g[POOL_DBG_ACTIVE] = stat_active;
      g[POOL_DBG_STEPS] = ls, g[POOL_DBG_WSTEPS] = s_wsteps, g[POOL_DBG_REFILLS] = stat_refills;
      g[POOL_DBG_CYCLES] = (unsigned long long)(__builtin_readcyclecounter() - clk0);
    }
#endif
  }
}

}  // namespace yt
