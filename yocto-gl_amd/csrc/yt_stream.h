// yt_stream.h — the between-bounce STREAMING scheduler of trace_samples (round 6; ythip_set_scheduler(ctx, 1)).
//
// k_trace (yt_kernels.h) is a persistent megakernel: one wavefront owns a 16 x 4 pixel tile for a whole batch, so
// the 64 rays a wavefront walks together are whatever its tile's paths happen to be doing — camera rays next to
// fifth-bounce rays going in every direction.  On incoherent workloads 10-22 of the 64 lanes of a VALU instruction
// are live, and no wavefront can trade rays with another.  This file is the other design — the one
// north_star names: "a wavefront (streaming) path tracer with SoA ray / hit / path-state buffers ... stream
// compaction for active-path sorting" — built so that it regroups rays BETWEEN bounces, at per-bounce cost:
//
//   * every pixel of the slice is in flight (one path slot per pixel of the tile grid, SoA path state in HBM:
//     ray 32 B, weight / radiance 32 B, rng 16 B, hit record 20 B); a path that ends is accumulated and its pixel's
//     next sample regenerated in place, so the queues never drain before the batch's tail;
//   * one GENERATION = every live path advances one bounce:
//         ks_scan     exclusive prefix over the histogram of the sort keys the previous generation emitted
//         ks_scatter  counting sort: slot -> its place in the ray queue (the rank came with the histogram's atomic)
//         ks_extend   intersect_scene_bvh for queue entries [0, n): the walk of k_intersect_batch and nothing else,
//                     wavefronts = 64 CONSECUTIVE queue entries = rays of one direction octant from one cell of
//                     the scene box (camera rays: of neighbouring tiles)
//         ks_shade    one iteration of the integrator's bounce loop per live path, in SLOT order (coalesced state,
//                     trace_state rows touched as in k_trace), incl. the deferred light-pdf walks, the end-of-sample
//                     accumulation and the regeneration; emits the next ray's sort key + histogram count;
//   * dependent launches on one stream; a generation's kernels all return at once when nothing is queued, and the
//     host stops enqueueing when a read-back of the queue length says zero.
//   * the path slots run as two such chains of generations on two streams (one chain's shade / sort launches fill the machine
//     while the other's extend launch drains), and once a chain's queue has shrunk to a quarter of its slots the rest of the
//     batch — a few scattered pixels still short of their samples — runs in ONE launch: ks_finish, a lane per queue entry,
//     extend and shade in turn on the same state.
//   * samplers: `path` (ks_shade runs the deferred light-pdf walks of the lanes that deferred), `pathdirect` (and the NEE ray
//     of a bounce, as k_trace's deferred stage does), `naive`, `pathtest`.  ythip_set_scheduler(ctx, 2) — the default — lets
//     the library choose between this scheduler and k_trace per trace_state by timing a batch of each.
//
// Bit-exact by construction: the per-pixel sequence of operations is k_trace's (the same start_sample / step_path /
// resolve_step / finish_sample on the same values) — only WHICH wavefront executes a ray's walk changes, and hit
// records do not depend on that.  Whole-trace_state digests equal the megakernel's (tests/test_gpu_stream.py).
//
// Restates the scheduling of yocto_trace.cpp:1595-1619 (trace_samples: parallel_for over pixels, each pixel's samples
// in order) around yocto_trace.cpp:453-596 (trace_path) and yocto_bvh.cpp:554-617 (intersect_scene_bvh).
#pragma once

#include "yt_kernels.h"

namespace yt {

constexpr unsigned SKEY_DEAD = 0xffffffffu;  // the slot has no ray for the next generation (its pixel's batch is done)
constexpr int      YT_STREAM_GEN_LOG = 8192;
constexpr int      YT_STREAM_MAX_GROUPS = 8;
constexpr int      PF_DEAD   = 0x80;         // Path flag of such a slot (ray_b.w)

struct DStream {
  // path state, one record per slot of the tile grid = per pixel (SoA of 16-B pieces; WgState's layout, in HBM) — arrays shared by the groups
  float4 *    ray_a, *ray_b, *wgt, *rad;
  ulonglong2* rng;
  float4*     hit_a;  // u, v, distance, instance (-1: miss)
  int*        hit_e;  // element
  unsigned *  key, *rank;  // the next ray's sort key, its arrival number inside the key's bin
  // one GROUP of path slots = one chain of generations on one stream (two groups overlap each other's launch tails)
  // The group's path slots: the grid's tiles (64 slots) dealt to the groups round-robin in chunks of 2^gshift tiles (16 by default),
  // so that the chains carry the same work whatever the image's top and bottom look like.  (As contiguous runs — the first form —
  // one chain of the corpus scenes got the sky and the other the scene: features1 306 -> 404 Msamples/s, materials1 392 -> 500 with
  // the deal; cfg2b's tail kernel ran 13.5 ms for one half and 3.5 for the other.  Tile by tile costs the closed boxes 2 %: a chain's
  // camera-ray bins and cells then span the whole image.)  slot0 = the group's offset into the per-group arrays (queue, step_log);
  // nslots = its slot count, a multiple of 64.
  int       slot0, nslots, gstride, goff, gshift;  // (chunks of 2^gshift tiles are dealt round-robin)
  unsigned *hist, *offs;    // per bin: count (zeroed by ks_scan), exclusive prefix
  int*      queue;          // the group's slots in key order
  int*      counts;         // [0] rays queued for the running generation (ks_scan), [1] generations run
  int*      gen_rays;       // profiling: queue length of generation g, for g < YT_STREAM_GEN_LOG (null: off)
  int*      step_log;       // profiling: the traversal steps of every queue entry of generation log_gen (null: off)
  int       log_gen;
  unsigned long long* stats;  // optional (profiling): [0] sum of lane steps, [1] 64 x longest lane per wavefront, [2] wavefronts, [3] rays
  int   nbins;        // bounce-ray bins: 8 octants x 2^(3 cell_bits) cells
  int   nprim_bins;   // camera-ray bins (groups of neighbouring tiles), after the bounce-ray bins
  int   cell_bits, prim_shift;
  int   order;        // 0: octant major, cell minor; 1: cell major, octant minor; 2: no sort (slot order)
  vec3f cell_lo, cell_scale;  // cell = (o - lo) * scale, the TLAS root box cut into 2^cell_bits cells per axis
};

// the k-th path slot of the group
YT_FN int stream_slot(const DStream& S, int k) {
  const int lt = k >> 6, lc = lt >> S.gshift, r = lt - (lc << S.gshift);  // the group's tile, its chunk of 2^gshift tiles, the tile inside it
  return ((((lc * S.gstride + S.goff) << S.gshift) + r) << 6) | (k & 63);
}

YT_FN unsigned spread3(unsigned x) {  // 10 bits -> every third bit
  x &= 0x3ff;
  x = (x | (x << 16)) & 0x30000ff;
  x = (x | (x << 8)) & 0x300f00f;
  x = (x | (x << 4)) & 0x30c30c3;
  x = (x | (x << 2)) & 0x9249249;
  return x;
}

// The sort key of a path's next ray.  Camera rays keep their tile neighbourhood (they ARE coherent); everything else goes by
// direction octant and by the Morton code of the origin's cell.
YT_FN unsigned stream_key(const DStream& S, int slot, vec3f o, vec3f d, bool primary) {  // slot: the PIXEL's (Path::vslot)
  if (S.order == 2) return (unsigned)(slot >> 6) % (unsigned)(S.nbins + S.nprim_bins);  // (wraps: bins are then a few tiles far apart)
  if (primary) return (unsigned)S.nbins + (unsigned)(slot >> S.prim_shift) % (unsigned)S.nprim_bins;
  const int   top = (1 << S.cell_bits) - 1;
  const float cx = (o.x - S.cell_lo.x) * S.cell_scale.x, cy = (o.y - S.cell_lo.y) * S.cell_scale.y, cz = (o.z - S.cell_lo.z) * S.cell_scale.z;
  // (NaN / out-of-box origins land in the first or last cell: the key only groups rays, it decides nothing)
  const unsigned ix = (unsigned)min_(max_((int)cx, 0), top), iy = (unsigned)min_(max_((int)cy, 0), top), iz = (unsigned)min_(max_((int)cz, 0), top);
  const unsigned cell   = spread3(ix) | (spread3(iy) << 1) | (spread3(iz) << 2);
  const unsigned octant = (d.x < 0 ? 1u : 0u) | (d.y < 0 ? 2u : 0u) | (d.z < 0 ? 4u : 0u);
  return S.order == 0 ? (octant << (3 * S.cell_bits)) | cell : (cell << 3) | octant;
}

// the path state's loads / stores in one place (measured as non-temporal accesses — 200 MB per generation that wash the tree
// out of the L2s —: -1 ... -3 %, docs/HISTORY.md)
template <typename T>
YT_FN T sld(const T* p) { return *p; }
template <typename T>
YT_FN void sst(T* p, T v) { *p = v; }

// a live slot's state <-> registers (load_path_rest / store_path of yt_kernels.h on the HBM arrays)
YT_FN void stream_load_rest(const DState& st, const DStream& S, int slot, Path& P, float4 rb) {
  float4 w = sld(S.wgt + slot), r = sld(S.rad + slot);
  auto   g = sld(S.rng + slot);
  int    pi, pj;
  P.vslot         = slot;
  P.pix           = slot_pixel(st, P.vslot, pi, pj);
  P.bounce        = __float_as_int(rb.z);
  int fw          = __float_as_int(rb.w);
  P.flags         = fw & 0xff;
  P.opbounce      = fw >> 8;
  P.weight        = {w.x, w.y, w.z};
  P.max_roughness = w.w;
  P.radiance      = {r.x, r.y, r.z};
  P.sidx          = __float_as_int(r.w);
  P.rng           = {g.x, g.y};
}
YT_FN void stream_store(const DStream& S, int slot, const Path& P) {
  sst(S.rng + slot, ulonglong2{P.rng.state, P.rng.inc});
  sst(S.ray_a + slot, float4{P.o.x, P.o.y, P.o.z, P.d.x});
  sst(S.ray_b + slot, float4{P.d.y, P.d.z, __int_as_float(P.bounce), __int_as_float(P.flags | (P.opbounce << 8))});
  sst(S.wgt + slot, float4{P.weight.x, P.weight.y, P.weight.z, P.max_roughness});
  sst(S.rad + slot, float4{P.radiance.x, P.radiance.y, P.radiance.z, __int_as_float(P.sidx)});
}

// The slot's entry in the next generation's queue: key + histogram count; the returning atomic IS the rank inside the bin.
// Camera rays of one wavefront (= one tile) share a bin: one atomic for all of them, ranks in lane order, so a tile's camera
// rays stay next to each other in the queue.  Must be called with the wavefront's live lanes converged on `cls`.
YT_FN void stream_emit(const DStream& S, int slot, const Path& P, int cls) {
  unsigned key = SKEY_DEAD, rank = 0;
  const bool   live = cls != OUT_DEAD;
  if (live) key = stream_key(S, P.vslot, P.o, P.d, cls == OUT_PRIMARY);
  // One atomic per DISTINCT key of the wavefront, not per lane (neighbouring pixels' rays share octants and cells: 10-20
  // distinct keys per 64 lanes; a per-lane returning atomic was the slowest part of the first version).  The lanes are
  // grouped key by key with ballots — no memory traffic —, every group's first lane then fetches the group's base with ONE
  // wave-level atomic instruction, and the others take it from that lane.
  const int          lane = (int)(threadIdx.x & 63);
  unsigned long long todo = __ballot(live);
  int                leader = lane;
  unsigned           prefix = 0, count = 0;
  while (todo) {
    const int                first = __ffsll((long long)todo) - 1;
    const unsigned           k     = (unsigned)__shfl((int)key, first);
    const unsigned long long m     = __ballot(live && key == k) & todo;
    if (live && key == k) leader = first, prefix = (unsigned)__popcll(m & ((1ull << lane) - 1ull)), count = (unsigned)__popcll(m);
    todo &= ~m;
  }
  unsigned base = 0;
  if (live && leader == lane) base = atomicAdd(&S.hist[key], count);
  base = (unsigned)__shfl((int)base, leader);
  rank = base + prefix;
  sst(S.key + slot, key);
  sst(S.rank + slot, rank);
}

#ifdef YT_STREAM_KERNELS  // the plain (non-template) kernels are compiled by ONE unit: yt_stream.hip defines this
// ---------------------------------------------------------------------------------------------------------------------
// head of the batch: every pixel's first camera ray (k_trace's prologue)
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(YT_BLOCK) ks_init(DScene sc, DState st, KParams kp, DStream S) {
  const int slot = stream_slot(S, (int)blockIdx.x * YT_BLOCK + (int)threadIdx.x);
  int       i, j;
  const int pix = slot_pixel(st, slot, i, j);
  int       cls = OUT_DEAD;
  Path      P;
  P.o = {0, 0, 0}, P.d = {0, 0, 0}, P.vslot = slot;
  if (pix >= 0) {
    auto r  = st.rngs[pix];
    P.rng   = {r.x, r.y};
    P.sidx  = 0;
    P.pix   = pix;
    start_sample(sc, st, kp, slot, P);
    stream_store(S, slot, P);
    cls = OUT_PRIMARY;
  } else {
    S.ray_b[slot] = {0, 0, 0, __int_as_float(PF_DEAD)};  // a slot of an edge tile outside the slice
  }
  stream_emit(S, slot, P, cls);
}

// ---------------------------------------------------------------------------------------------------------------------
// the counting sort between two generations
// ---------------------------------------------------------------------------------------------------------------------
// One workgroup: offs = exclusive prefix of hist, hist = 0, counts[0] = total.  The bins in tiles of 4096 (16 consecutive bins
// per thread as four uint4), a running carry from tile to tile.  (4 k - 40 k bins: 1 - 10 tiles.  The first version gave every
// thread a contiguous run of ALL its bins — strided, uncoalesced reads: 50 us per launch at 37 k bins, 460 us at 278 k.)
// FOUR wavefronts, not sixteen: the scan sits on its chain's critical path while the other chain's extend launch fills the
// machine, and a 1024-thread workgroup has to wait for one CU with sixteen free wave slots — it averaged 51 us per launch
// in a cfg2b batch (12 us alone); four slots are free somewhere at once.
constexpr int YT_SCAN_THREADS = 256;
__global__ void __launch_bounds__(YT_SCAN_THREADS) ks_scan(DStream S) {
  __shared__ unsigned s_wave[YT_SCAN_THREADS / 64];
  __shared__ unsigned s_carry;
  const int nb  = S.nbins + S.nprim_bins;  // (the arrays are padded to a multiple of 4096)
  const int tid = (int)threadIdx.x;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int t0 = 0; t0 < nb; t0 += 16 * YT_SCAN_THREADS) {
    const int b = t0 + 16 * tid;
    uint4     c[4];
    unsigned  part[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      c[j] = reinterpret_cast<const uint4*>(S.hist + b)[j];
      if (b + 4 * j + 0 >= nb) c[j].x = 0;
      if (b + 4 * j + 1 >= nb) c[j].y = 0;
      if (b + 4 * j + 2 >= nb) c[j].z = 0;
      if (b + 4 * j + 3 >= nb) c[j].w = 0;
      part[j] = sum;
      sum += c[j].x + c[j].y + c[j].z + c[j].w;
    }
    unsigned x = sum;  // inclusive scan over the workgroup: shuffles inside a wavefront, LDS across
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      unsigned y = (unsigned)__shfl_up((int)x, off);
      if ((tid & 63) >= off) x += y;
    }
    if ((tid & 63) == 63) s_wave[tid >> 6] = x;
    __syncthreads();
    unsigned before = s_carry, total = 0;
#pragma unroll
    for (int w = 0; w < YT_SCAN_THREADS / 64; w++) {
      if (w < (tid >> 6)) before += s_wave[w];
      total += s_wave[w];
    }
    const unsigned run0 = before + x - sum;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const unsigned run = run0 + part[j];
      reinterpret_cast<uint4*>(S.offs + b)[j] = {run, run + c[j].x, run + c[j].x + c[j].y, run + c[j].x + c[j].y + c[j].z};
      reinterpret_cast<uint4*>(S.hist + b)[j] = {0, 0, 0, 0};
    }
    __syncthreads();
    if (tid == 0) s_carry += total;
    __syncthreads();
  }
  if (tid == 0) {
    const unsigned total = s_carry;
    S.counts[0]          = (int)total;
    if (total) {
      const int g = S.counts[1];
      if (S.gen_rays && g < YT_STREAM_GEN_LOG) S.gen_rays[g] = (int)total;
      S.counts[1] = g + 1;
    }
  }
}

__global__ void __launch_bounds__(256) ks_scatter(DStream S) {
  if (S.counts[0] == 0) return;
  const int k = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (k >= S.nslots) return;
  const int      slot = stream_slot(S, k);
  const unsigned key  = S.key[slot];
  if (key != SKEY_DEAD) S.queue[S.offs[key] + S.rank[slot]] = slot;
}

#endif  // YT_STREAM_KERNELS

// ---------------------------------------------------------------------------------------------------------------------
// extend: the walk and nothing else (k_intersect_batch's shape: no shading state, no spills)
// ---------------------------------------------------------------------------------------------------------------------
#ifndef YT_STREAM_EXTEND_WAVES
#define YT_STREAM_EXTEND_WAVES 4
#endif
// profiling: how even are the walks of a wavefront?
YT_FN void stream_walk_stats(const DStream& S, unsigned steps) {
  unsigned sum = steps, mx = steps;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    sum += __shfl_xor(sum, off);
    unsigned o = __shfl_xor(mx, off);
    mx         = o > mx ? o : mx;
  }
  const unsigned long long rays = __popcll(__ballot(steps > 0));
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* c = S.stats + 8 * (blockIdx.x & 63);
    atomicAdd(&c[0], (unsigned long long)sum);
    atomicAdd(&c[1], 64ull * mx);
    atomicAdd(&c[2], 1ull);
    atomicAdd(&c[3], rays);
  }
}

template <bool WIDE, int TRI, bool PHASED>
__global__ void __launch_bounds__(YT_BLOCK, YT_STREAM_EXTEND_WAVES) ks_extend(DScene sc, DStream S) {
  __shared__ StackEntry s_stack[YT_LDS_DEPTH][YT_BLOCK];
  const int n = S.counts[0];
  const int i = (int)blockIdx.x * YT_BLOCK + (int)threadIdx.x;
  if ((int)blockIdx.x * YT_BLOCK >= n) return;
  unsigned steps = 0;
  if (i < n) {
    Stack stack;
    YT_STACK_INIT(stack, s_stack);
    Counters     cnt  = {0, 0, 0, 0, 0, 0, 0, 0};
    const int    slot = S.queue[i];
    const float4 ra = sld(S.ray_a + slot), rb = sld(S.ray_b + slot);
    const ray3f  ray = make_ray({ra.x, ra.y, ra.z}, {ra.w, rb.x, rb.y});
    const Hit    h   = traverse_any<false, WIDE, TRI, PHASED>(sc, ray, -1, false, stack, cnt);
    sst(S.hit_a + slot, float4{h.u, h.v, h.distance, __int_as_float(h.hit ? h.instance : -1)});
    sst(S.hit_e + slot, h.element);
    steps = cnt.steps + 1;
    if (S.step_log && S.counts[1] - 1 == S.log_gen) S.step_log[i] = (int)steps;
  }
  if (S.stats) stream_walk_stats(S, steps);
}

// ---------------------------------------------------------------------------------------------------------------------
// shade: one iteration of trace_path's bounce loop for every live slot, in slot order — k_trace's shade stage, its
// deferred light-pdf stage (run by the lanes that deferred, right here: the state is in registers) and its
// resolve_step (accumulate, regenerate), then the next ray's entry in the sort
// ---------------------------------------------------------------------------------------------------------------------
#ifndef YT_STREAM_SHADE_WAVES
#define YT_STREAM_SHADE_WAVES 4
#endif
// one live slot through one iteration of the bounce loop: state in, state out (stored), the slot's queue class returned
// ... on a path held in registers (P complete, isec = this iteration's hit): the iteration itself; returns the slot's queue class
template <int SAMPLER, int LP, int CLS>
YT_FN int stream_shade_path(const DScene& sc, const DState& st, const KParams& kp, int slot, bool stopped, Stack stack, Path& P) {
  constexpr bool MATTE = CLS == 1;
  constexpr int  PRIMS = MATTE ? 1 : (CLS == 3 ? 2 : 0);
  constexpr bool PEEK  = SAMPLER == YTHIP_SAMPLER_PATH || SAMPLER == YTHIP_SAMPLER_PATHTEST || SAMPLER == YTHIP_SAMPLER_NAIVE;
  static_assert(SAMPLER != YTHIP_SAMPLER_PATHDIRECT || LP == LP_DEFER, "pathdirect runs its walks in the deferred part");
  const int max_bounces = max_bounces_of<SAMPLER>(kp);
  ShadeEnv  E           = {sc, st, kp, slot};
  int       step;
  if constexpr (SAMPLER == YTHIP_SAMPLER_NAIVE) step = step_naive<SAMPLER>(E, P);
  else step = step_path<SAMPLER, LP, CLS>(E, P);
  if constexpr (LP == LP_DEFER) {
    if (step == STEP_DEFER) {  // the rest of the loop body behind the light pdf's instance walks (k_trace's walk stage)
      Counters cnt = {0, 0, 0, 0, 0, 0, 0, 0};
      int      nee = 0;
      if constexpr (SAMPLER == YTHIP_SAMPLER_PATHDIRECT) {
        // the NEE half of the loop body (yocto_trace.cpp:670-693) for the light direction step_path drew: pdf walks, the NEE
        // ray — walked right here, in slot order: neighbouring pixels' NEE rays aim at the same few lights and are coherent as they
        // come (a sort + extend launch of their own was measured: +-4 %, tools/experiments/r06_stream_pathdirect_two_stage.patch) —,
        // the emission it finds
        const float4 na = st.nee_a[slot];
        nee             = __float_as_int(na.w);
        if (nee) {
          const float4 nb  = st.nee_b[slot];
          const vec3f  inc = {na.x, na.y, na.z}, bsdfcos = {nb.x, nb.y, nb.z};
          const float  pdf = sample_lights_pdf<2, false, PRIMS>(sc, P.o, inc, &stack, &cnt);
          if (bsdfcos != vec3f{0, 0, 0} && pdf > 0) {
            const ray3f nray     = make_ray(P.o, inc);
            const Hit   nisec    = traverse_any<false, true, PRIMS>(sc, nray, -1, false, stack, cnt);
            const auto  emission = nee_emission<CLS>(sc, nisec, inc);
            P.radiance += P.weight * bsdfcos * emission / pdf;
          }
        }
      }
      step = STEP_END;
      if (nee != 2) {
        const float4 pd   = st.pend[slot];
        const float  lpdf = sample_lights_pdf<2, false, PRIMS>(sc, P.o, P.d, &stack, &cnt);
        P.weight *= vec3f{pd.x, pd.y, pd.z} / (0.5f * pd.w + 0.5f * lpdf);
        step = step_tail(P);
      }
    }
  }
  return resolve_step<PEEK>(sc, st, kp, slot, P, step, max_bounces, stopped);
}
template <int SAMPLER, int LP, int CLS>
YT_FN int stream_shade_slot(const DScene& sc, const DState& st, const KParams& kp, const DStream& S, int slot, float4 rb, bool stopped, Stack stack,
    Path& P) {
  const float4 ra = sld(S.ray_a + slot), ha = sld(S.hit_a + slot);
  P.o             = {ra.x, ra.y, ra.z};
  P.d             = {ra.w, rb.x, rb.y};
  const int inst  = __float_as_int(ha.w);
  P.isec          = {inst, sld(S.hit_e + slot), ha.x, ha.y, ha.z, inst >= 0};
  if (inst < 0) P.isec = {-1, -1, 0, 0, 0, false};
  stream_load_rest(st, S, slot, P, rb);
  const int cls = stream_shade_path<SAMPLER, LP, CLS>(sc, st, kp, slot, stopped, stack, P);
  if (cls == OUT_DEAD) P.flags |= PF_DEAD;
  stream_store(S, slot, P);
  return cls;
}

template <int SAMPLER, int LP, int CLS, bool WIDE>
__global__ void __launch_bounds__(YT_BLOCK, YT_STREAM_SHADE_WAVES) ks_shade(DScene sc, DState st, KParams kp, DStream S) {
  __shared__ StackEntry s_stack[LP == LP_DEFER ? YT_LDS_DEPTH : 1][YT_BLOCK];
  if (S.counts[0] == 0) return;  // nothing was queued for this generation: the batch is done
  const int  slot    = stream_slot(S, (int)blockIdx.x * YT_BLOCK + (int)threadIdx.x);
  const bool stopped = stop_requested(st.stop, st.stop_gen) || (blockIdx.x == 0 && relay_stop(st));
  const float4 rb = sld(S.ray_b + slot);
  const bool   live = !(__float_as_int(rb.w) & PF_DEAD);
  int  cls = OUT_DEAD;
  Path P;
  P.o = {0, 0, 0}, P.d = {0, 0, 0};
  if (live) {
    Stack stack;
    YT_STACK_INIT(stack, s_stack);
    cls = stream_shade_slot<SAMPLER, LP, CLS>(sc, st, kp, S, slot, rb, stopped, stack, P);
  }
  stream_emit(S, slot, P, cls);
}

// ---------------------------------------------------------------------------------------------------------------------
// finish: the tail of a batch without generations.  Once the queue is a fraction of the path slots (the pixels still short of
// their batch are few and scattered), every further generation is four launches and a drain for a handful of wavefronts, and
// waits for the slowest walk among them — 90-110 of the 270-320 generations of a 64-spp batch on the closed box / the hair hold
// 2-4 % of its rays and take 12-16 % of its time.  ks_finish takes the LAST sorted queue instead: one lane per entry carries its
// path slot through extend and shade, iteration after iteration, until the slot's pixel has taken its batch — the same two
// bodies as ks_extend / ks_shade on the same HBM state (the state goes through memory between them on purpose: the kernel then
// needs the registers of the larger body, not of both; with the path held in registers instead — k_trace's inner loop on a lane — the
// tail and a whole batch in this kernel take the same time), no sort, no launches, no generation waiting for anybody.  Per-pixel
// order untouched: still bit for bit the fused kernel's batch.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef YT_STREAM_FINISH_WAVES
#define YT_STREAM_FINISH_WAVES 4
#endif
template <int SAMPLER, int LP, int CLS, bool WIDE, int TRI, bool PHASED>
__global__ void __launch_bounds__(YT_BLOCK, YT_STREAM_FINISH_WAVES) ks_finish(DScene sc, DState st, KParams kp, DStream S) {
  __shared__ StackEntry s_stack[YT_LDS_DEPTH][YT_BLOCK];
  const int n = S.counts[0];
  const int i = (int)blockIdx.x * YT_BLOCK + (int)threadIdx.x;
  if (i >= n) return;
  int   slot = S.queue[i];
  Stack stack;
  YT_STACK_INIT(stack, s_stack);
  for (unsigned iter = 0;; iter++) {
    // (the slot made opaque once per phase: the compiler otherwise hoists every `array + slot` address of both bodies out of the loop
    //  and keeps them alive across the walk: classes 1-3 spill 0-40 VGPRs instead of 7-55 without it, materials1 with the whole batch
    //  in this kernel 82 -> 77 ms; the general class of `path` spills its 230-250 either way, and 3 waves per SIMD for it lost: 96 -> 104 ms)
    asm volatile("" : "+v"(slot));
    {  // extend (ks_extend's body)
      Counters     cnt = {0, 0, 0, 0, 0, 0, 0, 0};
      const float4 ra = sld(S.ray_a + slot), rb = sld(S.ray_b + slot);
      const ray3f  ray = make_ray({ra.x, ra.y, ra.z}, {ra.w, rb.x, rb.y});
      const Hit    h   = traverse_any<false, WIDE, TRI, PHASED>(sc, ray, -1, false, stack, cnt);
      sst(S.hit_a + slot, float4{h.u, h.v, h.distance, __int_as_float(h.hit ? h.instance : -1)});
      sst(S.hit_e + slot, h.element);
    }
    asm volatile("" : "+v"(slot) : : "memory");  // (the hit record and the path state are re-read, not carried in registers across the walk)
    bool stopped = stop_requested(st.stop, st.stop_gen);
    if (blockIdx.x == 0 && (iter & 15) == 0 && st.stop_host &&
        __hip_atomic_load(st.stop_host, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == st.stop_gen) {  // relay_stop, by whichever lanes are left
      __hip_atomic_store(st.stop, st.stop_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      stopped = true;
    }
    Path         P;
    const float4 rb  = sld(S.ray_b + slot);
    const int    cls = stream_shade_slot<SAMPLER, LP, CLS>(sc, st, kp, S, slot, rb, stopped, stack, P);
    if (cls == OUT_DEAD) break;
    asm volatile("" ::: "memory");
  }
}

}  // namespace yt
