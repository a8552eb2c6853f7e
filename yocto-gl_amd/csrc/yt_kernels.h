// yt_kernels.h — the wavefront (streaming) path tracer.
//
// One persistent path slot per pixel, SoA path state in HBM.  Per iteration:
//
//   k_extend    BVH traversal of every queued ray (the traversal kernel)
//   k_shade     ONE iteration of the integrator's bounce-loop body for every
//               queued path.  A path that ends is accumulated into the image
//               right there (tail of trace_sample) and, if its pixel still owes
//               samples to this batch, REGENERATED with the pixel's next camera
//               ray (head of trace_sample).  Samples of one pixel stay strictly
//               sequential (the per-pixel PCG stream and the running-mean lerp
//               are order-dependent), pixels are independent.
//   k_lightpdf  only for scenes with area lights: the nested instance walks of
//               sample_lights_pdf (up to 100 per light), deferred out of k_shade
//               so that kernel carries no traversal, then the rest of the loop
//               body (weight update, termination tests, russian roulette).
//
// Compaction is BLOCK-LOCAL and atomic-free: workgroup b (ONE wavefront, see
// YT_BLOCK in yt_bvh.h) owns the YT_BLOCK pixel slots [YT_BLOCK b, YT_BLOCK (b + 1))
// for the whole batch and, after every iteration,
// partitions them (wave ballots + an LDS prefix over its waves) into its own
// double-ended queue segment: regenerated (primary) rays from the front,
// continuing (bounce) rays from the back, dead slots dropped.  Waves of the next
// k_extend therefore stay homogeneous (primary rays of neighbouring pixels /
// bounce rays), there is no global counter to contend on (one device-scope
// atomic word saturates at ~88 ops/us on MI355X — measured 280 us per launch
// with per-wave atomics), and the order is deterministic.
//
// Restates libs/yocto/yocto_trace.cpp:338-1492 (sample_camera, sample_lights,
// sample_lights_pdf, the nine integrators, trace_sample).  The rng draw order at
// every multi-argument call site is the g++ (right-to-left) order, SURVEY.md
// Appendix A-13.
#pragma once

#include "yt_bvh.h"
#include "yt_shading.h"

#ifdef YT_FAST  // tolerance mode: multiply-adds written in this file may fuse (the traversal's, in yt_bvh.h, never do)
#pragma clang fp contract(fast)
#endif
namespace yt {

// Path flags (low byte) | opbounce << 8
enum {
  PF_HIT        = 1,   // trace_result.hit
  PF_VOLUME     = 2,   // volume_stack non-empty
  PF_NOEMIT     = 4,   // !next_emission            (pathdirect / pathmis)
  PF_SKIPEXTEND = 8,   // pathmis: use next_intersection instead of tracing (yocto_trace.cpp:795)
  PF_INVOL      = 16,  // furnace: in_volume
};

// How sample_lights_pdf's instance walks are executed
enum {
  LP_NONE   = 0,  // scene has no area lights: no traversal needed
  LP_DEFER  = 2,  // walks (and the NEE rays of pathdirect / pathmis) run in the walk stage of k_trace
  // (1 was LP_INLINE — walks inside the shade step, round 1's pathdirect / pathmis kernels: 1.2-1.4x
  //  slower and 1300+ spilled registers; tools/experiments/README.md)
};

// Device mirror of trace_state (yocto_trace.h:147-157) + wavefront path state.
struct DState {
  int width, height, row_begin, rows, npix;
  // column striping (multi-GPU load balance): the slice holds the
  // 16-pixel-wide tile columns col_first, col_first + col_stride, ... of the
  // frame, side by side in a local image `lwidth` pixels wide (the whole frame:
  // col_first 0, col_stride 1, lwidth == width)
  int lwidth, col_first, col_stride;
  // path slots: the slice is cut into 16 x YT_TILE_H pixel tiles (16x4 with one
  // wavefront per workgroup); workgroup (logical block) t owns the YT_BLOCK slots
  // of tile t for the whole batch.  Slots whose pixel falls outside the slice are
  // never queued.
  int tiles_x, tiles_y, nblocks, nslots;
  int sample_base;  // state.samples at the start of this batch
  int batch;        // samples to add per pixel in this batch
  int only_pix;     // >= 0: trace_sample() — only this local pixel takes the sample (one workgroup)
  // trace_state
  float4*     image;   // vec4f
  float*      albedo;  // vec3f
  float*      normal;  // vec3f
  int*        hits;
  ulonglong2* rngs;    // rng_state {state, inc}
  // rarely-touched path state, one slot per pixel of the tile grid (the hot
  // path state lives in LDS for the whole batch: WgState)
  float4* vol_a;    // volume: density.xyz, scattering.x
  float4* vol_b;    // volume: scattering.yz, scanisotropy, -
  float4* nhit_a;   // pathmis next_intersection: u, v, distance, instance
  int*    nhit_e;   // pathmis next_intersection: element
  float4* pend;     // deferred light pdf: bsdfcos.xyz (or scattering), bsdf pdf
  // deferred NEE (pathdirect, pathmis with LP_DEFER), per slot:
  //   pathdirect: nee_a = light direction xyz, state (0 none, 1 pending, 2 pending + path ends); nee_b = its bsdfcos
  //   pathmis:    nee_a / nee_b = light-sampled direction + bsdf pdf / its bsdfcos + flags (1 pass 0 ran, 2 pass 1 ran,
  //               4 volume event: `pend` holds the mixture terms instead); nee_c / nee_d = the same for the
  //               bsdf-sampled direction; nee_e = the weight factor of the continuation
  float4 *nee_a, *nee_b, *nee_c, *nee_d, *nee_e;
  // work counters (ythip_stats), may be null
  unsigned long long* counters;
  // cancellation: a device-visible flag the host raises when the caller's `stop` goes up
  // (yocto_trace.cpp:1636-1637 polls context.stop per sample); may be null
  int*       stop;       // device word every workgroup polls once per iteration
  const int* stop_host;  // pinned host word ythip_cancel writes; relayed into *stop by the kernel (ythip.hip, alloc_stop_word)
  int        stop_gen;   // this batch's number: the batch stops when the word equals it (begin_batch)
  // longest-tile-first launch order (yt_order.hip): workgroup b renders tile tile_perm[b]; every
  // workgroup records the cycles its tile took for the next launch's order.  null: off.
  const int* tile_perm;
  unsigned*  tile_cost;
  // pixel pool (opt-in, YTHIP_PIXEL_POOL=1; docs/HISTORY.md): fewer workgroups than tiles; a lane whose pixel has taken its
  // batch takes the next pixel of the queue (tiles in launch order, 64 entries each) instead of going idle.  null: off.
  int* pool_next;
  int  pool_total;  // queue length = nblocks * YT_BLOCK
};
YT_FN bool stop_requested(const int* stop, int gen) {
  return stop && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen;
}
// the relay: one look at the host's word (a read over the fabric: rare by construction), passed on to the device word
YT_FN bool relay_stop(const DState& st) {
  if (!st.stop_host || __hip_atomic_load(st.stop_host, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != st.stop_gen) return false;
  if (threadIdx.x == 0) __hip_atomic_store(st.stop, st.stop_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}

enum { CNT_RAYS = 0, CNT_NODES, CNT_TRIS, CNT_QUADS, CNT_LINES, CNT_POINTS, CNT_INST, CNT_SHADES, CNT_SAMPLES, CNT_NUM };

struct KParams {
  int   camera, sampler, falsecolor, bounces;
  float clamp;
  int   nocaustics, envhidden, tentfilter;
  int   has_env;  // !scene.environments.empty()
  int   hold;     // scheduling policy of k_trace (0 off, 1 hold back partial primary wavefronts)
  int   peek;     // resolve a continuing path's root-box miss in place (resolve_step), 0 off
};

#ifndef YT_TILE_W  // development builds: tile shape experiments (8 -> 8 x 8 tiles; single-GPU full frames only)
#define YT_TILE_W 16
#endif
constexpr int YT_TILE   = YT_TILE_W;                  // a workgroup's tile: 16 pixels wide ...
constexpr int YT_TILE_H = YT_BLOCK / YT_TILE;  // ... and YT_BLOCK / 16 tall (16 x 4 for one wavefront)
constexpr int YT_PROBE_ITERS = 48;  // iterations over which a workgroup measures its per-class traversal work
static_assert(YT_TILE * YT_TILE_H == YT_BLOCK, "one tile per workgroup");

// Block → tile mapping.  Hardware block b runs on XCD b % 8 (each XCD has its own
// 4 MB L2), so with the identity mapping each XCD serves every 8th tile of a
// tile row.  XCD-aware alternatives were measured on the 1M-triangle plane
// (k_extend per launch): identity 68 us; one contiguous band of tile rows per
// XCD 78 us (the sky band idles); tile rows ty = x (mod 8) per XCD 73 us (45
// tile rows do not divide by 8).  The tree's leaf level dominates the L2
// footprint and is private to a pixel neighbourhood under any mapping, so
// balance wins; the hook stays here in case a scene says otherwise.  (Re-measured
// with the adaptive-wait scheduler: vertical bands of tile columns per XCD, plain
// and interleaved in groups of 2 / 4 columns, gain 2-7 % on the plane and the
// instanced scene and LOSE 15-45 % on the Cornell box, whose column costs differ.)
// (Also measured: keeping "tile column mod 8 = XCD" — what the identity mapping gives
// for 1280 / 1920 / 2560 px — for widths whose tile-column count is not a multiple of
// 8: within the 1-2 % run-to-run noise, not adopted.)
YT_FN int logical_block(const DState& st) {
  int b = (int)blockIdx.x;
  if (b >= st.nblocks) return -1;
  if (st.only_pix >= 0) {  // trace_sample(): a one-workgroup launch on the pixel's tile
    int jl = st.only_pix / st.lwidth, il = st.only_pix - jl * st.lwidth;
    return b == 0 ? (jl / YT_TILE_H) * st.tiles_x + il / YT_TILE : -1;
  }
  if (st.tile_perm) return st.tile_perm[b];
  return b;
}

// Pixel of a path slot: index into the slice's trace_state arrays (row-major,
// as the reference lays them out), -1 when outside.  A wave covers 16 x 4
// pixels: 256-B runs of the image rows (8 x 8 quadrants traced 1 % faster but
// shaded 6 % slower: trace_state rows are touched in 128-B pieces).
YT_FN int slot_pixel(const DState& st, int slot, int& i, int& j) {
  int tile = slot / YT_BLOCK, w = slot & (YT_BLOCK - 1);
  int ty = tile / st.tiles_x, tx = tile - ty * st.tiles_x;
  int il = tx * YT_TILE + (w % YT_TILE);
  i      = (st.col_first + tx * st.col_stride) * YT_TILE + (w % YT_TILE);
  int jl = ty * YT_TILE_H + (w / YT_TILE);
  j      = st.row_begin + jl;
  return (i < st.width && jl < st.rows) ? jl * st.lwidth + il : -1;
}

// The counters are kept in CNT_BANKS copies (one 128-B line each, chosen by block
// index) that the host sums: a single device-scope atomic word saturates at
// ~88 ops/us on MI355X.
constexpr int CNT_BANKS = 64, CNT_STRIDE = 16;
YT_FN int     cnt_bank() { return (int)(blockIdx.x & (CNT_BANKS - 1)) * CNT_STRIDE; }

// Counting mode only: per-lane work counters → wavefront sum (butterfly over
// __shfl_xor) → one global atomic per counter per wave.  Must be called with
// the whole wavefront converged (idle lanes pass zeros).
YT_FN void flush_counters(unsigned long long* c, const Counters& cnt) {
  if (!c) return;
  unsigned  v[7]   = {cnt.rays, cnt.nodes, cnt.triangles, cnt.quads, cnt.lines, cnt.points, cnt.instances};
  const int idx[7] = {CNT_RAYS, CNT_NODES, CNT_TRIS, CNT_QUADS, CNT_LINES, CNT_POINTS, CNT_INST};
#pragma unroll
  for (int k = 0; k < 7; k++) {
    unsigned x = v[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
    if ((threadIdx.x & 63) == 0 && x) atomicAdd(&c[cnt_bank() + idx[k]], (unsigned long long)x);
  }
}
// +1 per active lane of the current (possibly diverged) control flow: one atomic per wave
YT_FN void count_lanes(unsigned long long* c, int idx) {
  if (!c) return;
  unsigned long long m = __ballot(1);
  if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicAdd(&c[cnt_bank() + idx], (unsigned long long)__popcll(m));
}

// ---------------------------------------------------------------------------
// shading-point helpers
// ---------------------------------------------------------------------------
struct Surface {
  frame3f        frame;
  DShape         shc;
  ythip_material mat;  // (by value: what is not used is never loaded — everything is inlined)
  elem4          e;
  vec2f          uv;
};

// TRI: what the caller knows about the scene's shapes (as traverse(): 1 triangle meshes only, 2 triangle and quad meshes only)
// LEAF (class 1: untextured triangle meshes): a mesh without vertex normals and colours is shaded from the hit's leaf record alone
// (TriPos, yt_scene.h) — its element's indices are not fetched; any other mesh goes through its indices as before
template <int TRI = 0, bool LEAF = false>
YT_FN Surface load_surface(const DScene& sc, int instance, int element, vec2f uv) {
  // The instance, shape and material records through the scalar cache whenever the lanes that shade together agree on
  // them (wave_uniform, yt_bvh.h): one instance per scene, or thousands of instances of one shape with one material, are
  // the common cases, and three records are 14 + 12 + 21 dwords per lane of vector loads otherwise.
  Surface        s;
  ythip_instance inst;
#define YT_LD_RECORD(p) (*(p))
  if (int u; SCALAR_LOADS && wave_uniform(instance, u)) inst = ldc_record(sc.instances + u);
  else inst = YT_LD_RECORD(sc.instances + instance);
  s.frame = ldframe(inst.frame);
  if (int u; SCALAR_LOADS && wave_uniform(inst.shape, u)) s.shc = ldc_record(sc.shapes + u);
  else s.shc = YT_LD_RECORD(sc.shapes + inst.shape);
  if (TRI == 1) s.shc.kind_eval = KIND_TRIANGLES;
  if (TRI == 2 && s.shc.kind_eval != KIND_TRIANGLES) s.shc.kind_eval = KIND_QUADS;
  if (int u; SCALAR_LOADS && wave_uniform(inst.material, u)) s.mat = ldc_record(sc.materials + u);
  else s.mat = YT_LD_RECORD(sc.materials + inst.material);
#undef YT_LD_RECORD
  if (LEAF && s.shc.normals < 0 && s.shc.colors < 0) s.e = {0, 0, 0, 0};
  else s.e = load_element(sc, s.shc, element);
  s.uv    = uv;
  return s;
}

// sample_lights — yocto_trace.cpp:361-388
YT_FN vec3f sample_lights(const DScene& sc, vec3f position, float rl, float rel, vec2f ruv) {
  if (sc.num_lights <= 0) return {0, 0, 0};  // (reference: out-of-bounds read; ytrace never gets here)
  auto        light_id = sample_uniform(sc.num_lights, rl);
  const DLight light = load_record(sc.lights, light_id);
  if (light.instance != YTHIP_INVALIDID) {
    const ythip_instance inst = load_record(sc.instances, light.instance);
    const DShape         sh   = load_record(sc.shapes, inst.shape);
    auto        element = sample_discrete(sc.cdf + light.cdf_offset, light.cdf_count, rel);
    auto        uv      = (sh.kind_eval == KIND_TRIANGLES) ? sample_triangle(ruv) : ruv;
    auto        e       = load_element(sc, sh, element);
    auto        lpos    = eval_position(sc, ldframe(inst.frame), sh, e, uv);
    return normalize(lpos - position);
  } else if (light.environment != YTHIP_INVALIDID) {
    const ythip_environment environment = load_record(sc.environments, light.environment);
    if (environment.emission_tex != YTHIP_INVALIDID) {
      const ythip_texture tex = load_record(sc.textures, environment.emission_tex);
      auto        idx = sample_discrete(sc.cdf + light.cdf_offset, light.cdf_count, rel);
      auto        uv  = vec2f{div_((idx % tex.width) + 0.5f, (float)tex.width), div_((idx / tex.width) + 0.5f, (float)tex.height)};
      float sx, cx, sy, cy;
      ytm::sincosf(uv.x * 2 * pif, &sx, &cx);
      ytm::sincosf(uv.y * pif, &sy, &cy);
      return transform_direction(ldframe(environment.frame), {cx * sy, cy, sx * sy});
    } else {
      return sample_sphere(ruv);
    }
  }
  return {0, 0, 0};
}

// sample_lights_pdf — yocto_trace.cpp:391-443.  WALK: 0 = no instance lights in
// the scene (no traversal code at all), 2 = the instance walks inline (the walk stage of k_trace).
template <int WALK, bool COUNT = true, int TRI = 0>
YT_FN float sample_lights_pdf(const DScene& sc, vec3f position, vec3f direction, Stack* st, Counters* cnt) {
  auto pdf = 0.0f;
  for (int l = 0; l < sc.num_lights; l++) {
    const auto light = SCALAR_LOADS ? ldc_record(sc.lights + l) : sc.lights[l];  // (l is wave-uniform: one scalar fetch)
    if (light.instance != YTHIP_INVALIDID) {
      if constexpr (WALK != 0) {
        // (the light is the same for every lane: its instance and shape records through the scalar cache)
        const auto inst           = SCALAR_LOADS ? ldc_record(sc.instances + light.instance) : sc.instances[light.instance];
        const auto sh             = SCALAR_LOADS ? ldc_record(sc.shapes + inst.shape) : sc.shapes[inst.shape];
        auto        frame         = ldframe(inst.frame);
        auto        lpdf          = 0.0f;
        auto        next_position = position;
        auto        area          = sc.cdf[light.cdf_offset + light.cdf_count - 1];
        for (auto bounce = 0; bounce < 100; bounce++) {
          ray3f ray  = make_ray(next_position, direction);
#ifdef YT_OWN_TREE
          Hit   isec = traverse_own<TRI>(sc, ray, light.instance, *st, *cnt);
#else
          Hit   isec = traverse<COUNT, false, TRI>(sc, ray, light.instance, false, *st, *cnt);  // (counts only in the counting launch)
#endif
          if (!isec.hit) break;
          vec3f lposition, lnormal;
          if constexpr (TRI == 1 && YT_LEAF_SHADE) {  // every shape a triangle mesh: the light's triangle as the walk read it
            const TriPos tp = load_tripos(sc, isec.leaf);
            lposition       = eval_position(sc, frame, sh, elem4{0, 0, 0, 0}, {isec.u, isec.v}, &tp);
            lnormal         = eval_element_normal(sc, frame, sh, elem4{0, 0, 0, 0}, &tp);
          } else {
            auto e    = load_element(sc, sh, isec.element);
            lposition = eval_position(sc, frame, sh, e, {isec.u, isec.v});
            lnormal   = eval_element_normal(sc, frame, sh, e);
          }
          lpdf += div_(distance_squared(lposition, position), fabs_(dot(lnormal, direction)) * area);
          next_position = lposition + direction * 1e-3f;
        }
        pdf += lpdf;
      }
    } else if (light.environment != YTHIP_INVALIDID) {
      const auto environment = SCALAR_LOADS ? ldc_record(sc.environments + light.environment) : sc.environments[light.environment];
      if (environment.emission_tex != YTHIP_INVALIDID) {
        const ythip_texture tex = SCALAR_LOADS ? ldc_record(sc.textures + environment.emission_tex) : ld_record(sc.textures + environment.emission_tex);
        auto        wl       = transform_direction(ldframe(sc.env_inv + 12 * light.environment), direction);
        auto        texcoord = vec2f{div_(ytm::atan2f(wl.z, wl.x), 2 * pif), div_(ytm::acosf(clamp_(wl.y, -1.0f, 1.0f)), pif)};
        if (texcoord.x < 0) texcoord.x += 1;
        auto i     = clamp_((int)(texcoord.x * tex.width), 0, tex.width - 1);
        auto j     = clamp_((int)(texcoord.y * tex.height), 0, tex.height - 1);
        auto cdf   = sc.cdf + light.cdf_offset;
        auto prob  = div_(sample_discrete_pdf(cdf, j * tex.width + i), cdf[light.cdf_count - 1]);
        auto angle = div_(2 * pif, (float)tex.width) * div_(pif, (float)tex.height) * ytm::sinf(div_(pif * (j + 0.5f), (float)tex.height));
        pdf += div_(prob, angle);
      } else {
        pdf += 1 / (4 * pif);
      }
    }
  }
  pdf *= sample_uniform_pdf(sc.num_lights);
  return pdf;
}

// ---------------------------------------------------------------------------
// One path, registers
// ---------------------------------------------------------------------------
struct Path {
  vec3f     o, d;         // ray
  Hit       isec;         // intersection for this iteration
  vec3f     weight, radiance;
  float     max_roughness;
  int       bounce, opbounce, flags, sidx;
  int       pix;    // pixel index of the slot in the slice's trace_state arrays
  int       vslot;  // the slot of the tile grid whose pixel this is (the path slot itself, unless the pixel pool dealt it)
  rng_state rng;
};

// What the loop body decided
enum {
  STEP_END   = 0,
  STEP_NEXT  = 1,  // bounce++
  STEP_RETRY = 2,  // opacity: same bounce
  STEP_DEFER = 3,  // light pdf pending: k_lightpdf finishes the iteration
};

struct ShadeEnv {
  const DScene&  sc;
  const DState&  st;
  const KParams& kp;
  int            slot;
#ifdef YT_TIMING
  long long t_geo = 0;  // cycle counter after the shading point has been evaluated
#endif
};

YT_FN volume_point load_volume(const DState& s, int slot) {
  float4 a = s.vol_a[slot], b = s.vol_b[slot];
  return {{a.x, a.y, a.z}, {a.w, b.x, b.y}, b.z};
}
YT_FN void store_volume(const DState& s, int slot, const material_point& m) {
  s.vol_a[slot] = {m.density.x, m.density.y, m.density.z, m.scattering.x};
  s.vol_b[slot] = {m.scattering.y, m.scattering.z, m.scanisotropy, 0};
}
// trace_result.albedo / .normal of the sample = those of its first surface hit.
// The running means of trace_state.albedo / .normal (yocto_trace.cpp:1477-1481)
// take exactly one term per sample, so the term is folded in right here, with
// the same lerp and weight trace_sample's tail uses, instead of being parked in
// per-slot arrays until the sample ends (saves 72 B of state traffic per sample).
YT_FN void set_first_hit(const DState& s, const Path& P, vec3f albedo, vec3f normal) {
  auto weight = rcp_((float)(s.sample_base + P.sidx + 1));
  auto alb    = lerp_(ld3(s.albedo, P.pix), albedo, weight);
  auto nrm    = lerp_(ld3(s.normal, P.pix), normal, weight);
  s.albedo[3 * P.pix] = alb.x, s.albedo[3 * P.pix + 1] = alb.y, s.albedo[3 * P.pix + 2] = alb.z;
  s.normal[3 * P.pix] = nrm.x, s.normal[3 * P.pix + 1] = nrm.y, s.normal[3 * P.pix + 2] = nrm.z;
}
YT_FN void count_shade(const DState& s) { count_lanes(s.counters, CNT_SHADES); }

// emission seen along `incoming` from a NEE ray's intersection
// (yocto_trace.cpp:678-687, 873-884).  CLS: the scene class (step_path)
template <int CLS = 0>
YT_FN vec3f nee_emission(const DScene& sc, const Hit& isec, vec3f incoming) {
  constexpr bool NOTEX = CLS == 1 || CLS == 2, OPAQUE = CLS == 3;
  constexpr int  PRIMS = CLS == 1 ? 1 : (OPAQUE ? 2 : 0);
  if (!isec.hit) return eval_environment(sc, incoming);
  auto s        = load_surface<PRIMS>(sc, isec.instance, isec.element, {isec.u, isec.v});
  auto material = eval_material<NOTEX, OPAQUE>(sc, s.shc, s.mat, s.e, s.uv);
  auto normal   = eval_shading_normal<NOTEX>(sc, s.frame, s.shc, s.mat, s.e, s.uv, -incoming);
  return eval_emission(material, normal, -incoming);
}

// The end of one loop-body iteration, after the weight update:
// weight check + russian roulette (yocto_trace.cpp:584-592).
YT_FN int step_tail(Path& P) {
  if (P.weight == vec3f{0, 0, 0} || !isfinite_(P.weight)) return STEP_END;
  if (P.bounce > 3) {
    auto rr_prob = min_((float)0.99, max_(P.weight));
    if (rand1f(P.rng) >= rr_prob) return STEP_END;
    P.weight *= rcp_(rr_prob);
  }
  return STEP_NEXT;
}

// ---------------------------------------------------------------------------
// trace_path / trace_pathdirect / trace_pathmis / trace_pathtest — one iteration
// of the bounce loop after the intersection (yocto_trace.cpp:453-1029)
// ---------------------------------------------------------------------------
// CLS: what is known about the resident scene (checked at upload; the assignments below only
// tell the compiler, the arithmetic on the live path is the same): 0 nothing; 1 "simple scene" —
// every material matte and untextured, every shape a triangle mesh; 2 no material references
// a texture (any material types, any shapes: the hair scene of configs[4]); 3 "opaque textured" —
// every material matte, glossy or reflective (nothing transmits: no volumes), textures only in the
// color and normal slots, triangle and quad meshes only: the scenes of the reference's own test
// corpus (materials1 / materials3 / shapes1 / instances1 / arealights1 / environments1).
// LEAFPOS (class 1 of the fused kernel, where the hit record comes straight from the walk): the shading point's position and
// geometric normal from the hit's leaf record (yt_scene.h: TriPos)
template <int SAMPLER, int LP, int CLS = 0, bool LEAFPOS = false>
YT_FN int step_path(ShadeEnv& E, Path& P) {
  static_assert(!LEAFPOS || CLS == 1, "leaf-record shading: triangle scenes");
  constexpr bool MATTE = CLS == 1, NOTEX = CLS == 1 || CLS == 2, OPAQUE = CLS == 3;
  constexpr int  PRIMS = MATTE ? 1 : (OPAQUE ? 2 : 0);
  const auto& sc = E.sc;
  const auto& kp = E.kp;
  constexpr bool DIRECT  = SAMPLER == YTHIP_SAMPLER_PATHDIRECT;
  constexpr bool MIS     = SAMPLER == YTHIP_SAMPLER_PATHMIS;
  constexpr bool TEST    = SAMPLER == YTHIP_SAMPLER_PATHTEST;
  constexpr bool VOLUMES = !TEST && !MATTE && !OPAQUE;
  static_assert(!(DIRECT || MIS) || LP == LP_DEFER, "the NEE samplers run their walks in the walk stage");
  const bool next_emission = !(P.flags & PF_NOEMIT);

  auto& isec = P.isec;
  if (!isec.hit) {
    if ((P.bounce > 0 || !kp.envhidden) && ((!DIRECT && !MIS) || next_emission))
      P.radiance += P.weight * eval_environment(sc, P.d);
    return STEP_END;
  }

  // handle transmission if inside a volume
  auto in_volume = false;
  volume_point vsdf;
  if (VOLUMES && (P.flags & PF_VOLUME)) {
    vsdf          = load_volume(E.st, E.slot);
    auto rd       = rand1f(P.rng);  // g++ order: rd, then rl
    auto rl       = rand1f(P.rng);
    auto distance = sample_transmittance(vsdf.density, isec.distance, rl, rd);
    P.weight *= eval_transmittance(vsdf.density, distance) /
                sample_transmittance_pdf(vsdf.density, distance, isec.distance);
    in_volume     = distance < isec.distance;
    isec.distance = distance;
  }

  if (!in_volume) {
    // prepare shading point
    auto outgoing = -P.d;
    auto s        = load_surface<PRIMS, LEAFPOS>(sc, isec.instance, isec.element, {isec.u, isec.v});
    TriPos        tri;
    const TriPos* tp = nullptr;
    if constexpr (LEAFPOS) {
      // (a mesh with vertex normals fetches its indices anyway: three more 16-B loads would only add to them — configs[3] -0.5 %)
      if (s.shc.normals < 0 && s.shc.colors < 0) tri = load_tripos(sc, isec.leaf), tp = &tri;
    }
    auto position = eval_shading_position(sc, s.frame, s.shc, s.e, s.uv, tp);
    auto normal   = eval_shading_normal<NOTEX>(sc, s.frame, s.shc, s.mat, s.e, s.uv, outgoing, tp);
    auto material = eval_material<NOTEX, OPAQUE>(sc, s.shc, s.mat, s.e, s.uv);
    count_shade(E.st);
#ifdef YT_TIMING
    asm volatile("" ::"v"(position.x), "v"(normal.x), "v"(material.color.x), "v"(material.roughness));
    E.t_geo = __builtin_readcyclecounter();
#endif
    if (TEST) material.type = YTHIP_MATTE;
    // MATTE (the "simple scene" variant): every material of the resident scene is matte
    // and untextured and every shape a triangle mesh (checked at upload), so this
    // assignment changes nothing — it tells the compiler, which then
    // drops the other seven lobes, the volume and texture code and the registers they pin
    if (MATTE) material.type = YTHIP_MATTE;

    // correct roughness
    if (!TEST && kp.nocaustics) {
      P.max_roughness    = max_(material.roughness, P.max_roughness);
      material.roughness = P.max_roughness;
    }

    // handle opacity
    if (!TEST && material.opacity < 1 && rand1f(P.rng) >= material.opacity) {
      if (P.opbounce++ > 128) return STEP_END;
      P.o = position + P.d * 1e-2f;
      return STEP_RETRY;
    }

    // set hit variables
    if (P.bounce == 0) {
      P.flags |= PF_HIT;
      set_first_hit(E.st, P, material.color, normal);
    }

    // accumulate emission
    if ((!DIRECT && !MIS) || next_emission) P.radiance += P.weight * eval_emission(material, normal, outgoing);

    // direct (pathdirect) — yocto_trace.cpp:670-693
    int nee_pending = 0;  // pathdirect with deferred NEE (LP_DEFER): the NEE ray waits for the walk stage
    if constexpr (DIRECT) {
      if (!is_delta(material)) {
        auto ruv      = rand2f(P.rng);  // g++ order: ruv, rel, rl
        auto rel      = rand1f(P.rng);
        auto rl       = rand1f(P.rng);
        auto incoming = sample_lights(sc, position, rl, rel, ruv);
        {
          // the light pdf (walks), the NEE ray and its emission have no rng draws and touch
          // nothing but P.radiance: they run in the walk stage, before the weight update
          auto bsdfcos       = eval_bsdfcos(material, normal, outgoing, incoming);
          E.st.nee_a[E.slot] = {incoming.x, incoming.y, incoming.z, __int_as_float(1)};
          E.st.nee_b[E.slot] = {bsdfcos.x, bsdfcos.y, bsdfcos.z, 0};
          nee_pending        = 1;
        }
        P.flags |= PF_NOEMIT;
      } else {
        P.flags &= ~PF_NOEMIT;
      }
    }

    // next direction
    auto incoming = vec3f{0, 0, 0};
    bool deferred = false;
    if (!is_delta(material)) {
      if constexpr (MIS) {
        // direct with MIS — yocto_trace.cpp:853-892 — with the walks deferred: both directions are
        // drawn and their lobe terms evaluated here (all the rng draws, in order); the light pdfs,
        // the two NEE rays, next_intersection and the emission run in the walk stage
        float4 ma = {0, 0, 0, 0}, mb = {0, 0, 0, 0}, mc = {0, 0, 0, 0}, md = {0, 0, 0, 0};
        int    mflags = 0;
        for (int pass = 0; pass < 2; pass++) {
          if (pass == 0) {
            auto ruv = rand2f(P.rng);
            auto rel = rand1f(P.rng);
            auto rl  = rand1f(P.rng);
            incoming = sample_lights(sc, position, rl, rel, ruv);
          } else {
            auto rn  = rand2f(P.rng);  // g++ order: rn, rnl
            auto rnl = rand1f(P.rng);
            incoming = sample_bsdfcos(material, normal, outgoing, rnl, rn);
          }
          if (incoming == vec3f{0, 0, 0}) break;
          auto lobe     = eval_lobe(material, normal, outgoing, incoming);
          auto bsdfcos  = lobe.f;
          auto bsdf_pdf = lobe.pdf;
          if (pass == 0)
            ma = {incoming.x, incoming.y, incoming.z, bsdf_pdf}, mb = {bsdfcos.x, bsdfcos.y, bsdfcos.z, 0};
          else
            mc = {incoming.x, incoming.y, incoming.z, bsdf_pdf}, md = {bsdfcos.x, bsdfcos.y, bsdfcos.z, 0};
          mflags |= 1 << pass;
        }
        // indirect: the factor of the weight update, applied after the NEE contributions
        auto wl   = eval_lobe(material, normal, outgoing, incoming);
        auto wfac = wl.f / wl.pdf;
        mb.w               = __int_as_float(mflags);
        E.st.nee_a[E.slot] = ma, E.st.nee_b[E.slot] = mb, E.st.nee_c[E.slot] = mc, E.st.nee_d[E.slot] = md;
        E.st.nee_e[E.slot] = {wfac.x, wfac.y, wfac.z, 0};
        deferred           = true;
        P.flags |= PF_NOEMIT;
      } else {
        if (rand1f(P.rng) < 0.5f) {
          auto rn  = rand2f(P.rng);
          auto rnl = rand1f(P.rng);
          incoming = sample_bsdfcos(material, normal, outgoing, rnl, rn);
        } else {
          auto ruv = rand2f(P.rng);
          auto rel = rand1f(P.rng);
          auto rl  = rand1f(P.rng);
          incoming = sample_lights(sc, position, rl, rel, ruv);
        }
        if (incoming == vec3f{0, 0, 0}) {
          if (DIRECT && nee_pending) {  // the path ends, its NEE contribution is still owed
            E.st.nee_a[E.slot].w = __int_as_float(2);
            P.o                  = position;
            return STEP_DEFER;
          }
          return STEP_END;
        }
        auto lobe  = eval_lobe(material, normal, outgoing, incoming);  // f and its pdf from one evaluation (yt_shading.h)
        auto f     = lobe.f;
        auto pdf_a = lobe.pdf;
        if constexpr (LP == LP_DEFER) {
          E.st.pend[E.slot] = {f.x, f.y, f.z, pdf_a};
          deferred          = true;
        } else {
          P.weight *= f / (0.5f * pdf_a + 0.5f * sample_lights_pdf<0>(sc, position, incoming, nullptr, nullptr));
        }
      }
    } else {
      incoming = sample_delta(material, normal, outgoing, rand1f(P.rng));
      if (DIRECT && incoming == vec3f{0, 0, 0}) return STEP_END;
      P.weight *= eval_delta(material, normal, outgoing, incoming) /
                  sample_delta_pdf(material, normal, outgoing, incoming);
      if (MIS) P.flags &= ~PF_NOEMIT;
    }

    // update volume stack
    if (VOLUMES && is_volumetric(s.mat) && dot(normal, outgoing) * dot(normal, incoming) < 0) {
      if (!(P.flags & PF_VOLUME)) {
        auto vmat = eval_material<NOTEX, OPAQUE>(sc, s.shc, s.mat, s.e, s.uv);
        store_volume(E.st, E.slot, vmat);
        P.flags |= PF_VOLUME;
      } else {
        P.flags &= ~PF_VOLUME;
      }
    }

    // setup next iteration
    P.o = position;
    P.d = incoming;
    if (deferred) return STEP_DEFER;
  } else {
    // volume scattering event
    auto outgoing = -P.d;
    auto position = P.o + P.d * isec.distance;
    auto incoming = vec3f{0, 0, 0};
    if (rand1f(P.rng) < 0.5f) {
      auto rn  = rand2f(P.rng);
      auto rnl = rand1f(P.rng);
      incoming = sample_scattering(vsdf, outgoing, rnl, rn);
      if (MIS) P.flags &= ~PF_NOEMIT;
    } else {
      auto ruv = rand2f(P.rng);
      auto rel = rand1f(P.rng);
      auto rl  = rand1f(P.rng);
      incoming = sample_lights(sc, position, rl, rel, ruv);
      if (MIS) P.flags &= ~PF_NOEMIT;
    }
    if (!MIS && incoming == vec3f{0, 0, 0}) return STEP_END;
    auto medium = eval_medium(vsdf, outgoing, incoming);
    auto f      = medium.f;
    auto pdf_a  = medium.pdf;
    P.o        = position;
    P.d        = incoming;
    if constexpr (LP == LP_DEFER) {
      E.st.pend[E.slot] = {f.x, f.y, f.z, pdf_a};
      if constexpr (DIRECT) E.st.nee_a[E.slot].w = __int_as_float(0);  // no NEE at a scattering event
      if constexpr (MIS) E.st.nee_b[E.slot].w = __int_as_float(4);     // the mixture terms of `pend` apply
      return STEP_DEFER;
    } else {
      P.weight *= f / (0.5f * pdf_a + 0.5f * sample_lights_pdf<0>(sc, position, incoming, nullptr, nullptr));
    }
  }
  return step_tail(P);
}

// ---------------------------------------------------------------------------
// trace_naive / trace_furnace — yocto_trace.cpp:1032-1108, 1247-1338
// ---------------------------------------------------------------------------
template <int SAMPLER>
YT_FN int step_naive(ShadeEnv& E, Path& P) {
  const auto& sc = E.sc;
  const auto& kp = E.kp;
  constexpr bool FURNACE = SAMPLER == YTHIP_SAMPLER_FURNACE;
  auto&          isec    = P.isec;
  if (!isec.hit) {
    if (P.bounce > 0 || !kp.envhidden) P.radiance += P.weight * eval_environment(sc, P.d);
    return STEP_END;
  }
  auto outgoing = -P.d;
  auto s        = load_surface(sc, isec.instance, isec.element, {isec.u, isec.v});
  // furnace uses eval_position (instance transform always); naive eval_shading_position
  auto position = FURNACE ? eval_position(sc, s.frame, s.shc, s.e, s.uv)
                          : eval_shading_position(sc, s.frame, s.shc, s.e, s.uv);
  auto normal   = eval_shading_normal(sc, s.frame, s.shc, s.mat, s.e, s.uv, outgoing);
  auto material = eval_material(sc, s.shc, s.mat, s.e, s.uv);
  count_shade(E.st);

  if (material.opacity < 1 && rand1f(P.rng) >= material.opacity) {
    if (P.opbounce++ > 128) return STEP_END;
    P.o = position + P.d * 1e-2f;
    return STEP_RETRY;
  }
  if (P.bounce == 0) {
    P.flags |= PF_HIT;
    set_first_hit(E.st, P, material.color, normal);
  }
  P.radiance += P.weight * eval_emission(material, normal, outgoing);

  auto incoming = vec3f{0, 0, 0};
  if (material.roughness != 0) {
    auto rn  = rand2f(P.rng);
    auto rnl = rand1f(P.rng);
    incoming = sample_bsdfcos(material, normal, outgoing, rnl, rn);
    if (incoming == vec3f{0, 0, 0}) return STEP_END;
    auto lobe = eval_lobe(material, normal, outgoing, incoming);
    P.weight *= lobe.f / lobe.pdf;
  } else {
    incoming = sample_delta(material, normal, outgoing, rand1f(P.rng));
    if (incoming == vec3f{0, 0, 0}) return STEP_END;
    P.weight *= eval_delta(material, normal, outgoing, incoming) /
                sample_delta_pdf(material, normal, outgoing, incoming);
  }
  int tail = step_tail(P);
  if (tail == STEP_END) return STEP_END;
  if (FURNACE) {
    if (dot(normal, outgoing) * dot(normal, incoming) < 0) P.flags ^= PF_INVOL;
  }
  P.o = position;
  P.d = incoming;
  return STEP_NEXT;
}

// ---------------------------------------------------------------------------
// trace_eyelight / trace_diagram — yocto_trace.cpp:1111-1244
// ---------------------------------------------------------------------------
template <int SAMPLER>
YT_FN int step_eyelight(ShadeEnv& E, Path& P) {
  const auto& sc = E.sc;
  const auto& kp = E.kp;
  constexpr bool DIAGRAM = SAMPLER == YTHIP_SAMPLER_DIAGRAM;
  auto&          isec    = P.isec;
  if (!isec.hit) {
    if (DIAGRAM) {
      P.radiance += P.weight * vec3f{1, 1, 1};
      // hit = true; albedo/normal stay as recorded at bounce 0 (zero if never hit)
      if (!(P.flags & PF_HIT)) set_first_hit(E.st, P, {0, 0, 0}, {0, 0, 0});
      P.flags |= PF_HIT;
    } else if (P.bounce > 0 || !kp.envhidden) {
      P.radiance += P.weight * eval_environment(sc, P.d);
    }
    return STEP_END;
  }
  auto outgoing = -P.d;
  auto s        = load_surface(sc, isec.instance, isec.element, {isec.u, isec.v});
  auto position = eval_shading_position(sc, s.frame, s.shc, s.e, s.uv);
  auto normal   = eval_shading_normal(sc, s.frame, s.shc, s.mat, s.e, s.uv, outgoing);
  auto material = eval_material(sc, s.shc, s.mat, s.e, s.uv);
  count_shade(E.st);

  if (material.opacity < 1 && rand1f(P.rng) >= material.opacity) {
    if (P.opbounce++ > 128) return STEP_END;
    P.o = position + P.d * 1e-2f;
    return STEP_RETRY;
  }
  if (P.bounce == 0) {
    P.flags |= PF_HIT;
    set_first_hit(E.st, P, material.color, normal);
  }
  auto incoming = outgoing;
  P.radiance += P.weight * eval_emission(material, normal, outgoing);
  P.radiance += P.weight * pif * eval_bsdfcos(material, normal, outgoing, incoming);

  if (!is_delta(material)) return STEP_END;
  incoming = sample_delta(material, normal, outgoing, rand1f(P.rng));
  if (incoming == vec3f{0, 0, 0}) return STEP_END;
  P.weight *= eval_delta(material, normal, outgoing, incoming) /
              sample_delta_pdf(material, normal, outgoing, incoming);
  if (P.weight == vec3f{0, 0, 0} || !isfinite_(P.weight)) return STEP_END;
  P.o = position;
  P.d = incoming;
  return STEP_NEXT;
}

// ---------------------------------------------------------------------------
// trace_falsecolor — yocto_trace.cpp:1341-1419
// ---------------------------------------------------------------------------
YT_FN vec3f hashed_color(int id) {
  // std::hash<int> is the identity in libstdc++: (size_t)id, sign-extended
  auto hashed = (uint64_t)(int64_t)id;
  auto rng    = make_rng(961748941ull, hashed);
  auto r      = rand3f(rng);
  auto c      = 0.5f + 0.5f * r;
  return {ytm::powf(c.x, 2.2f), ytm::powf(c.y, 2.2f), ytm::powf(c.z, 2.2f)};
}
YT_FN int step_falsecolor(ShadeEnv& E, Path& P) {
  const auto& sc   = E.sc;
  auto&       isec = P.isec;
  if (!isec.hit) return STEP_END;  // trace_result{}: radiance 0, hit false
  auto outgoing = -P.d;
  auto s        = load_surface(sc, isec.instance, isec.element, {isec.u, isec.v});
  auto position = eval_shading_position(sc, s.frame, s.shc, s.e, s.uv);
  auto normal   = eval_shading_normal(sc, s.frame, s.shc, s.mat, s.e, s.uv, outgoing);
  auto gnormal  = eval_element_normal(sc, s.frame, s.shc, s.e);
  auto texcoord = eval_texcoord(sc, s.shc, s.e, s.uv);
  auto material = eval_material(sc, s.shc, s.mat, s.e, s.uv);
  auto delta    = is_delta(material) ? 1.0f : 0.0f;
  count_shade(E.st);
  const auto& inst = sc.instances[isec.instance];

  auto result = vec3f{0, 0, 0};
  switch (E.kp.falsecolor) {
    case YTHIP_FC_POSITION: result = position * 0.5f + 0.5f; break;
    case YTHIP_FC_NORMAL: result = normal * 0.5f + 0.5f; break;
    case YTHIP_FC_FRONTFACING: result = dot(normal, -P.d) > 0 ? vec3f{0, 1, 0} : vec3f{1, 0, 0}; break;
    case YTHIP_FC_GNORMAL: result = gnormal * 0.5f + 0.5f; break;
    case YTHIP_FC_GFRONTFACING: result = dot(gnormal, -P.d) > 0 ? vec3f{0, 1, 0} : vec3f{1, 0, 0}; break;
    case YTHIP_FC_MTYPE: result = hashed_color(material.type); break;
    case YTHIP_FC_TEXCOORD: result = {fmodf(texcoord.x, 1.0f), fmodf(texcoord.y, 1.0f), 0}; break;
    case YTHIP_FC_COLOR: result = material.color; break;
    case YTHIP_FC_EMISSION: result = material.emission; break;
    case YTHIP_FC_ROUGHNESS: result = {material.roughness, material.roughness, material.roughness}; break;
    case YTHIP_FC_OPACITY: result = {material.opacity, material.opacity, material.opacity}; break;
    case YTHIP_FC_METALLIC: result = {material.metallic, material.metallic, material.metallic}; break;
    case YTHIP_FC_DELTA: result = {delta, delta, delta}; break;
    case YTHIP_FC_ELEMENT: result = hashed_color(isec.element); break;
    case YTHIP_FC_INSTANCE: result = hashed_color(isec.instance); break;
    case YTHIP_FC_SHAPE: result = hashed_color(inst.shape); break;
    case YTHIP_FC_MATERIAL: result = hashed_color(inst.material); break;
    case YTHIP_FC_HIGHLIGHT: {
      if (material.emission == vec3f{0, 0, 0}) material.emission = {0.2f, 0.2f, 0.2f};
      result = material.emission * fabs_(dot(-P.d, normal));
    } break;
    default: result = {0, 0, 0};
  }
  P.radiance = srgb_to_rgb(result);
  P.flags |= PF_HIT;
  set_first_hit(E.st, P, material.color, normal);
  return STEP_END;
}

// ===========================================================================
// Path slot I/O, accumulation, regeneration, compaction
// ===========================================================================

// Path state of the YT_BLOCK slots of a tile, resident in LDS for the whole batch
// (SoA of 16-B records: ds_read/write_b128).  Slots change threads at every
// compaction, so the state cannot stay in registers; keeping it in LDS instead
// of HBM removes 160 B of global traffic per path per bounce.  The pixel's PCG
// state travels here too and returns to trace_state.rngs when the pixel has
// taken its last sample of the batch.
struct WgState {
  float4     ray_a[YT_BLOCK];  // o.xyz, d.x
  float4     ray_b[YT_BLOCK];  // d.y, d.z, bounce, flags|opbounce<<8
  float4     wgt[YT_BLOCK];    // weight.xyz, max_roughness
  float4     rad[YT_BLOCK];    // radiance.xyz, samples done in this batch (int)
  ulonglong2 rng[YT_BLOCK];    // rng_state {state, inc}
  int        vslot[YT_BLOCK];  // Path::vslot
};

// everything but the ray, whose second record `rb` the caller already holds
YT_FN void load_path_rest(const DState& st, const WgState& W, int slot, Path& P, float4 rb) {
  const int l = slot & (YT_BLOCK - 1);
  float4    w = W.wgt[l], r = W.rad[l];
  auto      g = W.rng[l];
  int       pi, pj;
  P.vslot         = W.vslot[l];
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
YT_FN void load_path(const DState& st, const WgState& W, int slot, Path& P) {
  const int l  = slot & (YT_BLOCK - 1);
  float4    ra = W.ray_a[l], rb = W.ray_b[l];
  P.o = {ra.x, ra.y, ra.z};
  P.d = {ra.w, rb.x, rb.y};
  load_path_rest(st, W, slot, P, rb);
}

YT_FN void store_path(WgState& W, int slot, const Path& P) {
  const int l = slot & (YT_BLOCK - 1);
  W.rng[l]   = {P.rng.state, P.rng.inc};
  W.ray_a[l] = {P.o.x, P.o.y, P.o.z, P.d.x};
  W.ray_b[l] = {P.d.y, P.d.z, __int_as_float(P.bounce), __int_as_float(P.flags | (P.opbounce << 8))};
  W.wgt[l]   = {P.weight.x, P.weight.y, P.weight.z, P.max_roughness};
  W.rad[l]   = {P.radiance.x, P.radiance.y, P.radiance.z, __int_as_float(P.sidx)};
  W.vslot[l] = P.vslot;
}

// Head of trace_sample (yocto_trace.cpp:1464-1468): the pixel's next camera ray.
YT_FN void start_sample(const DScene& sc, const DState& st, const KParams& kp, int slot, Path& P) {
  int i, j;
  slot_pixel(st, P.vslot, i, j);
  // sample_camera(camera, ij, size, puv = rand2f, luv = rand2f, tent): g++ draws luv first
  auto luv = rand2f(P.rng);
  auto puv = rand2f(P.rng);
  // (the camera record is the same for every lane: through the scalar cache)
  const auto camera = SCALAR_LOADS ? ldc_record(sc.cameras + kp.camera) : sc.cameras[kp.camera];
  auto ray = sample_camera(camera, i, j, st.width, st.height, puv, luv, kp.tentfilter != 0);
  P.o = ray.o, P.d = ray.d;
  P.weight        = {1, 1, 1};
  P.radiance      = {0, 0, 0};
  P.max_roughness = 0;
  P.bounce = 0, P.opbounce = 0, P.flags = 0;
  if (st.nhit_a) {
    st.nhit_a[slot] = {0, 0, 0, __int_as_float(-1)};
    st.nhit_e[slot] = -1;
  }
}

// Tail of trace_sample (yocto_trace.cpp:1471-1491).
YT_FN void finish_sample(const DState& st, const KParams& kp, int slot, const Path& P) {
  int   sample   = st.sample_base + P.sidx;
  vec3f radiance = P.radiance;
  bool  hit      = (P.flags & PF_HIT) != 0;
  if (!isfinite_(radiance)) radiance = {0, 0, 0};
  if (max_(radiance) > kp.clamp) radiance = radiance * div_(kp.clamp, max_(radiance));
  auto      weight = rcp_((float)(sample + 1));
  const int pix    = P.pix;
  float4    im     = st.image[pix];
  vec4f     image  = {im.x, im.y, im.z, im.w};
  constexpr bool store = true;
  if (hit) {
    // albedo / normal were folded in at the first hit (set_first_hit)
    image = lerp_(image, vec4f{radiance.x, radiance.y, radiance.z, 1}, weight);
    if (store) st.hits[pix] += 1;
  } else {
    // no surface was hit: the path never left the camera ray's direction
    // (opacity skips only move the origin), so -P.d is -camera_ray.d
    vec3f normal = -P.d;
    vec3f alb    = ld3(st.albedo, pix);
    vec3f nrm    = ld3(st.normal, pix);
    if (!kp.envhidden && kp.has_env) {
      image = lerp_(image, vec4f{radiance.x, radiance.y, radiance.z, 1}, weight);
      alb   = lerp_(alb, vec3f{1, 1, 1}, weight);
      nrm   = lerp_(nrm, normal, weight);
      if (store) st.hits[pix] += 1;
    } else {
      image = lerp_(image, vec4f{0, 0, 0, 0}, weight);
      alb   = lerp_(alb, vec3f{0, 0, 0}, weight);
      nrm   = lerp_(nrm, normal, weight);
    }
    if (store) {
      st.albedo[3 * pix] = alb.x, st.albedo[3 * pix + 1] = alb.y, st.albedo[3 * pix + 2] = alb.z;
      st.normal[3 * pix] = nrm.x, st.normal[3 * pix + 1] = nrm.y, st.normal[3 * pix + 2] = nrm.z;
    }
  }
  if (store) st.image[pix] = {image.x, image.y, image.z, image.w};
  count_lanes(st.counters, CNT_SAMPLES);
}

// Path outcome classes for the compaction
enum { OUT_DEAD = 0, OUT_PRIMARY = 1, OUT_BOUNCE = 2, OUT_DEFER = 3 };

// The head of intersect_scene_bvh for the path's NEXT ray: true when the walk would end
// at the root (empty scene, or the ray misses the root box — yocto_bvh.cpp:554-590 with
// the ray of make_ray: tmin 1e-4, tmax flt_max).  Exactly the test `traverse` opens
// with, on the same operands, so deciding it here or there gives the same answer.
YT_FN bool misses_scene_root(const DScene& sc, vec3f o, vec3f d) {
  if (sc.tlas_ref == REF_NONE) return true;
  const vec3f dinv = {1 / d.x, 1 / d.y, 1 / d.z};
  float       t0;
  return !(slab<false>(o, dinv, ray_eps, sc.tlas_bmin, sc.tlas_bmax, t0) && t0 <= flt_max * BBOX_K);
}

// Applies a step decision: bounce bookkeeping, end-of-sample accumulation and
// regeneration.  Returns the queue class of the slot.
//
// PEEK (trace_path / trace_pathtest / trace_naive / trace_eyelight, not the counting launch): a continuing path whose
// next ray cannot enter the scene's root box takes the miss branch of its next loop
// iteration (yocto_trace.cpp:473-477) right here instead of going through the queue,
// the traversal prologue and the shade stage again.  On open scenes (configs[1]: every
// bounce ray leaves the plane's box) that is most bounce rays: a sample then costs one
// iteration instead of two, and the end-of-sample code (running means, the next
// camera ray) runs once per iteration with every lane active instead of twice with
// 16 % + 84 % of them.  Same operations on the same operands in the same per-path
// order: results are bit-identical (tested against the counting launch's flow).
template <bool PEEK = false>
YT_FN int resolve_step(const DScene& sc, const DState& st, const KParams& kp, int slot, Path& P, int step,
    int max_bounces, bool stopped = false) {
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
  finish_sample(st, kp, slot, P);
  P.sidx += 1;
  if (P.sidx < st.batch && !stopped) {  // (cancelled: the pixel stops at this sample boundary)
    start_sample(sc, st, kp, slot, P);
    return OUT_PRIMARY;
  }
  st.rngs[P.pix] = {P.rng.state, P.rng.inc};  // the pixel's stream goes back to trace_state
  if (st.pool_next && !stopped) {
    // pixel pool: the lanes that are here together take the next queue entries with one atomic
    while (true) {
      const unsigned long long here   = __ballot(1);
      const int                lane   = (int)(threadIdx.x & 63), leader = __ffsll((long long)here) - 1;
      int                      base   = 0;
      if (lane == leader) base = atomicAdd(st.pool_next, __popcll(here));
      base        = __shfl(base, leader);
      const int q = base + __popcll(here & ((1ull << lane) - 1ull));
      if (q >= st.pool_total) break;
      int tile = q / YT_BLOCK;
      if (st.tile_perm) tile = st.tile_perm[tile];
      const int vs = tile * YT_BLOCK + (q & (YT_BLOCK - 1));
      int       i, j;
      const int pix = slot_pixel(st, vs, i, j);
      if (pix < 0) continue;  // a slot of an edge tile outside the slice
      auto r  = st.rngs[pix];
      P.vslot = vs, P.pix = pix, P.rng = {r.x, r.y}, P.sidx = 0;
      start_sample(sc, st, kp, slot, P);
      return OUT_PRIMARY;
    }
  }
  return OUT_DEAD;
}

// Workgroup-local queue of the persistent kernel (LDS).  Partition of the 256
// slots a workgroup owns: class OUT_PRIMARY from the front of `queue`,
// OUT_BOUNCE from the back, starting after the `base` entries already there;
// OUT_DEFER into `lqueue`.  Every thread of the workgroup must call it.
// Returns {primaries, bounces, deferred} (workgroup-uniform).
struct WgQueues {
  int queue[YT_BLOCK];
  int lqueue[YT_BLOCK];
  int cnt[YT_BLOCK / 64][3];
  // traversal work seen so far by this workgroup, per ray class (0 camera rays,
  // 1 bounce rays): lane-steps and rays — the scheduling signal of k_trace
  unsigned work[2], rays[2];
};
YT_FN int3 block_partition(WgQueues& Q, int slot, int cls, int2 base, bool defer_class) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long mp = __ballot(cls == OUT_PRIMARY);
  unsigned long long mb = __ballot(cls == OUT_BOUNCE);
  unsigned long long md = defer_class ? __ballot(cls == OUT_DEFER) : 0ull;
  if (lane == 0) {
    Q.cnt[wave][0] = __popcll(mp);
    Q.cnt[wave][1] = __popcll(mb);
    Q.cnt[wave][2] = __popcll(md);
  }
  __syncthreads();  // also: every thread has read its entry of the previous queue
  int offp = base.x, offb = base.y, offd = 0, totp = base.x, totb = base.y, totd = 0;
#pragma unroll
  for (int w = 0; w < YT_BLOCK / 64; w++) {
    if (w < wave) {
      offp += Q.cnt[w][0];
      offb += Q.cnt[w][1];
      offd += Q.cnt[w][2];
    }
    totp += Q.cnt[w][0];
    totb += Q.cnt[w][1];
    totd += Q.cnt[w][2];
  }
  const unsigned long long below = (1ull << lane) - 1ull;
  if (cls == OUT_PRIMARY) Q.queue[offp + __popcll(mp & below)] = slot;
  if (cls == OUT_BOUNCE) Q.queue[YT_BLOCK - 1 - (offb + __popcll(mb & below))] = slot;
  if (defer_class && cls == OUT_DEFER) Q.lqueue[offd + __popcll(md & below)] = slot;
  __syncthreads();  // queue visible; cnt[] free for the next call
  return {totp, totb, totd};
}

template <int SAMPLER>
YT_FN int max_bounces_of(const KParams& kp) {
  return (SAMPLER == YTHIP_SAMPLER_EYELIGHT || SAMPLER == YTHIP_SAMPLER_DIAGRAM) ? max_(kp.bounces, 4)
                                                                              : kp.bounces;
}

// ===========================================================================
// k_trace — the whole of trace_samples for one tile, one launch per batch.
//
// A persistent workgroup — ONE wavefront, so that no wave ever waits for a sibling
// at the two barriers of an iteration (with four waves per workgroup that wait was
// 30-40 % of the wave time on interior scenes, yt_bvh.h) — owns its tile's
// YT_BLOCK path slots and loops
//     extend (BVH traversal)  →  shade (one bounce-loop body)  →
//     [deferred light-pdf walks]  →  block-local compaction
// until every pixel of the tile has taken its `batch` samples.  Workgroups
// never wait for each other: no per-iteration launch, no grid-wide tail, and the
// tile's path state (≈40 KB) stays in the XCD's L2 between iterations.  The
// ray and the hit record never leave registers between extend and shade.
// ===========================================================================
#ifndef YT_WAVES_PER_EU  // development builds: occupancy experiments (docs/HISTORY.md)
#define YT_WAVES_PER_EU 4
#endif
template <int SAMPLER, int LP, bool COUNT, bool WIDE, int CLS = 0>
__global__ void __launch_bounds__(YT_BLOCK, YT_WAVES_PER_EU)
    k_trace(DScene sc, DState st, KParams kp) {
  constexpr bool MATTE = CLS == 1;
  constexpr int  PRIMS = MATTE ? 1 : (CLS == 3 ? 2 : 0);  // what the walks know about the shapes (yt_bvh.h: TRI)
  // majority-phase scene walk (yt_bvh.h::traverse_phased) for the kernels where it wins:
  // simple scenes with area lights (closed rooms: every ray hits, bounce rays as long as
  // camera rays).  Measured, docs/HISTORY.md.
  constexpr bool PHASED_SCENE = MATTE && LP == LP_DEFER;
  static_assert(!(COUNT && WIDE), "work counters follow the reference's binary walk");
  constexpr bool MIS = SAMPLER == YTHIP_SAMPLER_PATHMIS;
  // root-box misses of continuing paths resolved in place (resolve_step); the counting
  // launch keeps the plain flow, whose ray / node counts are the reference's
  // (trace_naive and trace_eyelight have the same miss branch: yocto_trace.cpp:1048-1052, 1127-1131)
  constexpr bool PEEK = !COUNT && (SAMPLER == YTHIP_SAMPLER_PATH || SAMPLER == YTHIP_SAMPLER_PATHTEST ||
                                      SAMPLER == YTHIP_SAMPLER_NAIVE || SAMPLER == YTHIP_SAMPLER_EYELIGHT);
  __shared__ StackEntry s_stack[YT_LDS_DEPTH][YT_BLOCK];
  __shared__ WgQueues   Q;
  __shared__ WgState    W;
  int lb = logical_block(st);
  if (lb < 0) return;
  if (stop_requested(st.stop, st.stop_gen)) return;  // cancelled before this tile started
  int vtile = lb;  // the tile whose pixels the slots start on
  if (st.pool_next) {
    // pixel pool: the workgroup's first 64 queue entries = one tile, assigned STATICALLY — workgroup b starts on entries
    // [64 b, 64 b + 64) and the queue's head starts at 64 x the number of workgroups (enqueue_samples).  (the
    // first tile used to come from the same counter the refills advance by arbitrary lane counts, so a workgroup whose
    // prologue ran after somebody's refill could start on an unaligned entry and render another workgroup's pixels.)
    if ((int)blockIdx.x * YT_BLOCK >= st.pool_total) return;
    vtile = (int)blockIdx.x;
    if (st.tile_perm) vtile = st.tile_perm[vtile];
    lb = (int)blockIdx.x;
  }
  if ((lb & 63) == 0 && relay_stop(st)) return;      // (one tile in 64 also asks the host's word: a batch cancelled before any workgroup ran)
  const long long t_tile0 = st.tile_cost ? (long long)__builtin_readcyclecounter() : 0;
  const int tid = threadIdx.x;
  Stack     stack;
  YT_STACK_INIT(stack, s_stack);
  Counters  cnt         = {0, 0, 0, 0, 0, 0, 0, 0};
  const int max_bounces = max_bounces_of<SAMPLER>(kp);

  if (tid < 2) Q.work[tid] = 0, Q.rays[tid] = 0;
  // head of the batch: the first camera ray of every pixel of the tile
  int3 n;
  {
    int slot = lb * YT_BLOCK + tid, i, j;
    int pix  = slot_pixel(st, vtile * YT_BLOCK + tid, i, j);
    if (st.only_pix >= 0 && pix != st.only_pix) pix = -1;
    if (pix >= 0) {
      Path P;
      auto r  = st.rngs[pix];
      P.rng   = {r.x, r.y};
      P.sidx  = 0;
      P.pix   = pix;
      P.vslot = vtile * YT_BLOCK + tid;
      start_sample(sc, st, kp, slot, P);
      if (max_bounces <= 0 && SAMPLER != YTHIP_SAMPLER_FALSECOLOR) {
        // the reference's bounce loop never runs (yocto_trace.cpp:466, 1045, 1260): every
        // sample is its camera-ray draws + the tail of trace_sample with radiance 0, no hit
        while (true) {
          finish_sample(st, kp, slot, P);
          P.sidx += 1;
          if (P.sidx >= st.batch) break;
          start_sample(sc, st, kp, slot, P);
        }
        st.rngs[pix] = {P.rng.state, P.rng.inc};
        pix          = -1;
      } else {
        store_path(W, slot, P);
      }
    }
    n = block_partition(Q, slot, pix >= 0 ? OUT_PRIMARY : OUT_DEAD, {0, 0}, false);
  }

  for (int iter = 0; n.x + n.y > 0; iter++) {
    // entry `tid` of the queue: primaries from the front, then bounces from the back.
    //
    // Scheduling.  An iteration lasts as long as its slowest wavefront, so one that
    // traces ANY camera ray costs a full camera-ray traversal.  Where bounce rays
    // are much cheaper than camera rays (open scenes: most bounce rays leave
    // through the first box) it pays to let the regenerated camera rays WAIT while
    // bounce rays are pending and then trace them all together: per sample one
    // expensive iteration instead of 1 + (bounces) of them.  Where bounce rays cost
    // as much (interiors) everything runs at once, which keeps the lanes full.  The
    // workgroup decides from the traversal work it has measured itself; results do
    // not depend on the decision (pixels are independent).
    // once per iteration = at most one sample late; every 64th iteration this workgroup is a relay
    const bool stopped = stop_requested(st.stop, st.stop_gen) || (((iter + lb) & 63) == 63 && relay_stop(st));
    bool wait = false;
    if (kp.hold && n.y > 0 && n.x > 0) {
      float wp = (float)Q.work[0], rp = (float)Q.rays[0], wb = (float)Q.work[1], rb_ = (float)Q.rays[1];
      wait     = rp > 0 && rb_ > 0 && 4.0f * wb * rp < wp * rb_;  // mean bounce work < 1/4 mean camera-ray work
    }
    const int held = wait ? n.x : 0;
    const int np   = n.x - held;
    int  slot = -1, cls = OUT_DEAD;
    bool run  = false;
    if (tid < np) {
      slot = Q.queue[tid], run = true;
    } else if (tid < np + n.y) {
      slot = Q.queue[YT_BLOCK - 1 - (tid - np)], run = true;
    } else if (tid < n.x + n.y) {
      slot = Q.queue[np + (tid - np - n.y)], cls = OUT_PRIMARY;  // stays queued
    }
    unsigned work = 0;  // traversal steps of this lane's ray in this iteration
#ifdef YT_TIMING  // development builds: per-wave phase times (tools/devbuild.sh NAME -DYT_TIMING)
    const long long tm0 = __builtin_readcyclecounter();
    long long       tm1 = tm0, tm2 = tm0, tmS = tm0, tmG = 0;
#endif
    if (run) {
      Path   P;
      float4 ra = W.ray_a[slot & (YT_BLOCK - 1)], rb = W.ray_b[slot & (YT_BLOCK - 1)];
      P.o      = {ra.x, ra.y, ra.z};
      P.d      = {ra.w, rb.x, rb.y};
      int fw   = __float_as_int(rb.w);
      // ---- extend: intersect_scene_bvh ------------------------------------
      if (MIS && (fw & PF_SKIPEXTEND)) {  // pathmis: intersection = next_intersection
        float4 ha   = st.nhit_a[slot];
        int    inst = __float_as_int(ha.w);
        P.isec      = {inst, st.nhit_e[slot], ha.x, ha.y, ha.z, inst >= 0};
      } else {
        ray3f          ray = make_ray(P.o, P.d);
        const unsigned s0  = cnt.steps;
        P.isec             = traverse_any<COUNT, WIDE, PRIMS, PHASED_SCENE>(sc, ray, -1, false, stack, cnt);
        work               = cnt.steps - s0 + 1;
      }
#ifdef YT_TIMING
      tm1 = __builtin_readcyclecounter();
#endif
      // ---- shade: one iteration of the integrator's bounce loop -------------
      load_path_rest(st, W, slot, P, rb);
      P.flags &= ~PF_SKIPEXTEND;
      int step;
      {
        ShadeEnv E = {sc, st, kp, slot};
        if constexpr (SAMPLER == YTHIP_SAMPLER_PATH || SAMPLER == YTHIP_SAMPLER_PATHTEST ||
                      SAMPLER == YTHIP_SAMPLER_PATHDIRECT || SAMPLER == YTHIP_SAMPLER_PATHMIS) {
          constexpr bool LEAFPOS = YT_LEAF_SHADE && CLS == 1 && !MIS;  // (pathmis re-uses hit records kept in HBM without the leaf index)
          step = step_path<SAMPLER, LP, CLS, LEAFPOS>(E, P);
          if constexpr (MIS) {
            // pathmis re-tests the SAME next_intersection against `opacity` with a fresh random
            // number until it passes (yocto_trace.cpp:794-796 with :830-836; up to 128 times):
            // the retry needs no ray, so it loops here instead of costing a wavefront iteration
            // each.  Exactly what the next iteration would do: P.o moved by the step, the
            // intersection re-read from next_intersection (the volume branch edits isec.distance).
            while (step == STEP_RETRY && (P.flags & PF_NOEMIT)) {
              float4 ha   = st.nhit_a[slot];
              int    inst = __float_as_int(ha.w);
              P.isec      = {inst, st.nhit_e[slot], ha.x, ha.y, ha.z, inst >= 0};
              step        = step_path<SAMPLER, LP, CLS>(E, P);
            }
          }
#ifdef YT_TIMING
          tmG = E.t_geo;
#endif
        } else if constexpr (SAMPLER == YTHIP_SAMPLER_NAIVE) {
          step = step_naive<SAMPLER>(E, P);
        } else if constexpr (SAMPLER == YTHIP_SAMPLER_FURNACE) {
          // exit test at the top of the loop body — yocto_trace.cpp:1263-1266
          if (P.bounce > 0 && !(P.flags & PF_INVOL)) {
            P.radiance += P.weight * eval_environment(sc, P.d);
            step = STEP_END;
          } else {
            step = step_naive<SAMPLER>(E, P);
          }
        } else if constexpr (SAMPLER == YTHIP_SAMPLER_EYELIGHT || SAMPLER == YTHIP_SAMPLER_DIAGRAM) {
          step = step_eyelight<SAMPLER>(E, P);
        } else {
          step = step_falsecolor(E, P);
        }
      }
      if (MIS && step != STEP_END && (P.flags & PF_NOEMIT)) P.flags |= PF_SKIPEXTEND;
#ifdef YT_TIMING
      tmS = __builtin_readcyclecounter();
#endif
      cls = resolve_step<PEEK>(sc, st, kp, slot, P, step, max_bounces, stopped);
      store_path(W, slot, P);
    }
#ifdef YT_TIMING
    tm2 = __builtin_readcyclecounter();
#endif
    if (kp.hold && iter < YT_PROBE_ITERS) {  // per-class totals: butterflies over the wavefront, one LDS atomic per class per wave
      const bool               prim = tid < np;
      const unsigned long long mp = __ballot(run && prim && work > 0), mb = __ballot(run && !prim && work > 0);
      unsigned                 wp = prim ? work : 0, wb = prim ? 0 : work;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) wp += __shfl_xor(wp, off), wb += __shfl_xor(wb, off);
      if ((tid & 63) == 0) {
        if (mp) atomicAdd(&Q.work[0], wp), atomicAdd(&Q.rays[0], (unsigned)__popcll(mp));
        if (mb) atomicAdd(&Q.work[1], wb), atomicAdd(&Q.rays[1], (unsigned)__popcll(mb));
      }
    }
    n = block_partition(Q, slot, cls, {0, 0}, LP == LP_DEFER);
#ifdef YT_TIMING
    {  // extend | shade | probe + partition (incl. waiting for the workgroup's slower waves)
      unsigned tm_sum = work, tm_max = work;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        tm_sum += __shfl_xor(tm_sum, off);
        unsigned o = __shfl_xor(tm_max, off);
        tm_max     = o > tm_max ? o : tm_max;
      }
      const long long tm3     = __builtin_readcyclecounter();
      const bool      any_run = __ballot(run) != 0;
#ifdef YT_TIMING_OCC
      const unsigned long long occ_walk = (unsigned long long)__popcll(__ballot(run && work > 0)), occ_run = (unsigned long long)__popcll(__ballot(run));
#endif
      if ((tid & 63) == 0 && st.counters) {
        unsigned long long* c = st.counters + cnt_bank();
        atomicAdd(&c[9], (unsigned long long)(any_run ? tm1 - tm0 : 0));
        atomicAdd(&c[10], (unsigned long long)(any_run ? tm2 - tm1 : 0));
        atomicAdd(&c[11], (unsigned long long)(tm3 - tm2));
        atomicAdd(&c[12], (unsigned long long)(any_run ? 0 : tm2 - tm0));
        atomicAdd(&c[13], 1ull);
        // lane utilisation of the traversal: sum of lane steps vs 64 x the longest lane
        atomicAdd(&c[6], (unsigned long long)tm_sum);
        atomicAdd(&c[7], (unsigned long long)tm_max * 64ull);
#ifdef YT_TIMING_OCC  // instead of the shade split: how full the wavefront is (slots with a ray), weighted by the walk / by the iteration
        atomicAdd(&c[14], occ_walk * tm_max);
        atomicAdd(&c[15], occ_run * (unsigned long long)(tm2 - tm0));
        atomicAdd(&c[8], 64ull * (unsigned long long)(tm2 - tm0));
        if (false) {
#else
        // shade split (waves whose lane 0 shaded a hit): shading point | bsdf + sampling | finish + regenerate
        if (any_run && tmG) {
#endif
          atomicAdd(&c[14], (unsigned long long)(tmG - tm1));
          atomicAdd(&c[15], (unsigned long long)(tmS - tmG));
        }
#ifndef YT_TIMING_OCC
        if (any_run) atomicAdd(&c[8], (unsigned long long)(tm2 - tmS));
#endif
      }
    }
#endif

    // ---- deferred sample_lights_pdf walks + the rest of the loop body -------
    if constexpr (LP == LP_DEFER) {
      if (n.z > 0) {  // workgroup-uniform
        slot = -1, cls = OUT_DEAD;
        if (tid < n.z) {
          slot = Q.lqueue[tid];
          Path P;
          load_path(st, W, slot, P);
          int nee = 0;
          if constexpr (SAMPLER == YTHIP_SAMPLER_PATHDIRECT) {
            // the NEE half of the loop body (yocto_trace.cpp:670-693) for the light direction
            // the shade stage drew: pdf walks, the NEE ray, the emission it finds
            float4 na = st.nee_a[slot];
            nee       = __float_as_int(na.w);
            if (nee) {
              float4 nb      = st.nee_b[slot];
              vec3f  inc     = {na.x, na.y, na.z}, bsdfcos = {nb.x, nb.y, nb.z};
              auto   pdf     = sample_lights_pdf<2, COUNT, PRIMS>(sc, P.o, inc, &stack, &cnt);
              if (bsdfcos != vec3f{0, 0, 0} && pdf > 0) {
                ray3f nray     = make_ray(P.o, inc);
                Hit   nisec    = traverse_any<COUNT, WIDE, PRIMS>(sc, nray, -1, false, stack, cnt);
                auto  emission = nee_emission<CLS>(sc, nisec, inc);
                P.radiance += P.weight * bsdfcos * emission / pdf;
              }
            }
          }
          int step = STEP_END;
          if constexpr (SAMPLER == YTHIP_SAMPLER_PATHMIS) {
            // the walk half of the MIS loop body (yocto_trace.cpp:853-892)
            const float4 mb     = st.nee_b[slot];
            const int    mflags = __float_as_int(mb.w);
            if (!(mflags & 4)) {
              nee = 2;  // (the mixture update below does not apply)
              for (int pass = 0; pass < 2; pass++) {
                if (!(mflags & (1 << pass))) break;
                const float4 a = pass == 0 ? st.nee_a[slot] : st.nee_c[slot];
                const float4 b = pass == 0 ? mb : st.nee_d[slot];
                const vec3f  inc = {a.x, a.y, a.z}, bsdfcos = {b.x, b.y, b.z};
                const float  bsdf_pdf  = a.w;
                const float  light_pdf = sample_lights_pdf<2, COUNT, PRIMS>(sc, P.o, inc, &stack, &cnt);
                auto         heur      = [](float this_pdf, float other_pdf) {
                  return div_(this_pdf * this_pdf, this_pdf * this_pdf + other_pdf * other_pdf);
                };
                const float mis_weight = pass == 0 ? div_(heur(light_pdf, bsdf_pdf), light_pdf)
                                                   : div_(heur(bsdf_pdf, light_pdf), bsdf_pdf);
                if (bsdfcos != vec3f{0, 0, 0} && mis_weight != 0) {
                  ray3f nray  = make_ray(P.o, inc);
                  Hit   nisec = traverse_any<COUNT, WIDE, PRIMS>(sc, nray, -1, false, stack, cnt);
                  if (pass == 1) {  // next_intersection = intersection (persists across bounces)
                    st.nhit_a[slot] = {nisec.u, nisec.v, nisec.distance, __int_as_float(nisec.hit ? nisec.instance : -1)};
                    st.nhit_e[slot] = nisec.element;
                  }
                  auto emission = nee_emission<CLS>(sc, nisec, inc);
                  P.radiance += P.weight * bsdfcos * emission * mis_weight;
                }
              }
              const float4 wf = st.nee_e[slot];
              P.weight *= vec3f{wf.x, wf.y, wf.z};
              step = step_tail(P);
            }
          }
          if (nee != 2) {
            float4 pd = st.pend[slot];
            // weight *= f / (0.5 * pdf_a + 0.5 * sample_lights_pdf(position, incoming))
            auto lpdf = sample_lights_pdf<2, COUNT, PRIMS>(sc, P.o, P.d, &stack, &cnt);
            P.weight *= vec3f{pd.x, pd.y, pd.z} / (0.5f * pd.w + 0.5f * lpdf);
            step = step_tail(P);
          }
          cls      = resolve_step<PEEK>(sc, st, kp, slot, P, step, max_bounces, stopped);
          store_path(W, slot, P);
        }
        // append behind what the shade stage queued
        n = block_partition(Q, slot, cls, {n.x, n.y}, false);
      }
    }
  }
  if (COUNT || LP != LP_NONE) flush_counters(st.counters, cnt);
  if (st.tile_cost && !st.pool_next && threadIdx.x == 0) {
    const long long dt = ((long long)__builtin_readcyclecounter() - t_tile0) >> 6;  // 64-cycle units fit 32 bits
    st.tile_cost[lb]   = (unsigned)(dt < 0 ? 0 : (dt > 0xffffffffll ? 0xffffffffll : dt));
  }
}

// Test/parity entries ---------------------------------------------------------
#ifndef YT_IB_WAVES  // development builds: occupancy of the traversal-only kernel (tools/traversal_occupancy.py)
#define YT_IB_WAVES 1
#endif
template <bool COUNT, bool WIDE>
__global__ void __launch_bounds__(YT_BLOCK, YT_IB_WAVES) k_intersect_batch(DScene sc, const ythip_ray* rays,
    const int* instances, long long n, int find_any, ythip_hit* hits, unsigned long long* counters) {
  __shared__ StackEntry s_stack[YT_LDS_DEPTH][YT_BLOCK];
  long long      idx = (long long)blockIdx.x * YT_BLOCK + threadIdx.x;
  Counters       cnt = {0, 0, 0, 0, 0, 0, 0, 0};
  if (idx < n) {
    Stack stack;
    YT_STACK_INIT(stack, s_stack);
    auto  r   = rays[idx];
    ray3f ray = {{r.o[0], r.o[1], r.o[2]}, {r.d[0], r.d[1], r.d[2]}, r.tmin, r.tmax};
    Hit   h   = traverse_any<COUNT, WIDE>(sc, ray, instances ? instances[idx] : -1, find_any != 0, stack, cnt);
    // scene_intersection{} defaults when missed: instance -1, element -1, uv 0, distance 0
    if (!h.hit) h = {-1, -1, 0, 0, 0, false};
    hits[idx] = {h.instance, h.element, h.u, h.v, h.distance, h.hit ? 1 : 0};
  }
  if (COUNT) flush_counters(counters, cnt);
}

#ifdef YT_MISC_KERNELS  // the plain (non-template) kernels below are compiled by ONE unit: ythip.hip defines this
// ---------------------------------------------------------------------------
// Display path: tonemap(trace_state.image) on the device, so
// a viewer downloads 4 B/pixel instead of 16 — yocto_color.h:322-364 (tonemap,
// tonemap_filmic, rgb_to_srgb) and float_to_byte (yocto_math.h).
// ---------------------------------------------------------------------------
YT_FN float rgb_to_srgb1(float rgb) {  // yocto_color.h:239-242
  return (rgb <= 0.0031308f) ? 12.92f * rgb : (1 + 0.055f) * ytm::powf(rgb, 1 / 2.4f) - 0.055f;
}
YT_FN vec3f tonemap_filmic(vec3f hdr_) {  // yocto_color.h:322-329 (accurate_fit = false)
  auto hdr = hdr_ * 0.6f;
  auto ldr = (hdr * hdr * 2.51f + hdr * 0.03f) / (hdr * hdr * 2.43f + hdr * 0.59f + 0.14f);
  return {max_(0.0f, ldr.x), max_(0.0f, ldr.y), max_(0.0f, ldr.z)};
}
YT_FN vec3f tonemap(vec3f hdr, float exposure, bool filmic, bool srgb) {  // yocto_color.h:355-361
  auto rgb = hdr;
  if (exposure != 0) rgb *= ytm::exp2f(exposure);
  if (filmic) rgb = tonemap_filmic(rgb);
  if (srgb) rgb = {rgb_to_srgb1(rgb.x), rgb_to_srgb1(rgb.y), rgb_to_srgb1(rgb.z)};
  return rgb;
}
YT_FN unsigned char float_to_byte(float a) { return (unsigned char)clamp_((int)(a * 256), 0, 255); }
__global__ void __launch_bounds__(YT_BLOCK) k_tonemap(const float4* image, int n, float exposure, int filmic,
    int srgb, float4* outf, uchar4* outb) {
  int i = blockIdx.x * YT_BLOCK + threadIdx.x;
  if (i >= n) return;
  float4 h   = image[i];
  auto   ldr = tonemap({h.x, h.y, h.z}, exposure, filmic != 0, srgb != 0);
  if (outf) outf[i] = {ldr.x, ldr.y, ldr.z, h.w};
  if (outb) outb[i] = {float_to_byte(ldr.x), float_to_byte(ldr.y), float_to_byte(ldr.z), float_to_byte(h.w)};
}

// get_albedo_image / get_normal_image (yocto_trace.cpp:1769-1791): the vec3f guide
// buffer as the vec4f image a denoiser / viewer takes, alpha 1.
__global__ void __launch_bounds__(YT_BLOCK) k_guide_image(const float* rgb, int n, float4* out) {
  int i = blockIdx.x * YT_BLOCK + threadIdx.x;
  if (i >= n) return;
  out[i] = {rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2], 1.0f};
}

__global__ void __launch_bounds__(YT_BLOCK) k_camera_rays(DScene sc, DState st, KParams kp, ythip_ray* rays) {
  int slot = blockIdx.x * YT_BLOCK + threadIdx.x;
  if (slot >= st.npix) return;
  int  il = slot % st.lwidth, j = st.row_begin + slot / st.lwidth;
  int  i  = (st.col_first + (il / YT_TILE) * st.col_stride) * YT_TILE + il % YT_TILE;
  auto r  = st.rngs[slot];
  rng_state rng = {r.x, r.y};
  auto luv = rand2f(rng);
  auto puv = rand2f(rng);
  auto ray = sample_camera(sc.cameras[kp.camera], i, j, st.width, st.height, puv, luv, kp.tentfilter != 0);
  rays[slot] = {{ray.o.x, ray.o.y, ray.o.z}, {ray.d.x, ray.d.y, ray.d.z}, ray.tmin, ray.tmax};
}
#endif  // YT_MISC_KERNELS

}  // namespace yt
#ifdef YT_FAST
#pragma clang fp contract(off)
#endif

