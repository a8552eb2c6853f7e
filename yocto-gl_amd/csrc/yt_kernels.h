// yt_kernels.h — the wavefront (streaming) path tracer: SoA path state in HBM,
// one live path per pixel, and per bounce two kernels over the compacted queue
// of live paths:  k_extend (BVH traversal)  →  k_shade (one iteration of the
// integrator loop body, wave-ballot compaction of survivors into the next
// queue).  k_generate / k_accumulate are the head and tail of trace_sample.
//
// Restates libs/yocto/yocto_trace.cpp:338-1492 (sample_camera, sample_lights,
// sample_lights_pdf, the nine integrators, trace_sample).  The rng draw order at
// every multi-argument call site is the g++ (right-to-left) order, SURVEY.md
// Appendix A-13.
#pragma once

#include "yt_bvh.h"
#include "yt_shading.h"

namespace yt {

// Path flags (low byte) | opbounce << 8
enum {
  PF_HIT        = 1,   // trace_result.hit
  PF_VOLUME     = 2,   // volume_stack non-empty
  PF_NOEMIT     = 4,   // !next_emission            (pathdirect / pathmis)
  PF_SKIPEXTEND = 8,   // pathmis: use next_intersection instead of tracing (yocto_trace.cpp:795)
  PF_INVOL      = 16,  // furnace: in_volume
};

// Device mirror of trace_state (yocto_trace.h:147-157) + wavefront path state.
struct DState {
  int width, height, row_begin, rows, npix;
  // trace_state
  float4*     image;   // vec4f
  float*      albedo;  // vec3f
  float*      normal;  // vec3f
  int*        hits;
  ulonglong2* rngs;    // rng_state {state, inc}
  // path state, one slot per pixel (SoA of 16-B records: coalesced dwordx4)
  float4* ray_a;    // o.xyz, d.x
  float4* ray_b;    // d.y, d.z, bounce, flags|opbounce<<8
  float4* hit_a;    // u, v, distance, instance (-1 = miss)
  int*    hit_e;    // element
  float4* wgt;      // weight.xyz, max_roughness
  float4* rad;      // radiance.xyz, -
  float4* first_a;  // hit_albedo.xyz, hit_normal.x
  float2* first_b;  // hit_normal.yz          (before the first hit: -camera_ray.d)
  float4* vol_a;    // volume: density.xyz, scattering.x
  float4* vol_b;    // volume: scattering.yz, scanisotropy, -
  float4* nhit_a;   // pathmis next_intersection: u, v, distance, instance
  int*    nhit_e;   // pathmis next_intersection: element
  // queues of live pixel slots
  int* queue[2];
  int* qcount;  // [2]
  // work counters (ythip_stats), may be null
  unsigned long long* counters;
};

enum { CNT_RAYS = 0, CNT_NODES, CNT_TRIS, CNT_QUADS, CNT_LINES, CNT_POINTS, CNT_INST, CNT_SHADES, CNT_SAMPLES, CNT_NUM };

struct KParams {
  int   camera, sampler, falsecolor, bounces;
  float clamp;
  int   nocaustics, envhidden, tentfilter;
  int   has_env;  // !scene.environments.empty()
};

YT_FN void flush_counters(unsigned long long* c, const Counters& cnt) {
  if (!c) return;
  atomicAdd(&c[CNT_RAYS], (unsigned long long)cnt.rays);
  atomicAdd(&c[CNT_NODES], (unsigned long long)cnt.nodes);
  atomicAdd(&c[CNT_TRIS], (unsigned long long)cnt.triangles);
  atomicAdd(&c[CNT_QUADS], (unsigned long long)cnt.quads);
  atomicAdd(&c[CNT_LINES], (unsigned long long)cnt.lines);
  atomicAdd(&c[CNT_POINTS], (unsigned long long)cnt.points);
  atomicAdd(&c[CNT_INST], (unsigned long long)cnt.instances);
}

// Out-of-line traversal used from the shading kernel (light-pdf walks and the
// NEE rays of pathdirect/pathmis) so the big kernel carries one copy.
template <bool COUNT>
__device__ __noinline__ Hit trace_ray(const DScene& sc, vec3f o, vec3f d, int only_instance, Stack& st,
    Counters& cnt) {
  ray3f ray = make_ray(o, d);
  return traverse<COUNT>(sc, ray, only_instance, false, st, cnt);
}

// ---------------------------------------------------------------------------
// shading-point helpers
// ---------------------------------------------------------------------------
struct Surface {
  frame3f               frame;
  const DShape*         sh;
  const ythip_material* mat;
  elem4                 e;
  vec2f                 uv;
};
YT_FN Surface load_surface(const DScene& sc, int instance, int element, vec2f uv) {
  const auto& inst = sc.instances[instance];
  Surface     s;
  s.frame = ldframe(inst.frame);
  s.sh    = &sc.shapes[inst.shape];
  s.mat   = &sc.materials[inst.material];
  s.e     = load_element(sc, *s.sh, element);
  s.uv    = uv;
  return s;
}

// sample_lights — yocto_trace.cpp:361-388
YT_FN vec3f sample_lights(const DScene& sc, vec3f position, float rl, float rel, vec2f ruv) {
  if (sc.num_lights <= 0) return {0, 0, 0};  // (reference: out-of-bounds read; ytrace never gets here)
  auto        light_id = sample_uniform(sc.num_lights, rl);
  const auto& light    = sc.lights[light_id];
  if (light.instance != YTHIP_INVALIDID) {
    const auto& inst    = sc.instances[light.instance];
    const auto& sh      = sc.shapes[inst.shape];
    auto        element = sample_discrete(sc.cdf + light.cdf_offset, light.cdf_count, rel);
    auto        uv      = (sh.kind_eval == KIND_TRIANGLES) ? sample_triangle(ruv) : ruv;
    auto        e       = load_element(sc, sh, element);
    auto        lpos    = eval_position(sc, ldframe(inst.frame), sh, e, uv);
    return normalize(lpos - position);
  } else if (light.environment != YTHIP_INVALIDID) {
    const auto& environment = sc.environments[light.environment];
    if (environment.emission_tex != YTHIP_INVALIDID) {
      const auto& tex = sc.textures[environment.emission_tex];
      auto        idx = sample_discrete(sc.cdf + light.cdf_offset, light.cdf_count, rel);
      auto        uv  = vec2f{((idx % tex.width) + 0.5f) / tex.width, ((idx / tex.width) + 0.5f) / tex.height};
      return transform_direction(ldframe(environment.frame),
          {cosf(uv.x * 2 * pif) * sinf(uv.y * pif), cosf(uv.y * pif), sinf(uv.x * 2 * pif) * sinf(uv.y * pif)});
    } else {
      return sample_sphere(ruv);
    }
  }
  return {0, 0, 0};
}

// sample_lights_pdf — yocto_trace.cpp:391-443
template <bool COUNT>
YT_FN float sample_lights_pdf(const DScene& sc, vec3f position, vec3f direction, Stack& st, Counters& cnt) {
  auto pdf = 0.0f;
  for (int l = 0; l < sc.num_lights; l++) {
    const auto& light = sc.lights[l];
    if (light.instance != YTHIP_INVALIDID) {
      const auto& inst          = sc.instances[light.instance];
      const auto& sh            = sc.shapes[inst.shape];
      auto        frame         = ldframe(inst.frame);
      auto        lpdf          = 0.0f;
      auto        next_position = position;
      for (auto bounce = 0; bounce < 100; bounce++) {
        auto isec = trace_ray<COUNT>(sc, next_position, direction, light.instance, st, cnt);
        if (!isec.hit) break;
        auto e         = load_element(sc, sh, isec.element);
        auto lposition = eval_position(sc, frame, sh, e, {isec.u, isec.v});
        auto lnormal   = eval_element_normal(sc, frame, sh, e);
        auto area      = sc.cdf[light.cdf_offset + light.cdf_count - 1];
        lpdf += distance_squared(lposition, position) / (fabs_(dot(lnormal, direction)) * area);
        next_position = lposition + direction * 1e-3f;
      }
      pdf += lpdf;
    } else if (light.environment != YTHIP_INVALIDID) {
      const auto& environment = sc.environments[light.environment];
      if (environment.emission_tex != YTHIP_INVALIDID) {
        const auto& tex      = sc.textures[environment.emission_tex];
        auto        wl       = transform_direction(ldframe(sc.env_inv + 12 * light.environment), direction);
        auto        texcoord = vec2f{atan2f(wl.z, wl.x) / (2 * pif), acosf(clamp_(wl.y, -1.0f, 1.0f)) / pif};
        if (texcoord.x < 0) texcoord.x += 1;
        auto i     = clamp_((int)(texcoord.x * tex.width), 0, tex.width - 1);
        auto j     = clamp_((int)(texcoord.y * tex.height), 0, tex.height - 1);
        auto cdf   = sc.cdf + light.cdf_offset;
        auto prob  = sample_discrete_pdf(cdf, j * tex.width + i) / cdf[light.cdf_count - 1];
        auto angle = (2 * pif / tex.width) * (pif / tex.height) * sinf(pif * (j + 0.5f) / tex.height);
        pdf += prob / angle;
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
  int       bounce, opbounce, flags;
  rng_state rng;
};

// What the loop body decided
enum { STEP_END = 0, STEP_NEXT = 1 /* bounce++ */, STEP_RETRY = 2 /* opacity: same bounce */ };

struct ShadeEnv {
  const DScene&  sc;
  const DState&  st;
  const KParams& kp;
  Stack&         stack;
  Counters&      cnt;
  int            slot;
};

YT_FN volume_point load_volume(const DState& s, int slot) {
  float4 a = s.vol_a[slot], b = s.vol_b[slot];
  return {{a.x, a.y, a.z}, {a.w, b.x, b.y}, b.z};
}
YT_FN void store_volume(const DState& s, int slot, const material_point& m) {
  s.vol_a[slot] = {m.density.x, m.density.y, m.density.z, m.scattering.x};
  s.vol_b[slot] = {m.scattering.y, m.scattering.z, m.scanisotropy, 0};
}
YT_FN void set_first_hit(const DState& s, int slot, vec3f albedo, vec3f normal) {
  s.first_a[slot] = {albedo.x, albedo.y, albedo.z, normal.x};
  s.first_b[slot] = {normal.y, normal.z};
}

// emission seen along `incoming` from a NEE ray's intersection
// (yocto_trace.cpp:678-687, 873-884)
YT_FN vec3f nee_emission(const DScene& sc, const Hit& isec, vec3f incoming) {
  if (!isec.hit) return eval_environment(sc, incoming);
  auto s        = load_surface(sc, isec.instance, isec.element, {isec.u, isec.v});
  auto material = eval_material(sc, *s.sh, *s.mat, s.e, s.uv);
  auto normal   = eval_shading_normal(sc, s.frame, *s.sh, *s.mat, s.e, s.uv, -incoming);
  return eval_emission(material, normal, -incoming);
}

// ---------------------------------------------------------------------------
// trace_path / trace_pathdirect / trace_pathmis / trace_pathtest — one iteration
// of the bounce loop after the intersection (yocto_trace.cpp:453-1029)
// ---------------------------------------------------------------------------
template <int SAMPLER, bool COUNT>
YT_FN int step_path(ShadeEnv& E, Path& P) {
  const auto& sc = E.sc;
  const auto& kp = E.kp;
  constexpr bool DIRECT  = SAMPLER == YTHIP_SAMPLER_PATHDIRECT;
  constexpr bool MIS     = SAMPLER == YTHIP_SAMPLER_PATHMIS;
  constexpr bool TEST    = SAMPLER == YTHIP_SAMPLER_PATHTEST;
  constexpr bool VOLUMES = !TEST;
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
    auto s        = load_surface(sc, isec.instance, isec.element, {isec.u, isec.v});
    auto position = eval_shading_position(sc, s.frame, *s.sh, s.e, s.uv);
    auto normal   = eval_shading_normal(sc, s.frame, *s.sh, *s.mat, s.e, s.uv, outgoing);
    auto material = eval_material(sc, *s.sh, *s.mat, s.e, s.uv);
    if (COUNT && E.st.counters) atomicAdd(&E.st.counters[CNT_SHADES], 1ull);
    if (TEST) material.type = YTHIP_MATTE;

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
      set_first_hit(E.st, E.slot, material.color, normal);
    }

    // accumulate emission
    if ((!DIRECT && !MIS) || next_emission) P.radiance += P.weight * eval_emission(material, normal, outgoing);

    // direct (pathdirect) — yocto_trace.cpp:670-693
    if (DIRECT) {
      if (!is_delta(material)) {
        auto ruv      = rand2f(P.rng);  // g++ order: ruv, rel, rl
        auto rel      = rand1f(P.rng);
        auto rl       = rand1f(P.rng);
        auto incoming = sample_lights(sc, position, rl, rel, ruv);
        auto pdf      = sample_lights_pdf<COUNT>(sc, position, incoming, E.stack, E.cnt);
        auto bsdfcos  = eval_bsdfcos(material, normal, outgoing, incoming);
        if (bsdfcos != vec3f{0, 0, 0} && pdf > 0) {
          auto nisec    = trace_ray<COUNT>(sc, position, incoming, -1, E.stack, E.cnt);
          auto emission = nee_emission(sc, nisec, incoming);
          P.radiance += P.weight * bsdfcos * emission / pdf;
        }
        P.flags |= PF_NOEMIT;
      } else {
        P.flags &= ~PF_NOEMIT;
      }
    }

    // next direction
    auto incoming = vec3f{0, 0, 0};
    if (!is_delta(material)) {
      if (MIS) {
        // direct with MIS — yocto_trace.cpp:853-892
        for (int pass = 0; pass < 2; pass++) {
          const bool sample_light = pass == 0;
          if (sample_light) {
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
          auto bsdfcos    = eval_bsdfcos(material, normal, outgoing, incoming);
          auto light_pdf  = sample_lights_pdf<COUNT>(sc, position, incoming, E.stack, E.cnt);
          auto bsdf_pdf   = sample_bsdfcos_pdf(material, normal, outgoing, incoming);
          auto heur       = [](float this_pdf, float other_pdf) {
            return (this_pdf * this_pdf) / (this_pdf * this_pdf + other_pdf * other_pdf);
          };
          auto mis_weight = sample_light ? heur(light_pdf, bsdf_pdf) / light_pdf
                                         : heur(bsdf_pdf, light_pdf) / bsdf_pdf;
          if (bsdfcos != vec3f{0, 0, 0} && mis_weight != 0) {
            auto nisec = trace_ray<COUNT>(sc, position, incoming, -1, E.stack, E.cnt);
            if (!sample_light) {
              // next_intersection = intersection (persists across bounces)
              E.st.nhit_a[E.slot] = {nisec.u, nisec.v, nisec.distance, __int_as_float(nisec.hit ? nisec.instance : -1)};
              E.st.nhit_e[E.slot] = nisec.element;
            }
            auto emission = nee_emission(sc, nisec, incoming);
            P.radiance += P.weight * bsdfcos * emission * mis_weight;
          }
        }
        // indirect
        P.weight *= eval_bsdfcos(material, normal, outgoing, incoming) /
                    sample_bsdfcos_pdf(material, normal, outgoing, incoming);
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
        if (incoming == vec3f{0, 0, 0}) return STEP_END;
        P.weight *= eval_bsdfcos(material, normal, outgoing, incoming) /
                    (0.5f * sample_bsdfcos_pdf(material, normal, outgoing, incoming) +
                        0.5f * sample_lights_pdf<COUNT>(sc, position, incoming, E.stack, E.cnt));
      }
    } else {
      incoming = sample_delta(material, normal, outgoing, rand1f(P.rng));
      if (DIRECT && incoming == vec3f{0, 0, 0}) return STEP_END;
      P.weight *= eval_delta(material, normal, outgoing, incoming) /
                  sample_delta_pdf(material, normal, outgoing, incoming);
      if (MIS) P.flags &= ~PF_NOEMIT;
    }

    // update volume stack
    if (VOLUMES && is_volumetric(*s.mat) && dot(normal, outgoing) * dot(normal, incoming) < 0) {
      if (!(P.flags & PF_VOLUME)) {
        auto vmat = eval_material(sc, *s.sh, *s.mat, s.e, s.uv);
        store_volume(E.st, E.slot, vmat);
        P.flags |= PF_VOLUME;
      } else {
        P.flags &= ~PF_VOLUME;
      }
    }

    // setup next iteration
    P.o = position;
    P.d = incoming;
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
    P.weight *= eval_scattering(vsdf, outgoing, incoming) /
                (0.5f * sample_scattering_pdf(vsdf, outgoing, incoming) +
                    0.5f * sample_lights_pdf<COUNT>(sc, position, incoming, E.stack, E.cnt));
    P.o = position;
    P.d = incoming;
  }

  // check weight
  if (P.weight == vec3f{0, 0, 0} || !isfinite_(P.weight)) return STEP_END;

  // russian roulette
  if (P.bounce > 3) {
    auto rr_prob = min_((float)0.99, max_(P.weight));
    if (rand1f(P.rng) >= rr_prob) return STEP_END;
    P.weight *= 1 / rr_prob;
  }
  return STEP_NEXT;
}

// ---------------------------------------------------------------------------
// trace_naive / trace_furnace — yocto_trace.cpp:1032-1108, 1247-1338
// ---------------------------------------------------------------------------
template <int SAMPLER, bool COUNT>
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
  auto position = FURNACE ? eval_position(sc, s.frame, *s.sh, s.e, s.uv)
                          : eval_shading_position(sc, s.frame, *s.sh, s.e, s.uv);
  auto normal   = eval_shading_normal(sc, s.frame, *s.sh, *s.mat, s.e, s.uv, outgoing);
  auto material = eval_material(sc, *s.sh, *s.mat, s.e, s.uv);
  if (COUNT && E.st.counters) atomicAdd(&E.st.counters[CNT_SHADES], 1ull);

  if (material.opacity < 1 && rand1f(P.rng) >= material.opacity) {
    if (P.opbounce++ > 128) return STEP_END;
    P.o = position + P.d * 1e-2f;
    return STEP_RETRY;
  }
  if (P.bounce == 0) {
    P.flags |= PF_HIT;
    set_first_hit(E.st, E.slot, material.color, normal);
  }
  P.radiance += P.weight * eval_emission(material, normal, outgoing);

  auto incoming = vec3f{0, 0, 0};
  if (material.roughness != 0) {
    auto rn  = rand2f(P.rng);
    auto rnl = rand1f(P.rng);
    incoming = sample_bsdfcos(material, normal, outgoing, rnl, rn);
    if (incoming == vec3f{0, 0, 0}) return STEP_END;
    P.weight *= eval_bsdfcos(material, normal, outgoing, incoming) /
                sample_bsdfcos_pdf(material, normal, outgoing, incoming);
  } else {
    incoming = sample_delta(material, normal, outgoing, rand1f(P.rng));
    if (incoming == vec3f{0, 0, 0}) return STEP_END;
    P.weight *= eval_delta(material, normal, outgoing, incoming) /
                sample_delta_pdf(material, normal, outgoing, incoming);
  }
  if (P.weight == vec3f{0, 0, 0} || !isfinite_(P.weight)) return STEP_END;
  if (P.bounce > 3) {
    auto rr_prob = min_((float)0.99, max_(P.weight));
    if (rand1f(P.rng) >= rr_prob) return STEP_END;
    P.weight *= 1 / rr_prob;
  }
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
template <int SAMPLER, bool COUNT>
YT_FN int step_eyelight(ShadeEnv& E, Path& P) {
  const auto& sc = E.sc;
  const auto& kp = E.kp;
  constexpr bool DIAGRAM = SAMPLER == YTHIP_SAMPLER_DIAGRAM;
  auto&          isec    = P.isec;
  if (!isec.hit) {
    if (DIAGRAM) {
      P.radiance += P.weight * vec3f{1, 1, 1};
      P.flags |= PF_HIT;
      if (P.bounce != 0) {
        // hit=true with the albedo/normal recorded at bounce 0
      } else {
        set_first_hit(E.st, E.slot, {0, 0, 0}, {0, 0, 0});
      }
    } else if (P.bounce > 0 || !kp.envhidden) {
      P.radiance += P.weight * eval_environment(sc, P.d);
    }
    return STEP_END;
  }
  auto outgoing = -P.d;
  auto s        = load_surface(sc, isec.instance, isec.element, {isec.u, isec.v});
  auto position = eval_shading_position(sc, s.frame, *s.sh, s.e, s.uv);
  auto normal   = eval_shading_normal(sc, s.frame, *s.sh, *s.mat, s.e, s.uv, outgoing);
  auto material = eval_material(sc, *s.sh, *s.mat, s.e, s.uv);
  if (COUNT && E.st.counters) atomicAdd(&E.st.counters[CNT_SHADES], 1ull);

  if (material.opacity < 1 && rand1f(P.rng) >= material.opacity) {
    if (P.opbounce++ > 128) return STEP_END;
    P.o = position + P.d * 1e-2f;
    return STEP_RETRY;
  }
  if (P.bounce == 0) {
    P.flags |= PF_HIT;
    set_first_hit(E.st, E.slot, material.color, normal);
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
  return {powf(c.x, 2.2f), powf(c.y, 2.2f), powf(c.z, 2.2f)};
}
template <bool COUNT>
YT_FN int step_falsecolor(ShadeEnv& E, Path& P) {
  const auto& sc   = E.sc;
  auto&       isec = P.isec;
  if (!isec.hit) return STEP_END;  // trace_result{}: radiance 0, hit false, albedo 0, normal 0
  auto outgoing = -P.d;
  auto s        = load_surface(sc, isec.instance, isec.element, {isec.u, isec.v});
  auto position = eval_shading_position(sc, s.frame, *s.sh, s.e, s.uv);
  auto normal   = eval_shading_normal(sc, s.frame, *s.sh, *s.mat, s.e, s.uv, outgoing);
  auto gnormal  = eval_element_normal(sc, s.frame, *s.sh, s.e);
  auto texcoord = eval_texcoord(sc, *s.sh, s.e, s.uv);
  auto material = eval_material(sc, *s.sh, *s.mat, s.e, s.uv);
  auto delta    = is_delta(material) ? 1.0f : 0.0f;
  if (COUNT && E.st.counters) atomicAdd(&E.st.counters[CNT_SHADES], 1ull);
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
  set_first_hit(E.st, E.slot, material.color, normal);
  return STEP_END;
}

// ===========================================================================
// Kernels
// ===========================================================================

// k_generate: head of trace_sample (yocto_trace.cpp:1464-1468) for every pixel
// of the slice; initialises the path and enqueues it.
__global__ void __launch_bounds__(YT_BLOCK) k_generate(DScene sc, DState st, KParams kp) {
  int slot = blockIdx.x * YT_BLOCK + threadIdx.x;
  if (slot >= st.npix) return;
  int  i = slot % st.width, j = st.row_begin + slot / st.width;
  auto r = st.rngs[slot];
  rng_state rng = {r.x, r.y};
  // sample_camera(camera, ij, size, puv = rand2f, luv = rand2f, tent): g++ draws luv first
  auto luv = rand2f(rng);
  auto puv = rand2f(rng);
  auto ray = sample_camera(sc.cameras[kp.camera], i, j, st.width, st.height, puv, luv, kp.tentfilter != 0);
  st.rngs[slot]    = {rng.state, rng.inc};
  st.ray_a[slot]   = {ray.o.x, ray.o.y, ray.o.z, ray.d.x};
  st.ray_b[slot]   = {ray.d.y, ray.d.z, __int_as_float(0), __int_as_float(0)};
  st.wgt[slot]     = {1, 1, 1, 0};
  st.rad[slot]     = {0, 0, 0, 0};
  st.first_a[slot] = {0, 0, 0, -ray.d.x};
  st.first_b[slot] = {-ray.d.y, -ray.d.z};
  if (st.nhit_a) {
    st.nhit_a[slot] = {0, 0, 0, __int_as_float(-1)};
    st.nhit_e[slot] = -1;
  }
  st.queue[0][slot] = slot;
  if (slot == 0) {
    st.qcount[0] = st.npix;
    st.qcount[1] = 0;
  }
}

// k_extend: intersect_scene_bvh for every live path (the traversal kernel).
template <bool COUNT>
__global__ void __launch_bounds__(YT_BLOCK) k_extend(DScene sc, DState st, int q) {
  __shared__ int s_stack[YT_LDS_DEPTH][YT_BLOCK];
  int            idx = blockIdx.x * YT_BLOCK + threadIdx.x;
  // The other queue was fully consumed by the previous k_shade: reset its counter
  // before this iteration's k_shade appends to it (must happen even when the
  // live queue is empty, or a stale count would resurrect dead paths).
  if (idx == 0) st.qcount[q ^ 1] = 0;
  if (idx >= st.qcount[q]) return;
  int    slot = st.queue[q][idx];
  float4 a = st.ray_a[slot], b = st.ray_b[slot];
  int    flags = __float_as_int(b.w);
  if (flags & PF_SKIPEXTEND) {         // pathmis: intersection = next_intersection
    st.hit_a[slot] = st.nhit_a[slot];
    st.hit_e[slot] = st.nhit_e[slot];
    return;
  }
  Stack stack;
  stack.lds = &s_stack[0][threadIdx.x];
  Counters cnt = {0, 0, 0, 0, 0, 0, 0};
  // furnace: `if (bounce > 0 && !in_volume) { env; break; }` happens BEFORE the
  // intersection (yocto_trace.cpp:1263-1266); handled in k_shade by a flag check.
  ray3f ray = make_ray({a.x, a.y, a.z}, {a.w, b.x, b.y});
  Hit   h   = traverse<COUNT>(sc, ray, -1, false, stack, cnt);
  st.hit_a[slot] = {h.u, h.v, h.distance, __int_as_float(h.hit ? h.instance : -1)};
  st.hit_e[slot] = h.element;
  if (COUNT) flush_counters(st.counters, cnt);
}

// k_shade: one iteration of the integrator's bounce loop for every live path,
// then wave-ballot compaction of the survivors into the other queue.
template <int SAMPLER, bool COUNT>
__global__ void __launch_bounds__(YT_BLOCK) k_shade(DScene sc, DState st, KParams kp, int q) {
  __shared__ int s_stack[YT_LDS_DEPTH][YT_BLOCK];
  int            idx   = blockIdx.x * YT_BLOCK + threadIdx.x;
  bool           alive = false;
  int            slot  = -1;
  if (idx < st.qcount[q]) {
    slot = st.queue[q][idx];
    Path   P;
    float4 ra = st.ray_a[slot], rb = st.ray_b[slot];
    float4 ha = st.hit_a[slot];
    float4 w = st.wgt[slot], r = st.rad[slot];
    auto   g = st.rngs[slot];
    P.o             = {ra.x, ra.y, ra.z};
    P.d             = {ra.w, rb.x, rb.y};
    P.bounce        = __float_as_int(rb.z);
    int fw          = __float_as_int(rb.w);
    P.flags         = fw & 0xff;
    P.opbounce      = fw >> 8;
    int inst        = __float_as_int(ha.w);
    P.isec          = {inst, st.hit_e[slot], ha.x, ha.y, ha.z, inst >= 0};
    P.weight        = {w.x, w.y, w.z};
    P.max_roughness = w.w;
    P.radiance      = {r.x, r.y, r.z};
    P.rng           = {g.x, g.y};
    P.flags &= ~PF_SKIPEXTEND;

    Stack stack;
    stack.lds    = &s_stack[0][threadIdx.x];
    Counters cnt = {0, 0, 0, 0, 0, 0, 0};
    ShadeEnv E   = {sc, st, kp, stack, cnt, slot};

    int step;
    if (SAMPLER == YTHIP_SAMPLER_PATH || SAMPLER == YTHIP_SAMPLER_PATHDIRECT ||
        SAMPLER == YTHIP_SAMPLER_PATHMIS || SAMPLER == YTHIP_SAMPLER_PATHTEST) {
      step = step_path<SAMPLER, COUNT>(E, P);
    } else if (SAMPLER == YTHIP_SAMPLER_NAIVE) {
      step = step_naive<SAMPLER, COUNT>(E, P);
    } else if (SAMPLER == YTHIP_SAMPLER_FURNACE) {
      // exit test at the top of the loop body — yocto_trace.cpp:1263-1266
      if (P.bounce > 0 && !(P.flags & PF_INVOL)) {
        P.radiance += P.weight * eval_environment(sc, P.d);
        step = STEP_END;
      } else {
        step = step_naive<SAMPLER, COUNT>(E, P);
      }
    } else if (SAMPLER == YTHIP_SAMPLER_EYELIGHT || SAMPLER == YTHIP_SAMPLER_DIAGRAM) {
      step = step_eyelight<SAMPLER, COUNT>(E, P);
    } else {
      step = step_falsecolor<COUNT>(E, P);
    }

    if (SAMPLER == YTHIP_SAMPLER_PATHMIS && (P.flags & PF_NOEMIT)) P.flags |= PF_SKIPEXTEND;
    int max_bounces = (SAMPLER == YTHIP_SAMPLER_EYELIGHT || SAMPLER == YTHIP_SAMPLER_DIAGRAM)
                          ? max_(kp.bounces, 4)
                          : kp.bounces;
    if (step == STEP_NEXT) {
      P.bounce += 1;
      alive = P.bounce < max_bounces;
    } else if (step == STEP_RETRY) {
      alive = true;
    }

    st.rngs[slot] = {P.rng.state, P.rng.inc};
    st.rad[slot]  = {P.radiance.x, P.radiance.y, P.radiance.z, 0};
    if (alive) {
      st.ray_a[slot] = {P.o.x, P.o.y, P.o.z, P.d.x};
      st.ray_b[slot] = {P.d.y, P.d.z, __int_as_float(P.bounce), __int_as_float(P.flags | (P.opbounce << 8))};
      st.wgt[slot]   = {P.weight.x, P.weight.y, P.weight.z, P.max_roughness};
    } else {
      // keep the hit flag for k_accumulate
      st.ray_b[slot].w = __int_as_float(P.flags | (P.opbounce << 8));
    }
    if (COUNT) flush_counters(st.counters, cnt);
  }
  // stream compaction: wave ballot + one atomic per wave
  unsigned long long mask = __ballot(alive);
  if (mask) {
    int lane   = threadIdx.x & 63;
    int leader = __ffsll((long long)mask) - 1;
    int base   = 0;
    if (lane == leader) base = atomicAdd(&st.qcount[q ^ 1], __popcll(mask));
    base = __shfl(base, leader);
    if (alive) {
      int rank = __popcll(mask & ((1ull << lane) - 1ull));
      st.queue[q ^ 1][base + rank] = slot;
    }
  }
}


// k_accumulate: tail of trace_sample (yocto_trace.cpp:1471-1491).
template <bool COUNT>
__global__ void __launch_bounds__(YT_BLOCK) k_accumulate(DState st, KParams kp, int sample) {
  int slot = blockIdx.x * YT_BLOCK + threadIdx.x;
  if (slot >= st.npix) return;
  float4 r  = st.rad[slot];
  int    fl = __float_as_int(st.ray_b[slot].w);
  float4 fa = st.first_a[slot];
  float2 fb = st.first_b[slot];
  vec3f  radiance = {r.x, r.y, r.z};
  bool   hit      = (fl & PF_HIT) != 0;
  vec3f  albedo = {fa.x, fa.y, fa.z}, normal = {fa.w, fb.x, fb.y};
  if (!isfinite_(radiance)) radiance = {0, 0, 0};
  if (max_(radiance) > kp.clamp) radiance = radiance * (kp.clamp / max_(radiance));
  auto   weight = 1.0f / (sample + 1);
  float4 im     = st.image[slot];
  vec4f  image  = {im.x, im.y, im.z, im.w};
  vec3f  alb    = ld3(st.albedo, slot);
  vec3f  nrm    = ld3(st.normal, slot);
  // before the first hit `normal` holds -camera_ray.d
  if (hit) {
    image = lerp_(image, vec4f{radiance.x, radiance.y, radiance.z, 1}, weight);
    alb   = lerp_(alb, albedo, weight);
    nrm   = lerp_(nrm, normal, weight);
    st.hits[slot] += 1;
  } else if (!kp.envhidden && kp.has_env) {
    image = lerp_(image, vec4f{radiance.x, radiance.y, radiance.z, 1}, weight);
    alb   = lerp_(alb, vec3f{1, 1, 1}, weight);
    nrm   = lerp_(nrm, normal, weight);
    st.hits[slot] += 1;
  } else {
    image = lerp_(image, vec4f{0, 0, 0, 0}, weight);
    alb   = lerp_(alb, vec3f{0, 0, 0}, weight);
    nrm   = lerp_(nrm, normal, weight);
  }
  st.image[slot]          = {image.x, image.y, image.z, image.w};
  st.albedo[3 * slot]     = alb.x;
  st.albedo[3 * slot + 1] = alb.y;
  st.albedo[3 * slot + 2] = alb.z;
  st.normal[3 * slot]     = nrm.x;
  st.normal[3 * slot + 1] = nrm.y;
  st.normal[3 * slot + 2] = nrm.z;
  if (COUNT && st.counters && slot == 0) atomicAdd(&st.counters[CNT_SAMPLES], (unsigned long long)st.npix);
}

// Test/parity entries ---------------------------------------------------------
template <bool COUNT>
__global__ void __launch_bounds__(YT_BLOCK) k_intersect_batch(DScene sc, const ythip_ray* rays,
    const int* instances, long long n, int find_any, ythip_hit* hits, unsigned long long* counters) {
  __shared__ int s_stack[YT_LDS_DEPTH][YT_BLOCK];
  long long      idx = (long long)blockIdx.x * YT_BLOCK + threadIdx.x;
  if (idx >= n) return;
  Stack stack;
  stack.lds    = &s_stack[0][threadIdx.x];
  Counters cnt = {0, 0, 0, 0, 0, 0, 0};
  auto     r   = rays[idx];
  ray3f    ray = {{r.o[0], r.o[1], r.o[2]}, {r.d[0], r.d[1], r.d[2]}, r.tmin, r.tmax};
  Hit      h   = traverse<COUNT>(sc, ray, instances ? instances[idx] : -1, find_any != 0, stack, cnt);
  // scene_intersection{} defaults when missed: instance -1, element -1, uv 0, distance 0
  if (!h.hit) h = {-1, -1, 0, 0, 0, false};
  hits[idx] = {h.instance, h.element, h.u, h.v, h.distance, h.hit ? 1 : 0};
  if (COUNT) flush_counters(counters, cnt);
}

__global__ void __launch_bounds__(YT_BLOCK) k_camera_rays(DScene sc, DState st, KParams kp, ythip_ray* rays) {
  int slot = blockIdx.x * YT_BLOCK + threadIdx.x;
  if (slot >= st.npix) return;
  int  i = slot % st.width, j = st.row_begin + slot / st.width;
  auto r = st.rngs[slot];
  rng_state rng = {r.x, r.y};
  auto luv = rand2f(rng);
  auto puv = rand2f(rng);
  auto ray = sample_camera(sc.cameras[kp.camera], i, j, st.width, st.height, puv, luv, kp.tentfilter != 0);
  rays[slot] = {{ray.o.x, ray.o.y, ray.o.z}, {ray.d.x, ray.d.y, ray.d.z}, ray.tmin, ray.tmax};
}

}  // namespace yt
