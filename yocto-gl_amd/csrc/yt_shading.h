// yt_shading.h — the surface and volume lobes of the path, device-shaped.
//
// What the reference specifies (libs/yocto/yocto_shading.h:302-1111 and the material dispatch of
// libs/yocto/yocto_trace.cpp:166-335) is ARITHMETIC: for a shading point (material, normal,
// outgoing) and a direction `incoming`, the floats of eval_bsdfcos, sample_bsdfcos_pdf,
// sample_bsdfcos and their delta counterparts, in the reference's association order.  How that
// arithmetic is packaged is ours.  The reference evaluates f and its pdf through three independent
// per-material switches that each rebuild the up-normal, the half vector, the Fresnel term and the
// GGX distribution; every integrator of the path wants f AND pdf of the same direction pair (the
// weight update f / pdf, the MIS terms), so here
//
//   * eval_lobe(m, n, o, i) returns {f, pdf} of one direction pair from ONE evaluation of the
//     shared terms — each shared term is the same expression on the same operands in both of the
//     reference's functions (listed per lobe below), so both results keep their bits;
//   * the geometry terms are computed once per call as cosines and handed to small cosine-space
//     primitives (Fresnel, GGX D and G1) instead of being re-derived from vectors inside each;
//   * there is ONE material switch per question (eval_lobe, sample_lobe, and the three delta
//     forms), with `type` and `roughness == 0` tested once.
//
// The kernels inline all of it; what changes for the compiler is that the f and pdf computations
// of a lobe now live in one basic-block region with their common subexpressions written once,
// instead of in two switch statements it could not merge.
//
// Bit-exactness notes that are easy to get wrong:
//   * -n is NOT interchangeable with n under a dot product when the result can be zero
//     (dot(-n, i) and -dot(n, i) may differ in the sign of a zero), so every dot below is taken
//     on exactly the vectors the reference takes it on;
//   * a + b == b + a bit for bit, so the reference's `incoming + outgoing` / `outgoing + incoming`
//     half vectors are one value;
//   * g++ -O3 folds pow(x, 2.0f) style calls; nothing here depends on that (powf(x, 5) is a call).
#pragma once

#include "../../include/ythip.h"  // material types
#include "yt_material.h"

#ifdef YT_FAST  // tolerance mode: multiply-adds written in this file may fuse (the traversal's, in yt_bvh.h, never do)
#pragma clang fp contract(fast)
#endif
namespace yt {

// f = eval_bsdfcos, pdf = sample_bsdfcos_pdf of one (outgoing, incoming) pair
struct LobeEval {
  vec3f f;
  float pdf;
};

// ---------------------------------------------------------------------------
// cosine-space primitives
// ---------------------------------------------------------------------------
// fresnel_dielectric(eta, normal, dir) with c = dot(normal, dir) — yocto_shading.h:318-338
YT_FN float dielectric_fresnel(float eta, float c) {
  auto cosw  = fabs_(c);
  auto sin2  = 1 - cosw * cosw;
  auto eta2  = eta * eta;
  auto cos2t = 1 - div_(sin2, eta2);
  if (cos2t < 0) return 1;  // total internal reflection
  auto t0 = sqrt_(cos2t);
  auto t1 = eta * t0;
  auto t2 = eta * cosw;
  auto rs = div_(cosw - t1, cosw + t1);
  auto rp = div_(t0 - t2, t0 + t2);
  return div_(rs * rs + rp * rp, 2);
}
// fresnel_schlick(specular, normal, dir) with c = dot(normal, dir) — :308-315
YT_FN vec3f schlick_fresnel(vec3f specular, float c) {
  if (specular == vec3f{0, 0, 0}) return {0, 0, 0};
  return specular + (1 - specular) * ytm::powf(clamp_(1 - fabs_(c), 0.0f, 1.0f), 5.0f);
}
// fresnel_conductor(eta, etak = 0, normal, dir) with c = dot(normal, dir) — :341-366.  The
// reference is always called with etak = {0,0,0} on this path (:626, :696): etak2 = 0, and
// x - 0 == x, x + 0 * y == x for the finite eta2 that reflectivity_to_eta produces (<= 399^2), so
// the etak terms are dropped without changing a bit.
YT_FN vec3f conductor_fresnel(vec3f eta, float c) {
  if (c <= 0) return {0, 0, 0};
  auto cosw     = clamp_(c, (float)-1, (float)1);
  auto cos2     = cosw * cosw;
  auto sin2     = clamp_(1 - cos2, (float)0, (float)1);
  auto eta2     = eta * eta;
  auto t0       = eta2 - sin2;
  auto a2plusb2 = sqrt_(t0 * t0);
  auto t1       = a2plusb2 + cos2;
  auto a        = sqrt_((a2plusb2 + t0) / 2);
  auto t2       = 2 * a * cosw;
  auto rs       = (t1 - t2) / (t1 + t2);
  auto t3       = cos2 * a2plusb2 + sin2 * sin2;
  auto t4       = t2 * sin2;
  auto rp       = rs * (t3 - t4) / (t3 + t4);
  return (rp + rs) / 2;
}
// eta_to_reflectivity / reflectivity_to_eta — :369-376
YT_FN vec3f eta_to_reflectivity(vec3f eta) { return ((eta - 1) * (eta - 1)) / ((eta + 1) * (eta + 1)); }
YT_FN vec3f reflectivity_to_eta(vec3f reflectivity_) {
  auto reflectivity = clamp_(reflectivity_, 0.0f, 0.99f);
  return (1 + sqrt_(reflectivity)) / (1 - sqrt_(reflectivity));
}
// GGX: microfacet_distribution with nh = dot(normal, halfway) — :409-424
YT_FN float ggx_d(float roughness, float nh) {
  if (nh <= 0) return 0;
  auto roughness2 = roughness * roughness;
  auto cosine2    = nh * nh;
  return div_(roughness2, pif * (cosine2 * roughness2 + 1 - cosine2) * (cosine2 * roughness2 + 1 - cosine2));
}
// microfacet_shadowing1 with nd = dot(normal, dir), hd = dot(halfway, dir) — :427-447
YT_FN float ggx_g1(float roughness, float nd, float hd) {
  if (nd * hd <= 0) return 0;
  auto roughness2 = roughness * roughness;
  auto cosine2    = nd * nd;
  return div_(2 * fabs_(nd), fabs_(nd) + sqrt_(cosine2 - roughness2 * cosine2 + roughness2));
}
// sample_microfacet_pdf given D and nh — :474-479
YT_FN float ggx_pdf(float d, float nh) { return nh < 0 ? 0 : d * nh; }
// sample_microfacet (ggx) — :458-471
YT_FN vec3f sample_microfacet(float roughness, vec3f normal, vec2f rn) {
  auto  phi   = 2 * pif * rn.x;
  auto  theta = ytm::atanf(roughness * sqrt_(div_(rn.y, 1 - rn.y)));
  float sp, cp, st, ct;
  ytm::sincosf(phi, &sp, &cp);
  ytm::sincosf(theta, &st, &ct);
  return transform_direction(basis_fromz(normal), vec3f{cp * st, sp * st, ct});
}
// same_hemisphere — :302-305
YT_FN bool same_hemisphere(vec3f normal, vec3f outgoing, vec3f incoming) {
  return dot(normal, outgoing) * dot(normal, incoming) >= 0;
}

// The microfacet reflection term F D G / (4 (n.o) (n.i)) |n.i| of the rough lobes and the pdf of
// having sampled `halfway` and reflected about it, D (n.h) / (4 |o.h|): the pieces glossy,
// reflective, gltfpbr and the reflection halves of transparent / refractive share.
struct Specular {
  float d, g, nh;
};
YT_FN Specular specular_terms(float roughness, vec3f up, vec3f halfway, vec3f outgoing, vec3f incoming) {
  Specular s;
  s.nh = dot(up, halfway);
  s.d  = ggx_d(roughness, s.nh);
  s.g  = ggx_g1(roughness, dot(up, outgoing), dot(halfway, outgoing)) *
        ggx_g1(roughness, dot(up, incoming), dot(halfway, incoming));
  return s;
}

// ---------------------------------------------------------------------------
// the rough lobes: {f, pdf}
// ---------------------------------------------------------------------------
// matte — eval :554-559, pdf :568-573 (shared: the sidedness product)
YT_FN LobeEval lobe_matte(vec3f color, vec3f n, vec3f o, vec3f i) {
  auto ni = dot(n, i), no = dot(n, o);
  if (ni * no <= 0) return {{0, 0, 0}, 0};
  auto up = no <= 0 ? -n : n;
  return {color / pif * fabs_(ni), sample_hemisphere_cos_pdf(up, i)};
}
// glossy — eval :576-589, pdf :607-617.  Shared: up-normal, F1 = fresnel(ior, up, o) (the pdf's
// mixture weight IS the eval's F1), the half vector, D.
YT_FN LobeEval lobe_glossy(vec3f color, float ior, float roughness, vec3f n, vec3f o, vec3f i) {
  auto no = dot(n, o);
  if (dot(n, i) * no <= 0) return {{0, 0, 0}, 0};
  auto up      = no <= 0 ? -n : n;
  auto f1      = dielectric_fresnel(ior, dot(up, o));
  auto halfway = normalize(i + o);
  auto fh      = dielectric_fresnel(ior, dot(halfway, i));
  auto s       = specular_terms(roughness, up, halfway, o, i);
  auto ui = dot(up, i), uo = dot(up, o);
  LobeEval r;
  r.f   = color * (1 - f1) / pif * fabs_(ui) + vec3f{1, 1, 1} * fh * s.d * s.g / (4 * uo * ui) * fabs_(ui);
  r.pdf = div_(f1 * ggx_pdf(s.d, s.nh), 4 * fabs_(dot(o, halfway))) + (1 - f1) * sample_hemisphere_cos_pdf(up, i);
  return r;
}
// reflective (rough, colour-parametrised conductor) — eval :620-630, pdf :641-650
YT_FN LobeEval lobe_reflective(vec3f color, float roughness, vec3f n, vec3f o, vec3f i) {
  auto no = dot(n, o);
  if (dot(n, i) * no <= 0) return {{0, 0, 0}, 0};
  auto up      = no <= 0 ? -n : n;
  auto halfway = normalize(i + o);
  auto fh      = conductor_fresnel(reflectivity_to_eta(color), dot(halfway, i));
  auto s       = specular_terms(roughness, up, halfway, o, i);
  auto ui      = dot(up, i);
  LobeEval r;
  r.f   = fh * s.d * s.g / (4 * dot(up, o) * ui) * fabs_(ui);
  r.pdf = div_(ggx_pdf(s.d, s.nh), 4 * fabs_(dot(o, halfway)));
  return r;
}
// gltfpbr — eval :739-752, pdf :776-789.  Shared: reflectivity, F1 (the pdf's weight is mean(F1)).
YT_FN LobeEval lobe_gltfpbr(vec3f color, float ior, float roughness, float metallic, vec3f n, vec3f o, vec3f i) {
  auto no = dot(n, o);
  if (dot(n, i) * no <= 0) return {{0, 0, 0}, 0};
  auto reflectivity = lerp_(eta_to_reflectivity(vec3f{ior, ior, ior}), color, metallic);
  auto up           = no <= 0 ? -n : n;
  auto f1           = schlick_fresnel(reflectivity, dot(up, o));
  auto halfway      = normalize(i + o);
  auto fh           = schlick_fresnel(reflectivity, dot(halfway, i));
  auto s            = specular_terms(roughness, up, halfway, o, i);
  auto ui = dot(up, i), w = mean(f1);
  LobeEval r;
  r.f   = color * (1 - metallic) * (1 - f1) / pif * fabs_(ui) + fh * s.d * s.g / (4 * dot(up, o) * ui) * fabs_(ui);
  r.pdf = div_(w * ggx_pdf(s.d, s.nh), 4 * fabs_(dot(o, halfway))) + (1 - w) * sample_hemisphere_cos_pdf(up, i);
  return r;
}
// transparent (thin rough dielectric) — eval :792-812, pdf :833-846.  Transmission is evaluated on
// the direction mirrored back to the outgoing side.
YT_FN LobeEval lobe_transparent(vec3f color, float ior, float roughness, vec3f n, vec3f o, vec3f i) {
  auto no = dot(n, o);
  auto up = no <= 0 ? -n : n;
  LobeEval r;
  if (dot(n, i) * no >= 0) {
    auto halfway = normalize(i + o);
    auto fh      = dielectric_fresnel(ior, dot(halfway, o));
    auto s       = specular_terms(roughness, up, halfway, o, i);
    auto ui      = dot(up, i);
    r.f          = vec3f{1, 1, 1} * fh * s.d * s.g / (4 * dot(up, o) * ui) * fabs_(ui);
    r.pdf        = div_(fh * ggx_pdf(s.d, s.nh), 4 * fabs_(dot(o, halfway)));
  } else {
    auto reflected = reflect(-i, up);
    auto halfway   = normalize(reflected + o);
    auto fh        = dielectric_fresnel(ior, dot(halfway, o));
    auto s         = specular_terms(roughness, up, halfway, o, reflected);
    auto ur        = dot(up, reflected);
    r.f            = color * (1 - fh) * s.d * s.g / (4 * dot(up, o) * ur) * (fabs_(ur));
    auto d         = (1 - fh) * ggx_pdf(s.d, s.nh);
    r.pdf          = div_(d, 4 * fabs_(dot(o, halfway)));
  }
  return r;
}
// refractive (rough dielectric interface; subsurface shares it) — eval :881-907, pdf :937-956
YT_FN LobeEval lobe_refractive(vec3f color, float ior, float roughness, vec3f n, vec3f o, vec3f i) {
  auto no       = dot(n, o);
  auto entering = no >= 0;
  auto up       = entering ? n : -n;
  auto rel_ior  = entering ? ior : rcp_(ior);
  auto ni       = dot(n, i);
  LobeEval r;
  if (ni * no >= 0) {
    auto halfway = normalize(i + o);
    auto fh      = dielectric_fresnel(rel_ior, dot(halfway, o));
    auto s       = specular_terms(roughness, up, halfway, o, i);
    r.f          = vec3f{1, 1, 1} * fh * s.d * s.g / fabs_(4 * no * ni) * fabs_(ni);
    r.pdf        = div_(fh * ggx_pdf(s.d, s.nh), 4 * fabs_(dot(o, halfway)));
  } else {
    auto halfway = -normalize(rel_ior * i + o) * (entering ? 1.0f : -1.0f);
    auto fh      = dielectric_fresnel(rel_ior, dot(halfway, o));
    auto s       = specular_terms(roughness, up, halfway, o, i);
    // [Walter 2007] equations 21 (f) and 17 (pdf); the eval writes its dots (vector, halfway) /
    // (vector, normal), the pdf (halfway, vector): dot is commutative term by term
    auto oh = dot(o, halfway), ih = dot(i, halfway);
    r.f     = vec3f{1, 1, 1} * fabs_(div_(oh * ih, dot(o, n) * dot(i, n))) * (1 - fh) * s.d * s.g /
          sqr_(rel_ior * dot(halfway, i) + dot(halfway, o)) * fabs_(ni);
    r.pdf = div_((1 - fh) * ggx_pdf(s.d, s.nh) * fabs_(dot(halfway, i)), sqr_(rel_ior * dot(halfway, i) + dot(halfway, o)));
  }
  return r;
}

// eval_bsdfcos + sample_bsdfcos_pdf — yocto_trace.cpp:172-199, 270-297
YT_FN LobeEval eval_lobe(const material_point& m, vec3f n, vec3f o, vec3f i) {
  if (m.roughness == 0) return {{0, 0, 0}, 0};
  switch (m.type) {
    case YTHIP_MATTE: return lobe_matte(m.color, n, o, i);
    case YTHIP_GLOSSY: return lobe_glossy(m.color, m.ior, m.roughness, n, o, i);
    case YTHIP_REFLECTIVE: return lobe_reflective(m.color, m.roughness, n, o, i);
    case YTHIP_TRANSPARENT: return lobe_transparent(m.color, m.ior, m.roughness, n, o, i);
    case YTHIP_REFRACTIVE:
    case YTHIP_SUBSURFACE: return lobe_refractive(m.color, m.ior, m.roughness, n, o, i);
    case YTHIP_GLTFPBR: return lobe_gltfpbr(m.color, m.ior, m.roughness, m.metallic, n, o, i);
    default: return {{0, 0, 0}, 0};
  }
}
// the two halves for callers that want only one (everything is inlined: the other half is dead code)
YT_FN vec3f eval_bsdfcos(const material_point& m, vec3f n, vec3f o, vec3f i) { return eval_lobe(m, n, o, i).f; }
YT_FN float sample_bsdfcos_pdf(const material_point& m, vec3f n, vec3f o, vec3f i) { return eval_lobe(m, n, o, i).pdf; }

// ---------------------------------------------------------------------------
// sampling a direction from the rough lobes — yocto_trace.cpp:221-248 with
// yocto_shading.h:562-566, 592-604, 633-639, 755-773, 815-830, 910-934
// ---------------------------------------------------------------------------
// reflect about a sampled micro-normal; {0,0,0} when the result leaves the upper hemisphere
YT_FN vec3f reflect_in_hemisphere(vec3f up, vec3f o, vec3f halfway) {
  auto incoming = reflect(o, halfway);
  return same_hemisphere(up, o, incoming) ? incoming : vec3f{0, 0, 0};
}
YT_FN vec3f sample_lobe(const material_point& m, vec3f n, vec3f o, float rnl, vec2f rn) {
  if (m.roughness == 0) return {0, 0, 0};
  const auto no = dot(n, o);
  switch (m.type) {
    case YTHIP_MATTE: return sample_hemisphere_cos(no <= 0 ? -n : n, rn);
    case YTHIP_GLOSSY: {
      auto up = no <= 0 ? -n : n;
      if (rnl < dielectric_fresnel(m.ior, dot(up, o))) return reflect_in_hemisphere(up, o, sample_microfacet(m.roughness, up, rn));
      return sample_hemisphere_cos(up, rn);
    }
    case YTHIP_REFLECTIVE: {
      auto up = no <= 0 ? -n : n;
      return reflect_in_hemisphere(up, o, sample_microfacet(m.roughness, up, rn));
    }
    case YTHIP_GLTFPBR: {
      auto up           = no <= 0 ? -n : n;
      auto reflectivity = lerp_(eta_to_reflectivity(vec3f{m.ior, m.ior, m.ior}), m.color, m.metallic);
      if (rnl < mean(schlick_fresnel(reflectivity, dot(up, o)))) return reflect_in_hemisphere(up, o, sample_microfacet(m.roughness, up, rn));
      return sample_hemisphere_cos(up, rn);
    }
    case YTHIP_TRANSPARENT: {
      auto up      = no <= 0 ? -n : n;
      auto halfway = sample_microfacet(m.roughness, up, rn);
      if (rnl < dielectric_fresnel(m.ior, dot(halfway, o))) return reflect_in_hemisphere(up, o, halfway);
      auto incoming = -reflect(reflect(o, halfway), up);
      return same_hemisphere(up, o, incoming) ? vec3f{0, 0, 0} : incoming;
    }
    case YTHIP_REFRACTIVE:
    case YTHIP_SUBSURFACE: {
      auto entering = no >= 0;
      auto up       = entering ? n : -n;
      auto halfway  = sample_microfacet(m.roughness, up, rn);
      if (rnl < dielectric_fresnel(entering ? m.ior : rcp_(m.ior), dot(halfway, o))) return reflect_in_hemisphere(up, o, halfway);
      auto incoming = refract(o, halfway, entering ? rcp_(m.ior) : m.ior);
      return same_hemisphere(up, o, incoming) ? vec3f{0, 0, 0} : incoming;
    }
    default: return {0, 0, 0};
  }
}
YT_FN vec3f sample_bsdfcos(const material_point& m, vec3f n, vec3f o, float rnl, vec2f rn) { return sample_lobe(m, n, o, rnl, rn); }

// ---------------------------------------------------------------------------
// delta lobes (roughness == 0) — yocto_trace.cpp:201-218, 250-267, 299-316 with
// yocto_shading.h:692-713, 849-878, 959-1005, 1029-1053.  `abs(ior - 1) < 1e-3` compares in double.
// ---------------------------------------------------------------------------
// the Fresnel reflectance of the smooth dielectric interface seen from `o`, and which side `i` is on
struct DeltaDielectric {
  float fresnel, rel_ior;
  bool  same_side;
};
YT_FN DeltaDielectric delta_refractive_terms(float ior, vec3f n, vec3f o, vec3f i) {
  auto no       = dot(n, o);
  auto entering = no >= 0;
  auto up       = entering ? n : -n;
  auto rel_ior  = entering ? ior : rcp_(ior);
  return {dielectric_fresnel(rel_ior, dot(up, o)), rel_ior, dot(n, i) * no >= 0};
}
YT_FN vec3f eval_delta(const material_point& m, vec3f n, vec3f o, vec3f i) {
  if (m.roughness != 0) return {0, 0, 0};
  switch (m.type) {
    case YTHIP_REFLECTIVE: {
      auto no = dot(n, o);
      if (dot(n, i) * no <= 0) return {0, 0, 0};
      auto up = no <= 0 ? -n : n;
      return conductor_fresnel(reflectivity_to_eta(m.color), dot(up, o));
    }
    case YTHIP_TRANSPARENT: {
      auto no = dot(n, o);
      auto up = no <= 0 ? -n : n;
      auto fr = dielectric_fresnel(m.ior, dot(up, o));
      return dot(n, i) * no >= 0 ? vec3f{1, 1, 1} * fr : m.color * (1 - fr);
    }
    case YTHIP_REFRACTIVE: {
      if ((double)fabs_(m.ior - 1) < 1e-3) return dot(n, i) * dot(n, o) <= 0 ? vec3f{1, 1, 1} : vec3f{0, 0, 0};
      auto t = delta_refractive_terms(m.ior, n, o, i);
      return t.same_side ? vec3f{1, 1, 1} * t.fresnel : vec3f{1, 1, 1} * rcp_(t.rel_ior * t.rel_ior) * (1 - t.fresnel);
    }
    case YTHIP_VOLUMETRIC: return dot(n, i) * dot(n, o) >= 0 ? vec3f{0, 0, 0} : vec3f{1, 1, 1};  // passthrough
    default: return {0, 0, 0};
  }
}
YT_FN float sample_delta_pdf(const material_point& m, vec3f n, vec3f o, vec3f i) {
  if (m.roughness != 0) return 0;
  switch (m.type) {
    case YTHIP_REFLECTIVE: return dot(n, i) * dot(n, o) <= 0 ? 0.0f : 1.0f;
    case YTHIP_TRANSPARENT: {
      auto no = dot(n, o);
      auto up = no <= 0 ? -n : n;
      auto fr = dielectric_fresnel(m.ior, dot(up, o));
      return dot(n, i) * no >= 0 ? fr : 1 - fr;
    }
    case YTHIP_REFRACTIVE: {
      if ((double)fabs_(m.ior - 1) < 1e-3) return dot(n, i) * dot(n, o) < 0 ? 1.0f : 0.0f;
      auto t = delta_refractive_terms(m.ior, n, o, i);
      return t.same_side ? t.fresnel : (1 - t.fresnel);
    }
    case YTHIP_VOLUMETRIC: return dot(n, i) * dot(n, o) >= 0 ? 0.0f : 1.0f;
    default: return 0;
  }
}
YT_FN vec3f sample_delta(const material_point& m, vec3f n, vec3f o, float rnl) {
  if (m.roughness != 0) return {0, 0, 0};
  switch (m.type) {
    case YTHIP_REFLECTIVE: return reflect(o, dot(n, o) <= 0 ? -n : n);
    case YTHIP_TRANSPARENT: {
      auto up = dot(n, o) <= 0 ? -n : n;
      return rnl < dielectric_fresnel(m.ior, dot(up, o)) ? reflect(o, up) : -o;
    }
    case YTHIP_REFRACTIVE: {
      if ((double)fabs_(m.ior - 1) < 1e-3) return -o;
      auto entering = dot(n, o) >= 0;
      auto up       = entering ? n : -n;
      auto rel_ior  = entering ? m.ior : rcp_(m.ior);
      return rnl < dielectric_fresnel(rel_ior, dot(up, o)) ? reflect(o, up) : refract(o, up, rcp_(rel_ior));
    }
    case YTHIP_VOLUMETRIC: return -o;  // sample_passthrough
    default: return {0, 0, 0};
  }
}

// eval_emission — yocto_trace.cpp:166-169
YT_FN vec3f eval_emission(const material_point& m, vec3f normal, vec3f outgoing) {
  return dot(normal, outgoing) >= 0 ? m.emission : vec3f{0, 0, 0};
}

// ---------------------------------------------------------------------------
// volumes — yocto_shading.h:1061-1111, yocto_trace.cpp:318-335
// ---------------------------------------------------------------------------
YT_FN vec3f eval_transmittance(vec3f density, float distance) { return exp_(-density * distance); }
YT_FN float sample_transmittance(vec3f density, float max_distance, float rl, float rd) {
  auto channel  = clamp_((int)(rl * 3), 0, 2);
  auto dch      = at(density, channel);
  auto distance = (dch == 0) ? flt_max : div_(-ytm::logf(1 - rd), dch);
  return min_(distance, max_distance);
}
YT_FN float sample_transmittance_pdf(vec3f density, float distance, float max_distance) {
  return distance < max_distance ? div_(sum(density * exp_(-density * distance)), 3) : div_(sum(exp_(-density * max_distance)), 3);
}
// Henyey-Greenstein
YT_FN float eval_phasefunction(float anisotropy, vec3f outgoing, vec3f incoming) {
  auto cosine = -dot(outgoing, incoming);
  auto denom  = 1 + anisotropy * anisotropy - 2 * anisotropy * cosine;
  return div_(1 - anisotropy * anisotropy, 4 * pif * denom * sqrt_(denom));
}
YT_FN vec3f sample_phasefunction(float anisotropy, vec3f outgoing, vec2f rn) {
  auto cos_theta = 0.0f;
  if (fabs_(anisotropy) < 1e-3f) {
    cos_theta = 1 - 2 * rn.y;
  } else {
    auto square = div_(1 - anisotropy * anisotropy, 1 + anisotropy - 2 * anisotropy * rn.y);
    cos_theta   = div_(1 + anisotropy * anisotropy - square * square, 2 * anisotropy);
  }
  auto  sin_theta = sqrt_(max_(0.0f, 1 - cos_theta * cos_theta));
  float sp, cp;
  ytm::sincosf(2 * pif * rn.x, &sp, &cp);
  return basis_fromz(-outgoing) * vec3f{sin_theta * cp, sin_theta * sp, cos_theta};
}

// The volume-stack entry (depth <= 1 in the reference: yocto_trace.cpp:545-553 pushes only when
// empty and pops otherwise).  Only the fields the volume branch reads are kept.
struct volume_point {
  vec3f density, scattering;
  float scanisotropy;
};
// eval_scattering + sample_scattering_pdf of one direction pair — yocto_trace.cpp:318-323, 331-335
// (the phase function is evaluated once; the pdf IS it)
YT_FN LobeEval eval_medium(const volume_point& v, vec3f outgoing, vec3f incoming) {
  if (v.density == vec3f{0, 0, 0}) return {{0, 0, 0}, 0};
  auto phase = eval_phasefunction(v.scanisotropy, outgoing, incoming);
  return {v.scattering * v.density * phase, phase};
}
YT_FN vec3f sample_scattering(const volume_point& v, vec3f outgoing, float rnl, vec2f rn) {  // :325-329
  if (v.density == vec3f{0, 0, 0}) return {0, 0, 0};
  return sample_phasefunction(v.scanisotropy, outgoing, rn);
}

}  // namespace yt
#ifdef YT_FAST
#pragma clang fp contract(off)
#endif

