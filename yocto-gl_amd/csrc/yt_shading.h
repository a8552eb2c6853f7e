// yt_shading.h — device restatement of the BSDF lobes of
// libs/yocto/yocto_shading.h (eval / sample / pdf for matte, glossy,
// reflective, transparent, refractive, gltfpbr, passthrough; transmittance and
// the Henyey-Greenstein phase function) and of the material dispatch of
// libs/yocto/yocto_trace.cpp:166-335.  Expression trees are kept identical to
// the reference (association order matters for parity).
#pragma once

#include "yt_scene.h"

namespace yt {

// yocto_shading.h:302-305
YT_FN bool same_hemisphere(vec3f normal, vec3f outgoing, vec3f incoming) {
  return dot(normal, outgoing) * dot(normal, incoming) >= 0;
}
// fresnel_schlick — :308-315
YT_FN vec3f fresnel_schlick(vec3f specular, vec3f normal, vec3f outgoing) {
  if (specular == vec3f{0, 0, 0}) return {0, 0, 0};
  auto cosine = dot(normal, outgoing);
  return specular + (1 - specular) * ytm::powf(clamp_(1 - fabs_(cosine), 0.0f, 1.0f), 5.0f);
}
// fresnel_dielectric — :318-338
YT_FN float fresnel_dielectric(float eta, vec3f normal, vec3f outgoing) {
  auto cosw  = fabs_(dot(normal, outgoing));
  auto sin2  = 1 - cosw * cosw;
  auto eta2  = eta * eta;
  auto cos2t = 1 - sin2 / eta2;
  if (cos2t < 0) return 1;  // tir
  auto t0 = sqrt_(cos2t);
  auto t1 = eta * t0;
  auto t2 = eta * cosw;
  auto rs = (cosw - t1) / (cosw + t1);
  auto rp = (t0 - t2) / (t0 + t2);
  return (rs * rs + rp * rp) / 2;
}
// fresnel_conductor — :341-366
YT_FN vec3f fresnel_conductor(vec3f eta, vec3f etak, vec3f normal, vec3f outgoing) {
  auto cosw = dot(normal, outgoing);
  if (cosw <= 0) return {0, 0, 0};
  cosw       = clamp_(cosw, (float)-1, (float)1);
  auto cos2  = cosw * cosw;
  auto sin2  = clamp_(1 - cos2, (float)0, (float)1);
  auto eta2  = eta * eta;
  auto etak2 = etak * etak;

  auto t0       = eta2 - etak2 - sin2;
  auto a2plusb2 = sqrt_(t0 * t0 + 4 * eta2 * etak2);
  auto t1       = a2plusb2 + cos2;
  auto a        = sqrt_((a2plusb2 + t0) / 2);
  auto t2       = 2 * a * cosw;
  auto rs       = (t1 - t2) / (t1 + t2);

  auto t3 = cos2 * a2plusb2 + sin2 * sin2;
  auto t4 = t2 * sin2;
  auto rp = rs * (t3 - t4) / (t3 + t4);

  return (rp + rs) / 2;
}
// eta_to_reflectivity / reflectivity_to_eta — :369-376
YT_FN vec3f eta_to_reflectivity(vec3f eta) { return ((eta - 1) * (eta - 1)) / ((eta + 1) * (eta + 1)); }
YT_FN vec3f reflectivity_to_eta(vec3f reflectivity_) {
  auto reflectivity = clamp_(reflectivity_, 0.0f, 0.99f);
  return (1 + sqrt_(reflectivity)) / (1 - sqrt_(reflectivity));
}

// microfacet_distribution (ggx) — :409-424
YT_FN float microfacet_distribution(float roughness, vec3f normal, vec3f halfway) {
  auto cosine = dot(normal, halfway);
  if (cosine <= 0) return 0;
  auto roughness2 = roughness * roughness;
  auto cosine2    = cosine * cosine;
  return roughness2 /
         (pif * (cosine2 * roughness2 + 1 - cosine2) * (cosine2 * roughness2 + 1 - cosine2));
}
// microfacet_shadowing1 (ggx) — :427-447
YT_FN float microfacet_shadowing1(float roughness, vec3f normal, vec3f halfway, vec3f direction) {
  auto cosine  = dot(normal, direction);
  auto cosineh = dot(halfway, direction);
  if (cosine * cosineh <= 0) return 0;
  auto roughness2 = roughness * roughness;
  auto cosine2    = cosine * cosine;
  return 2 * fabs_(cosine) / (fabs_(cosine) + sqrt_(cosine2 - roughness2 * cosine2 + roughness2));
}
YT_FN float microfacet_shadowing(float roughness, vec3f normal, vec3f halfway, vec3f outgoing,
    vec3f incoming) {  // :450-455
  return microfacet_shadowing1(roughness, normal, halfway, outgoing) *
         microfacet_shadowing1(roughness, normal, halfway, incoming);
}
// sample_microfacet (ggx) — :458-471
YT_FN vec3f sample_microfacet(float roughness, vec3f normal, vec2f rn) {
  auto phi   = 2 * pif * rn.x;
  auto theta = ytm::atanf(roughness * sqrt_(rn.y / (1 - rn.y)));
  float sp, cp, st, ct;
  ytm::sincosf(phi, &sp, &cp);
  ytm::sincosf(theta, &st, &ct);
  auto local_half_vector = vec3f{cp * st, sp * st, ct};
  return transform_direction(basis_fromz(normal), local_half_vector);
}
// sample_microfacet_pdf — :474-479
YT_FN float sample_microfacet_pdf(float roughness, vec3f normal, vec3f halfway) {
  auto cosine = dot(normal, halfway);
  if (cosine < 0) return 0;
  return microfacet_distribution(roughness, normal, halfway) * cosine;
}

// ---- matte — :554-573 --------------------------------------------------------
YT_FN vec3f eval_matte(vec3f color, vec3f normal, vec3f outgoing, vec3f incoming) {
  if (dot(normal, incoming) * dot(normal, outgoing) <= 0) return {0, 0, 0};
  return color / pif * fabs_(dot(normal, incoming));
}
YT_FN vec3f sample_matte(vec3f color, vec3f normal, vec3f outgoing, vec2f rn) {
  auto up_normal = dot(normal, outgoing) <= 0 ? -normal : normal;
  return sample_hemisphere_cos(up_normal, rn);
}
YT_FN float sample_matte_pdf(vec3f color, vec3f normal, vec3f outgoing, vec3f incoming) {
  if (dot(normal, incoming) * dot(normal, outgoing) <= 0) return 0;
  auto up_normal = dot(normal, outgoing) <= 0 ? -normal : normal;
  return sample_hemisphere_cos_pdf(up_normal, incoming);
}

// ---- glossy — :576-617 -------------------------------------------------------
YT_FN vec3f eval_glossy(vec3f color, float ior, float roughness, vec3f normal, vec3f outgoing,
    vec3f incoming) {
  if (dot(normal, incoming) * dot(normal, outgoing) <= 0) return {0, 0, 0};
  auto up_normal = dot(normal, outgoing) <= 0 ? -normal : normal;
  auto F1        = fresnel_dielectric(ior, up_normal, outgoing);
  auto halfway   = normalize(incoming + outgoing);
  auto F         = fresnel_dielectric(ior, halfway, incoming);
  auto D         = microfacet_distribution(roughness, up_normal, halfway);
  auto G         = microfacet_shadowing(roughness, up_normal, halfway, outgoing, incoming);
  return color * (1 - F1) / pif * fabs_(dot(up_normal, incoming)) +
         vec3f{1, 1, 1} * F * D * G / (4 * dot(up_normal, outgoing) * dot(up_normal, incoming)) *
             fabs_(dot(up_normal, incoming));
}
YT_FN vec3f sample_glossy(vec3f color, float ior, float roughness, vec3f normal, vec3f outgoing,
    float rnl, vec2f rn) {
  auto up_normal = dot(normal, outgoing) <= 0 ? -normal : normal;
  if (rnl < fresnel_dielectric(ior, up_normal, outgoing)) {
    auto halfway  = sample_microfacet(roughness, up_normal, rn);
    auto incoming = reflect(outgoing, halfway);
    if (!same_hemisphere(up_normal, outgoing, incoming)) return {0, 0, 0};
    return incoming;
  } else {
    return sample_hemisphere_cos(up_normal, rn);
  }
}
YT_FN float sample_glossy_pdf(vec3f color, float ior, float roughness, vec3f normal, vec3f outgoing,
    vec3f incoming) {
  if (dot(normal, incoming) * dot(normal, outgoing) <= 0) return 0;
  auto up_normal = dot(normal, outgoing) <= 0 ? -normal : normal;
  auto halfway   = normalize(outgoing + incoming);
  auto F         = fresnel_dielectric(ior, up_normal, outgoing);
  return F * sample_microfacet_pdf(roughness, up_normal, halfway) / (4 * fabs_(dot(outgoing, halfway))) +
         (1 - F) * sample_hemisphere_cos_pdf(up_normal, incoming);
}

// ---- reflective (rough, color-parametrised) — :620-650 -----------------------
YT_FN vec3f eval_reflective(vec3f color, float roughness, vec3f normal, vec3f outgoing,
    vec3f incoming) {
  if (dot(normal, incoming) * dot(normal, outgoing) <= 0) return {0, 0, 0};
  auto up_normal = dot(normal, outgoing) <= 0 ? -normal : normal;
  auto halfway   = normalize(incoming + outgoing);
  auto F         = fresnel_conductor(reflectivity_to_eta(color), {0, 0, 0}, halfway, incoming);
  auto D         = microfacet_distribution(roughness, up_normal, halfway);
  auto G         = microfacet_shadowing(roughness, up_normal, halfway, outgoing, incoming);
  return F * D * G / (4 * dot(up_normal, outgoing) * dot(up_normal, incoming)) *
         fabs_(dot(up_normal, incoming));
}
YT_FN vec3f sample_reflective(vec3f color, float roughness, vec3f normal, vec3f outgoing, vec2f rn) {
  auto up_normal = dot(normal, outgoing) <= 0 ? -normal : normal;
  auto halfway   = sample_microfacet(roughness, up_normal, rn);
  auto incoming  = reflect(outgoing, halfway);
  if (!same_hemisphere(up_normal, outgoing, incoming)) return {0, 0, 0};
  return incoming;
}
YT_FN float sample_reflective_pdf(vec3f color, float roughness, vec3f normal, vec3f outgoing,
    vec3f incoming) {
  if (dot(normal, incoming) * dot(normal, outgoing) <= 0) return 0;
  auto up_normal = dot(normal, outgoing) <= 0 ? -normal : normal;
  auto halfway   = normalize(outgoing + incoming);
  return sample_microfacet_pdf(roughness, up_normal, halfway) / (4 * fabs_(dot(outgoing, halfway)));
}
// ---- reflective (delta) — :692-713 -------------------------------------------
YT_FN vec3f eval_reflective_delta(vec3f color, vec3f normal, vec3f outgoing, vec3f incoming) {
  if (dot(normal, incoming) * dot(normal, outgoing) <= 0) return {0, 0, 0};
  auto up_normal = dot(normal, outgoing) <= 0 ? -normal : normal;
  return fresnel_conductor(reflectivity_to_eta(color), {0, 0, 0}, up_normal, outgoing);
}
YT_FN vec3f sample_reflective_delta(vec3f color, vec3f normal, vec3f outgoing) {
  auto up_normal = dot(normal, outgoing) <= 0 ? -normal : normal;
  return reflect(outgoing, up_normal);
}
YT_FN float sample_reflective_delta_pdf(vec3f color, vec3f normal, vec3f outgoing, vec3f incoming) {
  if (dot(normal, incoming) * dot(normal, outgoing) <= 0) return 0;
  return 1;
}

// ---- gltfpbr — :739-789 ------------------------------------------------------
YT_FN vec3f eval_gltfpbr(vec3f color, float ior, float roughness, float metallic, vec3f normal,
    vec3f outgoing, vec3f incoming) {
  if (dot(normal, incoming) * dot(normal, outgoing) <= 0) return {0, 0, 0};
  auto reflectivity = lerp_(eta_to_reflectivity(vec3f{ior, ior, ior}), color, metallic);
  auto up_normal    = dot(normal, outgoing) <= 0 ? -normal : normal;
  auto F1           = fresnel_schlick(reflectivity, up_normal, outgoing);
  auto halfway      = normalize(incoming + outgoing);
  auto F            = fresnel_schlick(reflectivity, halfway, incoming);
  auto D            = microfacet_distribution(roughness, up_normal, halfway);
  auto G            = microfacet_shadowing(roughness, up_normal, halfway, outgoing, incoming);
  return color * (1 - metallic) * (1 - F1) / pif * fabs_(dot(up_normal, incoming)) +
         F * D * G / (4 * dot(up_normal, outgoing) * dot(up_normal, incoming)) *
             fabs_(dot(up_normal, incoming));
}
YT_FN vec3f sample_gltfpbr(vec3f color, float ior, float roughness, float metallic, vec3f normal,
    vec3f outgoing, float rnl, vec2f rn) {
  auto up_normal    = dot(normal, outgoing) <= 0 ? -normal : normal;
  auto reflectivity = lerp_(eta_to_reflectivity(vec3f{ior, ior, ior}), color, metallic);
  if (rnl < mean(fresnel_schlick(reflectivity, up_normal, outgoing))) {
    auto halfway  = sample_microfacet(roughness, up_normal, rn);
    auto incoming = reflect(outgoing, halfway);
    if (!same_hemisphere(up_normal, outgoing, incoming)) return {0, 0, 0};
    return incoming;
  } else {
    return sample_hemisphere_cos(up_normal, rn);
  }
}
YT_FN float sample_gltfpbr_pdf(vec3f color, float ior, float roughness, float metallic, vec3f normal,
    vec3f outgoing, vec3f incoming) {
  if (dot(normal, incoming) * dot(normal, outgoing) <= 0) return 0;
  auto up_normal    = dot(normal, outgoing) <= 0 ? -normal : normal;
  auto halfway      = normalize(outgoing + incoming);
  auto reflectivity = lerp_(eta_to_reflectivity(vec3f{ior, ior, ior}), color, metallic);
  auto F            = mean(fresnel_schlick(reflectivity, up_normal, outgoing));
  return F * sample_microfacet_pdf(roughness, up_normal, halfway) / (4 * fabs_(dot(outgoing, halfway))) +
         (1 - F) * sample_hemisphere_cos_pdf(up_normal, incoming);
}

// ---- transparent (rough) — :792-846 ------------------------------------------
YT_FN vec3f eval_transparent(vec3f color, float ior, float roughness, vec3f normal, vec3f outgoing,
    vec3f incoming) {
  auto up_normal = dot(normal, outgoing) <= 0 ? -normal : normal;
  if (dot(normal, incoming) * dot(normal, outgoing) >= 0) {
    auto halfway = normalize(incoming + outgoing);
    auto F       = fresnel_dielectric(ior, halfway, outgoing);
    auto D       = microfacet_distribution(roughness, up_normal, halfway);
    auto G       = microfacet_shadowing(roughness, up_normal, halfway, outgoing, incoming);
    return vec3f{1, 1, 1} * F * D * G / (4 * dot(up_normal, outgoing) * dot(up_normal, incoming)) *
           fabs_(dot(up_normal, incoming));
  } else {
    auto reflected = reflect(-incoming, up_normal);
    auto halfway   = normalize(reflected + outgoing);
    auto F         = fresnel_dielectric(ior, halfway, outgoing);
    auto D         = microfacet_distribution(roughness, up_normal, halfway);
    auto G         = microfacet_shadowing(roughness, up_normal, halfway, outgoing, reflected);
    return color * (1 - F) * D * G / (4 * dot(up_normal, outgoing) * dot(up_normal, reflected)) *
           (fabs_(dot(up_normal, reflected)));
  }
}
YT_FN vec3f sample_transparent(vec3f color, float ior, float roughness, vec3f normal, vec3f outgoing,
    float rnl, vec2f rn) {
  auto up_normal = dot(normal, outgoing) <= 0 ? -normal : normal;
  auto halfway   = sample_microfacet(roughness, up_normal, rn);
  if (rnl < fresnel_dielectric(ior, halfway, outgoing)) {
    auto incoming = reflect(outgoing, halfway);
    if (!same_hemisphere(up_normal, outgoing, incoming)) return {0, 0, 0};
    return incoming;
  } else {
    auto reflected = reflect(outgoing, halfway);
    auto incoming  = -reflect(reflected, up_normal);
    if (same_hemisphere(up_normal, outgoing, incoming)) return {0, 0, 0};
    return incoming;
  }
}
YT_FN float sample_transparent_pdf(vec3f color, float ior, float roughness, vec3f normal,
    vec3f outgoing, vec3f incoming) {
  auto up_normal = dot(normal, outgoing) <= 0 ? -normal : normal;
  if (dot(normal, incoming) * dot(normal, outgoing) >= 0) {
    auto halfway = normalize(incoming + outgoing);
    return fresnel_dielectric(ior, halfway, outgoing) *
           sample_microfacet_pdf(roughness, up_normal, halfway) / (4 * fabs_(dot(outgoing, halfway)));
  } else {
    auto reflected = reflect(-incoming, up_normal);
    auto halfway   = normalize(reflected + outgoing);
    auto d         = (1 - fresnel_dielectric(ior, halfway, outgoing)) *
             sample_microfacet_pdf(roughness, up_normal, halfway);
    return d / (4 * fabs_(dot(outgoing, halfway)));
  }
}
// ---- transparent (delta) — :849-878 ------------------------------------------
YT_FN vec3f eval_transparent_delta(vec3f color, float ior, vec3f normal, vec3f outgoing,
    vec3f incoming) {
  auto up_normal = dot(normal, outgoing) <= 0 ? -normal : normal;
  if (dot(normal, incoming) * dot(normal, outgoing) >= 0) {
    return vec3f{1, 1, 1} * fresnel_dielectric(ior, up_normal, outgoing);
  } else {
    return color * (1 - fresnel_dielectric(ior, up_normal, outgoing));
  }
}
YT_FN vec3f sample_transparent_delta(vec3f color, float ior, vec3f normal, vec3f outgoing, float rnl) {
  auto up_normal = dot(normal, outgoing) <= 0 ? -normal : normal;
  if (rnl < fresnel_dielectric(ior, up_normal, outgoing)) {
    return reflect(outgoing, up_normal);
  } else {
    return -outgoing;
  }
}
YT_FN float sample_transparent_delta_pdf(vec3f color, float ior, vec3f normal, vec3f outgoing,
    vec3f incoming) {
  auto up_normal = dot(normal, outgoing) <= 0 ? -normal : normal;
  if (dot(normal, incoming) * dot(normal, outgoing) >= 0) {
    return fresnel_dielectric(ior, up_normal, outgoing);
  } else {
    return 1 - fresnel_dielectric(ior, up_normal, outgoing);
  }
}

// ---- refractive (rough) — :881-956 -------------------------------------------
YT_FN vec3f eval_refractive(vec3f color, float ior, float roughness, vec3f normal, vec3f outgoing,
    vec3f incoming) {
  auto entering  = dot(normal, outgoing) >= 0;
  auto up_normal = entering ? normal : -normal;
  auto rel_ior   = entering ? ior : (1 / ior);
  if (dot(normal, incoming) * dot(normal, outgoing) >= 0) {
    auto halfway = normalize(incoming + outgoing);
    auto F       = fresnel_dielectric(rel_ior, halfway, outgoing);
    auto D       = microfacet_distribution(roughness, up_normal, halfway);
    auto G       = microfacet_shadowing(roughness, up_normal, halfway, outgoing, incoming);
    return vec3f{1, 1, 1} * F * D * G / fabs_(4 * dot(normal, outgoing) * dot(normal, incoming)) *
           fabs_(dot(normal, incoming));
  } else {
    auto halfway = -normalize(rel_ior * incoming + outgoing) * (entering ? 1.0f : -1.0f);
    auto F       = fresnel_dielectric(rel_ior, halfway, outgoing);
    auto D       = microfacet_distribution(roughness, up_normal, halfway);
    auto G       = microfacet_shadowing(roughness, up_normal, halfway, outgoing, incoming);
    // [Walter 2007] equation 21
    return vec3f{1, 1, 1} *
           fabs_((dot(outgoing, halfway) * dot(incoming, halfway)) /
                 (dot(outgoing, normal) * dot(incoming, normal))) *
           (1 - F) * D * G / sqr_(rel_ior * dot(halfway, incoming) + dot(halfway, outgoing)) *
           fabs_(dot(normal, incoming));
  }
}
YT_FN vec3f sample_refractive(vec3f color, float ior, float roughness, vec3f normal, vec3f outgoing,
    float rnl, vec2f rn) {
  auto entering  = dot(normal, outgoing) >= 0;
  auto up_normal = entering ? normal : -normal;
  auto halfway   = sample_microfacet(roughness, up_normal, rn);
  if (rnl < fresnel_dielectric(entering ? ior : (1 / ior), halfway, outgoing)) {
    auto incoming = reflect(outgoing, halfway);
    if (!same_hemisphere(up_normal, outgoing, incoming)) return {0, 0, 0};
    return incoming;
  } else {
    auto incoming = refract(outgoing, halfway, entering ? (1 / ior) : ior);
    if (same_hemisphere(up_normal, outgoing, incoming)) return {0, 0, 0};
    return incoming;
  }
}
YT_FN float sample_refractive_pdf(vec3f color, float ior, float roughness, vec3f normal,
    vec3f outgoing, vec3f incoming) {
  auto entering  = dot(normal, outgoing) >= 0;
  auto up_normal = entering ? normal : -normal;
  auto rel_ior   = entering ? ior : (1 / ior);
  if (dot(normal, incoming) * dot(normal, outgoing) >= 0) {
    auto halfway = normalize(incoming + outgoing);
    return fresnel_dielectric(rel_ior, halfway, outgoing) *
           sample_microfacet_pdf(roughness, up_normal, halfway) / (4 * fabs_(dot(outgoing, halfway)));
  } else {
    auto halfway = -normalize(rel_ior * incoming + outgoing) * (entering ? 1.0f : -1.0f);
    // [Walter 2007] equation 17
    return (1 - fresnel_dielectric(rel_ior, halfway, outgoing)) *
           sample_microfacet_pdf(roughness, up_normal, halfway) * fabs_(dot(halfway, incoming)) /
           sqr_(rel_ior * dot(halfway, incoming) + dot(halfway, outgoing));
  }
}
// ---- refractive (delta) — :959-1005 (`abs(ior-1) < 1e-3` compares in double) ---
YT_FN vec3f eval_refractive_delta(vec3f color, float ior, vec3f normal, vec3f outgoing,
    vec3f incoming) {
  if ((double)fabs_(ior - 1) < 1e-3)
    return dot(normal, incoming) * dot(normal, outgoing) <= 0 ? vec3f{1, 1, 1} : vec3f{0, 0, 0};
  auto entering  = dot(normal, outgoing) >= 0;
  auto up_normal = entering ? normal : -normal;
  auto rel_ior   = entering ? ior : (1 / ior);
  if (dot(normal, incoming) * dot(normal, outgoing) >= 0) {
    return vec3f{1, 1, 1} * fresnel_dielectric(rel_ior, up_normal, outgoing);
  } else {
    return vec3f{1, 1, 1} * (1 / (rel_ior * rel_ior)) *
           (1 - fresnel_dielectric(rel_ior, up_normal, outgoing));
  }
}
YT_FN vec3f sample_refractive_delta(vec3f color, float ior, vec3f normal, vec3f outgoing, float rnl) {
  if ((double)fabs_(ior - 1) < 1e-3) return -outgoing;
  auto entering  = dot(normal, outgoing) >= 0;
  auto up_normal = entering ? normal : -normal;
  auto rel_ior   = entering ? ior : (1 / ior);
  if (rnl < fresnel_dielectric(rel_ior, up_normal, outgoing)) {
    return reflect(outgoing, up_normal);
  } else {
    return refract(outgoing, up_normal, 1 / rel_ior);
  }
}
YT_FN float sample_refractive_delta_pdf(vec3f color, float ior, vec3f normal, vec3f outgoing,
    vec3f incoming) {
  if ((double)fabs_(ior - 1) < 1e-3)
    return dot(normal, incoming) * dot(normal, outgoing) < 0 ? 1.0f : 0.0f;
  auto entering  = dot(normal, outgoing) >= 0;
  auto up_normal = entering ? normal : -normal;
  auto rel_ior   = entering ? ior : (1 / ior);
  if (dot(normal, incoming) * dot(normal, outgoing) >= 0) {
    return fresnel_dielectric(rel_ior, up_normal, outgoing);
  } else {
    return (1 - fresnel_dielectric(rel_ior, up_normal, outgoing));
  }
}
// ---- passthrough — :1029-1053 --------------------------------------------------
YT_FN vec3f eval_passthrough(vec3f color, vec3f normal, vec3f outgoing, vec3f incoming) {
  if (dot(normal, incoming) * dot(normal, outgoing) >= 0) return vec3f{0, 0, 0};
  return vec3f{1, 1, 1};
}
YT_FN float sample_passthrough_pdf(vec3f color, vec3f normal, vec3f outgoing, vec3f incoming) {
  if (dot(normal, incoming) * dot(normal, outgoing) >= 0) return 0;
  return 1;
}

// ---- volumes — :1061-1111 --------------------------------------------------------
YT_FN vec3f eval_transmittance(vec3f density, float distance) { return exp_(-density * distance); }
YT_FN float sample_transmittance(vec3f density, float max_distance, float rl, float rd) {
  auto channel  = clamp_((int)(rl * 3), 0, 2);
  auto dch      = at(density, channel);
  auto distance = (dch == 0) ? flt_max : -ytm::logf(1 - rd) / dch;
  return min_(distance, max_distance);
}
YT_FN float sample_transmittance_pdf(vec3f density, float distance, float max_distance) {
  if (distance < max_distance) {
    return sum(density * exp_(-density * distance)) / 3;
  } else {
    return sum(exp_(-density * max_distance)) / 3;
  }
}
YT_FN float eval_phasefunction(float anisotropy, vec3f outgoing, vec3f incoming) {
  auto cosine = -dot(outgoing, incoming);
  auto denom  = 1 + anisotropy * anisotropy - 2 * anisotropy * cosine;
  return (1 - anisotropy * anisotropy) / (4 * pif * denom * sqrt_(denom));
}
YT_FN vec3f sample_phasefunction(float anisotropy, vec3f outgoing, vec2f rn) {
  auto cos_theta = 0.0f;
  if (fabs_(anisotropy) < 1e-3f) {
    cos_theta = 1 - 2 * rn.y;
  } else {
    auto square = (1 - anisotropy * anisotropy) / (1 + anisotropy - 2 * anisotropy * rn.y);
    cos_theta   = (1 + anisotropy * anisotropy - square * square) / (2 * anisotropy);
  }
  auto sin_theta      = sqrt_(max_(0.0f, 1 - cos_theta * cos_theta));
  auto phi            = 2 * pif * rn.x;
  float sp, cp;
  ytm::sincosf(phi, &sp, &cp);
  auto local_incoming = vec3f{sin_theta * cp, sin_theta * sp, cos_theta};
  return basis_fromz(-outgoing) * local_incoming;
}

// ===========================================================================
// Material dispatch — libs/yocto/yocto_trace.cpp:166-335
// ===========================================================================
YT_FN vec3f eval_emission(const material_point& m, vec3f normal, vec3f outgoing) {  // :166
  return dot(normal, outgoing) >= 0 ? m.emission : vec3f{0, 0, 0};
}
YT_FN vec3f eval_bsdfcos(const material_point& m, vec3f n, vec3f o, vec3f i) {  // :172-199
  if (m.roughness == 0) return {0, 0, 0};
  switch (m.type) {
    case YTHIP_MATTE: return eval_matte(m.color, n, o, i);
    case YTHIP_GLOSSY: return eval_glossy(m.color, m.ior, m.roughness, n, o, i);
    case YTHIP_REFLECTIVE: return eval_reflective(m.color, m.roughness, n, o, i);
    case YTHIP_TRANSPARENT: return eval_transparent(m.color, m.ior, m.roughness, n, o, i);
    case YTHIP_REFRACTIVE:
    case YTHIP_SUBSURFACE: return eval_refractive(m.color, m.ior, m.roughness, n, o, i);
    case YTHIP_GLTFPBR: return eval_gltfpbr(m.color, m.ior, m.roughness, m.metallic, n, o, i);
    default: return {0, 0, 0};
  }
}
YT_FN vec3f eval_delta(const material_point& m, vec3f n, vec3f o, vec3f i) {  // :201-218
  if (m.roughness != 0) return {0, 0, 0};
  switch (m.type) {
    case YTHIP_REFLECTIVE: return eval_reflective_delta(m.color, n, o, i);
    case YTHIP_TRANSPARENT: return eval_transparent_delta(m.color, m.ior, n, o, i);
    case YTHIP_REFRACTIVE: return eval_refractive_delta(m.color, m.ior, n, o, i);
    case YTHIP_VOLUMETRIC: return eval_passthrough(m.color, n, o, i);
    default: return {0, 0, 0};
  }
}
YT_FN vec3f sample_bsdfcos(const material_point& m, vec3f n, vec3f o, float rnl, vec2f rn) {  // :221-248
  if (m.roughness == 0) return {0, 0, 0};
  switch (m.type) {
    case YTHIP_MATTE: return sample_matte(m.color, n, o, rn);
    case YTHIP_GLOSSY: return sample_glossy(m.color, m.ior, m.roughness, n, o, rnl, rn);
    case YTHIP_REFLECTIVE: return sample_reflective(m.color, m.roughness, n, o, rn);
    case YTHIP_TRANSPARENT: return sample_transparent(m.color, m.ior, m.roughness, n, o, rnl, rn);
    case YTHIP_REFRACTIVE:
    case YTHIP_SUBSURFACE: return sample_refractive(m.color, m.ior, m.roughness, n, o, rnl, rn);
    case YTHIP_GLTFPBR: return sample_gltfpbr(m.color, m.ior, m.roughness, m.metallic, n, o, rnl, rn);
    default: return {0, 0, 0};
  }
}
YT_FN vec3f sample_delta(const material_point& m, vec3f n, vec3f o, float rnl) {  // :250-267
  if (m.roughness != 0) return {0, 0, 0};
  switch (m.type) {
    case YTHIP_REFLECTIVE: return sample_reflective_delta(m.color, n, o);
    case YTHIP_TRANSPARENT: return sample_transparent_delta(m.color, m.ior, n, o, rnl);
    case YTHIP_REFRACTIVE: return sample_refractive_delta(m.color, m.ior, n, o, rnl);
    case YTHIP_VOLUMETRIC: return -o;  // sample_passthrough
    default: return {0, 0, 0};
  }
}
YT_FN float sample_bsdfcos_pdf(const material_point& m, vec3f n, vec3f o, vec3f i) {  // :270-297
  if (m.roughness == 0) return 0;
  switch (m.type) {
    case YTHIP_MATTE: return sample_matte_pdf(m.color, n, o, i);
    case YTHIP_GLOSSY: return sample_glossy_pdf(m.color, m.ior, m.roughness, n, o, i);
    case YTHIP_REFLECTIVE: return sample_reflective_pdf(m.color, m.roughness, n, o, i);
    case YTHIP_TRANSPARENT: return sample_transparent_pdf(m.color, m.ior, m.roughness, n, o, i);
    case YTHIP_REFRACTIVE:
    case YTHIP_SUBSURFACE: return sample_refractive_pdf(m.color, m.ior, m.roughness, n, o, i);
    case YTHIP_GLTFPBR: return sample_gltfpbr_pdf(m.color, m.ior, m.roughness, m.metallic, n, o, i);
    default: return 0;
  }
}
YT_FN float sample_delta_pdf(const material_point& m, vec3f n, vec3f o, vec3f i) {  // :299-316
  if (m.roughness != 0) return 0;
  switch (m.type) {
    case YTHIP_REFLECTIVE: return sample_reflective_delta_pdf(m.color, n, o, i);
    case YTHIP_TRANSPARENT: return sample_transparent_delta_pdf(m.color, m.ior, n, o, i);
    case YTHIP_REFRACTIVE: return sample_refractive_delta_pdf(m.color, m.ior, n, o, i);
    case YTHIP_VOLUMETRIC: return sample_passthrough_pdf(m.color, n, o, i);
    default: return 0;
  }
}

// The volume-stack entry (depth <= 1 in the reference: yocto_trace.cpp:545-553
// pushes only when empty and pops otherwise).  Only the fields the volume branch
// reads are kept.
struct volume_point {
  vec3f density, scattering;
  float scanisotropy;
};
YT_FN vec3f eval_scattering(const volume_point& v, vec3f outgoing, vec3f incoming) {  // :318-323
  if (v.density == vec3f{0, 0, 0}) return {0, 0, 0};
  return v.scattering * v.density * eval_phasefunction(v.scanisotropy, outgoing, incoming);
}
YT_FN vec3f sample_scattering(const volume_point& v, vec3f outgoing, float rnl, vec2f rn) {  // :325-329
  if (v.density == vec3f{0, 0, 0}) return {0, 0, 0};
  return sample_phasefunction(v.scanisotropy, outgoing, rn);
}
YT_FN float sample_scattering_pdf(const volume_point& v, vec3f outgoing, vec3f incoming) {  // :331-335
  if (v.density == vec3f{0, 0, 0}) return 0;
  return eval_phasefunction(v.scanisotropy, outgoing, incoming);
}

}  // namespace yt
