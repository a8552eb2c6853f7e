// yt_material.h — material_point (yocto_scene.h:258-270): what eval_material hands to the lobes.
// On its own so that yt_shading.h compiles without the device scene (host build of the lobe
// arithmetic for tests/cpp/shading_check.cpp).
#pragma once

#include "yt_math.h"

namespace yt {

struct material_point {
  int   type;
  vec3f emission, color;
  float opacity, roughness, metallic, ior;
  vec3f density, scattering;
  float scanisotropy, trdepth;
};
constexpr float min_roughness = 0.03f * 0.03f;

}  // namespace yt
