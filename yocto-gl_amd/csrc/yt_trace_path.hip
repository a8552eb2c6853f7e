// yt_trace_path.hip — k_trace for `path` (the default sampler: yocto_trace.cpp:453-596) and `pathtest` (:937-1029).
// `path` comes in four scene classes (yt_kernels.h: step_path's CLS) x with / without the light-pdf walk stage; the
// classes 1-3 exist for the wide walk only (a scene the wide walk does not serve renders with the general kernels).
#include "yt_launch.h"

using namespace yt;

namespace ytl {

template <int LP, int CLS>
static void launch_class(const Launch& l) {
  hipLaunchKernelGGL((k_trace<YTHIP_SAMPLER_PATH, LP, false, true, CLS>), dim3(l.blocks), dim3(YT_BLOCK), 0, l.stream, *l.ds, *l.st,
      *l.kp);
}

int launch_path(const Launch& l) {
  const bool defer = l.lp == LP_DEFER;
  if (l.kp->sampler == YTHIP_SAMPLER_PATH) {
    const int cls = (!l.count && l.wide) ? l.cls : 0;
    switch (cls) {
      // 1: every material matte and untextured, every shape a triangle mesh — no other lobe, no volume code
      case 1: defer ? launch_class<LP_DEFER, 1>(l) : launch_class<LP_NONE, 1>(l); break;
      // 2: no material references a texture (any material types, any primitive kinds) — no texture / normal-map code
      case 2: defer ? launch_class<LP_DEFER, 2>(l) : launch_class<LP_NONE, 2>(l); break;
      // 3: "opaque textured" — matte / glossy / reflective materials, textures in the color and normal slots only, triangle
      //    and quad meshes (the scenes of the reference's own corpus): no transmission lobes, no volume code, two texture
      //    evaluators instead of five, no line / point intersectors
      case 3: defer ? launch_class<LP_DEFER, 3>(l) : launch_class<LP_NONE, 3>(l); break;
      default: defer ? launch_trace<YTHIP_SAMPLER_PATH, LP_DEFER>(l) : launch_trace<YTHIP_SAMPLER_PATH, LP_NONE>(l);
    }
    return 0;
  }
  if (l.kp->sampler == YTHIP_SAMPLER_PATHTEST) {
    defer ? launch_trace<YTHIP_SAMPLER_PATHTEST, LP_DEFER>(l) : launch_trace<YTHIP_SAMPLER_PATHTEST, LP_NONE>(l);
    return 0;
  }
  return 1;
}

}  // namespace ytl
