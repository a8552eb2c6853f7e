// yt_trace_nee_cls.hip — k_trace for `pathdirect` / `pathmis` by scene class (round 6; yt_kernels.h: step_path's CLS).  Until
// round 5 only `path` was specialised and the NEE samplers always ran the general class — 316 / 390 spilled VGPRs even on an
// all-matte triangle scene.  The classes exist for the wide walk only, as for `path` (yt_trace_path.hip); a unit of its own
// so that it compiles next to yt_trace_nee.hip.
#include "yt_launch.h"

using namespace yt;

namespace ytl {

template <int S, int CLS>
static void launch_class(const Launch& l) {
  hipLaunchKernelGGL((k_trace<S, LP_DEFER, false, true, CLS>), dim3(l.blocks), dim3(YT_BLOCK), 0, l.stream, *l.ds, *l.st, *l.kp);
}

// 0 = launched; 1 = no class kernel for this launch (the caller runs launch_nee)
int launch_nee_class(const Launch& l) {
#if !defined(YT_DEV_ONLY_PATH) || defined(YT_DEV_NEE)
  if (l.count || !l.wide || l.cls < 1 || l.cls > 3) return 1;
  if (l.kp->sampler == YTHIP_SAMPLER_PATHDIRECT) {
    l.cls == 1 ? launch_class<YTHIP_SAMPLER_PATHDIRECT, 1>(l) : l.cls == 2 ? launch_class<YTHIP_SAMPLER_PATHDIRECT, 2>(l) : launch_class<YTHIP_SAMPLER_PATHDIRECT, 3>(l);
    return 0;
  }
  if (l.kp->sampler == YTHIP_SAMPLER_PATHMIS) {
    l.cls == 1 ? launch_class<YTHIP_SAMPLER_PATHMIS, 1>(l) : l.cls == 2 ? launch_class<YTHIP_SAMPLER_PATHMIS, 2>(l) : launch_class<YTHIP_SAMPLER_PATHMIS, 3>(l);
    return 0;
  }
#endif
  (void)l;
  return 1;
}

}  // namespace ytl
