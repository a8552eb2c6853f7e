// yt_trace_nee.hip — k_trace for `pathdirect` (yocto_trace.cpp:599-722) and `pathmis` (:725-934): their NEE rays and
// light-pdf walks run in the walk stage (LP_DEFER).
#include "yt_launch.h"

using namespace yt;

namespace ytl {

int launch_nee(const Launch& l) {
#if !defined(YT_DEV_ONLY_PATH) || defined(YT_DEV_NEE)  // development builds (tools/devbuild.sh) leave these out unless asked: -DYT_DEV_NEE
  if (l.kp->sampler == YTHIP_SAMPLER_PATHDIRECT) {
    launch_trace<YTHIP_SAMPLER_PATHDIRECT, LP_DEFER>(l);
    return 0;
  }
  if (l.kp->sampler == YTHIP_SAMPLER_PATHMIS) {
    launch_trace<YTHIP_SAMPLER_PATHMIS, LP_DEFER>(l);
    return 0;
  }
#endif
  (void)l;
  return 1;
}

}  // namespace ytl
