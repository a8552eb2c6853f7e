// yt_trace_misc.hip — k_trace for `naive` (yocto_trace.cpp:1032-1108), `eyelight` / `diagram` (:1111-1244), `furnace`
// (:1247-1338) and `falsecolor` (:1341-1419): no light pdf, no walk stage.
#include "yt_launch.h"

using namespace yt;

namespace ytl {

int launch_misc(const Launch& l) {
  switch (l.kp->sampler) {
    case YTHIP_SAMPLER_NAIVE: launch_trace<YTHIP_SAMPLER_NAIVE, LP_NONE>(l); return 0;
#ifndef YT_DEV_ONLY_PATH  // development builds (tools/devbuild.sh): path / pathtest / naive only
    case YTHIP_SAMPLER_EYELIGHT: launch_trace<YTHIP_SAMPLER_EYELIGHT, LP_NONE>(l); return 0;
    case YTHIP_SAMPLER_DIAGRAM: launch_trace<YTHIP_SAMPLER_DIAGRAM, LP_NONE>(l); return 0;
    case YTHIP_SAMPLER_FURNACE: launch_trace<YTHIP_SAMPLER_FURNACE, LP_NONE>(l); return 0;
    case YTHIP_SAMPLER_FALSECOLOR: launch_trace<YTHIP_SAMPLER_FALSECOLOR, LP_NONE>(l); return 0;
#endif
    default: return 1;
  }
}

}  // namespace ytl
