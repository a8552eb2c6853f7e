// yt_launch.h — the k_trace instantiations, compiled in three units by sampler family so that they build in parallel
// and an edit of one family does not recompile the others:
//   yt_trace_path.hip   path (scene classes 0-3, with / without the light-pdf walk stage) and pathtest
//   yt_trace_nee.hip    pathdirect, pathmis (next-event estimation: NEE rays + light-pdf walks in the walk stage)
//   yt_trace_nee_cls.hip  the same two by scene class (1-3), wide walk
//   yt_trace_misc.hip   naive, eyelight, diagram, furnace, falsecolor
// Each exports one launcher; 0 = launched, 1 = not a sampler of this unit.
#pragma once

#include "yt_kernels.h"

namespace ytl {

struct Launch {
  hipStream_t        stream;
  int                blocks;  // workgroups: one persistent one-wave workgroup per 16x4 tile (or the pixel pool's count)
  const yt::DScene*  ds;
  const yt::DState*  st;
  const yt::KParams* kp;
  bool               count;  // the work-counting launch: binary walk, the reference's node / primitive counts
  bool               wide;   // the wide (grandchildren-record) walk
  int                lp;     // LP_NONE / LP_DEFER: does path / pathtest need the light-pdf walk stage (area lights present)
  int                cls;    // path / pathdirect / pathmis: the scene class (0 general, 1 matte triangles, 2 no textures, 3 opaque textured)
};

int launch_path(const Launch& l);
int launch_nee(const Launch& l);
int launch_nee_class(const Launch& l);  // yt_trace_nee_cls.hip: pathdirect / pathmis of scene classes 1-3 (1 = not served: launch_nee)
int launch_misc(const Launch& l);

// k_trace<S, LP, ...> by the walk the launch asks for
template <int S, int LP>
void launch_trace(const Launch& l) {
  using namespace yt;
  dim3 grid(l.blocks), block(YT_BLOCK);
  if (l.count)  // the counting launch walks binary: its counts are the reference's
    hipLaunchKernelGGL((k_trace<S, LP, true, false>), grid, block, 0, l.stream, *l.ds, *l.st, *l.kp);
  else if (l.wide)
    hipLaunchKernelGGL((k_trace<S, LP, false, true>), grid, block, 0, l.stream, *l.ds, *l.st, *l.kp);
  else
    hipLaunchKernelGGL((k_trace<S, LP, false, false>), grid, block, 0, l.stream, *l.ds, *l.st, *l.kp);
}

}  // namespace ytl
