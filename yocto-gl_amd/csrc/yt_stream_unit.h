// yt_stream_unit.h — the kernels of the streaming scheduler (yt_stream.h) for `path`, by scene class, and the launch of one
// generation, instantiated ONCE PER BUILD of the source: yt_stream.hip (bit-exact, namespace yt / ytl), yt_fast.hip (the tolerance
// mode, yt_fast / ytl_fast) and yt_owntree.hip (the own tree, yt_own / ytl_own — ks_extend's traverse_any is yt_own.h's walk there).
// The including unit defines YT_STREAM_KERNELS before anything else (the plain kernels of yt_stream.h are compiled per unit too:
// ks_init generates camera rays with the unit's arithmetic).  The host loop that enqueues generations and watches the queue
// length is enqueue_stream in ythip.hip; it reaches the non-default units through their extern "C" entries.
#pragma once

#include "yt_stream_launch.h"

namespace ytl {

namespace {
template <int LP, int CLS>
void stream_shade(const StreamLaunch& l) {
  using namespace yt;
  if constexpr (LP == LP_DEFER) {
    if (l.kp->sampler == YTHIP_SAMPLER_PATHDIRECT) {
      hipLaunchKernelGGL((ks_shade<YTHIP_SAMPLER_PATHDIRECT, LP, CLS, true>), dim3(l.ss->nslots / YT_BLOCK), dim3(YT_BLOCK), 0, l.stream, *l.ds, *l.st,
          *l.kp, *l.ss);
      return;
    }
  }
  if constexpr (CLS == 0) {  // `naive` and `pathtest`: the general class, as their fused kernels (yt_trace_misc.hip, yt_trace_path.hip)
    const dim3 grid(l.ss->nslots / YT_BLOCK), block(YT_BLOCK);
    if (l.kp->sampler == YTHIP_SAMPLER_PATHTEST) {
      hipLaunchKernelGGL((ks_shade<YTHIP_SAMPLER_PATHTEST, LP, 0, true>), grid, block, 0, l.stream, *l.ds, *l.st, *l.kp, *l.ss);
      return;
    }
    if constexpr (LP == LP_NONE) {
      if (l.kp->sampler == YTHIP_SAMPLER_NAIVE) {
        hipLaunchKernelGGL((ks_shade<YTHIP_SAMPLER_NAIVE, LP_NONE, 0, true>), grid, block, 0, l.stream, *l.ds, *l.st, *l.kp, *l.ss);
        return;
      }
    }
  }
  hipLaunchKernelGGL((ks_shade<YTHIP_SAMPLER_PATH, LP, CLS, true>), dim3(l.ss->nslots / YT_BLOCK), dim3(YT_BLOCK), 0, l.stream, *l.ds, *l.st, *l.kp,
      *l.ss);
}
template <int TRI>
void stream_extend(const StreamLaunch& l) {
  using namespace yt;
  if (l.phased)
    hipLaunchKernelGGL((ks_extend<true, TRI, true>), dim3(l.ss->nslots / YT_BLOCK), dim3(YT_BLOCK), 0, l.stream, *l.ds, *l.ss);
  else
    hipLaunchKernelGGL((ks_extend<true, TRI, false>), dim3(l.ss->nslots / YT_BLOCK), dim3(YT_BLOCK), 0, l.stream, *l.ds, *l.ss);
}
template <int LP, int CLS, int TRI>
void stream_finish_launch(const StreamLaunch& l) {
  using namespace yt;
  if constexpr (LP == LP_DEFER) {
    if (l.kp->sampler == YTHIP_SAMPLER_PATHDIRECT) {  // (never the majority-phase walk)
      hipLaunchKernelGGL((ks_finish<YTHIP_SAMPLER_PATHDIRECT, LP, CLS, true, TRI, false>), dim3(l.ss->nslots / YT_BLOCK), dim3(YT_BLOCK), 0, l.stream,
          *l.ds, *l.st, *l.kp, *l.ss);
      return;
    }
  }
  if constexpr (CLS == 0) {
    const dim3 grid(l.ss->nslots / YT_BLOCK), block(YT_BLOCK);
    if (l.kp->sampler == YTHIP_SAMPLER_PATHTEST) {
      hipLaunchKernelGGL((ks_finish<YTHIP_SAMPLER_PATHTEST, LP, 0, true, 0, false>), grid, block, 0, l.stream, *l.ds, *l.st, *l.kp, *l.ss);
      return;
    }
    if constexpr (LP == LP_NONE) {
      if (l.kp->sampler == YTHIP_SAMPLER_NAIVE) {
        hipLaunchKernelGGL((ks_finish<YTHIP_SAMPLER_NAIVE, LP_NONE, 0, true, 0, false>), grid, block, 0, l.stream, *l.ds, *l.st, *l.kp, *l.ss);
        return;
      }
    }
  }
  if (l.phased)
    hipLaunchKernelGGL((ks_finish<YTHIP_SAMPLER_PATH, LP, CLS, true, TRI, true>), dim3(l.ss->nslots / YT_BLOCK), dim3(YT_BLOCK), 0, l.stream, *l.ds, *l.st,
        *l.kp, *l.ss);
  else
    hipLaunchKernelGGL((ks_finish<YTHIP_SAMPLER_PATH, LP, CLS, true, TRI, false>), dim3(l.ss->nslots / YT_BLOCK), dim3(YT_BLOCK), 0, l.stream, *l.ds, *l.st,
        *l.kp, *l.ss);
}
}  // namespace

bool stream_supported(const StreamLaunch& l) {
  return (l.kp->sampler == YTHIP_SAMPLER_PATH || l.kp->sampler == YTHIP_SAMPLER_PATHDIRECT || l.kp->sampler == YTHIP_SAMPLER_PATHTEST ||
          l.kp->sampler == YTHIP_SAMPLER_NAIVE) &&
         l.kp->bounces > 0;
}

void stream_begin(const StreamLaunch& l) {
  using namespace yt;
  hipLaunchKernelGGL(ks_init, dim3(l.ss->nslots / YT_BLOCK), dim3(YT_BLOCK), 0, l.stream, *l.ds, *l.st, *l.kp, *l.ss);
  hipLaunchKernelGGL(ks_scan, dim3(1), dim3(YT_SCAN_THREADS), 0, l.stream, *l.ss);
}

// one generation of one group, ending with the scan of the keys it emitted (so counts[0] read after the launch is the
// length of the NEXT generation's queue: zero = the group is done)
void stream_generation(const StreamLaunch& l) {
  using namespace yt;
  const bool defer = l.lp == LP_DEFER;
  hipLaunchKernelGGL(ks_scatter, dim3((l.ss->nslots + 255) / 256), dim3(256), 0, l.stream, *l.ss);
  switch (l.cls) {
    case 1: stream_extend<1>(l); break;
    case 3: stream_extend<2>(l); break;
    default: stream_extend<0>(l); break;
  }
  switch (l.cls) {
    case 1: defer ? stream_shade<LP_DEFER, 1>(l) : stream_shade<LP_NONE, 1>(l); break;
    case 2: defer ? stream_shade<LP_DEFER, 2>(l) : stream_shade<LP_NONE, 2>(l); break;
    case 3: defer ? stream_shade<LP_DEFER, 3>(l) : stream_shade<LP_NONE, 3>(l); break;
    default: defer ? stream_shade<LP_DEFER, 0>(l) : stream_shade<LP_NONE, 0>(l); break;
  }
  hipLaunchKernelGGL(ks_scan, dim3(1), dim3(YT_SCAN_THREADS), 0, l.stream, *l.ss);
}

// the tail of the group's batch in one launch: the queue the last scan sized, sorted once more, then every entry's path slot carried
// to the end of its pixel's batch by one lane (ks_finish).  `blocks` covers the queue length the host read back.
void stream_finish(const StreamLaunch& l) {
  using namespace yt;
  const bool defer = l.lp == LP_DEFER;
  hipLaunchKernelGGL(ks_scatter, dim3((l.ss->nslots + 255) / 256), dim3(256), 0, l.stream, *l.ss);
  switch (l.cls) {
    case 1: defer ? stream_finish_launch<LP_DEFER, 1, 1>(l) : stream_finish_launch<LP_NONE, 1, 1>(l); break;
    case 2: defer ? stream_finish_launch<LP_DEFER, 2, 0>(l) : stream_finish_launch<LP_NONE, 2, 0>(l); break;
    case 3: defer ? stream_finish_launch<LP_DEFER, 3, 2>(l) : stream_finish_launch<LP_NONE, 3, 2>(l); break;
    default: defer ? stream_finish_launch<LP_DEFER, 0, 0>(l) : stream_finish_launch<LP_NONE, 0, 0>(l); break;
  }
}

}  // namespace ytl
