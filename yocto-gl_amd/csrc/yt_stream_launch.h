// yt_stream_launch.h — what ythip.hip (the host loop) and yt_stream.hip (the kernels) share of the streaming scheduler.
#pragma once

#include "yt_stream.h"

namespace ytl {

struct StreamLaunch {
  hipStream_t        stream;
  const yt::DScene*  ds;
  const yt::DState*  st;
  const yt::KParams* kp;
  const yt::DStream* ss;
  int                lp;      // LP_NONE / LP_DEFER (area lights present)
  int                cls;     // the scene class of `path` (yt_kernels.h: step_path's CLS)
  bool               phased;  // the majority-phase scene walk in ks_extend
};

bool stream_supported(const StreamLaunch& l);
void stream_begin(const StreamLaunch& l);       // ks_init + the first scan of the group l.ss, on l.stream
void stream_generation(const StreamLaunch& l);  // scatter, extend, shade, scan
void stream_finish(const StreamLaunch& l);      // scatter, then ks_finish: every queued path slot to the end of its pixel's batch

}  // namespace ytl
