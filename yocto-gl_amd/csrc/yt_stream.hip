// yt_stream.hip — the kernels of the streaming scheduler (yt_stream.h) for `path`, by scene class, and the launch of
// one generation.  The host loop that enqueues generations and watches the queue length is enqueue_stream in ythip.hip.
#define YT_STREAM_KERNELS 1
#include "yt_stream_launch.h"

using namespace yt;

namespace ytl {

namespace {
template <int LP, int CLS>
void shade(const StreamLaunch& l) {
  hipLaunchKernelGGL((ks_shade<YTHIP_SAMPLER_PATH, LP, CLS, true>), dim3(l.ss->nslots / YT_BLOCK), dim3(YT_BLOCK), 0, l.stream, *l.ds, *l.st, *l.kp,
      *l.ss);
}
template <int TRI>
void extend(const StreamLaunch& l) {
  if (l.phased)
    hipLaunchKernelGGL((ks_extend<true, TRI, true>), dim3(l.ss->nslots / YT_BLOCK), dim3(YT_BLOCK), 0, l.stream, *l.ds, *l.ss);
  else
    hipLaunchKernelGGL((ks_extend<true, TRI, false>), dim3(l.ss->nslots / YT_BLOCK), dim3(YT_BLOCK), 0, l.stream, *l.ds, *l.ss);
}
}  // namespace

bool stream_supported(const StreamLaunch& l) { return l.kp->sampler == YTHIP_SAMPLER_PATH && l.kp->bounces > 0; }

void stream_begin(const StreamLaunch& l) {
  hipLaunchKernelGGL(ks_init, dim3(l.ss->nslots / YT_BLOCK), dim3(YT_BLOCK), 0, l.stream, *l.ds, *l.st, *l.kp, *l.ss);
  hipLaunchKernelGGL(ks_scan, dim3(1), dim3(YT_SCAN_THREADS), 0, l.stream, *l.ss);
}

// one generation of one group, ending with the scan of the keys it emitted (so counts[0] read after the launch is the
// length of the NEXT generation's queue: zero = the group is done)
void stream_generation(const StreamLaunch& l) {
  const bool defer = l.lp == LP_DEFER;
  {
    hipLaunchKernelGGL(ks_scatter, dim3((l.ss->nslots + 255) / 256), dim3(256), 0, l.stream, *l.ss);
    switch (l.cls) {
      case 1: extend<1>(l); break;
      case 3: extend<2>(l); break;
      default: extend<0>(l); break;
    }
    switch (l.cls) {
      case 1: defer ? shade<LP_DEFER, 1>(l) : shade<LP_NONE, 1>(l); break;
      case 2: defer ? shade<LP_DEFER, 2>(l) : shade<LP_NONE, 2>(l); break;
      case 3: defer ? shade<LP_DEFER, 3>(l) : shade<LP_NONE, 3>(l); break;
      default: defer ? shade<LP_DEFER, 0>(l) : shade<LP_NONE, 0>(l); break;
    }
    hipLaunchKernelGGL(ks_scan, dim3(1), dim3(YT_SCAN_THREADS), 0, l.stream, *l.ss);
  }
}

}  // namespace ytl
