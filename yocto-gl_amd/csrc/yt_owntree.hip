// yt_owntree.hip — the OWN-TREE mode of trace_samples (ythip_params::fastmath = 2).
//
// The third build of the same source (yt_kernels.h): the tolerance mode's arithmetic (-DYT_FAST: hardware transcendentals,
// reciprocals, fused multiply-adds in shading) AND a traversal that no longer follows the reference's trees — yt_own.h: the
// device builder's SAH tree, collapsed two levels per node and compressed to 64-B nodes of 8-bit boxes, half the load
// instructions per step of the quad walk.  Radiance within the stated tolerance of the reference (tests/test_gpu_own_tree.py
// gates every BASELINE workload against oracle/_ref); hit records equal to the reference's except at ties and box-edge grazes
// (the agreement on the 200 k-ray batches is printed and asserted there).  ythip_intersect_batch is NOT in this unit: hit
// indices for a ray batch stay bit-exact.  ythip_own_intersect below is the test entry behind ythip_intersect_batch_own.
//
// Everything of ours is compiled into namespace yt_own / ytm_own here.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/ythip.h"

#define YT_FAST 1
#define YT_STREAM_KERNELS 1  // this unit's build of the streaming scheduler (yt_stream_unit.h)
#define ytl ytl_own
#define YT_OWN_TREE 1
#define yt yt_own
#define ytm ytm_own
#include "yt_kernels.h"
#include "yt_own.h"
#include "yt_stream_unit.h"

using namespace yt_own;

namespace {
template <int S, int LP, int CLS = 0>
void launch(hipStream_t stream, int blocks, const DScene& ds, const DState& st, const KParams& kp) {
  hipLaunchKernelGGL((k_trace<S, LP, false, true, CLS>), dim3(blocks), dim3(YT_BLOCK), 0, stream, ds, st, kp);
}
}  // namespace

// `ds`: the caller's DScene with the own tree's bvh fields swapped in (yt_ctx.h: BvhView).  Returns 0 = launched,
// 1 = no own-tree kernel for this sampler (diagram / falsecolor: debug views).
extern "C" int ythip_own_launch(void* stream, int blocks, const void* ds_, const void* st_, const void* kp_, int lp, int cls) {
  const DScene&  ds = *static_cast<const DScene*>(ds_);
  const DState&  st = *static_cast<const DState*>(st_);
  const KParams& kp = *static_cast<const KParams*>(kp_);
  hipStream_t    s  = static_cast<hipStream_t>(stream);
  const bool     defer = lp == LP_DEFER;
  switch (kp.sampler) {
    case YTHIP_SAMPLER_PATH:
      if (cls == 1) defer ? launch<YTHIP_SAMPLER_PATH, LP_DEFER, 1>(s, blocks, ds, st, kp) : launch<YTHIP_SAMPLER_PATH, LP_NONE, 1>(s, blocks, ds, st, kp);
      else if (cls == 2) defer ? launch<YTHIP_SAMPLER_PATH, LP_DEFER, 2>(s, blocks, ds, st, kp) : launch<YTHIP_SAMPLER_PATH, LP_NONE, 2>(s, blocks, ds, st, kp);
      else if (cls == 3) defer ? launch<YTHIP_SAMPLER_PATH, LP_DEFER, 3>(s, blocks, ds, st, kp) : launch<YTHIP_SAMPLER_PATH, LP_NONE, 3>(s, blocks, ds, st, kp);
      else defer ? launch<YTHIP_SAMPLER_PATH, LP_DEFER>(s, blocks, ds, st, kp) : launch<YTHIP_SAMPLER_PATH, LP_NONE>(s, blocks, ds, st, kp);
      return 0;
#ifndef YT_DEV_ONLY_PATH
    case YTHIP_SAMPLER_PATHTEST:
      defer ? launch<YTHIP_SAMPLER_PATHTEST, LP_DEFER>(s, blocks, ds, st, kp) : launch<YTHIP_SAMPLER_PATHTEST, LP_NONE>(s, blocks, ds, st, kp);
      return 0;
    case YTHIP_SAMPLER_PATHDIRECT:  // (by scene class since round 6, as `path`)
      if (cls == 1) launch<YTHIP_SAMPLER_PATHDIRECT, LP_DEFER, 1>(s, blocks, ds, st, kp);
      else if (cls == 2) launch<YTHIP_SAMPLER_PATHDIRECT, LP_DEFER, 2>(s, blocks, ds, st, kp);
      else if (cls == 3) launch<YTHIP_SAMPLER_PATHDIRECT, LP_DEFER, 3>(s, blocks, ds, st, kp);
      else launch<YTHIP_SAMPLER_PATHDIRECT, LP_DEFER>(s, blocks, ds, st, kp);
      return 0;
    case YTHIP_SAMPLER_PATHMIS:
      if (cls == 1) launch<YTHIP_SAMPLER_PATHMIS, LP_DEFER, 1>(s, blocks, ds, st, kp);
      else if (cls == 2) launch<YTHIP_SAMPLER_PATHMIS, LP_DEFER, 2>(s, blocks, ds, st, kp);
      else if (cls == 3) launch<YTHIP_SAMPLER_PATHMIS, LP_DEFER, 3>(s, blocks, ds, st, kp);
      else launch<YTHIP_SAMPLER_PATHMIS, LP_DEFER>(s, blocks, ds, st, kp);
      return 0;
    case YTHIP_SAMPLER_NAIVE: launch<YTHIP_SAMPLER_NAIVE, LP_NONE>(s, blocks, ds, st, kp); return 0;
    case YTHIP_SAMPLER_EYELIGHT: launch<YTHIP_SAMPLER_EYELIGHT, LP_NONE>(s, blocks, ds, st, kp); return 0;
    case YTHIP_SAMPLER_FURNACE: launch<YTHIP_SAMPLER_FURNACE, LP_NONE>(s, blocks, ds, st, kp); return 0;
#endif
    default: return 1;
  }
}

// the own walk on a ray batch (device pointers; instances: null = intersect_scene, else intersect_instance per ray)
extern "C" int ythip_own_intersect(void* stream, const void* ds_, const void* rays, const int* instances, long long n, void* hits) {
  const DScene& ds = *static_cast<const DScene*>(ds_);
  if (n <= 0) return 0;
  hipLaunchKernelGGL((k_intersect_batch<false, true>), dim3((unsigned)((n + YT_BLOCK - 1) / YT_BLOCK)), dim3(YT_BLOCK), 0,
      static_cast<hipStream_t>(stream), ds, static_cast<const ythip_ray*>(rays), instances, n, 0, static_cast<ythip_hit*>(hits),
      (unsigned long long*)nullptr);
  return 0;
}

// the streaming scheduler in this mode (ythip_set_scheduler 1 with fastmath = 2): `l` points at the caller's ytl::StreamLaunch — the
// same struct under this unit's namespaces, its DScene the one with the own tree's bvh fields swapped in
extern "C" void ythip_own_stream_begin(const void* l) { ytl::stream_begin(*static_cast<const ytl::StreamLaunch*>(l)); }
extern "C" void ythip_own_stream_generation(const void* l) { ytl::stream_generation(*static_cast<const ytl::StreamLaunch*>(l)); }
extern "C" void ythip_own_stream_finish(const void* l) { ytl::stream_finish(*static_cast<const ytl::StreamLaunch*>(l)); }
