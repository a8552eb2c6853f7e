// yt_fast.hip — the TOLERANCE mode of trace_samples (ythip_params::fastmath = 1).
//
// north_star asks for radiance "within a stated float tolerance" and bit-exact hit indices; the default build
// delivers far more — the reference's whole trace_state, byte for byte — and pays for it in the kernel that is
// bound by VALU issue: IEEE divisions (normalize = 3 of them), correctly rounded square roots, glibc's
// double-precision sinf / cosf / expf / logf / powf, no fused multiply-adds.  This translation unit compiles the
// SAME source (yt_kernels.h: same integrators, same rng streams and draw order, same traversal) with -DYT_FAST:
//
//   * shading / sampling / camera arithmetic: v_rcp / v_rsq / v_sqrt / v_sin / v_cos / v_exp / v_log_f32
//     (yt_math.h: div_, rcp_, sqrt_, normalize, vector operator/; yt_fastmath.h), multiply-adds of the shading
//     files fused (#pragma clang fp contract(fast) in yt_shading.h / yt_scene.h / yt_kernels.h / the samplers
//     of yt_math.h);
//   * the TRAVERSAL is untouched: yt_bvh.h and the root-box peek spell their divisions `/` and their square root
//     sqrt_ieee_, the vector helpers they share with shading are outside every contract pragma, and this unit is
//     compiled with the same -ffp-contract=off as the rest — for a given ray the hit record is the reference's.
//     (ythip_intersect_batch is not in this unit at all: it always runs the bit-exact kernels.)
//
// What it is checked against: tests/test_gpu_fastmath.py — BASELINE.md §3.5's statistical gates against the
// bit-exact render at equal spp (image mean within 0.5 %, 8x8-block MAE within the reference's own seed-to-seed
// spread), rng streams still the reference's wherever a path took the same decisions.
//
// Everything of ours is compiled into namespace yt_fast / ytm_fast here (the two defines below): the kernels of
// this unit and the bit-exact ones of ythip.hip are different functions with the same source names.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/ythip.h"

#define YT_FAST 1
#define YT_STREAM_KERNELS 1  // this unit's build of the streaming scheduler (yt_stream_unit.h)
#define ytl ytl_fast
#define yt yt_fast
#define ytm ytm_fast
#include "yt_kernels.h"
#include "yt_stream_unit.h"

using namespace yt_fast;

namespace {
template <int S, int LP, int CLS = 0>
void launch(hipStream_t stream, int blocks, const DScene& ds, const DState& st, const KParams& kp) {
  hipLaunchKernelGGL((k_trace<S, LP, false, true, CLS>), dim3(blocks), dim3(YT_BLOCK), 0, stream, ds, st, kp);
}
}  // namespace

// The dispatch of the tolerance mode: the wide-walk kernels only (a scene the wide walk cannot serve — trees of fewer
// than 64 primitives, trees too deep for its stack — renders with the bit-exact kernels: returns 1, nothing launched).
// `ds` / `st` / `kp` point at the caller's yt::DScene / yt::DState / yt::KParams: the same structs under this unit's
// namespace.  cls: the scene class of the path sampler (0 general, 1 all matte triangles, 2 no textures, 3 opaque textured).
extern "C" int ythip_fast_launch(void* stream, int blocks, const void* ds_, const void* st_, const void* kp_, int lp, int cls) {
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
    default: return 1;  // diagram / falsecolor: debug views, no tolerance build
  }
}

// the streaming scheduler in this mode (ythip_set_scheduler 1 with fastmath = 1): `l` points at the caller's ytl::StreamLaunch — the
// same struct under this unit's namespaces
extern "C" void ythip_fast_stream_begin(const void* l) { ytl::stream_begin(*static_cast<const ytl::StreamLaunch*>(l)); }
extern "C" void ythip_fast_stream_generation(const void* l) { ytl::stream_generation(*static_cast<const ytl::StreamLaunch*>(l)); }
extern "C" void ythip_fast_stream_finish(const void* l) { ytl::stream_finish(*static_cast<const ytl::StreamLaunch*>(l)); }
