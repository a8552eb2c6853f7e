// yt_fastmath.h — the TOLERANCE mode's transcendental functions (ythip_params::fastmath): the same
// names as yt_libm.h, evaluated by gfx950's transcendental unit (v_sin / v_cos / v_exp / v_log / v_rcp / v_rsq /
// v_sqrt_f32: one quarter-rate instruction each, about 1 ulp) instead of glibc's double-precision kernels.  Only
// the fast translation unit (yt_fast.hip, -DYT_FAST) sees this header; the bit-exact kernels keep yt_libm.h.
// Results are NOT the reference's bits — the tests of this mode are statistical (tests/test_gpu_fastmath.py).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define YT_LIBM_FN __device__ __forceinline__

namespace ytm {

YT_LIBM_FN float sinf(float x) { return __builtin_amdgcn_sinf(x * 0.15915494309189535f); }  // v_sin_f32 takes revolutions
YT_LIBM_FN float cosf(float x) { return __builtin_amdgcn_cosf(x * 0.15915494309189535f); }
YT_LIBM_FN void  sincosf(float x, float* s, float* c) {
  const float r = x * 0.15915494309189535f;
  *s = __builtin_amdgcn_sinf(r), *c = __builtin_amdgcn_cosf(r);
}
YT_LIBM_FN float exp2f(float x) { return __builtin_amdgcn_exp2f(x); }
YT_LIBM_FN float expf(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
YT_LIBM_FN float logf(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }  // v_log_f32 is log2
// pow for the bases the path raises (colours, cosines, Fresnel terms: x >= 0): 0^y = 0 for y > 0 (log2 0 = -inf)
YT_LIBM_FN float powf(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }
// the inverse trigonometric functions have no hardware form: the device library's float polynomials
YT_LIBM_FN float atanf(float x) { return ::atanf(x); }
YT_LIBM_FN float atan2f(float y, float x) { return ::atan2f(y, x); }
YT_LIBM_FN float acosf(float x) { return ::acosf(x); }

}  // namespace ytm
