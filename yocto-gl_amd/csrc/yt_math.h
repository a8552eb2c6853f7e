// yt_math.h — device-side scalar/vector math with the reference's exact
// operation order.  Every function is written so that, compiled with
// -ffp-contract=off (no FMA contraction) and IEEE div/sqrt, it produces the
// same bits as the g++-built reference on x86-64 (SSE, FLT_EVAL_METHOD 0).
//
// Restates the subset of libs/yocto/yocto_math.h used on the hot path:
//   min/max/clamp   yocto_math.h:1046-1051   (ternary, NaN-order-sensitive)
//   vec3f ops       yocto_math.h:1253-1370
//   mat3f/frame3f   yocto_math.h:1933-1986, 2108-2122, 2236-2280
// Do not "simplify" expressions here: association order is part of the spec.
#pragma once

#include <stdint.h>
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define YT_FN __device__ __forceinline__
#else  // the same arithmetic compiled for the host: tests/cpp/shading_check.cpp (g++, against the reference's headers)
#include <math.h>
#include <string.h>
#define YT_FN inline
static inline float __uint_as_float(unsigned u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}
#endif

#ifdef YT_FAST  // the tolerance mode's translation unit (yt_fast.hip): hardware transcendentals, reciprocals instead of divisions
#include "yt_fastmath.h"
#else
#include "yt_libm.h"  // the reference platform's libm (glibc 2.35), restated: ytm::sinf ... ytm::powf
#endif

namespace yt {

constexpr float pif     = 3.14159265358979323846f;  // (float)pi
constexpr float flt_max = 3.402823466e+38f;
constexpr float ray_eps = 1e-4f;  // yocto_geometry.h:125

struct vec2f {
  float x, y;
};
struct vec3f {
  float x, y, z;
};
struct vec4f {
  float x, y, z, w;
};
struct mat3f {
  vec3f x, y, z;
};
struct frame3f {
  vec3f x, y, z, o;
};

// scalar -------------------------------------------------------------------
YT_FN float fabs_(float a) { return a < 0 ? -a : a; }
// pow(x, 2.0f) as the g++ -O3 reference evaluates it: one multiplication (GCC expands integer
// exponents in [-1, 2] without -ffast-math)
YT_FN float sqr_(float a) { return a * a; }                // :1044
YT_FN float min_(float a, float b) { return (a < b) ? a : b; }       // :1046
YT_FN float max_(float a, float b) { return (a > b) ? a : b; }       // :1047
YT_FN float clamp_(float a, float lo, float hi) { return min_(max_(a, lo), hi); }
YT_FN int   min_(int a, int b) { return (a < b) ? a : b; }
YT_FN int   max_(int a, int b) { return (a > b) ? a : b; }
YT_FN int   clamp_(int a, int lo, int hi) { return min_(max_(a, lo), hi); }
YT_FN bool  isfinite_(float a) { return __builtin_isfinite(a); }
YT_FN float sqrt_ieee_(float a) { return __builtin_sqrtf(a); }  // IEEE (v_sqrt + fixup): the traversal's, in every mode
// Shading / sampling arithmetic goes through div_ / sqrt_ (and the vector operator/, normalize): the reference's IEEE
// operations in the bit-exact build; in the tolerance mode (-DYT_FAST) v_rcp / v_sqrt / v_rsq_f32, about 1 ulp each.
// The TRAVERSAL (yt_bvh.h, misses_scene_root) spells its divisions `/` and its root sqrt_ieee_: identical in both.
#ifdef YT_FAST
YT_FN float rcp_(float a) { return __builtin_amdgcn_rcpf(a); }
YT_FN float div_(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
YT_FN float sqrt_(float a) { return __builtin_amdgcn_sqrtf(a); }
YT_FN float rsqrt_(float a) { return __builtin_amdgcn_rsqf(a); }
#else
YT_FN float rcp_(float a) { return 1 / a; }
YT_FN float div_(float a, float b) { return a / b; }
YT_FN float sqrt_(float a) { return __builtin_sqrtf(a); }
#endif
YT_FN float lerp_(float a, float b, float u) { return a * (1 - u) + b * u; }

// vec2 -----------------------------------------------------------------------
YT_FN vec2f operator+(vec2f a, vec2f b) { return {a.x + b.x, a.y + b.y}; }
YT_FN vec2f operator*(vec2f a, float b) { return {a.x * b, a.y * b}; }
YT_FN vec2f operator*(vec2f a, vec2f b) { return {a.x * b.x, a.y * b.y}; }
YT_FN vec2f operator-(float a, vec2f b) { return {a - b.x, a - b.y}; }

// vec3 -----------------------------------------------------------------------
YT_FN bool operator==(vec3f a, vec3f b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
YT_FN bool operator!=(vec3f a, vec3f b) { return a.x != b.x || a.y != b.y || a.z != b.z; }
YT_FN vec3f operator-(vec3f a) { return {-a.x, -a.y, -a.z}; }
YT_FN vec3f operator+(vec3f a, vec3f b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
YT_FN vec3f operator+(vec3f a, float b) { return {a.x + b, a.y + b, a.z + b}; }
YT_FN vec3f operator+(float a, vec3f b) { return {a + b.x, a + b.y, a + b.z}; }
YT_FN vec3f operator-(vec3f a, vec3f b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
YT_FN vec3f operator-(vec3f a, float b) { return {a.x - b, a.y - b, a.z - b}; }
YT_FN vec3f operator-(float a, vec3f b) { return {a - b.x, a - b.y, a - b.z}; }
YT_FN vec3f operator*(vec3f a, vec3f b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
YT_FN vec3f operator*(vec3f a, float b) { return {a.x * b, a.y * b, a.z * b}; }
YT_FN vec3f operator*(float a, vec3f b) { return {a * b.x, a * b.y, a * b.z}; }
// (vector division is shading arithmetic: the traversal only ever divides scalars)
#ifdef YT_FAST
YT_FN vec3f operator/(vec3f a, vec3f b) { return {a.x * rcp_(b.x), a.y * rcp_(b.y), a.z * rcp_(b.z)}; }
YT_FN vec3f operator/(vec3f a, float b) {
  const float r = rcp_(b);
  return {a.x * r, a.y * r, a.z * r};
}
YT_FN vec3f operator/(float a, vec3f b) { return {a * rcp_(b.x), a * rcp_(b.y), a * rcp_(b.z)}; }
#else
YT_FN vec3f operator/(vec3f a, vec3f b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
YT_FN vec3f operator/(vec3f a, float b) { return {a.x / b, a.y / b, a.z / b}; }
YT_FN vec3f operator/(float a, vec3f b) { return {a / b.x, a / b.y, a / b.z}; }
#endif
YT_FN vec3f& operator+=(vec3f& a, vec3f b) { return a = a + b; }
YT_FN vec3f& operator*=(vec3f& a, vec3f b) { return a = a * b; }
YT_FN vec3f& operator*=(vec3f& a, float b) { return a = a * b; }

YT_FN float dot(vec3f a, vec3f b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // :1305
YT_FN vec3f cross(vec3f a, vec3f b) {                                              // :1308
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
YT_FN float length(vec3f a) { return sqrt_(dot(a, a)); }
YT_FN vec3f normalize(vec3f a) {  // :1314
#ifdef YT_FAST
  auto l2 = dot(a, a);
  return (l2 != 0) ? a * rsqrt_(l2) : a;
#else
  auto l = length(a);
  return (l != 0) ? a / l : a;
#endif
}
YT_FN float distance_squared(vec3f a, vec3f b) { return dot(a - b, a - b); }
YT_FN vec3f orthonormalize(vec3f a, vec3f b) { return normalize(a - b * dot(a, b)); }  // :1332
YT_FN vec3f reflect(vec3f w, vec3f n) { return -w + 2 * dot(n, w) * n; }              // :1336
YT_FN vec3f refract(vec3f w, vec3f n, float inv_eta) {                                 // :1339
  auto cosine = dot(n, w);
  auto k      = 1 + inv_eta * inv_eta * (cosine * cosine - 1);
  if (k < 0) return {0, 0, 0};  // tir
  return -w * inv_eta + (inv_eta * cosine - sqrt_(k)) * n;
}
YT_FN vec3f max_(vec3f a, float b) { return {max_(a.x, b), max_(a.y, b), max_(a.z, b)}; }
YT_FN vec3f min3_(vec3f a, vec3f b) { return {min_(a.x, b.x), min_(a.y, b.y), min_(a.z, b.z)}; }
YT_FN vec3f max3_(vec3f a, vec3f b) { return {max_(a.x, b.x), max_(a.y, b.y), max_(a.z, b.z)}; }
YT_FN vec3f clamp_(vec3f a, float lo, float hi) {
  return {clamp_(a.x, lo, hi), clamp_(a.y, lo, hi), clamp_(a.z, lo, hi)};
}
YT_FN vec3f lerp_(vec3f a, vec3f b, float u) { return a * (1 - u) + b * u; }  // :1362
YT_FN float max_(vec3f a) { return max_(max_(a.x, a.y), a.z); }               // :1369
YT_FN float min_(vec3f a) { return min_(min_(a.x, a.y), a.z); }
YT_FN float sum(vec3f a) { return a.x + a.y + a.z; }
YT_FN float mean(vec3f a) { return div_(sum(a), 3); }
YT_FN vec3f abs_(vec3f a) { return {fabs_(a.x), fabs_(a.y), fabs_(a.z)}; }
YT_FN vec3f sqrt_(vec3f a) { return {sqrt_(a.x), sqrt_(a.y), sqrt_(a.z)}; }
YT_FN vec3f exp_(vec3f a) { return {ytm::expf(a.x), ytm::expf(a.y), ytm::expf(a.z)}; }
YT_FN vec3f log_(vec3f a) { return {ytm::logf(a.x), ytm::logf(a.y), ytm::logf(a.z)}; }
YT_FN bool  isfinite_(vec3f a) { return isfinite_(a.x) && isfinite_(a.y) && isfinite_(a.z); }
YT_FN float at(vec3f a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// vec4 -----------------------------------------------------------------------
YT_FN vec4f operator+(vec4f a, vec4f b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
YT_FN vec4f operator*(vec4f a, float b) { return {a.x * b, a.y * b, a.z * b, a.w * b}; }
YT_FN vec4f lerp_(vec4f a, vec4f b, float u) { return a * (1 - u) + b * u; }
YT_FN vec3f xyz(vec4f a) { return {a.x, a.y, a.z}; }

// mat3 / frame -----------------------------------------------------------------
YT_FN vec3f operator*(const mat3f& a, vec3f b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // :1942
YT_FN mat3f transpose(const mat3f& a) {
  return {{a.x.x, a.y.x, a.z.x}, {a.x.y, a.y.y, a.z.y}, {a.x.z, a.y.z, a.z.z}};
}
// basis_fromz — yocto_math.h:1977-1986 (Pixar branchless ONB)
YT_FN mat3f basis_fromz(vec3f v) {
  auto z    = normalize(v);
  auto sign = copysignf(1.0f, z.z);
  auto a    = div_(-1.0f, sign + z.z);
  auto b    = z.x * z.y * a;
  auto x    = vec3f{1.0f + sign * z.x * z.x * a, sign * b, -sign * z.x};
  auto y    = vec3f{b, sign + z.y * z.y * a, -z.y};
  return {x, y, z};
}
YT_FN vec3f transform_point(const frame3f& a, vec3f b) {  // :2263
  return a.x * b.x + a.y * b.y + a.z * b.z + a.o;
}
YT_FN vec3f transform_vector(const frame3f& a, vec3f b) {  // :2266
  return a.x * b.x + a.y * b.y + a.z * b.z;
}
YT_FN vec3f transform_direction(const frame3f& a, vec3f b) {  // :2269
  return normalize(transform_vector(a, b));
}
// transform_normal(frame, n, non_rigid = false) — :2272-2280
YT_FN vec3f transform_normal(const frame3f& a, vec3f b) {
  return normalize(transform_vector(a, b));
}
YT_FN vec3f transform_direction(const mat3f& a, vec3f b) { return normalize(a * b); }  // :2236
// mat3 inverse = adjoint * (1/det) — :1967-1974
YT_FN mat3f inverse(const mat3f& a) {
  auto det = dot(a.x, cross(a.y, a.z));
  auto adj = transpose(mat3f{cross(a.y, a.z), cross(a.z, a.x), cross(a.x, a.y)});
  auto s   = rcp_(det);
  return {adj.x * s, adj.y * s, adj.z * s};
}
// transform_normal(frame, n, non_rigid = true): normalize(transpose(inverse(rot)) * n)
YT_FN vec3f transform_normal_nonrigid(const frame3f& a, vec3f b) {
  auto m = transpose(inverse(mat3f{a.x, a.y, a.z}));
  return normalize(m * b);
}

// ---------------------------------------------------------------------------
// PCG32 — libs/yocto/yocto_sampling.h:187-232 (integer-exact)
// ---------------------------------------------------------------------------
struct rng_state {
  uint64_t state, inc;
};
YT_FN uint32_t advance_rng(rng_state& rng) {
  uint64_t oldstate   = rng.state;
  rng.state           = oldstate * 6364136223846793005ULL + rng.inc;
  uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
  uint32_t rot        = (uint32_t)(oldstate >> 59u);
  return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}
YT_FN rng_state make_rng(uint64_t seed, uint64_t seq) {
  rng_state rng;
  rng.state = 0U;
  rng.inc   = (seq << 1u) | 1u;
  advance_rng(rng);
  rng.state += seed;
  advance_rng(rng);
  return rng;
}
YT_FN float rand1f(rng_state& rng) {
  uint32_t u = (advance_rng(rng) >> 9) | 0x3f800000u;
  return __uint_as_float(u) - 1.0f;
}
YT_FN vec2f rand2f(rng_state& rng) {
  auto x = rand1f(rng);
  auto y = rand1f(rng);
  return {x, y};
}
YT_FN vec3f rand3f(rng_state& rng) {
  auto x = rand1f(rng);
  auto y = rand1f(rng);
  auto z = rand1f(rng);
  return {x, y, z};
}

#ifdef YT_FAST  // tolerance mode: the samplers below may fuse their multiply-adds (vector helpers above: never — the traversal uses them)
#pragma clang fp contract(fast)
#endif
// ---------------------------------------------------------------------------
// Monte Carlo sampling — libs/yocto/yocto_sampling.h:252-398
// ---------------------------------------------------------------------------
YT_FN vec3f sample_sphere(vec2f ruv) {  // :277
  auto z   = 2 * ruv.y - 1;
  auto r   = sqrt_(clamp_(1 - z * z, 0.0f, 1.0f));
  auto  phi = 2 * pif * ruv.x;
  float sp, cp;
  ytm::sincosf(phi, &sp, &cp);  // (glibc's sincosf == its sinf and cosf: one argument reduction)
  return {r * cp, r * sp, z};
}
YT_FN vec3f sample_hemisphere_cos(vec3f normal, vec2f ruv) {  // :297
  auto z               = sqrt_(ruv.y);
  auto r               = sqrt_(1 - z * z);
  auto phi             = 2 * pif * ruv.x;
  float sp, cp;
  ytm::sincosf(phi, &sp, &cp);
  auto local_direction = vec3f{r * cp, r * sp, z};
  return transform_direction(basis_fromz(normal), local_direction);
}
YT_FN float sample_hemisphere_cos_pdf(vec3f normal, vec3f direction) {  // :304
  auto cosw = dot(normal, direction);
  return (cosw <= 0) ? 0 : div_(cosw, pif);
}
YT_FN vec2f sample_disk(vec2f ruv) {  // :339
  auto r   = sqrt_(ruv.y);
  auto phi = 2 * pif * ruv.x;
  float sp, cp;
  ytm::sincosf(phi, &sp, &cp);
  return {cp * r, sp * r};
}
YT_FN vec2f sample_triangle(vec2f ruv) {  // :354
  return {1 - sqrt_(ruv.x), ruv.y * sqrt_(ruv.x)};
}
YT_FN int sample_uniform(int size, float r) {  // :371
  return clamp_((int)(r * size), 0, size - 1);
}
YT_FN float sample_uniform_pdf(int size) { return div_((float)1, (float)size); }
// sample_discrete — :388-393 (std::upper_bound = first element > r)
YT_FN int sample_discrete(const float* cdf, int n, float r) {
  auto last = cdf[n - 1];
  r         = clamp_(r * last, (float)0, last - (float)0.00001);
  int lo = 0, len = n;
  while (len > 0) {
    int half = len >> 1;
    int mid  = lo + half;
    if (!(r < cdf[mid])) {
      lo  = mid + 1;
      len = len - half - 1;
    } else {
      len = half;
    }
  }
  return clamp_(lo, 0, n - 1);
}
YT_FN float sample_discrete_pdf(const float* cdf, int idx) {  // :395
  if (idx == 0) return cdf[0];
  return cdf[idx] - cdf[idx - 1];
}

#ifdef YT_FAST
#pragma clang fp contract(off)
#endif

}  // namespace yt
