// yt_libm.h — the libm of the REFERENCE PLATFORM, restated for the device.
//
// The reference is a g++ / glibc program: the only arithmetic on the trace path that does
// not live under libs/yocto is libm's — sinf, cosf (and sincosf where g++
// merges a pair), acosf, atanf, atan2f, expf, exp2f, logf, powf.  The device's own libm
// (ocml) differs from glibc's in the last ulp here and there, and one ulp in a sampled
// direction or a Fresnel term can flip a comparison further down the path: in round 1
// 1-5 % of the pixels took another path than the reference's within a few samples.
// Restating glibc's algorithms makes every float of the path the reference's.
//
// Third-party dependency, pinned: GNU libc 2.35 (Ubuntu GLIBC 2.35-0ubuntu3.x, the libm.so.6
// of this image, which is also the GPU box's), x86-64, as selected at run time on a CPU with
// FMA + AVX2 (the `_fma` ifunc variants).  Published algorithms:
//   * sinf / cosf / sincosf, expf / exp2f, logf, powf — ARM Optimized Routines (Szabolcs Nagy,
//     2017-2018; glibc sysdeps/ieee754/flt-32/{s_sinf,s_cosf,sincosf.h,e_expf,e_exp2f,e_logf,
//     e_powf}.c with the tables of sincosf_data.c, e_exp2f_data.c, e_logf_data.c,
//     e_powf_log2_data.c): double-precision kernels, one rounding to float at the end.  The
//     `_fma` variants are the same C compiled with -mfma -mavx2, where GCC contracts
//     `a * b + c` into fused multiply-adds; WHICH operations are fused is part of the result,
//     so the fused forms below follow the machine code of that libm.so.6 operation by
//     operation (fma() is exact on both sides: v_fma_f64 here, vfmadd*sd there).
//   * atanf, atan2f, acosf — Sun's fdlibm in float (s_atanf.c, e_atan2f.c, e_acosf.c): plain
//     float arithmetic in source order (no ifunc variant, baseline SSE2, nothing fused).
// The table values were read out of that libm.so.6 (they are the published ones).
// Upstream copyright notices and licence texts (LGPL-2.1-or-later for glibc, the Sun fdlibm notice):
// csrc/licenses/README.md.
//
// Checked in tests/cpp/libm_check.cpp (CPU, `-m "not gpu"`): this header compiled for the
// host against the live glibc — every one of the 2^32 arguments of the one-argument functions,
// 2^28 seeded pairs + the special values for atan2f / powf — and on the GPU box against
// what the device computes (tests/test_gpu_libm.py).  NaN results are NaNs (payload and sign
// of a NaN are not reproduced: no comparison on the path can see them).
#pragma once

#include <stdint.h>

#ifdef __HIPCC__
#define YT_LIBM_FN __device__ __forceinline__
#define YT_LIBM_TABLE static __device__ const
#define YT_LIBM_BIG __device__ __forceinline__
#else
#define YT_LIBM_FN static inline
#define YT_LIBM_BIG static inline
#define YT_LIBM_TABLE static const
#endif

namespace ytm {

YT_LIBM_FN uint32_t asuint(float f) { return __builtin_bit_cast(uint32_t, f); }
YT_LIBM_FN float    asfloat(uint32_t u) { return __builtin_bit_cast(float, u); }
YT_LIBM_FN uint64_t asuint64(double f) { return __builtin_bit_cast(uint64_t, f); }
YT_LIBM_FN double   asdouble(uint64_t u) { return __builtin_bit_cast(double, u); }
YT_LIBM_FN double   fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
YT_LIBM_FN float    nanf_() { return asfloat(0x7fc00000u); }

// ---------------------------------------------------------------------------
// sinf / cosf — sincosf.h, s_sinf.c, s_cosf.c
// ---------------------------------------------------------------------------
// __sincosf_table[0] (sincos_t: sign[4] = {1, -1, -1, 1}, hpi_inv, hpi, c0..c4, s1..s3).
// __sincosf_table[1] is the same with the cosine coefficients negated — it is selected
// when n & 2 — and IEEE arithmetic is sign-symmetric, so evaluating the cosine with table 0
// and negating the result gives the same bits; likewise sign[n & 3] is -1 exactly when
// (n + 1) & 2.  That keeps every constant a literal: no table loads in the dependency chain.
constexpr double SC_HPI_INV = 0x1.45f306dc9c883p+23, SC_HPI = 0x1.921fb54442d18p+0;
constexpr double SC_C0 = 0x1.0000000000000p+0, SC_C1 = -0x1.ffffffd0c621cp-2, SC_C2 = 0x1.55553e1068f19p-5,
                 SC_C3 = -0x1.6c087e89a359dp-10, SC_C4 = 0x1.99343027bf8c3p-16;
constexpr double SC_S1 = -0x1.555545995a603p-3, SC_S2 = 0x1.1107605230bc4p-7, SC_S3 = -0x1.994eb3774cf24p-13;
// 4/PI as 24 overlapping 32-bit words (__inv_pio4)
YT_LIBM_TABLE uint32_t inv_pio4[24] = {0xa2, 0xa2f9, 0xa2f983, 0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529,
    0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0, 0x34ddc0db,
    0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};

YT_LIBM_FN uint32_t abstop12(float x) { return (asuint(x) >> 20) & 0x7ff; }

// sinf_poly's two branches, as fused in the _fma build: the sine polynomial of x (x2 = x * x) ...
YT_LIBM_FN double sin_poly(double x, double x2) {
  double x3 = x * x2;
  double s1 = fma_(x2, SC_S3, SC_S2);
  double x7 = x3 * x2;
  double s  = fma_(x3, SC_S1, x);
  return fma_(s1, x7, s);
}
// ... and the cosine polynomial (table 0)
YT_LIBM_FN double cos_poly(double x2) {
  double x4 = x2 * x2;
  double c1 = fma_(x2, SC_C1, SC_C0);
  double c2 = fma_(x2, SC_C4, SC_C3);
  double x6 = x4 * x2;
  double c  = fma_(x4, SC_C2, c1);
  return fma_(c2, x6, c);
}
// reduce_fast: |x| < 120
YT_LIBM_FN double reduce_fast(double x, int* np) {
  double r = x * SC_HPI_INV;
  int    n = ((int32_t)r + 0x800000) >> 24;
  *np      = n;
  return fma_(-(double)n, SC_HPI, x);
}
// reduce_large: 120 <= |x| < inf
YT_LIBM_FN double reduce_large(uint32_t xi, int* np) {
  const uint32_t* arr   = &inv_pio4[(xi >> 26) & 15];
  int             shift = (xi >> 23) & 7;
  uint64_t        n, res0, res1, res2;
  xi = (xi & 0xffffff) | 0x800000;
  xi <<= shift;
  res0 = (uint64_t)(uint32_t)(xi * arr[0]);
  res1 = (uint64_t)xi * arr[4];
  res2 = (uint64_t)xi * arr[8];
  res0 = (res2 >> 32) | (res0 << 32);
  res0 += res1;
  n = (res0 + (1ULL << 61)) >> 62;
  res0 -= n << 62;
  double x = (double)(int64_t)res0;
  *np      = (int)n;
  return x * 0x1.921fb54442d18p-62;
}
// the argument reduction shared by sinf / cosf / sincosf: x reduced to [-pi/4, pi/4] times the
// quadrant sign, n the quadrant (n & 1: swap sine and cosine; m & 2, m = n (+ sign): negate the cosine)
YT_LIBM_FN bool sincos_reduce(float y, double* xs, double* x2, int* n, int* m) {
  double x = y;
  if (abstop12(y) < 0x42f) {  // |y| < 120
    x  = reduce_fast(x, n);
    *m = *n;
  } else if (abstop12(y) < 0x7f8) {
    uint32_t xi = asuint(y);
    x           = reduce_large(xi, n);
    *m          = *n + (int)(xi >> 31);
  } else {
    return false;
  }
  *xs = ((*m + 1) & 2) ? -x : x;  // x * sign[m & 3]
  *x2 = x * x;
  return true;
}

// (glibc branches off |y| < pi/4 (no reduction) and |y| < 2^-12 (sinf returns y, cosf 1) first.
// Both are what the general path computes anyway — there n = 0, x - 0 * hpi = x exactly and the
// polynomials round to y / 1 — so the wavefront runs ONE path for every argument below 120
// instead of diverging; equality with glibc is checked over all 2^32 arguments.)
YT_LIBM_FN float sinf(float y) {
  double xs, x2;
  int    n, m;
  if (!sincos_reduce(y, &xs, &x2, &n, &m)) return nanf_();
  if ((asuint(y) << 1) == 0) return y;  // sin(-0) = -0 (the polynomial would give +0)
  double c = cos_poly(x2), sv = sin_poly(xs, x2);
  return (float)((n & 1) ? ((m & 2) ? -c : c) : sv);
}
YT_LIBM_FN float cosf(float y) {
  double xs, x2;
  int    n, m;
  if (!sincos_reduce(y, &xs, &x2, &n, &m)) return nanf_();
  double c = cos_poly(x2), sv = sin_poly(xs, x2);
  return (float)((n & 1) ? sv : ((m & 2) ? -c : c));
}
// sincosf: glibc's gives exactly sinf's and cosf's values (same reduction, same
// polynomials; checked over all 2^32 arguments) — one reduction for both
YT_LIBM_FN void sincosf(float y, float* sinp, float* cosp) {
  double xs, x2;
  int    n, m;
  if (!sincos_reduce(y, &xs, &x2, &n, &m)) {
    *sinp = *cosp = nanf_();
    return;
  }
  double c  = cos_poly(x2);
  float  sv = (float)sin_poly(xs, x2), cv = (float)((m & 2) ? -c : c);
  if ((asuint(y) << 1) == 0) sv = y;  // sin(-0) = -0
  *sinp = (n & 1) ? cv : sv;
  *cosp = (n & 1) ? sv : cv;
}

// ---------------------------------------------------------------------------
// expf / exp2f — e_expf.c, e_exp2f.c (__exp2f_data, EXP2F_TABLE_BITS = 5)
// ---------------------------------------------------------------------------
YT_LIBM_TABLE uint64_t exp2f_tab[32] = {0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51,
    0x3fef72b83c7d517b, 0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1, 0x3fef06fe0a31b715,
    0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d, 0x3feebfdad5362a27, 0x3feeb42b569d4f82,
    0x3feeab07dd485429, 0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187,
    0x3feea589994cce13, 0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
    0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487,
    0x3fef7c97337b9b5f, 0x3fefa4afa2a490da, 0x3fefd0765b6e4540};
constexpr double EXP2F_SHIFT_SCALED = 0x1.8p+47;  // 0x1.8p52 / 32
constexpr double EXP2F_C0 = 0x1.c6af84b912394p-5, EXP2F_C1 = 0x1.ebfce50fac4f3p-3, EXP2F_C2 = 0x1.62e42ff0c52d6p-1;
constexpr double EXPF_SHIFT = 0x1.8p+52, EXPF_INVLN2N = 0x1.71547652b82fep+5;  // 32 / ln 2
constexpr double EXPF_C0 = 0x1.c6af84b912394p-20, EXPF_C1 = 0x1.ebfce50fac4f3p-13, EXPF_C2 = 0x1.62e42ff0c52d6p-6;

YT_LIBM_FN float oflowf(uint32_t sign) { return (sign ? -0x1p97f : 0x1p97f) * 0x1p97f; }
YT_LIBM_FN float uflowf(uint32_t sign) { return (sign ? -0x1p-95f : 0x1p-95f) * 0x1p-95f; }
YT_LIBM_FN float may_uflowf(uint32_t sign) { return (sign ? -0x1.4p-75f : 0x1.4p-75f) * 0x1.4p-75f; }

YT_LIBM_BIG float expf(float x) {
  double   xd     = (double)x;
  uint32_t abstop = abstop12(x);
  if (abstop >= 0x42b) {  // |x| >= 88 or x is nan
    if (asuint(x) == 0xff800000u) return 0.0f;
    if (abstop >= 0x7f8) return x + x;
    if (x > 0x1.62e42ep6f) return oflowf(0);
    if (x < -0x1.9fe368p6f) return uflowf(0);
    if (x < -0x1.9d1d9ep6f) return may_uflowf(0);
  }
  // x * N / ln2 = k + r; in the _fma build z + SHIFT and z - kd are single fused operations on x
  double   kd = fma_(EXPF_INVLN2N, xd, EXPF_SHIFT);
  uint64_t ki = asuint64(kd);
  kd -= EXPF_SHIFT;
  double   r = fma_(EXPF_INVLN2N, xd, -kd);
  uint64_t t = exp2f_tab[ki % 32];
  t += ki << (52 - 5);
  double s  = asdouble(t);
  double z  = fma_(EXPF_C0, r, EXPF_C1);
  double r2 = r * r;
  double y  = fma_(EXPF_C2, r, 1.0);
  y         = fma_(z, r2, y);
  y         = y * s;
  return (float)y;
}
YT_LIBM_BIG float exp2f(float x) {
  double   xd     = (double)x;
  uint32_t abstop = abstop12(x);
  if (abstop >= 0x430) {  // |x| >= 128 or x is nan
    if (asuint(x) == 0xff800000u) return 0.0f;
    if (abstop >= 0x7f8) return x + x;
    if (x > 0.0f) return oflowf(0);
    if (x <= -150.0f) return uflowf(0);
    if (x < -149.0f) return may_uflowf(0);
  }
  double   kd = xd + EXP2F_SHIFT_SCALED;
  uint64_t ki = asuint64(kd);
  kd -= EXP2F_SHIFT_SCALED;
  double   r = xd - kd;
  uint64_t t = exp2f_tab[ki % 32];
  t += ki << (52 - 5);
  double s  = asdouble(t);
  double z  = fma_(EXP2F_C0, r, EXP2F_C1);
  double r2 = r * r;
  double y  = fma_(EXP2F_C2, r, 1.0);
  y         = fma_(z, r2, y);
  y         = y * s;
  return (float)y;
}

// ---------------------------------------------------------------------------
// logf — e_logf.c (__logf_data, LOGF_TABLE_BITS = 4)
// ---------------------------------------------------------------------------
YT_LIBM_TABLE double logf_tab[16][2] = {{0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010b0p+0, -0x1.01eae7f513a67p-2}, {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8ea0p+0, -0x1.1aa2bc79c8100p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5}, {0x1.ca4b31f026aa0p-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3}, {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d224770p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2}, {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
constexpr double LOGF_LN2 = 0x1.62e42fefa39efp-1, LOGF_A0 = -0x1.00ea348b88334p-2, LOGF_A1 = 0x1.5575b0be00b6ap-2,
                 LOGF_A2 = -0x1.ffffef20a4123p-2;

YT_LIBM_BIG float logf(float x) {
  uint32_t ix = asuint(x);
  if (ix == 0x3f800000u) return 0.0f;
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
    // x < 0x1p-126 or inf or nan
    if (ix * 2 == 0) return -1.0f / 0.0f;    // __math_divzerof (1)
    if (ix == 0x7f800000u) return x;         // log(inf) == inf
    if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return nanf_();
    ix = asuint(x * 0x1p23f);  // subnormal: normalize
    ix -= 23u << 23;
  }
  uint32_t tmp  = ix - 0x3f330000u;
  int      i    = (int)((tmp >> 19) % 16);
  int      k    = (int32_t)tmp >> 23;
  uint32_t iz   = ix - (tmp & 0xff800000u);
  double   invc = logf_tab[i][0], logc = logf_tab[i][1];
  double   z    = (double)asfloat(iz);
  // log(x) = log1p(z/c-1) + log(c) + k*Ln2
  double r  = fma_(z, invc, -1.0);
  double y0 = fma_((double)k, LOGF_LN2, logc);
  double r2 = r * r;
  double y  = fma_(LOGF_A1, r, LOGF_A2);
  y         = fma_(LOGF_A0, r2, y);
  y         = fma_(y, r2, y0 + r);
  return (float)y;
}

// ---------------------------------------------------------------------------
// powf — e_powf.c (__powf_log2_data, POWF_LOG2_TABLE_BITS = 4, POWF_SCALE_BITS = 0)
// ---------------------------------------------------------------------------
YT_LIBM_TABLE double powf_log2_tab[16][2] = {{0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2},
    {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2}, {0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2},
    {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2}, {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2},
    {0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3}, {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3},
    {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4}, {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5},
    {0x1.0000000000000p+0, 0x0.0p+0}, {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},
    {0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3}, {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},
    {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2}, {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},
    {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
constexpr double POWF_A0 = 0x1.27616c9496e0bp-2, POWF_A1 = -0x1.71969a075c67ap-2, POWF_A2 = 0x1.ec70a6ca7baddp-2,
                 POWF_A3 = -0x1.7154748bef6c8p-1, POWF_A4 = 0x1.71547652ab82bp+0;

YT_LIBM_FN double powf_log2_inline(uint32_t ix) {
  uint32_t tmp  = ix - 0x3f330000u;
  int      i    = (int)((tmp >> 19) % 16);
  uint32_t top  = tmp & 0xff800000u;
  uint32_t iz   = ix - top;
  int      k    = (int32_t)top >> 23;
  double   invc = powf_log2_tab[i][0], logc = powf_log2_tab[i][1];
  double   z    = (double)asfloat(iz);
  // log2(x) = log1p(z/c-1)/ln2 + log2(c) + k
  double r  = fma_(z, invc, -1.0);
  double y0 = logc + (double)k;
  double r2 = r * r;
  double y  = fma_(POWF_A0, r, POWF_A1);
  double p  = fma_(POWF_A2, r, POWF_A3);
  double r4 = r2 * r2;
  double q  = fma_(POWF_A4, r, y0);
  q         = fma_(p, r2, q);
  y         = fma_(y, r4, q);
  return y;
}
YT_LIBM_FN float powf_exp2_inline(double xd, uint32_t sign_bias) {
  double   kd = xd + EXP2F_SHIFT_SCALED;
  uint64_t ki = asuint64(kd);
  kd -= EXP2F_SHIFT_SCALED;
  double   r   = xd - kd;
  uint64_t t   = exp2f_tab[ki % 32];
  uint64_t ski = ki + sign_bias;
  t += ski << (52 - 5);
  double s  = asdouble(t);
  double z  = fma_(EXP2F_C0, r, EXP2F_C1);
  double r2 = r * r;
  double y  = fma_(EXP2F_C2, r, 1.0);
  y         = fma_(z, r2, y);
  y         = y * s;
  return (float)y;
}
// 0 not an integer, 1 odd integer, 2 even integer (iy is the bit pattern of y)
YT_LIBM_FN int powf_checkint(uint32_t iy) {
  int e = (int)(iy >> 23 & 0xff);
  if (e < 0x7f) return 0;
  if (e > 0x7f + 23) return 2;
  if (iy & ((1u << (0x7f + 23 - e)) - 1)) return 0;
  if (iy & (1u << (0x7f + 23 - e))) return 1;
  return 2;
}
YT_LIBM_FN bool powf_zeroinfnan(uint32_t ix) { return 2 * ix - 1 >= 2u * 0x7f800000u - 1; }

YT_LIBM_BIG float powf(float x, float y) {
  uint32_t sign_bias = 0;
  uint32_t ix = asuint(x), iy = asuint(y);
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || powf_zeroinfnan(iy)) {
    // either (x < 0x1p-126 or inf or nan) or (y is 0 or inf or nan)
    if (powf_zeroinfnan(iy)) {
      if (2 * iy == 0) return ((ix ^ 0x00400000u) & 0x7fffffffu) > 0x7fc00000u ? x + y : 1.0f;  // (signalling nan ^ 0)
      if (ix == 0x3f800000u) return ((iy ^ 0x00400000u) & 0x7fffffffu) > 0x7fc00000u ? x + y : 1.0f;
      if (2 * ix > 2u * 0x7f800000u || 2 * iy > 2u * 0x7f800000u) return x + y;
      if (2 * ix == 2 * 0x3f800000u) return 1.0f;
      if ((2 * ix < 2 * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;  // |x| < 1 && y == inf or |x| > 1 && y == -inf
      return y * y;
    }
    if (powf_zeroinfnan(ix)) {
      float x2 = x * x;
      if ((ix & 0x80000000u) && powf_checkint(iy) == 1) {
        x2        = -x2;
        sign_bias = 1;
      }
      if (2 * ix == 0 && (iy & 0x80000000u)) return (sign_bias ? -1.0f : 1.0f) / 0.0f;  // __math_divzerof
      return (iy & 0x80000000u) ? 1 / x2 : x2;
    }
    // x and y are non-zero finite
    if (ix & 0x80000000u) {
      int yint = powf_checkint(iy);  // finite x < 0
      if (yint == 0) return nanf_();
      if (yint == 1) sign_bias = 1u << (5 + 11);
      ix &= 0x7fffffffu;
    }
    if (ix < 0x00800000u) {
      ix = asuint(x * 0x1p23f);  // normalize subnormal x so the exponent becomes negative
      ix &= 0x7fffffffu;
      ix -= 23u << 23;
    }
  }
  double logx  = powf_log2_inline(ix);
  double ylogx = (double)y * logx;  // cannot overflow, y is single precision
  if ((asuint64(ylogx) >> 47 & 0xffff) >= (asuint64(126.0) >> 47)) {
    // |y * log(x)| >= 126
    if (ylogx > 0x1.fffffffd1d571p+6) return oflowf(sign_bias);
    // (round-to-nearest: the WANT_ROUNDING test between 0x1.fffffffa3aae2p+6 and the bound above never fires)
    if (ylogx <= -150.0) return uflowf(sign_bias);
    if (ylogx < -149.0) return may_uflowf(sign_bias);
  }
  return powf_exp2_inline(ylogx, sign_bias);
}

// ---------------------------------------------------------------------------
// atanf, atan2f, acosf — fdlibm in float (s_atanf.c, e_atan2f.c, e_acosf.c): float
// arithmetic in source order, nothing fused (hipcc: -ffp-contract=off)
// ---------------------------------------------------------------------------
YT_LIBM_FN float fabsf_(float x) { return asfloat(asuint(x) & 0x7fffffffu); }

YT_LIBM_BIG float atanf(float x) {
  const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
  const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
  const float aT[11]    = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f,
         9.0908870101e-02f, -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f,
         -3.6531571299e-02f, 1.6285819933e-02f};
  float   w, s1, s2, z;
  int32_t hx = (int32_t)asuint(x), ix = hx & 0x7fffffff, id;
  // -DYT_LIBM_NO_TABLES (lead for round 5; off: unchanged): atanhi[id] / atanlo[id] with a run-time id become two
  // DEPENDENT loads from a constant table in global memory on the device (hipcc -S: global_load_dword, s_waitcnt vmcnt(0), twice) —
  // in every eval_environment.  The interval's two constants are picked where the interval is decided instead; same floats.
  float hi = 0, lo = 0;
  (void)hi, (void)lo;
  if (ix >= 0x4c000000) {  // |x| >= 2^25
    if (ix > 0x7f800000) return x + x;
    if (hx > 0) return atanhi[3] + atanlo[3];
    return -atanhi[3] - atanlo[3];
  }
  if (ix < 0x3ee00000) {                  // |x| < 0.4375
    if (ix < 0x31000000) return x;        // |x| < 2^-29
    id = -1;
  } else {
    x = fabsf_(x);
    if (ix < 0x3f980000) {    // |x| < 1.1875
      if (ix < 0x3f300000) {  // 7/16 <= |x| < 11/16
        id = 0, hi = atanhi[0], lo = atanlo[0];
        x  = (2.0f * x - 1.0f) / (2.0f + x);
      } else {  // 11/16 <= |x| < 19/16
        id = 1, hi = atanhi[1], lo = atanlo[1];
        x  = (x - 1.0f) / (x + 1.0f);
      }
    } else {
      if (ix < 0x401c0000) {  // |x| < 2.4375
        id = 2, hi = atanhi[2], lo = atanlo[2];
        x  = (x - 1.5f) / (1.0f + 1.5f * x);
      } else {  // 2.4375 <= |x| < 2^25
        id = 3, hi = atanhi[3], lo = atanlo[3];
        x  = -1.0f / x;
      }
    }
  }
  z  = x * x;
  w  = z * z;
  s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
  s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
  if (id < 0) return x - x * (s1 + s2);
  z = hi - ((x * (s1 + s2) - lo) - x);
  return (hx < 0) ? -z : z;
}

YT_LIBM_BIG float atan2f(float y, float x) {
  const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f,
              pi_lo = -8.7422776573e-08f;
  float   z;
  int32_t hx = (int32_t)asuint(x), ix = hx & 0x7fffffff;
  int32_t hy = (int32_t)asuint(y), iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;  // x or y is NaN
  if (hx == 0x3f800000) return atanf(y);                 // x = 1.0
  int32_t m = ((hy >> 31) & 1) | ((hx >> 30) & 2);       // 2 * sign(x) + sign(y)
  if (iy == 0) {                                         // y = 0
    switch (m) {
      case 0:
      case 1: return y;
      case 2: return pi + tiny;
      default: return -pi - tiny;
    }
  }
  if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;  // x = 0
  if (ix == 0x7f800000) {                                         // x is INF
    if (iy == 0x7f800000) {
      switch (m) {
        case 0: return pi_o_4 + tiny;
        case 1: return -pi_o_4 - tiny;
        case 2: return 3.0f * pi_o_4 + tiny;
        default: return -3.0f * pi_o_4 - tiny;
      }
    } else {
      switch (m) {
        case 0: return 0.0f;
        case 1: return -0.0f;
        case 2: return pi + tiny;
        default: return -pi - tiny;
      }
    }
  }
  if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;  // y is INF
  int32_t k = (iy - ix) >> 23;
  if (k > 60)
    z = pi_o_2 + 0.5f * pi_lo;  // |y / x| > 2^60
  else if (hx < 0 && k < -60)
    z = 0.0f;  // |y| / x < -2^60
  else
    z = atanf(fabsf_(y / x));
  switch (m) {
    case 0: return z;
    case 1: return asfloat(asuint(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}

YT_LIBM_FN float sqrtf_(float x) { return __builtin_sqrtf(x); }

YT_LIBM_BIG float acosf(float x) {
  const float pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f,
              pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f,
              pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f, qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f,
              qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
  float   z, p, q, r, w, s, c, df;
  int32_t hx = (int32_t)asuint(x), ix = hx & 0x7fffffff;
  if (ix == 0x3f800000) {  // |x| == 1
    if (hx > 0) return 0.0f;
    return pi + 2.0f * pio2_lo;
  } else if (ix > 0x3f800000) {
    return nanf_();  // |x| > 1 (or NaN)
  }
  if (ix < 0x3f000000) {  // |x| < 0.5
    if (ix <= 0x32800000) return pio2_hi + pio2_lo;  // |x| <= 2^-26
    z = x * x;
    p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    r = p / q;
    return pio2_hi - (x - (pio2_lo - x * r));
  } else if (hx < 0) {  // x < -0.5
    z = (1.0f + x) * 0.5f;
    p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    s = sqrtf_(z);
    r = p / q;
    w = r * s - pio2_lo;
    return pi - 2.0f * (s + w);
  } else {  // x > 0.5
    z  = (1.0f - x) * 0.5f;
    s  = sqrtf_(z);
    df = asfloat(asuint(s) & 0xfffff000u);
    c  = (z - df * df) / (s + df);
    p  = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    q  = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    r  = p / q;
    w  = r * s + c;
    return 2.0f * (df + w);
  }
}

}  // namespace ytm
