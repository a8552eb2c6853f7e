// yt_denoise.h — an on-device denoiser for the hand-off of yocto_trace.cpp:1794-1872
// (SURVEY.md §8(f) rank 3).
//
// The reference's `denoise_image` is Intel OIDN's "RT" filter (a neural network, not vendored:
// exts/openimagedenoise is a CMake finder) or, in the default build, a plain copy.  There is no
// reference algorithm to be in parity with; what the reference fixes is the INTERFACE: HDR
// colour + the first-hit albedo and normal means that trace_sample keeps, all three resident
// on the device here.  This file fills that slot with the classical guide-driven filter:
// an edge-avoiding à-trous wavelet (Dammertz, Sewtz, Hanika, Lensch, HPG 2010) on the
// albedo-demodulated colour, specified completely below so that tests/denoise_check.py can
// restate it in numpy:
//
//   den(p)  = max(albedo(p), 0.01)                    irr_0(p) = colour(p) / den(p)
//   for level l = 0 .. levels-1, step s = 2^l:
//     irr_{l+1}(p) = Σ_q w(p,q) irr_l(q) / Σ_q w(p,q),   q = p + s·(dx, dy), dx, dy ∈ [-2, 2], q inside
//     w(p,q) = k[|dx|]·k[|dy|] · exp(-(dn/σn² + da/σa² + dc·4^l/σc²)),    k = {3/8, 1/4, 1/16}
//     dn = |normal(p) - normal(q)|², da = |albedo(p) - albedo(q)|²,
//     dc = |irr_l(p) - irr_l(q)|² / (max(lum(irr_l(p)), lum(irr_l(q)))² + 1e-4)
//   out(p) = { irr_levels(p) · den(p), alpha(p) }
//
// Taps are visited row-major (dy outer, dx inner), sums are plain float adds in that order,
// -ffp-contract=off: the numpy restatement differs only through `exp` (v_exp_f32 here).
// HBM-bound stencil: per level and pixel 48 B read + 16 B written, the 24 neighbour taps come
// from L1 / L2; one thread per pixel, 64 x 4 pixel workgroups so a wavefront reads 1 KB runs.
#pragma once
#include <hip/hip_runtime.h>

namespace ytdn {

constexpr int BX = 64, BY = 4;

struct Params {
  int   width, height, levels;
  float inv_sn2, inv_sa2, inv_sc2;  // 1 / sigma^2
};

__device__ __forceinline__ float lum(float4 c) { return (c.x + c.y + c.z) / 3; }
__device__ __forceinline__ float dist2(float4 a, float4 b) {
  float x = a.x - b.x, y = a.y - b.y, z = a.z - b.z;
  return x * x + y * y + z * z;
}

// colour / albedo / normal of a trace_state (vec4f, vec3f, vec3f) → working images
__global__ void __launch_bounds__(BX* BY) k_prep(const float4* image, const float* albedo, const float* normal, int n,
    float4* irr, float4* gn, float4* ga) {
  int i = blockIdx.x * (BX * BY) + threadIdx.x;
  if (i >= n) return;
  float4 c = image[i];
  float4 a = {albedo[3 * i], albedo[3 * i + 1], albedo[3 * i + 2], 0};
  float4 m = {normal[3 * i], normal[3 * i + 1], normal[3 * i + 2], 0};
  float  dx = fmaxf(a.x, 0.01f), dy = fmaxf(a.y, 0.01f), dz = fmaxf(a.z, 0.01f);
  irr[i] = {c.x / dx, c.y / dy, c.z / dz, c.w};
  gn[i]  = m;
  ga[i]  = a;
}

__global__ void __launch_bounds__(BX* BY) k_atrous(const float4* __restrict__ in, const float4* __restrict__ gn,
    const float4* __restrict__ ga, float4* __restrict__ out, Params p, int step, float inv_sc2_l) {
  const int x = blockIdx.x * BX + threadIdx.x % BX, y = blockIdx.y * BY + threadIdx.x / BX;
  if (x >= p.width || y >= p.height) return;
  const int    i  = y * p.width + x;
  const float4 cp = in[i], np = gn[i], ap = ga[i];
  const float  lp = lum(cp);
  const float  k[3] = {3.0f / 8, 1.0f / 4, 1.0f / 16};
  float sx = 0, sy = 0, sz = 0, sw = 0;
#pragma unroll
  for (int dy = -2; dy <= 2; dy++) {
    const int qy = y + dy * step;
    if (qy < 0 || qy >= p.height) continue;
#pragma unroll
    for (int dx = -2; dx <= 2; dx++) {
      const int qx = x + dx * step;
      if (qx < 0 || qx >= p.width) continue;
      const int    j  = qy * p.width + qx;
      const float4 cq = in[j], nq = gn[j], aq = ga[j];
      const float  m  = fmaxf(lp, lum(cq));
      const float  dc = dist2(cp, cq) / (m * m + 1e-4f);
      const float  d  = dist2(np, nq) * p.inv_sn2 + dist2(ap, aq) * p.inv_sa2 + dc * inv_sc2_l;
      const float  w  = k[dx < 0 ? -dx : dx] * k[dy < 0 ? -dy : dy] * __expf(-d);
      sx += w * cq.x, sy += w * cq.y, sz += w * cq.z, sw += w;
    }
  }
  out[i] = {sx / sw, sy / sw, sz / sw, cp.w};
}

__global__ void __launch_bounds__(BX* BY) k_finish(const float4* irr, const float4* ga, int n, float4* out) {
  int i = blockIdx.x * (BX * BY) + threadIdx.x;
  if (i >= n) return;
  float4 c = irr[i], a = ga[i];
  out[i]   = {c.x * fmaxf(a.x, 0.01f), c.y * fmaxf(a.y, 0.01f), c.z * fmaxf(a.z, 0.01f), c.w};
}

}  // namespace ytdn
