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
// HBM-bound stencil: per level and pixel 48 B read + 16 B written.  k_atrous_lds (spacings up
// to 16) stages tile + halo through LDS; k_atrous (any spacing, and the cross-check) takes the
// 24 neighbour taps from L1 / L2, one thread per pixel, 64 x 4 pixel workgroups.
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

// The same level with the taps staged through LDS.  An à-trous level of spacing s is s x s
// independent dense 5x5 filters, one per residue class (x mod s, y mod s): a workgroup takes a
// TX x TY tile of ONE class's sub-image, loads tile + 2-pixel halo (in sub-image coordinates)
// of the three inputs into LDS once — 1.7 global loads per pixel and input instead of 25 — and
// runs the identical tap loop (same order, same arithmetic: bit-identical to k_atrous) on LDS.
// Lanes of a wavefront are `s` pixels apart in memory; the loads of a level still touch every
// byte once.  20.7 KB of LDS per workgroup.
constexpr int TX = 32, TY = 8, HALO = 2, LW = TX + 2 * HALO, LH = TY + 2 * HALO;
__global__ void __launch_bounds__(TX* TY) k_atrous_lds(const float4* __restrict__ in, const float4* __restrict__ gn,
    const float4* __restrict__ ga, float4* __restrict__ out, Params p, int step, float inv_sc2_l) {
  __shared__ float4 s_c[LH][LW], s_n[LH][LW], s_a[LH][LW];
  const int cx = blockIdx.z % step, cy = blockIdx.z / step;       // residue class
  const int u0 = blockIdx.x * TX, v0 = blockIdx.y * TY;            // tile origin in the sub-image
  if (cx + (long long)u0 * step >= p.width || cy + (long long)v0 * step >= p.height) return;  // empty tile (uniform)
  for (int e = threadIdx.x; e < LW * LH; e += TX * TY) {
    const int       lu = e % LW, lv = e / LW;
    const long long qx = cx + (long long)(u0 + lu - HALO) * step, qy = cy + (long long)(v0 + lv - HALO) * step;
    if (qx >= 0 && qx < p.width && qy >= 0 && qy < p.height) {
      const long long j = qy * p.width + qx;
      s_c[lv][lu] = in[j], s_n[lv][lu] = gn[j], s_a[lv][lu] = ga[j];
    }
  }
  __syncthreads();
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int x = cx + (u0 + tx) * step, y = cy + (v0 + ty) * step;
  if (x >= p.width || y >= p.height) return;
  const float4 cp = s_c[ty + HALO][tx + HALO], np = s_n[ty + HALO][tx + HALO], ap = s_a[ty + HALO][tx + HALO];
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
      const float4 cq = s_c[ty + HALO + dy][tx + HALO + dx], nq = s_n[ty + HALO + dy][tx + HALO + dx],
                   aq = s_a[ty + HALO + dy][tx + HALO + dx];
      const float  m  = fmaxf(lp, lum(cq));
      const float  dc = dist2(cp, cq) / (m * m + 1e-4f);
      const float  d  = dist2(np, nq) * p.inv_sn2 + dist2(ap, aq) * p.inv_sa2 + dc * inv_sc2_l;
      const float  w  = k[dx < 0 ? -dx : dx] * k[dy < 0 ? -dy : dy] * __expf(-d);
      sx += w * cq.x, sy += w * cq.y, sz += w * cq.z, sw += w;
    }
  }
  out[y * p.width + x] = {sx / sw, sy / sw, sz / sw, cp.w};
}

__global__ void __launch_bounds__(BX* BY) k_finish(const float4* irr, const float4* ga, int n, float4* out) {
  int i = blockIdx.x * (BX * BY) + threadIdx.x;
  if (i >= n) return;
  float4 c = irr[i], a = ga[i];
  out[i]   = {c.x * fmaxf(a.x, 0.01f), c.y * fmaxf(a.y, 0.01f), c.z * fmaxf(a.z, 0.01f), c.w};
}

}  // namespace ytdn
