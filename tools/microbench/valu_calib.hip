// valu_calib.hip — what does "VALU busy" mean on gfx950?  (calibrate the
// cycles-per-VALU-instruction figure the roofline uses with a VALU-only microbenchmark.)
//
// Kernels that do nothing but issue VALU instructions of one kind from W waves per SIMD
// (W = 1, 2, 4 — k_trace runs at 4), timed with hipEvents and with the shader clock, so
//   wave-instructions / (SIMD x cycle)
// at TRUE peak is known per instruction kind; the same binary run under
//   rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
// tells what the counters read at that peak.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int UNROLL = 64;  // VALU instructions per loop iteration (16 independent chains x 4)

// kind 0: v_fma_f32   1: v_mul_f32 + v_add_f32 (no contraction, as the tracer is built)
//      2: v_cndmask_b32 / v_cmp_lt_f32 pairs   3: v_min3_f32 / v_max3_f32
//      4: v_rcp_f32 (transcendental)           5: v_mul_lo_u32 (PCG's 64-bit multiply is made of these)
//      6: v_pk_mul_f32 (packed)                7: IEEE division a / b (the v_div_scale/fmas/fixup sequence)
template <int KIND>
__global__ void __launch_bounds__(64) k_valu(float* out, int iters, float seed, unsigned long long* cycles) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = seed + (float)(threadIdx.x + i);
  unsigned u[16];
#pragma unroll
  for (int i = 0; i < 16; i++) u[i] = (unsigned)(threadIdx.x * 16 + i) | 1u;
  const float b = seed * 1.0001f, c = seed * 0.5f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < UNROLL / 16; r++) {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (KIND == 1) { if (r & 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c)); else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); }
        if (KIND == 2) { if (r & 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c)); else asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(b) : "vcc"); }
        if (KIND == 3) { if (r & 1) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b)); }
        if (KIND == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        if (KIND == 5) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
        if (KIND == 6) { if (i % 2 == 0) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&a[i]) : "v"(*(const double*)&a[(i + 2) & 15])); }
        if (KIND == 7) { if (i < 4) a[i] = a[i] / (b + a[i + 4]); }
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += a[i] + (float)u[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = (unsigned long long)(t1 - t0);
}

template <int KIND>
void run(const char* name, int insts_per_iter, int waves_per_simd, int cus, float* d_out, unsigned long long* d_cyc) {
  const int iters = 20000;
  const int grid  = cus * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_valu<KIND>, dim3(grid), dim3(64), 0, 0, d_out, 100, 1.0f, d_cyc);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_valu<KIND>, dim3(grid), dim3(64), 0, 0, d_out, iters, 1.0f, d_cyc);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> cyc(grid);
  CHECK(hipMemcpy(cyc.data(), d_cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double mean = 0;
  for (auto c : cyc) mean += (double)c;
  mean /= grid;
  const double winst = (double)iters * insts_per_iter;  // wave-instructions per wave
  // per SIMD: waves_per_simd waves share it
  printf("%-28s waves/SIMD %d  %8.3f ms  shader cycles/wave %.3e  -> %.3f cycles per wave-instruction per SIMD "
         "(%.3f wave-instr / SIMD / cycle); clock %.2f GHz\n",
      name, waves_per_simd, ms, mean, mean / (winst * waves_per_simd), winst * waves_per_simd / mean,
      mean / (ms * 1e6));
}

int main() {
  int dev = 0, cus = 0;
  CHECK(hipSetDevice(dev));
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  float*              d_out;
  unsigned long long* d_cyc;
  CHECK(hipMalloc(&d_out, (size_t)cus * 4 * 8 * 64 * sizeof(float)));
  CHECK(hipMalloc(&d_cyc, (size_t)cus * 4 * 8 * sizeof(unsigned long long)));
  printf("device CUs %d\n", cus);
  for (int w : {1, 2, 4}) {
    run<0>("v_fma_f32", UNROLL, w, cus, d_out, d_cyc);
    run<1>("v_mul_f32 / v_add_f32", UNROLL, w, cus, d_out, d_cyc);
    run<2>("v_cmp_lt_f32 / v_cndmask_b32", UNROLL, w, cus, d_out, d_cyc);
    run<3>("v_min3_f32 / v_max3_f32", UNROLL, w, cus, d_out, d_cyc);
    run<4>("v_rcp_f32", UNROLL, w, cus, d_out, d_cyc);
    run<5>("v_mul_lo_u32", UNROLL, w, cus, d_out, d_cyc);
    run<6>("v_pk_mul_f32 (2 flop/lane)", UNROLL / 2, w, cus, d_out, d_cyc);
    run<7>("a / b (IEEE f32 division)", UNROLL / 4, w, cus, d_out, d_cyc);
  }
  return 0;
}
