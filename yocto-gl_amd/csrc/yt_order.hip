// yt_order.hip — longest-tile-first launch order for k_trace.
//
// k_trace gives every 16x4 tile its own one-wave workgroup; the hardware starts workgroups in
// blockIdx order as slots free up, so the end of a launch is a tail in which the last few
// (possibly expensive) tiles run alone.  Tile costs repeat from batch to batch (the same
// pixels), so every workgroup leaves its cycle count behind and the NEXT launch hands out the
// tiles most expensive first — the classic LPT rule; the tail then consists of the cheapest
// tiles.  Only the tile -> workgroup assignment changes: results are bit-identical.
//
// The order is a counting sort over 4096 logarithmic cost classes (6 % wide: the float
// exponent + 4 mantissa bits of the cost) — LPT does not care about the order inside a class —
// in three small kernels on the launch stream: no host work, no synchronisation (a library
// sort cost 1.4 ms of host time per batch when it was tried).
#include <hip/hip_runtime.h>

#include "yt_order.h"

namespace ytorder {

constexpr int NBINS = 4096, BLK = 256;

__device__ __forceinline__ int cost_class(unsigned cost) {  // larger cost -> smaller class id (descending order)
  unsigned k = __float_as_uint((float)cost) >> 19;           // 8 exponent + 4 mantissa bits, monotonic in cost
  return NBINS - 1 - (int)(k < (unsigned)NBINS ? k : (unsigned)NBINS - 1);
}

// (tiles of one frame fall into a handful of classes: the counts are gathered per block in LDS
// first, so the global atomics are one per block and non-empty class, not one per tile)
// (the class of every tile is computed ONCE and kept: the placement pass must see exactly what the
// histogram saw, or the result would not be a permutation — whatever happens to `cost` meanwhile)
__global__ void __launch_bounds__(BLK) k_hist(const unsigned* cost, int n, int* bins, int* cls) {
  __shared__ int s_cnt[NBINS];
  for (int k = threadIdx.x; k < NBINS; k += BLK) s_cnt[k] = 0;
  __syncthreads();
  int i = blockIdx.x * BLK + threadIdx.x;
  if (i < n) {
    const int c = cost_class(cost[i]);
    cls[i]      = c;
    atomicAdd(&s_cnt[c], 1);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < NBINS; k += BLK)
    if (s_cnt[k]) atomicAdd(&bins[k], s_cnt[k]);
}
// exclusive prefix sum of the 4096 class counts, one block
__global__ void __launch_bounds__(1024) k_scan(int* bins) {
  __shared__ int s_part[1024];
  const int t = threadIdx.x;
  int v[4], sum = 0;
  for (int k = 0; k < 4; k++) v[k] = bins[4 * t + k], sum += v[k];
  s_part[t] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    int o = t >= d ? s_part[t - d] : 0;
    __syncthreads();
    s_part[t] += o;
    __syncthreads();
  }
  int run = s_part[t] - sum;
  for (int k = 0; k < 4; k++) bins[4 * t + k] = run, run += v[k];
}
__global__ void __launch_bounds__(BLK) k_place(const int* cls, int n, int* bins, int* perm) {
  __shared__ int s_cnt[NBINS], s_base[NBINS];
  for (int k = threadIdx.x; k < NBINS; k += BLK) s_cnt[k] = 0;
  __syncthreads();
  const int i = blockIdx.x * BLK + threadIdx.x;
  const int c = i < n ? cls[i] : -1;
  int       local = 0;
  if (c >= 0) local = atomicAdd(&s_cnt[c], 1);  // rank inside this block's share of the class
  __syncthreads();
  for (int k = threadIdx.x; k < NBINS; k += BLK)
    if (s_cnt[k]) s_base[k] = atomicAdd(&bins[k], s_cnt[k]);  // the block's range inside the class
  __syncthreads();
  if (c >= 0) perm[s_base[c] + local] = i;
}

size_t temp_bytes(int n) { return (NBINS + (size_t)n) * sizeof(int); }  // class counts + one class per tile

// perm[k] = a tile of the k-th most expensive cost class (any order inside a class)
hipError_t order_by_cost(hipStream_t s, const unsigned* cost, int n, int* perm, void* temp, size_t temp_size) {
  if (temp_size < (NBINS + (size_t)n) * sizeof(int)) return hipErrorInvalidValue;
  int*       bins = (int*)temp;
  int*       cls  = bins + NBINS;
  hipError_t e    = hipMemsetAsync(bins, 0, NBINS * sizeof(int), s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_hist, dim3((n + BLK - 1) / BLK), dim3(BLK), 0, s, cost, n, bins, cls);
  hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, bins);
  hipLaunchKernelGGL(k_place, dim3((n + BLK - 1) / BLK), dim3(BLK), 0, s, cls, n, bins, perm);
  return hipGetLastError();
}

}  // namespace ytorder
