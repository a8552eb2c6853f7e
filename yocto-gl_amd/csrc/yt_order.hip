// yt_order.hip — longest-tile-first launch order for k_trace.
//
// k_trace gives every 16x4 tile its own one-wave workgroup; the hardware starts workgroups in
// blockIdx order as slots free up, so the end of a launch is a tail in which the last few
// (possibly expensive) tiles run alone.  Tile costs repeat from batch to batch (the same
// pixels), so every workgroup leaves its cycle count behind and the NEXT launch hands out the
// tiles most expensive first — the classic LPT rule; the tail then consists of the cheapest
// tiles.  Only the tile -> workgroup assignment changes: results are bit-identical.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "yt_order.h"

namespace ytorder {

__global__ void k_iota(int* v, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) v[i] = i;
}

size_t temp_bytes(int n) {
  size_t bytes = 0;
  (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, bytes, (const unsigned*)nullptr, (unsigned*)nullptr,
      (const int*)nullptr, (int*)nullptr, n);
  return bytes;
}

// perm[k] = tile with the k-th largest cost (ties: unspecified, any order is a valid assignment)
hipError_t order_by_cost(hipStream_t s, const unsigned* cost, int n, unsigned* keys_out, int* iota, int* perm,
    void* temp, size_t temp_size) {
  hipLaunchKernelGGL(k_iota, dim3((n + 255) / 256), dim3(256), 0, s, iota, n);
  return hipcub::DeviceRadixSort::SortPairsDescending(temp, temp_size, cost, keys_out, iota, perm, n, 0, 32, s);
}

}  // namespace ytorder
