// yt_order.h — longest-tile-first launch order (yt_order.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace ytorder {
size_t     temp_bytes(int n);
hipError_t order_by_cost(hipStream_t s, const unsigned* cost, int n, int* perm, void* temp, size_t temp_size);
}  // namespace ytorder
