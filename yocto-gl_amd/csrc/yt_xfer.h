// yt_xfer.h — host <-> device copies that never hand caller (pageable) memory to the DMA engines.
//
// Why (round 3, DESIGN.md §1 "first-process fault"): for a pageable source of more than a few KB
// hipMemcpy[Async] does not stage — ROCclr pins the CALLER's pages on the fly
// (`Locking to pool ... hostMem = <caller pointer>` in AMD_LOG_LEVEL=4, i.e.
// hsa_amd_memory_lock_to_pool: a userptr registration) and lets SDMA read them in place.  A userptr
// mapping lives at the mercy of the host kernel's MMU notifiers: when the kernel migrates, compacts
// or collapses those pages (NUMA balancing, khugepaged — busiest right after a box and a process
// come up, which is when round 2 saw it) the GPU mapping is torn down under the copy, and on this
// XNACK-less configuration the engine does not retry: the process dies with
// `Memory access fault by GPU ... on address <page-aligned HOST address>. Reason: Unknown`.
// Round 2's three faults were all in first uploads / first batches of a fresh box and carried a
// host address.  The library therefore moves every byte through pinned memory it owns
// (hipHostMalloc: driver-allocated GTT pages, never migrated, no userptr): a small ring of 4-MiB
// bounce buffers, CPU memcpy on one side, DMA on the other.  Memory the caller already pinned
// (ythip_scene_staging pools, hipHostMalloc / hipHostRegister of its own) is recognised and DMA'd
// from directly.  A side effect worth having: an upload no longer depends on the lifetime of the
// caller's buffer after the call returns.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstring>

namespace ytx {

struct Bounce {
  static constexpr size_t CHUNK = 4u << 20;
  static constexpr int    N     = 4;
  void*      buf[N]  = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev[N]   = {nullptr, nullptr, nullptr, nullptr};
  bool       busy[N] = {false, false, false, false};
  int        next    = 0;
  // small uploads share a chunk (one 48-byte copy used to cost a whole 4-MiB slot and, from the fifth call
  // on, an event wait): `cur` is the chunk being filled, `fill` its used bytes; its event is recorded when it is closed
  static constexpr size_t SMALL = 256u << 10;
  int         cur        = -1;
  size_t      fill       = 0;
  hipStream_t cur_stream = nullptr;

  hipError_t init() {
    if (buf[0]) return hipSuccess;
    for (int k = 0; k < N; k++) {
      hipError_t e = hipHostMalloc(&buf[k], CHUNK, hipHostMallocDefault);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
      if (e != hipSuccess) {
        destroy();
        return e;
      }
    }
    return hipSuccess;
  }
  void destroy() {
    for (int k = 0; k < N; k++) {
      if (ev[k]) (void)hipEventDestroy(ev[k]);
      if (buf[k]) (void)hipHostFree(buf[k]);
      ev[k] = nullptr, buf[k] = nullptr, busy[k] = false;
    }
    next = 0, cur = -1, fill = 0;
  }
  // the shared chunk of the small uploads is complete: from now on it is busy until its last copy has run
  hipError_t close_small() {
    if (cur < 0) return hipSuccess;
    hipError_t e = hipEventRecord(ev[cur], cur_stream);
    busy[cur]    = e == hipSuccess;
    cur = -1, fill = 0;
    return e;
  }
  // host memory the device can address as it is: pinned by the caller or by this library
  static bool pinned(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
      (void)hipGetLastError();  // (older runtimes report pageable memory as an error)
      return false;
    }
    return a.type == hipMemoryTypeHost || a.type == hipMemoryTypeManaged;
  }
  hipError_t acquire(int* k) {
    *k = next;
    if (busy[*k]) {
      hipError_t e = hipEventSynchronize(ev[*k]);
      if (e != hipSuccess) return e;
      busy[*k] = false;
    }
    next = (next + 1) % N;
    return hipSuccess;
  }

  // Stream-ordered on `s`.  On return `src` has been read completely unless it is pinned memory
  // (then it must stay valid until `s` has been synchronised, as with any asynchronous copy).
  hipError_t h2d(hipStream_t s, void* dst, const void* src, size_t bytes) {
    if (!bytes) return hipSuccess;
    if (bytes <= SMALL) {  // (no pointer query either: copying a few KB costs less than asking)
      hipError_t e = init();
      if (e != hipSuccess) return e;
      const size_t need = (bytes + 255) & ~(size_t)255;
      if (cur >= 0 && (fill + need > CHUNK || cur_stream != s) && (e = close_small()) != hipSuccess) return e;
      if (cur < 0) {
        if ((e = acquire(&cur)) != hipSuccess) return cur = -1, e;
        fill = 0, cur_stream = s;
      }
      std::memcpy((char*)buf[cur] + fill, src, bytes);
      e = hipMemcpyAsync(dst, (char*)buf[cur] + fill, bytes, hipMemcpyHostToDevice, s);
      fill += need;
      return e;
    }
    if (pinned(src)) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s);
    hipError_t e = init();
    if (e == hipSuccess) e = close_small();
    for (size_t off = 0; off < bytes && e == hipSuccess; off += CHUNK) {
      const size_t n = bytes - off < CHUNK ? bytes - off : CHUNK;
      int          k;
      if ((e = acquire(&k)) != hipSuccess) break;
      std::memcpy(buf[k], (const char*)src + off, n);
      e = hipMemcpyAsync((char*)dst + off, buf[k], n, hipMemcpyHostToDevice, s);
      if (e == hipSuccess) e = hipEventRecord(ev[k], s);
      busy[k] = e == hipSuccess;
    }
    return e;
  }

  // Complete on return (everything queued on `s` before it has finished too).
  hipError_t d2h(hipStream_t s, void* dst, const void* src, size_t bytes) {
    if (!bytes) return hipStreamSynchronize(s);
    if (pinned(dst)) {
      hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s);
      return e == hipSuccess ? hipStreamSynchronize(s) : e;
    }
    hipError_t e = init();
    if (e == hipSuccess) e = close_small();
    if (e != hipSuccess) return e;
    // the DMA of chunk c + 1 .. c + N - 1 runs while chunk c is copied out of its bounce buffer
    struct Pending {
      int    k;
      size_t off, n;
    } q[N];
    int    head = 0, count = 0;
    size_t off = 0;
    auto   retire = [&]() -> hipError_t {
      Pending&   p  = q[head];
      hipError_t e2 = hipEventSynchronize(ev[p.k]);
      if (e2 == hipSuccess) std::memcpy((char*)dst + p.off, buf[p.k], p.n);
      busy[p.k] = false;
      head      = (head + 1) % N, count--;
      return e2;
    };
    while (off < bytes && e == hipSuccess) {
      if (count == N && (e = retire()) != hipSuccess) break;
      const size_t n = bytes - off < CHUNK ? bytes - off : CHUNK;
      int          k;
      if ((e = acquire(&k)) != hipSuccess) break;
      e = hipMemcpyAsync(buf[k], (const char*)src + off, n, hipMemcpyDeviceToHost, s);
      if (e == hipSuccess) e = hipEventRecord(ev[k], s);
      if (e != hipSuccess) break;
      busy[k]                 = true;
      q[(head + count) % N]   = {k, off, n};
      count++;
      off += n;
    }
    while (count > 0) {
      hipError_t e2 = retire();
      if (e == hipSuccess) e = e2;
    }
    return e;
  }
};

}  // namespace ytx
