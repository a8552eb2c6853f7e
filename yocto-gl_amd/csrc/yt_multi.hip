// yt_multi.hip — one process, N GPUs: the multi-device half of include/ythip.h.
//
// SURVEY.md §8(b) asks for `ythip_create(device_ids[], n)`, §8(e) for pixel sharding
// with ONE exchange: the framebuffer gather over RCCL / xGMI.  A ythip_multi owns one
// ythip_ctx per device (full replica of scene + BVH + lights — the caller uploads to
// each, ythip_multi_ctx()), shards trace_state by 16-pixel tile columns dealt round-robin
// (rank r of n owns columns r, r + n, ...: every rank sees sky and ground alike,
// DESIGN.md §7), launches a batch on every device from one host thread
// (k_trace / k_pool are asynchronous launches on per-device streams) and gathers the
// `image` slices to device 0 with ncclSend / ncclRecv inside one group, followed by an
// un-permute kernel.  No data-path collective: pixels are independent
// (yocto_trace.cpp:1600-1612).
//
// RCCL is loaded at run time (dlopen), only when a gather between DISTINCT devices is
// asked for; a multi whose ranks share a device (a rehearsal of the sharding on one GPU —
// RCCL refuses duplicate devices in a communicator) gathers with device copies.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "yt_xfer.h"

#include "../../include/ythip.h"

namespace {

// the few RCCL entry points used, resolved from librccl at run time
typedef struct ncclComm* ncclComm_t;
typedef int              ncclResult_t;
enum { ncclFloat32 = 7 };
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*)                                    = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t)                                                      = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*)                                            = nullptr;
  ncclResult_t (*GroupStart)()                                                                 = nullptr;
  ncclResult_t (*GroupEnd)()                                                                   = nullptr;
  ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t)                 = nullptr;
  ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t)                       = nullptr;
  const char* (*GetErrorString)(ncclResult_t)                                                  = nullptr;
  bool load(std::string* err) {
    if (lib) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) {
      *err = std::string("librccl not found: ") + (dlerror() ? dlerror() : "");
      return false;
    }
#define SYM(field, name)                                        \
  field = (decltype(field))dlsym(lib, name);                    \
  if (!field) {                                                 \
    *err = std::string("librccl lacks ") + name;                \
    return false;                                               \
  }
    SYM(CommInitAll, "ncclCommInitAll");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(CommCount, "ncclCommCount");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    return true;
  }
};

constexpr int TILE = 16;  // tile-column width of the striping (yt_kernels.h YT_TILE)

// frame pixel (i, j) <- slice of rank (i / 16) % n, local column ((i / 16) / n) * 16 + i % 16
__global__ void __launch_bounds__(256) k_unstripe4(const float4* gathered, const long long* slice_offset,
    const int* slice_lwidth, int n, int width, int height, float4* frame) {
  long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)width * height) return;
  int j = (int)(idx / width), i = (int)(idx - (long long)j * width);
  int tc = i / TILE, r = tc % n, lc = (tc / n) * TILE + (i - tc * TILE);
  frame[idx] = gathered[slice_offset[r] + (long long)j * slice_lwidth[r] + lc];
}

}  // namespace

struct ythip_multi {
  int                     n = 0;
  std::vector<int>        devices;
  std::vector<ythip_ctx*> ctx;
  std::vector<hipStream_t> streams;   // per rank: the gather's stream on that rank's device
  std::string             err;
  bool                    distinct = true;
  int                     width = 0, height = 0;
  std::vector<int>        lwidth;
  std::vector<long long>  offset;  // float4 offset of each slice in the gathered buffer
  // gather
  Rccl                    rccl;
  std::vector<ncclComm_t> comms;
  int                     comm_ranks = 0;   // what RCCL reports (ncclCommCount)
  float4*                 d_gathered = nullptr;  // on devices[0]
  float4*                 d_frame    = nullptr;
  long long*              d_offset   = nullptr;
  int*                    d_lwidth   = nullptr;
  std::string             gather_mode = "none";
  ytx::Bounce             xfer;  // the frame goes to the caller through pinned memory (yt_xfer.h), on devices[0]
};

namespace {
thread_local std::string g_merr;
int mfail(ythip_multi* m, int code, const char* fmt, ...) {
  char    buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (m) m->err = buf;
  g_merr = buf;
  return code;
}
#define MHIP(m, call)                                                                                     \
  do {                                                                                                    \
    hipError_t e_ = (call);                                                                               \
    if (e_ != hipSuccess) return mfail(m, YTHIP_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));  \
  } while (0)
// error of a per-rank call, with the rank named
int rank_fail(ythip_multi* m, int r, int rc) {
  return mfail(m, rc, "rank %d (device %d): %s", r, m->devices[r], ythip_last_error(m->ctx[r]));
}
void free_gather(ythip_multi* m) {
  if (m->n == 0) return;
  (void)hipSetDevice(m->devices[0]);
  for (void* p : {(void*)m->d_gathered, (void*)m->d_frame, (void*)m->d_offset, (void*)m->d_lwidth})
    if (p) (void)hipFree(p);
  m->d_gathered = m->d_frame = nullptr;
  m->d_offset   = nullptr;
  m->d_lwidth   = nullptr;
}
}  // namespace

// full-frame host arrays <-> the ranks' slices (strides in 4-byte words per pixel)
namespace {
template <typename F>
void for_each_run(const ythip_multi* m, int r, F f) {  // f(frame pixel index, slice pixel index, run length)
  const int W = m->width, H = m->height, n = m->n, lw = m->lwidth[r];
  for (int j = 0; j < H; j++) {
    int lc = 0;
    for (int tc = r; tc * TILE < W; tc += n) {
      int len = std::min(TILE, W - tc * TILE);
      f((size_t)j * W + (size_t)tc * TILE, (size_t)j * lw + lc, len);
      lc += len;
    }
  }
}
}  // namespace

extern "C" {

int ythip_create_multi(const int* device_ids, int n, ythip_multi** out) {
  if (!out) return mfail(nullptr, YTHIP_ERR_INVALID, "out is null");
  *out = nullptr;
  if (!device_ids || n < 1 || n > 64) return mfail(nullptr, YTHIP_ERR_INVALID, "need 1..64 device ids");
  auto m = new ythip_multi{};
  m->n   = n;
  m->devices.assign(device_ids, device_ids + n);
  for (int a = 0; a < n; a++)
    for (int b = a + 1; b < n; b++)
      if (device_ids[a] == device_ids[b]) m->distinct = false;
  for (int r = 0; r < n; r++) {
    ythip_ctx* c  = nullptr;
    int        rc = ythip_create(device_ids[r], &c);
    if (rc) {
      std::string e = ythip_last_error(nullptr);
      for (auto p : m->ctx) ythip_destroy(p);
      delete m;
      return mfail(nullptr, rc, "rank %d: %s", r, e.c_str());
    }
    m->ctx.push_back(c);
    hipStream_t s = nullptr;
    if (hipSetDevice(device_ids[r]) != hipSuccess || hipStreamCreate(&s) != hipSuccess) {
      for (auto p : m->ctx) ythip_destroy(p);
      delete m;
      return mfail(nullptr, YTHIP_ERR_HIP, "rank %d: stream creation failed", r);
    }
    m->streams.push_back(s);
  }
  *out = m;
  return YTHIP_OK;
}

void ythip_destroy_multi(ythip_multi* m) {
  if (!m) return;
  free_gather(m);
  if (!m->devices.empty()) {
    (void)hipSetDevice(m->devices[0]);
    m->xfer.destroy();
  }
  for (auto c : m->comms)
    if (c && m->rccl.CommDestroy) m->rccl.CommDestroy(c);
  for (int r = 0; r < m->n; r++) {
    (void)hipSetDevice(m->devices[r]);
    if (m->streams[r]) (void)hipStreamDestroy(m->streams[r]);
    ythip_destroy(m->ctx[r]);
  }
  delete m;
}

int         ythip_multi_size(const ythip_multi* m) { return m ? m->n : 0; }
ythip_ctx*  ythip_multi_ctx(ythip_multi* m, int rank) { return (m && rank >= 0 && rank < m->n) ? m->ctx[rank] : nullptr; }
const char* ythip_multi_last_error(const ythip_multi* m) { return m ? m->err.c_str() : g_merr.c_str(); }

int ythip_multi_state_create(ythip_multi* m, int width, int height) {
  if (!m) return YTHIP_ERR_INVALID;
  if (width <= 0 || height <= 0) return mfail(m, YTHIP_ERR_INVALID, "bad state geometry %dx%d", width, height);
  // (a frame with fewer tile columns than ranks leaves the last ranks without pixels: they
  // hold no state and sit the batches out)
  m->lwidth.assign(m->n, 0);
  m->offset.assign(m->n, 0);
  long long off = 0;
  for (int r = 0; r < m->n; r++) {
    m->lwidth[r] = std::max(0, ythip_state_local_width(width, r, m->n));
    m->offset[r] = off;
    off += (long long)m->lwidth[r] * height;
    if (m->lwidth[r] == 0) continue;
    int rc = ythip_state_create_striped(m->ctx[r], width, height, 0, height, r, m->n);
    if (rc) return rank_fail(m, r, rc);
  }
  m->width = width, m->height = height;
  free_gather(m);  // sized per frame
  return YTHIP_OK;
}

int ythip_multi_state_upload(ythip_multi* m, const float* image, const float* albedo, const float* normal,
    const int32_t* hits, const uint64_t* rngs, int samples) {
  if (!m || m->width == 0) return mfail(m, YTHIP_ERR_STATE, "multi_state_create first");
  for (int r = 0; r < m->n; r++) {
    if (m->lwidth[r] == 0) continue;
    size_t np = (size_t)m->lwidth[r] * m->height;
    std::vector<float>    im(image ? np * 4 : 0), al(albedo ? np * 3 : 0), no(normal ? np * 3 : 0);
    std::vector<int32_t>  hi(hits ? np : 0);
    std::vector<uint64_t> rn(rngs ? np * 2 : 0);
    for_each_run(m, r, [&](size_t fp, size_t sp, int len) {
      if (image) std::memcpy(&im[sp * 4], image + fp * 4, (size_t)len * 16);
      if (albedo) std::memcpy(&al[sp * 3], albedo + fp * 3, (size_t)len * 12);
      if (normal) std::memcpy(&no[sp * 3], normal + fp * 3, (size_t)len * 12);
      if (hits) std::memcpy(&hi[sp], hits + fp, (size_t)len * 4);
      if (rngs) std::memcpy(&rn[sp * 2], rngs + fp * 2, (size_t)len * 16);
    });
    int rc = ythip_state_upload(m->ctx[r], image ? im.data() : nullptr, albedo ? al.data() : nullptr,
        normal ? no.data() : nullptr, hits ? hi.data() : nullptr, rngs ? rn.data() : nullptr, samples);
    if (rc) return rank_fail(m, r, rc);
  }
  return YTHIP_OK;
}

int ythip_multi_state_download(ythip_multi* m, float* image, float* albedo, float* normal, int32_t* hits,
    uint64_t* rngs, int* samples) {
  if (!m || m->width == 0) return mfail(m, YTHIP_ERR_STATE, "multi_state_create first");
  for (int r = 0; r < m->n; r++) {
    if (m->lwidth[r] == 0) continue;
    size_t np = (size_t)m->lwidth[r] * m->height;
    std::vector<float>    im(image ? np * 4 : 0), al(albedo ? np * 3 : 0), no(normal ? np * 3 : 0);
    std::vector<int32_t>  hi(hits ? np : 0);
    std::vector<uint64_t> rn(rngs ? np * 2 : 0);
    int                   s  = 0;
    int rc = ythip_state_download(m->ctx[r], image ? im.data() : nullptr, albedo ? al.data() : nullptr,
        normal ? no.data() : nullptr, hits ? hi.data() : nullptr, rngs ? rn.data() : nullptr, &s);
    if (rc) return rank_fail(m, r, rc);
    if (samples) *samples = s;
    for_each_run(m, r, [&](size_t fp, size_t sp, int len) {
      if (image) std::memcpy(image + fp * 4, &im[sp * 4], (size_t)len * 16);
      if (albedo) std::memcpy(albedo + fp * 3, &al[sp * 3], (size_t)len * 12);
      if (normal) std::memcpy(normal + fp * 3, &no[sp * 3], (size_t)len * 12);
      if (hits) std::memcpy(hits + fp, &hi[sp], (size_t)len * 4);
      if (rngs) std::memcpy(rngs + fp * 2, &rn[sp * 2], (size_t)len * 16);
    });
  }
  return YTHIP_OK;
}

// trace_samples on every device at once: enqueue everywhere, then wait everywhere; `stop`
// is relayed to every rank (ythip_cancel).  state.samples advances on all ranks or none.
int ythip_multi_trace_samples(ythip_multi* m, const ythip_params* params, const volatile int32_t* stop) {
  if (!m || !params) return mfail(m, YTHIP_ERR_INVALID, "null argument");
  if (stop && *stop) return mfail(m, YTHIP_ERR_CANCELLED, "cancelled");
  if (m->width == 0) return mfail(m, YTHIP_ERR_STATE, "multi_state_create first");
  std::vector<int> samples_before(m->n, 0);
  for (int r = 0; r < m->n; r++)
    if (m->lwidth[r]) (void)ythip_state_get_samples(m->ctx[r], &samples_before[r]);
  bool cancelled = false;
  // A rank on the streaming scheduler (ythip_set_scheduler 1 / 2) runs its batch as generations enqueued by a host loop that
  // returns when the batch is done: such ranks launch from a thread each — or they would render one after the other, and nobody
  // would relay `stop` meanwhile.
  bool streaming = false;
  for (int r = 0; r < m->n; r++) streaming = streaming || (m->lwidth[r] && ythip_may_stream(m->ctx[r], params));
  if (streaming) {  // (one rank too: somebody has to watch `stop` while the rank's host loop runs)
    std::vector<int>         rcs(m->n, 0);
    std::vector<std::thread> threads;
    std::atomic<int>         left{0};
    for (int r = 0; r < m->n; r++) {
      if (m->lwidth[r] == 0) continue;
      left++;
      threads.emplace_back([&, r] {
        rcs[r] = ythip_trace_samples(m->ctx[r], params, nullptr);  // (the waiting call: the one the measured choice may stream)
        left--;
      });
    }
    while (left.load() > 0) {
      if (stop && *stop && !cancelled) {
        for (int q = 0; q < m->n; q++) (void)ythip_cancel(m->ctx[q]);
        cancelled = true;
      }
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    for (auto& t : threads) t.join();
    for (int r = 0; r < m->n; r++) {
      if (!rcs[r] || (cancelled && rcs[r] == YTHIP_ERR_CANCELLED)) continue;
      for (int q = 0; q < m->n; q++) {  // all ranks or none: the ranks that did run take their batch back
        if (q == r || !m->lwidth[q]) continue;
        (void)ythip_sync(m->ctx[q]);
        (void)ythip_state_set_samples(m->ctx[q], samples_before[q]);
      }
      return rank_fail(m, r, rcs[r]);
    }
  } else {
    for (int r = 0; r < m->n; r++) {
      if (m->lwidth[r] == 0) continue;
      int rc = ythip_trace_samples_async(m->ctx[r], params);
      if (rc) {  // all ranks or none: the ranks that did launch finish their batch and take it back
        for (int q = 0; q < r; q++) {
          (void)ythip_sync(m->ctx[q]);
          if (m->lwidth[q]) (void)ythip_state_set_samples(m->ctx[q], samples_before[q]);
        }
        return rank_fail(m, r, rc);
      }
    }
  }
  if (stop) {
    for (int r = 0; r < m->n && !cancelled; r++)
      while (ythip_poll(m->ctx[r]) == 0) {
        if (*stop) {
          for (int q = 0; q < m->n; q++) (void)ythip_cancel(m->ctx[q]);
          cancelled = true;
          break;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
      }
  }
  for (int r = 0; r < m->n; r++) {
    int rc = ythip_sync(m->ctx[r]);
    if (rc) return rank_fail(m, r, rc);
  }
  if (cancelled) {
    for (int r = 0; r < m->n; r++)
      if (m->lwidth[r]) (void)ythip_state_set_samples(m->ctx[r], samples_before[r]);
    return mfail(m, YTHIP_ERR_CANCELLED, "cancelled");
  }
  return YTHIP_OK;
}

// The framebuffer gather: every rank's `image` slice to device 0 — RCCL
// send/recv in one group between distinct devices, device copies otherwise —, the
// un-permute into frame order there, one download.
int ythip_multi_get_image(ythip_multi* m, float* image) {
  if (!m || !image) return mfail(m, YTHIP_ERR_INVALID, "null argument");
  if (m->width == 0) return mfail(m, YTHIP_ERR_STATE, "multi_state_create first");
  const int       n    = m->n;
  const long long npix = (long long)m->width * m->height;
  MHIP(m, hipSetDevice(m->devices[0]));
  if (!m->d_gathered) {
    MHIP(m, hipMalloc((void**)&m->d_gathered, (size_t)npix * sizeof(float4)));
    MHIP(m, hipMalloc((void**)&m->d_frame, (size_t)npix * sizeof(float4)));
    MHIP(m, hipMalloc((void**)&m->d_offset, (size_t)n * sizeof(long long)));
    MHIP(m, hipMalloc((void**)&m->d_lwidth, (size_t)n * sizeof(int)));
    MHIP(m, hipMemcpy(m->d_offset, m->offset.data(), (size_t)n * sizeof(long long), hipMemcpyHostToDevice));
    MHIP(m, hipMemcpy(m->d_lwidth, m->lwidth.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice));
  }
  std::vector<const void*> src(n, nullptr);
  for (int r = 0; r < n; r++) {
    if (m->lwidth[r] == 0) continue;
    void* p  = nullptr;
    int   rc = ythip_state_device_image(m->ctx[r], &p);  // also drains that rank's kernel stream
    if (rc) return rank_fail(m, r, rc);
    src[r] = p;
  }
  // YTHIP_GATHER=copy: device copies even between distinct devices.  YTHIP_GATHER=rccl-self: the RCCL
  // branch on whatever distinct devices there are — with ONE device a one-rank communicator whose
  // rank sends its slice to itself (a self send / recv inside the group), which is how a 1-GPU box
  // executes this branch's ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd sequence at all
  // (RCCL refuses a communicator that lists a device twice): tests/test_gpu_round2.py.
  const char* force = std::getenv("YTHIP_GATHER");
  const bool  self  = force && std::string(force) == "rccl-self" && m->distinct;
  const bool  use_rccl = self || (m->distinct && n > 1 && !(force && std::string(force) == "copy"));
  if (use_rccl) {
    if (m->comms.empty()) {
      std::string e;
      if (!m->rccl.load(&e)) return mfail(m, YTHIP_ERR_HIP, "%s", e.c_str());
      m->comms.assign(n, nullptr);
      ncclResult_t rc = m->rccl.CommInitAll(m->comms.data(), n, m->devices.data());
      if (rc) return mfail(m, YTHIP_ERR_HIP, "ncclCommInitAll: %s", m->rccl.GetErrorString(rc));
      m->rccl.CommCount(m->comms[0], &m->comm_ranks);
    }
    // rank 0's own slice never leaves its device (a device copy; in the self-test mode it takes
    // the send / recv path like everybody else's)
    if (!self)
      MHIP(m, hipMemcpyAsync(m->d_gathered + m->offset[0], src[0], (size_t)m->lwidth[0] * m->height * sizeof(float4),
                  hipMemcpyDeviceToDevice, m->streams[0]));
    ncclResult_t rc = m->rccl.GroupStart();
    for (int r = self ? 0 : 1; r < n && !rc; r++) {
      if (m->lwidth[r] == 0) continue;
      size_t count = (size_t)m->lwidth[r] * m->height * 4;
      rc           = m->rccl.Send(src[r], count, ncclFloat32, 0, m->comms[r], m->streams[r]);
      if (!rc) rc = m->rccl.Recv(m->d_gathered + m->offset[r], count, ncclFloat32, r, m->comms[0], m->streams[0]);
    }
    ncclResult_t rc2 = m->rccl.GroupEnd();
    if (rc || rc2) return mfail(m, YTHIP_ERR_HIP, "rccl gather: %s", m->rccl.GetErrorString(rc ? rc : rc2));
    for (int r = 1; r < n; r++) {
      MHIP(m, hipSetDevice(m->devices[r]));
      MHIP(m, hipStreamSynchronize(m->streams[r]));
    }
    MHIP(m, hipSetDevice(m->devices[0]));
    m->gather_mode = self ? "rccl send/recv (incl. rank 0 to itself)" : "rccl send/recv";
  } else {
    for (int r = 0; r < n; r++) {
      if (m->lwidth[r] == 0) continue;
      size_t bytes = (size_t)m->lwidth[r] * m->height * sizeof(float4);
      if (m->devices[r] == m->devices[0])
        MHIP(m, hipMemcpyAsync(m->d_gathered + m->offset[r], src[r], bytes, hipMemcpyDeviceToDevice, m->streams[0]));
      else
        MHIP(m, hipMemcpyPeerAsync(m->d_gathered + m->offset[r], m->devices[0], src[r], m->devices[r], bytes, m->streams[0]));
    }
    m->gather_mode = n > 1 ? "device copies" : "none (one rank)";
  }
  hipLaunchKernelGGL(k_unstripe4, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, m->streams[0], m->d_gathered,
      m->d_offset, m->d_lwidth, n, m->width, m->height, m->d_frame);
  MHIP(m, m->xfer.d2h(m->streams[0], image, m->d_frame, (size_t)npix * sizeof(float4)));  // (complete on return)
  return YTHIP_OK;
}

// how the last gather moved the slices + the rank count RCCL's communicator reports
int ythip_multi_gather_info(const ythip_multi* m, char* mode, int mode_len, int* rccl_ranks) {
  if (!m) return YTHIP_ERR_INVALID;
  if (mode && mode_len > 0) std::snprintf(mode, (size_t)mode_len, "%s", m->gather_mode.c_str());
  if (rccl_ranks) *rccl_ranks = m->comm_ranks;
  return YTHIP_OK;
}

}  // extern "C"
