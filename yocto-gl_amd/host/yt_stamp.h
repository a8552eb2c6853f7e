//
// yt_stamp.h — content stamps of the host arrays the device mirrors (yocto_hiptrace.cpp's residency cache):
// a 64-bit hash of every byte of the large arrays, computed in 1-MiB pieces on a persistent pool of host
// threads, or — opt-in — the 256-element strided sample of rounds 1-3.  Plain C++17, no yocto headers
// (tests/cpp/stamp_check.cpp compiles it on its own).
//
#ifndef YT_STAMP_H
#define YT_STAMP_H

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace yocto::hip::stamp {

using hash_t = uint64_t;
inline hash_t fnv(const void* data, size_t bytes, hash_t h = 1469598103934665603ull) {
  auto p = (const unsigned char*)data;
  for (size_t k = 0; k < bytes; k++) h = (h ^ p[k]) * 1099511628211ull;
  return h;
}
template <typename T>
hash_t fnv_all(const std::vector<T>& v, hash_t h) {
  auto n = v.size();
  h      = fnv(&n, sizeof(n), h);
  return v.empty() ? h : fnv(v.data(), v.size() * sizeof(T), h);
}

// ---- the full-content hash ---------------------------------------------------------------
// one piece (<= 1 MiB): four independent multiply-rotate lanes over 32 bytes per step (the lanes keep the
// multiplier pipelines busy; a single chain would run at a third of the memory bandwidth), folded at the end
inline uint64_t mix64(uint64_t x) {
  x ^= x >> 32, x *= 0xd6e8feb86659fd93ull, x ^= x >> 32, x *= 0xd6e8feb86659fd93ull, x ^= x >> 32;
  return x;
}
inline uint64_t hash_piece(const unsigned char* p, size_t bytes) {
  uint64_t a = 0x9e3779b97f4a7c15ull ^ bytes, b = 0xbf58476d1ce4e5b9ull, c = 0x94d049bb133111ebull, d = 0x2545f4914f6cdd1dull;
  auto     rot = [](uint64_t x, int r) { return (x << r) | (x >> (64 - r)); };
  size_t   k   = 0;
  for (; k + 32 <= bytes; k += 32) {
    uint64_t w[4];
    std::memcpy(w, p + k, 32);
    a = rot(a ^ w[0], 29) * 0x9fb21c651e98df25ull;
    b = rot(b ^ w[1], 31) * 0xc2b2ae3d27d4eb4full;
    c = rot(c ^ w[2], 33) * 0x165667b19e3779f9ull;
    d = rot(d ^ w[3], 27) * 0x85ebca77c2b2ae63ull;
  }
  uint64_t tail[4] = {0, 0, 0, 0};
  std::memcpy(tail, p + k, bytes - k);
  a ^= tail[0], b ^= tail[1], c ^= tail[2], d ^= tail[3];
  return mix64(mix64(a) + rot(mix64(b), 17) + rot(mix64(c), 31) + rot(mix64(d), 47));
}

// A persistent pool of host threads for the hashing (the reference's own parallel_for spawns threads per call:
// ~100 us with 256 of them, which is what a whole 64-spp batch of configs[1] costs on the device).
class hash_pool {
 public:
  static hash_pool& get() {
    static hash_pool p;
    return p;
  }
  // runs fn(k) for k in [0, n) on the pool + the calling thread.  An exception thrown by fn (on any thread) is
  // rethrown here AFTER every worker has checked out of the job: the pool's bookkeeping stays balanced.
  template <typename F>
  void run(size_t n, F&& fn) {
    if (n == 0) return;
    if (n == 1 || threads.empty()) {
      for (size_t k = 0; k < n; k++) fn(k);
      return;
    }
    auto one_at_a_time = std::lock_guard{callers};  // (the shim's callers hold its residency lock anyway)
    std::function<void(size_t)> f = fn;
    {
      auto lock = std::unique_lock{m};
      job = &f, total = n, next.store(0), pending = threads.size(), generation++, failure = nullptr;
    }
    wake.notify_all();
    drain(f, n);
    auto lock = std::unique_lock{m};
    idle.wait(lock, [&] { return pending == 0; });
    job = nullptr;
    if (auto e = std::exchange(failure, nullptr)) std::rethrow_exception(e);
  }

 private:
  hash_pool() {
    unsigned n = std::thread::hardware_concurrency();
    n          = n > 1 ? std::min(n - 1, 31u) : 0;  // memory-bound work: more threads than channels buys nothing
    if (auto e = std::getenv("YOCTO_HIP_HASH_THREADS")) n = (unsigned)std::max(0, std::atoi(e));
    for (unsigned t = 0; t < n; t++) threads.emplace_back([this] { loop(); });
  }
  ~hash_pool() {
    {
      auto lock = std::unique_lock{m};
      quit      = true;
    }
    wake.notify_all();
    for (auto& t : threads) t.join();
  }
  void loop() {
    uint64_t seen = 0;
    while (true) {
      std::function<void(size_t)>* f;
      size_t                       n;
      {
        auto lock = std::unique_lock{m};
        wake.wait(lock, [&] { return quit || generation != seen; });
        if (quit) return;
        seen = generation, f = job, n = total;
      }
      drain(*f, n);
      auto lock = std::unique_lock{m};
      if (--pending == 0) idle.notify_one();
    }
  }
  // takes indices until none is left; the first exception ends the job for everybody (the rest of the indices is
  // skipped) and is kept for run() to rethrow
  void drain(std::function<void(size_t)>& f, size_t n) {
    try {
      for (size_t k; (k = next.fetch_add(1)) < n;) f(k);
    } catch (...) {
      next.store(n);
      auto lock = std::unique_lock{m};
      if (!failure) failure = std::current_exception();
    }
  }
  std::vector<std::thread>     threads;
  std::mutex                   m, callers;
  std::condition_variable      wake, idle;
  std::function<void(size_t)>* job = nullptr;
  size_t                       total = 0, pending = 0;
  std::atomic<size_t>          next{0};
  uint64_t                     generation = 0;
  bool                         quit       = false;
  std::exception_ptr           failure;
};

// how the large arrays are stamped: every byte (default) or the 256-element sample of rounds 1-3
inline std::atomic<int>& residency_mode() {
  static std::atomic<int> mode{[] {
    auto e = std::getenv("YOCTO_HIP_RESIDENCY");
    return e && std::string(e) == "sampled" ? 1 : 0;
  }()};
  return mode;
}

// Collects the large arrays of one object, then hashes them all in one go: `h` = hash of (size, address) of every
// array in order + (full mode) the 64-bit hashes of their 1-MiB pieces in order, or (sampled mode) 256 elements each.
struct array_hasher {
  struct span {
    const unsigned char* p;
    size_t               bytes, elem;
  };
  std::vector<span> spans;
  hash_t            h = 1469598103934665603ull;
  template <typename T>
  void add(const std::vector<T>& v) {
    auto n = v.size();
    auto a = (uintptr_t)v.data();
    h      = fnv(&n, sizeof(n), h);
    h      = fnv(&a, sizeof(a), h);
    if (n) spans.push_back({(const unsigned char*)v.data(), n * sizeof(T), sizeof(T)});
  }
  hash_t finish() {
    if (residency_mode().load() == 1) {
      for (auto& s : spans) {
        const size_t n = s.bytes / s.elem;
        if (n <= 256) {
          h = fnv(s.p, s.bytes, h);
          continue;
        }
        for (size_t k = 0; k < 256; k++) h = fnv(s.p + (size_t)((unsigned __int128)k * (n - 1) / 255) * s.elem, s.elem, h);
      }
      return h;
    }
    constexpr size_t PIECE = 1u << 20, INLINE_PIECES = 8;
    struct piece {
      const unsigned char* p;
      size_t               bytes;
    };
    std::vector<piece> pieces;
    for (auto& s : spans)
      for (size_t off = 0; off < s.bytes; off += PIECE) pieces.push_back({s.p + off, std::min(PIECE, s.bytes - off)});
    std::vector<uint64_t> out(pieces.size());
    // small scenes are hashed in place: waking up to 31 threads costs more than 8 MiB of hashing (~80 us)
    if (pieces.size() <= INLINE_PIECES)
      for (size_t k = 0; k < pieces.size(); k++) out[k] = hash_piece(pieces[k].p, pieces[k].bytes);
    else
      hash_pool::get().run(pieces.size(), [&](size_t k) { out[k] = hash_piece(pieces[k].p, pieces[k].bytes); });
    return out.empty() ? h : fnv(out.data(), out.size() * sizeof(uint64_t), h);
  }
};

}  // namespace yocto::hip::stamp

#endif
