// stamp_check.cpp — the content stamp of yocto-gl_amd/host/yt_stamp.h on its own (TEST INFRASTRUCTURE, CPU):
//   * full mode: flipping ANY single bit of a large array changes the stamp (2,000 random bits + the first and the
//     last byte), a resize or a move to another address changes it, equal content at the same address does not;
//   * sampled mode (the opt-in of rounds 1-3): a bit outside the 256-element sample is NOT seen — which is why the
//     default changed;
//   * what the full hash costs on this host.
#include <chrono>
#include <cstdio>
#include <random>

#include "../../yocto-gl_amd/host/yt_stamp.h"

using namespace yocto::hip::stamp;

static hash_t stamp(const std::vector<float>& a, const std::vector<int>& b) {
  array_hasher h;
  h.add(a), h.add(b);
  return h.finish();
}

int main() {
  int               failures = 0;
  std::mt19937_64   rng(7);
  std::vector<float> a(3 * 1000 * 1000 + 5);  // 12 MB: eleven and a half pieces
  std::vector<int>   b(777);
  for (auto& x : a) x = (float)(rng() % 100000) * 1e-3f;
  for (auto& x : b) x = (int)rng();
  residency_mode().store(0);
  const hash_t base = stamp(a, b);
  if (stamp(a, b) != base) failures++, std::printf("FAIL: the stamp is not a function of the content\n");
  auto flip = [&](size_t byte, int bit) { ((unsigned char*)a.data())[byte] ^= (unsigned char)(1u << bit); };
  int  missed = 0;
  for (int k = 0; k < 2002; k++) {
    size_t byte = k == 0 ? 0 : k == 1 ? a.size() * 4 - 1 : rng() % (a.size() * 4);
    int    bit  = (int)(rng() % 8);
    flip(byte, bit);
    if (stamp(a, b) == base) missed++;
    flip(byte, bit);
  }
  if (missed) failures++, std::printf("FAIL: %d of 2002 single-bit edits not seen by the full stamp\n", missed);
  b[400] ^= 1;
  if (stamp(a, b) == base) failures++, std::printf("FAIL: edit of the small array not seen\n");
  b[400] ^= 1;
  {
    auto c = a;  // same content, other address: an upload is due (the mirrors are keyed by storage)
    if (stamp(c, b) == base) failures++, std::printf("FAIL: a copy at another address has the same stamp\n");
    c = a;
    c.pop_back();
    if (stamp(c, b) == stamp(a, b)) failures++, std::printf("FAIL: a resize is not seen\n");
  }
  // the sample of rounds 1-3 misses almost every single-element edit
  residency_mode().store(1);
  const hash_t sbase = stamp(a, b);
  int          seen  = 0;
  for (int k = 0; k < 500; k++) {
    size_t byte = rng() % (a.size() * 4);
    flip(byte, 3);
    if (stamp(a, b) != sbase) seen++;
    flip(byte, 3);
  }
  std::printf("sampled stamp: %d of 500 single-bit edits seen (256 of %zu elements are looked at)\n", seen, a.size());
  if (seen > 5) failures++, std::printf("FAIL: the sampled stamp sees more than it can\n");
  residency_mode().store(0);
  // a job that throws leaves the pool usable: the exception reaches the caller, the next job runs
  {
    bool caught = false;
    try {
      hash_pool::get().run(1000, [](size_t k) {
        if (k == 137) throw std::runtime_error("job 137");
      });
    } catch (const std::runtime_error&) {
      caught = true;
    }
    if (!caught && std::thread::hardware_concurrency() > 1) failures++, std::printf("FAIL: the job's exception was swallowed\n");
    std::atomic<size_t> sum{0};
    hash_pool::get().run(1000, [&](size_t k) { sum += k; });
    if (sum != 999 * 1000 / 2) failures++, std::printf("FAIL: the pool lost indices after a throwing job\n");
  }
  // cost
  std::vector<float> big(25 * 1000 * 1000);  // 100 MB
  for (size_t k = 0; k < big.size(); k += 1024) big[k] = (float)k;
  (void)stamp(big, b);
  auto   t0 = std::chrono::steady_clock::now();
  hash_t x  = 0;
  for (int k = 0; k < 10; k++) x ^= stamp(big, b);
  double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / 10;
  std::printf("full stamp of 100 MB: %.2f ms (%.1f GB/s, %u hardware threads) [%llx]\n", ms, 0.1 / (ms * 1e-3),
      std::thread::hardware_concurrency(), (unsigned long long)x);
  std::printf(failures ? "stamp_check: %d FAILURES\n" : "stamp_check: OK\n", failures);
  return failures ? 1 : 0;
}
