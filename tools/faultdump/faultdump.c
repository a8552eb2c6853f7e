// faultdump.so — LD_PRELOAD diagnostic for "Memory access fault by GPU ..." (development tool,
// never shipped, never linked).
//
// ROCr reports a GPU page fault by printing one line with the faulting address and calling
// abort() from its event thread.  This interposer
//   * records every hipMalloc / hipHostMalloc / hipFree / hipHostFree / hipMemcpy[Async] the
//     process makes (pointer, size, host source, return address), and
//   * on SIGABRT writes that table plus /proc/self/maps to $FAULTDUMP_OUT (default
//     faultdump.<pid>.txt), so the address can be attributed to a device allocation, a host
//     buffer handed to a copy, a mapped file or nothing at all.
//
//   gcc -O2 -shared -fPIC -o faultdump.so faultdump.c -ldl
//   LD_PRELOAD=$PWD/faultdump.so FAULTDUMP_OUT=out.txt python ...
#define _GNU_SOURCE
#include <dlfcn.h>
#include <fcntl.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

typedef int   hipError_t;
typedef void* hipStream_t;

enum { K_MALLOC = 1, K_HOSTMALLOC, K_FREE, K_HOSTFREE, K_COPY, K_COPYASYNC, K_MEMSET, K_SYNC };
static const char* kind_name[] = {"?", "hipMalloc", "hipHostMalloc", "hipFree", "hipHostFree", "hipMemcpy",
    "hipMemcpyAsync", "hipMemset*", "sync"};

typedef struct {
  int      kind;
  int      rc;
  void*    a;  // allocation / destination
  void*    b;  // source
  size_t   n;
  void*    caller;
  uint64_t ns;
} rec_t;

#define MAXREC (1 << 16)
static rec_t         g_rec[MAXREC];
static volatile long g_n = 0;

static uint64_t now_ns(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

static void note(int kind, int rc, void* a, void* b, size_t n, void* caller) {
  long i = __sync_fetch_and_add(&g_n, 1);
  rec_t* r = &g_rec[i & (MAXREC - 1)];
  r->kind = kind, r->rc = rc, r->a = a, r->b = b, r->n = n, r->caller = caller, r->ns = now_ns();
}

static void* next(const char* name) {
  void* f = dlsym(RTLD_NEXT, name);
  if (!f) {  // the HIP runtime came in through a dlopen with RTLD_LOCAL (ctypes → libythip.so)
    void* h = dlopen("libamdhip64.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("libamdhip64.so.7", RTLD_NOW | RTLD_NOLOAD);
    if (h) f = dlsym(h, name);
  }
  if (!f) {
    fprintf(stderr, "faultdump: %s not found\n", name);
    _exit(111);
  }
  return f;
}

hipError_t hipMalloc(void** p, size_t n) {
  static hipError_t (*f)(void**, size_t);
  if (!f) f = next("hipMalloc");
  hipError_t rc = f(p, n);
  note(K_MALLOC, rc, p ? *p : 0, 0, n, __builtin_return_address(0));
  return rc;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags) {
  static hipError_t (*f)(void**, size_t, unsigned);
  if (!f) f = next("hipHostMalloc");
  hipError_t rc = f(p, n, flags);
  note(K_HOSTMALLOC, rc, p ? *p : 0, 0, n, __builtin_return_address(0));
  return rc;
}
hipError_t hipFree(void* p) {
  static hipError_t (*f)(void*);
  if (!f) f = next("hipFree");
  note(K_FREE, 0, p, 0, 0, __builtin_return_address(0));
  return f(p);
}
hipError_t hipHostFree(void* p) {
  static hipError_t (*f)(void*);
  if (!f) f = next("hipHostFree");
  note(K_HOSTFREE, 0, p, 0, 0, __builtin_return_address(0));
  return f(p);
}
hipError_t hipMemcpy(void* d, const void* s, size_t n, int kind) {
  static hipError_t (*f)(void*, const void*, size_t, int);
  if (!f) f = next("hipMemcpy");
  note(K_COPY, kind, d, (void*)s, n, __builtin_return_address(0));
  return f(d, s, n, kind);
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int kind, hipStream_t st) {
  static hipError_t (*f)(void*, const void*, size_t, int, hipStream_t);
  if (!f) f = next("hipMemcpyAsync");
  note(K_COPYASYNC, kind, d, (void*)s, n, __builtin_return_address(0));
  return f(d, s, n, kind, st);
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) {
  static hipError_t (*f)(void*, int, size_t, hipStream_t);
  if (!f) f = next("hipMemsetAsync");
  note(K_MEMSET, v, d, 0, n, __builtin_return_address(0));
  return f(d, v, n, st);
}
hipError_t hipMemset(void* d, int v, size_t n) {
  static hipError_t (*f)(void*, int, size_t);
  if (!f) f = next("hipMemset");
  note(K_MEMSET, v, d, 0, n, __builtin_return_address(0));
  return f(d, v, n);
}
hipError_t hipStreamSynchronize(hipStream_t st) {
  static hipError_t (*f)(hipStream_t);
  if (!f) f = next("hipStreamSynchronize");
  hipError_t rc = f(st);
  note(K_SYNC, rc, st, 0, 0, __builtin_return_address(0));
  return rc;
}

static void dump(int sig) {
  (void)sig;
  const char* out = getenv("FAULTDUMP_OUT");
  char        path[256];
  if (out)
    snprintf(path, sizeof(path), "%s", out);
  else
    snprintf(path, sizeof(path), "faultdump.%d.txt", (int)getpid());
  int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (fd < 0) return;
  char     line[512];
  long     n     = g_n;
  long     first = n > MAXREC ? n - MAXREC : 0;
  uint64_t t     = now_ns();
  int      len   = snprintf(line, sizeof(line), "# faultdump pid %d: %ld calls recorded, abort at t=0; times in ms before the abort\n",
      (int)getpid(), n);
  (void)!write(fd, line, len);
  for (long i = first; i < n; i++) {
    rec_t*  r = &g_rec[i & (MAXREC - 1)];
    Dl_info info;
    const char* so = "?";
    uintptr_t   off = 0;
    if (r->caller && dladdr(r->caller, &info) && info.dli_fname) {
      so  = strrchr(info.dli_fname, '/') ? strrchr(info.dli_fname, '/') + 1 : info.dli_fname;
      off = (uintptr_t)r->caller - (uintptr_t)info.dli_fbase;
    }
    len = snprintf(line, sizeof(line), "%-15s a=%p b=%p n=%zu arg=%d t=-%.3f from %s+0x%lx\n", kind_name[r->kind], r->a,
        r->b, r->n, r->rc, (double)(t - r->ns) * 1e-6, so, (unsigned long)off);
    (void)!write(fd, line, len);
  }
  (void)!write(fd, "# /proc/self/maps\n", 18);
  int m = open("/proc/self/maps", O_RDONLY);
  if (m >= 0) {
    char    buf[65536];
    ssize_t k;
    while ((k = read(m, buf, sizeof(buf))) > 0) (void)!write(fd, buf, (size_t)k);
    close(m);
  }
  close(fd);
}

__attribute__((constructor)) static void install(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_handler = dump;
  sa.sa_flags   = SA_RESETHAND;  // abort() re-raises with the default action afterwards
  sigaction(SIGABRT, &sa, 0);
}
