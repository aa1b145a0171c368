// tools/experiments/ioslow.c -- LD_PRELOAD shim for the stall hunt (profiles/r06_stall.txt): reports every ioctl / mmap / munmap /
// madvise / mprotect of the process that takes longer than IOSLOW_MS (default 3 ms), with CLOCK_MONOTONIC time (= Python's
// perf_counter) and a native backtrace, so that a 65 ms hipLaunchKernel can be pinned on the system call it sits in.
//   gcc -O2 -shared -fPIC tools/experiments/ioslow.c -o gpurun_in/ioslow.so -ldl
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static double thr_ms(void) { static double v = -1; if (v < 0) { const char* e = getenv("IOSLOW_MS"); v = e ? atof(e) : 3.0; } return v; }
static void report(const char* what, unsigned long a, unsigned long b, double t0, double dt) {
  fprintf(stderr, "[ioslow] t=%.6f %s a=0x%lx b=0x%lx  %.3f ms  tid %ld\n", t0, what, a, b, dt * 1e3, (long)syscall(SYS_gettid));
  void* bt[24];
  const int n = backtrace(bt, 24);
  backtrace_symbols_fd(bt, n, 2);
  fflush(stderr);
}

int ioctl(int fd, unsigned long req, ...) {
  static int (*real)(int, unsigned long, void*) = 0;
  if (!real) real = (int (*)(int, unsigned long, void*))dlsym(RTLD_NEXT, "ioctl");
  va_list ap; va_start(ap, req); void* arg = va_arg(ap, void*); va_end(ap);
  const double t0 = now_s();
  const int r = real(fd, req, arg);
  const double dt = now_s() - t0;
  if (dt * 1e3 > thr_ms()) report("ioctl(fd, req)", (unsigned long)fd, req, t0, dt);
  return r;
}
void* mmap(void* addr, size_t len, int prot, int flags, int fd, off_t off) {
  static void* (*real)(void*, size_t, int, int, int, off_t) = 0;
  if (!real) real = (void* (*)(void*, size_t, int, int, int, off_t))dlsym(RTLD_NEXT, "mmap");
  const double t0 = now_s();
  void* r = real(addr, len, prot, flags, fd, off);
  const double dt = now_s() - t0;
  if (dt * 1e3 > thr_ms()) report("mmap(len, flags)", (unsigned long)len, (unsigned long)flags, t0, dt);
  return r;
}
int munmap(void* addr, size_t len) {
  static int (*real)(void*, size_t) = 0;
  if (!real) real = (int (*)(void*, size_t))dlsym(RTLD_NEXT, "munmap");
  const double t0 = now_s();
  const int r = real(addr, len);
  const double dt = now_s() - t0;
  if (dt * 1e3 > thr_ms()) report("munmap(addr, len)", (unsigned long)addr, (unsigned long)len, t0, dt);
  return r;
}
int madvise(void* addr, size_t len, int adv) {
  static int (*real)(void*, size_t, int) = 0;
  if (!real) real = (int (*)(void*, size_t, int))dlsym(RTLD_NEXT, "madvise");
  const double t0 = now_s();
  const int r = real(addr, len, adv);
  const double dt = now_s() - t0;
  if (dt * 1e3 > thr_ms()) report("madvise(len, advice)", (unsigned long)len, (unsigned long)adv, t0, dt);
  return r;
}
int mprotect(void* addr, size_t len, int prot) {
  static int (*real)(void*, size_t, int) = 0;
  if (!real) real = (int (*)(void*, size_t, int))dlsym(RTLD_NEXT, "mprotect");
  const double t0 = now_s();
  const int r = real(addr, len, prot);
  const double dt = now_s() - t0;
  if (dt * 1e3 > thr_ms()) report("mprotect(len, prot)", (unsigned long)len, (unsigned long)prot, t0, dt);
  return r;
}
