// Host-side plumbing shared by every translation unit of libsonic_hip.so: status codes, the
// thread-local error message, device buffers, and the launch wrapper that optionally brackets a
// kernel with HIP events (bench.py's live roofline numbers come from these).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <map>
#include <mutex>
#include <thread>
#include <vector>
#include <string>
#include <vector>
#include "../../include/sonic_hip.h"

namespace sonic {

void set_error(const char* fmt, ...);

struct HipFail { int code; };

#define HIP_OK(expr)                                                                          \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      ::sonic::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      throw ::sonic::HipFail{SONIC_ERR_HIP};                                                  \
    }                                                                                         \
  } while (0)

// Per-kernel-name event timing (enabled by sonic_profile_enable).
struct Profiler {
  struct Rec { hipEvent_t a, b; };
  bool on = false;
  std::mutex mu;
  std::map<std::string, std::vector<Rec>> recs;
  std::map<std::string, std::pair<double, long>> totals;  // ms, launches
  void collect();  // waits for recorded events and folds them into totals
  void reset();
};
Profiler& profiler();

struct ScopedKernelTimer {
  const char* name; hipStream_t st; Profiler::Rec r; bool on;
  ScopedKernelTimer(const char* n, hipStream_t s) : name(n), st(s), on(profiler().on) {
    if (on) { hipEventCreate(&r.a); hipEventCreate(&r.b); hipEventRecord(r.a, st); }
  }
  ~ScopedKernelTimer() {
    if (on) { hipEventRecord(r.b, st); std::lock_guard<std::mutex> g(profiler().mu); profiler().recs[name].push_back(r); }
  }
};

// LAUNCH(kernel, grid, block, lds_bytes, stream, args...)
#define LAUNCH(kern, grid, block, lds, stream, ...)                                           \
  do {                                                                                        \
    ::sonic::ScopedKernelTimer _t(#kern, stream);                                             \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, stream, __VA_ARGS__);              \
    HIP_OK(hipGetLastError());                                                                \
  } while (0)

// Host threads that are joined when the group goes out of scope -- also when the code between their start and their join throws (ADVICE
// r05: a std::thread destroyed while joinable ends the process in std::terminate).  A drop-in for std::vector<std::thread>.
struct ThreadGroup {
  std::vector<std::thread> th;
  template <class... A> void emplace_back(A&&... a) { th.emplace_back(std::forward<A>(a)...); }
  void join() { for (auto& t : th) if (t.joinable()) t.join(); }
  std::vector<std::thread>::iterator begin() { return th.begin(); }
  std::vector<std::thread>::iterator end() { return th.end(); }
  ThreadGroup() {}
  ThreadGroup(const ThreadGroup&) = delete;
  ThreadGroup& operator=(const ThreadGroup&) = delete;
  ~ThreadGroup() { join(); }
};

// RAII device buffer
struct DevBuf {
  void* p = nullptr; size_t bytes = 0;
  DevBuf() {}
  explicit DevBuf(size_t n) { alloc(n); }
  DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; } return *this; }
  ~DevBuf() { release(); }
  void alloc(size_t n) { release(); if (n == 0) n = 16; HIP_OK(hipMalloc(&p, n)); bytes = n; }
  void ensure(size_t n) { if (n > bytes) alloc(n); }
  void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// the library must fail loudly when there is no GPU: every entry point goes through a DeviceScope (internal.hpp)

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace sonic
