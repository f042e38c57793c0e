// Common host/device helpers for the MI355X POGS engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace pogs_amd {

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

#define POGS_HIP_CHECK(expr)                                                      \
  do {                                                                            \
    hipError_t _e = (expr);                                                       \
    if (_e != hipSuccess) {                                                       \
      char _buf[512];                                                             \
      std::snprintf(_buf, sizeof(_buf), "HIP error %d (%s) at %s:%d: %s", (int)_e, \
                    hipGetErrorString(_e), __FILE__, __LINE__, #expr);            \
      throw ::pogs_amd::Error(_buf);                                              \
    }                                                                             \
  } while (0)

#define POGS_CHECK(cond, msg)                                                     \
  do {                                                                            \
    if (!(cond)) {                                                                \
      char _buf[512];                                                             \
      std::snprintf(_buf, sizeof(_buf), "%s (%s) at %s:%d", msg, #cond, __FILE__, \
                    __LINE__);                                                    \
      throw ::pogs_amd::Error(_buf);                                              \
    }                                                                             \
  } while (0)

// RAII device buffer.
template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  explicit DevBuf(size_t count) { alloc(count); }
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  DevBuf &operator=(DevBuf &&o) noexcept {
    if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void alloc(size_t count) {
    release();
    n = count;
    if (count) POGS_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T)));
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void zero(hipStream_t s) {
    if (n) POGS_HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(T), s));
  }
  T *get() const { return p; }
};

// RAII pinned host buffer.
template <typename T>
struct PinnedBuf {
  T *p = nullptr;
  size_t n = 0;
  PinnedBuf() = default;
  explicit PinnedBuf(size_t count) { alloc(count); }
  PinnedBuf(const PinnedBuf &) = delete;
  PinnedBuf &operator=(const PinnedBuf &) = delete;
  ~PinnedBuf() { if (p) (void)hipHostFree(p); }
  void alloc(size_t count) {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    n = count;
    if (count) POGS_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&p), count * sizeof(T), hipHostMallocMapped | hipHostMallocCoherent));
  }
  T *get() const { return p; }
};

inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int ceil_div(long long a, long long b) { return static_cast<int>((a + b - 1) / b); }

// 16-byte vector type per scalar type: float -> float4, double -> double2.
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  using type = float4;
  static constexpr int N = 4;
};
template <> struct Vec16<double> {
  using type = double2;
  static constexpr int N = 2;
};

// gfx950 geometry (MI355X): 256 CUs in 8 XCDs, 64-lane wavefronts.
constexpr int kWave = 64;
constexpr int kNumXcd = 8;

struct DeviceInfo {
  int device = 0;
  int num_cu = 256;
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) holds for the CURRENT device only.  One table per
// kernel instantiation (a function-local static of the caller) remembers what each device has been
// granted, so a process that drives several GPUs raises the limit on each of them.
constexpr int kMaxDevices = 64;
struct SmemGrants {
  std::atomic<size_t> bytes[kMaxDevices];   // zero-initialised as a static: nothing above the 48 KB default yet
};
inline void ensure_dynamic_smem(const void *fn, size_t want, SmemGrants &g) {
  if (want <= 48 * 1024) return;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  std::atomic<size_t> &slot = g.bytes[dev & (kMaxDevices - 1)];
  if (want <= slot.load(std::memory_order_acquire)) return;
  POGS_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(want)));
  size_t cur = slot.load(std::memory_order_relaxed);
  while (cur < want && !slot.compare_exchange_weak(cur, want)) {}
}

}  // namespace pogs_amd
