// Common host/device helpers for the MI355X POGS engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <unordered_map>
#include <cstdlib>

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

// ---- device memory pool --------------------------------------------------------------------
// The reference constructs and destroys its solver inside every PogsD/PogsS call
// (src/interface_c/pogs_c.cpp:19-20), so a caller of the one-shot ABI -- and of
// PogsAmdCreate/Destroy in a loop -- allocates and frees the whole working set (C2: 5.6 GB) per
// call.  On this runtime a fresh hipMalloc of that size costs tens of milliseconds, its first
// touch 100-180 ms more (page tables are populated lazily), and hipFree unmaps synchronously: the
// round-3 driver run saw 0.28 s of such stalls in the second handle of a process.  Blocks
// therefore go back to a per-device cache instead of the runtime and the next handle takes them
// from there: no map / unmap, no first touch.
//
//  * size classes: requests are rounded up (512 B to 4 KB, 4 KB to 1 MB, 2 MB above) and a cached
//    block is taken when it is at most 1/8 larger than the rounded request;
//  * ordering: a block may be released while work that uses it is still in flight (hipFree would
//    have waited for the device).  Releases are numbered; an acquire that picks a block released
//    after the last device-wide wait began waits for the device once, which covers every block
//    released up to then;
//  * bound: at most POGS_AMD_POOL_MB of idle memory per device (default 1/4 of the device's
//    memory; 0 turns the pool off), largest-first eviction; an allocation that fails is retried
//    after the cache has been emptied; PogsAmdPoolTrim() empties it on request.
struct PoolCounters {
  unsigned long long mallocs = 0, reuses = 0, frees = 0;
  double malloc_ms = 0, free_ms = 0;
  size_t cached_bytes = 0, live_bytes = 0, peak_cached_bytes = 0;
};

class DevicePool {
 public:
  static DevicePool &get() {
    static DevicePool *p = new DevicePool;   // never destroyed: handles may outlive static destruction
    return *p;
  }
  void *acquire(size_t bytes) {
    if (!bytes) return nullptr;
    int dev = 0;
    POGS_HIP_CHECK(hipGetDevice(&dev));
    const size_t want = size_class(bytes);
    bool need_sync = false;
    unsigned long long upto = 0;
    void *ptr = nullptr;
    {
      std::lock_guard<std::mutex> lock(mu_);
      PerDevice &pd = dev_[dev & (kMaxPoolDevices - 1)];
      auto it = pd.idle.lower_bound(want);
      if (it != pd.idle.end() && it->first <= want + want / 8) {
        ptr = it->second.p;
        need_sync = it->second.stamp > pd.synced_seq;
        upto = pd.release_seq;
        live_[ptr] = Live{it->first, dev};   // the block keeps the size it was allocated with
        pd.c.cached_bytes -= it->first;
        pd.c.live_bytes += it->first;
        pd.c.reuses++;
        pd.idle.erase(it);
      }
    }
    ptr = ptr ? finish_reuse(dev, ptr, need_sync, upto) : fresh(dev, want);
    // POGS_AMD_POOL_POISON=1 (test aid): every block starts as NaN bytes, so that code which counts
    // on fresh memory being zero fails the same way on its first handle as on a recycled block
    static const bool poison = [] { const char *e = std::getenv("POGS_AMD_POOL_POISON"); return e && e[0] == '1'; }();
    if (poison) {
      try {
        POGS_HIP_CHECK(hipMemset(ptr, 0xFF, bytes));
        POGS_HIP_CHECK(hipDeviceSynchronize());
      } catch (...) {
        release(ptr);   // the block is registered as live: hand it back instead of losing it
        throw;
      }
    }
    return ptr;
  }
  void release(void *p) {
    if (!p) return;
    std::unique_lock<std::mutex> lock(mu_);
    auto it = live_.find(p);
    if (it != live_.end() && dev_[it->second.device & (kMaxPoolDevices - 1)].cap == static_cast<size_t>(-1)) {
      // first release on this device: read the bound once, OUTSIDE the lock and with the BLOCK's device
      // current (hipMemGetInfo answers for the current device, which need not be the block's)
      const int bdev = it->second.device;
      lock.unlock();
      const size_t cap = query_capacity(bdev);
      lock.lock();
      PerDevice &pd0 = dev_[bdev & (kMaxPoolDevices - 1)];
      if (pd0.cap == static_cast<size_t>(-1)) pd0.cap = cap;
      it = live_.find(p);
    }
    if (it == live_.end()) {   // not ours (should not happen): hand it to the runtime
      lock.unlock();
      (void)hipFree(p);
      return;
    }
    const Live lv = it->second;
    live_.erase(it);
    PerDevice &pd = dev_[lv.device & (kMaxPoolDevices - 1)];
    pd.c.live_bytes -= lv.bytes;
    const size_t cap = pd.cap;
    if (lv.bytes > cap) {
      lock.unlock();
      timed_free(lv.device, p, pd);
      return;
    }
    Idle id;
    id.p = p;
    // inside a QuiescedScope the caller has waited for everything that used the block: stamp 0 = no
    // wait needed when it is handed out again
    id.stamp = quiesced_depth() > 0 ? 0 : ++pd.release_seq;
    pd.idle.emplace(lv.bytes, id);
    pd.c.cached_bytes += lv.bytes;
    // over the bound: give the largest idle blocks back to the runtime
    std::vector<void *> evict;
    while (pd.c.cached_bytes > cap && !pd.idle.empty()) {
      auto last = std::prev(pd.idle.end());
      if (last->second.p == p && pd.idle.size() > 1) {   // prefer to keep what was just released
        auto before = std::prev(last);
        last = before;
      }
      pd.c.cached_bytes -= last->first;
      evict.push_back(last->second.p);
      pd.idle.erase(last);
    }
    pd.c.peak_cached_bytes = std::max(pd.c.peak_cached_bytes, pd.c.cached_bytes);
    lock.unlock();
    for (void *q : evict) timed_free(lv.device, q, pd);
  }
  // gives every idle block of `device` (-1: all devices) back to the runtime; returns the bytes freed
  size_t trim(int device) {
    std::vector<std::pair<int, void *>> out;
    size_t bytes = 0;
    {
      std::lock_guard<std::mutex> lock(mu_);
      for (int d = 0; d < kMaxPoolDevices; ++d) {
        if (device >= 0 && d != (device & (kMaxPoolDevices - 1))) continue;
        PerDevice &pd = dev_[d];
        for (auto &kv : pd.idle) { out.emplace_back(d, kv.second.p); bytes += kv.first; }
        pd.idle.clear();
        pd.c.cached_bytes = 0;
      }
    }
    for (auto &dp : out) timed_free(dp.first, dp.second, dev_[dp.first]);
    return bytes;
  }
  static int &quiesced_depth() {
    static thread_local int depth = 0;
    return depth;
  }
  // RAII: the calling thread guarantees that nothing in flight uses the blocks it releases while the
  // scope is open (it has synchronised the streams that touched them)
  struct QuiescedScope {
    QuiescedScope() { ++quiesced_depth(); }
    ~QuiescedScope() { --quiesced_depth(); }
    QuiescedScope(const QuiescedScope &) = delete;
    QuiescedScope &operator=(const QuiescedScope &) = delete;
  };
  PoolCounters counters(int device) {
    std::lock_guard<std::mutex> lock(mu_);
    return dev_[device & (kMaxPoolDevices - 1)].c;
  }

 private:
  static constexpr int kMaxPoolDevices = 64;
  struct Idle { void *p; unsigned long long stamp; };
  struct Live { size_t bytes; int device; };
  struct PerDevice {
    std::multimap<size_t, Idle> idle;
    PoolCounters c;
    size_t cap = static_cast<size_t>(-1);   // -1: not read yet
    // every release takes the next release_seq; blocks stamped <= synced_seq were released before a
    // device-wide wait began, i.e. nothing in flight can still touch them
    unsigned long long release_seq = 0, synced_seq = 0;
  };
  std::mutex mu_;
  PerDevice dev_[kMaxPoolDevices];
  std::unordered_map<void *, Live> live_;

  static size_t size_class(size_t bytes) {
    const size_t g = bytes <= 4096 ? 512 : bytes <= (1u << 20) ? 4096 : (2u << 20);
    return (bytes + g - 1) / g * g;
  }
  static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }
  // idle bytes the pool may hold on `device` (no lock held; the device is made current for the query)
  static size_t query_capacity(int device) {
    if (const char *e = std::getenv("POGS_AMD_POOL_MB")) return static_cast<size_t>(std::max(0.0, std::atof(e)) * 1048576.0);
    int cur = -1;
    const bool sw = hipGetDevice(&cur) == hipSuccess && cur != device && hipSetDevice(device) == hipSuccess;
    size_t free_b = 0, total_b = 0;
    const size_t cap = hipMemGetInfo(&free_b, &total_b) == hipSuccess ? total_b / 4 : (static_cast<size_t>(16) << 30);
    if (sw) (void)hipSetDevice(cur);
    return cap;
  }
  void *finish_reuse(int dev, void *ptr, bool need_sync, unsigned long long upto) {
    if (need_sync) {
      // the block (and possibly others) was released with work in flight: one wait covers every
      // block released before it began.  (A handle's own blocks never take this path: ~Ctx waits for
      // the handle's stream and releases inside a QuiescedScope, so they come back already "synced";
      // what is left are temporaries of the set-up phases, released right after a stream wait too.)
      hipError_t e = hipDeviceSynchronize();
      if (e != hipSuccess) {
        release(ptr);   // registered as live above: back to the cache, not lost
        POGS_HIP_CHECK(e);
      }
      std::lock_guard<std::mutex> lock(mu_);
      PerDevice &pd = dev_[dev & (kMaxPoolDevices - 1)];
      pd.synced_seq = std::max(pd.synced_seq, upto);
    }
    return ptr;
  }
  void *fresh(int dev, size_t want) {
    void *p = nullptr;
    const double t0 = now_ms();
    hipError_t e = hipMalloc(&p, want);
    if (e == hipErrorOutOfMemory) {
      (void)hipGetLastError();
      trim(dev);
      e = hipMalloc(&p, want);
    }
    const double dt = now_ms() - t0;
    POGS_HIP_CHECK(e);
    std::lock_guard<std::mutex> lock(mu_);
    PerDevice &pd = dev_[dev & (kMaxPoolDevices - 1)];
    pd.c.mallocs++;
    pd.c.malloc_ms += dt;
    pd.c.live_bytes += want;
    live_[p] = Live{want, dev};
    return p;
  }
  void timed_free(int dev, void *p, PerDevice &pd) {
    const double t0 = now_ms();
    (void)hipFree(p);
    const double dt = now_ms() - t0;
    std::lock_guard<std::mutex> lock(mu_);
    pd.c.frees++;
    pd.c.free_ms += dt;
    (void)dev;
  }
};

// RAII device buffer (memory from the DevicePool).
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
    if (count) p = static_cast<T *>(DevicePool::get().acquire(count * sizeof(T)));
  }
  void release() {
    if (p) DevicePool::get().release(p);
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

// Fused multiply-adds are WRITTEN (every translation unit but gemm.hip is compiled with -ffp-contract=off,
// pogs_amd/build.py): dev::fma_ where one rounding is meant, plain operators where two are.  sq_acc / prod_acc: the
// fp64 scalar sums of the row and column functors, s += (double) x * y in one rounding (the form the contracting
// compiler had given every one of them).
namespace dev {
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <typename X, typename Y>
__device__ __forceinline__ void prod_acc(double &s, X x, Y y) { s = __builtin_fma(static_cast<double>(x), static_cast<double>(y), s); }
}  // namespace dev

}  // namespace pogs_amd
