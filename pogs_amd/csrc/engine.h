// Engine plumbing shared by the dense and sparse solvers: execution context,
// host-side ADMM control (stopping rules + adaptive rho), solver base class.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pogs_amd.h"
#include "common.h"
#include "dist.h"
#include "vec_kernels.h"

namespace pogs_amd {

inline double wall_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// HIP's current device is per thread: every extern "C" entry that touches a handle selects the
// handle's device for its duration and puts the caller's device back on exit (a handle may be
// used from another thread, or after the caller -- e.g. torch -- switched devices).
class DeviceGuard {
 public:
  explicit DeviceGuard(int dev) {
    if (dev < 0) return;
    if (hipGetDevice(&prev_) != hipSuccess) { prev_ = -1; }
    if (prev_ != dev) {
      POGS_HIP_CHECK(hipSetDevice(dev));
      restore_ = prev_ >= 0;
    }
  }
  ~DeviceGuard() {
    if (restore_) (void)hipSetDevice(prev_);
  }
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;
 private:
  int prev_ = -1;
  bool restore_ = false;
};

// HIP-event stopwatch for kernels launched on one stream (profile mode only).
class EventTimer {
 public:
  void enable(bool on) { on_ = on; }
  bool enabled() const { return on_; }
  // Time every k-th bracket only: an event record costs the stream 3-4 us of idle time on either
  // side of the kernel it brackets (rocprofv3 trace at C2: 5.9 us gaps around the pass over A with
  // every launch bracketed), i.e. measuring every launch slows down what is measured by ~1 %.
  void set_every(unsigned k) { every_ = k ? k : 1; }
  static constexpr size_t npos = static_cast<size_t>(-1);
  // returns the index of the pair (see drop), npos if this bracket is not sampled
  size_t begin(hipStream_t s) {
    active_ = on_ && (calls_++ % every_) == 0;
    if (!active_) return npos;
    if (used_ == ev_.size()) {
      hipEvent_t a, b;
      POGS_HIP_CHECK(hipEventCreate(&a));
      POGS_HIP_CHECK(hipEventCreate(&b));
      ev_.push_back({a, b});
      dropped_.push_back(0);
    }
    dropped_[used_] = 0;
    POGS_HIP_CHECK(hipEventRecord(ev_[used_].first, s));
    return used_;
  }
  // a launch that turned out to be a no-op (a guarded launch past the end of a device-side loop)
  // does not count as a launch of the kernel
  void drop(size_t idx) {
    if (on_ && idx < dropped_.size()) dropped_[idx] = 1;
  }
  void end(hipStream_t s) {
    if (!active_) return;
    POGS_HIP_CHECK(hipEventRecord(ev_[used_].second, s));
    ++used_;
    active_ = false;
  }
  // Sum of elapsed ms over all recorded pairs; the stream must be idle.
  double collect_ms(unsigned long long *count) {
    double tot = 0;
    size_t kept = 0;
    for (size_t i = 0; i < used_; ++i) {
      if (dropped_[i]) continue;
      float ms = 0;
      if (hipEventElapsedTime(&ms, ev_[i].first, ev_[i].second) == hipSuccess) tot += ms;
      ++kept;
    }
    if (count) *count += kept;
    used_ = 0;
    return tot;
  }
  ~EventTimer() {
    for (auto &p : ev_) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
  }
 private:
  bool on_ = false, active_ = false;
  unsigned every_ = 1;
  unsigned long long calls_ = 0;
  size_t used_ = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_;
  std::vector<char> dropped_;
};

// One-shot event pair for setup phases.
struct PhaseTimer {
  hipEvent_t a = nullptr, b = nullptr;
  hipStream_t s;
  explicit PhaseTimer(hipStream_t st) : s(st) {
    POGS_HIP_CHECK(hipEventCreate(&a));
    POGS_HIP_CHECK(hipEventCreate(&b));
    POGS_HIP_CHECK(hipEventRecord(a, s));
  }
  double stop_ms() {
    POGS_HIP_CHECK(hipEventRecord(b, s));
    POGS_HIP_CHECK(hipEventSynchronize(b));
    float ms = 0;
    POGS_HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    return ms;
  }
  ~PhaseTimer() {
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
  }
};

// Streams and the host-mapped scalar mirror are recycled across solver handles of a process:
// a one-shot solve creates and destroys a handle per
// call, and stream / mapped-host churn costs milliseconds per call plus occasional
// tens-of-milliseconds stalls inside the runtime.  Entries are never destroyed (a handful
// of streams and 4 KB blocks per device for the life of the process).
struct CtxResources {
  hipStream_t stream = nullptr;
  double *host = nullptr;       // kNumSlots + 1 doubles, hipHostMallocMapped | Coherent
  int device = -1;
};
inline std::mutex &ctx_pool_mutex() { static std::mutex *m = new std::mutex; return *m; }
inline std::vector<CtxResources> &ctx_pool() { static auto *v = new std::vector<CtxResources>; return *v; }
struct Ctx {
  int device = 0;
  int num_cu = 256;
  hipStream_t stream = nullptr;
  DistComm dist;
  size_t m_global = 0;
  DevBuf<double> S;            // device scalar block [kNumSlots]
  struct HostMirror { double *p = nullptr; } S_host;   // host-mapped mirror (CtxResources)
  DevBuf<double> spart;        // scalar partial sums scratch
  size_t spart_cap = 0;
  EventTimer stream_timer;
  PogsAmdStats stats;

  void init(int dev, int profile) {
    if (dev >= 0) POGS_HIP_CHECK(hipSetDevice(dev));
    POGS_HIP_CHECK(hipGetDevice(&device));
    hipDeviceProp_t prop;
    POGS_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    {
      std::lock_guard<std::mutex> lock(ctx_pool_mutex());
      auto &pool = ctx_pool();
      for (size_t i = 0; i < pool.size(); ++i)
        if (pool[i].device == device) {
          stream = pool[i].stream;
          S_host.p = pool[i].host;
          pool.erase(pool.begin() + static_cast<long>(i));
          break;
        }
    }
    if (!stream) {
      POGS_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
      // + the sequence word of fetch_scalars
      POGS_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&S_host.p), (kNumSlots + 1) * sizeof(double),
                                   hipHostMallocMapped | hipHostMallocCoherent));
    }
    S.alloc(kNumSlots);
    S.zero(stream);
    pub_counter.alloc(1);
    pub_counter.zero(stream);
    std::memset(S_host.p, 0, (kNumSlots + 1) * sizeof(double));
    {
      void *dp = nullptr;
      poll_fetch = true;
      if (hipHostGetDevicePointer(&dp, S_host.p, 0) == hipSuccess && dp) S_host_dev = static_cast<double *>(dp);
      else poll_fetch = false;
    }
    std::memset(&stats, 0, sizeof(stats));
    stream_timer.enable(profile != 0);
    stream_timer.set_every(profile > 1 ? static_cast<unsigned>(profile) : 1u);
  }
  void ensure_spart(size_t count) {
    if (count > spart_cap) {
      POGS_HIP_CHECK(hipStreamSynchronize(stream));
      spart.alloc(count);
      spart_cap = count;
    }
  }
  // Brings the scalar block to the host and waits for everything enqueued so far.  Default: a
  // one-wave kernel writes the block into the host-mapped mirror and then a sequence word the
  // host polls (the stream is in order, so seeing the word means all earlier work is done);
  // that saves the copy-engine round trip and the synchronize call of the plain copy, which is
  // the fallback if the mirror has no device address or the poll sees no progress.
  // Deferred scalar sums: jobs queued here run in the launch that publishes the scalar block
  // (sum_publish_kernel), so an iteration ends with one small launch instead of one per sum plus
  // the publish.  Whoever queues a job keeps its partials untouched until the next fetch.
  void queue_sum(const SumJob &j) {
    if (npending == kMaxSumJobs) flush_sums();
    pending[npending++] = j;
  }
  void flush_sums() {
    if (npending) launch_sum_jobs(pending, npending, stream);
    npending = 0;
  }
  // scalars that the next fetch takes from a packed all-reduce buffer (consumed by that fetch)
  void set_overlay(const ScalarOverlay &ov) { overlay = ov; }
  // polls the host-mapped sequence word until the publishing launch has raised it to `want`
  const double *wait_publish(unsigned long long want) {
    unsigned long long *seqp = reinterpret_cast<unsigned long long *>(S_host.p + kNumSlots);
    unsigned spins = 0, idle_seen = 0, checks = 0;
    double t_wait0 = 0;
    while (__atomic_load_n(seqp, __ATOMIC_ACQUIRE) != want) {
      if (++spins == (1u << 14)) {   // ~ every few hundred microseconds: surface a failed stream
        spins = 0;
        // Row shards: a peer that never joins a collective (it failed, or was killed) leaves this
        // rank's stream inside the all-reduce for ever.  After coll_timeout_s without the publish
        // (or as soon as RCCL reports an asynchronous error) the communicator is aborted and the
        // call fails with POGS_ERROR instead of hanging.
        if (dist.active() && (++checks & 255u) == 0) {
          const double now = wall_s();
          if (t_wait0 == 0) t_wait0 = now;
          const char *ae = dist.async_error();
          if (ae[0] || now - t_wait0 > coll_timeout_s()) {
            const std::string why = ae[0] ? std::string("RCCL reported: ") + ae
                                          : "no progress for " + std::to_string(static_cast<int>(coll_timeout_s())) +
                                                " s inside a collective (a peer rank did not join it)";
            dist.abort();
            poisoned = true;
            throw Error("row-sharded solve aborted: " + why);
          }
        }
        const hipError_t q = hipStreamQuery(stream);
        if (q == hipSuccess) {
          if (__atomic_load_n(seqp, __ATOMIC_ACQUIRE) == want) break;
          // the stream is idle (so the block on the device is final) but the word has not
          // shown up in the mirror: never spin on that -- copy the block and stop polling
          if (++idle_seen >= 3) {
            POGS_HIP_CHECK(hipMemcpy(S_host.p, S.p, kNumSlots * sizeof(double), hipMemcpyDeviceToHost));
            poll_fetch = false;
            return S_host.p;
          }
        } else if (q != hipErrorNotReady) {
          POGS_HIP_CHECK(q);
        }
      }
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    return S_host.p;
  }
  const double *fetch_scalars() {
    const ScalarOverlay ov = overlay;
    overlay = ScalarOverlay();
    if (!poll_fetch) {
      flush_sums();
      launch_apply_overlay(S.p, ov, stream);
      POGS_HIP_CHECK(hipMemcpyAsync(S_host.p, S.p, kNumSlots * sizeof(double), hipMemcpyDeviceToHost, stream));
      POGS_HIP_CHECK(hipStreamSynchronize(stream));
      return S_host.p;
    }
    const unsigned long long want = ++fetch_seq;
    if (npending) {
      launch_sum_publish(pending, npending, S.p, kNumSlots, S_host_dev,
                         reinterpret_cast<unsigned long long *>(S_host_dev + kNumSlots), want, pub_counter.p, stream, ov);
      npending = 0;
    } else {
      launch_publish_scalars(S.p, kNumSlots, S_host_dev, reinterpret_cast<unsigned long long *>(S_host_dev + kNumSlots),
                             want, stream, ov);
    }
    return wait_publish(want);
  }
  static double coll_timeout_s() {
    static const double v = [] {
      const char *e = std::getenv("POGS_AMD_COLL_TIMEOUT_S");
      const double t = e ? std::atof(e) : 300.0;
      return t > 0 ? t : 300.0;
    }();
    return v;
  }
  bool poll_fetch = true;
  ScalarOverlay overlay;
  SumJob pending[kMaxSumJobs];
  int npending = 0;
  DevBuf<unsigned> pub_counter;
  unsigned long long fetch_seq = 0;
  double *S_host_dev = nullptr;   // device address of the host-mapped mirror
  void sync() {
    flush_sums();
    POGS_HIP_CHECK(hipStreamSynchronize(stream));
  }
  // POGS_AMD_TRACE=1: host-side setup timeline on stderr (time to reach the mark on the host,
  // then the extra wait for the stream to drain) -- finds host stalls the kernel trace hides.
  void tmark(const char *label) {
    static const bool on = std::getenv("POGS_AMD_TRACE") != nullptr;
    if (!on) return;
    const double t1 = wall_s();
    POGS_HIP_CHECK(hipStreamSynchronize(stream));
    const double t2 = wall_s();
    // what the phase took from / gave to the HIP runtime (DevicePool, common.h): a phase that
    // stalls in hipMalloc / hipFree shows here, a first-touch stall shows as drain time
    const PoolCounters pc = DevicePool::get().counters(device);
    std::fprintf(stderr, "[pogs_amd trace] %-28s host +%8.3f ms, drain +%8.3f ms | pool: +%llu malloc %.3f ms, +%llu reuse, "
                 "+%llu free %.3f ms\n", label, (t1 - tmark_last) * 1e3, (t2 - t1) * 1e3,
                 pc.mallocs - tmark_pool.mallocs, pc.malloc_ms - tmark_pool.malloc_ms, pc.reuses - tmark_pool.reuses,
                 pc.frees - tmark_pool.frees, pc.free_ms - tmark_pool.free_ms);
    tmark_pool = pc;
    tmark_last = wall_s();
  }
  double tmark_last = 0;
  PoolCounters tmark_pool;
  bool poisoned = false;   // set when an error left the stream / communicator in an unknown state
  // an exception escaped a solve on this context: with row shards the peers are (or will be) inside
  // a collective this rank no longer takes part in -- abort the communicator so that they fail too,
  // and never hand this stream to another solver
  // Only when a collective of the failing call was (or may have been) enqueued: an argument or state
  // check that throws before any exchange leaves the peers nothing to wait for, and must not cost the
  // handle (and theirs) its communicator.
  unsigned long long coll_mark = 0;
  void on_entry() { coll_mark = dist.collectives(); }
  void on_error() {
    if (dist.active() && dist.collectives() != coll_mark) {
      dist.abort();
      poisoned = true;
    }
  }
  ~Ctx() {
    if (!stream) return;
    DeviceGuard guard(device);
    // a stream that faulted (or was left inside a failed collective) must not be handed to the
    // next solver: recycle it only if it drains cleanly
    const bool healthy = !poisoned && hipStreamSynchronize(stream) == hipSuccess;
    if (healthy) {
      std::lock_guard<std::mutex> lock(ctx_pool_mutex());
      CtxResources r;
      r.stream = stream;
      r.host = S_host.p;
      r.device = device;
      ctx_pool().push_back(r);
    } else {
      (void)hipStreamDestroy(stream);
      if (S_host.p) (void)hipHostFree(S_host.p);
    }
  }
};

// Host-side scalar logic of PogsImplementation::Solve for a separable objective
// (kUseExactTol = false): constants pogs.cpp:94-110, tolerances :199-201,270-273,
// projection tolerance :287-290, stopping rule :379-394, adaptive rho :402-466.
// Arithmetic is carried out in T exactly where the reference uses T.
template <typename T>
struct AdmmControl {
  // parameters
  T abs_tol = 0, rel_tol = 0;
  unsigned max_iter = 0;
  bool adaptive_rho = true, gap_stop = false;
  T rho0 = 1;
  size_t m_glob = 0, n = 0;
  // state
  T rho = 1, delta = 0, xi = 1;
  unsigned k = 0, kd = 0, ku = 0;
  T prev_nrm_r = 0;
  T nrm_r = 0, nrm_s = 0, gap = 0, eps_gap = 0, eps_pri = 0, eps_dua = 0;
  bool converged = false, finished = false;
  unsigned exact_iters = 0, rho_updates = 0;
  bool say_rho = false;   // verbose > 3: the reference's rho messages (pogs.cpp:432-459)
  T sqrtn_atol = 0, sqrtm_atol = 0, sqrtmn_atol = 0;

  static constexpr double kAlphaD = 1.7;
  T alpha() const { return static_cast<T>(kAlphaD); }

  void reset() {
    rho = rho0;
    delta = static_cast<T>(1.05);
    xi = 1;
    k = kd = ku = 0;
    prev_nrm_r = std::numeric_limits<T>::max();
    converged = finished = false;
    exact_iters = rho_updates = 0;
    sqrtn_atol = std::sqrt(static_cast<T>(n)) * abs_tol;
    sqrtm_atol = std::sqrt(static_cast<T>(m_glob)) * abs_tol;
    sqrtmn_atol = std::sqrt(static_cast<T>(m_glob + n)) * abs_tol;
  }

  // After the prox step: S holds the pre-projection sums.
  void set_pre(const double *S) {
    gap = std::abs(static_cast<T>(S[kGapX] + S[kGapY]));
    const T nz = static_cast<T>(std::sqrt(S[kWX2] + S[kWY2]));
    const T nz12 = static_cast<T>(std::sqrt(S[kHX2] + S[kHY2]));
    const T ny12 = static_cast<T>(std::sqrt(S[kHY2]));
    const T nx = static_cast<T>(std::sqrt(S[kWX2]));
    eps_gap = sqrtmn_atol + rel_tol * nz * nz12;
    eps_pri = sqrtm_atol + rel_tol * ny12;
    eps_dua = rho * (sqrtn_atol + rel_tol * nx);
  }
  T proj_tol() const {
    T tol = static_cast<T>(1e-2) * std::pow(std::min(prev_nrm_r, static_cast<T>(1)), static_cast<T>(0.5));
    return std::max(tol, static_cast<T>(1e-8));
  }
  // After the projection: cheap residual bounds; returns whether the exact
  // residuals must be evaluated.
  bool set_approx(const double *S, T nrmA) {
    nrm_s = rho * (nrmA * static_cast<T>(std::sqrt(S[kDYprev2])) + static_cast<T>(std::sqrt(S[kDXprev2])));
    nrm_r = nrmA * static_cast<T>(std::sqrt(S[kDX12])) + static_cast<T>(std::sqrt(S[kDY12]));
    return nrm_r < 10 * eps_pri && nrm_s < 10 * eps_dua;
  }
  void set_exact(const double *S) {
    nrm_r = static_cast<T>(std::sqrt(S[kExactR2]));
    nrm_s = rho * static_cast<T>(std::sqrt(S[kExactS2]));
    ++exact_iters;
  }
  // Returns true when the solve stops at this iteration (k is then final_iter).
  bool check_stop(bool exact) {
    converged = exact && nrm_r < eps_pri && nrm_s < eps_dua && (!gap_stop || gap < eps_gap);
    if (converged || k == max_iter - 1) {
      finished = true;
      return true;
    }
    return false;
  }
  // Adaptive rho; returns the factor zt must be multiplied by (1 = unchanged).
  T adapt() {
    T scale = 1;
    const T kDeltaMin = static_cast<T>(1.05), kGamma = static_cast<T>(1.01), kTau = static_cast<T>(0.8);
    const T kRhoMin = static_cast<T>(1e-4), kRhoMax = static_cast<T>(1e4), kKappa = static_cast<T>(0.9);
    const T kOne = 1, kZero = 0;
    if (adaptive_rho) {
      const unsigned kRhoUpdateFreq = 50u;
      const T kRhoChangeMax = static_cast<T>(1.5), kRhoChangeMin = static_cast<T>(0.67);
      const T kImbalanceThresh = static_cast<T>(10);
      if (k > 0 && k % kRhoUpdateFreq == 0 && eps_pri > kZero && eps_dua > kZero) {
        const T pri_n = nrm_r / eps_pri, dua_n = nrm_s / eps_dua;
        if (pri_n > kZero && dua_n > kZero) {
          const T imbalance = pri_n / dua_n;
          if (imbalance > kImbalanceThresh || imbalance < kOne / kImbalanceThresh) {
            T ratio = std::sqrt(imbalance);
            ratio = std::max(kRhoChangeMin, std::min(kRhoChangeMax, ratio));
            T rho_new = rho * ratio;
            rho_new = std::max(kRhoMin, std::min(kRhoMax, rho_new));
            if (std::abs(rho_new - rho) / rho > static_cast<T>(0.05)) {
              scale = rho / rho_new;
              rho = rho_new;
              ++rho_updates;
              if (say_rho) std::printf("spectral rho update: %e (imbalance=%.1f)\n", (double)rho, (double)imbalance);
            }
          }
        }
      } else if (nrm_s < xi * eps_dua && nrm_r > xi * eps_pri && kTau * static_cast<T>(k) > static_cast<T>(kd)) {
        if (rho < kRhoMax) {
          rho *= delta;
          scale = 1 / delta;
          delta = kGamma * delta;
          ku = k;
          ++rho_updates;
          if (say_rho) std::printf("+ rho %e\n", (double)rho);
        }
      } else if (nrm_s > xi * eps_dua && nrm_r < xi * eps_pri && kTau * static_cast<T>(k) > static_cast<T>(ku)) {
        if (rho > kRhoMin) {
          rho /= delta;
          scale = delta;
          delta = kGamma * delta;
          kd = k;
          ++rho_updates;
          if (say_rho) std::printf("- rho %e\n", (double)rho);
        }
      } else if (nrm_s < xi * eps_dua && nrm_r < xi * eps_pri) {
        xi *= kKappa;
      } else {
        delta = kDeltaMin;
      }
    }
    prev_nrm_r = nrm_r;
    return scale;
  }
  // What adapt() would do at this iteration if the residuals equalled the previous
  // iteration's (they drift slowly): returns the predicted (rho, zt scale) without
  // touching the state.  Used to speculate across a rho change (dense_solver.h).
  void predict(T *rho_out, T *scale_out) const {
    AdmmControl<T> c = *this;   // nrm_r, nrm_s, eps_* still hold the previous iteration's values
    c.say_rho = false;
    *scale_out = c.adapt();
    *rho_out = c.rho;
  }
  int status() const {
    if (!converged && k == max_iter - 1) return POGS_MAX_ITER;
    if (!converged) return POGS_NAN_FOUND;
    return POGS_SUCCESS;
  }
};

// ---- console output in the reference's format (src/cpu/pogs.cpp:26-27,185-196,382-388,485-500,
// src/include/pogs.h:168-186), so that a caller parsing the reference's verbose output keeps working
#define POGS_AMD_HBAR "----------------------------------------------------------------------------\n"
inline const char *status_string(int st) {
  switch (st) {
    case POGS_SUCCESS: return "Solved";
    case POGS_UNBOUNDED: return "Unbounded";
    case POGS_INFEASIBLE: return "Infeasible";
    case POGS_MAX_ITER: return "Reached max iter";
    case POGS_NAN_FOUND: return "Encountered NaN";
    case POGS_INVALID_CONE: return "Invalid cone found";
    default: return "Error";
  }
}
inline void print_banner(unsigned verbose) {
  if (verbose > 0)
    std::printf(POGS_AMD_HBAR
                "           POGS v0.4.0 - Proximal Graph Solver (MI355X / HIP engine)\n"
                "           graph-form ADMM of foges/pogs, rebuilt for gfx950\n");
  if (verbose > 1)
    std::printf(POGS_AMD_HBAR " Iter | pri res | pri tol | dua res | dua tol |   gap   | eps gap |"
                " pri obj\n" POGS_AMD_HBAR);
}
template <typename T>
inline bool wants_iter_line(unsigned verbose, const AdmmControl<T> &c) {
  return (verbose > 2 && c.k % 10 == 0) || (verbose > 1 && c.k % 100 == 0) || (verbose > 1 && c.converged);
}
template <typename T>
inline void print_iter_line(const AdmmControl<T> &c, double optval) {
  std::printf("%5d : %.2e  %.2e  %.2e  %.2e  %.2e  %.2e % .2e\n", static_cast<int>(c.k), (double)c.nrm_r,
              (double)c.eps_pri, (double)c.nrm_s, (double)c.eps_dua, (double)c.gap, (double)c.eps_gap, optval);
}
template <typename T>
inline void print_summary(int status, double t_total, double t_init, const AdmmControl<T> &c) {
  std::printf(POGS_AMD_HBAR
              "Status: %s\n"
              "Timing: Total = %3.2e s, Init = %3.2e s\n"
              "Iter  : %u\n",
              status_string(status), t_total, t_init, c.k);
  std::printf(POGS_AMD_HBAR
              "Error Metrics:\n"
              "Pri: "
              "|Ax - y|    / (abs_tol sqrt(m)     / rel_tol + |y|)          = %.2e\n"
              "Dua: "
              "|A'l + u|   / (abs_tol sqrt(n)     / rel_tol + |u|)          = %.2e\n"
              "Gap: "
              "|x'u + y'l| / (abs_tol sqrt(m + n) / rel_tol + |x,u| |y,l|)  = %.2e\n" POGS_AMD_HBAR,
              (double)(c.rel_tol * c.nrm_r / c.eps_pri), (double)(c.rel_tol * c.nrm_s / c.eps_dua),
              (double)(c.rel_tol * c.gap / c.eps_gap));
  std::fflush(stdout);
}

// verbose > 3: the reference closes with its per-iteration timing breakdown (pogs.cpp:501-506: prox,
// projection and residual evaluation timed separately on the host).  Here those three are one fused
// pass over A (dense) or share their launches (sparse), so the whole iteration is reported under
// `proj` -- the line keeps the reference's format for whoever parses it.
inline void print_timing_breakdown(double loop_s, unsigned iterations) {
  std::printf("Timing breakdown (per-iter avg): prox = %3.2e s, proj = %3.2e s, residual = %3.2e s\n", 0.0,
              loop_s / std::max(1u, iterations), 0.0);
  std::fflush(stdout);
}

struct SolveParams {
  double rho, abs_tol, rel_tol;
  unsigned max_iter, verbose;
  bool adaptive_rho, gap_stop;
};

struct FnHost {  // host SoA, element type = solver dtype
  const void *a, *b, *c, *d, *e;
  const int *h;
  // PogsAmdSolveFn / PogsAmdBeginRunFn: a field whose pointer is null holds ONE value for every element (a0 .. e0
  // in this order, h0) -- it is filled on the device instead of being built, converted and uploaded by the caller
  double s0[5] = {1.0, 0.0, 1.0, 0.0, 0.0};
  int h0 = 15;   // kZero
  int hval(size_t i) const { return h ? h[i] : h0; }
};

// Host-side scans of the function codes: one look when the codes are broadcast.
template <typename Pred>
inline bool all_h(const FnHost &f, size_t count, Pred pred) {
  if (!f.h) return count == 0 || pred(f.h0);
  for (size_t i = 0; i < count; ++i)
    if (!pred(f.h[i])) return false;
  return true;
}

// The six coefficient arrays of `src` into the device SoA `dst` (cnt elements each): copied from the host, or filled
// on the device where the host gave one value for all.
template <typename T>
inline void upload_fn(FnBuf<T> &dst, const FnHost &src, int cnt, hipStream_t s) {
  if (cnt <= 0) return;
  if (src.h) POGS_HIP_CHECK(hipMemcpyAsync(dst.h.p, src.h, cnt * sizeof(int), hipMemcpyHostToDevice, s));
  else launch_fill_int(dst.h.p, src.h0, static_cast<size_t>(cnt), s);
  const void *ptr[5] = {src.a, src.b, src.c, src.d, src.e};
  T *out[5] = {dst.a.p, dst.b.p, dst.c.p, dst.d.p, dst.e.p};
  for (int k = 0; k < 5; ++k) {
    if (ptr[k]) POGS_HIP_CHECK(hipMemcpyAsync(out[k], ptr[k], cnt * sizeof(T), hipMemcpyHostToDevice, s));
    else launch_fill<T>(out[k], static_cast<T>(src.s0[k]), static_cast<size_t>(cnt), s);
  }
}

// FunctionObj::CheckConsts (src/include/prox_lib.h:62-69): a negative c or e is not convex; the
// reference prints a warning per offending object and uses 0.  The clamp itself happens on the
// device (scale_objective_kernel); this prints the reference's messages, the first kMaxWarn per
// coefficient array and then a count, so that a bad million-element vector stays readable.
template <typename T>
inline unsigned warn_negative_coeffs(const FnHost &f, size_t count) {
  constexpr unsigned kMaxWarn = 8;
  const T *c = static_cast<const T *>(f.c), *e = static_cast<const T *>(f.e);
  unsigned nc = 0, ne = 0;
  // (a broadcast coefficient is one value: one look, the count of objects it stands for)
  if (!c && count && f.s0[2] < 0.0) nc = static_cast<unsigned>(std::min<size_t>(count, 0xFFFFFFFFu));
  if (!e && count && f.s0[4] < 0.0) ne = static_cast<unsigned>(std::min<size_t>(count, 0xFFFFFFFFu));
  for (unsigned w = 0; w < std::min(nc, kMaxWarn); ++w) std::printf("WARNING c < 0. Function not convex. Using c = 0");
  for (unsigned w = 0; w < std::min(ne, kMaxWarn); ++w) std::printf("WARNING e < 0. Function not convex. Using e = 0");
  for (size_t i = 0; i < count && (c || e); ++i) {
    // (the reference's messages carry no newline: Printf at prox_lib.h:64,66)
    if (c && c[i] < static_cast<T>(0) && nc++ < kMaxWarn) std::printf("WARNING c < 0. Function not convex. Using c = 0");
    if (e && e[i] < static_cast<T>(0) && ne++ < kMaxWarn) std::printf("WARNING e < 0. Function not convex. Using e = 0");
  }
  if (nc > kMaxWarn) std::printf("\nWARNING c < 0 in %u more function objects", nc - kMaxWarn);
  if (ne > kMaxWarn) std::printf("\nWARNING e < 0 in %u more function objects", ne - kMaxWarn);
  if (nc + ne) {
    std::printf("\n");
    std::fflush(stdout);
  }
  return nc + ne;
}

// Type-erased solver behind the C handle.
struct SolverBase {
  virtual ~SolverBase() {}
  virtual int dtype() const = 0;
  virtual int device() const = 0;   // HIP device the handle lives on (DeviceGuard in abi.hip)
  virtual int solve(const FnHost &f, const FnHost &g, const SolveParams &p, void *x, void *y, void *l,
                    void *mu, double *optval, unsigned *final_iter) = 0;
  virtual void begin_run(const FnHost &f, const FnHost &g, const SolveParams &p) = 0;
  virtual void iterate(unsigned iters, double *seconds, unsigned *solves) = 0;
  virtual void set_warm_start(const void *x0, const void *l0) = 0;
  virtual void get_equil(void *A_eq, void *d, void *e, double *nrmA) = 0;
  virtual void project(const void *x0, const void *y0, double tol, void *x, void *y) = 0;
  virtual void mul(char trans, double alpha, const void *x, double beta, void *y) = 0;
  virtual PogsAmdStats &stats() = 0;
  // guarded() in abi.hip: on_entry() when an entry point starts working on the handle, on_error()
  // when an exception leaves it
  virtual void on_entry() {}
  virtual void on_error() {}

 protected:
  // First statement of a derived destructor: waits for the handle's stream and, if that succeeded,
  // marks the calling thread "quiesced" until the LAST sub-object of the solver is gone (this base
  // is destroyed after every derived member) -- the device blocks the members release then go back
  // to the pool as safe to hand out at once, and no later handle pays a device-wide wait for them
  // (DevicePool::finish_reuse; ADVICE r04: hipDeviceSynchronize stalled every stream of the device).
  void begin_destroy(Ctx &ctx) {
    if (quiesce_.armed || !ctx.stream || ctx.poisoned) return;
    DeviceGuard guard(ctx.device);
    if (hipStreamSynchronize(ctx.stream) == hipSuccess) quiesce_.arm();
  }

 private:
  struct QuiesceOnDestroy {
    bool armed = false;
    void arm() { ++DevicePool::quiesced_depth(); armed = true; }
    ~QuiesceOnDestroy() { if (armed) --DevicePool::quiesced_depth(); }
  } quiesce_;
};

}  // namespace pogs_amd

struct PogsAmdSolver {
  std::unique_ptr<pogs_amd::SolverBase> impl;
};
