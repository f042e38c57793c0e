// RCCL (xGMI) sum all-reduce for the row-sharded solve.
//
// The reference is single-process (no collective anywhere);
// this is the MI355X-native addition of SURVEY.md section 8(e): one process per
// GPU, A row-sharded, and the only data exchanged per iteration is one n-vector
// (A_k^T y_k partials) plus a handful of scalars.
//
// RCCL is bound at run time with dlopen: if the process already has a librccl
// (PyTorch ships its own copy) that instance is reused, so the library never
// ends up with two RCCL / HIP runtimes in one address space.
//
// One collective per exchange: a vector and the scalar sums that travel with it are packed
// into one fp64 staging buffer (allreduce2 / allreduce3: pack kernel, ONE ncclAllReduce,
// unpack kernel); the one-pass dense iteration fills its pack buffer straight from the producing
// kernels and calls allreduce() on it (dense_solver.h).  The one place where two buffers of different
// type must travel at the same point (sparse.hip: a CG step's A^T q and its |q|^2 records) groups
// the two calls (group_begin / group_end) so that they are still one launch.
//
// Other transports: a unique id that starts with "POGS" (RCCL's ids are binary and never do) is served by the
// plug-in named by POGS_AMD_TRANSPORT_PLUGIN (transport_plugin.h); without that variable such an id is refused.
// The test-suite's communicators -- ranks as threads of one process, or as processes joined by shared memory,
// on ONE GPU, which RCCL refuses -- are such a plug-in (tests/transport/test_transport.hip); none of that code
// is in this library.
#pragma once
#include <exception>
#include <hip/hip_runtime.h>

#include <cstddef>

namespace pogs_amd {

constexpr int kUniqueIdBytes = 128;

class DistComm {
 public:
  DistComm() = default;
  ~DistComm();
  DistComm(const DistComm &) = delete;
  DistComm &operator=(const DistComm &) = delete;

  // Collective: every rank calls with the same unique id.
  void init(int rank, int world, const char *unique_id);
  bool active() const { return comm_ != nullptr || local_ != nullptr; }
  int rank() const { return rank_; }
  int world() const { return world_; }

  // In-place sum all-reduce on `stream`.  No-ops when not initialised.
  void allreduce(float *buf, size_t count, hipStream_t stream) const;
  void allreduce(double *buf, size_t count, hipStream_t stream) const;
  // out-of-place: out = sum over ranks of in (in is not written; in != out)
  void allreduce(const double *in, double *out, size_t count, hipStream_t stream) const;
  // The all-reduces issued between group_begin() and group_end() are independent of each other and
  // travel as ONE RCCL launch (ncclGroupStart / ncclGroupEnd): the n-vector and the record array of a
  // row-sharded CG step (sparse.hip).  A plug-in transport runs them one after the other.
  void group_begin() const;
  void group_end() const;
  // RAII form: ncclGroupEnd is called on every way out of the scope.  When an exception unwinds
  // through it (the first all-reduce failed, the communicator was aborted) the group is still closed
  // -- an open thread-local RCCL group would swallow every later call of the thread -- and a second
  // error from that closing call is dropped.
  class Group {
   public:
    explicit Group(const DistComm &c) : c_(c), exc_(std::uncaught_exceptions()) { c_.group_begin(); }
    ~Group() noexcept(false) {
      if (std::uncaught_exceptions() > exc_) {
        try { c_.group_end(); } catch (...) {}
      } else {
        c_.group_end();
      }
    }
    Group(const Group &) = delete;
    Group &operator=(const Group &) = delete;
   private:
    const DistComm &c_;
    int exc_;
  };
  // A vector and one / two scalar ranges as ONE all-reduce of a packed fp64 buffer.
  template <typename T>
  void allreduce2(T *buf, size_t count, double *scalars, size_t nscalars, hipStream_t stream);
  template <typename T>
  void allreduce3(T *buf, size_t count, double *s1, size_t n1, double *s2, size_t n2, hipStream_t stream);
  unsigned long long collectives() const { return ncoll_; }   // all-reduce calls issued so far
  // Ranks of the communicator as RCCL itself reports them (ncclCommCount); world() for a
  // plug-in transport, 0 when there is no communicator.
  int comm_nranks() const;
  // A collective that a peer never joined leaves this rank's stream inside the all-reduce kernel for
  // ever.  abort() tears the communicator down (ncclCommAbort: the kernel returns, later calls on
  // this handle fail); Ctx::wait_publish calls it when a publish does not arrive within
  // POGS_AMD_COLL_TIMEOUT_S seconds (default 300), and the entry point returns POGS_ERROR.
  void abort();
  bool aborted() const { return aborted_; }
  // "" or RCCL's description of an asynchronous error on the communicator (ncclCommGetAsyncError)
  const char *async_error() const;

  static void unique_id(char *out);  // fresh id (rank 0)

 private:
  void reduce_raw(const void *in, void *out, size_t count, int dtype, hipStream_t stream) const;
  int rank_ = 0, world_ = 1;
  void *comm_ = nullptr;
  void *local_ = nullptr;   // handle of a plug-in transport (transport_plugin.h)
  const void *plug_ = nullptr;   // its PogsAmdTransportApi
  double *pack_ = nullptr;  // fp64 staging of allreduce2 / allreduce3 (device)
  size_t pack_cap_ = 0;
  mutable unsigned long long ncoll_ = 0;
  bool aborted_ = false;
};

}  // namespace pogs_amd
