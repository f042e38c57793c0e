// RCCL binding (see dist.h).
#include "dist.h"

#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace pogs_amd {

namespace {

struct UniqueId {
  char internal[kUniqueIdBytes];
};
using ncclComm_t = void *;
constexpr int kNcclSum = 0, kNcclFloat = 7, kNcclDouble = 8;

struct RcclApi {
  int (*GetUniqueId)(UniqueId *) = nullptr;
  int (*CommInitRank)(ncclComm_t *, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*CommAbort)(ncclComm_t) = nullptr;
  int (*CommCount)(const ncclComm_t, int *) = nullptr;
  int (*CommGetAsyncError)(ncclComm_t, int *) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};

RcclApi &api() {
  static RcclApi a;
  static std::once_flag once;
  std::call_once(once, [] {
    void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) throw Error(std::string("cannot load librccl: ") + dlerror());
    auto sym = [&](const char *name) {
      void *p = dlsym(lib, name);
      if (!p) throw Error(std::string("librccl lacks symbol ") + name);
      return p;
    };
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.CommAbort = reinterpret_cast<decltype(a.CommAbort)>(sym("ncclCommAbort"));
    a.CommCount = reinterpret_cast<decltype(a.CommCount)>(sym("ncclCommCount"));
    a.CommGetAsyncError = reinterpret_cast<decltype(a.CommGetAsyncError)>(sym("ncclCommGetAsyncError"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
  });
  return a;
}

void check(int r, const char *what) {
  if (r != 0) {
    const char *msg = api().GetErrorString ? api().GetErrorString(r) : "?";
    throw Error(std::string("RCCL error in ") + what + ": " + msg);
  }
}

// ---- in-process test transport (see dist.h) --------------------------------------------
constexpr char kLocalTag[] = "POGSLOCAL:";

struct LocalGroup {
  int world = 0;
  bool host_staged = false;   // POGS_AMD_TEST_TRANSPORT=host
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  unsigned long long gen = 0;
  std::vector<std::vector<unsigned char>> slots;   // host-staged form
  // stream-ordered form: one device slot and two events per rank
  std::vector<void *> dslot;
  std::vector<size_t> dcap;
  std::vector<hipEvent_t> ready, done;
  std::vector<int> device;
  std::vector<unsigned long long> calls;   // collectives each rank has taken part in

  ~LocalGroup() {
    for (void *p : dslot) if (p) (void)hipFree(p);
    for (hipEvent_t e : ready) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : done) if (e) (void)hipEventDestroy(e);
  }

  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const unsigned long long g = gen;
    if (++arrived == world) {
      arrived = 0;
      ++gen;
      cv.notify_all();
      return;
    }
    // (the same limit as the RCCL path's, Ctx::wait_publish: POGS_AMD_COLL_TIMEOUT_S, at most 120 s here)
    const char *te = std::getenv("POGS_AMD_COLL_TIMEOUT_S");
    const double lim = std::min(120.0, te && std::atof(te) > 0 ? std::atof(te) : 120.0);
    if (!cv.wait_for(lk, std::chrono::duration<double>(lim), [&] { return gen != g; })) {
      // leave the group consistent for the ranks that did arrive: this collective is void
      --arrived;
      throw Error("local communicator: a rank did not reach the collective within " + std::to_string(static_cast<int>(lim)) + " s");
    }
  }
};

std::shared_ptr<LocalGroup> local_group(const std::string &key, int world, bool host_staged) {
  static std::mutex mu;
  static std::map<std::string, std::weak_ptr<LocalGroup>> groups;
  std::lock_guard<std::mutex> lk(mu);
  std::shared_ptr<LocalGroup> g = groups[key].lock();
  if (!g) {
    g = std::make_shared<LocalGroup>();
    g->world = world;
    g->host_staged = host_staged;
    g->slots.resize(world);
    g->dslot.assign(world, nullptr);
    g->dcap.assign(world, 0);
    g->ready.assign(world, nullptr);
    g->done.assign(world, nullptr);
    g->device.assign(world, -1);
    g->calls.assign(world, 0);
    groups[key] = g;
  }
  POGS_CHECK(g->world == world, "local communicator: ranks disagree on the world size");
  return g;
}

template <typename T>
void local_allreduce_host(LocalGroup &g, int rank, const T *in, T *out, size_t count, hipStream_t stream) {
  const size_t bytes = count * sizeof(T);
  std::vector<unsigned char> &mine = g.slots[rank];
  mine.resize(bytes);
  POGS_HIP_CHECK(hipMemcpyAsync(mine.data(), in, bytes, hipMemcpyDeviceToHost, stream));
  POGS_HIP_CHECK(hipStreamSynchronize(stream));
  g.barrier();
  std::vector<T> sum(count, static_cast<T>(0));
  for (int r = 0; r < g.world; ++r) {   // rank order: every rank forms the identical sum
    POGS_CHECK(g.slots[r].size() == bytes, "local communicator: ranks disagree on the element count");
    const T *p = reinterpret_cast<const T *>(g.slots[r].data());
    for (size_t i = 0; i < count; ++i) sum[i] += p[i];
  }
  g.barrier();   // nobody overwrites a slot that is still being read
  POGS_HIP_CHECK(hipMemcpyAsync(out, sum.data(), bytes, hipMemcpyHostToDevice, stream));
  POGS_HIP_CHECK(hipStreamSynchronize(stream));
}

// out[i] = slot_0[i] + slot_1[i] + ... in rank order (every rank forms the identical sum)
constexpr int kLocalMaxWorld = 16;
struct LocalSlots {
  const void *p[kLocalMaxWorld];
};
template <typename T>
__global__ void __launch_bounds__(256) local_sum_kernel(LocalSlots slots, int world, size_t count, T *out) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < count; i += static_cast<size_t>(gridDim.x) * 256) {
    T v = static_cast<T>(0);
    for (int r = 0; r < world; ++r) v += static_cast<const T *>(slots.p[r])[i];
    out[i] = v;
  }
}

// Stream-ordered exchange (dist.h): no call in here waits for a stream.  Per collective and rank:
//   wait (on the stream) for the peers' `done` of the previous collective -- my slot may still be read;
//   copy in -> own slot, record `ready`
//   ---- host barrier: every `ready` has been RECORDED (not necessarily reached) ----
//   wait (on the stream) for every peer's `ready`; sum kernel over the slots -> out; record `done`
//   ---- host barrier: every `done` has been recorded ----
template <typename T>
void local_allreduce_stream(LocalGroup &g, int rank, const T *in, T *out, size_t count, hipStream_t stream) {
  POGS_CHECK(g.world <= kLocalMaxWorld, "local communicator: at most 16 ranks");
  const size_t bytes = count * sizeof(T);
  int dev = 0;
  POGS_HIP_CHECK(hipGetDevice(&dev));
  if (!g.ready[rank]) {
    POGS_HIP_CHECK(hipEventCreateWithFlags(&g.ready[rank], hipEventDisableTiming));
    POGS_HIP_CHECK(hipEventCreateWithFlags(&g.done[rank], hipEventDisableTiming));
    g.device[rank] = dev;
  }
  if (bytes > g.dcap[rank]) {
    // the slot grows (the first collective of each size): drain the device first -- a peer's sum
    // kernel of the previous collective may still read the old slot
    POGS_HIP_CHECK(hipDeviceSynchronize());
    if (g.dslot[rank]) POGS_HIP_CHECK(hipFree(g.dslot[rank]));
    g.dslot[rank] = nullptr;
    POGS_HIP_CHECK(hipMalloc(&g.dslot[rank], bytes));
    g.dcap[rank] = bytes;
  }
  if (g.calls[rank] > 0) {
    for (int r = 0; r < g.world; ++r)
      if (r != rank) POGS_HIP_CHECK(hipStreamWaitEvent(stream, g.done[r], 0));
  }
  POGS_HIP_CHECK(hipMemcpyAsync(g.dslot[rank], in, bytes, hipMemcpyDeviceToDevice, stream));
  POGS_HIP_CHECK(hipEventRecord(g.ready[rank], stream));
  g.barrier();
  LocalSlots slots;
  for (int r = 0; r < g.world; ++r) {
    POGS_CHECK(g.device[r] == dev, "local communicator (stream-ordered form): all ranks must use one device");
    POGS_CHECK(g.dcap[r] >= bytes, "local communicator: ranks disagree on the element count");
    slots.p[r] = g.dslot[r];
    if (r != rank) POGS_HIP_CHECK(hipStreamWaitEvent(stream, g.ready[r], 0));
  }
  const unsigned grid = static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>(1024, (count + 255) / 256)));
  hipLaunchKernelGGL(local_sum_kernel<T>, dim3(grid), dim3(256), 0, stream, slots, g.world, count, out);
  POGS_HIP_CHECK(hipEventRecord(g.done[rank], stream));
  ++g.calls[rank];
  g.barrier();
}

template <typename T>
void local_allreduce(LocalGroup &g, int rank, const T *in, T *out, size_t count, hipStream_t stream) {
  if (g.host_staged) local_allreduce_host(g, rank, in, out, count, stream);
  else local_allreduce_stream(g, rank, in, out, count, stream);
}

}  // namespace

// pack: out[0..nv) = vec (as double), then the scalar ranges; unpack: the reverse
template <typename T>
__global__ void __launch_bounds__(256) pack_kernel(const T *vec, size_t nv, const double *s1, int n1, const double *s2,
                                                   int n2, double *out) {
  const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < nv) out[i] = static_cast<double>(vec[i]);
  if (blockIdx.x == 0) {
    const int t = threadIdx.x;
    if (t < n1) out[nv + t] = s1[t];
    else if (t < n1 + n2) out[nv + t] = s2[t - n1];
  }
}
template <typename T>
__global__ void __launch_bounds__(256) unpack_kernel(const double *in, size_t nv, T *vec, double *s1, int n1, double *s2,
                                                     int n2) {
  const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < nv) vec[i] = static_cast<T>(in[i]);
  if (blockIdx.x == 0) {
    const int t = threadIdx.x;
    if (t < n1) s1[t] = in[nv + t];
    else if (t < n1 + n2) s2[t - n1] = in[nv + t];
  }
}

DistComm::~DistComm() {
  if (comm_) {
    // (an aborted communicator is already gone; one whose collective may still be stuck is aborted,
    // not destroyed: ncclCommDestroy would wait for the stuck kernel)
    try { if (!aborted_) api().CommDestroy(comm_); } catch (...) {}
  }
  delete static_cast<std::shared_ptr<LocalGroup> *>(local_);
  if (pack_) (void)hipFree(pack_);
}

void DistComm::unique_id(char *out) {
  UniqueId id;
  check(api().GetUniqueId(&id), "ncclGetUniqueId");
  std::memcpy(out, id.internal, kUniqueIdBytes);
}

void DistComm::init(int rank, int world, const char *unique_id) {
  POGS_CHECK(world >= 1 && rank >= 0 && rank < world, "bad rank/world");
  rank_ = rank;
  world_ = world;
  if (std::strncmp(unique_id, kLocalTag, sizeof(kLocalTag) - 1) == 0) {
    const char *tt = std::getenv("POGS_AMD_TEST_TRANSPORT");
    POGS_CHECK(tt && (tt[0] == '1' || tt[0] == 'h'), "the in-process test transport needs POGS_AMD_TEST_TRANSPORT=1");
    const std::string key(unique_id, strnlen(unique_id, kUniqueIdBytes));
    local_ = new std::shared_ptr<LocalGroup>(local_group(key, world, tt[0] == 'h'));
    return;
  }
  UniqueId id;
  std::memcpy(id.internal, unique_id, kUniqueIdBytes);
  check(api().CommInitRank(&comm_, world, id, rank), "ncclCommInitRank");
}

int DistComm::comm_nranks() const {
  if (local_) return world_;
  if (!comm_ || aborted_) return 0;
  int n = 0;
  check(api().CommCount(comm_, &n), "ncclCommCount");
  return n;
}

void DistComm::abort() {
  if (comm_ && !aborted_) {
    aborted_ = true;
    try { (void)api().CommAbort(comm_); } catch (...) {}
  }
}

const char *DistComm::async_error() const {
  if (!comm_ || aborted_) return "";
  int e = 0;
  if (api().CommGetAsyncError(comm_, &e) != 0 || e == 0) return "";
  return api().GetErrorString ? api().GetErrorString(e) : "asynchronous RCCL error";
}

void DistComm::reduce_raw(const void *in, void *out, size_t count, int dtype, hipStream_t stream) const {
  if (count == 0 || !active()) return;
  if (aborted_) throw Error("the RCCL communicator of this handle was aborted (a collective timed out or failed)");
  ++ncoll_;
  if (local_) {
    LocalGroup &g = **static_cast<std::shared_ptr<LocalGroup> *>(local_);
    if (dtype == kNcclFloat) local_allreduce(g, rank_, static_cast<const float *>(in), static_cast<float *>(out), count, stream);
    else local_allreduce(g, rank_, static_cast<const double *>(in), static_cast<double *>(out), count, stream);
    return;
  }
  check(api().AllReduce(in, out, count, dtype, kNcclSum, comm_, stream), "ncclAllReduce");
}

void DistComm::group_begin() const {
  if (comm_ && !aborted_) check(api().GroupStart(), "ncclGroupStart");
}
void DistComm::group_end() const {
  if (comm_ && !aborted_) check(api().GroupEnd(), "ncclGroupEnd");
}

void DistComm::allreduce(float *buf, size_t count, hipStream_t stream) const {
  reduce_raw(buf, buf, count, kNcclFloat, stream);
}
void DistComm::allreduce(double *buf, size_t count, hipStream_t stream) const {
  reduce_raw(buf, buf, count, kNcclDouble, stream);
}
void DistComm::allreduce(const double *in, double *out, size_t count, hipStream_t stream) const {
  reduce_raw(in, out, count, kNcclDouble, stream);
}

template <typename T>
void DistComm::allreduce3(T *buf, size_t count, double *s1, size_t n1, double *s2, size_t n2, hipStream_t stream) {
  if (!active()) return;
  POGS_CHECK(n1 + n2 <= 256, "allreduce3: too many scalars");
  const size_t total = count + n1 + n2;
  if (total > pack_cap_) {
    POGS_HIP_CHECK(hipStreamSynchronize(stream));
    if (pack_) POGS_HIP_CHECK(hipFree(pack_));
    pack_ = nullptr;
    POGS_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&pack_), total * sizeof(double)));
    pack_cap_ = total;
  }
  const unsigned grid = static_cast<unsigned>((count + 255) / 256) + (count == 0 ? 1u : 0u);
  hipLaunchKernelGGL(pack_kernel<T>, dim3(grid), dim3(256), 0, stream, buf, count, s1, static_cast<int>(n1), s2,
                     static_cast<int>(n2), pack_);
  allreduce(pack_, total, stream);
  hipLaunchKernelGGL(unpack_kernel<T>, dim3(grid), dim3(256), 0, stream, pack_, count, buf, s1, static_cast<int>(n1), s2,
                     static_cast<int>(n2));
}
template <typename T>
void DistComm::allreduce2(T *buf, size_t count, double *scalars, size_t nscalars, hipStream_t stream) {
  allreduce3<T>(buf, count, scalars, nscalars, nullptr, 0, stream);
}
template void DistComm::allreduce3<float>(float *, size_t, double *, size_t, double *, size_t, hipStream_t);
template void DistComm::allreduce3<double>(double *, size_t, double *, size_t, double *, size_t, hipStream_t);
template void DistComm::allreduce2<float>(float *, size_t, double *, size_t, hipStream_t);
template void DistComm::allreduce2<double>(double *, size_t, double *, size_t, hipStream_t);

}  // namespace pogs_amd
