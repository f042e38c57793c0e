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
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  unsigned long long gen = 0;
  std::vector<std::vector<unsigned char>> slots;

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

std::shared_ptr<LocalGroup> local_group(const std::string &key, int world) {
  static std::mutex mu;
  static std::map<std::string, std::weak_ptr<LocalGroup>> groups;
  std::lock_guard<std::mutex> lk(mu);
  std::shared_ptr<LocalGroup> g = groups[key].lock();
  if (!g) {
    g = std::make_shared<LocalGroup>();
    g->world = world;
    g->slots.resize(world);
    groups[key] = g;
  }
  POGS_CHECK(g->world == world, "local communicator: ranks disagree on the world size");
  return g;
}

template <typename T>
void local_allreduce(LocalGroup &g, int rank, T *buf, size_t count, hipStream_t stream) {
  const size_t bytes = count * sizeof(T);
  std::vector<unsigned char> &mine = g.slots[rank];
  mine.resize(bytes);
  POGS_HIP_CHECK(hipMemcpyAsync(mine.data(), buf, bytes, hipMemcpyDeviceToHost, stream));
  POGS_HIP_CHECK(hipStreamSynchronize(stream));
  g.barrier();
  std::vector<T> sum(count, static_cast<T>(0));
  for (int r = 0; r < g.world; ++r) {   // rank order: every rank forms the identical sum
    POGS_CHECK(g.slots[r].size() == bytes, "local communicator: ranks disagree on the element count");
    const T *p = reinterpret_cast<const T *>(g.slots[r].data());
    for (size_t i = 0; i < count; ++i) sum[i] += p[i];
  }
  g.barrier();   // nobody overwrites a slot that is still being read
  POGS_HIP_CHECK(hipMemcpyAsync(buf, sum.data(), bytes, hipMemcpyHostToDevice, stream));
  POGS_HIP_CHECK(hipStreamSynchronize(stream));
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
    POGS_CHECK(tt && tt[0] == '1', "the in-process test transport needs POGS_AMD_TEST_TRANSPORT=1");
    const std::string key(unique_id, strnlen(unique_id, kUniqueIdBytes));
    local_ = new std::shared_ptr<LocalGroup>(local_group(key, world));
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

void DistComm::reduce_raw(void *buf, size_t count, int dtype, hipStream_t stream) const {
  if (count == 0 || !active()) return;
  if (aborted_) throw Error("the RCCL communicator of this handle was aborted (a collective timed out or failed)");
  ++ncoll_;
  if (local_) {
    LocalGroup &g = **static_cast<std::shared_ptr<LocalGroup> *>(local_);
    if (dtype == kNcclFloat) local_allreduce(g, rank_, static_cast<float *>(buf), count, stream);
    else local_allreduce(g, rank_, static_cast<double *>(buf), count, stream);
    return;
  }
  check(api().AllReduce(buf, buf, count, dtype, kNcclSum, comm_, stream), "ncclAllReduce");
}

void DistComm::allreduce(float *buf, size_t count, hipStream_t stream) const {
  reduce_raw(buf, count, kNcclFloat, stream);
}
void DistComm::allreduce(double *buf, size_t count, hipStream_t stream) const {
  reduce_raw(buf, count, kNcclDouble, stream);
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
