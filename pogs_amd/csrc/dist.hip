// RCCL binding (see dist.h).
#include "dist.h"

#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
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
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
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
    if (!cv.wait_for(lk, std::chrono::seconds(120), [&] { return gen != g; }))
      throw Error("local communicator: a rank did not reach the collective within 120 s");
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

DistComm::~DistComm() {
  if (comm_) {
    try { api().CommDestroy(comm_); } catch (...) {}
  }
  delete static_cast<std::shared_ptr<LocalGroup> *>(local_);
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
    const std::string key(unique_id, strnlen(unique_id, kUniqueIdBytes));
    local_ = new std::shared_ptr<LocalGroup>(local_group(key, world));
    return;
  }
  UniqueId id;
  std::memcpy(id.internal, unique_id, kUniqueIdBytes);
  check(api().CommInitRank(&comm_, world, id, rank), "ncclCommInitRank");
}

void DistComm::reduce_raw(void *buf, size_t count, int dtype, hipStream_t stream) const {
  if (local_ && count > 0) {
    LocalGroup &g = **static_cast<std::shared_ptr<LocalGroup> *>(local_);
    if (dtype == kNcclFloat) local_allreduce(g, rank_, static_cast<float *>(buf), count, stream);
    else local_allreduce(g, rank_, static_cast<double *>(buf), count, stream);
    return;
  }
  if (!comm_ || count == 0) return;
  check(api().AllReduce(buf, buf, count, dtype, kNcclSum, comm_, stream), "ncclAllReduce");
}

void DistComm::allreduce(float *buf, size_t count, hipStream_t stream) const {
  reduce_raw(buf, count, kNcclFloat, stream);
}
void DistComm::allreduce(double *buf, size_t count, hipStream_t stream) const {
  reduce_raw(buf, count, kNcclDouble, stream);
}

template <typename T>
void DistComm::allreduce2(T *buf, size_t count, double *scalars, size_t nscalars, hipStream_t stream) const {
  if (local_) {
    allreduce(buf, count, stream);
    allreduce(scalars, nscalars, stream);
    return;
  }
  if (!comm_) return;
  check(api().GroupStart(), "ncclGroupStart");
  allreduce(buf, count, stream);
  allreduce(scalars, nscalars, stream);
  check(api().GroupEnd(), "ncclGroupEnd");
}
template <typename T>
void DistComm::allreduce3(T *buf, size_t count, double *s1, size_t n1, double *s2, size_t n2,
                          hipStream_t stream) const {
  if (local_) {
    allreduce(buf, count, stream);
    allreduce(s1, n1, stream);
    allreduce(s2, n2, stream);
    return;
  }
  if (!comm_) return;
  check(api().GroupStart(), "ncclGroupStart");
  allreduce(buf, count, stream);
  allreduce(s1, n1, stream);
  allreduce(s2, n2, stream);
  check(api().GroupEnd(), "ncclGroupEnd");
}
template void DistComm::allreduce3<float>(float *, size_t, double *, size_t, double *, size_t, hipStream_t) const;
template void DistComm::allreduce3<double>(double *, size_t, double *, size_t, double *, size_t, hipStream_t) const;
template void DistComm::allreduce2<float>(float *, size_t, double *, size_t, hipStream_t) const;
template void DistComm::allreduce2<double>(double *, size_t, double *, size_t, hipStream_t) const;

}  // namespace pogs_amd
