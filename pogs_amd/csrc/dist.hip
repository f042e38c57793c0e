// RCCL binding (see dist.h).
#include "dist.h"

#include <dlfcn.h>

#include <cstring>
#include <mutex>

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

}  // namespace

DistComm::~DistComm() {
  if (comm_) {
    try { api().CommDestroy(comm_); } catch (...) {}
  }
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
  UniqueId id;
  std::memcpy(id.internal, unique_id, kUniqueIdBytes);
  check(api().CommInitRank(&comm_, world, id, rank), "ncclCommInitRank");
}

void DistComm::reduce_raw(void *buf, size_t count, int dtype, hipStream_t stream) const {
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
  if (!comm_) return;
  check(api().GroupStart(), "ncclGroupStart");
  allreduce(buf, count, stream);
  allreduce(scalars, nscalars, stream);
  check(api().GroupEnd(), "ncclGroupEnd");
}
template void DistComm::allreduce2<float>(float *, size_t, double *, size_t, hipStream_t) const;
template void DistComm::allreduce2<double>(double *, size_t, double *, size_t, hipStream_t) const;

}  // namespace pogs_amd
