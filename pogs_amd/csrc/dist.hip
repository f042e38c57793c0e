// RCCL binding (see dist.h).
#include "dist.h"

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "common.h"
#include "transport_plugin.h"

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

// ---- transport plug-in (transport_plugin.h): ids that start with "POGS" are not RCCL's ----------------
constexpr char kPluginTag[] = "POGS";

const PogsAmdTransportApi *plugin_api() {
  static const PogsAmdTransportApi *table = nullptr;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (table) return table;
  const char *path = std::getenv("POGS_AMD_TRANSPORT_PLUGIN");
  POGS_CHECK(path && path[0], "a unique id that starts with \"POGS\" names a transport plug-in: set POGS_AMD_TRANSPORT_PLUGIN "
                              "to its shared object (the test-suite's is tests/transport/libpogs_test_transport.so)");
  void *lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!lib) throw Error(std::string("cannot load the transport plug-in: ") + dlerror());
  auto entry = reinterpret_cast<PogsAmdTransportEntry>(dlsym(lib, "pogs_amd_transport"));
  POGS_CHECK(entry != nullptr, "the transport plug-in lacks the symbol pogs_amd_transport");
  const PogsAmdTransportApi *t = entry();
  POGS_CHECK(t && t->abi == POGS_AMD_TRANSPORT_ABI && t->open && t->allreduce && t->close && t->last_error,
             "the transport plug-in speaks another interface version");
  table = t;
  return table;
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
  if (local_ && plug_) static_cast<const PogsAmdTransportApi *>(plug_)->close(local_);
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
  if (std::strncmp(unique_id, kPluginTag, sizeof(kPluginTag) - 1) == 0) {
    const PogsAmdTransportApi *t = plugin_api();
    local_ = t->open(unique_id, rank, world);
    if (!local_) throw Error(std::string("transport plug-in: ") + t->last_error());
    plug_ = t;
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
    const PogsAmdTransportApi *t = static_cast<const PogsAmdTransportApi *>(plug_);
    if (t->allreduce(local_, rank_, in, out, count, dtype == kNcclFloat ? 0 : 1, stream) != 0)
      throw Error(std::string("transport plug-in: ") + t->last_error());
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
