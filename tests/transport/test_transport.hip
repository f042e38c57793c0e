// Test transports for the row-sharded solve: a plug-in of libpogs_amd.so (pogs_amd/csrc/transport_plugin.h).
// TEST INFRASTRUCTURE -- not part of the product library, not a performance path.
//
// RCCL refuses two ranks on one device, and the build / test boxes have one GPU.  These communicators let the
// engine's row-sharded decomposition itself be verified there:
//
//   id "POGSLOCAL:<key>"  ranks = THREADS of one process, each with its own solver, on the same GPU.
//       POGS_AMD_TEST_TRANSPORT=1 (default)  stream-ordered, like ncclAllReduce: every rank copies its buffer into a
//           device slot on ITS stream and records an event; the threads meet (host barrier, no stream is waited
//           for); every rank makes its stream wait for the peers' events and sums the slots in rank order with a
//           kernel.  The host never waits for the device, so a kernel that reads the result too early, or
//           overwrites an operand too soon, shows up as a wrong answer.
//       POGS_AMD_TEST_TRANSPORT=host         buffers staged through the host and summed there (hipStreamSynchronize
//           on both sides of the exchange).
//   id "POGSSHM:<key>"    ranks = PROCESSES (one per rank, as under torch.distributed.run) that may share one GPU,
//       joined by a POSIX shared-memory segment: buffers staged through the host, summed in rank order by every
//       rank.  It exists so that everything AROUND the collective of a multi-process run (launcher environment, the
//       128-byte id broadcast, per-rank shards, rank 0 working alone while the others wait) can be rehearsed on one
//       GPU (bench.py, POGS_AMD_BENCH_REHEARSAL=1; tests/test_gpu_bench.py).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../pogs_amd/csrc/transport_plugin.h"

namespace {

thread_local std::string g_error;

struct Fail : std::runtime_error {
  using std::runtime_error::runtime_error;
};
void need(bool ok, const char *what) {
  if (!ok) throw Fail(what);
}
void hip_ok(hipError_t e, const char *what) {
  if (e != hipSuccess) throw Fail(std::string(what) + ": " + hipGetErrorString(e));
}
double timeout_s() {
  const char *te = std::getenv("POGS_AMD_COLL_TIMEOUT_S");
  return std::min(120.0, te && std::atof(te) > 0 ? std::atof(te) : 120.0);
}

struct Transport {
  virtual ~Transport() = default;
  virtual void allreduce(int rank, const void *in, void *out, size_t count, int dtype, hipStream_t stream) = 0;
};

// ---------------------------------------------------------------- ranks = threads of one process
struct LocalGroup {
  int world = 0;
  bool host_staged = false;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  unsigned long long gen = 0;
  std::vector<std::vector<unsigned char>> slots;   // host-staged form
  std::vector<void *> dslot;                       // stream-ordered form: one device slot and two events per rank
  std::vector<size_t> dcap;
  std::vector<hipEvent_t> ready, done;
  std::vector<int> device;
  std::vector<unsigned long long> calls;

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
    const double lim = timeout_s();
    if (!cv.wait_for(lk, std::chrono::duration<double>(lim), [&] { return gen != g; })) {
      --arrived;   // leave the group consistent for the ranks that did arrive: this collective is void
      throw Fail("local communicator: a rank did not reach the collective within " + std::to_string(static_cast<int>(lim)) + " s");
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
  need(g->world == world, "local communicator: ranks disagree on the world size");
  return g;
}

constexpr int kMaxWorld = 16;
struct Slots {
  const void *p[kMaxWorld];
};
// out[i] = slot_0[i] + slot_1[i] + ... in rank order (every rank forms the identical sum)
template <typename T>
__global__ void __launch_bounds__(256) slot_sum_kernel(Slots slots, int world, size_t count, T *out) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < count; i += static_cast<size_t>(gridDim.x) * 256) {
    T v = static_cast<T>(0);
    for (int r = 0; r < world; ++r) v += static_cast<const T *>(slots.p[r])[i];
    out[i] = v;
  }
}

struct LocalTransport : Transport {
  std::shared_ptr<LocalGroup> g;

  template <typename T>
  void host(int rank, const T *in, T *out, size_t count, hipStream_t stream) {
    const size_t bytes = count * sizeof(T);
    std::vector<unsigned char> &mine = g->slots[rank];
    mine.resize(bytes);
    hip_ok(hipMemcpyAsync(mine.data(), in, bytes, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
    hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize");
    g->barrier();
    std::vector<T> sum(count, static_cast<T>(0));
    for (int r = 0; r < g->world; ++r) {
      need(g->slots[r].size() == bytes, "local communicator: ranks disagree on the element count");
      const T *p = reinterpret_cast<const T *>(g->slots[r].data());
      for (size_t i = 0; i < count; ++i) sum[i] += p[i];
    }
    g->barrier();   // nobody overwrites a slot that is still being read
    hip_ok(hipMemcpyAsync(out, sum.data(), bytes, hipMemcpyHostToDevice, stream), "hipMemcpyAsync");
    hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize");
  }

  // Stream-ordered exchange: no call in here waits for a stream.  Per collective and rank:
  //   wait (on the stream) for the peers' `done` of the previous collective -- my slot may still be read;
  //   copy in -> own slot, record `ready`
  //   ---- host barrier: every `ready` has been RECORDED (not necessarily reached) ----
  //   wait (on the stream) for every peer's `ready`; sum kernel over the slots -> out; record `done`
  //   ---- host barrier: every `done` has been recorded ----
  template <typename T>
  void ordered(int rank, const T *in, T *out, size_t count, hipStream_t stream) {
    LocalGroup &G = *g;
    need(G.world <= kMaxWorld, "local communicator: at most 16 ranks");
    const size_t bytes = count * sizeof(T);
    int dev = 0;
    hip_ok(hipGetDevice(&dev), "hipGetDevice");
    if (!G.ready[rank]) {
      hip_ok(hipEventCreateWithFlags(&G.ready[rank], hipEventDisableTiming), "hipEventCreate");
      hip_ok(hipEventCreateWithFlags(&G.done[rank], hipEventDisableTiming), "hipEventCreate");
      G.device[rank] = dev;
    }
    if (bytes > G.dcap[rank]) {
      // the slot grows (the first collective of each size): drain the device first -- a peer's sum
      // kernel of the previous collective may still read the old slot
      hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize");
      if (G.dslot[rank]) hip_ok(hipFree(G.dslot[rank]), "hipFree");
      G.dslot[rank] = nullptr;
      hip_ok(hipMalloc(&G.dslot[rank], bytes), "hipMalloc");
      G.dcap[rank] = bytes;
    }
    if (G.calls[rank] > 0) {
      for (int r = 0; r < G.world; ++r)
        if (r != rank) hip_ok(hipStreamWaitEvent(stream, G.done[r], 0), "hipStreamWaitEvent");
    }
    hip_ok(hipMemcpyAsync(G.dslot[rank], in, bytes, hipMemcpyDeviceToDevice, stream), "hipMemcpyAsync");
    hip_ok(hipEventRecord(G.ready[rank], stream), "hipEventRecord");
    G.barrier();
    Slots slots;
    for (int r = 0; r < G.world; ++r) {
      need(G.device[r] == dev, "local communicator (stream-ordered form): all ranks must use one device");
      need(G.dcap[r] >= bytes, "local communicator: ranks disagree on the element count");
      slots.p[r] = G.dslot[r];
      if (r != rank) hip_ok(hipStreamWaitEvent(stream, G.ready[r], 0), "hipStreamWaitEvent");
    }
    const unsigned grid = static_cast<unsigned>(std::max<size_t>(1, std::min<size_t>(1024, (count + 255) / 256)));
    hipLaunchKernelGGL(slot_sum_kernel<T>, dim3(grid), dim3(256), 0, stream, slots, G.world, count, out);
    hip_ok(hipEventRecord(G.done[rank], stream), "hipEventRecord");
    ++G.calls[rank];
    G.barrier();
  }

  void allreduce(int rank, const void *in, void *out, size_t count, int dtype, hipStream_t stream) override {
    if (dtype == 0) {
      if (g->host_staged) host(rank, static_cast<const float *>(in), static_cast<float *>(out), count, stream);
      else ordered(rank, static_cast<const float *>(in), static_cast<float *>(out), count, stream);
    } else {
      if (g->host_staged) host(rank, static_cast<const double *>(in), static_cast<double *>(out), count, stream);
      else ordered(rank, static_cast<const double *>(in), static_cast<double *>(out), count, stream);
    }
  }
};

// ---------------------------------------------------------------- ranks = processes, joined by shared memory
// Segment: a header (sense-reversing barrier words, zero-filled by the kernel at creation) and one slot of
// kShmSlotBytes per rank; longer buffers travel in chunks.  Every rank opens (O_CREAT) and sizes the segment
// -- both idempotent -- and rank 0 unlinks the name once all ranks are attached, so nothing outlives the run.
constexpr size_t kShmSlotBytes = size_t(32) << 20;
struct ShmHeader {
  std::atomic<unsigned> arrived;
  std::atomic<unsigned> gen;
  std::atomic<unsigned> world;     // 0 until the first rank writes it; every rank checks it
  std::atomic<unsigned> failed;    // a rank gave up: the others stop waiting
  unsigned char pad[48];
};
static_assert(sizeof(ShmHeader) == 64, "one cache line");

struct ShmTransport : Transport {
  int rank = 0, world = 0;
  std::string name;
  unsigned char *base = nullptr;
  size_t bytes = 0;
  std::vector<unsigned char> sum;

  ShmHeader *hdr() const { return reinterpret_cast<ShmHeader *>(base); }
  unsigned char *slot(int r) const { return base + sizeof(ShmHeader) + static_cast<size_t>(r) * kShmSlotBytes; }

  ~ShmTransport() override {
    if (base) munmap(base, bytes);
    if (rank == 0 && !name.empty()) shm_unlink(name.c_str());   // (already gone after a complete open)
  }

  void barrier() {
    ShmHeader *h = hdr();
    const unsigned g = h->gen.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == static_cast<unsigned>(world)) {
      h->arrived.store(0, std::memory_order_relaxed);
      h->gen.store(g + 1, std::memory_order_release);
      return;
    }
    const double lim = timeout_s();
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (h->gen.load(std::memory_order_acquire) == g) {
      if (h->failed.load(std::memory_order_relaxed)) throw Fail("shared-memory communicator: a peer rank failed");
      if (++spins < 2000) {
        sched_yield();
      } else {
        timespec ts{0, 50000};
        nanosleep(&ts, nullptr);
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > lim) {
          h->failed.store(1, std::memory_order_relaxed);
          throw Fail("shared-memory communicator: a rank did not reach the collective within " +
                     std::to_string(static_cast<int>(lim)) + " s");
        }
      }
    }
  }

  void open(const std::string &key, int r, int w) {
    rank = r;
    world = w;
    need(w >= 1 && w <= kMaxWorld, "shared-memory communicator: 1 to 16 ranks");
    name = "/pogs_amd_";
    for (char c : key) name.push_back((c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') ? c : '_');
    bytes = sizeof(ShmHeader) + static_cast<size_t>(w) * kShmSlotBytes;
    const int fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
    need(fd >= 0, "shm_open failed");
    if (ftruncate(fd, static_cast<off_t>(bytes)) != 0) {
      ::close(fd);
      throw Fail("ftruncate of the shared segment failed");
    }
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    ::close(fd);
    need(p != MAP_FAILED, "mmap of the shared segment failed");
    base = static_cast<unsigned char *>(p);
    unsigned expect = 0;
    if (!hdr()->world.compare_exchange_strong(expect, static_cast<unsigned>(w)))
      need(expect == static_cast<unsigned>(w), "shared-memory communicator: ranks disagree on the world size");
    barrier();                       // every rank is attached ...
    if (rank == 0) shm_unlink(name.c_str());   // ... so the name can go: the mappings stay
  }

  template <typename T>
  void run(const T *in, T *out, size_t count, hipStream_t stream) {
    const size_t per = kShmSlotBytes / sizeof(T);
    for (size_t off = 0; off < count; off += per) {
      const size_t c = std::min(per, count - off), nb = c * sizeof(T);
      hip_ok(hipMemcpyAsync(slot(rank), in + off, nb, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
      hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize");
      barrier();
      sum.resize(nb);
      T *s = reinterpret_cast<T *>(sum.data());
      std::memcpy(s, slot(0), nb);
      for (int r = 1; r < world; ++r) {   // rank order: every rank forms the identical sum
        const T *p = reinterpret_cast<const T *>(slot(r));
        for (size_t i = 0; i < c; ++i) s[i] += p[i];
      }
      barrier();   // nobody overwrites a slot that is still being read
      hip_ok(hipMemcpyAsync(out + off, s, nb, hipMemcpyHostToDevice, stream), "hipMemcpyAsync");
      hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize");
    }
  }

  void allreduce(int, const void *in, void *out, size_t count, int dtype, hipStream_t stream) override {
    if (dtype == 0) run(static_cast<const float *>(in), static_cast<float *>(out), count, stream);
    else run(static_cast<const double *>(in), static_cast<double *>(out), count, stream);
  }
};

constexpr char kLocalTag[] = "POGSLOCAL:";
constexpr char kShmTag[] = "POGSSHM:";

void *tp_open(const char *unique_id, int rank, int world) {
  try {
    const std::string key(unique_id, strnlen(unique_id, 128));
    const char *tt = std::getenv("POGS_AMD_TEST_TRANSPORT");
    if (key.compare(0, sizeof(kLocalTag) - 1, kLocalTag) == 0) {
      auto t = std::make_unique<LocalTransport>();
      t->g = local_group(key, world, tt && tt[0] == 'h');
      return t.release();
    }
    if (key.compare(0, sizeof(kShmTag) - 1, kShmTag) == 0) {
      auto t = std::make_unique<ShmTransport>();
      t->open(key.substr(sizeof(kShmTag) - 1), rank, world);
      return t.release();
    }
    throw Fail("test transport: the unique id names neither POGSLOCAL: nor POGSSHM:");
  } catch (const std::exception &e) {
    g_error = e.what();
    return nullptr;
  }
}

int tp_allreduce(void *h, int rank, const void *in, void *out, size_t count, int dtype, void *stream) {
  try {
    static_cast<Transport *>(h)->allreduce(rank, in, out, count, dtype, static_cast<hipStream_t>(stream));
    return 0;
  } catch (const std::exception &e) {
    g_error = e.what();
    return 1;
  }
}

void tp_close(void *h) { delete static_cast<Transport *>(h); }
const char *tp_last_error() { return g_error.c_str(); }

const PogsAmdTransportApi kApi = {POGS_AMD_TRANSPORT_ABI, tp_open, tp_allreduce, tp_close, tp_last_error};

}  // namespace

extern "C" const PogsAmdTransportApi *pogs_amd_transport(void) { return &kApi; }
