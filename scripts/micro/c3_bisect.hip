// Bisection between the library's two row-streaming skeletons at C3's shape (200000 x 5000 fp32, 256 threads x 5
// vectors): Sinkhorn-Knopp's pass (stream_rows_kernel, 0.58 ms) and the one-pass iteration kernel
// (stream_rows2_kernel, 0.63-0.65 ms).  Timing only: the kernels are the library's own templates (csrc/stream.h,
// csrc/ops.h), instantiated here with functors that morph one into the other one element at a time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I pogs_amd/csrc -I scripts/micro scripts/micro/c3_bisect.hip -o scripts/micro/bin/c3_bisect
//   scripts/micro/bin/c3_bisect [m n reps]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "ops.h"
#include "stream.h"
#include "stream_experiments.h"   // the prefetching and the functor-wavefront forms: measured, not shipped

using namespace pogs_amd;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// Sinkhorn-Knopp's row functor in the one-pass kernel's interface (prefetch + row with dot / accumulator arrays)
template <typename T>
struct Sk2Op {
  static constexpr int NS = 0;
  struct Pre {};
  T nn, c;
  T *d;
  __device__ __forceinline__ Pre prefetch(int) const { return Pre{}; }
  template <int N, int ND, int NA>
  __device__ __forceinline__ void row(int i, const Pre &, const T (&dot)[ND], double (&)[N], T (&u)[NA]) const {
    const T v = nn / (dot[0] + c);
    d[i] = v;
    u[0] = v;
    if constexpr (NA > 1) u[1] = ND > 1 ? dot[1] : v;
  }
  struct Post { T v; };
  __device__ __forceinline__ void commit(int i, const Post &q) const { d[i] = q.v; }
  template <int N, int ND, int NA>
  __device__ __forceinline__ Post compute(const Pre &, const T (&dot)[ND], double (&)[N], T (&u)[NA]) const {
    const T v = nn / (dot[0] + c);
    u[0] = v;
    if constexpr (NA > 1) u[1] = ND > 1 ? dot[1] : v;
    return Post{v};
  }
  template <int NA>
  __device__ __forceinline__ void uonly(int, T (&)[NA]) const {}
};
// ... with the six fp64 scalar sums of the iteration's functor and nothing else of it
template <typename T>
struct Sk2SumsOp {
  static constexpr int NS = 6;
  struct Pre {};
  T nn, c;
  T *d;
  __device__ __forceinline__ Pre prefetch(int) const { return Pre{}; }
  template <int N, int ND, int NA>
  __device__ __forceinline__ void row(int i, const Pre &, const T (&dot)[ND], double (&s)[N], T (&u)[NA]) const {
    const T v = nn / (dot[0] + c);
    d[i] = v;
#pragma unroll
    for (int k = 0; k < 6; ++k) s[k] += static_cast<double>(v) * (dot[0] + static_cast<T>(k));
    u[0] = v;
    if constexpr (NA > 1) u[1] = ND > 1 ? dot[1] : v;
  }
  template <int NA>
  __device__ __forceinline__ void uonly(int, T (&)[NA]) const {}
};
// the iteration's functor in the single-dot kernel's interface (operands loaded inside row())
template <typename T, bool LOGISTIC>
struct Fused1Op {
  static constexpr int NS = 6;
  FusedIterOp<T, LOGISTIC> f;
  template <int N>
  __device__ __forceinline__ T row(int i, T dot, double (&s)[N]) const {
    const auto pre = f.prefetch(i);
    T dots[1] = {dot};
    T u[1];
    f.template row<N, 1, 1>(i, pre, dots, s, u);
    return u[0];
  }
  __device__ __forceinline__ T u(int) const { return 0; }
};

struct Timer {
  hipEvent_t a, b;
  Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
  template <typename F>
  double run(F &&launch, int reps) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    std::vector<float> ms(reps);
    for (int i = 0; i < reps; ++i) {
      CK(hipEventRecord(a));
      launch();
      CK(hipEventRecord(b));
      CK(hipEventSynchronize(b));
      CK(hipEventElapsedTime(&ms[i], a, b));
    }
    std::sort(ms.begin(), ms.end());
    return ms[reps / 2];
  }
};

template <typename K>
int regs_of(K kernel) {
  hipFuncAttributes at;
  CK(hipFuncGetAttributes(&at, reinterpret_cast<const void *>(kernel)));
  return at.numRegs;
}

template <int NV, int TPB = 256>
int run(int m, int n, int reps, int table) {
  using T = float;
  const int n_pad = (n + 3) / 4 * 4;
  if (n_pad > TPB * NV * 4) { printf("n too wide for %d x %d\n", TPB, NV); return 1; }
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("device %s, %d CUs; A = %d x %d fp32 (%.2f GB), 256 threads x %d vectors, median of %d launches each\n", prop.name, ncu, m, n,
         4.0 * m * n_pad / 1e9, NV, reps);

  T *A, *xv, *x1, *vec[12], *cp0, *cp1;
  int *h;
  double *sp;
  const size_t gmax = static_cast<size_t>(ncu) * 4;
  CK(hipMalloc(&A, sizeof(T) * static_cast<size_t>(m) * n_pad));
  CK(hipMalloc(&xv, sizeof(T) * n_pad));
  CK(hipMalloc(&x1, sizeof(T) * n_pad));
  for (auto &p : vec) CK(hipMalloc(&p, sizeof(T) * m));
  CK(hipMalloc(&h, sizeof(int) * m));
  CK(hipMalloc(&cp0, sizeof(T) * gmax * n_pad));
  CK(hipMalloc(&cp1, sizeof(T) * gmax * n_pad));
  CK(hipMalloc(&sp, sizeof(double) * gmax * 8));
  {
    std::vector<T> host(static_cast<size_t>(m) * n_pad);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (static_cast<int>(s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
    for (auto &v : host) v = rnd() * 0.05f;
    CK(hipMemcpy(A, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    std::vector<T> v1(std::max(m, n_pad));
    for (auto &v : v1) v = rnd();
    CK(hipMemcpy(xv, v1.data(), n_pad * sizeof(T), hipMemcpyHostToDevice));
    CK(hipMemcpy(x1, v1.data(), n_pad * sizeof(T), hipMemcpyHostToDevice));
    for (int k = 0; k < 12; ++k) {
      for (int i = 0; i < m; ++i) v1[i] = (k == 5 || k == 7) ? 1.0f : (k == 6 || k == 8 || k == 9) ? 0.0f : rnd();   // a = c = 1, b = d = e = 0
      CK(hipMemcpy(vec[k], v1.data(), m * sizeof(T), hipMemcpyHostToDevice));
    }
    std::vector<int> hh(m, 0);   // kAbs: a cheap prox
    CK(hipMemcpy(h, hh.data(), m * sizeof(int), hipMemcpyHostToDevice));
  }
  // vec: 0 ynew, 1 ycur, 2 y12, 3 ytemp, 4 d (SK), 5 a, 6 b, 7 c, 8 d, 9 e, 10 y12s, 11 ytemps
  const FnView<T> fv{h, vec[5], vec[6], vec[7], vec[8], vec[9]};
  const FusedIterOp<T, false> fcheap{vec[0], vec[1], vec[2], vec[3], fv, 1.0f, 1.7f, 1.0f, vec[10], vec[11]};
  const FusedIterOp<T, true> flog{vec[0], vec[1], vec[2], vec[3], fv, 1.0f, 1.7f, 1.0f, vec[10], vec[11]};
  const SkRowOp<T> sk1{static_cast<T>(n), 1e-4f, vec[4]};
  const Sk2Op<T> sk2{static_cast<T>(n), 1e-4f, vec[4]};
  const Sk2SumsOp<T> sk2s{static_cast<T>(n), 1e-4f, vec[4]};

  StreamArgs<T> a1{};
  a1.A = A; a1.lda = n_pad; a1.m = m; a1.n_pad = n_pad; a1.xin = xv; a1.xin_add = nullptr; a1.xin_nrm2 = nullptr;
  a1.col_partials = cp0; a1.scalar_partials = sp;
  StreamArgs2<T> a2{A, static_cast<size_t>(n_pad), m, n_pad, xv, x1, cp0, cp1, sp};
  const size_t lds2 = static_cast<size_t>(n_pad) * sizeof(T);
  using F1C = Fused1Op<T, false>;
  using F1L = Fused1Op<T, true>;
  using FIC = FusedIterOp<T, false>;
  using FIL = FusedIterOp<T, true>;
  Timer tm;
  const double gb = 4.0 * m * n_pad / 1e9;
  auto report = [&](const char *name, int grid, int regs, double ms) {
    printf("%-86s grid %4d (%d/CU)  %3d VGPR  %.4f ms  %.0f GB/s\n", name, grid, grid / ncu, regs, ms, gb / ms * 1e3);
    fflush(stdout);
  };

#define ROWS(NAME, R, DOT, ACC, SQ, OPT, OP, GRID)                                                              \
  {                                                                                                             \
    auto k = stream_rows_kernel<T, TPB, NV, R, DOT, ACC, SQ, kFull, OPT>;                                       \
    const int g = (GRID);                                                                                       \
    report(NAME, g, regs_of(k), tm.run([&] { hipLaunchKernelGGL(k, dim3(g), dim3(TPB), 0, 0, a1, OP); }, reps)); \
  }
#define ROWS2(NAME, R, ND, NA, OPT, OP, GRID)                                                                    \
  {                                                                                                             \
    auto k = stream_rows2_kernel<T, TPB, NV, R, ND, NA, OPT>;                                                    \
    const int g = (GRID);                                                                                       \
    const size_t l = (ND > 1) ? lds2 : 0;                                                                       \
    report(NAME, g, regs_of(k), tm.run([&] { hipLaunchKernelGGL(k, dim3(g), dim3(TPB), l, 0, a2, OP); }, reps)); \
  }
#define ROWS2DB(NAME, R, ND, NA, BPC, OPT, OP, GRID)                                                             \
  {                                                                                                             \
    auto k = stream_rows2_db_kernel<T, TPB, NV, R, ND, NA, BPC, OPT>;                                            \
    const int g = (GRID);                                                                                       \
    const size_t l = (ND > 1) ? lds2 : 0;                                                                       \
    report(NAME, g, regs_of(k), tm.run([&] { hipLaunchKernelGGL(k, dim3(g), dim3(TPB), l, 0, a2, OP); }, reps)); \
  }

#define ROWS2FW(NAME, R, ND, NA, BPC, OPT, OP, GRID)                                                            \
  {                                                                                                             \
    auto k = stream_rows2_fw_kernel<T, TPB, NV, R, ND, NA, BPC, OPT>;                                            \
    const int g = (GRID);                                                                                       \
    const size_t l = stream2_fw_lds<T>(n_pad, TPB, NV, R, ND);                                                  \
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(l))); \
    int occ_ = 0;                                                                                               \
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_, k, TPB + kFwThreads, l));                            \
    printf("   (runtime's occupancy for the next line: %d workgroups per CU, %zu B of dynamic LDS)\n", occ_, l);  \
    report(NAME, g, regs_of(k), tm.run([&] { hipLaunchKernelGGL(k, dim3(g), dim3(TPB + kFwThreads), l, 0, a2, OP); }, reps)); \
  }

  for (int round = 0; round < 2; ++round) {
    printf("---- round %d\n", round);
    if constexpr (TPB == 512) {
      // table 8: C3's rows on 512 threads x 3 vectors (1536 slots for 1250 float4 columns: 19 % of the lanes idle, but
      // half the registers per thread, so four rows per step -- or more workgroups -- fit)
#define FAST8(NAME, R, BPC, X1, ALU, OPT, OP, GRID)                                                              \
  {                                                                                                             \
    auto k = stream_rows2_fast_kernel<T, TPB, NV, R, 2, 2, BPC, X1, ALU, OPT>;                                   \
    const int g = (GRID);                                                                                       \
    report(NAME, g, regs_of(k), tm.run([&] { hipLaunchKernelGGL(k, dim3(g), dim3(TPB), lds2, 0, a2, OP); }, reps)); \
  }
#define DBF8(NAME, R, BPC, OPT, OP, GRID)                                                                         \
  {                                                                                                             \
    auto k = stream_rows2_db_kernel<T, TPB, NV, R, 2, 2, BPC, OPT, true, true>;                                  \
    const int g = (GRID);                                                                                       \
    report(NAME, g, regs_of(k), tm.run([&] { hipLaunchKernelGGL(k, dim3(g), dim3(TPB), lds2, 0, a2, OP); }, reps)); \
  }
      FAST8("512x3 R2 x1 LDS  LDS sums Sk2Op    2/CU", 2, 2, false, false, Sk2Op<T>, sk2, 2 * ncu);
      FAST8("512x3 R4 x1 LDS  LDS sums Sk2Op    2/CU", 4, 2, false, false, Sk2Op<T>, sk2, 2 * ncu);
      FAST8("512x3 R2 x1 LDS  LDS sums logistic 2/CU", 2, 2, false, false, FIL, flog, 2 * ncu);
      FAST8("512x3 R4 x1 LDS  LDS sums logistic 2/CU", 4, 2, false, false, FIL, flog, 2 * ncu);
      FAST8("512x3 R4 x1 regs ALU sums logistic 2/CU", 4, 2, true, true, FIL, flog, 2 * ncu);
      FAST8("512x3 R4 x1 regs ALU sums cheap    2/CU", 4, 2, true, true, FIC, fcheap, 2 * ncu);
      FAST8("512x3 R4 x1 regs ALU sums Sk2Op    2/CU", 4, 2, true, true, Sk2Op<T>, sk2, 2 * ncu);
      FAST8("512x3 R3 x1 LDS  LDS sums logistic 2/CU", 3, 2, false, false, FIL, flog, 2 * ncu);
      FAST8("512x3 R3 x1 regs ALU sums logistic 2/CU", 3, 2, true, true, FIL, flog, 2 * ncu);
      FAST8("512x3 R2 x1 regs ALU sums logistic 2/CU", 2, 2, true, true, FIL, flog, 2 * ncu);
      FAST8("512x3 R2 x1 LDS  ALU sums logistic 3/CU", 2, 3, false, true, FIL, flog, 3 * ncu);
      FAST8("512x3 R8 x1 regs ALU sums logistic 1/CU", 8, 1, true, true, FIL, flog, 1 * ncu);
      FAST8("512x3 R6 x1 regs ALU sums logistic 1/CU", 6, 1, true, true, FIL, flog, 1 * ncu);
      DBF8("512x3 db R2+next x1 regs ALU sums logistic 2/CU", 2, 2, FIL, flog, 2 * ncu);
      DBF8("512x3 db R4+next x1 regs ALU sums logistic 1/CU", 4, 1, FIL, flog, 1 * ncu);
      DBF8("512x3 db R4+next x1 regs ALU sums Sk2Op    1/CU", 4, 1, Sk2Op<T>, sk2, 1 * ncu);
    } else if constexpr (NV == 5) {
      if (table == 7) {
        // table 7: the dots phase without its LDS latency chains (second dot vector in registers, ALU wavefront sums)
#define FAST(NAME, R, BPC, X1, ALU, OPT, OP, GRID)                                                               \
  {                                                                                                             \
    auto k = stream_rows2_fast_kernel<T, TPB, NV, R, 2, 2, BPC, X1, ALU, OPT>;                                   \
    const int g = (GRID);                                                                                       \
    report(NAME, g, regs_of(k), tm.run([&] { hipLaunchKernelGGL(k, dim3(g), dim3(TPB), lds2, 0, a2, OP); }, reps)); \
  }
        ROWS2("H2 rows2 Sk2Op    (two per CU)", 2, 2, 2, Sk2Op<T>, sk2, 2 * ncu);
        ROWS2("K2 rows2 cheap    (two per CU)", 2, 2, 2, FIC, fcheap, 2 * ncu);
        ROWS2("L2 rows2 logistic (two per CU)", 2, 2, 2, FIL, flog, 2 * ncu);
        ROWS2("L3 rows2 logistic (three per CU: shipped)", 2, 2, 2, FIL, flog, 3 * ncu);
        FAST("f  x1 in LDS, ALU sums, logistic, 2/CU", 2, 2, false, true, FIL, flog, 2 * ncu);
        FAST("f  x1 in regs, LDS sums, logistic, 2/CU", 2, 2, true, false, FIL, flog, 2 * ncu);
        FAST("F2 x1 in regs, ALU sums, logistic, 2/CU", 2, 2, true, true, FIL, flog, 2 * ncu);
        FAST("F2 x1 in regs, ALU sums, cheap,    2/CU", 2, 2, true, true, FIC, fcheap, 2 * ncu);
        FAST("F2 x1 in regs, ALU sums, Sk2Op,    2/CU", 2, 2, true, true, Sk2Op<T>, sk2, 2 * ncu);
        FAST("F3 x1 in LDS, ALU sums, logistic, 3/CU", 2, 3, false, true, FIL, flog, 3 * ncu);
#define DBF(NAME, R, BPC, OPT, OP, GRID)                                                                          \
  {                                                                                                             \
    auto k = stream_rows2_db_kernel<T, TPB, NV, R, 2, 2, BPC, OPT, true, true>;                                  \
    const int g = (GRID);                                                                                       \
    report(NAME, g, regs_of(k), tm.run([&] { hipLaunchKernelGGL(k, dim3(g), dim3(TPB), lds2, 0, a2, OP); }, reps)); \
  }
        DBF("D2 db R2+next, x1 in regs, ALU sums, logistic, 2/CU", 2, 2, FIL, flog, 2 * ncu);
        DBF("D2 db R2+next, x1 in regs, ALU sums, cheap,    2/CU", 2, 2, FIC, fcheap, 2 * ncu);
#define DBD(NAME, R, BPC, OPT, OP, GRID)                                                                          \
  {                                                                                                             \
    auto k = stream_rows2_db_kernel<T, TPB, NV, R, 2, 2, BPC, OPT, true, true, true>;                            \
    const int g = (GRID);                                                                                       \
    report(NAME, g, regs_of(k), tm.run([&] { hipLaunchKernelGGL(k, dim3(g), dim3(TPB), lds2, 0, a2, OP); }, reps)); \
  }
        // (late stores: the functor's results kept in registers and written one step late, after the next tile's wait --
        //  the in-order vmcnt counter then never waits for a store's acknowledgement.  With FusedIterOp split into
        //  compute() + commit() the logistic form measured 0.629 against 0.623 and the cheap one 0.640 against 0.648:
        //  nothing, the split was not kept in ops.h; the Sinkhorn-Knopp functor below carries it for the record.)
        DBD("E2 db R2+next, late stores,            Sk2Op,    2/CU", 2, 2, Sk2Op<T>, sk2, 2 * ncu);
#define DBL(NAME, LATE_, OPT, OP)                                                                                 \
  {                                                                                                             \
    auto k = stream_rows2_db_kernel<T, TPB, NV, 2, 2, 2, 2, OPT, true, true, false, LATE_>;                      \
    report(NAME, 2 * ncu, regs_of(k), tm.run([&] { hipLaunchKernelGGL(k, dim3(2 * ncu), dim3(TPB), lds2, 0, a2, OP); }, reps)); \
  }
        DBL("G2 db R2+next requested after the dots, logistic, 2/CU", 1, FIL, flog);
        DBL("G2 db R2+next requested after the dots, cheap,    2/CU", 1, FIC, fcheap);
        DBL("G2 db R2+next requested after the dots, Sk2Op,    2/CU", 1, Sk2Op<T>, sk2);
        DBL("g2 db R2+next one row early, one after the dots, logistic, 2/CU", 2, FIL, flog);
        DBL("g2 db R2+next one row early, one after the dots, cheap,    2/CU", 2, FIC, fcheap);
        DBL("g2 db R2+next one row early, one after the dots, Sk2Op,    2/CU", 2, Sk2Op<T>, sk2);
        DBF("D2 db R3+next, x1 in regs, ALU sums, logistic, 2/CU", 3, 2, FIL, flog, 2 * ncu);
        DBF("D2 db R3+next, x1 in regs, ALU sums, cheap,    2/CU", 3, 2, FIC, fcheap, 2 * ncu);
        DBF("D2 db R2+next, x1 in regs, ALU sums, Sk2Op,    2/CU", 2, 2, Sk2Op<T>, sk2, 2 * ncu);
        DBF("D2 db R1+next, x1 in regs, ALU sums, logistic, 2/CU", 1, 2, FIL, flog, 2 * ncu);
        DBF("D3 db R1+next, x1 in regs, ALU sums, logistic, 3/CU", 1, 3, FIL, flog, 3 * ncu);
        FAST("F2r3 x1 in regs, ALU sums, logistic, R3 2/CU", 3, 2, true, true, FIL, flog, 2 * ncu);
        FAST("F2r4 x1 in regs, ALU sums, logistic, R4 2/CU", 4, 2, true, true, FIL, flog, 2 * ncu);
      } else if (table == 6) {
        // table 6: where a workgroup step's time goes (cycle stamps by thread 0 of every workgroup, averaged)
        unsigned long long *stamps;
        CK(hipMalloc(&stamps, sizeof(unsigned long long) * gmax * 8));
        auto timed = [&](const char *name, auto kern, auto opv, int g, size_t l) {
          CK(hipMemset(stamps, 0, sizeof(unsigned long long) * gmax * 8));
          const double ms = tm.run([&] { hipLaunchKernelGGL(kern, dim3(g), dim3(TPB), l, 0, a2, opv, stamps); }, 5);
          std::vector<unsigned long long> hst(static_cast<size_t>(g) * 8);
          CK(hipMemcpy(hst.data(), stamps, hst.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
          double ph[6] = {0, 0, 0, 0, 0, 0}, steps = 0;
          for (int b = 0; b < g; ++b) {
            for (int k = 0; k < 6; ++k) ph[k] += static_cast<double>(hst[static_cast<size_t>(b) * 8 + k]);
            steps += static_cast<double>(hst[static_cast<size_t>(b) * 8 + 6]);
          }
          double tot = 0;
          for (int k = 0; k < 6; ++k) tot += ph[k];
          // cycles -> us: the whole launch is ms long and a workgroup's steps fill it
          const double us_per_step = ms * 1e3 / (steps / g);
          printf("%-58s grid %4d  %.4f ms  %.2f us/step | tile wait %4.1f %%  dots %4.1f %%  barrier %4.1f %%  functor %4.1f %%  barrier %4.1f %%  column sums %4.1f %%  (%.0f cycles/step stamped)\n",
                 name, g, ms, us_per_step, 100 * ph[0] / tot, 100 * ph[1] / tot, 100 * ph[2] / tot, 100 * ph[3] / tot, 100 * ph[4] / tot, 100 * ph[5] / tot, tot / steps);
          fflush(stdout);
        };
        timed("H2 Sk2Op, two per CU", stream_rows2_timed_kernel<T, TPB, NV, 2, 2, 2, Sk2Op<T>>, sk2, 2 * ncu, lds2);
        timed("H3 Sk2Op, three per CU", stream_rows2_timed_kernel<T, TPB, NV, 2, 2, 2, Sk2Op<T>>, sk2, 3 * ncu, lds2);
        timed("K2 lasso-type prox, two per CU", stream_rows2_timed_kernel<T, TPB, NV, 2, 2, 2, FIC>, fcheap, 2 * ncu, lds2);
        timed("K3 lasso-type prox, three per CU", stream_rows2_timed_kernel<T, TPB, NV, 2, 2, 2, FIC>, fcheap, 3 * ncu, lds2);
        timed("L2 logistic prox, two per CU", stream_rows2_timed_kernel<T, TPB, NV, 2, 2, 2, FIL>, flog, 2 * ncu, lds2);
        timed("L3 logistic prox, three per CU (shipped shape)", stream_rows2_timed_kernel<T, TPB, NV, 2, 2, 2, FIL>, flog, 3 * ncu, lds2);
        CK(hipFree(stamps));
      } else if (table == 9) {
        // table 9: the plain skeleton and the prefetching one with the same (Sinkhorn-Knopp) functor at two workgroups per CU,
        // for a counter run (scripts/pmc_wg_per_cu.sh 9): why is the prefetching skeleton's floor 7 % higher?
        ROWS2("H2 rows2 Sk2Op    (two per CU)", 2, 2, 2, Sk2Op<T>, sk2, 2 * ncu);
        {
          auto k = stream_rows2_db_kernel<T, TPB, NV, 2, 2, 2, 2, Sk2Op<T>, true, true>;
          report("D2 db R2+next, x1 in regs, ALU sums, Sk2Op, 2/CU", 2 * ncu, regs_of(k),
                 tm.run([&] { hipLaunchKernelGGL(k, dim3(2 * ncu), dim3(TPB), lds2, 0, a2, sk2); }, reps));
        }
      } else if (table == 5) {
        // table 5: the same kernel at two and at three workgroups per CU, for a counter run (rocprofv3 --pmc ...)
        ROWS("B  rows  R2 dot+acc     SkRowOp            (two per CU)", 2, true, true, false, SkRowOp<T>, sk1, 2 * ncu);
        ROWS("B3 rows  R2 dot+acc     SkRowOp            (three per CU)", 2, true, true, false, SkRowOp<T>, sk1, 3 * ncu);
      } else if (table == 4) {
        // table 4: the functor on a wavefront of its own, the column sums one step late (stream_rows2_fw_kernel)
        ROWS("A  rows  R2 dot+acc SQ  SkRowOp            (Sinkhorn-Knopp's pass)", 2, true, true, true, SkRowOp<T>, sk1, 2 * ncu);
        ROWS2("H2 rows2 R2 2 dot 2 acc Sk2Op              (two per CU)", 2, 2, 2, Sk2Op<T>, sk2, 2 * ncu);
        ROWS2("L3 rows2 R2 2 dot 2 acc FusedIterOp logistic (the shipped C3 pass)", 2, 2, 2, FIL, flog, 3 * ncu);
        ROWS2("K2 rows2 R2 2 dot 2 acc FusedIterOp cheap    (two per CU: shipped for lasso-type since round 6)", 2, 2, 2, FIC, fcheap, 2 * ncu);
        ROWS2FW("W2  fw R2 2 dot 2 acc logistic  (two per CU)", 2, 2, 2, 2, FIL, flog, 2 * ncu);
        ROWS2FW("W2c fw R2 2 dot 2 acc cheap     (two per CU)", 2, 2, 2, 2, FIC, fcheap, 2 * ncu);
        ROWS2FW("W2s fw R2 2 dot 2 acc Sk2Op     (two per CU)", 2, 2, 2, 2, Sk2Op<T>, sk2, 2 * ncu);
        ROWS2FW("X2  fw R1 2 dot 2 acc logistic  (two per CU)", 1, 2, 2, 2, FIL, flog, 2 * ncu);
        ROWS2FW("X3  fw R1 2 dot 2 acc logistic  (three per CU)", 1, 2, 2, 3, FIL, flog, 3 * ncu);
        ROWS2FW("X4  fw R1 2 dot 2 acc logistic  (four per CU)", 1, 2, 2, 4, FIL, flog, 4 * ncu);
        ROWS2FW("Y2  fw R3 2 dot 2 acc logistic  (two per CU)", 3, 2, 2, 2, FIL, flog, 2 * ncu);
        ROWS2FW("W2l fw R2 1 dot 1 acc logistic  (lean, two per CU)", 2, 1, 1, 2, FIL, flog, 2 * ncu);
      } else if (table == 1) {
        ROWS("A  rows  R2 dot+acc SQ  SkRowOp            (Sinkhorn-Knopp's pass)", 2, true, true, true, SkRowOp<T>, sk1, 2 * ncu);
        ROWS("B  rows  R2 dot+acc     SkRowOp            (no squaring)", 2, true, true, false, SkRowOp<T>, sk1, 2 * ncu);
        ROWS("B3 rows  R2 dot+acc     SkRowOp            (three per CU)", 2, true, true, false, SkRowOp<T>, sk1, 3 * ncu);
        ROWS("C  rows  R2 dot+acc     Fused1Op cheap     (the iteration's functor, lean, in the rows skeleton)", 2, true, true, false, F1C, F1C{fcheap}, 2 * ncu);
        ROWS("C3 rows  R2 dot+acc     Fused1Op cheap     (three per CU)", 2, true, true, false, F1C, F1C{fcheap}, 3 * ncu);
        ROWS("CL rows  R2 dot+acc     Fused1Op logistic", 2, true, true, false, F1L, F1L{flog}, 2 * ncu);
        ROWS2("D  rows2 R2 1 dot 1 acc Sk2Op              (SK's functor in the rows2 skeleton)", 2, 1, 1, Sk2Op<T>, sk2, 2 * ncu);
        ROWS2("D3 rows2 R2 1 dot 1 acc Sk2Op              (three per CU)", 2, 1, 1, Sk2Op<T>, sk2, 3 * ncu);
        ROWS2("E3 rows2 R2 1 dot 1 acc Sk2SumsOp          (+ six fp64 sums)", 2, 1, 1, Sk2SumsOp<T>, sk2s, 3 * ncu);
        ROWS2("F3 rows2 R2 2 dot 1 acc Sk2Op              (+ second dot from LDS)", 2, 2, 1, Sk2Op<T>, sk2, 3 * ncu);
        ROWS2("G3 rows2 R2 1 dot 2 acc Sk2Op              (+ second accumulator)", 2, 1, 2, Sk2Op<T>, sk2, 3 * ncu);
        ROWS2("H3 rows2 R2 2 dot 2 acc Sk2Op              (both)", 2, 2, 2, Sk2Op<T>, sk2, 3 * ncu);
        ROWS2("H2 rows2 R2 2 dot 2 acc Sk2Op              (both, two per CU)", 2, 2, 2, Sk2Op<T>, sk2, 2 * ncu);
        ROWS2("I3 rows2 R2 2 dot 2 acc Sk2SumsOp          (both + sums)", 2, 2, 2, Sk2SumsOp<T>, sk2s, 3 * ncu);
        ROWS2("J3 rows2 R2 1 dot 1 acc FusedIterOp cheap  (lean iteration pass)", 2, 1, 1, FIC, fcheap, 3 * ncu);
        ROWS2("J2 rows2 R2 1 dot 1 acc FusedIterOp cheap  (two per CU)", 2, 1, 1, FIC, fcheap, 2 * ncu);
        ROWS2("K3 rows2 R2 2 dot 2 acc FusedIterOp cheap  (full iteration pass, lasso)", 2, 2, 2, FIC, fcheap, 3 * ncu);
        ROWS2("L3 rows2 R2 2 dot 2 acc FusedIterOp logistic (the shipped C3 pass)", 2, 2, 2, FIL, flog, 3 * ncu);
        ROWS2DB("M3 rows2-db R1+next 2 dot 2 acc FusedIterOp logistic", 1, 2, 2, 3, FIL, flog, 3 * ncu);
        ROWS2DB("N3 rows2-db R1+next 2 dot 2 acc Sk2Op", 1, 2, 2, 3, Sk2Op<T>, sk2, 3 * ncu);
        ROWS2DB("O3 rows2-db R1+next 1 dot 1 acc Sk2Op", 1, 1, 1, 3, Sk2Op<T>, sk2, 3 * ncu);
        ROWS2DB("O4 rows2-db R1+next 1 dot 1 acc Sk2Op   (four per CU)", 1, 1, 1, 4, Sk2Op<T>, sk2, 4 * ncu);
      } else {
        // table 2: the finding of table 1 (two workgroups per CU stream 8 % faster than three, whatever the skeleton; three
        // are there to hide the functor) -- what hides the functor at TWO per CU?
        ROWS("A  rows  R2 dot+acc SQ  SkRowOp            (Sinkhorn-Knopp's pass)", 2, true, true, true, SkRowOp<T>, sk1, 2 * ncu);
        ROWS2("L3 rows2 R2 2 dot 2 acc FusedIterOp logistic (the shipped C3 pass)", 2, 2, 2, FIL, flog, 3 * ncu);
        ROWS2("L2 rows2 R2 2 dot 2 acc FusedIterOp logistic (two per CU)", 2, 2, 2, FIL, flog, 2 * ncu);
        ROWS2("K3 rows2 R2 2 dot 2 acc FusedIterOp cheap", 2, 2, 2, FIC, fcheap, 3 * ncu);
        ROWS2("K2 rows2 R2 2 dot 2 acc FusedIterOp cheap    (two per CU)", 2, 2, 2, FIC, fcheap, 2 * ncu);
        ROWS2("H2 rows2 R2 2 dot 2 acc Sk2Op              (two per CU)", 2, 2, 2, Sk2Op<T>, sk2, 2 * ncu);
        ROWS2DB("M2  db R1+next 2 dot 2 acc logistic  (two per CU, 168 regs)", 1, 2, 2, 3, FIL, flog, 2 * ncu);
        ROWS2DB("M3  db R1+next 2 dot 2 acc logistic  (three per CU)", 1, 2, 2, 3, FIL, flog, 3 * ncu);
        ROWS2DB("P2  db R2+next 2 dot 2 acc logistic  (two per CU, 256 regs)", 2, 2, 2, 2, FIL, flog, 2 * ncu);
        ROWS2DB("P2c db R2+next 2 dot 2 acc cheap     (two per CU)", 2, 2, 2, 2, FIC, fcheap, 2 * ncu);
        ROWS2DB("P2s db R2+next 2 dot 2 acc Sk2Op     (two per CU)", 2, 2, 2, 2, Sk2Op<T>, sk2, 2 * ncu);
        ROWS2DB("P1  db R2+next 2 dot 2 acc logistic  (ONE per CU)", 2, 2, 2, 2, FIL, flog, 1 * ncu);
        ROWS2DB("Q2  db R3+next 2 dot 2 acc logistic  (two per CU)", 3, 2, 2, 2, FIL, flog, 2 * ncu);
        ROWS2DB("Q1  db R4+next 2 dot 2 acc logistic  (ONE per CU, 512 regs)", 4, 2, 2, 1, FIL, flog, 1 * ncu);
        ROWS2DB("R2  db R2+next 1 dot 1 acc logistic  (lean, two per CU)", 2, 1, 1, 2, FIL, flog, 2 * ncu);
      }
    } else {
      // C2's shape: 256 x 10, one row per step, two workgroups per CU
      ROWS("A  rows  R2 dot+acc SQ  SkRowOp            (Sinkhorn-Knopp's pass)", 2, true, true, true, SkRowOp<T>, sk1, 2 * ncu);
      ROWS2("K2 rows2 R1 2 dot 2 acc FusedIterOp cheap  (the shipped C2 pass)", 1, 2, 2, FIC, fcheap, 2 * ncu);
      ROWS2("H2 rows2 R1 2 dot 2 acc Sk2Op", 1, 2, 2, Sk2Op<T>, sk2, 2 * ncu);
      ROWS2("K1 rows2 R1 2 dot 2 acc FusedIterOp cheap  (one per CU)", 1, 2, 2, FIC, fcheap, 1 * ncu);
      ROWS2DB("M2  db R1+next 2 dot 2 acc cheap     (two per CU, 256 regs)", 1, 2, 2, 2, FIC, fcheap, 2 * ncu);
      ROWS2DB("M2s db R1+next 2 dot 2 acc Sk2Op     (two per CU)", 1, 2, 2, 2, Sk2Op<T>, sk2, 2 * ncu);
      {
        auto k = stream_rows2_db_kernel<T, TPB, NV, 1, 2, 2, 2, FIC, false, true>;
        report("M2a db R1+next 2 dot 2 acc cheap, ALU sums (two per CU)", 2 * ncu, regs_of(k),
               tm.run([&] { hipLaunchKernelGGL(k, dim3(2 * ncu), dim3(TPB), lds2, 0, a2, fcheap); }, reps));
        auto kf = stream_rows2_fast_kernel<T, TPB, NV, 1, 2, 2, 2, false, true, FIC>;
        report("F2a rows2 R1 cheap, ALU sums, x1 in LDS (two per CU)", 2 * ncu, regs_of(kf),
               tm.run([&] { hipLaunchKernelGGL(kf, dim3(2 * ncu), dim3(TPB), lds2, 0, a2, fcheap); }, reps));
      }
      ROWS2DB("M1  db R1+next 2 dot 2 acc cheap     (ONE per CU)", 1, 2, 2, 2, FIC, fcheap, 1 * ncu);
      ROWS2DB("P1  db R2+next 2 dot 2 acc cheap     (ONE per CU, 512 regs)", 2, 2, 2, 1, FIC, fcheap, 1 * ncu);
      ROWS2DB("R2  db R1+next 1 dot 1 acc cheap     (lean, two per CU)", 1, 1, 1, 2, FIC, fcheap, 2 * ncu);
    }
  }
  for (auto &p : vec) CK(hipFree(p));
  CK(hipFree(A)); CK(hipFree(xv)); CK(hipFree(x1)); CK(hipFree(h)); CK(hipFree(cp0)); CK(hipFree(cp1)); CK(hipFree(sp));
  return 0;
}

int main(int argc, char **argv) {
  // c3_bisect [table] [reps]: 1 = the bisection between the two skeletons (C3's shape), 2 = what hides the functor at two
  // workgroups per CU (C3's shape), 3 = the same question at C2's shape (256 x 10), 4 = the functor wavefront (C3's shape)
  const int table = argc > 1 ? atoi(argv[1]) : 1, reps = argc > 2 ? atoi(argv[2]) : 15;
  if (table == 3) return run<10>(100000, 10000, reps, table);
  if (table == 8) return run<3, 512>(200000, 5000, reps, table);
  return run<5>(200000, 5000, reps, table);
}
