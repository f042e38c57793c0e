#pragma once
// Dense graph-form ADMM solver with the direct (Gram + Cholesky) projector.
//
// Reference call stack being replaced:
//   PogsD/PogsS -> Pogs<T,O> (src/interface_c/pogs_c.cpp:9-55)
//     -> PogsImplementation::_Init (src/cpu/pogs.cpp:59-88)
//          MatrixDense::Init/Equil (src/cpu/matrix/matrix_dense.cpp:85-200)
//          Norm2Est (src/cpu/include/equil_helper.h:107-135)
//          ProjectorDirect::Init (src/cpu/projector/projector_direct_dense.cpp:45-84)
//     -> PogsImplementation::Solve (src/cpu/pogs.cpp:91-581)
//          ProjectorDirect::Project (projector_direct_dense.cpp:87-175)
//
// HBM layout: A_eq row-major m x lda (lda = n rounded up to 16 bytes, padding
// columns zero); x-sized vectors have n_pad entries with zero padding;
// W = inv(chol(A^T A + I)) lower-triangular and U = W^T, both n x n_pad.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <thread>

#include "cg_kernels.h"
#include "engine.h"
#include "fused_cols.h"
#include "gemm.h"
#include "ops.h"
#include "reduce.h"
#include "stream.h"
#include "vec_kernels.h"

namespace pogs_amd {

void rand_uniform_host(float *x, size_t n);
void rand_uniform_host(double *x, size_t n);

namespace {

// diag(d) A diag(e) in two passes (MultDiag + NormEst(kNormFro) + the division by the norm,
// matrix_dense.cpp:181-186,215-237).  WRITE = false: nothing is stored -- the sum of squares and the largest
// |entry| of the scaled matrix only.  WRITE = true: dst = (src_ij (d_i e_j)) * post, the same two roundings as
// storing the scaled matrix and multiplying it by 1 / normA in a pass of its own, which this replaces
// (read + read/write instead of two read/writes of the matrix; src may be the caller's own buffer).
template <typename T, bool WRITE>
__global__ void __launch_bounds__(256) scale_de_kernel(const T *src, T *dst, size_t lda, int m, int n_pad, const T *d,
                                                       const T *e, T post, double *partials, double *max_partials) {
  using V = typename Vec16<T>::type;
  constexpr int VEC = Vec16<T>::N;
  __shared__ double s_red[4];
  __shared__ double s_max[4];
  const int vpr = n_pad / VEC;
  double acc[1] = {0.0};
  T amax = 0;
  for (int row = blockIdx.x; row < m; row += gridDim.x) {
    const T di = d[row];
    const T *rp = src + static_cast<size_t>(row) * lda;
    for (int v = threadIdx.x; v < vpr; v += 256) {
      V a = WRITE ? stream_load<V>(rp + v * VEC) : *reinterpret_cast<const V *>(rp + v * VEC);
      const V ev = *reinterpret_cast<const V *>(e + v * VEC);
      T *ap = reinterpret_cast<T *>(&a);
      const T *ep = reinterpret_cast<const T *>(&ev);
#pragma unroll
      for (int c = 0; c < VEC; ++c) {
        const T val = ap[c] * (di * ep[c]);
        if (WRITE) {
          ap[c] = val * post;
        } else {
          acc[0] += static_cast<double>(val) * val;
          amax = fmax(amax, fabs(val));
        }
      }
      if (WRITE) *reinterpret_cast<V *>(dst + static_cast<size_t>(row) * lda + v * VEC) = a;
    }
  }
  if (WRITE) return;
  dev::block_sum<1, 256>(acc, s_red);
  double mx = static_cast<double>(amax);
  for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
  if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = acc[0];
    max_partials[blockIdx.x] = fmax(fmax(s_max[0], s_max[1]), fmax(s_max[2], s_max[3]));
  }
}

// Tag: the streaming shapes this instantiation carries (stream.h: OnePlan / WindowPlans /
// AllPlans); dense_plan.hip builds one solver class per shape.
template <typename T, typename Tag = AllPlans>
class DenseSolver final : public SolverBase {
 public:
  ~DenseSolver() override { begin_destroy(ctx_); }
  DenseSolver(int ord, size_t m, size_t n, const void *A, int mem, const PogsAmdOptions *opt,
              const PogsAmdDist *dist) {
    const double t0 = wall_s();
    // The first launch of a kernel of this translation unit makes the runtime load its code
    // object (once per process; ~2 ms now that there is one ~0.8 MB object per streaming shape).
    // On the first solver of a process (per arithmetic type) a helper thread asks for a kernel's
    // attributes right away, so that the load overlaps the stream creation and the upload of A:
    // ~2 ms of the cold time to converge at C2 (0.1483 -> 0.1462 s).
    struct Joiner {
      std::thread t;
      ~Joiner() { if (t.joinable()) t.join(); }
    } preload;
    {
      int dev = opt ? opt->device : -1;
      if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = -1;
      static std::atomic<bool> preloaded{false};
      if (dev >= 0 && !preloaded.exchange(true))
        preload.t = std::thread([dev] {
          hipFuncAttributes fa;
          if (hipSetDevice(dev) != hipSuccess) return;
          (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(scale_de_kernel<T, true>));
          preload_vec_code();
          preload_gemm_code();
        });
    }
    ctx_.init(opt ? opt->device : -1, opt ? opt->profile : 0);
    m_ = static_cast<int>(m);
    n_ = static_cast<int>(n);
    POGS_CHECK(m > 0 && n > 0 && m < (1u << 31) && n < (1u << 31), "bad dimensions");
    if (dist && dist->world >= 1) {   // a 1-rank communicator is allowed (exercises the RCCL path)
      ctx_.dist.init(dist->rank, dist->world, dist->unique_id);
      ctx_.m_global = dist->m_global;
    } else {
      ctx_.m_global = m;
    }
    multi_ = ctx_.dist.active();
    tall_ = ctx_.m_global > n;   // projector_direct_dense.cpp:53,122,128
    POGS_CHECK(tall_ || !multi_, "row sharding needs m > n (SURVEY.md section 8(e))");
    // dense + CGLS: instantiated by the reference (pogs.cpp:1983-1984) but not reachable from
    // its C ABI; offered here as an option (no Gram / factorisation, any shape)
    use_cgls_ = opt && opt->projector == POGS_AMD_PROJ_CGLS;
    if (const char *ys = std::getenv("POGS_AMD_YSYNC")) ysync_ = std::max(0, std::atoi(ys));
    POGS_CHECK(!(use_cgls_ && multi_), "the CGLS projector is single-GPU");
    constexpr int VEC = Vec16<T>::N;
    n_pad_ = static_cast<int>(round_up(n, VEC));
    m_pad_ = static_cast<int>(round_up(m, VEC));
    k_ = tall_ ? n_ : m_;
    k_pad_ = static_cast<int>(round_up(k_, VEC));
    // m <= n with the direct projector: the solver keeps T = A^T (n stored rows of length m), so
    // the register-tiled row kernel sees rows of min(m, n) elements whatever n is; A x becomes a
    // column-sum pass and A^T u a row-dot pass (see the adaptors in ops.h)
    tmode_ = !tall_ && !use_cgls_;
    srows_ = tmode_ ? n_ : m_;
    scols_pad_ = tmode_ ? m_pad_ : n_pad_;
    lda_ = scols_pad_;
    planA_ = make_stream_plan<T>(scols_pad_, ctx_.num_cu);
    POGS_CHECK(planA_.ok, tmode_ || tall_ ? "min(m, n) too large for the register-tiled streaming kernel"
                                          : "n too large for the register-tiled streaming kernel (CGLS projector)");
    POGS_CHECK(planA_.xl ? Tag::windows : Tag::has(planA_.tpb, planA_.nv),
               "dense solver built for another streaming shape (abi.hip: make_dense_solver)");
    ctx_.tmark_last = t0;
    ctx_.tmark("ctx init");
    upload(ord, A, mem);
    if (mem != POGS_AMD_DEVICE) ctx_.sync();   // t_h2d_s is the whole copy from the host, not the time to enqueue it
    ctx_.stats.t_h2d_s = wall_s() - t0;
    ctx_.tmark("upload");
    alloc_state();
    ctx_.tmark("alloc_state");
    equilibrate();
    ctx_.tmark("equilibrate");
    if (use_cgls_) norm_est();   // direct projector: estimated from the Gram matrix inside factor()
    if (!use_cgls_) factor();
    ctx_.sync();
    ctx_.stats.t_init_s = wall_s() - t0;
  }

  int dtype() const override { return sizeof(T) == 4 ? POGS_AMD_F32 : POGS_AMD_F64; }
  int device() const override { return ctx_.device; }
  void on_entry() override { ctx_.on_entry(); }
  void on_error() override { ctx_.on_error(); }
  PogsAmdStats &stats() override { return ctx_.stats; }

  int solve(const FnHost &f, const FnHost &g, const SolveParams &p, void *x, void *y, void *l, void *mu,
            double *optval, unsigned *final_iter) override {
    const double t0 = wall_s();
    load_problem(f, g, p);
    cold_start();
    apply_warm_start();
    ctx_.sync();
    const double t1 = wall_s();
    if (ctx_.dist.rank() == 0) print_banner(p.verbose);
    while (!iteration(p.verbose)) {}
    ctx_.sync();
    const double t2 = wall_s();
    const int status = epilogue(x, y, l, mu, optval);
    *final_iter = ctl_.k;
    PogsAmdStats &st = ctx_.stats;
    st.t_loop_s = t2 - t1;
    st.t_total_s = st.t_init_s + (wall_s() - t0);
    st.iterations = ctl_.k + 1;
    st.exact_iters = ctl_.exact_iters;
    st.rho_updates = ctl_.rho_updates;
    st.rho_final = ctl_.rho;
    collect_stream_timer();
    if (p.verbose > 0 && ctx_.dist.rank() == 0) {
      print_summary(status, st.t_total_s, st.t_init_s, ctl_);
      if (p.verbose > 3) print_timing_breakdown(st.t_loop_s, st.iterations);
      std::printf("POGS-AMD dense/%s: status %d, iter %u, init %.3e s, loop %.3e s\n", use_cgls_ ? "cgls" : "direct",
                  status, ctl_.k, st.t_init_s, st.t_loop_s);
    }
    return status;
  }

  void begin_run(const FnHost &f, const FnHost &g, const SolveParams &p) override {
    load_problem(f, g, p);
    cold_start();
    apply_warm_start();
    ctx_.sync();
  }

  void set_warm_start(const void *x0, const void *l0) override {
    warm_x_.assign(static_cast<const T *>(x0), static_cast<const T *>(x0) + n_);
    warm_l_.assign(static_cast<const T *>(l0), static_cast<const T *>(l0) + m_);
    warm_pending_ = true;
  }

  void iterate(unsigned iters, double *seconds, unsigned *solves) override {
    POGS_CHECK(loaded_, "PogsAmdIterate before PogsAmdBeginRun / PogsAmdSolve: no problem is loaded");
    unsigned done = 0;
    ctx_.sync();
    const double t0 = wall_s();
    for (unsigned i = 0; i < iters; ++i) {
      if (ctl_.finished) {
        cold_start();
        ++done;
      }
      iteration(0);
    }
    ctx_.sync();
    const double t1 = wall_s();
    if (seconds) *seconds = t1 - t0;
    if (solves) *solves = done;
    ctx_.stats.t_loop_s += t1 - t0;
    ctx_.stats.iterations += iters;
    collect_stream_timer();
  }

  void get_equil(void *A_eq, void *d, void *e, double *nrmA) override {
    ctx_.sync();
    if (A_eq && tmode_) {
      DevBuf<T> rm(static_cast<size_t>(m_) * n_);
      launch_transpose<T>(A_.p, lda_, n_, m_, rm.p, n_, ctx_.stream);
      ctx_.sync();
      POGS_HIP_CHECK(hipMemcpy(A_eq, rm.p, static_cast<size_t>(m_) * n_ * sizeof(T), hipMemcpyDeviceToHost));
    } else if (A_eq) {
      POGS_HIP_CHECK(hipMemcpy2D(A_eq, n_ * sizeof(T), A_.p, lda_ * sizeof(T), n_ * sizeof(T), m_,
                                 hipMemcpyDeviceToHost));
    }
    if (d) POGS_HIP_CHECK(hipMemcpy(d, d_.p, m_ * sizeof(T), hipMemcpyDeviceToHost));
    if (e) POGS_HIP_CHECK(hipMemcpy(e, e_.p, n_ * sizeof(T), hipMemcpyDeviceToHost));
    if (nrmA) *nrmA = nrmA_;
  }

  // (x, y) = Proj_{y = A x}(x0, y0), projector_direct_dense.cpp:122-127.
  void project(const void *x0, const void *y0, double tol, void *x, void *y) override {
    hipStream_t s = ctx_.stream;
    POGS_HIP_CHECK(hipMemcpyAsync(xtemp_.p, x0, n_ * sizeof(T), hipMemcpyHostToDevice, s));
    POGS_HIP_CHECK(hipMemcpyAsync(ytemp_.p, y0, m_ * sizeof(T), hipMemcpyHostToDevice, s));
    if (use_cgls_) {
      x_[0].zero(s);
      cgls_project(xtemp_.p, ytemp_.p, x_[0].p, static_cast<T>(tol), nullptr);
      StreamArgs<T> a = argsA();
      a.xin = x_[0].p;
      launch_stream<T, true, false, false, kFull, Tag>(planA_, a, GemvNOp<T>{1, 0, y_[0].p}, s);
    } else if (tall_) {
      gemv_t_partials(ytemp_.p);
      finish_cols(StoreColOp<T>{1, 0, rhs_.p, n_}, nullptr, 0, 0);
      solve_gram(rhs_.p, xtemp_.p, GemvNOp<T>{1, 0, x_[0].p}, nullptr);
      StreamArgs<T> a = argsA();
      a.xin = x_[0].p;
      launch_stream<T, true, false, false, kFull, Tag>(planA_, a, GemvNOp<T>{1, 0, y_[0].p}, s);
    } else {
      // projector_direct_dense.cpp:128-135: t = (A A^T + I)^{-1} (A x0 - y0); x = x0 - A^T t; y = y0 + t
      t_mul_n(xtemp_.p, nullptr, ResidOp<T>{ytemp_.p, rhs_.p}, nullptr);
      solve_gram(rhs_.p, static_cast<const T *>(nullptr), GemvNOp<T>{1, 0, tmpn_.p}, nullptr);
      launch_axpby<T>(m_, static_cast<T>(1), ytemp_.p, static_cast<T>(0), y_[0].p, s);
      launch_axpby<T>(m_, static_cast<T>(1), tmpn_.p, static_cast<T>(1), y_[0].p, s);
      launch_axpby<T>(n_, static_cast<T>(1), xtemp_.p, static_cast<T>(0), x_[0].p, s);
      t_mul_t(tmpn_.p, StoreColOp<T>{static_cast<T>(-1), static_cast<T>(1), x_[0].p, n_}, nullptr);
    }
    POGS_HIP_CHECK(hipMemcpyAsync(x, x_[0].p, n_ * sizeof(T), hipMemcpyDeviceToHost, s));
    POGS_HIP_CHECK(hipMemcpyAsync(y, y_[0].p, m_ * sizeof(T), hipMemcpyDeviceToHost, s));
    ctx_.sync();
  }

  void mul(char trans, double alpha, const void *x, double beta, void *y) override {
    hipStream_t s = ctx_.stream;
    const bool tr = (trans == 't' || trans == 'T');
    if (tmode_) {
      if (!tr) {
        POGS_HIP_CHECK(hipMemcpyAsync(xtemp_.p, x, n_ * sizeof(T), hipMemcpyHostToDevice, s));
        POGS_HIP_CHECK(hipMemcpyAsync(ytemp_.p, y, m_ * sizeof(T), hipMemcpyHostToDevice, s));
        t_mul_n(xtemp_.p, nullptr, GemvNOp<T>{static_cast<T>(alpha), static_cast<T>(beta), ytemp_.p}, nullptr);
        POGS_HIP_CHECK(hipMemcpyAsync(y, ytemp_.p, m_ * sizeof(T), hipMemcpyDeviceToHost, s));
      } else {
        POGS_HIP_CHECK(hipMemcpyAsync(ytemp_.p, x, m_ * sizeof(T), hipMemcpyHostToDevice, s));
        POGS_HIP_CHECK(hipMemcpyAsync(xtemp_.p, y, n_ * sizeof(T), hipMemcpyHostToDevice, s));
        t_mul_t(ytemp_.p, StoreColOp<T>{static_cast<T>(alpha), static_cast<T>(beta), xtemp_.p, n_}, nullptr);
        POGS_HIP_CHECK(hipMemcpyAsync(y, xtemp_.p, n_ * sizeof(T), hipMemcpyDeviceToHost, s));
      }
      ctx_.sync();
      return;
    }
    if (!tr) {
      POGS_HIP_CHECK(hipMemcpyAsync(xtemp_.p, x, n_ * sizeof(T), hipMemcpyHostToDevice, s));
      POGS_HIP_CHECK(hipMemcpyAsync(ytemp_.p, y, m_ * sizeof(T), hipMemcpyHostToDevice, s));
      StreamArgs<T> a = argsA();
      a.xin = xtemp_.p;
      launch_stream<T, true, false, false, kFull, Tag>(planA_, a,
                                                  GemvNOp<T>{static_cast<T>(alpha), static_cast<T>(beta), ytemp_.p}, s);
      POGS_HIP_CHECK(hipMemcpyAsync(y, ytemp_.p, m_ * sizeof(T), hipMemcpyDeviceToHost, s));
    } else {
      POGS_HIP_CHECK(hipMemcpyAsync(ytemp_.p, x, m_ * sizeof(T), hipMemcpyHostToDevice, s));
      POGS_HIP_CHECK(hipMemcpyAsync(xtemp_.p, y, n_ * sizeof(T), hipMemcpyHostToDevice, s));
      gemv_t_partials(ytemp_.p);
      if (multi_) {
        finish_cols(StoreColOp<T>{1, 0, rhs_.p, n_}, nullptr, 0, 0);
        launch_axpby<T>(n_, static_cast<T>(alpha), rhs_.p, static_cast<T>(beta), xtemp_.p, s);
      } else {
        finish_cols(StoreColOp<T>{static_cast<T>(alpha), static_cast<T>(beta), xtemp_.p, n_}, nullptr, 0, 0);
      }
      POGS_HIP_CHECK(hipMemcpyAsync(y, xtemp_.p, n_ * sizeof(T), hipMemcpyDeviceToHost, s));
    }
    ctx_.sync();
  }

 private:
  // ---- setup ---------------------------------------------------------------
  void upload(int ord, const void *A, int mem) {
    hipStream_t s = ctx_.stream;
    A_.alloc(static_cast<size_t>(srows_) * lda_);
    const hipMemcpyKind kind = (mem == POGS_AMD_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    // A matrix that is already in HBM in the stored layout (same pitch, 16-byte aligned) is not copied: the
    // equilibration passes read the caller's buffer and its last pass writes the scaled matrix straight into
    // A_ (equilibrate()); the caller's buffer is never written.  4 GB less to copy at C2 (1.5 ms).
    const char *ae = std::getenv("POGS_AMD_ALIAS_INPUT");
    bool may_alias = mem == POGS_AMD_DEVICE && !(ae && ae[0] == '0') &&
                     (reinterpret_cast<uintptr_t>(A) % 16) == 0;
    if (may_alias) {
      // only a buffer that lives on THIS handle's device is read in place: one on another GPU (or
      // host-mapped memory) goes through the runtime's copy as before
      hipPointerAttribute_t attr;
      if (hipPointerGetAttributes(&attr, A) != hipSuccess) {
        (void)hipGetLastError();
        may_alias = false;
      } else {
        may_alias = attr.type == hipMemoryTypeDevice && attr.device == ctx_.device;
      }
    }
    if (tmode_) {
      // stored matrix = A^T, n rows of m: column-major input already is that; row-major is transposed
      if (lda_ != static_cast<size_t>(m_)) A_.zero(s);
      if (ord != ROW_MAJ && may_alias && lda_ == static_cast<size_t>(m_)) {
        A_src_ = static_cast<const T *>(A);
      } else if (ord != ROW_MAJ) {
        POGS_HIP_CHECK(hipMemcpy2DAsync(A_.p, lda_ * sizeof(T), A, m_ * sizeof(T), m_ * sizeof(T), n_, kind, s));
      } else {
        DevBuf<T> stage;
        const T *src = static_cast<const T *>(A);
        if (mem != POGS_AMD_DEVICE) {
          stage.alloc(static_cast<size_t>(m_) * n_);
          POGS_HIP_CHECK(hipMemcpyAsync(stage.p, A, static_cast<size_t>(m_) * n_ * sizeof(T), kind, s));
          src = stage.p;
        }
        launch_transpose<T>(src, n_, m_, n_, A_.p, lda_, s);
        ctx_.sync();   // stage is freed at scope exit
      }
    } else if (ord == ROW_MAJ && may_alias && lda_ == static_cast<size_t>(n_)) {
      A_src_ = static_cast<const T *>(A);
    } else if (ord == ROW_MAJ) {
      if (lda_ != static_cast<size_t>(n_)) A_.zero(s);
      POGS_HIP_CHECK(hipMemcpy2DAsync(A_.p, lda_ * sizeof(T), A, n_ * sizeof(T), n_ * sizeof(T), m_, kind, s));
    } else {
      // column-major m x n == row-major n x m: stage and transpose on the device.
      DevBuf<T> stage;
      const T *src = static_cast<const T *>(A);
      if (mem != POGS_AMD_DEVICE) {
        stage.alloc(static_cast<size_t>(m_) * n_);
        POGS_HIP_CHECK(hipMemcpyAsync(stage.p, A, static_cast<size_t>(m_) * n_ * sizeof(T), kind, s));
        src = stage.p;
      }
      if (lda_ != static_cast<size_t>(n_)) A_.zero(s);
      launch_transpose<T>(src, m_, n_, m_, A_.p, lda_, s);
      ctx_.sync();
    }
    ctx_.sync();
  }

  void alloc_state() {
    hipStream_t s = ctx_.stream;
    const size_t np = n_pad_;
    const size_t mp = m_pad_;   // y-sized vectors are vector-loaded by the row kernel when T = A^T is stored
    for (int i = 0; i < 2; ++i) { x_[i].alloc(np); y_[i].alloc(mp); x_[i].zero(s); y_[i].zero(s); }
    xt_.alloc(np); yt_.alloc(mp); xtemp_.alloc(np); ytemp_.alloc(mp);
    const size_t kp = std::max<size_t>(np, k_pad_);
    x12_.alloc(np); y12_.alloc(mp); rhs_.alloc(kp); tvec_.alloc(kp); tmpn_.alloc(kp);
    xt_.zero(s); yt_.zero(s); xtemp_.zero(s); ytemp_.zero(s); x12_.zero(s); y12_.zero(s);
    rhs_.zero(s); tvec_.zero(s); tmpn_.zero(s);
    d_.alloc(mp); d_.zero(s); e_.alloc(np); e_.zero(s);
    if (tmode_) { uvec_.alloc(mp); uvec_.zero(s); }
    xout_.alloc(np); yout_.alloc(m_); lout_.alloc(m_); muout_.alloc(np);
    f_.alloc(m_); g_.alloc(n_); fs_.alloc(m_); gs_.alloc(n_);
    colpart_.alloc(static_cast<size_t>(planA_.grid_max) * scols_pad_);
    ensure_xl(planA_, srows_, scols_pad_);
    if (use_cgls_) {
      cg_p_.alloc(np); cg_s_.alloc(np); cg_q_.alloc(m_); cg_r_.alloc(m_); cg_.alloc(kCgNumSlots);
      cg_p_.zero(s); cg_s_.zero(s); cg_.zero(s);
    }
    const char *fe = std::getenv("POGS_AMD_FUSED");
    fused_ok_ = (tall_ || tmode_) && !use_cgls_ && stream2_supported(planA_) && !(fe && fe[0] == '0');
    if (fused_ok_ && tmode_) {
      colpart2_.alloc(static_cast<size_t>(planA_.grid_max) * scols_pad_);
      x12s_.alloc(np); xtemps_.alloc(np);
      x12s_.zero(s); xtemps_.zero(s);
    } else if (fused_ok_) {
      colpart2_.alloc(static_cast<size_t>(planA_.grid_max) * np);
      if (multi_) {   // [A^T yhat | exact-dual-residual sums | 6 scalars] in fp64: one all-reduce per iteration
        pack_.alloc(2 * np + 8);
        pack_.zero(s);
      }
      y12s_.alloc(m_); ytemps_.alloc(m_);
      y12s_.zero(s); ytemps_.zero(s);
    }
    // scalar-partials scratch: [stream passes | column reductions, vector kernels | prox partials
    // of the one-pass iteration | its projection-tail partials] -- the last two have regions of
    // their own because that iteration sums everything in its closing launch (Ctx::queue_sum)
    const size_t vb = vec_blocks(n_) + vec_blocks(m_);
    const size_t r01 = static_cast<size_t>(planA_.grid_max) * 6 + std::max<size_t>(4096, vb * 3 + 64);
    sp_pre_off_ = r01;   // [y-half prox sums: vec_blocks(m) x 3 | pre_cols sums: column blocks x 4]
    sp_tail_off_ = r01 + vb * 3 + static_cast<size_t>(pre_cols_grid(n_pad_, Vec16<T>::N)) * 4 + 64;
    ctx_.ensure_spart(sp_tail_off_ + static_cast<size_t>(ctx_.num_cu) * 32);
  }

  StreamArgs<T> argsA() const {
    StreamArgs<T> a;
    a.A = A_.p; a.lda = lda_; a.m = srows_; a.n_pad = scols_pad_;
    a.xin = nullptr; a.xin_add = nullptr; a.xin_nrm2 = nullptr;
    a.col_partials = colpart_.p; a.scalar_partials = ctx_.spart.p;
    a.xl_scratch = xl_buf_.p;
    return a;
  }
  // scratch of the windowed passes (rows wider than one register tile), grown on demand
  void ensure_xl(const StreamPlan &p, int rows, int n_pad) {
    const size_t need = stream_xl_scratch<T>(p, rows, n_pad);
    if (need > xl_buf_.n) {
      ctx_.sync();
      xl_buf_.alloc(need);
    }
  }

  // ---- products on the transposed storage (tmode_): same contracts as a row-dot pass with a row
  // functor over the m rows of A / a column-sum pass with a column functor over its n columns
  template <typename RowOp>
  void t_mul_n(const T *xin, const T *xin_add, const RowOp &op, double *scalar_out, const double *x_nrm2 = nullptr) {
    hipStream_t s = ctx_.stream;
    if (!tmode_) {
      StreamArgs<T> a = argsA();
      a.xin = xin; a.xin_add = xin_add; a.xin_nrm2 = x_nrm2;
      ctx_.stream_timer.begin(s);
      launch_stream<T, true, false, false, kFull, Tag>(planA_, a, op, s);
      ctx_.stream_timer.end(s);
      if (RowOp::NS > 0 && scalar_out) sum_row_scalars(stream_grid<true, false>(planA_, srows_), RowOp::NS, scalar_out);
      return;
    }
    StreamArgs<T> a = argsA();
    ctx_.stream_timer.begin(s);
    launch_stream<T, false, true, false, kFull, Tag>(planA_, a, VecCoefOp<T>{xin, xin_add, x_nrm2}, s);
    ctx_.stream_timer.end(s);
    double *sp = ctx_.spart.p + static_cast<size_t>(planA_.grid_max) * 6;
    launch_reduce_cols<T, RowAsColOp<T, RowOp>>(colpart_.p, stream_grid<false, true>(planA_, srows_), scols_pad_,
                                                RowAsColOp<T, RowOp>{op, m_}, sp, s);
    if (RowOp::NS > 0 && scalar_out) {
      SumJob j{sp, reduce_cols_grid(scols_pad_, Vec16<T>::N), RowOp::NS, scalar_out};
      launch_sum_jobs(&j, 1, s);
    }
  }
  template <typename ColOp>
  void t_mul_t(const T *u, const ColOp &op, double *scalar_out) {
    hipStream_t s = ctx_.stream;
    if (!tmode_) {
      gemv_t_partials(u);
      finish_cols(op, scalar_out, 0, 0);
      return;
    }
    StreamArgs<T> a = argsA();
    a.xin = u;   // length m, zero-padded to the vector width
    ctx_.stream_timer.begin(s);
    launch_stream<T, true, false, false, kFull, Tag>(planA_, a, ColAsRowOp<T, ColOp>{op}, s);
    ctx_.stream_timer.end(s);
    if (ColOp::NS > 0 && scalar_out) sum_row_scalars(stream_grid<true, false>(planA_, srows_), ColOp::NS, scalar_out);
  }

  // Second stage of a column-sum pass.  With row shards the totals are
  // all-reduced (together with scalar slots [yslot, yslot+count) if count > 0)
  // before op runs.
  template <typename ColOp>
  void finish_cols(const ColOp &op, double *colop_scalar_out, int yslot, int yslot_count,
                   int nparts_override = -1, const T *partials_src = nullptr) {
    hipStream_t s = ctx_.stream;
    const T *colpart = partials_src ? partials_src : colpart_.p;
    const int nparts = nparts_override > 0 ? nparts_override : stream_grid<false, true>(planA_, m_);
    double *sp = ctx_.spart.p + static_cast<size_t>(planA_.grid_max) * 6;  // separate scratch region
    if (!multi_) {
      launch_reduce_cols<T, ColOp>(colpart, nparts, n_pad_, op, sp, s);
    } else {
      launch_reduce_cols<T, StoreColOp<T>>(colpart, nparts, n_pad_, StoreColOp<T>{1, 0, tmpn_.p, n_}, sp, s);
      if (yslot_count > 0) ctx_.dist.allreduce2<T>(tmpn_.p, n_pad_, ctx_.S.p + yslot, yslot_count, s);
      else ctx_.dist.allreduce(tmpn_.p, n_pad_, s);
      launch_reduce_cols<T, ColOp>(tmpn_.p, 1, n_pad_, op, sp, s);
    }
    if (ColOp::NS > 0 && colop_scalar_out) {
      SumJob j{sp, reduce_cols_grid(n_pad_, Vec16<T>::N), ColOp::NS, colop_scalar_out};
      sum_now_or_later(j);
    }
  }

  // Scalar sums of the one-pass iteration wait for its closing launch (single GPU); everywhere
  // else they run at once.
  void sum_now_or_later(const SumJob &j) {
    if (defer_sums_) ctx_.queue_sum(j);
    else launch_sum_jobs(&j, 1, ctx_.stream);
  }

  // col partials <- A^T u (ACC-only pass)
  void gemv_t_partials(const T *u) {
    StreamArgs<T> a = argsA();
    ctx_.stream_timer.begin(ctx_.stream);
    launch_stream<T, false, true, false, kFull, Tag>(planA_, a, GemvTOp<T>{1, u}, ctx_.stream);
    ctx_.stream_timer.end(ctx_.stream);
  }

  void sum_row_scalars(int grid, int ns, double *out) {
    SumJob j{ctx_.spart.p, grid, ns, out};
    launch_sum_jobs(&j, 1, ctx_.stream);
  }

  // MatrixDense::Equil without materialising A.^2 (matrix_dense.cpp:116-200,
  // equil_helper.h:140-164): 51 passes over A instead of 100.
  void equilibrate() {
    hipStream_t s = ctx_.stream;
    PhaseTimer pt(s);
    const double mg = static_cast<double>(ctx_.m_global), nn = n_;
    const T ce = static_cast<T>(1e-4) * static_cast<T>(mg + nn) / static_cast<T>(mg);   // equil_helper.h:152-153
    const T cd = static_cast<T>(1e-4) * static_cast<T>(mg + nn) / static_cast<T>(nn);   // :159-160
    const int gridACC = stream_grid<false, true>(planA_, srows_);
    const int gridBOTH = stream_grid<true, true>(planA_, srows_);
    StreamArgs<T> a = argsA();
    if (A_src_) a.A = A_src_;   // the caller's buffer (upload()): read-only until the scaled copy is written
    ctx_.tmark("  eq: start");
    // The reference runs a fixed 50 iterations (equil_helper.h:147).  After a few of them the only
    // thing that still moves is the common factor (d * a, e / a) -- the unregularised iteration
    // does not fix it, and the two regularisers pull it towards its fixed point at a rate of
    // ~1e-8 per iteration -- so every entry of the scaling vector changes by the same ratio
    // 1 + gamma (~1e-6).  The column functor measures that ratio (mean over the entries, in double)
    // and stamps the pass if any entry deviates from the previous pass's mean by more than 16 ulp
    // (fp64: 64 ulp = 1.4e-14, where the entries' own drift rates differ by a few 1e-15);
    // the first pass without a stamp (fp64: the second in a row) ends the loop, and the remaining
    // iterations are applied in closed form: the newer vector times the product of the growth
    // factors still to come, the other one divided by the matching product.
    // POGS_AMD_SK_FULL=1 runs all 50 passes.
    const char *sk_env = std::getenv("POGS_AMD_SK_FULL");
    const bool sk_probe = !(sk_env && sk_env[0] == '1');
    double *mark = sk_probe ? ctx_.S.p + kSkMark : nullptr;
    const T sk_tol = (std::is_same<T, float>::value ? 16 : 64) * std::numeric_limits<T>::epsilon();
    double r_ref = 0, gamma = 0, gamma_prev = 0;
    bool extrapolate = false, was_uniform = false;
    // after pass k (0-based): true if it was a pure common-factor pass; keeps r_ref current
    auto sk_uniform = [&](int k, int count) {
      if (!mark || k < 1) return false;
      const double *S = ctx_.fetch_scalars();
      const double r_mean = S[kSkRatio] / count;
      const bool uniform = k >= 2 && S[kSkMark] < k + 1.0 && r_mean > 0.5 && r_mean < 2.0;
      r_ref = r_mean;
      gamma_prev = gamma;
      gamma = r_mean - 1.0;
      // fp64 also uses the previous pass's gamma (below), so that one has to be clean as well
      if (std::getenv("POGS_AMD_TRACE"))
        std::fprintf(stderr, "[pogs_amd trace]   sk pass %d: gamma %.6e, %s\n", k, r_mean - 1.0, uniform ? "uniform" : "stamped");
      const bool fire = uniform && (std::is_same<T, float>::value || was_uniform);
      was_uniform = uniform;
      return fire;
    };
    // log of the product of the next `count` growth factors, the first of which is
    // (1 + gamma q^first).  In fp32 gamma is taken as constant (its own change over 50 iterations,
    // ~1e-6 relative, is far below fp32 resolution); in fp64 it is not: the drift slows down
    // geometrically as the common factor approaches its fixed point, and the ratio q of two
    // consecutive measurements carries that (second-order terms are ~1e-14).
    auto sk_log_growth = [&](int first, int count) {
      double q = 1.0;
      if (std::is_same<T, double>::value && gamma != 0 && gamma_prev != 0) {
        q = gamma / gamma_prev;
        if (!(q > 0.999 && q < 1.001)) q = 1.0;
      }
      double L = 0, gi = gamma * std::pow(q, first);
      for (int i = 0; i < count; ++i, gi *= q) L += std::log1p(gi);
      return L;
    };
    int k = 0;
    if (tmode_) {
      // stored rows are the columns of A: one fused pass per iteration, the row dot (with d) gives
      // e_j, the column sums (weighted by e_j) give d   (equil_helper.h:149-163, d = 1 to start)
      double *sp = ctx_.spart.p + static_cast<size_t>(planA_.grid_max) * 6;
      launch_fill<T>(d_.p, static_cast<T>(1), m_, s);
      for (; k < 50; ++k) {
        a.xin = d_.p;
        launch_stream<T, true, true, true, kFull, Tag>(planA_, a, SkRowOp<T>{static_cast<T>(mg), ce, e_.p}, s);
        launch_reduce_cols<T, SkColOp<T>>(
            colpart_.p, gridBOTH, scols_pad_,
            SkColOp<T>{static_cast<T>(nn), cd, d_.p, m_, mark, k + 1.0, sk_tol, static_cast<T>(r_ref)}, sp, s);
        SumJob j{sp, reduce_cols_grid(scols_pad_, Vec16<T>::N), 1, ctx_.S.p + kSkRatio};
        launch_sum_jobs(&j, 1, s);
        if (sk_uniform(k, m_)) { extrapolate = true; ++k; break; }
      }
      ctx_.stats.matvecs_init += k;
      if (extrapolate) {
        // state (e_{k-1}, d_k) after k passes, gamma measured on d_k / d_{k-1}; the reference ends
        // with (e_49, d_50): d_50 = d_k prod_{i=1..50-k} (1 + gamma_i), e_49 = f(d_49) = e_{k-1} d_{k-1} / d_49
        const double Ld = sk_log_growth(1, 50 - k);
        const double Le = -sk_log_growth(0, 50 - k);
        launch_scal<T>(d_.p, static_cast<T>(std::exp(Ld)), m_, s);
        launch_scal<T>(e_.p, static_cast<T>(std::exp(Le)), n_, s);
      }
    } else {
      launch_stream<T, false, true, true, kFull, Tag>(planA_, a, OnesOp<T>{}, s);
      ctx_.tmark("  eq: first pass");
      finish_cols(SkColOp<T>{static_cast<T>(mg), ce, e_.p, n_}, nullptr, 0, 0, gridACC);
      ctx_.tmark("  eq: first cols");
      for (; k < 50; ++k) {
        a.xin = e_.p;
        if (k < 49) {
          launch_stream<T, true, true, true, kFull, Tag>(planA_, a, SkRowOp<T>{static_cast<T>(nn), cd, d_.p}, s);
          finish_cols(SkColOp<T>{static_cast<T>(mg), ce, e_.p, n_, mark, k + 1.0, sk_tol, static_cast<T>(r_ref)},
                      ctx_.S.p + kSkRatio, 0, 0, gridBOTH);
          if (sk_uniform(k, n_)) { extrapolate = true; ++k; break; }
        } else {
          launch_stream<T, true, false, true, kFull, Tag>(planA_, a, SkRowOp<T>{static_cast<T>(nn), cd, d_.p}, s);
        }
      }
      ctx_.stats.matvecs_init += k + 1;
      if (extrapolate) {
        // state (d_k, e_k) after k loop passes, gamma measured on e_k / e_{k-1}; the reference ends
        // with (d_50, e_49): e_49 = e_k prod_{i=1..49-k} (1 + gamma_i), d_50 = g(e_49) = d_k e_{k-1} / e_49
        const double Le = sk_log_growth(1, 49 - k);
        const double Ld = -sk_log_growth(0, 50 - k);
        launch_scal<T>(d_.p, static_cast<T>(std::exp(Ld)), m_, s);
        launch_scal<T>(e_.p, static_cast<T>(std::exp(Le)), n_, s);
      }
    }
    ctx_.tmark("  eq: sk loop");
    launch_sqrt_inplace<T>(d_.p, m_, s);                                  // matrix_dense.cpp:176-177
    launch_sqrt_inplace<T>(e_.p, n_, s);
    const int sgrid = std::min(srows_, ctx_.num_cu * 8);
    // rows of the stored matrix are scaled by the first vector, its columns by the second
    const T *src = A_src_ ? A_src_ : A_.p;
    const T *dr = tmode_ ? e_.p : d_.p, *dc = tmode_ ? d_.p : e_.p;
    hipLaunchKernelGGL((scale_de_kernel<T, false>), dim3(sgrid), dim3(256), 0, s, src, static_cast<T *>(nullptr), lda_,
                       srows_, scols_pad_, dr, dc, static_cast<T>(1), ctx_.spart.p, ctx_.spart.p + sgrid);
    sum_row_scalars(sgrid, 1, ctx_.S.p + kFro2);
    launch_max_partials(ctx_.spart.p + sgrid, sgrid, ctx_.S.p + kAmax, s);
    if (multi_) ctx_.dist.allreduce(ctx_.S.p + kFro2, 1, s);
    const double *S = ctx_.fetch_scalars();
    const T normA = static_cast<T>(std::sqrt(S[kFro2])) /
                    std::sqrt(static_cast<T>(std::min<double>(mg, nn)));   // :215-218
    amax_ = S[kAmax] / static_cast<double>(normA);   // largest |entry| of the equilibrated matrix (this shard)
    hipLaunchKernelGGL((scale_de_kernel<T, true>), dim3(sgrid), dim3(256), 0, s, src, A_.p, lda_, srows_, scols_pad_,
                       dr, dc, static_cast<T>(1) / normA, static_cast<double *>(nullptr),
                       static_cast<double *>(nullptr));                    // :181,186
    A_src_ = nullptr;   // from here on the solver's own (equilibrated) copy
    const T invs = static_cast<T>(1) / std::sqrt(normA);                   // :191-192
    launch_scal<T>(d_.p, invs, m_, s);
    launch_scal<T>(e_.p, invs, n_, s);
    ctx_.stats.equil_ms = pt.stop_ms();
  }

  // Norm2Est (equil_helper.h:107-135), one fused pass per power iteration.
  void norm_est() {
    hipStream_t s = ctx_.stream;
    PhaseTimer pt(s);
    std::vector<T> x0(n_pad_, 0);
    rand_uniform_host(x0.data(), n_);
    POGS_HIP_CHECK(hipMemcpyAsync(xtemp_.p, x0.data(), n_pad_ * sizeof(T), hipMemcpyHostToDevice, s));
    ctx_.sync();
    T *xa = xtemp_.p, *xb = rhs_.p;
    const T kTol = static_cast<T>(1e-4);
    T norm_est = 0, last;
    const int grid = stream_grid<true, true>(planA_, srows_);
    unsigned i = 0;
    for (i = 0; tmode_ && i < 50; ++i) {
      // transposed storage: Sx = A (x / |x|) is a column-sum pass, x' = A^T Sx a row-dot pass
      last = norm_est;
      t_mul_n(xa, nullptr, StoreNormRowOp<T>{ytemp_.p}, ctx_.S.p + kPowSx2, (i == 0) ? nullptr : ctx_.S.p + kPowX2);
      t_mul_t(ytemp_.p, PowerColOp<T>{xb, n_}, ctx_.S.p + kPowX2);
      const double *S = ctx_.fetch_scalars();
      norm_est = static_cast<T>(std::sqrt(S[kPowX2])) / static_cast<T>(std::sqrt(S[kPowSx2]));
      std::swap(xa, xb);
      ctx_.stats.matvecs_init += 2;
      if (std::abs(last - norm_est) < kTol * norm_est) { ++i; break; }
    }
    for (; !tmode_ && i < 50; ++i) {
      last = norm_est;
      StreamArgs<T> a = argsA();
      a.xin = xa;
      a.xin_nrm2 = (i == 0) ? nullptr : ctx_.S.p + kPowX2;
      launch_stream<T, true, true, false, kFull, Tag>(planA_, a, PowerRowOp<T>{}, s);
      sum_row_scalars(grid, 1, ctx_.S.p + kPowSx2);
      // kPowX2 is read by the pass above (x normalisation) and rewritten here.
      // with shards |Sx|^2 travels in the same RCCL group as the column totals
      finish_cols(PowerColOp<T>{xb, n_}, ctx_.S.p + kPowX2, kPowSx2, 1, grid);
      const double *S = ctx_.fetch_scalars();
      const T normx = static_cast<T>(std::sqrt(S[kPowX2]));
      const T normSx = static_cast<T>(std::sqrt(S[kPowSx2]));
      norm_est = normx / normSx;
      std::swap(xa, xb);
      ctx_.stats.matvecs_init += 1;
      if (std::abs(last - norm_est) < kTol * norm_est) { ++i; break; }
    }
    nrmA_ = norm_est;
    ctx_.stats.nrmA = nrmA_;
    ctx_.stats.norm_est_iters = i;
    // leave the work vectors clean
    xtemp_.zero(s);
    rhs_.zero(s);
    ytemp_.zero(s);
    ctx_.stats.normest_ms = pt.stop_ms();
  }

  // Norm2Est (equil_helper.h:107-135) for m > n, run on G = A^T A instead of A: the
  // iteration x <- A^T (A x) is x <- G x and |A x|^2 = x^T G x, so each power step reads
  // the n x n lower triangle (0.2 GB at C2) instead of A (4 GB).  Same start vector, same
  // normalisation and stopping rule; G is already summed over shards.
  void norm_est_gram(T *G, size_t ld) {
    hipStream_t s = ctx_.stream;
    PhaseTimer pt(s);
    launch_zero_upper<T>(G, ld, n_, s);
    ctx_.tmark("  ne: zero_upper");
    std::vector<T> x0(n_pad_, 0);
    rand_uniform_host(x0.data(), n_);
    ctx_.tmark("  ne: rand host");
    POGS_HIP_CHECK(hipMemcpyAsync(xtemp_.p, x0.data(), n_pad_ * sizeof(T), hipMemcpyHostToDevice, s));
    ctx_.sync();
    ctx_.tmark("  ne: h2d");
    T *xa = xtemp_.p, *xb = rhs_.p;
    const T kTol = static_cast<T>(1e-4);
    T norm_est = 0, last;
    const int grid = stream_grid<true, true>(planW_, n_);
    double *sp = ctx_.spart.p + static_cast<size_t>(planA_.grid_max) * 6;
    unsigned i = 0;
    for (i = 0; i < 50; ++i) {
      last = norm_est;
      const double *nrm = (i == 0) ? nullptr : ctx_.S.p + kPowX2;
      StreamArgs<T> a;
      a.A = G; a.lda = ld; a.m = n_; a.n_pad = n_pad_;
      a.xin = xa; a.xin_add = nullptr; a.xin_nrm2 = nrm;
      a.col_partials = colpart_.p; a.scalar_partials = ctx_.spart.p;
      a.xl_scratch = xl_buf_.p;
      launch_stream<T, true, true, false, kLower, Tag>(planW_, a, SymRowOp<T>{G, ld, xa, nrm, tvec_.p}, s);
      launch_reduce_cols<T, SymColOp<T>>(colpart_.p, grid, n_pad_, SymColOp<T>{tvec_.p, xa, nrm, xb, n_}, sp, s);
      SumJob j{sp, reduce_cols_grid(n_pad_, Vec16<T>::N), 2, ctx_.S.p + kPowX2};   // -> kPowX2, kPowXGx
      launch_sum_jobs(&j, 1, s);
      const double *S = ctx_.fetch_scalars();
      const T normx = static_cast<T>(std::sqrt(S[kPowX2]));
      const T normSx = static_cast<T>(std::sqrt(S[kPowXGx]));
      norm_est = normx / normSx;
      std::swap(xa, xb);
      if (std::abs(last - norm_est) < kTol * norm_est) { ++i; break; }
    }
    nrmA_ = norm_est;
    ctx_.stats.nrmA = nrmA_;
    ctx_.stats.norm_est_iters = i;
    xtemp_.zero(s);
    rhs_.zero(s);
    tvec_.zero(s);
    ctx_.stats.normest_ms = pt.stop_ms();
  }

  // Norm2Est for m <= n on G = A A^T.  With y = A x^ (x^ the normalised iterate) the reference's
  // step x' = A^T (A x^), est = |x'| / |A x^|, x^ <- x' / |x'| reads  |x'|^2 = y^T G y,
  // |A x^| = |y|,  y <- G y / |x'|:  after ONE product with A (y0 = A x0, x0 the same random
  // start vector, un-normalised as in equil_helper.h:113-121) every power step is a symmetric
  // m x m product instead of two passes over A.
  void norm_est_gram_wide(T *G, size_t ld) {
    hipStream_t s = ctx_.stream;
    PhaseTimer pt(s);
    launch_zero_upper<T>(G, ld, k_, s);
    std::vector<T> x0(n_pad_, 0);
    rand_uniform_host(x0.data(), n_);
    POGS_HIP_CHECK(hipMemcpyAsync(xtemp_.p, x0.data(), n_pad_ * sizeof(T), hipMemcpyHostToDevice, s));
    ctx_.sync();
    T *ya = ytemp_.p, *yb = uvec_.p;
    t_mul_n(xtemp_.p, nullptr, StoreNormRowOp<T>{ya}, ctx_.S.p + kPowSx2);   // y0 = A x0, |y0|^2
    ctx_.stats.matvecs_init += 1;
    const T kTol = static_cast<T>(1e-4);
    T norm_est = 0, last;
    const int grid = stream_grid<true, true>(planW_, k_);
    double *sp = ctx_.spart.p + static_cast<size_t>(planA_.grid_max) * 6;
    double y2 = ctx_.fetch_scalars()[kPowSx2];   // |y^|^2 of the current iterate
    unsigned i = 0;
    for (i = 0; i < 50; ++i) {
      last = norm_est;
      // the stored iterate is w = G y^_prev; y^ = w / sqrt(y^_prev^T G y^_prev): normaliser = previous kPowXGx
      const double *nrm = (i == 0) ? nullptr : ctx_.S.p + kPowXGx;
      StreamArgs<T> a;
      a.A = G; a.lda = ld; a.m = k_; a.n_pad = k_pad_;
      a.xin = ya; a.xin_add = nullptr; a.xin_nrm2 = nrm;
      a.col_partials = colpart_.p; a.scalar_partials = ctx_.spart.p;
      a.xl_scratch = xl_buf_.p;
      launch_stream<T, true, true, false, kLower, Tag>(planW_, a, SymRowOp<T>{G, ld, ya, nrm, tvec_.p}, s);
      launch_reduce_cols<T, SymColOp<T>>(colpart_.p, grid, k_pad_, SymColOp<T>{tvec_.p, ya, nrm, yb, k_}, sp, s);
      double *tmp2 = ctx_.S.p + kPowX2;   // -> kPowX2 = |G y^|^2, kPowXGx = y^^T G y^ = |x'|^2
      const double prev_xgx = (i == 0) ? 1.0 : ctx_.S_host.p[kPowXGx];
      const double prev_w2 = (i == 0) ? y2 : ctx_.S_host.p[kPowX2];
      SumJob j{sp, reduce_cols_grid(k_pad_, Vec16<T>::N), 2, tmp2};
      launch_sum_jobs(&j, 1, s);
      const double *S = ctx_.fetch_scalars();
      y2 = prev_w2 / prev_xgx;                                   // |y^|^2 = |w|^2 / normaliser^2
      norm_est = static_cast<T>(std::sqrt(S[kPowXGx])) / static_cast<T>(std::sqrt(y2));
      std::swap(ya, yb);
      if (std::abs(last - norm_est) < kTol * norm_est) { ++i; break; }
    }
    nrmA_ = norm_est;
    ctx_.stats.nrmA = nrmA_;
    ctx_.stats.norm_est_iters = i;
    xtemp_.zero(s);
    ytemp_.zero(s);
    uvec_.zero(s);
    tvec_.zero(s);
    ctx_.stats.normest_ms = pt.stop_ms();
  }

  // ProjectorDirect::Init + the first-call factorisation (s = 1 always,
  // pogs.cpp:293,296): G = A^T A (m > n) or A A^T (m <= n) on MFMA tiles,
  // L L^T = G + I, W = L^{-1}, U = W^T.
  void factor() {
    hipStream_t s = ctx_.stream;
    const size_t ld = k_pad_;
    planW_ = make_stream_plan<T>(k_pad_, ctx_.num_cu);
    ensure_xl(planW_, k_, k_pad_);
    // One allocation, four k x k slabs: [G -> L | scratch | W = L^-1 | U = W^T].
    const size_t slab = static_cast<size_t>(k_) * ld;
    fac_.alloc(slab * 4);
    fac_.zero(s);
    T *G = fac_.p, *tmp = fac_.p + slab;
    Wp_ = fac_.p + 2 * slab;
    Up_ = fac_.p + 3 * slab;
    {
      PhaseTimer pt(s);
      // split-K: short K ranges keep the workgroups of an XCD in step on the same rows of A
      // (L2 hits; one long K range per tile measured 121 ms against 100 ms at C2), give every
      // CU work to the end of the launch, and form the fp32 K-sum as an ordered sum of short
      // sums: a sequential fp32 sum over 1e5 rows costs ~30 % more ADMM iterations at C2.
      // The K ranges are processed in rounds that write their partial products into the four
      // slabs of fac_ itself (4 ranges in the first round, 3 in the later ones: slab 0 carries
      // the running sum), added in range order -- no transient multi-GB allocation, whose
      // first-touch cost was seen to stall this phase by 100-180 ms now and then.
      const int kdim = tall_ ? m_ : n_;
      const long long tiles = static_cast<long long>((k_ + 127) / 128) * ((k_ + 127) / 128 + 1) / 2;
      int ksplit = 1;
      while (ksplit < 32 && kdim / (ksplit * 2) >= 2048 && (kdim / ksplit > 6400 || tiles * ksplit < 16LL * ctx_.num_cu * 3))
        ksplit *= 2;
      GemmArgs<T> g{k_, k_, kdim, A_.p, lda_, A_.p, lda_, G, ld, static_cast<T>(1), static_cast<T>(0)};
      g.kchunk = ksplit > 1 ? static_cast<int>(round_up((kdim + ksplit - 1) / ksplit, 32)) : 0;
      g.csplit_stride = slab;
      // where K ranges stay longer than ~6.4k rows the unit itself sums in chunks
      const int klen = ksplit > 1 ? g.kchunk : kdim;
      const int nacc = (klen + 6399) / 6400;
      // (fp32 only: the chunks bound the rounding of a long fp32 sum; an fp64 sum over 1e5 rows is exact to 1e-11,
      // and the one-level kernel runs two workgroups per CU where the two-level one has registers for one)
      g.kacc = (nacc > 1 && std::is_same<T, float>::value) ? static_cast<int>(round_up((klen + nacc - 1) / nacc, 32)) : 0;
      DevBuf<int> tmap;
      if (k_ > 16 * 128 && k_ < 65536 * 128) {
        const std::vector<int> order = gram_tile_order(k_);
        tmap.alloc(order.size());
        POGS_HIP_CHECK(hipMemcpyAsync(tmap.p, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice, s));
        ctx_.sync();   // order is a host temporary
        g.tile_map = tmap.p;
      }
      // fp32, K-major operand, enough rows: the fp16 matrix cores at (better than) fp32 accuracy --
      // operands scaled by a power of two into fp16 range and split in two fp16 parts, three
      // products (gemm.h).  1024-row K ranges, four at a time into the four slabs, each launch
      // adding to what the slabs hold; then the slabs are added in order.
      const char *gsel = std::getenv("POGS_AMD_GRAM");
      bool split16 = std::is_same<T, float>::value && (tall_ || tmode_) && kdim >= 8192 && k_ >= 256 &&
                     !(gsel && gsel[0] == 'f') && std::isfinite(amax_) && amax_ > 0;
      float scale16 = 1.f;
      if (split16) {
        int ex = 0;
        std::frexp(amax_, &ex);                       // amax_ = f * 2^ex, f in [0.5, 1)
        scale16 = std::ldexp(1.f, 14 - ex);           // largest scaled entry in [8192, 16384)
        split16 = std::isfinite(scale16) && scale16 > 0;
      }
      if (split16) {
        // The K dimension is cut into equal units of at most ~12800 rows, four per launch into the
        // four slabs (C2: 2 launches x 4 units of 12512 rows): long units pay the accumulator
        // read-add-write, the prologue and the first-copy latency less often, equal ones leave no
        // mostly-empty unit at the end.  The rows of a launch are first written as two fp16 images
        // in operand order (launch_split_f16: 168 MB per 4096 rows at C2), which the product kernel
        // copies straight into LDS (gemm.h).
        // 256 x 256 workgroup tiles (half the operand bytes per product of the 128 tile; one
        // accumulator set, i.e. a unit is ONE MFMA chain -- chains of 1024 .. 16384 rows give the same
        // 106 iterations at C2 and x within 6e-7 of each other, the distance the native fp32 product
        // is at) from n = 4096 on; the 128 tile below, with
        // 1024-row chains added to a second register set.  POGS_AMD_GRAM_TILE=128 forces the 128
        // tile (regression sweep of tests/test_gpu_dense.py).
        constexpr int kRows = 1024, kUnitCap = 12800;
        const int launches = (kdim + 4 * kUnitCap - 1) / (4 * kUnitCap);
        // (measured with 200000 rows, phase in ms, 128 | 256 tile: n = 3072 7.5 | 7.9, 4096 12.6 | 12.2, 5000 18.5 | 16.7,
        // 6144 25.9 | 21.8, 7168 34.4 | 30.0)
        int tile = k_ >= 4096 ? 256 : 128;
        if (const char *ev = std::getenv("POGS_AMD_GRAM_TILE")) tile = std::atoi(ev) == 256 ? 256 : 128;
        const int urows = static_cast<int>(round_up(static_cast<size_t>((kdim + 4 * launches - 1) / (4 * launches)), 32));
        const int nunits = (kdim + urows - 1) / urows;
        const int npad = static_cast<int>(round_up(k_, tile));
        DevBuf<unsigned char> img(static_cast<size_t>(2) * (4 * urows) * npad * 2);
        unsigned char *H = img.p, *L = img.p + static_cast<size_t>(4 * urows) * npad * 2;
        ctx_.tmark("  gram: images allocated");
        GramF16PArgs gp{H, L, npad, k_, reinterpret_cast<float *>(G), ld, 4, urows, slab, 0, g.tile_map, scale16};
        gp.tile = tile;
        gp.flush_rows = kRows;
        DevBuf<int> tmap256;
        if (tile == 256) {
          gp.tile_map = nullptr;
          if (k_ > 16 * 256) {
            const std::vector<int> order = gram_tile_order(k_, 256);
            tmap256.alloc(order.size());
            POGS_HIP_CHECK(hipMemcpyAsync(tmap256.p, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice, s));
            ctx_.sync();   // order is a host temporary
            gp.tile_map = tmap256.p;
          }
        }
        if (const char *rep = std::getenv("POGS_AMD_GRAM_REPEAT")) {
          // telemetry aid (scripts/gpu_pmc_gram.sh): the first launch's product `rep` times back to back -- seconds
          // of nothing but gram_f16s_kernel for a power / clock sampler to look at; the real launches below
          // overwrite what these leave in the slabs (the first one does not accumulate)
          gp.nslabs = std::min(4, nunits);
          gp.accumulate = 0;
          launch_split_f16(reinterpret_cast<const float *>(A_.p), lda_, kdim, k_, 0, gp.nslabs * urows, npad, scale16, H, L, s);
          for (int r = std::max(0, std::atoi(rep)); r > 0; --r) launch_gram_f16p(gp, s);
          ctx_.sync();
        }
        for (int u0 = 0; u0 < nunits; u0 += 4) {
          gp.nslabs = std::min(4, nunits - u0);
          gp.accumulate = u0 > 0 ? 1 : 0;
          launch_split_f16(reinterpret_cast<const float *>(A_.p), lda_, kdim, k_, u0 * urows, gp.nslabs * urows, npad,
                           scale16, H, L, s);
          launch_gram_f16p(gp, s);
        }
        const int nslabs_used = std::min(4, nunits);
        ctx_.sync();   // img is freed at scope exit
        launch_sum_slabs<T>(G, slab, nslabs_used, G, ld, k_, s);
        ksplit = 0;   // skip the fp32 rounds below
        POGS_HIP_CHECK(hipMemsetAsync(G + slab, 0, 3 * slab * sizeof(T), s));
      }
      for (int ks = 0; ks < ksplit;) {
        const bool first = ks == 0;
        const int nb = std::min(first ? 4 : 3, ksplit - ks);
        g.ks0 = ks;
        g.ksplit = nb;
        g.C = first ? G : G + slab;
        launch_gemm<T>(tall_ || tmode_, tall_ || tmode_, true, g, s);   // K-major when the stored rows are the K index
        if (ksplit > 1) launch_sum_slabs<T>(G, slab, first ? nb : nb + 1, G, ld, k_, s);   // in place: slab 0 is G
        ks += nb;
      }
      if (ksplit > 1) POGS_HIP_CHECK(hipMemsetAsync(G + slab, 0, 3 * slab * sizeof(T), s));
      ctx_.sync();   // tmap is freed at scope exit
      if (multi_) {
        // G = sum over the ranks of A_k^T A_k: only the lower block-triangle travels (half the bytes
        // of the k x ld square), packed into the scratch slab, ONE all-reduce, unpacked in place
        const size_t cnt = packed_lower_count(k_, ld);
        if (ld % Vec16<T>::N == 0 && cnt <= slab) {
          launch_pack_lower<T>(G, ld, k_, tmp, false, s);
          ctx_.dist.allreduce(tmp, cnt, s);
          launch_pack_lower<T>(G, ld, k_, tmp, true, s);
          POGS_HIP_CHECK(hipMemsetAsync(tmp, 0, cnt * sizeof(T), s));
        } else {
          ctx_.dist.allreduce(G, slab, s);
        }
      }
      ctx_.stats.gram_ms = pt.stop_ms();
      ctx_.stats.gram_flops = static_cast<double>(kdim) * k_ * k_;
    }
    ctx_.tmark("gram");
    if (tall_) norm_est_gram(G, ld);
    else if (tmode_) norm_est_gram_wide(G, ld);
    ctx_.tmark("norm_est_gram");
    launch_add_diag<T>(G, ld, k_, static_cast<T>(1), s);                 // projector_direct_dense.cpp:118-119
    {
      PhaseTimer pt(s);
      cholesky_lower<T>(G, ld, k_, Wp_, ld, s);
      ctx_.stats.chol_ms = pt.stop_ms();
    }
    ctx_.tmark("cholesky");
    {
      PhaseTimer pt(s);
      trtri_lower<T>(G, ld, k_, Wp_, ld, tmp, s);
      launch_transpose<T>(Wp_, ld, k_, k_, Up_, ld, s);
      ctx_.stats.trtri_ms = pt.stop_ms();
    }
    ctx_.sync();
    ctx_.tmark("trtri");
  }

  // x_out-functor( U (W (rhs + add)) ): the two triangular products that replace
  // linalg_cholesky_svx (gsl_linalg.h:57-61).
  template <typename TailOp>
  void solve_gram(const T *rhs, const T *add, const TailOp &tail, double *tail_scalars) {
    hipStream_t s = ctx_.stream;
    StreamArgs<T> a;
    a.A = Wp_; a.lda = k_pad_; a.m = k_; a.n_pad = k_pad_;
    a.xin = rhs; a.xin_add = add; a.xin_nrm2 = nullptr;
    double *part = defer_sums_ ? ctx_.spart.p + sp_tail_off_ : ctx_.spart.p;
    a.col_partials = nullptr; a.scalar_partials = part;
    a.xl_scratch = xl_buf_.p;
    launch_stream<T, true, false, false, kLower, Tag>(planW_, a, GemvNOp<T>{1, 0, tvec_.p}, s);
    a.A = Up_;
    a.xin = tvec_.p; a.xin_add = nullptr;
    launch_stream<T, true, false, false, kUpper, Tag>(planW_, a, tail, s);
    if (TailOp::NS > 0 && tail_scalars) {
      SumJob j{part, stream_grid<true, false>(planW_, k_), TailOp::NS, tail_scalars};
      sum_now_or_later(j);
    }
  }

  // The same solve for the x update of the one-pass iteration, as ONE sweep over W = L^-1:
  // x = W^T (W (rhs + add)) -- the row dot t_i = W_i . r is handed back as the coefficient of row
  // i in the column sums of the very pass that computed it (DOT + ACC on the lower triangle, like
  // the symmetric product of the norm estimate), and the projection tail runs as the column
  // functor of the second stage.  200 MB instead of 400 MB per iteration at C2; U is not read.
  template <typename TailColOp>
  void solve_gram_onepass(const T *rhs, const T *add, const TailColOp &tail, double *tail_scalars) {
    hipStream_t s = ctx_.stream;
    StreamArgs<T> a;
    a.A = Wp_; a.lda = k_pad_; a.m = k_; a.n_pad = k_pad_;
    a.xin = rhs; a.xin_add = add; a.xin_nrm2 = nullptr;
    a.col_partials = colpart_.p;   // free here: its sums were reduced into rhs before the solve
    a.scalar_partials = ctx_.spart.p;
    a.xl_scratch = xl_buf_.p;
    launch_stream<T, true, true, false, kLower, Tag>(planW_, a, IdentRowOp<T>{}, s);
    double *sp = ctx_.spart.p + sp_tail_off_;
    launch_reduce_cols<T, TailColOp>(colpart_.p, stream_grid<true, true>(planW_, k_), k_pad_, tail, sp, s);
    if (TailColOp::NS > 0 && tail_scalars) {
      SumJob j{sp, reduce_cols_grid(k_pad_, Vec16<T>::N), TailColOp::NS, tail_scalars};
      sum_now_or_later(j);
    }
  }

  // ProjectorCgls::Project on the dense operator up to (not including) the final y = A x
  // (projector_cgls.cpp:59-75, cgls.h:200-323).  x: warm start in, projected x out.
  // Ax_warm: A times the warm start if the caller has it (inside the ADMM loop it is the
  // previous y), which replaces the two initial matrix passes by vector algebra.
  // yacc (with Ax_warm): receives A x by the recurrence A x_warm + sum alpha_k q_k, so that the caller
  // needs no product for y = A x (cg_fused.h); untouched when the loop takes no step.  Returns the
  // number of CG steps taken.
  int cgls_project(const T *x0, const T *y0, T *x, T tol, const T *Ax_warm, T *yacc = nullptr) {
    hipStream_t s = ctx_.stream;
    const int bx = vec_blocks(n_);
    const double shift = 1.0;
    const double kEps = std::numeric_limits<T>::epsilon();
    double *vp = ctx_.spart.p + static_cast<size_t>(planA_.grid_max) * 6;   // vector-kernel partials
    auto sum_vp = [&](int blocks, double *out) {
      SumJob j{vp, blocks, 1, out};
      launch_sum_jobs(&j, 1, s);
    };
    auto pass_n = [&](const T *xin, auto op, double *out, int ns) {   // DOT pass over A
      StreamArgs<T> a = argsA();
      a.xin = xin;
      ctx_.stream_timer.begin(s);
      launch_stream<T, true, false, false, kFull, Tag>(planA_, a, op, s);
      ctx_.stream_timer.end(s);
      if (ns > 0) sum_row_scalars(stream_grid<true, false>(planA_, m_), ns, out);
      ctx_.stats.matvecs += 1;
    };
    auto pass_t = [&](const T *rin) {   // s = A^T r - shift x, |s|^2
      gemv_t_partials(rin);
      finish_cols(CgSColOp<T>{x, static_cast<T>(shift), cg_s_.p, n_}, ctx_.S.p + kCgS2, 0, 0);
      ctx_.stats.matvecs += 1;
    };
    if (Ax_warm) {
      hipLaunchKernelGGL(sub_norm_kernel<T>, dim3(vec_blocks(m_)), dim3(kVecTpb), 0, s, m_, y0, Ax_warm, cg_r_.p, vp);
      hipLaunchKernelGGL(sub_norm_kernel<T>, dim3(bx), dim3(kVecTpb), 0, s, n_, x, x0, x, vp);
    } else {
      hipLaunchKernelGGL(sub_norm_kernel<T>, dim3(bx), dim3(kVecTpb), 0, s, n_, x, x0, x, vp);
      sum_vp(bx, ctx_.S.p + kCgX2);
      pass_n(x0, SubDotOp<T>{y0, cg_q_.p}, nullptr, 0);                 // b = y0 - A x0
      const double *S0 = ctx_.fetch_scalars();
      if (std::sqrt(S0[kCgX2]) > 0.0) pass_n(x, SubDotOp<T>{cg_q_.p, cg_r_.p}, nullptr, 0);   // r = b - A x
      else POGS_HIP_CHECK(hipMemcpyAsync(cg_r_.p, cg_q_.p, m_ * sizeof(T), hipMemcpyDeviceToDevice, s));
    }
    pass_t(cg_r_.p);
    hipLaunchKernelGGL(set_gamma_kernel, dim3(1), dim3(1), 0, s, ctx_.S.p, cg_.p);
    hipLaunchKernelGGL(cg_update_p_kernel<T>, dim3(bx), dim3(kVecTpb), 0, s, n_, cg_.p, cg_s_.p, cg_p_.p, vp, true);
    sum_vp(bx, ctx_.S.p + kCgP2);
    const double *S = ctx_.fetch_scalars();
    const double norms0 = std::sqrt(S[kCgS2]);
    const int maxit = (norms0 < kEps) ? 0 : 500;
    int steps = 0;
    for (int k = 0; k < maxit; ++k) {
      pass_n(cg_p_.p, CgQRowOp<T>{cg_q_.p}, ctx_.S.p + kCgQ2, 1);     // q = A p
      hipLaunchKernelGGL(cg_alpha_kernel, dim3(1), dim3(1), 0, s, ctx_.S.p, cg_.p, shift, kEps);
      const int bm = vec_blocks(m_);
      hipLaunchKernelGGL(cg_update_xr_kernel<T>, dim3(bx + bm), dim3(kVecTpb), 0, s, n_, m_, cg_.p, cg_p_.p, x,
                         cg_q_.p, cg_r_.p, vp, bx,
                         (yacc && Ax_warm) ? (k == 0 ? Ax_warm : static_cast<const T *>(yacc)) : static_cast<const T *>(nullptr),
                         (yacc && Ax_warm) ? yacc : static_cast<T *>(nullptr));
      sum_vp(bx, ctx_.S.p + kCgX2);
      pass_t(cg_r_.p);
      hipLaunchKernelGGL(cg_beta_kernel, dim3(1), dim3(1), 0, s, ctx_.S.p, cg_.p);
      hipLaunchKernelGGL(cg_update_p_kernel<T>, dim3(bx), dim3(kVecTpb), 0, s, n_, cg_.p, cg_s_.p, cg_p_.p, vp, false);
      sum_vp(bx, ctx_.S.p + kCgP2);
      S = ctx_.fetch_scalars();
      const double norms = std::sqrt(S[kCgS2]), normx = std::sqrt(S[kCgX2]);
      ++ctx_.stats.cg_iters;
      ++steps;
      if ((norms <= norms0 * static_cast<double>(tol)) || (normx * static_cast<double>(tol) >= 1.0)) break;
    }
    launch_axpby<T>(n_, static_cast<T>(1), x0, static_cast<T>(1), x, s);   // x += x0
    return steps;
  }

  // ---- per-solve -----------------------------------------------------------
  void load_problem(const FnHost &f, const FnHost &g, const SolveParams &p) {
    hipStream_t s = ctx_.stream;
    auto up = [&](FnBuf<T> &dst, const FnHost &src, int cnt) {
      POGS_HIP_CHECK(hipMemcpyAsync(dst.h.p, src.h, cnt * sizeof(int), hipMemcpyHostToDevice, s));
      POGS_HIP_CHECK(hipMemcpyAsync(dst.a.p, src.a, cnt * sizeof(T), hipMemcpyHostToDevice, s));
      POGS_HIP_CHECK(hipMemcpyAsync(dst.b.p, src.b, cnt * sizeof(T), hipMemcpyHostToDevice, s));
      POGS_HIP_CHECK(hipMemcpyAsync(dst.c.p, src.c, cnt * sizeof(T), hipMemcpyHostToDevice, s));
      POGS_HIP_CHECK(hipMemcpyAsync(dst.d.p, src.d, cnt * sizeof(T), hipMemcpyHostToDevice, s));
      POGS_HIP_CHECK(hipMemcpyAsync(dst.e.p, src.e, cnt * sizeof(T), hipMemcpyHostToDevice, s));
    };
    up(f_, f, m_);
    up(g_, g, n_);
    warn_negative_coeffs<T>(f, m_);   // prox_lib.h:62-69 (the clamp is in scale_objective_kernel)
    warn_negative_coeffs<T>(g, n_);
    // the one-pass kernel evaluates prox_f inline: only for the cheap base functions
    bool all_cheap = true, all_logistic = true;
    if (tmode_) {   // transposed storage: it is prox_g that runs inside the pass
      all_logistic = false;
      for (int j = 0; j < n_; ++j) all_cheap = all_cheap && is_cheap_prox(g.h[j]);
    } else {
      for (int i = 0; i < m_; ++i) {
        all_cheap = all_cheap && is_cheap_prox(f.h[i]);
        all_logistic = all_logistic && f.h[i] == kLogistic;
      }
    }
    pre_cheap_ = true;
    for (int i = 0; i < m_ && pre_cheap_; ++i) pre_cheap_ = is_cheap_prox(f.h[i]);
    for (int j = 0; j < n_ && pre_cheap_; ++j) pre_cheap_ = is_cheap_prox(g.h[j]);
    fused_now_ = fused_ok_ && (all_cheap || all_logistic);
    fused_logistic_ = fused_now_ && all_logistic && !all_cheap;
    // scaled copies: h and b shared with the originals (pogs.cpp:608-617)
    launch_scale_objective<T>(f_.view(), fs_.a.p, fs_.c.p, fs_.d.p, fs_.e.p, d_.p, m_, true, s);
    launch_scale_objective<T>(g_.view(), gs_.a.p, gs_.c.p, gs_.d.p, gs_.e.p, e_.p, n_, false, s);
    ctl_ = AdmmControl<T>();
    ctl_.abs_tol = static_cast<T>(p.abs_tol);
    ctl_.rel_tol = static_cast<T>(p.rel_tol);
    ctl_.max_iter = p.max_iter;
    ctl_.adaptive_rho = p.adaptive_rho;
    ctl_.gap_stop = p.gap_stop;
    ctl_.say_rho = p.verbose > 3 && ctx_.dist.rank() == 0;
    ctl_.rho0 = static_cast<T>(p.rho);
    ctl_.m_glob = ctx_.m_global;
    ctl_.n = n_;
    loaded_ = true;
    ctx_.sync();  // the host coefficient arrays may be freed by the caller afterwards
  }
  FnView<T> fview() const { return FnView<T>{f_.h.p, fs_.a.p, f_.b.p, fs_.c.p, fs_.d.p, fs_.e.p}; }
  FnView<T> gview() const { return FnView<T>{g_.h.p, gs_.a.p, g_.b.p, gs_.c.p, gs_.d.p, gs_.e.p}; }

  void cold_start() {  // z = 0, zt = 0 (pogs.cpp:71-73,121-126)
    hipStream_t s = ctx_.stream;
    for (int i = 0; i < 2; ++i) { x_[i].zero(s); y_[i].zero(s); }
    xt_.zero(s); yt_.zero(s); xtemp_.zero(s); ytemp_.zero(s);
    cur_ = 0;
    zt_scale_ = 1;
    spec_valid_ = false;
    exact_mode_ = false;
    colparts_ = 0;
    proj_count_ = 0;
    ctl_.reset();
  }

  // (x0, lambda0) -> (z, z~): z = [x0 / e | A (x0 / e)], z~ = -(1/rho) [-A^T (l0 / d) | l0 / d]
  // (pogs.cpp:144-156).  Consumed once.
  void apply_warm_start() {
    if (!warm_pending_) return;
    warm_pending_ = false;
    hipStream_t s = ctx_.stream;
    const T rho = ctl_.rho;
    POGS_HIP_CHECK(hipMemcpyAsync(xtemp_.p, warm_x_.data(), n_ * sizeof(T), hipMemcpyHostToDevice, s));
    POGS_HIP_CHECK(hipMemcpyAsync(ytemp_.p, warm_l_.data(), m_ * sizeof(T), hipMemcpyHostToDevice, s));
    launch_scale_by<T>(n_, static_cast<T>(1), xtemp_.p, e_.p, true, x_[cur_].p, s);            // x = x0 / e
    launch_scale_by<T>(m_, static_cast<T>(1), ytemp_.p, d_.p, true, yt_.p, s);                  // l0 / d
    if (tmode_) {
      t_mul_n(x_[cur_].p, nullptr, GemvNOp<T>{1, 0, y_[cur_].p}, nullptr);                      // y = A x
      t_mul_t(yt_.p, StoreColOp<T>{static_cast<T>(1) / rho, 0, xt_.p, n_}, nullptr);            // xt = A^T (l0/d) / rho
    } else {
      StreamArgs<T> a = argsA();
      a.xin = x_[cur_].p;
      launch_stream<T, true, false, false, kFull, Tag>(planA_, a, GemvNOp<T>{1, 0, y_[cur_].p}, s);   // y = A x
      gemv_t_partials(yt_.p);
      finish_cols(StoreColOp<T>{static_cast<T>(1) / rho, 0, xt_.p, n_}, nullptr, 0, 0);         // xt = A^T (l0/d) / rho
    }
    launch_scal<T>(yt_.p, static_cast<T>(-1) / rho, m_, s);                                     // yt = -(l0/d) / rho
    ctx_.sync();
    xtemp_.zero(s);
    ytemp_.zero(s);
  }

  // One ADMM iteration (pogs.cpp:253-470).  Returns true when the solve stops.
  bool iteration(unsigned verbose) {
    if (fused_now_) return tmode_ ? iteration_fused_wide(verbose) : iteration_fused(verbose);
    hipStream_t s = ctx_.stream;
    const int nw = cur_ ^ 1;
    const bool multi = multi_;
    // (1) prox + gap/tolerance sums + over-relaxation
    AdmmPreArgs<T> pa;
    pa.n_x = n_; pa.n_y = m_;
    pa.g = gview(); pa.f = fview();
    pa.x_cur = x_[cur_].p; pa.y_cur = y_[cur_].p;
    pa.xt = xt_.p; pa.yt = yt_.p;
    pa.zt_scale = zt_scale_;
    pa.x12 = x12_.p; pa.y12 = y12_.p;
    pa.xtemp = xtemp_.p; pa.ytemp = ytemp_.p;
    pa.rho = ctl_.rho; pa.alpha = ctl_.alpha(); pa.cheap = pre_cheap_;
    pa.partials = ctx_.spart.p;
    pa.blocks_x = pre_blocks(n_);
    launch_admm_pre<T>(pa, s);
    {
      SumJob j[2] = {{ctx_.spart.p, pa.blocks_x, 3, ctx_.S.p + kGapX},
                     {ctx_.spart.p + static_cast<size_t>(pa.blocks_x) * 3, pre_blocks(m_), 3, ctx_.S.p + kGapY}};
      launch_sum_jobs(j, 2, s);
    }
    if (use_cgls_) {
      // (2c) CGLS projector (projector_cgls.cpp:52-88), warm-started with the previous x (pogs.cpp:281)
      POGS_HIP_CHECK(hipMemcpyAsync(x_[nw].p, x_[cur_].p, n_ * sizeof(T), hipMemcpyDeviceToDevice, s));
      // y = A x (projector_cgls.cpp:78): from the CG recurrence y_warm + sum alpha_k q_k, except every
      // ysync_-th projection, which takes the product itself (cg_fused.h; POGS_AMD_YSYNC)
      const bool ysync = ysync_ <= 0 || (proj_count_ % static_cast<unsigned long long>(ysync_)) == 0;
      ++proj_count_;
      const int steps = cgls_project(xtemp_.p, ytemp_.p, x_[nw].p, ctl_.proj_tol(), y_[cur_].p, ysync ? nullptr : y_[nw].p);
      if (ysync) {
        StreamArgs<T> a = argsA();
        a.xin = x_[nw].p;
        ctx_.stream_timer.begin(s);
        launch_stream<T, true, false, false, kFull, Tag>(planA_, a,
                                                    ProjTailOp<T>{y_[nw].p, y_[cur_].p, y12_.p, ytemp_.p}, s);
        ctx_.stream_timer.end(s);
        sum_row_scalars(stream_grid<true, false>(planA_, m_), 2, ctx_.S.p + kDYprev2);
        ctx_.stats.matvecs += 1;
      } else {
        if (steps == 0) POGS_HIP_CHECK(hipMemcpyAsync(y_[nw].p, y_[cur_].p, m_ * sizeof(T), hipMemcpyDeviceToDevice, s));
        double *spy = ctx_.spart.p;   // (the prox step's partials there have been summed above)
        launch_admm_tail<T>(m_, y_[nw].p, y_[cur_].p, y12_.p, ytemp_.p, spy, s);
        SumJob jy{spy, vec_blocks(m_), 2, ctx_.S.p + kDYprev2};
        launch_sum_jobs(&jy, 1, s);
      }
      double *sp2 = ctx_.spart.p + static_cast<size_t>(planA_.grid_max) * 6;
      launch_admm_tail<T>(n_, x_[nw].p, x_[cur_].p, x12_.p, xtemp_.p, sp2, s);
      SumJob jt{sp2, vec_blocks(n_), 2, ctx_.S.p + kDXprev2};
      launch_sum_jobs(&jt, 1, s);
    } else if (tall_) {
      // (2) projection: x = (G + I)^{-1} (xtemp + A^T ytemp), y = A x   (projector_direct_dense.cpp:122-127)
      gemv_t_partials(ytemp_.p);
      finish_cols(StoreColOp<T>{1, 0, rhs_.p, n_}, nullptr, kGapY, 3);   // with shards: gap/norm sums ride along
      solve_gram_onepass(rhs_.p, xtemp_.p, ProjTailSumColOp<T>{x_[nw].p, x_[cur_].p, x12_.p, xtemp_.p, n_},
                         ctx_.S.p + kDXprev2);
      StreamArgs<T> a = argsA();
      a.xin = x_[nw].p;
      ctx_.stream_timer.begin(s);
      launch_stream<T, true, false, false, kFull, Tag>(planA_, a,
                                                  ProjTailOp<T>{y_[nw].p, y_[cur_].p, y12_.p, ytemp_.p}, s);
      ctx_.stream_timer.end(s);
      sum_row_scalars(stream_grid<true, false>(planA_, m_), 2, ctx_.S.p + kDYprev2);
      if (multi) ctx_.dist.allreduce(ctx_.S.p + kDYprev2, 2, s);
    } else {
      // (2') m <= n: t = (A A^T + I)^{-1} (A xtemp - ytemp); x = xtemp - A^T t; y = ytemp + t   (:128-135)
      t_mul_n(xtemp_.p, nullptr, ResidOp<T>{ytemp_.p, rhs_.p}, nullptr);
      solve_gram_onepass(rhs_.p, static_cast<const T *>(nullptr),
                         ProjTailAddColOp<T>{y_[nw].p, y_[cur_].p, y12_.p, ytemp_.p, tmpn_.p, m_},
                         ctx_.S.p + kDYprev2);
      t_mul_t(tmpn_.p, ProjTailColOp<T>{x_[nw].p, x_[cur_].p, x12_.p, xtemp_.p, n_}, ctx_.S.p + kDXprev2);
    }
    if (!use_cgls_) ctx_.stats.matvecs += 2;
    const double *S = ctx_.fetch_scalars();
    ctl_.set_pre(S);
    bool exact = false;
    if (ctl_.set_approx(S, nrmA_)) {
      // (3) exact residuals in one fused pass (pogs.cpp:352-376)
      StreamArgs<T> a = argsA();
      const int grid = stream_grid<true, true>(planA_, srows_);
      if (tmode_) {
        // stored rows = columns of A: the row dot with u = y12 + c yt - yprev is (A^T u)_j (dual
        // residual), the column sums weighted by x12_j are A x12 (primal residual)
        launch_exact_u<T>(m_, y12_.p, yt_.p, y_[cur_].p, zt_scale_, uvec_.p, s);
        a.xin = uvec_.p;
        ctx_.stream_timer.begin(s);
        launch_stream<T, true, true, false, kFull, Tag>(planA_, a, ExactTRowOp<T>{x12_.p, xt_.p, x_[cur_].p, zt_scale_}, s);
        ctx_.stream_timer.end(s);
        sum_row_scalars(grid, 1, ctx_.S.p + kExactS2);
        double *sp = ctx_.spart.p + static_cast<size_t>(planA_.grid_max) * 6;
        launch_reduce_cols<T, ExactTColOp<T>>(colpart_.p, grid, scols_pad_, ExactTColOp<T>{y12_.p, m_}, sp, s);
        SumJob j{sp, reduce_cols_grid(scols_pad_, Vec16<T>::N), 1, ctx_.S.p + kExactR2};
        launch_sum_jobs(&j, 1, s);
      } else {
        a.xin = x12_.p;
        ctx_.stream_timer.begin(s);
        launch_stream<T, true, true, false, kFull, Tag>(planA_, a,
                                                   ExactRowOp<T>{y12_.p, yt_.p, y_[cur_].p, zt_scale_}, s);
        ctx_.stream_timer.end(s);
        sum_row_scalars(grid, 1, ctx_.S.p + kExactR2);
        finish_cols(ExactColOp<T>{x12_.p, xt_.p, x_[cur_].p, zt_scale_, n_}, ctx_.S.p + kExactS2, kExactR2, 1, grid);
      }
      ctx_.stats.matvecs += 1;
      S = ctx_.fetch_scalars();
      ctl_.set_exact(S);
      exact = true;
    }
    const bool stop = ctl_.check_stop(exact);
    log_iteration(verbose);
    if (stop) return true;
    // (4) dual update already sits in xtemp/ytemp (ProjTailOp): swap roles.
    std::swap(xt_, xtemp_);
    std::swap(yt_, ytemp_);
    cur_ = nw;
    zt_scale_ = ctl_.adapt();
    ++ctl_.k;
    return false;
  }

  // One ADMM iteration as ONE pass over A (two when the previous pass could not
  // speculate).  Same arithmetic as iteration(): the pass that forms y_{k+1} = A x_{k+1}
  // also (a) evaluates the exact primal residual of iteration k with a second dot
  // product, and (b) assuming rho stays, runs the y half of iteration k+1's prox /
  // over-relaxation per row and accumulates A^T yhat_{k+1} and the exact-dual-residual
  // column sums for k+1.  If rho changes the speculative results are dropped.
  bool iteration_fused(unsigned verbose) {
    hipStream_t s = ctx_.stream;
    const int nw = cur_ ^ 1;
    const int by = pre_blocks(m_);
    const int gridC = pre_cols_grid(n_pad_, Vec16<T>::N);
    // every scalar sum of the iteration that needs no exchange runs in the launch that publishes
    // the scalar block; on row shards the y-side sums travel in the tail of the pack buffer
    struct DeferGuard {
      bool &flag;
      DeferGuard(bool &f, bool on) : flag(f) { flag = on; }
      ~DeferGuard() { flag = false; }
    } defer_guard(defer_sums_, true);
    const bool spec = spec_valid_;
    double *pre_part = ctx_.spart.p + sp_pre_off_;              // [by][3] y-half prox sums (non-speculated iterations)
    double *pc_part = pre_part + static_cast<size_t>(by) * 3;    // [gridC][4] pre_cols sums
    double *tail = pack_.p ? pack_.p + 2 * static_cast<size_t>(n_pad_) : nullptr;   // row shards: 6 scalars
    const size_t pack_count = 2 * static_cast<size_t>(n_pad_) + 6;
    // Lean iterations (fp64 on one GPU).  With 16-byte vectors of two doubles the two-dot / two-accumulator
    // pass has registers for ONE row per step and one workgroup per CU: nothing covers the row functor and
    // the barriers, and it streams at 5.8 TB/s where the one-dot / one-accumulator form with two rows per
    // step (Sinkhorn-Knopp's pass) reaches 7.0.  The exact residuals it carries are only ever USED once the
    // approximate bounds fall below 10 x the tolerances (pogs.cpp:346-352) -- late in a solve, 11 of C2's 106
    // iterations.  Until then the pass leaves them out; the first iteration whose bounds ask for them
    // evaluates them in a pass of its own (the two-pass iteration's, below), drops the speculation so that
    // the next iteration rebuilds both column-sum sets (PreAccOp), and from there on the full pass runs.
    // Same arithmetic for everything that is used, so the same trajectory.
    constexpr bool kLeanType = std::is_same<T, double>::value;
    const bool lean = kLeanType && !multi_ && !exact_mode_;
    int nparts = colparts_ > 0 ? colparts_ : stream2_grid<2>(planA_, m_);
    if (!spec) {
      // (A') y half of the prox / over-relaxation, then (B) the column sums A^T yhat_k and
      // A^T (y12 + c yt - yprev) in a pass of their own -- a speculated iteration has both from
      // the previous pass (and its y-half sums on the host: spec_gap_ stands in for kGapY)
      AdmmPreArgs<T> pa;
      pa.n_x = 0; pa.n_y = m_;
      pa.g = gview(); pa.f = fview();
      pa.x_cur = x_[cur_].p; pa.y_cur = y_[cur_].p;
      pa.xt = xt_.p; pa.yt = yt_.p;
      pa.zt_scale = zt_scale_;
      pa.x12 = x12_.p; pa.y12 = y12_.p;
      pa.xtemp = xtemp_.p; pa.ytemp = ytemp_.p;
      pa.rho = ctl_.rho; pa.alpha = ctl_.alpha(); pa.cheap = pre_cheap_;
      pa.partials = pre_part;
      pa.blocks_x = 0;
      launch_admm_pre<T>(pa, s);
      const SumJob jy{pre_part, by, 3, ctx_.S.p + kGapY};
      StreamArgs2<T> a2{A_.p, lda_, m_, n_pad_, nullptr, nullptr, colpart_.p, colpart2_.p, ctx_.spart.p};
      ctx_.stream_timer.begin(s);
      launch_stream2<T, 0, 2, Tag>(planA_, a2, PreAccOp<T>{ytemp_.p, y12_.p, yt_.p, y_[cur_].p, zt_scale_}, s);
      ctx_.stream_timer.end(s);
      nparts = stream2_grid<0>(planA_, m_);
      ctx_.stats.matvecs += 1;
      if (!multi_) {
        sum_now_or_later(jy);
      } else {
        PackJobs pj;
        pj.j[0] = jy; pj.j[1] = jy; pj.njobs = 1;
        launch_pack_cols<T>(colpart_.p, colpart2_.p, nparts, n_pad_, pack_.p, pj, s);
        ctx_.dist.allreduce(pack_.p, pack_count, s);
        ScalarOverlay ov;
        ov.src = tail; ov.slot[0] = kGapY; ov.n[0] = 3;
        launch_apply_overlay(ctx_.S.p, ov, s);   // the pack buffer is reused before this iteration's fetch
      }
    }
    // (C) ONE launch for the column side: both second stages, the x half of the prox, the exact
    // dual residual (fused_cols.h)
    {
      PreColsArgs<T> pc;
      // a lean pass (the previous iteration's, when this one is speculated) forms the first set only: the
      // second is stale pool memory then, and its sum -- published as S[kExactS2], never used -- reads as 0
      pc.part0 = colpart_.p; pc.part1 = (lean && spec) ? nullptr : colpart2_.p; pc.nparts = nparts;
      pc.tot64 = pack_.p;
      pc.n = n_; pc.n_pad = n_pad_;
      pc.g = gview();
      pc.x_cur = x_[cur_].p; pc.xt = xt_.p;
      pc.zt_scale = zt_scale_; pc.rho = ctl_.rho; pc.alpha = ctl_.alpha();
      pc.x12 = x12_.p; pc.xtemp = xtemp_.p; pc.rhs = rhs_.p;
      pc.partials = pc_part;
      launch_pre_cols<T>(pc, multi_, s);
      sum_now_or_later(SumJob{pc_part, gridC, 3, ctx_.S.p + kGapX, 4, 0});
      sum_now_or_later(SumJob{pc_part, gridC, 1, ctx_.S.p + kExactS2, 4, 3});
    }
    // x = (G + I)^{-1} (xtemp + A^T yhat)
    solve_gram_onepass(rhs_.p, xtemp_.p, ProjTailSumColOp<T>{x_[nw].p, x_[cur_].p, x12_.p, xtemp_.p, n_},
                       ctx_.S.p + kDXprev2);
    // (D) the pass over A
    {
      StreamArgs2<T> a2{A_.p, lda_, m_, n_pad_, x_[nw].p, x12_.p, colpart_.p, colpart2_.p, ctx_.spart.p};
      // speculate on the rho the adaptive rule is expected to choose (the previous
      // iteration's residuals stand in for this one's)
      ctl_.predict(&rho_pred_, &zs_pred_);
      ctx_.stream_timer.begin(s);
      int grid = stream2_grid<2>(planA_, m_);
      if (fused_logistic_) {
        FusedIterOp<T, true> op{y_[nw].p, y_[cur_].p, y12_.p, ytemp_.p, fview(), rho_pred_, ctl_.alpha(), zs_pred_,
                                y12s_.p, ytemps_.p};
        if constexpr (kLeanType) {
          if (lean) { launch_stream2<T, 1, 1, Tag>(planA_, a2, op, s); grid = stream2_grid<1, 1>(planA_, m_); }
          else launch_stream2<T, 2, 2, Tag>(planA_, a2, op, s);
        } else {
          launch_stream2<T, 2, 2, Tag>(planA_, a2, op, s);
        }
      } else {
        FusedIterOp<T, false> op{y_[nw].p, y_[cur_].p, y12_.p, ytemp_.p, fview(), rho_pred_, ctl_.alpha(), zs_pred_,
                                 y12s_.p, ytemps_.p};
        if constexpr (kLeanType) {
          if (lean) { launch_stream2<T, 1, 1, Tag>(planA_, a2, op, s); grid = stream2_grid<1, 1>(planA_, m_); }
          else launch_stream2<T, 2, 2, Tag>(planA_, a2, op, s);
        } else {
          launch_stream2<T, 2, 2, Tag>(planA_, a2, op, s);
        }
      }
      ctx_.stream_timer.end(s);
      colparts_ = grid;
      const SumJob jd{ctx_.spart.p, grid, 3, ctx_.S.p + kDYprev2, 6, 0};
      const SumJob js{ctx_.spart.p, grid, 3, ctx_.S.p + kSpecGapY, 6, 3};
      if (!multi_) {
        sum_now_or_later(jd);
        sum_now_or_later(js);
      } else {
        // ONE collective per iteration: this iteration's y-residual sums, the speculative column
        // sums and y-half sums of the next one -- a single fp64 buffer, one ncclAllReduce; the
        // scalars reach the host through the publishing launch (ScalarOverlay)
        PackJobs pj;
        pj.j[0] = jd; pj.j[1] = js; pj.njobs = 2;
        launch_pack_cols<T>(colpart_.p, colpart2_.p, grid, n_pad_, pack_.p, pj, s);
        ctx_.dist.allreduce(pack_.p, pack_count, s);
        ScalarOverlay ov;
        ov.src = tail;
        ov.slot[0] = kDYprev2; ov.n[0] = 3;
        ov.slot[1] = kSpecGapY; ov.n[1] = 3;
        ctx_.set_overlay(ov);
      }
      ctx_.stats.matvecs += 1;
    }
    // (E) host decisions (pogs.cpp:270-273, 342-394)
    double S[kNumSlots];
    std::memcpy(S, ctx_.fetch_scalars(), sizeof(S));
    if (spec)
      for (int q = 0; q < 3; ++q) S[kGapY + q] = spec_gap_[q];
    ctl_.set_pre(S);
    bool exact = false;
    bool drop_spec = false;
    if (ctl_.set_approx(S, nrmA_)) {
      if (lean) {
        // the exact residuals of THIS iteration in a pass of their own (pogs.cpp:352-376; the same launches
        // as the two-pass iteration's step (3)), column partials into the free second set
        StreamArgs<T> a = argsA();
        a.xin = x12_.p;
        a.col_partials = colpart2_.p;
        const int g1 = stream_grid<true, true>(planA_, srows_);
        ctx_.stream_timer.begin(s);
        launch_stream<T, true, true, false, kFull, Tag>(planA_, a, ExactRowOp<T>{y12_.p, yt_.p, y_[cur_].p, zt_scale_}, s);
        ctx_.stream_timer.end(s);
        sum_row_scalars(g1, 1, ctx_.S.p + kExactR2);
        finish_cols(ExactColOp<T>{x12_.p, xt_.p, x_[cur_].p, zt_scale_, n_}, ctx_.S.p + kExactS2, kExactR2, 1, g1, colpart2_.p);
        ctx_.stats.matvecs += 1;
        const double *S2 = ctx_.fetch_scalars();
        S[kExactR2] = S2[kExactR2];
        S[kExactS2] = S2[kExactS2];
        exact_mode_ = true;
        drop_spec = true;   // the next iteration rebuilds both column-sum sets (the second one was never formed)
      }
      ctl_.set_exact(S);
      exact = true;
    }
    const bool stop = ctl_.check_stop(exact);
    log_iteration(verbose);
    if (stop) return true;
    std::swap(xt_, xtemp_);
    std::swap(yt_, ytemp_);            // yt = ytilde_{k+1}
    cur_ = nw;
    zt_scale_ = ctl_.adapt();
    if (!drop_spec && ctl_.rho == rho_pred_ && zt_scale_ == zs_pred_) {
      std::swap(ytemp_, ytemps_);      // ytemp = speculative yhat_{k+1}
      std::swap(y12_, y12s_);          // y12 = speculative y12_{k+1}
      for (int q = 0; q < 3; ++q) spec_gap_[q] = S[kSpecGapY + q];
      spec_valid_ = true;
      ctx_.stats.reserved[0] += 1;     // speculation hits
    } else {
      spec_valid_ = false;
      ctx_.stats.reserved[1] += 1;     // misses
    }
    ++ctl_.k;
    return false;
  }

  // The one-pass iteration for m <= n on the transposed storage: the mirror image of
  // iteration_fused with x and y (g and f) trading places.  The pass over T = A^T that forms
  // x_{k+1} = xhat_k - A^T t_k (dot 0, t_k from the m x m solve) also evaluates the exact dual
  // residual of iteration k (dot 1 with u_k = y12 + c yt - y), finishes the x half of k, runs the
  // x half of k+1 per stored row with the predicted rho, and accumulates A xhat_{k+1}
  // (next right-hand side) and A x12_{k+1} (next exact primal residual).
  bool iteration_fused_wide(unsigned verbose) {
    hipStream_t s = ctx_.stream;
    const int nw = cur_ ^ 1;
    const int bx = pre_blocks(n_), by = pre_blocks(m_);
    // (A) prox / over-relaxation: y half always, x half unless already speculated
    AdmmPreArgs<T> pa;
    pa.n_x = spec_valid_ ? 0 : n_; pa.n_y = m_;
    pa.g = gview(); pa.f = fview();
    pa.x_cur = x_[cur_].p; pa.y_cur = y_[cur_].p;
    pa.xt = xt_.p; pa.yt = yt_.p;
    pa.zt_scale = zt_scale_;
    pa.x12 = x12_.p; pa.y12 = y12_.p;
    pa.xtemp = xtemp_.p; pa.ytemp = ytemp_.p;
    pa.rho = ctl_.rho; pa.alpha = ctl_.alpha(); pa.cheap = pre_cheap_;
    pa.partials = ctx_.spart.p;
    pa.blocks_x = spec_valid_ ? 0 : bx;
    launch_admm_pre<T>(pa, s);
    if (spec_valid_) {
      SumJob j{ctx_.spart.p, by, 3, ctx_.S.p + kGapY};
      launch_sum_jobs(&j, 1, s);
      POGS_HIP_CHECK(hipMemcpyAsync(ctx_.S.p + kGapX, ctx_.S.p + kSpecGapX, 3 * sizeof(double),
                                    hipMemcpyDeviceToDevice, s));
    } else {
      SumJob j[2] = {{ctx_.spart.p, bx, 3, ctx_.S.p + kGapX},
                     {ctx_.spart.p + static_cast<size_t>(bx) * 3, by, 3, ctx_.S.p + kGapY}};
      launch_sum_jobs(j, 2, s);
    }
    // u_k = y12 + c yt - y: the second dot vector of the pass (exact dual residual, pogs.cpp:366-369)
    launch_exact_u<T>(m_, y12_.p, yt_.p, y_[cur_].p, zt_scale_, uvec_.p, s);
    int nparts;
    if (spec_valid_) {
      nparts = stream2_grid<2>(planA_, srows_);
    } else {
      // (B) column sums A xhat_k and A x12_k
      StreamArgs2<T> a2{A_.p, lda_, srows_, scols_pad_, nullptr, nullptr, colpart_.p, colpart2_.p, ctx_.spart.p};
      ctx_.stream_timer.begin(s);
      launch_stream2<T, 0, 2, Tag>(planA_, a2, PreAcc2Op<T>{xtemp_.p, x12_.p}, s);
      ctx_.stream_timer.end(s);
      nparts = stream2_grid<0>(planA_, srows_);
      ctx_.stats.matvecs += 1;
    }
    // (C) t = (A A^T + I)^{-1} (A xhat - yhat), y = yhat + t; exact primal residual |A x12 - y12|
    {
      double *sp = ctx_.spart.p + static_cast<size_t>(planA_.grid_max) * 6;
      launch_reduce_cols<T, ResidColOp<T>>(colpart_.p, nparts, scols_pad_, ResidColOp<T>{ytemp_.p, rhs_.p, m_}, sp, s);
      launch_reduce_cols<T, ExactTColOp<T>>(colpart2_.p, nparts, scols_pad_, ExactTColOp<T>{y12_.p, m_}, sp, s);
      SumJob j{sp, reduce_cols_grid(scols_pad_, Vec16<T>::N), 1, ctx_.S.p + kExactR2};
      launch_sum_jobs(&j, 1, s);
    }
    solve_gram_onepass(rhs_.p, static_cast<const T *>(nullptr),
                       ProjTailAddColOp<T>{y_[nw].p, y_[cur_].p, y12_.p, ytemp_.p, tmpn_.p, m_}, ctx_.S.p + kDYprev2);
    // (D) the pass over T
    {
      StreamArgs2<T> a2{A_.p, lda_, srows_, scols_pad_, tmpn_.p, uvec_.p, colpart_.p, colpart2_.p, ctx_.spart.p};
      ctl_.predict(&rho_pred_, &zs_pred_);
      ctx_.stream_timer.begin(s);
      FusedIterOp<T, false, true> op{x_[nw].p, x_[cur_].p, x12_.p, xtemp_.p, gview(), rho_pred_, ctl_.alpha(), zs_pred_,
                                     x12s_.p, xtemps_.p, xt_.p, zt_scale_};
      launch_stream2<T, 2, 2, Tag>(planA_, a2, op, s);
      ctx_.stream_timer.end(s);
      const int grid = stream2_grid<2>(planA_, srows_);
      SumJob j[2] = {{ctx_.spart.p, grid, 3, ctx_.S.p + kDXprev2, 6, 0},
                     {ctx_.spart.p, grid, 3, ctx_.S.p + kSpecGapX, 6, 3}};
      launch_sum_jobs(j, 2, s);
      ctx_.stats.matvecs += 1;
    }
    // (E) host decisions (pogs.cpp:270-273, 342-394)
    const double *S = ctx_.fetch_scalars();
    ctl_.set_pre(S);
    bool exact = false;
    if (ctl_.set_approx(S, nrmA_)) {
      ctl_.set_exact(S);
      exact = true;
    }
    const bool stop = ctl_.check_stop(exact);
    log_iteration(verbose);
    if (stop) return true;
    std::swap(xt_, xtemp_);            // xt = xtilde_{k+1}
    std::swap(yt_, ytemp_);
    cur_ = nw;
    zt_scale_ = ctl_.adapt();
    if (ctl_.rho == rho_pred_ && zt_scale_ == zs_pred_) {
      std::swap(xtemp_, xtemps_);      // xtemp = speculative xhat_{k+1}
      std::swap(x12_, x12s_);          // x12 = speculative x12_{k+1}
      spec_valid_ = true;
      ctx_.stats.reserved[0] += 1;
    } else {
      spec_valid_ = false;
      ctx_.stats.reserved[1] += 1;
    }
    ++ctl_.k;
    return false;
  }

  // sum f(y12) + sum g(x12) at the current prox point (pogs.cpp:385, 473)
  double eval_objective() {
    hipStream_t s = ctx_.stream;
    const int by = vec_blocks(m_), bx = vec_blocks(n_);
    const bool was_deferring = defer_sums_;
    defer_sums_ = false;
    launch_func_eval<T>(m_, fview(), y12_.p, ctx_.spart.p, s);
    launch_func_eval<T>(n_, gview(), x12_.p, ctx_.spart.p + by, s);
    SumJob j[2] = {{ctx_.spart.p, by, 1, ctx_.S.p + kFvalF}, {ctx_.spart.p + by, bx, 1, ctx_.S.p + kFvalG}};
    launch_sum_jobs(j, 2, s);
    if (multi_) ctx_.dist.allreduce(ctx_.S.p + kFvalF, 1, s);
    const double *S = ctx_.fetch_scalars();
    defer_sums_ = was_deferring;
    return static_cast<double>(static_cast<T>(S[kFvalF]) + static_cast<T>(S[kFvalG]));
  }
  // the reference's per-iteration line (pogs.cpp:382-388); every rank evaluates (the objective
  // sum is a collective on row shards), rank 0 prints
  void log_iteration(unsigned verbose) {
    if (!wants_iter_line(verbose, ctl_)) return;
    const double obj = eval_objective();
    if (ctx_.dist.rank() == 0) print_iter_line(ctl_, obj);
  }

  // optval, status, un-scaling, copy out (pogs.cpp:473-482, 510-518, 567-570).
  int epilogue(void *x, void *y, void *l, void *mu, double *optval) {
    hipStream_t s = ctx_.stream;
    const int by = vec_blocks(m_), bx = vec_blocks(n_);
    launch_func_eval<T>(m_, fview(), y12_.p, ctx_.spart.p, s);
    launch_func_eval<T>(n_, gview(), x12_.p, ctx_.spart.p + by, s);
    SumJob j[2] = {{ctx_.spart.p, by, 1, ctx_.S.p + kFvalF}, {ctx_.spart.p + by, bx, 1, ctx_.S.p + kFvalG}};
    launch_sum_jobs(j, 2, s);
    if (multi_) ctx_.dist.allreduce(ctx_.S.p + kFvalF, 1, s);
    UnscaleArgs<T> u;
    u.n_x = n_; u.n_y = m_;
    u.x12 = x12_.p; u.y12 = y12_.p; u.xt = xt_.p; u.yt = yt_.p;
    u.xprev = x_[cur_].p; u.yprev = y_[cur_].p; u.d = d_.p; u.e = e_.p;
    u.zt_scale = zt_scale_; u.rho = ctl_.rho;
    u.x_out = xout_.p; u.y_out = yout_.p; u.l_out = lout_.p; u.mu_out = muout_.p;
    launch_unscale<T>(u, s);
    POGS_HIP_CHECK(hipMemcpyAsync(x, xout_.p, n_ * sizeof(T), hipMemcpyDeviceToHost, s));
    POGS_HIP_CHECK(hipMemcpyAsync(y, yout_.p, m_ * sizeof(T), hipMemcpyDeviceToHost, s));
    POGS_HIP_CHECK(hipMemcpyAsync(l, lout_.p, m_ * sizeof(T), hipMemcpyDeviceToHost, s));
    if (mu) POGS_HIP_CHECK(hipMemcpyAsync(mu, muout_.p, n_ * sizeof(T), hipMemcpyDeviceToHost, s));
    const double *S = ctx_.fetch_scalars();
    *optval = static_cast<double>(static_cast<T>(S[kFvalF]) + static_cast<T>(S[kFvalG]));
    // the polled sequence word says the kernels are done; the D2H copies into the caller's
    // (pageable) buffers are only guaranteed complete after a synchronizing call
    POGS_HIP_CHECK(hipStreamSynchronize(s));
    return ctl_.status();
  }

  void collect_stream_timer() {
    ctx_.stats.reserved[2] = static_cast<double>(ctx_.dist.collectives());   // all-reduce calls since creation
    ctx_.stats.reserved[3] = static_cast<double>(ctx_.dist.comm_nranks());   // ranks as the communicator reports them
    if (!ctx_.stream_timer.enabled()) return;
    unsigned long long cnt = 0;
    ctx_.stats.stream_ms += ctx_.stream_timer.collect_ms(&cnt);
    ctx_.stats.stream_launches += cnt;
    ctx_.stats.stream_bytes += static_cast<double>(cnt) * m_ * n_ * sizeof(T);
  }

  Ctx ctx_;
  int m_ = 0, n_ = 0, n_pad_ = 0, k_ = 0, k_pad_ = 0;
  bool tall_ = true, multi_ = false, use_cgls_ = false;
  double amax_ = 0;             // max |entry| of the equilibrated matrix (fp16 scaling of the Gram product)
  bool defer_sums_ = false;     // inside iteration_fused on one GPU: sums wait for the closing launch
  size_t sp_pre_off_ = 0, sp_tail_off_ = 0;   // regions of ctx_.spart (alloc_state)
  double spec_gap_[3] = {0, 0, 0};            // y-half sums of the speculated iteration (host copy)
  bool tmode_ = false;          // A^T is what is stored (m <= n, direct projector)
  int m_pad_ = 0, srows_ = 0, scols_pad_ = 0;   // stored rows / padded stored row length
  DevBuf<T> xl_buf_;            // windowed passes: partial row dots per window + the coefficient vector
  DevBuf<T> uvec_;              // tmode_: y12 + c yt - yprev for the exact-residual pass
  DevBuf<T> x12s_, xtemps_;     // tmode_, one-pass iteration: speculative x12_{k+1}, xhat_{k+1}
  DevBuf<T> cg_p_, cg_s_, cg_q_, cg_r_;
  DevBuf<double> cg_;
  size_t lda_ = 0;
  StreamPlan planA_, planW_;
  const T *A_src_ = nullptr;    // upload() .. equilibrate(): the caller's device buffer standing in for A_
  bool pre_cheap_ = false;      // every f_i, g_j has a few-operation prox (admm_pre_kernel inlines it)
  T *Wp_ = nullptr, *Up_ = nullptr;   // views into fac_
  DevBuf<T> A_, fac_, d_, e_, colpart_, colpart2_, y12s_, ytemps_;
  DevBuf<double> pack_;         // row shards, one-pass iteration: the packed all-reduce buffer
  bool fused_ok_ = false, fused_now_ = false, fused_logistic_ = false, spec_valid_ = false;
  // fp64, one GPU, m > n: the pass leaves the exact residuals out (one dot product, one accumulator, two
  // rows per step) until the approximate bounds first ask for them (iteration_fused)
  bool exact_mode_ = false;
  int colparts_ = 0;            // workgroups (= column partials) of the pass that filled colpart_ last
  bool warm_pending_ = false;
  std::vector<T> warm_x_, warm_l_;
  T rho_pred_ = 1, zs_pred_ = 1;
  int ysync_ = 16;                       // dense CGLS option: y = A x explicitly every ysync_-th projection (0: always)
  unsigned long long proj_count_ = 0;
  DevBuf<T> x_[2], y_[2], xt_, yt_, xtemp_, ytemp_, x12_, y12_, rhs_, tvec_, tmpn_;
  DevBuf<T> xout_, yout_, lout_, muout_;
  FnBuf<T> f_, g_, fs_, gs_;
  AdmmControl<T> ctl_;
  bool loaded_ = false;   // load_problem has run: f, g and the control block are valid
  int cur_ = 0;
  T zt_scale_ = 1;
  T nrmA_ = 0;
};

}  // namespace

// One translation unit per arithmetic type (dense_f32.hip, dense_f64.hip): the HIP runtime loads a
// code object as a whole at its first kernel launch, and the row kernels come in ~700 variants
// per type (plan x mode x functor), so a float solve should not pay for the double kernels.
template <typename T, typename Tag>
SolverBase *make_dense_solver_t(int ord, size_t m, size_t n, const void *A, int mem, const PogsAmdOptions *opt,
                                const PogsAmdDist *dist) {
  return new DenseSolver<T, Tag>(ord, m, n, A, mem, opt, dist);
}

}  // namespace pogs_amd
