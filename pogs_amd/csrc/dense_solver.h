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
          dev::prod_acc(acc[0], val, val);
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
  void upload(int ord, const void *A, int mem);

  void alloc_state();

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
  void equilibrate();

  // Norm2Est (equil_helper.h:107-135), one fused pass per power iteration.
  void norm_est();

  // Norm2Est (equil_helper.h:107-135) for m > n, run on G = A^T A instead of A: the
  // iteration x <- A^T (A x) is x <- G x and |A x|^2 = x^T G x, so each power step reads
  // the n x n lower triangle (0.2 GB at C2) instead of A (4 GB).  Same start vector, same
  // normalisation and stopping rule; G is already summed over shards.
  void norm_est_gram(T *G, size_t ld);

  // Norm2Est for m <= n on G = A A^T.  With y = A x^ (x^ the normalised iterate) the reference's
  // step x' = A^T (A x^), est = |x'| / |A x^|, x^ <- x' / |x'| reads  |x'|^2 = y^T G y,
  // |A x^| = |y|,  y <- G y / |x'|:  after ONE product with A (y0 = A x0, x0 the same random
  // start vector, un-normalised as in equil_helper.h:113-121) every power step is a symmetric
  // m x m product instead of two passes over A.
  void norm_est_gram_wide(T *G, size_t ld);

  // ProjectorDirect::Init + the first-call factorisation (s = 1 always,
  // pogs.cpp:293,296): G = A^T A (m > n) or A A^T (m <= n) on MFMA tiles,
  // L L^T = G + I, W = L^{-1}, U = W^T.
  void factor();

  // x_out-functor( U (W (rhs + add)) ): the two triangular products that replace
  // linalg_cholesky_svx (gsl_linalg.h:57-61).
  template <typename TailOp>
  void solve_gram(const T *rhs, const T *add, const TailOp &tail, double *tail_scalars);

  // The same solve for the x update of the one-pass iteration, as ONE sweep over W = L^-1:
  // x = W^T (W (rhs + add)) -- the row dot t_i = W_i . r is handed back as the coefficient of row
  // i in the column sums of the very pass that computed it (DOT + ACC on the lower triangle, like
  // the symmetric product of the norm estimate), and the projection tail runs as the column
  // functor of the second stage.  200 MB instead of 400 MB per iteration at C2; U is not read.
  template <typename TailColOp>
  void solve_gram_onepass(const T *rhs, const T *add, const TailColOp &tail, double *tail_scalars);

  // ProjectorCgls::Project on the dense operator up to (not including) the final y = A x
  // (projector_cgls.cpp:59-75, cgls.h:200-323).  x: warm start in, projected x out.
  // Ax_warm: A times the warm start if the caller has it (inside the ADMM loop it is the
  // previous y), which replaces the two initial matrix passes by vector algebra.
  // yacc (with Ax_warm): receives A x by the recurrence A x_warm + sum alpha_k q_k, so that the caller
  // needs no product for y = A x (cg_fused.h); untouched when the loop takes no step.  Returns the
  // number of CG steps taken.
  int cgls_project(const T *x0, const T *y0, T *x, T tol, const T *Ax_warm, T *yacc = nullptr);

  // ---- per-solve -----------------------------------------------------------
  void load_problem(const FnHost &f, const FnHost &g, const SolveParams &p);
  FnView<T> fview() const { return FnView<T>{f_.h.p, fs_.a.p, f_.b.p, fs_.c.p, fs_.d.p, fs_.e.p}; }
  FnView<T> gview() const { return FnView<T>{g_.h.p, gs_.a.p, g_.b.p, gs_.c.p, gs_.d.p, gs_.e.p}; }

  void cold_start();  // z = 0, zt = 0 (pogs.cpp:71-73,121-126)

  // (x0, lambda0) -> (z, z~): z = [x0 / e | A (x0 / e)], z~ = -(1/rho) [-A^T (l0 / d) | l0 / d]
  // (pogs.cpp:144-156).  Consumed once.
  void apply_warm_start();

  // One ADMM iteration (pogs.cpp:253-470).  Returns true when the solve stops.
  bool iteration(unsigned verbose);

  // One ADMM iteration as ONE pass over A (two when the previous pass could not
  // speculate).  Same arithmetic as iteration(): the pass that forms y_{k+1} = A x_{k+1}
  // also (a) evaluates the exact primal residual of iteration k with a second dot
  // product, and (b) assuming rho stays, runs the y half of iteration k+1's prox /
  // over-relaxation per row and accumulates A^T yhat_{k+1} and the exact-dual-residual
  // column sums for k+1.  If rho changes the speculative results are dropped.
  bool iteration_fused(unsigned verbose);

  // The one-pass iteration for m <= n on the transposed storage: the mirror image of
  // iteration_fused with x and y (g and f) trading places.  The pass over T = A^T that forms
  // x_{k+1} = xhat_k - A^T t_k (dot 0, t_k from the m x m solve) also evaluates the exact dual
  // residual of iteration k (dot 1 with u_k = y12 + c yt - y), finishes the x half of k, runs the
  // x half of k+1 per stored row with the predicted rho, and accumulates A xhat_{k+1}
  // (next right-hand side) and A x12_{k+1} (next exact primal residual).
  bool iteration_fused_wide(unsigned verbose);

  // sum f(y12) + sum g(x12) at the current prox point (pogs.cpp:385, 473)
  double eval_objective();
  // the reference's per-iteration line (pogs.cpp:382-388); every rank evaluates (the objective
  // sum is a collective on row shards), rank 0 prints
  void log_iteration(unsigned verbose);

  // optval, status, un-scaling, copy out (pogs.cpp:473-482, 510-518, 567-570).
  int epilogue(void *x, void *y, void *l, void *mu, double *optval);

  void collect_stream_timer();

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

// Member definitions, by seam (included HERE, inside the namespaces of the class): the one-time setup, the projector,
// the iterations for m > n, the m <= n mirror.
#include "dense_setup.h"
#include "dense_factor.h"
#include "dense_iter.h"
#include "dense_wide.h"

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
