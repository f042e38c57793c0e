// DenseSolver: the one-time setup -- upload, work vectors, equilibration (MatrixDense::Equil,
// src/cpu/matrix/matrix_dense.cpp:116-200 with equil_helper.h:107-164) and the norm estimate.
// Member definitions of the class template declared in dense_solver.h, which includes this file once, right after the
// class, inside its namespaces (no include guard, no namespace of its own).

// ---- setup ---------------------------------------------------------------
template <typename T, typename Tag>
void DenseSolver<T, Tag>::upload(int ord, const void *A, int mem) {
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

template <typename T, typename Tag>
void DenseSolver<T, Tag>::alloc_state() {
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

// MatrixDense::Equil without materialising A.^2 (matrix_dense.cpp:116-200,
// equil_helper.h:140-164): 51 passes over A instead of 100.
template <typename T, typename Tag>
void DenseSolver<T, Tag>::equilibrate() {
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
template <typename T, typename Tag>
void DenseSolver<T, Tag>::norm_est() {
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
template <typename T, typename Tag>
void DenseSolver<T, Tag>::norm_est_gram(T *G, size_t ld) {
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
template <typename T, typename Tag>
void DenseSolver<T, Tag>::norm_est_gram_wide(T *G, size_t ld) {
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
