// DenseSolver: per-solve state and the ADMM iterations for m > n (PogsImplementation::Solve, src/cpu/pogs.cpp:91-581):
// the two-pass iteration and the one-pass iteration with rho speculation; objective, epilogue.
// Member definitions of the class template declared in dense_solver.h, which includes this file once, right after the
// class, inside its namespaces (no include guard, no namespace of its own).

// ---- per-solve -----------------------------------------------------------
template <typename T, typename Tag>
void DenseSolver<T, Tag>::load_problem(const FnHost &f, const FnHost &g, const SolveParams &p) {
  hipStream_t s = ctx_.stream;
  upload_fn<T>(f_, f, m_, s);
  upload_fn<T>(g_, g, n_, s);
  warn_negative_coeffs<T>(f, m_);   // prox_lib.h:62-69 (the clamp is in scale_objective_kernel)
  warn_negative_coeffs<T>(g, n_);
  // the one-pass kernel evaluates prox_f inline: only for the cheap base functions
  bool all_cheap = true, all_logistic = true;
  if (tmode_) {   // transposed storage: it is prox_g that runs inside the pass
    all_logistic = false;
    all_cheap = all_h(g, n_, [](int h) { return is_cheap_prox(h); });
  } else {
    all_cheap = all_h(f, m_, [](int h) { return is_cheap_prox(h); });
    all_logistic = all_h(f, m_, [](int h) { return h == kLogistic; });
  }
  pre_cheap_ = all_h(f, m_, [](int h) { return is_cheap_prox(h); }) && all_h(g, n_, [](int h) { return is_cheap_prox(h); });
  fused_now_ = fused_ok_ && (all_cheap || all_logistic);
  fused_logistic_ = fused_now_ && all_logistic && !all_cheap;
#ifndef POGS_NV5_CHEAP_BPC   // (0: three per CU for every solve at 256 x 5, as before round 6 -- A / B builds)
#define POGS_NV5_CHEAP_BPC 2
#endif
  planA_.bpc_override = (POGS_NV5_CHEAP_BPC > 0 && fused_now_ && !fused_logistic_ && !tmode_ && std::is_same<T, float>::value &&
                         planA_.tpb == 256 && planA_.nv == 5) ? POGS_NV5_CHEAP_BPC : 0;
  planA_.pf = POGS_NV5_PREFETCH != 0 && fused_logistic_ && !tmode_ && std::is_same<T, float>::value && planA_.tpb == 256 &&
              planA_.nv == 5;
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

template <typename T, typename Tag>
void DenseSolver<T, Tag>::cold_start() {  // z = 0, zt = 0 (pogs.cpp:71-73,121-126)
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
template <typename T, typename Tag>
void DenseSolver<T, Tag>::apply_warm_start() {
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
template <typename T, typename Tag>
bool DenseSolver<T, Tag>::iteration(unsigned verbose) {
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
template <typename T, typename Tag>
bool DenseSolver<T, Tag>::iteration_fused(unsigned verbose) {
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
#ifndef POGS_LEAN_F32_NV5   // (experiment switch, scripts/build_variant.py: the lean form in fp32 at 256 x 5, C3's shape)
#define POGS_LEAN_F32_NV5 0
#endif
  constexpr bool kLeanType = std::is_same<T, double>::value || (POGS_LEAN_F32_NV5 != 0 && Tag::has(256, 5) && !Tag::windows);
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

// sum f(y12) + sum g(x12) at the current prox point (pogs.cpp:385, 473)
template <typename T, typename Tag>
double DenseSolver<T, Tag>::eval_objective() {
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
template <typename T, typename Tag>
void DenseSolver<T, Tag>::log_iteration(unsigned verbose) {
  if (!wants_iter_line(verbose, ctl_)) return;
  const double obj = eval_objective();
  if (ctx_.dist.rank() == 0) print_iter_line(ctl_, obj);
}

// optval, status, un-scaling, copy out (pogs.cpp:473-482, 510-518, 567-570).
template <typename T, typename Tag>
int DenseSolver<T, Tag>::epilogue(void *x, void *y, void *l, void *mu, double *optval) {
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

template <typename T, typename Tag>
void DenseSolver<T, Tag>::collect_stream_timer() {
  ctx_.stats.reserved[2] = static_cast<double>(ctx_.dist.collectives());   // all-reduce calls since creation
  ctx_.stats.reserved[3] = static_cast<double>(ctx_.dist.comm_nranks());   // ranks as the communicator reports them
  if (!ctx_.stream_timer.enabled()) return;
  unsigned long long cnt = 0;
  ctx_.stats.stream_ms += ctx_.stream_timer.collect_ms(&cnt);
  ctx_.stats.stream_launches += cnt;
  ctx_.stats.stream_bytes += static_cast<double>(cnt) * m_ * n_ * sizeof(T);
}
