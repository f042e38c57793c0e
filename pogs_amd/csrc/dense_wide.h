// DenseSolver: the one-pass iteration for m <= n on transposed storage (projector_direct_dense.cpp:128-135, A A^T).
// Member definitions of the class template declared in dense_solver.h, which includes this file once, right after the
// class, inside its namespaces (no include guard, no namespace of its own).

// The one-pass iteration for m <= n on the transposed storage: the mirror image of
// iteration_fused with x and y (g and f) trading places.  The pass over T = A^T that forms
// x_{k+1} = xhat_k - A^T t_k (dot 0, t_k from the m x m solve) also evaluates the exact dual
// residual of iteration k (dot 1 with u_k = y12 + c yt - y), finishes the x half of k, runs the
// x half of k+1 per stored row with the predicted rho, and accumulates A xhat_{k+1}
// (next right-hand side) and A x12_{k+1} (next exact primal residual).
template <typename T, typename Tag>
bool DenseSolver<T, Tag>::iteration_fused_wide(unsigned verbose) {
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
