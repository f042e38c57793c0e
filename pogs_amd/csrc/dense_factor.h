// DenseSolver: the projector -- Gram product, Cholesky, W = L^-1 (ProjectorDirect::Init,
// src/cpu/projector/projector_direct_dense.cpp:45-84), the solves of ::Project (:87-175) and the matrix-free CGLS option
// (projector_cgls.cpp:52-88).
// Member definitions of the class template declared in dense_solver.h, which includes this file once, right after the
// class, inside its namespaces (no include guard, no namespace of its own).

// ProjectorDirect::Init + the first-call factorisation (s = 1 always,
// pogs.cpp:293,296): G = A^T A (m > n) or A A^T (m <= n) on MFMA tiles,
// L L^T = G + I, W = L^{-1}, U = W^T.
template <typename T, typename Tag>
void DenseSolver<T, Tag>::factor() {
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
template <typename T, typename Tag>
template <typename TailOp>
void DenseSolver<T, Tag>::solve_gram(const T *rhs, const T *add, const TailOp &tail, double *tail_scalars) {
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
template <typename T, typename Tag>
template <typename TailColOp>
void DenseSolver<T, Tag>::solve_gram_onepass(const T *rhs, const T *add, const TailColOp &tail, double *tail_scalars) {
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
template <typename T, typename Tag>
int DenseSolver<T, Tag>::cgls_project(const T *x0, const T *y0, T *x, T tol, const T *Ax_warm, T *yacc) {
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
