// Element-wise ADMM kernels (see vec_kernels.h).
#include "vec_kernels.h"

#include "reduce.h"
#include "stream.h"

namespace pogs_amd {

namespace {

template <typename T>
__global__ void __launch_bounds__(kVecTpb) scale_objective_kernel(FnView<T> fn, T *a, T *c, T *d, T *e,
                                                                  const T *scale, int n, bool divide) {
  const int i = blockIdx.x * kVecTpb + threadIdx.x;
  if (i >= n) return;
  const T s = scale[i];
  const T zero = 0;
  T ai = fn.a[i], ci = fn.c[i], di = fn.d[i], ei = fn.e[i];
  ci = ci < zero ? zero : ci;
  ei = ei < zero ? zero : ei;
  if (divide) {
    ai /= s; di /= s; ei /= s * s;
  } else {
    ai *= s; di *= s; ei *= s * s;
  }
  a[i] = ai; c[i] = ci; d[i] = di; e[i] = ei;
}

template <typename T, bool CHEAP>
__global__ void __launch_bounds__(kVecTpb) admm_pre_kernel(AdmmPreArgs<T> a) {
  __shared__ double s_red[3 * (kVecTpb / 64)];
  const bool is_x = static_cast<int>(blockIdx.x) < a.blocks_x;
  const int blk = is_x ? blockIdx.x : blockIdx.x - a.blocks_x;
  const int n = is_x ? a.n_x : a.n_y;
  const FnView<T> fn = is_x ? a.g : a.f;
  const FnUniform<T> uni = is_x ? a.ug : a.uf;
  const T *cur = is_x ? a.x_cur : a.y_cur;
  const T *zt = is_x ? a.xt : a.yt;
  T *z12 = is_x ? a.x12 : a.y12;
  T *ztemp = is_x ? a.xtemp : a.ytemp;
  if (a.cg_reset && blockIdx.x == 0 && threadIdx.x == 0) {
    a.cg_reset[0] = 0.0;
    a.cg_reset[1] = 0.0;
  }
  T *aux = is_x ? a.x_aux : a.y_aux;
  double acc[3] = {0.0, 0.0, 0.0};
  // all loads of the kPreU chunks first: the kernel is a latency chain of eight streams per element
  T prev[kPreU], ztv[kPreU], fa[kPreU], fb[kPreU], fc[kPreU], fd[kPreU], fe[kPreU];
  int fh[kPreU];
#pragma unroll
  for (int u = 0; u < kPreU; ++u) {
    const int i = (blk * kPreU + u) * kVecTpb + threadIdx.x;
    if (i < n) {
      prev[u] = cur[i]; ztv[u] = zt[i];
      fa[u] = fn.a[i]; fb[u] = fn.b[i];
      // (uniform per workgroup) an array that holds one value throughout is not read
      fh[u] = (uni.mask & 1) ? uni.h : fn.h[i];
      fc[u] = (uni.mask & 2) ? uni.c : fn.c[i];
      fd[u] = (uni.mask & 4) ? uni.d : fn.d[i];
      fe[u] = (uni.mask & 8) ? uni.e : fn.e[i];
    }
  }
#pragma unroll
  for (int u = 0; u < kPreU; ++u) {
    const int i = (blk * kPreU + u) * kVecTpb + threadIdx.x;
    if (i < n) {
      const T zs = a.zt_scale * ztv[u];
      const T v = prev[u] - zs;                                     // pogs.cpp:257
      const T h = CHEAP ? dev::ProxEvalCheap(fh[u], fa[u], fb[u], fc[u], fd[u], fe[u], v, a.rho)
                        : dev::ProxEval(fh[u], fa[u], fb[u], fc[u], fd[u], fe[u], v, a.rho);  // :263
      const T w = v - h;                                            // :267
      z12[i] = h;
      const T zt_new = zs + a.alpha * h + (static_cast<T>(1) - a.alpha) * prev[u];   // :276-278
      ztemp[i] = zt_new;
      if (aux) aux[i] = is_x ? prev[u] - zt_new : zt_new - prev[u];
      dev::prod_acc(acc[0], w, h);                         // :268
      dev::prod_acc(acc[1], w, w);
      dev::prod_acc(acc[2], h, h);
    }
  }
  dev::block_sum<3, kVecTpb>(acc, s_red);
  if (threadIdx.x == 0) {
    double *out = a.partials + static_cast<size_t>(blockIdx.x) * 3;
    out[0] = acc[0]; out[1] = acc[1]; out[2] = acc[2];
  }
}

template <typename T>
__global__ void __launch_bounds__(kVecTpb) admm_tail_kernel(int n, const T *znew, const T *zprev,
                                                            const T *z12, T *ztemp, double *partials) {
  __shared__ double s_red[2 * (kVecTpb / 64)];
  const int i = blockIdx.x * kVecTpb + threadIdx.x;
  double acc[2] = {0.0, 0.0};
  if (i < n) {
    const T zn = znew[i];
    const T p = zprev[i] - zn, q = z12[i] - zn;
    acc[0] = static_cast<double>(p) * p;
    acc[1] = static_cast<double>(q) * q;
    ztemp[i] -= zn;
  }
  dev::block_sum<2, kVecTpb>(acc, s_red);
  if (threadIdx.x == 0) {
    partials[static_cast<size_t>(blockIdx.x) * 2 + 0] = acc[0];
    partials[static_cast<size_t>(blockIdx.x) * 2 + 1] = acc[1];
  }
}

template <typename T>
__global__ void __launch_bounds__(kVecTpb) func_eval_kernel(int n, FnView<T> f, const T *v, double *partials) {
  __shared__ double s_red[kVecTpb / 64];
  const int i = blockIdx.x * kVecTpb + threadIdx.x;
  double acc[1] = {0.0};
  if (i < n) acc[0] = static_cast<double>(dev::FuncEval(f.h[i], f.a[i], f.b[i], f.c[i], f.d[i], f.e[i], v[i]));
  dev::block_sum<1, kVecTpb>(acc, s_red);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}

template <typename T>
__global__ void __launch_bounds__(kVecTpb) prox_eval_kernel(int n, FnView<T> f, T rho, const T *in, T *out) {
  const int i = blockIdx.x * kVecTpb + threadIdx.x;
  if (i < n) out[i] = dev::ProxEval(f.h[i], f.a[i], f.b[i], f.c[i], f.d[i], f.e[i], in[i], rho);
}

template <typename T>
__global__ void __launch_bounds__(kVecTpb) proj_subgrad_kernel(int n, FnView<T> f, const T *x, const T *v, T *out) {
  const int i = blockIdx.x * kVecTpb + threadIdx.x;
  if (i < n) out[i] = dev::ProjSubgradEval(f.h[i], f.a[i], f.b[i], f.c[i], f.d[i], f.e[i], v[i], x[i]);
}

template <typename T>
__global__ void __launch_bounds__(kVecTpb) unscale_kernel(UnscaleArgs<T> a, int blocks_x) {
  const bool is_x = static_cast<int>(blockIdx.x) < blocks_x;
  const int blk = is_x ? blockIdx.x : blockIdx.x - blocks_x;
  const int i = blk * kVecTpb + threadIdx.x;
  if (is_x) {
    if (i < a.n_x) {
      const T ei = a.e[i];
      a.x_out[i] = a.x12[i] * ei;                                               // pogs.cpp:518
      if (a.mu_out)
        a.mu_out[i] = -a.rho * (a.zt_scale * a.xt[i] - a.xprev[i] + a.x12[i]) / ei;  // :510-515
    }
  } else {
    if (i < a.n_y) {
      const T di = a.d[i];
      a.y_out[i] = a.y12[i] / di;                                               // :517
      a.l_out[i] = -a.rho * (a.zt_scale * a.yt[i] - a.yprev[i] + a.y12[i]) * di;     // :510-514
    }
  }
}

struct SumJobs {
  SumJob j[kMaxSumJobs];
};

// the whole workgroup strides over the partials of one scalar at a time (independent loads,
// four in flight per thread), then a fixed-order wave / workgroup reduction
__device__ __forceinline__ void run_sum_job(const SumJob &job, double (&s_w)[4]) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const size_t stride = job.stride > 0 ? job.stride : job.ns;
  for (int k = 0; k < job.ns; ++k) {
    const double *p = job.partials + job.offset + k;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int b = t;
    for (; b + 768 < job.nparts; b += 1024) {
      s0 += p[static_cast<size_t>(b) * stride];
      s1 += p[static_cast<size_t>(b + 256) * stride];
      s2 += p[static_cast<size_t>(b + 512) * stride];
      s3 += p[static_cast<size_t>(b + 768) * stride];
    }
    for (; b < job.nparts; b += 256) s0 += p[static_cast<size_t>(b) * stride];
    double s = dev::wave_sum((s0 + s1) + (s2 + s3));
    if (lane == 0) s_w[wave] = s;
    __syncthreads();
    if (t == 0) job.out[k] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) sum_jobs_kernel(SumJobs jobs) {
  __shared__ double s_w[4];
  run_sum_job(jobs.j[blockIdx.x], s_w);
}

// The iteration's closing launch: one workgroup per sum job, and the workgroup that finishes
// last (a device counter) copies the scalar block to the host-mapped mirror and raises the
// sequence word -- publish_scalars_kernel without a launch of its own.
__global__ void __launch_bounds__(256) sum_publish_kernel(SumJobs jobs, int njobs, double *S, int count,
                                                          double *host_S, unsigned long long *host_seq,
                                                          unsigned long long seq, unsigned *counter, ScalarOverlay ov) {
  __shared__ double s_w[4];
  __shared__ unsigned s_last;
  run_sum_job(jobs.j[blockIdx.x], s_w);
  if (threadIdx.x == 0) {
    __threadfence();   // this job's sums before the count
    s_last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == static_cast<unsigned>(njobs - 1);
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();     // the other workgroups' sums before the copy
  const int t = threadIdx.x;
  if (t == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (t < count) {
    double v = __hip_atomic_load(S + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int base = 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {   // scalars summed over the ranks in the pack buffer (see ScalarOverlay)
      if (t >= ov.slot[q] && t < ov.slot[q] + ov.n[q]) {
        v = ov.src[base + t - ov.slot[q]];
        S[t] = v;
      }
      base += ov.n[q];
    }
    host_S[t] = v;
  }
  __threadfence_system();
  __syncthreads();
  if (t == 0) __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <typename T>
__global__ void fill_kernel(T *p, T v, size_t n) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
template <typename T>
__global__ void sqrt_kernel(T *p, size_t n) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) p[i] = dev::Sqrt(p[i]);
}
template <typename T>
__global__ void scal_kernel(T *p, T alpha, size_t n) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) p[i] *= alpha;
}
template <typename T>
__global__ void axpby_kernel(size_t n, T a, const T *x, T b, T *y) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) y[i] = (b == static_cast<T>(0)) ? a * x[i] : a * x[i] + b * y[i];
}
template <typename T>
__global__ void scale_by_kernel(size_t n, T alpha, const T *in, const T *sc, bool divide, T *out) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) out[i] = divide ? alpha * in[i] / sc[i] : alpha * in[i] * sc[i];
}

inline dim3 grid1d(size_t n, int tpb = 256) { return dim3(static_cast<unsigned>((n + tpb - 1) / tpb)); }

}  // namespace

template <typename T>
void launch_scale_objective(FnView<T> fn, T *a, T *c, T *d, T *e, const T *scale, int n, bool divide,
                            hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(scale_objective_kernel<T>, dim3(vec_blocks(n)), dim3(kVecTpb), 0, s, fn, a, c, d, e,
                     scale, n, divide);
}

namespace {
template <typename T>
struct UniformProbeOut {
  int differs;   // bit k set: array k (h, c, d, e) holds more than one value
  int h0;
  T c0, d0, e0;
};
template <typename T>
__global__ void __launch_bounds__(256) uniform_probe_kernel(FnView<T> fn, int n, UniformProbeOut<T> *out) {
  const int h0 = fn.h[0];
  const T c0 = fn.c[0], d0 = fn.d[0], e0 = fn.e[0];
  int bad = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    bad |= (fn.h[i] != h0) ? 1 : 0;
    bad |= (fn.c[i] != c0) ? 2 : 0;   // (a NaN differs from itself: such an array is simply read as before)
    bad |= (fn.d[i] != d0) ? 4 : 0;
    bad |= (fn.e[i] != e0) ? 8 : 0;
  }
  if (bad) atomicOr(&out->differs, bad);
  if (blockIdx.x == 0 && threadIdx.x == 0) { out->h0 = h0; out->c0 = c0; out->d0 = d0; out->e0 = e0; }
}
}  // namespace
template <typename T>
FnUniform<T> probe_uniform(FnView<T> fn, int n, hipStream_t s) {
  FnUniform<T> u;
  if (n <= 0) return u;
  DevBuf<UniformProbeOut<T>> out(1);
  out.zero(s);
  hipLaunchKernelGGL(uniform_probe_kernel<T>, dim3(std::max(1, std::min((n + 255) / 256, 2048))), dim3(256), 0, s, fn, n, out.p);
  UniformProbeOut<T> h;
  POGS_HIP_CHECK(hipMemcpyAsync(&h, out.p, sizeof(h), hipMemcpyDeviceToHost, s));
  POGS_HIP_CHECK(hipStreamSynchronize(s));
  u.mask = ~h.differs & 15;
  u.h = h.h0; u.c = h.c0; u.d = h.d0; u.e = h.e0;
  return u;
}

template <typename T>
void launch_admm_pre(const AdmmPreArgs<T> &a, hipStream_t s) {
  const int blocks = a.blocks_x + pre_blocks(a.n_y);
  if (a.cheap) hipLaunchKernelGGL((admm_pre_kernel<T, true>), dim3(blocks), dim3(kVecTpb), 0, s, a);
  else hipLaunchKernelGGL((admm_pre_kernel<T, false>), dim3(blocks), dim3(kVecTpb), 0, s, a);
}

template <typename T>
void launch_admm_tail(int n, const T *znew, const T *zprev, const T *z12, T *ztemp, double *partials,
                      hipStream_t s) {
  hipLaunchKernelGGL(admm_tail_kernel<T>, dim3(vec_blocks(n)), dim3(kVecTpb), 0, s, n, znew, zprev, z12,
                     ztemp, partials);
}

template <typename T>
void launch_func_eval(int n, FnView<T> f, const T *v, double *partials, hipStream_t s) {
  hipLaunchKernelGGL(func_eval_kernel<T>, dim3(vec_blocks(n)), dim3(kVecTpb), 0, s, n, f, v, partials);
}

template <typename T>
void launch_prox_eval(int n, FnView<T> f, T rho, const T *in, T *out, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(prox_eval_kernel<T>, dim3(vec_blocks(n)), dim3(kVecTpb), 0, s, n, f, rho, in, out);
}

template <typename T>
void launch_proj_subgrad(int n, FnView<T> f, const T *x, const T *v, T *out, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(proj_subgrad_kernel<T>, dim3(vec_blocks(n)), dim3(kVecTpb), 0, s, n, f, x, v, out);
}

template <typename T>
void launch_unscale(const UnscaleArgs<T> &a, hipStream_t s) {
  const int bx = vec_blocks(a.n_x);
  hipLaunchKernelGGL(unscale_kernel<T>, dim3(bx + vec_blocks(a.n_y)), dim3(kVecTpb), 0, s, a, bx);
}

namespace {
// value of slot t: from the overlay if it covers t (then also stored into the device block)
__device__ __forceinline__ double overlay_value(double *S, int t, const ScalarOverlay &ov) {
  int base = 0;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (t >= ov.slot[q] && t < ov.slot[q] + ov.n[q]) {
      const double v = ov.src[base + t - ov.slot[q]];
      S[t] = v;
      return v;
    }
    base += ov.n[q];
  }
  return __hip_atomic_load(S + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void __launch_bounds__(64) publish_scalars_kernel(double *S, int count, double *host_S,
                                                           unsigned long long *host_seq, unsigned long long seq,
                                                           ScalarOverlay ov) {
  const int t = threadIdx.x;
  if (t < count) host_S[t] = overlay_value(S, t, ov);
  __threadfence_system();
  __syncthreads();
  if (t == 0) __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void __launch_bounds__(64) apply_overlay_kernel(double *S, ScalarOverlay ov) {
  const int t = threadIdx.x;
  if (t < kNumSlots) (void)overlay_value(S, t, ov);
}
}  // namespace

void launch_publish_scalars(double *S, int count, double *host_S, unsigned long long *host_seq,
                            unsigned long long seq, hipStream_t s, const ScalarOverlay &ov) {
  hipLaunchKernelGGL(publish_scalars_kernel, dim3(1), dim3(64), 0, s, S, count, host_S, host_seq, seq, ov);
}
void launch_apply_overlay(double *S, const ScalarOverlay &ov, hipStream_t s) {
  if (ov.n[0] + ov.n[1] > 0) hipLaunchKernelGGL(apply_overlay_kernel, dim3(1), dim3(64), 0, s, S, ov);
}

namespace {
__global__ void __launch_bounds__(256) max_partials_kernel(const double *partials, int n, double *out) {
  __shared__ double s_w[4];
  double v = 0.0;
  for (int b = threadIdx.x; b < n; b += 256) v = fmax(v, partials[b]);
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) *out = fmax(fmax(s_w[0], s_w[1]), fmax(s_w[2], s_w[3]));
}
}  // namespace

void preload_vec_code() {
  hipFuncAttributes fa;
  (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(max_partials_kernel));
}

void launch_max_partials(const double *partials, int n, double *out, hipStream_t s) {
  hipLaunchKernelGGL(max_partials_kernel, dim3(1), dim3(256), 0, s, partials, n, out);
}

namespace {
// (slot numbers of the CG block: cg_kernels.h CgSlot -- gamma 0, alpha 1, beta 2, delta 3, indefinite 4)
__global__ void __launch_bounds__(256) sum_cg_kernel(SumJob job, double *S, double *cg, int mode, double shift, double eps) {
  __shared__ double s_w[4];
  run_sum_job(job, s_w);   // ends with a barrier: thread 0 sees its own store
  if (threadIdx.x != 0) return;
  if (mode == 1) {
    double delta = S[kCgQ2] + shift * S[kCgP2];
    if (delta <= 0.0) cg[4] = 1.0;
    if (delta == 0.0) delta = eps;
    cg[3] = delta;
    cg[1] = cg[0] / delta;
  } else if (mode == 2) {
    const double g1 = cg[0], g = S[kCgS2];
    cg[0] = g;
    cg[2] = g / g1;
  } else if (mode == 3) {
    cg[0] = S[kCgS2];
    cg[4] = 0.0;
  }
}
}  // namespace

void launch_sum_cg(const SumJob &job, double *S, double *cg, int mode, double shift, double eps, hipStream_t s) {
  hipLaunchKernelGGL(sum_cg_kernel, dim3(1), dim3(256), 0, s, job, S, cg, mode, shift, eps);
}

void launch_sum_jobs(const SumJob *jobs, int njobs, hipStream_t s) {
  POGS_CHECK(njobs >= 1 && njobs <= kMaxSumJobs, "sum jobs");
  SumJobs j;
  for (int i = 0; i < kMaxSumJobs; ++i) j.j[i] = jobs[i < njobs ? i : 0];
  hipLaunchKernelGGL(sum_jobs_kernel, dim3(njobs), dim3(256), 0, s, j);
}

void launch_sum_publish(const SumJob *jobs, int njobs, double *S, int count, double *host_S,
                        unsigned long long *host_seq, unsigned long long seq, unsigned *counter, hipStream_t s,
                        const ScalarOverlay &ov) {
  POGS_CHECK(njobs >= 1 && njobs <= kMaxSumJobs && count <= 256, "sum jobs");
  SumJobs j;
  for (int i = 0; i < kMaxSumJobs; ++i) j.j[i] = jobs[i < njobs ? i : 0];
  hipLaunchKernelGGL(sum_publish_kernel, dim3(njobs), dim3(256), 0, s, j, njobs, S, count, host_S, host_seq, seq,
                     counter, ov);
}

namespace {
template <typename T>
__global__ void exact_u_vec_kernel(int m, const T *y12, const T *yt, const T *yprev, T c, T *u) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) u[i] = y12[i] + c * yt[i] - yprev[i];
}
}  // namespace
template <typename T>
void launch_exact_u(int m, const T *y12, const T *yt, const T *yprev, T c, T *u, hipStream_t s) {
  if (m) hipLaunchKernelGGL(exact_u_vec_kernel<T>, dim3((m + 255) / 256), dim3(256), 0, s, m, y12, yt, yprev, c, u);
}

template <typename T>
void launch_fill(T *p, T v, size_t n, hipStream_t s) {
  if (n) hipLaunchKernelGGL(fill_kernel<T>, grid1d(n), dim3(256), 0, s, p, v, n);
}
void launch_fill_int(int *p, int v, size_t n, hipStream_t s) {
  if (n) hipLaunchKernelGGL(fill_kernel<int>, grid1d(n), dim3(256), 0, s, p, v, n);
}
template <typename T>
void launch_sqrt_inplace(T *p, size_t n, hipStream_t s) {
  if (n) hipLaunchKernelGGL(sqrt_kernel<T>, grid1d(n), dim3(256), 0, s, p, n);
}
template <typename T>
void launch_scal(T *p, T alpha, size_t n, hipStream_t s) {
  if (n) hipLaunchKernelGGL(scal_kernel<T>, grid1d(n), dim3(256), 0, s, p, alpha, n);
}
template <typename T>
void launch_scale_by(size_t n, T alpha, const T *in, const T *sc, bool divide, T *out, hipStream_t s) {
  if (n) hipLaunchKernelGGL(scale_by_kernel<T>, grid1d(n), dim3(256), 0, s, n, alpha, in, sc, divide, out);
}
template <typename T>
void launch_axpby(size_t n, T a, const T *x, T b, T *y, hipStream_t s) {
  if (n) hipLaunchKernelGGL(axpby_kernel<T>, grid1d(n), dim3(256), 0, s, n, a, x, b, y);
}
// ---------------------------------------------------------------------------------------------
// Read-bandwidth probe (PogsAmdReadBandwidth, include/pogs_amd.h part 3): how fast THIS device reads a
// large array with nothing else to do -- the measured ceiling bench.py prints next to the data-sheet peak
// ("measured stream on the box and the datasheet value -- state both").  Two
// patterns, the best of which is returned: all workgroups marching through the array side by side with
// 16-byte non-temporal loads (the best of scripts/micro/read_bw.hip's table, profiles/r03_read_bw.txt),
// and the row-block shape of the one-pass iteration kernel (a workgroup reads whole 40 KB rows, two a step).
namespace {
typedef float bw_v4 __attribute__((ext_vector_type(4)));
template <int U>
__global__ void __launch_bounds__(256) bw_stride_kernel(const bw_v4 *__restrict__ a, size_t nvec, float *out) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  bw_v4 acc = {0, 0, 0, 0};
  for (; i + (U - 1) * stride < nvec; i += U * stride) {
    bw_v4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(a + i + u * stride);
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  for (; i < nvec; i += stride) acc += __builtin_nontemporal_load(a + i);
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;   // never true for the zero-filled array: keeps the loads
}
template <int NV, int R>
__global__ void __launch_bounds__(256) bw_rows_kernel(const bw_v4 *__restrict__ a, int rows, int rowvec, float *out) {
  bw_v4 acc = {0, 0, 0, 0};
  for (int r0 = blockIdx.x * R; r0 < rows; r0 += gridDim.x * R) {
    bw_v4 v[R][NV];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int c = k * 256 + static_cast<int>(threadIdx.x);
        v[r][k] = __builtin_nontemporal_load(a + static_cast<size_t>(min(r0 + r, rows - 1)) * rowvec + min(c, rowvec - 1));
      }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int k = 0; k < NV; ++k) acc += v[r][k];
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}
}  // namespace

double measure_read_bandwidth_gbs(size_t bytes, int reps, int *pattern) {
  const int rowvec = 2500, rows = static_cast<int>(std::max<size_t>(1, bytes / (16 * static_cast<size_t>(rowvec))));
  const size_t nvec = static_cast<size_t>(rows) * rowvec;
  DevBuf<bw_v4> a(nvec);
  DevBuf<float> out(16);
  hipStream_t s = nullptr;
  a.zero(s);
  hipEvent_t e0, e1;
  POGS_HIP_CHECK(hipEventCreate(&e0));
  POGS_HIP_CHECK(hipEventCreate(&e1));
  double best = 0;
  reps = std::max(1, reps);
  for (int pat = 0; pat < 2; ++pat) {
    auto launch = [&]() {
      if (pat == 0) hipLaunchKernelGGL((bw_stride_kernel<4>), dim3(512), dim3(256), 0, s, a.p, nvec, out.p);
      else hipLaunchKernelGGL((bw_rows_kernel<10, 2>), dim3(512), dim3(256), 0, s, a.p, rows, rowvec, out.p);
    };
    for (int i = 0; i < 2; ++i) launch();
    POGS_HIP_CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) launch();
    POGS_HIP_CHECK(hipEventRecord(e1, s));
    POGS_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0;
    POGS_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double gbs = static_cast<double>(nvec) * 16.0 * reps / (static_cast<double>(ms) * 1e-3) / 1e9;
    if (gbs > best) { best = gbs; if (pattern) *pattern = pat; }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  POGS_HIP_CHECK(hipDeviceSynchronize());   // a / out go back to the pool with nothing in flight
  return best;
}

// Diagnostic (PogsAmdWaveSumCheck): dev::wave_sum -- the wavefront total formed in the vector ALU -- next to the same
// butterfly through __shfl_xor, per wavefront of 64 consecutive inputs; every lane's total is written.
namespace {
template <typename T>
__global__ void __launch_bounds__(256) wave_sum_check_kernel(const T *in, size_t n, T *out_alu, T *out_lds) {
  const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;   // (n is a multiple of 64: whole wavefronts leave together)
  const T v = in[i];
  out_alu[i] = dev::wave_sum(v);
  out_lds[i] = dev::wave_sum_shfl(v);
}
}  // namespace
template <typename T>
void wave_sum_check(const T *in_host, size_t n, T *alu_host, T *lds_host) {
  POGS_CHECK(n > 0 && n % 64 == 0, "whole wavefronts of 64 values");
  DevBuf<T> in(n), a(n), b(n);
  hipStream_t s = nullptr;
  POGS_HIP_CHECK(hipMemcpyAsync(in.p, in_host, n * sizeof(T), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL((wave_sum_check_kernel<T>), dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, in.p, n, a.p, b.p);
  POGS_HIP_CHECK(hipGetLastError());
  POGS_HIP_CHECK(hipMemcpyAsync(alu_host, a.p, n * sizeof(T), hipMemcpyDeviceToHost, s));
  POGS_HIP_CHECK(hipMemcpyAsync(lds_host, b.p, n * sizeof(T), hipMemcpyDeviceToHost, s));
  POGS_HIP_CHECK(hipDeviceSynchronize());
}
template void wave_sum_check<float>(const float *, size_t, float *, float *);
template void wave_sum_check<double>(const double *, size_t, double *, double *);

#define POGS_INST(T)                                                                                     \
  template void launch_scale_objective<T>(FnView<T>, T *, T *, T *, T *, const T *, int, bool, hipStream_t); \
  template void launch_admm_pre<T>(const AdmmPreArgs<T> &, hipStream_t);                                 \
  template FnUniform<T> probe_uniform<T>(FnView<T>, int, hipStream_t);                                   \
  template void launch_admm_tail<T>(int, const T *, const T *, const T *, T *, double *, hipStream_t);   \
  template void launch_func_eval<T>(int, FnView<T>, const T *, double *, hipStream_t);                   \
  template void launch_prox_eval<T>(int, FnView<T>, T, const T *, T *, hipStream_t);                     \
  template void launch_proj_subgrad<T>(int, FnView<T>, const T *, const T *, T *, hipStream_t);          \
  template void launch_unscale<T>(const UnscaleArgs<T> &, hipStream_t);                                  \
  template void launch_exact_u<T>(int, const T *, const T *, const T *, T, T *, hipStream_t);            \
  template void launch_fill<T>(T *, T, size_t, hipStream_t);                                             \
  template void launch_sqrt_inplace<T>(T *, size_t, hipStream_t);                                        \
  template void launch_scal<T>(T *, T, size_t, hipStream_t);                                             \
  template void launch_scale_by<T>(size_t, T, const T *, const T *, bool, T *, hipStream_t);             \
  template void launch_axpby<T>(size_t, T, const T *, T, T *, hipStream_t);
POGS_INST(float)
POGS_INST(double)
#undef POGS_INST

}  // namespace pogs_amd
