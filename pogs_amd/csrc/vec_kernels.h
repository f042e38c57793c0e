// Element-wise ADMM kernels with wavefront-reduced scalar outputs.
// Reference: the vector algebra inside PogsImplementation::Solve
// (src/cpu/pogs.cpp:254-278, 342-348, 397-399, 473, 510-518), which the CPU
// path runs as ~25 separate BLAS-1 calls per iteration.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"
#include "prox.h"

namespace pogs_amd {

// Device scalar block layout (doubles).  The first group holds sums over rows
// (y-sized data): on a row-sharded solve the ones in use are all-reduced (contiguous
// ranges).  The second group holds sums over columns (replicated data).
enum Slot : int {
  kGapY = 0, kWY2, kHY2, kDYprev2, kDY12, kExactR2, kPowSx2, kFro2, kFvalF, kCgQ2,
  kAmax = 10,   // max |entry| of the scaled matrix before the Frobenius normalisation (local shard)
  kSkMark = 11, // last Sinkhorn-Knopp pass in which an entry left the common growth factor
  kGapX = 12, kWX2, kHX2, kDXprev2, kDX12, kExactS2, kPowX2, kPowXGx, kFvalG, kCgP2, kCgS2, kCgX2,
  kSpecGapY = 26, kSpecWY2, kSpecHY2,   // next iteration's y-half sums from the one-pass kernel
  kSpecGapX = 29, kSpecWX2, kSpecHX2,   // ... x-half sums (m <= n, transposed storage)
  kSkRatio = 32,   // sum over the entries of (new / old) of a Sinkhorn-Knopp pass
  // device-resident CGLS loop (cg_fused.h): the scalars of cgls.h:236-306 and its loop control
  kFcDone = 40,    // != 0: the projection's CG loop has ended (converged, or maxit steps)
  kFcSteps,        // CG steps taken by the current projection (adjacent to kFcDone: reset together)
  kFcNorms0,       // |s_0|^2
  kFcGamma0, kFcGamma1,   // gamma, ping-pong by step parity (read and written by different launches only)
  kFcAlpha, kFcBeta, kFcDelta, kFcIndef,
  kNumSlots = 52
};

template <typename T>
struct FnBuf {  // owning device SoA of function objects
  DevBuf<int> h;
  DevBuf<T> a, b, c, d, e;
  void alloc(size_t n) { h.alloc(n); a.alloc(n); b.alloc(n); c.alloc(n); d.alloc(n); e.alloc(n); }
  FnView<T> view() const { return FnView<T>{h.p, a.p, b.p, c.p, d.p, e.p}; }
};

// Which coefficient arrays of a (scaled) function vector hold one value throughout (bit 0: h, 1: c, 2: d, 3: e), and
// that value.  A lasso's f is (kSquare, a_i, b_i, 1, 0, 0) and its g (kAbs, a_j, 0, lambda, 0, 0): four of the six
// streams the prox step reads per element carry no information -- 16 of its 44 bytes per element at C4.
template <typename T>
struct FnUniform {
  int mask = 0;
  int h = 0;
  T c = 0, d = 0, e = 0;
};
// Probes the n function objects of `fn` (device arrays, scaled); waits for the stream.
template <typename T>
FnUniform<T> probe_uniform(FnView<T> fn, int n, hipStream_t s);

template <typename T>
struct AdmmPreArgs {
  int n_x, n_y;
  FnView<T> g, f;
  const T *x_cur, *y_cur;
  const T *xt, *yt;
  T zt_scale;
  T *x12, *y12;
  T *xtemp, *ytemp;
  T rho, alpha;
  double *partials;  // [blocks_x + blocks_y][3]; blocks = pre_blocks(n)
  int blocks_x;
  // optional (CGLS warm start, cg_fused.h): x_aux = x_cur - xtemp_new, y_aux = ytemp_new - y_cur;
  // cg_reset[0..1] = 0 (the CG loop's done flag and step count)
  T *x_aux = nullptr, *y_aux = nullptr;
  double *cg_reset = nullptr;
  // every f_i and g_j is one of the few-operation base functions (is_cheap_prox): the prox is inlined
  // instead of a call into the full library (Lambert W, cubic roots, Newton steps behind one switch)
  bool cheap = false;
  FnUniform<T> ug, uf;   // uniform coefficient arrays of g and f: not loaded (mask 0: everything is)
};

// clamp c,e >= 0 (FunctionObj::CheckConsts, prox_lib.h:62-69) and scale by the
// equilibration (PogsObjectiveSeparable::scale, pogs.cpp:608-617):
//   divide: a/=s, d/=s, e/=s^2 (f with s=d);  multiply: a*=s, d*=s, e*=s^2 (g with s=e).
template <typename T>
void launch_scale_objective(FnView<T> fn, T *a, T *c, T *d, T *e, const T *scale, int n, bool divide,
                            hipStream_t s);

constexpr int kVecTpb = 256;
inline int vec_blocks(int n) { return (n + kVecTpb - 1) / kVecTpb; }
// admm_pre_kernel / cgf_close_kernel: kPreU chunks of kVecTpb elements per workgroup -- a quarter of the
// workgroup reductions and of the partial sums the closing launch adds up (1.1e6 elements at C4)
constexpr int kPreU = 4;
inline int pre_blocks(int n) { return (n + kVecTpb * kPreU - 1) / (kVecTpb * kPreU); }

// pre-projection step: prox, gap / norm partials, over-relaxation.
// Writes partials [blocks][3] = {sum w*h, sum w^2, sum h^2}; x blocks first.
template <typename T>
void launch_admm_pre(const AdmmPreArgs<T> &a, hipStream_t s);

// element-wise projection tail (CGLS path): see ProjTailOp.  partials [blocks][2].
template <typename T>
void launch_admm_tail(int n, const T *znew, const T *zprev, const T *z12, T *ztemp, double *partials,
                      hipStream_t s);

// partials[b] = sum over the block of FuncEval(f_i, v_i).
template <typename T>
void launch_func_eval(int n, FnView<T> f, const T *v, double *partials, hipStream_t s);

// out[i] = ProxEval(f_i, in[i], rho)
template <typename T>
void launch_prox_eval(int n, FnView<T> f, T rho, const T *in, T *out, hipStream_t s);

// out[i] = ProjSubgradEval(f_i, v[i], x[i])   (prox_lib.h:468-493, 538-546)
template <typename T>
void launch_proj_subgrad(int n, FnView<T> f, const T *x, const T *v, T *out, hipStream_t s);

// Un-scaling of the outputs (pogs.cpp:510-518).
template <typename T>
struct UnscaleArgs {
  int n_x, n_y;
  const T *x12, *y12, *xt, *yt, *xprev, *yprev, *d, *e;
  T zt_scale, rho;
  T *x_out, *y_out, *l_out, *mu_out;
};
template <typename T>
void launch_unscale(const UnscaleArgs<T> &a, hipStream_t s);

// Up to 4 independent partial-sum jobs in one launch: out[k] = sum_b partials[b*ns+k].
struct SumJob {
  const double *partials;
  int nparts, ns;
  double *out;
  int stride = 0;   // doubles between consecutive partial records (0: ns)
  int offset = 0;   // first scalar of the record to sum
};
// Scalars that were summed over the ranks inside a packed buffer (dense_solver.h: pack buffer tail):
// the publishing launch takes slots [slot[q], slot[q] + n[q]) from src instead of the device
// block (and stores them there), q = 0, 1; src is read in that order.  n = {0, 0}: none.
struct ScalarOverlay {
  const double *src = nullptr;
  int slot[2] = {0, 0};
  int n[2] = {0, 0};
};
// Makes the runtime load this translation unit's code object (see preload_gemm_code).
void preload_vec_code();
constexpr int kMaxSumJobs = 8;
// launch_sum_jobs and launch_publish_scalars in one launch (see sum_publish_kernel); *counter is a
// zero-initialised device word the kernel leaves at zero.
void launch_sum_publish(const SumJob *jobs, int njobs, double *S, int count, double *host_S,
                        unsigned long long *host_seq, unsigned long long seq, unsigned *counter, hipStream_t s,
                        const ScalarOverlay &ov = ScalarOverlay());
// Copies the device scalar block to a host-mapped mirror and then raises *host_seq to `seq`
// (system-scope release): the host reads the block after polling the sequence word, with no
// copy-engine round trip and no stream-synchronize call.
void launch_publish_scalars(double *S, int count, double *host_S, unsigned long long *host_seq,
                            unsigned long long seq, hipStream_t s, const ScalarOverlay &ov = ScalarOverlay());
// S[slot ...] = src[...] for the overlay's ranges (the non-polling fetch path and rare call sites)
void launch_apply_overlay(double *S, const ScalarOverlay &ov, hipStream_t s);

// u = y12 + c yt - yprev   (pogs.cpp:366-368, y half)
template <typename T>
void launch_exact_u(int m, const T *y12, const T *yt, const T *yprev, T c, T *u, hipStream_t s);

// *out = max_b partials[b]
void launch_max_partials(const double *partials, int n, double *out, hipStream_t s);

void launch_sum_jobs(const SumJob *jobs, int njobs, hipStream_t s);

// One sum job and, in the same launch, the CGLS scalar that needs it (cg_kernels.h: the three
// one-thread kernels): mode 1 alpha = gamma / (S[kCgQ2] + shift S[kCgP2]) (cgls.h:262-271),
// mode 2 beta = S[kCgS2] / gamma, gamma = S[kCgS2] (:288-292), mode 3 gamma = S[kCgS2] (:245).
void launch_sum_cg(const SumJob &job, double *S, double *cg, int mode, double shift, double eps, hipStream_t s);

// Read-bandwidth probe of the current device (diagnostic; see vec_kernels.hip): GB/s of the better of two
// read patterns over `bytes` of zero-filled memory, `reps` timed launches each; *pattern: 0 = side-by-side
// grid stride, 1 = row blocks.
double measure_read_bandwidth_gbs(size_t bytes, int reps, int *pattern);

// Diagnostic: the wavefront sums of n (a multiple of 64) host values, by dev::wave_sum and by the __shfl_xor butterfly.
template <typename T> void wave_sum_check(const T *in_host, size_t n, T *alu_host, T *lds_host);

// Misc vector helpers.
template <typename T> void launch_fill(T *p, T v, size_t n, hipStream_t s);
void launch_fill_int(int *p, int v, size_t n, hipStream_t s);
template <typename T> void launch_sqrt_inplace(T *p, size_t n, hipStream_t s);
template <typename T> void launch_scal(T *p, T alpha, size_t n, hipStream_t s);
// out[i] = alpha * in[i] * (divide ? 1 / sc[i] : sc[i])   (warm start: x0 / e, lambda0 / d)
template <typename T> void launch_scale_by(size_t n, T alpha, const T *in, const T *sc, bool divide, T *out, hipStream_t s);
// y = a*x + b*y
template <typename T> void launch_axpby(size_t n, T a, const T *x, T b, T *y, hipStream_t s);

}  // namespace pogs_amd
