/* include/pogs_amd.h -- C ABI of the MI355X-native POGS graph-form ADMM engine.
 *
 *   minimize  sum_i f_i(y_i) + sum_j g_j(x_j)   subject to  y = A x
 *   f_i(v) = c_i h_i(a_i v - b_i) + d_i v + e_i v^2 / 2      (same for g_j)
 *
 * Part 1 is the drop-in boundary: the four graph-form entry points of the
 * reference's C interface, with identical names, argument order, argument
 * meaning, enum values and return codes, so that the reference's own
 * python/pogs/graph.py (ctypes) or a C caller can be pointed at libpogs_amd.so
 * instead of libpogs_cpu.so.  Every pointer in part 1 is a HOST pointer owned
 * by the caller; the library copies what it needs and writes exactly n, m, m
 * elements to x, y, l (reference: src/interface_c/pogs_c.cpp:19-52).
 *
 * Part 2 is an additive extension for what the one-shot ABI cannot express on
 * a GPU: a persistent handle (equilibration + factorisation reused across
 * solves; the reference offers this only through its C++ API,
 * src/cpu/pogs.cpp:113-114), device-resident inputs, row-sharded multi-GPU
 * solves over RCCL, iteration stepping for benchmarks, and statistics.
 *
 * All functions are extern "C", take plain pointers and sizes, and never throw.
 */
#ifndef POGS_AMD_H_
#define POGS_AMD_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------
 * Part 1 -- drop-in graph-form ABI
 * ------------------------------------------------------------------------- */

/* replaces: src/interface_c/pogs_c.h:51 */
enum ORD { COL_MAJ, ROW_MAJ };

/* replaces: src/interface_c/pogs_c.h:54-69 (order pinned by
 * tests/test_c_interface.cpp:149-154: ABS == 0, SQUARE == 14, ZERO == 15) */
enum FUNCTION { ABS, EXP, HUBER, IDENTITY, INDBOX01, INDEQ0, INDGE0, INDLE0,
                LOGISTIC, MAXNEG0, MAXPOS0, NEGENTR, NEGLOG, RECIPR, SQUARE, ZERO };

/* Return codes: the reference's PogsStatus (src/include/pogs.h:31-37).
 * NB 3 (not 1) means "max_iter reached"; *final_iter is the 0-based index of
 * the last iteration executed (src/cpu/pogs.cpp:391-393). */
enum POGS_STATUS { POGS_SUCCESS = 0, POGS_INFEASIBLE = 1, POGS_UNBOUNDED = 2,
                   POGS_MAX_ITER = 3, POGS_NAN_FOUND = 4, POGS_INVALID_CONE = 5,
                   POGS_ERROR = 6 };

/* replaces: src/interface_c/pogs_c.h:75-82 (PogsD).  Dense A, direct projector
 * (src/interface_c/pogs_c.cpp:19-20). */
int PogsD(enum ORD ord, size_t m, size_t n, const double *A,
          const double *f_a, const double *f_b, const double *f_c,
          const double *f_d, const double *f_e, const enum FUNCTION *f_h,
          const double *g_a, const double *g_b, const double *g_c,
          const double *g_d, const double *g_e, const enum FUNCTION *g_h,
          double rho, double abs_tol, double rel_tol, unsigned int max_iter,
          unsigned int verbose, int adaptive_rho, int gap_stop,
          double *x, double *y, double *l, double *optval, unsigned int *final_iter);

/* replaces: src/interface_c/pogs_c.h:84-91 (PogsS) */
int PogsS(enum ORD ord, size_t m, size_t n, const float *A,
          const float *f_a, const float *f_b, const float *f_c,
          const float *f_d, const float *f_e, const enum FUNCTION *f_h,
          const float *g_a, const float *g_b, const float *g_c,
          const float *g_d, const float *g_e, const enum FUNCTION *g_h,
          float rho, float abs_tol, float rel_tol, unsigned int max_iter,
          unsigned int verbose, int adaptive_rho, int gap_stop,
          float *x, float *y, float *l, float *optval, unsigned int *final_iter);

/* replaces: src/interface_c/pogs_c.h:99-108 (PogsSparseD).  ROW_MAJ = CSR with
 * ptr of length m+1, COL_MAJ = CSC with ptr of length n+1; int32 indices; CGLS
 * projector (src/interface_c/pogs_c.cpp:69-73). */
int PogsSparseD(enum ORD ord, size_t m, size_t n, size_t nnz,
                const double *data, const int *ptr, const int *ind,
                const double *f_a, const double *f_b, const double *f_c,
                const double *f_d, const double *f_e, const enum FUNCTION *f_h,
                const double *g_a, const double *g_b, const double *g_c,
                const double *g_d, const double *g_e, const enum FUNCTION *g_h,
                double rho, double abs_tol, double rel_tol, unsigned int max_iter,
                unsigned int verbose, int adaptive_rho, int gap_stop,
                double *x, double *y, double *l, double *optval,
                unsigned int *final_iter);

/* replaces: src/interface_c/pogs_c.h:110-119 (PogsSparseS) */
int PogsSparseS(enum ORD ord, size_t m, size_t n, size_t nnz,
                const float *data, const int *ptr, const int *ind,
                const float *f_a, const float *f_b, const float *f_c,
                const float *f_d, const float *f_e, const enum FUNCTION *f_h,
                const float *g_a, const float *g_b, const float *g_c,
                const float *g_d, const float *g_e, const enum FUNCTION *g_h,
                float rho, float abs_tol, float rel_tol, unsigned int max_iter,
                unsigned int verbose, int adaptive_rho, int gap_stop,
                float *x, float *y, float *l, float *optval,
                unsigned int *final_iter);

/* ---------------------------------------------------------------------------
 * Part 2 -- MI355X extension: persistent handle, device inputs, multi-GPU
 * ------------------------------------------------------------------------- */

typedef struct PogsAmdSolver PogsAmdSolver; /* opaque */

enum POGS_AMD_DTYPE { POGS_AMD_F32 = 0, POGS_AMD_F64 = 1 };
enum POGS_AMD_MEM { POGS_AMD_HOST = 0, POGS_AMD_DEVICE = 1 };
enum POGS_AMD_PROJECTOR { POGS_AMD_PROJ_DEFAULT = 0, /* dense: direct, sparse: CGLS */
                          POGS_AMD_PROJ_DIRECT = 1, POGS_AMD_PROJ_CGLS = 2 };

#define POGS_AMD_UNIQUE_ID_BYTES 128

/* Row-sharding descriptor.  world == 1 (or a NULL pointer) means single GPU.
 * Every rank holds m_local consecutive rows of A and the matching slices of
 * f, y, l; x-sized data (g, x) is replicated.  The only collectives are sum
 * all-reduces of n-vectors / n*n Gram / a few scalars over RCCL. */
typedef struct PogsAmdDist {
  int rank;
  int world;
  size_t m_global;                              /* total rows over all ranks  */
  char unique_id[POGS_AMD_UNIQUE_ID_BYTES];     /* from PogsAmdDistUniqueId() */
} PogsAmdDist;

typedef struct PogsAmdOptions {
  int device;          /* HIP device ordinal; -1 = current device                   */
  int projector;       /* enum POGS_AMD_PROJECTOR                                   */
  int profile;         /* 1: bracket the dominant kernels with HIP events; k > 1:
                          every k-th such launch only (an event record costs the
                          stream a few microseconds of idle time)                   */
  int reserved[5];
} PogsAmdOptions;

typedef struct PogsAmdStats {
  /* last solve / iterate call */
  double t_total_s, t_init_s, t_loop_s, t_h2d_s;
  unsigned int iterations;        /* executed (= final_iter + 1)                    */
  unsigned int exact_iters;       /* iterations that evaluated exact residuals      */
  unsigned int norm_est_iters;    /* power iterations used by the norm estimate     */
  unsigned int rho_updates;
  unsigned long long cg_iters;    /* total CGLS inner iterations                    */
  unsigned long long matvecs;     /* passes over A (dense) / SpMVs (sparse), loop   */
  unsigned long long matvecs_init;/* the same for the one-time setup                */
  double rho_final, nrmA;
  /* HIP-event timing of the dominant kernel (options.profile = 1), loop only */
  double stream_ms;               /* sum of durations of the A-streaming launches   */
  unsigned long long stream_launches;
  double stream_bytes;            /* algorithmic bytes those launches had to move   */
  /* one-time setup pieces, HIP-event timed */
  double equil_ms, normest_ms, gram_ms, chol_ms, trtri_ms;
  double gram_flops;
  double reserved[8];             /* [0] / [1]: one-pass iteration, rho predictions hit / missed;
                                     [2]: all-reduce calls issued by the handle so far;
                                     [3]: ranks of the handle's communicator as RCCL reports
                                          them (ncclCommCount; 0 without row shards)             */
} PogsAmdStats;

/* Fill `out` (POGS_AMD_UNIQUE_ID_BYTES) with a fresh RCCL unique id (rank 0
 * calls this and ships the bytes to the other ranks by any means). */
int PogsAmdDistUniqueId(char *out);

/* Create a solver for a dense m x n matrix.  `A` is a host or device pointer
 * (mem), row- or column-major (ord), of type dtype.  The matrix is copied,
 * equilibrated (reference: src/cpu/matrix/matrix_dense.cpp:116-200), its norm
 * estimated (src/cpu/include/equil_helper.h:107-135) and the projector set up
 * (src/cpu/projector/projector_direct_dense.cpp:45-84,116-121).  With dist,
 * m is the LOCAL row count. */
int PogsAmdCreateDense(PogsAmdSolver **out, int dtype, enum ORD ord, size_t m,
                       size_t n, const void *A, int mem,
                       const PogsAmdOptions *opt, const PogsAmdDist *dist);

/* Create a solver for a sparse matrix (CSR if ord == ROW_MAJ, else CSC);
 * data/ptr/ind are host or device pointers (mem).  With dist (may be NULL) the
 * matrix is this rank's block of m consecutive rows, given as CSR (SURVEY.md
 * section 8 f.3): CGLS with the A^T products and row sums all-reduced. */
int PogsAmdCreateSparse(PogsAmdSolver **out, int dtype, enum ORD ord, size_t m,
                        size_t n, size_t nnz, const void *data, const int *ptr,
                        const int *ind, int mem, const PogsAmdOptions *opt,
                        const PogsAmdDist *dist);

/* Cold-start solve (reference: PogsImplementation::Solve, src/cpu/pogs.cpp:91-581).
 * Coefficient and output pointers are HOST pointers of the solver's dtype
 * (f_*: m_local, g_*: n; x: n, y/l: m_local).  mu (length n) may be NULL. */
int PogsAmdSolve(PogsAmdSolver *s,
                 const void *f_a, const void *f_b, const void *f_c, const void *f_d,
                 const void *f_e, const int *f_h,
                 const void *g_a, const void *g_b, const void *g_c, const void *g_d,
                 const void *g_e, const int *g_h,
                 double rho, double abs_tol, double rel_tol, unsigned int max_iter,
                 unsigned int verbose, int adaptive_rho, int gap_stop,
                 void *x, void *y, void *l, void *mu, double *optval,
                 unsigned int *final_iter);

/* The same solve with BROADCAST coefficients: a field of f or g whose pointer is NULL holds one value (a0 .. e0, h0) for
 * every element and is filled on the device -- the caller neither builds nor converts nor uploads an array for it.  A
 * lasso's f is (h = SQUARE, a = 1, b = b_i, c = 1, d = 0, e = 0) and its g (h = ABS, a = 1, b = 0, c = lambda, d = e = 0):
 * one per-element array out of twelve (python/pogs/graph.py:428,431 builds m + n objects for them; at 2.5e6 elements
 * the twelve arrays are 60 MB of host work per solve).  Everything else as PogsAmdSolve / PogsAmdBeginRun. */
typedef struct PogsAmdFn {
  const void *a, *b, *c, *d, *e;   /* HOST arrays of the solver's dtype, or NULL */
  const int *h;                    /* HOST array of enum FUNCTION values, or NULL */
  double a0, b0, c0, d0, e0;       /* the value of a field whose pointer is NULL  */
  int h0;
} PogsAmdFn;
int PogsAmdSolveFn(PogsAmdSolver *s, const PogsAmdFn *f, const PogsAmdFn *g,
                   double rho, double abs_tol, double rel_tol, unsigned int max_iter,
                   unsigned int verbose, int adaptive_rho, int gap_stop,
                   void *x, void *y, void *l, void *mu, double *optval,
                   unsigned int *final_iter);
int PogsAmdBeginRunFn(PogsAmdSolver *s, const PogsAmdFn *f, const PogsAmdFn *g,
                      double rho, double abs_tol, double rel_tol, unsigned int max_iter,
                      int adaptive_rho, int gap_stop);

/* Benchmark stepping.  PogsAmdBeginRun loads f/g and the solve parameters and
 * resets the ADMM state to the cold start; PogsAmdIterate then advances exactly
 * `iters` ADMM iterations of real solves (restarting from the cold start each
 * time a solve converges or hits max_iter), and returns the elapsed seconds
 * measured with HIP events on the solver's stream. */
int PogsAmdBeginRun(PogsAmdSolver *s,
                    const void *f_a, const void *f_b, const void *f_c, const void *f_d,
                    const void *f_e, const int *f_h,
                    const void *g_a, const void *g_b, const void *g_c, const void *g_d,
                    const void *g_e, const int *g_h,
                    double rho, double abs_tol, double rel_tol, unsigned int max_iter,
                    int adaptive_rho, int gap_stop);
int PogsAmdIterate(PogsAmdSolver *s, unsigned int iters, double *seconds,
                   unsigned int *solves_completed);

/* Warm start for the NEXT PogsAmdSolve / PogsAmdBeginRun call only (reference: C++-only
 * SetInitX / SetInitLambda, src/include/pogs.h:112-119, consumed at src/cpu/pogs.cpp:144-180;
 * as there, x0 and l0 must be given together).  HOST pointers: x0 (n), l0 (m_local). */
int PogsAmdSetWarmStart(PogsAmdSolver *s, const void *x0, const void *l0);

int PogsAmdGetStats(const PogsAmdSolver *s, PogsAmdStats *out);
int PogsAmdResetStats(PogsAmdSolver *s);
void PogsAmdDestroy(PogsAmdSolver *s);

/* Last error message of the calling thread ("" if none). */
const char *PogsAmdLastError(void);

/* Device memory pool.  The reference builds and destroys its solver inside every one-shot call
 * (src/interface_c/pogs_c.cpp:19-20, 67-68), i.e. its working set is allocated and freed per
 * call; on the GPU that costs map / first-touch / unmap stalls of 0.1-0.3 s at 5 GB.  The
 * library therefore keeps the device blocks of destroyed handles (and of finished PogsD/PogsS
 * calls) in a per-device cache and hands them to the next handle.  At most POGS_AMD_POOL_MB
 * (environment; default a quarter of the device's memory, 0 = no caching) stay idle per device;
 * an allocation that fails is retried after the cache has been emptied. */
typedef struct PogsAmdPoolInfo {
  unsigned long long mallocs;   /* blocks taken from the HIP runtime                    */
  unsigned long long reuses;    /* blocks taken from the cache                          */
  unsigned long long frees;     /* blocks given back to the HIP runtime                 */
  double malloc_ms, free_ms;    /* host time spent inside hipMalloc / hipFree           */
  size_t cached_bytes;          /* idle in the cache now                                */
  size_t live_bytes;            /* in use by handles now                                */
  size_t peak_cached_bytes;
} PogsAmdPoolInfo;
int PogsAmdPoolStats(int device, PogsAmdPoolInfo *out);
/* Gives the idle blocks of `device` (-1: every device) back to the HIP runtime. */
int PogsAmdPoolTrim(int device, size_t *freed_bytes);

/* ---------------------------------------------------------------------------
 * Part 3 -- building blocks exported for parity tests (device pointers unless
 * noted).  Not needed by a drop-in caller.
 * ------------------------------------------------------------------------- */

/* out[i] = Prox{f_i}(in[i]) with penalty rho; SoA coefficients; all HOST
 * pointers of type dtype (reference: ProxEval, src/include/prox_lib.h:207-230,
 * 503-511).  The evaluation runs on the GPU. */
int PogsAmdProxEval(int dtype, size_t n, const int *h, const void *a, const void *b,
                    const void *c, const void *d, const void *e, double rho,
                    const void *in, void *out);
/* sum_i f_i(in[i])  (reference: FuncEval, src/include/prox_lib.h:326-349,520-529) */
int PogsAmdFuncEval(int dtype, size_t n, const int *h, const void *a, const void *b,
                    const void *c, const void *d, const void *e, const void *in,
                    double *out);
/* v_out[i] = ProjSubgrad{f_i}(v_in[i]) at x_in[i]: the point of the subdifferential of f_i at
 * x_in[i] closest to v_in[i] (reference: ProjSubgradEval, src/include/prox_lib.h:468-493,
 * 538-546; unused by the reference's solvers, part of its prox library).  HOST pointers. */
int PogsAmdProjSubgradEval(int dtype, size_t n, const int *h, const void *a, const void *b,
                           const void *c, const void *d, const void *e, const void *x_in,
                           const void *v_in, void *v_out);
/* Equilibrated matrix, scalings and norm estimate of a solver (HOST outputs,
 * any may be NULL): A_eq (m*n row-major), d (m), e (n). */
int PogsAmdGetEquil(const PogsAmdSolver *s, void *A_eq, void *d, void *e, double *nrmA);
/* Projection onto {y = A_eq x}: (x, y) = argmin |x-x0|^2 + |y-y0|^2 (HOST ptrs). */
int PogsAmdProject(PogsAmdSolver *s, const void *x0, const void *y0, double tol,
                   void *x, void *y);
/* y = alpha * op(A_eq) x + beta * y on the solver's operator (HOST ptrs);
 * trans = 'n' or 't' (reference: Matrix::Mul). */
int PogsAmdMul(PogsAmdSolver *s, char trans, double alpha, const void *x, double beta,
               void *y);
/* Diagnostic: GB/s at which `device` (-1: current) reads `bytes` of device memory with nothing else to
 * do -- the better of two read-only kernels (all workgroups side by side, 16-byte non-temporal loads; the
 * row-block shape of the dense pass), `reps` timed launches each.  bench.py prints it as
 * roofline.peak_measured next to the data-sheet peak.  *pattern (may be NULL): 0 or 1, which one won. */
int PogsAmdReadBandwidth(int device, size_t bytes, int reps, double *gb_per_s, int *pattern);
/* Diagnostic: the 64-lane wavefront sums behind every row dot-product and scalar of the engine are formed in the
 * vector ALU (v_permlane32_swap / v_permlane16_swap / DPP, csrc/reduce.h) in the order of the butterfly
 * `v += shfl_xor(v, 32, 16, 8, 4, 2, 1)`.  For n (a multiple of 64) HOST values this returns, per value, its
 * wavefront's total formed that way (alu) and by the butterfly through the LDS crossbar (lds): the two must agree
 * bit for bit (tests/test_gpu_dense.py). */
int PogsAmdWaveSumCheck(int dtype, size_t n, const void *in_host, void *alu_host, void *lds_host);
/* The Norm2Est start vector (reference: gsl::rand, src/cpu/include/gsl/gsl_rand.h:8-16). */
int PogsAmdRandUniform(int dtype, size_t n, void *out_host);

#ifdef __cplusplus
}
#endif
#endif /* POGS_AMD_H_ */
