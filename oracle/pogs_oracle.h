/* oracle/pogs_oracle.h -- TEST INFRASTRUCTURE (see pogs_oracle.cpp header).
 * C interface of the CPU restatement of the reference POGS hot path.  Only
 * tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may use it. */
#ifndef POGS_ORACLE_H_
#define POGS_ORACLE_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Sum-all-reduce of `count` doubles in place (row-sharded variant only). */
typedef void (*oracle_allreduce_fn)(void *ctx, double *buf, size_t count);

/* Optional introspection block.  Zero-initialise before the call. */
typedef struct OracleInfo {
  /* in */
  int use_cgls;            /* dense entry only: use the CGLS projector       */
  double *d_out;           /* optional, length m: equilibration row scaling   */
  double *e_out;           /* optional, length n: equilibration col scaling   */
  const void *warm_x;      /* optional, length n, element type T: x0           */
  const void *warm_l;      /* optional, length m, element type T: lambda0      */
  /* out */
  double nrmA;             /* Norm2Est of the equilibrated matrix             */
  unsigned norm_est_iters; /* power iterations executed                       */
  double rho_final;
  unsigned exact_iters;    /* iterations that evaluated exact residuals       */
  long cg_iters;           /* total CGLS inner iterations                     */
  long n_mul;              /* operator applications (A. or A^T.) incl. init   */
  double t_init, t_loop;   /* seconds                                         */
} OracleInfo;

#define ORACLE_DECL_DENSE(NAME, T)                                                   \
  int NAME(int ord, size_t m, size_t n, const T *A, const T *f_a, const T *f_b,      \
           const T *f_c, const T *f_d, const T *f_e, const int *f_h, const T *g_a,   \
           const T *g_b, const T *g_c, const T *g_d, const T *g_e, const int *g_h,   \
           T rho, T abs_tol, T rel_tol, unsigned max_iter, unsigned verbose,         \
           int adaptive_rho, int gap_stop, T *x, T *y, T *l, T *optval,              \
           unsigned *final_iter, OracleInfo *info);
ORACLE_DECL_DENSE(OraclePogsD, double)
ORACLE_DECL_DENSE(OraclePogsS, float)

#define ORACLE_DECL_SHARD(NAME, T)                                                   \
  int NAME(size_t m_local, size_t m_global, size_t n, const T *A, const T *f_a,      \
           const T *f_b, const T *f_c, const T *f_d, const T *f_e, const int *f_h,   \
           const T *g_a, const T *g_b, const T *g_c, const T *g_d, const T *g_e,     \
           const int *g_h, T rho, T abs_tol, T rel_tol, unsigned max_iter,           \
           unsigned verbose, int adaptive_rho, int gap_stop, T *x, T *y, T *l,       \
           T *optval, unsigned *final_iter, OracleInfo *info,                        \
           oracle_allreduce_fn fn, void *ctx);
ORACLE_DECL_SHARD(OraclePogsShardD, double)
ORACLE_DECL_SHARD(OraclePogsShardS, float)

#define ORACLE_DECL_SPARSE(NAME, T)                                                  \
  int NAME(int ord, size_t m, size_t n, size_t nnz, const T *data, const int *ptr,   \
           const int *ind, const T *f_a, const T *f_b, const T *f_c, const T *f_d,   \
           const T *f_e, const int *f_h, const T *g_a, const T *g_b, const T *g_c,   \
           const T *g_d, const T *g_e, const int *g_h, T rho, T abs_tol, T rel_tol,  \
           unsigned max_iter, unsigned verbose, int adaptive_rho, int gap_stop,      \
           T *x, T *y, T *l, T *optval, unsigned *final_iter, OracleInfo *info);
#define ORACLE_DECL_SPARSE_SHARD(NAME, T)                                            \
  int NAME(size_t m_local, size_t m_global, size_t n, size_t nnz, const T *data,     \
           const int *ptr, const int *ind, const T *f_a, const T *f_b, const T *f_c, \
           const T *f_d, const T *f_e, const int *f_h, const T *g_a, const T *g_b,   \
           const T *g_c, const T *g_d, const T *g_e, const int *g_h, T rho,          \
           T abs_tol, T rel_tol, unsigned max_iter, unsigned verbose,                \
           int adaptive_rho, int gap_stop, T *x, T *y, T *l, T *optval,              \
           unsigned *final_iter, OracleInfo *info, oracle_allreduce_fn fn, void *ctx);
ORACLE_DECL_SPARSE_SHARD(OraclePogsSparseShardD, double)
ORACLE_DECL_SPARSE_SHARD(OraclePogsSparseShardS, float)

ORACLE_DECL_SPARSE(OraclePogsSparseD, double)
ORACLE_DECL_SPARSE(OraclePogsSparseS, float)

#define ORACLE_DECL_PROX(NAME, FNAME, T)                                             \
  void NAME(size_t n, const int *h, const T *a, const T *b, const T *c, const T *d,  \
            const T *e, T rho, const T *in, T *out);                                 \
  double FNAME(size_t n, const int *h, const T *a, const T *b, const T *c,           \
               const T *d, const T *e, const T *in);
ORACLE_DECL_PROX(OracleProxEvalD, OracleFuncEvalD, double)
ORACLE_DECL_PROX(OracleProxEvalS, OracleFuncEvalS, float)

#define ORACLE_DECL_PROJSUB(NAME, T)                                                  \
  void NAME(size_t n, const int *h, const T *a, const T *b, const T *c, const T *d,  \
            const T *e, const T *x_in, const T *v_in, T *v_out);
ORACLE_DECL_PROJSUB(OracleProjSubgradEvalD, double)
ORACLE_DECL_PROJSUB(OracleProjSubgradEvalS, float)

double OracleProxRawD(int h, double v, double rho);
float OracleProxRawS(int h, float v, float rho);
void OracleRandS(float *x, size_t n);
void OracleRandD(double *x, size_t n);

void OracleProjectD(size_t m, size_t n, const double *A, const double *x0,
                    const double *y0, double s, double tol, int use_cgls, double *x,
                    double *y);
void OracleProjectS(size_t m, size_t n, const float *A, const float *x0,
                    const float *y0, float s, float tol, int use_cgls, float *x,
                    float *y);
void OracleEquilD(size_t m, size_t n, double *A_inout, double *d, double *e,
                  double *nrmA, unsigned *kpow);
void OracleEquilS(size_t m, size_t n, float *A_inout, float *d, float *e, float *nrmA,
                  unsigned *kpow);

#ifdef __cplusplus
}
#endif
#endif /* POGS_ORACLE_H_ */
