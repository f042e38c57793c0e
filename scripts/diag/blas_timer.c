/* LD_PRELOAD interposer: wall time and call counts of the BLAS/LAPACK entry points the reference's
 * dense path uses (src/cpu/include/gsl/gsl_blas.h, gsl_linalg.h), printed at exit.  Diagnostic
 * only (scripts/ref_collapse_probe.py): where does the compiled reference spend its loop time?
 *   gcc -O2 -shared -fPIC blas_timer.c -o blas_timer.so -ldl                                  */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static double cpu(void) { struct timespec t; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
enum { GEMV_N, GEMV_T, TRSV, SYRK, GEMM, POTRF_LIKE, NRM2, DOT, AXPY, NFN };
static const char *names[NFN] = {"sgemv N", "sgemv T", "strsv", "ssyrk", "sgemm", "potrf/other", "snrm2", "sdot", "saxpy"};
static double tw[NFN], tc[NFN];
static long cnt[NFN];
static double t_first = 0;
static void report(void) {
  fprintf(stderr, "[blas_timer] wall since first call %.1f s\n", now() - t_first);
  for (int i = 0; i < NFN; ++i)
    if (cnt[i]) fprintf(stderr, "[blas_timer] %-12s calls %7ld  wall %8.2f s  cpu/wall %5.2f  avg %8.3f ms\n", names[i], cnt[i], tw[i], tc[i] / (tw[i] > 0 ? tw[i] : 1), 1e3 * tw[i] / cnt[i]);
}
static void *sym(const char *n) {
  static int reg = 0;
  if (!reg) { reg = 1; atexit(report); t_first = now(); }
  static void *mkl = NULL;
  if (!mkl) mkl = dlopen(getenv("BLAS_TIMER_LIB") ? getenv("BLAS_TIMER_LIB") : "/opt/conda/lib/libmkl_rt.so", RTLD_NOW | RTLD_GLOBAL);
  void *p = mkl ? dlsym(mkl, n) : dlsym(RTLD_NEXT, n);
  if (!p) { fprintf(stderr, "[blas_timer] no %s\n", n); abort(); }
  return p;
}
#define BEGIN double w0 = now(), c0 = cpu();
#define END(i) tw[i] += now() - w0; tc[i] += cpu() - c0; cnt[i]++;

void cblas_sgemv(int order, int trans, int m, int n, float alpha, const float *A, int lda, const float *x, int incx, float beta, float *y, int incy) {
  static void (*f)(int, int, int, int, float, const float *, int, const float *, int, float, float *, int);
  if (!f) f = sym("cblas_sgemv");
  BEGIN f(order, trans, m, n, alpha, A, lda, x, incx, beta, y, incy); END(trans == 111 ? GEMV_N : GEMV_T)
}
void cblas_strsv(int order, int uplo, int trans, int diag, int n, const float *A, int lda, float *x, int incx) {
  static void (*f)(int, int, int, int, int, const float *, int, float *, int);
  if (!f) f = sym("cblas_strsv");
  BEGIN f(order, uplo, trans, diag, n, A, lda, x, incx); END(TRSV)
}
void cblas_ssyrk(int order, int uplo, int trans, int n, int k, float alpha, const float *A, int lda, float beta, float *C, int ldc) {
  static void (*f)(int, int, int, int, int, float, const float *, int, float, float *, int);
  if (!f) f = sym("cblas_ssyrk");
  BEGIN f(order, uplo, trans, n, k, alpha, A, lda, beta, C, ldc); END(SYRK)
}
float cblas_snrm2(int n, const float *x, int incx) {
  static float (*f)(int, const float *, int);
  if (!f) f = sym("cblas_snrm2");
  BEGIN float r = f(n, x, incx); END(NRM2) return r;
}
float cblas_sdot(int n, const float *x, int incx, const float *y, int incy) {
  static float (*f)(int, const float *, int, const float *, int);
  if (!f) f = sym("cblas_sdot");
  BEGIN float r = f(n, x, incx, y, incy); END(DOT) return r;
}
void cblas_saxpy(int n, float a, const float *x, int incx, float *y, int incy) {
  static void (*f)(int, float, const float *, int, float *, int);
  if (!f) f = sym("cblas_saxpy");
  BEGIN f(n, a, x, incx, y, incy); END(AXPY)
}
