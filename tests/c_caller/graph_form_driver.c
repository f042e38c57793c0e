/* A plain-C caller of the drop-in boundary (include/pogs_amd.h part 1), the way a C user of the
 * reference calls PogsD / PogsS / PogsSparseD (reference call site: examples/c/lasso.c:102-106).
 * Test infrastructure: tests/test_gpu_boundary.py compiles it with gcc against libpogs_amd.so and
 * compares what it prints with the python binding and the oracle on the same inputs; the CPU suite
 * only compiles and links it (C99, -Wall -Wextra -Werror -pedantic).
 *
 *   graph_form_driver <dense64|dense32|colmajor64|csr64|handle64> m n lambda <file>
 *
 * handle64: the additive part of the header from C -- PogsAmdCreateDense, then PogsAmdSolveFn with every field of f and g
 * but f.b given as a broadcast value (a PogsAmdFn whose NULL pointers stand for a0 .. e0, h0), PogsAmdDestroy.
 *
 * <file>: raw doubles, A (m x n, row-major; for csr64 its zeros are the sparsity pattern) followed
 * by b (m).  Output: one "key value..." line per result. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pogs_amd.h"

static void print_vec(const char *key, const double *v, size_t n) {
  size_t i;
  printf("%s", key);
  for (i = 0; i < n; ++i) printf(" %.17g", v[i]);
  printf("\n");
}

int main(int argc, char **argv) {
  size_t m, n, i, j, nnz = 0;
  double lambda, optval = 0;
  unsigned int final_iter = 0;
  int status = -1, kind;
  double *A, *b, *fa, *fb, *fc, *fd, *fe, *ga, *gb, *gc, *gd, *ge, *x, *y, *l;
  enum FUNCTION *fh, *gh;
  FILE *in;
  if (argc != 6) return 2;
  kind = !strcmp(argv[1], "dense64") ? 0 : !strcmp(argv[1], "dense32") ? 1 : !strcmp(argv[1], "colmajor64") ? 2
         : !strcmp(argv[1], "csr64") ? 3 : !strcmp(argv[1], "handle64") ? 4 : -1;
  if (kind < 0) return 2;
  m = (size_t)strtoul(argv[2], NULL, 10);
  n = (size_t)strtoul(argv[3], NULL, 10);
  lambda = strtod(argv[4], NULL);
  A = malloc(m * n * sizeof *A); b = malloc(m * sizeof *b);
  fa = malloc(m * sizeof *fa); fb = malloc(m * sizeof *fb); fc = malloc(m * sizeof *fc);
  fd = malloc(m * sizeof *fd); fe = malloc(m * sizeof *fe); fh = malloc(m * sizeof *fh);
  ga = malloc(n * sizeof *ga); gb = malloc(n * sizeof *gb); gc = malloc(n * sizeof *gc);
  gd = malloc(n * sizeof *gd); ge = malloc(n * sizeof *ge); gh = malloc(n * sizeof *gh);
  x = calloc(n, sizeof *x); y = calloc(m, sizeof *y); l = calloc(m, sizeof *l);
  if (!A || !b || !fa || !fb || !fc || !fd || !fe || !fh || !ga || !gb || !gc || !gd || !ge || !gh || !x || !y || !l) return 3;
  in = fopen(argv[5], "rb");
  if (!in || fread(A, sizeof *A, m * n, in) != m * n || fread(b, sizeof *b, m, in) != m) return 4;
  fclose(in);
  /* lasso: f_i = 1/2 (y_i - b_i)^2, g_j = lambda |x_j| */
  for (i = 0; i < m; ++i) { fa[i] = 1; fb[i] = b[i]; fc[i] = 1; fd[i] = 0; fe[i] = 0; fh[i] = SQUARE; }
  for (j = 0; j < n; ++j) { ga[j] = 1; gb[j] = 0; gc[j] = lambda; gd[j] = 0; ge[j] = 0; gh[j] = ABS; }

  if (kind == 0) {
    status = PogsD(ROW_MAJ, m, n, A, fa, fb, fc, fd, fe, fh, ga, gb, gc, gd, ge, gh, 1.0, 1e-4, 1e-4, 2500u, 0u, 1, 1,
                   x, y, l, &optval, &final_iter);
  } else if (kind == 2) {
    double *At = malloc(m * n * sizeof *At);
    if (!At) return 3;
    for (i = 0; i < m; ++i)
      for (j = 0; j < n; ++j) At[j * m + i] = A[i * n + j];
    status = PogsD(COL_MAJ, m, n, At, fa, fb, fc, fd, fe, fh, ga, gb, gc, gd, ge, gh, 1.0, 1e-4, 1e-4, 2500u, 0u, 1, 1,
                   x, y, l, &optval, &final_iter);
    free(At);
  } else if (kind == 4) {
    PogsAmdSolver *h = NULL;
    PogsAmdFn f, g;
    double *mu = calloc(n, sizeof *mu);
    if (!mu) return 3;
    memset(&f, 0, sizeof f);
    memset(&g, 0, sizeof g);
    f.b = fb;                                    /* the one per-element field of a lasso */
    f.a0 = 1; f.c0 = 1; f.d0 = 0; f.e0 = 0; f.h0 = SQUARE;
    g.a0 = 1; g.b0 = 0; g.c0 = lambda; g.d0 = 0; g.e0 = 0; g.h0 = ABS;
    status = PogsAmdCreateDense(&h, POGS_AMD_F64, ROW_MAJ, m, n, A, POGS_AMD_HOST, NULL, NULL);
    if (status == 0)
      status = PogsAmdSolveFn(h, &f, &g, 1.0, 1e-4, 1e-4, 2500u, 0u, 1, 1, x, y, l, mu, &optval, &final_iter);
    PogsAmdDestroy(h);
    free(mu);
  } else if (kind == 1) {
    /* every array narrowed to float, results widened for printing */
    float *A32 = malloc(m * n * sizeof *A32), *c32 = malloc((5 * (m + n)) * sizeof *c32);
    float *x32 = calloc(n, sizeof *x32), *y32 = calloc(m, sizeof *y32), *l32 = calloc(m, sizeof *l32), opt32 = 0;
    float *f32 = c32, *g32 = c32 + 5 * m;
    if (!A32 || !c32 || !x32 || !y32 || !l32) return 3;
    for (i = 0; i < m * n; ++i) A32[i] = (float)A[i];
    for (i = 0; i < m; ++i) { f32[i] = 1; f32[m + i] = (float)b[i]; f32[2 * m + i] = 1; f32[3 * m + i] = 0; f32[4 * m + i] = 0; }
    for (j = 0; j < n; ++j) { g32[j] = 1; g32[n + j] = 0; g32[2 * n + j] = (float)lambda; g32[3 * n + j] = 0; g32[4 * n + j] = 0; }
    status = PogsS(ROW_MAJ, m, n, A32, f32, f32 + m, f32 + 2 * m, f32 + 3 * m, f32 + 4 * m, fh, g32, g32 + n, g32 + 2 * n,
                   g32 + 3 * n, g32 + 4 * n, gh, 1.0f, 1e-4f, 1e-4f, 2500u, 0u, 1, 1, x32, y32, l32, &opt32, &final_iter);
    for (j = 0; j < n; ++j) x[j] = x32[j];
    for (i = 0; i < m; ++i) { y[i] = y32[i]; l[i] = l32[i]; }
    optval = opt32;
    free(A32); free(c32); free(x32); free(y32); free(l32);
  } else {
    int *ptr = malloc((m + 1) * sizeof *ptr), *ind;
    double *val;
    if (!ptr) return 3;
    for (i = 0; i < m * n; ++i) nnz += A[i] != 0.0;
    ind = malloc((nnz + 1) * sizeof *ind); val = malloc((nnz + 1) * sizeof *val);
    if (!ind || !val) return 3;
    nnz = 0;
    for (i = 0; i < m; ++i) {
      ptr[i] = (int)nnz;
      for (j = 0; j < n; ++j)
        if (A[i * n + j] != 0.0) { ind[nnz] = (int)j; val[nnz] = A[i * n + j]; ++nnz; }
    }
    ptr[m] = (int)nnz;
    status = PogsSparseD(ROW_MAJ, m, n, nnz, val, ptr, ind, fa, fb, fc, fd, fe, fh, ga, gb, gc, gd, ge, gh, 1.0, 1e-4, 1e-4,
                         2500u, 0u, 1, 1, x, y, l, &optval, &final_iter);
    free(ptr); free(ind); free(val);
  }
  printf("status %d\nfinal_iter %u\noptval %.17g\n", status, final_iter, optval);
  print_vec("x", x, n);
  print_vec("y", y, m);
  print_vec("l", l, m);
  free(A); free(b); free(fa); free(fb); free(fc); free(fd); free(fe); free(fh);
  free(ga); free(gb); free(gc); free(gd); free(ge); free(gh); free(x); free(y); free(l);
  return status == POGS_SUCCESS ? 0 : 1;
}
