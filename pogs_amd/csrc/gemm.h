// One-time dense factorisation pieces on the matrix cores (gfx950 MFMA):
//   * gemm:      C = alpha * op(A) op(B) + beta * C, f32 / f64 MFMA 16x16x4 tiles
//   * gram:      G = A^T A (lower tiles) -- the SYRK of
//                src/cpu/projector/projector_direct_dense.cpp:62-81
//   * cholesky:  blocked right-looking L L^T = G (gsl_linalg.h:36-55), with the
//                diagonal-block inverses kept so the panel solve is a GEMM
//   * trtri:     W = L^{-1} by recursive doubling, so that the per-iteration
//                triangular solves (gsl_linalg.h:57-61, two cblas_?trsv) become
//                two fully parallel triangular matrix-vector products.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "common.h"

namespace pogs_amd {

template <typename T>
struct GemmArgs {
  int M, N, K;
  const T *A; size_t lda;
  const T *B; size_t ldb;
  T *C; size_t ldc;
  T alpha, beta;
  // split-K (ksplit > 1): unit u computes rows [ks*kchunk, (ks+1)*kchunk) of the K range into
  // C + ks*csplit_stride, so one launch keeps every CU busy to the end and the K-sum is formed
  // in chunks (better fp32 accumulation); the caller adds the ksplit slabs in fixed order.
  int ksplit = 1;
  int kchunk = 0;
  size_t csplit_stride = 0;
  int ks0 = 0;   // K range of slab ks is [(ks0 + ks) * kchunk, (ks0 + ks + 1) * kchunk)
  // kacc > 0 (multiple of 32; lower_only Gram launches): every kacc rows of the K range the
  // register accumulators are added into a second set and cleared, so a long fp32 K-sum is
  // formed as an ordered sum of short ones (same error behaviour as split-K, no slabs).
  int kacc = 0;
  // batch > 1: `batch` independent products of the same shape; product q uses A + q*strideA,
  // B + q*strideB, C + q*strideC (one launch for a whole TRTRI level).
  int batch = 1;
  size_t strideA = 0, strideB = 0, strideC = 0;
  // optional tile order (device pointer, one entry (tile_i << 16 | tile_j) per computed tile):
  // workgroups that run together then share operand panels in L2 (see gram_tile_order)
  const int *tile_map = nullptr;
  // ktri: a triangular operand's zero blocks are skipped (the products left out are exact zeros, so
  // the result has the same bits).  1: op(B)(k, j) = 0 for k < j (lower-triangular B in (k, j)): a
  // tile's K range starts at its first column.  2: op(A)(i, k) = 0 for k > i: it ends after its last row.
  int ktri = 0;
};

// Asks the runtime for a kernel of gemm.hip, which makes it load that code object (helper thread
// of DenseSolver's constructor).
void preload_gemm_code();

// G += / = P^T P (lower tiles) for a K-major fp32 operand P (K rows of N, leading dimension ld) on
// the fp16 matrix cores at fp32 accuracy: gfx950 has no reduced-precision fast path for fp32
// inputs, but a scaled operand a s (s a power of two that puts the largest entry near 2^14) splits
// into two fp16 numbers h + l with 22 significant bits, and the three products hh, hl, lh carry
// everything above 2^-22 relative -- accumulated in fp32 by v_mfma_f32_32x32x16_f16 and unscaled
// by 1 / s^2.
// launch_split_f16 writes rows [k0, k0 + krows) of P (zeros past K, zeros in columns [N, npad)) as
// two fp16 images H and L in MFMA-operand order: [k / 8][npad][8], i.e. 16 bytes = eight
// consecutive k of one column, columns contiguous -- a wavefront's async global->LDS copy of 64
// columns is one 1 KB line and lands in LDS exactly as the matrix cores read it.  launch_gram_f16p
// then only copies (no staging registers, no LDS stores, no conversion) and multiplies.  npad: a
// multiple of the tile; krows: a multiple of 32; unit u of the launch covers image rows
// [u kchunk, (u + 1) kchunk), kchunk a multiple of 32 (and of flush_rows), and goes to slab u.
void launch_split_f16(const float *P, size_t ld, int K, int N, int k0, int krows, int npad, float scale,
                      void *H, void *L, hipStream_t s);
struct GramF16PArgs {
  const void *H, *L; int npad, N;
  float *C; size_t ldc;
  int nslabs, kchunk; size_t slab_stride;
  int accumulate;
  const int *tile_map;
  float scale;
  int tile = 128;                 // workgroup tile: 128 (4 waves) or 256 (8 waves); tile_map must match
  int flush_rows = 0;             // 128 tile: rows per MFMA chain inside a unit (0: the whole kchunk)
};
void launch_gram_f16p(const GramF16PArgs &g, hipStream_t s);

// Lower-triangular tile order in 8 x 8 super-tiles for an n x n Gram product: the ~64
// workgroups an XCD runs at a time then touch 16 operand panels instead of 65.
std::vector<int> gram_tile_order(int n, int tile = 128);

// A_KMAJ: op(A)(i,k) = A[k*lda + i], else A[i*lda + k].
// B_KMAJ: op(B)(k,j) = B[k*ldb + j], else B[j*ldb + k].
// lower_only: only tiles with tile_j <= tile_i are computed (M == N).
// All leading dimensions and sub-matrix offsets must keep 16-byte alignment.
template <typename T>
void launch_gemm(bool a_kmaj, bool b_kmaj, bool lower_only, const GemmArgs<T> &g, hipStream_t s);

// Diagonal block size of the blocked Cholesky / first TRTRI level.
template <typename T> struct CholBlock;
#ifndef POGS_CHOL_NB_F32   // (compile-time overrides for A / B builds: scripts/build_variant.py <tag> gemm.hip -D...)
#define POGS_CHOL_NB_F32 128
#endif
#ifndef POGS_CHOL_GROUP_F32
#define POGS_CHOL_GROUP_F32 2
#endif
template <> struct CholBlock<float> { static constexpr int NB = POGS_CHOL_NB_F32; };
template <> struct CholBlock<double> { static constexpr int NB = 64; };

// G (n x n, lower triangle valid, leading dim ldg) is overwritten by its Cholesky
// factor L; W (n x n, leading dim ldw, zero-initialised by the caller) receives
// the inverses of L's diagonal blocks.
template <typename T>
void cholesky_lower(T *G, size_t ldg, int n, T *W, size_t ldw, hipStream_t s);

// Completes W = L^{-1} (lower triangular) from the diagonal-block inverses.
// tmp must hold n * ldw elements.
template <typename T>
void trtri_lower(const T *L, size_t ldg, int n, T *W, size_t ldw, T *tmp, hipStream_t s);

// out (cols x rows, ld_out) = in^T, in is rows x cols with ld_in.
template <typename T>
void launch_transpose(const T *in, size_t ld_in, int rows, int cols, T *out, size_t ld_out, hipStream_t s);

// out = sum_s in[s * stride + .] over the lower 128-tiles of an n x n matrix (leading dim ld,
// ld a multiple of the 16-byte vector), s in fixed order: combines split-K slabs, which a
// lower_only launch leaves unwritten above the diagonal tiles.
template <typename T>
void launch_sum_slabs(const T *in, size_t stride, int nslabs, T *out, size_t ld, int n, hipStream_t s);

// zero the strictly upper triangle of an n x n matrix
template <typename T>
void launch_zero_upper(T *G, size_t ldg, int n, hipStream_t s);

// Lower block-triangle of an n x n matrix (rows of 128-row block b keep their first
// min(ld, 128 (b + 1)) columns) <-> a contiguous buffer: what a row-sharded Gram matrix sums over
// the ranks -- about half of the n x ld square.  Returns the element count.
size_t packed_lower_count(int n, size_t ld);
template <typename T>
void launch_pack_lower(T *G, size_t ld, int n, T *packed, bool unpack, hipStream_t s);

// G[i][i] += v
template <typename T>
void launch_add_diag(T *G, size_t ldg, int n, T v, hipStream_t s);

}  // namespace pogs_amd
