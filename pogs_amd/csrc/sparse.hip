// Sparse graph-form ADMM solver with the CGLS projector.
//
// Reference call stack being replaced:
//   PogsSparseD/S -> PogsSparse<T,O> (src/interface_c/pogs_c.cpp:57-108)
//     MatrixSparse::Init/Mul/Equil (src/cpu/matrix/matrix_sparse.cpp:97-302,
//       gsl_spblas.h:10-40, gsl_spmat.h:32-93): CSR plus its transpose, both
//       used as row-gather SpMVs
//     ProjectorCgls::Project (src/cpu/projector/projector_cgls.cpp:52-88)
//       -> cgls::Solve (src/cpu/include/cgls.h:200-323)
//     PogsImplementation::Solve (src/cpu/pogs.cpp:91-581)
//
// HBM layout: two CSR structures (A by rows, A^T by rows; int32 indices) -- kept for GetEquil,
// as the source of the tiled copies and as the fallback (POGS_AMD_SPMV=plain): "row blocks" of
// consecutive rows whose non-zeros fit one LDS tile, streamed with coalesced loads, products
// staged in LDS, rows reduced from there.  The solver itself runs on a tiled sliced-ELL copy of
// each (sell.h): x slices and row sums in LDS, uint16 local columns, no partial-sum traffic.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <limits>
#include <type_traits>
#include <vector>

#include "cg_fused.h"
#include "cg_kernels.h"
#include "engine.h"
#include "reduce.h"
#include "sell.h"
#include "vec_kernels.h"

namespace pogs_amd {

void rand_uniform_host(float *x, size_t n);
void rand_uniform_host(double *x, size_t n);

namespace {

constexpr int kSpTpb = 256;
constexpr int kSpCap = 4096;       // non-zeros staged in LDS per row block
constexpr int kSpMaxRows = 2048;   // rows per block cap (balance when rows are empty)

// ---------------------------------------------------------------------------
// Row functors (one thread per finished row; scalars accumulate in doubles)
// ---------------------------------------------------------------------------
template <typename T>
struct SpAxpbyOp {  // y[i] = alpha * dot + beta * yin[i]
  static constexpr int NS = 0;
  T alpha, beta;
  const T *yin;
  T *y;
  template <int N>
  __device__ __forceinline__ void row(int i, T dot, double (&)[N]) const {
    T v = alpha * dot;
    if (beta != static_cast<T>(0)) v += beta * yin[i];
    y[i] = v;
  }
};

template <typename T>
struct SpStoreOp {  // y[i] = dot, no sums: the local part of a row-sharded A^T product, before its all-reduce
  static constexpr int NS = 0;
  T *y;
  template <int N>
  __device__ __forceinline__ void row(int i, T dot, double (&)[N]) const { y[i] = dot; }
  struct In {};
  __device__ __forceinline__ In load(int) const { return In{}; }
  template <int N>
  __device__ __forceinline__ void apply(int i, T dot, const In &, double (&)[N]) const { y[i] = dot; }
};

template <typename T>
struct SpAxpbyNormOp {  // y[i] = alpha * dot + beta * yin[i]; s0 += y[i]^2
  static constexpr int NS = 1;
  T alpha, beta;
  const T *yin;
  T *y;
  template <int N>
  __device__ __forceinline__ void row(int i, T dot, double (&s)[N]) const {
    T v = alpha * dot;
    if (beta != static_cast<T>(0)) v += beta * yin[i];
    y[i] = v;
    dev::prod_acc(s[0], v, v);
  }
  // the same in two steps (sell.h: spmv_sell_fin_kernel requests the operands of several rows
  // before it uses any)
  struct In { T yin; };
  __device__ __forceinline__ In load(int i) const { return In{beta != static_cast<T>(0) ? yin[i] : static_cast<T>(0)}; }
  template <int N>
  __device__ __forceinline__ void apply(int i, T dot, const In &in, double (&s)[N]) const {
    T v = alpha * dot;
    if (beta != static_cast<T>(0)) v += beta * in.yin;
    y[i] = v;
    dev::prod_acc(s[0], v, v);
  }
};

template <typename T>
struct SpSkOp {  // out[i] = num / (dot + c)   (equil_helper.h:149-162)
  static constexpr int NS = 1;
  T num, c;
  T *out;
  // common-factor probe, see SkColOp in ops.h: sums new / old, stamps *mark when an entry's ratio
  // leaves r_ref by more than tol
  double *mark = nullptr;
  double stamp = 0;
  T tol = 0;
  T r_ref = 0;
  template <int N>
  __device__ __forceinline__ void row(int i, T dot, double (&s)[N]) const {
    const T v = num / (dot + c);
    const T old = out[i];
    const T r = old > static_cast<T>(0) ? v / old : static_cast<T>(0);
    s[0] += static_cast<double>(r);
    if (mark && !(fabs(r - r_ref) <= tol * r_ref)) *mark = stamp;
    out[i] = v;
  }
};

template <typename T>
struct SpTailOp {  // ProjTailOp for the y half: see ops.h
  static constexpr int NS = 2;
  T *znew;
  const T *zprev, *z12;
  T *ztemp;
  template <int N>
  __device__ __forceinline__ void row(int i, T dot, double (&s)[N]) const {
    znew[i] = dot;
    const T a = zprev[i] - dot, b = z12[i] - dot;
    dev::prod_acc(s[0], a, a);
    dev::prod_acc(s[1], b, b);
    ztemp[i] -= dot;
  }
  struct In { T zprev, z12, ztemp; };
  __device__ __forceinline__ In load(int i) const { return In{zprev[i], z12[i], ztemp[i]}; }
  template <int N>
  __device__ __forceinline__ void apply(int i, T dot, const In &in, double (&s)[N]) const {
    znew[i] = dot;
    const T a = in.zprev - dot, b = in.z12 - dot;
    dev::prod_acc(s[0], a, a);
    dev::prod_acc(s[1], b, b);
    ztemp[i] = in.ztemp - dot;
  }
};

template <typename T>
struct SpExactROp {  // r_i = (A x12)_i - y12_i (pogs.cpp:353-364)
  static constexpr int NS = 1;
  const T *y12;
  template <int N>
  __device__ __forceinline__ void row(int i, T dot, double (&s)[N]) const {
    const T r = dot - y12[i];
    dev::prod_acc(s[0], r, r);
  }
};

template <typename T>
struct SpExactSOp {  // s_j = (A^T u)_j + x12_j + c xt_j - xprev_j (pogs.cpp:366-373)
  static constexpr int NS = 1;
  const T *x12, *xt, *xprev;
  T zt_scale;
  template <int N>
  __device__ __forceinline__ void row(int j, T dot, double (&s)[N]) const {
    const T v = dot + x12[j] + zt_scale * xt[j] - xprev[j];
    dev::prod_acc(s[0], v, v);
  }
};

// ---------------------------------------------------------------------------
// SpMV kernel (CSR-stream with LDS staging)
// ---------------------------------------------------------------------------
template <typename T>
struct Csr {
  const T *val;
  const int *ind, *ptr, *blocks;
  int nrows, nblocks;
};

template <typename T, bool SQ, typename Op>
__global__ void __launch_bounds__(kSpTpb) spmv_kernel(Csr<T> A, const T *__restrict__ x, const double *x_nrm2,
                                                      Op op, double *scalar_partials) {
  constexpr int NS = Op::NS > 0 ? Op::NS : 1;
  __shared__ T s_prod[kSpCap];
  __shared__ T s_long[kSpTpb / 64];
  __shared__ double s_red[NS * (kSpTpb / 64)];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  double sacc[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) sacc[k] = 0.0;
  T xs = 1;
  if (x_nrm2) xs = static_cast<T>(1.0 / sqrt(*x_nrm2));

  for (int b = blockIdx.x; b < A.nblocks; b += gridDim.x) {
    const int r0 = A.blocks[b], r1 = A.blocks[b + 1];
    const int p0 = A.ptr[r0], p1 = A.ptr[r1];
    const int cnt = p1 - p0;
    if (cnt > kSpCap) {
      // one long row: the whole workgroup strides over it
      T s = 0;
      for (int k = t; k < cnt; k += kSpTpb) {
        T v = A.val[p0 + k];
        if (SQ) v *= v;
        s = sell_fma(v, x[A.ind[p0 + k]] * xs, s);
      }
      s = dev::wave_sum(s);
      if (lane == 0) s_long[wave] = s;
      __syncthreads();
      if (t == 0) {
        T tot = 0;
#pragma unroll
        for (int w = 0; w < kSpTpb / 64; ++w) tot += s_long[w];
        op.row(r0, tot, sacc);
      }
      __syncthreads();
      continue;
    }
    // stage val * x[ind] in LDS: 8 independent coalesced value/index loads and 8
    // gathers in flight per thread
    constexpr int U = 8;
    for (int k0 = 0; k0 < cnt; k0 += kSpTpb * U) {
      T v[U];
      int id[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = k0 + u * kSpTpb + t;
        const bool ok = k < cnt;
        v[u] = ok ? A.val[p0 + k] : static_cast<T>(0);
        id[u] = ok ? A.ind[p0 + k] : 0;
      }
      T xg[U];
#pragma unroll
      for (int u = 0; u < U; ++u) xg[u] = x[id[u]];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = k0 + u * kSpTpb + t;
        if (k < cnt) s_prod[k] = (SQ ? v[u] * v[u] : v[u]) * (xg[u] * xs);
      }
    }
    __syncthreads();
    const int nrows = r1 - r0;
    int tpr = 1;  // threads per row: a power of two <= 64, about a quarter of the mean row length
    while (tpr < 64 && tpr * 4 < cnt / (nrows > 0 ? nrows : 1)) tpr <<= 1;
    const int rpp = kSpTpb / tpr, lir = t % tpr, slot = t / tpr;
    for (int base = 0; base < nrows; base += rpp) {
      const int r = r0 + base + slot;
      T s = 0;
      if (r < r1) {
        const int a = A.ptr[r] - p0, e = A.ptr[r + 1] - p0;
        for (int k = a + lir; k < e; k += tpr) s += s_prod[k];
      }
      for (int off = tpr >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      if (lir == 0 && r < r1) op.row(r, s, sacc);
    }
    __syncthreads();
  }
  if (Op::NS > 0) {
    dev::block_sum<NS, kSpTpb>(sacc, s_red);
    if (t == 0) {
#pragma unroll
      for (int k = 0; k < NS; ++k) scalar_partials[static_cast<size_t>(blockIdx.x) * NS + k] = sacc[k];
    }
  }
}

// row r: sum of its ncb partial sums (one per column group) in group order, then the row functor
// (Measured and not kept: letting the workgroup that finishes last -- a device counter -- add the
// scalar partials and form the CGLS scalar, in place of the launch_sum_cg launch that follows.
// 2048 workgroups incrementing one address serialise in L2: +50 us per SpMV.)
template <typename T, typename Op>
__global__ void __launch_bounds__(256) reduce_parts_kernel(const T *__restrict__ part, int nrows, int ncb, Op op,
                                                           double *scalar_partials) {
  constexpr int NS = Op::NS > 0 ? Op::NS : 1;
  __shared__ double s_red[NS * 4];
  double sacc[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) sacc[k] = 0.0;
  for (int r = blockIdx.x * 256 + threadIdx.x; r < nrows; r += gridDim.x * 256) {
    T sum = part[r];
    for (int cb = 1; cb < ncb; ++cb) sum += part[static_cast<size_t>(cb) * nrows + r];
    op.row(r, sum, sacc);
  }
  if (Op::NS > 0) {
    dev::block_sum<NS, 256>(sacc, s_red);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < NS; ++k) scalar_partials[static_cast<size_t>(blockIdx.x) * NS + k] = sacc[k];
    }
  }
}

// the row functor applied to a finished vector of dot products (row-sharded solves: the
// A^T products are summed over the ranks before the functor sees them)
template <typename T, typename Op>
__global__ void __launch_bounds__(256) apply_rows_kernel(const T *__restrict__ dots, int nrows, Op op,
                                                         double *scalar_partials) {
  constexpr int NS = Op::NS > 0 ? Op::NS : 1;
  __shared__ double s_red[NS * 4];
  double sacc[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) sacc[k] = 0.0;
  for (int r = blockIdx.x * 256 + threadIdx.x; r < nrows; r += gridDim.x * 256) op.row(r, dots[r], sacc);
  if (Op::NS > 0) {
    dev::block_sum<NS, 256>(sacc, s_red);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < NS; ++k) scalar_partials[static_cast<size_t>(blockIdx.x) * NS + k] = sacc[k];
    }
  }
}

// ---------------------------------------------------------------------------
// One-time structure kernels
// ---------------------------------------------------------------------------
// *err |= 1 if ptr decreases somewhere, 2 if an index lies outside [0, ncols): checked before any
// kernel scatters through these arrays (a malformed CSR / CSC is an error return, not a fault)
__global__ void validate_csr_kernel(const int *ind, const int *ptr, int nrows, int ncols, size_t nnz, int *err) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  const size_t t0 = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  int bad = 0;
  for (size_t r = t0; r < static_cast<size_t>(nrows); r += stride)
    if (ptr[r + 1] < ptr[r]) bad |= 1;
  for (size_t k = t0; k < nnz; k += stride) {
    const int c = ind[k];
    if (c < 0 || c >= ncols) bad |= 2;
  }
  if (bad) atomicOr(err, bad);
}

__global__ void count_cols_kernel(const int *ind, size_t nnz, int *cnt) {
  for (size_t k = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; k < nnz;
       k += static_cast<size_t>(gridDim.x) * blockDim.x)
    atomicAdd(&cnt[ind[k]], 1);
}

// exclusive scan of cnt[0..n) into ptr[0..n], single workgroup of 1024 threads
__global__ void __launch_bounds__(1024) scan_kernel(const int *cnt, int n, int *ptr) {
  __shared__ int s_tot[1024];
  const int t = threadIdx.x;
  const int chunk = (n + 1023) / 1024;
  const int lo = t * chunk, hi = min(n, lo + chunk);
  int sum = 0;
  for (int i = lo; i < hi; ++i) sum += cnt[i];
  s_tot[t] = sum;
  __syncthreads();
  // Hillis-Steele inclusive scan over the 1024 chunk totals
  for (int off = 1; off < 1024; off <<= 1) {
    int v = (t >= off) ? s_tot[t - off] : 0;
    __syncthreads();
    s_tot[t] += v;
    __syncthreads();
  }
  int run = (t == 0) ? 0 : s_tot[t - 1];
  for (int i = lo; i < hi; ++i) {
    ptr[i] = run;
    run += cnt[i];
  }
  if (t == 1023) ptr[n] = s_tot[1023];
}

// Three-kernel exclusive scan for long arrays: per-tile (8192 items) local scan + tile totals,
// scan_kernel over the totals, then the tile offsets are added.
constexpr int kScanTile = 8192;

__global__ void __launch_bounds__(1024) scan_tiles_kernel(const int *cnt, int n, int *out, int *tile_tot) {
  __shared__ int s_w[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int base = blockIdx.x * kScanTile + t * 8;
  int v[8], sum = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    v[k] = (base + k < n) ? cnt[base + k] : 0;
    sum += v[k];
  }
  // inclusive scan of the thread sums: within the wave by shuffles, then across the 16 waves
  int inc = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(inc, off, 64);
    if (lane >= off) inc += o;
  }
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < wave; ++w) woff += s_w[w];
  int run = woff + inc - sum;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (base + k < n) out[base + k] = run;
    run += v[k];
  }
  if (t == 1023) tile_tot[blockIdx.x] = woff + inc;
}

__global__ void __launch_bounds__(1024) scan_add_kernel(int *out, int n, const int *tile_off, int ntiles) {
  const int base = blockIdx.x * kScanTile + threadIdx.x * 8;
  const int off = tile_off[blockIdx.x];
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (base + k < n) out[base + k] += off;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = tile_off[ntiles];
}

// scatter (row, val) of every non-zero into its column segment (order within a
// segment is fixed afterwards by sort_segments_kernel)
template <typename T>
__global__ void fill_transpose_kernel(const T *val, const int *ind, const int *ptr, int nrows, int *cursor,
                                      T *tval, int *tind) {
  const int lane = threadIdx.x & 63;
  const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nw = (gridDim.x * blockDim.x) >> 6;
  for (int r = w; r < nrows; r += nw) {
    for (int k = ptr[r] + lane; k < ptr[r + 1]; k += 64) {
      const int pos = atomicAdd(&cursor[ind[k]], 1);
      tind[pos] = r;
      tval[pos] = val[k];
    }
  }
}

// Sorts each segment by index with an all-ascending bitonic network, so the transposed matrix is
// exactly what the reference's stable csr2csc builds (gsl_spmat.h:32-55).  An input that repeats
// an entry (the same column twice in a row: the reference's gather product simply adds both,
// gsl_spblas.h:16-40) leaves ties, which the scatter above delivers in no particular order: they
// are broken by the value's bit pattern, so the stored order -- and with it every sum -- is the
// same from run to run (the reference's order among such ties is their CSR order; the two differ
// only in the association of three or more equal-index terms).  One workgroup per segment; LDS
// when it fits.
template <typename T>
__global__ void __launch_bounds__(256) sort_segments_kernel(const int *ptr, int nseg, int *ind, T *val) {
  constexpr int CAP = 2048;
  __shared__ int s_i[CAP];
  __shared__ T s_v[CAP];
  for (int seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
    const int p0 = ptr[seg], len = ptr[seg + 1] - p0;
    if (len <= 1) continue;
    const bool lds = len <= CAP;
    int *ki = lds ? s_i : ind + p0;
    T *kv = lds ? s_v : val + p0;
    if (lds) {
      for (int k = threadIdx.x; k < len; k += 256) { s_i[k] = ind[p0 + k]; s_v[k] = val[p0 + k]; }
    }
    __syncthreads();
    int np2 = 1;
    while (np2 < len) np2 <<= 1;
    for (int k = 2; k <= np2; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = threadIdx.x; i < np2; i += 256) {
          const int l = (j == (k >> 1)) ? (i ^ (k - 1)) : (i ^ j);
          if (l > i && l < len) {  // elements >= len act as +inf and never move
            const int a = ki[i], b = ki[l];
            const T va = kv[i], vb = kv[l];
            bool swap = a > b;
            if (a == b) {
              typename std::conditional<sizeof(T) == 4, unsigned, unsigned long long>::type ba, bb;
              __builtin_memcpy(&ba, &va, sizeof(T));
              __builtin_memcpy(&bb, &vb, sizeof(T));
              swap = ba > bb;
            }
            if (swap) {
              ki[i] = b; ki[l] = a;
              kv[i] = vb; kv[l] = va;
            }
          }
        }
        __syncthreads();
      }
    }
    if (lds) {
      for (int k = threadIdx.x; k < len; k += 256) { ind[p0 + k] = s_i[k]; val[p0 + k] = s_v[k]; }
    }
    __syncthreads();
  }
}

// val[k] *= drow[row] * ecol[ind[k]], one wavefront per row; partial sum of squares
template <typename T>
__global__ void __launch_bounds__(256) scale_csr_kernel(T *val, const int *ind, const int *ptr, int nrows,
                                                        const T *drow, const T *ecol, double *partials) {
  __shared__ double s_red[4];
  const int lane = threadIdx.x & 63;
  const int w = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = (gridDim.x * 256) >> 6;
  double acc[1] = {0.0};
  for (int r = w; r < nrows; r += nw) {
    const T dr = drow[r];
    for (int k = ptr[r] + lane; k < ptr[r + 1]; k += 64) {
      const T v = val[k] * (dr * ecol[ind[k]]);
      val[k] = v;
      dev::prod_acc(acc[0], v, v);
    }
  }
  dev::block_sum<1, 256>(acc, s_red);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}

// u = y12 + c yt - yprev   (pogs.cpp:366-368, y half)
template <typename T>
__global__ void exact_u_kernel(int m, const T *y12, const T *yt, const T *yprev, T c, T *u) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) u[i] = y12[i] + c * yt[i] - yprev[i];
}


// ---------------------------------------------------------------------------
template <typename T>
struct DevCsr {
  DevBuf<T> val;
  DevBuf<int> ind, ptr, blocks;
  int nrows = 0, ncols = 0, nblocks = 0;
  size_t nnz = 0;
  Csr<T> view() const { return Csr<T>{val.p, ind.p, ptr.p, blocks.p, nrows, nblocks}; }
  // tiled lane-stream copy (sell.h); sell_ready == false: not built, the plain kernel runs
  DevBuf<T> sval, part;
  DevBuf<unsigned short> sloc, srid;
  DevBuf<int> tile_unit;
  DevBuf<unsigned short> scnt;   // build temporaries kept until the values are final (refill_sell)
  DevBuf<unsigned> ssoff;
  DevBuf<unsigned> sdst;         // position of every CSR element in the tiled copy, kept from the first fill to the refill
  bool sell_ready = false;
  int rr_rows = 0, nrr = 0, ncb = 0, ncg = 1;
  int two = 0;                   // storage format (SellView::two)
  size_t sell_elems = 0;
  DevBuf<unsigned long long> stamps;      // debug time stamps (POGS_AMD_SELL_STAMPS)
  bool stamps_on = false;
  SellDims sdims() const { return SellDims{nrows, ncols, rr_rows, nrr, ncb, SellCfg<T>::BW}; }
  SellView<T> sview() const {
    return SellView<T>{sval.p, sloc.p, srid.p, tile_unit.p, nrows, ncols, rr_rows, nrr, ncb, ncg, two,
                       stamps_on ? stamps.p : nullptr};
  }
};

std::vector<int> make_row_blocks(const std::vector<int> &ptr, int nrows) {
  std::vector<int> blocks;
  blocks.push_back(0);
  int start = 0;
  while (start < nrows) {
    int end = start;
    long long cnt = 0;
    while (end < nrows && end - start < kSpMaxRows) {
      const long long rn = ptr[end + 1] - ptr[end];
      if (cnt + rn > kSpCap) break;
      cnt += rn;
      ++end;
    }
    if (end == start) ++end;  // a single row longer than a tile
    blocks.push_back(end);
    start = end;
  }
  return blocks;
}

template <typename T>
class SparseSolver final : public SolverBase {
 public:
  SparseSolver(int ord, size_t m, size_t n, size_t nnz, const void *data, const int *ptr, const int *ind, int mem,
               const PogsAmdOptions *opt, const PogsAmdDist *dist) {
    const double t0 = wall_s();
    ctx_.init(opt ? opt->device : -1, opt ? opt->profile : 0);
    POGS_CHECK(m > 0 && n > 0 && m < (1u << 31) && n < (1u << 31) && nnz < (1ull << 31), "bad dimensions");
    m_ = static_cast<int>(m);
    n_ = static_cast<int>(n);
    nnz_ = nnz;
    // row shards: this rank holds m consecutive rows as CSR; y-sized
    // state is local, x-sized state replicated, A^T products and row sums are all-reduced
    if (dist && dist->world >= 1) {
      POGS_CHECK(ord == ROW_MAJ, "a row shard must be given as CSR (ROW_MAJ)");
      ctx_.dist.init(dist->rank, dist->world, dist->unique_id);
      ctx_.m_global = dist->m_global;
    } else {
      ctx_.m_global = m;
    }
    multi_ = ctx_.dist.active();
    build_structure(ord, data, ptr, ind, mem);
    ctx_.stats.t_h2d_s = wall_s() - t0;
    alloc_state();
    print_stamps();
    equilibrate();
    norm_est();
    ctx_.sync();
    ctx_.stats.t_init_s = wall_s() - t0;
  }

  ~SparseSolver() override { begin_destroy(ctx_); }

  int dtype() const override { return sizeof(T) == 4 ? POGS_AMD_F32 : POGS_AMD_F64; }
  int device() const override { return ctx_.device; }
  void on_entry() override { ctx_.on_entry(); }
  void on_error() override { ctx_.on_error(); }
  PogsAmdStats &stats() override { return ctx_.stats; }

  int solve(const FnHost &f, const FnHost &g, const SolveParams &p, void *x, void *y, void *l, void *mu,
            double *optval, unsigned *final_iter) override {
    const double t0 = wall_s();
    load_problem(f, g, p);
    cold_start();
    apply_warm_start();
    ctx_.sync();
    const double t1 = wall_s();
    if (ctx_.dist.rank() == 0) print_banner(p.verbose);
    while (!iteration(p.verbose)) {}
    ctx_.sync();
    const double t2 = wall_s();
    const int status = epilogue(x, y, l, mu, optval);
    *final_iter = ctl_.k;
    PogsAmdStats &st = ctx_.stats;
    st.t_loop_s = t2 - t1;
    st.t_total_s = st.t_init_s + (wall_s() - t0);
    st.iterations = ctl_.k + 1;
    st.exact_iters = ctl_.exact_iters;
    st.rho_updates = ctl_.rho_updates;
    st.rho_final = ctl_.rho;
    collect_timer();
    if (p.verbose > 0 && ctx_.dist.rank() == 0) {
      print_summary(status, st.t_total_s, st.t_init_s, ctl_);
      if (p.verbose > 3) print_timing_breakdown(st.t_loop_s, st.iterations);
      std::printf("POGS-AMD sparse/cgls: status %d, iter %u, init %.3e s, loop %.3e s, cg %llu, spmv %llu\n", status,
                  ctl_.k, st.t_init_s, st.t_loop_s, st.cg_iters, st.matvecs);
    }
    return status;
  }

  void begin_run(const FnHost &f, const FnHost &g, const SolveParams &p) override {
    load_problem(f, g, p);
    cold_start();
    apply_warm_start();
    ctx_.sync();
  }

  void set_warm_start(const void *x0, const void *l0) override {
    warm_x_.assign(static_cast<const T *>(x0), static_cast<const T *>(x0) + n_);
    warm_l_.assign(static_cast<const T *>(l0), static_cast<const T *>(l0) + m_);
    warm_pending_ = true;
  }

  void iterate(unsigned iters, double *seconds, unsigned *solves) override {
    POGS_CHECK(loaded_, "PogsAmdIterate before PogsAmdBeginRun / PogsAmdSolve: no problem is loaded");
    unsigned done = 0;
    ctx_.sync();
    const double t0 = wall_s();
    for (unsigned i = 0; i < iters; ++i) {
      if (ctl_.finished) {
        cold_start();
        ++done;
      }
      iteration(0);
    }
    ctx_.sync();
    const double t1 = wall_s();
    if (seconds) *seconds = t1 - t0;
    if (solves) *solves = done;
    ctx_.stats.t_loop_s += t1 - t0;
    ctx_.stats.iterations += iters;
    collect_timer();
  }

  void get_equil(void *A_eq, void *d, void *e, double *nrmA) override {
    ctx_.sync();
    // A_eq: the equilibrated CSR values of the first copy, length nnz
    if (A_eq) POGS_HIP_CHECK(hipMemcpy(A_eq, A_.val.p, nnz_ * sizeof(T), hipMemcpyDeviceToHost));
    if (d) POGS_HIP_CHECK(hipMemcpy(d, d_.p, m_ * sizeof(T), hipMemcpyDeviceToHost));
    if (e) POGS_HIP_CHECK(hipMemcpy(e, e_.p, n_ * sizeof(T), hipMemcpyDeviceToHost));
    if (nrmA) *nrmA = nrmA_;
  }

  void project(const void *x0, const void *y0, double tol, void *x, void *y) override {
    hipStream_t s = ctx_.stream;
    POGS_HIP_CHECK(hipMemcpyAsync(xtemp_.p, x0, n_ * sizeof(T), hipMemcpyHostToDevice, s));
    POGS_HIP_CHECK(hipMemcpyAsync(ytemp_.p, y0, m_ * sizeof(T), hipMemcpyHostToDevice, s));
    x_[1].zero(s);  // cold start: x = 0
    cgls_project(xtemp_.p, ytemp_.p, x_[1].p, static_cast<T>(tol));
    spmv<false>(A_, x_[1].p, nullptr, SpAxpbyOp<T>{1, 0, nullptr, y_[1].p}, nullptr, 0);
    POGS_HIP_CHECK(hipMemcpyAsync(x, x_[1].p, n_ * sizeof(T), hipMemcpyDeviceToHost, s));
    POGS_HIP_CHECK(hipMemcpyAsync(y, y_[1].p, m_ * sizeof(T), hipMemcpyDeviceToHost, s));
    ctx_.sync();
  }

  void mul(char trans, double alpha, const void *x, double beta, void *y) override {
    hipStream_t s = ctx_.stream;
    const bool tr = (trans == 't' || trans == 'T');
    const int nin = tr ? m_ : n_, nout = tr ? n_ : m_;
    DevBuf<T> vin(nin), vout(nout);
    POGS_HIP_CHECK(hipMemcpyAsync(vin.p, x, nin * sizeof(T), hipMemcpyHostToDevice, s));
    POGS_HIP_CHECK(hipMemcpyAsync(vout.p, y, nout * sizeof(T), hipMemcpyHostToDevice, s));
    const SpAxpbyOp<T> op{static_cast<T>(alpha), static_cast<T>(beta), vout.p, vout.p};
    if (tr) spmv_t<false>(vin.p, op, nullptr);   // summed over the row shards
    else spmv<false>(A_, vin.p, nullptr, op, nullptr, 0);
    POGS_HIP_CHECK(hipMemcpyAsync(y, vout.p, nout * sizeof(T), hipMemcpyDeviceToHost, s));
    ctx_.sync();
  }

 private:
  // ---- structure -----------------------------------------------------------
  void build_structure(int ord, const void *data, const int *ptr, const int *ind, int mem) {
    hipStream_t s = ctx_.stream;
    // "first" copy = what the caller gave (CSR if ROW_MAJ, CSC = CSR of A^T otherwise)
    const int r1 = (ord == ROW_MAJ) ? m_ : n_, c1 = (ord == ROW_MAJ) ? n_ : m_;
    DevCsr<T> first, second;
    first.nrows = r1;
    second.nrows = c1;
    first.nnz = second.nnz = nnz_;
    const hipMemcpyKind kind = (mem == POGS_AMD_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    first.val.alloc(nnz_); first.ind.alloc(nnz_); first.ptr.alloc(r1 + 1);
    POGS_HIP_CHECK(hipMemcpyAsync(first.val.p, data, nnz_ * sizeof(T), kind, s));
    POGS_HIP_CHECK(hipMemcpyAsync(first.ind.p, ind, nnz_ * sizeof(int), kind, s));
    POGS_HIP_CHECK(hipMemcpyAsync(first.ptr.p, ptr, (r1 + 1) * sizeof(int), kind, s));
    std::vector<int> hptr(r1 + 1);
    if (mem == POGS_AMD_DEVICE) {
      POGS_HIP_CHECK(hipMemcpyAsync(hptr.data(), ptr, (r1 + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
      ctx_.sync();
    } else {
      std::memcpy(hptr.data(), ptr, (r1 + 1) * sizeof(int));
    }
    POGS_CHECK(hptr[0] == 0 && static_cast<size_t>(hptr[r1]) == nnz_, "ptr does not match nnz");
    {
      DevBuf<int> err(1);
      err.zero(s);
      hipLaunchKernelGGL(validate_csr_kernel, dim3(2048), dim3(256), 0, s, first.ind.p, first.ptr.p, r1, c1, nnz_, err.p);
      int herr = 0;
      POGS_HIP_CHECK(hipMemcpyAsync(&herr, err.p, sizeof(int), hipMemcpyDeviceToHost, s));
      ctx_.sync();
      POGS_CHECK((herr & 1) == 0, "sparse matrix: ptr is not non-decreasing");
      POGS_CHECK((herr & 2) == 0, "sparse matrix: an index lies outside [0, columns)");
    }
    // transpose on the device (gsl_spmat.h:32-55)
    second.val.alloc(nnz_); second.ind.alloc(nnz_); second.ptr.alloc(c1 + 1);
    DevBuf<int> cnt(c1 + 1), cursor(c1 + 1);
    cnt.zero(s);
    if (nnz_) hipLaunchKernelGGL(count_cols_kernel, dim3(2048), dim3(256), 0, s, first.ind.p, nnz_, cnt.p);
    exclusive_scan(cnt.p, c1, second.ptr.p);
    POGS_HIP_CHECK(hipMemcpyAsync(cursor.p, second.ptr.p, (c1 + 1) * sizeof(int), hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(fill_transpose_kernel<T>, dim3(2048), dim3(256), 0, s, first.val.p, first.ind.p, first.ptr.p,
                       r1, cursor.p, second.val.p, second.ind.p);
    hipLaunchKernelGGL(sort_segments_kernel<T>, dim3(std::min(c1, 65536)), dim3(256), 0, s, second.ptr.p, c1,
                       second.ind.p, second.val.p);
    first.ncols = c1;
    second.ncols = r1;
    const char *ev = std::getenv("POGS_AMD_SPMV");
    if (!(ev && ev[0] == 'p')) {   // POGS_AMD_SPMV=plain keeps the plain CSR kernel (testing aid)
      build_sell(first);
      build_sell(second);
    }
    // row blocks of the plain CSR kernel: only for a copy that did not get its tiled form (the host
    // walk over every row and the copy of the transposed ptr array cost ~5 ms at C4)
    auto set_blocks = [&](DevCsr<T> &M, const std::vector<int> &hp) {
      std::vector<int> b = make_row_blocks(hp, M.nrows);
      M.nblocks = static_cast<int>(b.size()) - 1;
      M.blocks.alloc(b.size());
      POGS_HIP_CHECK(hipMemcpy(M.blocks.p, b.data(), b.size() * sizeof(int), hipMemcpyHostToDevice));
    };
    if (!first.sell_ready) set_blocks(first, hptr);
    if (!second.sell_ready) {
      std::vector<int> hptr2(c1 + 1);
      POGS_HIP_CHECK(hipMemcpyAsync(hptr2.data(), second.ptr.p, (c1 + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
      ctx_.sync();
      set_blocks(second, hptr2);
    }
    if (ord == ROW_MAJ) { A_ = std::move(first); At_ = std::move(second); }
    else { At_ = std::move(first); A_ = std::move(second); }
    first_is_A_ = (ord == ROW_MAJ);
  }

  // ptr[0..n] = exclusive scan of cnt[0..n)
  void exclusive_scan(const int *cnt, int n, int *ptr) {
    hipStream_t s = ctx_.stream;
    if (n <= 4 * kScanTile) {
      hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, cnt, n, ptr);
      return;
    }
    const int ntiles = (n + kScanTile - 1) / kScanTile;
    DevBuf<int> tot(ntiles), off(ntiles + 1);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(ntiles), dim3(1024), 0, s, cnt, n, ptr, tot.p);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, tot.p, ntiles, off.p);
    hipLaunchKernelGGL(scan_add_kernel, dim3(ntiles), dim3(1024), 0, s, ptr, n, off.p, ntiles);
    ctx_.sync();   // temporaries are freed at scope exit
  }

  // Tiled lane-stream copy of M (sell.h): structure and values now, values again after the
  // equilibration has rescaled the CSR copy (refill_sell).  Skipped (the plain CSR kernel then
  // runs) when the bookkeeping could not be indexed with 32 bits or the padding would blow up.
  void build_sell(DevCsr<T> &M) {
    hipStream_t s = ctx_.stream;
    constexpr int BW = SellCfg<T>::BW, RRMAX = SellCfg<T>::RR;
    if (M.nnz == 0) return;
    const int ncb = (M.ncols + BW - 1) / BW;
    // rows per row range: as many as the LDS holds, fewer when the matrix would otherwise give
    // the chip less than ~2 workgroups per CU (column groups can only multiply by ncb)
    const long long want = static_cast<long long>(M.nrows) * ncb / (2LL * ctx_.num_cu);
    int rr_rows = static_cast<int>(round_up(static_cast<size_t>(std::max<long long>(512, std::min<long long>(RRMAX, want))), 64));
    rr_rows = std::min(rr_rows, RRMAX);
    // column groups: the count that fills whole rounds of workgroups (one per CU) best, with the
    // column blocks split evenly; ties go to fewer groups (fewer partial sums).  How well the launch
    // fills its rounds decides the SpMV time beyond its bytes -- C4, BW x RR -> workgroups -> SpMV:
    // 18432 x 16384 -> 246 (A) / 248 (A^T), one round each -> 152 us; 24576 x 12288 -> 489, two
    // rounds -> 165 us; 22528 x 14336 -> 420, 0.82 of two rounds -> 225 us.  (A joint search over
    // the row-range height and the group count by this fill model alone picked many small groups
    // -- 17 rounds of 28 groups -- and was slower, 250 us: partial sums and per-tile costs are not
    // in the model.  Left at the LDS-limit height.)
    int ncg = 1;
    double best = -1;
    for (int g = 1; g <= std::min(ncb, 32); ++g) {
      const long long nwg = static_cast<long long>((M.nrows + rr_rows - 1) / rr_rows) * g;
      const long long rounds = (nwg + ctx_.num_cu - 1) / ctx_.num_cu;
      const double fill = static_cast<double>(nwg) / static_cast<double>(rounds * ctx_.num_cu);
      const double even = (static_cast<double>(ncb) / g) / static_cast<double>((ncb + g - 1) / g);
      const double eff = fill * even;
      if (eff > best + 1e-9) { best = eff; ncg = g; }
    }
    // Second look with a byte model of ONE workgroup's path (the launch takes rounds x that):
    //   matrix bytes rr * blocks * (nnz per row and block) * 8  +  x slices blocks * BW * s * 0.15 (they
    //   mostly hit L2)  +  partial sums rr * 8 (written, then read by reduce_parts) when there are groups.
    // Candidates are built to fill k rounds of ~250 workgroups exactly: for g groups, nrr = k * 250 / g
    // row ranges of rows / nrr rows each (shorter than the LDS limit).  Calibrated on C4 (forced
    // configurations (round 2, forced through tuning switches since removed): 8128 rows x 1 group +1.9 %, 12288 x 3 +9 %, A^T 8064 x 4
    // +4 %, 16384 x 16 +3 % against the 16384 x 2 / x 8 the rule above picks there); a candidate replaces
    // that choice only when the model sees more than 5 % in it -- matrices whose row count leaves the
    // LDS-limit height with many groups (1.4e6 rows: 14 groups, 1204 workgroups; 4157 GB/s).
    {
      const double d = static_cast<double>(M.nnz) / static_cast<double>(M.nrows) / ncb;
      auto path_bytes = [&](int rr, int g) {
        const long long nwg = static_cast<long long>((M.nrows + rr - 1) / rr) * g;
        const long long rounds = (nwg + ctx_.num_cu - 1) / ctx_.num_cu;
        const double cbg = static_cast<double>((ncb + g - 1) / g);
        return static_cast<double>(rounds) * (rr * cbg * d * 8.0 + cbg * BW * sizeof(T) * 0.15 + (g > 1 ? rr * 8.0 : 0.0));
      };
      const double base = path_bytes(rr_rows, ncg);
      double best_c = base * 0.95;
      const int rr_hi = rr_rows, cap = std::max(1, ctx_.num_cu - 6);
      for (int g = 1; g <= std::min(ncb, 32); ++g)
        for (int k = 1; k <= 8; ++k) {
          const long long nrr_t = static_cast<long long>(k) * cap / g;
          if (nrr_t < 1) continue;
          const int rr = std::max(512, static_cast<int>(round_up(static_cast<size_t>((M.nrows + nrr_t - 1) / nrr_t), 64)));
          if (rr > rr_hi) continue;
          const double c = path_bytes(rr, g);
          if (c < best_c * (1 - 1e-3)) { best_c = c; rr_rows = rr; ncg = g; }
        }
    }
    const int nrr = (M.nrows + rr_rows - 1) / rr_rows;
    const long long ntiles = static_cast<long long>(nrr) * ncb;
    const long long nq = ntiles * rr_rows;
    if (ntiles >= (1LL << 30) || nq >= (1LL << 31)) return;
    // storage format: the planner lays the tile out both ways and the smaller matrix is kept (7 bytes per stored
    // fp32 element with two id slots per batch, 8 with a tag per element -- but the first needs padding when most
    // rows of a tile hold a single element).  POGS_AMD_SELL_FORMAT=tags / two pins it (tests, A/B measurements).
    static_assert(SellCfg<T>::BW <= 32768, "bit 15 of a local column is the row-end flag of the two-slot format");
    int want_two = -1;
    if (const char *f = std::getenv("POGS_AMD_SELL_FORMAT")) want_two = std::strcmp(f, "two") == 0 ? 1 : (std::strcmp(f, "tags") == 0 ? 0 : -1);
    {
      // The plan keeps 6 bytes per (row, column block) pair (count, stream offset) and 4 more (the second layout's
      // offsets) unless the tag format is pinned -- on a matrix with many column blocks and few non-zeros per row
      // that outweighs the matrix itself (5e6 x 5e6: 272 blocks x 5e6 rows x 10 B = 13.6 GB).  Beyond 4x the CSR
      // bytes, or half of what the device has free, the plain CSR kernel stays (the same exit as a padding blow-up).
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
      const double tmp_bytes = (want_two != 0 ? 10.0 : 6.0) * static_cast<double>(nq);
      const double csr_bytes = static_cast<double>(M.nnz) * (sizeof(T) + 4.0);
      if (tmp_bytes > 4.0 * csr_bytes + 64e6 || (free_b && tmp_bytes > 0.5 * static_cast<double>(free_b))) return;
    }
    M.rr_rows = rr_rows; M.nrr = nrr; M.ncb = ncb; M.ncg = ncg;
    const SellDims D = M.sdims();
    DevBuf<unsigned> soff2;
    M.scnt.alloc(nq); M.ssoff.alloc(nq);
    if (want_two != 0) soff2.alloc(nq);
    M.scnt.zero(s);
    const int g = std::max(1, std::min((M.nrows + 3) / 4, ctx_.num_cu * 32));   // a wavefront per row, four per workgroup
    hipLaunchKernelGGL(sell_count_kernel, dim3(g), dim3(256), 0, s, M.ind.p, M.ptr.p, D, M.scnt.p);
    DevBuf<int> nu(ntiles + 1), nu2, err(1);
    DevBuf<int> tile_unit2;
    if (soff2.p) { nu2.alloc(ntiles + 1); tile_unit2.alloc(ntiles + 1); }
    err.zero(s);
    M.tile_unit.alloc(ntiles + 1);
    const int gt = static_cast<int>(std::min<long long>(ntiles, ctx_.num_cu * 8));
    {
      static SmemGrants grants;   // (row ranges taller than 23 K rows: static + dynamic LDS of the planner pass 64 KB)
      ensure_dynamic_smem(reinterpret_cast<const void *>(&sell_plan_kernel), sell_plan_lds(rr_rows) + 32768, grants);
    }
    hipLaunchKernelGGL(sell_plan_kernel, dim3(gt), dim3(256), sell_plan_lds(rr_rows), s, M.scnt.p, D, nu.p,
                       M.ssoff.p, nu2.p, soff2.p, err.p);
    exclusive_scan(nu.p, static_cast<int>(ntiles), M.tile_unit.p);
    int tot = 0, tot2 = 0, herr = 0;
    POGS_HIP_CHECK(hipMemcpyAsync(&tot, M.tile_unit.p + ntiles, sizeof(int), hipMemcpyDeviceToHost, s));
    if (soff2.p) {
      exclusive_scan(nu2.p, static_cast<int>(ntiles), tile_unit2.p);
      POGS_HIP_CHECK(hipMemcpyAsync(&tot2, tile_unit2.p + ntiles, sizeof(int), hipMemcpyDeviceToHost, s));
    }
    POGS_HIP_CHECK(hipMemcpyAsync(&herr, err.p, sizeof(int), hipMemcpyDeviceToHost, s));
    ctx_.sync();
    M.two = 0;
    if (soff2.p && !(herr & 8) && tot2 > 0) {
      // bytes per stored element: value + local column + (2 ids per 4 | a tag)
      const double b2 = static_cast<double>(tot2) * (sizeof(T) + 3.0), b1 = static_cast<double>(tot) * (sizeof(T) + 4.0);
      if (want_two == 1 || b2 < b1) {
        M.two = 1;
        tot = tot2;
        M.ssoff = std::move(soff2);
        M.tile_unit = std::move(tile_unit2);
      }
    }
    // each layout has its own range check (bit 4: the tag layout's 23-bit stream offsets, bit 8: the two-slot
    // layout's 22-bit ones): only the chosen layout's decides whether the tiled copy is usable
    herr &= M.two ? ~4 : ~8;
    if (std::getenv("POGS_AMD_TRACE"))
      std::fprintf(stderr, "[pogs_amd trace] tiled copy %d x %d: %s, %.3f stored elements per non-zero\n", M.nrows, M.ncols,
                   M.two ? "two id slots per batch" : "a row tag per element",
                   static_cast<double>(tot) * 64.0 / static_cast<double>(M.nnz));
    // (a padding blow-up beyond 4x the non-zeros -- a few very long rows among many short ones in
    // a tile -- is left to the plain kernel)
    if (herr != 0 || tot <= 0 || static_cast<size_t>(tot) * 64 > 4 * M.nnz + (static_cast<size_t>(1) << 22)) {
      M.scnt.release(); M.ssoff.release(); M.tile_unit.release();
      return;
    }
    M.sell_elems = static_cast<size_t>(tot) * 64;
    M.sval.alloc(M.sell_elems);
    M.sloc.alloc(M.sell_elems);
    M.srid.alloc(M.two ? M.sell_elems / 2 : M.sell_elems);
    M.sval.zero(s);
    M.sloc.zero(s);
    // (tags: kSellNoRow everywhere but on row ends; two id slots: an unused slot names row 0 -- it is looked up, never written)
    POGS_HIP_CHECK(hipMemsetAsync(M.srid.p, M.two ? 0x00 : 0xFF, M.srid.n * sizeof(unsigned short), s));
    M.sell_ready = true;
    fill_sell(M, true);
    if (ncg > 1) M.part.alloc(static_cast<size_t>(ncg) * M.nrows);
    ctx_.sync();   // nu / err are freed at scope exit
  }

  // (re)writes the tiled values from M.val; with_loc also the local columns and the row tags
  void fill_sell(DevCsr<T> &M, bool with_loc) {
    if (!M.sell_ready) return;
    hipStream_t s = ctx_.stream;
    // the first fill records where every CSR element went (4 B per non-zero until refill_sell): the
    // values are written once more after equilibration, and walking the (row, tile) bookkeeping a
    // second time costs 6.7 ms per copy at C4 against 1 ms for a gather through that table
    if (with_loc && M.sell_elems < (static_cast<size_t>(1) << 32)) M.sdst.alloc(M.nnz);
    const int g = std::max(1, std::min((M.nrows + 3) / 4, ctx_.num_cu * 32));   // a wavefront per row, four per workgroup
    hipLaunchKernelGGL(sell_fill_kernel<T>, dim3(g), dim3(256), 0, s, M.val.p, M.ind.p, M.ptr.p, M.sdims(), M.scnt.p,
                       M.ssoff.p, M.tile_unit.p, M.sval.p, with_loc ? M.sloc.p : nullptr, M.srid.p,
                       with_loc ? M.sdst.p : nullptr, M.two);
    ctx_.sync();
  }
  // the values are final (equilibrated): refill and drop the build temporaries
  void refill_sell(DevCsr<T> &M) {
    if (M.sell_ready && M.sdst.p) {
      const int g = static_cast<int>(std::min<size_t>((M.nnz + 255) / 256, static_cast<size_t>(ctx_.num_cu) * 32));
      hipLaunchKernelGGL(sell_refill_kernel<T>, dim3(std::max(1, g)), dim3(256), 0, ctx_.stream, M.val.p, M.sdst.p, M.nnz,
                         M.sval.p);
      ctx_.sync();
    } else {
      fill_sell(M, false);
    }
    M.sdst.release();
    M.scnt.release();
    M.ssoff.release();
  }

  // Per-XCD streaming rates from time-stamped launches of M's SpMV (workgroup b: work units / duration,
  // summed per XCC id); `reps` launches after one untimed.  Debug / calibration aid.
  void measure_xcd_rates(DevCsr<T> &M, const T *xin, T *yout, int reps, double *rate, bool print) {
    hipStream_t s = ctx_.stream;
    const int nwg = M.nrr * M.ncg;
    M.stamps.alloc(static_cast<size_t>(nwg) * 4);
    std::vector<unsigned long long> h(static_cast<size_t>(nwg) * 4);
    std::vector<double> work(kNumXcd, 0.0), time(kNumXcd, 0.0), tmax(kNumXcd, 0.0);
    std::vector<int> cnt(kNumXcd, 0);
    double kernel_us = 0;
    for (int r = 0; r <= reps; ++r) {
      M.stamps_on = true;
      spmv<false>(M, xin, nullptr, SpAxpbyOp<T>{1, 0, nullptr, yout}, nullptr, 0);
      M.stamps_on = false;
      POGS_HIP_CHECK(hipMemcpyAsync(h.data(), M.stamps.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
      ctx_.sync();
      if (r == 0) continue;
      unsigned long long lo = ~0ull, hi = 0;
      for (int b = 0; b < nwg; ++b) {
        const unsigned long long t0 = h[4 * b], t1 = h[4 * b + 1];
        const int x = static_cast<int>(h[4 * b + 2]) & (kNumXcd - 1);
        const double us = static_cast<double>(t1 - t0) / 100.0;   // wall_clock64: 100 MHz
        work[x] += static_cast<double>(h[4 * b + 3]);
        time[x] += us;
        tmax[x] = std::max(tmax[x], us);
        cnt[x]++;
        lo = std::min(lo, t0);
        hi = std::max(hi, t1);
      }
      kernel_us += static_cast<double>(hi - lo) / 100.0;
    }
    for (int x = 0; x < kNumXcd; ++x) rate[x] = time[x] > 0 ? work[x] / time[x] : 1.0;
    if (print) {
      std::fprintf(stderr, "[pogs_amd stamps] %d x %d, %d workgroups, first start to last end %.1f us; per XCD mean us (max) [rate]:",
                   M.nrows, M.ncols, nwg, kernel_us / reps);
      double rs = 0;
      for (int x = 0; x < kNumXcd; ++x) rs += rate[x];
      for (int x = 0; x < kNumXcd; ++x)
        std::fprintf(stderr, " %.1f (%.1f) [%.3f]", cnt[x] ? time[x] / cnt[x] : 0.0, tmax[x], rate[x] * kNumXcd / rs);
      std::fprintf(stderr, "\n");
    }
  }

  // POGS_AMD_SELL_STAMPS=1 (diagnostic): per-XCD times of both SpMVs on stderr once per handle (sell.h: what they showed).
  void print_stamps() {
    const char *st = std::getenv("POGS_AMD_SELL_STAMPS");
    if (!(st && st[0] == '1')) return;
    DevBuf<T> vin(static_cast<size_t>(std::max(m_, n_))), vout(static_cast<size_t>(std::max(m_, n_)));
    launch_fill<T>(vin.p, static_cast<T>(1), vin.n, ctx_.stream);
    for (DevCsr<T> *M : {&A_, &At_}) {
      if (!M->sell_ready) continue;
      double r[kNumXcd];
      measure_xcd_rates(*M, vin.p, vout.p, 3, r, true);
    }
    ctx_.sync();
  }

  void alloc_state() {
    hipStream_t s = ctx_.stream;
    for (int i = 0; i < 2; ++i) { x_[i].alloc(n_); y_[i].alloc(m_); x_[i].zero(s); y_[i].zero(s); }
    xt_.alloc(n_); yt_.alloc(m_); xtemp_.alloc(n_); ytemp_.alloc(m_); x12_.alloc(n_); y12_.alloc(m_);
    xt_.zero(s); yt_.zero(s); xtemp_.zero(s); ytemp_.zero(s); x12_.zero(s); y12_.zero(s);
    d_.alloc(m_); e_.alloc(n_);
    cg_p_.alloc(n_); cg_s_.alloc(n_); cg_q_.alloc(m_); cg_r_.alloc(m_); cg_b_.alloc(m_); u_.alloc(m_);
    xout_.alloc(n_); yout_.alloc(m_); lout_.alloc(m_); muout_.alloc(n_);
    f_.alloc(m_); g_.alloc(n_); fs_.alloc(m_); gs_.alloc(n_);
    cg_.alloc(kCgNumSlots);
    cg_.zero(s);
    if (multi_) { tsum_.alloc(n_); cg_u_.alloc(n_); cg_u_.zero(s); tsum_.zero(s); }
    spmv_grid_ = ctx_.num_cu * 8;
    const size_t vb = vec_blocks(n_) + vec_blocks(m_);
    size_t sg = static_cast<size_t>(spmv_grid_);   // workgroups that write scalar partials in one launch
    if (A_.sell_ready) sg = std::max(sg, static_cast<size_t>(A_.nrr) * A_.ncg);
    if (At_.sell_ready) sg = std::max(sg, static_cast<size_t>(At_.nrr) * At_.ncg);
    // [SpMV / vector-kernel partials | |x|^2 partials of a CG step | |p|^2 partials]: the last two
    // are summed by the launch that publishes the scalars, so they keep regions of their own
    sp_cgx_off_ = std::max<size_t>(sg * 4 + 64, vb * 3 + 64);
    const size_t cgreg = static_cast<size_t>(std::max(vec_blocks(n_), kCgfBlocks)) + 8;   // (cg_fused.h: up to kCgfBlocks records)
    sp_cgp_off_ = sp_cgx_off_ + cgreg;
    sp_pre_off_ = sp_cgp_off_ + cgreg;   // prox-step sums (deferred on one GPU)
    ctx_.ensure_spart(sp_pre_off_ + vb * 3 + 8);
    // device-resident CGLS loop (cg_fused.h): both copies in the tiled layout; POGS_AMD_CG=h keeps
    // round 2's host-polled loop (cgls_project), which is also what the plain-CSR fallback runs
    const char *cg_env = std::getenv("POGS_AMD_CG");
    fused_cg_ = A_.sell_ready && At_.sell_ready && !(cg_env && cg_env[0] == 'h') && ctx_.poll_fetch;
    if (multi_) {
      // every rank must take the same path (the collectives of the two loops differ): all or none
      DevBuf<double> flag(1);
      const double mine = fused_cg_ ? 0.0 : 1.0;
      POGS_HIP_CHECK(hipMemcpyAsync(flag.p, &mine, sizeof(double), hipMemcpyHostToDevice, s));
      ctx_.dist.allreduce(flag.p, 1, s);
      double any = 0;
      POGS_HIP_CHECK(hipMemcpyAsync(&any, flag.p, sizeof(double), hipMemcpyDeviceToHost, s));
      POGS_HIP_CHECK(hipStreamSynchronize(s));
      if (any != 0.0) fused_cg_ = false;
    }
    if (fused_cg_) {
      // scalar records of the loop's products: one region for A^T products, one for A products
      cg_rec_cap_ = static_cast<size_t>(std::max({A_.nrr * A_.ncg, At_.nrr * At_.ncg, kCgfBlocks})) * 2;
      if (multi_) {
        // row shards: the |q|^2 records of the ranks are summed record by record (one all-reduce of the
        // record array, no folding launch), so the ranks agree on its length -- the largest -- and a
        // rank's unused tail stays zero; the sums land in a third region
        DevBuf<double> caps(static_cast<size_t>(ctx_.dist.world()));
        caps.zero(s);
        const double mine = static_cast<double>(cg_rec_cap_);
        POGS_HIP_CHECK(hipMemcpyAsync(caps.p + ctx_.dist.rank(), &mine, sizeof(double), hipMemcpyHostToDevice, s));
        ctx_.dist.allreduce(caps.p, caps.n, s);
        std::vector<double> all(caps.n);
        POGS_HIP_CHECK(hipMemcpyAsync(all.data(), caps.p, caps.n * sizeof(double), hipMemcpyDeviceToHost, s));
        POGS_HIP_CHECK(hipStreamSynchronize(s));
        for (double v : all) cg_rec_cap_ = std::max(cg_rec_cap_, static_cast<size_t>(v));
      }
      cg_rec_.alloc(cg_rec_cap_ * (multi_ ? 4 : 2));   // row shards: + the summed |q|^2 records, + |s|^2 records of U1
      cg_rec_.zero(s);
      const char *ys = std::getenv("POGS_AMD_YSYNC");
      if (ys) ysync_ = std::max(0, std::atoi(ys));
    }
  }

  // y_i = op(sum_k val * x[ind]) over the rows of M; scalar sums land in S[slot..slot+NS)
  // cg_mode != 0 (single GPU): the scalar sum and the CGLS scalar that consumes it run as one launch
  template <bool SQ, typename Op>
  void spmv(const DevCsr<T> &M, const T *x, const double *x_nrm2, const Op &op, double *scalar_out, int,
            bool timed = false, int cg_mode = 0) {
    hipStream_t s = ctx_.stream;
    int grid;
    if (timed) ctx_.stream_timer.begin(s);
    if (M.sell_ready) {
      constexpr size_t smem = sell_lds_bytes<T>();
      const int g1 = M.nrr * M.ncg;
      if (M.ncg == 1) {
        static SmemGrants grants;
        ensure_dynamic_smem(reinterpret_cast<const void *>(&spmv_sell_kernel<T, SQ, true, Op>), smem, grants);
        hipLaunchKernelGGL((spmv_sell_kernel<T, SQ, true, Op>), dim3(g1), dim3(kSellTpb), smem, s, M.sview(), x,
                           x_nrm2, op, static_cast<T *>(nullptr), ctx_.spart.p, static_cast<const double *>(nullptr));
        grid = g1;
      } else {
        static SmemGrants grants;
        ensure_dynamic_smem(reinterpret_cast<const void *>(&spmv_sell_kernel<T, SQ, false, Op>), smem, grants);
        hipLaunchKernelGGL((spmv_sell_kernel<T, SQ, false, Op>), dim3(g1), dim3(kSellTpb), smem, s, M.sview(), x,
                           x_nrm2, op, M.part.p, ctx_.spart.p, static_cast<const double *>(nullptr));
        grid = std::max(1, std::min((M.nrows + 255) / 256, spmv_grid_));
        hipLaunchKernelGGL((reduce_parts_kernel<T, Op>), dim3(grid), dim3(256), 0, s, M.part.p, M.nrows, M.ncg, op,
                           ctx_.spart.p);
      }
    } else {
      grid = std::max(1, std::min(M.nblocks, spmv_grid_));
      hipLaunchKernelGGL((spmv_kernel<T, SQ, Op>), dim3(grid), dim3(kSpTpb), 0, s, M.view(), x, x_nrm2, op,
                         ctx_.spart.p);
    }
    if (timed) {
      ctx_.stream_timer.end(s);
      ++timed_spmvs_;
    }
    if (Op::NS > 0 && scalar_out) {
      SumJob j{ctx_.spart.p, grid, Op::NS, scalar_out};
      if (cg_mode != 0) launch_sum_cg(j, ctx_.S.p, cg_.p, cg_mode, 1.0, std::numeric_limits<T>::epsilon(), s);
      else launch_sum_jobs(&j, 1, s);
    }
  }
  // A^T product: with row shards the n partial sums are all-reduced before the row functor runs
  template <bool SQ, typename Op>
  void spmv_t(const T *xin, const Op &op, double *scalar_out, bool timed = false, int cg_mode = 0) {
    if (!multi_) {
      spmv<SQ>(At_, xin, nullptr, op, scalar_out, 0, timed, cg_mode);
      return;
    }
    hipStream_t s = ctx_.stream;
    spmv<SQ>(At_, xin, nullptr, SpAxpbyOp<T>{1, 0, nullptr, tsum_.p}, nullptr, 0, timed);
    ctx_.dist.allreduce(tsum_.p, n_, s);
    const int grid = std::max(1, std::min((n_ + 255) / 256, spmv_grid_));
    hipLaunchKernelGGL((apply_rows_kernel<T, Op>), dim3(grid), dim3(256), 0, s, tsum_.p, n_, op, ctx_.spart.p);
    if (Op::NS > 0 && scalar_out) {
      SumJob j{ctx_.spart.p, grid, Op::NS, scalar_out};
      launch_sum_jobs(&j, 1, s);
    }
  }
  // A product of the device-resident CG loop (cg_fused.h): the SpMV and, with more than one column
  // group, the group reduction that runs the row functor; both return at once unless the loop's
  // done flag says `run_if_done` (-1: always run).  The functor's scalar records (one per block)
  // go to `rec`; returns how many there are.  *ev: index of the stream-timer pair.
  template <typename Op>
  int spmv_cg(const DevCsr<T> &M, const T *x, const Op &op, double *rec, int run_if_done, size_t *ev) {
    hipStream_t s = ctx_.stream;
    constexpr size_t smem = sell_lds_bytes<T>();
    const double *S = ctx_.S.p;
    const double *guard = run_if_done == 0 ? S + kFcDone : nullptr;
    const int g1 = M.nrr * M.ncg;
    int nrec;
    *ev = ctx_.stream_timer.begin(s);
    if (M.ncg == 1) {
      static SmemGrants grants;
      ensure_dynamic_smem(reinterpret_cast<const void *>(&spmv_sell_kernel<T, false, true, Op>), smem, grants);
      hipLaunchKernelGGL((spmv_sell_kernel<T, false, true, Op>), dim3(g1), dim3(kSellTpb), smem, s, M.sview(), x,
                         static_cast<const double *>(nullptr), op, static_cast<T *>(nullptr), rec, guard);
      nrec = g1;
    } else {
      static SmemGrants grants;
      ensure_dynamic_smem(reinterpret_cast<const void *>(&spmv_sell_kernel<T, false, false, Op>), smem, grants);
      hipLaunchKernelGGL((spmv_sell_kernel<T, false, false, Op>), dim3(g1), dim3(kSellTpb), smem, s, M.sview(), x,
                         static_cast<const double *>(nullptr), op, M.part.p, rec, guard);
      nrec = cgf_blocks(M.nrows);
      hipLaunchKernelGGL((cgf_reduce_kernel<T, Op>), dim3(nrec), dim3(kCgfTpb), 0, s, M.part.p, M.nrows, M.ncg, op, rec, S,
                         run_if_done);
    }
    ctx_.stream_timer.end(s);
    return nrec;
  }
  // sums of a y-sized quantity: add the other ranks' rows
  void reduce_y_scalars(double *slot, int count) {
    if (multi_) ctx_.dist.allreduce(slot, count, ctx_.stream);
  }

  // MatrixSparse::Equil (matrix_sparse.cpp:158-242): Sinkhorn-Knopp on the squared
  // entries (squared on the fly), D A E on both copies, Frobenius norm of the first.
  void equilibrate() {
    hipStream_t s = ctx_.stream;
    PhaseTimer pt(s);
    const double mg = static_cast<double>(ctx_.m_global), nn = n_;
    const T ce = static_cast<T>(1e-4) * static_cast<T>(mg + nn) / static_cast<T>(mg);
    const T cd = static_cast<T>(1e-4) * static_cast<T>(mg + nn) / static_cast<T>(nn);
    launch_fill<T>(d_.p, static_cast<T>(1), m_, s);
    launch_fill<T>(e_.p, static_cast<T>(1), n_, s);
    // 50 iterations in the reference (equil_helper.h:147).  As in DenseSolver::equilibrate (fp32):
    // once an iteration changes every entry of e (replicated, so the ranks of a sharded solve
    // agree) by one common ratio 1 + gamma -- the slow drift of the common factor (d * a, e / a)
    // -- the loop ends after its d update and the remaining iterations are applied in closed
    // form.  POGS_AMD_SK_FULL=1: all 50.
    const char *sk_env = std::getenv("POGS_AMD_SK_FULL");
    const bool sk_probe = std::is_same<T, float>::value && !(sk_env && sk_env[0] == '1');
    double *mark = sk_probe ? ctx_.S.p + kSkMark : nullptr;
    const T sk_tol = 16 * std::numeric_limits<T>::epsilon();
    double r_ref = 0, gamma = 0;
    bool extrapolate = false;
    int k = 0;
    while (k < 50) {
      spmv_t<true>(d_.p, SpSkOp<T>{static_cast<T>(mg), ce, e_.p, mark, k + 1.0, sk_tol, static_cast<T>(r_ref)},
                   ctx_.S.p + kSkRatio);
      spmv<true>(A_, e_.p, nullptr, SpSkOp<T>{static_cast<T>(nn), cd, d_.p}, nullptr, 0);
      ++k;
      if (mark && k >= 2) {
        const double *S = ctx_.fetch_scalars();
        const double r_mean = S[kSkRatio] / n_;
        const bool uniform = k >= 3 && S[kSkMark] < static_cast<double>(k) && r_mean > 0.5 && r_mean < 2.0;
        r_ref = r_mean;
        gamma = r_mean - 1.0;
        if (uniform) { extrapolate = true; break; }
      }
    }
    if (extrapolate) {
      // state (e_{k-1}, d_k) after k iterations; the reference ends with (e_49, d_50)
      const double f = std::pow(1.0 + gamma, 50 - k);
      launch_scal<T>(e_.p, static_cast<T>(f), n_, s);
      launch_scal<T>(d_.p, static_cast<T>(1.0 / f), m_, s);
    }
    ctx_.stats.matvecs_init += 2 * k;
    launch_sqrt_inplace<T>(d_.p, m_, s);
    launch_sqrt_inplace<T>(e_.p, n_, s);
    const int g = ctx_.num_cu * 8;
    double *pa = ctx_.spart.p, *pb = ctx_.spart.p + g;
    hipLaunchKernelGGL(scale_csr_kernel<T>, dim3(g), dim3(256), 0, s, A_.val.p, A_.ind.p, A_.ptr.p, m_, d_.p, e_.p, pa);
    hipLaunchKernelGGL(scale_csr_kernel<T>, dim3(g), dim3(256), 0, s, At_.val.p, At_.ind.p, At_.ptr.p, n_, e_.p, d_.p,
                       pb);
    SumJob j{first_is_A_ ? pa : pb, g, 1, ctx_.S.p + kFro2};   // first nnz only (matrix_sparse.cpp:257)
    launch_sum_jobs(&j, 1, s);
    reduce_y_scalars(ctx_.S.p + kFro2, 1);
    const double *S = ctx_.fetch_scalars();
    const T normA = static_cast<T>(std::sqrt(S[kFro2])) /
                    static_cast<T>(std::sqrt(std::min(mg, nn)));
    launch_scal<T>(A_.val.p, static_cast<T>(1) / normA, nnz_, s);
    launch_scal<T>(At_.val.p, static_cast<T>(1) / normA, nnz_, s);
    refill_sell(A_);
    refill_sell(At_);
    const T invs = static_cast<T>(1) / std::sqrt(normA);
    launch_scal<T>(d_.p, invs, m_, s);
    launch_scal<T>(e_.p, invs, n_, s);
    ctx_.stats.equil_ms = pt.stop_ms();
  }

  // Norm2Est (equil_helper.h:107-135)
  void norm_est() {
    hipStream_t s = ctx_.stream;
    PhaseTimer pt(s);
    std::vector<T> x0(n_);
    rand_uniform_host(x0.data(), n_);
    POGS_HIP_CHECK(hipMemcpyAsync(xtemp_.p, x0.data(), n_ * sizeof(T), hipMemcpyHostToDevice, s));
    ctx_.sync();
    const T kTol = static_cast<T>(1e-4);
    T norm_est = 0, last;
    unsigned i = 0;
    for (i = 0; i < 50; ++i) {
      last = norm_est;
      // Sx = A (x / |x|);  x' = A^T Sx
      spmv<false>(A_, xtemp_.p, (i == 0) ? nullptr : ctx_.S.p + kPowX2,
                  SpAxpbyNormOp<T>{1, 0, nullptr, cg_q_.p}, ctx_.S.p + kPowSx2, 0);
      reduce_y_scalars(ctx_.S.p + kPowSx2, 1);
      spmv_t<false>(cg_q_.p, SpAxpbyNormOp<T>{1, 0, nullptr, xtemp_.p}, ctx_.S.p + kPowX2);
      const double *S = ctx_.fetch_scalars();
      const T normx = static_cast<T>(std::sqrt(S[kPowX2]));
      const T normSx = static_cast<T>(std::sqrt(S[kPowSx2]));
      norm_est = normx / normSx;
      ctx_.stats.matvecs_init += 2;
      if (std::abs(last - norm_est) < kTol * norm_est) { ++i; break; }
    }
    nrmA_ = norm_est;
    ctx_.stats.nrmA = nrmA_;
    ctx_.stats.norm_est_iters = i;
    xtemp_.zero(s);
    ctx_.stats.normest_ms = pt.stop_ms();
  }

  // ---- per solve -----------------------------------------------------------
  void load_problem(const FnHost &f, const FnHost &g, const SolveParams &p) {
    hipStream_t s = ctx_.stream;
    upload_fn<T>(f_, f, m_, s);
    upload_fn<T>(g_, g, n_, s);
    warn_negative_coeffs<T>(f, m_);   // prox_lib.h:62-69 (the clamp is in scale_objective_kernel)
    warn_negative_coeffs<T>(g, n_);
    pre_cheap_ = all_h(f, m_, [](int h) { return is_cheap_prox(h); }) && all_h(g, n_, [](int h) { return is_cheap_prox(h); });
    launch_scale_objective<T>(f_.view(), fs_.a.p, fs_.c.p, fs_.d.p, fs_.e.p, d_.p, m_, true, s);
    launch_scale_objective<T>(g_.view(), gs_.a.p, gs_.c.p, gs_.d.p, gs_.e.p, e_.p, n_, false, s);
    // coefficient arrays that hold one value throughout (a lasso: h, c, d, e of both halves) are not streamed by the
    // prox step: 16 of its 44 bytes per element
    uni_f_ = probe_uniform<T>(fview(), m_, s);
    uni_g_ = probe_uniform<T>(gview(), n_, s);
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
    ctx_.sync();
  }
  FnView<T> fview() const { return FnView<T>{f_.h.p, fs_.a.p, f_.b.p, fs_.c.p, fs_.d.p, fs_.e.p}; }
  FnView<T> gview() const { return FnView<T>{g_.h.p, gs_.a.p, g_.b.p, gs_.c.p, gs_.d.p, gs_.e.p}; }

  void cold_start() {
    hipStream_t s = ctx_.stream;
    for (int i = 0; i < 2; ++i) { x_[i].zero(s); y_[i].zero(s); }
    xt_.zero(s); yt_.zero(s); xtemp_.zero(s); ytemp_.zero(s);
    cur_ = 0;
    zt_scale_ = 1;
    ctl_.reset();
    proj_count_ = 0;
    cg_pred_ = 1;
  }

  void sum_vec_partials(int blocks, double *out) {
    SumJob j{ctx_.spart.p, blocks, 1, out};
    launch_sum_jobs(&j, 1, ctx_.stream);
  }

  // ProjectorCgls::Project up to (not including) the final y = A x
  // (projector_cgls.cpp:59-75): x holds the warm start on entry, the projected x
  // on exit.
  // `Ax_warm` (optional): A times the warm-start x, if the caller already has it.  The
  // reference forms b = y0 - A x0 and r = b - A (x - x0) with two SpMVs (:65-68,
  // cgls.h:226-233); their sum is r = y0 - A x_warm, and inside the ADMM loop A x_warm is the
  // previous iteration's y (the projection always ends with y = A x), so both SpMVs vanish.
  // `x_warm` (with Ax_warm): where the warm start is read from -- x itself by default; the ADMM
  // loop passes the previous iterate, which saves the 2 MB device-to-device copy into x per
  // iteration (30 us at C4 through the runtime's blit kernel).
  void cgls_project(const T *x0, const T *y0, T *x, T tol, const T *Ax_warm = nullptr, const T *x_warm = nullptr) {
    hipStream_t s = ctx_.stream;
    const int bx = vec_blocks(n_);
    const double shift = 1.0;
    const double kEps = std::numeric_limits<T>::epsilon();
    if (Ax_warm) {
      // r = y0 - A x_warm ; x <- x - x0                                       (:62)
      hipLaunchKernelGGL(sub_norm_kernel<T>, dim3(vec_blocks(m_)), dim3(kVecTpb), 0, s, m_, y0, Ax_warm, cg_r_.p,
                         ctx_.spart.p);
      hipLaunchKernelGGL(sub_norm_kernel<T>, dim3(bx), dim3(kVecTpb), 0, s, n_, x_warm ? x_warm : x, x0, x, ctx_.spart.p);
    } else {
      // x <- x - x0, |x|^2                                                    (:62)
      hipLaunchKernelGGL(sub_norm_kernel<T>, dim3(bx), dim3(kVecTpb), 0, s, n_, x, x0, x, ctx_.spart.p);
      sum_vec_partials(bx, ctx_.S.p + kCgX2);
      // b = y0 - A x0                                                         (:65-68)
      spmv<false>(A_, x0, nullptr, SpAxpbyOp<T>{static_cast<T>(-1), static_cast<T>(1), y0, cg_b_.p}, nullptr, 0,
                  true);
      const double *S0 = ctx_.fetch_scalars();
      // r = b - A x (only if x != 0)                                          (cgls.h:226-233)
      if (std::sqrt(S0[kCgX2]) > 0.0) {
        spmv<false>(A_, x, nullptr, SpAxpbyOp<T>{static_cast<T>(-1), static_cast<T>(1), cg_b_.p, cg_r_.p}, nullptr,
                    0, true);
      } else {
        POGS_HIP_CHECK(hipMemcpyAsync(cg_r_.p, cg_b_.p, m_ * sizeof(T), hipMemcpyDeviceToDevice, s));
      }
    }
    const double *S;
    double normx;
    // Single GPU: the scalar sum after an SpMV and the one-thread CGLS update that consumes it are
    // one launch (launch_sum_cg), and |x|^2, |p|^2 -- needed by the host only -- are summed by the
    // launch that publishes the scalar block: 9 launches per CG step instead of 13.
    const bool fuse = !multi_;
    double *px = ctx_.spart.p + sp_cgx_off_, *pp = ctx_.spart.p + sp_cgp_off_;
    // s = A^T r - shift x ; p = s ; gamma = |s|^2                             (cgls.h:236-245)
    spmv_t<false>(cg_r_.p, SpAxpbyNormOp<T>{1, static_cast<T>(-shift), x, cg_s_.p}, ctx_.S.p + kCgS2, true, fuse ? 3 : 0);
    if (!fuse) hipLaunchKernelGGL(set_gamma_kernel, dim3(1), dim3(1), 0, s, ctx_.S.p, cg_.p);
    hipLaunchKernelGGL(cg_update_p_kernel<T>, dim3(bx), dim3(kVecTpb), 0, s, n_, cg_.p, cg_s_.p, cg_p_.p,
                       fuse ? pp : ctx_.spart.p, true);
    if (fuse) ctx_.queue_sum(SumJob{pp, bx, 1, ctx_.S.p + kCgP2});
    else sum_vec_partials(bx, ctx_.S.p + kCgP2);
    S = ctx_.fetch_scalars();
    const double norms0 = std::sqrt(S[kCgS2]);
    double norms = norms0;
    const int maxit = (norms < kEps) ? 0 : 500;                               // flag 1 / projector_cgls.cpp:17
    for (int k = 0; k < maxit; ++k) {
      // q = A p, |q|^2 ; alpha                                               (cgls.h:257-271)
      spmv<false>(A_, cg_p_.p, nullptr, SpAxpbyNormOp<T>{1, 0, nullptr, cg_q_.p}, ctx_.S.p + kCgQ2, 0, true, fuse ? 1 : 0);
      if (!fuse) {
        reduce_y_scalars(ctx_.S.p + kCgQ2, 1);
        hipLaunchKernelGGL(cg_alpha_kernel, dim3(1), dim3(1), 0, s, ctx_.S.p, cg_.p, shift, kEps);
      }
      // x += alpha p ; r -= alpha q ; |x|^2                                  (:274-277)
      const int bm = vec_blocks(m_);
      hipLaunchKernelGGL(cg_update_xr_kernel<T>, dim3(bx + bm), dim3(kVecTpb), 0, s, n_, m_, cg_.p, cg_p_.p, x,
                         cg_q_.p, cg_r_.p, fuse ? px : ctx_.spart.p, bx, static_cast<const T *>(nullptr),
                         static_cast<T *>(nullptr));
      if (fuse) ctx_.queue_sum(SumJob{px, bx, 1, ctx_.S.p + kCgX2});
      else sum_vec_partials(bx, ctx_.S.p + kCgX2);
      // s = A^T r - shift x ; |s|^2 ; beta ; p = s + beta p ; |p|^2          (:281-296)
      spmv_t<false>(cg_r_.p, SpAxpbyNormOp<T>{1, static_cast<T>(-shift), x, cg_s_.p}, ctx_.S.p + kCgS2, true, fuse ? 2 : 0);
      if (!fuse) hipLaunchKernelGGL(cg_beta_kernel, dim3(1), dim3(1), 0, s, ctx_.S.p, cg_.p);
      hipLaunchKernelGGL(cg_update_p_kernel<T>, dim3(bx), dim3(kVecTpb), 0, s, n_, cg_.p, cg_s_.p, cg_p_.p,
                         fuse ? pp : ctx_.spart.p, false);
      if (fuse) ctx_.queue_sum(SumJob{pp, bx, 1, ctx_.S.p + kCgP2});
      else sum_vec_partials(bx, ctx_.S.p + kCgP2);
      S = ctx_.fetch_scalars();
      norms = std::sqrt(S[kCgS2]);
      normx = std::sqrt(S[kCgX2]);
      ++ctx_.stats.cg_iters;
      const bool converged = (norms <= norms0 * static_cast<double>(tol)) || (normx * static_cast<double>(tol) >= 1.0);
      if (converged) break;                                                   // :301-305
    }
    // x <- x + x0                                                            (projector_cgls.cpp:75)
    launch_axpby<T>(n_, static_cast<T>(1), x0, static_cast<T>(1), x, s);
  }

  // (x0, lambda0) -> (z, z~)   (pogs.cpp:144-156), see dense_solver.h
  void apply_warm_start() {
    if (!warm_pending_) return;
    warm_pending_ = false;
    hipStream_t s = ctx_.stream;
    const T rho = ctl_.rho;
    POGS_HIP_CHECK(hipMemcpyAsync(xtemp_.p, warm_x_.data(), n_ * sizeof(T), hipMemcpyHostToDevice, s));
    POGS_HIP_CHECK(hipMemcpyAsync(ytemp_.p, warm_l_.data(), m_ * sizeof(T), hipMemcpyHostToDevice, s));
    launch_scale_by<T>(n_, static_cast<T>(1), xtemp_.p, e_.p, true, x_[cur_].p, s);
    spmv<false>(A_, x_[cur_].p, nullptr, SpAxpbyOp<T>{1, 0, nullptr, y_[cur_].p}, nullptr, 0);
    launch_scale_by<T>(m_, static_cast<T>(1), ytemp_.p, d_.p, true, yt_.p, s);
    spmv_t<false>(yt_.p, SpAxpbyOp<T>{static_cast<T>(1) / rho, 0, nullptr, xt_.p}, nullptr);
    launch_scal<T>(yt_.p, static_cast<T>(-1) / rho, m_, s);
    ctx_.sync();
    xtemp_.zero(s);
    ytemp_.zero(s);
  }

  // The prox step and the projection of one iteration with the device-resident CGLS loop
  // (cg_fused.h): 6 launches per CG step, one host poll.  Returns the published block.
  const double *prox_and_project_fused(const AdmmPreArgs<T> &pa0, int nw) {
    hipStream_t s = ctx_.stream;
    const int bx = pre_blocks(n_), bm = pre_blocks(m_);
    double *S = ctx_.S.p;
    double *rec_t = cg_rec_.p, *rec_a = cg_rec_.p + cg_rec_cap_;          // A^T products, A products
    double *rec_x = ctx_.spart.p + sp_cgx_off_, *rec_p = ctx_.spart.p + sp_cgp_off_;
    double *cpart = ctx_.spart.p;   // closing launch: [bx + bm][2]
    const double shift = 1.0, kEps = std::numeric_limits<T>::epsilon();
    const double tol = static_cast<double>(ctl_.proj_tol());
    T *x = x_[nw].p;
    // y = A x by the recurrence, except every ysync_-th projection (and the first), which takes the
    // product itself
    const bool ysync = ysync_ <= 0 || (proj_count_ % static_cast<unsigned long long>(ysync_)) == 0;
    ++proj_count_;
    // prox, sums, over-relaxation; r = y0 - A x_warm (A x_warm is the previous y, see cgls_project),
    // x <- x_warm - x0                                                        (projector_cgls.cpp:62)
    AdmmPreArgs<T> pa = pa0;
    pa.x_aux = x;
    pa.y_aux = cg_r_.p;
    pa.cg_reset = S + kFcDone;
    launch_admm_pre<T>(pa, s);
    std::vector<size_t> &ev = fused_events_;
    ev.clear();
    size_t e;
    // Row shards: q = A p and r are this rank's rows; x, p, s and
    // u = A^T r (summed over the ranks) are replicated.  A CG step needs two sums over all ranks, |q|^2
    // and A^T r_new = u - alpha A^T q: with t = A^T q formed BEFORE alpha is known, both travel at the
    // same point -- t as the n-vector of local column sums, |q|^2 as the array of per-block records
    // (summed record by record; every block of U1 then adds the records up as on one GPU) -- in ONE
    // grouped RCCL launch per step, on the stream, between the launches that produce and consume
    // them: 0 host polls.  u is set by the explicit product at the start of every projection, so the
    // recurrence u -= alpha t runs over the few steps of one projection only.
    double *rec_a_sum = multi_ ? cg_rec_.p + 2 * cg_rec_cap_ : rec_a;
    double *rec_s2 = multi_ ? cg_rec_.p + 3 * cg_rec_cap_ : rec_t;
    const int nrec_a_sum = static_cast<int>(cg_rec_cap_);
    const int gv = cgf_blocks(std::max(n_, m_)), gp = cgf_blocks(n_);
    // s = A^T r - shift x ; p = s ; |s_0|^2 records                           (cgls.h:236-245)
    int nrec_s0;
    if (!multi_) {
      nrec_s0 = spmv_cg(At_, cg_r_.p, SpCgInitOp<T>{static_cast<T>(shift), x, cg_s_.p, cg_p_.p}, rec_t, -1, &e);
    } else {
      spmv_cg(At_, cg_r_.p, SpStoreOp<T>{tsum_.p}, rec_t, -1, &e);
      ctx_.dist.allreduce(tsum_.p, n_, s);
      nrec_s0 = cgf_blocks(n_);
      SpCgInitOp<T> init{static_cast<T>(shift), x, cg_s_.p, cg_p_.p};
      init.u = cg_u_.p;
      hipLaunchKernelGGL((cgf_reduce_kernel<T, SpCgInitOp<T>>), dim3(nrec_s0), dim3(kCgfTpb), 0, s, tsum_.p, n_, 1, init,
                         rec_t, S, -1);
    }
    int enq = 0;
    auto step = [&]() {
      // q = A p, |q|^2 records                                               (cgls.h:257-260)
      int nrec_q = spmv_cg(A_, cg_p_.p, SpAxpbyNormOp<T>{1, 0, nullptr, cg_q_.p}, rec_a, 0, &e);
      ev.push_back(e);
      if (multi_) {
        // t = A^T q (this rank's rows), then t and the |q|^2 records over all ranks
        spmv_cg(At_, cg_q_.p, SpStoreOp<T>{tsum_.p}, rec_s2, 0, &e);
        ev.push_back(e);
        {
          DistComm::Group grp(ctx_.dist);   // one RCCL launch; closed on every way out (also by an exception)
          ctx_.dist.allreduce(tsum_.p, n_, s);
          ctx_.dist.allreduce(rec_a, rec_a_sum, static_cast<size_t>(nrec_a_sum), s);
        }
        nrec_q = nrec_a_sum;
      }
      // alpha ; x += alpha p ; r -= alpha q ; y_new += alpha q ; |x|^2        (:262-277, 298)
      // (row shards: also u -= alpha t ; s = u - shift x ; |s|^2, :281-286)
      CgfStepA<T> a;
      a.n = n_; a.m = m_; a.S = S;
      a.first = enq == 0; a.gslot = enq & 1;
      a.rec_s0 = rec_t; a.nrec_s0 = nrec_s0;
      a.rec_p = rec_p; a.nrec_p = gp;
      a.rec_q = rec_a_sum; a.nrec_q = nrec_q;
      a.shift = shift; a.eps = kEps;
      a.p = cg_p_.p; a.x = x; a.q = cg_q_.p; a.r = cg_r_.p;
      a.ycur = y_[cur_].p; a.ynew = ysync ? nullptr : y_[nw].p;
      a.rec_x = rec_x;
      a.u = multi_ ? cg_u_.p : nullptr; a.t = tsum_.p; a.s = cg_s_.p; a.rec_s = rec_s2;
      a.nb_n = gp;
      hipLaunchKernelGGL(cgf_step_a_kernel<T>, dim3(gv), dim3(kCgfTpb), 0, s, a);
      int nrec_s = gp;
      if (!multi_) {
        // s = A^T r - shift x ; |s|^2 records                                (:281-286)
        nrec_s = spmv_cg(At_, cg_r_.p, SpAxpbyNormOp<T>{1, static_cast<T>(-shift), x, cg_s_.p}, rec_t, 0, &e);
        ev.push_back(e);
      }
      // beta, gamma, the stopping test ; p = s + beta p ; |p|^2              (:288-305)
      CgfStepB<T> b;
      b.n = n_; b.S = S; b.k = enq;
      b.rec_s = rec_s2; b.nrec_s = nrec_s;
      b.rec_x = rec_x; b.nrec_x = gp;
      b.tol = tol; b.maxit = 500;                                             // projector_cgls.cpp:17
      b.s = cg_s_.p; b.p = cg_p_.p; b.rec_p = rec_p;
      hipLaunchKernelGGL(cgf_step_b_kernel<T>, dim3(gp), dim3(kCgfTpb), 0, s, b);
      ++enq;
    };
    auto close = [&]() {
      // x <- x + x0 (projector_cgls.cpp:75), the bookkeeping of both halves (y_new from the
      // recurrence), the iteration's sums and the publish
      CgfClose<T> c;
      c.n = n_; c.m = m_; c.S = S;
      c.x = x; c.xprev = x_[cur_].p; c.x12 = x12_.p; c.xtemp = xtemp_.p;
      c.ynew = ysync ? nullptr : y_[nw].p; c.yprev = y_[cur_].p; c.y12 = y12_.p; c.ytemp = ytemp_.p;
      c.part = cpart; c.blocks_x = bx;
      hipLaunchKernelGGL(cgf_close_kernel<T>, dim3(ysync ? bx : bx + bm), dim3(kVecTpb), 0, s, c);
      ctx_.queue_sum(SumJob{pa.partials, bx, 3, S + kGapX});
      ctx_.queue_sum(SumJob{cpart, bx, 2, S + kDXprev2});
      if (!multi_) {
        ctx_.queue_sum(SumJob{pa.partials + static_cast<size_t>(bx) * 3, bm, 3, S + kGapY});
        if (!ysync) ctx_.queue_sum(SumJob{cpart + static_cast<size_t>(bx) * 2, bm, 2, S + kDYprev2});
      } else {
        // the y-side sums are over this rank's rows: slots kGapY .. kDY12 are adjacent, one all-reduce
        static_assert(kWY2 == kGapY + 1 && kHY2 == kGapY + 2 && kDYprev2 == kGapY + 3 && kDY12 == kGapY + 4, "slot order");
        SumJob j[2] = {{pa.partials + static_cast<size_t>(bx) * 3, bm, 3, S + kGapY},
                       {cpart + static_cast<size_t>(bx) * 2, bm, 2, S + kDYprev2}};
        launch_sum_jobs(j, ysync ? 1 : 2, s);
        ctx_.dist.allreduce(S + kGapY, ysync ? 3 : 5, s);
      }
      return ctx_.fetch_scalars();
    };
    const int ahead = std::max(1, std::min(cg_pred_, 500));
    for (int k = 0; k < ahead; ++k) step();
    const double *Sh = close();
    int more = 1;
    while (Sh[kFcDone] == 0.0) {   // the loop needed more steps than the previous projection
      for (int k = 0; k < more && enq < 500; ++k) step();
      Sh = close();
      more = std::min(2 * more, 64);   // a growing chunk per host poll, not one step per poll
    }
    const int steps = static_cast<int>(Sh[kFcSteps]);
    cg_pred_ = std::max(1, steps);
    ctx_.stats.cg_iters += static_cast<unsigned long long>(steps);
    // launches that found the loop ended were no-ops
    for (size_t k = 2 * static_cast<size_t>(steps); k < ev.size(); ++k) ctx_.stream_timer.drop(ev[k]);
    timed_spmvs_ += 1 + 2 * static_cast<unsigned long long>(steps);
    if (ysync) {
      // y = A x fused with the y-half bookkeeping                             (projector_cgls.cpp:78)
      spmv<false>(A_, x, nullptr, SpTailOp<T>{y_[nw].p, y_[cur_].p, y12_.p, ytemp_.p}, S + kDYprev2, 0, true);
      reduce_y_scalars(S + kDYprev2, 2);
      Sh = ctx_.fetch_scalars();
    }
    return Sh;
  }

  bool iteration(unsigned verbose) {
    hipStream_t s = ctx_.stream;
    const int nw = cur_ ^ 1;
    AdmmPreArgs<T> pa;
    pa.n_x = n_; pa.n_y = m_;
    pa.g = gview(); pa.f = fview();
    pa.x_cur = x_[cur_].p; pa.y_cur = y_[cur_].p;
    pa.xt = xt_.p; pa.yt = yt_.p;
    pa.zt_scale = zt_scale_;
    pa.x12 = x12_.p; pa.y12 = y12_.p;
    pa.xtemp = xtemp_.p; pa.ytemp = ytemp_.p;
    pa.rho = ctl_.rho; pa.alpha = ctl_.alpha(); pa.cheap = pre_cheap_;
    pa.uf = uni_f_; pa.ug = uni_g_;
    pa.partials = ctx_.spart.p + sp_pre_off_;
    pa.blocks_x = pre_blocks(n_);
    const double *S;
    if (fused_cg_) {
      S = prox_and_project_fused(pa, nw);
    } else {
    launch_admm_pre<T>(pa, s);
    {
      SumJob j[2] = {{pa.partials, pa.blocks_x, 3, ctx_.S.p + kGapX},
                     {pa.partials + static_cast<size_t>(pa.blocks_x) * 3, pre_blocks(m_), 3, ctx_.S.p + kGapY}};
      if (!multi_) {
        // one GPU: summed by the launch that publishes the scalar block next (the first fetch of the
        // projection); the partials have a region of their own until then
        ctx_.queue_sum(j[0]);
        ctx_.queue_sum(j[1]);
      } else {
        launch_sum_jobs(j, 2, s);
        reduce_y_scalars(ctx_.S.p + kGapY, 3);
      }
    }
    // warm start with the previous x (pogs.cpp:281), then CGLS
    cgls_project(xtemp_.p, ytemp_.p, x_[nw].p, ctl_.proj_tol(), y_[cur_].p, x_[cur_].p);   // y_cur == A x_cur
    // y = A x fused with the y-half bookkeeping; x-half element-wise        (projector_cgls.cpp:78)
    spmv<false>(A_, x_[nw].p, nullptr, SpTailOp<T>{y_[nw].p, y_[cur_].p, y12_.p, ytemp_.p}, ctx_.S.p + kDYprev2, 0,
                true);
    reduce_y_scalars(ctx_.S.p + kDYprev2, 2);
    launch_admm_tail<T>(n_, x_[nw].p, x_[cur_].p, x12_.p, xtemp_.p, ctx_.spart.p, s);
    ctx_.queue_sum(SumJob{ctx_.spart.p, vec_blocks(n_), 2, ctx_.S.p + kDXprev2});   // x side: no exchange; summed by the fetch below
    S = ctx_.fetch_scalars();
    }
    ctl_.set_pre(S);
    bool exact = false;
    if (ctl_.set_approx(S, nrmA_)) {
      spmv<false>(A_, x12_.p, nullptr, SpExactROp<T>{y12_.p}, ctx_.S.p + kExactR2, 0, true);
      reduce_y_scalars(ctx_.S.p + kExactR2, 1);
      hipLaunchKernelGGL(exact_u_kernel<T>, dim3((m_ + 255) / 256), dim3(256), 0, s, m_, y12_.p, yt_.p, y_[cur_].p,
                         zt_scale_, u_.p);
      spmv_t<false>(u_.p, SpExactSOp<T>{x12_.p, xt_.p, x_[cur_].p, zt_scale_}, ctx_.S.p + kExactS2, true);
      S = ctx_.fetch_scalars();
      ctl_.set_exact(S);
      exact = true;
    }
    const bool stop = ctl_.check_stop(exact);
    if (wants_iter_line(verbose, ctl_)) {
      const double obj = eval_objective();
      if (ctx_.dist.rank() == 0) print_iter_line(ctl_, obj);
    }
    if (stop) return true;
    std::swap(xt_, xtemp_);
    std::swap(yt_, ytemp_);
    cur_ = nw;
    zt_scale_ = ctl_.adapt();
    ++ctl_.k;
    return false;
  }

  // sum f(y12) + sum g(x12) at the current prox point (pogs.cpp:385, 473)
  double eval_objective() {
    hipStream_t s = ctx_.stream;
    const int by = vec_blocks(m_), bx = vec_blocks(n_);
    launch_func_eval<T>(m_, fview(), y12_.p, ctx_.spart.p, s);
    launch_func_eval<T>(n_, gview(), x12_.p, ctx_.spart.p + by, s);
    SumJob j[2] = {{ctx_.spart.p, by, 1, ctx_.S.p + kFvalF}, {ctx_.spart.p + by, bx, 1, ctx_.S.p + kFvalG}};
    launch_sum_jobs(j, 2, s);
    reduce_y_scalars(ctx_.S.p + kFvalF, 1);
    const double *S = ctx_.fetch_scalars();
    return static_cast<double>(static_cast<T>(S[kFvalF]) + static_cast<T>(S[kFvalG]));
  }

  int epilogue(void *x, void *y, void *l, void *mu, double *optval) {
    hipStream_t s = ctx_.stream;
    const int by = vec_blocks(m_), bx = vec_blocks(n_);
    launch_func_eval<T>(m_, fview(), y12_.p, ctx_.spart.p, s);
    launch_func_eval<T>(n_, gview(), x12_.p, ctx_.spart.p + by, s);
    SumJob j[2] = {{ctx_.spart.p, by, 1, ctx_.S.p + kFvalF}, {ctx_.spart.p + by, bx, 1, ctx_.S.p + kFvalG}};
    launch_sum_jobs(j, 2, s);
    reduce_y_scalars(ctx_.S.p + kFvalF, 1);
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

  void collect_timer() {
    ctx_.stats.reserved[2] = static_cast<double>(ctx_.dist.collectives());   // all-reduce calls since creation
    ctx_.stats.reserved[3] = static_cast<double>(ctx_.dist.comm_nranks());   // ranks as the communicator reports them
    ctx_.stats.matvecs += timed_spmvs_;
    if (ctx_.stream_timer.enabled()) {
      unsigned long long cnt = 0;
      ctx_.stats.stream_ms += ctx_.stream_timer.collect_ms(&cnt);
      ctx_.stats.stream_launches += cnt;
      // algorithmic bytes of one SpMV: nnz (s + 4) + 4 (rows + 1) + s (rows + cols)
      const double per = static_cast<double>(nnz_) * (sizeof(T) + 4) + 4.0 * (0.5 * (m_ + n_) + 1) +
                         static_cast<double>(sizeof(T)) * (m_ + n_);
      ctx_.stats.stream_bytes += static_cast<double>(cnt) * per;
    }
    timed_spmvs_ = 0;
  }

  Ctx ctx_;
  int m_ = 0, n_ = 0;
  size_t nnz_ = 0;
  bool first_is_A_ = true;
  bool multi_ = false;
  DevBuf<T> tsum_;   // row shards: this rank's A^T partial sums before the all-reduce
  DevBuf<T> cg_u_;   // row shards, device-resident CG loop: A^T r over all ranks, kept by recurrence
  int spmv_grid_ = 2048;
  size_t sp_cgx_off_ = 0, sp_cgp_off_ = 0, sp_pre_off_ = 0;   // regions of ctx_.spart (alloc_state)
  unsigned long long timed_spmvs_ = 0;
  std::vector<size_t> fused_events_;
  bool warm_pending_ = false;
  std::vector<T> warm_x_, warm_l_;
  DevCsr<T> A_, At_;
  DevBuf<T> d_, e_;
  DevBuf<T> x_[2], y_[2], xt_, yt_, xtemp_, ytemp_, x12_, y12_;
  DevBuf<T> cg_p_, cg_s_, cg_q_, cg_r_, cg_b_, u_;
  DevBuf<T> xout_, yout_, lout_, muout_;
  DevBuf<double> cg_;
  // device-resident CGLS loop (cg_fused.h)
  bool fused_cg_ = false;
  int cg_pred_ = 1;              // CG steps enqueued ahead: what the previous projection took
  DevBuf<double> cg_rec_;        // scalar records of its products: [A^T products | A products]
  size_t cg_rec_cap_ = 0;
  int ysync_ = 16;               // y = A x explicitly every ysync_-th iteration (0: always), else by recurrence
  unsigned long long proj_count_ = 0;
  bool pre_cheap_ = false;       // every f_i, g_j has a few-operation prox (admm_pre_kernel inlines it)
  FnUniform<T> uni_f_, uni_g_;   // coefficient arrays of f / g that hold one value throughout (load_problem)
  FnBuf<T> f_, g_, fs_, gs_;
  AdmmControl<T> ctl_;
  bool loaded_ = false;   // load_problem has run: f, g and the control block are valid
  int cur_ = 0;
  T zt_scale_ = 1;
  T nrmA_ = 0;
};

}  // namespace

SolverBase *make_sparse_solver(int dtype, int ord, size_t m, size_t n, size_t nnz, const void *data, const int *ptr,
                               const int *ind, int mem, const PogsAmdOptions *opt, const PogsAmdDist *dist) {
  if (dtype == POGS_AMD_F32) return new SparseSolver<float>(ord, m, n, nnz, data, ptr, ind, mem, opt, dist);
  if (dtype == POGS_AMD_F64) return new SparseSolver<double>(ord, m, n, nnz, data, ptr, ind, mem, opt, dist);
  throw Error("unknown dtype");
}

}  // namespace pogs_amd
