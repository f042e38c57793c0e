// MFMA GEMM + blocked Cholesky + triangular inverse (see gemm.h).
#include "gemm.h"

#include <cmath>
#include <type_traits>

namespace pogs_amd {

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef double doublex4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64: one wavefront computes a
// 16x16 tile, D = A(16x4) B(4x16) + C.  Lane l supplies A[l&15][l>>4] and
// B[l>>4][l&15]; it owns 4 results in column l&15 at the rows given by row().
template <typename T> struct Mma;
template <> struct Mma<float> {
  using Acc = floatx4;
  static __device__ __forceinline__ Acc mma(float a, float b, Acc c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
};
template <> struct Mma<double> {
  using Acc = doublex4;
  static __device__ __forceinline__ Acc mma(double a, double b, Acc c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LS = 144;  // LDS row stride: 144 = 16 mod 32 (f32) and 288 = 32 mod 64 (f64 dwords)
constexpr int GT = 256;  // threads per workgroup: 4 waves as 2 x 2, each 64 x 64

template <typename T> struct TileVec {
  static constexpr int VEC = Vec16<T>::N;
  static constexpr int NVT = (BM * BK) / VEC / GT;  // 16-byte vectors per thread per operand tile
};

// Loads a 128 (i) x 16 (k) operand tile into registers, zero-filling out of range.
// FAST: the caller knows the whole tile is in range -- unconditional 16-byte loads.  With the
// guarded form every load sits in its own branch and the compiler, unable to count the loads in
// flight, waits `vmcnt(0)` before the tile is written to LDS: that also waits for the NEXT tile's
// loads, issued a moment ago, i.e. it drains the prefetch every k-tile (fp64 Gram at C2: MFMA busy
// 59 % at the full 2.38 GHz).  The unconditional form lets it wait for the older stage only.
template <typename T, bool KMAJ, bool FAST = false>
__device__ __forceinline__ void load_tile(const T *__restrict__ P, size_t ld, int i0, int k0, int ilim,
                                          int klim, typename Vec16<T>::type (&regs)[TileVec<T>::NVT]) {
  using V = typename Vec16<T>::type;
  constexpr int VEC = TileVec<T>::VEC;
  const int t = threadIdx.x;
#pragma unroll
  for (int p = 0; p < TileVec<T>::NVT; ++p) {
    const int q = t + GT * p;
    T tmp[VEC];
    if (FAST) {
      const int per_row = KMAJ ? BM / VEC : BK / VEC;
      const int r = q / per_row, c = (q % per_row) * VEC;
      const T *src = KMAJ ? P + static_cast<size_t>(k0 + r) * ld + (i0 + c) : P + static_cast<size_t>(i0 + r) * ld + (k0 + c);
      regs[p] = *reinterpret_cast<const V *>(src);
      continue;
    }
    if (KMAJ) {
      constexpr int per_row = BM / VEC;
      const int gk = k0 + q / per_row;
      const int gi = i0 + (q % per_row) * VEC;
      const T *src = P + static_cast<size_t>(gk) * ld + gi;
      if (gk < klim && gi + VEC <= ilim) {
        regs[p] = *reinterpret_cast<const V *>(src);
        continue;
      }
#pragma unroll
      for (int c = 0; c < VEC; ++c) tmp[c] = (gk < klim && gi + c < ilim) ? src[c] : static_cast<T>(0);
    } else {
      constexpr int per_row = BK / VEC;
      const int gi = i0 + q / per_row;
      const int gk = k0 + (q % per_row) * VEC;
      const T *src = P + static_cast<size_t>(gi) * ld + gk;
      if (gi < ilim && gk + VEC <= klim) {
        regs[p] = *reinterpret_cast<const V *>(src);
        continue;
      }
#pragma unroll
      for (int c = 0; c < VEC; ++c) tmp[c] = (gi < ilim && gk + c < klim) ? src[c] : static_cast<T>(0);
    }
    regs[p] = *reinterpret_cast<const V *>(tmp);
  }
}

// Writes the register tile to LDS as sm[k][i] (row stride LS).
template <typename T, bool KMAJ>
__device__ __forceinline__ void store_tile(T *sm, const typename Vec16<T>::type (&regs)[TileVec<T>::NVT]) {
  using V = typename Vec16<T>::type;
  constexpr int VEC = TileVec<T>::VEC;
  const int t = threadIdx.x;
#pragma unroll
  for (int p = 0; p < TileVec<T>::NVT; ++p) {
    const int q = t + GT * p;
    if (KMAJ) {
      constexpr int per_row = BM / VEC;
      const int k = q / per_row, i = (q % per_row) * VEC;
      *reinterpret_cast<V *>(sm + k * LS + i) = regs[p];
    } else {
      constexpr int per_row = BK / VEC;
      const int i = q / per_row, k = (q % per_row) * VEC;
      const T *tp = reinterpret_cast<const T *>(&regs[p]);
#pragma unroll
      for (int c = 0; c < VEC; ++c) sm[(k + c) * LS + i] = tp[c];
    }
  }
}

template <typename T, bool A_KMAJ, bool B_KMAJ, bool LOWER, bool TWOLEVEL>
__device__ __forceinline__ void gemm_body(GemmArgs<T> g) {
  using V = typename Vec16<T>::type;
  using Acc = typename Mma<T>::Acc;
  // two LDS stages per operand: the next k-tile is written while the current one is read,
  // one barrier per k-tile
  __shared__ __attribute__((aligned(16))) T sA[2][BK * LS];
  __shared__ __attribute__((aligned(16))) T sB[2][BK * LS];

  // XCD-aware unit order: workgroup b runs on XCD b % 8 (observed dispatch order); give each
  // XCD a contiguous range of units so neighbouring tiles (shared operand panels, same K
  // range) meet in one L2.  Units beyond the real count exit.
  const int tm_ = (g.M + BM - 1) / BM, tn_ = (g.N + BN - 1) / BN;
  const int ntiles = LOWER ? tm_ * (tm_ + 1) / 2 : tm_ * tn_;
  const int per_batch = ntiles * g.ksplit;
  const int nunits = per_batch * g.batch;
  const int per_xcd = (nunits + kNumXcd - 1) / kNumXcd;
  int unit = static_cast<int>(blockIdx.x % kNumXcd) * per_xcd + static_cast<int>(blockIdx.x / kNumXcd);
  if (unit >= nunits || static_cast<int>(blockIdx.x / kNumXcd) >= per_xcd) return;
  const int q = unit / per_batch;
  unit -= q * per_batch;
  g.A += q * g.strideA;
  g.B += q * g.strideB;
  g.C += q * g.strideC;
  const int ks = unit / ntiles;
  const int tile = unit % ntiles;
  const bool split = g.kchunk > 0;
  int kbeg = split ? (g.ks0 + ks) * g.kchunk : 0;
  int kend = split ? ((kbeg + g.kchunk < g.K) ? kbeg + g.kchunk : g.K) : g.K;
  T *Cout = g.C + static_cast<size_t>(ks) * g.csplit_stride;
  int ti, tj;
  if (g.tile_map) {
    const int e = g.tile_map[tile];
    ti = e >> 16;
    tj = e & 0xffff;
  } else if (LOWER) {
    const int p = tile;
    ti = static_cast<int>((sqrt(8.0 * p + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > p) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= p) ++ti;
    tj = p - ti * (ti + 1) / 2;
  } else if (g.ktri == 2) {
    // the K range grows with the tile row: consecutive units (= one XCD's share, see above) walk
    // down a tile column, so every XCD gets short and long rows alike
    ti = tile % tm_;
    tj = tile / tm_;
  } else {
    ti = tile / tn_;
    tj = tile % tn_;
  }
  const int i0 = ti * BM, j0 = tj * BN;
  if (g.ktri == 1) kbeg = max(kbeg, j0);              // BN is a multiple of BK
  if (g.ktri == 2) kend = min(kend, i0 + BM);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int l15 = lane & 15, lk = lane >> 4;

  Acc acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = 0;

  // Two register stages + two LDS stages: the global loads of k-tile t+2 are issued two
  // compute phases before they are written to LDS.  The k-tile count is rounded up to an
  // even number (load_tile zero-fills past kend), which keeps the loop body branch-free.
  V ra0[TileVec<T>::NVT], rb0[TileVec<T>::NVT], ra1[TileVec<T>::NVT], rb1[TileVec<T>::NVT];
  const int nk = ((kend - kbeg + BK - 1) / BK + 1) & ~1;
#define POGS_GEMM_COMPUTE(CA, CB)                                                        \
  _Pragma("unroll") for (int ks = 0; ks < BK / 4; ++ks) {                                \
    T af[4], bf[4];                                                                      \
    const T *pa = (CA) + (ks * 4 + lk) * LS + wm + l15;                                  \
    const T *pb = (CB) + (ks * 4 + lk) * LS + wn + l15;                                  \
    _Pragma("unroll") for (int a = 0; a < 4; ++a) af[a] = pa[a * 16];                    \
    _Pragma("unroll") for (int b = 0; b < 4; ++b) bf[b] = pb[b * 16];                    \
    _Pragma("unroll") for (int a = 0; a < 4; ++a)                                        \
      _Pragma("unroll") for (int b = 0; b < 4; ++b)                                      \
        acc[a][b] = Mma<T>::mma(af[a], bf[b], acc[a][b]);                                \
  }
  load_tile<T, A_KMAJ>(g.A, g.lda, i0, kbeg, g.M, kend, ra0);
  load_tile<T, B_KMAJ>(g.B, g.ldb, j0, kbeg, g.N, kend, rb0);
  store_tile<T, A_KMAJ>(sA[0], ra0);
  store_tile<T, B_KMAJ>(sB[0], rb0);
  load_tile<T, A_KMAJ>(g.A, g.lda, i0, kbeg + BK, g.M, kend, ra0);
  load_tile<T, B_KMAJ>(g.B, g.ldb, j0, kbeg + BK, g.N, kend, rb0);
  __syncthreads();
  Acc tot[TWOLEVEL ? 4 : 1][TWOLEVEL ? 4 : 1];
  if (TWOLEVEL) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) tot[a][b][r] = 0;
  }
  const int acc_tiles = TWOLEVEL ? g.kacc / BK : 0;   // even
  int until_flush = acc_tiles;
  // Interior tiles (all 128 rows of both operand panels in range) run the pairs of k-tiles whose
  // prefetched tiles kt + 2, kt + 3 lie wholly below kend with the unconditional loader; the last
  // pairs and edge tiles take the guarded one.
  const int nfull = (kend - kbeg) / BK;   // k-tiles wholly in range
  const bool interior = i0 + BM <= g.M && j0 + BN <= g.N;
  const int kt_fast = interior ? ((nfull - 3) & ~1) : 0;   // pairs kt < kt_fast prefetch tiles <= kt + 3 < nfull
#define POGS_GEMM_PAIR(FAST_)                                                                          \
  {                                                                                                      \
    /* even phase: LDS[0] = tile kt, stage 0 = tile kt+1 (in flight), request tile kt+2 */              \
    load_tile<T, A_KMAJ, FAST_>(g.A, g.lda, i0, kbeg + (kt + 2) * BK, g.M, kend, ra1);                   \
    load_tile<T, B_KMAJ, FAST_>(g.B, g.ldb, j0, kbeg + (kt + 2) * BK, g.N, kend, rb1);                   \
    POGS_GEMM_COMPUTE(sA[0], sB[0])                                                                      \
    store_tile<T, A_KMAJ>(sA[1], ra0);                                                                   \
    store_tile<T, B_KMAJ>(sB[1], rb0);                                                                   \
    __syncthreads();                                                                                     \
    /* odd phase: LDS[1] = tile kt+1, stage 1 = tile kt+2 (in flight), request tile kt+3 */             \
    load_tile<T, A_KMAJ, FAST_>(g.A, g.lda, i0, kbeg + (kt + 3) * BK, g.M, kend, ra0);                   \
    load_tile<T, B_KMAJ, FAST_>(g.B, g.ldb, j0, kbeg + (kt + 3) * BK, g.N, kend, rb0);                   \
    POGS_GEMM_COMPUTE(sA[1], sB[1])                                                                      \
    store_tile<T, A_KMAJ>(sA[0], ra1);                                                                   \
    store_tile<T, B_KMAJ>(sB[0], rb1);                                                                   \
    __syncthreads();                                                                                     \
    if (TWOLEVEL) {                                                                                      \
      until_flush -= 2;                                                                                  \
      if (until_flush == 0) {                                                                            \
        until_flush = acc_tiles;                                                                         \
        _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                    \
          _Pragma("unroll") for (int b = 0; b < 4; ++b) {                                                \
            tot[a][b] += acc[a][b];                                                                      \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) acc[a][b][r] = 0;                              \
          }                                                                                              \
      }                                                                                                  \
    }                                                                                                    \
  }
  int kt = 0;
  if constexpr (!TWOLEVEL) {   // (the two-level form has no registers for a second copy of the loop: fp64 would spill)
    for (; kt < kt_fast; kt += 2) POGS_GEMM_PAIR(true)
  }
  for (; kt < nk; kt += 2) POGS_GEMM_PAIR(false)
#undef POGS_GEMM_PAIR
#undef POGS_GEMM_COMPUTE
  if (TWOLEVEL) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] += tot[a][b];
  }

#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = i0 + wm + a * 16 + Mma<T>::row(lane, r);
        const int col = j0 + wn + b * 16 + l15;
        if (row < g.M && col < g.N) {
          T *c = Cout + static_cast<size_t>(row) * g.ldc + col;
          T v = g.alpha * acc[a][b][r];
          if (g.beta != static_cast<T>(0)) v += g.beta * *c;
          *c = v;
        }
      }
}

template <typename T, bool A_KMAJ, bool B_KMAJ, bool LOWER, bool TWOLEVEL = false>
__global__ void __launch_bounds__(GT) gemm_kernel(GemmArgs<T> g) {
  gemm_body<T, A_KMAJ, B_KMAJ, LOWER, TWOLEVEL>(g);
}
// The same body held to 256 registers, i.e. two workgroups per CU (2 x 72 KB of LDS): every fp64 form
// with one accumulator set.  Left alone the fp64 body takes 384 .. 417 registers (256 + 128 accumulators)
// and runs ONE wavefront per SIMD, so the matrix pipe idles whenever that wavefront reads LDS, stores a
// tile or waits at the barrier -- the fp64 Gram product of C2 kept it busy 59 % of the time at the full
// 2.38 GHz (it is not power-limited, unlike the fp16-split product).  Under the bound the K-major form
// (the Gram product) fits without spilling; the row-major and mixed forms (Cholesky updates, triangular
// inverse) spill 12 .. 23 registers and are faster all the same.  Measured at C2 in fp64 together with
// the branch-free loader above: Gram 228 -> 155 ms (68 TFLOP/s = 0.86 of the fp64 matrix peak),
// Cholesky 39.6 -> 29.1 ms, inverse 12.9 -> 9.9 ms.  The two-level form would spill 200+ and keeps
// gemm_kernel.
template <typename T, bool A_KMAJ, bool B_KMAJ, bool LOWER>
__global__ void __launch_bounds__(GT) __attribute__((amdgpu_waves_per_eu(2))) gemm_kernel_w2(GemmArgs<T> g) {
  gemm_body<T, A_KMAJ, B_KMAJ, LOWER, false>(g);
}

// The TS steps j = TS JB .. TS JB + TS - 1 of potrf_inv_kernel.  JB is a template parameter so
// that which register groups take part is known at compile time (rows a >= JB; L columns
// b >= JB; X columns b <= JB) and the register arrays are only indexed statically.
template <typename T, int NB, int TS, int JB>
__device__ __forceinline__ void potrf_steps(T (&Lr)[NB / TS][NB / TS], T (&Xr)[NB / TS][NB / TS], T (&colj)[2][NB],
                                            T (&rowj)[2][NB], int nb, int tx, int ty) {
  constexpr int NBK = NB / TS;
  for (int jl = 0; jl < TS; ++jl) {
    const int j = TS * JB + jl, par = jl & 1;
    if (j >= nb) break;
    if (tx == jl) {
#pragma unroll
      for (int a = JB; a < NBK; ++a) colj[par][ty + TS * a] = Lr[a][JB];
    }
    if (ty == jl) {
#pragma unroll
      for (int b = 0; b <= JB; ++b) rowj[par][tx + TS * b] = Xr[JB][b];
    }
    __syncthreads();
    // unconditional LDS reads (a read under a lane condition becomes a branch with its own wait)
    T lij[NBK], fl[NBK], fx[NBK];
    const T djj = colj[par][j];
#pragma unroll
    for (int a = JB; a < NBK; ++a) lij[a] = colj[par][ty + TS * a];
#pragma unroll
    for (int b = JB; b < NBK; ++b) fl[b] = colj[par][tx + TS * b];
#pragma unroll
    for (int b = 0; b <= JB; ++b) fx[b] = rowj[par][tx + TS * b];
    const T ljj = sqrt(djj);
    const T inv = static_cast<T>(1) / ljj;
#pragma unroll
    for (int a = JB; a < NBK; ++a) lij[a] = (ty + TS * a > j) ? lij[a] * inv : static_cast<T>(0);
#pragma unroll
    for (int b = JB; b < NBK; ++b) fl[b] = (tx + TS * b > j) ? fl[b] * inv : static_cast<T>(0);
#pragma unroll
    for (int b = 0; b <= JB; ++b) fx[b] = (tx + TS * b <= j) ? fx[b] * inv : static_cast<T>(0);
    // the owners finish column j of L and row j of X
    if (tx == jl) {
#pragma unroll
      for (int a = JB; a < NBK; ++a) {
        const int i = ty + TS * a;
        Lr[a][JB] = (i == j) ? ljj : ((i > j) ? lij[a] : Lr[a][JB]);
      }
    }
    if (ty == jl) {
#pragma unroll
      for (int b = 0; b <= JB; ++b) Xr[JB][b] = (tx + TS * b <= j) ? fx[b] : Xr[JB][b];
    }
#pragma unroll
    for (int a = JB; a < NBK; ++a) {
#pragma unroll
      for (int b = JB; b < NBK; ++b) Lr[a][b] -= lij[a] * fl[b];
#pragma unroll
      for (int b = 0; b <= JB; ++b) Xr[a][b] -= lij[a] * fx[b];
    }
  }
  if constexpr (JB + 1 < NBK) potrf_steps<T, NB, TS, JB + 1>(Lr, Xr, colj, rowj, nb, tx, ty);
}

// Cholesky of one NB x NB diagonal block together with the inverse of its factor, one
// workgroup, right-looking, the whole block in registers: thread (ty, tx) of a TS x TS layout
// owns the elements (ty + TS a, tx + TS b) of L and of X = L^-1.  Step j: the owners publish
// column j of L and row j of X (unscaled) through LDS, one barrier, then every thread applies
//   L[i][c] -= L[i][j] L[c][j]  (j < c, j < i)      X[i][c] -= L[i][j] X[j][c]  (c <= j < i)
// to its own registers; TS-row / TS-column groups that lie entirely outside the active region
// are skipped with uniform branches.  (A version that kept the block in LDS spent 340 us per
// 128-block on address arithmetic and LDS round trips; the Cholesky was 75 % potrf time.)
// TS = 16 (shipped): 256 threads with (NB / 16)^2 elements of each matrix; TS = 32: 1024 threads (16 wavefronts, four per
// SIMD) with a quarter of the per-step arithmetic each -- same bits, measured slower (kPotrfTS below).
template <typename T, int NB, int TS>
__global__ void __launch_bounds__(TS * TS) potrf_inv_kernel(T *G, size_t ldg, int nb, T *Winv, size_t ldw) {
  constexpr int NBK = NB / TS;
  static_assert(NB % TS == 0 && (TS == 16 || TS == 32), "thread layout");
  __shared__ T colj[2][NB], rowj[2][NB];
  const int t = threadIdx.x;
  const int tx = t % TS, ty = t / TS;
  T Lr[NBK][NBK], Xr[NBK][NBK];
#pragma unroll
  for (int a = 0; a < NBK; ++a)
#pragma unroll
    for (int b = 0; b < NBK; ++b) {
      const int i = ty + TS * a, c = tx + TS * b;
      const T unit = (i == c) ? static_cast<T>(1) : static_cast<T>(0);
      Lr[a][b] = (i < nb && c <= i) ? G[static_cast<size_t>(i) * ldg + c] : unit;
      Xr[a][b] = unit;
    }
  potrf_steps<T, NB, TS, 0>(Lr, Xr, colj, rowj, nb, tx, ty);
#pragma unroll
  for (int a = 0; a < NBK; ++a)
#pragma unroll
    for (int b = 0; b < NBK; ++b) {
      const int i = ty + TS * a, c = tx + TS * b;
      if (i < nb && c <= i) {
        G[static_cast<size_t>(i) * ldg + c] = Lr[a][b];
        Winv[static_cast<size_t>(i) * ldw + c] = Xr[a][b];
      }
    }
}
#ifndef POGS_POTRF_TS   // (A / B builds; 32 = 1024 threads was measured in round 6 and is slower: the 16-wavefront barrier of
#define POGS_POTRF_TS 16   //  every step costs more than the quarter of the arithmetic saves -- Cholesky 12.28 -> 13.4 ms at C2)
#endif
constexpr int kPotrfTS = POGS_POTRF_TS;

template <typename T>
__global__ void __launch_bounds__(256) transpose_kernel(const T *in, size_t ld_in, int rows, int cols, T *out,
                                                        size_t ld_out) {
  __shared__ T tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int k = 0; k < 32; k += 8) {
    const int r = by + ty + k, c = bx + tx;
    tile[ty + k][tx] = (r < rows && c < cols) ? in[static_cast<size_t>(r) * ld_in + c] : static_cast<T>(0);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 32; k += 8) {
    const int r = bx + ty + k, c = by + tx;  // out is cols x rows
    if (r < cols && c < rows) out[static_cast<size_t>(r) * ld_out + c] = tile[tx][ty + k];
  }
}

// out[r][c] = sum_s in[s][r][c] over the lower 128 x 128 tiles (c < 128 * (r / 128 + 1)),
// slabs added in index order.  One workgroup per row.
template <typename T>
__global__ void __launch_bounds__(256) sum_slabs_kernel(const T *in, size_t stride, int nslabs, T *out,
                                                        size_t ld, int n) {
  using V = typename Vec16<T>::type;
  constexpr int VEC = Vec16<T>::N;
  const int r = blockIdx.x;
  const int cend = min(n, (r / BM + 1) * BM);
  const size_t base = static_cast<size_t>(r) * ld;
  for (int c = threadIdx.x * VEC; c < cend; c += 256 * VEC) {
    V acc = *reinterpret_cast<const V *>(in + base + c);
    T *ap = reinterpret_cast<T *>(&acc);
    for (int sl = 1; sl < nslabs; ++sl) {
      const V v = *reinterpret_cast<const V *>(in + static_cast<size_t>(sl) * stride + base + c);
      const T *vp = reinterpret_cast<const T *>(&v);
#pragma unroll
      for (int k = 0; k < VEC; ++k) ap[k] += vp[k];
    }
    if (c + VEC > n) {   // padding columns of the last vector stay zero
#pragma unroll
      for (int k = 0; k < VEC; ++k)
        if (c + k >= n) ap[k] = 0;
    }
    *reinterpret_cast<V *>(out + base + c) = acc;
  }
}

template <typename T>
__global__ void zero_upper_kernel(T *G, size_t ldg, int n) {
  const int r = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n && c > r) G[static_cast<size_t>(r) * ldg + c] = 0;
}

template <typename T>
__global__ void add_diag_kernel(T *G, size_t ldg, int n, T v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) G[static_cast<size_t>(i) * ldg + i] += v;
}

template <typename T, bool A_KMAJ, bool B_KMAJ>
void launch_gemm_ab(bool lower, const GemmArgs<T> &g, hipStream_t s) {
  const int tm = (g.M + BM - 1) / BM, tn = (g.N + BN - 1) / BN;
  if (tm <= 0 || tn <= 0 || g.K <= 0) return;
  GemmArgs<T> gg = g;
  if (gg.ksplit < 1) gg.ksplit = 1;
  if (gg.batch < 1) gg.batch = 1;
  const int nt = (lower ? tm * (tm + 1) / 2 : tm * tn) * gg.ksplit * gg.batch;
  const int grid = (nt + kNumXcd - 1) / kNumXcd * kNumXcd;
  if (lower && gg.kacc > 0 && A_KMAJ == B_KMAJ) {
    hipLaunchKernelGGL((gemm_kernel<T, A_KMAJ, B_KMAJ, true, true>), dim3(grid), dim3(GT), 0, s, gg);
  } else if constexpr (std::is_same<T, double>::value) {
    if (lower) hipLaunchKernelGGL((gemm_kernel_w2<T, A_KMAJ, B_KMAJ, true>), dim3(grid), dim3(GT), 0, s, gg);
    else hipLaunchKernelGGL((gemm_kernel_w2<T, A_KMAJ, B_KMAJ, false>), dim3(grid), dim3(GT), 0, s, gg);
  } else {
    if (lower) hipLaunchKernelGGL((gemm_kernel<T, A_KMAJ, B_KMAJ, true>), dim3(grid), dim3(GT), 0, s, gg);
    else hipLaunchKernelGGL((gemm_kernel<T, A_KMAJ, B_KMAJ, false>), dim3(grid), dim3(GT), 0, s, gg);
  }
}

}  // namespace

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// 8 fp32 -> two vectors of 8 fp16: h = fp16(a s), l = fp16(a s - h)
__device__ __forceinline__ void split8_f16(const float (&v)[8], float sc, f16x8 &h, f16x8 &l) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float a = v[q] * sc;
    const _Float16 hh = static_cast<_Float16>(a);
    h[q] = hh;
    l[q] = static_cast<_Float16>(a - static_cast<float>(hh));
  }
}

constexpr int PBK = 32;   // image rows per step of the pre-split kernel: four 8-row groups, two MFMA k-steps

__global__ void __launch_bounds__(256) split_f16_kernel(const float *P, size_t ld, int K, int N, int k0, int npad,
                                                        float scale, f16x8 *H, f16x8 *L) {
  // one thread: eight consecutive rows of one column -> one 16-byte group of each image
  const int col = blockIdx.x * 256 + threadIdx.x;
  const int kg = blockIdx.y;
  if (col >= npad) return;
  float v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int k = k0 + kg * 8 + q;
    v[q] = (k < K && col < N) ? P[static_cast<size_t>(k) * ld + col] : 0.f;
  }
  f16x8 h, l;
  split8_f16(v, scale, h, l);
  const size_t o = static_cast<size_t>(kg) * npad + col;
  H[o] = h;
  L[o] = l;
}

// WM x WN waves, each TA x TB MFMA tiles of 32 x 32; the workgroup tile is square (TILE = WM TA 32
// = WN TB 32): 2 x 2 waves of 2 x 2 -> 128, 2 x 4 waves of 4 x 2 -> 256.  One step = 32 image
// rows.  LDS stage: [operand A/B][part h/l][8-row group 0..3][TILE columns] x 16 B, two stages
// (64 KB at 128, 128 KB at 256).  A step's one-KB lines are copied by the waves (eight each) with
// global_load_lds; the copy of step i + 1 is in flight while step i is multiplied.
template <int WM, int WN, int TA, int TB>
__global__ void __launch_bounds__(WM * WN * 64) __attribute__((amdgpu_waves_per_eu(2))) gram_f16p_kernel(GramF16PArgs g) {
  constexpr int TILE = WM * TA * 32;
  static_assert(TILE == WN * TB * 32, "square workgroup tile");
  constexpr int QC = TILE / 64;                       // 64-column lines per row group
  static_assert(2 * 2 * 4 * QC == WM * WN * 8, "eight lines per wave and step");
  extern __shared__ __attribute__((aligned(16))) unsigned char gram_lds[];
  typedef f16x8 Stage[2][2][4][TILE];
  Stage *sh = reinterpret_cast<Stage *>(gram_lds);
  const int tm = (g.N + TILE - 1) / TILE;
  const int ntiles = tm * (tm + 1) / 2;
  const int nunits = ntiles * g.nslabs;
  const int per_xcd = (nunits + kNumXcd - 1) / kNumXcd;
  const int unit = static_cast<int>(blockIdx.x % kNumXcd) * per_xcd + static_cast<int>(blockIdx.x / kNumXcd);
  if (unit >= nunits || static_cast<int>(blockIdx.x / kNumXcd) >= per_xcd) return;
  const int ks = unit / ntiles, tile = unit % ntiles;
  int ti, tj;
  if (g.tile_map) {
    const int e = g.tile_map[tile];
    ti = e >> 16;
    tj = e & 0xffff;
  } else {
    ti = static_cast<int>((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > tile) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
    tj = tile - ti * (ti + 1) / 2;
  }
  const int i0 = ti * TILE, j0 = tj * TILE;
  float *Cout = g.C + static_cast<size_t>(ks) * g.slab_stride;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = (wave / WN) * (TA * 32), wn = (wave % WN) * (TB * 32);
  const int r32 = lane & 31, kh = lane >> 5;

  floatx16 acc[TA][TB];
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nsteps = g.kchunk / PBK;
  const size_t kg0 = static_cast<size_t>(ks) * (g.kchunk / 8);
  auto issue = [&](int st, int step) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int q = wave * 8 + j;
      const int cq = q % QC, kg = (q / QC) % 4, part = (q / (QC * 4)) % 2, op = q / (QC * 8);
      const f16x8 *img = reinterpret_cast<const f16x8 *>(part ? g.L : g.H);
      const f16x8 *src = img + (kg0 + static_cast<size_t>(step) * 4 + kg) * g.npad + (op ? j0 : i0) + cq * 64 + lane;
      f16x8 *dst = &sh[st][op][part][kg][cq * 64];
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src),
                                       (__attribute__((address_space(3))) void *)(dst), 16, 0, 0);
    }
  };
  auto compute = [&](int st) {
    // both k-halves' fragments are requested up front: the second half's reads land while the
    // first half's products run
    f16x8 A[2][TA][2], B[2][TB][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int a = 0; a < TA; ++a) A[kk][a][p] = sh[st][0][p][kk * 2 + kh][wm + a * 32 + r32];
#pragma unroll
        for (int b = 0; b < TB; ++b) B[kk][b][p] = sh[st][1][p][kk * 2 + kh][wn + b * 32 + r32];
      }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) {
          floatx16 c = acc[a][b];   // small products first
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[kk][a][0], B[kk][b][1], c, 0, 0, 0);   // h l
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[kk][a][1], B[kk][b][0], c, 0, 0, 0);   // l h
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[kk][a][0], B[kk][b][0], c, 0, 0, 0);   // h h
          acc[a][b] = c;
        }
    __builtin_amdgcn_s_setprio(0);
  };
  // The MFMA accumulate truncates, so a chain is kept to flush_rows (1024) rows: after that many
  // the chain's sum is added (IEEE) to a second set of registers and the chain restarts.  Units
  // can then span several chains and pay their prologue, first-copy latency and the
  // read-add-write of the C tile once per kchunk rather than once per chain.
  constexpr bool TWO = TA * TB <= 4;   // the 256 tile has no registers left for a second set
  floatx16 sum[TWO ? TA : 1][TWO ? TB : 1];
  if (TWO) {
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
      for (int b = 0; b < TB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum[TWO ? a : 0][TWO ? b : 0][r] = 0.f;
  }
  const int fsteps = TWO && g.flush_rows > 0 ? g.flush_rows / PBK : nsteps + 1;
  int until_flush = fsteps;
  if (nsteps > 0) issue(0, 0);
  for (int i = 0; i < nsteps; ++i) {
    // each wave waits for its own lines, the barrier then covers everybody's -- and says that
    // the stage about to be refilled has been read by all
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (i + 1 < nsteps) issue((i + 1) & 1, i + 1);
    compute(i & 1);
    if (TWO && --until_flush == 0) {
      until_flush = fsteps;
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            sum[TWO ? a : 0][TWO ? b : 0][r] += acc[a][b][r];
            acc[a][b][r] = 0.f;
          }
    }
  }
  if (TWO) {
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
      for (int b = 0; b < TB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] += sum[TWO ? a : 0][TWO ? b : 0][r];
  }
  const float inv2 = 1.0f / (g.scale * g.scale);
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + wm + a * 32 + (r / 4) * 8 + kh * 4 + (r % 4);
        const int col = j0 + wn + b * 32 + r32;
        if (row < g.N && col < g.N) {
          float *c = Cout + static_cast<size_t>(row) * g.ldc + col;
          const float v = acc[a][b][r] * inv2;
          *c = g.accumulate ? *c + v : v;
        }
      }
}


// The 256 x 256 tile with the two wavefronts of every SIMD HALF A PHASE APART.
//
// gram_f16p_kernel<2, 4, 4, 2> meets at one barrier per 32-row step, after which all eight
// wavefronts first read their fragments from LDS (16 KB + 8 KB each: 192 KB per step, ~1500 LDS
// cycles at 128 B/clk) and then all multiply (48 MFMAs of 32 cycles each, two wavefronts per SIMD:
// ~3070 cycles): the matrix pipes idle while the LDS is read and the LDS idles while they run --
// 1536 + 3072 = 4608 cycles per step is what round 3's counters show (MFMA busy 62 %), not the
// 3072 the products need.  Here a phase is 16 image rows (half a step: 12 fragment reads, 24 MFMAs
// per wavefront) in two parts with a raw s_barrier after each,
//
//     R: request the phase's fragments                      | barrier |
//     M: issue this wavefront's copies for phase p + 3; wait for the fragments; 24 MFMAs | barrier |
//
// and the wavefronts of workgroup row 1 (wavefronts 4-7: the SECOND wavefront of each SIMD) run
// the same program one barrier late (one extra barrier before their loop, one after for row 0 --
// every wavefront executes the same number).  Between two barriers one wavefront of a SIMD is in M
// and the other in R: fragment reads and DMA writes overlap the other wavefront's products
// (cdna_hip_programming.md section 5, "8-phase template": the per-phase role split).
//
// LDS: a ring of four phase slots of 32 KB, [operand A/B][part h/l][8-row group 0..1][256] x 16 B.
// Copies (global_load_lds, 4 one-KB lines per wavefront and phase) are issued three phases ahead
// and waited for with a counted vmcnt, never 0 inside the loop:
//   RAW  the fragments of phase p are read in R(p); every wavefront has waited for its lines of
//        phase p in R(p - 1) (vmcnt(4): only phase p + 1's four lines may still be in flight) and
//        has then passed a barrier that the reader passed too -- for the late row the barrier after
//        its R(p - 1) is the one before the early row's R(p).
//   WAR  phase p + 3 lands in the slot of phase p - 1, whose fragment reads were retired
//        (lgkmcnt(0) at the head of M(p - 1)) before the barrier that precedes this M(p) -- for
//        either row.
constexpr int kGramSlots = 4;   // ring of 32 KB phase slots: copies run three phases ahead (five slots = all 160 KB of
                                // LDS, four phases ahead, measured: 29.9 against 28.3 ms for the phase at C2, not faster)
template <int N> __device__ __forceinline__ void gram_wait_lines() {
  static_assert(N == 4 || N == 8 || N == 12, "counted vmcnt");
  if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) gram_f16s_kernel(GramF16PArgs g) {
  constexpr int TILE = 256, TA = 4, TB = 2, WN = 4;
  constexpr int SLOTS = kGramSlots, AHEAD = kGramSlots - 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char gram_lds[];
  typedef f16x8 Slot[2][2][2][TILE];   // 32 KB
  Slot *sh = reinterpret_cast<Slot *>(gram_lds);
  const int tm = (g.N + TILE - 1) / TILE;
  const int ntiles = tm * (tm + 1) / 2;
  const int nunits = ntiles * g.nslabs;
  const int per_xcd = (nunits + kNumXcd - 1) / kNumXcd;
  const int unit = static_cast<int>(blockIdx.x % kNumXcd) * per_xcd + static_cast<int>(blockIdx.x / kNumXcd);
  if (unit >= nunits || static_cast<int>(blockIdx.x / kNumXcd) >= per_xcd) return;
  const int ks = unit / ntiles, tile = unit % ntiles;
  int ti, tj;
  if (g.tile_map) {
    const int e = g.tile_map[tile];
    ti = e >> 16;
    tj = e & 0xffff;
  } else {
    ti = static_cast<int>((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > tile) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
    tj = tile - ti * (ti + 1) / 2;
  }
  const int i0 = ti * TILE, j0 = tj * TILE;
  float *Cout = g.C + static_cast<size_t>(ks) * g.slab_stride;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int late = wave / WN;   // workgroup row 1 = the second wavefront of each SIMD
  const int wm = late * (TA * 32), wn = (wave % WN) * (TB * 32);
  const int r32 = lane & 31, kh = lane >> 5;

  floatx16 acc[TA][TB];
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nph = g.kchunk / 16;   // phases of this unit (kchunk is a multiple of 32: even)
  const size_t kg0 = static_cast<size_t>(ks) * (g.kchunk / 8);
  // this wavefront's four lines of a phase: line q = wave * 4 + j of [op][part][group][4 x 64 columns]
  const f16x8 *src_line[4];
  int dst_off[4];   // in f16x8 units inside a slot
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = wave * 4 + j;
    const int cq = q % 4, kg = (q / 4) % 2, part = (q / 8) % 2, op = q / 16;
    const f16x8 *img = reinterpret_cast<const f16x8 *>(part ? g.L : g.H);
    src_line[j] = img + (kg0 + kg) * g.npad + (op ? j0 : i0) + cq * 64 + lane;
    dst_off[j] = ((op * 2 + part) * 2 + kg) * TILE + cq * 64;
  }
  const size_t phase_stride = static_cast<size_t>(2) * g.npad;   // two 8-row groups per phase
  auto issue = [&](int ph) {
    f16x8 *slot = reinterpret_cast<f16x8 *>(&sh[ph % SLOTS]);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src_line[j] + phase_stride * ph),
                                       (__attribute__((address_space(3))) void *)(slot + dst_off[j]), 16, 0, 0);
  };

  // (units are thousands of rows long: nph > AHEAD always; shorter ones would only wait longer)
#pragma unroll
  for (int q = 0; q < AHEAD; ++q)
    if (q < nph) issue(q);
  if (nph >= AHEAD) gram_wait_lines<4 * (AHEAD - 1)>();
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                 // phase 0 is in LDS for everybody
  if (late) __builtin_amdgcn_s_barrier();       // the late row starts one barrier behind
  for (int p = 0; p < nph; ++p) {
    // ---- R: the phase's fragments (one register set: the previous phase's products are done)
    f16x8 A[TA][2], B[TB][2];
    {
      const Slot &S = sh[p % SLOTS];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int b = 0; b < TB; ++b) B[b][q] = S[1][q][kh][wn + b * 32 + r32];
#pragma unroll
        for (int a = 0; a < TA; ++a) A[a][q] = S[0][q][kh][wm + a * 32 + r32];
      }
    }
    __builtin_amdgcn_sched_barrier(0);          // the reads are requested BEFORE the barrier
    // my lines of phase p + 1 have landed (only those of phases p + 2 .. p + AHEAD - 1, four each, may still be in flight)
    if (p + AHEAD - 1 < nph) gram_wait_lines<4 * (AHEAD - 2)>();
    else if (AHEAD > 3 && p + 2 < nph) gram_wait_lines<4>();   // tail: fewer phases behind this one
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- M
    if (p + AHEAD < nph) issue(p + AHEAD);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
      for (int b = 0; b < TB; ++b) {
        floatx16 c = acc[a][b];   // small products first
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[a][0], B[b][1], c, 0, 0, 0);   // h l
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[a][1], B[b][0], c, 0, 0, 0);   // l h
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[a][0], B[b][0], c, 0, 0, 0);   // h h
        acc[a][b] = c;
      }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
  if (!late) __builtin_amdgcn_s_barrier();      // same number of barriers for every wavefront
  const float inv2 = 1.0f / (g.scale * g.scale);
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + wm + a * 32 + (r / 4) * 8 + kh * 4 + (r % 4);
        const int col = j0 + wn + b * 32 + r32;
        if (row < g.N && col < g.N) {
          float *c = Cout + static_cast<size_t>(row) * g.ldc + col;
          const float v = acc[a][b][r] * inv2;
          *c = g.accumulate ? *c + v : v;
        }
      }
}

}  // namespace

void launch_split_f16(const float *P, size_t ld, int K, int N, int k0, int krows, int npad, float scale, void *H,
                      void *L, hipStream_t s) {
  if (krows <= 0) return;
  hipLaunchKernelGGL(split_f16_kernel, dim3((npad + 255) / 256, krows / 8), dim3(256), 0, s, P, ld, K, N, k0, npad,
                     scale, static_cast<f16x8 *>(H), static_cast<f16x8 *>(L));
}

template <int WM, int WN, int TA, int TB>
static void launch_gram_f16p_cfg(const GramF16PArgs &g, hipStream_t s) {
  constexpr int TILE = WM * TA * 32;
  const int tm = (g.N + TILE - 1) / TILE;
  const int nunits = tm * (tm + 1) / 2 * g.nslabs;
  if (nunits <= 0) return;
  constexpr int kLds = 2 * 2 * 2 * 4 * TILE * 16;
  static SmemGrants grants;
  ensure_dynamic_smem(reinterpret_cast<const void *>(gram_f16p_kernel<WM, WN, TA, TB>), kLds, grants);
  const int grid = (nunits + kNumXcd - 1) / kNumXcd * kNumXcd;
  hipLaunchKernelGGL((gram_f16p_kernel<WM, WN, TA, TB>), dim3(grid), dim3(WM * WN * 64), kLds, s, g);
}

static void launch_gram_f16s(const GramF16PArgs &g, hipStream_t s) {
  constexpr int TILE = 256;
  const int tm = (g.N + TILE - 1) / TILE;
  const int nunits = tm * (tm + 1) / 2 * g.nslabs;
  if (nunits <= 0) return;
  constexpr int kLds = kGramSlots * 2 * 2 * 2 * TILE * 16;   // phase slots of 32 KB
  static SmemGrants grants;
  ensure_dynamic_smem(reinterpret_cast<const void *>(gram_f16s_kernel), kLds, grants);
  const int grid = (nunits + kNumXcd - 1) / kNumXcd * kNumXcd;
  hipLaunchKernelGGL(gram_f16s_kernel, dim3(grid), dim3(512), kLds, s, g);
}

void launch_gram_f16p(const GramF16PArgs &g, hipStream_t s) {
  if (g.tile == 256) launch_gram_f16s(g, s);
  else launch_gram_f16p_cfg<2, 2, 2, 2>(g, s);
}

void preload_gemm_code() {
  hipFuncAttributes fa;
  (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(split_f16_kernel));
}

std::vector<int> gram_tile_order(int n, int tile) {
  constexpr int G = 8;
  const int tm = (n + tile - 1) / tile;
  std::vector<int> order;
  order.reserve(static_cast<size_t>(tm) * (tm + 1) / 2);
  const int sm = (tm + G - 1) / G;
  for (int I = 0; I < sm; ++I)
    for (int J = 0; J <= I; ++J)
      for (int a = 0; a < G; ++a)
        for (int b = 0; b < G; ++b) {
          const int ti = I * G + a, tj = J * G + b;
          if (ti < tm && tj <= ti) order.push_back((ti << 16) | tj);
        }
  return order;
}

template <typename T>
void launch_gemm(bool a_kmaj, bool b_kmaj, bool lower_only, const GemmArgs<T> &g, hipStream_t s) {
  if (a_kmaj && b_kmaj) launch_gemm_ab<T, true, true>(lower_only, g, s);
  else if (!a_kmaj && !b_kmaj) launch_gemm_ab<T, false, false>(lower_only, g, s);
  else if (!a_kmaj && b_kmaj) launch_gemm_ab<T, false, true>(lower_only, g, s);
  else launch_gemm_ab<T, true, false>(lower_only, g, s);
}

// (Measured in round 5 and not kept: a LOOK-AHEAD form -- the trailing update of group k on a second stream, in three
// pieces with an event after the next group's block columns, while the main stream factorises group k + 1; the chain
// diagonal block -> panel product is 7 of the phase's 12.3 ms at n = 10000 and needs only those columns.  Same bits
// (the pieces are the tiles of the one launch, cut at tile boundaries), but slower on every shape: 13.4 against 12.3 ms
// (C2), 25.6 against 23.5 (C2 in fp64), 5.8 against 4.6 (C3) -- two cross-stream event waits per group cost more than
// the one-workgroup factorisation gains beside a GEMM that fills the device; profiles/NOTES_r05.md.)
template <typename T>
void cholesky_lower(T *G, size_t ldg, int n, T *W, size_t ldw, hipStream_t s) {
  constexpr int NB = CholBlock<T>::NB;
  // (measured at n = 10000, Cholesky phase in ms for groups of 1 | 2 | 4: fp32, 128-wide panels 13.5 | 12.3 | 12.2;
  // fp64, 64-wide panels 29.1 | 24.7 | 23.6; fp32 at n = 5000 4.6 | 4.6 | 4.9)
  constexpr int kCholGroup = sizeof(T) == 8 ? 4 : POGS_CHOL_GROUP_F32;
  // Panels are taken in GROUPS of kCholGroup: inside a group a panel's block column is updated by the
  // group's earlier panels only (left-looking), and the trailing matrix is updated once per group with
  // all of its panels (K = kCholGroup NB).  The trailing update is bound
  // by the read-add-write of the trailing matrix itself -- sum over the panels of (n - k)^2 / 2 elements,
  // 21 GB at n = 10000 in fp32 with 128-wide panels, 83 GB in fp64 with 64-wide ones -- so a group of g panels
  // divides that traffic by g; the chain diagonal block -> panel product is as long as before.
  auto factor_panel = [&](int o, int nb) {   // diagonal block + the panel below it; returns the rows below
    T *Gd = G + static_cast<size_t>(o) * ldg + o;
    T *Wd = W + static_cast<size_t>(o) * ldw + o;
    hipLaunchKernelGGL((potrf_inv_kernel<T, NB, kPotrfTS>), dim3(1), dim3(kPotrfTS * kPotrfTS), 0, s, Gd, ldg, nb, Wd, ldw);
    const int rem = n - o - nb;
    if (rem > 0) {
      T *L21 = G + static_cast<size_t>(o + nb) * ldg + o;
      // L21 <- A21 * inv(L11)^T, in place: one tile column, each workgroup reads
      // its own row block completely before it writes it.
      GemmArgs<T> g1{rem, nb, nb, L21, ldg, Wd, ldw, L21, ldg, static_cast<T>(1), static_cast<T>(0)};
      launch_gemm<T>(false, false, false, g1, s);
    }
    return rem;
  };
  for (int o = 0; o < n;) {
    int done = 0;   // columns of this group factorised so far
    int rem = 0;
    for (int j = 0; j < kCholGroup && o + done < n; ++j) {
      const int oj = o + done;
      const int nbj = (n - oj < NB) ? n - oj : NB;
      if (j > 0) {
        // this panel's block column (all rows from oj down) <- itself - (the group's panels so far) (their rows oj ..)^T
        T *Lg = G + static_cast<size_t>(oj) * ldg + o;
        T *Cj = G + static_cast<size_t>(oj) * ldg + oj;
        GemmArgs<T> gc{n - oj, nbj, done, Lg, ldg, Lg, ldg, Cj, ldg, static_cast<T>(-1), static_cast<T>(1)};
        launch_gemm<T>(false, false, false, gc, s);
      }
      rem = factor_panel(oj, nbj);
      done += nbj;
    }
    if (rem > 0) {
      // A33 <- A33 - [the group's panels] [..]^T (lower tiles), K = the group's columns, adjacent in storage
      const int o3 = o + done;
      T *Lp = G + static_cast<size_t>(o3) * ldg + o;
      T *A33 = G + static_cast<size_t>(o3) * ldg + o3;
      GemmArgs<T> g2{rem, rem, done, Lp, ldg, Lp, ldg, A33, ldg, static_cast<T>(-1), static_cast<T>(1)};
      launch_gemm<T>(false, false, true, g2, s);
    }
    o += done;
  }
}

template <typename T>
void trtri_lower(const T *L, size_t ldg, int n, T *W, size_t ldw, T *tmp, hipStream_t s) {
  constexpr int NB = CholBlock<T>::NB;
  for (long long sz = NB; sz < n; sz *= 2) {
    // pairs (a = [o, o+sz), b = [o+sz, o+2sz)) at o = 0, 2sz, 4sz, ...: all full pairs of a
    // level go out as one batched launch per product, a ragged last pair on its own
    const int na = static_cast<int>(sz);
    int full = 0;
    long long o = 0;
    for (; o + 2 * sz <= n; o += 2 * sz) ++full;
    auto level = [&](long long o0, int nb, int batch) {
      const T *Lba = L + static_cast<size_t>(o0 + sz) * ldg + o0;
      const T *Waa = W + static_cast<size_t>(o0) * ldw + o0;
      const T *Wbb = W + static_cast<size_t>(o0 + sz) * ldw + (o0 + sz);
      T *Wba = W + static_cast<size_t>(o0 + sz) * ldw + o0;
      T *Tba = tmp + static_cast<size_t>(o0 + sz) * ldw + o0;
      // T = L_ba W_aa ; W_ba = -W_bb T
      GemmArgs<T> g1{nb, na, na, Lba, ldg, Waa, ldw, Tba, ldw, static_cast<T>(1), static_cast<T>(0)};
      g1.batch = batch;
      g1.strideA = static_cast<size_t>(2 * sz) * (ldg + 1);
      g1.strideB = static_cast<size_t>(2 * sz) * (ldw + 1);
      g1.strideC = g1.strideB;
      g1.ktri = 1;   // W_aa is lower-triangular
      launch_gemm<T>(false, true, false, g1, s);
      GemmArgs<T> g2{nb, na, nb, Wbb, ldw, Tba, ldw, Wba, ldw, static_cast<T>(-1), static_cast<T>(0)};
      g2.batch = batch;
      g2.strideA = g2.strideB = g2.strideC = g1.strideB;
      g2.ktri = 2;   // W_bb is lower-triangular
      launch_gemm<T>(false, true, false, g2, s);
    };
    if (full > 0) level(0, na, full);
    if (o + sz < n) level(o, static_cast<int>(n - o - sz), 1);
  }
}

template <typename T>
void launch_transpose(const T *in, size_t ld_in, int rows, int cols, T *out, size_t ld_out, hipStream_t s) {
  dim3 grid((cols + 31) / 32, (rows + 31) / 32);
  hipLaunchKernelGGL(transpose_kernel<T>, grid, dim3(256), 0, s, in, ld_in, rows, cols, out, ld_out);
}

template <typename T>
void launch_sum_slabs(const T *in, size_t stride, int nslabs, T *out, size_t ld, int n, hipStream_t s) {
  hipLaunchKernelGGL(sum_slabs_kernel<T>, dim3(n), dim3(256), 0, s, in, stride, nslabs, out, ld, n);
}

template <typename T>
void launch_zero_upper(T *G, size_t ldg, int n, hipStream_t s) {
  hipLaunchKernelGGL(zero_upper_kernel<T>, dim3((n + 255) / 256, n), dim3(256), 0, s, G, ldg, n);
}

namespace {
__host__ __device__ inline size_t packed_lower_offset(int i, size_t ld, size_t *len) {
  const size_t b = static_cast<size_t>(i) / BM;
  const size_t L = (BM * (b + 1) < ld) ? BM * (b + 1) : ld;
  *len = L;
  return static_cast<size_t>(BM) * BM * (b * (b + 1) / 2) + (static_cast<size_t>(i) - BM * b) * L;
}
template <typename T>
__global__ void __launch_bounds__(256) pack_lower_kernel(T *G, size_t ld, int n, T *packed, bool unpack) {
  using V = typename Vec16<T>::type;
  constexpr int VEC = Vec16<T>::N;
  const int i = blockIdx.x;
  size_t len;
  const size_t off = packed_lower_offset(i, ld, &len);
  T *row = G + static_cast<size_t>(i) * ld, *pk = packed + off;
  for (size_t c = static_cast<size_t>(threadIdx.x) * VEC; c < len; c += 256 * VEC) {   // len, ld: multiples of VEC
    if (unpack) *reinterpret_cast<V *>(row + c) = *reinterpret_cast<const V *>(pk + c);
    else *reinterpret_cast<V *>(pk + c) = *reinterpret_cast<const V *>(row + c);
  }
}
}  // namespace

size_t packed_lower_count(int n, size_t ld) {
  if (n <= 0) return 0;
  size_t len;
  const size_t off = packed_lower_offset(n - 1, ld, &len);
  return off + len;
}

template <typename T>
void launch_pack_lower(T *G, size_t ld, int n, T *packed, bool unpack, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(pack_lower_kernel<T>, dim3(n), dim3(256), 0, s, G, ld, n, packed, unpack);
}

template <typename T>
void launch_add_diag(T *G, size_t ldg, int n, T v, hipStream_t s) {
  hipLaunchKernelGGL(add_diag_kernel<T>, dim3((n + 255) / 256), dim3(256), 0, s, G, ldg, n, v);
}

#define POGS_INST(T)                                                                           \
  template void launch_gemm<T>(bool, bool, bool, const GemmArgs<T> &, hipStream_t);            \
  template void cholesky_lower<T>(T *, size_t, int, T *, size_t, hipStream_t);                 \
  template void trtri_lower<T>(const T *, size_t, int, T *, size_t, T *, hipStream_t);         \
  template void launch_transpose<T>(const T *, size_t, int, int, T *, size_t, hipStream_t);    \
  template void launch_sum_slabs<T>(const T *, size_t, int, T *, size_t, int, hipStream_t);       \
  template void launch_zero_upper<T>(T *, size_t, int, hipStream_t);                           \
  template void launch_pack_lower<T>(T *, size_t, int, T *, bool, hipStream_t);                  \
  template void launch_add_diag<T>(T *, size_t, int, T, hipStream_t);
POGS_INST(float)
POGS_INST(double)
#undef POGS_INST

}  // namespace pogs_amd
