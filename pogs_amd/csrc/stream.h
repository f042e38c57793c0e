// Row-streaming kernel over a dense row-major matrix: the one HBM-bound kernel
// behind every pass over A (and over the triangular solve factors).
//
// For each block of R rows a workgroup
//   (1) loads the rows with 16-byte coalesced loads into registers (once),
//   (2) DOT: forms the R row dot-products with a register-resident vector,
//       reduces them wave -> workgroup, and hands each to a per-row functor
//       (which may store it, update row-local ADMM state, accumulate scalars),
//   (3) ACC: accumulates u_r * row_r into register-resident column sums,
//       where u_r is the functor's return value.
// So y = A x, x = A^T y and the fused "y = f(A x); x' = A^T y" all read A from
// HBM exactly once.  The reference does each of these as separate cblas_?gemv
// calls (src/cpu/matrix/matrix_dense.cpp:93-113); fusing them halves the
// matrix traffic of Sinkhorn-Knopp (equil_helper.h:149-163), the power
// iteration (equil_helper.h:121-123) and the exact-residual evaluation
// (pogs.cpp:352-376).
//
// Column sums are written per workgroup to `col_partials` and combined by
// reduce_cols (fixed order, no atomics => bit-reproducible).
#pragma once
#include <atomic>
#include <cstdlib>
#include <hip/hip_runtime.h>

#include "common.h"
#include "reduce.h"

namespace pogs_amd {

// Load of the matrix stream: every element is used once per pass, so it is marked
// non-temporal (POGS_AMD_NT_LOADS=0 at compile time falls back to plain loads).
#ifndef POGS_AMD_NT_LOADS
#define POGS_AMD_NT_LOADS 1
#endif
template <typename V, typename T>
__device__ __forceinline__ V stream_load(const T *p) {
#if POGS_AMD_NT_LOADS
  typedef unsigned int raw16 __attribute__((ext_vector_type(4)));
  static_assert(sizeof(V) == 16, "16-byte vectors");
  const raw16 raw = __builtin_nontemporal_load(reinterpret_cast<const raw16 *>(p));
  V out;
  __builtin_memcpy(&out, &raw, 16);
  return out;
#else
  return *reinterpret_cast<const V *>(p);
#endif
}

enum Tri : int { kFull = 0, kLower = 1, kUpper = 2 };

template <typename T>
struct StreamArgs {
  const T *A;              // row-major, leading dimension lda (multiple of 16 B)
  size_t lda;
  int m;                   // rows
  int n_pad;               // columns rounded up to the 16-byte vector width
  const T *xin;            // DOT: vector of length n_pad
  const T *xin_add;        // DOT: optional addend (x = xin * scale + xin_add)
  const double *xin_nrm2;  // DOT: optional device scalar; scale = 1/sqrt(*xin_nrm2)
  T *col_partials;         // ACC: [gridDim.x][n_pad]
  double *scalar_partials; // Op::NS > 0: [gridDim.x][Op::NS]
  int col0 = 0;            // first column of the window this launch covers (chunked wide rows)
  T *xl_scratch = nullptr; // chunked plans: (chunks + 1) * m elements
};

struct StreamPlan {
  int tpb = 0;   // threads per workgroup: 64, 256, 512 or 1024
  int nv = 0;    // 16-byte vectors per thread per row
  int grid_max = 0;    // most workgroups any launch of this plan uses (sizes the partial-sum buffers)
  int grid_dot = 0;    // single-dot kernels (stream_rows_kernel): the workgroups resident at once
  int num_cu = 0;      // (256-thread plans) the device's CUs: the one-pass kernel's grid is a multiple of it, per form
  bool ok = false;
  // rows wider than one register tile: the pass is run window by window (tpb * nv vectors of
  // columns each) -- column sums window-wise, row dots as partial dots that a small kernel adds
  // before the row functor runs; a fused row-dot + column-sum pass becomes those two in turn
  bool xl = false;
  // one-pass kernel (ND > 0): workgroups per CU of the NEXT launches when > 0 -- the solver sets it per solve
  // (dense_iter.h: load_problem).  At 256 x 5 the kernel is built for three per CU, which the logistic row functor
  // needs (its serial chain is exposed at two: 0.76 against 0.63 ms per pass at C3) and which by itself costs 8 %:
  // solves whose prox is a few operations run two per CU (0.622 against 0.655 ms; profiles/NOTES_r06.md section 4)
  int bpc_override = 0;
  bool pf = false;   // 256 x 5, fp32, slow row functor: stream_rows2_pf_kernel (set per solve, dense_iter.h)
};
template <typename T> inline int stream_window_cols(const StreamPlan &p) { return p.tpb * p.nv * Vec16<T>::N; }
template <typename T> inline int stream_windows(const StreamPlan &p, int n_pad) {
  const int w = stream_window_cols<T>(p);
  return p.xl ? (n_pad + w - 1) / w : 1;
}
// scratch elements a chunked plan needs for a matrix with `rows` rows
template <typename T> inline size_t stream_xl_scratch(const StreamPlan &p, int rows, int n_pad) {
  return p.xl ? static_cast<size_t>(stream_windows<T>(p, n_pad) + 1) * rows : 0;
}

// The workgroup shapes (threads, 16-byte vectors per thread and row) the streaming kernels are
// built for.  A dense solver uses exactly one of them (or the two window shapes) for all its
// passes, so pogs_amd/build.py compiles dense_plan.hip once per entry of this list and
// arithmetic type -- one small code object each instead of one 4 MB object per type, which is
// what the first launch of a process has to load (~3 ms per MB).  Keep the list in one line per
// the X(tpb, nv) form: build.py reads it.
#define POGS_STREAM_PLANS(X) \
  X(64, 1) X(64, 2) X(64, 4) X(256, 2) X(256, 3) X(256, 4) X(256, 5) X(256, 6) X(256, 8) X(256, 10) \
  X(512, 6) X(512, 8) X(512, 10) X(1024, 6) X(1024, 8)
// Which shapes a translation unit instantiates.
struct AllPlans {
  static constexpr bool has(int, int) { return true; }
  static constexpr bool windows = true;
};
template <int TPB, int NV>
struct OnePlan {
  static constexpr bool has(int t, int v) { return t == TPB && v == NV; }
  static constexpr bool windows = false;
};
struct WindowPlans {   // StreamPlan::xl: the two window shapes only
  static constexpr bool has(int, int) { return false; }
  static constexpr bool windows = true;
};

// Workgroups per CU the one-pass kernel (stream_rows2_kernel, ND > 0) is launched with: the plan's grid_max AND the
// kernel's register budget (__launch_bounds__'s second argument) -- at 256 x 5 the kernel sits at the edge of the 168
// VGPRs that three workgroups per CU leave a wavefront, and without the bound a change of a few registers silently
// costs a resident workgroup (measured in round 5: 174 VGPRs -> two per CU -> C3's pass 0.63 -> 0.88 ms).
#ifndef POGS_C3_LEAN_BLOCKS   // (experiment switches, whole-library builds with POGS_AMD_EXTRA_FLAGS: workgroups per CU at 256 x 5
#define POGS_C3_LEAN_BLOCKS 3 //  of the one-dot / one-accumulator form and of the full one)
#endif
#ifndef POGS_C3_FULL_BLOCKS
#define POGS_C3_FULL_BLOCKS 3
#endif
constexpr int stream2_blocks_per_cu(int tpb, int nv, int nd = 2, int na = 2) {
  return tpb == 64 ? 8
         : tpb == 256 ? (nv == 5 ? ((nd == 1 && na == 1) ? POGS_C3_LEAN_BLOCKS : POGS_C3_FULL_BLOCKS) : nv <= 2 ? 3 : 2)
                      : 1;
}
// (hipcc reads the second argument of __launch_bounds__ as wavefronts per SIMD, not workgroups per CU: 8 x 64 threads
// per CU are 2 per SIMD -- passed as 8 it bounded the 64-thread kernels to 64 VGPRs and they spilled)
constexpr int stream2_waves_per_simd(int tpb, int nv, int nd = 2, int na = 2) {
  return (stream2_blocks_per_cu(tpb, nv, nd, na) * (tpb / 64) + 3) / 4;
}

// Chooses the workgroup shape for rows of n_pad elements.
template <typename T>
inline StreamPlan make_stream_plan(int n_pad, int num_cu) {
  constexpr int VEC = Vec16<T>::N;
  const int vpr = n_pad / VEC;
  StreamPlan p;
  // <= 10 vectors per thread wherever possible: that is what the one-pass kernel's register
  // budget allows (stream2_supported), so 512-thread workgroups (one per CU, 256 VGPRs) take
  // rows of 2561..5120 vectors (fp32 n <= 20480, fp64 n <= 10240) before the 1024-thread shapes
  static const int nv64[] = {1, 2, 4};
  static const int nv256[] = {2, 3, 4, 5, 6, 8, 10};
  static const int nv512[] = {6, 8, 10};
  static const int nv1024[] = {6, 8};
  const char *xl_env = std::getenv("POGS_AMD_XL_LIMIT");
  const bool forced_xl = xl_env && vpr > std::atoi(xl_env);
  for (int nv : nv64)
    if (!forced_xl && vpr <= 64 * nv) { p.tpb = 64; p.nv = nv; p.grid_max = p.grid_dot = num_cu * 8; p.ok = true; return p; }
  for (int nv : nv256)
    if (!forced_xl && vpr <= 256 * nv) {
      p.tpb = 256; p.nv = nv; p.grid_dot = num_cu * 2;
      // at nv = 5 the one-pass iteration kernel takes two rows per step (stream2_rows_c), needs 167
      // VGPRs and fits three workgroups per CU: a slow row functor (logistic prox) hides better
      p.num_cu = num_cu;
      p.grid_max = num_cu * std::max(stream2_blocks_per_cu(256, nv), stream2_blocks_per_cu(256, nv, 1, 1));   // (nv <= 2: 8 rows per step, 158 VGPRs; 500000 x 2000: 0.694 -> 0.657 ms)
      p.ok = true;
      return p;
    }
  for (int nv : nv512)
    if (!forced_xl && vpr <= 512 * nv) { p.tpb = 512; p.nv = nv; p.grid_max = p.grid_dot = num_cu; p.ok = true; return p; }
  // POGS_AMD_XL_LIMIT=<vectors> (testing aid): rows wider than that take the windowed form below,
  // with small windows so that small test matrices span several of them
  int limit = 1024 * 8;
  bool small_windows = false;
  if (const char *ev = std::getenv("POGS_AMD_XL_LIMIT")) { limit = std::atoi(ev); small_windows = true; }
  if (vpr <= limit) {
    for (int nv : nv1024)
      if (vpr <= 1024 * nv) { p.tpb = 1024; p.nv = nv; p.grid_max = p.grid_dot = num_cu; p.ok = true; return p; }
  }
  p.xl = true;
  p.ok = true;
  if (small_windows) { p.tpb = 64; p.nv = 2; p.grid_max = p.grid_dot = num_cu * 8; }
  else { p.tpb = 256; p.nv = 8; p.grid_max = p.grid_dot = num_cu * 2; }
  return p;
}

namespace dev {

// The roundings of the row dots and column sums are WRITTEN OUT (round 6): this translation unit is compiled with
// -ffp-contract=off (pogs_amd/build.py), so a product is fused with a sum exactly where an fma() says so and nowhere
// else -- which products of a dot get fused no longer follows instruction selection (round 5: an edit to the wavefront
// reduction moved a 33 x 40001 problem by 1e-4 through the dots of OTHER kernels).  One form for every kernel:
//   vdot   a.x b.x  rounded, then the y, z, w products fused onto it in that order (what -ffp-contract=fast made of the
//          first dot of the one-pass kernel, the one on the trajectory; its second dot used to run unfused);
//   vfma   acc + u a  fused per element;   vscale_add  a s + b  fused per element;   vadd  plain adds.
__device__ __forceinline__ float vdot(const float4 &a, const float4 &b) {
  return fma_(a.w, b.w, fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)));
}
__device__ __forceinline__ double vdot(const double2 &a, const double2 &b) {
  return fma_(a.y, b.y, a.x * b.x);
}
__device__ __forceinline__ void vfma(float4 &acc, float u, const float4 &a) {
  acc.x = fma_(u, a.x, acc.x); acc.y = fma_(u, a.y, acc.y); acc.z = fma_(u, a.z, acc.z); acc.w = fma_(u, a.w, acc.w);
}
__device__ __forceinline__ void vfma(double2 &acc, double u, const double2 &a) {
  acc.x = fma_(u, a.x, acc.x); acc.y = fma_(u, a.y, acc.y);
}
__device__ __forceinline__ void vadd(float4 &acc, const float4 &a) {
  acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
}
__device__ __forceinline__ void vadd(double2 &acc, const double2 &a) {
  acc.x += a.x; acc.y += a.y;
}
__device__ __forceinline__ float4 vsq(const float4 &a) {
  return make_float4(a.x * a.x, a.y * a.y, a.z * a.z, a.w * a.w);
}
__device__ __forceinline__ double2 vsq(const double2 &a) { return make_double2(a.x * a.x, a.y * a.y); }
__device__ __forceinline__ float4 vscale_add(const float4 &a, float s, const float4 &b) {
  return make_float4(fma_(a.x, s, b.x), fma_(a.y, s, b.y), fma_(a.z, s, b.z), fma_(a.w, s, b.w));
}
__device__ __forceinline__ double2 vscale_add(const double2 &a, double s, const double2 &b) {
  return make_double2(fma_(a.x, s, b.x), fma_(a.y, s, b.y));
}
template <typename V> __device__ __forceinline__ V vzero();
template <> __device__ __forceinline__ float4 vzero<float4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <> __device__ __forceinline__ double2 vzero<double2>() { return make_double2(0.0, 0.0); }

}  // namespace dev

template <typename T, int TPB, int NV, int R, bool DOT, bool ACC, bool SQ, int TRI, typename Op>
__global__ void __launch_bounds__(TPB) stream_rows_kernel(StreamArgs<T> a, Op op) {
  using V = typename Vec16<T>::type;
  constexpr int VEC = Vec16<T>::N;
  constexpr int NW = TPB / 64;
  constexpr int NS = Op::NS > 0 ? Op::NS : 1;
  __shared__ T s_part[2 * R * NW];
  __shared__ T s_u[2 * R];
  __shared__ double s_red[NS * NW];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

  V xv[NV];
  V acc[NV];
  if (DOT) {
    T sc = 1;
    if (a.xin_nrm2) sc = static_cast<T>(1.0 / sqrt(*a.xin_nrm2));
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int col = a.col0 + (v * TPB + t) * VEC;
      V x = dev::vzero<V>();
      if (col < a.n_pad) {
        x = *reinterpret_cast<const V *>(a.xin + col);
        V add = dev::vzero<V>();
        if (a.xin_add) add = *reinterpret_cast<const V *>(a.xin_add + col);
        x = dev::vscale_add(x, sc, add);
      }
      xv[v] = x;
    }
  }
  if (ACC) {
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = dev::vzero<V>();
  }
  double sacc[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) sacc[k] = 0.0;

  const int nblk = (a.m + R - 1) / R;
  int slot = 0;
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x, slot ^= 1) {
    const int row0 = blk * R;
    V av[R][NV];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = row0 + r;
      const T *rp = a.A + static_cast<size_t>(row) * a.lda;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int col = a.col0 + (v * TPB + t) * VEC;
        bool ok = (col < a.n_pad) && (row < a.m);
        if (TRI == kLower) ok = ok && (col <= row);
        if (TRI == kUpper) ok = ok && (col + VEC - 1 >= row);
        V val = dev::vzero<V>();
        // the pass over A is read once per iteration (non-temporal); a triangular factor is 1/20
        // of that and comes back every iteration: plain loads, so it may stay in the Infinity Cache
        if (ok) val = (TRI == kFull) ? stream_load<V>(rp + col) : *reinterpret_cast<const V *>(rp + col);
        av[r][v] = SQ ? dev::vsq(val) : val;
      }
    }
    if (DOT) {
      T p[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        T s = 0;
#pragma unroll
        for (int v = 0; v < NV; ++v) s += dev::vdot(av[r][v], xv[v]);
        p[r] = dev::wave_sum(s);
      }
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) s_part[(slot * R + r) * NW + wave] = p[r];
      }
      __syncthreads();
      if (t < R) {
        const int row = row0 + t;
        T uval = 0;
        if (row < a.m) {
          T dot = 0;
#pragma unroll
          for (int w = 0; w < NW; ++w) dot += s_part[(slot * R + t) * NW + w];
          uval = op.row(row, dot, sacc);
        }
        if (ACC) s_u[slot * R + t] = uval;
      }
      if (ACC) __syncthreads();
    }
    if (ACC) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        T u;
        if (DOT) {
          u = s_u[slot * R + r];
        } else {
          u = (row0 + r < a.m) ? op.u(row0 + r) : static_cast<T>(0);
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) dev::vfma(acc[v], u, av[r][v]);
      }
    }
  }
  if (ACC) {
    T *out = a.col_partials + static_cast<size_t>(blockIdx.x) * a.n_pad;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int col = a.col0 + (v * TPB + t) * VEC;
      if (col < a.n_pad) *reinterpret_cast<V *>(out + col) = acc[v];
    }
  }
  if (Op::NS > 0) {
    __syncthreads();
    dev::block_sum<NS, TPB>(sacc, s_red);
    if (t == 0) {
#pragma unroll
      for (int k = 0; k < NS; ++k) a.scalar_partials[static_cast<size_t>(blockIdx.x) * NS + k] = sacc[k];
    }
  }
}

// ---------------------------------------------------------------------------
// stream_tri_kernel: the dot + column-sum pass over a LOWER TRIANGLE (the one sweep over W = L^-1
// of the x update, the symmetric product G x of the norm estimate).  In stream_rows_kernel a row
// of 100 entries costs a workgroup step -- a load latency, two barriers -- like a row of 10000, and
// half the steps of a triangle are spent on a quarter of its bytes.  Here the rows that end inside
// the first quarter (half) of the register tile's vectors are taken 4R (2R) at a time, in the
// registers the absent vectors leave free: 31 % fewer steps, more bytes in flight where the rows are
// short.  Same functor contract, same partial-sum layout, same grid as stream_rows_kernel.
// ---------------------------------------------------------------------------
template <typename T, int TPB, int NV, int RP, int NVP, int RMAX, int NSX, typename Op>
__device__ __forceinline__ void stream_tri_phase(const StreamArgs<T> &a, const Op &op, int row_begin, int row_end,
                                                 int blk_base, const typename Vec16<T>::type (&xv)[NV],
                                                 typename Vec16<T>::type (&acc)[NV], double (&sacc)[NSX], T *s_part,
                                                 T *s_u, int &slot) {
  using V = typename Vec16<T>::type;
  constexpr int VEC = Vec16<T>::N;
  constexpr int NW = TPB / 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nblk = (row_end - row_begin + RP - 1) / RP;
  // step b of this phase is step blk_base + b of the sweep; steps are dealt round-robin over the sweep
  int first = static_cast<int>(blockIdx.x) - blk_base % static_cast<int>(gridDim.x);
  if (first < 0) first += gridDim.x;
  for (int blk = first; blk < nblk; blk += gridDim.x, slot ^= 1) {
    const int row0 = row_begin + blk * RP;
    V av[RP][NVP];
#pragma unroll
    for (int r = 0; r < RP; ++r) {
      const int row = row0 + r;
      const T *rp = a.A + static_cast<size_t>(row) * a.lda;
#pragma unroll
      for (int v = 0; v < NVP; ++v) {
        const int col = (v * TPB + t) * VEC;
        V val = dev::vzero<V>();
        if (row < row_end && col <= row) val = *reinterpret_cast<const V *>(rp + col);
        av[r][v] = val;
      }
    }
    T p[RP];
#pragma unroll
    for (int r = 0; r < RP; ++r) {
      T s = 0;
#pragma unroll
      for (int v = 0; v < NVP; ++v) s += dev::vdot(av[r][v], xv[v]);
      p[r] = dev::wave_sum(s);
    }
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < RP; ++r) s_part[(slot * RMAX + r) * NW + wave] = p[r];
    }
    __syncthreads();
    if (t < RP) {
      const int row = row0 + t;
      T uval = 0;
      if (row < row_end) {
        T dot = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) dot += s_part[(slot * RMAX + t) * NW + w];
        uval = op.row(row, dot, sacc);
      }
      s_u[slot * RMAX + t] = uval;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RP; ++r) {
      const T u = s_u[slot * RMAX + r];
#pragma unroll
      for (int v = 0; v < NVP; ++v) dev::vfma(acc[v], u, av[r][v]);
    }
  }
}

template <int NV> struct TriPhases {
  static constexpr int q = NV / 4;                       // vectors of the rows taken 4R at a time
  static constexpr int h = (NV / 2 > q) ? NV / 2 : 0;    // ... 2R at a time
};
// first row of the half (hq = 1) / full (hq = 2) phase, and the steps before it
template <int TPB, int NV, int VEC, int R>
__host__ __device__ inline void tri_phase_bounds(int m, int (&row)[4], int (&base)[3]) {
  constexpr int q = TriPhases<NV>::q, h = TriPhases<NV>::h;
  row[0] = 0;
  row[1] = q > 0 ? (q * TPB * VEC < m ? q * TPB * VEC : m) : 0;
  row[2] = h > 0 ? (h * TPB * VEC < m ? h * TPB * VEC : m) : row[1];
  row[3] = m;
  base[0] = 0;
  base[1] = (row[1] - row[0] + 4 * R - 1) / (4 * R);
  base[2] = base[1] + (row[2] - row[1] + 2 * R - 1) / (2 * R);
}

template <typename T, int TPB, int NV, int R, typename Op>
__global__ void __launch_bounds__(TPB) stream_tri_kernel(StreamArgs<T> a, Op op) {
  using V = typename Vec16<T>::type;
  constexpr int VEC = Vec16<T>::N;
  constexpr int NW = TPB / 64;
  constexpr int NS = Op::NS > 0 ? Op::NS : 1;
  constexpr int NVQ = TriPhases<NV>::q, NVH = TriPhases<NV>::h;
  constexpr int RMAX = NVQ > 0 ? 4 * R : (NVH > 0 ? 2 * R : R);
  __shared__ T s_part[2 * RMAX * NW];
  __shared__ T s_u[2 * RMAX];
  __shared__ double s_red[NS * NW];
  const int t = threadIdx.x;

  V xv[NV];
  V acc[NV];
  T sc = 1;
  if (a.xin_nrm2) sc = static_cast<T>(1.0 / sqrt(*a.xin_nrm2));
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int col = (v * TPB + t) * VEC;
    V x = dev::vzero<V>();
    if (col < a.n_pad) {
      x = *reinterpret_cast<const V *>(a.xin + col);
      V add = dev::vzero<V>();
      if (a.xin_add) add = *reinterpret_cast<const V *>(a.xin_add + col);
      x = dev::vscale_add(x, sc, add);
    }
    xv[v] = x;
    acc[v] = dev::vzero<V>();
  }
  double sacc[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) sacc[k] = 0.0;

  int row[4], base[3];
  tri_phase_bounds<TPB, NV, VEC, R>(a.m, row, base);
  int slot = 0;
  if constexpr (NVQ > 0)
    stream_tri_phase<T, TPB, NV, 4 * R, NVQ, RMAX>(a, op, row[0], row[1], base[0], xv, acc, sacc, s_part, s_u, slot);
  if constexpr (NVH > 0)
    stream_tri_phase<T, TPB, NV, 2 * R, NVH, RMAX>(a, op, row[1], row[2], base[1], xv, acc, sacc, s_part, s_u, slot);
  stream_tri_phase<T, TPB, NV, R, NV, RMAX>(a, op, row[2], row[3], base[2], xv, acc, sacc, s_part, s_u, slot);

  T *out = a.col_partials + static_cast<size_t>(blockIdx.x) * a.n_pad;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int col = (v * TPB + t) * VEC;
    if (col < a.n_pad) *reinterpret_cast<V *>(out + col) = acc[v];
  }
  if (Op::NS > 0) {
    __syncthreads();
    dev::block_sum<NS, TPB>(sacc, s_red);
    if (t == 0) {
#pragma unroll
      for (int k = 0; k < NS; ++k) a.scalar_partials[static_cast<size_t>(blockIdx.x) * NS + k] = sacc[k];
    }
  }
}
// (Measured and not kept: one row per step for the dot + column-sum form at 256 x 10 -- 154 VGPRs,
// three workgroups per CU.  The triangular W sweep stayed at 40.7 us, the full passes got slower
// (Sinkhorn-Knopp pass 614 -> 653 us) and the second stage had 50 % more partials to add.)
// Rows per workgroup step by mode and workgroup size (register budget: the row
// tile costs 4*R*NV VGPRs, the x / column-sum vectors 4*NV each).
template <bool DOT, bool ACC, int TPB>
struct RowsPerStep {
  static constexpr int value = (TPB == 1024) ? ((DOT && ACC) ? 1 : 2) : ((DOT && ACC) ? 2 : 4);
};

// Number of workgroups a launch of this plan uses for m rows.
template <bool DOT, bool ACC>
inline int stream_grid(const StreamPlan &p, int m) {
  const int R = (p.tpb == 1024) ? ((DOT && ACC) ? 1 : 2) : ((DOT && ACC) ? 2 : 4);
  const int nblk = (m + R - 1) / R;
  return nblk < p.grid_dot ? (nblk > 0 ? nblk : 1) : p.grid_dot;
}

// ---- windowed form for rows wider than one register tile (StreamPlan::xl) -------------------
template <typename T>
struct XlStoreDotOp {   // partial row dots of one window
  static constexpr int NS = 0;
  T *out;
  template <int N>
  __device__ __forceinline__ T row(int i, T dot, double (&)[N]) const {
    out[i] = dot;
    return 0;
  }
  __device__ __forceinline__ T u(int) const { return 0; }
};
template <typename T>
struct XlVecUOp {       // column-sum coefficients from a vector
  static constexpr int NS = 0;
  const T *uvec;
  template <int N>
  __device__ __forceinline__ T row(int, T, double (&)[N]) const { return 0; }
  __device__ __forceinline__ T u(int i) const { return uvec[i]; }
};
// dot_i = sum of the window partials (window order), then the row functor; its return value is
// kept when a column-sum pass follows.  Scalar partials in the layout of the plain kernel.
template <typename T, typename Op, bool KEEP_U>
__global__ void __launch_bounds__(256) xl_apply_rows_kernel(const T *part, int m, int nwin, Op op, T *uvec,
                                                            double *scalar_partials) {
  constexpr int NS = Op::NS > 0 ? Op::NS : 1;
  __shared__ double s_red[NS * 4];
  double sacc[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) sacc[k] = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256) {
    T dot = part[i];
    for (int c = 1; c < nwin; ++c) dot += part[static_cast<size_t>(c) * m + i];
    const T u = op.row(i, dot, sacc);
    if (KEEP_U) uvec[i] = u;
  }
  if (Op::NS > 0) {
    dev::block_sum<NS, 256>(sacc, s_red);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < NS; ++k) scalar_partials[static_cast<size_t>(blockIdx.x) * NS + k] = sacc[k];
    }
  }
}

template <typename T, bool DOT, bool ACC, bool SQ, int TRI, typename Tag, typename Op>
void launch_stream_plain(const StreamPlan &p, const StreamArgs<T> &a, const Op &op, hipStream_t s, int grid) {
#define POGS_STREAM_CASE(TPB_, NV_)                                                             \
  if constexpr (Tag::has(TPB_, NV_)) {                                                          \
    if (p.tpb == TPB_ && p.nv == NV_) {                                                         \
      constexpr int R_ = RowsPerStep<DOT, ACC, TPB_>::value;                                    \
      /* (1024, 8): 128 VGPRs per thread -- the phased kernel spills 26-45 of them, the plain one 17-21 */ \
      if constexpr (DOT && ACC && !SQ && TRI == kLower && !(TPB_ == 1024 && NV_ == 8)) {        \
        if (a.col0 == 0) {                                              \
          hipLaunchKernelGGL((stream_tri_kernel<T, TPB_, NV_, R_, Op>), dim3(grid), dim3(TPB_), 0, s, a, op); \
          return;                                                                               \
        }                                                                                       \
      }                                                                                         \
      hipLaunchKernelGGL((stream_rows_kernel<T, TPB_, NV_, R_, DOT, ACC, SQ, TRI, Op>),         \
                         dim3(grid), dim3(TPB_), 0, s, a, op);                                  \
      return;                                                                                   \
    }                                                                                           \
  }
  POGS_STREAM_PLANS(POGS_STREAM_CASE)
#undef POGS_STREAM_CASE
  throw Error("no stream kernel instance for this plan in this translation unit");
}

// the two window shapes of a windowed plan only (keeps the number of kernel variants down)
template <typename T, bool DOT, bool ACC, bool SQ, int TRI, typename Tag, typename Op>
void launch_stream_window(const StreamPlan &p, const StreamArgs<T> &a, const Op &op, hipStream_t s, int grid) {
  if constexpr (Tag::windows) {
    if (p.tpb == 256 && p.nv == 8) {
      hipLaunchKernelGGL((stream_rows_kernel<T, 256, 8, RowsPerStep<DOT, ACC, 256>::value, DOT, ACC, SQ, TRI, Op>),
                         dim3(grid), dim3(256), 0, s, a, op);
      return;
    }
    if (p.tpb == 64 && p.nv == 2) {
      hipLaunchKernelGGL((stream_rows_kernel<T, 64, 2, RowsPerStep<DOT, ACC, 64>::value, DOT, ACC, SQ, TRI, Op>),
                         dim3(grid), dim3(64), 0, s, a, op);
      return;
    }
  }
  throw Error("no window kernel instance for this plan in this translation unit");
}

template <typename T, bool DOT, bool ACC, bool SQ, int TRI, typename Tag = AllPlans, typename Op>
void launch_stream(const StreamPlan &p, const StreamArgs<T> &a, const Op &op, hipStream_t s) {
  POGS_CHECK(p.ok, "matrix too wide for the row-streaming kernel");
  const int grid = stream_grid<DOT, ACC>(p, a.m);
  if (!p.xl) {
    launch_stream_plain<T, DOT, ACC, SQ, TRI, Tag, Op>(p, a, op, s, grid);
    return;
  }
  POGS_CHECK(a.xl_scratch != nullptr, "windowed pass without scratch");
  const int w = stream_window_cols<T>(p);
  const int nwin = stream_windows<T>(p, a.n_pad);
  T *part = a.xl_scratch, *uvec = a.xl_scratch + static_cast<size_t>(nwin) * a.m;
  StreamArgs<T> aw = a;
  if (DOT) {
    const int gdot = stream_grid<true, false>(p, a.m);
    for (int c = 0; c < nwin; ++c) {
      aw.col0 = c * w;
      launch_stream_window<T, true, false, SQ, TRI, Tag, XlStoreDotOp<T>>(p, aw, XlStoreDotOp<T>{part + static_cast<size_t>(c) * a.m},
                                                                     s, gdot);
    }
    hipLaunchKernelGGL((xl_apply_rows_kernel<T, Op, ACC>), dim3(grid), dim3(256), 0, s, part, a.m, nwin, op, uvec,
                       a.scalar_partials);
  }
  if (ACC) {
    for (int c = 0; c < nwin; ++c) {
      aw.col0 = c * w;
      if (DOT) launch_stream_window<T, false, true, SQ, TRI, Tag, XlVecUOp<T>>(p, aw, XlVecUOp<T>{uvec}, s, grid);
      else launch_stream_window<T, false, true, SQ, TRI, Tag, Op>(p, aw, op, s, grid);
    }
  }
}

// ---------------------------------------------------------------------------
// stream_rows2_kernel: the same pass with up to two dot products (two
// register-resident x vectors) and up to two column-sum accumulators per row.
// It is what makes a whole ADMM iteration a single pass over A:
//   dot[0] = A x_{k+1}  (the projection's y),  dot[1] = A x12_k  (exact primal residual),
//   acc[0] += yhat_{k+1} * row  (next iteration's A^T yhat, speculative),
//   acc[1] += u2_{k+1}  * row   (next iteration's exact dual residual).
// ND = 0 is the column-sum-only form (two u values per row from the functor).
// ---------------------------------------------------------------------------
template <typename T>
struct StreamArgs2 {
  const T *A;
  size_t lda;
  int m, n_pad;
  const T *xin0, *xin1;      // ND dot vectors (length n_pad)
  T *col_partials0, *col_partials1;  // NA accumulators: [gridDim.x][n_pad] each
  double *scalar_partials;   // [gridDim.x][Op::NS]
};

// (Measured and not kept: requesting the NEXT row tile before the current one is reduced, so that
// the memory pipe does not idle through a slow row functor -- the logistic prox, one lane per row,
// 0.35 us per step.  At the C3 shape the R * NV extra vector registers take the kernel from three
// to two workgroups per CU and the pass from 0.692 to 0.715 ms: resident workgroups hide the
// functor better than a prefetch does.)
// (Measured in round 5 and not kept either: the NEXT tile travelling global -> LDS by DMA (global_load_lds_dwordx4,
// every wavefront staging and reading back its own lines: no extra barrier, no registers) while this one is worked on.
// The 40 KB staging buffer leaves room for two workgroups per CU instead of three, i.e. 80 KB in flight per CU instead
// of up to 120: C3's pass 0.657 -> 0.725 ms.  The register file (512 KB per CU) holds more bytes in flight than the
// LDS (160 KB) can; profiles/NOTES_r05.md.)
template <typename T, int TPB, int NV, int R, int ND, int NA, typename Op>
__global__ void __launch_bounds__(TPB, (ND > 0 ? stream2_waves_per_simd(TPB, NV, ND, NA) : 1)) stream_rows2_kernel(StreamArgs2<T> a, Op op) {
  using V = typename Vec16<T>::type;
  constexpr int VEC = Vec16<T>::N;
  constexpr int NW = TPB / 64;
  constexpr int NS = Op::NS > 0 ? Op::NS : 1;
  constexpr int NDD = ND > 0 ? ND : 1;
  __shared__ T s_part[2 * R * NDD * NW];
  __shared__ T s_u[2 * R * NA];
  __shared__ double s_red[NS * NW];
  // the second dot vector lives in LDS (dynamic, n_pad elements): it is re-read per
  // row with conflict-free 16-byte reads and frees 4*NV VGPRs for a second resident wave
  extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
  T *s_x1 = reinterpret_cast<T *>(s_dyn);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

  V xv[NV];
  V acc[NA][NV];
  if (ND > 0) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int col = (v * TPB + t) * VEC;
      xv[v] = (col < a.n_pad) ? *reinterpret_cast<const V *>(a.xin0 + col) : dev::vzero<V>();
      if (ND > 1 && col < a.n_pad) *reinterpret_cast<V *>(s_x1 + col) = *reinterpret_cast<const V *>(a.xin1 + col);
    }
    if (ND > 1) __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < NA; ++q)
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[q][v] = dev::vzero<V>();
  double sacc[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) sacc[k] = 0.0;

  const int nblk = (a.m + R - 1) / R;
  int slot = 0;
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x, slot ^= 1) {
    const int row0 = blk * R;
    V av[R][NV];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = row0 + r;
      const T *rp = a.A + static_cast<size_t>(row) * a.lda;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int col = (v * TPB + t) * VEC;
        V val = dev::vzero<V>();
        if (col < a.n_pad && row < a.m) val = stream_load<V>(rp + col);
        av[r][v] = val;
      }
    }
    // the row functor's own operands (coefficients, y-vectors) are requested now, so
    // their latency overlaps the row tile's instead of following the reduction
    typename Op::Pre pre;
    if (ND > 0 && t < R && row0 + t < a.m) pre = op.prefetch(row0 + t);
    if (ND > 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        T s0 = 0, s1 = 0;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          s0 += dev::vdot(av[r][v], xv[v]);
          if (ND > 1) {
            const int col = (v * TPB + t) * VEC;
            if (col < a.n_pad) s1 += dev::vdot(av[r][v], *reinterpret_cast<const V *>(s_x1 + col));
          }
        }
        s0 = dev::wave_sum(s0);
        if (ND > 1) s1 = dev::wave_sum(s1);
        if (lane == 0) {
          s_part[((slot * R + r) * NDD + 0) * NW + wave] = s0;
          if (ND > 1) s_part[((slot * R + r) * NDD + 1) * NW + wave] = s1;
        }
      }
      __syncthreads();
      if (t < R) {
        const int row = row0 + t;
        T uu[NA];
#pragma unroll
        for (int q = 0; q < NA; ++q) uu[q] = 0;
        if (row < a.m) {
          T dots[NDD];
#pragma unroll
          for (int d = 0; d < ND; ++d) {
            T s = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += s_part[((slot * R + t) * NDD + d) * NW + w];
            dots[d] = s;
          }
          op.row(row, pre, dots, sacc, uu);
        }
#pragma unroll
        for (int q = 0; q < NA; ++q) s_u[(slot * R + t) * NA + q] = uu[q];
      }
      __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      T uu[NA];
      if (ND > 0) {
#pragma unroll
        for (int q = 0; q < NA; ++q) uu[q] = s_u[(slot * R + r) * NA + q];
      } else {
#pragma unroll
        for (int q = 0; q < NA; ++q) uu[q] = 0;
        if (row0 + r < a.m) op.uonly(row0 + r, uu);
      }
#pragma unroll
      for (int q = 0; q < NA; ++q)
#pragma unroll
        for (int v = 0; v < NV; ++v) dev::vfma(acc[q][v], uu[q], av[r][v]);
    }
  }
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    T *out = (q == 0 ? a.col_partials0 : a.col_partials1) + static_cast<size_t>(blockIdx.x) * a.n_pad;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int col = (v * TPB + t) * VEC;
      if (col < a.n_pad) *reinterpret_cast<V *>(out + col) = acc[q][v];
    }
  }
  if (Op::NS > 0) {
    __syncthreads();
    dev::block_sum<NS, TPB>(sacc, s_red);
    if (t == 0) {
#pragma unroll
      for (int k = 0; k < NS; ++k) a.scalar_partials[static_cast<size_t>(blockIdx.x) * NS + k] = sacc[k];
    }
  }
}

// ---------------------------------------------------------------------------
// stream_rows2_pf_kernel: the one-pass iteration kernel (two dots, two accumulators) with the NEXT tile in flight,
// for a SLOW row functor at 256 x 5 (the logistic prox: C3).
// What round 6's phase timing found (scripts/micro/c3_bisect.hip tables 6-8, profiles/NOTES_r06.md section 12): a
// step of stream_rows2_kernel at this shape is  wait for the tile 27 % | dots 33 % | functor 37 % (1.5 us, one
// lane per row) | column sums < 1 %,  and "dots" is not arithmetic: it is the ten dependent 16-byte LDS reads of
// the second dot vector and four six-level ds_bpermute wavefront sums.  Two workgroups per CU stream 8 % faster
// than three, but then nothing covers the functor.  This form:
//   * both dot vectors in registers and the wavefront sums in the vector ALU (wave_sum_valu: same tree, same bits):
//     the dots become ~0.4 us of arithmetic;
//   * the tile of step k + 1 (and its functor's operands) is requested before step k is reduced: 228 VGPRs, two
//     workgroups per CU, and the functor of step k runs under the load of step k + 1;
//   * every tile load UNCONDITIONAL (row and column clamped into the matrix; the dot vectors are zero in the lanes
//     past n_pad, the functor gives u = 0 to rows past m, and the partials of those lanes are never stored).  A
//     guarded load is a branch, and behind a branch the compiler cannot count the loads younger than the tile it is
//     about to use: it emits s_waitcnt vmcnt(0) -- waits for the NEXT tile too -- and the prefetch is gone.  (That is
//     what the round-3 and round-6 prefetching forms measured as "no gain" had done.)
// Measured (same boxes, shipped three-per-CU form -> this): 0.639 -> 0.617, 0.652 -> 0.627 ms over C3's matrix.
// Same arithmetic per row as stream_rows2_kernel; the grid differs (two per CU), so the column partials add the
// rows in another grouping: compared through the usual tolerances, not bit for bit.
// ---------------------------------------------------------------------------
#ifndef POGS_NV5_PREFETCH   // (0: the three-per-CU form for the logistic solves too, as before -- A / B builds)
#define POGS_NV5_PREFETCH 1
#endif
template <typename T, int TPB, int NV, int R, typename Op>
__global__ void __launch_bounds__(TPB, (2 * (TPB / 64) + 3) / 4) stream_rows2_pf_kernel(StreamArgs2<T> a, Op op) {
  using V = typename Vec16<T>::type;
  using Pre = typename Op::Pre;
  constexpr int VEC = Vec16<T>::N;
  constexpr int NW = TPB / 64;
  constexpr int NS = Op::NS > 0 ? Op::NS : 1;
  constexpr int ND = 2, NA = 2;
  __shared__ T s_part[2 * R * ND * NW];
  __shared__ T s_u[2 * R * NA];
  __shared__ double s_red[NS * NW];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

  V xv[NV], x1v[NV];
  V acc[NA][NV];
  int colc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int col = (v * TPB + t) * VEC;
    const bool in = col < a.n_pad;
    colc[v] = in ? col : a.n_pad - VEC;
    xv[v] = in ? *reinterpret_cast<const V *>(a.xin0 + col) : dev::vzero<V>();
    x1v[v] = in ? *reinterpret_cast<const V *>(a.xin1 + col) : dev::vzero<V>();
  }
#pragma unroll
  for (int q = 0; q < NA; ++q)
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[q][v] = dev::vzero<V>();
  double sacc[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) sacc[k] = 0.0;

  const int nblk = (a.m + R - 1) / R;
  int blk = blockIdx.x;
  V cur[R][NV];
  Pre pre_cur;
  if (t < R && blk < nblk && blk * R + t < a.m) pre_cur = op.prefetch(blk * R + t);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = blk * R + r;
    const T *rp = a.A + static_cast<size_t>(row < a.m ? row : a.m - 1) * a.lda;
#pragma unroll
    for (int v = 0; v < NV; ++v) cur[r][v] = stream_load<V>(rp + colc[v]);
  }
  int slot = 0;
  for (; blk < nblk; blk += gridDim.x, slot ^= 1) {
    const int row0 = blk * R;
    // the next tile of this workgroup: its functor's operands first (waiting for them next step then waits for
    // nothing younger), then the rows
    const int nblk_ = blk + gridDim.x, nrow0 = nblk_ * R;
    Pre pre_nxt;
    if (t < R && nblk_ < nblk && nrow0 + t < a.m) pre_nxt = op.prefetch(nrow0 + t);
    V nxt[R][NV];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = nrow0 + r;
      const T *rp = a.A + static_cast<size_t>(row < a.m ? row : a.m - 1) * a.lda;
#pragma unroll
      for (int v = 0; v < NV; ++v) nxt[r][v] = stream_load<V>(rp + colc[v]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      T s0 = 0, s1 = 0;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        s0 += dev::vdot(cur[r][v], xv[v]);
        s1 += dev::vdot(cur[r][v], x1v[v]);
      }
      s0 = dev::wave_sum_valu(s0);
      s1 = dev::wave_sum_valu(s1);
      if (lane == 0) {
        s_part[((slot * R + r) * ND + 0) * NW + wave] = s0;
        s_part[((slot * R + r) * ND + 1) * NW + wave] = s1;
      }
    }
    __syncthreads();
    if (t < R) {
      const int row = row0 + t;
      T uu[NA];
#pragma unroll
      for (int q = 0; q < NA; ++q) uu[q] = 0;
      if (row < a.m) {
        T dots[ND];
#pragma unroll
        for (int d = 0; d < ND; ++d) {
          T s = 0;
#pragma unroll
          for (int w = 0; w < NW; ++w) s += s_part[((slot * R + t) * ND + d) * NW + w];
          dots[d] = s;
        }
        op.row(row, pre_cur, dots, sacc, uu);
      }
#pragma unroll
      for (int q = 0; q < NA; ++q) s_u[(slot * R + t) * NA + q] = uu[q];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      T uu[NA];
#pragma unroll
      for (int q = 0; q < NA; ++q) uu[q] = s_u[(slot * R + r) * NA + q];
#pragma unroll
      for (int q = 0; q < NA; ++q)
#pragma unroll
        for (int v = 0; v < NV; ++v) dev::vfma(acc[q][v], uu[q], cur[r][v]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int v = 0; v < NV; ++v) cur[r][v] = nxt[r][v];
    pre_cur = pre_nxt;
  }
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    T *out = (q == 0 ? a.col_partials0 : a.col_partials1) + static_cast<size_t>(blockIdx.x) * a.n_pad;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int col = (v * TPB + t) * VEC;
      if (col < a.n_pad) *reinterpret_cast<V *>(out + col) = acc[q][v];
    }
  }
  if (Op::NS > 0) {
    __syncthreads();
    dev::block_sum<NS, TPB>(sacc, s_red);
    if (t == 0) {
#pragma unroll
      for (int k = 0; k < NS; ++k) a.scalar_partials[static_cast<size_t>(blockIdx.x) * NS + k] = sacc[k];
    }
  }
}

// Whether the two-dot / two-accumulator kernel fits the register file for this plan
// (row tile 4*R*NV + 2 x vectors 8*NV + 2 accumulators 8*NV VGPRs, R = 1).
inline bool stream2_supported(const StreamPlan &p) {
  if (!p.ok || p.xl) return false;
  if (p.tpb == 1024) return false;  // 128-VGPR budget: would spill
  return p.nv <= 10;
}
// Rows per workgroup step of the two-accumulator kernel: as many as the register file
// allows (4*NV*(R + 1 + 2) + ~70 VGPRs <= 256), so the per-row functor latency (one lane per
// row) is amortised over R rows.
// (one dot product and one accumulator -- the pass without the exact residuals -- leave room for a
// second row at 8 .. 10 vectors per thread: 4*NV*(R + 1 + 1) + ~70)
constexpr int stream2_rows_c(int nd, int nv, int na = 2) {
  return nd > 0 ? ((nd == 1 && na == 1 && nv > 8) ? 2 : (nv <= 2 ? 8 : nv <= 4 ? 4 : nv == 5 ? 2 : nv <= 6 ? 3 : nv <= 8 ? 2 : 1))
                : (nv <= 4 ? 8 : nv <= 8 ? 4 : 2);
}
template <int ND, int NA = 2>
inline int stream2_rows(const StreamPlan &p) { return stream2_rows_c(ND, p.nv, NA); }
template <int ND, int NA = 2>
inline int stream2_grid(const StreamPlan &p, int m) {
  const int R = stream2_rows<ND, NA>(p);
  const int nblk = (m + R - 1) / R;
  const int gmax = ND > 0 ? (p.tpb == 256 && p.num_cu > 0
                                 ? p.num_cu * ((p.pf && ND == 2 && NA == 2) ? 2
                                               : p.bpc_override > 0         ? p.bpc_override
                                                                            : stream2_blocks_per_cu(256, p.nv, ND, NA))
                                 : p.grid_max)
                          : p.grid_dot;   // (the column-sum-only form keeps the two-per-CU grid)
  return nblk < gmax ? (nblk > 0 ? nblk : 1) : gmax;
}

template <typename T, int ND, int NA, typename Tag = AllPlans, typename Op>
void launch_stream2(const StreamPlan &p, const StreamArgs2<T> &a, const Op &op, hipStream_t s) {
  POGS_CHECK(stream2_supported(p), "plan not supported by the two-accumulator kernel");
  const int grid = stream2_grid<ND, NA>(p, a.m);
  if constexpr (ND == 2 && NA == 2 && std::is_same<T, float>::value && Tag::has(256, 5)) {
    if (p.pf && p.tpb == 256 && p.nv == 5) {   // (the slow-functor form: next tile in flight, two workgroups per CU)
      hipLaunchKernelGGL((stream_rows2_pf_kernel<T, 256, 5, stream2_rows_c(2, 5, 2), Op>), dim3(grid), dim3(256), 0, s, a, op);
      return;
    }
  }
#define POGS_STREAM2_CASE(TPB_, NV_)                                                            \
  if constexpr (Tag::has(TPB_, NV_) && TPB_ != 1024)                                            \
  if (p.tpb == TPB_ && p.nv == NV_) {                                                           \
    constexpr int R_ = stream2_rows_c(ND, NV_, NA);                                             \
    const size_t lds = (ND > 1) ? static_cast<size_t>(a.n_pad) * sizeof(T) : 0;                 \
    static SmemGrants grants;   /* per device; concurrent solvers share it */                   \
    ensure_dynamic_smem(reinterpret_cast<const void *>(&stream_rows2_kernel<T, TPB_, NV_, R_, ND, NA, Op>), lds, grants); \
    hipLaunchKernelGGL((stream_rows2_kernel<T, TPB_, NV_, R_, ND, NA, Op>), dim3(grid),         \
                       dim3(TPB_), lds, s, a, op);                                              \
    return;                                                                                     \
  }
  POGS_STREAM_PLANS(POGS_STREAM2_CASE)
#undef POGS_STREAM2_CASE
  throw Error("no stream2 kernel instance for this plan in this translation unit");
}

// ---------------------------------------------------------------------------
// reduce_cols: out-of-kernel second stage of the column sums.
//   total[j] = sum_b partials[b][j]   (b in fixed order)
// and hands total[j] to a per-column functor.
// Workgroup = 256 threads = 32 column-vectors x 8 partial groups.
// ---------------------------------------------------------------------------
template <typename T, typename ColOp>
__global__ void __launch_bounds__(256) reduce_cols_kernel(const T *partials, int nparts, int n_pad,
                                                          ColOp op, double *scalar_partials) {
  using V = typename Vec16<T>::type;
  constexpr int VEC = Vec16<T>::N;
  constexpr int NS = ColOp::NS > 0 ? ColOp::NS : 1;
  __shared__ V s_v[8][32];
  __shared__ double s_red[NS * 4];
  const int cx = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int col = (blockIdx.x * 32 + cx) * VEC;
  V sum = dev::vzero<V>();
  if (col < n_pad) {
    // 8 independent loads in flight per thread (the partials sit in L2 / Infinity Cache)
    int b = g;
    for (; b + 56 < nparts; b += 64) {
      V v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q)
        v[q] = *reinterpret_cast<const V *>(partials + static_cast<size_t>(b + 8 * q) * n_pad + col);
#pragma unroll
      for (int q = 0; q < 8; ++q) dev::vadd(sum, v[q]);
    }
    for (; b < nparts; b += 8) {
      const V v = *reinterpret_cast<const V *>(partials + static_cast<size_t>(b) * n_pad + col);
      dev::vadd(sum, v);
    }
  }
  s_v[g][cx] = sum;
  __syncthreads();
  double sacc[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) sacc[k] = 0.0;
  if (g == 0 && col < n_pad) {
    V tot = s_v[0][cx];
#pragma unroll
    for (int q = 1; q < 8; ++q) dev::vadd(tot, s_v[q][cx]);
    const T *tp = reinterpret_cast<const T *>(&tot);
#pragma unroll
    for (int i = 0; i < VEC; ++i) op.col(col + i, tp[i], sacc);
  }
  if (ColOp::NS > 0) {
    __syncthreads();
    dev::block_sum<NS, 256>(sacc, s_red);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < NS; ++k) scalar_partials[static_cast<size_t>(blockIdx.x) * NS + k] = sacc[k];
    }
  }
}

inline int reduce_cols_grid(int n_pad, int vec) { return (n_pad / vec + 31) / 32; }

template <typename T, typename ColOp>
void launch_reduce_cols(const T *partials, int nparts, int n_pad, const ColOp &op,
                        double *scalar_partials, hipStream_t s) {
  const int grid = reduce_cols_grid(n_pad, Vec16<T>::N);
  hipLaunchKernelGGL((reduce_cols_kernel<T, ColOp>), dim3(grid), dim3(256), 0, s, partials, nparts,
                     n_pad, op, scalar_partials);
}

}  // namespace pogs_amd
