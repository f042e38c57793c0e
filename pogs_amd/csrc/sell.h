// Tiled sliced-ELL storage and SpMV for the sparse operator (gfx950).
//
// Reference operation: MatrixSparse::Mul -> spblas_gemv (src/cpu/matrix/matrix_sparse.cpp:139-155,
// src/cpu/include/gsl/gsl_spblas.h:10-40): a row-gather CSR SpMV, used for A and (on the
// transposed copy) for A^T.  On the GPU a random gather x[ind] from HBM/L2 costs one 64-byte
// request per non-zero and that request rate, not the bytes, bounds a CSR kernel (~1 TB/s at
// C4).  So the gather is served from LDS:
//
//   * the matrix is cut into TILES of RR rows x BW columns (fp32: <= 8192 x 24576).  A workgroup
//     (1024 threads, one per CU) owns one row range: it keeps the RR row sums in LDS, walks the
//     column blocks of its column group, and for each tile loads that block's slice of x into
//     LDS (96 KB) and gathers from there.  The row sums never leave LDS between column blocks:
//     no per-(block, row) partial sums travel through HBM.  (Only when a matrix has too few row
//     ranges to fill the chip are the column blocks split into a few column GROUPS, whose
//     partial row sums -- groups x rows values, not blocks x rows -- a second kernel adds in
//     group order.)
//   * inside a tile the rows are sorted by their non-zero count (stable) and packed into SLICES
//     of 64 rows of equal length L <= 32, stored column-major ([L][64]: a wavefront reads 256 B of
//     values and 128 B of uint16 local columns per step, fully coalesced).  Lane l of a slice owns
//     one row: L multiply-adds in column order, then ONE read-add-write of its row sum in LDS (a
//     row occurs once per tile, so no two lanes ever touch the same sum: no atomics, and the
//     summation order of a row -- ascending column blocks, ascending columns inside -- is fixed).
//     perm (uint16 per slot) names the row of each lane; rows longer than 32 inside one tile get
//     a slice of their own that the whole wavefront strides over.
//   * bytes per non-zero: 4 (value) + 2 (local column) + ~0.8 (perm, ~2.5 non-zeros per
//     (tile, row) at C4) -- below the 8 of plain CSR -- plus <2 % padding where a slice mixes two
//     lengths; no row offsets at all.
//
// Every slice has a 32-bit descriptor (unit offset << 6 | L; units of 64 elements; L = 0: long
// row) read through the scalar cache two slices ahead; values / columns of slice s + 32 are
// requested while slice s is multiplied (two register stages per wavefront, 16 independent
// wavefronts per CU, no workgroup barrier inside a tile).
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"
#include "reduce.h"

namespace pogs_amd {

constexpr int kSellTpb = 512;           // one workgroup per CU (LDS), 8 wavefronts with 256 VGPRs each
constexpr int kSellWaves = kSellTpb / 64;
constexpr int kSellLmax = 32;           // longest row kept as one lane's work
constexpr int kSellStage = 6;           // wave-rows of a slice held in registers
template <typename T> struct SellCfg;
template <> struct SellCfg<float> { static constexpr int BW = 24576, RR = 8192; };    // 96 KB + 32 KB of LDS
template <> struct SellCfg<double> { static constexpr int BW = 12288, RR = 4096; };   // 96 KB + 32 KB
constexpr unsigned short kSellNoRow = 0xFFFF;

template <typename T>
struct SellView {
  const T *val;
  const unsigned short *loc;    // column - first column of the tile's block
  const unsigned short *perm;   // [slice][64]: row - first row of the tile's range (kSellNoRow: padding lane)
  const unsigned *desc;         // [slice]
  const int *tile_ptr;          // [nrr * ncb + 1]: first slice of each tile
  int nrows, ncols;
  int rr_rows;                  // rows per row range (multiple of 64, <= SellCfg::RR)
  int nrr, ncb;                 // row ranges, column blocks
  int ncg, cb_per_group;        // column groups and their width in blocks
};

template <typename T>
struct SellRegs {
  T v[kSellStage];
  unsigned short l[kSellStage];
  unsigned short row;
};
constexpr int kSellRing = 8;            // slices in flight per wavefront
constexpr unsigned kSellDummy = 63u;    // L field of a padding slice (a tile's slice count is a multiple of 16)

// DIRECT (ncg == 1): the row functor runs here; otherwise part[cg * nrows + row] receives the
// column group's partial sums.
//
// Every wavefront walks ITS slices of the workgroup's tiles as one stream (slice w, w + 16, ... of
// each tile; the build pads every tile to a multiple of 16 slices, so all wavefronts cross a tile
// boundary at the same step): kSellRing slices are in flight per wavefront, also across tile
// boundaries -- a fetch needs no LDS -- and the x slice of the next tile waits in registers,
// requested one tile ahead, so a boundary costs two barriers and the LDS stores, no memory latency.
template <typename T, bool SQ, bool DIRECT, typename Op>
__global__ void __launch_bounds__(kSellTpb) spmv_sell_kernel(SellView<T> A, const T *__restrict__ x,
                                                             const double *x_nrm2, Op op, T *__restrict__ part,
                                                             double *scalar_partials) {
  constexpr int NS = Op::NS > 0 ? Op::NS : 1;
  constexpr int BW = SellCfg<T>::BW;
  constexpr int XR = BW / kSellTpb;            // x values per thread and tile
  extern __shared__ __attribute__((aligned(16))) unsigned char sell_smem[];
  T *s_x = reinterpret_cast<T *>(sell_smem);   // [BW]
  T *s_y = s_x + BW;                           // [RR]
  __shared__ double s_red[NS * kSellWaves];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  // consecutive workgroups (round-robin over the XCDs) take the column groups of one row range:
  // with 8 groups every XCD keeps re-reading the same eighth of x from its own L2
  const int cg = static_cast<int>(blockIdx.x) % A.ncg;
  const int rr = static_cast<int>(blockIdx.x) / A.ncg;
  const int row0 = rr * A.rr_rows;
  const int nr = min(A.rr_rows, A.nrows - row0);
  T xs = 1;
  if (x_nrm2) xs = static_cast<T>(1.0 / sqrt(*x_nrm2));
  for (int i = t; i < nr; i += kSellTpb) s_y[i] = 0;

  const T *__restrict__ a_val = A.val;
  const unsigned short *__restrict__ a_loc = A.loc;
  const unsigned short *__restrict__ a_perm = A.perm;
  const unsigned *__restrict__ a_desc = A.desc;
  const int *__restrict__ a_tptr = A.tile_ptr + static_cast<size_t>(rr) * A.ncb;   // this row range's tiles

  auto fetch = [&](unsigned d, int s, SellRegs<T> &R) {
    const int L = static_cast<int>(d & 63u);
    if (L == 0 || L == static_cast<int>(kSellDummy)) return;   // long row: streamed in consume(); padding: nothing
    const size_t off = static_cast<size_t>(d >> 6) * 64 + lane;
    R.row = __builtin_nontemporal_load(a_perm + static_cast<size_t>(s) * 64 + lane);
#pragma unroll
    for (int j = 0; j < kSellStage; ++j) {
      if (j < L) {   // L is wave-uniform: a scalar branch
        R.v[j] = __builtin_nontemporal_load(a_val + off + j * 64);
        R.l[j] = __builtin_nontemporal_load(a_loc + off + j * 64);
      }
    }
  };
  auto consume = [&](unsigned d, int s, const SellRegs<T> &R) {
    const int L = static_cast<int>(d & 63u);
    if (L == static_cast<int>(kSellDummy)) return;
    const size_t base = static_cast<size_t>(d >> 6) * 64;
    if (L > 0) {
      T acc = 0;
#pragma unroll
      for (int j = 0; j < kSellStage; ++j) {
        if (j < L) {
          const T v = R.v[j];
          acc += (SQ ? v * v : v) * s_x[R.l[j]];
        }
      }
      for (int j = kSellStage; j < L; ++j) {   // longer rows: the tail straight from memory
        const T v = __builtin_nontemporal_load(a_val + base + j * 64 + lane);
        const unsigned short c = __builtin_nontemporal_load(a_loc + base + j * 64 + lane);
        acc += (SQ ? v * v : v) * s_x[c];
      }
      if (R.row != kSellNoRow) s_y[R.row] += acc;
    } else {
      // one long row: perm holds {row, len low, len high}; the wavefront strides over it
      const unsigned short *pp = a_perm + static_cast<size_t>(s) * 64;
      const int row = pp[0];
      const int len = static_cast<int>(pp[1]) | (static_cast<int>(pp[2]) << 16);
      T acc = 0;
      for (int k = lane; k < len; k += 64) {
        const T v = a_val[base + k];
        acc += (SQ ? v * v : v) * s_x[a_loc[base + k]];
      }
      acc = dev::wave_sum(acc);
      if (lane == 0) s_y[row] += acc;
    }
  };

  const int cb0 = cg * A.cb_per_group, cb1 = min(A.ncb, cb0 + A.cb_per_group);
  // x slice of the next non-empty tile, in registers
  T xreg[XR];
  int x_cb = cb0 - 1;
  auto x_prefetch = [&]() {   // advance x_cb to the next non-empty tile and request its slice
    do { ++x_cb; } while (x_cb < cb1 && a_tptr[x_cb] == a_tptr[x_cb + 1]);
    if (x_cb >= cb1) return;
    const int c0 = x_cb * BW, w = min(BW, A.ncols - c0);
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const int c = i * kSellTpb + t;
      xreg[i] = (c < w) ? x[c0 + c] : static_cast<T>(0);
    }
  };
  // fetch cursor: this wavefront's next slice
  int f_cb = cb0 - 1, f_s = 0, f_end = 0;
  auto f_next = [&]() -> bool {
    while (f_s >= f_end) {
      if (++f_cb >= cb1) return false;
      f_s = a_tptr[f_cb] + wave;
      f_end = a_tptr[f_cb + 1];
    }
    return true;
  };
  int c_cb = cb0 - 1, c_end = 0;   // consume cursor: the tile whose x slice is in LDS ends at c_end

  SellRegs<T> R[kSellRing];
  unsigned d[kSellRing];
  int sidx[kSellRing];
  x_prefetch();
#pragma unroll
  for (int q = 0; q < kSellRing; ++q) {
    if (f_next()) {
      sidx[q] = f_s;
      d[q] = a_desc[f_s];
      fetch(d[q], f_s, R[q]);
      f_s += kSellWaves;
    } else {
      sidx[q] = -1;
      d[q] = kSellDummy;
    }
  }
  bool live = sidx[0] >= 0;
  while (live) {
#pragma unroll
    for (int q = 0; q < kSellRing; ++q) {
      if (sidx[q] < 0) { live = false; break; }   // uniform; the ring drains in order
      if (sidx[q] >= c_end) {
        // first slice of a later tile (the same step for every wavefront): swap the x slice
        do { ++c_cb; c_end = a_tptr[c_cb + 1]; } while (c_end <= sidx[q]);
        __syncthreads();   // the previous tile's gathers and row-sum updates are done
#pragma unroll
        for (int i = 0; i < XR; ++i) s_x[i * kSellTpb + t] = xreg[i] * xs;
        x_prefetch();
        __syncthreads();
      }
      consume(d[q], sidx[q], R[q]);
      if (f_next()) {
        sidx[q] = f_s;
        d[q] = a_desc[f_s];
        fetch(d[q], f_s, R[q]);
        f_s += kSellWaves;
      } else {
        sidx[q] = -1;
        d[q] = kSellDummy;
      }
    }
  }
  __syncthreads();
  double sacc[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) sacc[k] = 0.0;
  if (DIRECT) {
    for (int i = t; i < nr; i += kSellTpb) op.row(row0 + i, s_y[i], sacc);
    if (Op::NS > 0) {
      dev::block_sum<NS, kSellTpb>(sacc, s_red);
      if (t == 0) {
#pragma unroll
        for (int k = 0; k < NS; ++k) scalar_partials[static_cast<size_t>(blockIdx.x) * NS + k] = sacc[k];
      }
    }
  } else {
    T *out = part + static_cast<size_t>(cg) * A.nrows + row0;
    for (int i = t; i < nr; i += kSellTpb) out[i] = s_y[i];
  }
}

// ---------------------------------------------------------------------------------------------
// Build (device).  Temporaries per (tile, local row): cnt (non-zeros), slot (slice * 64 + lane of
// the row inside its tile), cursor (fill position) -- uint16 each: a tile is at most 24576 wide
// and holds at most RR + 32 slices of 64 slots.
// ---------------------------------------------------------------------------------------------
struct SellDims {
  int nrows, ncols, rr_rows, nrr, ncb, bw;
};

// cnt[tile * rr_rows + local row] = non-zeros of that row inside the tile (one thread per row)
__global__ void sell_count_kernel(const int *ind, const int *ptr, SellDims D, unsigned short *cnt) {
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < D.nrows; r += gridDim.x * blockDim.x) {
    const int rr = r / D.rr_rows, lr = r - rr * D.rr_rows;
    for (int k = ptr[r]; k < ptr[r + 1]; ++k) {
      const int cb = ind[k] / D.bw;
      cnt[(static_cast<size_t>(rr) * D.ncb + cb) * D.rr_rows + lr] += 1;
    }
  }
}

// One workgroup (256 threads) per tile.  Classes: 1..32 = rows of that length, 33 = longer.
// Slice order inside a tile: the long rows (one slice each, in row order), then the classes from
// 32 down to 1, rows of a class in row order (stable), 64 to a slice.
// FINAL = false: tile_ns[tile] = slices, tile_nu[tile] = 64-element units of the tile.
// FINAL = true : with tile_ptr / tile_uptr (exclusive scans of those) writes desc, perm, slot.
constexpr int kSellClasses = kSellLmax + 2;   // 0 (empty, unused), 1..32, 33
template <bool FINAL>
__global__ void __launch_bounds__(256) sell_plan_kernel(const unsigned short *cnt, SellDims D, int *tile_ns,
                                                        int *tile_nu, const int *tile_ptr, const int *tile_uptr,
                                                        unsigned *desc, unsigned short *perm, unsigned short *slot) {
  __shared__ unsigned short s_hist[kSellClasses][256];   // per thread and class: rows, then their exclusive prefix
  __shared__ int s_tot[kSellClasses];                    // rows per class
  __shared__ int s_sbase[kSellClasses], s_ubase[kSellClasses];   // first slice / first unit of a class in the tile
  __shared__ int s_lu[256];                              // per thread: units of its long rows, then exclusive prefix
  __shared__ int s_lutot;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int rpt = (D.rr_rows + 255) / 256;               // consecutive rows per thread (<= 32)
  for (int tile = blockIdx.x; tile < D.nrr * D.ncb; tile += gridDim.x) {
    const int rr = tile / D.ncb;
    const int nr = min(D.rr_rows, D.nrows - rr * D.rr_rows);
    const unsigned short *tc = cnt + static_cast<size_t>(tile) * D.rr_rows;
    for (int c = 0; c < kSellClasses; ++c) s_hist[c][t] = 0;
    int lu = 0;
    const int r_lo = t * rpt, r_hi = min(nr, r_lo + rpt);
    for (int r = r_lo; r < r_hi; ++r) {
      const int len = tc[r];
      if (len == 0) continue;
      const int c = len <= kSellLmax ? len : kSellLmax + 1;
      s_hist[c][t] += 1;
      if (c == kSellLmax + 1) lu += (len + 63) / 64;
    }
    s_lu[t] = lu;
    __syncthreads();
    // exclusive scans over the 256 threads: one wavefront per class (4 entries per lane)
    for (int c = 1 + wave; c <= kSellClasses; c += 4) {   // c == kSellClasses: the long-row units
      int v[4], sum = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q] = (c < kSellClasses) ? s_hist[c][lane * 4 + q] : s_lu[lane * 4 + q];
        sum += v[q];
      }
      int inc = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
      }
      int run = inc - sum;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (c < kSellClasses) s_hist[c][lane * 4 + q] = static_cast<unsigned short>(run);
        else s_lu[lane * 4 + q] = run;
        run += v[q];
      }
      if (lane == 63) {
        if (c < kSellClasses) s_tot[c] = inc;
        else s_lutot = inc;
      }
    }
    __syncthreads();
    if (t == 0) {
      int sb = s_tot[kSellLmax + 1], ub = s_lutot;   // the long rows come first
      s_sbase[kSellLmax + 1] = 0;
      s_ubase[kSellLmax + 1] = 0;
      for (int L = kSellLmax; L >= 1; --L) {
        const int ns = (s_tot[L] + 63) / 64;
        s_sbase[L] = sb;
        s_ubase[L] = ub;
        sb += ns;
        ub += ns * L;
      }
      s_sbase[0] = sb;   // real slices of the tile; padded to a multiple of the wavefronts per workgroup
      if (!FINAL) {
        tile_ns[tile] = (sb + kSellWaves - 1) / kSellWaves * kSellWaves;
        tile_nu[tile] = ub;
      }
    }
    __syncthreads();
    if (FINAL) {
      const int sp = tile_ptr[tile], up = tile_uptr[tile];
      for (int q = sp + s_sbase[0] + t; q < tile_ptr[tile + 1]; q += 256) desc[q] = 63u;   // padding slices (kSellDummy)
      // descriptors of the short classes
      for (int L = 1; L <= kSellLmax; ++L) {
        const int ns = (s_tot[L] + 63) / 64;
        for (int q = t; q < ns; q += 256)
          desc[sp + s_sbase[L] + q] = (static_cast<unsigned>(up + s_ubase[L] + q * L) << 6) | static_cast<unsigned>(L);
      }
      unsigned short *ts = slot + static_cast<size_t>(tile) * D.rr_rows;
      int rank[kSellClasses];   // running position of this thread inside each class (prefix + own rows so far)
      int lrun = s_lu[t];
#pragma unroll
      for (int c = 0; c < kSellClasses; ++c) rank[c] = 0;
      for (int r = r_lo; r < r_hi; ++r) {
        const int len = tc[r];
        if (len == 0) continue;
        if (len <= kSellLmax) {
          // (dynamic index into a small local array: the compiler keeps it in scratch; one-time code)
          const int k = s_hist[len][t] + rank[len];
          rank[len] += 1;
          const int sl = s_sbase[len] + (k >> 6), ln = k & 63;
          perm[(static_cast<size_t>(sp) + sl) * 64 + ln] = static_cast<unsigned short>(r);
          ts[r] = static_cast<unsigned short>(sl * 64 + ln);
        } else {
          const int sl = s_hist[kSellLmax + 1][t] + rank[kSellLmax + 1];
          rank[kSellLmax + 1] += 1;
          desc[sp + sl] = static_cast<unsigned>(up + lrun) << 6;   // L = 0
          unsigned short *pp = perm + (static_cast<size_t>(sp) + sl) * 64;
          pp[0] = static_cast<unsigned short>(r);
          pp[1] = static_cast<unsigned short>(len & 0xFFFF);
          pp[2] = static_cast<unsigned short>(len >> 16);
          ts[r] = static_cast<unsigned short>(sl);   // long rows: the slice itself (<= RR slices, fits)
          lrun += (len + 63) / 64;
        }
      }
    }
    __syncthreads();
  }
}

// copies every non-zero to its place (one thread per row, non-zeros in their CSR order); cursor
// must be zero on entry.  loc == nullptr: values only (after a rescaling of the CSR copy).
template <typename T>
__global__ void sell_fill_kernel(const T *val, const int *ind, const int *ptr, SellDims D, const unsigned short *cnt,
                                 const unsigned short *slot, unsigned short *cursor, const int *tile_ptr,
                                 const unsigned *desc, T *sval, unsigned short *sloc) {
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < D.nrows; r += gridDim.x * blockDim.x) {
    const int rr = r / D.rr_rows, lr = r - rr * D.rr_rows;
    for (int k = ptr[r]; k < ptr[r + 1]; ++k) {
      const int c = ind[k], cb = c / D.bw;
      const int tile = rr * D.ncb + cb;
      const size_t idx = static_cast<size_t>(tile) * D.rr_rows + lr;
      const int j = cursor[idx];
      cursor[idx] = static_cast<unsigned short>(j + 1);
      const int sl = slot[idx];
      const bool is_long = cnt[idx] > kSellLmax;
      const unsigned d = desc[tile_ptr[tile] + (is_long ? sl : (sl >> 6))];
      const size_t base = static_cast<size_t>(d >> 6) * 64;
      const size_t dst = is_long ? base + j : base + static_cast<size_t>(j) * 64 + (sl & 63);
      sval[dst] = val[k];
      if (sloc) sloc[dst] = static_cast<unsigned short>(c - cb * D.bw);
    }
  }
}

}  // namespace pogs_amd
