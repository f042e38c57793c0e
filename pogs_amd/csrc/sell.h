// Tiled lane-stream storage and SpMV for the sparse operator (gfx950).
//
// Reference operation: MatrixSparse::Mul -> spblas_gemv (src/cpu/matrix/matrix_sparse.cpp:139-155,
// src/cpu/include/gsl/gsl_spblas.h:10-40): a row-gather CSR SpMV, used for A and (on the
// transposed copy) for A^T.  On the GPU a random gather x[ind] from HBM/L2 costs one 64-byte
// request per non-zero and that request rate, not the bytes, bounds a CSR kernel (~1 TB/s at
// C4).  So the gather is served from LDS:
//
//   * the matrix is cut into TILES of RR rows x BW columns (fp32: <= 16384 x 18432).  A workgroup
//     (512 threads, one per CU) owns one row range: it keeps the RR row sums in LDS, walks the
//     column blocks of its column group, and for each tile puts that block's slice of x into LDS
//     (72 KB) and gathers from there.  The row sums never leave LDS between column blocks: no
//     per-(block, row) partial sums travel through HBM.  (Only when a matrix has too few row
//     ranges to fill the chip are the column blocks split into a few column GROUPS, whose partial
//     row sums -- groups x rows values, not blocks x rows -- a second kernel adds in group order.)
//   * inside a tile every one of the 512 lanes owns a STREAM of whole rows, one after the other:
//     element k of the streams of a wavefront is one 64-wide "wave-row" in memory (256 B of values,
//     128 B of uint16 local columns, 128 B of uint16 row tags -- three fully coalesced,
//     UNCONDITIONAL loads per wave-row, so the compiler can keep a fixed number of them in flight).
//     The tag is 0xFFFF except on the last element of a row, where it names the row: the lane
//     adds its running sum to that row's sum in LDS and starts over.  A row lies in exactly one
//     stream of a tile, so no two lanes ever touch the same sum (no atomics), and a row is summed
//     in ascending column blocks and CSR order inside a block -- a fixed order.
//     Rows are dealt to the streams longest first, a row only to streams whose lane has the row's residue mod 32
//     (round 6, sell_plan_kernel: the row-end flushes of a wavefront then never meet on an LDS bank), greedily onto the
//     least loaded one: the 512 stream lengths of a tile stay within a batch of each other (padding ~6 % at C4).
//   * bytes per non-zero: 4 (value) + 2 (local column) + 2 (tag) = the 8 of CSR (s + 4), no row
//     offsets and no x traffic from HBM.
//
//   * in memory a tile is a sequence of STEPS: step b holds batch b (4 elements per lane) of each of
//     the workgroup's 8 wavefronts side by side, so the workgroup reads 8 KB of values and 4 KB of
//     each index array contiguously per step and walks forward through its tiles -- one
//     sequential stream per array and workgroup instead of one per wavefront (round 4: 145 -> 142 us
//     at C4; putting the three arrays into one stream gave nothing more).
//
// Pipeline: a wavefront walks its batches of all tiles of the workgroup as one sequence with 8
// batches (24 loads) in flight, also across tile boundaries -- a fetch needs no LDS -- and the x
// slice of the next tile waits in registers, requested one tile ahead, so a tile boundary costs two
// barriers and the LDS stores, no memory latency.  What bounds the kernel is this load stream alone
// (round 4, profiles/NOTES_r04.md: with the gathers and row-sum updates compiled out it takes 142.8
// instead of 145.2 us; ring depths 4 / 6 / 8 are equal, 12 slower; the LDS reads of the next batch
// requested ahead of this batch's stores: no change).
//
// Round 5 (profiles/NOTES_r05.md) looked at everything around this stream once more, with per-workgroup time stamps
// (POGS_AMD_SELL_STAMPS) and a micro-benchmark of the bare load stream (scripts/micro/stream_pattern.hip):
//   * the bare stream of this kernel's bytes -- 248 workgroups, three arrays, 8 batches in flight -- takes 124-129 us
//     (6.6-6.8 TB/s), 132 us with the x slices and their LDS swaps; the MEAN workgroup of the real kernel takes
//     127-131 us: a workgroup streams as fast as the bare pattern does.  The kernel time (136-146 us) is its slowest one;
//   * the XCDs stream at the same rate (work / time within +-4 %): what differs between workgroups is the number of
//     column blocks (13 or 14).  Equal blocks (narrower, 112 instead of 109) made every workgroup as slow as the
//     slow ones: 1372 against 1377 it/s;
//   * where the streams lie in memory does not matter (all workgroups marching through one window: 124.2 against
//     123.6 us);
//   * the wait for the x slice in registers drains the 8-deep load ring at every tile boundary (vmcnt(11..3) in the
//     ISA).  Four extra producer wavefronts that own the slices (own vmcnt; the streaming wavefronts never wait for
//     more than their oldest batch, checked in the ISA) changed nothing: 1346 against 1340-1351 it/s -- the ring
//     refills faster than a tile takes;
// so what is left is bytes (8.4 per non-zero here).
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"
#include "reduce.h"

namespace pogs_amd {

constexpr int kSellTpb = 512;           // one workgroup per CU (LDS), 8 wavefronts with up to 256 VGPRs each
constexpr int kSellWaves = kSellTpb / 64;
constexpr int kSellStreams = kSellTpb;  // lane streams per tile
constexpr int kSellUB = 4;              // elements per lane and batch (a tile's stream length is a multiple of it)
#ifndef POGS_SELL_NB   // (compile-time override: ring-depth A / B runs through scripts/build_variant.py; two-slot format at C4: 6 = 8, 10 and 12 are 1.5-2 % slower)
#define POGS_SELL_NB 8
#endif
constexpr int kSellNB = POGS_SELL_NB;   // batches in flight per wavefront (3 x 16- / 8- / 4-byte loads each)
constexpr int kSellLmax = 32;           // sort classes: row lengths 1..32 each, longer rows together
__device__ __forceinline__ float sell_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double sell_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <typename T> struct SellCfg;
#ifndef POGS_SELL_BW_F32   // (compile-time overrides: tile-shape experiments through scripts/build_variant.py, scripts/spmv_probe.sh)
#define POGS_SELL_BW_F32 18432
#define POGS_SELL_RR_F32 16384
#endif
template <> struct SellCfg<float> { static constexpr int BW = POGS_SELL_BW_F32, RR = POGS_SELL_RR_F32; };   // 72 KB + 64 KB of LDS (measured at C4 against 24576 x 12288, 20480 x 16384, 22528 x 16384, 14336 x 16384: 165 / 155 / 163 / 155 us, this one 152; 16384 x 16384 -- every x slice on a 64 KB boundary -- 237)
template <> struct SellCfg<double> { static constexpr int BW = 12288, RR = 6144; };   // 96 KB + 48 KB
constexpr unsigned short kSellNoRow = 0xFFFF;
constexpr int kSellOffBits = 23;        // plan: stream offset of a row inside its tile (9 bits of stream above it)
constexpr int kSellOff2Bits = 22;       // (two id slots) the offset; bit 22: the row's end is the SECOND one of its batch

template <typename T>
struct SellView {
  const T *val;
  const unsigned short *loc;    // column - first column of the tile's block
  const unsigned short *rid;    // tags: kSellNoRow, or (row - first row of the range) on the last element of a row;
                                // two (below): per batch of a lane the rows of its first and second row end
  const int *tile_unit;         // [nrr * ncb + 1]: first 64-element unit of each tile; a tile holds 8 K units
  int nrows, ncols;
  int rr_rows;                  // rows per row range (<= SellCfg::RR)
  int nrr, ncb;                 // row ranges, column blocks
  int ncg;                      // column groups (an even split of the column blocks)
  int two;                      // 1: "two id slots" format -- bit 15 of loc marks the last element of a row and rid holds
                                // 2 ids per 4-element batch (no batch of a stream has more than 2 row ends: the planner
                                // orders a stream's rows that way); 7 bytes per stored fp32 element instead of 8
  // debug (POGS_AMD_SELL_STAMPS): per workgroup {start, end (100 MHz wall clock), XCC id, 64-element units walked}
  unsigned long long *stamps;
};

// One batch = kSellUB consecutive elements of every lane's stream, stored lane-major: the values of
// a lane are one 16-byte (fp32) vector, its local columns and row tags one 8-byte vector each, so a
// wavefront fetches a batch with three fully coalesced wide loads; batch b of wavefront w of a tile
// sits at element (first unit of the tile) * 64 + (b * 8 + w) * 256.
template <typename T, bool TWO>
struct SellBatch {
  T v[kSellUB];
  unsigned short c[kSellUB], r[TWO ? 1 : kSellUB];
  unsigned rr;   // (two id slots) both ids as loaded: first end's row in the low half, second's in the high half
};
#ifndef POGS_SELL_FLAT_FLUSH   // (0: the two-slot format's row ends flushed under exec masks, as the tag format's: round 5's A / B)
#define POGS_SELL_FLAT_FLUSH 1
#endif
constexpr bool kSellFlatFlush = POGS_SELL_FLAT_FLUSH != 0;
// LDS words of the streaming body: x slice, row sums, and (two id slots) one scratch word per lane for the flat flush
template <typename T> constexpr size_t sell_lds_bytes() {
  return (static_cast<size_t>(SellCfg<T>::BW) + SellCfg<T>::RR + kSellTpb) * sizeof(T);
}
constexpr unsigned short kSellEndBit = 0x8000;   // (two id slots) set in loc on the last element of a row
template <typename V> __device__ __forceinline__ V dev_vzero() {
  V v;
  __builtin_memset(&v, 0, sizeof(V));
  return v;
}
template <int BYTES> struct SellRaw;
template <> struct SellRaw<4> { typedef unsigned int type; };
template <> struct SellRaw<8> { typedef unsigned int type __attribute__((ext_vector_type(2))); };
template <> struct SellRaw<16> { typedef unsigned int type __attribute__((ext_vector_type(4))); };
template <typename E, int N, bool NT = true>
__device__ __forceinline__ void sell_load(const E *p, E (&out)[N]) {   // N * sizeof(E) bytes, non-temporal, in <= 16-byte pieces
  constexpr int BYTES = N * static_cast<int>(sizeof(E));
  constexpr int PIECE = BYTES >= 16 ? 16 : (BYTES >= 8 ? 8 : 4);
  static_assert(BYTES % PIECE == 0, "batch vectors are 4, 8 or 16 byte multiples");
  typedef typename SellRaw<PIECE>::type R;
  R raw[BYTES / PIECE];
#pragma unroll
  for (int i = 0; i < BYTES / PIECE; ++i)
    raw[i] = NT ? __builtin_nontemporal_load(reinterpret_cast<const R *>(p) + i) : *(reinterpret_cast<const R *>(p) + i);
  __builtin_memcpy(out, raw, BYTES);
}

// Read of the tile table at a wavefront-uniform index as a SCALAR load (s_load_dword through the
// constant address space; the table is written by the build kernels of an earlier launch only).
// As an ordinary global load it is a vector-memory instruction whose `s_waitcnt vmcnt(0)` also
// waits for every batch load in flight -- it drained the ring at each tile boundary.
__device__ __forceinline__ int sell_uniform_load(const int *p) {
  typedef const int __attribute__((address_space(4))) *cptr;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  return *(cptr)(p);
#pragma clang diagnostic pop
}

// The row sums of one (row range, column group) into s_y[0 .. nr): the streaming body shared by the
// SpMV kernels below.  s_x: BW elements of LDS, s_y: RR elements; ends with a barrier (s_y complete).
template <typename T, bool SQ, bool TWO>
__device__ __forceinline__ void sell_row_sums(const SellView<T> &A, const T *__restrict__ x, T xs, int rr, int cg, int nr,
                                              T *s_x, T *s_y) {
  constexpr int BW = SellCfg<T>::BW;
  constexpr int UB = kSellUB, NB = kSellNB;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  for (int i = t; i < nr; i += kSellTpb) s_y[i] = 0;

  const T *__restrict__ a_val = A.val;
  const unsigned short *__restrict__ a_loc = A.loc;
  const unsigned short *__restrict__ a_rid = A.rid;
  const int *__restrict__ a_tu = A.tile_unit + static_cast<size_t>(rr) * A.ncb;   // this row range's tiles
  auto tu = [&](int cb) { return sell_uniform_load(a_tu + cb); };

  auto fetch = [&](size_t e0, SellBatch<T, TWO> &B) {   // e0: first element of this lane's batch
#ifndef POGS_SELL_VAL_NT
#define POGS_SELL_VAL_NT 1
#endif
#ifndef POGS_SELL_LOC_NT
#define POGS_SELL_LOC_NT 1
#endif
    sell_load<T, UB, POGS_SELL_VAL_NT != 0>(a_val + e0, B.v);
    sell_load<unsigned short, UB, POGS_SELL_LOC_NT != 0>(a_loc + e0, B.c);
    // (measured at C4, round 5, alternating runs on one box: the values and the columns plain instead of non-temporal
    // cost 12 % / 7 % of the it/s; the id pair -- 1 byte per element -- is the other way round, plain 1283 against 1276 it/s.
    // As a 4-byte load into two uint16 it had lost its non-temporal mark in the optimiser anyway: it is loaded as one word now)
#ifndef POGS_SELL_RID_NT
#define POGS_SELL_RID_NT 0
#endif
    if constexpr (TWO) B.rr = POGS_SELL_RID_NT ? __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(a_rid + e0 / 2))
                                               : *reinterpret_cast<const unsigned *>(a_rid + e0 / 2);
    else sell_load<unsigned short, UB>(a_rid + e0, B.r);
  };
  T acc = 0;   // running sum of the lane's current row (rows never straddle tiles)
  // POGS_SELL_ABLATE (timing only, WRONG results; scripts/build_variant.py + scripts/spmv_probe.sh, round 6's table in
  // profiles/NOTES_r06.md): bit 0 -- every LDS access of the batch compiled out (the gathers read a register instead, the
  // row sums are not flushed); bit 1 -- the row-end / id-slot logic compiled out (no element ends a row).
#ifndef POGS_SELL_ABLATE
#define POGS_SELL_ABLATE 0
#endif
  constexpr bool kNoLds = (POGS_SELL_ABLATE & 1) != 0, kNoIds = (POGS_SELL_ABLATE & 2) != 0;
  auto consume = [&](const SellBatch<T, TWO> &B) {
    // (an ablation must not shrink the load stream: the id pair / tags stay loaded even where nothing reads them -- the
    // first form of these builds let the compiler drop that load, 1 of 7.35 bytes per element, and what it measured
    // was the bytes: profiles/NOTES_r06.md section 5)
    if constexpr (POGS_SELL_ABLATE != 0) {
      if constexpr (TWO) asm volatile("" ::"v"(B.rr));
      else asm volatile("" ::"v"(B.r[0]), "v"(B.r[UB - 1]));
    }
    T xg[UB];
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      if constexpr (kNoLds) xg[j] = static_cast<T>(B.c[j]);
      else xg[j] = s_x[TWO ? (B.c[j] & (kSellEndBit - 1)) : B.c[j]];
    }
    // running sums in registers first; the row ends of the batch are then flushed with INDEPENDENT
    // LDS accesses (all reads, then all writes).  A row belongs to one lane and ends once, so the
    // (up to UB) sums a lane flushes here are distinct and nobody else touches them -- written as
    // `s_y[r] += acc` inside the element loop the compiler must assume aliasing and the batch
    // becomes UB dependent LDS round trips.
    T fl[UB], old[UB];
    bool en[UB];
    unsigned short row[UB];
    bool seen = false;   // (two id slots) a row has ended earlier in this batch: the next end is the second slot's
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      const T v = B.v[j];
      acc = sell_fma(SQ ? v * v : v, xg[j], acc);   // fused, written out: this unit is compiled with -ffp-contract=off
      if constexpr (kNoIds) {
        en[j] = false;
        row[j] = 0;
      } else if constexpr (TWO) {
        en[j] = (B.c[j] & kSellEndBit) != 0;
        row[j] = static_cast<unsigned short>(seen ? B.rr >> 16 : B.rr & 0xFFFFu);
        seen = seen || en[j];
      } else {
        en[j] = B.r[j] != kSellNoRow;
        row[j] = B.r[j];
      }
      fl[j] = acc;
      acc = en[j] ? static_cast<T>(0) : acc;
    }
    if constexpr (kNoLds) {
      // (the sums must stay live: folded into the running sum, which the kernel's tail keeps)
#pragma unroll
      for (int j = 0; j < UB; ++j) acc += en[j] ? fl[j] : static_cast<T>(0);
    } else if constexpr (TWO && kSellFlatFlush) {
      // Flat flush: EVERY element reads and writes one LDS word -- a row end its row sum, any other element the lane's own
      // scratch word behind the row sums (conflict-free) -- so the batch is straight-line code: no exec mask is saved and
      // restored around each of the eight accesses, no branch skips a write that some lane of the wavefront needs anyway
      // (with ~0.45 row ends per element every one of the four positions has ends in every wavefront).  Unused id slots
      // hold 0, so a look-up of a non-end is in range whatever it names.
      // (one read / write pair per id SLOT instead of per element -- 4 + 2 + 2 LDS instructions per batch instead of 4 + 4 + 4 --
      // measured the same: 1272 against 1276 it/s at C4; not kept)
      int at[UB];
#pragma unroll
      for (int j = 0; j < UB; ++j) at[j] = en[j] ? static_cast<int>(row[j]) : SellCfg<T>::RR + t;
#pragma unroll
      for (int j = 0; j < UB; ++j) old[j] = s_y[at[j]];
#pragma unroll
      for (int j = 0; j < UB; ++j) s_y[at[j]] = old[j] + fl[j];
    } else {
#pragma unroll
      for (int j = 0; j < UB; ++j)
        if (en[j]) old[j] = s_y[row[j]];
#pragma unroll
      for (int j = 0; j < UB; ++j)
        if (en[j]) s_y[row[j]] = old[j] + fl[j];
    }
  };

  // column blocks of this group: an even split of the ncb blocks
  const int cb0 = static_cast<int>(static_cast<long long>(cg) * A.ncb / A.ncg);
  const int cb1 = static_cast<int>(static_cast<long long>(cg + 1) * A.ncb / A.ncg);
  // x slice of the next non-empty tile, in registers (16-byte pieces)
  using V = typename Vec16<T>::type;
  constexpr int VEC = Vec16<T>::N;
  constexpr int XV = BW / (kSellTpb * VEC);
  static_assert(BW % (kSellTpb * VEC) == 0, "the x slice is loaded as whole 16-byte vectors per thread (an override of the tile width must keep that)");
  V xreg[XV];
  int x_cb = cb0 - 1, x_w = 0;   // block whose slice is in xreg, its width
  auto x_prefetch = [&]() {   // advance x_cb to the next non-empty tile and request its slice
    do { ++x_cb; } while (x_cb < cb1 && tu(x_cb) == tu(x_cb + 1));
    if (x_cb >= cb1) return;
    const int c0 = x_cb * BW, w = min(BW, A.ncols - c0);
    if (w >= VEC) {
      // unconditional loads: a vector that straddles or lies past the end re-reads the last whole
      // vector of the slice (no branch per vector, a compile-time number of loads in flight); what
      // it really holds is sorted out in x_store, one tile later, when the data has long arrived
      // (touching the loaded value here would make the compiler wait for it -- and with it for the
      // whole batch ring -- at every tile boundary)
      const int c_last = w - VEC;
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const int c = (i * kSellTpb + t) * VEC;
        xreg[i] = *reinterpret_cast<const V *>(x + c0 + min(c, c_last));
      }
    } else {
      // a last column block narrower than one vector: thread 0 owns it
#pragma unroll
      for (int i = 0; i < XV; ++i) xreg[i] = dev_vzero<V>();
      if (t == 0) {
        T out[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) out[q] = q < w ? x[c0 + q] : static_cast<T>(0);
        __builtin_memcpy(&xreg[0], out, sizeof(V));
      }
    }
    x_w = w;
  };
  auto x_store = [&]() {
    // Every word of s_x has ONE writer: the vector that straddles the end of the slice takes its
    // leading elements from the tail of the last whole vector (which is what it loaded: elements
    // c .. w-1 sit at offset c - c_last in it) and zeros behind them; vectors past the end are zero.
    // (The ragged tail used to be stored by threads 0 .. VEC-2 on top of the owner's zero vector, a
    // write-write race between wavefronts.)
    if (x_w == BW) {   // (uniform) a full-width slice: every vector is what it loaded
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        T tmp[VEC];
        __builtin_memcpy(tmp, &xreg[i], sizeof(V));
#pragma unroll
        for (int q = 0; q < VEC; ++q) tmp[q] *= xs;
        V v;
        __builtin_memcpy(&v, tmp, sizeof(V));
        *reinterpret_cast<V *>(s_x + (i * kSellTpb + t) * VEC) = v;
      }
      return;
    }
    const int c_last = x_w - VEC;
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int c = (i * kSellTpb + t) * VEC;
      T in[VEC], out[VEC];
      __builtin_memcpy(in, &xreg[i], sizeof(V));
      const int sh = (x_w >= VEC && c > c_last) ? c - c_last : 0;   // 0: the vector is what it loaded
#pragma unroll
      for (int q = 0; q < VEC; ++q) {
        T e = in[q];
#pragma unroll
        for (int u = 1; u < VEC; ++u)
          if (sh == u) e = (q + u < VEC) ? in[q + u] : static_cast<T>(0);
        if (sh >= VEC) e = 0;
        out[q] = e * xs;
      }
      V v;
      __builtin_memcpy(&v, out, sizeof(V));
      *reinterpret_cast<V *>(s_x + c) = v;
    }
  };
  // The wavefront's batches of ALL tiles of the workgroup form one sequence (every wavefront has
  // the same number per tile, so all of them cross a tile boundary at the same step).  The ring is
  // filled with the first NB batches; every step consumes the oldest batch and refills its slot
  // with the next batch of the sequence -- ALL batch loads are unconditional (past the end the
  // last batch is requested again), so the number in flight is a compile-time constant and the
  // compiler waits with a counted vmcnt instead of draining the queue at every step.
  int tb = 0;   // batches of this wavefront over the whole group
  for (int cb = cb0; cb < cb1; ++cb) tb += (tu(cb + 1) - tu(cb)) / (kSellWaves * UB);
  int f_cb = cb0 - 1, f_b = 0, f_nb = 0;
  size_t f_e0 = 0;
  auto f_next = [&](size_t &e0, bool &first) {
    if (f_b >= f_nb) {   // to the next non-empty tile, if there is one
      int nx = f_cb + 1;
      while (nx < cb1 && tu(nx) == tu(nx + 1)) ++nx;
      if (nx < cb1) {
        const int u0 = tu(nx), u1 = tu(nx + 1);
        const int K = (u1 - u0) / kSellWaves;   // wave-rows per wavefront in this tile
        f_cb = nx;
        f_nb = K / UB;
        f_b = 0;
        f_e0 = static_cast<size_t>(u0) * 64 + static_cast<size_t>(wave) * (UB * 64) + lane * UB;
      } else {
        f_b = f_nb > 0 ? f_nb - 1 : 0;   // past the end: the last batch again (never consumed)
        f_cb = cb1;
      }
    }
    e0 = f_e0 + static_cast<size_t>(f_b) * (kSellWaves * UB * 64);
    first = f_b == 0 && f_cb < cb1;
    ++f_b;
  };
  SellBatch<T, TWO> B[NB];
  bool first[NB];
  x_prefetch();
  if (tb > 0) {
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      size_t e0;
      f_next(e0, first[q]);
      fetch(e0, B[q]);
    }
    for (int i = 0; i < tb; i += NB) {
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        if (i + q < tb) {   // uniform
          if (first[q]) {
            // first batch of a tile (the same step for every wavefront): swap the x slice
            __syncthreads();   // the previous tile's gathers and row-sum updates are done
            x_store();
            x_prefetch();
            __syncthreads();
          }
          consume(B[q]);
        }
        size_t e0;
        f_next(e0, first[q]);
        fetch(e0, B[q]);
      }
    }
  }
  if constexpr (kNoLds) {   // (ablation build: keeps the running sums, and with them the loads and products, alive)
    if (acc == static_cast<T>(123.456)) s_y[t] = acc;
  }
  __syncthreads();
}

// DIRECT (ncg == 1): the row functor runs here; otherwise part[cg * nrows + row] receives the
// column group's partial sums.  `guard` (may be null): a device word written by an EARLIER launch;
// non-zero turns this launch into a no-op (the device-resident CG loop, cg_fused.h: the host
// enqueues a loop's worth of launches without reading anything back).
template <typename T, bool SQ, bool DIRECT, typename Op>
__global__ void __launch_bounds__(kSellTpb) spmv_sell_kernel(SellView<T> A, const T *__restrict__ x,
                                                             const double *x_nrm2, Op op, T *__restrict__ part,
                                                             double *scalar_partials, const double *guard) {
  if (guard && *guard != 0.0) return;
  constexpr int NS = Op::NS > 0 ? Op::NS : 1;
  constexpr int BW = SellCfg<T>::BW;
  extern __shared__ __attribute__((aligned(16))) unsigned char sell_smem[];
  T *s_x = reinterpret_cast<T *>(sell_smem);   // [BW]
  T *s_y = s_x + BW;                           // [RR]
  __shared__ double s_red[NS * kSellWaves];
  const int t = threadIdx.x;
  unsigned long long t_start = 0;
  if (A.stamps) t_start = wall_clock64();
  // consecutive workgroups (round-robin over the XCDs) take the column groups of one row range:
  // with 8 groups every XCD keeps re-reading the same eighth of x from its own L2
  const int cg = static_cast<int>(blockIdx.x) % A.ncg;
  const int rr = static_cast<int>(blockIdx.x) / A.ncg;
  const int row0 = rr * A.rr_rows;
  const int nr = min(A.rr_rows, A.nrows - row0);
  T xs = 1;
  if (x_nrm2) xs = static_cast<T>(1.0 / sqrt(*x_nrm2));
  // (one uniform branch per launch: the two storage formats differ in what a batch fetches and how it names its rows)
  if (A.two) sell_row_sums<T, SQ, true>(A, x, xs, rr, cg, nr, s_x, s_y);
  else sell_row_sums<T, SQ, false>(A, x, xs, rr, cg, nr, s_x, s_y);
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
  if (A.stamps && t == 0) {
    const int cb0 = static_cast<int>(static_cast<long long>(cg) * A.ncb / A.ncg);
    const int cb1 = static_cast<int>(static_cast<long long>(cg + 1) * A.ncb / A.ncg);
    unsigned long long *st = A.stamps + 4 * static_cast<size_t>(blockIdx.x);
    st[0] = t_start;
    st[1] = wall_clock64();
    st[2] = __builtin_amdgcn_s_getreg(6164) & 15u;   // HW_REG_XCC_ID (id 20), bits [3:0]
    st[3] = static_cast<unsigned long long>(sell_uniform_load(A.tile_unit + static_cast<size_t>(rr) * A.ncb + cb1) -
                                            sell_uniform_load(A.tile_unit + static_cast<size_t>(rr) * A.ncb + cb0));
  }
}

// ---------------------------------------------------------------------------------------------
// Build (device).  Temporaries per (tile, local row): cnt (non-zeros, uint16: a tile is at most
// 24576 wide), soff (stream << 23 | offset of the row inside its stream).
// ---------------------------------------------------------------------------------------------
struct SellDims {
  int nrows, ncols, rr_rows, nrr, ncb, bw;
};

// cnt[tile * rr_rows + local row] = non-zeros of that row inside the tile; cnt is zero on entry.  A wavefront per row: in a
// row whose column blocks do not decrease the first element of every run of equal blocks walks forward to the run's end and
// writes its length; any other row is counted by one lane, element by element (as every row was, one thread per row, until
// round 5: 2.3 ms per copy at C4).
__global__ void __launch_bounds__(256) sell_count_kernel(const int *ind, const int *ptr, SellDims D, unsigned short *cnt) {
  const int lane = threadIdx.x & 63;
  const int w = static_cast<int>((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nw = static_cast<int>((gridDim.x * blockDim.x) >> 6);
  for (int r = w; r < D.nrows; r += nw) {
    const int rr = r / D.rr_rows, lr = r - rr * D.rr_rows;
    const int p0 = ptr[r], p1 = ptr[r + 1];
    bool mono = true;
    for (int k = p0 + 1 + lane; k < p1; k += 64) mono = mono && (ind[k] / D.bw >= ind[k - 1] / D.bw);
    mono = __all(mono);
    if (mono) {
      for (int k = p0 + lane; k < p1; k += 64) {
        const int cb = ind[k] / D.bw;
        if (k > p0 && ind[k - 1] / D.bw == cb) continue;   // not the first of its run
        int e = k + 1;
        while (e < p1 && ind[e] / D.bw == cb) ++e;
        cnt[(static_cast<size_t>(rr) * D.ncb + cb) * D.rr_rows + lr] = static_cast<unsigned short>(e - k);
      }
    } else if (lane == 0) {
      for (int k = p0; k < p1; ++k) {
        const int cb = ind[k] / D.bw;
        cnt[(static_cast<size_t>(rr) * D.ncb + cb) * D.rr_rows + lr] += 1;
      }
    }
  }
}

// One workgroup (256 threads) per tile.  The non-empty rows are ordered by class -- rows longer
// than 32 first, then lengths 32 down to 1, rows of a class in row order (a stable counting sort)
// -- and dealt to the 512 streams bank-aware (below).  Outputs:
// soff[tile * rr_rows + row] = stream << 23 | offset of the row in its stream, tile_nu[tile] =
// 64-element units of the tile = 8 * K, K = longest stream rounded up to the batch size.
// *err |= 4 if an offset does not fit its 23 bits (the caller then keeps the plain CSR kernel).
//
// The same deal laid out for the "two id slots" format (soff2, tile_nu2; SellView::two): a stream takes
// its rows from both ends of its list -- a short row from the back while the current batch of 4 has
// room for it and has seen fewer than two row ends, a long one from the front otherwise (it runs
// on into a later batch, whose first end it is), and zeros up to the batch boundary when two rows
// have ended and nothing left is long enough to leave the batch -- so that no batch holds more than
// two ends.  soff2 = stream << 23 | (second end of its batch) << 22 | offset.  A row's elements stay
// together and in order, so both layouts add up every row sum in the same order: same bits.
// With ~2.2 non-zeros per (row, tile) (C4) the streams need no extra padding for this; with mostly
// single-element rows they need a lot (the caller compares the two totals and takes the smaller
// matrix).  *err |= 8: an offset of this layout does not fit 22 bits (the caller keeps the tags).
constexpr int kSellClasses = kSellLmax + 2;   // 0 unused, 1..32, 33 = longer
// Bank-aware dealing (round 6).  A row end is flushed into its row sum with a dependent LDS read-add-write of
// s_y[local row]; a wavefront's 64 flushes of a batch position used to land on random banks, and those conflicts were
// measured as 10 % of the SpMV (123.7 -> 111.6 us with them compiled out, profiles/NOTES_r06.md section 5).  The LDS
// serves a 4-byte access in two groups of 32 lanes on 32 banks (8-byte: the same residues), so a flush is conflict-free
// whenever every lane's row satisfies  local row mod 32 == lane mod 32  -- whatever the other lanes flush at that moment,
// including the lanes of the flat flush that hit their own scratch word (RR + thread: the same residue).  The planner
// therefore deals the rows of residue class q only to the 16 streams {8 wavefronts} x {lanes q, q + 32}:
//   * rows in class order (long first), stably split by residue;
//   * one lane per residue deals its rows onto its 16 streams in serpentine order, a row only while its stream
//     stays within the budget B = the tile's elements / 512, rounded up to a batch (at least the longest row); a row that
//     does not fit goes to an overflow list (< 1 % of the rows at C4: the residue totals differ by a few per cent);
//   * overflow rows go, first fit, to any stream with room (their flushes may conflict -- they are few).
// A row's elements stay together, in order, in ONE lane's stream, so every row sum is added up exactly as before: the
// assignment changes the layout, not a single bit of the results.
constexpr int kSellResidues = 32;
constexpr int kSellOvfCap = 2048;
constexpr int kSellFillCb = 1024;   // sell_fill_kernel: column blocks it keeps running counts for (unsorted rows)
constexpr size_t sell_plan_lds(int rr_rows) { return 3 * static_cast<size_t>(rr_rows) * sizeof(unsigned short); }
__global__ void __launch_bounds__(256) sell_plan_kernel(const unsigned short *cnt, SellDims D, int *tile_nu,
                                                        unsigned *soff, int *tile_nu2, unsigned *soff2, int *err) {
  extern __shared__ unsigned short s_dyn[];               // three arrays of rr_rows entries
  unsigned short *s_sorted = s_dyn;                       // rows in class order; later (stream << 7 | position) per ROW
  unsigned short *s_xrow = s_dyn + D.rr_rows;             // rows split by residue; later the rows by stream
  unsigned short *s_len = s_dyn + 2 * D.rr_rows;          // the rows' lengths in this tile, by row: read from memory once
  unsigned short *s_sp = s_sorted, *s_by = s_xrow;
  __shared__ unsigned short s_hist[kSellClasses][256];    // per thread and class (then residue): entries, then their exclusive prefix
  __shared__ int s_tot[kSellClasses], s_cbase[kSellClasses];
  __shared__ int s_max[4], s_max2[4], s_sum[4];
  __shared__ int s_start[kSellStreams + 1], s_ld[kSellStreams];
  __shared__ unsigned short s_cnt[kSellStreams], s_ovf[kSellOvfCap], s_ovl[kSellOvfCap];
  __shared__ int s_novf, s_budget;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int rpt = (D.rr_rows + 255) / 256;               // consecutive rows per thread
  // exclusive scans of s_hist[c][0..255] for c = c0, c0 + 1, ..., c1 - 1: one wavefront per c (4 entries per lane)
  auto scan_hist = [&](int c0, int c1) {
    for (int c = c0 + wave; c < c1; c += 4) {
      int v[4], sum = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q] = s_hist[c][lane * 4 + q];
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
        s_hist[c][lane * 4 + q] = static_cast<unsigned short>(run);
        run += v[q];
      }
      if (lane == 63) s_tot[c] = inc;
    }
  };
  for (int tile = blockIdx.x; tile < D.nrr * D.ncb; tile += gridDim.x) {
    const int rr = tile / D.ncb;
    const int nr = min(D.rr_rows, D.nrows - rr * D.rr_rows);
    const unsigned short *tc = cnt + static_cast<size_t>(tile) * D.rr_rows;
    unsigned *to = soff + static_cast<size_t>(tile) * D.rr_rows;
    unsigned *to2 = soff2 ? soff2 + static_cast<size_t>(tile) * D.rr_rows : nullptr;
    for (int c = 0; c < kSellClasses; ++c) s_hist[c][t] = 0;
    // the tile's row lengths: read from memory ONCE, coalesced (every later phase reads them from LDS -- a thread walking
    // its 64 consecutive rows through global memory pays a memory latency per row)
    for (int r = t; r < nr; r += 256) s_len[r] = tc[r];
    __syncthreads();
    const int r_lo = t * rpt, r_hi = min(nr, r_lo + rpt);
    int my_sum = 0, my_max = 0;
    for (int r = r_lo; r < r_hi; ++r) {
      const int len = s_len[r];
      if (len == 0) continue;
      s_hist[len <= kSellLmax ? len : kSellLmax + 1][t] += 1;
      my_sum += len;
      my_max = max(my_max, len);
    }
    for (int off = 32; off > 0; off >>= 1) {
      my_sum += __shfl_xor(my_sum, off, 64);
      my_max = max(my_max, __shfl_xor(my_max, off, 64));
    }
    if (lane == 0) { s_sum[wave] = my_sum; s_max[wave] = my_max; }
    if (t == 0) s_novf = 0;
    for (int i = t; i < kSellStreams; i += 256) { s_cnt[i] = 0; s_ld[i] = 0; }
    __syncthreads();
    scan_hist(1, kSellClasses);
    __syncthreads();
    if (t == 0) {
      int b = 0;
      for (int c = kSellLmax + 1; c >= 1; --c) {
        s_cbase[c] = b;
        b += s_tot[c];
      }
      s_cbase[0] = b;   // non-empty rows of the tile
      const int total = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
      const int lmax = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
      const int mean = (total + kSellStreams - 1) / kSellStreams;
      s_budget = (max(mean, lmax) + kSellUB - 1) / kSellUB * kSellUB;
    }
    __syncthreads();
    {
      // (dynamic index into a small local array: scratch; one-time code)
      int rank[kSellClasses];
#pragma unroll
      for (int c = 0; c < kSellClasses; ++c) rank[c] = 0;
      for (int r = r_lo; r < r_hi; ++r) {
        const int len = s_len[r];
        if (len == 0) continue;
        const int c = len <= kSellLmax ? len : kSellLmax + 1;
        const int k = s_cbase[c] + s_hist[c][t] + rank[c];
        rank[c] += 1;
        s_sorted[k] = static_cast<unsigned short>(r);
      }
    }
    __syncthreads();
    const int nne = s_cbase[0];
    // ---- the class-ordered rows, stably split by residue (local row mod 32): a counting sort over chunks of the list
    const int cpt = (nne + 255) / 256;
    const int k_lo = min(nne, t * cpt), k_hi = min(nne, k_lo + cpt);
    for (int q = 0; q < kSellResidues; ++q) s_hist[q][t] = 0;
    for (int k = k_lo; k < k_hi; ++k) s_hist[s_sorted[k] & (kSellResidues - 1)][t] += 1;
    __syncthreads();
    scan_hist(0, kSellResidues);
    __syncthreads();
    if (t == 0) {
      int b = 0;
      for (int q = 0; q < kSellResidues; ++q) {
        s_cbase[q] = b;
        b += s_tot[q];
      }
      s_cbase[kSellResidues] = b;
    }
    __syncthreads();
    {
      int rank[kSellResidues];
#pragma unroll
      for (int q = 0; q < kSellResidues; ++q) rank[q] = 0;
      for (int k = k_lo; k < k_hi; ++k) {
        const int r = s_sorted[k], q = r & (kSellResidues - 1);
        const int d = s_cbase[q] + s_hist[q][t] + rank[q];
        rank[q] += 1;
        s_xrow[d] = static_cast<unsigned short>(r);
      }
    }
    __syncthreads();
    // ---- wavefront 0, one lane per residue q (lanes 0..31): the lane owns the 16 streams {w * 64 + q, w * 64 + q + 32} --
    // lanes q and q + 32 of every wavefront of the SpMV -- and deals the rows of residue q onto them in serpentine order
    // (the list is sorted, long rows first: every stream gets one row of every band of 16), a row only while its stream
    // stays within the budget.  Loads and counts live in LDS (stream s at word s: lane q touches banks q only).  The rows
    // that do not fit are then placed first-fit in stream order by the whole wavefront (lane L tests the streams
    // w * 64 + L), the search starting behind the last placement: ONE overflow row per stream and sweep -- several short
    // rows at the end of one stream would cost the two-slot layout a batch (two ends per batch), and the longest stream
    // sets the tile's length (measured: 1.11 against 1.065 stored elements per non-zero at C4 when they pile up).
    if (wave == 0) {
      const int B = s_budget;
      if (lane < kSellResidues) {
        const int q = lane;
        constexpr int NSL = kSellStreams / kSellResidues;   // 16 streams per residue
        int p = 0;
        const int k_end = s_cbase[q + 1];
        int r_nx = s_cbase[q] < k_end ? s_xrow[s_cbase[q]] : 0;
        int len_nx = s_len[r_nx];
        for (int k = s_cbase[q]; k < k_end; ++k, ++p) {
          const int r = r_nx, len = len_nx;   // (the next row and its length are requested a row ahead: two dependent LDS reads)
          r_nx = k + 1 < k_end ? s_xrow[k + 1] : 0;
          len_nx = s_len[r_nx];
          const int at = p & (NSL - 1), j = ((p / NSL) & 1) ? NSL - 1 - at : at;
          const int sx = (j >> 1) * 64 + q + 32 * (j & 1);
          const int load = s_ld[sx], c = s_cnt[sx];
          if (load + len <= B && c < 127) {
            s_ld[sx] = load + len;
            s_cnt[sx] = static_cast<unsigned short>(c + 1);
            s_sp[r] = static_cast<unsigned short>((sx << 7) | c);
          } else {
            const int o = atomicAdd(&s_novf, 1);
            if (o < kSellOvfCap) {
              s_ovf[o] = static_cast<unsigned short>(r);
              s_ovl[o] = static_cast<unsigned short>(len);
            } else {   // (list full: a pathological tile) the row stays with its stream, over the budget
              if (c >= 127) atomicOr(err, 4);   // cannot be laid out: the caller keeps the plain kernel
              s_ld[sx] = load + len;
              s_cnt[sx] = static_cast<unsigned short>(min(c + 1, 127));
              s_sp[r] = static_cast<unsigned short>((sx << 7) | min(c, 127));
            }
          }
        }
      }
      const int novf = min(__shfl(s_novf, 0, 64), kSellOvfCap);   // (every lane's appends are done: same wavefront)
      int cur = 0;
      for (int o = 0; o < novf; ++o) {
        const int r = s_ovf[o], len = s_ovl[o];
        // the first stream at or behind the cursor with room, wavefront by wavefront of the SpMV (stream w * 64 + lane):
        // one ballot per 64 streams, usually the first or second finds one
        const int cw = cur >> 6, cl = cur & 63;
        int sx = -1;
        for (int step = 0; step <= kSellStreams / 64 && sx < 0; ++step) {
          const int w = (cw + step) & (kSellStreams / 64 - 1);
          const int mine = w * 64 + lane;
          const bool room = s_ld[mine] + len <= B && s_cnt[mine] < 127;
          unsigned long long mask = __ballot(room);
          if (step == 0) mask &= ~0ull << cl;                                  // lanes at or behind the cursor
          if (step == kSellStreams / 64) mask &= cl ? ~(~0ull << cl) : 0ull;   // the cursor's wavefront again: lanes before it
          if (mask) sx = w * 64 + __ffsll(static_cast<long long>(mask)) - 1;
        }
        if (sx < 0) {   // nothing has room: the least loaded stream of all
          int lkey = 0x7fffffff;
#pragma unroll
          for (int w = 0; w < kSellStreams / 64; ++w) {
            const int m2 = w * 64 + lane;
            const int lc = (s_ld[m2] << 9) | m2;
            lkey = (s_cnt[m2] < 127 && lc < lkey) ? lc : lkey;
          }
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) lkey = min(lkey, __shfl_xor(lkey, off, 64));
          if (lkey == 0x7fffffff) { atomicOr(err, 4); lkey = 0; }
          sx = lkey & (kSellStreams - 1);
        }
        cur = (sx + 1) & (kSellStreams - 1);
        if (lane == 0) {
          const int c = s_cnt[sx];
          s_sp[r] = static_cast<unsigned short>((sx << 7) | min(c, 127));
          s_ld[sx] += len;
          s_cnt[sx] = static_cast<unsigned short>(min(c + 1, 127));
        }
      }
    }
    __syncthreads();
    // ---- rows by stream: exclusive prefix of the counts (one wavefront, 8 streams per lane), then every row to its place
    if (wave == 0) {
      int v[8], sum = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        v[q] = s_cnt[lane * 8 + q];
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
      for (int q = 0; q < 8; ++q) {
        s_start[lane * 8 + q] = run;
        run += v[q];
      }
      if (lane == 63) s_start[kSellStreams] = inc;
    }
    __syncthreads();
    for (int r = r_lo; r < r_hi; ++r) {
      if (s_len[r] == 0) continue;
      const int sp = s_sp[r];
      s_by[s_start[sp >> 7] + (sp & 127)] = static_cast<unsigned short>(r);
    }
    __syncthreads();
    int longest = 0, longest2 = 0;
    for (int sidx = t; sidx < kSellStreams; sidx += 256) {
      const int base = s_start[sidx], np = s_start[sidx + 1] - base;   // this stream's rows: s_by[base .. base + np), long first
      int run = 0;
      for (int p = 0; p < np; ++p) {
        const int r = s_by[base + p];
        if (run >= (1 << kSellOffBits)) atomicOr(err, 4);
        to[r] = (static_cast<unsigned>(sidx) << kSellOffBits) | static_cast<unsigned>(run & ((1 << kSellOffBits) - 1));
        run += s_len[r];
      }
      longest = max(longest, run);
      if (to2) {
        static_assert(kSellUB == 4, "the two-slot layout counts row ends per batch of 4");
        int i = 0, j = np - 1, pos = 0, ends = 0;   // front (long rows), back (short rows), stream position, ends in its batch
        int ri = 0, li = 0, rj = 0, lj = 0;
        if (np > 0) {
          ri = s_by[base]; li = s_len[ri];
          rj = s_by[base + j]; lj = s_len[rj];
        }
        while (i <= j) {
          const int slot = pos & 3;
          if (slot == 0) ends = 0;
          const int left = 4 - slot;
          int r, len;
          if (ends < 2 && lj <= left) {          // a short row that ends inside this batch
            r = rj; len = lj;
            --j;
            if (i <= j) { rj = s_by[base + j]; lj = s_len[rj]; }
          } else if (ends < 2 || li > left) {    // the front row: it ends in a later batch, or this batch has room for its end
            r = ri; len = li;
            ++i;
            if (i <= j) { ri = s_by[base + i]; li = s_len[ri]; }
          } else {                               // two ends and nothing leaves the batch: zeros up to its boundary
            pos += left;
            continue;
          }
          const int e = pos + len - 1;
          const bool same = (e >> 2) == (pos >> 2);
          const unsigned second = same ? static_cast<unsigned>(ends) : 0u;
          ends = same ? ends + 1 : 1;
          if (pos >= (1 << kSellOff2Bits)) atomicOr(err, 8);
          to2[r] = (static_cast<unsigned>(sidx) << kSellOffBits) | (second << kSellOff2Bits) |
                   static_cast<unsigned>(pos & ((1 << kSellOff2Bits) - 1));
          pos += len;
        }
        longest2 = max(longest2, pos);
      }
    }
    for (int off = 32; off > 0; off >>= 1) {
      longest = max(longest, __shfl_xor(longest, off, 64));
      longest2 = max(longest2, __shfl_xor(longest2, off, 64));
    }
    __syncthreads();   // (s_max was read for the budget)
    if (lane == 0) { s_max[wave] = longest; s_max2[wave] = longest2; }
    __syncthreads();
    if (t == 0) {
      const int K = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
      tile_nu[tile] = (K + kSellUB - 1) / kSellUB * kSellUB * kSellWaves;
      if (tile_nu2) {
        const int K2 = max(max(s_max2[0], s_max2[1]), max(s_max2[2], s_max2[3]));
        tile_nu2[tile] = (K2 + kSellUB - 1) / kSellUB * kSellUB * kSellWaves;
      }
    }
    __syncthreads();
  }
}

// copies every non-zero to its place: element j (in CSR order) of a row's non-zeros inside a tile goes to position
// offset + j of the row's stream.  One WAVEFRONT per row, a lane per non-zero: j is the number of the row's earlier
// non-zeros in the same column block -- for a row whose column blocks do not decrease (every row of the transposed
// copy, every row of a sorted input) the distance to the start of its run of equal blocks, found by walking back
// (~2 steps at C4); for any other row a count over the row's prefix.  (One thread per row with a cursor per (row, tile)
// in global memory -- 50 to 200 dependent read-modify-writes per thread -- took 8.7 ms per copy at C4.)
// sloc == nullptr: values only (after a rescaling of the CSR copy).
template <typename T>
__global__ void __launch_bounds__(256) sell_fill_kernel(const T *val, const int *ind, const int *ptr, SellDims D,
                                                        const unsigned short *cnt, const unsigned *soff, const int *tile_unit,
                                                        T *sval, unsigned short *sloc, unsigned short *srid, unsigned *dst_out,
                                                        int two) {
  // (rows whose column blocks are NOT in order -- an unsorted input: one running count per column block and wavefront in
  // LDS, the row taken 64 non-zeros at a time: an element's position is the count so far plus its rank among the lower
  // lanes of its chunk with the same block.  Linear in the row's length; the count over the row's prefix it replaces
  // was quadratic, minutes for a row of 1e5 unsorted non-zeros.  More than kSellFillCb column blocks: the prefix count.)
  __shared__ int s_run[4][kSellFillCb];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int w = static_cast<int>((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nw = static_cast<int>((gridDim.x * blockDim.x) >> 6);
  for (int r = w; r < D.nrows; r += nw) {
    const int rr = r / D.rr_rows, lr = r - rr * D.rr_rows;
    const int p0 = ptr[r], p1 = ptr[r + 1];
    bool mono = true;
    for (int k = p0 + 1 + lane; k < p1; k += 64) mono = mono && (ind[k] / D.bw >= ind[k - 1] / D.bw);
    mono = __all(mono);
    const bool counted = !mono && D.ncb <= kSellFillCb;
    if (counted)
      for (int q = lane; q < D.ncb; q += 64) s_run[wv][q] = 0;
    for (int k0 = p0; k0 < p1; k0 += 64) {   // (uniform trip count: the whole wavefront takes part in the shuffles)
      const int k = k0 + lane;
      const bool have = k < p1;
      const int c = have ? ind[k] : 0, cb = have ? c / D.bw : -1;
      int j = 0;
      if (mono) {
        if (have) {
          int kk = k;
          while (kk > p0 && ind[kk - 1] / D.bw == cb) --kk;
          j = k - kk;
        }
      } else if (counted) {
        int rank = 0, tot = 0;
        for (int l = 0; l < 64; ++l) {
          const int ocb = __shfl(cb, l, 64);
          rank += (l < lane && ocb == cb) ? 1 : 0;
          tot += (ocb == cb) ? 1 : 0;
        }
        const int base = have ? s_run[wv][cb] : 0;
        j = base + rank;
        if (have && rank == tot - 1) s_run[wv][cb] = base + tot;   // (the last lane of each block's group: one writer per word)
      } else if (have) {
        for (int q = p0; q < k; ++q) j += (ind[q] / D.bw == cb) ? 1 : 0;
      }
      if (!have) continue;
      const int tile = rr * D.ncb + cb;
      const size_t idx = static_cast<size_t>(tile) * D.rr_rows + lr;
      const unsigned so = soff[idx];
      const int sidx = static_cast<int>(so >> kSellOffBits);
      const int off = static_cast<int>(so & ((1u << (two ? kSellOff2Bits : kSellOffBits)) - 1));
      const int u0 = tile_unit[tile];
      const int k_el = off + j;   // position in the lane's stream: step k_el / UB, wavefront-, then lane-major inside the step
      const size_t dst = static_cast<size_t>(u0) * 64 +
                         (static_cast<size_t>(k_el / kSellUB) * kSellWaves + (sidx >> 6)) * (64 * kSellUB) +
                         (sidx & 63) * kSellUB + (k_el % kSellUB);
      sval[dst] = val[k];
      if (dst_out) dst_out[k] = static_cast<unsigned>(dst);
      if (sloc) {
        const bool last = j + 1 == cnt[idx];
        const int lc = c - cb * D.bw;
        if (!two) {
          sloc[dst] = static_cast<unsigned short>(lc);
          if (last) srid[dst] = static_cast<unsigned short>(lr);
        } else {
          sloc[dst] = static_cast<unsigned short>(lc | (last ? kSellEndBit : 0));
          if (last) srid[(dst - static_cast<size_t>(k_el % kSellUB)) / 2 + ((so >> kSellOff2Bits) & 1u)] = static_cast<unsigned short>(lr);
        }
      }
    }
  }
}

// sval[dst[k]] = val[k]: the values again (after a rescaling of the CSR copy), through the positions
// sell_fill_kernel recorded -- coalesced reads, no per-(row, tile) bookkeeping
template <typename T>
__global__ void __launch_bounds__(256) sell_refill_kernel(const T *__restrict__ val, const unsigned *__restrict__ dst,
                                                          size_t nnz, T *sval) {
  for (size_t k = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; k < nnz; k += static_cast<size_t>(gridDim.x) * 256)
    sval[dst[k]] = val[k];
}

}  // namespace pogs_amd
