// Does the PLACEMENT of the SpMV's load streams matter?  The tiled lane-stream SpMV (csrc/sell.h) has ~248 workgroups
// of 512 threads, each walking its own contiguous piece of three arrays (16 B values + 8 B columns + 8 B tags per lane and
// batch, 8 batches in flight per wavefront) -- 744 separate sequential streams, 6.3 TB/s of real traffic at C4 -- while the
// best plain read pattern (all workgroups side by side, profiles/r03_read_bw.txt) reaches 7.0 TB/s.  This probe runs the
// SpMV's load stream alone (no LDS work, no tiles) in three placements of the same bytes:
//   separate     workgroup g reads [g * S, (g + 1) * S) steps of each array            (the shipped layout)
//   interleaved  step j of workgroup g sits at (j * G + g): all workgroups march through ONE window per array
//   merged       like interleaved, the three arrays of a step in one 16 KB piece
//   hipcc --offload-arch=gfx950 -O3 stream_pattern.hip -o bin/stream_pattern
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
constexpr int TPB = 512, NB = 8;

template <int MODE, int TAGS>
__global__ void __launch_bounds__(TPB) walk(const u4 *__restrict__ val, const u2 *__restrict__ loc, const u2 *__restrict__ rid,
                                            int G, int S, unsigned *out) {
  const int t = threadIdx.x, g = blockIdx.x;
  auto at = [&](int j) -> size_t {   // index of this thread's vector of step j, in units of one vector per thread
    const int jj = j < S ? j : S - 1;
    return MODE == 0 ? (static_cast<size_t>(g) * S + jj) * TPB + t : (static_cast<size_t>(jj) * G + g) * TPB + t;
  };
  u4 v[NB];
  u2 c[NB], r[NB];
  unsigned acc = 0;
  auto fetch = [&](int j, int q) {
    const size_t i = at(j);
    if (MODE == 2) {   // one 16 KB piece per step: [values 8 KB | columns 4 KB | tags 4 KB]
      const char *base = reinterpret_cast<const char *>(val) + (i - t) / TPB * (TPB * (TAGS ? 32 : 24));
      v[q] = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(base) + t);
      c[q] = __builtin_nontemporal_load(reinterpret_cast<const u2 *>(base + TPB * 16) + t);
      if (TAGS) r[q] = __builtin_nontemporal_load(reinterpret_cast<const u2 *>(base + TPB * 24) + t);
    } else {
      v[q] = __builtin_nontemporal_load(val + i);
      c[q] = __builtin_nontemporal_load(loc + i);
      if (TAGS) r[q] = __builtin_nontemporal_load(rid + i);
    }
  };
#pragma unroll
  for (int q = 0; q < NB; ++q) fetch(q, q);
  for (int j = 0; j < S; j += NB) {
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      acc += v[q].x ^ v[q].y ^ v[q].z ^ v[q].w ^ c[q].x ^ c[q].y;
      if (TAGS) acc += r[q].x ^ r[q].y;
      fetch(j + NB + q, q);
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// The same separate streams, plus what a tile boundary adds in the SpMV: every TILE steps each thread requests 9 x 16 B of
// an L2-resident vector (the x slice: 72 KB per workgroup and tile, 246 MB per SpMV on top of the 846 MB of matrix -- but
// from L2, not HBM).  STORE: the slice also goes through LDS with the two barriers of the swap.
template <int STORE>
__global__ void __launch_bounds__(TPB) walk_x(const u4 *__restrict__ val, const u2 *__restrict__ loc, const u2 *__restrict__ rid,
                                              const u4 *__restrict__ xvec, int xvecs, int G, int S, int TILE, unsigned *out) {
  __shared__ u4 s_x[TPB * 9];
  const int t = threadIdx.x, g = blockIdx.x;
  auto at = [&](int j) -> size_t { return (static_cast<size_t>(g) * S + (j < S ? j : S - 1)) * TPB + t; };
  u4 v[NB];
  u2 c[NB], r[NB];
  u4 xr[9];
  unsigned acc = 0;
  auto fetch = [&](int j, int q) {
    const size_t i = at(j);
    v[q] = __builtin_nontemporal_load(val + i);
    c[q] = __builtin_nontemporal_load(loc + i);
    r[q] = __builtin_nontemporal_load(rid + i);
  };
  auto xload = [&](int tile) {
#pragma unroll
    for (int k = 0; k < 9; ++k) xr[k] = xvec[(static_cast<size_t>(tile) * 9 * TPB + k * TPB + t) % xvecs];
  };
  xload(0);
#pragma unroll
  for (int q = 0; q < NB; ++q) fetch(q, q);
  int next_tile = 0, tile = 0;
  for (int j = 0; j < S; j += NB) {
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      if (j + q == next_tile) {   // (uniform)
        if (STORE) {
          __syncthreads();
#pragma unroll
          for (int k = 0; k < 9; ++k) s_x[k * TPB + t] = xr[k];
        } else {
#pragma unroll
          for (int k = 0; k < 9; ++k) acc += xr[k].x;
        }
        xload(++tile);
        if (STORE) __syncthreads();
        next_tile += TILE;
      }
      acc += v[q].x ^ v[q].y ^ v[q].z ^ v[q].w ^ c[q].x ^ c[q].y ^ r[q].x ^ r[q].y;
      if (STORE) acc += s_x[(v[q].x & 0xFFF) % (TPB * 9)].x;
      fetch(j + NB + q, q);
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <typename F>
static double time_us(F &&launch, int reps = 20) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) launch();
  hipEventRecord(a, 0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1e3 / reps;
}

int main() {
  // C4: 1.05e8 stored elements = 51270 steps of 2048 elements; 248 workgroups -> 207 steps each
  const int S = 208;
  void *val, *loc, *rid;
  unsigned *out;
  const size_t steps_max = 512ull * S;
  hipMalloc(&val, steps_max * TPB * 32);
  hipMalloc(&loc, steps_max * TPB * 8);
  hipMalloc(&rid, steps_max * TPB * 8);
  hipMalloc(&out, 64);
  hipMemset(val, 1, steps_max * TPB * 32);
  hipMemset(loc, 1, steps_max * TPB * 8);
  hipMemset(rid, 1, steps_max * TPB * 8);
  {
    void *xv;
    const int xvecs = 131072;   // 2 MB: the C4 x vector, L2-resident
    hipMalloc(&xv, static_cast<size_t>(xvecs) * 16);
    hipMemset(xv, 1, static_cast<size_t>(xvecs) * 16);
    const int G = 248;
    const double gb3 = static_cast<double>(G) * S * TPB * 32 / 1e9;
    for (int tile : {15, 30, 1000000}) {
      double us = time_us([&] {
        hipLaunchKernelGGL((walk_x<0>), dim3(G), dim3(TPB), 0, 0, static_cast<const u4 *>(val), static_cast<const u2 *>(loc),
                           static_cast<const u2 *>(rid), static_cast<const u4 *>(xv), xvecs, G, S, tile, out);
      });
      std::printf("G %3d  separate + x slice every %7d steps, no LDS   %7.1f us  %6.0f GB/s (matrix bytes)\n", G, tile, us, gb3 / us * 1e6);
      us = time_us([&] {
        hipLaunchKernelGGL((walk_x<1>), dim3(G), dim3(TPB), 0, 0, static_cast<const u4 *>(val), static_cast<const u2 *>(loc),
                           static_cast<const u2 *>(rid), static_cast<const u4 *>(xv), xvecs, G, S, tile, out);
      });
      std::printf("G %3d  separate + x slice every %7d steps, LDS swap %7.1f us  %6.0f GB/s (matrix bytes)\n", G, tile, us, gb3 / us * 1e6);
    }
  }
  for (int G : {248, 256}) {
    const int Sg = G <= 256 ? S : S / 2;
    const double gb3 = static_cast<double>(G) * Sg * TPB * 32 / 1e9, gb2 = static_cast<double>(G) * Sg * TPB * 24 / 1e9;
#define RUN(MODE, TAGS, name)                                                                                         \
  {                                                                                                                   \
    const double us = time_us([&] {                                                                                   \
      hipLaunchKernelGGL((walk<MODE, TAGS>), dim3(G), dim3(TPB), 0, 0, static_cast<const u4 *>(val),                  \
                         static_cast<const u2 *>(loc), static_cast<const u2 *>(rid), G, Sg, out);                     \
    });                                                                                                               \
    std::printf("G %3d  %-28s %7.1f us  %6.0f GB/s\n", G, name, us, (TAGS ? gb3 : gb2) / us * 1e6);                   \
  }
    RUN(0, 1, "separate, 3 arrays (8 B/nz)")
    RUN(1, 1, "interleaved, 3 arrays")
    RUN(2, 1, "merged 16 KB steps")
    RUN(0, 0, "separate, 2 arrays (6 B/nz)")
    RUN(1, 0, "interleaved, 2 arrays")
    RUN(2, 0, "merged 12 KB steps")
  }
  return 0;
}
