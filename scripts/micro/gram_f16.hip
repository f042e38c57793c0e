// Development micro-benchmark: G = P^T P (lower 128 x 128 tiles) on the fp16 matrix cores: every
// fp32 operand (scaled by a power of two into fp16 range) is split into two fp16 parts
// (a s = h + l, 11 + 11 mantissa bits) and the three products hh, hl, lh are accumulated in fp32
// by v_mfma_f32_32x32x16_f16; the result is unscaled by 1 / s^2.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifndef REGSTAGES
#define REGSTAGES 1
#endif
constexpr int BM = 128, BK = 16, GT = 256, kNumXcd = 8;

struct GArgs {
  const float *P; size_t ld; int K, N;   // P is K x N row-major
  float *C; size_t ldc;                  // N x N, lower tiles
  int ksplit, kchunk; size_t cstride;
  int flush;   // k-tiles between two-level accumulator flushes (0: never)
  float scale; // power of two
};

// 8 fp32 -> two vectors of 8 fp16: h = fp16(a s), l = fp16(a s - h)
__device__ __forceinline__ void split8(const float (&v)[8], float sc, f16x8 &h, f16x8 &l) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float a = v[q] * sc;
    const _Float16 hh = static_cast<_Float16>(a);
    h[q] = hh;
    l[q] = static_cast<_Float16>(a - static_cast<float>(hh));
  }
}

__global__ void __launch_bounds__(GT) __attribute__((amdgpu_waves_per_eu(3))) gram_f16_kernel(GArgs g) {
  // [stage][operand][part][k8][i] 16-byte vectors
  __shared__ __attribute__((aligned(16))) f16x8 sh[2][2][2][2][BM];
  const int tm = (g.N + BM - 1) / BM;
  const int ntiles = tm * (tm + 1) / 2;
  const int nunits = ntiles * g.ksplit;
  const int per_xcd = (nunits + kNumXcd - 1) / kNumXcd;
  const int unit = (blockIdx.x % kNumXcd) * per_xcd + blockIdx.x / kNumXcd;
  if (unit >= nunits || (int)(blockIdx.x / kNumXcd) >= per_xcd) return;
  const int ks = unit / ntiles, tile = unit % ntiles;
  int ti = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
  while (ti * (ti + 1) / 2 > tile) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
  const int tj = tile - ti * (ti + 1) / 2;
  const int i0 = ti * BM, j0 = tj * BM;
  const int kbeg = ks * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
  float *Cout = g.C + (size_t)ks * g.cstride;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = t & 127, lk8 = t >> 7;           // loader role: column li, k8 group lk8
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int r32 = lane & 31, kh = lane >> 5;

  floatx16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const bool diag = (ti == tj);
  float va0[8], vb0[8], va1[8], vb1[8];
  auto gload = [&](int k0, float (&va)[8], float (&vb)[8]) {
    const int gi = i0 + li, gj = j0 + li;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = k0 + lk8 * 8 + q;
      const bool kok = k < kend;
      va[q] = (kok && gi < g.N) ? g.P[(size_t)k * g.ld + gi] : 0.f;
      if (!diag) vb[q] = (kok && gj < g.N) ? g.P[(size_t)k * g.ld + gj] : 0.f;
    }
  };
  auto lstore = [&](int st, const float (&va)[8], const float (&vb)[8]) {
    f16x8 h, l;
    split8(va, g.scale, h, l);
    sh[st][0][0][lk8][li] = h; sh[st][0][1][lk8][li] = l;
    if (!diag) {
      split8(vb, g.scale, h, l);
      sh[st][1][0][lk8][li] = h; sh[st][1][1][lk8][li] = l;
    }
  };
  const int bop = diag ? 0 : 1;
  auto compute = [&](int st) {
    f16x8 A[2][2], B[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        A[a][p] = sh[st][0][p][kh][wm + a * 32 + r32];
        B[a][p] = sh[st][bop][p][kh][wn + a * 32 + r32];
      }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        floatx16 c = acc[a][b];
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[a][0], B[b][1], c, 0, 0, 0);   // hl
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[a][1], B[b][0], c, 0, 0, 0);   // lh
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[a][0], B[b][0], c, 0, 0, 0);   // hh
        acc[a][b] = c;
      }
  };
  // two-level K-sum without a second accumulator set: every g.flush k-tiles the accumulators
  // are added into the workgroup's own output tile (read-modify-write, no other writer) and
  // cleared.  The MFMA accumulate truncates, so a long chain drifts in proportion to its
  // length; short chains + IEEE adds keep the sum more exact than a sequential fp32 sum.
  bool first = true;
  const float inv2 = 1.0f / (g.scale * g.scale);
  auto flush_out = [&]() {
    // the leading dimension is made opaque here so that the 64 tile addresses are recomputed
    // per flush instead of living in registers across the k loop (occupancy)
    size_t ldc = g.ldc;
    asm volatile("" : "+s"(ldc));
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = i0 + wm + a * 32 + (r / 4) * 8 + kh * 4 + (r % 4);
          const int col = j0 + wn + b * 32 + r32;
          if (row < g.N && col < g.N) {
            float *c = Cout + (size_t)row * ldc + col;
            const float val = acc[a][b][r] * inv2;
            *c = first ? val : *c + val;
          }
          acc[a][b][r] = 0.f;
          if (r % 8 == 7) asm volatile("" ::: "memory");   // 8 tile rows in flight at a time (registers)
        }
    first = false;
  };
  int until = g.flush;
  auto maybe_flush = [&](int steps) {
#ifdef NOFLUSH
    return;
#endif
    if (g.flush <= 0) return;
    until -= steps;
    if (until <= 0) {
      until = g.flush;
      flush_out();
    }
  };

#if REGSTAGES == 2
  // two register stages + two LDS stages; k-tile count rounded up to even (loads past kend are zero)
  const int nk = ((kend - kbeg + BK - 1) / BK + 1) & ~1;
  gload(kbeg, va0, vb0);
  lstore(0, va0, vb0);
  gload(kbeg + BK, va0, vb0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    gload(kbeg + (kt + 2) * BK, va1, vb1);
    compute(0);
    lstore(1, va0, vb0);
    __syncthreads();
    gload(kbeg + (kt + 3) * BK, va0, vb0);
    compute(1);
    lstore(0, va1, vb1);
    __syncthreads();
    maybe_flush(2);
  }
#else
  (void)va1; (void)vb1;
  const int nk = ((kend - kbeg + BK - 1) / BK + 1) & ~1;
  gload(kbeg, va0, vb0);
  lstore(0, va0, vb0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    gload(kbeg + (kt + 1) * BK, va0, vb0);
    compute(0);
    lstore(1, va0, vb0);
    __syncthreads();
    gload(kbeg + (kt + 2) * BK, va0, vb0);
    compute(1);
    lstore(0, va0, vb0);
    __syncthreads();
    maybe_flush(2);
  }
#endif
  flush_out();
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 512;
  const int ksplit = argc > 3 ? atoi(argv[3]) : 1;
  const int flush = argc > 4 ? atoi(argv[4]) : 0;
  const float scale = argc > 5 ? (float)atof(argv[5]) : 4096.0f;
  const bool check = (size_t)K * N <= (1u << 23);
  std::vector<float> P((size_t)K * N);
  unsigned s = 12345;
  for (auto &v : P) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 20001 - 10000) * 1e-4f * (1.0f / sqrtf((float)K)); }
  float *dP, *dC;
  const size_t slab = (size_t)N * N;
  hipMalloc(&dP, P.size() * 4); hipMalloc(&dC, slab * 4 * ksplit);
  hipMemcpy(dP, P.data(), P.size() * 4, hipMemcpyHostToDevice);
  hipMemset(dC, 0, slab * 4 * ksplit);
  GArgs g{dP, (size_t)N, K, N, dC, (size_t)N, ksplit, ((K + ksplit - 1) / ksplit + 31) / 32 * 32, slab, flush, scale};
  const int tm = (N + BM - 1) / BM, nunits = tm * (tm + 1) / 2 * ksplit;
  const int grid = (nunits + kNumXcd - 1) / kNumXcd * kNumXcd;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(gram_f16_kernel, dim3(grid), dim3(GT), 0, 0, g);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("K=%d N=%d ksplit=%d flush=%d: %.3f ms  %.1f TFLOP/s (fp32-equivalent, lower tiles incl. full diagonal tiles)\n", K, N, ksplit, flush, ms,
           (double)K * N * N / (ms * 1e-3) / 1e12);
  }
  if (check) {
    std::vector<float> C(slab * ksplit);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double emax = 0, gmax = 0, e32 = 0;
    for (int i = 0; i < N; i += 7) for (int j = 0; j <= i; j += 5) {
      double ref = 0; float f32 = 0;
      for (int k = 0; k < K; ++k) { ref += (double)P[(size_t)k * N + i] * P[(size_t)k * N + j]; f32 += P[(size_t)k * N + i] * P[(size_t)k * N + j]; }
      double got = 0;
      for (int q = 0; q < ksplit; ++q) got += C[q * slab + (size_t)i * N + j];
      emax = fmax(emax, fabs(got - ref)); gmax = fmax(gmax, fabs(ref)); e32 = fmax(e32, fabs((double)f32 - ref));
    }
    printf("max |G - ref| = %.3e (max |ref| %.3e; sequential fp32 sum: %.3e)\n", emax, gmax, e32);
  }
  return 0;
}
