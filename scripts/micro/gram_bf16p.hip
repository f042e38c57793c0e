// Development micro-benchmark, pre-split variant of gram_bf16.hip: a first pass writes the
// three bf16 parts of P in MFMA-operand order [part][k / 8][column][8] (16 bytes = the 8
// consecutive-k values one lane feeds to v_mfma_f32_32x32x16_bf16); the product kernel then
// only moves 16-byte vectors global -> LDS -> registers and issues MFMAs.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BK = 16, GT = 256, kNumXcd = 8;

// P (K x N fp32, ld) -> S[part][k8][n_pad] of u32x4; rows past K are zero
__global__ void __launch_bounds__(256) split_kernel(const float *P, size_t ld, int K, int N, u32x4 *S, size_t n_pad,
                                                    size_t part_stride) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int k8 = blockIdx.y;
  if (i >= (int)n_pad) return;
  float v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int k = k8 * 8 + q;
    v[q] = (k < K && i < N) ? P[(size_t)k * ld + i] : 0.f;
  }
  unsigned hb[8], mb[8], lb[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const unsigned u = __float_as_uint(v[q]);
    hb[q] = u & 0xffff0000u;
    const float r1 = v[q] - __uint_as_float(hb[q]);
    mb[q] = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mb[q]);
    lb[q] = __float_as_uint(r2) & 0xffff0000u;
  }
  u32x4 h, m, l;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    h[p] = (hb[2 * p] >> 16) | hb[2 * p + 1];
    m[p] = (mb[2 * p] >> 16) | mb[2 * p + 1];
    l[p] = (lb[2 * p] >> 16) | lb[2 * p + 1];
  }
  const size_t o = (size_t)k8 * n_pad + i;
  S[o] = h; S[part_stride + o] = m; S[2 * part_stride + o] = l;
}

struct GArgs {
  const u32x4 *S; size_t n_pad, part_stride; int K, N;
  float *C; size_t ldc;
  int ksplit, kchunk; size_t cstride;
};

__global__ void __launch_bounds__(GT) gram_kernel(GArgs g) {
  __shared__ __attribute__((aligned(16))) u32x4 sh[2][2][3][2][BM];
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
  const int kbeg = ks * g.kchunk, kend = min(g.K, kbeg + g.kchunk);   // multiples of 16 except the very end
  float *Cout = g.C + (size_t)ks * g.cstride;
  const bool diag = ti == tj;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = t & 127, lk8 = t >> 7;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int r32 = lane & 31, kh = lane >> 5;

  floatx16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  u32x4 ra[3], rb[3];
  const int k8_end = (kend + 7) / 8;
  auto gload = [&](int k0) {
    const int k8 = k0 / 8 + lk8;
    const bool ok = k8 < k8_end;
    const size_t oa = (size_t)k8 * g.n_pad + i0 + li, ob = (size_t)k8 * g.n_pad + j0 + li;
    const u32x4 z = {0, 0, 0, 0};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      ra[p] = ok ? g.S[p * g.part_stride + oa] : z;
      if (!diag) rb[p] = ok ? g.S[p * g.part_stride + ob] : z;
    }
  };
  auto lstore = [&](int st) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      sh[st][0][p][lk8][li] = ra[p];
      if (!diag) sh[st][1][p][lk8][li] = rb[p];
    }
  };
  const int bop = diag ? 0 : 1;
  auto compute = [&](int st) {
    bf16x8 A[2][3], B[2][3];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const u32x4 xa = sh[st][0][p][kh][wm + a * 32 + r32];
        const u32x4 xb = sh[st][bop][p][kh][wn + a * 32 + r32];
        A[a][p] = *reinterpret_cast<const bf16x8 *>(&xa);
        B[a][p] = *reinterpret_cast<const bf16x8 *>(&xb);
      }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        floatx16 c = acc[a][b];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[a][1], B[b][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[a][0], B[b][2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[a][2], B[b][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[a][0], B[b][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[a][1], B[b][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[a][0], B[b][0], c, 0, 0, 0);
        acc[a][b] = c;
      }
  };
  const int nk = ((kend - kbeg + BK - 1) / BK + 1) & ~1;
  gload(kbeg);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    gload(kbeg + (kt + 1) * BK);
    compute(0);
    lstore(1);
    __syncthreads();
    gload(kbeg + (kt + 2) * BK);
    compute(1);
    lstore(0);
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + wm + a * 32 + (r / 4) * 8 + kh * 4 + (r % 4);
        const int col = j0 + wn + b * 32 + r32;
        if (row < g.N && col < g.N) Cout[(size_t)row * g.ldc + col] = acc[a][b][r];
      }
}

int main(int argc, char **argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 512;
  const int ksplit = argc > 3 ? atoi(argv[3]) : 1;
  const bool check = (size_t)K * N <= (1u << 23);
  std::vector<float> P((size_t)K * N);
  unsigned s = 12345;
  for (auto &v : P) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 20001 - 10000) * 1e-4f * (1.0f / sqrtf((float)K)); }
  float *dP, *dC; u32x4 *dS;
  const size_t slab = (size_t)N * N;
  const size_t n_pad = (N + BM - 1) / BM * BM;
  const int kchunk = ((K + ksplit - 1) / ksplit + 31) / 32 * 32;
  const size_t k8n = ((size_t)kchunk * ksplit + 31) / 8 + 4;
  const size_t part_stride = k8n * n_pad;
  hipMalloc(&dP, P.size() * 4); hipMalloc(&dC, slab * 4 * ksplit); hipMalloc(&dS, part_stride * 3 * 16);
  hipMemcpy(dP, P.data(), P.size() * 4, hipMemcpyHostToDevice);
  hipMemset(dC, 0, slab * 4 * ksplit);
  hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
  GArgs g{dS, n_pad, part_stride, K, N, dC, (size_t)N, ksplit, kchunk, slab};
  const int tm = (N + BM - 1) / BM, nunits = tm * (tm + 1) / 2 * ksplit;
  const int grid = (nunits + kNumXcd - 1) / kNumXcd * kNumXcd;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(split_kernel, dim3((n_pad + 255) / 256, k8n), dim3(256), 0, 0, dP, (size_t)N, K, N, dS, n_pad, part_stride);
    hipEventRecord(e1);
    hipLaunchKernelGGL(gram_kernel, dim3(grid), dim3(GT), 0, 0, g);
    hipEventRecord(e2); hipEventSynchronize(e2);
    float ms1, ms2; hipEventElapsedTime(&ms1, e0, e1); hipEventElapsedTime(&ms2, e1, e2);
    printf("K=%d N=%d ksplit=%d: split %.3f ms + product %.3f ms  -> %.1f TFLOP/s fp32-equivalent\n", K, N, ksplit, ms1, ms2,
           (double)K * N * N / ((ms1 + ms2) * 1e-3) / 1e12);
  }
  if (check) {
    std::vector<float> C(slab * ksplit);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double emax = 0, gmax = 0;
    for (int i = 0; i < N; i += 7) for (int j = 0; j <= i; j += 5) {
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)P[(size_t)k * N + i] * P[(size_t)k * N + j];
      double got = 0;
      for (int q = 0; q < ksplit; ++q) got += C[q * slab + (size_t)i * N + j];
      emax = fmax(emax, fabs(got - ref)); gmax = fmax(gmax, fabs(ref));
    }
    printf("max |G - ref| = %.3e (max |ref| %.3e)\n", emax, gmax);
  }
  return 0;
}
