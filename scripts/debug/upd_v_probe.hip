// The rank-K update C -= X^T X of esl_chol.hpp, 256 x 128 tile (k_chol_update_lds) against the vendor-shaped 128 x 128 tile
// (k_chol_update_v): same bits?  how long?   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I object-oriented-slam_amd/csrc scripts/debug/upd_v_probe.hip -o /tmp/upd_v_probe
#include "esl_chol.hpp"
#include <cstdio>
#include <cstring>
#include <vector>
using namespace esl;

// ---- experiment (all loads compiler-visible): the shared operand requested TWO chunks ahead through two register staging sets
typedef double double2_v4 __attribute__((ext_vector_type(2)));
template <int BQSETS>
static __global__ __launch_bounds__(256, 2) void k_update_v4(double* __restrict__ M, long lda, long rows, long ncols, int K, const double* __restrict__ P, long ldp,
                                                             const int* __restrict__ kfirst) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const long b = blockIdx.x;
  long ti = (long)((sqrt(1.0 + 8.0 * (double)b) - 1.0) * 0.5);
  while (ti * (ti + 1) / 2 > b) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= b) ++ti;
  const long tj = b - ti * (ti + 1) / 2;
  const long i0 = ti * kVT, j0 = tj * kVT;
  if (i0 >= rows || j0 >= ncols) return;
  if (kfirst) {
    int ka = K, kb = K;
    for (long g = i0 / 64; g < (i0 + kVT) / 64; ++g) ka = kfirst[g] < ka ? kfirst[g] : ka;
    for (long g = j0 / 64; g < (j0 + kVT) / 64; ++g) kb = kfirst[g] < kb ? kfirst[g] : kb;
    int ks = ka > kb ? ka : kb;
    ks = (ks < K ? ks : K) & ~(kKC - 1);
    if (((K - ks) & kKC) != 0 && ks >= kKC) ks -= kKC;
    P += (long)ks * ldp; K -= ks;
    if (K <= 0) return;
  }
  if (!(i0 + kVT <= ldp && K % (2 * kKC) == 0 && K >= 4 * kKC)) { chol_update_tile_v<false>(sm, M, lda, rows, ncols, P, ldp, K, i0, j0, false); return; }
  constexpr int NT = 256, NJ = kVT / 16, QB = 4;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, r = lane & 15, kq = lane >> 4;
  const long iw = i0 + wave * 32;
  const bool skipw = (j0 > iw + 31) || iw >= rows;
  const char* PBu = reinterpret_cast<const char*>(P + j0);
  const char* PAu = reinterpret_cast<const char*>(P + i0);
  const unsigned offB = (unsigned)((2 * (t % 64) + (long)(t / 64) * ldp) * 8);
  const unsigned offA = (unsigned)((wave * 32 + 2 * r + (long)kq * ldp) * 8);
  auto gload_b = [&](double2_v4 (&rb)[QB], int kc) {
#pragma unroll
    for (int q = 0; q < QB; ++q) rb[q] = *reinterpret_cast<const double2_v4*>(PBu + (long)(kc + 4 * q) * ldp * 8 + offB);
  };
  auto gload_a = [&](double2_v4 (&a)[4], int kc) {
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) a[s2] = *reinterpret_cast<const double2_v4*>(PAu + (long)(kc + 4 * s2) * ldp * 8 + offA);
  };
  auto sstore = [&](const double2_v4 (&rb)[QB], int buf) {
#pragma unroll
    for (int q = 0; q < QB; ++q) { const int e = t + NT * q; *reinterpret_cast<double2_v4*>(sm + (buf * kKC + e / 64) * kVLd + 2 * (e % 64)) = rb[q]; }
  };
  double4_t acc[NJ][2];
#pragma unroll
  for (int x = 0; x < NJ; ++x) { acc[x][0] = double4_t{0, 0, 0, 0}; acc[x][1] = double4_t{0, 0, 0, 0}; }
  const int klast = K - kKC;
  auto clampk = [&](int k) { return k <= klast ? k : klast; };
  // chunk at kc in LDS buffer `buf`: request the shared operand of kc + 32 into rb_ld and this wave's rows of kc + 16 into a_nxt, multiply,
  // then put the shared operand of kc + 16 (rb_st: requested a chunk ago) into the other buffer
  auto chunk = [&](double2_v4 (&a_cur)[4], double2_v4 (&a_nxt)[4], double2_v4 (&rb_ld)[QB], double2_v4 (&rb_st)[QB], int kc, int buf) {
    gload_b(rb_ld, clampk(kc + 2 * kKC));
    gload_a(a_nxt, clampk(kc + kKC));
    if (!skipw) {
      const double* Bb = sm + buf * kKC * kVLd + r;
      double bq[BQSETS][NJ];
#pragma unroll
      for (int m = 0; m < NJ; ++m) bq[0][m] = Bb[kq * kVLd + m * 16];
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        if (BQSETS == 2 && s2 + 1 < 4) {
#pragma unroll
          for (int m = 0; m < NJ; ++m) bq[(s2 + 1) & 1][m] = Bb[(4 * (s2 + 1) + kq) * kVLd + m * 16];
        }
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) {
          acc[nj][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(bq[BQSETS == 2 ? (s2 & 1) : 0][nj], a_cur[s2].x, acc[nj][0], 0, 0, 0);
          acc[nj][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(bq[BQSETS == 2 ? (s2 & 1) : 0][nj], a_cur[s2].y, acc[nj][1], 0, 0, 0);
        }
        if (BQSETS == 1 && s2 + 1 < 4) {
#pragma unroll
          for (int m = 0; m < NJ; ++m) bq[0][m] = Bb[(4 * (s2 + 1) + kq) * kVLd + m * 16];
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    sstore(rb_st, buf ^ 1);
    __syncthreads();
  };
  double2_v4 a0[4], a1[4], rbA[QB], rbB[QB];
  gload_b(rbA, 0); gload_a(a0, 0); gload_b(rbB, kKC);
  sstore(rbA, 0);
  __syncthreads();
  for (int kc = 0; kc < K; kc += 2 * kKC) {
    chunk(a0, a1, rbA, rbB, kc, 0);          // loads j(kc + 32) -> rbA, stores rbB = j(kc + 16)
    chunk(a1, a0, rbB, rbA, kc + kKC, 1);    // loads j(kc + 48) -> rbB, stores rbA = j(kc + 32)
  }
  if (skipw) return;
  const bool full = (iw + 32 <= rows) && (j0 + kVT <= ncols) && (j0 + kVT - 1 <= iw) && ((lda & 1) == 0);
#pragma unroll
  for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (full) {
        double2_v4* p = reinterpret_cast<double2_v4*>(&M[(iw + 2 * r) + (j0 + nj * 16 + kq + 4 * g) * lda]);
        const double2_v4 cv = *p;
        *p = double2_v4{cv.x - acc[nj][0][g], cv.y - acc[nj][1][g]};
      } else {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const long col = j0 + nj * 16 + kq + 4 * g, row = iw + 2 * r + mi;
          if (row < rows && col < ncols && row >= col) M[row + col * lda] -= acc[nj][mi][g];
        }
      }
    }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 18000, K = argc > 2 ? atol(argv[2]) : 3744;
  const int stair = argc > 3 ? atoi(argv[3]) : 0, sb = argc > 4 ? atoi(argv[4]) : 0;
  const long rows = n + 1, lda = (rows + 127) / 128 * 128, ldx = lda;
  std::vector<double> hX((size_t)ldx * K), hM((size_t)lda * n);
  unsigned long long s = 12345;
  auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)((s >> 33) & 0xFFFFF) / 1048576.0 - 0.5; };
  std::vector<int> kf((rows + 256 + 63) / 64 + 8, 0);
  if (stair) for (size_t g = 0; g < kf.size(); ++g) kf[g] = (int)std::min<long>(K, (long)(0.18 * (double)(g * 64) * K / rows) / 6 * 6);   // first non-zero row per 64 columns
  for (long k = 0; k < K; ++k) for (long c = 0; c < ldx; ++c) hX[(size_t)k * ldx + c] = (c < rows && k >= kf[c / 64]) ? rnd() : 0.0;
  for (auto& v : hM) v = rnd();
  double *dX, *dM1, *dM2; int* dkf;
  CK(hipMalloc(&dX, hX.size() * 8)); CK(hipMalloc(&dM1, hM.size() * 8)); CK(hipMalloc(&dM2, hM.size() * 8)); CK(hipMalloc(&dkf, kf.size() * 4));
  CK(hipMemcpy(dX, hX.data(), hX.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dkf, kf.data(), kf.size() * 4, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)k_chol_update_lds<256, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCholLdsBig));
  CK(hipFuncSetAttribute((const void*)k_chol_update_v, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCholLdsV));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run_old = [&](double* M) {
    const long ntI = (rows + 255) / 256, ntJ = (n + 127) / 128, nblk = ntI * (ntI + 1);
    hipLaunchKernelGGL((k_chol_update_lds<256, 128>), dim3((unsigned)nblk), dim3(512), kCholLdsBig, st, M, lda, rows, n, 0, (int)K, 0L, (int)ntJ, 0, dX, ldx, (double*)nullptr, 0,
                       stair ? dkf : nullptr);
  };
  auto run_new = [&](double* M) {
    const long ntI = (rows + 127) / 128, ntJ = (n + 127) / 128, nblk = chol_v_grid(ntI, sb);
    hipLaunchKernelGGL(k_chol_update_v, dim3((unsigned)nblk), dim3(256), kCholLdsV, st, M, lda, rows, n, 0, (int)K, 0L, (int)ntJ, 0, dX, ldx, stair ? dkf : nullptr, sb);
  };
  const int v4 = argc > 5 ? atoi(argv[5]) : 0;   // 41: two fragment sets, 42: one
  auto run_v4 = [&](double* M) {
    const long ntI = (rows + 127) / 128, nblk = ntI * (ntI + 1) / 2;
    if (v4 == 41) hipLaunchKernelGGL(k_update_v4<2>, dim3((unsigned)nblk), dim3(256), kCholLdsV, st, M, lda, rows, n, (int)K, dX, ldx, stair ? dkf : nullptr);
    else hipLaunchKernelGGL(k_update_v4<1>, dim3((unsigned)nblk), dim3(256), kCholLdsV, st, M, lda, rows, n, (int)K, dX, ldx, stair ? dkf : nullptr);
  };
  for (int which = 0; which < 2; ++which) {
    double* M = which ? dM2 : dM1;
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipMemcpy(M, hM.data(), hM.size() * 8, hipMemcpyHostToDevice));
      CK(hipEventRecord(e0, st));
      if (which) { if (v4) run_v4(M); else run_new(M); } else run_old(M);
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipGetLastError());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    const double fl = (double)K * rows * n;   // lower triangle, 2 flops per term
    printf("%s  n=%ld K=%ld stair=%d sb=%d  %8.3f ms  %6.2f TF (full-triangle flops) = %.3f of 78.6\n", which ? (v4 == 41 ? "v4 (j two ahead)" : v4 == 42 ? "v4, 1 frag set  " : "v(128x128 x2/CU)") : "lds(256x128)    ", n, K, stair, which ? sb : 0, best, fl / best / 1e9,
           fl / best / 1e9 / 78.6);
  }
  std::vector<double> r1(hM.size()), r2(hM.size());
  CK(hipMemcpy(r1.data(), dM1, r1.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(), dM2, r2.size() * 8, hipMemcpyDeviceToHost));
  long diff = 0, changed = 0; double maxd = 0;
  for (long c = 0; c < n; ++c) for (long r = c; r < rows; ++r) {
    const size_t i = (size_t)c * lda + r;
    if (memcmp(&r1[i], &r2[i], 8)) { if (diff < 12) printf("first differing: row %ld col %ld  lds %.17g  v %.17g  before %.17g\n", r, c, r1[i], r2[i], hM[i]); ++diff; maxd = std::max(maxd, std::abs(r1[i] - r2[i])); }
    if (r1[i] != hM[i]) ++changed;
  }
  long above = 0;
  for (long c = 0; c < n; ++c) for (long r = 0; r < c; ++r) { const size_t i = (size_t)c * lda + r; if (r2[i] != hM[i]) ++above; }
  printf("lower-triangle entries that differ between the two kernels: %ld (max |d| %.3e); changed by the update: %ld; written above the diagonal by v: %ld\n", diff, maxd, changed, above);
  return diff != 0;
}
