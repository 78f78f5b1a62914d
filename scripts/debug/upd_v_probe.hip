// The rank-K update C -= X^T X of esl_chol.hpp, 256 x 128 tile (k_chol_update_lds) against the vendor-shaped 128 x 128 tile
// (k_chol_update_v): same bits?  how long?   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I object-oriented-slam_amd/csrc scripts/debug/upd_v_probe.hip -o /tmp/upd_v_probe
#include "esl_chol.hpp"
#include <cstdio>
#include <cstring>
#include <vector>
using namespace esl;
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
  for (int which = 0; which < 2; ++which) {
    double* M = which ? dM2 : dM1;
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipMemcpy(M, hM.data(), hM.size() * 8, hipMemcpyHostToDevice));
      CK(hipEventRecord(e0, st));
      if (which) run_new(M); else run_old(M);
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipGetLastError());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    const double fl = (double)K * rows * n;   // lower triangle, 2 flops per term
    printf("%s  n=%ld K=%ld stair=%d sb=%d  %8.3f ms  %6.2f TF (full-triangle flops) = %.3f of 78.6\n", which ? "v(128x128 x2/CU)" : "lds(256x128)    ", n, K, stair, which ? sb : 0, best, fl / best / 1e9,
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
