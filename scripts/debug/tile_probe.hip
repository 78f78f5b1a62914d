// Where do the ~20 % between the update tile's K loop and the FP64-MFMA peak go?  The loop of chol_update_tile<256,128,4,2> (esl_chol.hpp)
// rebuilt piece by piece on synthetic data: MODE 0 = LDS fragment reads + MFMAs only; 1 = + one barrier per chunk; 2 = + the
// register -> LDS staging stores; 3 = + the global loads that feed them (the whole loop).  256 workgroups x 512 threads, 104 KB LDS.
//   hipcc -O3 --offload-arch=gfx950 scripts/debug/tile_probe.hip -o /tmp/tile_probe && /tmp/tile_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));
constexpr int kKC = 16, BM = 256, BN = 128, WM = 4, WN = 2, NT = 512;
constexpr int kLdA = BM + 16, kLdB = BN + 16, SM = BM / WM, SN = BN / WN, MI = SM / 16, NJ = SN / 16;
constexpr int QA = BM * kKC / 2 / NT, QB = BN * kKC / 2 / NT;
template <int MODE, int KSTEP_READS /* 0: as the kernel (per k-step); 1: all of a chunk's fragments up front */>
__global__ __launch_bounds__(NT, 2) void k_probe(const double* __restrict__ P, long ldp, int K, double* out) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* As = sm;
  double* Bs = sm + 2 * kKC * kLdA;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, wi = wave / WN, wj = wave % WN, r = lane & 15, kq = lane >> 4;
  for (int i = t; i < 2 * kKC * (kLdA + kLdB); i += NT) sm[i] = 1e-3 * (i % 97);
  __syncthreads();
  double4_t acc[NJ][MI];
  for (int x = 0; x < NJ; ++x) for (int y = 0; y < MI; ++y) acc[x][y] = double4_t{0, 0, 0, 0};
  double2_t ra[QA], rb[QB];
  const double* pA[QA]; const double* pB[QB];
  const long i0 = (long)blockIdx.x * BM % 4096, j0 = (long)blockIdx.x * BN % 2048;
  for (int q = 0; q < QA; ++q) { const int e = t + NT * q; pA[q] = P + i0 + 2 * (e % (BM / 2)) + (long)(e / (BM / 2)) * ldp; ra[q] = double2_t{1e-3, 2e-3}; }
  for (int q = 0; q < QB; ++q) { const int e = t + NT * q; pB[q] = P + j0 + 2 * (e % (BN / 2)) + (long)(e / (BN / 2)) * ldp; rb[q] = double2_t{3e-3, 1e-3}; }
  int buf = 0;
  for (int kc = 0; kc < K; kc += kKC) {
    asm volatile("" ::: "memory");   // (the LDS image is re-read every chunk, as in the kernel)
    if (MODE >= 3) {
      for (int q = 0; q < QA; ++q) ra[q] = *reinterpret_cast<const double2_t*>(pA[q] + (long)(kc % 512) * ldp);
      for (int q = 0; q < QB; ++q) rb[q] = *reinterpret_cast<const double2_t*>(pB[q] + (long)(kc % 512) * ldp);
    }
    const double* Ab = As + buf * kKC * kLdA + wi * SM + r;
    const double* Bb = Bs + buf * kKC * kLdB + wj * SN + r;
    if (KSTEP_READS == 0) {
#pragma unroll
      for (int kk = 0; kk < kKC; kk += 4) {
        double a[MI], bq[NJ];
#pragma unroll
        for (int m = 0; m < MI; ++m) a[m] = Ab[(kk + kq) * kLdA + m * 16];
#pragma unroll
        for (int m = 0; m < NJ; ++m) bq[m] = Bb[(kk + kq) * kLdB + m * 16];
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) acc[nj][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(bq[nj], a[mi], acc[nj][mi], 0, 0, 0);
      }
    } else {
      double a[4][MI], bq[4][NJ];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int m = 0; m < MI; ++m) a[s][m] = Ab[(4 * s + kq) * kLdA + m * 16];
#pragma unroll
        for (int m = 0; m < NJ; ++m) bq[s][m] = Bb[(4 * s + kq) * kLdB + m * 16];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) acc[nj][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(bq[s][nj], a[s][mi], acc[nj][mi], 0, 0, 0);
    }
    if (MODE >= 2) {
#pragma unroll
      for (int q = 0; q < QA; ++q) { const int e = t + NT * q, pr = e % (BM / 2), k = e / (BM / 2); *reinterpret_cast<double2_t*>(As + ((buf ^ 1) * kKC + k) * kLdA + 2 * pr) = ra[q]; }
#pragma unroll
      for (int q = 0; q < QB; ++q) { const int e = t + NT * q, pr = e % (BN / 2), k = e / (BN / 2); *reinterpret_cast<double2_t*>(Bs + ((buf ^ 1) * kKC + k) * kLdB + 2 * pr) = rb[q]; }
    }
    if (MODE >= 1) __syncthreads();
    buf ^= 1;
  }
  double s = 0;
  for (int x = 0; x < NJ; ++x) for (int y = 0; y < MI; ++y) s += acc[x][y][0] + acc[x][y][1] + acc[x][y][2] + acc[x][y][3];
  out[(size_t)blockIdx.x * NT + t] = s;
}
template <int MODE, int KR>
void run(const char* name, const double* P, long ldp, double* out, int K) {
  const size_t lds = (size_t)(2 * kKC * (kLdA + kLdB)) * sizeof(double);
  hipFuncSetAttribute((const void*)k_probe<MODE, KR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 4;
  hipLaunchKernelGGL((k_probe<MODE, KR>), dim3(grid), dim3(NT), lds, 0, P, ldp, K, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_probe<MODE, KR>), dim3(grid), dim3(NT), lds, 0, P, ldp, K, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double fl = (double)grid * 2.0 * BM * BN * K;
  printf("%-64s %8.3f ms  %6.2f TFLOP/s  (%.1f %% of 78.6)\n", name, ms, fl / ms * 1e-9, fl / ms * 1e-9 / 78.6 * 100);
}
int main() {
  const long ldp = 18016, rows = 4096 + 512;
  double* P; hipMalloc(&P, (size_t)ldp * 1024 * sizeof(double)); hipMemset(P, 0, (size_t)ldp * 1024 * sizeof(double));
  double* out; hipMalloc(&out, (size_t)1024 * NT * sizeof(double));
  (void)rows;
  const int K = 4096;
  run<0, 0>("LDS fragment reads + MFMA (reads per k-step, as the kernel)", P, ldp, out, K);
  run<0, 1>("LDS fragment reads + MFMA (a chunk's reads up front)", P, ldp, out, K);
  run<1, 0>("+ one barrier per chunk", P, ldp, out, K);
  run<1, 1>("+ one barrier per chunk (reads up front)", P, ldp, out, K);
  run<2, 0>("+ staging stores registers -> LDS", P, ldp, out, K);
  run<2, 1>("+ staging stores (reads up front)", P, ldp, out, K);
  run<3, 0>("+ global loads (the whole K loop)", P, ldp, out, K);
  run<3, 1>("+ global loads (reads up front)", P, ldp, out, K);
  return 0;
}
