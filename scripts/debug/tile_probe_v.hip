// The K loop of chol_update_tile_v (esl_chol.hpp: 128 x 128 tile, four waves of 32 x 128, two workgroups per CU) rebuilt piece by piece
// on synthetic data: MODE 0 = MFMAs on register operands only; 1 = + the shared operand's LDS fragment reads; 2 = + one barrier per chunk;
// 3 = + the register -> LDS staging stores; 4 = + the global loads (the whole loop).
//   hipcc -O3 --offload-arch=gfx950 scripts/debug/tile_probe_v.hip -o scripts/debug/bin/tile_probe_v
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));
constexpr int kKC = 16, VT = 128, VLd = VT + 16, NT = 256, NJ = 8, QB = VT * kKC / 2 / NT;
template <int MODE, int WPS, int HALVES = 0>
__global__ __launch_bounds__(NT * (HALVES ? 2 : 1), HALVES ? 2 : WPS) void k_probe(const double* __restrict__ P, long ldp, int K, double* out) {
  extern __shared__ __attribute__((aligned(16))) double sm_all[];
  double* sm = sm_all;
  const int half = HALVES ? (threadIdx.x >> 8) : 0;
  const int t = threadIdx.x & 255, wave = t >> 6, lane = t & 63, r = lane & 15, kq = lane >> 4;
  int* bar = reinterpret_cast<int*>(sm + (HALVES ? 2 : 1) * 2 * kKC * VLd) + 32 * half;   // one counter per half, in LDS lines of their own
  if (HALVES && HALVES != 3) sm += half * 2 * kKC * VLd;
  for (int i = t; i < 2 * kKC * VLd; i += NT) sm[i] = 1e-3 * (i % 97);
  if (t == 0) *bar = 0;
  __syncthreads();
  int phase = 0;
  if (HALVES == 2 && blockIdx.x == 0 && lane == 0) { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); out[4096 * NT + (threadIdx.x >> 6)] = (double)((hw >> 4) & 3); }
  auto sync_half = [&]() {
    if (!HALVES || HALVES == 3) { __syncthreads(); return; }
    phase += 4;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < phase) __builtin_amdgcn_s_sleep(0);
    asm volatile("" ::: "memory");
  };
  double4_t acc[NJ][2];
  for (int x = 0; x < NJ; ++x) { acc[x][0] = double4_t{0, 0, 0, 0}; acc[x][1] = double4_t{0, 0, 0, 0}; }
  const long i0 = (long)(blockIdx.x * 2 + half) * VT % 4096, j0 = (long)(blockIdx.x * 7) * VT % 4096;
  const double* pB[QB];
  for (int q = 0; q < QB; ++q) { const int e = t + NT * q; pB[q] = P + j0 + 2 * (e % (VT / 2)) + (long)(e / (VT / 2)) * ldp; }
  const double* pA = P + i0 + wave * 32 + 2 * r + (long)kq * ldp;
  double2_t rb[QB], a0[4], a1[4];
  for (int q = 0; q < QB; ++q) rb[q] = double2_t{3e-3, 1e-3};
  for (int s = 0; s < 4; ++s) { a0[s] = double2_t{1e-3 * lane, 2e-3}; a1[s] = double2_t{1e-3, 2e-3 * lane}; }
  double bqs[NJ];
  for (int m = 0; m < NJ; ++m) bqs[m] = 1e-3 * (m + lane);
  auto chunk = [&](double2_t (&a_cur)[4], double2_t (&a_nxt)[4], int kn, int buf) {
    asm volatile("" ::: "memory");
    if (MODE >= 4) {
      for (int q = 0; q < (HALVES == 3 ? QB / 2 : QB); ++q) rb[q] = *reinterpret_cast<const double2_t*>(pB[HALVES == 3 ? 2 * q + half : q] + (long)(kn % 512) * ldp);
      for (int s = 0; s < 4; ++s) a_nxt[s] = *reinterpret_cast<const double2_t*>(pA + (long)(kn % 512 + 4 * s) * ldp);
    }
    const double* Bb = sm + buf * kKC * VLd + r;
    double bq[2][NJ];
    if (MODE >= 1) {
#pragma unroll
      for (int m = 0; m < NJ; ++m) bq[0][m] = Bb[kq * VLd + m * 16];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (MODE >= 1 && s + 1 < 4) {
#pragma unroll
        for (int m = 0; m < NJ; ++m) bq[(s + 1) & 1][m] = Bb[(4 * (s + 1) + kq) * VLd + m * 16];
      }
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj) {
        const double b = MODE >= 1 ? bq[s & 1][nj] : bqs[nj];
        acc[nj][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a_cur[s].x, acc[nj][0], 0, 0, 0);
        acc[nj][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a_cur[s].y, acc[nj][1], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE >= 3) {
#pragma unroll
      for (int q = 0; q < (HALVES == 3 ? QB / 2 : QB); ++q) { const int e = t + NT * (HALVES == 3 ? 2 * q + half : q); *reinterpret_cast<double2_t*>(sm + ((buf ^ 1) * kKC + e / (VT / 2)) * VLd + 2 * (e % (VT / 2))) = rb[q]; }
    }
    if (MODE >= 2) sync_half();
  };
  for (int kc = 0; kc < K; kc += 2 * kKC) { chunk(a0, a1, kc + kKC, 0); chunk(a1, a0, kc + 2 * kKC, 1); }
  double s = 0;
  for (int x = 0; x < NJ; ++x) for (int y = 0; y < 2; ++y) s += acc[x][y][0] + acc[x][y][1] + acc[x][y][2] + acc[x][y][3];
  out[((size_t)blockIdx.x * NT + t) % (4096 * NT)] = s;
}
template <int MODE, int WPS, int HALVES = 0>
void run(const char* name, const double* P, long ldp, double* out, int K) {
  const size_t lds = HALVES ? (size_t)(4 * kKC * VLd) * sizeof(double) + 256 + 72 * 1024 : (size_t)(2 * kKC * VLd) * sizeof(double) * (WPS == 1 ? 3 : 1);   // (WPS = 1: enough LDS that only one workgroup fits a CU)
  hipFuncSetAttribute((const void*)k_probe<MODE, WPS, HALVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = (HALVES ? 256 : 512) * 4;
  hipLaunchKernelGGL((k_probe<MODE, WPS, HALVES>), dim3(grid), dim3(NT * (HALVES ? 2 : 1)), lds, 0, P, ldp, K, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_probe<MODE, WPS, HALVES>), dim3(grid), dim3(NT * (HALVES ? 2 : 1)), lds, 0, P, ldp, K, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double fl = (double)grid * (HALVES ? 2 : 1) * 2.0 * VT * VT * K;
  printf("%-72s %8.3f ms  %6.2f TFLOP/s  (%.1f %% of 78.6)\n", name, ms, fl / ms * 1e-9, fl / ms * 1e-9 / 78.6 * 100);
}
int main() {
  const long ldp = 18016;
  double* P; hipMalloc(&P, (size_t)ldp * 1024 * sizeof(double)); hipMemset(P, 0, (size_t)ldp * 1024 * sizeof(double));
  double* out; hipMalloc(&out, ((size_t)4096 * NT + 64) * sizeof(double));
  const int K = 4096;
  run<0, 2>("two workgroups per CU: MFMAs on register operands", P, ldp, out, K);
  run<1, 2>("+ LDS fragment reads of the shared operand", P, ldp, out, K);
  run<2, 2>("+ one barrier per chunk", P, ldp, out, K);
  run<3, 2>("+ staging stores registers -> LDS", P, ldp, out, K);
  run<4, 2>("+ global loads (the whole K loop)", P, ldp, out, K);
  run<4, 2, 1>("one 512-thread workgroup per CU, two halves, LDS-counter barrier: whole loop", P, ldp, out, K);
  run<2, 2, 1>("   the same without the staging stores and global loads", P, ldp, out, K);
  run<4, 2, 2>("   (whole loop again; SIMD ids of the eight waves below)", P, ldp, out, K);
  run<4, 2, 3>("one 512-thread workgroup: ONE shared image of the staged operand, hardware barrier", P, ldp, out, K);
  run<2, 2, 3>("   the same without the staging stores and global loads", P, ldp, out, K);
  double simd[8]; hipMemcpy(simd, out + (size_t)4096 * NT, sizeof(simd), hipMemcpyDeviceToHost);
  printf("SIMD of waves 0..7 of workgroup 0:"); for (int i = 0; i < 8; ++i) printf(" %d", (int)simd[i]); printf("\n");
  return 0;
}
