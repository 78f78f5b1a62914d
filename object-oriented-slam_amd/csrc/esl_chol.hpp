// esl_chol.hpp — dense fp64 Cholesky (lower, column-major, in place) + solve for the reduced camera
// system of SLAM mode, written for gfx950: FP64 MFMA (v_mfma_f64_16x16x4_f64) for the panel solve
// and the trailing rank-NB update, LDS for the diagonal block.
//
// Replaces g2o's LinearSolverDense (Thirdparty/g2o/g2o/solvers/linear_solver_dense.h:65-113:
// Eigen::LDLT of the dense matrix, "ok" iff positive) on the Schur-reduced system that
// BlockSolver::solve forms (core/block_solver.hpp:367-486).
//
// Layout: M is (n+1) x n, column-major, leading dimension lda >= n+1.  Rows 0..n-1 hold the lower
// triangle of S; row n holds b^T.  Factoring the augmented matrix leaves y = L^-1 b in row n (the
// forward substitution rides along in the panel solve), so only the backward substitution
// L^T x = y is separate.
//
// Blocked right-looking, NB = 128:
//   k_potrf   one workgroup: diagonal block -> L11 in LDS, then Linv = L11^-1 (in-place trtri)
//   k_panel   P <- P * Linv^T           (rows below the diagonal block; MFMA, X*Y^T form)
//   k_update  C <- C - P_i * P_j^T      (lower-triangle tiles of the trailing matrix; MFMA)
#pragma once
#include <hip/hip_runtime.h>

namespace esl {

constexpr int kNB = 128;       // panel width
constexpr int kLdsPad = 129;   // column stride of the LDS copy of the diagonal block (bank-conflict pad)

typedef double double4_t __attribute__((ext_vector_type(4)));

// ---- diagonal block: Cholesky in LDS, then triangular inverse -----------------------------------------
// info[0] |= 1 when a pivot is not positive (g2o: LDLT not positive => solve fails => step rejected)
static __global__ __launch_bounds__(256) void k_chol_potrf(double* __restrict__ M, long lda, int k0, int nb,
                                                    double* __restrict__ Linv /* kNB x kNB col-major */,
                                                    int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* L = sm;                    // nb x nb, column stride kLdsPad
  double* v = sm + kNB * kLdsPad;    // nb
  const int t = threadIdx.x;
  for (int idx = t; idx < nb * nb; idx += 256) {
    const int i = idx % nb, j = idx / nb;
    L[i + j * kLdsPad] = (i >= j) ? M[(long)(k0 + i) + (long)(k0 + j) * lda] : 0.0;
  }
  __syncthreads();
  for (int j = 0; j < nb; ++j) {
    const double d = L[j + j * kLdsPad];
    if (!(d > 0)) {
      if (t == 0) atomicOr(info, 1);
    }
    const double s = sqrt(d);
    __syncthreads();
    for (int i = j + t; i < nb; i += 256) L[i + j * kLdsPad] = (i == j) ? s : L[i + j * kLdsPad] / s;
    __syncthreads();
    // trailing update: columns c > j, rows i >= c
    const int m = nb - j - 1;
    for (int idx = t; idx < m * m; idx += 256) {
      const int c = j + 1 + idx / m, i = j + 1 + idx % m;
      if (i >= c) L[i + c * kLdsPad] -= L[i + j * kLdsPad] * L[c + j * kLdsPad];
    }
    __syncthreads();
  }
  // write L11 back (the factor itself is part of the result)
  for (int idx = t; idx < nb * nb; idx += 256) {
    const int i = idx % nb, j = idx / nb;
    if (i >= j) M[(long)(k0 + i) + (long)(k0 + j) * lda] = L[i + j * kLdsPad];
  }
  // in-place inverse of the lower-triangular L (LAPACK dtrti2, lower): columns last to first
  for (int j = nb - 1; j >= 0; --j) {
    const double ajj = 1.0 / L[j + j * kLdsPad];
    for (int i = j + 1 + t; i < nb; i += 256) v[i] = L[i + j * kLdsPad];
    __syncthreads();
    for (int i = j + 1 + t; i < nb; i += 256) {
      double s = 0;
      for (int k = j + 1; k <= i; ++k) s += L[i + k * kLdsPad] * v[k];
      L[i + j * kLdsPad] = -s * ajj;
    }
    if (t == 0) L[j + j * kLdsPad] = ajj;
    __syncthreads();
  }
  for (int idx = t; idx < kNB * kNB; idx += 256) {
    const int i = idx % kNB, j = idx / kNB;
    Linv[i + j * kNB] = (i < nb && j < nb && i >= j) ? L[i + j * kLdsPad] : 0.0;
  }
}

// ---- MFMA micro-kernel: acc(64x64 per wave) += X[i0.., 0..K) * Y[j0.., 0..K)^T ------------------------------
// X, Y column-major (element (r,k) at X[r + k*ld]); rows beyond the limits read as zero.
// acc[mi][nj] is the 16x16 tile (mi,nj) in the f64 MFMA C/D layout: col = lane&15, row = (lane>>4) + 4*reg.
__device__ __forceinline__ void mfma_xyT_64x64(const double* __restrict__ X, long ldx, long xrow0, long xrows,
                                               const double* __restrict__ Y, long ldy, long yrow0, long yrows,
                                               int K, double4_t acc[4][4]) {
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, kq = lane >> 4;
  long xr[4], yr[4];
  bool xv[4], yv[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    xr[m] = xrow0 + m * 16 + r; xv[m] = xr[m] < xrows;
    yr[m] = yrow0 + m * 16 + r; yv[m] = yr[m] < yrows;
  }
  for (int kk = 0; kk < K; kk += 4) {
    double a[4], b[4];
    const long kcol = kk + kq;
    const bool kv = kcol < K;  // K need not be a multiple of 4 (last panel)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      a[m] = (xv[m] && kv) ? X[xr[m] + kcol * ldx] : 0.0;
      b[m] = (yv[m] && kv) ? Y[yr[m] + kcol * ldy] : 0.0;
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[nj], acc[mi][nj], 0, 0, 0);
  }
}

// ---- panel solve: P <- P * Linv^T for rows [r0, rows) of columns [k0, k0+nb) -------------------------------
// One wave owns 64 rows x all kNB columns (two 64-column halves): it reads its rows completely before
// writing them, so the in-place update is race-free.
static __global__ __launch_bounds__(256) void k_chol_panel(double* __restrict__ M, long lda, long rows, int k0, int nb,
                                                    const double* __restrict__ Linv) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long r0 = (long)k0 + nb + ((long)blockIdx.x * 4 + wave) * 64;
  if (r0 >= rows) return;
  double4_t acc0[4][4], acc1[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc0[i][j] = double4_t{0, 0, 0, 0}; acc1[i][j] = double4_t{0, 0, 0, 0}; }
  const double* P = M + (long)k0 * lda;
  mfma_xyT_64x64(P, lda, r0, rows, Linv, kNB, 0, kNB, nb, acc0);
  mfma_xyT_64x64(P, lda, r0, rows, Linv, kNB, 64, kNB, nb, acc1);
  // all reads of this wave's rows are done (acc depends on them); write back
  const int c = lane & 15, rq = lane >> 4;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int nj = 0; nj < 4; ++nj)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const long row = r0 + mi * 16 + rq + 4 * g;
        const int col0 = nj * 16 + c, col1 = 64 + nj * 16 + c;
        if (row < rows) {
          if (col0 < nb) M[row + (long)(k0 + col0) * lda] = acc0[mi][nj][g];
          if (col1 < nb) M[row + (long)(k0 + col1) * lda] = acc1[mi][nj][g];
        }
      }
}

// ---- trailing update: C <- C - P_i P_j^T over lower-triangle 128x128 tiles --------------------------------
// grid.x enumerates tile pairs (ti >= tj) of the trailing matrix; 4 waves = 2x2 sub-tiles of 64x64.
static __global__ __launch_bounds__(256) void k_chol_update(double* __restrict__ M, long lda, long rows, long ncols, int k0,
                                                     int nb, int ntiles) {
  // decode (ti, tj) with ti >= tj from the linear index
  const long b = blockIdx.x;
  long ti = (long)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
  while (ti * (ti + 1) / 2 > b) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= b) ++ti;
  const long tj = b - ti * (ti + 1) / 2;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long base = (long)k0 + nb;
  const long i0 = base + ti * 128 + (wave >> 1) * 64;
  const long j0 = base + tj * 128 + (wave & 1) * 64;
  if (i0 >= rows || j0 >= ncols) return;
  if (j0 > i0 + 63) return;  // strictly upper 64x64 sub-tile of a diagonal tile
  double4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = double4_t{0, 0, 0, 0};
  const double* P = M + (long)k0 * lda;
  mfma_xyT_64x64(P, lda, i0, rows, P, lda, j0, ncols, nb, acc);
  const int c = lane & 15, rq = lane >> 4;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int nj = 0; nj < 4; ++nj)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const long row = i0 + mi * 16 + rq + 4 * g;
        const long col = j0 + nj * 16 + c;
        if (row < rows && col < ncols && row >= col) M[row + col * lda] -= acc[mi][nj][g];
      }
  (void)ntiles;
}

// ---- backward substitution L^T x = y, panel by panel from the last ------------------------------------------
// z[c] = y[k0+c] - sum_{i >= k0+nb}^{n-1} L[i][k0+c] x[i]   (one workgroup per column: contiguous dot product)
static __global__ __launch_bounds__(256) void k_chol_backdot(const double* __restrict__ M, long lda, int n, int k0, int nb,
                                                      const double* __restrict__ x, double* __restrict__ z) {
  __shared__ double red[256];
  const int c = blockIdx.x;
  if (c >= nb) return;
  const double* col = M + (long)(k0 + c) * lda;
  double s = 0;
  for (long i = (long)k0 + nb + threadIdx.x; i < n; i += 256) s += col[i] * x[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) z[c] = col[n] - red[0];  // row n holds y^T
}
// x[k0 + r] = sum_c Linv[c][r] z[c]  (Linv^T z)
static __global__ __launch_bounds__(128) void k_chol_backsolve(const double* __restrict__ Linv, int k0, int nb,
                                                        const double* __restrict__ z, double* __restrict__ x) {
  const int r = threadIdx.x;
  if (r >= nb) return;
  double s = 0;
  for (int c = r; c < nb; ++c) s += Linv[c + r * kNB] * z[c];
  x[k0 + r] = s;
}

// Host driver.  M: (n+1) x n col-major (lda), Linv_ws: ceil(n/NB) * NB*NB doubles, z_ws: NB doubles,
// x: n doubles (output), info: device int (bit 0 set on a non-positive pivot).
inline hipError_t chol_factor_solve(double* M, long lda, int n, double* Linv_ws, double* z_ws, double* x, int* info,
                                    hipStream_t st) {
  const long rows = (long)n + 1;
  const size_t lds = (size_t)(kNB * kLdsPad + kNB) * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_chol_potrf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  int np = (n + kNB - 1) / kNB;
  for (int p = 0; p < np; ++p) {
    const int k0 = p * kNB, nb = (n - k0 < kNB) ? (n - k0) : kNB;
    double* Linv = Linv_ws + (size_t)p * kNB * kNB;
    hipLaunchKernelGGL(k_chol_potrf, dim3(1), dim3(256), lds, st, M, lda, k0, nb, Linv, info);
    const long below = rows - (k0 + nb);
    if (below > 0) {
      const int pb = (int)((below + 255) / 256);
      hipLaunchKernelGGL(k_chol_panel, dim3(pb), dim3(256), 0, st, M, lda, rows, k0, nb, Linv);
      const long tcols = (long)n - (k0 + nb);
      if (tcols > 0) {
        const long nt = (below + 127) / 128;  // row tiles (includes the b row)
        const long nblk = nt * (nt + 1) / 2;
        hipLaunchKernelGGL(k_chol_update, dim3((unsigned)nblk), dim3(256), 0, st, M, lda, rows, (long)n, k0, nb, (int)nt);
      }
    }
  }
  for (int p = np - 1; p >= 0; --p) {
    const int k0 = p * kNB, nb = (n - k0 < kNB) ? (n - k0) : kNB;
    hipLaunchKernelGGL(k_chol_backdot, dim3(nb), dim3(256), 0, st, M, lda, n, k0, nb, x, z_ws);
    hipLaunchKernelGGL(k_chol_backsolve, dim3(1), dim3(128), 0, st, Linv_ws + (size_t)p * kNB * kNB, k0, nb, z_ws, x);
  }
  return hipGetLastError();
}

}  // namespace esl
