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

// ---- diagonal block: blocked Cholesky in LDS (32-wide sub-blocks), then blocked triangular inverse --------------
// info[0] |= 1 when a pivot is not positive (g2o: LDLT not positive => solve fails => step rejected).
// The serial chain is 4 x 32 small column steps; the rank-32 updates and the 32x32 block products of the
// inverse use all 256 threads.  (The first version ran 128 full-size column steps + a serial trtri: 793 us.)
constexpr int kSB = 32;
static __global__ __launch_bounds__(256) void k_chol_potrf(double* __restrict__ M, long lda, int k0, int nb,
                                                    double* __restrict__ Linv /* kNB x kNB col-major */,
                                                    int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* L = sm;                          // nb x nb, column stride kLdsPad
  double* T = sm + kNB * kLdsPad;          // 96 x 32 temporary + 32-vector for the in-place inverse
  const int t = threadIdx.x;
#define LL(i, j) L[(i) + (j) * kLdsPad]
  for (int idx = t; idx < kNB * kNB; idx += 256) {
    const int i = idx % kNB, j = idx / kNB;
    LL(i, j) = (i < nb && j < nb && i >= j) ? M[(long)(k0 + i) + (long)(k0 + j) * lda] : ((i == j) ? 1.0 : 0.0);
  }
  __syncthreads();
  for (int c0 = 0; c0 < nb; c0 += kSB) {
    const int w = (nb - c0 < kSB) ? (nb - c0) : kSB;
    // 1. unblocked factorisation of the w x w diagonal sub-block
    for (int j = c0; j < c0 + w; ++j) {
      const double d = LL(j, j);
      if (!(d > 0) && t == 0) atomicOr(info, 1);
      const double s = sqrt(d), is = 1.0 / s;
      __syncthreads();
      if (t < c0 + w - j) { const int i = j + t; LL(i, j) = (i == j) ? s : LL(i, j) * is; }
      __syncthreads();
      const int m = c0 + w - j - 1;   // remaining columns inside the sub-block
      for (int idx = t; idx < m * m; idx += 256) {
        const int c = j + 1 + idx / m, i = j + 1 + idx % m;
        if (i >= c) LL(i, c) -= LL(i, j) * LL(c, j);
      }
      __syncthreads();
    }
    const int r0 = c0 + w;            // rows below the sub-block
    // 2. sub-panel solve: X D^T = A, one thread per row
    for (int i = r0 + t; i < nb; i += 256) {
      for (int c = c0; c < c0 + w; ++c) {
        double v = LL(i, c);
        for (int k = c0; k < c; ++k) v -= LL(i, k) * LL(c, k);
        LL(i, c) = v / LL(c, c);
      }
    }
    __syncthreads();
    // 3. rank-w update of the trailing lower triangle
    const int m2 = nb - r0;
    for (int idx = t; idx < m2 * m2; idx += 256) {
      const int c = r0 + idx / m2, i = r0 + idx % m2;
      if (i >= c) {
        double v = 0;
        for (int k = c0; k < c0 + w; ++k) v += LL(i, k) * LL(c, k);
        LL(i, c) -= v;
      }
    }
    __syncthreads();
  }
  // write L11 back (the factor itself is part of the result)
  for (int idx = t; idx < nb * nb; idx += 256) {
    const int i = idx % nb, j = idx / nb;
    if (i >= j) M[(long)(k0 + i) + (long)(k0 + j) * lda] = LL(i, j);
  }
  // ---- Linv = L^-1 IN PLACE in LDS (blocked dtrtri, lower): block columns right to left ---------------------------
  // (the first blocked version kept Linv in global memory: store-to-load round trips made it 355 us per panel)
  __syncthreads();
  constexpr int nB = kNB / kSB;
  double* v = T + 96 * kSB;   // 32 doubles: column copy for the unblocked diagonal inverse
  for (int J = nB - 1; J >= 0; --J) {
    const int c0 = J * kSB, r0 = c0 + kSB, mrows = kNB - r0;
    // (a) T = Linv[r0.., r0..] * L[r0.., c0..c0+32)    (lower-triangular times panel)
    for (int idx = t; idx < mrows * kSB; idx += 256) {
      const int i = idx % mrows, c = idx / mrows;
      double acc = 0;
      for (int k = 0; k <= i; ++k) acc += LL(r0 + i, r0 + k) * LL(r0 + k, c0 + c);
      T[i + c * 96] = acc;
    }
    __syncthreads();
    // (b) diagonal block inverted in place (dtrti2, lower): columns right to left, rows in parallel
    for (int jj = kSB - 1; jj >= 0; --jj) {
      const int j = c0 + jj;
      const double ajj = 1.0 / LL(j, j);
      if (t > jj && t < kSB) v[t] = LL(c0 + t, j);
      __syncthreads();
      if (t > jj && t < kSB) {
        double acc = 0;
        for (int k = jj + 1; k <= t; ++k) acc += LL(c0 + t, c0 + k) * v[k];
        LL(c0 + t, j) = -acc * ajj;
      }
      if (t == 0) LL(j, j) = ajj;
      __syncthreads();
    }
    // (c) panel = -T * Dinv
    for (int idx = t; idx < mrows * kSB; idx += 256) {
      const int i = idx % mrows, c = idx / mrows;
      double acc = 0;
      for (int k = c; k < kSB; ++k) acc += T[i + k * 96] * LL(c0 + k, c0 + c);
      LL(r0 + i, c0 + c) = -acc;
    }
    __syncthreads();
  }
  for (int idx = t; idx < kNB * kNB; idx += 256) {
    const int i = idx % kNB, j = idx / kNB;
    Linv[idx] = (i < nb && j < nb && i >= j) ? LL(i, j) : 0.0;
  }
#undef LL
}

// ---- MFMA micro-kernel: acc(64x64 per wave) += X[i0.., 0..K) * Y[j0.., 0..K)^T ------------------------------
// X, Y column-major (element (r,k) at X[r + k*ld]); rows beyond the limits read as zero.
// acc[mi][nj] is the 16x16 tile (mi,nj) in the f64 MFMA C/D layout: col = lane&15, row = (lane>>4) + 4*reg.
__device__ __forceinline__ void mfma_xyT_64x64(const double* __restrict__ X, long ldx, long xrow0, long xrows,
                                               const double* __restrict__ Y, long ldy, long yrow0, long yrows,
                                               int K, double4_t acc[4][4]) {
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, kq = lane >> 4;
  const double* xp[4];
  const double* yp[4];
  bool xv[4], yv[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const long xr = xrow0 + m * 16 + r, yr = yrow0 + m * 16 + r;
    xv[m] = xr < xrows; yv[m] = yr < yrows;
    xp[m] = X + (xv[m] ? xr : 0) + (long)kq * ldx;
    yp[m] = Y + (yv[m] ? yr : 0) + (long)kq * ldy;
  }
  // software pipeline: the operands of step kk+4 are in flight while the 16 MFMAs of step kk issue
  double a[4], b[4], an[4], bn[4];
  {
    const bool kv = kq < K;
#pragma unroll
    for (int m = 0; m < 4; ++m) { a[m] = (xv[m] && kv) ? xp[m][0] : 0.0; b[m] = (yv[m] && kv) ? yp[m][0] : 0.0; }
  }
  for (int kk = 0; kk < K; kk += 4) {
    const int kn = kk + 4;
    const bool kv = (kn + kq) < K;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      an[m] = (xv[m] && kv) ? xp[m][(long)kn * ldx] : 0.0;
      bn[m] = (yv[m] && kv) ? yp[m][(long)kn * ldy] : 0.0;
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[nj], acc[mi][nj], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) { a[m] = an[m]; b[m] = bn[m]; }
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
// XCD-aware tile order: the dispatcher places workgroup b on XCD b % 8, each XCD has its own 4 MiB L2.  Tiles are
// grouped into 8x8 SUPER-TILES (1024 x 1024 elements; operand working set 2 x 1024 x 128 x 8 B = 2 MiB, fits one
// L2) and a super-tile is handed to ONE XCD: workgroup b -> xcd = b % 8, q = b / 8, super-tile (q / 64) * 8 + xcd,
// tile q % 64 inside it.  4 waves = 2x2 sub-tiles of 64x64 each.
static __global__ __launch_bounds__(256) void k_chol_update(double* __restrict__ M, long lda, long rows, long ncols, int k0,
                                                     int nb, int nst /* super-tile rows */) {
  const long b = blockIdx.x;
  const long xcd = b & 7, q = b >> 3;
  const long sb = (q >> 6) * 8 + xcd;         // super-tile index in the lower triangle of super-tiles
  const long nsb = (long)nst * (nst + 1) / 2;
  if (sb >= nsb) return;
  long SI = (long)((sqrt(8.0 * (double)sb + 1.0) - 1.0) * 0.5);
  while (SI * (SI + 1) / 2 > sb) --SI;
  while ((SI + 1) * (SI + 2) / 2 <= sb) ++SI;
  const long SJ = sb - SI * (SI + 1) / 2;
  const long ti = SI * 8 + ((q & 63) >> 3), tj = SJ * 8 + (q & 7);
  if (tj > ti) return;
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
}

// ---- trailing update, LDS-staged: C[base.., base..) -= P P^T with P = columns [kcol0, kcol0 + K) -------------------
// Workgroup tile 256 (rows) x 128 (cols), 8 waves as 4 x 2 sub-tiles of 64 x 64; the K dimension is streamed
// through LDS in chunks of 16 (double-buffered: global -> registers while the MFMAs of the previous chunk issue,
// registers -> LDS, one barrier per chunk).  Operand traffic per tile is (256 + 128) x K doubles instead of the
// 4 x (64 + 64) x K each wave used to pull on its own, and K itself is 256 (two 128-wide inner panels per
// trailing update), which halves the read-modify-write traffic on C per flop.
// The product is formed TRANSPOSED (B-fragment as the MFMA A operand): in the f64 D layout lane & 15 is then the
// row index of C, so 16 lanes touch 128 contiguous bytes of the column-major matrix.
constexpr int kBM = 256, kBN = 128, kKC = 16, kLdA = 272, kLdB = 144;
static __global__ __launch_bounds__(512) void k_chol_update_lds(double* __restrict__ M, long lda, long rows, long ncols,
                                                                int kcol0, int K, long base, int ntJ) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* As = sm;                               // [2][kKC][kLdA]
  double* Bs = sm + 2 * kKC * kLdA;              // [2][kKC][kLdB]
  const long b = blockIdx.x;
  long ti = (long)((sqrt(1.0 + 4.0 * (double)b) - 1.0) * 0.5);
  while (ti * (ti + 1) > b) --ti;
  while ((ti + 1) * (ti + 2) <= b) ++ti;
  const long tj = b - ti * (ti + 1);             // 0 .. 2 ti + 1
  if (tj >= ntJ) return;
  const long i0 = base + ti * kBM, j0 = base + tj * kBN;
  if (i0 >= rows || j0 >= ncols) return;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int wi = wave >> 1, wj = wave & 1;
  const double* P = M + (long)kcol0 * lda;
  // global -> register staging: A chunk = 256 x 16 doubles = 2048 double2 (4 per thread), B chunk = 1024 double2 (2 per thread)
  typedef double double2_t __attribute__((ext_vector_type(2)));
  double2_t ra[4], rb[2];
  auto gload = [&](int kc) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = t + 512 * q, pr = e & 127, k = e >> 7;
      const long row = i0 + 2 * pr;
      const bool kv = (kc + k) < K;
      double2_t v = {0.0, 0.0};
      if (kv && row + 1 < rows) v = *reinterpret_cast<const double2_t*>(P + row + (long)(kc + k) * lda);
      else if (kv && row < rows) v[0] = P[row + (long)(kc + k) * lda];
      ra[q] = v;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = t + 512 * q, pr = e & 63, k = e >> 6;
      const long row = j0 + 2 * pr;
      const bool kv = (kc + k) < K;
      double2_t v = {0.0, 0.0};
      if (kv && row + 1 < rows) v = *reinterpret_cast<const double2_t*>(P + row + (long)(kc + k) * lda);
      else if (kv && row < rows) v[0] = P[row + (long)(kc + k) * lda];
      rb[q] = v;
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = t + 512 * q, pr = e & 127, k = e >> 7;
      *reinterpret_cast<double2_t*>(As + (buf * kKC + k) * kLdA + 2 * pr) = ra[q];
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = t + 512 * q, pr = e & 63, k = e >> 6;
      *reinterpret_cast<double2_t*>(Bs + (buf * kKC + k) * kLdB + 2 * pr) = rb[q];
    }
  };
  double4_t acc[4][4];   // acc[nj][mi]: rows = j (tile columns), cols = i (tile rows)
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = double4_t{0, 0, 0, 0};
  const bool skip = (j0 + wj * 64) > (i0 + wi * 64 + 63) || (i0 + wi * 64) >= rows || (j0 + wj * 64) >= ncols;  // sub-tile above the diagonal / outside
  const int r = lane & 15, kq = lane >> 4;
  gload(0);
  sstore(0);
  __syncthreads();
  int buf = 0;
  for (int kc = 0; kc < K; kc += kKC) {
    const bool more = kc + kKC < K;
    if (more) gload(kc + kKC);
    if (!skip) {
      const double* Ab = As + buf * kKC * kLdA + wi * 64 + r;
      const double* Bb = Bs + buf * kKC * kLdB + wj * 64 + r;
#pragma unroll
      for (int kk = 0; kk < kKC; kk += 4) {
        double a[4], bq[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) { a[m] = Ab[(kk + kq) * kLdA + m * 16]; bq[m] = Bb[(kk + kq) * kLdB + m * 16]; }
#pragma unroll
        for (int nj = 0; nj < 4; ++nj)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) acc[nj][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(bq[nj], a[mi], acc[nj][mi], 0, 0, 0);
      }
    }
    if (more) sstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  if (skip) return;
  const long iw = i0 + wi * 64, jw = j0 + wj * 64;
  const int rq = lane >> 4;
#pragma unroll
  for (int nj = 0; nj < 4; ++nj)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const long col = jw + nj * 16 + rq + 4 * g;   // D row  -> j
        const long row = iw + mi * 16 + r;            // D col  -> i (contiguous across lanes)
        if (row < rows && col < ncols && row >= col) M[row + col * lda] -= acc[nj][mi][g];
      }
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
// x[k0 + r] = sum_c Linv[c][r] z[c]  (Linv^T z): one wave per r, lanes over c (contiguous in memory)
static __global__ __launch_bounds__(256) void k_chol_backsolve(const double* __restrict__ Linv, int k0, int nb,
                                                        const double* __restrict__ z, double* __restrict__ x) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= nb) return;
  double s = 0;
  for (int c = lane; c < nb; c += 64) s += Linv[c + r * kNB] * z[c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) x[k0 + r] = s;
}

// Host driver.  M: (n+1) x n col-major (lda), Linv_ws: ceil(n/NB) * NB*NB doubles, z_ws: NB doubles,
// x: n doubles (output), info: device int (bit 0 set on a non-positive pivot).
inline hipError_t chol_factor_solve(double* M, long lda, int n, double* Linv_ws, double* z_ws, double* x, int* info,
                                    hipStream_t st) {
  const long rows = (long)n + 1;
  const size_t lds = (size_t)(kNB * kLdsPad + 96 * kSB + kSB) * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_chol_potrf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  int np = (n + kNB - 1) / kNB;
  const size_t lds_u = (size_t)(2 * kKC * (kLdA + kLdB)) * sizeof(double);
  static bool attr2_set = false;
  if (!attr2_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_chol_update_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_u);
    if (e != hipSuccess) return e;
    attr2_set = true;
  }
  auto launch_update = [&](int kcol0, int K, long base, long col_limit) {
    // trailing region: rows [base, rows), cols [base, col_limit)
    const long nrows = rows - base, nc = col_limit - base;
    if (nrows <= 0 || nc <= 0) return;
    const long ntI = (nrows + kBM - 1) / kBM, ntJ = (nc + kBN - 1) / kBN;
    const long nblk = ntI * (ntI + 1);   // sum over ti of (2 ti + 2) tiles; tiles with tj >= ntJ exit immediately
    hipLaunchKernelGGL(k_chol_update_lds, dim3((unsigned)nblk), dim3(512), lds_u, st, M, lda, rows, col_limit, kcol0, K, base,
                       (int)ntJ);
  };
  for (int p = 0; p < np; p += 2) {
    // outer panel = up to two 128-wide inner panels; the big trailing update uses K = 256
    const int k0 = p * kNB, nb1 = (n - k0 < kNB) ? (n - k0) : kNB;
    double* Linv1 = Linv_ws + (size_t)p * kNB * kNB;
    hipLaunchKernelGGL(k_chol_potrf, dim3(1), dim3(256), lds, st, M, lda, k0, nb1, Linv1, info);
    long below = rows - (k0 + nb1);
    if (below > 0) hipLaunchKernelGGL(k_chol_panel, dim3((unsigned)((below + 255) / 256)), dim3(256), 0, st, M, lda, rows, k0, nb1, Linv1);
    const int k1 = k0 + nb1;
    if (k1 >= n) break;
    const int nb2 = (n - k1 < kNB) ? (n - k1) : kNB;
    // bring the second inner panel's columns up to date (rank-nb1 update restricted to those columns)
    launch_update(k0, nb1, (long)k1, (long)k1 + nb2);
    double* Linv2 = Linv_ws + (size_t)(p + 1) * kNB * kNB;
    hipLaunchKernelGGL(k_chol_potrf, dim3(1), dim3(256), lds, st, M, lda, k1, nb2, Linv2, info);
    below = rows - (k1 + nb2);
    if (below > 0) hipLaunchKernelGGL(k_chol_panel, dim3((unsigned)((below + 255) / 256)), dim3(256), 0, st, M, lda, rows, k1, nb2, Linv2);
    // rank-(nb1+nb2) update of everything to the right of the outer panel
    launch_update(k0, nb1 + nb2, (long)k1 + nb2, (long)n);
  }
  for (int p = np - 1; p >= 0; --p) {
    const int k0 = p * kNB, nb = (n - k0 < kNB) ? (n - k0) : kNB;
    hipLaunchKernelGGL(k_chol_backdot, dim3(nb), dim3(256), 0, st, M, lda, n, k0, nb, x, z_ws);
    hipLaunchKernelGGL(k_chol_backsolve, dim3((nb + 3) / 4), dim3(256), 0, st, Linv_ws + (size_t)p * kNB * kNB, k0, nb, z_ws, x);
  }
  return hipGetLastError();
}

}  // namespace esl
