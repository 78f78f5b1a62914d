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
//   k_potrf2  one workgroup: diagonal block -> L11 in registers / LDS, then Linv = L11^-1 (in-place, recursive doubling)
//   k_panel   P <- P * Linv^T           (rows below the diagonal block; MFMA, X*Y^T form)
//   k_update  C <- C - P_i * P_j^T      (lower-triangle tiles of the trailing matrix; MFMA)
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace esl {

constexpr int kNB = 128;       // panel width
constexpr int kLdsPad = 129;   // column stride of the LDS copy of the diagonal block (bank-conflict pad)

typedef double double4_t __attribute__((ext_vector_type(4)));

// ---- diagonal block -----------------------------------------------------------------------------------------
// info[0] |= 1 when a pivot is not positive (g2o: LDLT not positive => solve fails => step rejected).
constexpr int kSB = 32;   // (LDS sizing only: the 96 x 32 temporary of the triangular inverse)
// value of `v` in lane `src` (a compile-time constant after unrolling) -> every lane: two v_readlane_b32 (a few cycles
// each); __shfl() would go through ds_bpermute, ~100 cycles of LDS-crossbar latency on a dependent chain
__device__ __forceinline__ double readlane_f64(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
// diagnostic: wall_clock64() marks (100 MHz) of the last k_chol_potrf2 launch, read by esl_selftest_cholesky when
// ESL_CHOL_TIMING is set
__device__ long long g_potrf_clk[16];
#define POTRF_MARK(k) do { if (t == 0) g_potrf_clk[k] = (long long)wall_clock64(); } while (0)
// ---- k_chol_potrf2: right-looking over the WHOLE 128 x 128 block in rank-4 steps ------------------------------------------------
// 528 threads each OWN a 4 x 4 block of the lower triangle in 16 registers for the whole factorisation; only the current block
// column (4 x 128 doubles) and the inverse of its diagonal block (10 doubles) travel through LDS.  32 steps of two barriers each
// (see the loop).  History of this kernel, n = 2,994 factor + solve: blocked form of round 1 (k_chol_potrf: 32 x 32 sub-blocks
// factored by one wave with v_readlane broadcasts) 4.5 ms; one column per step with one barrier (590 ns per column, all of it
// latency of LDS write -> 10-wave barrier -> LDS read) 4.0 ms; rank-4 steps (column loop 76 -> 28 us, kernel 118 -> 72 us) 2.86 ms.  Around the loop: every thread
// fetches its own block (load 10 -> 2 us) and writes its part of L back from registers; the triangular inverse is computed in
// place by recursive doubling from the 4 x 4 inverses the factorisation produces anyway (see below).
// (Also measured: round 1's blocked loop moved into this kernel's frame on eight waves with the sub-block inverses
// overlapped -- slower, 4.9 ms: at two waves per SIMD and 244 registers the wave-synchronous column steps went 12 -> 17 us.
// Round 1's kernel itself was deleted in round 3 after its equivalence runs.)
constexpr int kP2Threads = 640;
// WT: L and Linv leave with write-through (sc1) stores -- the persistent factorisation hands them to other workgroups inside the launch
template <bool WT>
__device__ __forceinline__ void chol_store(double* p, double v) {
  if constexpr (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
// Write-through stores in PAIRS (round 6).  An 8-byte sc1 store costs 2.7 x a 16-byte one per byte (the guide's store table), and in the
// product layouts of this file a lane holds ONE row: 16 lanes = 16 consecutive rows of a column.  Two values of the lane's row that belong
// to two different columns (x -> column a, y -> column b) are exchanged with the neighbouring lane (lane ^ 1: rows 2 m, 2 m + 1) so that
// the even lane stores rows {2 m, 2 m + 1} of column a and the odd lane the same rows of column b, 16 bytes each: half the store
// instructions, all of them wide.  `pa`, `pb`: this lane's own addresses of (row, a) and (row, b); `ok`: the lane's row is inside the
// matrix; `wide`: 16-byte alignment holds (even leading dimension and even first row).  A pair whose second row is outside falls
// back to 8-byte stores.  The inline-assembly store carries the wait states of the wide-store data hazard (see chol_update_tile_v).
#ifdef ESL_NO_WT_PAIRS   // (A/B switch: the 8-byte write-through stores everywhere but the update tile)
constexpr bool kWtPairs = false;
#else
constexpr bool kWtPairs = true;
#endif
__device__ __forceinline__ double chol_xchg_lane1(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0xB1, 0xF, 0xF, true), hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void chol_store_wt_pair(double* pa, double* pb, double x, double y, bool ok, bool wide) {
  typedef double double2_t __attribute__((ext_vector_type(2)));
  const bool odd = (threadIdx.x & 1) != 0;
  const double got = chol_xchg_lane1(odd ? x : y);
  const bool nb_ok = __builtin_amdgcn_mov_dpp((int)ok, 0xB1, 0xF, 0xF, true) != 0;
  if (kWtPairs && wide && ok && nb_ok) {
    double* p = odd ? pb - 1 : pa;
    const double2_t v = odd ? double2_t{got, y} : double2_t{x, got};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
  } else if (ok) {
    chol_store<true>(pa, x);
    chol_store<true>(pb, y);
  }
}
// ---- the triangular inverse of the factored diagonal block, in place in the LDS copy (see chol_potrf2_body) -------------------------------
// A function of its own, NOT inlined (round 6): inlined into k_chol_persist its register pressure (two tile columns of double4 accumulators)
// moved the merged kernel's allocation and the chain's fused row solve picked up spills -- 15.5 -> 24.7 us per panel, which ate the
// 8 us the inverse had gained.  The LDS copy is reached through the kernel's one dynamic-LDS symbol (offset 0), not through a pointer
// argument (that would decay to flat addressing).
template <int NT>
static __device__ __attribute__((noinline)) void chol_inv_levels() {
  extern __shared__ __attribute__((aligned(16))) double sm_inv[];
  double* L = sm_inv;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#define LL(i, j) L[(i) + (j) * kLdsPad]
  // ---- Linv = L^-1 IN PLACE by recursive doubling: the inverse of [A 0; C D] is [A^-1 0; -D^-1 C A^-1  D^-1].  Level b merges
  // pairs of adjacent b x b diagonal inverses; C (original L) is overwritten by its block of the inverse, which no later level
  // reads as L.  Rounds 3-5: b = 4, 8 one thread per output entry; b = 16, 32, 64 two MFMA products, the intermediate T = C A^-1 parked
  // in the (unused) block above the diagonal that mirrors C; 8 barriers in all, against the 32-step triangular inverses of
  // the four 32 x 32 sub-blocks + 3 dependent block columns of the blocked form (25.6 -> 16.4 us; n = 2,994 2.86 -> 2.60 ms).
  // Round 6: every level on the matrix cores, and no intermediate through LDS.  In the f64 MFMA layout the D fragment of the first product
  // (T = C A^-1: lane (r, kq), register g = T[kq + 4 g][r]) IS the B fragment of k-step g of the second (X = D^-1 T wants T[4 g + kq][r]):
  // a wave that computes a tile column of T keeps it in registers and goes straight on to the same tile column of X.
  //   levels 4, 8   wave s < 8 owns the 16 x 16 diagonal super-block s; the level's C / A^-1 / D^-1 positions are selected by index masks
  //                 and the two products are 4 + 4 MFMAs on the whole super-block (block-diagonal operands), wave-local: no barrier
  //                 between the two levels (LDS operations of one wave complete in order)
  //   level 16      four groups, one wave each: 4 + 4 MFMAs
  //   levels 32, 64 (group, tile column) per wave: the column's tiles of T, then of X, all in registers; one barrier before C is overwritten
  // 16.4 -> ~6 us (the one-entry-per-thread loops of levels 4, 8 were 5.1 us, the per-MFMA LDS round trips of the others 10.9).
  {
    const int r = lane & 15, kq = lane >> 4;
    if (wave < 8) {
      const int s0 = 16 * wave;
#pragma unroll
      for (int b = 4; b <= 8; b *= 2) {
        // position classes of the level inside the super-block: same 2b-group, (upper | lower) half of it
        auto grp = [b](int i) { return i / (2 * b); };
        auto low = [b](int i) { return (i / b) & 1; };
        double4_t tt = {0, 0, 0, 0}, xx = {0, 0, 0, 0};
        double av[4], bv[4], dv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int k = 4 * g + kq;
          const bool cpos = grp(r) == grp(k) && low(r) == 1 && low(k) == 0;             // C(i = r, k)
          const bool apos = grp(k) == grp(r) && low(k) == 0 && low(r) == 0 && k >= r;   // A^-1(k, j = r), lower
          const bool dpos = grp(r) == grp(k) && low(r) == 1 && low(k) == 1 && k <= r;   // D^-1(i = r, m = k), lower
          const double vc = LL(s0 + r, s0 + k), va = LL(s0 + k, s0 + r);
          av[g] = cpos ? vc : 0.0; bv[g] = apos ? va : 0.0; dv[g] = dpos ? vc : 0.0;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) tt = __builtin_amdgcn_mfma_f64_16x16x4f64(av[g], bv[g], tt, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) xx = __builtin_amdgcn_mfma_f64_16x16x4f64(dv[g], tt[g], xx, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int i = kq + 4 * g;
          if (grp(i) == grp(r) && low(i) == 1 && low(r) == 0) LL(s0 + i, s0 + r) = -xx[g];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the wave's own writes, before its own reads of the next level)
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
    if (t == 0) g_potrf_clk[14] = (long long)wall_clock64();   // (diagnostic: end of levels 4, 8)
    if (wave < 4) {   // level 16: group `wave`, rows / columns [32 wave, 32 wave + 32)
      const int r0 = 32 * wave;
      double4_t tt = {0, 0, 0, 0}, xx = {0, 0, 0, 0};
      double av[4], bv[4], dv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int k = 4 * g + kq;
        av[g] = LL(r0 + 16 + r, r0 + k);
        const double va = LL(r0 + k, r0 + r), vd = LL(r0 + 16 + r, r0 + 16 + k);
        bv[g] = (k >= r) ? va : 0.0; dv[g] = (k <= r) ? vd : 0.0;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) tt = __builtin_amdgcn_mfma_f64_16x16x4f64(av[g], bv[g], tt, 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) xx = __builtin_amdgcn_mfma_f64_16x16x4f64(dv[g], tt[g], xx, 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) LL(r0 + 16 + kq + 4 * g, r0 + r) = -xx[g];   // (C's own wave: nobody else reads this group)
    }
    __syncthreads();
    if (t == 0) g_potrf_clk[15] = (long long)wall_clock64();   // (diagnostic: end of level 16)
    auto level_cols = [&](auto bc) {
      constexpr int B = decltype(bc)::value, NTL = B / 16;   // (kNB / (2 B)) x NTL = 4 wave tasks at both levels
      double4_t X[NTL];
      const int mg = wave / NTL, jb = wave % NTL, r0 = mg * 2 * B;
      if (wave < 4) {
        double4_t Tc[NTL];
#pragma unroll
        for (int m = 0; m < NTL; ++m) { Tc[m] = double4_t{0, 0, 0, 0}; X[m] = double4_t{0, 0, 0, 0}; }
        const int j = jb * 16 + r;
        // T(:, jb) = C(:, k) A^-1(k, jb), k from the column's own tile on (A^-1 is lower); operands of step kk + 4 in flight under step kk
        double an[NTL], bn, ac[NTL], bc2;
        {
          const int k = jb * 16 + kq;
          const double va = LL(r0 + k, r0 + j);
          bc2 = (k >= j) ? va : 0.0;
#pragma unroll
          for (int m = 0; m < NTL; ++m) ac[m] = LL(r0 + B + m * 16 + r, r0 + k);
        }
        for (int kk = jb * 16; kk < B; kk += 4) {
          const int k = (kk + 4 < B ? kk + 4 : kk) + kq;
          const double va = LL(r0 + k, r0 + j);
          bn = (k >= j) ? va : 0.0;
#pragma unroll
          for (int m = 0; m < NTL; ++m) an[m] = LL(r0 + B + m * 16 + r, r0 + k);
#pragma unroll
          for (int m = 0; m < NTL; ++m) Tc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[m], bc2, Tc[m], 0, 0, 0);
          bc2 = bn;
#pragma unroll
          for (int m = 0; m < NTL; ++m) ac[m] = an[m];
        }
        // X(ib, jb) = D^-1(ib, m) T(m, jb), m <= ib
#pragma unroll
        for (int ib = 0; ib < NTL; ++ib) {
          const int i = ib * 16 + r;
#pragma unroll
          for (int m = 0; m <= ib; ++m) {
            double dv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int mm = m * 16 + 4 * g + kq;
              const double vd = LL(r0 + B + i, r0 + B + mm);
              dv[g] = (mm <= i) ? vd : 0.0;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) X[ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(dv[g], Tc[m][g], X[ib], 0, 0, 0);
          }
        }
      }
      __syncthreads();   // every column's reads of C are done
      if (wave < 4) {
#pragma unroll
        for (int ib = 0; ib < NTL; ++ib)
#pragma unroll
          for (int g = 0; g < 4; ++g) LL(r0 + B + ib * 16 + kq + 4 * g, r0 + jb * 16 + r) = -X[ib][g];
      }
      __syncthreads();
    };
    level_cols(std::integral_constant<int, 32>{});
    level_cols(std::integral_constant<int, 64>{});
  }
#undef LL
}
// NT = 640: every one of the 528 blocks has its own thread (the stand-alone kernel).  NT = 512 (the persistent kernel's workgroup
// shape): threads 0..15 own TWO blocks one after the other -- their first one lies in block column 0 and is final after step 0, so
// they store it there and then (global memory + the LDS copy) and take over blocks 512..527 (block columns 26..31, which only see
// work at the very end) in time for step 0's rank-4 update.
template <bool WT, int NT = kP2Threads>
__device__ __forceinline__ void chol_potrf2_body(double* __restrict__ sm, double* __restrict__ M, long lda, int k0, int nb,
                                                 double* __restrict__ Linv /* kNB x kNB col-major */, int* __restrict__ info) {
  static_assert(NT == 640 || NT == 512, "potrf2: 640 or 512 threads");
  double* L = sm;                          // kNB x kNB, column stride kLdsPad: the factor, later its inverse
  double* T = sm + kNB * kLdsPad;          // 96 x 32 temporary of the inverse; the column buffers of the factorisation before that
  double* dinv = T + 96 * kSB;             // kNB reciprocals of the diagonal
  const int t = threadIdx.x;
#define LL(i, j) L[(i) + (j) * kLdsPad]
  POTRF_MARK(0);
  // the thread's 4 x 4 block (bi >= bj) of the 32 x 32 grid of blocks, enumerated column by column: the blocks of the first
  // block columns are finished first, so whole waves drop out of the update as the factorisation moves right
  const bool own = t < 528;
  int bi = 0, bj = 0;
  auto block_of = [](int idx, int& obi, int& obj) {
    int off = 0; obj = 0;
    while (off + (32 - obj) <= idx) { off += 32 - obj; ++obj; }
    obi = obj + (idx - off);
  };
  if (own) block_of(t, bi, bj);
  double a[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * bi + r, j = 4 * bj + c;
      a[r][c] = (own && i < nb && j < nb && i >= j) ? M[(long)(k0 + i) + (long)(k0 + j) * lda] : ((i == j) ? 1.0 : 0.0);   // identity padding
    }
  // NT = 512: the second block of threads 0..15 is fetched NOW, with everything else (fetched at the hand-over inside step 0 it was a
  // global round trip on the critical path of the whole workgroup: 4 - 6 us of the chain's every step)
  double a2[4][4];
  int bi2 = 0, bj2 = 0;
  if constexpr (NT == 512) {
    if (t < 16) block_of(512 + t, bi2, bj2);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 4 * bi2 + r, j = 4 * bj2 + c;
        a2[r][c] = (t < 16 && i < nb && j < nb && i >= j) ? M[(long)(k0 + i) + (long)(k0 + j) * lda] : ((i == j) ? 1.0 : 0.0);
      }
  }
  // Rank-4 steps: block column J = the four columns of the 4 x 4 blocks (., J).  Per step, two barriers instead of four:
  //   (a) the owner of the diagonal block (J, J) factors it in registers (four dependent rsqrt chains) and publishes its
  //       inverse (10 numbers)                                                                       -> barrier A
  //   (b) the blocks below it solve X = A L_JJ^-T with that inverse, keep X (their part of L) and publish its four columns
  //                                                                                                   -> barrier B
  //   (c) every block to the right subtracts the rank-4 product of its rows' and its columns' entries; the owner of the NEXT
  //       diagonal block goes straight on to (a), so barrier A of step J + 1 is the barrier that ends step J.
  double* colbuf = T;                      // [4][kNB]: the four columns of the current block column (rows below the diagonal block)
  double* dblk = T + 4 * kNB;              // [10]: inverse of the current diagonal block, packed rows (0,0) (1,0) (1,1) (2,0) ...
  double xd[4][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};   // diagonal-block owners: the inverse of their factored block
  // 4 x 4 Cholesky of the thread's own block (lower triangle), in place; returns the inverse of the factor, publishes 1 / L(c,c)
  auto factor_diag = [&](int J) {
    double is[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double d = a[c][c];
#pragma unroll
      for (int k = 0; k < c; ++k) d -= a[c][k] * a[c][k];
      if (!(d > 0)) atomicOr(info, 1);
      double r = __builtin_amdgcn_rsq(d);   // 1/sqrt(d): hardware estimate + two Newton steps (no fp64 sqrt, no divide)
      r = r * (1.5 - 0.5 * d * r * r);
      r = r * (1.5 - 0.5 * d * r * r);
      is[c] = r;
      a[c][c] = d * r;
#pragma unroll
      for (int rr = c + 1; rr < 4; ++rr) {
        double v = a[rr][c];
#pragma unroll
        for (int k = 0; k < c; ++k) v -= a[rr][k] * a[c][k];
        a[rr][c] = v * r;
      }
      dinv[4 * J + c] = r;
    }
    // X = L^-1 (lower): X(c,c) = 1 / L(c,c), X(r,c) = -(sum_{k=c}^{r-1} L(r,k) X(k,c)) / L(r,r)
    double x[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      x[c][c] = is[c];
#pragma unroll
      for (int rr = c + 1; rr < 4; ++rr) {
        double v = 0;
#pragma unroll
        for (int k = c; k < rr; ++k) v += a[rr][k] * x[k][c];
        x[rr][c] = -v * is[rr];
      }
    }
    int q = 0;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int c = 0; c <= rr; ++c) { dblk[q++] = x[rr][c]; xd[rr][c] = x[rr][c]; }
  };
  if (own && bi == 0 && bj == 0) factor_diag(0);
  __syncthreads();                                      // barrier A of step 0
  POTRF_MARK(1);
  for (int J = 0; J < kNB / 4; ++J) {
    if (t == 0 && (J == 1 || J == 8 || J == 16 || J == 24)) g_potrf_clk[10 + (J + 7) / 8 - (J == 1 ? 1 : 0)] = (long long)wall_clock64();   // (diagnostic: 10 = J 1, 11 = J 8, 12 = J 16, 13 = J 24)
    if (own && bj == J && bi > J) {                     // (b) X = A L_JJ^-T: X(r,c) = sum_{k <= c} A(r,k) Linv(c,k)
      double li[10];
#pragma unroll
      for (int q = 0; q < 10; ++q) li[q] = dblk[q];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double a0 = a[r][0], a1 = a[r][1], a2 = a[r][2], a3 = a[r][3];
        a[r][0] = a0 * li[0];
        a[r][1] = a0 * li[1] + a1 * li[2];
        a[r][2] = a0 * li[3] + a1 * li[4] + a2 * li[5];
        a[r][3] = a0 * li[6] + a1 * li[7] + a2 * li[8] + a3 * li[9];
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) colbuf[c * kNB + 4 * bi + r] = a[r][c];
    }
    __syncthreads();                                    // barrier B
    if constexpr (NT == 512) {
      if (J == 0 && t < 16) {   // first block (t, 0) is final: store it, take over block 512 + t (original values; step 0's update follows)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 4 * bi + r, j = c;
            if (i >= j) {
              if (i < nb && j < nb) chol_store<WT>(&M[(long)(k0 + i) + (long)(k0 + j) * lda], a[r][c]);
              LL(i, j) = (bi == 0) ? xd[r][c] : a[r][c];
            }
          }
        bi = bi2; bj = bj2;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) a[r][c] = a2[r][c];
      }
    }
    if (own && bj > J) {                                // (c) rank-4 update
      double rv[4][4], cv[4][4];                        // [k][r]: entries of column k of the block column in this block's rows / columns
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) { rv[k][r] = colbuf[k * kNB + 4 * bi + r]; cv[k][r] = colbuf[k * kNB + 4 * bj + r]; }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double v = a[r][c];
#pragma unroll
          for (int k = 0; k < 4; ++k) v -= rv[k][r] * cv[k][c];
          a[r][c] = v;
        }
      if (bi == bj && bj == J + 1) factor_diag(J + 1);  // (a) of the next step
    }
    __syncthreads();                                    // barrier A of step J + 1
  }
  POTRF_MARK(2); POTRF_MARK(3); POTRF_MARK(4);   // (marks 2..4 of the blocked form collapse: the whole column loop is "factor32[0]")
  // L11 back to memory straight from the owners' registers (the factor itself is part of the result), and into LDS for the
  // inverse: off-diagonal 4 x 4 blocks as they are, diagonal blocks as their INVERSES (computed during the factorisation)
  if (own) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 4 * bi + r, j = 4 * bj + c;
        if (i >= j) {
          if constexpr (WT) {
            // write-through: the block's rows (2 p, 2 p + 1) of a column as one 16-byte store where both are stored (see chol_store_wt_pair)
            const bool pair = !kWtPairs ? false : (r & 1) == 0 ? (i + 1 < nb && j < nb && (lda & 1) == 0) : (i - 1 >= j && i < nb && j < nb && (lda & 1) == 0);
            if (pair) {
              if ((r & 1) == 0) {
                typedef double double2_t __attribute__((ext_vector_type(2)));
                const double2_t v = double2_t{a[r][c], a[r | 1][c]};
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(&M[(long)(k0 + i) + (long)(k0 + j) * lda]), "v"(v) : "memory");
              }
            } else if (i < nb && j < nb) chol_store<WT>(&M[(long)(k0 + i) + (long)(k0 + j) * lda], a[r][c]);
          } else if (i < nb && j < nb) chol_store<WT>(&M[(long)(k0 + i) + (long)(k0 + j) * lda], a[r][c]);
          LL(i, j) = (bi == bj) ? xd[r][c] : a[r][c];
        }
      }
  }
  __syncthreads();
  POTRF_MARK(5); POTRF_MARK(6); POTRF_MARK(7);
  chol_inv_levels<NT>();
  POTRF_MARK(8);
  if constexpr (WT && kWtPairs) {   // (write-through: 16-byte stores, see chol_store_wt_pair; the workspace is 16-byte aligned, kNB is even)
    typedef double double2_t __attribute__((ext_vector_type(2)));
    for (int idx = 2 * t; idx < kNB * kNB; idx += 2 * NT) {
      const int i = idx % kNB, j = idx / kNB;
      const double2_t v = double2_t{(i < nb && j < nb && i >= j) ? LL(i, j) : 0.0, (i + 1 < nb && j < nb && i + 1 >= j) ? LL(i + 1, j) : 0.0};
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(&Linv[idx]), "v"(v) : "memory");
    }
  } else
  for (int idx = t; idx < kNB * kNB; idx += NT) {
    const int i = idx % kNB, j = idx / kNB;
    chol_store<WT>(&Linv[idx], (i < nb && j < nb && i >= j) ? LL(i, j) : 0.0);
  }
  POTRF_MARK(9);
#undef LL
}
static __global__ __launch_bounds__(kP2Threads) void k_chol_potrf2(double* __restrict__ M, long lda, int k0, int nb,
                                                            double* __restrict__ Linv /* kNB x kNB col-major */,
                                                            int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  chol_potrf2_body<false, kP2Threads>(sm, M, lda, k0, nb, Linv, info);
}

// (diagnostic, ESL_CHOL_POTRF512=1: the 512-thread shape and the write-through stores of the persistent kernel's chain as a kernel
//  of their own -- separates what the shape costs from what sharing a kernel with the 212-register update tile costs)
template <bool WT>
static __global__ __launch_bounds__(512) void k_chol_potrf2_512(double* __restrict__ M, long lda, int k0, int nb,
                                                                double* __restrict__ Linv, int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  chol_potrf2_body<WT, 512>(sm, M, lda, k0, nb, Linv, info);
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
// One WORKGROUP owns 64 rows x all kNB columns; its 8 waves take 32 x 32 sub-tiles (2 row groups x 4 column groups), so a
// wave's dependent chain is 128 MFMAs, not the 1,024 of the first version (one wave per 64 rows x 128 columns: 86 us per
// launch whatever the height of the panel).  The rows are updated in place: every wave reads all K input columns of its
// rows, the barrier makes sure all of the workgroup's reads are done before anybody writes.
template <bool WT>
__device__ __forceinline__ void chol_panel_body(double* __restrict__ M, long lda, long rows, int k0, int nb,
                                                const double* __restrict__ Linv, long strip) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane & 15, kq = lane >> 4;
  const long r0 = (long)k0 + nb + strip * 64 + (wave >> 2) * 32;
  const int cg = (wave & 3) * 32;
  const double* P = M + (long)k0 * lda;
  double4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = double4_t{0, 0, 0, 0};
  const double* xp[2];
  const double* yp[2];
  bool xv[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const long xr = r0 + m * 16 + r;
    xv[m] = xr < rows;
    xp[m] = P + (xv[m] ? xr : 0) + (long)kq * lda;
    yp[m] = Linv + (cg + m * 16 + r) + (long)kq * kNB;
  }
  if (nb == kNB) {   // (rows past the end read row 0 -- xp[] is clamped -- and are dropped at the store: one straggling wave on the
                     //  two-deep path below used to set the duration of the whole launch, 22 us whatever the rest did)
    // full panel, rows inside: 16 k-steps of operands in flight at a time (a slot is reloaded with step s + 16 right after step s
    // used it).  Two steps deep, the loop waited for a global load per step: 25 us per launch whatever the panel height.
    double ra[16][2], rb[16][2];
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2)
#pragma unroll
      for (int m = 0; m < 2; ++m) { ra[s2][m] = xp[m][(long)(4 * s2) * lda]; rb[s2][m] = yp[m][(long)(4 * s2) * kNB]; }
    __builtin_amdgcn_sched_barrier(0);   // all 64 loads are issued HERE (left alone, the scheduler sinks them to three steps ahead of their use)
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(rb[s2][nj], ra[s2][mi], acc[mi][nj], 0, 0, 0);
        if (half == 0) {
#pragma unroll
          for (int m = 0; m < 2; ++m) { ra[s2][m] = xp[m][(long)(4 * (s2 + 16)) * lda]; rb[s2][m] = yp[m][(long)(4 * (s2 + 16)) * kNB]; }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  } else {
  // software pipeline, two k-steps deep
  double a[2], b[2], an[2], bn[2];
  {
    const bool kv = kq < nb;
#pragma unroll
    for (int m = 0; m < 2; ++m) { a[m] = (xv[m] && kv) ? xp[m][0] : 0.0; b[m] = kv ? yp[m][0] : 0.0; }
  }
  for (int kk = 0; kk < nb; kk += 4) {
    const int kn = kk + 4;
    const bool kv = (kn + kq) < nb;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      an[m] = (xv[m] && kv) ? xp[m][(long)kn * lda] : 0.0;
      bn[m] = kv ? yp[m][(long)kn * kNB] : 0.0;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int nj = 0; nj < 2; ++nj) acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[nj], a[mi], acc[mi][nj], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 2; ++m) { a[m] = an[m]; b[m] = bn[m]; }
  }
  }
  __syncthreads();   // every wave of the workgroup has read its input rows
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // the product is formed TRANSPOSED (Linv fragment as the MFMA A operand): D's lane & 15 is then the panel ROW, so 16 lanes
        // store 128 contiguous bytes of a column (row-major-in-lanes stores touched 64 different cache lines per instruction)
        const long row = r0 + mi * 16 + r;
        const int col = cg + nj * 16 + kq + 4 * g;
        if constexpr (WT) {
          if (nb == kNB) {   // write-through, full panel: the lane's two column blocks leave as one 16-byte store per lane pair
            if (nj == 0)
              chol_store_wt_pair(&M[row + (long)(k0 + col) * lda], &M[row + (long)(k0 + col + 16) * lda], acc[mi][0][g], acc[mi][1][g], row < rows,
                                 ((lda | (r0 + mi * 16)) & 1) == 0);
            continue;
          }
        }
        if (row < rows && col < nb) chol_store<WT>(&M[row + (long)(k0 + col) * lda], acc[mi][nj][g]);
      }
}
static __global__ __launch_bounds__(512) void k_chol_panel(double* __restrict__ M, long lda, long rows, int k0, int nb,
                                                    const double* __restrict__ Linv) {
  chol_panel_body<false>(M, lda, rows, k0, nb, Linv, (long)blockIdx.x);
}

// ---- trailing update, LDS-staged: C[base.., base..) -= P P^T with P = columns [kcol0, kcol0 + K) -------------------
// Workgroup tile 256 (rows) x 128 (cols), 8 waves as 4 x 2 sub-tiles of 64 x 64; the K dimension is streamed
// through LDS in chunks of 16 (double-buffered: global -> registers while the MFMAs of the previous chunk issue,
// registers -> LDS, one barrier per chunk).  Operand traffic per tile is (256 + 128) x K doubles instead of the
// 4 x (64 + 64) x K each wave used to pull on its own, and K itself is 256 (two 128-wide inner panels per
// trailing update), which halves the read-modify-write traffic on C per flop.
// The product is formed TRANSPOSED (B-fragment as the MFMA A operand): in the f64 D layout lane & 15 is then the
// row index of C, so 16 lanes touch 128 contiguous bytes of the column-major matrix.
// Small trailing matrices (the reduced system of a C3-sized graph, n ~ 3k) use a 128 x 64 tile (32 x 32 per wave): the
// chain of one wave is then 4 MFMAs per k-step instead of 16 and a trailing matrix of 1,500 rows still spreads over ~140
// workgroups; with the big tile it occupied 20 CUs for 86 us whatever its size.
constexpr int kKC = 16;
__device__ int g_chol_stats_on;     // ESL_CHOL_TIMING (esl_selftest_cholesky): diagnostics of the solver's kernels on
__device__ long long g_upd_clk[4];  // k_chol_update_lds, one tile of the last launch: shader clock ticks, 100 MHz ticks, K, tile rows
// One tile of C -= P P^T: rows [i0, i0 + BM) x columns [j0, j0 + BN) of M (lower triangle only), P = (rows x K) column-major with
// leading dimension ldp; `assign`: C = -P P^T (split-K slices).  WT: C leaves with write-through stores (persistent factorisation).
template <int BM, int BN, int WM, int WN, bool WT>
__device__ __forceinline__ void chol_update_tile(double* __restrict__ sm, double* __restrict__ M, long lda, long rows, long ncols,
                                                 const double* __restrict__ P, long ldp, int K, long i0, long j0, bool assign) {
  constexpr int NT = 64 * WM * WN;                // threads per workgroup
  constexpr int kLdA = BM + 16, kLdB = BN + 16;   // +16: the 4 k-groups of a fragment read land in different banks
  constexpr int SM = BM / WM, SN = BN / WN;       // rows x columns of one wave's sub-tile
  constexpr int MI = SM / 16, NJ = SN / 16;
  constexpr int QA = BM * kKC / 2 / NT, QB = BN * kKC / 2 / NT;   // double2 per thread per chunk
  static_assert(QA >= 1 && QB >= 1, "chunk smaller than the workgroup");
  double* As = sm;                               // [2][kKC][kLdA]
  double* Bs = sm + 2 * kKC * kLdA;              // [2][kKC][kLdB]
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int wi = wave / WN, wj = wave % WN;
  // global -> register staging: A chunk = BM x 16 doubles, B chunk = BN x 16 doubles, as double2
  typedef double double2_t __attribute__((ext_vector_type(2)));
  double2_t ra[QA], rb[QB];
  auto gload = [&](int kc) {
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int e = t + NT * q, pr = e % (BM / 2), k = e / (BM / 2);
      const long row = i0 + 2 * pr;
      const bool kv = (kc + k) < K;
      // edge tiles: clamped addresses + selects, no branches (a branchy load per element made the edge workgroups the stragglers)
      const long kcl = kv ? (long)(kc + k) : (long)(K - 1);
      const double v0 = P[(row < rows ? row : rows - 1) + kcl * ldp], v1 = P[(row + 1 < rows ? row + 1 : rows - 1) + kcl * ldp];
      ra[q] = double2_t{(kv && row < rows) ? v0 : 0.0, (kv && row + 1 < rows) ? v1 : 0.0};
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int e = t + NT * q, pr = e % (BN / 2), k = e / (BN / 2);
      const long row = j0 + 2 * pr;
      const bool kv = (kc + k) < K;
      const long kcl = kv ? (long)(kc + k) : (long)(K - 1);
      const double v0 = P[(row < rows ? row : rows - 1) + kcl * ldp], v1 = P[(row + 1 < rows ? row + 1 : rows - 1) + kcl * ldp];
      rb[q] = double2_t{(kv && row < rows) ? v0 : 0.0, (kv && row + 1 < rows) ? v1 : 0.0};
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int q = 0; q < QA; ++q) {
      const int e = t + NT * q, pr = e % (BM / 2), k = e / (BM / 2);
      *reinterpret_cast<double2_t*>(As + (buf * kKC + k) * kLdA + 2 * pr) = ra[q];
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int e = t + NT * q, pr = e % (BN / 2), k = e / (BN / 2);
      *reinterpret_cast<double2_t*>(Bs + (buf * kKC + k) * kLdB + 2 * pr) = rb[q];
    }
  };
  double4_t acc[NJ][MI];   // acc[nj][mi]: rows = j (tile columns), cols = i (tile rows)
#pragma unroll
  for (int x = 0; x < NJ; ++x)
#pragma unroll
    for (int y = 0; y < MI; ++y) acc[x][y] = double4_t{0, 0, 0, 0};
  const bool skip = (j0 + wj * SN) > (i0 + wi * SM + SM - 1) || (i0 + wi * SM) >= rows || (j0 + wj * SN) >= ncols;  // sub-tile above the diagonal / outside
  const int r = lane & 15, kq = lane >> 4;
  // interior tiles (all rows of the tile inside the matrix, K a multiple of the chunk): the same staging without a single
  // guard -- the guarded form costs ~250 instructions and 22 branches per 64 MFMAs, in phase on all 8 waves after each barrier
  const bool interior = (i0 + BM <= rows) && (K % kKC == 0);
  const double* pA[QA];
  const double* pB[QB];
#pragma unroll
  for (int q = 0; q < QA; ++q) { const int e = t + NT * q; pA[q] = P + i0 + 2 * (e % (BM / 2)) + (long)(e / (BM / 2)) * ldp; }
#pragma unroll
  for (int q = 0; q < QB; ++q) { const int e = t + NT * q; pB[q] = P + j0 + 2 * (e % (BN / 2)) + (long)(e / (BN / 2)) * ldp; }
  auto gload_fast = [&](int kc) {
#pragma unroll
    for (int q = 0; q < QA; ++q) ra[q] = *reinterpret_cast<const double2_t*>(pA[q] + (long)kc * ldp);
#pragma unroll
    for (int q = 0; q < QB; ++q) rb[q] = *reinterpret_cast<const double2_t*>(pB[q] + (long)kc * ldp);
  };
  // Round 6, persistent kernel only: the C tile's lines are requested (one dword per line and lane group, into a register nobody
  // reads) under the LAST chunk's MFMAs.  The tile was written write-through by whichever CU visited it last, usually on another
  // XCD: the epilogue's NJ dependent load groups each paid a fabric round trip (~2 us) in a task visited 216 times per worker at
  // n = 18,000.  The loads are older than the epilogue's, and loads complete in order, so the dummy register is dead by the time the
  // first real C value arrives (it is kept allocated until then).
  int c_dummy = 0;
  auto c_prefetch = [&]() {
    const long iw0 = i0 + wi * SM, jw0 = j0 + wj * SN;
    const bool full_sub = !assign && (iw0 + SM <= rows) && (jw0 + SN <= ncols) && (jw0 + SN - 1 <= iw0);
    if (full_sub) {
      // ONE pointer walks the sub-tile's column groups (column jw0 + kq + 4 c, c = 0 .. 4 NJ - 1), the MI row groups are immediate
      // offsets: 64 separately formed addresses took the merged kernel's scratch from 144 to 656 B
      const double* pc = &M[(iw0 + r) + (jw0 + kq) * lda];
#pragma unroll 1
      for (int c2 = 0; c2 < 4 * NJ; ++c2) {
        if constexpr (MI == 4)
          asm volatile("global_load_dword %0, %1, off\n\tglobal_load_dword %0, %1, off offset:128\n\tglobal_load_dword %0, %1, off offset:256\n\t"
                       "global_load_dword %0, %1, off offset:384" : "+v"(c_dummy) : "v"(pc) : "memory");
        else if constexpr (MI == 2)
          asm volatile("global_load_dword %0, %1, off\n\tglobal_load_dword %0, %1, off offset:128" : "+v"(c_dummy) : "v"(pc) : "memory");
        pc += 4 * lda;
      }
    }
  };
  auto mainloop = [&](auto fast) {
    if constexpr (decltype(fast)::value) gload_fast(0); else gload(0);
    sstore(0);
    __syncthreads();
    int buf = 0;
    for (int kc = 0; kc < K; kc += kKC) {
      const bool more = kc + kKC < K;
      if (more) { if constexpr (decltype(fast)::value) gload_fast(kc + kKC); else gload(kc + kKC); }
#ifndef ESL_NO_CPREFETCH
      else if constexpr (WT) c_prefetch();
#endif   // (the last chunk has no operands to fetch: its MFMAs cover the C tile's way into this XCD's L2)
      if (!skip) {
        const double* Ab = As + buf * kKC * kLdA + wi * SM + r;
        const double* Bb = Bs + buf * kKC * kLdB + wj * SN + r;
        if constexpr (WM * WN <= 4) {
          // one wave per SIMD (accumulators in AGPRs): nothing but this wave's own instruction stream hides the LDS latency, so the
          // fragments of k-step s + 1 are requested before the MFMAs of k-step s issue (two register sets)
          double a[2][MI], bq[2][NJ];
#pragma unroll
          for (int m = 0; m < MI; ++m) a[0][m] = Ab[kq * kLdA + m * 16];
#pragma unroll
          for (int m = 0; m < NJ; ++m) bq[0][m] = Bb[kq * kLdB + m * 16];
#pragma unroll
          for (int kk = 0; kk < kKC; kk += 4) {
            const int cur = (kk >> 2) & 1;
            if (kk + 4 < kKC) {
#pragma unroll
              for (int m = 0; m < MI; ++m) a[cur ^ 1][m] = Ab[(kk + 4 + kq) * kLdA + m * 16];
#pragma unroll
              for (int m = 0; m < NJ; ++m) bq[cur ^ 1][m] = Bb[(kk + 4 + kq) * kLdB + m * 16];
            }
#pragma unroll
            for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
              for (int mi = 0; mi < MI; ++mi) acc[nj][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(bq[cur][nj], a[cur][mi], acc[nj][mi], 0, 0, 0);
          }
        } else {
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
        }
      }
      if (more) sstore(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  };
  if (interior) mainloop(std::true_type{}); else mainloop(std::false_type{});
  if (skip) return;
  const long iw = i0 + wi * SM, jw = j0 + wj * SN;
  const int rq = lane >> 4;
  // C -= acc.  Sub-tiles that lie entirely inside the matrix and strictly below the diagonal (all but a sliver) take the
  // unguarded form: 4 MI independent loads in flight per column group, then the subtracts and stores.  The guarded form
  // compiled to load / wait / subtract / store one element at a time -- 64 exposed memory round trips per lane, which
  // cost more than the whole K loop of the tile.
  const bool full = (iw + SM <= rows) && (jw + SN <= ncols) && (jw + SN - 1 <= iw);
  if (full) {
    // Round 6: the loads of column group nj + 1 are issued BEFORE the stores of group nj (two register sets).  Written as "load 16,
    // store 16" per group the compiler keeps every group's loads behind the previous group's stores (same array), i.e. NJ exposed
    // round trips to a tile another CU wrote last -- ~2 us each, in a task that is visited 216 times per worker at n = 18,000.
    // Launch-path kernels only (n = 32,768: 215.9 -> 213.6 ms): the persistent kernel owns all 256 registers of its two waves per SIMD
    // and the second register set went to scratch there (128 -> 224 B, task bodies 33.6 -> 34.3 ms) -- it keeps the one-set form.
    if constexpr (WT && kWtPairs && NJ == 2 && MI == 2) {
      // the persistent kernel's quarter tiles (32 x 32 per wave): both column groups' C values first, then the lane's two columns of a
      // row leave with its neighbour's as 16-byte write-through stores (chol_store_wt_pair)
      double c2[2][MI][4];
#pragma unroll
      for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int g = 0; g < 4; ++g) c2[nj][mi][g] = assign ? 0.0 : M[(iw + mi * 16 + r) + (jw + nj * 16 + rq + 4 * g) * lda];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          chol_store_wt_pair(&M[(iw + mi * 16 + r) + (jw + rq + 4 * g) * lda], &M[(iw + mi * 16 + r) + (jw + 16 + rq + 4 * g) * lda], c2[0][mi][g] - acc[0][mi][g],
                             c2[1][mi][g] - acc[1][mi][g], true, ((lda | iw) & 1) == 0);
      asm volatile("" :: "v"(c_dummy));
      return;
    }
    constexpr int NSET = WT ? 1 : 2;
    double cv[NSET][MI][4];
    auto load_group = [&](int nj, double (&dst)[MI][4]) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int g = 0; g < 4; ++g) dst[mi][g] = assign ? 0.0 : M[(iw + mi * 16 + r) + (jw + nj * 16 + rq + 4 * g) * lda];
    };
    if constexpr (NSET == 2) load_group(0, cv[0]);
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
      if constexpr (NSET == 2) { if (nj + 1 < NJ) load_group(nj + 1, cv[(nj + 1) & 1]); }
      else load_group(nj, cv[0]);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int g = 0; g < 4; ++g) chol_store<WT>(&M[(iw + mi * 16 + r) + (jw + nj * 16 + rq + 4 * g) * lda], cv[nj & (NSET - 1)][mi][g] - acc[nj][mi][g]);
      if (nj == 0) asm volatile("" :: "v"(c_dummy));   // (the prefetch's destination register: allocated until the first real C values are in)
    }
    return;
  }
  // edge and diagonal sub-tiles: the same batching with clamped addresses and predicated stores (an `if` around the whole
  // read-modify-write compiles to one exposed memory round trip per element, and the workgroup doing that is the last to finish)
#pragma unroll
  for (int nj = 0; nj < NJ; ++nj) {
    double cv[MI][4];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const long col = jw + nj * 16 + rq + 4 * g;   // D row  -> j
        const long row = iw + mi * 16 + r;            // D col  -> i (contiguous across lanes)
        cv[mi][g] = assign ? 0.0 : M[(row < rows ? row : rows - 1) + (col < ncols ? col : ncols - 1) * lda];
      }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const long col = jw + nj * 16 + rq + 4 * g;
        const long row = iw + mi * 16 + r;
        if (row < rows && col < ncols && row >= col) chol_store<WT>(&M[row + col * lda], cv[mi][g] - acc[nj][mi][g]);
      }
  }
}

template <int BM, int BN, int WM = 4, int WN = 2>
static __global__ __launch_bounds__(64 * WM * WN, (WM * WN <= 4 ? 1 : 2)) void k_chol_update_lds(double* __restrict__ M, long lda, long rows, long ncols,
                                                                           int kcol0, int K, long base, int ntJ, int rect,
                                                                           const double* __restrict__ Pext, long ldp,
                                                                           double* __restrict__ part, int kper,
                                                                           const int* __restrict__ kfirst = nullptr) {
  static_assert((BM == 2 * BN || BM == BN) && BM % 64 == 0, "tile shape");
  constexpr int RT = BM / BN;                     // tile row ti of the lower triangle holds RT (ti + 1) tiles
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const long b = blockIdx.x;
  long ti, tj;
  if (rect) {                                    // a few tile columns of a tall region: ntJ tiles per tile row
    ti = b / ntJ; tj = b - ti * ntJ;
    if (tj > RT * ti + RT - 1) return;           // above the diagonal
  } else {                                       // the whole lower triangle: rows 0 .. ti-1 hold RT ti (ti + 1) / 2 tiles
    ti = (long)((sqrt(1.0 + 8.0 * (double)b / RT) - 1.0) * 0.5);
    while (RT * ti * (ti + 1) / 2 > b) --ti;
    while (RT * (ti + 1) * (ti + 2) / 2 <= b) ++ti;
    tj = b - RT * ti * (ti + 1) / 2;             // 0 .. RT (ti + 1) - 1
    if (tj >= ntJ) return;
  }
  const long i0 = base + ti * BM, j0 = base + tj * BN;
  if (i0 >= rows || j0 >= ncols) return;
  // the rank-K factor: columns [kcol0, kcol0 + K) of M itself (Cholesky trailing update), or an external (rows x K) column-major
  // matrix Pext with leading dimension ldp (C -= Pext Pext^T: the reduced ellipsoid system of the camera-first elimination, esl_cf.hpp)
  const double* P = Pext ? Pext : M + (long)kcol0 * lda;
  if (!Pext) ldp = lda;
  // split-K (gridDim.y > 1; small outputs with a long K: a 450 x 450 system has 20 tiles): slice blockIdx.y takes K range
  // [y kper, (y + 1) kper) and writes -acc into ITS copy of the output (part + y lda ncols; no read-modify-write); a reduction
  // kernel adds the slices to C in slice order -- deterministic, unlike atomics
  if (part) {
    const int kbeg = (int)blockIdx.y * kper;
    P += (long)kbeg * ldp;
    K = (K - kbeg < kper) ? K - kbeg : kper;
    M = part + (size_t)blockIdx.y * (size_t)lda * (size_t)ncols;
    if (K <= 0) return;
  }
  // kfirst (external factor only, round 5): kfirst[g] = the first row of Pext that can be non-zero in any of the columns
  // [64 g, 64 g + 64) -- esl_cf.hpp orders T's ellipsoids by first camera, so the separators' rows of X are structurally zero above a
  // staircase.  A product term (row r, column c) needs both factors non-zero: the tile starts at the later of its two earliest
  // rows, rounded down to a chunk.  Only exact zeros are skipped: the result has the same bits.
  if (kfirst) {
    int ka = K, kb = K;
    for (long g = i0 / 64; g < (i0 + BM) / 64; ++g) ka = kfirst[g] < ka ? kfirst[g] : ka;
    for (long g = j0 / 64; g < (j0 + BN) / 64; ++g) kb = kfirst[g] < kb ? kfirst[g] : kb;
    int ks = ka > kb ? ka : kb;
    ks = (ks < K ? ks : K) & ~(kKC - 1);
    P += (long)ks * ldp;
    K -= ks;
    if (K <= 0) return;
  }
  // (diagnostic, ESL_CHOL_TIMING: shader clock ticks against 100 MHz wall-clock ticks over one tile of the last launch -- is the
  //  K loop's distance from the MFMA peak a matter of the clock the part sustains under this load?)
  const bool clk_on = BM == 256 && K >= 512 && gridDim.x >= 2048 && g_chol_stats_on != 0 && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0;   // (a tile in the middle of a launch that fills the chip)
  long long c0 = 0, w0 = 0;
  if (clk_on) { c0 = (long long)clock64(); w0 = (long long)wall_clock64(); }
  chol_update_tile<BM, BN, WM, WN, false>(sm, M, lda, rows, ncols, P, ldp, K, i0, j0, part != nullptr);
  if (clk_on) { g_upd_clk[0] = (long long)clock64() - c0; g_upd_clk[1] = (long long)wall_clock64() - w0; g_upd_clk[2] = K; g_upd_clk[3] = BM; }
}

// ---- round 6: the update tile in the shape the part's own DGEMM library runs (scripts/debug/rocblas_dgemm_probe.py) ------------------
// The vendor DGEMM reaches 89 - 96 % of the FP64-MFMA peak at these sizes where the 256 x 128 tile above reaches 79 %.  Its kernels
// (the dispatched symbols name the shape: MT128x128x16, four waves of 32 x 128, one operand "direct to VGPR") differ in two ways that
// matter here: (1) a workgroup is FOUR waves, one per SIMD, and TWO workgroups share a CU -- when one of them sits in its barrier the
// other one's wave on the same SIMD owns the matrix core, where the eight waves of the big tile all wait together; (2) the operand
// no other wave needs (this wave's 32 rows) never touches LDS: every lane loads its own MFMA fragment from global memory (a lane's
// fragment of the f64 16x16x4 MFMA is ONE element: row r = lane & 15, k = lane >> 4), only the 128-column operand the four waves share
// is staged.  Half the LDS stores and half the LDS fragment reads per MFMA.
// Tile: rows [i0, i0 + 128) x columns [j0, j0 + 128); wave w owns rows i0 + 32 w .. + 31, interleaved in pairs -- lane r holds rows
// 2 r and 2 r + 1 (one 16-byte load per k-step, and the C tile's read-modify-write is 16-byte accesses too).  Same product
// orientation as above (D rows = tile columns, D columns = tile rows), the same k order inside and across MFMAs: the same bits.
constexpr int kVT = 128, kVLd = kVT + 16;
// a value every lane holds alike, as the compiler cannot know (it came through LDS or a vector load): into scalar registers
__device__ __forceinline__ int chol_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long chol_uniform(long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffL)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long)v >> 32));
  return (long)(((unsigned long)hi << 32) | lo);
}
constexpr size_t kCholLdsV = (size_t)(2 * kKC * kVLd) * sizeof(double);
// HV = 1 (the persistent factorisation, one 512-thread workgroup per CU): the workgroup's two halves -- waves 0-3 and 4-7, one wave per
// SIMD each (measured: the eight waves of a workgroup sit on SIMDs 3 0 2 1 3 0 2 1) -- run this function side by side on the upper and
// the lower 128 rows of a 256 x 128 tile, each with its own LDS buffers and its own barrier: a counter in LDS (`bar`, zeroed by the
// caller behind a workgroup barrier) that the four waves of a half bump and poll.  The piecewise probe (scripts/debug/tile_probe_v.hip):
// 86.4 % of the FP64-MFMA peak for the eight-wave 256 x 128 loop, 92.3 % for two four-wave workgroups, 89.7 % for two halves of one
// workgroup behind the LDS-counter barrier.
template <bool WT, int HV = 0>
__device__ __forceinline__ void chol_update_tile_v(double* __restrict__ sm, double* __restrict__ M, long lda, long rows, long ncols,
                                                   const double* __restrict__ P, long ldp, int K, long i0, long j0, bool assign, int* bar = nullptr) {
  typedef double double2_t __attribute__((ext_vector_type(2)));
  constexpr int NT = 256, NJ = kVT / 16, QB = kVT * kKC / 2 / NT;   // 8 column blocks; 4 double2 of the shared operand per thread and chunk
  const int t = HV ? (int)(threadIdx.x & 255) : (int)threadIdx.x, wave = t >> 6, lane = t & 63, r = lane & 15, kq = lane >> 4;
  const long iw = i0 + wave * 32;
  // The unguarded loop only needs the tile's 128 rows of P to be READABLE: rows [rows, ldp) of a column-major array are padding inside
  // the allocation, and what they hold cannot matter -- row r of either operand only enters row / column r of the product, and rows and
  // columns >= `rows` of C are never stored.  With a leading dimension that is a multiple of 128 (esl_slam.hip pads its own) the
  // LAST tile row takes the unguarded loop like every other: as the guarded form it ran 2 - 3 x longer, 141 tiles of it at the very
  // end of the launch -- a fixed ~2 ms on the rank-3,744 update.
  const bool interior = (i0 + kVT <= ldp) && (K % (2 * kKC) == 0);
  const bool skipw = (j0 > iw + 31) || iw >= rows;   // this wave's rows lie above the diagonal / under the matrix: staging and barriers only
  int phase = 0;
  auto sync = [&]() {
    if constexpr (HV == 0) __syncthreads();
    else {
      phase += 4;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < phase) __builtin_amdgcn_s_sleep(0);
      asm volatile("" ::: "memory");
    }
  };
  // the shared operand: thread t stages rows j0 + 2 (e % 64), k = e / 64 (e = t + 256 q), i.e. k = t / 64 + 4 q; this lane's own
  // fragments: rows iw + 2 r, iw + 2 r + 1 at k = kc + 4 s + kq.  Addresses = a wave-uniform base (scalar registers) + ONE 32-bit
  // per-lane byte offset each (eight 64-bit pointers per lane went to scratch in the persistent kernel)
  const char* PBu = reinterpret_cast<const char*>(P + j0);
  const char* PAu = reinterpret_cast<const char*>(P + i0);
  const unsigned offB = (unsigned)((2 * (t % (kVT / 2)) + (long)(t / (kVT / 2)) * ldp) * (long)sizeof(double));
  const unsigned offA = (unsigned)((wave * 32 + 2 * r + (long)kq * ldp) * (long)sizeof(double));
  double2_t rb[QB];
  auto gload_b = [&](int kc, auto fast) {
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      if constexpr (decltype(fast)::value) rb[q] = *reinterpret_cast<const double2_t*>(PBu + (long)(kc + 4 * q) * ldp * (long)sizeof(double) + offB);
      else {
        const int e = t + NT * q, k = e / (kVT / 2);
        const long row = j0 + 2 * (e % (kVT / 2));
        const bool kv = (kc + k) < K;
        const long kcl = kv ? (long)(kc + k) : (long)(K - 1);
        const double v0 = P[(row < rows ? row : rows - 1) + kcl * ldp], v1 = P[(row + 1 < rows ? row + 1 : rows - 1) + kcl * ldp];
        rb[q] = double2_t{(kv && row < rows) ? v0 : 0.0, (kv && row + 1 < rows) ? v1 : 0.0};
      }
    }
  };
  auto gload_a = [&](double2_t (&a)[4], int kc, auto fast) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if constexpr (decltype(fast)::value) a[s] = *reinterpret_cast<const double2_t*>(PAu + (long)(kc + 4 * s) * ldp * (long)sizeof(double) + offA);
      else {
        const long row = iw + 2 * r;
        const int k = kc + 4 * s + kq;
        const bool kv = k < K;
        const long kcl = kv ? (long)k : (long)(K - 1);
        const double v0 = P[(row < rows ? row : rows - 1) + kcl * ldp], v1 = P[(row + 1 < rows ? row + 1 : rows - 1) + kcl * ldp];
        a[s] = double2_t{(kv && row < rows) ? v0 : 0.0, (kv && row + 1 < rows) ? v1 : 0.0};
      }
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const int e = t + NT * q;
      *reinterpret_cast<double2_t*>(sm + (buf * kKC + e / (kVT / 2)) * kVLd + 2 * (e % (kVT / 2))) = rb[q];
    }
  };
  double4_t acc[NJ][2];
#pragma unroll
  for (int x = 0; x < NJ; ++x) { acc[x][0] = double4_t{0, 0, 0, 0}; acc[x][1] = double4_t{0, 0, 0, 0}; }
  const bool full = !skipw && (iw + 32 <= rows) && (j0 + kVT <= ncols) && (j0 + kVT - 1 <= iw) && ((lda & 1) == 0) && (((iw & 1) == 0));
  // persistent kernel: the C sub-tile's lines requested under the LAST chunk's MFMAs (see chol_update_tile: the tile was written
  // write-through by another CU, usually on another XCD; one dword per 128-byte line and lane group into a register nobody reads)
  int c_dummy = 0;
  auto c_prefetch = [&]() {
    if (full && !assign) {
      const double* pc = &M[(iw + 2 * r) + (j0 + kq) * lda];
#pragma unroll 1
      for (int c2 = 0; c2 < 4 * NJ; ++c2) {
        asm volatile("global_load_dword %0, %1, off" : "+v"(c_dummy) : "v"(pc) : "memory");
        pc += 4 * lda;
      }
    }
  };
  // one chunk: MFMAs on a_cur and LDS buffer `buf`; the operands of chunk `kn` (the next one; the last chunk names itself) are
  // requested first and the shared one goes into the other LDS buffer behind the MFMAs.  No branch around the loads or the stores:
  // with `if (more)` the compiler's wait-count pass joins the two paths at the first MFMA and waits for the loads it has just issued
  // (vmcnt(3) with twelve in flight: one exposed memory round trip per chunk).
  auto chunk = [&](double2_t (&a_cur)[4], double2_t (&a_nxt)[4], int kn, int buf, auto fast) {
    gload_b(kn, fast);
    gload_a(a_nxt, kn, fast);
    if (!skipw) {
      const double* Bb = sm + buf * kKC * kVLd + r;
      double bq[2][NJ];
#pragma unroll
      for (int m = 0; m < NJ; ++m) bq[0][m] = Bb[kq * kVLd + m * 16];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s + 1 < 4) {
#pragma unroll
          for (int m = 0; m < NJ; ++m) bq[(s + 1) & 1][m] = Bb[(4 * (s + 1) + kq) * kVLd + m * 16];
        }
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) {
          acc[nj][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(bq[s & 1][nj], a_cur[s].x, acc[nj][0], 0, 0, 0);
          acc[nj][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(bq[s & 1][nj], a_cur[s].y, acc[nj][1], 0, 0, 0);
        }
      }
    }
    // (the stores stay behind the MFMAs: hoisted to the middle of the chunk, as the scheduler likes to, they wait for loads that are
    //  half a chunk old -- 19.1 vs 18.7 ms on the rank-3,744 update)
    __builtin_amdgcn_sched_barrier(0);
    sstore(buf ^ 1);
    sync();
  };
  auto mainloop = [&](auto fast) {
    double2_t a0[4], a1[4];
    gload_b(0, fast); gload_a(a0, 0, fast);
    sstore(0);
    sync();
    if constexpr (decltype(fast)::value) {
      // an even number of whole chunks: ONE copy of the chunk pair (separate copies for the last chunks cost the persistent kernel
      // 80 scratch stores + 35 reloads per visit around them); the last chunk fetches nothing new -- its own operands again
      for (int kc = 0; kc < K; kc += 2 * kKC) {
        const bool last = kc + 2 * kKC >= K;
        if constexpr (WT) { if (last) c_prefetch(); }
        chunk(a0, a1, kc + kKC, 0, fast);
        chunk(a1, a0, last ? kc + kKC : kc + 2 * kKC, 1, fast);
      }
    } else {
      const int klast = ((K + kKC - 1) / kKC - 1) * kKC;   // first k of the last chunk
      int kc = 0;
      for (; kc + kKC <= klast; kc += 2 * kKC) {
        chunk(a0, a1, kc + kKC, 0, fast);
        chunk(a1, a0, kc + 2 * kKC <= klast ? kc + 2 * kKC : klast, 1, fast);
      }
      if (kc <= klast) chunk(a0, a1, klast, 0, fast);   // (an odd number of chunks)
    }
  };
  if (interior) mainloop(std::true_type{}); else mainloop(std::false_type{});
  if (skipw || j0 >= ncols) return;
  // C -= acc: acc[nj][mi][g] is row iw + 2 r + mi, column j0 + 16 nj + kq + 4 g
  if (full) {
    double2_t cv[2][4];
    auto load_group = [&](int nj, double2_t (&dst)[4]) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        dst[g] = assign ? double2_t{0.0, 0.0} : *reinterpret_cast<const double2_t*>(&M[(iw + 2 * r) + (j0 + nj * 16 + kq + 4 * g) * lda]);
    };
    load_group(0, cv[0]);
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
      if (nj + 1 < NJ) load_group(nj + 1, cv[(nj + 1) & 1]);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        double* p = &M[(iw + 2 * r) + (j0 + nj * 16 + kq + 4 * g) * lda];
        const double2_t v = double2_t{cv[nj & 1][g].x - acc[nj][0][g], cv[nj & 1][g].y - acc[nj][1][g]};
        // write-through as ONE 16-byte sc1 store: 8-byte sc1 stores cost 2.7 x per byte (the guide's store table; n = 18,000 37.25 ->
        // 36.37 ms, 24,000 80.7 -> 79.4 against two 8-byte atomic stores).  Inline assembly, so the compiler sees neither the store's
        // place in the vmcnt queue nor its hazard: a VALU write to the data registers of a store wider than 8 bytes needs wait states
        // behind it (the hazard recognizer pads its own stores; without the s_nop the next address computation landed in the stored data)
        if constexpr (WT) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
        else *reinterpret_cast<double2_t*>(p) = v;
      }
      if (nj == 0) asm volatile("" :: "v"(c_dummy));   // (the prefetch's destination register: allocated until the first real C values are in)
    }
    return;
  }
#pragma unroll
  for (int nj = 0; nj < NJ; ++nj) {
    double cv[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const long col = j0 + nj * 16 + kq + 4 * g, row = iw + 2 * r + mi;
        cv[mi][g] = assign ? 0.0 : M[(row < rows ? row : rows - 1) + (col < ncols ? col : ncols - 1) * lda];
      }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const long col = j0 + nj * 16 + kq + 4 * g, row = iw + 2 * r + mi;
        if (row < rows && col < ncols && row >= col) chol_store<WT>(&M[row + col * lda], cv[mi][g] - acc[nj][mi][g]);
      }
  }
}
// the launch: same arguments and tile enumeration as k_chol_update_lds<128, 128> (square tiles: tile row ti holds ti + 1 tiles)
// Tile order (whole-triangle launches): workgroups are dealt to the eight XCDs round-robin, and each XCD has an L2 of its own.  Row-major
// over the triangle, the ~64 tiles an XCD holds at one time are 64 different tile columns of one tile row: 1 + 64 operand panels stream
// through that L2.  `sb` > 0: ids are dealt so that XCD x works on a block of sb x sb tiles (2 sb panels) -- workgroup b belongs to
// super-block 8 (b / (8 sb sb)) + b % 8, tile (b % (8 sb sb)) / 8 of it; super-blocks in row-major order over their own triangle.
__host__ __device__ inline long chol_v_grid(long ntI, int sb) {
  if (sb <= 0) return ntI * (ntI + 1) / 2;
  const long nS = (ntI + sb - 1) / sb, nSB = nS * (nS + 1) / 2;
  return (nSB + 7) / 8 * 8 * sb * sb;
}
static __global__ __launch_bounds__(256, 2) void k_chol_update_v(double* __restrict__ M, long lda, long rows, long ncols, int kcol0, int K, long base, int ntJ,
                                                                 int rect, const double* __restrict__ Pext, long ldp,
                                                                 const int* __restrict__ kfirst = nullptr, int sb = 0, int assign = 0) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const long b = blockIdx.x;
  long ti, tj;
  if (rect) {
    ti = b / ntJ; tj = b - ti * ntJ;
    if (tj > ti) return;
  } else if (sb > 0) {
    const long per = 8L * sb * sb, B = 8 * (b / per) + (b & 7), e = (b % per) >> 3;
    long sI = (long)((sqrt(1.0 + 8.0 * (double)B) - 1.0) * 0.5);
    while (sI * (sI + 1) / 2 > B) --sI;
    while ((sI + 1) * (sI + 2) / 2 <= B) ++sI;
    const long sJ = B - sI * (sI + 1) / 2;
    ti = sI * sb + e / sb; tj = sJ * sb + e % sb;
    if (tj > ti || tj >= ntJ) return;
  } else {
    ti = (long)((sqrt(1.0 + 8.0 * (double)b) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > b) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= b) ++ti;
    tj = b - ti * (ti + 1) / 2;
    if (tj >= ntJ) return;
  }
  const long i0 = base + ti * kVT, j0 = base + tj * kVT;
  if (i0 >= rows || j0 >= ncols) return;
  const double* P = Pext ? Pext : M + (long)kcol0 * lda;
  if (!Pext) ldp = lda;
  if (kfirst) {   // (the staircase of esl_cf.hpp, as in k_chol_update_lds)
    int ka = K, kb = K;
    for (long g = i0 / 64; g < (i0 + kVT) / 64; ++g) ka = kfirst[g] < ka ? kfirst[g] : ka;
    for (long g = j0 / 64; g < (j0 + kVT) / 64; ++g) kb = kfirst[g] < kb ? kfirst[g] : kb;
    int ks = ka > kb ? ka : kb;
    ks = (ks < K ? ks : K) & ~(kKC - 1);
    if (((K - ks) & kKC) != 0 && ks >= kKC) ks -= kKC;   // (the tile's unguarded loop takes chunks in pairs: an odd count sent half the tiles down the guarded path)
    P += (long)ks * ldp;
    K -= ks;
    if (K <= 0) {
      if (assign)   // C = -P P^T: a tile the staircase skips altogether still has to leave zeros
        for (int e = threadIdx.x; e < kVT * kVT; e += 256) {
          const long row = i0 + e % kVT, col = j0 + e / kVT;
          if (row < rows && col < ncols && row >= col) M[row + col * lda] = 0.0;
        }
      return;
    }
  }
#ifdef ESL_V_FORCE_WT
  chol_update_tile_v<true>(sm, M, lda, rows, ncols, P, ldp, K, i0, j0, assign != 0);
#else
  chol_update_tile_v<false>(sm, M, lda, rows, ncols, P, ldp, K, i0, j0, assign != 0);
#endif
}

// ---- small systems: factor + both substitutions in ONE workgroup, the whole lower triangle in registers (round 6) ---------------
// An order-180 system (20 ellipsoids: BASELINE configs[4] in SLAM mode) is two diagonal blocks of the blocked path: 2 x potrf2 + 2 panel
// solves + 1 update + the back-substitution = 6 dependent launches, ~170 us, with fewer than 1e7 flops in them.  Here 1024 threads own
// the 4 x 4 blocks of the lower triangle (and of the right-hand side's row, a block row of its own under the padded matrix) for the
// whole solve -- up to two blocks per thread, enumerated column by column, so n <= 248 -- and run the rank-4 steps of k_chol_potrf2
// over ALL block columns: (a) the diagonal block's owner factors it and publishes the inverse of its factor, (b) the blocks under it
// become X = A L_JJ^-T and publish their four columns, (c) every block to the right subtracts its rank-4 product.  y = L^-1 b rides
// along as the last block row.  L^T x = y runs on the same registers, block row by block row from the last: the diagonal owner forms
// x_J = L_JJ^-T z_J, the blocks (J, j < J) subtract L(J, j)^T x_J from z_j (one writer per entry and step: no atomics, fixed order).
// Only x leaves the kernel (nobody reads the factor of these systems afterwards; the panels' inverses are not formed at all).
constexpr int kCsThreads = 1024, kCsKB = 2, kCsMaxRows = 256;
__host__ __device__ inline int chol_small_blocks(int n) { const int nbc = (n + 3) / 4; return nbc * (nbc + 1) / 2 + nbc; }
inline bool chol_small_fits(int n) { return n >= 1 && chol_small_blocks(n) <= kCsThreads * kCsKB && 4 * ((n + 3) / 4 + 1) <= kCsMaxRows; }
static __global__ __launch_bounds__(kCsThreads) void k_chol_small(const double* __restrict__ M, long lda, int n, double* __restrict__ x,
                                                                  int* __restrict__ info) {
  __shared__ double colbuf[4 * kCsMaxRows];   // [k][4 bi + r]: the four columns of the current block column
  __shared__ double dblk[12];                 // inverse of the current diagonal block's factor, packed rows
  __shared__ double zs[kCsMaxRows];           // back-substitution: what is left of y
  __shared__ double xs[4];
  const int t = threadIdx.x;
  const int NBC = (n + 3) / 4, NBR = NBC + 1, total = chol_small_blocks(n);
  int bi[kCsKB], bj[kCsKB];
  bool own[kCsKB];
  double a[kCsKB][4][4];
#pragma unroll
  for (int q = 0; q < kCsKB; ++q) {
    const int idx = t + q * kCsThreads;
    own[q] = idx < total;
    int off = 0, c = 0;
    if (own[q]) while (off + (NBR - c) <= idx) { off += NBR - c; ++c; }
    bj[q] = c; bi[q] = own[q] ? c + (idx - off) : 0;
    const bool rhs = bi[q] == NBC;              // the right-hand side's row: local row 0 of the last block row
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 4 * bi[q] + r, j = 4 * bj[q] + c2;
        const bool valid = own[q] && j < n && (rhs ? r == 0 : (i < n && i >= j));
        const long row = rhs ? (long)n : (long)i;
        const double v = valid ? M[row + (long)j * lda] : 0.0;
        a[q][r][c2] = valid ? v : ((!rhs && i == j) ? 1.0 : 0.0);   // identity padding behind the matrix
      }
  }
  // 4 x 4 Cholesky of block q (lower, in place); the block is REPLACED by the inverse of its factor (all that is needed afterwards)
  auto factor_diag = [&](double (&b)[4][4]) {
    double is[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double d = b[c][c];
#pragma unroll
      for (int k = 0; k < c; ++k) d -= b[c][k] * b[c][k];
      if (!(d > 0)) atomicOr(info, 1);
      double rq = __builtin_amdgcn_rsq(d);
      rq = rq * (1.5 - 0.5 * d * rq * rq);
      rq = rq * (1.5 - 0.5 * d * rq * rq);
      is[c] = rq;
      b[c][c] = d * rq;
#pragma unroll
      for (int rr = c + 1; rr < 4; ++rr) {
        double v = b[rr][c];
#pragma unroll
        for (int k = 0; k < c; ++k) v -= b[rr][k] * b[c][k];
        b[rr][c] = v * rq;
      }
    }
    double xi[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      xi[c][c] = is[c];
#pragma unroll
      for (int rr = c + 1; rr < 4; ++rr) {
        double v = 0;
#pragma unroll
        for (int k = c; k < rr; ++k) v += b[rr][k] * xi[k][c];
        xi[rr][c] = -v * is[rr];
      }
    }
    int p = 0;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        b[rr][c] = c <= rr ? xi[rr][c] : 0.0;
        if (c <= rr) dblk[p++] = xi[rr][c];
      }
  };
  if (t == 0) factor_diag(a[0]);   // block (0, 0) is thread 0's first
  __syncthreads();
  for (int J = 0; J < NBC; ++J) {
#pragma unroll
    for (int q = 0; q < kCsKB; ++q)
      if (own[q] && bj[q] == J && bi[q] > J) {   // (b) X = A L_JJ^-T
        double li[10];
#pragma unroll
        for (int p = 0; p < 10; ++p) li[p] = dblk[p];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double a0 = a[q][r][0], a1 = a[q][r][1], a2 = a[q][r][2], a3 = a[q][r][3];
          a[q][r][0] = a0 * li[0];
          a[q][r][1] = a0 * li[1] + a1 * li[2];
          a[q][r][2] = a0 * li[3] + a1 * li[4] + a2 * li[5];
          a[q][r][3] = a0 * li[6] + a1 * li[7] + a2 * li[8] + a3 * li[9];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) colbuf[c * kCsMaxRows + 4 * bi[q] + r] = a[q][r][c];
      }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kCsKB; ++q)
      if (own[q] && bj[q] > J) {                 // (c) rank-4 update
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          double rv[4], cv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) { rv[r] = colbuf[k * kCsMaxRows + 4 * bi[q] + r]; cv[r] = colbuf[k * kCsMaxRows + 4 * bj[q] + r]; }
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) a[q][r][c] -= rv[r] * cv[c];
        }
        if (bi[q] == bj[q] && bj[q] == J + 1) factor_diag(a[q]);   // (a) of the next step
      }
    __syncthreads();
  }
  // ---- L^T x = y.  zs <- y (local row 0 of the last block row); diagonal blocks hold the inverses of their factors
#pragma unroll
  for (int q = 0; q < kCsKB; ++q)
    if (own[q] && bi[q] == NBC) {
#pragma unroll
      for (int c = 0; c < 4; ++c) zs[4 * bj[q] + c] = a[q][0][c];
    }
  __syncthreads();
  for (int J = NBC - 1; J >= 0; --J) {
#pragma unroll
    for (int q = 0; q < kCsKB; ++q)
      if (own[q] && bi[q] == J && bj[q] == J) {   // x_J = L_JJ^-T z_J
        double z[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) z[r] = zs[4 * J + r];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double v = 0;
#pragma unroll
          for (int r = c; r < 4; ++r) v += a[q][r][c] * z[r];
          xs[c] = v;
          if (4 * J + c < n) x[4 * J + c] = v;
        }
      }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kCsKB; ++q)
      if (own[q] && bi[q] == J && bj[q] < J) {    // z_j -= L(J, j)^T x_J
        double xv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) xv[r] = xs[r];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double v = zs[4 * bj[q] + c];
#pragma unroll
          for (int r = 0; r < 4; ++r) v -= a[q][r][c] * xv[r];
          zs[4 * bj[q] + c] = v;
        }
      }
    __syncthreads();
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

// ---- backward substitution L^T x = y in ONE launch (round 4) ---------------------------------------------------------------
// The launch-per-panel form above is 2 x ceil(n / 128) dependent launches whose kernels are each a few microseconds of work
// (n = 18,000: 282 launches, ~6 ms of a 46 ms solve).  Here a workgroup owns one panel p at a time (claimed from a counter, see below); it streams
// the blocks L[panel r, panel p] for r = np - 1 .. p + 1 in the order the x_r appear, keeps per-lane partial sums of
// z_p = y_p - sum_r L_rp^T x_r in registers (ONE cross-lane reduction at the end, fixed order => bit-reproducible), then
// x_p = Linv_p^T z_p from an LDS copy of Linv_p staged at the start, and publishes x_p.  Block r - 1 is fetched before the
// flag of x_{r-1} is polled, so the only things on the chain per panel are: flag -> 1 KB of x -> 32 FMAs -> reduction -> the
// 128 x 128 product from LDS -> publish.
// Hand-off (guide G16, recipe R1): x_p is stored write-through (relaxed agent-scope atomic stores = sc1), every storing wave
// drains its stores, one lane stores the flag; the consumer polls the flag relaxed from ONE lane and reads x_r with sc1 loads
// (L1 bypassed: no acquire fence on the chain).  Flags are zeroed by a memset in front of every launch.  Every spin is bounded:
// on a timeout bit 1 of info is set and every waiter gives up (the host reports an error instead of hanging the device).
// Panels are CLAIMED, last panel first, from one counter (flags[np]; round 6): a workgroup takes the next unclaimed panel
// when it arrives and again whenever it has finished one.  Panel p waits only for panels > p, and every one of those was claimed
// earlier by a workgroup that had already started, i.e. is resident and will finish it -- no assumption about dispatch order, about
// all G workgroups fitting the device at once, or about a second round of a workgroup's own panels (round 5's static
// `p -= gridDim.x` made ticket b's second panel wait for first-round panels of tickets that might not have started on a CU-masked
// device).  G = min(np, CUs of the device).
// Read a word other workgroups of the SAME launch write: a device-scope atomic ADD OF ZERO, written as inline assembly.
// Measured on MI355X (round 4): an agent-scope (sc1) LOAD in a poll loop -- and every atomic the compiler or the L2 can treat as a
// read: fetch_or 0 is folded into a load, a compare-and-swap that fails writes nothing -- can keep returning the value the line had
// when the poller's XCD first fetched it, for as long as nothing evicts the line: with all 256 workgroups waiting (nobody streams
// data) the chain sat on "3" for 300 ms while a copy from another stream showed 4.  An atomic that WRITES has to own the line, and
// the task counter of these very kernels shows those are coherent across the XCDs.  ~1-3 us per poll; pollers sleep in between.
__device__ __forceinline__ int chol_peek(int* word) {
  int seen;
  const int zero = 0;
  asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(seen) : "v"(word), "v"(zero) : "memory");
  return seen;
}
constexpr int kBsThreads = 512;
constexpr int kBsMaxGrid = 256;
constexpr size_t kBsLds = (size_t)(kNB * kNB + kNB) * sizeof(double);
// sum over the 64 lanes of 16 per-lane values in 17 shuffles: after the four halving steps lane l holds the partial of value
// 8 b5 + 4 b4 + 2 b3 + b2 (b_k = bit k of l) summed over the lanes that agree with it in bits 1..0; two plain butterflies finish.
// The total of value v ends up in every lane of the quad { l : bits 5..2 of l spell v }.
__device__ __forceinline__ double bs_reduce16(double (&a)[16], int lane) {
  const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8, h2 = lane & 4;
  double b[8], c[4], d[2];
#pragma unroll
  for (int i = 0; i < 8; ++i) { const double mine = h5 ? a[8 + i] : a[i], send = h5 ? a[i] : a[8 + i]; b[i] = mine + __shfl_xor(send, 32, 64); }
#pragma unroll
  for (int i = 0; i < 4; ++i) { const double mine = h4 ? b[4 + i] : b[i], send = h4 ? b[i] : b[4 + i]; c[i] = mine + __shfl_xor(send, 16, 64); }
#pragma unroll
  for (int i = 0; i < 2; ++i) { const double mine = h3 ? c[2 + i] : c[i], send = h3 ? c[i] : c[2 + i]; d[i] = mine + __shfl_xor(send, 8, 64); }
  double e = (h2 ? d[1] : d[0]) + __shfl_xor(h2 ? d[0] : d[1], 4, 64);
  e += __shfl_xor(e, 2, 64);
  e += __shfl_xor(e, 1, 64);
  return e;
}
static __global__ __launch_bounds__(kBsThreads) void k_chol_backsub(const double* __restrict__ M, long lda, int n, int np,
                                                                    const double* __restrict__ Linv_ws, double* x,
                                                                    int* flags, int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  typedef double double2_t __attribute__((ext_vector_type(2)));
  double* Li = sm;                 // Linv_p, 128 x 128 column-major
  double* zs = sm + kNB * kNB;     // z_p
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  bool aborted = false;            // (thread 0 only; no static __shared__ here: it would shift the 16-byte alignment of sm, guide G17)
  const int c0 = 16 * wave;                                                   // this wave's 16 columns of the panel
  const int vcol = c0 + ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);   // bs_reduce16: whose total this lane ends up with
  for (;;) {
    __syncthreads();                                                           // the previous panel's readers of Li / zs are done
    if (t == 0) reinterpret_cast<int*>(zs)[0] = atomicAdd(&flags[np], 1);     // claim the next panel (the word behind the np panel flags)
    __syncthreads();
    const int p = np - 1 - reinterpret_cast<const int*>(zs)[0];
    __syncthreads();                                                           // everybody holds p before zs is written again
    if (p < 0) break;
    const int k0 = p * kNB, nb = (n - k0 < kNB) ? (n - k0) : kNB;
    {
      const double2_t* src = reinterpret_cast<const double2_t*>(Linv_ws + (size_t)p * kNB * kNB);
      double2_t* dst = reinterpret_cast<double2_t*>(Li);
#pragma unroll 4
      for (int e = t; e < kNB * kNB / 2; e += kBsThreads) dst[e] = src[e];
    }
    const double yv = (vcol < nb) ? M[(long)n + (long)(k0 + vcol) * lda] : 0.0;   // row n of the factored matrix: y = L^-1 b
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0;
    double2_t blk[16];
    auto load_blk = [&](int r) {   // rows r 128 + 2 lane, + 1 of this wave's 16 columns
      const long row = (long)r * kNB + 2 * lane;
      const double* src = M + row + (long)(k0 + c0) * lda;
      if ((long)r * kNB + kNB <= (long)n) {
#pragma unroll
        for (int i = 0; i < 16; ++i) blk[i] = *reinterpret_cast<const double2_t*>(src + (long)i * lda);
      } else {                      // the last panel's partial block: rows >= n hold y^T and padding
        const bool v0 = row < n, v1 = row + 1 < n;
#pragma unroll
        for (int i = 0; i < 16; ++i) { blk[i].x = v0 ? src[(long)i * lda] : 0.0; blk[i].y = v1 ? src[(long)i * lda + 1] : 0.0; }
      }
    };
    if (np - 1 > p) load_blk(np - 1);
    for (int r = np - 1; r > p; --r) {
      if (t == 0 && !aborted) {
        const long long t0 = (long long)wall_clock64();
        unsigned spins = 0;
        while (chol_peek(&flags[r]) == 0) {
          __builtin_amdgcn_s_sleep(2);
          if ((++spins & 255u) == 0) {
            if (chol_peek(info) & 2) { aborted = true; break; }
            if ((long long)wall_clock64() - t0 > 300000000LL) { atomicOr(info, 2); aborted = true; break; }   // 3 s at 100 MHz
          }
        }
      }
      __syncthreads();
      const long xr = (long)r * kNB + 2 * lane;
      const double x0 = (xr < n) ? __hip_atomic_load(x + xr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
      const double x1 = (xr + 1 < n) ? __hip_atomic_load(x + xr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] += blk[i].x * x0 + blk[i].y * x1;
      if (r - 1 > p) load_blk(r - 1);
    }
    const double tot = bs_reduce16(acc, lane);
    if ((lane & 3) == 0) zs[vcol] = yv - tot;
    __syncthreads();                                                           // z_p complete (and Li staged long ago)
    {
      const double z0 = zs[2 * lane], z1 = zs[2 * lane + 1];
      double o[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {                                           // x[r] = sum_c Linv(c, r) z[c]: column r of Linv is contiguous in c
        const double2_t v = *reinterpret_cast<const double2_t*>(Li + 2 * lane + (c0 + i) * kNB);
        o[i] = v.x * z0 + v.y * z1;
      }
      const double xo = bs_reduce16(o, lane);
      if ((lane & 3) == 0 && vcol < nb) __hip_atomic_store(x + k0 + vcol, xo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           // every storing wave drains its stores ...
    __syncthreads();
    if (t == 0) __hip_atomic_store(&flags[p], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... then ONE lane publishes
  }
}

// Distributed form (one process per GPU; SURVEY.md section 8 e).  Outer panel o (W x 128 columns) belongs to rank o mod R.
// Every rank holds the full-size matrix, but only the columns of ITS outer panels are kept up to date: the owner factors
// its panel (diagonal blocks, panel solves, inner updates), broadcasts the factored columns (+ the inverses of the diagonal
// blocks for the back-substitution), and every rank applies the rank-(W x 128) update to its own later panels only --
// n^3 / (3 R) flops per rank instead of the replicated n^3 / 3.  After the last panel every rank holds the whole factor,
// so the O(n^2) back-substitution runs replicated and all ranks end with the same x, bit for bit.
struct CholDist {
  int rank = 0, n_ranks = 1;
  void* user = nullptr;
  int (*bcast)(void* user, double* dev, size_t count, int root) = nullptr;   // on the factorisation's stream
  // staging buffer of chol_pack_len(n) doubles: a factored outer panel travels as ONE message holding its rows from the diagonal
  // down (+ the right-hand side's row) and the inverses of its diagonal blocks -- half of the whole-column volume on average.
  // Null: whole columns and the inverses in two messages.
  double* pack = nullptr;
  // OVERLAPPED form (round 3): a second staging buffer and a broadcast that takes the stream to run on.  The messages then travel
  // on their own stream: the owner of panel o + 1 brings it up to date with panel o FIRST, factors and packs it, and its message is
  // on the wire while every rank is still applying panel o to the rest of its panels (the single-GPU look-ahead carried over).
  double* pack2 = nullptr;
  int (*bcast_on)(void* user, double* dev, size_t count, int root, hipStream_t stream) = nullptr;
};
inline int chol_outer_panels(int n);
inline size_t chol_pack_len(int n) { const size_t w = (size_t)chol_outer_panels(n) * kNB; return ((size_t)n + 1) * w + w * kNB; }
// pack / unpack of rows [r0, rows) x columns [c0, c0 + w) of M (dir 0: M -> buf, 1: buf -> M); buf column-major, height rows - r0
static __global__ __launch_bounds__(256) void k_chol_pack(double* __restrict__ M, long lda, long rows, long r0, long c0, double* __restrict__ buf, int dir) {
  const long h = rows - r0, i = (long)blockIdx.x * 256 + threadIdx.x, cidx = blockIdx.y;
  if (i >= h) return;
  if (dir == 0) buf[i + cidx * h] = M[(r0 + i) + (c0 + cidx) * lda];
  else M[(r0 + i) + (c0 + cidx) * lda] = buf[i + cidx * h];
}
inline int chol_outer_panels(int n) { return (n >= 8192) ? 4 : 2; }   // inner 128-panels per outer panel

// C += sum over the K slices of their partial products (slice order), lower triangle only
static __global__ __launch_bounds__(256) void k_chol_splitk_reduce(double* __restrict__ M, long lda, long rows, long ncols,
                                                                   const double* __restrict__ part, int nsplit) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long col = t / rows, row = t - col * rows;
  if (col >= ncols || row < col) return;
  double s = 0;
  for (int y = 0; y < nsplit; ++y) s += part[(size_t)y * (size_t)lda * (size_t)ncols + row + col * lda];
  M[row + col * lda] += s;
}

// C[base.., base..col_limit) -= P P^T on the lower triangle of the (rows x *) matrix M; P = columns [kcol0, kcol0 + K) of M, or
// Pext (rows x K, leading dimension ldp) when given.  Needs chol_set_attributes() on the current device first.
// part (optional, with Pext and base = 0): workspace of kCholMaxSplit x lda x col_limit doubles -- few tiles and a long K are split
// over K (see the kernel).
constexpr size_t kCholLdsBig = (size_t)(2 * kKC * (256 + 16 + 128 + 16)) * sizeof(double);
constexpr size_t kCholLdsSmall = (size_t)(2 * kKC * (128 + 16 + 64 + 16)) * sizeof(double);
constexpr int kCholMaxSplit = 16;
// kfirst (optional, with Pext): per group of 64 columns the first row of Pext that can be non-zero there (see the kernel); the array
// covers rows + 256 columns.
inline bool chol_update_v_enabled() { static const int use_v = [] { const char* e = std::getenv("ESL_UPD_V"); return (e && std::atoi(e) == 0) ? 0 : 1; }(); return use_v != 0; }
// would chol_launch_update take the k_chol_update_v path for this region?  (The only path that can ASSIGN C = -P P^T instead of updating C.)
inline bool chol_update_v_applies(long lda, long rows, long base, long col_limit, bool ext, long ldp) {
  const long nrows = rows - base, nc = col_limit - base;
  if (nrows <= 0 || nc <= 0) return false;
  const bool whole = (nc >= nrows - 1);
  const long big_tiles = ((nrows + 255) / 256) * ((nc + 127) / 128) / (whole ? 2 : 1);
  static const int waves4 = [] { const char* e = std::getenv("ESL_UPD_WAVES"); return (e && std::atoi(e) == 4) ? 1 : 0; }();
  return big_tiles >= 1024 && chol_update_v_enabled() && !waves4 && (lda & 1) == 0 && (base & 1) == 0 && (!ext || (ldp & 1) == 0);
}
inline void chol_launch_update(double* M, long lda, long rows, hipStream_t stream, int kcol0, int K, long base, long col_limit,
                               const double* Pext = nullptr, long ldp = 0, double* part = nullptr, const int* kfirst = nullptr, bool assign = false) {
  // trailing region: rows [base, rows), cols [base, col_limit)
  const long nrows = rows - base, nc = col_limit - base;
  if (nrows <= 0 || nc <= 0) return;
  // the big tile only when it still gives every CU a few workgroups
  const bool whole = (nc >= nrows - 1);
  const long big_tiles = ((nrows + 255) / 256) * ((nc + 127) / 128) / (whole ? 2 : 1);
  if (big_tiles >= 1024) {   // (square 128 x 128 tiles, two workgroups per CU, measured at n = 32,768: 238.7 vs 231.6 ms -- dropped)
    const long ntI = (nrows + 255) / 256, ntJ = (nc + 127) / 128;
    const long nblk = whole ? ntI * (ntI + 1) : ntI * ntJ;   // triangle: sum over ti of (2 ti + 2) tiles
    static const int waves4 = [] { const char* e = std::getenv("ESL_UPD_WAVES"); return (e && std::atoi(e) == 4) ? 1 : 0; }();
    // round 6: the vendor-shaped 128 x 128 tile, two four-wave workgroups per CU (k_chol_update_v; ESL_UPD_V=0: the 256 x 128 tile).  Same bits.
    if (chol_update_v_applies(lda, rows, base, col_limit, Pext != nullptr, ldp)) {
      const long vI = (nrows + kVT - 1) / kVT, vJ = (nc + kVT - 1) / kVT;
      const long vblk = whole ? chol_v_grid(vI, 0) : vI * vJ;
      hipLaunchKernelGGL(k_chol_update_v, dim3((unsigned)vblk), dim3(256), kCholLdsV, stream, M, lda, rows, col_limit, kcol0, K, base, (int)vJ, whole ? 0 : 1, Pext, ldp,
                         Pext ? kfirst : nullptr, 0, assign ? 1 : 0);
      return;
    }
    if (waves4)   // (experiment, round 6) four waves of 128 x 64, one per SIMD, accumulators in AGPRs
      hipLaunchKernelGGL((k_chol_update_lds<256, 128, 2, 2>), dim3((unsigned)nblk), dim3(256), kCholLdsBig, stream, M, lda, rows, col_limit, kcol0, K,
                         base, (int)ntJ, whole ? 0 : 1, Pext, ldp, (double*)nullptr, 0, Pext ? kfirst : nullptr);
    else
    hipLaunchKernelGGL((k_chol_update_lds<256, 128>), dim3((unsigned)nblk), dim3(512), kCholLdsBig, stream, M, lda, rows, col_limit, kcol0, K,
                       base, (int)ntJ, whole ? 0 : 1, Pext, ldp, (double*)nullptr, 0, Pext ? kfirst : nullptr);
  } else {
    const long ntI = (nrows + 127) / 128, ntJ = (nc + 63) / 64;
    const long nblk = whole ? ntI * (ntI + 1) : ntI * ntJ;
    int nsplit = 1, kper = K;
    if (part && Pext && base == 0 && nblk < 128 && K >= 16 * kKC) {
      nsplit = (int)std::min<long>(kCholMaxSplit, std::max<long>(1, 512 / nblk));
      kper = ((K + nsplit - 1) / nsplit + kKC - 1) / kKC * kKC;
      nsplit = (K + kper - 1) / kper;
    }
    if (nsplit > 1) {
      hipLaunchKernelGGL((k_chol_update_lds<128, 64>), dim3((unsigned)nblk, (unsigned)nsplit), dim3(512), kCholLdsSmall, stream, M, lda, rows, col_limit,
                         kcol0, K, base, (int)ntJ, whole ? 0 : 1, Pext, ldp, part, kper);
      hipLaunchKernelGGL(k_chol_splitk_reduce, dim3((unsigned)((rows * col_limit + 255) / 256)), dim3(256), 0, stream, M, lda, rows, col_limit, part, nsplit);
    } else {
      hipLaunchKernelGGL((k_chol_update_lds<128, 64>), dim3((unsigned)nblk), dim3(512), kCholLdsSmall, stream, M, lda, rows, col_limit, kcol0, K,
                         base, (int)ntJ, whole ? 0 : 1, Pext, ldp, (double*)nullptr, 0, Pext ? kfirst : nullptr);
    }
  }
}

// ---- persistent factorisation (round 4) -----------------------------------------------------------------------------------------
// The launch-per-step form above leaves the chain of diagonal blocks exposed wherever the trailing matrix is too small to hide it
// (the last ~8,000 rows of an order-18,000 system, ~8 of its 44 ms) and its small kernels queue behind 110 us update tiles for a
// free CU.  Here ONE kernel, k_chol_persist, runs the whole factorisation: 256 workgroups of 512 threads, every one the only
// tenant of its CU (157 KB of LDS).  Workgroup 0 is the CHAIN: it walks the diagonal blocks (the potrf2 body).  The other 255 pull
// TASKS from a static list with one atomic counter.  Tasks, on the absolute tile grid (row tiles of 256, column tiles = the
// 128-wide panels):
//   S(k, i)     strip i (64 rows) of the panel solve of panel k                            (k_chol_panel's body)
//   u(k, R, J)  tile (R, J) -= X(R, k) X(J, k)^T, rank 128, J in the rest of k's outer panel (k_chol_update_lds's body); the
//               DIAGONAL tile of a panel takes these in quarters (task type 3, below)
//   U(o, R, J)  tile (R, J) -= X(R, o) X(J, o)^T, rank W x 128, J beyond outer panel o
// Dependencies are words in device memory (zeroed by a memset in front of the launch): pdone[k] (chain: L_kk, Linv_k published),
// sdone[k][R] (strips of row tile R solved for panel k; complete at ns[k][R]), ver[R][J] (updates applied to tile (R, J): every
// update carries its sequence number, waits for ver == seq and leaves ver = seq + 1 -- read-modify-write order and "tile final"
// in one word).  The list is in an order in which every task's prerequisites come earlier (or are the chain's), so a workgroup
// that spins on a word waits for work another RESIDENT workgroup already holds: no deadlock whatever the dispatch order; the
// next outer panel's chain-dependent tasks sit between batches of the previous outer panel's far updates (look-ahead without
// streams or events).  tests/test_chol_plan.py replays the list on the CPU.  Hand-offs follow the guide's recipe R1: payload
// stored write-through (sc1), every storing wave drains, one lane publishes the word; the consumer polls from one lane, ONE
// agent-scope acquire, plain loads.  Every spin is bounded (bit 1 of info + an abort word that stops all other spins).
// What the first versions taught (all measured on MI355X, round 4):
//   * TWO kernels (chain / workers) on two streams are co-resident only until something else touches the queues: a stream created
//     or destroyed anywhere in the process while they run gets them TIME-SLICED against each other, and a spinning consumer whose
//     producer is switched out turns a 40 ms factorisation into seconds.  One kernel cannot be split that way.
//   * a polled word must be read with a WRITING atomic (chol_peek: an atomic add of zero).  Agent-scope loads, fetch_or(0) (the
//     compiler folds it into a load) and failing compare-and-swaps are all served from the polling XCD's L2, which another XCD's
//     write-through store does not update: once every workgroup of an XCD waits, nothing evicts the line and the stale value is
//     read forever.  Pollers back to back saturate the atomic units (255 of them slowed the whole kernel 100x): workers sleep
//     ~2.7 us between polls, the chain ~0.1 us; a plain agent-scope load goes first (the words only grow: stale can under-report).
//   * where the time goes at n = 18,000 (ESL_CHOL_TIMING=1 prints it): workers spend 35.5 of 40 ms inside task bodies (the
//     rank-512 tiles run at ~73 % of a CU's MFMA peak) and 3.5 ms waiting; the chain is idle 31 ms -- and sets the pace over the
//     last three eighths of the panels, where a step costs potrf 62 us + 40 us (strips under the diagonal block -> the quarters of
//     the next diagonal tile -> the next block) against 3 - 70 us of trailing work.  Two schedule changes got it there from
//     62 + 68 us: the diagonal tile's rank-128 updates in four quarters (42 us for one workgroup), and the first diagonal tile of
//     the NEXT outer panel fed rank-128 updates panel by panel instead of waiting for its rank-512 tile (chol_tile_special).
//   * the chain then stopped waiting for other workgroups between two blocks: it solves the 128 rows under its block against the
//     inverse that is still in its LDS and applies the rank-128 update to the next diagonal block itself (chol_chain_solve_rows /
//     chol_chain_update_next; strips 0, 1 and the diagonal half of the quarters leave the task list; ESL_CHOL_FUSE=0 is the
//     form before).  Its own 25 us (14 + 10 + drain) replace 40 us of hand-offs: the step is 66 + 36 us, n = 18,000 40.5 -> 40.0 ms.
//     The chain's own arithmetic is now what the last eighths of the panels cost; the next lever is the 66 us of potrf2.
// 0: S(panel a, strip b)   1: u(panel a, row tile b, column tile c)   2: U(outer panel a, row tile b, column tile c)
// 3: a QUARTER of u(panel a, row tile b, column tile c & 0xffff): rows 256 b + 128 h, columns 128 J + 64 g with h = bit 16, g = bit 17
//    of c, bits 18-20 = the number of quarters of THIS update that are tasks, bits 21-23 = the number a whole update of the tile has
//    (those that meet the lower triangle and the matrix; fewer are tasks when the chain takes the diagonal block itself).
//    The DIAGONAL tile of the next panels takes its rank-128 updates in quarters: it sits between the solve of panel k and the
//    diagonal block of panel k + 1, i.e. on the critical path of the chain, where one workgroup needs 42 us for the whole tile and
//    four need ~12 us for a quarter each (round 4: 62 panels x 30 us of the last 8,000 columns of an order-18,000 system).
struct CholTask { int type, a, b, c; };
struct CholPlan {
  int n = 0, np = 0, W = 0, n_outer = 0, nR = 0;
  int fuse = 0;          // the chain solves the two strips under its diagonal block and updates the next diagonal block itself (below)
  int merge = 1;         // far updates of a tile taken `merge` outer panels at a time (round 6: one visit with K = merge x W x 128)
  int strip_merge = 0;   // the strips of a row tile away from the diagonal as one task (round 6)
  int sp = 1;            // leading column panels of an outer panel whose chain tiles are fed rank-128 updates panel by panel (chol_tile_special)
  std::vector<CholTask> tasks;
  std::vector<int> ns;   // [np][nR]
};
// Updates a tile (R, J) of the lower triangle receives before panel J is factored, in this order: nU rank-(W x 128) updates U(o), one
// per EARLIER outer panel, then rank-128 updates from the panels in front of J.  Ordinarily those are the J % W panels of J's own
// outer panel.  The DIAGONAL tile of an outer panel's FIRST column panel (J % W == 0, R == J / 2) is special: it takes the
// previous outer panel's contribution as W rank-128 updates (in quarters), one per panel as that panel is solved, instead of
// inside U(o - 1) -- otherwise the chain's step across an outer-panel boundary waits for a whole rank-512 tile update (~150 us)
// that cannot start before the LAST panel of the outer panel is solved.
// Round 6: SP leading column panels of every outer panel are special in this sense (default 2), in the diagonal tile's row AND the row tile
// under it (R == J / 2 + 1: its strips of panel J are what the next panels' chain tiles wait for).  Measured with SP = 1, R == J / 2
// only, by the panel's position in its outer panel over the chain-bound part: gap to the next block 60 / 36 / 27 / 27 us at n = 8,192
// against 26 us of the chain's own work -- at the first panel of an outer panel the fused stage waited ~34 us for the rank-(W x 128)
// update of the NEXT diagonal tile (ke / 2, ke + 1), which cannot start before the outer panel's last solve.  Making only that tile
// special moved the stall to the next step (27 / 68 / 27 / 27): there the chain waits for tile (ke / 2 + 1, ke + 1), whose update by panel
// ke needs panel ke's strips in row tile ke / 2 + 1, which wait for tile (ke / 2 + 1, ke) to be final -- another 150 us rank-512 update.
// A special tile receives, in this order: the rank-(W x 128) updates of outer panels 0 .. J / W - 2, then W rank-128 updates from the
// panels of outer panel J / W - 1, then J % W rank-128 updates from the earlier panels of its own outer panel.
__host__ __device__ __forceinline__ bool chol_tile_special(int R, int J, int W, int SP) { return J >= W && (J % W) < SP && (R == J / 2 || R == J / 2 + 1); }
__host__ __device__ __forceinline__ int chol_tile_nU(int R, int J, int W, int SP) { return J / W - (chol_tile_special(R, J, W, SP) ? 1 : 0); }
// index, among the rank-128 updates tile (R, J) receives, of the one from panel a (a in J's outer panel or, special tiles, the one before)
__host__ __device__ __forceinline__ int chol_tile_ridx(int R, int J, int W, int SP, int a) { return ((a / W == J / W && chol_tile_special(R, J, W, SP)) ? W : 0) + a % W; }
// sequence number of that update: the value of ver[R][J] it waits for (and leaves at + 1)
__host__ __device__ __forceinline__ int chol_tile_seq(int R, int J, int W, int SP, int a) { return chol_tile_nU(R, J, W, SP) + chol_tile_ridx(R, J, W, SP, a); }
__host__ __device__ __forceinline__ int chol_tile_final(int R, int J, int W, int SP) { return chol_tile_nU(R, J, W, SP) + (chol_tile_special(R, J, W, SP) ? W : 0) + J % W; }
// quarter (h, g) of tile (R, J): rows 256 R + 128 h .. + 128, columns 128 J + 64 g .. + 64
__host__ __device__ inline bool chol_quarter_live(int R, int J, int h, int g, long rows, int n) {
  const long i0 = 256L * R + 128L * h, j0 = 128L * J + 64L * g;
  return i0 < rows && j0 < n && i0 + 127 >= j0;
}
inline void chol_plan_build(int n, int W, int filler, CholPlan& pl, bool fuse = false, int merge = 1, bool strip_merge = false, int sp = 1) {
  const long rows = (long)n + 1;
  pl.fuse = fuse ? 1 : 0;
  pl.merge = merge < 1 ? 1 : merge;
  pl.strip_merge = strip_merge ? 1 : 0;
  pl.sp = sp < 1 ? 1 : (sp > W ? W : sp);
  const int SP = pl.sp;
  pl.n = n; pl.W = W; pl.np = (n + kNB - 1) / kNB; pl.n_outer = (pl.np + W - 1) / W; pl.nR = (int)((rows + 255) / 256);
  const int np = pl.np, nR = pl.nR;
  pl.ns.assign((size_t)np * nR, 0);
  pl.tasks.clear();
  auto live = [&](int R, int J) { return 256L * R + 255 >= 128L * J && 256L * R < rows; };   // the tile meets the lower triangle (or the b row)
  auto k0s = [&](int k) { const long k0 = 128L * k; return k0 + std::min<long>(kNB, n - k0); };   // first row under panel k's diagonal block
  auto strips = [&](int k) { const long k0 = 128L * k, nb = std::min<long>(kNB, n - k0), below = rows - (k0 + nb); return (int)(below > 0 ? (below + 63) / 64 : 0); };
  for (int k = 0; k < np; ++k) {
    const long k0 = 128L * k, nb = std::min<long>(kNB, n - k0);
    for (int i = 0; i < strips(k); ++i) pl.ns[(size_t)k * nR + (size_t)((k0 + nb + 64L * i) / 256)]++;
  }
  std::vector<CholTask> prevA, prevB;   // the previous outer panel's rank-(W x 128) updates: this outer panel's columns / everything beyond
  for (int o = 0; o < pl.n_outer; ++o) {
    const int kb = o * W, ke = std::min(np, kb + W);
    pl.tasks.insert(pl.tasks.end(), prevA.begin(), prevA.end());
    size_t bpos = 0;
    auto fill = [&]() { const size_t e = std::min(prevB.size(), bpos + (size_t)filler); pl.tasks.insert(pl.tasks.end(), prevB.begin() + bpos, prevB.begin() + e); bpos = e; };
    for (int k = kb; k < ke; ++k) {
      fill();                                              // (work for the others while the chain factors block k)
      // (fused: strips 0, 1 are the chain's.)  Round 6: the strips of ONE row tile are one task where nobody is waiting for the first
      // of them alone -- every consumer of a solved row tile waits for all of its strips (sdone == ns) anyway, and a strip is ~9 us of
      // work behind ~20 us of task hand-off.  The row tiles next to the diagonal block keep their 64-row strips: they feed the quarters
      // of the next diagonal tiles, i.e. the chain, and four workgroups finish them sooner than one.  c = strips in the task.
      for (int i = fuse ? 2 : 0; i < strips(k);) {
        const long R = (k0s(k) + 64L * i) / 256;
        int cnt = 1;
        if (pl.strip_merge && R >= k0s(k) / 256 + 2)
          while (i + cnt < strips(k) && (k0s(k) + 64L * (i + cnt)) / 256 == R) ++cnt;
        pl.tasks.push_back(CholTask{0, k, i, cnt});
        i += cnt;
      }
      bool any = false;
      auto quarters = [&](int R, int J) {
        // (q & 1 = h: which 128 rows, q >> 1 = g: which 64 columns.)  Fused chain: panel J - 1's update of the diagonal BLOCK of
        // tile (J / 2, J) -- the half h = J & 1 -- is the chain's; the workers keep the other half, if it is below the diagonal
        const bool chains = fuse && k == J - 1 && R == J / 2;
        int full = 0, mine = 0;
        for (int q = 0; q < 4; ++q)
          if (chol_quarter_live(R, J, q & 1, q >> 1, rows, n)) { ++full; if (!(chains && (q & 1) == (J & 1))) ++mine; }
        for (int q = 0; q < 4; ++q)
          if (chol_quarter_live(R, J, q & 1, q >> 1, rows, n) && !(chains && (q & 1) == (J & 1)))
            pl.tasks.push_back(CholTask{3, k, R, J | ((q & 1) << 16) | ((q >> 1) << 17) | (mine << 18) | (full << 21)});
      };
      // the next outer panel's first diagonal tile takes this panel's rank-128 contribution now (chol_tile_special); of all of the
      // panel's updates it is the one the chain will wait for soonest when k is the outer panel's last, so it goes first
      for (int J = ke; J < std::min(np, ke + W); ++J)
        for (int R = J / 2; R <= J / 2 + 1; ++R)
          if (chol_tile_special(R, J, W, SP) && live(R, J)) { if (!any) { fill(); any = true; } quarters(R, J); }
      for (int J = k + 1; J < ke; ++J)
        for (int R = J / 2; R < nR; ++R)
          if (live(R, J)) {
            if (!any) { fill(); any = true; }
            // quarters for the diagonal tile AND, when panel J's block is the lower half of its tile (J odd), for the tile under it:
            // its upper half holds the rows under block J, which the fused chain solves next -- as one task it is 42 us in the
            // chain's way on every other step
            if (R != J / 2 && R != (J + 1) / 2) { pl.tasks.push_back(CholTask{1, k, R, J}); continue; }
            quarters(R, J);
          }
    }
    pl.tasks.insert(pl.tasks.end(), prevB.begin() + bpos, prevB.end());
    prevA.clear(); prevB.clear();
    const int ne = std::min(np, ke + W);
    // Round 6: a tile's FAR updates -- those of outer panels o <= J / W - 2, which only fill the workers' time -- are taken `merge` outer
    // panels at a time: one visit of the tile with K = merge x W x 128 instead of `merge` visits (each visit pays ~22 us of task
    // hand-off, first-chunk staging and the read-modify-write of C beside ~127 us of K loop).  A group [g0, g0 + cnt) goes into the
    // list when its LAST outer panel is solved; the update of the outer panel right in front of the tile's own (o = J / W - 1, the one
    // the chain waits for) stays a visit of its own.  c = J | cnt << 16; sequence number g0 -> g0 + cnt.
    for (int J = ke; J < np; ++J)
      for (int R = J / 2; R < nR; ++R)
        if (live(R, J) && !(J < ne && chol_tile_special(R, J, W, SP))) {   // (special tiles took this outer panel's contribution panel by panel)
          if (J < ne) { prevA.push_back(CholTask{2, o, R, J | (1 << 16)}); continue; }
          const int far = J / W - 1;                     // far updates of this tile: o = 0 .. far - 1 (this o is one of them: J >= ne)
          // the groups of different tiles are STAGGERED (boundaries shifted by a hash of the tile): with one grid of boundaries for all
          // tiles every far task of `merge` outer panels enters the list at once and the panels in between have no filler work at all
          // (measured: waiting 2.7 -> 3.8 ms while the bodies fell 36.5 -> 34.1)
          const int shift = (R + 2 * J) % pl.merge, q = (o + shift) / pl.merge;
          const int g0 = std::max(0, q * pl.merge - shift), ge = std::min((q + 1) * pl.merge - shift, far) - 1;
          if (o == ge) prevB.push_back(CholTask{2, g0, R, J | ((ge - g0 + 1) << 16)});
        }
  }
}
// sync words: [0] task counter, [1] abort, [2] arrival tickets (0 = the chain), [4 ..) pdone[np], sdone[np][nR], ver[nR][np], quarters done [nR][np]
inline size_t chol_sync_words(int np, int nR) { return 4 + (size_t)np + 3 * (size_t)np * nR; }
__device__ long long g_chol_timeout_ticks = 300000000LL;   // 3 s at 100 MHz (ESL_CHOL_TIMEOUT_MS overrides it: debugging)
// Spin until *word >= want (ONE lane; bounded).  Every poll is a chol_peek -- a device-scope atomic -- so the pollers are RATE
// LIMITED: 255 workgroups polling back to back saturate the atomic units (~90 atomics per us on one word) and every other atomic
// of the launch -- the chain's own polls, the task counter, the publishes -- queues behind them: measured, the first 30 panels of an
// order-8,192 system then take 300 ms instead of 4.  SLEEP = argument of s_sleep (64 cycles each): workers wait ~2.7 us between
// polls (at most ~95 polls per us from all of them together), the chain -- one poller -- ~0.1 us.
template <int SLEEP>
__device__ __forceinline__ bool chol_wait_ge(int* word, int want, int* abortw, int* info) {
  // fast path: a plain agent-scope load.  The words only ever grow, so a stale copy can under-report but never over-report; most
  // waits of tasks deep in the list were satisfied long ago and this XCD has either never fetched the line or fetched it late enough
  if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
  if (chol_peek(word) >= want) return true;
  const long long t0 = (long long)wall_clock64();
  for (unsigned spins = 1;; ++spins) {
    __builtin_amdgcn_s_sleep(SLEEP);
    if (chol_peek(word) >= want) return true;
    if ((spins & 15u) == 0) {
      // (abort: a plain agent-scope load is enough -- a poller that keeps seeing a stale 0 runs into its own timeout)
      if (__hip_atomic_load(abortw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
      if ((long long)wall_clock64() - t0 > g_chol_timeout_ticks) {
        atomicOr(info, 2);
        __hip_atomic_store(abortw, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
  }
}

// ---- the chain's own step to the next diagonal block (round 4, "fused chain") -------------------------------------------------
// After chol_potrf2_body the LDS array L holds Linv_k.  The next diagonal block waits for two things that other workgroups used to
// deliver: the solve of the 128 rows under block k (strips 0, 1 of panel k) and the rank-128 update of block k + 1 with them --
// 40 us of task bodies, drains and polls per panel against ~8 us of arithmetic.  Here the chain does both itself:
//   stage B   X = A[r0 .. r0 + 128, panel k] Linv_k^T   (rows < `rows`: the right-hand side's row rides along), Linv from LDS, A from
//             global memory; X goes to global memory (it IS L's rows) and, once every wave has finished with Linv, into the LDS array
//   stage C   A[r0 + i, c1 + j] -= sum_c X(i, c) X(j, c),  i >= j: the lower triangle of the next diagonal block (and the
//             right-hand side's row when it lies in these rows), X from LDS
// Both products are formed transposed (lane & 15 = the matrix ROW), so every global access is 128 contiguous bytes per 16 lanes.
// 512 threads = 8 waves.  The caller provides the waits (stage B: the tile of these rows final for panel k; stage C: every earlier
// update of the next diagonal tile applied) and the publish afterwards.
template <bool WT>
__device__ __forceinline__ void chol_chain_solve_rows(double* __restrict__ sm, double* __restrict__ M, long lda, long rows, int k0) {
  double* L = sm;
#define LL(i, j) L[(i) + (j) * kLdsPad]
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, r = lane & 15, kq = lane >> 4;
  const long r0 = (long)k0 + kNB;
  const long row = r0 + 16 * wave + r;
  const bool rv = row < rows;
  const double* ap = M + (rv ? row : rows - 1) + (long)(k0 + kq) * lda;
  double a[32];
#pragma unroll
  for (int s2 = 0; s2 < 32; ++s2) a[s2] = ap[(long)(4 * s2) * lda];
  double4_t acc[8];
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) acc[jb] = double4_t{0, 0, 0, 0};
#pragma unroll
  for (int s2 = 0; s2 < 32; ++s2) {
    const double av = rv ? a[s2] : 0.0;
    const int k = 4 * s2 + kq;
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
      if (4 * s2 < 16 * jb + 16) {   // Linv(c, k) = 0 for k > c (compile-time after unrolling)
        const int c = 16 * jb + r;
        const double lv = (k <= c) ? LL(c, k) : 0.0;
        acc[jb] = __builtin_amdgcn_mfma_f64_16x16x4f64(lv, av, acc[jb], 0, 0, 0);   // D[c = 4 g + kq][row = r]
      }
  }
  // X -> global memory: lane r = row, 16 lanes store 128 contiguous bytes of a column
  if constexpr (WT) {
#pragma unroll
    for (int jb = 0; jb < 8; jb += 2)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        chol_store_wt_pair(&M[row + (long)(k0 + 16 * jb + 4 * g + kq) * lda], &M[row + (long)(k0 + 16 * (jb + 1) + 4 * g + kq) * lda], acc[jb][g], acc[jb + 1][g], rv,
                           ((lda | (r0 + 16 * wave)) & 1) == 0);
  } else if (rv) {
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
      for (int g = 0; g < 4; ++g) chol_store<WT>(&M[row + (long)(k0 + 16 * jb + 4 * g + kq) * lda], acc[jb][g]);
  }
  __syncthreads();   // every wave has read Linv: the array now holds X (rows past the matrix as zeros)
#pragma unroll
  for (int jb = 0; jb < 8; ++jb)
#pragma unroll
    for (int g = 0; g < 4; ++g) LL(16 * wave + r, 16 * jb + 4 * g + kq) = rv ? acc[jb][g] : 0.0;
  __syncthreads();
#undef LL
}
// tiles (ib >= jb) of the next diagonal block owned by wave w: w, w + 8, ... of the 36, at most 5
__device__ __forceinline__ void chol_chain_tile(int tl, int& ib, int& jb) {
  ib = 0; int off = 0;
  while (off + ib + 1 <= tl) { off += ib + 1; ++ib; }
  jb = tl - off;
}
// the current values of the wave's entries of the next diagonal block, requested BEFORE the solve of the rows (every earlier update
// of that tile has been waited for by then): fetched one tile at a time in front of its own products, each was ~2 us of exposed
// latency, five in a row
struct CholChainC { double v[5][4]; };
__device__ __forceinline__ void chol_chain_prefetch_next(const double* __restrict__ M, long lda, long rows, int n, int k0, CholChainC& c) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, kq = lane >> 4;
  const long c1 = (long)k0 + kNB;
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const int tl = wave + 8 * q;
    int ib, jb;
    chol_chain_tile(tl < 36 ? tl : 35, ib, jb);
    const long row = c1 + 16 * ib + r;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const long col = c1 + 16 * jb + 4 * g + kq;
      c.v[q][g] = M[(row < rows ? row : rows - 1) + (col < n ? col : (long)n - 1) * lda];
    }
  }
}
template <bool WT>
__device__ __forceinline__ void chol_chain_update_next(double* __restrict__ sm, double* __restrict__ M, long lda, long rows, int n, int k0,
                                                       const CholChainC& c) {
  double* L = sm;
#define LL(i, j) L[(i) + (j) * kLdsPad]
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, r = lane & 15, kq = lane >> 4;
  const long c1 = (long)k0 + kNB;   // first row AND first column of the next diagonal block
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const int tl = wave + 8 * q;
    if (tl >= 36) break;
    int ib, jb;
    chol_chain_tile(tl, ib, jb);
    double4_t acc = {0, 0, 0, 0};
#pragma unroll 8
    for (int kk = 0; kk < kNB; kk += 4) {
      const double xj = LL(16 * jb + r, kk + kq), xi = LL(16 * ib + r, kk + kq);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xj, xi, acc, 0, 0, 0);   // D[j = 4 g + kq][i = r]
    }
    const long row = c1 + 16 * ib + r;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const long col = c1 + 16 * jb + 4 * g + kq;
      if (row < rows && col < n && row >= col) chol_store<WT>(&M[row + col * lda], c.v[q][g] - acc[g]);
    }
  }
#undef LL
}
constexpr size_t kP2Lds = (size_t)(kNB * kLdsPad + 96 * kSB + kNB) * sizeof(double);
static_assert(kP2Lds >= kCholLdsBig, "the persistent kernel's LDS is sized by the diagonal-block role");
constexpr int kPwThreads = 512, kPwGrid = 256;
// diagnostics of the persistent kernel (ESL_CHOL_TIMING=1 in the self test): per workgroup {ticks waiting on dependency words, ticks
// inside task bodies, tasks, first tick, last tick} of the last launch, wall_clock64 ticks (100 MHz)
__device__ long long g_chol_stats[kPwGrid * 5];
__device__ long long g_chol_chain_log[2 * 1024];
__device__ long long g_chol_fuse_ticks[4];   // fused chain stage, totals of the last launch: waits, row solve, next-block update, drain + publish   // per diagonal block: tick its tile was final, tick its factor was published
constexpr size_t kPwLds = kP2Lds + 64 + 256;   // + the task slot words and the diagnostics accumulators + the two halves' barrier counters (slot[16], slot[48])
static_assert(2 * kCholLdsV <= kP2Lds, "two halves' staging buffers");
// ONE kernel, two roles (round 4, second form): workgroup 0 is the chain, workgroups 1..255 the workers.  (The first form ran the two
// roles as two kernels on two streams: correct and as fast -- but whether two queues of one process run side by side or in turns
// is the scheduler's business: creating or destroying any stream while the pair ran made it time-slice them, each role then only
// moved during its own quantum, and an order-18,000 factorisation went from 42 ms to its 3 s spin limit.)  The 157 KB of LDS the
// diagonal-block role needs make every workgroup the only tenant of its CU -- which the 104 KB of the update role did anyway.
static __global__ __launch_bounds__(kPwThreads) void k_chol_persist(double* __restrict__ M, long lda, int n, int np, int W, int nR,
                                                                   double* __restrict__ Linv_ws, const CholTask* __restrict__ tasks,
                                                                   int n_tasks, const int* __restrict__ ns, int* sync, int* info, int fuse, int SP) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  int* slot = reinterpret_cast<int*>(sm + kP2Lds / sizeof(double));   // (behind the roles' LDS: no static __shared__, guide G17)
  long long* lstat = reinterpret_cast<long long*>(slot + 4);           // diagnostics accumulate in LDS, not in registers (see below)
  const bool stats = g_chol_stats_on != 0;
  if (stats && threadIdx.x == 0) { lstat[0] = lstat[1] = lstat[2] = 0; lstat[3] = (long long)wall_clock64(); if (blockIdx.x == 0) { g_chol_fuse_ticks[0] = g_chol_fuse_ticks[1] = g_chol_fuse_ticks[2] = g_chol_fuse_ticks[3] = 0; } }
  int* pdone = sync + 4;
  int* sdone = pdone + np;
  int* ver = sdone + (size_t)np * nR;
  int* qdone = ver + (size_t)np * nR;
  const int t = threadIdx.x;
  const long rows = (long)n + 1;
  // the chain is whichever workgroup gets here FIRST (a ticket in sync[2]), not blockIdx 0: the workers' waits end only if the chain
  // is resident, and nothing guarantees that the dispatcher starts with block 0 (guide G16: no dispatch-order assumption)
  if (t == 0) { slot[2] = __hip_atomic_fetch_add(sync + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); slot[14] = 0; slot[15] = 0; }
  __syncthreads();
  const bool is_chain = slot[2] == 0;
  if (is_chain) {   // ---- the chain: diagonal blocks in order
    bool have_next = false;   // fused: block k already carries every update (the previous step applied the last one itself)
    for (int k = 0; k < np; ++k) {
      const int k0 = k * kNB, nb = (n - k0 < kNB) ? (n - k0) : kNB;
      if (t == 0) {
        const long long tw = stats ? (long long)wall_clock64() : 0;
        const bool ok = have_next || chol_wait_ge<2>(&ver[(size_t)(k / 2) * np + k], chol_tile_final(k / 2, k, W, SP), sync + 1, info);   // every update of the diagonal block's tile is in
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        slot[0] = ok ? 1 : 0;
        if (stats) { const long long now = (long long)wall_clock64(); lstat[0] += now - tw; lstat[4] = now; if (k < 1024) g_chol_chain_log[2 * k] = now; }
      }
      __syncthreads();
      if (!slot[0]) return;
      chol_potrf2_body<true, kPwThreads>(sm, M, lda, k0, nb, Linv_ws + (size_t)k * kNB * kNB, info);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores ...
      __syncthreads();
      if (t == 0) {
        __hip_atomic_store(&pdone[k], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... then ONE lane publishes
        if (stats) { const long long now = (long long)wall_clock64(); lstat[1] += now - lstat[4]; lstat[2] += 1; if (k < 1024) g_chol_chain_log[2 * k + 1] = now; }
      }
      have_next = false;
      const long below = rows - ((long)k0 + nb);
      if (fuse && below > 0) {
        // the rows under the block (strips 0, 1 of panel k: a full panel -- only the last one is short, and it has just the
        // right-hand side's row under it, which the same code handles) and the next diagonal block: chol_chain_solve_rows / _update_next
        const int R1 = (int)(((long)k0 + nb) / 256);
        const int nstr = below > 64 ? 2 : 1;
        const bool next = k + 1 < np;
        if (t == 0) {
          const long long tw = stats ? (long long)wall_clock64() : 0;
          bool ok = chol_wait_ge<2>(&ver[(size_t)R1 * np + k], chol_tile_final(R1, k, W, SP), sync + 1, info);
          if (ok && next) ok = chol_wait_ge<2>(&ver[(size_t)((k + 1) / 2) * np + k + 1], chol_tile_seq((k + 1) / 2, k + 1, W, SP, k), sync + 1, info);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          slot[0] = ok ? 1 : 0;
          if (stats) { const long long now = (long long)wall_clock64(); lstat[0] += now - tw; g_chol_fuse_ticks[0] += now - tw; lstat[4] = now; }
        }
        __syncthreads();
        if (!slot[0]) return;
        CholChainC cnext;
        if (next) chol_chain_prefetch_next(M, lda, rows, n, k0, cnext);
        if (nb == kNB) chol_chain_solve_rows<true>(sm, M, lda, rows, k0);
        else {   // the short last panel: one row (the right-hand side's) against a partial inverse -- the strip body
          chol_panel_body<true>(M, lda, rows, k0, nb, Linv_ws + (size_t)k * kNB * kNB, 0L);
        }
        if (stats && t == 0) { const long long now = (long long)wall_clock64(); g_chol_fuse_ticks[1] += now - lstat[4]; lstat[4] = now; }
        if (next) chol_chain_update_next<true>(sm, M, lda, rows, n, k0, cnext);
        if (stats && t == 0) { const long long now = (long long)wall_clock64(); g_chol_fuse_ticks[2] += now - lstat[4]; lstat[4] = now; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) {
          if (stats) { const long long now = (long long)wall_clock64(); g_chol_fuse_ticks[3] += now - lstat[4]; }
          __hip_atomic_fetch_add(&sdone[(size_t)k * nR + R1], nstr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (next) {
            // panel k's update of tile ((k + 1) / 2, k + 1): the workers' quarters (the half that is not the diagonal block) move the
            // sequence number on when they are done; when there are none (k + 1 odd: the other half lies above the diagonal) the chain does
            const int J = k + 1, R = J / 2;
            int mine = 0;
            for (int q = 0; q < 4; ++q) if ((q & 1) != (J & 1) && chol_quarter_live(R, J, q & 1, q >> 1, rows, n)) ++mine;
            if (mine == 0) __hip_atomic_store(&ver[(size_t)R * np + J], chol_tile_seq(R, J, W, SP, k) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        have_next = next;
      }
    }
    if (stats && t == 0) { for (int q = 0; q < 4; ++q) g_chol_stats[q] = lstat[q]; g_chol_stats[4] = (long long)wall_clock64(); }
    return;
  }
  for (;;) {               // ---- a worker: the next task of the list
    // (nothing but the kernel arguments is live across a task body: the task is re-read from the list afterwards -- with the
    //  descriptor, the publish address and the diagnostics kept in registers the update tile's 212 spilled 60 B per lane)
    if (t == 0) {
      if (slot[14] > 0) { slot[14] -= 1; slot[15] += 1; }   // the next strip of the task in hand: no new task, no waits, no publish in between
      else { slot[0] = __hip_atomic_fetch_add(sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); slot[15] = 0; }
    }
    __syncthreads();
    if (slot[0] >= n_tasks) {
      if (stats && t == 0) { const int me = slot[2] < kPwGrid ? slot[2] : kPwGrid - 1; for (int q = 0; q < 4; ++q) g_chol_stats[5 * me + q] = lstat[q]; g_chol_stats[5 * me + 4] = (long long)wall_clock64(); }
      return;
    }
    if (t == 0 && slot[15] == 0) {
      const CholTask tk = tasks[slot[0]];
      bool ok = true;
      const long long tw = stats ? (long long)wall_clock64() : 0;
      slot[14] = (tk.type == 0 && tk.c > 1) ? tk.c - 1 : 0;   // strips of this task after the first
      if (tk.type == 0) {
        const int k = tk.a, k0 = k * kNB, nb = (n - k0 < kNB) ? (n - k0) : kNB;
        const int R = (int)(((long)k0 + nb + 64L * tk.b) / 256);
        ok = chol_wait_ge<100>(&pdone[k], 1, sync + 1, info) && chol_wait_ge<100>(&ver[(size_t)R * np + k], chol_tile_final(R, k, W, SP), sync + 1, info);
      } else {
        const int R = tk.b, J = tk.c & 0xffff;
        const int cnt = (tk.type == 2) ? (tk.c >> 16) : 1;   // type 2: outer panels [a, a + cnt) in one visit
        const int ke = (tk.type != 2) ? tk.a + 1 : (((tk.a + cnt) * W < np) ? (tk.a + cnt) * W : np);
        // the LAST panel's strips of a row tile are solved only after every earlier panel of the same outer panel has solved its own
        // there and updated them (S waits for its tile to be final): one pair of words stands for all W panels
        // the three words are fetched TOGETHER first (agent-scope loads go to the fabric, ~1.5 us each one after the other; most
        // tasks find all three satisfied) and only the unsatisfied ones enter the polling wait
        int* w0 = &sdone[(size_t)(ke - 1) * nR + R];
        int* w1 = &sdone[(size_t)(ke - 1) * nR + J / 2];
        int* w2 = &ver[(size_t)R * np + J];
        const int want0 = ns[(size_t)(ke - 1) * nR + R], want1 = ns[(size_t)(ke - 1) * nR + J / 2];
        const int seq = (tk.type != 2) ? chol_tile_seq(R, J, W, SP, tk.a) : tk.a;
        const int v0 = __hip_atomic_load(w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), v1 = __hip_atomic_load(w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                  v2 = __hip_atomic_load(w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = (v0 >= want0 || chol_wait_ge<100>(w0, want0, sync + 1, info)) && (v1 >= want1 || chol_wait_ge<100>(w1, want1, sync + 1, info)) &&
             (v2 >= seq || chol_wait_ge<100>(w2, seq, sync + 1, info));
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      slot[1] = ok ? 1 : 0;
      slot[16] = 0; slot[48] = 0;   // the halves' barrier counters (chol_update_tile_v<.., 1>): every wave of the previous task is behind its last barrier
      if (stats) { const long long now = (long long)wall_clock64(); lstat[0] += now - tw; lstat[4] = now; }
    }
    __syncthreads();
    if (!slot[1]) return;
    {
      const CholTask tk = tasks[slot[0]];
      if (tk.type == 0) {
        const int k = tk.a, k0 = k * kNB, nb = (n - k0 < kNB) ? (n - k0) : kNB;
        // (a task of several strips -- tk.c, one row tile's -- comes back here once per strip without leaving the task: slot[15] = the strip
        //  it is at.  A loop around this body cost the merged kernel 540 B of scratch per lane, K loops included: 37.7 -> 46.7 ms.)
        chol_panel_body<true>(M, lda, rows, k0, nb, Linv_ws + (size_t)k * kNB * kNB, (long)(tk.b + slot[15]));
      } else {
        const int cnt = (tk.type == 2) ? (tk.c >> 16) : 1;
        const int kb = (tk.type != 2) ? tk.a : tk.a * W, ke = (tk.type != 2) ? tk.a + 1 : (((tk.a + cnt) * W < np) ? (tk.a + cnt) * W : np);
        const long c0 = (long)kb * kNB;
        const int K = (int)(((long)ke * kNB < n ? (long)ke * kNB : (long)n) - c0);
        if (tk.type == 3)
          chol_update_tile<128, 64, 4, 2, true>(sm, M, lda, rows, (long)n, M + c0 * lda, lda, K, 256L * tk.b + 128L * ((tk.c >> 16) & 1),
                                                128L * (tk.c & 0xffff) + 64L * ((tk.c >> 17) & 1), false);
        else {
#ifdef ESL_PERSIST_TILE_LDS
          chol_update_tile<256, 128, 4, 2, true>(sm, M, lda, rows, (long)n, M + c0 * lda, lda, K, 256L * tk.b, 128L * (tk.c & 0xffff), false);
#else
          {
          // round 6: the two halves of the workgroup take the tile's upper and lower 128 rows independently (chol_update_tile_v, HV = 1)
          const int hv = chol_uniform(t >> 8), Ku = chol_uniform(K);
          const long ih = 256L * chol_uniform(tk.b) + 128L * hv, jh = 128L * chol_uniform(tk.c & 0xffff), c0u = chol_uniform(c0);
          if (ih < rows && jh <= ih + 127)   // (a half under the matrix or above the diagonal has nothing to do)
            chol_update_tile_v<true, 1>(sm + hv * (2 * kKC * kVLd), M, lda, rows, (long)n, M + c0u * lda, lda, Ku, ih, jh, false, slot + 16 + 32 * hv);
          }
#endif
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores ...
    __syncthreads();
    if (t == 0 && slot[14] == 0) {                        // ... then ONE lane publishes (a task of several strips: after its last)
      const CholTask tk = tasks[slot[0]];
      if (tk.type == 0) {
        const int k = tk.a, k0 = k * kNB, nb = (n - k0 < kNB) ? (n - k0) : kNB;
        __hip_atomic_fetch_add(&sdone[(size_t)k * nR + (size_t)(((long)k0 + nb + 64L * tk.b) / 256)], tk.c > 1 ? tk.c : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (tk.type == 3) {
        // quarters of ONE update run side by side; the one that completes the set moves the tile's sequence number on.  (The counter
        // only grows: panel a is the (a % W + 1)-th rank-128 update of this tile, which belongs to a's own outer panel.)
        const int J = tk.c & 0xffff, mine = (tk.c >> 18) & 7, full = (tk.c >> 21) & 7;
        const int old = __hip_atomic_fetch_add(&qdone[(size_t)tk.b * np + J], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == full * chol_tile_ridx(tk.b, J, W, SP, tk.a) + mine)
          __hip_atomic_store(&ver[(size_t)tk.b * np + J], chol_tile_seq(tk.b, J, W, SP, tk.a) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        const int J = tk.c & 0xffff;
        __hip_atomic_store(&ver[(size_t)tk.b * np + J], (tk.type == 1) ? chol_tile_seq(tk.b, J, W, SP, tk.a) + 1 : tk.a + (tk.c >> 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (stats) { lstat[1] += (long long)wall_clock64() - lstat[4]; lstat[2] += 1; }
    }
    __syncthreads();                                      // (slot[0] is free for the next task)
  }
}

inline hipError_t chol_sync_alloc(void** p, size_t bytes) { return hipMalloc(p, bytes); }   // (polled words: read through chol_peek only)

// Per-context state of the host driver: the look-ahead stream and its events live on the CONTEXT's device and are used by one
// context only (two contexts sharing them could wait on each other's records), and hipFuncSetAttribute is per device.
struct CholRuntime {
  hipStream_t side = nullptr;
  std::vector<hipEvent_t> ev_panel, ev_trail;
  hipStream_t comm = nullptr;                               // the distributed factorisation's message stream
  hipEvent_t ev_begin = nullptr;                            // everything the caller's stream held when the factorisation began
  std::vector<hipEvent_t> ev_packed, ev_recv, ev_unpacked;  // per outer panel: message staged / received / copied out of its staging buffer
  bool attr_set = false;
  int* bs_flags = nullptr; int bs_flags_cap = 0;            // one "x_p published" word per 128-panel of k_chol_backsub
  // persistent factorisation: the task list / strip counts of the last (n, W) on the device, the sync words, begin / done events
  CholPlan plan;
  CholTask* d_tasks = nullptr; int* d_ns = nullptr; int* d_sync = nullptr;
  size_t d_tasks_cap = 0, d_ns_cap = 0, d_sync_cap = 0;
  int sw_persistent = -1;                                   // ESL_CHOL_PERSISTENT = 1 / 0 forces it on / off (default: by size)
  int stats_on = 0;                                         // g_chol_stats_on as last written (ESL_CHOL_TIMING)
  int sw_potrf512 = -1;                                     // ESL_CHOL_POTRF512=1: diagnostic kernel shape in the launch path
  int sw_backsub = -1;                                      // ESL_CHOL_BACKSUB_LAUNCHES=1 keeps the launch-per-panel form (A/B)
  int sw_small = -1;                                        // ESL_CHOL_SMALL=0: orders <= 248 on the blocked path too (A/B)
  int n_cu = 0;                                             // hipDeviceProp_t::multiProcessorCount of the context's device (grids of the one-launch forms)
  int fallbacks = 0;                                        // trials redone launch-per-step after a device-side hand-off timed out (esl_slam.hip)
  // optional event bracket around the FACTORISATION alone (not the back-substitution): mark(user, 1) before, mark(user, 0) after
  void (*prof_mark)(void* user, int begin) = nullptr;
  void* prof_user = nullptr;
  hipStream_t cf_side = nullptr;                            // camera-first elimination: the segments' products run here beside the rank-K update (esl_slam.hip)
  hipEvent_t cf_ev_x = nullptr, cf_ev_p = nullptr;
  hipError_t cf_overlap_init() {
    if (cf_side) return hipSuccess;
    // (a plain second stream.  Two full-size grids on two streams do not run side by side -- the second kernel's workgroups are placed
    //  when the first one's are all dispatched -- and with the side stream confined to every 2nd / 4th / 8th CU of every XCD
    //  (hipExtStreamCreateWithCUMask) the products simply ran later AND slower: the bracket around T's build 29.3 -> 32.9 / 40.7 / 57.2 ms.
    //  What the second stream does give: the products fill the update's last, partly empty round of tiles: 30.0 -> 29.3 ms.)
    hipError_t e = hipStreamCreateWithFlags(&cf_side, hipStreamNonBlocking); if (e != hipSuccess) return e;
    e = hipEventCreateWithFlags(&cf_ev_x, hipEventDisableTiming); if (e != hipSuccess) return e;
    return hipEventCreateWithFlags(&cf_ev_p, hipEventDisableTiming);
  }
  void release() {
    if (cf_side) { (void)hipStreamDestroy(cf_side); cf_side = nullptr; }
    if (cf_ev_x) { (void)hipEventDestroy(cf_ev_x); cf_ev_x = nullptr; }
    if (cf_ev_p) { (void)hipEventDestroy(cf_ev_p); cf_ev_p = nullptr; }
    if (bs_flags) { (void)hipFree(bs_flags); bs_flags = nullptr; bs_flags_cap = 0; }
    if (d_tasks) { (void)hipFree(d_tasks); d_tasks = nullptr; d_tasks_cap = 0; }
    if (d_ns) { (void)hipFree(d_ns); d_ns = nullptr; d_ns_cap = 0; }
    if (d_sync) { (void)hipFree(d_sync); d_sync = nullptr; d_sync_cap = 0; }
    plan = CholPlan();
    for (hipEvent_t e : ev_panel) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_trail) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_packed) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_recv) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_unpacked) (void)hipEventDestroy(e);
    ev_panel.clear(); ev_trail.clear(); ev_packed.clear(); ev_recv.clear(); ev_unpacked.clear();
    if (side) { (void)hipStreamDestroy(side); side = nullptr; }
    if (comm) { (void)hipStreamDestroy(comm); comm = nullptr; }
    if (ev_begin) { (void)hipEventDestroy(ev_begin); ev_begin = nullptr; }
    attr_set = false;
  }
};

inline hipError_t chol_set_attributes(CholRuntime& rt) {   // dynamic-LDS limits of the two big kernels, once per context (= per device)
  if (rt.attr_set) return hipSuccess;
  const size_t lds = (size_t)(kNB * kLdsPad + 96 * kSB + kNB) * sizeof(double);
  hipError_t e = hipFuncSetAttribute((const void*)k_chol_potrf2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)k_chol_potrf2_512<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)k_chol_potrf2_512<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)k_chol_update_lds<256, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCholLdsBig);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)k_chol_update_lds<256, 128, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCholLdsBig);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)k_chol_backsub, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBsLds);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)k_chol_persist, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPwLds);
  if (e != hipSuccess) return e;
  {
    const int on = std::getenv("ESL_CHOL_TIMING") ? 1 : 0;
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_chol_stats_on), &on, sizeof(on));
    if (e != hipSuccess) return e;
    rt.stats_on = on;
  }
  {
    int dev = 0;
    hipDeviceProp_t prop;
    e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return e;
    rt.n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : kPwGrid;
  }
  rt.attr_set = true;
  return hipSuccess;
}

inline bool chol_fuse_default() { const char* e = std::getenv("ESL_CHOL_FUSE"); return !(e && e[0] == '0'); }
// Far updates of a tile per visit (chol_plan_build).  Measured on MI355X (round 6, scripts/chol_bench.py).  With ONE grid of group
// boundaries for all tiles (1 / 2 / 4 / 8 outer panels per visit): n = 8,192 7.22 / 7.53 / 7.76 / 8.65 ms, 18,000 39.5 / 39.1 / 38.9 / 40.4 -- the
// time in task bodies falls as predicted (36.5 -> 34.1 ms) but the far work arrives in bursts and the waiting grows (2.7 -> 3.8 ms).
// With the boundaries STAGGERED by tile (1 / 2 / 3 / 4 / 6 per visit): n = 4,096 2.97 / 2.93 / 2.98, 8,192 7.35 / 7.12 / 7.18 / 7.70, 12,000
// 15.35 / 14.85 / 15.64 / 16.7, 18,000 39.38 / 37.94 / 37.67 / 38.0 / 39.2, 24,000 87.7 / 83.9 / 83.5 / 84.7 / 86.3.
// Default: 3 from n = 16,384, else 2; ESL_CHOL_MERGE overrides (1 = the round 4-5 list).
// The strips of a row tile away from the diagonal as ONE task (ESL_CHOL_STRIPS=1): measured and NOT the default -- n = 4,096 2.95 -> 3.57 ms,
// 8,192 7.00 -> 8.31, 12,000 15.0 -> 17.2, 18,000 37.7 -> 38.9, 24,000 83.4 -> 84.7: four workgroups finish a row tile's solve sooner than
// one, and every trailing update of that row tile waits for it.
inline bool chol_strip_merge_default() { const char* e = std::getenv("ESL_CHOL_STRIPS"); return e && e[0] == '1'; }
// leading column panels of an outer panel with special chain tiles (chol_tile_special): ESL_CHOL_SPECIAL, default 2 (1 = rounds 4-5)
inline int chol_special_default() { const char* e = std::getenv("ESL_CHOL_SPECIAL"); const int v = e ? std::atoi(e) : 2; return v < 1 ? 1 : (v > 8 ? 8 : v); }
inline int chol_merge_default(int n) {
  const char* e = std::getenv("ESL_CHOL_MERGE");
  const int m = e ? std::atoi(e) : (n >= 16384 ? 3 : 2);
  return m < 1 ? 1 : (m > 16 ? 16 : m);
}
// the persistent kernel on the caller's stream (one launch; the sync words are cleared in front of it)
inline hipError_t chol_factor_persistent(double* M, long lda, int n, double* Linv_ws, int* info, hipStream_t st, CholRuntime& rt) {
  const char* wenv = std::getenv("ESL_CHOL_W");   // (debugging: outer-panel width of the persistent form)
  const int W = wenv ? std::max(1, std::atoi(wenv)) : chol_outer_panels(n);
  // ESL_CHOL_FUSE=0: strips under the diagonal block and the next block's update as worker tasks (the first form of round 4; A/B)
  const bool fuse = chol_fuse_default();
  const int merge = chol_merge_default(n);
  const bool smerge = chol_strip_merge_default();
  const int spc = std::min(chol_special_default(), W);
  if (rt.plan.n != n || rt.plan.W != W || rt.plan.fuse != (fuse ? 1 : 0) || rt.plan.merge != merge || rt.plan.strip_merge != (smerge ? 1 : 0) || rt.plan.sp != spc) {
    { hipError_t e0 = hipStreamSynchronize(st); if (e0 != hipSuccess) return e0; }   // the old list / sync words may still be in use
    const char* fenv = std::getenv("ESL_CHOL_FILLER");   // (debugging: far-update tasks interleaved per chain-dependent group)
    chol_plan_build(n, W, fenv ? std::max(1, std::atoi(fenv)) : 128, rt.plan, fuse, merge, smerge, spc);
    auto grow = [](void** p, size_t* cap, size_t need) -> hipError_t {
      if (*cap >= need) return hipSuccess;
      if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
      hipError_t e = hipMalloc(p, need); if (e == hipSuccess) *cap = need;
      return e;
    };
    hipError_t e = grow((void**)&rt.d_tasks, &rt.d_tasks_cap, std::max<size_t>(rt.plan.tasks.size(), 1) * sizeof(CholTask)); if (e != hipSuccess) return e;
    e = grow((void**)&rt.d_ns, &rt.d_ns_cap, std::max<size_t>(rt.plan.ns.size(), 1) * sizeof(int)); if (e != hipSuccess) return e;
    {
      const size_t need = chol_sync_words(rt.plan.np, rt.plan.nR) * sizeof(int);
      if (rt.d_sync_cap < need) {
        if (rt.d_sync) { (void)hipFree(rt.d_sync); rt.d_sync = nullptr; rt.d_sync_cap = 0; }
        e = chol_sync_alloc((void**)&rt.d_sync, need); if (e != hipSuccess) return e;
        rt.d_sync_cap = need;
      }
    }
    // once per system size.  The stream is NON-BLOCKING (no implicit ordering with the null stream's copies): wait for whatever still
    // reads the old list, then ship the new one ON the stream, in front of the launch that reads it.  The source is
    // rt.plan (lives as long as the runtime); a pageable source is staged by the runtime before hipMemcpyAsync returns.
    // (The buffers above were re-allocated after the same wait would have been due: nothing of this context runs during a plan
    // change -- the previous factorisation's caller has read its info word, i.e. synchronised -- but the wait makes it a fact.)
    e = hipStreamSynchronize(st); if (e != hipSuccess) return e;
    if (!rt.plan.tasks.empty()) { e = hipMemcpyAsync(rt.d_tasks, rt.plan.tasks.data(), rt.plan.tasks.size() * sizeof(CholTask), hipMemcpyHostToDevice, st); if (e != hipSuccess) return e; }
    e = hipMemcpyAsync(rt.d_ns, rt.plan.ns.data(), rt.plan.ns.size() * sizeof(int), hipMemcpyHostToDevice, st); if (e != hipSuccess) return e;
  }
  const CholPlan& pl = rt.plan;
  if (const char* tm = std::getenv("ESL_CHOL_TIMEOUT_MS")) {
    const long long ticks = 100000LL * std::max(1, std::atoi(tm));
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_chol_timeout_ticks), &ticks, sizeof(ticks)); if (e != hipSuccess) return e;
  }
  {
    const int on = std::getenv("ESL_CHOL_TIMING") ? 1 : 0;
    if (on != rt.stats_on) { hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_chol_stats_on), &on, sizeof(on)); if (e != hipSuccess) return e; rt.stats_on = on; }
  }
  hipError_t e = hipMemsetAsync(rt.d_sync, 0, chol_sync_words(pl.np, pl.nR) * sizeof(int), st); if (e != hipSuccess) return e;
  // one workgroup per CU of THIS device (the kernel is correct for any grid >= 2: the first arrival is the chain, the others pull tasks)
  const int grid = std::max(2, std::min(rt.n_cu > 0 ? rt.n_cu : kPwGrid, kPwGrid));
  hipLaunchKernelGGL(k_chol_persist, dim3((unsigned)grid), dim3(kPwThreads), kPwLds, st, M, lda, n, pl.np, pl.W, pl.nR, Linv_ws, (const CholTask*)rt.d_tasks,
                     (int)pl.tasks.size(), (const int*)rt.d_ns, rt.d_sync, info, pl.fuse, pl.sp);
  return hipGetLastError();
}

// Host driver.  M: (n+1) x n col-major (lda), Linv_ws: ceil(n/NB) * NB*NB doubles, z_ws: NB doubles,
// x: n doubles (output), info: device int (bit 0 set on a non-positive pivot).  The current device must be the one `rt` and
// `st` belong to.
inline hipError_t chol_factor_solve(double* M, long lda, int n, double* Linv_ws, double* z_ws, double* x, int* info,
                                    hipStream_t st, CholRuntime& rt, const CholDist* dist = nullptr) {
  const long rows = (long)n + 1;
  const size_t lds = (size_t)(kNB * kLdsPad + 96 * kSB + kNB) * sizeof(double);
  int np = (n + kNB - 1) / kNB;
  { hipError_t e = chol_set_attributes(rt); if (e != hipSuccess) return e; }
  // the bracket closes on EVERY way out of the factorisation, error returns included (an early return used to leave the
  // scope open, and later brackets were skipped or mis-paired)
  struct ProfBracket {
    CholRuntime& r; bool open;
    explicit ProfBracket(CholRuntime& rr) : r(rr), open(rr.prof_mark != nullptr) { if (open) r.prof_mark(r.prof_user, 1); }
    void close() { if (open) { r.prof_mark(r.prof_user, 0); open = false; } }
    ~ProfBracket() { close(); }
  } bracket(rt);
  auto launch_update = [&](hipStream_t stream, int kcol0, int K, long base, long col_limit) {
    chol_launch_update(M, lda, rows, stream, kcol0, K, base, col_limit);
  };
  // Outer panels of W inner panels (W x 128 columns): the inner panels are factored one after the other on the caller's
  // stream, each followed by a rank-128 update of the rest of the outer panel; the trailing matrix then gets ONE rank-(W x 128)
  // update (W = 2 halves, W = 4 quarters the read-modify-write traffic on C and the per-tile prologue / epilogue per flop).
  // Look-ahead: that update is split -- the columns of the NEXT outer panel are updated on the caller's stream, the rest on a
  // second stream, so the next outer panel's potrf / panel solves (single-workgroup and short kernels that leave the chip
  // empty) run underneath the big update instead of in front of it.
  // small systems on one GPU: one workgroup, one launch (k_chol_small; ESL_CHOL_SMALL=0 keeps the blocked path: A/B and the tests
  // that compare the two)
  if (rt.sw_persistent < 0) { const char* sw = std::getenv("ESL_CHOL_PERSISTENT"); rt.sw_persistent = sw ? (sw[0] == '1' ? 1 : 0) : 2; }
  if (!(dist && dist->n_ranks > 1) && chol_small_fits(n)) {
    if (rt.sw_small < 0) { const char* sw = std::getenv("ESL_CHOL_SMALL"); rt.sw_small = (sw && sw[0] == '0') ? 0 : 1; }
    if (rt.sw_small == 1 && rt.sw_persistent != 1) {   // (ESL_CHOL_PERSISTENT=1 asks for that kernel at every size: the tests of its chain)
      hipLaunchKernelGGL(k_chol_small, dim3(1), dim3(kCsThreads), 0, st, (const double*)M, lda, n, x, info);
      return hipGetLastError();
    }
  }
  hipStream_t& side = rt.side;
  std::vector<hipEvent_t>& ev_panel = rt.ev_panel;
  std::vector<hipEvent_t>& ev_trail = rt.ev_trail;
  if (!side) { hipError_t e = hipStreamCreateWithFlags(&side, hipStreamNonBlocking); if (e != hipSuccess) return e; }
  const int W = chol_outer_panels(n);
  const int n_outer = (np + W - 1) / W;
  while ((int)ev_panel.size() < n_outer) {
    hipEvent_t a, b;
    hipError_t e = hipEventCreateWithFlags(&a, hipEventDisableTiming); if (e != hipSuccess) return e;
    e = hipEventCreateWithFlags(&b, hipEventDisableTiming); if (e != hipSuccess) return e;
    ev_panel.push_back(a); ev_trail.push_back(b);
  }
  // below ~8k unknowns the chain of single-workgroup potrf launches is the critical path whatever runs beside it (measured:
  // n = 2994 4.55 ms without, 4.75 ms with look-ahead), from 16k on it buys 12-15 %
  const bool lookahead = n >= 8192;
  if (dist && dist->n_ranks > 1 && dist->pack && dist->pack2 && dist->bcast_on) {
    // ---- distributed, messages overlapped with the trailing updates --------------------------------------------------------------
    // Streams: st (the caller's: all arithmetic), rt.comm (the broadcasts, in panel order -- the same order on every rank).
    // Buffers: panel o travels in pack (o even) / pack2 (o odd).  Events per panel o: ev_packed (owner: factored + staged, on st),
    // ev_recv (message complete, on comm), ev_unpacked (non-owners: copied out into M / Linv_ws, on st).  Ordering that keeps the two
    // buffers safe: the broadcast of panel o waits for ev_packed[o] on the owner and for ev_unpacked[o - 2] on the others (the last
    // reader of that buffer); the owner stages panel o on st after st has waited for ev_recv[o - 1], and comm runs in order, so the
    // broadcast of panel o - 2 (the last sender from that buffer) is over by then.
    // Every panel still receives the updates of panels 0, 1, 2, .. in that order: same bits as the form below.
    const int R = dist->n_ranks, me = dist->rank;
    if (!rt.comm) { hipError_t e = hipStreamCreateWithFlags(&rt.comm, hipStreamNonBlocking); if (e != hipSuccess) return e; }
    if (!rt.ev_begin) { hipError_t e = hipEventCreateWithFlags(&rt.ev_begin, hipEventDisableTiming); if (e != hipSuccess) return e; }
    {   // the message stream starts behind everything the caller's stream already holds (its earlier collectives on this communicator
        // among it: two collectives of one communicator must not run side by side)
      hipError_t e = hipEventRecord(rt.ev_begin, st); if (e != hipSuccess) return e;
      e = hipStreamWaitEvent(rt.comm, rt.ev_begin, 0); if (e != hipSuccess) return e;
    }
    while ((int)rt.ev_packed.size() < n_outer) {
      hipEvent_t a, b, c3;
      hipError_t e = hipEventCreateWithFlags(&a, hipEventDisableTiming); if (e != hipSuccess) return e;
      e = hipEventCreateWithFlags(&b, hipEventDisableTiming); if (e != hipSuccess) return e;
      e = hipEventCreateWithFlags(&c3, hipEventDisableTiming); if (e != hipSuccess) return e;
      rt.ev_packed.push_back(a); rt.ev_recv.push_back(b); rt.ev_unpacked.push_back(c3);
    }
    struct Geo { int p0, p1, c_begin, c_end; long h, w; size_t n_inv; double* buf; };
    auto geo = [&](int o) {
      Geo g;
      g.p0 = o * W; g.p1 = (g.p0 + W < np) ? g.p0 + W : np;
      g.c_begin = g.p0 * kNB; g.c_end = (g.p1 * kNB < n) ? g.p1 * kNB : n;
      g.h = rows - g.c_begin; g.w = g.c_end - g.c_begin; g.n_inv = (size_t)(g.p1 - g.p0) * kNB * kNB;
      g.buf = (o & 1) ? dist->pack2 : dist->pack;
      return g;
    };
    auto factor_and_stage = [&](int o) -> hipError_t {   // this rank owns panel o; everything on st
      const Geo g = geo(o);
      for (int p = g.p0; p < g.p1; ++p) {
        const int k0 = p * kNB, nb = (n - k0 < kNB) ? (n - k0) : kNB;
        double* Linv = Linv_ws + (size_t)p * kNB * kNB;
        hipLaunchKernelGGL(k_chol_potrf2, dim3(1), dim3(kP2Threads), lds, st, M, lda, k0, nb, Linv, info);
        const long below = rows - (k0 + nb);
        if (below > 0) hipLaunchKernelGGL(k_chol_panel, dim3((unsigned)((below + 63) / 64)), dim3(512), 0, st, M, lda, rows, k0, nb, Linv);
        if (p + 1 < g.p1) launch_update(st, k0, nb, (long)k0 + nb, (long)g.c_end);
      }
      hipLaunchKernelGGL(k_chol_pack, dim3((unsigned)((g.h + 255) / 256), (unsigned)g.w), dim3(256), 0, st, M, lda, rows, (long)g.c_begin, (long)g.c_begin, g.buf, 0);
      hipError_t e = hipMemcpyAsync(g.buf + (size_t)g.h * g.w, Linv_ws + (size_t)g.p0 * kNB * kNB, g.n_inv * sizeof(double), hipMemcpyDeviceToDevice, st);
      if (e != hipSuccess) return e;
      e = hipGetLastError(); if (e != hipSuccess) return e;
      return hipEventRecord(rt.ev_packed[o], st);
    };
    auto update_with = [&](int o, int o2) {   // panel o2 -= (its rows of panel o) (panel o's rows of panel o2's columns)^T
      const Geo g = geo(o), q = geo(o2);
      launch_update(st, g.c_begin, g.c_end - g.c_begin, (long)q.c_begin, (long)q.c_end);
    };
    if (0 % R == me) { hipError_t e = factor_and_stage(0); if (e != hipSuccess) return e; }
    for (int o = 0; o < n_outer; ++o) {
      const int owner = o % R;
      const Geo g = geo(o);
      hipError_t e;
      if (owner == me) e = hipStreamWaitEvent(rt.comm, rt.ev_packed[o], 0);
      else e = (o >= 2) ? hipStreamWaitEvent(rt.comm, rt.ev_unpacked[o - 2], 0) : hipSuccess;
      if (e != hipSuccess) return e;
      if (dist->bcast_on(dist->user, g.buf, (size_t)g.h * g.w + g.n_inv, owner, rt.comm)) return hipErrorUnknown;
      e = hipEventRecord(rt.ev_recv[o], rt.comm); if (e != hipSuccess) return e;
      e = hipStreamWaitEvent(st, rt.ev_recv[o], 0); if (e != hipSuccess) return e;
      if (owner != me) {
        hipLaunchKernelGGL(k_chol_pack, dim3((unsigned)((g.h + 255) / 256), (unsigned)g.w), dim3(256), 0, st, M, lda, rows, (long)g.c_begin, (long)g.c_begin, g.buf, 1);
        e = hipMemcpyAsync(Linv_ws + (size_t)g.p0 * kNB * kNB, g.buf + (size_t)g.h * g.w, g.n_inv * sizeof(double), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return e;
      }
      e = hipEventRecord(rt.ev_unpacked[o], st); if (e != hipSuccess) return e;
      // look-ahead: the next panel, if it is this rank's, before anything else
      const bool next_mine = o + 1 < n_outer && (o + 1) % R == me;
      if (next_mine) {
        update_with(o, o + 1);
        e = factor_and_stage(o + 1); if (e != hipSuccess) return e;
      }
      for (int o2 = o + 1 + (next_mine ? 1 : 0); o2 < n_outer; ++o2)
        if (o2 % R == me) update_with(o, o2);
    }
  } else if (dist && dist->n_ranks > 1) {
    for (int o = 0; o < n_outer; ++o) {
      const int owner = o % dist->n_ranks;
      const int p0 = o * W, p1 = (p0 + W < np) ? p0 + W : np;
      const int c_begin = p0 * kNB, c_end = (p1 * kNB < n) ? p1 * kNB : n;
      if (owner == dist->rank) {
        for (int p = p0; p < p1; ++p) {
          const int k0 = p * kNB, nb = (n - k0 < kNB) ? (n - k0) : kNB;
          double* Linv = Linv_ws + (size_t)p * kNB * kNB;
          hipLaunchKernelGGL(k_chol_potrf2, dim3(1), dim3(kP2Threads), lds, st, M, lda, k0, nb, Linv, info);
          const long below = rows - (k0 + nb);
          if (below > 0) hipLaunchKernelGGL(k_chol_panel, dim3((unsigned)((below + 63) / 64)), dim3(512), 0, st, M, lda, rows, k0, nb, Linv);
          if (p + 1 < p1) launch_update(st, k0, nb, (long)k0 + nb, (long)c_end);
        }
        hipError_t e = hipGetLastError(); if (e != hipSuccess) return e;
      }
      if (dist->pack) {   // one message: the panel from its diagonal down, then the diagonal blocks' inverses
        const long h = rows - c_begin, w = c_end - c_begin;
        const size_t n_inv = (size_t)(p1 - p0) * kNB * kNB;
        const dim3 grid((unsigned)((h + 255) / 256), (unsigned)w);
        if (owner == dist->rank) {
          hipLaunchKernelGGL(k_chol_pack, grid, dim3(256), 0, st, M, lda, rows, (long)c_begin, (long)c_begin, dist->pack, 0);
          hipError_t e = hipMemcpyAsync(dist->pack + (size_t)h * w, Linv_ws + (size_t)p0 * kNB * kNB, n_inv * sizeof(double), hipMemcpyDeviceToDevice, st);
          if (e != hipSuccess) return e;
        }
        if (dist->bcast(dist->user, dist->pack, (size_t)h * w + n_inv, owner)) return hipErrorUnknown;
        if (owner != dist->rank) {
          hipLaunchKernelGGL(k_chol_pack, grid, dim3(256), 0, st, M, lda, rows, (long)c_begin, (long)c_begin, dist->pack, 1);
          hipError_t e = hipMemcpyAsync(Linv_ws + (size_t)p0 * kNB * kNB, dist->pack + (size_t)h * w, n_inv * sizeof(double), hipMemcpyDeviceToDevice, st);
          if (e != hipSuccess) return e;
        }
      } else {
        // the factored columns (whole columns: the rows above the diagonal travel along, unused) and the diagonal blocks' inverses
        if (dist->bcast(dist->user, M + (size_t)c_begin * lda, (size_t)lda * (size_t)(c_end - c_begin), owner)) return hipErrorUnknown;
        if (dist->bcast(dist->user, Linv_ws + (size_t)p0 * kNB * kNB, (size_t)(p1 - p0) * kNB * kNB, owner)) return hipErrorUnknown;
      }
      // this rank's later outer panels
      for (int o2 = o + 1; o2 < n_outer; ++o2) {
        if (o2 % dist->n_ranks != dist->rank) continue;
        const int q0 = o2 * W, q1 = (q0 + W < np) ? q0 + W : np;
        const long b2 = (long)q0 * kNB, e2 = ((long)q1 * kNB < n) ? (long)q1 * kNB : n;
        launch_update(st, c_begin, c_end - c_begin, b2, e2);
      }
    }
  } else if (([&]() { // by size, measured on MI355X (profiles/r4_cholesky_microbench.txt), one launch vs launch per step: n = 2,994 2.43 vs
                      // 2.24 ms, 4,096 3.23 vs 3.24, 6,000 4.88 vs 5.75, 8,192 7.94 vs 8.95, 12,000 15.9 vs 17.9, 18,000 40.3 vs 43.9,
                      // 24,000 88.0 vs 91.8, 32,768 214.4 vs 214.5 (a draw: the launch path is kept there, it is the form that has run the
                      // order-59,994 reduced camera system since round 1)
                      return rt.sw_persistent == 1 || (rt.sw_persistent == 2 && n >= 4096 && n < 30000); })()) {
    hipError_t e = chol_factor_persistent(M, lda, n, Linv_ws, info, st, rt); if (e != hipSuccess) return e;
  } else {
  bool trail_pending = false;   // ev_trail[o - 1] has been recorded and not yet waited for
  for (int o = 0; o < n_outer; ++o) {
    const int p0 = o * W, p1 = (p0 + W < np) ? p0 + W : np;
    const int c_begin = p0 * kNB, c_end = (p1 * kNB < n) ? p1 * kNB : n;   // columns of this outer panel
    for (int p = p0; p < p1; ++p) {
      const int k0 = p * kNB, nb = (n - k0 < kNB) ? (n - k0) : kNB;
      double* Linv = Linv_ws + (size_t)p * kNB * kNB;
      if (rt.sw_potrf512 < 0) { const char* e5 = std::getenv("ESL_CHOL_POTRF512"); rt.sw_potrf512 = e5 ? std::atoi(e5) : 0; }
      if (rt.sw_potrf512 == 1) hipLaunchKernelGGL(k_chol_potrf2_512<true>, dim3(1), dim3(512), lds, st, M, lda, k0, nb, Linv, info);
      else if (rt.sw_potrf512 == 2) hipLaunchKernelGGL(k_chol_potrf2_512<false>, dim3(1), dim3(512), lds, st, M, lda, k0, nb, Linv, info);
      else hipLaunchKernelGGL(k_chol_potrf2, dim3(1), dim3(kP2Threads), lds, st, M, lda, k0, nb, Linv, info);
      const long below = rows - (k0 + nb);
      if (below > 0) hipLaunchKernelGGL(k_chol_panel, dim3((unsigned)((below + 63) / 64)), dim3(512), 0, st, M, lda, rows, k0, nb, Linv);
      // bring the rest of the outer panel's columns up to date (rank-nb update restricted to those columns)
      if (p + 1 < p1) launch_update(st, k0, nb, (long)k0 + nb, (long)c_end);
    }
    if (c_end >= n) break;
    const int K = c_end - c_begin;
    const int next_end = ((p1 + W) * kNB < n) ? (p1 + W) * kNB : n;        // end of the next outer panel's columns
    if (!lookahead || next_end >= n) {
      if (trail_pending) { hipError_t e = hipStreamWaitEvent(st, ev_trail[o - 1], 0); if (e != hipSuccess) return e; trail_pending = false; }
      launch_update(st, c_begin, K, (long)c_end, (long)n);
      continue;
    }
    hipError_t e = hipEventRecord(ev_panel[o], st); if (e != hipSuccess) return e;
    // the next outer panel's columns: written by the previous side-stream update as well -> wait for it first
    if (trail_pending) { e = hipStreamWaitEvent(st, ev_trail[o - 1], 0); if (e != hipSuccess) return e; trail_pending = false; }
    launch_update(st, c_begin, K, (long)c_end, (long)next_end);
    // everything to the right of them, underneath the next outer panel's factorisation
    e = hipStreamWaitEvent(side, ev_panel[o], 0); if (e != hipSuccess) return e;
    launch_update(side, c_begin, K, (long)next_end, (long)n);
    e = hipEventRecord(ev_trail[o], side); if (e != hipSuccess) return e;
    trail_pending = true;
  }
  if (trail_pending) { hipError_t e = hipStreamWaitEvent(st, ev_trail[n_outer - 2], 0); if (e != hipSuccess) return e; }
  }
  bracket.close();
  if (rt.sw_backsub < 0) { const char* sw = std::getenv("ESL_CHOL_BACKSUB_LAUNCHES"); rt.sw_backsub = (sw && sw[0] == '1') ? 1 : 0; }
  // the one-launch form reads the factor as 16-byte pairs: even leading dimension, 16-byte aligned base
  if (rt.sw_backsub == 0 && (lda & 1) == 0 && ((uintptr_t)M & 15) == 0 && ((uintptr_t)Linv_ws & 15) == 0) {
    if (rt.bs_flags_cap < np + 1) {   // np panel flags + the arrival-ticket word
      if (rt.bs_flags) { (void)hipFree(rt.bs_flags); rt.bs_flags = nullptr; rt.bs_flags_cap = 0; }
      const int cap = std::max(np + 1, 512);
      hipError_t e = chol_sync_alloc((void**)&rt.bs_flags, (size_t)cap * sizeof(int)); if (e != hipSuccess) return e;
      rt.bs_flags_cap = cap;
    }
    hipError_t e = hipMemsetAsync(rt.bs_flags, 0, (size_t)(np + 1) * sizeof(int), st); if (e != hipSuccess) return e;
    const int bs_grid = std::min(np, std::min(rt.n_cu > 0 ? rt.n_cu : kBsMaxGrid, kBsMaxGrid));
    hipLaunchKernelGGL(k_chol_backsub, dim3((unsigned)bs_grid), dim3(kBsThreads), kBsLds, st, M, lda, n, np, Linv_ws, x, rt.bs_flags, info);
    return hipGetLastError();
  }
  for (int p = np - 1; p >= 0; --p) {
    const int k0 = p * kNB, nb = (n - k0 < kNB) ? (n - k0) : kNB;
    hipLaunchKernelGGL(k_chol_backdot, dim3(nb), dim3(256), 0, st, M, lda, n, k0, nb, x, z_ws);
    hipLaunchKernelGGL(k_chol_backsolve, dim3((nb + 3) / 4), dim3(256), 0, st, Linv_ws + (size_t)p * kNB * kNB, k0, nb, z_ws, x);
  }
  return hipGetLastError();
}

}  // namespace esl
