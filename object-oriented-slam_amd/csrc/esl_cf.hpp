// esl_cf.hpp — SLAM mode, second elimination order: CAMERAS FIRST (round 3).
//
// g2o's BlockSolver eliminates the vertices marked as marginalised and hands the rest to the linear solver
// (Thirdparty/g2o/g2o/core/block_solver.hpp:367-486); which set that is does not change the solution of
//     [ A   W ] [x_c]   [b_c]        A = Hcc + lambda I  (cameras),   D = Hoo + lambda I  (ellipsoids)
//     [ W^T D ] [x_o] = [b_o]
// esl_kernels_slam.hpp eliminates the ellipsoids (D is block diagonal) and factors the dense reduced CAMERA system
// S = A - W D^-1 W^T of order 6 (F - 1).  In the reference's SLAM branch the camera block A is block TRIDIAGONAL -- the only
// camera-camera edges are the odometry edges between consecutive frames (src/core/Optimizer.cpp:142-158) -- and there are
// five times fewer ellipsoid unknowns than camera unknowns (C4: 18,000 vs 59,994).  Eliminating the cameras instead:
//     A = L L^T                         block-bidiagonal Cholesky, one sequential sweep over the cameras        k_cf_gather_B, k_cf_chain, k_cf_factor_blocks
//     X = L^-1 W,  y = L^-1 b_c         forward substitution, one lane per column (9 N + 1 columns)             k_cf_edge_scale, k_cf_forward
//     T = D - X^T X,  t = b_o - X^T y   rank-6(F-1) update on the FP64 matrix cores (k_chol_update_lds, external factor)
//     T x_o = t                         dense FP64-MFMA Cholesky of order 9 N (esl_chol.hpp)
//     x_c = L^-T (y - X x_o)            k_cf_z, k_cf_tridiag_back
// Flops per LM trial with X dense: (9N)^2 6F + (9N)^3 / 3 instead of (6F)^3 / 3 -- C4: 2.1e13 instead of 7.2e13, C3: 6.4e8 instead of
// 8.9e9; memory: X 8.6 GB + T 2.6 GB instead of S 28.8 GB.  Same x up to rounding (it is the same linear system), so every parity
// test of SLAM mode applies unchanged.  Positive-definiteness ("ok" of LinearSolverDense, solvers/linear_solver_dense.h:107-112)
// = all pivots of A and of T positive.
//
// Two refinements (both further down in this file), each the same factorisation in another order:
//   nested dissection of the camera chain  every stride-th camera a separator, the segments between them factored / substituted in
//                                          parallel: serial length F -> stride + F / stride;
//   X kept sparse                          (long chains, large T: C4) stride 16, the interior rows of X as compact per-segment
//                                          slabs, T assembled from per-segment MFMA products, the dense MFMA update only for the
//                                          separators' 6 F / 16 rows: C4 3.3e12 flops per trial instead of 2.1e13.
//
// Applicable when every odometry edge joins two cameras whose free-camera slots differ by one (the reference's chain); across GPUs
// with the replicated-graph communicator (esl_comm_set_replicated: the ranks divide T by outer panels); esl_lm_params::linear_solver
// selects (include/esl.h).
#pragma once
#include "esl_kernels_map.hpp"

namespace esl {

__device__ __forceinline__ double cf_readlane(double v, int src) {   // src: compile-time constant after unrolling
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// ---- A = L L^T for the block-tridiagonal camera matrix -------------------------------------------------------------------
// Three launches: (1) k_cf_gather_B, parallel: B_i = A_{i+1,i} from the odometry edges of slot pair (i, i + 1) -- od_start / od_edge,
// entry = edge * 2 + t; t = 1: the edge's first vertex is slot i, so its Hij block (rows = first vertex) is the TRANSPOSE of B_i;
// (2) k_cf_chain, ONE wave, sequential: only what is inherently serial, the Schur-complement recurrence
//         D_0 = Hcc_0 + lambda I,   D_{i+1} = Hcc_{i+1} + lambda I - G_i^T G_i,   G_i = L_ii^-1 B_i^T,   L_ii = chol(D_i)
//     Every lane carries the whole 6 x 6 problem in registers (no LDS hand-overs, no cross-lane traffic: a dependent LDS round trip
//     costs more than the ~60 instructions it would save); the next step's Hcc and B are in flight one step ahead.  Round 3's first
//     form (36 lanes, one entry each, blocks through LDS, odometry blocks gathered inside the loop) took 2.0 us per camera -- three
//     dependent global loads and ~18 dependent LDS round trips per step; this one 0.5 us.
// (3) k_cf_factor_blocks, parallel (one thread per camera): everything else the substitutions need, from L_ii and G_i:
//         Li = L_ii^-1,   Lo = L_{i,i-1} = G_{i-1}^T,   M_i = Li Lo   (X_i = Li W_i - M_i X_{i-1}),
//         N_i = Li^T L_{i+1,i}^T = Li^T G_i                            (x_i = Li^T z_i - N_i x_{i+1})
static __global__ __launch_bounds__(256) void k_cf_gather_B(int nf, const double* __restrict__ Aod, const int* __restrict__ od_start,
                                                            const int* __restrict__ od_edge, double* __restrict__ B) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long i = t / 36;
  if (i >= nf) return;
  const int e = (int)(t - i * 36), r = e / 6, c = e - 6 * r;
  double s = 0;
  if (i + 1 < nf)
    for (int q = od_start[i]; q < od_start[i + 1]; ++q) {
      const int es = od_edge[q];
      const double* Hij = Aod + (size_t)(es >> 1) * 90 + 54;
      s += (es & 1) ? Hij[c * 6 + r] : Hij[r * 6 + c];
    }
  B[t] = s;   // (block nf - 1 is zero: nothing follows the last camera)
}

// packed lower triangle: (r, c), r >= c, at r (r + 1) / 2 + c
#define CF_LT(r, c) ((r) * ((r) + 1) / 2 + (c))
// Memory never sits inside the serial loop: Hcc and B of kCfChainCh cameras are staged in LDS (the next chunk's global loads are
// issued before the chunk's steps run and land in LDS after them), L and G of the chunk collect in LDS and leave as coalesced
// stores.  (With the loads inside the loop the compiler serialised three scalar-load round trips per camera: 2.3 us per step.)
constexpr int kCfChainCh = 32;
// Nested dissection (see the file header): workgroup p runs the recurrence over segment p = cameras [p stride, p stride + seglen)
// starting from D = Hcc (no incoming G); one segment covering everything (stride >= nf) is the plain chain.
static __global__ __launch_bounds__(64) void k_cf_chain(int nf_all, const double* __restrict__ Hcc, const double* __restrict__ B, double lambda,
                                                        double* __restrict__ Lfac /* nf x 21: L_ii, packed lower */,
                                                        double* __restrict__ Gfac /* nf x 36: G_i = L_ii^-1 B_i^T, row-major */,
                                                        int* __restrict__ info, int stride, int seglen) {
  __shared__ double sH[kCfChainCh * 36], sB[kCfChainCh * 36], sL[kCfChainCh * 21], sG[kCfChainCh * 36];
  constexpr int kPer = kCfChainCh * 36 / 64;   // doubles per lane and array in one chunk
  const int lane = threadIdx.x;
  const int beg = (int)blockIdx.x * stride;
  const int nf = (beg + seglen < nf_all) ? beg + seglen : nf_all;   // end of this segment
  if (beg >= nf) return;
  double rh[kPer], rb[kPer];
  auto gload = [&](int i0) {
    const size_t lim = (size_t)nf_all * 36;
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      const size_t idx = (size_t)i0 * 36 + (size_t)q * 64 + lane;
      rh[q] = idx < lim ? Hcc[idx] : 0.0;
      rb[q] = idx < lim ? B[idx] : 0.0;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int q = 0; q < kPer; ++q) { sH[q * 64 + lane] = rh[q]; sB[q * 64 + lane] = rb[q]; }
  };
  gload(beg);
  lstore();
  __builtin_amdgcn_wave_barrier();
  bool ok = true;
  double g[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) g[k] = 0;
  for (int i0 = beg; i0 < nf; i0 += kCfChainCh) {
    const int len = (nf - i0 < kCfChainCh) ? nf - i0 : kCfChainCh;
    const bool more = i0 + kCfChainCh < nf;
    if (more) gload(i0 + kCfChainCh);
    for (int ii = 0; ii < len; ++ii) {
      // D_i = Hcc_i + lambda I - G_{i-1}^T G_{i-1}
      double d[21], b[36];
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c) {
          double v = sH[ii * 36 + r * 6 + c] + ((r == c) ? lambda : 0.0);
#pragma unroll
          for (int k = 0; k < 6; ++k) v -= g[k * 6 + r] * g[k * 6 + c];
          d[CF_LT(r, c)] = v;
        }
#pragma unroll
      for (int k = 0; k < 36; ++k) b[k] = sB[ii * 36 + k];
      // L_ii = chol(D_i), right-looking, in place
      double is[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const double p = d[CF_LT(j, j)];
        ok = ok && (p > 0);
        double q = __builtin_amdgcn_rsq(p);   // 1 / sqrt(p): hardware estimate + two Newton steps
        q = q * (1.5 - 0.5 * p * q * q);
        q = q * (1.5 - 0.5 * p * q * q);
        is[j] = q;
        d[CF_LT(j, j)] = p * q;
#pragma unroll
        for (int r = j + 1; r < 6; ++r) d[CF_LT(r, j)] *= q;
#pragma unroll
        for (int r = j + 1; r < 6; ++r)
#pragma unroll
          for (int c = j + 1; c <= r; ++c) d[CF_LT(r, c)] -= d[CF_LT(r, j)] * d[CF_LT(c, j)];
      }
      // G = L^-1 B^T by forward substitution: column c of G solves L g = (row c of B)^T
#pragma unroll
      for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          double v = b[c * 6 + r];
#pragma unroll
          for (int k = 0; k < r; ++k) v -= d[CF_LT(r, k)] * g[k * 6 + c];
          g[r * 6 + c] = v * is[r];
        }
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 21; ++k) sL[ii * 21 + k] = d[k];
#pragma unroll
        for (int k = 0; k < 36; ++k) sG[ii * 36 + k] = g[k];
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (more) lstore();                      // the next chunk's blocks (their loads ran under the steps above)
    for (int idx = lane; idx < len * 21; idx += 64) Lfac[(size_t)i0 * 21 + idx] = sL[idx];
    for (int idx = lane; idx < len * 36; idx += 64) Gfac[(size_t)i0 * 36 + idx] = sG[idx];
    __builtin_amdgcn_wave_barrier();
  }
  if (!ok && lane == 0) atomicOr(info, 1);
}

// stride > 0 (nested dissection): slots with i % stride == stride - 1 are separators (not touched here), the first camera of a
// segment has no predecessor (M = 0) and the last none to hand its x to (N = 0; its G couples it to the separator instead)
// scat (round 6; the separator chain's call): L_ii^-1 of chain position i ALSO goes to scat[(i scat_stride + scat_stride - 1) * 36 ..], the
// per-slot array the other kernels read (was a launch of its own, k_cf_sep_scatter)
static __global__ __launch_bounds__(64) void k_cf_factor_blocks(int nf, const double* __restrict__ Lfac, const double* __restrict__ Gfac,
                                                                double* __restrict__ Linv, double* __restrict__ Mm, double* __restrict__ Nn,
                                                                int stride, double* __restrict__ scat = nullptr, int scat_stride = 0) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= nf) return;
  const int pos = stride > 0 ? i % stride : i;
  if (stride > 0 && pos == stride - 1) return;
  const bool first = pos == 0, last = (i == nf - 1) || (stride > 0 && pos == stride - 2);
  double L[21], Li[36];
#pragma unroll
  for (int k = 0; k < 21; ++k) L[k] = Lfac[(size_t)i * 21 + k];
  // Li = L^-1 (lower): column c by forward substitution
#pragma unroll
  for (int c = 0; c < 6; ++c)
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      if (r < c) { Li[r * 6 + c] = 0; continue; }
      double v = (r == c) ? 1.0 : 0.0;
#pragma unroll
      for (int k = c; k < r; ++k) v -= L[CF_LT(r, k)] * Li[k * 6 + c];
      Li[r * 6 + c] = v / L[CF_LT(r, r)];
    }
#pragma unroll
  for (int k = 0; k < 36; ++k) Linv[(size_t)i * 36 + k] = Li[k];
  if (scat) {
#pragma unroll
    for (int k = 0; k < 36; ++k) scat[(size_t)(i * scat_stride + scat_stride - 1) * 36 + k] = Li[k];
  }
  // M_i = Li Lo, Lo = L_{i,i-1} = G_{i-1}^T (zero for the first camera)
  double Lo[36];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) Lo[r * 6 + c] = first ? 0.0 : Gfac[(size_t)(i - 1) * 36 + c * 6 + r];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double m = 0;
#pragma unroll
      for (int k = 0; k <= r; ++k) m += Li[r * 6 + k] * Lo[k * 6 + c];
      Mm[(size_t)i * 36 + r * 6 + c] = m;
    }
  // N_i = Li^T G_i (G of the last camera is zero: B_{nf-1} = 0)
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double nv = 0;
#pragma unroll
      for (int k = r; k < 6; ++k) nv += Li[k * 6 + r] * Gfac[(size_t)i * 36 + k * 6 + c];
      Nn[(size_t)i * 36 + r * 6 + c] = last ? 0.0 : nv;
    }
}
#undef CF_LT

// ---- V_k = Li_{slot(k)} W_{edge(k)} for every entry k of the per-ellipsoid edge lists (sorted by slot), vy_s = Li_s b_c,s ----
// thread = (list entry k, ellipsoid column b) or, behind those, one per free camera.  Records in LIST order ([k][b][a], a = camera
// row): the forward substitution finds the record of entry k of column b without going through the edge id.  bbox edges dropped by
// the NaN / visibility pre-check get V = 0 (the lists still hold them).
__device__ __forceinline__ void cf_edge_scale_one(const DevGraph& g, int n_list, const int* __restrict__ oe_u, const int* __restrict__ oe_slot,
                                                  const double* __restrict__ W, const double* __restrict__ Linv,
                                                  const double* __restrict__ bc, double* __restrict__ V,
                                                  double* __restrict__ vy, long t) {
  const long EU = (long)g.n_bbox + g.n_e3d;
  if (t < (long)n_list * 9) {
    const long k = t / 9;
    const int b = (int)(t - k * 9);
    const long u = oe_u[k];
    const int slot = oe_slot[k];
    const bool valid = !(u < g.n_bbox) || g.bb_valid[u];
    double w[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) w[q] = valid ? W[(long)(q * 9 + b) * EU + u] : 0.0;
    const double* Li = Linv + (size_t)slot * 36;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      double s = 0;
#pragma unroll
      for (int q = 0; q <= a; ++q) s += Li[a * 6 + q] * w[q];
      V[t * 6 + a] = s;
    }
  } else {
    const long s = t - (long)n_list * 9;
    if (s >= g.n_free_cams) return;
    const double* Li = Linv + (size_t)s * 36;
    double w[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) w[q] = bc[(size_t)s * 6 + q];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      double v = 0;
#pragma unroll
      for (int q = 0; q <= a; ++q) v += Li[a * 6 + q] * w[q];
      vy[(size_t)s * 6 + a] = v;
    }
  }
}
static __global__ __launch_bounds__(256) void k_cf_edge_scale(DevGraph g, int n_list, const int* __restrict__ oe_u, const int* __restrict__ oe_slot,
                                                              const double* __restrict__ W, const double* __restrict__ Linv,
                                                              const double* __restrict__ bc, double* __restrict__ V,
                                                              double* __restrict__ vy) {
  cf_edge_scale_one(g, n_list, oe_u, oe_slot, W, Linv, bc, V, vy, (long)blockIdx.x * 256 + threadIdx.x);
}

// ---- X = L^-1 [W | b_c]: one lane per column (9 per ellipsoid + the right-hand side), sequential over the cameras -----------
// X_i = V_i - M_i X_{i-1}; V_i is non-zero only where the column's ellipsoid has an edge at camera i.  Chunks of kCfFwdCh cameras,
// three phases each, so that no memory operation sits inside the serial loop:
//   (1) gather: the M blocks of the chunk -> LDS; every lane adds the V records of ITS column's list entries that fall into the
//       chunk into an LDS slab v[step][row][lane] (entry range of (ellipsoid, chunk) from a table built at upload: all loads of
//       a batch of entries go out together);
//   (2) kCfFwdCh steps on LDS only: x = v - M x, x written over v;
//   (3) the slab leaves as coalesced stores: Xt is the TRANSPOSE of X (element (column j, row k) at Xt[j + k ldx]) -- the layout
//       the MFMA rank-K update wants for its factor.
// History: list walk inside the loop with the next record prefetched: 1.08 us per camera (a store and a dependent load per step:
// the compiler's s_waitcnt vmcnt(0) at the loop head made every step wait for the previous step's stores); this form 0.2 us.
constexpr int kCfFwdCh = 16;
// MODE 0: blockIdx.y = segment p (cameras [p stride, p stride + seglen); one segment = the plain chain).  With nested
//   dissection (Zt != null) the lane also accumulates acc = sum_j Zt_j^T X_j over its segment -- the segment's contribution to the
//   right-hand side of the separator in FRONT of it -- into R[column + (6 (p - 1) + a) ldx]; list entries of separator slots are
//   not this kernel's (they fall outside [beg, end)).
// MODE 1 (SEP): the same recurrence over the separators (nf = number of separators, M = the separator chain's blocks), V dense from
//   R (k_cf_sep_rhs), rows written at the separators' slots k sep_stride + sep_stride - 1 (sep_stride = 0: at row block k).
// MODE 2 (X is sparse, see "sparse interior rows" below): segments of kCfFwdCh slots, workgroup = entry of the work list
//   (segment p, group of 64 COMPACT columns); compact column 9 i + b = column b of the i-th ellipsoid of the segment's list, the
//   column after the last one = the right-hand side; rows go to the segment's slab Xc (xoff / xld), acc to R at the T column.
constexpr int kCfSyT = 128;     // tile of the per-segment product kernel (compact columns)
constexpr int kCfSegRows = 96;   // rows of a slab: 6 (kCfFwdCh - 1) = 90, padded with zero rows to a multiple of the T kernel's row batch
// rank / unrank (round 5, null = identity): T's ellipsoids in the order of their FIRST free camera -- column block rank[o] of T, R, Xs
// belongs to ellipsoid o.  The separators' rows of X are then zero above a staircase (an ellipsoid no camera has seen yet has no
// entry in any R(k)), and the dense rank-K update skips them tile by tile (esl_chol.hpp, kfirst).
struct CfSegs {
  const int* fwork; const int* seg_start; const int* seg_obj; const long long* xoff; const int* xld;
  const int* rank; const int* unrank;
};
template <int MODE>
static __global__ __launch_bounds__(64) void k_cf_forward(int nf_all, int n_o, int n_chunks, const int* __restrict__ oe_cst /* [N][n_chunks + 1] */,
                                                          const int* __restrict__ oe_slot, const double* __restrict__ V,
                                                          const double* __restrict__ vy, const double* __restrict__ Mm,
                                                          double* __restrict__ Xt, long ldx, int stride, int seglen,
                                                          const double* __restrict__ Zt, double* __restrict__ R, int sep_stride, CfSegs sg) {
  constexpr bool SEP = MODE == 1;
  __shared__ double sM[kCfFwdCh * 36];
  __shared__ double sZ[kCfFwdCh * 36];
  __shared__ double sv[kCfFwdCh * 6 * 64];
  __shared__ double svy[kCfFwdCh * 6];
  const int lane = threadIdx.x;
  const int seg = SEP ? 0 : (MODE == 2 ? sg.fwork[2 * blockIdx.x] : (int)blockIdx.y);
  const int j = (MODE == 2 ? sg.fwork[2 * blockIdx.x + 1] : (int)blockIdx.x) * 64 + lane;
  const int ob = MODE == 2 ? sg.seg_start[seg] : 0;
  const int jr = MODE == 2 ? 9 * (sg.seg_start[seg + 1] - ob) : n_o;   // the right-hand side's column
  const bool on = j <= jr, rhs = j == jr, col = on && !rhs;
  const int oi = col ? j / 9 : 0, b = j - 9 * oi;
  const int o = (MODE == 2 && col) ? sg.seg_obj[ob + oi] : oi;
  const long tcol = MODE == 2 ? (rhs ? (long)n_o : 9L * (sg.rank ? sg.rank[o] : o) + b) : (long)j;   // this lane's column of T / R
  const int beg = SEP ? 0 : seg * stride;
  const int nf = SEP ? nf_all : ((beg + seglen < nf_all) ? beg + seglen : nf_all);
  if (beg >= nf) return;
  const bool use_z = !SEP && Zt != nullptr && seg >= 1;
  const int* cst = oe_cst + (size_t)o * (n_chunks + 1);
  double x[6] = {0, 0, 0, 0, 0, 0}, acc[6] = {0, 0, 0, 0, 0, 0};
  for (int i0 = beg; i0 < nf; i0 += kCfFwdCh) {
    const int ch = i0 / kCfFwdCh;
    const int len = (nf - i0 < kCfFwdCh) ? nf - i0 : kCfFwdCh;
    // (1) gather
    {   // (fixed trip counts with clamped addresses: all loads of the staging go out together)
      double tm[kCfFwdCh * 36 / 64], tz[kCfFwdCh * 36 / 64], ty[2];
#pragma unroll
      for (int q = 0; q < kCfFwdCh * 36 / 64; ++q) {
        const int idx = q * 64 + lane, cl = idx < len * 36 ? idx : 0;
        tm[q] = Mm[(size_t)i0 * 36 + cl];
        tz[q] = use_z ? Zt[(size_t)i0 * 36 + cl] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) { const int idx = q * 64 + lane; ty[q] = SEP ? 0.0 : vy[(size_t)i0 * 6 + (idx < len * 6 ? idx : 0)]; }
#pragma unroll
      for (int q = 0; q < kCfFwdCh * 36 / 64; ++q) { sM[q * 64 + lane] = tm[q]; sZ[q * 64 + lane] = tz[q]; }
#pragma unroll
      for (int q = 0; q < 2; ++q) if (q * 64 + lane < kCfFwdCh * 6) svy[q * 64 + lane] = ty[q];   // (used by the right-hand-side lane only)
    }
    if (SEP) {   // dense right-hand sides of the separators: rows 6 i0 .. of R
      // clamped addresses + selects, ALL 96 loads of the chunk in flight together: the grid is one wave per 64 columns (282 waves
      // at C4), so the bytes in flight per wave set the bandwidth -- a guarded load per element (96 dependent round trips per chunk)
      // and batches of 16 both took 2.2 ms per trial at C4's 625 separators
      const size_t jc = on ? (size_t)j : 0;
      {
        constexpr int kQ = kCfFwdCh * 6;
        double t[kQ];
#pragma unroll
        for (int q = 0; q < kQ; ++q) t[q] = R[jc + (size_t)(6 * i0 + (q < len * 6 ? q : 0)) * (size_t)ldx];
#pragma unroll
        for (int q = 0; q < kQ; ++q) sv[q * 64 + lane] = (on && q < len * 6) ? t[q] : 0.0;
      }
      __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll
      for (int q = 0; q < kCfFwdCh * 6; ++q) sv[q * 64 + lane] = 0.0;
      __builtin_amdgcn_wave_barrier();
      if (rhs) {
#pragma unroll
        for (int q = 0; q < kCfFwdCh * 6; ++q) sv[q * 64 + lane] = (q < len * 6) ? svy[q] : 0.0;
      } else if (col) {
        const int ka = cst[ch], kb = cst[ch + 1];
        for (int k = ka; k < kb; k += 4) {       // batches of four entries: slots and records of a batch in flight together
          int sl[4];
          double rec[4][6];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int kk = (k + e < kb) ? k + e : ka;
            sl[e] = oe_slot[kk];
            const double* src = V + ((size_t)kk * 9 + b) * 6;
#pragma unroll
            for (int a = 0; a < 6; ++a) rec[e][a] = src[a];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k + e < kb && sl[e] - i0 < len) {   // (slots behind the segment's end in this chunk: the separator's entries)
#pragma unroll
              for (int a = 0; a < 6; ++a) sv[((sl[e] - i0) * 6 + a) * 64 + lane] += rec[e][a];   // (a bbox and a 3-D edge of one camera add up)
            }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    // (2) the serial steps, LDS only
    for (int ii = 0; ii < len; ++ii) {
      const double* M = sM + ii * 36;
      double xn[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        double t = sv[(ii * 6 + a) * 64 + lane];
#pragma unroll
        for (int q = 0; q < 6; ++q) t -= M[a * 6 + q] * x[q];
        xn[a] = t;
      }
#pragma unroll
      for (int a = 0; a < 6; ++a) { x[a] = xn[a]; sv[(ii * 6 + a) * 64 + lane] = xn[a]; }
      if (use_z) {
        const double* Z = sZ + ii * 36;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int q = 0; q < 6; ++q) acc[a] += Z[q * 6 + a] * xn[q];
      }
    }
    // (3) out: batches of eight LDS reads, then their eight stores.  Straight-line code (fixed trip count, predicated stores): as a
    // run-time loop every batch waited for the previous batch's stores (the s_waitcnt vmcnt(0) at the loop head) -- hidden where
    // thousands of waves run, 18 us per chunk over the separators, where the grid is one wave per 64 columns
    if (on) {
#pragma unroll
      for (int q0 = 0; q0 < kCfFwdCh * 6; q0 += 8) {
        double t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = sv[((q0 + e < len * 6) ? q0 + e : 0) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (q0 + e < len * 6) {
            const int q = q0 + e;
            // separator k = i0 + q / 6 sits at slot k sep_stride + sep_stride - 1
            if (MODE == 2) {
              Xt[(size_t)sg.xoff[seg] + (size_t)(6 * (i0 - beg) + q) * (size_t)sg.xld[seg] + (size_t)j] = t[e];
            } else {
              const size_t row = (SEP && sep_stride > 0) ? (size_t)(6 * ((i0 + q / 6) * sep_stride + sep_stride - 1) + q % 6) : (size_t)(6 * i0 + q);
              Xt[(size_t)j + row * (size_t)ldx] = t[e];
            }
          }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (use_z && on) {
#pragma unroll
    for (int a = 0; a < 6; ++a) R[(size_t)tcol + (size_t)(6 * (seg - 1) + a) * (size_t)ldx] = acc[a];
  }
}

// ---- nested dissection of the camera chain (round 3) -------------------------------------------------------------------------------
// The chain kernels above are serial over the cameras (C4: 10k steps).  Every stride-th camera (slot i with i % stride == stride - 1)
// becomes a SEPARATOR; the segments between separators are independent chains (factor, forward and backward recurrences run one
// workgroup per segment), and the separators form a short block-tridiagonal system of their own.  With the interiors I ordered
// first and the separators S last, A = G G^T has
//     G = [ L_II   0   ]     L_II  block diagonal over the segments (the plain chain factor of each),
//         [ L_SI  L_SS ]     L_SI(s, j): for the LAST camera l of the segment in front of s the block G_l^T (what the chain computes
//                            anyway); for every camera j of the segment BEHIND s the block Z_j with Z_j^T =: Zt_j = -M_j Zt_{j-1},
//                            Zt_f = Li_f A(f, s) at the segment's first camera f (the forward recurrence applied to s's coupling),
//     L_SS = chol of the separators' Schur complement: diagonal A(s, s) - G_l^T G_l - sum_j Zt_j^T Zt_j, sub-diagonal
//            -G_e^T Zt_e (e = last camera of the segment between the two separators): block tridiagonal, factored by k_cf_chain.
// X = G^-1 [W | b_c] : interior rows segment by segment; separator rows X_S = L_SS^-1 (W_S - L_SI X_I) (k_cf_sep_rhs + the forward
// kernel on the separator chain).  T = D - X^T X does not care about the row order.  x_c = G^-T z : separators first (backward
// chain), then z_j -= Zt_j x_s(front) + [j last] G_j x_s(behind) and the per-segment backward recurrences.
// Serial length: stride - 1 + (F - 1) / stride steps instead of F - 1 (C4: 199 instead of 9,999; C3: 46 instead of 499).
constexpr int kCfMaxStride = 128;
// Zt of segment p >= 1 (workgroup p, lanes 0..5 = the six columns)
static __global__ __launch_bounds__(64) void k_cf_zt(int nf, int stride, const double* __restrict__ Linv, const double* __restrict__ Mm,
                                                     const double* __restrict__ B, double* __restrict__ Zt) {
  __shared__ double sM[kCfMaxStride * 36];
  const int p = blockIdx.x + 1, lane = threadIdx.x;
  const int beg = p * stride, end = (beg + stride - 1 < nf) ? beg + stride - 1 : nf;
  if (beg >= end) return;
  const int len = end - beg;
  {   // the segment's M blocks -> LDS (fixed trip count, clamped addresses: the loads go out together)
    constexpr int kQ = kCfMaxStride * 36 / 64;
    double t[kQ];
#pragma unroll
    for (int q = 0; q < kQ; ++q) { const int idx = q * 64 + lane; t[q] = Mm[(size_t)beg * 36 + (idx < len * 36 ? idx : 0)]; }
#pragma unroll
    for (int q = 0; q < kQ; ++q) sM[q * 64 + lane] = t[q];
  }
  __builtin_amdgcn_wave_barrier();
  const int c = lane < 6 ? lane : 5;
  double z[6];
  {   // Zt_f = Li_f A(f, s), A(f, s) = B[s] (rows f, columns s), s = f - 1
    const double* Li = Linv + (size_t)beg * 36;
    const double* Bs = B + (size_t)(beg - 1) * 36;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      double v = 0;
#pragma unroll
      for (int k = 0; k <= r; ++k) v += Li[r * 6 + k] * Bs[k * 6 + c];
      z[r] = v;
    }
  }
  for (int ii = 0; ii < len; ++ii) {
    if (ii > 0) {
      const double* M = sM + ii * 36;
      double zn[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        double v = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) v -= M[r * 6 + k] * z[k];
        zn[r] = v;
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) z[r] = zn[r];
    }
    if (lane < 6) {
#pragma unroll
      for (int r = 0; r < 6; ++r) Zt[(size_t)(beg + ii) * 36 + r * 6 + lane] = z[r];
    }
  }
}

// the separators' block-tridiagonal Schur complement: Hs[k] (diagonal, lambda included), Bs[k] = block (k + 1, k); thread = (k, entry)
static __global__ __launch_bounds__(256) void k_cf_sep_assemble(int nf, int stride, int n_sep, const double* __restrict__ Hcc, double lambda,
                                                                const double* __restrict__ Gfac, const double* __restrict__ Zt,
                                                                double* __restrict__ Hs, double* __restrict__ Bs) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int k = t / 36;
  if (k >= n_sep) return;
  const int e = t - k * 36, r = e / 6, c = e - 6 * r;
  const int s = k * stride + stride - 1, l = s - 1;
  double v = Hcc[(size_t)s * 36 + e] + ((r == c) ? lambda : 0.0);
#pragma unroll
  for (int q = 0; q < 6; ++q) v -= Gfac[(size_t)l * 36 + q * 6 + r] * Gfac[(size_t)l * 36 + q * 6 + c];
  const int beg = (k + 1) * stride, end = (beg + stride - 1 < nf) ? beg + stride - 1 : nf;   // the segment behind s
  for (int j = beg; j < end; ++j) {
#pragma unroll
    for (int q = 0; q < 6; ++q) v -= Zt[(size_t)j * 36 + q * 6 + r] * Zt[(size_t)j * 36 + q * 6 + c];
  }
  Hs[t] = v;
  double w = 0;
  if (k + 1 < n_sep) {   // block (k + 1, k) = -G_e^T Zt_e, e = last camera of that segment
    const int le = end - 1;
#pragma unroll
    for (int q = 0; q < 6; ++q) w -= Gfac[(size_t)le * 36 + q * 6 + r] * Zt[(size_t)le * 36 + q * 6 + c];
  }
  Bs[t] = w;
}

// right-hand sides of the separator rows of X: R(k) <- V_s - LiS_k (G_l^T X_l + acc), acc = what the forward kernel left in R(k);
// grid (column groups, separators), lane = column
static __global__ __launch_bounds__(64) void k_cf_sep_rhs(int nf, int n_o, int stride, int n_chunks, const int* __restrict__ oe_cst,
                                                          const int* __restrict__ oe_slot, const double* __restrict__ V,
                                                          const double* __restrict__ vy, const double* __restrict__ Gfac,
                                                          const double* __restrict__ LiS, const double* __restrict__ Xt, long ldx,
                                                          double* __restrict__ R, const int* __restrict__ cmap /* null: X dense in Xt */,
                                                          CfSegs sg) {
  const int j = blockIdx.x * 64 + threadIdx.x, k = blockIdx.y;
  if (j > n_o) return;
  const bool rhs = j == n_o;
  const int N1 = n_o / 9 + 1;
  const int s = k * stride + stride - 1, l = s - 1;
  double vs[6] = {0, 0, 0, 0, 0, 0};
  if (rhs) {
#pragma unroll
    for (int a = 0; a < 6; ++a) vs[a] = vy[(size_t)s * 6 + a];
  } else {
    const int tb = j / 9, b = j - 9 * tb;
    const int o = sg.unrank ? sg.unrank[tb] : tb;
    const int* cst = oe_cst + (size_t)o * (n_chunks + 1);
    const int ch = s / kCfFwdCh;
    for (int q = cst[ch]; q < cst[ch + 1]; ++q)
      if (oe_slot[q] == s) {
        const double* src = V + ((size_t)q * 9 + b) * 6;
#pragma unroll
        for (int a = 0; a < 6; ++a) vs[a] += src[a];
      }
  }
  bool has_acc = (k + 1) * stride < nf;   // a segment behind s exists (else R(k) was never written)
  double xl[6] = {0, 0, 0, 0, 0, 0};
  if (cmap) {   // X sparse: column j is zero over a segment unless its ellipsoid is in the segment's list
    const int tb2 = j / 9, bb = rhs ? 0 : j - 9 * tb2;
    const int ob = rhs ? N1 - 1 : (sg.unrank ? sg.unrank[tb2] : tb2);
    has_acc = has_acc && cmap[(size_t)(k + 1) * N1 + ob] >= 0;
    const int e = cmap[(size_t)k * N1 + ob];
    if (e >= 0) {
      const double* src = Xt + (size_t)sg.xoff[k] + (size_t)(9 * (e & 0xFFFFFF) + bb);
      const size_t ld = (size_t)sg.xld[k];
#pragma unroll
      for (int r = 0; r < 6; ++r) xl[r] = src[(size_t)(6 * (stride - 2) + r) * ld];
    }
  } else {
#pragma unroll
    for (int r = 0; r < 6; ++r) xl[r] = Xt[(size_t)j + (size_t)(6 * l + r) * (size_t)ldx];
  }
  double t[6];
#pragma unroll
  for (int a = 0; a < 6; ++a) t[a] = has_acc ? R[(size_t)j + (size_t)(6 * k + a) * (size_t)ldx] : 0.0;
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int r = 0; r < 6; ++r) t[a] += Gfac[(size_t)l * 36 + r * 6 + a] * xl[r];
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    double v = vs[a];
#pragma unroll
    for (int q = 0; q <= a; ++q) v -= LiS[(size_t)k * 36 + a * 6 + q] * t[q];
    R[(size_t)j + (size_t)(6 * k + a) * (size_t)ldx] = v;
  }
}

// back-substitution, interior right-hand sides once the separators' x is known: z_j -= Zt_j x_s(front) + [j last in segment] G_j x_s(behind)
static __global__ __launch_bounds__(256) void k_cf_back_prep(int nf, int stride, int n_sep, const double* __restrict__ Zt, const double* __restrict__ Gfac,
                                                             const double* __restrict__ xc, double* __restrict__ z) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int i = t / 6, a = t - 6 * i;
  if (i >= nf) return;
  const int p = i / stride, pos = i - p * stride;
  if (pos == stride - 1) return;   // a separator
  double v = 0;
  if (p >= 1) {
    const double* xs = xc + (size_t)(p * stride - 1) * 6;
#pragma unroll
    for (int q = 0; q < 6; ++q) v += Zt[(size_t)i * 36 + a * 6 + q] * xs[q];
  }
  if ((pos == stride - 2 || i == nf - 1) && p < n_sep) {
    const double* xs = xc + (size_t)(p * stride + stride - 1) * 6;
#pragma unroll
    for (int q = 0; q < 6; ++q) v += Gfac[(size_t)i * 36 + a * 6 + q] * xs[q];
  }
  z[t] -= v;
}

// ---- T <- D (block diagonal, lower triangle), row n_o <- b_o^T, zeros elsewhere ------------------------------------------------------------
// the whole of T in one launch (X dense: no memset in front): thread = entry (row, column) of the (n_o + 1) x n_o array
__device__ __forceinline__ void cf_T_init_full_one(const double* __restrict__ Hoo, const double* __restrict__ bo, double lambda,
                                                   double* __restrict__ T, long ldt, long n_o, long t) {
  const long col = t / ldt, row = t - col * ldt;
  if (col >= n_o) return;
  double v = 0;
  const long o = col / 9;
  const int c = (int)(col - 9 * o);
  if (row == n_o) v = bo[(size_t)o * 9 + c];
  else if (row < n_o && row / 9 == o && row >= col) {
    const int r = (int)(row - 9 * o);
    v = Hoo[(size_t)o * 45 + c * 9 - (c * (c - 1)) / 2 + (r - c)] + ((r == c) ? lambda : 0.0);
  }
  T[t] = v;
}
// V = Li W, vy = Li b_c AND the whole of T in one launch (X dense): blocks [0, nb_es) are k_cf_edge_scale's, the rest write T
static __global__ __launch_bounds__(256) void k_cf_edge_scale_T(DevGraph g, int nb_es, int n_list, const int* __restrict__ oe_u, const int* __restrict__ oe_slot,
                                                                const double* __restrict__ W, const double* __restrict__ Linv,
                                                                const double* __restrict__ bc, double* __restrict__ V, double* __restrict__ vy,
                                                                const double* __restrict__ Hoo, const double* __restrict__ bo, double lambda,
                                                                double* __restrict__ T, long ldt, long n_o) {
  if ((int)blockIdx.x < nb_es) cf_edge_scale_one(g, n_list, oe_u, oe_slot, W, Linv, bc, V, vy, (long)blockIdx.x * 256 + threadIdx.x);
  else cf_T_init_full_one(Hoo, bo, lambda, T, ldt, n_o, (long)(blockIdx.x - nb_es) * 256 + threadIdx.x);
}

// ---- z = y - X x_o: one workgroup per row of X (a contiguous row of Xt) -----------------------------------------------------------
static __global__ __launch_bounds__(256) void k_cf_z(const double* __restrict__ Xt, long ldx, int n_o, const double* __restrict__ xo,
                                                     double* __restrict__ z) {
  __shared__ double red[4];
  const double* row = Xt + (size_t)blockIdx.x * (size_t)ldx;
  double s = 0;
  for (int o = threadIdx.x; o < n_o; o += 256) s += row[o] * xo[o];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) z[blockIdx.x] = row[n_o] - (((red[0] + red[1]) + red[2]) + red[3]);
}

// ---- x_c = L^-T z: one wave, sequential from the last camera to the first; x_i = Li^T z_i - N_i x_{i+1} -------------------------
constexpr int kCfBackCh = 32;
// workgroup p: positions [p stride, p stride + seglen) of the chain (one segment = the plain chain).  map_stride > 0: position k
// of the chain is the separator at slot k map_stride + map_stride - 1 (z read from / x_c written to that slot).
static __global__ __launch_bounds__(64) void k_cf_tridiag_back(int nf_all, const double* __restrict__ Linv, const double* __restrict__ Nn,
                                                               const double* __restrict__ z, double* __restrict__ xc, int stride, int seglen,
                                                               int map_stride) {
  __shared__ double sLi[kCfBackCh * 36], sN[kCfBackCh * 36], sz[kCfBackCh * 6];
  const int lane = threadIdx.x;
  const int a = lane < 6 ? lane : 5;
  const int beg = (int)blockIdx.x * stride;
  const int nf = (beg + seglen < nf_all) ? beg + seglen : nf_all;
  if (beg >= nf) return;
  double xa = 0;                        // lanes 0..5: component a of x_{i+1}
  for (int hi = nf; hi > beg; hi -= kCfBackCh) {
    const int i0 = (hi - kCfBackCh > beg) ? hi - kCfBackCh : beg, len = hi - i0;
    __builtin_amdgcn_wave_barrier();
    {   // fixed trip counts with clamped addresses: all loads of the staging go out together (a run-time loop of load -> LDS write
        // pays one memory round trip per iteration)
      constexpr int kQ = kCfBackCh * 36 / 64, kZ = (kCfBackCh * 6 + 63) / 64;
      double tl[kQ], tn[kQ], tz[kZ];
#pragma unroll
      for (int q = 0; q < kQ; ++q) {
        const int idx = q * 64 + lane, cl = idx < len * 36 ? idx : 0;
        tl[q] = Linv[(size_t)i0 * 36 + cl]; tn[q] = Nn[(size_t)i0 * 36 + cl];
      }
#pragma unroll
      for (int q = 0; q < kZ; ++q) {
        const int idx = q * 64 + lane, cl = idx < len * 6 ? idx : 0;
        const size_t src = map_stride > 0 ? (size_t)((i0 + cl / 6) * map_stride + map_stride - 1) * 6 + cl % 6 : (size_t)i0 * 6 + cl;
        tz[q] = z[src];
      }
#pragma unroll
      for (int q = 0; q < kQ; ++q) { sLi[q * 64 + lane] = tl[q]; sN[q * 64 + lane] = tn[q]; }
#pragma unroll
      for (int q = 0; q < kZ; ++q) if (q * 64 + lane < kCfBackCh * 6) sz[q * 64 + lane] = tz[q];
    }
    __builtin_amdgcn_wave_barrier();
    for (int ii = len - 1; ii >= 0; --ii) {
      double xn[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) xn[k] = cf_readlane(xa, k);
      double s = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) s += sLi[ii * 36 + k * 6 + a] * sz[ii * 6 + k] - sN[ii * 36 + a * 6 + k] * xn[k];
      xa = s;
      const size_t dst = map_stride > 0 ? (size_t)((i0 + ii) * map_stride + map_stride - 1) * 6 : (size_t)(i0 + ii) * 6;
      if (lane < 6) xc[dst + lane] = s;
    }
  }
}

// ---- trial ellipsoids from x_o (the solution of the reduced ellipsoid system); scale partial of computeScale -------------------
__device__ __forceinline__ void cf_obj_update_one(const DevGraph& g, double lambda, const double* __restrict__ objs, const double* __restrict__ bo,
                                                  const double* __restrict__ xo, double* __restrict__ objs_trial, double* __restrict__ part, int o) {
  if (o >= g.n_objs) return;
  const Ell e = ell_load(objs + 10 * o);
  const bool active = (g.bb_start[o + 1] > g.bb_start[o]) || (g.e3_start[o + 1] > g.e3_start[o]) || g.gr_cnt[o] > 0;
  part[o * 4 + 3] = 1;
  if (!active) {   // never touched (sparse_optimizer.cpp:236-257)
    ell_store(e, objs_trial + 10 * o);
    part[o * 4 + 2] = 0;
    return;
  }
  double x[9], scale = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) { x[i] = xo[(size_t)o * 9 + i]; scale += x[i] * (lambda * x[i] + bo[(size_t)o * 9 + i]); }
  ell_store(ell_oplus(e, x), objs_trial + 10 * o);
  part[o * 4 + 2] = scale;
}

// ---- sparse interior rows of X (round 3) ---------------------------------------------------------------------------------------------
// Column o of X = G^-1 W is zero over segment p unless a camera INSIDE p sees ellipsoid o (the recurrence X_j = V_j - M_j X_{j-1}
// starts from zero at every segment), and non-zero only from the first such camera on.  A camera sees ~20 ellipsoids (the
// reference adds one bbox edge per detection, Optimizer.cpp:166-245), so over a 15-camera segment ~13 % of the C4 columns are
// live: the dense rank-6(F-1) update T -= X^T X spends 98 % of its 1.9e13 flops on zeros.  With the dissection stride fixed at
// kCfFwdCh = 16:
//   interior rows: compact slabs Xc_p [kCfSegRows][xld_p], columns = the segment's list (k_cf_forward<2>);
//   T(o1, o2) = D(o1) [o1 == o2] - sum over the segments p that list BOTH (bitmap AND), rows from the later first camera on, of
//               Xc_p(:, o1)^T Xc_p(:, o2)                                   k_cf_T_sparse: output stationary, one wave per 9 x 9 block,
//               v_mfma_f64_16x16x4 with 9 of the 16 rows / columns used; segments in ascending order -> bit-reproducible, no atomics;
//   separator rows (dense: the separator chain couples everything): Xs [6 n_sep][ldx], T -= Xs^T Xs on the MFMA update kernel
//               (k_chol_update_lds, external factor) -- 6 F / 16 rows instead of 6 F;
//   z = y - X x_o reads the slabs (k_cf_z_sparse).
// C4: 2.7e8 (block pair, camera) triples instead of 4e10; per LM trial 299 ms -> the dense separator update (K = 3,750) + this kernel.
typedef double cf_d4 __attribute__((ext_vector_type(4)));
constexpr int kCfTPer = 16;
constexpr int kCfTBatch = 8; // k-steps (of four rows) per batch of loads in k_cf_T_sparse   // ellipsoids o2 per workgroup of k_cf_T_sparse (4 waves x 4)
static __global__ __launch_bounds__(256) void k_cf_T_sparse(int N, int nw, const unsigned long long* __restrict__ mask, const int* __restrict__ cmap,
                                                            const long long* __restrict__ xoff, const int* __restrict__ xld,
                                                            const double* __restrict__ Xc, const double* __restrict__ Hoo,
                                                            const double* __restrict__ bo, double lambda, double* __restrict__ T, long ldt,
                                                            int o2_begin, int o2_end, const int* __restrict__ unrank) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t1 = blockIdx.x;                    // block row of T (its order: unrank, null = ellipsoid order); N = the right-hand side (row 9 N of T)
  const int o1 = (t1 == N || !unrank) ? t1 : unrank[t1];
  const int r = lane & 15, kq = lane >> 4;
  const int N1 = N + 1;
  const long n_o = 9L * N;
  for (int q = wave; q < kCfTPer; q += 4) {
    const int t2 = o2_begin + (int)blockIdx.y * kCfTPer + q;
    if (t2 >= o2_end || t2 > t1 || t2 >= N) continue;
    const int o2 = unrank ? unrank[t2] : t2;
    cf_d4 acc = {0, 0, 0, 0};
    for (int wd = 0; wd < nw; ++wd) {
      unsigned long long m = mask[(size_t)o1 * nw + wd] & mask[(size_t)o2 * nw + wd];
      if (!m) continue;
      // lane = bit: the column-map entries of all 64 segments of this word in one round trip
      const int pl = wd * 64 + lane;
      const bool mine = (m >> lane) & 1ull;
      const int e1l = mine ? cmap[(size_t)pl * N1 + o1] : 0, e2l = mine ? cmap[(size_t)pl * N1 + o2] : 0;
      const long long xol = mine ? xoff[pl] : 0;
      const int ldl = mine ? xld[pl] : 0;
      while (m) {
        const int bit = __builtin_ctzll(m);
        m &= m - 1;
        const int e1 = __builtin_amdgcn_readlane(e1l, bit), e2 = __builtin_amdgcn_readlane(e2l, bit);
        const int ld = __builtin_amdgcn_readlane(ldl, bit);
        const long long xo = ((long long)__builtin_amdgcn_readlane((int)(xol >> 32), bit) << 32) | (unsigned)__builtin_amdgcn_readlane((int)xol, bit);
        const int f1 = e1 >> 24, f2 = e2 >> 24, f = f1 > f2 ? f1 : f2;
        const double* pa = Xc + xo + 9 * (e1 & 0xFFFFFF) + r;
        const double* pb = Xc + xo + 9 * (e2 & 0xFFFFFF) + r;
        for (int k = (6 * f) & ~3; k < 6 * (kCfFwdCh - 1); k += 4 * kCfTBatch) {   // batches of k-steps: all loads of a batch in flight, then its MFMAs
          double av[kCfTBatch], bv[kCfTBatch];
#pragma unroll
          for (int u = 0; u < kCfTBatch; ++u) {
            int row = k + 4 * u + kq;
            row = row < kCfSegRows ? row : kCfSegRows - 1;              // (rows 90 .. 95 are zero)
            av[u] = pa[(long)row * ld];
            bv[u] = pb[(long)row * ld];
          }
#pragma unroll
          for (int u = 0; u < kCfTBatch; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
        }
      }
    }
    // D: acc[g] = C(i = kq + 4 g, j = r); block (o1, o2) of T (column major), rows 9 o1 + i, columns 9 o2 + j
    if (r < 9) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int i = kq + 4 * g;
        if (o1 == N) {
          if (i == 0) T[n_o + (9L * t2 + r) * ldt] = bo[(size_t)o2 * 9 + r] - acc[g];
        } else if (i < 9) {
          double base = 0;
          if (o1 == o2) {   // packed upper triangle of the symmetric block: (c, r) with c <= r at c * 9 - c (c - 1) / 2 + (r - c)
            const int lo = i < r ? i : r, hi = i < r ? r : i;
            base = Hoo[(size_t)o1 * 45 + lo * 9 - (lo * (lo - 1)) / 2 + (hi - lo)] + ((i == r) ? lambda : 0.0);
          }
          T[(9L * t1 + i) + (9L * t2 + r) * ldt] = base - acc[g];
        }
      }
    }
  }
}

// z = y - X x_o with the interior rows in the slabs: one workgroup per row (slot i, component a)
// xo_t: x_o in T's column order (= xo unless T's ellipsoids are permuted, CfSegs::rank): what a separator's dense row meets
static __global__ __launch_bounds__(256) void k_cf_z_sparse(int n_o, const double* __restrict__ Xs, long ldx, CfSegs sg, const double* __restrict__ Xc,
                                                            const double* __restrict__ xo, const double* __restrict__ xo_t, double* __restrict__ z) {
  __shared__ double red[4];
  const int t = blockIdx.x, i = t / 6, a = t - 6 * i;
  const int p = i / kCfFwdCh, pos = i - p * kCfFwdCh;
  double s = 0, y;
  if (pos == kCfFwdCh - 1) {   // separator p: a dense row
    const double* row = Xs + (size_t)(6 * p + a) * (size_t)ldx;
    for (int o = threadIdx.x; o < n_o; o += 256) s += row[o] * xo_t[o];
    y = row[n_o];
  } else {
    const int ob = sg.seg_start[p], m = 9 * (sg.seg_start[p + 1] - ob);
    const double* row = Xc + (size_t)sg.xoff[p] + (size_t)(6 * pos + a) * (size_t)sg.xld[p];
    for (int cc = threadIdx.x; cc < m; cc += 256) {
      const int ci = cc / 9;
      s += row[cc] * xo[(size_t)9 * sg.seg_obj[ob + ci] + (cc - 9 * ci)];
    }
    y = row[m];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) z[t] = y - (((red[0] + red[1]) + red[2]) + red[3]);
}

// ---- stored products (the default form of the sparse path) ---------------------------------------------------------------------------
// k_cf_T_sparse reads every slab column once per PARTNER column: ~450 GB of cache-line traffic at C4 (57 ms).  Blocking needs a
// dense tile, and the dense thing here is ONE segment's product P_p = Xc_p^T Xc_p over its compact columns:
//   k_cf_seg_syrk   one workgroup per 128 x 128 tile of one segment's product (work list of all (segment, tile) on and below the
//                   diagonal), four waves of 64 x 64, operands straight from the slab in MFMA layout (a slab is L2 resident while
//                   its tiles run), K = 96; the tile leaves as 9 x 9 BLOCKS (81 contiguous doubles per ellipsoid pair; column
//                   major inside) at boff[p] + i1 (i1 + 1) / 2 + i2, the right-hand side's row to Prhs;
//   k_cf_T_gather   T(o1, o2) = D [o1 == o2] - sum over the shared segments, ascending, of their stored block: 648 contiguous
//                   bytes per (pair, segment), every stored block read exactly once.
// C4: 14 GB of products written and read once, 3.7e11 flops of dense MFMA tiles.
static __global__ __launch_bounds__(256) void k_cf_seg_syrk(const int* __restrict__ twork, const int* __restrict__ seg_start,
                                                            const int* __restrict__ seg_first, const long long* __restrict__ xoff,
                                                            const int* __restrict__ xld, const double* __restrict__ Xc,
                                                            const long long* __restrict__ boff, double* __restrict__ P) {
  const int p = twork[3 * blockIdx.x], ti = twork[3 * blockIdx.x + 1], tj = twork[3 * blockIdx.x + 2];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, kq = lane >> 4;
  const int ob = seg_start[p], m = 9 * (seg_start[p + 1] - ob);   // the ellipsoids' compact columns (the right-hand side: k_cf_seg_rhs)
  const int c1b = ti * kCfSyT + (wave >> 1) * 64, c2b = tj * kCfSyT + (wave & 1) * 64;
  if (c1b + 63 < c2b || c1b >= m || c2b >= m) return;            // above the diagonal / past the last column
  const long ld = xld[p];
  const double* X = Xc + xoff[p];
  cf_d4 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int nj = 0; nj < 4; ++nj) acc[mi][nj] = cf_d4{0, 0, 0, 0};
  const double* pa = X + c1b + r;
  const double* pb = X + c2b + r;
  constexpr int kB = 3;   // k-steps per batch: the 24 loads of batch i + 1 are in flight while the 48 MFMAs of batch i issue
  constexpr int kNBatch = kCfSegRows / (4 * kB);
  static_assert(kNBatch * 4 * kB == kCfSegRows, "slab rows = whole batches");
  double a0[kB][4], b0[kB][4], a1[kB][4], b1[kB][4];
  auto load = [&](double (&a)[kB][4], double (&b)[kB][4], int it) {
    const int k = it * 4 * kB;
#pragma unroll
    for (int u = 0; u < kB; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        a[u][q] = pa[(long)(k + 4 * u + kq) * ld + 16 * q];
        b[u][q] = pb[(long)(k + 4 * u + kq) * ld + 16 * q];
      }
  };
  auto mac = [&](double (&a)[kB][4], double (&b)[kB][4]) {
#pragma unroll
    for (int u = 0; u < kB; ++u)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) acc[mi][nj] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[u][nj], a[u][mi], acc[mi][nj], 0, 0, 0);
  };
  // the columns are ordered by first camera: everything above row 6 first(c1b) is zero in all of this wave's c1 columns
  const int it0 = (6 * seg_first[ob + c1b / 9]) / (4 * kB);
  load(a0, b0, it0);
  for (int it = it0; it < kNBatch; it += 2) {
    if (it + 1 < kNBatch) load(a1, b1, it + 1);
    mac(a0, b0);
    if (it + 1 < kNBatch) {
      if (it + 2 < kNBatch) load(a0, b0, it + 2);
      mac(a1, b1);
    }
  }
  // The product is formed TRANSPOSED (the c2 fragment as the MFMA A operand): acc[mi][nj][g] = P(c1 = c1b + 16 mi + r, c2 = c2b + 16 nj +
  // kq + 4 g), so 16 lanes of a store walk down c1 = the contiguous direction of a stored block
  double* Pp = P + (size_t)boff[p] * 81;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int c1 = c1b + 16 * mi + r;
    if (c1 >= m) continue;
    const int i1 = c1 / 9, i = c1 - 9 * i1;
    const size_t row_off = (size_t)i1 * (i1 + 1) / 2;
#pragma unroll
    for (int nj = 0; nj < 4; ++nj)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c2 = c2b + 16 * nj + kq + 4 * g;
        if (c2 > c1) continue;
        const int i2 = c2 / 9, j = c2 - 9 * i2;
        Pp[(row_off + i2) * 81 + j * 9 + i] = acc[mi][nj][g];
      }
  }
}

// the right-hand side's row of every product: Prhs_p(c) = sum over the rows of Xc_p(row, c) Xc_p(row, m); workgroup = entry of the
// forward kernel's work list (segment, group of 64 compact columns), lane = column
static __global__ __launch_bounds__(64) void k_cf_seg_rhs(const int* __restrict__ fwork, const int* __restrict__ seg_start,
                                                          const long long* __restrict__ xoff, const int* __restrict__ xld,
                                                          const double* __restrict__ Xc, const long long* __restrict__ roff,
                                                          double* __restrict__ Prhs) {
  const int p = fwork[2 * blockIdx.x], c = fwork[2 * blockIdx.x + 1] * 64 + (int)threadIdx.x;
  const int m = 9 * (seg_start[p + 1] - seg_start[p]);
  const long ld = xld[p];
  const double* X = Xc + xoff[p];
  const int cc = c < m ? c : 0;
  double s = 0;
#pragma unroll 6
  for (int row = 0; row < 6 * (kCfFwdCh - 1); ++row) s += X[(long)row * ld + cc] * X[(long)row * ld + m];
  if (c < m) Prhs[roff[p] + c] = s;
}

static __global__ __launch_bounds__(256) void k_cf_T_gather(int N, int nw, const unsigned long long* __restrict__ mask, const int* __restrict__ cmap,
                                                            const long long* __restrict__ boff, const long long* __restrict__ roff,
                                                            const double* __restrict__ P, const double* __restrict__ Prhs,
                                                            const double* __restrict__ Hoo, const double* __restrict__ bo, double lambda,
                                                            double* __restrict__ T, long ldt, int o2_begin, int o2_end, const int* __restrict__ unrank,
                                                            int accumulate = 0) {
  // accumulate (round 6): T already holds -Xs^T Xs, written by the rank-K update in its assign form while the segments' products ran
  // beside it on a second stream (esl_slam.hip); (D - sum) + (-acc) is the same double as (D - sum) - acc
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t1 = blockIdx.x;                    // block row of T (its order: unrank, null = ellipsoid order); N = the right-hand side (row 9 N of T)
  const int o1 = (t1 == N || !unrank) ? t1 : unrank[t1];
  const int N1 = N + 1;
  const long n_o = 9L * N;
  const bool rhs = o1 == N;
  for (int q = wave; q < kCfTPer; q += 4) {
    const int t2 = o2_begin + (int)blockIdx.y * kCfTPer + q;
    if (t2 >= o2_end || t2 > t1 || t2 >= N) continue;
    const int o2 = unrank ? unrank[t2] : t2;
    double s0 = 0, s1 = 0;                      // entries lane and 64 + lane of the 81 (column major: entry = 9 j + i)
    const int eT0 = (lane % 9) * 9 + lane / 9, eT1 = lane < 17 ? ((64 + lane) % 9) * 9 + (64 + lane) / 9 : 0;   // the same entries of the transposed block
    for (int wd = 0; wd < nw; ++wd) {
      unsigned long long mk = mask[(size_t)o1 * nw + wd] & mask[(size_t)o2 * nw + wd];
      if (!mk) continue;
      // lane = bit: where this segment keeps the pair's block (one round trip for the word's 64 segments)
      // (the compact order of a segment is by first camera, not by ellipsoid: where o2 comes after o1 the stored block is the
      //  transpose of the one wanted)
      const int pl = wd * 64 + lane;
      long long offl = 0;
      int swl = 0;
      if ((mk >> lane) & 1ull) {
        const long long i2 = cmap[(size_t)pl * N1 + o2] & 0xFFFFFF;
        if (rhs) offl = roff[pl] + 9 * i2;
        else {
          const long long i1 = cmap[(size_t)pl * N1 + o1] & 0xFFFFFF;
          swl = i2 > i1;
          offl = (boff[pl] + (swl ? i2 * (i2 + 1) / 2 + i1 : i1 * (i1 + 1) / 2 + i2)) * 81;
        }
      }
      while (mk) {   // four segments' blocks in flight, added in ascending segment order
        long long off[4];
        bool have[4], sw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          have[u] = mk != 0;
          const int bit = have[u] ? __builtin_ctzll(mk) : 0;
          if (have[u]) mk &= mk - 1;
          off[u] = ((long long)__builtin_amdgcn_readlane((int)(offl >> 32), bit) << 32) | (unsigned)__builtin_amdgcn_readlane((int)offl, bit);
          sw[u] = __builtin_amdgcn_readlane(swl, bit) != 0;
        }
        double v0[4], v1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double* src = (rhs ? Prhs : P) + off[u];
          v0[u] = (have[u] && (!rhs || lane < 9)) ? src[sw[u] ? eT0 : lane] : 0.0;
          v1[u] = (have[u] && !rhs && lane < 17) ? src[sw[u] ? eT1 : 64 + lane] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { s0 += v0[u]; s1 += v1[u]; }
      }
    }
    if (rhs) {
      if (lane < 9) { double* tp = &T[n_o + (9L * t2 + lane) * ldt]; const double v = bo[(size_t)o2 * 9 + lane] - s0; *tp = accumulate ? v + *tp : v; }
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e = h * 64 + lane;
        if (e >= 81) continue;
        const int j = e / 9, i = e - 9 * j;
        if (o1 == o2 && i < j) continue;        // (above the diagonal: never stored, never read)
        double base = 0;
        if (o1 == o2) base = Hoo[(size_t)o1 * 45 + j * 9 - (j * (j - 1)) / 2 + (i - j)] + ((i == j) ? lambda : 0.0);
        double* tp = &T[(9L * t1 + i) + (9L * t2 + j) * ldt];
        const double v = base - (h ? s1 : s0);
        *tp = accumulate ? v + *tp : v;
      }
    }
  }
}


// x_o arrives from the solver in T's column order; everything else (trial states, the slabs' dot products, esl_lm_download) wants it
// by ellipsoid
static __global__ __launch_bounds__(256) void k_cf_xo_unpermute(int N, const int* __restrict__ rank, const double* __restrict__ xo_t,
                                                                double* __restrict__ xo) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= 9 * N) return;
  const int o = t / 9, i = t - 9 * o;
  xo[t] = xo_t[9 * rank[o] + i];
}

// ---- the camera-first form's index tables, generated on the device (round 4) ---------------------------------------------------
// The host ships only what it takes a sort to know (esl_slam.hip cf_ensure_impl): the per-ellipsoid lists in (slot, u) order and
// the (segment, ellipsoid) incidences with their rank inside the segment.  The dense tables -- N x (chunks + 1) entry ranges,
// segments x (N + 1) column map, the tile lists -- were 15 MB built by one host core at BASELINE configs[3]; here they are a few
// microseconds of scatter.
// oe_cst[o][chk] = first entry of ellipsoid o's list whose slot is >= chk * kCfFwdCh
static __global__ __launch_bounds__(256) void k_cf_make_cst(int N, int n_chunks, const int* __restrict__ oe_start, const int* __restrict__ oe_slot,
                                                            int* __restrict__ cst) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)N * (n_chunks + 1)) return;
  const int o = (int)(t / (n_chunks + 1)), chk = (int)(t - (long)o * (n_chunks + 1));
  int lo = oe_start[o], hi = oe_start[o + 1];
  const int want = chk * kCfFwdCh;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (oe_slot[mid] < want) lo = mid + 1; else hi = mid; }
  cst[t] = lo;
}
// incidence q = (segment p, ellipsoid o, first interior camera f, position `at` in the concatenated per-segment column lists):
// cmap / seg_obj / seg_first entries and the ellipsoid's segment bitmap (cmap preset to -1, mask to 0)
static __global__ __launch_bounds__(256) void k_cf_make_cmap(int n_inc, const int* __restrict__ inc /* [4][n_inc]: p, o, f, at */, int N1, int nw,
                                                             const int* __restrict__ seg_start, int* __restrict__ cmap, unsigned long long* __restrict__ mask,
                                                             int* __restrict__ seg_obj, int* __restrict__ seg_first) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= n_inc) return;
  const int p = inc[q], o = inc[n_inc + q], f = inc[2 * n_inc + q], at = inc[3 * n_inc + q];
  cmap[(long)p * N1 + o] = (f << 24) | (at - seg_start[p]);
  seg_obj[at] = o; seg_first[at] = f;
  atomicOr(&mask[(long)o * nw + p / 64], 1ull << (p & 63));
}
// per segment: the right-hand side's column (index = the segment's column count, "ellipsoid" N) and the work lists of the
// per-segment kernels -- (p, q) per 64 columns of the slab, (p, ti, tj) per tile of the product on and below the diagonal
static __global__ __launch_bounds__(64) void k_cf_make_work(int nseg, int N, int nw, const int* __restrict__ seg_start, const int* __restrict__ fw_off,
                                                            const long long* __restrict__ tw_off, int* __restrict__ cmap, unsigned long long* __restrict__ mask,
                                                            int* __restrict__ fwork, int* __restrict__ twork) {
  const int p = blockIdx.x;
  if (p >= nseg) return;
  const int cnt = seg_start[p + 1] - seg_start[p];
  if (threadIdx.x == 0) {
    cmap[(long)p * (N + 1) + N] = cnt;
    atomicOr(&mask[(long)N * nw + p / 64], 1ull << (p & 63));
  }
  const int nq = fw_off[p + 1] - fw_off[p];
  for (int q = threadIdx.x; q < nq; q += 64) { fwork[2 * (fw_off[p] + q)] = p; fwork[2 * (fw_off[p] + q) + 1] = q; }
  const int nt = (9 * cnt + kCfSyT - 1) / kCfSyT;
  const long long base = tw_off[p];
  for (int t = threadIdx.x; t < nt * (nt + 1) / 2; t += 64) {
    int ti = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while (ti * (ti + 1) / 2 > t) --ti;
    const int tj = t - ti * (ti + 1) / 2;
    twork[3 * (base + t)] = p; twork[3 * (base + t) + 1] = ti; twork[3 * (base + t) + 2] = tj;
  }
}
}  // namespace esl
