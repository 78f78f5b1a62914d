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
//     A = L L^T                         block-bidiagonal Cholesky, one sequential sweep over the cameras        k_cf_tridiag_factor
//     X = L^-1 W,  y = L^-1 b_c         forward substitution, one lane per column (9 N + 1 columns)             k_cf_edge_scale, k_cf_forward
//     T = D - X^T X,  t = b_o - X^T y   rank-6(F-1) update on the FP64 matrix cores (k_chol_update_lds, external factor)
//     T x_o = t                         dense FP64-MFMA Cholesky of order 9 N (esl_chol.hpp)
//     x_c = L^-T (y - X x_o)            k_cf_z, k_cf_tridiag_back
// Flops per LM trial: (9N)^2 6F + (9N)^3 / 3 instead of (6F)^3 / 3 -- C4: 2.1e13 instead of 7.2e13, C3: 6.4e8 instead of 8.9e9;
// memory: X 8.6 GB + T 2.6 GB instead of S 28.8 GB.  Same x up to rounding (it is the same linear system), so every parity
// test of SLAM mode applies unchanged.  Positive-definiteness ("ok" of LinearSolverDense, solvers/linear_solver_dense.h:107-112)
// = all pivots of A and of T positive.
//
// Applicable when every odometry edge joins two cameras whose free-camera slots differ by one (the reference's chain) and the run
// is not sharded; esl_lm_params::linear_solver selects (include/esl.h).
#pragma once
#include <climits>

#include "esl_kernels_map.hpp"

namespace esl {

__device__ __forceinline__ double cf_readlane(double v, int src) {   // src: compile-time constant after unrolling
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// ---- A = L L^T for the block-tridiagonal camera matrix: ONE wave, sequential over the free cameras -----------------------
// Lane (r, c) = lane / 6, lane % 6 owns entry (r, c) of every 6 x 6 block; the blocks travel through LDS (one wave: LDS
// operations execute in order, only the compiler is fenced).  Per camera i:
//   D_i = Hcc_i + lambda I - Lo Lo^T     (Lo = L_{i,i-1})        L_ii = chol(D_i),  Li = L_ii^-1
//   M_i = Li Lo                          forward substitution:   X_i = Li W_i - M_i X_{i-1}
//   B_i = A_{i+1,i} (odometry blocks),   Lo' = L_{i+1,i} = B_i Li^T
//   N_i = Li^T Lo'^T                     back substitution:      x_i = Li^T z_i - N_i x_{i+1}
// od_start / od_edge: odometry edges joining slots (i, i + 1), entry = edge * 2 + t; t = 1: the edge's first vertex is slot i
// (its Hij block, rows = first vertex, is then the TRANSPOSE of B_i).
static __global__ __launch_bounds__(64) void k_cf_tridiag_factor(int nf, const double* __restrict__ Hcc, const double* __restrict__ Aod,
                                                                 const int* __restrict__ od_start, const int* __restrict__ od_edge,
                                                                 double lambda, double* __restrict__ Linv, double* __restrict__ Mm,
                                                                 double* __restrict__ Nn, int* __restrict__ info) {
  __shared__ double sLo[36], sD[36], sL[36], sLi[36], sB[36], sIs[6];
  const int lane = threadIdx.x, r = lane / 6, c = lane - 6 * r;
  const bool on = lane < 36;
  if (on) { sLo[lane] = 0; sL[lane] = 0; sLi[lane] = 0; }
  bool ok = true;
  double h_next = (on && nf > 0) ? Hcc[lane] : 0.0;
  __builtin_amdgcn_wave_barrier();
  for (int i = 0; i < nf; ++i) {
    const double h = h_next;
    if (on && i + 1 < nf) h_next = Hcc[(size_t)(i + 1) * 36 + lane];   // next camera's block: in flight under this step
    double bsum = 0;                                                    // B_i entry (r, c), gathered early for the same reason
    if (on && i + 1 < nf) {
      for (int q = od_start[i]; q < od_start[i + 1]; ++q) {
        const int es = od_edge[q];
        const double* Hij = Aod + (size_t)(es >> 1) * 90 + 54;
        bsum += (es & 1) ? Hij[c * 6 + r] : Hij[r * 6 + c];
      }
    }
    // D_i
    if (on) {
      double d = h + ((r == c) ? lambda : 0.0);
#pragma unroll
      for (int k = 0; k < 6; ++k) d -= sLo[r * 6 + k] * sLo[c * 6 + k];
      sD[lane] = d;
    }
    __builtin_amdgcn_wave_barrier();
    // L_ii: right-looking, column by column
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const double p = sD[j * 6 + j];
      ok = ok && (p > 0);
      double is = __builtin_amdgcn_rsq(p);   // 1 / sqrt(p): hardware estimate + two Newton steps
      is = is * (1.5 - 0.5 * p * is * is);
      is = is * (1.5 - 0.5 * p * is * is);
      if (on && c == j && r >= j) sL[lane] = (r == j) ? p * is : sD[lane] * is;
      if (lane == 0) sIs[j] = is;
      __builtin_amdgcn_wave_barrier();
      if (on && r >= c && c > j) sD[lane] -= sL[r * 6 + j] * sL[c * 6 + j];
      __builtin_amdgcn_wave_barrier();
    }
    // Li = L_ii^-1 (lower): lane c < 6 solves column c by forward substitution
    if (lane < 6) {
      double col[6];
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) {
        double v = (rr == lane) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < rr; ++k) v -= sL[rr * 6 + k] * col[k];
        col[rr] = (rr >= lane) ? v * sIs[rr] : 0.0;
      }
#pragma unroll
      for (int rr = 0; rr < 6; ++rr) sLi[rr * 6 + lane] = col[rr];
    }
    __builtin_amdgcn_wave_barrier();
    if (on) {
      Linv[(size_t)i * 36 + lane] = sLi[lane];
      double m = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) m += sLi[r * 6 + k] * sLo[k * 6 + c];
      Mm[(size_t)i * 36 + lane] = m;
      sB[lane] = bsum;
    }
    __builtin_amdgcn_wave_barrier();
    double lo_new = 0;
    if (on && i + 1 < nf) {
#pragma unroll
      for (int k = 0; k < 6; ++k) lo_new += sB[r * 6 + k] * sLi[c * 6 + k];   // B Li^T
    }
    __builtin_amdgcn_wave_barrier();      // every lane has read the old Lo (M_i) before it is replaced
    if (on) sLo[lane] = lo_new;
    __builtin_amdgcn_wave_barrier();
    if (on) {
      double nv = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) nv += sLi[k * 6 + r] * sLo[c * 6 + k];      // Li^T Lo'^T
      Nn[(size_t)i * 36 + lane] = nv;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (!__all(ok) && lane == 0) atomicOr(info, 1);
}

// ---- V_e = Li_{slot(e)} W_e per edge (per-edge records [u][b][a], a = camera row), vy_s = Li_s b_c,s ------------------------
// thread = (edge u, ellipsoid column b) or, behind those, one per free camera.  Edges whose camera is fixed are never read;
// bbox edges dropped by the NaN / visibility pre-check get V = 0 (the per-ellipsoid lists still hold them).
static __global__ __launch_bounds__(256) void k_cf_edge_scale(DevGraph g, const double* __restrict__ W, const double* __restrict__ Linv,
                                                              const double* __restrict__ bc, double* __restrict__ V,
                                                              double* __restrict__ vy) {
  const long EU = (long)g.n_bbox + g.n_e3d;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t < EU * 9) {
    const long u = t / 9;
    const int b = (int)(t - u * 9);
    const int cam = (u < g.n_bbox) ? g.bb_cam[u] : g.e3_cam[u - g.n_bbox];
    const int obj = (u < g.n_bbox) ? g.bb_obj[u] : g.e3_obj[u - g.n_bbox];
    if (obj < 0) return;                               // slack slot of an appendable layout
    const int slot = g.cam_slot[cam];
    if (slot < 0) return;
    const bool valid = !(u < g.n_bbox) || g.bb_valid[u];
    double w[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) w[k] = valid ? W[(long)(k * 9 + b) * EU + u] : 0.0;
    const double* Li = Linv + (size_t)slot * 36;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      double s = 0;
#pragma unroll
      for (int k = 0; k <= a; ++k) s += Li[a * 6 + k] * w[k];
      V[u * 54 + b * 6 + a] = s;
    }
  } else {
    const long s = t - EU * 9;
    if (s >= g.n_free_cams) return;
    const double* Li = Linv + (size_t)s * 36;
    double w[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) w[k] = bc[(size_t)s * 6 + k];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      double v = 0;
#pragma unroll
      for (int k = 0; k <= a; ++k) v += Li[a * 6 + k] * w[k];
      vy[(size_t)s * 6 + a] = v;
    }
  }
}

// ---- X = L^-1 [W | b_c]: one lane per column (9 per ellipsoid + the right-hand side), sequential over the cameras -----------
// X_i = V_i - M_i X_{i-1}; V_i is non-zero only where the column's ellipsoid has an edge at camera i (its list is sorted by
// slot; the next edge's record is already in registers when its camera comes up).  The M blocks of kCfChunk cameras are staged
// in LDS by the wave (one wave per workgroup).  Xt is the TRANSPOSE of X: element (column j, row k) at Xt[j + k * ldx] --
// the layout the MFMA rank-K update wants for its factor, and coalesced for these stores.
constexpr int kCfChunk = 32;
static __global__ __launch_bounds__(64) void k_cf_forward(int nf, int n_o, const int* __restrict__ oe_start, const int* __restrict__ oe_u,
                                                          const int* __restrict__ oe_slot, const double* __restrict__ V,
                                                          const double* __restrict__ vy, const double* __restrict__ Mm,
                                                          double* __restrict__ Xt, long ldx) {
  __shared__ double sM[kCfChunk * 36];
  const int lane = threadIdx.x;
  const int j = blockIdx.x * 64 + lane;
  const bool on = j <= n_o, rhs = j == n_o;
  const int o = (on && !rhs) ? j / 9 : 0, b = j - 9 * o;
  int p = (on && !rhs) ? oe_start[o] : 0;
  const int pe = (on && !rhs) ? oe_start[o + 1] : 0;
  int nslot = rhs ? 0 : ((p < pe) ? oe_slot[p] : INT_MAX);
  double vn[6] = {0, 0, 0, 0, 0, 0};
  if (rhs && nf > 0) {
#pragma unroll
    for (int a = 0; a < 6; ++a) vn[a] = vy[a];
  } else if (p < pe) {
    const double* src = V + (size_t)oe_u[p] * 54 + b * 6;
#pragma unroll
    for (int a = 0; a < 6; ++a) vn[a] = src[a];
  }
  double x[6] = {0, 0, 0, 0, 0, 0};
  for (int i0 = 0; i0 < nf; i0 += kCfChunk) {
    const int len = (nf - i0 < kCfChunk) ? nf - i0 : kCfChunk;
    __builtin_amdgcn_wave_barrier();
    for (int idx = lane; idx < len * 36; idx += 64) sM[idx] = Mm[(size_t)i0 * 36 + idx];
    __builtin_amdgcn_wave_barrier();
    for (int ii = 0; ii < len; ++ii) {
      const int i = i0 + ii;
      double v[6] = {0, 0, 0, 0, 0, 0};
      while (nslot == i) {            // (a bbox and a 3-D edge of the same camera: two entries)
#pragma unroll
        for (int a = 0; a < 6; ++a) v[a] += vn[a];
        if (rhs) {
          nslot = (i + 1 < nf) ? i + 1 : INT_MAX;
          if (i + 1 < nf) {
#pragma unroll
            for (int a = 0; a < 6; ++a) vn[a] = vy[(size_t)(i + 1) * 6 + a];
          }
        } else {
          ++p;
          if (p < pe) {
            nslot = oe_slot[p];
            const double* src = V + (size_t)oe_u[p] * 54 + b * 6;
#pragma unroll
            for (int a = 0; a < 6; ++a) vn[a] = src[a];
          } else {
            nslot = INT_MAX;
          }
        }
      }
      const double* M = sM + ii * 36;
      double xn[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        double s = v[a];
#pragma unroll
        for (int k = 0; k < 6; ++k) s -= M[a * 6 + k] * x[k];
        xn[a] = s;
      }
      if (on) {
#pragma unroll
        for (int a = 0; a < 6; ++a) Xt[(size_t)j + (size_t)(6 * i + a) * (size_t)ldx] = xn[a];
      }
#pragma unroll
      for (int a = 0; a < 6; ++a) x[a] = xn[a];
    }
  }
}

// ---- T <- D (block diagonal, lower triangle) and row n_o <- b_o^T; the rest of T was cleared by a memset -----------------------
static __global__ __launch_bounds__(256) void k_cf_T_init(int n_objs, const double* __restrict__ Hoo, const double* __restrict__ bo,
                                                          double lambda, double* __restrict__ T, long ldt, long n_o) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long o = t / 90;
  if (o >= n_objs) return;
  const int e = (int)(t - o * 90);
  if (e < 81) {
    const int r = e / 9, c = e - 9 * r;
    if (r < c) return;
    // packed upper triangle of the symmetric block: (c, r) with c <= r at c * 9 - c (c - 1) / 2 + (r - c)
    const double v = Hoo[(size_t)o * 45 + c * 9 - (c * (c - 1)) / 2 + (r - c)] + ((r == c) ? lambda : 0.0);
    T[(9 * o + r) + (9 * o + c) * ldt] = v;
  } else {
    const int c = e - 81;
    T[n_o + (9 * o + c) * ldt] = bo[(size_t)o * 9 + c];
  }
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
static __global__ __launch_bounds__(64) void k_cf_tridiag_back(int nf, const double* __restrict__ Linv, const double* __restrict__ Nn,
                                                               const double* __restrict__ z, double* __restrict__ xc) {
  __shared__ double sLi[kCfChunk * 36], sN[kCfChunk * 36], sz[kCfChunk * 6];
  const int lane = threadIdx.x;
  const int a = lane < 6 ? lane : 5;
  double xa = 0;                        // lanes 0..5: component a of x_{i+1}
  for (int hi = nf; hi > 0; hi -= kCfChunk) {
    const int i0 = (hi - kCfChunk > 0) ? hi - kCfChunk : 0, len = hi - i0;
    __builtin_amdgcn_wave_barrier();
    for (int idx = lane; idx < len * 36; idx += 64) { sLi[idx] = Linv[(size_t)i0 * 36 + idx]; sN[idx] = Nn[(size_t)i0 * 36 + idx]; }
    for (int idx = lane; idx < len * 6; idx += 64) sz[idx] = z[(size_t)i0 * 6 + idx];
    __builtin_amdgcn_wave_barrier();
    for (int ii = len - 1; ii >= 0; --ii) {
      double xn[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) xn[k] = cf_readlane(xa, k);
      double s = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) s += sLi[ii * 36 + k * 6 + a] * sz[ii * 6 + k] - sN[ii * 36 + a * 6 + k] * xn[k];
      xa = s;
      if (lane < 6) xc[(size_t)(i0 + ii) * 6 + lane] = s;
    }
  }
}

// ---- trial ellipsoids from x_o (the solution of the reduced ellipsoid system); scale partial of computeScale -------------------
static __global__ __launch_bounds__(256) void k_cf_obj_update(DevGraph g, double lambda, const double* __restrict__ objs,
                                                              const double* __restrict__ bo, const double* __restrict__ xo,
                                                              double* __restrict__ objs_trial, double* __restrict__ part) {
  const int o = blockIdx.x * 256 + threadIdx.x;
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

}  // namespace esl
