// esl_kernels_map.hpp — mapping-mode kernels (all cameras fixed: the shipped reference setting,
// src/core/Optimizer.cpp:126,135).  With cameras fixed the normal equations are block-diagonal
// 9x9 per ellipsoid, so ONE WAVEFRONT OWNS ONE ELLIPSOID: lanes stride over that ellipsoid's edges
// (edge arrays are sorted by ellipsoid, so the loads are coalesced), accumulate J^T W J / J^T W r in
// registers and finish with a wave-wide shuffle reduction.  No atomics, deterministic sums.
//
// Replaces: BlockSolver::buildSystem (Thirdparty/g2o/g2o/core/block_solver.hpp:502-560) =
// per-edge linearizeOplus (core/base_binary_edge.hpp:131-205) + constructQuadraticForm (:55-120);
// BlockSolver::solve + LinearSolverDense (block_solver.hpp:356-365, solvers/linear_solver_dense.h:65-113);
// SparseOptimizer::update / computeActiveErrors / activeRobustChi2 (core/sparse_optimizer.cpp:61-114,422-435).
#pragma once
#include "esl_ctx.hpp"

namespace esl {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;
}

// J^T W J (packed upper 45) and J^T W r accumulation for a D x 9 Jacobian with scalar weight
template <int D>
__device__ __forceinline__ void accum_obj(const double* J, const double* r, double w, double* acc /*45+9*/) {
#pragma unroll
  for (int k = 0; k < D; ++k) {
    const double wr = w * r[k];
    int p = 0;
#pragma unroll
    for (int a = 0; a < 9; ++a) {
      const double wa = w * J[k * 9 + a];
      acc[45 + a] -= J[k * 9 + a] * wr;
#pragma unroll
      for (int c = a; c < 9; ++c) acc[p++] += wa * J[k * 9 + c];
    }
  }
}

// numeric central-difference Jacobian wrt the ellipsoid (g2o base_binary_edge.hpp:147-197)
template <class F>
__device__ __forceinline__ void numeric_jac_obj(const Ell& e, double delta, int D, double* J, F&& eval) {
  const double scalar = 1.0 / (2 * delta);
#pragma unroll
  for (int d = 0; d < 9; ++d) {
    double u[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double rp[9], rm[9];
    u[d] = delta;
    eval(ell_oplus(e, u), rp);
    u[d] = -delta;
    eval(ell_oplus(e, u), rm);
    for (int k = 0; k < D; ++k) J[k * 9 + d] = scalar * (rp[k] - rm[k]);
  }
}

// ---- NaN pre-check of bbox edges (Optimizer.cpp:234-243) ------------------------------------------
static __global__ void k_bbox_validate(DevGraph g, const double* __restrict__ cams, const double* __restrict__ objs,
                                int* __restrict__ n_dropped) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.n_bbox) return;
  const SE3 T = se3_load(cams + 7 * g.bb_cam[i]);
  const Ell e = ell_load(objs + 10 * g.bb_obj[i]);
  double r[4];
  res_bbox(T, e, g.K, g.bb_meas + 4 * i, r);
  const double w = g.bb_w[i];
  const double c = w * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
  const bool bad = (c != c);
  g.bb_valid[i] = bad ? 0 : 1;
  if (bad) atomicAdd(n_dropped, 1);
}

// ---- linearise: one wave per ellipsoid ------------------------------------------------------------
// out: Hoo[o*45..] packed upper, bo[o*9..], part[o*4+0] = chi2, part[o*4+1] = max |H_kk|
template <int JAC>
static __global__ __launch_bounds__(kWave* kWavesPerBlock) void k_map_linearize(
    DevGraph g, const double* __restrict__ cams, const double* __restrict__ objs, double delta,
    double* __restrict__ Hoo, double* __restrict__ bo, double* __restrict__ part) {
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (o >= g.n_objs) return;
  const Ell e = ell_load(objs + 10 * o);
  double acc[54];
#pragma unroll
  for (int i = 0; i < 54; ++i) acc[i] = 0.0;
  double chi = 0.0;

  // bbox edges
  for (int i = g.bb_start[o] + lane; i < g.bb_start[o + 1]; i += 64) {
    if (!g.bb_valid[i]) continue;
    const SE3 T = se3_load(cams + 7 * g.bb_cam[i]);
    double meas[4] = {g.bb_meas[4 * i], g.bb_meas[4 * i + 1], g.bb_meas[4 * i + 2], g.bb_meas[4 * i + 3]};
    const double w = g.bb_w[i];
    double r[4], J[36];
    if (JAC == ESL_JAC_ANALYTIC) {
      jac_bbox(T, e, g.K, meas, r, J, nullptr);
    } else {
      res_bbox(T, e, g.K, meas, r);
      numeric_jac_obj(e, delta, 4, J, [&](const Ell& ep, double* out) { res_bbox(T, ep, g.K, meas, out); });
    }
    chi += w * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
    accum_obj<4>(J, r, w, acc);
  }
  // 3-D edges: J = [[Jp 6x6, 0],[0, I3]]  (the scale rows of the residual are s - s_k)
  for (int i = g.e3_start[o] + lane; i < g.e3_start[o + 1]; i += 64) {
    const SE3 T = se3_load(cams + 7 * g.e3_cam[i]);
    const Ell m = ell_load(g.e3_meas + 10 * i);
    const double w = g.e3_w[i];
    double r[9], J[81];
    if (JAC == ESL_JAC_ANALYTIC) {
      jac_e3d(T, e, m, g.yt, r, J, nullptr);
    } else {
      res_e3d(T, e, m, g.yt, r);
      numeric_jac_obj(e, delta, 9, J, [&](const Ell& ep, double* out) { res_e3d(T, ep, m, g.yt, out); });
    }
    double c = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) c += r[k] * r[k];
    chi += w * c;
    accum_obj<9>(J, r, w, acc);
  }
  // gravity prior (information = grav_w)
  if (lane < g.gr_cnt[o]) {
    double J[9], r[1];
    if (JAC == ESL_JAC_ANALYTIC) {
      r[0] = jac_grav(e, g.grav_n, J);
    } else {
      r[0] = res_grav(e, g.grav_n);
      numeric_jac_obj(e, delta, 1, J, [&](const Ell& ep, double* out) { out[0] = res_grav(ep, g.grav_n); });
    }
    chi += g.grav_w * r[0] * r[0];
    accum_obj<1>(J, r, g.grav_w, acc);
  }
  // wave reduction
#pragma unroll
  for (int i = 0; i < 54; ++i) acc[i] = wave_sum(acc[i]);
  chi = wave_sum(chi);
  if (lane == 0) {
    double md = 0;
    int p = 0;
#pragma unroll
    for (int a = 0; a < 9; ++a) {
      md = fmax(md, fabs(acc[p]));
      p += 9 - a;
    }
#pragma unroll
    for (int i = 0; i < 45; ++i) Hoo[(size_t)o * 45 + i] = acc[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) bo[(size_t)o * 9 + i] = acc[45 + i];
    part[o * 4 + 0] = chi;
    part[o * 4 + 1] = md;
  }
}

// chi2 of every edge hanging on ellipsoid `o` for state `e` (wave-cooperative; result valid in lane 0)
__device__ __forceinline__ double obj_chi2(const DevGraph& g, const double* __restrict__ cams, const Ell& e,
                                           int o, int lane) {
  double chi = 0.0;
  for (int i = g.bb_start[o] + lane; i < g.bb_start[o + 1]; i += 64) {
    if (!g.bb_valid[i]) continue;
    const SE3 T = se3_load(cams + 7 * g.bb_cam[i]);
    double r[4];
    res_bbox(T, e, g.K, g.bb_meas + 4 * i, r);
    chi += g.bb_w[i] * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
  }
  for (int i = g.e3_start[o] + lane; i < g.e3_start[o + 1]; i += 64) {
    const SE3 T = se3_load(cams + 7 * g.e3_cam[i]);
    const Ell m = ell_load(g.e3_meas + 10 * i);
    double r[9];
    res_e3d(T, e, m, g.yt, r);
    double c = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) c += r[k] * r[k];
    chi += g.e3_w[i] * c;
  }
  if (lane < g.gr_cnt[o]) {
    const double r = res_grav(e, g.grav_n);
    chi += g.grav_w * r * r;
  }
  return wave_sum(chi);
}

// ---- LM trial: solve 9x9, retract, re-evaluate chi2; one wave per ellipsoid ---------------------------
// part[o*4+0] = chi2(new), part[o*4+2] = sum x (lambda x + b), part[o*4+3] = pivots positive ? 1 : 0
static __global__ __launch_bounds__(kWave* kWavesPerBlock) void k_map_try_step(
    DevGraph g, const double* __restrict__ cams, const double* __restrict__ objs, double lambda,
    const double* __restrict__ Hoo, const double* __restrict__ bo, double* __restrict__ xo,
    double* __restrict__ objs_trial, double* __restrict__ part) {
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (o >= g.n_objs) return;
  const bool active = (g.bb_start[o + 1] > g.bb_start[o]) || (g.e3_start[o + 1] > g.e3_start[o]) || g.gr_cnt[o] > 0;
  const Ell e = ell_load(objs + 10 * o);
  if (!active) {  // inactive vertex: not in g2o's index mapping, never touched (sparse_optimizer.cpp:236-257)
    if (lane == 0) {
      ell_store(e, objs_trial + 10 * o);
      part[o * 4 + 0] = 0; part[o * 4 + 2] = 0; part[o * 4 + 3] = 1;
    }
    return;
  }
  double x[9];
  int ok = 1;
  double scale = 0;
  if (lane == 0) {
    double b[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) b[i] = bo[(size_t)o * 9 + i];
    ok = ldlt_solve_packed<9>(Hoo + (size_t)o * 45, lambda, b, x) ? 1 : 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      scale += x[i] * (lambda * x[i] + b[i]);
      xo[(size_t)o * 9 + i] = x[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) x[i] = __shfl(x[i], 0, 64);
  const Ell en = ell_oplus(e, x);
  const double chi = obj_chi2(g, cams, en, o, lane);
  if (lane == 0) {
    ell_store(en, objs_trial + 10 * o);
    part[o * 4 + 0] = chi;
    part[o * 4 + 2] = scale;
    part[o * 4 + 3] = (double)ok;
  }
}

// ---- deterministic reduction of the per-vertex partials: out = {sum chi2, max maxdiag, sum scale, min ok}
static __global__ __launch_bounds__(256) void k_reduce_parts(const double* __restrict__ part, int n, double* __restrict__ out,
                                                      int accumulate) {
  __shared__ double s0[256], s1[256], s2[256], s3[256];
  double a = 0, b = 0, c = 0, d = 1;
  for (int i = threadIdx.x; i < n; i += 256) {
    a += part[i * 4 + 0];
    b = fmax(b, part[i * 4 + 1]);
    c += part[i * 4 + 2];
    d = fmin(d, part[i * 4 + 3]);
  }
  s0[threadIdx.x] = a; s1[threadIdx.x] = b; s2[threadIdx.x] = c; s3[threadIdx.x] = d;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      s0[threadIdx.x] += s0[threadIdx.x + s];
      s1[threadIdx.x] = fmax(s1[threadIdx.x], s1[threadIdx.x + s]);
      s2[threadIdx.x] += s2[threadIdx.x + s];
      s3[threadIdx.x] = fmin(s3[threadIdx.x], s3[threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (accumulate) {
      out[0] += s0[0]; out[1] = fmax(out[1], s1[0]); out[2] += s2[0]; out[3] = fmin(out[3], s3[0]);
    } else {
      out[0] = s0[0]; out[1] = s1[0]; out[2] = s2[0]; out[3] = s3[0];
    }
  }
}

}  // namespace esl
