// esl_kernels_slam.hpp — SLAM-mode kernels: free SE3 cameras + 9-DoF ellipsoids.
//
// This is the `bSLAM_mode = true` branch of the reference (src/core/Optimizer.cpp:126-158: camera 0
// fixed, odometry EdgeSE3Expmap between consecutive frames) solved the way g2o's BlockSolver does
// when the ellipsoids are marginalised (Thirdparty/g2o/g2o/core/block_solver.hpp:367-486):
//     S   = Hcc + lambda I - sum_o  W_o (Hoo + lambda I)^-1 W_o^T
//     b_s = b_c            - sum_o  W_o (Hoo + lambda I)^-1 b_o
//     x_c = S^-1 b_s ;  x_o = (Hoo + lambda I)^-1 (b_o - W_o^T x_c)
// which is the same solution as the dense LDLT of the whole system the shipped solver would compute.
//
// Work split:
//   k_slam_linearize_chunks  one wave per chunk (<= 64 bbox / <= 32 3-D edges of one ellipsoid): residuals, Jacobians wrt ellipsoid
//                      AND camera, per-edge W = Jc^T W Jo (6x9, SoA) and camera terms A = Jc^T W Jc, g = -Jc^T W r (records) to
//                      HBM, chunk partial of Hoo / b_o through the LDS transpose; k_slam_combine: Hoo, b_o, chi2 per ellipsoid
//   k_slam_odom        one lane per odometry edge
//   k_slam_cam_gather  one wave per free camera: Hcc, b_c from its edges (camera-side CSR; deterministic)
//   k_slam_prepare     one wave per ellipsoid: Dinv = (Hoo+lambda I)^-1, Y_e = W_e Dinv, b_s -= Y_e b_o
//   k_slam_schur_pull  one lane per block (c1, c2) of S: intersects the two cameras' edge lists, S[c1,c2] -= sum Y_e1 W_e2^T in
//                      list order (deterministic); the row camera's Y blocks, list and ellipsoid bitmap are staged in LDS per
//                      workgroup (round 1's scatter form with fp64 atomics was deleted in round 3)
//   dense Cholesky     esl_chol.hpp (FP64 MFMA)
//   k_slam_backsub     one wave per ellipsoid: x_o, retraction, trial states
//   k_slam_cam_update  one lane per camera: retraction exp(x_c) * Tcw
//   chi2 of the trial  k_slam_chi2_all (a wave per ellipsoid + a lane per odometry edge, one launch)
#pragma once
#include "esl_kernels_chunk.hpp"

namespace esl {

// numeric central-difference Jacobian wrt a camera (g2o base_binary_edge.hpp:147-170)
template <class F>
__device__ __forceinline__ void numeric_jac_cam(const SE3& T, double delta, int D, double* J, F&& eval) {
  const double scalar = 1.0 / (2 * delta);
#pragma unroll
  for (int d = 0; d < 6; ++d) {
    double u[6] = {0, 0, 0, 0, 0, 0};
    double rp[9], rm[9];
    u[d] = delta;
    eval(cam_oplus(T, u), rp);
    u[d] = -delta;
    eval(cam_oplus(T, u), rm);
    for (int k = 0; k < D; ++k) J[k * 6 + d] = scalar * (rp[k] - rm[k]);
  }
}

// the same for a 6-row residual with ONE loop body (six trips, the column placed by a select chain): the odometry edge's two
// Jacobians cost 24 fully inlined evaluations and 2,612 B of scratch per lane in the unrolled form (round 6)
template <class F>
__device__ __forceinline__ void numeric_jac_cam6_loop(const SE3& T, double delta, double* J, F&& eval) {
  const double scalar = 1.0 / (2 * delta);
  for (int d = 0; d < 6; ++d) {   // not unrolled
    double u[6], rp[6], rm[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) u[q] = (q == d) ? delta : 0.0;
    eval(cam_oplus(T, u), rp);
#pragma unroll
    for (int q = 0; q < 6; ++q) u[q] = (q == d) ? -delta : 0.0;
    eval(cam_oplus(T, u), rm);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
      for (int q = 0; q < 6; ++q) if (q == d) J[k * 6 + q] = scalar * (rp[k] - rm[k]);
    }
  }
}

// per-edge camera-side products: W = Jc^T w Jo written SoA (index k * EU + u); the camera terms A = Jc^T w Jc (21 packed) and
// g = -Jc^T w r (6) as a per-edge record ([u][27]: the camera gather reads a record as one coalesced row, lane = entry); Y and the
// pull kernel's copy of W are per-edge records ([u][54])
constexpr int kARec = 27;
template <int D>
__device__ __forceinline__ void store_cam_terms(const double* Jc, const double* Jo, const double* r, double w,
                                                double* __restrict__ W, double* __restrict__ A, long EU, long u) {
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 9; ++b) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < D; ++k) s += Jc[k * 6 + a] * w * Jo[k * 9 + b];
      W[(long)(a * 9 + b) * EU + u] = s;
    }
  int p = 0;
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int c = a; c < 6; ++c) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < D; ++k) s += Jc[k * 6 + a] * w * Jc[k * 6 + c];
      A[u * kARec + (p++)] = s;
    }
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    double s = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) s += Jc[k * 6 + a] * (w * r[k]);
    A[u * kARec + 21 + a] = -s;
  }
}

// The 3-D edge's Jacobians are structured: wrt the ellipsoid [[Jp 6x6, 0], [0, I3]], wrt the camera [[Jcp = Jp Ad 6x6], [0 3x6]] (the scale
// rows s - s_k do not see the camera).  W = Jc^T w Jo is therefore [Jcp^T w Jp | 0 (6 x 3)], A and g sum over the six pose rows only: the
// same values as store_cam_terms<9> on the padded 9 x 9 / 9 x 6 arrays (the dropped terms are exact zeros), from 36 + 36 + 9 live doubles
// instead of 81 + 54 + 9 (round 6: the padded form kept both arrays in scratch, 1,104 B per lane).
__device__ __forceinline__ void store_cam_terms_e3d(const double* Jcp, const double* Jp, const double* r, double w,
                                                    double* __restrict__ W, double* __restrict__ A, long EU, long u) {
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = 0; b < 9; ++b) {
      double s = 0;
      if (b < 6) {
#pragma unroll
        for (int k = 0; k < 6; ++k) s += Jcp[k * 6 + a] * w * Jp[k * 6 + b];
      }
      W[(long)(a * 9 + b) * EU + u] = s;
    }
  int p = 0;
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int c = a; c < 6; ++c) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) s += Jcp[k * 6 + a] * w * Jcp[k * 6 + c];
      A[u * kARec + (p++)] = s;
    }
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    double s = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) s += Jcp[k * 6 + a] * (w * r[k]);
    A[u * kARec + 21 + a] = -s;
  }
}
// 64-lane form of reduce_group_e3d_lds (one chunk per wave): the group's 18 entries from the structured Jacobian, then the LDS transpose
template <int G>
__device__ __forceinline__ void reduce_group_lds_e3d64(const double* Jp, const double* r, double w, int lane, double* __restrict__ out,
                                                       double* __restrict__ tr) {
  double v[18];
  fill_group_e3d<G>(v, Jp, r, w, std::make_integer_sequence<int, 18>{});
#pragma unroll
  for (int k = 0; k < 18; ++k) tr[k * kTrStride + lane] = v[k];
  __builtin_amdgcn_wave_barrier();
  const int k = lane % 18, p = lane / 18;
  double s0 = 0, s1 = 0;
  if (lane < 54) {
    const double* row = tr + k * kTrStride + p * 22;
    const int n = (p == 2) ? 20 : 22;
#pragma unroll
    for (int i = 0; i < 22; i += 2) {
      if (i < n) { s0 += row[i]; s1 += row[i + 1]; }
    }
  }
  double s = s0 + s1;
  const double sa = __shfl_down(s, 18, 64), sb = __shfl_down(s, 36, 64);
  __builtin_amdgcn_wave_barrier();
  if (lane < 18) out[G * 18 + lane] = (s + sa) + sb;
}

// One wave per CHUNK (<= 64 bbox edges or <= 32 3-D edges of one ellipsoid: the decomposition of mapping mode, built at upload),
// lane = edge: residual, Jacobians wrt ellipsoid AND camera, W / A of the edge to HBM, the 45 + 9 entries of the ellipsoid's
// J^T w J / -J^T w r summed over the wave through the LDS transpose of esl_kernels_chunk.hpp into a chunk partial (row of
// kChunkOut doubles: 45 packed H, 9 b, chi2).  One instantiation per edge type (very different register needs).
// Round 2's form ran one wave per ELLIPSOID over all its edges: 50 waves at C3 (186 us), its tail set by the ellipsoid with the
// most edges.
template <int JAC, int TYPE>
static __global__ __launch_bounds__(64 * kLinWaves) void k_slam_linearize_chunks(
    DevGraph g, ChunkTable ct, const int* __restrict__ ids, int n_ids, const double* __restrict__ cams, const double* __restrict__ objs,
    double delta, double* __restrict__ chunk_out, double* __restrict__ W, double* __restrict__ A) {
  __shared__ double tr_all[kLinWaves * kTrDoubles];
  double* tr = tr_all + (threadIdx.x >> 6) * kTrDoubles;
  const int lane = threadIdx.x & 63;
  const int seg = blockIdx.x * kLinWaves + (threadIdx.x >> 6);
  if (seg >= n_ids) return;
  const int ch = ids[seg];
  const int o = ct.obj[ch];
  const int i = ct.begin[ch] + lane;
  const bool in = i < ct.end[ch];
  const long EU = (long)g.n_bbox + g.n_e3d;
  const Ell e = ell_load(objs + 10 * o);
  double* out = chunk_out + (size_t)ch * kChunkOut;
  double chi = 0;
  if (TYPE == 0) {
    double r[4] = {0, 0, 0, 0}, Jo[36], w = 0;
#pragma unroll
    for (int k = 0; k < 36; ++k) Jo[k] = 0;
    if (JAC == ESL_JAC_NUMERIC) {
      // as the mapping-mode kernel (esl_kernels_chunk.hpp): the 18 states ell_oplus(e, +-delta e_d) are the same for every edge of the
      // chunk -- 18 lanes compute one each into the wave's LDS tile
      if (lane < 18) {
        double u[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) u[q] = (q == (lane >> 1)) ? ((lane & 1) ? -delta : delta) : 0.0;
        ell_store(ell_oplus(e, u), tr + 10 * lane);
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (in && g.bb_valid[i]) {
      const int ci = g.bb_cam[i];
      const SE3 T = se3_load(cams + 7 * ci);
      const bool cam_free = g.cam_slot[ci] >= 0;
      double meas[4] = {g.bb_meas[4 * i], g.bb_meas[4 * i + 1], g.bb_meas[4 * i + 2], g.bb_meas[4 * i + 3]};
      w = g.bb_w[i];
      double Jc[24];
      if (JAC == ESL_JAC_ANALYTIC) {
        // (Jc always, and through the entry without null tests: a select or a comparison of an array's address keeps it in scratch)
        jac_box_edge_both(g.bbox_mode, T, e, g.K, meas, r, Jo, Jc);
      } else {
        // one loop body per Jacobian, not 18 + 12 inlined evaluations (round 6); same perturbed states, same differences
        res_box_edge(g.bbox_mode, T, e, g.K, meas, r);
        const double scalar = 1.0 / (2 * delta);
        for (int d = 0; d < 9; ++d) {   // not unrolled
          double rp[4], rm[4];
          res_box_edge(g.bbox_mode, T, ell_load(tr + 20 * d), g.K, meas, rp);
          res_box_edge(g.bbox_mode, T, ell_load(tr + 20 * d + 10), g.K, meas, rm);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int q = 0; q < 9; ++q) if (q == d) Jo[k * 9 + q] = scalar * (rp[k] - rm[k]);
          }
        }
        if (cam_free) {
#pragma unroll
          for (int k = 0; k < 24; ++k) Jc[k] = 0;
          for (int d = 0; d < 6; ++d) {   // not unrolled
            double u[6], rp[4], rm[4];
#pragma unroll
            for (int q = 0; q < 6; ++q) u[q] = (q == d) ? delta : 0.0;
            res_box_edge(g.bbox_mode, cam_oplus(T, u), e, g.K, meas, rp);
#pragma unroll
            for (int q = 0; q < 6; ++q) u[q] = (q == d) ? -delta : 0.0;
            res_box_edge(g.bbox_mode, cam_oplus(T, u), e, g.K, meas, rm);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
              for (int q = 0; q < 6; ++q) if (q == d) Jc[k * 6 + q] = scalar * (rp[k] - rm[k]);
            }
          }
        }
      }
      chi = w * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
      if (cam_free) store_cam_terms<4>(Jc, Jo, r, w, W, A, EU, (long)i);
    }
    if (JAC == ESL_JAC_NUMERIC) __builtin_amdgcn_wave_barrier();   // the perturbed states have been read; the tile is reused
    reduce_group_lds<4, 0>(Jo, r, w, lane, out, tr);
    reduce_group_lds<4, 1>(Jo, r, w, lane, out, tr);
    reduce_group_lds<4, 2>(Jo, r, w, lane, out, tr);
  } else if (JAC == ESL_JAC_ANALYTIC) {   // structured Jacobians (see store_cam_terms_e3d)
    double r[9], Jp[36], w = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) r[k] = 0;
#pragma unroll
    for (int k = 0; k < 36; ++k) Jp[k] = 0;
    if (in) {
      const int ci = g.e3_cam[i];
      const SE3 T = se3_load(cams + 7 * ci);
      const bool cam_free = g.cam_slot[ci] >= 0;
      const Ell m = ell_load(g.e3_meas + 10 * i);
      w = g.e3_w[i];
      E3dHyp h;
      res_e3d(T, e, m, g.yt, r, &h);
      dlog_right_R(h.R, h.t, h.a, Jp);
      double cc = 0;
#pragma unroll
      for (int k = 0; k < 9; ++k) cc += r[k] * r[k];
      chi = w * cc;
      if (cam_free) {   // exp(d) Tcw == right perturbation of E by Ad((Tcw T_est)^-1) d  (jac_e3d)
        double Ad[36], Jcp[36];
        se3_adj(se3_inv(se3_mul(T, e.pose)), Ad);
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int b = 0; b < 6; ++b) {
            double sacc = 0;
#pragma unroll
            for (int k = 0; k < 6; ++k) sacc += Jp[a * 6 + k] * Ad[k * 6 + b];
            Jcp[a * 6 + b] = sacc;
          }
        store_cam_terms_e3d(Jcp, Jp, r, w, W, A, EU, (long)g.n_bbox + i);
      }
    }
    reduce_group_lds_e3d64<0>(Jp, r, w, lane, out, tr);
    reduce_group_lds_e3d64<1>(Jp, r, w, lane, out, tr);
    reduce_group_lds_e3d64<2>(Jp, r, w, lane, out, tr);
  } else {
    // g2o's central differences, in the structured form of the analytic branch and of the mapping-mode kernel (esl_kernels_chunk.hpp):
    // the six pose columns of the ellipsoid's Jacobian from E_0 exp(+-delta e_d) (the ellipsoid is perturbed on the right, so E_0 is
    // formed once), its scale block exactly the identity, the camera's six columns from exp(+-delta e_d) Tcw; both loops NOT
    // unrolled (one body, six trips; the column is placed by a select chain).  Round 6: the 30 fully inlined evaluations on padded
    // 9 x 9 / 9 x 6 arrays cost 6,596 B of scratch per lane and 0.40 ms per linearisation at C3 (analytic: 0.06).
    double r[9], Jp[36], w = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) r[k] = 0;
#pragma unroll
    for (int k = 0; k < 36; ++k) Jp[k] = 0;
    if (in) {
      const int ci = g.e3_cam[i];
      const SE3 T = se3_load(cams + 7 * ci);
      const bool cam_free = g.cam_slot[ci] >= 0;
      const Ell m = ell_load(g.e3_meas + 10 * i);
      w = g.e3_w[i];
      const SE3 E0 = e3d_E0(T, e, m);
      res_e3d_from_E0(E0, e.s, m.s, g.yt, r);
      const double scalar = 1.0 / (2 * delta);
      for (int d = 0; d < 6; ++d) {   // not unrolled
        double u[6], rp[9], rm[9];
#pragma unroll
        for (int q = 0; q < 6; ++q) u[q] = (q == d) ? delta : 0.0;
        res_e3d_from_E0(se3_mul(E0, se3_exp(u)), e.s, m.s, g.yt, rp);
#pragma unroll
        for (int q = 0; q < 6; ++q) u[q] = (q == d) ? -delta : 0.0;
        res_e3d_from_E0(se3_mul(E0, se3_exp(u)), e.s, m.s, g.yt, rm);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
#pragma unroll
          for (int q = 0; q < 6; ++q) if (q == d) Jp[k * 6 + q] = scalar * (rp[k] - rm[k]);
        }
      }
      double cc = 0;
#pragma unroll
      for (int k = 0; k < 9; ++k) cc += r[k] * r[k];
      chi = w * cc;
      if (cam_free) {
        double Jcp[36];
#pragma unroll
        for (int k = 0; k < 36; ++k) Jcp[k] = 0;
        for (int d = 0; d < 6; ++d) {   // not unrolled
          double u[6], rp[9], rm[9];
#pragma unroll
          for (int q = 0; q < 6; ++q) u[q] = (q == d) ? delta : 0.0;
          res_e3d_from_E0(e3d_E0(cam_oplus(T, u), e, m), e.s, m.s, g.yt, rp);
#pragma unroll
          for (int q = 0; q < 6; ++q) u[q] = (q == d) ? -delta : 0.0;
          res_e3d_from_E0(e3d_E0(cam_oplus(T, u), e, m), e.s, m.s, g.yt, rm);
#pragma unroll
          for (int k = 0; k < 6; ++k) {
#pragma unroll
            for (int q = 0; q < 6; ++q) if (q == d) Jcp[k * 6 + q] = scalar * (rp[k] - rm[k]);
          }
        }
        store_cam_terms_e3d(Jcp, Jp, r, w, W, A, EU, (long)g.n_bbox + i);
      }
    }
    reduce_group_lds_e3d64<0>(Jp, r, w, lane, out, tr);
    reduce_group_lds_e3d64<1>(Jp, r, w, lane, out, tr);
    reduce_group_lds_e3d64<2>(Jp, r, w, lane, out, tr);
  }
  chi = wave_sum(chi);
  if (lane == 0) out[54] = chi;
}

// Hoo (45 packed), b_o (9), chi2 and max |H_kk| of every ellipsoid from its chunk partials (chunk order: deterministic) + its
// gravity prior; one wave per ellipsoid, lane k < 55 = entry k
template <int JAC>
static __global__ __launch_bounds__(kWave* kWavesPerBlock) void k_slam_combine(DevGraph g, ChunkTable ct, const double* __restrict__ chunk_out,
                                                                         const double* __restrict__ objs, double delta,
                                                                         double* __restrict__ Hoo, double* __restrict__ bo,
                                                                         double* __restrict__ part) {
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (o >= g.n_objs) return;
  double v = 0;
  if (lane < 55)
    for (int ch = ct.ostart[o]; ch < ct.ostart[o + 1]; ++ch) v += chunk_out[(size_t)ch * kChunkOut + lane];
  const int ng = g.gr_cnt[o];
  if (ng > 0) {
    const Ell e = ell_load(objs + 10 * o);
    const double wg = g.grav_w * ng;
    double Jg[9], rg;
    if (JAC == ESL_JAC_ANALYTIC) rg = jac_grav(e, g.grav_n, Jg);
    else {
      rg = res_grav(e, g.grav_n);
      numeric_jac_obj(e, delta, 1, Jg, [&](const Ell& ep, double* o1) { o1[0] = res_grav(ep, g.grav_n); });
    }
    if (lane < 45) {
      double ja = 0, jc = 0;
#pragma unroll
      for (int q = 0; q < 9; ++q) { if (q == tri_a(lane)) ja = Jg[q]; if (q == tri_c(lane)) jc = Jg[q]; }
      v += (wg * ja) * jc;
    } else if (lane < 54) {
      double ja = 0;
#pragma unroll
      for (int q = 0; q < 9; ++q) if (q == lane - 45) ja = Jg[q];
      v -= ja * (wg * rg);
    } else if (lane == 54) {
      v += wg * rg * rg;
    }
  }
  if (lane < 45) Hoo[(size_t)o * 45 + lane] = v;
  else if (lane < 54) bo[(size_t)o * 9 + (lane - 45)] = v;
  bool diag = false;
#pragma unroll
  for (int a = 0, q = 0; a < 9; ++a) { if (lane == q) diag = true; q += 9 - a; }
  const double md = wave_max(diag ? fabs(v) : 0.0);
  const double chi = __shfl(v, 54, 64);
  if (lane == 0) { part[o * 4 + 0] = chi; part[o * 4 + 1] = md; part[o * 4 + 2] = 0; part[o * 4 + 3] = 1; }
}

// odometry edges: Aod[e*90 ..] = Hii(21) bi(6) Hjj(21) bj(6) Hij(36, row-major i x j); od_part[e] = chi2
// (launched with 128 threads: without the bound the compiler sizes the register file for 1,024 threads per workgroup -- 128 VGPRs --
//  and the analytic instantiation spilt 540 B per lane, the numeric one 4.3 KB)
template <int JAC>
static __global__ __launch_bounds__(128) void k_slam_odom(DevGraph g, const double* __restrict__ cams, double delta, double* __restrict__ Aod,
                            double* __restrict__ od_chi) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= g.n_odom) return;
  const int vi = g.od_i[e], vj = g.od_j[e];
  const bool fi = g.cam_slot[vi] >= 0, fj = g.cam_slot[vj] >= 0;
  double* out = Aod + (size_t)e * 90;
  if (!fi && !fj) {  // inactive edge (all vertices fixed)
    od_chi[e] = 0;
    for (int k = 0; k < 90; ++k) out[k] = 0;
    return;
  }
  const SE3 Ti = se3_load(cams + 7 * vi), Tj = se3_load(cams + 7 * vj), Z = se3_load(g.od_meas + 7 * e);
  double r[6], Ji[36], Jj[36];
  if (JAC == ESL_JAC_ANALYTIC) {
    jac_odom(Ti, Tj, Z, r, Ji, Jj);
  } else {
    res_odom(Ti, Tj, Z, r);
#pragma unroll
    for (int k = 0; k < 36; ++k) { Ji[k] = 0; Jj[k] = 0; }
    numeric_jac_cam6_loop(Ti, delta, Ji, [&](const SE3& Tp, double* o6) { res_odom(Tp, Tj, Z, o6); });
    numeric_jac_cam6_loop(Tj, delta, Jj, [&](const SE3& Tp, double* o6) { res_odom(Ti, Tp, Z, o6); });
  }
  double w[6], chi = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) { w[k] = g.od_info[6 * e + k]; chi += r[k] * w[k] * r[k]; }
  od_chi[e] = chi;
  int p = 0;
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int c = a; c < 6; ++c) {
      double s = 0, t = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) { s += Ji[k * 6 + a] * w[k] * Ji[k * 6 + c]; t += Jj[k * 6 + a] * w[k] * Jj[k * 6 + c]; }
      out[p] = fi ? s : 0.0;
      out[27 + p] = fj ? t : 0.0;
      ++p;
    }
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    double s = 0, t = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) { s += Ji[k * 6 + a] * w[k] * r[k]; t += Jj[k * 6 + a] * w[k] * r[k]; }
    out[21 + a] = fi ? -s : 0.0;
    out[48 + a] = fj ? -t : 0.0;
  }
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) s += Ji[k * 6 + a] * w[k] * Jj[k * 6 + c];
      out[54 + a * 6 + c] = (fi && fj) ? s : 0.0;
    }
}

// one WAVE per camera, lane k < 27 = entry k of {A (21 packed), g (6)}: the camera's edges are walked uniformly and every edge's
// record is one coalesced row (round 2: one lane per camera over SoA planes, 64 us at C3 for 8 waves of dependent loads);
// fixed edge order (bbox edges, 3-D edges, odometry) -> deterministic.  Hcc (36 full, row-major), bc (6); cam_part = {0, max diag, 0, 1}
static __global__ __launch_bounds__(kWave* kWavesPerBlock) void k_slam_cam_gather(DevGraph g, const double* __restrict__ A,
                                                                            const double* __restrict__ Aod, double* __restrict__ Hcc,
                                                                            double* __restrict__ bc, double* __restrict__ cam_part) {
  const int lane = threadIdx.x & 63;
  const int cidx = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (cidx >= g.n_cams) return;
  const int slot = g.cam_slot[cidx];
  if (slot < 0) {
    if (lane == 0) { cam_part[cidx * 4 + 0] = 0; cam_part[cidx * 4 + 1] = 0; cam_part[cidx * 4 + 2] = 0; cam_part[cidx * 4 + 3] = 1; }
    return;
  }
  const int k = lane < kARec ? lane : 0;
  double acc = 0;
  for (int q = g.cbb_start[cidx]; q < g.cbb_start[cidx + 1]; ++q) {
    const int e = g.cbb_edge[q];
    if (!g.bb_valid[e]) continue;
    acc += A[(long)e * kARec + k];
  }
  for (int q = g.ce3_start[cidx]; q < g.ce3_start[cidx + 1]; ++q) acc += A[((long)g.n_bbox + g.ce3_edge[q]) * kARec + k];
  if (g.shard_rank == 0)
    for (int q = g.cod_start[cidx]; q < g.cod_start[cidx + 1]; ++q) {
      const int es = g.cod_edge[q];
      acc += Aod[(size_t)(es >> 1) * 90 + ((es & 1) ? 27 : 0) + k];
    }
  // packed (a, c), a <= c, index p: row a starts at a (13 - a) / 2
  if (lane < 21) {
    int a = 0, base = 0;
    while (lane >= base + (6 - a)) { base += 6 - a; ++a; }
    const int cc = a + (lane - base);
    Hcc[(size_t)slot * 36 + a * 6 + cc] = acc;
    Hcc[(size_t)slot * 36 + cc * 6 + a] = acc;
  } else if (lane < 27) {
    bc[(size_t)slot * 6 + (lane - 21)] = acc;
  }
  bool diag = false;
#pragma unroll
  for (int a = 0, q = 0; a < 6; ++a) { if (lane == q) diag = true; q += 6 - a; }
  const double md = wave_max(diag ? fabs(acc) : 0.0);
  if (lane == 0) { cam_part[cidx * 4 + 0] = 0; cam_part[cidx * 4 + 1] = md; cam_part[cidx * 4 + 2] = 0; cam_part[cidx * 4 + 3] = 1; }
}

// max |Hcc_kk| per camera from the (possibly rank-summed) blocks
static __global__ void k_slam_cam_maxdiag(DevGraph g, const double* __restrict__ Hcc, double* __restrict__ cam_part) {
  const int cidx = blockIdx.x * blockDim.x + threadIdx.x;
  if (cidx >= g.n_cams) return;
  const int slot = g.cam_slot[cidx];
  double md = 0;
  if (slot >= 0)
    for (int a = 0; a < 6; ++a) md = fmax(md, fabs(Hcc[(size_t)slot * 36 + a * 6 + a]));
  cam_part[cidx * 4 + 1] = md;
}

// S <- 0 is done by memset; this adds the camera blocks: diagonal Hcc + lambda I, b_c into row n,
// odometry off-diagonal blocks into the LOWER triangle.
static __global__ void k_slam_S_init(DevGraph g, const double* __restrict__ Hcc, const double* __restrict__ bc,
                              const double* __restrict__ Aod, double lambda, double* __restrict__ S, long lda, long n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int nf = g.n_free_cams;
  if (g.shard_rank != 0) return;   // Hcc / b_c are rank-summed beforehand; they enter the summed S once
  if (t < nf) {
    for (int a = 0; a < 6; ++a) {
      for (int c = 0; c < 6; ++c)
        S[(long)(6 * t + a) + (long)(6 * t + c) * lda] = Hcc[(size_t)t * 36 + a * 6 + c] + ((a == c) ? lambda : 0.0);
      S[n + (long)(6 * t + a) * lda] = bc[(size_t)t * 6 + a];
    }
  } else if (t < nf + g.n_odom) {
    const int e = t - nf;
    const int si = g.cam_slot[g.od_i[e]], sj = g.cam_slot[g.od_j[e]];
    if (si < 0 || sj < 0) return;
    const double* Hij = Aod + (size_t)e * 90 + 54;  // 6x6 row-major: rows = vertex i, cols = vertex j
    for (int a = 0; a < 6; ++a)
      for (int c = 0; c < 6; ++c) {
        const double v = Hij[a * 6 + c];
        // block (si, sj) of H; store into the lower triangle
        if (si > sj) atomicAdd(&S[(long)(6 * si + a) + (long)(6 * sj + c) * lda], v);
        else atomicAdd(&S[(long)(6 * sj + c) + (long)(6 * si + a) * lda], v);
      }
  }
}

// one wave per ellipsoid: Dinv = (Hoo + lambda I)^-1 (lanes 0..8 each solve one unit vector),
// then Y_e = W_e Dinv for its free-camera edges and b_s[cam] -= Y_e b_o.  part[o*4+3] = pivots ok.
static __global__ __launch_bounds__(kWave* kWavesPerBlock) void k_slam_prepare(
    DevGraph g, double lambda, const double* __restrict__ Hoo, const double* __restrict__ bo,
    const double* __restrict__ W, double* __restrict__ Y, double* __restrict__ Dinv, double* __restrict__ part,
    double* __restrict__ Tb /* [6][EU]: Y_e b_o, the edge's share of b_s */,
    double* __restrict__ Wt /* [EU][9][6]: copy of W as per-edge records for k_slam_schur_pull */) {
  __shared__ double sD[kWavesPerBlock][81];
  __shared__ double sb[kWavesPerBlock][9];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int o = blockIdx.x * kWavesPerBlock + wv;
  if (o >= g.n_objs) return;
  const long EU = (long)g.n_bbox + g.n_e3d;
  int ok = 1;
  if (lane < 9) {
    double e[9], x[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) e[i] = (i == lane) ? 1.0 : 0.0;
    ok = ldlt_solve_packed<9>(Hoo + (size_t)o * 45, lambda, e, x) ? 1 : 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) { sD[wv][i * 9 + lane] = x[i]; Dinv[(size_t)o * 81 + i * 9 + lane] = x[i]; }
    sb[wv][lane] = bo[(size_t)o * 9 + lane];
  }
  ok = __all(ok);
  if (lane == 0) part[o * 4 + 3] = (double)ok;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  for (int q = g.ue_start[o] + lane; q < g.ue_start[o + 1]; q += 64) {
    const long u = g.ue_id[q];
    if (u < g.n_bbox && !g.bb_valid[u]) continue;
    double t[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      double wrow[9];
#pragma unroll
      for (int b = 0; b < 9; ++b) wrow[b] = W[(long)(a * 9 + b) * EU + u];
#pragma unroll
      for (int b = 0; b < 9; ++b) Wt[u * 54 + b * 6 + a] = wrow[b];
#pragma unroll
      for (int b = 0; b < 9; ++b) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) s += wrow[k] * sD[wv][k * 9 + b];
        Y[u * 54 + a * 9 + b] = s;
        t[a] += s * sb[wv][b];
      }
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) Tb[(long)a * EU + u] = t[a];
  }
}

// Schur complement, output-owner form (round 2; deterministic): one LANE per lower block (sp, sq) of S.  The lane intersects
// the two cameras' edge lists (both sorted by ellipsoid), and for every ellipsoid both cameras observe subtracts
// Y_e1 W_e2^T over the edge pairs (e1 of camera sp, e2 of camera sq) in list order -- the sum the scatter form below builds
// with fp64 atomics in whatever order the hardware serves them.  The diagonal lane (s, s) also folds its camera's share of
// b_s (sum over its edges of Y_e b_o, stored per edge by k_slam_prepare).  1-D grid: workgroup b = (row camera sp = b / groups,
// column group gx = b % groups), sq = gx * 64 + lane.
constexpr int kSchurYStride = 55;   // 54 doubles of one edge's Y block + 1 of padding (lanes read different edges: no bank conflicts)
// dynamic LDS of the staged form: the row camera's Y blocks, its list (ellipsoid, edge id) and a bitmap of its ellipsoids
inline size_t schur_pull_lds_bytes(int cu_max, int n_objs) {
  return (size_t)cu_max * kSchurYStride * sizeof(double) + (size_t)cu_max * 2 * sizeof(int) + (size_t)((n_objs + 31) / 32) * sizeof(unsigned);
}
// one (row edge e1, column edge e2) contribution: out += Y_e1 W_e2^T
#define ESL_SCHUR_ACCUM(yp, uq)                                                     \
  _Pragma("unroll") for (int b = 0; b < 9; ++b) {                                   \
    double wq[6];                                                                   \
    _Pragma("unroll") for (int c = 0; c < 6; ++c) wq[c] = Wt[(uq) * 54 + b * 6 + c]; \
    _Pragma("unroll") for (int a = 0; a < 6; ++a)                                   \
      _Pragma("unroll") for (int c = 0; c < 6; ++c) out[a * 6 + c] += yp[a * 9 + b] * wq[c]; \
  }
template <bool YLDS>
static __global__ __launch_bounds__(64) void k_slam_schur_pull(DevGraph g, const double* __restrict__ Wt, const double* __restrict__ Y,
                                                        const double* __restrict__ Tb, double* __restrict__ S, long lda, long n, unsigned groups) {
  extern __shared__ double ysh[];   // YLDS: [edge of the row camera][kSchurYStride], then robj / rid / rbits
  const int sp = (int)(blockIdx.x / groups), gx = (int)(blockIdx.x - (unsigned)sp * groups);
  const int sq = gx * 64 + (int)threadIdx.x;
  if (gx * 64 > sp) return;   // the whole workgroup lies right of the diagonal
  const long EU = (long)g.n_bbox + g.n_e3d;
  const int i0 = g.cu_start[sp];
  int i = i0, j = (sq <= sp) ? g.cu_start[sq] : 0;
  const int ie = g.cu_start[sp + 1], je = (sq <= sp) ? g.cu_start[sq + 1] : 0;
  double out[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) out[k] = 0;
  bool any = false;
  if (YLDS) {
    // Staged form.  Every lane of the workgroup intersects ITS camera's list with the same row camera's list: the row
    // camera's Y blocks (54 scattered 8-byte loads per edge), its list and a bitmap of its ellipsoids are put in LDS once.
    // A lane then walks only its own list (one load per entry instead of two dependent loads per merge step), tests the
    // bitmap, and on a hit finds the row camera's run of that ellipsoid by binary search in LDS.  Same ellipsoid order, same
    // (row edge, column edge) order inside a run as the merge below: bit-identical S.
    const int ne = ie - i0, nw = (g.n_objs + 31) / 32;
    int* robj = (int*)(ysh + (size_t)g.cu_max * kSchurYStride);
    int* rid = robj + g.cu_max;
    unsigned* rbits = (unsigned*)(rid + g.cu_max);
    for (int w = threadIdx.x; w < nw; w += 64) rbits[w] = 0u;
    for (int e = threadIdx.x; e < ne; e += 64) { robj[e] = g.cu_obj[i0 + e]; rid[e] = g.cu_id[i0 + e]; }
    __syncthreads();
    for (int e = threadIdx.x; e < ne; e += 64) atomicOr(&rbits[robj[e] >> 5], 1u << (robj[e] & 31));
    for (int idx = threadIdx.x; idx < ne * 54; idx += 64) {
      const int e = idx / 54, k = idx - e * 54;
      ysh[e * kSchurYStride + k] = Y[(long)rid[e] * 54 + k];
    }
    __syncthreads();
    if (sq > sp) return;
    int oj = (j < je) ? g.cu_obj[j] : -1;
    while (j < je) {
      int j1 = j + 1;
      int on = (j1 < je) ? g.cu_obj[j1] : -1;
      while (on == oj) { ++j1; on = (j1 < je) ? g.cu_obj[j1] : -1; }   // run of this ellipsoid (a bbox and a 3-D edge at most)
      if ((rbits[oj >> 5] >> (oj & 31)) & 1u) {
        int lo = 0, hi = ne;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (robj[mid] < oj) lo = mid + 1; else hi = mid; }
        int e1 = lo + 1;
        while (e1 < ne && robj[e1] == oj) ++e1;
        for (int e = lo; e < e1; ++e) {
          const long up = rid[e];
          if (up < g.n_bbox && !g.bb_valid[up]) continue;
          double yp[54];
#pragma unroll
          for (int k = 0; k < 54; ++k) yp[k] = ysh[e * kSchurYStride + k];
          for (int a2 = j; a2 < j1; ++a2) {
            const long uq = g.cu_id[a2];
            if (uq < g.n_bbox && !g.bb_valid[uq]) continue;
            any = true;
            ESL_SCHUR_ACCUM(yp, uq)
          }
        }
      }
      j = j1; oj = on;
    }
  } else {
    if (sq > sp) return;
    while (i < ie && j < je) {
      const int oi = g.cu_obj[i], oj = g.cu_obj[j];
      if (oi < oj) { ++i; continue; }
      if (oj < oi) { ++j; continue; }
      int i1 = i + 1, j1 = j + 1;                       // runs of this ellipsoid in both lists (a bbox and a 3-D edge at most)
      while (i1 < ie && g.cu_obj[i1] == oi) ++i1;
      while (j1 < je && g.cu_obj[j1] == oi) ++j1;
      for (int a1 = i; a1 < i1; ++a1) {
        const long up = g.cu_id[a1];
        if (up < g.n_bbox && !g.bb_valid[up]) continue;
        double yp[54];
#pragma unroll
        for (int k = 0; k < 54; ++k) yp[k] = Y[up * 54 + k];
        for (int a2 = j; a2 < j1; ++a2) {
          const long uq = g.cu_id[a2];
          if (uq < g.n_bbox && !g.bb_valid[uq]) continue;
          any = true;
          ESL_SCHUR_ACCUM(yp, uq)
        }
      }
      i = i1; j = j1;
    }
  }
  if (any) {
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int c = 0; c < 6; ++c) S[(long)(6 * sp + a) + (long)(6 * sq + c) * lda] -= out[a * 6 + c];
  }
  if (sp == sq) {
    double t[6] = {0, 0, 0, 0, 0, 0};
    for (int a1 = g.cu_start[sp]; a1 < ie; ++a1) {
      const long u = g.cu_id[a1];
      if (u < g.n_bbox && !g.bb_valid[u]) continue;
#pragma unroll
      for (int a = 0; a < 6; ++a) t[a] += Tb[(long)a * EU + u];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) S[n + (long)(6 * sp + a) * lda] -= t[a];
  }
}

// one wave per ellipsoid: x_o = Dinv (b_o - sum_e W_e^T x_c[cam_e]); trial ellipsoid; scale partial
static __global__ __launch_bounds__(kWave* kWavesPerBlock) void k_slam_backsub(
    DevGraph g, double lambda, const double* __restrict__ objs, const double* __restrict__ bo,
    const double* __restrict__ W, const double* __restrict__ Dinv, const double* __restrict__ xc,
    double* __restrict__ xo, double* __restrict__ objs_trial, double* __restrict__ part) {
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (o >= g.n_objs) return;
  const long EU = (long)g.n_bbox + g.n_e3d;
  const Ell e = ell_load(objs + 10 * o);
  const bool active = (g.bb_start[o + 1] > g.bb_start[o]) || (g.e3_start[o + 1] > g.e3_start[o]) || g.gr_cnt[o] > 0;
  if (!active) {
    if (lane == 0) { ell_store(e, objs_trial + 10 * o); part[o * 4 + 2] = 0; }
    return;
  }
  double t[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int q = g.ue_start[o] + lane; q < g.ue_start[o + 1]; q += 64) {
    const long u = g.ue_id[q];
    if (u < g.n_bbox && !g.bb_valid[u]) continue;
    const int slot = g.ue_slot[q];
    double x6[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) x6[a] = xc[(size_t)slot * 6 + a];
#pragma unroll
    for (int b = 0; b < 9; ++b) {
      double s = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a) s += W[(long)(a * 9 + b) * EU + u] * x6[a];
      t[b] += s;
    }
  }
#pragma unroll
  for (int b = 0; b < 9; ++b) t[b] = wave_sum(t[b]);
#pragma unroll
  for (int b = 0; b < 9; ++b) t[b] = __shfl(t[b], 0, 64);
  double x[9], bb[9];
  double scale = 0;
#pragma unroll
  for (int b = 0; b < 9; ++b) bb[b] = bo[(size_t)o * 9 + b];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    double s = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) s += Dinv[(size_t)o * 81 + i * 9 + k] * (bb[k] - t[k]);
    x[i] = s;
    scale += s * (lambda * s + bb[i]);
  }
  const Ell en = ell_oplus(e, x);
  if (lane == 0) {
    ell_store(en, objs_trial + 10 * o);
#pragma unroll
    for (int i = 0; i < 9; ++i) xo[(size_t)o * 9 + i] = x[i];
    part[o * 4 + 2] = scale;
  }
}

// one lane per camera: trial camera = exp(x_c) * Tcw for free cameras; scale partial
__device__ __forceinline__ void slam_cam_update_one(const DevGraph& g, double lambda, const double* __restrict__ cams, const double* __restrict__ xc,
                                                    const double* __restrict__ bc, double* __restrict__ cams_trial, double* __restrict__ cam_part, int cidx) {
  if (cidx >= g.n_cams) return;
  const SE3 T = se3_load(cams + 7 * cidx);
  const int slot = g.cam_slot[cidx];
  double scale = 0;
  if (slot < 0) {
    se3_store(T, cams_trial + 7 * cidx);
  } else {
    double u[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) { u[a] = xc[(size_t)slot * 6 + a]; scale += u[a] * (lambda * u[a] + bc[(size_t)slot * 6 + a]); }
    if (g.shard_rank != 0) scale = 0;   // camera part of computeScale counted once (b_c is the rank-summed vector)
    se3_store(cam_oplus(T, u), cams_trial + 7 * cidx);
  }
  cam_part[cidx * 4 + 0] = 0; cam_part[cidx * 4 + 1] = 0; cam_part[cidx * 4 + 2] = scale; cam_part[cidx * 4 + 3] = 1;
}
static __global__ void k_slam_cam_update(DevGraph g, double lambda, const double* __restrict__ cams,
                                  const double* __restrict__ xc, const double* __restrict__ bc,
                                  double* __restrict__ cams_trial, double* __restrict__ cam_part) {
  slam_cam_update_one(g, lambda, cams, xc, bc, cams_trial, cam_part, blockIdx.x * blockDim.x + threadIdx.x);
}

// ---- one launch for the trial's scalars (round 6): the two 4-tuple reductions and the odometry sum reduce_all used to launch (rounds 3-5: k_reduce_parts x 2, k_sum_into), in the same order and
// with the same trees (same bits), plus the fold of the dense solver's pivot flag into the "ok" partial -- at streaming sizes a
// trial is ~30 dependent launches of a few microseconds each, and these were four of them and a host round trip.
//   out[0..3] = {sum chi2, max maxdiag, sum scale, min ok} over the ellipsoids' and the cameras' partials, out[0] += sum of od_part
//   info != null: out[3] = 0 if bit 0 of *info is set (non-positive pivot: the step is rejected), out[4] = *info as a double
static __global__ __launch_bounds__(256) void k_slam_reduce_all(const double* __restrict__ obj_part, int n_objs, const double* __restrict__ cam_part, int n_cams,
                                                                const double* __restrict__ od_part, int n_od, double* __restrict__ out,
                                                                const int* __restrict__ info) {
  __shared__ double s0[256], s1[256], s2[256], s3[256];
  double r[4] = {0, 0, 0, 1};
  for (int pass = 0; pass < 3; ++pass) {
    const double* part = pass == 0 ? obj_part : (pass == 1 ? cam_part : od_part);
    const int n = pass == 0 ? n_objs : (pass == 1 ? n_cams : n_od);
    double a = 0, b = 0, c = 0, d = 1;
    if (pass < 2) {
      for (int i = threadIdx.x; i < n; i += 256) {
        a += part[i * 4 + 0];
        b = fmax(b, part[i * 4 + 1]);
        c += part[i * 4 + 2];
        d = fmin(d, part[i * 4 + 3]);
      }
    } else {
      for (int i = threadIdx.x; i < n; i += 256) a += part[i];
    }
    __syncthreads();   // (the previous pass's readers of the trees)
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
      if (pass == 0) { r[0] = s0[0]; r[1] = s1[0]; r[2] = s2[0]; r[3] = s3[0]; }
      else if (pass == 1) { r[0] += s0[0]; r[1] = fmax(r[1], s1[0]); r[2] += s2[0]; r[3] = fmin(r[3], s3[0]); }
      else if (n > 0) r[0] += s0[0];
    }
  }
  if (threadIdx.x == 0) {
    if (info) { const int f = info[0]; if (f & 1) r[3] = 0; out[4] = (double)f; }
    out[0] = r[0]; out[1] = r[1]; out[2] = r[2]; out[3] = r[3];
  }
}
// the pivot flag alone (runs where the flag is only final after a collective: replicated-graph ranks)
static __global__ void k_slam_fold_info(const int* __restrict__ info, double* __restrict__ out) {
  const int f = info[0];
  if (f & 1) out[3] = 0;
  out[4] = (double)f;
}

// chi2 of the trial states in ONE launch: blocks [0, nb_obj) take a wave per ellipsoid (its bbox, 3-D and gravity edges), the rest a lane per odometry edge
static __global__ __launch_bounds__(kWave* kWavesPerBlock) void k_slam_chi2_all(DevGraph g, int nb_obj, const double* __restrict__ cams,
                                                                                const double* __restrict__ objs, double* __restrict__ part,
                                                                                double* __restrict__ od_chi) {
  if ((int)blockIdx.x < nb_obj) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (o >= g.n_objs) return;
    const Ell e = ell_load(objs + 10 * o);
    const double chi = obj_chi2(g, cams, e, o, lane);
    if (lane == 0) part[o * 4 + 0] = chi;
    return;
  }
  const int e = ((int)blockIdx.x - nb_obj) * (kWave * kWavesPerBlock) + threadIdx.x;
  if (e >= g.n_odom) return;
  const int vi = g.od_i[e], vj = g.od_j[e];
  if (g.cam_slot[vi] < 0 && g.cam_slot[vj] < 0) { od_chi[e] = 0; return; }
  double r[6];
  res_odom(se3_load(cams + 7 * vi), se3_load(cams + 7 * vj), se3_load(g.od_meas + 7 * e), r);
  double chi = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) chi += r[k] * g.od_info[6 * e + k] * r[k];
  od_chi[e] = chi;
}

// xc <- solution vector x (n doubles) : plain copy kernel not needed (same layout: slot*6 + a)

}  // namespace esl
