// esl_kernels_fused.hpp — the mapping-mode LM trial with the sums of an ellipsoid kept inside one workgroup.
//
// The chunked pipeline (esl_kernels_chunk.hpp) scattered the <= 64-edge chunks of an ellipsoid over unrelated workgroups:
// every chunk wrote its 54 partial sums to HBM (2.9 MB per linearisation at C4) and k_lm_step gathered them again
// (3.9 MB, 3.4 us of dependent loads) before ONE lane per ellipsoid solved the 9x9 system with the whole matrix in
// registers (254 VGPRs, a 400-deep dependent chain, 16 of 256 lanes busy).  Here
//   k_group_linearize  a workgroup OWNS a few consecutive ellipsoids: the host cuts their edges into ITEMS (<= 64 bbox edges,
//                      or two <= 32-edge 3-D slices) and spreads the items over the workgroup's waves longest-first
//                      (FusedSched, built at upload); lane = edge, residual + analytic Jacobian in registers, J^T W J /
//                      -J^T W r summed through a wave-private LDS transpose (9 of the 54 sums per pass) into the LDS
//                      accumulator slot of the item's (wave, ellipsoid) pair; slots are added per ellipsoid in fixed order,
//                      the gravity prior is added, and 432 B per ELLIPSOID go to HBM;
//   k_group_step       decides the previous trial (lm_decide) and solves: one WAVE per ellipsoid, row-parallel 9x9 LDL^T on
//                      lanes 0..8 (pivot / row broadcasts with v_readlane), retraction -> trial state.
// A trial is linearised AT its trial state, so its chi2 needs no residual-only pass and, if it is accepted, its H, b are
// already the next iteration's system (accept = flip of LmCore::cur).
//
// A single kernel per trial (decide + solve + linearise in one launch, 16-wave workgroups owning 8 ellipsoids, everything
// resident at once) was built and measured first: 49 us per launch at C4 against 21 + 13 us for the two launches it was
// to replace -- the linearisation is bound by FP64 VALU issue (~4.2 us per item per SIMD slot however the waves are
// packed), so idling every SIMD during the serial decide + solve prefix and squeezing the Jacobian into 128 VGPRs (61
// spills) cost more than the second dispatch (profiles/r2_fused_variants.txt).
//
// Reference pieces replaced: BlockSolver::buildSystem / solve (Thirdparty/g2o/g2o/core/block_solver.hpp:354-365, 502-560),
// LinearSolverDense (solvers/linear_solver_dense.h:65-113), SparseOptimizer::update / computeActiveErrors /
// activeRobustChi2 (core/sparse_optimizer.cpp:61-114, 422-435), OptimizationAlgorithmLevenberg::solve /
// computeLambdaInit (core/optimization_algorithm_levenberg.cpp:61-180), the NaN pre-check of the bbox edges
// (src/core/Optimizer.cpp:234-243).
#pragma once
#include "esl_kernels_chunk.hpp"

namespace esl {

#ifndef ESL_FZ_OBJS       // tuning knobs (scripts/build_variants.sh builds side-by-side variants of the library)
#define ESL_FZ_OBJS 2
#endif
#ifndef ESL_FZ_WAVES
#define ESL_FZ_WAVES 4
#endif
#ifndef ESL_FZ_WAVES_PER_EU
#define ESL_FZ_WAVES_PER_EU 3
#endif
#ifndef ESL_FZ_SLOTS
#define ESL_FZ_SLOTS 16
#endif
constexpr int kFzObjs = ESL_FZ_OBJS;          // ellipsoids per workgroup
constexpr int kFzWaves = ESL_FZ_WAVES;        // waves per workgroup
constexpr int kFzThreads = 64 * kFzWaves;
constexpr int kSysStride = 56;                // per ellipsoid: 45 packed H | 9 b | chi2 | spare
constexpr int kFzPart = 4;                    // per workgroup: chi2 | scale (INIT: max diag) | ok | (INIT: dropped bbox edges)
constexpr int kFzSlots = ESL_FZ_SLOTS;         // accumulator slots per workgroup: one per (wave, ellipsoid) pair that occurs
constexpr int kFzRows = 9;                    // sums per pass through the transpose tile
constexpr int kFzTile = kFzRows * kTrStride;  // doubles per wave

struct FzItem {   // type 0: one bbox slice (A); type 1: two 3-D slices (A = lanes 0-31, B = lanes 32-63; empty slice: begin == end)
  int type, oa, ba, ea, ob, bb, eb, slots;   // o = ellipsoid index inside the workgroup; slots = slot of A | slot of B << 16
};
struct FusedSched {
  int n_wg = 0;
  const int* wg_obj = nullptr;       // n_wg + 1: first ellipsoid of each workgroup (<= kFzObjs ellipsoids, <= kFzSlots slots)
  const int* wave_start = nullptr;   // n_wg * kFzWaves + 1: items of (workgroup, wave)
  const FzItem* items = nullptr;
  const int* slot_obj = nullptr;     // n_wg * kFzSlots: ellipsoid (inside the workgroup) a slot belongs to, -1 = unused
};

template <int D, int G, int... T>
__device__ __forceinline__ void fz_fill(double* v, const double* J, const double* r, double w, std::integer_sequence<int, T...>) {
  ((v[T] = hb_entry<D, G * kFzRows + T>(J, r, w)), ...);
}
template <int G, int... T>
__device__ __forceinline__ void fz_fill_e3d(double* v, const double* Jp, const double* r, double w, std::integer_sequence<int, T...>) {
  ((v[T] = hb_entry_e3d<G * kFzRows + T>(Jp, r, w)), ...);
}
// wave-wide sums of entries [G*9, G*9+9) through the wave's transpose tile tr[9][65]: every lane parks its 9 values, 36 lanes
// add a quarter of a row each (16 columns), three shuffles finish -> the totals are returned in lanes 0..8
template <int D, int G>
__device__ __forceinline__ double fz_group(const double* J, const double* r, double w, int lane, double* __restrict__ tr) {
  double v[kFzRows];
  fz_fill<D, G>(v, J, r, w, std::make_integer_sequence<int, kFzRows>{});
#pragma unroll
  for (int k = 0; k < kFzRows; ++k) tr[k * kTrStride + lane] = v[k];
  __builtin_amdgcn_wave_barrier();
  const int k = lane % kFzRows, p = lane / kFzRows;   // lanes 0..35: quarter p of row k
  double s0 = 0, s1 = 0;
  if (lane < 4 * kFzRows) {
    const double* row = tr + k * kTrStride + p * 16;
#pragma unroll
    for (int i = 0; i < 16; i += 2) { s0 += row[i]; s1 += row[i + 1]; }
  }
  const double s = s0 + s1;
  const double sa = __shfl_down(s, kFzRows, 64), sb = __shfl_down(s, 2 * kFzRows, 64), sc = __shfl_down(s, 3 * kFzRows, 64);
  __builtin_amdgcn_wave_barrier();
  return ((s + sa) + sb) + sc;
}
// 3-D slices: the totals of a half wave are returned in its lanes t = 0..8
template <int G>
__device__ __forceinline__ double fz_group_e3d(const double* Jp, const double* r, double w, int lane64, double* __restrict__ tr) {
  double v[kFzRows];
  fz_fill_e3d<G>(v, Jp, r, w, std::make_integer_sequence<int, kFzRows>{});
#pragma unroll
  for (int k = 0; k < kFzRows; ++k) tr[k * kTrStride + lane64] = v[k];
  __builtin_amdgcn_wave_barrier();
  const int t = lane64 & 31, half = lane64 & 32;
  const int k = t % kFzRows, p = t / kFzRows;         // t = 0..17: half p (16 columns) of row k
  double s0 = 0, s1 = 0;
  if (t < 2 * kFzRows) {
    const double* row = tr + k * kTrStride + half + p * 16;
#pragma unroll
    for (int i = 0; i < 16; i += 2) { s0 += row[i]; s1 += row[i + 1]; }
  }
  const double s = s0 + s1;
  const double sa = __shfl_down(s, kFzRows, 64);
  __builtin_amdgcn_wave_barrier();
  return s + sa;
}

// one bbox slice of one ellipsoid (state in LDS) -> += this wave's accumulator of that ellipsoid
template <int JAC, bool VALIDATE>
__device__ __forceinline__ void fz_bbox_item(const DevGraph& g, const double* __restrict__ cams, const double* __restrict__ ell_lds,
                                             int begin, int end, double delta, double* __restrict__ tr, double* __restrict__ accE,
                                             int lane) {
  const Ell e = ell_load(ell_lds);
  const int i = begin + lane;
  const bool in = i < end;
  double r[4] = {0, 0, 0, 0}, J[36], w = 0, chi = 0;
  const bool act = in && (VALIDATE || g.bb_valid[i]);
  if (JAC == ESL_JAC_NUMERIC) {   // the 18 perturbed states are the same for every edge of the slice: 18 lanes compute one each
    if (lane < 18) {
      double u[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 9; ++q) u[q] = (q == (lane >> 1)) ? ((lane & 1) ? -delta : delta) : 0.0;
      ell_store(ell_oplus(e, u), tr + 10 * lane);
    }
    __builtin_amdgcn_wave_barrier();
  }
  bool bad = false;
  if (act) {
    const SE3 T = se3_load(cams + 7 * g.bb_cam[i]);
    double meas[4] = {g.bb_meas[4 * i], g.bb_meas[4 * i + 1], g.bb_meas[4 * i + 2], g.bb_meas[4 * i + 3]};
    w = g.bb_w[i];
    if (JAC == ESL_JAC_ANALYTIC) jac_bbox_t<true, false>(T, e, g.K, meas, r, J, nullptr);
    else {
      res_bbox(T, e, g.K, meas, r);
      const double scalar = 1.0 / (2 * delta);
      for (int d = 0; d < 9; ++d) {   // not unrolled: one body, 9 trips
        double rp[4], rm[4];
        res_bbox(T, ell_load(tr + 20 * d), g.K, meas, rp);
        res_bbox(T, ell_load(tr + 20 * d + 10), g.K, meas, rm);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
          for (int q = 0; q < 9; ++q) if (q == d) J[k * 9 + q] = scalar * (rp[k] - rm[k]);
        }
      }
    }
    chi = w * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
    if (VALIDATE) {   // Optimizer.cpp:234-243: an edge whose chi2 is NaN at the start state is dropped for the whole run
      bad = (chi != chi);
      g.bb_valid[i] = bad ? 0 : 1;
      if (bad) {
        chi = 0; w = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = 0;
#pragma unroll
        for (int k = 0; k < 36; ++k) J[k] = 0;
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < 36; ++k) J[k] = 0;
  }
  if (JAC == ESL_JAC_NUMERIC) __builtin_amdgcn_wave_barrier();   // the perturbed states have been read; the tile is reused
  { const double v = fz_group<4, 0>(J, r, w, lane, tr); if (lane < kFzRows) accE[lane] += v; }
  { const double v = fz_group<4, 1>(J, r, w, lane, tr); if (lane < kFzRows) accE[9 + lane] += v; }
  { const double v = fz_group<4, 2>(J, r, w, lane, tr); if (lane < kFzRows) accE[18 + lane] += v; }
  { const double v = fz_group<4, 3>(J, r, w, lane, tr); if (lane < kFzRows) accE[27 + lane] += v; }
  { const double v = fz_group<4, 4>(J, r, w, lane, tr); if (lane < kFzRows) accE[36 + lane] += v; }
  { const double v = fz_group<4, 5>(J, r, w, lane, tr); if (lane < kFzRows) accE[45 + lane] += v; }
  chi = wave_sum(chi);
  if (lane == 0) accE[54] += chi;
  if (VALIDATE) {
    const int nbad = __popcll(__ballot(bad));
    if (lane == 0 && nbad) accE[55] += (double)nbad;
  }
}

// two 3-D slices (<= 32 edges each) of ellipsoids of this workgroup; a half with begin == end idles
// `same`: both halves belong to ONE ellipsoid (more than 32 3-D edges on it) -> the lower half adds the upper half's totals
// to its own and is the only one to touch that ellipsoid's accumulator
template <int JAC>
__device__ __forceinline__ void fz_e3d_item(const DevGraph& g, const double* __restrict__ cams, const double* __restrict__ ell_lds /* of this half */,
                                            int begin, int end, double delta, double* __restrict__ tr, double* __restrict__ accE, int lane64,
                                            bool same) {
  const int t = lane64 & 31;
  const int i = begin + t;
  const bool in = i < end;
  const Ell e = ell_load(ell_lds);
  double r[9], Jp[36], w = 0, chi = 0;
  if (in) {
    const SE3 T = se3_load(cams + 7 * g.e3_cam[i]);
    const Ell m = ell_load(g.e3_meas + 10 * i);
    w = g.e3_w[i];
    if (JAC == ESL_JAC_ANALYTIC) jac_e3d_pose(T, e, m, g.yt, r, Jp);
    else {   // g2o's central differences for the 6 pose columns (the scale block of this edge is exactly the identity)
      const SE3 E0 = e3d_E0(T, e, m);
      res_e3d_from_E0(E0, e.s, m.s, g.yt, r);
      const double scalar = 1.0 / (2 * delta);
      for (int d = 0; d < 6; ++d) {   // not unrolled: one body, 6 trips
        double u[6] = {0, 0, 0, 0, 0, 0}, rp[9], rm[9];
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
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) chi += r[k] * r[k];
    chi *= w;
  } else {
#pragma unroll
    for (int k = 0; k < 36; ++k) Jp[k] = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) r[k] = 0;
  }
  const bool wr = begin < end && !(same && lane64 >= 32);
#define ESL_FZ_E3D_GROUP(G)                                           \
  {                                                                   \
    double v = fz_group_e3d<G>(Jp, r, w, lane64, tr);                 \
    const double u = __shfl_down(v, 32, 64);                          \
    if (same) v += u;                                                 \
    if (wr && t < kFzRows) accE[G * kFzRows + t] += v;                \
  }
  ESL_FZ_E3D_GROUP(0) ESL_FZ_E3D_GROUP(1) ESL_FZ_E3D_GROUP(2) ESL_FZ_E3D_GROUP(3) ESL_FZ_E3D_GROUP(4) ESL_FZ_E3D_GROUP(5)
#undef ESL_FZ_E3D_GROUP
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) chi += __shfl_xor(chi, off, 64);
  const double uc = __shfl_down(chi, 32, 64);
  if (same) chi += uc;
  if (wr && t == 0) accE[54] += chi;
}

__device__ __forceinline__ double fz_readlane(double v, int src) {   // src: compile-time constant after unrolling
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
// (H + lambda I) x = b for one 9x9 block, ROW-PARALLEL: lane i (< 9) of the calling wave owns row i.  Right-looking LDL^T
// without pivoting, pivots / rows broadcast with v_readlane; the triangular solves run on the same 9 lanes (the factor is
// transposed once through LDS so that lane i also owns column i).  One lane with the whole matrix in registers (the
// version of esl_math.hpp) needs ~250 VGPRs and a 400-deep dependent chain; this needs 9 + 9 doubles and ~100 steps.
// hb: 45 packed upper + 9 rhs (LDS); tl: 81 doubles of LDS scratch of this wave; returns x[0..8] in every lane and
// `ok` = all pivots positive (Eigen's LDLT::isPositive of linear_solver_dense.h:107-112).
__device__ __forceinline__ bool fz_solve9(const double* __restrict__ hb, double lambda, int lane, double* __restrict__ tl, double x[9]) {
  const int i = lane < 9 ? lane : 8;   // lanes >= 9 shadow row 8
  double a[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    const int r = i < j ? i : j, c = i < j ? j : i;
    a[j] = hb[r * 9 - (r * (r - 1)) / 2 + (c - r)] + ((i == j) ? lambda : 0.0);
  }
  double y = hb[45 + i], d = 0;
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const double dk = fz_readlane(a[k], k);
    ok = ok && (dk > 0);
    if (i == k) d = dk;
    const double lik = a[k] * (1.0 / dk);
#pragma unroll
    for (int j = k + 1; j < 9; ++j) {
      const double rkj = fz_readlane(a[j], k);
      if (i > k) a[j] -= lik * rkj;
    }
    if (i > k) a[k] = lik;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {   // L y = b
    const double yk = fz_readlane(y, k);
    if (i > k) y -= a[k] * yk;
  }
  double z = y / d;                 // D z = y
  if (lane < 9) {
#pragma unroll
    for (int j = 0; j < 9; ++j) tl[lane * 9 + j] = a[j];
  }
  __builtin_amdgcn_wave_barrier();
  double c[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) c[j] = tl[j * 9 + i];   // c[j] = L[j][i]: column i of L
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int k = 8; k >= 1; --k) {    // L^T x = z
    const double xk = fz_readlane(z, k);
    if (i < k) z -= c[k] * xk;
  }
#pragma unroll
  for (int j = 0; j < 9; ++j) x[j] = fz_readlane(z, j);
  return ok;
}

// packed upper-triangle index of the diagonal entry (a, a)
__device__ __forceinline__ constexpr int tri_diag(int a) { return a * 9 - a * (a - 1) / 2; }

// Linearisation of the workgroup's ellipsoids: residual + Jacobian of every edge at the linearisation point, J^T W J /
// -J^T W r / chi2 summed per ellipsoid inside the workgroup (+ the gravity prior, a unary edge on the ellipsoid) -> the
// ellipsoid's 54 sums + chi2 in HBM, one {chi2, max |H_kk|, -, dropped edges} partial per workgroup.
// st == null: linearise `objs_a` into `sys_a` (start of a run; VALIDATE: with the NaN pre-check of the bbox edges,
// Optimizer.cpp:234-243).  st != null: the device-resident LM state picks the pair -- the linearisation point is the TRIAL
// state (the pair that is NOT current) and the launch is a no-op once the run is done.
template <int JAC, bool VALIDATE>
static __global__ __launch_bounds__(kFzThreads, ESL_FZ_WAVES_PER_EU) void k_group_linearize(DevGraph g, FusedSched sc, const double* __restrict__ cams,
                                                                       const double* __restrict__ objs_a, const double* __restrict__ objs_b,
                                                                       double* __restrict__ sys_a, double* __restrict__ sys_b,
                                                                       const LmCore* __restrict__ st, double* __restrict__ part_out, double delta) {
  __shared__ double tr_all[kFzWaves * kFzTile];
  __shared__ double acc[kFzSlots][kSysStride];
  __shared__ double ellt[kFzObjs][10];
  __shared__ double grv[kFzObjs][12];    // gravity prior at the linearisation point: Jg[9], rg, weight
  __shared__ double esum[kFzObjs][kSysStride];
  __shared__ int slot_obj[kFzSlots];
  const double* objs = objs_a;
  double* sys_out = sys_a;
  if (st) {
    if (st->done) return;
    if (st->cur == 0) { objs = objs_b; sys_out = sys_b; }   // trial = the pair that is NOT current
  }
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wg = blockIdx.x, o0 = sc.wg_obj[wg], n_own = sc.wg_obj[wg + 1] - o0;
  for (int k = tid; k < kFzSlots * kSysStride; k += kFzThreads) (&acc[0][0])[k] = 0.0;
  if (tid < kFzSlots) slot_obj[tid] = sc.slot_obj[wg * kFzSlots + tid];
  if (tid < kFzObjs * 10) {
    const int e = tid / 10, k = tid - e * 10;
    ellt[e][k] = e < n_own ? objs[(size_t)(o0 + e) * 10 + k] : (k == 6 ? 1.0 : 0.0);
  }
  if (tid >= 64 && tid < 64 + kFzObjs) {   // one lane per ellipsoid (second wave when there is one): the gravity prior
    const int e = tid - 64;
    double* gv = grv[e];
#pragma unroll
    for (int k = 0; k < 12; ++k) gv[k] = 0;
    if (e < n_own && g.gr_cnt[o0 + e] > 0) {
      const Ell el = ell_load(objs + (size_t)(o0 + e) * 10);
      double Jg[9], rg;
      if (JAC == ESL_JAC_ANALYTIC) rg = jac_grav(el, g.grav_n, Jg);
      else {
        rg = res_grav(el, g.grav_n);
        numeric_jac_obj(el, delta, 1, Jg, [&](const Ell& ep, double* o1) { o1[0] = res_grav(ep, g.grav_n); });
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) gv[k] = Jg[k];
      gv[9] = rg; gv[10] = g.grav_w * g.gr_cnt[o0 + e];
    }
  }
  __syncthreads();
  {
    double* tr = tr_all + wave * kFzTile;
    const int i0 = sc.wave_start[wg * kFzWaves + wave], i1 = sc.wave_start[wg * kFzWaves + wave + 1];
    for (int it = i0; it < i1; ++it) {
      const FzItem m = sc.items[it];
      if (m.type == 0) {
        fz_bbox_item<JAC, VALIDATE>(g, cams, ellt[m.oa], m.ba, m.ea, delta, tr, acc[m.slots & 0xffff], lane);
      } else {
        const bool hiB = lane >= 32;
        fz_e3d_item<JAC>(g, cams, ellt[hiB ? m.ob : m.oa], hiB ? m.bb : m.ba, hiB ? m.eb : m.ea, delta, tr,
                         acc[hiB ? (m.slots >> 16) : (m.slots & 0xffff)], lane, m.oa == m.ob && m.eb > m.bb);
      }
    }
  }
  __syncthreads();
  // sums over the slots in fixed order (+ the gravity prior) -> the system of the linearised state
  for (int idx = tid; idx < kFzObjs * kSysStride; idx += kFzThreads) {
    const int e = idx / kSysStride, k = idx - e * kSysStride;
    double v = 0;
    for (int sl = 0; sl < kFzSlots; ++sl) {
      const int so = slot_obj[sl];
      if (so < 0) break;               // slots are handed out densely
      if (so == e) v += acc[sl][k];
    }
    const double* gv = grv[e];
    const double wgv = gv[10];
    if (k < 45) {
      const int a = tri_a(k), c2 = tri_c(k);
      v += (wgv * gv[a]) * gv[c2];
    } else if (k < 54) {
      v -= gv[k - 45] * (wgv * gv[9]);
    } else if (k == 54) {
      v += wgv * gv[9] * gv[9];
    }
    if (e < n_own && k < 55) sys_out[(size_t)(o0 + e) * kSysStride + k] = v;
    esum[e][k] = v;
  }
  __syncthreads();
  if (tid == 0) {
    double chi = 0, md = 0, nd = 0;
#pragma unroll
    for (int e = 0; e < kFzObjs; ++e) {
      chi += esum[e][54];
      nd += esum[e][55];
#pragma unroll
      for (int a = 0; a < 9; ++a) md = fmax(md, fabs(esum[e][tri_diag(a)]));
    }
    part_out[wg * kFzPart + 0] = chi;
    part_out[wg * kFzPart + 1] = md;
    part_out[wg * kFzPart + 2] = 1.0;
    part_out[wg * kFzPart + 3] = nd;
  }
}

// One LM trial's head: (1) every workgroup decides the PREVIOUS trial (lm_decide: accept / reject, lambda, stop rule) from
// the chi2 partials its linearisation left in part_lin and the {scale, ok} partials the previous launch of this kernel
// left in sp_in -- redundantly, so there is no grid-wide handshake; (2) one WAVE per ellipsoid: (H + lambda I) x = b from
// the 54 sums of the now-current state (row-parallel 9x9 LDL^T on lanes 0..8), retraction -> trial state.
// first: 0 = decide; 1 = initialise the LM state from the partials of the start state's linearisation (single GPU);
// 2 = from the gathered blocks (sharded).
constexpr int kStepWaves = 4;
static __global__ __launch_bounds__(64 * kStepWaves) void k_group_step(DevGraph g, const double* __restrict__ sys_a, const double* __restrict__ sys_b,
                                                                        double* __restrict__ objs_a, double* __restrict__ objs_b,
                                                                        const LmCore* __restrict__ in, LmCore* __restrict__ out,
                                                                        const double* __restrict__ part_lin, int n_lin,
                                                                        const double* __restrict__ sp_in, double* __restrict__ sp_out, int first,
                                                                        int max_iters, int max_trials, LmHostView* __restrict__ host,
                                                                        double* __restrict__ xo, const double* __restrict__ gathered, int n_ranks,
                                                                        double tau, int n_grav_total) {
  __shared__ double red[4 * kStepWaves];
  __shared__ double hbs[kStepWaves][56];
  __shared__ double tls[kStepWaves][81];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const bool writer = blockIdx.x == 0 && tid == 0;
  const int o = blockIdx.x * kStepWaves + wave;
  // the sums of this wave's ellipsoid from BOTH pairs (which one is current is only known after the decision)
  double va = 0, vb = 0;
  if (lane < 54 && o < g.n_objs) { va = sys_a[(size_t)o * kSysStride + lane]; vb = sys_b[(size_t)o * kSysStride + lane]; }
  double c = 0, s2 = 0, okv = 1, nd = 0;
  if (n_ranks == 0) {
    for (int b = tid; b < n_lin; b += 64 * kStepWaves) {
      c += part_lin[b * kFzPart + 0];
      if (first) { s2 = fmax(s2, part_lin[b * kFzPart + 1]); nd += part_lin[b * kFzPart + 3]; }
    }
    if (!first)
      for (int b = tid; b < (int)gridDim.x; b += 64 * kStepWaves) { s2 += sp_in[b * 2 + 0]; okv = fmin(okv, sp_in[b * 2 + 1]); }
  }
  LmCore s;
  if (first == 0) s = *in;
  if (first == 0 && s.done) {   // queued behind a finished run: pass the state on and leave
    if (writer) *out = s;
    return;
  }
  if (n_ranks == 0) {   // four block reductions behind one barrier pair
    c = wave_sum(c); okv = -wave_max(-okv); nd = wave_sum(nd);
    s2 = first ? wave_max(s2) : wave_sum(s2);
    if (lane == 0) { red[wave] = c; red[kStepWaves + wave] = s2; red[2 * kStepWaves + wave] = okv; red[3 * kStepWaves + wave] = nd; }
    __syncthreads();
    c = red[0]; s2 = red[kStepWaves]; okv = red[2 * kStepWaves]; nd = red[3 * kStepWaves];
#pragma unroll
    for (int k = 1; k < kStepWaves; ++k) {
      c += red[k];
      s2 = first ? fmax(s2, red[kStepWaves + k]) : s2 + red[kStepWaves + k];
      okv = fmin(okv, red[2 * kStepWaves + k]);
      nd += red[3 * kStepWaves + k];
    }
  }
  if (first) {   // computeLambdaInit + the bookkeeping of iteration 0
    bool any_edge;
    if (first == 2) {
      double chi = 0, md = 0, any = 0;
      for (int r = 0; r < n_ranks; ++r) { chi += gathered[r * 8 + 0]; md = fmax(md, gathered[r * 8 + 1]); any = fmax(any, gathered[r * 8 + 5]); }
      c = chi; s2 = md; any_edge = any > 0.5;
    } else {
      any_edge = (g.n_bbox - (int)nd > 0) || g.n_e3d > 0 || n_grav_total > 0;
      if (writer) host->n_dropped = (int)nd;
    }
    lm_core_init(s, c, s2, tau);
    if (!any_edge) {   // no active edge: nothing to optimise (stop_reason 3)
      s.done = 1; s.stop_reason = 3;
      if (writer) {
        *out = s;
        host->core = s;
        host->trace_len = 0;
        __hip_atomic_store(&host->done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      return;
    }
  } else {
    if (n_ranks > 0) {
      c = 0; s2 = 0; okv = 1;
      for (int r = 0; r < n_ranks; ++r) { c += gathered[r * 8 + 2]; s2 += gathered[r * 8 + 3]; okv = fmin(okv, gathered[r * 8 + 4]); }
    }
    lm_decide(s, c, s2, okv, max_iters, max_trials, writer, host);
    if (writer) {
      if (s.done) {   // results first, then the flags with release semantics: the host reads them as soon as it sees `done`
        host->core = s;
        __hip_atomic_store(&host->done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&host->seq, s.trial_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      } else {        // progress counter only (it throttles the host's enqueueing): no system-scope release fence per trial
        __hip_atomic_store(&host->seq, s.trial_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  if (writer) *out = s;
  if (s.done) return;
  const double* objs = s.cur ? objs_b : objs_a;
  double* objs_trial = s.cur ? objs_a : objs_b;
  const double lambda = s.lambda;
  if (lane < 54) hbs[wave][lane] = s.cur ? vb : va;
  __builtin_amdgcn_wave_barrier();
  double scale = 0, okd = 1;
  if (o < g.n_objs) {
    Ell e = ell_load(objs + 10 * (size_t)o);
    const bool active = g.gr_cnt[o] > 0 || g.bb_start[o + 1] > g.bb_start[o] || g.e3_start[o + 1] > g.e3_start[o];
    if (active) {   // an inactive vertex is never touched (sparse_optimizer.cpp:236-257)
      const double* hb = hbs[wave];
      double x[9];
      const bool ok = fz_solve9(hb, lambda, lane, tls[wave], x);
#pragma unroll
      for (int i = 0; i < 9; ++i) scale += x[i] * (lambda * x[i] + hb[45 + i]);
      if (lane < 9) {
        double xl = x[0];
#pragma unroll
        for (int i = 1; i < 9; ++i) if (lane == i) xl = x[i];
        xo[(size_t)o * 9 + lane] = xl;
      }
      okd = ok ? 1.0 : 0.0;
      e = ell_oplus(e, x);
    }
    if (lane == 0) ell_store(e, objs_trial + 10 * (size_t)o);
  }
  // this workgroup's share of the trial scalars, ellipsoids in order
  __syncthreads();
  if (lane == 0) { red[wave] = scale; red[kStepWaves + wave] = okd; }
  __syncthreads();
  if (tid == 0) {
    double sc2 = 0, ok2 = 1;
#pragma unroll
    for (int k = 0; k < kStepWaves; ++k) { sc2 += red[k]; ok2 = fmin(ok2, red[kStepWaves + k]); }
    sp_out[blockIdx.x * 2 + 0] = sc2; sp_out[blockIdx.x * 2 + 1] = ok2;
  }
}

// Sharded run: this rank's share of a launch's scalars -> dev_scal, the send buffer of the all-gather.
// init != 0: {chi2_lin, max_diag, -, -, -, has_edges}; else dev_scal[2..4] = {chi2_trial, scale, ok}
static __global__ __launch_bounds__(256) void k_fz_partials(const double* __restrict__ part, int n_wg, const double* __restrict__ sp, int n_sp,
                                                            int init, int n_bbox, int n_e3d, int n_grav, double* __restrict__ dev_scal,
                                                            LmHostView* __restrict__ host) {
  __shared__ double sm4[4];
  double c = 0, s2 = 0, okv = 1, nd = 0;
  for (int b = threadIdx.x; b < n_wg; b += 256) {
    c += part[b * kFzPart + 0];
    if (init) { s2 = fmax(s2, part[b * kFzPart + 1]); nd += part[b * kFzPart + 3]; }
  }
  if (!init)
    for (int b = threadIdx.x; b < n_sp; b += 256) { s2 += sp[b * 2 + 0]; okv = fmin(okv, sp[b * 2 + 1]); }
  c = block256_sum1(c, sm4);
  nd = block256_sum1(nd, sm4);
  okv = block256_min1(okv, sm4);
  s2 = init ? -block256_min1(-s2, sm4) : block256_sum1(s2, sm4);
  if (threadIdx.x == 0) {
    if (init) {
      dev_scal[0] = c; dev_scal[1] = s2;
      dev_scal[5] = ((n_bbox - (int)nd > 0) || n_e3d > 0 || n_grav > 0) ? 1.0 : 0.0;
      host->n_dropped = (int)nd;
    } else {
      dev_scal[2] = c; dev_scal[3] = s2; dev_scal[4] = okv;
    }
  }
}

}  // namespace esl
