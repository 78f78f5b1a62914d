// esl_kernels_chunk.hpp — mapping-mode kernels and the device-resident Levenberg-Marquardt control.
//
// The unit of work is a CHUNK = up to 64 bbox edges (or 32 3-D edges) hanging on one ellipsoid (table built at upload;
// edges are sorted by ellipsoid, so a chunk is a contiguous, coalesced slice).  The first cut (one wave per ellipsoid, 54
// accumulators per lane carried across rounds) ran at 1 wave/SIMD with scratch spills and a tail set by the ellipsoid with
// the most edges.
//   k_chunk_linearize_both  one wave per chunk (two 3-D chunks per wave), both edge types in one launch: lane = edge;
//                           residual + Jacobian in registers, the 45+9 entries of J^T W J / -J^T W r summed over the wave
//                           through a wave-private LDS transpose, written as a chunk partial; per-workgroup chi2 partial.
//                           The first linearisation of a run also performs the NaN pre-check of the bbox edges.
//   k_chunk_finalize_rows   once per run: per-ellipsoid diagonal of H -> max diag (lambda_0), chi2, LM state initialisation
//   k_lm_step_rows          one LM trial's head: every workgroup decides the previous trial (accept / reject, lambda, stop
//                           rule -- lm_decide), then solves its 4 ellipsoids (a wave each, row-parallel 9x9) for the new lambda -> trial state.
//                           A trial is linearised AT its trial state, so its chi2 is the sum of the chunk chi2 and its
//                           H, b are the next iteration's system if it is accepted: no residual-only pass.
//   k_lm_partials           sharded runs: this rank's share of a trial's scalars -> send buffer of the all-gather
//   k_chunk_linearize / k_obj_solve / k_chunk_chi2 / k_chunk_combine
//                           per-type numeric-Jacobian kernels and the pieces behind the step API (esl_lm_*)
// Reference pieces replaced: BlockSolver::buildSystem (core/block_solver.hpp:502-560), BlockSolver::solve +
// LinearSolverDense (:356-365, solvers/linear_solver_dense.h:65-113), SparseOptimizer::update / computeActiveErrors /
// activeRobustChi2 (core/sparse_optimizer.cpp:61-114, 422-435), OptimizationAlgorithmLevenberg::solve
// (core/optimization_algorithm_levenberg.cpp:61-164).
#pragma once
#include <cfloat>
#include <utility>

#include "esl_kernels_map.hpp"

namespace esl {

struct ChunkTable {
  int n_chunks = 0;
  int* obj = nullptr;     // ellipsoid of the chunk
  int* type = nullptr;    // 0 bbox, 1 3-D, 2 gravity
  int* begin = nullptr;   // first edge (index into the sorted edge arrays of that type)
  int* end = nullptr;
  int* ostart = nullptr;  // n_objs + 1 : chunks of each ellipsoid (contiguous)
};

constexpr int kChunkOut = 56;  // 45 packed H + 9 b + chi2 + pad
// waves per workgroup of the linearisation kernels.  Measured at C4: 1-wave workgroups make the launch 0.8 us shorter
// (finer spreading over the 1024 SIMDs) and k_lm_step 0.8 us longer (4x the per-workgroup chi2 partials to add): a wash.
constexpr int kLinWaves = 4;
#ifndef ESL_LIN_MIN_WAVES
#define ESL_LIN_MIN_WAVES 1   // waves per SIMD the linearisation is compiled for (tuning: scripts/build_variants.sh)
#endif

// packed upper-triangle index -> (a, c)
__device__ __forceinline__ constexpr int tri_a(int p) {
  int a = 0, base = 0;
  while (p >= base + (9 - a)) { base += 9 - a; ++a; }
  return a;
}
__device__ __forceinline__ constexpr int tri_c(int p) {
  int a = 0, base = 0;
  while (p >= base + (9 - a)) { base += 9 - a; ++a; }
  return a + (p - base);
}

// entry P of {J^T W J packed upper (45), -J^T W r (9)} for this lane's edge (P is a compile-time constant, so
// every J index below is static and J stays in registers)
template <int D, int P>
__device__ __forceinline__ double hb_entry(const double* J, const double* r, double w) {
  if constexpr (P < 45) {
    constexpr int a = tri_a(P), c = tri_c(P);
    double s = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) s += (w * J[k * 9 + a]) * J[k * 9 + c];
    return s;
  } else if constexpr (P < 54) {
    constexpr int a = P - 45;
    double s = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) s -= J[k * 9 + a] * (w * r[k]);
    return s;
  } else {
    return 0.0;
  }
}
template <int D, int G, int... T>
__device__ __forceinline__ void fill_group(double* v, const double* J, const double* r, double w, std::integer_sequence<int, T...>) {
  ((v[T] = hb_entry<D, G * 18 + T>(J, r, w)), ...);
}

// same for a 3-D edge, whose Jacobian is [[Jp 6x6, 0], [0, I3]] (Jp row-major 6x6, r has 9 rows)
template <int P>
__device__ __forceinline__ double hb_entry_e3d(const double* Jp, const double* r, double w) {
  if constexpr (P < 45) {
    constexpr int a = tri_a(P), c = tri_c(P);
    if constexpr (c < 6) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) s += (w * Jp[k * 6 + a]) * Jp[k * 6 + c];
      return s;
    } else if constexpr (a == c) {
      return w;
    } else {
      return 0.0;
    }
  } else if constexpr (P < 51) {
    constexpr int a = P - 45;
    double s = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) s -= Jp[k * 6 + a] * (w * r[k]);
    return s;
  } else if constexpr (P < 54) {
    return -(w * r[P - 45]);
  } else {
    return 0.0;
  }
}
template <int G, int... T>
__device__ __forceinline__ void fill_group_e3d(double* v, const double* Jp, const double* r, double w, std::integer_sequence<int, T...>) {
  ((v[T] = hb_entry_e3d<G * 18 + T>(Jp, r, w)), ...);
}
// ---- wave reductions through an LDS transpose -----------------------------------------------------------------------------
// (A shuffle-based reduce-scatter -- recursive halving, 4 v_cndmask + 2 ds_bpermute + 1 add per exchange -- was tried first: 40 %
// of the analytic bbox kernel's issue slots were selects and permutes.)  Every lane parks its 18 values in a wave-private LDS tile tr[18][65] (row =
// entry, column = lane; the odd stride keeps the column sums conflict-free) and 54 lanes add up a third of a row each:
// 18 ds_write + 22 ds_read + 22 adds per group instead of ~210 instructions.  LDS operations of one wave execute in order,
// so no barrier is needed, only the compiler must not reorder across the hand-over.
constexpr int kTrStride = 65, kTrDoubles = 18 * kTrStride;
template <int D, int G>
__device__ __forceinline__ void reduce_group_lds(const double* J, const double* r, double w, int lane, double* __restrict__ out,
                                                 double* __restrict__ tr) {
  double v[18];
  fill_group<D, G>(v, J, r, w, std::make_integer_sequence<int, 18>{});
#pragma unroll
  for (int k = 0; k < 18; ++k) tr[k * kTrStride + lane] = v[k];
  __builtin_amdgcn_wave_barrier();
  const int k = lane % 18, p = lane / 18;            // lanes 0..53: third p of row k (22, 22, 20 columns)
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
  __builtin_amdgcn_wave_barrier();                   // the tile is rewritten by the next group
  if (lane < 18) out[G * 18 + lane] = (s + sa) + sb;
}
// 3-D edges: two 32-lane chunks per wave; `lane64` is the lane in the wave, the chunk's columns are its own 32
template <int G>
__device__ __forceinline__ void reduce_group_e3d_lds(const double* Jp, const double* r, double w, int lane64, double* __restrict__ out,
                                                     double* __restrict__ tr) {
  double v[18];
  fill_group_e3d<G>(v, Jp, r, w, std::make_integer_sequence<int, 18>{});
#pragma unroll
  for (int k = 0; k < 18; ++k) tr[k * kTrStride + lane64] = v[k];
  __builtin_amdgcn_wave_barrier();
  const int t = lane64 & 31, half = lane64 & 32;
  double s0 = 0, s1 = 0;
  if (t < 18) {
    const double* row = tr + t * kTrStride + half;
#pragma unroll
    for (int i = 0; i < 32; i += 2) { s0 += row[i]; s1 += row[i + 1]; }
  }
  __builtin_amdgcn_wave_barrier();
  if (t < 18) out[G * 18 + t] = s0 + s1;
}

// One instantiation per edge type: the bbox and the 3-D code paths have very different register needs, and a
// kernel is allocated for the worse of its branches (the fused version spilled 304 B/lane = 69 MB of HBM
// writes per launch at C4, profiles/r1_pmc_traffic.json).  `ids` lists the chunks of this type.
// VALIDATE (bbox only): this is the first linearisation of a run and doubles as the reference's NaN pre-check of the
// bbox edges (Optimizer.cpp:234-243): an edge whose residual is NaN at the start state is marked invalid for the
// whole run and counted in *n_dropped (same residual code as k_bbox_validate, which the synchronous API still uses).
template <int JAC, int TYPE, int VALIDATE = 0, bool TANG = false>
__device__ __forceinline__ void chunk_linearize_body(const DevGraph& g, const ChunkTable& ct, const int* __restrict__ ids, int n_ids,
                                                     const double* __restrict__ cams, const double* __restrict__ objs, double delta,
                                                     double* __restrict__ chunk_out, double* __restrict__ wg_chi /* LDS, 8 slots */,
                                                     double* __restrict__ tr_all /* LDS, kTrDoubles per wave */, int block,
                                                     int* __restrict__ n_dropped = nullptr) {
  double* tr = tr_all + (threadIdx.x >> 6) * kTrDoubles;
  // bbox chunks hold <= 64 edges (one wave each); 3-D chunks hold <= 32 edges and TWO of them share a wave
  // (an ellipsoid has ~20 3-D edges: a whole wave per chunk ran at 31 % lane use).  Offsets < 32 keep the
  // shuffles of the reduce-scatter inside a half wave.
  constexpr int kSeg = (TYPE == 1) ? 32 : 64;
  const int lane = threadIdx.x & (kSeg - 1);
  const int seg = (block * kLinWaves + (threadIdx.x >> 6)) * (64 / kSeg) + ((threadIdx.x & 63) / kSeg);
  const bool seg_on = seg < n_ids;
  if (!seg_on) return;   // (3-D: the other half wave carries on; its shuffles never leave its 32 lanes)
  const int ch = ids[seg];
  const int o = ct.obj[ch];
  constexpr int type = TYPE;
  const int i = ct.begin[ch] + lane;
  const bool in = seg_on && i < ct.end[ch];
  const Ell e = ell_load(objs + 10 * o);
  double* out = chunk_out + (size_t)ch * kChunkOut;
  double chi = 0;
  if (type == 0) {
    double r[4] = {0, 0, 0, 0}, J[36], w = 0;
    const bool act = in && (VALIDATE || g.bb_valid[i]);
    if (JAC == ESL_JAC_NUMERIC) {
      // g2o perturbs the ELLIPSOID: the 18 states ell_oplus(e, +-delta e_d) are the same for every edge of the chunk, so
      // 18 lanes compute one each (same code, same values) and park them in the wave's LDS tile; every lane used to
      // redo all 18 retractions (exp, quaternion products, normalisations: half of the kernel's 10k instructions)
      const int l64 = threadIdx.x & 63;
      if (l64 < 18) {
        double u[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 9; ++q) u[q] = (q == (l64 >> 1)) ? ((l64 & 1) ? -delta : delta) : 0.0;
        ell_store(ell_oplus(e, u), tr + 10 * l64);
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (act) {
      const SE3 T = se3_load(cams + 7 * g.bb_cam[i]);
      double meas[4] = {g.bb_meas[4 * i], g.bb_meas[4 * i + 1], g.bb_meas[4 * i + 2], g.bb_meas[4 * i + 3]};
      w = g.bb_w[i];
      if (JAC == ESL_JAC_ANALYTIC) {
        if (TANG) jac_tangency_t<true, false>(T, e, g.K, meas, r, J, nullptr);
        else jac_bbox_t<true, false>(T, e, g.K, meas, r, J, nullptr);
      } else {
        // the residual kind is a template parameter here too: a run-time branch in front of each of the 19 evaluations cost
        // the numeric kernel 6 % (4,740 vs 5,070 LM it/s at C4)
        auto res = [&](const Ell& ee, double* out) { if (TANG) res_tangency(T, ee, g.K, meas, out); else res_bbox(T, ee, g.K, meas, out); };
        res(e, r);
        const double scalar = 1.0 / (2 * delta);
        for (int d = 0; d < 9; ++d) {   // not unrolled: one body, 9 trips
          double rp[4], rm[4];
          res(ell_load(tr + 20 * d), rp);
          res(ell_load(tr + 20 * d + 10), rm);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int q = 0; q < 9; ++q) if (q == d) J[k * 9 + q] = scalar * (rp[k] - rm[k]);
          }
        }
      }
      chi = w * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
      if (VALIDATE) {
        // VALIDATE == 2 adds checkVisibility (esl_graph::check_visibility); as a run-time flag inside the NaN-only form it cost the
        // first linearisation of EVERY run a third of its occupancy (203 instead of 144 registers)
        const bool bad = (chi != chi) || (VALIDATE == 2 && !bbox_edge_visible(T, e, g.K, g.img_rows, g.img_cols));
        g.bb_valid[i] = bad ? 0 : 1;
        if (bad) {   // dropped: contributes nothing (its J / r may hold NaN)
          atomicAdd(n_dropped, 1);
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
    reduce_group_lds<4, 0>(J, r, w, lane, out, tr);
    reduce_group_lds<4, 1>(J, r, w, lane, out, tr);
    reduce_group_lds<4, 2>(J, r, w, lane, out, tr);
  } else {
    double r[9], Jp[36], w = 0;
    if (in) {
      const SE3 T = se3_load(cams + 7 * g.e3_cam[i]);
      const Ell m = ell_load(g.e3_meas + 10 * i);
      w = g.e3_w[i];
      if (JAC == ESL_JAC_ANALYTIC) jac_e3d_pose(T, e, m, g.yt, r, Jp);
      else {
        // g2o's central differences for the 6 pose columns; the scale block of this edge is exactly the identity.
        // The ellipsoid is perturbed on the right (T_est exp(u)), so E_0 exp(u) is the perturbed relative pose: E_0 is
        // formed once instead of 13 times (the 13 full chains spilled 1.1 KB of scratch per lane).
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
    reduce_group_e3d_lds<0>(Jp, r, w, threadIdx.x & 63, out, tr);
    reduce_group_e3d_lds<1>(Jp, r, w, threadIdx.x & 63, out, tr);
    reduce_group_e3d_lds<2>(Jp, r, w, threadIdx.x & 63, out, tr);
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) chi += __shfl_xor(chi, off, 64);
    if (lane == 0 && seg_on) {
      out[54] = chi;
      wg_chi[(threadIdx.x >> 6) * 2 + ((threadIdx.x & 63) >> 5)] = chi;
    }
    return;
  }
  chi = wave_sum(chi);
  if (lane == 0) {
    out[54] = chi;
    wg_chi[(threadIdx.x >> 6) * 2] = chi;
  }
}
// chi2 of the chunks one workgroup linearised, summed in slot order -> blk_chi[workgroup]  (the LM step kernel adds
// these ~1.3k numbers instead of one per chunk)
__device__ __forceinline__ void wg_chi_begin(double* wg_chi) {
  if (threadIdx.x < 2 * kLinWaves) wg_chi[threadIdx.x] = 0;
  __syncthreads();
}
__device__ __forceinline__ void wg_chi_end(const double* wg_chi, double* __restrict__ blk_chi, int slot) {
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
#pragma unroll
    for (int k = 0; k < 2 * kLinWaves; ++k) t += wg_chi[k];
    blk_chi[slot] = t;
  }
}

// fixed-order block reduction helpers for the "last workgroup finishes" pattern
struct LmScalars {  // lives in mapped host memory
  double chi2_lin, max_diag, chi2_trial, scale, ok, lambda_used, pad0, pad1;
};

// Device-resident Levenberg-Marquardt control (g2o's OptimizationAlgorithmLevenberg::solve + the ORB-SLAM style
// stop rule of Optimizer.cpp): the accept / reject decision, the lambda update and the stop test of every trial run
// at the head of k_lm_step, so the host only enqueues launches and never sits between two kernels.
struct LmCore {
  double lambda, ni, currentChi, iniChi, chi2_initial, rho;
  int cur;            // which (ellipsoid buffer, chunk buffer) pair holds the current estimate and its system
  int it, qmax, nBad, total_trials, done, stop_reason, trial_seq;
};
struct LmHostView {   // mapped host memory, written by the device only: progress flags the host polls + the results
  int seq, done;
  int n_dropped, pad0;   // bbox edges dropped by the NaN pre-check (written by k_chunk_finalize)
  LmCore core;        // final state (valid once done)
  double trace_chi2[ESL_MAX_TRACE], trace_lambda[ESL_MAX_TRACE];
  int trace_trials[ESL_MAX_TRACE];
  int trace_len, pad;
  long long dbg_clk[8];   // diagnostic (ESL_LM_TIMING): wall_clock64 marks of workgroup 0 in the last live k_lm_step
};

__device__ __forceinline__ void lm_core_init(LmCore& s, double chi2, double max_diag, double tau) {
  s.lambda = tau * max_diag;   // computeLambdaInit
  s.ni = 2; s.currentChi = chi2; s.iniChi = chi2; s.chi2_initial = chi2; s.rho = 0;
  s.cur = 0; s.it = 0; s.qmax = 0; s.nBad = 0; s.total_trials = 0; s.done = 0; s.stop_reason = 0; s.trial_seq = 0;
}

// Publish this workgroup's results and learn whether it is the last one.  Recipe of the CDNA guide (G16):
// plain stores -> __syncthreads() -> ONE lane: agent-scope release + drained counter increment; the last
// workgroup does ONE agent-scope acquire before reading the others' results.  (A __threadfence() in every
// thread costs ~10-50 us per workgroup on gfx950.)
__device__ __forceinline__ bool last_block_arrives(unsigned int* ticket) {
  __shared__ bool last;
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = (t == gridDim.x - 1);
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return last;
}
// Variant for payloads that were stored WRITE-THROUGH (relaxed agent-scope atomic stores = sc1): no release fence,
// every storing wave only drains its stores (G16 recipe R1); ~5 us per workgroup cheaper than the release form.
__device__ __forceinline__ bool last_block_arrives_wt(unsigned int* ticket) {
  __shared__ bool last_wt;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_wt = (t == gridDim.x - 1);
    if (last_wt) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return last_wt;
}
__device__ double block256_sum(double v, double* sm) {
  sm[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
    __syncthreads();
  }
  const double r = sm[0];
  __syncthreads();
  return r;
}
__device__ double block256_max(double v, double* sm) {
  sm[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]);
    __syncthreads();
  }
  const double r = sm[0];
  __syncthreads();
  return r;
}

// Buffer selection shared by the LM kernels: without `st` a launch works on (objs_a -> chunk_a); with it, the
// device-resident LM state says which pair is current and the linearisation point is the TRIAL buffer.
//
// Both edge types in ONE launch: the first nb_e3 workgroups take the 3-D chunks (longest instruction stream first),
// the rest the bbox chunks.  Launched back to back the two kernels ran 18 + 17.5 us at C4 although neither fills the
// chip for long (3-D: ~0.7 waves per SIMD on a 4.8k-instruction stream; bbox: 3 waves per SIMD, 1.7k instructions);
// together they take 27 us.
template <int JAC, int VALIDATE = 0, bool TANG = false>
static __global__ __launch_bounds__(64 * kLinWaves, ESL_LIN_MIN_WAVES) void k_chunk_linearize_both(DevGraph g, ChunkTable ct, const int* __restrict__ ids_e3, int n_e3,
                                                                     int nb_e3, const int* __restrict__ ids_bb, int n_bb,
                                                                     const double* __restrict__ cams,
                                                                     const double* __restrict__ objs_a, const double* __restrict__ objs_b,
                                                                     double delta, double* __restrict__ chunk_a, double* __restrict__ chunk_b,
                                                                     double* __restrict__ blk_chi, const LmCore* __restrict__ st,
                                                                     int* __restrict__ n_dropped) {
  __shared__ double wg_chi[2 * kLinWaves];
  __shared__ double tr_all[kLinWaves * kTrDoubles];
  const double* objs = objs_a;
  double* chunk_out = chunk_a;
  if (st) {
    if (st->done) return;
    if (st->cur == 0) { objs = objs_b; chunk_out = chunk_b; }   // trial = the pair that is NOT current
  }
  wg_chi_begin(wg_chi);
  if ((int)blockIdx.x < nb_e3) chunk_linearize_body<JAC, 1>(g, ct, ids_e3, n_e3, cams, objs, delta, chunk_out, wg_chi, tr_all, blockIdx.x);
  else chunk_linearize_body<JAC, 0, VALIDATE, TANG>(g, ct, ids_bb, n_bb, cams, objs, delta, chunk_out, wg_chi, tr_all, blockIdx.x - nb_e3, n_dropped);
  wg_chi_end(wg_chi, blk_chi, blockIdx.x);
}
template <int JAC, int TYPE, int VALIDATE = 0, bool TANG = false>
static __global__ __launch_bounds__(64 * kLinWaves) void k_chunk_linearize(DevGraph g, ChunkTable ct, const int* __restrict__ ids, int n_ids,
                                                                const double* __restrict__ cams,
                                                                const double* __restrict__ objs_a, const double* __restrict__ objs_b,
                                                                double delta, double* __restrict__ chunk_a, double* __restrict__ chunk_b,
                                                                double* __restrict__ blk_chi, int blk_offset,
                                                                const LmCore* __restrict__ st, int* __restrict__ n_dropped) {
  __shared__ double wg_chi[2 * kLinWaves];
  __shared__ double tr_all[kLinWaves * kTrDoubles];
  const double* objs = objs_a;
  double* chunk_out = chunk_a;
  if (st) {
    if (st->done) return;
    if (st->cur == 0) { objs = objs_b; chunk_out = chunk_b; }
  }
  wg_chi_begin(wg_chi);
  chunk_linearize_body<JAC, TYPE, VALIDATE, TANG>(g, ct, ids, n_ids, cams, objs, delta, chunk_out, wg_chi, tr_all, blockIdx.x, n_dropped);
  wg_chi_end(wg_chi, blk_chi, blk_offset + blockIdx.x);
}

__device__ __forceinline__ void obj_solve_one(const DevGraph& g, const ChunkTable& ct, const double* __restrict__ chunk_out,
                                              const double* __restrict__ objs, int jac, double delta, double lambda,
                                              double* __restrict__ xo, double* __restrict__ objs_trial, double* __restrict__ part, int o,
                                              double& cg_out, double& scale_out, double& ok_out,
                                              const double* __restrict__ hb_sum = nullptr /* 54 pre-summed entries (LDS) */) {
  const int c0 = ct.ostart[o], c1 = ct.ostart[o + 1];
  const Ell e = ell_load(objs + 10 * o);
  const int ngrav = g.gr_cnt[o];
  if (c0 == c1 && ngrav == 0) {  // inactive vertex: never touched (sparse_optimizer.cpp:236-257)
    ell_store(e, objs_trial + 10 * o);
    part[o * 4 + 0] = 0; part[o * 4 + 2] = 0; part[o * 4 + 3] = 1;
    return;   // contributes cg = 0, scale = 0, ok = 1
  }
  double hb[54];
  if (hb_sum) {
#pragma unroll
    for (int k = 0; k < 54; ++k) hb[k] = hb_sum[k];
  } else {
#pragma unroll
    for (int k = 0; k < 54; ++k) hb[k] = 0;
    for (int ch = c0; ch < c1; ++ch) {
      const double* p = chunk_out + (size_t)ch * kChunkOut;
#pragma unroll
      for (int k = 0; k < 54; ++k) hb[k] += p[k];
    }
  }
  const double wg = g.grav_w * ngrav;
  if (ngrav > 0) {  // the gravity prior is a unary edge on this ellipsoid: linearise it here
    double Jg[9], rg;
    if (jac == ESL_JAC_ANALYTIC) rg = jac_grav(e, g.grav_n, Jg);
    else {
      rg = res_grav(e, g.grav_n);
      numeric_jac_obj(e, delta, 1, Jg, [&](const Ell& ep, double* o1) { o1[0] = res_grav(ep, g.grav_n); });
    }
    int p = 0;
#pragma unroll
    for (int a = 0; a < 9; ++a) {
#pragma unroll
      for (int cc = a; cc < 9; ++cc) hb[p++] += (wg * Jg[a]) * Jg[cc];
      hb[45 + a] -= Jg[a] * (wg * rg);
    }
  }
  double x[9];
  const bool ok = ldlt_solve_packed<9>(hb, lambda, hb + 45, x);
  double scale = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) { scale += x[i] * (lambda * x[i] + hb[45 + i]); xo[(size_t)o * 9 + i] = x[i]; }
  const Ell en = ell_oplus(e, x);
  ell_store(en, objs_trial + 10 * o);
  double cg = 0;
  if (ngrav > 0) { const double rg = res_grav(en, g.grav_n); cg = wg * rg * rg; }
  part[o * 4 + 0] = cg;   // chi2 of the gravity prior at the trial state
  part[o * 4 + 2] = scale;
  part[o * 4 + 3] = ok ? 1.0 : 0.0;
  cg_out = cg; scale_out = scale; ok_out = ok ? 1.0 : 0.0;
}


// ONE LANE PER ELLIPSOID: H, b from the chunk partials (fixed order => deterministic), gravity prior added,
// unrolled register LDL^T of the 9x9, retraction, trial state.  (A wave per ellipsoid left 63 lanes idle on a
// serial dependency chain; 2k ellipsoids are 32 full waves this way.)
// lambda < 0: lambda = tau * max_diag read from device memory (first LM iteration, computeLambdaInit).
static __global__ __launch_bounds__(64) void k_obj_solve(DevGraph g, ChunkTable ct, const double* __restrict__ chunk_out,
                                                         const double* __restrict__ objs, int jac, double delta, double lambda, double tau,
                                                         const double* __restrict__ dev_scal, double* __restrict__ xo,
                                                         double* __restrict__ objs_trial, double* __restrict__ part,
                                                         double* __restrict__ solve_part /* gridDim x 4 */) {
  if (lambda < 0) lambda = tau * dev_scal[1];
  const int o = blockIdx.x * 64 + threadIdx.x;
  double cg = 0, scale = 0, okd = 1;
  if (o < g.n_objs) obj_solve_one(g, ct, chunk_out, objs, jac, delta, lambda, xo, objs_trial, part, o, cg, scale, okd);
  // this workgroup's share of the trial scalars (fixed order: shuffle tree over the 64 lanes)
  cg = wave_sum(cg); scale = wave_sum(scale);
  okd = -wave_max(-okd);
  if (threadIdx.x == 0) {
    solve_part[blockIdx.x * 4 + 0] = cg; solve_part[blockIdx.x * 4 + 1] = 0;
    solve_part[blockIdx.x * 4 + 2] = scale; solve_part[blockIdx.x * 4 + 3] = okd;
  }
}

// block-wide reductions with ONE barrier pair: wave shuffle tree, then every thread adds the 4 wave totals in wave order
__device__ __forceinline__ double block256_sum1(double v, double* sm4) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm4[threadIdx.x >> 6] = v;
  __syncthreads();
  return ((sm4[0] + sm4[1]) + sm4[2]) + sm4[3];
}
__device__ __forceinline__ double block256_min1(double v, double* sm4) {
  v = -wave_max(-v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm4[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmin(fmin(sm4[0], sm4[1]), fmin(sm4[2], sm4[3]));
}

// One LM trial's accept / reject decision.  Control flow restated from g2o optimization_algorithm_levenberg.cpp:69-141
// (rho, lambda update, <= max_trials per iteration), the iteration loop of sparse_optimizer.cpp:357-425 and the
// "nBad >= 3" early stop; same order of operations as the host loop in esl_optimize_resident.  Every thread of every
// workgroup evaluates it on the same inputs (identical result); `writer` alone records trace / progress for the host.
__device__ __forceinline__ void lm_decide(LmCore& s, double chi, double sc, double okv, int max_iters, int max_trials, bool writer,
                                          LmHostView* __restrict__ host) {
  const double tempChi = (okv > 0.5) ? chi : DBL_MAX;
  const double rho = (s.currentChi - tempChi) / (sc + 1e-3);
  if (rho > 0 && isfinite(tempChi)) {
    const double t3 = 2 * rho - 1;
    double alpha = 1. - t3 * t3 * t3;   // g2o: pow(2 rho - 1, 3); the library pow() is ~200 dependent instructions here
    alpha = fmin(alpha, 2. / 3.);
    s.lambda *= fmax(1. / 3., alpha);
    s.ni = 2;
    s.currentChi = tempChi;
    s.cur ^= 1;          // discardTop: the trial pair (state + its linearisation) becomes the current one
  } else {
    s.lambda *= s.ni;    // pop: the current pair is untouched
    s.ni *= 2;
  }
  s.rho = rho;
  s.qmax++;
  s.trial_seq++;
  if (!(rho < 0 && s.qmax < max_trials)) {   // this LM iteration is over
    s.total_trials += s.qmax;
    if (writer && s.it < ESL_MAX_TRACE) {
      host->trace_chi2[s.it] = s.currentChi; host->trace_lambda[s.it] = s.lambda; host->trace_trials[s.it] = s.qmax;
      host->trace_len = s.it + 1;
    }
    if (s.qmax == max_trials || rho == 0) { s.done = 1; s.stop_reason = 1; }
    else {
      if ((s.iniChi - s.currentChi) * 1e3 < s.iniChi) s.nBad++; else s.nBad = 0;
      if (s.nBad >= 3) { s.done = 1; s.stop_reason = 2; }
    }
    s.it++;
    if (s.it >= max_iters) s.done = 1;
    s.iniChi = s.currentChi;
    s.qmax = 0;
  }
}

// One LM trial, device-driven (k_lm_step_rows below): (1) decide the PREVIOUS trial from the chi2 its linearisation left in
// blk_chi and the partials the previous step launch left in sp_in -- every workgroup does this redundantly, so there is no
// grid-wide handshake; (2) solve (H + lambda I) x = b per ellipsoid for the new lambda from the now-current chunk partials and
// write the next trial state.  LM state ping-pongs between `in` and `out` (other workgroups may still be reading `in`).
// first != 0: nothing to decide yet (state initialised by k_chunk_finalize_rows).
// Sharded run (n_ranks > 0): `gathered` holds every rank's 8-double block {chi2_lin, max_diag, chi2_trial, scale, ok,
// has_edges, -, -} (all-gathered on this stream); the decision adds them in rank order, so every rank decides alike.
// first: 1 = state in `in` is initialised (single GPU), 2 = initialise it here from the gathered linearisation scalars.
__device__ __forceinline__ double lm_readlane(double v, int src) {   // src: compile-time constant after unrolling
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
// (H + lambda I) x = b for one 9x9 block, ROW-PARALLEL: lane i (< 9) of the calling wave owns row i.  Right-looking LDL^T
// without pivoting, pivots / rows broadcast with v_readlane; the triangular solves run on the same 9 lanes (the factor is
// transposed once through LDS so that lane i also owns column i).  One lane with the whole matrix in registers
// (ldlt_solve_packed, esl_math.hpp) needs ~250 VGPRs and a 400-deep dependent chain; this needs 9 + 9 doubles and ~100 steps.
// hb: 45 packed upper + 9 rhs (LDS); tl: 81 doubles of LDS scratch of this wave; returns x[0..8] in every lane and
// `ok` = all pivots positive (Eigen's LDLT::isPositive, solvers/linear_solver_dense.h:107-112).
__device__ __forceinline__ bool lm_solve9_rows(const double* __restrict__ hb, double lambda, int lane, double* __restrict__ tl, double x[9]) {
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
    const double dk = lm_readlane(a[k], k);
    ok = ok && (dk > 0);
    if (i == k) d = dk;
    const double lik = a[k] * (1.0 / dk);
#pragma unroll
    for (int j = k + 1; j < 9; ++j) {
      const double rkj = lm_readlane(a[j], k);
      if (i > k) a[j] -= lik * rkj;
    }
    if (i > k) a[k] = lik;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {   // L y = b
    const double yk = lm_readlane(y, k);
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
    const double xk = lm_readlane(z, k);
    if (i < k) z -= c[k] * xk;
  }
#pragma unroll
  for (int j = 0; j < 9; ++j) x[j] = lm_readlane(z, j);
  return ok;
}

// The per-ellipsoid part runs on a WAVE per ellipsoid.  (Round 1's form gathered the chunk partials of 16 ellipsoids with all
// 256 threads -- (ellipsoid, entry) pairs, 3.4 us of scattered loads -- and then solved on 16 lanes with the whole 9x9 system
// in registers: 254 VGPRs, 3.9 us.)  Here lane k < 54 of the wave adds entry k over the ellipsoid's chunks
// (one coalesced 432-byte row per chunk, the rows of BOTH state pairs already in flight while the decision is taken), the
// gravity prior is linearised uniformly by the wave, and the system is solved row-parallel on lanes 0..8 (lm_solve9_rows).
constexpr int kStepWaves = 4;
// First iteration: chi2 of the linearisation point and max |H_kk| (computeLambdaInit).  One WAVE per ellipsoid: the chunk rows
// are read as coalesced rows (lane = entry, up to four rows in flight); the last workgroup reduces the per-workgroup partials in
// fixed order.  12.5 us (once per run): the ticket and the serial tail are what the time is, not the per-ellipsoid part.
static __global__ __launch_bounds__(64 * kStepWaves) void k_chunk_finalize_rows(DevGraph g, ChunkTable ct, const double* __restrict__ chunk_out,
                                                                                 const double* __restrict__ objs, int jac, double delta,
                                                                                 double* __restrict__ blk_part /* gridDim x 2 */,
                                                                                 unsigned int* __restrict__ ticket, double* __restrict__ dev_scal,
                                                                                 LmScalars* __restrict__ host, double tau, LmCore* __restrict__ st,
                                                                                 int* __restrict__ n_dropped, int n_grav,
                                                                                 LmHostView* __restrict__ hv) {
  __shared__ double sm[256];
  __shared__ double wpart[2 * kStepWaves];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int o = blockIdx.x * kStepWaves + wave;
  double chi = 0, md = 0;
  if (o < g.n_objs) {
    const int c0 = ct.ostart[o], c1 = ct.ostart[o + 1];
    double v = 0;                                   // lane < 54: entry `lane` of the summed system, lane 54: chi2
    if (lane < 55) {
      double r4[4] = {0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (c0 + q < c1) r4[q] = chunk_out[(size_t)(c0 + q) * kChunkOut + lane];
      v = ((r4[0] + r4[1]) + r4[2]) + r4[3];
      for (int ch = c0 + 4; ch < c1; ++ch) v += chunk_out[(size_t)ch * kChunkOut + lane];
    }
    // position of diagonal entry a in the packed upper triangle: 0 9 17 24 30 35 39 42 44
    int a_of = -1;
#pragma unroll
    for (int a = 0, q = 0; a < 9; ++a) { if (lane == q) a_of = a; q += 9 - a; }
    double gchi = 0;
    const int ng = g.gr_cnt[o];
    if (ng > 0) {   // gravity prior lives on the ellipsoid itself (see k_obj_solve); evaluated uniformly by the wave
      const Ell e = ell_load(objs + 10 * o);
      const double wg = g.grav_w * ng;
      double Jg[9], rg;
      if (jac == ESL_JAC_ANALYTIC) rg = jac_grav(e, g.grav_n, Jg);
      else {
        rg = res_grav(e, g.grav_n);
        numeric_jac_obj(e, delta, 1, Jg, [&](const Ell& ep, double* o1) { o1[0] = res_grav(ep, g.grav_n); });
      }
      gchi = wg * rg * rg;
#pragma unroll
      for (int a = 0; a < 9; ++a) if (a_of == a) v += wg * Jg[a] * Jg[a];
    }
    md = wave_max((a_of >= 0) ? fabs(v) : 0.0);
    chi = lm_readlane(v, 54) + gchi;
  }
  if (lane == 0) { wpart[wave] = chi; wpart[kStepWaves + wave] = md; }
  __syncthreads();
  if (tid == 0) {
    double c = 0, m = 0;
#pragma unroll
    for (int k = 0; k < kStepWaves; ++k) { c += wpart[k]; m = fmax(m, wpart[kStepWaves + k]); }
    __hip_atomic_store(&blk_part[2 * blockIdx.x], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&blk_part[2 * blockIdx.x + 1], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (last_block_arrives_wt(ticket)) {
    double c = 0, m = 0;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += 256) { c += blk_part[2 * b]; m = fmax(m, blk_part[2 * b + 1]); }
    c = block256_sum(c, sm);
    m = block256_max(m, sm);
    if (threadIdx.x == 0) {
      dev_scal[0] = c; dev_scal[1] = m;
      host->chi2_lin = c; host->max_diag = m;
      if (hv) {
        const int nd = *n_dropped;
        hv->n_dropped = nd;
        *n_dropped = 0;
        const bool any_edge = (g.n_bbox_edges - nd > 0) || g.n_e3d > 0 || n_grav > 0;
        dev_scal[5] = any_edge ? 1.0 : 0.0;
        if (st) {
          LmCore s;
          lm_core_init(s, c, m, tau);
          if (!any_edge) {
            s.done = 1; s.stop_reason = 3;
            hv->core = s;
            hv->trace_len = 0;
            __hip_atomic_store(&hv->done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
          }
          *st = s;
        }
      }
      *ticket = 0;
    }
  }
}

static __global__ __launch_bounds__(64 * kStepWaves) void k_lm_step_rows(DevGraph g, ChunkTable ct, const double* __restrict__ chunk_a,
                                                                          const double* __restrict__ chunk_b, double* __restrict__ objs_a,
                                                                          double* __restrict__ objs_b, const LmCore* __restrict__ in,
                                                                          LmCore* __restrict__ out, const double* __restrict__ blk_chi, int n_lin_blocks,
                                                                          const double* __restrict__ sp_in, double* __restrict__ sp_out, int first,
                                                                          int max_iters, int max_trials, LmHostView* __restrict__ host, int jac,
                                                                          double delta, double* __restrict__ xo, const double* __restrict__ gathered,
                                                                          int n_ranks, double tau) {
  __shared__ double sm12[3 * kStepWaves];
  __shared__ double hbs[kStepWaves][56];
  __shared__ double tls[kStepWaves][81];
  __shared__ double jgs[kStepWaves][12];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const bool writer = blockIdx.x == 0 && tid == 0;
  const int o = blockIdx.x * kStepWaves + wave;
  const bool own = o < g.n_objs;
  // the first four chunk rows of this wave's ellipsoid from BOTH pairs: which one is current is only known after the decision
  const int c0 = own ? ct.ostart[o] : 0, c1 = own ? ct.ostart[o + 1] : 0;
  double va[4] = {0, 0, 0, 0}, vb[4] = {0, 0, 0, 0};
  if (lane < 54) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (c0 + q < c1) { va[q] = chunk_a[(size_t)(c0 + q) * kChunkOut + lane]; vb[q] = chunk_b[(size_t)(c0 + q) * kChunkOut + lane]; }
  }
  // ... and its state from both pairs (one value per lane; broadcast after the decision)
  double ea = 0, eb = 0;
  int ngrav = 0;
  if (own) {
    if (lane < 10) { ea = objs_a[10 * (size_t)o + lane]; eb = objs_b[10 * (size_t)o + lane]; }
    ngrav = g.gr_cnt[o];
  }
  double c = 0, sc = 0, okv = 1;
  if (!first && n_ranks == 0) {
#pragma unroll 8
    for (int k = tid; k < n_lin_blocks; k += 64 * kStepWaves) c += blk_chi[k];
    for (int b = tid; b < (int)gridDim.x; b += 64 * kStepWaves) { c += sp_in[b * 4 + 0]; sc += sp_in[b * 4 + 2]; okv = fmin(okv, sp_in[b * 4 + 3]); }
  }
  LmCore s;
  if (first == 2) {
    double chi = 0, md = 0, any = 0;
    for (int r = 0; r < n_ranks; ++r) { chi += gathered[r * 8 + 0]; md = fmax(md, gathered[r * 8 + 1]); any = fmax(any, gathered[r * 8 + 5]); }
    lm_core_init(s, chi, md, tau);
    if (any < 0.5) {   // no rank has an active edge
      s.done = 1; s.stop_reason = 3;
      if (writer) {
        host->core = s;
        host->trace_len = 0;
        __hip_atomic_store(&host->done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  } else {
    s = *in;
  }
  if (s.done) {   // queued behind a finished run: pass the state on and leave
    if (writer) *out = s;
    return;
  }
  if (!first) {
    if (n_ranks > 0) {
      for (int r = 0; r < n_ranks; ++r) { c += gathered[r * 8 + 2]; sc += gathered[r * 8 + 3]; okv = fmin(okv, gathered[r * 8 + 4]); }
    } else {   // three block reductions behind one barrier pair
      c = wave_sum(c); sc = wave_sum(sc); okv = -wave_max(-okv);
      if (lane == 0) { sm12[wave] = c; sm12[kStepWaves + wave] = sc; sm12[2 * kStepWaves + wave] = okv; }
      __syncthreads();
      c = sm12[0]; sc = sm12[kStepWaves]; okv = sm12[2 * kStepWaves];
#pragma unroll
      for (int k = 1; k < kStepWaves; ++k) { c += sm12[k]; sc += sm12[kStepWaves + k]; okv = fmin(okv, sm12[2 * kStepWaves + k]); }
    }
    lm_decide(s, c, sc, okv, max_iters, max_trials, writer, host);
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
  const double* chunk_out = s.cur ? chunk_b : chunk_a;
  double* objs_trial = s.cur ? objs_a : objs_b;
  const double lambda = s.lambda;
  double cg = 0, scale = 0, okd = 1;
  if (own) {
    const double ev = s.cur ? eb : ea;
    double e10[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) e10[k] = lm_readlane(ev, k);
    Ell e = ell_load(e10);
    if (c0 < c1 || ngrav > 0) {   // an inactive vertex is never touched (sparse_optimizer.cpp:236-257)
      const double wg = g.grav_w * ngrav;
      if (ngrav > 0) {   // the gravity prior is a unary edge on this ellipsoid: linearised here (uniformly by the wave)
        double Jg[9], rg;
        if (jac == ESL_JAC_ANALYTIC) rg = jac_grav(e, g.grav_n, Jg);
        else {
          rg = res_grav(e, g.grav_n);
          numeric_jac_obj(e, delta, 1, Jg, [&](const Ell& ep, double* o1) { o1[0] = res_grav(ep, g.grav_n); });
        }
        if (lane == 0) {
#pragma unroll
          for (int k = 0; k < 9; ++k) jgs[wave][k] = Jg[k];
          jgs[wave][9] = rg;
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (lane < 54) {
        double v = 0;   // chunk order, as the serial sum
#pragma unroll
        for (int q = 0; q < 4; ++q) v += s.cur ? vb[q] : va[q];
        for (int ch = c0 + 4; ch < c1; ++ch) v += chunk_out[(size_t)ch * kChunkOut + lane];   // > 4 chunks: rare
        if (ngrav > 0) {
          const double* jg = jgs[wave];
          if (lane < 45) v += (wg * jg[tri_a(lane)]) * jg[tri_c(lane)];
          else v -= jg[lane - 45] * (wg * jg[9]);
        }
        hbs[wave][lane] = v;
      }
      __builtin_amdgcn_wave_barrier();
      const double* hb = hbs[wave];
      double x[9];
      const bool ok = lm_solve9_rows(hb, lambda, lane, tls[wave], x);
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
      if (ngrav > 0) { const double rg = res_grav(e, g.grav_n); cg = wg * rg * rg; }   // chi2 of the gravity prior at the trial state
    }
    if (lane == 0) ell_store(e, objs_trial + 10 * (size_t)o);
  }
  // this workgroup's share of the trial scalars, ellipsoids in order
  __syncthreads();
  if (lane == 0) { sm12[wave] = cg; sm12[kStepWaves + wave] = scale; sm12[2 * kStepWaves + wave] = okd; }
  __syncthreads();
  if (tid == 0) {
    double a = 0, b = 0, m = 1;
#pragma unroll
    for (int k = 0; k < kStepWaves; ++k) { a += sm12[k]; b += sm12[kStepWaves + k]; m = fmin(m, sm12[2 * kStepWaves + k]); }
    sp_out[blockIdx.x * 4 + 0] = a; sp_out[blockIdx.x * 4 + 1] = 0;
    sp_out[blockIdx.x * 4 + 2] = b; sp_out[blockIdx.x * 4 + 3] = m;
  }
}

// Sharded run: this rank's share of a trial's scalars -> dev_scal[2..4], the send buffer of the all-gather
static __global__ __launch_bounds__(256) void k_lm_partials(const double* __restrict__ blk_chi, int n_lin_blocks,
                                                            const double* __restrict__ sp, int n_step_blocks,
                                                            double* __restrict__ dev_scal) {
  __shared__ double sm4[4];
  double c = 0, sc = 0, okv = 1;
#pragma unroll 8
  for (int k = threadIdx.x; k < n_lin_blocks; k += 256) c += blk_chi[k];
  for (int b = threadIdx.x; b < n_step_blocks; b += 256) { c += sp[b * 4 + 0]; sc += sp[b * 4 + 2]; okv = fmin(okv, sp[b * 4 + 3]); }
  c = block256_sum1(c, sm4);
  sc = block256_sum1(sc, sm4);
  okv = block256_min1(okv, sm4);
  if (threadIdx.x == 0) { dev_scal[2] = c; dev_scal[3] = sc; dev_scal[4] = okv; }
}

// chi2 of the trial states, one instantiation per edge type (bbox: one wave per chunk; 3-D: two 32-edge chunks per
// wave).  FINAL: the last workgroup to arrive reduces all partials in fixed order -> mapped host memory.
template <int TYPE, bool FINAL>
static __global__ __launch_bounds__(256) void k_chunk_chi2(DevGraph g, ChunkTable ct, const int* __restrict__ ids, int n_ids,
                                                           const double* __restrict__ cams,
                                                           const double* __restrict__ objs_trial, const double* __restrict__ part,
                                                           double* __restrict__ chunk_chi, unsigned int* __restrict__ ticket,
                                                           double lambda, double tau, double* __restrict__ dev_scal,
                                                           LmScalars* __restrict__ host) {
  __shared__ double sm[256];
  constexpr int kSeg = (TYPE == 1) ? 32 : 64;
  const int lane = threadIdx.x & (kSeg - 1);
  const int seg = (blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / kSeg) + ((threadIdx.x & 63) / kSeg);
  const bool seg_on = seg < n_ids;
  if (seg_on) {
    const int ch = ids[seg];
    const int o = ct.obj[ch];
    const int i = ct.begin[ch] + lane;
    const bool in = seg_on && i < ct.end[ch];
    const Ell e = ell_load(objs_trial + 10 * o);
    double chi = 0;
    if (TYPE == 0) {
      if (in && g.bb_valid[i]) {
        double r[4];
        res_box_edge(g.bbox_mode, se3_load(cams + 7 * g.bb_cam[i]), e, g.K, g.bb_meas + 4 * i, r);
        chi = g.bb_w[i] * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
      }
    } else if (in) {
      double r[9];
      res_e3d(se3_load(cams + 7 * g.e3_cam[i]), e, ell_load(g.e3_meas + 10 * i), g.yt, r);
#pragma unroll
      for (int k = 0; k < 9; ++k) chi += r[k] * r[k];
      chi *= g.e3_w[i];
    }
#pragma unroll
    for (int off = kSeg / 2; off > 0; off >>= 1) chi += __shfl_xor(chi, off, 64);
    if (lane == 0 && seg_on) __hip_atomic_store(&chunk_chi[ch], chi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through
  }
  if (!FINAL) return;
  if (last_block_arrives_wt(ticket)) {
    double c = 0, sc = 0, okv = 1;
    for (int k = threadIdx.x; k < ct.n_chunks; k += 256) c += chunk_chi[k];
    for (int o = threadIdx.x; o < g.n_objs; o += 256) { c += part[o * 4 + 0]; sc += part[o * 4 + 2]; okv = fmin(okv, part[o * 4 + 3]); }
    c = block256_sum(c, sm);
    sc = block256_sum(sc, sm);
    okv = -block256_max(-okv, sm);
    if (threadIdx.x == 0) {
      host->chi2_trial = c; host->scale = sc; host->ok = okv;
      dev_scal[2] = c; dev_scal[3] = sc; dev_scal[4] = okv;   // send buffer of the RCCL exchange
      host->lambda_used = (lambda < 0) ? tau * dev_scal[1] : lambda;
      *ticket = 0;
    }
  }
}

// Hoo / b_o of every ellipsoid from the chunk partials (only used by esl_lm_download and the SLAM-free inspection path)
static __global__ void k_chunk_combine(DevGraph g, ChunkTable ct, const double* __restrict__ chunk_out,
                                       const double* __restrict__ objs, int jac, double delta, double* __restrict__ Hoo,
                                       double* __restrict__ bo) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int o = t / 54, k = t % 54;
  if (o >= g.n_objs) return;
  double s = 0;
  for (int ch = ct.ostart[o]; ch < ct.ostart[o + 1]; ++ch) s += chunk_out[(size_t)ch * kChunkOut + k];
  if (g.gr_cnt[o] > 0) {
    const Ell e = ell_load(objs + 10 * o);
    const double wg = g.grav_w * g.gr_cnt[o];
    double Jg[9], rg;
    if (jac == ESL_JAC_ANALYTIC) rg = jac_grav(e, g.grav_n, Jg);
    else {
      rg = res_grav(e, g.grav_n);
      numeric_jac_obj(e, delta, 1, Jg, [&](const Ell& ep, double* o1) { o1[0] = res_grav(ep, g.grav_n); });
    }
    if (k < 45) {
      int a = 0, base = 0;
      while (k >= base + (9 - a)) { base += 9 - a; ++a; }
      const int cc = a + (k - base);
      double ja = 0, jc = 0;
      for (int q = 0; q < 9; ++q) { if (q == a) ja = Jg[q]; if (q == cc) jc = Jg[q]; }
      s += (wg * ja) * jc;
    } else {
      double ja = 0;
      for (int q = 0; q < 9; ++q) if (q == k - 45) ja = Jg[q];
      s -= ja * (wg * rg);
    }
  }
  if (k < 45) Hoo[(size_t)o * 45 + k] = s;
  else bo[(size_t)o * 9 + (k - 45)] = s;
}

}  // namespace esl
