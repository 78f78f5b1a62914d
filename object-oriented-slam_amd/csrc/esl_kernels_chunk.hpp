// esl_kernels_chunk.hpp — mapping-mode kernels, second generation (the ones the LM driver launches).
//
// Why: the first cut (esl_kernels_map.hpp: one wave per ellipsoid, 54 accumulators per lane carried across
// rounds) ran at 1 wave/SIMD with scratch spills and a tail set by the ellipsoid with the most edges.
// Here the unit of work is a CHUNK = up to 64 edges of one type hanging on one ellipsoid (table built at
// upload, edges are already sorted by ellipsoid so a chunk is a contiguous, coalesced slice):
//   k_chunk_linearize   one wave per chunk: lane = edge; residual + Jacobian in registers, then the 45+9
//                       entries of J^T W J / -J^T W r are reduced across the wave with a transpose-reduce
//                       (reduce-scatter by recursive halving: ~21 exchanges per 18 values instead of 108),
//                       so nothing has to live across rounds -> ~4x the occupancy, uniform task size
//   k_chunk_finalize    (first LM iteration only) per-ellipsoid diagonal of H -> max diag (lambda_0), chi2
//   k_obj_solve         one wave per ellipsoid: sum its chunk partials in fixed order (deterministic),
//                       row-parallel LDL^T of the 9x9, retraction, trial state
//   k_chunk_chi2        one wave per chunk: chi2 of the trial state; the last workgroup to finish reduces
//                       everything in fixed order and writes the LM scalars straight into mapped host memory
// Reference pieces replaced: see esl_kernels_map.hpp.
#pragma once
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

// ---- reduce-scatter across the wave by recursive halving ------------------------------------------------
// In: N values per lane.  Out: the wave-wide total of entry `idx` (returned) in the lanes whose low bits are 0.
template <int N>
struct WaveRS {
  // cnt = number of REAL (non-padding) entries among the N this lane currently holds
  static __device__ __forceinline__ double run(const double* v, int lane, int& idx, int& cnt, int off) {
    constexpr int H = (N + 1) / 2;
    const bool hi = (lane & off) != 0;
    double nv[H];
#pragma unroll
    for (int i = 0; i < H; ++i) {
      const double lo_v = v[i];
      const double hi_v = (i + H < N) ? v[i + H] : 0.0;
      const double send = hi ? lo_v : hi_v;
      const double keep = hi ? hi_v : lo_v;
      nv[i] = keep + __shfl_xor(send, off, 64);
    }
    if (hi) { idx += H; cnt -= H; }
    else cnt = cnt < H ? cnt : H;
    return WaveRS<H>::run(nv, lane, idx, cnt, off >> 1);
  }
};
template <>
struct WaveRS<1> {
  static __device__ __forceinline__ double run(const double* v, int, int&, int&, int off) {
    double s = v[0];
    for (; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return s;
  }
};
// number of halving steps that distribute entries for N values (the remaining steps are plain adds)
constexpr int rs_steps(int n) { return n <= 1 ? 0 : 1 + rs_steps((n + 1) / 2); }

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
// half-wave version: `lane` is the lane inside the 32-lane segment; 18 -> 9 -> 5 -> 3 -> 2 -> 1 uses exactly the
// five offsets 16..1, so every lane of the segment ends up owning (at most) one entry
template <int G>
__device__ __forceinline__ void reduce_group_e3d(const double* Jp, const double* r, double w, int lane, double* __restrict__ out,
                                                 bool seg_on) {
  double v[18];
  fill_group_e3d<G>(v, Jp, r, w, std::make_integer_sequence<int, 18>{});
  int idx = 0, cnt = 18;
  const double tot = WaveRS<18>::run(v, lane, idx, cnt, 16);
  if (seg_on && cnt >= 1) out[G * 18 + idx] = tot;
}

// entries [G*18, G*18+18) reduced over the wave; the owning lanes write the totals
template <int D, int G>
__device__ __forceinline__ void reduce_group(const double* J, const double* r, double w, int lane, double* __restrict__ out) {
  double v[18];
  fill_group<D, G>(v, J, r, w, std::make_integer_sequence<int, 18>{});
  int idx = 0, cnt = 18;
  const double tot = WaveRS<18>::run(v, lane, idx, cnt, 32);
  constexpr int used = rs_steps(18);               // 5 distributing steps: offsets 32..2
  const int low_mask = (64 >> used) - 1;           // remaining low bits (here: bit 0)
  if ((lane & low_mask) == 0 && cnt >= 1) out[G * 18 + idx] = tot;   // cnt < 1: this lane ended on a padding slot
}

// One instantiation per edge type: the bbox and the 3-D code paths have very different register needs, and a
// kernel is allocated for the worse of its branches (the fused version spilled 304 B/lane = 69 MB of HBM
// writes per launch at C4, profiles/r1_pmc_traffic.json).  `ids` lists the chunks of this type.
template <int JAC, int TYPE>
static __global__ __launch_bounds__(256) void k_chunk_linearize(DevGraph g, ChunkTable ct, const int* __restrict__ ids, int n_ids,
                                                                const double* __restrict__ cams,
                                                                const double* __restrict__ objs, double delta,
                                                                double* __restrict__ chunk_out) {
  // bbox chunks hold <= 64 edges (one wave each); 3-D chunks hold <= 32 edges and TWO of them share a wave
  // (an ellipsoid has ~20 3-D edges: a whole wave per chunk ran at 31 % lane use).  Offsets < 32 keep the
  // shuffles of the reduce-scatter inside a half wave.
  constexpr int kSeg = (TYPE == 1) ? 32 : 64;
  const int lane = threadIdx.x & (kSeg - 1);
  const int seg = (blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / kSeg) + ((threadIdx.x & 63) / kSeg);
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
    const bool act = in && g.bb_valid[i];
    if (act) {
      const SE3 T = se3_load(cams + 7 * g.bb_cam[i]);
      double meas[4] = {g.bb_meas[4 * i], g.bb_meas[4 * i + 1], g.bb_meas[4 * i + 2], g.bb_meas[4 * i + 3]};
      w = g.bb_w[i];
      if (JAC == ESL_JAC_ANALYTIC) jac_bbox_t<true, false>(T, e, g.K, meas, r, J, nullptr);
      else {
        res_bbox(T, e, g.K, meas, r);
        numeric_jac_obj(e, delta, 4, J, [&](const Ell& ep, double* o4) { res_bbox(T, ep, g.K, meas, o4); });
      }
      chi = w * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
    } else {
#pragma unroll
      for (int k = 0; k < 36; ++k) J[k] = 0;
    }
    reduce_group<4, 0>(J, r, w, lane, out);
    reduce_group<4, 1>(J, r, w, lane, out);
    reduce_group<4, 2>(J, r, w, lane, out);
  } else {
    double r[9], Jp[36], w = 0;
    if (in) {
      const SE3 T = se3_load(cams + 7 * g.e3_cam[i]);
      const Ell m = ell_load(g.e3_meas + 10 * i);
      w = g.e3_w[i];
      if (JAC == ESL_JAC_ANALYTIC) jac_e3d_pose(T, e, m, g.yt, r, Jp);
      else {
        // g2o's central differences for the 6 pose columns; the scale block of this edge is exactly the identity
        res_e3d(T, e, m, g.yt, r);
        const double scalar = 1.0 / (2 * delta);
#pragma unroll
        for (int d = 0; d < 6; ++d) {
          double u[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, rp[9], rm[9];
          u[d] = delta; res_e3d(T, ell_oplus(e, u), m, g.yt, rp);
          u[d] = -delta; res_e3d(T, ell_oplus(e, u), m, g.yt, rm);
#pragma unroll
          for (int k = 0; k < 6; ++k) Jp[k * 6 + d] = scalar * (rp[k] - rm[k]);
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
    reduce_group_e3d<0>(Jp, r, w, lane, out, seg_on);
    reduce_group_e3d<1>(Jp, r, w, lane, out, seg_on);
    reduce_group_e3d<2>(Jp, r, w, lane, out, seg_on);
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) chi += __shfl_xor(chi, off, 64);
    if (lane == 0 && seg_on) out[54] = chi;
    return;
  }
  chi = wave_sum(chi);
  if (lane == 0) out[54] = chi;
}

// fixed-order block reduction helpers for the "last workgroup finishes" pattern
struct LmScalars {  // lives in mapped host memory
  double chi2_lin, max_diag, chi2_trial, scale, ok, lambda_used, pad0, pad1;
};

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

// first iteration: chi2 of the linearisation point and max |H_kk| (computeLambdaInit); one thread per ellipsoid,
// the last workgroup reduces the per-workgroup partials in fixed order
static __global__ __launch_bounds__(256) void k_chunk_finalize(DevGraph g, ChunkTable ct, const double* __restrict__ chunk_out,
                                                               const double* __restrict__ objs, int jac, double delta,
                                                               double* __restrict__ blk_part /* gridDim x 2 */,
                                                               unsigned int* __restrict__ ticket, double* __restrict__ dev_scal,
                                                               LmScalars* __restrict__ host) {
  __shared__ double sm[256];
  const int o = blockIdx.x * 256 + threadIdx.x;
  double chi = 0, md = 0;
  if (o < g.n_objs) {
    double diag[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int ch = ct.ostart[o]; ch < ct.ostart[o + 1]; ++ch) {
      const double* p = chunk_out + (size_t)ch * kChunkOut;
      chi += p[54];
      int q = 0;
#pragma unroll
      for (int a = 0; a < 9; ++a) { diag[a] += p[q]; q += 9 - a; }
    }
    if (g.gr_cnt[o] > 0) {  // gravity prior lives on the ellipsoid itself (see k_obj_solve)
      const Ell e = ell_load(objs + 10 * o);
      const double wg = g.grav_w * g.gr_cnt[o];
      double Jg[9], rg;
      if (jac == ESL_JAC_ANALYTIC) rg = jac_grav(e, g.grav_n, Jg);
      else {
        rg = res_grav(e, g.grav_n);
        numeric_jac_obj(e, delta, 1, Jg, [&](const Ell& ep, double* o1) { o1[0] = res_grav(ep, g.grav_n); });
      }
      chi += wg * rg * rg;
#pragma unroll
      for (int a = 0; a < 9; ++a) diag[a] += wg * Jg[a] * Jg[a];
    }
#pragma unroll
    for (int a = 0; a < 9; ++a) md = fmax(md, fabs(diag[a]));
  }
  chi = block256_sum(chi, sm);
  md = block256_max(md, sm);
  if (threadIdx.x == 0) {
    __hip_atomic_store(&blk_part[2 * blockIdx.x], chi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&blk_part[2 * blockIdx.x + 1], md, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (last_block_arrives_wt(ticket)) {
    double c = 0, m = 0;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += 256) { c += blk_part[2 * b]; m = fmax(m, blk_part[2 * b + 1]); }
    c = block256_sum(c, sm);
    m = block256_max(m, sm);
    if (threadIdx.x == 0) {
      dev_scal[0] = c; dev_scal[1] = m;
      host->chi2_lin = c; host->max_diag = m;
      *ticket = 0;
    }
  }
}

// ONE LANE PER ELLIPSOID: H, b from the chunk partials (fixed order => deterministic), gravity prior added,
// unrolled register LDL^T of the 9x9, retraction, trial state.  (A wave per ellipsoid left 63 lanes idle on a
// serial dependency chain; 2k ellipsoids are 32 full waves this way.)
// lambda < 0: lambda = tau * max_diag read from device memory (first LM iteration, computeLambdaInit).
static __global__ __launch_bounds__(64) void k_obj_solve(DevGraph g, ChunkTable ct, const double* __restrict__ chunk_out,
                                                         const double* __restrict__ objs, int jac, double delta, double lambda, double tau,
                                                         const double* __restrict__ dev_scal, double* __restrict__ xo,
                                                         double* __restrict__ objs_trial, double* __restrict__ part) {
  const int o = blockIdx.x * 64 + threadIdx.x;
  if (o >= g.n_objs) return;
  if (lambda < 0) lambda = tau * dev_scal[1];
  const int c0 = ct.ostart[o], c1 = ct.ostart[o + 1];
  const Ell e = ell_load(objs + 10 * o);
  const int ngrav = g.gr_cnt[o];
  if (c0 == c1 && ngrav == 0) {  // inactive vertex: never touched (sparse_optimizer.cpp:236-257)
    ell_store(e, objs_trial + 10 * o);
    part[o * 4 + 0] = 0; part[o * 4 + 2] = 0; part[o * 4 + 3] = 1;
    return;
  }
  double hb[54];
#pragma unroll
  for (int k = 0; k < 54; ++k) hb[k] = 0;
  for (int ch = c0; ch < c1; ++ch) {
    const double* p = chunk_out + (size_t)ch * kChunkOut;
#pragma unroll
    for (int k = 0; k < 54; ++k) hb[k] += p[k];
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
        res_bbox(se3_load(cams + 7 * g.bb_cam[i]), e, g.K, g.bb_meas + 4 * i, r);
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
