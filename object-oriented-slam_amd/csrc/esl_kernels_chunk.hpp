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
  static __device__ __forceinline__ double run(const double* v, int lane, int& idx, int off) {
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
    if (hi) idx += H;
    return WaveRS<H>::run(nv, lane, idx, off >> 1);
  }
};
template <>
struct WaveRS<1> {
  static __device__ __forceinline__ double run(const double* v, int, int&, int off) {
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

// entries [G*18, G*18+18) reduced over the wave; the owning lanes write the totals
template <int D, int G>
__device__ __forceinline__ void reduce_group(const double* J, const double* r, double w, int lane, double* __restrict__ out) {
  double v[18];
  fill_group<D, G>(v, J, r, w, std::make_integer_sequence<int, 18>{});
  int idx = 0;
  const double tot = WaveRS<18>::run(v, lane, idx, 32);
  constexpr int used = rs_steps(18);               // 5 distributing steps: offsets 32..2
  const int low_mask = (64 >> used) - 1;           // remaining low bits (here: bit 0)
  if ((lane & low_mask) == 0 && idx < 18) out[G * 18 + idx] = tot;
}

template <int JAC>
static __global__ __launch_bounds__(256) void k_chunk_linearize(DevGraph g, ChunkTable ct, const double* __restrict__ cams,
                                                                const double* __restrict__ objs, double delta,
                                                                double* __restrict__ chunk_out) {
  const int lane = threadIdx.x & 63;
  const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ch >= ct.n_chunks) return;
  const int o = ct.obj[ch], type = ct.type[ch];
  const int i = ct.begin[ch] + lane;
  const bool in = i < ct.end[ch];
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
      if (JAC == ESL_JAC_ANALYTIC) jac_bbox(T, e, g.K, meas, r, J, nullptr);
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
  } else if (type == 1) {
    double r[9], J[81], w = 0;
    if (in) {
      const SE3 T = se3_load(cams + 7 * g.e3_cam[i]);
      const Ell m = ell_load(g.e3_meas + 10 * i);
      w = g.e3_w[i];
      if (JAC == ESL_JAC_ANALYTIC) jac_e3d(T, e, m, g.yt, r, J, nullptr);
      else {
        res_e3d(T, e, m, g.yt, r);
        numeric_jac_obj(e, delta, 9, J, [&](const Ell& ep, double* o9) { res_e3d(T, ep, m, g.yt, o9); });
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) chi += r[k] * r[k];
      chi *= w;
    } else {
#pragma unroll
      for (int k = 0; k < 81; ++k) J[k] = 0;
#pragma unroll
      for (int k = 0; k < 9; ++k) r[k] = 0;
    }
    reduce_group<9, 0>(J, r, w, lane, out);
    reduce_group<9, 1>(J, r, w, lane, out);
    reduce_group<9, 2>(J, r, w, lane, out);
  } else {
    double r[1] = {0}, J[9], w = 0;
    if (in) {
      w = g.grav_w;
      if (JAC == ESL_JAC_ANALYTIC) r[0] = jac_grav(e, g.grav_n, J);
      else {
        r[0] = res_grav(e, g.grav_n);
        numeric_jac_obj(e, delta, 1, J, [&](const Ell& ep, double* o1) { o1[0] = res_grav(ep, g.grav_n); });
      }
      chi = w * r[0] * r[0];
    } else {
#pragma unroll
      for (int k = 0; k < 9; ++k) J[k] = 0;
    }
    reduce_group<1, 0>(J, r, w, lane, out);
    reduce_group<1, 1>(J, r, w, lane, out);
    reduce_group<1, 2>(J, r, w, lane, out);
  }
  chi = wave_sum(chi);
  if (lane == 0) out[54] = chi;
}

// fixed-order block reduction helpers for the "last workgroup finishes" pattern
struct LmScalars {  // lives in mapped host memory
  double chi2_lin, max_diag, chi2_trial, scale, ok, lambda_used, pad0, pad1;
};

__device__ __forceinline__ bool last_block_arrives(unsigned int* ticket) {
  __shared__ bool last;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int t = atomicAdd(ticket, 1u);
    last = (t == gridDim.x - 1);
  }
  __syncthreads();
  if (last) __threadfence();
  return last;
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
#pragma unroll
    for (int a = 0; a < 9; ++a) md = fmax(md, fabs(diag[a]));
  }
  chi = block256_sum(chi, sm);
  md = block256_max(md, sm);
  if (threadIdx.x == 0) { blk_part[2 * blockIdx.x] = chi; blk_part[2 * blockIdx.x + 1] = md; }
  if (last_block_arrives(ticket)) {
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

// one wave per ellipsoid: H, b from the chunk partials (fixed order), (H + lambda I) x = b, retraction.
// lambda < 0: lambda = tau * max_diag read from device memory (first LM iteration, computeLambdaInit).
static __global__ __launch_bounds__(256) void k_obj_solve(DevGraph g, ChunkTable ct, const double* __restrict__ chunk_out,
                                                          const double* __restrict__ objs, double lambda, double tau,
                                                          const double* __restrict__ dev_scal, double* __restrict__ xo,
                                                          double* __restrict__ objs_trial, double* __restrict__ part) {
  __shared__ double sH[4][56];
  __shared__ double sL[4][81];
  __shared__ double sx[4][9];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int o = blockIdx.x * 4 + wv;
  if (o >= g.n_objs) return;
  if (lambda < 0) lambda = tau * dev_scal[1];
  const int c0 = ct.ostart[o], c1 = ct.ostart[o + 1];
  const Ell e = ell_load(objs + 10 * o);
  if (c0 == c1) {  // inactive vertex: never touched (sparse_optimizer.cpp:236-257)
    if (lane == 0) { ell_store(e, objs_trial + 10 * o); part[o * 4 + 2] = 0; part[o * 4 + 3] = 1; }
    return;
  }
  if (lane < 54) {
    double s = 0;
    for (int ch = c0; ch < c1; ++ch) s += chunk_out[(size_t)ch * kChunkOut + lane];
    sH[wv][lane] = s;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  // row-parallel LDL^T: lane i (< 9) owns row i of L; D on the diagonal
  double* L = sL[wv];
  if (lane < 9) {
    for (int j = 0; j <= lane; ++j) {
      // packed index of (j, lane), j <= lane
      const int p = j * 9 - j * (j - 1) / 2 + (lane - j);
      L[lane * 9 + j] = sH[wv][p] + ((j == lane) ? lambda : 0.0);
    }
  }
  int ok = 1;
  for (int k = 0; k < 9; ++k) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane == k) {  // D_k
      double v = L[k * 9 + k];
      for (int j = 0; j < k; ++j) v -= L[k * 9 + j] * L[k * 9 + j] * L[j * 9 + j];
      L[k * 9 + k] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane > k && lane < 9) {  // column k of L, one row per lane
      double v = L[lane * 9 + k];
      for (int j = 0; j < k; ++j) v -= L[lane * 9 + j] * L[k * 9 + j] * L[j * 9 + j];
      L[lane * 9 + k] = v / L[k * 9 + k];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if (lane == 0) {
    double y[9], x[9];
    for (int i = 0; i < 9; ++i) { if (!(L[i * 9 + i] > 0)) ok = 0; }
    for (int i = 0; i < 9; ++i) { double s = sH[wv][45 + i]; for (int j = 0; j < i; ++j) s -= L[i * 9 + j] * y[j]; y[i] = s; }
    for (int i = 0; i < 9; ++i) y[i] /= L[i * 9 + i];
    for (int i = 8; i >= 0; --i) { double s = y[i]; for (int j = i + 1; j < 9; ++j) s -= L[j * 9 + i] * x[j]; x[i] = s; }
    double scale = 0;
    for (int i = 0; i < 9; ++i) { scale += x[i] * (lambda * x[i] + sH[wv][45 + i]); sx[wv][i] = x[i]; xo[(size_t)o * 9 + i] = x[i]; }
    const Ell en = ell_oplus(e, x);
    ell_store(en, objs_trial + 10 * o);
    part[o * 4 + 2] = scale;
    part[o * 4 + 3] = (double)ok;
  }
}

// one wave per chunk: chi2 of the trial states; last workgroup: fixed-order totals -> mapped host memory
static __global__ __launch_bounds__(256) void k_chunk_chi2(DevGraph g, ChunkTable ct, const double* __restrict__ cams,
                                                           const double* __restrict__ objs_trial, const double* __restrict__ part,
                                                           double* __restrict__ chunk_chi, unsigned int* __restrict__ ticket,
                                                           double lambda, double tau, const double* __restrict__ dev_scal,
                                                           LmScalars* __restrict__ host) {
  __shared__ double sm[256];
  const int lane = threadIdx.x & 63;
  const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ch < ct.n_chunks) {
    const int o = ct.obj[ch], type = ct.type[ch];
    const int i = ct.begin[ch] + lane;
    const bool in = i < ct.end[ch];
    const Ell e = ell_load(objs_trial + 10 * o);
    double chi = 0;
    if (type == 0) {
      if (in && g.bb_valid[i]) {
        double r[4];
        res_bbox(se3_load(cams + 7 * g.bb_cam[i]), e, g.K, g.bb_meas + 4 * i, r);
        chi = g.bb_w[i] * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
      }
    } else if (type == 1) {
      if (in) {
        double r[9];
        res_e3d(se3_load(cams + 7 * g.e3_cam[i]), e, ell_load(g.e3_meas + 10 * i), g.yt, r);
#pragma unroll
        for (int k = 0; k < 9; ++k) chi += r[k] * r[k];
        chi *= g.e3_w[i];
      }
    } else if (in) {
      const double r = res_grav(e, g.grav_n);
      chi = g.grav_w * r * r;
    }
    chi = wave_sum(chi);
    if (lane == 0) chunk_chi[ch] = chi;
  }
  if (last_block_arrives(ticket)) {
    double c = 0, sc = 0, okv = 1;
    for (int k = threadIdx.x; k < ct.n_chunks; k += 256) c += chunk_chi[k];
    for (int o = threadIdx.x; o < g.n_objs; o += 256) { sc += part[o * 4 + 2]; okv = fmin(okv, part[o * 4 + 3]); }
    c = block256_sum(c, sm);
    sc = block256_sum(sc, sm);
    okv = -block256_max(-okv, sm);
    if (threadIdx.x == 0) {
      host->chi2_trial = c; host->scale = sc; host->ok = okv;
      host->lambda_used = (lambda < 0) ? tau * dev_scal[1] : lambda;
      *ticket = 0;
    }
  }
}

// Hoo / b_o of every ellipsoid from the chunk partials (only used by esl_lm_download and the SLAM-free inspection path)
static __global__ void k_chunk_combine(DevGraph g, ChunkTable ct, const double* __restrict__ chunk_out, double* __restrict__ Hoo,
                                       double* __restrict__ bo) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int o = t / 54, k = t % 54;
  if (o >= g.n_objs) return;
  double s = 0;
  for (int ch = ct.ostart[o]; ch < ct.ostart[o + 1]; ++ch) s += chunk_out[(size_t)ch * kChunkOut + k];
  if (k < 45) Hoo[(size_t)o * 45 + k] = s;
  else bo[(size_t)o * 9 + (k - 45)] = s;
}

}  // namespace esl
