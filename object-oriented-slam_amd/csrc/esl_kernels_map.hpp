// esl_kernels_map.hpp — pieces shared by the mapping-mode and SLAM-mode kernels: wave reductions, the J^T W J
// accumulation for one edge, g2o's central-difference Jacobian of an ellipsoid vertex, the NaN pre-check of the bbox
// edges and the fixed-order reduction of per-vertex partials.  (The first-generation mapping kernels -- one wavefront per
// ellipsoid -- lived here; the chunked pipeline of esl_kernels_chunk.hpp replaced them.)
//
// Reference pieces: per-edge linearizeOplus (Thirdparty/g2o/g2o/core/base_binary_edge.hpp:131-205) +
// constructQuadraticForm (:55-120); the NaN filter of Optimizer.cpp:234-243; activeRobustChi2
// (core/sparse_optimizer.cpp:422-435).
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
  if (g.bb_obj[i] < 0) return;   // slack slot of an appendable layout
  const SE3 T = se3_load(cams + 7 * g.bb_cam[i]);
  const Ell e = ell_load(objs + 10 * g.bb_obj[i]);
  double r[4];
  res_box_edge(g.bbox_mode, T, e, g.K, g.bb_meas + 4 * i, r);
  const double w = g.bb_w[i];
  const double c = w * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
  const bool bad = (c != c) || (g.check_vis && !bbox_edge_visible(T, e, g.K, g.img_rows, g.img_cols));
  g.bb_valid[i] = bad ? 0 : 1;
  if (bad) atomicAdd(n_dropped, 1);
}

// chi2 of every edge hanging on ellipsoid `o` for state `e` (wave-cooperative; result valid in lane 0)
__device__ __forceinline__ double obj_chi2(const DevGraph& g, const double* __restrict__ cams, const Ell& e,
                                           int o, int lane) {
  double chi = 0.0;
  for (int i = g.bb_start[o] + lane; i < g.bb_start[o + 1]; i += 64) {
    if (!g.bb_valid[i]) continue;
    const SE3 T = se3_load(cams + 7 * g.bb_cam[i]);
    double r[4];
    res_box_edge(g.bbox_mode, T, e, g.K, g.bb_meas + 4 * i, r);
    chi += g.bb_w[i] * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
  }
  for (int i = g.e3_start[o] + lane; i < g.e3_start[o + 1]; i += 64) {
    if (g.e3_obj[i] < 0) continue;   // slack slot of the appendable layout (its all-zero "measurement" has no rotation: 0 x NaN)
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

// ---- esl_graph_append: new edges into their slots of the ellipsoid-sorted arrays ----------------------------------------
struct AppendBb { int slot, cam, obj, pad; double meas[4], w; };      // 56 B
struct AppendE3 { int slot, cam, obj, pad; double meas[10], w; };     // 104 B
static __global__ void k_append_scatter(DevGraph g, const AppendBb* __restrict__ rb, int nb, const AppendE3* __restrict__ re, int ne) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < nb) {
    const AppendBb r = rb[t];
    g.bb_cam[r.slot] = r.cam; g.bb_obj[r.slot] = r.obj; g.bb_w[r.slot] = r.w; g.bb_valid[r.slot] = 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) g.bb_meas[4 * (size_t)r.slot + k] = r.meas[k];
  } else if (t - nb < ne) {
    const AppendE3 r = re[t - nb];
    g.e3_cam[r.slot] = r.cam; g.e3_obj[r.slot] = r.obj; g.e3_w[r.slot] = r.w;
#pragma unroll
    for (int k = 0; k < 10; ++k) g.e3_meas[10 * (size_t)r.slot + k] = r.meas[k];
  }
}
// small tables (chunk table, id lists, per-ellipsoid counters) from the staged blob to their places: one launch instead of
// a dozen tiny copies
struct AppendCopies { int n; const int* src[12]; int* dst[12]; int count[12]; };
static __global__ void k_append_tables(AppendCopies cp) {
  for (int k = 0; k < cp.n; ++k)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cp.count[k]; i += gridDim.x * blockDim.x) cp.dst[k][i] = cp.src[k][i];
}

}  // namespace esl
