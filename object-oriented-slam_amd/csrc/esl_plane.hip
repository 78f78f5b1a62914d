// esl_plane.hip — esl_extract_ground_plane: the supporting plane of the scene from one depth image.
// Replaces EllipsoidSLAM::PlaneExtractor::extractGroundPlane / extractPlanes (reference src/plane/PlaneExtractor.cpp:22-183),
// the step in front of the per-frame fit (Tracking::ProcessGroundPlaneEstimation, src/core/Tracking.cpp:712-790).
//
// The reference delegates the two heavy steps to PCL (unpinned, not vendored): IntegralImageNormalEstimation
// (AVERAGE_3D_GRADIENT, smoothing size 10, depth-change factor 0.05; PlaneExtractor.cpp:54-61) and
// OrganizedMultiPlaneSegmentation::segmentAndRefine (min 100 inliers, angular threshold Plane.AngleThreshold degrees, distance
// threshold Plane.DistanceThreshold; :68-83).  As for the PCL steps of the fit (esl_fit.hip), they are restated as
// deterministic, order-independent definitions, which the CPU checker and an independent numpy / scipy version used by the
// tests restate as well:
//   points      every pixel: z = depth / scale as float, x = (u - cx) z / fx, y = (v - cy) z / fy (float, :38-42); depth 0 = no point
//   normals     average 3-D gradient over the (2R+1)^2 window, R = smoothing / 2: dh = sum_W (P(v, u+1) - P(v, u-1)),
//               dv = sum_W (P(v+1, u) - P(v-1, u)), n = dv x dh normalised and turned towards the camera; no normal where the
//               (2R+3)^2 neighbourhood holds a missing depth or two adjacent depths that differ by more than factor * z
//   segments    connected components (4-neighbourhood) of pixels with normals whose plane offsets d = -n.p differ by less than
//               the distance threshold and whose normals differ by less than the angular threshold (PCL's
//               PlaneCoefficientComparator); a lock-free union-find over the pixel grid
//   planes      per component of >= max(100, Plane.MinSize) pixels the least-squares plane of its points (centroid + smallest
//               eigenvector of the covariance, moments accumulated exactly in fixed point), d >= 0 (:95-96)
//   refinement  (round 5; the second half of segmentAndRefine, :82) PCL's OrganizedMultiPlaneSegmentation::refine: models =
//               components of >= 100 pixels (setMinInliers); two raster passes in which a pixel carrying a model's label pulls its
//               right / lower (second pass: left / upper) neighbour into the model when that neighbour belongs to no model and
//               lies within 0.02 m of the model's plane (PlaneRefinementComparator) -- labels move as the scan goes, so a model grows
//               through a whole run of such pixels.  Coefficients are NOT re-estimated (PCL does not); what grows are the inlier
//               lists: the sizes the reference filters by (:87) and ranks the ground candidates by (:160), and GetPoints().
//               Sequential by definition; here ONE WAVE walks the rows in order: a row's runs are disjoint, so every run is
//               walked by its own lane, and the pull into the next row is one independent test per column.  The result does not
//               depend on thread timing (k_plane_refine).  esl_plane_params::refine = 0 gives the segments alone.
//   models      (round 6) what PCL's segment() keeps: components of MORE than 100 pixels whose surface curvature
//               |lambda_min| / trace(cov) is below maximum_curvature_ = 0.001; the rest are "other" pixels a model may absorb
//   ground      among the planes whose normal is within 45 degrees of the camera's y axis (either sign, :139-146) the one with
//               the most pixels (:160-162; ties: the component with the smaller root pixel); sign such that the camera
//               centre is on the positive side (:165-167)
// Runs once per sequence (until it succeeds): nothing here is tuned for throughput, everything for determinism.
#include <cmath>
#include <string>
#include <vector>

#include "esl_ctx.hpp"

namespace esl {

struct PlaneArgs {
  const uint16_t* depth; int w, h;
  double fx, fy, cx, cy, scale;
  int R; double depth_factor, cos_ang, dist_th; int min_size;
  int min_inliers; double max_curv;   // PCL's model test (segment()): size > min_inliers, curvature < max_curv (<= 0: no test); min_size: an extra >= when no refinement follows
  float* nrm;            // 4 per pixel: nx ny nz d ; nx = NaN: no normal
  int* parent;           // union-find forest over the pixels
  int* cnt;              // pixels per root
  long long* mom;        // per root: 9 fixed-point sums  x y z xx xy xz yy yz zz
  double* out;           // [0..3] plane, [4] ok, [5] planes >= min size, [6] pixels of the chosen plane, [7] components with normals
  int* plane_of_root;    // esl_extract_planes: index of the plane a root belongs to, -1 = none
  double* list;          // esl_extract_planes: 5 per plane (a b c d pixels), in raster order of each segment's first pixel
  int list_cap;
  int* labels;           // esl_extract_planes: plane index per pixel, -1 = none
  int* blk;              // k_plane_flag / _scan / _emit: models per block of 256 roots, then their exclusive scan
};
constexpr double kFix1 = 4294967296.0;     // first moments in 2^-32 m   (65 m x 2^19 pixels x 2^32 < 2^63)
constexpr double kFix2 = 1073741824.0;     // second moments in 2^-30 m^2 (4300 m^2 x 2^19 pixels x 2^30 < 2^63)

__device__ __forceinline__ bool px_point(const PlaneArgs& a, int u, int v, float p[3]) {
  const uint16_t d = a.depth[(size_t)v * a.w + u];
  const float z = (float)((double)d / a.scale);
  p[2] = z;
  p[0] = (float)(((double)u - a.cx) * (double)z / a.fx);
  p[1] = (float)(((double)v - a.cy) * (double)z / a.fy);
  return d != 0;
}

// Round 6: the tile's points staged once.  The direct form below recomputes the point of every pixel of the (2R+1)^2 window four times
// per centre pixel -- three IEEE double divisions each, ~1,500 divisions per output pixel: 0.35-0.41 ms per 640 x 480 image, all of it
// the divider.  Here a workgroup computes the points of its 16 x 16 tile + halo (R + 1) ONCE into LDS (same expressions: same bits) and
// the validity test and the window sums read them there, in the same order.  Dynamic LDS: (16 + 2 (R + 1))^2 x 20 bytes (R = 5: 15.7 KB).
static __global__ __launch_bounds__(256) void k_plane_normals_lds(PlaneArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int R = a.R, H = R + 1, side = 16 + 2 * H, np = side * side;
  double* zz = reinterpret_cast<double*>(lds_raw);   // depth in metres, as the validity test forms it; 0: no depth or outside the image
  float* pxs = reinterpret_cast<float*>(zz + np);
  float* pys = pxs + np;
  float* pzs = pys + np;
  const int u0 = blockIdx.x * 16 - H, v0 = blockIdx.y * 16 - H;
  for (int i = threadIdx.x; i < np; i += 256) {
    const int lx = i % side, ly = i / side, x = u0 + lx, y = v0 + ly;
    double z = 0;
    float p[3] = {0, 0, 0};
    if (x >= 0 && y >= 0 && x < a.w && y < a.h) {
      const uint16_t d = a.depth[(size_t)y * a.w + x];
      if (d != 0) { z = (double)d / a.scale; (void)px_point(a, x, y, p); }
    }
    zz[i] = z; pxs[i] = p[0]; pys[i] = p[1]; pzs[i] = p[2];
  }
  __syncthreads();
  const int tu = threadIdx.x & 15, tv = threadIdx.x >> 4;
  const int u = blockIdx.x * 16 + tu, v = blockIdx.y * 16 + tv;
  if (u >= a.w || v >= a.h) return;
  float* o = a.nrm + 4 * ((size_t)v * a.w + u);
  const float qnan = __builtin_nanf("");
  o[0] = qnan; o[1] = o[2] = o[3] = 0;
  if (u - R - 1 < 0 || v - R - 1 < 0 || u + R + 1 >= a.w || v + R + 1 >= a.h) return;
  const int c0 = (tv + H) * side + (tu + H);        // this pixel in the tile
  // validity of the (2R+3)^2 neighbourhood: every depth present, no jump between horizontal / vertical neighbours
  for (int dy = -H; dy <= H; ++dy)
    for (int dx = -H; dx <= H; ++dx) {
      const int i = c0 + dy * side + dx;
      const double z = zz[i];
      if (z == 0) return;                           // (a depth of 0 is "no point"; any other depth is a positive number)
      if (dx + 1 <= H) { const double z2 = zz[i + 1]; if (fabs(z2 - z) > a.depth_factor * z) return; }
      if (dy + 1 <= H) { const double z2 = zz[i + side]; if (fabs(z2 - z) > a.depth_factor * z) return; }
    }
  double dh[3] = {0, 0, 0}, dv[3] = {0, 0, 0};
  for (int dy = -R; dy <= R; ++dy)
    for (int dx = -R; dx <= R; ++dx) {
      const int i = c0 + dy * side + dx;
      dh[0] += (double)pxs[i + 1] - (double)pxs[i - 1]; dv[0] += (double)pxs[i + side] - (double)pxs[i - side];
      dh[1] += (double)pys[i + 1] - (double)pys[i - 1]; dv[1] += (double)pys[i + side] - (double)pys[i - side];
      dh[2] += (double)pzs[i + 1] - (double)pzs[i - 1]; dv[2] += (double)pzs[i + side] - (double)pzs[i - side];
    }
  double n[3] = {dv[1] * dh[2] - dv[2] * dh[1], dv[2] * dh[0] - dv[0] * dh[2], dv[0] * dh[1] - dv[1] * dh[0]};
  const double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
  if (!(nn > 0)) return;
  const float p[3] = {pxs[c0], pys[c0], pzs[c0]};
  double s = 1.0 / nn;
  if ((n[0] * p[0] + n[1] * p[1] + n[2] * p[2]) * s > 0) s = -s;   // towards the camera at the origin
  const float nx = (float)(n[0] * s), ny = (float)(n[1] * s), nz = (float)(n[2] * s);
  o[0] = nx; o[1] = ny; o[2] = nz;
  o[3] = -(nx * p[0] + ny * p[1] + nz * p[2]);
}
// the direct form (tiles whose halo does not fit LDS: very wide smoothing windows)
static __global__ __launch_bounds__(256) void k_plane_normals(PlaneArgs a) {
  const int u = blockIdx.x * 16 + (threadIdx.x & 15), v = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (u >= a.w || v >= a.h) return;
  float* o = a.nrm + 4 * ((size_t)v * a.w + u);
  const float qnan = __builtin_nanf("");
  o[0] = qnan; o[1] = o[2] = o[3] = 0;
  const int R = a.R;
  if (u - R - 1 < 0 || v - R - 1 < 0 || u + R + 1 >= a.w || v + R + 1 >= a.h) return;
  // validity of the (2R+3)^2 neighbourhood: every depth present, no jump between horizontal / vertical neighbours
  for (int y = v - R - 1; y <= v + R + 1; ++y)
    for (int x = u - R - 1; x <= u + R + 1; ++x) {
      const uint16_t d = a.depth[(size_t)y * a.w + x];
      if (d == 0) return;
      const double z = (double)d / a.scale;
      if (x + 1 <= u + R + 1) { const double z2 = (double)a.depth[(size_t)y * a.w + x + 1] / a.scale; if (fabs(z2 - z) > a.depth_factor * z) return; }
      if (y + 1 <= v + R + 1) { const double z2 = (double)a.depth[(size_t)(y + 1) * a.w + x] / a.scale; if (fabs(z2 - z) > a.depth_factor * z) return; }
    }
  double dh[3] = {0, 0, 0}, dv[3] = {0, 0, 0};
  for (int y = v - R; y <= v + R; ++y)
    for (int x = u - R; x <= u + R; ++x) {
      float pr[3], pl[3], pd[3], pu[3];
      px_point(a, x + 1, y, pr); px_point(a, x - 1, y, pl); px_point(a, x, y + 1, pd); px_point(a, x, y - 1, pu);
      for (int k = 0; k < 3; ++k) { dh[k] += (double)pr[k] - (double)pl[k]; dv[k] += (double)pd[k] - (double)pu[k]; }
    }
  double n[3] = {dv[1] * dh[2] - dv[2] * dh[1], dv[2] * dh[0] - dv[0] * dh[2], dv[0] * dh[1] - dv[1] * dh[0]};
  const double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
  if (!(nn > 0)) return;
  float p[3];
  px_point(a, u, v, p);
  double s = 1.0 / nn;
  if ((n[0] * p[0] + n[1] * p[1] + n[2] * p[2]) * s > 0) s = -s;   // towards the camera at the origin
  const float nx = (float)(n[0] * s), ny = (float)(n[1] * s), nz = (float)(n[2] * s);
  o[0] = nx; o[1] = ny; o[2] = nz;
  o[3] = -(nx * p[0] + ny * p[1] + nz * p[2]);
}

__device__ __forceinline__ int uf_find(int* parent, int i) {
  while (true) {
    const int p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p == i) return i;
    const int gp = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gp != p) (void)atomicCAS(&parent[i], p, gp);   // path halving
    i = p;
  }
}
__device__ __forceinline__ void uf_union(int* parent, int a, int b) {
  while (true) {
    a = uf_find(parent, a); b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }      // the smaller index becomes the root
    if (atomicCAS(&parent[a], a, b) == a) return;
  }
}
__device__ __forceinline__ bool same_plane(const PlaneArgs& a, const float* ni, const float* nj) {
  if (ni[0] != ni[0] || nj[0] != nj[0]) return false;
  const float dot = ni[0] * nj[0] + ni[1] * nj[1] + ni[2] * nj[2];
  return fabsf(ni[3] - nj[3]) < (float)a.dist_th && dot > (float)a.cos_ang;
}
static __global__ __launch_bounds__(256) void k_plane_init(PlaneArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.w * a.h) return;
  a.parent[i] = i; a.cnt[i] = 0;
  for (int k = 0; k < 9; ++k) a.mom[(size_t)i * 9 + k] = 0;
}
// Round 6: two levels.  The one-level kernel below sends every pixel's two unions through global atomics (0.52 ms per 640 x 480 image:
// pointer chasing over a forest that 300k threads rewrite at once).  Here a workgroup first unites the pixels of its 32 x 32 tile in LDS
// (the same lock-free union-find on local indices -- row-major inside the tile, so "smaller local index" is "smaller pixel index" and the
// root of a local component is its first pixel, as in the global forest), writes every pixel's parent as its local root's pixel index (a
// forest of stars), and k_plane_union_borders then unites across the tile borders only (one pixel in 16).  The final roots are the
// components' first pixels either way: same segments, same order.
constexpr int kPlaneTile = 32;
__device__ __forceinline__ int uf_find_lds(int* parent, int i) {
  while (true) {
    const int p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (p == i) return i;
    const int gp = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (gp != p) (void)atomicCAS(&parent[i], p, gp);
    i = p;
  }
}
__device__ __forceinline__ void uf_union_lds(int* parent, int x, int y) {
  while (true) {
    x = uf_find_lds(parent, x); y = uf_find_lds(parent, y);
    if (x == y) return;
    if (x < y) { const int t = x; x = y; y = t; }
    if (atomicCAS(&parent[x], x, y) == x) return;
  }
}
static __global__ __launch_bounds__(256) void k_plane_union_tiles(PlaneArgs a) {
  __shared__ int lp[kPlaneTile * kPlaneTile];
  const int u0 = blockIdx.x * kPlaneTile, v0 = blockIdx.y * kPlaneTile;
  for (int l = threadIdx.x; l < kPlaneTile * kPlaneTile; l += 256) lp[l] = l;
  __syncthreads();
  for (int l = threadIdx.x; l < kPlaneTile * kPlaneTile; l += 256) {
    const int lx = l % kPlaneTile, ly = l / kPlaneTile, u = u0 + lx, v = v0 + ly;
    if (u >= a.w || v >= a.h) continue;
    const size_t i = (size_t)v * a.w + u;
    const float* ni = a.nrm + 4 * i;
    if (ni[0] != ni[0]) continue;
    if (lx + 1 < kPlaneTile && u + 1 < a.w && same_plane(a, ni, a.nrm + 4 * (i + 1))) uf_union_lds(lp, l, l + 1);
    if (ly + 1 < kPlaneTile && v + 1 < a.h && same_plane(a, ni, a.nrm + 4 * (i + a.w))) uf_union_lds(lp, l, l + kPlaneTile);
  }
  __syncthreads();
  for (int l = threadIdx.x; l < kPlaneTile * kPlaneTile; l += 256) {
    const int lx = l % kPlaneTile, ly = l / kPlaneTile, u = u0 + lx, v = v0 + ly;
    if (u >= a.w || v >= a.h) continue;
    const int rt = uf_find_lds(lp, l);
    a.parent[(size_t)v * a.w + u] = (v0 + rt / kPlaneTile) * a.w + (u0 + rt % kPlaneTile);
  }
}
// unions across the tile borders: thread = (tile row boundary pixel | tile column boundary pixel)
static __global__ __launch_bounds__(256) void k_plane_union_borders(PlaneArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.w * a.h) return;
  const int u = i % a.w, v = i / a.w;
  const bool right_edge = (u % kPlaneTile) == kPlaneTile - 1, bottom_edge = (v % kPlaneTile) == kPlaneTile - 1;
  if (!right_edge && !bottom_edge) return;
  const float* ni = a.nrm + 4 * (size_t)i;
  if (ni[0] != ni[0]) return;
  if (right_edge && u + 1 < a.w && same_plane(a, ni, a.nrm + 4 * (size_t)(i + 1))) uf_union(a.parent, i, i + 1);
  if (bottom_edge && v + 1 < a.h && same_plane(a, ni, a.nrm + 4 * (size_t)(i + a.w))) uf_union(a.parent, i, i + a.w);
}
// the one-level form (kept: ESL_PLANE_UNION=0, A/B)
static __global__ __launch_bounds__(256) void k_plane_union(PlaneArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.w * a.h) return;
  const int u = i % a.w, v = i / a.w;
  const float* ni = a.nrm + 4 * (size_t)i;
  if (ni[0] != ni[0]) return;
  if (u + 1 < a.w && same_plane(a, ni, a.nrm + 4 * (size_t)(i + 1))) uf_union(a.parent, i, i + 1);
  if (v + 1 < a.h && same_plane(a, ni, a.nrm + 4 * (size_t)(i + a.w))) uf_union(a.parent, i, i + a.w);
}
// Per-root pixel count and moment sums.  The sums are integers (fixed point), so any order gives the same bits; a wave first adds
// up the lanes that share a root (a floor segment of 200k pixels used to send 2 M atomics to the same ten addresses, which the
// memory system serialises at ~10 ns each: 18 ms of the 19.5 ms the whole extraction took) and one lane per (wave, root) issues
// the ten atomics.
static __global__ __launch_bounds__(256) void k_plane_moments(PlaneArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  int r = -1;
  long long v9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (i < a.w * a.h) {
    const float* ni = a.nrm + 4 * (size_t)i;
    if (ni[0] == ni[0]) {
      r = uf_find(a.parent, i);
      float p[3];
      px_point(a, i % a.w, i / a.w, p);
      const double x = p[0], y = p[1], z = p[2];
      const double d9[9] = {x * kFix1, y * kFix1, z * kFix1, x * x * kFix2, x * y * kFix2, x * z * kFix2, y * y * kFix2, y * z * kFix2, z * z * kFix2};
#pragma unroll
      for (int k = 0; k < 9; ++k) v9[k] = llrint(d9[k]);
    }
  }
  const int lane = threadIdx.x & 63;
  bool pending = r >= 0;
  while (__any(pending)) {
    const unsigned long long mask = __ballot(pending);
    const int leader = __ffsll((long long)mask) - 1;
    const int rl = __shfl(r, leader, 64);
    const bool mine = pending && r == rl;
    long long s[9];
    int cnt = mine ? 1 : 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) s[k] = mine ? v9[k] : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      cnt += __shfl_xor(cnt, off, 64);
#pragma unroll
      for (int k = 0; k < 9; ++k) s[k] += __shfl_xor(s[k], off, 64);
    }
    if (lane == leader) {
      atomicAdd(&a.cnt[rl], cnt);
      unsigned long long* m = (unsigned long long*)(a.mom + (size_t)rl * 9);
#pragma unroll
      for (int k = 0; k < 9; ++k) atomicAdd(&m[k], (unsigned long long)s[k]);
    }
    if (mine) pending = false;
  }
}

// symmetric 3x3 eigen-decomposition by cyclic Jacobi; returns the eigenvector of the smallest eigenvalue
__device__ void smallest_eigvec3(const double C[9], double n[3]) {
  double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int i = 0; i < 9; ++i) A[i] = C[i];
  for (int sweep = 0; sweep < 60; ++sweep) {
    const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    if (off < 1e-300) break;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 3; ++q) {
        const double apq = A[p * 3 + q];
        if (apq == 0.0) continue;
        const double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
        const double c = 1 / sqrt(t * t + 1), s = t * c;
        for (int k = 0; k < 3; ++k) { const double x = A[k * 3 + p], y = A[k * 3 + q]; A[k * 3 + p] = c * x - s * y; A[k * 3 + q] = s * x + c * y; }
        for (int k = 0; k < 3; ++k) { const double x = A[p * 3 + k], y = A[q * 3 + k]; A[p * 3 + k] = c * x - s * y; A[q * 3 + k] = s * x + c * y; }
        for (int k = 0; k < 3; ++k) { const double x = V[k * 3 + p], y = V[k * 3 + q]; V[k * 3 + p] = c * x - s * y; V[k * 3 + q] = s * x + c * y; }
      }
  }
  int m = 0;
  for (int j = 1; j < 3; ++j) if (A[j * 4] < A[m * 4]) m = j;
  for (int k = 0; k < 3; ++k) n[k] = V[k * 3 + m];
}

// returns PCL's surface curvature of the segment: |lambda_min| / trace(cov) (lambda_min as the Rayleigh quotient of the unit normal)
__device__ double plane_from_moments(const long long* m, int cnt, double pl[4]) {
  const double inv = 1.0 / (double)cnt;
  const double cx = (double)m[0] / kFix1 * inv, cy = (double)m[1] / kFix1 * inv, cz = (double)m[2] / kFix1 * inv;
  const double C[9] = {(double)m[3] / kFix2 * inv - cx * cx, (double)m[4] / kFix2 * inv - cx * cy, (double)m[5] / kFix2 * inv - cx * cz,
                       (double)m[4] / kFix2 * inv - cx * cy, (double)m[6] / kFix2 * inv - cy * cy, (double)m[7] / kFix2 * inv - cy * cz,
                       (double)m[5] / kFix2 * inv - cx * cz, (double)m[7] / kFix2 * inv - cy * cz, (double)m[8] / kFix2 * inv - cz * cz};
  double n[3];
  smallest_eigvec3(C, n);
  const double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
  pl[0] = n[0] / nn; pl[1] = n[1] / nn; pl[2] = n[2] / nn;
  pl[3] = -(pl[0] * cx + pl[1] * cy + pl[2] * cz);
  if (pl[3] < 0) for (int k = 0; k < 4; ++k) pl[k] = -pl[k];   // camera centre on the positive side (PlaneExtractor.cpp:95-96, 165-167)
  double lmin = 0;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) lmin += pl[i] * C[3 * i + j] * pl[j];
  const double tr = C[0] + C[4] + C[8];
  return tr != 0 ? fabs(lmin / tr) : 0.0;
}
// PCL's model test (OrganizedMultiPlaneSegmentation::segment): more than min_inliers pixels (strict) and curvature below
// maximum_curvature_; a.min_size: an additional >= (the Plane.MinSize filter folded in where no refinement pass can grow a segment)
__device__ __forceinline__ bool plane_is_model(const PlaneArgs& a, int r, int c, double pl[4]) {
  if (!(c > a.min_inliers && c >= a.min_size && c > 0)) return false;
  const double curv = plane_from_moments(a.mom + (size_t)r * 9, c, pl);
  return !(a.max_curv > 0) || curv < a.max_curv;
}

// the MODELS (plane_is_model) in raster order of their first pixel (= their root: the union-find keeps the smallest index on top), as
// extractPlanes stores them (PlaneExtractor.cpp:85-103).  Round 6: three small launches over all roots instead of ONE workgroup
// scanning 307,200 roots 256 at a time (1.5 ms): flags + per-block counts, a scan of the 1,200 counts, the emit.
static __global__ __launch_bounds__(256) void k_plane_flag(PlaneArgs a) {
  __shared__ int s_n;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const int N = a.w * a.h, r = blockIdx.x * 256 + threadIdx.x;
  const int c = r < N ? a.cnt[r] : 0;
  double pl[4];
  const int flag = (r < N && plane_is_model(a, r, c, pl)) ? 1 : 0;
  if (r < N) a.plane_of_root[r] = flag;
  if (flag) atomicAdd(&s_n, 1);
  __syncthreads();
  if (threadIdx.x == 0) a.blk[blockIdx.x] = s_n;
}
static __global__ __launch_bounds__(1024) void k_plane_scan(PlaneArgs a, int nb) {
  __shared__ int s_part[1024];
  const int t = threadIdx.x, per = (nb + 1023) / 1024, b0 = t * per;
  int sum = 0;
  for (int b = b0; b < b0 + per && b < nb; ++b) sum += a.blk[b];
  s_part[t] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = t >= off ? s_part[t - off] : 0;
    __syncthreads();
    s_part[t] += v;
    __syncthreads();
  }
  int run = s_part[t] - sum;
  for (int b = b0; b < b0 + per && b < nb; ++b) { const int v = a.blk[b]; a.blk[b] = run; run += v; }
  if (t == 1023) a.out[5] = s_part[1023];
}
static __global__ __launch_bounds__(256) void k_plane_emit(PlaneArgs a) {
  __shared__ int s_w[4];
  const int N = a.w * a.h, r = blockIdx.x * 256 + threadIdx.x;
  const int flag = r < N ? a.plane_of_root[r] : 0;
  const unsigned long long bal = __ballot(flag != 0);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) s_w[wv] = __popcll(bal);
  __syncthreads();
  int idx = a.blk[blockIdx.x] + __popcll(bal & ((1ull << lane) - 1ull));
  for (int k = 0; k < wv; ++k) idx += s_w[k];
  if (r < N) a.plane_of_root[r] = flag ? idx : -1;
  if (flag && idx < a.list_cap) {
    double pl[4];
    const int c = a.cnt[r];
    (void)plane_from_moments(a.mom + (size_t)r * 9, c, pl);
    for (int k = 0; k < 4; ++k) a.list[(size_t)idx * 5 + k] = pl[k];
    a.list[(size_t)idx * 5 + 4] = c;
  }
}
static void plane_list_launch(const PlaneArgs& a, hipStream_t st) {
  const int nb = (a.w * a.h + 255) / 256;
  hipLaunchKernelGGL(k_plane_flag, dim3((unsigned)nb), dim3(256), 0, st, a);
  hipLaunchKernelGGL(k_plane_scan, dim3(1), dim3(1024), 0, st, a, nb);
  hipLaunchKernelGGL(k_plane_emit, dim3((unsigned)nb), dim3(256), 0, st, a);
}
static __global__ __launch_bounds__(256) void k_plane_labels(PlaneArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.w * a.h) return;
  const float* ni = a.nrm + 4 * (size_t)i;
  a.labels[i] = ni[0] != ni[0] ? -1 : a.plane_of_root[uf_find(a.parent, i)];
}

// refinement pass, step 0: the label image PCL's refine() works on -- model index, -1 no label (no depth), -2 any other pixel
static __global__ __launch_bounds__(256) void k_plane_lab_init(PlaneArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.w * a.h) return;
  int l = -2;
  if (a.depth[i] == 0) l = -1;
  else if (a.nrm[4 * (size_t)i] == a.nrm[4 * (size_t)i]) { const int m = a.plane_of_root[uf_find(a.parent, i)]; l = m >= 0 ? m : -2; }
  a.labels[i] = l;
}
// OrganizedMultiPlaneSegmentation::refine (see the file header; the CPU checker under oracle/ runs the literal loops).
// ONE workgroup; per row: (1) the row's labels -> LDS, (2) run starts from the labels as they are BEFORE this row's sweep (a pixel of
// a model followed by an unlabelled-by-any-model pixel), (3) every run start walks its run -- runs end at the next pixel that is not
// "other", so they are disjoint --, (4) the labels go back and every column tests the ONE pixel of the next row (quirk kept: not when
// the in-row neighbour has no label).  Second pass mirrored, plus PCL's "left of column 0" = the last pixel of the row above.
// Round 6.  What a row cost in round 5's form (2.3 us, 2.2 ms per image) was the global-memory round trip of its labels and depths in
// front of every step, not the step: (a) a single wave without any barrier takes 7 us per row (ten columns per lane, every one a
// dependent LDS round trip), 256 threads 2.6 us, 1,024 threads 2.1 us; (b) with the next ROW prefetched one step ahead the step still
// waits, because a step is shorter than a memory round trip.  So: a ring of 2 x kPlaneRefB rows (labels AND depths) in LDS, the
// kPlaneRefB rows after them in flight in registers for a whole block of steps, a finished row leaves once, growth counts in LDS.
// Rows beyond the two a step works on are never modified before their turn, so the prefetched values are the ones the serial sweep
// would read.
// (c) what remained after that (1.7 ms) was the point of a tested pixel: three IEEE double divisions per test (px_point's, kept bit
// for bit), ~500 cycles on the one lane the whole workgroup waits for, once or twice per row and once per pixel of a run.  The
// points of a row are now computed by ALL threads when the row is parked in the ring (same expressions, same bits); a test is four
// multiply-adds.
constexpr int kPlaneRefThreads = 1024;
constexpr int kPlaneRefMaxW = 4096;                        // (B = 1; the B = 4 ring fits images up to kPlaneRefRingW columns)
constexpr int kPlaneRefRingW = 1152;
constexpr int kPlaneRefGrown = 2048;                       // models whose growth is counted in LDS (beyond: global atomics)
static size_t plane_refine_lds(int B, int w) { return ((size_t)(2 * B) * w * 4 + (size_t)w + kPlaneRefGrown) * sizeof(int); }
// B: rows per prefetch block; the ring holds 2 B rows
template <int B>
static __global__ __launch_bounds__(kPlaneRefThreads) void k_plane_refine(PlaneArgs a, int* __restrict__ grown, double thr, int n_cap) {
  extern __shared__ int s_row[];
  const int w = a.w, h = a.h, t = threadIdx.x;
  constexpr int RING = 2 * B;
  int* sLab = s_row;                                       // [RING][w]
  float* sPt = reinterpret_cast<float*>(s_row + RING * w); // [RING][3][w]: the pixel's point, as px_point computes it
  int* sF = s_row + 4 * RING * w;
  int* sG = sF + w;                                        // [kPlaneRefGrown]
  constexpr int kPer = (B == 1 ? kPlaneRefMaxW : kPlaneRefRingW + kPlaneRefThreads - 1) / kPlaneRefThreads;
  auto near = [&](int m, const float* P, int u) {           // P: the row's points
    const double* pl = a.list + (size_t)m * 5;
    return fabs(pl[0] * (double)P[u] + pl[1] * (double)P[w + u] + pl[2] * (double)P[2 * w + u] + pl[3]) < thr;
  };
  auto grow = [&](int m, int n) { if (m < kPlaneRefGrown) atomicAdd(&sG[m], n); else atomicAdd(&grown[m], n); };
  for (int m = t; m < kPlaneRefGrown; m += kPlaneRefThreads) sG[m] = 0;
  // LOGICAL rows: i = 0 .. h - 1 in sweep order (pass 1: image row i; pass 2: image row h - 1 - i); step i works on rows i and i + 1
  int pl_[B][kPer]; unsigned short pd_[B][kPer];
  auto fetch = [&](int i0, bool down) {                    // logical rows i0 .. i0 + B - 1 -> registers (rows outside the image: nothing)
#pragma unroll
    for (int r = 0; r < B; ++r)
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        const int i = i0 + r, v = down ? i : h - 1 - i, u = t + q * kPlaneRefThreads;
        const bool in = i >= 0 && i < h && u < w;
        pl_[r][q] = in ? a.labels[(size_t)v * w + u] : -1;
        pd_[r][q] = in ? a.depth[(size_t)v * w + u] : (unsigned short)0;
      }
  };
  auto park = [&](int i0, bool down) {
#pragma unroll
    for (int r = 0; r < B; ++r)
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        const int i = i0 + r, v = down ? i : h - 1 - i, u = t + q * kPlaneRefThreads, sl = (i % RING) * w;
        if (u < w) {
          sLab[sl + u] = pl_[r][q];
          const float z = (float)((double)pd_[r][q] / a.scale);   // px_point, expression for expression
          sPt[3 * sl + u] = (float)(((double)u - a.cx) * (double)z / a.fx);
          sPt[3 * sl + w + u] = (float)(((double)v - a.cy) * (double)z / a.fy);
          sPt[3 * sl + 2 * w + u] = z;
        }
      }
  };
  // ---- first pass: rightwards, downwards
  fetch(0, true); park(0, true); fetch(B, true); park(B, true);
  __syncthreads();
  for (int v = 0; v < h - 1; ++v) {
    if (v % B == 0) fetch(v + 2 * B, true);                 // the block after the ring's two: in flight for B steps
    int* sL = sLab + (v % RING) * w; int* sN = sLab + ((v + 1) % RING) * w;
    const float* sP = sPt + 3 * (v % RING) * w; const float* sPN = sPt + 3 * ((v + 1) % RING) * w;
    for (int u = t; u < w - 1; u += kPlaneRefThreads) sF[u] = (sL[u] >= 0 && sL[u + 1] == -2) ? 1 : 0;
    __syncthreads();
    for (int u = t; u < w - 1; u += kPlaneRefThreads)
      if (sF[u]) {
        const int m = sL[u];
        int j = u + 1, n = 0;
        while (j < w && sL[j] == -2 && near(m, sP, j)) { sL[j] = m; ++j; ++n; }
        if (n) grow(m, n);
      }
    __syncthreads();
    for (int u = t; u < w; u += kPlaneRefThreads) {
      const int cur = sL[u];
      a.labels[(size_t)v * w + u] = cur;
      if (u < w - 1 && cur >= 0 && sL[u + 1] != -1 && sN[u] == -2 && near(cur, sPN, u)) { sN[u] = cur; grow(cur, 1); }
    }
    __syncthreads();
    if (v % B == B - 1) { park(v + 1 + B, true); __syncthreads(); }   // rows v + 1 - B .. v are finished: their slots take the block that was in flight
  }
  for (int u = t; u < w; u += kPlaneRefThreads) a.labels[(size_t)(h - 1) * w + u] = sLab[((h - 1) % RING) * w + u];
  __threadfence_block();
  __syncthreads();
  // ---- second pass: leftwards, upwards (plus PCL's "left of column 0" = the last pixel of the row above); logical row i = image row h - 1 - i
  fetch(0, false); park(0, false); fetch(B, false); park(B, false);
  __syncthreads();
  for (int i = 0; i < h - 1; ++i) {
    const int v = h - 1 - i;
    if (i % B == 0) fetch(i + 2 * B, false);
    int* sL = sLab + (i % RING) * w; int* sN = sLab + ((i + 1) % RING) * w;
    const float* sP = sPt + 3 * (i % RING) * w; const float* sPN = sPt + 3 * ((i + 1) % RING) * w;
    for (int u = t; u < w; u += kPlaneRefThreads) sF[u] = (u >= 1 && sL[u] >= 0 && sL[u - 1] == -2) ? 1 : 0;
    __syncthreads();
    for (int u = t; u < w; u += kPlaneRefThreads)
      if (sF[u]) {
        const int m = sL[u];
        int j = u - 1, n = 0;
        while (j >= 0 && sL[j] == -2 && near(m, sP, j)) { sL[j] = m; --j; ++n; }
        if (n) grow(m, n);
      }
    __syncthreads();
    for (int u = t; u < w; u += kPlaneRefThreads) {
      const int cur = sL[u];
      a.labels[(size_t)v * w + u] = cur;
      if (u >= 1 && cur >= 0 && sL[u - 1] != -1 && sN[u] == -2 && near(cur, sPN, u)) { sN[u] = cur; grow(cur, 1); }
    }
    __syncthreads();
    if (t == 0) {                                          // column 0 comes last in PCL's sweep: "left" = the last pixel of the row above
      const int cur = sL[0];
      if (cur != -1 && sN[w - 1] != -1) {
        if (cur >= 0 && sN[w - 1] == -2 && near(cur, sPN, w - 1)) { sN[w - 1] = cur; grow(cur, 1); }
        if (cur >= 0 && sN[0] == -2 && near(cur, sPN, 0)) { sN[0] = cur; grow(cur, 1); }
      }
    }
    __syncthreads();
    if (i % B == B - 1) { park(i + 1 + B, false); __syncthreads(); }
  }
  for (int u = t; u < w; u += kPlaneRefThreads) a.labels[u] = sLab[((h - 1) % RING) * w + u];
  for (int m = t; m < kPlaneRefGrown && m < n_cap; m += kPlaneRefThreads) if (sG[m]) atomicAdd(&grown[m], sG[m]);
}

// one workgroup: every component of >= min_size pixels -> its plane; the ground-plane candidate with the most pixels wins
static __global__ __launch_bounds__(256) void k_plane_select(PlaneArgs a) {
  __shared__ int s_cnt[256], s_root[256], s_np[256], s_nc[256];
  int best_cnt = 0, best_root = -1, n_planes = 0, n_comp = 0;
  const int N = a.w * a.h;
  for (int r = threadIdx.x; r < N; r += 256) {
    const int c = a.cnt[r];
    if (c <= 0) continue;
    ++n_comp;
    double n[4];
    if (!plane_is_model(a, r, c, n)) continue;
    ++n_planes;
    // wall filter: the angle between the normal and the camera's y axis must not lie in (pi/4, 3 pi/4)
    const double th = acos(n[1] / sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]));
    if (th > M_PI / 4 && th < 3 * M_PI / 4) continue;
    if (c > best_cnt || (c == best_cnt && (best_root < 0 || r < best_root))) { best_cnt = c; best_root = r; }
  }
  s_cnt[threadIdx.x] = best_cnt; s_root[threadIdx.x] = best_root; s_np[threadIdx.x] = n_planes; s_nc[threadIdx.x] = n_comp;
  __syncthreads();
  if (threadIdx.x != 0) return;
  for (int k = 1; k < 256; ++k) {
    n_planes += s_np[k]; n_comp += s_nc[k];
    if (s_root[k] >= 0 && (s_cnt[k] > best_cnt || (s_cnt[k] == best_cnt && (best_root < 0 || s_root[k] < best_root)))) { best_cnt = s_cnt[k]; best_root = s_root[k]; }
  }
  for (int k = 0; k < 8; ++k) a.out[k] = 0;
  a.out[5] = n_planes; a.out[7] = n_comp;
  if (best_root < 0) return;
  double pl[4];
  (void)plane_from_moments(a.mom + (size_t)best_root * 9, best_cnt, pl);
  for (int k = 0; k < 4; ++k) a.out[k] = pl[k];
  a.out[4] = 1; a.out[6] = best_cnt;
}

}  // namespace esl

using namespace esl;

extern "C" void esl_plane_params_default(esl_plane_params* p) {
  p->min_size = 200; p->angle_threshold_deg = 5; p->distance_threshold = 0.1;   // Example/param/TUM3.yaml:36-38
  p->normal_smoothing = 10; p->max_depth_change_factor = 0.05;                   // PlaneExtractor.cpp:57-58
  p->min_inliers = 100;                                                           // PlaneExtractor.cpp:74
  p->refine = 1; p->refine_distance = 0.02;                                       // segmentAndRefine (:82); PlaneRefinementComparator's default
  p->max_curvature = 0.001;                                                       // PCL's maximum_curvature_ default (the reference does not set it)
}

namespace esl {
void plane_release(esl_ctx* c) {
  if (c->plane_slab) (void)hipFree(c->plane_slab);
  c->plane_slab = nullptr; c->plane_slab_cap = 0;
}
}  // namespace esl

namespace {
// every buffer of the plane step is a slice of ONE grow-only slab kept by the context (round 2 paid 33 MB of hipMalloc / hipFree
// per call): depth u16, normal + offset 4 x f32, union-find parent i32, per-root count i32 and nine i64 moments, results, and
// (esl_extract_planes) the root -> plane map, the plane list and the label image
struct Buf { void* p = nullptr; };
struct PlaneWork { Buf depth, nrm, par, cnt, mom, out, map, list, labels, grown, blk; int list_cap = 0; };
int plane_reserve(esl_ctx* c, size_t npx, int max_planes, int model_min, PlaneWork& w) {
  auto al = [](size_t v) { return (v + 255) / 256 * 256; };
  // the list holds PCL's MODELS when the refinement pass runs (components of >= min_inliers pixels: at most npx / min_inliers)
  w.list_cap = std::max(std::max(max_planes, 1), model_min > 0 ? (int)(npx / (size_t)model_min) + 2 : 1);
  const size_t sz[11] = {al(npx * 2), al(npx * 16), al(npx * 4), al(npx * 4), al(npx * 72), al(8 * sizeof(double)), al(npx * 4),
                         al((size_t)w.list_cap * 5 * sizeof(double)), al(npx * 4), al((size_t)w.list_cap * sizeof(int)), al(((npx + 255) / 256 + 1) * sizeof(int))};
  size_t need = 0;
  for (size_t v : sz) need += v;
  if (need > c->plane_slab_cap) {
    ESL_HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->plane_slab) (void)hipFree(c->plane_slab);
    c->plane_slab = nullptr; c->plane_slab_cap = 0;
    ESL_HIP_TRY(hipMalloc((void**)&c->plane_slab, need));
    c->plane_slab_cap = need;
  }
  Buf* b[11] = {&w.depth, &w.nrm, &w.par, &w.cnt, &w.mom, &w.out, &w.map, &w.list, &w.labels, &w.grown, &w.blk};
  size_t off = 0;
  for (int k = 0; k < 11; ++k) { b[k]->p = c->plane_slab + off; off += sz[k]; }
  return ESL_OK;
}

// normals -> segments -> moments on the context's stream; fills `a`
int plane_segment(esl_ctx* c, const uint16_t* depth, int32_t width, int32_t height, const double intr[5], const esl_plane_params* p,
                  PlaneWork& w, PlaneArgs& a, const char* who, int max_planes = 1) {
  if (!c || !depth || !intr || !p || width <= 0 || height <= 0) { set_error(std::string(who) + ": bad argument"); return ESL_ERR_INVALID; }
  if (p->normal_smoothing < 2 || p->distance_threshold <= 0 || p->angle_threshold_deg <= 0 || p->max_depth_change_factor <= 0 ||
      (p->refine && (!(p->refine_distance > 0) || p->min_inliers < 1))) {
    set_error(std::string(who) + ": bad parameters"); return ESL_ERR_INVALID;
  }
  ESL_HIP_TRY(hipSetDevice(c->device));
  const size_t npx = (size_t)width * height;
  if (const int rc = plane_reserve(c, npx, max_planes, p->refine ? p->min_inliers : 0, w)) return rc;
  ESL_HIP_TRY(hipMemcpyAsync(w.depth.p, depth, npx * 2, hipMemcpyHostToDevice, c->stream));
  a = PlaneArgs{};
  a.depth = (const uint16_t*)w.depth.p; a.w = width; a.h = height;
  a.fx = intr[0]; a.fy = intr[1]; a.cx = intr[2]; a.cy = intr[3]; a.scale = intr[4];
  a.R = p->normal_smoothing / 2; a.depth_factor = p->max_depth_change_factor;
  a.cos_ang = std::cos(p->angle_threshold_deg * 0.017453);   // the reference's degree-to-radian constant (PlaneExtractor.cpp:75)
  a.dist_th = p->distance_threshold; a.min_size = p->min_size; a.min_inliers = p->min_inliers; a.max_curv = p->max_curvature;
  a.nrm = (float*)w.nrm.p; a.parent = (int*)w.par.p; a.cnt = (int*)w.cnt.p; a.mom = (long long*)w.mom.p; a.out = (double*)w.out.p; a.blk = (int*)w.blk.p;
  const unsigned nb = (unsigned)((npx + 255) / 256);
  {
    const int side = 16 + 2 * (a.R + 1);
    const size_t lds = (size_t)side * side * 20;
    if (lds <= 48 * 1024) hipLaunchKernelGGL(k_plane_normals_lds, dim3((width + 15) / 16, (height + 15) / 16), dim3(256), lds, c->stream, a);
    else hipLaunchKernelGGL(k_plane_normals, dim3((width + 15) / 16, (height + 15) / 16), dim3(256), 0, c->stream, a);
  }
  hipLaunchKernelGGL(k_plane_init, dim3(nb), dim3(256), 0, c->stream, a);
  static const int tiled = [] { const char* e = std::getenv("ESL_PLANE_UNION"); return (e && e[0] == '0') ? 0 : 1; }();
  if (tiled) {
    hipLaunchKernelGGL(k_plane_union_tiles, dim3((width + kPlaneTile - 1) / kPlaneTile, (height + kPlaneTile - 1) / kPlaneTile), dim3(256), 0, c->stream, a);
    hipLaunchKernelGGL(k_plane_union_borders, dim3(nb), dim3(256), 0, c->stream, a);
  } else {
    hipLaunchKernelGGL(k_plane_union, dim3(nb), dim3(256), 0, c->stream, a);
  }
  hipLaunchKernelGGL(k_plane_moments, dim3(nb), dim3(256), 0, c->stream, a);
  ESL_HIP_TRY(hipGetLastError());
  return ESL_OK;
}
}  // namespace

namespace {
// segmentAndRefine: models (>= min_inliers) -> label image -> the two refinement passes; on the host: refined sizes, the Plane.MinSize
// filter, the wall filter and the largest candidate (the part of extractPlanes / extractGroundPlane behind the PCL call; a handful of
// planes).  labels_out (optional): plane index per pixel, -1 none.
struct RefinedPlanes { std::vector<double> planes; std::vector<int> sizes; int ground = -1; };
int plane_refined(esl_ctx* c, const esl_plane_params* p, PlaneWork& w, PlaneArgs& a, RefinedPlanes& out, int32_t* labels_out) {
  const size_t npx = (size_t)a.w * a.h;
  if (a.w > kPlaneRefMaxW) { set_error("plane refinement: images wider than 4096 pixels are not supported"); return ESL_ERR_INVALID; }
  a.min_size = 0;                                      // PCL's models (plane_is_model), in raster order of their first pixel; Plane.MinSize applies to the REFINED sizes below
  a.plane_of_root = (int*)w.map.p; a.list = (double*)w.list.p; a.list_cap = w.list_cap; a.labels = (int*)w.labels.p;
  plane_list_launch(a, c->stream);
  hipLaunchKernelGGL(k_plane_lab_init, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, c->stream, a);
  ESL_HIP_TRY(hipMemsetAsync(w.grown.p, 0, (size_t)w.list_cap * sizeof(int), c->stream));
  if (a.w > 1 && a.h > 1) {
    if (a.w <= kPlaneRefRingW) {
      const size_t lds = plane_refine_lds(4, a.w);
      ESL_HIP_TRY(hipFuncSetAttribute((const void*)k_plane_refine<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k_plane_refine<4>, dim3(1), dim3(kPlaneRefThreads), lds, c->stream, a, (int*)w.grown.p, p->refine_distance, w.list_cap);
    } else {
      const size_t lds = plane_refine_lds(1, a.w);
      ESL_HIP_TRY(hipFuncSetAttribute((const void*)k_plane_refine<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k_plane_refine<1>, dim3(1), dim3(kPlaneRefThreads), lds, c->stream, a, (int*)w.grown.p, p->refine_distance, w.list_cap);
    }
  }
  ESL_HIP_TRY(hipGetLastError());
  double h[8];
  ESL_HIP_TRY(hipMemcpyAsync(h, w.out.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  ESL_HIP_TRY(hipStreamSynchronize(c->stream));
  const int n_models = std::min((int)h[5], w.list_cap);
  std::vector<double> list((size_t)std::max(n_models, 1) * 5);
  std::vector<int> grown((size_t)std::max(n_models, 1), 0), idx((size_t)std::max(n_models, 1), -1);
  if (n_models) {
    ESL_HIP_TRY(hipMemcpyAsync(list.data(), w.list.p, (size_t)n_models * 5 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    ESL_HIP_TRY(hipMemcpyAsync(grown.data(), w.grown.p, (size_t)n_models * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  }
  if (labels_out) ESL_HIP_TRY(hipMemcpyAsync(labels_out, w.labels.p, npx * 4, hipMemcpyDeviceToHost, c->stream));
  ESL_HIP_TRY(hipStreamSynchronize(c->stream));
  int best_cnt = 0;
  for (int m = 0; m < n_models; ++m) {
    const int size = (int)list[(size_t)m * 5 + 4] + grown[m];
    if (size < p->min_size) continue;                   // PlaneExtractor.cpp:87
    const double* pl = &list[(size_t)m * 5];
    idx[m] = (int)out.sizes.size();
    out.planes.insert(out.planes.end(), pl, pl + 4);
    out.sizes.push_back(size);
    const double th = std::acos(pl[1] / std::sqrt(pl[0] * pl[0] + pl[1] * pl[1] + pl[2] * pl[2]));
    if (th > M_PI / 4 && th < 3 * M_PI / 4) continue;    // the wall filter (:139-146)
    if (size > best_cnt) { best_cnt = size; out.ground = idx[m]; }   // most inliers (:160-162); ties: the earlier model
  }
  if (labels_out)
    for (size_t i = 0; i < npx; ++i) { const int l = labels_out[i]; labels_out[i] = (l >= 0 && l < n_models) ? idx[l] : -1; }
  return ESL_OK;
}
}  // namespace

extern "C" int esl_extract_ground_plane(esl_ctx* c, const uint16_t* depth, int32_t width, int32_t height, const double intr[5],
                                        const esl_plane_params* p, double plane_out[4], int32_t* ok, int32_t* n_planes, int32_t* n_pixels) {
  if (!plane_out || !ok) { set_error("esl_extract_ground_plane: bad argument"); return ESL_ERR_INVALID; }
  *ok = 0;
  for (int k = 0; k < 4; ++k) plane_out[k] = 0;
  PlaneWork w;
  PlaneArgs a;
  if (const int rc = plane_segment(c, depth, width, height, intr, p, w, a, "esl_extract_ground_plane")) return rc;
  if (p->refine) {
    RefinedPlanes rp;
    if (const int rc = plane_refined(c, p, w, a, rp, nullptr)) return rc;
    if (n_planes) *n_planes = (int32_t)rp.sizes.size();
    if (n_pixels) *n_pixels = rp.ground >= 0 ? rp.sizes[(size_t)rp.ground] : 0;
    if (rp.ground >= 0) { *ok = 1; for (int k = 0; k < 4; ++k) plane_out[k] = rp.planes[(size_t)rp.ground * 4 + k]; }
    return ESL_OK;
  }
  hipLaunchKernelGGL(k_plane_select, dim3(1), dim3(256), 0, c->stream, a);
  ESL_HIP_TRY(hipGetLastError());
  double h[8];
  ESL_HIP_TRY(hipMemcpyAsync(h, w.out.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  ESL_HIP_TRY(hipStreamSynchronize(c->stream));
  *ok = h[4] > 0.5 ? 1 : 0;
  if (*ok) for (int k = 0; k < 4; ++k) plane_out[k] = h[k];
  if (n_planes) *n_planes = (int32_t)h[5];
  if (n_pixels) *n_pixels = (int32_t)h[6];
  return ESL_OK;
}

extern "C" int esl_extract_planes(esl_ctx* c, const uint16_t* depth, int32_t width, int32_t height, const double intr[5],
                                  const esl_plane_params* p, int32_t max_planes, double* planes_out, int32_t* sizes_out,
                                  int32_t* n_planes, int32_t* labels_out) {
  if (!n_planes || max_planes < 0 || (max_planes > 0 && (!planes_out || !sizes_out))) { set_error("esl_extract_planes: bad argument"); return ESL_ERR_INVALID; }
  *n_planes = 0;
  PlaneWork w;
  PlaneArgs a;
  if (const int rc = plane_segment(c, depth, width, height, intr, p, w, a, "esl_extract_planes", max_planes)) return rc;
  if (p->refine) {
    RefinedPlanes rp;
    if (const int rc = plane_refined(c, p, w, a, rp, labels_out)) return rc;
    *n_planes = (int32_t)rp.sizes.size();
    const int n = std::min(*n_planes, max_planes);
    for (int i = 0; i < n; ++i) {
      for (int k = 0; k < 4; ++k) planes_out[(size_t)i * 4 + k] = rp.planes[(size_t)i * 4 + k];
      sizes_out[i] = rp.sizes[(size_t)i];
    }
    return ESL_OK;
  }
  const size_t npx = (size_t)width * height;
  a.plane_of_root = (int*)w.map.p; a.list = (double*)w.list.p; a.list_cap = max_planes;
  plane_list_launch(a, c->stream);
  if (labels_out) {
    a.labels = (int*)w.labels.p;
    hipLaunchKernelGGL(k_plane_labels, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, c->stream, a);
  }
  ESL_HIP_TRY(hipGetLastError());
  double h[8];
  std::vector<double> list((size_t)std::max(max_planes, 1) * 5);
  ESL_HIP_TRY(hipMemcpyAsync(h, w.out.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  ESL_HIP_TRY(hipMemcpyAsync(list.data(), w.list.p, list.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (labels_out) ESL_HIP_TRY(hipMemcpyAsync(labels_out, w.labels.p, npx * 4, hipMemcpyDeviceToHost, c->stream));
  ESL_HIP_TRY(hipStreamSynchronize(c->stream));
  *n_planes = (int32_t)h[5];
  const int n = std::min(*n_planes, max_planes);
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < 4; ++k) planes_out[(size_t)i * 4 + k] = list[(size_t)i * 5 + k];
    sizes_out[i] = (int32_t)list[(size_t)i * 5 + 4];
  }
  return ESL_OK;
}
