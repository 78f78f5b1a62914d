// esl_fit.hip — esl_fit_frame: the single-frame ellipsoid fit for all boxes of a frame (the boxes are independent: no
// collective, replicas only), a short pipeline of launches replayed from a captured hipGraph:
//   k_fit_scan / k_fit_plane          (tile, box) grids: depth samples -> voxel hash -> world frame -> supporting-plane filter
//   k_fit_cl_*                        boxes with many samples: centre + Euclidean clustering over (tile, box) grids
//   k_fit_pre                         one workgroup per box: (clustering in LDS for small boxes,) cluster choice, PCA +
//                                     gravity alignment, 10 cm grid, symmetry context
//   k_fit_sym                         one workgroup per (box, plane hypothesis): the 1-edge symmetry LM
//   k_fit_post                        one workgroup per box: mirrored completion, extents, ellipsoid
//
// Replaces EllipsoidSLAM::EllipsoidExtractor::EstimateLocalEllipsoid (reference
// src/pca/EllipsoidExtractor.cpp:292-493) with its helpers: getPointCloudInRect
// (src/symmetry/PointCloudFilter.cpp:21-56), DownSamplePointCloudOnly (:249-261, PCL VoxelGrid),
// transformPointCloud (src/core/Geometry.cpp:44-66), ApplySupportingPlaneFilter (:562-578), GetCenter
// (:583-643), ApplyEuclideanFilter (:646-737, PCL EuclideanClusterExtraction), ProcessPCA (:170-196),
// AdjustChirality / AlignZAxisToGravity / calibRotMatAccordingToGroundPlane (:760-788, 207-267),
// Symmetry::estimateSymmetry (src/symmetry/Symmetry.cpp:88-128) with SymmetrySolver::GetPointCloudProb
// (src/symmetry/SymmetrySolver.cpp:49-138) and the 1-edge g2o LM (:217-333), ProcessPCANormalized (:495-531).
//
// PCL's unpinned pieces are replaced by deterministic, order-independent equivalents (same definitions as
// the CPU checker): voxel grid = hash table keyed by floorf(p/leaf) with EXACT fixed-point (2^-30 m)
// integer atomics for the centroids; Euclidean clusters = connected components by lock-free union-find
// over a 2 cm cell hash; 1-NN = brute force over the <= few hundred 10 cm voxels.
//
// Inside a kernel the stages of a box are separated by a barrier + agent-scope acquire (stage_sync); reductions are wave
// shuffles + one LDS hop.  ESL_FIT_TIMING=1 prints a per-stage breakdown, ESL_FIT_NO_GRAPH=1 forces direct launches.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <climits>
#include <cmath>
#include <vector>

#include "esl_ctx.hpp"

namespace esl {

constexpr int kFitThreads = 1024;
constexpr unsigned long long kEmpty = 0xFFFFFFFFFFFFFFFFull;
constexpr int kKeyOff = 1 << 20;
#define ESL_FIX 1073741824.0  // 2^30

// Per-frame values live in device memory (uploaded with the depth image), NOT in the kernel arguments: the launches of two
// frames with the same geometry (boxes, image size, workspace capacity) are then identical, which is what lets
// esl_fit_frame replay them from one captured hipGraph.
struct FitFrame { double Twc[7], intr[5], ground[4]; esl_fit_params p; };
struct FitShared;
struct FitArgs {
  const uint16_t* depth; int w, h;
  const double* bboxes; const int* labels; int n_boxes;
  const struct FitFrame* fr;   // everything that changes from frame to frame (pose, intrinsics, ground plane, parameters)
  // workspace (per box strides)
  long cap, H;
  unsigned long long* hk; long long* hsx; long long* hsy; long long* hsz; unsigned int* hcnt;
  float* pwx; float* pwy; float* pwz; unsigned long long* pkey;
  unsigned long long* ck; int* chead; int* nxt; int* parent;
  int* csize; unsigned long long* cminkey; unsigned long long* cmind;
  double* po; float* pof;
  double* out_ell; double* out_prob; int* out_status; double* out_dbg;
  double* out_sym;   // 16 per box: SymmetryOutputData (src/symmetry/Symmetry.h:16-32), see include/esl.h
  int wide;   // != 0: centre + clustering already ran as grid-wide kernels (boxes with many samples)
  struct FitShared* state;   // per box: the first kernel's shared block, read by the symmetry and the completion kernels
  long long* clk;   // optional (ESL_FIT_TIMING=1): 16 wall_clock64() marks per box written by thread 0 at the stage boundaries
};

__device__ __forceinline__ unsigned long long hash64(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return k;
}
// Stage boundary.  All traffic between stages stays inside ONE workgroup (= one CU, one vector L1): the barrier's
// workgroup-scope release drains this wave's stores / atomics to L2, and an agent-scope ACQUIRE drops L1 lines that an
// L2 atomic of another wave may have outdated.  The agent-scope RELEASE of a full __threadfence() (L2 write-back,
// issued by all 1024 threads) is what cross-CU visibility would need -- not this kernel -- and cost 10-50 us per stage.
__device__ __forceinline__ void stage_sync() {
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// symmetry-grid points of the box (object frame, float) staged in LDS for the brute-force 1-NN of the symmetry cost
constexpr int kSymLds = 1024;
// One raw LDS buffer, carved per stage (the stages are separated by barriers):
//   clustering (M <= kClLds points):  cell keys u32[kClCells] | cell starts i32[kClCells + 1] | cell-sorted points float4[kClLds] | parent i32[kClLds]
//                       afterwards:   min centre distance u64[kClLds] | min voxel key u64[kClLds] | size i32[kClLds] | parent = root of every point
//   (the symmetry kernel's candidate points float4[kSymLds] have their own array, g_sym_cand)
constexpr int kClLds = 2048, kClCells = 4096;
constexpr unsigned int kEmpty32 = 0xFFFFFFFFu;
constexpr int kLdsKey = 0;                                   // cell keys u32[kClCells]            | later: min centre distance u64[kClLds]
constexpr int kLdsCst = kClCells * 4;                        // cell starts i32[kClCells + 1]      | later: min voxel key u64[kClLds]
constexpr int kLdsPts = kLdsCst + kClCells * 4 + 256;        // cell-sorted points float4[kClLds]  | later: cluster size i32[kClLds]
constexpr int kLdsPar = kLdsPts + kClLds * 16;               // union-find parent i32[kClLds] (root of every point at the end)
__shared__ __attribute__((aligned(16))) unsigned char g_fit_lds[kLdsPar + kClLds * 4];
// the symmetry kernel's 1-NN candidates (its own array: k_fit_sym must not inherit the clustering tables' 72 KB)
__shared__ float4 g_sym_cand[1024];
// the box's points (world frame) for the LDS clustering path and what follows it in k_fit_pre: x[kClLds] | y[kClLds] | z[kClLds].
// Round 3 read them from global memory inside the neighbour-list walks -- one dependent L2 round trip per candidate pair.
// (gfx950 gives a workgroup up to 160 KB; k_fit_pre now takes 75 KB and runs one 1,024-thread workgroup per CU as before.)
__shared__ float g_fit_xyz[3 * kClLds];
// per-thread strips of neighbour indices found by the LDS clustering path (k_fit_pre)
constexpr int kNbCap = 16;
__shared__ unsigned short g_fit_nb[kFitThreads * kNbCap];
static_assert(kSymLds <= 1024, "symmetry candidates (float4) must fit g_sym_cand");
#define ESL_FIT_MARK(k) do { if (a.clk && tid == 0) a.clk[16 * b + (k)] = (long long)wall_clock64(); } while (0)

// block-wide sum, result to every thread (wave shuffle + LDS)
__device__ double block_sum(double v, double* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0;
  for (int i = 0; i < kFitThreads / 64; ++i) s += red[i];
  return s;
}
// N sums at once with ONE barrier pair (red: N x 16 doubles); same per-value summation order as block_sum
template <int N>
__device__ void block_sum_n(double (&v)[N], double* red) {
#pragma unroll
  for (int k = 0; k < N; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < N; ++k) red[k * (kFitThreads / 64) + (threadIdx.x >> 6)] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double s = 0;
    for (int i = 0; i < kFitThreads / 64; ++i) s += red[k * (kFitThreads / 64) + i];
    v[k] = s;
  }
}
__device__ double block_max(double v, double* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = red[0];
  for (int i = 1; i < kFitThreads / 64; ++i) s = fmax(s, red[i]);
  return s;
}

__device__ __forceinline__ unsigned long long vox_key(float x, float y, float z, float inv) {
  const int ix = (int)floorf(x * inv), iy = (int)floorf(y * inv), iz = (int)floorf(z * inv);
  return ((unsigned long long)(iz + kKeyOff) << 42) | ((unsigned long long)(iy + kKeyOff) << 21) | (unsigned long long)(ix + kKeyOff);
}
__device__ __forceinline__ void vox_insert(const FitArgs& a, long base, unsigned long long key, float x, float y, float z) {
  unsigned long long slot = hash64(key) & (unsigned long long)(a.H - 1);
  for (;;) {
    const unsigned long long prev = atomicCAS(&a.hk[base + slot], kEmpty, key);
    if (prev == kEmpty || prev == key) break;
    slot = (slot + 1) & (unsigned long long)(a.H - 1);
  }
  atomicAdd((unsigned long long*)&a.hsx[base + slot], (unsigned long long)llrint((double)x * ESL_FIX));
  atomicAdd((unsigned long long*)&a.hsy[base + slot], (unsigned long long)llrint((double)y * ESL_FIX));
  atomicAdd((unsigned long long*)&a.hsz[base + slot], (unsigned long long)llrint((double)z * ESL_FIX));
  atomicAdd(&a.hcnt[base + slot], 1u);
}

__device__ __forceinline__ unsigned long long cell_key(float x, float y, float z, double tol, int dx, int dy, int dz) {
  const long long cx = (long long)floor((double)x / tol) + dx, cy = (long long)floor((double)y / tol) + dy,
                  cz = (long long)floor((double)z / tol) + dz;
  return ((unsigned long long)(cz + kKeyOff) << 42) | ((unsigned long long)(cy + kKeyOff) << 21) | (unsigned long long)(cx + kKeyOff);
}
// find with path halving: a node is re-pointed at its grandparent.  Linking by index (larger root under smaller) alone
// grows chains hundreds of hops long inside a 4.5k-point cluster, and every hop is an L2 round trip.  The re-pointing is
// a plain (relaxed atomic) STORE: only roots are ever targets of the linking CAS and a non-root never becomes a root
// again, so the only concurrent writers of parent[i] are other halvings, all of them writing ancestors of i -- and a
// same-address read-modify-write costs ~10 ns of serialisation on this part where a store does not.
__device__ __forceinline__ int uf_find(int* parent, int i) {
  for (;;) {
    const int p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p == i) return i;
    const int gp = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gp == p) return p;
    __hip_atomic_store(&parent[i], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    i = gp;
  }
}

// The same walk on CACHED loads (workgroup-scope relaxed atomics = plain loads the compiler may not fold): an agent-scope load
// goes past this XCD's L2 to the fabric for every hop (~1 - 2 us, and a find is a chain of them); a cached copy of the forest may be
// stale, but a stale parent is an OLDER ancestor of the same node (roots only ever gain parents, halving only re-points at
// ancestors), so the walk still ends at an ancestor: "same ancestor" proves the two points are joined, and a stale "root" is
// caught by the linking CAS, whose return value is the truth (cl_union continues from it).  Only the union stage may use this --
// the statistics stage of the next kernel needs the real roots.
__device__ __forceinline__ int uf_find_cached(int* parent, int i) {
  for (;;) {
    const int p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (p == i) return i;
    const int gp = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (gp == p) return p;
    __hip_atomic_store(&parent[i], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    i = gp;
  }
}

__device__ __forceinline__ int uf_find_lds(int* parent, int i) {   // same, forest in LDS
  for (;;) {
    const int p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (p == i) return i;
    const int gp = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (gp == p) return p;
    (void)atomicCAS(&parent[i], p, gp);
    i = gp;
  }
}

// ---- tiny host/device linear algebra used by lane 0 -------------------------------------------------------
__device__ void jacobi3(const double* Ain, double* w, double* V) {
  double A[9];
  for (int i = 0; i < 9; ++i) A[i] = Ain[i];
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
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
  for (int i = 0; i < 3; ++i) w[i] = A[i * 4];
  for (int i = 0; i < 3; ++i) {
    int m = i;
    for (int j = i + 1; j < 3; ++j)
      if (w[j] < w[m]) m = j;
    if (m != i) {
      const double t = w[i]; w[i] = w[m]; w[m] = t;
      for (int k = 0; k < 3; ++k) { const double u = V[k * 3 + i]; V[k * 3 + i] = V[k * 3 + m]; V[k * 3 + m] = u; }
    }
  }
  for (int j = 0; j < 3; ++j) {
    int m = 0;
    for (int k = 1; k < 3; ++k)
      if (fabs(V[k * 3 + j]) > fabs(V[m * 3 + j])) m = k;
    if (V[m * 3 + j] < 0)
      for (int k = 0; k < 3; ++k) V[k * 3 + j] = -V[k * 3 + j];
  }
}
__device__ __forceinline__ void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ void xform(const Mat3& R, const double* t, const double* p, double* o) {
  o[0] = R.m[0] * p[0] + R.m[1] * p[1] + R.m[2] * p[2] + t[0];
  o[1] = R.m[3] * p[0] + R.m[4] * p[1] + R.m[5] * p[2] + t[1];
  o[2] = R.m[6] * p[0] + R.m[7] * p[1] + R.m[8] * p[2] + t[2];
}

// ---- g2o::plane (include/core/Plane.h:46-129) --------------------------------------------------------------
struct PlaneT { double p[4]; double dual; };
// g2o::plane::oplus(Vector3(az, 0, d)) as the symmetry LM uses it (elevation update always 0; Plane.cpp / SymmetrySolver
// retraction): the reference goes through atan2 -> cos/sin of the current normal's azimuth and elevation; those are
// algebraic in the normal (cos(atan2(y,x)) = x/r, ...), so only sincos(az) of the UPDATE is transcendental here.
// Same values to rounding; an FP64 atan2/sin/cos chain is ~2k serial instructions per call on one wave.
__device__ __forceinline__ void plane_update(PlaneT& pl, const double* u, bool dual) {
  const double az = u[0], dd = u[1];
  const double r2 = pl.p[0] * pl.p[0] + pl.p[1] * pl.p[1];
  const double r = sqrt(r2), n3 = sqrt(r2 + pl.p[2] * pl.p[2]);
  const double ca = r > 0 ? pl.p[0] / r : 1.0, sa = r > 0 ? pl.p[1] / r : 0.0;   // azimuth of the current normal
  const double cb = n3 > 0 ? r / n3 : 1.0, sb = n3 > 0 ? -pl.p[2] / n3 : 0.0;     // cos / sin of MINUS its elevation
  double sz, cz;
  sincos(az, &sz, &cz);
  const double q[3] = {ca * cb * cz - sa * sz, sa * cb * cz + ca * sz, -sb * cz};
  const double d = -pl.p[3] + dd;
  pl.p[0] = q[0]; pl.p[1] = q[1]; pl.p[2] = q[2]; pl.p[3] = -d;
  const double inv = 1. / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  for (int i = 0; i < 4; ++i) pl.p[i] *= inv;
  if (dual) pl.dual += u[2];
}
__device__ __forceinline__ void plane_another(const PlaneT& pl, double* out) {
  const double r = sqrt(pl.p[0] * pl.p[0] + pl.p[1] * pl.p[1]);
  out[0] = r > 0 ? pl.p[1] / r : 0.0; out[1] = r > 0 ? -pl.p[0] / r : -1.0; out[2] = 0; out[3] = -pl.dual;   // (sin az, -cos az, 0, -d2)
}
__device__ __forceinline__ void mirror_point(const double* p, const double* pl, double* r) {
  const double nn = sqrt(pl[0] * pl[0] + pl[1] * pl[1] + pl[2] * pl[2]);
  const double n[3] = {pl[0] / nn, pl[1] / nn, pl[2] / nn};
  const double sv = pl[0] * p[0] + pl[1] * p[1] + pl[2] * p[2] + pl[3];
  const double dis = fabs(sv) / nn;
  const double symbol = sv > 0 ? -1 : 1;
  r[0] = p[0] + 2 * symbol * dis * n[0]; r[1] = p[1] + 2 * symbol * dis * n[1]; r[2] = p[2] + 2 * symbol * dis * n[2];
}

struct SymCtx {
  const uint16_t* depth; int w, h;
  int bb[4];
  double K[4], scale, sigma;
  double P[12], camc[3];
  const double* po; const float* pof; int n;
};

// edge error (= -mean ln P) of one plane hypothesis, evaluated by ONE WAVE (lanes over mirrored points)
__device__ double sym_error_wave(const SymCtx& c, const PlaneT& pl, bool dual) {
  const int lane = threadIdx.x & 63;
  double p2[4];
  if (dual) plane_another(pl, p2);
  const int ns = dual ? 2 * c.n : c.n;
  double ln_total = 0;
  double invalid = 0;
  for (int i = lane; i < ns; i += 64) {
    const double* src = c.po + 3 * (long)(i < c.n ? i : i - c.n);
    double p[3];
    mirror_point(src, (i < c.n) ? pl.p : p2, p);
    double uh[3];
    for (int r = 0; r < 3; ++r) uh[r] = c.P[r * 4] * p[0] + c.P[r * 4 + 1] * p[1] + c.P[r * 4 + 2] * p[2] + c.P[r * 4 + 3];
    const double u = uh[0] / uh[2], v = uh[1] / uh[2];
    bool use_nn = true;
    const bool finite_pt = isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]);
    if (isfinite(u) && isfinite(v) && fabs(u) < 1e9 && fabs(v) < 1e9) {
      const int x = (int)u, y = (int)v;
      if (c.bb[0] < x && x < c.bb[2] && c.bb[1] < y && y < c.bb[3] && x >= 0 && y >= 0 && x < c.w && y < c.h) {
        const uint16_t d = c.depth[(size_t)y * c.w + x];
        const double realz = (double)d * sqrt((x - c.K[2]) * (x - c.K[2]) + c.K[0] * c.K[0] + (y - c.K[3]) * (y - c.K[3])) / c.K[0];
        const uint16_t dp = (uint16_t)realz;
        if (dp == 0) use_nn = false;
        else {
          const double depth = dp / c.scale;
          const double dx = c.camc[0] - p[0], dy = c.camc[1] - p[1], dz = c.camc[2] - p[2];
          if (sqrt(dx * dx + dy * dy + dz * dz) > depth) use_nn = false;
        }
      }
    }
    double dis = 0;
    if (use_nn) {
      if (!finite_pt) invalid += 1;
      else {
        double best = 1e300;
        if (c.n <= kSymLds) {
          // candidates broadcast from LDS as float4 (one ds_read_b128 each), four independent running minima: the plain
          // loop (three scalar LDS reads and a compare-and-branch per candidate) cost ~0.25 us PER CANDIDATE and was the
          // whole symmetry stage (22 us per evaluation round at 46 points)
          const float4* cand = g_sym_cand;
          double b0 = 1e300, b1 = 1e300, b2 = 1e300, b3 = 1e300;
          int j = 0;
          for (; j + 4 <= c.n; j += 4) {
            const float4 q0 = cand[j], q1 = cand[j + 1], q2 = cand[j + 2], q3 = cand[j + 3];
            const double x0 = p[0] - (double)q0.x, y0 = p[1] - (double)q0.y, z0 = p[2] - (double)q0.z;
            const double x1 = p[0] - (double)q1.x, y1 = p[1] - (double)q1.y, z1 = p[2] - (double)q1.z;
            const double x2 = p[0] - (double)q2.x, y2 = p[1] - (double)q2.y, z2 = p[2] - (double)q2.z;
            const double x3 = p[0] - (double)q3.x, y3 = p[1] - (double)q3.y, z3 = p[2] - (double)q3.z;
            b0 = fmin(b0, x0 * x0 + y0 * y0 + z0 * z0);
            b1 = fmin(b1, x1 * x1 + y1 * y1 + z1 * z1);
            b2 = fmin(b2, x2 * x2 + y2 * y2 + z2 * z2);
            b3 = fmin(b3, x3 * x3 + y3 * y3 + z3 * z3);
          }
          for (; j < c.n; ++j) {
            const float4 q0 = cand[j];
            const double x0 = p[0] - (double)q0.x, y0 = p[1] - (double)q0.y, z0 = p[2] - (double)q0.z;
            b0 = fmin(b0, x0 * x0 + y0 * y0 + z0 * z0);
          }
          best = fmin(fmin(b0, b1), fmin(b2, b3));
        } else {
          for (int j = 0; j < c.n; ++j) {
            const double dx = p[0] - (double)c.pof[3 * j], dy = p[1] - (double)c.pof[3 * j + 1], dz = p[2] - (double)c.pof[3 * j + 2];
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (d2 < best) best = d2;
          }
        }
        dis = sqrt(best);
      }
    }
    const double si = 1.0 / c.sigma;
    ln_total += -0.5 * si * si * dis * dis;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { ln_total += __shfl_xor(ln_total, off, 64); invalid += __shfl_xor(invalid, off, 64); }
  const double valid = (double)ns - invalid;
  const double aver = valid > 0 ? ln_total / valid : -INFINITY;
  return -aver;
}

// Eigen-style pivoted LDLT for n <= 3 (what g2o's LinearSolverDense runs on the 2x2 / 3x3 system).  N is a template parameter and
// every loop is unrolled with the pivot position resolved by compile-time cases: all indices are static, the matrix lives in
// registers.  (Round 3's runtime-n form indexed its arrays dynamically -> 184 bytes of scratch per lane in k_fit_sym, a dozen
// dependent scratch round trips in front of every trial evaluation of the symmetry LM.)  Same operations in the same order.
template <int N>
__device__ __forceinline__ bool ldlt_small(double* A, const double* b, double* x) {
  int tr[N];
  int sign = 0;
  bool stop = false;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if (stop) break;
    int big = k;
    double bv = fabs(A[k * N + k]);
#pragma unroll
    for (int i = k + 1; i < N; ++i) { const double v = fabs(A[i * N + i]); if (v > bv) { bv = v; big = i; } }
    tr[k] = big;
#pragma unroll
    for (int c = k + 1; c < N; ++c)
      if (big == c) {   // symmetric swap of rows / columns k and c of the lower triangle
#pragma unroll
        for (int j = 0; j < k; ++j) { const double t = A[k * N + j]; A[k * N + j] = A[c * N + j]; A[c * N + j] = t; }
#pragma unroll
        for (int i = c + 1; i < N; ++i) { const double t = A[i * N + k]; A[i * N + k] = A[i * N + c]; A[i * N + c] = t; }
#pragma unroll
        for (int i = k + 1; i < c; ++i) { const double t = A[i * N + k]; A[i * N + k] = A[c * N + i]; A[c * N + i] = t; }
        const double t = A[k * N + k]; A[k * N + k] = A[c * N + c]; A[c * N + c] = t;
      }
    if (k > 0) {
      double temp[N], sacc = 0;
#pragma unroll
      for (int j = 0; j < k; ++j) { temp[j] = A[j * N + j] * A[k * N + j]; sacc += A[k * N + j] * temp[j]; }
      A[k * N + k] -= sacc;
#pragma unroll
      for (int i = k + 1; i < N; ++i) {
        double t = 0;
#pragma unroll
        for (int j = 0; j < k; ++j) t += A[i * N + j] * temp[j];
        A[i * N + k] -= t;
      }
    }
    const double akk = A[k * N + k];
    const bool valid = fabs(akk) > 0;
    if (k == 0 && !valid) {
      sign = 0;
#pragma unroll
      for (int j = 0; j < N; ++j) tr[j] = j;
      stop = true;
    } else {
      if (valid) {
#pragma unroll
        for (int i = k + 1; i < N; ++i) A[i * N + k] /= akk;
      }
      if (sign == 1) { if (akk < 0) sign = 2; }
      else if (sign == -1) { if (akk > 0) sign = 2; }
      else if (sign == 0) { if (akk > 0) sign = 1; else if (akk < 0) sign = -1; }
    }
  }
  const bool positive = (sign == 1 || sign == 0);
  if (positive) {
    auto swap_x = [&](int i) {   // x[i] <-> x[tr[i]], tr[i] >= i
#pragma unroll
      for (int c = 0; c < N; ++c) if (c > i && tr[i] == c) { const double t = x[i]; x[i] = x[c]; x[c] = t; }
    };
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = b[i];
#pragma unroll
    for (int i = 0; i < N; ++i) swap_x(i);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double sacc = x[i];
#pragma unroll
      for (int j = 0; j < i; ++j) sacc -= A[i * N + j] * x[j];
      x[i] = sacc;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) { const double d = A[i * N + i]; x[i] = (fabs(d) > 2.2250738585072014e-308) ? x[i] / d : 0.0; }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
      double sacc = x[i];
#pragma unroll
      for (int j = i + 1; j < N; ++j) sacc -= A[j * N + i] * x[j];
      x[i] = sacc;
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) swap_x(i);
  }
  return positive;
}


__device__ __forceinline__ int symmetry_type(int label) {  // EllipsoidExtractor::LoadSymmetryPrior (:52-79)
  switch (label) {
    case 58: return 0;
    case 59: case 62: case 57: case 66: case 63: case 64: case 41: return 1;
    case 28: return 2;
    default: return -1;
  }
}

struct FitShared {
  double red[6 * (kFitThreads / 64)];
  int M, n0, n1, ncl, chosen, maxsize, only, status, ns0, npo, cnt;
  int cmin[3], cmax[3];   // cell-coordinate extent of the box's points (LDS clustering path)
  int nc, pre_ok, run_sym, stype;   // carried from the first kernel to the later ones
  unsigned long long minkey;
  double center[3], cen[3], cov[6];
  double Two[7], Tow[7];
  double Row[9];
  double prob[9];
  PlaneT planes[9];
  double best_plane[5];
  double Rmo[9], Tmo[3];
  SymCtx sc;
};

// ---- Euclidean clustering on global-memory tables (cell hash -> per-cell lists -> lock-free union-find) -------
// Thread t0 of `stride` cooperating threads; one workgroup (the box's) or a whole (tile, box) grid.
__device__ void cl_init(const FitArgs& a, long base, long pbase, int M, long t0, long stride) {
  for (long s = t0; s < a.H; s += stride) { a.ck[base + s] = kEmpty; a.chead[base + s] = -1; }
  for (long i = t0; i < M; i += stride) { a.parent[pbase + i] = (int)i; a.csize[pbase + i] = 0; a.cminkey[pbase + i] = kEmpty; a.cmind[pbase + i] = kEmpty; }
}
__device__ void cl_insert(const FitArgs& a, long base, long pbase, int M, double tol, long t0, long stride) {
  for (long i = t0; i < M; i += stride) {
    const unsigned long long key = cell_key(a.pwx[pbase + i], a.pwy[pbase + i], a.pwz[pbase + i], tol, 0, 0, 0);
    unsigned long long slot = hash64(key) & (unsigned long long)(a.H - 1);
    for (;;) {
      const unsigned long long prev = atomicCAS(&a.ck[base + slot], kEmpty, key);
      if (prev == kEmpty || prev == key) break;
      slot = (slot + 1) & (unsigned long long)(a.H - 1);
    }
    a.nxt[pbase + i] = atomicExch(&a.chead[base + slot], (int)i);
  }
}
// work item = (point i, one of its 27 neighbour cells): a point's own chain of 27 probes + list walks + unions was a
// 300 us serial path per thread at 8.5k points; split this way the longest chain is one cell's list
__device__ void cl_union(const FitArgs& a, long base, long pbase, int M, double tol, long t0, long stride) {
  const double tol2 = tol * tol;
  // every pair of neighbouring cells is visited from ONE side: the point's own cell (with j > i) and the 13 cells after it in
  // (z, y, x) order -- the same pairs as all 27 cells with j > i
  for (long w = t0; w < 14L * M; w += stride) {
    const int i = (int)(w / 14), cidx = (int)(w % 14) + 13;   // 13 = the own cell in the 3 x 3 x 3 numbering
    const int dx = cidx % 3 - 1, dy = (cidx / 3) % 3 - 1, dz = cidx / 9 - 1;
    const bool own = cidx == 13;
    const float xi = a.pwx[pbase + i], yi = a.pwy[pbase + i], zi = a.pwz[pbase + i];
    const unsigned long long key = cell_key(xi, yi, zi, tol, dx, dy, dz);
    unsigned long long slot = hash64(key) & (unsigned long long)(a.H - 1);
    int head = -1;
    for (;;) {
      const unsigned long long k = a.ck[base + slot];   // cell table and lists are read-only in this stage: cached loads
      if (k == key) { head = a.chead[base + slot]; break; }
      if (k == kEmpty) break;
      slot = (slot + 1) & (unsigned long long)(a.H - 1);
    }
    int ri = i;   // last known root of i
    for (int j = head; j >= 0; j = a.nxt[pbase + j]) {
      if (own && j <= i) continue;
      const double ddx = (double)xi - a.pwx[pbase + j], ddy = (double)yi - a.pwy[pbase + j], ddz = (double)zi - a.pwz[pbase + j];
      if (ddx * ddx + ddy * ddy + ddz * ddz <= tol2) {
        // cheap pre-check with cached loads: equal parents (even stale ones) mean the two were joined already --
        // components only ever merge, so a stale value can cost a redundant union, never a wrong skip
        if (a.parent[pbase + ri] == a.parent[pbase + j]) continue;
        int ra = ri, rb = j;
        for (;;) {
          ra = uf_find_cached(a.parent + pbase, ra); rb = uf_find_cached(a.parent + pbase, rb);
          if (ra == rb) break;
          if (ra < rb) { const int t = ra; ra = rb; rb = t; }
          const int seen = atomicCAS(&a.parent[pbase + ra], ra, rb);
          if (seen == ra) { ra = rb; break; }
          // lost the race -- or uf_find walked a STALE copy of the forest: its loads are served by this XCD's L2, which another
          // XCD's hook does not update (the eight L2s are not coherent with each other; measured in round 4 on the Cholesky's
          // flag words: such a line can stay stale for as long as nothing evicts it).  The atomic's return value is the truth:
          // continue from ra's real parent instead of re-reading the stale line and failing the same CAS again -- the
          // 150 - 400 us "contention" of this kernel at 8.5k points was this loop
          ra = seen;
        }
        ri = ra;
      }
    }
  }
}
__device__ void cl_stats(const FitArgs& a, long pbase, int M, const double* center, long t0, long stride) {
  // all lanes of a wave run the same number of rounds (t0's of a wave are consecutive, M is uniform)
  for (long i0 = t0 - (threadIdx.x & 63); i0 < M; i0 += stride) {
    const long i = i0 + (threadIdx.x & 63);
    const bool on = i < M;
    int r = -1;
    unsigned long long key = kEmpty, dbits = kEmpty;
    if (on) {
      r = uf_find(a.parent + pbase, (int)i);
      a.nxt[pbase + i] = r;  // root of every point (the list links are no longer needed)
      key = a.pkey[pbase + i];
      const double dx = center[0] - a.pwx[pbase + i], dy = center[1] - a.pwy[pbase + i], dz = center[2] - a.pwz[pbase + i];
      dbits = (unsigned long long)__double_as_longlong(sqrt(dx * dx + dy * dy + dz * dz));
    }
    // most waves see ONE root (the big cluster): combine in the wave, one lane issues the three atomics
    const int r0 = __shfl(r, __ffsll((long long)__ballot(on)) - 1, 64);
    if (__all(!on || r == r0)) {
      int cnt = on ? 1 : 0;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_xor(cnt, off, 64);
        const unsigned long long k2 = __shfl_xor(key, off, 64), d2 = __shfl_xor(dbits, off, 64);
        key = k2 < key ? k2 : key; dbits = d2 < dbits ? d2 : dbits;
      }
      if ((threadIdx.x & 63) == 0 && cnt > 0) {
        atomicAdd(&a.csize[pbase + r0], cnt);
        atomicMin(&a.cminkey[pbase + r0], key);
        atomicMin(&a.cmind[pbase + r0], dbits);
      }
    } else if (on) {
      atomicAdd(&a.csize[pbase + r], 1);
      atomicMin(&a.cminkey[pbase + r], key);
      atomicMin(&a.cmind[pbase + r], dbits);
    }
  }
}

// ---- 1-4 as grid-wide kernels: blockIdx.y = box, blockIdx.x = tile of the box's samples / hash slots ----------
// (a single 50k-sample box kept ONE compute unit busy for 0.7 ms here; the tables are global-memory hashes built with
// atomics anyway, so nothing ties these stages to one workgroup)
static __global__ __launch_bounds__(256) void k_fit_scan(FitArgs a) {
  const int b = blockIdx.y;
  const long base = (long)b * a.H;
  const double* bbox = a.bboxes + 4 * b;
  const double fx = a.fr->intr[0], fy = a.fr->intr[1], cx = a.fr->intr[2], cy = a.fr->intr[3], scale = a.fr->intr[4];
  if (a.clk && blockIdx.x == 0 && threadIdx.x == 0) a.clk[16 * b + 0] = (long long)wall_clock64();
  // 1 + 2. scan the box, voxel-hash at voxel_leaf
  const int x1 = (int)bbox[0], y1 = (int)bbox[1], x2 = (int)bbox[2], y2 = (int)bbox[3];
  const int st = a.fr->p.stride;
  const int nsx = x2 > x1 ? (x2 - x1 + st - 1) / st : 0, nsy = y2 > y1 ? (y2 - y1 + st - 1) / st : 0;
  const float inv = 1.0f / (float)a.fr->p.voxel_leaf;
  int mine = 0;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < (long)nsx * nsy; idx += (long)gridDim.x * blockDim.x) {
    const int x = x1 + (int)(idx % nsx) * st, y = y1 + (int)(idx / nsx) * st;
    if (x < 0 || y < 0 || x >= a.w || y >= a.h) continue;
    const uint16_t d = a.depth[(size_t)y * a.w + x];
    const double z = d / scale;
    if (z <= a.fr->p.depth_min || z > a.fr->p.depth_max) continue;
    const float px = (float)((x - cx) * z / fx), py = (float)((y - cy) * z / fy), pz = (float)z;
    vox_insert(a, base, vox_key(px, py, pz, inv), px, py, pz);
    ++mine;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&a.state[b].n0, mine);
}
static __global__ __launch_bounds__(256) void k_fit_plane(FitArgs a) {
  const int b = blockIdx.y;
  const long base = (long)b * a.H, pbase = (long)b * a.cap;
  if (a.clk && blockIdx.x == 0 && threadIdx.x == 0) a.clk[16 * b + 1] = (long long)wall_clock64();
  // 3 + 4. centroids -> world -> supporting-plane filter
  const SE3 Twc = se3_load(a.fr->Twc);
  const Mat3 Rwc = q_to_R(Twc.r);
  const double gn = sqrt(a.fr->ground[0] * a.fr->ground[0] + a.fr->ground[1] * a.fr->ground[1] + a.fr->ground[2] * a.fr->ground[2]);
  // whole waves walk the table together (H and the strides are multiples of 64), so the two counters take ONE atomic per
  // wave and round instead of one per voxel (8.5k same-address atomics for a 50k-sample box)
  __shared__ int s_cnt[12];
  const int lane = threadIdx.x & 63;
  for (long s0 = (long)blockIdx.x * blockDim.x + threadIdx.x - lane; s0 < a.H; s0 += (long)gridDim.x * blockDim.x) {
    const long s = s0 + lane;
    const unsigned int cnt = s < a.H ? a.hcnt[base + s] : 0u;
    const bool occ = cnt != 0;
    bool keep = false;
    double q[3] = {0, 0, 0};
    if (occ) {
      const double c = (double)cnt;
      const double p[3] = {(double)(float)((double)a.hsx[base + s] / c / ESL_FIX), (double)(float)((double)a.hsy[base + s] / c / ESL_FIX),
                           (double)(float)((double)a.hsz[base + s] / c / ESL_FIX)};
      xform(Rwc, Twc.t, p, q);
      const double dis = (a.fr->ground[0] * q[0] + a.fr->ground[1] * q[1] + a.fr->ground[2] * q[2] + a.fr->ground[3]) / gn;
      keep = dis > a.fr->p.plane_dist;
    }
    // same-address device atomics cost ~10 ns each however they are issued: one pair per WORKGROUP and round
    const unsigned long long m_occ = __ballot(occ), m_keep = __ballot(keep);
    const int wv = threadIdx.x >> 6;
    __syncthreads();   // (previous round's readers of s_cnt are done)
    if (lane == 0) { s_cnt[wv] = __popcll(m_occ); s_cnt[4 + wv] = __popcll(m_keep); }
    __syncthreads();
    if (threadIdx.x == 0) {
      const int t_occ = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3], t_keep = s_cnt[4] + s_cnt[5] + s_cnt[6] + s_cnt[7];
      if (t_occ) atomicAdd(&a.state[b].n1, t_occ);
      s_cnt[8] = t_keep ? atomicAdd(&a.state[b].M, t_keep) : 0;
    }
    __syncthreads();
    int base_i = s_cnt[8];
    for (int k = 0; k < wv; ++k) base_i += s_cnt[4 + k];
    if (keep) {
      const int i = base_i + __popcll(m_keep & ((1ull << lane) - 1ull));
      a.pwx[pbase + i] = (float)q[0]; a.pwy[pbase + i] = (float)q[1]; a.pwz[pbase + i] = (float)q[2];
      a.pkey[pbase + i] = a.hk[base + s];
    }
  }
}

// ---- 5-6 as grid-wide kernels (boxes with many samples): centre + table reset, cell hash, union, statistics ----
static __global__ __launch_bounds__(256) void k_fit_cl_init(FitArgs a) {
  __shared__ double red[4][4];
  const int b = blockIdx.y, tid = threadIdx.x;
  const long base = (long)b * a.H, pbase = (long)b * a.cap;
  FitShared* st = a.state + b;
  const int M = st->M;
  if (blockIdx.x == 0) {
    if (a.clk && tid == 0) a.clk[16 * b + 2] = (long long)wall_clock64();
    if (M < 1) { if (tid == 0) { a.out_status[b] = 4; a.out_dbg[16 * b] = st->n0; a.out_dbg[16 * b + 1] = st->n1; } }
    else {
      // 5. GetCenter: 10 x 10 samples around the box centre
      const double* bbox = a.bboxes + 4 * b;
      const double fx = a.fr->intr[0], fy = a.fr->intr[1], cx = a.fr->intr[2], cy = a.fr->intr[3], scale = a.fr->intr[4];
      double v[4] = {0, 0, 0, 0};
      if (tid < 100) {
        const int x = (int)((bbox[0] + bbox[2]) / 2.0), y = (int)((bbox[1] + bbox[3]) / 2.0);
        const int xd = (int)(fabs(bbox[0] - bbox[2]) / 4.0 / 10), yd = (int)(fabs(bbox[1] - bbox[3]) / 4.0 / 10);
        const int x_ = x + (tid / 10 - 5) * xd, y_ = y + (tid % 10 - 5) * yd;
        if (x_ >= 0 && y_ >= 0 && x_ < a.w && y_ < a.h) {
          const uint16_t d = a.depth[(size_t)y_ * a.w + x_];
          const float pz = (float)(d / scale);
          if (!(pz <= 0.1 || pz > a.fr->p.depth_max)) {
            v[0] = (float)((x_ - cx) * pz / fx); v[1] = (float)((y_ - cy) * pz / fy); v[2] = pz; v[3] = 1;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
        if ((tid & 63) == 0) red[tid >> 6][k] = v[k];
      }
      __syncthreads();
      if (tid == 0) {
        double t[4];
        for (int k = 0; k < 4; ++k) t[k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
        if (t[3] < 2) a.out_status[b] = 1;
        else {
          const SE3 Twc = se3_load(a.fr->Twc);
          const Mat3 Rwc = q_to_R(Twc.r);
          const double c[3] = {t[0] / t[3], t[1] / t[3], t[2] / t[3]};
          xform(Rwc, Twc.t, c, st->center);
        }
      }
    }
  }
  cl_init(a, base, pbase, M, (long)blockIdx.x * blockDim.x + tid, (long)gridDim.x * blockDim.x);
}
static __global__ __launch_bounds__(256) void k_fit_cl_insert(FitArgs a) {
  const int b = blockIdx.y;
  if (a.out_status[b] != 0) return;
  cl_insert(a, (long)b * a.H, (long)b * a.cap, a.state[b].M, a.fr->p.cluster_tolerance, (long)blockIdx.x * blockDim.x + threadIdx.x,
            (long)gridDim.x * blockDim.x);
}
static __global__ __launch_bounds__(256) void k_fit_cl_union(FitArgs a) {
  const int b = blockIdx.y;
  if (a.out_status[b] != 0) return;
  cl_union(a, (long)b * a.H, (long)b * a.cap, a.state[b].M, a.fr->p.cluster_tolerance, (long)blockIdx.x * blockDim.x + threadIdx.x,
           (long)gridDim.x * blockDim.x);
}
static __global__ __launch_bounds__(256) void k_fit_cl_stats(FitArgs a) {
  const int b = blockIdx.y;
  if (a.out_status[b] != 0) return;
  cl_stats(a, (long)b * a.cap, a.state[b].M, a.state[b].center, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

static __global__ __launch_bounds__(kFitThreads) void k_fit_pre(FitArgs a) {
  __shared__ FitShared S;
  const int b = blockIdx.x, tid = threadIdx.x;
  const long base = (long)b * a.H, pbase = (long)b * a.cap;
  const SE3 Twc = se3_load(a.fr->Twc);
  const Mat3 Rwc = q_to_R(Twc.r);
  const double gn = sqrt(a.fr->ground[0] * a.fr->ground[0] + a.fr->ground[1] * a.fr->ground[1] + a.fr->ground[2] * a.fr->ground[2]);
  const double* bbox = a.bboxes + 4 * b;
  const double fx = a.fr->intr[0], fy = a.fr->intr[1], cx = a.fr->intr[2], cy = a.fr->intr[3], scale = a.fr->intr[4];
  if (tid == 0) { S.M = a.state[b].M; S.n0 = a.state[b].n0; S.n1 = a.state[b].n1; for (int k = 0; k < 3; ++k) S.center[k] = a.state[b].center[k]; S.ncl = 0; S.chosen = -1; S.maxsize = 0; S.only = -1; S.status = 0; S.ns0 = 0; S.npo = 0; S.cnt = 0; S.minkey = kEmpty; }
  __syncthreads();
  const int M = S.M;
  if (M < 1) { if (tid == 0) { a.out_status[b] = 4; a.out_dbg[16 * b] = S.n0; a.out_dbg[16 * b + 1] = S.n1; } return; }
  float* lx = g_fit_xyz; float* ly = g_fit_xyz + kClLds; float* lz = g_fit_xyz + 2 * kClLds;
  bool pts_in_lds = false;   // the LDS clustering path ran: points in lx / ly / lz, root of point i in lroot[i]
  int* lroot = (int*)(g_fit_lds + kLdsPar);
  ESL_FIT_MARK(2);
  if (a.wide && a.out_status[b] != 0) return;   // the grid-wide centre stage already gave up on this box
  if (!a.wide) {
  // 5. GetCenter: 10 x 10 samples around the box centre (first two waves)
  {
    double sx = 0, sy = 0, sz = 0, cn = 0;
    if (tid < 100) {
      const int x = (int)((bbox[0] + bbox[2]) / 2.0), y = (int)((bbox[1] + bbox[3]) / 2.0);
      const int xd = (int)(fabs(bbox[0] - bbox[2]) / 4.0 / 10), yd = (int)(fabs(bbox[1] - bbox[3]) / 4.0 / 10);
      const int x_ = x + (tid / 10 - 5) * xd, y_ = y + (tid % 10 - 5) * yd;
      if (x_ >= 0 && y_ >= 0 && x_ < a.w && y_ < a.h) {
        const uint16_t d = a.depth[(size_t)y_ * a.w + x_];
        const float pz = (float)(d / scale);
        if (!(pz <= 0.1 || pz > a.fr->p.depth_max)) {
          sx = (float)((x_ - cx) * pz / fx); sy = (float)((y_ - cy) * pz / fy); sz = pz; cn = 1;
        }
      }
    }
    { double v4[4] = {sx, sy, sz, cn}; block_sum_n<4>(v4, S.red); sx = v4[0]; sy = v4[1]; sz = v4[2]; cn = v4[3]; }   // (one barrier pair, same per-value summation order as four block_sum calls)
    if (cn < 2) { if (tid == 0) a.out_status[b] = 1; return; }
    if (tid == 0) { const double c[3] = {sx / cn, sy / cn, sz / cn}; xform(Rwc, Twc.t, c, S.center); }
  }
  ESL_FIT_MARK(3);
  // 6. Euclidean clustering: cell hash + lock-free union-find
  const double tol = a.fr->p.cluster_tolerance, tol2 = tol * tol;
  // Same definition on two substrates: with M <= kClLds points and a cell extent that fits 10 bits per axis the cell
  // hash, the per-cell lists and the union-find forest live in LDS (the global-memory version is pointer chasing
  // through L2 at ~1 us a hop: 27 cells x probe + list walk per point made this stage 150-330 us per box).
  if (tid == 0) { for (int k = 0; k < 3; ++k) { S.cmin[k] = INT_MAX; S.cmax[k] = INT_MIN; } }
  __syncthreads();
  {
    int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
    for (int i = tid; i < M; i += kFitThreads) {
      const float px = a.pwx[pbase + i], py = a.pwy[pbase + i], pz = a.pwz[pbase + i];
      if (M <= kClLds) { lx[i] = px; ly[i] = py; lz[i] = pz; }
      const int cc[3] = {(int)(long long)floor((double)px / tol), (int)(long long)floor((double)py / tol), (int)(long long)floor((double)pz / tol)};
#pragma unroll
      for (int k = 0; k < 3; ++k) { lo[k] = min(lo[k], cc[k]); hi[k] = max(hi[k], cc[k]); }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {   // one LDS atomic per wave and axis instead of six per point
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) { lo[k] = min(lo[k], __shfl_xor(lo[k], off, 64)); hi[k] = max(hi[k], __shfl_xor(hi[k], off, 64)); }
      if ((tid & 63) == 0) { atomicMin(&S.cmin[k], lo[k]); atomicMax(&S.cmax[k], hi[k]); }
    }
  }
  __syncthreads();
  const bool lds_path = M <= kClLds && (long)S.cmax[0] - S.cmin[0] < 1020 && (long)S.cmax[1] - S.cmin[1] < 1020 &&
                        (long)S.cmax[2] - S.cmin[2] < 1020;
  if (lds_path) {
    // Cells as CONTIGUOUS runs of a cell-sorted copy of the points (round 4; round 3 chained a cell's points through a linked
    // list and read their coordinates from global memory): count per cell while inserting (the atomic's return value is the
    // point's rank in its cell), exclusive scan of the counts over the table, scatter (x, y, z, index) as one float4 per point --
    // a cell is then [cst[slot], cst[slot + 1]) and a candidate is one 16-byte LDS read.  The order inside a cell depends on the
    // atomics' order; the components and their roots (= smallest index: larger roots are always hooked under smaller ones) do
    // not.  Measured at BASELINE's 20-box frame (stage = tables + search + linking, mean per box): 46.6 us -> 51 us with this
    // alone (the list walk was NOT the cost) -> 37 us with search and linking in separate loops (below) -> half-space search.
    unsigned int* ckey = (unsigned int*)(g_fit_lds + kLdsKey);
    int* cst = (int*)(g_fit_lds + kLdsCst);
    float4* spts = (float4*)(g_fit_lds + kLdsPts);
    int* lpar = (int*)(g_fit_lds + kLdsPar);
    int* wsum = (int*)S.red;                                                  // 16 wave totals of the scan
    const int mx = S.cmin[0] - 1, my = S.cmin[1] - 1, mz = S.cmin[2] - 1;   // neighbour offsets stay >= 0
    for (int sidx = tid; sidx <= kClCells; sidx += kFitThreads) { if (sidx < kClCells) ckey[sidx] = kEmpty32; cst[sidx] = 0; }
    for (int i = tid; i < M; i += kFitThreads) lpar[i] = i;
    __syncthreads();
    auto key_of = [&](float x, float y, float z, int dx, int dy, int dz) -> unsigned int {
      const int cx_ = (int)(long long)floor((double)x / tol) + dx - mx, cy_ = (int)(long long)floor((double)y / tol) + dy - my,
                cz_ = (int)(long long)floor((double)z / tol) + dz - mz;
      return ((unsigned int)cz_ << 20) | ((unsigned int)cy_ << 10) | (unsigned int)cx_;
    };
    static_assert(kClLds <= 2 * kFitThreads && kClCells == 4 * kFitThreads, "two points and four table slots per thread");
    int my_slot[2] = {0, 0}, my_rank[2] = {0, 0};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int i = tid + t * kFitThreads;
      if (i < M) {
        const unsigned int key = key_of(lx[i], ly[i], lz[i], 0, 0, 0);
        unsigned int slot = (unsigned int)hash64(key) & (kClCells - 1);
        for (;;) {
          const unsigned int prev = atomicCAS(&ckey[slot], kEmpty32, key);
          if (prev == kEmpty32 || prev == key) break;
          slot = (slot + 1) & (kClCells - 1);
        }
        my_slot[t] = (int)slot; my_rank[t] = atomicAdd(&cst[slot], 1);
      }
    }
    __syncthreads();
    {  // exclusive scan of the per-slot counts, in place; cst[kClCells] = M
      const int v0 = cst[4 * tid], v1 = cst[4 * tid + 1], v2 = cst[4 * tid + 2], v3 = cst[4 * tid + 3];
      const int mine = v0 + v1 + v2 + v3;
      int incl = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const int up = __shfl_up(incl, off, 64); if ((tid & 63) >= off) incl += up; }
      if ((tid & 63) == 63) wsum[tid >> 6] = incl;
      __syncthreads();
      int base_w = 0;
      for (int w = 0; w < (tid >> 6); ++w) base_w += wsum[w];
      const int ex = base_w + incl - mine;
      cst[4 * tid] = ex; cst[4 * tid + 1] = ex + v0; cst[4 * tid + 2] = ex + v0 + v1; cst[4 * tid + 3] = ex + v0 + v1 + v2;
      if (tid == kFitThreads - 1) cst[kClCells] = ex + mine;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int i = tid + t * kFitThreads;
      if (i < M) spts[cst[my_slot[t]] + my_rank[t]] = make_float4(lx[i], ly[i], lz[i], __int_as_float(i));
    }
    __syncthreads();
    // Neighbour search and linking are SEPARATE loops.  Fused (round 3), the 64 lanes of a wave met their few true neighbours at
    // different candidates, so nearly every one of a point's ~200 candidate steps carried some lane's union (two finds + a CAS:
    // ~1,300 cycles of dependent LDS round trips) for the whole wave -- measured 80 us for a 1,628-voxel box.  Now a lane only
    // RECORDS the neighbours it finds (16-bit indices in a private LDS strip) and links them afterwards in a loop whose trip
    // count is the largest neighbour count of the wave (~10), not the candidate count.
    unsigned short* nb = (unsigned short*)g_fit_nb + tid * kNbCap;
    const float tol2_hi = (float)(tol2 * (1.0 + 1e-5)), tol2_lo = (float)(tol2 * (1.0 - 1e-5));
    auto link_all = [&](int i, int cnt) {
      for (int k = 0; k < cnt; ++k) {
        int ra = i, rb = nb[k];
        for (;;) {
          ra = uf_find_lds(lpar, ra); rb = uf_find_lds(lpar, rb);
          if (ra == rb) break;
          if (ra < rb) { const int tt = ra; ra = rb; rb = tt; }
          if (atomicCAS(&lpar[ra], ra, rb) == ra) break;
        }
      }
    };
#pragma unroll 1
    for (int t = 0; t < 2; ++t) {
      const int i = tid + t * kFitThreads;
      if (i >= M) continue;
      const float xi = lx[i], yi = ly[i], zi = lz[i];
      int cnt = 0;
      // every pair of neighbouring cells is visited from ONE side (the 13 cells "after" the point's own in (z, y, x) order) and the
      // own cell with j > i: the same set of pairs as all 27 cells with j > i, half the probes and half the candidates
      for (int dz = 0; dz <= 1; ++dz)
        for (int dy = (dz == 0 ? 0 : -1); dy <= 1; ++dy)
          for (int dx = ((dz == 0 && dy == 0) ? 0 : -1); dx <= 1; ++dx) {
            const bool own = dz == 0 && dy == 0 && dx == 0;
            const unsigned int key = key_of(xi, yi, zi, dx, dy, dz);
            unsigned int slot = (unsigned int)hash64(key) & (kClCells - 1);
            int e0 = 0, e1 = 0;
            for (;;) {
              const unsigned int k = ckey[slot];
              if (k == key) { e0 = cst[slot]; e1 = cst[slot + 1]; break; }
              if (k == kEmpty32) break;
              slot = (slot + 1) & (kClCells - 1);
            }
            for (int e = e0; e < e1; ++e) {
              // the candidate scan is ISSUE-bound on the box's one CU (1,628 points x ~216 candidates): a single-precision screen
              // (relative error < 1e-6) settles all but the candidates within 1e-5 of the radius; those take the reference's
              // double-precision test, so every decision is the reference's
              const float4 q = spts[e];
              const float fx = xi - q.x, fy = yi - q.y, fz = zi - q.z;
              const float d2f = fx * fx + fy * fy + fz * fz;
              if (d2f > tol2_hi) continue;
              const int j = __float_as_int(q.w);
              if (own && j <= i) continue;
              bool in = d2f < tol2_lo;
              if (!in) {
                const double ddx = (double)xi - q.x, ddy = (double)yi - q.y, ddz = (double)zi - q.z;
                in = ddx * ddx + ddy * ddy + ddz * ddz <= tol2;
              }
              if (in) {
                if (cnt == kNbCap) { link_all(i, cnt); cnt = 0; }   // (a strip holds kNbCap neighbours: rare)
                nb[cnt++] = (unsigned short)j;
              }
            }
          }
      link_all(i, cnt);
    }
    __syncthreads();
    ESL_FIT_MARK(9);
    // per-root statistics (the cell tables and the sorted copy are dead now: their space holds the three statistics arrays)
    unsigned long long* lmind = (unsigned long long*)(g_fit_lds + kLdsKey);
    unsigned long long* lmink = (unsigned long long*)(g_fit_lds + kLdsCst);
    int* lsize = (int*)(g_fit_lds + kLdsPts);
    for (int i = tid; i < M; i += kFitThreads) { lmind[i] = kEmpty; lmink[i] = kEmpty; lsize[i] = 0; }
    __syncthreads();
    for (int i = tid; i < M; i += kFitThreads) {
      const int r = uf_find_lds(lpar, i);
      a.nxt[pbase + i] = r;  // root of every point
      atomicAdd(&lsize[r], 1);
      atomicMin(&lmink[r], a.pkey[pbase + i]);
      const double dx = S.center[0] - lx[i], dy = S.center[1] - ly[i], dz = S.center[2] - lz[i];
      const double d = sqrt(dx * dx + dy * dy + dz * dz);
      atomicMin(&lmind[r], (unsigned long long)__double_as_longlong(d));
      lroot[i] = r;   // (= the forest's parent array: pointing i at its root is a valid compression while others still search)
    }
    pts_in_lds = true;
    __syncthreads();
    // (k_fit_post reads the sizes; the choice below reads the LDS copies: no global round trip, no stage_sync)
    for (int i = tid; i < M; i += kFitThreads) { a.csize[pbase + i] = lsize[i]; a.cminkey[pbase + i] = lmink[i]; a.cmind[pbase + i] = lmind[i]; }
  } else {
    cl_init(a, base, pbase, M, tid, kFitThreads);
    stage_sync();
    cl_insert(a, base, pbase, M, tol, tid, kFitThreads);
    stage_sync();
    cl_union(a, base, pbase, M, tol, tid, kFitThreads);
    stage_sync();
    cl_stats(a, pbase, M, S.center, tid, kFitThreads);
    stage_sync();
  }
  }   // !a.wide
  // choose the cluster: the only one, else the largest (ties: smaller voxel key) within center_dis of the centre
  const unsigned long long* s_mind = (const unsigned long long*)(g_fit_lds + kLdsKey);
  const unsigned long long* s_mink = (const unsigned long long*)(g_fit_lds + kLdsCst);
  const int* s_size = (const int*)(g_fit_lds + kLdsPts);
  auto size_of = [&](int r) { return pts_in_lds ? s_size[r] : __hip_atomic_load(&a.csize[pbase + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto mind_of = [&](int r) {
    return __longlong_as_double((long long)(pts_in_lds ? s_mind[r] : __hip_atomic_load(&a.cmind[pbase + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
  };
  auto mink_of = [&](int r) { return pts_in_lds ? s_mink[r] : __hip_atomic_load(&a.cminkey[pbase + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  for (int r = tid; r < M; r += kFitThreads) {
    const int sz = size_of(r);
    if (sz >= a.fr->p.min_cluster_size) {
      atomicAdd(&S.ncl, 1);
      atomicMax(&S.only, r);
      if (mind_of(r) < a.fr->p.center_dis) atomicMax(&S.maxsize, sz);
    }
  }
  __syncthreads();
  for (int r = tid; r < M; r += kFitThreads) {
    const int sz = size_of(r);
    if (sz >= a.fr->p.min_cluster_size && sz == S.maxsize && mind_of(r) < a.fr->p.center_dis) atomicMin(&S.minkey, mink_of(r));
  }
  __syncthreads();
  for (int r = tid; r < M; r += kFitThreads) {
    const int sz = size_of(r);
    if (sz >= a.fr->p.min_cluster_size && sz == S.maxsize && S.maxsize > 0 && mink_of(r) == S.minkey) S.chosen = r;
  }
  __syncthreads();
  if (tid == 0 && S.ncl == 1) S.chosen = S.only;
  __syncthreads();
  const int chosen = S.chosen;
  if (chosen < 0) {
    if (tid == 0) { a.out_status[b] = 2; a.out_dbg[16 * b] = S.n0; a.out_dbg[16 * b + 1] = S.n1; a.out_dbg[16 * b + 2] = M; a.out_dbg[16 * b + 3] = S.ncl; }
    return;
  }
  const int nc = size_of(chosen);
  ESL_FIT_MARK(4);
  // 7. PCA of the chosen cluster (two passes like PCL: centroid, then normalised covariance)
  {
    double sx = 0, sy = 0, sz = 0;
    for (int i = tid; i < M; i += kFitThreads) {
      if (pts_in_lds) { if (lroot[i] == chosen) { sx += lx[i]; sy += ly[i]; sz += lz[i]; } }
      else if (a.nxt[pbase + i] == chosen) { sx += a.pwx[pbase + i]; sy += a.pwy[pbase + i]; sz += a.pwz[pbase + i]; }
    }
    { double v3[3] = {sx, sy, sz}; block_sum_n<3>(v3, S.red); sx = v3[0]; sy = v3[1]; sz = v3[2]; }
    const double cen[3] = {sx / nc, sy / nc, sz / nc};
    double cv[6] = {0, 0, 0, 0, 0, 0};
    for (int i = tid; i < M; i += kFitThreads)
      if ((pts_in_lds ? lroot[i] : a.nxt[pbase + i]) == chosen) {
        const float px = pts_in_lds ? lx[i] : a.pwx[pbase + i], py = pts_in_lds ? ly[i] : a.pwy[pbase + i], pz = pts_in_lds ? lz[i] : a.pwz[pbase + i];
        const double d0 = px - cen[0], d1 = py - cen[1], d2 = pz - cen[2];
        cv[0] += d0 * d0; cv[1] += d0 * d1; cv[2] += d0 * d2; cv[3] += d1 * d1; cv[4] += d1 * d2; cv[5] += d2 * d2;
      }
    __syncthreads();   // S.red is reused
    block_sum_n<6>(cv, S.red);
    if (tid == 0) {
      const double cov[9] = {cv[0] / nc, cv[1] / nc, cv[2] / nc, cv[1] / nc, cv[3] / nc, cv[4] / nc, cv[2] / nc, cv[4] / nc, cv[5] / nc};
      double ev[3], rot[9];
      jacobi3(cov, ev, rot);
      {  // AdjustChirality
        const double c0[3] = {rot[0], rot[3], rot[6]}, c1[3] = {rot[1], rot[4], rot[7]};
        double c2[3];
        cross3(c0, c1, c2);
        rot[2] = c2[0]; rot[5] = c2[1]; rot[8] = c2[2];
      }
      const double nh[3] = {a.fr->ground[0] / gn, a.fr->ground[1] / gn, a.fr->ground[2] / gn};
      {  // AlignZAxisToGravity
        double maxc = 0; int maxid = -1; bool pos = true;
        for (int i = 0; i < 3; ++i) {
          const double c = rot[i] * nh[0] + rot[3 + i] * nh[1] + rot[6 + i] * nh[2];
          if (fabs(c) > maxc) { maxc = fabs(c); pos = c > 0; maxid = i; }
        }
        if (maxid < 0) maxid = 2;
        double z[3], x[3], y[3];
        for (int r = 0; r < 3; ++r) { z[r] = pos ? rot[r * 3 + maxid] : -rot[r * 3 + maxid]; x[r] = rot[r * 3 + (maxid + 1) % 3]; }
        cross3(z, x, y);
        for (int r = 0; r < 3; ++r) { rot[r * 3] = x[r]; rot[r * 3 + 1] = y[r]; rot[r * 3 + 2] = z[r]; }
      }
      {  // calibRotMatAccordingToGroundPlane
        const double z[3] = {rot[2], rot[5], rot[8]}, nrm[3] = {a.fr->ground[0], a.fr->ground[1], a.fr->ground[2]};
        double ax[3];
        cross3(z, nrm, ax);
        const double an2 = ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2];
        if (an2 > 0) { const double an = sqrt(an2); ax[0] /= an; ax[1] /= an; ax[2] /= an; }
        const double zn = sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
        const double ct = (nrm[0] * z[0] + nrm[1] * z[1] + nrm[2] * z[2]) / gn / zn;
        const double th = acos(ct), s = sin(th), c = cos(th);
        const double sa[3] = {s * ax[0], s * ax[1], s * ax[2]}, c1[3] = {(1 - c) * ax[0], (1 - c) * ax[1], (1 - c) * ax[2]};
        double Ra[9], t;
        t = c1[0] * ax[1]; Ra[1] = t - sa[2]; Ra[3] = t + sa[2];
        t = c1[0] * ax[2]; Ra[2] = t + sa[1]; Ra[6] = t - sa[1];
        t = c1[1] * ax[2]; Ra[5] = t - sa[0]; Ra[7] = t + sa[0];
        Ra[0] = c1[0] * ax[0] + c; Ra[4] = c1[1] * ax[1] + c; Ra[8] = c1[2] * ax[2] + c;
        double out[9];
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) out[i * 3 + j] = Ra[i * 3] * rot[j] + Ra[i * 3 + 1] * rot[3 + j] + Ra[i * 3 + 2] * rot[6 + j];
        for (int i = 0; i < 9; ++i) rot[i] = out[i];
      }
      const double xn = sqrt(rot[0] * rot[0] + rot[3] * rot[3] + rot[6] * rot[6]);
      const double x[3] = {rot[0] / xn, rot[3] / xn, rot[6] / xn};
      double y[3];
      cross3(nh, x, y);
      Mat3 Rwo;
      Rwo.m[0] = x[0]; Rwo.m[1] = y[0]; Rwo.m[2] = nh[0];
      Rwo.m[3] = x[1]; Rwo.m[4] = y[1]; Rwo.m[5] = nh[1];
      Rwo.m[6] = x[2]; Rwo.m[7] = y[2]; Rwo.m[8] = nh[2];
      SE3 Two;
      Two.r = q_from_R(Rwo);
      Two.t[0] = cen[0]; Two.t[1] = cen[1]; Two.t[2] = cen[2];
      const SE3 Tow = se3_inv(Two);
      se3_store(Two, S.Two); se3_store(Tow, S.Tow);
      const Mat3 Row = q_to_R(Tow.r);
      for (int i = 0; i < 9; ++i) S.Row[i] = Row.m[i];
    }
  }
  ESL_FIT_MARK(5);
  // 8. voxel grid at symmetry_grid over the cluster, then into the object frame
  // (the grid has at most nc cells: a table of >= 2 nc slots carved from the box's voxel table is enough -- clearing and
  // scanning all H slots cost a 50k-sample box 120 us here)
  FitArgs ag = a;
  {
    long Hs = 1024;
    while (Hs < 2L * nc) Hs <<= 1;
    ag.H = Hs < a.H ? Hs : a.H;
  }
  double* po = a.po + 12 * pbase;   // up to 4 * cap points
  float* pof = a.pof + 3 * pbase;
  // a grid cell -> the point the symmetry stage sees: centroid (fixed-point sums, exact) rounded to float, into the object frame
  auto emit = [&](unsigned int cnt, long long sxv, long long syv, long long szv) {
    const double c = (double)cnt;
    const double p[3] = {(double)(float)((double)sxv / c / ESL_FIX), (double)(float)((double)syv / c / ESL_FIX), (double)(float)((double)szv / c / ESL_FIX)};
    const int i = atomicAdd(&S.ns0, 1);
    double q[3];
    q[0] = S.Row[0] * p[0] + S.Row[1] * p[1] + S.Row[2] * p[2] + S.Tow[0];
    q[1] = S.Row[3] * p[0] + S.Row[4] * p[1] + S.Row[5] * p[2] + S.Tow[1];
    q[2] = S.Row[6] * p[0] + S.Row[7] * p[1] + S.Row[8] * p[2] + S.Tow[2];
    po[3 * i] = q[0]; po[3 * i + 1] = q[1]; po[3 * i + 2] = q[2];
    pof[3 * i] = (float)q[0]; pof[3 * i + 1] = (float)q[1]; pof[3 * i + 2] = (float)q[2];
  };
  const float inv_grid = 1.0f / (float)a.fr->p.symmetry_grid;
  if (pts_in_lds && ag.H <= 2048) {
    // the grid's hash in LDS (round 4): the clustering tables are dead, the points and their roots are still there.  Same keys, same
    // slots, same exact sums as the global-memory table below -- three global-atomic round trips and two stage_syncs shorter.
    const int Hg = (int)ag.H;
    unsigned long long* gk = (unsigned long long*)g_fit_lds;
    unsigned long long* gsx = gk + Hg; unsigned long long* gsy = gk + 2 * Hg; unsigned long long* gsz = gk + 3 * Hg;   // 32 Hg <= 65,536 < kLdsPar
    unsigned int* gcnt = (unsigned int*)g_fit_nb;
    static_assert(32 * 2048 <= kLdsPar && 4 * 2048 <= (int)sizeof(g_fit_nb), "symmetry grid table must fit under the roots");
    for (int sidx = tid; sidx < Hg; sidx += kFitThreads) { gk[sidx] = kEmpty; gsx[sidx] = 0; gsy[sidx] = 0; gsz[sidx] = 0; gcnt[sidx] = 0; }
    __syncthreads();
    for (int i = tid; i < M; i += kFitThreads)
      if (lroot[i] == chosen) {
        const float px = lx[i], py = ly[i], pz = lz[i];
        const unsigned long long key = vox_key(px, py, pz, inv_grid);
        unsigned int slot = (unsigned int)(hash64(key) & (unsigned long long)(Hg - 1));
        for (;;) {
          const unsigned long long prev = atomicCAS(&gk[slot], kEmpty, key);
          if (prev == kEmpty || prev == key) break;
          slot = (slot + 1) & (unsigned int)(Hg - 1);
        }
        atomicAdd(&gsx[slot], (unsigned long long)llrint((double)px * ESL_FIX));
        atomicAdd(&gsy[slot], (unsigned long long)llrint((double)py * ESL_FIX));
        atomicAdd(&gsz[slot], (unsigned long long)llrint((double)pz * ESL_FIX));
        atomicAdd(&gcnt[slot], 1u);
      }
    __syncthreads();
    for (int sidx = tid; sidx < Hg; sidx += kFitThreads) {
      const unsigned int cnt = gcnt[sidx];
      if (cnt) emit(cnt, (long long)gsx[sidx], (long long)gsy[sidx], (long long)gsz[sidx]);
    }
    __syncthreads();
  } else {
  for (long s = tid; s < ag.H; s += kFitThreads) { a.hk[base + s] = kEmpty; a.hsx[base + s] = 0; a.hsy[base + s] = 0; a.hsz[base + s] = 0; a.hcnt[base + s] = 0; }
  stage_sync();
  {
    for (int i = tid; i < M; i += kFitThreads)
      if ((pts_in_lds ? lroot[i] : a.nxt[pbase + i]) == chosen) {
        const float px = pts_in_lds ? lx[i] : a.pwx[pbase + i], py = pts_in_lds ? ly[i] : a.pwy[pbase + i], pz = pts_in_lds ? lz[i] : a.pwz[pbase + i];
        vox_insert(ag, base, vox_key(px, py, pz, inv_grid), px, py, pz);
      }
  }
  stage_sync();
  for (long s = tid; s < ag.H; s += kFitThreads) {
    const unsigned int cnt = __hip_atomic_load(&a.hcnt[base + s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cnt) emit(cnt, a.hsx[base + s], a.hsy[base + s], a.hsz[base + s]);
  }
  stage_sync();
  }
  const int ns0 = S.ns0;
  ESL_FIT_MARK(6);
  // 9. symmetry: 9 hypotheses, one wavefront each
  const int stype = symmetry_type(a.labels ? a.labels[b] : -1);
  const bool run_sym = a.fr->p.symmetry_open && stype > 0 && ns0 > 0;
  if (run_sym) {
    if (tid == 0) {
      SymCtx& c = S.sc;
      c.depth = a.depth; c.w = a.w; c.h = a.h;
      for (int k = 0; k < 4; ++k) c.bb[k] = (int)bbox[k];
      c.K[0] = fx; c.K[1] = fy; c.K[2] = cx; c.K[3] = cy; c.scale = scale; c.sigma = a.fr->p.symmetry_sigma;
      const SE3 Toc = se3_mul(se3_load(S.Tow), se3_load(a.fr->Twc));
      const SE3 Tco = se3_inv(Toc);
      const Mat3 Rco = q_to_R(Tco.r);
      for (int cc = 0; cc < 4; ++cc) {
        const double c0 = cc < 3 ? Rco.m[cc] : Tco.t[0], c1 = cc < 3 ? Rco.m[3 + cc] : Tco.t[1], c2 = cc < 3 ? Rco.m[6 + cc] : Tco.t[2];
        c.P[cc] = fx * c0 + cx * c2; c.P[4 + cc] = fy * c1 + cy * c2; c.P[8 + cc] = c2;
      }
      c.camc[0] = Toc.t[0]; c.camc[1] = Toc.t[1]; c.camc[2] = Toc.t[2];
      c.po = po; c.pof = pof; c.n = ns0;
    }
  }
  // hand the box over to k_fit_sym / k_fit_post
  __syncthreads();
  if (tid == 0) { S.nc = nc; S.M = M; S.pre_ok = 1; S.run_sym = run_sym ? 1 : 0; S.stype = stype; }
  __syncthreads();
  {
    const unsigned long long* src = (const unsigned long long*)&S;
    unsigned long long* dst = (unsigned long long*)(a.state + b);
    for (int k = tid; k < (int)(sizeof(FitShared) / 8); k += kFitThreads) dst[k] = src[k];
  }
}

// ---- symmetry: one workgroup per (box, plane hypothesis), 8 waves -------------------------------------------
// SymmetrySolver's 1-edge LM (delta = 1e-9 central differences) is a chain of ~40 cost evaluations per hypothesis;
// run by a single wave inside the box's workgroup it was the longest stage of a frame (180 us mean, 460 us worst).
// Here the 1 + 2 dim evaluations of an iteration's linearisation run on separate waves at once, and so do the LM's
// trials: wave k evaluates the step for the lambda that k rejections in a row would lead to (lambda *= ni, ni *= 2),
// the results are then consumed in the sequential order -- same arithmetic, same accept/reject sequence, same e_last.
constexpr int kSymThreads = 512, kSymWaves = kSymThreads / 64;
static __global__ __launch_bounds__(kSymThreads) void k_fit_sym(FitArgs a) {
  __shared__ SymCtx sc;
  __shared__ double ev[kSymWaves], s_ok[kSymWaves], s_x[kSymWaves][3], s_pl[kSymWaves][5];
  const int b = blockIdx.x / 9, h = blockIdx.x % 9, tid = threadIdx.x, wv = tid >> 6;
  FitShared* st = a.state + b;
  if (!st->pre_ok || !st->run_sym) return;
  long long t_start = 0, t_eval = 0, t_book = 0; int n_rounds = 0;
  if (a.clk) t_start = (long long)wall_clock64();
  {
    const unsigned long long* src = (const unsigned long long*)&st->sc;
    unsigned long long* dst = (unsigned long long*)&sc;
    for (int k = tid; k < (int)(sizeof(SymCtx) / 8); k += kSymThreads) dst[k] = src[k];
  }
  __syncthreads();
  const int n = sc.n;
  if (n <= kSymLds)
    for (int i = tid; i < n; i += kSymThreads) g_sym_cand[i] = make_float4(sc.pof[3 * i], sc.pof[3 * i + 1], sc.pof[3 * i + 2], 0.f);
  __syncthreads();
  const bool dual = st->stype == 2;
  const int dim = dual ? 3 : 2, iters = a.fr->p.symmetry_lm_iters;
  const double delta = 1e-9;
  PlaneT pl;
  {
    const int i = h / 3, m = h % 3;
    const double dis = -0.2 + 0.2 * i, ang = -(M_PI / 180.0 * 5) + (M_PI / 180.0 * 5) * m;
    pl.p[0] = sin(ang); pl.p[1] = -cos(ang); pl.p[2] = 0; pl.p[3] = -dis; pl.dual = 0;
  }
  // every thread carries the LM state; the evaluations come from ev[] -> identical decisions in all threads.
  // (Measured and dropped in round 4: the NEXT iteration's linearisation issued inside the trial round, around the first trial's
  // plane, on waves 1 .. 2 dim -- it saves a round whenever the first trial is accepted, but leaves only 8 - 2 dim trials in the
  // first batch, and these problems reject often enough that the slowest hypothesis of a box went from 10 to 13 rounds:
  // 77 -> 100 us per frame at BASELINE's 20-box frame.)
  double lambda = 0, ni = 2;
  int nbad = 0;
  if (wv == 0) { const double e = sym_error_wave(sc, pl, dual); if ((tid & 63) == 0) ev[0] = e; }
  __syncthreads();
  double e_last = ev[0];
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    // linearisation: wave 0 -> E(pl), wave 1 + 2d -> E(pl + delta e_d), wave 2 + 2d -> E(pl - delta e_d)
    ++n_rounds;
    const long long te0 = a.clk ? (long long)wall_clock64() : 0;
    if (wv < 1 + 2 * dim) {
      PlaneT pp = pl;
      if (wv > 0) {
        double u[3] = {0, 0, 0};
        const int d = (wv - 1) >> 1;
        u[d] = ((wv - 1) & 1) ? -delta : delta;
        plane_update(pp, u, dual);
      }
      const double e = sym_error_wave(sc, pp, dual);
      if ((tid & 63) == 0) ev[wv] = e;
    }
    if (a.clk && tid == 0) t_eval += (long long)wall_clock64() - te0;
    __syncthreads();
    const long long tb0 = a.clk ? (long long)wall_clock64() : 0;
    const double e0 = ev[0];
    e_last = e0;
    double cur = e0 * e0;
    const double ini = cur;
    // (all arrays statically indexed -> registers: a runtime `dim` in the subscripts put them in scratch memory)
    double J[3] = {0, 0, 0}, H3[9], bvec[3] = {0, 0, 0};
#pragma unroll
    for (int d = 0; d < 3; ++d) if (d < dim) J[d] = (1.0 / (2 * delta)) * (ev[1 + 2 * d] - ev[2 + 2 * d]);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      bvec[r] = J[r] * (-(1.0 * e0));
#pragma unroll
      for (int k = 0; k < 3; ++k) H3[r * 3 + k] = J[r] * 1.0 * J[k];   // row / column 2 are zero when the hypothesis has two parameters
    }
    if (it == 0) {
      double md = 0;
#pragma unroll
      for (int r = 0; r < 3; ++r) if (r < dim) md = fmax(md, fabs(H3[r * 3 + r]));
      lambda = 1e-5 * md; ni = 2; nbad = 0;
    }
    double rho = 0;
    int q = 0;
    bool more = true;
    if (a.clk && tid == 0) t_book += (long long)wall_clock64() - tb0;
    while (more) {
      // speculative batch: wave k tries the lambda reached after k further rejections
      ++n_rounds;
      {
        double lam_k = lambda, ni_k = ni;
        for (int k = 0; k < wv; ++k) { lam_k *= ni_k; ni_k *= 2; }
        double x[3] = {0, 0, 0};
        bool ok;
        if (dual) {
          double Mx[9];
#pragma unroll
          for (int r = 0; r < 9; ++r) Mx[r] = H3[r];
          Mx[0] += lam_k; Mx[4] += lam_k; Mx[8] += lam_k;
          ok = ldlt_small<3>(Mx, bvec, x);
        } else {
          double Mx[4] = {H3[0] + lam_k, H3[1], H3[3], H3[4] + lam_k};
          ok = ldlt_small<2>(Mx, bvec, x);
        }
        PlaneT pt = pl;
        plane_update(pt, x, dual);
        const double e = sym_error_wave(sc, pt, dual);
        if ((tid & 63) == 0) {   // everything the sequential consumption needs from this trial
          ev[wv] = e; s_ok[wv] = ok ? 1.0 : 0.0;
          for (int r = 0; r < 3; ++r) s_x[wv][r] = x[r];
          for (int r = 0; r < 4; ++r) s_pl[wv][r] = pt.p[r];
          s_pl[wv][4] = pt.dual;
        }
      }
      __syncthreads();
      for (int k = 0; k < kSymWaves && more; ++k) {   // consume in the sequential order (a rejected trial costs ~30 instructions)
        const double et = ev[k];
        e_last = et;
        const double tmp = (s_ok[k] > 0.5) ? et * et : 1.7976931348623157e308;
        double scale = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r) if (r < dim) scale += s_x[k][r] * (lambda * s_x[k][r] + bvec[r]);   // lambda == wave k's lam_k here
        rho = (cur - tmp) / (scale + 1e-3);
        if (rho > 0 && isfinite(tmp)) {
          double alpha = 1. - pow((2 * rho - 1), 3);
          alpha = alpha < 2. / 3. ? alpha : 2. / 3.;
          lambda *= (1. / 3. > alpha ? 1. / 3. : alpha);
          ni = 2; cur = tmp;
          for (int r = 0; r < 4; ++r) pl.p[r] = s_pl[k][r];
          pl.dual = s_pl[k][4];
        } else { lambda *= ni; ni *= 2; }
        q++;
        more = (rho < 0 && q < 10);
      }
      __syncthreads();   // ev[] is rewritten by the next batch / the next linearisation
    }
    if (q == 10 || rho == 0) break;
    if ((ini - cur) * 1e3 < ini) nbad++; else nbad = 0;
    if (nbad >= 3) break;
  }
  if (tid == 0) { st->prob[h] = exp(-e_last); st->planes[h] = pl; }
  if (a.clk && tid == 0) {   // diagnostic: the slowest hypothesis of the box (duration in ticks, rounds)
    const long long dur = (long long)wall_clock64() - t_start;
    atomicMax((unsigned long long*)&a.clk[16 * b + 10], (unsigned long long)dur);
    atomicMax((unsigned long long*)&a.clk[16 * b + 11], (unsigned long long)n_rounds);
    atomicMax((unsigned long long*)&a.clk[16 * b + 12], (unsigned long long)n);
    atomicMax((unsigned long long*)&a.clk[16 * b + 13], (unsigned long long)t_eval);
    atomicMax((unsigned long long*)&a.clk[16 * b + 14], (unsigned long long)t_book);
  }
}

// ---- completion: best hypothesis, mirrored points, extents, ellipsoid (one workgroup per box) ----------------
static __global__ __launch_bounds__(kFitThreads) void k_fit_post(FitArgs a) {
  __shared__ FitShared S;
  const int b = blockIdx.x, tid = threadIdx.x;
  const long pbase = (long)b * a.cap;
  if (!a.state[b].pre_ok) return;   // the first kernel already reported why this box has no ellipsoid
  {
    const unsigned long long* src = (const unsigned long long*)(a.state + b);
    unsigned long long* dst = (unsigned long long*)&S;
    for (int k = tid; k < (int)(sizeof(FitShared) / 8); k += kFitThreads) dst[k] = src[k];
  }
  __syncthreads();
  double* po = a.po + 12 * pbase;
  const int M = S.M, nc = S.nc, ns0 = S.ns0, stype = S.stype;
  const bool run_sym = S.run_sym != 0;
  int npo = ns0;
  double prob_sym = 1.0;
  if (run_sym) {
    if (tid == 0) {
      int best = 0;
      for (int k = 1; k < 9; ++k) if (S.prob[k] > S.prob[best]) best = k;
      for (int k = 0; k < 4; ++k) S.best_plane[k] = S.planes[best].p[k];
      S.best_plane[4] = S.planes[best].dual;
    }
    __syncthreads();
    prob_sym = S.prob[0];
    for (int k = 1; k < 9; ++k) prob_sym = fmax(prob_sym, S.prob[k]);
    // complete the cloud with the mirrored points
    PlaneT best;
    for (int k = 0; k < 4; ++k) best.p[k] = S.best_plane[k];
    best.dual = S.best_plane[4];
    double p2[4];
    plane_another(best, p2);
    for (int i = tid; i < ns0; i += kFitThreads) {
      double m1[3];
      mirror_point(po + 3 * i, best.p, m1);
      po[3 * (ns0 + i)] = m1[0]; po[3 * (ns0 + i) + 1] = m1[1]; po[3 * (ns0 + i) + 2] = m1[2];
      if (stype == 2) {
        double m2[3], m3[3];
        mirror_point(po + 3 * i, p2, m2);
        mirror_point(m1, p2, m3);
        po[3 * (2 * ns0 + i)] = m2[0]; po[3 * (2 * ns0 + i) + 1] = m2[1]; po[3 * (2 * ns0 + i) + 2] = m2[2];
        po[3 * (3 * ns0 + i)] = m3[0]; po[3 * (3 * ns0 + i) + 1] = m3[1]; po[3 * (3 * ns0 + i) + 2] = m3[2];
      }
    }
    npo = (stype == 2) ? 4 * ns0 : 2 * ns0;
    stage_sync();
    double sx = 0, sy = 0, sz = 0;
    for (int i = tid; i < npo; i += kFitThreads) { sx += po[3 * i]; sy += po[3 * i + 1]; sz += po[3 * i + 2]; }
    sx = block_sum(sx, S.red); sy = block_sum(sy, S.red); sz = block_sum(sz, S.red);
    if (tid == 0) {
      const double cc[3] = {sx / (double)npo, sy / (double)npo, sz / (double)npo};
      {   // SymmetryOutputData (EllipsoidExtractor.cpp:376-393, 415-423): planes via plane::transform(*pSE3Two) =
          // (Two^T)^-1 pi  ->  n' = R n, d' = d - t . n'; centre of the completed cloud in the world frame
        double* so = a.out_sym + 16 * b;
        const SE3 Two = se3_load(S.Two);
        const Mat3 Rwo = q_to_R(Two.r);
        for (int pl = 0; pl < (stype == 2 ? 2 : 1); ++pl) {
          const double* q = pl ? p2 : best.p;
          double n[3];
          for (int r = 0; r < 3; ++r) n[r] = Rwo.m[3 * r] * q[0] + Rwo.m[3 * r + 1] * q[1] + Rwo.m[3 * r + 2] * q[2];
          so[4 * pl] = n[0]; so[4 * pl + 1] = n[1]; so[4 * pl + 2] = n[2];
          so[4 * pl + 3] = q[3] - (Two.t[0] * n[0] + Two.t[1] * n[1] + Two.t[2] * n[2]);
        }
        so[8] = prob_sym; so[9] = (double)stype; so[10] = 1.0;
        for (int r = 0; r < 3; ++r) so[11 + r] = Rwo.m[3 * r] * cc[0] + Rwo.m[3 * r + 1] * cc[1] + Rwo.m[3 * r + 2] * cc[2] + Two.t[r];
      }
      const double nn = sqrt(best.p[0] * best.p[0] + best.p[1] * best.p[1] + best.p[2] * best.p[2]);
      const double x[3] = {best.p[0] / nn, best.p[1] / nn, best.p[2] / nn}, z[3] = {0, 0, 1};
      double y[3];
      cross3(z, x, y);
      Mat3 Rom;
      Rom.m[0] = x[0]; Rom.m[1] = y[0]; Rom.m[2] = z[0];
      Rom.m[3] = x[1]; Rom.m[4] = y[1]; Rom.m[5] = z[1];
      Rom.m[6] = x[2]; Rom.m[7] = y[2]; Rom.m[8] = z[2];
      SE3 Tom;
      Tom.r = q_from_R(Rom);
      Tom.t[0] = cc[0]; Tom.t[1] = cc[1]; Tom.t[2] = cc[2];
      const SE3 Tmo = se3_inv(Tom);
      const Mat3 Rmo = q_to_R(Tmo.r);
      for (int i = 0; i < 9; ++i) S.Rmo[i] = Rmo.m[i];
      S.Tmo[0] = Tmo.t[0]; S.Tmo[1] = Tmo.t[1]; S.Tmo[2] = Tmo.t[2];
      const SE3 Twm = se3_mul(se3_load(S.Two), Tom);
      se3_store(Twm, S.Two);
    }
    __syncthreads();
  }
  ESL_FIT_MARK(7);
  // 10. extents in the (re-centred) object frame, ellipsoid, back to the camera frame
  double mx = 0, my = 0, mz = 0;
  for (int i = tid; i < npo; i += kFitThreads) {
    double p[3] = {po[3 * i], po[3 * i + 1], po[3 * i + 2]};
    if (run_sym) {
      const double q0 = S.Rmo[0] * p[0] + S.Rmo[1] * p[1] + S.Rmo[2] * p[2] + S.Tmo[0];
      const double q1 = S.Rmo[3] * p[0] + S.Rmo[4] * p[1] + S.Rmo[5] * p[2] + S.Tmo[1];
      const double q2 = S.Rmo[6] * p[0] + S.Rmo[7] * p[1] + S.Rmo[8] * p[2] + S.Tmo[2];
      p[0] = q0; p[1] = q1; p[2] = q2;
    }
    mx = fmax(mx, fabs(p[0])); my = fmax(my, fabs(p[1])); mz = fmax(mz, fabs(p[2]));
  }
  mx = block_max(mx, S.red); my = block_max(my, S.red); mz = block_max(mz, S.red);
  if (tid == 0) {
    SE3 I;
    I.r = Quat{0, 0, 0, 1}; I.t[0] = I.t[1] = I.t[2] = 0;
    const SE3 g = se3_mul(se3_load(S.Two), I);
    const SE3 l = se3_mul(se3_inv(se3_load(a.fr->Twc)), g);
    double* o = a.out_ell + 10 * b;
    se3_store(l, o);
    o[7] = mx; o[8] = my; o[9] = mz;
    a.out_prob[b] = prob_sym;
    a.out_status[b] = 0;
    ESL_FIT_MARK(8);
    double* dbg = a.out_dbg + 16 * b;
    dbg[0] = S.n0; dbg[1] = S.n1; dbg[2] = M; dbg[3] = S.ncl; dbg[4] = nc; dbg[5] = ns0;
    for (int k = 0; k < 9; ++k) dbg[6 + k] = run_sym ? S.prob[k] : 0.0;
  }
}

}  // namespace esl

using namespace esl;

// ---- captured launch sequences of esl_fit_frame, keyed by everything the sequence depends on -------------------------
struct FitGraphKey {
  int n_boxes, w, h; long cap; int sym_open; const void* slab; const void* in; const void* out;
  bool operator==(const FitGraphKey& o) const {
    return n_boxes == o.n_boxes && w == o.w && h == o.h && cap == o.cap && sym_open == o.sym_open && slab == o.slab && in == o.in && out == o.out;
  }
};
struct FitGraphEntry { FitGraphKey key; hipGraphExec_t exec; };
static std::vector<FitGraphEntry>& fit_graphs(esl_ctx* c) {
  if (!c->fit_graph_cache) c->fit_graph_cache = new std::vector<FitGraphEntry>();
  return *(std::vector<FitGraphEntry>*)c->fit_graph_cache;
}
static hipGraphExec_t fit_graph_find(esl_ctx* c, const FitGraphKey& k) {
  for (auto& e : fit_graphs(c)) if (e.key == k) return e.exec;
  return nullptr;
}
static void fit_graph_store(esl_ctx* c, const FitGraphKey& k, hipGraphExec_t exec) {
  auto& v = fit_graphs(c);
  if (v.size() >= 16) { (void)hipGraphExecDestroy(v.front().exec); v.erase(v.begin()); }   // small FIFO
  v.push_back({k, exec});
}
namespace esl {
void fit_graphs_clear(esl_ctx* c) {
  if (!c->fit_graph_cache) return;
  auto& v = fit_graphs(c);
  for (auto& e : v) (void)hipGraphExecDestroy(e.exec);
  v.clear();
}
void fit_release(esl_ctx* c) {   // called by esl_ctx_destroy
  fit_graphs_clear(c);
  delete (std::vector<FitGraphEntry>*)c->fit_graph_cache; c->fit_graph_cache = nullptr;
  if (c->fit_slab) { (void)hipFree(c->fit_slab); c->fit_slab = nullptr; c->fit_slab_cap = 0; }
  if (c->fit_in) { (void)hipHostFree(c->fit_in); c->fit_in = nullptr; c->fit_in_cap = 0; }
  if (c->fit_out) { (void)hipHostFree(c->fit_out); c->fit_out = nullptr; c->fit_out_cap = 0; }
}
}  // namespace esl

extern "C" {

void esl_fit_params_default(esl_fit_params* p) {
  p->stride = 3; p->depth_scale = 5000; p->depth_min = 0.1; p->depth_max = 6.0; p->voxel_leaf = 0.01;
  p->plane_dist = 0.05; p->cluster_tolerance = 0.02; p->min_cluster_size = 100; p->center_dis = 0.5;
  p->symmetry_open = 1; p->symmetry_grid = 0.1; p->symmetry_sigma = 0.1; p->symmetry_lm_iters = 5;
}

static int fit_frame_impl(esl_ctx* c, const uint16_t* depth, int32_t width, int32_t height, const double* bboxes,
                          const int32_t* labels, int32_t n_boxes, const double Twc[7], const double intr[5],
                          const double ground[4], const esl_fit_params* p, double* ellipsoids_out, double* prob_out,
                          int32_t* status_out, double* debug_out, esl_fit_symmetry* sym_out) {
  if (!c || !depth || !bboxes || !Twc || !intr || !ground || !p || !ellipsoids_out || !prob_out || !status_out || width <= 0 ||
      height <= 0 || n_boxes < 0) {
    set_error("esl_fit_frame: bad argument");
    return ESL_ERR_INVALID;
  }
  if (p->stride < 1 || p->voxel_leaf <= 0 || p->cluster_tolerance <= 0 || p->symmetry_grid <= 0) {
    set_error("esl_fit_frame: bad parameters");
    return ESL_ERR_INVALID;
  }
  if (n_boxes == 0) return ESL_OK;
  ESL_HIP_TRY(hipSetDevice(c->device));
  FitArgs a{};
  a.w = width; a.h = height; a.n_boxes = n_boxes;
  long cap = 64;
  for (int b = 0; b < n_boxes; ++b) {
    const long w = (long)(bboxes[4 * b + 2] - bboxes[4 * b]) / p->stride + 2, h = (long)(bboxes[4 * b + 3] - bboxes[4 * b + 1]) / p->stride + 2;
    if (w > 0 && h > 0) cap = std::max(cap, w * h);
  }
  {   // workspace capacity in powers of two: consecutive frames with similar boxes then share one launch geometry
    long q = 1024;
    while (q < cap) q <<= 1;
    cap = q;
  }
  const long H = 2 * cap;
  a.cap = cap; a.H = H;
  a.wide = cap > 4 * kClLds ? 1 : 0;   // a box may hold more voxels than the LDS clustering path takes: go grid-wide
  // one slab for everything; the inputs (per-frame values | boxes | labels | depth) and the outputs are contiguous blocks
  const size_t B = (size_t)n_boxes;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t o_fr = take(sizeof(FitFrame)), o_bb = take(B * 32), o_lab = take(B * 4), o_depth = take((size_t)width * height * 2);
  const size_t in_bytes = off;
  const size_t o_hk = take(B * H * 8), o_hsx = take(B * H * 8), o_hsy = take(B * H * 8), o_hsz = take(B * H * 8), o_hcnt = take(B * H * 4);
  const size_t o_pwx = take(B * cap * 4), o_pwy = take(B * cap * 4), o_pwz = take(B * cap * 4), o_pkey = take(B * cap * 8);
  const size_t o_ck = take(B * H * 8), o_chead = take(B * H * 4), o_nxt = take(B * cap * 4), o_par = take(B * cap * 4);
  const size_t o_cs = take(B * cap * 4), o_cmk = take(B * cap * 8), o_cmd = take(B * cap * 8);
  const size_t o_po = take(B * cap * 12 * 8), o_pof = take(B * cap * 3 * 4);
  const size_t o_state = take(B * sizeof(FitShared));
  const size_t o_ell = take(B * 80), o_prob = take(B * 8), o_st = take(B * 4), o_dbg = take(B * 128), o_clk = take(B * 128), o_sym = take(B * 128);
  const size_t out_bytes = off - o_ell;
  const bool timing = std::getenv("ESL_FIT_TIMING") != nullptr;
  // grow-only device slab and pinned staging blocks owned by the context: a per-frame call pays no allocation
  if (off > c->fit_slab_cap) {
    ESL_HIP_TRY(hipStreamSynchronize(c->stream));
    fit_graphs_clear(c);
    if (c->fit_slab) { (void)hipFree(c->fit_slab); c->fit_slab = nullptr; c->fit_slab_cap = 0; }
    const size_t want = off + off / 4;
    ESL_HIP_TRY(hipMalloc((void**)&c->fit_slab, want));
    c->fit_slab_cap = want;
  }
  if (in_bytes > c->fit_in_cap || out_bytes > c->fit_out_cap) {
    ESL_HIP_TRY(hipStreamSynchronize(c->stream));
    fit_graphs_clear(c);
    if (c->fit_in) { (void)hipHostFree(c->fit_in); c->fit_in = nullptr; }
    if (c->fit_out) { (void)hipHostFree(c->fit_out); c->fit_out = nullptr; }
    c->fit_in_cap = std::max(c->fit_in_cap, in_bytes + in_bytes / 4);
    c->fit_out_cap = std::max(c->fit_out_cap, out_bytes + out_bytes / 4);
    ESL_HIP_TRY(hipHostMalloc((void**)&c->fit_in, c->fit_in_cap, hipHostMallocDefault));
    ESL_HIP_TRY(hipHostMalloc((void**)&c->fit_out, c->fit_out_cap, hipHostMallocDefault));
  }
  char* slab = c->fit_slab;
  a.fr = (const FitFrame*)(slab + o_fr);
  a.depth = (const uint16_t*)(slab + o_depth); a.bboxes = (const double*)(slab + o_bb); a.labels = (const int*)(slab + o_lab);
  a.hk = (unsigned long long*)(slab + o_hk); a.hsx = (long long*)(slab + o_hsx); a.hsy = (long long*)(slab + o_hsy);
  a.hsz = (long long*)(slab + o_hsz); a.hcnt = (unsigned int*)(slab + o_hcnt);
  a.pwx = (float*)(slab + o_pwx); a.pwy = (float*)(slab + o_pwy); a.pwz = (float*)(slab + o_pwz); a.pkey = (unsigned long long*)(slab + o_pkey);
  a.ck = (unsigned long long*)(slab + o_ck); a.chead = (int*)(slab + o_chead); a.nxt = (int*)(slab + o_nxt); a.parent = (int*)(slab + o_par);
  a.csize = (int*)(slab + o_cs); a.cminkey = (unsigned long long*)(slab + o_cmk); a.cmind = (unsigned long long*)(slab + o_cmd);
  a.po = (double*)(slab + o_po); a.pof = (float*)(slab + o_pof);
  a.state = (FitShared*)(slab + o_state);
  a.clk = timing ? (long long*)(slab + o_clk) : nullptr;
  a.out_ell = (double*)(slab + o_ell); a.out_prob = (double*)(slab + o_prob); a.out_status = (int*)(slab + o_st); a.out_dbg = (double*)(slab + o_dbg);
  a.out_sym = (double*)(slab + o_sym);
  hipStream_t st = c->stream;
  const bool host_timing = std::getenv("ESL_FIT_HOST_TIMING") != nullptr;   // diagnostic: where the host call's time goes
  auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double ht0 = host_timing ? now_us() : 0;
  // stage the inputs (the previous call synchronised the stream: the staging block is free)
  {
    FitFrame fr;
    for (int i = 0; i < 7; ++i) fr.Twc[i] = Twc[i];
    for (int i = 0; i < 5; ++i) fr.intr[i] = intr[i];
    for (int i = 0; i < 4; ++i) fr.ground[i] = ground[i];
    fr.p = *p;
    std::memcpy(c->fit_in + o_fr, &fr, sizeof(fr));
    std::memcpy(c->fit_in + o_bb, bboxes, B * 32);
    int32_t* lab = (int32_t*)(c->fit_in + o_lab);
    for (int b = 0; b < n_boxes; ++b) lab[b] = labels ? labels[b] : -1;
    std::memcpy(c->fit_in + o_depth, depth, (size_t)width * height * 2);
  }
  const double ht1 = host_timing ? now_us() : 0;
  int rc = ESL_OK;
  auto fail = [&](hipError_t e, const char* what) { set_error(std::string(what) + ": " + hipGetErrorString(e)); rc = ESL_ERR_HIP; };
  hipError_t e;
  const int sym_open = p->symmetry_open;
  // The frame's work as a stream of asynchronous operations.  Nothing in it depends on per-frame VALUES (those sit in the
  // staged input block), only on the geometry (boxes, image size, capacity): it can be captured once and replayed.
  auto enqueue = [&](bool bracket) {
    if ((e = hipMemcpyAsync(slab, c->fit_in, in_bytes, hipMemcpyHostToDevice, st)) != hipSuccess) { fail(e, "upload"); return; }
    if ((e = hipMemsetAsync(slab + o_hk, 0xFF, B * H * 8, st)) != hipSuccess) { fail(e, "clear hash"); return; }
    if ((e = hipMemsetAsync(slab + o_hsx, 0, (o_pwx - o_hsx), st)) != hipSuccess) { fail(e, "clear sums"); return; }
    if ((e = hipMemsetAsync(slab + o_state, 0, off - o_state, st)) != hipSuccess) { fail(e, "clear state + outputs"); return; }
    {
      ProfScope ps(bracket ? c : nullptr, 5);
      // box scan + voxel hash and plane filter over (tile, box) grids; per-box clustering + PCA (clustering grid-wide for
      // big boxes); the 9 plane hypotheses of every box side by side; completion
      // latency-bound walks (a sample / slot is a chain of 2-3 dependent loads and a few atomics): go wide rather than deep
      const long tiles_scan = std::min<long>(512, std::max<long>(1, (cap + 511) / 512));    // ~2 samples per thread
      const long tiles_hash = std::min<long>(512, std::max<long>(1, H / 512));              // ~2 slots per thread
      hipLaunchKernelGGL(k_fit_scan, dim3((unsigned)tiles_scan, n_boxes), dim3(256), 0, st, a);
      hipLaunchKernelGGL(k_fit_plane, dim3((unsigned)tiles_hash, n_boxes), dim3(256), 0, st, a);
      if (a.wide) {
        const long tiles_pts = std::min<long>(64, std::max<long>(1, cap / 1024));
        hipLaunchKernelGGL(k_fit_cl_init, dim3((unsigned)tiles_hash, n_boxes), dim3(256), 0, st, a);
        hipLaunchKernelGGL(k_fit_cl_insert, dim3((unsigned)tiles_pts, n_boxes), dim3(256), 0, st, a);
        hipLaunchKernelGGL(k_fit_cl_union, dim3((unsigned)std::min<long>(1024, tiles_pts * 14), n_boxes), dim3(256), 0, st, a);
        hipLaunchKernelGGL(k_fit_cl_stats, dim3((unsigned)tiles_pts, n_boxes), dim3(256), 0, st, a);
      }
      hipLaunchKernelGGL(k_fit_pre, dim3(n_boxes), dim3(kFitThreads), 0, st, a);
      if (sym_open) hipLaunchKernelGGL(k_fit_sym, dim3(n_boxes * 9), dim3(kSymThreads), 0, st, a);
      hipLaunchKernelGGL(k_fit_post, dim3(n_boxes), dim3(kFitThreads), 0, st, a);
      if ((e = hipGetLastError()) != hipSuccess) { fail(e, "k_fit_*"); return; }
    }
    if ((e = hipMemcpyAsync(c->fit_out, slab + o_ell, out_bytes, hipMemcpyDeviceToHost, st)) != hipSuccess) fail(e, "download");
  };
  // hipGraph replay unless events are wanted around the kernels (profiling / stage timing) or it is switched off
  const bool use_graph = !c->prof_on && !timing && std::getenv("ESL_FIT_NO_GRAPH") == nullptr;
  if (use_graph) {
    FitGraphKey key{n_boxes, width, height, cap, sym_open, slab, c->fit_in, c->fit_out};
    hipGraphExec_t exec = fit_graph_find(c, key);
    if (!exec) {
      hipGraph_t graph = nullptr;
      if ((e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)) != hipSuccess) fail(e, "begin capture");
      if (!rc) enqueue(false);
      hipError_t e2 = hipStreamEndCapture(st, &graph);
      if (!rc && e2 != hipSuccess) fail(e2, "end capture");
      if (!rc && (e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0)) != hipSuccess) fail(e, "instantiate graph");
      if (graph) (void)hipGraphDestroy(graph);
      if (!rc) fit_graph_store(c, key, exec);
    }
    if (!rc && (e = hipGraphLaunch(exec, st)) != hipSuccess) fail(e, "launch graph");
  } else {
    enqueue(true);
  }
  const double ht2 = host_timing ? now_us() : 0;
  if ((e = hipStreamSynchronize(st)) != hipSuccess && !rc) fail(e, "sync");
  if (host_timing) fprintf(stderr, "[esl_fit host, us] stage %.0f  enqueue %.0f  wait %.0f\n", ht1 - ht0, ht2 - ht1, now_us() - ht2);
  const char* outs_data = c->fit_out;
  // the four output arrays are contiguous in the slab: one D2H copy into the pinned block, scattered on the host
  if (!rc) {
    std::memcpy(ellipsoids_out, outs_data, B * 80);
    std::memcpy(prob_out, outs_data + (o_prob - o_ell), B * 8);
    std::memcpy(status_out, outs_data + (o_st - o_ell), B * 4);
    if (debug_out) std::memcpy(debug_out, outs_data + (o_dbg - o_ell), B * 128);
    if (sym_out) {
      const double* so = (const double*)(outs_data + (o_sym - o_ell));
      for (size_t bx = 0; bx < B; ++bx) {
        esl_fit_symmetry& o = sym_out[bx];
        for (int k = 0; k < 4; ++k) { o.plane[k] = so[16 * bx + k]; o.plane2[k] = so[16 * bx + 4 + k]; }
        o.prob = so[16 * bx + 8];
        o.result = so[16 * bx + 10] > 0.5 ? 1 : 0;
        o.symmetry_type = o.result ? (int32_t)so[16 * bx + 9] : -1;
        for (int k = 0; k < 3; ++k) o.center[k] = so[16 * bx + 11 + k];
      }
    }
    if (timing) {   // diagnostic: mean stage durations over the boxes (wall_clock64 ticks at 100 MHz)
      static const char* names[8] = {"scan+voxel", "plane filter", "centre", "cluster", "pca", "sym grid", "symmetry", "extents"};
      const long long* clk = (const long long*)(outs_data + (o_clk - o_ell));
      double sum[8] = {0}; int cnt[8] = {0};
      for (size_t bx = 0; bx < B; ++bx)
        for (int k = 0; k < 8; ++k) {
          const long long t0 = clk[16 * bx + k], t1 = clk[16 * bx + k + 1];
          if (t0 > 0 && t1 >= t0) { sum[k] += (double)(t1 - t0) * 0.01; cnt[k]++; }
        }
      fprintf(stderr, "[esl_fit timing, us per box]");
      for (int k = 0; k < 8; ++k) fprintf(stderr, " %s=%.1f", names[k], cnt[k] ? sum[k] / cnt[k] : 0.0);
      {  // inside the cluster stage (LDS path): centre mark -> neighbour search + linking done
        double s9 = 0, m9 = 0; int c9 = 0;
        for (size_t bx = 0; bx < B; ++bx) {
          const long long t0 = clk[16 * bx + 3], t1 = clk[16 * bx + 9];
          if (t0 > 0 && t1 >= t0) { const double d = (double)(t1 - t0) * 0.01; s9 += d; m9 = std::max(m9, d); c9++; }
        }
        if (c9) fprintf(stderr, " (cluster: tables + search + linking %.1f, max %.1f)", s9 / c9, m9);
      }
      size_t slow = 0; double slow_t = 0;
      for (size_t bx = 0; bx < B; ++bx) {
        long long last = clk[16 * bx];
        for (int k = 1; k <= 8; ++k) if (clk[16 * bx + k] > last) last = clk[16 * bx + k];
        const double t = (double)(last - clk[16 * bx]) * 0.01;
        if (t > slow_t) { slow_t = t; slow = bx; }
      }
      {
        double md = 0; long long mr = 0, mn = 0;
        double me = 0, mb = 0;
        for (size_t bx = 0; bx < B; ++bx) { md = std::max(md, (double)clk[16 * bx + 10] * 0.01); mr = std::max(mr, clk[16 * bx + 11]); mn = std::max(mn, clk[16 * bx + 12]);
          me = std::max(me, (double)clk[16 * bx + 13] * 0.01); mb = std::max(mb, (double)clk[16 * bx + 14] * 0.01); }
        fprintf(stderr, "\n[esl_fit timing] symmetry kernel: slowest hypothesis %.1f us, max evaluation rounds %lld, max points %lld; linearisation evaluations (wave 0) %.1f us, bookkeeping %.1f us", md, mr, mn, me, mb);
      }
      fprintf(stderr, "\n[esl_fit timing] slowest box %zu: %.1f us:", slow, slow_t);
      for (int k = 0; k < 8; ++k) {
        const long long t0 = clk[16 * slow + k], t1 = clk[16 * slow + k + 1];
        fprintf(stderr, " %s=%.1f", names[k], (t0 > 0 && t1 >= t0) ? (double)(t1 - t0) * 0.01 : 0.0);
      }
      fprintf(stderr, "\n");
    }
  }
  return rc;
}

int esl_fit_frame_debug(esl_ctx* c, const uint16_t* depth, int32_t width, int32_t height, const double* bboxes,
                        const int32_t* labels, int32_t n_boxes, const double Twc[7], const double intr[5],
                        const double ground[4], const esl_fit_params* p, double* ellipsoids_out, double* prob_out,
                        int32_t* status_out, double* debug_out) {
  return fit_frame_impl(c, depth, width, height, bboxes, labels, n_boxes, Twc, intr, ground, p, ellipsoids_out, prob_out, status_out,
                        debug_out, nullptr);
}

int esl_fit_frame(esl_ctx* c, const uint16_t* depth, int32_t width, int32_t height, const double* bboxes, const int32_t* labels,
                  int32_t n_boxes, const double Twc[7], const double intr[5], const double ground[4], const esl_fit_params* p,
                  double* ellipsoids_out, double* prob_out, int32_t* status_out) {
  return fit_frame_impl(c, depth, width, height, bboxes, labels, n_boxes, Twc, intr, ground, p, ellipsoids_out, prob_out, status_out,
                        nullptr, nullptr);
}

int esl_fit_frame_ex(esl_ctx* c, const uint16_t* depth, int32_t width, int32_t height, const double* bboxes, const int32_t* labels,
                     int32_t n_boxes, const double Twc[7], const double intr[5], const double ground[4], const esl_fit_params* p,
                     double* ellipsoids_out, double* prob_out, int32_t* status_out, esl_fit_symmetry* symmetry_out,
                     double* debug_out) {
  return fit_frame_impl(c, depth, width, height, bboxes, labels, n_boxes, Twc, intr, ground, p, ellipsoids_out, prob_out, status_out,
                        debug_out, symmetry_out);
}

}  // extern "C"
