/* esl_oracle_plane.c — CPU restatement of the ground-plane extraction.  TEST INFRASTRUCTURE ONLY (see esl_oracle.h).
 *
 * Follows EllipsoidSLAM::PlaneExtractor::extractPlanes / extractGroundPlane (reference src/plane/PlaneExtractor.cpp:22-104,
 * 107-183).  The two PCL calls (IntegralImageNormalEstimation :54-61, OrganizedMultiPlaneSegmentation::segmentAndRefine
 * :68-83; PCL is not vendored and not installed) are restated by the definitions written at the top of
 * object-oriented-slam_amd/csrc/esl_plane.hip; PARITY WITH PCL ITSELF IS UNPINNED, the rest (point back-projection, size
 * filter, sign, wall filter, largest-plane choice) follows the reference line by line.  An independent numpy / scipy version
 * of the same definitions is oracle/np_plane.py. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "esl_oracle.h"

static int px_point(const uint16_t* depth, int w, int u, int v, const double intr[5], float p[3]) {
  const uint16_t d = depth[(size_t)v * w + u];
  const float z = (float)((double)d / intr[4]);                       /* PlaneExtractor.cpp:38-40 */
  p[2] = z;
  p[0] = (float)(((double)u - intr[2]) * (double)z / intr[0]);        /* :41 */
  p[1] = (float)(((double)v - intr[3]) * (double)z / intr[1]);        /* :42 */
  return d != 0;
}

static void jacobi3_smallest(const double C[9], double n[3]) {
  double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  memcpy(A, C, sizeof(A));
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

static int find_root(int* parent, int i) {
  while (parent[i] != i) { parent[i] = parent[parent[i]]; i = parent[i]; }
  return i;
}
static void unite(int* parent, int a, int b) {
  a = find_root(parent, a); b = find_root(parent, b);
  if (a == b) return;
  if (a < b) parent[b] = a; else parent[a] = b;
}

/* least-squares plane of a segment from its exact moments; returns PCL's surface curvature of the segment, |lambda_min| / trace(cov)
 * (OrganizedMultiPlaneSegmentation::segment: eigen_value / eig_sum of the 3 x 3 covariance; lambda_min as the Rayleigh quotient of the
 * unit normal) */
static double plane_of(const long long* m, int cnt, double pl[4]) {
  const double f1 = 4294967296.0, f = 1073741824.0, inv = 1.0 / (double)cnt;   /* fixed point: 2^-32 m, 2^-30 m^2 */
  const double cx = (double)m[0] / f1 * inv, cy = (double)m[1] / f1 * inv, cz = (double)m[2] / f1 * inv;
  const double C[9] = {(double)m[3] / f * inv - cx * cx, (double)m[4] / f * inv - cx * cy, (double)m[5] / f * inv - cx * cz,
                       (double)m[4] / f * inv - cx * cy, (double)m[6] / f * inv - cy * cy, (double)m[7] / f * inv - cy * cz,
                       (double)m[5] / f * inv - cx * cz, (double)m[7] / f * inv - cy * cz, (double)m[8] / f * inv - cz * cz};
  double n[3];
  jacobi3_smallest(C, n);
  const double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
  pl[0] = n[0] / nn; pl[1] = n[1] / nn; pl[2] = n[2] / nn;
  pl[3] = -(pl[0] * cx + pl[1] * cy + pl[2] * cz);
  if (pl[3] < 0) for (int k = 0; k < 4; ++k) pl[k] = -pl[k];          /* PlaneExtractor.cpp:95-96 */
  double lmin = 0;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) lmin += pl[i] * C[3 * i + j] * pl[j];
  const double tr = C[0] + C[4] + C[8];
  return tr != 0 ? fabs(lmin / tr) : 0.0;
}

static int plane_core(const uint16_t* depth, int32_t w, int32_t h, const double intr[5], const esl_plane_params* p,
                      double plane_out[4], int32_t* ok, int32_t* n_planes, int32_t* n_pixels, float* normals_out,
                      int32_t max_planes, double* planes_out, int32_t* sizes_out, int32_t* labels_out) {
  const size_t npx = (size_t)w * h;
  float* nrm = (float*)malloc(npx * 4 * sizeof(float));
  int* parent = (int*)malloc(npx * sizeof(int));
  int* cnt = (int*)calloc(npx, sizeof(int));
  long long* mom = (long long*)calloc(npx * 9, sizeof(long long));
  if (!nrm || !parent || !cnt || !mom) { free(nrm); free(parent); free(cnt); free(mom); return -1; }
  const int R = p->normal_smoothing / 2;
  const double fac = p->max_depth_change_factor, scale = intr[4];
  /* normals: average 3-D gradient over the window */
  for (int v = 0; v < h; ++v)
    for (int u = 0; u < w; ++u) {
      float* o = nrm + 4 * ((size_t)v * w + u);
      o[0] = NAN; o[1] = o[2] = o[3] = 0;
      if (u - R - 1 < 0 || v - R - 1 < 0 || u + R + 1 >= w || v + R + 1 >= h) continue;
      int good = 1;
      for (int y = v - R - 1; y <= v + R + 1 && good; ++y)
        for (int x = u - R - 1; x <= u + R + 1; ++x) {
          const uint16_t d = depth[(size_t)y * w + x];
          if (d == 0) { good = 0; break; }
          const double z = (double)d / scale;
          if (x + 1 <= u + R + 1 && fabs((double)depth[(size_t)y * w + x + 1] / scale - z) > fac * z) { good = 0; break; }
          if (y + 1 <= v + R + 1 && fabs((double)depth[(size_t)(y + 1) * w + x] / scale - z) > fac * z) { good = 0; break; }
        }
      if (!good) continue;
      double dh[3] = {0, 0, 0}, dv[3] = {0, 0, 0};
      for (int y = v - R; y <= v + R; ++y)
        for (int x = u - R; x <= u + R; ++x) {
          float pr[3], pl[3], pd[3], pu[3];
          px_point(depth, w, x + 1, y, intr, pr); px_point(depth, w, x - 1, y, intr, pl);
          px_point(depth, w, x, y + 1, intr, pd); px_point(depth, w, x, y - 1, intr, pu);
          for (int k = 0; k < 3; ++k) { dh[k] += (double)pr[k] - (double)pl[k]; dv[k] += (double)pd[k] - (double)pu[k]; }
        }
      const double n[3] = {dv[1] * dh[2] - dv[2] * dh[1], dv[2] * dh[0] - dv[0] * dh[2], dv[0] * dh[1] - dv[1] * dh[0]};
      const double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
      if (!(nn > 0)) continue;
      float pt[3];
      px_point(depth, w, u, v, intr, pt);
      double s = 1.0 / nn;
      if ((n[0] * pt[0] + n[1] * pt[1] + n[2] * pt[2]) * s > 0) s = -s;
      const float nx = (float)(n[0] * s), ny = (float)(n[1] * s), nz = (float)(n[2] * s);
      o[0] = nx; o[1] = ny; o[2] = nz;
      o[3] = -(nx * pt[0] + ny * pt[1] + nz * pt[2]);
    }
  if (normals_out) memcpy(normals_out, nrm, npx * 4 * sizeof(float));
  /* segments: connected components under the plane-coefficient comparison */
  const float cos_ang = (float)cos(p->angle_threshold_deg * 0.017453), dist_th = (float)p->distance_threshold;   /* :75-76 */
  for (size_t i = 0; i < npx; ++i) parent[i] = (int)i;
  for (int v = 0; v < h; ++v)
    for (int u = 0; u < w; ++u) {
      const int i = v * w + u;
      const float* a = nrm + 4 * (size_t)i;
      if (a[0] != a[0]) continue;
      const int nb[2] = {u + 1 < w ? i + 1 : -1, v + 1 < h ? i + w : -1};
      for (int k = 0; k < 2; ++k) {
        if (nb[k] < 0) continue;
        const float* b = nrm + 4 * (size_t)nb[k];
        if (b[0] != b[0]) continue;
        const float dot = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
        if (fabsf(a[3] - b[3]) < dist_th && dot > cos_ang) unite(parent, i, nb[k]);
      }
    }
  for (int v = 0; v < h; ++v)
    for (int u = 0; u < w; ++u) {
      const int i = v * w + u;
      if (nrm[4 * (size_t)i] != nrm[4 * (size_t)i]) continue;
      const int r = find_root(parent, i);
      float pt[3];
      px_point(depth, w, u, v, intr, pt);
      const double x = pt[0], y = pt[1], z = pt[2], f1 = 4294967296.0, f = 1073741824.0;
      const double v9[9] = {x * f1, y * f1, z * f1, x * x * f, x * y * f, x * z * f, y * y * f, y * z * f, z * z * f};
      ++cnt[r];
      for (int k = 0; k < 9; ++k) mom[(size_t)r * 9 + k] += llrint(v9[k]);
    }
  /* PCL's models (OrganizedMultiPlaneSegmentation::segment, PCL 1.8 -- restated from its published source, UNPINNED):
   * segments of MORE than setMinInliers(100) pixels (:74; the test is a strict >) whose surface curvature is below maximum_curvature_
   * (0.001, PCL's default: the reference does not set it), in raster order of each segment's first pixel (= its root).  Anything else is
   * an "other" label that the refinement pass lets neighbouring models absorb.  Without the refinement pass a model below
   * Plane.MinSize can never become a plane, so the list then also starts at MinSize. */
  const int refine = p->refine != 0;
  const int extra_min = refine ? 0 : p->min_size;
  int* model_of_root = (int*)malloc(npx * sizeof(int));
  if (!model_of_root) { free(nrm); free(parent); free(cnt); free(mom); return -1; }
  int n_models = 0;
  for (size_t r = 0; r < npx; ++r) {
    model_of_root[r] = -1;
    if (!(cnt[r] > p->min_inliers && cnt[r] >= extra_min)) continue;
    double pl4[4];
    const double curv = plane_of(mom + r * 9, cnt[r], pl4);
    if (p->max_curvature > 0 && !(curv < p->max_curvature)) continue;
    model_of_root[r] = n_models++;
  }
  double* mpl = (double*)malloc((size_t)(n_models > 0 ? n_models : 1) * 4 * sizeof(double));
  int* msize = (int*)malloc((size_t)(n_models > 0 ? n_models : 1) * sizeof(int));
  int* lab = (int*)malloc(npx * sizeof(int));     /* model index; -1: no depth (never labelled); -2: any other pixel */
  if (!mpl || !msize || !lab) { free(model_of_root); free(mpl); free(msize); free(lab); free(nrm); free(parent); free(cnt); free(mom); return -1; }
  for (size_t r = 0; r < npx; ++r) {
    const int m = model_of_root[r];
    if (m < 0) continue;
    (void)plane_of(mom + r * 9, cnt[r], mpl + (size_t)m * 4);
    msize[m] = cnt[r];
  }
  for (size_t i = 0; i < npx; ++i) {
    if (depth[i] == 0) lab[i] = -1;
    else if (nrm[4 * i] != nrm[4 * i]) lab[i] = -2;
    else { const int m = model_of_root[find_root(parent, (int)i)]; lab[i] = m >= 0 ? m : -2; }
  }
  if (refine && n_models > 0 && w > 1 && h > 1) {
    /* OrganizedMultiPlaneSegmentation::refine as PCL 1.8 writes it (segmentAndRefine, PlaneExtractor.cpp:82; PCL is not vendored:
     * restated from its published source, UNPINNED): two raster passes in which a pixel of a model's label pulls its right / lower
     * (second pass: left / upper) neighbour into the model when that neighbour belongs to NO model and lies within the refinement
     * comparator's distance of the model's plane (PlaneRefinementComparator: |n.p + d| < 0.02 m, not depth dependent).  Labels move
     * as the scan goes, so a model grows along the scan direction through a whole run of such pixels.  The models' COEFFICIENTS are
     * not re-estimated (PCL does not): what changes is their inlier lists -- the size the reference filters by (:87) and ranks the
     * ground-plane candidates by (:160) and the points GetPoints() returns.  Two quirks kept: a pixel whose right (left) neighbour
     * has no label skips its lower (upper) neighbour too (the `continue`), and the second pass reads "the pixel to the left" of
     * column 0, i.e. the last pixel of the row above. */
    const double thr = p->refine_distance;
#define ESL_PL_TRY(cur_, nxt_, u_, v_) do { \
      const int cl_ = lab[cur_]; \
      if (cl_ >= 0 && lab[nxt_] == -2) { \
        float q_[3]; px_point(depth, w, (u_), (v_), intr, q_); \
        const double* pl_ = mpl + (size_t)cl_ * 4; \
        if (fabs(pl_[0] * (double)q_[0] + pl_[1] * (double)q_[1] + pl_[2] * (double)q_[2] + pl_[3]) < thr) { lab[nxt_] = cl_; ++msize[cl_]; } \
      } } while (0)
    for (int v = 0; v < h - 1; ++v)
      for (int u = 0; u < w - 1; ++u) {
        const size_t i = (size_t)v * w + u;
        if (lab[i] == -1 || lab[i + 1] == -1) continue;
        ESL_PL_TRY(i, i + 1, u + 1, v);
        if (lab[i + w] == -1) continue;
        ESL_PL_TRY(i, i + w, u, v + 1);
      }
    for (int v = h - 1; v >= 1; --v)
      for (int u = w - 1; u >= 0; --u) {
        const size_t i = (size_t)v * w + u;
        const size_t l = i - 1;                                   /* u == 0: the last pixel of row v - 1 (as PCL indexes it) */
        if (lab[i] == -1 || lab[l] == -1) continue;
        ESL_PL_TRY(i, l, (int)(l % (size_t)w), (int)(l / (size_t)w));
        if (lab[i - w] == -1) continue;
        ESL_PL_TRY(i, i - w, u, v - 1);
      }
#undef ESL_PL_TRY
  }
  /* planes of >= Plane.MinSize inliers (:87), wall filter, the largest one is the ground (:139-162; ties: the earlier model) */
  int best = -1, best_cnt = 0, planes = 0;
  int* plane_idx = (int*)malloc((size_t)(n_models > 0 ? n_models : 1) * sizeof(int));
  for (int m = 0; m < n_models; ++m) {
    plane_idx[m] = -1;
    if (msize[m] < p->min_size) continue;
    const double* pl = mpl + (size_t)m * 4;
    if (planes < max_planes && planes_out) { memcpy(planes_out + (size_t)planes * 4, pl, 4 * sizeof(double)); sizes_out[planes] = msize[m]; }
    plane_idx[m] = planes++;
    const double th = acos(pl[1] / sqrt(pl[0] * pl[0] + pl[1] * pl[1] + pl[2] * pl[2]));
    if (th > M_PI / 4 && th < 3 * M_PI / 4) continue;
    if (msize[m] > best_cnt) { best_cnt = msize[m]; best = m; }
  }
  if (labels_out)
    for (size_t i = 0; i < npx; ++i) labels_out[i] = lab[i] >= 0 ? plane_idx[lab[i]] : -1;
  if (ok) *ok = 0;
  if (plane_out) for (int k = 0; k < 4; ++k) plane_out[k] = 0;
  if (n_planes) *n_planes = planes;
  if (n_pixels) *n_pixels = 0;
  if (best >= 0) {
    if (plane_out) memcpy(plane_out, mpl + (size_t)best * 4, 4 * sizeof(double));   /* d >= 0: camera centre on the positive side (:165-167) */
    if (ok) *ok = 1;
    if (n_pixels) *n_pixels = best_cnt;
  }
  free(plane_idx); free(model_of_root); free(mpl); free(msize); free(lab);
  free(nrm); free(parent); free(cnt); free(mom);
  return 0;
}

int esl_oracle_extract_ground_plane(const uint16_t* depth, int32_t w, int32_t h, const double intr[5], const esl_plane_params* p,
                                    double plane_out[4], int32_t* ok, int32_t* n_planes, int32_t* n_pixels, float* normals_out) {
  return plane_core(depth, w, h, intr, p, plane_out, ok, n_planes, n_pixels, normals_out, 0, NULL, NULL, NULL);
}

int esl_oracle_extract_planes(const uint16_t* depth, int32_t w, int32_t h, const double intr[5], const esl_plane_params* p,
                              int32_t max_planes, double* planes_out, int32_t* sizes_out, int32_t* n_planes, int32_t* labels_out) {
  return plane_core(depth, w, h, intr, p, NULL, NULL, n_planes, NULL, NULL, max_planes, planes_out, sizes_out, labels_out);
}
