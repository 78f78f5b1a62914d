/*
 * esl_oracle_fit.c — CPU restatement of the single-frame ellipsoid fit
 * (EllipsoidExtractor::EstimateLocalEllipsoid, reference src/pca/EllipsoidExtractor.cpp:292-493 and
 * src/symmetry/{PointCloudFilter,Symmetry,SymmetrySolver}.cpp, include/core/Plane.h).
 * TEST INFRASTRUCTURE ONLY (see esl_oracle.h).
 *
 * The reference delegates four steps to PCL ("1.7+", unpinned, not vendored).  They are restated as
 * deterministic, ORDER-INDEPENDENT equivalents (SURVEY.md §7 "PCL semantics are unpinned"):
 *   VoxelGrid            float32 points, voxel = floorf(p * (1.0f/leaf)) per axis, one output point per
 *                        occupied voxel = centroid of its members (accumulated exactly in 2^-30 m fixed point)
 *   EuclideanCluster     connected components of "distance <= tolerance" on the float32 world points,
 *                        clusters >= MinClusterSize ordered by size (descending; ties: smaller voxel key first)
 *   compute3DCentroid /  double accumulation over the float32 points
 *   computeCovarianceMatrixNormalized
 *   KdTreeFLANN 1-NN     exact nearest neighbour (brute force) to the float32 copies of the object points
 * Eigen pieces: SelfAdjointEigenSolver -> cyclic Jacobi (ascending; eigenvector sign: largest component > 0),
 * Quaterniond(Matrix3d), AngleAxisd::toRotationMatrix restated from Eigen 3.3.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "esl_oracle.h"

typedef struct { double x, y, z; } v3;
typedef struct { float x, y, z; } f3;
typedef struct { uint64_t key; int64_t sx, sy, sz; uint32_t n; } vox;

#define KOFF (1 << 20)
static const double FIX = 1073741824.0; /* 2^30 */

static uint64_t vkey(float x, float y, float z, float inv) {
  int ix = (int)floorf(x * inv), iy = (int)floorf(y * inv), iz = (int)floorf(z * inv);
  return ((uint64_t)(iz + KOFF) << 42) | ((uint64_t)(iy + KOFF) << 21) | (uint64_t)(ix + KOFF);
}
static int cmp_vox(const void* a, const void* b) {
  uint64_t ka = ((const vox*)a)->key, kb = ((const vox*)b)->key;
  return ka < kb ? -1 : ka > kb;
}
/* VoxelGrid equivalent: in = float points, out = float centroids (+ key), canonical ascending-key order */
static int voxel_grid(const f3* in, int n, float leaf, f3* out, uint64_t* out_key) {
  if (n == 0) return 0;
  const float inv = 1.0f / leaf;
  vox* v = (vox*)malloc(sizeof(vox) * (size_t)n);
  for (int i = 0; i < n; ++i) {
    v[i].key = vkey(in[i].x, in[i].y, in[i].z, inv);
    v[i].sx = llrint((double)in[i].x * FIX); v[i].sy = llrint((double)in[i].y * FIX); v[i].sz = llrint((double)in[i].z * FIX);
    v[i].n = 1;
  }
  qsort(v, (size_t)n, sizeof(vox), cmp_vox);
  int m = 0;
  for (int i = 0; i < n;) {
    int j = i;
    int64_t sx = 0, sy = 0, sz = 0;
    while (j < n && v[j].key == v[i].key) { sx += v[j].sx; sy += v[j].sy; sz += v[j].sz; ++j; }
    double c = (double)(j - i);
    out[m].x = (float)((double)sx / c / FIX); out[m].y = (float)((double)sy / c / FIX); out[m].z = (float)((double)sz / c / FIX);
    if (out_key) out_key[m] = v[i].key;
    ++m;
    i = j;
  }
  free(v);
  return m;
}

/* ---- small linear algebra ----------------------------------------------------------------------------*/
static void quat_to_R(const double q[4] /* x y z w */, double R[9]) { /* Eigen toRotationMatrix */
  double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
  double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0], tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
static void R_to_quat(const double R[9], double q[4]) { /* Eigen Quaterniond(Matrix3d) */
  double t = R[0] + R[4] + R[8];
  if (t > 0) {
    t = sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
    q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 3 + i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
    q[i] = 0.5 * t; t = 0.5 / t;
    q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t;
    q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
  }
}
static void jacobi3(const double Ain[9], double w[3], double V[9]) {
  double A[9];
  memcpy(A, Ain, sizeof(A));
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0);
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    if (off < 1e-300) break;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double apq = A[p * 3 + q];
        if (apq == 0.0) continue;
        double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
        double c = 1 / sqrt(t * t + 1), s = t * c;
        for (int k = 0; k < 3; ++k) { double a = A[k * 3 + p], b = A[k * 3 + q]; A[k * 3 + p] = c * a - s * b; A[k * 3 + q] = s * a + c * b; }
        for (int k = 0; k < 3; ++k) { double a = A[p * 3 + k], b = A[q * 3 + k]; A[p * 3 + k] = c * a - s * b; A[q * 3 + k] = s * a + c * b; }
        for (int k = 0; k < 3; ++k) { double a = V[k * 3 + p], b = V[k * 3 + q]; V[k * 3 + p] = c * a - s * b; V[k * 3 + q] = s * a + c * b; }
      }
  }
  for (int i = 0; i < 3; ++i) w[i] = A[i * 4];
  for (int i = 0; i < 3; ++i) {
    int m = i;
    for (int j = i + 1; j < 3; ++j) if (w[j] < w[m]) m = j;
    if (m != i) {
      double t = w[i]; w[i] = w[m]; w[m] = t;
      for (int k = 0; k < 3; ++k) { double u = V[k * 3 + i]; V[k * 3 + i] = V[k * 3 + m]; V[k * 3 + m] = u; }
    }
  }
  for (int j = 0; j < 3; ++j) {
    int m = 0;
    for (int k = 1; k < 3; ++k) if (fabs(V[k * 3 + j]) > fabs(V[m * 3 + j])) m = k;
    if (V[m * 3 + j] < 0) for (int k = 0; k < 3; ++k) V[k * 3 + j] = -V[k * 3 + j];
  }
}
static void cross3(const double a[3], const double b[3], double c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}

/* ---- union-find for the Euclidean clustering --------------------------------------------------------*/
static int uf_find(int* p, int i) { while (p[i] != i) { p[i] = p[p[i]]; i = p[i]; } return i; }
static void uf_union(int* p, int a, int b) { a = uf_find(p, a); b = uf_find(p, b); if (a == b) return; if (a < b) p[b] = a; else p[a] = b; }

typedef struct { int64_t cx, cy, cz; int idx; } cellent;
static int cmp_cell(const void* a, const void* b) {
  const cellent *A = (const cellent*)a, *B = (const cellent*)b;
  if (A->cz != B->cz) return A->cz < B->cz ? -1 : 1;
  if (A->cy != B->cy) return A->cy < B->cy ? -1 : 1;
  if (A->cx != B->cx) return A->cx < B->cx ? -1 : 1;
  return A->idx - B->idx;
}

/* ---- g2o::plane pieces (include/core/Plane.h:46-129, src/core/Plane.cpp:97-144) -----------------------*/
typedef struct { double p[4]; double dual; } plane_t;
static void plane_from_dis_angle(plane_t* pl, double dis, double angle) { /* fromDisAngleTrans(dis, angle, 0) */
  pl->p[0] = sin(angle); pl->p[1] = -cos(angle); pl->p[2] = 0; pl->p[3] = -dis; pl->dual = 0;
}
static void plane_rotation(const double n[3], double R[9]) { /* plane::rotation: Rz(azimuth) * Ry(-elevation) */
  double az = atan2(n[1], n[0]), el = atan2(n[2], sqrt(n[0] * n[0] + n[1] * n[1]));
  double ca = cos(az), sa = sin(az), cb = cos(-el), sb = sin(-el);
  /* Rz = [ca -sa 0; sa ca 0; 0 0 1], Ry(b) = [cb 0 sb; 0 1 0; -sb 0 cb] */
  R[0] = ca * cb; R[1] = -sa; R[2] = ca * sb;
  R[3] = sa * cb; R[4] = ca;  R[5] = sa * sb;
  R[6] = -sb;     R[7] = 0;   R[8] = cb;
}
static void plane_oplus3(plane_t* pl, double az, double el, double dd) { /* plane::oplus / oplus_dual core */
  double s = sin(el), c = cos(el);
  double n[3] = {c * cos(az), c * sin(az), s};
  double R[9];
  plane_rotation(pl->p, R);
  double d = -pl->p[3] + dd;
  for (int i = 0; i < 3; ++i) pl->p[i] = R[i * 3] * n[0] + R[i * 3 + 1] * n[1] + R[i * 3 + 2] * n[2];
  pl->p[3] = -d;
  double nn = sqrt(pl->p[0] * pl->p[0] + pl->p[1] * pl->p[1] + pl->p[2] * pl->p[2]);
  for (int i = 0; i < 4; ++i) pl->p[i] *= (1. / nn);
}
static void plane_update(plane_t* pl, const double* u, int dual) {
  if (!dual) plane_oplus3(pl, u[0], 0.0, u[1]);                 /* VertexPlane::oplusImpl: (yaw, 0, dis) */
  else { plane_oplus3(pl, u[0], 0.0, u[1]); pl->dual += u[2]; } /* oplus_dual */
}
static void plane_another(const plane_t* pl, double out[4]) { /* GenerateAnotherPlaneVec */
  double az = atan2(pl->p[1], pl->p[0]);
  out[0] = sin(az); out[1] = -cos(az); out[2] = 0; out[3] = -pl->dual;
}
static v3 mirror_point(v3 p, const double pl[4]) { /* SymmetrySolver::GetSymmetryPointOfPlane (:193-214) */
  double nn = sqrt(pl[0] * pl[0] + pl[1] * pl[1] + pl[2] * pl[2]);
  double n[3] = {pl[0] / nn, pl[1] / nn, pl[2] / nn};
  double sv = pl[0] * p.x + pl[1] * p.y + pl[2] * p.z + pl[3];
  double dis = fabs(sv) / nn;
  double symbol = sv > 0 ? -1 : 1;
  v3 r = {p.x + 2 * symbol * dis * n[0], p.y + 2 * symbol * dis * n[1], p.z + 2 * symbol * dis * n[2]};
  return r;
}

typedef struct {
  const uint16_t* depth; int w, h;
  double bbox[4];
  double K[4], scale, sigma;
  double P[12];      /* K [R|t] of campose_cw (object frame) */
  double camc[3];    /* camera centre in the object frame */
  const v3* pts; int n;     /* object points (double) */
  const f3* ptsf;           /* their float32 copies (kd-tree input) */
} symctx;

/* SymmetrySolver::GetPointCloudProb (:49-138) on the mirrored cloud; returns the edge error = -aver_ln_P */
static double sym_cost(const symctx* c, const v3* sym, int ns) {
  double ln_total = 0;
  int invalid = 0;
  for (int i = 0; i < ns; ++i) {
    v3 p = sym[i];
    double uh[3];
    for (int r = 0; r < 3; ++r) uh[r] = c->P[r * 4] * p.x + c->P[r * 4 + 1] * p.y + c->P[r * 4 + 2] * p.z + c->P[r * 4 + 3];
    double u = uh[0] / uh[2], v = uh[1] / uh[2];
    int use_nn = 1;
    int finite_pt = isfinite(p.x) && isfinite(p.y) && isfinite(p.z);
    if (isfinite(u) && isfinite(v) && fabs(u) < 1e9 && fabs(v) < 1e9) {
      int x = (int)u, y = (int)v;
      if ((int)c->bbox[0] < x && x < (int)c->bbox[2] && (int)c->bbox[1] < y && y < (int)c->bbox[3] && x >= 0 && y >= 0 && x < c->w && y < c->h) {
        uint16_t d = c->depth[(size_t)y * c->w + x];
        /* Symmetry::getProjDepthMat (:141-161): radial depth, truncated to ushort */
        double realz = (double)d * sqrt((x - c->K[2]) * (x - c->K[2]) + c->K[0] * c->K[0] + (y - c->K[3]) * (y - c->K[3])) / c->K[0];
        uint16_t dp = (uint16_t)realz;
        if (dp == 0) use_nn = 0;
        else {
          double depth = dp / c->scale;
          double dx = c->camc[0] - p.x, dy = c->camc[1] - p.y, dz = c->camc[2] - p.z;
          if (sqrt(dx * dx + dy * dy + dz * dz) > depth) use_nn = 0;
        }
      }
    }
    double dis = 0;
    if (use_nn) {
      if (!finite_pt) { invalid++; }
      else {
        double best = 1e300;
        for (int j = 0; j < c->n; ++j) {
          double dx = p.x - (double)c->ptsf[j].x, dy = p.y - (double)c->ptsf[j].y, dz = p.z - (double)c->ptsf[j].z;
          double d2 = dx * dx + dy * dy + dz * dz;
          if (d2 < best) best = d2;
        }
        dis = sqrt(best);
      }
    }
    double si = 1.0 / c->sigma;
    ln_total += -0.5 * si * si * dis * dis;
  }
  int valid = ns - invalid;
  double aver = valid > 0 ? ln_total / (double)valid : -INFINITY;
  return -aver;
}

static double sym_error(const symctx* c, const plane_t* pl, int dual, v3* buf) {
  int ns = 0;
  for (int i = 0; i < c->n; ++i) buf[ns++] = mirror_point(c->pts[i], pl->p);
  if (dual) {
    double p2[4];
    plane_another(pl, p2);
    for (int i = 0; i < c->n; ++i) buf[ns++] = mirror_point(c->pts[i], p2);
  }
  return sym_cost(c, buf, ns);
}

/* one-edge LM (graph.optimize(5), SymmetrySolver.cpp:263/319) with g2o's numeric Jacobian and control flow;
 * returns the edge's _error as left by the LAST computeError call (which may belong to a rejected trial) */
static double sym_optimize(const symctx* c, plane_t* pl, int dual, int iters, v3* buf) {
  const int dim = dual ? 3 : 2;
  const double delta = 1e-9;
  double lambda = 0, ni = 2;
  int nbad = 0;
  double e_last = sym_error(c, pl, dual, buf);
  if (iters <= 0) return e_last;
  for (int it = 0; it < iters; ++it) {
    double e0 = sym_error(c, pl, dual, buf);
    e_last = e0;
    double cur = e0 * e0, ini = cur;
    double J[3], H[9], b[3];
    for (int d = 0; d < dim; ++d) {
      double u[3] = {0, 0, 0};
      plane_t pp = *pl, pm = *pl;
      u[d] = delta; plane_update(&pp, u, dual);
      u[d] = -delta; plane_update(&pm, u, dual);
      J[d] = (1.0 / (2 * delta)) * (sym_error(c, &pp, dual, buf) - sym_error(c, &pm, dual, buf));
    }
    for (int a = 0; a < dim; ++a) { b[a] = J[a] * (-(1.0 * e0)); for (int k = 0; k < dim; ++k) H[a * dim + k] = J[a] * 1.0 * J[k]; }
    if (it == 0) {
      double md = 0;
      for (int a = 0; a < dim; ++a) if (fabs(H[a * dim + a]) > md) md = fabs(H[a * dim + a]);
      lambda = 1e-5 * md; ni = 2; nbad = 0;
    }
    double rho = 0;
    int q = 0;
    do {
      plane_t bak = *pl;
      double M[9], x[3] = {0, 0, 0};
      for (int a = 0; a < dim * dim; ++a) M[a] = H[a];
      for (int a = 0; a < dim; ++a) M[a * dim + a] += lambda;
      int ok = esl_oracle_ldlt_solve(M, dim, b, x);
      plane_update(pl, x, dual);
      double et = sym_error(c, pl, dual, buf);
      e_last = et;
      double tmp = ok ? et * et : 1.7976931348623157e308;
      double scale = 0;
      for (int a = 0; a < dim; ++a) scale += x[a] * (lambda * x[a] + b[a]);
      rho = (cur - tmp) / (scale + 1e-3);
      if (rho > 0 && isfinite(tmp)) {
        double alpha = 1. - pow((2 * rho - 1), 3);
        alpha = alpha < 2. / 3. ? alpha : 2. / 3.;
        lambda *= (1. / 3. > alpha ? 1. / 3. : alpha);
        ni = 2; cur = tmp;
      } else { lambda *= ni; ni *= 2; *pl = bak; }
      q++;
    } while (rho < 0 && q < 10);
    if (q == 10 || rho == 0) break;
    if ((ini - cur) * 1e3 < ini) nbad++; else nbad = 0;
    if (nbad >= 3) break;
  }
  return e_last;
}

static int symmetry_type(int label) { /* EllipsoidExtractor::LoadSymmetryPrior (:52-79) */
  switch (label) { case 58: return 0; case 59: case 62: case 57: case 66: case 63: case 64: case 41: return 1; case 28: return 2; default: return -1; }
}

static int fit_one(const uint16_t* depth, int w, int h, const double bbox[4], int label, const double Twc[7],
                   const double intr[5], const double ground[4], const esl_fit_params* P, double ell[10], double* prob,
                   double dbg[16], double sym[16]) {
  for (int i = 0; i < 10; ++i) ell[i] = 0;
  if (sym) for (int i = 0; i < 16; ++i) sym[i] = 0;
  *prob = 0;
  const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3], scale = intr[4];
  /* 1. getPointCloudInRect (PointCloudFilter.cpp:21-56) */
  int x1 = (int)bbox[0], y1 = (int)bbox[1], x2 = (int)bbox[2], y2 = (int)bbox[3];
  int cap = ((x2 - x1) / P->stride + 2) * ((y2 - y1) / P->stride + 2);
  if (cap < 16) cap = 16;
  f3* pf = (f3*)malloc(sizeof(f3) * (size_t)cap);
  int n0 = 0;
  for (int y = y1; y < y2; y += P->stride)
    for (int x = x1; x < x2; x += P->stride) {
      if (x < 0 || y < 0 || x >= w || y >= h) continue;
      uint16_t d = depth[(size_t)y * w + x];
      double z = d / scale;
      if (z <= P->depth_min || z > P->depth_max) continue;
      pf[n0].x = (float)((x - cx) * z / fx); pf[n0].y = (float)((y - cy) * z / fy); pf[n0].z = (float)z; /* QuadricPointCloudToPcl: float */
      n0++;
    }
  if (dbg) dbg[0] = n0;
  /* 2. DownSamplePointCloudOnly(0.01) (:98) */
  f3* v1 = (f3*)malloc(sizeof(f3) * (size_t)(n0 + 1));
  uint64_t* k1 = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(n0 + 1));
  int n1 = voxel_grid(pf, n0, (float)P->voxel_leaf, v1, k1);
  free(pf);
  if (dbg) dbg[1] = n1;
  /* 3. transformPointCloud to world (:102), campose_wc.fromVector(pose): no normalisation */
  double Rwc[9];
  quat_to_R(&Twc[3], Rwc);
  double gn = sqrt(ground[0] * ground[0] + ground[1] * ground[1] + ground[2] * ground[2]);
  /* 4. ApplySupportingPlaneFilter (:562-578) */
  f3* pw = (f3*)malloc(sizeof(f3) * (size_t)(n1 + 1));
  uint64_t* kw = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(n1 + 1));
  int n2 = 0;
  for (int i = 0; i < n1; ++i) {
    double p[3] = {v1[i].x, v1[i].y, v1[i].z}, q[3];
    for (int r = 0; r < 3; ++r) q[r] = Rwc[r * 3] * p[0] + Rwc[r * 3 + 1] * p[1] + Rwc[r * 3 + 2] * p[2] + Twc[r];
    double dis = (ground[0] * q[0] + ground[1] * q[1] + ground[2] * q[2] + ground[3]) / gn;
    if (dis > P->plane_dist) { pw[n2].x = (float)q[0]; pw[n2].y = (float)q[1]; pw[n2].z = (float)q[2]; kw[n2] = k1[i]; n2++; }
  }
  free(v1); free(k1);
  if (dbg) dbg[2] = n2;
  if (n2 < 1) { free(pw); free(kw); return 4; }
  /* 5. GetCenter (:583-643) */
  double center[3];
  {
    int x = (int)((bbox[0] + bbox[2]) / 2.0), y = (int)((bbox[1] + bbox[3]) / 2.0);
    const int point_num = 10;
    int x_delta = (int)(fabs(bbox[0] - bbox[2]) / 4.0 / point_num), y_delta = (int)(fabs(bbox[1] - bbox[3]) / 4.0 / point_num);
    double sx = 0, sy = 0, sz = 0; int cnt = 0;
    for (int xi = -point_num / 2; xi < point_num / 2; ++xi)
      for (int yi = -point_num / 2; yi < point_num / 2; ++yi) {
        int x_ = x + xi * x_delta, y_ = y + yi * y_delta;
        if (x_ < 0 || y_ < 0 || x_ >= w || y_ >= h) continue;
        uint16_t d = depth[(size_t)y_ * w + x_];
        float pz = (float)(d / scale);
        if (pz <= 0.1 || pz > P->depth_max) continue;
        float px = (float)((x_ - cx) * pz / fx), py = (float)((y_ - cy) * pz / fy);
        sx += px; sy += py; sz += pz; cnt++;
      }
    if (cnt < 2) { free(pw); free(kw); return 1; }
    double c[3] = {sx / cnt, sy / cnt, sz / cnt};
    for (int r = 0; r < 3; ++r) center[r] = Rwc[r * 3] * c[0] + Rwc[r * 3 + 1] * c[1] + Rwc[r * 3 + 2] * c[2] + Twc[r];
  }
  /* 6. ApplyEuclideanFilter (:646-737) */
  int* parent = (int*)malloc(sizeof(int) * (size_t)n2);
  for (int i = 0; i < n2; ++i) parent[i] = i;
  {
    const double tol = P->cluster_tolerance, tol2 = tol * tol;
    cellent* ce = (cellent*)malloc(sizeof(cellent) * (size_t)n2);
    for (int i = 0; i < n2; ++i) {
      ce[i].cx = (int64_t)floor((double)pw[i].x / tol); ce[i].cy = (int64_t)floor((double)pw[i].y / tol);
      ce[i].cz = (int64_t)floor((double)pw[i].z / tol); ce[i].idx = i;
    }
    qsort(ce, (size_t)n2, sizeof(cellent), cmp_cell);
    /* for each point scan the 27 neighbouring cells by binary search on the sorted cell list */
    for (int a = 0; a < n2; ++a) {
      int i = ce[a].idx;
      for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
        cellent key = {ce[a].cx + dx, ce[a].cy + dy, ce[a].cz + dz, -1};
        int lo = 0, hi = n2;
        while (lo < hi) { int mid = (lo + hi) / 2; if (cmp_cell(&ce[mid], &key) < 0) lo = mid + 1; else hi = mid; }
        for (int b = lo; b < n2 && ce[b].cx == key.cx && ce[b].cy == key.cy && ce[b].cz == key.cz; ++b) {
          int j = ce[b].idx;
          if (j <= i) continue;
          double ddx = (double)pw[i].x - pw[j].x, ddy = (double)pw[i].y - pw[j].y, ddz = (double)pw[i].z - pw[j].z;
          if (ddx * ddx + ddy * ddy + ddz * ddz <= tol2) uf_union(parent, i, j);
        }
      }
    }
    free(ce);
  }
  int* csize = (int*)calloc((size_t)n2, sizeof(int));
  uint64_t* cminkey = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n2);
  double* cmind = (double*)malloc(sizeof(double) * (size_t)n2);
  for (int i = 0; i < n2; ++i) { cminkey[i] = UINT64_MAX; cmind[i] = 1e300; }
  for (int i = 0; i < n2; ++i) {
    int r = uf_find(parent, i);
    csize[r]++;
    if (kw[i] < cminkey[r]) cminkey[r] = kw[i];
    double dx = center[0] - pw[i].x, dy = center[1] - pw[i].y, dz = center[2] - pw[i].z;
    double d = sqrt(dx * dx + dy * dy + dz * dz);
    if (d < cmind[r]) cmind[r] = d;
  }
  int ncl = 0, chosen = -1, only = -1;
  for (int r = 0; r < n2; ++r) if (csize[r] >= P->min_cluster_size) { ncl++; only = r; }
  if (ncl == 1) chosen = only;
  else
    for (int r = 0; r < n2; ++r) {
      if (csize[r] < P->min_cluster_size || !(cmind[r] < P->center_dis)) continue;
      if (chosen < 0 || csize[r] > csize[chosen] || (csize[r] == csize[chosen] && cminkey[r] < cminkey[chosen])) chosen = r;
    }
  if (dbg) dbg[3] = ncl;
  if (chosen < 0) { free(pw); free(kw); free(parent); free(csize); free(cminkey); free(cmind); return 2; }
  int nc = csize[chosen];
  f3* pc = (f3*)malloc(sizeof(f3) * (size_t)nc);
  { int m = 0; for (int i = 0; i < n2; ++i) if (uf_find(parent, i) == chosen) pc[m++] = pw[i]; }
  free(pw); free(kw); free(parent); free(csize); free(cminkey); free(cmind);
  if (dbg) dbg[4] = nc;
  /* 7. ProcessPCA (:170-196) */
  double cen[3] = {0, 0, 0}, cov[9] = {0};
  for (int i = 0; i < nc; ++i) { cen[0] += pc[i].x; cen[1] += pc[i].y; cen[2] += pc[i].z; }
  for (int r = 0; r < 3; ++r) cen[r] /= nc;
  for (int i = 0; i < nc; ++i) {
    double d[3] = {pc[i].x - cen[0], pc[i].y - cen[1], pc[i].z - cen[2]};
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) cov[a * 3 + b] += d[a] * d[b];
  }
  for (int a = 0; a < 9; ++a) cov[a] /= nc;
  double ev[3], rot[9];
  jacobi3(cov, ev, rot);
  /* AdjustChirality (:760-764) */
  { double c0[3] = {rot[0], rot[3], rot[6]}, c1[3] = {rot[1], rot[4], rot[7]}, c2[3]; cross3(c0, c1, c2); rot[2] = c2[0]; rot[5] = c2[1]; rot[8] = c2[2]; }
  /* AlignZAxisToGravity (:207-267) */
  double nh[3] = {ground[0] / gn, ground[1] / gn, ground[2] / gn};
  {
    double maxc = 0; int maxid = -1, pos = 1;
    for (int i = 0; i < 3; ++i) {
      double c = rot[i] * nh[0] + rot[3 + i] * nh[1] + rot[6 + i] * nh[2];
      if (fabs(c) > maxc) { maxc = fabs(c); pos = c > 0; maxid = i; }
    }
    if (maxid < 0) maxid = 2;
    double z[3], x[3], y[3];
    for (int r = 0; r < 3; ++r) { z[r] = pos ? rot[r * 3 + maxid] : -rot[r * 3 + maxid]; x[r] = rot[r * 3 + (maxid + 1) % 3]; }
    cross3(z, x, y);
    for (int r = 0; r < 3; ++r) { rot[r * 3] = x[r]; rot[r * 3 + 1] = y[r]; rot[r * 3 + 2] = z[r]; }
  }
  /* ApplyGravityPrior -> calibRotMatAccordingToGroundPlane (:766-788) */
  {
    double z[3] = {rot[2], rot[5], rot[8]}, nrm[3] = {ground[0], ground[1], ground[2]}, ax[3];
    cross3(z, nrm, ax);
    double an2 = ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2];
    if (an2 > 0) { double an = sqrt(an2); ax[0] /= an; ax[1] /= an; ax[2] /= an; }
    double zn = sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
    double ct = (nrm[0] * z[0] + nrm[1] * z[1] + nrm[2] * z[2]) / gn / zn;
    double th = acos(ct);
    double s = sin(th), c = cos(th); /* Eigen AngleAxis::toRotationMatrix */
    double sa[3] = {s * ax[0], s * ax[1], s * ax[2]}, c1[3] = {(1 - c) * ax[0], (1 - c) * ax[1], (1 - c) * ax[2]};
    double Ra[9], t;
    t = c1[0] * ax[1]; Ra[1] = t - sa[2]; Ra[3] = t + sa[2];
    t = c1[0] * ax[2]; Ra[2] = t + sa[1]; Ra[6] = t - sa[1];
    t = c1[1] * ax[2]; Ra[5] = t - sa[0]; Ra[7] = t + sa[0];
    Ra[0] = c1[0] * ax[0] + c; Ra[4] = c1[1] * ax[1] + c; Ra[8] = c1[2] * ax[2] + c;
    double out[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out[i * 3 + j] = Ra[i * 3] * rot[j] + Ra[i * 3 + 1] * rot[3 + j] + Ra[i * 3 + 2] * rot[6 + j];
    memcpy(rot, out, sizeof(out));
  }
  /* object frame (:326-347) */
  double Two_q[4], Two_t[3] = {cen[0], cen[1], cen[2]};
  {
    double xn = sqrt(rot[0] * rot[0] + rot[3] * rot[3] + rot[6] * rot[6]);
    double x[3] = {rot[0] / xn, rot[3] / xn, rot[6] / xn}, y[3];
    cross3(nh, x, y);
    double Rwo[9] = {x[0], y[0], nh[0], x[1], y[1], nh[1], x[2], y[2], nh[2]};
    R_to_quat(Rwo, Two_q);
  }
  double Two7[7] = {Two_t[0], Two_t[1], Two_t[2], Two_q[0], Two_q[1], Two_q[2], Two_q[3]}, Tow7[7];
  esl_oracle_se3_inv(Two7, Tow7);
  /* 8. DownSamplePointCloudOnly(GridSize) (:322-324) then transform into the object frame (:347) */
  f3* v2 = (f3*)malloc(sizeof(f3) * (size_t)nc);
  int ns0 = voxel_grid(pc, nc, (float)P->symmetry_grid, v2, NULL);
  free(pc);
  if (dbg) dbg[5] = ns0;
  double Row[9];
  quat_to_R(&Tow7[3], Row);
  int capo = 4 * ns0 + 4;
  v3* po = (v3*)malloc(sizeof(v3) * (size_t)capo);
  for (int i = 0; i < ns0; ++i) {
    double p[3] = {v2[i].x, v2[i].y, v2[i].z};
    po[i].x = Row[0] * p[0] + Row[1] * p[1] + Row[2] * p[2] + Tow7[0];
    po[i].y = Row[3] * p[0] + Row[4] * p[1] + Row[5] * p[2] + Tow7[1];
    po[i].z = Row[6] * p[0] + Row[7] * p[1] + Row[8] * p[2] + Tow7[2];
  }
  free(v2);
  int npo = ns0;
  double prob_sym = 1.0;
  /* 9. symmetry (:350-446) */
  int stype = symmetry_type(label);
  if (P->symmetry_open && stype > 0 && ns0 > 0) {
    symctx c;
    c.depth = depth; c.w = w; c.h = h;
    memcpy(c.bbox, bbox, sizeof(c.bbox));
    c.K[0] = fx; c.K[1] = fy; c.K[2] = cx; c.K[3] = cy; c.scale = scale; c.sigma = P->symmetry_sigma;
    double Toc[7], Tco[7];
    esl_oracle_se3_mul(Tow7, Twc, Toc); /* campose_oc = SE3Tow * campose_wc (:371) */
    esl_oracle_se3_inv(Toc, Tco);
    double Rco[9];
    quat_to_R(&Tco[3], Rco);
    for (int cc = 0; cc < 4; ++cc) {
      double c0 = cc < 3 ? Rco[cc] : Tco[0], c1 = cc < 3 ? Rco[3 + cc] : Tco[1], c2 = cc < 3 ? Rco[6 + cc] : Tco[2];
      c.P[cc] = fx * c0 + cx * c2; c.P[4 + cc] = fy * c1 + cy * c2; c.P[8 + cc] = c2;
    }
    c.camc[0] = Toc[0]; c.camc[1] = Toc[1]; c.camc[2] = Toc[2];
    f3* pof = (f3*)malloc(sizeof(f3) * (size_t)ns0);
    for (int i = 0; i < ns0; ++i) { pof[i].x = (float)po[i].x; pof[i].y = (float)po[i].y; pof[i].z = (float)po[i].z; }
    c.pts = po; c.ptsf = pof; c.n = ns0;
    v3* buf = (v3*)malloc(sizeof(v3) * (size_t)(2 * ns0 + 2));
    /* GenerateInitPlanes (Symmetry.cpp:44-86): 3 distances x 3 angles */
    plane_t best; double bestp = -1; memset(&best, 0, sizeof(best));
    for (int i = 0; i < 3; ++i)
      for (int m = 0; m < 3; ++m) {
        plane_t pl;
        plane_from_dis_angle(&pl, -0.2 + 0.2 * i, -(M_PI / 180.0 * 5) + (M_PI / 180.0 * 5) * m);
        double e = sym_optimize(&c, &pl, stype == 2, P->symmetry_lm_iters, buf);
        double pr = exp(-e);
        if (dbg && i * 3 + m < 9) dbg[6 + i * 3 + m] = pr;
        if (pr > bestp) { bestp = pr; best = pl; }
      }
    prob_sym = bestp;
    free(buf); free(pof);
    /* complete the cloud with the mirrored points (:399-446) */
    int m0 = npo;
    for (int i = 0; i < m0; ++i) po[npo++] = mirror_point(po[i], best.p);
    if (stype == 2) {
      double p2[4];
      plane_another(&best, p2);
      for (int i = 0; i < m0; ++i) po[npo++] = mirror_point(po[i], p2);          /* sym2_1 */
      for (int i = 0; i < m0; ++i) po[npo++] = mirror_point(po[m0 + i], p2);     /* sym2_2 = mirror(sym1) */
    }
    double cc[3] = {0, 0, 0};
    for (int i = 0; i < npo; ++i) { cc[0] += po[i].x; cc[1] += po[i].y; cc[2] += po[i].z; }
    for (int r = 0; r < 3; ++r) cc[r] /= (double)npo;
    if (sym) { /* SymmetryOutputData (:376-393, 415-423): plane::transform(*pSE3Two) = (Two^T)^-1 pi (Plane.cpp:117-122) */
      double Rwo[9], p2[4];
      quat_to_R(&Two7[3], Rwo);
      plane_another(&best, p2);
      for (int pl = 0; pl < (stype == 2 ? 2 : 1); ++pl) {
        const double* q = pl ? p2 : best.p;
        double n[3];
        for (int r = 0; r < 3; ++r) n[r] = Rwo[r * 3] * q[0] + Rwo[r * 3 + 1] * q[1] + Rwo[r * 3 + 2] * q[2];
        sym[4 * pl] = n[0]; sym[4 * pl + 1] = n[1]; sym[4 * pl + 2] = n[2];
        sym[4 * pl + 3] = q[3] - (Two7[0] * n[0] + Two7[1] * n[1] + Two7[2] * n[2]);
      }
      sym[8] = bestp; sym[9] = stype; sym[10] = 1;
      for (int r = 0; r < 3; ++r) sym[11 + r] = Rwo[r * 3] * cc[0] + Rwo[r * 3 + 1] * cc[1] + Rwo[r * 3 + 2] * cc[2] + Two7[r];
    }
    double nn = sqrt(best.p[0] * best.p[0] + best.p[1] * best.p[1] + best.p[2] * best.p[2]);
    double x[3] = {best.p[0] / nn, best.p[1] / nn, best.p[2] / nn}, z[3] = {0, 0, 1}, y[3];
    cross3(z, x, y);
    double Rom[9] = {x[0], y[0], z[0], x[1], y[1], z[1], x[2], y[2], z[2]}, qom[4];
    R_to_quat(Rom, qom);
    double Tom[7] = {cc[0], cc[1], cc[2], qom[0], qom[1], qom[2], qom[3]}, Tmo[7];
    esl_oracle_se3_inv(Tom, Tmo);
    double Rmo[9];
    quat_to_R(&Tmo[3], Rmo);
    for (int i = 0; i < npo; ++i) {
      double p[3] = {po[i].x, po[i].y, po[i].z};
      po[i].x = Rmo[0] * p[0] + Rmo[1] * p[1] + Rmo[2] * p[2] + Tmo[0];
      po[i].y = Rmo[3] * p[0] + Rmo[4] * p[1] + Rmo[5] * p[2] + Tmo[1];
      po[i].z = Rmo[6] * p[0] + Rmo[7] * p[1] + Rmo[8] * p[2] + Tmo[2];
    }
    double Twm[7];
    esl_oracle_se3_mul(Two7, Tom, Twm); /* (*pSE3Two) * Tom */
    memcpy(Two7, Twm, sizeof(Twm));
  }
  /* 10. ProcessPCANormalized (:495-531) + ConstructEllipsoid (:270-290) + transforms (:455-469) */
  double mx = 0, my = 0, mz = 0;
  for (int i = 0; i < npo; ++i) { if (fabs(po[i].x) > mx) mx = fabs(po[i].x); if (fabs(po[i].y) > my) my = fabs(po[i].y); if (fabs(po[i].z) > mz) mz = fabs(po[i].z); }
  free(po);
  double I7[7] = {0, 0, 0, 0, 0, 0, 1}, g7[7], l7[7], Tcw[7];
  esl_oracle_se3_mul(Two7, I7, g7);      /* transform_from(*pSE3Two) */
  esl_oracle_se3_inv(Twc, Tcw);          /* campose_wc.fromVector(pose); .inverse() */
  esl_oracle_se3_mul(Tcw, g7, l7);       /* transform_from(campose_wc.inverse()) */
  for (int i = 0; i < 7; ++i) ell[i] = l7[i];
  ell[7] = mx; ell[8] = my; ell[9] = mz;
  *prob = prob_sym;
  return 0;
}

int esl_oracle_fit_frame_ex(const uint16_t* depth, int32_t width, int32_t height, const double* bboxes, const int32_t* labels,
                            int32_t n_boxes, const double Twc[7], const double intr[5], const double ground[4],
                            const esl_fit_params* p, double* ellipsoids_out, double* prob_out, int32_t* status_out,
                            double* debug_out, double* sym_out);

int esl_oracle_fit_frame(const uint16_t* depth, int32_t width, int32_t height, const double* bboxes, const int32_t* labels,
                         int32_t n_boxes, const double Twc[7], const double intr[5], const double ground[4],
                         const esl_fit_params* p, double* ellipsoids_out, double* prob_out, int32_t* status_out,
                         double* debug_out /* n_boxes x 16 or NULL */) {
  return esl_oracle_fit_frame_ex(depth, width, height, bboxes, labels, n_boxes, Twc, intr, ground, p, ellipsoids_out, prob_out,
                                 status_out, debug_out, NULL);
}

/* same + SymmetryOutputData per box: 16 doubles = plane (world, 4) | plane2 (4) | prob | type | result | centre (3) | pad */
int esl_oracle_fit_frame_ex(const uint16_t* depth, int32_t width, int32_t height, const double* bboxes, const int32_t* labels,
                            int32_t n_boxes, const double Twc[7], const double intr[5], const double ground[4],
                            const esl_fit_params* p, double* ellipsoids_out, double* prob_out, int32_t* status_out,
                            double* debug_out, double* sym_out /* n_boxes x 16 or NULL */) {
  for (int b = 0; b < n_boxes; ++b)
    status_out[b] = fit_one(depth, width, height, &bboxes[4 * b], labels ? labels[b] : -1, Twc, intr, ground, p,
                            &ellipsoids_out[10 * b], &prob_out[b], debug_out ? &debug_out[16 * b] : NULL,
                            sym_out ? &sym_out[16 * b] : NULL);
  return 0;
}
