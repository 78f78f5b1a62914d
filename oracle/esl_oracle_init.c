/*
 * esl_oracle_init.c — CPU restatement of Initializer::initializeQuadric (reference
 * src/core/Initializer.cpp:24-248).  TEST INFRASTRUCTURE ONLY (see esl_oracle.h).
 *
 * Eigen pieces restated: JacobiSVD null vector -> one-sided (Hestenes) Jacobi SVD, last right singular
 * vector; SelfAdjointEigenSolver -> cyclic Jacobi, eigenvalues ascending; 4x4 inverse/determinant ->
 * Gauss-Jordan with partial pivoting.  Eigen leaves the SIGN of the SVD null vector and of eigenvectors
 * unspecified; the convention here (documented in SURVEY.md A.7): q_hat is signed so that Q*_33 > 0 in
 * faithful mode, eigenvector columns are signed so that their largest-magnitude component is positive.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "esl_oracle.h"

/* cyclic Jacobi for a symmetric n x n matrix (row-major, n <= 4); eigenvalues ascending, columns of V */
static void sym_eig(const double* Ain, int n, double* w, double* V) {
  double A[16];
  memcpy(A, Ain, sizeof(double) * (size_t)(n * n));
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = (i == j);
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
    for (int p = 0; p < n; ++p) for (int q = p + 1; q < n; ++q) off += A[p * n + q] * A[p * n + q];
    if (off < 1e-300) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        double apq = A[p * n + q];
        if (apq == 0.0) continue;
        double theta = (A[q * n + q] - A[p * n + p]) / (2 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
        double c = 1 / sqrt(t * t + 1), s = t * c;
        for (int k = 0; k < n; ++k) { double akp = A[k * n + p], akq = A[k * n + q]; A[k * n + p] = c * akp - s * akq; A[k * n + q] = s * akp + c * akq; }
        for (int k = 0; k < n; ++k) { double apk = A[p * n + k], aqk = A[q * n + k]; A[p * n + k] = c * apk - s * aqk; A[q * n + k] = s * apk + c * aqk; }
        for (int k = 0; k < n; ++k) { double vkp = V[k * n + p], vkq = V[k * n + q]; V[k * n + p] = c * vkp - s * vkq; V[k * n + q] = s * vkp + c * vkq; }
      }
  }
  for (int i = 0; i < n; ++i) w[i] = A[i * n + i];
  for (int i = 0; i < n; ++i) { /* selection sort ascending */
    int m = i;
    for (int j = i + 1; j < n; ++j) if (w[j] < w[m]) m = j;
    if (m != i) {
      double t = w[i]; w[i] = w[m]; w[m] = t;
      for (int k = 0; k < n; ++k) { double u = V[k * n + i]; V[k * n + i] = V[k * n + m]; V[k * n + m] = u; }
    }
  }
  for (int j = 0; j < n; ++j) { /* sign convention */
    int m = 0;
    for (int k = 1; k < n; ++k) if (fabs(V[k * n + j]) > fabs(V[m * n + j])) m = k;
    if (V[m * n + j] < 0) for (int k = 0; k < n; ++k) V[k * n + j] = -V[k * n + j];
  }
}

static double inv_det(const double* Ain, int n, double* inv) { /* Gauss-Jordan, returns determinant */
  double A[16], det = 1;
  memcpy(A, Ain, sizeof(double) * (size_t)(n * n));
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) inv[i * n + j] = (i == j);
  for (int c = 0; c < n; ++c) {
    int p = c;
    for (int r = c + 1; r < n; ++r) if (fabs(A[r * n + c]) > fabs(A[p * n + c])) p = r;
    if (p != c) {
      for (int k = 0; k < n; ++k) { double t = A[c * n + k]; A[c * n + k] = A[p * n + k]; A[p * n + k] = t; t = inv[c * n + k]; inv[c * n + k] = inv[p * n + k]; inv[p * n + k] = t; }
      det = -det;
    }
    double d = A[c * n + c];
    det *= d;
    for (int k = 0; k < n; ++k) { A[c * n + k] /= d; inv[c * n + k] /= d; }
    for (int r = 0; r < n; ++r) if (r != c) {
      double f = A[r * n + c];
      for (int k = 0; k < n; ++k) { A[r * n + k] -= f * A[c * n + k]; inv[r * n + k] -= f * inv[c * n + k]; }
    }
  }
  return det;
}

/* planes of Initializer::getPlanesHomo (:58-91) + fromDetectionsToLines (:107-145); returns count, A is m x 10 */
static int build_rows(const double* poses, const double* boxes, int n, const double K[4], int rows, int cols, double* A) {
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const double* d = &boxes[4 * i];
    if (d[0] < 1 && d[1] < 1 && d[2] < 1 && d[3] < 1) continue;
    double Tcw[7], Twc[7];
    memcpy(Twc, &poses[7 * i], sizeof(Twc));
    { /* SE3Quat(Vector7d) normalises (se3quat.h:66-69) */
      double qn = sqrt(Twc[3] * Twc[3] + Twc[4] * Twc[4] + Twc[5] * Twc[5] + Twc[6] * Twc[6]);
      double sg = Twc[6] < 0 ? -1.0 : 1.0;
      for (int k = 3; k < 7; ++k) Twc[k] = sg * Twc[k] / qn;
    }
    esl_oracle_se3_inv(Twc, Tcw);
    /* P = K [R|t] */
    double x = Tcw[3], y = Tcw[4], z = Tcw[5], w = Tcw[6];
    double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                   2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                   2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
    double P[12];
    for (int c = 0; c < 4; ++c) {
      double c0 = c < 3 ? R[c] : Tcw[0], c1 = c < 3 ? R[3 + c] : Tcw[1], c2 = c < 3 ? R[6 + c] : Tcw[2];
      P[c] = K[0] * c0 + K[2] * c2; P[4 + c] = K[1] * c1 + K[3] * c2; P[8 + c] = c2;
    }
    double lines[4][3] = {{1, 0, -d[0]}, {0, 1, -d[1]}, {1, 0, -d[2]}, {0, 1, -d[3]}};
    int keep[4] = {d[0] > 0 && d[0] < cols - 1, d[1] > 0 && d[1] < rows - 1, d[2] > 0 && d[2] < cols - 1, d[3] > 0 && d[3] < rows - 1};
    for (int l = 0; l < 4; ++l) {
      if (!keep[l]) continue;
      double p[4];
      for (int c = 0; c < 4; ++c) p[c] = P[c] * lines[l][0] + P[4 + c] * lines[l][1] + P[8 + c] * lines[l][2];
      double* v = &A[10 * m++]; /* :157 */
      v[0] = p[0] * p[0]; v[1] = 2 * p[0] * p[1]; v[2] = 2 * p[0] * p[2]; v[3] = 2 * p[0] * p[3]; v[4] = p[1] * p[1];
      v[5] = 2 * p[1] * p[2]; v[6] = 2 * p[1] * p[3]; v[7] = p[2] * p[2]; v[8] = 2 * p[2] * p[3]; v[9] = p[3] * p[3];
    }
  }
  return m;
}

/* last right singular vector of A (m x 10) by one-sided Jacobi */
static void null_vector(double* A, int m, double q[10]) {
  double V[100];
  for (int i = 0; i < 10; ++i) for (int j = 0; j < 10; ++j) V[i * 10 + j] = (i == j);
  for (int sweep = 0; sweep < 60; ++sweep) {
    int rotated = 0;
    for (int p = 0; p < 10; ++p)
      for (int c = p + 1; c < 10; ++c) {
        double al = 0, be = 0, ga = 0;
        for (int r = 0; r < m; ++r) { al += A[r * 10 + p] * A[r * 10 + p]; be += A[r * 10 + c] * A[r * 10 + c]; ga += A[r * 10 + p] * A[r * 10 + c]; }
        if (fabs(ga) <= 1e-15 * sqrt(al * be) || ga == 0.0) continue;
        rotated = 1;
        double zeta = (be - al) / (2 * ga);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
        double cs = 1 / sqrt(1 + t * t), sn = cs * t;
        for (int r = 0; r < m; ++r) { double ap = A[r * 10 + p], aq = A[r * 10 + c]; A[r * 10 + p] = cs * ap - sn * aq; A[r * 10 + c] = sn * ap + cs * aq; }
        for (int r = 0; r < 10; ++r) { double vp = V[r * 10 + p], vq = V[r * 10 + c]; V[r * 10 + p] = cs * vp - sn * vq; V[r * 10 + c] = sn * vp + cs * vq; }
      }
    if (!rotated) break;
  }
  int best = 0; double bn = -1;
  for (int c = 0; c < 10; ++c) {
    double s = 0;
    for (int r = 0; r < m; ++r) s += A[r * 10 + c] * A[r * 10 + c];
    if (bn < 0 || s < bn) { bn = s; best = c; }
  }
  for (int r = 0; r < 10; ++r) q[r] = V[r * 10 + best];
}

static void rot_to_euler_zyx(const double R[9], double* roll, double* pitch, double* yaw) { /* matrix_utils.cpp:75-95 */
  *pitch = asin(-R[6]);
  if (fabs(*pitch - M_PI / 2.0) < 1.0e-3) { *roll = 0.0; *yaw = atan2(R[5] - R[1], R[2] + R[4]) + *roll; }
  else if (fabs(*pitch + M_PI / 2.0) < 1.0e-3) { *roll = 0.0; *yaw = atan2(R[5] - R[1], R[2] + R[4]) - *roll; }
  else { *roll = atan2(R[7], R[8]); *yaw = atan2(R[3], R[0]); }
}

int esl_oracle_init_quadric(const double* poses_Twc, const double* bboxes, int n, const double K[4], int rows, int cols,
                            int faithful, double ell_out[10], double qstar_out[16], int* ok) {
  *ok = 0;
  for (int i = 0; i < 10; ++i) ell_out[i] = 0;
  for (int i = 0; i < 16; ++i) qstar_out[i] = 0;
  double* A = (double*)malloc(sizeof(double) * 10 * (size_t)(4 * n + 1));
  int m = build_rows(poses_Twc, bboxes, n, K, rows, cols, A);
  if (m < 9) { free(A); return 0; } /* :38 */
  double q[10];
  null_vector(A, m, q);
  free(A);
  if (q[9] < 0) for (int i = 0; i < 10; ++i) q[i] = -q[i]; /* sign convention: Q*_33 > 0 */
  double Qs[16] = {q[0], q[1], q[2], q[3], q[1], q[4], q[5], q[6], q[2], q[5], q[7], q[8], q[3], q[6], q[8], q[9]};
  memcpy(qstar_out, Qs, sizeof(Qs));
  double t[3] = {Qs[3] / Qs[15], Qs[7] / Qs[15], Qs[11] / Qs[15]}; /* :229-231 */
  double s[3], Rm[9];
  if (faithful) {
    double Qi[16], Q[16], w4[4], V4[16];
    double det = inv_det(Qs, 4, Qi);
    double cb = cbrt(det);
    for (int i = 0; i < 16; ++i) Q[i] = Qi[i] * cb; /* :189 */
    for (int i = 0; i < 4; ++i) for (int j = i + 1; j < 4; ++j) { double a = 0.5 * (Q[i * 4 + j] + Q[j * 4 + i]); Q[i * 4 + j] = Q[j * 4 + i] = a; }
    sym_eig(Q, 4, w4, V4);
    int np = 0, nn = 0;
    for (int i = 0; i < 4; ++i) { np += w4[i] > 0; nn += w4[i] < 0; }
    if (!((np == 3 && nn == 1) || (np == 1 && nn == 3))) return 0; /* :198-204 */
    if (w4[3] > 0) { for (int i = 0; i < 16; ++i) Q[i] = -Q[i]; sym_eig(Q, 4, w4, V4); } /* :209-217 */
    double Q33[9] = {Q[0], Q[1], Q[2], Q[4], Q[5], Q[6], Q[8], Q[9], Q[10]};
    double tmp[16];
    double detQ = inv_det(Q, 4, tmp), detQ33 = inv_det(Q33, 3, tmp);
    double k = detQ / detQ33; /* :224 */
    for (int i = 0; i < 3; ++i) s[i] = sqrt(fabs(-k * (1.0 / w4[i]))); /* :220-227 */
    double w3[3];
    sym_eig(Q33, 3, w3, Rm); /* :233-235 */
  } else {
    /* exact decomposition: Q* / (-Q*_33) = [[R D R^T - t t^T, -t], [-t^T, -1]] */
    double sc = -1.0 / Qs[15], M[9], w3[3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[i * 3 + j] = Qs[i * 4 + j] * sc + t[i] * t[j];
    for (int i = 0; i < 3; ++i) for (int j = i + 1; j < 3; ++j) { double a = 0.5 * (M[i * 3 + j] + M[j * 3 + i]); M[i * 3 + j] = M[j * 3 + i] = a; }
    sym_eig(M, 3, w3, Rm);
    if (!(w3[0] > 0)) return 0;
    for (int i = 0; i < 3; ++i) s[i] = sqrt(w3[i]);
    /* right-handed frame */
    double det = Rm[0] * (Rm[4] * Rm[8] - Rm[5] * Rm[7]) - Rm[1] * (Rm[3] * Rm[8] - Rm[5] * Rm[6]) + Rm[2] * (Rm[3] * Rm[7] - Rm[4] * Rm[6]);
    if (det < 0) for (int r = 0; r < 3; ++r) Rm[r * 3 + 2] = -Rm[r * 3 + 2];
  }
  double roll, pitch, yaw;
  rot_to_euler_zyx(Rm, &roll, &pitch, &yaw);
  double v9[9] = {t[0], t[1], t[2], roll, pitch, yaw, s[0], s[1], s[2]};
  esl_oracle_obj_from_minimal(v9, ell_out); /* e.fromMinimalVector (:246) */
  *ok = 1;
  return 0;
}
