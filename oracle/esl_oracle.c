/*
 * esl_oracle.c — CPU restatement of the reference graph-optimisation path.  See esl_oracle.h:
 * TEST INFRASTRUCTURE ONLY; parity unpinned by the reference's own tests (it has none).
 *
 * All citations are relative to /root/reference.  The arithmetic that the reference delegates to
 * Eigen (not vendored; unpinned ">= 3.1.0", CMakeLists.txt:32) restates Eigen 3.3's published
 * algorithms: Quaternion<->matrix conversion, quaternion product / vector rotation, 3x3 cofactor
 * inverse, LDLT with diagonal pivoting.
 */
#include "esl_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------------
 * small fixed-size helpers
 * ----------------------------------------------------------------------------------------------*/
typedef struct { double x, y, z, w; } quat;            /* Eigen coeffs order x y z w */
typedef struct { quat r; double t[3]; } se3;           /* g2o::SE3Quat (types/se3quat.h:47-48) */
typedef struct { se3 pose; double s[3]; } ell;         /* g2o::ellipsoid pose + scale (include/core/Ellipsoid.h) */

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static double g_timing[3];
void esl_oracle_last_timing(double t[3]) { t[0] = g_timing[0]; t[1] = g_timing[1]; t[2] = g_timing[2]; }
/* split of the solve time of ESL_ORACLE_CAMFIRST: band check + chain factor + forward substitution (Y) | T = D - Y^T Y | pivoted
 * LDLT of T | back-substitution; summed over the trials of the last esl_oracle_optimize */
static double g_cf_timing[4];
void esl_oracle_last_camfirst_timing(double t[4]) { for (int k = 0; k < 4; ++k) t[k] = g_cf_timing[k]; }

/* Eigen QuaternionBase::operator* */
static quat q_mul(quat a, quat b) {
  quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
static quat q_conj(quat a) { quat r = {-a.x, -a.y, -a.z, a.w}; return r; }

/* Eigen QuaternionBase::_transformVector: v + w*uv + vec x uv, uv = 2 vec x v */
static void q_rot(quat q, const double v[3], double out[3]) {
  double ux = q.y * v[2] - q.z * v[1], uy = q.z * v[0] - q.x * v[2], uz = q.x * v[1] - q.y * v[0];
  ux += ux; uy += uy; uz += uz;
  double cx = q.y * uz - q.z * uy, cy = q.z * ux - q.x * uz, cz = q.x * uy - q.y * ux;
  out[0] = v[0] + q.w * ux + cx;
  out[1] = v[1] + q.w * uy + cy;
  out[2] = v[2] + q.w * uz + cz;
}

/* Eigen QuaternionBase::toRotationMatrix (row-major 3x3 out) */
static void q_to_R(quat q, double R[9]) {
  double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

/* Eigen quaternionbase_assign_impl<Other,3,3>: Quaterniond(R) */
static quat q_from_R(const double R[9]) {
  quat q;
  double t = R[0] + R[4] + R[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (R[7] - R[5]) * t;
    q.y = (R[2] - R[6]) * t;
    q.z = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 3 + i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    double c[3];
    t = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
    c[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (R[k * 3 + j] - R[j * 3 + k]) * t;
    c[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    c[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    q.x = c[0]; q.y = c[1]; q.z = c[2];
  }
  return q;
}

/* SE3Quat::normalizeRotation (se3quat.h:345-350) */
static quat q_normalize_pos(quat q) {
  if (q.w < 0) { q.x *= -1; q.y *= -1; q.z *= -1; q.w *= -1; }
  double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  q.x /= n; q.y /= n; q.z /= n; q.w /= n;
  return q;
}

static se3 se3_from7(const double v[7]) { /* SE3Quat::fromVector (se3quat.h:158-161): no normalisation */
  se3 T;
  T.t[0] = v[0]; T.t[1] = v[1]; T.t[2] = v[2];
  T.r.x = v[3]; T.r.y = v[4]; T.r.z = v[5]; T.r.w = v[6];
  return T;
}
static void se3_to7(se3 T, double v[7]) {
  v[0] = T.t[0]; v[1] = T.t[1]; v[2] = T.t[2];
  v[3] = T.r.x; v[4] = T.r.y; v[5] = T.r.z; v[6] = T.r.w;
}

/* SE3Quat::operator* (se3quat.h:110-116) */
static se3 se3_mul(se3 A, se3 B) {
  se3 R = A;
  double rt[3];
  q_rot(A.r, B.t, rt);
  R.t[0] += rt[0]; R.t[1] += rt[1]; R.t[2] += rt[2];
  R.r = q_normalize_pos(q_mul(A.r, B.r));
  return R;
}

/* SE3Quat::inverse (se3quat.h:129-134): conjugate, no renormalisation */
static se3 se3_inv(se3 A) {
  se3 R;
  R.r = q_conj(A.r);
  double nt[3] = {A.t[0] * -1., A.t[1] * -1., A.t[2] * -1.};
  q_rot(R.r, nt, R.t);
  return R;
}

static void skew(const double v[3], double m[9]) { /* se3_ops.hpp:28-39 */
  m[0] = 0;     m[1] = -v[2]; m[2] = v[1];
  m[3] = v[2];  m[4] = 0;     m[5] = -v[0];
  m[6] = -v[1]; m[7] = v[0];  m[8] = 0;
}
static void m3_mul(const double a[9], const double b[9], double c[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += a[i * 3 + k] * b[k * 3 + j];
      c[i * 3 + j] = s;
    }
}
static void m3_vec(const double a[9], const double v[3], double o[3]) {
  for (int i = 0; i < 3; ++i) o[i] = a[i * 3] * v[0] + a[i * 3 + 1] * v[1] + a[i * 3 + 2] * v[2];
}

/* SE3Quat::exp (se3quat.h:274-322).  NOTE the small-angle branch R = I + Omega + Omega^2, V = R. */
static se3 se3_exp(const double u[6]) {
  double omega[3] = {u[0], u[1], u[2]}, ups[3] = {u[3], u[4], u[5]};
  double theta = sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
  double Om[9], Om2[9], R[9], V[9];
  skew(omega, Om);
  m3_mul(Om, Om, Om2);
  static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (theta < 0.00001) {
    for (int i = 0; i < 9; ++i) { R[i] = I3[i] + Om[i] + Om2[i]; V[i] = R[i]; }
  } else {
    double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta);
    double c = (theta - sin(theta)) / pow(theta, 3);
    for (int i = 0; i < 9; ++i) {
      R[i] = I3[i] + a * Om[i] + b * Om2[i];
      V[i] = I3[i] + b * Om[i] + c * Om2[i];
    }
  }
  se3 T;
  T.r = q_normalize_pos(q_from_R(R)); /* SE3Quat(Quaterniond(R), V*upsilon) normalises (se3quat.h:60-62) */
  m3_vec(V, ups, T.t);
  return T;
}

/* SE3Quat::log (se3quat.h:229-266) */
static void se3_log(se3 T, double out[6]) {
  double R[9];
  q_to_R(T.r, R);
  double d = 0.5 * (R[0] + R[4] + R[8] - 1);
  double dR[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]}; /* deltaR, se3_ops.hpp:41-48 */
  double omega[3], Om[9], Om2[9], Vinv[9];
  static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (d > 0.99999) {
    for (int i = 0; i < 3; ++i) omega[i] = 0.5 * dR[i];
    skew(omega, Om);
    m3_mul(Om, Om, Om2);
    for (int i = 0; i < 9; ++i) Vinv[i] = I3[i] - 0.5 * Om[i] + (1. / 12.) * Om2[i];
  } else {
    double theta = acos(d);
    double f = theta / (2 * sqrt(1 - d * d));
    for (int i = 0; i < 3; ++i) omega[i] = f * dR[i];
    skew(omega, Om);
    m3_mul(Om, Om, Om2);
    double c = (1 - theta / (2 * tan(theta / 2))) / (theta * theta);
    for (int i = 0; i < 9; ++i) Vinv[i] = I3[i] - 0.5 * Om[i] + c * Om2[i];
  }
  double ups[3];
  m3_vec(Vinv, T.t, ups);
  for (int i = 0; i < 3; ++i) { out[i] = omega[i]; out[i + 3] = ups[i]; }
}

/* SE3Quat::to_homogeneous_matrix (se3quat.h:335-343), row-major 4x4 */
static void se3_to_M(se3 T, double M[16]) {
  double R[9];
  q_to_R(T.r, R);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) M[i * 4 + j] = R[i * 3 + j];
    M[i * 4 + 3] = T.t[i];
  }
  M[12] = 0; M[13] = 0; M[14] = 0; M[15] = 1;
}

static ell ell_from10(const double v[10]) { /* ellipsoid::fromVector (Ellipsoid.cpp:25-29) */
  ell e;
  e.pose = se3_from7(v);
  e.s[0] = v[7]; e.s[1] = v[8]; e.s[2] = v[9];
  return e;
}
static void ell_to10(ell e, double v[10]) {
  se3_to7(e.pose, v);
  v[7] = e.s[0]; v[8] = e.s[1]; v[9] = e.s[2];
}

/* ellipsoid::exp_update (Ellipsoid.cpp:38-47) */
static ell ell_exp_update(ell e, const double u[9]) {
  ell r;
  r.pose = se3_mul(e.pose, se3_exp(u));
  for (int i = 0; i < 3; ++i) r.s[i] = e.s[i] + u[6 + i];
  return r;
}

/* ellipsoid::generateQuadric (Ellipsoid.cpp:290-300) */
static void ell_quadric(ell e, double Q[16]) {
  double dstar[4];
  for (int i = 0; i < 3; ++i) dstar[i] = 1.0 / (1 / (e.s[i] * e.s[i])); /* Q_c.inverse() of a diagonal */
  dstar[3] = 1.0 / -1.0;
  double M[16];
  se3_to_M(e.pose, M);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += M[i * 4 + k] * dstar[k] * M[j * 4 + k];
      Q[i * 4 + j] = s;
    }
}

/* ellipsoid::getBoundingBoxFromProjection (Ellipsoid.cpp:303-307) =
 * projectOntoImageEllipse (:209-249) + getBoundingBoxFromEllipse (:252-273) */
static void ell_project_bbox(ell e, se3 Tcw, const double K[4], double out[4]) {
  double Q[16], Mc[16], P[12];
  ell_quadric(e, Q);
  se3_to_M(Tcw, Mc);
  /* generateProjectionMatrix (:276-287): P = K [I|0] Tcw */
  double Kf[9] = {K[0], 0, K[2], 0, K[1], K[3], 0, 0, 1};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += Kf[i * 3 + k] * Mc[k * 4 + j];
      P[i * 4 + j] = s;
    }
  double PQ[12], Cs[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += P[i * 4 + k] * Q[k * 4 + j];
      PQ[i * 4 + j] = s;
    }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += PQ[i * 4 + k] * P[j * 4 + k];
      Cs[i * 3 + j] = s;
    }
  /* C = C_star.inverse() : 3x3 cofactor inverse (Eigen compute_inverse_size3) */
  double C[9];
  double c00 = Cs[4] * Cs[8] - Cs[5] * Cs[7];
  double c10 = Cs[5] * Cs[6] - Cs[3] * Cs[8];
  double c20 = Cs[3] * Cs[7] - Cs[4] * Cs[6];
  double det = Cs[0] * c00 + Cs[1] * c10 + Cs[2] * c20;
  double inv = 1.0 / det;
  C[0] = c00 * inv;
  C[1] = (Cs[2] * Cs[7] - Cs[1] * Cs[8]) * inv;
  C[2] = (Cs[1] * Cs[5] - Cs[2] * Cs[4]) * inv;
  C[3] = c10 * inv;
  C[4] = (Cs[0] * Cs[8] - Cs[2] * Cs[6]) * inv;
  C[5] = (Cs[2] * Cs[3] - Cs[0] * Cs[5]) * inv;
  C[6] = c20 * inv;
  C[7] = (Cs[1] * Cs[6] - Cs[0] * Cs[7]) * inv;
  C[8] = (Cs[0] * Cs[4] - Cs[1] * Cs[3]) * inv;
  double n = C[8];
  for (int i = 0; i < 9; ++i) C[i] = C[i] / n; /* :215 */
  double a = C[0], b = C[1] * 2, c = C[4], d = C[2] * 2, ee = C[7] * 2; /* :226-231 */
  double theta = 1 / 2.0 * atan2(b, (a - c));
  double x_c = (b * ee - 2 * c * d) / (4 * a * c - b * b);
  double y_c = (b * d - 2 * a * ee) / (4 * a * c - b * b);
  double a_2 = 2 * (a * x_c * x_c + c * y_c * y_c + b * x_c * y_c - 1) / (a + c + sqrt((a - c) * (a - c) + b * b));
  double b_2 = 2 * (a * x_c * x_c + c * y_c * y_c + b * x_c * y_c - 1) / (a + c - sqrt((a - c) * (a - c) + b * b));
  double axis1 = sqrt(a_2), axis2 = sqrt(b_2);
  /* getBoundingBoxFromEllipse (:252-273) */
  double cos_theta_2 = cos(theta) * cos(theta);
  double sin_theta_2 = 1 - cos_theta_2;
  double x_limit = sqrt(axis1 * axis1 * cos_theta_2 + axis2 * axis2 * sin_theta_2);
  double y_limit = sqrt(axis1 * axis1 * sin_theta_2 + axis2 * axis2 * cos_theta_2);
  out[0] = x_c - x_limit; out[1] = y_c - y_limit; out[2] = x_c + x_limit; out[3] = y_c + y_limit;
}

/* ellipsoid::rotate_ellipsoid (Ellipsoid.cpp:75-90) */
static ell ell_rotate(ell e, double yaw) {
  ell r;
  se3 rot;
  quat q = {0, 0, sin(yaw * 0.5), cos(yaw * 0.5)};
  rot.r = q_normalize_pos(q); /* SE3Quat(Quaterniond, Vector3d) ctor normalises */
  rot.t[0] = rot.t[1] = rot.t[2] = 0;
  r.pose = se3_mul(e.pose, rot);
  r.s[0] = e.s[0]; r.s[1] = e.s[1]; r.s[2] = e.s[2];
  const double eps = 1e-6;
  if ((fabs(yaw - M_PI / 2.0) < eps) || (fabs(yaw + M_PI / 2.0) < eps) || (fabs(yaw - 3 * M_PI / 2.0) < eps)) {
    double t = r.s[0]; r.s[0] = r.s[1]; r.s[1] = t;
  }
  return r;
}

/* ellipsoid::ellipsoid_log_error_9dof (Ellipsoid.cpp:63-71) */
static void ell_log_error(ell self, ell newone, double res[9]) {
  se3 diff = se3_mul(se3_inv(newone.pose), self.pose);
  se3_log(diff, res);
  for (int i = 0; i < 3; ++i) res[6 + i] = self.s[i] - newone.s[i];
}

/* ellipsoid::min_log_error_9dof (Ellipsoid.cpp:92-117); first minimum wins (Eigen minCoeff) */
static void ell_min_log_error(ell self, ell newone, double res[9]) {
  static const double angles[4] = {-1, 0, 1, 2};
  double best = 0, err[9];
  for (int i = 0; i < 4; ++i) {
    ell rc = ell_rotate(newone, angles[i] * M_PI / 2.0);
    ell_log_error(self, rc, err);
    double n2 = 0;
    for (int k = 0; k < 9; ++k) n2 += err[k] * err[k];
    double nn = sqrt(n2);
    if (i == 0 || nn < best) { best = nn; memcpy(res, err, sizeof(err)); }
  }
}

/* ------------------------------------------------------------------------------------------------
 * residuals
 * ----------------------------------------------------------------------------------------------*/
/* Plane-tangency rows (esl_lm_params::bbox_residual = 1; an optional extra of the product, NOT a reference edge): the
 * constraint of Initializer.cpp:147-164, 271-284 written literally -- P = K [I|0] Tcw (3 x 4), the four bbox lines
 * l (Initializer.cpp:115-142), planes pi = P^T l scaled to a unit normal, r_k = pi^T Q* pi with the 4 x 4 dual quadric. */
static int g_bbox_residual = 0;
void esl_oracle_set_bbox_residual(int mode) { g_bbox_residual = mode; }
static void res_tangency(se3 Tcw, ell e, const double K[4], const double meas[4], double r[4]) {
  double Q[16], Mc[16], P[12];
  ell_quadric(e, Q);
  se3_to_M(Tcw, Mc);
  double Kf[9] = {K[0], 0, K[2], 0, K[1], K[3], 0, 0, 1};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += Kf[i * 3 + k] * Mc[k * 4 + j];
      P[i * 4 + j] = s;
    }
  const double lines[4][3] = {{1, 0, -meas[0]}, {0, 1, -meas[1]}, {1, 0, -meas[2]}, {0, 1, -meas[3]}};
  for (int k = 0; k < 4; ++k) {
    double pi[4];
    for (int j = 0; j < 4; ++j) pi[j] = P[j] * lines[k][0] + P[4 + j] * lines[k][1] + P[8 + j] * lines[k][2];
    double nn = sqrt(pi[0] * pi[0] + pi[1] * pi[1] + pi[2] * pi[2]);
    for (int j = 0; j < 4; ++j) pi[j] /= nn;
    double v = 0;
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) v += pi[a] * Q[a * 4 + b] * pi[b];
    r[k] = (meas[k] >= 5) ? v : 0.0;
  }
}
/* EdgeSE3EllipsoidProj::computeError (BasicEllipsoidEdges.cpp:102-112) */
static void res_bbox(se3 Tcw, ell e, const double K[4], const double meas[4], double r[4]) {
  if (g_bbox_residual == 1) { res_tangency(Tcw, e, K, meas, r); return; }
  double proj[4];
  ell_project_bbox(e, Tcw, K, proj);
  for (int i = 0; i < 4; ++i) r[i] = (meas[i] >= 5) ? proj[i] - meas[i] : 0.0;
}
/* checkVisibility (Optimizer.cpp:35-81), statement by statement */
static int in_image(double u, double v, int rows, int cols) { return (u > 0 && u < cols) && (v > 0 && v < rows); }
static int check_visibility(se3 Tcw, ell e, const double K[4], int rows, int cols) {
  double Mc[16], Q[16];
  se3_to_M(Tcw, Mc);
  double ch[4] = {e.pose.t[0], e.pose.t[1], e.pose.t[2], 1.0}, pc[4];
  for (int i = 0; i < 4; ++i) { pc[i] = 0; for (int k = 0; k < 4; ++k) pc[i] += Mc[i * 4 + k] * ch[k]; }
  if (pc[2] < 0) return 0;                                  /* behind the camera */
  se3 Twc = se3_inv(Tcw);
  double X[4] = {Twc.t[0], Twc.t[1], Twc.t[2], 1.0};
  ell_quadric(e, Q);
  /* point_in_Q = X^T (Q*)^-1 X: solve Q* y = X by Gaussian elimination with partial pivoting */
  double A[4][5];
  for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) A[i][j] = Q[i * 4 + j]; A[i][4] = X[i]; }
  for (int c = 0; c < 4; ++c) {
    int p = c;
    for (int r2 = c + 1; r2 < 4; ++r2) if (fabs(A[r2][c]) > fabs(A[p][c])) p = r2;
    for (int k = 0; k < 5; ++k) { double t = A[c][k]; A[c][k] = A[p][k]; A[p][k] = t; }
    for (int r2 = 0; r2 < 4; ++r2) if (r2 != c) { double f = A[r2][c] / A[c][c]; for (int k = 0; k < 5; ++k) A[r2][k] -= f * A[c][k]; }
  }
  double piq = 0;
  for (int i = 0; i < 4; ++i) piq += X[i] * (A[i][4] / A[i][i]);
  if (piq < 0) return 0;                                    /* the camera centre is inside the ellipsoid */
  double u = (K[0] * pc[0] + K[2] * pc[2]) / pc[2], v = (K[1] * pc[1] + K[3] * pc[2]) / pc[2];
  if (in_image(u, v, rows, cols)) return 1;
  double bb[4];
  ell_project_bbox(e, Tcw, K, bb);
  if (in_image(bb[0], bb[1], rows, cols) || in_image(bb[2], bb[3], rows, cols)) return 1;
  return 0;
}
/* EdgeSE3Ellipsoid9DOF::computeError (BasicEllipsoidEdges.cpp:68-77) */
static void res_e3d(se3 Tcw, ell e, ell meas, double r[9]) {
  se3 Twc = se3_inv(Tcw);
  ell mw; /* ellipsoid::transform_from (Ellipsoid.cpp:120-129) */
  mw.pose = se3_mul(Twc, meas.pose);
  mw.s[0] = meas.s[0]; mw.s[1] = meas.s[1]; mw.s[2] = meas.s[2];
  ell_min_log_error(e, mw, r);
}
/* EdgeEllipsoidGravityPlanePrior::computeError (BasicEllipsoidEdges.cpp:129-152) */
static void res_grav(ell e, const double nrm[4], double r[1]) {
  double R[9];
  q_to_R(e.pose.r, R);
  double z[3] = {R[2], R[5], R[8]};
  double dot = z[0] * nrm[0] + z[1] * nrm[1] + z[2] * nrm[2];
  double zn = sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
  double nn = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
  double c = dot / zn / nn;
  if (c > 1) c = c - 0.0001;
  else if (c < -1) c = c + 0.0001;
  r[0] = acos(c) - 0;
}
/* EdgeSE3Expmap::computeError (types_six_dof_expmap.h:90-99) */
static void res_odom(se3 Ti, se3 Tj, se3 Z, double r[6]) {
  se3 err = se3_mul(se3_mul(Z, Ti), se3_inv(Tj));
  se3_log(err, r);
}

/* VertexSE3Expmap::oplusImpl (types_six_dof_expmap.h:73-76) */
static se3 cam_oplus(se3 T, const double u[6]) { return se3_mul(se3_exp(u), T); }

/* ------------------------------------------------------------------------------------------------
 * exported primitive wrappers
 * ----------------------------------------------------------------------------------------------*/
void esl_oracle_se3_exp(const double u[6], double out[7]) { se3_to7(se3_exp(u), out); }
void esl_oracle_se3_log(const double T[7], double out[6]) { se3_log(se3_from7(T), out); }
void esl_oracle_se3_mul(const double A[7], const double B[7], double out[7]) {
  se3_to7(se3_mul(se3_from7(A), se3_from7(B)), out);
}
void esl_oracle_se3_inv(const double A[7], double out[7]) { se3_to7(se3_inv(se3_from7(A)), out); }
void esl_oracle_cam_oplus(const double cam[7], const double u[6], double out[7]) {
  se3_to7(cam_oplus(se3_from7(cam), u), out);
}
void esl_oracle_obj_oplus(const double obj[10], const double u[9], double out[10]) {
  ell_to10(ell_exp_update(ell_from10(obj), u), out);
}
/* SE3Quat::toXYZPRYVector (se3quat.h:184-207) + ellipsoid::toMinimalVector (Ellipsoid.cpp:143-149) */
void esl_oracle_obj_to_minimal(const double obj[10], double out[9]) {
  double qx = obj[3], qy = obj[4], qz = obj[5], qw = obj[6];
  out[0] = obj[0]; out[1] = obj[1]; out[2] = obj[2];
  out[3] = atan2(2 * (qw * qx + qy * qz), 1 - 2 * (qx * qx + qy * qy));
  out[4] = asin(2 * (qw * qy - qz * qx));
  out[5] = atan2(2 * (qw * qz + qx * qy), 1 - 2 * (qy * qy + qz * qz));
  out[6] = obj[7]; out[7] = obj[8]; out[8] = obj[9];
}
/* ellipsoid::fromMinimalVector (Ellipsoid.cpp:16-22) + zyx_euler_to_quat (src/utils/matrix_utils.cpp:40-54) */
void esl_oracle_obj_from_minimal(const double v[9], double out[10]) {
  double roll = v[3], pitch = v[4], yaw = v[5];
  double sy = sin(yaw * 0.5), cy = cos(yaw * 0.5), sp = sin(pitch * 0.5), cp = cos(pitch * 0.5);
  double sr = sin(roll * 0.5), cr = cos(roll * 0.5);
  quat q;
  q.w = cr * cp * cy + sr * sp * sy;
  q.x = sr * cp * cy - cr * sp * sy;
  q.y = cr * sp * cy + sr * cp * sy;
  q.z = cr * cp * sy - sr * sp * cy;
  q = q_normalize_pos(q); /* SE3Quat(posequat, t) ctor */
  out[0] = v[0]; out[1] = v[1]; out[2] = v[2];
  out[3] = q.x; out[4] = q.y; out[5] = q.z; out[6] = q.w;
  out[7] = v[6]; out[8] = v[7]; out[9] = v[8];
}
void esl_oracle_quadric(const double obj[10], double Q[16]) { ell_quadric(ell_from10(obj), Q); }
void esl_oracle_project_bbox(const double cam[7], const double obj[10], const double K[4], double out[4]) {
  ell_project_bbox(ell_from10(obj), se3_from7(cam), K, out);
}
void esl_oracle_res_bbox(const double cam[7], const double obj[10], const double K[4], const double meas[4],
                         double r[4]) {
  res_bbox(se3_from7(cam), ell_from10(obj), K, meas, r);
}
void esl_oracle_res_e3d(const double cam[7], const double obj[10], const double meas[10], double r[9]) {
  res_e3d(se3_from7(cam), ell_from10(obj), ell_from10(meas), r);
}
void esl_oracle_res_grav(const double obj[10], const double normal[4], double r[1]) {
  res_grav(ell_from10(obj), normal, r);
}
void esl_oracle_res_odom(const double ci[7], const double cj[7], const double meas[7], double r[6]) {
  /* e->setMeasurement(SE3Quat) then SE3Quat C(_measurement): copy */
  res_odom(se3_from7(ci), se3_from7(cj), se3_from7(meas), r);
}

/* ------------------------------------------------------------------------------------------------
 * numeric Jacobians (base_binary_edge.hpp:131-205, base_unary_edge.hpp:82-123): central
 * differences; J is row-major D x dim.
 * ----------------------------------------------------------------------------------------------*/
typedef struct {
  int type; /* 0 bbox, 1 e3d, 2 grav, 3 odom */
  const double* K;
  const double* meas;
  ell meas_e;
  se3 meas_T;
  const double* normal;
} edge_ctx;

static int edge_dim(int type) { return type == 0 ? 4 : type == 1 ? 9 : type == 2 ? 1 : 6; }

static void edge_eval(const edge_ctx* c, se3 v0cam, ell v1obj, se3 v1cam, double* r) {
  switch (c->type) {
    case 0: res_bbox(v0cam, v1obj, c->K, c->meas, r); break;
    case 1: res_e3d(v0cam, v1obj, c->meas_e, r); break;
    case 2: res_grav(v1obj, c->normal, r); break;
    default: res_odom(v0cam, v1cam, c->meas_T, r); break;
  }
}

/* which: 0 -> perturb vertex 0 (always a camera), 1 -> perturb vertex 1 (object, or camera for odom) */
static void edge_numeric_jac(const edge_ctx* c, se3 v0cam, ell v1obj, se3 v1cam, int which, double delta,
                             double* J, int dim) {
  const int D = edge_dim(c->type);
  const double scalar = 1.0 / (2 * delta);
  double add[9], ep[9], em[9];
  for (int d = 0; d < dim; ++d) add[d] = 0;
  for (int d = 0; d < dim; ++d) {
    add[d] = delta;
    if (which == 0) edge_eval(c, cam_oplus(v0cam, add), v1obj, v1cam, ep);
    else if (c->type == 3) edge_eval(c, v0cam, v1obj, cam_oplus(v1cam, add), ep);
    else edge_eval(c, v0cam, ell_exp_update(v1obj, add), v1cam, ep);
    add[d] = -delta;
    if (which == 0) edge_eval(c, cam_oplus(v0cam, add), v1obj, v1cam, em);
    else if (c->type == 3) edge_eval(c, v0cam, v1obj, cam_oplus(v1cam, add), em);
    else edge_eval(c, v0cam, ell_exp_update(v1obj, add), v1cam, em);
    add[d] = 0.0;
    for (int k = 0; k < D; ++k) J[k * dim + d] = scalar * (ep[k] - em[k]);
  }
}

void esl_oracle_jac_bbox(const double cam[7], const double obj[10], const double K[4], const double meas[4],
                         double delta, double Jc[24], double Jo[36]) {
  edge_ctx c; memset(&c, 0, sizeof(c)); c.type = 0; c.K = K; c.meas = meas;
  se3 T = se3_from7(cam); ell e = ell_from10(obj);
  edge_numeric_jac(&c, T, e, T, 0, delta, Jc, 6);
  edge_numeric_jac(&c, T, e, T, 1, delta, Jo, 9);
}
void esl_oracle_jac_e3d(const double cam[7], const double obj[10], const double meas[10], double delta,
                        double Jc[54], double Jo[81]) {
  edge_ctx c; memset(&c, 0, sizeof(c)); c.type = 1; c.meas_e = ell_from10(meas);
  se3 T = se3_from7(cam); ell e = ell_from10(obj);
  edge_numeric_jac(&c, T, e, T, 0, delta, Jc, 6);
  edge_numeric_jac(&c, T, e, T, 1, delta, Jo, 9);
}
void esl_oracle_jac_grav(const double obj[10], const double normal[4], double delta, double Jo[9]) {
  edge_ctx c; memset(&c, 0, sizeof(c)); c.type = 2; c.normal = normal;
  ell e = ell_from10(obj);
  edge_numeric_jac(&c, e.pose, e, e.pose, 1, delta, Jo, 9);
}
void esl_oracle_jac_odom(const double ci[7], const double cj[7], const double meas[7], double delta,
                         double Ji[36], double Jj[36]) {
  edge_ctx c; memset(&c, 0, sizeof(c)); c.type = 3; c.meas_T = se3_from7(meas);
  se3 Ti = se3_from7(ci), Tj = se3_from7(cj);
  ell dummy; memset(&dummy, 0, sizeof(dummy));
  edge_numeric_jac(&c, Ti, dummy, Tj, 0, delta, Ji, 6);
  edge_numeric_jac(&c, Ti, dummy, Tj, 1, delta, Jj, 6);
}

/* ------------------------------------------------------------------------------------------------
 * Eigen::LDLT (unblocked, lower, diagonal pivoting) + solve, as used by linear_solver_dense.h:107-111
 * A row-major n x n, lower triangle used.
 * ----------------------------------------------------------------------------------------------*/
int esl_oracle_ldlt_solve(double* A, int n, const double* b, double* x) {
  int* tr = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  double* temp = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  int sign = 0; /* 0 ZeroSign, 1 PositiveSemiDef, -1 NegativeSemiDef, 2 Indefinite */
  if (n == 1) {
    tr[0] = 0;
    if (A[0] > 0) sign = 1; else if (A[0] < 0) sign = -1;
  } else {
    for (int k = 0; k < n; ++k) {
      int big = k;
      double bv = fabs(A[k * n + k]);
      for (int i = k + 1; i < n; ++i) {
        double v = fabs(A[i * n + i]);
        if (v > bv) { bv = v; big = i; }
      }
      tr[k] = big;
      if (k != big) { /* symmetric swap on the lower triangle (Eigen LDLT.h ldlt_inplace<Lower>) */
        int s = n - big - 1;
        for (int j = 0; j < k; ++j) { double t = A[k * n + j]; A[k * n + j] = A[big * n + j]; A[big * n + j] = t; }
        for (int i = 0; i < s; ++i) {
          double t = A[(big + 1 + i) * n + k]; A[(big + 1 + i) * n + k] = A[(big + 1 + i) * n + big];
          A[(big + 1 + i) * n + big] = t;
        }
        for (int i = k + 1; i < big; ++i) { double t = A[i * n + k]; A[i * n + k] = A[big * n + i]; A[big * n + i] = t; }
        { double t = A[k * n + k]; A[k * n + k] = A[big * n + big]; A[big * n + big] = t; }
      }
      int rs = n - k - 1;
      if (k > 0) {
        for (int j = 0; j < k; ++j) temp[j] = A[j * n + j] * A[k * n + j];
        double s = 0;
        for (int j = 0; j < k; ++j) s += A[k * n + j] * temp[j];
        A[k * n + k] -= s;
        for (int i = 0; i < rs; ++i) {
          double* row = &A[(k + 1 + i) * n];
          double t = 0;
          for (int j = 0; j < k; ++j) t += row[j] * temp[j];
          row[k] -= t;
        }
      }
      double akk = A[k * n + k];
      int valid = fabs(akk) > 0;
      if (k == 0 && !valid) { sign = 0; for (int j = 0; j < n; ++j) tr[j] = j; break; }
      if (rs > 0 && valid)
        for (int i = 0; i < rs; ++i) A[(k + 1 + i) * n + k] /= akk;
      if (sign == 1) { if (akk < 0) sign = 2; }
      else if (sign == -1) { if (akk > 0) sign = 2; }
      else if (sign == 0) { if (akk > 0) sign = 1; else if (akk < 0) sign = -1; }
    }
  }
  int positive = (sign == 1 || sign == 0);
  if (positive && x) {
    for (int i = 0; i < n; ++i) x[i] = b[i];
    for (int i = 0; i < n; ++i) if (tr[i] != i) { double t = x[i]; x[i] = x[tr[i]]; x[tr[i]] = t; }
    for (int i = 0; i < n; ++i) { double s = x[i]; for (int j = 0; j < i; ++j) s -= A[i * n + j] * x[j]; x[i] = s; }
    const double tol = DBL_MIN;
    for (int i = 0; i < n; ++i) { double d = A[i * n + i]; x[i] = (fabs(d) > tol) ? x[i] / d : 0.0; }
    for (int i = n - 1; i >= 0; --i) { double s = x[i]; for (int j = i + 1; j < n; ++j) s -= A[j * n + i] * x[j]; x[i] = s; }
    for (int i = n - 1; i >= 0; --i) if (tr[i] != i) { double t = x[i]; x[i] = x[tr[i]]; x[tr[i]] = t; }
  }
  free(tr); free(temp);
  return positive;
}

/* ------------------------------------------------------------------------------------------------
 * graph in g2o's order
 * ----------------------------------------------------------------------------------------------*/
typedef struct {
  int type;      /* 0 bbox 1 e3d 2 grav 3 odom */
  int src;       /* index into the input arrays of that type */
  int v0, v1;    /* vertex ids: cams 0..F-1, objs F..F+N-1 ; grav: v0 = -1 */
  int D;
  double w[9];   /* diagonal information */
  double r[9];   /* current error */
} oedge;

typedef struct {
  const esl_graph* g;
  int F, N;
  se3* cams; ell* objs;
  se3* cams_bak; ell* objs_bak;
  unsigned char* fixed;       /* per vertex */
  int* hidx;                  /* per vertex: first row of its block in H, or -1 */
  int* order; int n_free_v;   /* free vertices in Hessian order */
  int n;                      /* total free dims */
  oedge* edges; int n_edges;
  int n_dropped;
} ograph;

static int vdim(const ograph* G, int v) { return v < G->F ? 6 : 9; }

static void oedge_eval(const ograph* G, oedge* e) {
  const esl_graph* g = G->g;
  double K[4] = {g->fx, g->fy, g->cx, g->cy};
  switch (e->type) {
    case 0: res_bbox(G->cams[e->v0], G->objs[e->v1 - G->F], K, &g->bbox_meas[4 * e->src], e->r); break;
    case 1: res_e3d(G->cams[e->v0], G->objs[e->v1 - G->F], ell_from10(&g->e3d_meas[10 * e->src]), e->r); break;
    case 2: res_grav(G->objs[e->v1 - G->F], g->grav_normal, e->r); break;
    default: res_odom(G->cams[e->v0], G->cams[e->v1], se3_from7(&g->odom_meas[7 * e->src]), e->r); break;
  }
}
static double oedge_chi2(const oedge* e) { /* base_edge.h:58-61 */
  double c = 0;
  for (int k = 0; k < e->D; ++k) c += e->r[k] * (e->w[k] * e->r[k]);
  return c;
}

static void ograph_free(ograph* G) {
  free(G->cams); free(G->objs); free(G->cams_bak); free(G->objs_bak);
  free(G->fixed); free(G->hidx); free(G->order); free(G->edges);
}

static int ograph_build(ograph* G, const esl_graph* g, const double* cams, const double* objs, int drop_nan) {
  memset(G, 0, sizeof(*G));
  G->g = g; G->F = g->n_cams; G->N = g->n_objs;
  int F = G->F, N = G->N, V = F + N;
  G->cams = (se3*)malloc(sizeof(se3) * (size_t)(F + 1));
  G->objs = (ell*)malloc(sizeof(ell) * (size_t)(N + 1));
  G->cams_bak = (se3*)malloc(sizeof(se3) * (size_t)(F + 1));
  G->objs_bak = (ell*)malloc(sizeof(ell) * (size_t)(N + 1));
  G->fixed = (unsigned char*)calloc((size_t)(V + 1), 1);
  G->hidx = (int*)malloc(sizeof(int) * (size_t)(V + 1));
  G->order = (int*)malloc(sizeof(int) * (size_t)(V + 1));
  for (int i = 0; i < F; ++i) { G->cams[i] = se3_from7(&cams[7 * i]); G->fixed[i] = g->cam_fixed ? g->cam_fixed[i] : 1; }
  for (int i = 0; i < N; ++i) G->objs[i] = ell_from10(&objs[10 * i]);
  int cap = g->n_bbox + g->n_e3d + g->n_grav + g->n_odom + 1;
  G->edges = (oedge*)calloc((size_t)cap, sizeof(oedge));
  int ne = 0;
  /* addEdge order: odometry inside the camera loop (Optimizer.cpp:142-158) ... */
  for (int i = 0; i < g->n_odom; ++i) {
    oedge* e = &G->edges[ne++];
    e->type = 3; e->src = i; e->v0 = g->odom_i[i]; e->v1 = g->odom_j[i]; e->D = 6;
    for (int k = 0; k < 6; ++k) e->w[k] = g->odom_info ? g->odom_info[6 * i + k] : 1.0;
  }
  /* ... then per ellipsoid: gravity prior, bbox edges (Optimizer.cpp:166-245); bucket by object, stable */
  int* cnt = (int*)calloc((size_t)(N + 2), sizeof(int));
  int* bb_sorted = (int*)malloc(sizeof(int) * (size_t)(g->n_bbox + 1));
  for (int i = 0; i < g->n_bbox; ++i) cnt[g->bbox_obj[i] + 1]++;
  for (int i = 0; i < N; ++i) cnt[i + 1] += cnt[i];
  {
    int* pos = (int*)malloc(sizeof(int) * (size_t)(N + 1));
    for (int i = 0; i < N; ++i) pos[i] = cnt[i];
    for (int i = 0; i < g->n_bbox; ++i) bb_sorted[pos[g->bbox_obj[i]]++] = i;
    free(pos);
  }
  double K[4] = {g->fx, g->fy, g->cx, g->cy};
  for (int o = 0; o < N; ++o) {
    for (int i = 0; i < g->n_grav; ++i)
      if (g->grav_obj[i] == o) {
        oedge* e = &G->edges[ne++];
        e->type = 2; e->src = i; e->v0 = -1; e->v1 = F + o; e->D = 1; e->w[0] = g->grav_weight;
      }
    for (int j = cnt[o]; j < cnt[o + 1]; ++j) {
      int i = bb_sorted[j];
      oedge tmp; memset(&tmp, 0, sizeof(tmp));
      tmp.type = 0; tmp.src = i; tmp.v0 = g->bbox_cam[i]; tmp.v1 = F + o; tmp.D = 4;
      for (int k = 0; k < 4; ++k) tmp.w[k] = g->bbox_weight[i];
      if (drop_nan || g->check_visibility) { /* NaN check at graph build (Optimizer.cpp:234-243) + optional visibility test (:35-81) */
        res_bbox(G->cams[tmp.v0], G->objs[o], K, &g->bbox_meas[4 * i], tmp.r);
        double c = oedge_chi2(&tmp);
        if (isnan(c)) { G->n_dropped++; continue; }
        if (g->check_visibility && !check_visibility(G->cams[tmp.v0], G->objs[o], K, g->image_rows, g->image_cols)) { G->n_dropped++; continue; }
      }
      G->edges[ne++] = tmp;
    }
  }
  free(cnt); free(bb_sorted);
  /* ... then 3-D edges in frame order (Optimizer.cpp:249-279) */
  for (int i = 0; i < g->n_e3d; ++i) {
    oedge* e = &G->edges[ne++];
    e->type = 1; e->src = i; e->v0 = g->e3d_cam[i]; e->v1 = F + g->e3d_obj[i]; e->D = 9;
    for (int k = 0; k < 9; ++k) e->w[k] = g->e3d_weight[i];
  }
  /* active edges: not all vertices fixed (sparse_optimizer.cpp:234) */
  int na = 0;
  for (int i = 0; i < ne; ++i) {
    oedge* e = &G->edges[i];
    int allfixed = (e->v0 < 0 || G->fixed[e->v0]) && G->fixed[e->v1];
    if (!allfixed) G->edges[na++] = *e;
  }
  G->n_edges = na;
  /* active vertices = those with >= 1 active edge; index mapping sorted by id (sparse_optimizer.cpp:166-190) */
  unsigned char* active = (unsigned char*)calloc((size_t)(V + 1), 1);
  for (int i = 0; i < na; ++i) { if (G->edges[i].v0 >= 0) active[G->edges[i].v0] = 1; active[G->edges[i].v1] = 1; }
  int n = 0, nv = 0;
  for (int v = 0; v < V; ++v) {
    if (active[v] && !G->fixed[v]) { G->hidx[v] = n; G->order[nv++] = v; n += vdim(G, v); }
    else G->hidx[v] = -1;
  }
  free(active);
  G->n = n; G->n_free_v = nv;
  return 0;
}

static double ograph_errors(ograph* G) { /* computeActiveErrors + activeRobustChi2 */
  double chi = 0;
  for (int i = 0; i < G->n_edges; ++i) { oedge_eval(G, &G->edges[i]); chi += oedge_chi2(&G->edges[i]); }
  return chi;
}

/* numeric Jacobians of one edge wrt its free vertices.  A: D x d0 (vertex 0), B: D x d1 (vertex 1) */
static void oedge_linearize(ograph* G, oedge* e, double delta, double* A, double* B) {
  const esl_graph* g = G->g;
  double K[4] = {g->fx, g->fy, g->cx, g->cy};
  edge_ctx c; memset(&c, 0, sizeof(c));
  c.type = e->type; c.K = K; c.normal = g->grav_normal;
  se3 v0; memset(&v0, 0, sizeof(v0));
  se3 v1c; memset(&v1c, 0, sizeof(v1c));
  ell v1o; memset(&v1o, 0, sizeof(v1o));
  if (e->type == 0) { c.meas = &g->bbox_meas[4 * e->src]; v0 = G->cams[e->v0]; v1o = G->objs[e->v1 - G->F]; }
  else if (e->type == 1) { c.meas_e = ell_from10(&g->e3d_meas[10 * e->src]); v0 = G->cams[e->v0]; v1o = G->objs[e->v1 - G->F]; }
  else if (e->type == 2) { v1o = G->objs[e->v1 - G->F]; }
  else { c.meas_T = se3_from7(&g->odom_meas[7 * e->src]); v0 = G->cams[e->v0]; v1c = G->cams[e->v1]; }
  if (e->v0 >= 0 && !G->fixed[e->v0]) edge_numeric_jac(&c, v0, v1o, v1c, 0, delta, A, 6);
  if (!G->fixed[e->v1]) edge_numeric_jac(&c, v0, v1o, v1c, 1, delta, B, vdim(G, e->v1));
}

/* constructQuadraticForm (base_binary_edge.hpp:55-120, base_unary_edge.hpp:43-72) into dense H (n x n) */
static void accumulate(const ograph* G, const oedge* e, const double* A, const double* B, double* H, double* b) {
  const int n = G->n, D = e->D;
  int i0 = (e->v0 >= 0) ? G->hidx[e->v0] : -1, i1 = G->hidx[e->v1];
  int d0 = (e->v0 >= 0) ? vdim(G, e->v0) : 0, d1 = vdim(G, e->v1);
  double omega_r[9];
  for (int k = 0; k < D; ++k) omega_r[k] = -(e->w[k] * e->r[k]);
  if (i0 >= 0) {
    for (int a = 0; a < d0; ++a) {
      double s = 0;
      for (int k = 0; k < D; ++k) s += A[k * d0 + a] * omega_r[k];
      b[i0 + a] += s;
      for (int c = 0; c < d0; ++c) {
        double h = 0;
        for (int k = 0; k < D; ++k) h += A[k * d0 + a] * e->w[k] * A[k * d0 + c];
        H[(size_t)(i0 + a) * n + i0 + c] += h;
      }
      if (i1 >= 0)
        for (int c = 0; c < d1; ++c) {
          double h = 0;
          for (int k = 0; k < D; ++k) h += A[k * d0 + a] * e->w[k] * B[k * d1 + c];
          H[(size_t)(i0 + a) * n + i1 + c] += h;
          H[(size_t)(i1 + c) * n + i0 + a] += h;
        }
    }
  }
  if (i1 >= 0) {
    for (int a = 0; a < d1; ++a) {
      double s = 0;
      for (int k = 0; k < D; ++k) s += B[k * d1 + a] * omega_r[k];
      b[i1 + a] += s;
      for (int c = 0; c < d1; ++c) {
        double h = 0;
        for (int k = 0; k < D; ++k) h += B[k * d1 + a] * e->w[k] * B[k * d1 + c];
        H[(size_t)(i1 + a) * n + i1 + c] += h;
      }
    }
  }
}

int esl_oracle_build_system(const esl_graph* g, const double* cams, const double* objs, double delta,
                            int drop_nan_bbox, double* H, double* b, int32_t* free_index, double* chi2) {
  ograph G;
  ograph_build(&G, g, cams, objs, drop_nan_bbox);
  int n = G.n;
  if (free_index) for (int v = 0; v < G.F + G.N; ++v) free_index[v] = G.hidx[v];
  if (H) {
    memset(H, 0, sizeof(double) * (size_t)n * (size_t)n);
    memset(b, 0, sizeof(double) * (size_t)n);
    double c = ograph_errors(&G);
    if (chi2) *chi2 = c;
    double A[9 * 6], B[9 * 9];
    for (int i = 0; i < G.n_edges; ++i) {
      oedge_linearize(&G, &G.edges[i], delta, A, B);
      accumulate(&G, &G.edges[i], A, B, H, b);
    }
  }
  ograph_free(&G);
  return n;
}

/* ------------------------------------------------------------------------------------------------
 * block storage for the "improved" solver: per-vertex diagonal blocks, per-edge cam-obj blocks
 * ----------------------------------------------------------------------------------------------*/
typedef struct {
  double* Hd;   /* per free vertex in order: d x d row-major, offsets hoff */
  size_t* hoff;
  double* b;    /* n */
  double* W;    /* per binary cam-obj edge with both free: 6 x 9 */
  double* Hcc;  /* odometry cam-cam blocks per odom edge: 6 x 6 (i,j) */
} blocksys;

/* solve mapping-mode block-diagonal system: per ellipsoid pivoted LDLT (bit-identical to the dense
 * pivoted LDLT of a block-diagonal matrix) */
static int solve_blockdiag(const ograph* G, const double* Hdense_unused, const blocksys* S, double lambda,
                           double* x) {
  (void)Hdense_unused;
  int ok = 1;
  for (int k = 0; k < G->n_free_v; ++k) {
    int v = G->order[k], d = vdim(G, v), base = G->hidx[v];
    double M[81];
    const double* Hb = S->Hd + S->hoff[k];
    for (int i = 0; i < d * d; ++i) M[i] = Hb[i];
    for (int i = 0; i < d; ++i) M[i * d + i] += lambda;
    if (!esl_oracle_ldlt_solve(M, d, &S->b[base], &x[base])) ok = 0;
  }
  return ok;
}

/* ------------------------------------------------------------------------------------------------
 * Levenberg-Marquardt (optimization_algorithm_levenberg.cpp:61-164, sparse_optimizer.cpp:354-419)
 * ----------------------------------------------------------------------------------------------*/
static void apply_update(ograph* G, const double* x) { /* SparseOptimizer::update (sparse_optimizer.cpp:422-435) */
  for (int k = 0; k < G->n_free_v; ++k) {
    int v = G->order[k];
    const double* u = &x[G->hidx[v]];
    if (v < G->F) G->cams[v] = cam_oplus(G->cams[v], u);
    else G->objs[v - G->F] = ell_exp_update(G->objs[v - G->F], u);
  }
}

/* Schur solve for SLAM mode ("improved" CPU path; block_solver.hpp:367-486 arithmetic):
 * eliminate ellipsoid blocks, dense pivoted LDLT on the reduced camera system, back-substitute. */
static int solve_schur(const ograph* G, const double* H, const double* b, double lambda, double* x) {
  const int n = G->n;
  int np = 0; /* camera dims come first in the ordering */
  for (int k = 0; k < G->n_free_v; ++k) if (G->order[k] < G->F) np += 6;
  const int nl = n - np;
  double* S = (double*)malloc(sizeof(double) * (size_t)(np > 0 ? np : 1) * (size_t)(np > 0 ? np : 1));
  double* bs = (double*)malloc(sizeof(double) * (size_t)(np + 1));
  for (int i = 0; i < np; ++i) {
    for (int j = 0; j < np; ++j) S[(size_t)i * np + j] = H[(size_t)i * n + j];
    S[(size_t)i * np + i] += lambda;
    bs[i] = b[i];
  }
  double* Dinv = (double*)malloc(sizeof(double) * 81 * (size_t)(nl / 9 + 1));
  int ok = 1;
  for (int l = 0; l < nl / 9; ++l) {
    int base = np + 9 * l;
    /* Dinv = (D + lambda I)^-1 via 9 solves with the pivoted LDLT */
    double M[81], e[9], col[9];
    for (int c = 0; c < 9; ++c) {
      for (int i = 0; i < 9; ++i) for (int j = 0; j < 9; ++j) M[i * 9 + j] = H[(size_t)(base + i) * n + base + j];
      for (int i = 0; i < 9; ++i) M[i * 9 + i] += lambda;
      for (int i = 0; i < 9; ++i) e[i] = (i == c) ? 1.0 : 0.0;
      if (!esl_oracle_ldlt_solve(M, 9, e, col)) ok = 0;
      for (int i = 0; i < 9; ++i) Dinv[l * 81 + i * 9 + c] = col[i];
    }
    /* cameras touching this landmark */
    double db[9];
    for (int i = 0; i < 9; ++i) { double s = 0; for (int j = 0; j < 9; ++j) s += Dinv[l * 81 + i * 9 + j] * b[base + j]; db[i] = s; }
    for (int ci = 0; ci < np; ci += 6) {
      int nz = 0;
      for (int i = 0; i < 6 && !nz; ++i) for (int j = 0; j < 9; ++j) if (H[(size_t)(ci + i) * n + base + j] != 0.0) { nz = 1; break; }
      if (!nz) continue;
      double BD[54];
      for (int i = 0; i < 6; ++i) for (int j = 0; j < 9; ++j) {
        double s = 0;
        for (int k = 0; k < 9; ++k) s += H[(size_t)(ci + i) * n + base + k] * Dinv[l * 81 + k * 9 + j];
        BD[i * 9 + j] = s;
      }
      for (int i = 0; i < 6; ++i) { double s = 0; for (int j = 0; j < 9; ++j) s += H[(size_t)(ci + i) * n + base + j] * db[j]; bs[ci + i] -= s; }
      for (int cj = 0; cj < np; cj += 6) {
        int nz2 = 0;
        for (int i = 0; i < 6 && !nz2; ++i) for (int j = 0; j < 9; ++j) if (H[(size_t)(cj + i) * n + base + j] != 0.0) { nz2 = 1; break; }
        if (!nz2) continue;
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) {
          double s = 0;
          for (int k = 0; k < 9; ++k) s += BD[i * 9 + k] * H[(size_t)(cj + j) * n + base + k];
          S[(size_t)(ci + i) * np + cj + j] -= s;
        }
      }
    }
  }
  if (np > 0) { if (!esl_oracle_ldlt_solve(S, np, bs, x)) ok = 0; }
  for (int l = 0; l < nl / 9 && ok; ++l) {
    int base = np + 9 * l;
    double c[9];
    for (int i = 0; i < 9; ++i) {
      double s = b[base + i];
      for (int j = 0; j < np; ++j) s -= H[(size_t)(base + i) * n + j] * x[j];
      c[i] = s;
    }
    for (int i = 0; i < 9; ++i) { double s = 0; for (int j = 0; j < 9; ++j) s += Dinv[l * 81 + i * 9 + j] * c[j]; x[base + i] = s; }
  }
  free(S); free(bs); free(Dinv);
  return ok;
}

/* Camera-first elimination for SLAM mode: the SAME damped system (H + lambda I) x = b with the OTHER block marginalised
 * (block_solver.hpp:367-486 eliminates whichever vertices carry the marginalised flag; x does not depend on the choice).
 * In the reference's SLAM branch the camera block is block TRIDIAGONAL -- odometry edges join consecutive frames only
 * (Optimizer.cpp:142-158) -- so A = Hpp + lambda I = L L^T with L block bidiagonal, Y = L^-1 [Hpl | b_p] by a forward
 * recurrence, T = Hll + lambda I - Y^T Y (order 9 N), x_l from the pivoted LDLT of T, x_p = L^-T (y - Y x_l).
 * Written as the plain chain (no dissection, Y dense): the independent CPU check of csrc/esl_cf.hpp's nested-dissection /
 * sparse-X form, and the measured CPU time of the GPU's own elimination order in bench.py.
 * Returns 1 ok, 0 not positive, -1 not applicable (camera block not block tridiagonal). */
static void chol6(const double* A, double* L, int* ok) { /* lower Cholesky of a 6 x 6 row-major block */
  for (int i = 0; i < 36; ++i) L[i] = 0;
  for (int j = 0; j < 6; ++j) {
    double d = A[j * 6 + j];
    for (int k = 0; k < j; ++k) d -= L[j * 6 + k] * L[j * 6 + k];
    if (!(d > 0)) { *ok = 0; d = 1; }
    double r = sqrt(d);
    L[j * 6 + j] = r;
    for (int i = j + 1; i < 6; ++i) {
      double v = A[i * 6 + j];
      for (int k = 0; k < j; ++k) v -= L[i * 6 + k] * L[j * 6 + k];
      L[i * 6 + j] = v / r;
    }
  }
}
static void fwd6(const double* L, double* v) { /* v <- L^-1 v */
  for (int i = 0; i < 6; ++i) { double s = v[i]; for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * v[k]; v[i] = s / L[i * 6 + i]; }
}
static void bwd6(const double* L, double* v) { /* v <- L^-T v */
  for (int i = 5; i >= 0; --i) { double s = v[i]; for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * v[k]; v[i] = s / L[i * 6 + i]; }
}
__attribute__((optimize("O3")))
static int solve_camfirst(const ograph* G, const double* H, const double* b, double lambda, double* x) {
  const int n = G->n;
  int np = 0;
  for (int k = 0; k < G->n_free_v; ++k) if (G->order[k] < G->F) np += 6;
  const int nl = n - np, nc = np / 6, m = nl + 1;
  if (nc == 0 || nl == 0) return -1;
  double tm0 = now_s();
  for (int i = 0; i < nc; ++i)          /* band check: blocks (i, j), j < i - 1, must vanish */
    for (int j = 0; j + 1 < i; ++j)
      for (int a = 0; a < 6; ++a) for (int c = 0; c < 6; ++c)
        if (H[(size_t)(6 * i + a) * n + 6 * j + c] != 0.0) return -1;
  int ok = 1;
  double* Ld = (double*)malloc(sizeof(double) * 36 * (size_t)nc);   /* L_ii */
  double* Lo = (double*)calloc(36 * (size_t)nc, sizeof(double));    /* L_{i,i-1} */
  double* Y = (double*)malloc(sizeof(double) * (size_t)np * (size_t)m);
  for (int i = 0; i < nc; ++i) {
    double A[36];
    for (int a = 0; a < 6; ++a) for (int c = 0; c < 6; ++c) A[a * 6 + c] = H[(size_t)(6 * i + a) * n + 6 * i + c] + (a == c ? lambda : 0.0);
    if (i > 0) {   /* L_{i,i-1} = A_{i,i-1} L_{i-1,i-1}^-T: row a of it solves L_{i-1,i-1} w = (row a of A_{i,i-1})^T */
      for (int a = 0; a < 6; ++a) {
        double w[6];
        for (int c = 0; c < 6; ++c) w[c] = H[(size_t)(6 * i + a) * n + 6 * (i - 1) + c];
        fwd6(&Ld[36 * (size_t)(i - 1)], w);
        for (int c = 0; c < 6; ++c) Lo[36 * (size_t)i + a * 6 + c] = w[c];
      }
      for (int a = 0; a < 6; ++a) for (int c = 0; c < 6; ++c) {
        double sum = 0;
        for (int k = 0; k < 6; ++k) sum += Lo[36 * (size_t)i + a * 6 + k] * Lo[36 * (size_t)i + c * 6 + k];
        A[a * 6 + c] -= sum;
      }
    }
    chol6(A, &Ld[36 * (size_t)i], &ok);
    /* Y_i = L_ii^-1 (R_i - L_{i,i-1} Y_{i-1}), R = [Hpl | b_p] */
    double* Yi = Y + (size_t)(6 * i) * m;
    for (int a = 0; a < 6; ++a) {
      for (int j = 0; j < nl; ++j) Yi[(size_t)a * m + j] = H[(size_t)(6 * i + a) * n + np + j];
      Yi[(size_t)a * m + nl] = b[6 * i + a];
    }
    if (i > 0) {
      const double* Yp = Y + (size_t)(6 * (i - 1)) * m;
      for (int a = 0; a < 6; ++a) for (int k = 0; k < 6; ++k) {
        const double l = Lo[36 * (size_t)i + a * 6 + k];
        if (l == 0.0) continue;
        for (int j = 0; j < m; ++j) Yi[(size_t)a * m + j] -= l * Yp[(size_t)k * m + j];
      }
    }
    const double* L = &Ld[36 * (size_t)i];
    for (int a = 0; a < 6; ++a) {
      for (int k = 0; k < a; ++k) { const double l = L[a * 6 + k]; for (int j = 0; j < m; ++j) Yi[(size_t)a * m + j] -= l * Yi[(size_t)k * m + j]; }
      const double d = L[a * 6 + a];
      for (int j = 0; j < m; ++j) Yi[(size_t)a * m + j] /= d;
    }
  }
  g_cf_timing[0] += now_s() - tm0; tm0 = now_s();
  /* T = Hll + lambda I - Y^T Y (lower triangle, then mirrored), t = b_l - Y^T y */
  double* T = (double*)malloc(sizeof(double) * (size_t)nl * (size_t)nl);
  double* t = (double*)malloc(sizeof(double) * (size_t)nl);
  for (int i = 0; i < nl; ++i) {
    for (int j = 0; j <= i; ++j) T[(size_t)i * nl + j] = H[(size_t)(np + i) * n + np + j];
    T[(size_t)i * nl + i] += lambda;
    t[i] = b[np + i];
  }
  for (int k = 0; k < np; ++k) {
    const double* yk = Y + (size_t)k * m;
    const double yy = yk[nl];
    for (int i = 0; i < nl; ++i) {
      const double a = yk[i];
      if (a == 0.0) continue;
      double* Ti = T + (size_t)i * nl;
      for (int j = 0; j <= i; ++j) Ti[j] -= a * yk[j];
      t[i] -= a * yy;
    }
  }
  for (int i = 0; i < nl; ++i) for (int j = i + 1; j < nl; ++j) T[(size_t)i * nl + j] = T[(size_t)j * nl + i];
  g_cf_timing[1] += now_s() - tm0; tm0 = now_s();
  if (!esl_oracle_ldlt_solve(T, nl, t, &x[np])) ok = 0;
  g_cf_timing[2] += now_s() - tm0; tm0 = now_s();
  /* z = y - Y x_l, x_p = L^-T z (backward along the chain) */
  for (int k = 0; k < np; ++k) {
    const double* yk = Y + (size_t)k * m;
    double sum = yk[nl];
    for (int j = 0; j < nl; ++j) sum -= yk[j] * x[np + j];
    x[k] = sum;
  }
  for (int i = nc - 1; i >= 0; --i) {
    double* xi = &x[6 * i];
    if (i + 1 < nc) for (int a = 0; a < 6; ++a) { double sum = 0; for (int k = 0; k < 6; ++k) sum += Lo[36 * (size_t)(i + 1) + k * 6 + a] * x[6 * (i + 1) + k]; xi[a] -= sum; }
    bwd6(&Ld[36 * (size_t)i], xi);
  }
  g_cf_timing[3] += now_s() - tm0;
  free(Ld); free(Lo); free(Y); free(T); free(t);
  return ok;
}

int esl_oracle_optimize(const esl_graph* g, double* cams_io, double* objs_io, const esl_lm_params* p,
                        int solver, esl_lm_report* out) {
  ograph G;
  g_bbox_residual = p->bbox_residual;   /* stays set: esl_oracle_res_bbox / jac_bbox / build_system follow the last run's mode */
  ograph_build(&G, g, cams_io, objs_io, p->drop_nan_bbox);
  memset(out, 0, sizeof(*out));
  g_timing[0] = g_timing[1] = g_timing[2] = 0;
  g_cf_timing[0] = g_cf_timing[1] = g_cf_timing[2] = g_cf_timing[3] = 0;
  int nb = 0;
  for (int i = 0; i < G.n_edges; ++i) if (G.edges[i].type == 0) nb++;
  out->n_bbox_valid = nb; out->n_bbox_dropped = G.n_dropped;
  const int n = G.n;
  if (n == 0 || G.n_edges == 0) { out->stop_reason = 3; ograph_free(&G); return 0; }
  int all_cams_fixed = 1;
  for (int k = 0; k < G.n_free_v; ++k) if (G.order[k] < G.F) all_cams_fixed = 0;
  const int use_blockdiag = (solver == ESL_ORACLE_BLOCK) && all_cams_fixed;

  double* H = NULL; double* Hwork = NULL;
  blocksys S; memset(&S, 0, sizeof(S));
  double* b = (double*)calloc((size_t)n, sizeof(double));
  double* x = (double*)calloc((size_t)n, sizeof(double));
  if (use_blockdiag) {
    S.hoff = (size_t*)malloc(sizeof(size_t) * (size_t)G.n_free_v);
    size_t off = 0;
    for (int k = 0; k < G.n_free_v; ++k) { S.hoff[k] = off; int d = vdim(&G, G.order[k]); off += (size_t)d * d; }
    S.Hd = (double*)calloc(off, sizeof(double));
    S.b = b;
  } else {
    H = (double*)malloc(sizeof(double) * (size_t)n * (size_t)n);
    if (solver == ESL_ORACLE_DENSE) Hwork = (double*)malloc(sizeof(double) * (size_t)n * (size_t)n);
  }
  /* map vertex -> position in order (for block storage) */
  int* vpos = (int*)malloc(sizeof(int) * (size_t)(G.F + G.N + 1));
  for (int v = 0; v < G.F + G.N; ++v) vpos[v] = -1;
  for (int k = 0; k < G.n_free_v; ++k) vpos[G.order[k]] = k;

  double lambda = -1, ni = 2;
  int nBad = 0;
  int it = 0, ok_outer = 1, total_trials = 0;
  double currentChi = 0;
  out->stop_reason = 0;
  for (it = 0; it < p->max_iters && ok_outer; ++it) {
    double t0 = now_s();
    currentChi = ograph_errors(&G);
    g_timing[2] += now_s() - t0;
    double tempChi = currentChi, iniChi = currentChi;
    if (it == 0) out->chi2_initial = currentChi;
    /* buildSystem (block_solver.hpp:502-560) */
    t0 = now_s();
    memset(b, 0, sizeof(double) * (size_t)n);
    if (use_blockdiag) {
      size_t tot = S.hoff[G.n_free_v - 1] + 81;
      memset(S.Hd, 0, sizeof(double) * tot);
    } else memset(H, 0, sizeof(double) * (size_t)n * (size_t)n);
    {
      double A[54], B[81];
      for (int i = 0; i < G.n_edges; ++i) {
        oedge* e = &G.edges[i];
        oedge_linearize(&G, e, p->numeric_delta, A, B);
        if (use_blockdiag) {
          /* only vertex 1 (the ellipsoid) is free */
          int k = vpos[e->v1], D = e->D;
          double* Hb = S.Hd + S.hoff[k];
          int base = G.hidx[e->v1];
          for (int a = 0; a < 9; ++a) {
            double s = 0;
            for (int q = 0; q < D; ++q) s += B[q * 9 + a] * (-(e->w[q] * e->r[q]));
            b[base + a] += s;
            for (int c = 0; c < 9; ++c) {
              double h = 0;
              for (int q = 0; q < D; ++q) h += B[q * 9 + a] * e->w[q] * B[q * 9 + c];
              Hb[a * 9 + c] += h;
            }
          }
        } else accumulate(&G, e, A, B, H, b);
      }
    }
    g_timing[0] += now_s() - t0;
    if (it == 0) { /* computeLambdaInit (:166-180) */
      double maxDiag = 0;
      if (use_blockdiag) {
        for (int k = 0; k < G.n_free_v; ++k) for (int j = 0; j < 9; ++j) {
          double v = fabs(S.Hd[S.hoff[k] + (size_t)j * 9 + j]); if (v > maxDiag) maxDiag = v; }
      } else for (int j = 0; j < n; ++j) { double v = fabs(H[(size_t)j * n + j]); if (v > maxDiag) maxDiag = v; }
      lambda = p->tau * maxDiag; ni = 2; nBad = 0;
    }
    double rho = 0;
    int qmax = 0;
    do {
      memcpy(G.cams_bak, G.cams, sizeof(se3) * (size_t)G.F); /* _optimizer->push() */
      memcpy(G.objs_bak, G.objs, sizeof(ell) * (size_t)G.N);
      t0 = now_s();
      int ok2;
      if (use_blockdiag) ok2 = solve_blockdiag(&G, NULL, &S, lambda, x);
      else if (solver == ESL_ORACLE_BLOCK) ok2 = solve_schur(&G, H, b, lambda, x);
      else if (solver == ESL_ORACLE_CAMFIRST) {
        ok2 = solve_camfirst(&G, H, b, lambda, x);
        if (ok2 < 0) ok2 = solve_schur(&G, H, b, lambda, x);   /* camera block not a chain: the other elimination */
      }
      else {
        memcpy(Hwork, H, sizeof(double) * (size_t)n * (size_t)n);
        for (int j = 0; j < n; ++j) Hwork[(size_t)j * n + j] += lambda; /* setLambda */
        ok2 = esl_oracle_ldlt_solve(Hwork, n, b, x);
      }
      g_timing[1] += now_s() - t0;
      /* NB: when the factorisation fails g2o leaves _x from the previous solve; the step is rejected anyway */
      apply_update(&G, x);
      t0 = now_s();
      tempChi = ograph_errors(&G);
      g_timing[2] += now_s() - t0;
      if (!ok2) tempChi = DBL_MAX;
      rho = currentChi - tempChi;
      double scale = 0; /* computeScale (:182-189) */
      for (int j = 0; j < n; ++j) scale += x[j] * (lambda * x[j] + b[j]);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - pow((2 * rho - 1), 3);
        alpha = alpha < (2. / 3.) ? alpha : (2. / 3.);
        double scaleFactor = (1. / 3.) > alpha ? (1. / 3.) : alpha;
        lambda *= scaleFactor;
        ni = 2;
        currentChi = tempChi;
      } else {
        lambda *= ni;
        ni *= 2;
        memcpy(G.cams, G.cams_bak, sizeof(se3) * (size_t)G.F); /* pop */
        memcpy(G.objs, G.objs_bak, sizeof(ell) * (size_t)G.N);
      }
      qmax++;
    } while (rho < 0 && qmax < p->max_trials);
    total_trials += qmax;
    if (it < ESL_MAX_TRACE) { out->trace_chi2[it] = currentChi; out->trace_lambda[it] = lambda; out->trace_trials[it] = qmax; out->trace_len = it + 1; }
    if (qmax == p->max_trials || rho == 0) { ok_outer = 0; out->stop_reason = 1; }
    else {
      if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
      if (nBad >= 3) { ok_outer = 0; out->stop_reason = 2; }
    }
  }
  out->iterations = it;
  out->total_trials = total_trials;
  out->chi2_final = currentChi;
  out->lambda_final = lambda;
  for (int i = 0; i < G.F; ++i) se3_to7(G.cams[i], &cams_io[7 * i]);
  for (int i = 0; i < G.N; ++i) ell_to10(G.objs[i], &objs_io[10 * i]);
  free(H); free(Hwork); free(S.Hd); free(S.hoff); free(b); free(x); free(vpos);
  ograph_free(&G);
  return 0;
}
