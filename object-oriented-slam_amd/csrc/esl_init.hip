// esl_init.hip — esl_init_quadric: quadric from >= 9 bbox tangent planes (SVD null vector), one
// wavefront per object.  Replaces EllipsoidSLAM::Initializer::initializeQuadric
// (reference src/core/Initializer.cpp:24-56) = getPlanesHomo (:58-91), fromDetectionsToLines (:107-145),
// getVectorFromPlanesHomo (:147-164), getQStarFromVectors (:166-184), getEllipsoidFromQStar (:186-248).
//
// The plane rows (<= 4 per observation, 10 monomials each) are staged in LDS; the null vector comes
// from a one-sided (Hestenes) Jacobi SVD whose column dot-products are lane-strided sums finished with
// a butterfly all-reduce; the 4x4 / 3x3 decompositions that follow are a few hundred flops on lane 0.
#include <cmath>

#include "esl_ctx.hpp"
#include "esl_kernels_map.hpp"

namespace esl {

__device__ __forceinline__ double wave_allsum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// cyclic Jacobi eigen-solver, symmetric N x N row-major; eigenvalues ascending, eigenvectors in columns,
// each column signed so that its largest-magnitude component is positive
template <int N>
__device__ void sym_eig(const double* Ain, double* w, double* V) {
  double A[N * N];
  for (int i = 0; i < N * N; ++i) A[i] = Ain[i];
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) V[i * N + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
    for (int p = 0; p < N; ++p)
      for (int q = p + 1; q < N; ++q) off += A[p * N + q] * A[p * N + q];
    if (off < 1e-300) break;
    for (int p = 0; p < N; ++p)
      for (int q = p + 1; q < N; ++q) {
        const double apq = A[p * N + q];
        if (apq == 0.0) continue;
        const double theta = (A[q * N + q] - A[p * N + p]) / (2 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
        const double c = 1 / sqrt(t * t + 1), s = t * c;
        for (int k = 0; k < N; ++k) { const double a = A[k * N + p], b = A[k * N + q]; A[k * N + p] = c * a - s * b; A[k * N + q] = s * a + c * b; }
        for (int k = 0; k < N; ++k) { const double a = A[p * N + k], b = A[q * N + k]; A[p * N + k] = c * a - s * b; A[q * N + k] = s * a + c * b; }
        for (int k = 0; k < N; ++k) { const double a = V[k * N + p], b = V[k * N + q]; V[k * N + p] = c * a - s * b; V[k * N + q] = s * a + c * b; }
      }
  }
  for (int i = 0; i < N; ++i) w[i] = A[i * N + i];
  for (int i = 0; i < N; ++i) {
    int m = i;
    for (int j = i + 1; j < N; ++j)
      if (w[j] < w[m]) m = j;
    if (m != i) {
      const double t = w[i]; w[i] = w[m]; w[m] = t;
      for (int k = 0; k < N; ++k) { const double u = V[k * N + i]; V[k * N + i] = V[k * N + m]; V[k * N + m] = u; }
    }
  }
  for (int j = 0; j < N; ++j) {
    int m = 0;
    for (int k = 1; k < N; ++k)
      if (fabs(V[k * N + j]) > fabs(V[m * N + j])) m = k;
    if (V[m * N + j] < 0)
      for (int k = 0; k < N; ++k) V[k * N + j] = -V[k * N + j];
  }
}

// Gauss-Jordan inverse with partial pivoting; returns the determinant
template <int N>
__device__ double inv_det(const double* Ain, double* inv) {
  double A[N * N], det = 1;
  for (int i = 0; i < N * N; ++i) A[i] = Ain[i];
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) inv[i * N + j] = (i == j) ? 1.0 : 0.0;
  for (int c = 0; c < N; ++c) {
    int p = c;
    for (int r = c + 1; r < N; ++r)
      if (fabs(A[r * N + c]) > fabs(A[p * N + c])) p = r;
    if (p != c) {
      for (int k = 0; k < N; ++k) {
        double t = A[c * N + k]; A[c * N + k] = A[p * N + k]; A[p * N + k] = t;
        t = inv[c * N + k]; inv[c * N + k] = inv[p * N + k]; inv[p * N + k] = t;
      }
      det = -det;
    }
    const double d = A[c * N + c];
    det *= d;
    for (int k = 0; k < N; ++k) { A[c * N + k] /= d; inv[c * N + k] /= d; }
    for (int r = 0; r < N; ++r)
      if (r != c) {
        const double f = A[r * N + c];
        for (int k = 0; k < N; ++k) { A[r * N + k] -= f * A[c * N + k]; inv[r * N + k] -= f * inv[c * N + k]; }
      }
  }
  return det;
}

// mode 0: initializeQuadric; 1: getEllipsoidFromQStar of a given Q* (extra[0..15]); 2: quadricErrorWithPlanes of a given
// ellipsoid (extra[0..9]).  a_glob != null: the plane matrix lives in global memory (more observations than LDS holds).
struct InitArgs { double K[4]; int n, rows, cols, faithful, mode; double extra[16]; double* a_glob; };

__device__ void decompose_qstar(const double* Qs, int faithful, double* __restrict__ out);

// out: [0..9] ellipsoid 10-vector, [10..25] Q* row-major, [26] ok, [27] plane error (mode 2)
static __global__ __launch_bounds__(64) void k_init_quadric(const double* __restrict__ poses, const double* __restrict__ boxes,
                                                            InitArgs a, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* V = sm;                                   // 10 x 10
  double* A = a.a_glob ? a.a_glob : sm + 100;       // m x 10
  const int lane = threadIdx.x;
  if (a.mode == 1) {
    if (lane == 0) {
      for (int i = 0; i < 16; ++i) out[10 + i] = a.extra[i];
      for (int i = 0; i < 10; ++i) out[i] = 0;
      out[26] = 0; out[27] = 0;
      decompose_qstar(a.extra, a.faithful, out);
    }
    return;
  }
  int m = 0;  // rows so far (wave-uniform)
  for (int base = 0; base < a.n; base += 64) {
    const int i = base + lane;
    double rowsv[4][10];
    int cnt = 0;
    if (i < a.n) {
      const double* d = boxes + 4 * i;
      if (!(d[0] < 1 && d[1] < 1 && d[2] < 1 && d[3] < 1)) {
        SE3 Twc = se3_load(poses + 7 * i);
        Twc.r = q_normalize_pos(Twc.r);  // SE3Quat(Vector7d) ctor normalises (se3quat.h:66-69)
        const SE3 Tcw = se3_inv(Twc);
        const Mat3 R = q_to_R(Tcw.r);
        double P[12];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const double c0 = c < 3 ? R.m[c] : Tcw.t[0], c1 = c < 3 ? R.m[3 + c] : Tcw.t[1], c2 = c < 3 ? R.m[6 + c] : Tcw.t[2];
          P[c] = a.K[0] * c0 + a.K[2] * c2; P[4 + c] = a.K[1] * c1 + a.K[3] * c2; P[8 + c] = c2;
        }
        const double lines[4][3] = {{1, 0, -d[0]}, {0, 1, -d[1]}, {1, 0, -d[2]}, {0, 1, -d[3]}};
        const bool keep[4] = {d[0] > 0 && d[0] < a.cols - 1, d[1] > 0 && d[1] < a.rows - 1, d[2] > 0 && d[2] < a.cols - 1,
                              d[3] > 0 && d[3] < a.rows - 1};
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          if (!keep[l]) continue;
          double p[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) p[c] = P[c] * lines[l][0] + P[4 + c] * lines[l][1] + P[8 + c] * lines[l][2];
          double* v = rowsv[cnt++];
          v[0] = p[0] * p[0]; v[1] = 2 * p[0] * p[1]; v[2] = 2 * p[0] * p[2]; v[3] = 2 * p[0] * p[3]; v[4] = p[1] * p[1];
          v[5] = 2 * p[1] * p[2]; v[6] = 2 * p[1] * p[3]; v[7] = p[2] * p[2]; v[8] = 2 * p[2] * p[3]; v[9] = p[3] * p[3];
        }
      }
    }
    // exclusive prefix sum of cnt over the wave -> compacted row offsets (keeps observation order)
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    const int excl = incl - cnt;
    for (int k = 0; k < cnt; ++k)
      for (int c = 0; c < 10; ++c) A[(size_t)(m + excl + k) * 10 + c] = rowsv[k][c];
    m += __shfl(incl, 63, 64);
  }
  for (int idx = lane; idx < 100; idx += 64) V[idx] = (idx / 10 == idx % 10) ? 1.0 : 0.0;
  __syncthreads();
  if (a.mode == 2) {   // quadricErrorWithPlanes (Initializer.cpp:271-284): sum over the planes of (v(pi) . q_hat)^2
    SE3 T = se3_load(a.extra);
    const Mat3 R = q_to_R(T.r);
    const double d[4] = {a.extra[7] * a.extra[7], a.extra[8] * a.extra[8], a.extra[9] * a.extra[9], -1.0};
    double Z[16], Qs[16];   // Q* = Z diag(a^2, b^2, c^2, -1) Z^T (Ellipsoid.cpp:290-300)
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) Z[i * 4 + j] = R.m[i * 3 + j]; Z[i * 4 + 3] = T.t[i]; Z[12 + i] = 0; }
    Z[15] = 1;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) { double v = 0; for (int k = 0; k < 4; ++k) v += Z[i * 4 + k] * d[k] * Z[j * 4 + k]; Qs[i * 4 + j] = v; }
    const double qh[10] = {Qs[0], Qs[1], Qs[2], Qs[3], Qs[5], Qs[6], Qs[7], Qs[10], Qs[11], Qs[15]};
    double err = 0;
    for (int r = lane; r < m; r += 64) {
      double v = 0;
      for (int k = 0; k < 10; ++k) v += A[(size_t)r * 10 + k] * qh[k];
      err += v * v;
    }
    err = wave_allsum(err);
    if (lane < 27) out[lane] = 0.0;
    if (lane == 0) { out[27] = err; out[26] = 1.0; }
    return;
  }
  if (m < 9) {  // at least 9 planes are needed (Initializer.cpp:38)
    if (lane < 28) out[lane] = 0.0;
    return;
  }
  // one-sided Jacobi SVD on the columns of A
  for (int sweep = 0; sweep < 60; ++sweep) {
    int rotated = 0;
    for (int p = 0; p < 10; ++p)
      for (int q = p + 1; q < 10; ++q) {
        double al = 0, be = 0, ga = 0;
        for (int r = lane; r < m; r += 64) {
          const double ap = A[(size_t)r * 10 + p], aq = A[(size_t)r * 10 + q];
          al += ap * ap; be += aq * aq; ga += ap * aq;
        }
        al = wave_allsum(al); be = wave_allsum(be); ga = wave_allsum(ga);
        if (fabs(ga) <= 1e-15 * sqrt(al * be) || ga == 0.0) continue;
        rotated = 1;
        const double zeta = (be - al) / (2 * ga);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
        const double cs = 1 / sqrt(1 + t * t), sn = cs * t;
        for (int r = lane; r < m; r += 64) {
          const double ap = A[(size_t)r * 10 + p], aq = A[(size_t)r * 10 + q];
          A[(size_t)r * 10 + p] = cs * ap - sn * aq;
          A[(size_t)r * 10 + q] = sn * ap + cs * aq;
        }
        if (lane < 10) {
          const double vp = V[lane * 10 + p], vq = V[lane * 10 + q];
          V[lane * 10 + p] = cs * vp - sn * vq;
          V[lane * 10 + q] = sn * vp + cs * vq;
        }
        __syncthreads();
      }
    if (!rotated) break;
  }
  int best = 0;
  double bn = -1;
  for (int c = 0; c < 10; ++c) {
    double s = 0;
    for (int r = lane; r < m; r += 64) s += A[(size_t)r * 10 + c] * A[(size_t)r * 10 + c];
    s = wave_allsum(s);
    if (bn < 0 || s < bn) { bn = s; best = c; }
  }
  if (lane != 0) return;
  double q[10];
  for (int r = 0; r < 10; ++r) q[r] = V[r * 10 + best];
  if (q[9] < 0)
    for (int r = 0; r < 10; ++r) q[r] = -q[r];  // sign convention: Q*_33 > 0
  const double Qs[16] = {q[0], q[1], q[2], q[3], q[1], q[4], q[5], q[6], q[2], q[5], q[7], q[8], q[3], q[6], q[8], q[9]};
  for (int i = 0; i < 16; ++i) out[10 + i] = Qs[i];
  for (int i = 0; i < 10; ++i) out[i] = 0;
  out[26] = 0; out[27] = 0;
  decompose_qstar(Qs, a.faithful, out);
}

// getEllipsoidFromQStar (Initializer.cpp:186-248); faithful = 0: the exact decomposition through Q_33
__device__ void decompose_qstar(const double* Qs, int faithful, double* __restrict__ out) {
  const double t[3] = {Qs[3] / Qs[15], Qs[7] / Qs[15], Qs[11] / Qs[15]};
  double s[3], Rm[9];
  if (faithful) {
    double Qi[16], Q[16], w4[4], V4[16], tmp[16];
    const double cb = cbrt(inv_det<4>(Qs, Qi));
    for (int i = 0; i < 16; ++i) Q[i] = Qi[i] * cb;
    for (int i = 0; i < 4; ++i)
      for (int j = i + 1; j < 4; ++j) { const double v = 0.5 * (Q[i * 4 + j] + Q[j * 4 + i]); Q[i * 4 + j] = v; Q[j * 4 + i] = v; }
    sym_eig<4>(Q, w4, V4);
    int np = 0, nn = 0;
    for (int i = 0; i < 4; ++i) { np += w4[i] > 0; nn += w4[i] < 0; }
    if (!((np == 3 && nn == 1) || (np == 1 && nn == 3))) return;
    if (w4[3] > 0) {
      for (int i = 0; i < 16; ++i) Q[i] = -Q[i];
      sym_eig<4>(Q, w4, V4);
    }
    const double Q33[9] = {Q[0], Q[1], Q[2], Q[4], Q[5], Q[6], Q[8], Q[9], Q[10]};
    const double k = inv_det<4>(Q, tmp) / inv_det<3>(Q33, tmp);
    for (int i = 0; i < 3; ++i) s[i] = sqrt(fabs(-k * (1.0 / w4[i])));
    double w3[3];
    sym_eig<3>(Q33, w3, Rm);
  } else {
    const double sc = -1.0 / Qs[15];
    double M[9], w3[3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) M[i * 3 + j] = Qs[i * 4 + j] * sc + t[i] * t[j];
    for (int i = 0; i < 3; ++i)
      for (int j = i + 1; j < 3; ++j) { const double v = 0.5 * (M[i * 3 + j] + M[j * 3 + i]); M[i * 3 + j] = v; M[j * 3 + i] = v; }
    sym_eig<3>(M, w3, Rm);
    if (!(w3[0] > 0)) return;
    for (int i = 0; i < 3; ++i) s[i] = sqrt(w3[i]);
    const double det = Rm[0] * (Rm[4] * Rm[8] - Rm[5] * Rm[7]) - Rm[1] * (Rm[3] * Rm[8] - Rm[5] * Rm[6]) + Rm[2] * (Rm[3] * Rm[7] - Rm[4] * Rm[6]);
    if (det < 0)
      for (int r = 0; r < 3; ++r) Rm[r * 3 + 2] = -Rm[r * 3 + 2];
  }
  // rot_to_euler_zyx (src/utils/matrix_utils.cpp:75-95) then ellipsoid::fromMinimalVector (Ellipsoid.cpp:16-22)
  double roll, pitch, yaw;
  pitch = asin(-Rm[6]);
  if (fabs(pitch - M_PI / 2.0) < 1.0e-3) { roll = 0.0; yaw = atan2(Rm[5] - Rm[1], Rm[2] + Rm[4]) + roll; }
  else if (fabs(pitch + M_PI / 2.0) < 1.0e-3) { roll = 0.0; yaw = atan2(Rm[5] - Rm[1], Rm[2] + Rm[4]) - roll; }
  else { roll = atan2(Rm[7], Rm[8]); yaw = atan2(Rm[3], Rm[0]); }
  const double sy = sin(yaw * 0.5), cy = cos(yaw * 0.5), sp = sin(pitch * 0.5), cp = cos(pitch * 0.5);
  const double sr = sin(roll * 0.5), cr = cos(roll * 0.5);
  Quat qq;
  qq.w = cr * cp * cy + sr * sp * sy;
  qq.x = sr * cp * cy - cr * sp * sy;
  qq.y = cr * sp * cy + sr * cp * sy;
  qq.z = cr * cp * sy - sr * sp * cy;
  qq = q_normalize_pos(qq);
  out[0] = t[0]; out[1] = t[1]; out[2] = t[2];
  out[3] = qq.x; out[4] = qq.y; out[5] = qq.z; out[6] = qq.w;
  out[7] = s[0]; out[8] = s[1]; out[9] = s[2];
  out[26] = 1.0;
}

}  // namespace esl

using namespace esl;

namespace {
struct DevBuf {   // released on every exit path
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t bytes) { ESL_HIP_TRY(hipMalloc(&p, bytes ? bytes : 8)); return ESL_OK; }
};
// the plane matrix (40 doubles per observation) stays in LDS while it fits what the context's device offers per workgroup
// (gfx950: 160 KB => up to 480 observations); beyond that it goes to HBM

int init_launch(esl_ctx* c, const double* poses_Twc, const double* bboxes, int32_t n, const double K[4], int32_t rows, int32_t cols,
                int32_t faithful, int mode, const double* extra, int n_extra, double h[28]) {
  ESL_HIP_TRY(hipSetDevice(c->device));
  DevBuf dp, db, dout, dA;
  int rc;
  if ((rc = dp.alloc((size_t)n * 7 * sizeof(double))) || (rc = db.alloc((size_t)n * 4 * sizeof(double))) || (rc = dout.alloc(28 * sizeof(double))))
    return rc;
  if (n) {
    ESL_HIP_TRY(hipMemcpyAsync(dp.p, poses_Twc, (size_t)n * 7 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    ESL_HIP_TRY(hipMemcpyAsync(db.p, bboxes, (size_t)n * 4 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  InitArgs a{};
  a.K[0] = K[0]; a.K[1] = K[1]; a.K[2] = K[2]; a.K[3] = K[3];
  a.n = n; a.rows = rows; a.cols = cols; a.faithful = faithful; a.mode = mode;
  for (int i = 0; i < n_extra; ++i) a.extra[i] = extra[i];
  size_t lds = 100 * sizeof(double);
  int lds_max = 0;
  ESL_HIP_TRY(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, c->device));
  if (lds + (size_t)n * 40 * sizeof(double) > (size_t)lds_max) {
    if ((rc = dA.alloc((size_t)n * 40 * sizeof(double)))) return rc;
    a.a_glob = (double*)dA.p;
  } else {
    lds += (size_t)n * 40 * sizeof(double);
  }
  ESL_HIP_TRY(hipFuncSetAttribute((const void*)k_init_quadric, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_init_quadric, dim3(1), dim3(64), lds, c->stream, (const double*)dp.p, (const double*)db.p, a, (double*)dout.p);
  ESL_HIP_TRY(hipGetLastError());
  ESL_HIP_TRY(hipMemcpyAsync(h, dout.p, 28 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  ESL_HIP_TRY(hipStreamSynchronize(c->stream));
  return ESL_OK;
}
}  // namespace

extern "C" int esl_init_quadric(esl_ctx* c, const double* poses_Twc, const double* bboxes, int32_t n, const double K[4],
                                int32_t rows, int32_t cols, int32_t faithful, double ellipsoid_out[10], double qstar_out[16],
                                int32_t* ok) {
  if (!c || !poses_Twc || !bboxes || !K || !ellipsoid_out || !qstar_out || !ok || n < 0) return ESL_ERR_INVALID;
  *ok = 0;
  for (int i = 0; i < 10; ++i) ellipsoid_out[i] = 0;
  for (int i = 0; i < 16; ++i) qstar_out[i] = 0;
  if (n < 3) return ESL_OK;  // < 9 planes possible only
  double h[28];
  const int rc = init_launch(c, poses_Twc, bboxes, n, K, rows, cols, faithful, 0, nullptr, 0, h);
  if (rc) return rc;
  for (int i = 0; i < 10; ++i) ellipsoid_out[i] = h[i];
  for (int i = 0; i < 16; ++i) qstar_out[i] = h[10 + i];
  *ok = h[26] > 0.5 ? 1 : 0;
  return ESL_OK;
}

extern "C" int esl_init_from_qstar(esl_ctx* c, const double qstar[16], int32_t faithful, double ellipsoid_out[10], int32_t* ok) {
  if (!c || !qstar || !ellipsoid_out || !ok) return ESL_ERR_INVALID;
  const double K[4] = {1, 1, 0, 0};
  double h[28];
  const int rc = init_launch(c, nullptr, nullptr, 0, K, 0, 0, faithful, 1, qstar, 16, h);
  if (rc) return rc;
  for (int i = 0; i < 10; ++i) ellipsoid_out[i] = h[i];
  *ok = h[26] > 0.5 ? 1 : 0;
  return ESL_OK;
}

extern "C" int esl_init_plane_error(esl_ctx* c, const double* poses_Twc, const double* bboxes, int32_t n, const double K[4],
                                    int32_t rows, int32_t cols, const double ellipsoid[10], double* error_out) {
  if (!c || !K || !ellipsoid || !error_out || n < 0 || (n && (!poses_Twc || !bboxes))) return ESL_ERR_INVALID;
  double h[28];
  const int rc = init_launch(c, poses_Twc, bboxes, n, K, rows, cols, 1, 2, ellipsoid, 10, h);
  if (rc) return rc;
  *error_out = h[27];
  return ESL_OK;
}
