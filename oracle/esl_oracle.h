/*
 * esl_oracle.h — CPU restatement (plain C, fp64, single thread) of the reference hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (object-oriented-slam_amd/, the C-ABI
 * library) may include, link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference ships no tests, golden vectors or fixtures for
 * this path and cannot be compiled here (every translation unit needs Eigen, which is absent, plus
 * OpenCV/PCL for the fit path).  The restatement is pinned instead by (i) analytic known-answer
 * tests (tests/test_oracle_kat.py), (ii) an independent numpy restatement
 * (oracle/np_oracle.py) that must agree with it, and (iii) golden vectors generated from both
 * (tests/golden/).  Each function cites the reference file:line it follows.
 */
#ifndef ESL_ORACLE_H_
#define ESL_ORACLE_H_

#include "../include/esl.h"

#ifdef __cplusplus
extern "C" {
#endif

/* linear-solver choice of the oracle LM */
#define ESL_ORACLE_DENSE 0  /* faithful: dense pivoted LDLT of the whole free system (linear_solver_dense.h:65-113) */
#define ESL_ORACLE_BLOCK 1  /* "CPU, improved over reference": per-ellipsoid 9x9 (mapping) / Schur + dense (SLAM) */
#define ESL_ORACLE_CAMFIRST 2 /* SLAM mode, cameras eliminated first along the odometry chain (plain block-bidiagonal factor, dense Y,
                               * pivoted LDLT of the reduced ELLIPSOID system): the same x as the other two; falls back to BLOCK when
                               * the camera block is not block tridiagonal.  The CPU counterpart of csrc/esl_cf.hpp. */

/* --- SE3 / ellipsoid primitives (g2o types/se3quat.h, src/core/Ellipsoid.cpp) ------------------*/
void esl_oracle_se3_exp(const double upd[6], double out[7]);
void esl_oracle_se3_log(const double T[7], double out[6]);
void esl_oracle_se3_mul(const double A[7], const double B[7], double out[7]);
void esl_oracle_se3_inv(const double A[7], double out[7]);
void esl_oracle_cam_oplus(const double cam[7], const double upd[6], double out[7]);
void esl_oracle_obj_oplus(const double obj[10], const double upd[9], double out[10]);
void esl_oracle_obj_to_minimal(const double obj[10], double out[9]);
void esl_oracle_obj_from_minimal(const double v[9], double out[10]);
void esl_oracle_quadric(const double obj[10], double Q[16]);
void esl_oracle_project_bbox(const double cam[7], const double obj[10], const double K[4], double out[4]);

/* --- residuals (src/core/BasicEllipsoidEdges.cpp, types_six_dof_expmap.h) ---------------------*/
void esl_oracle_res_bbox(const double cam[7], const double obj[10], const double K[4],
                         const double meas[4], double r[4]);
void esl_oracle_res_e3d(const double cam[7], const double obj[10], const double meas[10], double r[9]);
void esl_oracle_res_grav(const double obj[10], const double normal[4], double r[1]);
void esl_oracle_res_odom(const double cam_i[7], const double cam_j[7], const double meas[7], double r[6]);

/* --- numeric Jacobians, row-major D x dim (g2o core/base_binary_edge.hpp:131-205) ---------------*/
void esl_oracle_jac_bbox(const double cam[7], const double obj[10], const double K[4], const double meas[4],
                         double delta, double Jc[4 * 6], double Jo[4 * 9]);
void esl_oracle_jac_e3d(const double cam[7], const double obj[10], const double meas[10], double delta,
                        double Jc[9 * 6], double Jo[9 * 9]);
void esl_oracle_jac_grav(const double obj[10], const double normal[4], double delta, double Jo[9]);
void esl_oracle_jac_odom(const double cam_i[7], const double cam_j[7], const double meas[7], double delta,
                         double Ji[6 * 6], double Jj[6 * 6]);

/* --- dense pivoted LDLT (Eigen::LDLT as used by linear_solver_dense.h:107-111) -----------------*/
/* A is n x n row-major symmetric (overwritten), b -> x; returns 1 iff isPositive(). */
int esl_oracle_ldlt_solve(double* A, int n, const double* b, double* x);

/* --- the optimiser (src/core/Optimizer.cpp:83-317 + g2o LM) --------------------------------------*/
int esl_oracle_optimize(const esl_graph* g, double* cams_io, double* objs_io, const esl_lm_params* p,
                        int solver, esl_lm_report* out);

/* one linearisation: dense H (n x n row-major) and b of the free system in g2o's variable order.
 * free_index receives, for every vertex (cams then objs), the first row in H or -1.
 * Returns n (call with H == NULL to query n). */
int esl_oracle_build_system(const esl_graph* g, const double* cams, const double* objs, double delta,
                            int drop_nan_bbox, double* H, double* b, int32_t* free_index, double* chi2);

/* --- Initializer::initializeQuadric (src/core/Initializer.cpp:24-248) -------------------------------*/
int esl_oracle_init_quadric(const double* poses_Twc, const double* bboxes, int n, const double K[4], int rows, int cols,
                            int faithful, double ell_out[10], double qstar_out[16], int* ok);

/* --- EllipsoidExtractor::EstimateLocalEllipsoid (src/pca/EllipsoidExtractor.cpp:292-493) --------------------
 * debug_out (optional, 16 doubles per box): [0] in-range samples, [1] 1 cm voxels, [2] after plane filter,
 * [3] clusters >= MinClusterSize, [4] chosen cluster size, [5] GridSize voxels, [6..14] prob of the 9 hypotheses */
int esl_oracle_fit_frame(const uint16_t* depth, int32_t width, int32_t height, const double* bboxes, const int32_t* labels,
                         int32_t n_boxes, const double Twc[7], const double intr[5], const double ground[4],
                         const esl_fit_params* p, double* ellipsoids_out, double* prob_out, int32_t* status_out,
                         double* debug_out);
/* same + SymmetryOutputData (src/symmetry/Symmetry.h:16-32) per box, 16 doubles: plane (world) [0..3], plane2 [4..7],
 * prob [8], symmetry type [9], result flag [10], centre of the completed cloud (world) [11..13] */
int esl_oracle_fit_frame_ex(const uint16_t* depth, int32_t width, int32_t height, const double* bboxes, const int32_t* labels,
                            int32_t n_boxes, const double Twc[7], const double intr[5], const double ground[4],
                            const esl_fit_params* p, double* ellipsoids_out, double* prob_out, int32_t* status_out,
                            double* debug_out, double* sym_out);

/* Ground plane of one depth image (esl_oracle_plane.c; esl_extract_ground_plane of include/esl.h).  normals_out (optional):
 * 4 floats per pixel, nx ny nz d, nx = NaN where no normal exists. */
int esl_oracle_extract_ground_plane(const uint16_t* depth, int32_t width, int32_t height, const double intr[5],
                                    const esl_plane_params* p, double plane_out[4], int32_t* ok, int32_t* n_planes,
                                    int32_t* n_pixels, float* normals_out);
/* every plane segment (esl_extract_planes of include/esl.h) */
int esl_oracle_extract_planes(const uint16_t* depth, int32_t width, int32_t height, const double intr[5], const esl_plane_params* p,
                              int32_t max_planes, double* planes_out, int32_t* sizes_out, int32_t* n_planes, int32_t* labels_out);

/* 0: the reference's reprojection residual (default); 1: plane-tangency rows (esl_lm_params::bbox_residual).  Applies to
 * esl_oracle_res_bbox / jac_bbox / build_system; esl_oracle_optimize sets it from its params. */
void esl_oracle_set_bbox_residual(int mode);

/* timing helper for bench.py's cpu_baseline: seconds spent in linearise / solve / error evaluation
 * of the last esl_oracle_optimize call */
void esl_oracle_last_timing(double t[3]);
/* solver ESL_ORACLE_CAMFIRST, summed over the trials of the last run: chain factor + forward substitution | T = D - Y^T Y | pivoted
 * LDLT of T | back-substitution */
void esl_oracle_last_camfirst_timing(double t[4]);

#ifdef __cplusplus
}
#endif
#endif
