// InitializerEsl.cpp — drop-in body for EllipsoidSLAM::Initializer (reference include/core/Initializer.h:36-79,
// src/core/Initializer.cpp) on top of esl_init_quadric / esl_init_from_qstar / esl_init_plane_error.  Compile inside
// the reference tree INSTEAD OF src/core/Initializer.cpp with -DESL_BUILD_IN_REFERENCE_TREE (needs Eigen + the
// reference headers).  Every public member is defined: both initializeQuadric overloads, quadricErrorWithPlanes,
// getEllipsoidFromQStar, getInitializeResult (the private helpers of the reference body are not needed any more).
//
// Behaviour kept: results by value; success is the sticky member flag read through getInitializeResult()
// (Tracking.cpp:590-593); on failure a default-constructed ellipsoid is returned (Initializer.cpp:38-43).
#include "esl.h"

#ifdef ESL_BUILD_IN_REFERENCE_TREE
#include <vector>

#include "EslAdapterCtx.hpp"
#include "core/Initializer.h"

namespace EllipsoidSLAM {

namespace {
// pose_mat rows: (id) x y z qx qy qz qw -- the last 7 columns (pose.tail(7), Initializer.cpp:74); detection_mat rows: x1 y1 x2 y2 (accuracy)
void FlattenObservations(MatrixXd& pose_mat, MatrixXd& detection_mat, std::vector<double>& poses, std::vector<double>& boxes) {
  const int n = (int)pose_mat.rows(), pc = (int)pose_mat.cols();
  poses.resize((size_t)n * 7); boxes.resize((size_t)n * 4);
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < 7; ++k) poses[(size_t)i * 7 + k] = pose_mat(i, pc - 7 + k);
    for (int k = 0; k < 4; ++k) boxes[(size_t)i * 4 + k] = detection_mat(i, k);
  }
}
}  // namespace

Initializer::Initializer(int rows, int cols) { miImageRows = rows; miImageCols = cols; mbResult = false; }
bool Initializer::getInitializeResult() { return mbResult; }

g2o::ellipsoid Initializer::initializeQuadric(MatrixXd& pose_mat, MatrixXd& detection_mat, Matrix3d& calib) {
  mbResult = false;
  g2o::ellipsoid e;
  std::vector<double> poses, boxes;
  FlattenObservations(pose_mat, detection_mat, poses, boxes);
  const double K[4] = {calib(0, 0), calib(1, 1), calib(0, 2), calib(1, 2)};
  double e10[10], Q[16];
  int32_t ok = 0;
  {
    std::lock_guard<std::mutex> lock(esl_adapter::CtxMutex());
    esl_ctx* ctx = esl_adapter::SharedCtx();
    if (!ctx) return e;
    if (esl_init_quadric(ctx, poses.data(), boxes.data(), (int32_t)pose_mat.rows(), K, miImageRows, miImageCols, /*faithful=*/1, e10, Q,
                         &ok) != ESL_OK) {
      std::cerr << "esl_init_quadric: " << esl_last_error() << std::endl;
      return e;
    }
  }
  mbResult = ok != 0;
  if (!mbResult) return e;
  Vector10d v;
  for (int k = 0; k < 10; ++k) v[k] = e10[k];
  e.fromVector(v);
  e.setColor(Eigen::Vector3d(0, 0, 255));   // Initializer.cpp:52-54
  return e;
}

g2o::ellipsoid Initializer::initializeQuadric(Observations& obs, Matrix3d& calib) {
  const int n = (int)obs.size();
  MatrixXd pose_mat(n, 7), detection_mat(n, 5);
  int id = 0;
  for (auto* ob : obs) {                                            // getDetectionAndPoseMatFromObservations (Initializer.cpp:307-325)
    const Vector7d pos = ob->pFrame->cam_pose_Twc.toVector();
    for (int k = 0; k < 7; ++k) pose_mat(id, k) = pos[k];
    for (int k = 0; k < 4; ++k) detection_mat(id, k) = ob->bbox[k];
    detection_mat(id, 4) = ob->rate;
    ++id;
  }
  g2o::ellipsoid e = initializeQuadric(pose_mat, detection_mat, calib);
  if (getInitializeResult()) e.miLabel = obs[0]->label;
  return e;
}

double Initializer::quadricErrorWithPlanes(MatrixXd& pose_mat, MatrixXd& detection_mat, Matrix3d& calib, g2o::ellipsoid& e) {
  std::vector<double> poses, boxes;
  FlattenObservations(pose_mat, detection_mat, poses, boxes);
  const double K[4] = {calib(0, 0), calib(1, 1), calib(0, 2), calib(1, 2)};
  const Vector10d v = e.toVector();
  double e10[10], err = 0;
  for (int k = 0; k < 10; ++k) e10[k] = v[k];
  std::lock_guard<std::mutex> lock(esl_adapter::CtxMutex());
  esl_ctx* ctx = esl_adapter::SharedCtx();
  if (!ctx) return 0;
  if (esl_init_plane_error(ctx, poses.data(), boxes.data(), (int32_t)pose_mat.rows(), K, miImageRows, miImageCols, e10, &err) != ESL_OK)
    std::cerr << "esl_init_plane_error: " << esl_last_error() << std::endl;
  return err;
}

g2o::ellipsoid Initializer::getEllipsoidFromQStar(Matrix4d& QStar) {
  g2o::ellipsoid e;
  double Q[16], e10[10];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) Q[i * 4 + j] = QStar(i, j);
  int32_t ok = 0;
  {
    std::lock_guard<std::mutex> lock(esl_adapter::CtxMutex());
    esl_ctx* ctx = esl_adapter::SharedCtx();
    if (!ctx) { mbResult = false; return e; }
    if (esl_init_from_qstar(ctx, Q, /*faithful=*/1, e10, &ok) != ESL_OK) {
      std::cerr << "esl_init_from_qstar: " << esl_last_error() << std::endl;
      mbResult = false;
      return e;
    }
  }
  mbResult = ok != 0;      // the reference sets the member flag here too (Initializer.cpp:201-207)
  if (!mbResult) return e;
  Vector10d v;
  for (int k = 0; k < 10; ++k) v[k] = e10[k];
  e.fromVector(v);
  return e;
}

}  // namespace EllipsoidSLAM
#endif
