// InitializerEsl.cpp — drop-in body for EllipsoidSLAM::Initializer (reference include/core/Initializer.h:36-79,
// src/core/Initializer.cpp) on top of esl_init_quadric.  Compile inside the reference tree INSTEAD OF
// src/core/Initializer.cpp with -DESL_BUILD_IN_REFERENCE_TREE (needs Eigen + the reference headers).
//
// Behaviour kept: results by value; success is the sticky member flag read through getInitializeResult()
// (Tracking.cpp:590-593); on failure a default-constructed ellipsoid is returned (Initializer.cpp:38-43).
#include "esl.h"

#ifdef ESL_BUILD_IN_REFERENCE_TREE
#include <vector>

#include "core/Initializer.h"

namespace EllipsoidSLAM {

static esl_ctx* g_init_ctx = nullptr;

Initializer::Initializer(int rows, int cols) { miImageRows = rows; miImageCols = cols; mbResult = false; }
bool Initializer::getInitializeResult() { return mbResult; }

// pose_mat rows: x y z qx qy qz qw (Twc) ; detection_mat rows: x1 y1 x2 y2 (accuracy)
g2o::ellipsoid Initializer::initializeQuadric(MatrixXd& pose_mat, MatrixXd& detection_mat, Matrix3d& calib) {
  mbResult = false;
  g2o::ellipsoid e;
  const int n = (int)pose_mat.rows();
  std::vector<double> poses((size_t)n * 7), boxes((size_t)n * 4);
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < 7; ++k) poses[(size_t)i * 7 + k] = pose_mat(i, pose_mat.cols() - 7 + k);   // pose.tail(7), Initializer.cpp:74
    for (int k = 0; k < 4; ++k) boxes[(size_t)i * 4 + k] = detection_mat(i, k);
  }
  const double K[4] = {calib(0, 0), calib(1, 1), calib(0, 2), calib(1, 2)};
  double e10[10], Q[16];
  int ok = 0;
  if (!g_init_ctx && esl_ctx_create(0, &g_init_ctx) != ESL_OK) { std::cerr << "esl: " << esl_last_error() << std::endl; return e; }
  if (esl_init_quadric(g_init_ctx, poses.data(), boxes.data(), n, K, miImageRows, miImageCols, /*faithful=*/1, e10, Q, &ok) != ESL_OK) {
    std::cerr << "esl_init_quadric: " << esl_last_error() << std::endl;
    return e;
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
  MatrixXd pose_mat((int)obs.size(), 7), detection_mat((int)obs.size(), 5);
  int id = 0;
  for (auto* ob : obs) {                                            // Initializer.cpp:307-325
    pose_mat.row(id) = ob->pFrame->cam_pose_Twc.toVector();
    detection_mat.row(id) << ob->bbox[0], ob->bbox[1], ob->bbox[2], ob->bbox[3], ob->rate;
    ++id;
  }
  g2o::ellipsoid e = initializeQuadric(pose_mat, detection_mat, calib);
  if (getInitializeResult()) e.miLabel = obs[0]->label;
  return e;
}

}  // namespace EllipsoidSLAM
#endif
