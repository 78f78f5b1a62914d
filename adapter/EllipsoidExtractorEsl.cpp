// EllipsoidExtractorEsl.cpp — drop-in body for EllipsoidSLAM::EllipsoidExtractor::EstimateLocalEllipsoid
// (reference src/pca/EllipsoidExtractor.h:42-129, .cpp:292-493) on top of esl_fit_frame.  Compile inside the
// reference tree INSTEAD OF src/pca/EllipsoidExtractor.cpp with -DESL_BUILD_IN_REFERENCE_TREE (needs OpenCV for
// cv::Mat, Eigen, the reference headers; PCL is no longer needed by this class).
//
// Behaviour kept: the supporting plane is a BORROWED pointer (SetSupportingPlane, .cpp:754-757); the result flag /
// state code pattern (GetResult, miSystemState 0..4, Tracking.cpp:338); prob = symmetry probability or 1.
// Not kept: the debug point clouds pushed into Map when visualisation is on (viewer-only).
#include "esl.h"

#ifdef ESL_BUILD_IN_REFERENCE_TREE
#include <src/config/Config.h>

#include "src/pca/EllipsoidExtractor.h"

namespace EllipsoidSLAM {

static esl_ctx* g_fit_ctx = nullptr;

EllipsoidExtractor::EllipsoidExtractor() { mResult = false; mbSetPlane = false; mbOpenVisualization = false; miExtractCount = 0; mbOpenSymmetry = false; }
bool EllipsoidExtractor::GetResult() { return mResult; }
void EllipsoidExtractor::SetSupportingPlane(g2o::plane* pPlane) { mpPlane = pPlane; mbSetPlane = true; }
void EllipsoidExtractor::OpenSymmetry() { mbOpenSymmetry = true; }   // the label -> symmetry-type table lives in the kernel
SymmetryOutputData EllipsoidExtractor::GetSymmetryOutputData() { return mSymmetryOutputData; }

g2o::ellipsoid EllipsoidExtractor::EstimateLocalEllipsoid(cv::Mat& depth, Eigen::Vector4d& bbox, int label, Eigen::VectorXd& pose,
                                                          camera_intrinsic& camera) {
  miExtractCount++;
  g2o::ellipsoid e;
  miSystemState = 0; mResult = false; mSymmetryOutputData.result = false;
  esl_fit_params p;
  esl_fit_params_default(&p);
  p.depth_max = Config::ReadValue<double>("EllipsoidExtractor_DEPTH_RANGE", 6);
  p.cluster_tolerance = Config::Get<double>("EllipsoidExtraction.Euclidean.ClusterTolerance");
  p.min_cluster_size = Config::Get<int>("EllipsoidExtraction.Euclidean.MinClusterSize");
  p.center_dis = Config::Get<double>("EllipsoidExtraction.Euclidean.CenterDis");
  p.symmetry_open = mbOpenSymmetry ? 1 : 0;
  p.symmetry_grid = Config::ReadValue<double>("EllipsoidExtraction.Symmetry.GridSize");
  p.symmetry_sigma = Config::ReadValue<double>("SymmetrySolver.Sigma");
  p.depth_scale = camera.scale;
  const double intr[5] = {camera.fx, camera.fy, camera.cx, camera.cy, camera.scale};
  const double Twc[7] = {pose[pose.size() - 7], pose[pose.size() - 6], pose[pose.size() - 5], pose[pose.size() - 4],
                         pose[pose.size() - 3], pose[pose.size() - 2], pose[pose.size() - 1]};
  const double ground[4] = {mpPlane->param[0], mpPlane->param[1], mpPlane->param[2], mpPlane->param[3]};
  double e10[10], prob = 0;
  int32_t state = 0, lab = label;
  if (!g_fit_ctx && esl_ctx_create(0, &g_fit_ctx) != ESL_OK) { std::cerr << "esl: " << esl_last_error() << std::endl; miSystemState = 4; return e; }
  // depth must be CV_16UC1 and continuous (Frame clones it: Frame.cpp:7-27)
  if (esl_fit_frame(g_fit_ctx, depth.ptr<uint16_t>(0), depth.cols, depth.rows, bbox.data(), &lab, 1, Twc, intr, ground, &p, e10, &prob,
                    &state) != ESL_OK) {
    std::cerr << "esl_fit_frame: " << esl_last_error() << std::endl;
    miSystemState = 4;
    return e;
  }
  miSystemState = state;
  if (state != 0) return e;
  Vector10d v;
  for (int k = 0; k < 10; ++k) v[k] = e10[k];
  e.fromVector(v);
  e.prob = prob;
  mResult = true;
  return e;
}

}  // namespace EllipsoidSLAM
#endif
