// EllipsoidExtractorEsl.cpp — drop-in body for EllipsoidSLAM::EllipsoidExtractor (reference
// src/pca/EllipsoidExtractor.h:42-129, .cpp:292-493) on top of esl_fit_frame_ex.  Compile inside the reference tree
// INSTEAD OF src/pca/EllipsoidExtractor.cpp with -DESL_BUILD_IN_REFERENCE_TREE (needs OpenCV for cv::Mat, Eigen, the
// reference headers; this class no longer calls into PCL).
//
// Every public member of the class is defined here (Tracking.cpp:299, 329, 338, 351, 643-650, 779 call them):
//   EstimateLocalEllipsoid, GetResult, GetSymmetryOutputData, SetSupportingPlane, OpenSymmetry, OpenVisualization,
//   ClearPointCloudList, GetPointCloudInProcess, GetPointCloudDebug.
// Behaviour kept: the supporting plane is a BORROWED pointer (.cpp:754-757); the result flag / state code pattern
// (GetResult, miSystemState 0..4, Tracking.cpp:338); prob = symmetry probability or 1; SymmetryOutputData carries the
// world-frame plane(s), prob, type and centre (.cpp:376-393, 415-423) that Tracking.cpp:351-372 consumes.
// Not kept: the intermediate point clouds stay on the device, so the visualisation lists this class used to push into
// Map ("EllipsoidExtractor.*", .cpp:809-834) are not produced; ClearPointCloudList still clears that name space, the two
// cloud getters return empty clouds, SymmetryOutputData::pCloud / pBorders are NULL.
#include "esl.h"

#ifdef ESL_BUILD_IN_REFERENCE_TREE
#include <src/config/Config.h>

#include "EslAdapterCtx.hpp"
#include "src/pca/EllipsoidExtractor.h"

namespace EllipsoidSLAM {

EllipsoidExtractor::EllipsoidExtractor() {
  mResult = false; mbSetPlane = false; mpPlane = NULL; mbOpenVisualization = false; mpMap = NULL; miExtractCount = 0;
  mbOpenSymmetry = false; miSystemState = 0; miEuclideanFilterState = 0;
  mpPoints = new EllipsoidSLAM::PointCloud; mpPointsDebug = new EllipsoidSLAM::PointCloud;   // stay empty (see header note)
  mSymmetryOutputData.result = false; mSymmetryOutputData.pCloud = NULL; mSymmetryOutputData.pBorders = NULL;
}
bool EllipsoidExtractor::GetResult() { return mResult; }
void EllipsoidExtractor::SetSupportingPlane(g2o::plane* pPlane) { mpPlane = pPlane; mbSetPlane = true; }
void EllipsoidExtractor::OpenSymmetry() {   // the label -> symmetry-type table (LoadSymmetryPrior, .cpp:52-79) lives in the kernel
  std::cout << std::endl << " * Open Symmetry Estimation. " << std::endl;
  mbOpenSymmetry = true;
}
SymmetryOutputData EllipsoidExtractor::GetSymmetryOutputData() { return mSymmetryOutputData; }
void EllipsoidExtractor::OpenVisualization(Map* pMap) { mbOpenVisualization = true; mpMap = pMap; }
void EllipsoidExtractor::ClearPointCloudList() {
  if (mbOpenVisualization && mpMap) mpMap->DeletePointCloudList("EllipsoidExtractor", 1);   // partial matching (.cpp:801-807)
}
EllipsoidSLAM::PointCloud* EllipsoidExtractor::GetPointCloudInProcess() { return mpPoints; }
EllipsoidSLAM::PointCloud* EllipsoidExtractor::GetPointCloudDebug() { return mpPointsDebug; }

g2o::ellipsoid EllipsoidExtractor::EstimateLocalEllipsoid(cv::Mat& depth, Eigen::Vector4d& bbox, int label, Eigen::VectorXd& pose,
                                                          camera_intrinsic& camera) {
  miExtractCount++;
  g2o::ellipsoid e;
  miSystemState = 0; mResult = false; mSymmetryOutputData.result = false;
  if (!mbSetPlane || !mpPlane) {   // the reference asserts here (.cpp:92); without the assert it would dereference garbage
    std::cerr << "EllipsoidExtractor: please set the supporting plane first." << std::endl;
    miSystemState = 4;
    return e;
  }
  esl_fit_params p;
  esl_fit_params_default(&p);
  p.depth_max = Config::ReadValue<double>("EllipsoidExtractor_DEPTH_RANGE", 6);
  p.cluster_tolerance = Config::Get<double>("EllipsoidExtraction.Euclidean.ClusterTolerance");
  p.min_cluster_size = Config::Get<int>("EllipsoidExtraction.Euclidean.MinClusterSize");
  p.center_dis = Config::Get<double>("EllipsoidExtraction.Euclidean.CenterDis");
  p.symmetry_open = mbOpenSymmetry ? 1 : 0;
  if (mbOpenSymmetry) {
    p.symmetry_grid = Config::ReadValue<double>("EllipsoidExtraction.Symmetry.GridSize");
    p.symmetry_sigma = Config::ReadValue<double>("SymmetrySolver.Sigma");
  }
  p.depth_scale = camera.scale;
  const double intr[5] = {camera.fx, camera.fy, camera.cx, camera.cy, camera.scale};
  const int np = (int)pose.size();   // pose.tail(7) (.cpp:369)
  const double Twc[7] = {pose[np - 7], pose[np - 6], pose[np - 5], pose[np - 4], pose[np - 3], pose[np - 2], pose[np - 1]};
  const double ground[4] = {mpPlane->param[0], mpPlane->param[1], mpPlane->param[2], mpPlane->param[3]};
  const double box[4] = {bbox[0], bbox[1], bbox[2], bbox[3]};
  double e10[10], prob = 0;
  int32_t state = 0, lab = label;
  esl_fit_symmetry sym;
  {
    std::lock_guard<std::mutex> lock(esl_adapter::CtxMutex());
    esl_ctx* ctx = esl_adapter::SharedCtx();
    if (!ctx) { miSystemState = 4; return e; }
    // depth must be CV_16UC1 and continuous (Frame clones it: Frame.cpp:7-27)
    if (esl_fit_frame_ex(ctx, depth.ptr<uint16_t>(0), depth.cols, depth.rows, box, &lab, 1, Twc, intr, ground, &p, e10, &prob, &state,
                         &sym, NULL) != ESL_OK) {
      std::cerr << "esl_fit_frame: " << esl_last_error() << std::endl;
      miSystemState = 4;
      return e;
    }
  }
  miSystemState = state;
  if (state != 0) return e;
  if (sym.result) {   // what Tracking.cpp:351-372 reads
    mSymmetryOutputData.result = true;
    mSymmetryOutputData.pCloud = NULL;
    mSymmetryOutputData.pBorders = NULL;
    mSymmetryOutputData.prob = sym.prob;
    mSymmetryOutputData.symmetryType = sym.symmetry_type;
    for (int k = 0; k < 4; ++k) { mSymmetryOutputData.planeVec[k] = sym.plane[k]; mSymmetryOutputData.planeVec2[k] = sym.plane2[k]; }
    for (int k = 0; k < 3; ++k) mSymmetryOutputData.center[k] = sym.center[k];
  }
  Vector10d v;
  for (int k = 0; k < 10; ++k) v[k] = e10[k];
  e.fromVector(v);
  e.prob = prob;
  mResult = true;
  return e;
}

}  // namespace EllipsoidSLAM
#endif
