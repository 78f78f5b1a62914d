// PlaneExtractorEsl.cpp — drop-in body for EllipsoidSLAM::PlaneExtractor (reference src/plane/PlaneExtractor.h:41-72,
// src/plane/PlaneExtractor.cpp) on top of esl_extract_planes.  Compile inside the reference tree INSTEAD OF
// src/plane/PlaneExtractor.cpp with -DESL_BUILD_IN_REFERENCE_TREE (needs the reference headers; PCL only for the cloud
// types of the class's public interface, none of its algorithms is called any more).  Every public member is defined.
//
// Behaviour kept: extractPlanes fills the coefficient list (4x1 float, d >= 0), the per-plane clouds and the dense cloud
// (the viewer reads them, Tracking.cpp:735-755); extractGroundPlane applies the reference's own choice to that list (wall
// filter around the camera's y axis, most points, camera centre on the positive side) and returns false when nothing
// qualifies.  Not kept: PCL's boundary refinement of the segments (segmentAndRefine) — see include/esl.h.
#include "esl.h"

#ifdef ESL_BUILD_IN_REFERENCE_TREE
#include <cmath>
#include <vector>

#include "EslAdapterCtx.hpp"
#include "src/config/Config.h"
#include "src/plane/PlaneExtractor.h"

namespace EllipsoidSLAM {

namespace {
const int kMaxPlanes = 256;

esl_plane_params ReadPlaneParams() {
  esl_plane_params p;
  esl_plane_params_default(&p);
  p.min_size = Config::Get<int>("Plane.MinSize");                       // PlaneExtractor.cpp:64-66
  p.angle_threshold_deg = Config::Get<float>("Plane.AngleThreshold");
  p.distance_threshold = Config::Get<float>("Plane.DistanceThreshold");
  return p;
}
}  // namespace

PlaneExtractor::PlaneExtractor(const string& settings) {
  std::cout << "Init plane extractor using : " << settings << std::endl;
  Config::Init();
  Config::SetParameterFile(settings);
  mParam.fx = Config::Get<double>("Camera.fx");
  mParam.fy = Config::Get<double>("Camera.fy");
  mParam.cx = Config::Get<double>("Camera.cx");
  mParam.cy = Config::Get<double>("Camera.cy");
  mParam.scale = Config::Get<double>("DepthMapFactor");
}

void PlaneExtractor::SetParam(PlaneExtractorParam& param) { mParam = param; }
std::vector<PointCloudPCL> PlaneExtractor::GetPoints() { return mvPlanePoints; }
std::vector<PointCloudPCL> PlaneExtractor::GetPotentialGroundPlanePoints() { return mvPotentialGroundPlanePoints; }
std::vector<cv::Mat> PlaneExtractor::GetCoefficients() { return mvPlaneCoefficients; }
PointCloudPCL::Ptr PlaneExtractor::GetCloudDense() { return mpCloudDense; }

void PlaneExtractor::extractPlanes(const cv::Mat& imDepth) {
  mvPlaneCoefficients.clear();
  mvPlanePoints.clear();
  const int row_start = mParam.RangeOpen ? imDepth.rows - mParam.RangeHeight : 0;   // PlaneExtractor.cpp:26-30
  const int rows = imDepth.rows - row_start, cols = imDepth.cols;
  // the dense cloud exactly as the reference builds it (PlaneExtractor.cpp:31-51): the viewer reads it, nothing else does
  PointCloudPCL::Ptr inputCloud(new PointCloudPCL());
  std::vector<uint16_t> depth((size_t)rows * cols);
  for (int m = row_start; m < imDepth.rows; ++m) {
    const ushort* line = imDepth.ptr<ushort>(m);
    for (int n = 0; n < cols; ++n) {
      depth[(size_t)(m - row_start) * cols + n] = line[n];
      pcl::PointXYZRGB p;
      p.z = line[n] / mParam.scale;
      p.x = (n - mParam.cx) * p.z / mParam.fx;
      p.y = (m - mParam.cy) * p.z / mParam.fy;
      p.r = 0; p.g = 0; p.b = 250;
      inputCloud->points.push_back(p);
    }
  }
  inputCloud->height = rows;
  inputCloud->width = cols;
  mpCloudDense = inputCloud;
  if (rows <= 0 || cols <= 0) return;

  const esl_plane_params pp = ReadPlaneParams();
  const double intr[5] = {mParam.fx, mParam.fy, mParam.cx, mParam.cy - row_start, mParam.scale};   // rows counted from row_start
  std::vector<double> planes((size_t)kMaxPlanes * 4);
  std::vector<int32_t> sizes(kMaxPlanes), labels((size_t)rows * cols);
  int32_t n_planes = 0;
  {
    std::lock_guard<std::mutex> lock(esl_adapter::CtxMutex());
    esl_ctx* ctx = esl_adapter::SharedCtx();
    if (!ctx) { std::cerr << "PlaneExtractor: no HIP device (there is no CPU fallback)" << std::endl; return; }
    if (esl_extract_planes(ctx, depth.data(), cols, rows, intr, &pp, kMaxPlanes, planes.data(), sizes.data(), &n_planes, labels.data()) != ESL_OK) {
      std::cerr << "PlaneExtractor: " << esl_last_error() << std::endl;
      return;
    }
  }
  const int n = n_planes < kMaxPlanes ? n_planes : kMaxPlanes;
  std::vector<PointCloudPCL> clouds(n);
  for (size_t i = 0; i < labels.size(); ++i)
    if (labels[i] >= 0 && labels[i] < n) clouds[labels[i]].points.push_back(inputCloud->points[i]);
  for (int i = 0; i < n; ++i) {
    cv::Mat coef(4, 1, CV_32F);                                                        // PlaneExtractor.cpp:90-96
    for (int k = 0; k < 4; ++k) coef.at<float>(k) = (float)planes[(size_t)i * 4 + k];
    mvPlanePoints.push_back(clouds[i]);
    mvPlaneCoefficients.push_back(coef);
  }
}

bool PlaneExtractor::extractGroundPlane(const cv::Mat& depth, g2o::plane& plane) {
  mParam.RangeOpen = false;
  mParam.RangeHeight = depth.rows / 2;
  extractPlanes(depth);
  if (mvPlaneCoefficients.size() < 1) return false;
  // PlaneExtractor.cpp:126-162: wall filter, then the plane with the most points
  int best = -1;
  size_t best_size = 0;
  for (size_t i = 0; i < mvPlaneCoefficients.size(); ++i) {
    cv::Mat& coeff = mvPlaneCoefficients[i];
    const double a = coeff.at<float>(0), b = coeff.at<float>(1), c = coeff.at<float>(2);
    const double theta = std::acos(b / std::sqrt(a * a + b * b + c * c));
    if (theta > M_PI / 4 && theta < 3 * M_PI / 4) continue;
    mvPotentialGroundPlanePoints.push_back(mvPlanePoints[i]);
    if (best < 0 || mvPlanePoints[i].points.size() > best_size) { best = (int)i; best_size = mvPlanePoints[i].points.size(); }
  }
  if (best < 0) {
    std::cout << "Please let the camera be parallel to the ground for initialization." << std::endl;
    return false;
  }
  cv::Mat& g = mvPlaneCoefficients[best];
  Eigen::Vector4d vec(g.at<float>(0), g.at<float>(1), g.at<float>(2), g.at<float>(3));
  if (vec[3] < 0) for (int k = 0; k < 4; ++k) vec[k] = -vec[k];   // camera centre on the positive side (:165-167)
  plane.param = vec;
  return true;
}

}  // namespace EllipsoidSLAM
#endif  // ESL_BUILD_IN_REFERENCE_TREE
