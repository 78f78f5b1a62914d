// esl_ref_surface.hpp — COMPILE-ONLY stand-ins for the headers the adapters are built against in the reference tree
// (Eigen, OpenCV's cv::Mat, the reference's Config / Frame / Map / ellipsoid / plane and the four class declarations
// whose BODIES adapter/*.cpp replace).  Test infrastructure for tests/test_adapter_link.py: it lets this container
// (no Eigen, no OpenCV, no PCL) compile and LINK all three adapters with -DESL_BUILD_IN_REFERENCE_TREE together with
// Tracking's call sites, so that a member Tracking calls but an adapter forgot to define is a link error here and not at
// the maintainer's desk.  Only what the adapters and those call sites touch is declared; the public member signatures of
// Optimizer / Initializer / EllipsoidExtractor / PlaneExtractor are those of reference include/core/Optimizer.h:13-30,
// include/core/Initializer.h:36-79, src/pca/EllipsoidExtractor.h:42-129, src/plane/PlaneExtractor.h:41-72.  Nothing here is used by the product.
#pragma once
#include <cstdint>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW

namespace Eigen {
template <class T, int R, int C>
class Matrix {   // column-major, R / C = -1: run-time size
  std::vector<T> d_;
  int r_, c_;

 public:
  Matrix() : d_((R > 0 ? R : 0) * (C > 0 ? C : 0)), r_(R > 0 ? R : 0), c_(C > 0 ? C : 0) {}
  Matrix(int r, int c) : d_((size_t)r * c), r_(r), c_(c) {}
  Matrix(T a, T b, T c3) : Matrix() { d_[0] = a; d_[1] = b; d_[2] = c3; }
  Matrix(T a, T b, T c3, T d4) : Matrix() { d_[0] = a; d_[1] = b; d_[2] = c3; d_[3] = d4; }
  T& operator()(int i, int j) { return d_[(size_t)j * r_ + i]; }
  const T& operator()(int i, int j) const { return d_[(size_t)j * r_ + i]; }
  T& operator[](int i) { return d_[i]; }
  const T& operator[](int i) const { return d_[i]; }
  T& operator()(int i) { return d_[i]; }
  int rows() const { return r_; }
  int cols() const { return c_; }
  int size() const { return r_ * c_; }
  T* data() { return d_.data(); }
  const Matrix& transpose() const { return *this; }
  void resize(int r, int c) { d_.assign((size_t)r * c, T()); r_ = r; c_ = c; }
};
template <class T, int R, int C>
std::ostream& operator<<(std::ostream& os, const Matrix<T, R, C>& m) {
  for (int i = 0; i < m.size(); ++i) os << (i ? " " : "") << m[i];
  return os;
}
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<double, -1, -1> MatrixXd;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 4, 1> Vector4d;
struct VectorXd : Matrix<double, -1, 1> {
  VectorXd() {}
  explicit VectorXd(int n) : Matrix<double, -1, 1>(n, 1) {}
};
}  // namespace Eigen
using namespace Eigen;
typedef Eigen::Matrix<double, 7, 1> Vector7d;
typedef Eigen::Matrix<double, 9, 1> Vector9d;
typedef Eigen::Matrix<double, 10, 1> Vector10d;

typedef unsigned short ushort;
#define CV_32F 5
namespace cv {
struct Mat {
  int rows = 0, cols = 0;
  std::vector<uint16_t> px;     // a 16-bit depth image ...
  std::vector<float> f32;       // ... or a small float matrix (Mat(rows, cols, CV_32F))
  Mat() {}
  Mat(int r, int c, int) : rows(r), cols(c), f32((size_t)r * c) {}
  template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(px.data()) + (size_t)r * cols; }
  template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(px.data()) + (size_t)r * cols; }
  template <class T> T& at(int i) { return f32[i]; }
};
}  // namespace cv

#include <memory>
namespace pcl {   // the two cloud types of PlaneExtractor's public interface
struct PointXYZRGB { float x = 0, y = 0, z = 0; unsigned char r = 0, g = 0, b = 0; };
template <class P>
struct PointCloud {
  typedef std::shared_ptr<PointCloud<P>> Ptr;
  std::vector<P> points;
  unsigned width = 0, height = 0;
  size_t size() const { return points.size(); }
};
}  // namespace pcl
typedef pcl::PointCloud<pcl::PointXYZRGB> PointCloudPCL;
using std::string;

namespace g2o {
struct SE3Quat {   // types/se3quat.h:110-134, 345-350: the two operations the SLAM branch of the Optimizer adapter needs
  Vector7d v;
  SE3Quat() { v[6] = 1; }
  Vector7d toVector() const { return v; }
  static void rot(const double q[4], const double p[3], double out[3]) {   // Eigen: v + 2 w (u x v) + 2 u x (u x v)
    const double ux = q[1] * p[2] - q[2] * p[1], uy = q[2] * p[0] - q[0] * p[2], uz = q[0] * p[1] - q[1] * p[0];
    out[0] = p[0] + 2 * (q[3] * ux + q[1] * uz - q[2] * uy);
    out[1] = p[1] + 2 * (q[3] * uy + q[2] * ux - q[0] * uz);
    out[2] = p[2] + 2 * (q[3] * uz + q[0] * uy - q[1] * ux);
  }
  SE3Quat inverse() const {   // conjugate, t' = q^-1 (-t); not normalised
    SE3Quat r;
    const double qc[4] = {-v[3], -v[4], -v[5], v[6]}, nt[3] = {-v[0], -v[1], -v[2]};
    double t[3];
    rot(qc, nt, t);
    for (int k = 0; k < 3; ++k) { r.v[k] = t[k]; r.v[3 + k] = qc[k]; }
    r.v[6] = qc[3];
    return r;
  }
  SE3Quat operator*(const SE3Quat& b) const {   // t = ta + Ra tb, q = qa qb, then w >= 0 and unit norm (normalizeRotation)
    SE3Quat r;
    const double qa[4] = {v[3], v[4], v[5], v[6]}, tb[3] = {b.v[0], b.v[1], b.v[2]};
    double t[3];
    rot(qa, tb, t);
    const double ax = v[3], ay = v[4], az = v[5], aw = v[6], bx = b.v[3], by = b.v[4], bz = b.v[5], bw = b.v[6];
    double q[4] = {aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw,
                   aw * bw - ax * bx - ay * by - az * bz};
    if (q[3] < 0) for (double& c : q) c = -c;
    double n = 0;
    for (double c : q) n += c * c;
    n = __builtin_sqrt(n);
    for (int k = 0; k < 3; ++k) r.v[k] = v[k] + t[k];
    for (int k = 0; k < 4; ++k) r.v[3 + k] = q[k] / n;
    return r;
  }
};
class ellipsoid {
 public:
  Vector10d vec;
  Vector9d vec_minimal;
  double prob = 0;
  int miLabel = 0, miInstanceID = 0;
  Vector3d color;
  void fromVector(const Vector10d& x) { vec = x; for (int k = 0; k < 3; ++k) { vec_minimal[k] = x[k]; vec_minimal[6 + k] = x[7 + k]; } }
  Vector10d toVector() const { return vec; }
  Vector9d toMinimalVector() const { return vec_minimal; }
  void setColor(const Vector3d& c, double = 1.0) { color = c; }
};
class plane {
 public:
  Vector4d param;
};
}  // namespace g2o
using namespace g2o;

namespace EllipsoidSLAM {
struct PointXYZRGB { double x, y, z; unsigned char r, g, b; int size = 1; };
typedef std::vector<PointXYZRGB> PointCloud;
struct camera_intrinsic { double fx, fy, cx, cy, scale; };

class Config {   // src/config/Config.h: Get<T>(key) reads the yaml, ReadValue<T>(key, default) a run-time override first
 public:
  static std::map<std::string, double>& values() { static std::map<std::string, double> m; return m; }
  template <class T> static T Get(const std::string& key) { return T(values()[key]); }
  template <class T> static T ReadValue(const std::string& key, T = 0) { return T(values()[key]); }
  static void Init() {}
  static void SetParameterFile(const std::string&) {}
};

class Frame {
 public:
  int frame_seq_id = 0;
  cv::Mat frame_img;
  g2o::SE3Quat cam_pose_Tcw, cam_pose_Twc;
  std::vector<g2o::ellipsoid*> mpLocalObjects;
};

class Map {
 public:
  std::map<int, g2o::ellipsoid*> ells;
  int deleted_lists = 0;
  void addEllipsoid(g2o::ellipsoid* e) { ells[e->miInstanceID] = e; }
  std::map<int, g2o::ellipsoid*> GetAllEllipsoidsMap() { return ells; }
  bool DeletePointCloudList(const std::string&, int = 0) { ++deleted_lists; return true; }
};

class Observation {
 public:
  int label;
  Vector4d bbox;
  double rate;
  Frame* pFrame;
  int instance;
};
typedef std::vector<Observation*> Observations;

class SymmetryOutputData {   // src/symmetry/Symmetry.h:16-32
 public:
  bool result;
  PointCloud* pCloud;
  Vector4d planeVec, planeVec2;
  double prob;
  PointCloud* pBorders;
  Vector3d center;
  int symmetryType;
};

class Optimizer {   // include/core/Optimizer.h:13-30
 public:
  Optimizer();
  void GlobalObjectGraphOptimization(std::vector<Frame*>& pFrames, Map* pMap, int rows, int cols, Matrix3d& mCalib,
                                     std::map<int, Observations>& objectObservations, bool save_graph = false,
                                     bool withAssociation = false, bool check_visibility = false);
  void SetGroundPlane(Vector4d& normal);

 private:
  bool mbGroundPlaneSet;
  Vector4d mGroundPlaneNormal;
};

class Initializer {   // include/core/Initializer.h:36-79 (public part + the members the body keeps)
 public:
  Initializer(int rows, int cols);
  g2o::ellipsoid initializeQuadric(MatrixXd& pose_mat, MatrixXd& detection_mat, Matrix3d& calib);
  g2o::ellipsoid initializeQuadric(Observations& obs, Matrix3d& calib);
  double quadricErrorWithPlanes(MatrixXd& pose_mat, MatrixXd& detection_mat, Matrix3d& calib, g2o::ellipsoid& e);
  g2o::ellipsoid getEllipsoidFromQStar(Matrix4d& QStar);
  bool getInitializeResult();

 private:
  bool mbResult;
  int miImageRows, miImageCols;
};

class EllipsoidExtractor {   // src/pca/EllipsoidExtractor.h:42-129 (public part + the members the body keeps)
 public:
  EllipsoidExtractor();
  void OpenSymmetry();
  void SetSupportingPlane(g2o::plane* pPlane);
  g2o::ellipsoid EstimateLocalEllipsoid(cv::Mat& depth, Eigen::Vector4d& bbox, int label, Eigen::VectorXd& pose, camera_intrinsic& camera);
  void OpenVisualization(Map* pMap);
  void ClearPointCloudList();
  bool GetResult();
  SymmetryOutputData GetSymmetryOutputData();
  EllipsoidSLAM::PointCloud* GetPointCloudInProcess();
  EllipsoidSLAM::PointCloud* GetPointCloudDebug();

 private:
  bool mResult;
  int miEuclideanFilterState, miSystemState;
  EllipsoidSLAM::PointCloud *mpPoints, *mpPointsDebug;
  bool mbSetPlane;
  g2o::plane* mpPlane;
  SymmetryOutputData mSymmetryOutputData;
  bool mbOpenVisualization;
  Map* mpMap;
  int miExtractCount;
  bool mbOpenSymmetry;
};

struct PlaneExtractorParam {   // src/plane/PlaneExtractor.h:32-39
  double fx, fy, cx, cy;
  double scale;
  bool RangeOpen = false;
  int RangeHeight;
};

class PlaneExtractor {   // src/plane/PlaneExtractor.h:41-72
 public:
  PlaneExtractor() {}
  PlaneExtractor(const string& settings);
  bool extractGroundPlane(const cv::Mat& depth, g2o::plane& plane);
  void extractPlanes(const cv::Mat& depth);
  void SetParam(PlaneExtractorParam& param);
  std::vector<PointCloudPCL> GetPoints();
  std::vector<PointCloudPCL> GetPotentialGroundPlanePoints();
  std::vector<cv::Mat> GetCoefficients();
  PointCloudPCL::Ptr GetCloudDense();

 private:
  int mParamRangeHeight;
  PlaneExtractorParam mParam;
  std::vector<PointCloudPCL> mvPlanePoints;
  std::vector<PointCloudPCL> mvPotentialGroundPlanePoints;
  std::vector<cv::Mat> mvPlaneCoefficients;
  PointCloudPCL::Ptr mpCloudDense;
};
}  // namespace EllipsoidSLAM
