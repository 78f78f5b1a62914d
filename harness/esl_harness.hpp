// esl_harness.hpp — build-owned counterpart of the reference's Tracking-side harness and dataset I/O,
// so that the TUM fr3_cabinet clip (BASELINE.json configs[0]) runs end to end through the C-ABI without OpenCV / PCL /
// Pangolin / Eigen.  Host code only: every number on the hot path still comes from libesl_hip.so (or, in the tests, from
// the CPU checker) through the Backend the Tracker is instantiated with.
//
// What is restated (reference file:line):
//   dataset layout + association         src/tum_rgbd/io.cpp:14-48 (loadDataset), 156-244 (associate.txt /
//                                        associateGroundtruth.txt, |dt| < 1 ms matching, gt stamp minus 2 digits),
//                                        277-290 (bbox/<rgb stamp>.txt: id x1 y1 x2 y2 label rate instance; README.md:69)
//   text readers / writers               src/utils/dataprocess_utils.cpp:72-127 (split on " \t,", setprecision(12))
//   frame construction                   src/core/Frame.cpp:7-27 (Twc = pose.tail(7) as is, Tcw = inverse, sequence id)
//   per-frame flow                       src/core/Tracking.cpp:172-186 (GrabPoseAndObjects), 493-564 (UpdateObjectObservation),
//                                        286-375 (UpdateDepthEllipsoidEstimation), 377-413 (3-D associations),
//                                        421-475 (key-observation check), 566-638 (JudgeInitialization: SVD from >= 15
//                                        observations, else the latest single-frame ellipsoid), 219-231 (optimise when the map
//                                        holds an ellipsoid), 810-852 (RefreshObjectHistory), 855-896 (SaveObjectHistory)
//   border filter                        src/utils/dataprocess_utils.cpp:150-190 (calibrateMeasurement)
//   graph assembly                       adapter/OptimizerEsl.cpp (esl_adapter::Flatten = src/core/Optimizer.cpp:127-279)
//   outputs                              src/core/System.cpp:75-91 (objects.txt), src/core/Optimizer.cpp:281-288 (graph summary)
//   data association                   src/core/DataAssociation.cpp:16-135 (used when the clip carries no instance ids)
//   ground plane                       src/core/Tracking.cpp:690-799, src/core/System.cpp:46 (backend: esl_extract_ground_plane)
// Not restated: the viewer and the dense builder.
#pragma once
#include <dirent.h>
#include <zlib.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "../adapter/OptimizerEsl.cpp"   // esl_adapter::Flatten / MakeGraph (+ esl.h)

namespace esl_harness {

// ---- f-2: on-disk formats ---------------------------------------------------------------------------------------------
// split on " \t," with token compression (boost::split(..., token_compress_on), dataprocess_utils.cpp:84,136)
inline std::vector<std::string> split_tokens(const std::string& line) {
  std::vector<std::string> out;
  std::string cur;
  bool in_sep = false;
  for (char ch : line) {
    if (ch == ' ' || ch == '\t' || ch == ',' || ch == '\r') {
      if (!in_sep) { out.push_back(cur); cur.clear(); in_sep = true; }
    } else { cur.push_back(ch); in_sep = false; }
  }
  out.push_back(cur);
  return out;
}
inline std::vector<std::vector<std::string>> read_string_table(const std::string& path, int drop_lines = 0) {
  std::vector<std::vector<std::string>> rows;
  std::ifstream fin(path);
  std::string line;
  for (int i = 0; i < drop_lines; ++i) std::getline(fin, line);
  while (std::getline(fin, line)) rows.push_back(split_tokens(line));
  return rows;
}
// readDataFromFile: a numeric matrix, one row per line (an empty file = no rows)
inline std::vector<std::vector<double>> read_number_table(const std::string& path) {
  std::vector<std::vector<double>> rows;
  for (const auto& toks : read_string_table(path)) {
    std::vector<double> r;
    for (const auto& t : toks) if (!t.empty()) r.push_back(std::stod(t));
    if (!r.empty()) rows.push_back(r);
  }
  return rows;
}
// saveMatToFile: setprecision(12), single blanks
inline bool save_number_table(const std::vector<std::vector<double>>& rows, const std::string& path) {
  std::ofstream fout(path);
  if (!fout) return false;
  for (const auto& r : rows) {
    for (size_t m = 0; m < r.size(); ++m) {
      fout << std::setprecision(12) << r[m];
      if (m + 1 < r.size()) fout << " ";
    }
    fout << std::endl;
  }
  return true;
}

// 16-bit grey-scale PNG (the TUM depth images; cv::imread(..., IMREAD_UNCHANGED) in io.cpp:86): non-interlaced, filters 0-4
inline bool read_png16(const std::string& path, std::vector<uint16_t>& img, int& w, int& h) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::vector<unsigned char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (buf.size() < 8 || std::memcmp(buf.data(), sig, 8) != 0) return false;
  auto be32 = [&](size_t p) { return ((uint32_t)buf[p] << 24) | ((uint32_t)buf[p + 1] << 16) | ((uint32_t)buf[p + 2] << 8) | buf[p + 3]; };
  std::vector<unsigned char> idat;
  int depth = 0, ctype = -1, interlace = 0;
  w = h = 0;
  for (size_t p = 8; p + 12 <= buf.size();) {
    const uint32_t len = be32(p);
    const std::string type((const char*)&buf[p + 4], 4);
    if (p + 12 + len > buf.size()) return false;
    if (type == "IHDR") { w = (int)be32(p + 8); h = (int)be32(p + 12); depth = buf[p + 16]; ctype = buf[p + 17]; interlace = buf[p + 20]; }
    else if (type == "IDAT") idat.insert(idat.end(), buf.begin() + p + 8, buf.begin() + p + 8 + len);
    else if (type == "IEND") break;
    p += 12 + len;
  }
  if (w <= 0 || h <= 0 || ctype != 0 || interlace != 0 || (depth != 16 && depth != 8)) return false;
  const int bpp = depth / 8;
  const size_t stride = (size_t)w * bpp;
  std::vector<unsigned char> raw((stride + 1) * (size_t)h);
  uLongf out_len = (uLongf)raw.size();
  if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) return false;
  std::vector<unsigned char> prev(stride, 0), cur(stride);
  img.assign((size_t)w * h, 0);
  for (int y = 0; y < h; ++y) {
    const unsigned char* row = raw.data() + (stride + 1) * (size_t)y;
    const int ft = row[0];
    for (size_t x = 0; x < stride; ++x) {
      const int a = x >= (size_t)bpp ? cur[x - bpp] : 0, b = prev[x], c = x >= (size_t)bpp ? prev[x - bpp] : 0;
      int v = row[1 + x];
      if (ft == 1) v += a;
      else if (ft == 2) v += b;
      else if (ft == 3) v += (a + b) / 2;
      else if (ft == 4) { const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
      else if (ft != 0) return false;
      cur[x] = (unsigned char)(v & 0xff);
    }
    for (int x = 0; x < w; ++x) img[(size_t)y * w + x] = bpp == 2 ? (uint16_t)((cur[2 * x] << 8) | cur[2 * x + 1]) : cur[x];
    prev.swap(cur);
  }
  return true;
}

struct Detection { double id, x1, y1, x2, y2, label, rate, instance; };   // one row of bbox/<stamp>.txt

// TUMRGBD::Dataset (io.cpp): frames = the files of rgb/ sorted by their stamp; depth and pose by stamp association
class Dataset {
 public:
  bool load(std::string dir) {
    if (!dir.empty() && dir.back() != '/') dir.push_back('/');
    dir_ = dir;
    stamps_.clear();
    if (DIR* d = opendir((dir + "rgb/").c_str())) {
      while (dirent* e = readdir(d)) {
        std::string n = e->d_name;
        if (n == "." || n == "..") continue;
        const size_t dot = n.rfind('.');
        stamps_.push_back(dot == std::string::npos ? n : n.substr(0, dot));   // bare name = rgb time stamp (io.cpp:236-244)
      }
      closedir(d);
    } else return false;
    std::sort(stamps_.begin(), stamps_.end(), [](const std::string& a, const std::string& b) { return std::atof(a.c_str()) < std::atof(b.c_str()); });
    for (const auto& r : read_string_table(dir + "groundtruth.txt")) {   // stamp x y z qx qy qz qw
      if (r.size() < 8 || r[0].empty() || r[0][0] == '#') continue;
      std::array<double, 7> p;
      for (int k = 0; k < 7; ++k) p[k] = std::stod(r[k + 1]);
      gt_[r[0]] = p;
    }
    std::map<std::string, std::string> rgb2depth_path, rgb2depth, rgb2gt;
    for (const auto& r : read_string_table(dir + "associate.txt")) if (r.size() >= 4) { rgb2depth[r[0]] = r[2]; rgb2depth_path[r[0]] = r[3]; }
    for (const auto& r : read_string_table(dir + "associateGroundtruth.txt"))
      if (r.size() >= 3) rgb2gt[r[0]] = r[2].size() > 2 ? r[2].substr(0, r[2].size() - 2) : r[2];   // drop two trailing digits (io.cpp:205-208)
    depth_path_.assign(stamps_.size(), ""); gt_stamp_.assign(stamps_.size(), "");
    for (size_t i = 0; i < stamps_.size(); ++i) {
      const auto it = by_number(rgb2depth, stamps_[i]);
      if (it != rgb2depth.end()) depth_path_[i] = rgb2depth_path[it->first];
      const auto ig = by_number(rgb2gt, stamps_[i]);
      if (ig != rgb2gt.end()) gt_stamp_[i] = ig->second;
    }
    return true;
  }
  int size() const { return (int)stamps_.size(); }
  const std::string& stamp(int i) const { return stamps_[i]; }
  // findFrameUsingID: false when the depth image or the pose is missing for this frame
  bool frame(int i, std::vector<uint16_t>& depth, int& w, int& h, std::array<double, 7>& pose) const {
    if (i < 0 || i >= size() || depth_path_[i].empty() || gt_stamp_[i].empty()) return false;
    bool found = false;
    for (const auto& kv : gt_)
      if (std::fabs(std::atof(kv.first.c_str()) - std::atof(gt_stamp_[i].c_str())) < 0.001) { pose = kv.second; found = true; break; }
    if (!found) return false;
    return read_png16(dir_ + depth_path_[i], depth, w, h);
  }
  std::vector<Detection> detections(int i) const {   // getDetectionMat: bbox/<rgb stamp>.txt
    std::vector<Detection> out;
    for (const auto& r : read_number_table(dir_ + "bbox/" + stamps_[i] + ".txt"))
      if (r.size() >= 8) out.push_back(Detection{r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]});
    return out;
  }

 private:
  template <class M>
  static typename M::const_iterator by_number(const M& m, const std::string& stamp) {   // AssociateWithNumber (io.cpp:96-109)
    for (auto it = m.begin(); it != m.end(); ++it) {
      if (it->first.empty() || stamp.empty()) continue;
      if (std::fabs(std::atof(it->first.c_str()) - std::atof(stamp.c_str())) < 0.001) return it;
    }
    return m.end();
  }
  std::string dir_;
  std::vector<std::string> stamps_, depth_path_, gt_stamp_;
  std::map<std::string, std::array<double, 7>> gt_;
};

// ---- small SE3 helpers on 7-vectors x y z qx qy qz qw (g2o::SE3Quat semantics, types/se3quat.h:110-134) ------------------
using Vec7 = std::array<double, 7>;
using Vec10 = std::array<double, 10>;
inline void q_rot(const double* q, const double v[3], double out[3]) {   // Eigen quaternion * vector (no normalisation)
  const double ux = q[1] * v[2] - q[2] * v[1], uy = q[2] * v[0] - q[0] * v[2], uz = q[0] * v[1] - q[1] * v[0];
  const double tx = 2 * ux, ty = 2 * uy, tz = 2 * uz;
  out[0] = v[0] + q[3] * tx + (q[1] * tz - q[2] * ty);
  out[1] = v[1] + q[3] * ty + (q[2] * tx - q[0] * tz);
  out[2] = v[2] + q[3] * tz + (q[0] * ty - q[1] * tx);
}
inline Vec7 se3_inverse(const Vec7& T) {   // conjugate, t' = -(q^-1 t); not normalised
  Vec7 r;
  const double qc[4] = {-T[3], -T[4], -T[5], T[6]};
  const double nt[3] = {-T[0], -T[1], -T[2]};
  q_rot(qc, nt, r.data());
  r[3] = qc[0]; r[4] = qc[1]; r[5] = qc[2]; r[6] = qc[3];
  return r;
}
inline Vec7 se3_mul(const Vec7& A, const Vec7& B) {   // operator*: normalises, w >= 0
  Vec7 r;
  double rt[3];
  q_rot(&A[3], &B[0], rt);
  for (int k = 0; k < 3; ++k) r[k] = A[k] + rt[k];
  const double ax = A[3], ay = A[4], az = A[5], aw = A[6], bx = B[3], by = B[4], bz = B[5], bw = B[6];
  double q[4] = {aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw,
                 aw * bw - ax * bx - ay * by - az * bz};
  if (q[3] < 0) for (double& v : q) v = -v;
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int k = 0; k < 4; ++k) r[3 + k] = q[k] / n;
  return r;
}
// ellipsoid::toMinimalVector: x y z roll pitch yaw a b c (se3quat.h:184-207)
inline std::array<double, 9> to_minimal(const Vec10& e) {
  const double qx = e[3], qy = e[4], qz = e[5], qw = e[6];
  return {e[0], e[1], e[2], std::atan2(2 * (qw * qx + qy * qz), 1 - 2 * (qx * qx + qy * qy)), std::asin(2 * (qw * qy - qz * qx)),
          std::atan2(2 * (qw * qz + qx * qy), 1 - 2 * (qy * qy + qz * qz)), e[7], e[8], e[9]};
}

// ---- f-1: Tracking-side logic --------------------------------------------------------------------------------------------
struct Settings {   // Example/param/TUM3.yaml + src/config/Config.cpp:29-30
  double fx = 535.4, fy = 539.2, cx = 320.1, cy = 247.6, scale = 5000.0;
  int rows = 480, cols = 640;
  int border_pixels = 10, length_limit = 0;          // Measurement.Border.Pixels / LengthLimit.Pixels
  int min_init_frames = 15;                          // Tracking_MINIMUM_INITIALIZATION_FRAME
  bool keyframe_check_close = true;                  // Tracking.KeyFrameCheck.Close: 1
  double scale_3d = 10000, gravity_scale = 100;      // Optimizer.Edges.*
  bool gravity_open = true, depth_ellipsoid = true, optimization = true, symmetry = true;
  esl_fit_params fit;                                // filled by the backend's defaults, then overridden from here
  esl_plane_params plane = {200, 5.0, 0.1, 10, 0.05, 100, 1, 0.02, 0.001};   // Plane.MinSize / AngleThreshold / DistanceThreshold + PlaneExtractor.cpp:57-58, 74 + segmentAndRefine's pass (:82)
  bool with_association = true;                      // rgbd.cpp:73
  bool slam_mode = false;                            // Optimizer.cpp:126 bSLAM_mode (a constant false in the reference)
  bool check_visibility = false;                     // GlobalObjectGraphOptimization's last argument (false at Tracking.cpp:226)
};

// calibrateMeasurement: true = the box touches the border band (or is too small); the box is overwritten as the reference does
inline bool border_filter(double m[4], int rows, int cols, int border, int min_size) {
  const int xl = (int)(m[2] - m[0]), yl = (int)(m[3] - m[1]);
  if (xl < min_size || yl < min_size) return true;
  double cal[4] = {-1, -1, -1, -1};
  int ok = 0;
  if (m[0] > border && m[0] < cols - 1 - border) { cal[0] = m[0]; ++ok; }
  if (m[2] > border && m[2] < cols - 1 - border) { cal[2] = m[2]; ++ok; }
  if (m[1] > border && m[1] < rows - 1 - border) { cal[1] = m[1]; ++ok; }
  if (m[3] > border && m[3] < rows - 1 - border) { cal[3] = m[3]; ++ok; }
  for (int k = 0; k < 4; ++k) m[k] = cal[k];
  return ok != 4;
}

// the types esl_adapter::Flatten walks (member names as in the reference's Frame / Observation / g2o::ellipsoid)
struct Pose7 {
  Vec7 v;
  Vec7 toVector() const { return v; }
  Pose7 inverse() const { return Pose7{se3_inverse(v)}; }                        // g2o::SE3Quat::inverse
  Pose7 operator*(const Pose7& b) const { return Pose7{se3_mul(v, b.v)}; }       // g2o::SE3Quat::operator*
};
struct Ell {
  Vec10 v{};
  double prob = 0;
  int miLabel = 0, miInstanceID = -1;
  Vec10 toVector() const { return v; }
};
struct Frame {
  Pose7 cam_pose_Tcw, cam_pose_Twc;
  int frame_seq_id = 0;
  double timestamp = 0;
  std::vector<Detection> mmObservations;
  std::vector<Ell*> mpLocalObjects;
};
struct Observation { int label; double bbox[4]; double rate; Frame* pFrame; };
using Observations = std::vector<Observation*>;
struct Observation3D { Ell* pObj; Frame* pFrame; };

struct GraphInfo { int frame, objects, vertices, edges_2d, valid_2d, invalid_2d, edges_3d, gravity, lm_iterations; double chi2_initial, chi2_final; };

// DataAssociationSolver (src/core/DataAssociation.cpp:16-135): only the 3-D mode exists.  Cost = distance between the centre
// of a map ellipsoid and the centre of the frame's single-frame ellipsoid (moved to the world); rows are served in turn, each
// takes its cheapest column if that is below 1 m and blocks it for the later rows, otherwise a new instance id is created.
// As in the reference the value returned for a match is the COLUMN of the map ellipsoid in ascending-instance order.
class DataAssociationSolver {
 public:
  std::vector<int> Solve(const Frame* f, const std::map<int, Ell*>& map_ells) {
    const size_t n = f->mmObservations.size();
    std::vector<int> assoc(n, 0), rows;
    for (size_t i = 0; i < n; ++i) {
      if (i < f->mpLocalObjects.size() && f->mpLocalObjects[i] != nullptr) rows.push_back((int)i);
      else assoc[i] = -1;
    }
    if (rows.empty()) return assoc;
    std::vector<std::vector<double>> cost;
    for (int i : rows) {
      Vec7 pose;
      for (int k = 0; k < 7; ++k) pose[k] = f->mpLocalObjects[i]->v[k];
      const Vec7 w = se3_mul(f->cam_pose_Twc.v, pose);   // transform_from(campose_wc)
      std::vector<double> row;
      for (const auto& kv : map_ells) {
        const double dx = kv.second->v[0] - w[0], dy = kv.second->v[1] - w[1], dz = kv.second->v[2] - w[2];
        row.push_back(std::sqrt(dx * dx + dy * dy + dz * dz));
      }
      cost.push_back(row);
    }
    const std::vector<int> m = SolveCostMat(cost);
    for (size_t r = 0; r < rows.size(); ++r) assoc[rows[r]] = m[r];
    return assoc;
  }
  std::vector<int> SolveCostMat(std::vector<std::vector<double>> cost) {
    const double kDisThresh = 1.0;
    std::vector<int> out(cost.size());
    for (size_t i = 0; i < cost.size(); ++i) {
      if (cost[i].empty()) { out[i] = next_instance_++; continue; }
      size_t best = 0;
      for (size_t c = 1; c < cost[i].size(); ++c) if (cost[i][c] < cost[i][best]) best = c;
      if (cost[i][best] < kDisThresh) {
        out[i] = (int)best;
        for (auto& row : cost) row[best] = 999;
      } else out[i] = next_instance_++;
    }
    return out;
  }

 private:
  int next_instance_ = 0;
};

template <class Backend>
class Tracker {
 public:
  Tracker(Backend& be, const Settings& s) : be_(be), s_(s) {}
  // Tracking::ProcessGroundPlaneEstimation's outcome (world frame); until then no ellipsoid is extracted (Tracking.cpp:316)
  void SetGroundPlane(const double plane[4]) { for (int k = 0; k < 4; ++k) ground_[k] = plane[k]; ground_set_ = true; ground_state_ = 2; }
  // Tracking::OpenGroundPlaneEstimation (Tracking.cpp:690-702): the supporting plane is extracted from the depth image of the
  // first frame on which the extraction succeeds
  void OpenGroundPlaneEstimation() { ground_state_ = 1; ground_set_ = false; }
  int GetGroundPlaneEstimationState() const { return ground_state_; }
  const double* ground_plane() const { return ground_; }
  int ground_frame() const { return ground_frame_; }

  // Tracking::GrabPoseAndObjects
  bool Grab(double timestamp, const Vec7& pose_Twc, const std::vector<Detection>& dets, const uint16_t* depth, int w, int h) {
    Frame* f = new Frame();
    f->timestamp = timestamp;
    f->cam_pose_Twc.v = pose_Twc;
    f->cam_pose_Tcw.v = se3_inverse(pose_Twc);
    f->frame_seq_id = (int)frames_.size();
    f->mmObservations = dets;
    frames_.push_back(f);
    if (!UpdateObjectObservation(f, depth, w, h)) return false;
    if (!JudgeInitialization()) return false;
    if (s_.optimization && !map_order_.empty()) {
      if (!Optimize()) return false;
      RefreshObjectHistory();
    }
    return true;
  }

  bool SaveObjects(const std::string& path) const {   // System::SaveObjectsToFile: instance + 10-vector per row, insertion order
    std::vector<std::vector<double>> rows;
    for (int inst : map_order_) {
      const Ell* e = map_.at(inst);
      std::vector<double> r{(double)e->miInstanceID};
      r.insert(r.end(), e->v.begin(), e->v.end());
      rows.push_back(r);
    }
    return save_number_table(rows, path);
  }
  bool SaveObjectHistory(const std::string& path) const {   // Tracking.cpp:855-896
    std::ofstream out(path);
    if (!out) return false;
    out << history_.size() << std::endl;
    for (const auto& kv : history_) {
      out << kv.first << " " << kv.second.size() << std::endl;
      for (const auto& vec : kv.second)
        for (size_t i = 0; i < vec.size(); ++i) out << vec[i] << (i + 1 == vec.size() ? "\n" : " ");
    }
    return true;
  }
  const std::vector<GraphInfo>& graph_log() const { return log_; }
  const std::map<int, Ell*>& map_ellipsoids() const { return map_; }
  const std::vector<Frame*>& frames() const { return frames_; }
  const std::vector<double>& last_optimized_cams() const { return last_cams_; }   // SLAM mode: Tcw of every frame after the last run
  int fits_attempted = 0, fits_ok = 0;

 private:
  // Tracking::ProcessGroundPlaneEstimation (Tracking.cpp:716-799) without the viewer and the manual check
  bool ProcessGroundPlaneEstimation(Frame* f, const uint16_t* depth, int w, int h) {
    const double intr[5] = {s_.fx, s_.fy, s_.cx, s_.cy, s_.scale};
    double pc[4];
    int ok = 0;
    if (be_.ground_plane(depth, w, h, intr, &s_.plane, pc, &ok) != 0) return false;
    if (!ok) return true;                                   // " * Estimate Ground Plane Fails ": tried again on the next frame
    // g2o::plane::transform(Twc) (src/core/Plane.cpp:117-122: pi' = Twc^-T pi): n' = R n, d' = d - t . n'
    const Vec7& T = f->cam_pose_Twc.v;
    double n[3];
    q_rot(&T[3], pc, n);
    ground_[0] = n[0]; ground_[1] = n[1]; ground_[2] = n[2];
    ground_[3] = pc[3] - (T[0] * n[0] + T[1] * n[1] + T[2] * n[2]);
    ground_set_ = true; ground_state_ = 2; ground_frame_ = f->frame_seq_id;
    return true;
  }
  bool UpdateObjectObservation(Frame* f, const uint16_t* depth, int w, int h) {
    // 1.1 ground-plane estimation (Tracking.cpp:498-499)
    if (ground_state_ == 1 && !ProcessGroundPlaneEstimation(f, depth, w, h)) return false;
    // 1.2 single-frame ellipsoid estimation (UpdateDepthEllipsoidEstimation)
    if (s_.depth_ellipsoid) {
      for (const Detection& d : f->mmObservations) {
        double m[4] = {d.x1, d.y1, d.x2, d.y2};
        const bool is_border = border_filter(m, s_.rows, s_.cols, s_.border_pixels, s_.length_limit);
        bool c3 = false;
        if (s_.with_association && (int)std::lround(d.instance) < 0) c3 = true;
        Ell* extracted = nullptr;
        if (!is_border && ground_set_ && !c3) {
          const double intr[5] = {s_.fx, s_.fy, s_.cx, s_.cy, s_.scale};
          double e10[10], prob = 0;
          int state = 0;
          esl_fit_params p = s_.fit;
          p.symmetry_open = s_.symmetry ? 1 : 0;
          p.depth_scale = s_.scale;
          ++fits_attempted;
          if (be_.fit(depth, w, h, m, (int)std::lround(d.label), f->cam_pose_Twc.v.data(), intr, ground_, &p, e10, &prob, &state) != 0) return false;
          if (state == 0) {
            ++fits_ok;
            extracted = new Ell();
            for (int k = 0; k < 10; ++k) extracted->v[k] = e10[k];
            extracted->prob = prob;
          }
        }
        f->mpLocalObjects.push_back(extracted);
      }
    }
    // 1.3 data association: the instance column of the clip (GetMannualAssociation) or the nearest-centre solver
    std::vector<int> assoc;
    if (s_.with_association) for (const Detection& d : f->mmObservations) assoc.push_back((int)std::lround(d.instance));
    else assoc = da_.Solve(f, map_);
    const std::vector<bool> key = CheckKeyObservations(f, assoc);
    // Update3DObservationDataAssociation
    if (s_.depth_ellipsoid)
      for (size_t i = 0; i < assoc.size(); ++i) {
        if (f->mpLocalObjects[i] == nullptr || assoc[i] < 0) continue;
        if (!key[i]) { f->mpLocalObjects[i] = nullptr; continue; }
        obs3d_[assoc[i]].push_back(Observation3D{f->mpLocalObjects[i], f});
        f->mpLocalObjects[i]->miInstanceID = assoc[i];
      }
    // [2] 2-D observations
    for (size_t i = 0; i < assoc.size(); ++i) {
      const Detection& d = f->mmObservations[i];
      if (assoc[i] < 0 || !key[i]) continue;
      double m[4] = {d.x1, d.y1, d.x2, d.y2};
      if (border_filter(m, s_.rows, s_.cols, s_.border_pixels, s_.length_limit)) continue;
      Observation* ob = new Observation{(int)d.label, {m[0], m[1], m[2], m[3]}, d.rate, f};
      obs_[assoc[i]].push_back(ob);
    }
    return true;
  }

  // checkKeyFrameForInstances: a new observation of an instance counts when the camera moved since its last one
  std::vector<bool> CheckKeyObservations(const Frame* cur, const std::vector<int>& assoc) const {
    const double dis_thr = s_.keyframe_check_close ? 0.0 : 0.4, ang_thr = s_.keyframe_check_close ? 0.0 : M_PI / 180.0 * 15;
    std::vector<bool> out(assoc.size(), false);
    for (size_t i = 0; i < assoc.size(); ++i) {
      if (assoc[i] < 0) continue;
      const auto it = obs_.find(assoc[i]);
      if (it == obs_.end()) { out[i] = true; continue; }
      const Vec7 diff = se3_mul(se3_inverse(cur->cam_pose_Twc.v), it->second.back()->pFrame->cam_pose_Twc.v);
      const double dis = std::sqrt(diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2]);
      const double vn = std::sqrt(diff[3] * diff[3] + diff[4] * diff[4] + diff[5] * diff[5]);
      const double angle = 2 * std::atan2(vn, std::fabs(diff[6]));   // Eigen::AngleAxisd(quaternion).angle()
      out[i] = dis > dis_thr || angle > ang_thr;
    }
    return out;
  }

  bool JudgeInitialization() {
    // 1. SVD from the 2-D observations once there are enough of them
    for (const auto& kv : obs_) {
      if (map_.count(kv.first)) continue;
      const Observations& obs = kv.second;
      if ((int)obs.size() < s_.min_init_frames) continue;
      std::vector<double> poses, boxes;
      for (const Observation* ob : obs) {
        poses.insert(poses.end(), ob->pFrame->cam_pose_Twc.v.begin(), ob->pFrame->cam_pose_Twc.v.end());
        boxes.insert(boxes.end(), ob->bbox, ob->bbox + 4);
      }
      const double K[4] = {s_.fx, s_.fy, s_.cx, s_.cy};
      double e10[10];
      int ok = 0;
      if (be_.init_quadric(poses.data(), boxes.data(), (int)obs.size(), K, s_.rows, s_.cols, e10, &ok) != 0) return false;
      if (ok) {
        Ell* e = new Ell();
        for (int k = 0; k < 10; ++k) e->v[k] = e10[k];
        e->miLabel = obs[0]->label;
        e->miInstanceID = kv.first;
        AddEllipsoid(e);
      }
    }
    // 2. otherwise from the latest single-frame ellipsoid of the instance (camera frame -> world)
    if (s_.depth_ellipsoid)
      for (const auto& kv : obs3d_) {
        if (map_.count(kv.first)) continue;
        const Observation3D& o3 = kv.second.back();
        Ell* e = new Ell(*o3.pObj);
        Vec7 pose;
        for (int k = 0; k < 7; ++k) pose[k] = o3.pObj->v[k];
        const Vec7 w = se3_mul(o3.pFrame->cam_pose_Twc.v, pose);   // ellipsoid::transform_from
        for (int k = 0; k < 7; ++k) e->v[k] = w[k];
        AddEllipsoid(e);
      }
    return true;
  }
  void AddEllipsoid(Ell* e) { map_[e->miInstanceID] = e; map_order_.push_back(e->miInstanceID); }

  bool Optimize() {   // Optimizer::GlobalObjectGraphOptimization through the adapter's flattening
    const bool grav = ground_set_ && s_.gravity_open;
    esl_adapter::Options opt;
    opt.slam_mode = s_.slam_mode; opt.check_visibility = s_.check_visibility; opt.rows = s_.rows; opt.cols = s_.cols;
    esl_adapter::FlatGraph fg = esl_adapter::Flatten(frames_, map_, obs_, s_.scale_3d, grav, opt.slam_mode);
    const double K[4] = {s_.fx, s_.fy, s_.cx, s_.cy};
    esl_graph g = esl_adapter::MakeGraph(fg, K, ground_set_ ? ground_ : nullptr, s_.gravity_scale, opt);
    esl_lm_report rep;
    if (be_.optimize(&g, fg.cams.data(), fg.objs.data(), &rep) != 0) return false;
    for (size_t o = 0; o < fg.instance_of_obj.size(); ++o) {
      Ell* e = map_[fg.instance_of_obj[o]];
      for (int k = 0; k < 10; ++k) e->v[k] = fg.objs[o * 10 + k];
    }
    // as in the reference the optimised camera poses are NOT written back to the frames (Optimizer.cpp:294-306 copies the
    // ellipsoids only); the last run's poses are kept for the output file cameras_slam.txt
    if (s_.slam_mode) last_cams_ = fg.cams;
    log_.push_back(GraphInfo{(int)frames_.size() - 1, g.n_objs, g.n_cams + g.n_objs, g.n_bbox, rep.n_bbox_valid, rep.n_bbox_dropped, g.n_e3d, g.n_grav,
                             rep.iterations, rep.chi2_initial, rep.chi2_final});
    return true;
  }

  void RefreshObjectHistory() {
    for (const auto& kv : map_) {
      const int n_obs = obs_.count(kv.first) ? (int)obs_.at(kv.first).size() : 0;
      std::vector<double> row{(double)n_obs, 1.0};
      const auto mv = to_minimal(kv.second->v);
      row.insert(row.end(), mv.begin(), mv.end());
      auto& h = history_[kv.first];
      if (!h.empty() && (int)std::lround(h.back()[0]) == n_obs) h.back() = row;
      else h.push_back(row);
    }
  }

  Backend& be_;
  Settings s_;
  DataAssociationSolver da_;
  double ground_[4] = {0, 0, 1, 0};
  bool ground_set_ = false;
  int ground_state_ = 0, ground_frame_ = -1;
  std::vector<Frame*> frames_;
  std::map<int, Observations> obs_;
  std::map<int, std::vector<Observation3D>> obs3d_;
  std::map<int, Ell*> map_;
  std::vector<int> map_order_;
  std::map<int, std::vector<std::vector<double>>> history_;
  std::vector<GraphInfo> log_;
  std::vector<double> last_cams_;
};

// Example/interface/rgbd.cpp: every frame of the clip through the tracker, then objects.txt / object_history.txt and the
// per-optimisation graph summaries (graph_log.txt: one row per GlobalObjectGraphOptimization call)
template <class Backend>
int run_clip(Backend& be, const std::string& dataset_dir, const std::string& out_dir, const double ground[4], Settings s) {
  Dataset ds;
  if (!ds.load(dataset_dir)) { std::fprintf(stderr, "cannot read dataset %s\n", dataset_dir.c_str()); return 2; }
  Tracker<Backend> tr(be, s);
  if (ground) tr.SetGroundPlane(ground); else tr.OpenGroundPlaneEstimation();   // System.cpp:46
  std::vector<uint16_t> depth;
  int n_valid = 0;
  for (int i = 0; i < ds.size(); ++i) {
    int w = 0, h = 0;
    Vec7 pose;
    if (!ds.frame(i, depth, w, h, pose)) continue;   // rgbd.cpp:61-75: invalid frames are skipped
    ++n_valid;
    if (!tr.Grab(std::atof(ds.stamp(i).c_str()), pose, ds.detections(i), depth.data(), w, h)) { std::fprintf(stderr, "backend failure in frame %d\n", i); return 3; }
  }
  std::string od = out_dir;
  if (!od.empty() && od.back() != '/') od.push_back('/');
  tr.SaveObjects(od + "objects.txt");
  tr.SaveObjectHistory(od + "object_history.txt");
  std::vector<std::vector<double>> rows;
  for (const GraphInfo& gi : tr.graph_log())
    rows.push_back({(double)gi.frame, (double)gi.objects, (double)gi.vertices, (double)gi.edges_2d, (double)gi.valid_2d, (double)gi.invalid_2d,
                    (double)gi.edges_3d, (double)gi.gravity, (double)gi.lm_iterations, gi.chi2_initial, gi.chi2_final});
  save_number_table(rows, od + "graph_log.txt");
  if (s.slam_mode) {   // Tcw of every frame as the last optimisation left them (the reference drops them)
    std::vector<std::vector<double>> crow;
    const std::vector<double>& lc = tr.last_optimized_cams();
    for (size_t i = 0; i + 7 <= lc.size(); i += 7) crow.push_back(std::vector<double>(lc.begin() + i, lc.begin() + i + 7));
    save_number_table(crow, od + "cameras_slam.txt");
  }
  if (tr.GetGroundPlaneEstimationState() == 2) {
    const double* gp = tr.ground_plane();
    save_number_table({{(double)tr.ground_frame(), gp[0], gp[1], gp[2], gp[3]}}, od + "ground_plane.txt");
  }
  std::printf("frames %d (valid %d), fits %d / %d ok, objects %zu, optimisations %zu\n", ds.size(), n_valid, tr.fits_ok, tr.fits_attempted,
              tr.map_ellipsoids().size(), tr.graph_log().size());
  return 0;
}

}  // namespace esl_harness
