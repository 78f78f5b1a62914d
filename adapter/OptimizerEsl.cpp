// OptimizerEsl.cpp — drop-in body for EllipsoidSLAM::Optimizer that routes
// GlobalObjectGraphOptimization through the C-ABI of libesl_hip.so (include/esl.h).
//
// Compile this file INSIDE the reference tree INSTEAD OF src/core/Optimizer.cpp with
// -DESL_BUILD_IN_REFERENCE_TREE (it needs the reference's own headers: Eigen, Frame.h, Map.h, Config.h).
// No Eigen in the development container: tests/test_adapter_flatten.py runs the flattening logic through the
// template below with tiny stand-in types, tests/test_adapter_link.py compiles and LINKS all three adapters against
// compile-only stand-ins of the reference headers (tests/adapter_stubs/) together with Tracking's call sites.
//
// What it mirrors (reference src/core/Optimizer.cpp):
//   :88-90   config keys Optimizer.Edges.3DEllipsoid.Scale / GravityPrior.Open / GravityPrior.Scale
//   :126     bSLAM_mode -- a constant `false` in the reference; here the config key Optimizer.SLAMMode (absent from the
//            shipped yaml => 0 => the shipped mapping mode), esl_adapter::Options::slam_mode in the template below
//   :127-139 one camera vertex per frame (Tcw), all fixed in mapping mode; SLAM mode: only frame 0 fixed (:137)
//   :142-158 SLAM mode: one EdgeSE3Expmap per consecutive frame pair, vertices (prev, curr), measurement
//            curr_Tcw * prev_Tcw^-1 of the INPUT poses (both from cam_pose_Twc.inverse()), information I6
//   :166-180 one ellipsoid vertex per instance that exists in the map (ascending instance id)
//   :183-196 gravity prior per ellipsoid when a ground plane is set, information = Scale^2
//   :201-245 bbox edges only if the instance has > 2 observations; information = rate * I4;
//            NaN pre-check (done on the GPU: esl_lm_params::drop_nan_bbox = 1); the optional visibility test
//            (checkVisibility, :35-81, argument check_visibility with rows / cols) also runs on the GPU:
//            esl_graph::check_visibility / image_rows / image_cols
//   :249-279 one 3-D edge per non-null local ellipsoid of every frame, information = Scale * prob * I9
//   :290-291 optimize(10)
//   :294-306 write the estimates back into the map's ellipsoids IN PLACE (pose, scale, vec_minimal)
#include <cstdint>
#include <map>
#include <vector>

#include "esl.h"

namespace esl_adapter {

// Generic flattening: works on any Frame/Observation/ellipsoid types that expose the members the
// reference's own types have.  `GetVec7(se3)` / `GetVec10(ellipsoid)` are the toVector() calls.
struct FlatGraph {
  std::vector<double> cams, objs;               // F x 7 (Tcw), N x 10
  std::vector<int> instance_of_obj;             // N
  std::vector<int32_t> bbox_cam, bbox_obj, e3d_cam, e3d_obj, grav_obj;
  std::vector<double> bbox_meas, bbox_weight, e3d_meas, e3d_weight;
  // SLAM mode only (Optimizer.cpp:126-158); empty in mapping mode
  std::vector<uint8_t> cam_fixed;               // F: 1 for frame 0
  std::vector<int32_t> odom_i, odom_j;          // vertex 0 = frame i - 1, vertex 1 = frame i
  std::vector<double> odom_meas;                // (F - 1) x 7
};

// What the reference hard-codes or passes as arguments and a caller may now choose
struct Options {
  bool slam_mode = false;          // bSLAM_mode (Optimizer.cpp:126): cameras free except frame 0 + odometry edges
  bool check_visibility = false;   // argument check_visibility (Optimizer.h:20-22; false at Tracking.cpp:226)
  int rows = 0, cols = 0;          // image size the visibility test uses
};

template <class FramePtrVec, class EllipsoidMap, class ObservationMap>
FlatGraph Flatten(const FramePtrVec& frames, const EllipsoidMap& map_ellipsoids, const ObservationMap& object_observations,
                  double scale_3d, bool gravity_on, bool slam_mode = false) {
  FlatGraph f;
  std::map<int, int> obj_index;  // instance id -> ellipsoid vertex index
  int fi = 0;
  for (const auto& fr : frames) {
    const auto v = fr->cam_pose_Tcw.toVector();
    for (int k = 0; k < 7; ++k) f.cams.push_back(v[k]);
    if (slam_mode) {
      f.cam_fixed.push_back(fi == 0 ? 1 : 0);                   // vSE3->setFixed(frame_index == 0), Optimizer.cpp:137
      if (fi > 0) {                                             // Optimizer.cpp:142-158
        const auto prev_Tcw = frames[fi - 1]->cam_pose_Twc.inverse();
        const auto curr_Tcw = fr->cam_pose_Twc.inverse();
        const auto odom_val = curr_Tcw * prev_Tcw.inverse();
        const auto z = odom_val.toVector();
        for (int k = 0; k < 7; ++k) f.odom_meas.push_back(z[k]);
        f.odom_i.push_back(fi - 1);                             // setVertex(0, vSE3Vertex[frame_index - 1])
        f.odom_j.push_back(fi);                                 // setVertex(1, vSE3Vertex[frame_index])
      }
    }
    ++fi;
  }
  for (const auto& inst_obs : object_observations) {           // std::map => ascending instance id (Optimizer.cpp:166)
    const int instance = inst_obs.first;
    auto it = map_ellipsoids.find(instance);
    if (it == map_ellipsoids.end()) continue;                   // not initialised yet (Optimizer.cpp:169-170)
    const int o = (int)f.instance_of_obj.size();
    obj_index[instance] = o;
    f.instance_of_obj.push_back(instance);
    const auto v = it->second->toVector();
    for (int k = 0; k < 10; ++k) f.objs.push_back(v[k]);
    if (gravity_on) f.grav_obj.push_back(o);
    const auto& obs = inst_obs.second;
    if (obs.size() > 2) {                                       // Optimizer.cpp:201
      for (const auto* ob : obs) {
        f.bbox_cam.push_back(ob->pFrame->frame_seq_id);         // Optimizer.cpp:211-212
        f.bbox_obj.push_back(o);
        for (int k = 0; k < 4; ++k) f.bbox_meas.push_back(ob->bbox[k]);
        f.bbox_weight.push_back(ob->rate);
      }
    }
  }
  int frame_index = 0;
  for (const auto& fr : frames) {                               // Optimizer.cpp:250-279
    for (const auto* obj : fr->mpLocalObjects) {
      if (obj == nullptr) continue;
      auto it = obj_index.find(obj->miInstanceID);
      if (it == obj_index.end()) continue;
      f.e3d_cam.push_back(frame_index);
      f.e3d_obj.push_back(it->second);
      const auto v = obj->toVector();
      for (int k = 0; k < 10; ++k) f.e3d_meas.push_back(v[k]);
      f.e3d_weight.push_back(scale_3d * obj->prob);
    }
    ++frame_index;
  }
  return f;
}

inline esl_graph MakeGraph(const FlatGraph& f, const double K[4], const double ground[4], double grav_scale,
                           const Options& opt = Options()) {
  esl_graph g{};
  g.fx = K[0]; g.fy = K[1]; g.cx = K[2]; g.cy = K[3];
  g.n_cams = (int32_t)(f.cams.size() / 7);
  g.n_objs = (int32_t)(f.objs.size() / 10);
  // mapping mode (bSLAM_mode = false, Optimizer.cpp:126): all cameras fixed = NULL; SLAM mode: the flags Flatten built
  g.cam_fixed = f.cam_fixed.empty() ? nullptr : f.cam_fixed.data();
  g.n_odom = (int32_t)f.odom_i.size();
  g.odom_i = f.odom_i.data(); g.odom_j = f.odom_j.data();
  g.odom_meas = f.odom_meas.data();
  g.odom_info = nullptr;  // identity: inv_sigma = 1 (Optimizer.cpp:152-155)
  g.check_visibility = opt.check_visibility ? 1 : 0;
  g.image_rows = opt.rows; g.image_cols = opt.cols;
  g.n_bbox = (int32_t)f.bbox_cam.size();
  g.bbox_cam = f.bbox_cam.data(); g.bbox_obj = f.bbox_obj.data();
  g.bbox_meas = f.bbox_meas.data(); g.bbox_weight = f.bbox_weight.data();
  g.n_e3d = (int32_t)f.e3d_cam.size();
  g.e3d_cam = f.e3d_cam.data(); g.e3d_obj = f.e3d_obj.data();
  g.e3d_meas = f.e3d_meas.data(); g.e3d_weight = f.e3d_weight.data();
  g.n_grav = (int32_t)f.grav_obj.size();
  g.grav_obj = f.grav_obj.data();
  for (int k = 0; k < 4; ++k) g.grav_normal[k] = ground ? ground[k] : 0.0;
  g.grav_weight = grav_scale * grav_scale;  // inv_sigma^2 (Optimizer.cpp:188-191)
  return g;
}

}  // namespace esl_adapter

#ifdef ESL_BUILD_IN_REFERENCE_TREE
#include <src/config/Config.h>

#include <fstream>

#include "EslAdapterCtx.hpp"
#include "include/core/Optimizer.h"

namespace EllipsoidSLAM {

Optimizer::Optimizer() { mbGroundPlaneSet = false; }
void Optimizer::SetGroundPlane(Vector4d& normal) { mbGroundPlaneSet = true; mGroundPlaneNormal = normal; }

// save_graph and withAssociation are unused by the reference; rows / cols only feed checkVisibility (Optimizer.cpp:35-81),
// which Tracking never switches on (Tracking.cpp:226) -- it is passed through all the same.
// Optimizer.SLAMMode = 1 selects the reference's bSLAM_mode branch (Optimizer.cpp:126-158; a compile-time `false` there): the
// key is absent from the shipped yaml files, cv::FileStorage returns 0 for a missing key, so the default is the shipped mapping
// mode.  As in the reference, only the ellipsoids are written back (Optimizer.cpp:294-306) -- the optimised camera poses stay in
// the solver's vertices and are dropped.
void Optimizer::GlobalObjectGraphOptimization(std::vector<Frame*>& pFrames, Map* pMap, int rows, int cols, Matrix3d& mCalib,
                                              std::map<int, Observations>& objectObservations, bool, bool, bool check_visibility) {
  const double scale3d = Config::Get<double>("Optimizer.Edges.3DEllipsoid.Scale");
  const bool grav = mbGroundPlaneSet && Config::Get<int>("Optimizer.Edges.GravityPrior.Open") == 1;
  const double grav_scale = Config::Get<double>("Optimizer.Edges.GravityPrior.Scale");
  esl_adapter::Options opt;
  opt.slam_mode = Config::Get<int>("Optimizer.SLAMMode") == 1;
  opt.check_visibility = check_visibility; opt.rows = rows; opt.cols = cols;
  std::map<int, g2o::ellipsoid*> ells = pMap->GetAllEllipsoidsMap();
  esl_adapter::FlatGraph f = esl_adapter::Flatten(pFrames, ells, objectObservations, scale3d, grav, opt.slam_mode);
  const double K[4] = {mCalib(0, 0), mCalib(1, 1), mCalib(0, 2), mCalib(1, 2)};
  double ground[4] = {0, 0, 0, 0};
  if (mbGroundPlaneSet)
    for (int k = 0; k < 4; ++k) ground[k] = mGroundPlaneNormal[k];
  esl_graph g = esl_adapter::MakeGraph(f, K, mbGroundPlaneSet ? ground : nullptr, grav_scale, opt);
  esl_lm_params p; esl_lm_params_default(&p);
  esl_lm_report rep;
  {
    std::lock_guard<std::mutex> lock(esl_adapter::CtxMutex());
    esl_ctx* ctx = esl_adapter::SharedCtx();
    if (!ctx) return;
    if (esl_optimize(ctx, &g, f.cams.data(), f.objs.data(), &p, &rep) != ESL_OK) {
      std::cerr << "esl_optimize: " << esl_last_error() << std::endl;  // the reference never throws here
      return;
    }
  }
  // graph summary the reference prints before optimising (Optimizer.cpp:281-288)
  std::cout << " -- GRAPH INFORMATION : " << std::endl;
  std::cout << " * Object Num : " << g.n_objs << std::endl;
  std::cout << " * Vertices: " << g.n_cams + g.n_objs << std::endl;
  std::cout << " * 2d Edges [Valid/Invalid] : " << g.n_bbox << " [" << rep.n_bbox_valid << "/" << rep.n_bbox_dropped << "]" << std::endl;
  std::cout << " * 3d Edges : " << g.n_e3d << std::endl;
  std::cout << " * Gravity edges: " << g.n_grav << std::endl << std::endl;
  for (size_t o = 0; o < f.instance_of_obj.size(); ++o) {      // Optimizer.cpp:294-306: in-place write-back
    g2o::ellipsoid* e = ells[f.instance_of_obj[o]];
    Vector10d v; for (int k = 0; k < 10; ++k) v[k] = f.objs[o * 10 + k];
    e->fromVector(v);                                           // keeps label / colour / instance / prob
  }
  // object list (Optimizer.cpp:308-316): instance, minimal vector, label of every optimised ellipsoid, ascending instance
  std::ofstream out_obj("./object_list.txt");
  for (size_t o = 0; o < f.instance_of_obj.size(); ++o) {
    const g2o::ellipsoid* e = ells[f.instance_of_obj[o]];
    out_obj << f.instance_of_obj[o] << "\t" << e->toMinimalVector().transpose() << "\t" << e->miLabel << std::endl;
  }
  out_obj.close();
}

}  // namespace EllipsoidSLAM
#endif
