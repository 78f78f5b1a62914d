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
//   :127-139 one camera vertex per frame (Tcw), all fixed in mapping mode
//   :166-180 one ellipsoid vertex per instance that exists in the map (ascending instance id)
//   :183-196 gravity prior per ellipsoid when a ground plane is set, information = Scale^2
//   :201-245 bbox edges only if the instance has > 2 observations; information = rate * I4;
//            NaN pre-check (done on the GPU: esl_lm_params::drop_nan_bbox = 1)
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
};

template <class FramePtrVec, class EllipsoidMap, class ObservationMap>
FlatGraph Flatten(const FramePtrVec& frames, const EllipsoidMap& map_ellipsoids, const ObservationMap& object_observations,
                  double scale_3d, bool gravity_on) {
  FlatGraph f;
  std::map<int, int> obj_index;  // instance id -> ellipsoid vertex index
  for (const auto& fr : frames) {
    const auto v = fr->cam_pose_Tcw.toVector();
    for (int k = 0; k < 7; ++k) f.cams.push_back(v[k]);
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

inline esl_graph MakeGraph(const FlatGraph& f, const double K[4], const double ground[4], double grav_scale) {
  esl_graph g{};
  g.fx = K[0]; g.fy = K[1]; g.cx = K[2]; g.cy = K[3];
  g.n_cams = (int32_t)(f.cams.size() / 7);
  g.n_objs = (int32_t)(f.objs.size() / 10);
  g.cam_fixed = nullptr;  // mapping mode: bSLAM_mode = false (Optimizer.cpp:126)
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

// rows, cols, save_graph, withAssociation are unused by the reference unless check_visibility is set, and Tracking never
// sets it (Tracking.cpp:226; the visibility check of Optimizer.cpp:35-81 is dead code there)
void Optimizer::GlobalObjectGraphOptimization(std::vector<Frame*>& pFrames, Map* pMap, int, int, Matrix3d& mCalib,
                                              std::map<int, Observations>& objectObservations, bool, bool, bool) {
  const double scale3d = Config::Get<double>("Optimizer.Edges.3DEllipsoid.Scale");
  const bool grav = mbGroundPlaneSet && Config::Get<int>("Optimizer.Edges.GravityPrior.Open") == 1;
  const double grav_scale = Config::Get<double>("Optimizer.Edges.GravityPrior.Scale");
  std::map<int, g2o::ellipsoid*> ells = pMap->GetAllEllipsoidsMap();
  esl_adapter::FlatGraph f = esl_adapter::Flatten(pFrames, ells, objectObservations, scale3d, grav);
  const double K[4] = {mCalib(0, 0), mCalib(1, 1), mCalib(0, 2), mCalib(1, 2)};
  double ground[4] = {0, 0, 0, 0};
  if (mbGroundPlaneSet)
    for (int k = 0; k < 4; ++k) ground[k] = mGroundPlaneNormal[k];
  esl_graph g = esl_adapter::MakeGraph(f, K, mbGroundPlaneSet ? ground : nullptr, grav_scale);
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
