// tracking_calls.cpp — the call sites of reference src/core/Tracking.cpp that touch the four replaced classes
// (:121-122 construction, :226 optimisation, :299 ClearPointCloudList, :329 EstimateLocalEllipsoid, :338 GetResult,
// :351 GetSymmetryOutputData, :590-593 initializeQuadric + getInitializeResult, :643-650 OpenDepthEllipsoid,
// :692-699/:720 ground-plane estimation, :779/:784 ground plane), linked against adapter/*.cpp built with -DESL_BUILD_IN_REFERENCE_TREE and the stand-in
// headers of this directory.  tests/test_adapter_link.py builds it; without arguments it only proves that everything
// links and that a machine without a HIP device gets the reference's failure pattern (flags false, nothing thrown);
// with a scene file it replays a small Tracking-shaped sequence on the GPU and prints what the classes returned.
#include <cstdio>
#include <fstream>
#include <set>

#include "esl_ref_surface.hpp"

using namespace EllipsoidSLAM;

static void print10(const char* tag, int id, const Vector10d& v) {
  std::printf("%s %d", tag, id);
  for (int k = 0; k < 10; ++k) std::printf(" %.17g", v[k]);
  std::printf("\n");
}

int main(int argc, char** argv) {
  Config::values()["Optimizer.Edges.3DEllipsoid.Scale"] = 10000;
  Config::values()["Optimizer.Edges.GravityPrior.Open"] = 1;
  Config::values()["Optimizer.Edges.GravityPrior.Scale"] = 100;
  Config::values()["EllipsoidExtractor_DEPTH_RANGE"] = 6;
  Config::values()["EllipsoidExtraction.Euclidean.ClusterTolerance"] = 0.02;
  Config::values()["EllipsoidExtraction.Euclidean.MinClusterSize"] = 100;
  Config::values()["EllipsoidExtraction.Euclidean.CenterDis"] = 0.5;
  Config::values()["EllipsoidExtraction.Symmetry.GridSize"] = 0.1;
  Config::values()["SymmetrySolver.Sigma"] = 0.1;
  int mRows = 480, mCols = 640;
  Matrix3d mCalib;
  camera_intrinsic mCamera{535.4, 539.2, 320.1, 247.6, 5000.0};
  g2o::plane mGroundPlane;
  mGroundPlane.param = Vector4d(0, 0, 1, 0);
  std::vector<Frame*> mvpFrames;
  std::map<int, Observations> mmObjectObservations;
  struct Local { int frame, inst; Vector10d e; double prob; };
  std::vector<Local> locals;
  bool have_fit = false;
  int fit_label = 0;
  Vector7d fit_Twc;
  Vector4d fit_box;
  cv::Mat fit_depth;
  if (argc > 1) {
    std::ifstream in(argv[1]);
    in >> mCamera.fx >> mCamera.fy >> mCamera.cx >> mCamera.cy >> mCamera.scale >> mRows >> mCols;
    for (int k = 0; k < 4; ++k) in >> mGroundPlane.param[k];
    int F = 0, NB = 0, NL = 0;
    in >> F;
    for (int i = 0; i < F; ++i) {
      Frame* f = new Frame();
      f->frame_seq_id = i;
      for (int k = 0; k < 7; ++k) in >> f->cam_pose_Twc.v[k];
      for (int k = 0; k < 7; ++k) in >> f->cam_pose_Tcw.v[k];
      mvpFrames.push_back(f);
    }
    in >> NB;
    for (int i = 0; i < NB; ++i) {
      int fr, inst;
      Observation* ob = new Observation();
      in >> fr >> inst >> ob->bbox[0] >> ob->bbox[1] >> ob->bbox[2] >> ob->bbox[3] >> ob->rate >> ob->label;
      ob->pFrame = mvpFrames[fr];
      ob->instance = inst;
      mmObjectObservations[inst].push_back(ob);
    }
    in >> NL;
    for (int i = 0; i < NL; ++i) {
      Local l;
      in >> l.frame >> l.inst;
      for (int k = 0; k < 10; ++k) in >> l.e[k];
      in >> l.prob;
      locals.push_back(l);
    }
    int hf = 0;
    in >> hf;
    have_fit = hf != 0;
    if (have_fit) {
      std::string path;
      in >> fit_depth.cols >> fit_depth.rows;
      for (int k = 0; k < 7; ++k) in >> fit_Twc[k];
      in >> fit_box[0] >> fit_box[1] >> fit_box[2] >> fit_box[3] >> fit_label >> path;
      fit_depth.px.resize((size_t)fit_depth.cols * fit_depth.rows);
      std::ifstream raw(path, std::ios::binary);
      raw.read(reinterpret_cast<char*>(fit_depth.px.data()), (std::streamsize)fit_depth.px.size() * 2);
    }
  }
  cv::Mat plane_depth;
  bool have_plane = false;
  if (argc > 1 && have_fit) {   // optional last record: a depth image for the ground-plane estimation
    std::ifstream in(argv[1]);
    std::string tok, path;
    while (in >> tok) if (tok == "GROUNDDEPTH") { in >> plane_depth.cols >> plane_depth.rows >> path; have_plane = true; }
    if (have_plane) {
      plane_depth.px.resize((size_t)plane_depth.cols * plane_depth.rows);
      std::ifstream raw(path, std::ios::binary);
      raw.read(reinterpret_cast<char*>(plane_depth.px.data()), (std::streamsize)plane_depth.px.size() * 2);
    }
  }
  mCalib(0, 0) = mCamera.fx; mCalib(1, 1) = mCamera.fy; mCalib(0, 2) = mCamera.cx; mCalib(1, 2) = mCamera.cy; mCalib(2, 2) = 1;

  // Tracking.cpp:121-122
  Map* mpMap = new Map();
  Initializer* mpInitializer = new Initializer(mRows, mCols);
  Optimizer* mpOptimizer = new Optimizer;
  // Tracking.cpp:643-650 (OpenDepthEllipsoid)
  EllipsoidExtractor* mpEllipsoidExtractor = new EllipsoidExtractor;
  mpEllipsoidExtractor->OpenVisualization(mpMap);
  mpEllipsoidExtractor->OpenSymmetry();
  // a fit before the supporting plane is known must fail cleanly (the reference asserts)
  {
    cv::Mat d;
    d.rows = 4; d.cols = 4; d.px.assign(16, 0);
    Vector4d b(0, 0, 3, 3);
    VectorXd pose(7);
    pose[6] = 1;
    g2o::ellipsoid e = mpEllipsoidExtractor->EstimateLocalEllipsoid(d, b, 28, pose, mCamera);
    std::printf("NOPLANE %d\n", mpEllipsoidExtractor->GetResult() ? 1 : 0);
    (void)e;
  }
  // Tracking.cpp:692-699 (OpenGroundPlaneEstimation) and :720 (ProcessGroundPlaneEstimation)
  Config::values()["Plane.MinSize"] = 200;
  Config::values()["Plane.AngleThreshold"] = 5;
  Config::values()["Plane.DistanceThreshold"] = 0.1;
  PlaneExtractor* pPlaneExtractor = new PlaneExtractor;
  {
    PlaneExtractorParam param;
    param.fx = mCamera.fx; param.fy = mCamera.fy; param.cx = mCamera.cx; param.cy = mCamera.cy; param.scale = mCamera.scale;
    pPlaneExtractor->SetParam(param);
    cv::Mat d = plane_depth;
    if (!have_plane) { d.rows = 32; d.cols = 32; d.px.assign(32 * 32, 5000); }
    g2o::plane groundPlane;
    const bool result = pPlaneExtractor->extractGroundPlane(d, groundPlane);
    std::printf("GROUND %d", result ? 1 : 0);
    if (result) for (int k = 0; k < 4; ++k) std::printf(" %.9g", groundPlane.param[k]);
    std::printf("\nPLANES %zu", pPlaneExtractor->GetCoefficients().size());
    for (auto& c : pPlaneExtractor->GetPoints()) std::printf(" %zu", c.size());
    std::printf("\nDENSE %zu %zu\n", pPlaneExtractor->GetCloudDense() ? pPlaneExtractor->GetCloudDense()->size() : 0,
                pPlaneExtractor->GetPotentialGroundPlanePoints().size());
  }
  // Tracking.cpp:779, 784
  mpEllipsoidExtractor->SetSupportingPlane(&mGroundPlane);
  mpOptimizer->SetGroundPlane(mGroundPlane.param);

  // Tracking.cpp:299-372 (UpdateDepthEllipsoidEstimation)
  mpEllipsoidExtractor->ClearPointCloudList();
  if (have_fit) {
    VectorXd pose(7);
    for (int k = 0; k < 7; ++k) pose[k] = fit_Twc[k];
    g2o::ellipsoid e = mpEllipsoidExtractor->EstimateLocalEllipsoid(fit_depth, fit_box, fit_label, pose, mCamera);
    const bool ok = mpEllipsoidExtractor->GetResult();
    SymmetryOutputData s = mpEllipsoidExtractor->GetSymmetryOutputData();
    std::printf("FITFLAGS %d %d %d %.17g %.17g\n", ok ? 1 : 0, s.result ? 1 : 0, s.result ? s.symmetryType : -1, e.prob, s.result ? s.prob : 0.0);
    print10("FIT", 0, e.toVector());
    if (s.result) {
      std::printf("SYM");
      for (int k = 0; k < 4; ++k) std::printf(" %.17g", s.planeVec[k]);
      for (int k = 0; k < 4; ++k) std::printf(" %.17g", s.planeVec2[k]);
      for (int k = 0; k < 3; ++k) std::printf(" %.17g", s.center[k]);
      std::printf("\n");
    }
  }
  std::printf("CLOUDS %zu %zu %d\n", mpEllipsoidExtractor->GetPointCloudInProcess()->size(), mpEllipsoidExtractor->GetPointCloudDebug()->size(),
              mpMap->deleted_lists);

  // Tracking.cpp:575-603 (UpdateObjectInitialization, SVD part)
  std::set<int> existInstances;
  for (auto iter = mmObjectObservations.begin(); iter != mmObjectObservations.end(); iter++) {
    if (existInstances.find(iter->first) != existInstances.end()) continue;
    Observations obs = iter->second;
    if ((int)obs.size() < 3) continue;
    g2o::ellipsoid e = mpInitializer->initializeQuadric(obs, mCalib);
    e.miInstanceID = iter->first;
    std::printf("INITFLAG %d %d\n", iter->first, mpInitializer->getInitializeResult() ? 1 : 0);
    if (mpInitializer->getInitializeResult()) {
      g2o::ellipsoid* pBox = new g2o::ellipsoid(e);
      mpMap->addEllipsoid(pBox);
      print10("INIT", iter->first, pBox->toVector());
      // the two public helpers nobody calls in the reference, through the same surface
      MatrixXd pose_mat((int)obs.size(), 7), det_mat((int)obs.size(), 5);
      for (int i = 0; i < (int)obs.size(); ++i) {
        const Vector7d pv = obs[i]->pFrame->cam_pose_Twc.toVector();
        for (int k = 0; k < 7; ++k) pose_mat(i, k) = pv[k];
        for (int k = 0; k < 4; ++k) det_mat(i, k) = obs[i]->bbox[k];
        det_mat(i, 4) = obs[i]->rate;
      }
      std::printf("PLANEERR %d %.17g\n", iter->first, mpInitializer->quadricErrorWithPlanes(pose_mat, det_mat, mCalib, *pBox));
    }
  }
  {
    Matrix4d Q;   // Q* of the axis-aligned ellipsoid (0.3, 0.2, 0.5) centred at (1, -2, 0.5)
    const double s2[3] = {0.09, 0.04, 0.25}, t[3] = {1, -2, 0.5};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Q(i, j) = (i == j ? s2[i] : 0.0) - t[i] * t[j];
    for (int i = 0; i < 3; ++i) { Q(i, 3) = -t[i]; Q(3, i) = -t[i]; }
    Q(3, 3) = -1;
    g2o::ellipsoid e = mpInitializer->getEllipsoidFromQStar(Q);
    std::printf("QSTARFLAG %d\n", mpInitializer->getInitializeResult() ? 1 : 0);
    print10("QSTAR", 0, e.toVector());
  }
  // local 3-D observations of the frames (Tracking.cpp:376-413 stores them in pFrame->mpLocalObjects)
  for (const Local& l : locals) {
    g2o::ellipsoid* pe = new g2o::ellipsoid();
    pe->fromVector(l.e);
    pe->prob = l.prob;
    pe->miInstanceID = l.inst;
    mvpFrames[l.frame]->mpLocalObjects.push_back(pe);
  }
  // Tracking.cpp:226
  bool withAssociation = false;
  std::map<int, Vector10d> before;   // the estimates the optimisation starts from (it overwrites the map's ellipsoids in place)
  for (auto& kv : mpMap->GetAllEllipsoidsMap()) before[kv.first] = kv.second->toVector();
  mpOptimizer->GlobalObjectGraphOptimization(mvpFrames, mpMap, mRows, mCols, mCalib, mmObjectObservations, true, withAssociation);
  for (auto& kv : mpMap->GetAllEllipsoidsMap()) print10("OPT", kv.first, kv.second->toVector());
  // the same call with the reference's bSLAM_mode branch (Optimizer.cpp:126-158; config key Optimizer.SLAMMode here) and the
  // last argument of the signature, check_visibility (Optimizer.h:20-22), from the same start
  for (auto& kv : mpMap->GetAllEllipsoidsMap()) kv.second->fromVector(before[kv.first]);
  Config::values()["Optimizer.SLAMMode"] = 1;
  mpOptimizer->GlobalObjectGraphOptimization(mvpFrames, mpMap, mRows, mCols, mCalib, mmObjectObservations, true, withAssociation, true);
  for (auto& kv : mpMap->GetAllEllipsoidsMap()) print10("OPTSLAM", kv.first, kv.second->toVector());
  Config::values()["Optimizer.SLAMMode"] = 0;
  std::printf("LINK-OK\n");
  return 0;
}
