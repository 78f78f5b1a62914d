// harness_oracle_main.cpp — TEST INFRASTRUCTURE: the Tracking-side harness (harness/esl_harness.hpp) instantiated with the
// CPU checker (oracle/libesl_oracle.so) instead of libesl_hip.so, so that the GPU run of a whole clip has something to be
// compared with frame by frame.  Built and run only by tests/test_harness.py; the product (harness/esl_rgbd.cpp) never
// links the oracle.
#include "../harness/esl_harness.hpp"
#include "../oracle/esl_oracle.h"

struct OracleBackend {
  esl_lm_params lm;
  int solver = 1;   // per-ellipsoid blocks: bit-identical to the dense LDLT in mapping mode (tests/test_oracle_cross.py)
  int fit(const uint16_t* depth, int w, int h, const double box[4], int label, const double Twc[7], const double intr[5], const double ground[4],
          const esl_fit_params* p, double e10[10], double* prob, int* state) {
    int32_t lab = label, st = 0;
    esl_oracle_fit_frame(depth, w, h, box, &lab, 1, Twc, intr, ground, p, e10, prob, &st, nullptr);
    *state = st;
    return 0;
  }
  int ground_plane(const uint16_t* depth, int w, int h, const double intr[5], const esl_plane_params* p, double plane[4], int* ok) {
    int32_t k = 0;
    const int rc = esl_oracle_extract_ground_plane(depth, w, h, intr, p, plane, &k, nullptr, nullptr, nullptr);
    *ok = k;
    return rc;
  }
  int init_quadric(const double* poses, const double* boxes, int n, const double K[4], int rows, int cols, double e10[10], int* ok) {
    double Q[16];
    return esl_oracle_init_quadric(poses, boxes, n, K, rows, cols, 1, e10, Q, ok), 0;
  }
  int optimize(const esl_graph* g, double* cams, double* objs, esl_lm_report* rep) { return esl_oracle_optimize(g, cams, objs, &lm, solver, rep); }
};

int main(int argc, char** argv) {
  if (argc < 3) return 1;
  OracleBackend be;
  be.lm.max_iters = 10; be.lm.max_trials = 10; be.lm.tau = 1e-5; be.lm.jacobian_mode = 0; be.lm.numeric_delta = 1e-9; be.lm.linear_solver = 0; be.lm.drop_nan_bbox = 1;
  esl_harness::Settings s;
  s.fit.stride = 3; s.fit.depth_scale = 5000; s.fit.depth_min = 0.1; s.fit.depth_max = 6.0; s.fit.voxel_leaf = 0.01; s.fit.plane_dist = 0.05;
  s.fit.cluster_tolerance = 0.02; s.fit.min_cluster_size = 100; s.fit.center_dis = 0.5; s.fit.symmetry_open = 1; s.fit.symmetry_grid = 0.1;
  s.fit.symmetry_sigma = 0.1; s.fit.symmetry_lm_iters = 5;
  double ground[4] = {0, 0, 1, 0};
  bool have_ground = false;   // default: estimated from the depth images as the reference does (System.cpp:46)
  for (int i = 3; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--ground" && i + 4 < argc) { for (int k = 0; k < 4; ++k) ground[k] = std::atof(argv[i + 1 + k]); i += 4; have_ground = true; }
    else if (a == "--delta" && i + 1 < argc) be.lm.numeric_delta = std::atof(argv[++i]);
    else if (a == "--no-symmetry") s.symmetry = false;
    else if (a == "--auto-association") s.with_association = false;
    else if (a == "--sym-iters" && i + 1 < argc) s.fit.symmetry_lm_iters = std::atoi(argv[++i]);
    else if (a == "--slam-mode") { s.slam_mode = true; be.solver = 0; }   // bSLAM_mode + the faithful dense pivoted LDLT of the whole free system
    else if (a == "--check-visibility") s.check_visibility = true;
    else if (a == "--jacobian" && i + 1 < argc) ++i;   // the checker only has g2o's numeric scheme
    else return 1;
  }
  return esl_harness::run_clip(be, argv[1], argv[2], have_ground ? ground : nullptr, s);
}
