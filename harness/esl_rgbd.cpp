// esl_rgbd.cpp — the reference's Example/interface/rgbd.cpp on the MI355X backend: a TUM-RGB-D style clip (depth images,
// ground-truth poses, one bbox file per frame) through the Tracking-side logic of esl_harness.hpp with every fit, every
// quadric initialisation and every graph optimisation done by libesl_hip.so through the C-ABI of include/esl.h.
//
//   esl_rgbd <dataset_dir> <out_dir> [--ground a b c d] [--jacobian numeric|analytic] [--delta 1e-9] [--no-symmetry] [--sym-iters n] [--auto-association] [--slam-mode] [--check-visibility]
//
// Writes objects.txt (System::SaveObjectsToFile), object_history.txt (Tracking::SaveObjectHistory) and graph_log.txt (one
// row per GlobalObjectGraphOptimization call: frame, objects, vertices, 2-D edges, valid, invalid, 3-D edges, gravity edges,
// LM iterations, chi2 before / after).  The supporting plane is estimated from the depth images (esl_extract_ground_plane; the reference does the same
// with PCL, src/plane/PlaneExtractor.cpp) unless --ground gives it in the world frame.  There is no CPU fallback: without a HIP device this fails.
#include "esl_harness.hpp"

struct HipBackend {
  esl_ctx* ctx = nullptr;
  esl_lm_params lm;
  int fit(const uint16_t* depth, int w, int h, const double box[4], int label, const double Twc[7], const double intr[5], const double ground[4],
          const esl_fit_params* p, double e10[10], double* prob, int* state) {
    int32_t lab = label, st = 0;
    const int rc = esl_fit_frame(ctx, depth, w, h, box, &lab, 1, Twc, intr, ground, p, e10, prob, &st);
    *state = st;
    if (rc) std::fprintf(stderr, "esl_fit_frame: %s\n", esl_last_error());
    return rc;
  }
  int ground_plane(const uint16_t* depth, int w, int h, const double intr[5], const esl_plane_params* p, double plane[4], int* ok) {
    int32_t k = 0;
    const int rc = esl_extract_ground_plane(ctx, depth, w, h, intr, p, plane, &k, nullptr, nullptr);
    *ok = k;
    if (rc) std::fprintf(stderr, "esl_extract_ground_plane: %s\n", esl_last_error());
    return rc;
  }
  int init_quadric(const double* poses, const double* boxes, int n, const double K[4], int rows, int cols, double e10[10], int* ok) {
    double Q[16];
    int32_t k = 0;
    const int rc = esl_init_quadric(ctx, poses, boxes, n, K, rows, cols, /*faithful=*/1, e10, Q, &k);
    *ok = k;
    if (rc) std::fprintf(stderr, "esl_init_quadric: %s\n", esl_last_error());
    return rc;
  }
  int optimize(const esl_graph* g, double* cams, double* objs, esl_lm_report* rep) {
    const int rc = esl_optimize(ctx, g, cams, objs, &lm, rep);
    if (rc) std::fprintf(stderr, "esl_optimize: %s\n", esl_last_error());
    return rc;
  }
};

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s dataset_dir out_dir [--ground a b c d] [--jacobian numeric|analytic] [--delta d] [--no-symmetry] [--sym-iters n]\n", argv[0]); return 1; }
  HipBackend be;
  esl_lm_params_default(&be.lm);
  esl_harness::Settings s;
  esl_fit_params_default(&s.fit);
  double ground[4] = {0, 0, 1, 0};
  bool have_ground = false;   // default: estimated from the depth images as the reference does (System.cpp:46)
  for (int i = 3; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--ground" && i + 4 < argc) { for (int k = 0; k < 4; ++k) ground[k] = std::atof(argv[i + 1 + k]); i += 4; have_ground = true; }
    else if (a == "--jacobian" && i + 1 < argc) { be.lm.jacobian_mode = std::string(argv[++i]) == "analytic" ? ESL_JAC_ANALYTIC : ESL_JAC_NUMERIC; }
    else if (a == "--delta" && i + 1 < argc) be.lm.numeric_delta = std::atof(argv[++i]);
    else if (a == "--no-symmetry") s.symmetry = false;
    else if (a == "--auto-association") s.with_association = false;   // ignore the instance column: DataAssociationSolver
    else if (a == "--sym-iters" && i + 1 < argc) s.fit.symmetry_lm_iters = std::atoi(argv[++i]);
    else if (a == "--slam-mode") s.slam_mode = true;             // Optimizer.cpp:126 bSLAM_mode: frame 0 fixed + odometry edges
    else if (a == "--check-visibility") s.check_visibility = true;
    else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 1; }
  }
  if (esl_ctx_create(0, &be.ctx) != ESL_OK) { std::fprintf(stderr, "esl: %s\n", esl_last_error()); return 4; }
  const int rc = esl_harness::run_clip(be, argv[1], argv[2], have_ground ? ground : nullptr, s);
  esl_ctx_destroy(be.ctx);
  return rc;
}
