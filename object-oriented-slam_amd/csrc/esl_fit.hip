// esl_fit.hip — single-frame fit + quadric initialisation entry points (kernels land in a later milestone)
#include "esl_ctx.hpp"
extern "C" {
void esl_fit_params_default(esl_fit_params* p) {
  p->stride = 3; p->depth_scale = 5000; p->depth_min = 0.1; p->depth_max = 6.0; p->voxel_leaf = 0.01;
  p->plane_dist = 0.05; p->cluster_tolerance = 0.02; p->min_cluster_size = 100; p->center_dis = 0.5;
  p->symmetry_open = 1; p->symmetry_grid = 0.1; p->symmetry_sigma = 0.1; p->symmetry_lm_iters = 5;
}
int esl_fit_frame(esl_ctx*, const uint16_t*, int32_t, int32_t, const double*, const int32_t*, int32_t, const double*,
                  const double*, const double*, const esl_fit_params*, double*, double*, int32_t*) {
  esl::set_error("esl_fit_frame: not built yet");
  return ESL_ERR_STATE;
}
}
