// esl_slam.hip — placeholder until the Schur path lands (next milestone)
#include "esl_slam.hpp"
namespace esl {
int slam_alloc(esl_ctx*) { return ESL_OK; }
int slam_linearize(esl_ctx*) { set_error("SLAM mode not built yet"); return ESL_ERR_STATE; }
int slam_build_reduced(esl_ctx*, double, void**, int64_t*) { set_error("SLAM mode not built yet"); return ESL_ERR_STATE; }
int slam_try_step(esl_ctx*, double) { set_error("SLAM mode not built yet"); return ESL_ERR_STATE; }
}  // namespace esl
