// esl_slam.hpp — SLAM-mode (free cameras) normal equations, Schur complement and dense solve.
#pragma once
#include "esl_ctx.hpp"

namespace esl {
int slam_alloc(esl_ctx* c);
int slam_linearize(esl_ctx* c);
int slam_build_reduced(esl_ctx* c, double lambda, void** dev_ptr, int64_t* n);
int slam_try_step(esl_ctx* c, double lambda);
}  // namespace esl
