// esl_slam.hpp — SLAM-mode (free cameras) normal equations, Schur complement and dense solve.
#pragma once
#include "esl_ctx.hpp"

namespace esl {
int slam_alloc(esl_ctx* c);
void slam_forget(esl_ctx* c);    // the graph goes away: interior pointers of the SLAM blobs are dropped, the blobs stay (free_graph)
void slam_release(esl_ctx* c);   // + the blobs themselves (esl_ctx_destroy)
void slam_trim(esl_ctx* c, bool keep_lists);   // esl_ctx_trim: the solver blobs (and, without free cameras, the list blob) are released
int slam_linearize(esl_ctx* c);
// full_sum (sharded runs): true = every rank ends up with the SUM of the shards' partial systems (all-reduce: callers that read
// S itself -- esl_lm_reduced_system, the residual diagnostic); false = the form the factorisation that follows wants (per-panel
// reduce to the panel's owner when the distributed factorisation is on, else the all-reduce)
int slam_build_reduced(esl_ctx* c, double lambda, bool full_sum, void** dev_ptr, int64_t* n);
int slam_try_step(esl_ctx* c, double lambda);
void slam_release_runtime(esl_ctx* c);   // the context's CholRuntime (esl_ctx_destroy)
}  // namespace esl
