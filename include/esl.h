/*
 * esl.h — C-ABI of the MI355X-native EllipsoidSLAM hot-path backend.
 *
 * Drop-in boundary (SURVEY.md §8 b).  The reference has no FFI; its boundary is the public
 * surface of three C++ classes that `Tracking` instantiates directly.  Each entry point
 * below names the reference member it replaces:
 *
 *   esl_optimize*        <- EllipsoidSLAM::Optimizer::GlobalObjectGraphOptimization
 *                           (reference include/core/Optimizer.h:20-23, src/core/Optimizer.cpp:83-317)
 *                           + g2o::SparseOptimizer::optimize / OptimizationAlgorithmLevenberg::solve
 *                           (Thirdparty/g2o/g2o/core/sparse_optimizer.cpp:354-419,
 *                            core/optimization_algorithm_levenberg.cpp:61-164)
 *   esl_lm_*  (step API) <- the pieces of the above that exchange scalars between shards:
 *                           computeActiveErrors/activeRobustChi2 (sparse_optimizer.cpp:61-114),
 *                           BlockSolver::buildSystem/setLambda/solve (core/block_solver.hpp:354-604),
 *                           computeLambdaInit/computeScale (optimization_algorithm_levenberg.cpp:166-189)
 *   esl_fit_frame        <- EllipsoidSLAM::EllipsoidExtractor::EstimateLocalEllipsoid
 *                           (reference src/pca/EllipsoidExtractor.h:57-58, .cpp:292-493)
 *   esl_init_quadric     <- EllipsoidSLAM::Initializer::initializeQuadric
 *                           (reference include/core/Initializer.h:47-50, src/core/Initializer.cpp:24-56)
 *
 * Conventions (identical to the reference's own vectors):
 *   SE3 7-vector      = x y z qx qy qz qw        (g2o::SE3Quat::toVector, types/se3quat.h:144-155)
 *   ellipsoid 10-vec  = x y z qx qy qz qw a b c  (g2o::ellipsoid::toVector, src/core/Ellipsoid.cpp:152-157)
 *   camera states are Tcw (world -> camera), ellipsoid poses are object -> world.
 * All arrays are caller-allocated, plain pointers and sizes; no exceptions cross this boundary;
 * every function returns an esl_status.  A context is bound to one HIP device and one stream and
 * is re-entrant per context (not thread-safe within one context).
 *
 * There is NO CPU fallback behind these symbols: without a usable HIP device the compute entry
 * points return ESL_ERR_NO_DEVICE.
 */
#ifndef ESL_H_
#define ESL_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ESL_ABI_VERSION 5   /* 2: esl_lm_params::bbox_residual, esl_fit_frame_ex, esl_init_from_qstar / esl_init_plane_error, esl_graph_append
                             * 3: esl_linear_solver gains ESL_SOLVER_REDUCED_CAMERA / ESL_SOLVER_REDUCED_ELLIPSOID, esl_lm_solver_used,
                             *    esl_lm_solver_stats, esl_comm_set_replicated, ESL_PROF_KINDS 9
                             * 4: ESL_PROF_KINDS 10 (class 9: the dense factorisation alone), esl_ctx_trim, esl_comm_set_replicated refuses a
                             *    mode change under a resident graph, esl_graph_append accepts SLAM-mode graphs (free cameras + odometry)
                             * 5: esl_lm_params::e3d_half_turn (1 = the yaw-hypothesis minimum exactly as Ellipsoid.cpp:92-117 writes it),
                             *    esl_plane_params::max_curvature (PCL's maximum_curvature_ model test; min_inliers is a strict >) */
#define ESL_MAX_TRACE 32

typedef enum {
  ESL_OK = 0,
  ESL_ERR_NO_DEVICE = 1,   /* no HIP device / HIP runtime error on create */
  ESL_ERR_INVALID = 2,     /* bad argument (null pointer, negative size, index out of range) */
  ESL_ERR_HIP = 3,         /* HIP runtime call failed; see esl_last_error() */
  ESL_ERR_STATE = 4,       /* call sequence error (e.g. step API before esl_lm_begin) */
  ESL_ERR_ALLOC = 5
} esl_status;

typedef struct esl_ctx esl_ctx;

/* ---- graph description (SoA, host pointers) ------------------------------------------------
 * Mirrors what Optimizer.cpp:127-279 assembles into g2o objects. */
typedef struct {
  double fx, fy, cx, cy;          /* Kalib (EdgeSE3EllipsoidProj::setKalib, BasicEllipsoidEdges.cpp:114) */
  int32_t n_cams;                 /* VertexSE3Expmap count  (Optimizer.cpp:127-139) */
  int32_t n_objs;                 /* VertexEllipsoid count  (Optimizer.cpp:175-180) */
  const uint8_t* cam_fixed;       /* n_cams flags; NULL = all fixed (mapping mode, Optimizer.cpp:126,135) */

  /* EdgeSE3EllipsoidProj (Optimizer.cpp:201-245): information = weight * I4 */
  int32_t n_bbox;
  const int32_t* bbox_cam;
  const int32_t* bbox_obj;
  const double* bbox_meas;        /* n_bbox x 4 : x1 y1 x2 y2; entries < 5 are ignored (BasicEllipsoidEdges.cpp:109) */
  const double* bbox_weight;      /* n_bbox     : Observation::rate */

  /* EdgeSE3Ellipsoid9DOF (Optimizer.cpp:249-279): information = weight * I9
   * (the caller passes weight = Optimizer.Edges.3DEllipsoid.Scale * prob) */
  int32_t n_e3d;
  const int32_t* e3d_cam;
  const int32_t* e3d_obj;
  const double* e3d_meas;         /* n_e3d x 10 : local (camera-frame) ellipsoid */
  const double* e3d_weight;       /* n_e3d */

  /* EdgeEllipsoidGravityPlanePrior (Optimizer.cpp:183-196): information = grav_weight
   * (the caller passes GravityPrior.Scale^2) */
  int32_t n_grav;
  const int32_t* grav_obj;
  double grav_normal[4];          /* ground plane param; only [0..2] enter the residual */
  double grav_weight;

  /* EdgeSE3Expmap odometry (Optimizer.cpp:142-158; SLAM mode only): information = diag(odom_info) */
  int32_t n_odom;
  const int32_t* odom_i;          /* vertex 0 */
  const int32_t* odom_j;          /* vertex 1 */
  const double* odom_meas;        /* n_odom x 7 */
  const double* odom_info;        /* n_odom x 6 ; NULL = identity */

  /* the optional visibility test of the bbox edges (checkVisibility, Optimizer.cpp:35-81; argument check_visibility of
   * GlobalObjectGraphOptimization, false at the reference's only call site): an edge also needs the ellipsoid in front of its
   * camera, the camera outside the ellipsoid and the projected centre or a corner of the projected box inside the image.
   * Evaluated once at the start state together with the NaN test; failing edges count as dropped. */
  int32_t check_visibility;       /* 0 = off (as shipped) */
  int32_t image_rows, image_cols; /* Optimizer.h:20-23 rows, cols; only read when check_visibility != 0 */
} esl_graph;

typedef enum {
  ESL_JAC_NUMERIC = 0,   /* central differences, delta = numeric_delta (g2o base_binary_edge.hpp:147-197) */
  ESL_JAC_ANALYTIC = 1   /* closed-form Jacobians of the same residuals */
} esl_jacobian_mode;

typedef enum {
  /* All cameras fixed (mapping mode, the shipped setting): 9x9 blocks per ellipsoid.  Free cameras (SLAM mode): one of the two
   * block eliminations below -- both solve EXACTLY the system g2o's LinearSolverDense solves with one pivoted LDLT of the whole
   * free system (solvers/linear_solver_dense.h:65-113; that literal form exists only in the CPU checker, oracle/ ORACLE_DENSE,
   * which the tests compare against); which block is eliminated first is what BlockSolver's setMarginalized decides
   * (core/block_solver.hpp:367-486).  AUTO takes the one with fewer flops that applies. */
  ESL_SOLVER_AUTO = 0,
  /* Schur complement onto the CAMERAS: S = Hcc - sum_o W_o (Hoo + lambda I)^-1 W_o^T, dense FP64-MFMA Cholesky of order
   * 6 (F - 1) -- the reduced camera system of BASELINE.json's north star.  Always applicable; the form sharded runs use. */
  ESL_SOLVER_REDUCED_CAMERA = 1,
  /* Cameras first (round 3): in the reference's SLAM branch the camera block is block TRIDIAGONAL (odometry edges join
   * consecutive frames only, Optimizer.cpp:142-158), so it has a block-bidiagonal Cholesky factor L; the reduced system
   * T = (Hoo + lambda I) - (L^-1 W)^T (L^-1 W) over the ELLIPSOIDS (order 9 N) is formed with an FP64-MFMA rank-6(F-1) update and
   * factored densely.  (9N)^2 6F + (9N)^3/3 flops instead of (6F)^3/3 (C4: 2.1e13 vs 7.2e13); for long chains the camera chain is
   * dissected into 16-camera segments and X = L^-1 W kept sparse (per-segment MFMA products, the dense update only for the
   * separator cameras' rows: C4 3.3e12 flops; esl_lm_solver_stats reports the shape).  Needs every odometry edge to join two free
   * cameras that are neighbours in free-camera order, and either one GPU or the replicated-graph communicator
   * (esl_comm_set_replicated); ESL_ERR_INVALID otherwise. */
  ESL_SOLVER_REDUCED_ELLIPSOID = 2
} esl_linear_solver;

typedef enum {
  ESL_BBOX_REPROJECTION = 0,  /* r = bbox(Q*, P) - z, the reference's EdgeSE3EllipsoidProj (BasicEllipsoidEdges.cpp:90-112) */
  ESL_BBOX_TANGENCY = 1       /* OPTIONAL EXTRA, off for parity: r_k = pi_k^T Q* pi_k for the four planes pi_k = P^T l_k through the
                               * camera centre and the bbox lines (unit normals): the tangency constraint the reference only uses in
                               * its SVD initialiser (Initializer.cpp:147-164, 271-284) as an edge; same weight, same < 5 px rule */
} esl_bbox_residual;

typedef struct {
  int32_t max_iters;        /* 10  (Optimizer.cpp:291) */
  int32_t max_trials;       /* 10  (optimization_algorithm_levenberg.cpp:49) */
  double tau;               /* 1e-5 (optimization_algorithm_levenberg.cpp:45) */
  int32_t jacobian_mode;    /* esl_jacobian_mode */
  double numeric_delta;     /* 1e-9 */
  int32_t linear_solver;    /* esl_linear_solver (SLAM mode only; ignored when all cameras are fixed) */
  int32_t drop_nan_bbox;    /* 1: pre-evaluate bbox edges and drop those with NaN chi2 (Optimizer.cpp:234-243) */
  int32_t bbox_residual;    /* esl_bbox_residual; 0 = the reference */
  int32_t e3d_half_turn;    /* 3-D edge, choice among the four yaw hypotheses (Ellipsoid.cpp:92-117).  0 (default): a hypothesis whose relative
                             * rotation is within ~1.4e-6 rad of a HALF TURN is not eligible -- the reference's log has no branch for theta -> pi
                             * (se3quat.h:229-266: 0/0, decided by the last bit of the trace), its true norm is pi and never the minimum.
                             * 1: the minimum exactly as the reference writes it (minCoeff over the four norms, half turns included): bit-level
                             * faithful where the log is well defined, implementation-defined where the reference itself is (ABI 5) */
} esl_lm_params;

typedef struct {
  int32_t iterations;       /* outer iterations run (SparseOptimizer::optimize return value) */
  int32_t total_trials;     /* sum of inner LM trials */
  int32_t n_bbox_valid;
  int32_t n_bbox_dropped;
  int32_t stop_reason;      /* 0 max_iters, 1 Terminate(trials==max or rho==0), 2 Terminate(nBad>=3), 3 nothing to optimise */
  double chi2_initial;
  double chi2_final;
  double lambda_final;
  int32_t trace_len;
  double trace_chi2[ESL_MAX_TRACE];     /* chi2 after each outer iteration */
  double trace_lambda[ESL_MAX_TRACE];   /* lambda after each outer iteration */
  int32_t trace_trials[ESL_MAX_TRACE];  /* inner trials of each outer iteration */
} esl_lm_report;

/* Partial sums one shard contributes; summed (chi2, scale) or max-ed (max_diag) across shards. */
typedef struct {
  double chi2;
  double max_diag;
  double scale;      /* sum_j x_j (lambda x_j + b_j) */
  int32_t solve_ok;  /* 1 if every local factorisation had positive pivots (AND across shards) */
  int32_t pad;
} esl_lm_partials;

/* ---- context ---------------------------------------------------------------------------------*/
int esl_abi_version(void);
const char* esl_last_error(void);
int esl_device_count(void);
int esl_ctx_create(int device_id, esl_ctx** out);
int esl_ctx_destroy(esl_ctx* ctx);
int esl_ctx_synchronize(esl_ctx* ctx);
/* The context keeps its device buffers in grow-only blobs from graph to graph (no hipMalloc per esl_optimize: Optimizer.cpp:127,166,250
 * rebuilds its whole graph every frame).  esl_ctx_trim releases the SLAM-mode SOLVER blobs (camera-first set / reduced camera
 * system: tens of GB at 10k cameras) -- and the SLAM list blob when the resident graph has no free cameras; whatever a later
 * trial step needs is built again.  Uploading a graph WITHOUT free cameras over a SLAM-mode one trims by itself. */
int esl_ctx_trim(esl_ctx* ctx);
void esl_lm_params_default(esl_lm_params* p);

/* ---- one-shot optimiser (host buffers in, host buffers out) ------------------------------------*/
int esl_optimize(esl_ctx* ctx, const esl_graph* g, double* cams_io /* n_cams x 7 */,
                 double* objs_io /* n_objs x 10 */, const esl_lm_params* p, esl_lm_report* out);

/* ---- resident / step API (graph + states stay in HBM; used by bench and by the sharded driver) --*/
int esl_graph_upload(esl_ctx* ctx, const esl_graph* g);
int esl_states_upload(esl_ctx* ctx, const double* cams, const double* objs);
int esl_states_download(esl_ctx* ctx, double* cams, double* objs);
int esl_optimize_resident(esl_ctx* ctx, const esl_lm_params* p, esl_lm_report* out);
/* device-side copy of the current states (snapshot) and back (restore): lets a caller re-run the
 * optimiser from the same initial estimate without touching PCIe (the reference re-optimises the
 * whole graph every frame, Optimizer.cpp:127,166,250). */
int esl_states_snapshot(esl_ctx* ctx);
int esl_states_restore(esl_ctx* ctx);

/* ---- streaming: extend the device-resident graph instead of rebuilding it (BASELINE configs[4]) --------------------------
 * The reference rebuilds every g2o vertex and edge for every frame (Optimizer.cpp:127, 166, 250).  esl_graph_append adds the
 * observations of a new frame to the graph that is already in HBM: new cameras get the indices n_cams.., new ellipsoids
 * n_objs.. (their states are given; the states of the existing vertices -- e.g. the ellipsoids the previous frame's run
 * optimised -- stay where they are), edge indices refer to the extended numbering.  The result is indistinguishable from
 * esl_graph_upload of the concatenated graph (edges of an ellipsoid keep their arrival order).  Edges are stored sorted by
 * ellipsoid with slack behind every ellipsoid's slice: an append writes the new edges into free slots (one staged copy + two
 * small launches, independent of the size of the graph) and re-lays the arrays out, with doubled slack, only when a slice or an
 * array is full.
 * SLAM mode (ABI 4; the reference's bSLAM_mode branch, Optimizer.cpp:126-158): a delta may bring FREE cameras (new_cam_fixed) and
 * odometry edges; the edge records and the states still go into the resident arrays in place, the camera-side index tables
 * (camera slots, per-camera edge lists, the odometry arrays: a few bytes per edge) are rebuilt from the host's mirror of the layout
 * and shipped as one blob, and the solver's lists follow on the next trial step -- no re-upload, no re-sort of the graph.  One GPU
 * (no communicator). */
typedef struct {
  int32_t n_new_cams; const double* new_cams;   /* n_new_cams x 7 (Tcw) */
  int32_t n_new_objs; const double* new_objs;   /* n_new_objs x 10 */
  int32_t n_bbox; const int32_t* bbox_cam; const int32_t* bbox_obj; const double* bbox_meas; const double* bbox_weight;
  int32_t n_e3d; const int32_t* e3d_cam; const int32_t* e3d_obj; const double* e3d_meas; const double* e3d_weight;
  int32_t n_grav; const int32_t* grav_obj;      /* gravity priors on (new or old) ellipsoids */
  /* ABI 4 -- SLAM mode, all optional (a zero-initialised tail = the mapping-mode delta of ABI 2/3): */
  const uint8_t* new_cam_fixed;                 /* n_new_cams flags; null: the new cameras are fixed */
  int32_t n_odom; const int32_t* odom_i; const int32_t* odom_j;   /* odometry edges (previous frame, current frame), extended numbering */
  const double* odom_meas;                      /* n_odom x 7: Z = Tcw_j * Tcw_i^-1 */
  const double* odom_info;                      /* n_odom x 6 diagonal information, or null = ones */
} esl_graph_delta;
int esl_graph_append(esl_ctx* ctx, const esl_graph_delta* d);
/* sizes of the resident graph (after uploads / appends) */
int esl_graph_sizes(esl_ctx* ctx, int32_t* n_cams, int32_t* n_objs, int32_t* n_bbox, int32_t* n_e3d, int32_t* relayouts);

/* per-kernel timing with HIP events recorded on the context's own stream.
 * kernel ids: 0 linearise, 1 LM trial (solve + retract + chi2), 2 reduced-system build (Schur complement / camera-first: factor of
 *             the camera block + X + the rank-K update), 3 dense Cholesky + solves, 4 reductions / misc, 5 single-frame fit,
 *             6 RCCL all-reduce of the reduced system (sharded SLAM), 7 the MFMA rank-K update T -= X^T X of the camera-first
 *             form alone (nested inside class 2), 8 the block products of the sparse interior rows of X (k_cf_T_sparse, nested
 *             inside class 2; with class 8 present class 7 is the dense update with the separators' rows only), 9 the dense
 *             FACTORISATION of the reduced system alone (nested inside class 3: one k_chol_persist launch from 4,096 unknowns on,
 *             the launch-per-step chain otherwise; class 3 minus class 9 = back-substitutions).
 * esl_profile_enable(ctx, 0) off; 1 = bracket only kernel class 0, and inside esl_optimize_resident's device-driven
 * mapping run only ONE linearisation launch per run (cheap enough to stay on inside a timed region: an event record
 * is a barrier packet between two otherwise back-to-back dispatches); 2 = bracket every launch of every class.
 * esl_profile_get drains the events: count[k] launches, total_ms[k] summed durations. */
#define ESL_PROF_KINDS 10
int esl_profile_enable(esl_ctx* ctx, int enable);
int esl_profile_get(esl_ctx* ctx, int64_t count[ESL_PROF_KINDS], double total_ms[ESL_PROF_KINDS]);

int esl_lm_begin(esl_ctx* ctx, const esl_lm_params* p, int32_t* n_valid, int32_t* n_dropped);
/* residuals at current states -> chi2 ; then H,b -> max_diag */
int esl_lm_linearize(esl_ctx* ctx, esl_lm_partials* out);
/* SLAM mode only: build this shard's reduced camera system for `lambda` and return its device pointer.
 * Layout: (n+1) x n doubles, column-major with leading dimension lda; rows 0..n-1 hold the LOWER
 * triangle of S = Hcc + lambda I - sum_o W_o (Hoo + lambda I)^-1 W_o^T, row n holds b_s^T.
 * A multi-GPU caller may sum the lda*n doubles across shards in place (SURVEY.md §8 e). */
int esl_lm_reduced_system(esl_ctx* ctx, double lambda, void** dev_ptr, int64_t* n, int64_t* lda);
/* SLAM mode diagnostic (tests at sizes no CPU checker reaches): re-builds the reduced camera system for the lambda of
 * the last esl_lm_try_step (the factorisation overwrote it) and returns |S x_c - b_s| / |b_s| for the x_c that step solved. */
int esl_lm_reduced_residual(esl_ctx* ctx, double* rel_residual_out);
/* backup states, solve (H + lambda I) x = b, apply x, recompute chi2; states stay updated */
int esl_lm_try_step(esl_ctx* ctx, double lambda, esl_lm_partials* out);
/* accept != 0: discard backup; accept == 0: restore states from backup */
int esl_lm_commit(esl_ctx* ctx, int accept);
/* which esl_linear_solver the last SLAM-mode trial step of this context ran with (ESL_SOLVER_REDUCED_CAMERA or
 * ESL_SOLVER_REDUCED_ELLIPSOID: what ESL_SOLVER_AUTO resolved to); 0 before any SLAM-mode step */
int esl_lm_solver_used(esl_ctx* ctx, int32_t* solver_out);
/* shape of the camera-first elimination of the resident graph as the last trial step ran it (zeros when it did not):
 * stats[0] form of X: 0 dense rows, 1 sparse with stored per-segment products, 2 sparse, blocks of T straight from the slabs;
 * [1] dissection stride (0: plain chain), [2] separators, [3] segments, [4] flops of the per-segment products (lower block triangles, rows from a block row's first camera on),
 * [5] bytes of the stored products, [6] bytes of the compact slabs, [7] rows of the dense MFMA update (K, padded),
 * [8] flops that update executes when T's ellipsoids are ordered by first camera and the structurally zero rows above the
 *     staircase are skipped tile by tile (0: the closed form n (n + 1) K applies), [9] 1 if T is in that order. */
#define ESL_SOLVER_STATS 10
int esl_lm_solver_stats(esl_ctx* ctx, double stats[ESL_SOLVER_STATS]);

/* inspection (tests, debugging): copy one device array of the current linearisation to the host.
 * which: 0 Hoo (n_objs x 45 packed upper 9x9), 1 bo (n_objs x 9), 2 xo (n_objs x 9, last trial),
 *        3 Hcc (n_free_cams x 36), 4 bc (n_free_cams x 6), 5 xc (n_free_cams x 6, last trial),
 *        6 reduced system ((n+1) x n column-major, lda = round_up(n+1,16): lda*n doubles), 7 trial ellipsoids (n_objs x 10),
 *        8 trial cameras (n_cams x 7), 9 (SLAM mode) the per-edge camera-ellipsoid blocks W = Jc^T Omega Jo as [54][n_bbox + n_e3d]:
 *        entry (a, b) of edge u at [(a * 9 + b) * (n_bbox + n_e3d) + u], u = position of the edge after the stable sort by
 *        ellipsoid (bbox edges first, 3-D edges from n_bbox on); zero for edges of fixed cameras.
 *        count = number of doubles the caller's buffer holds. */
int esl_lm_download(esl_ctx* ctx, int32_t which, double* dst, int64_t count);

/* ---- multi-GPU exchange inside the library (RCCL over xGMI) ---------------------------------------------------
 * One process per GPU.  Rank 0 creates an id (esl_comm_unique_id), the host language ships its 128 bytes to the
 * other ranks (torch.distributed / MPI / a file), every rank calls esl_comm_init.  Afterwards
 * esl_optimize_resident is COLLECTIVE: each rank optimises its own shard (its ellipsoids with all their edges,
 * cameras replicated) under g2o's single global LM control — the LM scalars of every linearisation / trial are
 * exchanged with one ncclAllGather on the context's stream and reduced in rank order on every rank, so all ranks
 * take bit-identical accept/reject decisions (SURVEY.md §8 e).  RCCL is dlopen()ed at esl_comm_init time.
 * SLAM mode (free cameras): the shards additionally ncclAllReduce(sum) the camera blocks Hcc, b_c once per
 * linearisation and the partial reduced camera system S (+ b_s row) once per trial; odometry edges, lambda I on
 * the camera blocks and the camera part of the LM scale are counted on rank 0 only, so the summed system equals
 * the single-GPU one up to summation order.  Every rank then factorises the same S (replicated Cholesky: it is
 * O(n^3) but 0.2-3 s at the C4 size, while the sharded part is the O(edges) linearise + Schur build). */
int esl_comm_unique_id(char out[128]);
int esl_comm_init(esl_ctx* ctx, int32_t n_ranks, int32_t rank, const char id[128]);
/* Same collective semantics over a transport the HOST supplies (MPI, gloo, a test harness): `fn` must sum
 * `count` doubles of `host_buf` in place over all ranks and return 0; every rank calls it in the same order with
 * the same count.  The library stages device buffers through pinned host memory in <= 64 MiB pieces.  Meant for
 * hosts without a GPU-aware transport and for tests; RCCL (esl_comm_init) is the fast path. */
typedef int (*esl_host_allreduce_fn)(void* user, double* host_buf, int64_t count);
int esl_comm_init_host(esl_ctx* ctx, int32_t n_ranks, int32_t rank, esl_host_allreduce_fn fn, void* user);
/* REPLICATED-GRAPH mode of a communicator (round 3; call after esl_comm_init / esl_comm_init_host, on every rank): every rank has
 * uploaded the WHOLE graph (the same esl_graph_upload everywhere: 15 MB at BASELINE configs[3]) instead of a shard of the
 * ellipsoids.  Nothing is summed over ranks then (every rank linearises everything -- 0.4 ms at configs[3] -- and takes the same LM
 * decisions by construction); the ranks divide the DENSE SOLVE of SLAM mode: the outer panels (512 columns) of the reduced system
 * are dealt cyclically, a rank forms only its own panels (camera-first elimination: its share of the rank-6(F-1) MFMA update;
 * reduced camera system: built everywhere, factored distributed), the owner factors a panel and ncclBroadcasts it, every rank
 * updates its own later panels (the broadcasts run on their own stream under those updates; ESL_CHOL_DIST_OVERLAP=0 puts them
 * back on the compute stream), the back-substitution runs replicated -- all ranks end with bit-identical states.  This is the
 * multi-GPU form of ESL_SOLVER_REDUCED_ELLIPSOID (the sharded form below sums shard contributions and uses the reduced camera
 * system).  Mapping-mode runs on such a communicator are simply replicated.
 * CONTRACT: the library cannot tell a shard from a whole graph -- the caller asserts that the graph resident on EVERY rank is the
 * same whole graph (switching the mode on over ellipsoid shards would leave every rank with its own partial sums, silently). */
int esl_comm_set_replicated(esl_ctx* ctx, int replicated);
int esl_comm_destroy(esl_ctx* ctx);

/* host-only helper: balanced partition of ellipsoids (with all their edges) over n_parts shards.
 * part_of_obj receives n_objs entries.  (SURVEY.md §8 e) */
int esl_partition_objects(const esl_graph* g, int32_t n_parts, int32_t* part_of_obj);

/* ---- single-frame fit -----------------------------------------------------------------------*/
typedef struct {
  int32_t stride;               /* 3  (PointCloudFilter.cpp:31-32) */
  double depth_scale;           /* 5000 (TUM3.yaml Camera.scale) */
  double depth_min, depth_max;  /* 0.1, 6 (PointCloudFilter.cpp:38; Config.cpp:29) */
  double voxel_leaf;            /* 0.01 (EllipsoidExtractor.cpp:98) */
  double plane_dist;            /* 0.05 (EllipsoidExtractor.cpp:570) */
  double cluster_tolerance;     /* 0.02 (TUM3.yaml:8) */
  int32_t min_cluster_size;     /* 100  (TUM3.yaml:9) */
  double center_dis;            /* 0.5  (TUM3.yaml:10) */
  int32_t symmetry_open;        /* 1 */
  double symmetry_grid;         /* 0.1 (TUM3.yaml:23) */
  double symmetry_sigma;        /* 0.1 (TUM3.yaml:26) */
  int32_t symmetry_lm_iters;    /* 5 in the reference; 0 = score the 9 hypotheses only */
} esl_fit_params;

void esl_fit_params_default(esl_fit_params* p);

int esl_fit_frame(esl_ctx* ctx, const uint16_t* depth, int32_t width, int32_t height,
                  const double* bboxes /* B x 4 */, const int32_t* labels /* B */, int32_t n_boxes,
                  const double Twc[7], const double intr[5] /* fx fy cx cy scale */,
                  const double ground[4], const esl_fit_params* p,
                  double* ellipsoids_out /* B x 10, camera frame */, double* prob_out /* B */,
                  int32_t* status_out /* B, mirrors miSystemState 0..4 */);

/* same, plus per-box stage counters for tests/profiling (16 doubles per box: [0] in-range samples, [1] voxels at
 * voxel_leaf, [2] after the supporting-plane filter, [3] clusters >= min_cluster_size, [4] chosen cluster size,
 * [5] voxels at symmetry_grid, [6..14] probability of the 9 symmetry hypotheses); debug_out may be NULL */
int esl_fit_frame_debug(esl_ctx* ctx, const uint16_t* depth, int32_t width, int32_t height, const double* bboxes,
                        const int32_t* labels, int32_t n_boxes, const double Twc[7], const double intr[5],
                        const double ground[4], const esl_fit_params* p, double* ellipsoids_out, double* prob_out,
                        int32_t* status_out, double* debug_out);

/* SymmetryOutputData (reference src/symmetry/Symmetry.h:16-32) as EstimateLocalEllipsoid fills it
 * (src/pca/EllipsoidExtractor.cpp:376-393, 415-423) and Tracking consumes it (src/core/Tracking.cpp:351-372):
 * result = 0 when no symmetry estimation ran for the box (symmetry closed, label without a symmetry type, fit failed);
 * planes are in the WORLD frame (A B C D of A x + B y + C z + D = 0, not normalised further), plane2 only for
 * symmetry_type 2 (dual reflection), prob = exp(-cost) of the winning hypothesis, center = centre of the completed
 * (mirrored) cloud in the world frame.  The point-cloud pointers of the reference struct stay on the device. */
typedef struct {
  int32_t result;
  int32_t symmetry_type;   /* 1 reflection, 2 dual reflection (EllipsoidExtractor::LoadSymmetryPrior, :52-79); -1 when result == 0 */
  double plane[4];
  double plane2[4];
  double prob;
  double center[3];
} esl_fit_symmetry;

/* esl_fit_frame + the symmetry outputs (what EllipsoidExtractor::GetSymmetryOutputData() returns, EllipsoidExtractor.h:61);
 * symmetry_out: n_boxes entries or NULL; debug_out as in esl_fit_frame_debug or NULL */
int esl_fit_frame_ex(esl_ctx* ctx, const uint16_t* depth, int32_t width, int32_t height, const double* bboxes,
                     const int32_t* labels, int32_t n_boxes, const double Twc[7], const double intr[5],
                     const double ground[4], const esl_fit_params* p, double* ellipsoids_out, double* prob_out,
                     int32_t* status_out, esl_fit_symmetry* symmetry_out, double* debug_out);

/* ---- ground plane -------------------------------------------------------------------------------*/
/* Replaces EllipsoidSLAM::PlaneExtractor (reference include/src/plane/PlaneExtractor.h:34-77, src/plane/PlaneExtractor.cpp):
 * SetParam's fields + the two constants extractPlanes passes to PCL's normal estimation (:57-58, :74). */
typedef struct esl_plane_params {
  int32_t min_size;               /* Plane.MinSize: smallest segment kept as a plane (PlaneExtractor.cpp:87) */
  double angle_threshold_deg;     /* Plane.AngleThreshold */
  double distance_threshold;      /* Plane.DistanceThreshold */
  int32_t normal_smoothing;       /* setNormalSmoothingSize(10) */
  double max_depth_change_factor; /* setMaxDepthChangeFactor(0.05) */
  int32_t min_inliers;            /* setMinInliers(100) */
  /* ABI 4: the second half of segmentAndRefine (PlaneExtractor.cpp:82) -- PCL's OrganizedMultiPlaneSegmentation::refine: two raster
   * passes in which a model (segment of > min_inliers pixels that passes the curvature test below) absorbs neighbouring pixels that belong to no model and lie within
   * refine_distance of its plane.  Grows the inlier lists (sizes, labels, which candidate is the largest), not the coefficients. */
  int32_t refine;                 /* 1 (default): as the reference; 0: segments only (the round 2-4 behaviour) */
  double refine_distance;         /* PlaneRefinementComparator's distance threshold: 0.02 m, not depth dependent */
  /* ABI 5: which segments PCL's segment() keeps as MODELS -- size > min_inliers (strict) and surface curvature
   * |lambda_min| / trace(cov) of the segment's points < maximum_curvature_ (PCL's default 0.001; the reference does not set it).  A
   * segment that fails either test is an "other" label, which the refinement pass lets neighbouring models absorb.  <= 0: no test. */
  double max_curvature;           /* 0.001 */
} esl_plane_params;
void esl_plane_params_default(esl_plane_params* p);
/* PlaneExtractor::extractGroundPlane(depth, plane) (PlaneExtractor.cpp:107-183): the largest plane segment of the depth image
 * whose normal is within 45 degrees of the camera's y axis, in the camera frame, normalised, camera centre on the positive
 * side.  *ok = 0 when no segment qualifies (the reference returns false).  n_planes (segments of >= min_size pixels) and
 * n_pixels (pixels of the returned segment) may be NULL.  The caller moves the plane to the world with the frame's pose as
 * Tracking::ProcessGroundPlaneEstimation does (src/core/Tracking.cpp:741-744). */
int esl_extract_ground_plane(esl_ctx* ctx, const uint16_t* depth, int32_t width, int32_t height, const double intr[5],
                             const esl_plane_params* p, double plane_out[4], int32_t* ok, int32_t* n_planes, int32_t* n_pixels);

/* PlaneExtractor::extractPlanes + GetCoefficients + GetPoints (PlaneExtractor.cpp:22-104, 113-123): every plane segment of
 * >= max(min_size, min_inliers) pixels, in raster order of each segment's first pixel.  planes_out: max_planes x 4 (camera
 * frame, normalised, d >= 0), sizes_out: pixels per plane; *n_planes = segments found (only the first max_planes are
 * written); labels_out: width x height plane index per pixel (-1: none; indices >= max_planes are reported as found) or NULL.
 * Labels with indices are what GetPoints() needs to rebuild the per-plane clouds on the host. */
int esl_extract_planes(esl_ctx* ctx, const uint16_t* depth, int32_t width, int32_t height, const double intr[5],
                       const esl_plane_params* p, int32_t max_planes, double* planes_out, int32_t* sizes_out,
                       int32_t* n_planes, int32_t* labels_out);

/* ---- SVD quadric initialisation -----------------------------------------------------------------*/
int esl_init_quadric(esl_ctx* ctx, const double* poses_Twc /* n x 7 */, const double* bboxes /* n x 4 */,
                     int32_t n, const double K[4], int32_t rows, int32_t cols, int32_t faithful,
                     double ellipsoid_out[10], double qstar_out[16], int32_t* ok);

/* Initializer::getEllipsoidFromQStar (reference include/core/Initializer.h:53, src/core/Initializer.cpp:186-248): pose and
 * half-axes from a dual quadric Q* (row-major 4x4).  faithful != 0 reproduces the reference's decomposition (4x4
 * eigenvalues for the half-axes), 0 the exact one through Q_33. */
int esl_init_from_qstar(esl_ctx* ctx, const double qstar[16], int32_t faithful, double ellipsoid_out[10], int32_t* ok);
/* Initializer::quadricErrorWithPlanes (include/core/Initializer.h:50, src/core/Initializer.cpp:271-284): sum over the
 * bbox tangent planes pi of (pi^T Q* pi)^2 for the given ellipsoid (10-vector, world frame). */
int esl_init_plane_error(esl_ctx* ctx, const double* poses_Twc /* n x 7 */, const double* bboxes /* n x 4 */, int32_t n,
                         const double K[4], int32_t rows, int32_t cols, const double ellipsoid[10], double* error_out);

/* ---- diagnostics ---------------------------------------------------------------------------------*/
/* Dense FP64-MFMA Cholesky factor + solve (the reduced-camera solver of SLAM mode) on a procedurally generated,
 * strictly diagonally dominant n x n system; returns the time of factor + solve (HIP events) and |A x - b| / |b|.
 * A known-answer test of the solver and the micro-benchmark behind the "mfma" roofline numbers. */
int esl_selftest_cholesky(esl_ctx* ctx, int32_t n, double* ms_out, double* rel_residual_out);
/* test / debug: the static task list of the persistent dense factorisation for an order-n system (outer panels of W 128-panels,
 * `filler` far-update tasks between the chain-dependent groups): 4 int32 per task {type, a, b, c} -- 0: S(panel a, strip b),
 * 1: u(panel a, row tile b, column tile c), 2: U(outer panel a, row tile b, column tile c), 3: one 128 x 64 QUARTER of a rank-128 update
 * u (c = column tile | quarter << 16: the diagonal tiles' updates run as quarters) --, ns[np][nR] = strips of row tile R that
 * panel k solves, meta = {np, n_outer, nR, n_tasks, W}.  Host code only (no device needed), but the list depends on the environment:
 * ESL_CHOL_FUSE=0 puts strips 0, 1 of every panel and the diagonal halves of the quarter tasks back on the list (the chain does
 * them itself by default).  Call with null buffers for the sizes. */
int esl_debug_chol_plan(int32_t n, int32_t W, int32_t filler, int32_t* tasks_out, int64_t cap_tasks, int32_t* ns_out, int64_t cap_ns, int32_t meta_out[5]);

#ifdef __cplusplus
}
#endif
#endif /* ESL_H_ */
