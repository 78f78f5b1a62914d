// esl_ctx.hpp — context / device-resident graph shared by the translation units of libesl_hip.so
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/esl.h"
#include "esl_math.hpp"

namespace esl {

void set_error(const std::string& s);
#define ESL_HIP_TRY(expr)                                                                         \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) {                                                                       \
      esl::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                          \
      return ESL_ERR_HIP;                                                                         \
    }                                                                                             \
  } while (0)

// Device-resident graph: edges sorted by ellipsoid (CSR), camera-side CSR for the SLAM-mode gather.
struct DevGraph {
  int n_cams = 0, n_objs = 0;
  int n_bbox = 0, n_e3d = 0, n_odom = 0;   // extents of the edge arrays (appendable layout: including the slack behind the slices)
  int n_bbox_edges = 0;                    // bbox edges that exist (= n_bbox unless the layout has slack)
  double K[4];
  double grav_n[3];
  double grav_w = 0;
  int check_vis = 0, img_rows = 0, img_cols = 0;   // esl_graph::check_visibility / image_rows / image_cols
  int bbox_mode = 0;   // esl_lm_params::bbox_residual of the current run (0 reprojection, 1 plane tangency)
  YawTable yt;
  // per ellipsoid
  int* bb_start = nullptr;   // n_objs + 1
  int* e3_start = nullptr;   // n_objs + 1
  int* gr_cnt = nullptr;     // n_objs : number of gravity-prior edges on this ellipsoid
  // bbox edges, sorted by ellipsoid (stable)
  int* bb_cam = nullptr; int* bb_obj = nullptr;
  double* bb_meas = nullptr; double* bb_w = nullptr;
  unsigned char* bb_valid = nullptr;
  // 3-D edges, sorted by ellipsoid (stable)
  int* e3_cam = nullptr; int* e3_obj = nullptr;
  double* e3_meas = nullptr; double* e3_w = nullptr;
  // odometry
  int* od_i = nullptr; int* od_j = nullptr; double* od_meas = nullptr; double* od_info = nullptr;
  // cameras
  unsigned char* cam_fixed = nullptr;  // n_cams
  int* cam_slot = nullptr;             // n_cams: index among free cameras or -1
  int n_free_cams = 0;
  // camera-side CSR over (bbox, e3d) edges for H_cc / b_c gathers (SLAM mode)
  int* cbb_start = nullptr; int* cbb_edge = nullptr;
  int* ce3_start = nullptr; int* ce3_edge = nullptr;
  int* cod_start = nullptr; int* cod_edge = nullptr;  // odometry edges touching each camera (edge*2 + side)
  // per ellipsoid: unified list of edges whose camera is free (u = bbox index, or n_bbox + 3-D index)
  int* ue_start = nullptr; int* ue_id = nullptr; int* ue_slot = nullptr;
  int n_ue = 0;
  // per free camera (slot): its edges sorted by (ellipsoid, u) -- the deterministic Schur complement walks two of these lists
  int* cu_start = nullptr; int* cu_obj = nullptr; int* cu_id = nullptr;
  int cu_max = 0;   // longest per-camera list (k_slam_schur_pull stages the row camera's Y blocks in LDS when they fit)
  // sharding (SLAM mode): odometry edges, the camera blocks' lambda and the camera part of the LM scale are
  // replicated on every rank and must enter the summed system once -> only shard_rank 0 contributes them
  int shard_rank = 0;
};

struct ChunkTableFwd;
struct LmState {
  bool begun = false;
  esl_lm_params p;
  bool slam = false;          // any free camera
  double lambda_used = 0;
  bool have_trial = false;
};

}  // namespace esl

// device blob + its pinned staging twin + the event of the last staging -> device copy
struct BlobArena { char* dev = nullptr; size_t cap = 0; char* host = nullptr; size_t host_cap = 0; hipEvent_t ev = nullptr; };
struct esl_ctx;
namespace esl {
// RAII event bracket around a group of launches of one kernel class
struct ProfScope {
  esl_ctx* c; int slot;
  ProfScope(esl_ctx* ctx, int kind);
  ~ProfScope();
};
int prof_drain(esl_ctx* c);
void fit_graphs_clear(esl_ctx* c);
void fit_release(esl_ctx* c);
void plane_release(esl_ctx* c);
// all-gather the 8-double dev_scal block of every rank into c->dev_gather (device), ordered on the context's stream
int comm_gather_scalars_device(esl_ctx* c);
// SLAM mode: in-place sum over ranks of a device buffer (RCCL all-reduce); no-op without a communicator
int comm_allreduce_sum(esl_ctx* c, double* dev_buf, size_t count, hipStream_t st = nullptr);
int comm_reduce_sum_root(esl_ctx* c, double* dev_buf, size_t count, int root);   // sum over the ranks, delivered to root
int comm_bcast(esl_ctx* c, double* dev_buf, size_t count, int root, hipStream_t st = nullptr);
// replicated-graph communicator: compare this graph's structure fingerprint across the ranks (once per upload); ESL_ERR_STATE on a mismatch
int comm_check_replicated(esl_ctx* c);
// {sum, max, sum, min} of 4 device scalars over ranks -> host
int comm_reduce4(esl_ctx* c, const double* dev_src4, double out[4]);
}  // namespace esl

struct esl_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  esl::DevGraph g;
  bool graph_loaded = false;
  bool states_loaded = false;
  // states: current and trial (pointer swap on accept)
  double* cams = nullptr; double* cams_trial = nullptr;   // n_cams x 7
  double* objs = nullptr; double* objs_trial = nullptr;   // n_objs x 10
  // mapping-mode system
  double* Hoo = nullptr;      // n_objs x 45 packed upper
  double* bo = nullptr;       // n_objs x 9
  double* xo = nullptr;       // n_objs x 9
  double* obj_part = nullptr; // n_objs x 4 : chi2, maxdiag, scale, ok
  // SLAM-mode system
  double* Hcc = nullptr;      // n_free_cams x 36 diagonal blocks
  double* bc = nullptr;       // n_free_cams x 6
  double* xc = nullptr;       // n_free_cams x 6 (solution of the reduced system)
  double* Wbb = nullptr;      // n_bbox x 54  (H_co blocks, 6x9 row-major)
  double* Abb = nullptr;      // n_bbox x 27 (per-edge camera contributions: 21 packed + 6)
  double* Aod = nullptr;      // n_odom x (2*27 + 36): per-edge Hii, bi, Hjj, bj packed, Hij full
  double* Dinv = nullptr;     // n_objs x 81
  double* Yb = nullptr;       // [EU][54] : W D^-1, one record per edge
  double* Wt = nullptr;       // [EU][9][6] : W as per-edge records (k_slam_schur_pull)
  double* Tb = nullptr;       // [6][EU] : Y_e b_o per edge (its share of b_s), summed per camera by the Schur kernel
  double* S = nullptr;        // n x (n+1) column-major reduced system [S | b_s], n = 6 n_free_cams
  int64_t S_n = 0;
  double* cam_part = nullptr; // n_cams x 4
  double* od_part = nullptr;  // n_odom : chi2 per odometry edge
  int* chol_info = nullptr;
  // reductions
  double* host_part = nullptr;  // pinned: 8 doubles
  double* dev_part = nullptr;
  esl::LmState lm;
  double* cams_snap = nullptr; double* objs_snap = nullptr;
  // profiling (HIP events on this stream)
  bool prof_on = false;
  int prof_level = 0;   // 1: only kernel class 0 (linearise) is bracketed; 2: every class
  unsigned prof_runs = 0; // level 1: device-driven runs since esl_profile_enable (every 4th one carries the sampled launch)
  bool prof_gate = true; // level 1 inside a device-driven run: bracket ONE linearisation per run (event records break
                         // the back-to-back dispatch: bracketing all of them cost 13 % of the C4 run)
  std::vector<hipEvent_t> prof_ev;   // pairs
  std::vector<int> prof_kind;
  size_t prof_used = 0;
  int64_t prof_count[ESL_PROF_KINDS] = {0};
  double prof_ms[ESL_PROF_KINDS] = {0};
  size_t cap_cams = 0, cap_objs = 0;
  // host copies kept from upload (edge -> camera after sorting by ellipsoid)
  std::vector<int> h_bb_cam, h_bb_obj, h_e3_cam, h_e3_obj, h_cam_slot, h_od_i, h_od_j;
  // chunked mapping-mode pipeline (esl_kernels_chunk.hpp)
  int n_chunks = 0;
  int *ck_obj = nullptr, *ck_type = nullptr, *ck_begin = nullptr, *ck_end = nullptr, *ck_ostart = nullptr;
  int *ck_ids_bb = nullptr, *ck_ids_e3 = nullptr; int n_ids_bb = 0, n_ids_e3 = 0;
  double* chunk_out = nullptr;   // n_chunks x 56
  double* chunk_out2 = nullptr;  // second buffer: speculative linearisation of the trial state
  hipEvent_t ev_try = nullptr;
  double* chunk_chi = nullptr;   // n_chunks
  double* blk_part = nullptr;    // per-workgroup partials of k_chunk_finalize
  double* solve_part = nullptr;  // per-workgroup partials of k_obj_solve / k_lm_step (4 each)
  double* blk_chi = nullptr;     // per-workgroup chi2 of the last linearisation
  unsigned int* tickets = nullptr;  // 2 arrival counters
  double* dev_scal = nullptr;    // {chi2_lin, max_diag}
  void* append_img = nullptr;    // host image of the appendable layout (esl_graph_append, esl_capi.hip)
  char* append_dev = nullptr; size_t append_dev_cap = 0;   // device scratch of an append's staged blob
  char* slam_tab_dev = nullptr; size_t slam_tab_cap = 0;   // SLAM-mode append: the camera-side tables + odometry arrays, rebuilt per append (esl_capi.hip)
  // grow-only arenas behind esl_graph_upload (esl_capi.hip)
  char* arena_graph = nullptr; size_t arena_graph_cap = 0;
  char* arena_work = nullptr;  size_t arena_work_cap = 0;
  char* stage_host = nullptr;  size_t stage_host_cap = 0;   // pinned staging blob
  // grow-only blobs of SLAM mode (esl_slam.hip BlobStage): what slam_alloc builds at upload / what the first trial's solver needs
  // arena_solve: the camera-first form's tables, slabs and T; arena_S: the reduced camera system S + its solver workspace.  Separate
  // blobs since round 5: a graph normally runs ONE form, but esl_lm_reduced_system / esl_lm_reduced_residual build S
  // beside a camera-first run, and with one shared blob that re-laid out (and memset) the camera-first set under the caller's feet.
  BlobArena arena_slam, arena_solve, arena_S;
  const int *h_cu_start = nullptr, *h_cu_obj = nullptr, *h_cu_id = nullptr;   // the per-camera lists in arena_slam's staging blob
  std::vector<int> h_ue_start, h_ue_id, h_ue_slot;   // host copy of the unified per-ellipsoid edge lists (the camera-first tables are built from it on first use)
  char* fit_slab = nullptr; size_t fit_slab_cap = 0;        // esl_fit_frame's device slab (esl_fit.hip)
  char* fit_in = nullptr;  size_t fit_in_cap = 0;           // pinned staging of its inputs / outputs
  char* fit_out = nullptr; size_t fit_out_cap = 0;
  char* plane_slab = nullptr; size_t plane_slab_cap = 0;    // esl_extract_ground_plane / esl_extract_planes (esl_plane.hip)
  void* fit_graph_cache = nullptr;                          // captured launch sequences (std::vector<FitGraphEntry>)
  bool parts_fresh = false;      // host_part holds the last trial's scalars already (slam_try_step fetched them with the solver's flag)
  bool cams_match_snap = false;  // cameras untouched since the snapshot: esl_states_restore skips their copy
  int n_grav_edges = 0;
  void* lm_dev = nullptr;        // device-resident LM state (LmCore[2], esl_kernels_chunk.hpp)
  void* lm_host = nullptr;       // mapped pinned LmHostView the host polls
  void* lm_host_dev = nullptr;   // its device alias
  void* host_scal = nullptr;     // mapped pinned LmScalars
  void* host_scal_dev = nullptr; // its device alias
  bool sys_combined = false;
  // RCCL exchange (esl_comm.hip)
  void* comm = nullptr;          // ncclComm_t
  int comm_ranks = 1, comm_rank = 0;
  bool comm_replicated = false;  // esl_comm_set_replicated: every rank holds the whole graph; only the dense solve is divided
  esl_host_allreduce_fn host_allreduce = nullptr;   // host-staged transport (esl_comm_init_host)
  void* host_user = nullptr;
  double* host_stage = nullptr;
  double* dev_gather = nullptr;  // 8 x n_ranks
  double* host_gather = nullptr; // pinned
  double* Linv_ws = nullptr;  // ceil(n/NB) x NB x NB
  double* z_ws = nullptr;
  int64_t S_lda = 0;
  // camera-first elimination (esl_cf.hpp; SLAM mode, cameras chained by the odometry edges): per-ellipsoid edge lists sorted by
  // camera slot, the odometry edges of every consecutive slot pair, the block-bidiagonal factor (Linv, M, N), per-edge V = Linv W,
  // X^T = (L^-1 [W | b_c])^T, the reduced ellipsoid system T (+ b row) and its solver workspaces.  Big buffers on first use.
  bool cf_chain_ok = false;
  bool cf_ready = false;         // every buffer of cf_ensure exists and is initialised (esl_slam.hip)
  bool cf_unavailable = false;   // ESL_SOLVER_AUTO: the camera-first buffers did not fit; this graph runs the reduced camera system
  int *cf_oe_start = nullptr, *cf_oe_u = nullptr, *cf_oe_slot = nullptr, *cf_od_start = nullptr, *cf_od_edge = nullptr, *cf_oe_cst = nullptr;
  int cf_n_list = 0, cf_n_chunks = 0;
  double *cf_Linv = nullptr, *cf_M = nullptr, *cf_N = nullptr, *cf_V = nullptr, *cf_vy = nullptr, *cf_z = nullptr;
  double *cf_B = nullptr, *cf_Lfac = nullptr, *cf_G = nullptr;
  // nested dissection of the camera chain (esl_cf.hpp): every cf_stride-th slot is a separator (0 = plain chain)
  int cf_stride = 0, cf_n_sep = 0, cf_n_seg = 1;
  double *cf_Zt = nullptr, *cf_Hs = nullptr, *cf_Bs = nullptr, *cf_LfacS = nullptr, *cf_GS = nullptr, *cf_LiS = nullptr, *cf_MS = nullptr,
         *cf_NS = nullptr, *cf_R = nullptr;
  double *cf_Xt = nullptr, *cf_T = nullptr, *cf_Linv_ws = nullptr, *cf_part = nullptr;
  int64_t cf_ldx = 0, cf_kpad = 0, cf_ldt = 0;
  // sparse interior rows of X (esl_cf.hpp, "X is sparse"): segments of kCfFwdCh slots; per segment the sorted list of ellipsoids
  // with an edge at one of its interior cameras (seg_start / seg_obj), the column map cmap[p (N + 1) + o] = first << 24 | index
  // (-1: column o of X is zero over segment p; entry N = the right-hand side), per ellipsoid the bitmap of its segments, the
  // compact slabs Xc (xoff / xld) and the separators' dense rows Xs
  bool cf_sp_built = false, cf_sparse = false;
  int cf_sp_nseg = 0, cf_sp_nw = 0, cf_n_fwork = 0;
  double cf_sp_flops = 0;   // FMAs of the per-segment products: block lower triangles, rows from each block row's first camera on
  int *cf_seg_start = nullptr, *cf_seg_obj = nullptr, *cf_seg_first = nullptr, *cf_cmap = nullptr, *cf_xld = nullptr, *cf_fwork = nullptr;
  unsigned long long* cf_mask = nullptr;
  long long* cf_xoff = nullptr;
  size_t cf_xc_len = 0;
  double *cf_Xc = nullptr, *cf_Xs = nullptr;
  int64_t cf_kpad_s = 0;
  // per-segment products P_p = Xc_p^T Xc_p in 9 x 9 blocks (block (i1, i2), i1 >= i2, of segment p at boff[p] + i1 (i1 + 1) / 2 + i2),
  // the right-hand side's row of every product (roff), the tile list of the product kernel; cf_sp_form: 1 = products stored and
  // gathered (k_cf_seg_syrk + k_cf_T_gather), 2 = blocks of T computed from the slabs directly (k_cf_T_sparse: no P, more traffic)
  int cf_sp_form = 0, cf_n_twork = 0;
  long long *cf_boff = nullptr, *cf_roff = nullptr;
  int* cf_twork = nullptr;
  size_t cf_p_blocks = 0, cf_prhs_len = 0;
  double *cf_P = nullptr, *cf_Prhs = nullptr;
  // round 5: T's ellipsoids in the order of their first free camera (sparse form): column block cf_rank[o] of T / R / Xs belongs to
  // ellipsoid o; cf_kfirst[g] = first row of Xs that can be non-zero in columns [64 g, 64 g + 64) (the rank-K update skips the rest);
  // cf_xo_t = x_o as the solver returns it (T's order); cf_upd_flops = what the dense update executes with the staircase
  bool cf_tperm = false;
  int *cf_rank = nullptr, *cf_unrank = nullptr, *cf_kfirst = nullptr;
  double* cf_xo_t = nullptr;
  double cf_upd_flops = 0;
  int lm_solver_used = 0;   // esl_linear_solver the last trial step ran with (1 reduced camera system, 2 reduced ellipsoid system)
  // per-context runtime of the dense solver (esl_chol.hpp CholRuntime: look-ahead stream + events on THIS device, one-time
  // kernel attributes of this device) and of the Schur kernel; created on first use, released with the context
  void* chol_rt = nullptr;
  void* chol_prof_scope = nullptr;   // the open ProfScope (class 9) around a dense factorisation (esl_slam.hip chol_prof_mark)
  unsigned long long graph_fp = 0;   // order-independent fingerprint of the uploaded graph's structure (esl_graph_upload / _append; comm_check_replicated)
  bool repl_checked = false;         // replicated-graph communicator: this graph's fingerprint has been compared across the ranks
  double *chol_pack = nullptr, *chol_pack2 = nullptr;   // staging buffers of the distributed factorisation's panel messages (esl_chol.hpp CholDist)
  size_t chol_pack_len = 0;
  bool schur_attr_set = false;
  // switches resolved ONCE when the communicator is created (every rank must take the same collective sequence for the whole
  // run; an environment read per trial could change mid-run): ESL_CHOL_DIST = 1 / 0 forces the distributed factorisation
  // on / off, unset = by size
  int sw_chol_dist = -1;
  bool sw_chol_overlap = true;   // the distributed factorisation's messages on their own stream under the trailing updates
};
