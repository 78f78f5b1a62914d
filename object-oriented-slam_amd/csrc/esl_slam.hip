// esl_slam.hip — host orchestration of SLAM mode (free cameras): linearise, Schur complement onto the
// cameras, dense FP64-MFMA Cholesky of the reduced camera system, back-substitution, trial states.
// See esl_kernels_slam.hpp for the maths and the reference citations.
#include "esl_slam.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "esl_cf.hpp"
#include "esl_chol.hpp"
#include "esl_kernels_slam.hpp"

namespace esl {

// trial ellipsoids AND trial cameras of the camera-first form in one launch (round 6): blocks [0, nb_obj) are k_cf_obj_update's, the
// rest k_slam_cam_update's
static __global__ __launch_bounds__(256) void k_cf_updates(DevGraph g, int nb_obj, double lambda, const double* __restrict__ objs, const double* __restrict__ bo,
                                                           const double* __restrict__ xo, double* __restrict__ objs_trial, double* __restrict__ obj_part,
                                                           const double* __restrict__ cams, const double* __restrict__ xc, const double* __restrict__ bc,
                                                           double* __restrict__ cams_trial, double* __restrict__ cam_part) {
  if ((int)blockIdx.x < nb_obj) cf_obj_update_one(g, lambda, objs, bo, xo, objs_trial, obj_part, blockIdx.x * 256 + threadIdx.x);
  else slam_cam_update_one(g, lambda, cams, xc, bc, cams_trial, cam_part, ((int)blockIdx.x - nb_obj) * 256 + threadIdx.x);
}

template <class T>
static int al(T** dst, size_t n) {
  if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
  ESL_HIP_TRY(hipMalloc((void**)dst, std::max<size_t>(n, 1) * sizeof(T)));
  return ESL_OK;
}

// ---- grow-only blobs of SLAM mode (round 4) -------------------------------------------------------------------------------
// Rounds 2-3 gave every table and work buffer of SLAM mode its own hipFree + hipMalloc + pageable copy at every upload (~40 of
// them) and freed the solver's multi-GB buffers with the graph: 21 ms per esl_graph_upload at BASELINE configs[3], paid by every
// esl_optimize call of the drop-in adapter.  Now BlobArenas (esl_ctx.hpp): arena_slam holds what slam_alloc makes at upload,
// arena_solve what the first camera-first trial step needs (tables + slabs + T), arena_S the reduced camera system (round 5: a blob
// of its own, so that esl_lm_reduced_system beside a camera-first run no longer re-lays out that form).  A BlobStage plans a layout
// (256-byte aligned items, uploads first), grows the arena and its pinned staging blob when needed, lets the caller build the
// uploaded tables IN the staging blob, and ships them with one copy.  Pointers handed out are interior pointers: never hipFree()d,
// only forgotten (slam_forget) -- the arenas live until the context is destroyed (slam_release).
namespace {
struct BlobStage {
  struct Item { void** dst; size_t bytes, off; bool upload; };
  std::vector<Item> items;
  size_t total = 0, up_end = 0;
  BlobArena* a = nullptr;
  template <class T> int up(T** dst, size_t n) { return add((void**)dst, n * sizeof(T), true); }      // -> item index
  template <class T> int work(T** dst, size_t n) { return add((void**)dst, n * sizeof(T), false); }
  int add(void** dst, size_t bytes, bool upload) {
    const size_t off = (total + 255) / 256 * 256;
    items.push_back({dst, bytes, off, upload});
    total = off + std::max<size_t>(bytes, 8);
    if (upload) up_end = total;
    return (int)items.size() - 1;
  }
  // grows the arena / the staging blob; afterwards host<T>(i) is where item i's table is to be built
  int reserve(BlobArena& arena) {
    a = &arena;
    if (a->ev) ESL_HIP_TRY(hipEventSynchronize(a->ev));   // the previous blob's copy still reads the staging memory
    else ESL_HIP_TRY(hipEventCreateWithFlags(&a->ev, hipEventDisableTiming));
    if (up_end > a->host_cap) {
      if (a->host) { (void)hipHostFree(a->host); a->host = nullptr; a->host_cap = 0; }
      const size_t want = up_end + up_end / 2 + 4096;
      ESL_HIP_TRY(hipHostMalloc((void**)&a->host, want, hipHostMallocDefault));
      a->host_cap = want;
    }
    if (total > a->cap) {
      if (a->dev) { (void)hipFree(a->dev); a->dev = nullptr; a->cap = 0; }
      // growth slack only where it is cheap: 12.5 % of the 28.8 GB reduced camera system of BASELINE configs[3] is 3.6 GB that can turn
      // a fit into an out-of-memory; blobs of that size are sized by ONE graph shape, not by a growing sequence of them
      const size_t want = total + (total < ((size_t)1 << 30) ? total / 8 : 0) + 4096;
      ESL_HIP_TRY(hipMalloc((void**)&a->dev, want));
      a->cap = want;
    }
    for (const Item& it : items) *it.dst = a->dev + it.off;
    return ESL_OK;
  }
  template <class T> T* host(int i) const { return (T*)(a->host + items[(size_t)i].off); }
  int ship(hipStream_t st) {
    if (up_end) ESL_HIP_TRY(hipMemcpyAsync(a->dev, a->host, up_end, hipMemcpyHostToDevice, st));
    ESL_HIP_TRY(hipEventRecord(a->ev, st));
    return ESL_OK;
  }
};
double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

static void cf_forget(esl_ctx* c);
// the graph goes away: every interior pointer is forgotten, the arenas stay (esl_capi.hip free_graph)
void slam_forget(esl_ctx* c) {
  DevGraph& g = c->g;
  g.ue_start = g.ue_id = g.ue_slot = g.cu_start = g.cu_obj = g.cu_id = nullptr;
  c->Hcc = c->bc = c->xc = c->Wbb = c->Abb = c->Aod = c->Dinv = c->Yb = c->Wt = c->Tb = c->cam_part = c->od_part = c->z_ws = nullptr;
  c->cf_od_start = c->cf_od_edge = nullptr;
  c->S = c->Linv_ws = nullptr;
  cf_forget(c);
  c->cf_chain_ok = false; c->cf_unavailable = false;
  c->h_ue_start.clear(); c->h_ue_id.clear(); c->h_ue_slot.clear();
  c->h_cu_start = c->h_cu_obj = c->h_cu_id = nullptr;
}
static void blob_release(BlobArena& a) {
  if (a.dev) (void)hipFree(a.dev);
  if (a.host) (void)hipHostFree(a.host);
  if (a.ev) (void)hipEventDestroy(a.ev);
  a = BlobArena{};
}
void slam_release(esl_ctx* c) { slam_forget(c); blob_release(c->arena_slam); blob_release(c->arena_solve); blob_release(c->arena_S); }
// esl_ctx_trim: drop the solver blobs (tens of GB at BASELINE configs[3]); the next trial step that needs one builds it again.
// keep_lists: the resident graph has free cameras, so arena_slam (its lists, W, Y ...) is still in use.
void slam_trim(esl_ctx* c, bool keep_lists) {
  cf_forget(c);
  c->S = c->Linv_ws = nullptr;
  c->cf_unavailable = false;
  blob_release(c->arena_solve); blob_release(c->arena_S);
  if (!keep_lists) { slam_forget(c); blob_release(c->arena_slam); }
}

// SLAM-mode lists of a freshly uploaded graph + its per-graph work buffers.  All lists come out of counting sorts (no comparison
// sort: the per-ellipsoid order "ascending u" falls out of walking the edge arrays in order, "ascending (slot, u)" out of walking
// the per-camera lists in slot order).
int slam_alloc(esl_ctx* c) {
  DevGraph& g = c->g;
  const int N = g.n_objs, F = g.n_cams, nf = g.n_free_cams;
  const bool timing = std::getenv("ESL_UPLOAD_HOST_TIMING") != nullptr;
  const double t0 = timing ? now_us() : 0;
  // unified per-ellipsoid list of edges whose camera is free (u = bounding-box edge i, or n_bbox + 3-D edge i), ascending u
  std::vector<int>& start = c->h_ue_start;
  std::vector<int>& id = c->h_ue_id;
  std::vector<int>& slot = c->h_ue_slot;
  start.assign((size_t)N + 1, 0);
  // (appendable layout, esl_graph_append in SLAM mode: the arrays have slack behind every ellipsoid's slice -- free slots carry obj = -1)
  for (int i = 0; i < g.n_bbox; ++i) if (c->h_bb_obj[i] >= 0 && c->h_cam_slot[c->h_bb_cam[i]] >= 0) ++start[(size_t)c->h_bb_obj[i] + 1];
  for (int i = 0; i < g.n_e3d; ++i) if (c->h_e3_obj[i] >= 0 && c->h_cam_slot[c->h_e3_cam[i]] >= 0) ++start[(size_t)c->h_e3_obj[i] + 1];
  for (int o = 0; o < N; ++o) start[(size_t)o + 1] += start[o];
  const size_t nue = (size_t)start[N];
  id.resize(nue); slot.resize(nue);
  {
    std::vector<int> cur(start.begin(), start.end() - 1);
    for (int i = 0; i < g.n_bbox; ++i) {
      if (c->h_bb_obj[i] < 0) continue;
      const int s = c->h_cam_slot[c->h_bb_cam[i]];
      if (s >= 0) { const int at = cur[c->h_bb_obj[i]]++; id[at] = i; slot[at] = s; }
    }
    for (int i = 0; i < g.n_e3d; ++i) {
      if (c->h_e3_obj[i] < 0) continue;
      const int s = c->h_cam_slot[c->h_e3_cam[i]];
      if (s >= 0) { const int at = cur[c->h_e3_obj[i]]++; id[at] = g.n_bbox + i; slot[at] = s; }
    }
  }
  g.n_ue = (int)nue;
  // odometry edges of every pair of neighbouring slots (the camera chain of esl_cf.hpp): the camera-first form applies when ALL
  // odometry edges between free cameras join neighbours.  (Its other tables are built by the first trial step that picks it.)
  std::vector<int> ods((size_t)std::max(nf, 1) + 1, 0), ode;
  bool chain = true;
  {
    std::vector<std::pair<int, int>> pe;   // (lower slot, 2 e + orientation)
    for (int e = 0; e < g.n_odom; ++e) {
      const int si = c->h_cam_slot[c->h_od_i[e]], sj = c->h_cam_slot[c->h_od_j[e]];
      if (si < 0 || sj < 0) continue;                       // a fixed end: the edge only adds to the free end's diagonal block
      if (si == sj + 1) pe.push_back({sj, e * 2});          // rows = first vertex = the higher slot: B as stored
      else if (sj == si + 1) pe.push_back({si, e * 2 + 1});
      else chain = false;                                   // (si == sj: a self edge; |si - sj| > 1: fill-in outside the band)
    }
    for (auto& p : pe) ++ods[(size_t)p.first + 1];
    for (int k = 0; k < nf; ++k) ods[(size_t)k + 1] += ods[k];
    ode.resize(pe.size());
    std::vector<int> cur(ods.begin(), ods.end() - 1);
    for (auto& p : pe) ode[(size_t)cur[p.first]++] = p.second;   // (edges of one pair stay in ascending e)
  }
  c->cf_chain_ok = chain && nf > 0 && N > 0;
  const double t1 = timing ? now_us() : 0;
  // layout: uploads first, then the work buffers
  const size_t EU = (size_t)g.n_bbox + g.n_e3d;
  const int64_t n = 6 * (int64_t)nf;
  c->S_n = n;
  c->S_lda = ((n + 1 + 127) / 128) * 128;   // (whole 128-row tiles for the update kernels: chol_update_tile_v reads pad rows instead of guarding the last tile row)
  BlobStage st;
  const int i_ust = st.up(&g.ue_start, start.size()), i_uid = st.up(&g.ue_id, nue), i_usl = st.up(&g.ue_slot, nue);
  const int i_cst = st.up(&g.cu_start, (size_t)nf + 1), i_cob = st.up(&g.cu_obj, nue), i_cid = st.up(&g.cu_id, nue);
  const int i_ods = st.up(&c->cf_od_start, ods.size()), i_ode = st.up(&c->cf_od_edge, ode.size());
  st.work(&c->Hcc, (size_t)nf * 36); st.work(&c->bc, (size_t)nf * 6);
  st.work(&c->Wbb, EU * 54);    // unified W  [54][EU]
  st.work(&c->Abb, EU * 27);    // unified A  [EU][27] (per-edge records)
  st.work(&c->Yb, EU * 54);     // unified Y  [EU][54]
  st.work(&c->Wt, EU * 54);     // W once more as per-edge records [EU][9][6]: what k_slam_schur_pull gathers
  st.work(&c->Tb, EU * 6);      // Y_e b_o    [6][EU]
  st.work(&c->Aod, (size_t)g.n_odom * 90); st.work(&c->Dinv, (size_t)N * 81);
  st.work(&c->cam_part, (size_t)F * 4); st.work(&c->od_part, (size_t)g.n_odom);
  st.work(&c->xc, (size_t)n); st.work(&c->z_ws, (size_t)kNB);
  if (int rc = st.reserve(c->arena_slam)) return rc;
  std::memcpy(st.host<int>(i_ust), start.data(), start.size() * sizeof(int));
  if (nue) { std::memcpy(st.host<int>(i_uid), id.data(), nue * sizeof(int)); std::memcpy(st.host<int>(i_usl), slot.data(), nue * sizeof(int)); }
  std::memcpy(st.host<int>(i_ods), ods.data(), ods.size() * sizeof(int));
  if (!ode.empty()) std::memcpy(st.host<int>(i_ode), ode.data(), ode.size() * sizeof(int));
  // the same edges per free camera, sorted by (ellipsoid, u): two of these lists are intersected per block of S.  Walking the
  // unified list in order (ellipsoids ascending, u ascending inside one) keeps every camera's list sorted.
  {
    int* cstart = st.host<int>(i_cst); int* cobj = st.host<int>(i_cob); int* cid = st.host<int>(i_cid);
    c->h_cu_start = cstart; c->h_cu_obj = cobj; c->h_cu_id = cid;   // (staging memory: valid until the next upload; cf_ensure_impl reads them)
    std::fill(cstart, cstart + nf + 1, 0);
    for (size_t k = 0; k < nue; ++k) ++cstart[(size_t)slot[k] + 1];
    g.cu_max = 0;
    for (int s = 0; s < nf; ++s) { g.cu_max = std::max(g.cu_max, cstart[(size_t)s + 1]); cstart[(size_t)s + 1] += cstart[s]; }
    std::vector<int> cur(cstart, cstart + nf);
    for (int o = 0; o < N; ++o)
      for (int k = start[o]; k < start[(size_t)o + 1]; ++k) { const int at = cur[slot[k]]++; cobj[at] = o; cid[at] = id[k]; }
  }
  if (int rc = st.ship(c->stream)) return rc;
  // W / A of edges that are never written (fixed camera) must not hold NaN garbage where they are summed
  ESL_HIP_TRY(hipMemsetAsync(c->Abb, 0, std::max<size_t>(EU, 1) * 27 * sizeof(double), c->stream));
  ESL_HIP_TRY(hipMemsetAsync(c->Wbb, 0, std::max<size_t>(EU, 1) * 54 * sizeof(double), c->stream));
  if (timing) {
    const double t2 = now_us();
    ESL_HIP_TRY(hipStreamSynchronize(c->stream));
    fprintf(stderr, "[slam_alloc host, us] lists %.0f  layout + per-camera lists + enqueue %.0f  wait %.0f  (blob %.1f MB uploaded, %.1f MB in all)\n",
            t1 - t0, t2 - t1, now_us() - t2, st.up_end / 1e6, st.total / 1e6);
  }
  return ESL_OK;
}

static CholRuntime& chol_rt(esl_ctx* c);
static __global__ void k_info_to_double(const int* __restrict__ info, double* __restrict__ d) { d[0] = info[0] ? 1.0 : 0.0; }
static __global__ void k_double_to_info(const double* __restrict__ d, int* __restrict__ info) { if (d[0] > 0.5) info[0] |= 1; }
// the reduced camera system S (28.8 GB at C4) lives in a grow-only blob of its own, kept from graph to graph (esl_ctx_trim drops
// it); the pointer esl_lm_reduced_system hands out stays valid until the next graph upload, trim or context destroy -- its
// CONTENT only until the next trial step that factors S in place
static int slam_ensure_S(esl_ctx* c) {
  if (c->S) return ESL_OK;
  const size_t n = (size_t)c->S_n, np = (n + kNB - 1) / kNB;
  BlobStage st;
  st.work(&c->S, (size_t)c->S_lda * n); st.work(&c->Linv_ws, np * kNB * kNB);
  const int rc = st.reserve(c->arena_S);
  if (rc) { c->S = c->Linv_ws = nullptr; return rc; }
  return st.ship(c->stream);
}

// ---- camera-first elimination: host side (kernels and the maths: esl_cf.hpp) ---------------------------------------------
// staging buffer of the distributed factorisation (one message per outer panel: esl_chol.hpp CholDist::pack)
static int chol_pack_ensure(esl_ctx* c, int n) {
  const size_t need = chol_pack_len(n);
  if (c->chol_pack && c->chol_pack2 && c->chol_pack_len >= need) return ESL_OK;
  int rc = al(&c->chol_pack, need);
  if (rc) return rc;
  if ((rc = al(&c->chol_pack2, need))) return rc;
  c->chol_pack_len = need;
  return ESL_OK;
}
// CholDist of this context: one packed message per outer panel, on its own stream unless switched off (ESL_CHOL_DIST_OVERLAP=0)
static int chol_dist_fill(esl_ctx* c, int n, CholDist& d) {
  d.rank = c->comm_rank; d.n_ranks = c->comm_ranks; d.user = c;
  d.bcast = [](void* u, double* dev, size_t count, int root) { return comm_bcast((esl_ctx*)u, dev, count, root); };
  const int rc = chol_pack_ensure(c, n);
  if (rc) return rc;
  d.pack = c->chol_pack;
  if (c->sw_chol_overlap) {
    d.pack2 = c->chol_pack2;
    d.bcast_on = [](void* u, double* dev, size_t count, int root, hipStream_t st) { return comm_bcast((esl_ctx*)u, dev, count, root, st); };
  }
  return ESL_OK;
}
static bool cf_applicable(const esl_ctx* c) { return c->cf_chain_ok && (!c->comm || c->comm_replicated); }
// replicated-graph communicator (esl_comm_set_replicated): the reduced ellipsoid system's outer panels are dealt to the ranks
static bool cf_dist(const esl_ctx* c) {
  if (!c->comm || !c->comm_replicated || c->comm_ranks < 2) return false;
  if (c->sw_chol_dist >= 0) return c->sw_chol_dist == 1;
  return 9L * c->g.n_objs >= 8192;
}
// esl_lm_params::linear_solver -> the form this trial runs with; < 0: the request cannot be served
static int slam_pick_solver(const esl_ctx* c) {
  const int want = c->lm.p.linear_solver;
  if (want == ESL_SOLVER_REDUCED_CAMERA) return ESL_SOLVER_REDUCED_CAMERA;
  if (want == ESL_SOLVER_REDUCED_ELLIPSOID) return cf_applicable(c) ? ESL_SOLVER_REDUCED_ELLIPSOID : -1;
  if (!cf_applicable(c) || c->cf_unavailable) return ESL_SOLVER_REDUCED_CAMERA;
  const double nc = (double)c->S_n, no = 9.0 * c->g.n_objs;
  return (no * no * nc + no * no * no / 3.0 < nc * nc * nc / 3.0) ? ESL_SOLVER_REDUCED_ELLIPSOID : ESL_SOLVER_REDUCED_CAMERA;
}
template <class T>
static void fr(T** p) { if (*p) { (void)hipFree(*p); *p = nullptr; } }
// everything cf_ensure hands out (interior pointers of the solver arena: forgotten, not freed)
static void cf_forget(esl_ctx* c) {
  c->cf_oe_start = c->cf_oe_u = c->cf_oe_slot = c->cf_oe_cst = nullptr;
  c->cf_seg_start = c->cf_seg_obj = c->cf_seg_first = c->cf_cmap = c->cf_xld = c->cf_fwork = c->cf_twork = nullptr;
  c->cf_mask = nullptr; c->cf_xoff = c->cf_boff = c->cf_roff = nullptr;
  c->cf_Linv = c->cf_M = c->cf_N = c->cf_V = c->cf_B = c->cf_Lfac = c->cf_G = c->cf_vy = c->cf_z = nullptr;
  c->cf_T = c->cf_Linv_ws = c->cf_part = nullptr;
  c->cf_Zt = c->cf_Hs = c->cf_Bs = c->cf_LfacS = c->cf_GS = c->cf_LiS = c->cf_MS = c->cf_NS = c->cf_R = nullptr;
  c->cf_Xc = c->cf_Xs = c->cf_P = c->cf_Prhs = c->cf_Xt = nullptr;
  c->cf_rank = c->cf_unrank = c->cf_kfirst = nullptr; c->cf_xo_t = nullptr; c->cf_tperm = false; c->cf_upd_flops = 0;
  c->cf_ready = false; c->cf_sp_built = c->cf_sparse = false;
}
static int cf_ensure_impl(esl_ctx* c);
// The tables and buffers of the camera-first form exist and are initialised <=> cf_ready.  A failed allocation (dense X alone is
// 8.6 GB at BASELINE configs[3]) forgets the set again: the next trial step starts from scratch instead of launching on null
// pointers.
static int cf_ensure(esl_ctx* c) {
  if (c->cf_ready) return ESL_OK;
  const int rc = cf_ensure_impl(c);
  if (rc) { cf_forget(c); (void)hipGetLastError(); return rc; }
  c->cf_ready = true;
  return ESL_OK;
}
// Built by the FIRST trial step that picks the camera-first form -- not at upload (a graph whose trials run the reduced
// camera system, or that is sharded over ranks, never pays for them):
//   - every ellipsoid's free-camera edges sorted by (slot, u), and the entry range of every (ellipsoid, chunk of kCfFwdCh slots) in
//     those lists: the forward substitution's gather phase;
//   - when X = L^-1 W is kept SPARSE over runs of cameras (esl_cf.hpp): per segment of kCfFwdCh slots (the last one the separator)
//     the ellipsoids seen by its interior cameras, where each one's column starts, per ellipsoid the bitmap of its segments, the
//     work lists of the per-segment kernels;
//   - the factor of the camera block, V, X (slabs or dense), T and its solver workspace.
// No comparison sort on the host, and the dense index tables (entry ranges, column map, tile lists) are generated on the device.
static int cf_ensure_impl(esl_ctx* c) {
  const DevGraph& g = c->g;
  const int N = g.n_objs, nfi = g.n_free_cams;
  const size_t nf = (size_t)nfi, EU = (size_t)g.n_bbox + g.n_e3d, n_o = (size_t)9 * N;
  const bool timing = std::getenv("ESL_UPLOAD_HOST_TIMING") != nullptr;
  const double t0 = timing ? now_us() : 0;
  c->cf_ldx = (int64_t)((n_o + 1 + 127) / 128 * 128);   // (whole 128-row tiles: the update tiles read the pad rows instead of guarding their last tile row, chol_update_tile_v)
  c->cf_kpad = (int64_t)((6 * nf + kKC - 1) / kKC * kKC);
  c->cf_ldt = c->cf_ldx;
  const std::vector<int>& start = c->h_ue_start;
  const size_t nue = c->h_ue_id.size();
  const int nch = (nfi + kCfFwdCh - 1) / kCfFwdCh;
  c->cf_n_list = (int)nue; c->cf_n_chunks = nch;
  if ((size_t)std::max(N, 1) * (size_t)(nch + 1) > ((size_t)1 << 28)) {
    set_error("camera-first elimination: the (ellipsoid, camera chunk) range table of this graph exceeds 1 GB");
    return ESL_ERR_ALLOC;
  }
  // (slot, u) order per ellipsoid = the per-camera lists of slam_alloc (ellipsoids, then u, ascending inside a slot) walked in slot
  // order and dealt to the ellipsoids.  The same walk meets, for the sparse form, every (segment, ellipsoid) pair first at the
  // ellipsoid's FIRST interior camera of the segment, slot by slot: the incidences come out sorted by (segment, first camera,
  // ellipsoid) -- the compact column order of a segment's slab (upper-trapezoidal: column c is zero above row 6 first(c); a tile of
  // the segment's product starts at the first camera of its first column) -- with no sort at all.
  const bool want_sp = c->cf_chain_ok && nfi >= 256 && N > 0 && (size_t)nch * (size_t)(N + 1) <= ((size_t)1 << 28) && !std::getenv("ESL_CF_NO_ND");
  const int nseg = nch, nw = (nseg + 63) / 64, N1 = N + 1;
  std::vector<int> os(nue), ou(nue);
  std::vector<int> inc_p, inc_o, inc_first;
  std::vector<int> seg_start, xld, fw_off;
  std::vector<long long> xoff, boff, roff, tw_off;
  size_t n_fwork = 0, n_twork = 0;
  double fl = 0;
  {
    std::vector<int> cur(start.begin(), start.end() - 1), last_seg(want_sp ? (size_t)N : 0, -1);
    const int* cstart = c->h_cu_start; const int* cobj = c->h_cu_obj; const int* cid = c->h_cu_id;
    if (want_sp) { inc_p.reserve(nue / 4 + 16); inc_o.reserve(nue / 4 + 16); inc_first.reserve(nue / 4 + 16); }
    for (int sl = 0; sl < nfi; ++sl) {
      const int p = sl / kCfFwdCh, pos = sl - p * kCfFwdCh;
      const bool interior = want_sp && pos != kCfFwdCh - 1;   // (the last slot of a segment is its separator)
      for (int q = cstart[sl]; q < cstart[sl + 1]; ++q) {
        const int o = cobj[q], at = cur[o]++;
        os[at] = sl; ou[at] = cid[q];
        if (interior && last_seg[o] != p) { last_seg[o] = p; inc_p.push_back(p); inc_o.push_back(o); inc_first.push_back(pos); }
      }
    }
  }
  const size_t n_inc = inc_p.size();
  if (want_sp) {
    seg_start.assign((size_t)nseg + 1, 0); xld.assign((size_t)nseg, 0); fw_off.assign((size_t)nseg + 1, 0);
    xoff.assign((size_t)nseg + 1, 0); boff.assign((size_t)nseg + 1, 0); roff.assign((size_t)nseg + 1, 0); tw_off.assign((size_t)nseg + 1, 0);
    for (size_t q = 0; q < n_inc; ++q) ++seg_start[(size_t)inc_p[q] + 1];
    for (int p = 0; p < nseg; ++p) seg_start[(size_t)p + 1] += seg_start[p];
    for (size_t q = 0; q < n_inc; ++q)    // block row k of the product: k + 1 blocks, rows from its first camera on
      fl += 81.0 * (double)(q - (size_t)seg_start[inc_p[q]] + 1) * (6.0 * (kCfFwdCh - 1 - inc_first[q]));
    for (int p = 0; p < nseg; ++p) {
      const int cnt = seg_start[(size_t)p + 1] - seg_start[p];
      const int m = 9 * cnt + 1;
      xld[p] = (m + 15) / 16 * 16 + 16;                          // + 16: the T kernel reads 16 columns from any column start
      xoff[(size_t)p + 1] = xoff[p] + (long long)kCfSegRows * xld[p];
      fw_off[(size_t)p + 1] = fw_off[p] + (m + 63) / 64;
      boff[(size_t)p + 1] = boff[p] + (long long)cnt * (cnt + 1) / 2;
      roff[(size_t)p + 1] = roff[p] + 9LL * cnt;
      const long long nt = (9 * cnt + kCfSyT - 1) / kCfSyT;
      tw_off[(size_t)p + 1] = tw_off[p] + nt * (nt + 1) / 2;     // tiles of the product on and below the diagonal (ellipsoid columns only)
    }
    n_fwork = (size_t)fw_off[nseg]; n_twork = (size_t)tw_off[nseg];
    c->cf_sp_nseg = nseg; c->cf_sp_nw = nw; c->cf_n_fwork = (int)n_fwork; c->cf_sp_flops = fl;
    c->cf_xc_len = (size_t)xoff[nseg] + 2 * kCfSyT;   // (+: the product kernel's last tile reads past the last slab's columns)
    c->cf_n_twork = (int)n_twork; c->cf_p_blocks = (size_t)boff[nseg]; c->cf_prhs_len = (size_t)roff[nseg];
    c->cf_sp_built = true;
    // X sparse or dense (esl_cf.hpp, "sparse interior rows"): flops of the dense rank-6 nf update against the separators' dense
    // rows plus the per-segment products at a quarter of the MFMA rate; ESL_CF_SPARSE=1 / 0 forces it (A/B and the parity tests)
    const double dense = (double)n_o * (double)n_o * 6.0 * (double)nf;
    const double sparse = (double)n_o * (double)n_o * 6.0 * (double)(nf / kCfFwdCh) + 24.0 * c->cf_sp_flops;   // (cf_sp_flops: FMAs of the products' lower triangles)
    const char* sw = std::getenv("ESL_CF_SPARSE");
    c->cf_sparse = sw ? sw[0] != '0' : (n_o >= 2048 && sparse < 0.5 * dense);
    // the per-segment products are stored when they fit beside everything else (C4: 14 GB); else (or ESL_CF_SPARSE=2) the blocks
    // of T come straight from the slabs (k_cf_T_sparse).  Decided from the graph's size against the device's TOTAL memory, not from
    // what happens to be free at the first trial: the two forms add in different orders, so the bits of T must not depend on what
    // else occupies the GPU; every rank of a replicated run takes the same decision by construction.
    // esl_lm_solver_stats reports the form.
    size_t free_b = 0, total_b = 0;
    ESL_HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const double n_sep = (double)(nf / kCfFwdCh);
    const double need = 8.0 * (81.0 * (double)c->cf_p_blocks + (double)c->cf_xc_len + 2.0 * (double)c->cf_ldt * (double)n_o +
                               2.0 * 6.0 * n_sep * (double)c->cf_ldx /* Xs, R */ + 54.0 * (double)EU /* V */);
    c->cf_sp_form = (sw && sw[0] == '2') || need > 0.4 * (double)total_b ? 2 : 1;
    if (!c->cf_sparse) c->cf_sp_built = false;   // (none of the sparse-form tables are shipped)
  }
  const bool sp = c->cf_sparse;
  // nested dissection of the camera chain from 48 free cameras on: stride = 16 x round(sqrt(nf) / 16) in [16, 128] (a multiple
  // of the forward substitution's chunk, so that segments start on chunk boundaries), kCfFwdCh when X is kept sparse;
  // ESL_CF_NO_ND=1 keeps the plain chain (A/B)
  c->cf_stride = 0; c->cf_n_sep = 0; c->cf_n_seg = 1;
  // (ESL_CF_ND_MIN: first free-camera count that is dissected -- default 48, three segments; 128 until round 6: at 119 free cameras the
  //  plain chain's two serial sweeps, k_cf_chain 72 us + k_cf_forward 69 us, were a quarter of a streaming-size trial)
  static const size_t nd_min = [] { const char* e = std::getenv("ESL_CF_ND_MIN"); const int v = e ? std::atoi(e) : 48; return (size_t)std::max(v, 2 * kCfFwdCh); }();
  const bool nd = nf >= nd_min && !std::getenv("ESL_CF_NO_ND");
  if (nd) {
    int sd = 16 * (int)std::max(1.0, std::floor(std::sqrt((double)nf) / 16.0 + 0.5));
    sd = std::min(sd, kCfMaxStride);
    if (sp) sd = kCfFwdCh;
    c->cf_stride = sd; c->cf_n_sep = (int)(nf / sd); c->cf_n_seg = (int)((nf + sd - 1) / sd);
  }
  // T's ellipsoids by first free camera (sparse form; ESL_CF_TPERM=0: ellipsoid order, A/B): rank by a counting sort over the
  // first slot of every ellipsoid's (slot-sorted) list, ellipsoids without a free-camera edge last; kfirst per 64 columns of T: the
  // first separator whose right-hand side R(k) can hold the column's ellipsoid is the one in FRONT of the segment of its first
  // camera (an interior camera's segment hands its sum to the separator before it; conservative for a separator camera itself),
  // and the right-hand side's ROW (row n_o of T, X^T y) is dense from the start
  std::vector<int> h_rank, h_unrank, h_kfirst;
  double upd_flops = 0;
  const bool tperm = sp && N > 0 && !(std::getenv("ESL_CF_TPERM") && std::getenv("ESL_CF_TPERM")[0] == '0');
  if (tperm) {
    std::vector<int> first((size_t)N, nfi), cnt((size_t)nfi + 2, 0);
    for (int o = 0; o < N; ++o) if (start[(size_t)o + 1] > start[o]) first[o] = os[(size_t)start[o]];
    for (int o = 0; o < N; ++o) ++cnt[(size_t)first[o] + 1];
    for (int s2 = 0; s2 <= nfi; ++s2) cnt[(size_t)s2 + 1] += cnt[s2];
    h_rank.assign((size_t)N, 0); h_unrank.assign((size_t)N, 0);
    for (int o = 0; o < N; ++o) { const int t = cnt[first[o]]++; h_rank[o] = t; h_unrank[t] = o; }
    const int Ks = (int)((6 * (size_t)c->cf_n_sep + kKC - 1) / kKC * kKC);
    const size_t ng = (n_o + 1 + 256) / 64 + 8;
    h_kfirst.assign(ng, Ks);
    for (size_t col = 0; col < n_o; ++col) {
      const int o = h_unrank[col / 9];
      const int k0 = std::max(0, first[o] / kCfFwdCh - 1);
      const int row = std::min(6 * k0, Ks);
      h_kfirst[col / 64] = std::min(h_kfirst[col / 64], row);
    }
    h_kfirst[n_o / 64] = 0;   // the right-hand side's row
    // flops the update executes (k_chol_update_v, the default tile of the big launches): per 128 x 128 tile of the lower triangle, K from
    // the later of the tile's two earliest rows on, rounded down to a chunk pair as the kernel does; a tile on the diagonal counts its
    // lower triangle.  (The 256 x 128 tile of ESL_UPD_V=0 starts a little earlier: it executes ~1 % more.)
    const long nti = ((long)n_o + 1 + 127) / 128;
    for (long ti = 0; ti < nti; ++ti) {
      const int ka = std::min(h_kfirst[(size_t)(ti * 2)], h_kfirst[(size_t)(ti * 2 + 1)]);
      for (long tj = 0; tj <= ti && tj * 128 < (long)n_o; ++tj) {
        const int kb = std::min(h_kfirst[(size_t)(tj * 2)], h_kfirst[(size_t)(tj * 2 + 1)]);
        int ks = std::min(std::max(ka, kb), Ks) & ~(kKC - 1);
        if (((Ks - ks) & kKC) != 0 && ks >= kKC) ks -= kKC;
        const double rows_t = (double)std::min<long>(128, (long)n_o + 1 - ti * 128), cols_t = (double)std::min<long>(128, (long)n_o - tj * 128);
        const double in_triangle = tj < ti ? 1.0 : 0.5 * (cols_t + 1.0) / cols_t;
        upd_flops += 2.0 * rows_t * cols_t * (double)(Ks - ks) * in_triangle;
      }
    }
  }
  c->cf_tperm = tperm; c->cf_upd_flops = upd_flops;
  const double t1 = timing ? now_us() : 0;
  // ---- layout: what the host ships first, then the device-generated tables (esl_cf.hpp k_cf_make_*), then the buffers ----
  BlobStage st;
  const int i_st = st.up(&c->cf_oe_start, start.size()), i_ou = st.up(&c->cf_oe_u, nue), i_os = st.up(&c->cf_oe_slot, nue);
  int* d_inc = nullptr; int* d_fw_off = nullptr; long long* d_tw_off = nullptr;
  int i_inc = -1, i_ss = -1, i_xld = -1, i_xoff = -1, i_boff = -1, i_roff = -1, i_fwo = -1, i_two = -1, i_rk = -1, i_ur = -1, i_kf = -1;
  if (tperm) { i_rk = st.up(&c->cf_rank, h_rank.size()); i_ur = st.up(&c->cf_unrank, h_unrank.size()); i_kf = st.up(&c->cf_kfirst, h_kfirst.size()); }
  if (sp) {
    i_inc = st.up(&d_inc, 4 * n_inc); i_ss = st.up(&c->cf_seg_start, (size_t)nseg + 1); i_xld = st.up(&c->cf_xld, (size_t)nseg);
    i_xoff = st.up(&c->cf_xoff, (size_t)nseg + 1); i_boff = st.up(&c->cf_boff, (size_t)nseg + 1); i_roff = st.up(&c->cf_roff, (size_t)nseg + 1);
    i_fwo = st.up(&d_fw_off, (size_t)nseg + 1); i_two = st.up(&d_tw_off, (size_t)nseg + 1);
  }
  st.work(&c->cf_oe_cst, (size_t)std::max(N, 1) * (nch + 1));
  if (sp) {
    st.work(&c->cf_cmap, (size_t)nseg * N1); st.work(&c->cf_mask, (size_t)N1 * nw); st.work(&c->cf_seg_obj, n_inc); st.work(&c->cf_seg_first, n_inc);
    st.work(&c->cf_fwork, 2 * n_fwork); st.work(&c->cf_twork, 3 * n_twork);
  }
  st.work(&c->cf_Linv, nf * 36); st.work(&c->cf_M, nf * 36); st.work(&c->cf_N, nf * 36); st.work(&c->cf_V, EU * 54);
  st.work(&c->cf_B, nf * 36); st.work(&c->cf_Lfac, nf * 21); st.work(&c->cf_G, nf * 36); st.work(&c->cf_vy, nf * 6); st.work(&c->cf_z, nf * 6);
  st.work(&c->cf_T, (size_t)c->cf_ldt * n_o); st.work(&c->cf_Linv_ws, ((n_o + kNB - 1) / kNB) * kNB * kNB);
  if (tperm) st.work(&c->cf_xo_t, n_o);
  if (n_o <= 1024) st.work(&c->cf_part, (size_t)kCholMaxSplit * (size_t)c->cf_ldt * n_o);   // split-K workspace of small systems
  if (nd) {
    const size_t ns = (size_t)std::max(c->cf_n_sep, 1);
    st.work(&c->cf_Zt, nf * 36); st.work(&c->cf_Hs, ns * 36); st.work(&c->cf_Bs, ns * 36); st.work(&c->cf_LfacS, ns * 21);
    st.work(&c->cf_GS, ns * 36); st.work(&c->cf_LiS, ns * 36); st.work(&c->cf_MS, ns * 36); st.work(&c->cf_NS, ns * 36);
    st.work(&c->cf_R, ns * 6 * (size_t)c->cf_ldx);
  }
  if (sp) {
    c->cf_kpad_s = (int64_t)((6 * (size_t)c->cf_n_sep + kKC - 1) / kKC * kKC);
    st.work(&c->cf_Xc, c->cf_xc_len); st.work(&c->cf_Xs, (size_t)c->cf_ldx * (size_t)c->cf_kpad_s);
    if (c->cf_sp_form == 1) { st.work(&c->cf_P, c->cf_p_blocks * 81); st.work(&c->cf_Prhs, c->cf_prhs_len); }
  } else {
    st.work(&c->cf_Xt, (size_t)c->cf_ldx * (size_t)c->cf_kpad);
  }
  if (int rc = st.reserve(c->arena_solve)) return rc;
  // ---- fill the staging blob, ship it, generate the dense tables ----
  std::memcpy(st.host<int>(i_st), start.data(), start.size() * sizeof(int));
  if (nue) { std::memcpy(st.host<int>(i_ou), ou.data(), nue * sizeof(int)); std::memcpy(st.host<int>(i_os), os.data(), nue * sizeof(int)); }
  if (tperm) {
    std::memcpy(st.host<int>(i_rk), h_rank.data(), h_rank.size() * sizeof(int));
    std::memcpy(st.host<int>(i_ur), h_unrank.data(), h_unrank.size() * sizeof(int));
    std::memcpy(st.host<int>(i_kf), h_kfirst.data(), h_kfirst.size() * sizeof(int));
  }
  if (sp) {
    int* inc = st.host<int>(i_inc);
    if (n_inc) {
      std::memcpy(inc, inc_p.data(), n_inc * sizeof(int)); std::memcpy(inc + n_inc, inc_o.data(), n_inc * sizeof(int));
      std::memcpy(inc + 2 * n_inc, inc_first.data(), n_inc * sizeof(int));
      for (size_t q = 0; q < n_inc; ++q) inc[3 * n_inc + q] = (int)q;   // (already in column order)
    }
    std::memcpy(st.host<int>(i_ss), seg_start.data(), seg_start.size() * sizeof(int));
    std::memcpy(st.host<int>(i_xld), xld.data(), xld.size() * sizeof(int));
    std::memcpy(st.host<int>(i_fwo), fw_off.data(), fw_off.size() * sizeof(int));
    std::memcpy(st.host<long long>(i_xoff), xoff.data(), xoff.size() * sizeof(long long));
    std::memcpy(st.host<long long>(i_boff), boff.data(), boff.size() * sizeof(long long));
    std::memcpy(st.host<long long>(i_roff), roff.data(), roff.size() * sizeof(long long));
    std::memcpy(st.host<long long>(i_two), tw_off.data(), tw_off.size() * sizeof(long long));
  }
  if (int rc = st.ship(c->stream)) return rc;
  if (N > 0) {
    const long nt = (long)N * (nch + 1);
    hipLaunchKernelGGL(k_cf_make_cst, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, c->stream, N, nch, c->cf_oe_start, c->cf_oe_slot, c->cf_oe_cst);
  }
  if (sp) {
    ESL_HIP_TRY(hipMemsetAsync(c->cf_cmap, 0xFF, (size_t)nseg * N1 * sizeof(int), c->stream));
    ESL_HIP_TRY(hipMemsetAsync(c->cf_mask, 0, (size_t)N1 * nw * sizeof(unsigned long long), c->stream));
    if (n_inc)
      hipLaunchKernelGGL(k_cf_make_cmap, dim3((unsigned)((n_inc + 255) / 256)), dim3(256), 0, c->stream, (int)n_inc, d_inc, N1, nw, c->cf_seg_start, c->cf_cmap,
                         c->cf_mask, c->cf_seg_obj, c->cf_seg_first);
    hipLaunchKernelGGL(k_cf_make_work, dim3((unsigned)nseg), dim3(64), 0, c->stream, nseg, N, nw, c->cf_seg_start, d_fw_off, d_tw_off, c->cf_cmap, c->cf_mask,
                       c->cf_fwork, c->cf_twork);
  }
  // ---- what must be zero ----
  if (nd) ESL_HIP_TRY(hipMemsetAsync(c->cf_Zt, 0, nf * 36 * sizeof(double), c->stream));
  if (sp) {
    // slabs: pad columns and rows 90 .. 95 stay zero; Xs: rows 6 n_sep .. kpad_s (the K padding of the rank-K update) stay zero;
    // T: the blocks above the diagonal are never written
    ESL_HIP_TRY(hipMemsetAsync(c->cf_Xc, 0, std::max<size_t>(c->cf_xc_len, 1) * sizeof(double), c->stream));
    ESL_HIP_TRY(hipMemsetAsync(c->cf_Xs, 0, (size_t)c->cf_ldx * (size_t)c->cf_kpad_s * sizeof(double), c->stream));
    ESL_HIP_TRY(hipMemsetAsync(c->cf_T, 0, (size_t)c->cf_ldt * n_o * sizeof(double), c->stream));
  } else {
    // rows 6 nf .. kpad of X (the K padding of the rank-K update) and the columns of never-written edges stay zero
    ESL_HIP_TRY(hipMemsetAsync(c->cf_Xt, 0, (size_t)c->cf_ldx * (size_t)c->cf_kpad * sizeof(double), c->stream));
  }
  ESL_HIP_TRY(hipMemsetAsync(c->cf_V, 0, std::max<size_t>(EU, 1) * 54 * sizeof(double), c->stream));
  if (timing)
    fprintf(stderr, "[cf_ensure host, us] sorts + structure %.0f  layout + tables + enqueue %.0f  (blob %.1f MB uploaded, %.1f MB in all; sparse %d form %d)\n",
            t1 - t0, now_us() - t1, st.up_end / 1e6, st.total / 1e6, (int)sp, c->cf_sp_form);
  return ESL_OK;
}
// One LM trial with the cameras eliminated first: x_c, x_o, trial states, chi2 of the trial (same outputs as the other form)
static int slam_try_step_cf(esl_ctx* c, double lambda) {
  const DevGraph& g = c->g;
  const int N = g.n_objs, F = g.n_cams, nf = g.n_free_cams;
  const int n_o = 9 * N;
  int rc = cf_ensure(c);
  if (rc) return rc;
  const long ldx = (long)c->cf_ldx, ldt = (long)c->cf_ldt;
  CholRuntime& rt = chol_rt(c);
  {
    ProfScope ps(c, 2);   // "reduced-system build": factor of the camera block, X = L^-1 [W | b_c], T = D - X^T X
    ESL_HIP_TRY(hipMemsetAsync(c->chol_info, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(k_cf_gather_B, dim3((unsigned)(((long)nf * 36 + 255) / 256)), dim3(256), 0, c->stream, nf, c->Aod, c->cf_od_start, c->cf_od_edge, c->cf_B);
    const int st = c->cf_stride, ns = c->cf_n_sep, nseg = c->cf_n_seg;
    const bool nd = st > 0 && ns > 0;
    const unsigned cg = (unsigned)((n_o + 1 + 63) / 64);
    const long nt = (long)c->cf_n_list * 9 + nf;
    // V = Li W, vy = Li b_c; X dense: the same launch writes the whole of T (D's lower blocks, b_o's row, zeros) -- nothing reads T before
    // the rank-K update further down
    auto launch_edge_scale = [&]() {
      const int nb_es = (int)((nt + 255) / 256);
      if (!c->cf_sparse)
        hipLaunchKernelGGL(k_cf_edge_scale_T, dim3((unsigned)(nb_es + (int)((ldt * (long)n_o + 255) / 256))), dim3(256), 0, c->stream, g, nb_es, c->cf_n_list, c->cf_oe_u,
                           c->cf_oe_slot, c->Wbb, c->cf_Linv, c->bc, c->cf_V, c->cf_vy, c->Hoo, c->bo, lambda, c->cf_T, ldt, (long)n_o);
      else
        hipLaunchKernelGGL(k_cf_edge_scale, dim3((unsigned)nb_es), dim3(256), 0, c->stream, g, c->cf_n_list, c->cf_oe_u, c->cf_oe_slot, c->Wbb, c->cf_Linv, c->bc,
                           c->cf_V, c->cf_vy);
    };
    if (!nd) {
      hipLaunchKernelGGL(k_cf_chain, dim3(1), dim3(64), 0, c->stream, nf, c->Hcc, c->cf_B, lambda, c->cf_Lfac, c->cf_G, c->chol_info, nf, nf);
      hipLaunchKernelGGL(k_cf_factor_blocks, dim3((unsigned)((nf + 63) / 64)), dim3(64), 0, c->stream, nf, c->cf_Lfac, c->cf_G, c->cf_Linv, c->cf_M, c->cf_N, 0);
      launch_edge_scale();
      hipLaunchKernelGGL(k_cf_forward<0>, dim3(cg), dim3(64), 0, c->stream, nf, n_o, c->cf_n_chunks, c->cf_oe_cst, c->cf_oe_slot, c->cf_V, c->cf_vy,
                         c->cf_M, c->cf_Xt, ldx, nf, nf, (const double*)nullptr, (double*)nullptr, 0, CfSegs{});
    } else {   // nested dissection (esl_cf.hpp): segments in parallel, then the separators' own short chain
      hipLaunchKernelGGL(k_cf_chain, dim3((unsigned)nseg), dim3(64), 0, c->stream, nf, c->Hcc, c->cf_B, lambda, c->cf_Lfac, c->cf_G, c->chol_info, st, st - 1);
      hipLaunchKernelGGL(k_cf_factor_blocks, dim3((unsigned)((nf + 63) / 64)), dim3(64), 0, c->stream, nf, c->cf_Lfac, c->cf_G, c->cf_Linv, c->cf_M, c->cf_N, st);
      if (nseg > 1) hipLaunchKernelGGL(k_cf_zt, dim3((unsigned)(nseg - 1)), dim3(64), 0, c->stream, nf, st, c->cf_Linv, c->cf_M, c->cf_B, c->cf_Zt);
      hipLaunchKernelGGL(k_cf_sep_assemble, dim3((unsigned)((ns * 36 + 255) / 256)), dim3(256), 0, c->stream, nf, st, ns, c->Hcc, lambda, c->cf_G, c->cf_Zt,
                         c->cf_Hs, c->cf_Bs);
      hipLaunchKernelGGL(k_cf_chain, dim3(1), dim3(64), 0, c->stream, ns, c->cf_Hs, c->cf_Bs, 0.0, c->cf_LfacS, c->cf_GS, c->chol_info, ns, ns);
      hipLaunchKernelGGL(k_cf_factor_blocks, dim3((unsigned)((ns + 63) / 64)), dim3(64), 0, c->stream, ns, c->cf_LfacS, c->cf_GS, c->cf_LiS, c->cf_MS, c->cf_NS, 0,
                         c->cf_Linv, st);   // (+ the separators' L_ii^-1 into the per-slot array: was k_cf_sep_scatter)
      launch_edge_scale();
      if (!c->cf_sparse) {
        hipLaunchKernelGGL(k_cf_forward<0>, dim3(cg, (unsigned)nseg), dim3(64), 0, c->stream, nf, n_o, c->cf_n_chunks, c->cf_oe_cst, c->cf_oe_slot, c->cf_V,
                           c->cf_vy, c->cf_M, c->cf_Xt, ldx, st, st - 1, (const double*)c->cf_Zt, c->cf_R, st, CfSegs{});
        hipLaunchKernelGGL(k_cf_sep_rhs, dim3(cg, (unsigned)ns), dim3(64), 0, c->stream, nf, n_o, st, c->cf_n_chunks, c->cf_oe_cst, c->cf_oe_slot, c->cf_V, c->cf_vy,
                           c->cf_G, c->cf_LiS, c->cf_Xt, ldx, c->cf_R, (const int*)nullptr, CfSegs{});
        hipLaunchKernelGGL(k_cf_forward<1>, dim3(cg), dim3(64), 0, c->stream, ns, n_o, c->cf_n_chunks, c->cf_oe_cst, c->cf_oe_slot, c->cf_V, c->cf_vy,
                           c->cf_MS, c->cf_Xt, ldx, ns, ns, (const double*)nullptr, c->cf_R, st, CfSegs{});
      } else {   // interior rows into the segments' compact slabs, the separators' dense rows into Xs (row block k = separator k)
        const CfSegs sg{c->cf_fwork, c->cf_seg_start, c->cf_seg_obj, c->cf_xoff, c->cf_xld, c->cf_rank, c->cf_unrank};
        hipLaunchKernelGGL(k_cf_forward<2>, dim3((unsigned)c->cf_n_fwork), dim3(64), 0, c->stream, nf, n_o, c->cf_n_chunks, c->cf_oe_cst, c->cf_oe_slot, c->cf_V,
                           c->cf_vy, c->cf_M, c->cf_Xc, ldx, st, st - 1, (const double*)c->cf_Zt, c->cf_R, st, sg);
        hipLaunchKernelGGL(k_cf_sep_rhs, dim3(cg, (unsigned)ns), dim3(64), 0, c->stream, nf, n_o, st, c->cf_n_chunks, c->cf_oe_cst, c->cf_oe_slot, c->cf_V, c->cf_vy,
                           c->cf_G, c->cf_LiS, c->cf_Xc, ldx, c->cf_R, (const int*)c->cf_cmap, sg);
        hipLaunchKernelGGL(k_cf_forward<1>, dim3(cg), dim3(64), 0, c->stream, ns, n_o, c->cf_n_chunks, c->cf_oe_cst, c->cf_oe_slot, c->cf_V, c->cf_vy,
                           c->cf_MS, c->cf_Xs, ldx, ns, ns, (const double*)nullptr, c->cf_R, 0, CfSegs{});
      }
    }
    // this rank's columns of T: everything, or with the replicated-graph communicator its outer panels (dealt cyclically, as the
    // distributed factorisation below expects them)
    const int Wp = chol_outer_panels(n_o), n_outer = ((n_o + kNB - 1) / kNB + Wp - 1) / Wp;
    const bool dist = cf_dist(c);
    // Round 6: the segments' products (HBM-bound: 16.7 GB written) run on a second stream BESIDE the separators' rank-K update
    // (MFMA-bound): the update ASSIGNS T = -Xs^T Xs instead of updating what the gather wrote, and the gather, which now comes last,
    // adds its D - sum to that -- (D - sum) + (-acc) is the same double as (D - sum) - acc, so T has the same bits.  Single rank, X
    // kept sparse with stored products, the update on its k_chol_update_v path (the one with an assign form); ESL_CF_OVERLAP=0: the
    // serial order.
    static const bool overlap_on = !(std::getenv("ESL_CF_OVERLAP") && std::getenv("ESL_CF_OVERLAP")[0] == '0');
    const int Ks_ = (int)c->cf_kpad_s;
    const bool overlap = overlap_on && c->cf_sparse && c->cf_sp_form == 1 && !dist && Ks_ > 0 &&
                         chol_update_v_applies(ldt, (long)n_o + 1, 0, (long)n_o, true, ldx) && rt.cf_overlap_init() == hipSuccess;
    ESL_HIP_TRY(chol_set_attributes(rt));
    auto launch_products = [&](hipStream_t st2) {
      hipLaunchKernelGGL(k_cf_seg_syrk, dim3((unsigned)c->cf_n_twork), dim3(256), 0, st2, c->cf_twork, c->cf_seg_start, c->cf_seg_first, c->cf_xoff,
                         c->cf_xld, c->cf_Xc, c->cf_boff, c->cf_P);
      hipLaunchKernelGGL(k_cf_seg_rhs, dim3((unsigned)c->cf_n_fwork), dim3(64), 0, st2, c->cf_fwork, c->cf_seg_start, c->cf_xoff, c->cf_xld, c->cf_Xc,
                         c->cf_roff, c->cf_Prhs);
    };
    auto launch_gather = [&](int accumulate) {
      for (int op = dist ? c->comm_rank : 0; op < (dist ? n_outer : 1); op += dist ? c->comm_ranks : 1) {
        const long c_begin = dist ? (long)op * Wp * kNB : 0, c_end = dist ? std::min<long>((long)(op + 1) * Wp * kNB, (long)n_o) : (long)n_o;
        const int o2b = (int)(c_begin / 9), o2e = (int)((c_end + 8) / 9);   // (a block that straddles a panel boundary: on both sides)
        const dim3 grid((unsigned)(N + 1), (unsigned)((o2e - o2b + kCfTPer - 1) / kCfTPer));
        if (c->cf_sp_form == 1)
          hipLaunchKernelGGL(k_cf_T_gather, grid, dim3(256), 0, c->stream, N, c->cf_sp_nw, c->cf_mask, c->cf_cmap, c->cf_boff, c->cf_roff, c->cf_P, c->cf_Prhs,
                             c->Hoo, c->bo, lambda, c->cf_T, ldt, o2b, o2e, (const int*)c->cf_unrank, accumulate);
        else
          hipLaunchKernelGGL(k_cf_T_sparse, grid, dim3(256), 0, c->stream, N, c->cf_sp_nw, c->cf_mask, c->cf_cmap, c->cf_xoff, c->cf_xld, c->cf_Xc, c->Hoo, c->bo,
                             lambda, c->cf_T, ldt, o2b, o2e, (const int*)c->cf_unrank);
      }
    };
    if (overlap) {
      // (the slabs Xc were complete after k_cf_forward<2>; the event is recorded here, behind the separators' short chain, which the
      //  products do not need but which costs them 0.3 ms of head start at most)
      ESL_HIP_TRY(hipEventRecord(rt.cf_ev_x, c->stream));
      ESL_HIP_TRY(hipStreamWaitEvent(rt.cf_side, rt.cf_ev_x, 0));
      launch_products(rt.cf_side);
      ESL_HIP_TRY(hipEventRecord(rt.cf_ev_p, rt.cf_side));
      {
        ProfScope pk(c, 7);   // the rank-K update (assign form), with the products running beside it
        chol_launch_update(c->cf_T, ldt, (long)n_o + 1, c->stream, 0, Ks_, 0, (long)n_o, c->cf_Xs, ldx, nullptr, c->cf_kfirst, true);
      }
      ESL_HIP_TRY(hipStreamWaitEvent(c->stream, rt.cf_ev_p, 0));
      {
        ProfScope pk(c, 8);   // (what is left of the products' path on this stream: the gather)
        launch_gather(1);
      }
      ESL_HIP_TRY(hipGetLastError());
    } else {
    if (!c->cf_sparse) {   // (T was written whole by the edge-scale launch)
    } else {   // T = D - (interior rows)^T (interior rows), block by block (every block on and below the diagonal is written)
      ProfScope pk(c, 8);
      if (c->cf_sp_form == 1)   // the segments' products (all of them on every rank: 3 % of a trial)
        launch_products(c->stream);
      launch_gather(0);
    }
    ESL_HIP_TRY(hipGetLastError());
    {
      ProfScope pk(c, 7);   // the rank-K update alone (nested in class 2): the MFMA roofline kernel of this form
      const double* Xf = c->cf_sparse ? c->cf_Xs : c->cf_Xt;
      const int K = (int)(c->cf_sparse ? c->cf_kpad_s : c->cf_kpad);
      if (dist) {
        for (int op = c->comm_rank; op < n_outer; op += c->comm_ranks) {
          const long c_begin = (long)op * Wp * kNB, c_end = std::min<long>((long)(op + 1) * Wp * kNB, (long)n_o);
          chol_launch_update(c->cf_T, ldt, (long)n_o + 1, c->stream, 0, K, c_begin, c_end, Xf, ldx, nullptr, c->cf_sparse ? c->cf_kfirst : nullptr);
        }
      } else if (K > 0) {
        chol_launch_update(c->cf_T, ldt, (long)n_o + 1, c->stream, 0, K, 0, (long)n_o, Xf, ldx, c->cf_part, c->cf_sparse ? c->cf_kfirst : nullptr);
      }
    }
    }
    ESL_HIP_TRY(hipGetLastError());
  }
  {
    ProfScope ps(c, 3);   // dense Cholesky of the reduced ellipsoid system + the camera back-substitution
    double* xsol = (c->cf_sparse && c->cf_tperm) ? c->cf_xo_t : c->xo;   // the solver's x_o: in T's column order
    if (cf_dist(c)) {
      CholDist d;
      if ((rc = chol_dist_fill(c, n_o, d))) return rc;
      ESL_HIP_TRY(chol_factor_solve(c->cf_T, ldt, n_o, c->cf_Linv_ws, c->z_ws, xsol, c->chol_info, c->stream, rt, &d));
    } else {
      ESL_HIP_TRY(chol_factor_solve(c->cf_T, ldt, n_o, c->cf_Linv_ws, c->z_ws, xsol, c->chol_info, c->stream, rt));
    }
    if (xsol != c->xo) hipLaunchKernelGGL(k_cf_xo_unpermute, dim3((unsigned)((n_o + 255) / 256)), dim3(256), 0, c->stream, N, c->cf_rank, xsol, c->xo);
    if (c->cf_sparse)
      hipLaunchKernelGGL(k_cf_z_sparse, dim3((unsigned)(6 * nf)), dim3(256), 0, c->stream, n_o, c->cf_Xs, ldx,
                         CfSegs{c->cf_fwork, c->cf_seg_start, c->cf_seg_obj, c->cf_xoff, c->cf_xld, c->cf_rank, c->cf_unrank}, c->cf_Xc, c->xo, xsol, c->cf_z);
    else
      hipLaunchKernelGGL(k_cf_z, dim3((unsigned)(6 * nf)), dim3(256), 0, c->stream, c->cf_Xt, ldx, n_o, c->xo, c->cf_z);
    if (c->cf_stride > 0 && c->cf_n_sep > 0) {
      const int st = c->cf_stride, ns = c->cf_n_sep;
      hipLaunchKernelGGL(k_cf_tridiag_back, dim3(1), dim3(64), 0, c->stream, ns, c->cf_LiS, c->cf_NS, c->cf_z, c->xc, ns, ns, st);            // separators
      hipLaunchKernelGGL(k_cf_back_prep, dim3((unsigned)((6 * nf + 255) / 256)), dim3(256), 0, c->stream, nf, st, ns, c->cf_Zt, c->cf_G, c->xc, c->cf_z);
      hipLaunchKernelGGL(k_cf_tridiag_back, dim3((unsigned)c->cf_n_seg), dim3(64), 0, c->stream, nf, c->cf_Linv, c->cf_N, c->cf_z, c->xc, st, st - 1, 0);  // segments
    } else {
      hipLaunchKernelGGL(k_cf_tridiag_back, dim3(1), dim3(64), 0, c->stream, nf, c->cf_Linv, c->cf_N, c->cf_z, c->xc, nf, nf, 0);
    }
    ESL_HIP_TRY(hipGetLastError());
  }
  {
    ProfScope ps(c, 1);
    const int nb_obj = (N + 255) / 256, nb_cam = (F + 255) / 256;
    hipLaunchKernelGGL(k_cf_updates, dim3((unsigned)(nb_obj + nb_cam)), dim3(256), 0, c->stream, g, nb_obj, lambda, c->objs, c->bo, c->xo, c->objs_trial, c->obj_part,
                       c->cams, c->xc, c->bc, c->cams_trial, c->cam_part);
    const int nb_co = (N + kWavesPerBlock - 1) / kWavesPerBlock, nb_od = (g.n_odom + kWave * kWavesPerBlock - 1) / (kWave * kWavesPerBlock);
    hipLaunchKernelGGL(k_slam_chi2_all, dim3((unsigned)(nb_co + nb_od)), dim3(kWave * kWavesPerBlock), 0, c->stream, g, nb_co, c->cams_trial, c->objs_trial, c->obj_part,
                       c->od_part);
    ESL_HIP_TRY(hipGetLastError());
  }
  return ESL_OK;
}

// Sharded runs: factor the reduced camera system distributed over the ranks (esl_chol.hpp, CholDist) instead of all-reducing
// it and factoring it on every rank.  Default from 8,192 unknowns (below that the chain of diagonal blocks is the whole cost and
// a broadcast per panel only adds to it); ESL_CHOL_DIST=1 / 0 forces it on / off.
static bool slam_dist_chol(esl_ctx* c) {
  if (!c->comm || c->comm_ranks < 2) return false;
  if (c->sw_chol_dist >= 0) return c->sw_chol_dist == 1;   // ESL_CHOL_DIST as read when the communicator was created
  return c->S_n >= 8192;
}
// event bracket around the dense FACTORISATION alone (class 9, nested in class 3): k_chol_persist, or the launch-per-step chain
static void chol_prof_mark(void* user, int begin) {
  esl_ctx* c = (esl_ctx*)user;
  if (begin) { if (!c->chol_prof_scope) c->chol_prof_scope = new ProfScope(c, 9); }
  else if (c->chol_prof_scope) { delete (ProfScope*)c->chol_prof_scope; c->chol_prof_scope = nullptr; }
}
static CholRuntime& chol_rt(esl_ctx* c) {
  if (!c->chol_rt) {
    CholRuntime* rt = new CholRuntime();
    rt->prof_mark = chol_prof_mark; rt->prof_user = c;
    c->chol_rt = rt;
  }
  return *(CholRuntime*)c->chol_rt;
}
void slam_release_runtime(esl_ctx* c) {
  if (c->chol_prof_scope) { delete (ProfScope*)c->chol_prof_scope; c->chol_prof_scope = nullptr; }
  if (!c->chol_rt) return;
  ((CholRuntime*)c->chol_rt)->release();
  delete (CholRuntime*)c->chol_rt;
  c->chol_rt = nullptr;
  c->schur_attr_set = false;
}

// info: the dense solver's flag word, folded into the "ok" partial and handed on as out[4] (null: no fold -- linearisation, or a
// replicated-graph run, where the flag is only final after its collective: k_slam_fold_info there)
static int reduce_all(esl_ctx* c, const int* info = nullptr) {
  const DevGraph& g = c->g;
  hipLaunchKernelGGL(k_slam_reduce_all, dim3(1), dim3(256), 0, c->stream, c->obj_part, g.n_objs, c->cam_part, g.n_cams, c->od_part,
                     (g.n_odom && g.shard_rank == 0) ? g.n_odom : 0, c->dev_part, info);
  ESL_HIP_TRY(hipGetLastError());
  return ESL_OK;
}

int slam_linearize(esl_ctx* c) {
  c->parts_fresh = false;
  const DevGraph& g = c->g;
  const int N = g.n_objs, F = g.n_cams;
  const bool an = c->lm.p.jacobian_mode == ESL_JAC_ANALYTIC;
  const double delta = c->lm.p.numeric_delta;
  {
    ProfScope ps(c, 0);
    if (N > 0) {
      ChunkTable ct;
      ct.n_chunks = c->n_chunks; ct.obj = c->ck_obj; ct.type = c->ck_type; ct.begin = c->ck_begin; ct.end = c->ck_end; ct.ostart = c->ck_ostart;
      const dim3 block(64 * kLinWaves);
      const int nb_e3 = (c->n_ids_e3 + kLinWaves - 1) / kLinWaves, nb_bb = (c->n_ids_bb + kLinWaves - 1) / kLinWaves;
      // 3-D chunks first (the longer instruction stream), then the bbox chunks, then the per-ellipsoid sums
      if (an) {
        if (nb_e3) hipLaunchKernelGGL((k_slam_linearize_chunks<ESL_JAC_ANALYTIC, 1>), dim3(nb_e3), block, 0, c->stream, g, ct, c->ck_ids_e3, c->n_ids_e3, c->cams,
                                      c->objs, delta, c->chunk_out, c->Wbb, c->Abb);
        if (nb_bb) hipLaunchKernelGGL((k_slam_linearize_chunks<ESL_JAC_ANALYTIC, 0>), dim3(nb_bb), block, 0, c->stream, g, ct, c->ck_ids_bb, c->n_ids_bb, c->cams,
                                      c->objs, delta, c->chunk_out, c->Wbb, c->Abb);
        hipLaunchKernelGGL(k_slam_combine<ESL_JAC_ANALYTIC>, dim3((N + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kWave * kWavesPerBlock), 0, c->stream, g, ct,
                           c->chunk_out, c->objs, delta, c->Hoo, c->bo, c->obj_part);
      } else {
        if (nb_e3) hipLaunchKernelGGL((k_slam_linearize_chunks<ESL_JAC_NUMERIC, 1>), dim3(nb_e3), block, 0, c->stream, g, ct, c->ck_ids_e3, c->n_ids_e3, c->cams,
                                      c->objs, delta, c->chunk_out, c->Wbb, c->Abb);
        if (nb_bb) hipLaunchKernelGGL((k_slam_linearize_chunks<ESL_JAC_NUMERIC, 0>), dim3(nb_bb), block, 0, c->stream, g, ct, c->ck_ids_bb, c->n_ids_bb, c->cams,
                                      c->objs, delta, c->chunk_out, c->Wbb, c->Abb);
        hipLaunchKernelGGL(k_slam_combine<ESL_JAC_NUMERIC>, dim3((N + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kWave * kWavesPerBlock), 0, c->stream, g, ct,
                           c->chunk_out, c->objs, delta, c->Hoo, c->bo, c->obj_part);
      }
    }
    if (g.n_odom) {
      const dim3 grid((g.n_odom + 127) / 128), block(128);
      if (an) hipLaunchKernelGGL(k_slam_odom<ESL_JAC_ANALYTIC>, grid, block, 0, c->stream, g, c->cams, delta, c->Aod, c->od_part);
      else hipLaunchKernelGGL(k_slam_odom<ESL_JAC_NUMERIC>, grid, block, 0, c->stream, g, c->cams, delta, c->Aod, c->od_part);
    }
    hipLaunchKernelGGL(k_slam_cam_gather, dim3((F + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kWave * kWavesPerBlock), 0, c->stream, g, c->Abb, c->Aod,
                       c->Hcc, c->bc, c->cam_part);
    if (c->comm && !c->comm_replicated) {   // camera blocks of all shards: every rank ends up with the TOTAL Hcc, b_c (3.4 MB at 10k cameras)
      int rc2 = comm_allreduce_sum(c, c->Hcc, (size_t)g.n_free_cams * 36);
      if (!rc2) rc2 = comm_allreduce_sum(c, c->bc, (size_t)g.n_free_cams * 6);
      if (rc2) return rc2;
      hipLaunchKernelGGL(k_slam_cam_maxdiag, dim3((F + 127) / 128), dim3(128), 0, c->stream, g, c->Hcc, c->cam_part);
    }
  }
  ESL_HIP_TRY(hipGetLastError());
  ProfScope ps2(c, 4);
  return reduce_all(c);
}

int slam_build_reduced(esl_ctx* c, double lambda, bool full_sum, void** dev_ptr, int64_t* n_out) {
  const DevGraph& g = c->g;
  const int N = g.n_objs;
  const long n = (long)c->S_n, lda = (long)c->S_lda;
  if (const int rcs = slam_ensure_S(c)) return rcs;
  {
    ProfScope ps(c, 2);
    ESL_HIP_TRY(hipMemsetAsync(c->S, 0, (size_t)lda * (size_t)n * sizeof(double), c->stream));
    const int tot = g.n_free_cams + g.n_odom;
    hipLaunchKernelGGL(k_slam_S_init, dim3((tot + 127) / 128), dim3(128), 0, c->stream, g, c->Hcc, c->bc, c->Aod, lambda, c->S,
                       lda, n);
    if (N > 0) {
      const dim3 grid((N + kWavesPerBlock - 1) / kWavesPerBlock), block(kWave * kWavesPerBlock);
      hipLaunchKernelGGL(k_slam_prepare, grid, block, 0, c->stream, g, lambda, c->Hoo, c->bo, c->Wbb, c->Yb, c->Dinv, c->obj_part, c->Tb,
                         c->Wt);
      if (g.n_free_cams > 0) {
        // the row camera's Y blocks (54 doubles per edge, padded to 55), its list and its ellipsoid bitmap go through LDS when they
        // fit 64 KB; one workgroup per (row camera, group of 64 column cameras) on a 1-D grid (grid.y would cap the cameras at 65,535)
        const size_t ylds = schur_pull_lds_bytes(g.cu_max, g.n_objs);
        const unsigned groups = (unsigned)((g.n_free_cams + 63) / 64);
        const dim3 sgrid(groups * (unsigned)g.n_free_cams);
        if (ylds <= 65536) {
          if (!c->schur_attr_set) { ESL_HIP_TRY(hipFuncSetAttribute((const void*)k_slam_schur_pull<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536)); c->schur_attr_set = true; }
          hipLaunchKernelGGL(k_slam_schur_pull<true>, sgrid, dim3(64), ylds, c->stream, g, c->Wt, c->Yb, c->Tb, c->S, lda, n, groups);
        } else {
          hipLaunchKernelGGL(k_slam_schur_pull<false>, sgrid, dim3(64), 0, c->stream, g, c->Wt, c->Yb, c->Tb, c->S, lda, n, groups);
        }
      }
    }
  }
  ESL_HIP_TRY(hipGetLastError());
  if (c->comm && c->comm_replicated) {
    // whole graph on every rank: S is complete everywhere, nothing to sum
  } else if (c->comm && !full_sum && slam_dist_chol(c)) {
    // distributed factorisation: every outer panel of the summed system goes to its owner only (ncclReduce per panel)
    ProfScope ps(c, 6);
    const int W = chol_outer_panels((int)n), np = (int)((n + kNB - 1) / kNB), n_outer = (np + W - 1) / W;
    for (int o = 0; o < n_outer; ++o) {
      const long c_begin = (long)o * W * kNB, c_end = std::min<long>((long)(o + 1) * W * kNB, (long)n);
      const int rc2 = comm_reduce_sum_root(c, c->S + (size_t)c_begin * lda, (size_t)lda * (size_t)(c_end - c_begin), o % c->comm_ranks);
      if (rc2) return rc2;
    }
  } else if (c->comm) {   // sum of the shards' partial reduced systems (and b_s row): the RCCL all-reduce over xGMI of SURVEY.md §8 e
    ProfScope ps(c, 6);
    int rc2 = comm_allreduce_sum(c, c->S, (size_t)lda * (size_t)n);
    if (rc2) return rc2;
  }
  if (dev_ptr) *dev_ptr = c->S;
  if (n_out) *n_out = n;
  c->lm.lambda_used = lambda;
  return ESL_OK;
}

int slam_try_step(esl_ctx* c, double lambda) {
  c->parts_fresh = false;
  const DevGraph& g = c->g;
  const int N = g.n_objs, F = g.n_cams;
  int solver = slam_pick_solver(c);
  if (solver < 0) {
    set_error("ESL_SOLVER_REDUCED_ELLIPSOID needs odometry edges that join neighbouring free cameras only and, across GPUs, the "
              "replicated-graph communicator (esl_comm_set_replicated); a single GPU always qualifies");
    return ESL_ERR_INVALID;
  }
  int rc;
  if (solver == ESL_SOLVER_REDUCED_ELLIPSOID && c->lm.p.linear_solver == ESL_SOLVER_AUTO && (rc = cf_ensure(c)) != ESL_OK) {
    // AUTO: the camera-first form does not fit beside what else lives on the device -> the reduced camera system for this graph
    if (c->comm && c->comm_ranks > 1) return rc;   // (every rank must take the same form: no unilateral fall-back)
    c->cf_unavailable = true;
    solver = ESL_SOLVER_REDUCED_CAMERA;
  }
  c->lm_solver_used = solver;
  if (solver == ESL_SOLVER_REDUCED_ELLIPSOID) {
    if ((rc = slam_try_step_cf(c, lambda))) return rc;
  } else {
  if ((rc = slam_build_reduced(c, lambda, false, nullptr, nullptr))) return rc;
  {
    ProfScope ps(c, 3);
    ESL_HIP_TRY(hipMemsetAsync(c->chol_info, 0, sizeof(int), c->stream));
    if (c->comm && slam_dist_chol(c)) {
      CholDist d;
      { const int rcp = chol_dist_fill(c, (int)c->S_n, d); if (rcp) return rcp; }
      ESL_HIP_TRY(chol_factor_solve(c->S, (long)c->S_lda, (int)c->S_n, c->Linv_ws, c->z_ws, c->xc, c->chol_info, c->stream, chol_rt(c), &d));
    } else {
      ESL_HIP_TRY(chol_factor_solve(c->S, (long)c->S_lda, (int)c->S_n, c->Linv_ws, c->z_ws, c->xc, c->chol_info, c->stream, chol_rt(c)));
    }
  }
  {
    ProfScope ps(c, 1);
    const dim3 grid((N + kWavesPerBlock - 1) / kWavesPerBlock), block(kWave * kWavesPerBlock);
    if (N > 0)
      hipLaunchKernelGGL(k_slam_backsub, grid, block, 0, c->stream, g, lambda, c->objs, c->bo, c->Wbb, c->Dinv, c->xc, c->xo,
                         c->objs_trial, c->obj_part);
    hipLaunchKernelGGL(k_slam_cam_update, dim3((F + 127) / 128), dim3(128), 0, c->stream, g, lambda, c->cams, c->xc, c->bc,
                       c->cams_trial, c->cam_part);
    const int nb_co = (N + kWavesPerBlock - 1) / kWavesPerBlock, nb_od = (g.n_odom + kWave * kWavesPerBlock - 1) / (kWave * kWavesPerBlock);
    if (nb_co + nb_od > 0)
      hipLaunchKernelGGL(k_slam_chi2_all, dim3((unsigned)(nb_co + nb_od)), dim3(kWave * kWavesPerBlock), 0, c->stream, g, nb_co, c->cams_trial, c->objs_trial,
                         c->obj_part, c->od_part);
  }
  ESL_HIP_TRY(hipGetLastError());
  }
  const bool info_by_collective = c->comm && c->comm_replicated && c->comm_ranks > 1;
  {
    ProfScope ps2(c, 4);
    if ((rc = reduce_all(c, info_by_collective ? nullptr : c->chol_info))) return rc;
  }
  if (info_by_collective) {
    // replicated-graph run: no LM scalars are exchanged, so the pivot check of the panels' OWNERS has to reach every rank here
    // (one 8-byte all-reduce per trial) -- otherwise the owner would reject the step on its flag and the others on the NaNs that
    // came with the broadcast panel, two different exits of the trial loop
    hipLaunchKernelGGL(k_info_to_double, dim3(1), dim3(1), 0, c->stream, c->chol_info, c->dev_scal + 6);
    if ((rc = comm_allreduce_sum(c, c->dev_scal + 6, 1))) return rc;
    hipLaunchKernelGGL(k_double_to_info, dim3(1), dim3(1), 0, c->stream, c->dev_scal + 6, c->chol_info);
    hipLaunchKernelGGL(k_slam_fold_info, dim3(1), dim3(1), 0, c->stream, (const int*)c->chol_info, c->dev_part);
    ESL_HIP_TRY(hipGetLastError());
  }
  // the trial's four scalars and the solver's flag in ONE copy and ONE wait (the flag is already folded into the "ok" partial on the
  // device); esl_lm_try_step's read_parts finds them in the pinned block (parts_fresh)
  ESL_HIP_TRY(hipMemcpyAsync(c->host_part, c->dev_part, 5 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  ESL_HIP_TRY(hipStreamSynchronize(c->stream));
  const int info = (int)c->host_part[4];
  c->parts_fresh = true;
  if (info & 2) {
    // A spin of the one-launch forms ran into its wall-clock bound (3 s): the kernel was preempted or time-sliced for that long
    // (two processes on one GPU, a debugger, a profiler) or a producer never became resident.  The factorisation itself is not
    // wrong, so the trial is redone ONCE with the launch-per-step factorisation and back-substitution, which have no device-side
    // waits, and the context keeps that form from then on.  Not in a multi-rank run: the other ranks would not redo
    // their collectives.
    CholRuntime& rt = chol_rt(c);
    const bool can_retry = !(c->comm && c->comm_ranks > 1) && !(rt.sw_persistent == 0 && rt.sw_backsub == 1);
    if (!can_retry) { set_error("dense solver: a device-side hand-off timed out (k_chol_backsub / persistent panel kernel)"); return ESL_ERR_HIP; }
    rt.sw_persistent = 0; rt.sw_backsub = 1; rt.fallbacks += 1;
    fprintf(stderr, "[libesl_hip] a device-side hand-off of the dense solver timed out; this context now uses the launch-per-step form\n");
    ESL_HIP_TRY(hipMemsetAsync(c->chol_info, 0, sizeof(int), c->stream));
    c->parts_fresh = false;
    return slam_try_step(c, lambda);
  }
  return ESL_OK;
}

// ---- diagnostic: residual of the reduced camera system ------------------------------------------------------------------
// y += S x for the LOWER-triangle storage of S (column-major, lda): workgroup = (row tile of 256, column panel of 64) with
// row tile >= column panel; loads run down the columns (coalesced); an entry S_ij (i > j) serves y_i and y_j.
static __global__ __launch_bounds__(256) void k_symv_lower(const double* __restrict__ M, long lda, long n, const double* __restrict__ x,
                                                           double* __restrict__ y) {
  __shared__ double ycol[64];
  __shared__ double xcol[64];
  const long j0 = (long)blockIdx.y * 64, i0 = (long)blockIdx.x * 256;
  if (i0 + 255 < j0) return;                         // tile entirely above the diagonal
  const long i = i0 + threadIdx.x;
  if (threadIdx.x < 64) { ycol[threadIdx.x] = 0; xcol[threadIdx.x] = (j0 + threadIdx.x < n) ? x[j0 + threadIdx.x] : 0.0; }
  __syncthreads();
  const double xi = i < n ? x[i] : 0.0;
  double acc = 0;
  if (i < n) {
    for (int jj = 0; jj < 64; ++jj) {
      const long j = j0 + jj;
      if (j >= n || j > i) break;
      const double a = M[i + j * lda];
      acc += a * xcol[jj];
      if (i > j) atomicAdd(&ycol[jj], a * xi);
    }
    if (acc != 0) atomicAdd(&y[i], acc);
  }
  __syncthreads();
  if (threadIdx.x < 64 && j0 + threadIdx.x < n && ycol[threadIdx.x] != 0) atomicAdd(&y[j0 + threadIdx.x], ycol[threadIdx.x]);
}
static __global__ __launch_bounds__(256) void k_resid_norms(const double* __restrict__ M, long lda, long n, const double* __restrict__ y,
                                                            double* __restrict__ out2) {
  const long j = (long)blockIdx.x * 256 + threadIdx.x;
  double r2 = 0, b2 = 0;
  if (j < n) { const double b = M[n + j * lda], r = y[j] - b; r2 = r * r; b2 = b * b; }
  r2 = wave_sum(r2); b2 = wave_sum(b2);
  if ((threadIdx.x & 63) == 0) { atomicAdd(&out2[0], r2); atomicAdd(&out2[1], b2); }
}

}  // namespace esl

extern "C" int esl_lm_reduced_residual(esl_ctx* c, double* rel_residual_out) {
  using namespace esl;
  if (!c || !rel_residual_out) return ESL_ERR_INVALID;
  if (!c->lm.begun || !c->lm.slam || !c->xc || c->S_n <= 0) { set_error("esl_lm_reduced_residual: needs a SLAM-mode trial step"); return ESL_ERR_STATE; }
  // (works after a step of either form: x_c of the camera-first elimination solves the same reduced camera system)
  ESL_HIP_TRY(hipSetDevice(c->device));
  const long n = (long)c->S_n, lda = (long)c->S_lda;
  // the factorisation overwrote S and b_s; full_sum: in a sharded run EVERY rank needs the summed system here (the per-panel
  // reduce of the distributed factorisation would leave unsummed partials outside a rank's own panels)
  int rc = slam_build_reduced(c, c->lm.lambda_used, true, nullptr, nullptr);
  if (rc) return rc;
  double* y = nullptr;
  ESL_HIP_TRY(hipMalloc((void**)&y, ((size_t)n + 2) * sizeof(double)));
  hipError_t e = hipMemsetAsync(y, 0, ((size_t)n + 2) * sizeof(double), c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_symv_lower, dim3((unsigned)((n + 255) / 256), (unsigned)((n + 63) / 64)), dim3(256), 0, c->stream, c->S, lda, n,
                       c->xc, y);
    hipLaunchKernelGGL(k_resid_norms, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->S, lda, n, y, y + n);
    e = hipGetLastError();
  }
  double h2[2] = {0, 0};
  if (e == hipSuccess) e = hipMemcpyAsync(h2, y + n, sizeof(h2), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(y);
  if (e != hipSuccess) { set_error(std::string("esl_lm_reduced_residual: ") + hipGetErrorString(e)); return ESL_ERR_HIP; }
  *rel_residual_out = h2[1] > 0 ? std::sqrt(h2[0] / h2[1]) : 0.0;
  return ESL_OK;
}

// ---- dense-solver self test / micro-benchmark ------------------------------------------------------------------------
// A(i,j) = h(min,max) in [-1,1) off the diagonal (a hash, so the matrix never has to be kept), A(i,i) = n: strictly
// diagonally dominant, hence SPD.  b = A * 1 + known perturbation -> the solve must return x with A x = b.
namespace esl {
__device__ __forceinline__ double st_entry(long i, long j, long n) {
  if (i == j) return (double)n;
  const unsigned long long a = (unsigned long long)(i < j ? i : j), b = (unsigned long long)(i < j ? j : i);
  unsigned long long h = a * 0x9E3779B97F4A7C15ull ^ (b + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full;
  h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
  return (double)(h >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}
__device__ __forceinline__ double st_rhs(long i) { return 1.0 + 0.001 * (double)(i % 97); }
static __global__ void k_selftest_fill(double* __restrict__ M, long lda, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i > n || j >= n) return;
  double v = 0;
  if (i == n) v = st_rhs(j);          // row n carries b^T
  else if (i >= j) v = st_entry(i, j, n);
  M[i + j * lda] = v;
}
// one workgroup per row: r_i = b_i - sum_j A(i,j) x_j ;  accumulates |r|^2 and |b|^2
static __global__ __launch_bounds__(256) void k_selftest_resid(const double* __restrict__ x, long n, double* __restrict__ out2) {
  __shared__ double red[256];
  const long i = blockIdx.x;
  double s = 0;
  for (long j = threadIdx.x; j < n; j += 256) s += st_entry(i, j, n) * x[j];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
  if (threadIdx.x == 0) {
    const double b = st_rhs(i), r = b - red[0];
    atomicAdd(&out2[0], r * r);
    atomicAdd(&out2[1], b * b);
  }
}
}  // namespace esl

extern "C" int esl_selftest_cholesky(esl_ctx* c, int32_t n, double* ms_out, double* rel_residual_out) {
  using namespace esl;
  if (!c || n < 1 || !ms_out || !rel_residual_out) return ESL_ERR_INVALID;
  ESL_HIP_TRY(hipSetDevice(c->device));
  const long lda = (((long)n + 1 + 127) / 128) * 128;
  double *M = nullptr, *Linv = nullptr, *z = nullptr, *x = nullptr, *out2 = nullptr;
  int* info = nullptr;
  const size_t np = (size_t)((n + kNB - 1) / kNB);
  int rc = ESL_OK;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  auto cleanup = [&]() {
    if (M) (void)hipFree(M); if (Linv) (void)hipFree(Linv); if (z) (void)hipFree(z); if (x) (void)hipFree(x);
    if (out2) (void)hipFree(out2); if (info) (void)hipFree(info);
    if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1);
  };
#define ST_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); cleanup(); return ESL_ERR_HIP; } } while (0)
  ST_TRY(hipMalloc((void**)&M, (size_t)lda * (size_t)n * sizeof(double)));
  ST_TRY(hipMalloc((void**)&Linv, np * kNB * kNB * sizeof(double)));
  ST_TRY(hipMalloc((void**)&z, kNB * sizeof(double)));
  ST_TRY(hipMalloc((void**)&x, (size_t)n * sizeof(double)));
  ST_TRY(hipMalloc((void**)&out2, 2 * sizeof(double)));
  ST_TRY(hipMalloc((void**)&info, sizeof(int)));
  ST_TRY(hipEventCreate(&e0)); ST_TRY(hipEventCreate(&e1));
  ST_TRY(hipMemsetAsync(out2, 0, 2 * sizeof(double), c->stream));
  ST_TRY(hipMemsetAsync(info, 0, sizeof(int), c->stream));
  hipLaunchKernelGGL(k_selftest_fill, dim3((unsigned)((n + 1 + 255) / 256), (unsigned)n), dim3(256), 0, c->stream, M, lda, (long)n);
  ST_TRY(hipEventRecord(e0, c->stream));
  ST_TRY(chol_factor_solve(M, lda, n, Linv, z, x, info, c->stream, chol_rt(c)));
  ST_TRY(hipEventRecord(e1, c->stream));
  if (std::getenv("ESL_CHOL_DEBUG")) {   // a stream created and destroyed while the factorisation runs: the disturbance that made the
    hipStream_t w2 = nullptr;             // scheduler time-slice the first (two-kernel) persistent form
    if (hipStreamCreateWithFlags(&w2, hipStreamNonBlocking) == hipSuccess) (void)hipStreamDestroy(w2);
  }
  hipLaunchKernelGGL(k_selftest_resid, dim3((unsigned)n), dim3(256), 0, c->stream, x, (long)n, out2);
  double h2[2] = {0, 0};
  int hinfo = 0;
  ST_TRY(hipMemcpyAsync(h2, out2, sizeof(h2), hipMemcpyDeviceToHost, c->stream));
  ST_TRY(hipMemcpyAsync(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  ST_TRY(hipStreamSynchronize(c->stream));
  float ms = 0;
  ST_TRY(hipEventElapsedTime(&ms, e0, e1));
#undef ST_TRY
  if (std::getenv("ESL_CHOL_TIMING")) {
    std::vector<long long> ps((size_t)kPwGrid * 5, 0);
    if (hipMemcpyFromSymbol(ps.data(), HIP_SYMBOL(g_chol_stats), ps.size() * sizeof(long long)) == hipSuccess && ps[2] > 0) {
      // persistent kernel, last launch: workgroup 0 = the chain of diagonal blocks, the others pull tasks off the list
      long long t_first = ps[3], t_last = ps[4];
      double w_wait = 0, w_body = 0, w_tasks = 0, w_span = 0, w_wait_max = 0, w_end_min = 1e300, w_end_max = 0; int nw = 0;
      for (int b = 1; b < kPwGrid; ++b) {
        const long long* q = &ps[(size_t)b * 5];
        if (q[4] == 0) continue;
        t_first = std::min(t_first, q[3]); t_last = std::max(t_last, q[4]);
      }
      for (int b = 1; b < kPwGrid; ++b) {
        const long long* q = &ps[(size_t)b * 5];
        if (q[4] == 0) continue;
        ++nw; w_wait += (double)q[0]; w_body += (double)q[1]; w_tasks += (double)q[2]; w_span += (double)(q[4] - q[3]);
        w_wait_max = std::max(w_wait_max, (double)q[0]);
        w_end_min = std::min(w_end_min, (double)(q[4] - t_first)); w_end_max = std::max(w_end_max, (double)(q[4] - t_first));
      }
      fprintf(stderr, "[k_chol_persist, last launch] span %.2f ms | chain: %lld blocks, waiting %.2f ms, factoring %.2f ms | %d workers: mean waiting %.2f ms (max %.2f), "
                      "in task bodies %.2f ms, %.0f tasks each, alive %.2f ms, first / last exit at %.2f / %.2f ms\n",
              (double)(t_last - t_first) * 1e-5, ps[2], (double)ps[0] * 1e-5, (double)ps[1] * 1e-5, nw, nw ? w_wait / nw * 1e-5 : 0.0, w_wait_max * 1e-5,
              nw ? w_body / nw * 1e-5 : 0.0, nw ? w_tasks / nw : 0.0, nw ? w_span / nw * 1e-5 : 0.0, w_end_min * 1e-5, w_end_max * 1e-5);
    }
    {
      std::vector<long long> cl(2048, 0);
      const int npl = std::min(1024, (n + kNB - 1) / kNB);
      if (ps[2] > 0 && hipMemcpyFromSymbol(cl.data(), HIP_SYMBOL(g_chol_chain_log), cl.size() * sizeof(long long)) == hipSuccess && npl > 8) {
        // the chain's timeline, averaged over eighths of the panels: factor time and the gap between one factor published and the next tile final
        long long ft[4] = {0, 0, 0, 0};
        if (hipMemcpyFromSymbol(ft, HIP_SYMBOL(g_chol_fuse_ticks), sizeof(ft)) == hipSuccess && (ft[1] | ft[2]))
          fprintf(stderr, "[k_chol_persist chain, fused stage, us per panel] waits %.1f  rows under the block %.1f  next block's update %.1f  drain %.1f\n",
                  (double)ft[0] * 0.01 / npl, (double)ft[1] * 0.01 / npl, (double)ft[2] * 0.01 / npl, (double)ft[3] * 0.01 / npl);
        fprintf(stderr, "[k_chol_persist chain, us per panel by eighth of the matrix: potrf / gap to the next block]");
        for (int e = 0; e < 8; ++e) {
          const int k0 = e * npl / 8, k1 = std::min(npl - 1, (e + 1) * npl / 8);
          double f = 0, g = 0; int c = 0;
          for (int k = k0; k < k1; ++k) { f += (double)(cl[2 * k + 1] - cl[2 * k]); g += (double)(cl[2 * k + 2] - cl[2 * k + 1]); ++c; }
          if (c) fprintf(stderr, " %.0f/%.0f", f / c * 0.01, g / c * 0.01);
        }
        fprintf(stderr, "\n");
        {   // the same over the last three eighths (where the chain sets the pace), by the panel's position in its outer panel
          const int Wp = chol_outer_panels(n);
          std::vector<double> f(Wp, 0), g(Wp, 0); std::vector<int> cn(Wp, 0);
          for (int k = 5 * npl / 8; k < npl - 1; ++k) { f[k % Wp] += (double)(cl[2 * k + 1] - cl[2 * k]); g[k % Wp] += (double)(cl[2 * k + 2] - cl[2 * k + 1]); ++cn[k % Wp]; }
          fprintf(stderr, "[k_chol_persist chain, last 3/8 of the panels, us by k mod W: potrf / gap]");
          for (int q = 0; q < Wp; ++q) if (cn[q]) fprintf(stderr, " %.0f/%.0f", f[q] / cn[q] * 0.01, g[q] / cn[q] * 0.01);
          fprintf(stderr, "\n");
        }
      }
    }
    {
      long long uc[4] = {0, 0, 0, 0};
      if (hipMemcpyFromSymbol(uc, HIP_SYMBOL(g_upd_clk), sizeof(uc)) == hipSuccess && uc[1] > 0)
        fprintf(stderr, "[k_chol_update_lds, one %lld-row tile of the last launch, K = %lld] %.1f us at a shader clock of %.0f MHz -> MFMA peak at that clock %.1f TFLOP/s\n",
                uc[3], uc[2], (double)uc[1] * 0.01, (double)uc[0] / (double)uc[1] * 100.0, 78.6 * ((double)uc[0] / (double)uc[1] * 100.0) / 2400.0);
    }
    long long clk[16];
    if (hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_potrf_clk), sizeof(clk)) == hipSuccess) {
      static const char* names[9] = {"load", "rank-4 steps", "-", "-", "-", "store L", "-", "inverse", "store Linv"};
      fprintf(stderr, "[k_chol_potrf2, last launch, us into the rank-4 loop at step 1 / 8 / 16 / 24: %.1f %.1f %.1f %.1f]\n", (double)(clk[10] - clk[1]) * 0.01,
              (double)(clk[11] - clk[1]) * 0.01, (double)(clk[12] - clk[1]) * 0.01, (double)(clk[13] - clk[1]) * 0.01);
      fprintf(stderr, "[k_chol_potrf2, last launch, triangular inverse: levels 4 + 8 %.1f us, level 16 %.1f us, levels 32 + 64 %.1f us]\n", (double)(clk[14] - clk[7]) * 0.01,
              (double)(clk[15] - clk[14]) * 0.01, (double)(clk[8] - clk[15]) * 0.01);
      fprintf(stderr, "[k_chol_potrf2, last launch, us]");
      for (int k = 0; k < 9; ++k) fprintf(stderr, " %s=%.1f", names[k], (double)(clk[k + 1] - clk[k]) * 0.01);
      fprintf(stderr, "\n");
    }
  }
  *ms_out = ms;
  *rel_residual_out = h2[1] > 0 ? std::sqrt(h2[0] / h2[1]) : 0.0;
  if (hinfo & 2) { set_error("self test: a device-side hand-off of the dense solver timed out"); rc = ESL_ERR_HIP; }
  else if (hinfo) { set_error("self test: non-positive pivot"); rc = ESL_ERR_STATE; }
  cleanup();
  return rc;
}


extern "C" int esl_lm_solver_stats(esl_ctx* c, double* st) {
  if (!c || !st) return ESL_ERR_INVALID;
  for (int k = 0; k < ESL_SOLVER_STATS; ++k) st[k] = 0;
  if (c->lm_solver_used != ESL_SOLVER_REDUCED_ELLIPSOID || !c->cf_ready) return ESL_OK;
  st[0] = c->cf_sparse ? c->cf_sp_form : 0;
  st[1] = c->cf_stride; st[2] = c->cf_n_sep; st[3] = c->cf_n_seg;
  if (c->cf_sparse) {
    st[4] = 2.0 * c->cf_sp_flops;   // (FMAs of the block lower triangles, rows from each block row's first camera on)
    st[5] = c->cf_sp_form == 1 ? 8.0 * 81.0 * (double)c->cf_p_blocks : 0.0;
    st[6] = 8.0 * (double)c->cf_xc_len;
  }
  st[7] = (double)(c->cf_sparse ? c->cf_kpad_s : c->cf_kpad);
  st[8] = c->cf_tperm ? c->cf_upd_flops : 0.0;   // (0: the whole lower triangle times K)
  st[9] = c->cf_tperm ? 1.0 : 0.0;
  return ESL_OK;
}
extern "C" int esl_lm_solver_used(esl_ctx* c, int32_t* solver_out) {
  if (!c || !solver_out) return ESL_ERR_INVALID;
  *solver_out = c->lm_solver_used;
  return ESL_OK;
}

// test / debug: the task list of the persistent dense factorisation (esl_chol.hpp) for an order-n system with outer panels of W
// 128-panels -- a pure host function (tests/test_chol_plan.py replays it on the CPU: order, dependency words, result)
extern "C" int esl_debug_chol_plan(int32_t n, int32_t W, int32_t filler, int32_t* tasks_out /* 4 per task */, int64_t cap_tasks,
                                   int32_t* ns_out /* np x nR */, int64_t cap_ns, int32_t meta_out[5] /* np, n_outer, nR, n_tasks, W */) {
  if (n < 1 || W < 1 || !meta_out) return ESL_ERR_INVALID;
  esl::CholPlan pl;
  esl::chol_plan_build(n, W, filler, pl, esl::chol_fuse_default(), esl::chol_merge_default(n), esl::chol_strip_merge_default(), std::min(esl::chol_special_default(), (int)W));
  meta_out[0] = pl.np; meta_out[1] = pl.n_outer; meta_out[2] = pl.nR; meta_out[3] = (int32_t)pl.tasks.size(); meta_out[4] = pl.W;
  if (tasks_out) {
    if ((int64_t)pl.tasks.size() > cap_tasks) return ESL_ERR_INVALID;
    for (size_t i = 0; i < pl.tasks.size(); ++i) { tasks_out[4 * i] = pl.tasks[i].type; tasks_out[4 * i + 1] = pl.tasks[i].a; tasks_out[4 * i + 2] = pl.tasks[i].b; tasks_out[4 * i + 3] = pl.tasks[i].c; }
  }
  if (ns_out) {
    if ((int64_t)pl.ns.size() > cap_ns) return ESL_ERR_INVALID;
    for (size_t i = 0; i < pl.ns.size(); ++i) ns_out[i] = pl.ns[i];
  }
  return ESL_OK;
}
