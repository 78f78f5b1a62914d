// esl_capi.hip — C-ABI of libesl_hip.so (include/esl.h): context, graph upload (grow-only arenas), LM drivers.
//
// Mapping mode (cameras fixed, the shipped setting): esl_optimize_resident enqueues the whole Levenberg-Marquardt run
// ahead of the device; the control flow of OptimizationAlgorithmLevenberg::solve
// (Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-164) and SparseOptimizer::optimize
// (core/sparse_optimizer.cpp:354-419) runs in k_lm_step (esl_kernels_chunk.hpp), the host only watches a progress
// counter in mapped memory.  SLAM mode and the step API (esl_lm_*) keep the same statements on the host.  All arithmetic
// on states, residuals, Jacobians and normal equations runs in the HIP kernels of esl_kernels_*.hpp.  There is no CPU
// fallback: without a HIP device every compute entry point fails with ESL_ERR_NO_DEVICE.
#include <algorithm>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "esl_ctx.hpp"
#include "esl_kernels_map.hpp"
#include "esl_kernels_chunk.hpp"
#include "esl_slam.hpp"

namespace esl {
static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }
}  // namespace esl

using namespace esl;

namespace esl {
ProfScope::ProfScope(esl_ctx* ctx, int kind) : c(ctx), slot(-1) {
  if (!c || !c->prof_on) return;
  if (c->prof_level < 2 && (kind != 0 || !c->prof_gate)) return;
  if (c->prof_used + 2 > c->prof_ev.size()) {
    if (c->prof_ev.size() >= 16384) { prof_drain(c); }
    else {
      for (int i = 0; i < 512; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; c->prof_ev.push_back(e); }
      c->prof_kind.resize(c->prof_ev.size() / 2);
    }
  }
  slot = (int)c->prof_used;
  c->prof_kind[slot / 2] = kind;
  c->prof_used += 2;
  (void)hipEventRecord(c->prof_ev[slot], c->stream);
}
ProfScope::~ProfScope() {
  if (slot >= 0) (void)hipEventRecord(c->prof_ev[slot + 1], c->stream);
}
int prof_drain(esl_ctx* c) {
  if (hipStreamSynchronize(c->stream) != hipSuccess) return ESL_ERR_HIP;
  for (size_t s = 0; s + 1 < c->prof_used; s += 2) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->prof_ev[s], c->prof_ev[s + 1]) == hipSuccess) {
      const int k = c->prof_kind[s / 2];
      c->prof_count[k]++;
      c->prof_ms[k] += ms;
    }
  }
  c->prof_used = 0;
  return ESL_OK;
}
}  // namespace esl

template <class T>
static int dev_upload(T** dst, const T* src, size_t n, hipStream_t st) {
  if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
  ESL_HIP_TRY(hipMalloc((void**)dst, std::max<size_t>(n, 1) * sizeof(T)));
  if (n) ESL_HIP_TRY(hipMemcpyAsync(*dst, src, n * sizeof(T), hipMemcpyHostToDevice, st));
  return ESL_OK;
}
template <class T>
static int dev_alloc(T** dst, size_t n) {
  if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
  ESL_HIP_TRY(hipMalloc((void**)dst, std::max<size_t>(n, 1) * sizeof(T)));
  return ESL_OK;
}
template <class T>
static void dev_free(T** p) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
}

template <class T>
static void forget(T** p) { *p = nullptr; }   // interior pointer of an arena

// ---- arenas ---------------------------------------------------------------------------------------
// Device pointers handed out by the stages are INTERIOR pointers of c->arena_graph / c->arena_work: never hipFree()d
// one by one (free_graph only forgets them), the arenas grow and are released with the context.
static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static int arena_reserve(char** dev, size_t* cap, size_t need) {
  if (need <= *cap) return ESL_OK;
  if (*dev) { (void)hipFree(*dev); *dev = nullptr; *cap = 0; }
  const size_t want = need + need / 2 + 4096;
  ESL_HIP_TRY(hipMalloc((void**)dev, want));
  *cap = want;
  return ESL_OK;
}
struct UploadStage {   // arrays are packed straight into the context's pinned staging block (grown on demand, kept)
  struct Fix { void** dst; size_t off; };
  esl_ctx* c;
  size_t used = 0;
  int err = ESL_OK;
  std::vector<Fix> fixes;
  explicit UploadStage(esl_ctx* ctx) : c(ctx) {}
  bool reserve(size_t need) {
    if (need <= c->stage_host_cap) return true;
    const size_t want = std::max(need + need / 2 + 4096, 2 * c->stage_host_cap);
    char* nb = nullptr;
    if (hipHostMalloc((void**)&nb, want, hipHostMallocDefault) != hipSuccess) { esl::set_error("hipHostMalloc (upload staging) failed"); err = ESL_ERR_HIP; return false; }
    if (c->stage_host) { std::memcpy(nb, c->stage_host, used); (void)hipHostFree(c->stage_host); }
    c->stage_host = nb; c->stage_host_cap = want;
    return true;
  }
  template <class T>
  void add(T** dst, const T* src, size_t n) {
    const size_t off = align_up(used, 256), bytes = std::max<size_t>(n, 1) * sizeof(T);
    if (err || !reserve(off + bytes)) return;
    if (n) std::memcpy(c->stage_host + off, src, n * sizeof(T));
    used = off + bytes;
    fixes.push_back({(void**)dst, off});
  }
  // the caller fills the returned block itself (no intermediate copy); after begin() the block never moves
  template <class T>
  T* alloc(T** dst, size_t n) {
    const size_t off = align_up(used, 256), bytes = std::max<size_t>(n, 1) * sizeof(T);
    if (err || !reserve(off + bytes)) return nullptr;
    used = off + bytes;
    fixes.push_back({(void**)dst, off});
    return (T*)(c->stage_host + off);
  }
  // sizes both ends for `bound` bytes up front, so that flush() can ship finished sections while the host packs the next
  int begin(size_t bound) {
    if (!reserve(bound)) return err;
    if (int rc = arena_reserve(&c->arena_graph, &c->arena_graph_cap, std::max<size_t>(bound, 256))) return err = rc;
    early = true;
    return ESL_OK;
  }
  int flush() {
    if (err) return err;
    if (!early || used > c->arena_graph_cap) return ESL_OK;   // commit() ships everything
    if (used > sent) ESL_HIP_TRY(hipMemcpyAsync(c->arena_graph + sent, c->stage_host + sent, used - sent, hipMemcpyHostToDevice, c->stream));
    sent = used;
    return ESL_OK;
  }
  int commit() {
    if (err) return err;
    const size_t need = std::max<size_t>(used, 256);
    if (need > c->arena_graph_cap) sent = 0;   // the arena moves: ship the whole blob again
    int rc = arena_reserve(&c->arena_graph, &c->arena_graph_cap, need);
    if (rc) return rc;
    if (used > sent) ESL_HIP_TRY(hipMemcpyAsync(c->arena_graph + sent, c->stage_host + sent, used - sent, hipMemcpyHostToDevice, c->stream));
    sent = used;
    for (const Fix& f : fixes) *f.dst = c->arena_graph + f.off;
    return ESL_OK;
  }
  size_t sent = 0;
  bool early = false;
};
struct WorkStage {
  struct Fix { void** dst; size_t off; };
  size_t total = 0;
  std::vector<Fix> fixes;
  template <class T>
  void add(T** dst, size_t n) {
    const size_t off = align_up(total, 256);
    total = off + std::max<size_t>(n, 1) * sizeof(T);
    fixes.push_back({(void**)dst, off});
  }
  int commit(esl_ctx* c) {
    int rc = arena_reserve(&c->arena_work, &c->arena_work_cap, std::max<size_t>(total, 256));
    if (rc) return rc;
    for (const Fix& f : fixes) *f.dst = c->arena_work + f.off;
    return ESL_OK;
  }
};

extern "C" {

int esl_abi_version(void) { return ESL_ABI_VERSION; }
const char* esl_last_error(void) { return g_err.c_str(); }

int esl_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void esl_lm_params_default(esl_lm_params* p) {
  p->max_iters = 10;
  p->max_trials = 10;
  p->tau = 1e-5;
  p->jacobian_mode = ESL_JAC_NUMERIC;
  p->numeric_delta = 1e-9;
  p->linear_solver = ESL_SOLVER_AUTO;
  p->drop_nan_bbox = 1;
  p->bbox_residual = ESL_BBOX_REPROJECTION;
  p->e3d_half_turn = 0;
}

int esl_ctx_create(int device_id, esl_ctx** out) {
  if (!out) return ESL_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    set_error("no HIP device visible: libesl_hip has no CPU fallback");
    return ESL_ERR_NO_DEVICE;
  }
  if (device_id < 0 || device_id >= n) { set_error("device id out of range"); return ESL_ERR_INVALID; }
  ESL_HIP_TRY(hipSetDevice(device_id));
  esl_ctx* c = new esl_ctx();
  c->device = device_id;
  auto init = [&]() -> int {
    ESL_HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    ESL_HIP_TRY(hipHostMalloc((void**)&c->host_part, 16 * sizeof(double), hipHostMallocDefault));
    ESL_HIP_TRY(hipMalloc((void**)&c->dev_part, 16 * sizeof(double)));
    ESL_HIP_TRY(hipMalloc((void**)&c->chol_info, 4 * sizeof(int)));
    ESL_HIP_TRY(hipMemset(c->chol_info, 0, 4 * sizeof(int)));
    ESL_HIP_TRY(hipMalloc((void**)&c->tickets, 4 * sizeof(unsigned int)));
    ESL_HIP_TRY(hipMemset(c->tickets, 0, 4 * sizeof(unsigned int)));
    ESL_HIP_TRY(hipMalloc((void**)&c->dev_scal, 8 * sizeof(double)));
    ESL_HIP_TRY(hipEventCreateWithFlags(&c->ev_try, hipEventDisableTiming));
    ESL_HIP_TRY(hipHostMalloc(&c->host_scal, sizeof(LmScalars), hipHostMallocMapped));
    ESL_HIP_TRY(hipHostGetDevicePointer(&c->host_scal_dev, c->host_scal, 0));
    std::memset(c->host_scal, 0, sizeof(LmScalars));
    ESL_HIP_TRY(hipMalloc(&c->lm_dev, 2 * sizeof(LmCore)));   // ping-pong pair
    ESL_HIP_TRY(hipMemset(c->lm_dev, 0, 2 * sizeof(LmCore)));
    ESL_HIP_TRY(hipHostMalloc(&c->lm_host, sizeof(LmHostView), hipHostMallocMapped));
    ESL_HIP_TRY(hipHostGetDevicePointer(&c->lm_host_dev, c->lm_host, 0));
    std::memset(c->lm_host, 0, sizeof(LmHostView));
    return ESL_OK;
  };
  const int rc = init();
  if (rc) { const std::string msg = g_err; esl_ctx_destroy(c); set_error(msg); return rc; }   // a half-built context is released
  *out = c;
  return ESL_OK;
}

static void image_release(esl_ctx* c);
static void free_graph(esl_ctx* c) {
  DevGraph& g = c->g;
  image_release(c);   // the appendable host image describes the graph that goes away
  (void)hipStreamSynchronize(c->stream);   // nothing may still read the arenas that are about to be rewritten
  forget(&g.bb_start); forget(&g.e3_start); forget(&g.gr_cnt);
  forget(&g.bb_cam); forget(&g.bb_obj); forget(&g.bb_meas); forget(&g.bb_w); forget(&g.bb_valid);
  forget(&g.e3_cam); forget(&g.e3_obj); forget(&g.e3_meas); forget(&g.e3_w);
  forget(&g.od_i); forget(&g.od_j); forget(&g.od_meas); forget(&g.od_info);
  forget(&g.cam_fixed); forget(&g.cam_slot);
  forget(&g.cbb_start); forget(&g.cbb_edge); forget(&g.ce3_start); forget(&g.ce3_edge);
  forget(&g.cod_start); forget(&g.cod_edge);
  forget(&c->ck_obj); forget(&c->ck_type); forget(&c->ck_begin); forget(&c->ck_end); forget(&c->ck_ostart);
  forget(&c->chunk_out); forget(&c->chunk_out2); forget(&c->chunk_chi); forget(&c->blk_part); forget(&c->solve_part); forget(&c->blk_chi);
  forget(&c->ck_ids_bb); forget(&c->ck_ids_e3);
  c->n_chunks = 0;
  forget(&c->cams); forget(&c->cams_trial); forget(&c->objs); forget(&c->objs_trial);
  forget(&c->Hoo); forget(&c->bo); forget(&c->xo); forget(&c->obj_part);
  // SLAM-mode tables and buffers are interior pointers of the context's grow-only blobs (esl_slam.hip): forgotten, not freed
  slam_forget(c);
  dev_free(&c->chol_pack); dev_free(&c->chol_pack2); c->chol_pack_len = 0;
  dev_free(&c->cams_snap); dev_free(&c->objs_snap);
  c->S_n = 0;
  c->graph_loaded = false;
  c->states_loaded = false;
  c->lm.begun = false;
  c->repl_checked = false;
}

int esl_ctx_destroy(esl_ctx* c) {
  if (!c) return ESL_OK;
  (void)hipSetDevice(c->device);
  free_graph(c);
  slam_release(c);
  if (c->arena_graph) (void)hipFree(c->arena_graph);
  if (c->arena_work) (void)hipFree(c->arena_work);
  if (c->stage_host) (void)hipHostFree(c->stage_host);
  if (c->append_dev) (void)hipFree(c->append_dev);
  if (c->slam_tab_dev) (void)hipFree(c->slam_tab_dev);
  fit_release(c);
  plane_release(c);
  for (hipEvent_t e : c->prof_ev) (void)hipEventDestroy(e);
  if (c->host_part) (void)hipHostFree(c->host_part);
  dev_free(&c->dev_part);
  dev_free(&c->chol_info);
  dev_free(&c->tickets); dev_free(&c->dev_scal);
  if (c->host_scal) (void)hipHostFree(c->host_scal);
  if (c->lm_host) (void)hipHostFree(c->lm_host);
  if (c->lm_dev) (void)hipFree(c->lm_dev);
  if (c->ev_try) (void)hipEventDestroy(c->ev_try);
  esl_comm_destroy(c);
  slam_release_runtime(c);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return ESL_OK;
}

// Release the grow-only SOLVER blobs of SLAM mode (the camera-first set: 19.8 GB at BASELINE configs[3]; the reduced camera system:
// 28.8 GB) -- and, when the resident graph has no free cameras, the SLAM list blob too.  The next trial step that needs one
// builds it again (index tables included).  Device pointers obtained from esl_lm_reduced_system die here.
int esl_ctx_trim(esl_ctx* c) {
  if (!c) return ESL_ERR_INVALID;
  ESL_HIP_TRY(hipSetDevice(c->device));
  ESL_HIP_TRY(hipStreamSynchronize(c->stream));
  slam_trim(c, c->graph_loaded && c->g.n_free_cams > 0);
  return ESL_OK;
}

int esl_ctx_synchronize(esl_ctx* c) {
  if (!c) return ESL_ERR_INVALID;
  ESL_HIP_TRY(hipStreamSynchronize(c->stream));
  return ESL_OK;
}

// ---------------------------------------------------------------------------------------------------
// graph upload: validate, sort edges by ellipsoid (stable), build CSR
// ---------------------------------------------------------------------------------------------------
// order-independent structure fingerprint (comm_check_replicated): a 64-bit hash per edge / flag, summed
static inline unsigned long long fp_hash(unsigned long long a, unsigned long long b, unsigned long long c3) {
  unsigned long long h = 1469598103934665603ull;
  h = (h ^ a) * 1099511628211ull; h = (h ^ b) * 1099511628211ull; h = (h ^ c3) * 1099511628211ull;
  h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32;
  return h;
}
static unsigned long long fp_edges(int tag, int n, const int32_t* a, const int32_t* b) {
  unsigned long long s = 0;
  for (int i = 0; i < n; ++i) s += fp_hash((unsigned long long)tag, (unsigned)a[i], (unsigned)b[i]);
  return s;
}
static unsigned long long fp_fixed(int first_cam, int n, const uint8_t* fixed) {   // null = all fixed
  unsigned long long s = 0;
  for (int i = 0; i < n; ++i) s += fp_hash(4, (unsigned)(first_cam + i), fixed ? (fixed[i] ? 1u : 0u) : 1u);
  return s;
}
static int validate_graph(const esl_graph* g) {
  if (!g || g->n_cams < 0 || g->n_objs < 0 || g->n_bbox < 0 || g->n_e3d < 0 || g->n_grav < 0 || g->n_odom < 0) {
    set_error("esl_graph: null or negative size");
    return ESL_ERR_INVALID;
  }
  if ((g->n_bbox && (!g->bbox_cam || !g->bbox_obj || !g->bbox_meas || !g->bbox_weight)) ||
      (g->n_e3d && (!g->e3d_cam || !g->e3d_obj || !g->e3d_meas || !g->e3d_weight)) ||
      (g->n_grav && !g->grav_obj) || (g->n_odom && (!g->odom_i || !g->odom_j || !g->odom_meas))) {
    set_error("esl_graph: null edge array with non-zero count");
    return ESL_ERR_INVALID;
  }
  for (int i = 0; i < g->n_bbox; ++i)
    if (g->bbox_cam[i] < 0 || g->bbox_cam[i] >= g->n_cams || g->bbox_obj[i] < 0 || g->bbox_obj[i] >= g->n_objs) {
      set_error("esl_graph: bbox edge index out of range");
      return ESL_ERR_INVALID;
    }
  for (int i = 0; i < g->n_e3d; ++i)
    if (g->e3d_cam[i] < 0 || g->e3d_cam[i] >= g->n_cams || g->e3d_obj[i] < 0 || g->e3d_obj[i] >= g->n_objs) {
      set_error("esl_graph: 3-D edge index out of range");
      return ESL_ERR_INVALID;
    }
  for (int i = 0; i < g->n_grav; ++i)
    if (g->grav_obj[i] < 0 || g->grav_obj[i] >= g->n_objs) { set_error("esl_graph: gravity edge index out of range"); return ESL_ERR_INVALID; }
  for (int i = 0; i < g->n_odom; ++i)
    if (g->odom_i[i] < 0 || g->odom_i[i] >= g->n_cams || g->odom_j[i] < 0 || g->odom_j[i] >= g->n_cams) {
      set_error("esl_graph: odometry edge index out of range");
      return ESL_ERR_INVALID;
    }
  return ESL_OK;
}

static void csr_by_key(const int32_t* key, int n, int n_keys, int* start, int* perm, std::vector<int>& pos) {
  std::fill(start, start + n_keys + 1, 0);
  for (int i = 0; i < n; ++i) start[(size_t)key[i] + 1]++;
  for (int k = 0; k < n_keys; ++k) start[(size_t)k + 1] += start[k];
  pos.assign(start, start + n_keys);
  for (int i = 0; i < n; ++i) perm[(size_t)pos[key[i]]++] = i;
}
static void csr_by_key(const int32_t* key, int n, int n_keys, std::vector<int>& start, std::vector<int>& perm) {
  std::vector<int> pos;
  start.resize((size_t)n_keys + 1);
  perm.resize((size_t)n);
  csr_by_key(key, n, n_keys, start.data(), perm.data(), pos);
}

int esl_graph_upload(esl_ctx* c, const esl_graph* g) {
  if (!c) return ESL_ERR_INVALID;
  const bool host_timing = std::getenv("ESL_UPLOAD_HOST_TIMING") != nullptr;   // diagnostic: where the host call's time goes
  auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double ht[16]; int nht = 0;
  auto mark = [&] { if (host_timing && nht < 16) ht[nht++] = now_us(); };
  mark();
  int rc = validate_graph(g);
  if (rc) return rc;
  ESL_HIP_TRY(hipSetDevice(c->device));
  free_graph(c);
  c->graph_fp = fp_edges(1, g->n_bbox, g->bbox_cam, g->bbox_obj) + fp_edges(2, g->n_e3d, g->e3d_cam, g->e3d_obj) + fp_edges(3, g->n_odom, g->odom_i, g->odom_j) +
                fp_fixed(0, g->n_cams, g->cam_fixed);
  DevGraph& d = c->g;
  hipStream_t st = c->stream;
  // every array of the device-resident graph goes into ONE staging blob -> one H2D copy into a grow-only arena;
  // work buffers come from a second arena (a per-frame esl_optimize used to pay ~45 hipMalloc/hipFree pairs here)
  UploadStage up(c);
  WorkStage wk;
  d.n_cams = g->n_cams; d.n_objs = g->n_objs; d.n_bbox = g->n_bbox; d.n_e3d = g->n_e3d; d.n_odom = g->n_odom;
  d.n_bbox_edges = g->n_bbox;
  d.K[0] = g->fx; d.K[1] = g->fy; d.K[2] = g->cx; d.K[3] = g->cy;
  d.grav_n[0] = g->grav_normal[0]; d.grav_n[1] = g->grav_normal[1]; d.grav_n[2] = g->grav_normal[2];
  d.grav_w = g->grav_weight;
  d.check_vis = g->check_visibility ? 1 : 0; d.img_rows = g->image_rows; d.img_cols = g->image_cols;
  {  // rotate_ellipsoid's yaw table (src/core/Ellipsoid.cpp:78, 100): yaw = k*pi/2, k = -1,0,1,2
    const double ang[4] = {-1, 0, 1, 2};
    double hs[4], hc[4];
    for (int k = 0; k < 4; ++k) { const double yaw = ang[k] * M_PI / 2.0; hs[k] = std::sin(yaw * 0.5); hc[k] = std::cos(yaw * 0.5); }
    yaw_table_fill(d.yt, hs, hc);
  }
  const int N = g->n_objs, F = g->n_cams;
  {  // upper bound of the staging blob: lets finished sections go over PCIe while the host packs the next one
    auto blk = [](size_t n, size_t sz) { return align_up(std::max<size_t>(n, 1) * sz, 256) + 256; };
    const size_t nb = (size_t)g->n_bbox, ne = (size_t)g->n_e3d, no = (size_t)g->n_odom;
    const size_t max_chunks = nb / 64 + ne / 32 + 2 * (size_t)N + 2;
    const size_t bound = 3 * blk((size_t)N + 1, 4) + 3 * blk(nb, 4) + blk(nb * 4, 8) + blk(nb, 8) + blk(nb, 1) + 3 * blk((size_t)F + 1, 4) +
                         3 * blk(ne, 4) + blk(ne * 10, 8) + blk(ne, 8) + blk((size_t)N, 4) + 7 * blk(max_chunks, 4) + blk((size_t)F, 1) +
                         blk((size_t)F, 4) + 2 * blk(no, 4) + blk(no * 7, 8) + blk(no * 6, 8) + blk(no * 2, 4);
    if ((rc = up.begin(bound))) return rc;
  }
  std::vector<int> perm, perm3, pos, h_bb_start, h_e3_start;
  mark();   // 1: validated, staging sized
  {
    // edges are gathered in ellipsoid order straight into the pinned blob (C4: ~15 MB); the blob's edge arrays then go over
    // PCIe while the host builds the camera-side lists, the chunk table and the camera slots.  (Splitting the gather over
    // helper threads was measured and is not faster: profiles/r2_fused_experiments.txt)
    const size_t nb = (size_t)g->n_bbox, ne = (size_t)g->n_e3d;
    csr_by_key(g->bbox_obj, g->n_bbox, N, h_bb_start, perm);
    csr_by_key(g->e3d_obj, g->n_e3d, N, h_e3_start, perm3);
    int* b_start = up.alloc(&d.bb_start, (size_t)N + 1);
    int* b_cam = up.alloc(&d.bb_cam, nb);
    int* b_obj = up.alloc(&d.bb_obj, nb);
    double* b_meas = up.alloc(&d.bb_meas, nb * 4);
    double* b_w = up.alloc(&d.bb_w, nb);
    unsigned char* b_valid = up.alloc(&d.bb_valid, nb);
    int* e_start = up.alloc(&d.e3_start, (size_t)N + 1);
    int* e_cam = up.alloc(&d.e3_cam, ne);
    int* e_obj = up.alloc(&d.e3_obj, ne);
    double* e_meas = up.alloc(&d.e3_meas, ne * 10);
    double* e_w = up.alloc(&d.e3_w, ne);
    if (up.err) return up.err;
    std::copy(h_bb_start.begin(), h_bb_start.end(), b_start);
    std::copy(h_e3_start.begin(), h_e3_start.end(), e_start);
    mark();   // 2: both edge sets sorted by ellipsoid
    for (size_t k = 0; k < nb; ++k) {
      const size_t i = (size_t)perm[k];
      b_cam[k] = g->bbox_cam[i]; b_obj[k] = g->bbox_obj[i]; b_w[k] = g->bbox_weight[i];
      std::memcpy(b_meas + k * 4, g->bbox_meas + i * 4, 4 * sizeof(double));
    }
    if (nb) std::memset(b_valid, 1, nb);
    for (size_t k = 0; k < ne; ++k) {
      const size_t i = (size_t)perm3[k];
      e_cam[k] = g->e3d_cam[i]; e_obj[k] = g->e3d_obj[i]; e_w[k] = g->e3d_weight[i];
      std::memcpy(e_meas + k * 10, g->e3d_meas + i * 10, 10 * sizeof(double));
    }
    mark();   // 3: gathered
    if ((rc = up.flush())) return rc;
    c->h_bb_cam.assign(b_cam, b_cam + nb); c->h_bb_obj.assign(b_obj, b_obj + nb);
    c->h_e3_cam.assign(e_cam, e_cam + ne); c->h_e3_obj.assign(e_obj, e_obj + ne);
    // camera-side CSRs over the SORTED edges
    int* cs = up.alloc(&d.cbb_start, (size_t)F + 1);
    int* cp = up.alloc(&d.cbb_edge, nb);
    int* cs3 = up.alloc(&d.ce3_start, (size_t)F + 1);
    int* cp3 = up.alloc(&d.ce3_edge, ne);
    if (up.err) return up.err;
    csr_by_key(b_cam, g->n_bbox, F, cs, cp, pos);
    csr_by_key(e_cam, g->n_e3d, F, cs3, cp3, pos);
  }
  mark();   // 4: camera-side lists
  // gravity
  {
    std::vector<int> cnt((size_t)N, 0);
    for (int i = 0; i < g->n_grav; ++i) cnt[g->grav_obj[i]]++;
    c->n_grav_edges = g->n_grav;
    for (int o = 0; o < N; ++o)
      if (cnt[o] > 64) { set_error("more than 64 gravity edges on one ellipsoid"); return ESL_ERR_INVALID; }
    up.add(&d.gr_cnt, cnt.data(), cnt.size());
    // chunk table: per ellipsoid, in g2o's edge order (gravity, bbox, 3-D), <= 64 edges of one type per chunk
    std::vector<int> co, cty, cb, ce, cos((size_t)N + 1, 0);
    for (int o = 0; o < N; ++o) {
      cos[o] = (int)co.size();
      for (int b0 = h_bb_start[o]; b0 < h_bb_start[o + 1]; b0 += 64) { co.push_back(o); cty.push_back(0); cb.push_back(b0); ce.push_back(std::min(b0 + 64, h_bb_start[o + 1])); }
      for (int b0 = h_e3_start[o]; b0 < h_e3_start[o + 1]; b0 += 32) { co.push_back(o); cty.push_back(1); cb.push_back(b0); ce.push_back(std::min(b0 + 32, h_e3_start[o + 1])); }
    }
    cos[N] = (int)co.size();
    c->n_chunks = (int)co.size();
    up.add(&c->ck_obj, co.data(), co.size());
    up.add(&c->ck_type, cty.data(), cty.size());
    up.add(&c->ck_begin, cb.data(), cb.size());
    up.add(&c->ck_end, ce.data(), ce.size());
    up.add(&c->ck_ostart, cos.data(), cos.size());
    {
      std::vector<int> ib, ie;
      for (int k = 0; k < (int)cty.size(); ++k) (cty[k] == 0 ? ib : ie).push_back(k);
      c->n_ids_bb = (int)ib.size(); c->n_ids_e3 = (int)ie.size();
      up.add(&c->ck_ids_bb, ib.data(), ib.size());
      up.add(&c->ck_ids_e3, ie.data(), ie.size());
    }
    wk.add(&c->chunk_out, (size_t)c->n_chunks * kChunkOut);
    wk.add(&c->chunk_out2, (size_t)c->n_chunks * kChunkOut);
    wk.add(&c->chunk_chi, (size_t)c->n_chunks);
    wk.add(&c->blk_part, (size_t)((N + kStepWaves - 1) / kStepWaves + 2) * 2);   // k_chunk_finalize[_rows]: one pair per workgroup
    wk.add(&c->solve_part, (size_t)((N + kStepWaves - 1) / kStepWaves + (N + 63) / 64 + 2) * 4 * 2);   // k_obj_solve / k_lm_step* (x2: ping-pong)
    wk.add(&c->blk_chi, (size_t)(c->n_chunks + 2));                 // <= one workgroup per chunk
  }
  mark();   // 5: chunk table
  // cameras + odometry
  {
    std::vector<unsigned char> fixed((size_t)F, 1);
    std::vector<int> slot((size_t)F, -1);
    int nf = 0;
    if (g->cam_fixed)
      for (int i = 0; i < F; ++i) fixed[i] = g->cam_fixed[i] ? 1 : 0;
    // a free camera only enters the system if it has an active edge (sparse_optimizer.cpp:236-257)
    std::vector<unsigned char> touched((size_t)F, 0);
    for (int i = 0; i < g->n_bbox; ++i) touched[g->bbox_cam[i]] = 1;
    for (int i = 0; i < g->n_e3d; ++i) touched[g->e3d_cam[i]] = 1;
    for (int i = 0; i < g->n_odom; ++i)
      if (!(fixed[g->odom_i[i]] && fixed[g->odom_j[i]])) { touched[g->odom_i[i]] = 1; touched[g->odom_j[i]] = 1; }
    for (int i = 0; i < F; ++i)
      if (!fixed[i] && touched[i]) slot[i] = nf++;
    d.n_free_cams = nf;
    c->h_cam_slot = slot;
    c->h_od_i.assign(g->odom_i, g->odom_i + g->n_odom); c->h_od_j.assign(g->odom_j, g->odom_j + g->n_odom);
    up.add(&d.cam_fixed, fixed.data(), fixed.size());
    up.add(&d.cam_slot, slot.data(), slot.size());
    std::vector<double> info((size_t)g->n_odom * 6, 1.0);
    if (g->odom_info) std::copy(g->odom_info, g->odom_info + (size_t)g->n_odom * 6, info.begin());
    up.add(&d.od_i, g->odom_i, (size_t)g->n_odom);
    up.add(&d.od_j, g->odom_j, (size_t)g->n_odom);
    up.add(&d.od_meas, g->odom_meas, (size_t)g->n_odom * 7);
    up.add(&d.od_info, info.data(), info.size());
    // camera-side CSR over odometry edges: entry = edge*2 + side
    std::vector<int> key((size_t)g->n_odom * 2), cs, cp;
    for (int i = 0; i < g->n_odom; ++i) { key[(size_t)2 * i] = g->odom_i[i]; key[(size_t)2 * i + 1] = g->odom_j[i]; }
    csr_by_key(key.data(), g->n_odom * 2, F, cs, cp);
    up.add(&d.cod_start, cs.data(), cs.size());
    up.add(&d.cod_edge, cp.data(), cp.size());
  }
  // states + mapping-mode system
  wk.add(&c->cams, (size_t)F * 7);
  wk.add(&c->cams_trial, (size_t)F * 7);
  wk.add(&c->objs, (size_t)N * 10);
  wk.add(&c->objs_trial, (size_t)N * 10);
  wk.add(&c->Hoo, (size_t)N * 45);
  wk.add(&c->bo, (size_t)N * 9);
  wk.add(&c->xo, (size_t)N * 9);
  wk.add(&c->obj_part, (size_t)N * 4);
  mark();   // 6: cameras + odometry
  if ((rc = up.commit())) return rc;
  if ((rc = wk.commit(c))) return rc;
  ESL_HIP_TRY(hipMemsetAsync(c->obj_part, 0, std::max<size_t>(N, 1) * 4 * sizeof(double), st));
  if (d.n_free_cams == 0 && c->arena_solve.cap + c->arena_S.cap + c->arena_slam.cap > ((size_t)64 << 20)) {
    // a mapping-mode graph follows a SLAM-mode one: the solver blobs of the old graph (tens of GB at BASELINE configs[3]) would stay
    // pinned for the life of the context
    slam_trim(c, false);
  }
  if (d.n_free_cams > 0) {
    if ((rc = slam_alloc(c))) return rc;
  }
  mark();   // 7: commit + work arena + slam_alloc
  ESL_HIP_TRY(hipStreamSynchronize(st));
  mark();   // 8: copies landed
  if (host_timing) {
    std::fprintf(stderr, "esl_graph_upload us:");
    for (int i = 1; i < nht; ++i) std::fprintf(stderr, " %.0f", ht[i] - ht[i - 1]);
    std::fprintf(stderr, "  (validate, sort, gather, camera lists, chunks, cameras, commit, wait)\n");
  }
  c->graph_loaded = true;
  return ESL_OK;
}

// ---------------------------------------------------------------------------------------------------
// esl_graph_append: the appendable layout (mapping mode).  HostImage is the exact host copy of the device arrays: edges
// sorted by ellipsoid, every ellipsoid owning a slice [begin, begin + cap) of which cnt entries are used.
// ---------------------------------------------------------------------------------------------------
namespace {
struct HostImage {
  bool appendable = false;          // the device arrays have this image's capacities
  int n_cams = 0, n_objs = 0, cap_cams = 0, cap_objs = 0, n_grav = 0, relayouts = 0;
  size_t cap_bb = 0, cap_e3 = 0, used_bb = 0, used_e3 = 0;
  std::vector<int> bb_begin, bb_cnt, bb_cap, e3_begin, e3_cnt, e3_cap, gr_cnt;
  std::vector<int> bb_cam, bb_obj, e3_cam, e3_obj;
  std::vector<double> bb_meas, bb_w, e3_meas, e3_w;
  // SLAM mode (round 5): which cameras are fixed (empty = all of them) and the odometry edges
  std::vector<unsigned char> cam_fixed;
  std::vector<int> od_i, od_j;
  std::vector<double> od_meas, od_info;
  bool slam() const {
    if (!od_i.empty()) return true;
    for (unsigned char f : cam_fixed) if (!f) return true;
    return false;
  }
};
size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// chunk table of an image: per ellipsoid its bbox chunks (<= 64 edges) then its 3-D chunks (<= 32), as esl_graph_upload
void image_chunks(const HostImage& im, std::vector<int>& co, std::vector<int>& cty, std::vector<int>& cb, std::vector<int>& ce,
                  std::vector<int>& cos, std::vector<int>& ib, std::vector<int>& ie) {
  co.clear(); cty.clear(); cb.clear(); ce.clear(); ib.clear(); ie.clear();
  cos.assign((size_t)im.n_objs + 1, 0);
  for (int o = 0; o < im.n_objs; ++o) {
    cos[o] = (int)co.size();
    const int b0 = im.bb_begin[o], b1 = b0 + im.bb_cnt[o], e0 = im.e3_begin[o], e1 = e0 + im.e3_cnt[o];
    for (int b = b0; b < b1; b += 64) { ib.push_back((int)co.size()); co.push_back(o); cty.push_back(0); cb.push_back(b); ce.push_back(std::min(b + 64, b1)); }
    for (int b = e0; b < e1; b += 32) { ie.push_back((int)co.size()); co.push_back(o); cty.push_back(1); cb.push_back(b); ce.push_back(std::min(b + 32, e1)); }
  }
  cos[im.n_objs] = (int)co.size();
}
}  // namespace

static void image_release(esl_ctx* c) {
  delete (HostImage*)c->append_img;
  c->append_img = nullptr;
}

// the image of the graph that is resident now (compact layout of esl_graph_upload), read back from the device
static int image_from_device(esl_ctx* c, HostImage& im) {
  const DevGraph& g = c->g;
  const int N = g.n_objs;
  im = HostImage();
  im.n_cams = g.n_cams; im.n_objs = N; im.n_grav = c->n_grav_edges;
  std::vector<int> bs((size_t)N + 1), es((size_t)N + 1);
  im.gr_cnt.assign((size_t)N, 0);
  im.bb_cam.resize(g.n_bbox); im.bb_obj.resize(g.n_bbox); im.bb_meas.resize((size_t)g.n_bbox * 4); im.bb_w.resize(g.n_bbox);
  im.e3_cam.resize(g.n_e3d); im.e3_obj.resize(g.n_e3d); im.e3_meas.resize((size_t)g.n_e3d * 10); im.e3_w.resize(g.n_e3d);
  auto dl = [&](void* dst, const void* src, size_t bytes) -> int {
    if (bytes) ESL_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    return ESL_OK;
  };
  int rc;
  if ((rc = dl(bs.data(), g.bb_start, bs.size() * 4)) || (rc = dl(es.data(), g.e3_start, es.size() * 4)) || (rc = dl(im.gr_cnt.data(), g.gr_cnt, (size_t)N * 4)) ||
      (rc = dl(im.bb_cam.data(), g.bb_cam, (size_t)g.n_bbox * 4)) || (rc = dl(im.bb_obj.data(), g.bb_obj, (size_t)g.n_bbox * 4)) ||
      (rc = dl(im.bb_meas.data(), g.bb_meas, (size_t)g.n_bbox * 32)) || (rc = dl(im.bb_w.data(), g.bb_w, (size_t)g.n_bbox * 8)) ||
      (rc = dl(im.e3_cam.data(), g.e3_cam, (size_t)g.n_e3d * 4)) || (rc = dl(im.e3_obj.data(), g.e3_obj, (size_t)g.n_e3d * 4)) ||
      (rc = dl(im.e3_meas.data(), g.e3_meas, (size_t)g.n_e3d * 80)) || (rc = dl(im.e3_w.data(), g.e3_w, (size_t)g.n_e3d * 8)))
    return rc;
  im.cam_fixed.assign((size_t)g.n_cams, 1);
  im.od_i.resize((size_t)g.n_odom); im.od_j.resize((size_t)g.n_odom); im.od_meas.resize((size_t)g.n_odom * 7); im.od_info.resize((size_t)g.n_odom * 6);
  if ((rc = dl(im.cam_fixed.data(), g.cam_fixed, (size_t)g.n_cams)) || (rc = dl(im.od_i.data(), g.od_i, (size_t)g.n_odom * 4)) ||
      (rc = dl(im.od_j.data(), g.od_j, (size_t)g.n_odom * 4)) || (rc = dl(im.od_meas.data(), g.od_meas, (size_t)g.n_odom * 56)) ||
      (rc = dl(im.od_info.data(), g.od_info, (size_t)g.n_odom * 48)))
    return rc;
  ESL_HIP_TRY(hipStreamSynchronize(c->stream));
  im.bb_begin.assign(bs.begin(), bs.end() - 1); im.e3_begin.assign(es.begin(), es.end() - 1);
  im.bb_cnt.resize(N); im.e3_cnt.resize(N);
  for (int o = 0; o < N; ++o) { im.bb_cnt[o] = bs[o + 1] - bs[o]; im.e3_cnt[o] = es[o + 1] - es[o]; }
  im.bb_cap = im.bb_cnt; im.e3_cap = im.e3_cnt;
  im.used_bb = im.cap_bb = (size_t)g.n_bbox; im.used_e3 = im.cap_e3 = (size_t)g.n_e3d;
  im.cap_cams = im.n_cams; im.cap_objs = N;
  return ESL_OK;
}

// a fresh image with slack: every slice doubled (+ 64 / 32), arrays and vertex counts with room for as much again
static void image_relayout(const HostImage& old, HostImage& im, const std::vector<int>& add_bb, const std::vector<int>& add_e3, int n_cams, int n_objs) {
  im = HostImage();
  im.appendable = true;
  im.relayouts = old.relayouts + 1;
  im.n_cams = n_cams; im.n_objs = n_objs; im.n_grav = old.n_grav;
  im.cam_fixed = old.cam_fixed; im.od_i = old.od_i; im.od_j = old.od_j; im.od_meas = old.od_meas; im.od_info = old.od_info;
  im.cap_cams = n_cams * 2 + 64; im.cap_objs = n_objs * 2 + 16;
  im.bb_begin.resize(n_objs); im.bb_cnt.assign(n_objs, 0); im.bb_cap.resize(n_objs);
  im.e3_begin.resize(n_objs); im.e3_cnt.assign(n_objs, 0); im.e3_cap.resize(n_objs);
  im.gr_cnt.assign(n_objs, 0);
  size_t pb = 0, pe = 0;
  for (int o = 0; o < n_objs; ++o) {
    const int cb = (o < old.n_objs ? old.bb_cnt[o] : 0) + add_bb[o], ce = (o < old.n_objs ? old.e3_cnt[o] : 0) + add_e3[o];
    im.bb_begin[o] = (int)pb; im.bb_cap[o] = (int)round_up((size_t)2 * cb + 64, 64); pb += im.bb_cap[o];
    im.e3_begin[o] = (int)pe; im.e3_cap[o] = (int)round_up((size_t)2 * ce + 32, 32); pe += im.e3_cap[o];
    if (o < old.n_objs) im.gr_cnt[o] = old.gr_cnt[o];
  }
  im.used_bb = pb; im.used_e3 = pe;
  im.cap_bb = pb + pb / 2 + 64 * 64; im.cap_e3 = pe + pe / 2 + 32 * 64;     // room for ellipsoids that do not exist yet
  im.bb_cam.assign(im.cap_bb, 0); im.bb_obj.assign(im.cap_bb, -1); im.bb_meas.assign(im.cap_bb * 4, 0.0); im.bb_w.assign(im.cap_bb, 0.0);
  im.e3_cam.assign(im.cap_e3, 0); im.e3_obj.assign(im.cap_e3, -1); im.e3_meas.assign(im.cap_e3 * 10, 0.0); im.e3_w.assign(im.cap_e3, 0.0);
  for (int o = 0; o < old.n_objs; ++o) {
    for (int k = 0; k < old.bb_cnt[o]; ++k) {
      const size_t s0 = (size_t)old.bb_begin[o] + k, s1 = (size_t)im.bb_begin[o] + k;
      im.bb_cam[s1] = old.bb_cam[s0]; im.bb_obj[s1] = old.bb_obj[s0]; im.bb_w[s1] = old.bb_w[s0];
      for (int q = 0; q < 4; ++q) im.bb_meas[s1 * 4 + q] = old.bb_meas[s0 * 4 + q];
    }
    im.bb_cnt[o] = old.bb_cnt[o];
    for (int k = 0; k < old.e3_cnt[o]; ++k) {
      const size_t s0 = (size_t)old.e3_begin[o] + k, s1 = (size_t)im.e3_begin[o] + k;
      im.e3_cam[s1] = old.e3_cam[s0]; im.e3_obj[s1] = old.e3_obj[s0]; im.e3_w[s1] = old.e3_w[s0];
      for (int q = 0; q < 10; ++q) im.e3_meas[s1 * 10 + q] = old.e3_meas[s0 * 10 + q];
    }
    im.e3_cnt[o] = old.e3_cnt[o];
  }
}

// SLAM mode of the appendable layout: everything that is indexed by CAMERA -- fixed flags, free-camera slots, the per-camera lists
// of bbox / 3-D / odometry edges (positions in the ellipsoid-sorted arrays, ascending: the order esl_graph_upload produces, so the
// camera blocks are summed in the same order) -- and the odometry arrays are rebuilt from the host image and shipped as ONE blob
// into a grow-only buffer of their own (a few bytes per edge: 2 MB at BASELINE configs[3], nothing at streaming sizes); then the
// solver-side lists of esl_slam.hip (slam_alloc) are rebuilt from the same mirror.  The edge records and the states stay where
// they are.
static int slam_tables_refresh(esl_ctx* c, const HostImage& im) {
  DevGraph& d = c->g;
  const int F = im.n_cams, N = im.n_objs, no = (int)im.od_i.size();
  std::vector<unsigned char> fixed((size_t)F, 1);
  for (int i = 0; i < F && i < (int)im.cam_fixed.size(); ++i) fixed[i] = im.cam_fixed[i] ? 1 : 0;
  // a free camera only enters the system if it has an active edge (sparse_optimizer.cpp:236-257)
  std::vector<unsigned char> touched((size_t)F, 0);
  std::vector<int> nbc((size_t)F + 1, 0), nec((size_t)F + 1, 0);
  size_t nb = 0, ne = 0;
  for (int o = 0; o < N; ++o) {
    for (int k = 0; k < im.bb_cnt[o]; ++k) { const int cm = im.bb_cam[(size_t)im.bb_begin[o] + k]; touched[cm] = 1; ++nbc[(size_t)cm + 1]; ++nb; }
    for (int k = 0; k < im.e3_cnt[o]; ++k) { const int cm = im.e3_cam[(size_t)im.e3_begin[o] + k]; touched[cm] = 1; ++nec[(size_t)cm + 1]; ++ne; }
  }
  for (int e = 0; e < no; ++e)
    if (!(fixed[im.od_i[e]] && fixed[im.od_j[e]])) { touched[im.od_i[e]] = 1; touched[im.od_j[e]] = 1; }
  std::vector<int> slot((size_t)F, -1);
  int nf = 0;
  for (int i = 0; i < F; ++i) if (!fixed[i] && touched[i]) slot[i] = nf++;
  for (int i = 0; i < F; ++i) { nbc[(size_t)i + 1] += nbc[i]; nec[(size_t)i + 1] += nec[i]; }
  std::vector<int> cbe(std::max<size_t>(nb, 1)), cee(std::max<size_t>(ne, 1));
  {
    std::vector<int> pb(nbc.begin(), nbc.end() - 1), pe(nec.begin(), nec.end() - 1);
    for (int o = 0; o < N; ++o) {   // ascending position = ellipsoids ascending, arrival order inside one
      for (int k = 0; k < im.bb_cnt[o]; ++k) { const int at = im.bb_begin[o] + k; cbe[(size_t)pb[im.bb_cam[at]]++] = at; }
      for (int k = 0; k < im.e3_cnt[o]; ++k) { const int at = im.e3_begin[o] + k; cee[(size_t)pe[im.e3_cam[at]]++] = at; }
    }
  }
  std::vector<int> key((size_t)no * 2), cos, coe;
  for (int e = 0; e < no; ++e) { key[(size_t)2 * e] = im.od_i[e]; key[(size_t)2 * e + 1] = im.od_j[e]; }
  csr_by_key(key.data(), no * 2, F, cos, coe);
  // one staged blob
  struct Put { void** dst; const void* src; size_t bytes, off; };
  std::vector<Put> puts;
  size_t total = 0;
  auto put = [&](void* dstp, const void* src, size_t bytes) {
    const size_t off = align_up(total, 256);
    puts.push_back({(void**)dstp, src, bytes, off});
    total = off + std::max<size_t>(bytes, 8);
  };
  put(&d.cam_fixed, fixed.data(), (size_t)F); put(&d.cam_slot, slot.data(), (size_t)F * 4);
  put(&d.cbb_start, nbc.data(), nbc.size() * 4); put(&d.cbb_edge, cbe.data(), nb * 4);
  put(&d.ce3_start, nec.data(), nec.size() * 4); put(&d.ce3_edge, cee.data(), ne * 4);
  put(&d.cod_start, cos.data(), cos.size() * 4); put(&d.cod_edge, coe.data(), coe.size() * 4);
  put(&d.od_i, im.od_i.data(), (size_t)no * 4); put(&d.od_j, im.od_j.data(), (size_t)no * 4);
  put(&d.od_meas, im.od_meas.data(), (size_t)no * 56); put(&d.od_info, im.od_info.data(), (size_t)no * 48);
  UploadStage st(c);   // (the context's pinned staging block; the caller has synchronised the stream: it is free)
  if (!st.reserve(total)) return st.err;
  for (const Put& p : puts) if (p.bytes) std::memcpy(c->stage_host + p.off, p.src, p.bytes);
  int rc = arena_reserve(&c->slam_tab_dev, &c->slam_tab_cap, total);
  if (rc) return rc;
  ESL_HIP_TRY(hipMemcpyAsync(c->slam_tab_dev, c->stage_host, total, hipMemcpyHostToDevice, c->stream));
  for (const Put& p : puts) *p.dst = c->slam_tab_dev + p.off;
  d.n_free_cams = nf; d.n_odom = no;
  // the mirrors slam_alloc reads: position-indexed, free slots carry obj = -1
  c->h_cam_slot = slot;
  c->h_od_i = im.od_i; c->h_od_j = im.od_j;
  c->h_bb_cam.assign(im.bb_cam.begin(), im.bb_cam.begin() + (long)im.used_bb); c->h_bb_obj.assign(im.bb_obj.begin(), im.bb_obj.begin() + (long)im.used_bb);
  c->h_e3_cam.assign(im.e3_cam.begin(), im.e3_cam.begin() + (long)im.used_e3); c->h_e3_obj.assign(im.e3_obj.begin(), im.e3_obj.begin() + (long)im.used_e3);
  slam_forget(c);
  ESL_HIP_TRY(hipStreamSynchronize(c->stream));   // the staging block is free again (slam_alloc stages through its own blob)
  if (nf > 0 && (rc = slam_alloc(c))) return rc;
  return ESL_OK;
}

// the whole image to the device (capacities included) + the states; replaces what esl_graph_upload laid out
static int image_upload(esl_ctx* c, const HostImage& im, const double* cams, const double* objs, const double K[4], const double grav_n[3], double grav_w,
                        const YawTable& yt, int check_vis, int rows, int cols) {
  free_graph(c);
  DevGraph& d = c->g;
  UploadStage up(c);
  WorkStage wk;
  d = DevGraph();
  d.n_cams = im.n_cams; d.n_objs = im.n_objs; d.n_bbox = (int)im.used_bb; d.n_e3d = (int)im.used_e3; d.n_odom = 0;
  int nb = 0;
  for (int o = 0; o < im.n_objs; ++o) nb += im.bb_cnt[o];
  d.n_bbox_edges = nb;
  for (int k = 0; k < 4; ++k) d.K[k] = K[k];
  for (int k = 0; k < 3; ++k) d.grav_n[k] = grav_n[k];
  d.grav_w = grav_w; d.yt = yt; d.check_vis = check_vis; d.img_rows = rows; d.img_cols = cols;
  const size_t cN = (size_t)im.cap_objs, cF = (size_t)im.cap_cams;
  std::vector<int> start((size_t)im.cap_objs + 1, 0), gr((size_t)im.cap_objs, 0);
  // start[o] = begin of ellipsoid o's slice, start[n_objs] = end of the laid-out extent (a real end sentinel: a kernel that walks
  // start[o] .. start[o + 1] sees the slice WITH its slack, whose slots carry obj = -1, weight 0 and -- bbox -- valid = 0)
  for (int o = 0; o < im.n_objs; ++o) { start[o] = im.bb_begin[o]; gr[o] = im.gr_cnt[o]; }
  start[im.n_objs] = (int)im.used_bb;
  up.add(&d.bb_start, start.data(), start.size());
  for (int o = 0; o < im.n_objs; ++o) start[o] = im.e3_begin[o];
  start[im.n_objs] = (int)im.used_e3;
  up.add(&d.e3_start, start.data(), start.size());
  up.add(&d.gr_cnt, gr.data(), gr.size());
  up.add(&d.bb_cam, im.bb_cam.data(), im.cap_bb); up.add(&d.bb_obj, im.bb_obj.data(), im.cap_bb);
  up.add(&d.bb_meas, im.bb_meas.data(), im.cap_bb * 4); up.add(&d.bb_w, im.bb_w.data(), im.cap_bb);
  std::vector<unsigned char> valid(im.cap_bb, 0);   // slack slots stay invalid; an append marks the slots it fills
  for (int o = 0; o < im.n_objs; ++o)
    for (int k = 0; k < im.bb_cnt[o]; ++k) valid[(size_t)im.bb_begin[o] + k] = 1;
  up.add(&d.bb_valid, valid.data(), valid.size());
  up.add(&d.e3_cam, im.e3_cam.data(), im.cap_e3); up.add(&d.e3_obj, im.e3_obj.data(), im.cap_e3);
  up.add(&d.e3_meas, im.e3_meas.data(), im.cap_e3 * 10); up.add(&d.e3_w, im.e3_w.data(), im.cap_e3);
  std::vector<int> co, cty, cb, ce, cos, ib, ie;
  image_chunks(im, co, cty, cb, ce, cos, ib, ie);
  const size_t cap_chunks = im.cap_bb / 64 + im.cap_e3 / 32 + 2 * cN + 2;
  c->n_chunks = (int)co.size(); c->n_ids_bb = (int)ib.size(); c->n_ids_e3 = (int)ie.size();
  auto padded = [&](std::vector<int> v, size_t n) { v.resize(n, 0); return v; };
  { auto v = padded(co, cap_chunks); up.add(&c->ck_obj, v.data(), v.size()); }
  { auto v = padded(cty, cap_chunks); up.add(&c->ck_type, v.data(), v.size()); }
  { auto v = padded(cb, cap_chunks); up.add(&c->ck_begin, v.data(), v.size()); }
  { auto v = padded(ce, cap_chunks); up.add(&c->ck_end, v.data(), v.size()); }
  { auto v = padded(cos, cN + 1); up.add(&c->ck_ostart, v.data(), v.size()); }
  { auto v = padded(ib, cap_chunks); up.add(&c->ck_ids_bb, v.data(), v.size()); }
  { auto v = padded(ie, cap_chunks); up.add(&c->ck_ids_e3, v.data(), v.size()); }
  std::vector<unsigned char> fixed(cF, 1);
  std::vector<int> slot(cF, -1), one(2, 0);
  up.add(&d.cam_fixed, fixed.data(), fixed.size());
  up.add(&d.cam_slot, slot.data(), slot.size());
  d.n_free_cams = 0;
  c->h_cam_slot.assign(cF, -1); c->h_bb_cam.clear(); c->h_bb_obj.clear(); c->h_e3_cam.clear(); c->h_e3_obj.clear();
  const double dz[8] = {0};
  up.add(&d.od_i, one.data(), 1); up.add(&d.od_j, one.data(), 1); up.add(&d.od_meas, dz, 7); up.add(&d.od_info, dz, 6);
  up.add(&d.cbb_start, one.data(), 2); up.add(&d.cbb_edge, one.data(), 1); up.add(&d.ce3_start, one.data(), 2); up.add(&d.ce3_edge, one.data(), 1);
  up.add(&d.cod_start, one.data(), 2); up.add(&d.cod_edge, one.data(), 1);
  wk.add(&c->chunk_out, cap_chunks * kChunkOut); wk.add(&c->chunk_out2, cap_chunks * kChunkOut);
  wk.add(&c->chunk_chi, cap_chunks);
  wk.add(&c->blk_part, (size_t)((cN + kStepWaves - 1) / kStepWaves + 2) * 2);
  wk.add(&c->solve_part, (size_t)((cN + kStepWaves - 1) / kStepWaves + (cN + 63) / 64 + 2) * 4 * 2);
  wk.add(&c->blk_chi, cap_chunks + 2);
  wk.add(&c->cams, cF * 7); wk.add(&c->cams_trial, cF * 7);
  wk.add(&c->objs, cN * 10); wk.add(&c->objs_trial, cN * 10);
  wk.add(&c->Hoo, cN * 45); wk.add(&c->bo, cN * 9); wk.add(&c->xo, cN * 9); wk.add(&c->obj_part, cN * 4);
  int rc;
  if ((rc = up.commit())) return rc;
  if ((rc = wk.commit(c))) return rc;
  ESL_HIP_TRY(hipMemsetAsync(c->obj_part, 0, std::max<size_t>(cN, 1) * 4 * sizeof(double), c->stream));
  if (im.n_cams) ESL_HIP_TRY(hipMemcpyAsync(c->cams, cams, (size_t)im.n_cams * 7 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if (im.n_objs) ESL_HIP_TRY(hipMemcpyAsync(c->objs, objs, (size_t)im.n_objs * 10 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  ESL_HIP_TRY(hipStreamSynchronize(c->stream));
  c->n_grav_edges = im.n_grav;
  c->graph_loaded = true; c->states_loaded = true; c->cams_match_snap = false; c->lm.begun = false;
  if (im.slam()) return slam_tables_refresh(c, im);
  return ESL_OK;
}

int esl_graph_sizes(esl_ctx* c, int32_t* n_cams, int32_t* n_objs, int32_t* n_bbox, int32_t* n_e3d, int32_t* relayouts) {
  if (!c || !c->graph_loaded) return ESL_ERR_STATE;
  const HostImage* im = (const HostImage*)c->append_img;
  int ne = c->g.n_e3d;
  if (im) { ne = 0; for (int o = 0; o < im->n_objs; ++o) ne += im->e3_cnt[o]; }
  if (n_cams) *n_cams = c->g.n_cams;
  if (n_objs) *n_objs = c->g.n_objs;
  if (n_bbox) *n_bbox = c->g.n_bbox_edges;
  if (n_e3d) *n_e3d = ne;
  if (relayouts) *relayouts = im ? im->relayouts : 0;
  return ESL_OK;
}

int esl_graph_append(esl_ctx* c, const esl_graph_delta* dl) {
  if (!c || !dl) return ESL_ERR_INVALID;
  if (!c->graph_loaded || !c->states_loaded) { set_error("esl_graph_append: upload a graph and its states first"); return ESL_ERR_STATE; }
  if (c->comm) { set_error("esl_graph_append: one GPU only (no communicator)"); return ESL_ERR_STATE; }
  if (dl->n_odom < 0 || (dl->n_odom && (!dl->odom_i || !dl->odom_j || !dl->odom_meas))) { set_error("esl_graph_delta: null odometry array or negative size"); return ESL_ERR_INVALID; }
  if (dl->n_new_cams < 0 || dl->n_new_objs < 0 || dl->n_bbox < 0 || dl->n_e3d < 0 || dl->n_grav < 0 || (dl->n_new_cams && !dl->new_cams) ||
      (dl->n_new_objs && !dl->new_objs) || (dl->n_bbox && (!dl->bbox_cam || !dl->bbox_obj || !dl->bbox_meas || !dl->bbox_weight)) ||
      (dl->n_e3d && (!dl->e3d_cam || !dl->e3d_obj || !dl->e3d_meas || !dl->e3d_weight)) || (dl->n_grav && !dl->grav_obj)) {
    set_error("esl_graph_delta: null array or negative size");
    return ESL_ERR_INVALID;
  }
  ESL_HIP_TRY(hipSetDevice(c->device));
  int rc;
  if (!c->append_img) {
    HostImage* im0 = new HostImage();
    if ((rc = image_from_device(c, *im0))) { delete im0; return rc; }
    c->append_img = im0;
  }
  HostImage& im = *(HostImage*)c->append_img;
  const int F = im.n_cams + dl->n_new_cams, N = im.n_objs + dl->n_new_objs;
  for (int i = 0; i < dl->n_bbox; ++i)
    if (dl->bbox_cam[i] < 0 || dl->bbox_cam[i] >= F || dl->bbox_obj[i] < 0 || dl->bbox_obj[i] >= N) { set_error("esl_graph_delta: bbox edge index out of range"); return ESL_ERR_INVALID; }
  for (int i = 0; i < dl->n_e3d; ++i)
    if (dl->e3d_cam[i] < 0 || dl->e3d_cam[i] >= F || dl->e3d_obj[i] < 0 || dl->e3d_obj[i] >= N) { set_error("esl_graph_delta: 3-D edge index out of range"); return ESL_ERR_INVALID; }
  for (int i = 0; i < dl->n_grav; ++i)
    if (dl->grav_obj[i] < 0 || dl->grav_obj[i] >= N) { set_error("esl_graph_delta: gravity edge index out of range"); return ESL_ERR_INVALID; }
  for (int i = 0; i < dl->n_odom; ++i)
    if (dl->odom_i[i] < 0 || dl->odom_i[i] >= F || dl->odom_j[i] < 0 || dl->odom_j[i] >= F) { set_error("esl_graph_delta: odometry edge index out of range"); return ESL_ERR_INVALID; }
  c->graph_fp += fp_edges(1, dl->n_bbox, dl->bbox_cam, dl->bbox_obj) + fp_edges(2, dl->n_e3d, dl->e3d_cam, dl->e3d_obj) + fp_edges(3, dl->n_odom, dl->odom_i, dl->odom_j) +
                 fp_fixed(im.n_cams, dl->n_new_cams, dl->new_cam_fixed);
  c->repl_checked = false;
  auto add_cameras_and_odometry = [&](HostImage& I) {   // the delta's camera flags and odometry edges into an image whose n_cams is already F
    I.cam_fixed.resize((size_t)F, 1);
    for (int i = 0; i < dl->n_new_cams; ++i) I.cam_fixed[(size_t)(F - dl->n_new_cams + i)] = dl->new_cam_fixed ? (dl->new_cam_fixed[i] ? 1 : 0) : 1;
    for (int i = 0; i < dl->n_odom; ++i) {
      I.od_i.push_back(dl->odom_i[i]); I.od_j.push_back(dl->odom_j[i]);
      for (int q = 0; q < 7; ++q) I.od_meas.push_back(dl->odom_meas[(size_t)i * 7 + q]);
      for (int q = 0; q < 6; ++q) I.od_info.push_back(dl->odom_info ? dl->odom_info[(size_t)i * 6 + q] : 1.0);
    }
  };
  std::vector<int> add_bb((size_t)N, 0), add_e3((size_t)N, 0), add_gr((size_t)N, 0);
  for (int i = 0; i < dl->n_bbox; ++i) add_bb[dl->bbox_obj[i]]++;
  for (int i = 0; i < dl->n_e3d; ++i) add_e3[dl->e3d_obj[i]]++;
  for (int i = 0; i < dl->n_grav; ++i) add_gr[dl->grav_obj[i]]++;
  for (int o = 0; o < N; ++o)
    if ((o < im.n_objs ? im.gr_cnt[o] : 0) + add_gr[o] > 64) { set_error("more than 64 gravity edges on one ellipsoid"); return ESL_ERR_INVALID; }
  // does everything fit the slack of the resident layout?
  bool fits = im.appendable && F <= im.cap_cams && N <= im.cap_objs;
  size_t ub = im.used_bb, ue = im.used_e3;
  if (fits) {
    for (int o = 0; o < im.n_objs && fits; ++o) fits = im.bb_cnt[o] + add_bb[o] <= im.bb_cap[o] && im.e3_cnt[o] + add_e3[o] <= im.e3_cap[o];
    for (int o = im.n_objs; o < N; ++o) { ub += round_up((size_t)2 * add_bb[o] + 64, 64); ue += round_up((size_t)2 * add_e3[o] + 32, 32); }
    fits = fits && ub <= im.cap_bb && ue <= im.cap_e3;
  }
  const DevGraph gd = c->g;   // K, gravity, yaw table, visibility settings survive a re-layout
  std::vector<AppendBb> rb;
  std::vector<AppendE3> re;
  auto place = [&](HostImage& I, bool record) {   // the new edges into the free slots of I (arrival order inside an ellipsoid)
    for (int i = 0; i < dl->n_bbox; ++i) {
      const int o = dl->bbox_obj[i];
      const size_t s = (size_t)I.bb_begin[o] + I.bb_cnt[o]++;
      I.bb_cam[s] = dl->bbox_cam[i]; I.bb_obj[s] = o; I.bb_w[s] = dl->bbox_weight[i];
      for (int q = 0; q < 4; ++q) I.bb_meas[s * 4 + q] = dl->bbox_meas[(size_t)i * 4 + q];
      if (record) {
        AppendBb r{(int)s, dl->bbox_cam[i], o, 0, {0, 0, 0, 0}, dl->bbox_weight[i]};
        for (int q = 0; q < 4; ++q) r.meas[q] = dl->bbox_meas[(size_t)i * 4 + q];
        rb.push_back(r);
      }
    }
    for (int i = 0; i < dl->n_e3d; ++i) {
      const int o = dl->e3d_obj[i];
      const size_t s = (size_t)I.e3_begin[o] + I.e3_cnt[o]++;
      I.e3_cam[s] = dl->e3d_cam[i]; I.e3_obj[s] = o; I.e3_w[s] = dl->e3d_weight[i];
      for (int q = 0; q < 10; ++q) I.e3_meas[s * 10 + q] = dl->e3d_meas[(size_t)i * 10 + q];
      if (record) {
        AppendE3 r{(int)s, dl->e3d_cam[i], o, 0, {0}, dl->e3d_weight[i]};
        for (int q = 0; q < 10; ++q) r.meas[q] = dl->e3d_meas[(size_t)i * 10 + q];
        re.push_back(r);
      }
    }
    for (int o = 0; o < N; ++o) I.gr_cnt[o] += add_gr[o];
    I.n_grav += dl->n_grav;
  };
  if (!fits) {
    // full re-layout with doubled slack: the states that live on the device come back first
    std::vector<double> cams((size_t)F * 7), objs((size_t)N * 10);
    if (im.n_cams) ESL_HIP_TRY(hipMemcpyAsync(cams.data(), c->cams, (size_t)im.n_cams * 7 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (im.n_objs) ESL_HIP_TRY(hipMemcpyAsync(objs.data(), c->objs, (size_t)im.n_objs * 10 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    ESL_HIP_TRY(hipStreamSynchronize(c->stream));
    for (size_t k = 0; k < (size_t)dl->n_new_cams * 7; ++k) cams[(size_t)im.n_cams * 7 + k] = dl->new_cams[k];
    for (size_t k = 0; k < (size_t)dl->n_new_objs * 10; ++k) objs[(size_t)im.n_objs * 10 + k] = dl->new_objs[k];
    HostImage* ni = new HostImage();
    image_relayout(im, *ni, add_bb, add_e3, F, N);
    place(*ni, false);
    add_cameras_and_odometry(*ni);
    c->append_img = nullptr;   // free_graph (inside image_upload) must not drop the image under construction
    rc = image_upload(c, *ni, cams.data(), objs.data(), gd.K, gd.grav_n, gd.grav_w, gd.yt, gd.check_vis, gd.img_rows, gd.img_cols);
    delete &im;
    if (rc) { delete ni; return rc; }
    c->append_img = ni;
    dev_free(&c->cams_snap); dev_free(&c->objs_snap);
    return ESL_OK;
  }
  // incremental: slices of the new ellipsoids at the end of the arrays, edges into free slots, tables rebuilt on the host
  im.bb_begin.resize(N); im.bb_cnt.resize(N, 0); im.bb_cap.resize(N); im.e3_begin.resize(N); im.e3_cnt.resize(N, 0); im.e3_cap.resize(N);
  im.gr_cnt.resize(N, 0);
  for (int o = im.n_objs; o < N; ++o) {
    im.bb_begin[o] = (int)im.used_bb; im.bb_cap[o] = (int)round_up((size_t)2 * add_bb[o] + 64, 64); im.used_bb += im.bb_cap[o];
    im.e3_begin[o] = (int)im.used_e3; im.e3_cap[o] = (int)round_up((size_t)2 * add_e3[o] + 32, 32); im.used_e3 += im.e3_cap[o];
  }
  const int F0 = im.n_cams, N0 = im.n_objs;
  im.n_cams = F; im.n_objs = N;
  place(im, true);
  add_cameras_and_odometry(im);
  std::vector<int> co, cty, cb, ce, cos, ib, ie, bbs((size_t)N + 1, 0), e3s((size_t)N + 1, 0);
  image_chunks(im, co, cty, cb, ce, cos, ib, ie);
  for (int o = 0; o < N; ++o) { bbs[o] = im.bb_begin[o]; e3s[o] = im.e3_begin[o]; }
  bbs[N] = (int)im.used_bb; e3s[N] = (int)im.used_e3;   // end sentinel, as in image_upload
  // one staged blob: tables | bbox records | 3-D records
  UploadStage st(c);   // (re)uses the context's pinned staging block; `fixes` is not used here
  struct Part { size_t off, bytes; };
  auto put = [&](const void* src, size_t bytes) {
    const size_t off = align_up(st.used, 256);
    if (!st.reserve(off + std::max<size_t>(bytes, 8))) return Part{0, 0};
    if (bytes) std::memcpy(c->stage_host + off, src, bytes);
    st.used = off + std::max<size_t>(bytes, 8);
    return Part{off, bytes};
  };
  const Part p_co = put(co.data(), co.size() * 4), p_ty = put(cty.data(), cty.size() * 4), p_cb = put(cb.data(), cb.size() * 4), p_ce = put(ce.data(), ce.size() * 4),
             p_os = put(cos.data(), cos.size() * 4), p_ib = put(ib.data(), ib.size() * 4), p_ie = put(ie.data(), ie.size() * 4),
             p_gr = put(im.gr_cnt.data(), im.gr_cnt.size() * 4), p_bs = put(bbs.data(), bbs.size() * 4), p_es = put(e3s.data(), e3s.size() * 4),
             p_rb = put(rb.data(), rb.size() * sizeof(AppendBb)), p_re = put(re.data(), re.size() * sizeof(AppendE3));
  if (st.err) return st.err;
  if ((rc = arena_reserve(&c->append_dev, &c->append_dev_cap, st.used))) return rc;
  ESL_HIP_TRY(hipMemcpyAsync(c->append_dev, c->stage_host, st.used, hipMemcpyHostToDevice, c->stream));
  DevGraph& g = c->g;
  g.n_cams = F; g.n_objs = N; g.n_bbox = (int)im.used_bb; g.n_e3d = (int)im.used_e3; g.n_bbox_edges += dl->n_bbox;
  c->n_chunks = (int)co.size(); c->n_ids_bb = (int)ib.size(); c->n_ids_e3 = (int)ie.size();
  c->n_grav_edges = im.n_grav;
  AppendCopies cp{};
  auto tab = [&](const Part& p, int* dst) { cp.src[cp.n] = (const int*)(c->append_dev + p.off); cp.dst[cp.n] = dst; cp.count[cp.n] = (int)(p.bytes / 4); ++cp.n; };
  tab(p_co, c->ck_obj); tab(p_ty, c->ck_type); tab(p_cb, c->ck_begin); tab(p_ce, c->ck_end); tab(p_os, c->ck_ostart); tab(p_ib, c->ck_ids_bb);
  tab(p_ie, c->ck_ids_e3); tab(p_gr, g.gr_cnt); tab(p_bs, g.bb_start); tab(p_es, g.e3_start);
  hipLaunchKernelGGL(k_append_tables, dim3(16), dim3(256), 0, c->stream, cp);
  const int n_rec = (int)(rb.size() + re.size());
  if (n_rec)
    hipLaunchKernelGGL(k_append_scatter, dim3((n_rec + 127) / 128), dim3(128), 0, c->stream, g, (const AppendBb*)(c->append_dev + p_rb.off), (int)rb.size(),
                       (const AppendE3*)(c->append_dev + p_re.off), (int)re.size());
  ESL_HIP_TRY(hipGetLastError());
  if (dl->n_new_cams) ESL_HIP_TRY(hipMemcpyAsync(c->cams + (size_t)F0 * 7, dl->new_cams, (size_t)dl->n_new_cams * 7 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if (dl->n_new_objs) ESL_HIP_TRY(hipMemcpyAsync(c->objs + (size_t)N0 * 10, dl->new_objs, (size_t)dl->n_new_objs * 10 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  ESL_HIP_TRY(hipStreamSynchronize(c->stream));   // the caller's buffers and the staging block are free again
  dev_free(&c->cams_snap); dev_free(&c->objs_snap);
  c->cams_match_snap = false;
  c->lm.begun = false;
  if (im.slam()) {   // the camera-indexed tables + the solver's lists (SLAM mode)
    // Not failure-atomic by itself: records, states and the host image are committed above, the camera tables are rebuilt here.  If
    // that fails (out of memory in arena_reserve / slam_alloc) the device tables may be freed or describe the old graph -- drop the
    // graph altogether, so that the next call is an error ("upload graph first") and never a read of stale tables.
    const int rcs = slam_tables_refresh(c, im);
    if (rcs) { const std::string msg = esl_last_error(); free_graph(c); set_error(msg + " (esl_graph_append: the graph was dropped, upload it again)"); return rcs; }
  }
  return ESL_OK;
}

int esl_states_upload(esl_ctx* c, const double* cams, const double* objs) {
  if (!c || !c->graph_loaded) { set_error("esl_states_upload: no graph loaded"); return ESL_ERR_STATE; }
  if ((c->g.n_cams && !cams) || (c->g.n_objs && !objs)) return ESL_ERR_INVALID;
  ESL_HIP_TRY(hipSetDevice(c->device));
  if (c->g.n_cams) ESL_HIP_TRY(hipMemcpyAsync(c->cams, cams, (size_t)c->g.n_cams * 7 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if (c->g.n_objs) ESL_HIP_TRY(hipMemcpyAsync(c->objs, objs, (size_t)c->g.n_objs * 10 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  ESL_HIP_TRY(hipStreamSynchronize(c->stream));
  c->states_loaded = true;
  c->cams_match_snap = false;
  c->lm.begun = false;
  return ESL_OK;
}

int esl_states_download(esl_ctx* c, double* cams, double* objs) {
  if (!c || !c->states_loaded) { set_error("esl_states_download: no states"); return ESL_ERR_STATE; }
  ESL_HIP_TRY(hipSetDevice(c->device));
  if (cams && c->g.n_cams) ESL_HIP_TRY(hipMemcpyAsync(cams, c->cams, (size_t)c->g.n_cams * 7 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (objs && c->g.n_objs) ESL_HIP_TRY(hipMemcpyAsync(objs, c->objs, (size_t)c->g.n_objs * 10 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  ESL_HIP_TRY(hipStreamSynchronize(c->stream));
  return ESL_OK;
}


// ---------------------------------------------------------------------------------------------------
// chunked mapping-mode pipeline (esl_kernels_chunk.hpp): launch helpers
// ---------------------------------------------------------------------------------------------------
static ChunkTable chunk_table(const esl_ctx* c) {
  ChunkTable t;
  t.n_chunks = c->n_chunks; t.obj = c->ck_obj; t.type = c->ck_type; t.begin = c->ck_begin; t.end = c->ck_end; t.ostart = c->ck_ostart;
  return t;
}
// Linearise at `src_objs` into `dst_chunk` (defaults: the context's current estimate and chunk_out).
// st != null: the device-resident LM state picks the buffers instead -- the linearisation point is the TRIAL state of
// the pair (c->objs/chunk_out, c->objs_trial/chunk_out2) and the launch is a no-op once the run is done.
// validate: first linearisation of a device-driven run -- also performs the NaN pre-check of the bbox edges.
static int map_launch_linearize(esl_ctx* c, bool finalize, const double* src_objs = nullptr, double* dst_chunk = nullptr,
                                const LmCore* st = nullptr, bool validate = false) {
  const DevGraph& g = c->g;
  if (!src_objs) src_objs = c->objs;
  if (!dst_chunk) dst_chunk = c->chunk_out;
  const double* objs_b = st ? c->objs_trial : nullptr;
  double* chunk_b = st ? c->chunk_out2 : nullptr;
  if (st) { src_objs = c->objs; dst_chunk = c->chunk_out; }
  const ChunkTable ct = chunk_table(c);
  const bool an = c->lm.p.jacobian_mode == ESL_JAC_ANALYTIC;
  const int nb_e3 = (c->n_ids_e3 + 2 * kLinWaves - 1) / (2 * kLinWaves);   // two 32-edge chunks per wave
  const int nb_bb = (c->n_ids_bb + kLinWaves - 1) / kLinWaves;
  if (ct.n_chunks > 0) {
    ProfScope ps(c, 0);
    const dim3 block(64 * kLinWaves);
    int* cnt = c->chol_info + 2;
    if (an && g.bbox_mode) {   // plane-tangency rows instead of the reprojection residual (never NaN: nothing to validate, but the
                               // flags are (re)set to valid by the VALIDATE instantiation)
      if (validate && g.check_vis)
        hipLaunchKernelGGL((k_chunk_linearize_both<ESL_JAC_ANALYTIC, 2, true>), dim3(nb_e3 + nb_bb), block, 0, c->stream, g, ct, c->ck_ids_e3,
                           c->n_ids_e3, nb_e3, c->ck_ids_bb, c->n_ids_bb, c->cams, src_objs, objs_b, c->lm.p.numeric_delta, dst_chunk,
                           chunk_b, c->blk_chi, st, cnt);
      else if (validate)
        hipLaunchKernelGGL((k_chunk_linearize_both<ESL_JAC_ANALYTIC, 1, true>), dim3(nb_e3 + nb_bb), block, 0, c->stream, g, ct, c->ck_ids_e3,
                           c->n_ids_e3, nb_e3, c->ck_ids_bb, c->n_ids_bb, c->cams, src_objs, objs_b, c->lm.p.numeric_delta, dst_chunk,
                           chunk_b, c->blk_chi, st, cnt);
      else
        hipLaunchKernelGGL((k_chunk_linearize_both<ESL_JAC_ANALYTIC, 0, true>), dim3(nb_e3 + nb_bb), block, 0, c->stream, g, ct, c->ck_ids_e3,
                           c->n_ids_e3, nb_e3, c->ck_ids_bb, c->n_ids_bb, c->cams, src_objs, objs_b, c->lm.p.numeric_delta, dst_chunk,
                           chunk_b, c->blk_chi, st, cnt);
    } else if (an) {   // both edge types in one launch, 3-D workgroups first
      if (validate && g.check_vis)
        hipLaunchKernelGGL((k_chunk_linearize_both<ESL_JAC_ANALYTIC, 2>), dim3(nb_e3 + nb_bb), block, 0, c->stream, g, ct, c->ck_ids_e3,
                           c->n_ids_e3, nb_e3, c->ck_ids_bb, c->n_ids_bb, c->cams, src_objs, objs_b, c->lm.p.numeric_delta, dst_chunk,
                           chunk_b, c->blk_chi, st, cnt);
      else if (validate)
        hipLaunchKernelGGL((k_chunk_linearize_both<ESL_JAC_ANALYTIC, 1>), dim3(nb_e3 + nb_bb), block, 0, c->stream, g, ct, c->ck_ids_e3,
                           c->n_ids_e3, nb_e3, c->ck_ids_bb, c->n_ids_bb, c->cams, src_objs, objs_b, c->lm.p.numeric_delta, dst_chunk,
                           chunk_b, c->blk_chi, st, cnt);
      else
        hipLaunchKernelGGL((k_chunk_linearize_both<ESL_JAC_ANALYTIC, 0>), dim3(nb_e3 + nb_bb), block, 0, c->stream, g, ct, c->ck_ids_e3,
                           c->n_ids_e3, nb_e3, c->ck_ids_bb, c->n_ids_bb, c->cams, src_objs, objs_b, c->lm.p.numeric_delta, dst_chunk,
                           chunk_b, c->blk_chi, st, cnt);
    } else {    // numeric Jacobians: one kernel per edge type (very different register needs), the long tasks first
      if (nb_e3 > 0)
        hipLaunchKernelGGL((k_chunk_linearize<ESL_JAC_NUMERIC, 1, 0>), dim3(nb_e3), block, 0, c->stream, g, ct, c->ck_ids_e3,
                           c->n_ids_e3, c->cams, src_objs, objs_b, c->lm.p.numeric_delta, dst_chunk, chunk_b, c->blk_chi, 0, st, cnt);
      if (nb_bb > 0) {
        auto launch_bb = [&](auto kern) {
          hipLaunchKernelGGL(kern, dim3(nb_bb), block, 0, c->stream, g, ct, c->ck_ids_bb, c->n_ids_bb, c->cams, src_objs, objs_b,
                             c->lm.p.numeric_delta, dst_chunk, chunk_b, c->blk_chi, nb_e3, st, cnt);
        };
        const int vmode = !validate ? 0 : (g.check_vis ? 2 : 1);
        if (g.bbox_mode) { if (vmode == 2) launch_bb(k_chunk_linearize<ESL_JAC_NUMERIC, 0, 2, true>); else if (vmode == 1) launch_bb(k_chunk_linearize<ESL_JAC_NUMERIC, 0, 1, true>); else launch_bb(k_chunk_linearize<ESL_JAC_NUMERIC, 0, 0, true>); }
        else { if (vmode == 2) launch_bb(k_chunk_linearize<ESL_JAC_NUMERIC, 0, 2, false>); else if (vmode == 1) launch_bb(k_chunk_linearize<ESL_JAC_NUMERIC, 0, 1, false>); else launch_bb(k_chunk_linearize<ESL_JAC_NUMERIC, 0, 0, false>); }
      }
    }
  }
  ESL_HIP_TRY(hipGetLastError());
  c->sys_combined = false;
  if (finalize) {
    ProfScope ps(c, 4);
    hipLaunchKernelGGL(k_chunk_finalize_rows, dim3(std::max(1, (g.n_objs + kStepWaves - 1) / kStepWaves)), dim3(64 * kStepWaves), 0, c->stream, g, ct, dst_chunk, src_objs,
                       c->lm.p.jacobian_mode, c->lm.p.numeric_delta, c->blk_part,
                       c->tickets, c->dev_scal, (LmScalars*)c->host_scal_dev, c->lm.p.tau, (LmCore*)nullptr, (int*)nullptr, 0,
                       (LmHostView*)nullptr);
    ESL_HIP_TRY(hipGetLastError());
  }
  return ESL_OK;
}
// solve (H + lambda I) x = b per ellipsoid from the chunk partials, retract -> objs_trial.
// lambda < 0: use tau * max_diag from device memory (first iteration)
static int map_launch_solve(esl_ctx* c, double lambda, const double* chunk = nullptr) {
  const DevGraph& g = c->g;
  if (!chunk) chunk = c->chunk_out;
  ProfScope ps(c, 1);
  hipLaunchKernelGGL(k_obj_solve, dim3((g.n_objs + 63) / 64), dim3(64), 0, c->stream, g, chunk_table(c), chunk, c->objs,
                     c->lm.p.jacobian_mode, c->lm.p.numeric_delta, lambda,
                     c->lm.p.tau, c->dev_scal, c->xo, c->objs_trial, c->obj_part, c->solve_part);
  ESL_HIP_TRY(hipGetLastError());
  return ESL_OK;
}
// solve + chi2 of the trial state with the residual-only kernels (step API / sharded loop: H, b of the trial are not wanted)
static int map_launch_try(esl_ctx* c, double lambda, const double* chunk = nullptr) {
  const DevGraph& g = c->g;
  const ChunkTable ct = chunk_table(c);
  int rc = map_launch_solve(c, lambda, chunk);
  if (rc) return rc;
  {
    ProfScope ps(c, 1);
    // 3-D chunks first (no reduction), then the bbox chunks whose last workgroup reduces everything
    if (c->n_ids_e3 > 0)
      hipLaunchKernelGGL((k_chunk_chi2<1, false>), dim3((c->n_ids_e3 + 7) / 8), dim3(256), 0, c->stream, g, ct, c->ck_ids_e3, c->n_ids_e3,
                         c->cams, c->objs_trial, c->obj_part, c->chunk_chi, c->tickets + 1, lambda, c->lm.p.tau, c->dev_scal,
                         (LmScalars*)c->host_scal_dev);
    hipLaunchKernelGGL((k_chunk_chi2<0, true>), dim3(std::max(1, (c->n_ids_bb + 3) / 4)), dim3(256), 0, c->stream, g, ct, c->ck_ids_bb,
                       c->n_ids_bb, c->cams, c->objs_trial, c->obj_part, c->chunk_chi, c->tickets + 1, lambda, c->lm.p.tau, c->dev_scal,
                       (LmScalars*)c->host_scal_dev);
  }
  ESL_HIP_TRY(hipGetLastError());
  return ESL_OK;
}
static int map_combine(esl_ctx* c) {
  if (c->sys_combined || c->g.n_objs == 0) return ESL_OK;
  hipLaunchKernelGGL(k_chunk_combine, dim3((c->g.n_objs * 54 + 255) / 256), dim3(256), 0, c->stream, c->g, chunk_table(c),
                     c->chunk_out, c->objs, c->lm.p.jacobian_mode, c->lm.p.numeric_delta, c->Hoo, c->bo);
  ESL_HIP_TRY(hipGetLastError());
  c->sys_combined = true;
  return ESL_OK;
}

// ---------------------------------------------------------------------------------------------------
// step API
// ---------------------------------------------------------------------------------------------------
static int read_parts(esl_ctx* c, double out[4]) {
  if (c->parts_fresh) {   // (slam_try_step copied them together with the dense solver's flag: one wait per trial)
    c->parts_fresh = false;
    for (int i = 0; i < 4; ++i) out[i] = c->host_part[i];
    return ESL_OK;
  }
  ESL_HIP_TRY(hipMemcpyAsync(c->host_part, c->dev_part, 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  ESL_HIP_TRY(hipStreamSynchronize(c->stream));
  for (int i = 0; i < 4; ++i) out[i] = c->host_part[i];
  return ESL_OK;
}

// NaN pre-check of the bbox edges at the start state (Optimizer.cpp:234-243); the dropped count stays on the device
// (c->chol_info + 2) -- the synchronous caller reads it back, the device-driven run lets k_chunk_finalize report it.
static int lm_begin_enqueue(esl_ctx* c, const esl_lm_params* p, bool validate_in_linearize = false) {
  if (!c->graph_loaded || !c->states_loaded) { set_error("esl_lm_begin: upload graph and states first"); return ESL_ERR_STATE; }
  ESL_HIP_TRY(hipSetDevice(c->device));
  c->lm.p = *p;
  c->g.bbox_mode = p->bbox_residual == ESL_BBOX_TANGENCY ? 1 : 0;
  c->g.yt.as_written = p->e3d_half_turn ? 1 : 0;
  c->lm.slam = c->g.n_free_cams > 0;
  c->lm.have_trial = false;
  int* cnt = c->chol_info + 2;
  const bool validate = p->drop_nan_bbox != 0 || c->g.check_vis != 0;   // the visibility test rides on the NaN pre-check's pass
  if (validate_in_linearize && c->g.n_bbox && validate) {   // counter is zero: context creation, k_chunk_finalize, or esl_lm_begin's read-back
    c->lm.begun = true;
    return ESL_OK;
  }
  ESL_HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(int), c->stream));
  if (c->g.n_bbox) {
    if (validate) {
      hipLaunchKernelGGL(k_bbox_validate, dim3((c->g.n_bbox + 255) / 256), dim3(256), 0, c->stream, c->g, c->cams, c->objs, cnt);
      ESL_HIP_TRY(hipGetLastError());
    } else {
      ESL_HIP_TRY(hipMemsetAsync(c->g.bb_valid, 1, (size_t)c->g.n_bbox, c->stream));
    }
  }
  c->lm.begun = true;
  return ESL_OK;
}

int esl_lm_begin(esl_ctx* c, const esl_lm_params* p, int32_t* n_valid, int32_t* n_dropped) {
  if (!c || !p) return ESL_ERR_INVALID;
  // the replicated-graph promise is checked on the step API as well (once per uploaded graph; the same point of every rank's sequence)
  if (c->graph_loaded) { const int rcr = comm_check_replicated(c); if (rcr) return rcr; }
  int rc = lm_begin_enqueue(c, p);
  if (rc) return rc;
  int dropped = 0;
  if (c->g.n_bbox && (p->drop_nan_bbox || c->g.check_vis)) {
    ESL_HIP_TRY(hipMemcpyAsync(&dropped, c->chol_info + 2, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    // leave the counter at zero: a device-driven mapping run on this context counts into it without clearing it first
    ESL_HIP_TRY(hipMemsetAsync(c->chol_info + 2, 0, sizeof(int), c->stream));
    ESL_HIP_TRY(hipStreamSynchronize(c->stream));
  }
  if (n_valid) *n_valid = c->g.n_bbox_edges - dropped;
  if (n_dropped) *n_dropped = dropped;
  return ESL_OK;
}

int esl_lm_linearize(esl_ctx* c, esl_lm_partials* out) {
  if (!c || !out) return ESL_ERR_INVALID;
  if (!c->lm.begun) { set_error("esl_lm_linearize before esl_lm_begin"); return ESL_ERR_STATE; }
  ESL_HIP_TRY(hipSetDevice(c->device));
  if (c->lm.slam) {
    int rc = slam_linearize(c);
    if (rc) return rc;
    double v[4];
    rc = (c->comm && !c->comm_replicated) ? comm_reduce4(c, c->dev_part, v) : read_parts(c, v);
    if (rc) return rc;
    out->chi2 = v[0]; out->max_diag = v[1]; out->scale = 0; out->solve_ok = 1; out->pad = 0;
    return ESL_OK;
  }
  int rc = map_launch_linearize(c, true);
  if (rc) return rc;
  ESL_HIP_TRY(hipStreamSynchronize(c->stream));
  const LmScalars* h = (const LmScalars*)c->host_scal;
  out->chi2 = h->chi2_lin; out->max_diag = h->max_diag; out->scale = 0; out->solve_ok = 1; out->pad = 0;
  return ESL_OK;
}

int esl_lm_try_step(esl_ctx* c, double lambda, esl_lm_partials* out) {
  if (!c || !out) return ESL_ERR_INVALID;
  if (!c->lm.begun) { set_error("esl_lm_try_step before esl_lm_begin"); return ESL_ERR_STATE; }
  ESL_HIP_TRY(hipSetDevice(c->device));
  if (c->lm.slam) {
    int rc = slam_try_step(c, lambda);
    if (rc) return rc;
    double v[4];
    rc = (c->comm && !c->comm_replicated) ? comm_reduce4(c, c->dev_part, v) : read_parts(c, v);
    if (rc) return rc;
    out->chi2 = v[0]; out->max_diag = 0; out->scale = v[2]; out->solve_ok = (c->g.n_objs == 0 || v[3] > 0.5) ? 1 : 0; out->pad = 0;
  } else {
    int rc = map_launch_try(c, lambda);
    if (rc) return rc;
    ESL_HIP_TRY(hipStreamSynchronize(c->stream));
    const LmScalars* h = (const LmScalars*)c->host_scal;
    out->chi2 = h->chi2_trial; out->max_diag = 0; out->scale = h->scale; out->solve_ok = h->ok > 0.5 ? 1 : 0; out->pad = 0;
  }
  c->lm.lambda_used = lambda;
  c->lm.have_trial = true;
  return ESL_OK;
}

int esl_lm_commit(esl_ctx* c, int accept) {
  if (!c) return ESL_ERR_INVALID;
  if (!c->lm.begun || !c->lm.have_trial) { set_error("esl_lm_commit without a trial step"); return ESL_ERR_STATE; }
  if (accept) {  // discardTop: the trial states become the estimate
    std::swap(c->objs, c->objs_trial);
    if (c->lm.slam) { std::swap(c->cams, c->cams_trial); c->cams_match_snap = false; }
  }
  c->lm.have_trial = false;
  return ESL_OK;
}

int esl_lm_reduced_system(esl_ctx* c, double lambda, void** dev_ptr, int64_t* n, int64_t* lda) {
  if (!c || !dev_ptr || !n || !lda) return ESL_ERR_INVALID;
  if (!c->lm.begun) return ESL_ERR_STATE;
  ESL_HIP_TRY(hipSetDevice(c->device));
  if (!c->lm.slam) { *dev_ptr = nullptr; *n = 0; *lda = 0; return ESL_OK; }
  *lda = c->S_lda;
  int rc = slam_build_reduced(c, lambda, true, dev_ptr, n);
  if (rc) return rc;
  ESL_HIP_TRY(hipStreamSynchronize(c->stream));
  return ESL_OK;
}

// Mapping mode on one GPU: the whole LM run is enqueued ahead of the device.  One trial = two launches:
//   k_lm_step              decide the previous trial (accept / reject, lambda, stop test -- on the device), then solve
//                          per ellipsoid for the new lambda -> next trial state
//   k_chunk_linearize_*    linearisation AT the trial state: its chi2 sum is the trial's chi2 (no separate residual
//                          pass) and, if the trial gets accepted, its H, b are already the next iteration's system
// The host keeps a few trials queued, watches the progress counter in mapped memory and stops enqueueing when the
// device reports `done`; launches already queued behind a finished run exit at their first instruction.
static int optimize_mapping_device(esl_ctx* c, const esl_lm_params* p, esl_lm_report* out) {
  int rc;
  const DevGraph& g = c->g;
  LmHostView* hv = (LmHostView*)c->lm_host;
  LmCore* core = (LmCore*)c->lm_dev;   // [2]
  const bool sharded = c->comm != nullptr && !c->comm_replicated;   // collective run: every rank issues the same sequence of all-gathers
  if (!sharded && (g.n_objs == 0 || (g.n_bbox == 0 && g.n_e3d == 0 && c->n_grav_edges == 0))) {   // empty graph
    out->stop_reason = 3;
    return ESL_OK;
  }
  hv->trace_len = 0;
  __atomic_store_n(&hv->seq, 0, __ATOMIC_RELAXED);
  __atomic_store_n(&hv->done, 0, __ATOMIC_RELEASE);
  // linearisation of the start state + chi2, max diag, LM state initialisation (computeLambdaInit) into core[0]
  c->prof_gate = false;
  rc = map_launch_linearize(c, false, nullptr, nullptr, nullptr, g.n_bbox > 0 && (p->drop_nan_bbox != 0 || g.check_vis != 0));
  c->prof_gate = true;
  if (rc) return rc;
  {
    ProfScope ps(c, 4);
    hipLaunchKernelGGL(k_chunk_finalize_rows, dim3(std::max(1, (g.n_objs + kStepWaves - 1) / kStepWaves)), dim3(64 * kStepWaves), 0, c->stream, g, chunk_table(c), c->chunk_out,
                       c->objs, c->lm.p.jacobian_mode, c->lm.p.numeric_delta, c->blk_part, c->tickets, c->dev_scal,
                       (LmScalars*)c->host_scal_dev, p->tau, sharded ? (LmCore*)nullptr : core, c->chol_info + 2, c->n_grav_edges,
                       (LmHostView*)c->lm_host_dev);
    ESL_HIP_TRY(hipGetLastError());
  }
  if (sharded && (rc = comm_gather_scalars_device(c))) return rc;   // collective: every rank's {chi2, max diag, has_edges}
  if (p->max_iters <= 0) {   // nothing to iterate: report the start state
    ESL_HIP_TRY(hipStreamSynchronize(c->stream));
    const LmScalars* h = (const LmScalars*)c->host_scal;
    out->chi2_initial = out->chi2_final = h->chi2_lin;
    out->n_bbox_dropped = hv->n_dropped;
    out->n_bbox_valid = g.n_bbox_edges - hv->n_dropped;
    if (hv->done) out->stop_reason = hv->core.stop_reason;
    return ESL_OK;
  }
  const bool sample_run = c->prof_on && c->prof_level == 1 && (c->prof_runs++ % 4 == 0);
  const int max_total = p->max_iters * std::max(1, p->max_trials);
  const int depth = 2;   // trials kept in flight ahead of the device's progress counter
  const int n_step_blocks = std::max(1, (g.n_objs + kStepWaves - 1) / kStepWaves);
  const int batch = 4;   // sharded: trials enqueued per round (a fixed number, so that all ranks issue the same collectives)
  const int n_lin_blocks = (c->n_ids_e3 + 2 * kLinWaves - 1) / (2 * kLinWaves) + (c->n_ids_bb + kLinWaves - 1) / kLinWaves;
  // launch k carries trial k's solve and trial k-1's decision: one launch more than there are trials
  int enq = 0, seen = 0;
  bool done = false;
  while (!done) {
    const int enq_limit = sharded ? enq + batch : max_total + 1;
    if (sharded && enq > max_total + batch) { set_error("device-side LM did not terminate"); return ESL_ERR_STATE; }
    while (enq < enq_limit && (sharded || enq - seen < depth)) {
      const LmCore* in = core + (enq & 1);
      LmCore* nxt = core + ((enq + 1) & 1);
      {
        ProfScope ps(c, 1);
        hipLaunchKernelGGL(k_lm_step_rows, dim3(n_step_blocks), dim3(64 * kStepWaves), 0, c->stream, g, chunk_table(c), c->chunk_out,
                           c->chunk_out2, c->objs, c->objs_trial, in, nxt, c->blk_chi, n_lin_blocks, c->solve_part + 4 * n_step_blocks * (enq & 1),
                           c->solve_part + 4 * n_step_blocks * ((enq + 1) & 1), enq == 0 ? (sharded ? 2 : 1) : 0, p->max_iters, p->max_trials,
                           (LmHostView*)c->lm_host_dev, c->lm.p.jacobian_mode, c->lm.p.numeric_delta, c->xo,
                           sharded ? c->dev_gather : (const double*)nullptr, sharded ? c->comm_ranks : 0, p->tau);
        ESL_HIP_TRY(hipGetLastError());
      }
      // the second trial's linearisation (always a live launch when it exists) of every FOURTH run is the sampled one: an event
      // pair splits two back-to-back dispatches and costs that trial ~20 us -- one per run was 10 % of a 0.2 ms run
      c->prof_gate = (enq == 1) && (sample_run);
      rc = map_launch_linearize(c, false, nullptr, nullptr, nxt);
      c->prof_gate = true;
      if (rc) return rc;
      if (sharded) {   // this rank's share of the trial's scalars, then everybody's (one small all-gather on the stream)
        hipLaunchKernelGGL(k_lm_partials, dim3(1), dim3(256), 0, c->stream, c->blk_chi, n_lin_blocks,
                           c->solve_part + 4 * n_step_blocks * ((enq + 1) & 1), n_step_blocks, c->dev_scal);
        ESL_HIP_TRY(hipGetLastError());
        ProfScope ps(c, 6);
        if ((rc = comm_gather_scalars_device(c))) return rc;
      }
      ++enq;
    }
    // wait for the device to decide at least one more trial (sharded: all but the last of the trials enqueued so far --
    // a condition every rank evaluates on the same sequence of decisions): spin on the mapped counter, fall back to a
    // stream sync
    const int want = sharded ? enq - 1 : seen + 1;
    int s = seen;
    for (int spin = 0; spin < 400000; ++spin) {
      s = __atomic_load_n(&hv->seq, __ATOMIC_ACQUIRE);
      if (s >= want || __atomic_load_n(&hv->done, __ATOMIC_ACQUIRE)) break;
    }
    if (s < want && !__atomic_load_n(&hv->done, __ATOMIC_ACQUIRE)) {
      ESL_HIP_TRY(hipStreamSynchronize(c->stream));
      s = __atomic_load_n(&hv->seq, __ATOMIC_ACQUIRE);
      if (s < want && !hv->done) { set_error("device-side LM made no progress"); return ESL_ERR_STATE; }
    }
    seen = s;
    done = __atomic_load_n(&hv->done, __ATOMIC_ACQUIRE) != 0;
  }
  // `done` was stored with release semantics after the results: they are visible without draining the stream (the
  // one or two no-op launches still queued touch nothing the caller can see)
  if (sharded) ESL_HIP_TRY(hipStreamSynchronize(c->stream));   // leave no collective in flight behind the caller's back
  if (std::getenv("ESL_LM_TIMING")) {
    (void)hipStreamSynchronize(c->stream);
    fprintf(stderr, "[k_lm_step, workgroup 0, last live launch, us] decision=%.2f gather=%.2f solve=%.2f\n",
            (double)(hv->dbg_clk[1] - hv->dbg_clk[0]) * 0.01, (double)(hv->dbg_clk[2] - hv->dbg_clk[1]) * 0.01,
            (double)(hv->dbg_clk[3] - hv->dbg_clk[2]) * 0.01);
  }
  const LmCore& r = hv->core;
  out->n_bbox_dropped = hv->n_dropped;
  out->n_bbox_valid = g.n_bbox_edges - hv->n_dropped;
  if (r.cur) {   // the current estimate (and its system, for the inspection API) live in the second pair
    std::swap(c->objs, c->objs_trial);
    std::swap(c->chunk_out, c->chunk_out2);
  }
  c->sys_combined = false;
  c->lm.have_trial = false;
  out->iterations = r.it;
  out->total_trials = r.total_trials;
  out->stop_reason = r.stop_reason;
  out->chi2_initial = r.chi2_initial;
  out->chi2_final = r.currentChi;
  out->lambda_final = r.lambda;
  out->trace_len = hv->trace_len;
  for (int k = 0; k < hv->trace_len && k < ESL_MAX_TRACE; ++k) {
    out->trace_chi2[k] = hv->trace_chi2[k]; out->trace_lambda[k] = hv->trace_lambda[k]; out->trace_trials[k] = hv->trace_trials[k];
  }
  return ESL_OK;
}

int esl_optimize_resident(esl_ctx* c, const esl_lm_params* p, esl_lm_report* out) {
  if (!c || !p || !out) return ESL_ERR_INVALID;
  if (p->linear_solver != ESL_SOLVER_AUTO && p->linear_solver != ESL_SOLVER_REDUCED_CAMERA && p->linear_solver != ESL_SOLVER_REDUCED_ELLIPSOID) {
    set_error("esl_lm_params::linear_solver: unknown solver"); return ESL_ERR_INVALID;
  }
  if (p->bbox_residual != ESL_BBOX_REPROJECTION && p->bbox_residual != ESL_BBOX_TANGENCY) { set_error("esl_lm_params::bbox_residual: unknown mode"); return ESL_ERR_INVALID; }
  if (p->e3d_half_turn != 0 && p->e3d_half_turn != 1) { set_error("esl_lm_params::e3d_half_turn: 0 (half turns not eligible) or 1 (as the reference writes it)"); return ESL_ERR_INVALID; }
  std::memset(out, 0, sizeof(*out));
  if (c->graph_loaded) { const int rcr = comm_check_replicated(c); if (rcr) return rcr; }
  if (c->graph_loaded && c->g.n_free_cams == 0) {   // mapping mode: the LM runs on the device, nothing waits on the host
    int rc0 = lm_begin_enqueue(c, p, true);
    if (rc0) return rc0;
    return optimize_mapping_device(c, p, out);
  }
  int32_t nv = 0, nd = 0;
  int rc = esl_lm_begin(c, p, &nv, &nd);
  if (rc) return rc;
  out->n_bbox_valid = nv;
  out->n_bbox_dropped = nd;
  const DevGraph& g = c->g;
  const bool any_edge = (nv > 0) || g.n_e3d > 0 || (g.n_odom > 0 && g.n_free_cams > 0);
  bool any_grav = false;
  if (!any_edge) {  // gravity edges alone also make a graph
    std::vector<int> cnt((size_t)std::max(g.n_objs, 1));
    if (g.n_objs) ESL_HIP_TRY(hipMemcpy(cnt.data(), g.gr_cnt, (size_t)g.n_objs * sizeof(int), hipMemcpyDeviceToHost));
    for (int o = 0; o < g.n_objs; ++o) any_grav = any_grav || cnt[o] > 0;
  }
  if (!any_edge && !any_grav) { out->stop_reason = 3; return ESL_OK; }

  // SLAM mode (free cameras): host-driven LM over the step API -- each trial's dense factorisation dwarfs the host
  // round trip (optimization_algorithm_levenberg.cpp:69-141, statement for statement)
  double lambda = -1, ni = 2;
  int nBad = 0, it = 0, total_trials = 0;
  bool ok_outer = true;
  double currentChi = 0;
  for (it = 0; it < p->max_iters && ok_outer; ++it) {
    esl_lm_partials lin;
    if ((rc = esl_lm_linearize(c, &lin))) return rc;
    currentChi = lin.chi2;
    const double iniChi = currentChi;
    if (it == 0) {
      out->chi2_initial = currentChi;
      lambda = p->tau * lin.max_diag;  // computeLambdaInit
      ni = 2;
      nBad = 0;
    }
    double rho = 0;
    int qmax = 0;
    do {
      esl_lm_partials tr;
      if ((rc = esl_lm_try_step(c, lambda, &tr))) return rc;
      double tempChi = tr.solve_ok ? tr.chi2 : DBL_MAX;
      rho = (currentChi - tempChi) / (tr.scale + 1e-3);
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - std::pow((2 * rho - 1), 3);
        alpha = std::min(alpha, 2. / 3.);
        lambda *= std::max(1. / 3., alpha);
        ni = 2;
        currentChi = tempChi;
        if ((rc = esl_lm_commit(c, 1))) return rc;
      } else {
        lambda *= ni;
        ni *= 2;
        if ((rc = esl_lm_commit(c, 0))) return rc;
      }
      qmax++;
    } while (rho < 0 && qmax < p->max_trials);
    total_trials += qmax;
    if (it < ESL_MAX_TRACE) {
      out->trace_chi2[it] = currentChi; out->trace_lambda[it] = lambda; out->trace_trials[it] = qmax;
      out->trace_len = it + 1;
    }
    if (qmax == p->max_trials || rho == 0) { ok_outer = false; out->stop_reason = 1; }
    else {
      if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
      if (nBad >= 3) { ok_outer = false; out->stop_reason = 2; }
    }
  }
  out->iterations = it;
  out->total_trials = total_trials;
  out->chi2_final = currentChi;
  out->lambda_final = lambda;
  return ESL_OK;
}

int esl_optimize(esl_ctx* c, const esl_graph* g, double* cams_io, double* objs_io, const esl_lm_params* p,
                 esl_lm_report* out) {
  if (!c || !g || !p || !out) return ESL_ERR_INVALID;
  int rc;
  if ((rc = esl_graph_upload(c, g))) return rc;
  if ((rc = esl_states_upload(c, cams_io, objs_io))) return rc;
  if ((rc = esl_optimize_resident(c, p, out))) return rc;
  return esl_states_download(c, cams_io, objs_io);
}

int esl_states_snapshot(esl_ctx* c) {
  if (!c || !c->states_loaded) { set_error("esl_states_snapshot: no states"); return ESL_ERR_STATE; }
  ESL_HIP_TRY(hipSetDevice(c->device));
  const size_t nc = (size_t)c->g.n_cams * 7, no = (size_t)c->g.n_objs * 10;
  if (!c->cams_snap) ESL_HIP_TRY(hipMalloc((void**)&c->cams_snap, std::max<size_t>(nc, 1) * sizeof(double)));
  if (!c->objs_snap) ESL_HIP_TRY(hipMalloc((void**)&c->objs_snap, std::max<size_t>(no, 1) * sizeof(double)));
  if (nc) ESL_HIP_TRY(hipMemcpyAsync(c->cams_snap, c->cams, nc * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  if (no) ESL_HIP_TRY(hipMemcpyAsync(c->objs_snap, c->objs, no * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  c->cams_match_snap = true;
  return ESL_OK;
}

int esl_states_restore(esl_ctx* c) {
  if (!c || !c->cams_snap || !c->objs_snap) { set_error("esl_states_restore: no snapshot"); return ESL_ERR_STATE; }
  ESL_HIP_TRY(hipSetDevice(c->device));
  const size_t nc = (size_t)c->g.n_cams * 7, no = (size_t)c->g.n_objs * 10;
  if (nc && !c->cams_match_snap) ESL_HIP_TRY(hipMemcpyAsync(c->cams, c->cams_snap, nc * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  c->cams_match_snap = true;
  if (no) ESL_HIP_TRY(hipMemcpyAsync(c->objs, c->objs_snap, no * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  c->lm.begun = false;
  return ESL_OK;
}

int esl_profile_enable(esl_ctx* c, int enable) {
  if (!c) return ESL_ERR_INVALID;
  ESL_HIP_TRY(hipSetDevice(c->device));
  if (c->prof_on) prof_drain(c);
  c->prof_on = enable != 0;
  c->prof_level = enable;
  c->prof_runs = 0;
  for (int k = 0; k < ESL_PROF_KINDS; ++k) { c->prof_count[k] = 0; c->prof_ms[k] = 0; }
  return ESL_OK;
}

int esl_profile_get(esl_ctx* c, int64_t count[ESL_PROF_KINDS], double total_ms[ESL_PROF_KINDS]) {
  if (!c || !count || !total_ms) return ESL_ERR_INVALID;
  ESL_HIP_TRY(hipSetDevice(c->device));
  int rc = prof_drain(c);
  if (rc) return rc;
  for (int k = 0; k < ESL_PROF_KINDS; ++k) { count[k] = c->prof_count[k]; total_ms[k] = c->prof_ms[k]; }
  return ESL_OK;
}

int esl_lm_download(esl_ctx* c, int32_t which, double* dst, int64_t count) {
  if (!c || !dst || count < 0) return ESL_ERR_INVALID;
  if (!c->graph_loaded) { set_error("esl_lm_download: no graph"); return ESL_ERR_STATE; }
  const DevGraph& g = c->g;
  const double* src = nullptr;
  int64_t n = 0;
  switch (which) {
    case 0: src = c->Hoo; n = (int64_t)g.n_objs * 45; break;
    case 1: src = c->bo; n = (int64_t)g.n_objs * 9; break;
    case 2: src = c->xo; n = (int64_t)g.n_objs * 9; break;
    case 3: src = c->Hcc; n = (int64_t)g.n_free_cams * 36; break;
    case 4: src = c->bc; n = (int64_t)g.n_free_cams * 6; break;
    case 5: src = c->xc; n = (int64_t)g.n_free_cams * 6; break;
    case 6: src = c->S; n = c->S_lda * c->S_n; break;
    case 7: src = c->objs_trial; n = (int64_t)g.n_objs * 10; break;
    case 8: src = c->cams_trial; n = (int64_t)g.n_cams * 7; break;
    case 9: src = c->lm.slam ? c->Wbb : nullptr; n = ((int64_t)g.n_bbox + g.n_e3d) * 54; break;
    default: set_error("esl_lm_download: unknown array"); return ESL_ERR_INVALID;
  }
  if (!src || count < n) { set_error("esl_lm_download: array not available or buffer too small"); return ESL_ERR_INVALID; }
  ESL_HIP_TRY(hipSetDevice(c->device));
  if ((which == 0 || which == 1) && !c->lm.slam && c->lm.begun) { int rc = map_combine(c); if (rc) return rc; }
  if (n) ESL_HIP_TRY(hipMemcpyAsync(dst, src, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  ESL_HIP_TRY(hipStreamSynchronize(c->stream));
  return ESL_OK;
}

// host-only: balanced partition of ellipsoids by edge count (greedy longest-processing-time)
int esl_partition_objects(const esl_graph* g, int32_t n_parts, int32_t* part_of_obj) {
  if (!g || n_parts <= 0 || (!part_of_obj && g->n_objs > 0)) return ESL_ERR_INVALID;
  int rc = validate_graph(g);
  if (rc) return rc;
  const int N = g->n_objs;
  std::vector<int64_t> load((size_t)N, 1);
  for (int i = 0; i < g->n_bbox; ++i) load[g->bbox_obj[i]] += 4;
  for (int i = 0; i < g->n_e3d; ++i) load[g->e3d_obj[i]] += 9;
  std::vector<int> order((size_t)N);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return load[a] > load[b]; });
  std::vector<int64_t> tot((size_t)n_parts, 0);
  for (int k = 0; k < N; ++k) {
    int best = 0;
    for (int q = 1; q < n_parts; ++q)
      if (tot[q] < tot[best]) best = q;
    part_of_obj[order[k]] = best;
    tot[best] += load[order[k]];
  }
  return ESL_OK;
}

}  // extern "C"

#ifdef ESL_ISA_PROBE
// scripts/isa_count.py compiles this TU with -DESL_ISA_PROBE to get the per-edge-type instruction streams of the
// fused linearisation kernel as separate symbols (analysis only; never part of libesl_hip.so).
namespace esl {
template __global__ void k_chunk_linearize<ESL_JAC_ANALYTIC, 0, false>(DevGraph, ChunkTable, const int*, int, const double*, const double*,
                                                                       const double*, double, double*, double*, double*, int,
                                                                       const LmCore*, int*);
template __global__ void k_chunk_linearize<ESL_JAC_ANALYTIC, 1, false>(DevGraph, ChunkTable, const int*, int, const double*, const double*,
                                                                       const double*, double, double*, double*, double*, int,
                                                                       const LmCore*, int*);
}  // namespace esl
#endif
