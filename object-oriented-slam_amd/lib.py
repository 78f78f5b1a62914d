"""ctypes binding of csrc/libesl_hip.so (the C-ABI of include/esl.h).

The library is the product: if it is missing or no HIP device is usable, everything here raises —
there is no CPU fallback (the CPU restatement under oracle/ is test infrastructure and is never
imported from this package).
"""
import ctypes as C
import os
import subprocess
import time

import numpy as np

from . import abi

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# ESL_HIP_LIB: load another build of the SAME library (kernel tuning variants side by side); never a fallback
LIB_PATH = os.environ.get("ESL_HIP_LIB") or os.path.join(_CSRC, "libesl_hip.so")
_lib = None

EXPORTS = [
    "esl_abi_version", "esl_last_error", "esl_device_count", "esl_ctx_create", "esl_ctx_destroy",
    "esl_ctx_synchronize", "esl_ctx_trim", "esl_lm_params_default", "esl_optimize", "esl_graph_upload", "esl_graph_append", "esl_graph_sizes", "esl_states_upload",
    "esl_states_download", "esl_optimize_resident", "esl_states_snapshot", "esl_states_restore", "esl_profile_enable", "esl_profile_get", "esl_lm_begin", "esl_lm_linearize", "esl_lm_reduced_system", "esl_lm_reduced_residual",
    "esl_lm_try_step", "esl_lm_commit", "esl_lm_solver_used", "esl_lm_solver_stats", "esl_lm_download", "esl_comm_unique_id", "esl_comm_init", "esl_comm_init_host", "esl_comm_set_replicated", "esl_comm_destroy", "esl_partition_objects", "esl_fit_params_default", "esl_fit_frame", "esl_fit_frame_debug", "esl_fit_frame_ex", "esl_selftest_cholesky", "esl_debug_chol_plan",
    "esl_init_quadric", "esl_init_from_qstar", "esl_init_plane_error", "esl_plane_params_default", "esl_extract_ground_plane", "esl_extract_planes",
]


class EslError(RuntimeError):
    pass


def build(force=False):
    """Compile every HIP translation unit for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", _CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-j4", "-C", _CSRC])
    return LIB_PATH


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EslError(f"{LIB_PATH} is missing: run __graft_entry__.build() (no CPU fallback exists)")
        L = C.CDLL(LIB_PATH)
        L.esl_last_error.restype = C.c_char_p
        for name in EXPORTS:
            if not hasattr(L, name):
                raise EslError(f"libesl_hip.so does not export {name}")
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        msg = load().esl_last_error().decode(errors="replace")
        raise EslError(f"{what} failed with esl_status {rc}: {msg}")


def device_count():
    return int(load().esl_device_count())


_dp = C.POINTER(C.c_double)


class Context:
    """One HIP device + stream + device-resident graph/states (esl_ctx)."""

    def __init__(self, device=0):
        L = load()
        self._h = C.c_void_p()
        self.device = int(device)
        _check(L.esl_ctx_create(C.c_int(device), C.byref(self._h)), "esl_ctx_create")
        self._graph = None

    def close(self):
        if self._h:
            load().esl_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- one shot -----------------------------------------------------------------------------
    def optimize(self, graph, cams, objs, params=None):
        p = params if params is not None else abi.default_lm_params()
        g = graph.c_struct()
        cams = np.array(cams, dtype=np.float64, order="C").reshape(-1, 7).copy()
        objs = np.array(objs, dtype=np.float64, order="C").reshape(-1, 10).copy()
        rep = abi.EslLmReport()
        _check(load().esl_optimize(self._h, C.byref(g), cams.ctypes.data_as(_dp), objs.ctypes.data_as(_dp),
                                   C.byref(p), C.byref(rep)), "esl_optimize")
        return cams, objs, rep.as_dict()

    # ---- resident / step API --------------------------------------------------------------------
    def upload_graph(self, graph):
        self._graph = graph
        g = graph.c_struct()
        _check(load().esl_graph_upload(self._h, C.byref(g)), "esl_graph_upload")

    def append_graph(self, new_cams=(), new_objs=(), bbox=None, e3d=None, grav_obj=(), new_cam_fixed=None, odom=None):
        """esl_graph_append: bbox = (cam, obj, meas (n,4), weight), e3d = (cam, obj, meas (n,10), weight); indices in the extended
        numbering (new cameras / ellipsoids follow the existing ones).  SLAM mode: new_cam_fixed = flags of the new cameras (None:
        fixed), odom = (i, j, meas (n,7)[, info (n,6)])."""
        keep = []

        def arr(a, dt, tail=None):
            a = np.ascontiguousarray(a, dtype=dt)
            if tail is not None and a.size:
                a = a.reshape((-1,) + tail)
            keep.append(a)
            return a
        d = abi.EslGraphDelta()
        nc, no = arr(new_cams, np.float64, (7,)), arr(new_objs, np.float64, (10,))
        d.n_new_cams, d.n_new_objs = (len(nc) if nc.size else 0), (len(no) if no.size else 0)
        d.new_cams = nc.ctypes.data_as(_dp) if nc.size else _dp(); d.new_objs = no.ctypes.data_as(_dp) if no.size else _dp()
        ip = C.POINTER(C.c_int32)
        if bbox is not None and len(bbox[0]):
            bc, bo, bm, bw = arr(bbox[0], np.int32), arr(bbox[1], np.int32), arr(bbox[2], np.float64, (4,)), arr(bbox[3], np.float64)
            d.n_bbox = len(bc); d.bbox_cam = bc.ctypes.data_as(ip); d.bbox_obj = bo.ctypes.data_as(ip)
            d.bbox_meas = bm.ctypes.data_as(_dp); d.bbox_weight = bw.ctypes.data_as(_dp)
        if e3d is not None and len(e3d[0]):
            ec, eo, em, ew = arr(e3d[0], np.int32), arr(e3d[1], np.int32), arr(e3d[2], np.float64, (10,)), arr(e3d[3], np.float64)
            d.n_e3d = len(ec); d.e3d_cam = ec.ctypes.data_as(ip); d.e3d_obj = eo.ctypes.data_as(ip)
            d.e3d_meas = em.ctypes.data_as(_dp); d.e3d_weight = ew.ctypes.data_as(_dp)
        go = arr(grav_obj, np.int32)
        if go.size:
            d.n_grav = len(go); d.grav_obj = go.ctypes.data_as(ip)
        if new_cam_fixed is not None and d.n_new_cams:
            fx = arr(new_cam_fixed, np.uint8)
            assert fx.size == d.n_new_cams
            d.new_cam_fixed = fx.ctypes.data_as(C.POINTER(C.c_uint8))
        if odom is not None and len(odom[0]):
            oi, oj, om = arr(odom[0], np.int32), arr(odom[1], np.int32), arr(odom[2], np.float64, (7,))
            d.n_odom = len(oi); d.odom_i = oi.ctypes.data_as(ip); d.odom_j = oj.ctypes.data_as(ip); d.odom_meas = om.ctypes.data_as(_dp)
            if len(odom) > 3 and odom[3] is not None:
                d.odom_info = arr(odom[3], np.float64, (6,)).ctypes.data_as(_dp)
        _check(load().esl_graph_append(self._h, C.byref(d)), "esl_graph_append")
        if self._graph is not None:
            self._graph = _Sizes(self._graph.n_cams + d.n_new_cams, self._graph.n_objs + d.n_new_objs)

    def graph_sizes(self):
        v = [C.c_int32(0) for _ in range(5)]
        _check(load().esl_graph_sizes(self._h, *[C.byref(x) for x in v]), "esl_graph_sizes")
        return dict(zip(("n_cams", "n_objs", "n_bbox", "n_e3d", "relayouts"), [x.value for x in v]))

    def upload_states(self, cams, objs):
        cams = np.ascontiguousarray(cams, dtype=np.float64)
        objs = np.ascontiguousarray(objs, dtype=np.float64)
        _check(load().esl_states_upload(self._h, cams.ctypes.data_as(_dp), objs.ctypes.data_as(_dp)), "esl_states_upload")

    def download_states(self):
        cams = np.zeros((self._graph.n_cams, 7))
        objs = np.zeros((self._graph.n_objs, 10))
        _check(load().esl_states_download(self._h, cams.ctypes.data_as(_dp), objs.ctypes.data_as(_dp)), "esl_states_download")
        return cams, objs

    def optimize_resident(self, params=None):
        p = params if params is not None else abi.default_lm_params()
        rep = abi.EslLmReport()
        _check(load().esl_optimize_resident(self._h, C.byref(p), C.byref(rep)), "esl_optimize_resident")
        return rep.as_dict()

    def lm_begin(self, params=None):
        p = params if params is not None else abi.default_lm_params()
        nv, nd = C.c_int32(0), C.c_int32(0)
        _check(load().esl_lm_begin(self._h, C.byref(p), C.byref(nv), C.byref(nd)), "esl_lm_begin")
        return nv.value, nd.value

    def lm_linearize(self):
        out = abi.EslLmPartials()
        _check(load().esl_lm_linearize(self._h, C.byref(out)), "esl_lm_linearize")
        return out

    def lm_try_step(self, lam):
        out = abi.EslLmPartials()
        _check(load().esl_lm_try_step(self._h, C.c_double(lam), C.byref(out)), "esl_lm_try_step")
        return out

    def lm_commit(self, accept):
        _check(load().esl_lm_commit(self._h, C.c_int(1 if accept else 0)), "esl_lm_commit")

    def lm_solver_used(self):
        """esl_linear_solver of the last SLAM-mode trial step: 1 reduced camera system, 2 reduced ellipsoid system (cameras first)."""
        v = C.c_int32(0)
        _check(load().esl_lm_solver_used(self._h, C.byref(v)), "esl_lm_solver_used")
        return v.value

    def lm_solver_stats(self):
        """Shape of the camera-first elimination as the last trial ran it (esl_lm_solver_stats)."""
        st = (C.c_double * 10)()   # ESL_SOLVER_STATS
        _check(load().esl_lm_solver_stats(self._h, st), "esl_lm_solver_stats")
        names = ["x_form", "stride", "separators", "segments", "product_flops", "product_bytes", "slab_bytes", "dense_update_rows",
                 "dense_update_flops_executed", "t_ordered_by_first_camera"]
        return {n: float(st[i]) for i, n in enumerate(names)}

    def lm_reduced_system(self, lam):
        ptr, n, lda = C.c_void_p(), C.c_int64(0), C.c_int64(0)
        _check(load().esl_lm_reduced_system(self._h, C.c_double(lam), C.byref(ptr), C.byref(n), C.byref(lda)),
               "esl_lm_reduced_system")
        return ptr.value, n.value, lda.value

    def lm_reduced_residual(self):
        """|S x_c - b_s| / |b_s| of the last SLAM-mode trial step (the reduced system is re-built for it)."""
        r = C.c_double(0)
        _check(load().esl_lm_reduced_residual(self._h, C.byref(r)), "esl_lm_reduced_residual")
        return r.value

    def lm_download(self, which, count):
        out = np.zeros(int(count))
        _check(load().esl_lm_download(self._h, C.c_int32(which), out.ctypes.data_as(_dp), C.c_int64(int(count))),
               "esl_lm_download")
        return out

    def snapshot_states(self):
        _check(load().esl_states_snapshot(self._h), "esl_states_snapshot")

    def restore_states(self):
        _check(load().esl_states_restore(self._h), "esl_states_restore")

    def profile_enable(self, level=2):
        """0/False off; 1 only the linearise kernels (cheap, for timed regions); 2/True every kernel class."""
        lv = 2 if level is True else int(level)
        _check(load().esl_profile_enable(self._h, C.c_int(lv)), "esl_profile_enable")

    def profile_get(self):
        cnt = (C.c_int64 * 10)()   # ESL_PROF_KINDS
        ms = (C.c_double * 10)()
        _check(load().esl_profile_get(self._h, cnt, ms), "esl_profile_get")
        names = ["linearize", "lm_trial", "schur_build", "cholesky_solve", "reduce", "k5", "shard_allreduce", "rank_k_update", "sparse_block_products",
                 "dense_factorisation"]
        return {n: dict(count=int(cnt[i]), total_ms=float(ms[i])) for i, n in enumerate(names) if cnt[i]}

    def init_quadric(self, poses_Twc, bboxes, K, rows=480, cols=640, faithful=1):
        """Initializer::initializeQuadric: returns (ellipsoid 10-vector, Q* 4x4, ok)."""
        poses = np.ascontiguousarray(poses_Twc, dtype=np.float64).reshape(-1, 7)
        boxes = np.ascontiguousarray(bboxes, dtype=np.float64).reshape(-1, 4)
        Kd = np.ascontiguousarray(K, dtype=np.float64)
        e = np.zeros(10); Q = np.zeros(16); ok = C.c_int32(0)
        _check(load().esl_init_quadric(self._h, poses.ctypes.data_as(_dp), boxes.ctypes.data_as(_dp), C.c_int32(len(poses)),
                                       Kd.ctypes.data_as(_dp), C.c_int32(rows), C.c_int32(cols), C.c_int32(faithful),
                                       e.ctypes.data_as(_dp), Q.ctypes.data_as(_dp), C.byref(ok)), "esl_init_quadric")
        return e, Q.reshape(4, 4), bool(ok.value)

    def init_from_qstar(self, qstar, faithful=1):
        """Initializer::getEllipsoidFromQStar: (ellipsoid 10-vector, ok)."""
        Q = np.ascontiguousarray(qstar, dtype=np.float64).reshape(16)
        e = np.zeros(10); ok = C.c_int32(0)
        _check(load().esl_init_from_qstar(self._h, Q.ctypes.data_as(_dp), C.c_int32(faithful), e.ctypes.data_as(_dp), C.byref(ok)),
               "esl_init_from_qstar")
        return e, bool(ok.value)

    def init_plane_error(self, poses_Twc, bboxes, K, ellipsoid, rows=480, cols=640):
        """Initializer::quadricErrorWithPlanes."""
        poses = np.ascontiguousarray(poses_Twc, dtype=np.float64).reshape(-1, 7)
        boxes = np.ascontiguousarray(bboxes, dtype=np.float64).reshape(-1, 4)
        Kd = np.ascontiguousarray(K, dtype=np.float64); e = np.ascontiguousarray(ellipsoid, dtype=np.float64)
        err = C.c_double(0)
        _check(load().esl_init_plane_error(self._h, poses.ctypes.data_as(_dp), boxes.ctypes.data_as(_dp), C.c_int32(len(poses)),
                                           Kd.ctypes.data_as(_dp), C.c_int32(rows), C.c_int32(cols), e.ctypes.data_as(_dp),
                                           C.byref(err)), "esl_init_plane_error")
        return err.value

    def fit_frame(self, depth, bboxes, labels, Twc, intr, ground, params=None):
        """EllipsoidExtractor::EstimateLocalEllipsoid for every box of one frame.
        Returns (ellipsoids (B,10) in the camera frame, prob (B,), status (B,), debug (B,16))."""
        p = params if params is not None else default_fit_params()
        depth = np.ascontiguousarray(depth, dtype=np.uint16)
        h, w = depth.shape
        boxes = np.ascontiguousarray(bboxes, dtype=np.float64).reshape(-1, 4)
        B = len(boxes)
        lab = np.ascontiguousarray(labels, dtype=np.int32).reshape(-1)
        Twc = np.ascontiguousarray(Twc, dtype=np.float64); intr = np.ascontiguousarray(intr, dtype=np.float64)
        ground = np.ascontiguousarray(ground, dtype=np.float64)
        ell = np.zeros((B, 10)); prob = np.zeros(B); st = np.zeros(B, dtype=np.int32); dbg = np.zeros((B, 16))
        fn = load().esl_fit_frame_debug
        cargs = (self._h, depth.ctypes.data_as(C.POINTER(C.c_uint16)), C.c_int32(w), C.c_int32(h),
                 boxes.ctypes.data_as(_dp), lab.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int32(B),
                 Twc.ctypes.data_as(_dp), intr.ctypes.data_as(_dp), ground.ctypes.data_as(_dp),
                 C.byref(p), ell.ctypes.data_as(_dp), prob.ctypes.data_as(_dp),
                 st.ctypes.data_as(C.POINTER(C.c_int32)), dbg.ctypes.data_as(_dp))
        t0 = time.perf_counter()
        rc = fn(*cargs)
        self.last_call_s = time.perf_counter() - t0      # the C-ABI call alone (bench.py reports it next to the Python-inclusive time)
        _check(rc, "esl_fit_frame")
        return ell, prob, st, dbg

    def fit_frame_ex(self, depth, bboxes, labels, Twc, intr, ground, params=None):
        """fit_frame + SymmetryOutputData per box (EllipsoidExtractor::GetSymmetryOutputData): returns
        (ellipsoids, prob, status, debug, sym) with sym a dict of arrays result (B,), symmetry_type (B,), plane (B,4),
        plane2 (B,4), prob (B,), center (B,3)."""
        p = params if params is not None else default_fit_params()
        depth = np.ascontiguousarray(depth, dtype=np.uint16)
        h, w = depth.shape
        boxes = np.ascontiguousarray(bboxes, dtype=np.float64).reshape(-1, 4)
        B = len(boxes)
        lab = np.ascontiguousarray(labels, dtype=np.int32).reshape(-1)
        Twc = np.ascontiguousarray(Twc, dtype=np.float64); intr = np.ascontiguousarray(intr, dtype=np.float64)
        ground = np.ascontiguousarray(ground, dtype=np.float64)
        ell = np.zeros((B, 10)); prob = np.zeros(B); st = np.zeros(B, dtype=np.int32); dbg = np.zeros((B, 16))
        sym = (abi.EslFitSymmetry * max(B, 1))()
        _check(load().esl_fit_frame_ex(self._h, depth.ctypes.data_as(C.POINTER(C.c_uint16)), C.c_int32(w), C.c_int32(h),
                                       boxes.ctypes.data_as(_dp), lab.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int32(B),
                                       Twc.ctypes.data_as(_dp), intr.ctypes.data_as(_dp), ground.ctypes.data_as(_dp),
                                       C.byref(p), ell.ctypes.data_as(_dp), prob.ctypes.data_as(_dp),
                                       st.ctypes.data_as(C.POINTER(C.c_int32)), sym, dbg.ctypes.data_as(_dp)), "esl_fit_frame_ex")
        out = dict(result=np.array([sym[b].result for b in range(B)], dtype=np.int32),
                   symmetry_type=np.array([sym[b].symmetry_type for b in range(B)], dtype=np.int32),
                   plane=np.array([list(sym[b].plane) for b in range(B)]).reshape(B, 4),
                   plane2=np.array([list(sym[b].plane2) for b in range(B)]).reshape(B, 4),
                   prob=np.array([sym[b].prob for b in range(B)]),
                   center=np.array([list(sym[b].center) for b in range(B)]).reshape(B, 3))
        return ell, prob, st, dbg, out

    def extract_ground_plane(self, depth, intr, params=None):
        """PlaneExtractor::extractGroundPlane: dict(ok, plane (camera frame, 4), n_planes, n_pixels)."""
        p = params if params is not None else abi.default_plane_params()
        depth = np.ascontiguousarray(depth, dtype=np.uint16)
        h, w = depth.shape
        intr = np.ascontiguousarray(intr, dtype=np.float64)
        plane = np.zeros(4); ok = C.c_int32(0); npl = C.c_int32(0); npx = C.c_int32(0)
        _check(load().esl_extract_ground_plane(self._h, depth.ctypes.data_as(C.POINTER(C.c_uint16)), C.c_int32(w), C.c_int32(h),
                                               intr.ctypes.data_as(_dp), C.byref(p), plane.ctypes.data_as(_dp), C.byref(ok),
                                               C.byref(npl), C.byref(npx)), "esl_extract_ground_plane")
        return dict(ok=bool(ok.value), plane=plane, n_planes=npl.value, n_pixels=npx.value)

    def extract_planes(self, depth, intr, params=None, max_planes=64):
        """PlaneExtractor::extractPlanes: dict(n_planes, planes (n, 4), sizes (n,), labels (h, w) plane index or -1)."""
        p = params if params is not None else abi.default_plane_params()
        depth = np.ascontiguousarray(depth, dtype=np.uint16)
        h, w = depth.shape
        intr = np.ascontiguousarray(intr, dtype=np.float64)
        planes = np.zeros((max(max_planes, 1), 4)); sizes = np.zeros(max(max_planes, 1), dtype=np.int32); n = C.c_int32(0)
        labels = np.zeros((h, w), dtype=np.int32)
        _check(load().esl_extract_planes(self._h, depth.ctypes.data_as(C.POINTER(C.c_uint16)), C.c_int32(w), C.c_int32(h),
                                         intr.ctypes.data_as(_dp), C.byref(p), C.c_int32(max_planes), planes.ctypes.data_as(_dp),
                                         sizes.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n),
                                         labels.ctypes.data_as(C.POINTER(C.c_int32))), "esl_extract_planes")
        k = min(n.value, max_planes)
        return dict(n_planes=n.value, planes=planes[:k], sizes=sizes[:k], labels=labels)

    def selftest_cholesky(self, n):
        """(ms, relative residual) of the dense FP64-MFMA Cholesky factor + solve on a generated SPD system."""
        ms, res = C.c_double(0), C.c_double(0)
        _check(load().esl_selftest_cholesky(self._h, C.c_int32(n), C.byref(ms), C.byref(res)), "esl_selftest_cholesky")
        return ms.value, res.value

    def comm_init(self, n_ranks, rank, unique_id):
        """Join the RCCL communicator; afterwards optimize_resident() is collective over all ranks."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        _check(load().esl_comm_init(self._h, C.c_int32(n_ranks), C.c_int32(rank), buf), "esl_comm_init")

    def comm_init_host(self, n_ranks, rank, allreduce):
        """Collective mode over a host-supplied transport: allreduce(np.ndarray float64) sums in place over ranks."""
        def _cb(_user, buf, count):
            try:
                allreduce(np.ctypeslib.as_array(buf, shape=(count,)))
                return 0
            except Exception:   # an exception must not unwind through the C frames
                import traceback; traceback.print_exc()
                return 1
        self._host_cb = HOST_ALLREDUCE_FN(_cb)   # keep alive as long as the context uses it
        _check(load().esl_comm_init_host(self._h, C.c_int32(n_ranks), C.c_int32(rank), self._host_cb, None), "esl_comm_init_host")

    def comm_set_replicated(self, on=True):
        """Every rank holds the whole graph; the ranks divide the dense solve of SLAM mode (esl_comm_set_replicated)."""
        _check(load().esl_comm_set_replicated(self._h, C.c_int(1 if on else 0)), "esl_comm_set_replicated")

    def comm_destroy(self):
        _check(load().esl_comm_destroy(self._h), "esl_comm_destroy")

    def synchronize(self):
        _check(load().esl_ctx_synchronize(self._h), "esl_ctx_synchronize")

    def trim(self):
        """Release the grow-only solver blobs of SLAM mode (esl_ctx_trim); they are rebuilt on demand."""
        _check(load().esl_ctx_trim(self._h), "esl_ctx_trim")


class _Sizes:
    """vertex counts of a resident graph that has been extended by append_graph (download_states needs them)"""

    def __init__(self, n_cams, n_objs):
        self.n_cams, self.n_objs = n_cams, n_objs


HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64)


def comm_unique_id():
    buf = C.create_string_buffer(128)
    _check(load().esl_comm_unique_id(buf), "esl_comm_unique_id")
    return bytes(buf.raw)


def default_fit_params(**kw):
    """Reference settings (Example/param/TUM3.yaml, PointCloudFilter.cpp:31-38, EllipsoidExtractor.cpp:98,570)."""
    p = abi.EslFitParams()
    load().esl_fit_params_default(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def chol_plan(n, W, filler=128):
    """The persistent dense factorisation's static task list for an order-n system (esl_debug_chol_plan; host only):
    dict(np, n_outer, nR, W, tasks (n_tasks, 4) int32 {type, a, b, c}, ns (np, nR) int32)."""
    meta = (C.c_int32 * 5)()
    ip = C.POINTER(C.c_int32)
    _check(load().esl_debug_chol_plan(C.c_int32(n), C.c_int32(W), C.c_int32(filler), ip(), C.c_int64(0), ip(), C.c_int64(0), meta), "esl_debug_chol_plan")
    np_, n_outer, nR, n_tasks, W_ = [int(v) for v in meta]
    tasks = np.zeros((max(n_tasks, 1), 4), dtype=np.int32); ns = np.zeros((np_, nR), dtype=np.int32)
    _check(load().esl_debug_chol_plan(C.c_int32(n), C.c_int32(W), C.c_int32(filler), tasks.ctypes.data_as(ip), C.c_int64(n_tasks),
                                      ns.ctypes.data_as(ip), C.c_int64(ns.size), meta), "esl_debug_chol_plan")
    return dict(np=np_, n_outer=n_outer, nR=nR, W=W_, tasks=tasks[:n_tasks], ns=ns)


def partition_objects(graph, n_parts):
    """Host-only balanced partition of ellipsoids over shards (works without a GPU)."""
    g = graph.c_struct()
    out = np.zeros(graph.n_objs, dtype=np.int32)
    _check(load().esl_partition_objects(C.byref(g), C.c_int32(n_parts), out.ctypes.data_as(C.POINTER(C.c_int32))),
           "esl_partition_objects")
    return out
