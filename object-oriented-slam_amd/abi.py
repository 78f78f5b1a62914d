"""ctypes mirror of include/esl.h (the C-ABI of the MI355X backend).

Only plain-old-data layouts live here.  The product binding (lib.py) builds its argument structs
from these classes; the test-only CPU checker reuses the same layouts so that both sides of a
parity test are always called with byte-identical inputs.
"""
import ctypes as C

import numpy as np

ESL_MAX_TRACE = 32

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)


class EslGraph(C.Structure):
    _fields_ = [
        ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
        ("n_cams", C.c_int32), ("n_objs", C.c_int32),
        ("cam_fixed", c_uint8_p),
        ("n_bbox", C.c_int32), ("bbox_cam", c_int32_p), ("bbox_obj", c_int32_p),
        ("bbox_meas", c_double_p), ("bbox_weight", c_double_p),
        ("n_e3d", C.c_int32), ("e3d_cam", c_int32_p), ("e3d_obj", c_int32_p),
        ("e3d_meas", c_double_p), ("e3d_weight", c_double_p),
        ("n_grav", C.c_int32), ("grav_obj", c_int32_p),
        ("grav_normal", C.c_double * 4), ("grav_weight", C.c_double),
        ("n_odom", C.c_int32), ("odom_i", c_int32_p), ("odom_j", c_int32_p),
        ("odom_meas", c_double_p), ("odom_info", c_double_p),
        ("check_visibility", C.c_int32), ("image_rows", C.c_int32), ("image_cols", C.c_int32),
    ]


class EslGraphDelta(C.Structure):
    """esl_graph_delta: what one new frame adds to the resident graph (esl_graph_append)."""
    _fields_ = [
        ("n_new_cams", C.c_int32), ("new_cams", c_double_p), ("n_new_objs", C.c_int32), ("new_objs", c_double_p),
        ("n_bbox", C.c_int32), ("bbox_cam", c_int32_p), ("bbox_obj", c_int32_p), ("bbox_meas", c_double_p), ("bbox_weight", c_double_p),
        ("n_e3d", C.c_int32), ("e3d_cam", c_int32_p), ("e3d_obj", c_int32_p), ("e3d_meas", c_double_p), ("e3d_weight", c_double_p),
        ("n_grav", C.c_int32), ("grav_obj", c_int32_p),
        ("new_cam_fixed", C.POINTER(C.c_uint8)), ("n_odom", C.c_int32), ("odom_i", c_int32_p), ("odom_j", c_int32_p),
        ("odom_meas", c_double_p), ("odom_info", c_double_p),
    ]


class EslLmParams(C.Structure):
    _fields_ = [
        ("max_iters", C.c_int32), ("max_trials", C.c_int32), ("tau", C.c_double),
        ("jacobian_mode", C.c_int32), ("numeric_delta", C.c_double),
        ("linear_solver", C.c_int32), ("drop_nan_bbox", C.c_int32), ("bbox_residual", C.c_int32), ("e3d_half_turn", C.c_int32),
    ]


class EslLmReport(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32), ("total_trials", C.c_int32),
        ("n_bbox_valid", C.c_int32), ("n_bbox_dropped", C.c_int32), ("stop_reason", C.c_int32),
        ("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("lambda_final", C.c_double),
        ("trace_len", C.c_int32),
        ("trace_chi2", C.c_double * ESL_MAX_TRACE), ("trace_lambda", C.c_double * ESL_MAX_TRACE),
        ("trace_trials", C.c_int32 * ESL_MAX_TRACE),
    ]

    def as_dict(self):
        n = self.trace_len
        return dict(iterations=self.iterations, total_trials=self.total_trials,
                    n_bbox_valid=self.n_bbox_valid, n_bbox_dropped=self.n_bbox_dropped,
                    stop_reason=self.stop_reason, chi2_initial=self.chi2_initial,
                    chi2_final=self.chi2_final, lambda_final=self.lambda_final,
                    trace_chi2=list(self.trace_chi2[:n]), trace_lambda=list(self.trace_lambda[:n]),
                    trace_trials=list(self.trace_trials[:n]))


class EslLmPartials(C.Structure):
    _fields_ = [("chi2", C.c_double), ("max_diag", C.c_double), ("scale", C.c_double),
                ("solve_ok", C.c_int32), ("pad", C.c_int32)]


class EslFitParams(C.Structure):
    _fields_ = [
        ("stride", C.c_int32), ("depth_scale", C.c_double), ("depth_min", C.c_double),
        ("depth_max", C.c_double), ("voxel_leaf", C.c_double), ("plane_dist", C.c_double),
        ("cluster_tolerance", C.c_double), ("min_cluster_size", C.c_int32), ("center_dis", C.c_double),
        ("symmetry_open", C.c_int32), ("symmetry_grid", C.c_double), ("symmetry_sigma", C.c_double),
        ("symmetry_lm_iters", C.c_int32),
    ]


class EslFitSymmetry(C.Structure):
    """SymmetryOutputData (reference src/symmetry/Symmetry.h:16-32) as esl_fit_frame_ex returns it."""
    _fields_ = [("result", C.c_int32), ("symmetry_type", C.c_int32), ("plane", C.c_double * 4),
                ("plane2", C.c_double * 4), ("prob", C.c_double), ("center", C.c_double * 3)]


class EslPlaneParams(C.Structure):
    """esl_plane_params (PlaneExtractorParam + the PCL constants of PlaneExtractor.cpp:57-58, 74)."""
    _fields_ = [("min_size", C.c_int32), ("angle_threshold_deg", C.c_double), ("distance_threshold", C.c_double),
                ("normal_smoothing", C.c_int32), ("max_depth_change_factor", C.c_double), ("min_inliers", C.c_int32),
                ("refine", C.c_int32), ("refine_distance", C.c_double), ("max_curvature", C.c_double)]


def default_plane_params(**kw):
    """Example/param/TUM3.yaml:36-38 + PlaneExtractor.cpp:57-58, 74."""
    p = EslPlaneParams(min_size=200, angle_threshold_deg=5.0, distance_threshold=0.1, normal_smoothing=10,
                       max_depth_change_factor=0.05, min_inliers=100, refine=1, refine_distance=0.02, max_curvature=0.001)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


SOLVER_AUTO, SOLVER_REDUCED_CAMERA, SOLVER_REDUCED_ELLIPSOID = 0, 1, 2   # esl_linear_solver (include/esl.h)


def default_lm_params(**kw):
    """Reference settings: optimize(10) (Optimizer.cpp:291), tau 1e-5, 10 trials
    (optimization_algorithm_levenberg.cpp:45-49), delta 1e-9 (base_binary_edge.hpp:147)."""
    p = EslLmParams(max_iters=10, max_trials=10, tau=1e-5, jacobian_mode=0, numeric_delta=1e-9,
                    linear_solver=0, drop_nan_bbox=1, bbox_residual=0, e3d_half_turn=0)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def _arr(a, dtype, shape_tail=None):
    a = np.ascontiguousarray(a, dtype=dtype)
    if shape_tail is not None and a.size:
        a = a.reshape((-1,) + tuple(shape_tail))
    return a


class Graph:
    """Host-side SoA graph.  Keeps the numpy arrays alive and exposes an EslGraph view."""

    def __init__(self, K, n_cams, n_objs, cam_fixed=None,
                 bbox_cam=(), bbox_obj=(), bbox_meas=(), bbox_weight=(),
                 e3d_cam=(), e3d_obj=(), e3d_meas=(), e3d_weight=(),
                 grav_obj=(), grav_normal=(0, 0, 1, 0), grav_weight=0.0,
                 odom_i=(), odom_j=(), odom_meas=(), odom_info=None, check_visibility=0, image_rows=480, image_cols=640):
        self.K = tuple(float(k) for k in K)
        self.n_cams, self.n_objs = int(n_cams), int(n_objs)
        self.cam_fixed = None if cam_fixed is None else _arr(cam_fixed, np.uint8)
        self.bbox_cam = _arr(bbox_cam, np.int32); self.bbox_obj = _arr(bbox_obj, np.int32)
        self.bbox_meas = _arr(bbox_meas, np.float64, (4,)); self.bbox_weight = _arr(bbox_weight, np.float64)
        self.e3d_cam = _arr(e3d_cam, np.int32); self.e3d_obj = _arr(e3d_obj, np.int32)
        self.e3d_meas = _arr(e3d_meas, np.float64, (10,)); self.e3d_weight = _arr(e3d_weight, np.float64)
        self.grav_obj = _arr(grav_obj, np.int32)
        self.grav_normal = tuple(float(v) for v in grav_normal); self.grav_weight = float(grav_weight)
        self.odom_i = _arr(odom_i, np.int32); self.odom_j = _arr(odom_j, np.int32)
        self.odom_meas = _arr(odom_meas, np.float64, (7,))
        self.odom_info = None if odom_info is None else _arr(odom_info, np.float64, (6,))
        self.check_visibility, self.image_rows, self.image_cols = int(check_visibility), int(image_rows), int(image_cols)
        assert len(self.bbox_cam) == len(self.bbox_obj) == len(self.bbox_weight) == len(self.bbox_meas.reshape(-1, 4))
        assert len(self.e3d_cam) == len(self.e3d_obj) == len(self.e3d_weight) == len(self.e3d_meas.reshape(-1, 10))
        assert len(self.odom_i) == len(self.odom_j) == len(self.odom_meas.reshape(-1, 7))

    @staticmethod
    def _p(a, ptype):
        return a.ctypes.data_as(ptype) if a is not None and a.size else ptype()

    def c_struct(self):
        g = EslGraph()
        g.fx, g.fy, g.cx, g.cy = self.K
        g.n_cams, g.n_objs = self.n_cams, self.n_objs
        g.cam_fixed = self._p(self.cam_fixed, c_uint8_p)
        g.n_bbox = len(self.bbox_cam)
        g.bbox_cam = self._p(self.bbox_cam, c_int32_p); g.bbox_obj = self._p(self.bbox_obj, c_int32_p)
        g.bbox_meas = self._p(self.bbox_meas, c_double_p); g.bbox_weight = self._p(self.bbox_weight, c_double_p)
        g.n_e3d = len(self.e3d_cam)
        g.e3d_cam = self._p(self.e3d_cam, c_int32_p); g.e3d_obj = self._p(self.e3d_obj, c_int32_p)
        g.e3d_meas = self._p(self.e3d_meas, c_double_p); g.e3d_weight = self._p(self.e3d_weight, c_double_p)
        g.n_grav = len(self.grav_obj)
        g.grav_obj = self._p(self.grav_obj, c_int32_p)
        g.grav_normal = (C.c_double * 4)(*self.grav_normal); g.grav_weight = self.grav_weight
        g.n_odom = len(self.odom_i)
        g.odom_i = self._p(self.odom_i, c_int32_p); g.odom_j = self._p(self.odom_j, c_int32_p)
        g.odom_meas = self._p(self.odom_meas, c_double_p); g.odom_info = self._p(self.odom_info, c_double_p)
        g.check_visibility, g.image_rows, g.image_cols = self.check_visibility, self.image_rows, self.image_cols
        return g

    def subset_objects(self, keep):
        """Sub-graph holding only the ellipsoids in `keep` (ascending), re-indexed; cameras are kept.
        Used by the sharded driver (SURVEY.md §8 e) and by bench.py's bounded CPU sample."""
        keep = np.asarray(keep, dtype=np.int64)
        remap = -np.ones(self.n_objs, dtype=np.int64)
        remap[keep] = np.arange(len(keep))
        mb = remap[self.bbox_obj] >= 0 if len(self.bbox_obj) else np.zeros(0, bool)
        me = remap[self.e3d_obj] >= 0 if len(self.e3d_obj) else np.zeros(0, bool)
        mg = remap[self.grav_obj] >= 0 if len(self.grav_obj) else np.zeros(0, bool)
        return Graph(self.K, self.n_cams, len(keep), self.cam_fixed,
                     self.bbox_cam[mb], remap[self.bbox_obj[mb]] if mb.any() else (),
                     self.bbox_meas.reshape(-1, 4)[mb], self.bbox_weight[mb],
                     self.e3d_cam[me], remap[self.e3d_obj[me]] if me.any() else (),
                     self.e3d_meas.reshape(-1, 10)[me], self.e3d_weight[me],
                     remap[self.grav_obj[mg]] if mg.any() else (), self.grav_normal, self.grav_weight,
                     self.odom_i, self.odom_j, self.odom_meas, self.odom_info, self.check_visibility, self.image_rows, self.image_cols)

    def subset(self, cams, objs):
        """Sub-graph on the cameras `cams` and the ellipsoids `objs` (both ascending, re-indexed in that order): the edges whose
        two ends are both kept.  Used by the full-size parity tests (a sample of BASELINE configs[3] the CPU checker can hold)."""
        cams = np.asarray(cams, dtype=np.int64); objs = np.asarray(objs, dtype=np.int64)
        rc = -np.ones(self.n_cams, dtype=np.int64); rc[cams] = np.arange(len(cams))
        ro = -np.ones(self.n_objs, dtype=np.int64); ro[objs] = np.arange(len(objs))
        mb = (rc[self.bbox_cam] >= 0) & (ro[self.bbox_obj] >= 0) if len(self.bbox_cam) else np.zeros(0, bool)
        me = (rc[self.e3d_cam] >= 0) & (ro[self.e3d_obj] >= 0) if len(self.e3d_cam) else np.zeros(0, bool)
        mg = ro[self.grav_obj] >= 0 if len(self.grav_obj) else np.zeros(0, bool)
        mo = (rc[self.odom_i] >= 0) & (rc[self.odom_j] >= 0) if len(self.odom_i) else np.zeros(0, bool)
        fixed = None if self.cam_fixed is None else self.cam_fixed[cams]
        return Graph(self.K, len(cams), len(objs), fixed,
                     rc[self.bbox_cam[mb]], ro[self.bbox_obj[mb]], self.bbox_meas.reshape(-1, 4)[mb], self.bbox_weight[mb],
                     rc[self.e3d_cam[me]], ro[self.e3d_obj[me]], self.e3d_meas.reshape(-1, 10)[me], self.e3d_weight[me],
                     ro[self.grav_obj[mg]], self.grav_normal, self.grav_weight,
                     rc[self.odom_i[mo]], rc[self.odom_j[mo]], self.odom_meas.reshape(-1, 7)[mo],
                     None if self.odom_info is None else self.odom_info.reshape(-1, 6)[mo],
                     self.check_visibility, self.image_rows, self.image_cols)
