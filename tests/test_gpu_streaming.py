"""BASELINE.json configs[4] — streaming: per frame the single-frame fit of its boxes and the re-optimisation of the graph that
has accumulated so far, with the graph EXTENDED on the device (esl_graph_append) instead of rebuilt (the reference rebuilds every
g2o object per frame, Optimizer.cpp:127, 166, 250).  The appended graph must be indistinguishable from an upload of the
concatenated one — same chunks, same sums: bit-identical runs — and every frame's result must match the CPU checker."""
import numpy as np
import pytest

from test_gpu_optimizer import group_rel_err

pytestmark = pytest.mark.gpu


def frame_slices(g, f):
    """edges of frame f (camera index f), bbox edges only once their ellipsoid has > 2 observations (Optimizer.cpp:201): the
    frame in which an ellipsoid reaches 3 observations contributes its earlier ones too"""
    cnt_before = np.bincount(g.bbox_obj[g.bbox_cam < f], minlength=g.n_objs)
    cnt_now = np.bincount(g.bbox_obj[g.bbox_cam <= f], minlength=g.n_objs)
    mb = np.zeros(len(g.bbox_cam), bool)
    for i in range(len(g.bbox_cam)):
        o, cam = g.bbox_obj[i], g.bbox_cam[i]
        if cnt_now[o] <= 2 or cam > f:
            continue
        if cam == f or cnt_before[o] <= 2:
            mb[i] = True
    return mb, g.e3d_cam == f


def graph_upto(pkg, g, f):
    mb = g.bbox_cam <= f
    cnt = np.bincount(g.bbox_obj[mb], minlength=g.n_objs)
    mb &= cnt[g.bbox_obj] > 2
    me = g.e3d_cam <= f
    return pkg.Graph(g.K, f + 1, g.n_objs, None, g.bbox_cam[mb], g.bbox_obj[mb], g.bbox_meas.reshape(-1, 4)[mb], g.bbox_weight[mb],
                     g.e3d_cam[me], g.e3d_obj[me], g.e3d_meas.reshape(-1, 10)[me], g.e3d_weight[me], g.grav_obj, g.grav_normal, g.grav_weight)


def per_obj_err(a, b):
    """the north-star metric per ellipsoid: max of |dt| / |t|, |dq|, |ds| / |s|"""
    a, b = np.asarray(a).reshape(-1, 10), np.asarray(b).reshape(-1, 10)
    et = np.linalg.norm(a[:, :3] - b[:, :3], axis=1) / np.linalg.norm(b[:, :3], axis=1)
    eq = np.linalg.norm(a[:, 3:7] - b[:, 3:7], axis=1)
    es = np.linalg.norm(a[:, 7:] - b[:, 7:], axis=1) / np.linalg.norm(b[:, 7:], axis=1)
    return np.maximum(np.maximum(et, eq), es)


def yaw_hypothesis_gap(cam7, obj10, meas10):
    """relative gap between the two smallest of the four yaw-hypothesis residual norms of a 3-D edge (Ellipsoid.cpp:92-117);
    a gap near 0 is a kink of the min() the edge's residual takes"""
    from oracle import np_oracle as npo
    Tcw = npo.T_from7(cam7); To, s = npo.obj_from10(obj10); Tm, sm = npo.obj_from10(meas10)
    Tmw = npo.T_inv(Tcw) @ Tm
    norms = []
    for k in (-1, 0, 1, 2):
        a = k * np.pi / 2
        Rz = np.eye(4); Rz[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
        sk = sm.copy()
        if k in (-1, 1):
            sk[0], sk[1] = sm[1], sm[0]
        with np.errstate(all="ignore"):
            norms.append(np.linalg.norm(np.concatenate([npo.se3_log(npo.T_inv(Tmw @ Rz) @ To), s - sk])))
    norms = np.sort(np.nan_to_num(norms, nan=np.inf))
    return float((norms[1] - norms[0]) / norms[0])


def constraint_of(po, gf, cams, objs, oid):
    """how well the graph pins ellipsoid oid at the state `objs`: (bbox edges, 3-D edges, condition number of its 9 x 9 block of
    the checker's normal equations, smallest yaw-hypothesis gap of its 3-D edges)"""
    sub = gf.subset_objects([oid])
    H, b, fidx, _ = po.build_system(sub, cams, objs[oid:oid + 1], delta=1e-6)
    w = np.linalg.eigvalsh(H)
    cond = float(w[-1] / max(w[0], 1e-300)) if w[0] > 0 else float("inf")
    gaps = [yaw_hypothesis_gap(cams[sub.e3d_cam[k]], objs[oid], sub.e3d_meas.reshape(-1, 10)[k]) for k in range(len(sub.e3d_cam))]
    return len(sub.bbox_cam), len(sub.e3d_cam), cond, (min(gaps) if gaps else float("inf"))


def weak_constraint_report(po, gf, cams, objs, oid):
    nb, ne, cond, gap = constraint_of(po, gf, cams, objs, oid)
    return "ellipsoid %d: %d bbox + %d 3-D edges, cond(H_oo) %.1e, yaw-hypothesis gap %.2e" % (oid, nb, ne, cond, gap)


def test_streaming_sequence_append_equals_rebuild_and_checker(pkg, po, ctx):
    n_frames = 60
    g, c, o, _ = pkg.synth.make_graph(n_frames, 20, 20 * n_frames, seed=3)
    sc = pkg.synth.make_depth_scene(n_objs=20, seed=7, spread=1.6, size=(0.1, 0.3))
    P = pkg.lib.default_fit_params()
    p = pkg.default_lm_params(jacobian_mode=1)
    ctx_inc = pkg.Context(0)
    objs_inc = o.copy(); objs_reb = o.copy()
    e_fit_ref, p_fit_ref, st_ref, _ = po.fit_frame(sc["depth"], sc["bboxes"], sc["labels"], sc["Twc"], sc["intr"], sc["ground"])
    relayouts = 0
    for f in range(n_frames):
        # the frame's single-frame fits (20 boxes) ...
        e_fit, p_fit, st, _ = ctx_inc.fit_frame(sc["depth"], sc["bboxes"], sc["labels"], sc["Twc"], sc["intr"], sc["ground"], P)
        np.testing.assert_array_equal(st, st_ref)
        if f in (0, n_frames - 1):
            np.testing.assert_allclose(e_fit[st == 0], e_fit_ref[st_ref == 0], atol=1e-4)
        # ... then the re-optimisation of everything seen so far: (a) device-resident graph extended by this frame's edges
        mb, me = frame_slices(g, f)
        if f == 0:
            g0 = graph_upto(pkg, g, 0)
            ctx_inc.upload_graph(g0); ctx_inc.upload_states(c[:1], objs_inc)
        else:
            ctx_inc.append_graph(new_cams=c[f:f + 1], bbox=(g.bbox_cam[mb], g.bbox_obj[mb], g.bbox_meas.reshape(-1, 4)[mb], g.bbox_weight[mb]),
                                 e3d=(g.e3d_cam[me], g.e3d_obj[me], g.e3d_meas.reshape(-1, 10)[me], g.e3d_weight[me]))
        rep_inc = ctx_inc.optimize_resident(p)
        _, objs_inc = ctx_inc.download_states()
        # (b) the same frame with the whole graph rebuilt and re-uploaded, as the reference does
        gf = graph_upto(pkg, g, f)
        objs_before = objs_reb.copy()
        _, objs_reb, rep_reb = ctx.optimize(gf, c[:f + 1], objs_reb, p)
        sizes = ctx_inc.graph_sizes()
        assert (sizes["n_cams"], sizes["n_bbox"], sizes["n_e3d"]) == (f + 1, len(gf.bbox_cam), len(gf.e3d_cam))
        assert rep_inc["trace_chi2"] == rep_reb["trace_chi2"] and rep_inc["trace_trials"] == rep_reb["trace_trials"], f
        np.testing.assert_array_equal(objs_inc, objs_reb)
        relayouts = sizes["relayouts"]
        # (c) the checker on this frame's problem (same start state, like for like: numeric Jacobians at delta = 1e-6 -- the 3-D
        #     edge takes the minimum over four yaw hypotheses, so analytic and numeric LM paths may part at a kink)
        # Checked frames: 0, 6 and every sixth from 30 on (+ the last).  Frames 12 .. 24 are left out on purpose (SURVEY.md section 7:
        # keep parity fixtures away from yaw kinks): there most ellipsoids still hang on 3-D edges alone, whose residual is a minimum
        # over four yaw hypotheses, and the two implementations' central differences across such a kink differ in the last bits, so
        # the two LM runs can leave a kink on different sides (measured: states 7.7e-3 / 1.8e-4 apart at frames 12 / 24 with chi2
        # equal to 3e-5 / 5e-7).  Everything checked holds the north star's 1e-4 outright (measured 8e-10 .. 2.2e-6).
        if f in (0, 6) or (f >= 30 and f % 6 == 0) or f == n_frames - 1:
            pn = pkg.default_lm_params(jacobian_mode=0, numeric_delta=1e-6)
            _, o_orc, r_orc = po.optimize(gf, c[:f + 1], objs_before, pn, solver=1)
            _, o_gpu, r_gpu = ctx.optimize(gf, c[:f + 1], objs_before, pn)
            assert r_gpu["chi2_final"] == pytest.approx(r_orc["chi2_final"], rel=1e-7), f     # measured 3e-10 .. 9e-9
            assert r_gpu["n_bbox_valid"] == r_orc["n_bbox_valid"]
            err = group_rel_err(o_gpu, o_orc)
            print("streaming frame %d: GPU vs checker chi2 rel %.2e, states %.2e" % (f, abs(r_gpu["chi2_final"] / r_orc["chi2_final"] - 1), err))
            assert err < 1e-4, (f, err)
    assert 1 <= relayouts <= 4, relayouts      # 60 appends, a handful of re-layouts (slack doubles)
    ctx_inc.close()


def test_append_new_ellipsoids_gravity_and_errors(pkg, ctx):
    g, c, o, _ = pkg.synth.make_graph(30, 6, 240, seed=19)
    p = pkg.default_lm_params(jacobian_mode=1)
    # start with 4 ellipsoids and 20 cameras, then add 2 ellipsoids (with gravity priors) + 10 cameras + all their edges at once
    keepb = (g.bbox_obj < 4) & (g.bbox_cam < 20); keepe = (g.e3d_obj < 4) & (g.e3d_cam < 20)
    g0 = pkg.Graph(g.K, 20, 4, None, g.bbox_cam[keepb], g.bbox_obj[keepb], g.bbox_meas.reshape(-1, 4)[keepb], g.bbox_weight[keepb],
                   g.e3d_cam[keepe], g.e3d_obj[keepe], g.e3d_meas.reshape(-1, 10)[keepe], g.e3d_weight[keepe], np.arange(4), g.grav_normal, g.grav_weight)
    cx = pkg.Context(0)
    cx.upload_graph(g0); cx.upload_states(c[:20], o[:4])
    with pytest.raises(pkg.EslError, match="out of range"):
        cx.append_graph(bbox=([25], [0], [[10, 10, 50, 50]], [1.0]))
    cx.append_graph(new_cams=c[20:], new_objs=o[4:], bbox=(g.bbox_cam[~keepb], g.bbox_obj[~keepb], g.bbox_meas.reshape(-1, 4)[~keepb], g.bbox_weight[~keepb]),
                    e3d=(g.e3d_cam[~keepe], g.e3d_obj[~keepe], g.e3d_meas.reshape(-1, 10)[~keepe], g.e3d_weight[~keepe]), grav_obj=[4, 5])
    rep = cx.optimize_resident(p)
    _, o_app = cx.download_states()
    # reference result: upload of the whole graph with the edges in the same per-ellipsoid order (old ones first)
    ob = np.concatenate([np.nonzero(keepb)[0], np.nonzero(~keepb)[0]]); oe = np.concatenate([np.nonzero(keepe)[0], np.nonzero(~keepe)[0]])
    gw = pkg.Graph(g.K, 30, 6, None, g.bbox_cam[ob], g.bbox_obj[ob], g.bbox_meas.reshape(-1, 4)[ob], g.bbox_weight[ob], g.e3d_cam[oe], g.e3d_obj[oe],
                   g.e3d_meas.reshape(-1, 10)[oe], g.e3d_weight[oe], np.arange(6), g.grav_normal, g.grav_weight)
    _, o_ref, r_ref = ctx.optimize(gw, c, o, p)
    assert rep["trace_chi2"] == r_ref["trace_chi2"]
    np.testing.assert_array_equal(o_app, o_ref)
    # a second, small append goes into the slack (no further re-layout), numeric Jacobians work on the slack layout too
    r0 = cx.graph_sizes()["relayouts"]
    cx.append_graph(bbox=([3], [1], [[100, 100, 200, 200.0]], [0.5]))
    assert cx.graph_sizes()["relayouts"] == r0
    rep2 = cx.optimize_resident(pkg.default_lm_params(jacobian_mode=0, numeric_delta=1e-6))
    assert rep2["n_bbox_valid"] == len(g.bbox_cam) + 1 and np.isfinite(rep2["chi2_final"])
    # SLAM-mode graphs are rejected
    gs, cs, os_, _ = pkg.synth.make_graph(10, 2, 30, seed=1, slam=True)
    cx.upload_graph(gs); cx.upload_states(cs, os_)
    with pytest.raises(pkg.EslError, match="mapping-mode"):
        cx.append_graph(new_cams=cs[:1])
    cx.close()


def test_fit_and_optimise_overlap_on_two_contexts(pkg, ctx):
    """Two contexts = two HIP streams: a worker thread fits frames on one while the main thread optimises on the other (the
    pipelined streaming mode of bench.py).  Same results as when the calls run one after the other, every time."""
    import threading
    sc = pkg.synth.make_depth_scene(n_objs=6, seed=4, spread=1.2, size=(0.1, 0.3))
    P = pkg.lib.default_fit_params()
    g, c, o, _ = pkg.synth.make_graph(40, 8, 400, seed=8)
    p = pkg.default_lm_params(jacobian_mode=1)
    e_ref, pr_ref, st_ref = ctx.fit_frame(sc["depth"], sc["bboxes"], sc["labels"], sc["Twc"], sc["intr"], sc["ground"], P)[:3]
    _, o_ref, r_ref = ctx.optimize(g, c, o, p)
    fctx = pkg.Context(0)
    fits, errs = [], []

    def worker():
        try:
            for _ in range(12):
                fits.append(fctx.fit_frame(sc["depth"], sc["bboxes"], sc["labels"], sc["Twc"], sc["intr"], sc["ground"], P)[:3])
        except Exception as e:   # noqa: BLE001
            errs.append(e)
    try:
        th = threading.Thread(target=worker)
        th.start()
        runs = [ctx.optimize(g, c, o, p) for _ in range(12)]
        th.join(120)
        assert not th.is_alive() and not errs, errs
    finally:
        fctx.close()
    for _, oo, rr in runs:
        assert rr["trace_trials"] == r_ref["trace_trials"]
        np.testing.assert_array_equal(oo, o_ref)
    for e, pr, st in fits:
        assert np.array_equal(st, st_ref)
        np.testing.assert_allclose(e, e_ref, atol=1e-7)      # run-to-run spread of the fit itself
