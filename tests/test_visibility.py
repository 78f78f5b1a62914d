"""The optional visibility test of the bbox edges (checkVisibility, reference src/core/Optimizer.cpp:35-81; the argument
check_visibility of GlobalObjectGraphOptimization, which the reference's only call site leaves false): an edge is kept iff
the ellipsoid is in front of its camera, the camera is outside the ellipsoid and the projected centre or a corner of the
projected box lies inside the image.  esl_graph::check_visibility switches it on."""
import numpy as np
import pytest


def vis_graph(pkg, po):
    """6 ellipsoids seen by 30 cameras + hand-made offenders on ellipsoid 0: a camera looking away (centre behind it), a camera
    inside the ellipsoid, a camera that sees it far outside a 640x480 image"""
    from oracle import np_fit
    g, c, o, _ = pkg.synth.make_graph(30, 6, 260, seed=21)
    c = c.copy()
    ids = np.nonzero(g.bbox_obj == 0)[0][:3]
    cams = g.bbox_cam[ids]
    Twc = np_fit.se3_inv(c[cams[0]]); Twc[3:] = np_fit.q_mul(Twc[3:], np.array([0, 1.0, 0, 0])); c[cams[0]] = np_fit.se3_inv(Twc)   # turn by 180 deg about y
    Twc = np_fit.se3_inv(c[cams[1]]); Twc[:3] = o[0][:3] + 0.01; c[cams[1]] = np_fit.se3_inv(Twc)                                   # inside
    Twc = np_fit.se3_inv(c[cams[2]]); Twc[3:] = np_fit.q_mul(Twc[3:], np.array([0, np.sin(0.6), 0, np.cos(0.6)])); c[cams[2]] = np_fit.se3_inv(Twc)  # 69 deg
    gv = pkg.Graph(g.K, g.n_cams, g.n_objs, None, g.bbox_cam, g.bbox_obj, g.bbox_meas, g.bbox_weight, g.e3d_cam, g.e3d_obj, g.e3d_meas,
                   g.e3d_weight, g.grav_obj, g.grav_normal, g.grav_weight, check_visibility=1, image_rows=480, image_cols=640)
    return g, gv, c, o


def test_checker_drops_invisible_edges(pkg, po):
    g, gv, c, o = vis_graph(pkg, po)
    p = pkg.default_lm_params(numeric_delta=1e-6, max_iters=1)
    r0 = po.optimize(g, c, o, p, solver=1)[2]
    r1 = po.optimize(gv, c, o, p, solver=1)[2]
    assert r1["n_bbox_dropped"] > r0["n_bbox_dropped"] and r1["n_bbox_dropped"] >= 2
    assert r1["n_bbox_valid"] + r1["n_bbox_dropped"] == len(g.bbox_cam)


@pytest.mark.gpu
@pytest.mark.parametrize("jac", [0, 1])
def test_gpu_visibility_check_matches_checker(pkg, po, ctx, jac):
    g, gv, c, o = vis_graph(pkg, po)
    p = pkg.default_lm_params(jacobian_mode=jac, numeric_delta=1e-6)
    _, oo, ro = po.optimize(gv, c, o, pkg.default_lm_params(numeric_delta=1e-6), solver=1)
    _, og, rg = ctx.optimize(gv, c, o, p)
    assert (rg["n_bbox_valid"], rg["n_bbox_dropped"]) == (ro["n_bbox_valid"], ro["n_bbox_dropped"])
    assert rg["chi2_initial"] == pytest.approx(ro["chi2_initial"], rel=1e-9)
    np.testing.assert_allclose(rg["trace_chi2"][:2], ro["trace_chi2"][:2], rtol=1e-5)
    np.testing.assert_allclose(og[:, :3], oo[:, :3], atol=1e-4)
    # step API / SLAM-mode validation kernel takes the same decision
    ctx.upload_graph(gv); ctx.upload_states(c, o)
    nv, nd = ctx.lm_begin(p)
    assert (nv, nd) == (ro["n_bbox_valid"], ro["n_bbox_dropped"])
    # and with the test off nothing but NaN edges goes
    _, _, r_off = ctx.optimize(g, c, o, p)
    assert r_off["n_bbox_dropped"] < rg["n_bbox_dropped"]
