"""Plane-tangency rows pi^T Q* pi (esl_lm_params::bbox_residual = 1): an OPTIONAL residual of the product, off for parity — the
reference has no such edge, it uses the constraint only in its SVD initialiser (src/core/Initializer.cpp:147-164, 271-284; the
north star words it as "plane-tangency edges").  CPU: known answers of the checker's literal 4x4 formulation.  GPU: the
camera-frame closed form of csrc/esl_math.hpp (residual, analytic Jacobians wrt ellipsoid and camera) against it."""
import numpy as np
import pytest


@pytest.fixture()
def tang(po):
    po.set_bbox_residual(1)
    yield po
    po.set_bbox_residual(0)


def test_tangency_vanishes_on_the_exact_bbox(pkg, tang):
    """the four planes through the camera centre and the lines of the EXACT projected box touch the ellipsoid: r = 0; a box
    that is too large gives planes that miss it (r < 0: pi^T Q* pi is negative outside the tangent cone), too small > 0"""
    K = pkg.synth.TUM3_K
    cam = np.array([0.2, -0.1, 0.3, 0.05, -0.02, 0.03, 1.0]); cam[3:] /= np.linalg.norm(cam[3:])
    obj = tang.obj_from_minimal([0.3, -0.2, 2.5, 0.1, -0.2, 0.7, 0.4, 0.25, 0.6])
    tang.set_bbox_residual(0)
    box = tang.project_bbox(cam, obj, K)
    tang.set_bbox_residual(1)
    np.testing.assert_allclose(tang.res_bbox(cam, obj, K, box), 0, atol=1e-12)
    grow = box + np.array([-20, -20, 20, 20.0])
    assert (tang.res_bbox(cam, obj, K, grow) < 0).all()
    assert (tang.res_bbox(cam, obj, K, box - np.array([-20, -20, 20, 20.0])) > 0).all()
    m = box.copy(); m[1] = 3.0       # entries < 5 px are ignored like in the reprojection residual
    assert tang.res_bbox(cam, obj, K, m)[1] == 0


def test_tangency_lm_recovers_the_ellipsoid_on_the_checker(pkg, tang):
    g, c, o, truth = pkg.synth.make_graph(40, 4, 300, seed=8, frac_3d=0.0)
    p = pkg.default_lm_params(numeric_delta=1e-6, bbox_residual=1)
    _, oo, rep = tang.optimize(g, c, o, p, solver=1)
    assert rep["n_bbox_dropped"] == 0 and rep["chi2_final"] < 0.05 * rep["chi2_initial"]
    assert np.abs(oo[:, :3] - truth["objs"][:, :3]).max() < 0.1


@pytest.mark.gpu
@pytest.mark.parametrize("jac", [0, 1])
def test_gpu_tangency_linearisation_matches_checker(pkg, tang, ctx, jac):
    from test_gpu_fullsize import unpack45
    g, c, o, _ = pkg.synth.make_graph(30, 6, 250, seed=12)
    H, b, fidx, chi = tang.build_system(g, c, o, delta=1e-6)
    ctx.upload_graph(g); ctx.upload_states(c, o)
    ctx.lm_begin(pkg.default_lm_params(jacobian_mode=jac, numeric_delta=1e-6, bbox_residual=1))
    part = ctx.lm_linearize()
    assert part.chi2 == pytest.approx(chi, rel=1e-10)
    Hg = ctx.lm_download(0, g.n_objs * 45).reshape(-1, 45)
    bg = ctx.lm_download(1, g.n_objs * 9).reshape(-1, 9)
    for k in range(g.n_objs):
        i = fidx[g.n_cams + k]
        np.testing.assert_allclose(unpack45(Hg[k]), H[i:i + 9, i:i + 9], atol=3e-6 * np.abs(H[i:i + 9, i:i + 9]).max())
        np.testing.assert_allclose(bg[k], b[i:i + 9], atol=3e-6 * np.abs(b[i:i + 9]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("slam", [False, True])
@pytest.mark.parametrize("jac", [0, 1])
def test_gpu_tangency_lm_matches_checker(pkg, tang, ctx, slam, jac):
    """mapping mode (device-driven LM, the analytic run goes through the TANG instantiation of the linearisation kernel) and
    SLAM mode (camera Jacobians of the tangency rows, Schur, Cholesky) against the checker's dense solve"""
    g, c, o, _ = pkg.synth.make_graph(30, 6, 250, seed=13, slam=slam)
    p = pkg.default_lm_params(numeric_delta=1e-6, bbox_residual=1)
    co, oo, ro = tang.optimize(g, c, o, p, solver=0)
    cg, og, rg = ctx.optimize(g, c, o, pkg.default_lm_params(jacobian_mode=jac, numeric_delta=1e-6, bbox_residual=1))
    assert rg["n_bbox_dropped"] == 0
    n = min(len(rg["trace_chi2"]), len(ro["trace_chi2"]), 3)
    np.testing.assert_allclose(rg["trace_chi2"][:n], ro["trace_chi2"][:n], rtol=1e-5)
    assert rg["chi2_final"] == pytest.approx(ro["chi2_final"], rel=1e-4)
    np.testing.assert_allclose(og[:, :3], oo[:, :3], atol=2e-4)
    np.testing.assert_allclose(og[:, 7:], oo[:, 7:], rtol=2e-3)
    if slam:
        np.testing.assert_allclose(cg[:, :3], co[:, :3], atol=2e-4)
    # the reference's residual is untouched by the option: a default run still takes the reprojection edge
    tang.set_bbox_residual(0)
    _, o0, r0 = ctx.optimize(g, c, o, pkg.default_lm_params(jacobian_mode=1))
    assert abs(r0["chi2_initial"] - rg["chi2_initial"]) > 1e-3 * r0["chi2_initial"]


@pytest.mark.gpu
def test_unknown_modes_are_rejected(pkg, ctx):
    g, c, o, _ = pkg.synth.make_graph(10, 2, 30, seed=1)
    for kw in (dict(bbox_residual=7), dict(linear_solver=7)):
        with pytest.raises(pkg.EslError, match="esl_status 2"):
            ctx.optimize(g, c, o, pkg.default_lm_params(**kw))
