"""Committed golden vectors (tests/golden/optimizer_vectors.npz, written by tests/golden/gen_golden.py from the C
restatement): per-edge residuals / Jacobians and whole LM runs.  CPU: the restatement (and the independent numpy
restatement) must keep reproducing them.  GPU: the HIP path must land on the stored LM results."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "optimizer_vectors.npz")


def test_oracle_reproduces_golden_residuals_and_jacobians(po):
    G = np.load(GOLD)
    K, c, o = G["K"], G["cams"], G["objs"]
    for k in range(len(G["bbox_cam"])):
        i, j, m = G["bbox_cam"][k], G["bbox_obj"][k], G["bbox_meas"][k]
        np.testing.assert_allclose(po.res_bbox(c[i], o[j], K, m), G["res_bbox"][k], atol=1e-12)
        np.testing.assert_allclose(po.jac_bbox(c[i], o[j], K, m, 1e-6)[1], G["jac_bbox_obj"][k], atol=1e-9)
    for k in range(len(G["e3d_cam"])):
        np.testing.assert_allclose(po.res_e3d(c[G["e3d_cam"][k]], o[G["e3d_obj"][k]], G["e3d_meas"][k]), G["res_e3d"][k], atol=1e-12)


def test_numpy_restatement_agrees_with_golden_residuals():
    from oracle import np_oracle as npo
    G = np.load(GOLD)
    K, c, o = tuple(G["K"]), G["cams"], G["objs"]
    for k in range(len(G["bbox_cam"])):
        T, (To, s) = npo.T_from7(c[G["bbox_cam"][k]]), npo.obj_from10(o[G["bbox_obj"][k]])
        np.testing.assert_allclose(npo.res_bbox(T, To, s, K, G["bbox_meas"][k]), G["res_bbox"][k], atol=5e-8)
    for k in range(len(G["e3d_cam"])):
        T, (To, s) = npo.T_from7(c[G["e3d_cam"][k]]), npo.obj_from10(o[G["e3d_obj"][k]])
        np.testing.assert_allclose(npo.res_e3d(T, To, s, G["e3d_meas"][k]), G["res_e3d"][k], atol=1e-10)


@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("mode", ["map", "slam"])
def test_oracle_reproduces_golden_lm_runs(po, pkg, seed, mode):
    G = np.load(GOLD)
    g, c, o, _ = pkg.synth.make_graph(20, 5, 80, seed=seed, slam=(mode == "slam"))
    co, ob, rep = po.optimize(g, c, o, pkg.default_lm_params(numeric_delta=1e-6), solver=0)
    tag = f"lm_{seed}_{mode}"
    assert rep["trace_trials"] == list(G[tag + "_trials"])
    np.testing.assert_allclose(rep["trace_chi2"], G[tag + "_chi2"], rtol=1e-10)
    np.testing.assert_allclose(ob, G[tag + "_objs"], atol=1e-10)
    np.testing.assert_allclose(co, G[tag + "_cams"], atol=1e-10)


def test_oracle_reproduces_first_iteration_of_the_mid_size_slam_fixture(po, pkg):
    """tests/golden/mid_slam_run.npz (the camera-first checker's whole optimize(10) on 2,000 cameras / 300 ellipsoids, two minutes on
    one core) is what tests/test_gpu_slam.py holds the GPU's BASELINE-configs[3] form to; here the checker re-runs its FIRST iteration
    (~25 s) and must land on the stored trace: the fixture is this restatement's output, not a stale file."""
    import importlib.util
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("gen_golden_mid_slam", os.path.join(gdir, "gen_golden_mid_slam.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    G = np.load(os.path.join(gdir, "mid_slam_run.npz"))
    g, c, o, _ = pkg.synth.make_graph(**gen.ARGS)
    _, _, rep = po.optimize(g, c, o, pkg.default_lm_params(numeric_delta=float(G["numeric_delta"]), max_iters=1), solver=po.ORACLE_CAMFIRST)
    assert rep["trace_trials"] == [int(G["trace_trials"][0])]
    assert rep["chi2_initial"] == pytest.approx(float(G["chi2_initial"]), rel=1e-12)
    assert rep["trace_chi2"][0] == pytest.approx(float(G["trace_chi2"][0]), rel=1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("mode,jac", [("map", 0), ("map", 1), ("slam", 0), ("slam", 1)])
def test_gpu_lands_on_golden_lm_results(ctx, pkg, seed, mode, jac):
    """The committed LM runs (faithful dense checker, numeric Jacobians at delta = 1e-6) against the GPU with numeric (like for
    like) and analytic Jacobians -- at delta = 1e-6 the two differ by ~1e-7 in J and land within 1e-7 of each other here.
    Tolerances are what the runs measure (printed below), with a factor of ~2: mapping seeds 1, 2: objective 1e-8, states 9e-8;
    mapping seed 0 (an ellipsoid with 3 observations: 8 iterations along a flat valley): 4.4e-5 / 1.3e-5; SLAM (20 cameras, tiny
    and gauge-soft): objective 5e-6, states 2.4e-5.  All inside the north star's 1e-4."""
    G = np.load(GOLD)
    g, c, o, _ = pkg.synth.make_graph(20, 5, 80, seed=seed, slam=(mode == "slam"))
    cg, og, rep = ctx.optimize(g, c, o, pkg.default_lm_params(jacobian_mode=jac, numeric_delta=1e-6))
    tag = f"lm_{seed}_{mode}"
    chi = G[tag + "_chi2"]
    n = min(len(chi), len(rep["trace_chi2"]))
    d_chi = float(np.abs(np.array(rep["trace_chi2"][:n]) / chi[:n] - 1).max())
    d_obj = float(np.abs(og - G[tag + "_objs"]).max())
    d_cam = float(np.abs(cg - G[tag + "_cams"]).max())
    print("golden %s jac %d: chi2 trace rel %.2e (n %d of %d), objs %.2e, cams %.2e" % (tag, jac, d_chi, n, len(chi), d_obj, d_cam))
    if mode == "map":
        t_chi, t_state = (1e-4, 3e-5) if seed == 0 else (1e-7, 2e-7)
    else:
        t_chi, t_state = 1e-5, 5e-5
    assert rep["trace_trials"][:n] == list(G[tag + "_trials"])[:n]
    assert d_chi < t_chi
    assert rep["chi2_final"] == pytest.approx(chi[-1], rel=t_chi)
    assert d_obj < t_state and d_cam < t_state
