"""Corners of the device-resident LM loop (esl_optimize_resident, mapping mode) against the CPU restatement: iteration /
trial limits, the early-stop rule, graphs whose only edges are gravity priors or NaN bbox edges, back-to-back runs on
one context (the control block, the NaN counter and the queued no-op launches must leave nothing behind), and a
sharded run in which one shard is empty."""
import numpy as np
import pytest

from test_gpu_optimizer import assert_traces_match, group_rel_err
from test_gpu_sharded import ThreadAllreduce

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("max_iters,max_trials", [(1, 10), (3, 1), (10, 2), (25, 10)])
def test_iteration_and_trial_limits_match_oracle(pkg, po, ctx, max_iters, max_trials):
    g, c, o, _ = pkg.synth.make_graph(40, 8, 350, seed=31)
    p = pkg.default_lm_params(numeric_delta=1e-6, max_iters=max_iters, max_trials=max_trials)
    _, oo, ro = po.optimize(g, c, o, p, solver=1)
    _, og, rg = ctx.optimize(g, c, o, p)
    assert rg["iterations"] == ro["iterations"] <= max_iters
    assert rg["stop_reason"] == ro["stop_reason"]
    assert_traces_match(rg, ro)
    np.testing.assert_allclose(og, oo, atol=2e-5)


def test_gravity_only_graph(pkg, po, ctx):
    g0, c, o, _ = pkg.synth.make_graph(10, 4, 0, seed=2)
    g = pkg.Graph(g0.K, g0.n_cams, g0.n_objs, None, grav_obj=np.arange(g0.n_objs), grav_normal=g0.grav_normal, grav_weight=g0.grav_weight)
    p = pkg.default_lm_params(numeric_delta=1e-6)
    _, oo, ro = po.optimize(g, c, o, p, solver=1)
    _, og, rg = ctx.optimize(g, c, o, p)
    assert rg["chi2_initial"] == pytest.approx(ro["chi2_initial"], rel=1e-9)
    assert rg["chi2_final"] == pytest.approx(ro["chi2_final"], rel=1e-5, abs=1e-12)
    np.testing.assert_allclose(og, oo, atol=2e-5)


def test_all_bbox_edges_nan_is_an_empty_graph(pkg, po, ctx):
    """Every camera sits inside the only ellipsoid: each bbox edge is NaN at the start state and is dropped
    (Optimizer.cpp:234-243); nothing is left to optimise and the states come back untouched."""
    K = pkg.synth.TUM3_K
    o = np.array([[0, 0, 0, 0, 0, 0, 1, 2.0, 2.0, 2.0]])
    c = np.tile([0.1, 0.0, 0.0, 0, 0, 0, 1.0], (4, 1)); c[:, 0] = [0.1, -0.1, 0.2, 0.0]
    g = pkg.Graph(K, 4, 1, None, np.arange(4), np.zeros(4, int), np.tile([100, 100, 300, 300.0], (4, 1)), np.ones(4))
    p = pkg.default_lm_params()
    _, oo, ro = po.optimize(g, c, o, p, solver=1)
    _, og, rg = ctx.optimize(g, c, o, p)
    assert ro["n_bbox_dropped"] == 4
    assert (rg["n_bbox_valid"], rg["n_bbox_dropped"], rg["stop_reason"]) == (0, 4, 3)
    assert rg["iterations"] == 0
    np.testing.assert_array_equal(og, o)


def test_back_to_back_runs_leave_no_state_behind(pkg, ctx):
    """Different graphs and parameters on one context, interleaved: every run equals the same run on a fresh context."""
    cases = []
    for seed, kw in [(3, dict(jacobian_mode=1)), (4, dict(jacobian_mode=0, numeric_delta=1e-6)), (5, dict(jacobian_mode=1, max_iters=2)),
                     (3, dict(jacobian_mode=1))]:
        g, c, o, _ = pkg.synth.make_graph(30 + seed, 6 + seed, 200 + 40 * seed, seed=seed)
        cases.append((g, c, o, pkg.default_lm_params(**kw)))
    fresh = []
    for g, c, o, p in cases:
        cx = pkg.Context(0)
        _, oo, rr = cx.optimize(g, c, o, p)
        fresh.append((oo, rr))
        cx.close()
    for (g, c, o, p), (oo, rr) in zip(cases, fresh):
        _, og, rg = ctx.optimize(g, c, o, p)
        assert rg["trace_chi2"] == rr["trace_chi2"] and rg["trace_trials"] == rr["trace_trials"]
        assert (rg["iterations"], rg["stop_reason"], rg["n_bbox_dropped"]) == (rr["iterations"], rr["stop_reason"], rr["n_bbox_dropped"])
        np.testing.assert_array_equal(og, oo)


def test_sharded_run_with_an_empty_shard(pkg, ctx):
    """3 shards, the last one holds no ellipsoid: it must still take part in every collective and report the same run."""
    import threading
    g, c, o, _ = pkg.synth.make_graph(24, 2, 150, seed=17)
    p = pkg.default_lm_params(jacobian_mode=1)
    ctx.upload_graph(g); ctx.upload_states(c, o)
    ref = ctx.optimize_resident(p)
    _, ro = ctx.download_states()
    n = 3
    part = np.array([0, 1])            # ellipsoid 0 -> shard 0, ellipsoid 1 -> shard 1, shard 2 empty
    ar = ThreadAllreduce(n)
    ctxs, reps, errs = [], [None] * n, [None] * n
    for r in range(n):
        idx = np.nonzero(part == r)[0]
        cx = pkg.Context(0)
        cx.upload_graph(g.subset_objects(idx)); cx.upload_states(c, o[idx])
        cx.comm_init_host(n, r, ar.make(r))
        ctxs.append((cx, idx))

    def work(r):
        try:
            reps[r] = ctxs[r][0].optimize_resident(p)
        except Exception as e:
            errs[r] = e
            ar.bar.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    [t.start() for t in th]
    [t.join(120) for t in th]
    assert errs == [None] * n, errs
    out = np.zeros_like(o)
    for cx, idx in ctxs:
        _, oo = cx.download_states()
        out[idx] = oo[:len(idx)]
        cx.comm_destroy(); cx.close()
    for rep in reps:
        assert rep["iterations"] == ref["iterations"] and rep["trace_trials"] == ref["trace_trials"]
        np.testing.assert_allclose(rep["trace_chi2"], ref["trace_chi2"], rtol=1e-12)
    np.testing.assert_allclose(out, ro, rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize("seed", range(12))
def test_random_graph_sweep_matches_oracle(pkg, po, ctx, seed):
    """A sweep over graph shapes (3-80 cameras, 1-12 ellipsoids, with / without 3-D and gravity edges, both Jacobian modes):
    same LM trajectory as the CPU restatement and final ellipsoids within the north-star tolerance."""
    rng = np.random.default_rng(1000 + seed)
    n_cams, n_objs = int(rng.integers(3, 80)), int(rng.integers(1, 13))
    n_edges = int(rng.integers(n_objs * 3, n_objs * 60))
    g, c, o, _ = pkg.synth.make_graph(n_cams, n_objs, n_edges, seed=seed, gravity=bool(seed % 3))
    if seed % 4 == 0:   # bbox edges only
        g = pkg.Graph(g.K, g.n_cams, g.n_objs, None, g.bbox_cam, g.bbox_obj, g.bbox_meas, g.bbox_weight,
                      grav_obj=g.grav_obj, grav_normal=g.grav_normal, grav_weight=g.grav_weight)
    jac = seed % 2
    p = pkg.default_lm_params(jacobian_mode=jac, numeric_delta=1e-6)
    _, oo, ro = po.optimize(g, c, o, pkg.default_lm_params(numeric_delta=1e-6), solver=1)
    _, og, rg = ctx.optimize(g, c, o, p)
    assert rg["n_bbox_dropped"] == ro["n_bbox_dropped"]
    assert_traces_match(rg, ro, rtol=1e-5)
    assert group_rel_err(og, oo) < 1e-4   # |dt| / |t|, |dq|, |ds| / |s| per ellipsoid


def test_one_ellipsoid_with_many_chunks(pkg, po, ctx):
    """An ellipsoid seen 1,500 times: 24 bbox chunks + dozens of 3-D chunks on ONE vertex (k_lm_step's gather walks more than
    its four batched chunks, the chunk partials of one 9x9 system span many workgroups)."""
    g, c, o, _ = pkg.synth.make_graph(1500, 1, 3000, seed=41)
    assert np.bincount(g.bbox_obj).max() > 64 * 4
    for jac in (1, 0):
        p = pkg.default_lm_params(jacobian_mode=jac, numeric_delta=1e-6)
        _, oo, ro = po.optimize(g, c, o, pkg.default_lm_params(numeric_delta=1e-6), solver=1)
        _, og, rg = ctx.optimize(g, c, o, p)
        assert_traces_match(rg, ro, rtol=1e-5)
        assert group_rel_err(og, oo) < 1e-4
