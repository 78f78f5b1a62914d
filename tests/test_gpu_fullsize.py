"""BASELINE.json full-size configuration (configs[3]: 10k cams / 2k ellipsoids / ~200k bbox edges) through
size-independent properties — the oracle would need minutes on the whole graph, so:
  * the linearisation is independent per ellipsoid in mapping mode: H_oo / b_o of a random SAMPLE of ellipsoids
    must equal the oracle's linearisation of the sub-graph holding only those ellipsoids;
  * permuting the edge arrays of the input changes nothing (edges are re-sorted by ellipsoid at upload);
  * two runs are bit-identical (all reductions are fixed-order);
  * analytic and numeric (delta 1e-6) Jacobians reach the same minimum; chi2 never increases across iterations;
  * the sum of the ellipsoid-sharded sub-problems' chi2 equals the chi2 of the whole (sharding invariance)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c4(pkg):
    return pkg.synth.make_config("C4", seed=0)


def unpack45(Hp):
    H = np.zeros((9, 9))
    H[np.triu_indices(9)] = Hp
    return H + np.triu(H, 1).T


def test_c4_sampled_linearisation_matches_oracle(pkg, po, ctx, c4):
    g, c, o, _ = c4
    assert g.n_cams == 10000 and g.n_objs == 2000 and len(g.bbox_cam) == 200000
    ctx.upload_graph(g); ctx.upload_states(c, o)
    ctx.lm_begin(pkg.default_lm_params(jacobian_mode=1))
    part = ctx.lm_linearize()
    Hg = ctx.lm_download(0, g.n_objs * 45).reshape(-1, 45)
    bg = ctx.lm_download(1, g.n_objs * 9).reshape(-1, 9)
    ids = np.sort(np.random.default_rng(0).choice(g.n_objs, size=32, replace=False))
    sub = g.subset_objects(ids)
    H, b, fidx, chi = po.build_system(sub, c, o[ids], delta=1e-6)
    for k, oid in enumerate(ids):
        i = fidx[g.n_cams + k]
        np.testing.assert_allclose(unpack45(Hg[oid]), H[i:i + 9, i:i + 9], atol=3e-6 * np.abs(H[i:i + 9, i:i + 9]).max())
        np.testing.assert_allclose(bg[oid], b[i:i + 9], atol=3e-6 * np.abs(b[i:i + 9]).max())
    assert part.max_diag >= np.abs(np.diag(H)).max() * (1 - 1e-9)


def test_c4_deterministic_permutation_invariant_and_monotone(pkg, ctx, c4):
    g, c, o, _ = c4
    p = pkg.default_lm_params(jacobian_mode=1)
    _, o1, r1 = ctx.optimize(g, c, o, p)
    _, o2, r2 = ctx.optimize(g, c, o, p)
    assert np.array_equal(o1, o2) and r1["trace_chi2"] == r2["trace_chi2"]          # bit-identical reruns
    tr = [r1["chi2_initial"]] + r1["trace_chi2"]
    assert all(tr[k + 1] <= tr[k] * (1 + 1e-12) for k in range(len(tr) - 1))          # LM never accepts an increase
    assert r1["chi2_final"] < 0.1 * r1["chi2_initial"]
    rng = np.random.default_rng(1)
    pb, pe = rng.permutation(len(g.bbox_cam)), rng.permutation(len(g.e3d_cam))
    gp = pkg.Graph(g.K, g.n_cams, g.n_objs, None, g.bbox_cam[pb], g.bbox_obj[pb], g.bbox_meas.reshape(-1, 4)[pb], g.bbox_weight[pb],
                   g.e3d_cam[pe], g.e3d_obj[pe], g.e3d_meas.reshape(-1, 10)[pe], g.e3d_weight[pe], g.grav_obj, g.grav_normal,
                   g.grav_weight)
    _, o3, r3 = ctx.optimize(gp, c, o, p)
    assert r3["chi2_final"] == pytest.approx(r1["chi2_final"], rel=1e-9)
    np.testing.assert_allclose(o3, o1, atol=1e-8)


def test_c4_numeric_and_analytic_reach_the_same_minimum(pkg, ctx, c4):
    g, c, o, _ = c4
    _, oa, ra = ctx.optimize(g, c, o, pkg.default_lm_params(jacobian_mode=1))
    _, on, rn = ctx.optimize(g, c, o, pkg.default_lm_params(jacobian_mode=0, numeric_delta=1e-6))
    assert rn["chi2_final"] == pytest.approx(ra["chi2_final"], rel=1e-7)
    np.testing.assert_allclose(on, oa, atol=1e-5)


def test_c4_sharding_invariance_of_chi2(pkg, ctx, c4):
    """chi2 of the whole graph = sum over the ellipsoid shards (what the multi-GPU exchange adds up)."""
    g, c, o, _ = c4
    ctx.upload_graph(g); ctx.upload_states(c, o)
    ctx.lm_begin(pkg.default_lm_params(jacobian_mode=1))
    whole = ctx.lm_linearize()
    part = pkg.lib.partition_objects(g, 4)
    tot, md = 0.0, 0.0
    for r in range(4):
        ids = np.nonzero(part == r)[0]
        ctx.upload_graph(g.subset_objects(ids)); ctx.upload_states(c, o[ids])
        ctx.lm_begin(pkg.default_lm_params(jacobian_mode=1))
        s = ctx.lm_linearize()
        tot += s.chi2; md = max(md, s.max_diag)
    assert tot == pytest.approx(whole.chi2, rel=1e-12)
    assert md == whole.max_diag
