"""SLAM mode (free cameras, odometry, Schur + dense FP64-MFMA Cholesky) against the CPU restatement.

The reference ships this branch switched off (bSLAM_mode = false, Optimizer.cpp:126); parity is
"what g2o would do with that graph": the oracle's dense LDLT of the whole system (ORACLE_DENSE)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def lower_to_full(S, n, lda):
    M = S.reshape(n, lda).T  # column-major (lda x n) -> M[row, col]
    L = M[:n, :n]
    full = np.tril(L) + np.tril(L, -1).T
    return full, M[n, :n].copy()


def cam_err(a, b):
    from oracle import np_oracle as npo
    e = 0.0
    for x, y in zip(a, b):
        d = npo.se3_log(npo.T_inv(npo.T_from7(x)) @ npo.T_from7(y))
        e = max(e, np.linalg.norm(d))
    return e


@pytest.mark.parametrize("jac,delta,tol", [(0, 1e-6, 5e-6), (1, 1e-6, 5e-6)])
def test_slam_linearisation_matches_oracle(pkg, po, ctx, jac, delta, tol):
    g, c, o, _ = pkg.synth.make_graph(30, 6, 200, seed=3, slam=True)
    H, b, fidx, chi = po.build_system(g, c, o, delta=1e-6)
    ctx.upload_graph(g); ctx.upload_states(c, o)
    ctx.lm_begin(pkg.default_lm_params(jacobian_mode=jac, numeric_delta=delta))
    part = ctx.lm_linearize()
    assert part.chi2 == pytest.approx(chi, rel=1e-9)
    nf = int((~g.cam_fixed.astype(bool)).sum())
    Hcc = ctx.lm_download(3, nf * 36).reshape(nf, 6, 6)
    bc = ctx.lm_download(4, nf * 6).reshape(nf, 6)
    free = [i for i in range(g.n_cams) if not g.cam_fixed[i]]
    for s, ci in enumerate(free):
        i = fidx[ci]
        assert i == 6 * s
        np.testing.assert_allclose(Hcc[s], H[i:i + 6, i:i + 6], atol=tol * np.abs(H[i:i + 6, i:i + 6]).max())
        np.testing.assert_allclose(bc[s], b[i:i + 6], atol=tol * max(np.abs(b[i:i + 6]).max(), 1.0))
    assert part.max_diag == pytest.approx(np.abs(np.diag(H)).max(), rel=1e-5)
    # reduced system vs a numpy Schur complement of the oracle's dense H
    lam = 1e-5 * part.max_diag
    ptr, n, lda = ctx.lm_reduced_system(lam)
    assert n == 6 * nf
    S, bs = lower_to_full(ctx.lm_download(6, lda * n), n, lda)
    Hl = H + lam * np.eye(len(b))
    Hpp, Hpl, Hll = Hl[:n, :n], Hl[:n, n:], Hl[n:, n:]
    S_ref = Hpp - Hpl @ np.linalg.solve(Hll, Hpl.T)
    bs_ref = b[:n] - Hpl @ np.linalg.solve(Hll, b[n:])
    np.testing.assert_allclose(S, S_ref, atol=tol * np.abs(S_ref).max())
    np.testing.assert_allclose(bs, bs_ref, atol=tol * np.abs(bs_ref).max())


@pytest.mark.parametrize("solver", [1, 2])
@pytest.mark.parametrize("n_cams", [12, 40, 100])
def test_dense_cholesky_solves_reduced_system(pkg, ctx, n_cams, solver):
    """x_c equals numpy's solve of the reduced camera system S, b_s (1, 2 and 5 panels of 128) -- from the MFMA Cholesky of S itself
    (solver 1) and from the camera-first elimination (solver 2: block-bidiagonal factor of the camera block, reduced ELLIPSOID
    system, back-substitution), which must solve the same linear system."""
    g, c, o, _ = pkg.synth.make_graph(n_cams, 8, 40 * n_cams // 4, seed=4, slam=True)
    ctx.upload_graph(g); ctx.upload_states(c, o)
    ctx.lm_begin(pkg.default_lm_params(jacobian_mode=1, linear_solver=solver))
    part = ctx.lm_linearize()
    lam = 1e-5 * part.max_diag
    ptr, n, lda = ctx.lm_reduced_system(lam)
    S, bs = lower_to_full(ctx.lm_download(6, lda * n), n, lda)
    out = ctx.lm_try_step(lam)
    assert out.solve_ok == 1 and ctx.lm_solver_used() == solver
    xc = ctx.lm_download(5, n)
    ref = np.linalg.solve(S, bs)
    np.testing.assert_allclose(xc, ref, atol=1e-9 * np.abs(ref).max() + 1e-12, rtol=1e-7)
    assert ctx.lm_reduced_residual() < 1e-12
    ctx.lm_commit(False)


@pytest.mark.parametrize("solver", [1, 2])
@pytest.mark.parametrize("jac", [0, 1])
def test_slam_lm_matches_faithful_dense_oracle(pkg, po, ctx, jac, solver):
    g, c, o, _ = pkg.synth.make_graph(30, 6, 200, seed=3, slam=True)
    p = pkg.default_lm_params(numeric_delta=1e-6)
    co, oo, ro = po.optimize(g, c, o, p, solver=0)  # dense LDLT of the whole system, like LinearSolverDense
    cg, og, rg = ctx.optimize(g, c, o, pkg.default_lm_params(jacobian_mode=jac, numeric_delta=1e-6, linear_solver=solver))
    assert ctx.lm_solver_used() == solver
    n = min(len(rg["trace_chi2"]), len(ro["trace_chi2"]))
    print("SLAM LM (30 cams) jac %d solver %d vs dense checker: chi2 trace rel %.2e, cams %.2e, centres %.2e, scales rel %.2e" % (
        jac, solver, float(np.abs(np.array(rg["trace_chi2"][:n]) / np.array(ro["trace_chi2"][:n]) - 1).max()), cam_err(cg, co),
        float(np.abs(og[:, :3] - oo[:, :3]).max()), float(np.abs(og[:, 7:] / oo[:, 7:] - 1).max())))
    # measured (both Jacobian modes: at delta = 1e-6 the checker's differences are good to 1e-7): objective 1.1e-7, cameras 1.4e-6,
    # centres 4.1e-7, scales 1.2e-7
    np.testing.assert_allclose(rg["trace_chi2"][:n], ro["trace_chi2"][:n], rtol=5e-7)
    assert rg["chi2_final"] == pytest.approx(ro["chi2_final"], rel=5e-7)
    assert rg["trace_trials"][:n] == ro["trace_trials"][:n]
    assert np.array_equal(cg[0], c[0])  # camera 0 is fixed (Optimizer.cpp:138)
    assert cam_err(cg, co) < 5e-6
    np.testing.assert_allclose(og[:, :3], oo[:, :3], atol=2e-6)
    np.testing.assert_allclose(og[:, 7:], oo[:, 7:], rtol=1e-6)


def obj_rel(a, b):
    return float((np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)).max())


@pytest.fixture(scope="module")
def c3_dense(pkg, po):
    """BASELINE.json configs[2] in SLAM mode (n = 6*499 + 9*50 = 3,444 unknowns) through the FAITHFUL restatement: g2o's
    LinearSolverDense = pivoted LDLT of the whole system every trial (solvers/linear_solver_dense.h:65-113), numeric
    Jacobians; once at the reference's delta = 1e-9 and once at 1e-6 (~30 s each on one host core)."""
    g, c, o, _ = pkg.synth.make_config("C3", seed=0, slam=True)
    runs = {d: po.optimize(g, c, o, pkg.default_lm_params(numeric_delta=d), solver=po.ORACLE_DENSE) for d in (1e-9, 1e-6)}
    return g, c, o, runs


def test_c3_slam_first_iteration_matches_faithful_dense_oracle(pkg, po, ctx, c3_dense):
    """One LM iteration from the same start pins the SOLVER: Schur complement + FP64-MFMA Cholesky + back-substitution
    must give the step of the dense pivoted LDLT of the full 3,444 x 3,444 system.  Same Jacobian scheme on both sides
    (numeric, delta = 1e-6) -> cameras, ellipsoid poses AND scales to 1e-6 (north star: 1e-4)."""
    g, c, o, _ = c3_dense
    p1 = pkg.default_lm_params(numeric_delta=1e-6, max_iters=1)
    co, oo, ro = po.optimize(g, c, o, p1, solver=po.ORACLE_DENSE)
    for jac, solver in ((0, 1), (1, 1), (0, 2), (1, 2)):   # both eliminations: reduced camera system / cameras first
        cg, og, rg = ctx.optimize(g, c, o, pkg.default_lm_params(jacobian_mode=jac, numeric_delta=1e-6, max_iters=1, linear_solver=solver))
        assert ctx.lm_solver_used() == solver
        assert rg["trace_trials"] == ro["trace_trials"]
        assert rg["chi2_final"] == pytest.approx(ro["chi2_final"], rel=1e-7)
        assert cam_err(cg, co) < 1e-6
        assert obj_rel(og, oo) < 1e-6
        np.testing.assert_allclose(og[:, 7:], oo[:, 7:], rtol=1e-6)


def test_c3_slam_full_run_vs_faithful_dense_delta_1e9(pkg, po, ctx, c3_dense):
    """The whole optimize(10) against the reference's scheme (dense LDLT, delta = 1e-9).
    What CAN agree to 1e-4 does: the objective after every iteration (the LM trajectory) to 2e-5 relative, the accept /
    reject sequence exactly.  The final STATES of this graph are not reproducible to 1e-4 by the reference itself: g2o
    stops after 5 iterations on the ORB-SLAM rule (chi2 moved < 0.1 %), short of convergence along the weakly
    constrained camera directions, and its own result moves by ~6e-2 (camera log-distance) and ~1.4e-4 (ellipsoids)
    when nothing but the differentiation step changes from 1e-9 to 1e-6 (measured below, both runs of the restatement).
    The bar used here: the GPU must be as close to the reference (delta = 1e-9) as the reference is to itself under
    that change, with a factor 2 of slack — and to 1e-4 on the ellipsoids when compared like for like (delta 1e-6)."""
    g, c, o, runs = c3_dense
    (c9, o9, r9), (c6, o6, r6) = runs[1e-9], runs[1e-6]
    floor_cam, floor_obj = cam_err(c6, c9), obj_rel(o6, o9)
    cg, og, rg = ctx.optimize(g, c, o, pkg.default_lm_params(jacobian_mode=1))           # product default: analytic, solver AUTO
    assert ctx.lm_solver_used() == 2             # C3: 450 ellipsoid unknowns against 2,994 camera unknowns -> cameras first
    assert rg["trace_trials"] == r9["trace_trials"] and rg["stop_reason"] == r9["stop_reason"]
    np.testing.assert_allclose(rg["trace_chi2"], r9["trace_chi2"], rtol=2e-5)
    assert rg["chi2_final"] < 0.1 * rg["chi2_initial"]
    assert cam_err(cg, c9) < 2 * floor_cam + 1e-4, (cam_err(cg, c9), floor_cam)
    assert obj_rel(og, o9) < 2 * floor_obj + 1e-4, (obj_rel(og, o9), floor_obj)
    # like for like: numeric delta = 1e-6 on both sides
    cn, on, rn = ctx.optimize(g, c, o, pkg.default_lm_params(jacobian_mode=0, numeric_delta=1e-6))
    np.testing.assert_allclose(rn["trace_chi2"], r6["trace_chi2"], rtol=1e-6)
    # the two eliminations against each other over the whole run (same linear systems, different arithmetic: cond(S) amplifies
    # the rounding of either): like for like, reduced camera system forced
    c1, o1, r1 = ctx.optimize(g, c, o, pkg.default_lm_params(jacobian_mode=0, numeric_delta=1e-6, linear_solver=1))
    assert ctx.lm_solver_used() == 1 and r1["trace_trials"] == rn["trace_trials"]
    print("C3 SLAM: cameras-first vs reduced-camera elimination: cams %.2e objs %.2e chi2 rel %.2e" % (
        cam_err(cn, c1), obj_rel(on, o1), float(np.abs(np.array(rn["trace_chi2"]) / np.array(r1["trace_chi2"]) - 1).max())))
    assert cam_err(c1, c6) < 1e-4 and obj_rel(o1, o6) < 1e-5
    print("C3 SLAM parity: reference self-distance (delta 1e-9 vs 1e-6) cams %.2e objs %.2e | GPU analytic vs 1e-9: cams %.2e objs %.2e | "
          "GPU numeric 1e-6 vs oracle 1e-6: cams %.2e objs %.2e" % (floor_cam, floor_obj, cam_err(cg, c9), obj_rel(og, o9), cam_err(cn, c6), obj_rel(on, o6)))
    assert obj_rel(on, o6) < 1e-5          # measured 8.6e-7
    assert cam_err(cn, c6) < 1e-4          # measured 1.0e-5 (north star: 1e-4)
    # the product default (analytic Jacobians) against the checker at delta = 1e-6, whose central differences are good to ~1e-7
    print("C3 SLAM parity: GPU analytic vs oracle 1e-6: cams %.2e objs %.2e, chi2 trace rel %.2e" % (
        cam_err(cg, c6), obj_rel(og, o6), float(np.abs(np.array(rg["trace_chi2"]) / np.array(r6["trace_chi2"]) - 1).max())))
    assert cam_err(cg, c6) < 1e-4 and obj_rel(og, o6) < 1e-4


@pytest.mark.parametrize("n", [1, 7, 130, 777, 3000, 8192, 8704, 9001, 18000])
def test_dense_cholesky_selftest_residual(ctx, n):
    """Known-answer test of the MFMA Cholesky on a generated diagonally dominant system (sizes straddle the 128-wide
    inner panels and the outer panels of 2 / 4 of them; from 8,192 on the factorisation runs with the look-ahead split on two
    streams, rectangular head updates and both tile sizes -- 8,704: the last outer panel is full and only the right-hand side's row
    is below it; 18,000: the reduced ellipsoid system of C4): |A x - b| / |b| at fp64 round-off."""
    ms, res = ctx.selftest_cholesky(n)
    print("dense Cholesky self test n = %d: %.3f ms, residual %.2e" % (n, ms, res))
    assert res < 1e-12, (n, res)
    assert ms >= 0


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 17, 64, 127, 128, 129, 180, 181, 182, 183, 247, 248])
def test_small_dense_cholesky_one_workgroup(pkg, monkeypatch, n):
    """k_chol_small (round 6): orders <= 248 are factored and solved by ONE workgroup with the lower triangle in registers (rank-4 steps
    over all block columns, y = L^-1 b as a block row of its own, L^T x = y on the same registers).  Residual at round-off at sizes
    around the 4-wide blocks, the second block per thread (from 181 on) and the limit; the blocked path (ESL_CHOL_SMALL=0: potrf2 +
    panel + update launches) gives the same x to round-off on the same generated system (their residuals agree in magnitude, and both
    are checked here -- the blocked path has no other test at these sizes now)."""
    cx = pkg.Context(0)
    try:
        ms, res = cx.selftest_cholesky(n)
    finally:
        cx.close()
    monkeypatch.setenv("ESL_CHOL_SMALL", "0")
    cx = pkg.Context(0)
    try:
        ms0, res0 = cx.selftest_cholesky(n)
    finally:
        cx.close()
    print("one-workgroup solve n = %d: %.3f ms, residual %.2e | blocked path %.3f ms, residual %.2e" % (n, ms, res, ms0, res0))
    assert res < 5e-15 and res0 < 5e-15, (n, res, res0)


@pytest.mark.parametrize("n", [1, 7, 129, 130, 257, 777, 1153, 3000, 4097, 5000, 8192, 9001, 18000])
def test_persistent_dense_cholesky_selftest_residual(pkg, monkeypatch, n):
    """the persistent form of the dense factorisation (k_chol_persist: ONE launch, workgroup 0 walks the diagonal blocks, the others
    pull tasks off a static list, dependency words instead of ~1,000 launches and stream events; the CPU replay of its task list is
    tests/test_chol_plan.py) forced on at every size, against the same generated SPD systems as the launch-per-step form: residual
    at round-off, and the same x twice (the residual's own sum is an atomic reduction: its last bits vary, a different x would move
    it by a sizeable fraction of itself)"""
    monkeypatch.setenv("ESL_CHOL_PERSISTENT", "1")
    monkeypatch.setenv("ESL_CHOL_DEBUG", "1")     # the self test creates and destroys a stream while the kernel runs (what broke the two-kernel form)
    cx = pkg.Context(0)
    try:
        ms, res = cx.selftest_cholesky(n)
        ms2, res2 = cx.selftest_cholesky(n)
    finally:
        cx.close()
        monkeypatch.delenv("ESL_CHOL_PERSISTENT", raising=False)
        monkeypatch.delenv("ESL_CHOL_DEBUG", raising=False)
    print("persistent Cholesky n = %d: %.3f ms (second call %.3f ms), residual %.1e" % (n, ms, ms2, res))
    assert res < 1e-13 and abs(res2 - res) <= 1e-9 * res


def test_launch_path_update_tiles_agree(pkg):
    """The launch-per-step factorisation (ESL_CHOL_PERSISTENT=0) with the rank-K update in its two tile shapes -- k_chol_update_v (round 6:
    128 x 128, four waves of 32 x 128, two workgroups per CU, the default) and k_chol_update_lds<256, 128> (ESL_UPD_V=0) -- on the same
    generated systems: both at round-off, and the same x (the two kernels accumulate every entry in the same order: bit-identical C,
    scripts/debug/upd_v_probe.hip; the residual's own sum is an atomic reduction, so it is compared to 1e-9 of itself).  The switch is read
    once per process: one subprocess per setting."""
    import subprocess, sys, os, json
    code = ("import importlib, json, sys; sys.path.insert(0, %r); pkg = importlib.import_module('object-oriented-slam_amd'); cx = pkg.Context(0); "
            "print(json.dumps([cx.selftest_cholesky(n) for n in (9001, 12000)])); cx.close()") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for v in ("1", "0"):
        env = dict(os.environ, ESL_CHOL_PERSISTENT="0", ESL_UPD_V=v)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[v] = json.loads(r.stdout.strip().splitlines()[-1])
    print("launch path, update tile v / lds: (ms, residual) per size:", out["1"], out["0"])
    for (ms1, res1), (ms0, res0) in zip(out["1"], out["0"]):
        assert res1 < 1e-13 and res0 < 1e-13
        assert abs(res1 - res0) <= 1e-9 * res0


@pytest.mark.parametrize("n", [130, 1153, 4097, 9001])
def test_persistent_cholesky_without_the_fused_chain(pkg, monkeypatch, n):
    """ESL_CHOL_FUSE=0: the strips under the diagonal block and the next block's update as worker tasks (the first form of round 4,
    kept as the A/B switch of the fused chain)"""
    monkeypatch.setenv("ESL_CHOL_PERSISTENT", "1")
    monkeypatch.setenv("ESL_CHOL_FUSE", "0")
    cx = pkg.Context(0)
    try:
        ms, res = cx.selftest_cholesky(n)
    finally:
        cx.close()
    assert res < 1e-13


def test_slam_runs_are_bitwise_reproducible(pkg, ctx):
    """SLAM mode has no order-dependent reduction left (round 1's Schur complement scattered into S with fp64 atomics): the
    same graph twice gives the same bits, trace and states."""
    g, c, o, _ = pkg.synth.make_graph(60, 12, 700, seed=17, slam=True)
    for solver in (1, 2):
        p = pkg.default_lm_params(jacobian_mode=1, linear_solver=solver)
        runs = [ctx.optimize(g, c, o, p) for _ in range(3)]
        for cc, oo, rep in runs[1:]:
            assert rep["trace_chi2"] == runs[0][2]["trace_chi2"]
            np.testing.assert_array_equal(cc, runs[0][0])
            np.testing.assert_array_equal(oo, runs[0][1])


def test_camera_first_elimination_applicability(pkg, ctx):
    """ESL_SOLVER_REDUCED_ELLIPSOID needs the reference's chain (odometry edges between neighbouring free cameras only):
    a loop-closure style odometry edge makes the camera block non-tridiagonal -> the forced solver is refused (ESL_ERR_INVALID),
    AUTO falls back to the reduced camera system and still matches the run without that edge's special treatment; graphs with
    fixed cameras inside the chain, without any odometry edge, and with a camera that has odometry edges only, are served."""
    g, c, o, _ = pkg.synth.make_graph(24, 5, 160, seed=6, slam=True)
    from oracle import pyoracle as po
    # (a) an extra odometry edge 3 -> 11
    Z = po.se3_mul(c[11], po.se3_inv(c[3]))
    ga = pkg.Graph(g.K, g.n_cams, g.n_objs, g.cam_fixed, g.bbox_cam, g.bbox_obj, g.bbox_meas, g.bbox_weight, g.e3d_cam, g.e3d_obj, g.e3d_meas,
                   g.e3d_weight, g.grav_obj, g.grav_normal, g.grav_weight, odom_i=np.append(g.odom_i, 3), odom_j=np.append(g.odom_j, 11),
                   odom_meas=np.vstack([g.odom_meas.reshape(-1, 7), Z]))
    with pytest.raises(pkg.EslError, match="esl_status 2"):
        ctx.optimize(ga, c, o, pkg.default_lm_params(jacobian_mode=1, linear_solver=2))
    ca, oa, ra = ctx.optimize(ga, c, o, pkg.default_lm_params(jacobian_mode=1, numeric_delta=1e-6))
    assert ctx.lm_solver_used() == 1
    co, oo, ro = po.optimize(ga, c, o, pkg.default_lm_params(numeric_delta=1e-6), solver=0)
    assert ra["trace_trials"][:2] == ro["trace_trials"][:2] and cam_err(ca, co) < 1e-4
    # (b) cameras 0 and 9 fixed (the chain is cut in two), (c) no odometry at all, (d) reversed vertex order of some odometry edges
    fixed = g.cam_fixed.copy(); fixed[9] = 1
    oi, oj, om = g.odom_i.copy(), g.odom_j.copy(), g.odom_meas.reshape(-1, 7).copy()
    for k in (2, 5, 17):
        oi[k], oj[k] = oj[k], oi[k]
        om[k] = po.se3_inv(om[k])
    variants = {
        "cut": pkg.Graph(g.K, g.n_cams, g.n_objs, fixed, g.bbox_cam, g.bbox_obj, g.bbox_meas, g.bbox_weight, g.e3d_cam, g.e3d_obj, g.e3d_meas,
                         g.e3d_weight, g.grav_obj, g.grav_normal, g.grav_weight, odom_i=g.odom_i, odom_j=g.odom_j, odom_meas=g.odom_meas),
        "no_odometry": pkg.Graph(g.K, g.n_cams, g.n_objs, g.cam_fixed, g.bbox_cam, g.bbox_obj, g.bbox_meas, g.bbox_weight, g.e3d_cam, g.e3d_obj,
                                 g.e3d_meas, g.e3d_weight, g.grav_obj, g.grav_normal, g.grav_weight),
        "reversed_edges": pkg.Graph(g.K, g.n_cams, g.n_objs, g.cam_fixed, g.bbox_cam, g.bbox_obj, g.bbox_meas, g.bbox_weight, g.e3d_cam, g.e3d_obj,
                                    g.e3d_meas, g.e3d_weight, g.grav_obj, g.grav_normal, g.grav_weight, odom_i=oi, odom_j=oj, odom_meas=om),
    }
    # This 24-camera graph is poorly conditioned (cond(S) ~ 1e3 on top of the 1e-7 by which two implementations' central differences
    # differ: x of the first step is reproducible to 3e-5 only -- measured, scripts/debug/cut_graph.py).  So: the two eliminations
    # against EACH OTHER tightly (they solve the same linear systems), each against the checker at the north star's 1e-4.
    for name, gv in variants.items():
        p6 = pkg.default_lm_params(numeric_delta=1e-6, max_iters=2)
        co, oo, ro = po.optimize(gv, c, o, p6, solver=0)
        runs = {}
        for solver in (1, 2):
            cg, og, rg = ctx.optimize(gv, c, o, pkg.default_lm_params(jacobian_mode=0, numeric_delta=1e-6, max_iters=2, linear_solver=solver))
            assert ctx.lm_solver_used() == solver
            assert rg["trace_trials"] == ro["trace_trials"], (name, solver)
            # against the checker: the first iteration (from the second on the difference of the first -- 3e-5 -- is amplified again:
            # "no_odometry" leaves the cameras without any camera-camera constraint and reaches 1.04e-4 on chi2 after two)
            assert rg["trace_chi2"][0] == pytest.approx(ro["trace_chi2"][0], rel=1e-4), (name, solver)
            np.testing.assert_allclose(rg["trace_chi2"], ro["trace_chi2"], rtol=5e-4, err_msg=name)
            assert cam_err(cg, co) < 5e-4 and obj_rel(og, oo) < 5e-4, (name, solver, cam_err(cg, co), obj_rel(og, oo))
            runs[solver] = (cg, og, rg)
        print("variant %s: eliminations vs each other: chi2 rel %.2e cams %.2e objs %.2e" % (
            name, float(np.abs(np.array(runs[1][2]["trace_chi2"]) / np.array(runs[2][2]["trace_chi2"]) - 1).max()), cam_err(runs[1][0], runs[2][0]),
            obj_rel(runs[1][1], runs[2][1])))
        np.testing.assert_allclose(runs[1][2]["trace_chi2"], runs[2][2]["trace_chi2"], rtol=1e-6, err_msg=name)
        assert cam_err(runs[1][0], runs[2][0]) < 1e-6 and obj_rel(runs[1][1], runs[2][1]) < 1e-6


@pytest.mark.parametrize("n_cams", [129, 161, 200, 500])
def test_nested_dissection_equals_plain_chain(pkg, monkeypatch, n_cams):
    """Camera-first elimination with the camera chain cut into segments (nested dissection, from 128 free cameras on: every
    stride-th camera a separator, segments factored / substituted in parallel, the separators' own block-tridiagonal system)
    against the same elimination over the plain chain (ESL_CF_NO_ND=1) and against the reduced camera system: the same linear
    systems, so x of the first trial to rounding and the whole LM run alike.  Sizes: 128 free cameras (the threshold, 8 segments
    of stride 16), 160 (the last camera IS a separator), 199 (short last segment), 499 (C3's count, stride 16)."""
    g, c, o, _ = pkg.synth.make_graph(n_cams, 12, 12 * n_cams, seed=31, slam=True)
    nf = n_cams - 1
    out = {}
    for tag, env, solver in (("nd", None, 2), ("plain", "1", 2), ("camera", None, 1)):
        if env:
            monkeypatch.setenv("ESL_CF_NO_ND", env)
        else:
            monkeypatch.delenv("ESL_CF_NO_ND", raising=False)
        cx = pkg.Context(0)
        try:
            cx.upload_graph(g); cx.upload_states(c, o)
            cx.lm_begin(pkg.default_lm_params(jacobian_mode=1, linear_solver=solver))
            part = cx.lm_linearize()
            lam = 1e-5 * part.max_diag
            tr = cx.lm_try_step(lam)
            assert tr.solve_ok == 1 and cx.lm_solver_used() == solver
            xc, xo = cx.lm_download(5, 6 * nf), cx.lm_download(2, 9 * g.n_objs)
            res = cx.lm_reduced_residual()
            cx.lm_commit(False)
            cc, oo, rep = cx.optimize(g, c, o, pkg.default_lm_params(jacobian_mode=1, linear_solver=solver))
            out[tag] = (xc, xo, tr.chi2, res, cc, oo, rep)
        finally:
            cx.close()
    monkeypatch.delenv("ESL_CF_NO_ND", raising=False)
    ref = out["plain"]
    for tag in ("nd", "camera"):
        xc, xo, chi, res, cc, oo, rep = out[tag]
        d_xc = float(np.abs(xc - ref[0]).max() / np.abs(ref[0]).max()); d_xo = float(np.abs(xo - ref[1]).max() / np.abs(ref[1]).max())
        print("n_cams %d, %s vs plain chain: x_c %.2e x_o %.2e, |Sx-b|/|b| %.2e, run: chi2 rel %.2e cams %.2e" % (
            n_cams, tag, d_xc, d_xo, res, abs(rep["chi2_final"] / ref[6]["chi2_final"] - 1), cam_err(cc, ref[4])))
        # measured: x 1e-14 .. 5e-13, residual 5e-16, whole run chi2 1e-14, cameras 1e-11
        assert res < 1e-13 and d_xc < 1e-10 and d_xo < 1e-10
        assert chi == pytest.approx(ref[2], rel=1e-12)
        assert rep["trace_trials"] == ref[6]["trace_trials"]
        np.testing.assert_allclose(rep["trace_chi2"], ref[6]["trace_chi2"], rtol=1e-11)
        assert cam_err(cc, ref[4]) < 1e-9 and obj_rel(oo, ref[5]) < 1e-9


@pytest.mark.parametrize("n_cams,n_objs,per_cam", [(257, 40, 6), (300, 80, 10), (500, 50, 10), (1000, 300, 8)])
def test_sparse_interior_rows_equal_dense_rows(pkg, monkeypatch, n_cams, n_objs, per_cam):
    """Camera-first elimination with X = G^-1 W kept SPARSE (esl_cf.hpp: segments of 16 slots, compact slabs of the columns a
    segment's cameras see, T assembled block by block from the segments two ellipsoids share -- from the stored per-segment
    products (ESL_CF_SPARSE=1) or straight from the slabs (=2) --, the separators' dense rows on the MFMA update) against the dense X of the same elimination (ESL_CF_SPARSE=0) and the reduced camera system: one linear system,
    so x of the first trial to rounding and the LM run alike.  Sizes: 256 free cameras (the last one IS a separator), 299 (short
    last segment), 499 (C3's count), 999 with 300 ellipsoids (most columns dead in most segments)."""
    g, c, o, _ = pkg.synth.make_graph(n_cams, n_objs, per_cam * n_cams, seed=41, slam=True)
    nf = n_cams - 1
    out = {}
    for tag, env, solver in (("sparse", "1", 2), ("sparse_direct", "2", 2), ("dense", "0", 2), ("camera", None, 1)):
        if env:
            monkeypatch.setenv("ESL_CF_SPARSE", env)
        else:
            monkeypatch.delenv("ESL_CF_SPARSE", raising=False)
        cx = pkg.Context(0)
        try:
            cx.upload_graph(g); cx.upload_states(c, o)
            cx.lm_begin(pkg.default_lm_params(jacobian_mode=1, linear_solver=solver))
            part = cx.lm_linearize()
            tr = cx.lm_try_step(1e-5 * part.max_diag)
            assert tr.solve_ok == 1 and cx.lm_solver_used() == solver
            xc, xo = cx.lm_download(5, 6 * nf), cx.lm_download(2, 9 * g.n_objs)
            res = cx.lm_reduced_residual()
            prof_classes = None
            cx.lm_commit(False)
            cx.profile_enable(2)
            cc, oo, rep = cx.optimize(g, c, o, pkg.default_lm_params(jacobian_mode=1, linear_solver=solver))
            prof_classes = set(cx.profile_get())
            out[tag] = (xc, xo, tr.chi2, res, cc, oo, rep, prof_classes)
        finally:
            cx.close()
    monkeypatch.delenv("ESL_CF_SPARSE", raising=False)
    assert "sparse_block_products" in out["sparse"][7] and "sparse_block_products" not in out["dense"][7]   # the form under test ran
    ref = out["dense"]
    assert "sparse_block_products" in out["sparse_direct"][7]
    for tag in ("sparse", "sparse_direct", "camera"):
        xc, xo, chi, res, cc, oo, rep, _ = out[tag]
        d_xc = float(np.abs(xc - ref[0]).max() / np.abs(ref[0]).max()); d_xo = float(np.abs(xo - ref[1]).max() / np.abs(ref[1]).max())
        print("n_cams %d, %s vs dense X: x_c %.2e x_o %.2e, |Sx-b|/|b| %.2e, run: chi2 rel %.2e cams %.2e" % (
            n_cams, tag, d_xc, d_xo, res, abs(rep["chi2_final"] / ref[6]["chi2_final"] - 1), cam_err(cc, ref[4])))
        assert res < 1e-13 and d_xc < 1e-10 and d_xo < 1e-10
        assert chi == pytest.approx(ref[2], rel=1e-12)
        assert rep["trace_trials"] == ref[6]["trace_trials"]
        np.testing.assert_allclose(rep["trace_chi2"], ref[6]["trace_chi2"], rtol=1e-11)
        assert cam_err(cc, ref[4]) < 1e-9 and obj_rel(oo, ref[5]) < 1e-9


def test_sparse_interior_rows_on_awkward_structures(pkg, monkeypatch):
    """The sparse form's tables on structures a synthetic trajectory does not produce: an ellipsoid seen only by SEPARATOR cameras (in
    no segment's list: its column of X lives in the separators' rows alone), one seen only by the fixed camera 0 (no column of X at
    all), one seen inside a single segment, a segment none of whose interior cameras sees anything (an empty slab), and the last
    segment short.  Stored products (=1) and blocks straight from the slabs (=2) against dense X and the reduced camera system."""
    g0, c, o, _ = pkg.synth.make_graph(330, 14, 3300, seed=43, slam=True)

    def keep(cam, obj):
        k = np.ones(len(cam), bool)
        k &= ~((obj == 0) & ~((cam % 16 == 0) & (cam > 0)))          # ellipsoid 0: separator cameras only (slot = cam - 1, separators at slot % 16 == 15)
        k &= ~((obj == 1) & (cam != 0))                              # ellipsoid 1: the fixed camera only
        k &= ~((obj == 2) & ~((cam >= 17) & (cam <= 30)))            # ellipsoid 2: inside segment 1
        k &= ~((cam >= 81) & (cam <= 95))                            # segment 5: no interior observation
        return k
    kb, k3 = keep(g0.bbox_cam, g0.bbox_obj), keep(g0.e3d_cam, g0.e3d_obj)
    # make sure the special ellipsoids still have what they need
    extra_cam = np.array([16, 32, 48, 0, 0, 20, 25], np.int32); extra_obj = np.array([0, 0, 0, 1, 1, 2, 2], np.int32)
    bb, _, _ = pkg.synth.project_bboxes(c, o, g0.K, extra_cam, extra_obj)
    okb = np.isfinite(bb).all(1)
    g = pkg.Graph(g0.K, g0.n_cams, g0.n_objs, g0.cam_fixed,
                  np.concatenate([g0.bbox_cam[kb], extra_cam[okb]]), np.concatenate([g0.bbox_obj[kb], extra_obj[okb]]),
                  np.concatenate([g0.bbox_meas.reshape(-1, 4)[kb], bb[okb]]), np.concatenate([g0.bbox_weight[kb], np.full(okb.sum(), 0.7)]),
                  g0.e3d_cam[k3], g0.e3d_obj[k3], g0.e3d_meas.reshape(-1, 10)[k3], g0.e3d_weight[k3],
                  g0.grav_obj, g0.grav_normal, g0.grav_weight, g0.odom_i, g0.odom_j, g0.odom_meas, g0.odom_info)
    cams_of = lambda ob: set(np.concatenate([g.bbox_cam[g.bbox_obj == ob], g.e3d_cam[g.e3d_obj == ob]]).tolist())
    assert cams_of(0) and all(cm % 16 == 0 and cm > 0 for cm in cams_of(0)) and cams_of(1) == {0} and cams_of(2) <= set(range(17, 31))
    assert not (set(range(81, 96)) & set(np.concatenate([g.bbox_cam, g.e3d_cam]).tolist()))
    nf = g.n_cams - 1
    out = {}
    for tag, env, solver in (("stored", "1", 2), ("direct", "2", 2), ("dense", "0", 2), ("camera", None, 1)):
        if env:
            monkeypatch.setenv("ESL_CF_SPARSE", env)
        else:
            monkeypatch.delenv("ESL_CF_SPARSE", raising=False)
        cx = pkg.Context(0)
        try:
            cx.upload_graph(g); cx.upload_states(c, o)
            cx.lm_begin(pkg.default_lm_params(jacobian_mode=1, linear_solver=solver))
            part = cx.lm_linearize()
            tr = cx.lm_try_step(1e-5 * part.max_diag)
            assert cx.lm_solver_used() == solver
            st = cx.lm_solver_stats()
            xc, xo = cx.lm_download(5, 6 * nf), cx.lm_download(2, 9 * g.n_objs)
            res = cx.lm_reduced_residual() if tr.solve_ok == 1 else float("nan")
            cx.lm_commit(False)
            cc, oo, rep = cx.optimize(g, c, o, pkg.default_lm_params(jacobian_mode=1, linear_solver=solver))
            out[tag] = (xc, xo, tr.chi2, tr.solve_ok, res, cc, oo, rep, st)
        finally:
            cx.close()
    monkeypatch.delenv("ESL_CF_SPARSE", raising=False)
    assert out["stored"][8]["x_form"] == 1 and out["direct"][8]["x_form"] == 2 and out["dense"][8]["x_form"] == 0
    assert out["stored"][8]["separators"] == nf // 16 and out["stored"][8]["segments"] == (nf + 15) // 16
    ref = out["dense"]
    for tag in ("stored", "direct", "camera"):
        xc, xo, chi, ok, res, cc, oo, rep, _ = out[tag]
        assert ok == ref[3]
        d_xc = float(np.abs(xc - ref[0]).max() / np.abs(ref[0]).max()); d_xo = float(np.abs(xo - ref[1]).max() / np.abs(ref[1]).max())
        print("awkward structures, %s vs dense X: x_c %.2e x_o %.2e, |Sx-b|/|b| %.2e, run: chi2 rel %.2e cams %.2e" % (
            tag, d_xc, d_xo, res, abs(rep["chi2_final"] / ref[7]["chi2_final"] - 1), cam_err(cc, ref[5])))
        assert d_xc < 1e-9 and d_xo < 1e-9 and chi == pytest.approx(ref[2], rel=1e-11)
        assert rep["trace_trials"] == ref[7]["trace_trials"]
        np.testing.assert_allclose(rep["trace_chi2"], ref[7]["trace_chi2"], rtol=1e-10)
        assert cam_err(cc, ref[5]) < 1e-8 and obj_rel(oo, ref[6]) < 1e-8


def test_sparse_camera_first_trial_matches_cpu_camera_first_checker(pkg, po, monkeypatch):
    """The sparse camera-first path (nested dissection of a 2,047-camera chain into 128 segments of 16 slots, X kept sparse,
    per-segment products, separators' rows on the MFMA update: every piece of esl_cf.hpp live, as at BASELINE configs[3]) against the
    CPU checker at a size the checker reaches: ONE LM iteration from the same start, numeric Jacobians at delta = 1e-4 on both
    sides (round 6: at 1e-6 the checker's central differences carry ~1e-7 of roundoff noise, which was the whole 7.9e-6 this test
    used to measure on the cameras -- see test_mid_size_slam_full_run_matches_cpu_camera_first_checker).  The checker eliminates the cameras first too, but as the PLAIN chain with a dense Y and a pivoted LDLT
    (oracle/esl_oracle.c solve_camfirst; tests/test_oracle_cross.py holds it to the faithful dense LDLT of the whole system) --
    none of the dissection / sparsity machinery."""
    g, c, o, _ = pkg.synth.make_graph(2048, 150, 8 * 2048, seed=43, slam=True)
    p1 = pkg.default_lm_params(numeric_delta=1e-4, max_iters=1)
    co, oo, ro = po.optimize(g, c, o, p1, solver=po.ORACLE_CAMFIRST)
    monkeypatch.setenv("ESL_CF_SPARSE", "1")
    cx = pkg.Context(0)
    try:
        for jac in (0, 1):
            cg, og, rg = cx.optimize(g, c, o, pkg.default_lm_params(jacobian_mode=jac, numeric_delta=1e-4, max_iters=1, linear_solver=2))
            st = cx.lm_solver_stats()
            assert cx.lm_solver_used() == 2 and st["x_form"] == 1 and st["stride"] == 16 and st["separators"] == 127
            print("2,048-camera chain, sparse camera-first trial (jac %d) vs CPU camera-first checker: chi2 rel %.2e, cameras %.2e, ellipsoids %.2e, trials %s / %s"
                  % (jac, abs(rg["chi2_final"] / ro["chi2_final"] - 1), cam_err(cg, co), obj_rel(og, oo), rg["trace_trials"], ro["trace_trials"]))
            assert rg["trace_trials"] == ro["trace_trials"]
            assert rg["chi2_final"] == pytest.approx(ro["chi2_final"], rel=1e-8)
            assert cam_err(cg, co) < 1e-6 and obj_rel(og, oo) < 5e-7   # measured (round 6): 1.7e-7 / 7.8e-8
    finally:
        cx.close()
        monkeypatch.delenv("ESL_CF_SPARSE", raising=False)


def test_mid_size_slam_full_run_matches_cpu_camera_first_checker(pkg, monkeypatch):
    """A WHOLE optimize(10) in SLAM mode on the form BASELINE configs[3] runs (chain dissected into 16-slot segments, X kept sparse,
    stored per-segment products, separators' rows on the MFMA update, dense factorisation of the 2,700 ellipsoid unknowns) against the
    CPU checker's camera-first restatement (plain chain, dense Y, pivoted LDLT: none of that machinery) -- 11,994 camera + 2,700
    ellipsoid unknowns, the graph bench.py times as `mid_2k_cams_300_ellipsoids`.  The checker's runs are a committed fixture (2.5
    minutes each on one core: tests/golden/gen_golden_mid_slam.py; tests/test_golden_vectors.py re-runs a first iteration on the CPU).

    Round 6 ("resolve the 1.08e-4 camera discrepancy instead of widening the assert").  It was the CHECKER's
    numeric-Jacobian roundoff, not the elimination: (a) the checker's camera-first solve agrees with a dense Cholesky solve of its own
    (H + lambda I) x = b refined in long double to 7e-14 (scripts/debug/camfirst_checker_precision.py), so its linear algebra is
    exact; (b) its central differences carry a noise ~ eps_f / delta (error functions good to ~1e-13): re-run at delta = 1e-5 and
    1e-4 the checker moves 1.1e-4 away from its own delta = 1e-6 run on the cameras -- and TOWARD the GPU: the GPU (analytic or
    numeric at any delta, either elimination: 2e-7 among themselves) is 1.08e-4 / 4.2e-6 / 8e-7 from the checker at 1e-6 / 1e-5 / 1e-4.
    The like-for-like step is therefore delta = 1e-4 on both sides, held to 5e-6 on the cameras and 2e-7 on the ellipsoids
    (north_star: 1e-4); the older steps stay in the fixture and the sequence above is asserted."""
    import os
    sys_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_golden_mid_slam", os.path.join(sys_path, "gen_golden_mid_slam.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    G = np.load(os.path.join(sys_path, "mid_slam_run.npz"))
    delta = float(G["numeric_delta"])
    assert delta == gen.DELTA == 1e-4
    g, c, o, _ = pkg.synth.make_graph(**gen.ARGS)
    assert list(G["n_edges"]) == [len(g.bbox_cam), len(g.e3d_cam), len(g.odom_i)]
    co, oo = G["cams"], G["objs"]
    ro = dict(trace_chi2=list(G["trace_chi2"]), trace_trials=list(G["trace_trials"]), iterations=int(G["iterations"]), stop_reason=int(G["stop_reason"]),
              chi2_final=float(G["chi2_final"]), chi2_initial=float(G["chi2_initial"]))
    assert ro["iterations"] >= 3 and ro["chi2_final"] < ro["chi2_initial"]
    monkeypatch.setenv("ESL_CF_SPARSE", "1")
    cx = pkg.Context(0)
    try:
        cg, og, rg = cx.optimize(g, c, o, pkg.default_lm_params(jacobian_mode=0, numeric_delta=delta, linear_solver=2))
        st = cx.lm_solver_stats()
        assert cx.lm_solver_used() == 2 and st["x_form"] == 1 and st["stride"] == 16 and st["separators"] == 1999 // 16
        tr = float(np.abs(np.array(rg["trace_chi2"]) / np.array(ro["trace_chi2"]) - 1).max()) if rg["trace_trials"] == ro["trace_trials"] else float("nan")
        print("2,000 cams / 300 ellipsoids, full SLAM run, sparse camera-first (numeric %g) vs CPU camera-first checker at the same step: %d iterations, "
              "trials %s / %s, chi2 trace rel %.2e, final chi2 rel %.2e, cameras %.2e, ellipsoids %.2e" % (
                  delta, rg["iterations"], rg["trace_trials"], ro["trace_trials"], tr, abs(rg["chi2_final"] / ro["chi2_final"] - 1), cam_err(cg, co), obj_rel(og, oo)))
        assert [int(t) for t in rg["trace_trials"]] == [int(t) for t in ro["trace_trials"]] and rg["stop_reason"] == ro["stop_reason"]
        assert rg["chi2_initial"] == pytest.approx(ro["chi2_initial"], rel=1e-9)
        # measured on MI355X (round 6): chi2 final 8.7e-13, cameras 9.4e-7, ellipsoids 1e-8
        np.testing.assert_allclose(rg["trace_chi2"], ro["trace_chi2"], rtol=1e-9)
        assert obj_rel(og, oo) < 2e-7 and cam_err(cg, co) < 5e-6
        c1, o1, r1 = cx.optimize(g, c, o, pkg.default_lm_params(jacobian_mode=0, numeric_delta=delta, linear_solver=1))
        print("   the GPU's reduced-camera elimination on the same graph: vs its camera-first run chi2 rel %.2e cameras %.2e ellipsoids %.2e | vs the checker "
              "cameras %.2e ellipsoids %.2e" % (abs(r1["chi2_final"] / rg["chi2_final"] - 1), cam_err(c1, cg), obj_rel(o1, og), cam_err(c1, co), obj_rel(o1, oo)))
        assert r1["trace_trials"] == rg["trace_trials"] and r1["chi2_final"] == pytest.approx(rg["chi2_final"], rel=1e-9)
        assert cam_err(c1, cg) < 1e-6 and obj_rel(o1, oo) < 2e-7 and cam_err(c1, co) < 5e-6
        # the product default (analytic Jacobians) against the checker at its three finite steps: the checker converges to it as its
        # own roundoff noise (~ 1 / delta) goes down; and the checker against itself, which is that noise
        ca, oa, ra = cx.optimize(g, c, o, pkg.default_lm_params(jacobian_mode=1, linear_solver=2))
        seq = [(sfx, cam_err(ca, G["cams" + sfx]), obj_rel(oa, G["objs" + sfx]), abs(ra["chi2_final"] / float(G["chi2_final" + sfx]) - 1)) for sfx in ("_d6", "_d5", "")]
        print("   analytic Jacobians (%d iterations) vs the checker at delta 1e-6 / 1e-5 / 1e-4: cameras %s, ellipsoids %s, final chi2 %s" % (
            ra["iterations"], " / ".join("%.2e" % s[1] for s in seq), " / ".join("%.2e" % s[2] for s in seq), " / ".join("%.1e" % s[3] for s in seq)))
        print("   checker against itself: delta 1e-6 vs 1e-4 cameras %.2e, 1e-5 vs 1e-4 %.2e, 1e-9 (g2o's) vs 1e-4 %.2e (ellipsoids %.2e)" % (
            cam_err(G["cams_d6"], co), cam_err(G["cams_d5"], co), cam_err(G["cams_d9"], co), obj_rel(G["objs_d9"], oo)))
        assert seq[0][1] > seq[1][1] > seq[2][1]                       # monotone: the distance IS the checker's noise
        assert seq[0][1] < 2e-4 and seq[1][1] < 2e-5 and seq[2][1] < 5e-6 and seq[2][2] < 2e-7
        assert cam_err(G["cams_d6"], co) > 10 * seq[2][1]               # the checker is further from itself (1e-6 vs 1e-4) than the GPU is from its 1e-4 run
        assert ra["chi2_final"] == pytest.approx(ro["chi2_final"], rel=1e-8)
        # the reference's own step (delta = 1e-9) on the GPU: its distance from everything else is the GPU's own Jacobian noise at that step,
        # two orders below the checker's (1.7e-2 on the cameras) -- inside north_star's 1e-4 on the objective and the ellipsoids
        c9, o9, r9 = cx.optimize(g, c, o, pkg.default_lm_params(jacobian_mode=0, numeric_delta=1e-9, linear_solver=2))
        print("   numeric 1e-9 (g2o's step) on the GPU vs the checker at 1e-4: %d iterations, final chi2 rel %.2e, cameras %.2e, ellipsoids %.2e" % (
            r9["iterations"], abs(r9["chi2_final"] / ro["chi2_final"] - 1), cam_err(c9, co), obj_rel(o9, oo)))
        assert r9["chi2_final"] == pytest.approx(ro["chi2_final"], rel=1e-6)
        assert obj_rel(o9, oo) < 1e-4 and cam_err(c9, co) < 1e-3 and cam_err(c9, co) < 0.1 * cam_err(G["cams_d9"], co)
    finally:
        cx.close()
        monkeypatch.delenv("ESL_CF_SPARSE", raising=False)


def test_ctx_trim_releases_the_solver_blobs_and_they_come_back(pkg):
    """esl_ctx_trim (ABI 4): the grow-only solver blobs of SLAM mode -- the camera-first set, the reduced camera system --
    go back to the device; the next trial step builds what it needs again (index tables included) and lands on the same bits.  A
    mapping-mode graph uploaded over a SLAM-mode one trims by itself.  S lives in a blob of its own: esl_lm_reduced_system beside a
    camera-first run no longer re-lays that run's tables out (the pointer it returns stays valid across the next trial step)."""
    import ctypes

    def free_bytes():   # hipMemGetInfo of the HIP runtime libesl_hip.so itself runs on (no second runtime in the process)
        pkg.lib.load()
        # the runtime libesl_hip.so is linked against, by the path it is mapped from (no soname of one ROCm major version hardcoded)
        paths = [ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64.so" in ln]
        if not paths:
            pytest.skip("libamdhip64 is not mapped into this process")
        hip = ctypes.CDLL(paths[0])
        fr, tot = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipMemGetInfo(ctypes.byref(fr), ctypes.byref(tot)) == 0
        return fr.value
    g, c, o, _ = pkg.synth.make_graph(600, 120, 6000, seed=12, slam=True)
    p2 = pkg.default_lm_params(jacobian_mode=1, linear_solver=2, max_iters=3)
    cx = pkg.Context(0)
    try:
        c2, o2, r2 = cx.optimize(g, c, o, p2)
        cx.upload_graph(g); cx.upload_states(c, o)
        c1, o1, r1 = None, None, cx.optimize_resident(pkg.default_lm_params(jacobian_mode=1, linear_solver=1, max_iters=3))   # S: 3,594^2 doubles = 103 MB beside the camera-first set
        free_before = free_bytes()
        cx.trim()
        free_after = free_bytes()
        assert free_after - free_before > 100e6, (free_before, free_after)
        cx.upload_states(c, o)
        ca, oa = cx.download_states()
        rb = cx.optimize_resident(p2)
        cb, ob = cx.download_states()
        assert rb["trace_chi2"] == r2["trace_chi2"] and np.array_equal(cb, c2) and np.array_equal(ob, o2)
        # the reduced camera system beside a camera-first run: the pointer survives the next camera-first trial
        cx.upload_states(c, o)
        cx.lm_begin(p2)
        part = cx.lm_linearize()
        ptr, n, lda = cx.lm_reduced_system(1e-4 * part.max_diag)
        S0 = cx.lm_download(6, lda * n)
        tr = cx.lm_try_step(1e-4 * part.max_diag)
        assert cx.lm_solver_used() == 2 and tr.solve_ok == 1
        ptr2, n2, lda2 = cx.lm_reduced_system(1e-4 * part.max_diag)
        assert (ptr2, n2, lda2) == (ptr, n, lda) and np.array_equal(cx.lm_download(6, lda * n), S0)
        cx.lm_commit(False)
        # a mapping-mode graph over a SLAM-mode one: the SLAM blobs go by themselves
        gm, cm, om, _ = pkg.synth.make_graph(600, 120, 6000, seed=12, slam=False)
        free_slam = free_bytes()
        cx.upload_graph(gm); cx.upload_states(cm, om)
        assert free_bytes() - free_slam > 100e6
        assert np.isfinite(cx.optimize_resident(pkg.default_lm_params(jacobian_mode=1))["chi2_final"])
    finally:
        cx.close()
