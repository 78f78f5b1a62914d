"""C restatement vs the independent numpy restatement."""
import numpy as np
import pytest

from oracle import np_oracle as npo

K = (535.4, 539.2, 320.1, 247.6)


def rand_scene(rng):
    from oracle import pyoracle as po
    cam = po.se3_exp(np.concatenate([0.3 * rng.standard_normal(3), rng.standard_normal(3) * 0.3]))
    obj = po.obj_from_minimal(np.concatenate([rng.uniform(-0.5, 0.5, 2), [rng.uniform(2.5, 4)],
                                              rng.uniform(-1, 1, 3) * [0.3, 0.3, 3],
                                              rng.uniform(0.15, 0.6, 3)]))
    return cam, obj


def test_residuals_agree(po):
    rng = np.random.default_rng(0)
    for _ in range(64):
        cam, obj = rand_scene(rng)
        Tc = npo.T_from7(cam); To, s = npo.obj_from10(obj)
        meas = rng.uniform(50, 400, 4)
        r1 = po.res_bbox(cam, obj, K, meas); r2 = npo.res_bbox(Tc, To, s, K, meas)
        if np.isnan(r1).any():
            assert np.isnan(r2).any()
        else:
            np.testing.assert_allclose(r1, r2, atol=5e-8)  # the reference's inverse/atan2 chain loses ~1e-9 px
        m = np.concatenate([po.se3_exp(np.concatenate([0.5 * rng.standard_normal(3), [0.1, 0.1, 3.0]])), rng.uniform(0.15, 0.6, 3)])
        np.testing.assert_allclose(po.res_e3d(cam, obj, m), npo.res_e3d(Tc, To, s, m), atol=1e-11)
        np.testing.assert_allclose(po.res_grav(obj, [0.02, -0.01, 1, 0]), npo.res_grav(To, [0.02, -0.01, 1, 0]), atol=1e-9)
        cam2, _ = rand_scene(rng)
        Z = po.se3_mul(po.se3_mul(cam2, po.se3_inv(cam)), po.se3_exp(0.01 * rng.standard_normal(6)))
        np.testing.assert_allclose(po.res_odom(cam, cam2, Z), npo.res_odom(Tc, npo.T_from7(cam2), npo.T_from7(Z)), atol=1e-11)


def test_bbox_nan_semantics_agree(po):
    """camera inside / ellipsoid through the principal plane: both restatements give NaN."""
    rng = np.random.default_rng(1)
    n_nan = 0
    for _ in range(200):
        cam, obj = rand_scene(rng)
        obj = obj.copy(); obj[2] = rng.uniform(-0.6, 0.9)  # near / across the camera plane
        r1 = po.project_bbox(cam, obj, K)
        r2 = npo.project_bbox(npo.T_from7(cam), *npo.obj_from10(obj), K)
        assert np.isnan(r1).any() == np.isnan(r2).any()
        n_nan += int(np.isnan(r1).any())
    assert n_nan > 20


@pytest.mark.parametrize("slam", [False, True])
def test_lm_trajectories_agree(po, pkg, slam):
    """delta = 1e-6 keeps the numeric-Jacobian noise at 1e-6 so the two LM runs stay on one trajectory."""
    g, c, o, _ = pkg.synth.make_config("tiny", seed=1, slam=slam)
    p = pkg.default_lm_params(numeric_delta=1e-6)
    co, oo, rep = po.optimize(g, c, o, p, solver=0)
    cn, on, rn = npo.optimize(g, c, o, delta=1e-6)
    assert rep["trace_trials"] == [t[2] for t in rn["trace"]]
    np.testing.assert_allclose(rep["trace_chi2"], [t[0] for t in rn["trace"]], rtol=1e-5)
    np.testing.assert_allclose(oo, on, atol=2e-5)
    np.testing.assert_allclose(co, cn, atol=2e-5)


def test_block_solver_is_bit_identical_in_mapping_mode(po, pkg):
    """per-ellipsoid pivoted LDLT == dense pivoted LDLT of the block-diagonal system"""
    g, c, o, _ = pkg.synth.make_config("tiny", seed=2)
    _, o1, r1 = po.optimize(g, c, o, solver=0)
    _, o2, r2 = po.optimize(g, c, o, solver=1)
    assert np.array_equal(o1, o2) and r1["trace_chi2"] == r2["trace_chi2"]


def test_schur_matches_dense_in_slam_mode(po, pkg):
    g, c, o, _ = pkg.synth.make_graph(30, 6, 200, seed=3, slam=True)
    p = pkg.default_lm_params(numeric_delta=1e-6)
    c1, o1, r1 = po.optimize(g, c, o, p, solver=0)
    c2, o2, r2 = po.optimize(g, c, o, p, solver=1)
    assert r1["trace_trials"] == r2["trace_trials"]
    np.testing.assert_allclose(o1, o2, atol=1e-5)
    np.testing.assert_allclose(c1, c2, atol=1e-5)


@pytest.mark.parametrize("n_cams,n_objs,seed", [(30, 6, 3), (120, 12, 9)])
def test_camera_first_checker_matches_dense_in_slam_mode(po, pkg, n_cams, n_objs, seed):
    """solver 2 of the checker (cameras eliminated first along the odometry chain, oracle/esl_oracle.c solve_camfirst: block-
    bidiagonal factor, dense Y, pivoted LDLT of the reduced ELLIPSOID system) solves the same damped system as the faithful dense
    pivoted LDLT (linear_solver_dense.h:65-113) and as the Schur complement onto the cameras: the same LM run.  It is what
    tests/test_gpu_slam.py holds the GPU's sparse camera-first form to at 2,048 cameras."""
    g, c, o, _ = pkg.synth.make_graph(n_cams, n_objs, 8 * n_cams, seed=seed, slam=True)
    p = pkg.default_lm_params(numeric_delta=1e-6, max_iters=3)
    c0, o0, r0 = po.optimize(g, c, o, p, solver=po.ORACLE_DENSE)
    for solver in (po.ORACLE_BLOCK, po.ORACLE_CAMFIRST):
        c2, o2, r2 = po.optimize(g, c, o, p, solver=solver)
        assert r2["trace_trials"] == r0["trace_trials"]
        # (three exact solvers of one ill-conditioned system: measured 1e-7 on the objective, the figure the existing Schur test has)
        np.testing.assert_allclose(r2["trace_chi2"], r0["trace_chi2"], rtol=1e-6)
        np.testing.assert_allclose(o2, o0, atol=1e-5)
        np.testing.assert_allclose(c2, c0, atol=1e-5)


def test_camera_first_checker_falls_back_when_the_cameras_are_no_chain(po, pkg):
    """a loop-closure odometry edge (first to last camera) breaks the block-tridiagonal structure: solver 2 must take the other
    elimination and still return the dense checker's run"""
    g, c, o, _ = pkg.synth.make_graph(24, 5, 160, seed=6, slam=True)
    oi, oj = np.append(g.odom_i, 1), np.append(g.odom_j, 23)
    om = np.vstack([g.odom_meas.reshape(-1, 7), g.odom_meas.reshape(-1, 7)[:1]])
    gl = pkg.Graph(g.K, g.n_cams, g.n_objs, g.cam_fixed, g.bbox_cam, g.bbox_obj, g.bbox_meas, g.bbox_weight, g.e3d_cam, g.e3d_obj, g.e3d_meas,
                   g.e3d_weight, g.grav_obj, g.grav_normal, g.grav_weight, odom_i=oi, odom_j=oj, odom_meas=om)
    p = pkg.default_lm_params(numeric_delta=1e-6, max_iters=2)
    c1, o1, r1 = po.optimize(gl, c, o, p, solver=po.ORACLE_BLOCK)
    c2, o2, r2 = po.optimize(gl, c, o, p, solver=po.ORACLE_CAMFIRST)
    assert r2["trace_chi2"] == r1["trace_chi2"] and np.array_equal(o2, o1) and np.array_equal(c2, c1)   # the very same code path
    c0, o0, r0 = po.optimize(gl, c, o, pkg.default_lm_params(numeric_delta=1e-6, max_iters=1), solver=po.ORACLE_DENSE)
    c3, o3, r3 = po.optimize(gl, c, o, pkg.default_lm_params(numeric_delta=1e-6, max_iters=1), solver=po.ORACLE_CAMFIRST)
    assert r3["trace_trials"] == r0["trace_trials"]
    np.testing.assert_allclose(o3, o0, atol=1e-6)
    np.testing.assert_allclose(c3, c0, atol=1e-6)
