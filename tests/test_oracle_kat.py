"""Analytic known-answer tests that pin the CPU restatement.

The reference ships no tests; these KATs + the independent numpy restatement (test_oracle_cross.py)
are what anchors oracle/esl_oracle.c.
"""
import numpy as np
import pytest
from scipy.linalg import expm, logm

K = [535.4, 539.2, 320.1, 247.6]  # Example/param/TUM3.yaml:62-65
I7 = [0, 0, 0, 0, 0, 0, 1.0]


def hat(u):
    w, v = u[:3], u[3:]
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    M[:3, 3] = v
    return M


def T_of(v7):
    from oracle import np_oracle as npo
    return npo.T_from7(np.asarray(v7, float))


def test_sphere_bbox(po):
    # sphere r=0.5 at (0,0,3): half widths f*r/sqrt(z^2-r^2)
    bb = po.project_bbox(I7, [0, 0, 3, 0, 0, 0, 1, .5, .5, .5], K)
    hw = K[0] * 0.5 / np.sqrt(9 - 0.25)
    hh = K[1] * 0.5 / np.sqrt(9 - 0.25)
    np.testing.assert_allclose(bb, [K[2] - hw, K[3] - hh, K[2] + hw, K[3] + hh], rtol=0, atol=1e-9)
    np.testing.assert_allclose(bb, [229.6008824, 156.45856517, 410.5991176, 338.74143483], atol=1e-7)


def test_general_ellipsoid_bbox(po):
    obj = po.obj_from_minimal([0.3, -0.2, 2.5, 0.1, -0.2, 0.7, 0.4, 0.25, 0.6])
    bb = po.project_bbox(I7, obj, K)
    np.testing.assert_allclose(bb, [310.39067695, 133.0746008, 467.84972116, 279.31235526], atol=1e-7)
    # brute force: project surface points, the bbox must enclose them tightly
    from oracle import np_oracle as npo
    T, s = npo.obj_from10(obj)
    th, ph = np.meshgrid(np.linspace(0, np.pi, 400), np.linspace(0, 2 * np.pi, 800))
    P = np.stack([s[0] * np.sin(th) * np.cos(ph), s[1] * np.sin(th) * np.sin(ph), s[2] * np.cos(th)], -1).reshape(-1, 3)
    Pw = P @ T[:3, :3].T + T[:3, 3]
    u = K[0] * Pw[:, 0] / Pw[:, 2] + K[2]
    v = K[1] * Pw[:, 1] / Pw[:, 2] + K[3]
    assert abs(u.min() - bb[0]) < 0.05 and abs(u.max() - bb[2]) < 0.05
    assert abs(v.min() - bb[1]) < 0.05 and abs(v.max() - bb[3]) < 0.05


def test_minimal_roundtrip(po):
    v = np.array([0.3, -0.2, 2.5, 0.1, -0.2, 0.7, 0.4, 0.25, 0.6])
    np.testing.assert_allclose(po.obj_to_minimal(po.obj_from_minimal(v)), v, atol=1e-14)


@pytest.mark.parametrize("theta", [1e-3, 1.0, 3.0])
def test_exp_matches_matrix_exponential(po, theta):
    rng = np.random.default_rng(3)
    w = rng.standard_normal(3); w *= theta / np.linalg.norm(w)
    u = np.concatenate([w, rng.standard_normal(3)])
    T = T_of(po.se3_exp(u))
    np.testing.assert_allclose(T, expm(hat(u)), atol=1e-12)
    np.testing.assert_allclose(po.se3_log(po.se3_exp(u)), u, atol=1e-9 if theta < 3 else 1e-7)


def test_log_matches_matrix_log(po):
    rng = np.random.default_rng(4)
    for _ in range(10):
        u = np.concatenate([0.8 * rng.standard_normal(3), rng.standard_normal(3)])
        T7 = po.se3_exp(u)
        L = np.real(logm(T_of(T7)))
        ref = np.array([L[2, 1], L[0, 2], L[1, 0], L[0, 3], L[1, 3], L[2, 3]])
        np.testing.assert_allclose(po.se3_log(T7), ref, atol=1e-10)


def test_small_angle_branch_quirk(po):
    # theta < 1e-5: R = I + W + W^2 and V = R (se3quat.h:290-301), NOT the Rodrigues series
    u = np.array([3e-6, -2e-6, 1e-6, 0.5, -0.25, 0.125])
    T = T_of(po.se3_exp(u))
    W = hat(u)[:3, :3]
    V = np.eye(3) + W + W @ W
    np.testing.assert_allclose(T[:3, 3], V @ u[3:], atol=1e-16)
    true_t = expm(hat(u))[:3, 3]  # true series has V = I + W/2 + ...
    assert np.abs(T[:3, 3] - true_t).max() > 1e-7
    # log(exp(delta)) = delta to first order for the numeric-differentiation step
    d = np.zeros(6); d[1] = 1e-9; d[4] = 1e-9
    np.testing.assert_allclose(po.se3_log(po.se3_exp(d)), d, atol=1e-17)


def test_mul_inv_and_sign(po):
    rng = np.random.default_rng(5)
    A = po.se3_exp(rng.standard_normal(6)); B = po.se3_exp(rng.standard_normal(6))
    AB = po.se3_mul(A, B)
    assert AB[6] >= 0  # normalizeRotation keeps w >= 0
    np.testing.assert_allclose(T_of(AB), T_of(A) @ T_of(B), atol=1e-13)
    np.testing.assert_allclose(T_of(po.se3_mul(A, po.se3_inv(A))), np.eye(4), atol=1e-13)


def test_e3d_yaw_hypothesis_zero_residual(po):
    # est = meas_w rotated by +90 deg yaw with a/b swapped -> residual through hypothesis k=1 is just
    # the small extra twist we add (an exact 180-degree hypothesis would hit log()'s acos(-1)
    # singularity — in the reference too — so keep a 1e-3 offset)
    rng = np.random.default_rng(6)
    cam = po.se3_exp(0.3 * rng.standard_normal(6))
    meas = np.concatenate([po.se3_exp(np.array([0.1, -0.2, 0.3, 0.5, 0.2, 2.0])), [0.4, 0.25, 0.6]])
    Twc = po.se3_inv(cam)
    mw = po.se3_mul(Twc, meas[:7])
    rot = [0, 0, 0, 0, 0, np.sin(np.pi / 4), np.cos(np.pi / 4)]
    twist = np.array([1e-3, -2e-3, 3e-3, 0.01, 0.02, -0.01])
    est = np.concatenate([po.se3_mul(po.se3_mul(mw, rot), po.se3_exp(twist)), [0.25 + 0.01, 0.4, 0.6 - 0.02]])
    r = po.res_e3d(cam, est, meas)
    np.testing.assert_allclose(r[:6], twist, atol=2e-8)  # log() small-angle branch: omega = dR/2, O(theta^3)
    np.testing.assert_allclose(r[6:], [0.01, 0, -0.02], atol=1e-12)
    # hypothesis k=0 (no yaw, no swap)
    est2 = np.concatenate([po.se3_mul(mw, po.se3_exp(twist)), [0.4, 0.25 + 0.03, 0.6]])
    r2 = po.res_e3d(cam, est2, meas)
    np.testing.assert_allclose(r2[:6], twist, atol=2e-8)
    np.testing.assert_allclose(r2[6:], [0, 0.03, 0], atol=1e-12)


def test_gravity_residual(po):
    # object tilted by 0.3 rad about x: angle between its z axis and (0,0,1) is 0.3
    obj = np.concatenate([[1, 2, 3], [np.sin(0.15), 0, 0, np.cos(0.15)], [1, 1, 1]])
    np.testing.assert_allclose(po.res_grav(obj, [0, 0, 1, 0]), [0.3], atol=1e-12)
    np.testing.assert_allclose(po.res_grav(obj, [0, 0, 2, 5]), [0.3], atol=1e-12)


def test_bbox_measurement_mask(po):
    obj = [0, 0, 3, 0, 0, 0, 1, .5, .5, .5]
    bb = po.project_bbox(I7, obj, K)
    r = po.res_bbox(I7, obj, K, [-1, 4.99, 400, 5])
    np.testing.assert_allclose(r, [0, 0, bb[2] - 400, bb[3] - 5], atol=1e-12)


def test_bbox_nan_when_camera_inside(po):
    bb = po.project_bbox(I7, [0, 0, 0.1, 0, 0, 0, 1, 1, 1, 1], K)
    assert np.isnan(bb).any()


def test_odometry_residual_zero(po):
    rng = np.random.default_rng(7)
    Ti = po.se3_exp(rng.standard_normal(6)); Tj = po.se3_exp(rng.standard_normal(6))
    Z = po.se3_mul(Tj, po.se3_inv(Ti))  # Optimizer.cpp:143-146
    np.testing.assert_allclose(po.res_odom(Ti, Tj, Z), 0, atol=1e-12)


def test_ldlt_matches_numpy_and_detects_indefinite(po):
    rng = np.random.default_rng(8)
    A = rng.standard_normal((12, 12)); A = A @ A.T + 1e-3 * np.eye(12)
    b = rng.standard_normal(12)
    ok, x = po.ldlt_solve(A, b)
    assert ok
    np.testing.assert_allclose(x, np.linalg.solve(A, b), rtol=1e-9)
    A[3, 3] = -5.0
    ok, _ = po.ldlt_solve(A, b)
    assert not ok


def test_numeric_jacobian_is_central_difference(po):
    rng = np.random.default_rng(9)
    cam = po.se3_exp(0.1 * rng.standard_normal(6))
    obj = po.obj_from_minimal([0.3, -0.2, 2.5, 0.1, -0.2, 0.7, 0.4, 0.25, 0.6])
    meas = po.project_bbox(cam, obj, K) + 1.0
    Jc, Jo = po.jac_bbox(cam, obj, K, meas, 1e-6)
    for d in range(9):
        u = np.zeros(9); u[d] = 1e-6
        col = (po.res_bbox(cam, po.obj_oplus(obj, u), K, meas) - po.res_bbox(cam, po.obj_oplus(obj, -u), K, meas)) / 2e-6
        np.testing.assert_allclose(Jo[:, d], col, atol=1e-12, rtol=0)
    for d in range(6):
        u = np.zeros(6); u[d] = 1e-6
        col = (po.res_bbox(po.cam_oplus(cam, u), obj, K, meas) - po.res_bbox(po.cam_oplus(cam, -u), obj, K, meas)) / 2e-6
        np.testing.assert_allclose(Jc[:, d], col, atol=1e-12, rtol=0)
