"""BASELINE.json configs[4] — streaming: per frame the single-frame fit of its boxes and the re-optimisation of the graph that
has accumulated so far, with the graph EXTENDED on the device (esl_graph_append) instead of rebuilt (the reference rebuilds every
g2o object per frame, Optimizer.cpp:127, 166, 250).  The appended graph must be indistinguishable from an upload of the
concatenated one — same chunks, same sums: bit-identical runs — and every frame's result must match the CPU checker."""
import numpy as np
import pytest

from test_gpu_optimizer import group_rel_err

pytestmark = pytest.mark.gpu

# lock-step tolerances (lockstep_frame): both sides difference numerically at delta = 1e-6, so H and b carry ~1e-7 of differencing
# noise each (per-block tolerance inside lockstep_frame); the other pieces are plain fp64 arithmetic
LOCK_TOL = dict(lin_chi2=1e-9, H=1.0, b=1.0, solve_own=1e-12, solve_ref=1.0, retract=1e-9, trial_chi2=1e-9, scale=1e-9)   # H, b, solve_ref: multiples of the per-block tolerance


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


def unpack45(Hp):
    H = np.zeros((9, 9))
    H[np.triu_indices(9)] = Hp
    return H + np.triu(H, 1).T


def without_gravity(pkg, gf):
    """the frame's graph minus the gravity priors.  The lock-step comparison runs on this graph: the reference's gravity edge
    (BasicEllipsoidEdges.cpp:129-152) computes acos(c) of a cosine that LM drives to 1, and "if (c > 1) c -= 1e-4" turns a LAST-BIT
    excess of c over 1 into a residual of acos(1 - 1e-4) = 0.0141 rad: with its information of 1e4 that is a jump of 2.0 in chi2 and,
    through the central difference, a diagonal entry of 1e4 (0.0141 / 2e-6)^2 = 5e11 in H (measured on the GPU box, frame 1: the
    checker's max diag(H) 1.14e12 against 2.97e5 on the GPU at the same state -- lambda_0 = tau max diag follows).  Which side of 1
    the cosine lands on is decided by the rounding of a dot product and two norms, so no two implementations -- nor the checker at
    two values of delta -- agree on this edge at a converged state.  It is a property of the reference (SURVEY.md section 7:
    "numerically degenerate"), reproduced faithfully on both sides, and it is why the free-running comparison below carries a
    reproducibility criterion; the edge itself is held to the checker where it is well defined (tests/test_gpu_optimizer.py
    test_gravity_only_graph, every LM parity test with tilted ellipsoids)."""
    return pkg.Graph(gf.K, gf.n_cams, gf.n_objs, None, gf.bbox_cam, gf.bbox_obj, gf.bbox_meas, gf.bbox_weight, gf.e3d_cam, gf.e3d_obj, gf.e3d_meas,
                     gf.e3d_weight, (), gf.grav_normal, 0.0)


def half_turn_edges(gf, cams, objs, margin=1e-8):
    """Ellipsoids with a 3-D edge one of whose four yaw hypotheses is (within `margin` of cos = -1) a HALF TURN: there the reference's
    log (se3quat.h:229-266, no branch for theta -> pi) is 0/0 -- omega = theta / (2 sqrt(1 - d^2)) vee(R - R^T) with d rounded to a
    double, anything between 0 and inf at d = -1 + 1e-16 and still 1e-16 / (1 + d) in relative error next to it -- so which value the
    minimum over the hypotheses (Ellipsoid.cpp:92-117) sees there is a matter of an implementation's last bits.  Round 6 (VERDICT r5
    item 9): the lock-step comparison runs the product with esl_lm_params::e3d_half_turn = 1 (the minimum exactly as the reference
    writes it) against the UNCORRECTED checker and SKIPS what touches these edges; nothing edits the checker's output any more."""
    from oracle import np_oracle as npo
    hit = set()
    meas = gf.e3d_meas.reshape(-1, 10)
    for e in range(len(gf.e3d_cam)):
        k = int(gf.e3d_obj[e])
        Tcw = npo.T_from7(cams[gf.e3d_cam[e]]); To, _ = npo.obj_from10(objs[k]); Tm, _ = npo.obj_from10(meas[e])
        Tmw = npo.T_inv(Tcw) @ Tm
        for q in (-1, 0, 1, 2):
            a = q * np.pi / 2
            Rz = np.eye(4); Rz[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
            E = npo.T_inv(Tmw @ Rz) @ To
            if 0.5 * (np.trace(E[:3, :3]) - 1) < -1 + margin:
                hit.add(k)
    return hit


def lockstep_frame(pkg, po, cx, gf, cams, objs0, params):
    """One frame's LM run driven through the step API with g2o's control flow (optimization_algorithm_levenberg.cpp:61-164) and,
    at EVERY linearisation and EVERY trial, the checker put on the GPU's own current state.  No LM run is compared with another
    LM run here -- nothing can amplify a last-bit difference into another accept/reject sequence -- and every piece is compared in
    the form in which it is WELL CONDITIONED (a weakly observed ellipsoid's 9 x 9 block has a condition number of 1e7 .. 1e12, so
    the solution of its damped system moves by 1e-2 when the Jacobians move by 1e-7: the checker against itself does):
      linearise   chi2; every ellipsoid's H_oo block and b_o against the checker's (Frobenius norms; b = -sum J^T Omega r cancels to ~0
                  at a minimum, so its differencing noise is measured against |b| + sqrt(max H_ii chi2), its Cauchy-Schwarz bound).  The
                  tolerance of a block is 5e-6 + 3 x the checker's OWN spread of that block between delta and 0.9 delta; blocks
                  the checker does not reproduce itself to 1e-5 are counted (`noisy`) and not compared (with the gravity prior in
                  the graph that was most blocks of the early frames -- see without_gravity -- without it there should be none);
      solve       the GPU's step x as a solution of its OWN system, |(H + lambda I) x - b| / (|H + lambda I| |x| + |b|) per
                  ellipsoid (normwise backward error: LinearSolverDense's job, linear_solver_dense.h:65-113), and of the checker's
                  system with the block's tolerance and b's noise scale;
      retract     the GPU's trial state against the checker's ellipsoid::exp_update (Ellipsoid.cpp:38-47) of the GPU's x;
      errors      the checker's chi2 AT the GPU's trial state against the GPU's (computeActiveErrors / activeRobustChi2);
      scale       computeScale (optimization_algorithm_levenberg.cpp:182-189) from the GPU's own x and b.
    Mapping mode (all cameras fixed).  Returns the worst figures (H / b / solve_ref as multiples of their tolerance) and the run's
    decisions."""
    F, N = gf.n_cams, gf.n_objs
    cx.upload_graph(gf); cx.upload_states(cams, objs0)
    cx.lm_begin(params)
    worst = dict(lin_chi2=0.0, H=0.0, b=0.0, solve_own=0.0, solve_ref=0.0, retract=0.0, trial_chi2=0.0, scale=0.0, blocks=0, noisy=0, half_turn=0)
    lam, ni, nbad, it, ok_outer, trials, chi0 = -1.0, 2.0, 0, 0, True, [], 0.0
    nrm = lambda M: max(float(np.abs(M).max()), 1e-300)
    while it < params.max_iters and ok_outer:
        _, cur_objs = cx.download_states()
        lin = cx.lm_linearize()
        H, b, fidx, chi = po.build_system(gf, cams, cur_objs, delta=params.numeric_delta)
        H2, b2, _, _ = po.build_system(gf, cams, cur_objs, delta=0.9 * params.numeric_delta)
        fooled = half_turn_edges(gf, cams, cur_objs)   # (their blocks and, while there are any, the chi2 totals are not compared)
        worst["half_turn"] += len(fooled)
        Hg = cx.lm_download(0, N * 45).reshape(N, 45); bg = cx.lm_download(1, N * 9).reshape(N, 9)
        if it == 0:
            chi0 = chi                      # chi2 differences are measured against chi + 1e-9 of the frame's start value: ellipsoids held
                                            # by a single 3-D edge are fitted EXACTLY once the gravity prior is out of the graph
                                            # (chi2 -> 1e-20, where "relative" has no meaning)
        if not fooled:
            worst["lin_chi2"] = max(worst["lin_chi2"], abs(lin.chi2 - chi) / (chi + 1e-9 * chi0))
        tol = np.zeros(N); sb = np.zeros(N)
        for k in range(N):
            i = fidx[F + k]
            if i < 0:
                assert not Hg[k].any()
                continue
            Hr, br = H[i:i + 9, i:i + 9], b[i:i + 9]
            # b = -sum J^T Omega r cancels to ~0 at a minimum: its differencing noise scales with sqrt(H_ii chi2) (Cauchy-Schwarz), not with b
            # (... + 1e-7 max H_ii: a difference in b that moves the Gauss-Newton step by less than 1e-7 x the tolerance is none)
            sb[k] = np.linalg.norm(br) + np.sqrt(nrm(np.diag(Hr)) * chi) + 1e-7 * nrm(np.diag(Hr))
            spread = max(np.linalg.norm(H2[i:i + 9, i:i + 9] - Hr) / np.linalg.norm(Hr), np.linalg.norm(b2[i:i + 9] - br) / sb[k])
            worst["blocks"] += 1
            if k in fooled:                      # a half-turn hypothesis among its 3-D edges' four: the reference's value is 0/0 there -- skipped (counted in half_turn)
                tol[k] = np.inf
                continue
            if spread > 1e-5:                    # the checker does not reproduce this block itself: counted, not compared
                worst["noisy"] += 1
                tol[k] = np.inf
                continue
            tol[k] = 5e-6 + 3 * spread      # measured on the GPU box (all 60 frames): H 0.29, b 0.08, solve 0.08 of it
            worst["H"] = max(worst["H"], np.linalg.norm(unpack45(Hg[k]) - Hr) / np.linalg.norm(Hr) / tol[k])
            worst["b"] = max(worst["b"], np.linalg.norm(bg[k] - br) / sb[k] / tol[k])
        cur = ini = lin.chi2
        if it == 0:
            lam, ni, nbad = params.tau * lin.max_diag, 2.0, 0
            if np.isfinite(tol).all():
                assert lin.max_diag == pytest.approx(np.abs(np.diag(H)).max(), rel=1e-5)
        q, rho = 0, 0.0
        while True:
            tr = cx.lm_try_step(lam)
            xg = cx.lm_download(2, N * 9).reshape(N, 9)
            trial_gpu = cx.lm_download(7, N * 10).reshape(N, 10)
            trial_ref, sc_terms = cur_objs.copy(), []
            for k in range(N):
                i = fidx[F + k]
                if i < 0:
                    continue
                A = H[i:i + 9, i:i + 9] + lam * np.eye(9); Ag = unpack45(Hg[k]) + lam * np.eye(9)
                berr = lambda M, r, s: np.linalg.norm(M @ xg[k] - r) / (np.linalg.norm(M) * np.linalg.norm(xg[k]) + s + 1e-300)
                worst["solve_own"] = max(worst["solve_own"], berr(Ag, bg[k], np.linalg.norm(bg[k])))
                worst["solve_ref"] = max(worst["solve_ref"], berr(A, b[i:i + 9], sb[k]) / tol[k])
                trial_ref[k] = po.obj_oplus(cur_objs[k], xg[k])
                sc_terms.append(xg[k] * (lam * xg[k] + bg[k]))
            worst["retract"] = max(worst["retract"], float(per_obj_err(trial_gpu, trial_ref).max()))
            chi_ref = po.build_system(gf, cams, trial_gpu, delta=params.numeric_delta)[3]
            if not half_turn_edges(gf, cams, trial_gpu):
                worst["trial_chi2"] = max(worst["trial_chi2"], abs(tr.chi2 - chi_ref) / (chi_ref + 1e-9 * chi0))
            sc = np.concatenate(sc_terms) if sc_terms else np.zeros(1)
            worst["scale"] = max(worst["scale"], abs(tr.scale - sc.sum()) / max(np.abs(sc).sum(), 1e-300))
            tmp = tr.chi2 if tr.solve_ok else 1.7976931348623157e308
            rho = (cur - tmp) / (tr.scale + 1e-3)
            if rho > 0 and np.isfinite(tmp):
                lam *= max(1. / 3., min(1. - (2 * rho - 1) ** 3, 2. / 3.)); ni = 2.0; cur = tmp
                cx.lm_commit(True)
            else:
                lam *= ni; ni *= 2
                cx.lm_commit(False)
            q += 1
            if not (rho < 0 and q < params.max_trials):
                break
        trials.append(q)
        it += 1
        if q == params.max_trials or rho == 0:
            ok_outer = False
        else:
            nbad = nbad + 1 if (ini - cur) * 1e3 < ini else 0
            ok_outer = nbad < 3
    return worst, dict(iterations=it, trace_trials=trials, chi2_final=cur)


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
    lock_worst = dict(lin_chi2=0.0, H=0.0, b=0.0, solve_own=0.0, solve_ref=0.0, retract=0.0, trial_chi2=0.0, scale=0.0)
    lock_blocks = lock_noisy = lock_half = 0
    lock_failures = []
    n_reproducible = 0
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
        # (c) the checker, on EVERY frame (round 4: none is left out), numeric Jacobians at delta = 1e-6 on both sides:
        #   (c1) LOCK-STEP: every linearisation and every trial step of the GPU's own LM run against the checker at the same state
        #        (lockstep_frame) -- the parity statement proper, immune to the chaos of (c2);
        #   (c2) FREE-RUNNING: the two complete LM runs from the same start.  Evaluation parity always (the checker's chi2 at the
        #        GPU's end state).  End states to the north star's 1e-4 wherever the reference's OWN result is reproducible: its
        #        run converged before the 10-iteration cap and moves < 1e-5 when only its differencing step changes by 10 %.
        #        In the first ~27 frames it often is not: ellipsoids with three or four bbox edges from one viewpoint, or a single
        #        3-D edge, sit in flat valleys, LM's lambda and accept/reject are GLOBAL (one weak ellipsoid redirects the run of
        #        all), and the runs stop at the iteration cap mid-way -- measured on the GPU box, the checker against ITSELF at
        #        delta 0.9e-6 / 1.1e-6: up to 2e-2 (frame 11), 3e-3 (frames 10, 19).  The yaw-hypothesis explanation of round 3
        #        did not survive measurement: the two best hypotheses of every offending 3-D edge are 20x .. 190x apart.
        pn = pkg.default_lm_params(jacobian_mode=0, numeric_delta=1e-6)
        # (lock-step: the product with the 3-D edge's minimum AS THE REFERENCE WRITES IT, e3d_half_turn = 1, against the uncorrected checker)
        worst, r_lock = lockstep_frame(pkg, po, ctx, without_gravity(pkg, gf), c[:f + 1], objs_before,
                                       pkg.default_lm_params(jacobian_mode=0, numeric_delta=1e-6, e3d_half_turn=1))
        for k in lock_worst:
            lock_worst[k] = max(lock_worst[k], worst[k])
        lock_blocks += worst["blocks"]; lock_noisy += worst["noisy"]; lock_half += worst["half_turn"]
        bad = {k: float(worst[k]) for k in LOCK_TOL if not worst[k] < LOCK_TOL[k]}
        if bad:
            lock_failures.append("frame %d: %s" % (f, {k: "%.2e" % v for k, v in bad.items()}))
        _, o_orc, r_orc = po.optimize(gf, c[:f + 1], objs_before, pn, solver=1)
        _, o_gpu, r_gpu = ctx.optimize(gf, c[:f + 1], objs_before, pn)
        assert r_gpu["n_bbox_valid"] == r_orc["n_bbox_valid"]
        chi_at_gpu = po.build_system(gf, c[:f + 1], o_gpu, delta=1e-6)[3]
        assert r_gpu["chi2_final"] == pytest.approx(chi_at_gpu, rel=1e-9), f
        self_err = 0.0
        for d in (0.9e-6, 1.1e-6):
            _, o_alt, _ = po.optimize(gf, c[:f + 1], objs_before, pkg.default_lm_params(jacobian_mode=0, numeric_delta=d), solver=1)
            self_err = max(self_err, float(per_obj_err(o_alt, o_orc).max()))
        err = float(per_obj_err(o_gpu, o_orc).max())
        reproducible = self_err < 1e-5 and r_orc["iterations"] < pn.max_iters and r_gpu["iterations"] < pn.max_iters
        n_reproducible += reproducible
        print("streaming frame %2d: lock-step H %.2f b %.2f solve %.2f of tolerance (%d of %d blocks noisy) | free-running GPU vs checker chi2 rel %.2e states %.2e, checker vs itself %.2e, iterations %d / %d%s"
              % (f, worst["H"], worst["b"], worst["solve_ref"], worst["noisy"], worst["blocks"], abs(r_gpu["chi2_final"] / r_orc["chi2_final"] - 1), err, self_err, r_gpu["iterations"],
                 r_orc["iterations"], "" if reproducible else "  (reference not reproducible here)"))
        if reproducible:
            assert err < 1e-4, (f, err)
            assert r_gpu["chi2_final"] == pytest.approx(r_orc["chi2_final"], rel=1e-6), f
    print("streaming, all %d frames in lock-step (e3d_half_turn = 1 vs the uncorrected checker): worst %s; %d of %d linearised blocks the checker does not reproduce itself to 1e-5, "
          "%d skipped for a half-turn hypothesis (0/0 in the reference); %d frames where the reference's LM run reproduces itself"
          % (n_frames, {k: "%.1e" % v for k, v in lock_worst.items()}, lock_noisy, lock_blocks, lock_half, n_reproducible))
    assert not lock_failures, "lock-step beyond tolerance: " + "; ".join(lock_failures)
    assert n_reproducible >= 30 and lock_noisy <= 0.02 * lock_blocks and lock_half <= 0.1 * lock_blocks
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
    # SLAM-mode graphs are accepted since ABI 4 (test_slam_mode_append_equals_rebuild); a bad odometry index is not
    gs, cs, os_, _ = pkg.synth.make_graph(10, 2, 30, seed=1, slam=True)
    cx.upload_graph(gs); cx.upload_states(cs, os_)
    with pytest.raises(pkg.EslError, match="odometry edge index"):
        cx.append_graph(new_cams=cs[:1], new_cam_fixed=[0], odom=([9], [11], [[0, 0, 0, 0, 0, 0, 1.0]]))
    cx.append_graph(new_cams=cs[-1:], new_cam_fixed=[0], odom=([9], [10], [[0, 0, 0, 0, 0, 0, 1.0]]))
    assert cx.graph_sizes()["n_cams"] == 11
    assert np.isfinite(cx.optimize_resident(pkg.default_lm_params(jacobian_mode=1))["chi2_final"])
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


def test_half_turn_yaw_hypothesis_is_never_taken(pkg, po, ctx):
    """Regression of a state the streaming sequence reaches (round 4, found by the lock-step comparison).  Ellipsoid 15 is held by ONE
    3-D edge (camera 0) and the gravity prior until frame 17: LM matches the measurement's yaw to rounding and keeps a tilt residual
    of 1.88e-2 rad against gravity.  The yaw + 180 degrees hypothesis of min_log_error_9dof (Ellipsoid.cpp:92-117) is then a rotation
    by pi - 1e-9: the reference's SE3Quat::log (se3quat.h:229-266) has no branch for that, its omega there is pi delta / sqrt(1 - d^2)
    with d = cos(theta) rounded to a double -- between 0 and inf.  The product used to take that hypothesis at 7e-13 (chi2 of the edge 0
    instead of 3.21, analytic H entries of 1e16 .. 1e20); both restatements keep the true minimum.  The fixture holds the ellipsoid
    states the GPU run had in front of frame 11 (tests/golden/half_turn_hypothesis.npz); graph, cameras and measurements are the
    synthetic sequence's."""
    import os
    from oracle import np_oracle as npo
    g, c, _, _ = pkg.synth.make_graph(60, 20, 20 * 60, seed=3)
    gf = graph_upto(pkg, g, 11)
    before = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "half_turn_hypothesis.npz"))["objs_before_frame11"]
    k = 15
    sub = gf.subset_objects([k])
    assert len(sub.bbox_cam) == 0 and len(sub.e3d_cam) == 1
    # the configuration: hypothesis yaw + 180 degrees is a half turn, the true minimum is the plain one at 1.88e-2
    Tcw = npo.T_from7(c[sub.e3d_cam[0]]); To, s = npo.obj_from10(before[k]); Tm, sm = npo.obj_from10(sub.e3d_meas.reshape(-1, 10)[0])
    Rz = np.diag([-1.0, -1.0, 1.0, 1.0])
    E = npo.T_inv(npo.T_inv(Tcw) @ Tm @ Rz) @ To
    assert 0.5 * (np.trace(E[:3, :3]) - 1) < -1 + 1e-12
    r_ref = po.res_e3d(c[sub.e3d_cam[0]], before[k], sub.e3d_meas.reshape(-1, 10)[0])
    assert np.linalg.norm(r_ref) == pytest.approx(1.8779e-2, rel=1e-3)
    for gs in (without_gravity(pkg, sub), sub):
        H, b, fidx, chi = po.build_system(gs, c[:12], before[k:k + 1], delta=1e-6)
        for jac in (0, 1):
            ctx.upload_graph(gs); ctx.upload_states(c[:12], before[k:k + 1])
            ctx.lm_begin(pkg.default_lm_params(jacobian_mode=jac, numeric_delta=1e-6))
            lin = ctx.lm_linearize()
            Hg, bg = unpack45(ctx.lm_download(0, 45)), ctx.lm_download(1, 9)
            assert lin.chi2 == pytest.approx(chi, rel=1e-9)
            np.testing.assert_allclose(bg, b, atol=2e-6 * np.abs(b).max() + 1e-6)
            np.testing.assert_allclose(Hg, H, atol=5e-6 * np.abs(H).max())


def test_all_four_yaw_hypotheses_half_turns_do_not_leave_the_edge_without_a_residual(pkg, po, ctx):
    """the half-turn exclusion above must leave something to take.  An estimate that is the measurement turned by pi about a
    HORIZONTAL axis (upside-down) makes E_k = Rz_k^T E_0 a half turn for EVERY yaw hypothesis; round 4's code then took hypothesis 0
    unconditionally with best = DBL_MAX bookkeeping.  Now all four stay eligible and the first minimum as written wins -- what the
    reference and the checker compute there.  The log of a half turn is ill-defined in the reference (a division by sin(pi)), so the
    VALUES are not comparable; what must hold: the call returns, the edge's chi2 is either non-finite on both sides or at least
    weight * (pi - 1e-3)^2 (a half turn is never a small residual), and an optimisation from there leaves finite states."""
    from oracle import np_oracle as npo
    K = pkg.synth.TUM3_K
    cam = np.array([[0, 0, 0, 0, 0, 0, 1.0]])
    est = np.array([[0.2, -0.1, 2.5, 0, 0, 0, 1.0, 0.3, 0.25, 0.4]])
    meas = est[0].copy()
    meas[3:7] = [1.0, 0.0, 0.0, 0.0]            # the same ellipsoid turned by pi about its x axis (camera frame = world: Tcw = I)
    g = pkg.Graph(K, 1, 1, None, e3d_cam=[0], e3d_obj=[0], e3d_meas=[meas], e3d_weight=[1e4])
    for q in (-1, 0, 1, 2):                      # the configuration: every hypothesis is a half turn
        a = q * np.pi / 2
        Rz = np.eye(4); Rz[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
        E = npo.T_inv(npo.obj_from10(meas)[0] @ Rz) @ npo.obj_from10(est[0])[0]
        assert 0.5 * (np.trace(E[:3, :3]) - 1) < -1 + 1e-12
    with np.errstate(all="ignore"):
        r_ref = po.res_e3d(cam[0], est[0], meas)
    for jac in (0, 1):
        ctx.upload_graph(g); ctx.upload_states(cam, est)
        ctx.lm_begin(pkg.default_lm_params(jacobian_mode=jac, numeric_delta=1e-6))
        chi = ctx.lm_linearize().chi2
        print("all four hypotheses half turns (jac %d): GPU chi2 %r, checker |r|^2 w %r" % (jac, chi, 1e4 * float(np.dot(r_ref, r_ref))))
        assert (not np.isfinite(chi)) or chi >= 1e4 * (np.pi - 1e-3) ** 2
        _, oo, rep = ctx.optimize(g, cam, est, pkg.default_lm_params(jacobian_mode=jac, numeric_delta=1e-6))
        assert np.all(np.isfinite(oo))


def slam_graph_upto(pkg, g, f):
    """the SLAM-mode graph of frames 0 .. f: camera 0 fixed, one odometry edge per consecutive pair (Optimizer.cpp:126-158)"""
    mb, me, mo = g.bbox_cam <= f, g.e3d_cam <= f, g.odom_j <= f
    return pkg.Graph(g.K, f + 1, g.n_objs, g.cam_fixed[:f + 1], g.bbox_cam[mb], g.bbox_obj[mb], g.bbox_meas.reshape(-1, 4)[mb], g.bbox_weight[mb],
                     g.e3d_cam[me], g.e3d_obj[me], g.e3d_meas.reshape(-1, 10)[me], g.e3d_weight[me], g.grav_obj, g.grav_normal, g.grav_weight,
                     g.odom_i[mo], g.odom_j[mo], g.odom_meas.reshape(-1, 7)[mo])


@pytest.mark.parametrize("solver", [1, 2])
def test_slam_mode_append_equals_rebuild(pkg, ctx, solver):
    """esl_graph_append in SLAM mode (ABI 4): a frame's delta = one FREE camera, its odometry edge to the previous
    frame and its bbox / 3-D edges, into the device-resident graph; the camera-indexed tables and the solver's lists are rebuilt from
    the host's mirror of the layout.  After every frame the appended graph must BE the graph esl_graph_upload makes of the
    concatenated edges: same LM trace and states (same edge order inside an ellipsoid and inside a camera -> same sums), with the
    reduced camera system and with the camera-first elimination.  The first append re-lays the compact upload out with slack (one
    relayout); the later frames go into the slack.  Measured (round 5): linearisation, damped solve and the whole LM run are
    BIT-IDENTICAL on all 43 frames with both eliminations."""
    F, N = 48, 8
    g, c, o, _ = pkg.synth.make_graph(F, N, 14 * F, seed=9, slam=True)
    p = pkg.default_lm_params(jacobian_mode=1, max_iters=3, linear_solver=solver)
    f0 = 4
    ctx.upload_graph(slam_graph_upto(pkg, g, f0)); ctx.upload_states(c[:f0 + 1], o)
    ctx.optimize_resident(p)
    ref = pkg.Context(0)
    worst = dict(chi2=0.0, cams=0.0, objs=0.0)
    try:
        for f in range(f0 + 1, F):
            mb, me, mo = g.bbox_cam == f, g.e3d_cam == f, g.odom_j == f
            ctx.append_graph(new_cams=c[f:f + 1], new_cam_fixed=[0],
                             bbox=(g.bbox_cam[mb], g.bbox_obj[mb], g.bbox_meas.reshape(-1, 4)[mb], g.bbox_weight[mb]),
                             e3d=(g.e3d_cam[me], g.e3d_obj[me], g.e3d_meas.reshape(-1, 10)[me], g.e3d_weight[me]),
                             odom=(g.odom_i[mo], g.odom_j[mo], g.odom_meas.reshape(-1, 7)[mo]))
            cams_in, objs_in = ctx.download_states()
            assert cams_in.shape == (f + 1, 7) and np.array_equal(cams_in[f], c[f])
            ref.upload_graph(slam_graph_upto(pkg, g, f)); ref.upload_states(cams_in, objs_in)
            # the linearisation and one damped solve, quantity by quantity: the appended layout must give the SAME BITS as the compact one
            # (same edge order inside every ellipsoid, every camera and every chunk -> same sums)
            nf = f
            lin = []
            for cx_ in (ctx, ref):
                cx_.lm_begin(p)
                part = cx_.lm_linearize()
                tr = cx_.lm_try_step(1e-4 * part.max_diag)
                lin.append(dict(chi2=part.chi2, max_diag=part.max_diag, Hoo=cx_.lm_download(0, N * 45), bo=cx_.lm_download(1, N * 9), Hcc=cx_.lm_download(3, nf * 36),
                                bc=cx_.lm_download(4, nf * 6), xc=cx_.lm_download(5, nf * 6), xo=cx_.lm_download(2, N * 9), trial_chi2=tr.chi2, ok=tr.solve_ok))
                cx_.lm_commit(False)
            for key in ("chi2", "max_diag", "Hoo", "bo", "Hcc", "bc", "xc", "xo", "trial_chi2", "ok"):
                a_, b_ = np.asarray(lin[0][key]), np.asarray(lin[1][key])
                assert np.array_equal(a_, b_), "frame %d: %s of the appended graph differs from the rebuilt one by %.3e (relative to %.3e)" % (
                    f, key, float(np.abs(a_ - b_).max()), float(np.abs(b_).max()))
            ra = ctx.optimize_resident(p)
            ca, oa = ctx.download_states()
            assert ctx.lm_solver_used() == solver
            rr = ref.optimize_resident(p)
            cr, orr = ref.download_states()
            assert ra["trace_trials"] == rr["trace_trials"] and ra["n_bbox_valid"] == rr["n_bbox_valid"], (f, ra["trace_trials"], rr["trace_trials"], ra["trace_chi2"], rr["trace_chi2"])
            worst["chi2"] = max(worst["chi2"], float(np.abs(np.array(ra["trace_chi2"]) / np.array(rr["trace_chi2"]) - 1).max()))
            worst["cams"] = max(worst["cams"], float(np.abs(ca - cr).max())); worst["objs"] = max(worst["objs"], float(np.abs(oa - orr).max()))
        sz = ctx.graph_sizes()
        print("SLAM-mode append vs rebuild, %d frames, solver %d: worst chi2 trace rel %.2e, cameras %.2e, ellipsoids %.2e; relayouts %d" % (
            F - f0 - 1, solver, worst["chi2"], worst["cams"], worst["objs"], sz["relayouts"]))
        assert sz["n_cams"] == F and sz["relayouts"] >= 1
        assert worst["chi2"] < 1e-11 and worst["cams"] < 1e-10 and worst["objs"] < 1e-10
    finally:
        ref.close()


def test_e3d_half_turn_switch_is_inert_away_from_half_turns_and_validated(pkg, ctx):
    """esl_lm_params::e3d_half_turn (ABI 5): 1 takes the minimum over the four yaw hypotheses exactly as
    Ellipsoid.cpp:92-117 writes it, 0 (default) excludes hypotheses within 1.4e-6 rad of a half turn.  On a graph without such a
    hypothesis the two are the SAME run, bit for bit; on the half-turn fixture (ellipsoid 15 in front of frame 11) the default keeps
    the true minimum and the as-written form is whatever 0/0 gives -- finite or not, but never compared; any other value is refused."""
    import os
    g, c, o, _ = pkg.synth.make_graph(30, 6, 240, seed=5)
    for jac in (0, 1):
        r = [ctx.optimize(g, c, o, pkg.default_lm_params(jacobian_mode=jac, numeric_delta=1e-6, e3d_half_turn=h)) for h in (0, 1)]
        assert r[0][2]["trace_chi2"] == r[1][2]["trace_chi2"] and r[0][2]["trace_trials"] == r[1][2]["trace_trials"]
        np.testing.assert_array_equal(r[0][1], r[1][1])
    with pytest.raises(pkg.EslError, match="e3d_half_turn"):
        ctx.optimize(g, c, o, pkg.default_lm_params(e3d_half_turn=2))
    gs, cs, _, _ = pkg.synth.make_graph(60, 20, 20 * 60, seed=3)
    sub = without_gravity(pkg, graph_upto(pkg, gs, 11).subset_objects([15]))
    before = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "half_turn_hypothesis.npz"))["objs_before_frame11"]
    chi = []
    for h in (0, 1):
        ctx.upload_graph(sub); ctx.upload_states(cs[:12], before[15:16])
        ctx.lm_begin(pkg.default_lm_params(jacobian_mode=1, e3d_half_turn=h))
        chi.append(ctx.lm_linearize().chi2)
    print("half-turn fixture, chi2 of the one 3-D edge: guarded %.6g, as written %r" % (chi[0], chi[1]))
    assert chi[0] == pytest.approx(sub.e3d_weight[0] * 1.8779e-2 ** 2, rel=2e-3)
