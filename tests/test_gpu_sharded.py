"""Sharded optimisation inside the library (SURVEY.md §8 e) on ONE GPU: two contexts, one shard each, driven by
two host threads; the exchange runs over the host-staged transport (esl_comm_init_host) with a thread barrier
standing in for the wire.  The sharded run must reproduce the single-context run: same accept/reject sequence,
same chi2 trace (up to summation order), same final states.

Mapping mode: only the LM scalars cross.  SLAM mode: the camera blocks Hcc, b_c and the partial reduced camera
systems are summed over the shards; odometry, lambda I and the camera part of the LM scale are counted once."""
import threading

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class ThreadAllreduce:
    """In-place sum over `n` threads, rank order fixed (rank 0 adds 0,1,2,...), result identical on all ranks."""

    def __init__(self, n):
        self.n = n
        self.bar = threading.Barrier(n, timeout=60)
        self.bufs = [None] * n
        self.total = None
        self.calls = [0] * n
        self.doubles = [0] * n

    def make(self, rank):
        def fn(arr):
            self.calls[rank] += 1
            self.doubles[rank] += arr.size
            self.bufs[rank] = arr
            self.bar.wait()
            if rank == 0:
                sizes = {b.size for b in self.bufs}
                assert len(sizes) == 1, "ranks disagree on the collective's size: %r" % sizes
                tot = self.bufs[0].copy()
                for r in range(1, self.n):
                    tot += self.bufs[r]
                self.total = tot
            self.bar.wait()
            arr[:] = self.total
            self.bar.wait()
        return fn


def run_sharded(pkg, g, cams, objs, params, n_shards=2):
    part = pkg.lib.partition_objects(g, n_shards)
    ar = ThreadAllreduce(n_shards)
    ctxs, ids, reps, errs = [], [], [None] * n_shards, [None] * n_shards
    for r in range(n_shards):
        idx = np.nonzero(part == r)[0]
        c = pkg.Context(0)
        c.upload_graph(g.subset_objects(idx)); c.upload_states(cams, objs[idx])
        c.comm_init_host(n_shards, r, ar.make(r))
        ctxs.append(c); ids.append(idx)

    def work(r):
        try:
            reps[r] = ctxs[r].optimize_resident(params)
        except Exception as e:   # surfaced below; a dead rank must not leave the other one in the barrier
            errs[r] = e
            ar.bar.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(n_shards)]
    [t.start() for t in th]
    [t.join(120) for t in th]
    assert not any(t.is_alive() for t in th), "sharded run hung"
    assert errs == [None] * n_shards, errs
    out_objs = np.zeros_like(objs)
    out_cams = []
    for r, c in enumerate(ctxs):
        cc, oo = c.download_states()
        out_objs[ids[r]] = oo
        out_cams.append(cc)
        c.comm_destroy(); c.close()
    return reps, out_cams, out_objs, ar


def test_mapping_two_shards_match_single_context(pkg, ctx):
    g, c, o, _ = pkg.synth.make_graph(40, 10, 400, seed=11)
    p = pkg.default_lm_params(jacobian_mode=1)
    ctx.upload_graph(g); ctx.upload_states(c, o)
    ref = ctx.optimize_resident(p)
    _, ro = ctx.download_states()
    reps, cams, objs, ar = run_sharded(pkg, g, c, o, p)
    for rep in reps:
        assert rep["iterations"] == ref["iterations"]
        assert rep["trace_trials"] == ref["trace_trials"]
        np.testing.assert_allclose(rep["trace_chi2"], ref["trace_chi2"], rtol=1e-12)
    assert reps[0]["trace_chi2"] == reps[1]["trace_chi2"]          # bit-identical decisions on every rank
    assert reps[0]["trace_lambda"] == reps[1]["trace_lambda"]
    # block-diagonal system: an ellipsoid's arithmetic does not depend on which shard holds it.  (Not bit-equal: the
    # single-context run is device-driven and its first linearisation is the instantiation that also validates the
    # bbox edges, the sharded run is host-driven -- two compilations of the same formulas, last-ulp differences.)
    np.testing.assert_allclose(objs, ro, rtol=1e-11, atol=1e-13)
    assert ar.calls[0] == ar.calls[1] > 0


@pytest.mark.parametrize("jac", [0, 1])
def test_slam_two_shards_match_single_context(pkg, ctx, jac):
    g, c, o, _ = pkg.synth.make_graph(30, 8, 300, seed=5, slam=True)
    p = pkg.default_lm_params(jacobian_mode=jac, numeric_delta=1e-6)
    ctx.upload_graph(g); ctx.upload_states(c, o)
    ref = ctx.optimize_resident(p)
    rc, ro = ctx.download_states()
    assert ref["chi2_final"] < 0.5 * ref["chi2_initial"]
    reps, cams, objs, ar = run_sharded(pkg, g, c, o, p)
    n = len(ref["trace_chi2"])
    for rep in reps:
        assert rep["chi2_initial"] == pytest.approx(ref["chi2_initial"], rel=1e-12)
        # same trajectory while chi2 is still moving.  The partial sums are formed in another association (per shard, then
        # summed over the shards): ~1e-13 relative on the first iterations, growing along the trajectory (numeric
        # Jacobians amplify it).  Each side is reproducible run to run since the Schur complement lost its atomics.
        k = next((i for i in range(1, n) if abs(ref["trace_chi2"][i] - ref["trace_chi2"][i - 1]) < 1e-7 * ref["trace_chi2"][i]), n)
        k = max(k, 3)
        np.testing.assert_allclose(rep["trace_chi2"][:2], ref["trace_chi2"][:2], rtol=1e-7)   # x cond(S) already after one step
        np.testing.assert_allclose(rep["trace_chi2"][:k], ref["trace_chi2"][:k], rtol=1e-6)
        assert rep["trace_trials"][:k] == ref["trace_trials"][:k]
        assert rep["chi2_final"] == pytest.approx(ref["chi2_final"], rel=1e-4)   # converged tail: accept / reject there is round-off
    assert reps[0]["trace_chi2"] == reps[1]["trace_chi2"]
    assert reps[0]["trace_lambda"] == reps[1]["trace_lambda"]
    # every rank solved the same summed camera system: identical camera trajectories on both, equal to single-GPU
    np.testing.assert_array_equal(cams[0], cams[1])
    np.testing.assert_allclose(cams[0], rc, atol=2e-7)
    np.testing.assert_allclose(objs, ro, atol=2e-6)
    # traffic: the reduced system dominates what crosses (lda*n doubles per trial)
    assert ar.doubles[0] == ar.doubles[1] > 0


def test_slam_first_reduced_system_is_the_sum_of_the_shards(pkg, ctx):
    """One linearisation + one reduced system, compared entry by entry with the single-context one."""
    g, c, o, _ = pkg.synth.make_graph(24, 6, 200, seed=9, slam=True)
    p = pkg.default_lm_params(jacobian_mode=1)
    ctx.upload_graph(g); ctx.upload_states(c, o)
    ctx.lm_begin(p)
    lin = ctx.lm_linearize()
    lam = 1e-5 * lin.max_diag
    _, n, lda = ctx.lm_reduced_system(lam)
    S_ref = ctx.lm_download(6, lda * n)
    Hcc_ref = ctx.lm_download(3, n * 6)

    part = pkg.lib.partition_objects(g, 2)
    ar = ThreadAllreduce(2)
    out = [None, None]
    errs = [None, None]

    def work(r):
        try:
            idx = np.nonzero(part == r)[0]
            cx = pkg.Context(0)
            cx.upload_graph(g.subset_objects(idx)); cx.upload_states(c, o[idx])
            cx.comm_init_host(2, r, ar.make(r))
            cx.lm_begin(p)
            l2 = cx.lm_linearize()
            cx.lm_reduced_system(lam)
            out[r] = (l2.chi2, l2.max_diag, cx.lm_download(6, lda * n), cx.lm_download(3, n * 6))
            cx.comm_destroy(); cx.close()
        except Exception as e:
            errs[r] = e
            ar.bar.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join(120) for t in th]
    assert errs == [None, None], errs
    for r in range(2):
        chi2, md, S, Hcc = out[r]
        assert chi2 == pytest.approx(lin.chi2, rel=1e-12)
        assert md == pytest.approx(lin.max_diag, rel=1e-12)
        np.testing.assert_allclose(Hcc, Hcc_ref, rtol=1e-11, atol=1e-9 * np.abs(Hcc_ref).max())
        np.testing.assert_allclose(S, S_ref, rtol=0, atol=1e-12 * np.abs(S_ref).max())
    np.testing.assert_array_equal(out[0][2], out[1][2])


@pytest.mark.parametrize("slam", [False, True])
def test_rccl_single_rank_communicator_matches_plain_run(pkg, ctx, slam):
    """The RCCL transport itself (ncclAllGather / ncclAllReduce enqueued on the context's stream) with a 1-rank
    communicator: same LM run as without a communicator (the partial sums are formed in a different order: ~1e-15)."""
    g, c, o, _ = pkg.synth.make_graph(30, 8, 300, seed=21, slam=slam)
    p = pkg.default_lm_params(jacobian_mode=1)
    ctx.upload_graph(g); ctx.upload_states(c, o)
    ref = ctx.optimize_resident(p)
    rc, ro = ctx.download_states()
    cx = pkg.Context(0)
    try:
        cx.upload_graph(g); cx.upload_states(c, o)
        cx.comm_init(1, 0, pkg.lib.comm_unique_id())
        rep = cx.optimize_resident(p)
        cc, oo = cx.download_states()
        # a second collective run on the same communicator (state of the exchange buffers carries nothing over)
        cx.upload_states(c, o)
        rep2 = cx.optimize_resident(p)
        cx.comm_destroy()
    finally:
        cx.close()
    for r in (rep, rep2):
        assert r["iterations"] == ref["iterations"]
        assert r["trace_trials"] == ref["trace_trials"]
        np.testing.assert_allclose(r["trace_chi2"], ref["trace_chi2"], rtol=1e-9)
        assert r["stop_reason"] == ref["stop_reason"]
    np.testing.assert_allclose(oo, ro, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(cc, rc, rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("slam", [False, True, 2])
def test_rccl_two_ranks_on_two_gpus(pkg, ctx, slam, tmp_path):
    """The real thing when the box has it: two processes, one GPU each, ncclAllGather / ncclAllReduce over the fabric between
    them (esl_comm_init with a 2-rank communicator).  Must take the single-context run's decisions on both ranks.  Skipped on
    a one-GPU box (the driver's GPU test box has one; the 8-GPU bench exercises the same code through bench.py --gpus N)."""
    import subprocess
    import sys
    if pkg.lib.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    g, c, o, _ = pkg.synth.make_graph(120 if slam else 30, 10 if slam else 8, 1500 if slam else 300, seed=21, slam=slam)
    p = pkg.default_lm_params(jacobian_mode=1)
    ctx.upload_graph(g); ctx.upload_states(c, o)
    ref = ctx.optimize_resident(p)
    rc, ro = ctx.download_states()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_two_rank_worker.py")
    id_file = str(tmp_path / "nccl_id.bin")
    env = dict(os.environ, ESL_CHOL_DIST="1")      # SLAM: ncclReduce per outer panel + ncclBroadcast of the factored panels
    procs = [subprocess.Popen([sys.executable, worker, str(r), id_file, str(tmp_path / f"rank{r}.npz"), str(int(slam))], env=env)
             for r in range(2)]
    for pr in procs:
        assert pr.wait(timeout=180) == 0
    outs = [np.load(str(tmp_path / f"rank{r}.npz")) for r in range(2)]
    objs = np.zeros_like(ro)
    for z in outs:
        assert int(z["iterations"]) == ref["iterations"] and list(z["trace_trials"]) == ref["trace_trials"]
        np.testing.assert_allclose(z["trace_chi2"], ref["trace_chi2"], rtol=1e-9 if not slam else 1e-7)
        objs[z["idx"]] = z["objs"]
    np.testing.assert_array_equal(outs[0]["trace_chi2"], outs[1]["trace_chi2"])      # bit-identical decisions on both ranks
    np.testing.assert_allclose(objs, ro, atol=1e-9 if not slam else 1e-6)
    if slam:
        np.testing.assert_array_equal(outs[0]["cams"], outs[1]["cams"])
        np.testing.assert_allclose(outs[0]["cams"], rc, atol=1e-6)


@pytest.mark.parametrize("n_shards", [2, 3])
def test_slam_distributed_factorisation_matches_replicated(pkg, ctx, monkeypatch, n_shards):
    """ESL_CHOL_DIST=1: every outer panel of the reduced camera system is reduced to its owner, factored there, broadcast, and
    each shard updates only its own later panels (esl_chol.hpp, CholDist) -- against the same sharded run with the replicated
    factorisation and against the single-context run.  714 unknowns = 6 inner / 3 outer panels, dealt 0, 1, 0 (2 shards) or
    0, 1, 2 (3 shards)."""
    g, c, o, _ = pkg.synth.make_graph(120, 10, 1500, seed=33, slam=True)
    p = pkg.default_lm_params(jacobian_mode=1)
    ctx.upload_graph(g); ctx.upload_states(c, o)
    ref = ctx.optimize_resident(p)
    rc, ro = ctx.download_states()
    assert ref["chi2_final"] < 0.5 * ref["chi2_initial"]
    monkeypatch.setenv("ESL_CHOL_DIST", "0")
    reps_r, cams_r, objs_r, ar_r = run_sharded(pkg, g, c, o, p, n_shards=n_shards)
    monkeypatch.setenv("ESL_CHOL_DIST", "1")
    reps_d, cams_d, objs_d, ar = run_sharded(pkg, g, c, o, p, n_shards=n_shards)
    # the distributed path really ran: 3 reductions + 3 broadcasts (one packed message per outer panel) per trial instead of one
    # all-reduce of the whole system
    assert ar.calls[0] == ar_r.calls[0] + 5 * sum(reps_d[0]["trace_trials"])
    for rep in reps_d:
        assert rep["iterations"] == ref["iterations"] and rep["trace_trials"] == ref["trace_trials"]
        np.testing.assert_allclose(rep["trace_chi2"], reps_r[0]["trace_chi2"], rtol=1e-9)     # same sums, another elimination order
        np.testing.assert_allclose(rep["trace_chi2"][:3], ref["trace_chi2"][:3], rtol=1e-6)
    for r in range(1, n_shards):
        assert reps_d[r]["trace_chi2"] == reps_d[0]["trace_chi2"]                               # identical decisions on every rank
        np.testing.assert_array_equal(cams_d[r], cams_d[0])                                     # ... and identical camera updates
    np.testing.assert_allclose(cams_d[0], cams_r[0], atol=1e-9)
    np.testing.assert_allclose(objs_d, objs_r, atol=1e-9)
    np.testing.assert_allclose(cams_d[0], rc, atol=2e-6)
    np.testing.assert_allclose(objs_d, ro, atol=2e-5)


def run_replicated(pkg, g, cams, objs, params, n_ranks=2):
    """n contexts on one GPU, EVERY one with the whole graph, communicator in replicated-graph mode (esl_comm_set_replicated)"""
    ar = ThreadAllreduce(n_ranks)
    ctxs, reps, errs, used = [], [None] * n_ranks, [None] * n_ranks, [0] * n_ranks
    for r in range(n_ranks):
        c = pkg.Context(0)
        c.comm_init_host(n_ranks, r, ar.make(r))
        c.comm_set_replicated(True)                      # the mode first, then the graph (ABI 4: no mode change under a resident graph)
        c.upload_graph(g); c.upload_states(cams, objs)
        ctxs.append(c)

    def work(r):
        try:
            reps[r] = ctxs[r].optimize_resident(params)
            used[r] = ctxs[r].lm_solver_used()
        except Exception as e:   # noqa: BLE001
            errs[r] = e
            ar.bar.abort()
    th = [threading.Thread(target=work, args=(r,)) for r in range(n_ranks)]
    [t.start() for t in th]
    [t.join(120) for t in th]
    assert not any(t.is_alive() for t in th), "replicated run hung"
    assert errs == [None] * n_ranks, errs
    states = [c.download_states() for c in ctxs]
    for c in ctxs:
        c.comm_destroy(); c.close()
    return reps, states, used, ar


@pytest.mark.parametrize("n_ranks,solver,sparse", [(2, 0, False), (3, 0, False), (2, 1, False), (2, 2, True), (3, 2, True)])
def test_slam_replicated_graph_divides_the_dense_solve(pkg, ctx, monkeypatch, n_ranks, solver, sparse):
    """esl_comm_set_replicated: every rank holds the whole graph, nothing is summed over ranks, the ranks divide the dense solve --
    camera-first elimination (solver AUTO -> 2): each rank forms only the outer panels of the reduced ELLIPSOID system it owns,
    the owner factors a panel and broadcasts it, own later panels updated, back-substitution replicated; reduced camera system
    (solver 1): built everywhere, factored distributed.  ESL_CHOL_DIST=1 forces the distribution at this size (40 ellipsoids =
    360 unknowns = 2 outer panels of 256; 720 camera unknowns = 3).  Must be the single-context run, identical on every rank.
    sparse: 300 cameras with X kept sparse (ESL_CF_SPARSE=1, esl_cf.hpp): the per-segment products on every rank, the blocks of T
    gathered and the separators' rows applied for a rank's own panels only."""
    if sparse:
        monkeypatch.setenv("ESL_CF_SPARSE", "1")
    g, c, o, _ = pkg.synth.make_graph(300 if sparse else 121, 40, 2400, seed=37, slam=True)
    p = pkg.default_lm_params(jacobian_mode=1, linear_solver=solver)
    ctx.upload_graph(g); ctx.upload_states(c, o)
    ref = ctx.optimize_resident(p)
    used_ref = ctx.lm_solver_used()
    rc, ro = ctx.download_states()
    assert ref["chi2_final"] < 0.5 * ref["chi2_initial"] and used_ref == (1 if solver == 1 else 2)
    monkeypatch.setenv("ESL_CHOL_DIST", "1")
    reps, states, used, ar = run_replicated(pkg, g, c, o, p, n_ranks)
    assert used == [used_ref] * n_ranks
    assert (ctx.lm_solver_stats()["x_form"] > 0) == sparse
    n_tr = sum(ref["trace_trials"])
    # per trial: one message per outer panel (its rows from the diagonal down + the diagonal blocks' inverses) and the 8-byte pivot flag
    n_outer = 2 if used_ref == 2 else 3
    # (+ 1: the once-per-graph check that every rank holds the same graph, esl_comm.hip comm_check_replicated)
    assert ar.calls[0] == n_tr * (n_outer + 1) + 1, (ar.calls[0], n_tr)
    for r in range(n_ranks):
        assert reps[r]["trace_trials"] == ref["trace_trials"] and reps[r]["stop_reason"] == ref["stop_reason"]
        np.testing.assert_allclose(reps[r]["trace_chi2"], ref["trace_chi2"], rtol=1e-9)
        assert reps[r]["trace_chi2"] == reps[0]["trace_chi2"] and reps[r]["trace_lambda"] == reps[0]["trace_lambda"]   # bit-identical ranks
        np.testing.assert_array_equal(states[r][0], states[0][0])
        np.testing.assert_array_equal(states[r][1], states[0][1])
    print("replicated graph, %d ranks, solver %d: chi2 rel %.2e cams %.2e objs %.2e vs one context" % (
        n_ranks, used_ref, float(np.abs(np.array(reps[0]["trace_chi2"]) / np.array(ref["trace_chi2"]) - 1).max()),
        float(np.abs(states[0][0] - rc).max()), float(np.abs(states[0][1] - ro).max())))
    np.testing.assert_allclose(states[0][0], rc, atol=1e-8)
    np.testing.assert_allclose(states[0][1], ro, atol=1e-8)


@pytest.mark.parametrize("n_ranks", [2, 3])
def test_distributed_factorisation_overlapped_messages_same_bits(pkg, ctx, monkeypatch, n_ranks):
    """The distributed factorisation with its messages on their own stream (the owner of panel o + 1 updates, factors and stages it
    before anything else; two staging buffers; esl_chol.hpp) against the same factorisation with the messages on the compute stream
    (ESL_CHOL_DIST_OVERLAP=0): every panel still receives the updates of panels 0, 1, 2, .. in that order, so the two must agree BIT
    FOR BIT -- on a system of 1,350 ellipsoid unknowns = 6 outer panels (both staging buffers reused, look-ahead on every rank)."""
    g, c, o, _ = pkg.synth.make_graph(121, 150, 6000, seed=39, slam=True)
    p = pkg.default_lm_params(jacobian_mode=1, linear_solver=2)
    ctx.upload_graph(g); ctx.upload_states(c, o)
    ref = ctx.optimize_resident(p)
    rc, ro = ctx.download_states()
    monkeypatch.setenv("ESL_CHOL_DIST", "1")
    runs = {}
    for ov in ("1", "0"):
        monkeypatch.setenv("ESL_CHOL_DIST_OVERLAP", ov)
        reps, states, used, ar = run_replicated(pkg, g, c, o, p, n_ranks)
        assert used == [2] * n_ranks
        n_outer = (9 * 150 + 255) // 256
        assert ar.calls[0] == sum(ref["trace_trials"]) * (n_outer + 1) + 1
        for r in range(1, n_ranks):
            np.testing.assert_array_equal(states[r][0], states[0][0]); np.testing.assert_array_equal(states[r][1], states[0][1])
        runs[ov] = (reps[0], states[0])
    monkeypatch.delenv("ESL_CHOL_DIST_OVERLAP", raising=False)
    assert runs["1"][0]["trace_chi2"] == runs["0"][0]["trace_chi2"] and runs["1"][0]["trace_lambda"] == runs["0"][0]["trace_lambda"]
    np.testing.assert_array_equal(runs["1"][1][0], runs["0"][1][0])
    np.testing.assert_array_equal(runs["1"][1][1], runs["0"][1][1])
    assert runs["1"][0]["trace_trials"] == ref["trace_trials"]
    np.testing.assert_allclose(runs["1"][1][0], rc, atol=1e-8)
    np.testing.assert_allclose(runs["1"][1][1], ro, atol=1e-8)


def test_replicated_mode_is_a_checked_contract(pkg, ctx):
    """esl_comm_set_replicated (ABI 4): (i) the mode cannot flip under a resident graph -- ESL_ERR_STATE instead of silently summing
    whole graphs / treating shards as complete (this was a comment); (ii) "the same whole graph on every rank" is compared
    across the ranks by the first optimisation: two contexts in replicated mode, one of them holding only a shard, must BOTH refuse."""
    g, c, o, _ = pkg.synth.make_graph(40, 10, 400, seed=5, slam=True)
    ar = ThreadAllreduce(2)
    cx = pkg.Context(0)
    try:
        cx.comm_init_host(2, 0, ar.make(0))
        cx.upload_graph(g); cx.upload_states(c, o)
        with pytest.raises(pkg.EslError, match="resident graph"):
            cx.comm_set_replicated(True)
    finally:
        cx.comm_destroy(); cx.close()
    ctxs = []
    for r in range(2):
        cx = pkg.Context(0)
        cx.comm_init_host(2, r, ar.make(r))
        cx.comm_set_replicated(True)          # the sanctioned order: mode, then graph; rank 1 cheats with a shard
        ctxs.append(cx)
    try:
        idx = np.arange(g.n_objs // 2)
        ctxs[0].upload_graph(g); ctxs[0].upload_states(c, o)
        ctxs[1].upload_graph(g.subset_objects(idx)); ctxs[1].upload_states(c, o[idx])
        errs = [None, None]

        def work(r):
            try:
                ctxs[r].optimize_resident(pkg.default_lm_params(jacobian_mode=1))
            except Exception as e:   # noqa: BLE001
                errs[r] = e
        th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
        [t.start() for t in th]
        [t.join(60) for t in th]
        assert not any(t.is_alive() for t in th), "the contract check hung"
        for e in errs:
            assert isinstance(e, pkg.EslError) and "DIFFERENT graphs" in str(e), errs
    finally:
        for cx in ctxs:
            cx.comm_destroy(); cx.close()
