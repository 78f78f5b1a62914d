"""world_size-2 gloo test of the SLAM-mode shard protocol (SURVEY.md section 8 e): the sequence of collectives
csrc/esl_slam.hip and esl_chol.hpp (CholDist) issue over RCCL -- all-reduce of the camera blocks Hcc / b_c per linearisation,
per-panel reduce of the partial reduced camera system to the panel's owner, owner factors + broadcast, every rank updating its own
later panels, replicated back-substitution, all-gather of the LM scalars -- restated in object-oriented-slam_amd/parallel.py
(ShardedSlamLM) over torch.distributed with numpy arithmetic and the CPU checker's linearisation as the engine.

Asserted: the sharded run IS the unsharded LM run of the whole graph (faithful dense checker: same trial counts, chi2 trace,
cameras, ellipsoids); x_c is bit-identical on both ranks in every trial; both ranks issue the same collective sequence; odometry,
lambda on the camera blocks and the camera part of the LM scale enter once; the distributed factorisation equals the replicated
one (all-reduce of S, every rank factors everything)."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleSlamShardEngine:
    """one shard's linearisation / residuals / retraction from the checker's primitives"""

    def __init__(self, po, pkg, graph, cams, objs):
        self.po, self.pkg, self.g = po, pkg, graph
        self.cams = np.asarray(cams, float).reshape(-1, 7).copy()
        self.objs = np.asarray(objs, float).reshape(-1, 10).copy()
        self.free = [i for i in range(graph.n_cams) if not graph.cam_fixed[i]]
        self.n_free_cams, self.n_objs = len(self.free), graph.n_objs
        self.delta = 1e-6

    def linearize(self):
        H, b, fidx, chi = self.po.build_system(self.g, self.cams, self.objs, delta=self.delta, drop_nan=0)
        # g2o's order: active free cameras by id, then ellipsoids.  A camera none of THIS shard's edges touches is not in the shard's
        # system (sparse_optimizer.cpp:236-257): expand to the global free-camera numbering every rank shares (zero rows / columns)
        nc, no = 6 * self.n_free_cams, 9 * self.n_objs
        idx = np.full(len(b), -1)
        for k, i in enumerate(self.free):
            if fidx[i] >= 0:
                idx[fidx[i]:fidx[i] + 6] = np.arange(6 * k, 6 * k + 6)
        for k in range(self.n_objs):
            f = fidx[self.g.n_cams + k]
            assert f >= 0
            idx[f:f + 9] = np.arange(nc + 9 * k, nc + 9 * k + 9)
        assert (idx >= 0).all()
        Hf, bf = np.zeros((nc + no, nc + no)), np.zeros(nc + no)
        Hf[np.ix_(idx, idx)] = H
        bf[idx] = b
        return Hf, bf, chi

    def chi2_at(self, cams, objs):
        return self.po.build_system(self.g, cams, objs, delta=self.delta, drop_nan=0)[3]

    def retract(self, xc, xo):
        cams, objs = self.cams.copy(), self.objs.copy()
        for k, i in enumerate(self.free):
            cams[i] = self.po.cam_oplus(self.cams[i], xc[6 * k:6 * k + 6])
        for k in range(self.n_objs):
            objs[k] = self.po.obj_oplus(self.objs[k], xo[9 * k:9 * k + 9])
        return cams, objs

    def commit(self, cams, objs):
        self.cams, self.objs = cams, objs


def _shard(pkg, g, o, world, rank):
    """rank r's sub-graph: its ellipsoids with all their edges; the odometry edges only on rank 0 (csrc: shard_rank == 0)"""
    part = pkg.lib.partition_objects(g, world)
    ids = np.nonzero(part == rank)[0]
    sub = g.subset_objects(ids)
    if rank != 0:
        sub = pkg.Graph(sub.K, sub.n_cams, sub.n_objs, sub.cam_fixed, sub.bbox_cam, sub.bbox_obj, sub.bbox_meas, sub.bbox_weight, sub.e3d_cam,
                        sub.e3d_obj, sub.e3d_meas, sub.e3d_weight, sub.grav_obj, sub.grav_normal, sub.grav_weight)
    return sub, ids


def _worker(rank, world, port, q, dfac):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = importlib.import_module("object-oriented-slam_amd")
        par = importlib.import_module("object-oriented-slam_amd.parallel")
        from oracle import pyoracle as po
        g, c, o, _ = pkg.synth.make_graph(14, 6, 120, seed=23, slam=True)
        sub, ids = _shard(pkg, g, o, world, rank)
        eng = OracleSlamShardEngine(po, pkg, sub, c, o[ids])
        lm = par.ShardedSlamLM(eng, dist, panel=12, distributed_factor=dfac)
        rep = lm.optimize(pkg.default_lm_params(numeric_delta=1e-6, max_iters=4))
        q.put((rank, ids.tolist(), eng.cams.tolist(), eng.objs.tolist(), rep, lm.log, [x.tolist() for x in lm.xc_log]))
    except Exception:   # noqa: BLE001 -- surfaced by the parent; a dead rank must not leave the other one in a collective
        import traceback
        q.put((rank, "error", traceback.format_exc()))
        os._exit(1)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dfac", [True, False])
def test_sharded_slam_world2_matches_unsharded(pkg, po, dfac):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + (7 if dfac else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, dfac)) for r in range(world)]
    for p in procs:
        p.start()
    results = []
    for _ in range(world):
        r = q.get(timeout=180)
        if len(r) == 3 and r[1] == "error":
            for p in procs:
                p.kill()
            pytest.fail("rank %d: %s" % (r[0], r[2]))
        results.append(r)
    results.sort(key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g, c, o, _ = pkg.synth.make_graph(14, 6, 120, seed=23, slam=True)
    params = pkg.default_lm_params(numeric_delta=1e-6, max_iters=4)
    co, oo, ro = po.optimize(g, c, o, params, solver=po.ORACLE_DENSE)      # the faithful dense LDLT of the whole free system
    merged = np.zeros_like(oo)
    for rank, ids, cams, objs, rep, log, xcs in results:
        merged[ids] = np.asarray(objs)
        assert rep["trace_trials"] == ro["trace_trials"]                       # identical LM decisions on every rank
        # (14 cameras, gauge-soft: the summation order of two shards moves the result of this small graph by ~1e-5; the world-1 run of
        #  the same class differs from the dense checker by 1.2e-6 on chi2 and 1.2e-5 on the states)
        np.testing.assert_allclose(rep["trace_chi2"], ro["trace_chi2"], rtol=1e-5)
        np.testing.assert_allclose(np.asarray(cams), co, atol=1e-4)            # cameras are replicated: every rank ends with all of them
    np.testing.assert_allclose(merged, oo, atol=1e-4)
    (_, _, cams0, _, rep0, log0, xc0), (_, _, cams1, _, rep1, log1, xc1) = results
    assert rep0["trace_chi2"] == rep1["trace_chi2"] and rep0["trace_lambda"] == rep1["trace_lambda"]   # bit-identical control
    assert cams0 == cams1
    assert xc0 == xc1 and len(xc0) == rep0["total_trials"]                     # replicated back-substitution: the same bits
    assert log0 == log1                                                          # the same collective sequence on both ranks
    nc = 6 * 13
    n_pan = (nc + 11) // 12
    per_lin, per_trial = 3, (2 * n_pan if dfac else 1) + 1                       # Hcc + b_c + scalars ; reduce + broadcast per panel (or one all-reduce) + scalars
    assert len(log0) == rep0["iterations"] * per_lin + rep0["total_trials"] * per_trial
    if dfac:
        owners = [int(name.split("->")[1]) for name, _ in log0 if name.startswith("reduce")]
        assert owners[:n_pan] == [p % world for p in range(n_pan)]              # panels dealt cyclically
        sizes = [n for name, n in log0 if name.startswith("broadcast")][:n_pan]
        # one packed message per panel: its rows from the diagonal down + the rhs row, then the inverse of its diagonal block
        assert sizes == [(nc + 1 - 12 * p) * min(12, nc - 12 * p) + min(12, nc - 12 * p) ** 2 for p in range(n_pan)]


def _worker_replicated(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = importlib.import_module("object-oriented-slam_amd")
        par = importlib.import_module("object-oriented-slam_amd.parallel")
        from oracle import pyoracle as po
        g, c, o, _ = pkg.synth.make_graph(14, 6, 120, seed=23, slam=True)
        eng = OracleSlamShardEngine(po, pkg, g, c, o)          # the WHOLE graph on every rank
        lm = par.ReplicatedSlamLM(eng, dist, panel=18)
        rep = lm.optimize(pkg.default_lm_params(numeric_delta=1e-6, max_iters=4))
        q.put((rank, eng.cams.tolist(), eng.objs.tolist(), rep, lm.log, [x.tolist() for x in lm.xc_log]))
    except Exception:   # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc()))
        os._exit(1)
    dist.barrier()
    dist.destroy_process_group()


def test_replicated_slam_world2_divides_the_dense_solve(pkg, po):
    """The N > 1 default of SLAM mode (esl_comm_set_replicated) restated over gloo: every rank holds the whole graph,
    the cameras are eliminated first, the ranks divide the reduced ELLIPSOID system by outer panels (own panels assembled, owner
    factors + one packed broadcast per panel, own later panels updated, replicated back-substitution), the pivot flag is
    all-reduced.  Asserted: the run IS the dense checker's LM run; both ranks end with the same bits; the collective sequence is
    one broadcast per panel with the expected owners and message sizes + one all-reduce of one double per trial, nothing else."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_replicated, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = []
    for _ in range(world):
        r = q.get(timeout=180)
        if len(r) == 3 and r[1] == "error":
            for p in procs:
                p.kill()
            pytest.fail("rank %d: %s" % (r[0], r[2]))
        results.append(r)
    results.sort(key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g, c, o, _ = pkg.synth.make_graph(14, 6, 120, seed=23, slam=True)
    params = pkg.default_lm_params(numeric_delta=1e-6, max_iters=4)
    co, oo, ro = po.optimize(g, c, o, params, solver=po.ORACLE_DENSE)
    (_, cams0, objs0, rep0, log0, xc0), (_, cams1, objs1, rep1, log1, xc1) = results
    assert rep0["trace_trials"] == ro["trace_trials"]
    np.testing.assert_allclose(rep0["trace_chi2"], ro["trace_chi2"], rtol=1e-5)
    np.testing.assert_allclose(np.asarray(cams0), co, atol=1e-4)
    np.testing.assert_allclose(np.asarray(objs0), oo, atol=1e-4)
    assert rep0 == rep1 and cams0 == cams1 and objs0 == objs1 and xc0 == xc1     # the same bits on both ranks
    assert log0 == log1
    no = 9 * 6
    n_pan = (no + 17) // 18
    assert len(log0) == rep0["total_trials"] * (n_pan + 1)                       # nothing per linearisation, no scalar gather
    per = log0[:n_pan + 1]
    assert [name for name, _ in per] == ["broadcast<-%d" % (p % world) for p in range(n_pan)] + ["all_reduce"]
    assert [n for _, n in per] == [(no + 1 - 18 * p) * min(18, no - 18 * p) + min(18, no - 18 * p) ** 2 for p in range(n_pan)] + [1]

