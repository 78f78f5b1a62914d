"""world_size-2 gloo test of the sharded LM control flow (object-oriented-slam_amd/parallel.py).

The production engine is the HIP context; here a CPU checker built from the oracle's primitives
stands in for it so that the partition + scalar-exchange protocol is exercised on a GPU-less box:
the sharded run must reproduce the unsharded LM of the whole graph (same trial counts, same chi2
trace, same final ellipsoids)."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleShardEngine:
    """lm_begin / lm_linearize / lm_try_step / lm_commit on one shard, from oracle primitives."""

    def __init__(self, po, pkg, graph, cams, objs):
        self.po, self.pkg, self.g = po, pkg, graph
        self.cams = np.asarray(cams, float).reshape(-1, 7)
        self.objs = np.asarray(objs, float).reshape(-1, 10).copy()
        self.trial = None

    def lm_begin(self, p):
        self.delta = p.numeric_delta
        return len(self.g.bbox_cam), 0

    def _system(self, objs):
        H, b, fidx, chi = self.po.build_system(self.g, self.cams, objs, delta=self.delta, drop_nan=0)
        return H, b, fidx, chi

    def lm_linearize(self):
        self.H, self.b, self.fidx, chi = self._system(self.objs)
        out = self.pkg.abi.EslLmPartials()
        out.chi2 = chi
        out.max_diag = float(np.abs(np.diag(self.H)).max()) if len(self.b) else 0.0
        return out

    def lm_try_step(self, lam):
        n = len(self.b)
        x = np.zeros(n); ok = 1
        new = self.objs.copy()
        for k in range(self.g.n_objs):
            i = self.fidx[self.g.n_cams + k]
            if i < 0:
                continue
            A = self.H[i:i + 9, i:i + 9] + lam * np.eye(9)
            good, xk = self.po.ldlt_solve(A, self.b[i:i + 9])
            ok = ok and good
            x[i:i + 9] = xk
            new[k] = self.po.obj_oplus(self.objs[k], xk)
        self.trial = new
        out = self.pkg.abi.EslLmPartials()
        out.chi2 = self._system(new)[3]
        out.scale = float(x @ (lam * x + self.b))
        out.solve_ok = 1 if ok else 0
        return out

    def lm_commit(self, accept):
        if accept:
            self.objs = self.trial
        self.trial = None


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("object-oriented-slam_amd")
    par = importlib.import_module("object-oriented-slam_amd.parallel")
    from oracle import pyoracle as po
    g, c, o, _ = pkg.synth.make_graph(40, 8, 300, seed=21)
    part = pkg.lib.partition_objects(g, world)
    sub, osub, ids = par.shard_graph(g, o, world, part)[rank]
    eng = OracleShardEngine(po, pkg, sub, c, osub)
    lm = par.ShardedLM(eng, dist, torch.device("cpu"))
    rep = lm.optimize(pkg.default_lm_params(numeric_delta=1e-6))
    q.put((rank, ids.tolist(), eng.objs.tolist(), rep, lm.n_collectives))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_lm_world2_matches_unsharded(pkg, po):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g, c, o, _ = pkg.synth.make_graph(40, 8, 300, seed=21)
    params = pkg.default_lm_params(numeric_delta=1e-6)
    _, oo, ro = po.optimize(g, c, o, params, solver=1)
    merged = np.zeros_like(oo)
    for rank, ids, objs, rep, ncoll in results:
        merged[ids] = np.asarray(objs)
        assert rep["trace_trials"] == ro["trace_trials"]          # identical LM decisions on every rank
        np.testing.assert_allclose(rep["trace_chi2"], ro["trace_chi2"], rtol=1e-9)
        assert ncoll == rep["iterations"] + rep["total_trials"]   # one collective per linearisation / trial
    np.testing.assert_allclose(merged, oo, atol=1e-8)


def test_single_rank_path_equals_oracle(pkg, po):
    par = importlib.import_module("object-oriented-slam_amd.parallel")
    g, c, o, _ = pkg.synth.make_graph(30, 5, 200, seed=22)
    eng = OracleShardEngine(po, pkg, g, c, o)
    rep = par.ShardedLM(eng).optimize(pkg.default_lm_params(numeric_delta=1e-6))
    _, oo, ro = po.optimize(g, c, o, pkg.default_lm_params(numeric_delta=1e-6), solver=1)
    assert rep["trace_trials"] == ro["trace_trials"]
    np.testing.assert_allclose(eng.objs, oo, atol=1e-9)
