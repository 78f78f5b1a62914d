"""Sharded Levenberg-Marquardt driver: one process per GPU, ellipsoids (with all their edges)
partitioned across ranks, camera states replicated (SURVEY.md §8 e).

In mapping mode (cameras fixed, the shipped reference setting) the per-ellipsoid 9x9 systems are
independent; the only coupling is g2o's *global* LM control (one lambda, one accept/reject for the
whole graph: Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-164).  Each rank
therefore runs the step API of the C-ABI on its shard and the ranks exchange four scalars per
linearisation / per trial with ONE all_gather (RCCL over xGMI on GPUs, gloo in the CPU tests); the
reduction over ranks is done in rank order on every rank, so all ranks take bit-identical decisions.

`engine` is anything with lm_begin / lm_linearize / lm_try_step / lm_commit (the HIP Context in
production; the tests plug in a CPU checker to exercise this control flow without a GPU).
"""
import math

import torch

from . import abi

_DBL_MAX = 1.7976931348623157e308


class ShardedLM:
    def __init__(self, engine, dist=None, device=None, force_collectives=False):
        self.engine = engine
        use = dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or force_collectives)
        self.dist = dist if use else None
        self.device = device if device is not None else torch.device("cpu")
        self.world = self.dist.get_world_size() if self.dist else 1
        self.n_collectives = 0

    def _gather(self, vals):
        """all ranks' partials, reduced in rank order: (sum chi2, max max_diag, sum scale, min ok)"""
        if not self.dist:
            return vals
        t = torch.tensor(vals, dtype=torch.float64, device=self.device)
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        self.n_collectives += 1
        rows = torch.stack(out).cpu().tolist()
        chi2 = 0.0; md = 0.0; sc = 0.0; ok = 1.0
        for r in rows:
            chi2 += r[0]; md = max(md, r[1]); sc += r[2]; ok = min(ok, r[3])
        return [chi2, md, sc, ok]

    def optimize(self, params=None):
        p = params if params is not None else abi.default_lm_params()
        nv, nd = self.engine.lm_begin(p)
        rep = dict(iterations=0, total_trials=0, n_bbox_valid=nv, n_bbox_dropped=nd, stop_reason=0,
                   chi2_initial=0.0, chi2_final=0.0, lambda_final=0.0, trace_chi2=[], trace_lambda=[], trace_trials=[])
        lam, ni, nbad = -1.0, 2.0, 0
        cur = 0.0
        it = 0
        ok_outer = True
        while it < p.max_iters and ok_outer:
            lin = self.engine.lm_linearize()
            chi2, md, _, _ = self._gather([lin.chi2, lin.max_diag, 0.0, 1.0])
            cur = chi2
            ini = cur
            if it == 0:
                rep["chi2_initial"] = cur
                lam, ni, nbad = p.tau * md, 2.0, 0   # computeLambdaInit
            q, rho = 0, 0.0
            while True:
                tr = self.engine.lm_try_step(lam)
                tchi, _, scale, ok = self._gather([tr.chi2, 0.0, tr.scale, float(tr.solve_ok)])
                tmp = tchi if ok > 0.5 else _DBL_MAX
                rho = (cur - tmp) / (scale + 1e-3)
                if rho > 0 and math.isfinite(tmp):
                    alpha = min(1. - (2 * rho - 1) ** 3, 2. / 3.)
                    lam *= max(1. / 3., alpha)
                    ni = 2.0
                    cur = tmp
                    self.engine.lm_commit(True)
                else:
                    lam *= ni
                    ni *= 2
                    self.engine.lm_commit(False)
                q += 1
                if not (rho < 0 and q < p.max_trials):
                    break
            rep["total_trials"] += q
            rep["trace_chi2"].append(cur); rep["trace_lambda"].append(lam); rep["trace_trials"].append(q)
            it += 1
            if q == p.max_trials or rho == 0:
                ok_outer = False; rep["stop_reason"] = 1
            else:
                nbad = nbad + 1 if (ini - cur) * 1e3 < ini else 0
                if nbad >= 3:
                    ok_outer = False; rep["stop_reason"] = 2
        rep["iterations"] = it
        rep["chi2_final"] = cur
        rep["lambda_final"] = lam
        return rep


def shard_graph(graph, objs, n_parts, part_of_obj):
    """Split a graph into per-rank sub-graphs by ellipsoid; returns [(sub_graph, objs_subset, obj_ids)]."""
    import numpy as np
    out = []
    for r in range(n_parts):
        ids = np.nonzero(np.asarray(part_of_obj) == r)[0]
        out.append((graph.subset_objects(ids), np.asarray(objs).reshape(-1, 10)[ids], ids))
    return out
