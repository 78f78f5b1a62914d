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


class ShardedSlamLM:
    """SLAM mode (free cameras) over ellipsoid shards: the exchange protocol of csrc/esl_slam.hip + esl_chol.hpp (CholDist),
    restated over torch.distributed with numpy arithmetic so that it runs -- and is tested -- without a GPU (gloo, world_size 2:
    tests/test_sharded_slam.py).  One LM trial on R ranks:

      linearise   every rank: H, b of ITS ellipsoids' edges (odometry edges: rank 0 only)
                  ALL-REDUCE (sum) of the diagonal camera blocks Hcc (nf x 6 x 6) and b_c          [1 collective]
      trial       every rank: partial reduced camera system  S_r = - sum_{o in shard} W_o (Hoo_o + lambda I)^-1 W_o^T
                  (+ on rank 0: the summed Hcc + lambda I and the odometry off-diagonal blocks), b_s likewise
                  the outer panels (`panel` columns) are dealt cyclically: REDUCE of panel p to its owner p mod R [1 per panel]
                  right-looking loop: the owner factors its panel, BROADCAST of ONE packed message -- the panel's rows
                  from its diagonal down + the inverse of its diagonal block                                  [1 per panel]
                  every rank applies the rank-`panel` update to ITS OWN later panels only
                  back-substitution replicated: all ranks hold the whole factor -> bit-identical x_c everywhere
                  x_o local; ALL-GATHER of {chi2, scale, ok}, reduced in rank order -> identical decisions  [1 collective]

    `engine`: linearize() -> (H, b, chi2) dense over [free cameras | the shard's ellipsoids]; chi2_at(cams, objs) -> float;
    retract(xc, xo) -> (cams, objs) trial states; commit(cams, objs); attributes n_free_cams, n_objs."""

    def __init__(self, engine, dist, panel=12, distributed_factor=True):
        self.e, self.dist, self.panel, self.dfac = engine, dist, int(panel), distributed_factor
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.log = []          # (collective, size in doubles) in issue order: the tests compare it across ranks

    # -- collectives (float64 tensors over the process group) ---------------------------------------------------------------
    def _allreduce(self, a):
        import numpy as np
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
        self.dist.all_reduce(t)
        self.log.append(("all_reduce", t.numel()))
        return t.numpy().reshape(a.shape)

    def _reduce(self, a, dst):
        import numpy as np
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
        self.dist.reduce(t, dst=dst)
        self.log.append(("reduce->%d" % dst, t.numel()))
        return t.numpy().reshape(a.shape)      # meaningful on dst only

    def _bcast(self, a, src):
        import numpy as np
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
        self.dist.broadcast(t, src=src)
        self.log.append(("broadcast<-%d" % src, t.numel()))
        return t.numpy().reshape(a.shape)

    def _gather(self, vals):
        t = torch.tensor(vals, dtype=torch.float64)
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        self.log.append(("all_gather", t.numel()))
        return [o.tolist() for o in out]

    # -- one linearisation: summed camera blocks -----------------------------------------------------------------------------
    def _linearize(self):
        import numpy as np
        H, b, chi2 = self.e.linearize()
        nc = 6 * self.e.n_free_cams
        diag = np.stack([H[6 * k:6 * k + 6, 6 * k:6 * k + 6] for k in range(self.e.n_free_cams)]) if nc else np.zeros((0, 6, 6))
        diag_sum = self._allreduce(diag)
        bc_sum = self._allreduce(b[:nc])
        off = H[:nc, :nc].copy()                      # odometry off-diagonal blocks: non-zero on rank 0 only
        for k in range(self.e.n_free_cams):
            off[6 * k:6 * k + 6, 6 * k:6 * k + 6] = 0
        md_obj = float(np.abs(np.diag(H)[nc:]).max()) if H.shape[0] > nc else 0.0
        md_cam = float(max((np.abs(np.diag(d)).max() for d in diag_sum), default=0.0))
        self.sys = dict(H=H, b=b, nc=nc, diag=diag_sum, bc=bc_sum, off=off)
        return chi2, max(md_obj, md_cam if self.rank == 0 else 0.0)

    # -- one trial: partial S, distributed factorisation, replicated back-substitution -----------------------------------------
    def _solve(self, lam):
        import numpy as np
        s = self.sys
        H, b, nc = s["H"], s["b"], s["nc"]
        no = H.shape[0] - nc
        W, D = H[:nc, nc:], H[nc:, nc:] + lam * np.eye(no)
        ok = 1.0
        Dinv = np.zeros_like(D)
        for k in range(no // 9):                     # block diagonal: 9 x 9 per ellipsoid
            blk = D[9 * k:9 * k + 9, 9 * k:9 * k + 9]
            try:
                np.linalg.cholesky(blk)
            except np.linalg.LinAlgError:
                ok = 0.0
            Dinv[9 * k:9 * k + 9, 9 * k:9 * k + 9] = np.linalg.inv(blk)
        Y = W @ Dinv
        S = -(Y @ W.T)
        bs = -(Y @ b[nc:])
        if self.rank == 0:                           # the summed camera blocks, lambda I and the odometry blocks enter ONCE
            S = S + s["off"]
            for k in range(nc // 6):
                S[6 * k:6 * k + 6, 6 * k:6 * k + 6] += s["diag"][k] + lam * np.eye(6)
            bs = bs + s["bc"]
        M = np.vstack([S, bs[None, :]])             # (nc + 1) x nc: the right-hand side rides along as row nc (esl_chol.hpp)
        P = self.panel
        n_pan = (nc + P - 1) // P
        if not self.dfac:
            M = self._allreduce(M)
        else:
            for p in range(n_pan):                   # every outer panel to its owner only
                cols = slice(p * P, min((p + 1) * P, nc))
                red = self._reduce(M[:, cols], dst=p % self.world)
                if p % self.world == self.rank:
                    M[:, cols] = red
        for p in range(n_pan):
            c0, c1 = p * P, min((p + 1) * P, nc)
            owner = p % self.world
            if not self.dfac or owner == self.rank:  # factor the panel: diagonal block, then the rows below (incl. the rhs row)
                try:
                    L11 = np.linalg.cholesky(M[c0:c1, c0:c1])
                except np.linalg.LinAlgError:
                    ok = 0.0
                    L11 = np.eye(c1 - c0)
                M[c0:c1, c0:c1] = L11
                M[c1:, c0:c1] = np.linalg.solve(L11, M[c1:, c0:c1].T).T
            if self.dfac:      # ONE packed message: the panel's rows from its diagonal down (+ the rhs row), then the inverse of its diagonal block
                w = c1 - c0
                Linv = np.linalg.inv(M[c0:c1, c0:c1]) if owner == self.rank else np.zeros((w, w))
                msg = self._bcast(np.concatenate([M[c0:, c0:c1].ravel(), Linv.ravel()]), src=owner)
                M[c0:, c0:c1] = msg[:(nc + 1 - c0) * w].reshape(nc + 1 - c0, w)
                self.last_Linv = msg[(nc + 1 - c0) * w:].reshape(w, w)       # (csrc: the back-substitution's Linv blocks)
            for q in range(p + 1, n_pan):            # trailing update: own later panels only (all of them when replicated)
                if self.dfac and q % self.world != self.rank:
                    continue
                d0, d1 = q * P, min((q + 1) * P, nc)
                M[d0:, d0:d1] -= M[d0:, c0:c1] @ M[d0:d1, c0:c1].T
        L = np.tril(M[:nc, :nc])
        y = M[nc, :nc]                               # L^-1 b_s (the forward substitution rode along)
        xc = np.linalg.solve(L.T, y) if nc else np.zeros(0)
        xo = Dinv @ (b[nc:] - W.T @ xc)
        scale = float(xo @ (lam * xo + b[nc:]))
        if self.rank == 0:
            scale += float(xc @ (lam * xc + s["bc"]))   # b_c is the rank-summed vector: the camera part is counted once
        return xc, xo, scale, ok

    def optimize(self, params=None):
        p = params if params is not None else abi.default_lm_params()
        rep = dict(iterations=0, total_trials=0, stop_reason=0, chi2_initial=0.0, chi2_final=0.0, trace_chi2=[], trace_lambda=[], trace_trials=[])
        lam, ni, nbad, it, ok_outer, cur = -1.0, 2.0, 0, 0, True, 0.0
        self.xc_log = []
        while it < p.max_iters and ok_outer:
            chi2, md = self._linearize()
            rows = self._gather([chi2, md, 0.0, 1.0])
            cur = sum(r[0] for r in rows); mdg = max(r[1] for r in rows)
            ini = cur
            if it == 0:
                rep["chi2_initial"] = cur
                lam, ni, nbad = p.tau * mdg, 2.0, 0
            q, rho = 0, 0.0
            while True:
                xc, xo, scale, ok = self._solve(lam)
                self.xc_log.append(xc.copy())
                cams, objs = self.e.retract(xc, xo)
                rows = self._gather([self.e.chi2_at(cams, objs), 0.0, scale, ok])
                tchi = sum(r[0] for r in rows); sc = sum(r[2] for r in rows); okg = min(r[3] for r in rows)
                tmp = tchi if okg > 0.5 else _DBL_MAX
                rho = (cur - tmp) / (sc + 1e-3)
                if rho > 0 and math.isfinite(tmp):
                    lam *= max(1. / 3., min(1. - (2 * rho - 1) ** 3, 2. / 3.)); ni = 2.0; cur = tmp
                    self.e.commit(cams, objs)
                else:
                    lam *= ni; ni *= 2
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
        rep["iterations"] = it; rep["chi2_final"] = cur; rep["lambda_final"] = lam
        return rep


class ReplicatedSlamLM(ShardedSlamLM):
    """SLAM mode with the REPLICATED graph (esl_comm_set_replicated, csrc/esl_slam.hip): every rank holds the whole graph and
    linearises it (bit-identically, nothing is summed over ranks); the cameras are eliminated first -- A = Hcc + lambda I is block
    tridiagonal along the odometry chain, X = L^-1 [W | b_c] -- and the ranks divide the reduced ELLIPSOID system
    T = D - X^T X by outer panels: a rank forms only the columns of ITS panels, the owner factors a panel and broadcasts one packed
    message (rows from the diagonal down + the inverse of the diagonal block), every rank updates its own later panels, the
    back-substitution is replicated; the pivot flag is all-reduced (one double) so that every rank rejects the same trials.
    Per trial: one broadcast per panel + one all-reduce.  `engine` as for ShardedSlamLM, over the WHOLE graph."""

    def _linearize(self):
        import numpy as np
        H, b, chi2 = self.e.linearize()
        nc = 6 * self.e.n_free_cams
        self.sys = dict(H=H, b=b, nc=nc)
        return chi2, float(np.abs(np.diag(H)).max())

    def _gather(self, vals):       # nothing to gather: every rank holds every term
        return [list(vals)]

    def _solve(self, lam):
        import numpy as np
        s = self.sys
        H, b, nc = s["H"], s["b"], s["nc"]
        no = H.shape[0] - nc
        ok = 1.0
        A = H[:nc, :nc] + lam * np.eye(nc)
        try:
            L = np.linalg.cholesky(A)                         # (csrc: the block-bidiagonal chain + its dissection; same factor)
        except np.linalg.LinAlgError:
            ok, L = 0.0, np.eye(nc)
        X = np.linalg.solve(L, H[:nc, nc:])                  # nc x no
        y = np.linalg.solve(L, b[:nc])
        D = H[nc:, nc:] + lam * np.eye(no)
        P = self.panel
        n_pan = (no + P - 1) // P
        M = np.zeros((no + 1, no))                            # lower triangle of T, the rhs row rides along
        for p in range(n_pan):                                # this rank's panels only
            if p % self.world != self.rank:
                continue
            c0, c1 = p * P, min((p + 1) * P, no)
            M[c0:no, c0:c1] = D[c0:, c0:c1] - X[:, c0:].T @ X[:, c0:c1]
            M[no, c0:c1] = b[nc + c0:nc + c1] - X[:, c0:c1].T @ y
        for p in range(n_pan):
            c0, c1 = p * P, min((p + 1) * P, no)
            w, owner = c1 - c0, p % self.world
            if owner == self.rank:
                try:
                    L11 = np.linalg.cholesky(M[c0:c1, c0:c1])
                except np.linalg.LinAlgError:
                    ok, L11 = 0.0, np.eye(w)
                M[c0:c1, c0:c1] = L11
                M[c1:, c0:c1] = np.linalg.solve(L11, M[c1:, c0:c1].T).T
                Linv = np.linalg.inv(L11)
            else:
                Linv = np.zeros((w, w))
            msg = self._bcast(np.concatenate([M[c0:, c0:c1].ravel(), Linv.ravel()]), src=owner)
            M[c0:, c0:c1] = msg[:(no + 1 - c0) * w].reshape(no + 1 - c0, w)
            for q in range(p + 1, n_pan):                     # own later panels only
                if q % self.world != self.rank:
                    continue
                d0, d1 = q * P, min((q + 1) * P, no)
                M[d0:, d0:d1] -= M[d0:, c0:c1] @ M[d0:d1, c0:c1].T
        ok = 1.0 if float(self._allreduce(np.array([1.0 - ok]))[0]) < 0.5 else 0.0     # a non-positive pivot anywhere rejects the trial everywhere
        Lt = np.tril(M[:no, :no])
        xo = np.linalg.solve(Lt.T, M[no, :no]) if no else np.zeros(0)
        xc = np.linalg.solve(L.T, y - X @ xo) if nc else np.zeros(0)
        scale = float(xo @ (lam * xo + b[nc:])) + float(xc @ (lam * xc + b[:nc]))
        return xc, xo, scale, ok

