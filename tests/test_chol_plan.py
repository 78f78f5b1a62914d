"""The persistent dense factorisation (csrc/esl_chol.hpp, round 4) on the CPU: its STATIC TASK LIST is replayed by a sequential
executor that owns the same dependency words as the device code (pdone / sdone / ver) and numpy versions of the three task bodies
and the chain's potrf.  What this pins without a GPU:
  * the list is a schedule: taken strictly in list order by ONE worker (the chain advancing whenever its tile is final) no task
    ever finds a prerequisite missing -- so on the device a workgroup that spins always waits for work a resident workgroup
    already holds (no deadlock for any number of workgroups, any dispatch order);
  * every read-modify-write of a tile happens at its sequence number, every strip is solved once, every word ends at its expected
    value;
  * executing the list IS a Cholesky factorisation: L and the forward-substituted right-hand side row equal numpy's.
The device kernels reuse the bodies of the launch-per-step kernels (k_chol_potrf2 / k_chol_panel / k_chol_update_lds) that the GPU
suite already holds to 1e-15 residuals; the GPU tests run the pair itself (tests/test_gpu_slam.py)."""
import numpy as np
import pytest


def replay(pkg, n, W, filler, seed=0, fuse=True):
    # fuse: the chain itself solves the (<= 128) rows under its diagonal block -- strips 0, 1 of the panel -- and applies the panel's
    # update to the NEXT diagonal block (ESL_CHOL_FUSE, on by default; the caller sets the environment before building the plan)
    pl = pkg.lib.chol_plan(n, W, filler)
    np_, nR, tasks, ns = pl["np"], pl["nR"], pl["tasks"], pl["ns"]
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((n, n))
    A = G @ G.T + n * np.eye(n)
    b = rng.standard_normal(n)
    rows = n + 1
    M = np.zeros((rows, n))
    M[:n] = np.tril(A); M[n] = b
    M0 = M.copy()
    pdone = np.zeros(np_, int); sdone = np.zeros((np_, nR), int); ver = np.zeros((nR, np_), int)
    Linv = [None] * np_
    chain_k = [0]
    # what a tile receives before its panel is factored (csrc/esl_chol.hpp chol_tile_special / _nU / _final): the diagonal tile of an
    # outer panel's first column panel takes the previous outer panel's contribution as W rank-128 updates instead of one rank-(128 W)
    # (round 6: the first SP column panels of every outer panel, ESL_CHOL_SPECIAL, default 2; the diagonal tile's row and the row tile under it.
    #  A special tile takes its W rank-128 updates of the previous outer panel first, then the J % W of its own outer panel.)
    import os
    SP = max(1, min(int(os.environ.get("ESL_CHOL_SPECIAL", "2")), W))
    special = lambda R, J: J >= W and (J % W) < SP and R in (J // 2, J // 2 + 1)
    n_big = lambda R, J: J // W - (1 if special(R, J) else 0)
    final = lambda R, J: n_big(R, J) + (W if special(R, J) else 0) + J % W
    ridx = lambda R, J, a: (W if (a // W == J // W and special(R, J)) else 0) + a % W     # which of the tile's rank-128 updates panel a's is
    seq_of = lambda R, J, a: n_big(R, J) + ridx(R, J, a)
    nb_of = lambda k: min(128, n - 128 * k)

    chain_stage = [0]      # 0: block chain_k is next to be factored; 1 (fused): factored, its rows below and the next block's update pending
    have_next = [False]    # fused: the block to be factored already carries every update (the chain applied the last one)
    quarter_live = lambda R, J, h, g: 256 * R + 128 * h < rows and 128 * J + 64 * g < n and 256 * R + 128 * h + 127 >= 128 * J + 64 * g

    def advance_chain():
        while chain_k[0] < np_:
            k = chain_k[0]; k0, nb = 128 * k, nb_of(k)
            if chain_stage[0] == 0:
                if not (have_next[0] or ver[k // 2, k] >= final(k // 2, k)):
                    return
                if not have_next[0]:
                    assert ver[k // 2, k] == final(k // 2, k)
                D = np.tril(M[k0:k0 + nb, k0:k0 + nb]); D = D + np.tril(D, -1).T
                L = np.linalg.cholesky(D)
                M[k0:k0 + nb, k0:k0 + nb] = np.triu(M[k0:k0 + nb, k0:k0 + nb], 1) + L
                Linv[k] = np.linalg.inv(L)
                pdone[k] = 1
                have_next[0] = False
                if fuse and rows - (k0 + nb) > 0:
                    chain_stage[0] = 1
                else:
                    chain_k[0] += 1
                continue
            # fused stage: rows r0 .. r0 + 128 of panel k, then block k + 1 -= X X^T
            r0 = k0 + nb; r1 = min(r0 + 128, rows); R1 = r0 // 256; nxt = k + 1 < np_
            below = rows - r0; nstr = 2 if below > 64 else 1
            if ver[R1, k] < final(R1, k):
                return
            if nxt and ver[(k + 1) // 2, k + 1] < seq_of((k + 1) // 2, k + 1, k):
                return
            assert ver[R1, k] == final(R1, k)
            if nxt:
                assert nb == 128 and ver[(k + 1) // 2, k + 1] == seq_of((k + 1) // 2, k + 1, k), "the next diagonal tile is not at the chain's sequence number"
            M[r0:r1, k0:k0 + nb] = M[r0:r1, k0:k0 + nb] @ Linv[k].T
            sdone[k, R1] += nstr
            if nxt:
                c1 = r0; i1 = min(c1 + 128, rows); j1 = min(c1 + 128, n)
                X = M[c1:i1, k0:k0 + nb]
                upd = X @ X[:j1 - c1].T
                rr, cc = np.meshgrid(np.arange(c1, i1), np.arange(c1, j1), indexing="ij")
                mask = rr >= cc
                blk = M[c1:i1, c1:j1]
                blk[mask] -= upd[mask]
                J = k + 1; R = J // 2
                mine = sum(1 for hh in (0, 1) for gg in (0, 1) if hh != (J & 1) and quarter_live(R, J, hh, gg))
                if mine == 0:
                    ver[R, J] = seq_of(R, J, k) + 1
            have_next[0] = nxt
            chain_stage[0] = 0
            chain_k[0] += 1

    counts = {0: 0, 1: 0, 2: 0, 3: 0}
    qdone = np.zeros((nR, np_), int)
    for ty, a, bb, c in tasks:
        advance_chain()
        counts[int(ty)] += 1
        if ty == 3:
            # a quarter of the rank-128 update of the NEXT panels' diagonal tile: the quarters of one update may run side by side (all
            # of them see the tile at the update's sequence number), the last one moves the number on
            R, J, h, g, cnt, full = bb, c & 0xFFFF, (c >> 16) & 1, (c >> 17) & 1, (c >> 18) & 7, (c >> 21) & 7
            k = a
            chains = fuse and k == J - 1 and R == J // 2     # the chain applies this update to the diagonal block (the half h == J & 1) itself
            assert (R in (J // 2, (J + 1) // 2) or special(R, J)) and J > k and 1 <= cnt <= full <= 4
            assert (k // W == J // W) or (special(R, J) and k // W == J // W - 1)
            seq = seq_of(R, J, a)
            assert sdone[k, R] == ns[k, R] and sdone[k, J // 2] == ns[k, J // 2], ("quarter before its operands", a, R, J)
            assert ver[R, J] == seq, ("quarter out of sequence", a, R, J, ver[R, J], seq)
            i0 = 256 * R + 128 * h; j0 = 128 * J + 64 * g
            assert i0 < rows and j0 < n and i0 + 127 >= j0
            i1, j1 = min(i0 + 128, rows), min(j0 + 64, n)
            c0, c1 = 128 * k, 128 * k + 128
            upd = M[i0:i1, c0:c1] @ M[j0:j1, c0:c1].T
            rr, cc = np.meshgrid(np.arange(i0, i1), np.arange(j0, j1), indexing="ij")
            mask = rr >= cc
            blk = M[i0:i1, j0:j1]
            blk[mask] -= upd[mask]
            qdone[R, J] += 1
            # the live quarters of the tile cover its whole lower-triangle part, and cnt is their number
            live_q = [(hh, gg) for hh in (0, 1) for gg in (0, 1) if quarter_live(R, J, hh, gg)]
            task_q = [(hh, gg) for hh, gg in live_q if not (chains and hh == (J & 1))]
            assert full == len(live_q) and cnt == len(task_q) and (h, g) in task_q
            if qdone[R, J] == full * ridx(R, J, a) + cnt:
                ver[R, J] = seq + 1
            continue
        if ty == 0:
            # (round 6) c = strips in the task: i .. i + c - 1, all of ONE row tile that is not next to the diagonal block
            k, i, scnt = a, bb, max(int(c), 1); k0, nb = 128 * k, nb_of(k)
            counts[0] += scnt - 1
            R = (k0 + nb + 64 * i) // 256
            assert scnt == 1 or R >= (k0 + nb) // 256 + 2, "strips next to the diagonal block stay single"
            for q in range(i, i + scnt):
                r0 = k0 + nb + 64 * q; r1 = min(r0 + 64, rows)
                assert r0 // 256 == R
                assert r0 < rows and (q >= 2 or not fuse), "strips 0, 1 are the fused chain's"
                assert pdone[k] == 1, ("S before its diagonal block", k, q)
                assert ver[R, k] == final(R, k), ("S on a tile that is not final", k, q, ver[R, k], final(R, k))
                M[r0:r1, k0:k0 + nb] = M[r0:r1, k0:k0 + nb] @ Linv[k].T
                sdone[k, R] += 1
        else:
            # type 2 (round 6): c = J | cnt << 16 -- the tile takes the rank-(W x 128) updates of outer panels [a, a + cnt) in ONE visit
            R, J, cnt = bb, c & 0xFFFF, (c >> 16) if ty == 2 else 1
            assert cnt >= 1
            ks = [a] if ty == 1 else list(range(a * W, min(np_, (a + cnt) * W)))
            seq = seq_of(R, J, a) if ty == 1 else a
            # (the device code polls the LAST panel's words only: S waits for its tile to be final, so the last panel's strips of a
            #  row tile are solved after every earlier panel of the outer panel has solved its own there -- asserted here for all)
            assert ns[ks[-1], R] > 0 and ns[ks[-1], J // 2] > 0
            for k in ks:
                assert sdone[k, R] == ns[k, R], ("update before its row operand", ty, a, R, J, k)
                assert sdone[k, J // 2] == ns[k, J // 2], ("update before its column operand", ty, a, R, J, k)
                if ty == 1:
                    assert k // W == J // W and J > k and R not in (J // 2, (J + 1) // 2)   # (the diagonal tile and the one under an odd panel's block take their rank-128 updates in quarters)
                else:
                    assert J >= min(np_, (a + cnt) * W) and not (special(R, J) and J // W == a + cnt)
                    assert cnt == 1 or a + cnt <= J // W - 1, "only FAR updates are merged: the one in front of the tile's own outer panel is a visit of its own"
            assert ver[R, J] == seq, ("update out of sequence", ty, a, R, J, ver[R, J], seq)
            c0, c1 = 128 * ks[0], min(128 * (ks[-1] + 1), n)
            i0, i1 = 256 * R, min(256 * R + 256, rows); j0, j1 = 128 * J, min(128 * J + 128, n)
            P_i, P_j = M[i0:i1, c0:c1], M[j0:j1, c0:c1]
            upd = P_i @ P_j.T
            rr, cc = np.meshgrid(np.arange(i0, i1), np.arange(j0, j1), indexing="ij")
            mask = rr >= cc                        # the kernel writes the lower triangle only (row n = the right-hand side's row)
            blk = M[i0:i1, j0:j1]
            blk[mask] -= upd[mask]
            ver[R, J] = seq + cnt
    advance_chain()
    assert chain_k[0] == np_, "the chain never got its last blocks"
    # every word at its final value
    assert (sdone == ns).all()
    for J in range(np_):
        for R in range(nR):
            live = 256 * R + 255 >= 128 * J and 256 * R < rows
            assert ver[R, J] == (final(R, J) if live else 0), (R, J)
    Lref = np.linalg.cholesky(A)
    np.testing.assert_allclose(np.tril(M[:n]), Lref, rtol=0, atol=1e-9 * np.abs(Lref).max())
    np.testing.assert_allclose(M[n], np.linalg.solve(Lref, b), rtol=0, atol=1e-9 * np.abs(b).max())   # y = L^-1 b rides along as row n
    return pl, counts, M0


@pytest.mark.parametrize("n,W,filler", [(1, 2, 128), (7, 2, 128), (128, 2, 128), (130, 2, 4), (450, 2, 128), (777, 2, 3), (1000, 4, 2), (1153, 4, 128),
                                         (2994, 2, 128), (2500, 4, 16)])
def test_task_list_is_a_schedule_and_a_cholesky(pkg, n, W, filler):
    pl, counts, _ = replay(pkg, n, W, filler)
    np_ = pl["np"]
    assert counts[0] == int(pl["ns"].sum()) - sum(min(2, -(-(n + 1 - (128 * k + min(128, n - 128 * k))) // 64)) for k in range(np_))   # (the fused chain's strips are not tasks)
    print("n %d W %d: %d panels, %d tasks (%d strips, %d rank-128 tiles + %d quarters of diagonal tiles, %d rank-%d tiles)"
          % (n, W, np_, len(pl["tasks"]), counts[0], counts[1], counts[3], counts[2], 128 * W))


@pytest.mark.parametrize("n,W,filler", [(130, 2, 4), (777, 2, 3), (1153, 4, 128), (2500, 4, 16)])
def test_task_list_without_the_fused_chain(pkg, monkeypatch, n, W, filler):
    """ESL_CHOL_FUSE=0: every strip and every quarter is a worker task (the first form of the round, kept for A/B)"""
    monkeypatch.setenv("ESL_CHOL_FUSE", "0")
    pl, counts, _ = replay(pkg, n, W, filler, fuse=False)
    assert counts[0] == int(pl["ns"].sum())


@pytest.mark.parametrize("n,W,filler,merge", [(3000, 4, 128, 2), (2500, 4, 16, 2), (2994, 2, 128, 3), (2500, 2, 7, 4), (4000, 4, 128, 4)])
def test_task_list_with_merged_far_updates(pkg, monkeypatch, n, W, filler, merge):
    """ESL_CHOL_MERGE (round 6): a tile's far rank-(W x 128) updates taken `merge` outer panels per visit -- still a schedule, still a
    Cholesky factorisation, fewer type-2 tasks for the same flops"""
    monkeypatch.setenv("ESL_CHOL_MERGE", "1")
    base = pkg.lib.chol_plan(n, W, filler)
    monkeypatch.setenv("ESL_CHOL_MERGE", str(merge))
    pl, counts, _ = replay(pkg, n, W, filler)
    t, t0 = pl["tasks"], base["tasks"]
    k2 = lambda tt: int((tt[tt[:, 0] == 2][:, 3] >> 16).sum())
    assert k2(t) == k2(t0) == int((t0[:, 0] == 2).sum())          # the same outer-panel contributions ...
    assert int((t[:, 0] == 2).sum()) < int((t0[:, 0] == 2).sum())  # ... in fewer visits
    assert int((t[:, 0] == 2).sum() + 0) == counts[2] and (t[t[:, 0] == 2][:, 3] >> 16).max() == merge


@pytest.mark.parametrize("n,W,filler", [(2500, 4, 16), (4000, 4, 128)])
def test_task_list_with_row_tile_strip_tasks(pkg, monkeypatch, n, W, filler):
    """ESL_CHOL_STRIPS=1 (round 6; measured slower, not the default): the strips of a row tile away from the diagonal block as one task"""
    monkeypatch.setenv("ESL_CHOL_STRIPS", "1")
    pl, counts, _ = replay(pkg, n, W, filler)
    t = pl["tasks"]
    assert (t[t[:, 0] == 0][:, 3] > 1).any()


def test_look_ahead_order_of_the_list(pkg):
    """the next outer panel's chain-dependent tasks must not queue behind ALL far updates of the previous one (that would be the
    launch-per-step order without any look-ahead): the first strip of panel W comes before the last far update of outer panel 0"""
    pl = pkg.lib.chol_plan(18000, 4, 128)
    t = pl["tasks"]
    first_s = int(np.nonzero((t[:, 0] == 0) & (t[:, 1] == 4))[0][0])
    far0 = np.nonzero((t[:, 0] == 2) & (t[:, 1] == 0))[0]
    look0 = far0[(t[far0, 3] & 0xFFFF) < 8]
    assert look0.max() < first_s < far0.max()
    assert len(t) > 25000 and pl["np"] == 141 and pl["nR"] == 71   # (round 6: far updates three outer panels per visit, far strips per row tile: 87,651 -> ~37,000 tasks)
