#!/usr/bin/env python
"""bench.py — LM iterations/s of the quadric graph optimisation on MI355X (BASELINE.json metric).

One "step" = one Optimizer::GlobalObjectGraphOptimization-equivalent call (up to 10 LM iterations,
reference Optimizer.cpp:291) over one device-resident synthetic graph of the BASELINE.json
configs[3] shape (10k cams / 2k ellipsoids / ~200k bbox edges, SURVEY.md §8 d), restarted from the
same initial estimate every step (device-side state restore, no PCIe in the timed region).
Default = configs[3] AS NAMED: SLAM mode (free cameras, camera 0 fixed, odometry edges: the reference's bSLAM_mode
branch) with the Schur solve; `--mode mapping` times the reference's shipped setting (all cameras fixed), which the
default line carries as the `mapping` record.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--mode slam|mapping] [--config C4|C3] [--solver auto|camera|ellipsoid]

N > 1 is launched by the driver through torch.distributed.run, one rank per GPU (RCCL).  The path
shards by ellipsoid (SURVEY.md §8 e): the ellipsoids of the ONE named graph are partitioned over the ranks
(strong scaling, cameras replicated); the exchange in mapping mode is one 64-byte all-gather of the LM scalars per
trial, in SLAM mode additionally the all-reduce of the camera blocks and of the reduced camera system.
value = LM iterations of that one global optimisation / max-over-ranks wall time.

The default single-GPU line also carries: `slam_reduced_camera` (the same C4 steps with the reduced CAMERA system of the
north star, FP64-MFMA roofline of its Cholesky) when the timed steps ran the camera-first elimination, `mapping` (C4 with
all cameras fixed: HBM roofline of the linearisation, repeat blocks, its own CPU baseline), `slam_c3`, `fit` (per-frame
ellipsoid fit), `ground_plane`, `streaming_c5`, `cpu_baseline` and `host`.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_MFMA_PEAK_TF = 78.6          # SURVEY.md §8 d (vendor sheet): FP64 matrix
FP64_MFMA_MEASURED_TF = 78.2      # scripts/mfma_peak.hip on the GPU box (profiles/r2_fp64_ceilings.txt); round 1's 35.9 was a benchmark artefact


def algorithmic_bytes_linearize(g, slam):
    """SURVEY.md §8 d: per bbox edge 48 B read (+432 B H_co written in SLAM mode), per 3-D edge 96 B
    (+432 B), per camera 56 B read (+216 B written in SLAM mode), per ellipsoid 80 B read + 432 B written."""
    nb, n3 = len(g.bbox_cam), len(g.e3d_cam)
    b = nb * 48 + n3 * 96 + g.n_cams * 56 + g.n_objs * (80 + 432)
    if slam:
        b += (nb + n3) * 432 + g.n_cams * 216
    return b


def valu_issue_floor(g, avg_ms):
    """What actually bounds the linearisation: FP64 VALU issue, not HBM (DESIGN.md 'kernel rooflines').
    floor = sum over waves of their VALU instruction count x 4 cycles (a wave64 op on a 16-lane SIMD) spread over
    1024 SIMDs at 2.4 GHz; instruction counts are the static ones of profiles/r2_isa_counts.json."""
    try:
        isa = json.load(open(os.path.join(ROOT, "profiles", "r2_isa_counts.json")))
        bb, e3 = isa["k_chunk_linearize<1, 0, 0, false>"], isa["k_chunk_linearize<1, 1, 0, false>"]   # <JAC, TYPE, VALIDATE, TANG>
    except Exception:  # noqa: BLE001
        return None
    cb = np.bincount(g.bbox_obj, minlength=g.n_objs) if len(g.bbox_obj) else np.zeros(1, int)
    ce = np.bincount(g.e3d_obj, minlength=g.n_objs) if len(g.e3d_obj) else np.zeros(1, int)
    waves_bb = int(np.ceil(cb / 64).sum())
    waves_e3 = int(np.ceil(np.ceil(ce / 32).sum() / 2))
    v_bb, v_e3 = bb["f64"] + bb["valu_other"], e3["f64"] + e3["valu_other"]
    cycles = (waves_bb * v_bb + waves_e3 * v_e3) * 4.0 / 1024.0
    floor_ms = cycles / 2.4e9 * 1e3
    return {"bbox_waves": waves_bb, "bbox_valu_instr_per_wave": v_bb, "e3d_waves": waves_e3, "e3d_valu_instr_per_wave": v_e3,
            "simds": 1024, "clock_ghz": 2.4, "floor_ms": floor_ms, "frac": floor_ms / avg_ms if avg_ms > 0 else None,
            "floor_ms_at_1p89_ghz": floor_ms * 2.4 / 1.89,
            "note": "kernel duration includes ~4 us of dispatch; static instruction counts (both sides of branches); a pure "
                    "v_fma_f64 stream pulls the shader clock down to 1.89 GHz on this part (profiles/r2_fp64_ceilings.txt)"}


def host_info():
    """CPU model and core count of the box the CPU legs run on."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"cpu_model": model, "nproc": os.cpu_count(), "cores_usable": len(os.sched_getaffinity(0))}


class pinned_to_one_core:
    """taskset -c <core> for the CPU legs: the single-threaded restatement must not migrate while it is timed."""

    def __enter__(self):
        self.old = os.sched_getaffinity(0)
        self.core = min(self.old)
        try:
            os.sched_setaffinity(0, {self.core})
        except OSError:
            self.core = None
        return self

    def __exit__(self, *exc):
        try:
            os.sched_setaffinity(0, self.old)
        except OSError:
            pass


def cpu_baseline(pkg, g, c, o, params, budget_s=25.0):
    """The CPU restatement (oracle/, single thread, pinned to one core) timed on a bounded sample of the same workload."""
    from oracle import pyoracle as po
    with pinned_to_one_core() as pin:
        out = _cpu_baseline_pinned(pkg, po, g, c, o, params)
    out["pinned_to_core"] = pin.core
    return out


def _cpu_baseline_pinned(pkg, po, g, c, o, params):
    # the WHOLE graph (round 4; rounds 1-3 timed the first 400 ellipsoids and scaled by 5): 200k bbox + 40k 3-D edges with numeric
    # Jacobians are ~6-12 s per optimize(10) on one core
    n_s = g.n_objs
    sub = g.subset_objects(np.arange(n_s))
    t0 = time.perf_counter()
    _, _, rep = po.optimize(sub, c, o[:n_s], params, solver=po.ORACLE_BLOCK)
    dt = time.perf_counter() - t0
    tm = po.last_timing()
    its = max(rep["iterations"], 1)
    per_iter_sample = dt / its
    scale = g.n_objs / n_s                       # linear in edges for the block ("improved") solver
    block_it_s = 1.0 / (per_iter_sample * scale)
    # faithful dense LDLT (linear_solver_dense.h:65-113): time one n=9*64 dense solve, extrapolate n^3
    n_d = min(g.n_objs, 64)
    subd = g.subset_objects(np.arange(n_d))
    pd = pkg.default_lm_params(max_iters=1, numeric_delta=params.numeric_delta)
    po.optimize(subd, c, o[:n_d], pd, solver=po.ORACLE_DENSE)
    td = po.last_timing()
    dense_solve_full = td["solve_s"] * (g.n_objs / n_d) ** 3
    return {
        "value": block_it_s, "unit": "LM iterations/s", "cores": 1, "kind": "port",
        "sample": (f"oracle/esl_oracle.c (CPU restatement, g2o's numeric Jacobians at delta = 1e-9 -- the GPU side of this "
                   f"line runs analytic ones, see speedup note --, per-ellipsoid LDLT = 'improved over reference' solver), "
                   f"ALL {g.n_objs} ellipsoids with all their edges, {its} LM iterations in {dt:.1f} s: a run, nothing scaled"),
        "split_s": tm,
        "faithful_dense_ldlt": {
            "note": ("reference's LinearSolverDense factorises the whole 9N x 9N system every trial; measured on "
                     f"N={n_d} and extrapolated by (N/{n_d})^3"),
            "solve_s_per_trial_extrapolated": dense_solve_full,
            "iterations_per_s_extrapolated": 1.0 / (per_iter_sample * scale + dense_solve_full),
        },
    }


def fit_bench(pkg, ctx, with_cpu=True):
    """Second half of BASELINE.json's metric: per-frame ellipsoid-fit ms.
    C2 = one box with ~50k in-range depth samples (1280x960, stride 1); C5 frame = 20 boxes on a 640x480 frame (stride 3)."""
    out = {}
    cases = {
        "c2_1box_50k_points": (pkg.synth.make_depth_scene(width=1280, height=960, n_objs=1, seed=11, size=(0.23, 0.28)), dict(stride=1), [28]),
        "c5_20boxes_640x480": (pkg.synth.make_depth_scene(n_objs=20, seed=7, spread=1.6, size=(0.1, 0.3)), dict(stride=3), None),
    }
    for name, (sc, kw, labels) in cases.items():
        P = pkg.lib.default_fit_params(**kw)
        lab = sc["labels"] if labels is None else labels
        boxes = sc["bboxes"][:len(lab)]
        args = (sc["depth"], boxes, lab, sc["Twc"], sc["intr"], sc["ground"], P)
        for _ in range(3):
            res = ctx.fit_frame(*args)
        n = 20
        t0 = time.perf_counter()           # host-call time: the captured hipGraph is replayed (no events inside)
        t_abi = 0.0
        for _ in range(n):
            res = ctx.fit_frame(*args)
            t_abi += ctx.last_call_s
        dt_py = (time.perf_counter() - t0) / n
        dt = t_abi / n                     # the C-ABI call (esl_fit_frame) alone; dt_py adds the ctypes / numpy wrapper
        ctx.profile_enable(True)           # kernel time: direct launches bracketed by HIP events
        for _ in range(n):
            res = ctx.fit_frame(*args)
        prof = ctx.profile_get().get("k5", dict(count=1, total_ms=0.0))
        ctx.profile_enable(False)
        k_ms = prof["total_ms"] / max(prof["count"], 1)
        # SURVEY.md §8 d: 2 B per depth sample scanned + 32 B per occupied 1 cm voxel written and read once
        abytes = 2.0 * float(res[3][:, 0].sum()) + 32.0 * float(res[3][:, 1].sum())
        ach = abytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        entry = {"ms_per_frame_host_call": 1e3 * dt, "ms_per_frame_python_call": 1e3 * dt_py, "ms_per_frame_kernel": k_ms,
                 "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                              "traffic": None, "algorithmic_bytes_per_frame": abytes,
                              "note": "latency-bound pointer chasing (hash inserts, union-find, 1-edge LM): a few hundred KB per frame, "
                                      "nowhere near a bandwidth roof; the lever is launch structure, not bytes (DESIGN.md §4)"},
                 "boxes": len(lab), "ok_boxes": int((res[2] == 0).sum()), "samples": int(res[3][:, 0].sum()),
                 "note": "host call = the C-ABI call esl_fit_frame: staging + H2D of the depth image + kernels + D2H, replayed from a captured hipGraph (PCIe-inclusive); python call = the same through the ctypes / numpy wrapper of this repo (what the loop of this script sees); kernel = HIP events around direct launches"}
        if with_cpu:
            from oracle import pyoracle as po
            Po = po.default_fit_params(**kw)
            with pinned_to_one_core():
                t0 = time.perf_counter()
                m = 5
                for _ in range(m):
                    po.fit_frame(sc["depth"], boxes, lab, sc["Twc"], sc["intr"], sc["ground"], Po)
                entry["cpu_port_ms_per_frame"] = 1e3 * (time.perf_counter() - t0) / m
        out[name] = entry
    return out


SOLVER_NAMES = {1: "reduced camera system (ellipsoids eliminated, dense Cholesky of order 6(F-1))",
                2: "reduced ellipsoid system (cameras eliminated first: block-bidiagonal factor, rank-6(F-1) MFMA update, dense Cholesky of order 9N)"}


def slam_flops(n_c, n_o, solver, stats=None):
    """algorithmic flops of one damped solve (SURVEY.md section 8 d): the dense figure n_c^3/3 + 2 n_c^2 of the reduced camera
    system, and what the camera-first elimination actually executes (stats = esl_lm_solver_stats: with X kept sparse the MFMA
    update only carries the separators' rows and the interior rows go through the per-segment products)"""
    dense = n_c ** 3 / 3.0 + 2.0 * n_c * n_c
    if solver == 2:
        sparse = bool(stats and stats.get("x_form", 0) > 0)
        k = 6.0 * stats["separators"] if sparse else float(n_c)   # rows of X in the dense update
        rank_k = float(n_o) * (n_o + 1) * k                       # lower triangle of X^T X: n_o (n_o + 1) / 2 entries x 2 k
        if sparse and stats.get("dense_update_flops_executed", 0) > 0:
            # T's ellipsoids ordered by first camera: the separators' rows are zero above a staircase and the update skips them tile by
            # tile -- the flops it EXECUTES (esl_lm_solver_stats), not the closed form, are what `achieved` may be priced on
            rank_k = float(stats["dense_update_flops_executed"])
        chol = n_o ** 3 / 3.0 + 2.0 * n_o * n_o
        prod = stats["product_flops"] if sparse else 0.0         # (lower block triangle of every segment's product, live rows only)
        return {"dense_figure": dense, "rank_k_update": rank_k, "rank_k_rows": k, "rank_k_closed_form": float(n_o) * (n_o + 1) * k,
                "segment_products": prod, "cholesky": chol, "actual": rank_k + prod + chol + 2.0 * n_o * n_c}
    return {"dense_figure": dense, "cholesky": dense, "actual": dense}


def _pmc_traffic(sparse, key):
    """HBM bytes per launch of one kernel of the C4 SLAM trial from the newest committed PMC passes (FETCH_SIZE x 2 + WRITE_SIZE,
    scripts/pmc_summary.py): a committed file, not this run."""
    for tag in ("r6", "r5", "r4"):
        try:
            path = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic_c4_slam.json")
            pmc = json.load(open(path))
            if bool(pmc.get("x_sparse")) != sparse:
                continue
            if key == "rank_k_update_launch":
                return pmc[key]["traffic_bytes_per_launch"], os.path.relpath(path, ROOT)
            hit = [v for k, v in pmc["kernels"].items() if key in k]
            if hit:
                return hit[0]["traffic_bytes_per_launch"], os.path.relpath(path, ROOT)
        except Exception:  # noqa: BLE001
            continue
    return None, None


def slam_roofline(prof, n_c, n_o, solver, trials, stats=None, dt=None):
    """MFMA roofline of the DOMINANT kernel of a SLAM-mode run -- the kernel class with the largest time per trial in the timed
    region -- from the HIP-event classes of esl_profile_get (events on the library's own stream around every launch); the other
    MFMA kernel of the trial as `secondary`, and `trial_frac` = the flops a trial actually executes / the trial's wall time / peak."""
    fl = slam_flops(n_c, n_o, solver, stats)
    ch = prof.get("cholesky_solve", dict(count=0, total_ms=0.0))
    bd = prof.get("schur_build", dict(count=0, total_ms=0.0))
    fa = prof.get("dense_factorisation", dict(count=0, total_ms=0.0))
    ceiling = {"value": FP64_MFMA_MEASURED_TF, "unit": "TFLOP/s",
               "note": "register-only v_mfma_f64_16x16x4_f64 stream, accumulators in VGPRs: one MFMA per 64 cycles per SIMD at 2.4 GHz "
                       "(scripts/mfma_peak.hip, profiles/r2_fp64_ceilings.txt)"}
    solve_ms = (ch["total_ms"] + bd["total_ms"]) / max(ch["count"], 1)       # everything between the linearisation and x, per trial
    dense_equiv = fl["dense_figure"] / (solve_ms * 1e-3) / 1e12 if solve_ms > 0 else 0.0
    trial_ms = (1e3 * dt / trials) if (dt and trials) else None
    one_launch = 4096 <= n_o < 30000   # esl_chol.hpp chol_factor_solve: the persistent kernel's size range
    if solver == 2:
        rk = prof.get("rank_k_update", dict(count=0, total_ms=0.0))
        rk_avg = rk["total_ms"] / max(rk["count"], 1)
        rk_ach = fl["rank_k_update"] / (rk_avg * 1e-3) / 1e12 if rk_avg > 0 else 0.0
        fa_avg = fa["total_ms"] / max(fa["count"], 1)
        fac_flops = n_o ** 3 / 3.0
        fa_ach = fac_flops / (fa_avg * 1e-3) / 1e12 if fa_avg > 0 else 0.0
        ch_avg = ch["total_ms"] / max(ch["count"], 1)
        sparse = bool(stats and stats.get("x_form", 0) > 0)
        c4 = n_c == 59994 and n_o == 18000
        tile = None if n_o >= 8192 else "128,64 (split-K)"
        k_rows = int(fl["rank_k_rows"])
        sp = prof.get("sparse_block_products", dict(count=0, total_ms=0.0))
        sp_avg = sp["total_ms"] / max(sp["count"], 1)
        rk_traffic, rk_src = _pmc_traffic(sparse, "rank_k_update_launch") if c4 else (None, None)
        fa_traffic, fa_src = _pmc_traffic(sparse, "k_chol_persist") if (c4 and one_launch) else (None, None)
        rec_rk = {"kernel": "%s, the launch that carries the rank-%d update T -= X^T X of the reduced ellipsoid system (order %d)%s" % (
                      ("k_chol_update_lds<%s>" % tile) if tile else "k_chol_update_v (128 x 128 tiles, two four-wave workgroups per CU)", k_rows, n_o, "; X kept sparse: these are the separators' rows, the interior rows go through the per-segment products"
                      if sparse else ""),
                  "bound": "mfma", "achieved": rk_ach, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": rk_ach / FP64_MFMA_PEAK_TF,
                  "traffic": rk_traffic, "traffic_source": rk_src, "algorithmic_flops_per_launch": fl["rank_k_update"], "avg_launch_ms": rk_avg,
                  "launches": rk["count"], "ms_per_trial": rk["total_ms"] / max(ch["count"], 1),
                  "flops_note": ("executed flops: T's ellipsoids are ordered by first camera, the structurally zero rows of the separators' X above the "
                                 "staircase are skipped tile by tile (closed form without the skip: %.4g)" % fl["rank_k_closed_form"])
                  if fl["rank_k_update"] != fl["rank_k_closed_form"] else "closed form n_o (n_o + 1) K"}
        rec_fa = {"kernel": ("k_chol_persist: the whole dense Cholesky factorisation of the reduced ellipsoid system T (order %d) in ONE launch" % n_o) if one_launch
                  else ("dense Cholesky factorisation of T (order %d): k_chol_potrf2 + k_chol_panel + k_chol_update_v / _lds, a launch per step" % n_o),
                  "bound": "mfma", "achieved": fa_ach, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": fa_ach / FP64_MFMA_PEAK_TF,
                  "traffic": fa_traffic, "traffic_source": fa_src, "algorithmic_flops_per_launch": fac_flops, "avg_launch_ms": fa_avg,
                  "launches": fa["count"], "ms_per_trial": fa["total_ms"] / max(ch["count"], 1),
                  "flops_note": "n^3 / 3 (the factorisation alone; the back-substitutions, 2 n^2, are outside this bracket)"}
        dominant, other = (rec_fa, rec_rk) if rec_fa["ms_per_trial"] >= rec_rk["ms_per_trial"] else (rec_rk, rec_fa)
        out = dict(dominant)
        out.update({"selection": "the kernel class with the largest time per trial in the timed region (HIP events around every launch, "
                                 "esl_profile_enable level 2): dense factorisation %.2f ms, rank-K update %.2f ms, per-segment products + gather %.2f ms" % (
                                     rec_fa["ms_per_trial"], rec_rk["ms_per_trial"], sp["total_ms"] / max(ch["count"], 1)),
                    "secondary": other, "measured_ceiling": ceiling,
                    "linear_solve_ms_per_trial": solve_ms, "cholesky_order_9N_ms_per_trial": ch_avg,
                    "cholesky_order_9N_tflops": (fl["cholesky"] / (ch_avg * 1e-3) / 1e12) if ch_avg > 0 else 0.0,
                    "actual_flops_per_trial": fl["actual"],
                    "trial_ms": trial_ms,
                    "trial_frac": (fl["actual"] / (trial_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TF) if trial_ms else None,
                    "trial_frac_note": "flops one LM trial actually executes (rank-K update + per-segment products + factorisation + substitutions) / "
                                       "wall time per trial of the timed region / FP64-MFMA peak",
                    "dense_figure": {"flops_per_trial": fl["dense_figure"], "equivalent_tflops": dense_equiv, "frac_of_peak": dense_equiv / FP64_MFMA_PEAK_TF,
                                     "note": "SURVEY.md section 8 d: a cheaper exact solve is reported against the dense reduced-camera figure AND its own "
                                             "operation count; equivalent = n_c^3/3 flops / the time of the whole linear solve of a trial (it can exceed "
                                             "the peak: the flops were not executed)"}})
        if sparse:
            out["x_sparse"] = {"form": int(stats["x_form"]), "stride": int(stats["stride"]), "separators": int(stats["separators"]),
                               "segments": int(stats["segments"]), "segment_product_flops": fl["segment_products"],
                               "segment_products_ms_per_trial": sp_avg,
                               "segment_products_tflops": fl["segment_products"] / (sp_avg * 1e-3) / 1e12 if sp_avg > 0 else 0.0,
                               "stored_products_bytes": stats["product_bytes"], "slab_bytes": stats["slab_bytes"],
                               "dense_X_update_flops_avoided": float(n_o) * (n_o + 1) * (n_c - fl["rank_k_rows"])}
        return out
    # reduced camera system: the factorisation of S is the trial
    avg = (fa["total_ms"] / max(fa["count"], 1)) if fa["count"] else ch["total_ms"] / max(ch["count"], 1)
    fac_flops = n_c ** 3 / 3.0 if fa["count"] else fl["cholesky"]
    ach = fac_flops / (avg * 1e-3) / 1e12 if avg > 0 else 0.0
    return {"kernel": "dense Cholesky factorisation of the reduced camera system (order %d): %s" % (
                n_c, "k_chol_persist, one launch" if 4096 <= n_c < 30000 else "k_chol_potrf2 + k_chol_panel + k_chol_update_v, a launch per step with look-ahead"),
            "bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TF, "traffic": None,
            "algorithmic_flops_per_launch": fac_flops, "avg_launch_ms": avg, "launches": fa["count"] or ch["count"], "measured_ceiling": ceiling,
            "linear_solve_ms_per_trial": solve_ms, "trial_ms": trial_ms,
            "trial_frac": (fl["actual"] / (trial_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TF) if trial_ms else None}


def slam_run(pkg, ctx, g, c, o, solver, steps, warmup, jacobian=1, barrier=None):
    """K timed optimize(10) steps in SLAM mode on the resident graph (states restored device-side every step); HIP-event classes on
    (an event pair per kernel class and trial: nothing against the milliseconds of a trial)."""
    params = pkg.default_lm_params(jacobian_mode=jacobian, linear_solver=solver)
    for _ in range(warmup):
        ctx.restore_states(); ctx.optimize_resident(params)
    ctx.profile_enable(2)
    (barrier or ctx.synchronize)()
    t0 = time.perf_counter()
    its = trials = 0
    rep = None
    for _ in range(steps):
        ctx.restore_states()
        rep = ctx.optimize_resident(params)
        its += rep["iterations"]; trials += rep["total_trials"]
    (barrier or ctx.synchronize)()
    dt = time.perf_counter() - t0
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    used = ctx.lm_solver_used()
    n_c, n_o = 6 * int((~g.cam_fixed.astype(bool)).sum()), 9 * g.n_objs
    return {"value": its / dt, "unit": "LM iterations/s", "ms_per_optimize": 1e3 * dt / steps, "steps": steps, "warmup": warmup,
            "lm_iterations_per_step": its / steps, "lm_trials_per_step": trials / steps, "linear_solver": used,
            "linear_solver_name": SOLVER_NAMES.get(used, "?"), "unknowns": {"cameras": n_c, "ellipsoids": n_o},
            "chi2": {"initial": rep["chi2_initial"], "final": rep["chi2_final"], "trace": rep["trace_chi2"], "trials": rep["trace_trials"]}, "kernel_ms": prof,
            "roofline": slam_roofline(prof, n_c, n_o, used, trials, ctx.lm_solver_stats(), dt=dt), "_dt": dt, "_its": its, "_trials": trials}


def slam_workload(name, g):
    n = int((~g.cam_fixed.astype(bool)).sum())
    return (f"{name} SLAM mode (the reference's bSLAM_mode branch, Optimizer.cpp:126-158): {g.n_cams} cams ({n} free), {g.n_objs} ellipsoids, "
            f"{len(g.bbox_cam)} bbox + {len(g.e3d_cam)} 3-D + {len(g.grav_obj)} gravity + {len(g.odom_i)} odometry edges; analytic Jacobians; optimize(10) per step")


def slam_c3_bench(pkg, ctx, with_cpu=True):
    """BASELINE.json configs[2] with free cameras: both eliminations, several full optimize(10) steps each; the CPU restatement
    (block-Schur solver, one pinned core) runs the whole graph and provides the rates the C4 CPU figure is extrapolated from."""
    g, c, o, _ = pkg.synth.make_config("C3", seed=0, slam=True)
    ctx.upload_graph(g); ctx.upload_states(c, o); ctx.snapshot_states()
    out = {"workload": slam_workload("C3", g)}
    for solver, key in ((2, "reduced_ellipsoid"), (1, "reduced_camera")):
        r = slam_run(pkg, ctx, g, c, o, solver, steps=5, warmup=1)
        out[key] = {k: v for k, v in r.items() if not k.startswith("_")}
    out["value"] = max(out["reduced_ellipsoid"]["value"], out["reduced_camera"]["value"])
    cpu = None
    if with_cpu:
        from oracle import pyoracle as po
        n = 6 * int((~g.cam_fixed.astype(bool)).sum())
        with pinned_to_one_core() as pin:
            t0 = time.perf_counter()
            _, _, ro = po.optimize(g, c, o, pkg.default_lm_params(), solver=po.ORACLE_BLOCK)
            dtc = time.perf_counter() - t0
        tm = po.last_timing()
        nt = max(ro["total_trials"], 1)
        cpu = {"n": n, "edges": len(g.bbox_cam) + len(g.e3d_cam), "lin_s_per_it": tm["linearize_s"] / max(ro["iterations"], 1),
               "err_s_per_trial": tm["errors_s"] / nt, "ldlt_flops_per_s": (n ** 3 / 3.0) * nt / max(tm["solve_s"], 1e-9)}
        out["cpu_baseline"] = {
            "value": ro["iterations"] / dtc, "unit": "LM iterations/s", "cores": 1, "kind": "port", "pinned_to_core": pin.core,
            "sample": f"oracle/esl_oracle.c, whole C3 SLAM graph, numeric Jacobians (delta 1e-9), block-Schur solver with a dense pivoted LDLT of "
                      f"the {n} x {n} reduced camera system ('improved over reference'): {ro['iterations']} LM iterations / {ro['total_trials']} trials "
                      f"in {dtc:.1f} s", "split_s": tm}
        out["speedup_vs_cpu_port"] = out["value"] / out["cpu_baseline"]["value"]
        # the GPU's own elimination order ON THE CPU (round 4: oracle/esl_oracle.c solve_camfirst -- cameras first along the odometry
        # chain, dense Y, pivoted LDLT of the reduced ellipsoid system): a RUN, at C3 and at a mid-size graph, not an estimate
        cf = {}
        for tag, (gg, cc, oo), iters in (("c3", (g, c, o), 10), ("mid_2k_cams_300_ellipsoids", pkg.synth.make_graph(2000, 300, 16000, seed=41, slam=True)[:3], 1)):
            with pinned_to_one_core():
                t0 = time.perf_counter()
                _, _, rc_ = po.optimize(gg, cc, oo, pkg.default_lm_params(max_iters=iters), solver=po.ORACLE_CAMFIRST)
                dtc = time.perf_counter() - t0
            tmc, tcf = po.last_timing(), po.last_camfirst_timing()
            n_c, n_o, ntr = 6 * int((~gg.cam_fixed.astype(bool)).sum()), 9 * gg.n_objs, max(rc_["total_trials"], 1)
            cf[tag] = {"value": rc_["iterations"] / dtc, "unit": "LM iterations/s", "cores": 1, "kind": "port", "n_cameras": n_c, "n_ellipsoids": n_o,
                       "edges": len(gg.bbox_cam) + len(gg.e3d_cam), "iterations": rc_["iterations"], "trials": rc_["total_trials"], "seconds": dtc,
                       "split_s": {**tmc, **tcf},
                       "rates": {"syrk_flops_per_s": float(n_o) * n_o * n_c * ntr / max(tcf["syrk_s"], 1e-9),
                                 "ldlt_flops_per_s": (n_o ** 3 / 3.0) * ntr / max(tcf["ldlt_s"], 1e-9),
                                 "forward_s_per_nc_no": tcf["chain_forward_s"] / ntr / (float(n_c) * n_o),
                                 "lin_s_per_it_per_edge": tmc["linearize_s"] / max(rc_["iterations"], 1) / (len(gg.bbox_cam) + len(gg.e3d_cam)),
                                 "err_s_per_trial_per_edge": tmc["errors_s"] / ntr / (len(gg.bbox_cam) + len(gg.e3d_cam))}}
            if tag != "c3":   # the GPU on the same mid-size graph (camera-first, one iteration), for the ratio at a second size
                ctx.upload_graph(gg); ctx.upload_states(cc, oo); ctx.snapshot_states()
                pm = pkg.default_lm_params(jacobian_mode=1, max_iters=iters, linear_solver=2)
                ctx.optimize_resident(pm)
                t0 = time.perf_counter(); n_it = 0
                for _ in range(5):
                    ctx.restore_states(); n_it += ctx.optimize_resident(pm)["iterations"]
                ctx.synchronize()
                cf[tag]["gpu_value"] = n_it / (time.perf_counter() - t0)
                cf[tag]["speedup_gpu_vs_cpu_same_elimination"] = cf[tag]["gpu_value"] / cf[tag]["value"]
        cf["c3"]["speedup_gpu_vs_cpu_same_elimination"] = out["reduced_ellipsoid"]["value"] / cf["c3"]["value"]
        out["cpu_same_elimination"] = cf
        cpu["camera_first"] = cf
    return out, cpu


def cpu_baseline_c4_slam(g, cpu, its, trials, gpu_flops_per_trial=None):
    """C4 SLAM on one CPU core is not runnable (7.2e13 flop per trial): extrapolated from the rates the restatement reached on C3.
    gpu_flops_per_trial: what the GPU's elimination executes per trial -- priced at the same CPU flop rate it gives the baseline a
    CPU would reach WITH that elimination (same_elimination_estimate): the part of the speed-up that is hardware, not algorithm."""
    n = 6 * int((~g.cam_fixed.astype(bool)).sum())
    e4 = len(g.bbox_cam) + len(g.e3d_cam)
    per_trial = (n ** 3 / 3.0) / cpu["ldlt_flops_per_s"] + cpu["err_s_per_trial"] * e4 / cpu["edges"]
    per_it = cpu["lin_s_per_it"] * e4 / cpu["edges"] + per_trial * (trials / max(its, 1))
    same = None
    if cpu.get("camera_first"):
        # the camera-first elimination as the CPU restatement runs it (dense Y): per trial n_o^2 n_c flops of T = D - Y^T Y + n_o^3 / 3 of
        # its LDLT + the forward substitution, each at the rate MEASURED at the larger of the two sizes run above (both are reported)
        cfm = cpu["camera_first"]
        big = cfm["mid_2k_cams_300_ellipsoids"]["rates"]; small = cfm["c3"]["rates"]
        n_o = 9.0 * g.n_objs
        pt = n_o * n_o * n / big["syrk_flops_per_s"] + (n_o ** 3 / 3.0) / big["ldlt_flops_per_s"] + big["forward_s_per_nc_no"] * n * n_o + big["err_s_per_trial_per_edge"] * e4
        # (linearisation per edge from the SMALLER run: the restatement accumulates into a dense H, whose clearing grows with n^2 and
        #  would inflate a per-edge figure taken at the mid size)
        pi = min(big["lin_s_per_it_per_edge"], small["lin_s_per_it_per_edge"]) * e4 + pt * (trials / max(its, 1))
        same = {"value": 1.0 / pi, "unit": "LM iterations/s", "seconds_per_iteration": pi,
                "anchors": {"c3": small, "mid_2k_cams_300_ellipsoids": big},
                "note": "EXTRAPOLATED to C4 from two MEASURED runs of the CPU restatement with the GPU's elimination order (cameras first; "
                        "oracle/esl_oracle.c solve_camfirst, dense Y): n_o^2 n_c flops of the reduced-system build at "
                        f"{big['syrk_flops_per_s'] / 1e9:.2f} GFLOP/s (C3: {small['syrk_flops_per_s'] / 1e9:.2f}), n_o^3/3 of its pivoted LDLT at "
                        f"{big['ldlt_flops_per_s'] / 1e9:.2f} GFLOP/s (C3: {small['ldlt_flops_per_s'] / 1e9:.2f}) -> {pi:.0f} s per LM iteration"}
        if gpu_flops_per_trial:
            same["gpu_flops_per_trial_for_scale"] = gpu_flops_per_trial
    block = {"value": 1.0 / per_it, "unit": "LM iterations/s", "seconds_per_iteration": per_it,
             "note": f"block-Schur restatement (ellipsoids eliminated, dense pivoted LDLT of the {n} x {n} reduced camera system) EXTRAPOLATED, not run "
                     f"({n}^3/3 = {n ** 3 / 3.0:.2e} flop per trial): timed on the whole C3 SLAM graph in this run; linearisation and error evaluation "
                     f"scaled by the edge count, the reduced solve by the LDLT flop rate it reached there ({cpu['ldlt_flops_per_s'] / 1e9:.2f} GFLOP/s)"}
    if same is None:
        return {"value": block["value"], "unit": "LM iterations/s", "cores": 1, "kind": "port", "sample": block["note"], "block_schur_extrapolation": block}
    # the headline CPU figure is the one CLOSER TO A RUN: the restatement with the GPU's own elimination, measured at two
    # sizes in this invocation and scaled by its measured flop rates; the block-Schur extrapolation (one anchor, n^3 scaling over a
    # factor 20 in n) stays as a note
    cfm = cpu["camera_first"]
    return {"value": same["value"], "unit": "LM iterations/s", "cores": 1, "kind": "port", "seconds_per_iteration": same["seconds_per_iteration"],
            "sample": (f"camera-first CPU restatement RUN here, 1 pinned core: C3 SLAM ({cfm['c3']['iterations']} it, {cfm['c3']['seconds']:.1f} s) + 2000 cams/300 "
                       f"ellipsoids ({cfm['mid_2k_cams_300_ellipsoids']['iterations']} it, {cfm['mid_2k_cams_300_ellipsoids']['seconds']:.1f} s); C4 = "
                       f"their measured rates x C4 op counts ({same['seconds_per_iteration']:.0f} s/it): estimate"),
            "sample_detail": ("oracle/esl_oracle.c with the camera-first elimination (ESL_ORACLE_CAMFIRST: block Cholesky along the odometry chain, dense Y, pivoted "
                       "LDLT of the reduced ellipsoid system), numeric Jacobians, one pinned core, RUN in this invocation on C3 SLAM "
                       f"({cfm['c3']['iterations']} iterations in {cfm['c3']['seconds']:.1f} s) and on 2,000 cameras / 300 ellipsoids "
                       f"({cfm['mid_2k_cams_300_ellipsoids']['iterations']} iteration in {cfm['mid_2k_cams_300_ellipsoids']['seconds']:.1f} s); C4 itself "
                       f"({same['seconds_per_iteration']:.0f} s per iteration) does not fit a bench: its figure is those runs' measured rates times C4's "
                       "operation counts -- an estimate anchored on runs, labelled as such"),
            "anchors": same["anchors"], "estimate_note": same["note"], "block_schur_extrapolation": block}


def mapping_bench(pkg, ctx, config="C4", jacobian="analytic", steps=20, warmup=10, blocks=10, extra=True):
    """The shipped setting of the reference (bSLAM_mode = false: all cameras fixed, per-ellipsoid 9x9 systems): device-driven LM,
    HBM roofline of the linearisation kernel from a sampled HIP-event pair (one launch of every fourth run)."""
    g, c, o, _ = pkg.synth.make_config(config, seed=0, slam=False)
    params = pkg.default_lm_params(jacobian_mode=1 if jacobian == "analytic" else 0)
    ctx.upload_graph(g); ctx.upload_states(c, o); ctx.snapshot_states()

    def one_step():
        ctx.restore_states()
        return ctx.optimize_resident(params)
    for _ in range(warmup):
        one_step()
    ctx.profile_enable(0 if os.environ.get("ESL_BENCH_NO_PROFILE") == "1" else 1)
    ctx.synchronize()
    t0 = time.perf_counter()
    iters = trials = 0
    rep = None
    for _ in range(steps):
        rep = one_step()
        iters += rep["iterations"]; trials += rep["total_trials"]
    ctx.synchronize()
    dt = time.perf_counter() - t0
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    blk = []
    for _ in range(blocks):
        ctx.synchronize()
        tb = time.perf_counter()
        ib = 0
        for _ in range(steps):
            ib += one_step()["iterations"]
        ctx.synchronize()
        blk.append(ib / (time.perf_counter() - tb))
    if extra:
        ctx.profile_enable(2)
        for _ in range(3):
            one_step()
        for k, v in ctx.profile_get().items():
            if k != "linearize":
                prof[k + " (extra untimed pass)"] = v
        ctx.profile_enable(False)
    lin = prof.get("linearize", dict(count=0, total_ms=0.0))
    avg_ms = lin["total_ms"] / max(lin["count"], 1)
    abytes = algorithmic_bytes_linearize(g, False)
    achieved = abytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic = None
    try:   # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE, see the file) -- same workload only
        if config == "C4" and jacobian == "analytic":
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r4_pmc_traffic_device_lm.json")))["kernels"]
            traffic = [v for k, v in pmc.items() if "k_chunk_linearize_both<1, 0" in k or "k_chunk_linearize_both<1, false" in k][0]["traffic_bytes_per_launch"]
    except Exception:  # noqa: BLE001
        traffic = None
    roof = {"kernel": "k_chunk_linearize_both (bbox + 3-D chunks, one launch)", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": "profiles/r4_pmc_traffic_device_lm.json (committed PMC passes, not this run)",
            "algorithmic_bytes_per_launch": abytes, "avg_launch_ms": avg_ms, "launches": lin["count"],
            "sampling": "HIP events around ONE linearisation launch (the second trial's) of every FOURTH optimize() of the timed region: an event pair "
                        "splits two back-to-back dispatches and costs that trial ~20 us"}
    if jacobian == "analytic":
        roof["valu_issue_floor"] = valu_issue_floor(g, avg_ms)
    out = {"value": iters / dt, "unit": "LM iterations/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warmup,
           "workload": f"{config} synthetic graph: {g.n_cams} cams, {g.n_objs} ellipsoids, {len(g.bbox_cam)} bbox + {len(g.e3d_cam)} 3-D + {len(g.grav_obj)} gravity "
                       f"edges; mapping mode (all cameras fixed: the reference as shipped, Optimizer.cpp:126); {jacobian} Jacobians; optimize(10) per step",
           "lm_iterations_per_step": iters / steps, "lm_trials_per_step": trials / steps, "kernel_ms": prof,
           "chi2": {"initial": rep["chi2_initial"], "final": rep["chi2_final"], "trace": rep["trace_chi2"], "trials": rep["trace_trials"]}, "roofline": roof}
    if blk:
        out["repeat_blocks"] = {"blocks": len(blk), "steps_per_block": steps, "median": float(np.median(blk)), "min": float(np.min(blk)), "max": float(np.max(blk)),
                                "note": "LM iterations/s of further K-step blocks (HIP events off); `value` is the first, timed block"}
    return out, (g, c, o)


def ground_plane_bench(pkg, ctx, with_cpu=True):
    """The step in front of the fit: PlaneExtractor::extractGroundPlane on one 640 x 480 depth frame."""
    sc = pkg.synth.make_depth_scene(n_objs=3, seed=5)
    for _ in range(2):
        r = ctx.extract_ground_plane(sc["depth"], sc["intr"])
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        r = ctx.extract_ground_plane(sc["depth"], sc["intr"])
    out = {"ms_per_frame_host_call": 1e3 * (time.perf_counter() - t0) / n, "ok": bool(r["ok"]), "planes": int(r["n_planes"]), "pixels_of_the_plane": int(r["n_pixels"]),
           "frame": "640x480 synthetic (ground + 3 ellipsoids)",
           "note": "host call incl. H2D of the depth image; buffers from a grow-only context slab (round 2: 33 MB of hipMalloc per call); round 5: the call now "
                   "includes PCL's refinement pass of segmentAndRefine (PlaneExtractor.cpp:82) -- two raster passes that are sequential over the rows by "
                   "definition, one workgroup, ~3.5 ms of the call; runs once per sequence (Tracking.cpp:498-499)"}
    p0 = pkg.abi.default_plane_params(refine=0)
    ctx.extract_ground_plane(sc["depth"], sc["intr"], p0)
    t0 = time.perf_counter()
    for _ in range(n):
        r0 = ctx.extract_ground_plane(sc["depth"], sc["intr"], p0)
    out["segments_only_refine_0"] = {"ms_per_frame_host_call": 1e3 * (time.perf_counter() - t0) / n, "planes": int(r0["n_planes"]), "pixels_of_the_plane": int(r0["n_pixels"])}
    if with_cpu:
        from oracle import pyoracle as po
        with pinned_to_one_core():
            t0 = time.perf_counter()
            po.extract_ground_plane(sc["depth"], sc["intr"])
            out["cpu_port_ms_per_frame"] = 1e3 * (time.perf_counter() - t0)
    return out


def streaming_bench(pkg, ctx, n_frames=120):
    """BASELINE.json configs[4]: streaming RGB-D at 30 fps, 20 boxes per frame: per frame = single-frame fit of the
    20 boxes + re-optimisation of the whole accumulated graph.  Two ways of keeping the graph: "rebuild" = what the
    reference does every frame (Optimizer.cpp:127,166,250: every vertex and edge again; here: sort + pack + one H2D of the
    whole graph), "append" = esl_graph_append (the frame's edges into the slack of the device-resident arrays).  Reports
    sustained ms/frame of both; `ms_per_frame` is the append path."""
    sc = pkg.synth.make_depth_scene(n_objs=20, seed=7, spread=1.6, size=(0.1, 0.3))
    P = pkg.lib.default_fit_params()
    g, c, o, _ = pkg.synth.make_graph(n_frames, 20, 20 * n_frames, seed=3)
    params = pkg.default_lm_params(jacobian_mode=1)
    meas, e3m = g.bbox_meas.reshape(-1, 4), g.e3d_meas.reshape(-1, 10)
    # per-frame edge sets, prepared outside the timed loops (the tracker would hand them over as they arrive)
    full, delta = [], []
    cnt_prev = np.zeros(g.n_objs, int)
    for f in range(n_frames):
        mb = g.bbox_cam <= f
        cnt = np.bincount(g.bbox_obj[mb], minlength=g.n_objs)
        mbf = mb & (cnt[g.bbox_obj] > 2)                  # 2-D edges only for objects with > 2 observations (Optimizer.cpp:201)
        me = g.e3d_cam <= f
        full.append((mbf, me))
        new = mbf & ((g.bbox_cam == f) | (cnt_prev[g.bbox_obj] <= 2))
        delta.append((new, g.e3d_cam == f))
        cnt_prev = cnt
    out = {"frames": n_frames, "boxes_per_frame": 20}
    ctx.fit_frame(sc["depth"], sc["bboxes"], sc["labels"], sc["Twc"], sc["intr"], sc["ground"], P)  # warm-up
    for mode in ("rebuild", "append"):
        t_fit = t_opt = 0.0
        objs = o.copy()
        its = 0
        t0 = time.perf_counter()
        for f in range(n_frames):
            ta = time.perf_counter()
            ctx.fit_frame(sc["depth"], sc["bboxes"], sc["labels"], sc["Twc"], sc["intr"], sc["ground"], P)
            tb = time.perf_counter()
            if mode == "rebuild" or f == 0:
                mb, me = full[f]
                gf = pkg.Graph(g.K, f + 1, g.n_objs, None, g.bbox_cam[mb], g.bbox_obj[mb], meas[mb], g.bbox_weight[mb],
                               g.e3d_cam[me], g.e3d_obj[me], e3m[me], g.e3d_weight[me], g.grav_obj, g.grav_normal, g.grav_weight)
                if mode == "rebuild":
                    _, objs, rep = ctx.optimize(gf, c[:f + 1], objs, params)
                else:
                    ctx.upload_graph(gf); ctx.upload_states(c[:1], objs)
                    rep = ctx.optimize_resident(params)
            else:
                mb, me = delta[f]
                ctx.append_graph(new_cams=c[f:f + 1], bbox=(g.bbox_cam[mb], g.bbox_obj[mb], meas[mb], g.bbox_weight[mb]),
                                 e3d=(g.e3d_cam[me], g.e3d_obj[me], e3m[me], g.e3d_weight[me]))
                rep = ctx.optimize_resident(params)
            if mode == "append":
                _, objs = ctx.download_states()       # the tracker reads the ellipsoids back every frame
            its += rep["iterations"]
            tc = time.perf_counter()
            t_fit += tb - ta; t_opt += tc - tb
        dt = time.perf_counter() - t0
        out[mode] = {"ms_per_frame": 1e3 * dt / n_frames, "fps": n_frames / dt, "fit_ms_per_frame": 1e3 * t_fit / n_frames,
                     "reoptimize_ms_per_frame": 1e3 * t_opt / n_frames, "lm_iterations_per_frame": its / n_frames}
        if mode == "append":
            out[mode]["relayouts"] = ctx.graph_sizes()["relayouts"]
    # "pipelined": the fit of frame f+1 on a SECOND context (its own HIP stream, driven by a worker thread) while frame f's edges
    # are appended and the graph is re-optimised on the first -- two contexts are how the C-ABI exposes two streams; ctypes
    # releases the GIL for the duration of the calls, so the two host calls and their kernels really overlap.
    import queue
    import threading
    fctx = pkg.Context(ctx.device)
    try:
        fctx.fit_frame(sc["depth"], sc["bboxes"], sc["labels"], sc["Twc"], sc["intr"], sc["ground"], P)   # warm-up (captures its graph)
        todo, done = queue.Queue(), queue.Queue()

        def fit_worker():
            while True:
                f = todo.get()
                if f is None:
                    return
                done.put((f, fctx.fit_frame(sc["depth"], sc["bboxes"], sc["labels"], sc["Twc"], sc["intr"], sc["ground"], P)))
        th = threading.Thread(target=fit_worker, daemon=True)
        th.start()
        objs = o.copy()
        its = 0
        t0 = time.perf_counter()
        todo.put(0)
        for f in range(n_frames):
            done.get()                                   # frame f's ellipsoids (they become its 3-D edges)
            if f + 1 < n_frames:
                todo.put(f + 1)                          # next frame's fit runs under this frame's append + LM
            if f == 0:
                mb, me = full[0]
                gf = pkg.Graph(g.K, 1, g.n_objs, None, g.bbox_cam[mb], g.bbox_obj[mb], meas[mb], g.bbox_weight[mb],
                               g.e3d_cam[me], g.e3d_obj[me], e3m[me], g.e3d_weight[me], g.grav_obj, g.grav_normal, g.grav_weight)
                ctx.upload_graph(gf); ctx.upload_states(c[:1], objs)
            else:
                mb, me = delta[f]
                ctx.append_graph(new_cams=c[f:f + 1], bbox=(g.bbox_cam[mb], g.bbox_obj[mb], meas[mb], g.bbox_weight[mb]),
                                 e3d=(g.e3d_cam[me], g.e3d_obj[me], e3m[me], g.e3d_weight[me]))
            rep = ctx.optimize_resident(params)
            _, objs = ctx.download_states()
            its += rep["iterations"]
        dt = time.perf_counter() - t0
        todo.put(None)
        th.join(10)
        out["pipelined"] = {"ms_per_frame": 1e3 * dt / n_frames, "fps": n_frames / dt, "lm_iterations_per_frame": its / n_frames,
                            "note": "fit of frame f+1 on a second context / stream under frame f's append + re-optimisation"}
    finally:
        fctx.close()
    # the same stream in SLAM MODE (the reference's bSLAM_mode branch: camera 0 fixed, every new camera free, one odometry edge per frame):
    # esl_graph_append with free cameras (ABI 4) -- records and states in place, the camera-indexed tables rebuilt from the host mirror
    try:
        gs, cs, os_, _ = pkg.synth.make_graph(n_frames, 20, 20 * n_frames, seed=3, slam=True)
        ms, e3s, oms = gs.bbox_meas.reshape(-1, 4), gs.e3d_meas.reshape(-1, 10), gs.odom_meas.reshape(-1, 7)
        f0 = 2
        mb, me, mo = gs.bbox_cam <= f0, gs.e3d_cam <= f0, gs.odom_j <= f0
        g0 = pkg.Graph(gs.K, f0 + 1, gs.n_objs, gs.cam_fixed[:f0 + 1], gs.bbox_cam[mb], gs.bbox_obj[mb], ms[mb], gs.bbox_weight[mb], gs.e3d_cam[me], gs.e3d_obj[me], e3s[me],
                       gs.e3d_weight[me], gs.grav_obj, gs.grav_normal, gs.grav_weight, gs.odom_i[mo], gs.odom_j[mo], oms[mo])
        ctx.upload_graph(g0); ctx.upload_states(cs[:f0 + 1], os_)
        ctx.optimize_resident(params)
        its = 0
        t0 = time.perf_counter()
        for f in range(f0 + 1, n_frames):
            ctx.fit_frame(sc["depth"], sc["bboxes"], sc["labels"], sc["Twc"], sc["intr"], sc["ground"], P)
            mb, me, mo = gs.bbox_cam == f, gs.e3d_cam == f, gs.odom_j == f
            ctx.append_graph(new_cams=cs[f:f + 1], new_cam_fixed=[0], bbox=(gs.bbox_cam[mb], gs.bbox_obj[mb], ms[mb], gs.bbox_weight[mb]),
                             e3d=(gs.e3d_cam[me], gs.e3d_obj[me], e3s[me], gs.e3d_weight[me]), odom=(gs.odom_i[mo], gs.odom_j[mo], oms[mo]))
            its += ctx.optimize_resident(params)["iterations"]
            ctx.download_states()
        dt = time.perf_counter() - t0
        nfr = n_frames - f0 - 1
        out["append_slam_mode"] = {"ms_per_frame": 1e3 * dt / nfr, "fps": nfr / dt, "lm_iterations_per_frame": its / nfr, "frames": nfr,
                                   "final_free_cameras": n_frames - 1, "relayouts": ctx.graph_sizes()["relayouts"],
                                   "note": "fit of 20 boxes + esl_graph_append of one FREE camera with its odometry edge and its ~20 + 4 edges + re-optimisation of "
                                           "the whole graph (cameras and ellipsoids, host-driven LM with the Schur solve) per frame; the graph grows to "
                                           f"{n_frames - 1} free cameras"}
    except Exception as e:  # noqa: BLE001
        out["append_slam_mode"] = {"error": str(e)}
    out["ms_per_frame"] = out["append"]["ms_per_frame"]
    out["fps"] = out["append"]["fps"]
    out["final_graph_edges"] = int(full[-1][0].sum() + full[-1][1].sum())
    out["note"] = ("host-call times incl. PCIe (depth upload every frame; rebuild: the whole graph every frame, append: the frame's edges); "
                   "the fit replays a captured hipGraph, the LM loop is device-driven (no host in the loop, nothing to capture)")
    return out


DRIVER_LINE_MAX = 4096
_LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
_ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic_flops_per_launch",
              "algorithmic_bytes_per_launch", "avg_launch_ms", "launches", "trial_frac")
_CPU_KEYS = ("value", "unit", "cores", "kind", "sample")


def _clip(v, n):
    return v if not isinstance(v, str) or len(v) <= n else v[:n - 3] + "..."


def driver_line(out, extras_path="bench_extras.json"):
    """The ONE stdout line the driver parses (the 24 KB record was dropped): the contract's keys, `config`, `roofline` and
    `cpu_baseline` with their prescribed fields, strings clipped -- always < DRIVER_LINE_MAX bytes.  Everything else (secondary
    kernels, the other configs, chi2 traces, anchors) is the side file `--extras` and stderr."""
    line = {k: out[k] for k in _LINE_KEYS if k in out}
    line["config"] = {k: _clip(v, 260) for k, v in out.get("config", {}).items() if not isinstance(v, (dict, list))}
    if out.get("roofline"):
        line["roofline"] = {k: _clip(out["roofline"][k], 200) for k in _ROOF_KEYS if k in out["roofline"]}
    if out.get("cpu_baseline"):
        line["cpu_baseline"] = {k: _clip(out["cpu_baseline"][k], 200) for k in _CPU_KEYS if k in out["cpu_baseline"]}
    line["extras"] = extras_path
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 3 in SLAM mode, 20 in mapping mode)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default: 1 in SLAM mode, 10 in mapping mode)")
    ap.add_argument("--mode", default="slam", choices=["mapping", "slam"],
                    help="slam = BASELINE.json configs[3] as named (free cameras, Schur solve); mapping = the reference as shipped (cameras fixed)")
    ap.add_argument("--config", default="C4")
    ap.add_argument("--jacobian", default="analytic", choices=["analytic", "numeric"])
    ap.add_argument("--solver", default="auto", choices=["auto", "camera", "ellipsoid"], help="esl_linear_solver of the SLAM-mode steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the timed region (no mapping / fit / streaming / C3 / reduced-camera records)")
    ap.add_argument("--extras", default=os.path.join(ROOT, "bench_extras.json"),
                    help="side file for everything that is not the driver's line (the full record: secondary kernels, other configs, anchors)")
    a = ap.parse_args()
    slam = a.mode == "slam"
    if a.steps is None:
        a.steps = 3 if slam else 20
    if a.warmup is None:
        a.warmup = 1 if slam else 10

    # stdout carries exactly ONE line (the JSON).  Libraries (RCCL prints a banner at exit) write to fd 1 too,
    # so keep a private copy of the real stdout and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    force_dist = os.environ.get("ESL_BENCH_FORCE_DIST") == "1"   # exercise the RCCL exchange on a single GPU
    # ESL_BENCH_HOST_TRANSPORT=1: the library's collectives run over esl_comm_init_host with a gloo all-reduce of torch.distributed
    # behind the callback instead of RCCL, and the ranks share the visible devices round-robin -- so the N-rank branch below (real
    # process boundaries, the replicated-graph communicator, the panel messages on their own stream, the ellipsoid-sharded form)
    # executes on a box with ONE GPU (tests/test_bench_multirank.py).  A correctness vehicle, not a timing: never a bench number.
    host_transport = os.environ.get("ESL_BENCH_HOST_TRANSPORT") == "1"
    if host_transport:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("gloo" if host_transport else "nccl", rank=rank, world_size=world)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    pkg = importlib.import_module("object-oriented-slam_amd")
    solver = {"auto": 0, "camera": 1, "ellipsoid": 2}[a.solver]
    # ONE graph of the named shape; N > 1: its ellipsoids (with all their edges) partitioned over the ranks, cameras (and
    # odometry) replicated -> strong scaling of one global LM in both modes
    sharded = world > 1 or force_dist
    ctx = pkg.Context(local_rank)
    with_cpu = not a.no_cpu_baseline
    out = None

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()

    if not sharded and slam:
        # ---- the headline: BASELINE.json configs[3] as named -- 10k cams / 2k ellipsoids / ~200k edges, free cameras, Schur solve
        g, c, o, _ = pkg.synth.make_config(a.config, seed=0, slam=True)
        ctx.upload_graph(g); ctx.upload_states(c, o); ctx.snapshot_states()
        r = slam_run(pkg, ctx, g, c, o, solver, a.steps, a.warmup, jacobian=1 if a.jacobian == "analytic" else 0, barrier=barrier)
        out = {
            "metric": "LM iterations/sec (cams+ellipsoids)", "value": r["value"], "unit": "LM iterations/s",
            "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * r["_dt"] / a.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": slam_workload(a.config, g).replace("analytic", a.jacobian), "lm_iterations_per_step": r["lm_iterations_per_step"],
                       "lm_trials_per_step": r["lm_trials_per_step"], "linear_solver": r["linear_solver_name"],
                       "unknowns": r["unknowns"], "parallelism": "single GPU"},
            "kernel_ms": r["kernel_ms"], "chi2": r["chi2"], "roofline": r["roofline"], "host": host_info(),
        }
        if not a.no_extras:
            if r["linear_solver"] == 2:   # the elimination the north star names, beside the one that was timed: 2 steps after 1 warm-up
                rc_ = slam_run(pkg, ctx, g, c, o, 1, steps=2, warmup=1, jacobian=1 if a.jacobian == "analytic" else 0)
                out["slam_reduced_camera"] = {k: v for k, v in rc_.items() if not k.startswith("_")}
                out["slam_reduced_camera"]["workload"] = slam_workload(a.config, g)
            if a.jacobian == "analytic":
                # the C-ABI's DEFAULT Jacobians are the reference's (g2o's central differences, delta = 1e-9: base_binary_edge.hpp:147-197);
                # the timed region above runs the analytic ones -- the same steps with the faithful setting, beside it
                rn = slam_run(pkg, ctx, g, c, o, solver, steps=2, warmup=1, jacobian=0)
                out["slam_numeric_jacobians"] = {"value": rn["value"], "unit": "LM iterations/s", "ms_per_optimize": rn["ms_per_optimize"],
                                                 "lm_iterations_per_step": rn["lm_iterations_per_step"], "linearize_ms_per_launch":
                                                 rn["kernel_ms"].get("linearize", {}).get("total_ms", 0.0) / max(rn["kernel_ms"].get("linearize", {}).get("count", 1), 1),
                                                 "chi2_final": rn["chi2"]["final"],
                                                 "note": "numeric Jacobians as g2o computes them (the esl_lm_params default); 2 steps after 1 warm-up"}
            m, _ = mapping_bench(pkg, ctx, a.config, a.jacobian)
            out["mapping"] = m
            if a.jacobian == "analytic":
                mn, _ = mapping_bench(pkg, ctx, a.config, "numeric", steps=5, warmup=2, blocks=0, extra=False)
                out["mapping"]["numeric_jacobians"] = {"value": mn["value"], "unit": "LM iterations/s", "ms_per_step": mn["ms_per_step"],
                                                       "note": "the same graph with g2o's central differences (delta = 1e-9), the C-ABI default"}
            out["fit"] = fit_bench(pkg, ctx, with_cpu=with_cpu)
            out["ground_plane"] = ground_plane_bench(pkg, ctx, with_cpu=with_cpu)
            out["streaming_c5"] = streaming_bench(pkg, ctx)
            c3, cpu_rates = slam_c3_bench(pkg, ctx, with_cpu=with_cpu)
            out["slam_c3"] = c3
            if with_cpu and cpu_rates:
                out["cpu_baseline"] = cpu_baseline_c4_slam(g, cpu_rates, r["_its"], r["_trials"], r["roofline"].get("actual_flops_per_trial"))
                out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
                out["mapping"]["cpu_baseline"] = cpu_baseline(pkg, *mapping_graph(pkg, a.config), pkg.default_lm_params())
                out["mapping"]["speedup_vs_cpu_port"] = out["mapping"]["value"] / out["mapping"]["cpu_baseline"]["value"]
                out["speedup_note"] = ("GPU: analytic Jacobians; CPU port: g2o's numeric Jacobians, the same elimination order as the GPU; the C4 CPU figure "
                                       "is an estimate anchored on two measured runs (cpu_baseline.sample); a reported baseline, not a kernel-quality figure "
                                       "(that is roofline.frac / roofline.trial_frac)")
    elif not sharded:
        # ---- mapping mode as the timed region (--mode mapping): the reference as shipped
        m, (g, c, o) = mapping_bench(pkg, ctx, a.config, a.jacobian, steps=a.steps, warmup=a.warmup)
        out = {"metric": "LM iterations/sec (cams+ellipsoids)", "value": m["value"], "unit": "LM iterations/s", "n_gpus": 1, "steps": a.steps,
               "warmup": a.warmup, "ms_per_step": m["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
               "data": "synthetic", "config": {"workload": m["workload"], "lm_iterations_per_step": m["lm_iterations_per_step"],
                                               "lm_trials_per_step": m["lm_trials_per_step"], "parallelism": "single GPU"},
               "kernel_ms": m["kernel_ms"], "chi2": m["chi2"], "roofline": m["roofline"], "host": host_info()}
        if "repeat_blocks" in m:
            out["repeat_blocks"] = m["repeat_blocks"]
        if not a.no_extras:
            out["fit"] = fit_bench(pkg, ctx, with_cpu=with_cpu)
            out["ground_plane"] = ground_plane_bench(pkg, ctx, with_cpu=with_cpu)
            out["streaming_c5"] = streaming_bench(pkg, ctx)
        if with_cpu:
            out["cpu_baseline"] = cpu_baseline(pkg, g, c, o, pkg.default_lm_params())
            out["speedup_vs_cpu_port"] = out["value"] / out["cpu_baseline"]["value"]
    else:
        # ---- N ranks (one per GPU, RCCL): the ellipsoids of the ONE named graph partitioned over the ranks, cameras replicated
        # Mapping mode: the ellipsoids (with all their edges) are partitioned over the ranks, 64 bytes of LM scalars per trial cross.
        # SLAM mode: every rank holds the WHOLE graph (15 MB; its linearisation is 0.4 ms of a 350 ms trial) and the communicator
        # runs in replicated-graph mode: the ranks divide the dense solve -- each forms its own outer panels of the reduced ellipsoid
        # system (its share of the rank-59,994 MFMA update), the owner factors a panel and broadcasts it (esl_comm_set_replicated)
        g_full, c, o_full, _ = pkg.synth.make_config(a.config, seed=0, slam=slam)
        first_replicated = slam and os.environ.get("ESL_BENCH_SHARDED_SLAM") != "1"
        params = pkg.default_lm_params(jacobian_mode=1 if a.jacobian == "analytic" else 0, linear_solver=solver)

        def ranked_run(replicated, steps, warmup):
            """one N-rank run of the named graph: communicator (mode first, then the graph: esl.h ABI 4), warm-up, timed steps bracketed by
            barrier + synchronize, max over ranks; returns the JSON record on rank 0 (None elsewhere)"""
            mine = np.arange(g_full.n_objs) if replicated else np.nonzero(pkg.lib.partition_objects(g_full, world) == rank)[0]
            g, o = g_full.subset_objects(mine), o_full[mine]
            runner = None
            # preferred: the library's own RCCL exchange (collectives on its stream); fallback: the Python step-API driver with
            # torch.distributed collectives
            try:
                if os.environ.get("ESL_BENCH_PY_EXCHANGE") == "1":
                    raise RuntimeError("python exchange requested")
                ctx.comm_destroy()
                if host_transport:
                    def gloo_sum(buf):   # in place over the ranks: the callback contract of esl_comm_init_host
                        dist.all_reduce(torch.from_numpy(buf))
                    ctx.comm_init_host(world, rank, gloo_sum)
                    exchange = "host transport (esl_comm_init_host over gloo): NOT a timing"
                else:
                    uid = [pkg.lib.comm_unique_id() if rank == 0 else None]
                    dist.broadcast_object_list(uid, src=0)
                    ctx.comm_init(world, rank, uid[0])
                    exchange = "rccl-native"
                if replicated:
                    ctx.comm_set_replicated(True)
                    exchange += ", replicated graph: broadcast of the factored panels (+ an 8-byte all-reduce of the pivot flag) per trial"
            except Exception as e:  # noqa: BLE001
                if replicated:
                    raise
                print(f"[bench] native RCCL exchange unavailable ({e}); using torch.distributed step driver", file=sys.stderr)
                par = importlib.import_module("object-oriented-slam_amd.parallel")
                runner = par.ShardedLM(ctx, dist, device=torch.device("cuda", local_rank), force_collectives=force_dist)
                exchange = "torch.distributed all_gather"
            ctx.upload_graph(g); ctx.upload_states(c, o); ctx.snapshot_states()
            # (ellipsoid shards cannot run the camera-first form -- esl.h ESL_SOLVER_REDUCED_ELLIPSOID --: a forced --solver ellipsoid applies to the replicated run only)
            prm = params if (replicated or solver != 2) else pkg.default_lm_params(jacobian_mode=params.jacobian_mode, linear_solver=0)

            def one_step():
                ctx.restore_states()
                return runner.optimize(prm) if runner is not None else ctx.optimize_resident(prm)
            for _ in range(warmup):
                one_step()
            ctx.profile_enable(0 if os.environ.get("ESL_BENCH_NO_PROFILE") == "1" else (2 if slam else 1))
            barrier()
            t0 = time.perf_counter()
            iters = trials = 0
            rep = None
            for _ in range(steps):
                rep = one_step()
                iters += rep["iterations"]; trials += rep["total_trials"]
            barrier()
            dt = time.perf_counter() - t0
            prof = ctx.profile_get()
            ctx.profile_enable(False)
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if host_transport else f"cuda:{local_rank}")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            if rank != 0:
                return None
            if slam:
                n_c, n_o = 6 * int((~g_full.cam_fixed.astype(bool)).sum()), 9 * g_full.n_objs
                used = ctx.lm_solver_used()
                st = ctx.lm_solver_stats()
                roof = slam_roofline(prof, n_c, n_o, used, trials, st, dt=dt)
                if used == 2:   # rank 0 ran its share (its own outer panels: one update launch per owned panel, a launch-per-step factorisation)
                    rk = prof.get("rank_k_update", dict(count=0, total_ms=0.0))
                    avg = rk["total_ms"] / max(rk["count"], 1)
                    fl = slam_flops(n_c, n_o, 2, st)["rank_k_update"] / world
                    roof = {"kernel": "k_chol_update_v: rank 0's launches of the separators' rank-K update (one per owned outer panel of T)",
                            "bound": "mfma", "achieved": fl / (avg * 1e-3) / 1e12 if avg > 0 else 0.0, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                            "traffic": None, "algorithmic_flops_per_launch": fl, "avg_launch_ms": avg, "launches": rk["count"],
                            "trial_ms": roof.get("trial_ms"), "trial_frac_of_n_gpus_peak": (roof["trial_frac"] / world) if roof.get("trial_frac") else None,
                            "kernel_ms_rank0": {k: v["total_ms"] / max(trials, 1) for k, v in prof.items()}}
                    roof["frac"] = roof["achieved"] / FP64_MFMA_PEAK_TF
                roof["note"] = ("rank 0's launches; every rank executes 1 / n_gpus of the update flops (its own outer panels) and the factorisation is distributed"
                                if replicated else "rank 0's launches; ellipsoid shards: partial reduced camera systems summed over the ranks, factorisation distributed")
                wl = slam_workload(a.config, g_full)
            else:
                lin = prof.get("linearize", dict(count=0, total_ms=0.0))
                avg_ms = lin["total_ms"] / max(lin["count"], 1)
                abytes = algorithmic_bytes_linearize(g, False)
                ach = abytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
                roof = {"kernel": "k_chunk_linearize_both (rank 0's shard)", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": abytes, "avg_launch_ms": avg_ms, "launches": lin["count"]}
                wl = (f"{a.config} synthetic graph: {g_full.n_cams} cams, {g_full.n_objs} ellipsoids, {len(g_full.bbox_cam)} bbox + {len(g_full.e3d_cam)} 3-D + "
                      f"{len(g_full.grav_obj)} gravity edges; mapping mode; {a.jacobian} Jacobians; optimize(10) per step")
            return {"metric": "LM iterations/sec (cams+ellipsoids)", "value": iters / dt, "unit": "LM iterations/s", "n_gpus": world, "steps": steps,
                    "warmup": warmup, "ms_per_step": 1e3 * dt / steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                    "dtype": "f64", "data": "synthetic",
                    "config": {"workload": wl, "lm_iterations_per_step": iters / steps, "lm_trials_per_step": trials / steps,
                               "parallelism": (f"replicated graph, dense solve divided over {world} ranks" if replicated else f"ellipsoid-sharded x{world}"), "lm_scalar_exchange": exchange,
                               "linear_solver": SOLVER_NAMES.get(ctx.lm_solver_used(), "-") if slam else "per-ellipsoid 9x9 blocks"},
                    "kernel_ms": prof, "chi2": {"initial": rep["chi2_initial"], "final": rep["chi2_final"], "trace": rep["trace_chi2"], "trials": rep["trace_trials"]},
                    "roofline": roof, "host": host_info()}

        out = ranked_run(first_replicated, a.steps, a.warmup)
        if slam and first_replicated and os.environ.get("ESL_BENCH_ONE_LINE") != "1" and not a.no_extras:
            # the design BASELINE.json's north_star NAMES, beside the one that was timed: the ellipsoids partitioned over
            # the ranks, camera blocks all-reduced, every rank's partial reduced CAMERA system summed panel by panel to its owner, the
            # factorisation distributed.  16x the flops of the camera-first form at this shape: 1 step after 1 warm-up.
            try:
                ctx.trim()
                second = ranked_run(False, 1, 1)
                if rank == 0:
                    out["slam_ellipsoid_partition"] = {k: v for k, v in second.items() if k != "host"}
            except Exception as e:  # noqa: BLE001  (the timed line above must survive a failure of the extra record)
                if rank == 0:
                    out["slam_ellipsoid_partition"] = {"error": str(e)}
    final_line = None
    if rank == 0 and out is not None:
        final_line = json.dumps(driver_line(out, os.path.relpath(a.extras, ROOT) if os.path.abspath(a.extras).startswith(ROOT) else a.extras))
        assert len(final_line) < DRIVER_LINE_MAX, len(final_line)
        full = json.dumps(out)
        try:
            with open(a.extras, "w") as f:
                f.write(full + "\n")
        except OSError as e:
            print(f"[bench] could not write {a.extras}: {e}", file=sys.stderr)
        print("[bench] full record (also in %s):\n%s" % (a.extras, full), file=sys.stderr)
    if sharded:
        dist.destroy_process_group()
    ctx.close()
    if final_line is not None:
        os.write(json_fd, (final_line + "\n").encode())
    os.close(json_fd)


def mapping_graph(pkg, config):
    g, c, o, _ = pkg.synth.make_config(config, seed=0, slam=False)
    return g, c, o


if __name__ == "__main__":
    main()
