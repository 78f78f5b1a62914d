"""bench.py's N-rank branch across REAL process boundaries on a box with one GPU ("make the N > 1 path executable
before a node ever sees it"): `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` with
ESL_BENCH_HOST_TRANSPORT=1 -- both ranks on device 0, the library's collectives over esl_comm_init_host with a gloo all-reduce
behind the callback.  What runs is what the driver's 8-GPU launch runs except for the wire: rank / world from the environment,
the replicated-graph communicator (esl_comm_set_replicated), the distributed factorisation with its panel messages on their own
stream, the ellipsoid-sharded forms, the max-over-ranks timing and the single JSON line from rank 0.  Every mode must take the
LM run of `--gpus 1` (same accept / reject sequence, same chi2 trace up to the summation order of the sharded sums)."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON_DEFAULT = ["--steps", "1", "--warmup", "0", "--config", "C3", "--no-cpu-baseline", "--no-extras"]


DRIVER_LINE_MAX = 4096
LINE_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"}


def run_bench(extra_args, n_ranks, env_extra, port, common=None):
    """returns the FULL record (the side file `--extras`), after checking the one stdout line the driver parses: < 4 KB, the
    contract's keys, and the same numbers as the full record (a 24 KB line was dropped by the driver)"""
    COMMON = list(common if common is not None else COMMON_DEFAULT)
    extras = os.path.join(tempfile.mkdtemp(prefix="esl_bench_"), "extras.json")
    COMMON += ["--extras", extras]
    env = dict(os.environ)
    env.update(env_extra)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    if n_ranks == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + COMMON + extra_args
    else:
        env["ESL_BENCH_HOST_TRANSPORT"] = "1"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n_ranks)] + COMMON + extra_args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])      # rank 0 prints ONE line, the other ranks nothing
    assert len(lines[0]) < DRIVER_LINE_MAX, len(lines[0])
    line = json.loads(lines[0])
    assert LINE_KEYS <= set(line), LINE_KEYS - set(line)
    assert {"kernel", "bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"])
    assert "workload" in line["config"]
    full = json.load(open(extras))
    for k in ("value", "ms_per_step", "n_gpus", "steps", "warmup"):
        assert full[k] == line[k]
    return full


@pytest.fixture(scope="module")
def single():
    return {mode: run_bench(["--mode", mode], 1, {}, 0) for mode in ("slam", "mapping")}


@pytest.mark.parametrize("mode,env,tol", [
    ("slam", {"ESL_CHOL_DIST": "1"}, 1e-9),                                   # replicated graph, reduced ELLIPSOID system divided over the ranks, overlapped panel messages
    ("slam", {"ESL_CHOL_DIST": "1", "ESL_CHOL_DIST_OVERLAP": "0"}, 1e-9),     # ... with the messages on the compute stream
    ("slam", {"ESL_BENCH_SHARDED_SLAM": "1", "ESL_CHOL_DIST": "1"}, 1e-7),    # ellipsoid shards: all-reduce of the camera blocks, per-panel reduce, distributed factorisation of S
    ("slam", {"ESL_BENCH_SHARDED_SLAM": "1", "ESL_CHOL_DIST": "0"}, 1e-7),    # ... with the all-reduced S factored on every rank
    ("mapping", {}, 1e-11),                                                   # ellipsoid shards, one 64-byte all-gather per trial
])
def test_two_processes_on_one_gpu_take_the_single_gpu_run(single, mode, env, tol):
    port = 29600 + (os.getpid() % 300) + 7 * len(env)
    multi = run_bench(["--mode", mode], 2, env, port)
    ref = single[mode]
    assert multi["n_gpus"] == 2 and multi["steps"] == 1 and multi["scaling"] == "strong" and multi["value"] > 0
    assert "host transport" in multi["config"]["lm_scalar_exchange"]
    print("bench.py --gpus 2 (%s, %s): %s | chi2 %s" % (mode, env, multi["config"]["parallelism"], multi["chi2"]["trace"]))
    assert multi["chi2"]["trials"] == ref["chi2"]["trials"]
    np.testing.assert_allclose(multi["chi2"]["trace"], ref["chi2"]["trace"], rtol=tol)
    assert multi["chi2"]["initial"] == pytest.approx(ref["chi2"]["initial"], rel=1e-12)
    if mode == "slam" and "ESL_BENCH_SHARDED_SLAM" not in env:
        assert "replicated graph" in multi["config"]["parallelism"]


def test_two_rank_default_line_carries_both_slam_designs(single):
    """`bench.py --gpus N` as the driver launches it (no --no-extras) reports the timed replicated-graph run AND, as the record
    `slam_ellipsoid_partition`, the design BASELINE.json's north_star names: ellipsoids partitioned over the ranks, camera blocks
    all-reduced, the partial reduced camera systems summed to the panels' owners.  Both must take the
    single-GPU LM run."""
    port = 29600 + (os.getpid() % 300) + 71
    multi = run_bench(["--mode", "slam"], 2, {"ESL_CHOL_DIST": "1"}, port, common=["--steps", "1", "--warmup", "0", "--config", "C3", "--no-cpu-baseline"])
    ref = single["slam"]
    assert "replicated graph" in multi["config"]["parallelism"]
    sec = multi["slam_ellipsoid_partition"]
    assert "ellipsoid-sharded x2" in sec["config"]["parallelism"] and sec["n_gpus"] == 2 and sec["value"] > 0
    assert "reduced camera system" in sec["config"]["linear_solver"]
    for rec, tol in ((multi, 1e-9), (sec, 1e-7)):
        assert rec["chi2"]["trials"] == ref["chi2"]["trials"]
        np.testing.assert_allclose(rec["chi2"]["trace"], ref["chi2"]["trace"], rtol=tol)
    print("both designs in one line: replicated %.1f it/s, ellipsoid partition %.1f it/s (host transport: not timings)" % (multi["value"], sec["value"]))
