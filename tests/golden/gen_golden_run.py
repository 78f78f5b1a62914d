#!/usr/bin/env python
"""Generates tests/golden/cabinet_run.npz: the reference's demo clip (Example/dataset/cabinet, read from /root/reference in the
development container) end to end through the Tracking-side harness with the CPU checker as its backend
(tests/harness_oracle_main.cpp): the final objects.txt rows, object_history.txt and one graph summary per optimisation —
the counts the reference prints at src/core/Optimizer.cpp:282-288.  Two settings: the reference's (numeric Jacobians at
delta = 1e-9, 5-iteration symmetry LM) and a noise-free one (delta = 1e-6, no symmetry LM) for tight comparisons; and
(round 3) the same two with the reference's bSLAM_mode branch switched on (Optimizer.cpp:126-158: frame 0 fixed, odometry
edges; the checker then solves with the faithful dense pivoted LDLT), which also records the camera poses of the last run.

  python tests/golden/gen_golden_run.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness_util as hu  # noqa: E402

REF = "/root/reference/Example/dataset/cabinet/"


def main():
    out = {}
    with tempfile.TemporaryDirectory() as td:
        import pathlib
        exe = hu.build_oracle_harness(pathlib.Path(td))
        for tag, args in (("ref", []), ("tight", ["--delta", "1e-6", "--sym-iters", "0"]), ("slamref", ["--slam-mode"]),
                          ("slamtight", ["--slam-mode", "--delta", "1e-6", "--sym-iters", "0"])):
            od = os.path.join(td, tag)
            os.makedirs(od)
            print(subprocess.check_output([exe, REF, od, "--ground", "0", "0", "1", "0"] + args).decode().strip())
            out[tag + "_objects"] = np.array(hu.read_table(os.path.join(od, "objects.txt")))
            out[tag + "_graph_log"] = np.array(hu.read_table(os.path.join(od, "graph_log.txt")))
            out[tag + "_history"] = np.array(open(os.path.join(od, "object_history.txt")).read())
            if "slam" in tag:
                out[tag + "_cameras"] = np.array(hu.read_table(os.path.join(od, "cameras_slam.txt")))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cabinet_run.npz"), **out)
    print("objects (reference settings):", out["ref_objects"])


if __name__ == "__main__":
    main()
