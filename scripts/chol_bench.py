#!/usr/bin/env python
"""Micro-benchmark of the dense FP64-MFMA Cholesky (esl_selftest_cholesky): TFLOP/s for a few sizes."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("object-oriented-slam_amd")
ctx = pkg.Context(0)
sizes = [int(a) for a in sys.argv[1:]] or [2994, 8192, 16384, 24576]
for n in sizes:
    best = None
    for _ in range(2):
        ms, res = ctx.selftest_cholesky(n)
        best = ms if best is None else min(best, ms)
    flops = n ** 3 / 3.0 + 2.0 * n * n
    print(f"n={n:6d}  {best:9.2f} ms  {flops / best / 1e9:7.2f} TFLOP/s  ({100 * flops / best / 1e9 / 78.6:5.1f} % of 78.6)  residual {res:.1e}", flush=True)
