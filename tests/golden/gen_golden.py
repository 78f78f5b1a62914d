#!/usr/bin/env python
"""Generates the committed fixtures under tests/golden/ (run in the development container only: it reads the
reference's demo DATA files under /root/reference/Example/dataset/cabinet — depth PNGs, bbox txt, groundtruth —
and stores inputs + the CPU restatement's outputs).  No reference source text is stored.

  python tests/golden/gen_golden.py
"""
import importlib
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/Example/dataset/cabinet/"
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    pkg = importlib.import_module("object-oriented-slam_amd")
    from oracle import pyoracle as po
    # ---- cabinet frames ------------------------------------------------------
    gt = {l.split()[0]: np.array(l.split()[1:], float) for l in open(REF + "groundtruth.txt") if not l.startswith("#")}
    frames = ["1341841278.8427", "1341841300.3656"]
    depth, boxes, labels, poses = [], [], [], []
    for name in frames:
        depth.append(np.array(Image.open(REF + "depth/" + name + ".png")).astype(np.uint16))
        row = open(REF + "bbox/" + name + ".txt").read().split()
        boxes.append([float(v) for v in row[1:5]]); labels.append(int(row[5]))
        poses.append(gt[name])
    intr = np.array([535.4, 539.2, 320.1, 247.6, 5000.0])       # Example/param/TUM3.yaml:62-70
    ground = np.array([0.0, 0.0, 1.0, 0.0])                     # mocap world: floor = z 0 (stand-in for PlaneExtractor's output)
    exp = {}
    for iters in (0, 5):
        for fi in range(len(frames)):
            e, p, st, dbg = po.fit_frame(depth[fi], [boxes[fi]], [labels[fi]], poses[fi], intr, ground,
                                         po.default_fit_params(symmetry_lm_iters=iters))
            exp[f"ell_{fi}_{iters}"] = e[0]; exp[f"prob_{fi}_{iters}"] = p[0]; exp[f"status_{fi}_{iters}"] = st[0]
            exp[f"dbg_{fi}_{iters}"] = dbg[0]
    np.savez_compressed(os.path.join(OUT, "cabinet_fit.npz"), depth=np.array(depth), boxes=np.array(boxes),
                        labels=np.array(labels), poses=np.array(poses), intr=intr, ground=ground, **exp)
    # ---- optimiser: per-edge residuals / Jacobians and LM traces from the C restatement ------------------
    rng = np.random.default_rng(0)
    K = pkg.synth.TUM3_K
    g, c, o, _ = pkg.synth.make_graph(20, 5, 80, seed=0)
    vec = dict(K=np.array(K), cams=c, objs=o, bbox_cam=g.bbox_cam, bbox_obj=g.bbox_obj, bbox_meas=g.bbox_meas.reshape(-1, 4))
    vec["res_bbox"] = np.array([po.res_bbox(c[i], o[j], K, m) for i, j, m in zip(g.bbox_cam, g.bbox_obj, g.bbox_meas.reshape(-1, 4))])
    vec["jac_bbox_obj"] = np.array([po.jac_bbox(c[i], o[j], K, m, 1e-6)[1] for i, j, m in zip(g.bbox_cam, g.bbox_obj, g.bbox_meas.reshape(-1, 4))])
    vec["res_e3d"] = np.array([po.res_e3d(c[i], o[j], m) for i, j, m in zip(g.e3d_cam, g.e3d_obj, g.e3d_meas.reshape(-1, 10))])
    vec["e3d_cam"] = g.e3d_cam; vec["e3d_obj"] = g.e3d_obj; vec["e3d_meas"] = g.e3d_meas.reshape(-1, 10)
    for seed in range(3):
        for slam in (False, True):
            gg, cc, oo, _ = pkg.synth.make_graph(20, 5, 80, seed=seed, slam=slam)
            co, ob, rep = po.optimize(gg, cc, oo, pkg.default_lm_params(numeric_delta=1e-6), solver=0)
            tag = f"lm_{seed}_{'slam' if slam else 'map'}"
            vec[tag + "_objs"] = ob; vec[tag + "_cams"] = co
            vec[tag + "_chi2"] = np.array(rep["trace_chi2"]); vec[tag + "_trials"] = np.array(rep["trace_trials"])
    np.savez_compressed(os.path.join(OUT, "optimizer_vectors.npz"), **vec)
    print("wrote", os.listdir(OUT))


if __name__ == "__main__":
    main()
