#!/usr/bin/env python
"""Generates tests/golden/cabinet_fit_np.npz: the single-frame fit of ALL detections of the reference's demo clip
(Example/dataset/cabinet: 58 frames, 51 non-empty bbox files) computed by the INDEPENDENT numpy/scipy restatement
oracle/np_fit.py — not by the C restatement the product is otherwise compared with.  Run in the development
container only (reads the reference's DATA files; needs scipy + PIL).  Stored: inputs (depth cropped to the bbox — the
fit never reads a pixel outside it —, x-differenced so that deflate gets PNG-like ratios; bbox rows; ground-truth
poses) and expected outputs.  No reference source text is stored.

  python tests/golden/gen_golden_np.py
"""
import glob
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/Example/dataset/cabinet/"
OUT = os.path.join(ROOT, "tests", "golden", "cabinet_fit_np.npz")


def pack_depth(depth, box):
    """crop to the bbox (+1 px), difference along x with uint16 wrap-around"""
    h, w = depth.shape
    x0, y0 = max(int(box[0]) - 1, 0), max(int(box[1]) - 1, 0)
    x1, y1 = min(int(box[2]) + 2, w), min(int(box[3]) + 2, h)
    crop = depth[y0:y1, x0:x1].astype(np.uint16)
    diff = crop.copy()
    diff[:, 1:] = crop[:, 1:] - crop[:, :-1]
    return diff, np.array([x0, y0, w, h], dtype=np.int32)


def unpack_depth(diff, meta):
    x0, y0, w, h = [int(v) for v in meta]
    crop = np.cumsum(diff.astype(np.uint16), axis=1, dtype=np.uint16)
    depth = np.zeros((h, w), dtype=np.uint16)
    depth[y0:y0 + crop.shape[0], x0:x0 + crop.shape[1]] = crop
    return depth


def main():
    from oracle import np_fit
    gt = {l.split()[0]: np.array(l.split()[1:], float) for l in open(REF + "groundtruth.txt") if not l.startswith("#")}
    intr = np.array([535.4, 539.2, 320.1, 247.6, 5000.0])       # Example/param/TUM3.yaml:62-70
    ground = np.array([0.0, 0.0, 1.0, 0.0])                     # mocap world: floor = z 0 (stand-in for PlaneExtractor's output)
    names = sorted(os.path.basename(f)[:-4] for f in glob.glob(REF + "bbox/*.txt"))
    store = dict(intr=intr, ground=ground)
    det_frame, boxes, labels, rates, poses = [], [], [], [], []
    keys = ("status", "ell", "prob", "counts", "hyp_prob", "sym_type", "plane_world", "plane2_world", "center_world")
    exp = {f"{k}_{it}": [] for k in keys for it in (0, 5)}
    frame_names, frame_poses = [], []
    for fi, name in enumerate(names):
        frame_names.append(name)
        frame_poses.append(gt[name])
        rows = [l.split() for l in open(REF + "bbox/" + name + ".txt") if l.strip()]
        if not rows:
            continue
        depth = np.array(Image.open(REF + "depth/" + name + ".png")).astype(np.uint16)
        for row in rows:
            box = [float(v) for v in row[1:5]]
            k = len(boxes)
            d, meta = pack_depth(depth, box)
            assert np.array_equal(unpack_depth(d, meta)[int(box[1]):int(box[3]) + 1, int(box[0]):int(box[2]) + 1],
                                  depth[int(box[1]):int(box[3]) + 1, int(box[0]):int(box[2]) + 1])
            store[f"depth_{k}"] = d
            store[f"depth_meta_{k}"] = meta
            det_frame.append(fi); boxes.append(box); labels.append(int(row[5])); rates.append(float(row[6])); poses.append(gt[name])
            for it in (0, 5):
                r = np_fit.fit_one(unpack_depth(d, meta), box, int(row[5]), gt[name], intr, ground, symmetry_lm_iters=it)
                for kk in keys:
                    exp[f"{kk}_{it}"].append(r[kk])
            print(name, "status", r["status"], "prob %.6f" % r["prob"], flush=True)
    # the clip only holds label 28 (dual reflection): re-run every 5th detection as a single-reflection class (62) and as
    # a class without symmetry (58) so that those branches of EstimateLocalEllipsoid are pinned too (outputs only)
    alt = {f"alt_{k}": [] for k in keys}
    alt_det, alt_label = [], []
    for k in range(0, len(boxes), 5):
        for lab in (62, 58):
            r = np_fit.fit_one(unpack_depth(store[f"depth_{k}"], store[f"depth_meta_{k}"]), boxes[k], lab, poses[k], intr, ground,
                               symmetry_lm_iters=5)
            alt_det.append(k); alt_label.append(lab)
            for kk in keys:
                alt[f"alt_{kk}"].append(r[kk])
    store.update({k: np.array(v) for k, v in alt.items()})
    store.update(alt_det=np.array(alt_det), alt_label=np.array(alt_label))
    store.update(det_frame=np.array(det_frame), boxes=np.array(boxes), labels=np.array(labels), rates=np.array(rates),
                 poses=np.array(poses), frame_names=np.array(frame_names), frame_poses=np.array(frame_poses))
    store.update({k: np.array(v) for k, v in exp.items()})
    np.savez_compressed(OUT, **store)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(boxes), "detections in", len(names), "frames")


if __name__ == "__main__":
    main()
