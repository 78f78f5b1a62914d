#!/usr/bin/env python
"""Generates tests/golden/cabinet_plane.npz: the ground plane of two depth frames of the reference's demo clip
(Example/dataset/cabinet, first frame and frame 35; round 6: frame 30, used until then, has floor and cabinet chained into ONE segment of curvature 0.034, which PCL's maximum_curvature_ = 0.001 drops -- no ground plane there) computed by the INDEPENDENT numpy / scipy version oracle/np_plane.py.
Run in the development container only (reads the reference's DATA files; needs scipy + PIL).  Stored: the two depth
images (x-differenced so that deflate gets PNG-like ratios), the mocap poses of the frames, the expected plane, plane
count, pixel count, the planes' sizes and the count of pixels with a normal -- with PCL's refinement pass (round 5) and, for the
segments alone, the round 2-4 counts.  No reference source text is stored.

  python tests/golden/gen_golden_plane.py
"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/Example/dataset/cabinet/"
OUT = os.path.join(ROOT, "tests", "golden", "cabinet_plane.npz")


def main():
    from oracle import np_plane
    intr = np.array([535.4, 539.2, 320.1, 247.6, 5000.0])       # Example/param/TUM3.yaml:62-70
    lines = [l.split() for l in open(REF + "associateGroundtruth.txt")]
    assoc = {a.split()[0]: a.split()[3] for a in open(REF + "associate.txt")}
    store = dict(intr=intr)
    for k, li in enumerate((0, 35)):
        l = lines[li]
        depth = np.array(Image.open(REF + assoc[l[0]])).astype(np.uint16)
        diff = depth.copy()
        diff[:, 1:] = depth[:, 1:] - depth[:, :-1]
        r = np_plane.extract_ground_plane(depth, intr)                      # with PCL's refinement pass (the reference: segmentAndRefine)
        r0 = np_plane.extract_ground_plane(depth, intr, refine=False)        # the segments alone (rounds 2-4)
        store[f"depth_{k}"] = diff
        store[f"pose_{k}"] = np.array(l[3:10], float)
        store[f"plane_{k}"] = r["plane"]
        store[f"counts_{k}"] = np.array([r["n_planes"], r["n_pixels"], int(np.isfinite(r["normals"][..., 0]).sum())])
        store[f"sizes_{k}"] = r["sizes"]
        store[f"counts_norefine_{k}"] = np.array([r0["n_planes"], r0["n_pixels"]])
        assert np.array_equal(r["plane"], r0["plane"])                       # the pass grows the inlier lists, not the coefficients
        print(li, r["plane"], r["n_planes"], r["n_pixels"], list(r["sizes"]), "| without the refinement pass:", r0["n_planes"], r0["n_pixels"])
    np.savez_compressed(OUT, **store)
    print(OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    main()
