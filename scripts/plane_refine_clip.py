#!/usr/bin/env python
"""What PCL's refinement pass (segmentAndRefine's second half, PlaneExtractor.cpp:82) changes on the reference's demo clip:
every depth frame of Example/dataset/cabinet through the CPU checker with and without the pass -- planes found, size of the ground
plane, whether another plane becomes the ground, and where the ground lies in the world (mocap pose of the frame).
Development container only (reads the reference's DATA files).

    python scripts/plane_refine_clip.py > profiles/r5_plane_refine_clip.txt
"""
import importlib
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/Example/dataset/cabinet/"


def main():
    abi = importlib.import_module("object-oriented-slam_amd.abi")
    from oracle import pyoracle as po
    from oracle import np_fit
    intr = np.array([535.4, 539.2, 320.1, 247.6, 5000.0])
    lines = [l.split() for l in open(REF + "associateGroundtruth.txt")]
    assoc = {a.split()[0]: a.split()[3] for a in open(REF + "associate.txt")}
    print("frame | planes (no pass -> with) | ground pixels (no pass -> with, growth) | same ground plane | world floor: angle to z [deg], offset [m]")
    grow, changed, more = [], 0, 0
    for k, l in enumerate(lines):
        depth = np.array(Image.open(REF + assoc[l[0]])).astype(np.uint16)
        a = po.extract_ground_plane(depth, intr, abi.default_plane_params(refine=0))
        b = po.extract_ground_plane(depth, intr, abi.default_plane_params(refine=1))
        same = a["ok"] == b["ok"] and np.array_equal(a["plane"], b["plane"])
        T = np_fit.se3_mat(np.array(l[3:10], float))
        n = T[:3, :3] @ b["plane"][:3]
        d = b["plane"][3] - T[:3, 3] @ n
        g = (b["n_pixels"] / a["n_pixels"] - 1) if a["n_pixels"] else float("nan")
        grow.append(g); changed += not same; more += b["n_planes"] > a["n_planes"]
        print("%5d | %d -> %d | %6d -> %6d  %+5.1f %% | %s | %.2f  %+.3f" % (k, a["n_planes"], b["n_planes"], a["n_pixels"], b["n_pixels"], 100 * g,
                                                                          "yes" if same else "NO", np.degrees(np.arccos(min(1.0, abs(n[2])))), d))
    print("\n%d frames: the pass grows the ground plane's inlier list by %.1f %% on average (%.1f .. %.1f %%), finds more planes on %d frames, and changes WHICH "
          "plane is the ground on %d frames.  The coefficients of a plane never move (PCL does not re-estimate them), so the floor the tracker gets "
          "is the same plane whenever the same segment wins." % (len(lines), 100 * np.nanmean(grow), 100 * np.nanmin(grow), 100 * np.nanmax(grow), more, changed))


if __name__ == "__main__":
    main()
