"""Helpers of tests/test_harness.py: a minimal 16-bit PNG writer (zlib only) and the reconstruction of the reference's demo
clip in its on-disk layout from the committed fixture tests/golden/cabinet_fit_np.npz (depth cropped to the detection's box:
the pipeline never reads a pixel outside it)."""
import os
import struct
import subprocess
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_png16(path, img, filters=None):
    """img: (h, w) uint16.  filters: per-row PNG filter types (0-4) to exercise the reader; default 0."""
    img = np.ascontiguousarray(img, dtype=np.uint16)
    h, w = img.shape
    raw = img.astype(">u2").tobytes()
    stride = 2 * w
    out = bytearray()
    prev = bytes(stride)
    for y in range(h):
        row = raw[y * stride:(y + 1) * stride]
        ft = 0 if filters is None else int(filters[y % len(filters)])
        enc = bytearray(stride)
        for x in range(stride):
            a = row[x - 2] if x >= 2 else 0
            b = prev[x]
            c = prev[x - 2] if x >= 2 else 0
            if ft == 0: p = 0
            elif ft == 1: p = a
            elif ft == 2: p = b
            elif ft == 3: p = (a + b) // 2
            else:
                pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            enc[x] = (row[x] - p) & 0xff
        out.append(ft); out += enc
        prev = row

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(bytes(out), 6)) + chunk(b"IEND", b"")
    open(path, "wb").write(png)


def write_png16_fast(path, img):
    """filter 0 only, vectorised"""
    img = np.ascontiguousarray(img, dtype=np.uint16)
    h, w = img.shape
    rows = np.zeros((h, 2 * w + 1), dtype=np.uint8)
    rows[:, 1:] = img.astype(">u2").view(np.uint8).reshape(h, 2 * w)

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(rows.tobytes(), 1)) + chunk(b"IEND", b"")
    open(path, "wb").write(png)


def unpack_depth(diff, meta):
    x0, y0, w, h = [int(v) for v in meta]
    crop = np.cumsum(diff.astype(np.uint16), axis=1, dtype=np.uint16)
    depth = np.zeros((h, w), dtype=np.uint16)
    depth[y0:y0 + crop.shape[0], x0:x0 + crop.shape[1]] = crop
    return depth


def write_cabinet_clip(G, out_dir):
    """The clip in the layout src/tum_rgbd/io.cpp reads: rgb/<stamp>.jpg (names only), depth/<stamp>.png, groundtruth.txt,
    associate.txt, associateGroundtruth.txt, bbox/<stamp>.txt (id x1 y1 x2 y2 label rate instance; README.md:69)."""
    for d in ("rgb", "depth", "bbox"):
        os.makedirs(os.path.join(out_dir, d), exist_ok=True)
    names = [str(n) for n in G["frame_names"]]
    det_of_frame = {int(f): k for k, f in enumerate(G["det_frame"])}
    with open(os.path.join(out_dir, "groundtruth.txt"), "w") as gt, open(os.path.join(out_dir, "associate.txt"), "w") as asc, \
            open(os.path.join(out_dir, "associateGroundtruth.txt"), "w") as ag:
        for i, n in enumerate(names):
            pose = " ".join("%.4f" % v for v in G["frame_poses"][i])
            stamp6 = "%.6f" % float(n)
            gt.write(f"{n} {pose}\n")
            asc.write(f"{stamp6} rgb/{n}.jpg {stamp6} depth/{n}.png\n")
            ag.write(f"{stamp6} rgb/{n}.jpg {stamp6} {pose}\n")
            open(os.path.join(out_dir, "rgb", n + ".jpg"), "wb").close()
            k = det_of_frame.get(i)
            if k is None:
                write_png16_fast(os.path.join(out_dir, "depth", n + ".png"), np.zeros((480, 640), np.uint16))
                open(os.path.join(out_dir, "bbox", n + ".txt"), "w").close()
            else:
                write_png16_fast(os.path.join(out_dir, "depth", n + ".png"), unpack_depth(G[f"depth_{k}"], G[f"depth_meta_{k}"]))
                b = G["boxes"][k]
                with open(os.path.join(out_dir, "bbox", n + ".txt"), "w") as f:
                    f.write("0 %g %g %g %g %d %g 0\n" % (b[0], b[1], b[2], b[3], int(G["labels"][k]), float(G["rates"][k])))


def build_oracle_harness(tmp_path):
    exe = str(tmp_path / "harness_oracle")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "harness_oracle_main.cpp"),
                           "-L", os.path.join(ROOT, "oracle"), "-lesl_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lz", "-o", exe])
    return exe


def read_table(path):
    return [[float(v) for v in l.split()] for l in open(path) if l.strip()]
