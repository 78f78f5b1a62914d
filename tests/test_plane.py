"""Ground-plane extraction: esl_extract_ground_plane against the CPU checker, the independent numpy / scipy
version and the fixture made from two depth frames of the reference's demo clip (tests/golden/gen_golden_plane.py).

Replaces PlaneExtractor::extractGroundPlane (reference src/plane/PlaneExtractor.cpp:107-183).  The PCL steps inside it are
restated by the definitions at the top of csrc/esl_plane.hip (parity with PCL itself is unpinned); everything after them
follows the reference: segments of >= Plane.MinSize pixels, d >= 0, the wall filter around the camera's y axis, the largest
segment wins.

Tolerances: the segmentation is integer work on float32 normals computed in one fixed order, so pixel / segment counts are
exact between the HIP path and the C checker; the plane comes from fixed-point moments that are identical on both sides
(1e-9).  The numpy version sums in a different order and in floating point: counts exact on these inputs, plane 1e-5."""
import os

import numpy as np
import pytest

from oracle import np_plane
from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INTR = np.array([535.4, 539.2, 320.1, 247.6, 5000.0])


def fixture(extra=False):
    z = np.load(os.path.join(ROOT, "tests", "golden", "cabinet_plane.npz"))
    frames = []
    for k in range(2):
        depth = np.cumsum(z[f"depth_{k}"].astype(np.uint16), axis=1, dtype=np.uint16)
        frames.append((depth, z[f"pose_{k}"], z[f"plane_{k}"], z[f"counts_{k}"]) + ((z[f"sizes_{k}"], z[f"counts_norefine_{k}"]) if extra else ()))
    return z["intr"], frames


def _abi():
    from importlib import import_module
    return import_module("object-oriented-slam_amd.abi")


def scene(h=240, w=320, tilt_deg=25.0, height=1.3, wall=3.0, noise=0.0, seed=0, hole=True, floor=True):
    """a floor seen from a camera pitched down by tilt_deg at `height` above it, a fronto-parallel wall behind"""
    fx = fy = 535.4 * w / 640
    cx, cy = 320.1 * w / 640, 247.6 * h / 480
    u, v = np.meshgrid(np.arange(w), np.arange(h))
    dx, dy = (u - cx) / fx, (v - cy) / fy
    th = np.deg2rad(tilt_deg)
    n = np.array([0.0, np.cos(th), np.sin(th)])
    den = n[1] * dy + n[2]
    with np.errstate(divide="ignore"):
        zf = np.where(den > 1e-6, height / den, np.inf)
    z = np.minimum(zf, wall) if floor else np.full((h, w), wall)
    rng = np.random.default_rng(seed)
    depth = np.clip(np.round(z * 5000 + rng.normal(0, noise, z.shape)), 0, 65535).astype(np.uint16)
    if hole:
        depth[h // 5:h // 5 + 10, w // 3:w // 3 + 40] = 0
    return depth, np.array([fx, fy, cx, cy, 5000.0]), np.array([-n[0], -n[1], -n[2], height])


# ------------------------------------------------------------------------------------------------ CPU
def test_checker_matches_independent_version_on_the_clip_frames():
    intr, frames = fixture()
    for depth, pose, plane, counts in frames:
        r = po.extract_ground_plane(depth, intr, want_normals=True)
        assert r["ok"]
        np.testing.assert_allclose(r["plane"], plane, atol=1e-5)
        assert [r["n_planes"], r["n_pixels"], int(np.isfinite(r["normals"][..., 0]).sum())] == list(counts)


def test_refinement_pass_on_the_clip_frames():
    """segmentAndRefine's second half (PlaneExtractor.cpp:82; PCL's OrganizedMultiPlaneSegmentation::refine, restated: unpinned): the C
    checker's literal two raster passes and the numpy version's per-row scans give the same label image, pixel for pixel; the pass
    grows the floor by 12 % / 13 % on the two frames (0 and 35) and lets segments of 101 .. 199 pixels reach Plane.MinSize; the coefficients
    of a plane do not move.  refine = 0 is the round 2-4 behaviour (the stored counts)."""
    intr, frames = fixture(extra=True)
    for depth, pose, plane, counts, sizes, counts0 in frames:
        r, q = po.extract_planes(depth, intr), np_plane.extract_ground_plane(depth, intr)
        assert r["n_planes"] == q["n_planes"] == counts[0] and list(r["sizes"]) == list(q["sizes"]) == list(sizes)
        assert np.array_equal(r["labels"], q["labels"])
        assert [int((r["labels"] == k).sum()) for k in range(r["n_planes"])] == list(r["sizes"])
        r0 = po.extract_planes(depth, intr, _abi().default_plane_params(refine=0))
        g0 = po.extract_ground_plane(depth, intr, _abi().default_plane_params(refine=0))
        assert [g0["n_planes"], g0["n_pixels"]] == list(counts0) and counts0[1] < counts[1] and counts0[0] <= counts[0]
        # every pixel a segment had, it keeps; the planes that exist without the pass keep their coefficients
        keep = r0["labels"] >= 0
        assert np.all(r["labels"][keep] >= 0)
        for k0 in range(r0["n_planes"]):
            k1 = int(r["labels"][r0["labels"] == k0][0])
            np.testing.assert_array_equal(r["planes"][k1], r0["planes"][k0])
            assert r["sizes"][k1] >= r0["sizes"][k0]


def test_clip_floor_is_the_mocap_floor():
    """the extracted plane, moved to the world with the frame's mocap pose (Tracking.cpp:741-744), is z = 0 within the
    calibration slop of the recording (2.5 degrees, 15 cm)"""
    from oracle import np_fit
    intr, frames = fixture()
    for depth, pose, plane, counts in frames:
        T = np_fit.se3_mat(pose)
        n = T[:3, :3] @ plane[:3]
        d = plane[3] - T[:3, 3] @ n
        assert np.degrees(np.arccos(abs(n[2]))) < 2.5 and abs(d) < 0.15


def test_known_plane_and_failure_cases():
    depth, intr, want = scene()
    r = po.extract_ground_plane(depth, intr)
    assert r["ok"] and r["n_pixels"] > 10000
    np.testing.assert_allclose(r["plane"], want, atol=2e-3)          # depth quantised to 0.2 mm
    q = np_plane.extract_ground_plane(depth, intr)
    # (a noise-free synthetic floor puts whole rows of pixels EXACTLY at the refinement distance of the far wall's strips: the two
    #  versions' planes differ in the 12th digit and split such ties differently -- 30 of 76,800 labels; the clip frames and the noisy
    #  scenes below agree pixel for pixel)
    assert abs(q["n_pixels"] - r["n_pixels"]) <= 0.002 * r["n_pixels"] and q["n_planes"] == r["n_planes"]
    np.testing.assert_allclose(q["plane"], r["plane"], atol=1e-5)
    q0, r0 = np_plane.extract_ground_plane(depth, intr, refine=False), po.extract_ground_plane(depth, intr, _abi().default_plane_params(refine=0))
    assert q0["n_pixels"] == r0["n_pixels"] and q0["n_planes"] == r0["n_planes"] and r0["n_pixels"] < r["n_pixels"]
    np.testing.assert_array_equal(r0["plane"], r["plane"])          # the refinement pass grows the inlier lists, not the coefficients
    # only a wall: planes exist, none passes the wall filter (PlaneExtractor.cpp:139-146) -> false
    depth, intr, _ = scene(floor=False, hole=False)
    r = po.extract_ground_plane(depth, intr)
    assert not r["ok"] and r["n_planes"] == 1 and np.all(r["plane"] == 0)
    # no depth at all
    r = po.extract_ground_plane(np.zeros((60, 80), np.uint16), intr)
    assert not r["ok"] and r["n_planes"] == 0
    # an image smaller than one normal window
    r = po.extract_ground_plane(np.full((8, 8), 5000, np.uint16), intr)
    assert not r["ok"] and r["n_planes"] == 0


def test_noisy_scene_checker_vs_independent_version():
    for seed in range(3):
        depth, intr, want = scene(noise=4.0, seed=seed, tilt_deg=20 + 10 * seed)
        r = po.extract_ground_plane(depth, intr, want_normals=True)
        q = np_plane.extract_ground_plane(depth, intr)
        va, vb = np.isfinite(r["normals"][..., 0]), np.isfinite(q["normals"][..., 0])
        assert np.array_equal(va, vb)
        assert r["ok"] and q["ok"]
        assert abs(r["n_pixels"] - q["n_pixels"]) <= 0.002 * r["n_pixels"]     # float32 ties at the thresholds
        np.testing.assert_allclose(r["plane"], q["plane"], atol=1e-4)
        np.testing.assert_allclose(r["plane"], want, atol=1e-2)


def test_all_planes_in_raster_order_with_labels():
    """extractPlanes / GetCoefficients / GetPoints: every segment of >= min size, ordered by its first pixel, and the label image"""
    intr, frames = fixture()
    depth = frames[1][0]
    r = po.extract_planes(depth, intr)
    q = np_plane.extract_ground_plane(depth, intr)
    assert r["n_planes"] == q["n_planes"] == len(r["planes"]) and r["n_planes"] >= 2
    assert np.array_equal(r["sizes"], q["sizes"]) and np.array_equal(r["labels"], q["labels"])
    # (the checker's moments are fixed point, 2^-30 m^2: a sliver of ~120 pixels a few centimetres across -- the 220-pixel plane of this
    #  frame before it grows -- has a covariance of 1e-4 m^2 and a normal good to ~1e-4; the large planes agree to 1e-5)
    for k in range(r["n_planes"]):
        np.testing.assert_allclose(r["planes"][k], q["planes"][k], atol=1e-5 if r["sizes"][k] >= 1000 else 2e-4)
    assert [int((r["labels"] == k).sum()) for k in range(r["n_planes"])] == list(r["sizes"])
    first = [int(np.flatnonzero(r["labels"].reshape(-1) == k)[0]) for k in range(r["n_planes"])]
    assert first == sorted(first)
    assert np.all(r["planes"][:, 3] >= 0)
    np.testing.assert_allclose(np.linalg.norm(r["planes"][:, :3], axis=1), 1.0, atol=1e-12)
    # the ground plane is the largest of those that pass the wall filter
    g = po.extract_ground_plane(depth, intr)
    k = int(np.argmax(r["sizes"]))
    np.testing.assert_array_equal(g["plane"], r["planes"][k])
    # capacity smaller than the number of planes: the count is still reported, labels keep their indices
    c = po.extract_planes(depth, intr, max_planes=1)
    assert c["n_planes"] == r["n_planes"] and len(c["planes"]) == 1 and np.array_equal(c["labels"], r["labels"])


def test_min_size_and_thresholds_are_honoured():
    from importlib import import_module
    abi = import_module("object-oriented-slam_amd.abi")
    depth, intr, _ = scene(h=120, w=160)
    r = po.extract_ground_plane(depth, intr)
    big = po.extract_ground_plane(depth, intr, abi.default_plane_params(min_size=r["n_pixels"] + 1))
    assert r["ok"] and not big["ok"]
    same = po.extract_ground_plane(depth, intr, abi.default_plane_params(min_size=50))   # setMinInliers(100) still applies
    assert same["n_planes"] == po.extract_ground_plane(depth, intr, abi.default_plane_params(min_size=100))["n_planes"]


def test_pcl_model_test_strict_size_and_curvature():
    """OrganizedMultiPlaneSegmentation::segment keeps a segment as a MODEL only if it has MORE than min_inliers pixels and its surface
    curvature |lambda_min| / trace(cov) is below maximum_curvature_ = 0.001 (restated from PCL 1.8's published source,
    unpinned like the rest of the PCL steps).  On the first clip frame the segmentation chains a 63,078-pixel component through the
    cabinet (curvature 0.17) and a 456-pixel one at 0.002: with the test they are "other" pixels the floor may absorb (floor 183,535 ->
    184,337 inliers), without it (max_curvature = 0, the round-5 behaviour) they are planes.  Both CPU restatements agree either way."""
    abi = _abi()
    intr, frames = fixture()
    depth = frames[0][0]
    on, off = po.extract_planes(depth, intr), po.extract_planes(depth, intr, abi.default_plane_params(max_curvature=0.0))
    q_on, q_off = np_plane.extract_ground_plane(depth, intr), np_plane.extract_ground_plane(depth, intr, max_curvature=0.0)
    assert list(on["sizes"]) == list(q_on["sizes"]) == [184337]
    assert list(off["sizes"]) == list(q_off["sizes"]) and len(off["sizes"]) == 4 and off["sizes"][0] == 183535
    assert np.array_equal(on["labels"], q_on["labels"]) and np.array_equal(off["labels"], q_off["labels"])
    # strict >: a segment of exactly min_inliers pixels is not a model
    d, it, _ = scene(h=120, w=160)
    r = po.extract_planes(d, it, abi.default_plane_params(refine=0))
    n = int(r["sizes"].max())
    assert po.extract_ground_plane(d, it, abi.default_plane_params(refine=0, min_size=1, min_inliers=n - 1))["ok"]
    assert not po.extract_ground_plane(d, it, abi.default_plane_params(refine=0, min_size=1, min_inliers=n))["ok"]
    assert np_plane.extract_ground_plane(d, it, refine=False, min_size=1, min_inliers=n - 1)["ok"]
    assert not np_plane.extract_ground_plane(d, it, refine=False, min_size=1, min_inliers=n)["ok"]


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_gpu_pcl_model_test_matches_checker(ctx):
    abi = _abi()
    intr, frames = fixture()
    for depth in (frames[0][0], frames[1][0]):
        for kw in (dict(), dict(max_curvature=0.0), dict(refine=0), dict(refine=0, max_curvature=0.0), dict(min_inliers=455), dict(min_inliers=456, max_curvature=0.0)):
            p = abi.default_plane_params(**kw)
            g, r = ctx.extract_planes(depth, intr, p), po.extract_planes(depth, intr, p)
            assert g["n_planes"] == r["n_planes"] and np.array_equal(g["sizes"], r["sizes"]) and np.array_equal(g["labels"], r["labels"]), kw
            gg, rr = ctx.extract_ground_plane(depth, intr, p), po.extract_ground_plane(depth, intr, p)
            assert (gg["ok"], gg["n_planes"], gg["n_pixels"]) == (rr["ok"], rr["n_planes"], rr["n_pixels"]), kw


@pytest.mark.gpu
def test_gpu_ground_plane_matches_checker_and_fixture(ctx):
    intr, frames = fixture()
    for depth, pose, plane, counts in frames:
        g = ctx.extract_ground_plane(depth, intr)
        r = po.extract_ground_plane(depth, intr)
        assert g["ok"] and [g["n_planes"], g["n_pixels"]] == [r["n_planes"], r["n_pixels"]] == list(counts[:2])
        np.testing.assert_allclose(g["plane"], r["plane"], atol=1e-9)
        np.testing.assert_allclose(g["plane"], plane, atol=1e-5)


@pytest.mark.gpu
def test_gpu_all_planes_match_checker(ctx):
    intr, frames = fixture()
    cases = [(f[0], intr) for f in frames] + [scene(noise=3.0, seed=5)[:2], scene(h=123, w=211, noise=2.0)[:2],
                                                  scene(h=90, w=1300, noise=2.0, seed=2)[:2]]   # (wider than the refinement kernel's LDS ring: its two-row form)
    for depth, it in cases:
        g, r = ctx.extract_planes(depth, it), po.extract_planes(depth, it)
        assert g["n_planes"] == r["n_planes"] and np.array_equal(g["sizes"], r["sizes"]) and np.array_equal(g["labels"], r["labels"])
        np.testing.assert_allclose(g["planes"], r["planes"], atol=1e-9)
    g = ctx.extract_planes(frames[1][0], intr, max_planes=1)
    assert g["n_planes"] >= 2 and len(g["planes"]) == 1
    g = ctx.extract_planes(np.zeros((60, 80), np.uint16), INTR)
    assert g["n_planes"] == 0 and np.all(g["labels"] == -1)


@pytest.mark.gpu
def test_gpu_ground_plane_synthetic_and_edge_cases(ctx):
    from importlib import import_module
    abi = import_module("object-oriented-slam_amd.abi")
    lib = import_module("object-oriented-slam_amd.lib")
    for kw in (dict(), dict(noise=4.0, seed=1), dict(h=123, w=211, noise=2.0), dict(floor=False, hole=False), dict(tilt_deg=60.0)):
        depth, intr, want = scene(**kw)
        g = ctx.extract_ground_plane(depth, intr)
        r = po.extract_ground_plane(depth, intr)
        assert g["ok"] == r["ok"] and g["n_planes"] == r["n_planes"] and g["n_pixels"] == r["n_pixels"], kw
        np.testing.assert_allclose(g["plane"], r["plane"], atol=1e-9)
    for depth in (np.zeros((60, 80), np.uint16), np.full((8, 8), 5000, np.uint16)):
        g = ctx.extract_ground_plane(depth, INTR)
        assert not g["ok"] and g["n_planes"] == 0 and np.all(g["plane"] == 0)
    # twice the same answer: the extraction is order independent
    depth, intr, _ = scene(noise=5.0, seed=7)
    a, b = ctx.extract_ground_plane(depth, intr), ctx.extract_ground_plane(depth, intr)
    assert np.array_equal(a["plane"], b["plane"]) and a["n_pixels"] == b["n_pixels"]
    with pytest.raises(lib.EslError):
        ctx.extract_ground_plane(depth, intr, abi.default_plane_params(distance_threshold=0.0))


@pytest.mark.gpu
def test_gpu_clip_runs_from_raw_depth_with_estimated_ground(tmp_path):
    """the whole run with nothing given but the clip: a scene with a floor and a box on it; the harness estimates the floor
    on the first frame (Tracking.cpp:498-499) on the GPU exactly as the checker harness does on the CPU"""
    import subprocess
    import harness_util as hu
    d = str(tmp_path / "clip")
    for sub in ("depth", "rgb", "bbox"):
        os.makedirs(os.path.join(d, sub))
    depth, intr, want = scene(h=480, w=640, noise=2.0, seed=3)
    stamps = ["1.0000", "1.1000"]
    with open(os.path.join(d, "groundtruth.txt"), "w") as gt, open(os.path.join(d, "associate.txt"), "w") as asc, \
            open(os.path.join(d, "associateGroundtruth.txt"), "w") as ag:
        for s in stamps:
            open(os.path.join(d, "rgb", s + ".jpg"), "w").close()
            hu.write_png16_fast(os.path.join(d, "depth", s + ".png"), depth)
            gt.write("%s 0 0 0 0 0 0 1\n" % s)
            ag.write("%s00 rgb/%s.jpg %s00 0 0 0 0 0 0 1\n" % (s, s, s))
            asc.write("%s00 rgb/%s.jpg %s00 depth/%s.png\n" % (s, s, s, s))
            open(os.path.join(d, "bbox", s + ".txt"), "w").write("")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "harness")])
    exe_o = hu.build_oracle_harness(tmp_path)
    outs = []
    for exe, tag in ((os.path.join(ROOT, "harness", "esl_rgbd"), "gpu"), (exe_o, "cpu")):
        od = str(tmp_path / tag)
        os.makedirs(od)
        subprocess.check_call([exe, d, od])
        outs.append(np.array(hu.read_table(os.path.join(od, "ground_plane.txt")))[0])
    assert outs[0][0] == 0 and outs[1][0] == 0                        # estimated on the first frame
    np.testing.assert_allclose(outs[0][1:], outs[1][1:], atol=1e-9)
    np.testing.assert_allclose(outs[0][1:], want, atol=1e-2)          # identity pose: world = camera
