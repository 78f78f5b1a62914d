"""BASELINE.json configs[0] — the TUM fr3_cabinet demo clip END TO END: dataset files -> Tracking-side
gating -> single-frame fit of every detection -> object initialisation -> global optimisation after every frame ->
objects.txt.  The logic above the C-ABI is harness/esl_harness.hpp (build-owned restatement of src/core/Tracking.cpp:172-638,
src/tum_rgbd/io.cpp, src/utils/dataprocess_utils.cpp); the product driver harness/esl_rgbd.cpp runs it on libesl_hip.so, the
test driver tests/harness_oracle_main.cpp on the CPU checker.

CPU: the file formats (16-bit PNG with every filter type, stamp association, bbox rows, objects.txt precision), the gating
rules on hand-made clips, and the whole clip (rebuilt from tests/golden/cabinet_fit_np.npz, byte-identical to the
reference's data inside every bounding box) on the checker against tests/golden/cabinet_run.npz.
GPU: the same clip through esl_rgbd: same graph after every frame (vertex / edge counts, dropped edges), same final object.
"""
import os
import subprocess

import numpy as np
import pytest

import harness_util as hu

ROOT = hu.ROOT
GOLD_FIT = os.path.join(ROOT, "tests", "golden", "cabinet_fit_np.npz")
GOLD_RUN = os.path.join(ROOT, "tests", "golden", "cabinet_run.npz")


@pytest.fixture(scope="module")
def oracle_exe(tmp_path_factory):
    return hu.build_oracle_harness(tmp_path_factory.mktemp("harness"))


@pytest.fixture(scope="module")
def clip_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("cabinet")
    hu.write_cabinet_clip(np.load(GOLD_FIT), str(d))
    return str(d)


MOCAP_FLOOR = ["--ground", "0", "0", "1", "0"]   # the clips of these tests carry the bbox crops only: the floor is given (mocap z = 0)


def run(exe, clip, out, *args, ground=MOCAP_FLOOR):
    os.makedirs(out, exist_ok=True)
    txt = subprocess.check_output([exe, clip, out] + list(ground) + list(args), stderr=subprocess.STDOUT).decode()
    return (txt, np.array(hu.read_table(os.path.join(out, "objects.txt"))), np.array(hu.read_table(os.path.join(out, "graph_log.txt"))),
            open(os.path.join(out, "object_history.txt")).read())


# ------------------------------------------------------------------------------------------------ CPU
def test_png16_reader_handles_every_filter_type(tmp_path):
    """harness read_png16 against a writer that uses filter types 0-4 row by row (+ PIL's adaptive filters when available)"""
    src = os.path.join(str(tmp_path), "png_probe.cpp")
    open(src, "w").write('#include "%s/harness/esl_harness.hpp"\nint main(int c, char** v) { std::vector<uint16_t> im; int w, h;'
                         ' if (!esl_harness::read_png16(v[1], im, w, h)) return 1; std::printf("%%d %%d", w, h); unsigned long long s = 0;'
                         ' for (size_t i = 0; i < im.size(); ++i) s += (unsigned long long)im[i] * (i %% 251 + 1); std::printf(" %%llu %%u %%u\\n", s, im[0], im.back()); }\n' % ROOT)
    exe = os.path.join(str(tmp_path), "png_probe")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), src, "-lz", "-o", exe])
    rng = np.random.default_rng(0)
    img = (rng.integers(0, 65536, size=(37, 53)) * (rng.random((37, 53)) < 0.7)).astype(np.uint16)
    img[5:20, 10:40] = np.arange(30, dtype=np.uint16)[None, :] * 300 + 1000      # smooth region: sub / up / paeth all matter
    want = "%d %d %d %d %d" % (53, 37, int((img.reshape(-1).astype(np.uint64) * (np.arange(img.size) % 251 + 1).astype(np.uint64)).sum()), img[0, 0], img[-1, -1])
    p = os.path.join(str(tmp_path), "a.png")
    hu.write_png16(p, img, filters=[0, 1, 2, 3, 4])
    assert subprocess.check_output([exe, p]).decode().strip() == want
    hu.write_png16_fast(p, img)
    assert subprocess.check_output([exe, p]).decode().strip() == want
    try:
        from PIL import Image
        Image.fromarray(img).save(p)
        assert subprocess.check_output([exe, p]).decode().strip() == want
    except ImportError:
        pass
    open(p, "wb").write(b"not a png")
    assert subprocess.call([exe, p]) == 1


def test_gating_rules_on_a_hand_made_clip(tmp_path, oracle_exe):
    """3 frames, no usable depth (fits fail -> no 3-D initialisation): border boxes are ignored, instance -1 is ignored, a frame
    without a pose or a depth association is skipped, fewer than 15 observations never initialise an object (no optimisation)."""
    d = str(tmp_path / "clip")
    for s in ("rgb", "depth", "bbox"):
        os.makedirs(os.path.join(d, s))
    stamps = ["10.1000", "10.2000", "10.3000", "10.4000"]
    with open(os.path.join(d, "groundtruth.txt"), "w") as gt, open(os.path.join(d, "associate.txt"), "w") as asc, \
            open(os.path.join(d, "associateGroundtruth.txt"), "w") as ag:
        for k, s in enumerate(stamps):
            open(os.path.join(d, "rgb", s + ".jpg"), "w").close()
            hu.write_png16_fast(os.path.join(d, "depth", s + ".png"), np.zeros((480, 640), np.uint16))
            if k != 2:   # frame 2 has no ground-truth association -> skipped
                gt.write("%s %g 0 0 0 0 0 1\n" % (s, 0.1 * k))
                ag.write("%s00 rgb/%s.jpg %s00 %g 0 0 0 0 0 1\n" % (s, s, s, 0.1 * k))
            asc.write("%s00 rgb/%s.jpg %s00 depth/%s.png\n" % (s, s, s, s))
    rows = {"10.1000": ["0 100 100 300 300 28 0.9 0", "1 5 100 300 300 28 0.9 0", "2 100 100 300 300 28 0.9 -1"],
            "10.2000": ["0 100,100\t300 300 28 0.9 0"], "10.3000": ["0 100 100 300 300 28 0.9 0"], "10.4000": []}
    for s, r in rows.items():
        open(os.path.join(d, "bbox", s + ".txt"), "w").write("".join(x + "\n" for x in r))
    txt, objs, log, hist = run(oracle_exe, d, str(tmp_path / "out"))
    assert "frames 4 (valid 3)" in txt
    assert "fits 0 / 2 ok" in txt                 # attempted: frame 0 one non-border box with a valid instance, frame 1 one; none succeeds
    assert objs.size == 0 and log.size == 0 and hist.split() == ["0"]


def test_cabinet_clip_on_the_checker_matches_golden(oracle_exe, clip_dir, tmp_path):
    R = np.load(GOLD_RUN)
    for tag, args in (("ref", []), ("tight", ["--delta", "1e-6", "--sym-iters", "0"]), ("slamref", ["--slam-mode"]),
                      ("slamtight", ["--slam-mode", "--delta", "1e-6", "--sym-iters", "0"])):
        txt, objs, log, hist = run(oracle_exe, clip_dir, str(tmp_path / tag), *args)
        if "slam" in tag:   # the reference's bSLAM_mode branch (Optimizer.cpp:126-158) through the adapter's flattening
            cams = np.array(hu.read_table(os.path.join(str(tmp_path / tag), "cameras_slam.txt")))
            np.testing.assert_allclose(cams, R[tag + "_cameras"], rtol=0, atol=1e-9)
            assert cams.shape == (58, 7) and np.abs(objs - R[tag.replace("slam", "") + "_objects"]).max() > 1e-3   # a different optimum than mapping
        assert "frames 58 (valid 58), fits 49 / 49 ok, objects 1, optimisations 58" in txt
        np.testing.assert_allclose(objs, R[tag + "_objects"], rtol=0, atol=1e-9)
        np.testing.assert_array_equal(log[:, :8], R[tag + "_graph_log"][:, :8])                  # the graph after every frame
        np.testing.assert_allclose(log[:, 9:], R[tag + "_graph_log"][:, 9:], rtol=1e-7)
    # shape of the run: the object appears with the first successful fit (3-D initialisation, Tracking.cpp:606-627), 2-D edges
    # join once the instance has more than two observations (Optimizer.cpp:201), one gravity prior, a camera per frame
    g = R["ref_graph_log"]
    assert g[0].tolist()[:8] == [0, 1, 2, 0, 0, 0, 1, 1] and g[2, 3] == 3 and g[-1, 2] == 59 and g[-1, 3] == 49 and g[-1, 6] == 49
    assert (g[:, 5] == 0).all()
    assert str(R["ref_history"]).splitlines()[1] == "0 49"
    # objects.txt: instance + 10-vector, 12 significant digits (System.cpp:75-91, dataprocess_utils.cpp:113)
    line = open(str(tmp_path / "ref" / "objects.txt")).read().split()
    assert len(line) == 11 and line[0] == "0" and all(len(t.replace("-", "").replace(".", "").lstrip("0")) <= 12 or "e" in t for t in line[1:])


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_cabinet_clip_in_slam_mode_on_gpu(oracle_exe, clip_dir, tmp_path):
    """The reference's SLAM branch (frame 0 fixed, odometry edges, Optimizer.cpp:126-158) reached through the class surface:
    harness -> esl_adapter::Flatten(slam_mode) -> esl_optimize -> Schur complement + dense Cholesky on the GPU, against the
    checker harness, which solves the same graphs with the faithful dense pivoted LDLT of the whole free system."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "harness")])
    exe = os.path.join(ROOT, "harness", "esl_rgbd")
    R = np.load(GOLD_RUN)
    targs = ["--slam-mode", "--delta", "1e-6", "--sym-iters", "0"]
    txt, o_gpu, g_gpu, _ = run(exe, clip_dir, str(tmp_path / "g_slam"), *targs)
    assert "frames 58 (valid 58), fits 49 / 49 ok, objects 1, optimisations 58" in txt
    ref_log = R["slamtight_graph_log"]
    np.testing.assert_array_equal(g_gpu[:, :8], ref_log[:, :8])                        # the graph after every frame
    # ... and the LM iteration counts, except where the optimum is an exact zero of the objective (frame 0: one 3-D edge, chi2
    # 3e-4 -> 1e-147; how many iterations round-off keeps "improving" such a run is not a property of the algorithm)
    live = ref_log[:, 10] > 1e-100
    np.testing.assert_array_equal(g_gpu[live, 8], ref_log[live, 8])
    c_gpu = np.array(hu.read_table(os.path.join(str(tmp_path / "g_slam"), "cameras_slam.txt")))
    d_obj = np.abs(o_gpu - R["slamtight_objects"]).max()
    d_cam = np.abs(c_gpu - R["slamtight_cameras"]).max()
    d_chi = (np.abs(g_gpu[:, 9:] - ref_log[:, 9:]) / np.maximum(np.abs(ref_log[:, 9:]), 1e-9)).max()
    print("cabinet clip, SLAM mode, GPU vs checker (delta 1e-6): object %.2e, cameras %.2e, chi2 rel %.2e" % (d_obj, d_cam, d_chi))
    assert d_obj < 1e-6 and d_cam < 1e-6 and d_chi < 1e-6
    # the product default (analytic Jacobians) against the checker at the reference's delta = 1e-9: north-star tolerance
    _, o_an, g_an, _ = run(exe, clip_dir, str(tmp_path / "g_slam_an"), "--slam-mode", "--jacobian", "analytic")
    np.testing.assert_array_equal(g_an[:, :8], R["slamref_graph_log"][:, :8])
    rel = np.linalg.norm(o_an[0, 1:] - R["slamref_objects"][0, 1:]) / np.linalg.norm(R["slamref_objects"][0, 1:])
    print("cabinet clip, SLAM mode, GPU analytic vs checker (reference settings): relative difference of the final ellipsoid %.2e" % rel)
    assert rel < 1e-4


@pytest.mark.gpu
def test_cabinet_clip_with_automatic_association_on_gpu(clip_dir, tmp_path):
    """f-4 on the product: DataAssociationSolver (src/core/DataAssociation.cpp:16-135) driven by the ellipsoids the GPU fits, the
    instance column of the clip ignored -- one object throughout, the run of the golden fixture"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "harness")])
    exe = os.path.join(ROOT, "harness", "esl_rgbd")
    R = np.load(GOLD_RUN)
    txt, objs, log, _ = run(exe, clip_dir, str(tmp_path / "auto_gpu"), "--auto-association", "--delta", "1e-6", "--sym-iters", "0")
    assert "objects 1" in txt and len(log) == 58
    np.testing.assert_array_equal(log[:, :8], R["tight_graph_log"][:, :8])
    np.testing.assert_allclose(objs, R["tight_objects"], rtol=0, atol=1e-5)


@pytest.mark.gpu
def test_cabinet_clip_end_to_end_on_gpu(oracle_exe, clip_dir, tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "harness")])
    exe = os.path.join(ROOT, "harness", "esl_rgbd")
    # (i) noise-free settings on both sides (numeric Jacobians at delta = 1e-6, symmetry hypotheses scored without their LM)
    targs = ["--delta", "1e-6", "--sym-iters", "0"]
    _, o_ref, g_ref, _ = run(oracle_exe, clip_dir, str(tmp_path / "o_tight"), *targs)
    txt, o_gpu, g_gpu, h_gpu = run(exe, clip_dir, str(tmp_path / "g_tight"), *targs)
    assert "frames 58 (valid 58), fits 49 / 49 ok, objects 1, optimisations 58" in txt
    np.testing.assert_array_equal(g_gpu[:, :8], g_ref[:, :8])
    np.testing.assert_allclose(g_gpu[:, 9:], g_ref[:, 9:], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(o_gpu, o_ref, rtol=0, atol=1e-5)
    assert h_gpu.splitlines()[1] == "0 49"
    # (ii) the reference's settings (delta = 1e-9, 5-iteration symmetry LM) and the product default (analytic Jacobians)
    _, o_ref9, g_ref9, _ = run(oracle_exe, clip_dir, str(tmp_path / "o_ref"))
    for extra in ([], ["--jacobian", "analytic"]):
        _, o_g, g_g, _ = run(exe, clip_dir, str(tmp_path / ("g_ref" + "_".join(extra))), *extra)
        np.testing.assert_array_equal(g_g[:, :8], g_ref9[:, :8])
        rel = np.linalg.norm(o_g[0, 1:] - o_ref9[0, 1:]) / np.linalg.norm(o_ref9[0, 1:])
        print("cabinet clip, GPU %s vs checker (reference settings): relative difference of the final ellipsoid %.2e" % (extra or ["numeric"], rel))
        assert rel < 1e-4


def test_data_association_solver(tmp_path):
    """DataAssociationSolver (src/core/DataAssociation.cpp:16-135) on hand-made ellipsoids: nearest centre below 1 m wins and is
    blocked for later rows, anything else opens a new instance, rows without a single-frame ellipsoid get -1."""
    src = os.path.join(str(tmp_path), "da_probe.cpp")
    open(src, "w").write('''#include "%s/harness/esl_harness.hpp"
using namespace esl_harness;
int main() {
  DataAssociationSolver da;
  Frame f; f.cam_pose_Twc.v = {1, 0, 0, 0, 0, 0, 1};                 // camera at x = 1, no rotation
  std::map<int, Ell*> map;
  auto ell = [](double x, double y, double z) { Ell* e = new Ell(); e->v = {x, y, z, 0, 0, 0, 1, .1, .1, .1}; return e; };
  f.mmObservations.resize(4);
  f.mpLocalObjects = {ell(0, 0, 2), nullptr, ell(0.2, 0, 2), ell(5, 5, 5)};   // camera frame
  std::vector<int> a = da.Solve(&f, map);                            // empty map: every valid row opens an instance
  std::printf("%%d %%d %%d %%d\\n", a[0], a[1], a[2], a[3]);
  map[0] = ell(1.05, 0, 2); map[1] = ell(1.2, 0, 2.1); map[2] = ell(9, 9, 9);
  a = da.Solve(&f, map);                                             // row 0 -> column 0 (0.05 m), row 2 -> column 1, row 3 -> new (3)
  std::printf("%%d %%d %%d %%d\\n", a[0], a[1], a[2], a[3]);
  map.erase(1);
  a = da.Solve(&f, map);                                             // row 0 takes column 0; row 2 finds it blocked -> new instance
  std::printf("%%d %%d %%d %%d\\n", a[0], a[1], a[2], a[3]);
}
''' % ROOT)
    exe = os.path.join(str(tmp_path), "da_probe")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), src, "-lz", "-o", exe])
    out = subprocess.check_output([exe]).decode().split("\n")
    assert out[0] == "0 -1 1 2" and out[1] == "0 -1 1 3" and out[2] == "0 -1 4 5"


def test_cabinet_clip_with_automatic_association(oracle_exe, clip_dir, tmp_path):
    """the clip without its instance column: one object throughout (every single-frame ellipsoid lands within 1 m of it)"""
    txt, objs, log, hist = run(oracle_exe, clip_dir, str(tmp_path / "auto"), "--auto-association")
    assert "objects 1" in txt and len(log) == 58
    R = np.load(GOLD_RUN)
    np.testing.assert_allclose(objs, R["ref_objects"], rtol=0, atol=1e-9)
