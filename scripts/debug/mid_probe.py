import importlib, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("object-oriented-slam_amd")
from oracle import np_oracle as npo
def cam_err(a, b):
    return max(np.linalg.norm(npo.se3_log(npo.T_inv(npo.T_from7(x)) @ npo.T_from7(y))) for x, y in zip(a, b))
def obj_rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())
g, c, o, _ = pkg.synth.make_graph(n_cams=2000, n_objs=300, n_bbox_target=16000, seed=41, slam=True)
refs = {"1e-6": np.load(os.path.join(ROOT, "tests/golden/mid_slam_run.npz")), "1e-5": np.load(os.path.join(ROOT, "scripts/debug/tmp_mid/mid_1e-05.npz")),
        "1e-4": np.load(os.path.join(ROOT, "scripts/debug/tmp_mid/mid_0.0001.npz"))}
os.environ["ESL_CF_SPARSE"] = "1"
cx = pkg.Context(0)
runs = {}
for tag, jac, d, solver in (("num1e-5", 0, 1e-5, 2), ("num1e-6", 0, 1e-6, 2), ("num1e-4", 0, 1e-4, 2), ("analytic", 1, 1e-6, 2), ("num1e-5/S", 0, 1e-5, 1)):
    cg, og, rg = cx.optimize(g, c, o, pkg.default_lm_params(jacobian_mode=jac, numeric_delta=d, linear_solver=solver))
    runs[tag] = (cg, og, rg)
    print(tag, rg["iterations"], rg["trace_trials"], " ".join("| vs checker(%s): cams %.2e objs %.2e chi2 %.1e" % (k, cam_err(cg, r["cams"]), obj_rel(og, r["objs"]),
          abs(rg["chi2_final"] / float(r["chi2_final"] if "chi2_final" in r else r["chi2"]) - 1)) for k, r in refs.items()), flush=True)
for a in runs:
    print(a, "vs analytic: cams %.2e objs %.2e" % (cam_err(runs[a][0], runs["analytic"][0]), obj_rel(runs[a][1], runs["analytic"][1])))
