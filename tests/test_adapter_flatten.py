"""Compile-check the reference-side adapter's flattening logic (adapter/OptimizerEsl.cpp) with
minimal stand-in types and run it against a hand-built scene: vertex order, the >2-observation
rule, un-initialised instances and null local objects must behave as Optimizer.cpp:166-279."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
#include <array>
#include <cstdio>
#include <map>
#include <vector>
#include "../harness/esl_harness.hpp"   // includes ../adapter/OptimizerEsl.cpp; only its SE3 helpers are used here
struct V7 {
  std::array<double,7> a;
  std::array<double,7> toVector() const { return a; }
  V7 inverse() const { return V7{esl_harness::se3_inverse(a)}; }
  V7 operator*(const V7& b) const { return V7{esl_harness::se3_mul(a, b.a)}; }
};
struct Ell { std::array<double,10> a; int miInstanceID; double prob; std::array<double,10> toVector() const { return a; } };
struct Frame { V7 cam_pose_Tcw, cam_pose_Twc; int frame_seq_id; std::vector<Ell*> mpLocalObjects; };
struct Obs { Frame* pFrame; std::array<double,4> bbox; double rate; };
int main() {
  std::vector<Frame*> frames;
  for (int i = 0; i < 4; ++i) {
    Frame* f = new Frame(); f->frame_seq_id = i;
    const double h = 0.15 * i;   // Twc: rotation about z by 0.3 i rad, position (i, 0.5 i, 0)
    f->cam_pose_Twc.a = {double(i), 0.5 * i, 0, 0, 0, std::sin(h), std::cos(h)};
    f->cam_pose_Tcw = f->cam_pose_Twc.inverse();
    frames.push_back(f);
  }
  Ell e7{{1,2,3,0,0,0,1,.1,.2,.3}, 7, 1.0}, e2{{4,5,6,0,0,0,1,.4,.5,.6}, 2, 1.0};
  std::map<int, Ell*> ells{{7,&e7},{2,&e2}};
  std::map<int, std::vector<Obs*>> obs;
  for (int i = 0; i < 4; ++i) obs[7].push_back(new Obs{frames[i], {10.+i,20,30,40}, 0.5});
  for (int i = 0; i < 2; ++i) obs[2].push_back(new Obs{frames[i], {1,2,3,4}, 0.9});     // only 2 obs -> no bbox edges
  obs[9].push_back(new Obs{frames[0], {1,2,3,4}, 0.9});                                  // instance 9 not in the map
  Ell l7{{0,0,1,0,0,0,1,.1,.1,.1}, 7, 0.8}, l9{{0,0,1,0,0,0,1,.1,.1,.1}, 9, 0.8};
  frames[1]->mpLocalObjects = {nullptr, &l7, &l9};
  esl_adapter::FlatGraph f = esl_adapter::Flatten(frames, ells, obs, 10000.0, true);
  const double K[4] = {1,2,3,4}, ground[4] = {0,0,1,0};
  esl_graph g = esl_adapter::MakeGraph(f, K, ground, 100.0);
  std::printf("%d %d %d %d %d ", g.n_cams, g.n_objs, g.n_bbox, g.n_e3d, g.n_grav);
  std::printf("%d %d ", f.instance_of_obj[0], f.instance_of_obj[1]);
  std::printf("%d %d %g %g %g\n", g.bbox_obj[0], g.e3d_cam[0], g.e3d_weight[0], g.grav_weight, g.bbox_meas[4]);
  std::printf("%d %d %d %d\n", g.cam_fixed == nullptr, g.n_odom, g.check_visibility, g.image_rows);
  // SLAM branch (Optimizer.cpp:126-158) + the visibility arguments
  esl_adapter::Options opt; opt.slam_mode = true; opt.check_visibility = true; opt.rows = 480; opt.cols = 640;
  esl_adapter::FlatGraph fs = esl_adapter::Flatten(frames, ells, obs, 10000.0, true, true);
  esl_graph gs = esl_adapter::MakeGraph(fs, K, ground, 100.0, opt);
  std::printf("%d %d %d %d %d %d %d %d %d %d\n", gs.n_odom, gs.cam_fixed[0], gs.cam_fixed[1], gs.cam_fixed[3], gs.odom_i[0], gs.odom_j[0],
              gs.odom_i[2], gs.odom_j[2], gs.odom_info == nullptr, gs.n_bbox);
  std::printf("%d %d %d\n", gs.check_visibility, gs.image_rows, gs.image_cols);
  for (int e = 0; e < gs.n_odom; ++e) { for (int k = 0; k < 7; ++k) std::printf("%.17g ", gs.odom_meas[7 * e + k]); std::printf("\n"); }
  for (int i = 0; i < 4; ++i) { for (int k = 0; k < 7; ++k) std::printf("%.17g ", gs.cam_fixed ? fs.cams[7 * i + k] : 0.0); std::printf("\n"); }
  return 0;
}
'''


def test_adapter_flatten_compiles_and_orders_like_the_reference():
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "drv.cpp")
        open(src, "w").write(DRIVER.replace("../harness", os.path.join(ROOT, "harness")))
        exe = os.path.join(td, "drv")
        subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), src, "-lz", "-o", exe])
        lines = subprocess.check_output([exe]).decode().splitlines()
        out = lines[0].split()
    # 4 cams, 2 ellipsoids (instances 2 then 7: ascending id), 4 bbox edges (instance 7 only), 1 3-D edge, 2 gravity
    assert out[:5] == ["4", "2", "4", "1", "2"]
    assert out[5:7] == ["2", "7"]
    assert out[7] == "1" and out[8] == "1"          # bbox edges hang on vertex 1 (= instance 7); 3-D edge from frame 1
    assert float(out[9]) == 8000.0                   # Scale * prob
    assert float(out[10]) == 10000.0                 # GravityPrior.Scale^2
    assert float(out[11]) == 11.0
    # mapping mode (the shipped setting): no camera flags, no odometry edges, visibility test off
    assert lines[1].split() == ["1", "0", "0", "0"]
    # SLAM branch (Optimizer.cpp:126-158): frame 0 fixed, one odometry edge per consecutive pair with vertices (i - 1, i), identity
    # information; the ellipsoid / bbox part of the graph is unchanged; check_visibility, rows, cols reach the esl_graph
    assert lines[2].split() == ["3", "1", "0", "0", "0", "1", "2", "3", "1", "4"]
    assert lines[3].split() == ["1", "480", "640"]
    import numpy as np
    from oracle import pyoracle as po
    Z = np.array([[float(v) for v in l.split()] for l in lines[4:7]])
    Tcw = np.array([[float(v) for v in l.split()] for l in lines[7:11]])
    for i in range(1, 4):                      # measurement = Tcw_i * Tcw_{i-1}^-1 of the input poses (Optimizer.cpp:143-146)
        want = po.se3_mul(Tcw[i], po.se3_inv(Tcw[i - 1]))
        np.testing.assert_allclose(Z[i - 1], want, rtol=0, atol=1e-15)
        # and with the input poses the odometry residual log(Z * T_{i-1} * T_i^-1) vanishes (types_six_dof_expmap.h:85-99)
        assert np.abs(po.res_odom(Tcw[i - 1], Tcw[i], Z[i - 1])).max() < 1e-15
