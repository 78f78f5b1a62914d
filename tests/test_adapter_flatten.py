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
#include "../adapter/OptimizerEsl.cpp"
struct V7 { std::array<double,7> a; std::array<double,7> toVector() const { return a; } };
struct Ell { std::array<double,10> a; int miInstanceID; double prob; std::array<double,10> toVector() const { return a; } };
struct Frame { V7 cam_pose_Tcw; int frame_seq_id; std::vector<Ell*> mpLocalObjects; };
struct Obs { Frame* pFrame; std::array<double,4> bbox; double rate; };
int main() {
  std::vector<Frame*> frames;
  for (int i = 0; i < 4; ++i) { Frame* f = new Frame(); f->frame_seq_id = i; f->cam_pose_Tcw.a = {double(i),0,0,0,0,0,1}; frames.push_back(f); }
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
  return 0;
}
'''


def test_adapter_flatten_compiles_and_orders_like_the_reference():
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "drv.cpp")
        open(src, "w").write(DRIVER.replace("../adapter", os.path.join(ROOT, "adapter")))
        exe = os.path.join(td, "drv")
        subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    # 4 cams, 2 ellipsoids (instances 2 then 7: ascending id), 4 bbox edges (instance 7 only), 1 3-D edge, 2 gravity
    assert out[:5] == ["4", "2", "4", "1", "2"]
    assert out[5:7] == ["2", "7"]
    assert out[7] == "1" and out[8] == "1"          # bbox edges hang on vertex 1 (= instance 7); 3-D edge from frame 1
    assert float(out[9]) == 8000.0                   # Scale * prob
    assert float(out[10]) == 10000.0                 # GravityPrior.Scale^2
    assert float(out[11]) == 11.0
