"""One rank of the two-GPU RCCL run of tests/test_gpu_sharded.py::test_rccl_two_ranks_on_two_gpus (one process per GPU; the
128-byte communicator id travels through a file).  argv: rank, id file, output npz, mode: 0 mapping (sharded), 1 SLAM (sharded),
2 SLAM with the whole graph on both ranks and the communicator in replicated-graph mode (the ranks divide the dense solve)."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("object-oriented-slam_amd")
rank, id_file, out_file, slam, repl = int(sys.argv[1]), sys.argv[2], sys.argv[3], sys.argv[4] in ("1", "2"), sys.argv[4] == "2"
g, c, o, _ = pkg.synth.make_graph(120 if slam else 30, 10 if slam else 8, 1500 if slam else 300, seed=21, slam=slam)
part = pkg.lib.partition_objects(g, 2)
idx = np.arange(g.n_objs) if repl else np.nonzero(part == rank)[0]
ctx = pkg.Context(rank)                       # one GPU per rank
if rank == 0:
    uid = bytes(pkg.lib.comm_unique_id())
    with open(id_file + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(id_file + ".tmp", id_file)
else:
    t0 = time.time()
    while not os.path.exists(id_file):
        if time.time() - t0 > 60:
            sys.exit("no communicator id after 60 s")
        time.sleep(0.05)
    uid = open(id_file, "rb").read()
ctx.comm_init(2, rank, uid)
if repl:
    ctx.comm_set_replicated(True)     # the mode first, then the graph (ABI 4)
ctx.upload_graph(g.subset_objects(idx)); ctx.upload_states(c, o[idx])
rep = ctx.optimize_resident(pkg.default_lm_params(jacobian_mode=1))
cc, oo = ctx.download_states()
ctx.comm_destroy(); ctx.close()
np.savez(out_file, idx=idx, cams=cc, objs=oo, iterations=rep["iterations"], trace_chi2=np.array(rep["trace_chi2"]),
         trace_trials=np.array(rep["trace_trials"]), stop_reason=rep["stop_reason"])
