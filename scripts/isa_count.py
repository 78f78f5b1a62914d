#!/usr/bin/env python
"""Static instruction mix of the gfx950 kernels (device-only compile to assembly, then count by class).

    python scripts/isa_count.py [tag]      # writes profiles/<tag>_isa_counts.json + .md (tag defaults to r2)

Classes: f64 = v_*_f64 (FP64 VALU, 4 cycles per wave64 on a 16-lane SIMD), slow = v_div_*/v_rcp/v_sqrt/v_rsq f64,
valu_other = every other v_* (integer, moves, compares, selects), salu = s_*, lds = ds_* (the wave reduce-scatter's
ds_bpermute / ds_swizzle traffic), mem = global/scratch/flat.  Static counts: both sides of a branch are counted."""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "object-oriented-slam_amd", "csrc")
OUT = os.path.join(ROOT, "profiles")


def assemble(src, extra=()):
    s = os.path.join("/tmp", os.path.basename(src) + ".s")
    cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-std=c++17", "-fPIC", "--cuda-device-only", "-S",
           "-o", s, os.path.join(CSRC, src), *extra]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return s


def count(path):
    fn, stats = None, {}
    for line in open(path):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            fn = m.group(1); stats[fn] = collections.Counter(); continue
        if fn is None:
            continue
        t = line.strip().split()
        if not t or t[0].startswith((".", ";")) or t[0].endswith(":"):
            m2 = re.search(r"; (ScratchSize|NumVgprs|Occupancy|NumAgprs): (\d+)", line)
            if m2:
                stats[fn][m2.group(1)] = int(m2.group(2))
            continue
        op, c = t[0], stats[fn]
        c["total"] += 1
        if op.endswith("_f64") or "_f64_" in op:
            c["f64"] += 1
            if op.startswith(("v_div", "v_rcp", "v_sqrt", "v_rsq")):
                c["slow"] += 1
        elif op.startswith("v_"):
            c["valu_other"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "scratch_", "flat_")):
            c["mem"] += 1
    out = {}
    for k, v in stats.items():
        if v["total"] < 50:
            continue
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void ", "").replace("esl::", "")
        out[name] = dict(total=v["total"], f64=v["f64"], slow=v["slow"], valu_other=v["valu_other"], salu=v["salu"], lds=v["lds"],
                         mem=v["mem"], vgpr=v["NumVgprs"], agpr=v["NumAgprs"], scratch_bytes=v["ScratchSize"], occupancy=v["Occupancy"])
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
    res = {}
    res.update(count(assemble("esl_capi.hip", ["-DESL_ISA_PROBE"])))
    res.update(count(assemble("esl_slam.hip")))
    res.update(count(assemble("esl_fit.hip", ["-ffp-contract=off"])))
    res.update(count(assemble("esl_init.hip")))
    res.update(count(assemble("esl_plane.hip", ["-ffp-contract=off"])))
    os.makedirs(OUT, exist_ok=True)
    json.dump(res, open(os.path.join(OUT, tag + "_isa_counts.json"), "w"), indent=1, sort_keys=True)
    with open(os.path.join(OUT, tag + "_isa_counts.md"), "w") as f:
        f.write("Static gfx950 instruction mix per kernel (scripts/isa_count.py; both sides of every branch counted).\n\n")
        f.write("| kernel | total | f64 VALU | of which div/rcp/sqrt | other VALU | SALU | LDS | mem | VGPR | AGPR | scratch B | waves/SIMD |\n")
        f.write("|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for k in sorted(res):
            v = res[k]
            f.write(f"| `{k}` | {v['total']} | {v['f64']} | {v['slow']} | {v['valu_other']} | {v['salu']} | {v['lds']} | {v['mem']} | "
                    f"{v['vgpr']} | {v['agpr']} | {v['scratch_bytes']} | {v['occupancy']} |\n")
    print(open(os.path.join(OUT, tag + "_isa_counts.md")).read())


if __name__ == "__main__":
    sys.exit(main())
