#!/usr/bin/env python
"""Where does a kernel touch its scratch?  Per basic block of the gfx950 ISA: loop depth, instruction count, MFMAs, scratch
loads / stores -- the table that shows whether spilled registers are reloaded inside a hot loop or in a prologue.

    python scripts/isa_scratch_blocks.py esl_slam.hip k_chol_persist > profiles/r5_persist_scratch_isa.md

Device-only compile to assembly with the Makefile's flags (hipcc cross-compiles gfx950 without a GPU)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "object-oriented-slam_amd", "csrc")


def main():
    src, kern = sys.argv[1], sys.argv[2]
    asm = os.path.join("/tmp", src + ".s")
    subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "--cuda-device-only", "-S", "-o", asm,
                    os.path.join(CSRC, src)], check=True, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    start = [i for i, ln in enumerate(lines) if re.match(r"^_Z\S*" + kern + r"\S*:", ln)][0]
    end = [i for i, ln in enumerate(lines) if i > start and ".end_amdhsa_kernel" in ln][0]
    info = {}
    for ln in lines[end:end + 80]:
        m = re.search(r"; (NumVgprs|NumAgprs|ScratchSize|Occupancy): (\d+)", ln)
        if m:
            info.setdefault(m.group(1), int(m.group(2)))
    blocks, cur = [], dict(name="entry", depth=0, mfma=0, ld=0, st=0, n=0)
    for ln in lines[start:end]:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", ln)
        if m:
            blocks.append(cur)
            d = re.search(r"Depth=(\d+)", ln)
            cur = dict(name=m.group(1), depth=int(d.group(1)) if d else 0, mfma=0, ld=0, st=0, n=0)
            continue
        t = ln.strip()
        if not t or t[0] in ";.":
            d = re.search(r"Depth=(\d+)", t)
            if d and cur["n"] == 0:
                cur["depth"] = max(cur["depth"], int(d.group(1)))
            continue
        cur["n"] += 1
        cur["mfma"] += "v_mfma" in t
        cur["ld"] += t.startswith("scratch_load")
        cur["st"] += t.startswith("scratch_store")
    blocks.append(cur)
    print(f"# {kern}: where its scratch is touched (gfx950 ISA, hipcc -O3; scripts/isa_scratch_blocks.py {src} {kern})\n")
    print(f"Kernel info: NumVgprs {info.get('NumVgprs')}, NumAgprs {info.get('NumAgprs')}, ScratchSize {info.get('ScratchSize')} B/lane, occupancy "
          f"{info.get('Occupancy')} waves / SIMD.\n")
    print(f"{len(blocks)} basic blocks, {sum(b['mfma'] for b in blocks)} MFMA instructions, {sum(b['ld'] + b['st'] for b in blocks)} scratch instructions.  "
          "Blocks with >= 16 MFMAs or any scratch access:\n")
    print("| block | loop depth | instructions | MFMAs | scratch loads | scratch stores |")
    print("|---|---|---|---|---|---|")
    for b in blocks:
        if b["mfma"] >= 16 or b["ld"] or b["st"]:
            print(f"| `{b['name']}` | {b['depth']} | {b['n']} | {b['mfma']} | {b['ld']} | {b['st']} |")
    hot = [b for b in blocks if b["mfma"] >= 16]
    bad = [b for b in hot if b["ld"] + b["st"]]
    print(f"\nBlocks with >= 16 MFMAs: {len(hot)}; of those with a scratch access: {len(bad)}"
          + (" (" + ", ".join(f"{b['name']}: {b['ld'] + b['st']} in {b['n']} instructions" for b in bad) + ")" if bad else "") + ".")


if __name__ == "__main__":
    main()
