#!/usr/bin/env python
"""HBM traffic per dispatch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd sqlite output).

    python scripts/pmc_summary.py <fetch_dir> <write_dir> <out.json> [kernel-substring ...]

Units and corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are KiB per dispatch
summed over the counter's instances; on gfx950 FETCH_SIZE prices the 128-B requests of wide coalesced reads as 64 B, so
it is doubled; WRITE_SIZE is taken as is."""
import glob
import json
import sqlite3
import statistics
import sys


def per_kernel(dirname, counter):
    out = {}
    for f in glob.glob(dirname + "/**/*.db", recursive=True):
        db = sqlite3.connect(f)
        rows = db.execute("select kernel_name, dispatch_id, sum(value), max(grid_size) from counters_collection where counter_name = ? "
                          "group by kernel_name, dispatch_id", (counter,)).fetchall()
        for name, _, val, grid in rows:
            out.setdefault(name.split("(")[0], []).append((val, grid))
    return out


def main():
    fdir, wdir, dst = sys.argv[1:4]
    want = sys.argv[4:]
    fetch, write = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    res = {"units": "KiB per dispatch (median over live dispatches); fetch_bytes_corrected = 2 x FETCH_SIZE x 1024 (gfx950), "
                    "write_bytes = WRITE_SIZE x 1024; traffic = their sum", "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if want and not any(w in k for w in want):
            continue
        fv = [v for v, _ in fetch.get(k, [])]
        wv = [v for v, _ in write.get(k, [])]
        if not fv or not wv:
            continue
        # a device-driven LM run leaves a few no-op dispatches of the same kernels (they exit at the first instruction):
        # keep the dispatches above 20 % of the largest one
        live_f = [v for v in fv if v >= 0.2 * max(fv)] or fv
        live_w = [v for v in wv if v >= 0.2 * max(wv)] or wv
        f_kib, w_kib = statistics.median(live_f), statistics.median(live_w)
        res["kernels"][k] = {"FETCH_SIZE_KiB": f_kib, "WRITE_SIZE_KiB": w_kib, "dispatches": len(fv), "live_dispatches": len(live_f),
                             "fetch_bytes_corrected": 2 * f_kib * 1024, "write_bytes": w_kib * 1024,
                             "traffic_bytes_per_launch": 2 * f_kib * 1024 + w_kib * 1024}
    # the launch the bench line's roofline is quoted on: the rank-K update T -= X^T X of the camera-first elimination = the
    # dispatch of the big update kernel (round 6: k_chol_update_v; before: k_chol_update_lds<256, 128>) with the LARGEST grid (the whole
    # lower triangle of T; the trailing updates of the factorisation that follows cover less and less of it)
    big = [k for k in fetch if ("k_chol_update_v" in k or "k_chol_update_lds<256" in k) and k in write]
    big.sort(key=lambda k: -max(g for _, g in fetch[k]))
    if big:
        k = big[0]
        gmax = max(g for _, g in fetch[k])
        fv = [v for v, g in fetch[k] if g == gmax]; wv = [v for v, g in write[k] if g == gmax]
        f_kib, w_kib = statistics.median(fv), statistics.median(wv)
        res["rank_k_update_launch"] = {"kernel": k, "grid_threads": gmax, "dispatches": len(fv), "fetch_bytes_corrected": 2 * f_kib * 1024,
                                       "write_bytes": w_kib * 1024, "traffic_bytes_per_launch": 2 * f_kib * 1024 + w_kib * 1024}
    res["x_sparse"] = any("k_cf_T_gather" in k or "k_cf_T_sparse" in k for k in fetch)
    json.dump(res, open(dst, "w"), indent=1)
    if "rank_k_update_launch" in res:
        v = res["rank_k_update_launch"]
        print(f"rank-K update launch (grid {v['grid_threads']}): fetch {v['fetch_bytes_corrected']/1e6:.2f} MB write {v['write_bytes']/1e6:.2f} MB, X sparse: {res['x_sparse']}")
    for k, v in res["kernels"].items():
        print(f"{k:60s} fetch {v['fetch_bytes_corrected']/1e6:8.2f} MB  write {v['write_bytes']/1e6:7.2f} MB  ({v['live_dispatches']}/{v['dispatches']} live)")


if __name__ == "__main__":
    main()
