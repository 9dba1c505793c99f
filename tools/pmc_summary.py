#!/usr/bin/env python3
"""Builds profiles/rNN_counters.json -- the counter summary bench.py reads -- from rocprofv3 output directories.

    python tools/pmc_summary.py --out profiles/r02_counters.json --label "..." DIR [DIR ...]

Every DIR is the `-d` directory of ONE rocprofv3 pass over the same bench command:
    rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 bench.py ...      (durations)
    rocprofv3 --kernel-trace --pmc C1 C2 .. --output-format csv -d DIR -- python3 bench.py ... (counters; FETCH_SIZE and
                                                                                               WRITE_SIZE in passes of their own)
For each kernel family (name prefix) the mean per dispatch of every counter found and the mean kernel-trace duration
of the stats pass are written, together with what the kernel ran on (grid, workgroup, VGPRs).  No number is typed in.
"""
import argparse
import csv
import glob
import json
import os
import re

FAMILIES = {            # JSON key -> substring of the demangled kernel name
    "tvl1_persist_kernel": "tvl1_persist_kernel",
    "tvl1_chunk_kernel": "tvl1_chunk_kernel",
    "tvl1_warp_kernel": "tvl1_warp_kernel",
    "nl_fused_kernel": "nl_fused_kernel",
    "nl_primal_kernel": "nl_primal_kernel",
    "nl_dual_kernel": "nl_dual_kernel",
    "nl_warp_kernel": "nl_warp_kernel",
    "grow_kernel": "grow_kernel",
}
KB_COUNTERS = ("FETCH_SIZE", "WRITE_SIZE")


def families(name, grid, want_grid):
    """JSON keys a dispatch counts for: its family, when no grid filter is set for it or the grid matches, and every
    `family@tag` entry whose grid matches (the bench's secondary legs launch the same kernels on other schedules)."""
    keys = []
    for key, sub in FAMILIES.items():
        if sub not in name:
            continue
        if key not in want_grid or want_grid[key] == grid:
            keys.append(key)
        keys += [k for k, g in want_grid.items() if k.startswith(key + "@") and g == grid]
    return keys


def source_hash():
    """The stamp bench.py compares (bench.kernel_source_hash): sha256 over csrc/*.hip|hpp|inc."""
    import hashlib
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "faldoi-ipol_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp", ".inc")):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--label", default="")
    ap.add_argument("--chunk-tile", default="", help="R,NW,chunk,pairs,W,H of the chunk kernel's schedule in these runs")
    ap.add_argument("--persist-tile", default="", help="R,NW,k,pairs,WxH,warps,inner of the persistent kernel's schedule in these runs")
    ap.add_argument("--tile", action="append", default=[], metavar="KEY=R,NW,chunk,pairs,WxH",
                    help="the schedule behind another chunk-kernel entry (KEY = tvl1_chunk_kernel@tag)")
    ap.add_argument("--grid", action="append", default=[], metavar="FAMILY[@TAG]=THREADS",
                    help="count only the dispatches of FAMILY whose grid has THREADS work-items (the bench's secondary legs launch the "
                         "same kernels on other sizes); FAMILY@TAG=THREADS adds a second entry of that family for another grid")
    ap.add_argument("dirs", nargs="+")
    a = ap.parse_args()
    want_grid = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.grid}
    acc = {}                                      # family -> counter -> [sum, count]
    durs = {}                                     # family -> every kernel-trace duration of the duration pass (ns)
    meta = {}
    pmc_passes = 0
    for d in a.dirs:
        cc = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
        kt = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)
        for path in cc:
            pmc_passes += 1
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    for fam in families(row["Kernel_Name"], int(row["Grid_Size"]), want_grid):
                        s = acc.setdefault(fam, {}).setdefault(row["Counter_Name"], [0.0, 0])
                        s[0] += float(row["Counter_Value"])
                        s[1] += 1
                        m = meta.setdefault(fam, {})
                        m["name"] = row["Kernel_Name"].split("(")[0].replace("void ", "")
                        m["grid_threads"] = int(row["Grid_Size"])
                        m["workgroup"] = int(row["Workgroup_Size"])
                        m["vgprs"] = int(row["VGPR_Count"])
                        m["lds_bytes"] = int(row["LDS_Block_Size"])
        if not cc:                                # a duration pass: kernel-trace only (counters perturb durations)
            for path in kt:
                with open(path, newline="") as f:
                    for row in csv.DictReader(f):
                        grid = int(row["Grid_Size_X"]) * int(row["Grid_Size_Y"]) * int(row["Grid_Size_Z"])
                        for fam in families(row["Kernel_Name"], grid, want_grid):
                            s = acc.setdefault(fam, {}).setdefault("_dur_ns", [0.0, 0])
                            s[0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                            s[1] += 1
                            durs.setdefault(fam, []).append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    out = {"_label": a.label, "_source": [os.path.relpath(d) for d in a.dirs], "_pmc_passes": pmc_passes,
           "_source_hash": source_hash()}
    for fam, counters in acc.items():
        e = dict(meta.get(fam, {}))
        for c, (tot, n) in sorted(counters.items()):
            if c == "_dur_ns":
                e["avg_us"] = round(tot / n / 1e3, 3)
                e["launches_timed"] = n
                d = sorted(durs.get(fam, []))
                if d:                             # the median is what bench.py compares its live duration with (a mean is pulled up by
                    e["median_us"] = round((d[(len(d) - 1) // 2] + d[len(d) // 2]) / 2e3, 3)   # the odd slow launch and the profiler)
                    e["min_us"] = round(d[0] / 1e3, 3)
                    e["max_us"] = round(d[-1] / 1e3, 3)
            elif c in KB_COUNTERS:
                e[c + "_KB"] = round(tot / n, 3)
            else:
                e[c] = round(tot / n, 2)
        out[fam] = e
    if a.chunk_tile and "tvl1_chunk_kernel" in out:
        r, nw, k, pairs, w, h = (int(x) for x in re.split("[,x]", a.chunk_tile))
        out["tvl1_chunk_kernel"].update({"tile": [r, nw], "chunk": k, "pairs": pairs, "size": [w, h]})
    for kv in a.tile:
        key, val = kv.split("=")
        if key in out:
            r, nw, k, pairs, w, h = (int(x) for x in re.split("[,x]", val))
            out[key].update({"tile": [r, nw], "chunk": k, "pairs": pairs, "size": [w, h]})
    if a.persist_tile and "tvl1_persist_kernel" in out:
        r, nw, k, pairs, w, h, warps, inner = (int(x) for x in re.split("[,x]", a.persist_tile))
        out["tvl1_persist_kernel"].update({"tile": [r, nw], "chunk": k, "pairs": pairs, "size": [w, h], "schedule": [warps, inner]})
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", a.out, "families:", sorted(k for k in out if not k.startswith("_")))


if __name__ == "__main__":
    main()
