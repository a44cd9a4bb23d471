#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs written by tools/profile.sh into one text summary (per-kernel averages).

`--traffic <workload>` also merges that workload's HBM bytes per launch — 2 * FETCH_SIZE + WRITE_SIZE of the tuned kernel, KiB ->
bytes, FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-byte read requests at 64 B) — into
profiles/traffic_latest.json, stamped with the hash of the kernel sources the library was built from (bench.py reports
`roofline.traffic` only while that stamp matches the sources it runs).  No number in that file is typed by hand."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_traffic(workload: str, per_kernel: dict) -> None:
    """per_kernel: kernel name -> {counter -> average per dispatch}."""
    sys.path.insert(0, ROOT)
    import bench

    tuned = {k: v for k, v in per_kernel.items() if "k_r32x16" in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v}
    if len(tuned) != 1:
        print(f"== traffic: expected one tuned kernel with FETCH_SIZE and WRITE_SIZE, found {sorted(tuned)} — nothing written")
        return
    c = next(iter(tuned.values()))
    total = int(round((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0))
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    stamp = bench.kernel_source_stamp()
    try:
        cur = json.load(open(path))
    except Exception:
        cur = {}
    if cur.get("kernel_source_stamp") != stamp:  # measurements of other sources do not carry over
        cur = {}
    cur["_note"] = ("HBM bytes per launch from rocprofv3 PMC passes (tools/profile.sh -> tools/summarize_prof.py --traffic): 2*FETCH_SIZE + "
                    "WRITE_SIZE, KiB -> bytes; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B). "
                    "bench.py reports these only while kernel_source_stamp matches the kernel sources it runs.")
    cur["kernel_source_stamp"] = stamp
    cur[workload] = total
    json.dump(cur, open(path, "w"), indent=2)
    scratch = os.path.join(ROOT, "gpurun_out")  # the GPU box only hands gpurun_out/ back: the copy to commit as profiles/traffic_latest.json
    if os.path.isdir(scratch):
        json.dump(cur, open(os.path.join(scratch, "traffic_latest.json"), "w"), indent=2)
    print(f"== traffic: {workload} = {total} bytes per launch (FETCH_SIZE {c['FETCH_SIZE']:.6g} KiB, WRITE_SIZE {c['WRITE_SIZE']:.6g} KiB) -> {path}")


def main():
    root = sys.argv[1]
    traffic_wl = sys.argv[sys.argv.index("--traffic") + 1] if "--traffic" in sys.argv else None
    per_kernel = defaultdict(dict)
    # kernel stats
    for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
        print("== kernel stats (rocprofv3 --kernel-trace --stats):", os.path.relpath(f, root))
        for row in csv.DictReader(open(f)):
            print("  {Name:70.70s} calls={Calls} avg_ns={AverageNs} min_ns={MinNs} max_ns={MaxNs} pct={Percentage}".format(**row))
    for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_trace.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        # steady-state duration: the stats pass launches several hundred times (the GPU needs ~300 launches from idle to reach its
        # steady clocks), the average over the second half of each kernel's dispatches is the one to compare with
        # bench.py's kernel_ms
        by = defaultdict(list)
        for r in rows:
            if "Start_Timestamp" in r and "End_Timestamp" in r:
                by[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, d in by.items():
            if "sgx" in k and len(d) >= 8:
                h = d[len(d) // 2:]
                print(f"== steady state: {k[:70]} second half of {len(d)} dispatches: avg_ns={sum(h) / len(h):.0f} min_ns={min(h)} max_ns={max(h)}")
        if rows:
            r = rows[-1]
            keys = [k for k in ("Kernel_Name", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                                "Workgroup_Size", "Grid_Size") if k in r]
            print("== last dispatch:", {k: r[k] for k in keys})
    # counters: average per dispatch of each kernel (skip the first, warm-up, dispatch)
    for d in sorted(glob.glob(os.path.join(root, "*"))):
        if not os.path.isdir(d) or os.path.basename(d) == "stats":
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(list))
            for row in csv.DictReader(open(f)):
                acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
            print(f"== pmc pass {os.path.basename(d)}")
            for k, cs in acc.items():
                if "sgx" not in k:
                    continue
                print(f"  {k[:90]}")
                for c, v in cs.items():
                    vv = v[1:] if len(v) > 1 else v
                    per_kernel[k][c] = sum(vv) / len(vv)
                    print(f"      {c:40s} avg/dispatch = {sum(vv) / len(vv):.6g}   (n={len(vv)})")
    if traffic_wl:
        write_traffic(traffic_wl, per_kernel)


if __name__ == "__main__":
    main()
