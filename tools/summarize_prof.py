#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs written by tools/profile.sh into one text summary (per-kernel averages)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    # kernel stats
    for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
        print("== kernel stats (rocprofv3 --kernel-trace --stats):", os.path.relpath(f, root))
        for row in csv.DictReader(open(f)):
            print("  {Name:70.70s} calls={Calls} avg_ns={AverageNs} min_ns={MinNs} max_ns={MaxNs} pct={Percentage}".format(**row))
    for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_trace.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
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
                    print(f"      {c:40s} avg/dispatch = {sum(vv) / len(vv):.6g}   (n={len(vv)})")


if __name__ == "__main__":
    main()
