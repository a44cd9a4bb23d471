#!/usr/bin/env python3
"""Condenses tools/profile_droop.sh: per-dispatch duration series of the tuned kernel (steady half: mean, spread, by launch parity)
and the per-dispatch averages of the counter passes."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for d in sorted(glob.glob(os.path.join(root, "trace_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "k_r32x16" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
        gap = [(int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3 for i in range(len(rows) - 1)]
        if len(dur) < 8:
            continue
        h = dur[len(dur) // 2:]
        mean = sum(h) / len(h)
        sd = (sum((x - mean) ** 2 for x in h) / len(h)) ** 0.5
        ev, od = h[0::2], h[1::2]
        print(f"== {os.path.basename(d)}: {len(dur)} dispatches; second half: mean {mean:.1f} us, sd {sd:.1f} ({100 * sd / mean:.1f} %), min {min(h):.1f}, max {max(h):.1f}; "
              f"even launches {sum(ev) / len(ev):.1f}, odd {sum(od) / len(od):.1f}; median gap {sorted(gap)[len(gap) // 2]:.1f} us")
        blocks = [dur[i:i + max(1, len(dur) // 16)] for i in range(0, len(dur), max(1, len(dur) // 16))]
        print("   per sixteenth of the run (mean us):", " ".join(f"{sum(b) / len(b):.0f}" for b in blocks))
        print("   last 24 launches (us):", " ".join(f"{x:.0f}" for x in dur[-24:]))
for d in sorted(glob.glob(os.path.join(root, "*"))):
    base = os.path.basename(d)
    if not os.path.isdir(d) or base.startswith("trace_"):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "k_r32x16" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        if acc:
            print(f"== {base}: " + "  ".join(f"{c}={sum(v[1:]) / max(1, len(v) - 1):.6g}" for c, v in acc.items()))
