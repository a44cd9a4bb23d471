#!/usr/bin/env python3
"""Per-dispatch effective shader clock from one rocprofv3 run that has both --kernel-trace and --pmc GRBM_GUI_ACTIVE:
clock = GRBM_GUI_ACTIVE / 8 XCDs / duration (MI355X_MICROARCH.md, DVFS give-back).  Prints the series in blocks.
    python tools/clock_series.py <rocprof output dir>"""
import csv
import glob
import os
import sys

root = sys.argv[1]
tr = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
cc = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
if not tr or not cc:
    sys.exit(f"{root}: need kernel_trace.csv and counter_collection.csv")
dur = {}
for r in csv.DictReader(open(tr[0])):
    if "k_r32x16" in r["Kernel_Name"]:
        dur[r["Dispatch_Id"]] = (int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
gui = {}
for r in csv.DictReader(open(cc[0])):
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
        gui[r["Dispatch_Id"]] = gui.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
ser = sorted((dur[d][0], dur[d][1], gui[d] / 8.0 / dur[d][1] / 1e3) for d in gui)  # (start, us, GHz)
if not ser:
    sys.exit("no dispatches matched")
n = len(ser)
blk = max(1, n // 16)
print(f"{os.path.basename(root)}: {n} dispatches; per block of {blk}: duration us / effective clock GHz")
for i in range(0, n, blk):
    b = ser[i:i + blk]
    print(f"   launches {i:4d}..{i + len(b) - 1:4d}: {sum(x[1] for x in b) / len(b):8.1f} us   {sum(x[2] for x in b) / len(b):.3f} GHz   (t = {(b[0][0] - ser[0][0]) / 1e6:7.1f} ms)")
