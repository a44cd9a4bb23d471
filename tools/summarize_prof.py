#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs written by tools/profile.sh into one text summary (per-kernel averages).

`--traffic <workload>` also merges that workload's HBM bytes per launch — 2 * FETCH_SIZE + WRITE_SIZE of the tuned kernel, KiB ->
bytes, FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-byte read requests at 64 B) — into
profiles/traffic_latest.json, stamped with the hash of the kernel sources the library was built from (bench.py reports
`roofline.traffic` only while that stamp matches the sources it runs).  No number in that file is typed by hand."""
import csv
import re
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# the kernels whose bytes make up a workload's step (substring of the demangled name); everything else in the trace — torch's fills,
# the forward STFT that makes the istft leg's input, the one-off kernel-spectrum build — is not the leg's traffic
LEG_KERNELS = {"linear_power": ("k_r32x16",), "mel_power": ("k_r32x16",), "mel_db": ("k_r32x16",), "stft": ("k_r32x16",),
               "istft": ("k_istft",), "linear_power_f64": ("k_d32x16",), "mel_db_f64": ("k_d32x16",), "linear_db_f64": ("k_d32x16",), "mfcc": ("k_r32x16",), "config4": ("k_r32x16",), "fft2d": ("k_r32x16", "k_c2c1024"), "convolve_fft": ("k_r32x16", "k_colconv1024", "k_c2r1024")}


def write_traffic(workload: str, per_kernel: dict, sums: dict, durations: dict, iters: int) -> None:
    """per_kernel: kernel name -> {counter -> average per dispatch}; sums: kernel name -> {counter -> (sum over dispatches, dispatches)}."""
    sys.path.insert(0, ROOT)
    import bench

    want = LEG_KERNELS.get(workload, ("k_r32x16",))
    kernels, total = {}, 0.0
    for k, cs in sums.items():
        if not any(w in k for w in want) or "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
            continue
        f_sum, f_n = cs["FETCH_SIZE"]
        w_sum, w_n = cs["WRITE_SIZE"]
        per_step = f_n / iters
        fetch = f_sum / f_n * 1024.0  # bytes per dispatch as the counter reports them
        write = w_sum / w_n * 1024.0
        b = (2.0 * fetch + write) * per_step
        m = re.search(r"(k_\w+(?:<[^>]*>)?)", k)
        short = m.group(1) if m else k[:90]
        kernels[short] = {"dispatches_per_step": per_step, "FETCH_SIZE_bytes_per_dispatch": fetch, "WRITE_SIZE_bytes_per_dispatch": write,
                          "bytes_per_step": int(round(b)), "steady_avg_us": durations.get(k)}
        total += b
    if not kernels:
        print(f"== traffic: no kernel of {want} with FETCH_SIZE and WRITE_SIZE in the passes — nothing written")
        return
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        cur = json.load(open(path))
    except Exception:
        cur = {}
    if "entries" not in cur:
        cur = {"entries": {}}
    cur["_note"] = ("HBM bytes per step from rocprofv3 PMC passes (tools/profile.sh -> tools/summarize_prof.py --traffic): sum over the "
                    "leg's kernels of dispatches per step x (2*FETCH_SIZE + WRITE_SIZE), KiB -> bytes; FETCH_SIZE doubled per "
                    "MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B; calibrated there on 16-B-per-lane streams — the raw "
                    "counter values per dispatch are kept next to the total).  bench.py reports an entry only while its stamp matches the "
                    "sources of the kernels it was taken on.")
    cur["entries"][workload] = {"bytes": int(round(total)), "stamp": bench.kernel_source_stamp(bench.STAMP_FAMILY.get(workload, "stft")),
                                "iters": iters, "kernels": kernels}
    json.dump(cur, open(path, "w"), indent=2)
    scratch = os.path.join(ROOT, "gpurun_out")  # the GPU box only hands gpurun_out/ back: the copy to commit as profiles/traffic_latest.json
    if os.path.isdir(scratch):
        json.dump(cur, open(os.path.join(scratch, "traffic_latest.json"), "w"), indent=2)
    print(f"== traffic: {workload} = {int(round(total))} bytes per step over {len(kernels)} kernel(s) -> {path}")
    for k, v in kernels.items():
        print(f"      {k}: {v['dispatches_per_step']:.2f} dispatches/step, FETCH {v['FETCH_SIZE_bytes_per_dispatch'] / 1e6:.2f} MB (x2), WRITE {v['WRITE_SIZE_bytes_per_dispatch'] / 1e6:.2f} MB, "
              f"{v['bytes_per_step'] / 1e6:.1f} MB/step, steady {v['steady_avg_us']} us")


def main():
    root = sys.argv[1]
    traffic_wl = sys.argv[sys.argv.index("--traffic") + 1] if "--traffic" in sys.argv else None
    per_kernel = defaultdict(dict)
    sums = defaultdict(dict)
    durations = {}
    pmc_iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 12
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
                durations[k] = round(sum(h) / len(h) / 1e3, 2)
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
                    sums[k][c] = (sum(v), len(v))
                    print(f"      {c:40s} avg/dispatch = {sum(vv) / len(vv):.6g}   (n={len(vv)})")
    if traffic_wl:
        write_traffic(traffic_wl, per_kernel, sums, durations, pmc_iters)


if __name__ == "__main__":
    main()
