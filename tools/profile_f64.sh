#!/bin/bash
# f64 n_fft = 1024 / hop 256 linear power through the register-tiled generic kernel: SQ counters (two passes) + kernel stats.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/f64prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SGX_PROF_NFFT=${SGX_PROF_NFFT:-1024} SGX_PROF_HOP=${SGX_PROF_HOP:-256} SGX_PROF_DTYPE=${SGX_PROF_DTYPE:-float64}
DRV="python3 $ROOT/tools/prof_driver.py ${SGX_PROF_WORKLOAD:-linear_power} 4"
pmc() { name=$1; shift; timeout 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $DRV > $OUT/$name.log 2>&1; }
pmc sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE &&
pmc sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT &&
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $DRV > $OUT/stats.log 2>&1
python3 - <<PY
import csv, glob, collections
for name in ("sq1", "sq2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            if "reg_radix" in k or "r32x16" in k:
                print(name, k, {c: "%.4g" % x for c, x in v.items()})
for f in glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:600])
PY
