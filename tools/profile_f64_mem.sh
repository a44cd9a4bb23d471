#!/bin/bash
# HBM traffic of the f64 n_fft = 1024 generic kernel (separate --pmc passes, MI355X_MICROARCH.md's recipe)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/f64mem
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SGX_PROF_NFFT=${SGX_PROF_NFFT:-1024} SGX_PROF_HOP=${SGX_PROF_HOP:-256} SGX_PROF_DTYPE=${SGX_PROF_DTYPE:-float64}
DRV="python3 $ROOT/tools/prof_driver.py ${SGX_PROF_WORKLOAD:-linear_power} 4"
pmc() { name=$1; shift; timeout 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $DRV > $OUT/$name.log 2>&1; }
pmc fetch FETCH_SIZE && pmc write WRITE_SIZE && pmc tcc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum
python3 - <<PY
import csv, glob, collections
for name in ("fetch", "write", "tcc"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, v in acc.items():
            if "reg_radix" in k:
                print(name, k, {c: "%.4g per launch" % (x / 4) for c, x in v.items()})
PY
