#!/bin/bash
# Share of 64-byte write requests to memory (the rest are 32-byte partial writes) for one prof_driver workload.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/wrreq_${1:-linear_power}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT -- python3 $ROOT/tools/prof_driver.py ${1:-linear_power} 4 > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:50]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        if "sgx" in k: print(k, {c: "%.4g" % (x / 4) for c, x in v.items()})
PY
