#!/bin/bash
# MFMA utilisation of the matrix-core bank epilogue (ERB-64): PMC pass only.  (Round 1 also forced Mel-80 onto it through a
# switch the library no longer has: 11 % of the f32 matrix peak, profiles/r01_mfma_pmc.txt.)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma_erb -- python3 $ROOT/tools/prof_driver.py erb_power 4 > $OUT/mfma_erb.log 2>&1
cd $ROOT
for d in mfma_erb; do
  f=$(find $OUT/$d -name '*counter_collection.csv' | head -1)
  echo "== $d"
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_r32x16" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"   {k:28s} avg/dispatch = {sum(v) / len(v):.6g} (n={len(v)})")
PY
done
