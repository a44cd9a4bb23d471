#!/bin/bash
# Round 3, VERDICT item 5: where do the B = 2048 droop of the per-bin outputs and the spread of the complex STFT's launch times come from?
# Per-dispatch durations (kernel trace) and translation / clock / write-request counters at B = 256 and B = 2048.  Run on the GPU box
# from the repo root: bash tools/profile_droop.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/droop
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for wl in linear_power stft; do for B in 256 2048; do
  tag=${wl}_B$B
  it=$([ $B = 256 ] && echo 600 || echo 80)
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -- python3 $ROOT/tools/prof_driver.py $wl $it $B > $OUT/trace_$tag.log 2>&1
  pmc() { name=$1; shift; timeout 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/${name}_$tag -- python3 $ROOT/tools/prof_driver.py $wl 12 $B > $OUT/${name}_$tag.log 2>&1; }
  pmc tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum
  pmc clk GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
  pmc wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum
  pmc tag TCC_TAG_STALL_sum TCC_BUSY_sum TCC_HIT_sum TCC_MISS_sum
done; done
cd $ROOT
python3 tools/summarize_droop.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
