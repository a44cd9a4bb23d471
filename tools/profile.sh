#!/bin/bash
# Usage (on the GPU box, from the repo root): bash tools/profile.sh <tag> [workload]
# Writes rocprofv3 kernel stats + PMC passes under gpurun_out/prof_<tag>/ (copy summaries into profiles/).
set -u
TAG=${1:-r01}; WL=${2:-linear_power}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_${WL}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the stats pass launches 800 times: the clocks need ~300 launches from idle; summarize_prof.py reports the second half's average
case $WL in fft2d|convolve_fft) NST=120;; istft|*_f64|config4) NST=400;; *) NST=800;; esac
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tools/prof_driver.py $WL $NST > $OUT/stats.log 2>&1
DRV="python3 $ROOT/tools/prof_driver.py $WL 12"
pmc() { name=$1; shift; timeout 180 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $DRV > $OUT/$name.log 2>&1; }
pmc sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE
pmc sq3 SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INSTS_SMEM
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pmc tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum
pmc ta TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum
cd $ROOT
python3 tools/summarize_prof.py $OUT --traffic $WL --iters 12 > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
