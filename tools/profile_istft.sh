#!/bin/bash
# PMC passes over the fused inverse STFT kernel (tools/bench_istft.py as the driver).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_istft
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
DRV="python3 $ROOT/tools/bench_istft.py"
pmc() { name=$1; shift; timeout 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $DRV > $OUT/$name.log 2>&1; }
pmc sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pmc tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum
cd $ROOT
python3 tools/summarize_prof.py $OUT 2>&1 | grep -A12 "istft1024"
