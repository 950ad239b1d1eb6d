#!/bin/bash
# Memory-path counters of the trace kernel for ONE variant (the caller sets RT_LAYOUT / RT_HIP_LIB / ... in the environment):
#   tools/prof_layout.sh <tag> <config> [frames_per_launch=16] [launches=3]
# Separate --pmc passes (never combined with trace domains other than --kernel-trace), summary under gpurun_out/<tag>/.
TAG=$1; CFG=$2; FPL=${3:-16}; N=${4:-3}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CHILD="python $R/bench.py --pmc-child --config $CFG --steps $N --warmup 1 --frames-per-launch $FPL"
for pass in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --kernel-trace -d $OUT/pmc_$name -o pmc -- $CHILD > /dev/null 2> $OUT/pmc_$name.err
done
echo "# $TAG: config $CFG, $FPL frames per launch, $N launches (+1 warm-up); RT_LAYOUT=${RT_LAYOUT:-default} lib=${RT_HIP_LIB:-product}; per-dispatch averages" > $OUT/summary.txt
python $R/tools/prof_summary.py $OUT 2>&1 | grep -E "^   [A-Z]|^== (HBM|L2)" >> $OUT/summary.txt
cat $OUT/summary.txt
