#!/bin/bash
# memory-pipe counters of the trace kernel: is the vector memory path (TA / TCP = L1) a bottleneck?
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for pass in "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --kernel-trace -d $OUT/pmc_$name -o pmc -- python $R/bench.py --no-cpu-baseline "$@" > /dev/null 2> $OUT/pmc_$name.err
done
python $R/tools/prof_summary.py $OUT 2>&1 | grep -E "^   [A-Z]"
