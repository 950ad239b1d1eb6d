#!/bin/bash
# usage: tools/prof.sh <tag> [bench args...]   (run on the GPU box via gpurun)
# kernel-trace + stats of bench.py, then separate PMC passes; outputs under gpurun_out/<tag>/
# RT_TWO_STREAMS=0 + --no-batched: exactly one trace kernel per frame on one stream, so that
# per-dispatch durations and counters are those of one launch (bench.py's roofline pass uses the same mode)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RT_TWO_STREAMS=0 RT_COALESCE=0 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/bench.py --no-cpu-baseline --no-batched --no-pmc --no-secondary "$@" > $OUT/trace_bench.json 2> $OUT/trace.err
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  RT_TWO_STREAMS=0 RT_COALESCE=0 rocprofv3 --pmc $pass --kernel-trace -d $OUT/pmc_$name -o pmc -- python $R/bench.py --no-cpu-baseline --no-batched --no-pmc --no-secondary "$@" > /dev/null 2> $OUT/pmc_$name.err
done
python $R/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
