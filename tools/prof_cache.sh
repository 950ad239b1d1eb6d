#!/bin/bash
# Counters of the trace kernel for ONE variant of the top-of-tree cache (the caller sets RT_HOT_KB / RT_WAVES_PER_GROUP / RT_HIP_LIB):
#   tools/prof_cache.sh <tag> <config> [frames_per_launch=16] [launches=3]
# Separate --pmc passes (never combined with trace domains other than --kernel-trace), summary under gpurun_out/<tag>/.
TAG=$1; CFG=$2; FPL=${3:-16}; N=${4:-3}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CHILD="python $R/bench.py --pmc-child --config $CFG --steps $N --warmup 1 --frames-per-launch $FPL"
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
            "TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
            "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" \
            "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_IFETCH_LEVEL GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --kernel-trace -d $OUT/pmc_$name -o pmc -- $CHILD > /dev/null 2> $OUT/pmc_$name.err || echo "pass failed: $pass" >> $OUT/failed.txt
done
echo "# $TAG: config $CFG, $FPL frames per launch, $N launches (+1 warm-up); RT_HOT_KB=${RT_HOT_KB:-auto} RT_WAVES_PER_GROUP=${RT_WAVES_PER_GROUP:-12} lib=${RT_HIP_LIB:-product}; per-dispatch averages" > $OUT/summary.txt
python $R/tools/prof_summary.py $OUT 2>&1 | grep -E "^   [A-Z]|^== " >> $OUT/summary.txt
cat $OUT/failed.txt 2>/dev/null >> $OUT/summary.txt
rm -rf $OUT/pmc_*/
cat $OUT/summary.txt
