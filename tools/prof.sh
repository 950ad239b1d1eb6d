#!/bin/bash
# usage: tools/prof.sh <tag> <config> [frames_per_launch=16] [launches=4]   (run on the GPU box via gpurun)
# rocprofv3 kernel-trace + stats, then separate PMC passes, over `python bench.py --pmc-child`: the scene of
# `--config`, every trace-kernel dispatch = one launch of <frames_per_launch> frames, one kernel per launch on one
# stream (RT_TWO_STREAMS=0 RT_COALESCE=0) — the mode of bench.py's roofline pass, so per-dispatch durations and
# counters are those of `roofline.avg_launch_ms` / `valu_insts_per_launch`.  Outputs under gpurun_out/<tag>/.
TAG=$1; CFG=$2; FPL=${3:-16}; N=${4:-4}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CHILD="python $R/bench.py --pmc-child --config $CFG --steps $N --warmup 1 --frames-per-launch $FPL"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CHILD > $OUT/trace.out 2> $OUT/trace.err
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --kernel-trace -d $OUT/pmc_$name -o pmc -- $CHILD > /dev/null 2> $OUT/pmc_$name.err
done
echo "# config $CFG, $FPL frames per launch, $N launches (+1 warm-up launch); per-dispatch figures = per $FPL-frame launch" > $OUT/summary.txt
python $R/tools/prof_summary.py $OUT >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
