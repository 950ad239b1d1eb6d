#!/bin/bash
# The last GPU action of a round: bench lines of every config + rocprofv3 profiles of the final build.
# usage (GPU box): tools/final_round.sh <tag>      -> gpurun_out/<tag>/
TAG=${1:-final}; R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
for c in 3 4 5 6; do
  timeout 400 python bench.py --config $c --steps 33 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/bench_n1_config$c.json 2> $OUT/bench_n1_config$c.err
done
RT_PHASES=1 RT_PHASE_FRAMES=16 timeout 300 python tools/qb.py 2,3,4,6 16 > $OUT/phase_profile_16_frames_per_launch.txt 2>&1
for c in 2 3 4; do timeout 420 tools/prof.sh $TAG/prof_config$c $c 16 4 > /dev/null 2>&1; done
timeout 600 tools/prof.sh $TAG/prof_config5 5 16 2 > /dev/null 2>&1
for c in 2 3 4 5; do cp $OUT/prof_config$c/summary.txt $OUT/config${c}_16frames_per_launch_rocprof_summary.txt 2>/dev/null; cp $OUT/prof_config$c/summary.json $OUT/config${c}_16frames_per_launch_rocprof_summary.json 2>/dev/null; rm -rf $OUT/prof_config$c/trace $OUT/prof_config$c/pmc_*; done
ls -la $OUT
