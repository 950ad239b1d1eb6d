#!/bin/bash
# The last GPU action of a round: GPU test log, bench lines of every config + rocprofv3 profiles of the final build.
# usage (GPU box): tools/final_round.sh <tag>      -> gpurun_out/<tag>/
TAG=${1:-final}; R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -6 > $OUT/gpu_tests.txt
python __graft_entry__.py smoke 2>&1 | tail -1 >> $OUT/gpu_tests.txt
timeout 600 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_n1_k20.json 2> $OUT/bench_n1_k20.err
for c in 3 4 5 6; do
  timeout 400 python bench.py --config $c --steps 33 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/bench_n1_config$c.json 2> $OUT/bench_n1_config$c.err
done
RT_PHASES=1 RT_PHASE_FRAMES=16 timeout 300 python tools/qb.py 2,3,4,6 16 > $OUT/phase_profile_16_frames_per_launch.txt 2>&1
for c in 2 3 4; do timeout 600 tools/prof.sh $TAG/prof_config$c $c 16 4 > /dev/null 2>&1; done
timeout 900 tools/prof.sh $TAG/prof_config5 5 16 2 > /dev/null 2>&1
for c in 2 3 4 5; do cp $OUT/prof_config$c/summary.txt $OUT/config${c}_16frames_per_launch_rocprof_summary.txt 2>/dev/null; cp $OUT/prof_config$c/summary.json $OUT/config${c}_16frames_per_launch_rocprof_summary.json 2>/dev/null; rm -rf $OUT/prof_config$c/trace $OUT/prof_config$c/pmc_*; done
python tools/setup_time.py 4,5 > $OUT/setup_time.txt 2>&1
ls -la $OUT
