#!/bin/bash
# The last GPU action of a round: GPU test log, bench lines of every config + rocprofv3 profiles of the final build.
# usage (GPU box): tools/final_round.sh <tag>      -> gpurun_out/<tag>/          (every step under its own timeout, ~20 min in all)
TAG=${1:-final}; R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout -k 5 420 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $OUT/gpu_tests.txt
timeout -k 5 120 python __graft_entry__.py smoke 2>&1 | tail -1 >> $OUT/gpu_tests.txt
timeout -k 5 420 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
timeout -k 5 240 python bench.py --steps 20 --warmup 5 --no-secondary > $OUT/bench_n1_k20.json 2> $OUT/bench_n1_k20.err
for c in 3 4 5 6; do
  timeout -k 5 240 python bench.py --config $c --steps 33 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/bench_n1_config$c.json 2> $OUT/bench_n1_config$c.err
done
for c in 2 3 4; do timeout -k 5 300 tools/prof.sh $TAG/prof_config$c $c 16 3 > /dev/null 2>&1; done
for c in 2 3 4; do cp $OUT/prof_config$c/summary.txt $OUT/config${c}_16frames_per_launch_rocprof_summary.txt 2>/dev/null; cp $OUT/prof_config$c/summary.json $OUT/config${c}_16frames_per_launch_rocprof_summary.json 2>/dev/null; rm -rf $OUT/prof_config$c/trace $OUT/prof_config$c/pmc_*; done
RT_PHASES=1 RT_PHASE_FRAMES=16 timeout -k 5 200 python tools/qb.py 2,3,4,6 16 > $OUT/phase_profile_16_frames_per_launch.txt 2>&1
timeout -k 5 120 python tools/setup_time.py 4,5 > $OUT/setup_time.txt 2>&1
ls -la $OUT; cat $OUT/gpu_tests.txt; for f in $OUT/bench_n1*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    r=d.get("roofline") or {}
    mp=r.get("memory_path") or {}
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], "frac", r.get("frac"), "mem", mp.get("frac"), mp.get("ta_busy"), "parity", (d.get("parity") or {}).get("bit_identical") if isinstance(d.get("parity"),dict) else d.get("parity"), ((d.get("parity") or {}).get("vs_reference_text") if isinstance(d.get("parity"),dict) else None))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
P
done
