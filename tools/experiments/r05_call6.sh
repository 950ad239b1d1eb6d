#!/bin/bash
# round 5, GPU call 6: frames-per-launch budget at 20 ms / slabs sized once — the driver's K = 20 shape, the default K, partitions, the tests that touch it, a short soak
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r05b
timeout -k 5 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused or frames or held_back or alternating or single_frame_behind or batch" > gpurun_out/r05b/fuse_tests.txt 2>&1; tail -3 gpurun_out/r05b/fuse_tests.txt
for i in 1 2 3; do timeout -k 5 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-pmc > gpurun_out/r05b/bench_k20_$i.json 2> gpurun_out/r05b/bench_k20_$i.err; done
timeout -k 5 200 python bench.py --no-secondary --no-cpu-baseline --no-pmc > gpurun_out/r05b/bench_k65.json 2> gpurun_out/r05b/bench_k65.err
for c in 3 4; do timeout -k 5 200 python bench.py --config $c --steps 33 --warmup 3 --no-secondary --no-cpu-baseline --no-pmc > gpurun_out/r05b/bench_c$c.json 2> gpurun_out/r05b/bench_c$c.err; done
RT_VERBOSE=1 timeout -k 5 100 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-pmc 2>&1 | grep -E "fused launches" | head -5
timeout -k 5 150 python tools/partition_emulation.py > gpurun_out/r05b/partition_emulation.txt 2>&1; cat gpurun_out/r05b/partition_emulation.txt | grep '^config'
timeout -k 5 240 python tools/soak2.py 60 1 2,3,6 > gpurun_out/r05b/soak2.txt 2>&1; tail -2 gpurun_out/r05b/soak2.txt
for f in gpurun_out/r05b/bench_*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    print(sys.argv[1].split("/")[-1], round(d["value"]), round(d["ms_per_step"],4), d.get("value_fused_launches_one_stream") and round(d["value_fused_launches_one_stream"]), (d.get("parity") or {}).get("bit_identical") if isinstance(d.get("parity"),dict) else None)
except Exception as e:
    print(sys.argv[1], "unreadable", e, open(sys.argv[1].replace(".json",".err")).read()[-400:])
P
done
