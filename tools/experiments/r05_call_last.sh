#!/bin/bash
# the last GPU call of round 5 (4.8 GPU-minutes left): the suite and smoke on HEAD, then the driver's bench line with and without the spin-up
TAG=r05_final4; R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout -k 5 30 python tools/golden_check.py > $OUT/golden.txt 2>&1
timeout -k 5 200 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > $OUT/gpu_tests.txt
timeout -k 5 40 python __graft_entry__.py smoke 2>&1 | tail -1 >> $OUT/gpu_tests.txt
timeout -k 5 60 python bench.py --steps 20 --warmup 5 --no-secondary > $OUT/bench_n1_k20.json 2> $OUT/bench_n1_k20.err
timeout -k 5 40 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-pmc --spinup-ms 0 > $OUT/bench_n1_k20_no_spinup.json 2> $OUT/bench_n1_k20_no_spinup.err
timeout -k 5 40 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-pmc > $OUT/bench_n1_k20_b.json 2>/dev/null
timeout -k 5 40 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-pmc --spinup-ms 0 > $OUT/bench_n1_k20_no_spinup_b.json 2>/dev/null
timeout -k 5 90 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
cat $OUT/golden.txt $OUT/gpu_tests.txt; for f in $OUT/bench_n1*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    print(sys.argv[1].split("/")[-1], round(d["value"]), round(d["ms_per_step"],4), d.get("spinup",{}).get("frames"), (d.get("parity") or {}).get("bit_identical"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
P
done
