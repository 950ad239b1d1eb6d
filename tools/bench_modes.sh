#!/bin/bash
# bench.py's own timed regions (K back-to-back rt_render_frame, the driver's pattern) for several library / environment variants:
#   tools/bench_modes.sh <tag> "<configs>" "name|ENV=.. ENV=.." ...      -> gpurun_out/<tag>/modes.txt
TAG=$1; CFGS=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
for c in $CFGS; do
  for v in "$@"; do
    name=${v%%|*}; envs=${v#*|}
    line=$(env $envs timeout -k 5 400 python $R/bench.py --config $c --steps 33 --warmup 3 --no-cpu-baseline --no-pmc --no-secondary 2>$OUT/err_${c}_$name.txt | grep '^{' | tail -1)
    python - "$c" "$name" "$line" <<'P' | tee -a $OUT/modes.txt
import json,sys
c,name,line=sys.argv[1:4]
try:
    d=json.loads(line); r=d["regions"]
    print(f"config {c} {name:>22s}: value {d['value']:8.0f} ({d['ms_per_step']:7.3f} ms)  median of {r['n']} regions {d['value_median_of_regions']:8.0f} ({r['ms_per_step_median']:7.3f} ms; {min(r['values']):.0f} .. {max(r['values']):.0f})  fused one stream {d['value_fused_launches_one_stream'] or 0:8.0f}  batched {(d['batched_api'] or {}).get('ms_per_frame',0):7.3f} ms  diag {d['diagnostics']}")
except Exception as e:
    print(f"config {c} {name}: unreadable ({e})")
P
  done
done
