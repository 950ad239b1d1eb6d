#!/bin/bash
# Reproduce the N=2 one-device bench path many times; keep the full output of every failing run.
# usage: tools/repro_n2.sh [runs] [outdir]
N=${1:-30}; OUT=${2:-gpurun_out/repro_n2}
mkdir -p "$OUT"
fails=0
for i in $(seq 1 "$N"); do
  RT_BENCH_ONE_DEVICE=1 RT_BENCH_BACKEND=gloo OMP_NUM_THREADS=1 timeout 300 \
    python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > "$OUT/run_$i.out" 2> "$OUT/run_$i.err"
  rc=$?
  if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "run $i rc=$rc" >> "$OUT/summary.txt"; grep -E "^\[rank" "$OUT/run_$i.err" | head -40 >> "$OUT/summary.txt"
  else rm -f "$OUT/run_$i.err"; tail -c 600 "$OUT/run_$i.out" | grep -o '"gathered_image_complete": [a-z]*' >> "$OUT/summary.txt"; rm -f "$OUT/run_$i.out"; fi
done
echo "runs=$N fails=$fails" | tee -a "$OUT/summary.txt"
