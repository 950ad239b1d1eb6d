#!/bin/bash
# FETCH_SIZE calibration on the GPU box: tools/fetch_calib.sh <outfile>
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; mkdir -p $(dirname $OUT)
cd /tmp && export TMPDIR=/tmp
B=$R/tools/ubench/fetch_calib
{
echo "round 4 — FETCH_SIZE calibration for divergent per-lane record fetches (tools/ubench/fetch_calib.hip, tools/fetch_calib.sh), MI355X"
for pat in 0 1 2 3; do
  for log in 26; do
    [ $pat = 0 ] && l=$((log + 2)) || l=$log
    $B $pat $l
    for pass in "FETCH_SIZE" "TCC_MISS_sum TCC_HIT_sum TCC_REQ_sum"; do
      rm -rf /tmp/fc; rocprofv3 --pmc $pass --kernel-trace -d /tmp/fc -o fc -- $B $pat $l > /dev/null 2>&1
      python - <<PY
import glob, sqlite3
for d in glob.glob("/tmp/fc/**/*.db", recursive=True):
    cur = sqlite3.connect(d).cursor()
    for name, v in cur.execute("select counter_name, sum(value) from counters_collection where kernel_name like '%k_read%' group by counter_name"):
        print("      %-24s %.6g" % (name, v))
PY
    done
  done
done
} > $OUT 2>&1
cat $OUT
