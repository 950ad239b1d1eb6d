#!/bin/bash
# A/B runs of library builds on the GPU box: tools/ab.sh <outfile> <configs> <frames> <lib-suffix>...   ("" = the product build)
OUT=$1; CFGS=$2; FR=$3; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $(dirname $OUT)
for rep in 1 2; do
for v in "$@"; do
  lib=$R/ray-tracing_amd/lib/libraytrace_hip${v:+_$v}.so
  echo "=== ${v:-product} (rep $rep)" >> $OUT
  RT_HIP_LIB=$lib python $R/tools/qb.py $CFGS $FR 2>&1 | grep -E "golden|config" >> $OUT
done
done
cat $OUT
