#!/bin/bash
# A/B on the GPU box: tools/qab.sh <outfile> <configs> <frames> "<ENV=.. ENV=..>" ...   (each argument = one variant's environment; "-" = none)
OUT=$1; CFGS=$2; FR=$3; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $(dirname $OUT)
for v in "$@"; do
  echo "=== $v" >> $OUT
  if [ "$v" = "-" ]; then timeout 300 python $R/tools/qb.py $CFGS $FR 2>&1 | grep -E "golden|config|Error|error|rror" >> $OUT
  else env $v timeout 300 python $R/tools/qb.py $CFGS $FR 2>&1 | grep -E "golden|config|Error|error|rror" >> $OUT; fi
done
cat $OUT
