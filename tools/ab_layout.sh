#!/bin/bash
# A/B runs of device-memory layouts and library builds on the GPU box (round 5, profiles/r05_ab_layout*.txt):
#   tools/ab_layout.sh <outfile> <configs> <frames> <reps> <variant>...
# variant = "name|lib-suffix|ENV=value ENV=value ..."   (lib-suffix "" = the product build; RT_LAYOUT etc. in the ENV part)
# Every run starts with the golden fixtures through the HIP path (tools/qb.py): a layout that changed a bit says so.
OUT=$1; CFGS=$2; FR=$3; REPS=$4; shift 4
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $(dirname $OUT)
for rep in $(seq 1 $REPS); do
for v in "$@"; do
  name=${v%%|*}; rest=${v#*|}; suf=${rest%%|*}; envs=${rest#*|}
  lib=$R/ray-tracing_amd/lib/libraytrace_hip${suf:+_$suf}.so
  echo "=== $name (rep $rep)  lib=${suf:-product} env: $envs" >> $OUT
  env $envs RT_HIP_LIB=$lib python $R/tools/qb.py $CFGS $FR 2>&1 | grep -E "golden|config|rror" >> $OUT
done
done
cat $OUT
