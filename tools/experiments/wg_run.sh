#!/bin/bash
# workgroup-kernel experiment on the GPU box: tools/wg_run.sh <outfile> <configs> <frames> "<ENV..>" ...   ("-" = shipped kernel, product library)
OUT=$1; CFGS=$2; FR=$3; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $(dirname $R/gpurun_out/$OUT)
W=$R/ray-tracing_amd/lib/libraytrace_hip_wg.so
for v in "$@"; do
  echo "=== $v" >> $R/gpurun_out/$OUT
  if [ "$v" = "-" ]; then timeout 200 python $R/tools/qb.py $CFGS $FR 2>&1 | grep -E "golden|config|rror|atchdog" >> $R/gpurun_out/$OUT
  else env RT_HIP_LIB=$W $v timeout 200 python $R/tools/qb.py $CFGS $FR 2>&1 | grep -E "golden|config|rror|atchdog|util|workgroups per CU" >> $R/gpurun_out/$OUT; fi
done
cat $R/gpurun_out/$OUT
