#!/bin/bash
# round-4 record of the queued-stages experiment (GPU box): tools/experiments_q_run.sh <outdir>
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; mkdir -p $OUT
Q=$R/ray-tracing_amd/lib/libraytrace_hip_queued.so
{
echo "=== shipped kernel (product library)"; python $R/tools/qb.py 3,4,6 16 2>&1 | grep -E "golden|config"
echo "=== queued stages (RT_QUEUED=1, R = 3 chains per lane, flush / refill 16, starve 32)"; RT_HIP_LIB=$Q RT_QUEUED=1 python $R/tools/qb.py 3,4,6 16 2>&1 | grep -E "golden|config"
echo "=== queued stages, flush / refill 8"; RT_HIP_LIB=$Q RT_QUEUED=1 RT_Q_FLUSH=8 RT_Q_REFILL=8 python $R/tools/qb.py 4 16 2>&1 | grep -E "config"
echo "=== queued stages, flush / refill 32"; RT_HIP_LIB=$Q RT_QUEUED=1 RT_Q_FLUSH=32 RT_Q_REFILL=32 python $R/tools/qb.py 4 16 2>&1 | grep -E "config"
echo "=== queued stages, phase profile (stats instantiation), 16 frames per launch"; RT_HIP_LIB=$Q RT_QUEUED=1 RT_PHASES=1 RT_PHASE_FRAMES=16 python $R/tools/qb.py 3,4,6 16 2>&1 | grep -v "^config [346]: "
} > $OUT/queued_stages.txt 2>&1
cat $OUT/queued_stages.txt
