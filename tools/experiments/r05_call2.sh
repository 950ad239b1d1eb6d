#!/bin/bash
# round 5, GPU call 2: what the vector-memory path charges (microbenchmark + counters), VALU issue costs at the measured clock,
# second repetition of the layout finalists on configs 2-6, L1 access rates of the other configs
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r05
timeout 300 tools/ubench/vmem_gather > gpurun_out/r05/vmem_gather.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TA_FLAT_READ_WAVEFRONTS_sum --kernel-trace -d $R/gpurun_out/r05/vmem_gather_pmc -o pmc -- $R/tools/ubench/vmem_gather -1 20000 > $R/gpurun_out/r05/vmem_gather_pmc.log 2>&1)
python tools/ubench/vmem_gather_pmc.py gpurun_out/r05/vmem_gather_pmc 20000 gpurun_out/r05/vmem_gather_pmc.log > gpurun_out/r05/vmem_gather_counters.txt 2>&1
timeout 300 tools/ubench/valu_ops > gpurun_out/r05/valu_op_rates.txt 2>&1
V=(
 "dense||RT_LAYOUT=dense"
 "arena||RT_LAYOUT=arena"
 "pre,arena||RT_LAYOUT=pre,arena"
 "nolpt-dense||RT_LAYOUT=dense RT_LPT=0"
 "nolpt-pre,arena||RT_LAYOUT=pre,arena RT_LPT=0"
)
timeout 900 tools/ab_layout.sh gpurun_out/r05/ab_layout_sweep2.txt 2,3,4,5,6 16 2 "${V[@]}" > /dev/null 2>&1
for cfg in 2 3 6; do
  RT_LAYOUT=dense timeout 300 tools/prof_layout.sh r05/mem_dense_c$cfg $cfg 16 3 > /dev/null 2>&1
done
RT_LAYOUT=pre,arena timeout 300 tools/prof_layout.sh r05/mem_prearena_c6 6 16 3 > /dev/null 2>&1
cat gpurun_out/r05/vmem_gather.txt gpurun_out/r05/vmem_gather_counters.txt; head -60 gpurun_out/r05/valu_op_rates.txt; grep -E "===|config" gpurun_out/r05/ab_layout_sweep2.txt
