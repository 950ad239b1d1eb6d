#!/bin/bash
# round 5, GPU call 1: layout A/B (bit-exact gate + ms/frame), memory-path counters, the one-rank RCCL run, the GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r05
(cd /tmp && rocprofv3 -L > $R/gpurun_out/r05/counters_avail.txt 2>&1; true)
V=(
 "dense||RT_LAYOUT=dense"
 "pre||RT_LAYOUT=pre"
 "hot6||RT_LAYOUT=hot=6"
 "hot10||RT_LAYOUT=hot=10"
 "align||RT_LAYOUT=align"
 "arena||RT_LAYOUT=arena"
 "pre,arena||RT_LAYOUT=pre,arena"
 "pre,arena,palign||RT_LAYOUT=pre,arena,palign"
 "pre,hot8,arena||RT_LAYOUT=pre,hot=8,arena"
 "trint-dense|trint|RT_LAYOUT=dense"
 "trint-pre,arena|trint|RT_LAYOUT=pre,arena"
 "xcd-control-lpt|xcd|RT_LAYOUT=dense RT_XCD_AFFINITY=1"
 "xcd-bands|xcd|RT_LAYOUT=dense RT_XCD_AFFINITY=1 RT_LPT=0"
 "xcd-blocks|xcd|RT_LAYOUT=dense RT_XCD_AFFINITY=2"
 "nolpt||RT_LAYOUT=dense RT_LPT=0"
 "dense-again||RT_LAYOUT=dense"
)
timeout 900 tools/ab_layout.sh gpurun_out/r05/ab_layout_sweep1.txt 4,5,6 16 1 "${V[@]}" > /dev/null 2>&1
for cfg in 4 5; do
  RT_LAYOUT=dense timeout 300 tools/prof_layout.sh r05/mem_dense_c$cfg $cfg 16 3 > /dev/null 2>&1
  RT_LAYOUT=pre,arena timeout 300 tools/prof_layout.sh r05/mem_prearena_c$cfg $cfg 16 3 > /dev/null 2>&1
done
timeout 600 python -m pytest tests/test_zz_dist_gpu.py -x -q -m gpu -k "one_rank" > gpurun_out/r05/nccl_one_rank.txt 2>&1
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r05/gpu_suite.txt 2>&1
tail -5 gpurun_out/r05/gpu_suite.txt; tail -3 gpurun_out/r05/nccl_one_rank.txt; cat gpurun_out/r05/ab_layout_sweep1.txt | grep -E "===|config|golden" | head -120
