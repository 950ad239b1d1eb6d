#!/bin/bash
# round 5, GPU call 5: pair-fetch with both rounds' loads in flight (RT_PAIR_FETCH=2), VALU issue costs (sparse / chip), a longer soak
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r05
timeout -k 5 200 tools/ubench/valu_ops > gpurun_out/r05/valu_op_rates_v2.txt 2>&1
PF=$R/ray-tracing_amd/lib/libraytrace_hip_pairfetch2.so
if RT_HIP_LIB=$PF timeout -k 5 150 python tools/golden_check.py > gpurun_out/r05/pairfetch2_golden.txt 2>&1; then
  V=( "product||" "pairfetch2|pairfetch2|" "pairfetch|pairfetch|" )
  timeout -k 5 400 tools/ab_layout.sh gpurun_out/r05/ab_pairfetch2.txt 3,4,5,6 16 2 "${V[@]}" > /dev/null 2>&1
  RT_HIP_LIB=$PF timeout -k 5 200 tools/prof_layout.sh r05/mem_pairfetch2_c4 4 16 3 > /dev/null 2>&1
fi
cat gpurun_out/r05/pairfetch2_golden.txt
grep -hE "===|config|golden" gpurun_out/r05/ab_pairfetch2.txt 2>/dev/null; cat gpurun_out/r05/mem_pairfetch2_c4/summary.txt 2>/dev/null | grep -E 'GRBM|TA_BUSY_avr|TOTAL_CACHE|VMEM|INSTS_VALU|WAIT_ANY|WAVE_CYC'
timeout -k 5 420 python tools/soak2.py 120 2 2,3,6 > gpurun_out/r05/soak2_long.txt 2>&1
tail -3 gpurun_out/r05/soak2_long.txt
grep -E 'sparse.*(mul|sub|fma|min|max3|cnds|cmp |mov|lshr|addu|bfe|rcp|cvt|mad24|pkmul |fmas|muls)' gpurun_out/r05/valu_op_rates_v2.txt | head -40
