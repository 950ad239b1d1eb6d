#!/bin/bash
# round 5, GPU call 4: the re-cut launch path (GPU suite + soak), the frames-per-launch budget (partition emulation), then — LAST, behind a
# golden gate with a short timeout: a kernel that does not end occupies the GPU for the rest of the call — the pair-cooperative inner step
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r05
timeout -k 5 600 python -m pytest tests -x -q -m gpu > gpurun_out/r05/gpu_suite_call4.txt 2>&1
tail -4 gpurun_out/r05/gpu_suite_call4.txt
timeout -k 5 150 python tools/partition_emulation.py > gpurun_out/r05/partition_emulation.txt 2>&1
cat gpurun_out/r05/partition_emulation.txt
timeout -k 5 330 python tools/soak2.py 60 1 2,3,6 > gpurun_out/r05/soak2.txt 2>&1
tail -4 gpurun_out/r05/soak2.txt
PF=$R/ray-tracing_amd/lib/libraytrace_hip_pairfetch.so
if RT_HIP_LIB=$PF timeout -k 5 150 python tools/golden_check.py > gpurun_out/r05/pairfetch_golden.txt 2>&1; then
  V=( "product||" "pairfetch|pairfetch|" )
  timeout -k 5 300 tools/ab_layout.sh gpurun_out/r05/ab_pairfetch_fixed.txt 2,3,4,5,6 16 1 "${V[@]}" > /dev/null 2>&1
  timeout -k 5 200 tools/ab_layout.sh gpurun_out/r05/ab_pairfetch_fixed_rep2.txt 3,4,6 16 1 "${V[@]}" > /dev/null 2>&1
  RT_HIP_LIB=$PF timeout -k 5 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_pin.py -x -q -m gpu > gpurun_out/r05/gpu_suite_pairfetch_fixed.txt 2>&1
  RT_HIP_LIB=$PF timeout -k 5 200 tools/prof_layout.sh r05/mem_pairfetch_c4 4 16 3 > /dev/null 2>&1
fi
cat gpurun_out/r05/pairfetch_golden.txt
tail -4 gpurun_out/r05/gpu_suite_pairfetch_fixed.txt 2>/dev/null; grep -hE "===|config|golden" gpurun_out/r05/ab_pairfetch_fixed.txt gpurun_out/r05/ab_pairfetch_fixed_rep2.txt 2>/dev/null; cat gpurun_out/r05/mem_pairfetch_c4/summary.txt 2>/dev/null
