#!/bin/bash
# round 5, GPU call 3: the pair-cooperative inner step (make pair-fetch) against the product build, counters, the GPU suite with the new default layout
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r05
V=(
 "product||"
 "pairfetch|pairfetch|"
 "pairfetch2|pairfetch2|"
)
timeout 900 tools/ab_layout.sh gpurun_out/r05/ab_pairfetch.txt 2,3,4,5,6 16 2 "${V[@]}" > /dev/null 2>&1
RT_HIP_LIB=$R/ray-tracing_amd/lib/libraytrace_hip_pairfetch.so timeout 300 tools/prof_layout.sh r05/mem_pairfetch_c4 4 16 3 > /dev/null 2>&1
RT_HIP_LIB=$R/ray-tracing_amd/lib/libraytrace_hip_pairfetch.so timeout 300 tools/prof_layout.sh r05/mem_pairfetch_c6 6 16 3 > /dev/null 2>&1
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r05/gpu_suite_call3.txt 2>&1
RT_HIP_LIB=$R/ray-tracing_amd/lib/libraytrace_hip_pairfetch.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_pin.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r05/gpu_suite_pairfetch.txt 2>&1
tail -5 gpurun_out/r05/gpu_suite_call3.txt; tail -5 gpurun_out/r05/gpu_suite_pairfetch.txt; grep -E "===|config|golden" gpurun_out/r05/ab_pairfetch.txt; cat gpurun_out/r05/mem_pairfetch_c4/summary.txt
