#!/bin/bash
# usage: multi2.sh <name> <waves> [py-patch ...] [-- sed-exprs...]
name=$1; waves=$2; shift 2
d=/tmp/probe/v_$name; rm -rf $d; mkdir -p $d/b; cp -r /tmp/probe/a/include $d/include; cp -r /root/repo/ray-tracing_amd/csrc $d/b/csrc; cd $d/b/csrc
while [ $# -gt 0 ]; do
  if [ "$1" == "--" ]; then shift; break; fi
  python $1 $d/b/csrc || echo "PATCH FAILED $1"; shift
done
for e in "$@"; do sed -i "$e" rt_kernels.h; done
/opt/rocm/bin/hipcc -fno-slp-vectorize -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -DRT_MIN_WAVES_PER_SIMD=$waves -c rt_context.hip -o $d/x.o -Rpass-analysis=kernel-resource-usage 2> $d/ru.txt
echo "$name (waves $waves): $(grep -A12 'Function Name: _ZN3rtk15rt_trace_kernelILb0ELb0ELb0' $d/ru.txt | grep -E ' VGPRs:|VGPRs Spill' | sed 's/.*remark: [^ ]* *//;s/\[-R.*//' | tr '\n' ' ') $(grep -c 'error' $d/ru.txt) errors"
