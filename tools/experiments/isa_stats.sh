#!/bin/bash
# build the asm of the library and print register / spill / copy statistics of the main trace kernel
cd /root/repo/ray-tracing_amd/csrc && make 2>&1 | grep -E "error|warning" | head
make asm >/dev/null 2>&1
grep -A8 "Function Name: _ZN3rtk15rt_trace_kernelILb0ELb0" /root/repo/build/asm/resource_usage.txt | grep -E "VGPRs:|Spill|ScratchSize|Occupancy" | sed 's/.*:0: *//; s/ \[-Rpass.*//' | tr '\n' ';'; echo
python - <<'PY'
import re
s=open('/root/repo/build/asm/rt_context-hip-amdgcn-amd-amdhsa-gfx950.s').read()
i=s.index('_ZN3rtk15rt_trace_kernelILb0ELb0ELb0EEEv5KArgs:')
j=s.index('.Lfunc_end',i)
body=s[i:j].split('\n')
def stats(lines,name):
    ins=[l for l in lines if l.strip() and l.strip()[0] not in '.;' and not l.strip().endswith(':')]
    print(name,'insts',len(ins),'valu',sum(l.strip().startswith('v_') for l in ins),'v_mov',sum(bool(re.search(r'\bv_mov_b(32|64)',l)) for l in ins),'readlane',sum('v_readlane' in l for l in ins),'writelane',sum('v_writelane' in l for l in ins),'scratch',sum('scratch_' in l for l in ins))
stats(body,'kernel')
a=[n for n,l in enumerate(body) if 'begin_intersect' in l and 'exit' in l]
b=[n for n,l in enumerate(body) if 'traverse' in l and 'exit' in l]
if a and b: stats(body[a[0]:b[0]],'traverse section')
PY
