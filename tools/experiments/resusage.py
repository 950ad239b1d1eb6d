"""Registers / spills / scratch of the trace kernels from `make -C ray-tracing_amd/csrc asm` (build/asm/resource_usage.txt)."""
import re, sys
t = open('build/asm/resource_usage.txt').read()
pat = sys.argv[1] if len(sys.argv) > 1 else 'rt_trace'
KEYS = [("VGPR", "VGPRs"), ("SGPR", "SGPRs"), ("spillV", "VGPRs Spill"), ("spillS", "SGPRs Spill"), ("scratch", r"ScratchSize \[bytes/lane\]"), ("occ", r"Occupancy \[waves/SIMD\]")]
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    name = b.split('\n')[0].split(' ')[0]
    if not re.search(pat, name):
        continue
    vals = []
    for label, k in KEYS:
        m = re.search(k + r": (\d+)", b)
        vals.append("%s %s" % (label, m.group(1) if m else "?"))
    print("%-60s %s" % (name[:60], " ".join(vals)))
