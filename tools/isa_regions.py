"""Static ISA count of the trace kernel per source REGION (sequential attribution: an instruction
belongs to the last rt_kernels.h line seen before it).  Needs an asm built with -gline-tables-only.
usage: python tools/isa_regions.py <file.s> <mangled kernel name>"""
import collections, re, sys

import os
ANCHORS = [  # (region name, text that starts it in rt_kernels.h); a region runs to the next anchor
    ("rand/sky/optics helpers", "float rand_normal(uint32_t* state)"), ("tri_test", "void tri_test("),
    ("begin: spheres", "void begin_intersect("), ("begin: root filter", "Root filter, in lockstep"),
    ("trav: entry vote", "bool traverse("), ("trav A: next model", "---- A: next model"), ("trav B: inner", "---- B: one inner node"),
    ("trav C: leaf", "---- C: one leaf"), ("trav: bottom vote", "        RT_TRAV_VOTE();\n    } while"), ("traverse_flat", "void traverse_flat("),
    ("resolve_hit", "void resolve_hit("), ("cold_args/wave_sum", "const RT_CAS KArgs& cold_args()"), ("prologue", ") rt_trace_kernel(const KArgs a)"),
    ("refill", "---- hand pixels to idle lanes"), ("frame end / accumulate", "phase_mark<STATS>(st, PH_LOOP)"), ("raygen", "next camera ray of this pixel"),
    ("begin call", "phase_mark<STATS>(st, PH_SPHERES)"), ("pool exchange", "void pool_exchange("), ("pool call / loop tail", "if constexpr (POOL) {\n            pool_exchange"), ("sky", "/* the rest of one iteration of Trace's bounce loop — RC:488-538 */"),
    ("shade: common", "resolve the winner"), ("shade: glass", "phase_mark<STATS>(st, PH_GLASS)"),
    ("shade: opaque", "bool isSpecular = mat.specularProbability >= uSpec"), ("roulette/end path", "RC:535-538 Russian roulette"),
    ("epilogue", "exact work counters"), ("(other kernels)", "---- test hooks (rt_debug_*)")]
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracing_amd", "csrc", "rt_kernels.h")).read()
starts = []
for name, text in ANCHORS:
    assert src.count(text) == 1, (name, src.count(text))
    starts.append((src[:src.index(text)].count("\n") + 1, name))
starts.sort()
REGIONS = [(n, a, (starts[k + 1][0] - 1) if k + 1 < len(starts) else 10**9) for k, (a, n) in enumerate(starts)]

s = open(sys.argv[1]).read()
i = s.index(sys.argv[2] + ":")
j = s.index(".Lfunc_end", i)
files = {}
for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s):
    files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
region = "?"
cnt, valu, trans = collections.Counter(), collections.Counter(), collections.Counter()
for l in s[i:j].split("\n"):
    t = l.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m:
        if files.get(int(m.group(1))) == "rt_kernels.h":
            ln = int(m.group(2))
            region = next((n for n, a, b in REGIONS if a <= ln <= b), f"line {ln}")
        continue
    if not t or t[0] in ".;" or t.endswith(":"):
        continue
    cnt[region] += 1
    if t.startswith("v_"):
        valu[region] += 1
    if re.match(r"v_(rcp|rsq|sqrt|div_scale|div_fmas|div_fixup|exp|log|sin|cos)", t):
        trans[region] += 1
print(f"{'region':28s} {'insts':>6s} {'VALU':>6s} {'div/sqrt-seq':>12s}")
for n, _, _ in REGIONS:
    if cnt[n]:
        print(f"{n:28s} {cnt[n]:6d} {valu[n]:6d} {trans[n]:12d}")
for n in cnt:
    if n not in [r[0] for r in REGIONS]:
        print(f"{n:28s} {cnt[n]:6d} {valu[n]:6d} {trans[n]:12d}")
print(f"{'total':28s} {sum(cnt.values()):6d} {sum(valu.values()):6d} {sum(trans.values()):12d}")
