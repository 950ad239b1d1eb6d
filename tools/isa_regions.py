"""Static ISA count of the trace kernel per source REGION (sequential attribution: an instruction
belongs to the last rt_kernels.h line seen before it).  Needs an asm built with -gline-tables-only.
usage: python tools/isa_regions.py <file.s> <mangled kernel name>"""
import collections, re, sys

REGIONS = [  # (name, first line, last line) in rt_kernels.h
    ("rand/sky/optics helpers", 104, 186), ("tri_test", 188, 216), ("begin: spheres", 236, 290), ("begin: root filter", 291, 349),
    ("trav: vote", 365, 387), ("trav A: next model", 388, 415), ("trav B: inner", 416, 481), ("trav C: leaf", 482, 502),
    ("trav: suspend check", 503, 511), ("resolve_hit", 561, 581), ("prologue", 604, 652), ("refill", 653, 704),
    ("frame end / accumulate", 705, 742), ("raygen", 743, 760), ("begin call", 761, 768), ("sky", 769, 777),
    ("shade: common", 778, 792), ("shade: glass", 793, 808), ("shade: opaque", 809, 817), ("roulette/end path", 818, 836),
    ("epilogue", 837, 866)]

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
