"""Static ISA histogram of one kernel by source line (needs an asm built with -gline-tables-only).
usage: python tools/isa_lines.py <file.s> <mangled kernel name> [lo hi]   (lo/hi: restrict to rt_kernels.h lines)"""
import collections, re, sys

s = open(sys.argv[1]).read()
i = s.index(sys.argv[2] + ":")
j = s.index(".Lfunc_end", i)
files = {}
for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s):
    files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
cur = ("?", 0)
cnt, valu = collections.Counter(), collections.Counter()
for l in s[i:j].split("\n"):
    t = l.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    if not t or t[0] in ".;" or t.endswith(":"):
        continue
    cnt[cur] += 1
    if t.startswith("v_"):
        valu[cur] += 1
print("instructions", sum(cnt.values()), "VALU", sum(valu.values()))
byfile = collections.Counter()
for (f, l), c in cnt.items():
    byfile[f] += c
print(dict(byfile))
for f in sorted(byfile):
    rows = sorted((l, c, valu[(f, l)]) for (ff, l), c in cnt.items() if ff == f)
    print("==", f)
    for l, c, v in rows:
        print(f"  {l:5d}  {c:5d}  valu {v:5d}")
