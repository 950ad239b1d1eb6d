"""Per-dispatch counters of `vmem_gather -1 <iters>` under rocprofv3 --pmc: accesses / busy cycles per load instruction per CU.
usage: python tools/ubench/vmem_gather_pmc.py <rocprof output dir> <iters> <log of the run>"""
import glob, os, re, sqlite3, sys
out, iters, log = sys.argv[1], int(sys.argv[2]), sys.argv[3]
lines = [l for l in open(log) if l.startswith("dispatch ")]
meta = []
for l in lines:
    m = re.match(r"dispatch (\d+): (\S+) pattern (\d+), (\d+) iterations x (\d+) loads x (\d+) waves per CU \| (.*)", l)
    meta.append((m.group(2), int(m.group(5)), int(m.group(6)), m.group(7).strip()))
for d in sorted(glob.glob(os.path.join(out, "**", "*.db"), recursive=True)):
    cur = sqlite3.connect(d).cursor()
    rows = list(cur.execute("select dispatch_id, counter_name, sum(value) from counters_collection where kernel_name like '%gather%' group by dispatch_id, counter_name order by dispatch_id"))
    ids = sorted({r[0] for r in rows})
    print("# per wave64 load instruction per CU (256 CUs): counter / (waves per CU x iterations x loads per iteration)")
    for n, did in enumerate(ids):
        if n >= len(meta):
            break
        lvl, lpi, wpc, name = meta[n]
        instr = wpc * iters * lpi
        vals = {c: v for (i, c, v) in rows if i == did}
        print("%s %-18s " % (lvl, name.split()[0]) + "  ".join("%s %.2f" % (c.replace("_sum", ""), v / 256.0 / instr) for c, v in sorted(vals.items())))
