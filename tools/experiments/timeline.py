"""Kernel timeline from a rocprofv3 --kernel-trace rocpd database: start/end of the last N trace-kernel
dispatches relative to the first of them, with the queue/stream each ran on.
usage: python tools/timeline.py <dir with *.db> [N]"""
import glob, os, sqlite3, sys

n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
for d in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)):
    cur = sqlite3.connect(d).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = list(cur.execute(f"select start, end, {qcol}, grid_x, name from kernels where name like '%rt_trace%' or name like '%rt_accumulate%' or name like '%rt_order%' order by start"))
    rows = rows[-n:]
    t0 = rows[0][0]
    for s, e, q, g, name in rows:
        print(f"  start {(s - t0) / 1e3:9.1f} us  end {(e - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  queue {q}  grid {g:6d}  {name[:40]}")
    print("  columns:", cols)
