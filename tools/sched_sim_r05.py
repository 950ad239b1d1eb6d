"""Round 5, VERDICT r4 item 7: phases B / C once more through the wave-schedule model (tools/sched_sim2.py on the oracle's work logs in
scratch/trace_cfg{3,4,6}.npz, tools/sched_trace.py) BEFORE touching the kernel — small leaves tested inside the inner-step burst instead of
waiting for a C vote.  Also counts what round 5 found to be the binding resource of the BVH kernels: L1 accesses (one per lane and load
instruction: 4 per inner step, 3 per triangle test, 4 per model entry).
usage: python tools/sched_sim_r05.py > profiles/r05_sched_sim.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sched_sim2 as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOADS = {"A": 4, "B": 4, "C": 3}


def many(pixels, fn, per=6, **kw):
    tot = S.Acc()
    for w in range(len(pixels) // 64 // per):
        a = fn(pixels[w * per * 64:(w + 1) * per * 64], per, **kw)
        for d, s in ((tot.cost, a.cost), (tot.execs, a.execs), (tot.lanes, a.lanes)):
            for k, v in s.items():
                d[k] = d.get(k, 0) + v
        tot.segments += a.segments
    return tot


for cfg in (3, 4, 6):
    path = os.path.join(ROOT, "scratch", f"trace_cfg{cfg}.npz")
    if not os.path.exists(path):
        print(f"config {cfg}: no trace (python tools/sched_trace.py {cfg} 48 {path})")
        continue
    pixels = S.parse(path)
    print(f"==== config {cfg}: {len(pixels) // 64} tiles of 8x8 pixels (instructions per 64 segments; phase:instr@lane-utilisation; L1 accesses per 64 segments)")
    base = None
    for name, kw in (("shipped schedule (burst of 3 inner steps, leaves wait for a C vote)", {}),
                     ("leaves of 1 triangle tested inside the inner-step burst", dict(leaf_in_burst=1)),
                     ("leaves of <= 2 triangles tested inside the burst", dict(leaf_in_burst=2)),
                     ("leaves of <= 4 triangles tested inside the burst", dict(leaf_in_burst=4))):
        a = many(pixels, S.sim_base, **kw)
        tot = a.report(name)
        k = 64.0 / max(1, a.segments)
        acc = sum(a.lanes.get(p, 0) * n for p, n in LOADS.items()) * k
        base = base or (tot, acc)
        print(f"       -> instructions x{base[0] / tot:.3f} of the shipped schedule, L1 accesses {acc:.0f} (x{base[1] / acc:.3f}), votes {a.execs.get('VOTE', 0) * k:.1f} per 64 segments")
