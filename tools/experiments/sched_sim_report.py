"""Runs the scheduler models of tools/sched_sim2.py on work-log traces (tools/sched_trace.py) and prints one table
per trace: the shipped schedule and the wider-than-wave designs of DESIGN.md §9.2.
usage: python tools/sched_sim_report.py scratch/trace_cfg3.npz [more traces]  > profiles/r02_sched_sim.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tools/: sched_sim2.py
import sched_sim2 as S  # noqa: E402

for path in sys.argv[1:]:
    pixels = S.parse(path)
    n = len(pixels)
    print(f"==== {path}: {n // 64} tiles of 8x8 pixels, 8 spp  (instructions per 64 segments; phase:instr@lane-utilisation)")
    base = S.sim_base(pixels[:64 * 24], 24).report("shipped schedule (one chain per lane, majority vote, suspension at 3/8)")

    def rel(name, acc, extra=""):
        k = acc.report(name)
        print(f"       -> x{base / k:.2f} {extra}")

    rel("two chains per lane, one parked between segments (exit when 24 lanes idle)", S.sim_k2(pixels[:64 * 24], 24, exit_idle=24, pass2_min=16))
    for K in (2, 4):
        px = [pixels[j % n] for j in range(6 * K * 64)]
        rel(f"K={K} rays per lane, per-lane addressable state, one vote over all phases", S.sim_kpool(px, 6 * K, K=K))
    for NW, PJ, eso in ((8, 4, False), (8, 6, False), (8, 8, False), (8, 8, True)):
        px = [pixels[j % n] for j in range(6 * (NW + PJ) * 64)]
        rel(f"column pool NW={NW} waves + PJ={PJ} parked rows (swap 35, vote 60 instr){' only empty-stack rays move' if eso else ''}",
            S.sim_colpool(px, 6 * (NW + PJ), NW=NW, PJ=PJ, swap_cost=35, vote_cost=60, empty_stack_only=eso))
    px = [pixels[j % n] for j in range(6 * 8 * 64)]
    for budget, sc, cs in ((500, 45, False), (800, 80, False), (500, 35, True)):
        t = {}
        acc = S.sim_sorted(px, 6 * 8, NW=8, every=10 ** 9, budget=budget, sort_cost=sc, timing=t, column_sort=cs)
        rel(f"8-wave workgroup, {'per-column' if cs else 'global'} sort by phase every {budget} instr (sort {sc} instr/wave)", acc,
            f"; with barrier waits x{base / t['simd_instr_per_64seg']:.2f}")
    sys.stdout.flush()
