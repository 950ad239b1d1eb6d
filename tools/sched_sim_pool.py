"""Trace-driven model of a WORKGROUP-SHARED CHAIN POOL for the FLAT trace kernel (round 6, after tools/sched_sim_flat.py showed that no
schedule of ONE wave's 64 chains gets past 0.59-0.63 lane utilisation): the waves of a workgroup exchange pixel chains through LDS so that
a wave executes sky OR shade for (nearly) all of its lanes instead of both for half of them each.

Model (same work logs, same phase costs as sched_sim_flat.py — its `shipped` row is reproduced by pool(0)):
  a workgroup = G waves x 64 lanes, one chain per lane, plus P pool slots in LDS holding chains that wait for `sky` or for `shade`
  (a waiting chain = its ray / hit / path state, 10 resp. 20 dwords; the pixel bookkeeping is addressed by chain, not by lane);
  every wave iteration: [finish / refill / raygen] -> intersect (all lanes with a ray) -> EXCHANGE -> sky and / or shade
  EXCHANGE: the wave counts its lanes that need sky (nS) and shade (nH) and looks at the pool's two queue lengths; it picks as target
  the phase with the larger (own + pooled) demand; lanes of the OTHER phase deposit their chains (while slots are free), lanes without a
  chain (just deposited, or idle because the pixel queue is empty) withdraw chains of the target phase; then the target phase runs, and
  the other phase runs too only for lanes that could not deposit (pool full).  A lane left empty refills from the pixel pool at the top
  of the next iteration like today (more chains in flight, bounded by the slots).
  costs: EXCH_OUT / EXCH_IN wave instructions per exchange with the lanes that take part (LDS b128 writes / reads + queue bookkeeping)

usage: python tools/sched_sim_pool.py scratch/trace_cfg2_512.npz [G=8] [tiles_per_wave=4]"""
import sys
from collections import deque

from sched_sim_flat import COST, Chain, parse, P_ROOTS

EXCH_OUT, EXCH_IN, EXCH_VOTE = 35, 40, 45   # 5 ds_write_b128 + id / rank / atomic bookkeeping; the reverse; the two ballots + queue-length reads


def run(tiles, G, slots, tiles_per_wave, policy="majority"):
    tot = dict(instr=0.0, useful=0.0, segs=0, iters=0, dep=0, wd=0, both=0)
    per_wg = G * tiles_per_wave
    for g0 in range(0, len(tiles) - per_wg + 1, per_wg):
        pixels = [px for t in tiles[g0:g0 + per_wg] for px in t]
        pixels.reverse()
        lanes = [[None] * 64 for _ in range(G)]
        q = {"SKY": deque(), "SHADE": deque()}

        def charge(name, n_lanes, times=1.0, cost=None):
            c = (COST[name] if cost is None else cost) * times
            tot["instr"] += c
            tot["useful"] += c * n_lanes / 64.0

        alive = True
        while alive:
            alive = False
            for w in range(G):
                L = lanes[w]
                # finish + refill
                fin = sum(1 for c in L if c is not None and c.need == "DONE")
                took = 0
                for l in range(64):
                    if L[l] is not None and L[l].need == "DONE":
                        L[l] = None
                    if L[l] is None and pixels:
                        L[l] = Chain(pixels.pop())
                        took += 1
                if fin:
                    charge("FINISH", fin)
                if took:
                    charge("REFILL", took)
                have = [c for c in L if c is not None]
                if not have and not (q["SKY"] or q["SHADE"]):
                    continue
                alive = True
                tot["iters"] += 1
                charge("LOOP", len(have))
                rg = [c for c in have if c.need == "RAYGEN"]
                if rg:
                    for c in rg:
                        c.need = "INTERSECT"
                    charge("RAYGEN", len(rg))
                it = [c for c in have if c.need == "INTERSECT"]
                if it:
                    for c in it:
                        c.need = "SKY" if c.outcome() == "K" else "SHADE"
                    n = len(it)
                    charge("BEGIN", n)
                    charge("ROOTS", 0.18 * 64, P_ROOTS)
                    charge("TRI", n)
                    tot["segs"] += n
                # EXCHANGE
                nS = sum(1 for c in L if c is not None and c.need == "SKY")
                nH = sum(1 for c in L if c is not None and c.need == "SHADE")
                if slots:
                    charge("VOTE", 64, cost=EXCH_VOTE)
                    # cost-weighted demand would favour shade (430 vs 265); plain counts measured better in the model
                    target = "SHADE" if nH + len(q["SHADE"]) >= nS + len(q["SKY"]) else "SKY"
                    other = "SKY" if target == "SHADE" else "SHADE"
                    free = slots - len(q["SKY"]) - len(q["SHADE"])
                    dep = 0
                    for l in range(64):
                        if L[l] is not None and L[l].need == other and free > 0:
                            q[other].append(L[l])
                            L[l] = None
                            free -= 1
                            dep += 1
                    wd = 0
                    for l in range(64):
                        if L[l] is None and q[target]:
                            L[l] = q[target].popleft()
                            wd += 1
                    if dep:
                        charge("X", dep, cost=EXCH_OUT)
                    if wd:
                        charge("X", wd, cost=EXCH_IN)
                    tot["dep"] += dep
                    tot["wd"] += wd
                sk = [c for c in L if c is not None and c.need == "SKY"]
                sh = [c for c in L if c is not None and c.need == "SHADE"]
                if sk and sh:
                    tot["both"] += 1
                if sk:
                    for c in sk:
                        c.after_shading()
                    charge("SKY", len(sk))
                    charge("END", len(sk))
                if sh:
                    glass = sum(1 for c in sh if c.outcome() == "G")
                    ended = sum(1 for c in sh if c.after_shading())
                    charge("SHADE", len(sh))
                    if glass:
                        charge("GLASS", glass)
                    if ended:
                        charge("END", ended)
    return tot


def main():
    tiles = parse(sys.argv[1])
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    tpw = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    print(f"{len(tiles)} tiles; workgroups of {G} waves, {tpw} tiles per wave; VALU instructions per 64 segments, lane utilisation")
    base = None
    for slots in (0, 64, 128, 256, 512, 1024):
        r = run(tiles, G, slots, tpw)
        per = r["instr"] / r["segs"] * 64
        if base is None:
            base = per
        print(f"  pool({slots:4d})  {per:8.1f} instr / 64 segments ({per / base - 1:+.1%})   lane utilisation {r['useful'] / r['instr']:.3f}   "
              f"{r['segs'] / r['iters']:5.1f} segments / iteration   deposits / iteration {r['dep'] / r['iters']:5.1f}  withdrawals {r['wd'] / r['iters']:5.1f}  "
              f"iterations running both phases {r['both'] / r['iters']:.2f}")


if __name__ == "__main__":
    main()
