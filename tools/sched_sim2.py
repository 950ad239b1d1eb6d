"""Trace-driven model of the trace kernel's wave-level scheduling (tools/sched_trace.py makes the traces
from the oracle's per-pixel work logs).  Compares, in wave-instructions per 64 segments and lane
utilisation per phase:
    base : the round-1 kernel (one pixel chain per lane, majority vote, suspension at 3/8)
    k2   : two pixel chains per lane (one traversing, one parked between segments), swap inside the
           traversal loop, shading for whichever chain of a lane waits for it
usage: python tools/sched_sim2.py scratch/trace_cfg3.npz [waves]
Costs per wave-level execution are the measured ones of round 1."""
import sys
import numpy as np

COST = dict(A=149, B=99, C=121, VOTE=35, GLASS=670, RAYGEN=310, BEGIN=230, SHADE=285, SKY=250, REFILL=280, FINISH=40, SWAP=30)
SPP = 8


DEPTHS = {}  # id(token list) -> stack entries held while waiting for each token


def parse(npz):
    z = np.load(npz)
    data, offs = z["data"], z["offs"]
    pixels = []
    for p in range(len(offs) - 1):
        b = data[offs[p]:offs[p + 1]].tolist()
        samples, i, n = [], 0, len(b)
        cur_sample = None
        while i < n:
            t = b[i]
            if t == 82:  # R
                cur_sample = []
                samples.append(cur_sample)
                i += 1
            elif t == 83:  # S: segment
                i += 1
                toks, model, deps, depth = [], None, [], []
                while i < n and b[i] in (65, 66, 67):
                    if b[i] == 65:
                        if model is not None and model != ["A", "B"]:
                            toks += model
                            deps += depth
                        model = ["A"]
                        depth = [0]
                        i += 1
                    elif b[i] == 66:
                        model.append("B")
                        depth.append(b[i + 1])
                        i += 2
                    else:
                        model.append(b[i + 1])  # leaf with n tests (int)
                        depth.append(b[i + 2])
                        i += 3
                if model is not None and model != ["A", "B"]:
                    toks += model
                    deps += depth
                DEPTHS[id(toks)] = deps
                outcome = chr(b[i]) if i < n and b[i] in (75, 79, 71) else "E"
                if outcome != "E":
                    i += 1
                cur_sample.append((toks, outcome))
            else:
                raise ValueError(t)
        pixels.append(samples)
    return pixels


class Chain:
    __slots__ = ("px", "si", "gi", "pos", "state", "toks", "outcome", "lit")

    def __init__(self):
        self.state = "D"

    def assign(self, px):
        self.px, self.si, self.gi, self.state = px, 0, 0, "G"

    def raygen_or_finish(self):
        """G: returns 'fin' if the pixel is complete, else 'ray' (state -> I)"""
        if self.si >= len(self.px):
            self.state = "D"
            return "fin"
        self.gi = 0
        self.state = "I"
        return "ray"

    def begin(self):
        self.toks, self.outcome = self.px[self.si][self.gi]
        self.pos = 0
        self.state = "T" if self.toks else "S"

    def phase(self):
        t = self.toks[self.pos]
        return t if isinstance(t, str) else "C"

    def step(self):
        self.pos += 1
        if self.pos >= len(self.toks):
            self.state = "S"

    def shade(self):
        """S -> I (path continues) or G (path over)"""
        self.gi += 1
        if self.gi >= len(self.px[self.si]):
            self.si += 1
            self.state = "G"
        else:
            self.state = "I"


class Acc:
    def __init__(self):
        self.cost = {}
        self.execs = {}
        self.lanes = {}
        self.segments = 0

    def run(self, ph, nlanes, cost=None, mult=1):
        c = (COST[ph] if cost is None else cost) * mult
        self.cost[ph] = self.cost.get(ph, 0) + c
        self.execs[ph] = self.execs.get(ph, 0) + mult
        self.lanes[ph] = self.lanes.get(ph, 0) + nlanes

    def report(self, name):
        tot = sum(self.cost.values())
        k = 64.0 / max(1, self.segments)
        print(f"  {name}: {tot * k:7.0f} instr / 64 segments", end="   ")
        print(" ".join(f"{p}:{self.cost[p] * k:.0f}@{self.lanes[p] / (64.0 * self.execs[p]):.2f}" for p in sorted(self.cost)))
        return tot * k


def traverse_loop(chains_of, acc, lanes_idx, thr_num, thr_den, burst=3, swap=None, thr_abs=None, leaf_in_burst=0):
    """chains_of(i) -> the chain of lane i that is traversing (or None).  swap(i) -> tries to bring in lane i's
    other chain when the active one is finished; returns the new traversing chain or None."""
    def active_list():
        return [(i, c) for i in lanes_idx for c in (chains_of(i),) if c is not None and c.state == "T"]
    act = active_list()
    entered = len(act)
    if not act:
        return
    while True:
        if thr_abs is not None:
            if len(act) <= thr_abs():
                break
        elif len(act) * thr_den <= entered * thr_num:
            break
        cnt = {"A": 0, "B": 0, "C": 0}
        for _, c in act:
            cnt[c.phase()] += 1
        acc.run("VOTE", len(act))
        if cnt["A"] >= cnt["B"] and cnt["A"] >= cnt["C"]:
            served = [c for _, c in act if c.phase() == "A"]
            acc.run("A", len(served))
            for c in served:
                c.step()
        elif cnt["B"] >= cnt["C"]:
            for b in range(burst):
                served = [c for _, c in act if c.state == "T" and c.phase() == "B"]
                if not served:
                    break
                acc.run("B", len(served))
                for c in served:
                    c.step()
                if leaf_in_burst:  # round 5: a lane that reaches a small leaf tests it inside the burst instead of waiting for a C vote
                    small = [c for c in served if c.state == "T" and c.phase() == "C" and c.toks[c.pos] <= leaf_in_burst]
                    if small:
                        rem = [c.toks[c.pos] for c in small]
                        for k in range(max(rem)):
                            acc.run("C", sum(1 for r in rem if r > k))
                        for c in small:
                            c.step()
        else:
            served = [c for _, c in act if c.phase() == "C"]
            rem = [c.toks[c.pos] for c in served]
            m = max(rem)
            # per-triangle executions: lanes with fewer triangles idle
            for k in range(m):
                acc.run("C", sum(1 for r in rem if r > k))
            for c in served:
                c.step()
        if swap is not None:
            want = [i for i, c in act if c.state != "T"]
            if want:
                sw = [i for i in want if swap(i, False)]
                if sw:
                    acc.run("SWAP", len(sw))
                    for i in sw:
                        swap(i, True)
        act = active_list()
        if not act:
            break


def sim_base(pixels, tiles_per_wave, leaf_in_burst=0):
    acc = Acc()
    pool = list(pixels[: tiles_per_wave * 64])
    lanes = [Chain() for _ in range(64)]
    idx = range(64)
    while True:
        idle = [c for c in lanes if c.state == "D"]
        if idle and pool:
            n = 0
            for c in idle:
                if pool:
                    c.assign(pool.pop(0))
                    n += 1
            acc.run("REFILL", n)
        if all(c.state == "D" for c in lanes):
            break
        g = [c for c in lanes if c.state == "G"]
        if g:
            res = [c.raygen_or_finish() for c in g]
            if "ray" in res:
                acc.run("RAYGEN", res.count("ray"))
            if "fin" in res:
                acc.run("FINISH", res.count("fin"))
        b = [c for c in lanes if c.state == "I"]
        if b:
            acc.run("BEGIN", len(b))
            for c in b:
                c.begin()
                acc.segments += 1
        traverse_loop(lambda i: lanes[i], acc, idx, 3, 8, leaf_in_burst=leaf_in_burst)
        s = [c for c in lanes if c.state == "S"]
        shade_stage(s, acc)
    return acc


def shade_stage(s, acc, glass_min=0):
    if not s:
        return
    sky = [c for c in s if c.outcome == "K"]
    hit = [c for c in s if c.outcome in ("O", "G")]
    gl = [c for c in s if c.outcome == "G"]
    if sky:
        acc.run("SKY", len(sky))
    if hit:
        acc.run("SHADE", len(hit))
    if gl:
        acc.run("GLASS", len(gl))
    for c in s:
        c.shade()


def sim_k2(pixels, tiles_per_wave, exit_idle=24, passes=2, pass2_min=16):
    """Two chains per lane.  act[i] = index of the chain whose traversal state is in the lane's registers."""
    acc = Acc()
    pool = list(pixels[: tiles_per_wave * 64])
    ch = [[Chain(), Chain()] for _ in range(64)]
    act = [0] * 64
    idx = range(64)

    def trav_chain(i):
        return ch[i][act[i]]

    def swap(i, do):
        o = ch[i][1 - act[i]]
        if o.state == "T" and o.pos == 0:  # parked ready ray
            if do:
                act[i] = 1 - act[i]
            return True
        return False

    while True:
        idle = [c for pair in ch for c in pair if c.state == "D"]
        if idle and pool:
            n = 0
            for c in idle:
                if pool:
                    c.assign(pool.pop(0))
                    n += 1
            acc.run("REFILL", n)
        if all(c.state == "D" for pair in ch for c in pair):
            break
        # service passes: per lane one chain that is not mid-traversal and needs S / G / I work
        for p in range(passes):
            pick = []
            for i in idx:
                a, o = ch[i][act[i]], ch[i][1 - act[i]]
                cands = [c for c in (a, o) if c.state in ("S", "G", "I")]
                if cands:
                    cands.sort(key=lambda c: "SGI".index(c.state))
                    pick.append((i, cands[0]))
            if not pick or (p > 0 and len(pick) < pass2_min):
                break
            # lanes whose picked chain is the parked one swap it in (and back afterwards if the other is mid-traversal)
            nsw = sum(1 for i, c in pick if c is not ch[i][act[i]])
            if nsw:
                acc.run("SWAP", nsw, mult=2)
            s = [c for _, c in pick if c.state == "S"]
            shade_stage(s, acc)
            g = [c for _, c in pick if c.state == "G"]
            if g:
                res = [c.raygen_or_finish() for c in g]
                if "ray" in res:
                    acc.run("RAYGEN", res.count("ray"))
                if "fin" in res:
                    acc.run("FINISH", res.count("fin"))
            b = [c for _, c in pick if c.state == "I"]
            if b:
                acc.run("BEGIN", len(b))
                for c in b:
                    c.begin()
                    acc.segments += 1
        # make the traversing chain the active one where the active one has nothing to traverse
        for i in idx:
            a, o = ch[i][act[i]], ch[i][1 - act[i]]
            if a.state != "T" and o.state == "T":
                act[i] = 1 - act[i]
        def thr():
            live = sum(1 for pair in ch if pair[0].state != "D" or pair[1].state != "D")
            return live * (64 - exit_idle) // 64
        traverse_loop(trav_chain, acc, idx, 0, 1, swap=swap, thr_abs=thr)
    return acc


if __name__ == "__main__":
    path = sys.argv[1]
    pixels = parse(path)
    ntile = len(pixels) // 64
    print(f"{path}: {ntile} tiles")
    # each simulated wave gets 6 tiles (a real wave renders ~5 of the 32,400 tiles)
    per = 6
    nw = ntile // per
    def many(fn, **kw):
        tot = Acc()
        for w in range(nw):
            a = fn(pixels[w * per * 64:(w + 1) * per * 64], per, **kw)
            for d, s in ((tot.cost, a.cost), (tot.execs, a.execs), (tot.lanes, a.lanes)):
                for k, v in s.items():
                    d[k] = d.get(k, 0) + v
            tot.segments += a.segments
        return tot
    b = many(sim_base).report("base")
    for ei in (16, 24, 32):
        for p2 in (16, 32):
            k = many(sim_k2, exit_idle=ei, pass2_min=p2).report(f"k2 exit_idle={ei} pass2_min={p2}")
            print(f"       -> x{b / k:.2f}")


def sim_kpool(pixels, tiles_per_wave, K=2, ovh_trav=15, ovh_shade=50, rare_min=12, burst=3):
    """K chains per lane, every chain's state addressable (LDS / register select): each iteration the wave
    runs ONE phase, for every lane that has a chain wanting it (one chain per lane).  Phases:
    A B C (traversal), H = shade hit (+ glass for the glass lanes), Y = sky, R = raygen/finish, I = begin_intersect."""
    acc = Acc()
    pool = list(pixels[: tiles_per_wave * 64])
    ch = [[Chain() for _ in range(K)] for _ in range(64)]
    W = dict(A=COST["A"], B=COST["B"], C=COST["C"] * 2.2, H=COST["SHADE"], Y=COST["SKY"], R=COST["RAYGEN"], I=COST["BEGIN"], L=COST["GLASS"])

    def want(c):
        s = c.state
        if s == "T":
            return c.phase()
        if s == "S":
            return "Y" if c.outcome == "K" else ("L" if c.outcome == "G" and getattr(c, "lit", False) else "H")
        if s == "G":
            return "R"
        if s == "I":
            return "I"
        return None

    while True:
        idle = [c for lane in ch for c in lane if c.state == "D"]
        if idle and pool:
            n = 0
            for c in idle:
                if pool:
                    c.assign(pool.pop(0))
                    n += 1
            acc.run("REFILL", n)
        if all(c.state == "D" for lane in ch for c in lane):
            break
        # per phase: one chain per lane
        by = {}
        for lane in ch:
            seen = set()
            for c in lane:
                w = want(c)
                if w is not None and w not in seen:
                    seen.add(w)
                    by.setdefault(w, []).append(c)
        acc.run("VOTE", sum(1 for lane in ch if any(c.state != "D" for c in lane)), cost=45)
        # choose: most lanes served; rare phases only when nothing else has >= rare_min lanes
        best = max(by, key=lambda p: len(by[p]))
        served = by[best]
        if best == "A":
            acc.run("A", len(served), cost=COST["A"] + ovh_trav)
            for c in served:
                c.step()
        elif best == "B":
            acc.run("B", len(served), cost=COST["B"] + ovh_trav)
            for c in served:
                c.step()
            for b in range(burst - 1):  # same chains continue while they stay at inner nodes
                served = [c for c in served if c.state == "T" and c.phase() == "B"]
                if len(served) < 24:
                    break
                acc.run("B", len(served))
                for c in served:
                    c.step()
        elif best == "C":
            rem = [c.toks[c.pos] for c in served]
            acc.run("C", len(served), cost=ovh_trav, mult=1)
            acc.execs["C"] -= 1
            acc.lanes["C"] -= len(served)
            for k in range(max(rem)):
                acc.run("C", sum(1 for r in rem if r > k))
            for c in served:
                c.step()
        elif best == "H":
            acc.run("SHADE", len(served), cost=COST["SHADE"] + ovh_shade)
            for c in served:
                if c.outcome == "G":
                    c.lit = True  # glass part still to do (deferred: phase L)
                else:
                    c.shade()
        elif best == "L":
            acc.run("GLASS", len(served), cost=COST["GLASS"] + ovh_shade)
            for c in served:
                c.lit = False
                c.shade()
        elif best == "Y":
            acc.run("SKY", len(served), cost=COST["SKY"] + ovh_shade)
            for c in served:
                c.shade()
        elif best == "R":
            res = [c.raygen_or_finish() for c in served]
            acc.run("RAYGEN", len(served), cost=COST["RAYGEN"] + ovh_shade)
        elif best == "I":
            acc.run("BEGIN", len(served), cost=COST["BEGIN"] + ovh_shade // 2)
            for c in served:
                c.begin()
                acc.segments += 1
    return acc


if __name__ == "__main__":
    def many_k(fn, K, **kw):
        tot = Acc()
        n = len(pixels)
        for w in range(4):
            px = [pixels[(w * 977 * 64 + j) % n] for j in range(per * K * 64)]
            a = fn(px, per * K, K=K, **kw)
            for d, s in ((tot.cost, a.cost), (tot.execs, a.execs), (tot.lanes, a.lanes)):
                for k, v in s.items():
                    d[k] = d.get(k, 0) + v
            tot.segments += a.segments
        return tot
    for K in (1, 2, 3, 4):
        k = many_k(sim_kpool, K).report(f"kpool K={K}")
        print(f"       -> x{b / k:.2f}")


def sim_noA(pixels, tiles_per_wave, xform=110, load=25, thr=(3, 8), burst=3):
    """base schedule, but the model switch (phase A) is not a voted phase: the local rays of all candidate
    models are computed in the begin_intersect stage (loop to the largest candidate count of the wave)
    and a lane that finishes a model picks the next one up inline (cost folded into the next step)."""
    acc = Acc()
    pool = list(pixels[: tiles_per_wave * 64])
    lanes = [Chain() for _ in range(64)]
    idx = range(64)
    while True:
        idle = [c for c in lanes if c.state == "D"]
        if idle and pool:
            n = 0
            for c in idle:
                if pool:
                    c.assign(pool.pop(0))
                    n += 1
            acc.run("REFILL", n)
        if all(c.state == "D" for c in lanes):
            break
        g = [c for c in lanes if c.state == "G"]
        if g:
            res = [c.raygen_or_finish() for c in g]
            if "ray" in res:
                acc.run("RAYGEN", res.count("ray"))
            if "fin" in res:
                acc.run("FINISH", res.count("fin"))
        b = [c for c in lanes if c.state == "I"]
        if b:
            acc.run("BEGIN", len(b))
            ncand = []
            for c in b:
                c.begin()
                acc.segments += 1
                ncand.append(sum(1 for t in c.toks if t == "A"))
                c.toks = [t for t in c.toks if t != "A"]
                if not c.toks:
                    c.state = "S"
            for k in range(max(ncand)):
                acc.run("A", sum(1 for n in ncand if n > k), cost=xform)
        traverse_loop(lambda i: lanes[i], acc, idx, thr[0], thr[1], burst=burst)
        s = [c for c in lanes if c.state == "S"]
        shade_stage(s, acc)
    return acc


if __name__ == "__main__":
    for thr in ((3, 8), (1, 4), (1, 2)):
        k = many(sim_noA, thr=thr).report(f"noA thr={thr}")
        print(f"       -> x{b / k:.2f}")


def sim_sorted(pixels, tiles_total, NW=4, every=4, sort_cost=45, burst=3, budget=0, timing=None, column_sort=False):
    """NW waves of one workgroup; every `every` iterations all rays of the workgroup are re-sorted by the
    phase they want next (state moved through LDS, `sort_cost` instructions per wave) and dealt out in
    runs of 64; between sorts each wave executes the phase most of its lanes want."""
    acc = Acc()
    pool = list(pixels[: tiles_total * 64])
    lanes = [Chain() for _ in range(64 * NW)]
    order = "ABCHLYRI"

    def want(c):
        s = c.state
        if s == "T":
            return c.phase()
        if s == "S":
            return "Y" if c.outcome == "K" else ("L" if c.outcome == "G" and c.lit else "H")
        if s == "G":
            return "R"
        if s == "I":
            return "I"
        return "Z"
    for c in lanes:
        c.lit = False
    it = 0
    wcost = [0.0] * NW   # instructions each wave issued in the current epoch
    simd_time = 0.0      # sum over epochs of the slowest SIMD (waves w and w + NW/2 share one)
    def close_epoch():
        nonlocal simd_time
        half = max(1, NW // 2)
        simd_time += max(wcost[i] + (wcost[i + half] if i + half < NW else 0) for i in range(half))
        for i in range(NW):
            wcost[i] = 0.0
    while True:
        idle = [c for c in lanes if c.state == "D"]
        if idle and pool:
            n = 0
            for c in idle:
                if pool:
                    c.assign(pool.pop(0))
                    c.lit = False
                    n += 1
            acc.run("REFILL", n, mult=max(1, n // 64))
        if all(c.state == "D" for c in lanes):
            break
        if every and it % every == 0:
            close_epoch()
            if column_sort:
                # every lane column (the NW rays held by lane l of each wave) is sorted on its own: a ray never
                # leaves its column, so its LDS-resident data stays bank-conflict free
                for l in range(64):
                    col = [lanes[w * 64 + l] for w in range(NW)]
                    col.sort(key=lambda c: (order + "Z").index(want(c)))
                    for w in range(NW):
                        lanes[w * 64 + l] = col[w]
            else:
                lanes.sort(key=lambda c: (order + "Z").index(want(c)))
            acc.run("SORT", sum(1 for c in lanes if c.state != "D"), cost=sort_cost, mult=NW)
            for i in range(NW):
                wcost[i] += sort_cost
        it += 1
        for w in range(NW):
            if budget and wcost[w] >= budget:
                continue  # this wave already waits at the barrier
            before = sum(acc.cost.values())
            wl = lanes[w * 64:(w + 1) * 64]
            by = {}
            for c in wl:
                p = want(c)
                if p != "Z":
                    by.setdefault(p, []).append(c)
            if not by:
                continue
            acc.run("VOTE", sum(len(v) for v in by.values()), cost=30)
            best = max(by, key=lambda p: len(by[p]))
            served = by[best]
            if best == "A":
                acc.run("A", len(served))
                for c in served:
                    c.step()
            elif best == "B":
                for b in range(burst):
                    served = [c for c in served if c.state == "T" and c.phase() == "B"]
                    if not served or (b and len(served) < 20):
                        break
                    acc.run("B", len(served))
                    for c in served:
                        c.step()
            elif best == "C":
                rem = [c.toks[c.pos] for c in served]
                for k in range(max(rem)):
                    acc.run("C", sum(1 for r in rem if r > k))
                for c in served:
                    c.step()
            elif best == "H":
                acc.run("SHADE", len(served))
                for c in served:
                    if c.outcome == "G":
                        c.lit = True
                    else:
                        c.shade()
            elif best == "L":
                acc.run("GLASS", len(served))
                for c in served:
                    c.lit = False
                    c.shade()
            elif best == "Y":
                acc.run("SKY", len(served))
                for c in served:
                    c.shade()
            elif best == "R":
                [c.raygen_or_finish() for c in served]
                acc.run("RAYGEN", len(served))
            elif best == "I":
                acc.run("BEGIN", len(served))
                for c in served:
                    c.begin()
                    acc.segments += 1
            wcost[w] += sum(acc.cost.values()) - before
        if budget and all(wcost[w] >= budget or not any(c.state != "D" for c in lanes[w * 64:(w + 1) * 64]) for w in range(NW)):
            it = 0  # everybody reached the barrier: sort at the top of the next round
    close_epoch()
    if timing is not None:
        timing["simd_instr_per_64seg"] = 32.0 * NW * simd_time / max(1, acc.segments)  # comparable with instr / 64 segments of one wave
    return acc


if __name__ == "__main__":
    n = len(pixels)
    for NW in (2, 4, 8):
        for every in (0, 1, 2, 4, 8):
            px = [pixels[j % n] for j in range(6 * NW * 64)]
            k = sim_sorted(px, 6 * NW, NW=NW, every=every).report(f"sorted NW={NW} every={every}")
            print(f"       -> x{b / k:.2f}")


def sim_colpool(pixels, tiles_total, NW=8, PJ=4, swap_cost=22, vote_cost=45, burst=3, weight=False, empty_stack_only=False):
    """Lock-free ray pool: NW waves hold one ray per lane in registers; PJ x 64 more rays are parked in LDS,
    slot (j, col).  Lane l only ever swaps with the PJ slots of column l (bank-conflict-free).  Each iteration a
    wave picks the phase X that the most lanes could serve (own ray wants X, or a parked X-ray in the column),
    misfit lanes swap their ray for a parked X-ray, then X runs.  Waves take turns (round robin)."""
    acc = Acc()
    pool = list(pixels[: tiles_total * 64])
    own = [[Chain() for _ in range(64)] for _ in range(NW)]
    park = [[Chain() for _ in range(64)] for _ in range(PJ)]
    allc = [c for w in own for c in w] + [c for r in park for c in r]
    for c in allc:
        c.lit = False

    def want(c):
        s = c.state
        if s == "T":
            return c.phase()
        if s == "S":
            return "Y" if c.outcome == "K" else ("L" if c.outcome == "G" and c.lit else "H")
        if s == "G":
            return "R"
        if s == "I":
            return "I"
        return None
    Wt = dict(A=1.0, B=1.0, C=1.0, H=1.0, L=1.0, Y=1.0, R=1.0, I=1.0)
    while True:
        idle = [c for c in allc if c.state == "D"]
        if idle and pool:
            n = 0
            for c in idle:
                if pool:
                    c.assign(pool.pop(0))
                    c.lit = False
                    n += 1
            acc.run("REFILL", n, mult=max(1, n // 64))
        if all(c.state == "D" for c in allc):
            break
        for w in range(NW):
            wl = own[w]
            # servable lanes per phase
            cnt = {}
            ownw = [want(c) for c in wl]
            for l in range(64):
                ps = set()
                if ownw[l] is not None:
                    ps.add(ownw[l])
                o = wl[l]
                locked = empty_stack_only and o.state == "T" and DEPTHS[id(o.toks)][o.pos] > 0
                for j in range(0 if locked else PJ):
                    p = want(park[j][l])
                    if p is not None:
                        ps.add(p)
                for p in ps:
                    cnt[p] = cnt.get(p, 0) + 1
            if not cnt:
                continue
            acc.run("VOTE", 64, cost=vote_cost)
            best = max(cnt, key=lambda p: cnt[p] * Wt[p])
            nsw = 0
            for l in range(64):
                o = wl[l]
                if empty_stack_only and o.state == "T" and DEPTHS[id(o.toks)][o.pos] > 0:
                    continue  # a ray with entries on its lane's traversal stack stays with that lane
                if ownw[l] != best:
                    for j in range(PJ):
                        if want(park[j][l]) == best:
                            wl[l], park[j][l] = park[j][l], wl[l]
                            nsw += 1
                            break
            if nsw:
                acc.run("SWAP", nsw, cost=swap_cost)
            served = [c for c in wl if want(c) == best]
            if best == "A":
                acc.run("A", len(served))
                for c in served:
                    c.step()
            elif best == "B":
                for b in range(burst):
                    served = [c for c in served if c.state == "T" and c.phase() == "B"]
                    if not served or (b and len(served) < 24):
                        break
                    acc.run("B", len(served))
                    for c in served:
                        c.step()
            elif best == "C":
                rem = [c.toks[c.pos] for c in served]
                for k in range(max(rem)):
                    acc.run("C", sum(1 for r in rem if r > k))
                for c in served:
                    c.step()
            elif best == "H":
                acc.run("SHADE", len(served))
                for c in served:
                    if c.outcome == "G":
                        c.lit = True
                    else:
                        c.shade()
            elif best == "L":
                acc.run("GLASS", len(served))
                for c in served:
                    c.lit = False
                    c.shade()
            elif best == "Y":
                acc.run("SKY", len(served))
                for c in served:
                    c.shade()
            elif best == "R":
                [c.raygen_or_finish() for c in served]
                acc.run("RAYGEN", len(served))
            elif best == "I":
                acc.run("BEGIN", len(served))
                for c in served:
                    c.begin()
                    acc.segments += 1
    return acc


if __name__ == "__main__":
    n = len(pixels)
    for NW, PJ in ((8, 2), (8, 4), (8, 8), (4, 4), (4, 8), (2, 4), (1, 4), (1, 8)):
        px = [pixels[j % n] for j in range(6 * (NW + PJ) * 64)]
        k = sim_colpool(px, 6 * (NW + PJ), NW=NW, PJ=PJ).report(f"colpool NW={NW} PJ={PJ}")
        print(f"       -> x{b / k:.2f}")
