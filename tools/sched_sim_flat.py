"""Trace-driven model of the FLAT trace kernel's wave-level schedule (the headline workload, BASELINE config 2), VERDICT r5 item 6:
would parking lanes by phase — sky / shade / raygen executed only when enough lanes want them — raise the lane utilisation above the
shipped 0.58, and by how much in wave-instructions per segment?  Work logs = the oracle's per-pixel token streams (R new sample, S segment,
K sky, O / G opaque / glass hit; tools/sched_trace.py), whole 8x8 tiles; a wave = 64 lanes, one pixel chain per lane (quirk Q13: a pixel's
samples and bounces are ONE serial RNG chain), idle lanes take the pool tile's next pixels like the kernel does.

Policies:
  shipped      every iteration: [finish/raygen] -> intersect -> sky | shade(+glass) for whichever lanes need it (rt_kernels.h, FLAT)
  park(T)      a lane whose next phase is sky (resp. shade, raygen) WAITS until >= T lanes of the wave want that phase or no lane can
               intersect; waiting lanes idle (one chain per lane: their registers hold the parked chain)
  defer(d)     sky evaluation is a pure function of (dir, transmittance, pathLight) and draws no random number: a miss is stashed (9 floats
               in LDS) and the lane goes on with its next sample; path results are added to the pixel in sample order, so every path end
               queues behind a pending one; depth d per lane; the sky phase runs when a lane's queue is full (all pending lanes join)
  two(T)       two pixel chains per lane (the second one parked in LDS, 18 rows per wave): a lane intersects with whichever chain has a
               ray, phases run when >= T lanes have a chain waiting for them; a swap costs SWAP instructions per iteration

Costs (VALU instructions per wave-level execution) are static counts of the shipped FLAT kernel's phases, scaled so that `shipped`
reproduces the measured 1,433 VALU instructions per iteration (9.056e9 per 16-frame launch / 6.32e6 iterations, profiles/pmc_summary.json).
usage: python tools/sched_sim_flat.py scratch/trace_cfg2.npz [waves=48]"""
import sys
import numpy as np

COST = dict(LOOP=45, REFILL=110, FINISH=70, RAYGEN=185, BEGIN=150, ROOTS=52, TRI=175, SHADE=430, GLASS=300, SKY=265, END=15, VOTE=12, STASH=14, SWAP=40)
SPP = 8
P_ROOTS = 1.12  # sphere_roots executions per iteration (phase profile), lane util 0.18: cost charged per iteration


def parse(npz):
    z = np.load(npz)
    data, offs = z["data"], z["offs"]
    pixels = []
    for p in range(len(offs) - 1):
        b = bytes(data[offs[p]:offs[p + 1]].tolist())
        samples, cur, i = [], None, 0
        while i < len(b):
            t = b[i]
            if t == 82:      # R
                cur = []
                samples.append(cur)
                i += 1
            elif t == 83:    # S
                i += 1
            elif t == 65:    # A (model)
                i += 1
            elif t == 66:    # B depth
                i += 2
            elif t == 67:    # C n depth
                i += 3
            elif t in (75, 79, 71):   # K O G
                cur.append(chr(t))
                i += 1
            else:
                raise ValueError(t)
        pixels.append(samples)
    return [pixels[k:k + 64] for k in range(0, len(pixels), 64)]


class Chain:
    """one pixel: samples -> segments; state = what the chain needs next"""
    __slots__ = ("samples", "s", "k", "need")

    def __init__(self, samples):
        self.samples, self.s, self.k = samples, 0, 0
        self.need = "RAYGEN"   # RAYGEN -> INTERSECT -> SKY | SHADE -> (INTERSECT | RAYGEN | DONE)

    def outcome(self):
        return self.samples[self.s][self.k]

    def after_shading(self):
        """advance past the current segment; returns True if the path ended"""
        seg = self.samples[self.s]
        ended = self.outcome() == "K" or self.k + 1 >= len(seg)
        if ended:
            self.s += 1
            self.k = 0
            self.need = "DONE" if self.s >= len(self.samples) else "RAYGEN"
        else:
            self.k += 1
            self.need = "INTERSECT"
        return ended


def run(tiles, policy, arg, waves):
    tot = dict(instr=0.0, useful=0.0, segs=0, iters=0)
    per_wave = max(1, len(tiles) // waves)
    for w in range(waves):
        pool = [px for t in tiles[w * per_wave:(w + 1) * per_wave] for px in t]
        pool.reverse()
        nchains = 2 if policy == "two" else 1
        lanes = [[] for _ in range(64)]
        pend = [0] * 64   # defer: queued path results per lane (sky pending among them)
        sky_pending = [0] * 64

        def charge(name, n_lanes, times=1.0):
            tot["instr"] += COST[name] * times
            tot["useful"] += COST[name] * times * n_lanes / 64.0

        while True:
            # refill
            took = 0
            for l in range(64):
                lanes[l] = [c for c in lanes[l] if c.need != "DONE"]
                while len(lanes[l]) < nchains and pool:
                    lanes[l].append(Chain(pool.pop()))
                    took += 1
            if took:
                charge("REFILL", took)
            if not any(lanes):
                break
            tot["iters"] += 1
            charge("LOOP", sum(1 for l in lanes if l))
            if policy == "two":
                charge("SWAP", 64)
            T = arg if policy in ("park", "two") else 1

            def wanting(what):
                return [(l, c) for l in range(64) for c in lanes[l] if c.need == what]

            def can_intersect():
                return [l for l in range(64) if any(c.need == "INTERSECT" for c in lanes[l])]

            # RAYGEN
            rg = wanting("RAYGEN")
            if policy in ("park", "two"):
                lanes_rg = {l for l, _ in rg}
                go = len(lanes_rg) >= T or not can_intersect()
                charge("VOTE", 64)
            else:
                go = bool(rg)
            if rg and go:
                seen = set()
                for l, c in rg:
                    if l in seen:
                        continue   # one chain per lane per execution
                    seen.add(l)
                    c.need = "INTERSECT"
                charge("RAYGEN", len(seen))
            # INTERSECT: one chain per lane
            seen = []
            for l in range(64):
                for c in lanes[l]:
                    if c.need == "INTERSECT":
                        c.need = "SKY" if c.outcome() == "K" else "SHADE"
                        seen.append(l)
                        break
            if seen:
                n = len(seen)
                charge("BEGIN", n)
                charge("ROOTS", 0.18 * 64, P_ROOTS)
                charge("TRI", n)
                tot["segs"] += n
            # SKY
            sk = wanting("SKY")
            if policy == "defer":
                # stash: the lane goes on; results queue in sample order
                for l, c in sk:
                    pend[l] += 1
                    sky_pending[l] += 1
                    c.after_shading()
                if sk:
                    charge("STASH", len(sk))
                full = [l for l in range(64) if pend[l] >= arg]
                if full or not pool and not can_intersect() and any(sky_pending):
                    while any(sky_pending[l] and (pend[l] >= arg or True) for l in full) or (not full and any(sky_pending)):
                        act = [l for l in range(64) if sky_pending[l]]
                        if not act:
                            break
                        charge("SKY", len(act))
                        for l in act:
                            sky_pending[l] -= 1
                        if not full:
                            continue
                        if all(sky_pending[l] == 0 for l in full):
                            break
                    for l in range(64):
                        if sky_pending[l] == 0:
                            pend[l] = 0
            else:
                lanes_sk = {l for l, _ in sk}
                go = (len(lanes_sk) >= T or not can_intersect()) if policy in ("park", "two") else bool(sk)
                if sk and go:
                    seen2 = set()
                    for l, c in sk:
                        if l in seen2:
                            continue
                        seen2.add(l)
                        c.after_shading()
                    charge("SKY", len(seen2))
                    charge("END", len(seen2))
            # SHADE
            sh = wanting("SHADE")
            lanes_sh = {l for l, _ in sh}
            go = (len(lanes_sh) >= T or not can_intersect()) if policy in ("park", "two") else bool(sh)
            if sh and go:
                seen3, glass, ended = set(), 0, 0
                for l, c in sh:
                    if l in seen3:
                        continue
                    seen3.add(l)
                    glass += c.outcome() == "G"
                    was_defer_end = c.after_shading()
                    ended += was_defer_end
                    if policy == "defer" and was_defer_end and pend[l]:
                        pend[l] += 1   # a known value queues behind the pending sky
                charge("SHADE", len(seen3))
                if glass:
                    charge("GLASS", glass)
                if ended:
                    charge("END", ended)
            # pixel finish
            fin = sum(1 for l in range(64) for c in lanes[l] if c.need == "DONE")
            if fin:
                charge("FINISH", fin)
    return tot


def main():
    tiles = parse(sys.argv[1])
    waves = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    waves = min(waves, len(tiles))
    base = None
    print(f"{len(tiles)} tiles of 8x8 pixels, {waves} model waves ({len(tiles) // waves} tiles each); VALU instructions per 64 segments, lane utilisation")
    for name, pol, arg in (("shipped", "shipped", 0), ("park(8)", "park", 8), ("park(16)", "park", 16), ("park(24)", "park", 24), ("park(32)", "park", 32),
                           ("defer(1)", "defer", 1), ("defer(2)", "defer", 2), ("defer(4)", "defer", 4),
                           ("two(16)", "two", 16), ("two(32)", "two", 32), ("two(48)", "two", 48)):
        r = run(tiles, pol, arg, waves)
        per = r["instr"] / r["segs"] * 64
        if base is None:
            base = per
        print(f"  {name:10s} {per:8.1f} instr / 64 segments ({per / base - 1:+.1%})   lane utilisation {r['useful'] / r['instr']:.3f}   "
              f"{r['instr'] / r['iters']:7.1f} instr / iteration, {r['segs'] / r['iters']:5.1f} segments / iteration")


if __name__ == "__main__":
    main()
