"""Trace-driven model of the QUEUED-STAGES kernel variant (round 4; the same work logs and per-phase costs as
tools/sched_sim2.py): every wave owns R x 64 pixel chains ("slots", records in device memory); 64 of them are in
the wave's traversal lanes, the others wait in per-wave index queues between stages:
    rayQ  -> traversal lanes (continuous refill, A / B / C majority vote as shipped)      -> hitQ
    hitQ  -> SHADE batch of up to 64 (sky / hit / glass), survivors: BEGIN -> rayQ, ended paths -> camQ
    camQ  -> CAMERA batch of up to 64 (finish pixel / refill / camera ray), BEGIN -> rayQ
usage: python tools/sched_sim_queued.py scratch/trace_cfg4.npz"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tools/: sched_sim2.py
from sched_sim2 import COST, Acc, Chain, parse, sim_base, shade_stage

OVH = dict(TLOAD=30, TSTORE=25, SLOAD=45, ISTORE=40, CLOAD=35, SAVE=50)


def sim_queued(pixels, tiles_per_wave, R=3, refill_min=8, burst=3, shade_min=64, cam_min=64):
    acc = Acc()
    pool = list(pixels[: tiles_per_wave * 64])
    chains = [Chain() for _ in range(64 * R)]
    lanes = [None] * 64
    rayQ, hitQ, camQ = [], [], list(chains)

    def begin(cs):
        if not cs:
            return
        acc.run("BEGIN", len(cs))
        acc.run("ISTORE", len(cs), cost=OVH["ISTORE"])
        for c in cs:
            c.begin()
            acc.segments += 1
            (rayQ if c.state == "T" else hitQ).append(c)

    def busy():
        return sum(1 for c in lanes if c is not None)

    def stage_S(batch):
        acc.run("SLOAD", len(batch), cost=OVH["SLOAD"] + (OVH["SAVE"] if busy() else 0))
        shade_stage(batch, acc)
        begin([c for c in batch if c.state == "I"])
        camQ.extend(c for c in batch if c.state == "G")

    def stage_C(batch):
        acc.run("CLOAD", len(batch), cost=OVH["CLOAD"] + (OVH["SAVE"] if busy() else 0))
        fin = 0
        for c in batch:
            if c.state == "G" and c.si >= len(c.px):
                c.raygen_or_finish()
                fin += 1
        if fin:
            acc.run("FINISH", fin)
        need = [c for c in batch if c.state == "D"]
        n = 0
        for c in need:
            if pool:
                c.assign(pool.pop(0))
                n += 1
        if n:
            acc.run("REFILL", n)
        g = [c for c in batch if c.state == "G"]
        if g:
            acc.run("RAYGEN", len(g))
            for c in g:
                assert c.raygen_or_finish() == "ray"
        begin([c for c in batch if c.state == "I"])

    def cam_work():
        return [c for c in camQ if c.state == "G" or pool]

    while True:
        if len(hitQ) >= shade_min:
            b, hitQ[:] = hitQ[:64], hitQ[64:]
            stage_S(b)
            continue
        cw = cam_work()
        if len(cw) >= cam_min:
            b = cw[:64]
            for c in b:
                camQ.remove(c)
            stage_C(b)
            continue
        act = busy()
        free = 64 - act
        if rayQ and (free >= refill_min or act == 0):
            n = min(free, len(rayQ))
            take, rayQ[:] = rayQ[:n], rayQ[n:]
            acc.run("TLOAD", n, cost=OVH["TLOAD"])
            k = 0
            for i in range(64):
                if lanes[i] is None and k < n:
                    lanes[i] = take[k]
                    k += 1
            act = busy()
        if act and any(c is not None and c.state == "T" for c in lanes):
            live = [c for c in lanes if c is not None and c.state == "T"]
            cnt = {"A": 0, "B": 0, "C": 0}
            for c in live:
                cnt[c.phase()] += 1
            acc.run("VOTE", len(live))
            if cnt["A"] >= cnt["B"] and cnt["A"] >= cnt["C"]:
                served = [c for c in live if c.phase() == "A"]
                acc.run("A", len(served))
                for c in served:
                    c.step()
            elif cnt["B"] >= cnt["C"]:
                for _ in range(burst):
                    served = [c for c in live if c.state == "T" and c.phase() == "B"]
                    if not served:
                        break
                    acc.run("B", len(served))
                    for c in served:
                        c.step()
            else:
                served = [c for c in live if c.phase() == "C"]
                rem = [c.toks[c.pos] for c in served]
                for k in range(max(rem)):
                    acc.run("C", sum(1 for r in rem if r > k))
                for c in served:
                    c.step()
            done = [i for i in range(64) if lanes[i] is not None and lanes[i].state != "T"]
            # finished lanes are flushed when enough have gathered (or nothing else is active)
            ndone_total = done
            if done and (len(done) >= refill_min or len(done) == busy()):
                acc.run("TSTORE", len(done), cost=OVH["TSTORE"])
                for i in done:
                    hitQ.append(lanes[i])
                    lanes[i] = None
            continue
        # nothing to traverse: partial batches
        if hitQ:
            b, hitQ[:] = hitQ[:64], hitQ[64:]
            stage_S(b)
        elif cw:
            b = cw[:64]
            for c in b:
                camQ.remove(c)
            stage_C(b)
        else:
            break
    return acc


if __name__ == "__main__":
    for path in sys.argv[1:]:
        pixels = parse(path)
        ntile = len(pixels) // 64
        print(f"==== {path}: {ntile} tiles")

        def many(fn, per, **kw):
            tot = Acc()
            for w in range(ntile // per):
                a = fn(pixels[w * per * 64:(w + 1) * per * 64], per, **kw)
                for d, s in ((tot.cost, a.cost), (tot.execs, a.execs), (tot.lanes, a.lanes)):
                    for k, v in s.items():
                        d[k] = d.get(k, 0) + v
                tot.segments += a.segments
            return tot
        b = many(sim_base, 24).report("shipped schedule")
        for R in (2, 3, 4):
            for rm in (4, 8, 16):
                k = many(sim_queued, 24, R=R, refill_min=rm).report(f"queued stages R={R} refill_min={rm}")
                print(f"       -> x{b / k:.2f}")


def sim_lane_affine(pixels, tiles_per_wave, K=3, shade_min=48, cam_min=48, burst=3, swap_cost=25):
    """The on-chip variant: every LANE owns K chains (registers / LDS of that lane; nothing moves between lanes).  One of a lane's chains
    may be in traversal (it has the lane's LDS stack); the others wait at a segment boundary.  A lane whose traversing chain finishes
    continues with another of ITS chains that is ready to traverse, if it has one.  Shade / camera batches run when at least
    shade_min / cam_min lanes have a chain waiting for that stage (one chain per lane per batch), or when nothing can traverse."""
    acc = Acc()
    pool = list(pixels[: tiles_per_wave * 64])
    ch = [[Chain() for _ in range(K)] for _ in range(64)]
    trav = [None] * 64  # the chain of lane i that holds the stack

    def begin(cs):
        if not cs:
            return
        acc.run("BEGIN", len(cs))
        for c in cs:
            c.begin()
            acc.segments += 1

    def lanes_with(state_pred):
        out = []
        for i in range(64):
            for c in ch[i]:
                if c is not trav[i] and state_pred(c):
                    out.append((i, c))
                    break
        return out

    while True:
        # lanes without a traversing chain pick up one of their ready ones (state T, not started)
        sw = 0
        for i in range(64):
            if trav[i] is None or trav[i].state != "T":
                trav[i] = None
                for c in ch[i]:
                    if c.state == "T":
                        trav[i] = c
                        sw += 1
                        break
        if sw:
            acc.run("SWAP", sw, cost=swap_cost)
        act = [c for c in trav if c is not None]
        need_s = lanes_with(lambda c: c.state == "S")
        need_c = lanes_with(lambda c: c.state in ("G", "D") and (c.state == "G" or pool))
        if len(need_s) >= shade_min or (not act and need_s):
            batch = [c for _, c in need_s]
            shade_stage(batch, acc)
            begin([c for c in batch if c.state == "I"])
            continue
        if len(need_c) >= cam_min or (not act and need_c):
            batch = [c for _, c in need_c]
            fin = 0
            for c in batch:
                if c.state == "G" and c.si >= len(c.px):
                    c.raygen_or_finish()
                    fin += 1
            if fin:
                acc.run("FINISH", fin)
            n = 0
            for c in batch:
                if c.state == "D" and pool:
                    c.assign(pool.pop(0))
                    n += 1
            if n:
                acc.run("REFILL", n)
            g = [c for c in batch if c.state == "G"]
            if g:
                acc.run("RAYGEN", len(g))
                for c in g:
                    c.raygen_or_finish()
            begin([c for c in batch if c.state == "I"])
            continue
        if not act:
            break
        cnt = {"A": 0, "B": 0, "C": 0}
        for c in act:
            cnt[c.phase()] += 1
        acc.run("VOTE", len(act))
        if cnt["A"] >= cnt["B"] and cnt["A"] >= cnt["C"]:
            served = [c for c in act if c.phase() == "A"]
            acc.run("A", len(served))
            for c in served:
                c.step()
        elif cnt["B"] >= cnt["C"]:
            for _ in range(burst):
                served = [c for c in act if c.state == "T" and c.phase() == "B"]
                if not served:
                    break
                acc.run("B", len(served))
                for c in served:
                    c.step()
        else:
            served = [c for c in act if c.phase() == "C"]
            rem = [c.toks[c.pos] for c in served]
            for k in range(max(rem)):
                acc.run("C", sum(1 for r in rem if r > k))
            for c in served:
                c.step()
    return acc


if __name__ == "__main__" and len(sys.argv) > 1:
    for path in sys.argv[1:]:
        pixels = parse(path)
        ntile = len(pixels) // 64
        def many2(fn, per, **kw):
            tot = Acc()
            for w in range(ntile // per):
                a = fn(pixels[w * per * 64:(w + 1) * per * 64], per, **kw)
                for d, s in ((tot.cost, a.cost), (tot.execs, a.execs), (tot.lanes, a.lanes)):
                    for k, v in s.items():
                        d[k] = d.get(k, 0) + v
                tot.segments += a.segments
            return tot
        print(f"==== {path}: lane-affine on-chip variants")
        b = many2(sim_base, 24).report("shipped schedule")
        for K in (2, 3, 4):
            for sm in (32, 48):
                k = many2(sim_lane_affine, 24, K=K, shade_min=sm, cam_min=sm).report(f"lane-affine K={K} batch at {sm} lanes")
                print(f"       -> x{b / k:.2f}")
