"""Monte-Carlo model of the traversal scheduling: one wave (64 lanes, majority vote, suspension)
vs an idealised two-wave pool (128 rays, each wave executes the best phase for 64 of them)."""
import random, sys
random.seed(3)
COST = {"A": 160, "B": 140, "C": 125}   # C per triangle
SHADE = 2800

def make_ray(p_big, models_mean):
    toks = []
    nm = 1 + (random.random() < (models_mean - 1) % 1) + int(models_mean - 1)
    for _ in range(nm):
        toks.append("A")
        if random.random() < p_big:      # icosphere-like: deep walk with leaves interleaved
            steps = random.randint(18, 38)
            leaves = random.randint(4, 8)
        else:                            # cube / quad
            steps = random.randint(1, 3)
            leaves = random.randint(1, 2)
        seq = ["B"] * steps
        pos = sorted(random.sample(range(1, steps + leaves), min(leaves, steps + leaves - 1)))
        out, bi = [], 0
        total = steps + leaves
        lp = set(pos)
        for i in range(total):
            if i in lp:
                out.append(("C", random.randint(1, 4)))
            elif bi < steps:
                out.append("B"); bi += 1
        out.append(("C", random.randint(1, 4)))
        toks += out
    return toks

def phase_of(tok):
    return tok[0] if isinstance(tok, tuple) else tok

def sim_wave(p_big, models_mean, rounds=3000, suspend=3/8, burst=3, lanes=64):
    rays = [make_ray(p_big, models_mean) for _ in range(lanes)]
    pos = [0] * lanes
    cost = 0; segs = 0; trav_cost = 0; lane_steps = 0; execs = 0
    for _ in range(rounds):
        active = [i for i in range(lanes) if pos[i] < len(rays[i])]
        entered = len(active)
        while True:
            active = [i for i in range(lanes) if pos[i] < len(rays[i])]
            if len(active) <= entered * suspend: break
            cnt = {"A": 0, "B": 0, "C": 0}
            for i in active: cnt[phase_of(rays[i][pos[i]])] += 1
            ph = "A" if cnt["A"] >= cnt["B"] and cnt["A"] >= cnt["C"] else ("B" if cnt["B"] >= cnt["C"] else "C")
            if ph == "B":
                for b in range(burst):
                    served = [i for i in active if pos[i] < len(rays[i]) and phase_of(rays[i][pos[i]]) == "B"]
                    if not served: break
                    for i in served: pos[i] += 1
                    c = COST["B"] + (40 if b == 0 else 10); cost += c; trav_cost += c; lane_steps += len(served); execs += 1
            elif ph == "C":
                served = [i for i in active if phase_of(rays[i][pos[i]]) == "C"]
                ntri = max(rays[i][pos[i]][1] for i in served)
                for i in served: pos[i] += 1
                c = COST["C"] * ntri + 40; cost += c; trav_cost += c
            else:
                served = [i for i in active if phase_of(rays[i][pos[i]]) == "A"]
                for i in served: pos[i] += 1
                c = COST["A"] + 40; cost += c; trav_cost += c
        fin = [i for i in range(lanes) if pos[i] >= len(rays[i])]
        if fin:
            cost += SHADE
            for i in fin:
                rays[i] = make_ray(p_big, models_mean); pos[i] = 0; segs += 1
    return cost / segs * 64, trav_cost / segs * 64, lane_steps / max(execs, 1) / lanes

def sim_pool(p_big, models_mean, rounds=3000, lanes=128, width=64):
    """Idealised pool: every time step two 'waves' each serve up to `width` rays of one phase, chosen
    greedily from the pool; finished rays are shaded in batches of `width` (cost SHADE each)."""
    rays = [make_ray(p_big, models_mean) for _ in range(lanes)]
    pos = [0] * lanes
    cost = 0; segs = 0; fin_pending = []
    for _ in range(rounds * 20):
        for wave in range(2):
            active = [i for i in range(lanes) if pos[i] < len(rays[i])]
            by = {"A": [], "B": [], "C": []}
            for i in active: by[phase_of(rays[i][pos[i]])].append(i)
            # finished rays waiting for shading count as a phase of their own
            fin = [i for i in range(lanes) if pos[i] >= len(rays[i])]
            best = max(by, key=lambda p: min(len(by[p]), width) / (COST[p] * (2.5 if p == "C" else 1)))
            if len(fin) >= width or (len(fin) > 0 and not active):
                batch = fin[:width]
                cost += SHADE
                for i in batch:
                    rays[i] = make_ray(p_big, models_mean); pos[i] = 0; segs += 1
                continue
            served = by[best][:width]
            if not served: continue
            if best == "C":
                ntri = max(rays[i][pos[i]][1] for i in served)
                cost += COST["C"] * ntri + 40
            else:
                cost += COST[best] + 40
            for i in served: pos[i] += 1
    return cost / max(segs, 1) * 64

for name, p_big, mm in (("config6-like", 0.16, 2.36), ("config3-like", 0.05, 2.18)):
    w = sim_wave(p_big, mm)
    p = sim_pool(p_big, mm)
    print(f"{name}: one wave: {w[0]:.0f} instr / 64 segments (traversal {w[1]:.0f}, inner-step lane utilisation {w[2]:.2f});  ideal 2-wave pool: {p:.0f}  -> x{w[0]/p:.2f}")
