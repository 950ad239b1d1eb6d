"""Extended run of the random-scene fuzz of tests/test_gpu_fuzz.py: seeds [first, first + n) through the shipped and the stats kernel
instantiations against the CPU oracle — images bit for bit, exact work counters, no root-filter violation.
RT_FUZZ_SECONDS=s stops taking new seeds after s seconds (a bounded GPU lease) and reports what was done.
usage: python tools/scene_fuzz.py [n=200] [first=1000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
import test_gpu_fuzz as F

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
pkg = g.load_package(); api = pkg.load_library(); orc = g.load_oracle()
bad, t0, segs = 0, time.time(), 0
budget = float(os.environ.get("RT_FUZZ_SECONDS", "0"))
last = first - 1
for seed in range(first, first + n):
    if budget and time.time() - t0 > budget:
        break
    last = seed
    out = []
    for lib, tr, stats in F._three(api, orc):
        sc, render_seed = F.random_scene(pkg, seed)
        if stats:
            tr.enable_stats(True)
        mgr = sc.make_manager(tr, lib)
        mgr.OnEnable(renderSeed=render_seed)
        mgr.RenderFrames(sc.frames)
        acc = tr.read_accumulated()
        c = tr.counters()
        viol = tr.phase_profile()["filter_violations"][0] if stats else 0
        out.append((acc, [c[k] for k in F.KEYS], viol))
        tr.close()
    try:
        ca = F._compare(out, f"seed {seed}")
        segs += ca[0]
    except AssertionError as e:
        bad += 1
        print("MISMATCH", str(e)[:300])
print(f"SCENE FUZZ {'OK' if not bad else 'MISMATCH x %d' % bad}: seeds {first}..{last}, {segs} segments compared, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
