"""Soak test on the GPU: thousands of per-frame launches interleaved with reads, resets, parameter and
model updates, then a bit comparison of the final buffers with the same frames rendered by the
plainest schedule (one stream, unfused, identity order).  usage: python tools/soak.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
pkg = g.load_package(); api = pkg.load_library()


def run(plain):
    for k, v in (("RT_TWO_STREAMS", "0"), ("RT_FUSE_FRAMES", "0"), ("RT_LPT", "0"), ("RT_COALESCE", "0")):
        if plain:
            os.environ[k] = v
        else:
            os.environ.pop(k, None)
    tr = api.create_tracer(0)
    sc = pkg.scenes.get(3)
    mgr = sc.make_manager(tr, api, 640, 360)
    mgr.OnEnable(renderSeed=9)
    rng = np.random.default_rng(5)
    sig = []
    t0 = time.time()
    frames = 0
    for r in range(rounds):
        n = int(rng.integers(1, 40))
        if plain or r % 3:
            for _ in range(n):
                mgr.RenderFrame()
        else:
            mgr.RenderFrames(n)
        frames += n
        ev = int(rng.integers(0, 6))
        if ev == 0:
            sig.append(tr.read_accumulated().view(np.uint32).sum(dtype=np.uint64))
        elif ev == 1:
            mgr.models[8].transform = pkg.Transform((1.2, 0.3 + 0.1 * (r % 5), -0.5), (10 * r, 20, 0), (0.7, 0.7, 0.7))
            mgr.ResetAccumulatedRender()
        elif ev == 2:
            mgr.divergeStrength = 0.3 + 0.1 * (r % 7)
        elif ev == 3:
            sig.append(tr.counters()["segments"])
    acc = tr.read_accumulated().copy(); frm = tr.read_frame().copy()
    dt = time.time() - t0
    tr.close()
    return acc, frm, sig, frames, dt


a, fa, sa, n, dt = run(False)
print(f"default schedule: {n} frames in {dt:.1f} s")
b, fb, sb, n2, dt2 = run(True)
print(f"plain schedule:   {n2} frames in {dt2:.1f} s")
ok = n == n2 and np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(fa.view(np.uint32), fb.view(np.uint32)) and [int(x) for x in sa] == [int(x) for x in sb]
print("SOAK", "OK: identical buffers, checkpoints and counters" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
