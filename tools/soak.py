"""Soak test on the GPU: thousands of per-frame launches interleaved with reads, resets, parameter and
model updates, then a bit comparison of the final buffers with the same frames rendered by the
plainest schedule (one stream, unfused, identity order).  usage: python tools/soak.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
pkg = g.load_package(); api = pkg.load_library()


PLAIN = (("RT_TWO_STREAMS", "0"), ("RT_FUSE_FRAMES", "0"), ("RT_LPT", "0"), ("RT_COALESCE", "0"))
# SOAK_OFF="RT_ALTERNATE RT_LPT": run the 'default' side with these features switched off too (to find which one a mismatch needs)
OFF = os.environ.get("SOAK_OFF", "").split()


def run(plain):
    for k, v in PLAIN + (("RT_ALTERNATE", "0"),):
        if plain or k in OFF:
            os.environ[k] = v
        else:
            os.environ.pop(k, None)
    tr = api.create_tracer(0)
    sc = pkg.scenes.get(3)
    mgr = sc.make_manager(tr, api, 640, 360)
    mgr.OnEnable(renderSeed=9)
    rng = np.random.default_rng(5)
    sig = []
    kinds = []
    imgs = []
    events = []
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
        events.append((r, n, 'frames' if (plain or r % 3) else 'batch', ev, frames))
        if ev == 0:
            img = tr.read_accumulated().copy(); imgs.append(img)
            sig.append(int(img.view(np.uint32).sum(dtype=np.uint64)))
            kinds.append(f"acc@{frames}")
        elif ev == 1:
            mgr.models[8].transform = pkg.Transform((1.2, 0.3 + 0.1 * (r % 5), -0.5), (10 * r, 20, 0), (0.7, 0.7, 0.7))
            mgr.ResetAccumulatedRender()
        elif ev == 2:
            mgr.divergeStrength = 0.3 + 0.1 * (r % 7)
        elif ev == 3:
            sig.append(tr.counters()["segments"])
            kinds.append(f"segments@{frames}")
    acc = tr.read_accumulated().copy(); frm = tr.read_frame().copy()
    dt = time.time() - t0
    tr.close()
    run.kinds = kinds; run.imgs = imgs; run.events = events
    return acc, frm, sig, frames, dt


a, fa, sa, n, dt = run(False)
imgs_a, events_a = run.imgs, run.events
print(f"default schedule: {n} frames in {dt:.1f} s")
b, fb, sb, n2, dt2 = run(True)
print(f"plain schedule:   {n2} frames in {dt2:.1f} s")
ok = n == n2 and np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(fa.view(np.uint32), fb.view(np.uint32)) and [int(x) for x in sa] == [int(x) for x in sb]
if not ok:
    da = a.view(np.uint32) != b.view(np.uint32); df = fa.view(np.uint32) != fb.view(np.uint32)
    bad = [i for i, (x, y) in enumerate(zip(sa, sb)) if int(x) != int(y)]
    print(f"  frames {n} / {n2}; accumulated differs in {int(da.any(axis=-1).sum())} pixels, FrameRender in {int(df.any(axis=-1).sum())}; "
          f"checkpoints differing: {len(bad)} of {len(sa)} (first {bad[:5]}: {[(run.kinds[i], int(sa[i]), int(sb[i])) for i in bad[:5]]})")
    k = 0
    for i, kind in enumerate(run.kinds):
        if kind.startswith("acc@"):
            if i in bad:
                d = imgs_a[k].view(np.uint32) != run.imgs[k].view(np.uint32)
                ys, xs = np.nonzero(d.any(axis=-1))
                ratio = (imgs_a[k][ys[0], xs[0]] / np.maximum(run.imgs[k][ys[0], xs[0]], 1e-30))
                print(f"  checkpoint {i} {kind}: {len(ys)} pixels differ, rows {ys.min()}..{ys.max()} (distinct rows {len(set(ys.tolist()))}), cols {xs.min()}..{xs.max()}; alpha {imgs_a[k][ys[0], xs[0]][3]} vs {run.imgs[k][ys[0], xs[0]][3]}; ratio {ratio}")
            k += 1
    first = int(run.kinds[bad[0]].split('@')[1])
    print("  events around the first bad checkpoint (round, n, how, event, frames so far):")
    for e in events_a:
        if first - 150 <= e[4] <= first + 80: print("   ", e)
    if da.any():
        ys, xs = np.nonzero(da.any(axis=-1)); print(f"  accumulated: rows {ys.min()}..{ys.max()}, cols {xs.min()}..{xs.max()}; e.g. {a[ys[0], xs[0]]} vs {b[ys[0], xs[0]]}")
print("SOAK", "OK: identical buffers, checkpoints and counters" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
