"""Ad-hoc GPU check: HIP vs oracle parity on small renders + full-size timing."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g

pkg = g.load_package(); api = pkg.load_library(); orc = g.load_oracle()
print(api.version())

def render(lib, tr, cfg, w, h, frames, stats=False):
    sc = pkg.scenes.get(cfg)
    mgr = sc.make_manager(tr, lib, w, h)
    mgr.OnEnable(renderSeed=1)
    if stats and hasattr(tr, "enable_stats"): tr.enable_stats(True)
    tr.reset_counters()
    t = time.time(); mgr.RenderFrames(frames); acc = tr.read_accumulated(); dt = time.time() - t
    return acc, tr.counters(), dt

for cfg, (w, h), frames in ((1, (256, 256), 2), (2, (240, 136), 2), (3, (240, 136), 2), (4, (160, 90), 1)):
    kw = dict(subdivisions=4) if cfg == 4 else {}
    a, ca, ta = render(api, api.create_tracer(0), cfg, w, h, frames, stats=True)
    b, cb, tb = render(orc, orc.create_tracer(8), cfg, w, h, frames)
    same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
    nbad = int(np.sum(np.any(a.view(np.uint32) != b.view(np.uint32), axis=-1)))
    keys = ["segments", "innerSteps", "leafSteps", "triTests", "sphereTests", "modelVisits", "pixelFrames"]
    print(f"config {cfg} {w}x{h}x{frames}: bit-identical={same} bad_pixels={nbad}/{w*h} counters_equal={all(ca[k]==cb[k] for k in keys)} gpu {ta:.3f}s cpu {tb:.3f}s")
    if not same:
        d = np.abs(a - b); print("   max abs diff", np.nanmax(d), "first bad", np.argwhere(np.any(a.view(np.uint32) != b.view(np.uint32), axis=-1))[:5])
    if not all(ca[k]==cb[k] for k in keys): print("  ", ca, cb)

# full-size timing, config 2 and 3
for cfg, frames in ((2, 5), (3, 3)):
    tr = api.create_tracer(0)
    sc = pkg.scenes.get(cfg); mgr = sc.make_manager(tr, api); mgr.OnEnable(renderSeed=1)
    mgr.RenderFrames(1); tr.synchronize()
    tr.reset_counters(); tr.timer_begin(); mgr.RenderFrames(frames); tr.timer_end()
    c = tr.counters()
    print(f"config {cfg} full size: {frames} frames gpuMs={c['gpuMs']:.2f} segments={c['segments']} Mrays/s={c['segments']/c['gpuMs']/1e3:.1f}")
