"""The driver's timed pattern on config 2 (W warm-up RenderFrame() calls, synchronise, K x rt_render_frame, synchronise), alone, for
`rocprofv3 --kernel-trace` timelines (tools/timeline.py).  usage: python tools/k20_pattern.py [K=20] [W=5] [config=2]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = int(sys.argv[3]) if len(sys.argv) > 3 else 2
pkg = g.load_package(); api = pkg.load_library()
tr = api.create_tracer(0)
mgr = pkg.scenes.get(cfg).make_manager(tr, api); mgr.OnEnable(renderSeed=1)
for _ in range(W):
    mgr.RenderFrame()
tr.synchronize()
best = None
for rep in range(3):
    tr.reset_counters(); tr.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        tr.render_frame()
    tr.synchronize()
    dt = time.perf_counter() - t0
    seg = tr.counters()["segments"]
    print(f"rep {rep}: {K} frames in {dt * 1e3:.3f} ms = {dt / K * 1e3:.4f} ms/frame, {seg / dt / 1e6:.0f} Mrays/s")
tr.close()
