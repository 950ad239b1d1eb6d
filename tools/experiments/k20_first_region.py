"""The driver's region (5 warm-up frames, then 20 timed) in fresh contexts: the FIRST region of a context against the ones after it.
usage: python tools/experiments/k20_first_region.py [trials=4]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package(); api = pkg.load_library()
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 4
WARM = int(os.environ.get("K20_WARMUP", "5"))
K = 20
for trial in range(trials):
    sc = pkg.scenes.get(2)
    tr = api.create_tracer(0)
    mgr = sc.make_manager(tr, api)
    mgr.bvhOnGpu = True
    mgr.OnEnable(renderSeed=1)
    tr.synchronize()
    for _ in range(WARM):
        mgr.RenderFrame()
    tr.synchronize()
    ms = []
    for rep in range(4):
        t0 = time.perf_counter()
        for _ in range(K):
            tr.render_frame()
        tr.synchronize()
        ms.append(1e3 * (time.perf_counter() - t0))
    print(f"trial {trial}: warm-up {WARM} frames, regions of {K} frames: " + "  ".join(f"{m:6.2f} ms" for m in ms), flush=True)
    tr.close()
