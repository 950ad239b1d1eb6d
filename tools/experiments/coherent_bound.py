"""The no-divergence ceiling: Mrays/s when all 64 lanes of every wave trace the same pixel (identical chains, lane
utilisation 1 in every phase) against the shipped schedule, same scenes, same library build (make coherent).
A coherent frame does 64x the work per pixel (4 frames per launch there).   usage: python tools/coherent_bound.py [configs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
os.environ["RT_HIP_LIB"] = os.path.join(g.PKG_DIR, "lib", "libraytrace_hip_coherent.so")
pkg = g.load_package(); api = pkg.load_library()
cfgs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "2,3,4,6").split(",")]
for cfg in cfgs:
    row = {}
    for coh in (0, 1):
        os.environ["RT_DEBUG_COHERENT"] = str(coh)
        sc = pkg.scenes.get(cfg)
        w, h = sc.width, sc.height   # full size either way: the queue must hold many more items than the chip holds waves
        tr = api.create_tracer(0)
        mgr = sc.make_manager(tr, api, w, h); mgr.OnEnable(renderSeed=1)
        mgr.RenderFrames(1); tr.synchronize()
        best = 0.0
        for rep in range(1 if coh else 3):
            tr.reset_counters(); tr.timer_begin(); tr.render_frames(4 if coh else 16); tr.timer_end()
            c = tr.counters()
            lanes = 64 if coh else 1
            best = max(best, c["segments"] / c["gpuMs"] / 1e3)
        row[coh] = best
        tr.close()
    print(f"config {cfg}: shipped {row[0]:9.1f} Mrays/s   all lanes coherent {row[1]:9.1f} Mrays/s   ceiling / shipped = x{row[1] / row[0]:.2f}")
