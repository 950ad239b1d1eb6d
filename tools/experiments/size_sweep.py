import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package(); api = pkg.load_library()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for (w, h) in ((480, 270), (960, 540), (1920, 1080), (3840, 2160)):
    tr = api.create_tracer(0)
    mgr = pkg.scenes.get(cfg).make_manager(tr, api, w, h); mgr.OnEnable(renderSeed=1)
    mgr.RenderFrames(1); tr.synchronize()
    best = 1e9
    for rep in range(3):
        tr.reset_counters(); tr.timer_begin(); tr.render_frames(3); tr.timer_end(); c = tr.counters()
        best = min(best, c["gpuMs"] / 3)
    print(f"config {cfg} {w}x{h}: {best:.3f} ms/frame  {c['segments']/3/best/1e3:.1f} Mrays/s  ({best*1e6/(w*h):.1f} ns/pixel)")
    tr.close()
