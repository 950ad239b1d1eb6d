import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
pkg = g.load_package(); api = pkg.load_library()
cfg, frames = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 48
for n in (4, 8):
    tr = api.create_tracer(0); tr.set_partition(8, 0, n)
    sc = pkg.scenes.get(cfg); mgr = sc.make_manager(tr, api); mgr.OnEnable(renderSeed=1)
    for _ in range(3): mgr.RenderFrame()
    tr.synchronize(); tr.reset_counters(); tr.timer_begin()
    for _ in range(frames): tr.render_frame()
    tr.timer_end(); c = tr.counters()
    print(f"config {cfg} partition 1/{n}: {c['gpuMs']/frames:.4f} ms/frame  {c['segments']/c['gpuMs']/1e3:.0f} Mrays/s per rank")
    tr.close()
