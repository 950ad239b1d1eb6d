"""Emulate one rank of an N-GPU strong-scaling run (partition 0 of N) and sweep the persistent grid size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package(); api = pkg.load_library()
cfg = int(sys.argv[1]); parts = int(sys.argv[2])
tr = api.create_tracer(0); tr.set_partition(8, 0, parts)
mgr = pkg.scenes.get(cfg).make_manager(tr, api); mgr.OnEnable(renderSeed=1)
mgr.RenderFrames(3); tr.synchronize()
best = 1e9
for rep in range(3):
    tr.reset_counters(); tr.timer_begin(); tr.render_frames(5); tr.timer_end(); c = tr.counters()
    best = min(best, c["gpuMs"] / 5)
print(f"config {cfg} part 0/{parts} grid={os.environ.get('RT_GRID','auto')}: {best:.3f} ms/frame  {c['segments']/5/best/1e3:.1f} Mrays/s")
