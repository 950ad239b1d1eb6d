"""One GPU rendering partition 1/N of the image (cyclic 8-row strips) for N = 1, 2, 4, 8, configs 2 and 5: the per-rank time of an
N-GPU strong-scaling run (the data path has no communication).  usage: python tools/partition_emulation.py  (on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); api = pkg.load_library()
for cfg, frames in ((2, 48), (5, 8)):
    for n in (1, 2, 4, 8):
        tr = api.create_tracer(0); tr.set_partition(8, 0, n)
        sc = pkg.scenes.get(cfg); mgr = sc.make_manager(tr, api); mgr.OnEnable(renderSeed=1)
        for _ in range(3): mgr.RenderFrame()
        tr.synchronize(); tr.reset_counters(); tr.timer_begin()
        for _ in range(frames): tr.render_frame()
        tr.timer_end(); c = tr.counters()
        print(f"config {cfg} partition 1/{n}: {c['gpuMs']/frames:.3f} ms/frame (back-to-back rt_render_frame, coalesced)  {c['segments']/c['gpuMs']/1e3:.0f} Mrays/s per rank -> x{n} ranks = {n*c['segments']/c['gpuMs']/1e3:.0f}")
        tr.close()
