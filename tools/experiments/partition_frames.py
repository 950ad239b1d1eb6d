"""One GPU rendering partition 1/N of config C with rt_render_frames(k) for k = 1, 2, 4, 8, 16 frames per launch.
usage: python tools/partition_frames.py <config> <N>   (on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package(); api = pkg.load_library()
cfg, n = int(sys.argv[1]), int(sys.argv[2])
tr = api.create_tracer(0); tr.set_partition(8, 0, n)
sc = pkg.scenes.get(cfg); mgr = sc.make_manager(tr, api); mgr.OnEnable(renderSeed=1)
mgr.RenderFrames(20); tr.synchronize()
for k in (1, 2, 4, 8, 16):
    tr.reset_counters(); tr.timer_begin()
    for _ in range(4): tr.render_frames(k)
    tr.timer_end(); c = tr.counters()
    print(f"config {cfg} partition 1/{n}: rt_render_frames({k}) x4: {c['gpuMs']/(4*k):.3f} ms/frame, {c['gpuMs']/4:.3f} ms/launch, {c['segments']/(4*k):.3e} seg/frame")
tr.close()
