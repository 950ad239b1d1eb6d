"""Scenes with more models than one 64-bit candidate mask (the MANY kernel instantiation): ms per 960x540 frame.
usage: python tools/many_models_bench.py [model counts]      (RT_HIP_LIB selects the library)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as g
pkg = g.load_package(); api = pkg.load_library()
import test_gpu_fuzz as tf
for n in [int(a) for a in (sys.argv[1:] or ["100", "200", "333", "1000"])]:
    sc = tf.crowded_scene(pkg, n, 3, seed=21)
    tr = api.create_tracer(0)
    mgr = sc.make_manager(tr, api, 960, 540); mgr.OnEnable(renderSeed=5)
    mgr.RenderFrames(2); tr.synchronize()
    best = 1e9
    for _ in range(3):
        tr.reset_counters(); tr.timer_begin(); tr.render_frames(16); tr.timer_end()
        c = tr.counters(); best = min(best, c["gpuMs"] / 16)
    print(f"{n:5d} models: {best:7.3f} ms/frame   {c['segments'] / 16 / best / 1e3:8.1f} Mrays/s")
    tr.close()
