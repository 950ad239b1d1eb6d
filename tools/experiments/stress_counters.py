#!/usr/bin/env python
"""Stress the reset_counters -> render -> counters sequence with P processes sharing one GPU.
Every iteration must count exactly the same segments (the frame index is pinned); prints the number of deviating iterations.
    python tools/stress_counters.py [procs] [iters]       (RT_HIP_LIB selects the library)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def child(iters):
    import __graft_entry__ as g
    pkg = g.load_package(); api = pkg.load_library()
    tr = api.create_tracer(0)
    sc = pkg.scenes.get(2)
    mgr = sc.make_manager(tr, api, 960, 540)
    mgr.OnEnable(renderSeed=1)
    want, bad = None, 0
    for i in range(iters):
        mgr.numAccumulatedFrames = 1
        mgr.SetShaderParams()
        tr.reset_counters()
        tr.render_frames(3)
        got = tr.counters()["segments"]
        if want is None: want = got
        elif got != want:
            bad += 1
            print(f"pid {os.getpid()} iter {i}: {got} != {want}", flush=True)
    print(f"pid {os.getpid()}: {bad} of {iters} iterations deviate", flush=True)
    tr.close()

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(int(sys.argv[2]))
    else:
        procs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
        iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "child", str(iters)]) for _ in range(procs)]
        sys.exit(max(p.wait() for p in ps))
