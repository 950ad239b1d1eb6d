"""Quick GPU benchmark of several configs: Mrays/s + ms/frame (device events), golden check first.
usage: python tools/qb.py [configs, default 2,3,4] [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g

pkg = g.load_package(); api = pkg.load_library()
cfgs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "2,3,4").split(",")]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 5

# correctness gate: golden fixtures through the HIP path
import importlib.util
spec = importlib.util.spec_from_file_location("mg", os.path.join(g.ROOT, "tests", "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
bad = []
for case in sorted(mg.CASES):
    z = np.load(os.path.join(g.ROOT, "tests", "golden", case + ".npz"))
    tr = api.create_tracer(0); tr.enable_stats(True)
    acc, cnt = mg.render_case(pkg, api, tr, case); tr.close()
    if not (np.array_equal(acc.view(np.uint32), z["accumulated"].view(np.uint32)) and cnt.tolist() == z["counters"].tolist()):
        bad.append(case)
print("golden:", "ALL BIT-EXACT" if not bad else "MISMATCH " + str(bad))

for cfg in cfgs:
    t0 = time.time()
    tr = api.create_tracer(0)
    sc = pkg.scenes.get(cfg); mgr = sc.make_manager(tr, api); mgr.OnEnable(renderSeed=1)
    tb = time.time() - t0
    mgr.RenderFrames(1); tr.synchronize()
    best = None
    for rep in range(3):
        tr.reset_counters(); tr.timer_begin(); tr.render_frames(frames); tr.timer_end()
        c = tr.counters()
        ms = c["gpuMs"] / frames
        if best is None or ms < best[0]: best = (ms, c["segments"] / c["gpuMs"] / 1e3)
    print(f"config {cfg}: {best[0]:8.3f} ms/frame  {best[1]:9.1f} Mrays/s   (segments/frame {c['segments']/frames:.3e}, setup {tb:.1f}s)")
    tr.close()

if os.environ.get("RT_PHASES"):
    for cfg in cfgs:
        tr = api.create_tracer(0)
        sc = pkg.scenes.get(cfg); mgr = sc.make_manager(tr, api); mgr.OnEnable(renderSeed=1)
        nph = int(os.environ.get("RT_PHASE_FRAMES", "1"))  # frames in the profiled launch (1 = the drain of a single-frame launch included)
        tr.enable_stats(True); tr.reset_counters(); mgr.RenderFrames(nph)
        c = tr.counters(); ph = tr.phase_profile()
        print(f"config {cfg} phase profile ({nph} frame(s) in one launch, {c['segments']} segments):")
        inner_lanes = ph["inner"][1] or 1
        if not ph["inner"][1]:   # no trees: the FLAT kernel, whose counter in that slot is the chain pool's
            v = ph.pop("inner_from_lds_cache", (0, 0))[0]
            print(f"   chains handed over through the workgroup's LDS pool: {v} = {v / max(1, ph['loop'][0]):.1f} per wave iteration")
            ph.pop("inner_on_one_node_48_lanes", None); ph.pop("inner_on_one_node_3_of_4_active", None)
        for key, label in (("inner_from_lds_cache", "inner steps served by the LDS top-of-tree cache"),
                           ("inner_on_one_node_48_lanes", "inner steps with >= 48 lanes of the wave on ONE node"),
                           ("inner_on_one_node_3_of_4_active", "inner steps with >= 3/4 of >= 16 active lanes on ONE node")):
            if key not in ph: continue
            v = ph.pop(key, (0, 0))[0]
            print(f"   {label}: {v} lane-steps = {v / inner_lanes:.3f} of all inner lane-steps")
        for k, (e, l) in ph.items():
            if k == 'filter_violations': print('   filter_violations', e); continue
            if e: print(f"   {k:14s} wave-execs {e:12d}  lanes {l:13d}  util {l/(64*e):.3f}  execs/segment*64 {e*64/c['segments']:.2f}")
        tr.close()
