"""Extended run of tests/test_ref_pin.py::test_random_model_scenes: the oracle's restatement against the reference's own shader text
compiled as C++ (oracle/_ref/libref.so) on random model scenes (spheres removed: the reference has no sphere buffer), seeds [first, first+n):
FrameRender, AccumulatedRender and the shader's own counters bit for bit.  CPU only.    usage: python tools/ref_fuzz.py [n=200] [first=100]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
import test_ref_pin as T
from test_gpu_fuzz import random_scene

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
pkg = g.load_package(); orc = g.load_oracle(); ref = T.ref_lib.load(pkg)
if ref is None:
    raise SystemExit("oracle/_ref/libref.so absent")
bad = done = 0
t0 = time.time()
for seed in range(first, first + n):
    sc0, render_seed = random_scene(pkg, seed)
    if not sc0.models:
        continue
    try:
        out = T.render_pair(pkg, orc, ref, lambda: random_scene(pkg, seed)[0], sc0.width, sc0.height, sc0.frames, render_seed)
        T.assert_same(out, f"fuzz {seed}")
        done += 1
    except AssertionError as e:
        bad += 1
        print("MISMATCH", seed, str(e)[:200])
print(f"REF FUZZ {'OK' if not bad else 'MISMATCH x %d' % bad}: {done} scenes with models of seeds {first}..{first + n - 1}, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
