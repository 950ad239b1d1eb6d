"""Extended run of tests/test_ref_pin.py::test_random_model_scenes: the oracle's restatement against the reference's own shader text
compiled as C++ (oracle/_ref/libref.so) on random model scenes (spheres removed: the reference has no sphere buffer), seeds [first, first+n):
FrameRender, AccumulatedRender and the shader's own counters bit for bit.  `--spheres`: the same generator WITH its spheres against
libref_spheres.so (the reference's text + the declared sphere hook, make_ref.py S1; test_random_scenes_with_spheres_and_models extended).
CPU only.    usage: python tools/ref_fuzz.py [--spheres] [n=200] [first=100]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
import test_ref_pin as T
from test_gpu_fuzz import random_scene

argv = [a for a in sys.argv[1:] if a != "--spheres"]
with_spheres = "--spheres" in sys.argv[1:]
n = int(argv[0]) if len(argv) > 0 else 200
first = int(argv[1]) if len(argv) > 1 else 100
pkg = g.load_package(); orc = g.load_oracle(); ref = T.ref_lib.load(pkg, "spheres") if with_spheres else T.ref_lib.load(pkg)
if ref is None:
    raise SystemExit("oracle/_ref/libref%s.so absent" % ("_spheres" if with_spheres else ""))
render = T.render_pair_with_spheres if with_spheres else T.render_pair
bad = done = 0
t0 = time.time()
for seed in range(first, first + n):
    sc0, render_seed = random_scene(pkg, seed)
    if not sc0.models and not (with_spheres and sc0.spheres):
        continue
    try:
        out = render(pkg, orc, ref, lambda: random_scene(pkg, seed)[0], sc0.width, sc0.height, sc0.frames, render_seed)
        T.assert_same(out, f"fuzz {seed}")
        done += 1
    except AssertionError as e:
        bad += 1
        print("MISMATCH", seed, str(e)[:200])
print(f"REF FUZZ {'OK' if not bad else 'MISMATCH x %d' % bad}: {done} scenes {'with spheres and models' if with_spheres else 'with models'} of seeds {first}..{first + n - 1}, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
