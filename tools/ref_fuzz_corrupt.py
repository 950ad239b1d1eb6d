"""The oracle against the reference's own text on scenes NO builder makes: real scenes with node / model fields overwritten (child indices anywhere inside
the buffer, leaf ranges moved or overlapping, nodes shared between subtrees, models pointed at other meshes' nodes) that rt_validate_scene ACCEPTS.
What the library accepts it must render like the reference (RC:234-287 resolves a child as the model's nodeOffset + startIndex and a leaf as the model's
triOffset + startIndex); tests/test_layout.py checks that the device records of such scenes are the caller's tree, this tool that the oracle's reading
of them is the reference's: FrameRender, AccumulatedRender and the shader's counters bit for bit.  CPU only.
usage: python tools/ref_fuzz_corrupt.py [n=300] [seed=1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
import test_ref_pin as T

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pkg = g.load_package(); api = pkg.load_library(); orc = g.load_oracle(); ref = T.ref_lib.load(pkg)
if ref is None:
    raise SystemExit("oracle/_ref/libref.so absent")
rng = np.random.default_rng(seed)
CFGS = [(3, {}), (4, {"subdivisions": 2}), (6, {})]
W, H = 48, 27
done = refused = bad = 0
t0 = time.time()
it = 0
while done < n_cases and it < 50 * n_cases:
    it += 1
    cfg, kw = CFGS[it % len(CFGS)]
    out = []
    corrupted = None
    for lib in (orc, ref):
        sc = pkg.scenes.get(cfg, **kw)
        sc.spheres = []
        tr = lib.create_tracer(T.THREADS)
        mgr = sc.make_manager(tr, orc, W, H)
        mgr.OnEnable(renderSeed=it)
        if corrupted is None:
            d = mgr.CreateAllMeshData(mgr.models)
            m, t, n = d["meshInfo"].copy(), d["triangles"].copy(), d["nodes"].copy()
            for _ in range(int(rng.integers(1, 4))):
                what = int(rng.integers(0, 4))
                i = int(rng.integers(0, len(n)))
                if what == 0:
                    n["startIndex"][i] = int(rng.integers(0, len(n)))
                elif what == 1 and n["triangleCount"][i] > 0:
                    n["startIndex"][i] = int(rng.integers(0, 40))               # a leaf that names other triangles of its mesh
                elif what == 2 and n["triangleCount"][i] > 0:
                    n["triangleCount"][i] = int(rng.integers(1, 9))
                else:
                    j = int(rng.integers(0, len(n)))
                    n["startIndex"][i], n["triangleCount"][i] = n["startIndex"][j], n["triangleCount"][j]   # node i becomes a second parent of j's children / triangles
            try:
                api.validate_scene_arrays(m, t, n, mgr._pack_spheres())
            except pkg.abi.RtError:
                refused += 1
                tr.close()
                break
            corrupted = (m, t, n)
        tr.upload_scene(*corrupted, mgr._pack_spheres())
        for _ in range(2):
            mgr.RenderFrame()
        out.append((tr.read_accumulated(), tr.read_frame(), tr.counters(), tr.frame()))
        tr.close()
    if len(out) < 2:
        continue
    try:
        T.assert_same(out, f"corrupted case {it}")
        done += 1
    except AssertionError as e:
        bad += 1
        done += 1
        print("MISMATCH", it, str(e)[:300])
print(f"REF FUZZ (corrupted, accepted scenes) {'OK' if not bad else 'MISMATCH x %d' % bad}: {done} scenes rendered by both, {refused} refused by rt_validate_scene, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
