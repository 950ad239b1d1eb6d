"""Fuzz of the BVH builders: random meshes (uniform soups, clusters, coplanar sheets, duplicated and degenerate triangles, huge and tiny
coordinates, a few non-finite vertices) of 1 ... N triangles — rt_build_bvh_mt (1 / 5 threads), rt_build_bvh_gpu and random batches
through rt_build_bvh_gpu_batch (one forest) against rt_build_bvh, and against the reference's own BVH.cs compiled as C++
(oracle/_ref/libref_bvh.so) where it travelled: nodes, triangle order and statistics byte for byte, refusals with the same status.
usage: python tools/bvh_fuzz.py [meshes=200] [max_tris=6000] [seed=1]        (GPU parts are skipped without a device)"""
import importlib.util, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g

n_meshes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
max_tris = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
pkg = g.load_package(); api = pkg.load_library(); a = pkg.abi
spec = importlib.util.spec_from_file_location("rt_ref_lib", os.path.join(g.ROOT, "oracle", "ref_lib.py"))
ref_lib = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref_lib)
ref = ref_lib.load_bvh(pkg)
have_gpu = os.path.exists("/dev/kfd")
rng = np.random.default_rng(seed)


def random_mesh(k):
    kind = k % 8
    nt = int(rng.integers(1, max_tris)) if k % 11 else int(rng.integers(1, 12))
    nv = max(3, int(nt * rng.uniform(0.3, 1.5)))
    if kind == 0:
        v = rng.uniform(-1, 1, (nv, 3))
    elif kind == 1:   # clusters
        c = rng.uniform(-5, 5, (8, 3)); v = c[rng.integers(0, 8, nv)] + rng.normal(0, 0.05, (nv, 3))
    elif kind == 2:   # coplanar sheet (one zero-size axis)
        v = rng.uniform(-1, 1, (nv, 3)); v[:, int(rng.integers(0, 3))] = 0.25
    elif kind == 3:   # few distinct vertices: many identical triangles / centroids
        v = rng.uniform(-1, 1, (max(3, nv // 50), 3)); nv = len(v)
    elif kind == 4:   # huge coordinates (costs near overflow)
        v = rng.uniform(-1, 1, (nv, 3)) * 10.0 ** rng.uniform(10, 19)
    elif kind == 5:   # tiny coordinates (denormal extents)
        v = rng.uniform(-1, 1, (nv, 3)) * 1e-30
    elif kind == 6:   # a line
        t = rng.uniform(-1, 1, nv); v = np.stack([t, 2 * t, -t], axis=1)
    else:             # a few non-finite vertices
        v = rng.uniform(-1, 1, (nv, 3)); v[rng.integers(0, nv, 2), rng.integers(0, 3, 2)] = [np.inf, np.nan]
    v = v.astype(np.float32)
    idx = rng.integers(0, nv, 3 * nt).astype(np.int32)
    nrm = rng.normal(0, 1, (nv, 3)).astype(np.float32)
    return v, nrm, idx


def build(fn, *args):
    try:
        n, t, s = fn(*args)
        s = dict(s); s.pop("timeMs", None)
        return ("ok", n.tobytes(), t.tobytes(), tuple(sorted(s.items())))
    except a.RtError as e:
        return ("err", e.status)


bad = 0
meshes, wants = [], []
for k in range(n_meshes):
    v, nrm, idx = random_mesh(k)
    q = int(rng.integers(0, 3))
    want = build(api.build_bvh_arrays, v, nrm, idx, q)
    got = {"mt1": build(api.build_bvh_arrays_mt, v, nrm, idx, q, 1), "mt5": build(api.build_bvh_arrays_mt, v, nrm, idx, q, 5)}
    if ref is not None:
        got["BVH.cs"] = build(ref.build_bvh_arrays, v, nrm, idx, q)
    if have_gpu:
        got["gpu"] = build(api.build_bvh_arrays_gpu, v, nrm, idx, q)
    for name, r in got.items():
        if r != want:
            bad += 1
            print(f"MISMATCH mesh {k} (kind {k % 8}, {len(idx) // 3} triangles, quality {q}): {name} {r[0]} {r[1] if r[0] == 'err' else ''} vs host {want[0]} {want[1] if want[0] == 'err' else ''}")
    if q == 1:
        meshes.append((v, nrm, idx)); wants.append(want)
if have_gpu:   # random batches of the quality-1 meshes as forests
    for b in range(max(4, n_meshes // 10)):
        pick = sorted(rng.choice(len(meshes), size=int(rng.integers(2, min(12, len(meshes)))), replace=False).tolist())
        try:
            nd, tr, per = api.build_bvh_arrays_gpu_batch([meshes[i] for i in pick], 1)
        except a.RtError as e:
            st = [w for i, w in enumerate(wants) if i in pick and w[0] == "err"]
            if not st or st[0][1] != e.status:
                bad += 1; print(f"MISMATCH batch {b} {pick}: refused with {e.status}, single builds say {[w[0] for i, w in enumerate(wants) if i in pick]}")
            continue
        for i, (noff, toff, stats) in zip(pick, per):
            w = wants[i]
            s = dict(stats); s.pop("timeMs", None)
            nn = len(w[1]) // a.node_dtype.itemsize if w[0] == "ok" else 0
            ok = w[0] == "ok" and nd[noff:noff + nn].tobytes() == w[1] and tr[toff:toff + len(meshes[i][2]) // 3].tobytes() == w[2] and tuple(sorted(s.items())) == w[3]
            if not ok:
                bad += 1; print(f"MISMATCH batch {b} mesh {i}: forest differs from the single build ({w[0]})")
print(f"BVH FUZZ {'OK' if not bad else 'MISMATCH x %d' % bad}: {n_meshes} meshes (refused by all builders alike: {sum(1 for w in wants if w[0] == 'err')} of the {len(wants)} quality-1 ones), gpu {have_gpu}, BVH.cs {ref is not None}")
sys.exit(1 if bad else 0)
