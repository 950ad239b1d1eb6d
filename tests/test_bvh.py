"""BVH: the product's host builder (rt_build_bvh) emits exactly the tree of the
literal restatement of BVH.cs in the oracle, and the oracle's traversal finds the
same closest hit as brute force over all triangles (property test)."""
import ctypes as C

import os

import numpy as np
import pytest


def both(api, orc, mesh, quality):
    a = api.build_bvh_arrays(mesh.vertices, mesh.normals, mesh.triangles, quality)
    b = orc.build_bvh_arrays(mesh.vertices, mesh.normals, mesh.triangles, quality)
    return a, b


def mesh_cases(pkg):
    m = pkg.meshes
    rng = np.random.default_rng(7)
    # triangle soup: overlapping random triangles (ragged leaves, SAH refusing splits)
    v = rng.uniform(-1, 1, (300, 3)).astype(np.float32)
    soup = m.Mesh(v, np.tile([[0, 1, 0]], (300, 1)), rng.integers(0, 300, 3 * 500).astype(np.int32), "soup")
    # degenerate: every triangle identical (all centroids equal -> single leaf > 127 triangles)
    same = m.Mesh(v[:3], np.tile([[0, 1, 0]], (3, 1)), np.tile([0, 1, 2], 200).astype(np.int32), "same")
    # flat mesh in the y=0 plane (a zero-size axis: NaN in CeilToInt(axisSize/maxAxis*K) cannot occur, 0 can)
    flat_v = np.stack([rng.uniform(-1, 1, 200), np.zeros(200), rng.uniform(-1, 1, 200)], axis=1).astype(np.float32)
    flat = m.Mesh(flat_v, np.tile([[0, 1, 0]], (200, 1)), rng.integers(0, 200, 3 * 150).astype(np.int32), "flat")
    # a single point (all sizes 0 -> 0/0 NaN in the split count)
    point = m.Mesh(np.zeros((3, 3), np.float32), np.tile([[0, 1, 0]], (3, 1)), np.tile([0, 1, 2], 12).astype(np.int32), "point")
    one = m.Mesh(v[:3], np.tile([[0, 1, 0]], (3, 1)), np.array([0, 1, 2], np.int32), "one")
    return [m.quad(), m.cube(), m.rounded_cube(6), m.icosphere(3), m.icosphere(3, 1.0, 4), soup, same, flat, point, one]


@pytest.mark.parametrize("quality", [0, 1, 2])
def test_product_builder_equals_oracle_builder(pkg, api, orc, quality):
    for mesh in mesh_cases(pkg):
        (n1, t1, s1), (n2, t2, s2) = both(api, orc, mesh, quality)
        assert n1.tobytes() == n2.tobytes(), (mesh.name, quality)
        assert t1.tobytes() == t2.tobytes(), (mesh.name, quality)
        s1.pop("timeMs"), s2.pop("timeMs")
        assert s1 == s2, (mesh.name, quality)


def test_empty_mesh_builds_a_single_empty_leaf(pkg, api, orc):
    mesh = pkg.meshes.Mesh(np.zeros((3, 3), np.float32), np.zeros((3, 3), np.float32), np.zeros(0, np.int32), "empty")
    (n1, t1, _), (n2, t2, _) = both(api, orc, mesh, 1)
    assert len(n1) == len(n2) == 1 and len(t1) == len(t2) == 0
    assert n1[0]["triangleCount"] == 0 and n1.tobytes() == n2.tobytes()


def test_overflow_scale_coordinates_are_refused_not_overrun(pkg, api, orc):
    """Coordinates around 1e19+: node areas overflow to inf, every split cost is inf, and BVH.cs's
    (axis 0, pos 0) fallback peels off an empty child per level down to MaxDepth — 65 nodes for 3
    triangles against a documented out_nodes capacity of 6, with 0-triangle leaves the shader would
    read as inner nodes.  Both builders must refuse (RT_ERR_SCENE) without writing past the capacity."""
    rng = np.random.default_rng(3)
    v = (rng.uniform(-1, 1, (9, 3)) * 3e19).astype(np.float32)
    mesh = pkg.meshes.Mesh(v, np.tile([[0, 1, 0]], (9, 1)), np.arange(9, dtype=np.int32), "huge")
    for lib in (api, orc):
        with pytest.raises(pkg.abi.RtError) as e:
            lib.build_bvh_arrays(mesh.vertices, mesh.normals, mesh.triangles, pkg.abi.BVH_QUALITY_HIGH)
        assert e.value.status == pkg.abi.RT_ERR_SCENE
        try:  # Low quality: a tree within the capacity, or the same refusal
            n, _, st = lib.build_bvh_arrays(mesh.vertices, mesh.normals, mesh.triangles, pkg.abi.BVH_QUALITY_LOW)
            assert len(n) <= 6 and st["leafMinTriCount"] > 0
        except pkg.abi.RtError as err:
            assert err.status == pkg.abi.RT_ERR_SCENE
    if hasattr(api, "build_bvh_arrays_mt"):
        with pytest.raises(pkg.abi.RtError):
            api.build_bvh_arrays_mt(mesh.vertices, mesh.normals, mesh.triangles, 1, 4)
    # the same mesh at a sane scale builds
    small = pkg.meshes.Mesh(v / np.float32(3e19), mesh.normals, mesh.triangles, "small")
    (n1, _, _), (n2, _, _) = both(api, orc, small, 1)
    assert n1.tobytes() == n2.tobytes() and len(n1) <= 6


def test_tree_structure_invariants(pkg, api):
    """What the kernel relies on: children adjacent (BVH:161-162), inner root keeps
    triangleCount -1 (BVH:61), inner non-root 0, leaves partition the triangle range,
    depth <= 32, child bounds inside parent bounds."""
    mesh = pkg.meshes.icosphere(4, 1.0, 9)
    nodes, tris, stats = api.build_bvh_arrays(mesh.vertices, mesh.normals, mesh.triangles, 1)
    assert nodes[0]["triangleCount"] == -1 and nodes[0]["startIndex"] == 1
    covered = np.zeros(len(tris), dtype=int)
    stack = [(0, 0)]
    maxdepth = 0
    while stack:
        i, d = stack.pop()
        n = nodes[i]
        maxdepth = max(maxdepth, d)
        if n["triangleCount"] > 0:
            covered[n["startIndex"]: n["startIndex"] + n["triangleCount"]] += 1
            t = tris[n["startIndex"]: n["startIndex"] + n["triangleCount"]]
            pts = np.concatenate([t["posA"], t["posB"], t["posC"]])
            assert np.all(pts.min(0) == n["boundsMin"]) and np.all(pts.max(0) == n["boundsMax"])
        else:
            assert i == 0 or n["triangleCount"] == 0
            a = n["startIndex"]
            assert a % 2 == 1  # pairs sit at (odd, even) indices: node 0 is the root
            for c in (a, a + 1):
                assert np.all(nodes[c]["boundsMin"] >= n["boundsMin"]) and np.all(nodes[c]["boundsMax"] <= n["boundsMax"])
                stack.append((c, d + 1))
    assert np.all(covered == 1)
    assert maxdepth == stats["leafDepthMax"] <= 32
    assert stats["triangleCount"] == len(tris) == mesh.triangle_count
    assert stats["totalNodeCount"] == len(nodes)


def _scene_tracer(pkg, orc, cfg, **kw):
    tr = orc.create_tracer(1)
    sc = pkg.scenes.get(cfg, **kw)
    mgr = sc.make_manager(tr, orc, 32, 18)
    mgr.OnEnable(renderSeed=1)
    return tr, sc


@pytest.mark.parametrize("cfg,kw", [(2, {}), (3, {}), (4, {"subdivisions": 3})])
def test_bvh_traversal_equals_bruteforce(pkg, orc, cfg, kw):
    tr, sc = _scene_tracer(pkg, orc, cfg, **kw)
    rng = np.random.default_rng(cfg)
    n = 3000
    o = rng.uniform(-2.5, 2.5, (n, 3)) + np.array([0, 2, 0])
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # axis-parallel rays exercise invDir = inf
    d[:60] = np.eye(3)[rng.integers(0, 3, 60)] * rng.choice([-1, 1], (60, 1))
    hits = 0
    for i in range(n):
        a = (C.c_float * 10)()
        b = (C.c_float * 2)()
        oo = (C.c_float * 3)(*o[i])
        dd = (C.c_float * 3)(*d[i])
        orc.ray_collision(tr.h, oo, dd, a)
        orc.ray_collision_bruteforce(tr.h, oo, dd, b)
        assert a[0] == b[0]
        if a[0]:
            hits += 1
            assert a[2] == b[1], (i, a[2], b[1])
    assert hits > n // 4
    tr.close()


@pytest.mark.parametrize("threads", [1, 2, 3, 7, 16])
def test_parallel_builder_is_byte_identical_for_any_thread_count(pkg, api, orc, threads):
    """rt_build_bvh_mt: chunk-ordered sweeps + privately built subtrees numbered afterwards in the
    reference's allocation order -> the same bytes as the literal single-thread restatement."""
    for mesh, q in ((pkg.meshes.icosphere(5, 1.0, 11), 1), (pkg.meshes.icosphere(5, 1.0, 12), 0), (pkg.meshes.rounded_cube(40), 1)):
        assert mesh.triangle_count > 8192  # above the threshold where threads are used at all
        n0, t0, s0 = orc.build_bvh_arrays(mesh.vertices, mesh.normals, mesh.triangles, q)
        n1, t1, s1 = api.build_bvh_arrays_mt(mesh.vertices, mesh.normals, mesh.triangles, q, threads)
        assert n0.tobytes() == n1.tobytes() and t0.tobytes() == t1.tobytes()
        s0.pop("timeMs"), s1.pop("timeMs")
        assert s0 == s1


@pytest.mark.gpu
@pytest.mark.parametrize("quality", [0, 1, 2])
def test_gpu_builder_equals_host_builder(pkg, api, quality):
    """rt_build_bvh_gpu (level-synchronous: ordered chunk reductions, prefix-sum + pointer-jumping partition,
    pre-order numbering) emits the host builder's bytes: nodes, triangle order, statistics."""
    cases = mesh_cases(pkg) + [pkg.meshes.icosphere(5, 1.0, 3), pkg.meshes.rounded_cube(14)]
    empty = pkg.meshes.Mesh(np.zeros((3, 3), np.float32), np.zeros((3, 3), np.float32), np.zeros(0, np.int32), "empty")
    for mesh in cases + [empty]:
        n1, t1, s1 = api.build_bvh_arrays(mesh.vertices, mesh.normals, mesh.triangles, quality)
        n2, t2, s2 = api.build_bvh_arrays_gpu(mesh.vertices, mesh.normals, mesh.triangles, quality)
        assert len(n1) == len(n2), (mesh.name, quality, len(n1), len(n2))
        assert n1.tobytes() == n2.tobytes(), (mesh.name, quality)
        assert t1.tobytes() == t2.tobytes(), (mesh.name, quality)
        s1.pop("timeMs"), s2.pop("timeMs")
        assert s1 == s2, (mesh.name, quality, s1, s2)


@pytest.mark.gpu
def test_gpu_builder_big_meshes_and_refusal(pkg, api):
    """81,920 and 327,680 triangles (the BASELINE config 4/5 mesh class and 4x that) byte-identical; input whose
    reference-shaped tree is malformed is refused like on the host."""
    import time
    for sub, seed in ((6, 4), (7, 9)):
        mesh = pkg.meshes.icosphere(sub, 1.0, seed)
        t0 = time.perf_counter()
        n1, t1, s1 = api.build_bvh_arrays(mesh.vertices, mesh.normals, mesh.triangles, 1)
        t1s = time.perf_counter()
        n2, t2, s2 = api.build_bvh_arrays_gpu(mesh.vertices, mesh.normals, mesh.triangles, 1)
        t2s = time.perf_counter()
        assert n1.tobytes() == n2.tobytes() and t1.tobytes() == t2.tobytes()
        print(f"{len(t1)} triangles: host (threads auto) {1e3 * (t1s - t0):.1f} ms, gpu {1e3 * (t2s - t1s):.1f} ms (builder-internal {s1['timeMs']:.1f} / {s2['timeMs']:.1f})")
    rng = np.random.default_rng(3)
    v = (rng.uniform(-1, 1, (9, 3)) * 3e19).astype(np.float32)
    with pytest.raises(pkg.abi.RtError) as e:
        api.build_bvh_arrays_gpu(v, np.tile([[0, 1, 0]], (9, 1)), np.arange(9, dtype=np.int32), 1)
    assert e.value.status == pkg.abi.RT_ERR_SCENE


@pytest.mark.gpu
def test_gpu_builder_leaves_the_callers_device_state_alone(pkg, api):
    """rt_build_bvh_gpu keeps its scratch between calls; rt_build_bvh_gpu_release frees it, a build after that works
    again, bad input leaves 0 nodes behind, and every error path reports n_nodes = 0."""
    import ctypes as C
    mesh = pkg.meshes.icosphere(3, 1.0, 2)
    n0, t0, _ = api.build_bvh_arrays(mesh.vertices, mesh.normals, mesh.triangles, 1)
    for _ in range(2):
        n1, t1, _ = api.build_bvh_arrays_gpu(mesh.vertices, mesh.normals, mesh.triangles, 1)
        assert n0.tobytes() == n1.tobytes() and t0.tobytes() == t1.tobytes()
        api.build_bvh_gpu_release()
    # an index out of range: refused, n_nodes stays 0
    bad = mesh.triangles.copy()
    bad[5] = len(mesh.vertices) + 3
    nodes = np.zeros(2 * len(bad) // 3, dtype=pkg.abi.node_dtype)
    tris = np.zeros(len(bad) // 3, dtype=pkg.abi.triangle_dtype)
    nn = C.c_int(77)
    rc = api.build_bvh_gpu(0, mesh.vertices.ctypes.data, mesh.normals.ctypes.data, len(mesh.vertices), bad.ctypes.data, len(bad), 1,
                           nodes.ctypes.data, C.byref(nn), tris.ctypes.data, None)
    assert rc == pkg.abi.RT_ERR_INVALID_ARG and nn.value == 0


@pytest.mark.gpu
def test_gpu_batch_builder_writes_the_concatenated_arrays(pkg, api):
    """rt_build_bvh_gpu_batch: the meshes of a scene in one call; nodes and triangles come out concatenated exactly as
    CreateAllMeshData (RCM:206-236) concatenates per-mesh host builds, with each mesh's nodeOffset / triOffset; builds from
    several threads share the one scratch pool and still give the same bytes."""
    import threading
    meshes = [pkg.meshes.cube(), pkg.meshes.icosphere(4, 1.0, 7), pkg.meshes.quad(), pkg.meshes.rounded_cube(8), pkg.meshes.icosphere(5, 1.0, 2)]
    for quality in (0, 1, 2):
        nd, tr, per = api.build_bvh_arrays_gpu_batch([(m.vertices, m.normals, m.triangles) for m in meshes], quality)
        no = to = 0
        for m, (noff, toff, stats) in zip(meshes, per):
            n1, t1, s1 = api.build_bvh_arrays(m.vertices, m.normals, m.triangles, quality)
            assert (noff, toff) == (no, to)
            assert nd[no:no + len(n1)].tobytes() == n1.tobytes() and tr[to:to + len(t1)].tobytes() == t1.tobytes(), (m.name, quality)
            s1.pop("timeMs"), stats.pop("timeMs")
            assert s1 == stats
            no, to = no + len(n1), to + len(t1)
        assert len(nd) == no and len(tr) == to
    # the forest path with every awkward mesh class at once (soup, identical triangles, flat, point, one triangle), in both orders,
    # and the mesh-by-mesh path (RT_BVH_NO_FOREST) for the same batch
    cases = mesh_cases(pkg)
    for order in (cases, cases[::-1]):
        for env in (None, "1"):
            if env:
                os.environ["RT_BVH_NO_FOREST"] = env
            try:
                nd, tr, per = api.build_bvh_arrays_gpu_batch([(m.vertices, m.normals, m.triangles) for m in order], 1)
            finally:
                os.environ.pop("RT_BVH_NO_FOREST", None)
            no = to = 0
            for m, (noff, toff, stats) in zip(order, per):
                n1, t1, s1 = api.build_bvh_arrays(m.vertices, m.normals, m.triangles, 1)
                assert (noff, toff) == (no, to)
                assert nd[no:no + len(n1)].tobytes() == n1.tobytes() and tr[to:to + len(t1)].tobytes() == t1.tobytes(), (m.name, env)
                s1.pop("timeMs"), stats.pop("timeMs")
                assert s1 == stats, (m.name, env)
                no, to = no + len(n1), to + len(t1)
            assert len(nd) == no and len(tr) == to
    # an empty mesh in the batch: its one-node tree comes from the host, the batch goes mesh by mesh
    empty = pkg.meshes.Mesh(np.zeros((3, 3), np.float32), np.zeros((3, 3), np.float32), np.zeros(0, np.int32), "empty")
    nd, tr, per = api.build_bvh_arrays_gpu_batch([(m.vertices, m.normals, m.triangles) for m in (meshes[0], empty, meshes[2])], 1)
    assert [p[0] for p in per] == [0, per[1][0], per[1][0] + 1] and per[1][2]["triangleCount"] == 0
    # an empty batch is fine; a bad mesh in the middle reports its error
    nd, tr, per = api.build_bvh_arrays_gpu_batch([], 1)
    assert len(nd) == 0 and len(tr) == 0 and per == []
    bad = meshes[1].triangles.copy()
    bad[4] = 10**6
    with pytest.raises(pkg.abi.RtError):
        api.build_bvh_arrays_gpu_batch([(meshes[0].vertices, meshes[0].normals, meshes[0].triangles), (meshes[1].vertices, meshes[1].normals, bad)], 1)
    # concurrent builds from short-lived threads: serialised on the process-wide pool, same bytes, nothing left per thread
    ref = api.build_bvh_arrays(meshes[4].vertices, meshes[4].normals, meshes[4].triangles, 1)
    out = [None] * 6
    def work(k):
        out[k] = api.build_bvh_arrays_gpu(meshes[4].vertices, meshes[4].normals, meshes[4].triangles, 1)
    th = [threading.Thread(target=work, args=(k,)) for k in range(6)]
    [t.start() for t in th]
    [t.join() for t in th]
    for o in out:
        assert o[0].tobytes() == ref[0].tobytes() and o[1].tobytes() == ref[1].tobytes()
    api.build_bvh_gpu_release()


def _run_bvh_fuzz(*args):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "bvh_fuzz.py"), *[str(x) for x in args]], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "BVH FUZZ OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_builder_fuzz_host_threads_and_reference_text(pkg):
    """tools/bvh_fuzz.py without a device: random soups, clusters, sheets, duplicates, huge / tiny / non-finite coordinates — the host builder
    on 1 and 5 threads (and the reference's own BVH.cs compiled as C++, where oracle/_ref holds it) against rt_build_bvh, byte for byte,
    refusals alike."""
    out = _run_bvh_fuzz(80, 2500, 11)
    assert "80 meshes" in out


@pytest.mark.gpu
def test_builder_fuzz_gpu_single_and_forest(pkg):
    """The same fuzz with rt_build_bvh_gpu and random batches through rt_build_bvh_gpu_batch (one forest per batch)."""
    out = _run_bvh_fuzz(120, 5000, 12)
    assert "gpu True" in out
