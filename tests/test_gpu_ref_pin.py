"""HIP kernels (through the C ABI) against the REFERENCE'S OWN TEXT compiled as C++ (oracle/_ref/libref.so, built by
oracle/make_ref.py in the container that has /root/reference and shipped to the GPU box as a prebuilt library).

tests/test_ref_pin.py pins the oracle to that library on the CPU; here the product is compared with it directly — no
restatement in between: FrameRender and AccumulatedRender of both kernel instantiations bit for bit, and the shader's own
`stats` counters (RC:254 triangle tests, RC:271 box tests / 2 = inner steps) against the kernel's exact counters.  The
reference has no sphere buffer, so these are the BVH scenes: BASELINE configs 3-5 classes and all five reference scene files.
"""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("rt_ref_lib", os.path.join(ROOT, "oracle", "ref_lib.py"))
ref_lib = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(ref_lib)


@pytest.fixture(scope="module")
def ref(pkg):
    lib = ref_lib.load(pkg)
    if lib is None:
        pytest.skip("oracle/_ref/libref.so did not travel with the snapshot")
    return lib


CASES = [
    ("config3", 3, {}, 240, 135, 3),
    ("config3_ragged", 3, {}, 61, 35, 2),
    ("config4_dof", 4, {"subdivisions": 5}, 160, 90, 2),
    ("config5_class", 5, {"subdivisions": 3, "n_meshes": 12}, 128, 72, 1),
    ("glass_balls.unity", 6, {}, 139, 78, 2),
    ("glass_dragon.unity", 7, {}, 96, 54, 2),
    ("sphere_refract.unity", 8, {}, 80, 45, 2),
    ("splash.unity", 9, {}, 96, 54, 2),
    ("text.unity", 10, {}, 88, 50, 2),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_hip_equals_the_compiled_reference_text(pkg, api, orc, ref, case):
    name, cfg, kw, w, h, frames = case
    threads = min(64, os.cpu_count() or 8)
    out = []
    for lib, tr, stats in ((api, api.create_tracer(0), False), (api, api.create_tracer(0), True), (ref, ref.create_tracer(threads), False)):
        if stats:
            tr.enable_stats(True)
        sc = pkg.scenes.get(cfg, **kw)
        mgr = sc.make_manager(tr, orc if lib is ref else lib, w, h)  # the reference text has no BVH builder (BVH.cs is C#)
        mgr.OnEnable(renderSeed=11)
        for _ in range(frames):
            mgr.RenderFrame()
        out.append((tr.read_accumulated(), tr.read_frame(), tr.counters()))
        tr.close()
    (a0, f0, c0), (a1, f1, c1), (ar, fr, cr) = out
    for which, a, f in (("shipped", a0, f0), ("stats", a1, f1)):
        assert np.array_equal(a.view(np.uint32), ar.view(np.uint32)), f"{name}: {which} AccumulatedRender != reference text"
        assert np.array_equal(f.view(np.uint32), fr.view(np.uint32)), f"{name}: {which} FrameRender != reference text"
    assert c0["segments"] == cr["segments"] == c1["segments"]
    assert c1["triTests"] == cr["triTests"], (name, c1, cr)        # stats[0], RC:254
    assert c1["innerSteps"] == cr["innerSteps"], (name, c1, cr)    # stats[1] / 2, RC:271


def test_gpu_bvh_builder_equals_the_compiled_reference_bvh_text(pkg, api):
    """rt_build_bvh_gpu and the forest of rt_build_bvh_gpu_batch against BVH.cs itself (oracle/_ref/libref_bvh.so: the reference's C#
    text compiled as C++): nodes, triangle order and BuildStats byte for byte — no restatement in between."""
    ref_bvh = ref_lib.load_bvh(pkg)
    if ref_bvh is None:
        pytest.skip("oracle/_ref/libref_bvh.so did not travel with the snapshot")
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_bvh
    meshes = test_bvh.mesh_cases(pkg) + [pkg.meshes.icosphere(5, 1.0, 2), pkg.meshes.icosphere(6, 1.0, 4)]

    def same(a, b):
        sa, sb = dict(a[2]), dict(b[2])
        sa.pop("timeMs"), sb.pop("timeMs")
        return a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes() and sa == sb

    for quality in (0, 1, 2):
        refs = [ref_bvh.build_bvh_arrays(m.vertices, m.normals, m.triangles, quality) for m in meshes]
        for m, r in zip(meshes, refs):
            assert same(api.build_bvh_arrays_gpu(m.vertices, m.normals, m.triangles, quality), r), (m.name, quality)
        nd, tr, per = api.build_bvh_arrays_gpu_batch([(m.vertices, m.normals, m.triangles) for m in meshes], quality)
        for m, r, (noff, toff, stats) in zip(meshes, refs, per):
            assert same((nd[noff:noff + len(r[0])], tr[toff:toff + len(r[1])], stats), r), ("batch", m.name, quality)
    api.build_bvh_gpu_release()
