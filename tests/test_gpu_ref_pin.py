"""HIP kernels (through the C ABI) against the REFERENCE'S OWN TEXT compiled as C++ (oracle/_ref/libref.so, built by
oracle/make_ref.py in the container that has /root/reference and shipped to the GPU box as a prebuilt library).

tests/test_ref_pin.py pins the oracle to that library on the CPU; here the product is compared with it directly — no
restatement in between: FrameRender and AccumulatedRender of both kernel instantiations bit for bit, and the shader's own
`stats` counters (RC:254 triangle tests, RC:271 box tests / 2 = inner steps) against the kernel's exact counters.  The
reference has no sphere buffer, so libref.so covers the BVH scenes (BASELINE configs 3-5 classes, all five reference scene
files); the sphere scenes — BASELINE configs 1 and 2, the headline image — are compared with libref_spheres.so, the same text plus
the one declared semantic rewrite of make_ref.py (the reference's own RaySphere called from its commented call site, RC:341).
A prebuilt library that MANIFEST.json / oracle/REF_EXPECTED.json do not vouch for fails these tests (it does not skip them).
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


def _checked(pkg, variant, name):
    lib = ref_lib.load(pkg, variant)
    if lib is None:
        pytest.skip(f"oracle/_ref/{name} did not travel with the snapshot")
    # this box cannot rebuild the library (no reference checkout): one that was built from other sources or by another recipe than
    # oracle/REF_EXPECTED.json names FAILS the suite — a stale checker that still answers must not pass for the reference
    why = ref_lib.stale_reason([name])
    if why:
        pytest.fail("stale reference library: " + why)
    return lib


@pytest.fixture(scope="module")
def ref(pkg):
    return _checked(pkg, "", "libref.so")


@pytest.fixture(scope="module")
def ref_spheres(pkg):
    return _checked(pkg, "spheres", "libref_spheres.so")


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


SPHERE_CASES = [  # BASELINE config 1 exactly; config 2 = the headline image; spheres + BVH meshes together
    ("config1_exact", lambda pkg: pkg.scenes.get(1), 256, 256, 1),
    ("config2_headline", lambda pkg: pkg.scenes.get(2), 480, 270, 2),
    ("config2_ragged", lambda pkg: pkg.scenes.get(2), 61, 35, 3),
    ("fuzz3_spheres_and_meshes", lambda pkg: __import__("test_gpu_fuzz").random_scene(pkg, 3)[0], None, None, None),
    ("fuzz5_spheres_and_meshes", lambda pkg: __import__("test_gpu_fuzz").random_scene(pkg, 5)[0], None, None, None),
]


@pytest.mark.parametrize("case", SPHERE_CASES, ids=[c[0] for c in SPHERE_CASES])
def test_hip_equals_the_reference_text_with_the_sphere_hook(pkg, api, orc, ref_spheres, case):
    """The sphere path — two-phase pre-test + exact roots in the kernels (rt_kernels.h begin_intersect) — against the reference's
    own RaySphere (RC:289-332) called from the place of its commented call (RC:341): libref_spheres.so is the reference's text
    plus that declared hook (oracle/make_ref.py S1), nothing of the builder's restatement in between."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    name, factory, w, h, frames = case
    threads = min(64, os.cpu_count() or 8)
    out = []
    for lib, tr, stats in ((api, api.create_tracer(0), False), (api, api.create_tracer(0), True), (ref_spheres, ref_spheres.create_tracer(threads), False)):
        if stats:
            tr.enable_stats(True)
        sc = factory(pkg)
        mgr = sc.make_manager(tr, orc if lib is ref_spheres else lib, w or sc.width, h or sc.height)
        mgr.OnEnable(renderSeed=11)
        for _ in range(frames or sc.frames):
            mgr.RenderFrame()
        out.append((tr.read_accumulated(), tr.read_frame(), tr.counters()))
        tr.close()
    (a0, f0, c0), (a1, f1, c1), (ar, fr, cr) = out
    for which, a, f in (("shipped", a0, f0), ("stats", a1, f1)):
        assert np.array_equal(a.view(np.uint32), ar.view(np.uint32)), f"{name}: {which} AccumulatedRender != reference text + sphere hook"
        assert np.array_equal(f.view(np.uint32), fr.view(np.uint32)), f"{name}: {which} FrameRender != reference text + sphere hook"
    assert c0["segments"] == cr["segments"] == c1["segments"]
    assert c1["triTests"] == cr["triTests"] and c1["innerSteps"] == cr["innerSteps"], (name, c1, cr)


def test_gpu_bvh_builder_equals_the_compiled_reference_bvh_text(pkg, api):
    """rt_build_bvh_gpu and the forest of rt_build_bvh_gpu_batch against BVH.cs itself (oracle/_ref/libref_bvh.so: the reference's C#
    text compiled as C++): nodes, triangle order and BuildStats byte for byte — no restatement in between."""
    ref_bvh = ref_lib.load_bvh(pkg)
    if ref_bvh is None:
        pytest.skip("oracle/_ref/libref_bvh.so did not travel with the snapshot")
    why = ref_lib.stale_reason(["libref_bvh.so"])
    if why:
        pytest.fail("stale reference library: " + why)
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
