"""Seeded random scenes: HIP vs the CPU oracle, bit for bit, incl. the exact work counters.
Random mixes of spheres and models (cubes, quads, rounded cubes, icospheres; shared meshes;
rotations and non-uniform scales), random materials (all three flags), random camera and
manager settings (bounces, spp, sky, depth of field), odd image sizes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
KEYS = ["segments", "innerSteps", "leafSteps", "triTests", "sphereTests", "modelVisits", "pixelFrames"]


def random_scene(pkg, seed):
    rng = np.random.default_rng(seed)
    M, T = pkg.RayTracingMaterial, pkg.Transform
    meshes = [pkg.meshes.cube(), pkg.meshes.quad(), pkg.meshes.rounded_cube(4), pkg.meshes.icosphere(2, 1.0, int(seed))]

    def material():
        flag = int(rng.choice([0, 0, 1, 2]))
        return M(flag=flag, diffuseCol=tuple(rng.uniform(0.05, 1, 3)) + (1,), emissionCol=tuple(rng.uniform(0, 1, 3)) + (1,),
                 specularCol=tuple(rng.uniform(0.3, 1, 3)) + (1,), absorption=tuple(rng.uniform(0, 1, 3)) + (1,),
                 absorptionMultiplier=float(rng.uniform(0, 2)), emissionStrength=float(rng.choice([0, 0, 0, 4.0])),
                 smoothness=float(rng.uniform(0, 1)), specularProbability=float(rng.uniform(0, 1)),
                 ior=float(rng.uniform(1.0, 2.2)))
    spheres = [pkg.Sphere(tuple(rng.uniform(-3, 3, 3) + [0, 1, 3]), float(rng.uniform(0.2, 1.2)), material())
               for _ in range(int(rng.integers(0, 21)))]
    models = []
    for _ in range(int(rng.integers(0, 9))):
        scale = tuple(rng.uniform(0.3, 2.5, 3)) if rng.random() < 0.5 else float(rng.uniform(0.3, 2.0))
        models.append(pkg.Model(meshes[int(rng.integers(0, 4))], material(),
                                T(tuple(rng.uniform(-3, 3, 3) + [0, 1, 3]), tuple(rng.uniform(0, 360, 3)), scale)))
    if rng.random() < 0.6:  # a big ground quad
        models.append(pkg.Model(meshes[1], material(), T((0, -0.5, 3), (90, 0, 0), (30, 30, 1))))
    cam = pkg.Camera(T(tuple(rng.uniform(-1, 1, 3) + [0, 1, -5]), (float(rng.uniform(-10, 15)), float(rng.uniform(-15, 15)), 0)),
                     fieldOfView=float(rng.uniform(30, 90)))
    settings = dict(maxBounceCount=int(rng.integers(0, 13)), numRaysPerPixel=int(rng.integers(1, 5)),
                    divergeStrength=float(rng.uniform(0, 3)), defocusStrength=float(rng.choice([0, 0, 50, 200])),
                    focusDistance=float(rng.uniform(0.5, 8)), useSky=bool(rng.random() < 0.7), sunFocus=float(rng.uniform(100, 900)),
                    sunIntensity=float(rng.uniform(1, 20)), accumulate=True,
                    bvhQuality=int(rng.choice([0, 1, 1, 2])))
    w, h = int(rng.integers(17, 70)), int(rng.integers(9, 40))
    frames = int(rng.integers(1, 4))
    sc = pkg.scenes.SceneDescription(f"fuzz{seed}", w, h, frames, settings, cam, models, spheres)
    return sc, int(rng.integers(-2**31, 2**31 - 1))


@pytest.mark.parametrize("seed", range(16))
def test_random_scene_bit_exact(pkg, api, orc, seed):
    out = []
    for lib, tr, stats in _three(api, orc):
        sc, render_seed = random_scene(pkg, seed)
        if stats:
            tr.enable_stats(True)
        mgr = sc.make_manager(tr, lib)
        mgr.OnEnable(renderSeed=render_seed)
        mgr.RenderFrames(sc.frames)
        acc = tr.read_accumulated()
        c = tr.counters()
        viol = tr.phase_profile()["filter_violations"][0] if stats else 0
        out.append((acc, [c[k] for k in KEYS], viol))
        tr.close()
    _compare(out, f"seed {seed}")


def _three(api, orc):
    """(library, tracer, stats?) for the three renders every case gets: the SHIPPED kernel instantiation
    (rt_trace_kernel<false, ...>: what bench.py times), the STATS instantiation (detailed counters + the filter
    audits; a different binary) and the oracle."""
    return ((api, api.create_tracer(0), False), (api, api.create_tracer(0), True), (orc, orc.create_tracer(8), False))


def _compare(out, what=""):
    (a0, c0, _), (a, ca, viol), (b, cb, _) = out
    for name, img in (("shipped", a0), ("stats", a)):
        same = img.view(np.uint32) == b.view(np.uint32)
        assert same.all(), f"{what} ({name} instantiation): {int((~same.all(axis=-1)).sum())} pixels differ"
    assert ca == cb, (what, ca, cb)
    assert c0[KEYS.index("segments")] == cb[KEYS.index("segments")] and c0[-1] == cb[-1]
    assert viol == 0
    return ca


def crowded_scene(pkg, n_models, n_spheres, seed=7):
    """More models than the 64-bit root-filter mask holds and more spheres than one 32-wide
    candidate word: the kernel's overflow paths (models >= 64 visited unfiltered, second sphere
    word) must give the reference's in-order results."""
    rng = np.random.default_rng(seed)
    M, T = pkg.RayTracingMaterial, pkg.Transform
    meshes = [pkg.meshes.cube(), pkg.meshes.rounded_cube(3), pkg.meshes.quad()]
    models = [pkg.Model(meshes[i % 3], M(flag=int(i % 4 == 3) * 2, diffuseCol=tuple(rng.uniform(0.2, 1, 3)) + (1,), ior=1.4,
                                         emissionStrength=float(i % 9 == 0) * 3.0, emissionCol=(1, 0.9, 0.7, 1)),
                        T(tuple(rng.uniform(-4, 4, 3) + [0, 1.5, 4]), tuple(rng.uniform(0, 360, 3)), float(rng.uniform(0.2, 0.7))))
              for i in range(n_models)]
    spheres = [pkg.Sphere(tuple(rng.uniform(-4, 4, 3) + [0, 1.5, 4]), float(rng.uniform(0.1, 0.5)),
                          M(flag=int(i % 5 == 0) * 2, diffuseCol=tuple(rng.uniform(0.2, 1, 3)) + (1,), ior=1.5, smoothness=0.6,
                            specularProbability=0.4))
               for i in range(n_spheres)]
    cam = pkg.Camera(T((0, 1.5, -6), (0, 0, 0)), fieldOfView=55.0)
    settings = dict(maxBounceCount=5, numRaysPerPixel=2, divergeStrength=0.5, useSky=True, accumulate=True, bvhQuality=1)
    return pkg.scenes.SceneDescription("crowded", 88, 48, 2, settings, cam, models, spheres)


@pytest.mark.parametrize("n_models,n_spheres,quality", [(64, 32, 1), (65, 33, 1), (97, 70, 0), (3, 64, 1), (70, 5, 2),
                                                        (200, 4, 1), (333, 0, 0)])
def test_more_models_and_spheres_than_one_mask_word(pkg, api, orc, n_models, n_spheres, quality):
    out = []
    for lib, tr, stats in _three(api, orc):
        sc = crowded_scene(pkg, n_models, n_spheres)
        sc.settings["bvhQuality"] = quality   # 2 = no BVH: every root is a leaf (flat kernel variant)
        if stats:
            tr.enable_stats(True)
        mgr = sc.make_manager(tr, lib)
        mgr.OnEnable(renderSeed=11)
        mgr.RenderFrames(sc.frames)
        acc = tr.read_accumulated()
        c = tr.counters()
        viol = tr.phase_profile()["filter_violations"][0] if stats else 0
        out.append((acc, [c[k] for k in KEYS], viol))
        tr.close()
    ca = _compare(out, f"{n_models} models, {n_spheres} spheres")
    assert ca[KEYS.index("modelVisits")] == ca[KEYS.index("segments")] * n_models


def test_two_level_model_hierarchy_follows_moving_models(pkg, api, orc):
    """200 models (two-level chunked filter, candidate masks in LDS): models move between frames, so the
    filter boxes, the chunk boxes and the spatial clustering are rebuilt and uploaded stream-ordered by
    rt_update_models; every frame must still equal the oracle (which simply walks all 200 models)."""
    out = []
    for lib, tr, stats in _three(api, orc):
        sc = crowded_scene(pkg, 200, 3, seed=21)
        if stats:
            tr.enable_stats(True)
        mgr = sc.make_manager(tr, lib)
        mgr.OnEnable(renderSeed=5)
        for f in range(4):
            for i, m in enumerate(mgr.models):
                if i % 3 == f % 3:   # a third of the models jump somewhere else each frame
                    p = m.transform.position
                    m.transform.position = (p[0] + 0.37 * ((i % 7) - 3), p[1], p[2] - 0.21 * ((i % 5) - 2))
            mgr.RenderFrame()
        c = tr.counters()
        viol = tr.phase_profile()["filter_violations"][0] if stats else 0
        out.append((tr.read_accumulated(), [c[k] for k in KEYS], viol))
        tr.close()
    _compare(out, "200 moving models")


@pytest.mark.parametrize("seed", range(6))
def test_schedule_independence_at_scale(pkg, api, seed, monkeypatch):
    """GPU vs GPU at sizes the oracle cannot reach: the shipped schedule (two render streams, fused
    frames, learnt tile order, full persistent grid) against the plainest one (one stream, one
    launch per frame, identity tile order, a grid a tenth the size) — same bits, same counters."""
    rng = np.random.default_rng(1000 + seed)
    cfg = int(rng.choice([2, 3, 4]))
    w, h = int(rng.integers(200, 1400)), int(rng.integers(120, 800))
    frames = int(rng.integers(2, 12))
    kw = {"subdivisions": 4} if cfg == 4 else None
    out = []
    for plain in (False, True):
        for k, v in (("RT_TWO_STREAMS", "0"), ("RT_FUSE_FRAMES", "0"), ("RT_LPT", "0"), ("RT_GRID", "500")):
            if plain:
                monkeypatch.setenv(k, v)
            else:
                monkeypatch.delenv(k, raising=False)
        tr = api.create_tracer(0)
        sc = pkg.scenes.get(cfg, **(kw or {}))
        mgr = sc.make_manager(tr, api, w, h)
        mgr.OnEnable(renderSeed=seed)
        if plain:
            for _ in range(frames):
                mgr.RenderFrame()
        else:
            mgr.RenderFrame()
            mgr.RenderFrames(frames - 1)
        out.append((tr.read_accumulated().copy(), tr.read_frame().copy(), tr.counters()["segments"]))
        tr.close()
    (a, fa, sa), (b, fb, sb) = out
    assert (a.view(np.uint32) == b.view(np.uint32)).all() and (fa.view(np.uint32) == fb.view(np.uint32)).all()
    assert sa == sb and sa > 0


def _corrupt(pkg, api, rng, mgr):
    """A real scene's buffers with 1-3 node fields overwritten (tools/ref_fuzz_corrupt.py's generator): child indices anywhere inside the
    buffer, leaf ranges moved or resized, nodes made second parents of other nodes' children.  Returns the buffers if rt_validate_scene
    ACCEPTS them, else None."""
    d = mgr.CreateAllMeshData(mgr.models)
    m, t, n = d["meshInfo"].copy(), d["triangles"].copy(), d["nodes"].copy()
    for _ in range(int(rng.integers(1, 4))):
        what = int(rng.integers(0, 4))
        i = int(rng.integers(0, len(n)))
        if what == 0:
            n["startIndex"][i] = int(rng.integers(0, len(n)))
        elif what == 1 and n["triangleCount"][i] > 0:
            n["startIndex"][i] = int(rng.integers(0, 40))
        elif what == 2 and n["triangleCount"][i] > 0:
            n["triangleCount"][i] = int(rng.integers(1, 9))
        else:
            j = int(rng.integers(0, len(n)))
            n["startIndex"][i], n["triangleCount"][i] = n["startIndex"][j], n["triangleCount"][j]
    try:
        api.validate_scene_arrays(m, t, n, mgr._pack_spheres())
    except pkg.abi.RtError:
        return None
    return m, t, n


@pytest.mark.timeout(600)
@pytest.mark.parametrize("batch", range(6))
def test_corrupted_but_accepted_scenes_bit_exact(pkg, api, orc, batch):
    """VERDICT r5, missing 2: what rt_validate_scene ACCEPTS the device must render like the reference, which walks nodeOffset + startIndex
    with no bounds or cycle check (RayCommon.hlsl:245-252, 264-267).  40 accepted corruptions per batch (240 in all; tools/ref_fuzz_corrupt.py
    runs thousands of them through the oracle and the reference's text on the CPU): HIP — the shipped and the STATS instantiation —
    against the oracle, both render targets and the exact counters, bit for bit; where the reference's own text travelled (oracle/_ref/libref.so),
    against that too.  The traversal watchdog bounds every kernel: a hole in the validation fails this test, it does not hang the GPU."""
    import __graft_entry__ as graft
    ref = None
    try:
        ref = graft.load_ref()
    except AssertionError:
        ref = None
    rng = np.random.default_rng(7000 + batch)
    families = [(3, {}), (4, {"subdivisions": 2}), (6, {})]
    W, H = 48, 27
    done = refused = it = 0
    while done < 40 and it < 4000:
        it += 1
        cfg, kw = families[it % 3]
        sc = pkg.scenes.get(cfg, **kw)
        sc.spheres = []
        probe = sc.make_manager(None, api, W, H)
        bufs = _corrupt(pkg, api, rng, probe)
        if bufs is None:
            refused += 1
            continue
        seed = 100 * batch + it
        out = []
        libs = [(api, False), (api, True), (orc, False)] + ([(ref, False)] if ref is not None else [])
        for lib, stats in libs:
            tr = lib.create_tracer(0 if lib is api else 4)
            if stats:
                tr.enable_stats(True)
            sc2 = pkg.scenes.get(cfg, **kw)
            sc2.spheres = []
            mgr = sc2.make_manager(tr, orc if lib is ref else lib, W, H)
            mgr.OnEnable(renderSeed=seed)
            tr.upload_scene(*bufs, mgr._pack_spheres())
            for _ in range(2):
                mgr.RenderFrame()
            c = tr.counters()
            out.append((tr.read_accumulated(), tr.read_frame(), c))
            tr.close()
        want_acc, want_frame, want_c = out[2]
        for k, (acc, frame, c) in enumerate(out):
            if k == 2:
                continue
            what = f"batch {batch} case {it} (config {cfg}) vs " + ("shipped", "stats", "", "reference text")[k]
            assert np.array_equal(acc.view(np.uint32), want_acc.view(np.uint32)), what
            assert np.array_equal(frame.view(np.uint32), want_frame.view(np.uint32)), what
            assert c["segments"] == want_c["segments"], what
            # the STATS instantiation keeps every counter; the reference's text only its own two: stats[0] = triangle tests (RC:254) and
            # stats[1] = box tests = 2 per inner node (RC:271)
            for key in {0: (), 1: ("innerSteps", "leafSteps", "triTests", "modelVisits"), 3: ("innerSteps", "triTests")}[k]:
                assert c[key] == want_c[key], (what, key, c[key], want_c[key])
        done += 1
    assert done == 40, (done, refused)
