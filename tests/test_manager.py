"""Host logic above the C ABI: the RayComputeManager mirror, transforms, mesh
de-duplication (RCM:206-236), frame counter protocol (RCM:69-95), partition maths."""
import numpy as np
import pytest


def test_transform_matches_unity_conventions(pkg):
    T = pkg.Transform
    # +90 deg about X maps the Unity quad normal (0,0,-1) to +Y (used for the ground quad)
    r = T(euler=(90, 0, 0)).rotation_matrix()
    assert np.allclose(r @ [0, 0, -1], [0, 1, 0], atol=1e-12)
    # +Y rotation turns +Z (forward) towards +X in Unity's left-handed convention
    assert np.allclose(T(euler=(0, 90, 0)).forward, [1, 0, 0], atol=1e-12)
    # +X rotation pitches forward DOWN
    assert T(euler=(30, 0, 0)).forward[1] < 0
    t = T(position=(1, 2, 3), euler=(10, 20, 30), scale=(2, 3, 4))
    m = t.localToWorldMatrix
    assert np.allclose(m @ t.worldToLocalMatrix, np.eye(4), atol=1e-12)
    assert np.allclose(m[:3, 3], [1, 2, 3])
    assert np.allclose(np.linalg.norm(m[:3, :3], axis=0), [2, 3, 4])
    # Unity memory order: column-major, m[c*4+r]
    flat = pkg.manager.matrix_to_abi(m)
    assert flat[12] == np.float32(1) and flat[13] == np.float32(2) and flat[14] == np.float32(3)
    assert flat[1] == np.float32(m[1, 0])


def test_material_defaults_and_packing(pkg):
    m = pkg.RayTracingMaterial()
    p = m.pack()
    # RayTracingMaterial.cs:29-38
    assert p["flag"] == 0 and list(p["diffuseCol"]) == [1, 1, 1, 1] and list(p["emissionCol"]) == [0, 0, 0, 0]
    assert p["smoothness"] == 0 and p["specularProbability"] == 1 and p["ior"] == 1 and p["emissionStrength"] == 0
    with pytest.raises(AttributeError):
        pkg.RayTracingMaterial(nope=1)


class FakeTracer:
    """Records the ABI calls the manager makes (no library needed)."""

    def __init__(self):
        self.calls = []
        self._frame = 0

    def __getattr__(self, name):
        def f(*a):
            self.calls.append((name, a))
        return f

    def set_params(self, p):
        self.calls.append(("set_params", (p.frame, p.renderSeed, p.maxBounceCount)))
        self._frame = p.frame


def test_manager_call_sequence_and_frame_counter(pkg, api):
    tr = FakeTracer()
    sc = pkg.scenes.get(3)
    mgr = sc.make_manager(tr, api, 64, 36)
    mgr.OnEnable(renderSeed=42)
    names = [c[0] for c in tr.calls]
    # RCM:69-76 + 115-124: resize, upload (first InitFrame only), update models, params, reset kernel
    assert names == ["resize", "upload_scene", "update_models", "set_params", "reset_accumulation"]
    assert mgr.numAccumulatedFrames == 1 and tr.calls[3][1] == (1, 42, 8)
    tr.calls.clear()
    mgr.RenderFrame()
    assert [c[0] for c in tr.calls] == ["update_models", "set_params", "render_frame"]  # no re-upload: hasBVH latch
    assert tr.calls[1][1][0] == 1 and mgr.numAccumulatedFrames == 2  # Frame used THEN incremented (RCM:165,94)
    mgr.RenderFrame()
    assert mgr.numAccumulatedFrames == 3
    mgr.accumulate = False
    mgr.RenderFrame()
    assert mgr.numAccumulatedFrames == 3  # RCM:94 only counts when accumulating
    mgr.rayTracingEnabled = False
    tr.calls.clear()
    mgr.RenderFrame()
    assert tr.calls == []  # RCM:86


def test_create_all_mesh_data_dedupes_meshes(pkg, api):
    sc = pkg.scenes.get(3)
    mgr = sc.make_manager(FakeTracer(), api, 64, 36)
    data = mgr.CreateAllMeshData(mgr.models)
    info = data["meshInfo"]
    assert len(info) == 9
    # 6 cubes share one BVH, the quad its own, both rounded cubes share one (RCM:214-223)
    offsets = sorted(set((int(i["nodeOffset"]), int(i["triOffset"])) for i in info))
    assert len(offsets) == 3
    assert len(data["triangles"]) == 12 + 2 + 1728 == sc.unique_triangles()
    cubes = [i for i, m in enumerate(mgr.models) if m.Mesh.name == "Cube"]
    assert len(set(int(info[i]["nodeOffset"]) for i in cubes)) == 1
    # every model's root is where its offsets say
    for i in info:
        root = data["nodes"][i["nodeOffset"]]
        assert root["triangleCount"] != 0


def test_params_follow_manager_fields(pkg, api):
    sc = pkg.scenes.get(2)
    mgr = sc.make_manager(FakeTracer(), api, 192, 108)
    mgr.renderSeed = 7
    mgr.numAccumulatedFrames = 5
    p = mgr.params()
    assert (p.maxBounceCount, p.numRaysPerPixel, p.frame, p.renderSeed, p.useSky, p.accumulate) == (8, 8, 5, 7, 1, 1)
    assert list(p.dirToSun) == [0, -1, 0]  # RCM:176: sunTransform == null -> Vector3.down
    h = 1.0 * np.tan(np.radians(30)) * 2
    assert abs(p.viewParams[1] - h) < 1e-6 and abs(p.viewParams[0] - h * 192 / 108) < 1e-5
    mgr.sunTransform = pkg.Transform(euler=(90, 0, 0))  # forward = straight down -> dirToSun = up
    assert np.allclose(list(mgr.params().dirToSun), [0, 1, 0], atol=1e-6)


def test_scene_inventory(pkg):
    s = pkg.scenes
    c1, c2, c3 = s.get(1), s.get(2), s.get(3)
    assert (c1.width, c1.height, len(c1.spheres), len(c1.models)) == (256, 256, 4, 0)
    assert (c2.width, c2.height, len(c2.spheres), len(c2.models), c2.settings["numRaysPerPixel"], c2.settings["maxBounceCount"]) == (1920, 1080, 16, 1, 8, 8)
    assert c3.spp() == 64 and len(c3.models) == 9
    c4 = s.get(4, subdivisions=3)
    assert c4.settings["defocusStrength"] == 100.0 and c4.settings["focusDistance"] == 5.3 and c4.spp() == 32
    for sc in (c2, c3, c4):
        for m in sc.models:
            assert pkg.meshes.check_winding(m.Mesh) == 1.0
    assert pkg.meshes.icosphere(6).triangle_count == 81920


@pytest.mark.parametrize("H,world", [(1080, 1), (1080, 2), (1080, 8), (2160, 8), (45, 4), (7, 3)])
def test_row_partition_covers_image_exactly_once(pkg, H, world):
    d = pkg.dist
    seen = np.zeros(H, int)
    for r in range(world):
        rows = d.global_rows_of(r, world, H)
        seen[rows] += 1
        assert np.all(np.diff(rows) > 0)
        # strips of 8 rows dealt cyclically
        assert all((row // 8) % world == r for row in rows)
    assert np.all(seen == 1)
    assert d.max_local_rows(world, H) == len(d.global_rows_of(0, world, H))
