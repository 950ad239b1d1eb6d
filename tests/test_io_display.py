"""The rows either side of the hot path: scene files in, display / image files /
checkpoints out.  CPU tests drive the oracle through the same host code; the GPU tests hold the HIP
library to the same bits."""
import json
import os

import numpy as np
import pytest

from conftest import render


def test_scene_json_round_trip_renders_identically(pkg, orc, tmp_path):
    sc = pkg.scenes.get(3)
    path = tmp_path / "scene.json"
    pkg.sceneio.save_scene(str(path), sc, mesh_specs={"Cube": {"type": "cube"}, "Quad": {"type": "quad"},
                                                      "RoundedCube": {"type": "rounded_cube", "k": 12}})
    sc2 = pkg.sceneio.load_scene(str(path))
    assert len(sc2.models) == 9 and sc2.unique_triangles() == sc.unique_triangles()
    imgs = []
    for s in (sc, sc2):
        tr = orc.create_tracer(8)
        mgr = s.make_manager(tr, orc, 48, 27)
        mgr.OnEnable(renderSeed=5)
        mgr.RenderFrames(1)
        imgs.append(tr.read_accumulated())
        tr.close()
    assert np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32))


def test_scene_json_with_arrays_and_spheres(pkg, tmp_path):
    sc = pkg.scenes.get(2)
    d = pkg.sceneio.scene_to_dict(sc)
    assert d["meshes"]["Quad"]["type"] == "arrays" and len(d["spheres"]) == 16
    sc2 = pkg.sceneio.scene_from_dict(json.loads(json.dumps(d)))
    assert [s.radius for s in sc2.spheres] == [s.radius for s in sc.spheres]
    assert sc2.settings == sc.settings
    with pytest.raises(ValueError):
        pkg.sceneio.scene_from_dict({"settings": {"nope": 1}})


def test_obj_loader(pkg, tmp_path):
    # a unit cube with per-face normals as quads (the shape of the reference's cube_rounded2.obj export)
    lines = ["# test"]
    corners = [(x, y, z) for x in (-.5, .5) for y in (-.5, .5) for z in (-.5, .5)]
    for c in corners:
        lines.append("v %g %g %g" % c)
    faces = {(1, 0, 0): [4, 6, 7, 5], (-1, 0, 0): [0, 1, 3, 2], (0, 1, 0): [2, 3, 7, 6], (0, -1, 0): [0, 4, 5, 1],
             (0, 0, 1): [1, 5, 7, 3], (0, 0, -1): [0, 2, 6, 4]}
    for n in faces:
        lines.append("vn %g %g %g" % n)
    for ni, (n, q) in enumerate(faces.items()):
        lines.append("f " + " ".join(f"{v + 1}//{ni + 1}" for v in q))
    p = tmp_path / "cube.obj"
    p.write_text("\n".join(lines))
    m = pkg.sceneio.load_obj(str(p))
    assert m.triangle_count == 12 and len(m.vertices) == 24
    assert pkg.meshes.check_winding(m) == 1.0
    mu = pkg.sceneio.load_obj(str(p), unity_import=True)
    assert pkg.meshes.check_winding(mu) == 1.0 and np.allclose(mu.vertices[:, 0], -m.vertices[:, 0])


@pytest.mark.skipif(not os.path.exists("/root/reference/Assets/Graphics/cube_rounded2.obj"), reason="reference asset not present")
def test_reference_rounded_cube_obj_loads(pkg, api):
    m = pkg.sceneio.load_obj("/root/reference/Assets/Graphics/cube_rounded2.obj")
    assert m.triangle_count == 1724           # 858 quads + 8 triangles
    assert pkg.meshes.check_winding(m) == 1.0
    nodes, tris, stats = api.build_bvh_arrays(m.vertices, m.normals, m.triangles)
    assert stats["triangleCount"] == 1724 and stats["leafDepthMax"] <= 32


def test_display_matches_reference_formula(pkg, orc):
    """Display.shader: tex / Frame with Frame = numAccumulatedFrames AFTER the increment (off-by-one)."""
    tr = orc.create_tracer(4)
    acc, mgr = render(pkg, orc, tr, 2, 40, 24, 3)
    disp = pkg.display.RayTraceDisplay(mgr)
    assert disp.frame_divisor() == 4                      # 3 frames rendered, counter at 4
    img = disp.OnRenderImage()
    want = (acc.astype(np.float32) * (np.float32(1.0) / np.float32(4))).astype(np.float32)
    assert np.array_equal(img, want)
    assert np.allclose(disp.average(), acc[..., :3] / 3.0)
    mgr.accumulate = False
    assert disp.frame_divisor() == 1
    assert np.array_equal(disp.OnRenderImage(), tr.read_frame())
    tr.close()


def test_srgb8_and_png(pkg, orc, tmp_path):
    tr = orc.create_tracer(4)
    acc, mgr = render(pkg, orc, tr, 1, 32, 32, 2)
    disp = pkg.display.RayTraceDisplay(mgr)
    img8 = disp.srgb8()
    assert img8.shape == (32, 32, 4) and img8.dtype == np.uint8 and np.all(img8[..., 3] == 255)
    lin = np.clip(acc[::-1, :, :3].astype(np.float64) / 3.0, 0, 1)
    ref = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * lin ** (1 / 2.4) - 0.055)
    assert np.max(np.abs(img8[..., :3].astype(np.int32) - np.floor(ref * 255 + 0.5).astype(np.int32))) <= 1
    disp.save_png(str(tmp_path / "a.png"))
    disp.save_pfm(str(tmp_path / "a.pfm"))
    from PIL import Image
    back = np.asarray(Image.open(tmp_path / "a.png"))
    assert np.array_equal(back, img8)
    assert (tmp_path / "a.pfm").stat().st_size == len(b"PF\n32 32\n-1.0\n") + 32 * 32 * 12
    tr.close()


def _checkpoint_roundtrip(pkg, lib, make_tracer, tmp_path, cfg):
    t1 = make_tracer()
    straight, _ = render(pkg, lib, t1, cfg, 48, 27, 5, seed=7)
    t1.close()
    t2 = make_tracer()
    _, mgr = render(pkg, lib, t2, cfg, 48, 27, 2, seed=7)
    ck = str(tmp_path / "ck.npz")
    pkg.display.save_checkpoint(ck, mgr)
    t2.close()
    t3 = make_tracer()
    mgr3 = pkg.scenes.get(cfg).make_manager(t3, lib, 48, 27)
    meta = pkg.display.load_checkpoint(ck, mgr3)
    assert meta["numAccumulatedFrames"] == 3 and meta["renderSeed"] == 7
    mgr3.RenderFrames(3)
    resumed = t3.read_accumulated()
    t3.close()
    assert np.array_equal(straight.view(np.uint32), resumed.view(np.uint32))


def test_checkpoint_resume_is_bit_identical_cpu(pkg, orc, tmp_path):
    _checkpoint_roundtrip(pkg, orc, lambda: orc.create_tracer(4), tmp_path, 3)


@pytest.mark.gpu
def test_checkpoint_resume_is_bit_identical_gpu(pkg, api, tmp_path):
    _checkpoint_roundtrip(pkg, api, lambda: api.create_tracer(0), tmp_path, 3)
    _checkpoint_roundtrip(pkg, api, lambda: api.create_tracer(0), tmp_path, 2)


@pytest.mark.gpu
def test_display_and_srgb8_match_oracle_on_gpu(pkg, api, orc):
    out = []
    for lib, tr in ((api, api.create_tracer(0)), (orc, orc.create_tracer(8))):
        _, mgr = render(pkg, lib, tr, 2, 96, 54, 3)
        d = pkg.display.RayTraceDisplay(mgr)
        out.append((d.OnRenderImage(), d.srgb8(), d.srgb8(flip_y=False)))
        tr.close()
    for a, b in zip(*out):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.mark.gpu
def test_json_scene_renders_on_gpu_like_the_builtin(pkg, api, tmp_path):
    sc = pkg.scenes.get(4, subdivisions=3)
    p = str(tmp_path / "s.json")
    pkg.sceneio.save_scene(p, sc)
    imgs = []
    for s in (sc, pkg.sceneio.load_scene(p)):
        tr = api.create_tracer(0)
        mgr = s.make_manager(tr, api, 64, 36)
        mgr.OnEnable(renderSeed=3)
        mgr.RenderFrames(2)
        imgs.append(tr.read_accumulated())
        tr.close()
    assert np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32))
