"""The C-ABI library loads on a CPU-only host, exports every symbol rt_abi.h
declares, matches the header's struct sizes, and fails loudly without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "rt_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_all_exported(pkg, api):
    names = header_functions()
    assert len(names) >= 25
    assert sorted(pkg.hip.ABI_SYMBOLS) == names, "hip.ABI_SYMBOLS is out of sync with include/rt_abi.h"
    for n in names:
        assert hasattr(api.lib, n), f"libraytrace_hip.so does not export {n}"


def test_version_string(api):
    assert api.version() == b"raytrace_hip gfx950 abi=1"


def test_struct_sizes_match_reference_layouts(pkg):
    a = pkg.abi
    # RC:64-76 (88), RC:78-85 (224), RC:49-53 (72), RC:87-95 (32)
    assert a.material_dtype.itemsize == 88
    assert a.model_dtype.itemsize == 224
    assert a.triangle_dtype.itemsize == 72
    assert a.node_dtype.itemsize == 32
    assert a.sphere_dtype.itemsize == 104
    assert a.model_dtype.fields["worldToLocal"][1] == 8
    assert a.model_dtype.fields["localToWorld"][1] == 72
    assert a.model_dtype.fields["material"][1] == 136
    assert a.material_dtype.fields["flag"][1] == 84
    header = open(os.path.join(ROOT, "include", "rt_abi.h")).read()
    for name, size in (("RtMaterial", 88), ("RtModel", 224), ("RtTriangle", 72), ("RtBVHNode", 32), ("RtSphere", 104)):
        assert f"sizeof({name}) == {size}" in header
    assert C.sizeof(a.RtParams) == 8 + 6 * 4 + 4 * 4 + 9 * 4 + 64
    assert C.sizeof(a.RtCounters) == 64


def test_code_object_is_gfx950(pkg):
    blob = open(pkg.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"rt_trace_kernel" in blob


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful on a host without a GPU")
def test_create_fails_loudly_without_gpu(pkg, api):
    with pytest.raises(pkg.abi.RtError) as e:
        api.create_tracer(0)
    assert e.value.status == pkg.abi.RT_ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful on a host without a GPU")
def test_gpu_bvh_builder_fails_loudly_without_gpu(pkg, api):
    m = pkg.meshes.cube()
    with pytest.raises(pkg.abi.RtError) as e:
        api.build_bvh_arrays_gpu(m.vertices, m.normals, m.triangles)
    assert e.value.status == pkg.abi.RT_ERR_NO_DEVICE


def test_host_helpers_work_without_gpu(pkg, api):
    vp = api.view_params(60.0, 16 / 9, 1.0)
    assert abs(vp[1] - 2 * 0.57735027) < 1e-6 and abs(vp[0] - vp[1] * 16 / 9) < 1e-6 and vp[2] == 1.0
    m = pkg.meshes.cube()
    nodes, tris, stats = api.build_bvh_arrays(m.vertices, m.normals, m.triangles)
    assert stats["triangleCount"] == 12 and len(tris) == 12
    # bad input is an error code, not a crash
    bad = m.triangles.copy()
    bad[0] = 99
    with pytest.raises(pkg.abi.RtError):
        api.build_bvh_arrays(m.vertices, m.normals, bad)


def test_product_never_touches_the_oracle(pkg):
    """No file of the product package mentions the oracle library."""
    pdir = os.path.dirname(pkg.__file__)
    for dirpath, _, files in os.walk(pdir):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in text and "oracle_lib" not in text and "oracle/" not in text.replace("oracle/rt_oracle.cpp", ""), f


def _scene_arrays(pkg, api, cfg=4, **kw):
    sc = pkg.scenes.get(cfg, **kw)
    mgr = sc.make_manager(None, api)
    mgr.renderSeed = 1
    return mgr.CreateAllMeshData(mgr.models), mgr._pack_spheres()


def test_validate_scene_is_the_host_half_of_upload_scene(pkg, api, monkeypatch):
    """rt_validate_scene needs no device: the same checks and the same re-layout as rt_upload_scene.  A large scene with several
    meshes is converted on worker threads (one segment of the pair array per mesh); the sequential walk must see the same tree."""
    a = pkg.abi
    data, spheres = _scene_arrays(pkg, api, 4)                   # 81,920-triangle mesh + room: 166,010 nodes
    assert len(data["nodes"]) >= 1 << 16 and len({int(m["nodeOffset"]) for m in data["meshInfo"]}) >= 2
    infos = []
    for env in ({}, {"RT_SEQUENTIAL_PREPARE": "1"}, {"RT_HOST_THREADS": "1"}):
        for k in ("RT_SEQUENTIAL_PREPARE", "RT_HOST_THREADS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        infos.append(api.validate_scene_arrays(data["meshInfo"], data["triangles"], data["nodes"], spheres))
    n_inner = int((data["nodes"]["triangleCount"] <= 0).sum())
    for info in infos:
        assert info["max_height"] == infos[1]["max_height"] >= 10 and info["flat"] == 0 and info["n_filtered"] == infos[1]["n_filtered"]
    assert infos[1]["n_pairs"] == n_inner                       # the sequential walk: one pair per inner node
    n_meshes = len({int(m["nodeOffset"]) for m in data["meshInfo"]})
    assert n_inner <= infos[0]["n_pairs"] <= n_inner + 2 * n_meshes and infos[2]["n_pairs"] == infos[0]["n_pairs"]   # at most a slot of slack per mesh
    small, sp = _scene_arrays(pkg, api, 3)
    info = api.validate_scene_arrays(small["meshInfo"], small["triangles"], small["nodes"], sp)
    assert info["n_pairs"] == int((small["nodes"]["triangleCount"] <= 0).sum()) and info["flat"] == 0
    assert api.validate_scene_arrays(None, None, None, None) == {"n_pairs": 0, "max_height": 1, "flat": 1, "n_filtered": 0, "prepare_ms": pytest.approx(0, abs=50)}


@pytest.mark.parametrize("sequential", [False, True])
def test_validate_scene_refuses_what_upload_scene_refuses(pkg, api, monkeypatch, sequential):
    import numpy as np
    a = pkg.abi
    if sequential:
        monkeypatch.setenv("RT_SEQUENTIAL_PREPARE", "1")
    else:
        monkeypatch.delenv("RT_SEQUENTIAL_PREPARE", raising=False)
    data, spheres = _scene_arrays(pkg, api, 4)
    info, tris, nodes = data["meshInfo"], data["triangles"], data["nodes"]

    def refused(models, triangles, nds, what):
        with pytest.raises(a.RtError) as e:
            api.validate_scene_arrays(models, triangles, nds, spheres)
        assert e.value.status == a.RT_ERR_SCENE and what in str(e.value), str(e.value)

    big = max(range(len(info)), key=lambda i: 0 if i + 1 == len(info) else int(info[i + 1]["nodeOffset"]) - int(info[i]["nodeOffset"]))
    # a model that re-uses the big mesh with a triangle offset that runs its leaves past the buffer
    bad = info.copy()
    bad[-1]["nodeOffset"] = info[big]["nodeOffset"]
    bad[-1]["triOffset"] = len(tris) - 1
    refused(bad, tris, nodes, "out of bounds")
    # a cycle: the root's first child names its own sibling pair as its children
    root = int(info[big]["nodeOffset"])
    first = root + int(nodes[root]["startIndex"])
    assert nodes[root]["triangleCount"] <= 0 and nodes[first]["triangleCount"] <= 0
    cyc = nodes.copy()
    cyc[first]["startIndex"] = nodes[root]["startIndex"]
    refused(info, tris, cyc, "cycle")
    # a child index outside the node buffer, an empty mesh, a leaf past the triangle buffer, bad offsets
    oob = nodes.copy()
    oob[first]["startIndex"] = len(nodes)
    refused(info, tris, oob, "child index out of bounds")
    empty = nodes.copy()
    empty[root]["triangleCount"] = 0
    refused(info, tris, empty, "empty mesh")
    refused(info, tris[: len(tris) // 2], nodes, "out of")
    off = info.copy()
    off[0]["nodeOffset"] = len(nodes)
    refused(off, tris, nodes, "out of range")
    # a mesh that reaches into another mesh's nodes (outside its window) goes to the sequential walk — which refuses it when the grafted nodes would have to
    # mean two things: a child index is the MODEL's nodeOffset + startIndex (RC:265-266), so inner nodes read under another offset are another tree (round 5;
    # rounds 1-4 reused the first reading)
    if not sequential:
        other = next(i for i in range(len(info)) if int(info[i]["nodeOffset"]) != root)
        graft = nodes.copy()
        o_root = int(info[other]["nodeOffset"])
        if graft[o_root]["triangleCount"] <= 0:
            # the other mesh's root now has the big mesh's root pair as children (indices are relative to ITS nodeOffset)
            graft[o_root]["startIndex"] = first - o_root
            try:
                got = api.validate_scene_arrays(info, tris, graft, spheres)
                assert got["n_pairs"] > 0
            except a.RtError as e:      # a triangle-range refusal (the grafted leaves use the other mesh's triOffset) or the two-offsets refusal
                assert e.status == a.RT_ERR_SCENE and ("out of bounds" in str(e) or "two different nodeOffsets" in str(e))


def test_validate_scene_fuzz_parallel_and_sequential_agree(pkg, api, monkeypatch):
    """Corrupted scenes (random child indices, triangle counts, bit flips in the node array, model offsets, links between subtrees,
    NaN / inf bounds): never a crash or a hang, and the worker-thread preparation accepts and refuses exactly what the sequential walk
    does — with the same status, tree height and filter count."""
    import numpy as np
    a = pkg.abi
    data, sph = _scene_arrays(pkg, api, 4)
    rng = np.random.default_rng(1)

    def outcome(models, tris, nodes, seq):
        if seq:
            monkeypatch.setenv("RT_SEQUENTIAL_PREPARE", "1")
        else:
            monkeypatch.delenv("RT_SEQUENTIAL_PREPARE", raising=False)
        try:
            i = api.validate_scene_arrays(models, tris, nodes, sph)
            return ("ok", i["max_height"], i["flat"], i["n_filtered"])
        except a.RtError as e:
            return ("err", e.status)

    seen = {"ok": 0, "err": 0}
    for it in range(150):
        nodes, models, tris = data["nodes"].copy(), data["meshInfo"].copy(), data["triangles"]
        kind = it % 6
        for _ in range(int(rng.integers(1, 6))):
            i = int(rng.integers(0, len(nodes)))
            if kind == 0:
                nodes[i]["startIndex"] = int(rng.integers(-5, len(nodes) + 5))
            elif kind == 1:
                nodes[i]["triangleCount"] = int(rng.integers(-3, 200))
            elif kind == 2:
                raw = nodes.view(np.uint8)
                raw[int(rng.integers(0, raw.size))] ^= np.uint8(1 << int(rng.integers(0, 8)))
            elif kind == 3:
                m = int(rng.integers(0, len(models)))
                models[m]["nodeOffset"] = int(rng.integers(-2, len(nodes) + 2))
                models[m]["triOffset"] = int(rng.integers(-2, len(tris) + 2))
            elif kind == 4:
                nodes[i]["startIndex"] = nodes[int(rng.integers(0, len(nodes)))]["startIndex"]
            else:
                nodes[i]["boundsMin"][int(rng.integers(0, 3))] = [np.nan, np.inf, -np.inf][int(rng.integers(0, 3))]
        o1, o2 = outcome(models, tris, nodes, False), outcome(models, tris, nodes, True)
        assert o1 == o2, (it, kind, o1, o2)
        seen[o1[0]] += 1
    assert seen["ok"] > 10 and seen["err"] > 10, seen


def test_header_is_plain_c_and_links_from_a_c_program(pkg, api, tmp_path):
    """include/rt_abi.h is the boundary a C# / C / Rust host binds: it must compile as C99 (pedantic), and a C program must link
    against libraytrace_hip.so and call it without a GPU (version string, host-side scene validation, the host BVH builder)."""
    import subprocess
    src = tmp_path / "host.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "rt_abi.h"
int main(void)
{
    RtSceneInfo info;
    float verts[9] = {0, 0, 0, 1, 0, 0, 0, 1, 0}, normals[9] = {0, 0, 1, 0, 0, 1, 0, 0, 1};
    int32_t idx[3] = {0, 1, 2};
    RtBVHNode nodes[2];
    RtTriangle tri[1];
    RtBvhStats stats;
    RtModel model;
    int n_nodes = 0;
    if (rt_build_bvh(verts, normals, 3, idx, 3, RT_BVH_QUALITY_HIGH, nodes, &n_nodes, tri, &stats) != RT_OK || n_nodes != 1) return 2;
    memset(&model, 0, sizeof model);
    model.worldToLocal[0] = model.worldToLocal[5] = model.worldToLocal[10] = model.worldToLocal[15] = 1.0f;
    model.localToWorld[0] = model.localToWorld[5] = model.localToWorld[10] = model.localToWorld[15] = 1.0f;
    if (rt_validate_scene(&model, 1, tri, 1, nodes, n_nodes, NULL, 0, &info) != RT_OK || !info.flat) return 3;
    nodes[0].triangleCount = 5; /* a leaf that runs past the triangle buffer */
    if (rt_validate_scene(&model, 1, tri, 1, nodes, n_nodes, NULL, 0, &info) != RT_ERR_SCENE) return 4;
    printf("%s | %s\n", rt_version(), rt_last_error(NULL));
    return 0;
}
''')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "host"
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lraytrace_hip", "-Wl,-rpath," + libdir])
    out = subprocess.check_output([str(exe)], text=True)
    assert "raytrace_hip gfx950" in out and "out of bounds" in out
