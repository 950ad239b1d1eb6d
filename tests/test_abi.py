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
