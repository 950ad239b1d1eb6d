"""host/cpp: the compiled host above the C ABI (RayComputeManager mirror in C++ + rt_bench CLI).
CPU: it builds, and the buffers its CreateAllMeshData / ShaderParams produce are byte-identical to
the Python host's.  GPU: the image it renders equals the oracle's render of the buffers it uploaded."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "host", "cpp", "rt_bench")


@pytest.fixture(scope="module")
def bench(api):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host", "cpp")], stdout=subprocess.DEVNULL)
    assert os.path.exists(BENCH)
    return BENCH


class _Null:
    def __getattr__(self, n):
        return lambda *a: None


def _load(pkg, prefix):
    a = pkg.abi
    return (np.fromfile(prefix + ".models.bin", dtype=a.model_dtype), np.fromfile(prefix + ".triangles.bin", dtype=a.triangle_dtype),
            np.fromfile(prefix + ".nodes.bin", dtype=a.node_dtype), np.fromfile(prefix + ".spheres.bin", dtype=a.sphere_dtype),
            a.RtParams.from_buffer_copy(open(prefix + ".params.bin", "rb").read()))


@pytest.mark.parametrize("cfg", [2, 3])
def test_cpp_host_buffers_equal_python_host(pkg, api, bench, tmp_path, cfg):
    prefix = str(tmp_path / f"c{cfg}")
    subprocess.check_call([bench, "--config", str(cfg), "--width", "64", "--height", "36", "--scene-only", "--dump", prefix])
    models, tris, nodes, spheres, params = _load(pkg, prefix)
    mgr = pkg.scenes.get(cfg).make_manager(_Null(), api, 64, 36)
    mgr.renderSeed, mgr.numAccumulatedFrames = 1, 1
    d = mgr.CreateAllMeshData(mgr.models)
    assert tris.tobytes() == d["triangles"].tobytes() and nodes.tobytes() == d["nodes"].tobytes()
    assert models.tobytes() == d["meshInfo"].tobytes()
    assert spheres.tobytes() == mgr._pack_spheres().tobytes()
    p = mgr.params()
    p.abi_version, p.struct_size = 1, C.sizeof(pkg.abi.RtParams)
    assert bytes(p) == bytes(params)


def test_cpp_host_fails_loudly_without_gpu(bench):
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    r = subprocess.run([bench, "--config", "2", "--width", "32", "--height", "32", "--frames", "1"], capture_output=True, text=True)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_cpp_host_gpu_bvh_batch_gives_the_same_buffers(pkg, bench, tmp_path):
    """--bvh-gpu: CreateAllMeshData through one rt_build_bvh_gpu_batch call (the scene's meshes as one forest) — the buffers the
    compiled host would upload are byte for byte the ones its host-builder path makes."""
    a, b = str(tmp_path / "host"), str(tmp_path / "gpu")
    subprocess.check_call([bench, "--config", "3", "--width", "64", "--height", "36", "--scene-only", "--dump", a])
    subprocess.check_call([bench, "--config", "3", "--width", "64", "--height", "36", "--scene-only", "--bvh-gpu", "--dump", b])
    for x, y in zip(_load(pkg, a)[:4], _load(pkg, b)[:4]):
        assert x.tobytes() == y.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [2, 3])
def test_cpp_host_render_equals_oracle(pkg, orc, bench, tmp_path, cfg):
    prefix = str(tmp_path / f"g{cfg}")
    out = subprocess.check_output([bench, "--config", str(cfg), "--width", "96", "--height", "54", "--frames", "3", "--warmup", "1",
                                   "--dump", prefix], text=True)
    info = json.loads(out.strip().splitlines()[-1])
    assert info["frames"] == 3 and info["segments"] > 0
    models, tris, nodes, spheres, params = _load(pkg, prefix)
    got = np.fromfile(prefix + ".accumulated.bin", dtype=np.float32).reshape(54, 96, 4)
    tr = orc.create_tracer(8)
    tr.resize(96, 54)
    tr.upload_scene(models, tris, nodes, spheres)
    tr.set_params(params)
    tr.reset_accumulation()
    tr.render_frames(3)
    want = tr.read_accumulated()
    assert tr.counters()["segments"] == info["segments"]
    tr.close()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_cpp_host_several_devices_from_one_process(pkg, orc, bench, tmp_path, devices):
    """rt_create_multi through the compiled host: one context per listed device (the same GPU several times
    = virtual shards), cyclic 8-row strips, rt_gather_accumulated -> the oracle's image, bit for bit."""
    prefix = str(tmp_path / "m3")
    out = subprocess.check_output([bench, "--config", "3", "--width", "96", "--height", "70", "--frames", "3", "--devices", devices,
                                   "--dump", prefix], text=True)
    info = json.loads(out.strip().splitlines()[-1])
    assert info["devices"] == len(devices.split(",")) and info["segments"] > 0
    models, tris, nodes, spheres, params = _load(pkg, prefix)
    got = np.fromfile(prefix + ".accumulated.bin", dtype=np.float32).reshape(70, 96, 4)
    tr = orc.create_tracer(8)
    tr.resize(96, 70)
    tr.upload_scene(models, tris, nodes, spheres)
    tr.set_params(params)
    tr.reset_accumulation()
    tr.render_frames(3)
    want = tr.read_accumulated()
    assert tr.counters()["segments"] == info["segments"]
    tr.close()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
