"""The N>1 path on CPU: world_size 2 and 3, gloo backend.  Each rank renders only its
cyclic row strips, one gather assembles the image; result == single-process render."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world,cfg", [(2, 2), (3, 3)])
def test_row_tiled_render_equals_single(world, cfg):
    port = 29500 + (os.getpid() + world) % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "_dist_worker.py"), str(cfg)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "DIST_OK" in p.stdout


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
@pytest.mark.parametrize("world,cfg", [(2, 3), (3, 2)])
def test_partitioned_contexts_bound_targets_and_gather_on_the_gpu(world, cfg):
    """rt_set_partition + rt_bind_render_targets + the gather, together, on the real library (one device,
    `world` processes, gloo for the collective) == the oracle's single image."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(HERE, "_dist_worker.py"), "gpu", str(cfg)]
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "DIST_GPU_OK" in p.stdout


@pytest.mark.gpu
def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts the two ranks itself (strong scaling on the
    BASELINE image); on a 1-GPU box both ranks share device 0 and the collective runs over gloo."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, RT_BENCH_ONE_DEVICE="1", RT_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-pmc"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["resolution"] == [1920, 1080]
    assert d["value"] > 0 and d["gather_ms"] is not None and d["steps"] == 3
