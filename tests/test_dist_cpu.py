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


def test_one_rank_job_still_runs_the_collective():
    """gather_image(always_collective=True) at world size 1 (bench.py's RT_BENCH_FORCE_DIST hook, which puts RCCL's gather under
    test on a 1-GPU box): the collective runs and the de-interleave returns the tile unchanged."""
    code = (
        "import os, sys, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {os.path.dirname(HERE)!r})\n"
        "import __graft_entry__ as g\n"
        "pkg = g.load_package()\n"
        "dist.init_process_group('gloo', rank=0, world_size=1)\n"
        "t = torch.arange(37 * 5 * 4, dtype=torch.float32).reshape(37, 5, 4)\n"
        "calls = []\n"
        "real = dist.gather\n"
        "def spy(*a, **k):\n"
        "    calls.append(1)\n"
        "    return real(*a, **k)\n"
        "dist.gather = spy\n"
        "out = pkg.dist.gather_image(t, 0, 1, 37, always_collective=True)\n"
        "assert calls == [1] and out is not t and torch.equal(out, t)\n"
        "assert pkg.dist.gather_image(t, 0, 1, 37) is t and calls == [1]\n"
        "dist.destroy_process_group()\n"
        "print('ONE_RANK_OK')\n")
    port = 29500 + (os.getpid() + 7) % 2000
    env = dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0 and "ONE_RANK_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
