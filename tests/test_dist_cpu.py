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
