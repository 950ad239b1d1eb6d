"""A short run of the randomised API soak (tools/soak2.py): random sequences of rt_render_frame bursts, rt_render_frames(n) with every
residue mod 16, reads, resets, updates, resizes — on one context, a partitioned one and through rt_create_multi — under the default schedule
(two streams, fused launches, held-back frames, tile-order learning) against the plainest one: every checkpoint and the final buffers bit for
bit.  The long form found the lost-frame race of round 4 (profiles/r04_soak.txt)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("cfg", [2, 3])
def test_random_api_sequences_do_not_depend_on_the_schedule(cfg):
    env = {k: v for k, v in os.environ.items() if not k.startswith("RT_") or k in ("RT_HIP_LIB",)}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak2.py"), "45", "1", str(cfg)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "SOAK2 OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
