"""Worker for tests/test_dist_cpu.py: rank r of a world_size-N gloo group renders ONLY its
cyclic 8-row strips (CPU oracle stands in for the GPU), the tiles are gathered with
ray_tracing_amd.dist.gather_image, and rank 0 checks the result against a plain full render."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    W, H, frames, cfg = 40, 45, 2, int(sys.argv[1])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = graft.load_package()
    orc = graft.load_oracle()
    tr = orc.create_tracer(1)
    mgr = pkg.scenes.get(cfg).make_manager(tr, orc, W, H)
    mgr.OnEnable(renderSeed=3)
    rows = pkg.dist.global_rows_of(rank, world, H)
    for f in range(frames):
        for s in pkg.dist.strips_of(rank, world, H):
            mgr.numAccumulatedFrames = 1 + f
            mgr.SetShaderParams()
            orc.set_row_window(tr.h, s * 8, min(s * 8 + 8, H))
            tr.render_frame()
    full_local = tr.read_accumulated()
    # rows this rank does not own were never rendered
    other = np.setdiff1d(np.arange(H), rows)
    assert np.all(full_local[other] == 0)
    local = torch.from_numpy(np.ascontiguousarray(full_local[rows]))
    img = pkg.dist.gather_image(local, rank, world, H, dst=0)
    segs = torch.tensor([tr.counters()["segments"]], dtype=torch.float64)
    dist.all_reduce(segs)
    if rank == 0:
        ref = orc.create_tracer(2)
        m2 = pkg.scenes.get(cfg).make_manager(ref, orc, W, H)
        m2.OnEnable(renderSeed=3)
        m2.RenderFrames(frames)
        want = ref.read_accumulated()
        got = img.numpy()
        assert got.shape == want.shape
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "tiled render != single render"
        assert int(segs.item()) == ref.counters()["segments"]
        print("DIST_OK", world, int(segs.item()))
    else:
        assert img is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
