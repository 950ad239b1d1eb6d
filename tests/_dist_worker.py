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


def main_gpu():
    """The real library on one device: every rank = a context restricted by rt_set_partition, rendering into
    torch tensors bound with rt_bind_render_targets, tiles gathered by ray_tracing_amd.dist (gloo here because
    RCCL refuses two ranks on one GPU) — the exact code path of `bench.py --gpus N`, checked against the oracle."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    W, H, frames, cfg = 104, 77, 3, int(sys.argv[2])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = graft.load_package()
    api = pkg.load_library()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    tracer = api.create_tracer(0)
    tiled = pkg.dist.TiledTracer(tracer, rank, world, device)
    mgr = pkg.scenes.get(cfg).make_manager(tracer, api, W, H)
    mgr.OnEnable(renderSeed=3)
    tiled.bind(W, H)
    tracer.reset_accumulation()
    for _ in range(frames):
        mgr.RenderFrame()
    img = tiled.gather_accumulated(H, comm_device=torch.device("cpu"))
    segs = torch.tensor([tracer.counters()["segments"]], dtype=torch.float64)
    dist.all_reduce(segs)
    if rank == 0:
        orc = graft.load_oracle()
        ref = orc.create_tracer(4)
        m2 = pkg.scenes.get(cfg).make_manager(ref, orc, W, H)
        m2.OnEnable(renderSeed=3)
        m2.RenderFrames(frames)
        want = ref.read_accumulated()
        got = img.numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "tiled GPU render != oracle"
        assert int(segs.item()) == ref.counters()["segments"]
        print("DIST_GPU_OK", world, int(segs.item()))
    tracer.close()
    dist.barrier()
    dist.destroy_process_group()


def main():
    if sys.argv[1] == "gpu":
        return main_gpu()
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
