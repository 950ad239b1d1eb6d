"""Multi-GPU image tiling: one process per GPU, cyclic 8-row strips, one gather.

The reference is single-GPU.  Every pixel depends only on (id.xy, Resolution,
Frame, renderSeed, params, scene) (RayCompute.compute:13-16, RayCommon.hlsl:550-552),
so the image shards with no data-path communication: each rank renders strips
s with s % world == rank using GLOBAL pixel ids (rt_set_partition), keeps its
tile-local accumulation buffer across frames, and the only exchange is one
gather of the RGBA32F tile buffers to the root at readback — `torch.distributed`
gather, i.e. RCCL over xGMI with backend "nccl", gloo on CPU in the tests.
"""
import numpy as np

STRIP_ROWS = 8


def strips_of(rank, world, height, strip_rows=STRIP_ROWS):
    n_strips = (height + strip_rows - 1) // strip_rows
    return list(range(rank, n_strips, world))


def global_rows_of(rank, world, height, strip_rows=STRIP_ROWS):
    """Global row index of each packed local row of `rank` (same rule as rt_local_to_global_row)."""
    rows = []
    for s in strips_of(rank, world, height, strip_rows):
        rows.extend(range(s * strip_rows, min((s + 1) * strip_rows, height)))
    return np.asarray(rows, dtype=np.int64)


def max_local_rows(world, height, strip_rows=STRIP_ROWS):
    return max(len(global_rows_of(r, world, height, strip_rows)) for r in range(world))


def gather_image(local, rank, world, height, dst=0, strip_rows=STRIP_ROWS, group=None, always_collective=False):
    """Gather the packed per-rank row tiles (torch tensors [rows_r, W, 4]) into the full
    [H, W, 4] image on `dst` (returns None elsewhere).  One collective; ranks with
    fewer rows are padded to the common size."""
    import torch
    import torch.distributed as dist

    if world == 1 and not always_collective:  # (always_collective: a one-rank job still runs the gather — bench.py, RT_BENCH_FORCE_DIST)
        return local
    width = local.shape[1]
    pad_rows = max_local_rows(world, height, strip_rows)
    padded = torch.zeros((pad_rows, width, 4), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    if rank == dst:
        bufs = [torch.empty_like(padded) for _ in range(world)]
        dist.gather(padded, gather_list=bufs, dst=dst, group=group)
        out = torch.empty((height, width, 4), dtype=local.dtype, device=local.device)
        for r in range(world):
            rows = torch.as_tensor(global_rows_of(r, world, height, strip_rows), device=local.device)
            out[rows] = bufs[r][: len(rows)]
        return out
    dist.gather(padded, gather_list=None, dst=dst, group=group)
    return None


class TiledTracer:
    """A HipTracer restricted to this rank's strips, rendering into torch tensors so the
    tiles can be handed to RCCL without a copy."""

    def __init__(self, tracer, rank, world, device, always_collective=False):
        self.always_collective = always_collective
        self.tracer = tracer
        self.rank = rank
        self.world = world
        self.device = device
        self.frame_t = None
        self.accum_t = None
        tracer.set_partition(STRIP_ROWS, rank, world)

    def bind(self, width, height):
        import torch
        rows = self.tracer.local_rows()
        assert rows == len(global_rows_of(self.rank, self.world, height))
        self.frame_t = torch.zeros((rows, width, 4), dtype=torch.float32, device=self.device)
        self.accum_t = torch.zeros((rows, width, 4), dtype=torch.float32, device=self.device)
        torch.cuda.synchronize(self.device)
        self.tracer.bind_render_targets(self.frame_t.data_ptr(), self.accum_t.data_ptr())

    def gather_accumulated(self, height, dst=0, comm_device=None):
        """One collective: the accumulation tiles of all ranks -> full image on `dst`.
        comm_device: where the collective runs (default: the tiles' own device, i.e. RCCL)."""
        self.tracer.synchronize()
        local = self.accum_t if comm_device is None or comm_device == self.accum_t.device else self.accum_t.to(comm_device)
        return gather_image(local, self.rank, self.world, height, dst, always_collective=self.always_collective)
