#!/usr/bin/env python
"""bench.py — Mrays/s of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: 1920x1080, 8 spp, 8 bounces, 16 analytic
spheres + checkered ground quad, sky on (ray_tracing_amd.scenes.config2).  One
"step" = one frame = one launch of the trace kernel over the whole image (the
reference's RenderFrame, RayComputeManager.cs:84-95); successive steps are
successive Frame indices of a progressive render, exactly like the reference.
Scene and render targets are resident in HBM before the timed region.

"rays" = path segments = CalculateRayCollision calls (RayCommon.hlsl:487),
counted exactly by the kernel.  For N > 1 the image is split into cyclic 8-row
strips (one process per GPU, no data-path collective); the single RCCL gather
of the tiles happens at readback, after the timed steps, and is reported
separately (`gather_ms`).

Scaling is WEAK by default: per-GPU work is fixed at 1920x1080 pixels — the image
area grows with N at the same 16:9 view (N=4 is the north_star's 3840x2160), so
every rank renders ~2.07 Mpixels of the same scene.  `--scaling strong` keeps the
1920x1080 image for every N instead (a pixel's 8 samples x 9 segments are one
serial chain, so a 1/8 image is bounded by that chain).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(pkg, scene_id, width, height, min_s=10.0, max_frames=4, threads=1):
    """The CPU oracle (a port of the reference's loop; ONE thread unless told otherwise) timed on
    a bounded sample of the same workload: whole frames 1, 2, ... of the same scene at the same
    resolution until at least `min_s` seconds of wall time are spent."""
    orc = graft.load_oracle()
    tr = orc.create_tracer(threads=threads)
    sc = pkg.scenes.get(scene_id)
    mgr = sc.make_manager(tr, orc, width, height)
    mgr.OnEnable(renderSeed=1)
    tr.reset_counters()
    t0 = time.perf_counter()
    frames = 0
    while frames < max_frames and (frames == 0 or time.perf_counter() - t0 < min_s):
        mgr.RenderFrame()
        frames += 1
    dt = time.perf_counter() - t0
    c = tr.counters()
    tr.close()
    return {
        "value": c["segments"] / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
        "sample": f"oracle/rt_oracle.cpp (g++ -O2, strict fp32), {threads} thread(s): frames 1..{frames} of the same scene at "
                  f"{width}x{height} ({c['segments']} segments in {dt:.1f} s)",
        "nproc": os.cpu_count(),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, help="scene id (default 2 = the headline workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batched", action="store_true", help="skip the informational rt_render_frames(K) pass (profile runs)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    # Test hooks (single-GPU boxes): RT_BENCH_ONE_DEVICE=1 puts every rank on GPU 0 and
    # RT_BENCH_BACKEND=gloo replaces RCCL, so the N>1 code path can be exercised without N GPUs.
    dev_index = 0 if os.environ.get("RT_BENCH_ONE_DEVICE") else local_rank
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    comm_device = device if backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    pkg = graft.load_package()
    api = pkg.load_library()
    tracer = api.create_tracer(dev_index)
    scene = pkg.scenes.get(args.config)
    W, H = scene.width, scene.height
    if world > 1 and args.scaling == "weak":
        # same view, N x the pixels: both sides scaled by sqrt(N), kept multiples of 8
        f = world ** 0.5
        W, H = int(round(W * f / 8)) * 8, int(round(H * f / 8)) * 8
    tiled = None
    if world > 1:
        tiled = pkg.dist.TiledTracer(tracer, rank, world, device)
    mgr = scene.make_manager(tracer, api, W, H)
    mgr.OnEnable(renderSeed=1)  # resize + BVH build + upload + reset: everything resident in HBM
    if tiled:
        tiled.bind(W, H)
        tracer.reset_accumulation()
    n_models, n_spheres = len(scene.models), len(scene.spheres)

    def barrier():
        if world > 1:
            dist.barrier()
        tracer.synchronize()
        torch.cuda.synchronize()

    # ---- warmup
    for _ in range(args.warmup):
        mgr.RenderFrame()
    barrier()
    first_frame = tracer.frame()

    # ---- timed: exactly K steps
    tracer.reset_counters()
    barrier()
    t0 = time.perf_counter()
    tracer.timer_begin()
    for _ in range(args.steps):
        tracer.render_frame()
    tracer.timer_end()
    barrier()
    t1 = time.perf_counter()
    timed = tracer.counters()
    elapsed = t1 - t0
    segments = timed["segments"]

    # ---- untimed replay of the same K frames with the detailed counters on
    # (identical work: the frame index and seed decide every ray) -> algorithmic bytes
    tracer.reset_counters()
    tracer.enable_stats(True)
    mgr.numAccumulatedFrames = first_frame
    mgr.SetShaderParams()
    tracer.render_frames(args.steps)
    stats = tracer.counters()
    tracer.enable_stats(False)
    assert stats["segments"] == segments, (stats["segments"], segments)

    # ---- informational: the same K frames through the batched API (rt_render_frames: up to 16
    # frames per launch, each pixel runs its frames back to back; identical final buffers)
    batched = None
    if not args.no_batched:
        mgr.numAccumulatedFrames = first_frame
        mgr.SetShaderParams()
        tracer.reset_counters()
        barrier()
        b0 = time.perf_counter()
        tracer.render_frames(args.steps)
        barrier()
        batched_elapsed = time.perf_counter() - b0
        batched = {"what": "rt_render_frames(K): frames fused up to 16 per launch (this rank)",
                   "value": tracer.counters()["segments"] / batched_elapsed / 1e6, "unit": "Mrays/s",
                   "ms_per_frame": batched_elapsed / args.steps * 1e3}

    # ---- roofline pass (rank 0): the timed pass above launches every frame as two kernels on two
    # streams that overlap in time, so "duration of one launch" is not defined there.  The same K
    # frames are rendered once more by a context restricted to ONE kernel per frame on one stream
    # (RT_TWO_STREAMS=0): HIP events around those K launches give the kernel's average launch
    # duration, the figure rocprofv3 --kernel-trace reports for the same mode (profiles/).
    single_ms = None
    if rank == 0:
        if os.environ.get("RT_TWO_STREAMS") == "0":
            single_ms = timed["gpuMs"] / args.steps
        else:
            os.environ["RT_TWO_STREAMS"] = "0"
            t1s = api.create_tracer(dev_index)
            del os.environ["RT_TWO_STREAMS"]
            if tiled:
                t1s.set_partition(pkg.dist.STRIP_ROWS, rank, world)
            m1 = scene.make_manager(t1s, api, W, H)
            m1.OnEnable(renderSeed=1)
            for _ in range(args.warmup):
                m1.RenderFrame()
            t1s.synchronize()
            t1s.reset_counters()
            t1s.timer_begin()
            for _ in range(args.steps):
                t1s.render_frame()
            t1s.timer_end()
            one = t1s.counters()
            assert one["segments"] == segments, (one["segments"], segments)
            single_ms = one["gpuMs"] / args.steps
            t1s.close()

    # ---- readback: the one collective of the multi-GPU path
    gather_ms = None
    if tiled:
        barrier()
        g0 = time.perf_counter()
        full = tiled.gather_accumulated(H, comm_device=comm_device)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        del full

    if world > 1:
        t = torch.tensor([elapsed, float(timed["gpuMs"])], dtype=torch.float64, device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms_max = t[0].item(), t[1].item()
        s = torch.tensor([segments, pkg.abi.algorithmic_bytes(stats, n_models, n_spheres)], dtype=torch.float64, device=comm_device)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        total_segments, total_bytes = s[0].item(), s[1].item()
    else:
        kernel_ms_max = timed["gpuMs"]
        total_segments = float(segments)
        total_bytes = float(pkg.abi.algorithmic_bytes(stats, n_models, n_spheres))

    if rank == 0:
        # roofline of the dominant (only) kernel, rt_trace_kernel: algorithmic bytes of THIS
        # rank's launch / its average duration from HIP events on the launch stream
        my_bytes = pkg.abi.algorithmic_bytes(stats, n_models, n_spheres) / args.steps
        my_ms = single_ms
        achieved = my_bytes / (my_ms * 1e-3) / 1e9
        traffic = None
        prof = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(prof):
            with open(prof) as f:
                pj = json.load(f)
            key = f"config{args.config}_n{world}"
            traffic = pj.get(key, {}).get("hbm_bytes_per_launch")
        out = {
            "metric": "Mrays/s at 1920x1080, 8 spp, 8 bounces; per-channel L2 vs reference",
            "value": total_segments / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{scene.name}: {W}x{H}, {scene.settings['numRaysPerPixel']} spp/frame, "
                            f"{scene.settings['maxBounceCount']} bounces, {n_spheres} spheres + {n_models} models "
                            f"({scene.unique_triangles()} triangles); "
                            + (f"BASELINE.json configs[{args.config - 1}]" if args.config <= 5 else "the reference's own scene file, not a BASELINE config"),
                "parallelism": "single GPU" if world == 1 else f"{W}x{H} image row-tiled, cyclic 8-row strips over {world} GPUs "
                               f"({args.scaling} scaling), RCCL gather at readback",
                "renderSeed": 1, "first_timed_frame": first_frame,
            },
            "segments_per_step": total_segments / args.steps,
            "mpaths_per_s": W * H * scene.settings["numRaysPerPixel"] * args.steps / elapsed / 1e6,
            "resolution": [W, H],
            "kernel_ms_per_step": kernel_ms_max / args.steps,
            "gather_ms": gather_ms,
            "launches": "2 kernels per frame on 2 streams (disjoint tile halves, overlapping in time)"
                        if os.environ.get("RT_TWO_STREAMS") != "0" else "1 kernel per frame",
            "batched_api": batched,
            "parity": "bit-identical to oracle/ on tests/ (pytest -m gpu); max rel err 0",
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": "rt_trace_kernel<false, *> (whole-frame launches of the roofline pass; the two half-frame "
                          "launches per step of the timed pass run the same code as rt_trace_half_kernel)",
                "algorithmic_bytes_per_launch": my_bytes, "avg_launch_ms": my_ms,
                "note": "algorithmic bytes = the reference loop's loads for the counted work (SURVEY.md 8(d)); "
                        "the scene is cache/SGPR resident, so frac can exceed what HBM itself delivers. "
                        "avg_launch_ms: one kernel per frame on one stream (RT_TWO_STREAMS=0 pass), HIP events",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, args.config, W, H)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            # informational: the same port on the host's cores (row bands of the image per thread)
            nthr = min(os.cpu_count() or 1, 64)
            if nthr > 1:
                out["cpu_baseline_threads"] = cpu_baseline(pkg, args.config, W, H, min_s=4.0, max_frames=64, threads=nthr)
        print(json.dumps(out), flush=True)

    tracer.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
