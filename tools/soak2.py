"""Randomised API soak, wider than tools/soak.py: random call sequences (rt_render_frame bursts, rt_render_frames(n) with every residue
mod 16, reads of either buffer, counters, resets, model / sphere / parameter updates, accumulate off and on, resizes, flushes, synchronises, stats on / off, checkpoint
write-back, display reads, scene re-uploads, switches to a caller-provided stream and back)
on one context, on a partitioned context and through rt_create_multi — each run under the default schedule and under the plainest one (one
stream, one launch per frame, identity order, nothing held back); every checkpoint and the final buffers must agree bit for bit.
usage: python tools/soak2.py [rounds=120] [seeds=3] [configs=2,3,6]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
try:  # torch first: it carries its own HIP runtime, and the one loaded first owns the device for the process
    import torch
    caller_stream = torch.cuda.Stream() if torch.cuda.is_available() else None   # a caller-provided stream for rt_set_stream
except Exception:
    caller_stream = None
import __graft_entry__ as g

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
configs = [int(c) for c in (sys.argv[3] if len(sys.argv) > 3 else "2,3,6").split(",")]
pkg = g.load_package(); api = pkg.load_library()
PLAIN = (("RT_TWO_STREAMS", "0"), ("RT_FUSE_FRAMES", "0"), ("RT_LPT", "0"), ("RT_COALESCE", "0"), ("RT_ALTERNATE", "0"))


def run(plain, cfg, seed, mode):
    for k, v in PLAIN:
        if plain:
            os.environ[k] = v
        else:
            os.environ.pop(k, None)
    if mode == "multi":
        tr = api.create_multi_tracer([0, 0, 0])
    else:
        tr = api.create_tracer(0)
    sizes = [(320, 184), (200, 120)]
    w, h = sizes[0]
    if mode == "part":
        tr.set_partition(8, 1, 3)
    sc = pkg.scenes.get(cfg)
    mgr = sc.make_manager(tr, api, w, h)
    mgr.OnEnable(renderSeed=seed)
    rng = np.random.default_rng(seed * 1000 + cfg)
    stats_on, on_caller, bound = [False], [False], [None]
    sig, log = [], []
    frames = 0
    for r in range(rounds):
        how = int(rng.integers(0, 4))
        # (round 5: fused launches hold 16 ... 64 frames, the cap follows the measured frame time: batches around every boundary)
        n = int(rng.integers(1, 40)) if how else int(rng.choice([1, 15, 16, 17, 31, 32, 33, 2, 18, 47, 48, 49, 63, 64, 65, 100, 129]))
        if how == 0:
            mgr.RenderFrames(n)                       # rt_render_frames(n): full fused launches + the remainder
        elif how == 1:
            for _ in range(n):
                mgr.RenderFrame()                     # InitFrame + rt_render_frame, as the reference's Update loop
        else:
            mgr.InitFrame()
            for _ in range(n):
                tr.render_frame()                     # back to back: held back and fused by the library
            if mgr.accumulate:
                mgr.numAccumulatedFrames += n
        frames += n
        ev = int(rng.integers(0, 18))
        log.append((r, how, n, ev))
        if ev == 0:
            sig.append(("acc", r, int(np.ascontiguousarray(tr.read_accumulated()).view(np.uint32).sum(dtype=np.uint64))))
        elif ev == 1:
            sig.append(("frame", r, int(np.ascontiguousarray(tr.read_frame()).view(np.uint32).sum(dtype=np.uint64))))
        elif ev == 2 and len(mgr.models):
            i = int(rng.integers(0, len(mgr.models)))
            mgr.models[i].transform = pkg.Transform(tuple(float(x) for x in rng.uniform(-1.5, 1.5, 3)), (float(rng.uniform(0, 360)), 20.0, 0.0),
                                                    tuple(float(x) for x in rng.uniform(0.5, 1.1, 3)))
            if rng.integers(0, 2):
                mgr.ResetAccumulatedRender()
        elif ev == 3:
            mgr.divergeStrength = float(rng.uniform(0.0, 1.0))
        elif ev == 4 and mode != "multi":
            sig.append(("segments", r, tr.counters()["segments"]))
        elif ev == 5:
            mgr.ResetAccumulatedRender()
        elif ev == 6:
            mgr.accumulate = not mgr.accumulate       # RCM:94: the frame counter stands still while accumulation is off
            mgr.SetShaderParams()
        elif ev == 7 and mode == "single" and bound[0] is None:
            w, h = sizes[1] if (w, h) == sizes[0] else sizes[0]
            mgr.screenSize = (w, h)                   # a resize: new targets, accumulation restarts (InitTexturesAndBuffers, RCM:126-141)
            mgr._sized = False
            mgr.ResetAccumulatedRender()
        elif ev == 8 and hasattr(tr, "synchronize"):
            tr.synchronize()
        elif ev == 9 and len(getattr(mgr, "spheres", [])):
            i = int(rng.integers(0, len(mgr.spheres)))
            mgr.spheres[i].radius = float(rng.uniform(0.2, 0.6))
            mgr.tracer.update_spheres(mgr._pack_spheres())
        elif ev == 10:
            mgr.numRaysPerPixel = int(rng.integers(1, 4))
        elif ev == 11 and mode != "multi":
            stats_on[0] = not stats_on[0]
            tr.enable_stats(stats_on[0])              # the other kernel instantiation (exact counters, audits) from here on
        elif ev == 12 and mode != "multi":
            tr.write_accumulated(tr.read_accumulated())   # checkpoint / resume: the sum goes out and comes back
        elif ev == 13 and mode != "multi":
            img = tr.display_srgb8(max(1, mgr.numAccumulatedFrames - 1)) if r % 2 else tr.display(max(1, mgr.numAccumulatedFrames - 1))
            sig.append(("display", r, int(np.ascontiguousarray(img).view(np.uint8).sum(dtype=np.uint64))))
        elif ev == 14:
            mgr.hasBVH = False                        # the scene again: CreateAllMeshData + rt_upload_scene with frames possibly in flight
            mgr.InitFrame()
        elif ev == 15 and mode == "single" and caller_stream is not None:
            on_caller[0] = not on_caller[0]
            tr.set_stream(caller_stream.cuda_stream if on_caller[0] else None)   # the caller's stream order is then the contract
        elif ev == 16 and hasattr(tr, "flush"):
            tr.flush()
        elif ev == 17 and mode != "multi" and caller_stream is not None:
            if bound[0] is None:                      # caller-owned render targets (e.g. tensors for an RCCL gather), frames possibly in flight
                rows = tr.local_rows()
                bound[0] = (torch.empty((rows, w, 4), dtype=torch.float32, device="cuda"), torch.empty((rows, w, 4), dtype=torch.float32, device="cuda"))
                tr.bind_render_targets(bound[0][0].data_ptr(), bound[0][1].data_ptr())
            else:
                tr.bind_render_targets(None, None)    # back to the library's own buffers
                tr.synchronize()
                bound[0] = None
            mgr.ResetAccumulatedRender()              # the new targets hold nothing yet
    acc = np.ascontiguousarray(tr.read_accumulated()).copy(); frm = np.ascontiguousarray(tr.read_frame()).copy()
    if on_caller[0]:
        tr.set_stream(None)
    if bound[0] is not None:
        tr.bind_render_targets(None, None)
        tr.synchronize()
    tr.close()
    return acc, frm, sig, frames, log


bad = 0
t0 = time.time()
for cfg in configs:
    for seed in range(1, seeds + 1):
        for mode in ("single", "part", "multi"):
            a, fa, sa, n, log = run(False, cfg, seed, mode)
            b, fb, sb, n2, _ = run(True, cfg, seed, mode)
            ok = n == n2 and a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(fa.view(np.uint32), fb.view(np.uint32)) and sa == sb
            if not ok:
                bad += 1
                first = next((i for i, (x, y) in enumerate(zip(sa, sb)) if x != y), None)
                print(f"MISMATCH config {cfg} seed {seed} {mode}: frames {n}/{n2}, final acc equal {np.array_equal(a.view(np.uint32), b.view(np.uint32))}, "
                      f"frame equal {np.array_equal(fa.view(np.uint32), fb.view(np.uint32))}, first differing checkpoint {first}: {sa[first] if first is not None else None} vs {sb[first] if first is not None else None}")
                if first is not None:
                    r0 = sa[first][1]
                    print("   calls before it (round, how[0=batch,1=RenderFrame,2=burst], n, event):", [e for e in log if r0 - 6 <= e[0] <= r0])
            else:
                print(f"ok config {cfg} seed {seed} {mode}: {n} frames, {len(sa)} checkpoints")
print(f"SOAK2 {'OK' if not bad else 'MISMATCH in %d runs' % bad} ({time.time() - t0:.0f} s)")
sys.exit(1 if bad else 0)
