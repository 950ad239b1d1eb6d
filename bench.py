#!/usr/bin/env python
"""bench.py — Mrays/s of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config C] [--scaling strong|weak]

Workload (default) = BASELINE.json configs[1]: 1920x1080, 8 spp, 8 bounces, 16 analytic spheres +
checkered ground quad, sky on (ray_tracing_amd.scenes.config2).  `--config 5` is the north_star's
8-GPU case (3840x2160, 12 bounces, 983k triangles).  One "step" = one frame = the reference's Dispatch
of the whole image (RayComputeManager.cs:84-95); successive steps are successive Frame indices of a
progressive render.  Scene and render targets are resident in HBM before the timed region.

"rays" = path segments = CalculateRayCollision calls (RayCommon.hlsl:487), counted exactly by the kernel.
`value` times K x rt_render_frame (the Dispatch only) issued back to back — the library holds frames requested
while earlier ones still execute back (16 ... 64: a budget of ~20 ms per launch at the measured frame time) and launches them
fused, bit-identical to one launch per frame;
`value_one_kernel_per_frame` is the same work at exactly one kernel per frame (the roofline pass);
`value_with_initframe` times K x the mirror's RenderFrame() = InitFrame (UpdateModels + SetShaderParams every
frame, RCM:115-124) + Dispatch.

N > 1: one process per GPU.  With WORLD_SIZE unset, `python bench.py --gpus N` spawns the N ranks itself
(torch.distributed.run, 127.0.0.1); under torch.distributed.run it is one rank.  The image is split into
cyclic 8-row strips (no data-path collective), scaling is STRONG by default (the BASELINE image is fixed,
each rank renders 1/N of its rows); the one RCCL gather of the accumulation tiles happens at readback,
after the timed steps, and is reported as `gather_ms`.  `--scaling weak` grows the image with N instead.

roofline (rank 0, N = 1): the trace kernels are bound by instruction issue at the occupancy their registers allow, not by memory
(round 6 removed 39-47 % of the BVH kernels' L1 accesses with an LDS top-of-tree cache and their frame time moved by
1-2 %: `roofline.memory_path` — L1 accesses per clock per CU, TA busy — is an observation, not a roof), so
`bound` = "valu": achieved = SQ_INSTS_VALU per launch / average launch time, peak = 256 CU x 4 SIMD x
2.4 GHz / 2 cycles per wave64 instruction, frac = achieved/peak x lane utilisation
(SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU)).  A "launch" is the dominant one of the timed region: the
fused launch of up to 16 frames that back-to-back rt_render_frame calls leave as (`frames_per_launch`).  The
counters come from rocprofv3 PMC passes over a child run of THIS script in the same invocation (separate passes
for the SQ counters, FETCH_SIZE and WRITE_SIZE); if rocprofv3 is unavailable the committed profiles/ summary
is replayed and labelled so.

`value` is the driver's contract and nothing else: W warm-up steps, a barrier, exactly K timed steps, first timed Frame = W + 1 on every
rank (asserted with an all_gather).  There is no hidden spin-up (round 5's wall-clock-bounded one made ranks render different frame
counts).  This GPU's clocks do ramp for 20-30 ms after idling (profiles/r05_clock_ramp.txt), so a 13 ms region scatters by
+-5 %: the line therefore ALSO carries `regions` — R (default 7) further back-to-back regions of the same progressive render, each at
least K steps and at least 50 ms long, each bracketed like the timed one — and `value_median_of_regions`, the figure A/B comparisons
quote.  The number of steps per region is derived from an all-reduced time, so every rank renders the same frames.

Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import json
import math
import os
import shutil
import socket
import sqlite3
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK = 256 * 4 * 2.4e9 / 2.0          # wave64 VALU instructions / s: 256 CUs x 4 SIMD-32, 2 cycles each
N_SIMD, CLOCK_HZ = 256 * 4, 2.4e9
N_CU = 256
# The vector-memory path (round 5; tools/ubench/vmem_gather.hip, profiles/r05_memory_path.txt): a CU's L1 (TCP) takes about ONE
# access per clock — a wave64 load is charged one access per group of neighbouring lanes that fall into one 128-byte line, 64 accesses in
# 64.4 clocks when every lane has its own line — and no fewer than ~16 clocks per load instruction.  Round 5 read the BVH kernels' 0.83-0.92
# as their roof; round 6's LDS top-of-tree cache took them to 0.50-0.53 with 1-2 % less time (profiles/r06_what_binds_the_bvh_kernels.txt):
# reported as an observation, `binding: False`.
L1_ACCESS_PEAK = N_CU * CLOCK_HZ * 1.0
KERNEL_LIKE = "%rt_trace%kernel<false%"    # the non-stats instantiations (whole-frame and half-frame names)


# --------------------------------------------------------------------------- helpers
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one per GPU) ourselves."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def pin_to_gpu_numa_node(torch, dev_index):
    """One process per GPU: keep this rank's host threads on the cores of its GPU's NUMA node (the launch calls and the
    gather's staging then stay on the socket the GPU hangs off).  Best effort: returns the CPU list used, or None."""
    try:
        pr = torch.cuda.get_device_properties(dev_index)  # torch exposes PCI domain / bus / device as integers
        bdf = "%04x:%02x:%02x.0" % (int(getattr(pr, "pci_domain_id", 0)), int(pr.pci_bus_id), int(pr.pci_device_id))
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read().strip())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return f"numa node {node}: {len(cpus)} cpus"
    except Exception:
        return None


def cpu_baseline(pkg, scene_id, width, height, min_s=10.0, max_frames=4, threads=1):
    """The CPU oracle (a port of the reference's loop) on a bounded sample of the same workload: whole
    frames 1, 2, ... of the same scene at the same resolution until `min_s` seconds are spent.  The
    single-thread figure is pinned to one core (the `taskset -c <cpu>` of SURVEY.md §8(d))."""
    orc = graft.load_oracle()
    pinned = None
    old = None
    if threads == 1 and hasattr(os, "sched_setaffinity"):
        old = os.sched_getaffinity(0)
        pinned = min(old)
        os.sched_setaffinity(0, {pinned})
    try:
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
    finally:
        if old is not None:
            os.sched_setaffinity(0, old)
    return {
        "value": c["segments"] / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
        "sample": f"oracle/rt_oracle.cpp (g++ -O2, strict fp32), {threads} thread(s)"
                  + (f" pinned to cpu {pinned}" if pinned is not None else "")
                  + f": frames 1..{frames} of the same scene at {width}x{height} ({c['segments']} segments in {dt:.1f} s)",
        "nproc": os.cpu_count(),
    }


def cpu_baseline_reference(pkg, scene_id, width, height, min_s=8.0, max_frames=2):
    """The REFERENCE'S OWN loop on one pinned host core: oracle/_ref/libref.so is the reference's shader text (RayCommon.hlsl +
    RayCompute.compute) compiled as C++ by oracle/make_ref.py in the container that has the reference checkout; it travels to the GPU box
    as a prebuilt library.  A scene with analytic spheres (the headline) is timed on libref_spheres.so — the same text + the ONE declared
    sphere hook S1 (the reference's own RaySphere called from the commented call site RC:341) — and says so: kind "reference+S1".
    None when the library is not there."""
    sc = pkg.scenes.get(scene_id)
    variant = "spheres" if sc.spheres else ""
    ref = graft.load_ref(variant)
    if ref is None:
        return None
    orc = graft.load_oracle()  # BVH builder + camera helper of the manager mirror (BVH.cs is C#, not part of the shader text)
    old = os.sched_getaffinity(0) if hasattr(os, "sched_setaffinity") else None
    pinned = min(old) if old else None
    if old:
        os.sched_setaffinity(0, {pinned})
    try:
        tr = ref.create_tracer(threads=1)
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
    finally:
        if old:
            os.sched_setaffinity(0, old)
    lib = "libref_spheres.so (the reference's HLSL text + the declared sphere hook S1, compiled as C++" if variant else "libref.so (the reference's HLSL text compiled as C++"
    return {"value": c["segments"] / dt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "reference+S1" if variant else "reference",
            "sample": f"oracle/_ref/{lib}, g++ -O2, strict fp32), 1 thread pinned to cpu {pinned}: "
                      f"frames 1..{frames} of the same scene at {width}x{height} ({c['segments']} segments in {dt:.1f} s)",
            "nproc": os.cpu_count()}


def parity_check(pkg, api, dev_index, scene_id, width, height, strips):
    """In-run parity: frame 1 of the benchmarked scene at the benchmarked size on a fresh context, 8-row
    strips re-rendered by the oracle — bitwise comparison + per-channel relative L2 of the strips — and, where the library
    travelled, by the REFERENCE'S OWN TEXT: oracle/_ref/libref.so for a model-only scene, libref_spheres.so (that text + the one
    declared sphere hook of oracle/make_ref.py) for a scene with analytic spheres, i.e. the headline config."""
    import numpy as np
    tr = api.create_tracer(dev_index)
    scene = pkg.scenes.get(scene_id)
    mgr = scene.make_manager(tr, api, width, height)
    mgr.OnEnable(renderSeed=1)
    mgr.RenderFrame()
    gpu = tr.read_accumulated()
    tr.close()
    rows = np.concatenate([np.arange(s * 8, min(height, s * 8 + 8)) for s in strips])
    orc = graft.load_oracle()

    def strips_of(lib, threads):
        c = lib.create_tracer(threads)
        m2 = pkg.scenes.get(scene_id).make_manager(c, orc, width, height)  # (BVHs from the oracle's builder either way)
        m2.OnEnable(renderSeed=1)
        for s in strips:
            m2.numAccumulatedFrames = 1
            m2.SetShaderParams()
            lib.set_row_window(c.h, s * 8, min(height, s * 8 + 8))
            c.render_frame()
        img = c.read_accumulated()
        c.close()
        return img[rows]

    def compare(a, b):
        ident = bool(np.array_equal(a.view(np.uint32), b.view(np.uint32)))
        l2 = [float(np.sqrt(np.sum((a[..., k].astype(np.float64) - b[..., k]) ** 2) / max(float(np.sum(b[..., k].astype(np.float64) ** 2)), 1e-300)))
              for k in range(3)]
        mx = float(np.max(np.abs(a.astype(np.float64) - b) / np.maximum(np.abs(b.astype(np.float64)), 1e-30)))
        return ident, l2, mx

    ident, l2, mx = compare(gpu[rows], strips_of(orc, min(os.cpu_count() or 1, 16)))
    out = {"checked_in_this_run": f"frame 1 at {width}x{height}, 8-row strips {list(strips)} vs oracle/ ({len(rows) * width} pixels)",
           "bit_identical": ident, "rel_l2_per_channel": l2, "max_rel_err": mx}
    try:
        variant = "spheres" if scene.spheres else ""
        ref = graft.load_ref(variant)   # raises when the travelling library is stale against oracle/REF_EXPECTED.json
        if ref is not None:
            ri, rl2, rmx = compare(gpu[rows], strips_of(ref, min(os.cpu_count() or 1, 64)))
            out["vs_reference_text"] = {
                "library": "oracle/_ref/" + ("libref_spheres.so (the reference's HLSL text + the declared sphere hook, make_ref.py S1)" if variant
                                             else "libref.so (the reference's HLSL text compiled as C++)"),
                "same_strips": True, "bit_identical": ri, "rel_l2_per_channel": rl2, "max_rel_err": rmx}
    except Exception as e:  # the oracle comparison above stands; say why the reference-text leg is missing
        out["vs_reference_text"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def oracle_strips_check(pkg, scene_id, width, height, image, frames, strips):
    """`image` (H, W, 4: an accumulation buffer after `frames` frames from a reset, renderSeed 1) against the CPU oracle on
    the given 8-row strips: bitwise + per-channel relative L2."""
    import numpy as np
    orc = graft.load_oracle()
    c = orc.create_tracer(min(os.cpu_count() or 1, 16))
    m2 = pkg.scenes.get(scene_id).make_manager(c, orc, width, height)
    m2.OnEnable(renderSeed=1)
    for f in range(frames):
        for s in strips:
            m2.numAccumulatedFrames = 1 + f
            m2.SetShaderParams()
            orc.set_row_window(c.h, s * 8, min(height, s * 8 + 8))
            c.render_frame()
    cpu = c.read_accumulated()
    c.close()
    rows = np.concatenate([np.arange(s * 8, min(height, s * 8 + 8)) for s in strips])
    a, b = np.ascontiguousarray(image[rows]), np.ascontiguousarray(cpu[rows])
    ident = bool(np.array_equal(a.view(np.uint32), b.view(np.uint32)))
    l2 = [float(np.sqrt(np.sum((a[..., k].astype(np.float64) - b[..., k]) ** 2) / max(float(np.sum(b[..., k].astype(np.float64) ** 2)), 1e-300)))
          for k in range(3)]
    return {"what": f"frames 1..{frames} from a reset at {width}x{height}, 8-row strips {list(strips)} of the GATHERED image vs oracle/ "
                    f"({len(rows) * width} pixels)", "bit_identical": ident, "rel_l2_per_channel": l2}


# --------------------------------------------------------------------------- PMC passes (rocprofv3 around a child run)
def pmc_child(args):
    """Child of the PMC passes: one kernel per frame on one stream, `steps` frames, nothing else (no torch)."""
    os.environ["RT_TWO_STREAMS"] = "0"
    os.environ["RT_COALESCE"] = "0"
    pkg = graft.load_package()
    api = pkg.load_library()
    tr = api.create_tracer(0)
    sc = pkg.scenes.get(args.config)
    mgr = sc.make_manager(tr, api)
    mgr.OnEnable(renderSeed=1)
    for _ in range(args.warmup + args.steps):   # every trace-kernel dispatch of the child is one fused launch of --frames-per-launch frames
        tr.render_frames(args.frames_per_launch)
    tr.synchronize()
    tr.close()


def run_pmc_pass(counters, config, steps, warmup, outdir, tag, fpl=16):
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    d = os.path.join(outdir, tag)
    cmd = [exe, "--pmc"] + counters + ["--kernel-trace", "-d", d, "-o", "pmc", "--",
                                       sys.executable, os.path.abspath(__file__), "--pmc-child", "--config", str(config),
                                       "--steps", str(steps), "--warmup", str(warmup), "--frames-per-launch", str(fpl)]
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    except Exception:
        return None
    if r.returncode != 0:
        return None
    res = {}
    for db in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        cur = sqlite3.connect(db).cursor()
        q = ("select counter_name, count(*), avg(v) from (select counter_name, dispatch_id, sum(value) as v from counters_collection "
             f"where kernel_name like '{KERNEL_LIKE}' group by counter_name, dispatch_id) group by counter_name")
        try:
            for name, n, avg in cur.execute(q):
                res[name] = {"n": n, "avg": avg}
            # only the last `steps` dispatches are the measured ones, but warmup frames do the same work: averaged together
        except sqlite3.Error:
            return None
    return res or None


def collect_pmc(config, steps, warmup, with_traffic, fpl=16):
    """SQ pass (+ FETCH_SIZE and WRITE_SIZE passes) over child runs of this script; None if rocprofv3 cannot run."""
    out = tempfile.mkdtemp(prefix="rt_pmc_", dir="/tmp")
    try:
        sq = run_pmc_pass(["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVES", "SQ_BUSY_CYCLES"], config, steps, warmup, out, "sq", fpl)
        if not sq or "SQ_INSTS_VALU" not in sq:
            return None
        r = {"valu_insts_per_launch": sq["SQ_INSTS_VALU"]["avg"],
             "lane_util": sq["SQ_THREAD_CYCLES_VALU"]["avg"] / (64.0 * sq["SQ_ACTIVE_INST_VALU"]["avg"]),
             "launches_sampled": sq["SQ_INSTS_VALU"]["n"]}
        # dynamic VALU instruction types (two passes: the SQ block counts up to 8 at a time, kept small for safety)
        m1 = run_pmc_pass(["SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32"], config, steps, warmup, out, "m1", fpl)
        m2 = run_pmc_pass(["SQ_INSTS_VALU", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT"], config, steps, warmup, out, "m2", fpl)
        if m1 and m2 and "SQ_INSTS_VALU_TRANS_F32" in m1 and "SQ_INSTS_VALU_CVT" in m2:
            r["valu_types"] = {"arith": m1["SQ_INSTS_VALU_ADD_F32"]["avg"] + m1["SQ_INSTS_VALU_MUL_F32"]["avg"] + m1["SQ_INSTS_VALU_FMA_F32"]["avg"],
                               "trans": m1["SQ_INSTS_VALU_TRANS_F32"]["avg"], "int32": m2["SQ_INSTS_VALU_INT32"]["avg"],
                               "int64": m2["SQ_INSTS_VALU_INT64"]["avg"], "cvt": m2["SQ_INSTS_VALU_CVT"]["avg"]}
        # the vector-memory path (round 5): L1 (TCP) accesses and TA busy cycles per launch, with the launch's own cycle count
        mp = run_pmc_pass(["TCP_TOTAL_CACHE_ACCESSES_sum", "TA_BUSY_avr", "GRBM_GUI_ACTIVE", "SQ_INSTS_VMEM_RD"], config, steps, warmup, out, "mp", fpl)
        if mp and "TCP_TOTAL_CACHE_ACCESSES_sum" in mp and "GRBM_GUI_ACTIVE" in mp:
            cycles = mp["GRBM_GUI_ACTIVE"]["avg"] / 8.0          # the counter sums the 8 XCDs
            r["memory_path"] = {"l1_accesses_per_launch": mp["TCP_TOTAL_CACHE_ACCESSES_sum"]["avg"], "gpu_cycles_per_launch": cycles,
                                "l1_accesses_per_clk_per_cu": mp["TCP_TOTAL_CACHE_ACCESSES_sum"]["avg"] / N_CU / cycles,
                                "ta_busy": (mp["TA_BUSY_avr"]["avg"] / cycles) if "TA_BUSY_avr" in mp else None,
                                "l1_accesses_per_vmem_read_inst": (mp["TCP_TOTAL_CACHE_ACCESSES_sum"]["avg"] / mp["SQ_INSTS_VMEM_RD"]["avg"]) if mp.get("SQ_INSTS_VMEM_RD", {}).get("avg") else None}
        if with_traffic:
            rd = run_pmc_pass(["FETCH_SIZE"], config, steps, warmup, out, "rd", fpl)
            wr = run_pmc_pass(["WRITE_SIZE"], config, steps, warmup, out, "wr", fpl)
            if rd and wr and "FETCH_SIZE" in rd and "WRITE_SIZE" in wr:
                # MI355X_MICROARCH.md §HBM: KiB units; gfx950 FETCH_SIZE adds 64 B per 128-B line fill -> x2, calibrated for streaming AND
                # for the traversal's divergent 64-B / 48-B per-lane record fetches (profiles/r04_fetch_size_calibration.txt); WRITE_SIZE as is
                r["hbm_read_bytes"] = rd["FETCH_SIZE"]["avg"] * 1024 * 2
                r["hbm_write_bytes"] = wr["WRITE_SIZE"]["avg"] * 1024
        return r
    finally:
        shutil.rmtree(out, ignore_errors=True)


def replayed_pmc(config, fpl=16):
    prof = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if not os.path.exists(prof):
        return None
    with open(prof) as f:
        p = json.load(f).get(f"config{config}_n1")
    if not p or p.get("frames_per_launch", 1) != fpl:
        return None
    r = {"valu_insts_per_launch": p["valu_insts"], "lane_util": p["valu_lane_utilisation"],
         "hbm_read_bytes": p.get("read_bytes"), "hbm_write_bytes": p.get("write_bytes"),
         "replayed_from": "profiles/pmc_summary.json (" + p.get("source", "?") + ")"}
    if p.get("valu_types"):
        r["valu_types"] = p["valu_types"]
    return r


def launch_profile(pkg, api, dev_index, scene_id, W, H, launches, frames_per_launch, partition=None):
    """Average duration of one trace-kernel launch of `frames_per_launch` frames, one kernel per launch on one stream
    (RT_TWO_STREAMS=0, RT_COALESCE=0), HIP events on the launch stream.  Returns (ms per launch, segments per launch)."""
    prev = {k: os.environ.get(k) for k in ("RT_TWO_STREAMS", "RT_COALESCE")}
    os.environ["RT_TWO_STREAMS"] = "0"
    os.environ["RT_COALESCE"] = "0"
    try:
        t = api.create_tracer(dev_index)
    finally:
        for k, v in prev.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    if partition:
        t.set_partition(*partition)
    sc = pkg.scenes.get(scene_id)
    m = sc.make_manager(t, api, W, H)
    m.OnEnable(renderSeed=1)
    t.render_frames(frames_per_launch)
    t.synchronize()
    t.reset_counters()
    t.timer_begin()
    for _ in range(launches):
        t.render_frames(frames_per_launch)
    t.timer_end()
    c = t.counters()
    t.close()
    return c["gpuMs"] / launches, c["segments"] / launches


def memory_path_roofline(pmc, launch_ms):
    """The vector-memory path of the trace kernels: L1 accesses per second against one access per clock per CU (see L1_ACCESS_PEAK).
    NOT the binding roof (round 6: the LDS top-of-tree cache removed 39-47 % of these accesses, the frame time moved by 1-2 %)."""
    mp = pmc.get("memory_path")
    if not mp:
        return None
    achieved = mp["l1_accesses_per_launch"] / (launch_ms * 1e-3)
    return {"bound": "l1_access", "binding": False, "achieved": achieved / 1e9, "peak": L1_ACCESS_PEAK / 1e9, "unit": "G L1 accesses/s", "frac": achieved / L1_ACCESS_PEAK,
            "frac_at_measured_clock": mp["l1_accesses_per_clk_per_cu"], "ta_busy": mp["ta_busy"],
            "l1_accesses_per_vmem_read_inst": mp["l1_accesses_per_vmem_read_inst"], "l1_accesses_per_launch": mp["l1_accesses_per_launch"],
            "peak_derivation": "256 CUs x 2.4 GHz x 1 L1 (TCP) access per clock: tools/ubench/vmem_gather.hip measures 64 accesses of a wave64 global_load_dwordx4 "
                               "(every lane its own 128-byte line) in 64.4 clocks and one access per group of neighbouring lanes in one line (profiles/r05_vmem_gather_ubench.txt, "
                               "r05_vmem_gather_counters.txt); frac_at_measured_clock uses the launch's own GRBM_GUI_ACTIVE cycles instead of 2.4 GHz"}


def valu_roofline(pmc, launch_ms, segments_per_launch, variant="bvh"):
    achieved = pmc["valu_insts_per_launch"] / (launch_ms * 1e-3)
    r = {"bound": "valu", "achieved": achieved / 1e9, "peak": VALU_PEAK / 1e9, "unit": "Gwave-inst/s",
         "lane_util": pmc["lane_util"], "frac": achieved / VALU_PEAK * pmc["lane_util"],
         "valu_busy": achieved / VALU_PEAK, "valu_insts_per_launch": pmc["valu_insts_per_launch"], "avg_launch_ms": launch_ms,
         "valu_insts_per_segment": pmc["valu_insts_per_launch"] / max(1, segments_per_launch),
         "counters": "replayed: " + pmc["replayed_from"] if "replayed_from" in pmc else
                     f"rocprofv3 --pmc passes over a child run of this script, in this invocation ({pmc.get('launches_sampled')} launches)"}
    if pmc.get("valu_types"):   # the hardware's dynamic instruction types, as shares (informational)
        total = pmc["valu_insts_per_launch"]
        shares = {k: v / max(1.0, total) for k, v in pmc["valu_types"].items()}
        shares["other"] = max(0.0, 1.0 - sum(shares.values()))
        r["valu_type_shares"] = shares
    # (rounds 3-4 also printed `peak_at_mix` / `valu_busy_at_mix`: the 2-cycle peak re-priced with per-class issue costs from a microbenchmark.
    # Round 5 re-measured those costs at the kernel's own clock in >= 20 ms kernels: they depend on how many waves the microbenchmark keeps
    # resident and on the chip's power state (fp32 mul 2.9 -> 3.5 cycles, one wave per SIMD: 7.5), not on the instruction alone, and a busy
    # fraction of 1.04 was the result.  Dropped: `valu_busy` at the architectural 2-cycle rate is the measurement; what binds the BVH kernels
    # is `memory_path`.  profiles/r05_valu_op_rates.txt)
    if pmc.get("hbm_read_bytes") is not None and pmc.get("hbm_write_bytes") is not None:
        r["traffic"] = pmc["hbm_read_bytes"] + pmc["hbm_write_bytes"]
        r["hbm_physical"] = {"bytes_per_launch": r["traffic"], "GBps": r["traffic"] / (launch_ms * 1e-3) / 1e9,
                             "frac_of_8TBps": r["traffic"] / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    else:
        r["traffic"] = None
    return r


# --------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=65, help="timed frames (default 65: a progressive render's steady state = the first frame at once + 4 coalesced launches of 16)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--regions", type=int, default=7, help="further timed regions after the K steps (each >= K steps and >= --region-ms long) -> value_median_of_regions; 0 = off")
    ap.add_argument("--region-ms", type=float, default=50.0, help="minimum length of one of the --regions regions")
    ap.add_argument("--config", type=int, default=2, help="scene id (default 2 = the headline workload; 5 = the 8-GPU 3840x2160 case)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batched", action="store_true", help="skip the informational rt_render_frames(K) pass (profile runs)")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 PMC passes (roofline counters are then replayed from profiles/)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the BVH-workload roofline block (configs 3 and 4)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--frames-per-launch", type=int, default=16, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.pmc_child:
        return pmc_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    # Test hooks (single-GPU boxes): RT_BENCH_ONE_DEVICE=1 puts every rank on GPU 0 and
    # RT_BENCH_BACKEND=gloo replaces RCCL, so the N>1 code path can be exercised without N GPUs.
    dev_index = 0 if os.environ.get("RT_BENCH_ONE_DEVICE") else local_rank
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    pinned_cpus = (pin_to_gpu_numa_node(torch, dev_index)
                   if (world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1") and not os.environ.get("RT_BENCH_ONE_DEVICE") else None)
    device = torch.device("cuda", dev_index)
    comm_device = device if backend == "nccl" else torch.device("cpu")
    # RT_BENCH_FORCE_DIST=1: a ONE-rank job takes the distributed path all the same (process group on the chosen backend, partition
    # 1/1, bound render targets, the gather collective, all_reduce / all_gather_object) — the driver's SCALE run's RCCL code executes
    # on a 1-GPU box (tests/test_zz_dist_gpu.py)
    dist_on = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
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
    if dist_on:
        tiled = pkg.dist.TiledTracer(tracer, rank, world, device, always_collective=True)
    mgr = scene.make_manager(tracer, api, W, H)
    mgr.bvhOnGpu = True         # CreateAllMeshData through rt_build_bvh_gpu_batch: the scene's meshes as one forest, byte-identical trees
    mgr.bvhDevice = dev_index   # on this rank's GPU
    t_load = time.perf_counter()
    mgr.OnEnable(renderSeed=1)  # resize + BVH build + upload + reset: everything resident in HBM
    tracer.synchronize()
    scene_load = {"ms": (time.perf_counter() - t_load) * 1e3,
                  "what": "OnEnable: rt_resize + CreateAllMeshData (GPU BVH builder, one forest) + rt_upload_scene + parameters + reset, "
                          "first call of the process (module load and scratch allocation included); outside the timed region",
                  "bvh_build_ms": sum(float(st.get("timeMs", 0.0)) for st in getattr(mgr, "bvhStats", {}).values())}
    if tiled:
        tiled.bind(W, H)
        tracer.reset_accumulation()
    n_models, n_spheres = len(scene.models), len(scene.spheres)
    diagnostics = []          # failures of the informational passes: reported in the line, never fatal, never skipping a collective
    accumulated = [0]         # frames added into the accumulation buffer since its last reset (every pass below adds K)

    def barrier():
        if dist_on:
            dist.barrier()
        tracer.synchronize()
        torch.cuda.synchronize()

    def replay_from(frame):
        """Point the mirror (and the library) back at `frame`: the next frames rendered are frame, frame+1, ..."""
        mgr.numAccumulatedFrames = frame
        mgr.SetShaderParams()

    # ---- warmup
    for _ in range(args.warmup):
        mgr.RenderFrame()
    accumulated[0] += args.warmup
    skew_ms = float(os.environ.get("RT_BENCH_RANK_SKEW_MS", "0") or 0)   # test hook: odd ranks fall behind by this much before every pass
    def skew():
        if skew_ms > 0 and rank % 2 == 1:
            time.sleep(skew_ms * 1e-3)
    skew()
    barrier()
    first_frame = tracer.frame()
    if first_frame != 1 + args.warmup:
        diagnostics.append(f"rank {rank}: first timed Frame is {first_frame}, expected 1 + warmup = {1 + args.warmup}")
    if dist_on:   # every rank times the SAME frame indices (they seed the RNG, RC:552, and the gathered image adds them up per strip)
        firsts = [None] * world
        dist.all_gather_object(firsts, first_frame)
        if len(set(firsts)) != 1:
            diagnostics.append(f"rank {rank}: ranks disagree on the first timed Frame: {firsts}")

    # ---- timed: exactly K steps
    tracer.reset_counters()
    barrier()
    t0 = time.perf_counter()
    tracer.timer_begin()
    for _ in range(args.steps):
        tracer.render_frame()
    tracer.timer_end()
    tracer.synchronize()          # this rank's K steps are done ...
    torch.cuda.synchronize()
    t1 = time.perf_counter()      # ... at t1; the job's time is the MAX over ranks of t1 - t0 (taken below), so the closing
    barrier()                     # barrier brackets the region without adding the collective's own latency to every rank
    accumulated[0] += args.steps
    timed = tracer.counters()
    elapsed = t1 - t0
    segments = timed["segments"]

    # ---- R further regions of the same progressive render (>= K steps and >= --region-ms each): the median is what a 1 % change shows up in
    # (a single 13 ms region scatters by +-5 % with the device's clock ramp).  Steps per region from an ALL-REDUCED time: identical on every rank.
    regions = None
    if args.regions > 0:
        job_elapsed = elapsed
        if dist_on:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=comm_device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            job_elapsed = tt[0].item()
        per_step = max(job_elapsed / args.steps, 1e-6)
        region_steps = int(min(4096, max(args.steps, math.ceil(args.region_ms * 1e-3 / per_step))))
        r_el, r_seg = [], []
        for _ in range(args.regions):
            skew()
            tracer.reset_counters()
            barrier()
            r0 = time.perf_counter()
            for _ in range(region_steps):
                tracer.render_frame()
            tracer.synchronize()
            torch.cuda.synchronize()
            r_el.append(time.perf_counter() - r0)
            barrier()
            r_seg.append(tracer.counters()["segments"])
            accumulated[0] += region_steps
        regions = {"n": args.regions, "steps_each": region_steps, "elapsed_s": r_el, "segments": r_seg,
                   "first_frame": first_frame + args.steps}

    # ---- the same K frames through the host mirror's RenderFrame(): InitFrame + Dispatch per frame (RCM:84-95)
    mgr.numAccumulatedFrames = first_frame
    tracer.reset_counters()
    skew()
    barrier()
    i0 = time.perf_counter()
    for _ in range(args.steps):
        mgr.RenderFrame()
    barrier()
    init_elapsed = time.perf_counter() - i0
    accumulated[0] += args.steps
    init_segments = tracer.counters()["segments"]
    if init_segments != segments:   # same frame indices, same seed: the exact counters must agree
        diagnostics.append(f"rank {rank}: RenderFrame() pass counted {init_segments} segments, timed pass {segments}")

    # ---- untimed replay of the same K frames with the detailed counters on
    # (identical work: the frame index and seed decide every ray) -> algorithmic bytes
    tracer.reset_counters()
    tracer.enable_stats(True)
    replay_from(first_frame)
    tracer.render_frames(args.steps)
    accumulated[0] += args.steps
    stats = tracer.counters()
    lds_cache = None
    chain_pool = None
    try:   # BVH scenes: how many inner steps (lane-steps) the LDS top-of-tree cache served in the replayed K frames
        ph = tracer.phase_profile()
        if not ph["inner"][1] and ph["loop"][0]:
            # scenes without trees (the FLAT kernel): chains handed from wave to wave through the workgroup's LDS pool, and what that does to the two
            # exclusive phases behind the intersection (wave-level executions and the lanes active in them, STATS instantiation, the timed frames)
            util = lambda k: (ph[k][1] / (64.0 * ph[k][0])) if ph[k][0] else None
            chain_pool = {"chains_deposited": ph["inner_from_lds_cache"][0], "wave_iterations": ph["loop"][0],
                          "deposits_per_wave_iteration": ph["inner_from_lds_cache"][0] / ph["loop"][0],
                          "lane_util_sky": util("sky"), "lane_util_shade": util("shade_hit"), "lane_util_raygen": util("raygen"),
                          "what": "the FLAT kernel's workgroups (16 waves) share two LDS queues of pixel chains waiting for the sky / for the shade phase: a wave "
                                  "deposits the lanes of its minority phase and withdraws chains of its majority phase (rt_kernels.h, pool_exchange); 0 deposits = "
                                  "single-wave workgroups (RT_POOL=0, or a launch with too few items per wave)"}
        if ph["inner"][1]:
            lds_cache = {"inner_lane_steps": ph["inner"][1], "served_from_lds": ph["inner_from_lds_cache"][0],
                         "fraction": ph["inner_from_lds_cache"][0] / ph["inner"][1],
                         "steps_with_48_lanes_on_one_node": ph["inner_on_one_node_48_lanes"][0] / ph["inner"][1],
                         "what": "the BVH kernels' workgroups (up to 12 waves) share an LDS copy of the top of the scene's trees; counted by the STATS instantiation over the timed frames"}
    except Exception as e:  # informational
        diagnostics.append(f"rank {rank}: phase profile: {type(e).__name__}: {e}")
    tracer.enable_stats(False)
    if stats["segments"] != segments:
        diagnostics.append(f"rank {rank}: stats replay counted {stats['segments']} segments, timed pass {segments}")

    # ---- informational: the same K frames through the batched API (rt_render_frames: up to 16
    # frames per launch as (tile, frame) items; identical final buffers)
    batched = None
    if not args.no_batched:
        replay_from(first_frame)
        tracer.reset_counters()
        barrier()
        b0 = time.perf_counter()
        tracer.render_frames(args.steps)
        barrier()
        batched_elapsed = time.perf_counter() - b0
        accumulated[0] += args.steps
        batched = {"what": "rt_render_frames(K): frames fused up to 16 per launch (this rank)",
                   "value": tracer.counters()["segments"] / batched_elapsed / 1e6, "unit": "Mrays/s",
                   "ms_per_frame": batched_elapsed / args.steps * 1e3}

    # ---- roofline pass (rank 0): the dominant kernel of the timed region is the FUSED launch (K back-to-back
    # rt_render_frame calls leave as launches of up to 16 frames).  The same frames are rendered once more by a context
    # restricted to one kernel per launch on one stream, `fpl` frames per launch: HIP events around those launches give
    # the kernel's average launch duration, the figure rocprofv3 --kernel-trace reports for the same mode (profiles/).
    # A second short pass at one frame per launch gives the reference's dispatch granularity for comparison.
    single_ms = launch_ms = launch_segments = None
    fpl = max(1, min(16, args.steps))
    if rank == 0:
        part = (pkg.dist.STRIP_ROWS, rank, world) if tiled else None
        try:
            launch_ms, launch_segments = launch_profile(pkg, api, dev_index, args.config, W, H, max(2, min(6, args.steps // fpl + 1)), fpl, part)
            single_ms, seg1 = launch_profile(pkg, api, dev_index, args.config, W, H, min(args.steps, 8), 1, part)
        except Exception as e:  # informational
            diagnostics.append(f"rank 0: launch_profile: {type(e).__name__}: {e}")

    # ---- readback: the one collective of the multi-GPU path (every rank takes part whatever happened above)
    gather_ms = None
    gather_error = None
    gather_alpha_ok = None
    gathered = None
    if tiled:
        barrier()
        g0 = time.perf_counter()
        try:
            full = tiled.gather_accumulated(H, comm_device=comm_device)
            torch.cuda.synchronize()
            gather_ms = (time.perf_counter() - g0) * 1e3
            if rank == 0:
                assert tuple(full.shape) == (H, W, 4)
                # alpha counts the additions (RCC:22 adds 1 per frame): every pixel of every rank's strips got every frame
                alpha = full[..., 3]
                gather_alpha_ok = bool((alpha == float(accumulated[0])).all().item())
                gathered = {"alpha_expected": accumulated[0], "alpha_min": float(alpha.min().item()), "alpha_max": float(alpha.max().item())}
                if not gather_alpha_ok:   # a rank rendered other frames than rank 0 counted: the headline would describe an incoherent image
                    diagnostics.append(f"rank 0: gathered image incomplete: alpha in [{gathered['alpha_min']}, {gathered['alpha_max']}], expected {accumulated[0]} everywhere")
            del full
        except Exception as e:  # the timed result stands on its own; report the collective's failure instead of losing the line
            gather_ms = None
            gather_error = f"{type(e).__name__}: {e}"
            diagnostics.append(f"rank {rank}: gather: {gather_error}")
        # ---- and the gathered image itself against the oracle: a fresh progressive render of 2 frames on all ranks,
        # gathered, sample strips (one per rank at least) re-rendered by the CPU oracle on rank 0
        try:
            mgr.ResetAccumulatedRender()
            accumulated[0] = 0
            mgr.RenderFrames(2)
            accumulated[0] += 2
            full = tiled.gather_accumulated(H, comm_device=comm_device)
            if rank == 0:
                n_strips = (H + 7) // 8
                strips = sorted(set([r for r in range(min(world, n_strips))] + [n_strips // 2 + r for r in range(min(world, 3))] + [n_strips - 1]))
                strips = [s_ for s_ in strips if 0 <= s_ < n_strips][:12]
                gp = oracle_strips_check(pkg, args.config, W, H, full.cpu().numpy(), 2, strips)
                gp["ranks_covered"] = sorted({s_ % world for s_ in strips})
                gathered = dict(gathered or {}, parity_vs_oracle=gp)
            del full
        except Exception as e:
            diagnostics.append(f"rank {rank}: gathered-image parity: {type(e).__name__}: {e}")

    # ---- who took part: one record per rank, collected with the job's own backend (RCCL when N GPUs are there)
    devices_seen = None
    if dist_on:
        props = torch.cuda.get_device_properties(dev_index)
        me = {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "name": props.name,
              "uuid": str(getattr(props, "uuid", "")), "pci_bus_id": getattr(props, "pci_bus_id", None),
              "pci_device_id": getattr(props, "pci_device_id", None), "pid": os.getpid(), "backend": dist.get_backend(),
              "pinned_cpus": pinned_cpus}
        devices_seen = [None] * world
        dist.all_gather_object(devices_seen, me)

    if dist_on:
        t = torch.tensor([elapsed, float(timed["gpuMs"]), init_elapsed], dtype=torch.float64, device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms_max, init_elapsed = t[0].item(), t[1].item(), t[2].item()
        s = torch.tensor([segments, pkg.abi.algorithmic_bytes(stats, n_models, n_spheres)], dtype=torch.float64, device=comm_device)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        total_segments, total_bytes = s[0].item(), s[1].item()
        if regions:
            re_ = torch.tensor(regions["elapsed_s"], dtype=torch.float64, device=comm_device)
            rs_ = torch.tensor([float(x) for x in regions["segments"]], dtype=torch.float64, device=comm_device)
            dist.all_reduce(re_, op=dist.ReduceOp.MAX)
            dist.all_reduce(rs_, op=dist.ReduceOp.SUM)
            regions["elapsed_s"], regions["segments"] = re_.tolist(), rs_.tolist()
    else:
        kernel_ms_max = timed["gpuMs"]
        total_segments = float(segments)
        total_bytes = float(pkg.abi.algorithmic_bytes(stats, n_models, n_spheres))

    diagnostics_all = diagnostics
    if dist_on:
        lists = [None] * world
        dist.all_gather_object(lists, diagnostics)
        diagnostics_all = [d_ for l_ in lists for d_ in l_]

    if rank == 0:
        my_bytes = pkg.abi.algorithmic_bytes(stats, n_models, n_spheres) / args.steps
        spp, mb = scene.settings["numRaysPerPixel"], scene.settings["maxBounceCount"]
        # ---- VALU roofline of the dominant (only) kernel
        pmc = None
        if not dist_on and not args.no_pmc:
            pmc = collect_pmc(args.config, 3, 1, with_traffic=True, fpl=fpl)
        if pmc is None and not dist_on:
            pmc = replayed_pmc(args.config, fpl)
        roof = None
        if pmc is not None:
            roof = valu_roofline(pmc, launch_ms, launch_segments, "flat" if args.config <= 2 else "bvh")
            roof["frames_per_launch"] = fpl
            roof["memory_path"] = memory_path_roofline(pmc, launch_ms)
            roof["kernel"] = (f"rt_trace_kernel<false, *>, launches of {fpl} frames ((tile, frame) work items; one kernel per launch on one "
                              "stream in the roofline pass) — the form the timed K back-to-back rt_render_frame calls are launched in")
            roof["peak_derivation"] = "256 CU x 4 SIMD-32 x 2.4 GHz / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md)"
            roof["secondary_hbm_algorithmic"] = {
                "what": "SURVEY.md 8(d) algorithmic bytes (the reference loop's loads for the counted work) / launch time; the scene is "
                        "SGPR/L2 resident, so this is NOT a roof and may exceed 8 TB/s — kept for continuity with round 1",
                "bytes_per_launch": my_bytes * fpl, "GBps": my_bytes * fpl / (launch_ms * 1e-3) / 1e9}
            roof["secondary_hbm_algorithmic"]["frac_of_peak"] = roof["secondary_hbm_algorithmic"]["GBps"] / HBM_PEAK_GBS
            # more than the memory system can move: the figure describes the REFERENCE's loads, most of which are SGPR / cache hits here
            roof["secondary_hbm_algorithmic"]["void"] = roof["secondary_hbm_algorithmic"]["frac_of_peak"] > 1.0
            roof["fetch_size_factor"] = {"factor": 2.0, "calibration": "profiles/r04_fetch_size_calibration.txt (streaming and divergent 64-B / 48-B per-lane "
                                                                       "record fetches: FETCH_SIZE adds 64 B per 128-B line fill)"}
        parity = parity_check(pkg, api, dev_index, args.config, W, H, (3, H // 16, H // 8 - 2)) if not dist_on else None
        out = {
            "metric": f"Mrays/s at {W}x{H}, {spp} spp, {mb} bounces; per-channel L2 vs reference",
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
                "workload": f"{scene.name}: {W}x{H}, {spp} spp/frame, {mb} bounces, {n_spheres} spheres + {n_models} models "
                            f"({scene.unique_triangles()} triangles); "
                            + (f"BASELINE.json configs[{args.config - 1}]" if args.config <= 5 else "the reference's own scene file, not a BASELINE config"),
                "parallelism": "single GPU" if world == 1 else f"{W}x{H} image row-tiled, cyclic 8-row strips over {world} GPUs "
                               f"({args.scaling} scaling), RCCL gather at readback",
                "renderSeed": 1, "first_timed_frame": first_frame,
            },
            "segments_per_step": total_segments / args.steps,
            "mpaths_per_s": W * H * spp * args.steps / elapsed / 1e6,
            "resolution": [W, H],
            "kernel_ms_per_step": kernel_ms_max / args.steps,
            "value_median_of_regions": None, "regions": None,
            "value_with_initframe": total_segments / init_elapsed / 1e6,
            "ms_per_step_with_initframe": init_elapsed / args.steps * 1e3,
            "gather_ms": gather_ms, "gather_error": gather_error, "gathered_image_complete": gather_alpha_ok,
            "gathered_image": gathered,
            "ranks_seen": (dist.get_world_size() if dist_on else 1), "devices_seen": devices_seen,
            "distinct_devices": (len({(d_["uuid"], d_["pci_bus_id"]) for d_ in devices_seen}) if devices_seen else 1),
            "diagnostics": diagnostics_all,
            "launches": ("K x rt_render_frame, back to back: the library starts an idle GPU at once (frame 1: 2 kernels on 2 streams, disjoint "
                         "tile halves) and holds frames requested while earlier ones still execute back, up to 16, to launch them fused as "
                         "(tile, frame) work items whose per-frame colours are added in frame order afterwards (same bits); "
                         "RT_COALESCE=0 = one launch per call")
                        if os.environ.get("RT_COALESCE") != "0" else
                        ("2 kernels per frame on 2 streams (disjoint tile halves, overlapping in time)"
                         if os.environ.get("RT_TWO_STREAMS") != "0" else "1 kernel per frame"),
            "value_one_kernel_per_frame": (segments / args.steps / (single_ms * 1e-3) / 1e6) if single_ms else None,
            "value_fused_launches_one_stream": (launch_segments / (launch_ms * 1e-3) / 1e6) if launch_ms else None,
            "batched_api": batched,
            "lds_top_of_tree_cache": lds_cache,
            "chain_pool": chain_pool,
            "scene_load": scene_load,
            "parity": parity if parity is not None else "checked at N=1 (pytest -m gpu and the N=1 bench line)",
            "roofline": roof,
        }
        if regions:
            vals = [sg / el / 1e6 for sg, el in zip(regions["segments"], regions["elapsed_s"])]
            out["value_median_of_regions"] = sorted(vals)[len(vals) // 2]
            out["regions"] = {"n": regions["n"], "steps_each": regions["steps_each"], "first_frame": regions["first_frame"],
                              "values": vals, "min": min(vals), "max": max(vals),
                              "ms_per_step_median": sorted(el / regions["steps_each"] * 1e3 for el in regions["elapsed_s"])[len(vals) // 2],
                              "what": f"{regions['n']} further back-to-back regions of the same progressive render after the K timed steps, each {regions['steps_each']} steps "
                                      f"(>= K and >= {args.region_ms:g} ms), each bracketed by barrier + synchronise like the timed one; max time over ranks, segments summed; "
                                      "`value` stays the driver's W + K contract"}
        # ---- the same roofline for BVH workloads (north_star's roofline clause is about BVH traversal, RC:234-287)
        if not dist_on and args.config == 2 and not args.no_secondary:
            sec = {}
            for cfg in (3, 4):
                sc2 = pkg.scenes.get(cfg)
                ms2, seg2 = launch_profile(pkg, api, dev_index, cfg, sc2.width, sc2.height, 2, 16)
                p2 = (None if args.no_pmc else collect_pmc(cfg, 2, 1, with_traffic=False, fpl=16)) or replayed_pmc(cfg)
                if p2 is None:
                    continue
                r2 = valu_roofline(p2, ms2, seg2)
                r2["frames_per_launch"] = 16
                r2["memory_path"] = memory_path_roofline(p2, ms2)   # the binding roof of the BVH workloads
                r2["mrays_per_s_fused_launches"] = seg2 / (ms2 * 1e-3) / 1e6
                r2["workload"] = f"{sc2.name}: {sc2.width}x{sc2.height}, {sc2.unique_triangles()} triangles, BASELINE.json configs[{cfg - 1}]"
                if cfg == 3 and not args.no_cpu_baseline:
                    # the reference's own loop (its shader text compiled as C++) on one host core, a bounded sample at 1/16 of the area
                    cb = cpu_baseline_reference(pkg, cfg, sc2.width // 4, sc2.height // 4)
                    if cb:
                        r2["cpu_baseline"] = cb
                        r2["gpu_over_cpu"] = r2["mrays_per_s_fused_launches"] / cb["value"]
                sec[f"config{cfg}"] = r2
            out["secondary"] = sec
        if not dist_on and not args.no_cpu_baseline:
            # the reference's own loop (its text; + the declared sphere hook S1 for a sphere scene) on one pinned core, a bounded sample: the
            # same scene at 1/4 x 1/4 of the resolution for the headline (rays/s does not depend on the resolution to first order; one
            # 1920x1080 frame would be ~45 s of this library); the builder's port stays beside it
            port = cpu_baseline(pkg, args.config, W, H)
            try:
                refb = cpu_baseline_reference(pkg, args.config, max(8, W // 4), max(8, H // 4), min_s=10.0, max_frames=4)
            except Exception as e:
                refb = None
                port["reference_leg_error"] = f"{type(e).__name__}: {e}"
            out["cpu_baseline"] = refb if refb else port
            if refb:
                out["cpu_baseline_port"] = port
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            # informational: the same port on the host's cores (row bands of the image per thread)
            nthr = min(os.cpu_count() or 1, 64)
            if nthr > 1:
                out["cpu_baseline_threads"] = cpu_baseline(pkg, args.config, W, H, min_s=4.0, max_frames=64, threads=nthr)
        if diagnostics_all:
            # a segment-count disagreement between the passes is what the lost-counter race of round 2 looked like: the line is
            # still printed (with its diagnostics), but it carries no headline value and the run fails
            out["value_unchecked"] = out["value"]
            out["value"] = None
        print(json.dumps(out), flush=True)

    tracer.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if diagnostics_all:
        raise SystemExit("bench.py: diagnostics present: " + "; ".join(diagnostics_all))


def run():
    """main() with every failure tagged by rank on stderr (a launcher shows only the tail of one rank's output)."""
    rank = os.environ.get("RANK", "0")
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:  # noqa: BLE001 - report, then fail the process
        import traceback
        sys.stderr.write(f"[rank {rank}] bench.py failed: {type(e).__name__}: {e}\n")
        for line in traceback.format_exc().splitlines():
            sys.stderr.write(f"[rank {rank}] {line}\n")
        sys.stderr.flush()
        # a rank that leaves while its peers wait in a collective must not hang on interpreter teardown
        os._exit(1)


if __name__ == "__main__":
    run()
