"""bench.py's output contract, checked on the committed bench lines (profiles/r02_bench_n1*.json and r03_bench_n1*.json are
verbatim stdout lines of runs on an MI355X; r03_bench_n{2,8}_*.json are the multi-rank lines of a 1-GPU box) and on the script's own defaults — no GPU needed.  The GPU box re-creates such a line at
round end; what is checked here is that the fields the driver and the judge read exist, are typed, and are
self-consistent."""
import ast
import glob
import json
import math
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r02_bench_n1*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r03_bench_n1*.json"))
               + glob.glob(os.path.join(ROOT, "profiles", "r04_bench_n1*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r05_bench_n1*.json")))
R04 = sorted(glob.glob(os.path.join(ROOT, "profiles", "r04_bench_n1*.json")))
R05 = sorted(glob.glob(os.path.join(ROOT, "profiles", "r05_bench_n1*.json")))
MULTI = sorted(glob.glob(os.path.join(ROOT, "profiles", "r03_bench_n[28]_*.json")))


def load(path):
    with open(path) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_there_are_committed_bench_lines():
    assert os.path.join(ROOT, "profiles", "r02_bench_n1.json") in LINES
    assert os.path.join(ROOT, "profiles", "r03_bench_n1.json") in LINES
    assert len(LINES) >= 7 and len(MULTI) >= 2


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_required_fields(path):
    d = load(path)
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        if key == "cpu_baseline" and "_config" in os.path.basename(path) and any(r in os.path.basename(path) for r in ("r03", "r04", "r05")):
            continue   # the round-3 lines of the other configs were taken with --no-cpu-baseline (the headline line has it)
        assert key in d, key
        assert isinstance(d[key], typ), (key, type(d[key]))
    assert "vs_baseline" in d and d["vs_baseline"] is None      # BASELINE.md holds no published number for this metric
    assert d["unit"] == "Mrays/s" and d["metric"].startswith("Mrays/s")
    assert d["higher_is_better"] is True and d["dtype"] == "f32" and "synthetic" in d["data"]
    assert d["scaling"] in ("weak", "strong")
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["n_gpus"] == 1


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_value_is_whole_job_throughput(path):
    d = load(path)
    segments = d["segments_per_step"]
    assert math.isclose(d["value"], segments / (d["ms_per_step"] * 1e-3) / 1e6, rel_tol=1e-6)


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_roofline_object(path):
    r = load(path)["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "valu" and r["unit"] == "Gwave-inst/s"
    assert math.isclose(r["peak"], 256 * 4 * 2.4e9 / 2 / 1e9)
    # achieved = VALU instructions per launch / launch time; frac = achieved / peak x lane utilisation: a real roof
    assert math.isclose(r["achieved"], r["valu_insts_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9, rel_tol=1e-6)
    assert math.isclose(r["frac"], r["achieved"] / r["peak"] * r["lane_util"], rel_tol=1e-6)
    assert 0.0 < r["frac"] < 1.0 and 0.0 < r["lane_util"] <= 1.0 and 0.0 < r["valu_busy"] <= 1.0
    assert "rocprofv3" in r["counters"]                          # collected in the run, not replayed
    assert r["traffic"] is None or r["traffic"] > 0


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_cpu_baseline_and_parity_objects(path):
    d = load(path)
    if "cpu_baseline" not in d:
        assert "_config" in os.path.basename(path) and any(r in os.path.basename(path) for r in ("r03", "r04", "r05"))
        assert d["parity"]["bit_identical"] is True
        return
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "Mrays/s" and c["value"] > 0
    assert "pinned" in c["sample"]
    p = d["parity"]
    assert p["bit_identical"] is True and p["max_rel_err"] == 0.0 and "oracle" in p["checked_in_this_run"]


def test_round4_lines_exist():
    assert os.path.join(ROOT, "profiles", "r04_bench_n1.json") in R04 and os.path.join(ROOT, "profiles", "r04_bench_n1_k20.json") in R04


@pytest.mark.parametrize("path", R04, ids=[os.path.basename(p) for p in R04])
def test_round4_roofline_is_self_consistent(path):
    """"saturated" is a number in the line (peak_at_mix from hardware-counted instruction types x the kernel's
    static class make-up), the algorithmic-bytes figure that exceeds the memory system is marked void, FETCH_SIZE's factor is
    the calibrated one, and a run with diagnostics carries no value."""
    d = load(path)
    assert d["diagnostics"] == [] and d["value"] is not None
    r = d["roofline"]
    for key in ("peak_at_mix", "cycles_per_valu_inst_at_mix", "valu_busy_at_mix", "frac_at_mix", "valu_type_shares", "peak_at_mix_derivation"):
        assert key in r, key
    assert 2.0 <= r["cycles_per_valu_inst_at_mix"] <= 8.2
    assert math.isclose(r["peak_at_mix"], 1024 * 2.4e9 / r["cycles_per_valu_inst_at_mix"] / 1e9, rel_tol=1e-6)
    assert r["peak_at_mix"] < r["peak"]                                 # the 2-cycle rate holds for fp32 add / mul / fma on VGPRs only
    assert math.isclose(r["valu_busy_at_mix"], r["achieved"] / r["peak_at_mix"], rel_tol=1e-6)
    assert math.isclose(r["frac_at_mix"], r["valu_busy_at_mix"] * r["lane_util"], rel_tol=1e-6)
    assert 0.5 < r["valu_busy_at_mix"] < 1.15                           # a model: a few per cent above 1 is its error bar, not a measurement
    assert math.isclose(sum(r["valu_type_shares"].values()), 1.0, rel_tol=1e-6)
    sec = r["secondary_hbm_algorithmic"]
    assert sec["void"] is (sec["frac_of_peak"] > 1.0) and math.isclose(sec["frac_of_peak"], sec["GBps"] / 8000.0, rel_tol=1e-9)
    assert r["fetch_size_factor"]["factor"] == 2.0 and "r04_fetch_size_calibration" in r["fetch_size_factor"]["calibration"]
    assert os.path.exists(os.path.join(ROOT, "profiles", "r04_fetch_size_calibration.txt")) and os.path.exists(os.path.join(ROOT, "profiles", "isa_mix.json"))
    if "scene_load" in d:      # lines of the final build: the scene set-up (GPU BVH forest + upload) is reported beside the timed region
        assert d["scene_load"]["ms"] > 0 and d["scene_load"]["bvh_build_ms"] >= 0 and "outside the timed region" in d["scene_load"]["what"]


def test_round5_lines_exist():
    names = {os.path.basename(p) for p in R05}
    assert {"r05_bench_n1.json", "r05_bench_n1_k20.json", "r05_bench_n1_config3.json", "r05_bench_n1_config4.json", "r05_bench_n1_config5.json",
            "r05_bench_n1_config6.json"} <= names


@pytest.mark.parametrize("path", R05, ids=[os.path.basename(p) for p in R05])
def test_round5_roofs_and_reference_text_parity(path):
    """(weak 3 / item 6) a busy fraction cannot exceed 1 — the `*_at_mix` construct is gone from the line; (item 1) the roof that binds
    the BVH kernels is in the line: L1 accesses per second against one per clock per CU, measured in the run; (item 4) the headline image is compared in the run with
    the reference's text + the declared sphere hook, the BVH configs with the reference's text."""
    d = load(path)
    assert d["diagnostics"] == [] and d["value"] is not None
    r = d["roofline"]
    assert not [k for k in r if k.endswith("_at_mix")]
    assert 0.0 < r["valu_busy"] <= 1.0 and 0.0 < r["lane_util"] <= 1.0 and math.isclose(r["frac"], r["valu_busy"] * r["lane_util"], rel_tol=1e-6)
    mp = r["memory_path"]
    assert mp["bound"] == "l1_access" and mp["unit"] == "G L1 accesses/s" and math.isclose(mp["peak"], 256 * 2.4, rel_tol=1e-9)
    assert math.isclose(mp["achieved"], mp["l1_accesses_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9, rel_tol=1e-6)
    assert math.isclose(mp["frac"], mp["achieved"] / mp["peak"], rel_tol=1e-9)
    assert 0.0 < mp["frac"] <= 1.0 and 0.0 < mp["frac_at_measured_clock"] <= 1.0 and 0.0 < mp["ta_busy"] <= 1.0
    headline = "_config" not in os.path.basename(path)
    if headline:
        assert mp["frac"] < 0.4 and r["valu_busy"] > 0.6          # the sphere scene: VALU issue binds, the memory path idles
    else:
        assert mp["frac"] > 0.75 and mp["frac"] > r["frac"]        # the BVH scenes: at the chip's gather rate
    ref = d["parity"]["vs_reference_text"]
    assert ref["bit_identical"] is True and ref["max_rel_err"] == 0.0 and ref["same_strips"] is True
    assert ("libref_spheres.so" in ref["library"]) is headline and ("libref.so" in ref["library"]) is (not headline)
    if headline and "secondary" in d:
        for cfg in ("config3", "config4"):
            assert d["secondary"][cfg]["memory_path"]["frac"] > 0.75
        assert d["secondary"]["config3"]["cpu_baseline"]["kind"] == "reference" and d["cpu_baseline"]["kind"] == "port"


@pytest.mark.parametrize("path", MULTI, ids=[os.path.basename(p) for p in MULTI])
def test_multi_rank_lines_carry_their_own_evidence(path):
    """N > 1: who took part, that the gathered image holds every frame of every rank, and that it equals the oracle's."""
    d = load(path)
    n = d["n_gpus"]
    assert n in (2, 8) and d["ranks_seen"] == n and len(d["devices_seen"]) == n
    assert sorted(r["rank"] for r in d["devices_seen"]) == list(range(n))
    assert all(r["backend"] in ("gloo", "nccl") and r["name"] for r in d["devices_seen"])
    assert d["gather_ms"] > 0 and d["gather_error"] is None
    assert d["gathered_image_complete"] is True
    g = d["gathered_image"]
    assert g["alpha_min"] == g["alpha_max"] == g["alpha_expected"]
    assert g["parity_vs_oracle"]["bit_identical"] is True and g["parity_vs_oracle"]["ranks_covered"] == list(range(n))
    assert d["diagnostics"] == []
    assert math.isclose(d["value"], d["segments_per_step"] / (d["ms_per_step"] * 1e-3) / 1e6, rel_tol=1e-6)


def test_bench_defaults_and_flags():
    """Default invocation = 1 GPU, the headline config, a K/W that finishes in minutes; the flags the driver passes exist."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    args = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument" and node.args:
            name = node.args[0].value if isinstance(node.args[0], ast.Constant) else None
            default = next((kw.value.value for kw in node.keywords if kw.arg == "default" and isinstance(kw.value, ast.Constant)), None)
            args[name] = default
    assert args["--gpus"] == 1 and args["--config"] == 2
    assert 1 <= args["--steps"] <= 200 and 0 <= args["--warmup"] <= 10
    assert args["--scaling"] == "strong"


def test_round6_no_hidden_spinup_and_regions_are_collective():
    """the wall-clock-bounded spin-up is gone (ranks rendered different frame counts); `value` is the driver's W + K
    contract with first timed Frame = W + 1, asserted across ranks; the extra regions are sized from an all-reduced time; an incomplete
    gathered image is a diagnostic (the run fails).  The round-5 A/B lines stay as history."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "--spinup-ms" not in src and "spinup[" not in src
    assert '"--regions"' in src and "value_median_of_regions" in src
    assert "first_frame != 1 + args.warmup" in src and "dist.all_gather_object(firsts, first_frame)" in src
    assert "gathered image incomplete" in src
    # the steps of a region come from the job's (all-reduced) elapsed time, never from a rank's own clock
    body = src[src.index("regions = None"):src.index("# ---- the same K frames through the host mirror")]
    assert "dist.all_reduce(tt, op=dist.ReduceOp.MAX)" in body and "per_step = max(job_elapsed / args.steps" in body
    assert "time.perf_counter() - t_spin" not in src
    on = load(os.path.join(ROOT, "profiles", "r05_bench_n1_k20.json"))
    assert on["spinup"]["frames"] >= 4      # round 5's line, kept: what the like-for-like series must NOT be compared with
    r06 = sorted(glob.glob(os.path.join(ROOT, "profiles", "r06_bench_n1*.json")))
    for path in r06:
        d = load(path)
        assert "spinup" not in d and d["config"]["first_timed_frame"] == 1 + d["warmup"]
        g = d["regions"]
        assert g["n"] >= 7 and g["steps_each"] >= d["steps"] and len(g["values"]) == g["n"]
        assert g["min"] <= d["value_median_of_regions"] <= g["max"]
        assert g["ms_per_step_median"] * g["steps_each"] >= 45.0          # regions of >= 50 ms (timer slack)
        if "cpu_baseline" in d:
            assert d["cpu_baseline"]["kind"] in ("reference", "reference+S1") and d["cpu_baseline_port"]["kind"] == "port"
