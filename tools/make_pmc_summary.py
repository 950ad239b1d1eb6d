"""Build profiles/pmc_summary.json (read by bench.py for roofline.traffic) from tools/prof.sh outputs:
usage: [FPL=16] python tools/make_pmc_summary.py <key>=<gpurun_out/prof_dir> ...   (FPL = frames per launch of the profiled runs,
tools/prof.sh's third argument; bench.py only replays an entry whose frames_per_launch matches its roofline pass)"""
import json, os, sys
out = {}
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_summary.json")
if os.path.exists(path):
    out = json.load(open(path))
for arg in sys.argv[1:]:
    key, d = arg.split("=")
    s = json.load(open(os.path.join(d, "summary.json")))
    p = s["pmc"]
    rd = p["FETCH_SIZE"]["avg"] * 1024 * 2      # KiB -> B, x2: gfx950 FETCH_SIZE under-count (MI355X_MICROARCH.md §HBM)
    wr = p["WRITE_SIZE"]["avg"] * 1024
    out[key] = {
        "frames_per_launch": int(os.environ.get("FPL", "16")),
        "hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr,
        "l2_hit_rate": p["TCC_HIT_sum"]["avg"] / (p["TCC_HIT_sum"]["avg"] + p["TCC_MISS_sum"]["avg"]),
        "valu_insts": p["SQ_INSTS_VALU"]["avg"], "valu_lane_utilisation": p["SQ_THREAD_CYCLES_VALU"]["avg"] / (p["SQ_ACTIVE_INST_VALU"]["avg"] * 64),
        "kernel_avg_us_rocprof": [k["avg_us"] for k in s["kernel_stats"] if "rt_trace_kernel<false" in k["name"]][0],
        "source": d,
    }
    if "SQ_INSTS_VALU_TRANS_F32" in p and "SQ_INSTS_VALU_CVT" in p:  # tools/prof.sh's instruction-type passes (bench.py peak_at_mix)
        out[key]["valu_types"] = {"arith": p["SQ_INSTS_VALU_ADD_F32"]["avg"] + p["SQ_INSTS_VALU_MUL_F32"]["avg"] + p["SQ_INSTS_VALU_FMA_F32"]["avg"],
                                  "trans": p["SQ_INSTS_VALU_TRANS_F32"]["avg"], "int32": p["SQ_INSTS_VALU_INT32"]["avg"],
                                  "int64": p["SQ_INSTS_VALU_INT64"]["avg"], "cvt": p["SQ_INSTS_VALU_CVT"]["avg"]}
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
