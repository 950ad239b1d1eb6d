"""Summarise the rocprofv3 (rocpd sqlite) outputs of tools/prof.sh: kernel stats,
dispatch resources and per-dispatch PMC averages -> text on stdout + summary.json."""
import glob, json, os, sqlite3, sys

out = sys.argv[1]
res = {}
for d in sorted(glob.glob(os.path.join(out, "trace", "**", "*.db"), recursive=True)):
    cur = sqlite3.connect(d).cursor()
    print("== kernel stats (rocprofv3 --kernel-trace --stats):")
    res["kernel_stats"] = []
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("   %-70s calls=%4d total_us=%10.1f avg_us=%9.2f pct=%5.1f" % (name[:70], calls, total, avg, pct))
        res["kernel_stats"].append({"name": name, "calls": calls, "total_us": total, "avg_us": avg, "pct": pct})
    rows = list(cur.execute("select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, grid_x, workgroup_x, "
                            "count(*), avg(duration), min(duration), max(duration) from kernels where name like '%rt_trace%' group by name"))
    for r in rows:
        print("== dispatch: %s vgpr=%s agpr=%s sgpr=%s lds=%s scratch=%s grid=%s wg=%s n=%d avg_ns=%.0f min_ns=%d max_ns=%d" % r)
        res.setdefault("dispatch", []).append(dict(zip(
            ["name", "vgpr", "agpr", "sgpr", "lds", "scratch", "grid", "wg", "n", "avg_ns", "min_ns", "max_ns"], r)))
res["pmc"] = {}
print("== PMC per dispatch of rt_trace_kernel<false,*> (avg over n dispatches; separate passes):")
for d in sorted(glob.glob(os.path.join(out, "pmc_*", "**", "*.db"), recursive=True)):
    cur = sqlite3.connect(d).cursor()
    q = ("select counter_name, count(*), avg(v), min(v), max(v) from (select counter_name, dispatch_id, sum(value) as v "
         "from counters_collection where kernel_name like '%rt_trace_kernel<false%' group by counter_name, dispatch_id) group by counter_name")
    for name, n, avg, mn, mx in cur.execute(q):
        print("   %-26s n=%3d avg=%.6g min=%.6g max=%.6g" % (name, n, avg, mn, mx))
        res["pmc"][name] = {"n": n, "avg": avg, "min": mn, "max": mx}
p = res["pmc"]
if "FETCH_SIZE" in p:
    # MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the
    # bytes of wide coalesced reads -> doubled (upper correction); WRITE_SIZE uncalibrated, taken as is.
    rd = p["FETCH_SIZE"]["avg"] * 1024 * 2
    wr = p.get("WRITE_SIZE", {"avg": 0})["avg"] * 1024
    res["hbm_bytes_per_launch"] = rd + wr
    print("== HBM traffic per launch: read %.3f MB (FETCH_SIZE x1024 x2 gfx950 correction) + write %.3f MB = %.3f MB"
          % (rd / 1e6, wr / 1e6, (rd + wr) / 1e6))
if "TCC_HIT_sum" in p:
    print("== L2 hit rate: %.4f" % (p["TCC_HIT_sum"]["avg"] / (p["TCC_HIT_sum"]["avg"] + p["TCC_MISS_sum"]["avg"])))
if "SQ_THREAD_CYCLES_VALU" in p and "SQ_ACTIVE_INST_VALU" in p:
    # lanes active per VALU instruction: SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64) (both in quad-cycles x lanes)
    print("== VALU lane utilisation ~ %.3f" % (p["SQ_THREAD_CYCLES_VALU"]["avg"] / (p["SQ_ACTIVE_INST_VALU"]["avg"] * 64 / 4 * 4)))
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
