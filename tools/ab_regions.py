"""A/B of run-time variants on one GPU, the way VERDICT r5 asks for it: per (config, variant) the MEDIAN over >= 7 back-to-back regions
of >= 50 ms each (fused launches through rt_render_frames, HIP events around each region), variants interleaved round-robin so that
clock / thermal drift hits all of them alike.  A variant is a set of environment variables read when the context is made / the scene is
uploaded (RT_WAVES_PER_GROUP, RT_HOT_KB, RT_LAYOUT, RT_SUSPEND, RT_HIP_LIB is NOT one: one library per process).

usage: python tools/ab_regions.py CONFIGS "name:K=V,K=V" "name2:..." [--regions=7] [--region-ms=50] [--golden]
   e.g. python tools/ab_regions.py 3,4,6 "wave1:RT_WAVES_PER_GROUP=1" "g12:RT_WAVES_PER_GROUP=12" """
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g


class Env:
    def __init__(self, env): self.env = env
    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.env}; os.environ.update(self.env)
    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = dict((a.split("=", 1) + ["1"])[:2] for a in sys.argv[1:] if a.startswith("--"))
    regions = int(opts.get("--regions", 7)); region_ms = float(opts.get("--region-ms", 50.0))
    cfgs = [int(c) for c in args[0].split(",")]
    variants = []
    for v in args[1:]:
        name, _, kv = v.partition(":")
        variants.append((name, dict(x.split("=", 1) for x in kv.split(",") if x)))
    # per-context adaptive state would differ between variants (the launch tuner's 3/8 vs 4/8, the frames-per-launch budget): pinned unless a variant sets it
    os.environ.setdefault("RT_SUSPEND", "3")
    os.environ.setdefault("RT_FUSE_CAP", "16")
    pkg = g.load_package(); api = pkg.load_library()
    if "--golden" in opts:
        import importlib.util
        spec = importlib.util.spec_from_file_location("mg", os.path.join(g.ROOT, "tests", "golden", "make_golden.py"))
        mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
        for name, env in variants:
            with Env(env):
                bad = []
                for case in sorted(mg.CASES):
                    z = np.load(os.path.join(g.ROOT, "tests", "golden", case + ".npz"))
                    for stats in (False, True):
                        tr = api.create_tracer(0); tr.enable_stats(stats)
                        acc, cnt = mg.render_case(pkg, api, tr, case); tr.close()
                        ok = np.array_equal(acc.view(np.uint32), z["accumulated"].view(np.uint32)) and (not stats or cnt.tolist() == z["counters"].tolist())
                        if not ok: bad.append((case, stats))
            print(f"golden[{name}]:", "ALL BIT-EXACT (both instantiations)" if not bad else "MISMATCH " + str(bad), flush=True)
    for cfg in cfgs:
        ctxs = []
        for name, env in variants:
            with Env(env):
                tr = api.create_tracer(0)
                sc = pkg.scenes.get(cfg); mgr = sc.make_manager(tr, api); mgr.OnEnable(renderSeed=1)
                tr.render_frames(16); tr.synchronize()
            tr.reset_counters(); tr.timer_begin(); tr.render_frames(16); tr.timer_end()
            c = tr.counters(); ms = c["gpuMs"] / 16
            n = max(16, int(math.ceil(region_ms / ms)))
            ctxs.append([name, tr, n, [], []])
        for r in range(regions):
            for name, tr, n, mss, rates in ctxs:
                tr.reset_counters(); tr.timer_begin(); tr.render_frames(n); tr.timer_end()
                c = tr.counters()
                mss.append(c["gpuMs"] / n); rates.append(c["segments"] / c["gpuMs"] / 1e3)
        base = None
        for name, tr, n, mss, rates in ctxs:
            med = sorted(mss)[len(mss) // 2]
            if base is None: base = med
            print(f"config {cfg} {name:>14s}: median {med:8.4f} ms/frame ({med / base - 1:+.1%} vs first)  min {min(mss):8.4f} max {max(mss):8.4f}  "
                  f"{sorted(rates)[len(rates) // 2]:9.1f} Mrays/s  ({regions} regions x {n} frames)", flush=True)
            tr.close()


if __name__ == "__main__":
    main()
