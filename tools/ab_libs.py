"""A/B of library BUILDS on one GPU in ONE process: per (config, build) the MEDIAN over >= 7 back-to-back regions of >= 50 ms each (fused
launches through rt_render_frames, device events around each region), the builds interleaved round-robin region by region so that clock /
thermal drift hits all of them alike (tools/ab_regions.py does the same for run-time variants of one build).  Every build first renders
the golden fixtures through both kernel instantiations: a build that changed a bit says so and is not timed.

usage: python tools/ab_libs.py CONFIGS name=lib-suffix[:K=V,K=V] ... [--regions=7] [--region-ms=50] [--no-golden]
   lib-suffix "" = the product build (ray-tracing_amd/lib/libraytrace_hip.so), "x" = libraytrace_hip_x.so; K=V = environment of that build's contexts
   e.g. python tools/ab_libs.py 3,4,6 base= pair=expTRI_PAIR top=expSTACK_TOP"""
import math, os, sys
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
    os.environ.setdefault("RT_SUSPEND", "3")      # per-context adaptive state pinned, as in ab_regions.py
    os.environ.setdefault("RT_FUSE_CAP", "16")
    pkg = g.load_package()
    libdir = os.path.join(g.ROOT, "ray-tracing_amd", "lib")
    builds = []
    for v in args[1:]:
        name, _, rest = v.partition("=")
        suf, _, kv = rest.partition(":")
        env = dict(x.split("=", 1) for x in kv.split(",") if x)
        builds.append((name, pkg.hip.HipApi(os.path.join(libdir, f"libraytrace_hip{'_' + suf if suf else ''}.so")), env))
    if "--no-golden" not in opts:
        import importlib.util
        spec = importlib.util.spec_from_file_location("mg", os.path.join(g.ROOT, "tests", "golden", "make_golden.py"))
        mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
        good = []
        for name, api, env in builds:
            bad = []
            with Env(env):
                for case in sorted(mg.CASES):
                    z = np.load(os.path.join(g.ROOT, "tests", "golden", case + ".npz"))
                    for stats in (False, True):
                        tr = api.create_tracer(0); tr.enable_stats(stats)
                        acc, cnt = mg.render_case(pkg, api, tr, case); tr.close()
                        if not (np.array_equal(acc.view(np.uint32), z["accumulated"].view(np.uint32)) and (not stats or cnt.tolist() == z["counters"].tolist())):
                            bad.append((case, stats))
            print(f"golden[{name}]:", "ALL BIT-EXACT (both instantiations)" if not bad else "MISMATCH " + str(bad) + " — not timed", flush=True)
            if not bad: good.append((name, api, env))
        builds = good
    for cfg in cfgs:
        ctxs = []
        for name, api, env in builds:
            with Env(env):
                tr = api.create_tracer(0)
                mgr = pkg.scenes.get(cfg).make_manager(tr, api); mgr.OnEnable(renderSeed=1)
                tr.render_frames(16); tr.synchronize()
            tr.reset_counters(); tr.timer_begin(); tr.render_frames(16); tr.timer_end()
            ms = tr.counters()["gpuMs"] / 16
            ctxs.append([name, tr, max(16, int(math.ceil(region_ms / ms))), [], []])
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
