"""The golden fixtures through the HIP path named by RT_HIP_LIB (default: the product build): exit 0 = all bit-exact (images and exact counters).
usage: [RT_HIP_LIB=...] python tools/golden_check.py      (on the GPU box; a gate in front of A/B runs of kernel variants)"""
import importlib.util, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g

pkg = g.load_package(); api = pkg.load_library()
spec = importlib.util.spec_from_file_location("mg", os.path.join(g.ROOT, "tests", "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
bad = []
for case in sorted(mg.CASES):
    z = np.load(os.path.join(g.ROOT, "tests", "golden", case + ".npz"))
    for stats in (True, False):
        tr = api.create_tracer(0); tr.enable_stats(stats)
        acc, cnt = mg.render_case(pkg, api, tr, case); tr.close()
        same = np.array_equal(acc.view(np.uint32), z["accumulated"].view(np.uint32)) and (not stats or cnt.tolist() == z["counters"].tolist())
        if not same:
            bad.append((case, "stats" if stats else "shipped"))
print("golden:", "ALL BIT-EXACT (both kernel instantiations)" if not bad else "MISMATCH " + str(bad))
sys.exit(1 if bad else 0)
