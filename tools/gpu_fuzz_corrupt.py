"""The GPU form of tools/ref_fuzz_corrupt.py, long: batches of 40 corrupted-but-ACCEPTED scenes (tests/test_gpu_fuzz.py's generator and check: node
fields overwritten so that children point anywhere inside the buffer, leaf ranges moved or resized, nodes shared between subtrees) through HIP — the
shipped and the STATS instantiation — against the oracle and, where it travelled, the reference's own text: both render targets and the exact counters,
bit for bit.  The suite runs batches 0..5 (240 scenes); this runs batches [first, first + n).   usage: python tools/gpu_fuzz_corrupt.py [n=77] [first=6]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
import test_gpu_fuzz as F

n = int(sys.argv[1]) if len(sys.argv) > 1 else 77
first = int(sys.argv[2]) if len(sys.argv) > 2 else 6
pkg = g.load_package(); api = pkg.load_library(); orc = g.load_oracle()
budget = float(os.environ.get("RT_FUZZ_SECONDS", "0"))
t0, done = time.time(), 0
for batch in range(first, first + n):
    if budget and time.time() - t0 > budget:
        break
    F.test_corrupted_but_accepted_scenes_bit_exact.__wrapped__(pkg, api, orc, batch) if hasattr(F.test_corrupted_but_accepted_scenes_bit_exact, "__wrapped__") else F.test_corrupted_but_accepted_scenes_bit_exact(pkg, api, orc, batch)
    done += 1
print(f"GPU CORRUPT FUZZ OK: batches {first}..{first + done - 1} = {40 * done} corrupted-but-accepted scenes, HIP (shipped + STATS) == oracle"
      f"{' == reference text' if os.path.exists(os.path.join(ROOT, 'oracle', '_ref', 'libref.so')) else ''}, {time.time() - t0:.0f} s")
