import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
pkg = g.load_package(); api = pkg.load_library()
sc = pkg.scenes.get(5)
mgr = sc.make_manager(None, api); mgr.bvhOnGpu = True; mgr.renderSeed = 1
for rep in range(3):
    if rep == 2: os.environ["RT_BVH_DEBUG"] = "1"
    t0 = time.perf_counter(); data = mgr.CreateAllMeshData(mgr.models); print("CreateAllMeshData %.1f ms" % (1e3 * (time.perf_counter() - t0)))
