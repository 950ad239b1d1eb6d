"""Host (rt_build_bvh_mt, auto threads) vs GPU (rt_build_bvh_gpu) BVH build of displaced icospheres, best of 4 calls
each (the GPU builder keeps its device scratch between calls, like a scene build that calls it once per mesh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); api = pkg.load_library()
for sub in [int(a) for a in (sys.argv[1:] or ["5", "6", "7"])]:
    m = pkg.meshes.icosphere(sub, 1.0, 4)
    res = {}
    for name, fn in (("host", lambda: api.build_bvh_arrays(m.vertices, m.normals, m.triangles, 1)),
                     ("gpu", lambda: api.build_bvh_arrays_gpu(m.vertices, m.normals, m.triangles, 1))):
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); n, t, s = fn(); dt = time.perf_counter() - t0
            best = min(best, s["timeMs"])
        res[name] = (best, n.tobytes(), t.tobytes())
    same = res["host"][1:] == res["gpu"][1:]
    print(f"{m.triangle_count:8d} triangles: host {res['host'][0]:7.2f} ms   gpu {res['gpu'][0]:7.2f} ms   x{res['host'][0] / res['gpu'][0]:.2f}   byte-identical: {same}")
