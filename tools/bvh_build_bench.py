"""Host BVH build: rt_build_bvh_mt at several thread counts vs the literal single-thread restatement (oracle)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); api = pkg.load_library(); orc = g.load_oracle()
for sub in (5, 6, 7):
    m = pkg.meshes.icosphere(sub, 1.0, 4)
    n0, t0, s0 = orc.build_bvh_arrays(m.vertices, m.normals, m.triangles)
    line = f"{m.triangle_count:8d} tris: oracle(1 thread, one sweep per candidate) {s0['timeMs']:8.1f} ms |"
    for th in (1, 2, 4, 8, 16):
        best = 1e9
        for _ in range(3):
            n1, t1, s1 = api.build_bvh_arrays_mt(m.vertices, m.normals, m.triangles, 1, th)
            best = min(best, s1["timeMs"])
        assert n1.tobytes() == n0.tobytes() and t1.tobytes() == t0.tobytes()
        line += f" {th}T {best:7.1f}"
    print(line + "  (identical output)")
