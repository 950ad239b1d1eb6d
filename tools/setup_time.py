"""Scene-load phases (ms): CreateAllMeshData (BVH builds: host, multi-threaded / GPU, one rt_build_bvh_gpu_batch call) and rt_upload_scene.
usage: python tools/setup_time.py [configs, default 4,5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); api = pkg.load_library()
m0 = pkg.meshes.icosphere(3, 1.0, 1)
api.build_bvh_arrays_gpu(m0.vertices, m0.normals, m0.triangles)   # first GPU build of the process: module load + scratch pool
for cfg in [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "4,5").split(",")]:
    for gpu_bvh in (False, True):
        sc = pkg.scenes.get(cfg)
        tr = api.create_tracer(0)
        mgr = sc.make_manager(tr, api)
        mgr.bvhOnGpu = gpu_bvh
        mgr.renderSeed = 1
        mgr.InitTexturesAndBuffers()
        best = None
        for rep in range(2):
            t0 = time.perf_counter()
            data = mgr.CreateAllMeshData(mgr.models)
            t1 = time.perf_counter()
            best = min(best or 1e9, t1 - t0)
        t1 = time.perf_counter()
        tr.upload_scene(data["meshInfo"], data["triangles"], data["nodes"], mgr._pack_spheres())
        tr.synchronize()
        t2 = time.perf_counter()
        print(f"config {cfg} bvhOnGpu={gpu_bvh}: CreateAllMeshData {1e3*best:8.1f} ms (tris {len(data['triangles'])}, nodes {len(data['nodes'])}), rt_upload_scene {1e3*(t2-t1):8.1f} ms")
        tr.close()
