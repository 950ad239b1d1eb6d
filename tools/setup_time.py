import os, sys, time
sys.path.insert(0, '/root/repo')
import __graft_entry__ as g
pkg = g.load_package(); api = pkg.load_library()
for cfg in (4, 5):
    for gpu_bvh in (False, True):
        sc = pkg.scenes.get(cfg)
        tr = api.create_tracer(0)
        mgr = sc.make_manager(tr, api)
        mgr.bvhOnGpu = gpu_bvh
        mgr.renderSeed = 1
        mgr.InitTexturesAndBuffers()
        t0 = time.perf_counter()
        data = mgr.CreateAllMeshData(mgr.models)
        t1 = time.perf_counter()
        tr.upload_scene(data["meshInfo"], data["triangles"], data["nodes"], mgr._pack_spheres())
        tr.synchronize()
        t2 = time.perf_counter()
        print(f"config {cfg} bvhOnGpu={gpu_bvh}: CreateAllMeshData {1e3*(t1-t0):8.1f} ms (tris {len(data['triangles'])}, nodes {len(data['nodes'])}), rt_upload_scene {1e3*(t2-t1):8.1f} ms")
        tr.close()
