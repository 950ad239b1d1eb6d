"""Host-inclusive rates of the headline configuration: the reference keeps its render targets on the device (RCM:84-95 blits the
RenderTexture), so `value` has no host transfer in it.  This tool times what a host that DOES take the pixels over PCIe sees:
  (a) K frames, then ONE rt_read_accumulated (the end of a render job);  (b) rt_read_accumulated after EVERY frame (a host-side viewer).
usage: python tools/pcie_rate.py [config=2] [K=20]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pkg = g.load_package(); api = pkg.load_library()
sc = pkg.scenes.get(cfg)
tr = api.create_tracer(0)
mgr = sc.make_manager(tr, api)
mgr.bvhOnGpu = True
mgr.OnEnable(renderSeed=1)
for _ in range(5):
    mgr.RenderFrame()
tr.read_accumulated()
tr.synchronize()


def timed(body):
    best = None
    for _ in range(3):
        tr.reset_counters()
        tr.synchronize()
        t0 = time.perf_counter()
        body()
        tr.synchronize()
        dt = time.perf_counter() - t0
        seg = tr.counters()["segments"]
        if best is None or dt < best[0]:
            best = (dt, seg)
    return best


def device_only():
    for _ in range(K):
        tr.render_frame()


def one_read():
    for _ in range(K):
        tr.render_frame()
    tr.read_accumulated()


def read_every_frame():
    for _ in range(K):
        tr.render_frame()
        tr.read_accumulated()


W, H = sc.width, sc.height
mb = W * H * 16 / 1e6
for name, body in (("device only (bench.py's region)", device_only), (f"K frames + one rt_read_accumulated ({mb:.1f} MB)", one_read),
                   (f"rt_read_accumulated after every frame ({mb:.1f} MB each)", read_every_frame)):
    dt, seg = timed(body)
    print(f"config {cfg} {W}x{H} K={K}: {name:58s} {1e3 * dt / K:7.3f} ms per frame  {seg / dt / 1e6:9.0f} Mrays/s")
t0 = time.perf_counter(); tr.read_accumulated(); dt = time.perf_counter() - t0
print(f"one rt_read_accumulated alone: {1e3 * dt:.2f} ms = {mb / 1e3 / dt:.1f} GB/s into a pageable numpy array")
