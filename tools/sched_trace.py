"""Dump the per-pixel work logs (oracle_trace_pixel_schedule) of whole 8x8 tiles of a configuration,
for tools/sched_sim2.py.  usage: python tools/sched_trace.py <config> <n_tiles> <out.npz>
Tiles are spread over the 1920x1080 image on a regular grid (the kernel hands whole tiles to waves)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g

cfg, ntiles, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
pkg = g.load_package()
orc = g.load_oracle()
orc._bind("trace_pixel_schedule", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int])
tr = orc.create_tracer(1)
sc = pkg.scenes.get(cfg)
mgr = sc.make_manager(tr, orc)
mgr.OnEnable(renderSeed=1)
mgr.InitFrame()
W, H = sc.width, sc.height
tx, ty = W // 8, H // 8
rng = np.random.default_rng(5)
# a jittered grid of tiles
side = int(np.ceil(np.sqrt(ntiles * tx / ty)))
tiles = []
for j in range(int(np.ceil(ntiles / side))):
    for i in range(side):
        if len(tiles) < ntiles:
            tiles.append((int((i + rng.random()) * tx / side) % tx, int((j + rng.random()) * ty / max(1, int(np.ceil(ntiles / side)))) % ty))
buf = (C.c_uint8 * (1 << 22))()
logs, offs = [], [0]
for (cx, cy) in tiles:
    for s in range(64):
        x, y = cx * 8 + (s & 7), cy * 8 + (s >> 3)
        n = orc.trace_pixel_schedule(tr.h, x, y, 0, buf, len(buf))
        assert n <= len(buf)
        logs.append(np.frombuffer(buf, dtype=np.uint8, count=n).copy())
        offs.append(offs[-1] + n)
np.savez_compressed(out, data=np.concatenate(logs), offs=np.array(offs, dtype=np.int64), tiles=np.array(tiles), n_models=len(sc.models))
print(f"config {cfg}: {len(tiles)} tiles, {offs[-1]} bytes of work log")
