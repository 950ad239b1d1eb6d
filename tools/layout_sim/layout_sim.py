"""Which device-memory layout (RT_LAYOUT, ray-tracing_amd/csrc/rt_layout.h) touches how many cache lines — ranked on the CPU
before GPU time is spent.

The CPU oracle logs the nodes each pixel's rays visit (oracle_trace_pixel_nodes: RayTriangleBVH's pops, RC:241-283); every layout's
record addresses come from rt_debug_layout (the product's own host code); tools/layout_sim/cache_sim.c replays the streams of one
XCD's resident lanes (768 waves x 64 lanes, one pixel chain each, round-robin) through a 4 MB, 16-way, 128-byte-line LRU cache.
Reported per layout: lines per fetched record (what the vector-memory path is charged), hit rate of the warm cache, distinct lines
(the working set).  A model, not a measurement: no L1, no Infinity Cache, every lane advances one record per turn, the root
filter's skipped root steps are fetched — it ranks layouts, the GPU A/B (profiles/r05_ab_layout*.txt) decides.

usage: python tools/layout_sim/layout_sim.py <config> [--tiles 768] [--region all|block] [--layouts a;b;c] [--out file]
"""
import argparse
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
LEAF = 0x80000000
TRI_SPACE = 1 << 36  # separate spaces are far apart


def build_sim():
    so = os.path.join(HERE, "cache_sim.so")
    src = os.path.join(HERE, "cache_sim.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, src])
    lib = C.CDLL(so)

    class SimOut(C.Structure):
        _fields_ = [(k, C.c_uint64) for k in ("line_accesses", "line_hits", "records", "record_lines", "warm_accesses", "warm_hits", "distinct_lines")]
    lib.simulate.restype = C.c_int
    lib.simulate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(SimOut)]
    return lib, SimOut


def address_maps(models, nodes, lay):
    """node index -> (byte address, bytes) of what a visit of that node fetches: for a first-child index the pair record of its parent,
    for a leaf node its run of triangle records"""
    pair_space = lay["pair_space"]
    tri_off = 0 if lay["arena"] else TRI_SPACE
    pair_addr = np.full(len(nodes), -1, dtype=np.int64)
    leaf_addr = np.full(len(nodes), -1, dtype=np.int64)
    leaf_bytes = np.zeros(len(nodes), dtype=np.int64)
    codes = pair_space.view(np.uint32)
    big = lay["big_leaves"]
    done = set()
    for mi, m in enumerate(models):
        node_off, base = int(m["nodeOffset"]), int(lay["tri_base"][mi])
        if (node_off, base) in done:
            continue
        done.add((node_off, base))
        root, code = nodes[node_off], int(lay["root_codes"][mi])

        def leaf(ni, code):
            c, start = (code >> 24) & 0x7F, code & 0xFFFFFF
            if c == 0:
                start, c = (int(x) for x in big[start])
            leaf_addr[ni] = tri_off + (base + start) * 16
            leaf_bytes[ni] = c * 48
        if root["triangleCount"] > 0:
            leaf(node_off, code)
            continue
        stack = [(node_off, code)]
        while stack:
            ni, unit = stack.pop()
            first = node_off + int(nodes[ni]["startIndex"])
            pair_addr[first] = unit * 16
            for side in range(2):
                cc = int(codes[unit * 4 + 12 + side])
                if cc & LEAF:
                    leaf(first + side, cc)
                else:
                    stack.append((first + side, cc))
    return pair_addr, leaf_addr, leaf_bytes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", type=int)
    ap.add_argument("--tiles", type=int, default=768, help="resident waves of one XCD (32 CUs x 24)")
    ap.add_argument("--region", default="all", choices=["all", "block"], help="all: the XCD's tiles are spread over the image (longest-chain-first order); "
                    "block: they come from one of 4 x 2 blocks of it (RT_XCD_AFFINITY=2)")
    ap.add_argument("--layouts", default="dense;pre;hot=6;hot=10;align;arena;pre,arena;pre,arena,palign;pre,hot=8,arena")
    ap.add_argument("--cache-mb", type=float, default=4.0)
    ap.add_argument("--subdivisions", type=int, default=None)
    ap.add_argument("--out")
    a = ap.parse_args()
    lib, SimOut = build_sim()
    pkg = g.load_package()
    api = pkg.load_library()
    orc = g.load_oracle()
    orc._bind("trace_pixel_nodes", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int])
    kw = {"subdivisions": a.subdivisions} if a.subdivisions else {}
    sc = pkg.scenes.get(a.config, **kw)
    tr = orc.create_tracer(1)
    mgr = sc.make_manager(tr, orc)
    mgr.OnEnable(renderSeed=1)
    mgr.InitFrame()
    data = mgr.CreateAllMeshData(mgr.models)
    models, tris, nodes = data["meshInfo"], data["triangles"], data["nodes"]
    W, H = sc.width, sc.height
    tx, ty = W // 8, H // 8
    rng = np.random.default_rng(7)
    if a.region == "all":
        tiles = rng.choice(tx * ty, size=a.tiles, replace=False)
    else:  # block (1, 0) of 4 x 2: a block that looks at the middle of the scene
        bx0, bx1, by0, by1 = tx // 4, tx // 2, 0, ty // 2
        cand = np.array([y * tx + x for y in range(by0, by1) for x in range(bx0, bx1)])
        tiles = rng.choice(cand, size=min(a.tiles, len(cand)), replace=False)
    t0 = time.time()
    buf = (C.c_uint32 * (1 << 22))()
    streams = []
    for t in tiles:
        cx, cy = int(t) % tx, int(t) // tx
        for s in range(64):
            n = orc.trace_pixel_nodes(tr.h, cx * 8 + (s & 7), cy * 8 + (s >> 3), 1, buf, len(buf))
            assert n <= len(buf)
            v = np.frombuffer(buf, dtype=np.uint32, count=n)
            streams.append(v[v != 0xFFFFFFFF].copy())
    tr.close()
    starts = np.zeros(len(streams) + 1, dtype=np.uint64)
    starts[1:] = np.cumsum([len(s) for s in streams])
    visits = np.concatenate(streams)
    is_leaf = (visits & LEAF) != 0
    idx = (visits & 0x7FFFFFFF).astype(np.int64)
    lines_out = [f"# config {a.config}{' subdivisions ' + str(a.subdivisions) if a.subdivisions else ''}: {len(tiles)} tiles ({a.region}) = {len(streams)} pixel chains of frame 1, "
                 f"{len(visits)} node visits ({int(is_leaf.sum())} leaves), traced in {time.time() - t0:.1f} s; cache {a.cache_mb} MB, 16-way, 128-byte lines",
                 f"# {'layout':22s} {'spaces MB':>10s} {'lines/record':>12s} {'pair l/r':>9s} {'leaf l/r':>9s} {'warm hit':>9s} {'miss lines/visit':>16s} {'distinct MB':>11s}"]
    print("\n".join(lines_out))
    for name in a.layouts.split(";"):
        lay = api.layout_arrays(models, tris, nodes, name)
        pair_addr, leaf_addr, leaf_bytes = address_maps(models, nodes, lay)
        addr = np.where(is_leaf, leaf_addr[idx], pair_addr[idx])
        nbytes = np.where(is_leaf, leaf_bytes[idx], 64).astype(np.uint32)
        assert (addr >= 0).all()
        addr = addr.astype(np.uint64)
        out = SimOut()
        rc = lib.simulate(starts.ctypes.data, len(streams), addr.ctypes.data, nbytes.ctypes.data, 128, int(a.cache_mb * (1 << 20)), 16, 0.3, C.byref(out))
        assert rc == 0
        a64 = addr.astype(np.int64)
        nl = ((a64 + nbytes - 1) // 128 - a64 // 128 + 1)
        size_mb = (len(lay["pair_space"]) + (0 if lay["arena"] else len(lay["tri_space"]))) / 1e6
        warm_hit = out.warm_hits / max(1, out.warm_accesses)
        row = (f"  {lay['used']:22s} {size_mb:10.2f} {out.record_lines / out.records:12.3f} {nl[~is_leaf].mean():9.3f} {nl[is_leaf].mean():9.3f} "
               f"{warm_hit:9.4f} {(1 - warm_hit) * out.record_lines / out.records:16.4f} {out.distinct_lines * 128 / 1e6:11.2f}")
        print(row, flush=True)
        lines_out.append(row)
    if a.out:
        with open(a.out, "a") as f:
            f.write("\n".join(lines_out) + "\n\n")


if __name__ == "__main__":
    main()
