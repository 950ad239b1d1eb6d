#!/usr/bin/env python
"""Render a scene (BASELINE config id or a JSON scene file, see ray_tracing_amd/sceneio.py) on the GPU:

    python tools/rt_render.py 3 --frames 8 --png out.png
    python tools/rt_render.py scene.json --size 960x540 --frames 32 --png out.png --pfm out.pfm --checkpoint ck.npz
    python tools/rt_render.py scene.json --resume ck.npz --frames 32 --png more.png
    python tools/rt_render.py "Assets/Scenes/Glass Balls.unity" --stand-in Icosphere.obj=icosphere:4 --dump-json balls.json

A Unity scene file is converted by ray_tracing_amd/unityscene.py; meshes that only exist inside the
engine or are missing on disk need `--stand-in NAME=SPEC` (SPEC: cube | quad | rounded_cube |
icosphere:SUBDIV[:DISPLACEMENT_SEED[:RADIUS]] | a JSON mesh spec); a stand-in has to have the
asset's native size, the scene only stores the Transform on top of it.
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--size", default=None, help="WxH override")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--png"); ap.add_argument("--pfm"); ap.add_argument("--checkpoint"); ap.add_argument("--resume")
    ap.add_argument("--dump-json", help="write the scene as JSON and exit (no GPU needed)")
    ap.add_argument("--assets", help="Unity Assets directory (for .unity scenes; default: the scene's project)")
    ap.add_argument("--stand-in", action="append", default=[], metavar="NAME=SPEC", help="mesh stand-in for a .unity scene")
    a = ap.parse_args()
    pkg = g.load_package()

    def mesh_spec(text):
        if text.startswith("{"):
            return json.loads(text)
        parts = text.split(":")
        spec = {"type": parts[0]}
        if parts[0] == "icosphere":
            spec["subdivisions"] = int(parts[1]) if len(parts) > 1 else 3
            if len(parts) > 2 and parts[2]:
                spec["displacement_seed"] = int(parts[2])
            if len(parts) > 3:
                spec["radius"] = float(parts[3])   # the stand-in must have the asset's native size
        return spec
    if a.scene.isdigit():
        scene = pkg.scenes.get(int(a.scene))
    elif a.scene.lower().endswith(".unity"):
        stand = dict((kv.split("=", 1)[0], mesh_spec(kv.split("=", 1)[1])) for kv in a.stand_in)
        scene = pkg.sceneio.load_scene(a.scene, assets_dir=a.assets, stand_ins=stand)
    else:
        scene = pkg.sceneio.load_scene(a.scene)
    if a.dump_json:
        pkg.sceneio.save_scene(a.dump_json, scene)
        return
    w, h = (int(x) for x in a.size.split("x")) if a.size else (scene.width, scene.height)
    api = pkg.load_library()
    tr = api.create_tracer(0)
    mgr = scene.make_manager(tr, api, w, h)
    if a.resume:
        meta = pkg.display.load_checkpoint(a.resume, mgr)
        print("resumed at frame", meta["numAccumulatedFrames"])
    else:
        mgr.OnEnable(renderSeed=a.seed)
    frames = a.frames if a.frames is not None else scene.frames
    tr.reset_counters(); tr.timer_begin()
    mgr.RenderFrames(frames)
    tr.timer_end(); c = tr.counters()
    print(json.dumps({"scene": scene.name, "size": [w, h], "frames": frames, "spp_total": (mgr.numAccumulatedFrames - 1) * mgr.numRaysPerPixel,
                      "gpu_ms": c["gpuMs"], "Mrays_per_s": c["segments"] / max(c["gpuMs"], 1e-9) / 1e3}))
    disp = pkg.display.RayTraceDisplay(mgr)
    if a.png: disp.save_png(a.png)
    if a.pfm: disp.save_pfm(a.pfm)
    if a.checkpoint: pkg.display.save_checkpoint(a.checkpoint, mgr)


if __name__ == "__main__":
    main()
