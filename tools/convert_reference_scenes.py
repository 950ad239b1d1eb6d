#!/usr/bin/env python
"""Transcribe the reference's own scene files (Assets/Scenes/*.unity) into ray-tracing_amd/scenes_data/*.json with
ray_tracing_amd/unityscene.py — parameters only (transforms, materials, manager and camera settings); meshes that are
engine resources or missing blobs upstream become the declared procedural stand-ins below, and `cube_rounded2.obj` (an
asset of the reference) is replaced by the procedural rounded cube so that no reference file is copied.
Needs /root/reference (build machine only).   usage: python tools/convert_reference_scenes.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

REF_SCENES = "/root/reference/Assets/Scenes"
STAND_INS = {"Icosphere.obj": {"type": "icosphere", "subdivisions": 4},
             "Dragon_80K.obj": {"type": "icosphere", "subdivisions": 5, "displacement_seed": 4, "radius": 0.2},
             "Water.fbx": {"type": "quad"}, "Text.fbx": {"type": "cube"}}
SCENES = {"Glass Balls": "glass_balls", "Glass Dragon": "glass_dragon", "Sphere Refract": "sphere_refract", "Splash": "splash", "Text": "text"}


def transcribe(unity_name):
    pkg = g.load_package()
    d, notes = pkg.unityscene.load_unity_scene(os.path.join(REF_SCENES, unity_name + ".unity"), stand_ins=STAND_INS)
    replaced = []
    for name, spec in d["meshes"].items():
        if spec.get("type") == "obj":   # an asset file of the reference: not copied; procedural stand-in of the same shape class
            assert os.path.basename(spec["path"]) == "cube_rounded2.obj", spec
            d["meshes"][name] = {"type": "rounded_cube", "k": 12, "stand_in": True}
            replaced.append(f"{name} -> procedural rounded cube (1,728 triangles; the asset has 1,724)")
    d["name"] = SCENES[unity_name]
    d["provenance"] = (f"parameters transcribed by ray_tracing_amd/unityscene.py (tools/convert_reference_scenes.py) from the reference's "
                       f"Assets/Scenes/{unity_name}.unity; " + "; ".join(notes + replaced))
    return d


if __name__ == "__main__":
    out = os.path.join(ROOT, "ray-tracing_amd", "scenes_data")
    for unity_name, short in SCENES.items():
        if short == "glass_balls":
            continue   # committed in round 1 with its own provenance text; tests/test_unityscene.py re-checks it
        d = transcribe(unity_name)
        with open(os.path.join(out, short + ".json"), "w") as f:
            json.dump(d, f, indent=1)
        print(short, len(d["models"]), "models", d["settings"])
