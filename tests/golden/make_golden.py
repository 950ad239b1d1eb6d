"""Generates the golden fixtures of tests/golden/ from the CPU oracle.

The reference ships no golden images and its HLSL cannot run
here, so the fixtures pin OUR restatement: any later change to oracle/rt_oracle.cpp,
include/rt_math.h, the BVH builders or the scene generators that alters a single
bit shows up as a golden mismatch.  Run from the repo root:

    python tests/golden/make_golden.py

Each .npz holds the raw accumulation SUM buffer (RGBA32F, row 0 = bottom) after
`frames` frames at renderSeed 1, plus the exact work counters.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import __graft_entry__ as graft  # noqa: E402

# name -> (config, width, height, frames, scene kwargs, manager tweaks)
CASES = {
    "config1_64x64_f2": (1, 64, 64, 2, {}, {}),
    "config2_96x54_f2": (2, 96, 54, 2, {}, {}),
    "config3_80x45_f2": (3, 80, 45, 2, {}, {}),
    "config4s3_64x36_f1": (4, 64, 36, 1, {"subdivisions": 3}, {}),
    # odd size (ragged 8x8 tiles, W and H not multiples of 8) + non-default seed handling via frames
    "config3_37x23_f3": (3, 37, 23, 3, {}, {}),
    # BVH quality Low and Disabled produce different trees, same image path
    "config3_48x27_lowq": (3, 48, 27, 1, {}, {"bvhQuality": 0}),
    "config3_48x27_nobvh": (3, 48, 27, 1, {}, {"bvhQuality": 2}),
    # the reference's own Glass Balls scene (scenes_data/glass_balls.json): 17 models, checkered walls, glass
    "glassballs_72x40_f2": (6, 72, 40, 2, {}, {}),
}
KEYS = ["segments", "innerSteps", "leafSteps", "triTests", "sphereTests", "modelVisits", "pixelFrames"]


def render_case(pkg, lib, tracer, case):
    cfg, w, h, frames, skw, tweaks = CASES[case]
    sc = pkg.scenes.get(cfg, **skw)
    mgr = sc.make_manager(tracer, lib, w, h)
    for k, v in tweaks.items():
        setattr(mgr, k, v)
    mgr.OnEnable(renderSeed=1)
    tracer.reset_counters()
    mgr.RenderFrames(frames)
    acc = tracer.read_accumulated()
    c = tracer.counters()
    return acc, np.array([c[k] for k in KEYS], dtype=np.uint64)


if __name__ == "__main__":
    pkg = graft.load_package()
    orc = graft.load_oracle()
    for case in CASES:
        tr = orc.create_tracer(8)
        acc, counters = render_case(pkg, orc, tr, case)
        tr.close()
        np.savez_compressed(os.path.join(HERE, case + ".npz"), accumulated=acc, counters=counters)
        print(case, acc.shape, counters.tolist())
