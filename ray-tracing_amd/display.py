"""The step after the hot path: what the reference shows on screen, image files, and
checkpoint / resume of a progressive render.

`RayTraceDisplay` mirrors Assets/Scripts/Tracer/RayTraceDisplay.cs: it picks the texture
and the `Frame` divisor exactly as `OnRenderImage` does (RTD:9-23) and runs the Display
pass (Display.shader:42-47, `tex / Frame`) on the device through `rt_display`.  Note the
reference's off-by-one: `numAccumulatedFrames` has already been incremented when the blit
runs, so N accumulated frames are shown divided by N+1.  `average()` is the unbiased
`sum / alpha` for people who want the estimator rather than the reference's picture.
"""
import json
import os
import struct
import zlib

import numpy as np


class RayTraceDisplay:
    def __init__(self, raytracer):
        self.raytracer = raytracer  # a RayComputeManager

    def frame_divisor(self):
        m = self.raytracer
        return m.numAccumulatedFrames if m.accumulate else 1  # RTD:14

    def OnRenderImage(self):
        """HDR float image the reference would blit (rows bottom-up, RGBA)."""
        m = self.raytracer
        return m.tracer.display(self.frame_divisor(), use_accumulated=bool(m.accumulate))  # RTD:16-17

    def srgb8(self, flip_y=True):
        """The same after the back buffer's linear->sRGB conversion, RGBA8, top row first."""
        m = self.raytracer
        return m.tracer.display_srgb8(self.frame_divisor(), use_accumulated=bool(m.accumulate), flip_y=flip_y)

    def average(self):
        """sum / alpha: the plain Monte-Carlo mean (alpha holds the true frame count, RCC:22)."""
        acc = self.raytracer.tracer.read_accumulated()
        return acc[..., :3] / np.maximum(acc[..., 3:4], 1.0)

    def save_png(self, path):
        write_png(path, self.srgb8(flip_y=True))

    def save_pfm(self, path):
        write_pfm(path, self.OnRenderImage()[..., :3])


def write_png(path, rgba8):
    """Minimal PNG writer (RGBA8, rows top-down) — no imaging library needed on the GPU box."""
    h, w, _ = rgba8.shape
    raw = b"".join(b"\x00" + rgba8[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def write_pfm(path, rgb):
    """Portable float map, rows bottom-up like the render targets (negative scale = little endian)."""
    h, w, _ = rgb.shape
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (w, h))
        f.write(np.ascontiguousarray(rgb, dtype="<f4").tobytes())


# ---------------------------------------------------------------- checkpoint / resume
def save_checkpoint(path, manager):
    """Everything a progressive render needs to continue bit-identically: the accumulation sum,
    the frame counter and the seed (the scene itself is the caller's)."""
    tr = manager.tracer
    acc = tr.read_accumulated()
    meta = dict(width=int(tr.width), height=int(tr.height), numAccumulatedFrames=int(manager.numAccumulatedFrames),
                renderSeed=int(manager.renderSeed), numRaysPerPixel=int(manager.numRaysPerPixel),
                maxBounceCount=int(manager.maxBounceCount))
    np.savez_compressed(path, accumulated=acc, meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))


def load_checkpoint(path, manager):
    """Restore into a manager that has the same scene and size: after this, RenderFrame continues
    the sequence Frame = saved counter, saved counter + 1, ..."""
    if not path.endswith(".npz") and not os.path.exists(path):
        path = path + ".npz"
    z = np.load(path)
    meta = json.loads(bytes(z["meta"]).decode())
    tr = manager.tracer
    manager.renderSeed = meta["renderSeed"]
    manager.hasBVH = getattr(manager, "hasBVH", False)
    manager.numAccumulatedFrames = 1
    manager.InitFrame()  # sizes, scene upload, params
    if (tr.width, tr.height) != (meta["width"], meta["height"]):
        raise ValueError("checkpoint resolution %dx%d != %dx%d" % (meta["width"], meta["height"], tr.width, tr.height))
    tr.write_accumulated(z["accumulated"])
    manager.numAccumulatedFrames = meta["numAccumulatedFrames"]
    manager.SetShaderParams()
    return meta
