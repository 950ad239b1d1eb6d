"""Loader and context wrapper for libraytrace_hip.so (the product path).

There is no fallback: if the shared library is missing or no HIP device is
usable, this raises — rendering never silently runs anywhere else.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_HIP_LIB") or os.path.join(_HERE, "lib", "libraytrace_hip.so")  # RT_HIP_LIB: A/B builds

# every symbol include/rt_abi.h declares (tests check the library exports them all)
ABI_SYMBOLS = [
    "rt_create", "rt_destroy", "rt_last_error", "rt_set_stream", "rt_resize", "rt_set_partition", "rt_local_rows",
    "rt_local_to_global_row", "rt_bind_render_targets", "rt_get_render_targets", "rt_upload_scene", "rt_update_models",
    "rt_update_spheres", "rt_set_params", "rt_reset_accumulation", "rt_render_frame", "rt_render_frames",
    "rt_synchronize", "rt_get_frame", "rt_read_frame", "rt_read_accumulated", "rt_display", "rt_display_srgb8",
    "rt_write_accumulated", "rt_timer_begin", "rt_timer_end",
    "rt_enable_stats", "rt_reset_counters", "rt_get_counters", "rt_build_bvh", "rt_build_bvh_mt", "rt_camera_view_params", "rt_version",
    "rt_debug_intersect", "rt_debug_math_eval", "rt_debug_phase_profile",
]


class HipApi(abi.CApi):
    _EXTRA = {
        "create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
        "set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
        "set_partition": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
        "local_rows": (C.c_int, [C.c_void_p]),
        "local_to_global_row": (C.c_int, [C.c_void_p, C.c_int]),
        "bind_render_targets": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
        "get_render_targets": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
        "synchronize": (C.c_int, [C.c_void_p]),
        "timer_begin": (C.c_int, [C.c_void_p]),
        "timer_end": (C.c_int, [C.c_void_p]),
        "enable_stats": (C.c_int, [C.c_void_p, C.c_int]),
        "build_bvh_mt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.POINTER(C.c_int), C.c_void_p, C.POINTER(abi.RtBvhStats)]),
        "debug_intersect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
        "debug_math_eval": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
        "debug_phase_profile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    }

    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        super().__init__(path, "rt_")
        for name, (res, args) in self._EXTRA.items():
            self._bind(name, res, args)

    def build_bvh_arrays_mt(self, verts, normals, indices, quality=abi.BVH_QUALITY_HIGH, threads=0):
        """rt_build_bvh_mt: same output as build_bvh_arrays, on `threads` host threads."""
        verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
        normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        indices = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1)
        ntri = len(indices) // 3
        nodes = np.zeros(2 * max(1, ntri), dtype=abi.node_dtype)
        tris = np.zeros(ntri, dtype=abi.triangle_dtype)
        n_nodes = C.c_int(0)
        stats = abi.RtBvhStats()
        rc = self.build_bvh_mt(verts.ctypes.data, normals.ctypes.data, len(verts), indices.ctypes.data, len(indices),
                               int(quality), int(threads), nodes.ctypes.data, C.byref(n_nodes), tris.ctypes.data, C.byref(stats))
        if rc != abi.RT_OK:
            raise abi.RtError(rc, "build_bvh_mt failed")
        return nodes[: n_nodes.value].copy(), tris, stats.as_dict()

    def create_tracer(self, device_id=0):
        h = C.c_void_p()
        rc = self.create(device_id, C.byref(h))
        if rc != abi.RT_OK:
            msg = self.last_error(None)
            raise abi.RtError(rc, msg.decode() if msg else "rt_create failed")
        return HipTracer(self, h.value)


class HipTracer(abi.Tracer):
    """An RtContext on one MI355X."""

    def set_stream(self, stream_ptr):
        self._check(self.api.set_stream(self.h, stream_ptr))

    def set_partition(self, strip_rows, part_index, part_count):
        self._check(self.api.set_partition(self.h, strip_rows, part_index, part_count))

    def local_rows(self):
        return self.api.local_rows(self.h)

    def local_to_global_rows(self):
        n = self.local_rows()
        return np.array([self.api.local_to_global_row(self.h, i) for i in range(n)], dtype=np.int64)

    def bind_render_targets(self, frame_ptr, accum_ptr):
        self._check(self.api.bind_render_targets(self.h, frame_ptr, accum_ptr))

    def render_targets(self):
        f, a = C.c_void_p(), C.c_void_p()
        self._check(self.api.get_render_targets(self.h, C.byref(f), C.byref(a)))
        return f.value, a.value

    def synchronize(self):
        self._check(self.api.synchronize(self.h))

    def timer_begin(self):
        self._check(self.api.timer_begin(self.h))

    def timer_end(self):
        self._check(self.api.timer_end(self.h))

    def enable_stats(self, on=True):
        self._check(self.api.enable_stats(self.h, 1 if on else 0))

    def debug_intersect(self, origins, dirs):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        out = np.zeros((len(o), 10), dtype=np.float32)
        self._check(self.api.debug_intersect(self.h, o.ctypes.data, d.ctypes.data, len(o), out.ctypes.data))
        return out

    PHASES = ["loop", "raygen", "spheres", "traverse_call", "model", "inner", "tri", "shade_hit", "sky", "sphere_roots", "glass", "refill"]

    def phase_profile(self):
        out = np.zeros(2 * len(self.PHASES) + 1, dtype=np.uint64)
        self._check(self.api.debug_phase_profile(self.h, out.ctypes.data, len(out)))
        prof = {p: (int(out[2 * i]), int(out[2 * i + 1])) for i, p in enumerate(self.PHASES)}
        prof["filter_violations"] = (int(out[-1]), 0)
        return prof

    def debug_math_eval(self, op, x, y=None):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.ascontiguousarray(np.zeros_like(x) if y is None else y, dtype=np.float32)
        out = np.empty_like(x)
        self._check(self.api.debug_math_eval(self.h, int(op), x.ctypes.data, y.ctypes.data, out.ctypes.data, len(x)))
        return out


_api = None


def load_library():
    global _api
    if _api is None:
        _api = HipApi()
    return _api


def scatter_rows(local_image, global_rows, height):
    """Place the packed local rows of one partition into a full-height image."""
    out = np.zeros((height,) + local_image.shape[1:], dtype=local_image.dtype)
    out[global_rows] = local_image
    return out
