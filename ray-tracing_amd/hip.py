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
    "rt_enable_stats", "rt_reset_counters", "rt_get_counters", "rt_build_bvh", "rt_build_bvh_mt", "rt_build_bvh_gpu", "rt_build_bvh_gpu_release", "rt_camera_view_params", "rt_version",
    "rt_debug_intersect", "rt_debug_math_eval", "rt_debug_phase_profile", "rt_build_bvh_gpu_batch",
    "rt_flush", "rt_validate_scene", "rt_debug_layout", "rt_debug_layout_free", "rt_debug_fused_frames_cap",
    "rt_create_multi", "rt_destroy_multi", "rt_multi_count", "rt_multi_context", "rt_multi_resize", "rt_multi_upload_scene",
    "rt_multi_update_models", "rt_multi_update_spheres", "rt_multi_set_params", "rt_multi_reset_accumulation",
    "rt_multi_render_frame", "rt_multi_render_frames", "rt_multi_synchronize", "rt_gather_accumulated", "rt_gather_frame",
    "rt_multi_get_counters", "rt_multi_last_gather_ms", "rt_multi_peer_access", "rt_gather_accumulated_to_device", "rt_gather_frame_to_device",
    "rt_gather_rccl",
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
        "flush": (C.c_int, [C.c_void_p]),
        "timer_begin": (C.c_int, [C.c_void_p]),
        "timer_end": (C.c_int, [C.c_void_p]),
        "enable_stats": (C.c_int, [C.c_void_p, C.c_int]),
        "build_bvh_mt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.POINTER(C.c_int), C.c_void_p, C.POINTER(abi.RtBvhStats)]),
        "build_bvh_gpu": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                    C.POINTER(C.c_int), C.c_void_p, C.POINTER(abi.RtBvhStats)]),
        "build_bvh_gpu_release": (None, []),
        "build_bvh_gpu_batch": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
        "validate_scene": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
        "debug_layout": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_char_p, C.c_void_p]),
        "debug_layout_free": (None, [C.c_void_p]),
        "debug_fused_frames_cap": (C.c_int, [C.c_void_p]),
        "debug_intersect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
        "debug_math_eval": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
        "debug_phase_profile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
        "create_multi": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
        "destroy_multi": (None, [C.c_void_p]),
        "multi_count": (C.c_int, [C.c_void_p]),
        "multi_context": (C.c_void_p, [C.c_void_p, C.c_int]),
        "multi_resize": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
        "multi_upload_scene": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
        "multi_update_models": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
        "multi_update_spheres": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
        "multi_set_params": (C.c_int, [C.c_void_p, C.POINTER(abi.RtParams)]),
        "multi_reset_accumulation": (C.c_int, [C.c_void_p]),
        "multi_render_frame": (C.c_int, [C.c_void_p]),
        "multi_render_frames": (C.c_int, [C.c_void_p, C.c_int]),
        "multi_synchronize": (C.c_int, [C.c_void_p]),
        "gather_accumulated": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
        "gather_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
        "multi_get_counters": (C.c_int, [C.c_void_p, C.POINTER(abi.RtCounters)]),
        "multi_last_gather_ms": (C.c_double, [C.c_void_p]),
        "multi_peer_access": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "gather_accumulated_to_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
        "gather_frame_to_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
        "gather_rccl": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    }

    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        super().__init__(path, "rt_")
        for name, (res, args) in self._EXTRA.items():
            self._bind(name, res, args)

    def validate_scene_arrays(self, models, triangles, nodes, spheres=None):
        """rt_validate_scene: the host half of rt_upload_scene (no device needed).  Returns {n_pairs, max_height, flat, n_filtered,
        prepare_ms}; raises RtError with the status and message rt_upload_scene would give."""
        m, nm = abi._ptr(models, abi.model_dtype)
        t, nt = abi._ptr(triangles, abi.triangle_dtype)
        n, nn = abi._ptr(nodes, abi.node_dtype)
        sp, ns = abi._ptr(spheres, abi.sphere_dtype)
        class _Info(C.Structure):
            _fields_ = [("n_pairs", C.c_int32), ("max_height", C.c_int32), ("flat", C.c_int32), ("n_filtered", C.c_int32), ("prepare_ms", C.c_float)]
        info = _Info()
        rc = self.validate_scene(m.ctypes.data if nm else None, nm, t.ctypes.data if nt else None, nt,
                                 n.ctypes.data if nn else None, nn, sp.ctypes.data if ns else None, ns, C.byref(info))
        if rc != abi.RT_OK:
            raise abi.RtError(rc, (self.last_error(None) or b"").decode(errors="replace"))
        return {k: getattr(info, k) for k, _ in _Info._fields_}

    def layout_arrays(self, models, triangles, nodes, layout=None):
        """rt_debug_layout: the device-memory layout rt_upload_scene would produce (no device needed), as numpy copies:
        {pair_space, tri_space (None: arena), norm_space (uint8), big_leaves (n,2), root_codes, tri_base, arena, used}."""
        m, nm = abi._ptr(models, abi.model_dtype)
        t, nt = abi._ptr(triangles, abi.triangle_dtype)
        n, nn = abi._ptr(nodes, abi.node_dtype)

        class _Dump(C.Structure):
            _fields_ = [("pair_space", C.c_void_p), ("pair_bytes", C.c_size_t), ("tri_space", C.c_void_p), ("tri_bytes", C.c_size_t),
                        ("norm_space", C.c_void_p), ("norm_bytes", C.c_size_t), ("big_leaves", C.c_void_p), ("n_big_leaves", C.c_size_t),
                        ("root_codes", C.c_void_p), ("tri_base", C.c_void_p), ("n_models", C.c_int32), ("arena", C.c_int32), ("used", C.c_char * 64)]
        d = _Dump()
        rc = self.debug_layout(m.ctypes.data if nm else None, nm, t.ctypes.data if nt else None, nt, n.ctypes.data if nn else None, nn,
                               layout.encode() if layout is not None else None, C.byref(d))
        if rc != abi.RT_OK:
            raise abi.RtError(rc, (self.last_error(None) or b"").decode(errors="replace"))

        def arr(ptr, count, dtype):
            if not ptr or not count:
                return np.zeros(0, dtype=dtype)
            return np.frombuffer((C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr), dtype=dtype).copy()
        try:
            return {"pair_space": arr(d.pair_space, d.pair_bytes, np.uint8), "tri_space": None if d.arena else arr(d.tri_space, d.tri_bytes, np.uint8),
                    "norm_space": arr(d.norm_space, d.norm_bytes, np.uint8), "big_leaves": arr(d.big_leaves, 2 * d.n_big_leaves, np.uint32).reshape(-1, 2),
                    "root_codes": arr(d.root_codes, d.n_models, np.uint32), "tri_base": arr(d.tri_base, d.n_models, np.int32),
                    "arena": bool(d.arena), "used": d.used.decode()}
        finally:
            self.debug_layout_free(C.byref(d))

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

    def build_bvh_arrays_gpu(self, verts, normals, indices, quality=abi.BVH_QUALITY_HIGH, device_id=0):
        """rt_build_bvh_gpu: same output as build_bvh_arrays, built on the GPU."""
        verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
        normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        indices = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1)
        ntri = len(indices) // 3
        nodes = np.zeros(2 * max(1, ntri), dtype=abi.node_dtype)
        tris = np.zeros(ntri, dtype=abi.triangle_dtype)
        n_nodes = C.c_int(0)
        stats = abi.RtBvhStats()
        rc = self.build_bvh_gpu(int(device_id), verts.ctypes.data, normals.ctypes.data, len(verts), indices.ctypes.data, len(indices),
                                int(quality), nodes.ctypes.data, C.byref(n_nodes), tris.ctypes.data, C.byref(stats))
        if rc != abi.RT_OK:
            raise abi.RtError(rc, "build_bvh_gpu failed")
        return nodes[: n_nodes.value].copy(), tris, stats.as_dict()

    def build_bvh_arrays_gpu_batch(self, meshes, quality=abi.BVH_QUALITY_HIGH, device_id=0):
        """rt_build_bvh_gpu_batch: [(verts, normals, indices), ...] -> (nodes, triangles, [(nodeOffset, triOffset, stats)]) — the
        concatenated arrays of CreateAllMeshData, written in place by the library (no per-mesh copies on this side)."""
        n = len(meshes)
        vs = [np.ascontiguousarray(m[0], dtype=np.float32).reshape(-1, 3) for m in meshes]
        ns = [np.ascontiguousarray(m[1], dtype=np.float32).reshape(-1, 3) for m in meshes]
        ix = [np.ascontiguousarray(m[2], dtype=np.int32).reshape(-1) for m in meshes]
        ntri = [len(i) // 3 for i in ix]
        nodes = np.empty(sum(2 * max(1, t) for t in ntri), dtype=abi.node_dtype)
        tris = np.empty(sum(ntri), dtype=abi.triangle_dtype)
        ptrs = lambda arrs: (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        ints = lambda vals: (C.c_int * n)(*vals)
        n_nodes, node_off, tri_off = ints([0] * n), ints([0] * n), ints([0] * n)
        stats = (abi.RtBvhStats * n)()
        rc = self.build_bvh_gpu_batch(int(device_id), n, ptrs(vs), ptrs(ns), ints([len(v) for v in vs]), ptrs(ix), ints([len(i) for i in ix]),
                                      int(quality), nodes.ctypes.data, n_nodes, node_off, tris.ctypes.data, tri_off, stats)
        if rc != abi.RT_OK:
            raise abi.RtError(rc, "build_bvh_gpu_batch failed")
        total = (node_off[n - 1] + n_nodes[n - 1]) if n else 0
        return nodes[:total], tris, [(node_off[k], tri_off[k], stats[k].as_dict()) for k in range(n)]

    def create_tracer(self, device_id=0):
        h = C.c_void_p()
        rc = self.create(device_id, C.byref(h))
        if rc != abi.RT_OK:
            msg = self.last_error(None)
            raise abi.RtError(rc, msg.decode() if msg else "rt_create failed")
        return HipTracer(self, h.value)

    def create_multi_tracer(self, device_ids):
        return MultiTracer(self, list(device_ids))


class MultiTracer:
    """rt_create_multi: n contexts (one per device id) in THIS process, cyclic 8-row strips, gather at
    readback.  Quacks like a Tracer for the RayComputeManager mirror (resize / upload_scene / update_models /
    set_params / reset_accumulation / render_frame(s) / read_accumulated)."""

    def __init__(self, api, device_ids):
        self.api = api
        ids = (C.c_int * len(device_ids))(*device_ids)
        h = C.c_void_p()
        rc = api.create_multi(ids, len(device_ids), C.byref(h))
        if rc != abi.RT_OK:
            msg = api.last_error(None)
            raise abi.RtError(rc, msg.decode() if msg else "rt_create_multi failed")
        self.h = h.value
        self.size = (0, 0)

    def _check(self, rc):
        if rc != abi.RT_OK:
            msg = self.api.last_error(self.api.multi_context(self.h, 0))
            raise abi.RtError(rc, msg.decode() if msg else "")

    def close(self):
        if self.h:
            self.api.destroy_multi(self.h)
            self.h = None

    def context(self, i):
        """Borrowed per-device context (owned by the multi handle: closing it is a no-op)."""
        t = _BorrowedTracer(self.api, self.api.multi_context(self.h, i))
        t.width, t.height = self.size
        return t

    def resize(self, w, h):
        self._check(self.api.multi_resize(self.h, w, h))
        self.size = (w, h)

    def upload_scene(self, models, triangles, nodes, spheres=None):
        m, nm = abi._ptr(models, abi.model_dtype)
        t, nt = abi._ptr(triangles, abi.triangle_dtype)
        n, nn = abi._ptr(nodes, abi.node_dtype)
        sp, ns = abi._ptr(spheres, abi.sphere_dtype)
        self._check(self.api.multi_upload_scene(
            self.h, m.ctypes.data if nm else None, nm, t.ctypes.data if nt else None, nt,
            n.ctypes.data if nn else None, nn, sp.ctypes.data if ns else None, ns))

    def update_models(self, models):
        m, nm = abi._ptr(models, abi.model_dtype)
        self._check(self.api.multi_update_models(self.h, m.ctypes.data if nm else None, nm))

    def update_spheres(self, spheres):
        sp, ns = abi._ptr(spheres, abi.sphere_dtype)
        self._check(self.api.multi_update_spheres(self.h, sp.ctypes.data if ns else None, ns))

    def set_params(self, params):
        params.abi_version = abi.RT_ABI_VERSION
        params.struct_size = C.sizeof(abi.RtParams)
        self._check(self.api.multi_set_params(self.h, C.byref(params)))

    def reset_accumulation(self):
        self._check(self.api.multi_reset_accumulation(self.h))

    def render_frame(self):
        self._check(self.api.multi_render_frame(self.h))

    def render_frames(self, n):
        self._check(self.api.multi_render_frames(self.h, n))

    def synchronize(self):
        self._check(self.api.multi_synchronize(self.h))

    def _gather(self, fn):
        w, h = self.size
        out = np.zeros((h, w, 4), dtype=np.float32)
        self._check(fn(self.h, out.ctypes.data, out.nbytes))
        return out

    def read_accumulated(self):
        return self._gather(self.api.gather_accumulated)

    def read_frame(self):
        return self._gather(self.api.gather_frame)

    def gather_accumulated_to_device(self, root, device_ptr, nbytes):
        """rt_gather_accumulated_to_device: the image lands in device memory of context `root`'s GPU (e.g. a torch tensor)."""
        self._check(self.api.gather_accumulated_to_device(self.h, int(root), device_ptr, nbytes))

    def gather_frame_to_device(self, root, device_ptr, nbytes):
        self._check(self.api.gather_frame_to_device(self.h, int(root), device_ptr, nbytes))

    def last_gather_ms(self):
        return float(self.api.multi_last_gather_ms(self.h))

    def counters(self):
        c = abi.RtCounters()
        self._check(self.api.multi_get_counters(self.h, C.byref(c)))
        return c.as_dict()


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

    def gather_rccl(self, nccl_comm, root, device_ptr, nbytes, accumulated=True):
        """rt_gather_rccl: this rank's packed tile to `root` over the caller's ncclComm_t (RCCL), de-interleaved into the whole image
        at device_ptr on root (others pass None / 0)."""
        self._check(self.api.gather_rccl(self.h, nccl_comm, int(root), 1 if accumulated else 0, device_ptr, nbytes))

    def fused_frames_cap(self):
        return self.api.debug_fused_frames_cap(self.h)

    def synchronize(self):
        self._check(self.api.synchronize(self.h))

    def flush(self):
        self._check(self.api.flush(self.h))

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
        n = len(self.PHASES)
        out = np.zeros(3 * n + 1, dtype=np.uint64)
        self._check(self.api.debug_phase_profile(self.h, out.ctypes.data, len(out)))
        prof = {p: (int(out[2 * i]), int(out[2 * i + 1])) for i, p in enumerate(self.PHASES)}
        prof["filter_violations"] = (int(out[2 * n]), 0)
        # round 6: inner steps (lane-steps) served by the LDS top-of-tree cache, and taken while >= 48 lanes / >= 3/4 of >= 16 active lanes
        # of the wave stood on ONE node (the case for a scalar top-of-tree path)
        prof["inner_from_lds_cache"] = (int(out[2 * n + 1]), 0)
        prof["inner_on_one_node_48_lanes"] = (int(out[2 * n + 2]), 0)
        prof["inner_on_one_node_3_of_4_active"] = (int(out[2 * n + 3]), 0)
        return prof

    def debug_math_eval(self, op, x, y=None):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.ascontiguousarray(np.zeros_like(x) if y is None else y, dtype=np.float32)
        out = np.empty_like(x)
        self._check(self.api.debug_math_eval(self.h, int(op), x.ctypes.data, y.ctypes.data, out.ctypes.data, len(x)))
        return out


class _BorrowedTracer(HipTracer):
    def close(self):
        self.h = None


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
