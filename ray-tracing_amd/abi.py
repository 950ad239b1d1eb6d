"""ctypes / numpy view of include/rt_abi.h.

The numpy dtypes are the byte-exact layouts of the reference's structured
buffers (RayCommon.hlsl:49-95, RayTracingMaterial.cs:15-27,
RayComputeManager.cs:256-263); `CApi` binds the entry points of a shared
library exporting the rt_abi.h surface under a given symbol prefix.
"""
import ctypes as C
import os

import numpy as np

RT_ABI_VERSION = 1

RT_OK = 0
RT_ERR_INVALID_ARG = -1
RT_ERR_ABI_MISMATCH = -2
RT_ERR_NO_DEVICE = -3
RT_ERR_HIP = -4
RT_ERR_STATE = -5
RT_ERR_SCENE = -6
RT_ERR_OOM = -7

MATERIAL_DEFAULT = 0
MATERIAL_CHECKERED = 1
MATERIAL_GLASS = 2

BVH_QUALITY_LOW = 0
BVH_QUALITY_HIGH = 1
BVH_QUALITY_DISABLED = 2

material_dtype = np.dtype([
    ("diffuseCol", "<f4", 4), ("emissionCol", "<f4", 4), ("specularCol", "<f4", 4), ("absorption", "<f4", 4),
    ("absorptionStrength", "<f4"), ("emissionStrength", "<f4"), ("smoothness", "<f4"),
    ("specularProbability", "<f4"), ("ior", "<f4"), ("flag", "<i4")])
model_dtype = np.dtype([
    ("nodeOffset", "<i4"), ("triOffset", "<i4"), ("worldToLocal", "<f4", 16), ("localToWorld", "<f4", 16),
    ("material", material_dtype)])
triangle_dtype = np.dtype([
    ("posA", "<f4", 3), ("posB", "<f4", 3), ("posC", "<f4", 3),
    ("normA", "<f4", 3), ("normB", "<f4", 3), ("normC", "<f4", 3)])
node_dtype = np.dtype([
    ("boundsMin", "<f4", 3), ("boundsMax", "<f4", 3), ("startIndex", "<i4"), ("triangleCount", "<i4")])
sphere_dtype = np.dtype([("centre", "<f4", 3), ("radius", "<f4"), ("material", material_dtype)])

assert material_dtype.itemsize == 88
assert model_dtype.itemsize == 224
assert triangle_dtype.itemsize == 72
assert node_dtype.itemsize == 32
assert sphere_dtype.itemsize == 104


class RtParams(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("struct_size", C.c_uint32),
        ("maxBounceCount", C.c_int32), ("numRaysPerPixel", C.c_int32), ("frame", C.c_int32),
        ("renderSeed", C.c_int32), ("useSky", C.c_int32), ("accumulate", C.c_int32),
        ("defocusStrength", C.c_float), ("divergeStrength", C.c_float),
        ("sunFocus", C.c_float), ("sunIntensity", C.c_float),
        ("sunColour", C.c_float * 3), ("dirToSun", C.c_float * 3), ("viewParams", C.c_float * 3),
        ("camLocalToWorld", C.c_float * 16)]


class RtCounters(C.Structure):
    _fields_ = [
        ("segments", C.c_uint64), ("innerSteps", C.c_uint64), ("leafSteps", C.c_uint64),
        ("triTests", C.c_uint64), ("sphereTests", C.c_uint64), ("modelVisits", C.c_uint64),
        ("pixelFrames", C.c_uint64), ("gpuMs", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class RtBvhStats(C.Structure):
    _fields_ = [
        ("triangleCount", C.c_int32), ("totalNodeCount", C.c_int32), ("leafNodeCount", C.c_int32),
        ("leafDepthMax", C.c_int32), ("leafDepthMin", C.c_int32), ("leafDepthSum", C.c_int32),
        ("leafMaxTriCount", C.c_int32), ("leafMinTriCount", C.c_int32), ("quality", C.c_int32),
        ("timeMs", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def algorithmic_bytes(c, n_models, n_spheres):
    """bytes the reference's loop would move for the counted work.

    96 B per inner step (popped node re-read + two children, RC:245,266-267),
    32 B per leaf step + 72 B per triangle test (RC:245,252), 224 B per
    (segment x model) (RC:349), 104 B per (segment x sphere), 48 B per pixel per
    frame (RCC:18,22: 16 B write + 16 B read + 16 B write).
    """
    d = c if isinstance(c, dict) else c.as_dict()
    return (96 * d["innerSteps"] + 32 * d["leafSteps"] + 72 * d["triTests"]
            + 224 * d["segments"] * n_models + 104 * d["segments"] * n_spheres + 48 * d["pixelFrames"])


class RtError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"rt status {status}: {message}")
        self.status = status


def _ptr(a, dtype):
    if a is None:
        return None, 0
    a = np.ascontiguousarray(a, dtype=dtype)
    return a, len(a)


class CApi:
    """Binds the rt_abi.h entry points of `path` whose symbols start with `prefix`."""

    # name -> (restype, argtypes); ctx pointers are void*
    _SIGS = {
        "destroy": (None, [C.c_void_p]),
        "last_error": (C.c_char_p, [C.c_void_p]),
        "resize": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
        "upload_scene": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_int]),
        "update_models": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
        "update_spheres": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
        "set_params": (C.c_int, [C.c_void_p, C.POINTER(RtParams)]),
        "reset_accumulation": (C.c_int, [C.c_void_p]),
        "render_frame": (C.c_int, [C.c_void_p]),
        "render_frames": (C.c_int, [C.c_void_p, C.c_int]),
        "get_frame": (C.c_int, [C.c_void_p]),
        "read_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
        "read_accumulated": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
        "display": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
        "display_srgb8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
        "write_accumulated": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
        "reset_counters": (C.c_int, [C.c_void_p]),
        "get_counters": (C.c_int, [C.c_void_p, C.POINTER(RtCounters)]),
        "build_bvh": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                C.POINTER(C.c_int), C.c_void_p, C.POINTER(RtBvhStats)]),
        "camera_view_params": (C.c_int, [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float * 3)]),
        "version": (C.c_char_p, []),
    }

    def __init__(self, path, prefix):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.path = path
        self.prefix = prefix
        self.lib = C.CDLL(path, mode=getattr(os, "RTLD_LOCAL", 0) | getattr(os, "RTLD_NOW", 2))
        for name, (res, args) in self._SIGS.items():
            self._bind(name, res, args)

    def _bind(self, name, res, args):
        fn = getattr(self.lib, self.prefix + name)
        fn.restype = res
        fn.argtypes = args
        setattr(self, name, fn)
        return fn

    def has(self, name):
        return hasattr(self.lib, self.prefix + name)

    # ---- host helpers that need no context -------------------------------
    def build_bvh_arrays(self, verts, normals, indices, quality=BVH_QUALITY_HIGH):
        """BVH.cs ctor (BVH:26): returns (nodes, triangles, stats-dict)."""
        verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
        normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        indices = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1)
        ntri = len(indices) // 3
        nodes = np.zeros(2 * max(1, ntri), dtype=node_dtype)
        tris = np.zeros(ntri, dtype=triangle_dtype)
        n_nodes = C.c_int(0)
        stats = RtBvhStats()
        rc = self.build_bvh(verts.ctypes.data, normals.ctypes.data, len(verts), indices.ctypes.data, len(indices),
                            int(quality), nodes.ctypes.data, C.byref(n_nodes), tris.ctypes.data, C.byref(stats))
        if rc != RT_OK:
            raise RtError(rc, "build_bvh failed")
        return nodes[: n_nodes.value].copy(), tris, stats.as_dict()

    def view_params(self, fov_deg, aspect, focus_distance):
        out = (C.c_float * 3)()
        rc = self.camera_view_params(fov_deg, aspect, focus_distance, C.byref(out))
        if rc != RT_OK:
            raise RtError(rc, "camera_view_params failed")
        return [out[0], out[1], out[2]]


class Tracer:
    """One rendering context of a CApi library (the HIP library in the product;
    tests wrap the oracle library in the same class to swap implementations)."""

    def __init__(self, api, handle):
        self.api = api
        self.h = C.c_void_p(handle)
        self.width = 0
        self.height = 0
        self.n_models = 0
        self.n_spheres = 0

    def _check(self, rc):
        if rc != RT_OK:
            msg = self.api.last_error(self.h)
            raise RtError(rc, msg.decode() if msg else "")

    def close(self):
        if self.h:
            self.api.destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def resize(self, w, h):
        self._check(self.api.resize(self.h, w, h))
        self.width, self.height = w, h

    def local_rows(self):
        return self.height

    def upload_scene(self, models, triangles, nodes, spheres=None):
        m, nm = _ptr(models, model_dtype)
        t, nt = _ptr(triangles, triangle_dtype)
        n, nn = _ptr(nodes, node_dtype)
        s, ns = _ptr(spheres, sphere_dtype)
        self._check(self.api.upload_scene(
            self.h, m.ctypes.data if nm else None, nm, t.ctypes.data if nt else None, nt,
            n.ctypes.data if nn else None, nn, s.ctypes.data if ns else None, ns))
        self.n_models, self.n_spheres = nm, ns

    def update_models(self, models):
        m, nm = _ptr(models, model_dtype)
        self._check(self.api.update_models(self.h, m.ctypes.data if nm else None, nm))

    def update_spheres(self, spheres):
        s, ns = _ptr(spheres, sphere_dtype)
        self._check(self.api.update_spheres(self.h, s.ctypes.data if ns else None, ns))

    def set_params(self, params):
        params.abi_version = RT_ABI_VERSION
        params.struct_size = C.sizeof(RtParams)
        self._check(self.api.set_params(self.h, C.byref(params)))

    def reset_accumulation(self):
        self._check(self.api.reset_accumulation(self.h))

    def render_frame(self):
        self._check(self.api.render_frame(self.h))

    def render_frames(self, n):
        self._check(self.api.render_frames(self.h, n))

    def frame(self):
        return self.api.get_frame(self.h)

    def _read(self, fn):
        out = np.empty((self.local_rows(), self.width, 4), dtype=np.float32)
        self._check(fn(self.h, out.ctypes.data, out.nbytes))
        return out

    def read_frame(self):
        return self._read(self.api.read_frame)

    def read_accumulated(self):
        return self._read(self.api.read_accumulated)

    def display(self, frame, use_accumulated=True):
        """Display.shader: tex / Frame (HDR float image, local rows, row 0 = bottom)."""
        out = np.empty((self.local_rows(), self.width, 4), dtype=np.float32)
        self._check(self.api.display(self.h, int(frame), 1 if use_accumulated else 0, out.ctypes.data, out.nbytes))
        return out

    def display_srgb8(self, frame, use_accumulated=True, flip_y=True):
        out = np.empty((self.local_rows(), self.width, 4), dtype=np.uint8)
        self._check(self.api.display_srgb8(self.h, int(frame), 1 if use_accumulated else 0, 1 if flip_y else 0, out.ctypes.data, out.nbytes))
        return out

    def write_accumulated(self, image):
        image = np.ascontiguousarray(image, dtype=np.float32)
        self._check(self.api.write_accumulated(self.h, image.ctypes.data, image.nbytes))

    def reset_counters(self):
        self._check(self.api.reset_counters(self.h))

    def counters(self):
        c = RtCounters()
        self._check(self.api.get_counters(self.h, C.byref(c)))
        return c.as_dict()
