"""Python loader for oracle/_ref/libref.so — the reference's own shader text compiled as C++ (oracle/make_ref.py).
TEST INFRASTRUCTURE: only tests/, __graft_entry__ and bench.py's cpu_baseline leg import this.  Built from
/root/reference when that checkout is present (this container); on the GPU box the prebuilt library travels with the
snapshot and is used as it is."""
import ctypes as C
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("rt_make_ref", os.path.join(_HERE, "make_ref.py"))
make_ref = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_ref)


def lib_path(variant=""):
    return os.path.join(_HERE, "_ref", {"ieee": "libref_ieee.so", "spheres": "libref_spheres.so"}.get(variant, "libref.so"))


def stale_reason(libraries=("libref.so",)):
    """None, or why the libraries under oracle/_ref must not be trusted (make_ref.check_manifest): built from other reference
    sources or another recipe than oracle/REF_EXPECTED.json names, or not the files MANIFEST.json describes."""
    return make_ref.check_manifest(list(libraries))


def build(variant="", force=False):
    """(Re)build when the reference checkout is there and an input is newer than the library; else use what exists."""
    path = lib_path(variant)
    if make_ref.available():
        deps = [os.path.join(_HERE, f) for f in ("make_ref.py", "ref_compat.h", "ref_driver.h", "../include/rt_math.h", "../include/rt_abi.h")]
        deps += [os.path.join(make_ref.SHADER_DIR, s) for s in make_ref.SOURCES]
        if force or not os.path.exists(path) or any(os.path.getmtime(d) > os.path.getmtime(path) for d in deps):
            make_ref.build(ieee=(variant == "ieee"), spheres=(variant == "spheres"), quiet=True)
    return path if os.path.exists(path) else None


def load(pkg, variant=""):
    """Returns the `ref_` API bound like the oracle's (abi.Tracer drives it), or None when the library cannot be had."""
    path = build(variant)
    if path is None:
        return None
    abi = pkg.abi

    class RefApi:
        prefix = "ref_"
        # the subset of rt_abi.h the reference's dispatcher has (no spheres, display, BVH build: not in RC / RCC)
        _SIGS = {k: abi.CApi._SIGS[k] for k in (
            "destroy", "last_error", "resize", "upload_scene", "update_models", "update_spheres", "set_params",
            "reset_accumulation", "render_frame", "render_frames", "get_frame", "read_frame", "read_accumulated",
            "reset_counters", "get_counters", "version")}

        def __init__(self):
            self.path = path
            self.lib = C.CDLL(path, mode=getattr(os, "RTLD_LOCAL", 0) | getattr(os, "RTLD_NOW", 2))
            for name, (res, args) in self._SIGS.items():
                self._bind(name, res, args)
            f3 = C.POINTER(C.c_float)
            self._bind("create", C.c_int, [C.POINTER(C.c_void_p)])
            self._bind("set_threads", C.c_int, [C.c_void_p, C.c_int])
            self._bind("set_row_window", C.c_int, [C.c_void_p, C.c_int, C.c_int])
            self._bind("next_random", C.c_uint32, [C.POINTER(C.c_uint32)])
            self._bind("random_value", C.c_float, [C.POINTER(C.c_uint32)])
            self._bind("random_direction", None, [C.POINTER(C.c_uint32), f3])
            self._bind("random_point_in_circle", None, [C.POINTER(C.c_uint32), f3])
            self._bind("ray_box", C.c_float, [f3, f3, f3, f3])
            self._bind("ray_triangle", None, [f3, f3, C.c_void_p, C.c_int, f3])
            self._bind("ray_sphere", None, [f3, f3, f3, C.c_float, f3])
            self._bind("reflectance", C.c_float, [f3, f3, C.c_float, C.c_float])
            self._bind("refract", None, [f3, f3, C.c_float, C.c_float, f3])
            self._bind("environment_light", None, [C.POINTER(abi.RtParams), f3, f3])
            self._bind("material_colour", None, [C.c_void_p, f3, f3, C.c_int, f3])
            self._bind("ray_collision", None, [C.c_void_p, f3, f3, f3])
            self._bind("trace_pixel", None, [C.c_void_p, C.c_int, C.c_int, C.c_int, f3])

        def _bind(self, name, res, args):
            fn = getattr(self.lib, self.prefix + name)
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)

        def create_tracer(self, threads=1):
            h = C.c_void_p()
            rc = self.create(C.byref(h))
            if rc != abi.RT_OK:
                raise abi.RtError(rc, "ref_create failed")
            t = abi.Tracer(self, h.value)
            self.set_threads(t.h, threads)
            return t

    return RefApi()


# ---- the reference's BVH builder (BVH.cs compiled as C++): oracle/_ref/libref_bvh.so ---------------------------------
def bvh_lib_path():
    return os.path.join(_HERE, "_ref", "libref_bvh.so")


def build_bvh_lib(force=False):
    path = bvh_lib_path()
    if make_ref.bvh_available():
        deps = [os.path.join(_HERE, f) for f in ("make_ref.py", "ref_bvh_compat.h", "ref_bvh_driver.h", "../include/rt_abi.h")] + [make_ref.BVH_SOURCE]
        if force or not os.path.exists(path) or any(os.path.getmtime(d) > os.path.getmtime(path) for d in deps):
            make_ref.build_bvh(quiet=True)
    return path if os.path.exists(path) else None


def load_bvh(pkg):
    """build_bvh_arrays(verts, normals, indices, quality) of the reference's own BVH.cs text, or None when the library cannot be had."""
    path = build_bvh_lib()
    if path is None:
        return None
    import numpy as np
    abi = pkg.abi

    class RefBvh:
        def __init__(self):
            self.path = path
            self.lib = C.CDLL(path, mode=getattr(os, "RTLD_LOCAL", 0) | getattr(os, "RTLD_NOW", 2))
            self.lib.ref_bvh_build.restype = C.c_int
            self.lib.ref_bvh_build.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int),
                                               C.c_void_p, C.POINTER(abi.RtBvhStats)]
            self.lib.ref_bvh_version.restype = C.c_char_p

        def version(self):
            return self.lib.ref_bvh_version()

        def build_bvh_arrays(self, verts, normals, indices, quality=abi.BVH_QUALITY_HIGH):
            verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
            normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
            indices = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1)
            ntri = len(indices) // 3
            nodes = np.zeros(2 * max(1, ntri), dtype=abi.node_dtype)
            tris = np.zeros(ntri, dtype=abi.triangle_dtype)
            n_nodes = C.c_int(0)
            stats = abi.RtBvhStats()
            rc = self.lib.ref_bvh_build(verts.ctypes.data, normals.ctypes.data, len(verts), indices.ctypes.data, len(indices), int(quality),
                                        nodes.ctypes.data, C.byref(n_nodes), tris.ctypes.data, C.byref(stats))
            if rc != abi.RT_OK:
                raise abi.RtError(rc, "ref_bvh_build failed")
            return nodes[: n_nodes.value].copy(), tris, stats.as_dict()

    return RefBvh()
