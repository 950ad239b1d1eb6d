"""Python loader for the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force=False, variant=""):
    """variant "" = the contract of include/rt_math.h; "ieee" = the RT_MATH_IEEE reading (bracket tests only)."""
    name = "liboracle_ieee.so" if variant == "ieee" else "liboracle.so"
    path = os.path.join(_HERE, name)
    if force or not os.path.exists(path) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(path)
            for f in ("rt_oracle.cpp", "../include/rt_math.h", "../include/rt_abi.h")):
        subprocess.check_call(["make", "-C", _HERE, "-B", name], stdout=subprocess.DEVNULL)
    return path


def load(pkg, variant=""):
    """Returns (api, tracer_factory) for the oracle, bound with the package's generic CApi."""
    abi = pkg.abi

    class OracleApi(abi.CApi):
        def __init__(self):
            super().__init__(build(variant=variant), "oracle_")
            L = self.lib
            self._bind("create", C.c_int, [C.POINTER(C.c_void_p)])
            self._bind("set_threads", C.c_int, [C.c_void_p, C.c_int])
            self._bind("set_row_window", C.c_int, [C.c_void_p, C.c_int, C.c_int])
            f3 = C.POINTER(C.c_float)
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
            self._bind("ray_collision_bruteforce", None, [C.c_void_p, f3, f3, f3])
            self._bind("trace_pixel", None, [C.c_void_p, C.c_int, C.c_int, C.c_int, f3])
            self._bind("math_eval", None, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int])
            del L

        def create_tracer(self, threads=1):
            h = C.c_void_p()
            rc = self.create(C.byref(h))
            if rc != abi.RT_OK:
                raise abi.RtError(rc, "oracle_create failed")
            t = abi.Tracer(self, h.value)
            self.set_threads(t.h, threads)
            return t

    return OracleApi()
