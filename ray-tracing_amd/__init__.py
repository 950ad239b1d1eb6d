"""raytrace-mi355x: MI355X-native path tracer behind the call surface of
SebLague/Ray-Tracing's RayComputeManager (see include/rt_abi.h).

The directory is named `ray-tracing_amd`; it is imported as the module
`ray_tracing_amd` through `__graft_entry__.load_package()`.
"""
from . import abi, dist, display, manager, meshes, sceneio, scenes, unityscene  # noqa: F401
from .hip import HipApi, HipTracer, LIB_PATH, load_library  # noqa: F401
from .manager import Camera, MatrixTransform, Model, RayComputeManager, RayTracingMaterial, Sphere, Transform  # noqa: F401
