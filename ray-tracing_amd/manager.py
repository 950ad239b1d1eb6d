"""Host-side mirror of the reference dispatcher.

`RayComputeManager` keeps the reference's field and method names
(Assets/Scripts/Tracer/RayComputeManager.cs, "RCM") so code and tests written
against the Unity component read the same here; the Unity objects it touches are
replaced by plain data classes:

    UnityEngine.Transform / Camera   -> Transform, Camera (TRS -> Matrix4x4)
    Model (Types/Model.cs)           -> Model  (mesh + material + transform)
    RayTracingMaterial (struct)      -> RayTracingMaterial
    ComputeShader + ComputeBuffers   -> a `Tracer` (rt_abi.h context)

The manager only talks to its tracer through the C ABI, so the same class drives
libraytrace_hip.so (the product) and, in the tests, the CPU oracle.
"""
import math

import numpy as np

from . import abi


class RayTracingMaterial:
    """Types/RayTracingMaterial.cs:4-38 (88-byte struct)."""

    def __init__(self, **kw):
        self.SetDefaultValues()
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)

    def SetDefaultValues(self):  # MAT:29-38
        self.flag = abi.MATERIAL_DEFAULT
        self.diffuseCol = (1.0, 1.0, 1.0, 1.0)
        self.emissionCol = (0.0, 0.0, 0.0, 0.0)
        self.specularCol = (1.0, 1.0, 1.0, 1.0)
        self.absorption = (0.0, 0.0, 0.0, 0.0)
        self.absorptionMultiplier = 0.0
        self.emissionStrength = 0.0
        self.smoothness = 0.0
        self.specularProbability = 1.0
        self.ior = 1.0

    @staticmethod
    def _col(c):
        c = tuple(float(x) for x in c)
        return c if len(c) == 4 else c + (1.0,)

    def pack(self):
        m = np.zeros((), dtype=abi.material_dtype)
        m["diffuseCol"] = self._col(self.diffuseCol)
        m["emissionCol"] = self._col(self.emissionCol)
        m["specularCol"] = self._col(self.specularCol)
        m["absorption"] = self._col(self.absorption)
        m["absorptionStrength"] = self.absorptionMultiplier
        m["emissionStrength"] = self.emissionStrength
        m["smoothness"] = self.smoothness
        m["specularProbability"] = self.specularProbability
        m["ior"] = self.ior
        m["flag"] = int(self.flag)
        return m


def _quat_from_euler_deg(x, y, z):
    """Unity Quaternion.Euler: rotation z, then x, then y (extrinsic), left-handed."""
    hx, hy, hz = (math.radians(a) * 0.5 for a in (x, y, z))
    cx, sx, cy, sy, cz, sz = math.cos(hx), math.sin(hx), math.cos(hy), math.sin(hy), math.cos(hz), math.sin(hz)
    # q = qy * qx * qz
    qw = cy * cx * cz + sy * sx * sz
    qx = cy * sx * cz + sy * cx * sz
    qy = sy * cx * cz - cy * sx * sz
    qz = cy * cx * sz - sy * sx * cz
    return (qx, qy, qz, qw)


class Transform:
    """Position / rotation (Euler degrees, Unity order) / scale -> localToWorldMatrix."""

    def __init__(self, position=(0, 0, 0), euler=(0, 0, 0), scale=(1, 1, 1)):
        self.position = tuple(float(v) for v in position)
        self.euler = tuple(float(v) for v in euler)
        s = (scale, scale, scale) if np.isscalar(scale) else scale
        self.scale = tuple(float(v) for v in s)

    def rotation_matrix(self):
        x, y, z, w = _quat_from_euler_deg(*self.euler)
        return np.array([
            [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float64)

    @property
    def localToWorldMatrix(self):
        m = np.eye(4, dtype=np.float64)
        m[:3, :3] = self.rotation_matrix() @ np.diag(self.scale)
        m[:3, 3] = self.position
        return m

    @property
    def worldToLocalMatrix(self):
        return np.linalg.inv(self.localToWorldMatrix)

    @property
    def forward(self):
        return self.rotation_matrix()[:, 2]


class MatrixTransform:
    """A Transform given directly by its localToWorldMatrix (4x4, row/col) — e.g. the product of a
    parent chain read from a Unity scene file (unityscene.py)."""

    def __init__(self, matrix):
        self.matrix = np.asarray(matrix, dtype=np.float64).reshape(4, 4)

    @property
    def localToWorldMatrix(self):
        return self.matrix

    @property
    def worldToLocalMatrix(self):
        return np.linalg.inv(self.matrix)

    @property
    def forward(self):
        f = self.matrix[:3, 2]
        return f / (np.linalg.norm(f) or 1.0)


def matrix_to_abi(m):
    """4x4 (row, col) -> Unity Matrix4x4 memory order (column-major 16 floats)."""
    return (np.asarray(m, dtype=np.float64).T.reshape(16) + 0.0).astype(np.float32)  # + 0.0: no negative zeros


class Camera:
    """The fields of UnityEngine.Camera that RCM:183-190 reads."""

    def __init__(self, transform=None, fieldOfView=60.0, aspect=16.0 / 9.0):
        self.transform = transform or Transform()
        self.fieldOfView = float(fieldOfView)
        self.aspect = float(aspect)


class Model:
    """Types/Model.cs: a mesh + RayTracingMaterial + Transform (MDL:5,14)."""

    def __init__(self, mesh, material=None, transform=None, name=None):
        self.Mesh = mesh
        self.material = material or RayTracingMaterial()
        self.transform = transform or Transform()
        self.name = name or mesh.name
        self.logBVHStats = False


class Sphere:
    """Analytic sphere (extension S1, include/rt_abi.h RtSphere)."""

    def __init__(self, centre, radius, material=None):
        self.centre = tuple(float(v) for v in centre)
        self.radius = float(radius)
        self.material = material or RayTracingMaterial()


class RayComputeManager:
    """RCM:7-264 without the MonoBehaviour: same settings, same call sequence."""

    def __init__(self, tracer, api, width, height, camera=None, models=(), spheres=()):
        # Main settings — RCM:9-19
        self.rayTracingEnabled = True
        self.accumulate = True
        self.bvhQuality = abi.BVH_QUALITY_HIGH
        self.maxBounceCount = 4
        self.numRaysPerPixel = 1
        self.defocusStrength = 0.0
        self.divergeStrength = 0.3
        self.focusDistance = 1.0
        # Sky settings — RCM:21-27
        self.useSky = False
        self.sunFocus = 500.0
        self.sunIntensity = 10.0
        self.sunColor = (1.0, 1.0, 1.0)
        self.sunTransform = None
        # Info — RCM:36-42
        self.numAccumulatedFrames = 0
        self.renderSeed = 0
        self.screenSize = (width, height)

        self.tracer = tracer
        self.api = api
        self.camera = camera or Camera(aspect=width / height)
        self.models = list(models)
        self.spheres = list(spheres)
        self.meshInfo = None
        self.hasBVH = False
        self.bvhStats = {}
        self.bvhOnGpu = False  # CreateAllMeshData builds the BVHs with rt_build_bvh_gpu (same bytes)
        self._sized = False

    # RCM:61-67
    def OnEnable(self, renderSeed=None):
        self.hasBVH = False
        # reference: new System.Random().Next() (non-deterministic); callers pass a fixed seed
        self.renderSeed = int(np.random.SeedSequence().generate_state(1)[0] & 0x7FFFFFFF) if renderSeed is None else int(renderSeed)
        self.ResetAccumulatedRender()

    # RCM:69-76
    def ResetAccumulatedRender(self):
        self.numAccumulatedFrames = 1
        self.InitFrame()
        self.tracer.reset_accumulation()

    # RCM:84-95
    def RenderFrame(self):
        if not self.rayTracingEnabled:
            return
        self.InitFrame()
        self.tracer.render_frame()
        if self.accumulate:
            self.numAccumulatedFrames += 1

    def RenderFrames(self, n):
        """n x RenderFrame with one InitFrame (nothing changes between frames)."""
        if not self.rayTracingEnabled:
            return
        self.InitFrame()
        self.tracer.render_frames(n)
        if self.accumulate:
            self.numAccumulatedFrames += n

    # RCM:115-124
    def InitFrame(self):
        self.InitTexturesAndBuffers()
        self.InitBVH()
        self.UpdateModels()
        self.SetShaderParams()

    # RCM:126-141
    def InitTexturesAndBuffers(self):
        if not self._sized:
            self.tracer.resize(*self.screenSize)
            self._sized = True

    # RCM:143-161
    def InitBVH(self):
        if self.hasBVH:
            return
        self.hasBVH = True
        data = self.CreateAllMeshData(self.models)
        self.meshInfo = data["meshInfo"]
        self.tracer.upload_scene(self.meshInfo, data["triangles"], data["nodes"], self._pack_spheres())

    def _pack_spheres(self):
        out = np.zeros(len(self.spheres), dtype=abi.sphere_dtype)
        for i, s in enumerate(self.spheres):
            out[i]["centre"] = s.centre
            out[i]["radius"] = s.radius
            out[i]["material"] = s.material.pack()
        return out

    # RCM:206-236
    def CreateAllMeshData(self, models):
        tris, nodes = [], []
        n_tris = n_nodes = 0
        meshLookup = {}
        meshInfo = np.zeros(len(models), dtype=abi.model_dtype)
        if getattr(self, "bvhOnGpu", False) and hasattr(self.api, "build_bvh_arrays_gpu_batch") and len(models):
            # every distinct mesh in ONE call: the library writes the concatenated arrays itself (rt_build_bvh_gpu_batch)
            uniq = []
            for model in models:
                if id(model.Mesh) not in meshLookup:
                    meshLookup[id(model.Mesh)] = len(uniq)
                    uniq.append(model.Mesh)
            nd, tr, per = self.api.build_bvh_arrays_gpu_batch([(m.vertices, m.normals, m.triangles) for m in uniq], self.bvhQuality,
                                                              getattr(self, "bvhDevice", 0))
            for m, (_, _, stats) in zip(uniq, per):
                self.bvhStats[m.name] = stats
            for i, model in enumerate(models):
                meshInfo[i]["nodeOffset"], meshInfo[i]["triOffset"] = per[meshLookup[id(model.Mesh)]][:2]
                meshInfo[i]["worldToLocal"] = matrix_to_abi(model.transform.worldToLocalMatrix)
                meshInfo[i]["localToWorld"] = matrix_to_abi(model.transform.localToWorldMatrix)
                meshInfo[i]["material"] = model.material.pack()
            return {"meshInfo": meshInfo, "triangles": tr, "nodes": nd}
        for i, model in enumerate(models):
            key = id(model.Mesh)
            if key not in meshLookup:
                meshLookup[key] = (n_nodes, n_tris)
                # the tree is the reference's whichever builder makes it (byte-identical): the multi-threaded host
                # builder, or — bvhOnGpu — rt_build_bvh_gpu
                on_gpu = getattr(self, "bvhOnGpu", False) and hasattr(self.api, "build_bvh_arrays_gpu")
                build = self.api.build_bvh_arrays_gpu if on_gpu else self.api.build_bvh_arrays
                extra = (getattr(self, "bvhDevice", 0),) if on_gpu else ()
                nd, tr, stats = build(model.Mesh.vertices, model.Mesh.normals, model.Mesh.triangles, self.bvhQuality, *extra)
                self.bvhStats[model.Mesh.name] = stats
                tris.append(tr)
                nodes.append(nd)
                n_tris += len(tr)
                n_nodes += len(nd)
            meshInfo[i]["nodeOffset"], meshInfo[i]["triOffset"] = meshLookup[key]
            meshInfo[i]["worldToLocal"] = matrix_to_abi(model.transform.worldToLocalMatrix)
            meshInfo[i]["localToWorld"] = matrix_to_abi(model.transform.localToWorldMatrix)
            meshInfo[i]["material"] = model.material.pack()
        return {
            "meshInfo": meshInfo,
            "triangles": np.concatenate(tris) if tris else np.zeros(0, dtype=abi.triangle_dtype),
            "nodes": np.concatenate(nodes) if nodes else np.zeros(0, dtype=abi.node_dtype),
        }

    # RCM:192-204
    def UpdateModels(self):
        for i, model in enumerate(self.models):
            self.meshInfo[i]["worldToLocal"] = matrix_to_abi(model.transform.worldToLocalMatrix)
            self.meshInfo[i]["localToWorld"] = matrix_to_abi(model.transform.localToWorldMatrix)
            self.meshInfo[i]["material"] = model.material.pack()
        if len(self.models):
            self.tracer.update_models(self.meshInfo)

    # RCM:163-190 (SetShaderParams + UpdateCameraParams)
    def SetShaderParams(self):
        self.tracer.set_params(self.params())

    def params(self):
        p = abi.RtParams()
        p.maxBounceCount = int(self.maxBounceCount)
        p.numRaysPerPixel = int(self.numRaysPerPixel)
        p.frame = int(self.numAccumulatedFrames)
        p.renderSeed = int(self.renderSeed)
        p.useSky = 1 if self.useSky else 0
        p.accumulate = 1 if self.accumulate else 0
        p.defocusStrength = float(self.defocusStrength)
        p.divergeStrength = float(self.divergeStrength)
        p.sunFocus = float(self.sunFocus)
        p.sunIntensity = float(self.sunIntensity)
        p.sunColour[:] = [float(c) for c in self.sunColor[:3]]
        # RCM:176: sunTransform == null ? Vector3.down : -sunTransform.forward
        d = (0.0, -1.0, 0.0) if self.sunTransform is None else tuple(-self.sunTransform.forward)
        p.dirToSun[:] = [float(np.float32(c)) for c in d]
        vp = self.api.view_params(self.camera.fieldOfView, self.camera.aspect, self.focusDistance)
        p.viewParams[:] = vp
        p.camLocalToWorld[:] = matrix_to_abi(self.camera.transform.localToWorldMatrix).tolist()
        return p
