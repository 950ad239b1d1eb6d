"""The step before the hot path: scene description files.

The reference's scenes are Unity YAML (`Assets/Scenes/*.unity`) that only the Unity editor can
read.  This module defines a small JSON format carrying exactly what `RayComputeManager`
consumes — the manager's inspector fields (RCM:9-42), the camera, and per `Model` a mesh, a
`RayTracingMaterial` (RayTracingMaterial.cs:15-27) and a transform — plus an OBJ loader for
triangle meshes such as the reference's `Assets/Graphics/cube_rounded2.obj`.

    {"name": "...", "width": 1920, "height": 1080, "frames": 8,
     "settings": {"maxBounceCount": 8, "numRaysPerPixel": 8, "useSky": false, ...},
     "camera": {"position": [0,1.9,-5.67], "euler": [0,0,0], "fieldOfView": 54.5},
     "meshes": {"room_cube": {"type": "cube"}, "blob": {"type": "icosphere", "subdivisions": 4, "displacement_seed": 4},
                "thing": {"type": "obj", "path": "thing.obj"}},
     "models": [{"mesh": "room_cube", "name": "Floor",
                 "transform": {"position": [0,-0.075,-1], "euler": [0,0,90], "scale": [0.15,5.65,12]},
                 "material": {"diffuseCol": [0.86,0.86,0.86,1], "specularProbability": 0}}],
     "spheres": [{"centre": [0,1,0], "radius": 1, "material": {"flag": 2, "ior": 1.5}}]}
"""
import json
import os

import numpy as np

from . import meshes
from .manager import Camera, MatrixTransform, Model, RayTracingMaterial, Sphere, Transform
from .scenes import SceneDescription

MATERIAL_FIELDS = ["flag", "diffuseCol", "emissionCol", "specularCol", "absorption", "absorptionMultiplier",
                   "emissionStrength", "smoothness", "specularProbability", "ior"]
SETTING_FIELDS = ["accumulate", "bvhQuality", "maxBounceCount", "numRaysPerPixel", "defocusStrength", "divergeStrength",
                  "focusDistance", "useSky", "sunFocus", "sunIntensity", "sunColor"]


def load_obj(path, unity_import=False):
    """Wavefront OBJ -> Mesh (positions + per-corner normals, polygons fan-triangulated (0,1,2),(0,2,3)...).
    unity_import=True applies what Unity's importer does (mirror X, swap winding); the default
    keeps the file verbatim — either is self-consistent with RayTriangle's winding rule as long
    as cross(B-A, C-A) points along the vertex normals (meshes.check_winding)."""
    pos, nrm, corners, faces = [], [], {}, []
    verts, norms = [], []
    with open(path) as f:
        for line in f:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                pos.append([float(p[1]), float(p[2]), float(p[3])])
            elif p[0] == "vn":
                nrm.append([float(p[1]), float(p[2]), float(p[3])])
            elif p[0] == "f":
                idx = []
                for tok in p[1:]:
                    parts = tok.split("/")
                    vi = int(parts[0])
                    ni = int(parts[2]) if len(parts) > 2 and parts[2] else 0
                    vi = vi - 1 if vi > 0 else len(pos) + vi
                    ni = (ni - 1 if ni > 0 else len(nrm) + ni) if ni else -1
                    key = (vi, ni)
                    if key not in corners:
                        corners[key] = len(verts)
                        verts.append(pos[vi])
                        norms.append(nrm[ni] if ni >= 0 else None)
                    idx.append(corners[key])
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
    v = np.asarray(verts, dtype=np.float64).reshape(-1, 3)
    f = np.asarray(faces, dtype=np.int64).reshape(-1, 3)
    n = np.zeros_like(v)
    missing = [i for i, x in enumerate(norms) if x is None]
    for i, x in enumerate(norms):
        if x is not None:
            n[i] = x
    if missing:  # no vn in the file: area-weighted vertex normals
        fn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
        acc = np.zeros_like(v)
        for k in range(3):
            np.add.at(acc, f[:, k], fn)
        ln = np.linalg.norm(acc, axis=1, keepdims=True)
        acc = acc / np.where(ln > 0, ln, 1)
        n[missing] = acc[missing]
    if unity_import:
        v[:, 0] *= -1
        n[:, 0] *= -1
        f = f[:, [0, 2, 1]]
    return meshes.Mesh(v, n, f.reshape(-1), os.path.basename(path))


def _mesh_from_spec(spec, base_dir):
    t = spec.get("type")
    if t == "cube":
        return meshes.cube()
    if t == "quad":
        return meshes.quad()
    if t == "rounded_cube":
        return meshes.rounded_cube(int(spec.get("k", 12)), float(spec.get("radius", 0.18)))
    if t == "icosphere":
        return meshes.icosphere(int(spec.get("subdivisions", 2)), float(spec.get("radius", 1.0)),
                                spec.get("displacement_seed"), float(spec.get("displacement", 0.12)))
    if t == "obj":
        return load_obj(os.path.join(base_dir, spec["path"]), bool(spec.get("unity_import", False)))
    if t == "arrays":
        return meshes.Mesh(spec["vertices"], spec["normals"], spec["triangles"], spec.get("name", "arrays"))
    raise ValueError(f"unknown mesh type {t!r}")


def _material(d):
    return RayTracingMaterial(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in (d or {}).items()})


def _transform(d):
    d = d or {}
    if "matrix" in d:  # localToWorldMatrix given directly (e.g. a Unity parent chain, unityscene.py)
        return MatrixTransform(d["matrix"])
    return Transform(d.get("position", (0, 0, 0)), d.get("euler", (0, 0, 0)), d.get("scale", (1, 1, 1)))


class _ForwardOnly:
    def __init__(self, forward):
        self.forward = np.asarray(forward, dtype=np.float64)


def scene_from_dict(d, base_dir="."):
    mesh_objs = {name: _mesh_from_spec(spec, base_dir) for name, spec in d.get("meshes", {}).items()}
    models = [Model(mesh_objs[m["mesh"]], _material(m.get("material")), _transform(m.get("transform")), m.get("name"))
              for m in d.get("models", [])]
    spheres = [Sphere(s["centre"], s["radius"], _material(s.get("material"))) for s in d.get("spheres", [])]
    cam = d.get("camera", {})
    w, h = int(d.get("width", 1920)), int(d.get("height", 1080))
    camera = Camera(_transform(cam), float(cam.get("fieldOfView", 60.0)), w / h)
    settings = {k: (tuple(v) if isinstance(v, list) else v) for k, v in d.get("settings", {}).items()}
    for k in settings:
        if k not in SETTING_FIELDS:
            raise ValueError(f"unknown manager setting {k!r}")
    if "sunForward" in d:  # RCM:176 reads sunTransform.forward only
        settings["sunTransform"] = _ForwardOnly(d["sunForward"])
    return SceneDescription(d.get("name", "scene"), w, h, int(d.get("frames", 1)), settings, camera, models, spheres)


def load_scene(path, **unity_kw):
    """A JSON scene, or a Unity `.unity` scene file (converted by unityscene.py; unity_kw =
    assets_dir / stand_ins / width / height / frames)."""
    if path.lower().endswith(".unity"):
        from . import unityscene
        d, _notes = unityscene.load_unity_scene(path, **unity_kw)
        return scene_from_dict(d, os.path.dirname(os.path.abspath(path)))
    with open(path) as f:
        return scene_from_dict(json.load(f), os.path.dirname(os.path.abspath(path)))


def scene_to_dict(scene, mesh_specs=None):
    """Inverse of scene_from_dict.  Meshes are written as `arrays` unless `mesh_specs` maps a mesh
    name to a procedural spec."""
    mesh_specs = mesh_specs or {}
    out_meshes, names = {}, {}
    for m in scene.models:
        if id(m.Mesh) not in names:
            name = m.Mesh.name
            while name in out_meshes:
                name += "_"
            names[id(m.Mesh)] = name
            out_meshes[name] = mesh_specs.get(m.Mesh.name) or {
                "type": "arrays", "vertices": m.Mesh.vertices.tolist(), "normals": m.Mesh.normals.tolist(),
                "triangles": m.Mesh.triangles.tolist()}

    def mat(x):
        return {k: (list(getattr(x, k)) if isinstance(getattr(x, k), tuple) else getattr(x, k)) for k in MATERIAL_FIELDS}

    def tf(t):
        if isinstance(t, MatrixTransform):
            return {"matrix": t.matrix.tolist()}
        return {"position": list(t.position), "euler": list(t.euler), "scale": list(t.scale)}
    cam = tf(scene.camera.transform)
    cam["fieldOfView"] = scene.camera.fieldOfView
    return {
        "name": scene.name, "width": scene.width, "height": scene.height, "frames": scene.frames,
        "settings": {k: (list(v) if isinstance(v, tuple) else v) for k, v in scene.settings.items() if k != "sunTransform"},
        **({"sunForward": [float(x) for x in scene.settings["sunTransform"].forward]} if scene.settings.get("sunTransform") is not None else {}),
        "camera": cam, "meshes": out_meshes,
        "models": [{"mesh": names[id(m.Mesh)], "name": m.name, "transform": tf(m.transform), "material": mat(m.material)}
                   for m in scene.models],
        "spheres": [{"centre": list(s.centre), "radius": s.radius, "material": mat(s.material)} for s in scene.spheres],
    }


def save_scene(path, scene, mesh_specs=None):
    with open(path, "w") as f:
        json.dump(scene_to_dict(scene, mesh_specs), f)
