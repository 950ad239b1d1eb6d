"""Unity scene files (`Assets/Scenes/*.unity`, Unity's YAML dialect) -> the JSON scene format of
`sceneio` — the on-disk side of the boundary for the reference's own scenes.

What `RayComputeManager` would see after Unity loaded the scene:

    Model[]   = every enabled `Model` component (Types/Model.cs) on an active GameObject
                (RCM:118 `FindObjectsByType<Model>(FindObjectsInactive.Exclude, ...)`), with
                  material   <- the component's serialized `material` struct (RayTracingMaterial.cs:15-27)
                  mesh       <- the MeshFilter it references (`m_Mesh` {fileID, guid})
                  transform  <- localToWorldMatrix of its Transform: the chain of parent
                                TRS (position, quaternion, scale) up to the scene root
    manager   = the `RayComputeManager` component's inspector fields (RCM:9-42)
    camera    = the Camera's world matrix and vertical field of view (RCM:183-190)
    sun       = `sunTransform`'s world forward axis (RCM:176)

Not reproducible from the file: the order of `models` (sorted by runtime InstanceID, RCM:118 —
file order is used; it only decides exact closest-hit ties, quirk Q12) and engine-side meshes.
Unity's built-in Cube / Quad are rebuilt exactly (meshes.py); assets are resolved through the
`.meta` files of an assets directory — an `.obj` that exists is loaded with Unity's import
convention, anything else (FBX, missing files, the built-in Sphere) needs a stand-in given in
`stand_ins`, and is marked as such in the output.
"""
import math
import os
import re

import numpy as np

UNITY_BUILTIN_GUID = "0000000000000000e000000000000000"
BUILTIN_MESHES = {10202: ("Cube", {"type": "cube"}), 10210: ("Quad", {"type": "quad"})}
# built-in meshes whose vertex data lives inside the engine: only available as stand-ins
BUILTIN_NAMES = {10206: "Cylinder", 10207: "Sphere", 10208: "Capsule", 10209: "Plane"}
MATERIAL_COLOURS = ["diffuseCol", "emissionCol", "specularCol", "absorption"]
MATERIAL_SCALARS = ["absorptionMultiplier", "emissionStrength", "smoothness", "specularProbability", "ior", "flag"]
MANAGER_FIELDS = ["accumulate", "bvhQuality", "maxBounceCount", "numRaysPerPixel", "defocusStrength", "divergeStrength",
                  "focusDistance", "useSky", "sunFocus", "sunIntensity"]

_HEADER = re.compile(r"^--- !u!(\d+) &(-?\d+)(\s+stripped)?\s*$")


def parse_documents(text):
    """{fileID: (classID, type name, fields)} of every object in a Unity YAML file."""
    import yaml
    docs, cur, body = {}, None, []

    def flush():
        if cur is None:
            return
        data = yaml.safe_load("\n".join(body)) or {}
        if isinstance(data, dict) and len(data) == 1:
            (name, fields), = data.items()
            docs[cur[1]] = (cur[0], name, fields or {})

    for line in text.splitlines():
        m = _HEADER.match(line)
        if m:
            flush()
            cur, body = (int(m.group(1)), int(m.group(2))), []
        elif line.startswith("%"):
            continue
        elif cur is not None:
            body.append(line)
    flush()
    return docs


def _v3(d, default):
    return np.array([float(d.get(k, v)) for k, v in zip("xyz", default)], dtype=np.float64) if d else np.array(default, dtype=np.float64)


def quaternion_matrix(q):
    """Rotation matrix of a (normalised) quaternion x, y, z, w — UnityEngine.Quaternion's convention."""
    x, y, z, w = q
    n = math.sqrt(x * x + y * y + z * z + w * w) or 1.0
    x, y, z, w = x / n, y / n, z / n, w / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float64)


def local_matrix(tf):
    """Matrix4x4.TRS(m_LocalPosition, m_LocalRotation, m_LocalScale)."""
    r = tf.get("m_LocalRotation") or {}
    q = [float(r.get(k, d)) for k, d in zip("xyzw", (0, 0, 0, 1))]
    m = np.eye(4)
    m[:3, :3] = quaternion_matrix(q) @ np.diag(_v3(tf.get("m_LocalScale"), (1, 1, 1)))
    m[:3, 3] = _v3(tf.get("m_LocalPosition"), (0, 0, 0))
    return m


class UnityScene:
    def __init__(self, text):
        self.docs = parse_documents(text)
        self.transform_of_go = {}   # GameObject fileID -> Transform fileID
        self.components = {}        # GameObject fileID -> [component fileIDs]
        for fid, (cls, name, f) in self.docs.items():
            go = (f.get("m_GameObject") or {}).get("fileID") if isinstance(f, dict) else None
            if go:
                self.components.setdefault(go, []).append(fid)
                if name in ("Transform", "RectTransform"):
                    self.transform_of_go[go] = fid
        self._world = {}

    def fields(self, fid):
        return self.docs[fid][2]

    def world_matrix(self, transform_id):
        """localToWorldMatrix: parent chain of TRS matrices, root first."""
        if transform_id not in self._world:
            tf = self.fields(transform_id)
            m = local_matrix(tf)
            father = (tf.get("m_Father") or {}).get("fileID", 0)
            if father and father in self.docs:
                m = self.world_matrix(father) @ m
            self._world[transform_id] = m
        return self._world[transform_id]

    def active(self, go_id):
        """activeInHierarchy: the GameObject and all its ancestors have m_IsActive."""
        go = self.fields(go_id)
        if not int(go.get("m_IsActive", 1)):
            return False
        tf = self.transform_of_go.get(go_id)
        father = (self.fields(tf).get("m_Father") or {}).get("fileID", 0) if tf else 0
        if father and father in self.docs:
            return self.active((self.fields(father).get("m_GameObject") or {}).get("fileID"))
        return True

    def behaviours(self, *required):
        """MonoBehaviours that carry all the given serialized fields, in file order."""
        return [fid for fid, (cls, name, f) in self.docs.items() if name == "MonoBehaviour" and all(k in f for k in required)]


def scan_asset_guids(assets_dir):
    """guid -> asset path, from the .meta files under a Unity Assets directory."""
    out = {}
    for root, _, files in os.walk(assets_dir):
        for fn in files:
            if fn.endswith(".meta"):
                try:
                    with open(os.path.join(root, fn)) as f:
                        m = re.search(r"^guid:\s*([0-9a-f]{32})", f.read(), re.M)
                except OSError:
                    continue
                if m:
                    out[m.group(1)] = os.path.join(root, fn[:-5])
    return out


def _colour(d):
    d = d or {}
    return [float(d.get(k, 0.0)) for k in "rgba"]


def convert(text, assets_dir=None, stand_ins=None, width=None, height=None, frames=1, name="unity_scene"):
    """Unity scene text -> (scene dict for sceneio.scene_from_dict, list of notes).

    stand_ins: {mesh name: mesh spec}, mesh name = the asset's file name ("Icosphere.obj") or
    "builtin:Sphere"; used when the real mesh cannot be had."""
    sc = UnityScene(text)
    notes = []
    stand_ins = dict(stand_ins or {})
    guid_paths = scan_asset_guids(assets_dir) if assets_dir else {}
    meshes, models = {}, []

    def mesh_key(ref):
        fid, guid = int(ref.get("fileID", 0)), str(ref.get("guid", ""))
        if guid == UNITY_BUILTIN_GUID:
            if fid in BUILTIN_MESHES:
                key, spec = BUILTIN_MESHES[fid]
                meshes.setdefault(key, dict(spec))
                return key
            label = "builtin:" + BUILTIN_NAMES.get(fid, str(fid))
        else:
            path = guid_paths.get(guid)
            label = os.path.basename(path) if path else "guid:" + guid
            if path and path.lower().endswith(".obj") and os.path.exists(path) and label not in stand_ins:
                meshes.setdefault(label, {"type": "obj", "path": os.path.abspath(path), "unity_import": True})
                return label
        if label not in stand_ins:
            raise KeyError(f"mesh {label!r} (fileID {fid}, guid {guid}) is not available: pass a stand-in for it")
        if label not in meshes:
            meshes[label] = dict(stand_ins[label], stand_in=True)
            notes.append(f"mesh {label}: stand-in {stand_ins[label]}")
        return label

    for fid in sc.behaviours("material", "meshFilter"):            # Types/Model.cs
        f = sc.fields(fid)
        go = (f.get("m_GameObject") or {}).get("fileID")
        if not int(f.get("m_Enabled", 1)) or not go or not sc.active(go):
            continue
        mf = (f.get("meshFilter") or {}).get("fileID", 0)
        if not mf or mf not in sc.docs:                            # OnValidate would fill it from the same GameObject
            mf = next((c for c in sc.components.get(go, []) if sc.docs[c][1] == "MeshFilter"), 0)
        if not mf:
            notes.append(f"model on GameObject {go}: no MeshFilter, skipped")
            continue
        key = mesh_key(sc.fields(mf).get("m_Mesh") or {})
        mat = f.get("material") or {}
        material = {k: _colour(mat.get(k)) for k in MATERIAL_COLOURS}
        material.update({k: (int(mat.get(k, 0)) if k == "flag" else float(mat.get(k, 0.0))) for k in MATERIAL_SCALARS})
        models.append({"mesh": key, "name": str(sc.fields(go).get("m_Name", "")), "material": material,
                       "transform": {"matrix": sc.world_matrix(sc.transform_of_go[go]).tolist()}})

    out = {"name": name, "frames": int(frames), "meshes": meshes, "models": models, "spheres": []}
    managers = sc.behaviours("maxBounceCount", "numRaysPerPixel")   # Tracer/RayComputeManager.cs
    settings = {}
    if managers:
        f = sc.fields(managers[0])
        for k in MANAGER_FIELDS:
            if k in f:
                v = f[k]
                settings[k] = bool(int(v)) if k in ("accumulate", "useSky") else (int(v) if k in ("bvhQuality", "maxBounceCount", "numRaysPerPixel") else float(v))
        if "sunColor" in f:
            settings["sunColor"] = _colour(f["sunColor"])[:3]
        sun = (f.get("sunTransform") or {}).get("fileID", 0)
        if sun and sun in sc.docs:
            fw = sc.world_matrix(sun)[:3, 2]
            out["sunForward"] = (fw / (np.linalg.norm(fw) or 1.0)).tolist()
        size = f.get("screenSize") or {}
        width = width or int(size.get("x", 0)) or None
        height = height or int(size.get("y", 0)) or None
    else:
        notes.append("no RayComputeManager component found: default settings")
    out["settings"] = settings
    out["width"], out["height"] = int(width or 1920), int(height or 1080)

    cams = [fid for fid, (cls, nm, f) in sc.docs.items() if nm == "Camera" and int(f.get("m_Enabled", 1))
            and sc.active((f.get("m_GameObject") or {}).get("fileID"))]
    if cams:
        f = sc.fields(cams[0])
        go = (f.get("m_GameObject") or {}).get("fileID")
        out["camera"] = {"matrix": sc.world_matrix(sc.transform_of_go[go]).tolist(), "fieldOfView": float(f.get("field of view", 60.0))}
    else:
        notes.append("no active Camera found: default camera")
    return out, notes


def load_unity_scene(path, assets_dir=None, stand_ins=None, **kw):
    with open(path, encoding="utf-8") as f:
        text = f.read()
    if assets_dir is None:  # <project>/Assets/Scenes/x.unity -> <project>/Assets
        guess = os.path.dirname(os.path.dirname(os.path.abspath(path)))
        assets_dir = guess if os.path.basename(guess) == "Assets" else None
    kw.setdefault("name", os.path.splitext(os.path.basename(path))[0])
    return convert(text, assets_dir, stand_ins, **kw)
