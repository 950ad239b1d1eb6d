"""The five BASELINE.json configurations as synthetic scenes.

The reference ships Unity scene files that need the Unity engine and binary mesh
blobs that are missing from the snapshot, so each configuration is rebuilt here
from procedural meshes (meshes.py) with the reference's material / manager
parameter ranges (Assets/Scenes/*.unity manager blocks: maxBounceCount,
divergeStrength 1.5 / 0.3, defocusStrength 0 / 100, focusDistance 1 / 5.3, ...).
Everything is deterministic: fixed renderSeed, seeded LCG for placement.
"""
from . import abi, meshes
from .manager import Camera, Model, RayTracingMaterial, Sphere, Transform


class SceneDescription:
    def __init__(self, name, width, height, frames, settings, camera, models=(), spheres=(), note=""):
        self.name = name
        self.width = width
        self.height = height
        self.frames = frames          # frames to accumulate for the configuration's spp
        self.settings = settings      # RayComputeManager fields
        self.camera = camera
        self.models = list(models)
        self.spheres = list(spheres)
        self.note = note

    def spp(self):
        return self.frames * self.settings.get("numRaysPerPixel", 1)

    def unique_triangles(self):
        seen, n = set(), 0
        for m in self.models:
            if id(m.Mesh) not in seen:
                seen.add(id(m.Mesh))
                n += m.Mesh.triangle_count
        return n

    def make_manager(self, tracer, api, width=None, height=None, manager_cls=None):
        from .manager import RayComputeManager
        cls = manager_cls or RayComputeManager
        w = width or self.width
        h = height or self.height
        self.camera.aspect = w / h
        mgr = cls(tracer, api, w, h, camera=self.camera, models=self.models, spheres=self.spheres)
        for k, v in self.settings.items():
            if not hasattr(mgr, k):
                raise AttributeError(k)
            setattr(mgr, k, v)
        return mgr


def _mat(**kw):
    return RayTracingMaterial(**kw)


def config1():
    """CPU reference case: 256x256, 1 spp, 4 bounces, 3 analytic spheres + 1 emissive, sky on."""
    spheres = [
        Sphere((-2.2, 1.0, 0.0), 1.0, _mat(diffuseCol=(0.9, 0.2, 0.2, 1))),
        Sphere((0.0, 1.0, 0.5), 1.0, _mat(diffuseCol=(0.2, 0.8, 0.3, 1), smoothness=0.8, specularProbability=0.5)),
        Sphere((2.2, 1.0, 0.0), 1.0, _mat(diffuseCol=(0.25, 0.35, 0.9, 1))),
        Sphere((0.0, 4.5, 1.0), 1.2, _mat(diffuseCol=(0, 0, 0, 1), emissionCol=(1.0, 0.95, 0.8, 1), emissionStrength=8.0)),
    ]
    cam = Camera(Transform(position=(0, 1, -6)), fieldOfView=60.0, aspect=1.0)
    settings = dict(maxBounceCount=4, numRaysPerPixel=1, divergeStrength=0.3, defocusStrength=0.0, focusDistance=1.0,
                    useSky=True, sunFocus=500.0, sunIntensity=10.0, accumulate=True)
    return SceneDescription("config1_spheres_cpu", 256, 256, 1, settings, cam, spheres=spheres)


def config2():
    """Headline metric case: 1920x1080, 8 spp, 8 bounces, 16 spheres + checkered ground quad, sky on."""
    rnd = meshes._lcg(2)
    spheres = []
    for i in range(16):
        gx, gz = i % 4, i // 4
        r = 0.3 + 0.6 * rnd()
        x = (gx - 1.5) * 2.2 + (rnd() - 0.5) * 0.8
        z = (gz - 1.5) * 2.2 + (rnd() - 0.5) * 0.8
        col = (0.25 + 0.7 * rnd(), 0.25 + 0.7 * rnd(), 0.25 + 0.7 * rnd(), 1.0)
        sm, sp = 0.5 + 0.5 * rnd(), 0.1 + 0.9 * rnd()
        if i in (5, 10):
            m = _mat(diffuseCol=(0, 0, 0, 1), emissionCol=col, emissionStrength=6.0)
        elif i % 3 == 0:
            m = _mat(diffuseCol=col)
        elif i % 3 == 1:
            m = _mat(diffuseCol=col, smoothness=sm, specularProbability=sp)
        else:
            m = _mat(flag=abi.MATERIAL_GLASS, ior=1.5, smoothness=1.0, specularProbability=1.0,
                     absorption=(1 - col[0], 1 - col[1], 1 - col[2], 1), absorptionMultiplier=0.6)
        spheres.append(Sphere((x, r, z), r, m))
    ground = Model(meshes.quad(), _mat(flag=abi.MATERIAL_CHECKERED, diffuseCol=(0.82, 0.82, 0.82, 1),
                                       emissionCol=(0.28, 0.28, 0.33, 1), specularProbability=0.0),
                   Transform(position=(0, 0, 0), euler=(90, 0, 0), scale=(40, 40, 1)), name="Ground")
    cam = Camera(Transform(position=(0, 2.6, -8.8), euler=(12, 0, 0)), fieldOfView=60.0, aspect=16 / 9)
    settings = dict(maxBounceCount=8, numRaysPerPixel=8, divergeStrength=1.5, defocusStrength=0.0, focusDistance=1.0,
                    useSky=True, sunFocus=500.0, sunIntensity=10.0, accumulate=True)
    return SceneDescription("config2_16spheres_quad", 1920, 1080, 1, settings, cam, models=[ground], spheres=spheres)


def _room(cube, quad_mesh, half_w=2.75, height=4.0, z_front=-7.0, z_back=5.0):
    """Cornell-style room in the manner of 'Glass Dragon.unity': walls are Unity cubes
    (checkered side walls, floor / ceiling rotated 90 deg about Z), a front quad, a
    box light under the ceiling."""
    t = 0.15
    depth = z_back - z_front
    zc = 0.5 * (z_back + z_front)
    width = 2 * half_w + t
    white = dict(diffuseCol=(0.86, 0.86, 0.86, 1), specularProbability=0.0)
    models = [
        Model(cube, _mat(**white), Transform((0, -t / 2, zc), (0, 0, 90), (t, width, depth)), "Floor"),
        Model(cube, _mat(**white), Transform((0, height + t / 2, zc), (0, 0, 90), (t, width, depth)), "Ceiling"),
        Model(cube, _mat(flag=abi.MATERIAL_CHECKERED, diffuseCol=(0.85, 0.2, 0.18, 1), emissionCol=(0.6, 0.12, 0.1, 1),
                         smoothness=0.042, specularProbability=0.06),
              Transform((-half_w, height / 2, zc), (0, 0, 0), (t, height + 2 * t, depth)), "WallLeft"),
        Model(cube, _mat(flag=abi.MATERIAL_CHECKERED, diffuseCol=(0.2, 0.75, 0.25, 1), emissionCol=(0.12, 0.5, 0.15, 1),
                         smoothness=0.015, specularProbability=0.039),
              Transform((half_w, height / 2, zc), (0, 0, 0), (t, height + 2 * t, depth)), "WallRight"),
        Model(cube, _mat(**white), Transform((0, height / 2, z_back + t / 2), (0, 0, 0), (width, height + 2 * t, t)), "WallBack"),
        Model(quad_mesh, _mat(**white), Transform((0, height / 2, z_front), (0, 180, 0), (width, height + 2 * t, 1)), "WallFront"),
        Model(cube, _mat(diffuseCol=(0, 0, 0, 1), emissionCol=(1.0, 0.90, 0.53, 1), emissionStrength=15.0),
              Transform((0, height - 0.043 - 0.04, 0.5), (0, 0, 90), (0.086, 1.6, 1.6)), "Light"),
    ]
    return models


def config3():
    """Cornell room (Unity cubes + quad) + glass rounded cube + opaque rounded cube; ~1.7k unique
    triangles, 9 models; 'two-level' = model loop over per-mesh BVHs (the reference has no TLAS).
    1920x1080, 64 spp = 8 frames x 8 spp, 8 bounces, sky off."""
    cube, quad_mesh, rc = meshes.cube(), meshes.quad(), meshes.rounded_cube(12)
    models = _room(cube, quad_mesh)
    models += [
        Model(rc, _mat(flag=abi.MATERIAL_GLASS, ior=1.5, smoothness=1.0, specularProbability=1.0,
                       absorption=(0.1, 0.35, 0.6, 1), absorptionMultiplier=0.4),
              Transform((-1.0, 0.62, 0.6), (0, 30, 0), 1.2), "GlassRoundedCube"),
        Model(rc, _mat(diffuseCol=(0.85, 0.5, 0.2, 1), smoothness=0.6, specularProbability=0.3),
              Transform((1.1, 0.52, -0.2), (0, -20, 0), 1.0), "OpaqueRoundedCube"),
    ]
    cam = Camera(Transform(position=(0, 1.9, -5.67)), fieldOfView=54.5, aspect=16 / 9)
    settings = dict(maxBounceCount=8, numRaysPerPixel=8, divergeStrength=1.5, defocusStrength=0.0, focusDistance=1.0,
                    useSky=False, accumulate=True)
    return SceneDescription("config3_cornell_1k7tris", 1920, 1080, 8, settings, cam, models=models)


def _dragon_glass():
    # 'Glass Dragon.unity' dragon material: absorption (0.91,0.79,0.25) x 1.5, smoothness 0.85,
    # specularProbability 0.888, ior 1.5
    return _mat(flag=abi.MATERIAL_GLASS, ior=1.5, smoothness=0.85, specularProbability=0.888,
                absorption=(0.91, 0.79, 0.25, 1), absorptionMultiplier=1.5)


def config4(subdivisions=6):
    """'Bunny-class' mesh: displaced icosphere-6 (81,920 triangles, seed 4) in the Cornell room,
    glass like the reference's dragon, plus an opaque rounded cube; 32 spp = 4 frames x 8,
    depth of field on (defocusStrength 100, focusDistance 5.3 as in 'Sphere Refract.unity').
    The Stanford bunny itself is not available offline."""
    cube, quad_mesh, rc = meshes.cube(), meshes.quad(), meshes.rounded_cube(12)
    blob = meshes.icosphere(subdivisions, 1.0, displacement_seed=4)
    models = _room(cube, quad_mesh)
    models += [
        Model(blob, _dragon_glass(), Transform((-0.35, 1.18, 0.1), (0, 25, 0), 1.0), "Blob80k"),
        Model(rc, _mat(diffuseCol=(0.3, 0.45, 0.85, 1), smoothness=0.3, specularProbability=0.2),
              Transform((1.55, 0.42, -0.9), (0, -35, 0), 0.8), "OpaqueRoundedCube"),
    ]
    cam = Camera(Transform(position=(0, 1.9, -5.67)), fieldOfView=54.5, aspect=16 / 9)
    settings = dict(maxBounceCount=8, numRaysPerPixel=8, divergeStrength=1.5, defocusStrength=100.0, focusDistance=5.3,
                    useSky=False, accumulate=True)
    return SceneDescription("config4_blob82k_dof", 1920, 1080, 4, settings, cam, models=models)


def config5(subdivisions=6, n_meshes=12):
    """~1M unique triangles: 12 distinct displaced icosphere-6 meshes (seeds 100..111) in a larger
    room; 3840x2160, 256 spp = 32 frames x 8, 12 bounces. Intended to be row-tiled over 8 GPUs."""
    cube, quad_mesh = meshes.cube(), meshes.quad()
    models = _room(cube, quad_mesh, half_w=4.2, height=5.0, z_front=-8.0, z_back=6.0)
    rnd = meshes._lcg(5)
    for i in range(n_meshes):
        blob = meshes.icosphere(subdivisions, 1.0, displacement_seed=100 + i)
        gx, gy = i % 4, i // 4
        pos = ((gx - 1.5) * 1.9, 0.75 + gy * 1.45, 0.4 + (gy % 2) * 0.9 + (rnd() - 0.5) * 0.4)
        kind = i % 3
        if kind == 0:
            m = _dragon_glass()
        elif kind == 1:
            m = _mat(diffuseCol=(0.3 + 0.6 * rnd(), 0.3 + 0.6 * rnd(), 0.3 + 0.6 * rnd(), 1), smoothness=0.7,
                     specularProbability=0.25)
        else:
            m = _mat(diffuseCol=(0.3 + 0.6 * rnd(), 0.3 + 0.6 * rnd(), 0.3 + 0.6 * rnd(), 1))
        models.append(Model(blob, m, Transform(pos, (0, 360 * rnd(), 0), 0.62), f"Blob{i}"))
    cam = Camera(Transform(position=(0, 2.4, -6.9)), fieldOfView=54.5, aspect=16 / 9)
    settings = dict(maxBounceCount=12, numRaysPerPixel=8, divergeStrength=1.5, defocusStrength=0.0, focusDistance=1.0,
                    useSky=False, accumulate=True)
    return SceneDescription("config5_1Mtris_multimesh", 3840, 2160, 32, settings, cam, models=models)


def glass_balls(numRaysPerPixel=8, width=1920, height=1080):
    """One of the reference's own scenes, `Assets/Scenes/Glass Balls.unity`: Cornell room with
    checkered walls, five ceiling lights and six glass balls — 17 models, every parameter as
    serialized in the scene file (scenes_data/glass_balls.json, written by unityscene.py), except
    that `Icosphere.obj` (a missing blob upstream) is a subdivision-4 icosphere and that the
    manager's 1 ray per pixel per frame is raised to the BASELINE's 8 spp per frame."""
    import os
    from . import sceneio
    sc = sceneio.load_scene(os.path.join(os.path.dirname(os.path.abspath(__file__)), "scenes_data", "glass_balls.json"))
    sc.name = "glass_balls_reference_scene"
    sc.width, sc.height = width, height
    sc.settings["numRaysPerPixel"] = numRaysPerPixel
    return sc


def reference_scene(short_name, numRaysPerPixel=None, width=None, height=None):
    """The reference's other scene files — `Glass Dragon`, `Sphere Refract`, `Splash`, `Text` (.unity) — as transcribed by
    tools/convert_reference_scenes.py into scenes_data/<short_name>.json: every transform, material, manager and camera
    setting as serialized upstream (maxBounceCount 32 in three of them, depth of field in `Sphere Refract`, Quality.Low
    BVHs in `Splash`, 18 models in `Text`); meshes that are engine resources or missing blobs are the declared stand-ins."""
    import os
    from . import sceneio
    sc = sceneio.load_scene(os.path.join(os.path.dirname(os.path.abspath(__file__)), "scenes_data", short_name + ".json"))
    sc.name = short_name + "_reference_scene"
    if width and height:
        sc.width, sc.height = width, height
    if numRaysPerPixel:
        sc.settings["numRaysPerPixel"] = numRaysPerPixel
    return sc


def _ref(short_name):
    return lambda **kw: reference_scene(short_name, **kw)


CONFIGS = {1: config1, 2: config2, 3: config3, 4: config4, 5: config5, 6: glass_balls,
           7: _ref("glass_dragon"), 8: _ref("sphere_refract"), 9: _ref("splash"), 10: _ref("text")}


def get(config_id, **kw):
    return CONFIGS[int(config_id)](**kw)
