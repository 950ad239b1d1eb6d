"""Known-answer tests that pin the CPU oracle (the reference has no tests or golden
vectors of its own — SURVEY.md §8(c) — so these are hand-derived from the shader
source: RayCommon.hlsl "RC").  Each expected value is computed here independently
of the oracle (pure Python / closed form)."""
import ctypes as C
import math

import numpy as np
import pytest


def f3(*v):
    return (C.c_float * len(v))(*v)


# ---------------------------------------------------------------- RNG (RC:127-138)
def pcg_py(state):
    state = (state * 747796405 + 2891336453) & 0xFFFFFFFF
    result = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
    result = ((result >> 22) ^ result) & 0xFFFFFFFF
    return state, result


@pytest.mark.parametrize("seed", [0, 1, 12345, 719393 + 1, 0xFFFFFFFF, 0x80000000])
def test_pcg_matches_independent_python(orc, seed):
    st = C.c_uint32(seed)
    s = seed
    for _ in range(50):
        s, want = pcg_py(s)
        got = orc.next_random(C.byref(st))
        assert got == want and st.value == s


def test_random_value_is_u32_over_2pow32_in_fp32(orc):
    st = C.c_uint32(7)
    s = 7
    for _ in range(200):
        s, r = pcg_py(s)
        want = np.float32(np.float32(r) / np.float32(4294967296.0))
        assert orc.random_value(C.byref(st)) == want
    # float(0xFFFFFFFF) rounds to 2^32 -> exactly 1.0 is reachable (quirk Q4)
    assert np.float32(np.float32(0xFFFFFFFF) / np.float32(4294967296.0)) == 1.0


def test_random_direction_is_unit_and_consumes_six_draws(orc):
    st = C.c_uint32(99)
    out = f3(0, 0, 0)
    orc.random_direction(C.byref(st), out)
    assert abs(math.sqrt(sum(v * v for v in out)) - 1) < 1e-6
    s = 99
    for _ in range(6):
        s, _r = pcg_py(s)
    assert st.value == s


def test_random_point_in_circle_uses_pi_3_1415(orc):
    st = C.c_uint32(5)
    out = f3(0, 0)
    orc.random_point_in_circle(C.byref(st), out)
    s, r1 = pcg_py(5)
    s, r2 = pcg_py(s)
    u1, u2 = r1 / 4294967296.0, r2 / 4294967296.0
    ang = u1 * 2 * 3.1415  # RC:2 PI = 3.1415 (quirk Q3), NOT math.pi
    assert abs(out[0] - math.cos(ang) * math.sqrt(u2)) < 2e-6
    assert abs(out[1] - math.sin(ang) * math.sqrt(u2)) < 2e-6
    assert st.value == s


# ---------------------------------------------------------------- ray / box (RC:219-231)
def test_ray_box_cases(orc):
    bmin, bmax = f3(-1, -1, -1), f3(1, 1, 1)
    # from outside along +z: enters at z=-1 -> t=4
    assert orc.ray_box(f3(0, 0, -5), f3(0, 0, 1), bmin, bmax) == 4.0
    # origin inside: shader returns 0 (BVH.cs's CPU version returns tmax — not the oracle)
    assert orc.ray_box(f3(0, 0, 0), f3(0, 0, 1), bmin, bmax) == 0.0
    # box behind the ray
    assert orc.ray_box(f3(0, 0, 5), f3(0, 0, 1), bmin, bmax) == np.inf
    # miss sideways
    assert orc.ray_box(f3(3, 0, -5), f3(0, 0, 1), bmin, bmax) == np.inf
    # axis-parallel ray (invDir = +inf on x,y) strictly inside the slabs: (-inf, +inf) -> hit at t=4
    assert orc.ray_box(f3(0.5, 0, -5), f3(0, 0, 1), bmin, bmax) == 4.0
    # ... grazing exactly ON a slab plane: 0*inf = NaN; HLSL min/max return the non-NaN operand, so
    # x gives t1 = t2 = -inf (on the max plane) or +inf (on the min plane) -> tFar < tNear -> miss
    assert orc.ray_box(f3(1, 0, -5), f3(0, 0, 1), bmin, bmax) == np.inf
    assert orc.ray_box(f3(-1, 0, -5), f3(0, 0, 1), bmin, bmax) == np.inf
    # diagonal
    d = 1 / math.sqrt(3)
    t = orc.ray_box(f3(-3, -3, -3), f3(d, d, d), bmin, bmax)
    assert abs(t - 2 * math.sqrt(3)) < 1e-5


# ---------------------------------------------------------------- ray / triangle (RC:188-215)
def make_tri(pkg, a, b, c, n=(0, 0, -1)):
    t = np.zeros((), dtype=pkg.abi.triangle_dtype)
    t["posA"], t["posB"], t["posC"] = a, b, c
    t["normA"] = t["normB"] = t["normC"] = n
    return t


def ray_tri(orc, pos, d, tri, cull):
    out = f3(*([0] * 6))
    orc.ray_triangle(f3(*pos), f3(*d), tri.ctypes.data, int(cull), out)
    return list(out)


def test_ray_triangle_front_back_and_edges(orc, pkg):
    # cross(B-A, C-A) = (0,0,-1): front face looks toward -z
    tri = make_tri(pkg, (0, 0, 0), (0, 1, 0), (1, 0, 0))
    hit, back, dst, nx, ny, nz = ray_tri(orc, (0.25, 0.25, -2), (0, 0, 1), tri, True)
    assert hit == 1 and back == 0 and dst == 2.0 and (nx, ny, nz) == (0, 0, -1)
    # from behind: culled for opaque...
    hit, back, dst, *_ = ray_tri(orc, (0.25, 0.25, 2), (0, 0, -1), tri, True)
    assert hit == 0 and back == 1
    # ...but hit for glass (cull off), normal flipped by sign(det)
    hit, back, dst, nx, ny, nz = ray_tri(orc, (0.25, 0.25, 2), (0, 0, -1), tri, False)
    assert hit == 1 and back == 1 and dst == 2.0 and nz == 1.0
    # exactly on an edge (u = 0) and on a vertex count as hits (>= 0, RC:207)
    assert ray_tri(orc, (0.0, 0.5, -1), (0, 0, 1), tri, True)[0] == 1
    assert ray_tri(orc, (0.0, 0.0, -1), (0, 0, 1), tri, True)[0] == 1
    # just outside
    assert ray_tri(orc, (-1e-3, 0.5, -1), (0, 0, 1), tri, True)[0] == 0
    assert ray_tri(orc, (0.6, 0.6, -1), (0, 0, 1), tri, True)[0] == 0
    # dst must be > 0: origin on the plane is a miss
    assert ray_tri(orc, (0.25, 0.25, 0), (0, 0, 1), tri, True)[0] == 0
    # parallel ray: |det| < 1e-8
    assert ray_tri(orc, (0.25, 0.25, -1), (1, 0, 0), tri, False)[0] == 0


def test_ray_triangle_interpolates_normals(orc, pkg):
    t = make_tri(pkg, (0, 0, 0), (0, 1, 0), (1, 0, 0))
    t["normA"], t["normB"], t["normC"] = (0, 0, -1), (0, 1, -1), (1, 0, -1)
    _, _, _, nx, ny, nz = ray_tri(orc, (0.25, 0.5, -1), (0, 0, 1), t, True)
    # w=0.25 (A), u=0.5 (B), v=0.25 (C): n ~ (0.25, 0.5, -1) normalised
    want = np.array([0.25, 0.5, -1.0]) / np.linalg.norm([0.25, 0.5, -1.0])
    assert np.allclose([nx, ny, nz], want, atol=1e-6)


# ---------------------------------------------------------------- ray / sphere (RC:289-332)
def ray_sphere(orc, pos, d, c, r):
    out = f3(*([0] * 6))
    orc.ray_sphere(f3(*pos), f3(*d), f3(*c), r, out)
    return list(out)


def test_ray_sphere_outside_inside_behind(orc):
    hit, back, dst, nx, ny, nz = ray_sphere(orc, (0, 0, -5), (0, 0, 1), (0, 0, 0), 1.0)
    assert hit == 1 and back == 0 and dst == 4.0 and (nx, ny, nz) == (0, 0, -1)
    # inside: far root, normal flipped, flagged backface (RC:315-320)
    hit, back, dst, nx, ny, nz = ray_sphere(orc, (0, 0, 0), (0, 0, 1), (0, 0, 0), 1.0)
    assert hit == 1 and back == 1 and dst == 1.0 and (nx, ny, nz) == (0, 0, -1)
    # sphere behind the ray
    assert ray_sphere(orc, (0, 0, 5), (0, 0, 1), (0, 0, 0), 1.0)[0] == 0
    # miss
    assert ray_sphere(orc, (2, 0, -5), (0, 0, 1), (0, 0, 0), 1.0)[0] == 0
    # tangent (discriminant == 0) is a hit
    assert ray_sphere(orc, (1, 0, -5), (0, 0, 1), (0, 0, 0), 1.0)[0] == 1
    # direction need not be unit: a = dot(d,d)
    hit, _, dst, *_ = ray_sphere(orc, (0, 0, -5), (0, 0, 2), (0, 0, 0), 1.0)
    assert hit == 1 and dst == 2.0


# ---------------------------------------------------------------- optics (RC:383-417)
def test_fresnel_normal_incidence_and_tir(orc):
    r = orc.reflectance(f3(0, 0, 1), f3(0, 0, -1), 1.0, 1.5)
    assert abs(r - ((1.0 - 1.5) / (1.0 + 1.5)) ** 2) < 1e-7
    # grazing from the dense side beyond the critical angle -> total internal reflection
    s = math.sin(math.radians(60))
    assert orc.reflectance(f3(s, 0, math.cos(math.radians(60))), f3(0, 0, -1), 1.5, 1.0) == 1.0
    # RC:391-392 quirk: both denominators are the SAME expression; at 45 deg, n=1 -> 1.5:
    ci = math.cos(math.radians(45))
    sr2 = (1 / 1.5) ** 2 * (1 - ci * ci)
    cr = math.sqrt(1 - sr2)
    den = 1.0 * ci + 1.5 * cr
    want = (((1.0 * ci - 1.5 * cr) / den) ** 2 + ((1.5 * ci - 1.0 * cr) / den) ** 2) / 2
    got = orc.reflectance(f3(math.sin(math.radians(45)), 0, ci), f3(0, 0, -1), 1.0, 1.5)
    assert abs(got - want) < 1e-6


def test_snell_refraction(orc):
    out = f3(0, 0, 0)
    th = math.radians(30)
    orc.refract(f3(math.sin(th), 0, math.cos(th)), f3(0, 0, -1), 1.0, 1.5, out)
    # sin(theta_t) = sin(theta_i)/1.5
    assert abs(out[0] - math.sin(th) / 1.5) < 1e-6 and abs(out[2] - math.sqrt(1 - (math.sin(th) / 1.5) ** 2)) < 1e-6
    # TIR returns the zero vector (RC:413)
    th = math.radians(80)
    orc.refract(f3(math.sin(th), 0, math.cos(th)), f3(0, 0, -1), 1.5, 1.0, out)
    assert list(out) == [0, 0, 0]


# ---------------------------------------------------------------- sky + checker (RC:167-183, 450-466)
def test_environment_light(orc, pkg):
    p = pkg.abi.RtParams()
    p.useSky, p.sunFocus, p.sunIntensity = 1, 500.0, 10.0
    p.sunColour[:] = [1, 1, 1]
    p.dirToSun[:] = [0, 1, 0]
    out = f3(0, 0, 0)
    orc.environment_light(C.byref(p), f3(0, -1, 0), out)  # straight down: ground colour
    assert np.allclose(list(out), [0.35, 0.3, 0.35], atol=1e-6)
    orc.environment_light(C.byref(p), f3(0, 1, 0), out)   # zenith + full sun: pow(1, 2) * 10
    assert np.allclose(list(out), [0.08 + 10, 0.37 + 10, 0.73 + 10], atol=1e-4)
    orc.environment_light(C.byref(p), f3(1, 0, 0), out)   # horizon: t=0 -> horizon white, sun pow(0,2)=0
    assert np.allclose(list(out), [1, 1, 1], atol=1e-6)
    p.useSky = 0
    orc.environment_light(C.byref(p), f3(0, 1, 0), out)
    assert list(out) == [0, 0, 0]


def test_checker_material_colour(orc, pkg):
    m = pkg.RayTracingMaterial(flag=pkg.abi.MATERIAL_CHECKERED, diffuseCol=(1, 0, 0, 1), emissionCol=(0, 0, 1, 1),
                               specularCol=(0, 1, 0, 1)).pack()
    out = f3(0, 0, 0)

    def col(pos, n, spec=0):
        orc.material_colour(m.ctypes.data, f3(*pos), f3(*n), spec, out)
        return list(out)
    # floor (normal +y): cells of 1/1.5 in xz; floor(1.5*x) parity
    assert col((0.1, 0, 0.1), (0, 1, 0)) == [1, 0, 0]        # (0,0) same parity -> diffuse
    assert col((0.7, 0, 0.1), (0, 1, 0)) == [0, 0, 1]        # (1,0) differ -> emissionCol is the alt colour
    assert col((0.7, 0, 0.7), (0, 1, 0)) == [1, 0, 0]
    assert col((-0.1, 0, 0.1), (0, 1, 0)) == [0, 0, 1]       # floor(-0.15) = -1 -> mod2 -> 1
    # wall with dominant x normal uses (z, y); dominant z uses (x, y)
    assert col((5, 0.7, 0.1), (1, 0, 0)) == [0, 0, 1]
    assert col((0.7, 0.1, 5), (0, 0, 1)) == [0, 0, 1]
    # specular bounce -> specularCol
    assert col((0.1, 0, 0.1), (0, 1, 0), 1) == [0, 1, 0]


def test_camera_view_params(orc, api):
    for lib in (orc, api):
        w, h, d = lib.view_params(60.0, 2.0, 3.0)
        assert abs(h - 3.0 * math.tan(math.radians(30)) * 2) < 1e-5 and abs(w - 2 * h) < 1e-5 and d == 3.0
    assert orc.view_params(54.5, 16 / 9, 5.3) == api.view_params(54.5, 16 / 9, 5.3)
