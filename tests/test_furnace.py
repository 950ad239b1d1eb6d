"""Furnace tests: an integrator-level pin of Trace (RayCommon.hlsl:479-542) that does not pass
through anyone's restatement of it.

A closed, sky-less enclosure whose every surface has albedo a and emits e has a known radiance field: every path segment
hits a surface; a hit adds e x throughput (RC:529) BEFORE the throughput is multiplied by the albedo (RC:531-532); the
roulette (RC:535-538) continues with probability p = max(throughput) = a and divides by p, so the throughput is back at 1
at every hit and a path that makes N hits returns exactly e*N.  N is a geometric variable truncated by the INCLUSIVE bounce
loop (RC:485: i <= MaxBounceCount, i.e. at most M = MaxBounceCount + 1 hits):

    P(N >= k) = a^(k-1), k = 1..M        E[L] = e (1 - a^M) / (1 - a)   (= e/(1-a) truncated at M terms)

Checked on the oracle (CPU) and on the HIP kernel (GPU), for an analytic sphere seen from inside (RaySphere RC:289-332:
inside hit = far root, flipped normal), an inward-facing triangle mesh with a real BVH, and a room of inward-facing quads:
  * white furnace, a = 1: EVERY pixel equals e*M exactly — no leak, no lost or extra bounce, emission order;
  * a < 1, 1 sample per pixel: every pixel is an integer multiple k*e, k in 1..M, and the histogram of k is the truncated
    geometric distribution (chi-square) — the roulette's survival probability and its 1/p compensation;
  * a < 1, many samples: the image mean is within 4 standard errors of e (1 - a^M)/(1 - a); coloured albedo (p = max channel)
    per channel e (1 - a_c^M)/(1 - a_c); mirror-like walls (the specular branch of RC:521-533) give the same;
  * HIP == oracle bit for bit on all of them (GPU test).
"""
import numpy as np
import pytest


def inward(mesh, pkg):
    """The same surface seen from inside: winding and normals reversed (opaque materials cull back faces, RC:355)."""
    idx = mesh.triangles.reshape(-1, 3)[:, ::-1].copy()
    return pkg.meshes.Mesh(mesh.vertices, -mesh.normals, idx.reshape(-1), mesh.name + "_inward")


def furnace(pkg, kind, albedo, e, max_bounce, spp, size=48, specular=False):
    M, T = pkg.RayTracingMaterial, pkg.Transform
    a3 = (albedo,) * 3 if np.isscalar(albedo) else tuple(albedo)
    if specular:   # every hit takes the specular branch: colour = specularCol, direction = mirror (smoothness 1)
        mat = M(diffuseCol=(0.1, 0.9, 0.3, 1), specularCol=a3 + (1,), specularProbability=1.0, smoothness=1.0,
                emissionCol=(1, 1, 1, 1), emissionStrength=e)
    else:          # specularProbability 0: the diffuse branch (isSpecular only if the draw is exactly 0)
        mat = M(diffuseCol=a3 + (1,), specularCol=(0.3, 0.2, 0.9, 1), specularProbability=0.0, smoothness=0.7,
                emissionCol=(1, 1, 1, 1), emissionStrength=e)
    models, spheres = [], []
    if kind == "sphere":
        spheres = [pkg.Sphere((0.3, 1.0, 0.5), 3.0, mat)]
    elif kind == "mesh":      # 1,280 inward-facing triangles, BVH with inner nodes
        models = [pkg.Model(inward(pkg.meshes.icosphere(3), pkg), mat, T((0.3, 1.0, 0.5), (10, 20, 30), 3.0))]
    elif kind == "room":      # six inward-facing quads = six models with leaf roots (+ the sphere list stays empty)
        q = pkg.meshes.quad()  # normal (0,0,-1): a quad at +z looking back at the origin is already inward
        s = 4.0
        for pos, rot in (((0, 0, 2), (0, 0, 0)), ((0, 0, -2), (0, 180, 0)), ((2, 0, 0), (0, 90, 0)), ((-2, 0, 0), (0, -90, 0)),
                         ((0, 2, 0), (-90, 0, 0)), ((0, -2, 0), (90, 0, 0))):
            models.append(pkg.Model(q, mat, T(pos, rot, (s, s, 1.0))))
    else:
        raise ValueError(kind)
    cam = pkg.Camera(T((0.2, 0.1 if kind == "room" else 0.8, -0.3), (5, 15, 0)), fieldOfView=70.0)
    settings = dict(maxBounceCount=max_bounce, numRaysPerPixel=spp, divergeStrength=1.0, defocusStrength=0.0, focusDistance=1.0,
                    useSky=False, accumulate=True, bvhQuality=1)
    return pkg.scenes.SceneDescription(f"furnace_{kind}", size, size, 1, settings, cam, models, spheres)


def radiance(lib, tr, scene, frames=1, seed=7):
    mgr = scene.make_manager(tr, lib)
    mgr.OnEnable(renderSeed=seed)
    mgr.RenderFrames(frames)
    acc = tr.read_accumulated()
    assert np.all(acc[..., 3] == frames)
    return acc, acc[..., :3].astype(np.float64) / frames


def expected(a, e, M):
    a = np.asarray(a, dtype=np.float64)
    return e * np.where(a == 1.0, float(M), (1.0 - a ** M) / np.where(a == 1.0, 1.0, 1.0 - a))


def hits_variance(a, M):
    """Variance of the truncated geometric hit count N: P(N >= k) = a^(k-1), k = 1..M."""
    k = np.arange(1, M + 1, dtype=np.float64)
    tail = a ** (k - 1)
    m1 = tail.sum()
    m2 = ((2 * k - 1) * tail).sum()
    return m2 - m1 * m1


def check_white(lib, tr, pkg, kind, specular=False):
    e, mb = 0.25, 6
    sc = furnace(pkg, kind, 1.0, e, mb, spp=3, size=40, specular=specular)
    _, img = radiance(lib, tr, sc, frames=2)
    assert np.all(img == e * (mb + 1)), (kind, np.unique(img)[:8])   # exact: 7 additions of 0.25 x 1


def check_histogram(lib, tr, pkg, kind):
    a, e, mb = 0.5, 0.5, 5
    M = mb + 1
    sc = furnace(pkg, kind, a, e, mb, spp=1, size=96)
    _, img = radiance(lib, tr, sc)
    k = img[..., 0] / e
    assert np.all(img[..., 0] == img[..., 1]) and np.all(img[..., 1] == img[..., 2])
    assert np.all(k == np.round(k)) and k.min() >= 1 and k.max() <= M       # e*N exactly: throughput is 1 at every hit
    n = k.size
    obs = np.array([(k == j).sum() for j in range(1, M + 1)], dtype=np.float64)
    p = np.array([a ** (j - 1) * (1 - a) for j in range(1, M)] + [a ** (M - 1)])
    chi2 = float(((obs - n * p) ** 2 / (n * p)).sum())
    assert chi2 < 27.9, (kind, chi2, obs.tolist(), (n * p).tolist())       # 5 d.o.f., p = 4e-5: a wrong survival probability gives thousands
    return k


def check_mean(lib, tr, pkg, kind, albedo, specular=False):
    e, mb, spp, frames, size = 0.75, 8, 8, 2, 40
    M = mb + 1
    sc = furnace(pkg, kind, albedo, e, mb, spp, size, specular)
    _, img = radiance(lib, tr, sc, frames)
    paths = size * size * spp * frames
    a3 = np.array((albedo,) * 3 if np.isscalar(albedo) else albedo, dtype=np.float64)
    want = expected(a3, e, M)
    got = img.reshape(-1, 3).mean(axis=0)
    amax = a3.max()
    for c in range(3):
        if a3[c] == amax:   # the roulette follows this channel: exact truncated-geometric variance
            sd = e * np.sqrt(hits_variance(a3[c], M) / paths)
        else:               # weights (a_c/amax)^j on the same hit count: bounded by the leading channel's spread
            sd = e * np.sqrt(hits_variance(amax, M) / paths)
        assert abs(got[c] - want[c]) < 4.0 * sd + 1e-6 * want[c], (kind, albedo, c, got[c], want[c], sd)


KINDS = ["sphere", "mesh", "room"]


# ------------------------------------------------------------------ the oracle (CPU)
@pytest.mark.parametrize("kind", KINDS)
def test_oracle_white_furnace_is_exact(pkg, orc, kind):
    tr = orc.create_tracer(8)
    check_white(orc, tr, pkg, kind)
    check_white(orc, tr, pkg, kind, specular=True)
    tr.close()


@pytest.mark.parametrize("kind", KINDS)
def test_oracle_hit_count_is_truncated_geometric(pkg, orc, kind):
    tr = orc.create_tracer(8)
    check_histogram(orc, tr, pkg, kind)
    tr.close()


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("albedo", [0.6, 0.9, (0.8, 0.5, 0.2)])
def test_oracle_furnace_radiance(pkg, orc, kind, albedo):
    tr = orc.create_tracer(8)
    check_mean(orc, tr, pkg, kind, albedo)
    tr.close()


def test_oracle_furnace_radiance_specular_walls(pkg, orc):
    tr = orc.create_tracer(8)
    check_mean(orc, tr, pkg, "room", 0.7, specular=True)
    check_mean(orc, tr, pkg, "sphere", (0.3, 0.9, 0.6), specular=True)
    tr.close()


def test_furnace_detects_a_wrong_integrator():
    """The checks have teeth: emission added after the albedo multiply, an exclusive bounce loop or roulette without
    compensation each move the expectation by far more than the tolerance."""
    a, e, M = 0.6, 0.75, 9
    right = expected(a, e, M)
    sd = e * np.sqrt(hits_variance(a, M) / (40 * 40 * 16))
    assert abs(a * right - right) > 50 * sd                       # light += emission * (throughput * albedo)
    # i < MaxBounceCount (one hit fewer): the white furnace sees it exactly (e*(M-1) != e*M in every pixel), the histogram
    # sees a missing last bucket, and the mean sees it wherever a^(M-1) is not small (albedo 0.9 is in the list above)
    sd9 = e * np.sqrt(hits_variance(0.9, M) / (40 * 40 * 16))
    assert abs(expected(0.9, e, M - 1) - expected(0.9, e, M)) > 10 * sd9
    no_comp = e * sum(a ** (2 * j) for j in range(M))             # roulette without the 1/p: throughput a^j AND survival a^j
    assert abs(no_comp - right) > 50 * sd


# ------------------------------------------------------------------ the HIP kernel (GPU): same physics, and the oracle's bits
@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
def test_hip_white_furnace_is_exact(pkg, api, kind):
    tr = api.create_tracer(0)
    check_white(api, tr, pkg, kind)
    check_white(api, tr, pkg, kind, specular=True)
    tr.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
def test_hip_hit_count_is_truncated_geometric(pkg, api, orc, kind):
    tr, ref = api.create_tracer(0), orc.create_tracer(8)
    k = check_histogram(api, tr, pkg, kind)
    k2 = check_histogram(orc, ref, pkg, kind)
    assert np.array_equal(k, k2)
    tr.close(), ref.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("albedo", [0.6, (0.8, 0.5, 0.2)])
def test_hip_furnace_radiance(pkg, api, orc, kind, albedo):
    tr = api.create_tracer(0)
    check_mean(api, tr, pkg, kind, albedo)
    check_mean(api, tr, pkg, kind, albedo, specular=True)
    # and bit for bit the oracle's image
    sc = furnace(pkg, kind, albedo, 0.75, 8, 4, 32)
    a, _ = radiance(api, tr, sc, 2)
    ref = orc.create_tracer(8)
    b, _ = radiance(orc, ref, furnace(pkg, kind, albedo, 0.75, 8, 4, 32), 2)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    tr.close(), ref.close()
