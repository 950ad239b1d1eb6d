"""Procedural meshes for the synthetic scenes.

The reference's meshes are Unity built-ins (Cube / Quad — engine resources, not in
the repository), `cube_rounded2.obj`, and blobs that are missing from the snapshot
(Dragon_80K.obj, Icosphere.obj — SURVEY.md §2 row 12).  These generators produce
stand-ins of the same shape class and triangle counts.  Every triangle obeys the
winding convention RayTriangle implies (RayCommon.hlsl:192-195,206-209):
cross(B-A, C-A) points along the outward vertex normal, otherwise opaque surfaces
are back-face culled (RC:355).
"""
import numpy as np


class Mesh:
    """verts (N,3) f32, normals (N,3) f32, indices (3*T,) i32 — what BVH.cs's
    constructor takes (Mesh.vertices / .triangles / .normals, RCM:218)."""

    def __init__(self, verts, normals, indices, name="mesh"):
        self.vertices = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
        self.normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        self.triangles = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1)
        self.name = name
        assert len(self.vertices) == len(self.normals)

    @property
    def triangle_count(self):
        return len(self.triangles) // 3


def quad():
    """Unity built-in Quad: 4 verts (+-0.5, +-0.5, 0), normal (0,0,-1), 2 triangles."""
    v = np.array([[-0.5, -0.5, 0], [0.5, -0.5, 0], [-0.5, 0.5, 0], [0.5, 0.5, 0]], dtype=np.float32)
    n = np.tile(np.array([[0, 0, -1]], dtype=np.float32), (4, 1))
    idx = np.array([0, 3, 1, 3, 0, 2], dtype=np.int32)
    return Mesh(v, n, idx, "Quad")


def cube():
    """Unity built-in Cube: side 1, centred, 24 verts (4 per face, face normals), 12 tris."""
    verts, norms, idx = [], [], []
    for axis in range(3):
        for sgn in (-1.0, 1.0):
            nrm = np.zeros(3)
            nrm[axis] = sgn
            u = np.zeros(3)
            v = np.zeros(3)
            u[(axis + 1) % 3] = 1.0
            v[(axis + 2) % 3] = 1.0
            if sgn < 0:  # keep cross(u, v) == outward normal
                u, v = v, u
            base = len(verts)
            for (a, b) in ((-0.5, -0.5), (0.5, -0.5), (0.5, 0.5), (-0.5, 0.5)):
                verts.append(nrm * 0.5 + u * a + v * b)
                norms.append(nrm)
            idx += [base, base + 1, base + 2, base, base + 2, base + 3]
    return Mesh(np.array(verts), np.array(norms), np.array(idx), "Cube")


def rounded_cube(k=12, radius=0.18):
    """Unit cube with rounded edges, k x k quads per face (k=12 -> 1,728 triangles,
    the size class of the reference's cube_rounded2.obj with 1,724). Smooth normals."""
    verts, norms, idx = [], [], []
    inner = 0.5 - radius
    for axis in range(3):
        for sgn in (-1.0, 1.0):
            nrm = np.zeros(3)
            nrm[axis] = sgn
            u = np.zeros(3)
            v = np.zeros(3)
            u[(axis + 1) % 3] = 1.0
            v[(axis + 2) % 3] = 1.0
            if sgn < 0:
                u, v = v, u
            base = len(verts)
            for j in range(k + 1):
                for i in range(k + 1):
                    q = nrm * 0.5 + u * (i / k - 0.5) + v * (j / k - 0.5)
                    c = np.clip(q, -inner, inner)
                    d = q - c
                    ln = np.linalg.norm(d)
                    nn = d / ln if ln > 1e-12 else nrm
                    verts.append(c + nn * radius)
                    norms.append(nn)
            for j in range(k):
                for i in range(k):
                    a = base + j * (k + 1) + i
                    b = a + 1
                    c2 = a + (k + 1) + 1
                    d2 = a + (k + 1)
                    idx += [a, b, c2, a, c2, d2]
    return Mesh(np.array(verts), np.array(norms), np.array(idx), "RoundedCube")


def _lcg(seed):
    """Small deterministic generator (numerical-recipes LCG) so scenes do not depend
    on numpy's RNG stream."""
    state = [seed & 0xFFFFFFFF]

    def nxt():
        state[0] = (state[0] * 1664525 + 1013904223) & 0xFFFFFFFF
        return state[0] / 4294967296.0

    return nxt


def icosphere(subdivisions=2, radius=1.0, displacement_seed=None, displacement=0.12):
    """Icosphere with 20*4^s triangles (s=6 -> 81,920, the 'bunny/dragon class').
    With a seed, vertices are displaced radially by 3 octaves of smooth lobes and
    the smooth normals are recomputed (area-weighted)."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(subdivisions):
        edges = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0)
        es = np.sort(edges, axis=1)
        uniq, inv = np.unique(es, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        mid = v[uniq[:, 0]] + v[uniq[:, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        base = len(v)
        v = np.concatenate([v, mid], axis=0)
        nf = len(f)
        m01 = base + inv[0:nf]
        m12 = base + inv[nf:2 * nf]
        m20 = base + inv[2 * nf:3 * nf]
        f = np.concatenate([
            np.stack([f[:, 0], m01, m20], axis=1), np.stack([f[:, 1], m12, m01], axis=1),
            np.stack([f[:, 2], m20, m12], axis=1), np.stack([m01, m12, m20], axis=1)], axis=0)
    # The seed icosahedron above is wound so that cross(B-A, C-A) points inward for
    # this face table; check and flip so it points outward.
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    outward = np.einsum("ij,ij->i", np.cross(b - a, c - a), a + b + c) > 0
    f[~outward] = f[~outward][:, [0, 2, 1]]

    if displacement_seed is not None:
        rnd = _lcg(displacement_seed)
        scale = np.ones(len(v))
        for octave in range(3):
            nl = 6 * (2 ** octave)
            amp = displacement / (2 ** octave)
            for _ in range(nl):
                z = 2 * rnd() - 1
                ph = 2 * np.pi * rnd()
                r = (1 - z * z) ** 0.5
                d = np.array([r * np.cos(ph), r * np.sin(ph), z])
                sharp = 4.0 * (2 ** octave)
                sgn = 1.0 if rnd() < 0.6 else -1.0
                scale += sgn * amp * np.exp(sharp * (v @ d - 1.0))
        v = v * scale[:, None]
        a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
        fn = np.cross(b - a, c - a)
        n = np.zeros_like(v)
        for k in range(3):
            np.add.at(n, f[:, k], fn)
        n /= np.linalg.norm(n, axis=1, keepdims=True)
    else:
        n = v.copy()
    return Mesh(v * radius, n, f.reshape(-1), f"Icosphere{subdivisions}" + ("" if displacement_seed is None else f"_d{displacement_seed}"))


def check_winding(mesh):
    """Fraction of triangles whose cross(B-A, C-A) agrees with the mean vertex normal."""
    f = mesh.triangles.reshape(-1, 3)
    v = mesh.vertices.astype(np.float64)
    n = mesh.normals.astype(np.float64)
    fn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    mn = n[f[:, 0]] + n[f[:, 1]] + n[f[:, 2]]
    return float(np.mean(np.einsum("ij,ij->i", fn, mn) > 0))
