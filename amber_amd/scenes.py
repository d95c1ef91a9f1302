"""Synthetic scenes of BASELINE.json beyond the reference's built-in Cornell box (host-side data only)."""
from __future__ import annotations

import numpy as np

from . import api


def random_spheres(n: int = 1_000_000, seed: int = 7):
    """BASELINE config 3: n spheres, centres uniform in [-1,1]^3, radius = cbrt(1/n)*(0.5+U),
    materials 2 % DiffuseLight(10), 78 % Lambertian(.7), 10 % Specular(.9), 10 % Refraction(1.5); thin lens at z = +4
    (focal .05, focus 4, radius .01, 6 blades).  The reference has no such scene; the random stream is numpy's
    MT19937(seed) (the survey's definition names mt19937_64(7); only the distribution matters).
    Returns the keyword arguments of HostScene.create_arrays / oracle_binding.Scene.create."""
    rng = np.random.Generator(np.random.MT19937(seed))
    centres = rng.uniform(-1.0, 1.0, (n, 3)).astype(np.float32)
    radius = (np.cbrt(1.0 / n) * (0.5 + rng.random(n))).astype(np.float32)
    u = rng.random(n)
    material = np.where(u < 0.02, 0, np.where(u < 0.80, 1, np.where(u < 0.90, 2, 3))).astype(np.uint32)
    params = np.zeros((n, 12), np.float32)
    params[:, :3], params[:, 3] = centres, radius
    materials = [(api.MAT_DIFFUSE_LIGHT, (10.0, 10.0, 10.0), 0.0), (api.MAT_LAMBERTIAN, (0.7, 0.7, 0.7), 0.0),
                 (api.MAT_SPECULAR, (0.9, 0.9, 0.9), 0.0), (api.MAT_REFRACTION, (1.0, 1.0, 1.0), 1.5)]
    return dict(kinds=np.full(n, api.PRIM_SPHERE, np.uint32), material_index=material, params=params, materials=materials,
                transform=[1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 4, 0, 0, 0, 1], focal_length=0.05, focus_distance=4.0, radius=0.01, n_blades=6)


def as_tuples(scene_kwargs):
    """The same scene as lists of tuples (for oracle_binding.Scene.create / HostScene.create)."""
    k = scene_kwargs
    n_par = {0: 9, 1: 4, 2: 7, 3: 8}
    objects = [(int(kind), int(m), [float(x) for x in p[: n_par[int(kind)]]]) for kind, m, p in zip(k["kinds"], k["material_index"], k["params"])]
    return dict(objects=objects, materials=k["materials"], transform=k["transform"], focal_length=k["focal_length"],
                focus_distance=k["focus_distance"], radius=k["radius"], n_blades=k["n_blades"])


def mesh_room(subdivisions: int = 8, seed: int = 3):
    """A triangle-mesh stress scene for engine BVH (not a BASELINE config): a Cornell-like room of 10 wall triangles and
    a ceiling light, containing a bumpy icosphere of 20 * 4**subdivisions triangles (subdivisions = 8: 1.3 M) with a
    glossy material.  Returns the keyword arguments of HostScene.create_arrays."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)], np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
                  (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)], np.int64)
    for _ in range(subdivisions):
        e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
        ue, inv = np.unique(e, axis=0, return_inverse=True)
        mid = v[ue[:, 0]] + v[ue[:, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        m = len(v) + inv.reshape(3, -1)                       # midpoint index of edges (0,1), (1,2), (2,0) of every face
        v = np.concatenate([v, mid])
        a, b, c = f[:, 0], f[:, 1], f[:, 2]
        f = np.concatenate([np.stack([a, m[0], m[2]], 1), np.stack([b, m[1], m[0]], 1), np.stack([c, m[2], m[1]], 1), np.stack([m[0], m[1], m[2]], 1)])
    rng = np.random.Generator(np.random.MT19937(seed))
    bump = 1.0 + 0.02 * np.sin(9.0 * v[:, 0]) * np.sin(7.0 * v[:, 1] + 1.0) * np.sin(8.0 * v[:, 2] + 2.0) + 0.002 * rng.standard_normal(len(v))
    p = (np.array([0.1, -0.35, -0.1]) + 0.6 * v * bump[:, None]).astype(np.float32)
    tri = np.concatenate([p[f[:, 0]], p[f[:, 1]], p[f[:, 2]]], axis=1)                       # (n, 9)
    c8 = np.array([(-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1)], np.float32)
    quads = [(0, 1, 2, 3, 1), (0, 4, 5, 1, 1), (3, 2, 6, 7, 1), (0, 3, 7, 4, 2), (1, 5, 6, 2, 3)]    # back, floor, ceiling, left (red), right (green)
    walls, wall_mat = [], []
    for a, b, c, d, m in quads:
        walls += [np.concatenate([c8[a], c8[b], c8[c]]), np.concatenate([c8[c], c8[d], c8[a]])]; wall_mat += [m, m]
    L = np.array([(-0.3, 0.99, -0.3), (0.3, 0.99, -0.3), (0.3, 0.99, 0.3), (-0.3, 0.99, 0.3)], np.float32)
    walls += [np.concatenate([L[0], L[1], L[2]]), np.concatenate([L[2], L[3], L[0]])]; wall_mat += [0, 0]
    params = np.zeros((len(walls) + len(tri), 12), np.float32)
    params[: len(walls), :9] = np.array(walls); params[len(walls):, :9] = tri
    material = np.concatenate([np.array(wall_mat, np.uint32), np.full(len(tri), 4, np.uint32)])
    materials = [(api.MAT_DIFFUSE_LIGHT, (20.0, 20.0, 20.0), 0.0), (api.MAT_LAMBERTIAN, (0.75, 0.75, 0.75), 0.0), (api.MAT_LAMBERTIAN, (0.75, 0.25, 0.25), 0.0),
                 (api.MAT_LAMBERTIAN, (0.25, 0.75, 0.25), 0.0), (api.MAT_PHONG, (0.8, 0.8, 0.8), 40.0)]
    return dict(kinds=np.full(len(params), api.PRIM_TRIANGLE, np.uint32), material_index=material, params=params, materials=materials,
                transform=[1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 4, 0, 0, 0, 1], focal_length=0.05, focus_distance=4.0, radius=0.01, n_blades=6)


def cornell_plus(k: int, seed: int = 5):
    """The Cornell box (etude::CornelBox(0.050, 0.050, 6), read back from the host model) plus k extra objects -- small diffuse quads of two triangles
    scattered through the room, and one small sphere when k is odd: scenes on both sides of the engine switches (32 / 80 objects).
    Returns the keyword arguments of HostScene.create_arrays / oracle_binding.Scene.create_arrays (the aperture blades are inserted by create)."""
    objs, mats, lens = api.HostScene.cornell_box().flatten()
    arr = np.frombuffer(objs, dtype=np.dtype([("kind", np.uint32), ("material", np.uint32), ("p", np.float32, (12,))])).copy()
    n_blades = int(lens.n_blades)
    body = arr[n_blades:]
    materials = [(int(m.kind), tuple(float(x) for x in m.rho[:]), float(m.param)) for m in mats]
    diffuse = next(i for i, m in enumerate(materials) if m[0] == api.MAT_LAMBERTIAN)
    rng = np.random.default_rng(seed)
    extra = np.zeros(k, arr.dtype)
    for i in range(0, k - 1, 2):
        c = rng.uniform([-0.85, -0.95, -0.85], [0.85, 0.2, 0.85]); a = rng.normal(size=3) * 0.06; b = rng.normal(size=3) * 0.06
        q = [c, c + a, c + a + b, c + b]
        extra[i]["p"][:9] = np.concatenate([q[0], q[1], q[2]]); extra[i + 1]["p"][:9] = np.concatenate([q[2], q[3], q[0]])
        extra[i]["material"] = extra[i + 1]["material"] = diffuse
    if k % 2:
        extra[k - 1]["kind"] = api.PRIM_SPHERE; extra[k - 1]["p"][:4] = [*rng.uniform(-0.8, 0.8, 3), 0.05]; extra[k - 1]["material"] = diffuse
    allo = np.concatenate([body, extra])
    g = np.array(lens.global_[:], np.float32).reshape(3, 3); o = np.array(lens.origin[:], np.float32)
    transform = [g[0, 0], g[0, 1], g[0, 2], o[0], g[1, 0], g[1, 1], g[1, 2], o[1], g[2, 0], g[2, 1], g[2, 2], o[2], 0, 0, 0, 1]
    return dict(kinds=allo["kind"].copy(), material_index=allo["material"].copy(), params=allo["p"].copy(), materials=materials, transform=[float(x) for x in transform],
                focal_length=0.050, focus_distance=float(lens.focus_distance), radius=0.050, n_blades=n_blades)
