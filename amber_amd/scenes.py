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
