"""Random small scenes for the engine fuzzer (tools/fuzz_engines.py) and its regression test: triangles, exact and
perturbed parallelograms, fans, coplanar clutter, slivers, spheres, disks with NON-unit normals, cylinders, all
materials, thin and pinhole lenses.  Returns the keyword arguments of HostScene.create / oracle Scene.create."""
import numpy as np


def f32(v):
    return [float(np.float32(x)) for x in v]


def random_scene(rng, big, normal_scale=1.0, scale=1.0, offset=(0.0, 0.0, 0.0)):
    """normal_scale multiplies the (already non-unit) disk normals: sampled directions then have lengths far from 1.
    scale / offset apply a similarity transform to all geometry and to the camera position (precision stress)."""
    mats = [(4, tuple(rng.uniform(5, 40, 3)), 0.0), (0, tuple(rng.uniform(0.2, 0.9, 3)), 0.0), (1, tuple(rng.uniform(0.5, 0.95, 3)), float(rng.choice([4.0, 32.0, 256.0]))),
            (2, tuple(rng.uniform(0.7, 0.95, 3)), 0.0), (3, (1.0, 1.0, 1.0), float(rng.choice([1.125, 1.333, 1.5]))), (0, (0.5, 0.5, 0.5), 0.0)]
    objs = []
    budget = int(rng.integers(40, 400)) if big else int(rng.integers(3, 27))      # leaves room for <= 6 aperture blades in the 32-object engines

    def mat():
        return int(rng.integers(0, len(mats)))

    def point(scale=1.0):
        return rng.uniform(-1, 1, 3) * scale
    while len(objs) < budget:
        kind = rng.choice(["tri", "quad", "skewquad", "nearquad", "fan", "coplanar", "axisquad", "sphere", "disk", "cyl", "sliver"],
                          p=[0.12, 0.16, 0.1, 0.08, 0.08, 0.08, 0.12, 0.12, 0.05, 0.05, 0.04])
        if kind == "tri":
            a, b, c = point(), point(), point()
            objs.append((0, mat(), f32(np.concatenate([a, b, c]))))
        elif kind in ("quad", "skewquad", "nearquad", "axisquad"):
            a = point()
            if kind == "axisquad":                                   # axis-aligned rectangle (Cornell-like walls)
                ax = int(rng.integers(0, 3)); e1 = np.zeros(3); e2 = np.zeros(3)
                e1[(ax + 1) % 3] = rng.uniform(0.3, 2); e2[(ax + 2) % 3] = rng.uniform(0.3, 2)
                a = np.round(a * 4) / 4
            else:
                e1, e2 = point(1.2), point(1.2)
            b, c, d = a + e1, a + e1 + e2, a + e2
            if kind == "skewquad":
                d = d + 0.3 * e1                                     # coplanar but NOT a parallelogram
            if kind == "nearquad":
                d = d + rng.uniform(-1, 1, 3) * 1e-6                 # parallelogram up to rounding-size noise
            m = mat()
            order = int(rng.integers(0, 4))                          # different corner orders / shared diagonals
            if order == 0: t1, t2 = (a, b, c), (c, d, a)
            elif order == 1: t1, t2 = (b, c, a), (a, c, d)
            elif order == 2: t1, t2 = (a, b, d), (b, c, d)
            else: t1, t2 = (c, a, b), (d, a, c)
            objs.append((0, m, f32(np.concatenate(t1)))); objs.append((0, m if rng.random() < 0.8 else mat(), f32(np.concatenate(t2))))
        elif kind == "fan":
            c0, n = point(), int(rng.integers(3, 7))
            u = point(); u /= np.linalg.norm(u); w = np.cross(u, point()); w /= np.linalg.norm(w); r = rng.uniform(0.1, 0.8)
            vs = [c0 + r * (np.cos(2 * np.pi * k / n) * u + np.sin(2 * np.pi * k / n) * w) for k in range(n)]
            m = mat()
            for k in range(n):
                objs.append((0, m, f32(np.concatenate([c0, vs[k], vs[(k + 1) % n]]))))
        elif kind == "coplanar":                                     # unrelated triangles in one plane (shared plane record, no pairing)
            o0, u, w = point(), point(), point()
            for _ in range(int(rng.integers(2, 5))):
                p = [o0 + rng.uniform(-1, 1) * u + rng.uniform(-1, 1) * w for _ in range(3)]
                objs.append((0, mat(), f32(np.concatenate(p))))
        elif kind == "sliver":
            a = point(); e = point()
            objs.append((0, mat(), f32(np.concatenate([a, a + e, a + e * (1 + 1e-4) + rng.uniform(-1, 1, 3) * 1e-5]))))
        elif kind == "sphere":
            objs.append((1, mat(), f32(list(point()) + [rng.uniform(0.05, 0.6)])))
        elif kind == "disk":
            objs.append((2, mat(), f32(list(point()) + list(point() * normal_scale) + [rng.uniform(0.1, 0.8)])))
        else:
            n = point(); n /= np.linalg.norm(n)
            objs.append((3, mat(), f32(list(point()) + list(n) + [rng.uniform(0.05, 0.4), rng.uniform(0.2, 1.0)])))
    objs = objs[:budget]
    z = float(rng.uniform(2.5, 5))
    tf = [1, 0, 0, float(rng.uniform(-0.3, 0.3)), 0, 1, 0, float(rng.uniform(-0.3, 0.3)), 0, 0, 1, z, 0, 0, 0, 1]
    if rng.random() < 0.25:
        lens = dict(focal_length=0.045, focus_distance=z, radius=0.02, n_blades=0)             # pinhole
    else:
        lens = dict(focal_length=0.05, focus_distance=z, radius=float(rng.choice([0.01, 0.05])), n_blades=int(rng.integers(3, 7)))
    if scale != 1.0 or any(offset):
        off = np.array(offset, np.float64)

        def pos(v):
            return list(np.array(v, np.float64) * scale + off)
        moved = []
        for kind, m, p in objs:
            if kind == 0:
                p2 = pos(p[0:3]) + pos(p[3:6]) + pos(p[6:9])
            elif kind == 1:
                p2 = pos(p[0:3]) + [p[3] * scale]
            elif kind == 2:
                p2 = pos(p[0:3]) + list(p[3:6]) + [p[6] * scale]
            else:
                p2 = pos(p[0:3]) + list(p[3:6]) + [p[6] * scale, p[7] * scale]
            moved.append((kind, m, f32(p2)))
        objs = moved
        t3 = pos([tf[3], tf[7], tf[11]])
        tf = [1, 0, 0, float(np.float32(t3[0])), 0, 1, 0, float(np.float32(t3[1])), 0, 0, 1, float(np.float32(t3[2])), 0, 0, 0, 1]
        lens = dict(lens, focus_distance=float(np.float32(lens["focus_distance"] * scale)))
    return dict(objects=objs, materials=[(k, tuple(float(x) for x in rho), p) for k, rho, p in mats], transform=tf, **lens)


def scene_for_seed(seed, scaled=False, extreme=False):
    """The scene tools/fuzz_engines.py builds for `seed` under its --scaled / --extreme switches."""
    rng = np.random.default_rng(seed)
    kw = {}
    if scaled:
        s = float(10.0 ** rng.uniform(-2, 3))
        kw = dict(scale=s, offset=tuple(float(x) for x in rng.uniform(-100, 100, 3) * s * float(rng.integers(0, 2))))
    return random_scene(rng, seed % 4 == 3, normal_scale=(10.0 ** rng.uniform(-2, 2)) if extreme else 1.0, **kw), rng
