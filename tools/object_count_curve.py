"""Kernel time against the number of objects, across the engine switch at 33 objects: the Cornell box plus k extra objects (small diffuse / glossy
quads of two triangles and a few spheres scattered through the room), 1024 x 1024 @ 128 spp, every engine that accepts the scene."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import amber_amd as A


def cornell_plus(k, seed=5):
    objs, mats, lens = A.HostScene.cornell_box().flatten()
    arr = np.frombuffer(objs, dtype=np.dtype([("kind", np.uint32), ("material", np.uint32), ("p", np.float32, (12,))])).copy()
    n_blades = lens.n_blades
    body = arr[n_blades:]                                      # create() inserts the aperture blades itself, first (cornel_box.cc:62-64)
    materials = [(m.kind, tuple(m.rho[:]), m.param) for m in mats]
    rng = np.random.default_rng(seed)
    extra = np.zeros(k, arr.dtype)
    diffuse = [i for i, m in enumerate(materials) if m[0] == 0][:1] or [0]
    for i in range(0, k - 1, 2):                               # a small quad = two triangles sharing an edge (a parallelogram: one filter record)
        c = rng.uniform([-0.85, -0.95, -0.85], [0.85, 0.2, 0.85]); a = rng.normal(size=3) * 0.06; b = rng.normal(size=3) * 0.06
        q = [c, c + a, c + a + b, c + b]
        extra[i]["kind"] = 0; extra[i]["p"][:9] = np.concatenate([q[0], q[1], q[2]])
        extra[i + 1]["kind"] = 0; extra[i + 1]["p"][:9] = np.concatenate([q[2], q[3], q[0]])
        extra[i]["material"] = extra[i + 1]["material"] = diffuse[0]
    if k % 2:
        extra[k - 1]["kind"] = 1; extra[k - 1]["p"][:4] = [*rng.uniform(-0.8, 0.8, 3), 0.05]; extra[k - 1]["material"] = diffuse[0]
    allo = np.concatenate([body, extra])
    g = np.array(lens.global_[:], np.float32).reshape(3, 3); o = np.array(lens.origin[:], np.float32)
    transform = [g[0, 0], g[0, 1], g[0, 2], o[0], g[1, 0], g[1, 1], g[1, 2], o[1], g[2, 0], g[2, 1], g[2, 2], o[2], 0, 0, 0, 1]
    return A.HostScene.create_arrays(kinds=allo["kind"], material_index=allo["material"], params=allo["p"], materials=materials, transform=transform,
                                     focal_length=0.050, focus_distance=float(lens.focus_distance), radius=0.050, n_blades=int(n_blades)), len(allo) + n_blades


counts = [int(x) for x in sys.argv[1:]] or [0, 4, 7, 8, 16, 32, 64, 100, 200, 400, 1000]
print("%-8s %10s %10s %12s %12s   rays" % ("objects", "auto", "list", "bvh (auto)", "bvh items"), flush=True)
for k in counts:
    hs, n = cornell_plus(k)
    row, rays = [], None
    for engine, flags in ((A.ENGINE_AUTO, 0), (A.ENGINE_LIST, 0), (A.ENGINE_BVH, 0), (A.ENGINE_BVH, A.PT_FLAG_BVH_ITEMS)):
        if engine == A.ENGINE_LIST and n > 300: row.append(float("nan")); continue
        pt = A.PathTracer(hs, A.Sensor.default(1024, 1024), seed=12345, engine=engine, flags=flags)
        pt.render_pass(0, 8); pt.sync(); pt.clear()
        pt.render_pass(0, 128); pt.sync()
        row.append(pt.kernel_time()[1]); r = pt.ray_count(); pt.close()
        assert rays is None or rays == r, (rays, r)
        rays = r
    print("%-8d %10.2f %10.2f %12.2f %12.2f   %d" % (n, row[0], row[1], row[2], row[3], rays), flush=True)
