"""Kernel time against the number of objects, across the engine switch at 33 objects: the Cornell box plus k extra objects (small diffuse / glossy
quads of two triangles and a few spheres scattered through the room), 1024 x 1024 @ 128 spp, every engine that accepts the scene."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import amber_amd as A


def cornell_plus(k, seed=5):
    from amber_amd import scenes
    kw = scenes.cornell_plus(k, seed)
    return A.HostScene.create_arrays(**kw), len(kw["kinds"]) + kw["n_blades"]


counts = [int(x) for x in sys.argv[1:]] or [0, 4, 7, 8, 16, 24, 32, 39, 48, 64, 80, 96, 103, 200, 400, 1000]
print("%-8s %10s %10s %10s %12s %12s   rays" % ("objects", "auto", "two-phase", "list", "bvh (auto)", "bvh items"), flush=True)
for k in counts:
    hs, n = cornell_plus(k)
    row, rays = [], None
    for engine, flags in ((A.ENGINE_AUTO, 0), (A.ENGINE_TWO_PHASE, 0), (A.ENGINE_LIST, 0), (A.ENGINE_BVH, 0), (A.ENGINE_BVH, A.PT_FLAG_BVH_ITEMS)):
        if (engine == A.ENGINE_LIST and n > 300) or (engine == A.ENGINE_TWO_PHASE and n > 128): row.append(float("nan")); continue
        pt = A.PathTracer(hs, A.Sensor.default(1024, 1024), seed=12345, engine=engine, flags=flags)
        pt.render_pass(0, 8); pt.sync(); pt.clear()
        pt.render_pass(0, 128); pt.sync()
        row.append(pt.kernel_time()[1]); r = pt.ray_count(); pt.close()
        assert rays is None or rays == r, (rays, r)
        rays = r
    print("%-8d %10.2f %10.2f %10.2f %12.2f %12.2f   %d" % (n, row[0], row[1], row[2], row[3], row[4], rays), flush=True)
