"""Runs the same renders many times (fresh handles) and checks that images and ray counts never change: config 2 at 256 spp,
config 3 at 16 spp, a striped band, and (round 5) the 1M-triangle terrain, the room mesh through the path-granular BVH kernel, a 49-object scene through
the two-phase engine's groups, config 3 and the terrain through engine REFERENCE_BVH."""
import sys, os, hashlib; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import amber_amd as A
from amber_amd import scenes, workloads
import tempfile
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
box = A.HostScene.cornell_box(); sph = A.HostScene.create_arrays(**scenes.random_spheres(1_000_000, 7))
_d = tempfile.mkdtemp()
terrain = A.HostScene.import_file(workloads.terrain_mesh(16, 56).write(_d)); room = A.HostScene.import_file(workloads.room_mesh(3).write(_d))
plus = A.HostScene.create_arrays(**scenes.cornell_plus(24))
seen = {}
for i in range(n):
    for name, make, spp in (("config 2 @256", lambda: A.PathTracer(box, A.Sensor.default(1024, 1024), seed=12345), 256),
                            ("config 3 @16", lambda: A.PathTracer(sph, A.Sensor.default(1920, 1080), seed=1), 16),
                            ("terrain @8", lambda: A.PathTracer(terrain, A.Sensor.default(1920, 1080), seed=2), 8),
                            ("room mesh @64", lambda: A.PathTracer(room, A.Sensor.default(1024, 1024), seed=2), 64),
                            ("49 objects @64", lambda: A.PathTracer(plus, A.Sensor.default(1024, 1024), seed=2), 64),
                            ("config 3 through engine REFERENCE_BVH @16 (the host build runs on threads: same tree every time)", lambda: A.PathTracer(sph, A.Sensor.default(1920, 1080), seed=1, engine=A.ENGINE_REFERENCE_BVH), 16),
                            ("terrain through engine REFERENCE_BVH @8", lambda: A.PathTracer(terrain, A.Sensor.default(1920, 1080), seed=2, engine=A.ENGINE_REFERENCE_BVH), 8),
                            ("config 5 stripes @64", lambda: A.PathTracer(box, A.Sensor.default(3840, 2160), seed=3, max_depth=16, rows=(8, 2160), stripe=(8, 64)), 64)):
        pt = make(); pt.render_pass(0, spp); img, rays = pt.download(); pt.close()
        seen.setdefault(name, set()).add((rays, hashlib.sha1(img.tobytes()).hexdigest()))
    if i % 10 == 0: print("round", i, {k: len(v) for k, v in seen.items()}, flush=True)
print({k: (len(v), next(iter(v))[0]) for k, v in seen.items()})
assert all(len(v) == 1 for v in seen.values())
print("soak ok: %d rounds" % n)
