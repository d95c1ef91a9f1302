"""Kernel time of engine BVH on config 3 as a function of the samples per launch (looks for per-launch fixed costs)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import amber_amd as A
from amber_amd import scenes
hs = A.HostScene.create_arrays(**scenes.random_spheres(int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 7))
for (W, H) in ((1920, 1080), (480, 270)):
    pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=1)
    pt.render_pass(0, 8); pt.sync()
    for spp in (8, 8, 16, 32, 64, 128, 256):
        pt.clear(); pt.render_pass(0, spp); pt.sync(); n, ms = pt.kernel_time(); r = pt.ray_count()
        print("%dx%d spp %4d: %8.1f ms  %6.1f Mrays/s  (%d launches)" % (W, H, spp, ms, r / ms / 1e3, n), flush=True)
    pt.close()
