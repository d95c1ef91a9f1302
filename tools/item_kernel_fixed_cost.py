"""pt_bvh_megakernel: kernel time against samples per launch -- fixed part (start + tail) and slope -- for the 1M-sphere scene and the 1M-triangle terrain."""
import sys, os, tempfile; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import amber_amd as A
from amber_amd import scenes, workloads as WL
d = tempfile.mkdtemp()
for name, hs, spps in (("1M spheres", A.HostScene.create_arrays(**scenes.random_spheres(1_000_000, 7)), (8, 16, 32, 64, 128, 256)),
                       ("terrain 1.04M triangles", A.HostScene.import_file(WL.terrain_mesh(16, 56).write(d)), (8, 16, 32, 64, 128))):
    pt = A.PathTracer(hs, A.Sensor.default(1920, 1080), seed=1)
    pt.render_pass(0, 8); pt.sync(); pt.clear()
    xs, ys = [], []
    for spp in spps:
        best = 1e9
        for rep in range(2):
            k0, m0 = pt.kernel_time(); pt.render_pass(0, spp); pt.sync(); k1, m1 = pt.kernel_time(); best = min(best, m1 - m0)
        xs.append(spp); ys.append(best)
        print("%-24s %4d spp: %8.2f ms (%.3f ms per spp)" % (name, spp, best, best / spp), flush=True)
    b, a = np.polyfit(xs, ys, 1)
    print("%-24s fit: %.2f ms fixed + %.4f ms per spp" % (name, a, b), flush=True)
    pt.close()
