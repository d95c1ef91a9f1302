"""Ray count of BASELINE config 3 under increasingly conservative sphere bounds: the count must stop changing."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import amber_amd.api as api
from amber_amd import scenes
k = scenes.random_spheres(1_000_000, 7)
ref = None
for path in sys.argv[1:]:
    api._lib = None; api._LIB_PATH = api._ROOT / "lib" / path
    import amber_amd as A
    A.load_library()
    hs = A.HostScene.create_arrays(**k); pt = A.PathTracer(hs, A.Sensor.default(1920, 1080), seed=1)
    pt.render_pass(0, 256); img, rays = pt.download(); n, ms = pt.kernel_time()
    same = "" if ref is None else ("image identical to first: %s" % np.array_equal(ref.view(np.uint32), img.view(np.uint32)))
    if ref is None: ref = img
    print("%-28s rays %d  kernel %.0f ms  %s" % (path, rays, ms, same)); pt.close()
