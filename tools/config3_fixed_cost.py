"""Config 3 (1M spheres, engine BVH): kernel time against samples per launch -- fixed part (start + tail) and slope."""
import sys, os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import amber_amd as A
from amber_amd import scenes
flags = next((int(a) for a in sys.argv[1:] if a.isdigit()), 0)
import amber_amd.api as api
for a in sys.argv[1:]:
    if a.endswith(".so"): api._LIB_PATH = api._ROOT / "lib" / a
hs = A.HostScene.create_arrays(**scenes.random_spheres(1_000_000, 7))
pt = A.PathTracer(hs, A.Sensor.default(1920, 1080), seed=1, flags=flags)
pt.render_pass(0, 8); pt.sync(); pt.clear()
xs, ys = [], []
for spp in (16, 32, 64, 128, 256):
    best = 1e9
    for rep in range(2):
        k0, m0 = pt.kernel_time(); pt.render_pass(0, spp); pt.sync(); k1, m1 = pt.kernel_time(); best = min(best, m1 - m0)
    xs.append(spp); ys.append(best)
    print("%4d spp: %8.2f ms (%.3f ms per spp)" % (spp, best, best / spp), flush=True)
b, a = np.polyfit(xs, ys, 1)
print("fit: %.2f ms fixed + %.4f ms per spp; 256 spp: %.1f ms, of which fixed %.1f %%" % (a, b, a + 256 * b, 100 * a / (a + 256 * b)))
