"""pt_megakernel on config 2's frame: kernel time against samples per launch -- the fixed part of a launch (start-up + tail), which
is what strong scaling over N GPUs pays N times.  python tools/launch_fixed_cost.py [lib.so ...]"""
import sys, os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import amber_amd.api as api
import amber_amd as A
libs = [a for a in sys.argv[1:] if a.endswith(".so")] or ["libamber_hip.so"]
for path in libs:
    api._lib = None; api._LIB_PATH = api._ROOT / "lib" / path
    lib = A.load_library()
    pt = A.PathTracer(A.HostScene.cornell_box(), A.Sensor.default(1024, 1024))
    pt.render_pass(0, 8); pt.sync(); pt.clear()
    xs, ys = [], []
    for spp in (16, 32, 64, 128, 256, 512, 1024):
        best = 1e9
        for rep in range(3):
            k0, m0 = pt.kernel_time(); pt.render_pass(0, spp); pt.sync(); k1, m1 = pt.kernel_time(); best = min(best, m1 - m0)
        xs.append(spp); ys.append(best)
        print("%s %5d spp: %8.3f ms  (%.4f ms per spp)" % (path, spp, best, best / spp), flush=True)
    b, a = np.polyfit(xs, ys, 1)
    print("%s fit: %.3f ms fixed + %.5f ms per spp (1024 spp: %.2f ms); at 128 spp (one rank of 8) the fixed part is %.1f %%" % (path, a, b, a + 1024 * b, 100 * a / (a + 128 * b)))
    pt.close()
os._exit(0)
