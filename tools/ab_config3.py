"""A/B of library builds on BASELINE config 3 (1M spheres, engine BVH) in ONE process, interleaved rounds."""
import os, sys, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
libs = [a for a in sys.argv[1:] if a.endswith('.so')]
spp = next((int(a) for a in sys.argv[1:] if a.isdigit()), 64)
import amber_amd.api as api
import amber_amd as A
from amber_amd import scenes
kw = scenes.random_spheres(1_000_000, 7)
handles, res = {}, {}
for path in libs:
    api._lib = None; api._LIB_PATH = api._ROOT / "lib" / path
    lib = A.load_library()
    hs = A.HostScene.create_arrays(**kw); pt = A.PathTracer(hs, A.Sensor.default(1920, 1080), seed=1)
    handles[path] = (lib, hs, pt)
for rnd in range(3):
    for path in libs:
        lib, hs, pt = handles[path]; api._lib = lib
        pt.clear(); pt.render_pass(0, spp); pt.sync(); n, ms = pt.kernel_time()
        res.setdefault(path, []).append(ms)
        print(path, rnd, "%.1f ms" % ms, pt.ray_count(), flush=True)
for path in libs:
    print("%-32s median %.1f ms  (%s)" % (path, statistics.median(res[path]), " ".join("%.1f" % x for x in res[path])))
sys.stdout.flush(); os._exit(0)   # several copies of the library are loaded: skip their exit-time teardown (it can abort)
