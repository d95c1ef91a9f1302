"""A/B of two or more builds of the library on config 2 in ONE process, interleaved rounds (cdna guide rule 24)."""
import ctypes as C, os, sys, time, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
libs = [a for a in sys.argv[1:] if a.endswith('.so')]
spp = next((int(a) for a in sys.argv[1:] if a.isdigit()), 256)
import amber_amd.api as api
res = {}
handles = {}
for path in libs:
    api._lib = None; api._LIB_PATH = api._ROOT / "lib" / path
    import amber_amd as A
    lib = A.load_library()
    sc = A.HostScene.cornell_box(); pt = A.PathTracer(sc, A.Sensor.default(1024, 1024))
    handles[path] = (lib, sc, pt)
for rnd in range(5):
    for path in libs:
        lib, sc, pt = handles[path]; api._lib = lib
        pt.clear(); pt.render_pass(0, spp); pt.sync(); n, ms = pt.kernel_time()
        res.setdefault(path, []).append(ms)
for path in libs:
    print("%-28s median %.2f ms  min %.2f  (%s)" % (path, statistics.median(res[path]), min(res[path]), " ".join("%.1f" % x for x in res[path])))
sys.stdout.flush(); os._exit(0)   # several copies of the library are loaded: skip their exit-time teardown (it can abort)
