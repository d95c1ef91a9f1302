"""Finds the paths on which two builds of the BVH engine disagree (config 3) and asks the exhaustive LIST engine who is right."""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import amber_amd.api as api
from amber_amd import scenes
W, H, spp = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 256
k = scenes.random_spheres(1_000_000, 7)
imgs = {}; tracers = {}
for path in ("libamber_hip.so", "libamber_hip_submul.so"):
    api._lib = None; api._LIB_PATH = api._ROOT / "lib" / path
    import amber_amd as A
    lib = A.load_library()
    hs = A.HostScene.create_arrays(**k); pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=1)
    pt.render_pass(0, spp); imgs[path], rays = pt.download(); print(path, "rays", rays)
    tracers[path] = (lib, hs, pt)
a, b = imgs.values()
diff = np.argwhere((a.view(np.uint32) != b.view(np.uint32)).any(2))
print("differing pixels:", len(diff))
lib, hs, _ = tracers["libamber_hip.so"]; api._lib = lib
lst = A.PathTracer(hs, A.Sensor.default(W, H), seed=1, engine=A.ENGINE_LIST)
for (y, x) in diff[:8]:
    px = np.full(spp, y * W + x, np.uint32); sm = np.arange(spp, dtype=np.uint32)
    res = {}
    for path, (l, h_, pt) in tracers.items():
        api._lib = l; res[path] = pt.kat_trace(px, sm, 12)
    api._lib = lib; ref = lst.kat_trace(px, sm, 12)
    for path in res:
        bad = np.nonzero(res[path][1] != ref[1])[0]
        print("pixel", (x, y), path, "samples whose cast count differs from LIST:", bad.tolist())
        for s_ in bad[:2]:
            n = int(max(res[path][1][s_], ref[1][s_]))
            print("   sample", s_, "objects", [int(np.int32(v)) for v in res[path][0][s_, :n, 0]], "LIST", [int(np.int32(v)) for v in ref[0][s_, :n, 0]])
