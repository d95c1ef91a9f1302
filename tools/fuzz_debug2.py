"""Which path differs between two engines on a fuzz scene?   python tools/fuzz_debug2.py seed scaled extreme"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import amber_amd as A
from fuzz_scenes import scene_for_seed
seed = int(sys.argv[1]); scaled = sys.argv[2] == "1"; extreme = sys.argv[3] == "1"
sc, _ = scene_for_seed(seed, scaled=scaled, extreme=extreme)
hs = A.HostScene.create(**sc)
W, H, spp = 48, 40, 6
print("library", os.path.basename(str(A.library_path())), "objects", len(sc["objects"]), "blades", sc["n_blades"])
res = {}
for e, name in ((A.ENGINE_LIST, "list"), (A.ENGINE_BVH, "bvh")):
    pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=seed, engine=e)
    pt.render_pass(0, spp); img, rays = pt.download()
    res[name] = (img, rays, pt.kat_signatures(0, spp), pt)
print("rays", res["list"][1], res["bvh"][1], "differing pixels", np.argwhere((res["list"][0].view(np.uint32) != res["bvh"][0].view(np.uint32)).any(axis=2)).tolist())
ds = np.argwhere(res["list"][2] != res["bvh"][2])
print("KAT-kernel signatures differ on", len(ds), "paths", ds[:5].tolist())
for y, x, k in ds[:3]:
    for name in ("list", "bvh"):
        rec, casts = res[name][3].kat_trace(np.array([y * W + x], np.uint32), np.array([k], np.uint32), 8)
        print("  ", name, "pixel (%d,%d) sample %d:" % (x, y, k), [(int(np.int32(r[0])), "%.9g" % float(r[1:2].view(np.float32)[0])) for r in rec[0][: int(casts[0])]])
