"""Where do the per-pixel masks disagree with Phase A?  Renders a fuzz scene with the two-phase engine, masks on and off,
and prints the pixels that differ.    python tools/pixel_mask_debug.py seed [scaled] [extreme]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import amber_amd as A
from fuzz_scenes import scene_for_seed
seed = int(sys.argv[1]); scaled = len(sys.argv) > 2 and sys.argv[2] == "1"; extreme = len(sys.argv) > 3 and sys.argv[3] == "1"
sc, _ = scene_for_seed(seed, scaled=scaled, extreme=extreme)
hs = A.HostScene.create(**sc)
W, H, spp = 48, 40, 6
res = {}
for m in ("0", "1"):
    os.environ["AMBER_PIXEL_MASK"] = m
    pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=seed, engine=A.ENGINE_TWO_PHASE)
    pt.render_pass(0, spp); res[m] = pt.download()
    sig = pt.render_signatures(0, spp); res["s" + m] = sig
    pt.close()
print("rays", res["0"][1], res["1"][1])
d = np.argwhere((res["0"][0].view(np.uint32) != res["1"][0].view(np.uint32)).any(axis=2))
print("differing pixels", d.tolist()[:20])
ds = np.argwhere(res["s0"] != res["s1"])
print("paths whose signatures differ:", len(ds), ds[:10].tolist())
pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=seed, engine=A.ENGINE_LIST)
for y, x, k in ds[:6]:
    rec, casts = pt.kat_trace(np.array([y * W + x], np.uint32), np.array([k], np.uint32), 4)
    print("  pixel (%d,%d) sample %d: first hits (object, t):" % (x, y, k), [(int(np.int32(r[0])), float(r[1:2].view(np.float32)[0])) for r in rec[0][: int(casts[0])][:3]], "kind/params of first:", sc["objects"][int(np.int32(rec[0][0][0]))] if np.int32(rec[0][0][0]) >= 0 and np.int32(rec[0][0][0]) < len(sc["objects"]) else "blade/none")
