"""Probe one eye ray of a fuzz scene: exact (o, d) from kat_eye, closest hit by engines LIST and BVH, the object's record."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import amber_amd as A
from fuzz_scenes import scene_for_seed
seed, pixel, sample = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sc, _ = scene_for_seed(seed, scaled="--scaled" in sys.argv, extreme="--extreme" in sys.argv)
W, H = (128, 96) if "--heavy" in sys.argv else (48, 40)
hs = A.HostScene.create(**sc)
res = {}
for e in (1, 3):
    pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=seed, engine=e)
    eye = pt.kat_eye(np.array([pixel], np.uint32), np.array([sample], np.uint32))[0]
    o, d = eye[:3].copy(), eye[3:6].copy()
    res[e] = pt.kat_cast(o[None, :], d[None, :]); pt.close()
    print("engine", e, "o", [repr(float(x)) for x in o], "d", [repr(float(x)) for x in d], "|d|^2-1 = %.3e" % (float(np.dot(d.astype(np.float64), d.astype(np.float64))) - 1.0),
          "-> obj", res[e][0][0], "t", res[e][1][0])
nb = max(1, sc["n_blades"])
idx = int(res[1][0][0])
if idx >= nb:
    k, m, p = sc["objects"][idx - nb]
    print("object", idx, "kind", k, "material", sc["materials"][m], "params", [repr(float(np.float32(x))) for x in p])
print("n objects", len(sc["objects"]) + nb, "transform", sc["transform"], "lens", {k: sc[k] for k in ("focal_length", "focus_distance", "radius", "n_blades")})
