"""Debug helper for tools/fuzz_engines.py: for one seed, find the paths on which two engines disagree and print them
next to the oracle's trace.   python tools/fuzz_debug.py <seed> <engine_a> <engine_b>"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import amber_amd as A
import oracle_binding as O
import importlib.util
seed, ea, eb = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 1, int(sys.argv[3]) if len(sys.argv) > 3 else 3
sys.path.insert(0, os.path.join(R, "tests")); from fuzz_scenes import scene_for_seed
sc, rng = scene_for_seed(seed, scaled="--scaled" in sys.argv, extreme="--extreme" in sys.argv)
W, H, spp = (128, 96, 12) if "--heavy" in sys.argv else (48, 40, 6)
hs = A.HostScene.create(**sc); osc = O.Scene.create(**sc)
px = np.repeat(np.arange(W * H, dtype=np.uint32), spp); sm = np.tile(np.arange(spp, dtype=np.uint32), W * H)
tr = {}
for e in (ea, eb):
    pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=seed, engine=e)
    tr[e] = pt.kat_trace(px, sm, 24); pt.close()
(ra, ca), (rb, cb) = tr[ea], tr[eb]
bad = np.nonzero(ca != cb)[0]
print("paths with different cast counts:", len(bad), "objects:", len(sc["objects"]))
for i in bad[:4]:
    n, orec, eye = osc.trace(W, H, seed, int(px[i] % W), int(px[i] // W), int(sm[i]), max_bounces=24)
    print("pixel", px[i], "sample", sm[i], "casts", ea, ca[i], eb, cb[i], "oracle", n)
    for b in range(max(ca[i], cb[i])):
        def rec(r, c):
            return ("obj %d t %.9g" % (np.int32(r[i, b, 0]), r[i, b, 1:2].view(np.float32)[0])) if b < c[i] else "-"
        o = ("obj %d t %.9g pos %s" % (orec[b].object, orec[b].t, [round(x, 6) for x in orec[b].pos])) if b < n else "-"
        print("   bounce", b, "|", rec(ra, ca), "|", rec(rb, cb), "| oracle", o)
    k = min(ca[i], cb[i])

# closest-hit comparison of the two engines on rays from the first disagreeing bounce's origin
if len(bad):
    i = bad[0]
    b0 = 0
    while b0 < min(ca[i], cb[i]) and ra[i, b0, 0] == rb[i, b0, 0]:
        b0 += 1
    o = ra[i, b0 - 1, 2:5].view(np.float32) if b0 > 0 else None
    if o is not None:
        rng = np.random.default_rng(1)
        n = 200000
        d = rng.normal(size=(n, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        d[: n // 2] *= rng.uniform(0.2, 3.0, (n // 2, 1)).astype(np.float32)          # non-unit directions too
        org = np.tile(o, (n, 1)).astype(np.float32)
        res = {}
        for e in (ea, eb):
            pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=seed, engine=e)
            res[e] = pt.kat_cast(org, d); pt.close()
        diff = np.nonzero(res[ea][0] != res[eb][0])[0]
        print("kat_cast from", o, ":", len(diff), "of", n, "rays differ")
        for k in diff[:5]:
            print("   d", d[k], "engine", ea, "obj", res[ea][0][k], "t", res[ea][1][k], "| engine", eb, "obj", res[eb][0][k], "t", res[eb][1][k])
