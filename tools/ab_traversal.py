"""A/B of library builds on engine BVH's traversal alone (bvh_trace_rate_kernel), config 3's scene: eye rays in work-unit order and
shuffled, secondary rays of the same paths; marginal rates from two repeat counts, mean wave rounds per ray; answers checked against
the first library's.   python tools/ab_traversal.py libA.so libB.so ..."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import os; os.environ.setdefault("AMBER_AMD_LIB", "libamber_hip_lab.so")   # known-answer entry points / lab schedulers: the lab build (include/amber_hip_lab.h)
import amber_amd.api as api
import amber_amd as A
from amber_amd import scenes
libs = [a for a in sys.argv[1:] if a.endswith(".so")] or ["libamber_hip.so"]
kw = scenes.random_spheres(1_000_000, 7)
W, H, n_pix, spp = 1920, 1080, 20000, 32
rng = np.random.default_rng(5)
px = np.repeat(rng.integers(0, W * H, n_pix).astype(np.uint32), spp); sm = np.tile(np.arange(spp, dtype=np.uint32), n_pix)
sets, ref = None, {}
for path in libs:
    api._lib = None; api._LIB_PATH = api._ROOT / "lib" / path
    lib = A.load_library()
    hs = A.HostScene.create_arrays(**kw); pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=1)
    if sets is None:
        eye = pt.kat_eye(px, sm)
        org = np.ascontiguousarray(eye[:, 0:3], np.float32); dirs = np.ascontiguousarray(eye[:, 3:6], np.float32)
        rec, casts = pt.kat_trace(px, sm, 6)
        obj = rec[:, :, 0].view(np.int32); pos = rec[:, :, 2:5].view(np.float32)
        so, sd = [], []
        for k in range(1, 6):
            ok = (obj[:, k - 1] >= 0) & (obj[:, k] >= 0) & (casts > k)
            o = pos[ok, k - 1]; d = pos[ok, k] - o
            ln = np.linalg.norm(d, axis=1, keepdims=True); keep = ln[:, 0] > 1e-4
            so.append(o[keep]); sd.append((d[keep] / ln[keep]).astype(np.float32))
        so = np.ascontiguousarray(np.concatenate(so), np.float32); sd = np.ascontiguousarray(np.concatenate(sd), np.float32)
        p1 = rng.permutation(len(org)); p2 = rng.permutation(len(so))
        sets = [("eye rays, work-unit order", org, dirs), ("eye rays, shuffled", np.ascontiguousarray(org[p1]), np.ascontiguousarray(dirs[p1])),
                ("secondary rays, path order", so, sd), ("secondary rays, shuffled", np.ascontiguousarray(so[p2]), np.ascontiguousarray(sd[p2]))]
    for name, o, d in sets:
        rounds = np.zeros(len(o), np.uint32)
        ob, t, m0 = pt.kat_traversal_rate(o, d, waves=5, refill_min=16, repeats=10, rounds=rounds)
        _, _, m1 = pt.kat_traversal_rate(o, d, waves=5, refill_min=16, repeats=40)
        key = name
        if key not in ref: ref[key] = (ob.copy(), t.copy())
        same = np.array_equal(ob, ref[key][0]) and np.array_equal(t[ob >= 0].view(np.uint32), ref[key][1][ob >= 0].view(np.uint32))
        print("%-22s %-28s %7.2f / %7.2f ms -> marginal %7.1f Mrays/s, rounds mean %.2f max %d  %s" % (path, name, m0, m1, len(o) * 30 / (m1 - m0) / 1e3, rounds.mean(), rounds.max(), "same answers" if same else "MISMATCH"), flush=True)
    pt.close(); hs.close()
os._exit(0)
