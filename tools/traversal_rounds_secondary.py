"""Config 3: wave rounds per ray for the SECONDARY rays of traced paths (exact origins and directions from kat_trace records)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import os; os.environ.setdefault("AMBER_AMD_LIB", "libamber_hip_lab.so")   # known-answer entry points / lab schedulers: the lab build (include/amber_hip_lab.h)
import amber_amd as A
from amber_amd import scenes
n_pix, spp, W, H = 8000, 32, 1920, 1080
hs = A.HostScene.create_arrays(**scenes.random_spheres(1_000_000, 7))
pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=1)
rng = np.random.default_rng(11)
px = np.repeat(rng.integers(0, W * H, n_pix).astype(np.uint32), spp); sm = np.tile(np.arange(spp, dtype=np.uint32), n_pix)
rec, casts = pt.kat_trace(px, sm, 6)
obj = rec[:, :, 0].view(np.int32); pos = rec[:, :, 2:5].view(np.float32)
so, sd = [], []
for k in range(1, 6):
    ok = (obj[:, k - 1] >= 0) & (obj[:, k] >= 0) & (casts > k)
    o = pos[ok, k - 1]; d = pos[ok, k] - o
    ln = np.linalg.norm(d, axis=1, keepdims=True); keep = ln[:, 0] > 1e-6
    so.append(o[keep]); sd.append((d[keep] / ln[keep]).astype(np.float32))
so = np.ascontiguousarray(np.concatenate(so), np.float32); sd = np.ascontiguousarray(np.concatenate(sd), np.float32)
print("secondary rays", len(so), "finite", np.isfinite(so).all(), np.isfinite(sd).all(), "max |o|", np.abs(so).max())
rounds = np.zeros(len(so), np.uint32)
ob, t, ms = pt.kat_traversal_rate(so, sd, waves=5, refill_min=16, repeats=1, rounds=rounds)
r = rounds.astype(np.float64)
print("%.2f ms; rounds mean %.1f p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f max %.0f" % (ms, r.mean(), *np.percentile(r, [50, 90, 99, 99.9, 100])))
for i in np.argsort(rounds)[-8:]:
    print("  rounds %d obj %d t %s o %s d %s |d|^2-1 %.3g" % (rounds[i], ob[i], t[i], so[i], sd[i], float((sd[i].astype(np.float64) ** 2).sum() - 1)))
