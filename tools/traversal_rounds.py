"""Config 3: distribution of the wave rounds a ray is in flight for in bvh_trace_rate_kernel, for eye rays in work-unit order with
lock-step waves (refill 64) and decoupled lanes (refill 16), identical rays (each ray 64 times), and tiny launches (fixed cost)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import os; os.environ.setdefault("AMBER_AMD_LIB", "libamber_hip_lab.so")   # known-answer entry points / lab schedulers: the lab build (include/amber_hip_lab.h)
import amber_amd as A
from amber_amd import scenes
n_pix, spp, W, H = 20000, 64, 1920, 1080
hs = A.HostScene.create_arrays(**scenes.random_spheres(1_000_000, 7))
pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=1)
first = (H // 2) * W + 100
px = np.repeat(np.arange(first, first + n_pix, dtype=np.uint32), spp); sm = np.tile(np.arange(spp, dtype=np.uint32), n_pix)
eye = pt.kat_eye(px, sm)
org = np.ascontiguousarray(eye[:, 0:3], np.float32); dirs = np.ascontiguousarray(eye[:, 3:6], np.float32)

def stats(name, o, d, waves, refill, repeats=1):
    rounds = np.zeros(len(o), np.uint32)
    obj, t, ms = pt.kat_traversal_rate(o, d, waves=waves, refill_min=refill, repeats=repeats, rounds=rounds)
    r = rounds.astype(np.float64)
    q = np.percentile(r, [50, 90, 99, 99.9, 100])
    line = "%-40s %d waves refill %2d x%d: %8.2f ms  rounds mean %.1f  p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f max %.0f" % (name, waves, refill, repeats, ms, r.mean(), *q)
    if len(o) % 64 == 0:
        g = r.reshape(-1, 64); line += "  | per 64 in order: mean of max %.1f" % g.max(axis=1).mean()
    print(line, flush=True)
    return rounds, obj

for n in (64, 1024, 16384):
    stats("first %d eye rays" % n, org[:n].copy(), dirs[:n].copy(), 5, 16)
miss = None
for refill in (64, 16, 1):
    rounds, obj = stats("eye rays, work-unit order", org, dirs, 5, refill)
    if refill == 64:
        print("   hit share %.3f; rounds of hits mean %.1f, of misses mean %.1f" % ((obj >= 0).mean(), rounds[obj >= 0].mean(), rounds[obj < 0].mean() if (obj < 0).any() else 0))
        worst = np.argsort(rounds)[-5:]
        for i in worst: print("   slow ray %d: rounds %d obj %d o %s d %s" % (i, rounds[i], obj[i], org[i], dirs[i]))
rep = np.repeat(np.arange(0, len(org), 64), 64)
stats("each pixel's sample 0, 64 times", np.ascontiguousarray(org[rep]), np.ascontiguousarray(dirs[rep]), 5, 64)
stats("each pixel's sample 0, 64 times", np.ascontiguousarray(org[rep]), np.ascontiguousarray(dirs[rep]), 5, 64, repeats=20)
stats("eye rays, work-unit order", org, dirs, 5, 64, repeats=20)
stats("eye rays, work-unit order", org, dirs, 8, 64, repeats=20)
