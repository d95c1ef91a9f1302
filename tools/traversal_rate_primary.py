"""Config 3 (1M spheres): how fast would a PRIMARY-ray pass of its own be?  Eye rays in work-unit order (all samples of a pixel
next to each other, pixels consecutive: what a wave of such a pass would hold) against the same rays shuffled and against the
secondary rays of the same paths, through bvh_trace_rate_kernel; marginal rates from two repeat counts (a launch has a fixed
tail of slow rays).   python tools/traversal_rate_primary.py [pixels] [spp]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import os; os.environ.setdefault("AMBER_AMD_LIB", "libamber_hip_lab.so")   # known-answer entry points / lab schedulers: the lab build (include/amber_hip_lab.h)
import amber_amd as A
from amber_amd import scenes
n_pix = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W, H = 1920, 1080
hs = A.HostScene.create_arrays(**scenes.random_spheres(1_000_000, 7))
pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=1)
rng = np.random.default_rng(5)
first = (H // 2) * W + 100
px = np.repeat(np.arange(first, first + n_pix, dtype=np.uint32), spp)
sm = np.tile(np.arange(spp, dtype=np.uint32), n_pix)
eye = pt.kat_eye(px, sm)
org = np.ascontiguousarray(eye[:, 0:3], np.float32); dirs = np.ascontiguousarray(eye[:, 3:6], np.float32)
rec, casts = pt.kat_trace(px, sm, 6)
obj = rec[:, :, 0].view(np.int32); pos = rec[:, :, 2:5].view(np.float32)
so, sd = [], []
for k in range(1, 6):
    ok = (obj[:, k - 1] >= 0) & (obj[:, k] >= 0) & (casts > k)
    o = pos[ok, k - 1]; d = pos[ok, k] - o
    ln = np.linalg.norm(d, axis=1, keepdims=True); keep = ln[:, 0] > 1e-6
    so.append(o[keep]); sd.append((d[keep] / ln[keep]).astype(np.float32))
so = np.ascontiguousarray(np.concatenate(so), np.float32); sd = np.ascontiguousarray(np.concatenate(sd), np.float32)
print("eye rays %d (hit share %.3f), secondary rays with a hit %d, casts per path %.3f" % (len(org), (obj[:, 0] >= 0).mean(), len(so), casts.mean()), flush=True)

def marginal(name, o, d, waves, refill, r0=10, r1=40):
    _, _, m0 = pt.kat_traversal_rate(o, d, waves=waves, refill_min=refill, repeats=r0)
    _, _, m1 = pt.kat_traversal_rate(o, d, waves=waves, refill_min=refill, repeats=r1)
    print("  %-34s %d waves refill %2d: %7.2f / %7.2f ms -> marginal %8.1f Mrays/s" % (name, waves, refill, m0, m1, len(o) * (r1 - r0) / (m1 - m0) / 1e3), flush=True)

perm = rng.permutation(len(org))
ps = rng.permutation(len(so))
for waves in (5, 8):
    for refill in (16, 64):
        marginal("eye rays, work-unit order", org, dirs, waves, refill)
    marginal("eye rays, shuffled", np.ascontiguousarray(org[perm]), np.ascontiguousarray(dirs[perm]), waves, 16)
    marginal("secondary rays, path order", so, sd, waves, 16)
    marginal("secondary rays, shuffled", np.ascontiguousarray(so[ps]), np.ascontiguousarray(sd[ps]), waves, 16)
