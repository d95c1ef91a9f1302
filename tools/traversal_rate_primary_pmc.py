"""Two launches of bvh_trace_rate_kernel for rocprofv3 --pmc: eye rays of config 3 in work-unit order, then the same rays shuffled."""
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
perm = np.random.default_rng(5).permutation(len(org))
waves = int(os.environ.get("WAVES", "5")); refill = int(os.environ.get("REFILL", "16"))
_, _, m0 = pt.kat_traversal_rate(org, dirs, waves=waves, refill_min=refill, repeats=40)
_, _, m1 = pt.kat_traversal_rate(np.ascontiguousarray(org[perm]), np.ascontiguousarray(dirs[perm]), waves=waves, refill_min=refill, repeats=40)
print("ordered %.2f ms, shuffled %.2f ms, %d rays x 40" % (m0, m1, len(org)))
