"""How fast is engine BVH's traversal on its own?  (EXPERIMENTS.md, engine BVH.)  Builds a realistic ray set of BASELINE config 3 (1M
spheres) -- exact eye rays plus the secondary rays of traced paths (origin = the previous hit point, direction towards the next) --
and runs them through bvh_trace_rate_kernel: the render kernels' resumable traversal in a kernel that does nothing else, at 4, 5,
6 and 8 waves per SIMD and several refill thresholds.  Checks every answer against amber_hip_kat_cast.
    python tools/traversal_rate.py [n_paths] [repeats]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import os; os.environ.setdefault("AMBER_AMD_LIB", "libamber_hip_lab.so")   # known-answer entry points / lab schedulers: the lab build (include/amber_hip_lab.h)
import amber_amd as A
from amber_amd import scenes
n_paths = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 12
W, H = 1920, 1080
hs = A.HostScene.create_arrays(**scenes.random_spheres(1_000_000, 7))
pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=1)
rng = np.random.default_rng(3)
# paths of random pixels (a set of consecutive pixels makes every wave walk the same nodes at the same time: 0.5 Grays/s)
px = rng.integers(0, W * H, n_paths).astype(np.uint32)
sm = rng.integers(0, 256, n_paths).astype(np.uint32)
eye = pt.kat_eye(px, sm)                                   # (n, 7): origin, direction, weight
maxb = 8
rec, casts = pt.kat_trace(px, sm, maxb)                    # (n, maxb, 11): object, t, pos[3], ...
obj = rec[:, :, 0].view(np.int32)
pos = rec[:, :, 2:5].view(np.float32)
org, dirs = [eye[:, 0:3]], [eye[:, 3:6]]
for k in range(1, maxb):
    ok = (obj[:, k - 1] >= 0) & (obj[:, k] >= 0) & (casts > k)
    o = pos[ok, k - 1]; d = pos[ok, k] - o
    ln = np.linalg.norm(d, axis=1, keepdims=True)
    keep = ln[:, 0] > 1e-6
    org.append(o[keep]); dirs.append((d[keep] / ln[keep]).astype(np.float32))
org = np.ascontiguousarray(np.concatenate(org), np.float32); dirs = np.ascontiguousarray(np.concatenate(dirs), np.float32)
perm = rng.permutation(len(org)); org, dirs = np.ascontiguousarray(org[perm]), np.ascontiguousarray(dirs[perm])   # mixed bounce depths per wave, like the render
n = len(org)
print("ray set: %d rays (%d eye rays + %d secondary rays of %d paths, shuffled), walked %d times per launch = %.1f M rays" % (n, n_paths, n - n_paths, n_paths, repeats, n * repeats / 1e6))
ref_obj, ref_t, _, _ = pt.kat_cast(org, dirs)
best = None
for waves in (4, 5, 6, 8):
    for refill in (1, 8, 16, 32, 64):
        o2, t2, ms = pt.kat_traversal_rate(org, dirs, waves=waves, refill_min=refill, repeats=repeats)
        same = np.array_equal(o2, ref_obj) and np.array_equal(t2[ref_obj >= 0].view(np.uint32), ref_t[ref_obj >= 0].view(np.uint32))
        rate = n * repeats / ms / 1e3
        print("  %d waves/SIMD, refill at %2d idle lanes: %7.2f ms  %8.1f Mrays/s  %s" % (waves, refill, ms, rate, "answers == kat_cast" if same else "MISMATCH"), flush=True)
        if not same: sys.exit(1)
        if best is None or rate > best[0]: best = (rate, waves, refill)
print("best: %.1f Mrays/s at %d waves/SIMD, refill %d  (the render kernel, which also shades: ~5 800 Mrays/s on this scene)" % best)
