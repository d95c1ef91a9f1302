"""BASELINE config 3 at FULL size (1M spheres, 1920x1080 @ 256 spp, 1.37e9 rays) against the oracle's restatement of the REFERENCE BVH
(the List scan the engine's semantics follow is O(n) per ray: not possible at this size).  Expected: identical except on the paths where
the reference's binary32 sphere test accepts a ray that misses the sphere's geometric box -- hits its BVH loses and its List keeps
(DESIGN.md section 5).   python tools/config3_full_parity.py [spp]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import amber_amd as A
import oracle_binding as O
from amber_amd import scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W, H, seed = 1920, 1080, 1
kw = scenes.random_spheres(1_000_000, 7)
pt = A.PathTracer(A.HostScene.create_arrays(**kw), A.Sensor.default(W, H), seed=seed)
pt.render_pass(0, spp); img, rays = pt.download()
print("GPU (engine BVH, List semantics): %d rays" % rays, flush=True)
t = time.time(); osc = O.Scene.create_arrays(**kw, accel=O.ACCEL_BVH); tb = time.time() - t
t = time.time(); ref, cnt = osc.render_xorshift(W, H, seed, 0, spp, math=O.MATH_LIBM, threads=min(16, os.cpu_count() or 1)); dt = time.time() - t
diff = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
err = np.sqrt(((img.astype(np.float64) - ref) ** 2).sum(axis=2)); mag = np.sqrt((ref.astype(np.float64) ** 2).sum(axis=2))
rel = np.where(mag > 0, err / np.where(mag > 0, mag, 1), 0)
print("oracle(XorShift, reference BVH, live libm): build %.1f s, %d rays in %.1f s (%.1f Mrays/s)" % (tb, cnt.casts, dt, cnt.casts / dt / 1e6))
print("pixels differing %d of %d (%.2e); over the 1e-4 relative-L2 tolerance %d; ray count delta %d (%.1e of the rays); lit pixels %d"
      % (int(diff.sum()), W * H, diff.mean(), int((rel > 1e-4).sum()), int(rays) - int(cnt.casts), abs(int(rays) - int(cnt.casts)) / rays, int((ref > 0).any(axis=2).sum())))
