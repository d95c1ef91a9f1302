"""BASELINE config 2 at FULL size against the oracle: Cornell 1024x1024 @ 1024 spp, every pixel, every sample (2.25e9 rays; the oracle
needs about half a minute on 16 threads).  Image bits and ray count; the oracle's List acceleration (the semantics the engine
implements) and, optionally, its restatement of the reference BVH.   python tools/full_frame_parity.py [spp] [--bvh]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import amber_amd as A
import oracle_binding as O
spp = next((int(a) for a in sys.argv[1:] if a.isdigit()), 1024)
W = H = 1024; seed = 12345
threads = min(16, os.cpu_count() or 1)
pt = A.PathTracer(A.HostScene.cornell_box(), A.Sensor.default(W, H), seed=seed)
t = time.time(); pt.render_pass(0, spp); img, rays = pt.download(); print("GPU: %d rays in %.2f s (incl. first-launch probe and download)" % (rays, time.time() - t), flush=True)
for accel, name in ((O.ACCEL_LIST, "List"),) + (((O.ACCEL_BVH, "reference BVH"),) if "--bvh" in sys.argv else ()):
    osc = O.Scene.cornell(accel)
    t = time.time(); ref, cnt = osc.render_xorshift(W, H, seed, 0, spp, math=O.MATH_LIBM, threads=threads); dt = time.time() - t
    diff = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
    print("oracle(XorShift, %s, live libm, %d threads): %d rays in %.1f s (%.1f Mrays/s); pixels differing %d of %d; ray count delta %d; lit pixels %d"
          % (name, threads, cnt.casts, dt, cnt.casts / dt / 1e6, int(diff.sum()), W * H, int(rays) - int(cnt.casts), int((ref > 0).any(axis=2).sum())), flush=True)
