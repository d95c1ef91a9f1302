"""BASELINE configs 2, 4 and 5 at FULL size on one MI355X (the multi-GPU configs rendered by a single rank): wall time,
kernel time, rays, Mrays/s and image statistics.  Config 5 uses the build-side max depth 16."""
import os, sys, time, hashlib
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import amber_amd as A
hs = A.HostScene.cornell_box()
for name, (W, H, spp, depth) in {"config 2": (1024, 1024, 1024, 0), "config 4": (2048, 2048, 4096, 0), "config 5": (3840, 2160, 8192, 16)}.items():
    pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=12345, max_depth=depth)
    pt.render_pass(0, 8); pt.sync(); pt.clear()
    t = time.time()
    first = 0
    while first < spp:
        n = min(1024, spp - first)
        pt.render_pass(first, n); first += n
    pt.sync(); wall = time.time() - t
    nl, ms = pt.kernel_time()
    img, rays = pt.download()
    img = img / np.float32(spp)
    print("%s: %dx%d @ %d spp%s: wall %.3f s, kernel %.3f s in %d launches, %d rays (%.3f per path), %.1f Mrays/s; image mean %.6g, finite %s, "
          "non-zero pixels %.4f, sha1 %s" % (name, W, H, spp, (", max depth %d" % depth) if depth else "", wall, ms * 1e-3, nl, rays, rays / (W * H * spp),
                                            rays / wall / 1e6, float(img.mean()), bool(np.isfinite(img).all()), float((img.sum(axis=2) > 0).mean()),
                                            hashlib.sha1(img.tobytes()).hexdigest()[:16]), flush=True)
    pt.close()
