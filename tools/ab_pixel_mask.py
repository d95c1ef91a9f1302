"""pt_megakernel with and without the per-pixel candidate masks of the primary rays (AMBER_PIXEL_MASK=0 at create), one process,
interleaved rounds; images and ray counts must be bit-identical.   python tools/ab_pixel_mask.py [spp] [width]"""
import os, sys, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import amber_amd as A
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 512
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
sc = A.HostScene.cornell_box(); sn = A.Sensor.default(W, W)
os.environ["AMBER_PIXEL_MASK"] = "0"; off = A.PathTracer(sc, sn)
os.environ["AMBER_PIXEL_MASK"] = "1"; on = A.PathTracer(sc, sn)
res, out = {"off": [], "on": []}, {}
for rnd in range(5):
    for name, pt in (("off", off), ("on", on)):
        pt.clear(); pt.render_pass(0, spp); pt.sync(); n, ms = pt.kernel_time()
        res[name].append(ms)
        img, rays = pt.download(); out[name] = (img.view(np.uint32).copy(), rays)
same = out["on"][1] == out["off"][1] and np.array_equal(out["on"][0], out["off"][0])
for k, v in res.items():
    print("masks %-3s median %.2f ms (%s)" % (k, statistics.median(v), " ".join("%.1f" % x for x in v)))
print("bit-identical:", same, "| rays", out["on"][1])
sys.exit(0 if same else 1)
