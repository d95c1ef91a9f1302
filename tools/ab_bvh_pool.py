"""Engine BVH on BASELINE config 3 (1M spheres, 1920x1080): round 2's pt_bvh_megakernel (the default scheduler)
against pt_bvh_pool_kernel (AMBER_BVH_POOL=1 at create), in ONE process, interleaved rounds; images and ray counts must be bit-identical.
    python tools/ab_bvh_pool.py [spp] [n_spheres]"""
import os, sys, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import os; os.environ.setdefault("AMBER_AMD_LIB", "libamber_hip_lab.so")   # known-answer entry points / lab schedulers: the lab build (include/amber_hip_lab.h)
import amber_amd as A
from amber_amd import scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
hs = A.HostScene.create_arrays(**scenes.random_spheres(n, 7))
sn = A.Sensor.default(1920, 1080)
os.environ["AMBER_BVH_POOL"] = "0"; legacy = A.PathTracer(hs, sn, seed=1)
os.environ["AMBER_BVH_POOL"] = "1"; pool = A.PathTracer(hs, sn, seed=1)
res, out = {"legacy": [], "pool": []}, {}
for rnd in range(3):
    for name, pt in (("legacy", legacy), ("pool", pool)):
        pt.clear(); pt.render_pass(0, spp); pt.sync(); n0, ms0 = pt.kernel_time()
        img, rays = pt.download()
        out[name] = (img.view(np.uint32).copy(), rays)
        res[name].append(ms0 if rnd == 0 else ms0)   # kernel_time is reset by clear()
        print(name, rnd, "%.1f ms in %d launches, %d rays" % (ms0, n0, rays), flush=True)
same = out["legacy"][1] == out["pool"][1] and np.array_equal(out["legacy"][0], out["pool"][0])
for k, v in res.items():
    print("%-8s median %.1f ms (%s)" % (k, statistics.median(v), " ".join("%.1f" % x for x in v)))
print("bit-identical:", same, "| rays", out["pool"][1], "| Mrays/s pool %.1f" % (out["pool"][1] / statistics.median(res["pool"]) / 1e3))
sys.exit(0 if same else 1)
