"""What sets the fixed cost of a pt_megakernel launch?  Two measurements on config 2's frame:
 (1) kernel time = fixed + slope * spp for several path-length caps (max_depth): if the fixed part shrinks with the cap, the tail of a
     launch is the critical path of its longest paths (total-internal-reflection chains survive Russian roulette with p = 0.9375 per
     bounce), not a scheduling loss;
 (2) the latency of one wave iteration (one bounce of <= 64 rays) when few waves run: a launch of 8 claims (8 waves, one per claim,
     each alone on its SIMD) against the same paths spread over the whole frame at full occupancy.
python tools/launch_tail_floor.py"""
import sys, os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import amber_amd as A
sc = A.HostScene.cornell_box()
def fit(max_depth, rows=None):
    pt = A.PathTracer(sc, A.Sensor.default(1024, 1024), max_depth=max_depth, rows=rows)
    pt.render_pass(0, 8); pt.sync(); pt.clear()
    xs, ys, rays = [], [], []
    for spp in (32, 64, 128, 256, 512):
        best = 1e9
        for rep in range(3):
            k0, m0 = pt.kernel_time(); r0 = pt.ray_count(); pt.render_pass(0, spp); pt.sync(); k1, m1 = pt.kernel_time(); r1 = pt.ray_count(); best = min(best, m1 - m0)
        xs.append(spp); ys.append(best); rays.append(r1 - r0)
    b, a = np.polyfit(xs, ys, 1)
    pt.close()
    return a, b, rays[-1] / (1024 * (rows[1] - rows[0] if rows else 1024) * 512)
print("(1) fixed part of a launch against the path-length cap (config 2's frame, 32 .. 512 spp, best of 3):")
for md in (0, 64, 32, 16, 8, 4, 2):
    a, b, rpp = fit(md)
    print("  max_depth %3s: %.3f ms fixed + %.5f ms per spp   (%.3f rays per path)" % (md if md else "none", a, b, rpp), flush=True)
print("(2) one wave iteration, alone and in a crowd:")
pt = A.PathTracer(sc, A.Sensor.default(1024, 1024), rows=(600, 601))           # 1024 pixels: at 1024 spp per pixel a claim is one pixel
pt.render_pass(0, 8); pt.sync(); pt.clear()
for spp, note in ((8, "8192 paths = 8 claims: 8 waves, each alone on its SIMD"), (64, "64 claims"), (1024, "1024 claims: one per pixel, 1024 waves of 6144")):
    best = 1e9
    for rep in range(5):
        k0, m0 = pt.kernel_time(); r0 = pt.ray_count(); pt.render_pass(0, spp); pt.sync(); k1, m1 = pt.kernel_time(); r1 = pt.ray_count(); best = min(best, m1 - m0)
    paths = 1024 * spp; rays = r1 - r0
    claims = (paths + 1023) // 1024
    iters = rays / min(claims, 6144) / 64 / 0.75                                     # wave iterations of the busiest wave at the kernel's mean lane utilisation
    print("  %5d spp on one row (%s): %.3f ms, %d rays; ~%.0f iterations per wave -> %.2f us per iteration" % (spp, note, best, rays, iters, best * 1e3 / iters), flush=True)
os._exit(0)
