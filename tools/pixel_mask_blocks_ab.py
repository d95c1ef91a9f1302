"""Does the block size of the per-pixel candidate masks (AMBER_PIXEL_MASK_BLOCK = 1 / 2 / 4) cost the headline kernel anything?  Config 2 at its full
1024 spp, three handles in one process, interleaved rounds; kernel ms of the 1016-spp launch."""
import os, sys, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import amber_amd as A
hs = A.HostScene.cornell_box(); sn = A.Sensor.default(1024, 1024)
pts = {}
for block in ("1", "2", "4"):
    os.environ["AMBER_PIXEL_MASK_BLOCK"] = block
    pts[block] = A.PathTracer(hs, sn, seed=12345)
    pts[block].render_pass(0, 8); pts[block].sync()
res = {b: [] for b in pts}
for rnd in range(6):
    for b, pt in pts.items():
        pt.clear(); pt.render_pass(0, 1024); pt.sync(); res[b].append(pt.kernel_time()[1])
for b in pts:
    print("block %s: median %.3f ms  (%s)  rays %d" % (b, statistics.median(res[b]), " ".join("%.2f" % x for x in res[b]), pts[b].ray_count()), flush=True)
