"""Per-pixel candidate masks of the primary rays computed per BLOCK of pixels (AMBER_PIXEL_MASK_BLOCK = 1 / 2 / 4): duration of pixel_mask_kernel,
candidates per pixel, and what the coarser masks cost the render (config 2 at 256 spp, kernel ms)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import os; os.environ.setdefault("AMBER_AMD_LIB", "libamber_hip_lab.so")   # known-answer entry points / lab schedulers: the lab build (include/amber_hip_lab.h)
import amber_amd as A
hs = A.HostScene.cornell_box(); sn = A.Sensor.default(1024, 1024)
for block in ("1", "2", "4"):
    os.environ["AMBER_PIXEL_MASK_BLOCK"] = block
    pt = A.PathTracer(hs, sn, seed=12345)
    masks, ms = pt.pixel_masks()
    cand = np.unpackbits(masks.view(np.uint8)).sum() / masks.size
    res = []
    for _ in range(3):
        pt.clear(); pt.render_pass(0, 8); pt.render_pass(8, 248); pt.sync(); res.append(pt.kernel_time()[1])
    print("block %s: pixel_mask_kernel %.3f ms, %.3f candidates per pixel; 256 spp: kernel %s ms, %d rays" % (block, ms, cand, " ".join("%.2f" % x for x in res), pt.ray_count()), flush=True)
    pt.close()
