"""Creates, uses and destroys many handles (both engines, signatures, light tracing) and checks that device memory comes back."""
import sys, os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import os; os.environ.setdefault("AMBER_AMD_LIB", "libamber_hip_lab.so")   # known-answer entry points / lab schedulers: the lab build (include/amber_hip_lab.h)
import amber_amd as A
from amber_amd import scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
box = A.HostScene.cornell_box()
sph = A.HostScene.create_arrays(**scenes.random_spheres(20000, 3))
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
pt = A.PathTracer(box, A.Sensor.default(64, 64)); pt.render_pass(0, 8); pt.download(); pt.close()      # library warm-up
f0 = free(); ref = None
for i in range(n):
    hs, flags = (box, 0) if i % 3 == 0 else (sph, A.api.PT_FLAG_BVH_POOL if i % 3 == 2 else 0)
    pt = A.PathTracer(hs, A.Sensor.default(96, 64), seed=7, flags=flags)
    pt.render_pass(0, 16); pt.render_pass(16, 5)
    img, rays = pt.download()
    if i % 5 == 0: pt.render_signatures(0, 4)
    if i % 7 == 0: pt.lt_trace(0, 1)
    key = (i % 3 != 0, rays, img.tobytes())
    if i < 3: ref = ref or {}; ref[i % 3] = key
    else: assert key == ref[i % 3] or (i % 3 == 2 and key[1:] == ref[1][1:]), i
    pt.close()
f1 = free()
print("handles %d; free device memory before %.1f MB, after %.1f MB (delta %.1f MB)" % (n, f0 / 2**20, f1 / 2**20, (f0 - f1) / 2**20))
assert f0 - f1 < 64 * 2**20
print("churn ok")
