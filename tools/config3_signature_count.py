import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, time
import os; os.environ.setdefault("AMBER_AMD_LIB", "libamber_hip_lab.so")   # known-answer entry points / lab schedulers: the lab build (include/amber_hip_lab.h)
import amber_amd as A, oracle_binding as O
from amber_amd import scenes
kw = scenes.random_spheres(1_000_000, 7)
hb = A.HostScene.create_arrays(**kw)
t = time.time(); ob = O.Scene.create_arrays(**kw, accel=O.ACCEL_BVH); print("oracle BVH build %.1f s" % (time.time() - t))
pt = A.PathTracer(hb, A.Sensor.default(1920, 1080), seed=1, rows=(536, 540))
sg = pt.render_signatures(0, 16)
t = time.time(); so = ob.path_signatures(1920, 1080, 1, 0, 16, (536, 540), threads=16); print("oracle signatures %.1f s" % (time.time() - t))
d = sg != so
lo = (sg & np.uint64(0xffffffff)) != (so & np.uint64(0xffffffff)); hi = (sg >> np.uint64(32)) != (so >> np.uint64(32))
print("paths", sg.size, "differing", int(d.sum()), "objects differ", int(lo.sum()), "distances differ", int(hi.sum()))
