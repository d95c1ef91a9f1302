"""Distribution of path lengths (casts per path) of config 2: how long can the last paths of a launch keep a wave busy?"""
import sys, os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import os; os.environ.setdefault("AMBER_AMD_LIB", "libamber_hip_lab.so")   # known-answer entry points / lab schedulers: the lab build (include/amber_hip_lab.h)
import amber_amd as A
pt = A.PathTracer(A.HostScene.cornell_box(), A.Sensor.default(1024, 1024), seed=12345)
rng = np.random.default_rng(1)
hist = np.zeros(4096, np.int64)
n = 4_000_000
for rep in range(6):
    px = rng.integers(0, 1024 * 1024, n).astype(np.uint32); sm = rng.integers(0, 1024, n).astype(np.uint32)
    rec, casts = pt.kat_trace(px, sm, 1)
    hist += np.bincount(np.minimum(casts, 4095), minlength=4096)
tot = hist.sum(); c = np.arange(4096)
print("paths %d, mean casts %.3f, max %d" % (tot, (hist * c).sum() / tot, c[hist > 0].max()))
tail = hist[::-1].cumsum()[::-1] / tot
for k in (2, 4, 8, 16, 24, 32, 48, 64, 96, 128, 192, 256):
    print("  P(casts >= %3d) = %.3g" % (k, tail[k]))
