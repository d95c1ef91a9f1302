"""Bisects (rows, samples) of config 3 at 480x270 for the path that makes a launch 40x slower, then prints its trace."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import os; os.environ.setdefault("AMBER_AMD_LIB", "libamber_hip_lab.so")   # known-answer entry points / lab schedulers: the lab build (include/amber_hip_lab.h)
import amber_amd as A
from amber_amd import scenes
kw = scenes.random_spheres(1_000_000, 7)
hs = A.HostScene.create_arrays(**kw)
W, H = 480, 270
sn = A.Sensor.default(W, H)

def t_of(rows, first, n):
    pt = A.PathTracer(hs, sn, seed=1, rows=rows)
    pt.render_pass(first, n); pt.sync(); _, ms = pt.kernel_time(); pt.close()
    return ms

r0, r1, s0, s1 = 0, H, 128, 256
print("all", t_of((r0, r1), s0, s1 - s0), flush=True)
while r1 - r0 > 1:
    m = (r0 + r1) // 2
    ta = t_of((r0, m), s0, s1 - s0)
    if ta > 300: r1 = m
    else: r0 = m
    print("rows", r0, r1, "%.0f" % ta, flush=True)
while s1 - s0 > 1:
    m = (s0 + s1) // 2
    ta = t_of((r0, r1), s0, m - s0)
    if ta > 300: s1 = m
    else: s0 = m
    print("samples", s0, s1, "%.0f" % ta, flush=True)
pt = A.PathTracer(hs, sn, seed=1)
worst = (0, -1)
for x in range(W):
    px = np.array([x + r0 * W], np.uint32); sm = np.array([s0], np.uint32)
    t = time.time(); rec, casts = pt.kat_trace(px, sm, 64); dt = time.time() - t
    if dt > worst[0]: worst = (dt, x, rec, casts)
dt, x, rec, casts = worst
print("slow path: pixel x=%d y=%d sample=%d: kat_trace %.3f s, casts %d" % (x, r0, s0, dt, casts[0]))
for b in range(min(int(casts[0]), 64)):
    r = rec[0, b]
    print(b, "obj", np.int32(r[0]), "t", r[1:2].view(np.float32)[0], "pos", r[2:5].view(np.float32), "w", r[5:8].view(np.float32), "meas", r[8:11].view(np.float32))
