"""pt vs lt on sub-scenes (tests/test_integrators_converge.py): which objects make the two integrators disagree?
   python tools/convergence_probe.py  (GPU box).  Findings are recorded in DESIGN.md section 11."""
import sys, os, copy; R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np, amber_amd as amber
from test_integrators_converge import SCENE, W, H
np.set_printoptions(precision=3, linewidth=220)
sensor = amber.Sensor.default(W, H)
def blocks(a): return a[:, 6:18].reshape(a.shape[0], 4, 3, 6, 4, 3).sum(axis=(2, 4, 5))
def run2(name, objs, radius=None, K=8, show=()):
    SC = copy.deepcopy(SCENE); SC["objects"] = objs
    if radius: SC["radius"] = radius
    hs = amber.HostScene.create(**SC)
    pt = np.stack([hs.render(sensor, 65536, seed=1000 + k, samples_per_launch=65536, algorithm="pt")[0] for k in range(K)])
    lt = np.stack([hs.render(sensor, 200000, seed=2000 + k, samples_per_launch=20000, algorithm="lt")[0] for k in range(K)])
    bp, bl = blocks(pt), blocks(lt)
    print("==", name, "sum ratio %.4f" % (lt.mean(0).sum() / pt.mean(0).sum()))
    print(" ratio blocks lt/pt\n", bl.mean(0) / np.maximum(bp.mean(0), 1e-30))
    for (i, j) in show:
        print("  block", (i, j), "pt batches", bp[:, i, j] * 1e5, "\n              lt batches", bl[:, i, j] * 1e5)
O = SCENE["objects"]
floor = [O[4], O[5]]
run2("floor + disk light + diffuse sphere", [O[0]] + floor + [(1, 1, [-0.6, -0.55, -0.8, 0.45])], show=[(1, 3), (1, 4), (2, 3)])
run2("disk light + diffuse sphere, no floor", [O[0]] + [(1, 1, [-0.6, -0.55, -0.8, 0.45])], show=[(1, 3), (1, 4), (2, 3)])
run2("same, max aperture 0.15", [O[0]] + [(1, 1, [-0.6, -0.55, -0.8, 0.45])], radius=0.15, show=[(1, 3), (1, 4), (2, 3)])
