"""Light-tracing mismatch helper: ray counts of every engine and the oracle for one fuzz seed, pass by pass."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import amber_amd as A
import oracle_binding as O
from fuzz_scenes import scene_for_seed
seed = int(sys.argv[1]); depth = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sc, rng = scene_for_seed(seed, scaled="--scaled" in sys.argv, extreme="--extreme" in sys.argv)
W, H = 48, 40
hs = A.HostScene.create(**sc); osc = O.Scene.create(**sc)
for i, o in enumerate(sc["objects"]):
    print(i + max(1, sc["n_blades"]), "kind", o[0], "mat", sc["materials"][o[1]][0], [round(x, 4) for x in o[2]])
for s in range(3):
    row = []
    for e in (1, 2, 3):
        try:
            pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=seed, engine=e, max_depth=depth)
            rec, rays = pt.lt_trace(s, 1, capacity=1 << 14); pt.close(); row.append((e, rays, len(rec)))
        except Exception as ex:
            row.append((e, str(ex)[:30]))
    _, cnt, orec = osc.render_lt(W, H, seed, s, 1, max_depth=depth)
    print("pass", s, row, "oracle", cnt.casts, len(orec))
