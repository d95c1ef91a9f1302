"""One fuzz scene, one engine, with timing: python tools/fuzz_one.py <seed> <engine> [spp]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import amber_amd as A
seed, engine = int(sys.argv[1]), int(sys.argv[2]); spp = int(sys.argv[3]) if len(sys.argv) > 3 else 6
sys.path.insert(0, os.path.join(R, "tests")); from fuzz_scenes import scene_for_seed
sc, rng = scene_for_seed(seed, scaled="--scaled" in sys.argv, extreme="--extreme" in sys.argv)
print("seed", seed, "objects", len(sc["objects"]), "blades", sc["n_blades"], flush=True)
hs = A.HostScene.create(**sc)
t = time.time(); pt = A.PathTracer(hs, A.Sensor.default(48, 40), seed=seed, engine=engine); print("create %.2f s" % (time.time() - t), flush=True)
t = time.time(); pt.render_pass(0, spp); img, rays = pt.download(); print("engine %d: %.2f s, rays %d, %.2f rays/path" % (engine, time.time() - t, rays, rays / (48 * 40 * spp)), flush=True)
