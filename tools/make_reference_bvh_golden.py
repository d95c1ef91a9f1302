"""Writes tests/golden/reference_bvh_trees.json: node / leaf counts, depth, object order (its FNV-1a64) and pre-order digest of the reference's BVH
(acceleration_bvh.h:134-312) as the ORACLE builds it, for the Cornell box, six seeded random scenes and 30 000 random spheres.  The Cornell entry is
tied to the reference itself through the survey's recorded images (tests/test_oracle_pin.py renders them through this tree).  A regression record:
tests/test_reference_bvh_build.py checks the oracle and the product's builder (ref_bvh_build.h) against it, so a change of the C++ library's std::sort
(the order of equal centres) or of either restatement shows up by name.     python tools/make_reference_bvh_golden.py"""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import oracle_binding as O
from amber_amd import scenes
from fuzz_scenes import scene_for_seed


def fnv(a):
    h = 14695981039346656037
    for b in np.ascontiguousarray(a).view(np.uint8).tolist():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def entry(name, osc):
    nodes, leaves, depth = osc.bvh_stats()
    return {"scene": name, "nodes": nodes, "leaves": leaves, "depth": depth, "order_fnv1a64": str(fnv(osc.bvh_order())), "digest": str(osc.bvh_digest())}


out = [entry("cornell", O.Scene.cornell(O.ACCEL_BVH))]
for seed in (3, 7, 500, 501, 502, 503):
    sc, _ = scene_for_seed(seed)
    out.append(entry("fuzz_scenes.scene_for_seed(%d)" % seed, O.Scene.create(**sc, accel=O.ACCEL_BVH)))
out.append(entry("scenes.random_spheres(30000)", O.Scene.create_arrays(**scenes.random_spheres(30000), accel=O.ACCEL_BVH)))
path = os.path.join(R, "tests", "golden", "reference_bvh_trees.json")
json.dump({"made_by": "tools/make_reference_bvh_golden.py (oracle/amber_oracle.cc, g++ / libstdc++ of this image)", "trees": out}, open(path, "w"), indent=1)
print(open(path).read())
