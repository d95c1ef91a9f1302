"""Build options of the engine that are not the default stay correct: the 4-wide BVH (-DAMBER_BVH_WIDE=1, `make wide`) must
give the images and ray counts of engine LIST, bit for bit, like the default 2-wide tree."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import amber_amd as A
from amber_amd import scenes
assert str(A.library_path()).endswith("libamber_hip_wide.so")
hs = A.HostScene.create_arrays(**scenes.random_spheres(20000, 3))
res = {}
for e in (A.ENGINE_BVH, A.ENGINE_LIST):
    pt = A.PathTracer(hs, A.Sensor.default(96, 64), seed=5, engine=e)
    pt.render_pass(0, 6); img, rays = pt.download(); pt.close()
    res[e] = (img.view(np.uint32).copy(), rays)
assert res[A.ENGINE_BVH][1] == res[A.ENGINE_LIST][1], (res[A.ENGINE_BVH][1], res[A.ENGINE_LIST][1])
assert np.array_equal(res[A.ENGINE_BVH][0], res[A.ENGINE_LIST][0])
box = {}
for e in (A.ENGINE_BVH, A.ENGINE_LIST):                       # few objects: leaves next to the root, absent children
    pt = A.PathTracer(A.HostScene.cornell_box(), A.Sensor.default(64, 64), seed=2, engine=e)
    pt.render_pass(0, 8); box[e] = pt.download()[0].view(np.uint32).copy(); pt.close()
assert np.array_equal(box[A.ENGINE_BVH], box[A.ENGINE_LIST])
print("WIDE OK", res[A.ENGINE_BVH][1])
"""


@pytest.mark.gpu
def test_four_wide_bvh_build_matches_engine_list():
    subprocess.run(["make", "-C", str(ROOT / "amber_amd" / "csrc"), "wide"], check=True, capture_output=True, timeout=900)
    env = dict(os.environ, AMBER_AMD_LIB="libamber_hip_wide.so")
    r = subprocess.run([sys.executable, "-c", SCRIPT % str(ROOT)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "WIDE OK" in r.stdout, r.stdout + r.stderr


STAMPS_SCRIPT = r"""
import sys, hashlib
sys.path.insert(0, %r)
import amber_amd as A
assert str(A.library_path()).endswith(%r)
sc = A.HostScene.cornell_box(); sn = A.Sensor.default(256, 256)
seen = set()
for rnd in range(4):                      # the miscompile this guards against showed as run-to-run differences (stale scratch)
    for e in (A.ENGINE_TWO_PHASE, A.ENGINE_LIST, A.ENGINE_BVH):
        pt = A.PathTracer(sc, sn, engine=e); pt.render_pass(0, 32); img, rays = pt.download(); pt.close()
        seen.add((rays, hashlib.sha1(img.tobytes()).hexdigest()))
assert len(seen) == 1, seen
print("SAME PATHS", seen)
"""


@pytest.mark.gpu
def test_diagnostic_build_renders_the_same_paths():
    """The -DAMBER_STAMPS build (tools/stamps.py, tools/bvh_counters.py) adds clocks and counters, nothing else: every engine
    must cast exactly the rays of engine LIST and produce its image, every time.  Round 2's stamped two-phase kernel lost 5 % of
    its rays, differently in every run: hipcc had placed VGPR spill stores in front of the `s_or_b64 exec` of a join block, so the
    Phong lanes reloaded stale scratch (EXPERIMENTS.md; tools/check_spill_placement.py lints the ISA for the pattern).  The
    product library runs the same loop: its kernels are built at their register caps, where such spills would appear first."""
    subprocess.run(["make", "-C", str(ROOT / "amber_amd" / "csrc"), "stamps"], check=True, capture_output=True, timeout=900)
    for lib in ("libamber_hip_stamps.so", "libamber_hip.so"):
        env = dict(os.environ, AMBER_AMD_LIB=lib)
        r = subprocess.run([sys.executable, "-c", STAMPS_SCRIPT % (str(ROOT), lib)], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0 and "SAME PATHS" in r.stdout, lib + ": " + r.stdout + r.stderr
