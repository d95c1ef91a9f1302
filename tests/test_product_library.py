"""libamber_hip.so, the PRODUCT, next to the lab build the rest of the suite runs on.

CPU: `nm -D` of the product lists the documented ABI -- include/amber_hip.h, include/amber_host.h, the C++ interface of the host object model
(namespace amber: bin/amber links against it) -- and nothing of the laboratory: no known-answer entry point, no helper, no signature hook.
GPU: a child process that loads ONLY the product renders the Cornell box (two-phase), the Cornell box through ENGINE_BVH (both schedulers), a
deep sphere tree and an imported mesh; this process (lab build) renders the same: identical image bits and ray counts.  The product refuses
what it does not contain (engine WAVEFRONT, AMBER_PT_FLAG_BVH_POOL) with AMBER_EINVAL."""
import json
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "amber_amd" / "lib"


def _exports(path):
    out = subprocess.run(["nm", "-D", "--defined-only", str(path)], capture_output=True, text=True, check=True).stdout
    return [line.split()[-1] for line in out.splitlines() if line.strip()]


def test_product_exports_only_the_documented_abi(amber):
    from amber_amd.api import ABI_SYMBOLS, LAB_SYMBOLS
    names = _exports(LIB / "libamber_hip.so")
    c_names = sorted(n for n in names if not n.startswith("_Z") and not n.startswith("__hip_cuid_"))     # (__hip_cuid_*: one marker per translation unit, emitted by hipcc)
    assert c_names == sorted(ABI_SYMBOLS), sorted(set(c_names) ^ set(ABI_SYMBOLS))
    demangled = subprocess.run(["c++filt"], input="\n".join(n for n in names if n.startswith("_Z")), capture_output=True, text=True, check=True).stdout.splitlines()
    for d in demangled:
        # the host object model's C++ interface, and standard-library templates instantiated for it (weak, vague linkage)
        assert re.search(r"\bamber::(scene|rendering|cli|prelude|postprocess|raytracer|etude)\b", d) or d.startswith(("std::", "void std::", "typeinfo", "vtable", "guard variable")) or "std::" in d, d
        assert "kat" not in d and "signature" not in d.lower(), d
    lab = set(_exports(LIB / "libamber_hip_lab.so"))
    assert set(LAB_SYMBOLS) <= lab and set(ABI_SYMBOLS) <= lab and not (set(LAB_SYMBOLS) & set(names))


CHILD = r"""
import os, sys, tempfile, json
sys.path.insert(0, {root!r})
import numpy as np
import amber_amd as A
from amber_amd import scenes, workloads
assert A.library_path().name == "libamber_hip.so" and not A.is_lab()
out = {{}}
def render(name, hs, W, H, spp, **kw):
    pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=11, **kw); pt.render_pass(0, spp); img, rays = pt.download(); pt.close()
    np.save(os.path.join({tmp!r}, name + ".npy"), img); out[name] = int(rays)
cornell = A.HostScene.cornell_box()
render("cornell", cornell, 160, 120, 24)
render("cornell_bvh", cornell, 160, 120, 24, engine=A.ENGINE_BVH)
render("cornell_items", cornell, 160, 120, 24, engine=A.ENGINE_BVH, flags=A.PT_FLAG_BVH_ITEMS)
render("cornell_list", cornell, 160, 120, 24, engine=A.ENGINE_LIST)
render("spheres", A.HostScene.create_arrays(**scenes.random_spheres(60_000, 7)), 192, 108, 16)
render("mesh", A.HostScene.import_file(workloads.room_mesh(3).write({tmp!r})), 128, 128, 16)
img, st = cornell.render(A.Sensor.default(64, 48), 16, seed=3); np.save(os.path.join({tmp!r}, "adapter.npy"), img); out["adapter"] = int(st["rays"])
refused = []
for kw in (dict(engine=A.ENGINE_WAVEFRONT), dict(engine=A.ENGINE_BVH, flags=A.PT_FLAG_BVH_POOL)):
    try:
        A.PathTracer(cornell, A.Sensor.default(32, 32), **kw); refused.append(None)
    except A.AmberError as e:
        refused.append(str(e))
out["refused"] = refused
out["lab_symbols"] = [s for s in A.api.LAB_SYMBOLS if hasattr(A.load_library(), s)]
print("RESULT " + json.dumps(out))
"""


@pytest.mark.gpu
def test_product_renders_the_same_bits_as_the_lab_build(amber, tmp_path):
    from amber_amd import scenes, workloads
    assert amber.is_lab()
    env = dict(os.environ, AMBER_AMD_LIB="libamber_hip.so")
    p = subprocess.run([sys.executable, "-c", CHILD.format(root=str(ROOT), tmp=str(tmp_path))], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    assert res["lab_symbols"] == []
    assert all(r and "lab build" in r and "error -1" in r for r in res["refused"]), res["refused"]

    def render(hs, W, H, spp, **kw):
        pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=11, **kw); pt.render_pass(0, spp); img, rays = pt.download(); pt.close()
        return img, rays
    cornell = amber.HostScene.cornell_box()
    mine = {"cornell": render(cornell, 160, 120, 24), "cornell_bvh": render(cornell, 160, 120, 24, engine=amber.ENGINE_BVH),
            "cornell_items": render(cornell, 160, 120, 24, engine=amber.ENGINE_BVH, flags=amber.PT_FLAG_BVH_ITEMS),
            "cornell_list": render(cornell, 160, 120, 24, engine=amber.ENGINE_LIST),
            "spheres": render(amber.HostScene.create_arrays(**scenes.random_spheres(60_000, 7)), 192, 108, 16),
            "mesh": render(amber.HostScene.import_file(workloads.room_mesh(3).write(tmp_path / "lab")), 128, 128, 16)}
    img, st = cornell.render(amber.Sensor.default(64, 48), 16, seed=3)
    mine["adapter"] = (img, st["rays"])
    for name, (img, rays) in mine.items():
        got = np.load(tmp_path / (name + ".npy"))
        assert res[name] == rays, (name, res[name], rays)
        assert np.array_equal(got.view(np.uint32), img.view(np.uint32)), name
    assert np.array_equal(mine["cornell"][0].view(np.uint32), mine["cornell_bvh"][0].view(np.uint32))      # and the engines agree, as everywhere
