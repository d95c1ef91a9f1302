"""The imported-mesh workloads of engine BVH (amber_amd/workloads.py): where the reference's `--scene` users land
(/root/reference/src/amber/cli/application.cc:74-87, import.cc:49-167: every imported scene is a triangle mesh under Scene::Create<BVH>).

CPU: the generators' OBJ + MTL text read back through cli::ImportScene gives exactly the arrays they promise (object order, vertex bits,
materials, lens).  GPU: a band of each workload at its bench frame against the oracle with the config-3 assertions (tests/bvh_parity.py) --
0 differing pixels and equal ray counts against oracle(List), every difference from the reference's BVH classified and bounded:
  (i)   the Cornell box through ENGINE_BVH (both schedulers: pt_megakernel<ENGINE_BVH> on the shallow tree, pt_bvh_megakernel),
  (ii)  Cornell-like room + 1 304-triangle mesh,
  (iii) 1.04 M-triangle displaced terrain with needle triangles at the seams.
"""
import os
from pathlib import Path

import numpy as np
import pytest

import oracle_binding as O
from amber_amd import workloads as WL
from bvh_parity import bits, check_band, check_reference_engine

ROOT = Path(__file__).resolve().parent.parent


def _import(amber, wl, tmp_path):
    return amber.HostScene.import_file(wl.write(tmp_path))


@pytest.mark.parametrize("make", [lambda: WL.room_mesh(2), lambda: WL.terrain_mesh(3, 8)], ids=["room_mesh", "terrain"])
def test_generated_obj_imports_as_promised(amber, oracle, tmp_path, make):
    wl = make()
    kw = wl.arrays()
    n = len(kw["kinds"])
    assert n == wl.n_triangles
    objs, mats, lens = _import(amber, wl, tmp_path).flatten()
    arr = np.frombuffer(objs, dtype=np.dtype([("kind", np.uint32), ("material", np.uint32), ("p", np.float32, (12,))]))
    assert len(arr) == n + 6 and lens.first_blade_object == n and lens.n_blades == 6          # aperture objects last (import.cc:155-157)
    assert (arr["kind"][:n] == 0).all()
    assert np.array_equal(bits(arr["p"][:n, :9]), bits(kw["params"]))
    for i in range(0, n, max(1, n // 97)):
        m, (kind, rho, param) = mats[arr["material"][i]], kw["materials"][kw["material_index"][i]]
        assert m.kind == kind and tuple(m.rho[:]) == tuple(np.float32(x) for x in rho) and m.param == np.float32(param), i
    # the lens of the `#camera` line: same bits as the oracle's MakeThinLens on the expected transform (import.cc:130-154)
    osc = O.Scene.create_arrays(**kw, accel=O.ACCEL_LIST | O.BLADES_LAST)
    origin, g_, l_, fd, sd, pa = osc.lens()
    assert np.array_equal(bits(lens.origin[:]), bits(origin)) and np.array_equal(bits(lens.global_[:]), bits(g_)) and np.array_equal(bits(lens.local_[:]), bits(l_))
    assert bits([lens.focus_distance, lens.sensor_distance, lens.p_area]).tolist() == bits([fd, sd, pa]).tolist()


def test_terrain_seams_are_needles():
    wl = WL.terrain_mesh(4, 16)
    p = wl.arrays()["params"].astype(np.float64).reshape(-1, 3, 3)
    e = np.stack([np.linalg.norm(p[:, 1] - p[:, 0], axis=1), np.linalg.norm(p[:, 2] - p[:, 1], axis=1), np.linalg.norm(p[:, 0] - p[:, 2], axis=1)], 1)
    area2 = np.linalg.norm(np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]), axis=1)
    aspect = e.max(1) ** 2 / np.maximum(area2, 1e-300)
    seams = sum(len(f) for name, _, f in wl.groups if name == "seams")
    assert seams > 0 and (aspect > 100).sum() >= seams * 0.6 and aspect.max() > 1000 and np.median(aspect) < 4


@pytest.mark.gpu
def test_cornell_through_engine_bvh_band_against_both_oracles(amber, oracle):
    """(i) the engine switch at 33 objects: config 2's frame through ENGINE_BVH, both of its schedulers."""
    hs, osc = amber.HostScene.cornell_box(), O.Scene.cornell(O.ACCEL_BVH_CONS)
    for flags, name in ((0, "Cornell, pt_megakernel<ENGINE_BVH>"), (amber.api.PT_FLAG_BVH_ITEMS, "Cornell, pt_bvh_megakernel")):
        st = check_band(amber, hs, osc, 1024, 1024, 12345, 64, (600, 632), max_ref_pixels=8, max_ref_ray_delta=64, engine=amber.ENGINE_BVH, flags=flags, label=name)
    check_reference_engine(amber, hs, 1024, 1024, 12345, 64, (600, 632), st["reference_image"], st["reference_casts"], label="Cornell")


@pytest.mark.gpu
def test_room_mesh_band_against_both_oracles(amber, oracle, tmp_path):
    """(ii) a typical imported scene: 1 304 triangles, the tree 12 deep -> the path-granular kernel; and the item kernel."""
    wl = WL.room_mesh(3)
    hs, osc = _import(amber, wl, tmp_path), O.Scene.create_arrays(**wl.arrays(), accel=O.ACCEL_BVH_CONS | O.BLADES_LAST)
    for flags, name in ((0, "room mesh, engine auto"), (amber.api.PT_FLAG_BVH_ITEMS, "room mesh, pt_bvh_megakernel")):
        st = check_band(amber, hs, osc, 1024, 1024, 7, 64, (640, 672), max_ref_pixels=32, max_ref_ray_delta=256, flags=flags, label=name)
        assert st["lit"] > 0.25
    check_reference_engine(amber, hs, 1024, 1024, 7, 64, (640, 672), st["reference_image"], st["reference_casts"], label="room mesh")


@pytest.mark.gpu
def test_terrain_band_against_both_oracles(amber, oracle, tmp_path):
    """(iii) 1.04 M triangles, 40 000 of them needles: two bands of the bench frame (the far terrain under grazing rays; the near one)."""
    wl = WL.terrain_mesh(16, 56)
    assert wl.n_triangles > 1_000_000
    hs, osc = _import(amber, wl, tmp_path), O.Scene.create_arrays(**wl.arrays(), accel=O.ACCEL_BVH_CONS | O.BLADES_LAST)
    for rows in ((200, 224), (800, 824)):
        st = check_band(amber, hs, osc, 1920, 1080, 3, 32, rows, max_ref_pixels=256, max_ref_ray_delta=2048, label="terrain")
        assert st["lit"] > 0.25
        check_reference_engine(amber, hs, 1920, 1080, 3, 32, rows, st["reference_image"], st["reference_casts"], label="terrain")
    # the product kernel's paths one by one (hit-object sequence, hit distances): 4 rows at 8 spp
    osc.set_accel(O.ACCEL_BVH_CONS)
    so = osc.path_signatures(1920, 1080, 3, 0, 8, (700, 704), threads=16)
    pt = amber.PathTracer(hs, amber.Sensor.default(1920, 1080), seed=3, rows=(700, 704))
    assert np.array_equal(pt.render_signatures(0, 8), so)
    pt.close()


@pytest.mark.gpu
def test_engine_bvh_picks_its_scheduler_from_the_tree_depth(amber):
    """Shallow trees (depth <= 12) render with the path-granular kernel, deeper ones and AMBER_PT_FLAG_BVH_ITEMS with pt_bvh_megakernel; scenes with
    triangles shade in batches of 40, sphere scenes of 52 (AMBER_DEBUG_BVH prints the choice at create)."""
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); import amber_amd as A; from amber_amd import scenes; sn = A.Sensor.default(32, 32); c = A.HostScene.cornell_box();"
            "A.PathTracer(c, sn, engine=A.ENGINE_BVH).close(); A.PathTracer(c, sn, engine=A.ENGINE_BVH, flags=A.PT_FLAG_BVH_ITEMS).close();"
            "A.PathTracer(A.HostScene.create_arrays(**scenes.random_spheres(50_000, 7)), sn).close()") % str(ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, AMBER_DEBUG_BVH="1"))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stderr.splitlines() if "scheduler" in l]
    assert len(lines) == 3, p.stderr
    assert "depth 6; scheduler pt_megakernel<ENGINE_BVH>, shading batch 40" in lines[0]
    assert "scheduler pt_bvh_megakernel, shading batch 40" in lines[1]
    assert "scheduler pt_bvh_megakernel, shading batch 52" in lines[2]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [8, 20, 55, 103])
def test_two_phase_over_groups_of_32_objects(amber, oracle, extra):
    """Between 33 and 128 objects the two-phase engine runs one Phase-A program per group of 32 (AUTO up to 80 objects): the Cornell box plus `extra`
    small quads -- 33, 45, 80 and 128 objects -- a band against both oracles like every other workload, the product kernel's paths one by one, and every
    other engine bit for bit.  (8 extra: exactly 33 objects, the second group holds ONE object.)"""
    from amber_amd import scenes
    kw = scenes.cornell_plus(extra)
    n = len(kw["kinds"]) + kw["n_blades"]
    hs, osc = amber.HostScene.create_arrays(**kw), O.Scene.create_arrays(**kw, accel=O.ACCEL_BVH_CONS)
    W = H = 512
    check_band(amber, hs, osc, W, H, 12345, 32, (300, 332), max_ref_pixels=16, max_ref_ray_delta=128, engine=amber.ENGINE_TWO_PHASE, label=f"two-phase, {n} objects")
    rows = (336, 340)
    so = osc.set_accel(O.ACCEL_BVH_CONS).path_signatures(W, H, 12345, 0, 16, rows, threads=16)
    ref = None
    for engine, flags in ((amber.ENGINE_TWO_PHASE, 0), (amber.ENGINE_AUTO, 0), (amber.ENGINE_LIST, 0), (amber.ENGINE_BVH, 0), (amber.ENGINE_BVH, amber.PT_FLAG_BVH_ITEMS)):
        pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=12345, rows=rows, engine=engine, flags=flags)
        assert np.array_equal(pt.render_signatures(0, 16), so), (engine, flags)
        pt.render_pass(0, 16)
        img, rays = pt.download()
        pt.close()
        if ref is None:
            ref = (bits(img).copy(), rays)
        assert rays == ref[1] and np.array_equal(bits(img), ref[0]), (engine, flags)
    # light tracing through the grouped engine (origin slots of light triangles beyond group 0)
    a = amber.PathTracer(hs, amber.Sensor.default(64, 48), seed=3, engine=amber.ENGINE_TWO_PHASE).lt_trace(0, 2)
    b = amber.PathTracer(hs, amber.Sensor.default(64, 48), seed=3, engine=amber.ENGINE_LIST).lt_trace(0, 2)
    assert a[1] == b[1] and a[0].tobytes() == b[0].tobytes()


@pytest.mark.gpu
def test_two_phase_refuses_more_than_128_objects(amber):
    from amber_amd import scenes
    from amber_amd.api import AmberError
    hs = amber.HostScene.create_arrays(**scenes.cornell_plus(104))
    with pytest.raises(AmberError, match="at most 128 objects"):
        amber.PathTracer(hs, amber.Sensor.default(32, 32), engine=amber.ENGINE_TWO_PHASE)
    amber.PathTracer(hs, amber.Sensor.default(32, 32)).close()               # AUTO: engine BVH


@pytest.mark.gpu
def test_small_imported_mesh_runs_the_grouped_two_phase_engine_with_the_blades_last(amber, oracle, tmp_path):
    """cli::ImportScene puts the aperture blades LAST (import.cc:155-157); the grouped two-phase engine deals them into group 0 all the same (their slots,
    the primary rounds' masks).  A 44-triangle imported room (50 objects): AUTO against the oracle and every other engine, light tracing included."""
    wl = WL.room_mesh(0)
    assert wl.n_triangles == 44
    hs, osc = _import(amber, wl, tmp_path), O.Scene.create_arrays(**wl.arrays(), accel=O.ACCEL_BVH_CONS | O.BLADES_LAST)
    _, _, lens = hs.flatten()
    assert lens.first_blade_object == 44
    W = H = 384
    check_band(amber, hs, osc, W, H, 5, 48, (200, 232), max_ref_pixels=16, max_ref_ray_delta=128, label="imported room, 50 objects, engine auto (two groups)")
    rows = (180, 184)
    so = osc.set_accel(O.ACCEL_BVH_CONS).path_signatures(W, H, 5, 0, 24, rows, threads=16)
    for engine in (amber.ENGINE_AUTO, amber.ENGINE_TWO_PHASE, amber.ENGINE_LIST, amber.ENGINE_BVH):
        pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=5, rows=rows, engine=engine)
        assert np.array_equal(pt.render_signatures(0, 24), so), engine
        pt.close()
    a = amber.PathTracer(hs, amber.Sensor.default(64, 48), seed=3).lt_trace(0, 16)
    b = amber.PathTracer(hs, amber.Sensor.default(64, 48), seed=3, engine=amber.ENGINE_LIST).lt_trace(0, 16)
    assert a[1] == b[1] and a[0].tobytes() == b[0].tobytes(), (a[1], b[1], len(a[0]), len(b[0]))
    assert a[1] > 2 * 64 * 48 * 16                                       # light paths bounce (the aperture is a 1-cm target: splats are rare)
