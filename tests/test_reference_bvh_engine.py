"""AMBER_ENGINE_REFERENCE_BVH on the GPU: the reference's own tree (ref_bvh_build.h, proved equal to the oracle's restatement of
acceleration_bvh.h:134-312 on the CPU in tests/test_reference_bvh_build.py) walked in the order of BVH::Node::Cast (:340-403).
The bar is the reference's own closest hit: image bits, ray counts and every path's hit sequence == oracle(ACCEL_BVH) -- no tie or
lost-hit pixel left over.  (The bands of config 3 and of the mesh workloads: tests/test_config3_parity_gpu.py, test_mesh_workloads.py.)"""
import numpy as np
import pytest

import oracle_binding as O
from fuzz_scenes import scene_for_seed

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_config1_frame_equals_the_reference_bvh_oracle(amber):
    """BASELINE config 1 (Cornell 256 x 256 @ 16 spp), the whole frame; and config 2's frame size on 24 rows at 128 spp."""
    hs, osc = amber.HostScene.cornell_box(), O.Scene.cornell(O.ACCEL_BVH)
    for (w, h, spp, rows) in ((256, 256, 16, (0, 256)), (1024, 1024, 128, (500, 524))):
        pt = amber.PathTracer(hs, amber.Sensor.default(w, h), seed=12345, rows=rows, engine=amber.ENGINE_REFERENCE_BVH)
        pt.render_pass(0, spp)
        img, rays = pt.download()
        pt.close()
        full = np.zeros((h, w, 3), np.float32)
        _, cnt = osc.render_xorshift(w, h, 12345, 0, spp, math=O.MATH_LIBM, threads=16, rows=rows, out=full)
        assert rays == cnt.casts
        assert np.array_equal(bits(img), bits(full[rows[0]:rows[1]]))


def test_paths_one_by_one_against_the_reference_bvh_oracle(amber):
    """hit-object sequence and hit distances of every path (the kernel's signature instantiation): Cornell and a 300-object soup"""
    hs, osc = amber.HostScene.cornell_box(), O.Scene.cornell(O.ACCEL_BVH)
    pt = amber.PathTracer(hs, amber.Sensor.default(256, 256), seed=5, rows=(96, 112), engine=amber.ENGINE_REFERENCE_BVH)
    assert np.array_equal(pt.render_signatures(0, 32), osc.path_signatures(256, 256, 5, 0, 32, (96, 112), threads=16))
    pt.close()
    sc, _ = scene_for_seed(7)
    hs, osc = amber.HostScene.create(**sc), O.Scene.create(**sc, accel=O.ACCEL_BVH)
    pt = amber.PathTracer(hs, amber.Sensor.default(96, 64), seed=7, engine=amber.ENGINE_REFERENCE_BVH)
    assert np.array_equal(pt.render_signatures(0, 16), osc.path_signatures(96, 64, 7, 0, 16, (0, 64), threads=16))
    pt.close()


def test_random_scenes_equal_the_reference_bvh_oracle_where_the_list_engines_do_not(amber):
    """Coplanar clutter, fans, coincident objects, every primitive kind and material, both lenses: image bits and ray count ==
    oracle(ACCEL_BVH).  The scenes are built to contain exact distance ties, which List and the reference's BVH resolve differently:
    the count of scenes on which engine AUTO (List semantics) differs from this engine shows the test bites."""
    W, H, spp = 64, 48, 8
    list_differs = 0
    for seed in range(900, 948):
        sc, _ = scene_for_seed(seed)
        hs, osc = amber.HostScene.create(**sc), O.Scene.create(**sc, accel=O.ACCEL_BVH)
        images = {}
        for engine in (amber.ENGINE_REFERENCE_BVH, amber.ENGINE_AUTO):
            pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=seed, engine=engine)
            pt.render_pass(0, spp)
            images[engine] = pt.download()
            pt.close()
        oimg, cnt = osc.render_xorshift(W, H, seed, 0, spp, math=O.MATH_LIBM, threads=16)
        img, rays = images[amber.ENGINE_REFERENCE_BVH]
        assert rays == cnt.casts, seed
        assert np.array_equal(bits(img), bits(oimg)), seed
        list_differs += int(not np.array_equal(bits(images[amber.ENGINE_AUTO][0]), bits(oimg)))
    print(f"\n48 random scenes: engine REFERENCE_BVH == oracle(reference BVH) on all; the List engines differ from it on {list_differs}")


def test_light_tracing_through_the_references_tree(amber):
    """algorithm_lt.cc on the same engine: the splat records == the oracle's with its reference BVH"""
    lights = dict(
        materials=[(4, (3.0, 2.0, 1.0), 0.0), (0, (0.6, 0.6, 0.6), 0.0), (2, (0.9, 0.9, 0.9), 0.0), (3, (1.0, 1.0, 1.0), 1.5)],
        objects=[
            (1, 0, [0.0, 1.2, 0.0, 0.25]), (0, 0, [-0.4, 1.0, -0.4, 0.4, 1.0, -0.4, 0.0, 1.0, 0.4]),
            (0, 1, [-3, -1, -3, 3, -1, 3, 3, -1, -3]), (0, 1, [-3, -1, -3, -3, -1, 3, 3, -1, 3]),
            (1, 2, [0.7, -0.6, -0.3, 0.4]), (1, 3, [0.0, -0.5, 0.8, 0.45]),
        ],
        transform=[1, 0, 0, 0, 0, 1, 0, 0.2, 0, 0, 1, 2.6, 0, 0, 0, 1], focal_length=0.05, focus_distance=2.6, radius=0.45, n_blades=5,
    )
    hs, osc = amber.HostScene.create(**lights), O.Scene.create(**lights, accel=O.ACCEL_BVH)
    pt = amber.PathTracer(hs, amber.Sensor.default(48, 36), seed=13, engine=amber.ENGINE_REFERENCE_BVH)
    rec, rays = pt.lt_trace(2, 400)
    pt.close()
    _, cnt, oref = osc.render_lt(48, 36, 13, 2, 400)
    assert rays == cnt.casts and len(rec) == len(oref) and len(rec) > 30
    got = np.stack([rec["path"], rec["sample"], rec["bounce"], rec["pixel"], *[rec["rgb"][:, c].view(np.uint32) for c in range(3)]], 1)
    assert np.array_equal(got, oref)


def test_a_scene_the_reference_keeps_in_one_leaf_and_the_engine_id_gap(amber):
    """two objects: the reference's root is a leaf (no inner node to flatten); engine id 5 is not public"""
    sc = dict(objects=[(1, 0, (0.0, 0.0, 0.0, 0.5)), (1, 1, (0.0, 2.0, 0.0, 0.3))], materials=[(0, (0.7, 0.7, 0.7), 0.0), (4, (5.0, 5.0, 5.0), 0.0)],
              transform=[1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 4, 0, 0, 0, 1], focal_length=0.05, focus_distance=4.0, radius=0.01, n_blades=1)
    hs, osc = amber.HostScene.create(**sc), O.Scene.create(**sc, accel=O.ACCEL_BVH)
    pt = amber.PathTracer(hs, amber.Sensor.default(64, 64), seed=1, engine=amber.ENGINE_REFERENCE_BVH)
    pt.render_pass(0, 16)
    img, rays = pt.download()
    pt.close()
    oimg, cnt = osc.render_xorshift(64, 64, 1, 0, 16, math=O.MATH_LIBM)
    assert rays == cnt.casts and np.array_equal(bits(img), bits(oimg)) and img.sum() > 0
    with pytest.raises(Exception):
        amber.PathTracer(hs, amber.Sensor.default(64, 64), seed=1, engine=5)
    # a NaN centre: the reference's build would hand std::sort an order that is none (undefined behaviour); this engine refuses the scene, the others render it
    sc["objects"] = sc["objects"] + [(1, 0, (float("nan"), 0.0, 0.0, 0.1))]
    hs = amber.HostScene.create(**sc)
    with pytest.raises(Exception, match="NaN"):
        amber.PathTracer(hs, amber.Sensor.default(64, 64), seed=1, engine=amber.ENGINE_REFERENCE_BVH)
    amber.PathTracer(hs, amber.Sensor.default(64, 64), seed=1).close()


def test_command_line_with_engine_6_writes_the_reference_bvh_image(amber, tmp_path):
    """bin/amber --scene <obj> --engine 6: Algorithm<RGB>::Render on the imported scene through the reference's own tree; output.exr ==
    oracle(ACCEL_BVH) of the same objects (mean of the passes, x mirrored: cli/image.cc:50)."""
    import subprocess
    from pathlib import Path
    from amber_amd import workloads
    from test_output_stage import parse_exr
    wl = workloads.room_mesh(1)
    exe = Path(amber.library_path()).parent.parent / "bin" / "amber"
    out = str(tmp_path / "cli")
    r = subprocess.run([str(exe), "--algorithm", "pt", "--scene", str(wl.write(tmp_path)), "--width", "96", "--height", "64", "--spp", "16", "--seed", "5",
                        "--samples-per-launch", "16", "--engine", "6", "--output", out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    osc = O.Scene.create_arrays(**wl.arrays(), accel=O.ACCEL_BVH | O.BLADES_LAST)
    oimg, _ = osc.render_xorshift(96, 64, 5, 0, 16, math=O.MATH_LIBM)
    assert np.array_equal(bits(parse_exr(out + ".exr")), bits((oimg / np.float32(16))[:, ::-1]))


def test_stripes_of_three_ranks_through_the_references_tree_equal_the_whole_frame(amber):
    """the sharding of bench.py --gpus N (interleaved 8-row stripes, one handle per rank) with this engine: every rank's rows are the whole frame's rows,
    the ray counts add up -- a 300-object soup at 96 x 64"""
    from amber_amd.distributed import stripe_partition
    sc, _ = scene_for_seed(11)
    hs = amber.HostScene.create(**sc)
    W, H, spp = 96, 64, 16
    pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=4, engine=amber.ENGINE_REFERENCE_BVH)
    pt.render_pass(0, spp)
    full, rays = pt.download()
    pt.close()
    total = 0
    for part in stripe_partition(H, 3):
        pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=4, rows=part["rows"], stripe=part["stripe"], engine=amber.ENGINE_REFERENCE_BVH)
        pt.render_pass(0, spp)
        img, r = pt.download()
        pt.close()
        total += r
        assert np.array_equal(bits(img), bits(full[part["index"]]))
    assert total == rays
