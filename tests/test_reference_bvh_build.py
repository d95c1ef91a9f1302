"""The product's builder of the REFERENCE's BVH (amber_amd/csrc/hip/ref_bvh_build.h, engine AMBER_ENGINE_REFERENCE_BVH) against the
oracle's restatement of acceleration_bvh.h:134-312 on the same objects: node and leaf counts, depth, the object order the build's
sorts leave behind (it decides distance ties inside a leaf) and a digest of the pre-order walk (topology, every box, every leaf
range).  CPU only: the builder is compiled host-only into a small driver (tests/cpp/ref_bvh_dump.hip)."""
import shutil
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import amber_amd as A               # noqa: E402
import oracle_binding as O          # noqa: E402
from amber_amd import scenes, workloads   # noqa: E402
from fuzz_scenes import scene_for_seed    # noqa: E402

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = tmp_path_factory.mktemp("ref_bvh") / "ref_bvh_dump"
    subprocess.run(["hipcc", "--cuda-host-only", "-O2", "-std=c++17", "-ffp-contract=off", "-pthread", "-o", str(exe),
                    str(ROOT / "tests" / "cpp" / "ref_bvh_dump.hip")], check=True, capture_output=True, timeout=600)
    return exe


def product_tree(driver, hs, tmp_path):
    objs, _, _ = hs.flatten()
    (tmp_path / "objects.bin").write_bytes(bytes(objs))
    r = subprocess.run([str(driver), str(tmp_path / "objects.bin"), str(tmp_path / "order.bin")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    words = r.stdout.split()
    stats = {words[i]: int(words[i + 1]) for i in range(0, len(words), 2)}
    return stats, np.fromfile(tmp_path / "order.bin", np.uint32)


def check(driver, tmp_path, hs, osc, label):
    stats, order = product_tree(driver, hs, tmp_path)
    nodes, leaves, depth = osc.bvh_stats()
    assert (stats["nodes"], stats["leaves"], stats["depth"]) == (nodes, leaves, depth), (label, stats, (nodes, leaves, depth))
    assert np.array_equal(order, osc.bvh_order()), label
    assert stats["digest"] == osc.bvh_digest(), label
    return stats


def test_cornell_box_tree_is_the_references(driver, tmp_path):
    stats = check(driver, tmp_path, A.HostScene.cornell_box(), O.Scene.cornell(), "Cornell")
    assert (stats["nodes"], stats["leaves"]) == (3, 2)          # root + 2 leaves


def test_random_scenes_with_coplanar_clutter_and_every_primitive_kind(driver, tmp_path):
    for seed in list(range(500, 524)) + [3, 7, 11]:              # seed % 4 == 3: hundreds of objects
        sc, _ = scene_for_seed(seed)
        check(driver, tmp_path, A.HostScene.create(**sc), O.Scene.create(**sc, accel=O.ACCEL_BVH), "seed %d" % seed)


def test_equal_centres_keep_the_order_the_references_sorts_leave(driver, tmp_path):
    # 300 spheres on 40 distinct centres (several radii each) and 60 copies of one triangle: every sort meets long runs of equal keys
    rng = np.random.default_rng(17)
    centres = rng.uniform(-1, 1, (40, 3)).astype(np.float32)
    objects = [(1, 0, tuple(float(x) for x in centres[i % 40]) + (0.02 + 0.01 * (i % 7),)) for i in range(300)]
    tri = tuple(float(x) for x in rng.uniform(-1, 1, 9).astype(np.float32))
    objects += [(0, 0, tri)] * 60
    sc = dict(objects=objects, materials=[(0, (0.5, 0.5, 0.5), 0.0)],
              transform=[1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 4, 0, 0, 0, 1], focal_length=0.05, focus_distance=4.0, radius=0.01, n_blades=6)
    check(driver, tmp_path, A.HostScene.create(**sc), O.Scene.create(**sc, accel=O.ACCEL_BVH), "equal centres")


def test_mid_size_scenes(driver, tmp_path):
    kw = scenes.cornell_plus(100)
    check(driver, tmp_path, A.HostScene.create_arrays(**kw), O.Scene.create_arrays(**kw, accel=O.ACCEL_BVH), "Cornell + 100 quads")
    kw = scenes.random_spheres(30000)
    stats = check(driver, tmp_path, A.HostScene.create_arrays(**kw), O.Scene.create_arrays(**kw, accel=O.ACCEL_BVH), "30 000 spheres")
    assert stats["depth"] > 12
    mesh = workloads.room_mesh(3)
    kw = mesh.arrays()
    check(driver, tmp_path, A.HostScene.create_arrays(**kw), O.Scene.create_arrays(**kw, accel=O.ACCEL_BVH), "room mesh")


def test_trees_equal_the_committed_records(driver, tmp_path):
    """tests/golden/reference_bvh_trees.json (tools/make_reference_bvh_golden.py): the oracle today and the product's builder both reproduce the
    recorded trees -- counts, depth, object order, digest -- so a different std::sort (order of equal centres) or a changed restatement is named."""
    import json

    def fnv(a):
        h = 14695981039346656037
        for b in np.ascontiguousarray(a).view(np.uint8).tolist():
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h

    golden = {t["scene"]: t for t in json.loads((ROOT / "tests" / "golden" / "reference_bvh_trees.json").read_text())["trees"]}
    cases = [("cornell", A.HostScene.cornell_box(), O.Scene.cornell(O.ACCEL_BVH))]
    for seed in (3, 7, 500, 501, 502, 503):
        sc, _ = scene_for_seed(seed)
        cases.append(("fuzz_scenes.scene_for_seed(%d)" % seed, A.HostScene.create(**sc), O.Scene.create(**sc, accel=O.ACCEL_BVH)))
    kw = scenes.random_spheres(30000)
    cases.append(("scenes.random_spheres(30000)", A.HostScene.create_arrays(**kw), O.Scene.create_arrays(**kw, accel=O.ACCEL_BVH)))
    assert {c[0] for c in cases} == set(golden)
    for name, hs, osc in cases:
        g = golden[name]
        nodes, leaves, depth = osc.bvh_stats()
        assert (nodes, leaves, depth, str(fnv(osc.bvh_order())), str(osc.bvh_digest())) == (g["nodes"], g["leaves"], g["depth"], g["order_fnv1a64"], g["digest"]), name
        stats, order = product_tree(driver, hs, tmp_path)
        assert (stats["nodes"], stats["leaves"], stats["depth"], str(fnv(order)), str(stats["digest"])) == (g["nodes"], g["leaves"], g["depth"], g["order_fnv1a64"], g["digest"]), name
