"""Host object model, C ABI surface and error behaviour (no GPU needed)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

import oracle_binding as O

ROOT = Path(__file__).resolve().parent.parent


def _declared(*headers):
    out = set()
    for header in headers:
        text = (ROOT / "include" / header).read_text()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        out |= set(re.findall(r"\b(amber_(?:hip|host)_[a-z0-9_]+)\s*\(", text))
    return out


def test_library_exports_every_declared_symbol(amber):
    """The loaded library (the lab build under pytest) has every entry point of the three headers; the binding's lists are the headers'."""
    lib = amber.load_library()
    from amber_amd.api import ABI_SYMBOLS, LAB_SYMBOLS
    product, lab = _declared("amber_hip.h", "amber_host.h"), _declared("amber_hip_lab.h")
    assert len(product) >= 20 and product == set(ABI_SYMBOLS)
    assert lab == set(LAB_SYMBOLS) and not (lab & product)
    for name in product | (lab if amber.is_lab() else set()):
        assert hasattr(lib, name), name
    assert lib.amber_hip_abi_version() == 3


def test_struct_layouts_match_the_header(amber):
    from amber_amd import api
    assert C.sizeof(api.FlatObject) == 56 and C.sizeof(api.FlatMaterial) == 24
    assert C.sizeof(api.FlatThinLens) == 4 * (3 + 9 + 9 + 3 + 3)
    assert C.sizeof(api.Sensor) == 16 and C.sizeof(api.PtParams) == 48


def _resolved(objs, mats):
    out = []
    for o in objs:
        m = mats[o.material]
        out.append((o.kind, tuple(np.array(o.p[:], np.float32).view(np.uint32)), m.kind,
                    tuple(np.array(m.rho[:], np.float32).view(np.uint32)), np.float32(m.param).view(np.uint32), np.float32(m.r0).view(np.uint32)))
    return out


def test_cornell_box_flattening_equals_the_oracles_scene(amber, oracle):
    """etude::CornelBox built through the product's Make* factories == the oracle's independent restatement, bit for bit."""
    objs, mats, lens = amber.HostScene.cornell_box().flatten()
    osc = O.Scene.cornell(O.ACCEL_LIST)
    oobjs, omats = osc.objects(), osc.materials()
    assert len(objs) == len(oobjs) == 25 and lens.n_blades == 6 and lens.first_blade_object == 0
    got = _resolved(objs, mats)
    for g, (kind, mat, p, n) in zip(got, oobjs):
        mk, rho, param, r0 = omats[mat]
        pp = np.zeros(12, np.float32)
        if kind == 0:
            pp[:9], pp[9:] = p, n
        else:
            pp[:4] = p[:4]
        assert g == (kind, tuple(pp.view(np.uint32)), mk, tuple(rho.view(np.uint32)), param.view(np.uint32), r0.view(np.uint32))
    origin, g_, l_, fd, sd, pa = osc.lens()
    assert np.array_equal(np.array(lens.origin[:], np.float32), origin)
    assert np.array_equal(np.array(lens.global_[:], np.float32), g_) and np.array_equal(np.array(lens.local_[:], np.float32), l_)
    assert (np.float32(lens.focus_distance), np.float32(lens.sensor_distance), np.float32(lens.p_area)) == (fd, sd, pa)
    # application.cc:69-73 + cornel_box.cc:50-61: sensor distance of the thin lens
    assert np.float32(lens.sensor_distance) == np.float32(1) / (np.float32(1) / np.float32(0.05) - np.float32(1) / np.float32(4))


SCENE = dict(
    materials=[(4, (30.0, 30.0, 30.0), 0.0), (0, (0.7, 0.6, 0.5), 0.0), (1, (0.9, 0.9, 0.9), 32.0), (2, (0.8, 0.8, 0.8), 0.0), (3, (1.0, 1.0, 1.0), 1.5)],
    objects=[
        (2, 0, [0.0, 1.5, 0.0, 0.0, -1.0, 0.0, 0.6]),                      # disk light facing down
        (0, 1, [-3, -1, -3, 3, -1, 3, 3, -1, -3]), (0, 1, [-3, -1, -3, -3, -1, 3, 3, -1, 3]),   # floor
        (3, 2, [-0.8, -1.0, 0.0, 0.0, 1.0, 0.0, 0.3, 0.9]),               # phong cylinder
        (1, 3, [0.7, -0.6, -0.3, 0.4]), (1, 4, [0.0, -0.7, 0.8, 0.3]),    # mirror and glass spheres
    ],
    transform=[1, 0, 0, 0, 0, 1, 0, 0.2, 0, 0, 1, 5, 0, 0, 0, 1], focal_length=0.05, focus_distance=5.0, radius=0.02, n_blades=5,
)


def test_generic_scene_with_all_primitive_kinds(amber, oracle):
    hs = amber.HostScene.create(**SCENE)
    objs, mats, lens = hs.flatten()
    osc = O.Scene.create(**SCENE)
    oobjs, omats = osc.objects(), osc.materials()
    assert len(objs) == len(oobjs) == 5 + 6 and lens.n_blades == 5
    for o, (kind, mat, p, n) in zip(objs, oobjs):
        assert o.kind == kind
        k = {0: 9, 1: 4, 2: 7, 3: 8}[kind]
        assert np.array_equal(np.array(o.p[:k], np.float32).view(np.uint32), p[:k].view(np.uint32))
        if kind == 0:
            assert np.array_equal(np.array(o.p[9:12], np.float32).view(np.uint32), n.view(np.uint32))
        m, (mk, rho, param, r0) = mats[o.material], omats[mat]
        assert (m.kind, np.float32(m.param), np.float32(m.r0)) == (mk, param, r0)
        assert np.array_equal(np.array(m.rho[:], np.float32), rho)


def test_errors_are_loud(amber):
    lib = amber.load_library()
    sc = amber.HostScene.cornell_box()
    if amber.device_count() == 0:
        with pytest.raises(amber.AmberError, match="no HIP device"):
            amber.PathTracer(sc, amber.Sensor.default(32, 32))            # no CPU fallback
    with pytest.raises(amber.AmberError, match="Unknown algorithm"):      # cli::UnknownAlgorithmError, algorithm_factory.h:30-37
        sc.render(amber.Sensor.default(8, 8), 1, algorithm="bdpt")
    with pytest.raises(amber.AmberError):
        amber.HostScene.create(objects=[(0, 7, [0] * 9)], materials=[(0, (1, 1, 1), 0)], transform=SCENE["transform"],
                               focal_length=0.05, focus_distance=5.0, radius=0.02, n_blades=5)   # material index out of range
    with pytest.raises(amber.AmberError):
        amber.HostScene.create(objects=[(9, 0, [0] * 9)], materials=[(0, (1, 1, 1), 0)], transform=SCENE["transform"],
                               focal_length=0.05, focus_distance=5.0, radius=0.02, n_blades=5)   # unknown primitive
    # malformed flat scenes are rejected before any device work
    from amber_amd import api
    fs = api.FlatSceneC()
    h = C.c_void_p()
    p = api.PtParams()
    s = amber.Sensor.default(8, 8)
    assert lib.amber_hip_pt_create(C.byref(fs), C.byref(s), C.byref(p), C.byref(h)) == -1
    assert b"no objects" in lib.amber_hip_last_error()
    assert lib.amber_hip_pt_render_pass(None, 0, 1) == -1


def test_partition_rows():
    from amber_amd.distributed import partition_rows
    for h, g in [(1024, 1), (1024, 2), (1024, 8), (2160, 8), (100, 3), (8, 4), (20, 6)]:
        bands = partition_rows(h, g)
        assert len(bands) == g and bands[0][0] == 0 and bands[-1][1] == h
        for (a0, a1), (b0, b1) in zip(bands, bands[1:]):
            assert a1 == b0 and a0 <= a1
        assert all(y0 % 8 == 0 for y0, y1 in bands if y1 > y0)
        rows = [y1 - y0 for y0, y1 in bands]
        assert max(rows) - min(rows) < 16 or h < 8 * g      # one tile row of slack + a truncated last tile


def test_stripe_partition_covers_every_row_once():
    from amber_amd.distributed import stripe_partition
    for h, g, s in [(1024, 8, 8), (1024, 1, 8), (2160, 8, 8), (100, 3, 4), (7, 4, 8), (52, 3, 4)]:
        parts = stripe_partition(h, g, s)
        allrows = np.concatenate([p["index"] for p in parts])
        assert sorted(allrows.tolist()) == list(range(h))
        for r, p in enumerate(parts):
            y0, y1 = p["rows"]
            ys = np.arange(y0, y1)
            if p["stripe"]:
                S, period = p["stripe"]
                assert np.array_equal(ys[(ys - y0) % period < S], p["index"])      # exactly the rows the engine will own
            else:
                assert np.array_equal(ys, p["index"])
        counts = [len(p["index"]) for p in parts]
        assert max(counts) - min(counts) <= s


def test_no_spill_code_in_front_of_an_exec_restore():
    """The compiled product kernels are free of the one hipcc miscompile this code base has met (tools/check_spill_placement.py,
    EXPERIMENTS.md): VGPR spill stores / reloads placed before the `s_or_b64 exec` of a join block run for the lanes of the
    `if` body only, and the others later read stale scratch -- round 2's diagnostic build lost 5 % of its rays that way.
    hipcc cross-compiles without a GPU, so this runs everywhere; the known-bad pattern itself is a fixture."""
    import shutil
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root / "tools"))
    import check_spill_placement as lint
    bad = (Path(__file__).parent / "golden" / "misplaced_spill_r02_stamps.s").read_text()
    found = list(lint.scan(bad))
    assert len(found) == 3 and all("Folded Spill" in t for _, _, _, t in found)          # the lint sees the real thing
    if shutil.which("hipcc") is None:
        import pytest
        pytest.skip("no hipcc on this box")
    product = lint.compile_to_asm([])
    assert list(lint.scan(product)) == []                                                  # the product
    assert list(lint.scan(lint.compile_to_asm(["-DAMBER_LAB"]))) == []                     # the lab build the suite runs on
    # The headline kernel lives at its SGPR limit: 198 of its instructions move scalars to VGPR lanes and back.  Round 5 met the price of one more --
    # a generalisation of the two-phase closest hit that read ONE more field of the scene record made it 209 and the kernel 1.2 % slower
    # (tools/ab_lib.py over the round's commits, EXPERIMENTS.md) -- so the count is pinned here: a change that raises it must be measured.
    name = "_ZN12_GLOBAL__N_113pt_megakernelILi2ELb0ELb0EEEvNS_10RenderArgsE"
    body = product[product.index("\n" + name + ":"):]
    body = body[:body.index("s_endpgm")]
    lane_ops = sum(1 for line in body.splitlines() if line.strip().startswith(("v_readlane_b32", "v_writelane_b32")))
    assert 0 < lane_ops <= 198, lane_ops
