"""GPU parity tests: the HIP engine (through the C ABI) against the CPU oracle -- bit-exact.

Everything the engine computes is binary32/binary64 arithmetic in the oracle's operation order; sin / cos / pow are
glibc's own double-precision kernels executed operation for operation (DESIGN.md "Numerics"), so the bar is equality
of bits, not a tolerance: object indices, hit distances, positions, weights, measurements, images and ray counts.
The oracle runs in the mode that restates the loaded library's arithmetic (O.MATH_DEVICE: GLIBC for the product);
test_parity_against_live_libm.py compares with the oracle calling the host's libm, as the reference does.
"""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

import oracle_binding as O
from test_host_model import SCENE

pytestmark = pytest.mark.gpu
GOLD = np.load(Path(__file__).parent / "golden" / "oracle_vectors.npz")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def cornell(amber):
    return amber.HostScene.cornell_box(), O.Scene.cornell(O.ACCEL_LIST)


@pytest.fixture(scope="module")
def generic(amber):
    return amber.HostScene.create(**SCENE), O.Scene.create(**SCENE)


def test_extension_is_loaded_and_device_present(amber):
    assert amber.library_path().exists()
    assert amber.device_count() >= 1


def test_device_math_bit_exact(amber, oracle):
    """sin/cos/pow of the engine == the oracle mode that restates them, and (product build, FMA-capable x86-64 host)
    == the host's live libm, which is what the reference calls."""
    modes = [O.MATH_DEVICE]
    fb = (C.c_float * 2)()
    live_equals_restatement = oracle.oracle_math_compare(0, O.MATH_LIBM, O.MATH_GLIBC, 0, 1 << 18, 0.0, fb) == 0
    if O.MATH_DEVICE == O.MATH_GLIBC and live_equals_restatement:
        modes.append(O.MATH_LIBM)
    k = np.arange(0, 1 << 24, 257, dtype=np.float32)
    x = np.concatenate([(np.float32(2.0) * np.float32(3.14159274)) * (k * np.float32(2.0 ** -24)),      # the path's phi = 2 pi u
                        np.linspace(-119.0, 119.0, 20001),
                        [0.0, 6.2831855, 1e-8, np.pi / 4, np.pi / 2, 0.75, 0.78539816, 2.0 ** -12, 2.4414e-4, -0.3, 1e-40]]).astype(np.float32)
    g = amber.kat_math(0, x)
    s, c = C.c_float(), C.c_float()
    for mode in modes:
        for i in range(len(x)):
            oracle.oracle_sincos(x[i], mode, C.byref(s), C.byref(c))
            assert bits(g[i]).tolist() == bits([s.value, c.value]).tolist(), (mode, x[i])
    rng = np.random.default_rng(0)
    xy = np.stack([rng.random(20000), rng.choice([1 / 257, 1 / 33, 0.5, 1 / 3, 2.0, 7.5], 20000)], 1).astype(np.float32)
    xy[:10] = [[0, 0.5], [1, 0.3], [1e-38, 0.25], [0.5, 0], [2.0 ** -24, 1 / 257], [1e-42, 0.5], [3.0, 80.0], [0.5, 160.0], [0.5, 149.5], [1.0000001, 1e9]]
    g = amber.kat_math(1, xy)
    for mode in modes:
        for i in range(len(xy)):
            assert bits(g[i]) == bits(oracle.oracle_pow(xy[i, 0], xy[i, 1], mode)), (mode, xy[i])
    # x^4, x^5 in binary64 (std::pow(float, int)): equal to the restating mode bit for bit; against the live pow() the
    # last bit of the DOUBLE may differ (glibc's pow is not correctly rounded) but never the value rounded to binary32
    xs = np.concatenate([rng.random(5000), rng.random(1000) * 1e-3, [0.0, 1.0, 0.5]]).astype(np.float32)
    for n, kmode in ((4, 2), (5, 3)):
        g = amber.kat_math(kmode, xs)
        ref = np.array([oracle.oracle_pow_i(float(v), n, O.MATH_DEVICE) for v in xs])
        assert np.array_equal(g.view(np.uint64), ref.view(np.uint64)), n
        live = np.array([oracle.oracle_pow_i(float(v), n, O.MATH_LIBM) for v in xs])
        assert np.array_equal(g.astype(np.float32).view(np.uint32), live.astype(np.float32).view(np.uint32)), n


def _check_casts(pt, osc, org, d):
    obj, t, pos, nrm = pt.kat_cast(org, d)
    for i in range(len(org)):
        oi, ot, op, on = osc.cast(org[i], d[i])
        assert obj[i] == oi, (i, org[i], d[i])
        if oi >= 0:
            assert bits(t[i]) == bits(ot) and np.array_equal(bits(pos[i]), bits(op)) and np.array_equal(bits(nrm[i]), bits(on)), i
    return obj


def test_closest_hit_cornell(amber, cornell):
    hs, osc = cornell
    pt = amber.PathTracer(hs, amber.Sensor.default(32, 32))
    rng = np.random.default_rng(11)
    n = 3000
    org = rng.uniform(-0.99, 0.99, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    # edge cases: axis-parallel rays (det = 0 for walls), rays along triangle diagonals (u+v = 1), origins on surfaces
    # (kEPS self-hit rejection), rays grazing the sphere tangent points on the floor, zero direction
    org[:6] = [[0, 0, 0], [0, 0, 0], [0.5, -1.0, 0.5], [0.4, -1.0, -0.5], [0, 0, 0], [0, 0.5, 0]]
    d[:6] = [[1, 0, 0], [0, -1, 0], [0, 1, 0], [0, 1, 0], [0.70710678, -0.70710678, 0], [0, 0, 0]]
    obj = _check_casts(pt, osc, org, d)
    assert (obj >= 0).mean() > 0.8                                   # the box is open towards the camera
    # golden vectors
    obj, t, pos, nrm = pt.kat_cast(GOLD["cast_org"], GOLD["cast_dir"])
    assert np.array_equal(obj, GOLD["cast_obj"])
    hit = obj >= 0
    assert np.array_equal(bits(t[hit]), bits(GOLD["cast_t"][hit]))
    assert np.array_equal(bits(pos[hit]), bits(GOLD["cast_pos"][hit])) and np.array_equal(bits(nrm[hit]), bits(GOLD["cast_n"][hit]))


def test_closest_hit_all_primitive_kinds(amber, generic):
    hs, osc = generic
    pt = amber.PathTracer(hs, amber.Sensor.default(32, 32))
    rng = np.random.default_rng(12)
    n = 3000
    org = (rng.uniform(-1.5, 1.5, (n, 3)) + [0, 0.3, 0]).astype(np.float32)
    d = rng.normal(size=(n, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    obj = _check_casts(pt, osc, org, d)
    kinds = {k for k, *_ in [osc.objects()[i] for i in set(obj[obj >= 0].tolist())]}
    assert kinds == {0, 1, 2, 3}                                          # triangles, spheres, disk, cylinder all hit


def test_eye_rays(amber, cornell):
    hs, osc = cornell
    W, H = 80, 56
    pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=99)
    px = np.arange(0, W * H, 7, dtype=np.uint32)
    sm = (px * 2654435761 % 4096).astype(np.uint32)
    eye = pt.kat_eye(px, sm)
    for i in range(0, len(px), 5):
        _, _, e = osc.trace(W, H, 99, int(px[i] % W), int(px[i] // W), int(sm[i]), max_bounces=1)
        assert np.array_equal(bits(eye[i]), bits(e)), i


def test_material_sampling(amber, oracle, generic, cornell):
    """Every material kind incl. total internal reflection, both refraction branches and multi-try Phong."""
    for hs, osc in (cornell, generic):
        pt = amber.PathTracer(hs, amber.Sensor.default(32, 32))
        _, mats, _ = hs.flatten()
        rng = np.random.default_rng(21)
        n = 4000
        mat = rng.integers(0, len(mats), n).astype(np.uint32)
        nrm = rng.normal(size=(n, 3)); nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
        do = rng.normal(size=(n, 3)); do = (do / np.linalg.norm(do, axis=1, keepdims=True)).astype(np.float32)
        do[: n // 4] = (nrm[: n // 4] * 0.05 + do[: n // 4]).astype(np.float32)   # plenty of grazing directions (TIR, Phong rejections)
        do = (do / np.linalg.norm(do, axis=1, keepdims=True)).astype(np.float32)
        state = rng.integers(1, 2 ** 63, n).astype(np.uint64)
        di, w, st = pt.kat_sample(mat, nrm, do, state)
        kinds_seen, multi_try, tir = set(), 0, 0
        for i in range(n):
            m = mats[mat[i]]
            om = O.OMaterial(m.kind, (C.c_float * 3)(*m.rho), m.param)
            u = np.zeros(64, np.float64)
            oracle.oracle_xorshift_uniforms(int(state[i]), 64, u.ctypes.data)
            odi, ow = (C.c_float * 3)(), (C.c_float * 3)()
            used = oracle.oracle_sample_material(C.byref(om), O.f3(nrm[i]), O.f3(do[i]), u.ctypes.data_as(C.POINTER(C.c_double)), 64,
                                                 O.MATH_DEVICE, odi, ow)
            assert used <= 64
            assert np.array_equal(bits(di[i]), bits(list(odi))) and np.array_equal(bits(w[i]), bits(list(ow))), (i, m.kind)
            # the device consumed exactly as many draws as the oracle
            s = int(state[i])
            for _ in range(used):
                s ^= (s << 13) & 0xFFFFFFFFFFFFFFFF; s ^= s >> 7; s ^= (s << 17) & 0xFFFFFFFFFFFFFFFF
            assert s == int(st[i])
            kinds_seen.add(m.kind)
            multi_try += (m.kind == 1 and used > 2)
            tir += (m.kind == 3 and used == 0)
        assert kinds_seen >= {0, 1, 2, 3, 4}
        assert multi_try > 0 and tir > 0


def _compare_traces(pt, osc, W, H, seed, px, sm, maxb=12, max_depth=0):
    rec, casts = pt.kat_trace(px, sm, maxb)
    for i in range(len(px)):
        n, orec, _ = osc.trace(W, H, seed, int(px[i] % W), int(px[i] // W), int(sm[i]), max_depth=max_depth, max_bounces=maxb)
        assert n == casts[i], i
        for b in range(min(n, maxb)):
            assert np.int32(rec[i, b, 0]) == orec[b].object
            if orec[b].object >= 0:
                exp = np.array([orec[b].t, *orec[b].pos, *orec[b].weight, *orec[b].measurement], np.float32)
                assert np.array_equal(rec[i, b, 1:], exp.view(np.uint32)), (i, b)
    return casts


def test_path_traces(amber, cornell, generic):
    for (hs, osc), (W, H) in ((cornell, (72, 40)), (generic, (48, 48))):
        pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=4242)
        rng = np.random.default_rng(31)
        px = rng.integers(0, W * H, 600).astype(np.uint32)
        sm = rng.integers(0, 2 ** 20, 600).astype(np.uint32)
        casts = _compare_traces(pt, osc, W, H, 4242, px, sm)
        assert casts.max() >= 6 and 1.2 < casts.mean() < 4.5                # the generic scene is open: most paths leave early


def test_path_traces_golden(amber, cornell):
    hs, _ = cornell
    W, H, seed = int(GOLD["W"]), int(GOLD["H"]), int(GOLD["seed"])
    pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=seed)
    rec, casts = pt.kat_trace(GOLD["trace_px"], GOLD["trace_sample"], GOLD["trace_rec"].shape[1])
    assert np.array_equal(casts, GOLD["trace_casts"])
    gold = GOLD["trace_rec"]
    assert np.array_equal(rec[:, :, 0], gold[:, :, 0])                      # object index of every bounce (-1 = miss)
    hit = gold[:, :, 0].view(np.int32) >= 0
    assert hit.sum() > 500 and np.array_equal(rec[hit], gold[hit])            # t, position, weight, measurement bits
    assert np.array_equal(bits(pt.kat_eye(GOLD["trace_px"], GOLD["trace_sample"])), bits(GOLD["trace_eye"]))
    pt = amber.PathTracer(hs, amber.Sensor.default(48, 48), seed=seed)
    pt.render_pass(0, 8)
    img, rays = pt.download()
    assert rays == int(GOLD["img48_casts"]) and np.array_equal(bits(img), bits(GOLD["img48_sum"]))


def test_max_depth_extension(amber, cornell):
    hs, osc = cornell
    W = H = 40
    pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=5, max_depth=3)
    px = np.arange(0, W * H, 3, dtype=np.uint32); sm = (px % 11).astype(np.uint32)
    casts = _compare_traces(pt, osc, W, H, 5, px, sm, max_depth=3)
    assert casts.max() == 3


@pytest.mark.parametrize("which,W,H,passes", [("cornell", 128, 96, [(0, 5), (5, 11)]), ("cornell", 67, 45, [(0, 7)]), ("generic", 64, 64, [(3, 70), (73, 40)]), ("cornell", 40, 40, [(0, 100)])])
def test_render_images_bit_exact(amber, cornell, generic, which, W, H, passes):
    """Whole-image parity incl. ragged sizes (not multiples of the 8x8 tile), several passes and sample offsets."""
    hs, osc = cornell if which == "cornell" else generic
    pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=2026)
    ref = np.zeros((H, W, 3), np.float32)
    total = 0
    for first, n in passes:
        pt.render_pass(first, n)
        _, c = osc.render_xorshift(W, H, 2026, first, n, out=ref)
        total += c.casts
    img, rays = pt.download()
    assert rays == total
    assert np.array_equal(bits(img), bits(ref))
    # clear() resets both the framebuffer and the ray counter; a re-render reproduces the image
    pt.clear()
    z, r0 = pt.download()
    assert r0 == 0 and not z.any()
    for first, n in passes:
        pt.render_pass(first, n)
    again, _ = pt.download()
    assert np.array_equal(bits(again), bits(img))


def test_bands_tile_the_image(amber, cornell):
    """Band handles (multi-GPU sharding unit) reproduce the single-handle image exactly, ray counts add up."""
    from amber_amd.distributed import partition_rows
    hs, _ = cornell
    W, H, spp = 96, 100, 6
    sn = amber.Sensor.default(W, H)
    full = amber.PathTracer(hs, sn, seed=8)
    full.render_pass(0, spp)
    img, rays = full.download()
    out, tot = np.zeros_like(img), 0
    for y0, y1 in partition_rows(H, 3):
        band = amber.PathTracer(hs, sn, seed=8, rows=(y0, y1))
        band.render_pass(0, spp)
        b, r = band.download()
        assert b.shape == (y1 - y0, W, 3)
        out[y0:y1] = b; tot += r
    assert tot == rays and np.array_equal(bits(out), bits(img))


def test_algorithm_render_honours_the_context_contract(amber, cornell):
    """Algorithm<RGB>::Render through cli::Context(1, spp): spp passes, mean = sum / passes (accumulator.h:88-95)."""
    hs, osc = cornell
    W, H, spp = 64, 48, 10
    img, st = hs.render(amber.Sensor.default(W, H), spp, seed=77, samples_per_launch=4)    # launches of 4, 4, 2 passes
    assert st["passes"] == spp and st["launches"] == 3
    ref = np.zeros((H, W, 3), np.float32)
    casts = 0
    for first, n in [(0, 4), (4, 4), (8, 2)]:
        _, c = osc.render_xorshift(W, H, 77, first, n, out=ref)
        casts += c.casts
    assert st["rays"] == casts
    assert np.array_equal(bits(img), bits(ref / np.float32(spp)))


def test_full_size_properties(amber, cornell):
    """BASELINE config 2 geometry (1024x1024) at a reduced sample count: properties that need no oracle run.
    determinism, band invariance, rays-per-path and hit statistics of the reference scene."""
    hs, _ = cornell
    W = H = 1024
    spp = 32
    sn = amber.Sensor.default(W, H)
    a = amber.PathTracer(hs, sn, seed=1)
    a.render_pass(0, spp)
    img, rays = a.download()
    a.clear(); a.render_pass(0, spp)
    img2, rays2 = a.download()
    assert rays == rays2 and np.array_equal(bits(img), bits(img2))           # deterministic
    rpp = rays / (W * H * spp)
    assert 2.05 < rpp < 2.14                                                 # reference: 2.094 casts per path
    top = amber.PathTracer(hs, sn, seed=1, rows=(0, 512)); top.render_pass(0, spp)
    bot = amber.PathTracer(hs, sn, seed=1, rows=(512, 1024)); bot.render_pass(0, spp)
    (t, rt), (b, rb) = top.download(), bot.download()
    assert rt + rb == rays and np.array_equal(bits(np.concatenate([t, b])), bits(img))
    assert np.isfinite(img).all() and (img >= 0).all()
    lit = (img.max(2) > 0).mean()
    assert 0.0001 < lit < 0.02                                                # light is hit with probability ~2e-5 per hit
    # spot-check 64 random pixels of the full-size image against the oracle (same seeds, same sums)
    osc = O.Scene.cornell(O.ACCEL_LIST)
    rng = np.random.default_rng(2)
    for x, y in zip(rng.integers(0, W, 64), rng.integers(0, H, 64)):
        s = np.zeros(3, np.float32)
        for k in range(spp):
            n, rec, _ = osc.trace(W, H, 1, int(x), int(y), k, max_bounces=1)
            # measurement after the last hit: trace again with a deep buffer only if the path was long
            if n > 1:
                n, rec, _ = osc.trace(W, H, 1, int(x), int(y), k, max_bounces=n)
            last = None
            for bnc in range(n):
                if rec[bnc].object >= 0:
                    last = rec[bnc]
            if last is not None:
                s = s + np.array(last.measurement[:], np.float32)
        assert np.array_equal(bits(img[y, x]), bits(s))


def test_interleaved_stripes_tile_the_image(amber, cornell):
    """The multi-GPU sharding unit (every N-th stripe of rows) reproduces the single-handle image, ragged sizes included."""
    from amber_amd.distributed import stripe_partition
    hs, _ = cornell
    W, H, spp = 72, 61, 40
    sn = amber.Sensor.default(W, H)
    full = amber.PathTracer(hs, sn, seed=8)
    full.render_pass(0, spp)
    img, rays = full.download()
    for world, stripe in ((3, 4), (8, 8)):
        out, tot = np.full_like(img, np.nan), 0
        for part in stripe_partition(H, world, stripe):
            if len(part["index"]) == 0:
                continue
            pt = amber.PathTracer(hs, sn, seed=8, rows=part["rows"], stripe=part["stripe"])
            assert np.array_equal(pt.row_index, part["index"])
            pt.render_pass(0, spp)
            b, r = pt.download()
            out[part["index"]] = b; tot += r
        assert tot == rays and np.array_equal(bits(out), bits(img))


def test_engines_agree_at_full_size(amber, cornell, generic):
    """LIST (exact test of every object) and TWO_PHASE (conservative filter + exact tests of the candidates) must be
    indistinguishable: same image bits and same ray count on 1.3e8 rays of BASELINE config 2's frame, same per-ray
    answers on random and adversarial rays, and both equal to the oracle on a small frame."""
    hs, osc = cornell
    W = H = 1024
    spp = 64
    sn = amber.Sensor.default(W, H)
    res = {}
    for eng in (amber.ENGINE_LIST, amber.ENGINE_TWO_PHASE):
        pt = amber.PathTracer(hs, sn, seed=3, engine=eng)
        pt.render_pass(0, spp)
        res[eng] = pt.download()
        pt.close()
    assert res[amber.ENGINE_LIST][1] == res[amber.ENGINE_TWO_PHASE][1]
    assert np.array_equal(bits(res[amber.ENGINE_LIST][0]), bits(res[amber.ENGINE_TWO_PHASE][0]))
    rng = np.random.default_rng(77)
    n = 200000
    org = rng.uniform(-1.0, 1.0, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    # adversarial: origins exactly on walls / water plane / sphere surfaces, grazing and axis-parallel directions
    org[:20000, 1] = -1.0; org[20000:40000, 0] = 1.0; org[40000:60000, 1] = -0.5; org[60000:70000, 2] = -1.0
    d[70000:80000, 1] = 0.0; d[80000:90000, 1] *= 1e-4
    c, r = np.array([0.4, -0.6, -0.5], np.float32), np.float32(0.4)
    on_sphere = rng.normal(size=(10000, 3)); on_sphere /= np.linalg.norm(on_sphere, axis=1, keepdims=True)
    org[90000:100000] = (c + r * on_sphere).astype(np.float32)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    for (scene, _), (w, h) in ((cornell, (32, 32)), (generic, (32, 32))):
        a = amber.PathTracer(scene, amber.Sensor.default(w, h), engine=amber.ENGINE_LIST).kat_cast(org, d)
        b = amber.PathTracer(scene, amber.Sensor.default(w, h), engine=amber.ENGINE_TWO_PHASE).kat_cast(org, d)
        assert np.array_equal(a[0], b[0])
        hit = a[0] >= 0
        assert hit.sum() > 50000
        for k in (1, 2, 3):
            assert np.array_equal(bits(a[k][hit]), bits(b[k][hit]))
    # engine LIST against the oracle as well (the rest of this file runs the default engine, TWO_PHASE for these scenes)
    pt = amber.PathTracer(hs, amber.Sensor.default(64, 64), seed=12, engine=amber.ENGINE_LIST)
    pt.render_pass(0, 40)
    img, rays = pt.download()
    ref, cnt = osc.render_xorshift(64, 64, 12, 0, 40)
    assert rays == cnt.casts and np.array_equal(bits(img), bits(ref))


def _mixed_scene(n_spheres, seed):
    """Random spheres (BASELINE config 3 generator) plus a floor, a disk light and a cylinder: all four primitive kinds."""
    from amber_amd import scenes
    k = scenes.random_spheres(n_spheres, seed)
    extra_kinds = np.array([0, 0, 2, 3], np.uint32)
    extra_mat = np.array([1, 1, 0, 2], np.uint32)
    extra = np.zeros((4, 12), np.float32)
    extra[0, :9] = [-3, -1.2, -3, 3, -1.2, 3, 3, -1.2, -3]
    extra[1, :9] = [-3, -1.2, -3, -3, -1.2, 3, 3, -1.2, 3]
    extra[2, :7] = [0, 1.6, 0, 0, -1, 0, 0.8]
    extra[3, :8] = [1.3, -1.2, 0.2, 0, 1, 0, 0.15, 1.0]
    k["kinds"] = np.concatenate([k["kinds"], extra_kinds]); k["material_index"] = np.concatenate([k["material_index"], extra_mat])
    k["params"] = np.concatenate([k["params"], extra])
    return k


def test_bvh_engine_matches_oracle(amber):
    """Engine BVH (host-built flattened BVH, LDS stack) == oracle List semantics on a 3000-object scene."""
    from amber_amd import scenes
    k = _mixed_scene(3000, 11)
    hs = amber.HostScene.create_arrays(**k)
    osc = O.Scene.create(**scenes.as_tuples(k), accel=O.ACCEL_LIST)
    W = H = 40
    pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=21)          # AUTO -> BVH (> 32 objects)
    rng = np.random.default_rng(5)
    n = 1500
    org = rng.uniform(-1.1, 1.1, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    d[:50, 0] = 0.0; d[50:100, 1] = 0.0; d[100:150] = [0, 0, -1]              # zero direction components (inf reciprocals)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    obj = _check_casts(pt, osc, org, d)
    assert (obj >= 0).mean() > 0.7
    px = rng.integers(0, W * H, 300).astype(np.uint32); sm = rng.integers(0, 1000, 300).astype(np.uint32)
    casts = _compare_traces(pt, osc, W, H, 21, px, sm)
    assert casts.max() >= 5
    pt.render_pass(0, 3)
    img, rays = pt.download()
    ref, cnt = osc.render_xorshift(W, H, 21, 0, 3)
    assert rays == cnt.casts and np.array_equal(bits(img), bits(ref)) and (img > 0).any()


def test_bvh_engine_matches_list_engine_on_a_larger_scene(amber):
    k = _mixed_scene(20000, 3)
    hs = amber.HostScene.create_arrays(**k)
    sn = amber.Sensor.default(64, 64)
    rng = np.random.default_rng(9)
    n = 60000
    org = rng.uniform(-1.2, 1.2, (n, 3)).astype(np.float32); org[: n // 3] = [0, 0, 4]
    d = rng.normal(size=(n, 3)); d[: n // 3] = np.array([0, 0, -1.0]) + rng.normal(0, 0.15, (n // 3, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    a = amber.PathTracer(hs, sn, engine=amber.ENGINE_BVH).kat_cast(org, d)
    b = amber.PathTracer(hs, sn, engine=amber.ENGINE_LIST).kat_cast(org, d)
    assert np.array_equal(a[0], b[0])
    hit = a[0] >= 0
    assert hit.mean() > 0.7
    for j in (1, 2, 3):
        assert np.array_equal(bits(a[j][hit]), bits(b[j][hit]))
    # the Cornell box through the BVH engine as well (flat wall boxes, the aperture)
    hs2, osc2 = amber.HostScene.cornell_box(), O.Scene.cornell(O.ACCEL_LIST)
    pt = amber.PathTracer(hs2, amber.Sensor.default(48, 48), seed=5, engine=amber.ENGINE_BVH)
    pt.render_pass(0, 40)
    img, rays = pt.download()
    ref, cnt = osc2.render_xorshift(48, 48, 5, 0, 40)
    assert rays == cnt.casts and np.array_equal(bits(img), bits(ref))


def test_million_spheres_bvh_properties(amber):
    """BASELINE config 3 geometry (1M random spheres, deep BVH) at a reduced frame: determinism and sanity."""
    import time
    from amber_amd import scenes
    t0 = time.time()
    hs = amber.HostScene.create_arrays(**scenes.random_spheres(1_000_000, 7))
    W, H, spp = 480, 270, 8
    pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=1)
    t_build = time.time() - t0
    pt.render_pass(0, spp)
    img, rays = pt.download()
    n, ms = pt.kernel_time()
    pt.clear(); pt.render_pass(0, spp)
    img2, rays2 = pt.download()
    assert rays == rays2 and np.array_equal(bits(img), bits(img2))
    assert np.isfinite(img).all() and (img >= 0).all() and (img > 0).mean() > 0.02
    rpp = rays / (W * H * spp)
    assert 1.5 < rpp < 12
    print(f"\\n1M spheres: scene+BVH build {t_build:.1f} s, {W}x{H}@{spp}: {ms:.1f} ms, {rays / ms / 1e3:.1f} Mrays/s, {rpp:.2f} rays/path")


def test_million_spheres_closest_hits_against_the_oracle_list(amber):
    """BASELINE config 3's own scene (1M spheres): closest hits of engine BVH against the oracle's List acceleration
    (acceleration_list.h:51-68: every sphere tested, 6e8 exact sphere tests on the CPU) -- object, distance, position, normal."""
    from amber_amd import scenes
    k = scenes.random_spheres(1_000_000, 7)
    hs = amber.HostScene.create_arrays(**k)
    osc = O.Scene.create_arrays(**k, accel=O.ACCEL_LIST)
    pt = amber.PathTracer(hs, amber.Sensor.default(64, 64), seed=3)
    rng = np.random.default_rng(77)
    n = 600
    org = rng.uniform(-1.0, 1.0, (n, 3)).astype(np.float32)
    org[: n // 3] = [0.0, 0.0, 4.0]                                             # camera rays: long traversals through the whole cloud
    d = rng.normal(size=(n, 3)); d[: n // 3] = [0, 0, -1] + rng.normal(0, 0.12, (n // 3, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    d[-40:-20, 1] = 0.0; d[-20:] = np.sign(d[-20:]) * [1, 0, 0]                  # zero components / axis-parallel rays
    nz = np.linalg.norm(d, axis=1) > 0
    d[nz] = (d[nz] / np.linalg.norm(d[nz], axis=1, keepdims=True)).astype(np.float32)
    obj = _check_casts(pt, osc, org, d)
    assert (obj >= 0).mean() > 0.9
    # whole paths too: 120 (pixel, sample) pairs, every bounce
    px = rng.integers(0, 64 * 64, 120).astype(np.uint32); sm = rng.integers(0, 256, 120).astype(np.uint32)
    casts = _compare_traces(pt, osc, 64, 64, 3, px, sm)
    assert casts.max() >= 4


def test_million_spheres_at_full_size(amber):
    """BASELINE config 3 at its real size, 1920x1080 @ 256 spp: deterministic, finite, and the path statistics of the scene."""
    from amber_amd import scenes
    hs = amber.HostScene.create_arrays(**scenes.random_spheres(1_000_000, 7))
    W, H, spp = 1920, 1080, 256
    pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=1)
    pt.render_pass(0, spp)
    img, rays = pt.download()
    n, ms = pt.kernel_time()
    pt.clear(); pt.render_pass(0, spp)
    img2, rays2 = pt.download()
    assert rays == rays2 and np.array_equal(bits(img), bits(img2))
    assert np.isfinite(img).all() and (img >= 0).all() and (img > 0).mean() > 0.5
    rpp = rays / (W * H * spp)
    assert 2.4 < rpp < 2.8                                                       # 2.59 rays per path (profiles/r01_config3_bvh.txt)
    # half the resolution at a quarter of the samples estimates the same sensor: the total power (a pixel's value scales
    # with its area, lens_thin.cc:138-148) agrees within 2 %
    small = amber.PathTracer(hs, amber.Sensor.default(W // 2, H // 2), seed=1)
    small.render_pass(0, 64)
    simg, _ = small.download()
    assert abs((img.astype(np.float64).sum() / spp) / (simg.astype(np.float64).sum() / 64) - 1) < 0.02
    print(f"\nconfig 3 full size: {ms / n:.1f} ms per launch, {rays / (ms / n) / 1e3:.1f} Mrays/s, {rpp:.3f} rays/path")


def test_wavefront_engine_is_bit_identical(amber, cornell, generic):
    """Engine WAVEFRONT (SoA ray queues in HBM, one launch per bounce, ballot/prefix-sum compaction) == default engine."""
    for (hs, osc), (W, H, passes) in ((cornell, (96, 80, [(0, 40), (40, 7)])), (generic, (64, 64, [(5, 33)]))):
        sn = amber.Sensor.default(W, H)
        a = amber.PathTracer(hs, sn, seed=17, engine=amber.ENGINE_WAVEFRONT)
        b = amber.PathTracer(hs, sn, seed=17)
        ref = np.zeros((H, W, 3), np.float32); casts = 0
        for first, n in passes:
            a.render_pass(first, n); b.render_pass(first, n)
            _, c = osc.render_xorshift(W, H, 17, first, n, out=ref); casts += c.casts
        (ia, ra), (ib, rb) = a.download(), b.download()
        assert ra == rb == casts
        assert np.array_equal(bits(ia), bits(ib)) and np.array_equal(bits(ia), bits(ref))
    hs, _ = cornell
    sn = amber.Sensor.default(512, 512)
    a = amber.PathTracer(hs, sn, seed=2, engine=amber.ENGINE_WAVEFRONT, rows=(8, 512), stripe=(8, 16)); a.render_pass(0, 64)
    b = amber.PathTracer(hs, sn, seed=2, rows=(8, 512), stripe=(8, 16)); b.render_pass(0, 64)
    (ia, ra), (ib, rb) = a.download(), b.download()
    assert ra == rb and np.array_equal(bits(ia), bits(ib))


def test_pinhole_lens(amber):
    """MakePinholeLens (lens_pinhole.cc:31-106): eye rays, path traces and an image against the oracle."""
    pin = dict(SCENE, n_blades=0, focal_length=0.045)          # n_blades = 0 selects the pinhole; focal_length = sensor distance
    hs, osc = amber.HostScene.create(**pin), O.Scene.create(**pin)
    objs, mats, lens = hs.flatten()
    assert lens.kind == 1 and lens.n_blades == 1 and np.float32(lens.sensor_distance) == np.float32(0.045)
    W, H = 56, 40
    pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=6)
    px = np.arange(0, W * H, 3, dtype=np.uint32); sm = (px % 7).astype(np.uint32)
    eye = pt.kat_eye(px, sm)
    for i in range(0, len(px), 9):
        _, _, e = osc.trace(W, H, 6, int(px[i] % W), int(px[i] // W), int(sm[i]), max_bounces=1)
        assert np.array_equal(bits(eye[i]), bits(e))
    assert np.all(eye[:, :3] == np.array(lens.origin[:], np.float32))   # every ray leaves the pinhole itself
    _compare_traces(pt, osc, W, H, 6, px[:400], sm[:400])
    pt.render_pass(0, 48)
    img, rays = pt.download()
    ref, cnt = osc.render_xorshift(W, H, 6, 0, 48)
    assert rays == cnt.casts and np.array_equal(bits(img), bits(ref)) and (img > 0).any()


def test_baseline_configs_4_and_5_geometry(amber, cornell):
    """BASELINE configs 4 (2048x2048, 4-GPU shard) and 5 (3840x2160, max depth 16, 8-GPU shard) at a reduced sample
    count: one rank's stripe set rendered alone equals the same rows of the full frame, ray counts add up over ranks,
    max_depth truncation never exceeds 16 casts per path and only lowers the ray count."""
    from amber_amd.distributed import stripe_partition
    hs, osc = cornell
    for (W, H, world, depth, spp) in ((2048, 2048, 4, 0, 4), (3840, 2160, 8, 16, 2)):
        sn = amber.Sensor.default(W, H)
        full = amber.PathTracer(hs, sn, seed=4, max_depth=depth)
        full.render_pass(0, spp)
        img, rays = full.download()
        full.close()
        assert np.isfinite(img).all()
        parts = stripe_partition(H, world)
        tot = 0
        for r in (0, world - 1):
            pt = amber.PathTracer(hs, sn, seed=4, max_depth=depth, rows=parts[r]["rows"], stripe=parts[r]["stripe"])
            pt.render_pass(0, spp)
            b, rr = pt.download(); pt.close()
            assert np.array_equal(bits(b), bits(img[parts[r]["index"]]))
            tot += rr
        assert 0 < tot < rays
        if depth:
            free = amber.PathTracer(hs, sn, seed=4); free.render_pass(0, spp)
            rays_free = free.ray_count(); free.close()
            assert rays < rays_free                                             # truncation only removes tails
            px = np.arange(0, W * H, 9973, dtype=np.uint32); sm = (px % spp).astype(np.uint32)
            pt = amber.PathTracer(hs, sn, seed=4, max_depth=depth)
            _, casts = pt.kat_trace(px, sm, depth)
            assert casts.max() <= depth


def test_bvh_bounds_are_conservative_enough(amber, cornell):
    """The BVH may only cull what the exact tests would reject.  (1) Cornell 1024x1024@64 through engine BVH equals the
    default engine on all 1.4e8 rays.  (2) On the 1M-sphere scene the image must not change when the sphere bounds are
    made 16x more conservative (AMBER_BVH_SPHERE_SLACK test hook): the reference's binary32 discriminant accepts rays
    that miss a small distant sphere geometrically, and bounds derived from the geometry alone lose those hits."""
    import os
    from amber_amd import scenes
    hs, _ = cornell
    sn = amber.Sensor.default(1024, 1024)
    res = []
    for eng in (amber.ENGINE_AUTO, amber.ENGINE_BVH):
        pt = amber.PathTracer(hs, sn, seed=6, engine=eng); pt.render_pass(0, 64); res.append(pt.download()); pt.close()
    assert res[0][1] == res[1][1] and np.array_equal(bits(res[0][0]), bits(res[1][0]))
    k = scenes.random_spheres(1_000_000, 7)
    out = []
    for slack in (None, "256"):
        if slack: os.environ["AMBER_BVH_SPHERE_SLACK"] = slack
        try:
            hb = amber.HostScene.create_arrays(**k)
            pt = amber.PathTracer(hb, amber.Sensor.default(960, 540), seed=1); pt.render_pass(0, 64); out.append(pt.download()); pt.close()
        finally:
            os.environ.pop("AMBER_BVH_SPHERE_SLACK", None)
    assert out[0][1] == out[1][1] and np.array_equal(bits(out[0][0]), bits(out[1][0]))


def test_edge_sizes_and_engine_thresholds(amber, cornell):
    """Ragged and minimal inputs: 1x1 and 1xN frames, zero samples, sample offsets near 2^32, a band of one row,
    scenes of exactly 32 objects (last two-phase size) and 33 objects (first BVH size), an all-miss scene."""
    hs, osc = cornell
    for (W, H) in ((1, 1), (1, 9), (13, 1), (9, 9)):
        pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=3)
        pt.render_pass(0, 0)                                              # no samples: a no-op
        z, r0 = pt.download()
        assert r0 == 0 and not z.any()
        first = 2 ** 32 - 21
        pt.render_pass(first, 20)                                         # highest legal sample indices
        img, rays = pt.download()
        ref, cnt = osc.render_xorshift(W, H, 3, first, 20)
        assert rays == cnt.casts and np.array_equal(bits(img), bits(ref))
        with pytest.raises(amber.AmberError, match="overflow"):
            pt.render_pass(2 ** 32 - 5, 10)
    pt = amber.PathTracer(hs, amber.Sensor.default(40, 30), seed=3, rows=(17, 18))     # a one-row band
    pt.render_pass(0, 24)
    b, r = pt.download()
    ref, cnt = osc.render_xorshift(40, 30, 3, 0, 24, rows=(17, 18))
    assert b.shape == (1, 40, 3) and r == cnt.casts and np.array_equal(bits(b[0]), bits(ref[17]))
    with pytest.raises(amber.AmberError, match="bad row band"):
        amber.PathTracer(hs, amber.Sensor.default(40, 30), rows=(10, 31))
    # 32 vs 33 objects: 6 blades + n spheres in a row
    base = dict(materials=[(4, (5.0, 5.0, 5.0), 0.0), (0, (0.6, 0.6, 0.6), 0.0)], transform=[1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 5, 0, 0, 0, 1],
                focal_length=0.05, focus_distance=5.0, radius=0.02, n_blades=6)
    for n_sph, engine_is_bvh in ((26, False), (27, True)):
        objs = [(1, i % 2, [-1.3 + 0.1 * i, 0.1 * np.sin(i), 0.0, 0.045]) for i in range(n_sph)]
        sc = dict(base, objects=objs)
        h2, o2 = amber.HostScene.create(**sc), O.Scene.create(**sc)
        assert len(h2.flatten()[0]) == 6 + n_sph
        pt = amber.PathTracer(h2, amber.Sensor.default(48, 32), seed=1)
        pt.render_pass(0, 16)
        img, rays = pt.download()
        ref, cnt = o2.render_xorshift(48, 32, 1, 0, 16)
        assert rays == cnt.casts and np.array_equal(bits(img), bits(ref)) and (img > 0).any()
    # nothing in front of the camera: every path is one cast and a miss
    sc = dict(base, objects=[(1, 1, [0.0, 0.0, 50.0, 1.0])])
    h3 = amber.HostScene.create(**sc)
    pt = amber.PathTracer(h3, amber.Sensor.default(16, 16), seed=1); pt.render_pass(0, 8)
    img, rays = pt.download()
    assert rays == 16 * 16 * 8 and not img.any()


def test_light_tracing(amber, cornell):
    """rendering::LightTracing (algorithm_lt.cc:112-163) on the same kernels: light sampling over all four primitive kinds
    (LightSet order, SampleSurfacePoint, HemispherePSA), SampleImportance incl. refraction, lens Response and the splat
    records, bit for bit against the oracle; then the Algorithm::Render adapter (`--algorithm lt`)."""
    lights = dict(
        materials=[(4, (30.0, 20.0, 10.0), 0.0), (0, (0.7, 0.6, 0.5), 0.0), (2, (0.8, 0.8, 0.8), 0.0), (3, (1.0, 1.0, 1.0), 1.5), (4, (5.0, 5.0, 9.0), 0.0)],
        objects=[
            (2, 0, [0.0, 1.5, 0.0, 0.0, -1.0, 0.0, 0.6]),                      # disk light
            (0, 4, [-1.0, 1.4, -1.0, -1.0, 1.4, 1.0, -0.5, 1.4, 0.0]),        # triangle light
            (1, 0, [1.2, 0.8, 0.0, 0.15]),                                     # sphere light
            (3, 4, [-1.4, -0.5, 0.5, 0.0, 1.0, 0.0, 0.1, 0.6]),               # cylinder light
            (0, 1, [-3, -1, -3, 3, -1, 3, 3, -1, -3]), (0, 1, [-3, -1, -3, -3, -1, 3, 3, -1, 3]),
            (1, 2, [0.7, -0.6, -0.3, 0.4]), (1, 3, [0.0, -0.5, 0.8, 0.45]),
        ],
        transform=[1, 0, 0, 0, 0, 1, 0, 0.2, 0, 0, 1, 2.6, 0, 0, 0, 1], focal_length=0.05, focus_distance=2.6, radius=0.45, n_blades=5,
    )
    hs, osc = amber.HostScene.create(**lights), O.Scene.create(**lights)
    W, H = 48, 36
    pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=13)
    rec, rays = pt.lt_trace(2, 400)
    ref_img, cnt, oref = osc.render_lt(W, H, 13, 2, 400)
    assert rays == cnt.casts and len(rec) == len(oref) and len(rec) > 30    # only light that lands on the 36 mm sensor counts
    got = np.stack([rec["path"], rec["sample"], rec["bounce"], rec["pixel"], *[rec["rgb"][:, c].view(np.uint32) for c in range(3)]], 1)
    assert np.array_equal(got, oref)
    assert rec["bounce"].max() >= 3 and len(np.unique(rec["pixel"])) > 20
    # engine BVH and LIST produce the same records (closest hit is engine independent)
    for eng in (amber.ENGINE_LIST, amber.ENGINE_BVH):
        r2, rays2 = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=13, engine=eng).lt_trace(2, 400)
        assert rays2 == rays and np.array_equal(r2, rec)
    # adapter: mean image over passes, pass images summed in order
    img, st = hs.render(amber.Sensor.default(W, H), 150, seed=13, samples_per_launch=64, algorithm="lt")
    oimg, ocnt, _ = osc.render_lt(W, H, 13, 0, 150)
    assert st["passes"] == 150 and st["rays"] == ocnt.casts
    assert np.array_equal(bits(img), bits(oimg / np.float32(150))) and (img > 0).any()
    # the Cornell box: its 5 cm aperture is hit by ~1e-4 of the light paths; whatever arrives must match exactly
    hc, oc = cornell
    pt = amber.PathTracer(hc, amber.Sensor.default(64, 64), seed=3)
    rec, rays = pt.lt_trace(0, 48)
    _, cnt, oref = oc.render_lt(64, 64, 3, 0, 48)
    assert rays == cnt.casts and len(rec) == len(oref)
    if len(rec):
        got = np.stack([rec["path"], rec["sample"], rec["bounce"], rec["pixel"], *[rec["rgb"][:, c].view(np.uint32) for c in range(3)]], 1)
        assert np.array_equal(got, oref)


def test_fuzz_regressions(amber):
    """Scenes from tools/fuzz_engines.py that once broke an engine: disks with non-unit normals give directions of
    non-unit length, for which the reference's sphere test is not geometric (seed 5: engine BVH culled such hits) and
    for which its Phong rejection loop never ends (seeds 1037, 1039: bounded at AMBER_PHONG_MAX_TRIES in engine and
    oracle); plus a few ordinary seeds.  All engines and the oracle must agree bit for bit."""
    from fuzz_scenes import scene_for_seed
    W, H, spp = 48, 40, 6
    # 31296 (--extreme): a disk normal of length ~40 sends a path to coordinates of 1e16 and back; the two-phase filter's
    # tolerances do not hold out there and the engine has to bypass it.  209769 (--extreme --scaled): a scene 1e4 of its
    # size away from the world origin; the filter's affine maps must work in centred coordinates.
    # 308053 (--scaled --heavy): a needle triangle (edge ratio 1e4); the reference's u + v test accepts rays 2.5e-3 beyond
    # its short edge, outside the geometric box: engine BVH widens needle boxes by that reach.
    # 520075 (--extreme --scaled, round 3): a path that CARRIES a NaN measurement across bounces; since pt_megakernel's primary rounds a
    # ray changes lanes (parked, popped by another lane) and its carried measurement has to travel with it.
    # 209769 also caught round 3's per-pixel candidate masks: an aperture of radius 0.01 at world coordinates of 1.7e4 is smaller than
    # the binary32 grid there, so a ray's origin lies in several blades at once for the exact test (DevLens.edge_tol).
    for seed, scaled, extreme, heavy in ((5, 0, 0, 0), (1037, 0, 0, 0), (1039, 0, 0, 0), (2, 0, 0, 0), (11, 0, 0, 0), (16, 0, 0, 0), (40, 0, 0, 0), (31296, 0, 1, 0),
                                        (209769, 1, 1, 0), (308053, 1, 0, 1), (520075, 1, 1, 0)):
        big = seed % 4 == 3
        W, H, spp = (128, 96, 12) if heavy else (48, 40, 6)
        sc, _ = scene_for_seed(seed, scaled=bool(scaled), extreme=bool(extreme))
        hs = amber.HostScene.create(**sc)
        n_obj = len(sc["objects"]) + max(1, sc["n_blades"])
        engines = [amber.ENGINE_LIST, amber.ENGINE_BVH, amber.ENGINE_WAVEFRONT] + ([amber.ENGINE_TWO_PHASE] if n_obj <= 128 else [])
        ref = None
        for e in engines:
            pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=seed, engine=e)
            pt.render_pass(0, spp)
            img, rays = pt.download(); pt.close()
            if ref is None:
                ref = (bits(img).copy(), rays)
            assert rays == ref[1] and np.array_equal(bits(img), ref[0]), (seed, e)
        if not big:
            oimg, cnt = O.Scene.create(**sc).render_xorshift(W, H, seed, 0, spp)
            assert cnt.casts == ref[1] and np.array_equal(bits(oimg), ref[0]), seed
