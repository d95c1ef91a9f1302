"""GPU parity tests added in round 3 (all through the C ABI, all bit-exact against the oracle):

  * the PRODUCT render kernels' own path signatures (amber_hip_pt_signatures) against the oracle's, path by path;
  * BASELINE config 2's whole 1024x1024 frame at 64 spp against oracle(XorShift, reference BVH, live libm);
  * BASELINE config 5's max_depth = 16 against the oracle on full-width rows of the 3840x2160 frame;
  * path records (accumulation without owners): scenes where most paths end on a light, launches that need the density probe,
    launches that run out of record slots and are repeated;
  * engine BVH's second scheduler (AMBER_PT_FLAG_BVH_POOL, pt_bvh_pool_kernel) against the default and the oracle;
  * light tracing sharded by light-path index (N handles) and long pass ranges (more than 2^31 work units);
  * time-adaptive launch batches of Algorithm::Render;
  * 200 fresh fuzz scenes per round (the first seed advances with tests/golden/fuzz_round.json).
"""
import json
from pathlib import Path

import numpy as np
import pytest

import oracle_binding as O
from test_gpu_parity import _mixed_scene, bits

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent


@pytest.fixture(scope="module")
def cornell(amber):
    return amber.HostScene.cornell_box(), O.Scene.cornell(O.ACCEL_LIST)


def _split(sig):
    return sig & np.uint64(0xffffffff), sig >> np.uint64(32)


def test_product_kernel_signatures_against_the_oracle(amber, cornell):
    """algorithm_pt.cc:125-160, path by path, from the kernels that RENDER: pt_megakernel (two-phase and list), pt_bvh_megakernel
    and pt_bvh_pool_kernel instantiated with the hashing on -- same work queue, ray pool and device functions as the product
    instantiation.  Object sequence and hit distances of every path == oracle (List), == the per-thread KAT kernel."""
    hs, osc = cornell
    W = H = 256
    rows, spp, seed = (100, 108), 64, 12345
    so = osc.path_signatures(W, H, seed, 0, spp, rows, threads=16)
    for engine, flags in ((amber.ENGINE_TWO_PHASE, 0), (amber.ENGINE_LIST, 0), (amber.ENGINE_BVH, 0), (amber.ENGINE_BVH, amber.api.PT_FLAG_BVH_ITEMS), (amber.ENGINE_BVH, amber.api.PT_FLAG_BVH_POOL)):
        pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=seed, rows=rows, engine=engine, flags=flags)
        pt.render_pass(0, 8)                                           # state left by a render must not matter
        img0, rays0 = pt.download()
        sp = pt.render_signatures(0, spp)
        img1, rays1 = pt.download()
        assert rays0 == rays1 and np.array_equal(bits(img0), bits(img1))   # the signature launch leaves framebuffer and ray count alone
        assert np.array_equal(sp, so), engine
        assert np.array_equal(sp, pt.kat_signatures(0, spp)), engine
        # a later sample range, an odd count
        assert np.array_equal(pt.render_signatures(1000, 13), osc.path_signatures(W, H, seed, 1000, 13, rows, threads=16)), engine
        pt.close()
    # a scene of all four primitive kinds and every material, through the pool kernel (triangle hits carry u, v through LDS)
    from amber_amd import scenes
    k = _mixed_scene(3000, 11)
    hm, om = amber.HostScene.create_arrays(**k), O.Scene.create(**scenes.as_tuples(k), accel=O.ACCEL_LIST)
    sm = om.path_signatures(40, 40, 21, 0, 24, (0, 40), threads=16)
    for flags in (amber.api.PT_FLAG_BVH_POOL, 0):                      # ... and through engine BVH's default kernel (pt_bvh_megakernel)
        pt = amber.PathTracer(hm, amber.Sensor.default(40, 40), seed=21, flags=flags)
        assert np.array_equal(pt.render_signatures(0, 24), sm), flags
        pt.close()
    # BASELINE config 3's own scene and kernel: tests/test_config3_parity_gpu.py (against an oracle with List semantics at BVH speed)


def test_config2_whole_frame_at_64spp_against_the_oracle(amber, cornell):
    """BASELINE config 2's frame, every one of the 1024 rows, 64 samples per pixel (6.7e7 paths, 1.4e8 rays), with the
    host's live libm:
      * against the oracle's List acceleration (acceleration_list.h:51-68, the semantics the engine implements): image bits
        and ray count, exactly;
      * against the oracle's restatement of the reference's BVH (acceleration_bvh.h:340-403): identical except for the paths
        that meet an EXACT distance tie between two objects (a ray through the shared edge of two triangles), where the BVH
        keeps the first-visited leaf's hit and List the lower index -- a handful of paths in 6.7e7; when the two objects have
        different materials such a path continues differently (SURVEY.md Appendix C lists exact ties as a permitted
        difference).  CLASSIFIED here, pixel by pixel (tests/bvh_parity.py): every differing pixel holds a path whose first differing cast
        is an exact tie, and nothing else differs."""
    from bvh_parity import classify_pixels
    hs, _ = cornell
    W = H = 1024
    spp, seed = 64, 12345
    pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=seed)
    pt.render_pass(0, spp)
    img, rays = pt.download()
    ref, cnt = O.Scene.cornell(O.ACCEL_LIST).render_xorshift(W, H, seed, 0, spp, math=O.MATH_LIBM, threads=16)
    assert rays == cnt.casts, (rays, cnt.casts)
    assert np.array_equal(bits(img), bits(ref))
    osc = O.Scene.cornell(O.ACCEL_BVH_CONS)
    refb, cntb = osc.set_accel(O.ACCEL_BVH).render_xorshift(W, H, seed, 0, spp, math=O.MATH_LIBM, threads=16)
    differing = (bits(img) != bits(refb)).any(axis=2)                     # == oracle(List) != oracle(BVH): the GPU image IS oracle(List)'s, bit for bit
    causes = classify_pixels(osc, W, H, seed, list(zip(*np.nonzero(differing))), spp)
    for px, found in causes.items():
        assert found and all(c["cause"].startswith("exact distance tie") for c in found), (px, found)
    print(f"\nconfig 2 frame @ {spp} spp vs the reference's BVH: {int(differing.sum())} of {W * H} pixels differ, every one an exact-tie path "
          f"({sum(len(f) for f in causes.values())} paths); ray count {rays} vs {cntb.casts} ({rays - cntb.casts:+d})")


def test_config5_max_depth_16_against_the_oracle(amber, cornell):
    """BASELINE config 5 (3840x2160, max depth 16): 8 full-width rows at 16 spp against oracle.render_xorshift(max_depth=16)
    -- image bits, ray count, path signatures -- and per-bounce traces with the same truncation."""
    from test_gpu_parity import _compare_traces
    hs, osc = cornell
    W, H, depth, spp, seed = 3840, 2160, 16, 16, 12345
    sn = amber.Sensor.default(W, H)
    for rows in ((1076, 1080), (1500, 1504)):                          # light row / spheres and water (long refraction chains)
        pt = amber.PathTracer(hs, sn, seed=seed, max_depth=depth, rows=rows)
        pt.render_pass(0, spp)
        img, rays = pt.download()
        full = np.zeros((H, W, 3), np.float32)
        _, cnt = osc.render_xorshift(W, H, seed, 0, spp, max_depth=depth, threads=16, rows=rows, out=full)
        assert rays == cnt.casts and np.array_equal(bits(img), bits(full[rows[0]:rows[1]]))
        assert np.array_equal(pt.render_signatures(0, spp), osc.path_signatures(W, H, seed, 0, spp, rows, max_depth=depth, threads=16))
        _, cnt_free = osc.render_xorshift(W, H, seed, 0, spp, threads=16, rows=rows)
        assert cnt.casts < cnt_free.casts                               # the truncation bites on these rows
        pt.close()
    pt = amber.PathTracer(hs, sn, seed=seed, max_depth=depth)
    rng = np.random.default_rng(16)
    px = (rng.integers(1400, 1700, 500) * W + rng.integers(0, W, 500)).astype(np.uint32)
    sm = rng.integers(0, 8192, 500).astype(np.uint32)
    casts = _compare_traces(pt, osc, W, H, seed, px, sm, maxb=depth, max_depth=depth)
    assert casts.max() == depth


LIGHT_BOX = dict(
    # a room whose ceiling and two walls are lights: most paths end on one
    materials=[(4, (3.0, 2.0, 1.0), 0.0), (0, (0.7, 0.7, 0.7), 0.0), (2, (0.9, 0.9, 0.9), 0.0), (3, (1.0, 1.0, 1.0), 1.5), (4, (0.5, 1.5, 2.5), 0.0)],
    objects=[
        (0, 0, [-2, 1.5, -2, 2, 1.5, -2, 2, 1.5, 2]), (0, 0, [-2, 1.5, -2, 2, 1.5, 2, -2, 1.5, 2]),            # ceiling light (normal -y: facing down)
        (0, 4, [-2, -1, -2, 2, 1.5, -2, -2, 1.5, -2]), (0, 4, [-2, -1, -2, 2, -1, -2, 2, 1.5, -2]),            # back wall light (normal +z)
        (0, 1, [-2, -1, -2, 2, -1, 2, 2, -1, -2]), (0, 1, [-2, -1, -2, -2, -1, 2, 2, -1, 2]),                  # floor
        (1, 2, [0.6, -0.5, 0.0, 0.5]), (1, 3, [-0.7, -0.55, 0.4, 0.45]), (1, 4, [0.0, 0.9, 0.0, 0.3]),
        (2, 0, [-1.9, 0.2, 0.0, 1.0, 0.0, 0.0, 0.9]), (3, 1, [1.5, -1.0, -1.0, 0.0, 1.0, 0.0, 0.2, 1.2]),
    ],
    transform=[1, 0, 0, 0, 0, 1, 0, 0.1, 0, 0, 1, 3.2, 0, 0, 0, 1], focal_length=0.05, focus_distance=3.2, radius=0.02, n_blades=6,
)


def test_path_records_when_most_paths_reach_a_light(amber):
    """Accumulation without owners under load: > 50 % of the paths end with a non-zero measurement, so every wave appends
    records all the time.  Small frame: all engines against the oracle.  Larger frame (2.8e7 paths per pass range): the first
    launch is the one-chunk density probe, the rest is sized from it; repeated renders and split passes stay bit-identical."""
    hs, osc = amber.HostScene.create(**LIGHT_BOX), O.Scene.create(**LIGHT_BOX)
    W, H, spp = 96, 64, 40
    ref, cnt = osc.render_xorshift(W, H, 5, 0, spp)
    assert (ref.sum(axis=2) > 0).mean() > 0.9
    lit_paths = None
    for engine, flags in ((amber.ENGINE_TWO_PHASE, 0), (amber.ENGINE_LIST, 0), (amber.ENGINE_BVH, 0), (amber.ENGINE_BVH, amber.api.PT_FLAG_BVH_ITEMS), (amber.ENGINE_BVH, amber.api.PT_FLAG_BVH_POOL),
                          (amber.ENGINE_WAVEFRONT, 0)):
        pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=5, engine=engine, flags=flags)
        pt.render_pass(0, spp)
        img, rays = pt.download()
        assert rays == cnt.casts and np.array_equal(bits(img), bits(ref)), (engine, flags)
        pt.close()
    # fraction of paths that carry a measurement: from single-sample renders of the oracle
    one, _ = osc.render_xorshift(W, H, 5, 0, 1, chunk=1)
    lit_paths = (one.sum(axis=2) > 0).mean()
    assert lit_paths > 0.5, lit_paths
    # the large frame: probe launch + density-sized launches, against the oracle and against itself
    W, H, spp = 768, 768, 48
    sn = amber.Sensor.default(W, H)
    ref, cnt = osc.render_xorshift(W, H, 7, 0, spp, threads=16)
    for engine, flags in ((amber.ENGINE_AUTO, 0), (amber.ENGINE_BVH, amber.api.PT_FLAG_BVH_POOL)):
        pt = amber.PathTracer(hs, sn, seed=7, engine=engine, flags=flags)
        pt.render_pass(0, spp)
        img, rays = pt.download()
        n_launch, _ = pt.kernel_time()
        assert n_launch >= 2                                            # the probe chunk, then the rest
        assert rays == cnt.casts and np.array_equal(bits(img), bits(ref)), (engine, flags)
        pt.clear()
        pt.render_pass(0, 16); pt.render_pass(16, 32)                   # split on a chunk boundary: the same sums
        img2, rays2 = pt.download()
        assert rays2 == rays and np.array_equal(bits(img2), bits(img))
        pt.close()


def test_a_launch_that_runs_out_of_record_slots_is_repeated(amber, cornell):
    """The record buffer is sized from the PREVIOUS launch's density.  A handle that has only seen the Cornell box's dark
    rows (no path reaches the light) and then renders... the same handle cannot change scene, so the density is made wrong
    the other way round: a first launch over sample indices whose paths all miss the light, then a launch 100x longer.  Whatever
    the sizing, a launch that needs more slots than it was given must leave no trace and be repeated: image and ray count equal
    the oracle's."""
    hs, osc = amber.HostScene.create(**LIGHT_BOX), O.Scene.create(**LIGHT_BOX)
    W, H = 640, 512
    sn = amber.Sensor.default(W, H)
    # rows that look at the floor only see light by reflection; rows at the top look straight at the ceiling light: a band
    # handle over the whole frame whose first launch is ONE sample (density of that one sample), then 63 more
    import os
    os.environ["AMBER_TEST_RECORD_DENSITY_SCALE"] = "0.02"              # test hook (read once, at create): pretend the measured density was 50x lower
    try:
        pt = amber.PathTracer(hs, sn, seed=11)
        pt.render_pass(0, 8)
        pt.render_pass(8, 56)                                           # sized 50x too small: runs out of slots, is repeated
        img, rays = pt.download()
    finally:
        os.environ.pop("AMBER_TEST_RECORD_DENSITY_SCALE", None)
    n_launch, _ = pt.kernel_time()
    assert n_launch >= 3                                                # probe, the failed launch, its repetition
    ref, cnt = osc.render_xorshift(W, H, 11, 0, 64, threads=16)
    assert rays == cnt.casts and np.array_equal(bits(img), bits(ref))


def test_bvh_pool_scheduler_is_bit_identical(amber):
    """Engine BVH's two schedulers -- lanes own work items (pt_bvh_megakernel, the default) / per-wave ray pool with in-place
    swaps (pt_bvh_pool_kernel, AMBER_PT_FLAG_BVH_POOL) -- against each other and the oracle: all primitive kinds, several
    passes, sample offsets, a one-row band, stripes; then the 1M-sphere scene of BASELINE config 3 (deep stacks: the levels
    beyond the LDS part live in global memory) and fuzz scenes with degenerate directions (carried NaN measurements)."""
    from amber_amd import scenes
    from fuzz_scenes import scene_for_seed
    POOL = amber.api.PT_FLAG_BVH_POOL
    k = _mixed_scene(3000, 11)
    hs = amber.HostScene.create_arrays(**k)
    osc = O.Scene.create(**scenes.as_tuples(k), accel=O.ACCEL_LIST)
    W, H = 61, 45
    sn = amber.Sensor.default(W, H)
    pool, dflt = amber.PathTracer(hs, sn, seed=21, flags=POOL), amber.PathTracer(hs, sn, seed=21)
    ref = np.zeros((H, W, 3), np.float32); casts = 0
    for first, n in ((0, 5), (5, 11), (1000, 3)):
        pool.render_pass(first, n); dflt.render_pass(first, n)
        _, c = osc.render_xorshift(W, H, 21, first, n, out=ref); casts += c.casts
    (ip, rp), (idf, rd) = pool.download(), dflt.download()
    assert rp == rd == casts and np.array_equal(bits(ip), bits(idf)) and np.array_equal(bits(ip), bits(ref))
    band = amber.PathTracer(hs, sn, seed=21, flags=POOL, rows=(17, 18)); band.render_pass(0, 9)
    b, _ = band.download()
    rb, _ = osc.render_xorshift(W, H, 21, 0, 9, rows=(17, 18))
    assert np.array_equal(bits(b[0]), bits(rb[17]))
    st = amber.PathTracer(hs, sn, seed=21, flags=POOL, rows=(8, H), stripe=(8, 16)); st.render_pass(0, 9)
    s_img, _ = st.download()
    full, _ = osc.render_xorshift(W, H, 21, 0, 9)
    assert np.array_equal(bits(s_img), bits(full[st.row_index]))
    # BASELINE config 3's scene at a reduced frame
    hm = amber.HostScene.create_arrays(**scenes.random_spheres(1_000_000, 7))
    sm = amber.Sensor.default(480, 270)
    a, b2 = amber.PathTracer(hm, sm, seed=1, flags=POOL), amber.PathTracer(hm, sm, seed=1)
    a.render_pass(0, 16); b2.render_pass(0, 16)
    (ia, ra), (ib, rb2) = a.download(), b2.download()
    assert ra == rb2 and np.array_equal(bits(ia), bits(ib)) and (ia > 0).mean() > 0.02
    a.close(); b2.close()
    # degenerate fuzz scenes (non-unit normals: NaN / inf weights travel along the path)
    for seed, scaled, extreme in ((5, 0, 0), (1037, 0, 0), (31296, 0, 1), (209769, 1, 1), (7, 0, 1), (23, 0, 1)):
        sc, _ = scene_for_seed(seed, scaled=bool(scaled), extreme=bool(extreme))
        hf = amber.HostScene.create(**sc)
        res = []
        for eng, fl in ((amber.ENGINE_LIST, 0), (amber.ENGINE_BVH, 0), (amber.ENGINE_BVH, amber.api.PT_FLAG_BVH_ITEMS), (amber.ENGINE_BVH, POOL)):
            pt = amber.PathTracer(hf, amber.Sensor.default(48, 40), seed=seed, engine=eng, flags=fl)
            pt.render_pass(0, 6); res.append(pt.download()); pt.close()
        for img, rays in res[1:]:
            assert rays == res[0][1] and np.array_equal(bits(img), bits(res[0][0])), seed


def test_light_tracing_shards_by_path_index_and_splits_long_ranges(amber, cornell):
    """(a) amber_hip_lt_trace_range: the splats of disjoint light-path ranges, merged, are the splats of the whole pass.
    (b) HipLightTracing::Render with several handles (here on one device) == one handle, bit for bit (algorithm_lt.cc:82-95
    parallelises lt like pt).  (c) a range of passes with more than 2^31 (light path, pass) units is traced in several
    launches instead of failing."""
    lights = dict(
        materials=[(4, (30.0, 20.0, 10.0), 0.0), (0, (0.7, 0.6, 0.5), 0.0), (2, (0.8, 0.8, 0.8), 0.0), (3, (1.0, 1.0, 1.0), 1.5)],
        objects=[(2, 0, [0.0, 1.5, 0.0, 0.0, -1.0, 0.0, 0.6]), (1, 0, [1.2, 0.8, 0.0, 0.15]),
                 (0, 1, [-3, -1, -3, 3, -1, 3, 3, -1, -3]), (0, 1, [-3, -1, -3, -3, -1, 3, 3, -1, 3]),
                 (1, 2, [0.7, -0.6, -0.3, 0.4]), (1, 3, [0.0, -0.5, 0.8, 0.45])],
        transform=[1, 0, 0, 0, 0, 1, 0, 0.2, 0, 0, 1, 2.6, 0, 0, 0, 1], focal_length=0.05, focus_distance=2.6, radius=0.45, n_blades=5)
    hs, osc = amber.HostScene.create(**lights), O.Scene.create(**lights)
    W, H = 48, 36
    sn = amber.Sensor.default(W, H)
    pt = amber.PathTracer(hs, sn, seed=13)
    whole, rays = pt.lt_trace(2, 200)
    parts, tot = [], 0
    for p0, p1 in ((0, 500), (500, 501), (501, W * H)):
        rec, r = pt.lt_trace(2, 200, paths=(p0, p1))
        assert len(rec) == 0 or (rec["path"].min() >= p0 and rec["path"].max() < p1)
        parts.append(rec); tot += r
    merged = np.concatenate(parts)
    merged = merged[np.lexsort((merged["bounce"], merged["path"], merged["sample"]))]
    assert tot == rays and len(whole) > 20 and merged.tobytes() == whole.tobytes()
    for engine in (amber.ENGINE_AUTO, amber.ENGINE_BVH):
        one, st1 = hs.render(sn, 96, seed=13, samples_per_launch=40, algorithm="lt", devices=[0])
        many, stn = hs.render(sn, 96, seed=13, samples_per_launch=40, algorithm="lt", devices=[0, 0, 0])
        assert st1["rays"] == stn["rays"] and st1["passes"] == stn["passes"] == 96
        assert np.array_equal(bits(one), bits(many)) and (one > 0).any()
    oimg, ocnt, _ = osc.render_lt(W, H, 13, 0, 96)
    assert st1["rays"] == ocnt.casts and np.array_equal(bits(one), bits(oimg / np.float32(96)))
    # (c) 1920 x 1080 light paths x 1100 passes = 2.28e9 units > 2^31: several launches inside one call
    hc, _ = cornell
    big = amber.PathTracer(hc, amber.Sensor.default(1920, 1080), seed=3)
    rec, rays_big = big.lt_trace(0, 1100, capacity=1 << 20)
    a, ra = big.lt_trace(0, 1096, capacity=1 << 20)
    b, rb = big.lt_trace(1096, 4, capacity=1 << 20)
    assert rays_big == ra + rb and len(rec) == len(a) + len(b) and rec.tobytes() == np.concatenate([a, b]).tobytes()
    assert rays_big > 2 ** 31


def test_render_batches_adapt_to_time(amber, cornell):
    """HipPathTracingOptions.samples_per_launch = 0 (the default): Algorithm::Render grows its batches while they are short (at least doubling, at most to the measured 100-ms batch).
    All batches are multiples of the accumulation chunk, so the image is that of one launch -- bit for bit -- and of the oracle."""
    hs, osc = cornell
    W, H, spp = 96, 64, 200
    auto, st = hs.render(amber.Sensor.default(W, H), spp, seed=9)                          # samples_per_launch = 0: adaptive
    one, st1 = hs.render(amber.Sensor.default(W, H), spp, seed=9, samples_per_launch=spp)
    assert st["passes"] == st1["passes"] == spp and st["rays"] == st1["rays"]
    assert 2 <= st["launches"] < 20                                                          # 8, then what the measured rate says fills 100 ms (here: the rest): not 25 launches of 8
    assert np.array_equal(bits(auto), bits(one))
    ref, cnt = osc.render_xorshift(W, H, 9, 0, spp)
    assert st["rays"] == cnt.casts and np.array_equal(bits(auto), bits(ref / np.float32(spp)))


def test_fresh_fuzz_scenes(amber):
    """200 fuzz scenes nobody has rendered before: the first seed comes from tests/golden/fuzz_round.json, which every round
    advances.  The flag set rotates over plain / --scaled / --extreme / --heavy; every engine that accepts the scene (LIST, BVH
    with both schedulers, WAVEFRONT, TWO_PHASE) must agree bit for bit, light tracing between the work-queue engines on every
    4th scene, and the oracle on three of every eight scenes (one each of plain, --scaled, --extreme)."""
    from fuzz_scenes import scene_for_seed
    cfg = json.loads((ROOT / "golden" / "fuzz_round.json").read_text())
    first = int(cfg["first_seed"])
    n = int(cfg.get("n_scenes", 200))
    POOL = amber.api.PT_FLAG_BVH_POOL
    for seed in range(first, first + n):
        mode = seed % 4
        scaled, extreme, heavy = mode == 1, mode == 2, (mode == 3 and seed % 16 == 3)
        W, H, spp = (128, 96, 12) if heavy else (48, 40, 6)
        sc, rng = scene_for_seed(seed, scaled=scaled, extreme=extreme)
        n_obj = len(sc["objects"]) + max(1, sc["n_blades"])
        hs = amber.HostScene.create(**sc)
        engines = [(amber.ENGINE_LIST, 0), (amber.ENGINE_BVH, 0), (amber.ENGINE_BVH, amber.api.PT_FLAG_BVH_ITEMS), (amber.ENGINE_BVH, POOL), (amber.ENGINE_WAVEFRONT, 0)] + ([(amber.ENGINE_TWO_PHASE, 0)] if n_obj <= 128 else [])
        ref = None
        for e, fl in engines:
            pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=seed, engine=e, flags=fl)
            pt.render_pass(0, spp)
            img, rays = pt.download(); pt.close()
            if ref is None:
                ref = (bits(img).copy(), rays)
            assert rays == ref[1] and np.array_equal(bits(img), ref[0]), (seed, e, fl)
        if seed % 4 == 0:
            lref = None
            for e in [x for x, fl in engines if x != amber.ENGINE_WAVEFRONT and fl == 0]:
                pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=seed, engine=e)
                rec, lrays = pt.lt_trace(0, 3, capacity=1 << 14); pt.close()
                if lref is None:
                    lref = (rec.tobytes(), lrays)
                assert lrays == lref[1] and rec.tobytes() == lref[0], (seed, e)
        if seed % 8 < 3:                                               # plain, --scaled and --extreme scenes of the small kind
            oimg, cnt = O.Scene.create(**sc).render_xorshift(W, H, seed, 0, spp)
            assert cnt.casts == ref[1] and np.array_equal(bits(oimg), ref[0]), seed


def test_traversal_on_its_own_gives_the_same_closest_hits(amber):
    """bvh_trace_rate_kernel (engine BVH's resumable traversal in a kernel that does nothing else: the measurement of DESIGN.md
    section 5) against amber_hip_kat_cast, at every occupancy and at three refill thresholds, incl. rays with zero components,
    NaN rays and misses, on a scene of all four primitive kinds."""
    k = _mixed_scene(20000, 3)
    hs = amber.HostScene.create_arrays(**k)
    pt = amber.PathTracer(hs, amber.Sensor.default(64, 64), engine=amber.ENGINE_BVH)
    rng = np.random.default_rng(9)
    n = 50000
    org = rng.uniform(-1.3, 1.3, (n, 3)).astype(np.float32); org[: n // 4] = [0, 0, 4]
    d = rng.normal(size=(n, 3)); d[: n // 4] = np.array([0, 0, -1.0]) + rng.normal(0, 0.2, (n // 4, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    d[-30:-20, 0] = 0.0; d[-20:-10] = [0, 1, 0]; d[-10:-5] = np.nan; org[-5:] = [50, 50, 50]
    ref_obj, ref_t, _, _ = pt.kat_cast(org, d)
    hit = ref_obj >= 0
    assert 0.5 < hit.mean() < 1.0
    for waves in (4, 5, 6, 8):
        for refill in (1, 16, 64):
            obj, t, ms = pt.kat_traversal_rate(org, d, waves=waves, refill_min=refill, repeats=2)
            assert np.array_equal(obj, ref_obj) and np.array_equal(bits(t[hit]), bits(ref_t[hit])) and ms > 0, (waves, refill)


def test_rays_almost_parallel_to_an_axis_stay_cheap_and_exact(amber):
    """Round 3 found engine BVH's slab slack shared by the three axes: a ray with |d.y| = 4e-7 (the middle rows of a frame) or a zero
    component switched off the culling of the other axes and walked every box of the sheet it lies in -- 40 000 nodes, 8 000 wave rounds on
    the 1M-sphere scene against a median of 10.  The slack is per axis now and a parallel axis stays in the test with a clamped 1/d
    (dev_bvh.h BvhOperands).  Closest hits must equal the List scan's bit for bit, and no ray may be in flight for more than 100 rounds."""
    from amber_amd import scenes
    kw = scenes.random_spheres(200_000, 7)
    hs = amber.HostScene.create_arrays(**kw)
    bvh = amber.PathTracer(hs, amber.Sensor.default(1920, 1080), seed=1, engine=amber.ENGINE_BVH)
    lst = amber.PathTracer(hs, amber.Sensor.default(1920, 1080), seed=1, engine=amber.ENGINE_LIST)
    rng = np.random.default_rng(21)
    n = 6000
    org = rng.uniform(-1.0, 1.0, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    tiny = np.array([0.0, -0.0, 4e-7, -4e-7, 1e-12, -1e-13, 1e-20, 1e-30, 1e-39, -1e-42], np.float64)      # incl. denormals: 1/d overflows
    ax = rng.integers(0, 3, n)
    d[np.arange(n), ax] = tiny[rng.integers(0, len(tiny), n)]
    two = rng.random(n) < 0.2                                                    # a fifth of them parallel to two axes at once
    d[two, (ax[two] + 1) % 3] = tiny[rng.integers(0, len(tiny), int(two.sum()))]
    d /= np.linalg.norm(d, axis=1, keepdims=True)                               # unit length again (a direction far from unit length widens
    d = d.astype(np.float32)                                                    #  every box on purpose: BvhOperands' slack_len)
    # plus the eye rays of the frame's middle row and column (|d.y| or |d.x| of a pixel pitch and below)
    px = np.concatenate([540 * 1920 + np.arange(0, 1920, 2), np.arange(0, 1080, 2) * 1920 + 960]).astype(np.uint32)
    eye = bvh.kat_eye(np.repeat(px, 4), np.tile(np.arange(4, dtype=np.uint32), len(px)))
    org = np.ascontiguousarray(np.concatenate([org, eye[:, 0:3]]), np.float32); d = np.ascontiguousarray(np.concatenate([d, eye[:, 3:6]]), np.float32)
    ref_obj, ref_t, _, _ = lst.kat_cast(org, d)
    obj, t, _, _ = bvh.kat_cast(org, d)
    hit = ref_obj >= 0
    assert 0.3 < hit.mean() < 1.0
    assert np.array_equal(obj, ref_obj) and np.array_equal(bits(t[hit]), bits(ref_t[hit]))
    rounds = np.zeros(len(org), np.uint32)
    obj2, t2, _ = bvh.kat_traversal_rate(org, d, waves=5, refill_min=16, repeats=1, rounds=rounds)
    assert np.array_equal(obj2, ref_obj) and np.array_equal(bits(t2[hit]), bits(ref_t[hit]))
    assert rounds.max() <= 100, (int(rounds.max()), org[rounds.argmax()], d[rounds.argmax()])
    assert np.median(rounds) <= 15


def test_candidate_masks_of_the_primary_rays_change_nothing(amber, cornell):
    """pt_megakernel's per-pixel candidate masks (round 3: a primary round skips the candidate filter and tests the pixel's own
    candidates) against the same kernel without them (AMBER_PIXEL_MASK=0 at create): images, ray counts and the product kernel's
    path signatures, on the Cornell box (thin lens; striped rows; sample counts that make generation rounds straddle pixels; a
    second launch at an offset), on a pinhole camera and on the all-primitive test scene."""
    import os
    from test_host_model import SCENE

    def pair(make):
        os.environ["AMBER_PIXEL_MASK"] = "0"
        try:
            off = make()
        finally:
            os.environ["AMBER_PIXEL_MASK"] = "1"
        try:
            return off, make()
        finally:
            del os.environ["AMBER_PIXEL_MASK"]

    hs, _ = cornell
    cases = [
        ("cornell 96x80, 24 + 40 spp", lambda: amber.PathTracer(hs, amber.Sensor.default(96, 80), seed=3, engine=amber.ENGINE_TWO_PHASE), ((0, 24), (24, 40))),
        ("cornell 256x256 stripes 1 of 3, 64 spp", lambda: amber.PathTracer(hs, amber.Sensor.default(256, 256), seed=5, rows=(8, 256), stripe=(8, 24), engine=amber.ENGINE_TWO_PHASE), ((7, 64),)),
        ("pinhole 56x40, 48 spp", lambda: amber.PathTracer(amber.HostScene.create(**dict(SCENE, n_blades=0, focal_length=0.045)), amber.Sensor.default(56, 40), seed=6, engine=amber.ENGINE_TWO_PHASE), ((0, 48),)),
        ("all primitive kinds 64x48, 33 spp", lambda: amber.PathTracer(amber.HostScene.create(**SCENE), amber.Sensor.default(64, 48), seed=9, engine=amber.ENGINE_TWO_PHASE), ((2, 33),)),
    ]
    for name, make, launches in cases:
        off, on = pair(make)
        for first, n in launches:
            off.render_pass(first, n); on.render_pass(first, n)
        a, ra = off.download(); b, rb = on.download()
        assert ra == rb and np.array_equal(bits(a), bits(b)) and (a > 0).any(), name
        first, n = launches[0]
        assert np.array_equal(off.render_signatures(first, n), on.render_signatures(first, n)), name
        off.close(); on.close()


def test_config2_at_full_size_against_the_oracle(amber, cornell):
    """BASELINE config 2 as the bench renders it -- Cornell 1024x1024 @ 1024 spp, one launch, 2.25e9 rays -- against oracle(XorShift, List,
    live libm) on EVERY pixel and EVERY sample: image bits and the ray count (half a minute of oracle time on the box's 16 host threads;
    profiles/r03_full_frame_parity.txt also has the oracle's reference-BVH run: 0 pixels differ, 163 rays of exact-tie paths)."""
    hs, _ = cornell
    osc = O.Scene.cornell(O.ACCEL_LIST)
    W = H = 1024
    pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=12345)
    pt.render_pass(0, 1024)
    img, rays = pt.download()
    ref, cnt = osc.render_xorshift(W, H, 12345, 0, 1024, math=O.MATH_LIBM, threads=16)
    assert rays == cnt.casts == 2248938869
    assert np.array_equal(bits(img), bits(ref)) and 8000 < (ref > 0).any(axis=2).sum() < 9000
    pt.close()
