"""GPU tests added in round 4 (through the C ABI; the oracle is the checker):

  * BASELINE config 1's workload (Cornell 256x256 @ 16 spp, the reference's own CPU-runnable case) rendered on the GPU against the oracle --
    the configuration no longer lacks a GPU leg;
  * the per-pixel candidate masks of the primary rays (pixel_mask_kernel, 16 lanes per pixel since this round) are supersets of what
    the pixel's eye rays actually hit, on the Cornell box, on stripes and on fuzz scenes; the kernel's duration is bounded;
  * launches of every size give the image of one launch (a wave claims 1024 paths at a time, whatever is left of the queue);
  * one rank's stripes of BASELINE configs 4 and 5 at their real frame and samples per launch against the oracle (round 5).
"""
import numpy as np
import pytest

import oracle_binding as O
from fuzz_scenes import scene_for_seed

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_config1_workload_on_the_gpu(amber):
    """BASELINE configs[0]: Cornell box 256x256 @ 16 spp (`make pt`, CMakeLists.txt:86-98; application.cc:68-94).  Image bits and ray count
    against oracle(XorShift, List, live libm); against the reference's BVH nothing differs either unless a path meets an exact tie."""
    W = H = 256
    spp, seed = 16, 12345
    img, stats = amber.HostScene.cornell_box().render(amber.Sensor.default(W, H), spp, seed=seed)      # through Algorithm<RGB>::Render
    ref, cnt = O.Scene.cornell(O.ACCEL_LIST).render_xorshift(W, H, seed, 0, spp, math=O.MATH_LIBM, threads=16)
    assert stats["rays"] == cnt.casts
    assert np.array_equal(bits(img), bits(ref / np.float32(spp)))
    # against the reference's own BVH: only pixels that hold an exact-tie path may differ, each one classified (tests/bvh_parity.py)
    from bvh_parity import classify_pixels
    osc = O.Scene.cornell(O.ACCEL_BVH_CONS)
    refb, cntb = osc.set_accel(O.ACCEL_BVH).render_xorshift(W, H, seed, 0, spp, math=O.MATH_LIBM, threads=16)
    differing = (bits(ref) != bits(refb)).any(axis=2)
    causes = classify_pixels(osc, W, H, seed, list(zip(*np.nonzero(differing))), spp)
    for px, found in causes.items():
        assert found and all(c["cause"].startswith("exact distance tie") for c in found), (px, found)
    print(f"\nconfig 1 vs the reference's BVH: {int(differing.sum())} pixels differ (exact ties), ray count {cnt.casts} vs {cntb.casts}")


def _first_hits(pt, W, pixels, samples):
    rec, casts = pt.kat_trace(pixels, samples, 1)
    return rec[:, 0, 0].view(np.int32)


def _check_masks(amber, hs, n_objects, W, H, seed, rows=None, stripe=None, n=60_000):
    pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=seed, rows=rows, stripe=stripe)
    masks, ms = pt.pixel_masks()
    slots, always_mask = pt.object_slots(n_objects)
    rng = np.random.default_rng(seed)
    local = rng.integers(0, masks.size, n)
    lrow, x = local // W, local % W
    y = pt.row_index[lrow]
    pix = (y * W + x).astype(np.uint32)
    sm = rng.integers(0, 4096, n).astype(np.uint32)
    obj = _first_hits(pt, W, pix, sm)
    hit = obj >= 0
    slot = slots[obj[hit]]
    m = masks.reshape(-1)[local[hit]] | np.uint32(always_mask)                     # blades (added per ray) and objects without a filter record
    covered = ((m >> np.minimum(slot, 31).astype(np.uint32)) & 1).astype(bool) & (slot < 32)
    bad = ~covered
    assert not bad.any(), (int(bad.sum()), obj[hit][bad][:5], pix[hit][bad][:5])
    pt.close()
    return masks, ms


def test_pixel_masks_cover_every_primary_hit(amber):
    """The mask of a pixel must contain the closest hit of every eye ray of the pixel (then skipping Phase A for primary rays changes
    nothing): 60 000 random (pixel, sample) pairs per case, first hits from the known-answer trace kernel."""
    hs = amber.HostScene.cornell_box()
    masks, ms = _check_masks(amber, hs, 25, 1024, 1024, 1)
    print(f"\npixel_mask_kernel, config 2's frame: {ms:.3f} ms; mean candidates per pixel {np.mean([bin(int(v)).count('1') for v in masks.reshape(-1)[::97]]):.2f}")
    assert ms > 0                                                          # (its duration is bench.py's business: cold_start)
    _check_masks(amber, hs, 25, 3840, 2160, 2, rows=(8, 2160), stripe=(8, 64), n=30_000)      # rank 1 of 8, config 5's frame
    _check_masks(amber, hs, 25, 333, 77, 3, n=20_000)                        # odd sizes: the last group of 16 lanes is partly idle
    for seed in range(910_000, 910_030):
        kw, _ = scene_for_seed(seed, scaled=seed % 3 == 1)
        if len(kw["objects"]) + kw["n_blades"] > 32 or seed % 4 == 3:
            continue
        hsf = amber.HostScene.create(**kw)
        _check_masks(amber, hsf, len(kw["objects"]) + max(1, kw["n_blades"]), 96, 64, seed, n=6_000)


def test_launch_size_does_not_change_the_image(amber):
    """Whatever the split into launches and whatever their size -- down to a band with fewer paths than the grid has waves -- the image and
    the ray count are those of one launch (and of the oracle)."""
    hs = amber.HostScene.cornell_box()
    W, H, seed = 200, 120, 77
    ref, cnt = O.Scene.cornell(O.ACCEL_LIST).render_xorshift(W, H, seed, 0, 96, threads=16)
    for plan in ([96], [8, 88], [8, 8, 16, 64], [32, 32, 32]):
        pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=seed)
        first = 0
        for n in plan:
            pt.render_pass(first, n); first += n
        img, rays = pt.download()
        assert rays == cnt.casts and np.array_equal(bits(img), bits(ref)), plan
        pt.close()
    # a one-row band of a large frame: fewer paths than waves
    pt = amber.PathTracer(hs, amber.Sensor.default(1024, 1024), seed=seed, rows=(600, 601))
    pt.render_pass(0, 24)
    img, rays = pt.download()
    full = np.zeros((1024, 1024, 3), np.float32)
    _, c1 = O.Scene.cornell(O.ACCEL_LIST).render_xorshift(1024, 1024, seed, 0, 24, threads=16, rows=(600, 601), out=full)
    assert rays == c1.casts and np.array_equal(bits(img), bits(full[600:601]))


@pytest.mark.parametrize("name,W,H,world,depth,y0", [("config 4", 2048, 2048, 4, 0, 1408), ("config 5", 3840, 2160, 8, 16, 1472)])
def test_one_ranks_stripes_of_configs_4_and_5_at_their_real_size(amber, name, W, H, world, depth, y0):
    """BASELINE configs 4 (Cornell 2048x2048 on 4 GPUs) and 5 (3840x2160, max depth 16, on 8 GPUs): four consecutive 8-row stripes of rank 0's
    share (amber_amd.distributed.stripe_partition: every world-th stripe), through the spheres and the water, at the REAL frame and the real
    1024 samples of a launch, against oracle(XorShift, List, live libm): image bits and ray count (the whole share:
    tools/full_share_parity.py, profiles/r04_configs45_share_parity.txt)."""
    from amber_amd.distributed import stripe_partition
    spp, seed = 1024, 12345
    part = stripe_partition(H, world)[0]
    period = 8 * world
    assert part["stripe"] == (8, period) and y0 % period == 0
    pt = amber.PathTracer(amber.HostScene.cornell_box(), amber.Sensor.default(W, H), seed=seed, max_depth=depth, rows=(y0, y0 + 4 * period), stripe=(8, period))
    assert [int(y) for y in pt.row_index[::8]] == [y0 + k * period for k in range(4)] and set(pt.row_index) <= set(part["index"])
    pt.render_pass(0, spp)
    img, rays = pt.download()
    pt.close()
    osc = O.Scene.cornell(O.ACCEL_LIST)
    full = np.zeros((H, W, 3), np.float32)
    casts = 0
    for k in range(4):
        _, cnt = osc.render_xorshift(W, H, seed, 0, spp, math=O.MATH_LIBM, max_depth=depth, threads=16, rows=(y0 + k * period, y0 + k * period + 8), out=full)
        casts += cnt.casts
    assert rays == casts, (name, rays, casts)
    assert np.array_equal(bits(img), bits(full[pt.row_index])), name
    assert rays > 2 * 32 * W * spp                                         # spheres and water: more than two casts per path
    print(f"\n{name}: rank 0's stripes at rows {[y0 + k * period for k in range(4)]} of {W}x{H} @ {spp} spp" + (f", max depth {depth}" if depth else "") + f": {rays} rays, bit-identical to the oracle")
