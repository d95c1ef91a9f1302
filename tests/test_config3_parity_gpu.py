"""BASELINE config 3 (1M random spheres, 1920x1080 @ 256 spp) on the GPU against an oracle with the ENGINE'S semantics.

Every engine implements the reference's List acceleration (/root/reference/include/amber/raytracer/acceleration_list.h:51-68).  The
reference's BVH (acceleration_bvh.h:340-403) is a different function of the ray on this scene: primitive_sphere.cc:75-107 accepts rays
that pass slightly OUTSIDE a sphere (binary32 discriminant), its BVH culls with the geometric box and loses those hits, its List keeps
them.  oracle(ACCEL_BVH_CONS) evaluates List at BVH speed (proved equal to the plain scan in tests/test_oracle_conservative_bvh.py, on
the CPU), so here the comparison is exact:

  (i)   a 64-row band of the real frame at the real 256 samples: image bits and ray count GPU == oracle(List via conservative BVH);
  (ii)  the pixels on which the GPU differs from oracle(reference BVH) are EXACTLY the pixels on which the oracle's two accelerations
        differ from each other;
  (iii) every one of those pixels is attributed: it holds a path on whose first differing cast the plain List scan over all 1M spheres
        returns what the GPU's semantics returned, and the reference BVH lost that hit because the accepted hit lies outside the
        sphere's geometric box (Primitive::BoundingBox through the reference's own slab test, aabb.cc:28-62: the ray misses the box, or
        enters it only behind the accepted distance) or resolved an exact distance tie differently -- nothing else.
tools/config3_full_parity.py runs the same three steps on the whole frame (profiles/r04_config3_full_parity.txt).
"""
import numpy as np
import pytest

import oracle_binding as O
from amber_amd import scenes

pytestmark = pytest.mark.gpu
W, H, SEED, SPP = 1920, 1080, 1, 256


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def config3(amber):
    kw = scenes.random_spheres(1_000_000, 7)
    return amber.HostScene.create_arrays(**kw), O.Scene.create_arrays(**kw, accel=O.ACCEL_BVH_CONS)


def classify_pixels(osc, pixels, spp=SPP, seed=SEED):
    """For every (y, x): the paths of the pixel on which oracle(List via conservative BVH) and oracle(reference BVH) part ways, each with
    the cause of its first differing cast.  Returns {(y, x): [dict, ...]}."""
    out = {}
    for y, x in pixels:
        found = []
        for k in range(spp):
            c = osc.classify_path(W, H, seed, int(x), int(y), k, O.ACCEL_BVH_CONS, O.ACCEL_BVH, math=O.MATH_LIBM)
            if c is None:
                continue
            assert c["object_a"] == c["object_list"] and c["t_bits"][0] == c["t_bits"][2], (y, x, k, c)   # conservative BVH == the plain scan
            if c["exact_tie"]:
                c["cause"] = "exact distance tie between two objects"
            elif c["object_list"] >= 0 and not c["list_object_box_hit"]:
                c["cause"] = "lost hit: the reference's sphere test accepts a ray that misses the sphere's geometric box"
            elif c["object_list"] >= 0 and not c["list_hit_inside_box"]:
                c["cause"] = "lost hit: the accepted distance lies in front of the point where the ray enters the sphere's geometric box"
            else:
                c["cause"] = "unexplained"
            c["sample"] = k
            found.append(c)
        out[(int(y), int(x))] = found
    return out


def test_config3_band_at_256spp_equals_the_list_oracle_and_every_difference_from_the_reference_bvh_is_attributed(amber, config3):
    hs, osc = config3
    rows = (508, 572)
    pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=SEED, rows=rows)
    pt.render_pass(0, SPP)
    img, rays = pt.download()
    pt.close()
    full = np.zeros((H, W, 3), np.float32)
    _, cnt = osc.set_accel(O.ACCEL_BVH_CONS).render_xorshift(W, H, SEED, 0, SPP, math=O.MATH_LIBM, threads=16, rows=rows, out=full)
    cons = full[rows[0]:rows[1]]
    # (i)
    assert rays == cnt.casts, (rays, cnt.casts)
    assert int((bits(img) != bits(cons)).any(axis=2).sum()) == 0
    # (ii)
    full_b = np.zeros((H, W, 3), np.float32)
    _, cnt_b = osc.set_accel(O.ACCEL_BVH).render_xorshift(W, H, SEED, 0, SPP, math=O.MATH_LIBM, threads=16, rows=rows, out=full_b)
    ref = full_b[rows[0]:rows[1]]
    gpu_vs_ref = (bits(img) != bits(ref)).any(axis=2)
    cons_vs_ref = (bits(cons) != bits(ref)).any(axis=2)
    assert np.array_equal(gpu_vs_ref, cons_vs_ref)
    # (iii)
    pixels = [(y + rows[0], x) for y, x in zip(*np.nonzero(gpu_vs_ref))]
    causes = classify_pixels(osc, pixels)
    n_paths = 0
    for px, found in causes.items():
        assert found, f"pixel {px} differs but no path of it does"
        for c in found:
            assert c["cause"] != "unexplained", (px, c)
        n_paths += len(found)
    print(f"\nconfig 3 rows {rows} @ {SPP} spp: {rays} rays, GPU == oracle(List) on all {img.shape[0] * W} pixels; against the reference's BVH "
          f"{len(pixels)} pixels differ ({rays - cnt_b.casts:+d} rays), every one attributed ({n_paths} paths: "
          f"{sum(c['cause'].startswith('lost hit') for f in causes.values() for c in f)} lost grazing hits, "
          f"{sum(c['cause'].startswith('exact') for f in causes.values() for c in f)} ties)")
    assert len(pixels) > 0                                                 # the band was chosen to contain some


def test_config3_product_kernel_signatures_equal_the_list_oracle(amber, config3):
    """algorithm_pt.cc:125-160 path by path on config 3's scene, from the kernel that renders it (pt_bvh_megakernel with the hashing on):
    hit-object sequence and hit distances of every path of eight rows at 16 spp == oracle(List via conservative BVH) == the per-thread
    known-answer kernel; the second scheduler (pt_bvh_pool_kernel) as well."""
    hs, osc = config3
    osc.set_accel(O.ACCEL_BVH_CONS)
    for rows in ((536, 540), (20, 24)):
        so = osc.path_signatures(W, H, SEED, 0, 16, rows, threads=16)
        for flags in (0, amber.api.PT_FLAG_BVH_POOL):
            pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=SEED, rows=rows, flags=flags)
            sg = pt.render_signatures(0, 16)
            assert np.array_equal(sg, so), (rows, flags, int((sg != so).sum()))
            if flags == 0:
                assert np.array_equal(sg, pt.kat_signatures(0, 16))
            pt.close()
        assert len(np.unique(so & np.uint64(0xffffffff))) > 1000           # thousands of different hit sequences


def test_config3_closest_hits_equal_the_list_oracle_on_many_rays(amber, config3):
    """a5 on the 1M-sphere scene at scale: 60 000 rays of the configuration's own paths (all bounces) + 20 000 rays aimed at sphere
    rims, through the known-answer cast kernel: object and distance bits == oracle(List via conservative BVH)."""
    from test_oracle_conservative_bvh import _rim_rays
    hs, osc = config3
    osc.set_accel(O.ACCEL_BVH_CONS)
    o1, d1 = osc.collect_rays(W, H, SEED, 0, 2, (300, 306), 60_000)
    o2, d2 = _rim_rays(scenes.random_spheres(1_000_000, 7), 20_000, 9)
    o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    io, to = osc.cast_many(o, d, O.ACCEL_BVH_CONS, threads=16)
    pt = amber.PathTracer(hs, amber.Sensor.default(64, 64), seed=3)
    obj, t, _, _ = pt.kat_cast(o, d)
    pt.close()
    hit = io >= 0
    assert np.array_equal(obj, io)
    assert np.array_equal(bits(t)[hit], bits(to)[hit])
