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

from bvh_parity import bits, check_band, check_reference_engine


@pytest.fixture(scope="module")
def config3(amber):
    kw = scenes.random_spheres(1_000_000, 7)
    return amber.HostScene.create_arrays(**kw), O.Scene.create_arrays(**kw, accel=O.ACCEL_BVH_CONS)


BAND = (508, 572)


@pytest.fixture(scope="module")
def band(amber, config3):
    hs, osc = config3
    # hard bound against the reference's own BVH; the full frame has 117 such pixels of 2 073 600 (profiles/r04_config3_full_parity.txt)
    return check_band(amber, hs, osc, W, H, SEED, SPP, BAND, max_ref_pixels=32, max_ref_ray_delta=512, label="config 3")      # measured on this band: 15 pixels, -132 rays


def test_config3_band_at_256spp_equals_the_list_oracle_and_every_difference_from_the_reference_bvh_is_attributed(band):
    assert band["differ_from_reference_bvh"] > 0                            # the band was chosen to contain some


def test_config3_band_through_the_references_own_tree_equals_the_reference_bvh_oracle(amber, config3, band):
    """AMBER_ENGINE_REFERENCE_BVH: the same band, now including the pixels of the test above -- what the reference's code renders through its BVH
    (acceleration_bvh.h:134-403), bit for bit."""
    check_reference_engine(amber, config3[0], W, H, SEED, SPP, BAND, band["reference_image"], band["reference_casts"], label="config 3")


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
