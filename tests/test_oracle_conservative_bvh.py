"""The oracle's conservative-BVH mode (ORACLE_ACCEL_BVH_CONS, oracle/amber_oracle.cc BVH::CastCons) IS the List scan.

The engine implements the reference's List semantics (/root/reference/include/amber/raytracer/acceleration_list.h:51-68: the closest
hit over all objects, the lower index on a tie).  On the 1M-sphere scene of BASELINE config 3 the plain scan costs a millisecond per
ray, and the reference's own BVH (acceleration_bvh.h:340-403) is a DIFFERENT function of the ray there: it culls with geometric boxes
while primitive_sphere.cc:75-107 accepts rays that pass slightly outside the sphere.  ACCEL_BVH_CONS evaluates List through the
reference's tree; the GPU parity tests of config 3 compare against it.  This file is its proof, on the CPU:

  * 1M spheres: equal to the plain scan -- object and distance bits -- on > 1e5 rays of the configuration's own paths (eye rays and
    every later bounce, from the top, the middle and the bottom of the frame) and on 4e4 rays aimed at the RIM of a sphere, where the
    two reference accelerations part ways; the reference BVH differs from the scan on some of them (counted, must be > 0);
  * the fuzzer's scenes (all four primitive kinds, needles, non-unit disk normals, scaled and offset copies): images, ray counts and
    path signatures equal to the scan's.
"""
import numpy as np
import pytest

import oracle_binding as O
from amber_amd import scenes
from fuzz_scenes import scene_for_seed


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def million():
    return O.Scene.create_arrays(**scenes.random_spheres(1_000_000, 7), accel=O.ACCEL_BVH_CONS), scenes.random_spheres(1_000_000, 7)


def _rim_rays(kw, n, seed):
    """Rays from points in and around the scene towards the silhouette of a random sphere, offset from the rim by a factor
    1 + k with |k| from 1e-8 to 1e-2 on either side: the rays on which a geometric box and the binary32 discriminant disagree."""
    rng = np.random.default_rng(seed)
    which = rng.integers(0, len(kw["params"]), n)
    c, r = kw["params"][which, :3].astype(np.float64), kw["params"][which, 3].astype(np.float64)
    o = np.where(rng.random((n, 1)) < 0.5, rng.uniform(-1, 1, (n, 3)), rng.uniform(-1, 1, (n, 3)) * 0.1 + np.array([0, 0, 4.0]))
    to_c = c - o
    dist = np.linalg.norm(to_c, axis=1, keepdims=True)
    u = to_c / dist
    side = np.cross(u, rng.standard_normal((n, 3)))
    side /= np.linalg.norm(side, axis=1, keepdims=True)
    k = 10.0 ** rng.uniform(-8, -2, n) * rng.choice([-1.0, 1.0], n)
    rim = (r * (1.0 + k))[:, None]                                        # impact parameter
    d = u * np.sqrt(np.maximum(dist ** 2 - rim ** 2, 0.0)) + side * rim   # tangent-ish direction (not normalised yet)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d *= (1.0 + rng.uniform(-3e-6, 3e-6, (n, 1)))                         # the drift sampled directions carry
    return o.astype(np.float32), d.astype(np.float32)


def test_conservative_bvh_equals_the_list_scan_on_the_million_sphere_scene(million):
    sc, kw = million
    W, H, seed = 1920, 1080, 1
    sets = []
    for rows in ((8, 10), (538, 541), (1070, 1072)):                     # 7 rows x 1920 px x 3 spp x ~2.6 casts
        sc.set_accel(O.ACCEL_BVH_CONS)
        sets.append(sc.collect_rays(W, H, seed, 0, 3, rows, 60_000))
    sets.append(_rim_rays(kw, 40_000, 5))
    o, d = np.concatenate([s[0] for s in sets]), np.concatenate([s[1] for s in sets])
    assert len(o) >= 140_000
    il, tl = sc.cast_many(o, d, O.ACCEL_LIST)
    ic, tc = sc.cast_many(o, d, O.ACCEL_BVH_CONS)
    ib, tb = sc.cast_many(o, d, O.ACCEL_BVH)
    assert (il >= 0).sum() > 100_000                                      # most rays hit something
    assert np.array_equal(ic, il) and np.array_equal(bits(tc), bits(tl))
    lost = int(((ib != il) | (bits(tb) != bits(tl))).sum())
    print(f"\n{len(o)} rays: conservative BVH == List on all; the reference's BVH differs from its List on {lost}")
    assert lost > 0                                                       # the reason this mode exists
    # ... and every one of those rays is the reference's own List / BVH disagreement, by its own slab test (aabb.cc:28-62) on the GEOMETRIC
    # box of the sphere List found (Primitive::BoundingBox, primitive_sphere.cc:69-73): the ray misses that box, or enters it only behind
    # the distance the sphere test accepted -- or the two accelerations report the same distance for two different spheres (an exact tie)
    L = O.load()
    import ctypes as C
    causes = {"tie": 0, "misses the box": 0, "hit in front of the box": 0}
    for i in np.nonzero((ib != il) | (bits(tb) != bits(tl)))[0]:
        assert il[i] >= 0, "the reference BVH found a hit its List does not have"
        if ib[i] >= 0 and bits(tb[i]) == bits(tl[i]):
            causes["tie"] += 1
            continue
        c, r = kw["params"][il[i], :3], kw["params"][il[i], 3]
        mn, mx = (c - r).astype(np.float32), (c + r).astype(np.float32)
        tin, tout = C.c_float(), C.c_float()
        passes = lambda tmax: bool(L.oracle_aabb_intersect(O.f3(mn), O.f3(mx), O.f3(o[i]), O.f3(d[i]), np.float32(tmax), C.byref(tin), C.byref(tout)))
        if not passes(3.4028234663852886e38):
            causes["misses the box"] += 1
        else:
            assert not passes(tl[i]), (i, il[i], ib[i], tl[i], tb[i])         # nothing else may explain a difference
            causes["hit in front of the box"] += 1
    print("   causes:", causes)
    # the blocked scan of oracle_cast_many is the plain loop of Scene::Cast
    for i in (0, 1, 70_000, len(o) - 1):
        idx, t, _, _ = sc.set_accel(O.ACCEL_LIST).cast(o[i], d[i])
        assert idx == il[i] and (idx < 0 or bits(t) == bits(tl[i]))


def test_conservative_bvh_equals_the_list_scan_on_fuzz_scenes():
    n_bvh_differs = 0
    for seed in list(range(900_000, 900_036)):
        scaled, extreme = seed % 3 == 1, seed % 3 == 2
        kw, _ = scene_for_seed(seed, scaled=scaled, extreme=extreme)
        sc = O.Scene.create(**kw, accel=O.ACCEL_BVH_CONS)
        w, h, spp = 24, 18, 6
        out = {}
        for accel in (O.ACCEL_LIST, O.ACCEL_BVH_CONS, O.ACCEL_BVH):
            sc.set_accel(accel)
            img, cnt = sc.render_xorshift(w, h, seed, 0, spp, math=O.MATH_GLIBC, threads=8)
            out[accel] = (img, cnt.casts, sc.path_signatures(w, h, seed, 0, spp, (0, h), math=O.MATH_GLIBC, threads=8))
        img_l, casts_l, sig_l = out[O.ACCEL_LIST]
        img_c, casts_c, sig_c = out[O.ACCEL_BVH_CONS]
        assert casts_c == casts_l and np.array_equal(sig_c, sig_l) and np.array_equal(bits(img_c), bits(img_l)), seed
        n_bvh_differs += int((out[O.ACCEL_BVH][2] != sig_l).sum())
    print(f"\nfuzz scenes: the reference's BVH differs from its List on {n_bvh_differs} paths (ties and grazing hits)")


def test_switching_accelerations_needs_the_right_scene():
    sc = O.Scene.cornell(O.ACCEL_LIST)
    with pytest.raises(ValueError):
        sc.set_accel(O.ACCEL_BVH)
    sc = O.Scene.cornell(O.ACCEL_BVH)
    with pytest.raises(ValueError):
        sc.set_accel(O.ACCEL_BVH_CONS)
    sc.set_accel(O.ACCEL_LIST).set_accel(O.ACCEL_BVH)
