"""Shared by the GPU parity tests of engine BVH (config 3, the imported-mesh workloads, Cornell through ENGINE_BVH).

Every engine that AUTO can choose implements the reference's List acceleration (/root/reference/include/amber/raytracer/acceleration_list.h:51-68)
(AMBER_ENGINE_REFERENCE_BVH implements the reference's BVH: check_reference_engine below).  The
reference's BVH (acceleration_bvh.h:340-403) is a different function of the ray wherever a primitive test accepts a hit outside the
primitive's geometric box (binary32 sphere discriminants, barycentrics of needle triangles) or two objects tie exactly.  So a band is
checked in three steps:
  (i)   image bits and ray count: GPU == oracle(ACCEL_BVH_CONS) -- List's answer at BVH speed (tests/test_oracle_conservative_bvh.py);
  (ii)  the pixels on which the GPU differs from oracle(reference BVH) are EXACTLY those on which the oracle's two accelerations differ,
        and their number and the ray-count difference stay under a hard bound (a regression against the reference's real behaviour
        cannot hide behind the engine-shaped oracle);
  (iii) every such pixel holds a path on whose first differing cast the plain List scan over all objects returns what the GPU's semantics
        returned, while the reference BVH lost that hit (the ray misses the object's Primitive::BoundingBox through the reference's own
        slab test, aabb.cc:28-62, or enters it only behind the accepted distance) or resolved an exact distance tie the other way.
"""
import numpy as np

import oracle_binding as O


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def classify_pixels(osc, W, H, seed, pixels, spp, max_depth=0):
    """{(y, x): [dict, ...]}: the paths of every pixel on which oracle(List via conservative BVH) and oracle(reference BVH) part ways."""
    out = {}
    for y, x in pixels:
        found = []
        for k in range(spp):
            c = osc.classify_path(W, H, seed, int(x), int(y), k, O.ACCEL_BVH_CONS, O.ACCEL_BVH, math=O.MATH_LIBM, max_depth=max_depth)
            if c is None:
                continue
            assert c["object_a"] == c["object_list"] and c["t_bits"][0] == c["t_bits"][2], (y, x, k, c)   # conservative BVH == the plain scan
            if c["exact_tie"]:
                c["cause"] = "exact distance tie between two objects"
            elif c["object_list"] >= 0 and not c["list_object_box_hit"]:
                c["cause"] = "lost hit: the reference's primitive test accepts a ray that misses the object's geometric box"
            elif c["object_list"] >= 0 and not c["list_hit_inside_box"]:
                c["cause"] = "lost hit: the accepted distance lies in front of the point where the ray enters the object's geometric box"
            else:
                c["cause"] = "unexplained"
            c["sample"] = k
            found.append(c)
        out[(int(y), int(x))] = found
    return out


def check_band(amber, hs, osc, W, H, seed, spp, rows, max_ref_pixels, max_ref_ray_delta, engine=0, flags=0, max_depth=0, threads=16, label=""):
    """Steps (i)-(iii) on rows [rows[0], rows[1]) at `spp` samples.  osc: an oracle scene created with ACCEL_BVH_CONS.  Returns a dict of counts."""
    pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=seed, rows=rows, engine=engine, flags=flags, max_depth=max_depth)
    pt.render_pass(0, spp)
    img, rays = pt.download()
    pt.close()
    full = np.zeros((H, W, 3), np.float32)
    _, cnt = osc.set_accel(O.ACCEL_BVH_CONS).render_xorshift(W, H, seed, 0, spp, math=O.MATH_LIBM, max_depth=max_depth, threads=threads, rows=rows, out=full)
    cons = full[rows[0]:rows[1]]
    assert rays == cnt.casts, (label, rays, cnt.casts)                                                     # (i)
    assert int((bits(img) != bits(cons)).any(axis=2).sum()) == 0, label
    full_b = np.zeros((H, W, 3), np.float32)
    _, cnt_b = osc.set_accel(O.ACCEL_BVH).render_xorshift(W, H, seed, 0, spp, math=O.MATH_LIBM, max_depth=max_depth, threads=threads, rows=rows, out=full_b)
    ref = full_b[rows[0]:rows[1]]
    gpu_vs_ref = (bits(img) != bits(ref)).any(axis=2)                                                      # (ii)
    cons_vs_ref = (bits(cons) != bits(ref)).any(axis=2)
    assert np.array_equal(gpu_vs_ref, cons_vs_ref), label
    n_diff, ray_delta = int(gpu_vs_ref.sum()), int(rays) - int(cnt_b.casts)
    assert n_diff <= max_ref_pixels and abs(ray_delta) <= max_ref_ray_delta, (label, n_diff, ray_delta)
    pixels = [(y + rows[0], x) for y, x in zip(*np.nonzero(gpu_vs_ref))]                                   # (iii)
    causes = classify_pixels(osc.set_accel(O.ACCEL_BVH_CONS), W, H, seed, pixels, spp, max_depth=max_depth)
    for px, found in causes.items():
        assert found, f"{label}: pixel {px} differs but no path of it does"
        for c in found:
            assert c["cause"] != "unexplained", (label, px, c)
    flat = [c for f in causes.values() for c in f]
    out = dict(reference_image=ref.copy(), reference_casts=int(cnt_b.casts), rays=int(rays), pixels=img.shape[0] * W, differ_from_reference_bvh=n_diff, ray_delta_to_reference_bvh=ray_delta, paths=len(flat),
               lost_hits=sum(c["cause"].startswith("lost hit") for c in flat), ties=sum(c["cause"].startswith("exact") for c in flat),
               lit=float((img.sum(axis=2) > 0).mean()))
    print(f"\n{label}: rows {rows} @ {spp} spp: {rays} rays, GPU == oracle(List) on all {out['pixels']} pixels; against the reference's BVH {n_diff} pixels differ "
          f"({ray_delta:+d} rays), every one attributed ({out['paths']} paths: {out['lost_hits']} lost hits, {out['ties']} ties)")
    return out


def check_reference_engine(amber, hs, W, H, seed, spp, rows, reference_image, reference_casts, max_depth=0, label=""):
    """AMBER_ENGINE_REFERENCE_BVH (the reference's own tree, its own traversal order) on the same band: image bits and ray count
    == oracle(ACCEL_BVH), i.e. what the reference's code computes through its BVH from the same random numbers -- including the pixels on which the other engines differ."""
    pt = amber.PathTracer(hs, amber.Sensor.default(W, H), seed=seed, rows=rows, engine=amber.ENGINE_REFERENCE_BVH, max_depth=max_depth)
    pt.render_pass(0, spp)
    img, rays = pt.download()
    ms = pt.kernel_time()[1]
    pt.close()
    differing = int((bits(img) != bits(reference_image)).any(axis=2).sum())
    print(f"\n{label}: engine REFERENCE_BVH, rows {rows} @ {spp} spp: {rays} rays in {ms:.1f} ms, {differing} pixels differ from oracle(reference BVH)")
    assert rays == reference_casts, (label, rays, reference_casts)
    assert differing == 0, (label, differing)
