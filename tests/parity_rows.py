"""Row-level parity of the HIP engine against the oracle calling the host's LIVE libm -- the reference's arithmetic.

TEST INFRASTRUCTURE (uses oracle/): imported by tests/test_parity_against_live_libm.py and by bench.py's cpu_baseline leg,
where the oracle acts as the checker of the rows it renders there.  Workload = BASELINE.json configs[1] (Cornell box
1024x1024 @ 1024 spp) restricted to a few full-width rows at ALL samples, so every sample index of the headline run is
checked on those pixels.

For every band of rows:
  * image: the GPU's row sums against oracle(XorShift sampler, reference BVH, MATH_LIBM) -- bits, and the north star's
    tolerance (per-pixel relative L2 <= 1e-4; a pixel whose oracle value is 0 must be 0);
  * rays: Scene::Cast counts;
  * paths: per-path signatures (hash of the hit-object sequence | hash of the hit distances) from the PRODUCT kernel
    (amber_hip_pt_signatures; cross-checked against amber_hip_kat_signatures) and oracle_path_signatures: `diverged_paths` = paths whose object sequences AND hit
    distances differ, `inexact_paths` = same objects but some hit distance differs in a bit, `tie_paths` = every hit
    distance identical but a different object index somewhere: an exact distance tie between two objects (a ray through
    the shared edge of two triangles), which the reference's BVH and its List acceleration resolve differently
    (acceleration_bvh.h:340-403 keeps the first-visited leaf's hit, acceleration_list.h:51-68 the lower index; the
    engine implements List -- SURVEY.md Appendix C lists exact ties as a permitted difference);
  * `engine_reference_bvh`: the same rows through AMBER_ENGINE_REFERENCE_BVH, which walks the reference's own tree in the reference's order:
    differing pixels, ray-count difference and differing path signatures against the same oracle -- all zero, ties included.
"""
from __future__ import annotations

import numpy as np

import oracle_binding as O

DEFAULT_BANDS = ((250, 254), (508, 512), (700, 704), (900, 904))      # wall / back wall + light row / spheres / floor + water


def bands_for(height: int):
    """DEFAULT_BANDS scaled to another frame height (4 rows each, clipped)."""
    if height == 1024:
        return DEFAULT_BANDS
    out = []
    for y0, _ in DEFAULT_BANDS:
        a = min(max(0, y0 * height // 1024), max(0, height - 1))
        out.append((a, min(height, a + 4)))
    return tuple(out)


def compare_rows(amber, width: int = 1024, spp: int = 1024, seed: int = 12345, bands=None, threads: int = 8,
                 math: int = O.MATH_LIBM, accel: int = O.ACCEL_BVH, signatures: bool = True, tol: float = 1e-4) -> dict:
    bands = bands_for(width) if bands is None else bands
    hs = amber.HostScene.cornell_box()
    sensor = amber.Sensor.default(width, width)
    osc = O.Scene.cornell(accel)
    out = dict(rows=0, paths=0, rays_gpu=0, rays_oracle=0, pixels=0, pixels_differing=0, pixels_over_tol=0, max_rel_l2=0.0,
               diverged_paths=0 if signatures else None, inexact_paths=0 if signatures else None, tie_paths=0 if signatures else None,
               against=f"oracle(XorShift, {'BVH' if accel == O.ACCEL_BVH else 'List'}, "
                       f"{ {O.MATH_LIBM: 'live libm', O.MATH_GLIBC: 'glibc restatement', O.MATH_PORTABLE: 'portable'}[math] })",
               bands=[list(b) for b in bands], tolerance=tol)
    # the same rows through AMBER_ENGINE_REFERENCE_BVH (the reference's own tree and traversal order): against the same oracle nothing may be left over
    ref_engine = dict(pixels_differing=0, cast_delta=0, paths_differing=0 if signatures else None) if accel == O.ACCEL_BVH else None
    for y0, y1 in bands:
        pt = amber.PathTracer(hs, sensor, seed=seed, rows=(y0, y1))
        pt.render_pass(0, spp)
        g, rays = pt.download()
        full = np.zeros((width, width, 3), np.float32)
        _, cnt = osc.render_xorshift(width, width, seed, 0, spp, math=math, threads=threads, rows=(y0, y1), out=full)
        o = full[y0:y1]
        out["rows"] += y1 - y0
        out["paths"] += (y1 - y0) * width * spp
        out["pixels"] += (y1 - y0) * width
        out["rays_gpu"] += int(rays)
        out["rays_oracle"] += int(cnt.casts)
        out["pixels_differing"] += int((g.view(np.uint32) != o.view(np.uint32)).any(axis=2).sum())
        err = np.sqrt(((g.astype(np.float64) - o) ** 2).sum(axis=2))
        ref = np.sqrt((o.astype(np.float64) ** 2).sum(axis=2))
        rel = np.where(ref > 0, err / np.where(ref > 0, ref, 1), np.where(err > 0, np.inf, 0.0))
        out["pixels_over_tol"] += int((rel > tol).sum())
        out["max_rel_l2"] = float(max(out["max_rel_l2"], rel.max()))
        if signatures:
            # from the kernel that rendered the rows above (pt_megakernel's signature instantiation), not only from the per-thread
            # known-answer kernel: both must tell the same story
            sg = pt.render_signatures(0, spp)
            out["signature_kernel_mismatches"] = out.get("signature_kernel_mismatches", 0) + int((sg != pt.kat_signatures(0, spp)).sum())
            so = osc.path_signatures(width, width, seed, 0, spp, (y0, y1), math=math, threads=threads)
            lo_g, lo_o = sg & np.uint64(0xffffffff), so & np.uint64(0xffffffff)
            hi_g, hi_o = sg >> np.uint64(32), so >> np.uint64(32)
            out["diverged_paths"] += int(((lo_g != lo_o) & (hi_g != hi_o)).sum())
            out["inexact_paths"] += int(((lo_g == lo_o) & (hi_g != hi_o)).sum())
            out["tie_paths"] += int(((lo_g != lo_o) & (hi_g == hi_o)).sum())
        pt.close()
        if ref_engine is not None:
            pr = amber.PathTracer(hs, sensor, seed=seed, rows=(y0, y1), engine=amber.ENGINE_REFERENCE_BVH)
            pr.render_pass(0, spp)
            gr, rr = pr.download()
            ref_engine["pixels_differing"] += int((gr.view(np.uint32) != o.view(np.uint32)).any(axis=2).sum())
            ref_engine["cast_delta"] += int(rr) - int(cnt.casts)
            if signatures:
                ref_engine["paths_differing"] += int((pr.render_signatures(0, spp) != so).sum())
            pr.close()
    out["cast_delta"] = out["rays_gpu"] - out["rays_oracle"]
    if ref_engine is not None:
        out["engine_reference_bvh"] = ref_engine
    return out


def compare_image_rows(image: np.ndarray, width: int, spp: int, seed: int, bands=None, threads: int = 8, math: int = O.MATH_LIBM,
                       accel: int = O.ACCEL_BVH, tol: float = 1e-4, launches=None) -> dict:
    """The image leg of compare_rows for an image rendered ELSEWHERE -- bench.py's gathered image of an N > 1 run (row sums over all samples
    of the Cornell box, width x width): the rows of `bands` against the oracle, bits and the north star's tolerance.  `launches` =
    [(first_sample, n)] as the bench issued them (the summation order splits on launch boundaries)."""
    bands = bands_for(width) if bands is None else bands
    osc = O.Scene.cornell(accel)
    out = dict(rows=0, pixels=0, pixels_differing=0, pixels_over_tol=0, max_rel_l2=0.0, rays_oracle=0, bands=[list(b) for b in bands], tolerance=tol,
               against=f"oracle(XorShift, {'BVH' if accel == O.ACCEL_BVH else 'List'}, live libm) on rows of the gathered image")
    for y0, y1 in bands:
        full = np.zeros((width, width, 3), np.float32)
        for first, n in (launches or [(0, spp)]):                      # the oracle adds every chunk sum onto the running pixel value, launch after launch, as the engine does
            _, cnt = osc.render_xorshift(width, width, seed, first, n, math=math, threads=threads, rows=(y0, y1), out=full)
            out["rays_oracle"] += int(cnt.casts)
        g, o = np.ascontiguousarray(image[y0:y1], np.float32), full[y0:y1]
        out["rows"] += y1 - y0
        out["pixels"] += (y1 - y0) * width
        out["pixels_differing"] += int((g.view(np.uint32) != o.view(np.uint32)).any(axis=2).sum())
        err = np.sqrt(((g.astype(np.float64) - o) ** 2).sum(axis=2))
        ref = np.sqrt((o.astype(np.float64) ** 2).sum(axis=2))
        rel = np.where(ref > 0, err / np.where(ref > 0, ref, 1), np.where(err > 0, np.inf, 0.0))
        out["pixels_over_tol"] += int((rel > tol).sum())
        out["max_rel_l2"] = float(max(out["max_rel_l2"], rel.max()))
    return out
