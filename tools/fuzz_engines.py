"""Engine fuzzer: random small scenes (triangles, exact and perturbed parallelograms, fans, coplanar clutter, spheres,
disks, cylinders, all materials, thin and pinhole lenses) rendered by every engine that accepts them; all images and
ray counts must be bit-identical (the engines differ only in how they FIND the closest hit).  Optionally checks a
subset against the CPU oracle.     python tools/fuzz_engines.py [n_scenes] [first_seed] [--oracle]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import os; os.environ.setdefault("AMBER_AMD_LIB", "libamber_hip_lab.so")   # known-answer entry points / lab schedulers: the lab build (include/amber_hip_lab.h)
import amber_amd as A

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n_scenes = int(args[0]) if args else 200
first_seed = int(args[1]) if len(args) > 1 else 0
use_oracle = "--oracle" in sys.argv
scaled = "--scaled" in sys.argv      # similarity transform: scale 10^U(-2,3), offset up to 100 scales (coordinate precision stress)
use_stripes = "--stripes" in sys.argv # also render the image as three interleaved stripe sets (one process per GPU does this)
extreme = "--extreme" in sys.argv    # disk normals scaled by 10^U(-2,2): direction lengths from 0.01 to 100
use_lt = "--lt" in sys.argv           # also compare light tracing (splat records) between the work-queue engines
use_ref = "--reference-bvh" in sys.argv   # also render through AMBER_ENGINE_REFERENCE_BVH and compare with oracle(ACCEL_BVH): image bits, ray count (every scene)
if use_oracle or use_ref:
    sys.path.insert(0, os.path.join(R, "tests"))
    import oracle_binding as O
n_ref_differs = 0


sys.path.insert(0, os.path.join(R, "tests"))
from fuzz_scenes import scene_for_seed


t0 = time.time()
W, H, spp = (128, 96, 12) if "--heavy" in sys.argv else (48, 40, 6)      # --heavy: 13x more paths per scene (rare edge-on rays)
n_pairs_total = 0
for seed in range(first_seed, first_seed + n_scenes):
    t_scene = time.time()
    big = seed % 4 == 3
    sc, rng = scene_for_seed(seed, scaled=scaled, extreme=extreme)
    n_obj = len(sc["objects"]) + max(1, sc["n_blades"])
    hs = A.HostScene.create(**sc)
    # engine BVH twice: its default scheduler (pt_megakernel<ENGINE_BVH> on these shallow trees) and pt_bvh_megakernel (PT_FLAG_BVH_ITEMS)
    engines = [(A.ENGINE_LIST, 0), (A.ENGINE_BVH, 0), (A.ENGINE_BVH, A.api.PT_FLAG_BVH_ITEMS), (A.ENGINE_WAVEFRONT, 0)] + ([(A.ENGINE_TWO_PHASE, 0)] if n_obj <= 128 else [])
    ref = None
    for e, fl in engines:
        pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=seed, engine=e, flags=fl)
        pt.render_pass(0, spp)
        img, rays = pt.download(); pt.close()
        if ref is None:
            ref = (img.view(np.uint32).copy(), rays)
        elif rays != ref[1] or not np.array_equal(img.view(np.uint32), ref[0]):
            bad = np.argwhere(img.view(np.uint32) != ref[0])
            print("MISMATCH seed %d engine %d vs LIST: rays %d vs %d, %d differing values, first at %s" % (seed, e, rays, ref[1], len(bad), bad[:3].tolist()), flush=True)
            sys.exit(1)
    if use_stripes:
        from amber_amd.distributed import stripe_partition
        parts = stripe_partition(H, 3)
        full = ref[0].reshape(H, W, 3)
        tot = 0
        for part in parts:
            pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=seed, rows=part["rows"], stripe=part["stripe"])
            pt.render_pass(0, spp)                                  # (splitting a pass changes the summation order by design)
            img, rays = pt.download(); pt.close(); tot += rays
            if not np.array_equal(img.view(np.uint32), full[part["index"]]):
                print("STRIPE MISMATCH seed %d rows %s" % (seed, part["rows"]), flush=True)
                sys.exit(1)
        if tot != ref[1]:
            print("STRIPE RAY COUNT MISMATCH seed %d: %d vs %d" % (seed, tot, ref[1]), flush=True)
            sys.exit(1)
    if use_lt:
        lref = None
        lt_depth = int(rng.integers(0, 2)) * 5
        for e, fl in [x for x in engines if x[0] != A.ENGINE_WAVEFRONT]:
            pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=seed, engine=e, flags=fl, max_depth=lt_depth)
            rec, lrays = pt.lt_trace(0, 3, capacity=1 << 14); pt.close()
            if lref is None:
                lref = (rec.tobytes(), lrays)
            elif lrays != lref[1] or rec.tobytes() != lref[0]:
                print("LT MISMATCH seed %d engine %d vs LIST: rays %d vs %d, %d records" % (seed, e, lrays, lref[1], len(rec)), flush=True)
                sys.exit(1)
    if use_oracle and seed % 8 == 0 and not big:
        osc = O.Scene.create(**sc)
        oimg, cnt = osc.render_xorshift(W, H, seed, 0, spp)
        if cnt.casts != ref[1] or not np.array_equal(oimg.view(np.uint32), ref[0]):
            print("ORACLE MISMATCH seed %d: rays %d vs %d" % (seed, cnt.casts, ref[1]), flush=True)
            sys.exit(1)
    if use_ref:
        pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=seed, engine=A.ENGINE_REFERENCE_BVH)
        pt.render_pass(0, spp)
        rimg, rrays = pt.download(); pt.close()
        obvh = O.Scene.create(**sc, accel=O.ACCEL_BVH)
        oimg, cnt = obvh.render_xorshift(W, H, seed, 0, spp)
        if cnt.casts != rrays or not np.array_equal(oimg.view(np.uint32), rimg.view(np.uint32)):
            print("REFERENCE-BVH MISMATCH seed %d: rays %d vs oracle(BVH) %d, %d differing values" % (seed, rrays, cnt.casts, int((oimg.view(np.uint32) != rimg.view(np.uint32)).sum())), flush=True)
            sys.exit(1)
        n_ref_differs += int(not np.array_equal(rimg.view(np.uint32), ref[0]))
        if use_lt:
            pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=seed, engine=A.ENGINE_REFERENCE_BVH, max_depth=lt_depth)
            rec, lrays = pt.lt_trace(0, 3, capacity=1 << 14); pt.close()
            _, lcnt, orec = obvh.render_lt(W, H, seed, 0, 3, max_depth=lt_depth)
            got = np.stack([rec["path"], rec["sample"], rec["bounce"], rec["pixel"], *[rec["rgb"][:, c].view(np.uint32) for c in range(3)]], 1) if len(rec) else np.zeros((0, 7), np.uint32)
            if lrays != lcnt.casts or not np.array_equal(got, orec):
                print("REFERENCE-BVH LT MISMATCH seed %d: rays %d vs %d, %d vs %d records" % (seed, lrays, lcnt.casts, len(rec), len(orec)), flush=True)
                sys.exit(1)
    if time.time() - t_scene > 5.0:
        print("seed %d slow: %.1f s (%d objects)" % (seed, time.time() - t_scene, n_obj), flush=True)
    if (seed - first_seed) % 25 == 24:
        print("seed %d ok (%d objects, %.2f rays/path) %.0f s" % (seed, n_obj, ref[1] / (W * H * spp), time.time() - t0), flush=True)
print("fuzz ok: %d scenes, engines agree bit for bit%s" % (n_scenes, " (and with the oracle on every 8th small scene)" if use_oracle else ""))
if use_ref: print("engine REFERENCE_BVH == oracle(reference BVH) on all %d scenes; the List engines differ from it on %d of them" % (n_scenes, n_ref_differs))
