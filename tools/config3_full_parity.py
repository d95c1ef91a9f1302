"""BASELINE config 3 at FULL size (1M spheres, 1920x1080 @ 256 spp, 1.37e9 rays): the three steps of tests/test_config3_parity_gpu.py on
the whole frame.
  (i)   GPU == oracle(List semantics through the conservative BVH, live libm): image bits and ray count;
  (ii)  the pixels on which the GPU differs from oracle(REFERENCE BVH) == the pixels on which the oracle's two accelerations differ;
  (iii) every such pixel attributed to a path whose first differing cast is either a hit the reference's BVH loses because the ray misses
        the sphere's geometric box (its List, like the engine, keeps it) or an exact distance tie;
  (iv)  (round 5) the same frame through AMBER_ENGINE_REFERENCE_BVH == oracle(REFERENCE BVH): image bits and ray count, no pixel left over.
python tools/config3_full_parity.py [spp] [threads] [config3 | terrain | room_mesh]     (config 3: about two minutes of oracle time on 16 host threads;
terrain / room_mesh, round 5: the imported-mesh workloads of bench.py at their full frames, read back through cli::ImportScene)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import amber_amd as A
import oracle_binding as O
from amber_amd import scenes, workloads
# usage: config3_full_parity.py [spp] [threads] [workload]     workload: config3 (default) | terrain | room_mesh   (round 5: the imported-mesh workloads at full size)
args = [a for a in sys.argv[1:]]
workload = next((a for a in args if not a.isdigit()), "config3")
nums = [int(a) for a in args if a.isdigit()]
W, H, seed = (1024, 1024, 7) if workload == "room_mesh" else (1920, 1080, 3 if workload == "terrain" else 1)
spp = nums[0] if nums else {"config3": 256, "terrain": 64, "room_mesh": 256}[workload]
threads = nums[1] if len(nums) > 1 else 16
if workload == "config3":
    kw = scenes.random_spheres(1_000_000, 7); blades_last = 0
    hs = A.HostScene.create_arrays(**kw)
else:
    import tempfile
    wl = workloads.terrain_mesh(16, 56) if workload == "terrain" else workloads.room_mesh(3)
    kw = wl.arrays(); blades_last = O.BLADES_LAST
    hs = A.HostScene.import_file(wl.write(tempfile.mkdtemp()))
pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=seed)
t = time.time(); pt.render_pass(0, spp); img, rays = pt.download(); tg = time.time() - t
pt.close()
print("%s, %dx%d @ %d spp, seed %d" % (workload, W, H, spp, seed))
print("GPU (engine BVH, List semantics): %d rays" % rays, flush=True)
t = time.time(); osc = O.Scene.create_arrays(**kw, accel=O.ACCEL_BVH_CONS | blades_last); tb = time.time() - t
bits = lambda a: a.view(np.uint32)
res = {}
for name, accel in (("List via the conservative BVH", O.ACCEL_BVH_CONS), ("reference BVH", O.ACCEL_BVH)):
    t = time.time(); ref, cnt = osc.set_accel(accel).render_xorshift(W, H, seed, 0, spp, math=O.MATH_LIBM, threads=threads); dt = time.time() - t
    res[accel] = (ref, cnt.casts)
    diff = (bits(img) != bits(ref)).any(axis=2)
    err = np.sqrt(((img.astype(np.float64) - ref) ** 2).sum(axis=2)); mag = np.sqrt((ref.astype(np.float64) ** 2).sum(axis=2))
    rel = np.where(mag > 0, err / np.where(mag > 0, mag, 1), 0)
    print("oracle(XorShift, %s, live libm): tree %.1f s, %d rays in %.1f s (%.1f Mrays/s)" % (name, tb, cnt.casts, dt, cnt.casts / dt / 1e6))
    print("  GPU vs this: pixels differing %d of %d (%.2e); over the 1e-4 relative-L2 tolerance %d; ray count delta %+d; lit pixels %d"
          % (int(diff.sum()), W * H, diff.mean(), int((rel > 1e-4).sum()), int(rays) - int(cnt.casts), int((ref > 0).any(axis=2).sum())), flush=True)
cons, refb = res[O.ACCEL_BVH_CONS][0], res[O.ACCEL_BVH][0]
gpu_vs_ref = (bits(img) != bits(refb)).any(axis=2); cons_vs_ref = (bits(cons) != bits(refb)).any(axis=2)
print("(ii) set of pixels GPU != reference BVH equals set of pixels List != reference BVH:", bool(np.array_equal(gpu_vs_ref, cons_vs_ref)))
from bvh_parity import classify_pixels
pixels = list(zip(*np.nonzero(gpu_vs_ref)))
t = time.time(); causes = classify_pixels(osc, W, H, seed, pixels, spp); dt = time.time() - t
tally = {}
unattributed = 0
for px, found in causes.items():
    if not found: unattributed += 1
    for c in found: tally[c["cause"]] = tally.get(c["cause"], 0) + 1
print("(iii) %d differing pixels classified in %.1f s: %d without a differing path; first differing casts by cause:" % (len(pixels), dt, unattributed))
for k, v in sorted(tally.items()): print("      %5d  %s" % (v, k))
for px, found in list(causes.items())[:12]:
    for c in found:
        print("      pixel (y %4d, x %4d) sample %3d cast %d: List object %7d t %.9g | reference BVH object %7d t %.9g | %s"
              % (px[0], px[1], c["sample"], c["cast"], c["object_list"], c["t_list"], c["object_b"], c["t_b"], c["cause"]))
# (iv) round 5: the same frame through AMBER_ENGINE_REFERENCE_BVH -- the reference's own tree and traversal order -- against oracle(reference BVH)
t = time.time(); pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=seed, engine=A.ENGINE_REFERENCE_BVH); tc = time.time() - t
pt.render_pass(0, spp); rimg, rrays = pt.download(); launches, ms = pt.kernel_time(); pt.close()
rdiff = int((bits(rimg) != bits(refb)).any(axis=2).sum())
print("(iv) engine REFERENCE_BVH: create %.1f s (the reference's build on the host), %d rays in %.1f ms (%d launches, %.2f Grays/s); pixels differing from oracle(reference BVH): %d of %d; ray count delta %+d"
      % (tc, rrays, ms, launches, rrays / ms / 1e6, rdiff, W * H, int(rrays) - int(res[O.ACCEL_BVH][1])), flush=True)
ok = rdiff == 0 and rrays == res[O.ACCEL_BVH][1] and int((bits(img) != bits(cons)).any(axis=2).sum()) == 0 and rays == res[O.ACCEL_BVH_CONS][1] and np.array_equal(gpu_vs_ref, cons_vs_ref) and unattributed == 0 and "unexplained" not in tally
print("RESULT:", "every difference of the List engines from the reference's BVH is the reference's own List/BVH disagreement; engine REFERENCE_BVH equals the reference's BVH on every pixel" if ok else "NOT fully attributed")
