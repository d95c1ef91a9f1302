"""Kernel time of AMBER_ENGINE_REFERENCE_BVH (the reference's own tree, walked in the reference's order) beside the fast engine of the same
scene, at reduced sample counts: config 3's 1M spheres, the 1.04M-triangle terrain, the room mesh, the Cornell box.
python tools/reference_bvh_engine.py [scale] [scene substring] [--reference-only]     (scale divides the sample counts, default 4; the last two for profiling one kernel:
tools/pmc_generic.sh refbvh "pt_megakernel<6" tools/reference_bvh_engine.py 4 "config 3" --reference-only)"""
import os, sys, time, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import amber_amd as A
from amber_amd import scenes, workloads

_args = [a for a in sys.argv[1:] if not a.startswith("--")]
scale = int(_args[0]) if _args else 4
only = _args[1] if len(_args) > 1 else ""
engines = (A.ENGINE_REFERENCE_BVH,) if "--reference-only" in sys.argv else (A.ENGINE_AUTO, A.ENGINE_REFERENCE_BVH)
tmp = tempfile.mkdtemp()
cases = [("config 3: 1M spheres", lambda: A.HostScene.create_arrays(**scenes.random_spheres(1_000_000, 7)), 1920, 1080, 256, 1),
         ("terrain: 1.04M triangles", lambda: A.HostScene.import_file(workloads.terrain_mesh(16, 56).write(tmp)), 1920, 1080, 64, 3),
         ("room mesh: 1 304 triangles", lambda: A.HostScene.import_file(workloads.room_mesh(3).write(tmp)), 1024, 1024, 256, 7),
         ("Cornell box", A.HostScene.cornell_box, 1024, 1024, 1024, 12345)]
print("%-28s %6s %14s %12s %12s %8s %10s" % ("scene", "spp", "rays (ref)", "fast ms", "reference ms", "ratio", "create s"))
for name, make, W, H, spp, seed in cases:
    if only not in name: continue
    hs = make(); n = max(8, spp // scale)
    out = {}
    for engine in engines:
        t = time.time(); pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=seed, engine=engine); tc = time.time() - t
        pt.render_pass(0, 8); pt.sync(); pt.clear()
        l0, m0 = pt.kernel_time()
        pt.render_pass(0, n); pt.sync()
        l1, m1 = pt.kernel_time()
        out[engine] = (m1 - m0, pt.ray_count(), tc); pt.close()
    r = out[A.ENGINE_REFERENCE_BVH]; f = out.get(A.ENGINE_AUTO, (float("nan"), 0, 0))
    print("%-28s %6d %14d %12.2f %12.2f %8.2f %10.2f" % (name, n, r[1], f[0], r[0], r[0] / f[0], r[2]), flush=True)
    hs.close()
