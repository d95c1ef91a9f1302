"""pt_bvh_megakernel's shading batch (lanes that must have finished their traversal before the wave shades) against the scene:
kernel ms per batch size for the 1M-sphere scene, the 1M-triangle terrain, an 82k-triangle room mesh and the Cornell box (item kernel).
The batch is a kernel argument; AMBER_BVH_SHADE_BATCH (read at create) overrides the handle's choice."""
import os, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import amber_amd as A
from amber_amd import scenes, workloads as WL
batches = [int(x) for x in sys.argv[1:]] or [20, 24, 28, 32, 36, 40, 44, 48, 52, 56]
d = tempfile.mkdtemp()
jobs = [("1M spheres 1920x1080@64", A.HostScene.create_arrays(**scenes.random_spheres(1_000_000, 7)), 1920, 1080, 64, 0),
        ("terrain 1M tris 1920x1080@64", A.HostScene.import_file(WL.terrain_mesh(16, 56).write(d)), 1920, 1080, 64, 0),
        ("room mesh 82k tris 1024^2@128", A.HostScene.import_file(WL.room_mesh(6).write(d)), 1024, 1024, 128, 0),
        ("Cornell, item kernel 1024^2@128", A.HostScene.cornell_box(), 1024, 1024, 128, A.api.PT_FLAG_BVH_ITEMS)]
print("%-34s" % "batch" + "".join("%8d" % b for b in batches) + "    auto", flush=True)
for name, hs, W, H, spp, flags in jobs:
    row = []
    for b in batches + [0]:
        if b: os.environ["AMBER_BVH_SHADE_BATCH"] = str(b)
        else: os.environ.pop("AMBER_BVH_SHADE_BATCH", None)
        pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=1, engine=A.ENGINE_BVH, flags=flags)
        pt.render_pass(0, 8); pt.sync(); pt.clear()
        pt.render_pass(0, spp); pt.sync()
        row.append(pt.kernel_time()[1]); pt.close()
    print("%-34s" % name + "".join("%8.2f" % x for x in row), flush=True)
