"""Engine BVH on a triangle mesh (scenes.mesh_room): kernel time against samples per launch, looking for outlier paths."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import amber_amd as A
from amber_amd import scenes
sub = int(sys.argv[1]) if len(sys.argv) > 1 else 8
t = time.time(); kw = scenes.mesh_room(sub); t_gen = time.time() - t
t = time.time(); hs = A.HostScene.create_arrays(**kw); t_scene = time.time() - t
W, H = 1280, 720
t = time.time(); pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=1); t_create = time.time() - t
print("mesh_room(%d): %d triangles; generate %.2f s, host scene %.2f s, flatten + BVH + upload %.2f s" % (sub, len(kw["kinds"]), t_gen, t_scene, t_create), flush=True)
pt.render_pass(0, 4); pt.sync()
for spp in (4, 8, 16, 32, 64, 128):
    pt.clear(); pt.render_pass(0, spp); pt.sync(); n, ms = pt.kernel_time(); r = pt.ray_count()
    print("%dx%d spp %4d: %8.1f ms  %7.1f Mrays/s  %.2f rays/path" % (W, H, spp, ms, r / ms / 1e3, r / (W * H * spp)), flush=True)
