"""BASELINE config 3: 1M random spheres (deep BVH), 1920x1080 @ 256 spp on one GPU (engine BVH). Prints Mrays/s."""
import sys, os, time; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R)
import amber_amd as A
from amber_amd import scenes
spp=int(sys.argv[1]) if len(sys.argv)>1 else 256
engine=int(sys.argv[2]) if len(sys.argv)>2 else 0   # AMBER_ENGINE_*: 0 auto (BVH), 4 wavefront
t=time.time(); hs=A.HostScene.create_arrays(**scenes.random_spheres(1_000_000,7)); t_scene=time.time()-t
sn=A.Sensor.default(1920,1080)
t=time.time(); pt=A.PathTracer(hs,sn,seed=1,engine=engine); t_create=time.time()-t
pt.render_pass(0,8); pt.sync(); pt.clear()
t=time.time(); pt.render_pass(0,spp); pt.sync(); wall=time.time()-t
n,ms=pt.kernel_time(); r=pt.ray_count()
print("config 3: host scene %.2f s, flatten+BVH build+upload %.2f s; 1920x1080@%d spp: kernel %.1f ms (wall %.1f), %d rays, %.1f Mrays/s, %.2f rays/path"%(t_scene,t_create,spp,ms,wall*1e3,r,r/ms/1e3,r/(1920*1080*spp)))
