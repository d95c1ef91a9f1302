"""Times every engine on BASELINE config 2 (Cornell 1024x1024) at a given spp: kernel ms (hipEvents) and Mrays/s."""
import sys, os, time; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R)
import os; os.environ.setdefault("AMBER_AMD_LIB", "libamber_hip_lab.so")   # known-answer entry points / lab schedulers: the lab build (include/amber_hip_lab.h)
import amber_amd as A
spp=int(sys.argv[1]) if len(sys.argv)>1 else 256
sc=A.HostScene.cornell_box(); sn=A.Sensor.default(1024,1024)
for name,eng in (("two_phase",A.ENGINE_TWO_PHASE),("list",A.ENGINE_LIST),("bvh",A.ENGINE_BVH),("wavefront",A.ENGINE_WAVEFRONT)):
    pt=A.PathTracer(sc,sn,engine=eng); pt.render_pass(0,32); pt.sync(); pt.clear()
    t=time.time(); pt.render_pass(0,spp); pt.sync(); wall=time.time()-t
    n,ms=pt.kernel_time(); r=pt.ray_count(); pt.close()
    print("%-10s %4d spp: kernel %.2f ms (wall %.2f ms)  %.1f Mrays/s"%(name,spp,ms,wall*1e3,r/ms/1e3))
