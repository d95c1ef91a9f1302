"""Engine BVH vs engine LIST (exhaustive exact scan) on the 1M-sphere scene: per-ray equality on random and path-like rays."""
import sys, os, time; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R)
import numpy as np, amber_amd as A
import os; os.environ.setdefault("AMBER_AMD_LIB", "libamber_hip_lab.so")   # known-answer entry points / lab schedulers: the lab build (include/amber_hip_lab.h)
from amber_amd import scenes
n_obj=int(sys.argv[1]) if len(sys.argv)>1 else 1_000_000
n=int(sys.argv[2]) if len(sys.argv)>2 else 20000
hs=A.HostScene.create_arrays(**scenes.random_spheres(n_obj,7)); sn=A.Sensor.default(64,64)
rng=np.random.default_rng(1)
org=rng.uniform(-1,1,(n,3)).astype(np.float32); d=rng.normal(size=(n,3)); d=(d/np.linalg.norm(d,axis=1,keepdims=True)).astype(np.float32)
b=A.PathTracer(hs,sn,engine=A.ENGINE_BVH); l=A.PathTracer(hs,sn,engine=A.ENGINE_LIST)
t=time.time(); rb=b.kat_cast(org,d); tb=time.time()-t
# second generation: rays leaving the hit points (origins ON sphere surfaces, like real bounce rays)
hit=rb[0]>=0; org2=rb[2][hit][:n//2]; d2=rng.normal(size=(len(org2),3)); d2=(d2/np.linalg.norm(d2,axis=1,keepdims=True)).astype(np.float32)
org=np.concatenate([org,org2]); d=np.concatenate([d,d2])
rb=b.kat_cast(org,d)
t=time.time(); rl=l.kat_cast(org,d); tl=time.time()-t
same=np.array_equal(rb[0],rl[0]); h=rl[0]>=0
print("rays",len(org),"hit frac %.3f"%h.mean(),"bvh %.2fs list %.2fs"%(tb,tl),"objects equal:",same,"t equal:",np.array_equal(rb[1][h].view(np.uint32),rl[1][h].view(np.uint32)))
if not same:
    bad=np.nonzero(rb[0]!=rl[0])[0]; print("mismatches",len(bad)); 
    for i in bad[:5]: print(i,org[i],d[i],"bvh",rb[0][i],rb[1][i],"list",rl[0][i],rl[1][i])
