import sys, time; import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,R+'/tests')
import numpy as np, amber_amd as A, oracle_binding as O
print('devices', A.device_count())
sc = A.HostScene.cornell_box(); osc = O.Scene.cornell(O.ACCEL_LIST)
# math
x = np.linspace(0, 6.2831855, 100001, dtype=np.float32)
g = A.kat_math(0, x)
L = O.load(); import ctypes as C
s,c = C.c_float(), C.c_float(); bad=0
for i in range(0,len(x),7):
    L.oracle_sincos(x[i],1,C.byref(s),C.byref(c))
    if np.float32(s.value).view(np.uint32)!=g[i,0].view(np.uint32) or np.float32(c.value).view(np.uint32)!=g[i,1].view(np.uint32): bad+=1
print('sincos mismatches', bad, 'max err vs numpy', np.abs(g[:,0]-np.sin(x.astype(np.float64))).max())
W=64; spp=16
sn = A.Sensor.default(W,W)
pt = A.PathTracer(sc, sn, seed=12345)
t=time.time(); pt.render_pass(0, spp); img, rays = pt.download(); print('gpu render', time.time()-t, 'rays', rays)
oi, cnt = osc.render_xorshift(W,W,12345,0,spp)
print('oracle casts', cnt.casts, 'equal rays', cnt.casts==rays)
print('bit-exact image', np.array_equal(img.view(np.uint32), oi.view(np.uint32)), 'nonzero', (img>0).sum(), (oi>0).sum(), 'maxdiff', np.abs(img-oi).max())
# traces
px = np.arange(0, W*W, 37, dtype=np.uint32); sm = (px % 5).astype(np.uint32)
rec, casts = pt.kat_trace(px, sm, 16)
nbad=0
for i,(p,s_) in enumerate(zip(px,sm)):
    n, orec, eye = osc.trace(W,W,12345,int(p%W),int(p//W),int(s_))
    if n!=casts[i]: nbad+=1; continue
    for b in range(min(n,16)):
        r=rec[i,b]
        if np.int32(r[0])!=orec[b].object or r[1]!=np.float32(orec[b].t).view(np.uint32) and orec[b].object>=0: nbad+=1; break
print('trace mismatches', nbad, 'of', len(px))
# bench
W=1024; sn=A.Sensor.default(W,W); pt=A.PathTracer(sc,sn)
pt.render_pass(0,4); pt.sync()
for spp in (16,64):
    pt.clear(); t=time.time(); pt.render_pass(0,spp); pt.sync(); dt=time.time()-t
    r=pt.ray_count(); n,ms=pt.kernel_time()
    print('1024^2 @%d spp: %.3fs  rays %d  %.1f Mrays/s  kernel_ms %.2f' % (spp, dt, r, r/dt/1e6, ms))
