import sys, os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,R+'/tests')
import numpy as np, amber_amd as A, oracle_binding as O, ctypes as C
sc = A.HostScene.cornell_box(); osc = O.Scene.cornell(O.ACCEL_LIST)
W=64; sn=A.Sensor.default(W,W); pt=A.PathTracer(sc,sn,seed=12345)
px = np.arange(0, W*W, 1, dtype=np.uint32); sm = (px % 5).astype(np.uint32)
eye = pt.kat_eye(px, sm)
rec, casts = pt.kat_trace(px, sm, 16)
bad_eye=0; shown=0; nbad=0
for i,(p,s_) in enumerate(zip(px,sm)):
    n, orec, oeye = osc.trace(W,W,12345,int(p%W),int(p//W),int(s_))
    if not np.array_equal(oeye.view(np.uint32), eye[i].view(np.uint32)):
        bad_eye+=1
        if bad_eye<4: print('EYE diff px',p,'gpu',eye[i],'ora',oeye, eye[i].view(np.uint32)-oeye.view(np.uint32))
    mism=None
    for b in range(min(max(n,casts[i]),16)):
        r=rec[i,b]
        g=(np.int32(r[0]), r[1:11].view(np.float32))
        o=(orec[b].object, np.array([orec[b].t]+list(orec[b].pos)+list(orec[b].weight)+list(orec[b].measurement),np.float32))
        if g[0]!=o[0] or (o[0]>=0 and not np.array_equal(g[1].view(np.uint32), o[1].view(np.uint32))):
            mism=(b,g,o); break
    if mism or n!=casts[i]:
        nbad+=1
        if shown<6:
            shown+=1; print('TRACE diff px',p,'sample',s_,'casts gpu',casts[i],'ora',n)
            if mism:
                b,g,o=mism; print('  bounce',b,'gpu obj',g[0],'ora obj',o[0]); print('   gpu',g[1]); print('   ora',o[1]); print('   ulp',g[1].view(np.int32)-o[1].view(np.int32))
                if b>0:
                    r=rec[i,b-1]; print('   prev obj', np.int32(r[0]))
print('eye mismatches',bad_eye,'trace mismatches',nbad,'of',len(px))
# pow check
rng=np.random.default_rng(1); xs=rng.random(20000).astype(np.float32); ys=np.full(20000,np.float32(1/257),np.float32)
gp=A.kat_math(1,np.stack([xs,ys],1)); L=O.load(); bp=0
for i in range(len(xs)):
    if np.float32(L.oracle_pow(xs[i],ys[i],1)).view(np.uint32)!=gp[i].view(np.uint32): bp+=1
print('pow mismatches',bp)
