"""Predicts strong-scaling efficiency on one GPU: time every band of an N-way split separately (what each rank would run)."""
import sys, os, time; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R)
import numpy as np, amber_amd as A
from amber_amd.distributed import partition_rows, stripe_partition
W=H=1024; spp=int(sys.argv[1]) if len(sys.argv)>1 else 1024
sc=A.HostScene.cornell_box(); sn=A.Sensor.default(W,H)
def run(rows, stripe=None):
    pt=A.PathTracer(sc,sn,rows=rows,stripe=stripe); pt.render_pass(0,8); pt.sync(); pt.clear()
    pt.render_pass(0,spp); pt.sync(); n,ms=pt.kernel_time(); r=pt.ray_count(); pt.close(); return ms,r
full,rf=run((0,H)); print('full %.2f ms rays %d'%(full,rf))
for n in (2,4,8):
    ts=[run(b) for b in partition_rows(H,n)]
    ms=[t for t,_ in ts]; rs=[r for _,r in ts]
    print('N=%d band ms'%n, ['%.1f'%m for m in ms], 'rays share', ['%.3f'%(r/rf) for r in rs], 'speedup(max) %.2f eff %.2f'%(full/max(ms), full/max(ms)/n))
for n in (2,4,8):
    ts=[run(p["rows"],p["stripe"]) for p in stripe_partition(H,n)]
    ms=[t for t,_ in ts]; rs=[r for _,r in ts]
    print('STRIPES N=%d ms'%n, ['%.1f'%m for m in ms], 'rays share', ['%.3f'%(r/rf) for r in rs], 'speedup(max) %.2f eff %.2f'%(full/max(ms), full/max(ms)/n))
