"""Is the fixed 0.6 ms of a pt_megakernel launch (tools/launch_fixed_cost.py) paid by a launch that follows another one without a
host synchronisation in between?  Kernel times by the handle's own events: 1 launch + sync, against 8 launches enqueued back to back."""
import sys, os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import amber_amd as A
pt = A.PathTracer(A.HostScene.cornell_box(), A.Sensor.default(1024, 1024))
pt.render_pass(0, 8); pt.sync(); pt.clear()
for spp in (32, 128, 1024):
    single = []
    for rep in range(4):
        k0, m0 = pt.kernel_time(); pt.render_pass(0, spp); pt.sync(); k1, m1 = pt.kernel_time(); single.append(m1 - m0)
    k0, m0 = pt.kernel_time()
    for rep in range(8): pt.render_pass(rep * spp, spp)
    pt.sync(); k1, m1 = pt.kernel_time()
    print("%5d spp: one launch then sync %.3f ms (best of 4); 8 launches back to back: %.3f ms each (%d launches)" % (spp, min(single), (m1 - m0) / (k1 - k0), k1 - k0), flush=True)
