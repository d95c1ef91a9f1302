"""One rank's share of config 2 on N GPUs (rank 0's stripes), three launches of 1024 spp: for rocprofv3 --kernel-trace --stats (which kernels make up a step at N = 8?)."""
import sys, os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import amber_amd as A
from amber_amd.distributed import stripe_partition
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
p = stripe_partition(1024, n)[0]
pt = A.PathTracer(A.HostScene.cornell_box(), A.Sensor.default(1024, 1024), rows=p["rows"], stripe=p["stripe"])
pt.render_pass(0, 8); pt.sync(); pt.clear()
import time
for i in range(3):
    t0 = time.perf_counter(); pt.render_pass(0, spp); pt.sync(); wall = time.perf_counter() - t0
    k, ms = pt.kernel_time()
    print("launch %d: wall %.3f ms, kernel_time() %.3f ms over %d launches" % (i, wall * 1e3, ms, k))
