"""Where the time between amber_hip_pt_create and the end of a handle's first launch goes (config 2's frame): create, the host side of the
first render_pass (allocations + enqueue), the wait for the GPU; with and without the per-pixel masks (AMBER_PIXEL_MASK=0)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import amber_amd as A
hs = A.HostScene.cornell_box()
sn = A.Sensor.default(1024, 1024)
A.PathTracer(hs, sn).close()
for label, env in (("default (masks per 4x4 block, enqueued by create)", None), ("AMBER_PIXEL_MASK=0", "0")):
    if env is None: os.environ.pop("AMBER_PIXEL_MASK", None)
    else: os.environ["AMBER_PIXEL_MASK"] = env
    best = None
    for _ in range(5):
        t0 = time.perf_counter(); pt = A.PathTracer(hs, sn, seed=12345)
        t1 = time.perf_counter(); pt.render_pass(0, 8)
        t2 = time.perf_counter(); pt.sync()
        t3 = time.perf_counter(); n, ms = pt.kernel_time()
        pt.render_pass(8, 1016)
        t4 = time.perf_counter(); pt.sync()
        t5 = time.perf_counter(); n2, ms2 = pt.kernel_time()
        pt.close()
        cur = ((t3 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, ms, (t4 - t3) * 1e3, (t5 - t4) * 1e3, ms2 - ms)
        if best is None or cur[0] < best[0]: best = cur
    print("%-50s create -> first launch done %.3f ms = create %.3f + render_pass(0, 8) host side %.3f + wait %.3f (probe kernel %.3f ms); second launch: host %.3f, wait %.3f, kernel %.3f" % ((label,) + best), flush=True)
