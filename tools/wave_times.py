"""When do the waves of a pt_megakernel launch start, make their first claim, find the queue empty and end?  Diagnostic -DAMBER_STAMPS
build (make stamps).  Looks for the fixed 0.5 ms of a launch (tools/launch_fixed_cost.py).   python tools/wave_times.py [spp]"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import os; os.environ.setdefault("AMBER_AMD_LIB", "libamber_hip_lab.so")   # known-answer entry points / lab schedulers: the lab build (include/amber_hip_lab.h)
import amber_amd.api as api
api._LIB_PATH = api._ROOT / "lib" / "libamber_hip_stamps.so"
import amber_amd as A
lib = A.load_library()
lib.amber_hip_pt_read_wave_times.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
pt = A.PathTracer(A.HostScene.cornell_box(), A.Sensor.default(1024, 1024))
pt.render_pass(0, 8); pt.sync(); pt.clear()
for rep in range(2):
    k0, m0 = pt.kernel_time(); pt.render_pass(0, spp); pt.sync(); k1, m1 = pt.kernel_time()
n_waves = 6144
t = np.zeros((n_waves, 4), np.uint64)
assert lib.amber_hip_pt_read_wave_times(pt._h, t.ctypes.data, n_waves) == 0
t = t[t[:, 0] > 0].astype(np.float64) / 100.0          # 100 MHz ticks -> us
t0 = t[:, 0].min()
print("%d spp: kernel %.3f ms by events (stamps build); %d waves reported" % (spp, m1 - m0, len(t)))
def q(name, x):
    print("  %-34s min %8.1f  p10 %8.1f  p50 %8.1f  p90 %8.1f  max %8.1f us" % (name, x.min(), *np.percentile(x, [10, 50, 90]), x.max()))
q("wave start", t[:, 0] - t0)
q("first claim done", t[:, 1] - t0)
q("queue found empty", t[:, 2] - t0)
q("wave end", t[:, 3] - t0)
q("end - queue empty (drain)", t[:, 3] - t[:, 2])
q("first claim - start", t[:, 1] - t[:, 0])
end = t[:, 3] - t0
print("  mean end %.1f us, last end %.1f us: idle lane-time at the end %.1f %% of the launch" % (end.mean(), end.max(), 100 * (end.max() - end.mean()) / end.max()))
