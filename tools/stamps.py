"""Section shares of one megakernel loop iteration from the diagnostic -DAMBER_STAMPS build (not the product)."""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import os; os.environ.setdefault("AMBER_AMD_LIB", "libamber_hip_lab.so")   # known-answer entry points / lab schedulers: the lab build (include/amber_hip_lab.h)
import amber_amd.api as api
api._LIB_PATH = api._ROOT / "lib" / "libamber_hip_stamps.so"
if not api._LIB_PATH.exists():                      # measurement builds are not kept in the tree: build on demand (30 s)
    import subprocess; subprocess.run(["make", "-C", str(api._ROOT / "csrc"), "stamps"], check=True, stdout=subprocess.DEVNULL)
import amber_amd as A
lib = A.load_library()
lib.amber_hip_pt_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong * 8)]
sc = A.HostScene.cornell_box(); sn = A.Sensor.default(1024, 1024)
for eng, name in ((A.ENGINE_TWO_PHASE, "two_phase"), (A.ENGINE_LIST, "list")):
    pt = A.PathTracer(sc, sn, engine=eng)
    pt.render_pass(0, 64); pt.sync()
    out = (C.c_ulonglong * 8)()
    assert lib.amber_hip_pt_read_stamps(pt._h, C.byref(out)) == 0
    v = list(out); tot = sum(v)
    names = ["0 acquire", "1 regenerate", "2 closest-hit phase A (list: n/a)", "3 closest-hit phase B (list: whole scan)", "4 resolve+material+Le", "5 sample+pow/sincos", "6 RR/update/loop", "7 -"]
    print(name, "rays", pt.ray_count())
    trips, tests = v[7] & 0xffffffff, v[7] >> 32   # NOTE: 32-bit halves overflow on long runs; fine at 64 spp
    tot -= v[7]
    for n, x in zip(names[:7], v[:7]):
        print("   %-42s %6.2f %%" % (n, 100.0 * x / tot))
    if trips:
        rays = pt.ray_count()
        print("   phase B triangle loop: %.2f exact tests per ray, %.1f lanes busy per trip" % (tests / rays, tests / trips))
