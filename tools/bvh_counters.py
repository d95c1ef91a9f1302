"""Lane / wave-trip counters of engine BVH's divergent loops (diagnostic -DAMBER_STAMPS build, not the product):
node visits, leaf sphere tests, traversal rounds, shading calls -- how many lanes are busy each time a wave runs them."""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import os; os.environ.setdefault("AMBER_AMD_LIB", "libamber_hip_lab.so")   # known-answer entry points / lab schedulers: the lab build (include/amber_hip_lab.h)
import amber_amd.api as api
CLOCKS = "--clocks" in sys.argv
if CLOCKS: sys.argv.remove("--clocks")
api._LIB_PATH = api._ROOT / "lib" / os.environ.get("AMBER_COUNTERS_LIB", "libamber_hip_clocks.so" if CLOCKS else "libamber_hip_stamps.so")
if not api._LIB_PATH.exists():
    import subprocess; subprocess.run(["make", "-C", str(api._ROOT / "csrc"), "clocks" if CLOCKS else "stamps"], check=True, stdout=subprocess.DEVNULL)
import amber_amd as A
from amber_amd import scenes
lib = A.load_library()
lib.amber_hip_pt_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong * 8)]
n_spheres = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
hs = A.HostScene.create_arrays(**scenes.random_spheres(n_spheres, 7))
LEGACY = os.environ.get("AMBER_BVH_POOL") != "1"       # pt_bvh_megakernel (default) or, with AMBER_BVH_POOL=1, pt_bvh_pool_kernel
pt = A.PathTracer(hs, A.Sensor.default(1920, 1080), seed=1)
pt.render_pass(0, 8); pt.sync(); pt.clear()             # the handle's first launch is a probe of one chunk
z = (C.c_ulonglong * 8)()
assert lib.amber_hip_pt_read_stamps(pt._h, C.byref(z)) == 0; base = list(z)
pt.render_pass(0, spp); pt.sync()
out = (C.c_ulonglong * 8)()
assert lib.amber_hip_pt_read_stamps(pt._h, C.byref(out)) == 0
v = [x - y for x, y in zip(out, base)]; rays = pt.ray_count()
print("engine BVH (%s), %d spheres, 1920x1080 @ %d spp: %d rays" % ("pt_bvh_megakernel" if LEGACY else "pt_bvh_pool_kernel", n_spheres, spp, rays))
if CLOCKS:
    tot = sum(v[:7])      # clocks are kept per lane: a lane is charged for the sections its exec bit is set in
    names = ("work acquisition", "path regeneration + BvhBegin", "N-phase (inner nodes; incl. lanes stopped at a 2nd leaf)", "S-phase (leaf tests; incl. lanes without a leaf)",
             "round control (traversing lanes)", "shading + BvhBegin", "not traversing: waiting for the shading batch") if LEGACY else (
             "swap / trigger checks before a shading pass (all lanes)", "end of a round (all lanes: incl. lanes not in flight during the round)", "N-phase (inner nodes; lanes in flight)",
             "S-phase (leaf tests; lanes in flight)", "round head (lanes in flight)", "shading pass: shade, new paths, slab operands, LDS (all lanes)", "swap / trigger checks before a round (all lanes)")
    for k, name in enumerate(names):
        print("   %-60s %6.2f %% of lane time" % (name, 100.0 * v[k] / tot))
    sys.exit(0)
for k, name in enumerate(("inner-node visits", "leaf sphere tests", "traversal rounds", "path regenerations" if LEGACY else "shading calls")):
    lanes, trips = v[2 * k], v[2 * k + 1]
    print("   %-20s %7.2f per ray, %9.3e wave trips (%6.2f per ray-lane... %5.1f lanes busy per trip = %4.1f %%)"
          % (name, lanes / rays, trips, trips * 64 / rays, lanes / max(trips, 1), 100.0 * lanes / max(trips, 1) / 64))
