"""Engine BVH where the reference's `--scene` users land: kernel time of
  cornell_bvh     the Cornell box through ENGINE_BVH (and two-phase / list next to it: the cliff at the 33-object engine switch),
  room_mesh       Cornell-like room + 1 304-triangle imported mesh,
  terrain         1.04 M-triangle displaced terrain with needle triangles at the seams,
the mesh scenes written as OBJ + MTL and read back through cli::ImportScene (amber/import.cc).
Usage: python tools/mesh_workloads.py [name ...] [--spp N] [--counters]   (--counters: lane / wave-trip counters, the -DAMBER_STAMPS build)"""
import ctypes as C, os, sys, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import amber_amd.api as api
args = [a for a in sys.argv[1:] if not a.startswith("--")]
CLOCKS = "--clocks" in sys.argv                     # the same sites as per-lane clocks (libamber_hip_clocks.so): where the lane time goes
COUNTERS = "--counters" in sys.argv or CLOCKS
SPP = int(sys.argv[sys.argv.index("--spp") + 1]) if "--spp" in sys.argv else 0
if "--spp" in sys.argv: args.remove(str(SPP))
if COUNTERS:
    api._LIB_PATH = api._ROOT / "lib" / os.environ.get("AMBER_COUNTERS_LIB", "libamber_hip_clocks.so" if CLOCKS else "libamber_hip_stamps.so")
import amber_amd as A
from amber_amd import workloads as WL
lib = A.load_library()
if COUNTERS: lib.amber_hip_pt_read_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong * 8)]


def stamps(pt):
    z = (C.c_ulonglong * 8)()
    assert lib.amber_hip_pt_read_stamps(pt._h, C.byref(z)) == 0
    return list(z)


def run(label, hs, W, H, spp, engine=A.ENGINE_AUTO, paths=None):
    """paths: None = the library's choice; False / True = AMBER_BVH_PATHS=0 / unset (pt_bvh_megakernel against pt_megakernel<ENGINE_BVH> on shallow trees)"""
    if COUNTERS and paths is None: return 0.0                # the counter sites belong to pt_bvh_megakernel: only the item kernel is counted
    if paths is False: os.environ["AMBER_BVH_PATHS"] = "0"
    else: os.environ.pop("AMBER_BVH_PATHS", None)
    t = time.time(); pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=1, engine=engine); t_create = time.time() - t
    pt.render_pass(0, 8); pt.sync(); pt.clear()
    base = stamps(pt) if COUNTERS else None
    pt.render_pass(0, spp); pt.sync()
    n, ms = pt.kernel_time(); r = pt.ray_count()
    print("%-34s %dx%d @ %4d spp: create %.2f s, kernel %8.2f ms (%d launches), %11d rays, %8.1f Mrays/s, %.2f rays/path, contract frac %.4f"
          % (label, W, H, spp, t_create, ms, n, r, r / ms / 1e3, r / (W * H * spp), r * 96 / (ms * 1e-3) / 8e12), flush=True)
    if COUNTERS and engine in (A.ENGINE_AUTO, A.ENGINE_BVH):
        v = [x - y for x, y in zip(stamps(pt), base)]
        if CLOCKS:
            tot = sum(v[:7])
            for k, name in enumerate(("work acquisition", "path regeneration + BvhBegin", "N-phase (inner nodes)", "S-phase (leaf tests)", "round control", "shading + BvhBegin", "waiting for the shading batch")):
                print("      %-34s %6.2f %% of lane time" % (name, 100.0 * v[k] / tot), flush=True)
            pt.close()
            return ms
        for k, name in enumerate(("inner-node visits", "leaf tests (stage 2)", "traversal rounds", "path regenerations")):
            lanes, trips = v[2 * k], v[2 * k + 1]
            print("      %-22s %7.2f per ray, %6.2f wave trips per 64 rays, %5.1f %% of the lanes busy per trip" % (name, lanes / r, trips * 64 / r, 100.0 * lanes / max(trips, 1) / 64), flush=True)
    pt.close()
    return ms


names = args or ["cornell_bvh", "room_mesh", "terrain"]
for name in names:
    if name == "cornell_bvh":
        hs = A.HostScene.cornell_box()
        spp = SPP or 256
        t2 = run("Cornell, two-phase (auto)", hs, 1024, 1024, spp, A.ENGINE_TWO_PHASE) if not COUNTERS else None
        ti = run("Cornell, ENGINE_BVH, item kernel", hs, 1024, 1024, spp, A.ENGINE_BVH, paths=False)
        tb = run("Cornell, ENGINE_BVH", hs, 1024, 1024, spp, A.ENGINE_BVH)
        if t2: print("      engine BVH / two-phase = %.2fx (item kernel %.2fx)" % (tb / t2, ti / t2))
    else:
        if name.startswith("room_mesh"): wl = WL.room_mesh(int(name[9:] or 3))
        else: wl = WL.terrain_mesh(16, 56) if name == "terrain" else WL.terrain_mesh(int(name.split("_")[1]), int(name.split("_")[2]))
        d = tempfile.mkdtemp()
        t = time.time(); path = wl.write(d); t_w = time.time() - t
        t = time.time(); hs = A.HostScene.import_file(path); t_i = time.time() - t
        print("%s: %d triangles; OBJ written in %.1f s, imported in %.1f s" % (wl.name, wl.n_triangles, t_w, t_i), flush=True)
        if name.startswith("room_mesh"):
            run(wl.name + " (BVH, item kernel)", hs, 1024, 1024, SPP or 256, paths=False)
            run(wl.name + " (engine auto = BVH)", hs, 1024, 1024, SPP or 256)
        else: run(wl.name + " (engine auto = BVH)", hs, 1920, 1080, SPP or 64, paths=False if COUNTERS else None)
    hs.close()
