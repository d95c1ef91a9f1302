"""ctypes binding of include/amber_hip.h and include/amber_host.h (no compute in Python)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_ROOT = Path(__file__).resolve().parent
# AMBER_AMD_LIB selects another build next to the product library: libamber_hip_lab.so (tests/ and the tools that use the known-answer entry
# points, signatures or the measured-and-kept schedulers) or a measurement build (stamps, portable math); default = the product
_LIB_PATH = _ROOT / "lib" / os.environ.get("AMBER_AMD_LIB", "libamber_hip.so")
_lib = None


class AmberError(RuntimeError):
    """Raised for every non-zero return code of the C ABI (message = amber_hip_last_error())."""


# ---- plain-data structs (include/amber_hip.h) ---------------------------------------------------
class FlatObject(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("material", C.c_uint32), ("p", C.c_float * 12)]


class FlatMaterial(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("rho", C.c_float * 3), ("param", C.c_float), ("r0", C.c_float)]


class FlatThinLens(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("global_", C.c_float * 9), ("local_", C.c_float * 9),
                ("focus_distance", C.c_float), ("sensor_distance", C.c_float), ("p_area", C.c_float),
                ("n_blades", C.c_uint32), ("first_blade_object", C.c_uint32), ("kind", C.c_uint32)]


class FlatLight(C.Structure):
    _fields_ = [("object", C.c_uint32), ("cum_power", C.c_float), ("pdf_area", C.c_float), ("irradiance", C.c_float * 3)]


class FlatSceneC(C.Structure):
    _fields_ = [("objects", C.POINTER(FlatObject)), ("n_objects", C.c_uint32),
                ("materials", C.POINTER(FlatMaterial)), ("n_materials", C.c_uint32), ("lens", FlatThinLens),
                ("lights", C.POINTER(FlatLight)), ("n_lights", C.c_uint32)]


class Splat(C.Structure):
    _fields_ = [("path", C.c_uint32), ("sample", C.c_uint32), ("bounce", C.c_uint32), ("pixel", C.c_uint32),
                ("rgb", C.c_float * 3), ("pad", C.c_uint32)]


class Sensor(C.Structure):
    """rendering::Sensor(width, height, scene_width, scene_height) -- application.cc:89-94."""
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("scene_width", C.c_float), ("scene_height", C.c_float)]

    @classmethod
    def default(cls, width: int, height: int) -> "Sensor":
        # application.cc:89-94: Sensor(w, h, 0.036, 0.036 / w * h) -- doubles narrowed to real_type
        return cls(width, height, np.float32(0.036), np.float32(0.036 / width * height))


class PtParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("max_depth", C.c_uint32), ("device", C.c_int32), ("row_begin", C.c_uint32),
                ("row_end", C.c_uint32), ("stream", C.c_void_p), ("engine", C.c_uint32), ("stripe_rows", C.c_uint32),
                ("stripe_period", C.c_uint32), ("reserved", C.c_uint32)]


class HostStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("passes", C.c_uint64), ("launches", C.c_uint32), ("pad", C.c_uint32), ("kernel_ms", C.c_double)]


PRIM_TRIANGLE, PRIM_SPHERE, PRIM_DISK, PRIM_CYLINDER = 0, 1, 2, 3
MAT_LAMBERTIAN, MAT_PHONG, MAT_SPECULAR, MAT_REFRACTION, MAT_DIFFUSE_LIGHT, MAT_EYE = 0, 1, 2, 3, 4, 5
ENGINE_AUTO, ENGINE_LIST, ENGINE_TWO_PHASE, ENGINE_BVH, ENGINE_WAVEFRONT = 0, 1, 2, 3, 4
ENGINE_REFERENCE_BVH = 6     # the reference's own tree and traversal order: every ray gets the hit the reference's BVH gives it (include/amber_hip.h)
PT_FLAG_NULL_STREAM, PT_FLAG_BVH_POOL, PT_FLAG_BVH_ITEMS = 1, 2, 4

# every symbol include/amber_hip.h and include/amber_host.h declare: what libamber_hip.so (the product) exports
ABI_SYMBOLS = [
    "amber_hip_pt_create", "amber_hip_pt_render_pass", "amber_hip_pt_clear", "amber_hip_pt_sync",
    "amber_hip_pt_download", "amber_hip_pt_device_framebuffer", "amber_hip_pt_stream", "amber_hip_pt_local_rows", "amber_hip_pt_kernel_time", "amber_hip_pt_destroy",
    "amber_hip_last_error", "amber_hip_abi_version", "amber_hip_math_mode", "amber_hip_device_count", "amber_hip_lt_trace", "amber_hip_lt_trace_range",
    "amber_host_cornell_box", "amber_host_scene_import", "amber_host_scene_create", "amber_host_scene_destroy", "amber_host_scene_flatten",
    "amber_host_pt_create", "amber_host_render", "amber_host_render_devices", "amber_host_last_error", "amber_host_tonemap", "amber_host_export",
]
# what include/amber_hip_lab.h declares: libamber_hip_lab.so (the same sources with -DAMBER_LAB) exports these as well
LAB_SYMBOLS = [
    "amber_hip_kat_cast", "amber_hip_kat_sample", "amber_hip_kat_eye", "amber_hip_kat_trace", "amber_hip_kat_math", "amber_hip_kat_signatures", "amber_hip_pt_signatures",
    "amber_hip_kat_traversal_rate", "amber_hip_kat_pixel_masks",
]
PRODUCT_LIB, LAB_LIB = "libamber_hip.so", "libamber_hip_lab.so"


def library_path() -> Path:
    return _LIB_PATH


def build_library(force: bool = False) -> Path:
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU).  make is incremental, so it
    always runs where a compiler exists: an edited .hip/.h can never be tested against a stale library.  A box
    without hipcc uses the prebuilt library that travelled with the tree."""
    import shutil
    if shutil.which(os.environ.get("HIPCC", "hipcc")) or not _LIB_PATH.exists():
        subprocess.run(["make", "-C", str(_ROOT / "csrc")] + (["-B"] if force else []), check=True)
    return _LIB_PATH


def is_lab() -> bool:
    """True when the loaded library is a lab build (it has the entry points of include/amber_hip_lab.h)."""
    return hasattr(load_library(), "amber_hip_kat_cast")


def load_library() -> C.CDLL:
    """Load libamber_hip.so (or the build AMBER_AMD_LIB names).  Fails loudly: there is no CPU implementation to fall back to."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise AmberError(f"{_LIB_PATH} is missing: build it with `make -C amber_amd/csrc` "
                         "(or __graft_entry__.build()); amber_amd has no CPU fallback")
    lib = C.CDLL(str(_LIB_PATH))
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32
    lib.amber_hip_last_error.restype = C.c_char_p
    lib.amber_host_last_error.restype = C.c_char_p
    lib.amber_hip_pt_create.argtypes = [C.POINTER(FlatSceneC), C.POINTER(Sensor), C.POINTER(PtParams), C.POINTER(vp)]
    lib.amber_hip_pt_render_pass.argtypes = [vp, u32, u32]
    lib.amber_hip_pt_clear.argtypes = [vp]
    lib.amber_hip_pt_sync.argtypes = [vp]
    lib.amber_hip_pt_download.argtypes = [vp, vp, C.POINTER(u64)]
    lib.amber_hip_pt_device_framebuffer.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    lib.amber_hip_pt_local_rows.argtypes = [vp, C.POINTER(u32)]
    lib.amber_hip_pt_stream.argtypes = [vp, C.POINTER(vp)]
    lib.amber_hip_pt_kernel_time.argtypes = [vp, C.POINTER(u32), C.POINTER(C.c_double)]
    lib.amber_hip_pt_destroy.argtypes = [vp]
    lib.amber_hip_pt_destroy.restype = None
    if hasattr(lib, "amber_hip_lt_trace"):     # absent only in older builds loaded by tools/ab_lib.py
        lib.amber_hip_lt_trace.argtypes = [vp, u32, u32, vp, u32, C.POINTER(u32), C.POINTER(u64)]
    if hasattr(lib, "amber_hip_lt_trace_range"):
        lib.amber_hip_lt_trace_range.argtypes = [vp, u32, u32, u32, u32, vp, u32, C.POINTER(u32), C.POINTER(u64)]
    if hasattr(lib, "amber_hip_kat_cast"):         # the lab build (include/amber_hip_lab.h); the product exports none of these
        lib.amber_hip_kat_cast.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp]
        lib.amber_hip_kat_sample.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp]
        lib.amber_hip_kat_eye.argtypes = [vp, u32, vp, vp, vp]
        lib.amber_hip_kat_trace.argtypes = [vp, u32, vp, vp, u32, vp, vp]
        lib.amber_hip_kat_math.argtypes = [i32, i32, u32, vp, vp]
        lib.amber_hip_kat_signatures.argtypes = [vp, u32, u32, vp]
        lib.amber_hip_kat_pixel_masks.argtypes = [vp, vp, vp, C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
        lib.amber_hip_kat_traversal_rate.argtypes = [vp, u32, vp, vp, u32, u32, u32, vp, vp, C.POINTER(C.c_double), vp]
        lib.amber_hip_pt_signatures.argtypes = [vp, u32, u32, vp]
    lib.amber_host_cornell_box.restype = vp
    lib.amber_host_cornell_box.argtypes = [C.c_float, C.c_float, u32]
    lib.amber_host_scene_import.restype = vp
    lib.amber_host_scene_import.argtypes = [C.c_char_p]
    lib.amber_host_scene_create.restype = vp
    lib.amber_host_scene_create.argtypes = [C.POINTER(FlatObject), u32, C.POINTER(FlatMaterial), u32, C.POINTER(C.c_float),
                                            C.c_float, C.c_float, C.c_float, u32, C.c_int]   # n_blades == 0 selects the pinhole lens
    lib.amber_host_scene_destroy.argtypes = [vp]
    lib.amber_host_scene_destroy.restype = None
    lib.amber_host_scene_flatten.argtypes = [vp, C.POINTER(FlatObject), C.POINTER(u32), C.POINTER(FlatMaterial), C.POINTER(u32),
                                             C.POINTER(FlatThinLens)]
    lib.amber_host_pt_create.argtypes = [vp, C.POINTER(Sensor), C.POINTER(PtParams), C.POINTER(vp)]
    lib.amber_host_render.argtypes = [vp, C.c_char_p, C.POINTER(Sensor), u32, u64, u32, C.c_int, u32, vp, C.POINTER(HostStats)]
    lib.amber_host_render_devices.argtypes = [vp, C.c_char_p, C.POINTER(Sensor), u32, u64, u32, C.POINTER(C.c_int), u32, u32, vp, C.POINTER(HostStats)]
    lib.amber_host_tonemap.argtypes = [vp, u32, u32, vp]
    lib.amber_host_export.argtypes = [vp, u32, u32, C.c_char_p, C.c_char_p]
    _lib = lib
    return lib


def _check(rc: int, host: bool = False) -> None:
    if rc != 0:
        lib = load_library()
        msg = (lib.amber_host_last_error() if host else lib.amber_hip_last_error()) or b""
        if host and not msg:
            msg = lib.amber_hip_last_error() or b""
        raise AmberError(f"amber error {rc}: {msg.decode(errors='replace')}")


def device_count() -> int:
    return int(load_library().amber_hip_device_count())


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


class HostScene:
    """Handle of a scene built by the C++ host object model (amber::scene::Scene)."""

    def __init__(self, handle):
        if not handle:
            raise AmberError("scene creation failed: " + (load_library().amber_host_last_error() or b"").decode())
        self._h = C.c_void_p(handle)

    @classmethod
    def cornell_box(cls, focal_length: float = 0.050, aperture_radius: float = 0.050, n_blades: int = 6) -> "HostScene":
        """etude::CornelBox(0.050, 0.050, 6) -- application.cc:68-73."""
        return cls(load_library().amber_host_cornell_box(focal_length, aperture_radius, n_blades))

    @classmethod
    def import_file(cls, filename) -> "HostScene":
        """cli::ImportScene(filename) + Scene::Create<BVH> (import.cc:49-167, application.cc:74-86); OBJ + MTL subset."""
        return cls(load_library().amber_host_scene_import(str(filename).encode()))

    @classmethod
    def create(cls, objects, materials, transform, focal_length, focus_distance, radius, n_blades, accel: int = 0) -> "HostScene":
        """objects: list of (kind, material, params...) ; materials: list of (kind, (r,g,b), param)."""
        objs = (FlatObject * max(1, len(objects)))()
        for i, (kind, mat, params) in enumerate(objects):
            objs[i].kind, objs[i].material = kind, mat
            for j, v in enumerate(params):
                objs[i].p[j] = v
        mats = (FlatMaterial * max(1, len(materials)))()
        for i, (kind, rho, param) in enumerate(materials):
            mats[i].kind, mats[i].param = kind, param
            for j in range(3):
                mats[i].rho[j] = rho[j]
        t = (C.c_float * 16)(*[float(x) for x in transform])
        return cls(load_library().amber_host_scene_create(objs, len(objects), mats, len(materials), t, focal_length,
                                                          focus_distance, radius, n_blades, accel))

    @classmethod
    def create_arrays(cls, kinds, material_index, params, materials, transform, focal_length, focus_distance, radius, n_blades,
                      accel: int = 0) -> "HostScene":
        """Bulk form of create(): kinds (n,) u32, material_index (n,) u32, params (n,12) f32 as numpy arrays."""
        n = len(kinds)
        dt = np.dtype([("kind", np.uint32), ("material", np.uint32), ("p", np.float32, (12,))])
        arr = np.zeros(n, dt)
        arr["kind"], arr["material"] = kinds, material_index
        arr["p"][:, : np.asarray(params).shape[1]] = params
        assert arr.itemsize == C.sizeof(FlatObject)
        mats = (FlatMaterial * max(1, len(materials)))()
        for i, (kind, rho, param) in enumerate(materials):
            mats[i].kind, mats[i].param = kind, param
            for j in range(3):
                mats[i].rho[j] = rho[j]
        t = (C.c_float * 16)(*[float(x) for x in transform])
        return cls(load_library().amber_host_scene_create(arr.ctypes.data_as(C.POINTER(FlatObject)), n, mats, len(materials), t,
                                                          focal_length, focus_distance, radius, n_blades, accel))

    def flatten(self):
        lib = load_library()
        no, nm = C.c_uint32(0), C.c_uint32(0)
        _check(lib.amber_host_scene_flatten(self._h, None, C.byref(no), None, C.byref(nm), None), host=True)
        objs, mats, lens = (FlatObject * no.value)(), (FlatMaterial * nm.value)(), FlatThinLens()
        _check(lib.amber_host_scene_flatten(self._h, objs, C.byref(no), mats, C.byref(nm), C.byref(lens)), host=True)
        return objs, mats, lens

    def render(self, sensor: Sensor, spp: int, seed: int = 12345, max_depth: int = 0, device: int = 0,
               samples_per_launch: int = 0, algorithm: str = "pt", devices=None):
        """Algorithm<RGB>::Render through cli::MakeAlgorithm(algorithm) and cli::Context(1, spp).
        devices: list of HIP ordinals -> one engine handle per entry inside the one Render() call (HipPathTracingOptions.devices)."""
        out = np.empty((sensor.height, sensor.width, 3), np.float32)
        st = HostStats()
        devs = list(devices) if devices else [device]
        arr = (C.c_int * len(devs))(*devs)
        _check(load_library().amber_host_render_devices(self._h, algorithm.encode(), C.byref(sensor), spp, seed, max_depth, arr, len(devs),
                                                        samples_per_launch, out.ctypes.data, C.byref(st)), host=True)
        return out, {"rays": st.rays, "passes": st.passes, "launches": st.launches, "kernel_ms": st.kernel_ms}

    def close(self):
        if self._h:
            load_library().amber_host_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PathTracer:
    """amber_hip_pt handle: the device-side engine for one band of the framebuffer on one GPU."""

    def __init__(self, scene: HostScene, sensor: Sensor, seed: int = 12345, max_depth: int = 0, device: int = 0,
                 rows=None, stream: int | None = None, engine: int = ENGINE_AUTO, stripe=None, flags: int = 0):
        """rows = (y0, y1) contiguous band; stripe = (S, period) keeps only rows with (y - y0) % period < S;
        flags = AMBER_PT_FLAG_* bits (PT_FLAG_BVH_POOL: engine BVH with the per-wave ray pool scheduler)."""
        self.sensor = sensor
        rb, re = rows if rows is not None else (0, sensor.height)
        s_rows, s_period = stripe if stripe else (0, 0)
        p = PtParams(seed, max_depth, device, rb, re, stream, engine, s_rows, s_period, flags)
        h = C.c_void_p()
        _check(load_library().amber_host_pt_create(scene._h, C.byref(sensor), C.byref(p), C.byref(h)), host=True)
        self._h = h
        ys = np.arange(rb, re)
        self.row_index = ys[(ys - rb) % s_period < s_rows] if s_rows else ys   # global row of every local row
        n = C.c_uint32()
        _check(load_library().amber_hip_pt_local_rows(self._h, C.byref(n)))
        assert n.value == len(self.row_index)
        self.rows = (rb, re)

    @property
    def band_shape(self):
        return (len(self.row_index), self.sensor.width, 3)

    def render_pass(self, first_sample: int, n_samples: int) -> None:
        _check(load_library().amber_hip_pt_render_pass(self._h, first_sample, n_samples))

    def clear(self) -> None:
        _check(load_library().amber_hip_pt_clear(self._h))

    def sync(self) -> None:
        _check(load_library().amber_hip_pt_sync(self._h))

    def download(self):
        out = np.empty(self.band_shape, np.float32)
        rays = C.c_uint64()
        _check(load_library().amber_hip_pt_download(self._h, out.ctypes.data, C.byref(rays)))
        return out, rays.value

    def ray_count(self) -> int:
        rays = C.c_uint64()
        _check(load_library().amber_hip_pt_download(self._h, None, C.byref(rays)))
        return rays.value

    def stream(self) -> int:
        """hipStream_t (as an integer) the handle's work is enqueued on; wrap it with torch.cuda.ExternalStream to order
        collectives after the render."""
        p = C.c_void_p()
        _check(load_library().amber_hip_pt_stream(self._h, C.byref(p)))
        return p.value or 0

    def device_framebuffer(self):
        p, n = C.c_void_p(), C.c_uint64()
        _check(load_library().amber_hip_pt_device_framebuffer(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def kernel_time(self):
        n, ms = C.c_uint32(), C.c_double()
        _check(load_library().amber_hip_pt_kernel_time(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def lt_trace(self, first_sample: int, n_samples: int, capacity: int = 1 << 16, paths=None):
        """Light tracing (algorithm_lt.cc): splats of W*H light paths per pass (or of the light paths [paths[0], paths[1])),
        sorted (pass, path, bounce).  Returns (structured numpy array of records, ray count)."""
        dt = np.dtype([("path", np.uint32), ("sample", np.uint32), ("bounce", np.uint32), ("pixel", np.uint32), ("rgb", np.float32, (3,)), ("pad", np.uint32)])
        while True:
            out = np.zeros(capacity, dt)
            n, rays = C.c_uint32(), C.c_uint64()
            if paths is None:
                rc = load_library().amber_hip_lt_trace(self._h, first_sample, n_samples, out.ctypes.data, capacity, C.byref(n), C.byref(rays))
            else:
                rc = load_library().amber_hip_lt_trace_range(self._h, first_sample, n_samples, paths[0], paths[1], out.ctypes.data, capacity,
                                                             C.byref(n), C.byref(rays))
            if rc == -4 and n.value > capacity:
                capacity = n.value
                continue
            _check(rc)
            return out[: n.value], rays.value

    # ---- known-answer entry points -----------------------------------------------------------
    def kat_cast(self, origins, dirs):
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        n = len(o)
        obj, t = np.empty(n, np.int32), np.empty(n, np.float32)
        pos, nrm = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32)
        _check(load_library().amber_hip_kat_cast(self._h, n, o.ctypes.data, d.ctypes.data, obj.ctypes.data, t.ctypes.data,
                                                 pos.ctypes.data, nrm.ctypes.data))
        return obj, t, pos, nrm

    def kat_sample(self, material, normals, dirs_out, rng_state):
        m = np.ascontiguousarray(material, np.uint32)
        nn, dd = _f32(normals).reshape(-1, 3), _f32(dirs_out).reshape(-1, 3)
        st = np.ascontiguousarray(rng_state, np.uint64).copy()
        n = len(m)
        di, w = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32)
        _check(load_library().amber_hip_kat_sample(self._h, n, m.ctypes.data, nn.ctypes.data, dd.ctypes.data, st.ctypes.data,
                                                   di.ctypes.data, w.ctypes.data))
        return di, w, st

    def kat_eye(self, pixel, sample):
        p, s = np.ascontiguousarray(pixel, np.uint32), np.ascontiguousarray(sample, np.uint32)
        out = np.empty((len(p), 7), np.float32)
        _check(load_library().amber_hip_kat_eye(self._h, len(p), p.ctypes.data, s.ctypes.data, out.ctypes.data))
        return out

    def kat_trace(self, pixel, sample, max_bounces: int = 16):
        p, s = np.ascontiguousarray(pixel, np.uint32), np.ascontiguousarray(sample, np.uint32)
        rec = np.zeros((len(p), max_bounces, 11), np.uint32)
        casts = np.zeros(len(p), np.uint32)
        _check(load_library().amber_hip_kat_trace(self._h, len(p), p.ctypes.data, s.ctypes.data, max_bounces, rec.ctypes.data,
                                                  casts.ctypes.data))
        return rec, casts

    def kat_traversal_rate(self, origins, dirs, waves: int = 5, refill_min: int = 16, repeats: int = 3, rounds=None):
        """Engine BVH's traversal alone: returns (object index, t, best kernel ms) for the rays; `rounds` (uint32 array of len(rays),
        optional) receives the number of wave rounds each ray was in flight for."""
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        n = len(o)
        obj, t, ms = np.empty(n, np.int32), np.empty(n, np.float32), C.c_double()
        _check(load_library().amber_hip_kat_traversal_rate(self._h, n, o.ctypes.data, d.ctypes.data, waves, refill_min, repeats, t.ctypes.data,
                                                           obj.ctypes.data, C.byref(ms), rounds.ctypes.data if rounds is not None else None))
        return obj, t, ms.value

    def kat_signatures(self, first_sample: int, n_samples: int) -> np.ndarray:
        """(rows, width, n_samples) uint64: low word = hash of the path's hit-object sequence, high word = hash of its hit distances."""
        rows, width, _ = self.band_shape
        out = np.zeros((rows, width, n_samples), np.uint64)
        _check(load_library().amber_hip_kat_signatures(self._h, first_sample, n_samples, out.ctypes.data))
        return out

    def pixel_masks(self):
        """(masks (rows, width) uint32, duration of pixel_mask_kernel in ms): the candidate mask of every band pixel's eye rays (two-phase engine)."""
        rows, width, _ = self.band_shape
        masks = np.zeros((rows, width), np.uint32)
        ms = C.c_double()
        _check(load_library().amber_hip_kat_pixel_masks(self._h, masks.ctypes.data, None, None, C.byref(ms)))
        return masks, ms.value

    def object_slots(self, n_objects: int):
        """(filter-program slot (mask bit) of every scene object, 0xffffffff = none; mask of the slots that are candidates of every ray -- objects
        without a filter record and the aperture blades).  Works with AMBER_PIXEL_MASK=0 and on an empty band: it does not need the mask kernel."""
        slots = np.zeros(n_objects, np.uint32)
        always = C.c_uint32()
        _check(load_library().amber_hip_kat_pixel_masks(self._h, None, slots.ctypes.data, C.byref(always), None))
        return slots, always.value

    def render_signatures(self, first_sample: int, n_samples: int) -> np.ndarray:
        """kat_signatures' layout and meaning, produced by the PRODUCT render kernel (its signature instantiation)."""
        rows, width, _ = self.band_shape
        out = np.zeros((rows, width, n_samples), np.uint64)
        _check(load_library().amber_hip_pt_signatures(self._h, first_sample, n_samples, out.ctypes.data))
        return out

    def close(self):
        if getattr(self, "_h", None):
            load_library().amber_hip_pt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def kat_math(mode: int, x, device: int = 0) -> np.ndarray:
    """The engine's sin/cos/pow on device: mode 0 sincos(x[i]) -> (n,2) ; mode 1 pow(x[i,0], x[i,1]) -> (n,) ;
    mode 2 / 3: x[i]^4 / x[i]^5 in binary64 -> (n,) float64."""
    x = _f32(x)
    n = len(x)
    out = np.empty((n,) if mode == 1 else (n, 2), np.float32)
    _check(load_library().amber_hip_kat_math(device, mode, n, x.ctypes.data, out.ctypes.data))
    return out.view(np.float64).reshape(n) if mode >= 2 else out


MATH_PORTABLE, MATH_GLIBC = 1, 2


def math_mode() -> int:
    """amber_hip_math_mode(): MATH_GLIBC for the product build, MATH_PORTABLE for -DAMBER_BUILD_PORTABLE_MATH measurement builds."""
    return int(load_library().amber_hip_math_mode())


def tonemap(image) -> np.ndarray:
    """Filmic -> Gamma 2.2 -> 8-bit (postprocess/filmic.cc, gamma.cc), image (H, W, 3) float32."""
    img = _f32(image)
    h, w, _ = img.shape
    out = np.empty((h, w, 3), np.uint8)
    _check(load_library().amber_host_tonemap(img.ctypes.data, w, h, out.ctypes.data), host=True)
    return out


def export(image, png_path=None, exr_path=None) -> None:
    """cli::ExportPNG (tone-mapped) / cli::ExportEXR (raw), x-mirrored as the reference writes them."""
    img = _f32(image)
    h, w, _ = img.shape
    _check(load_library().amber_host_export(img.ctypes.data, w, h, png_path.encode() if png_path else None,
                                            exr_path.encode() if exr_path else None), host=True)
