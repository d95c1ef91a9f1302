"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE (see oracle/amber_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "oracle" / "liboracle.so"

ACCEL_BVH, ACCEL_LIST, ACCEL_BVH_CONS = 0, 1, 2   # BVH_CONS: List's answer through the reference's tree (amber_oracle.h)
BLADES_LAST = 0x100          # ORACLE_BLADES_LAST: aperture blades after the objects (cli::ImportScene order)
ACCUM_CHUNK = 8    # include/amber_hip.h AMBER_ACCUM_CHUNK
MATH_LIBM, MATH_PORTABLE, MATH_GLIBC = 0, 1, 2
# The oracle mode that restates the arithmetic the loaded libamber_hip.so executes (amber_hip_math_mode()): GLIBC for the
# product build.  conftest's `amber` fixture switches it to PORTABLE when a -DAMBER_BUILD_PORTABLE_MATH measurement build is loaded.
MATH_DEVICE = MATH_GLIBC


def _m(math):
    return MATH_DEVICE if math is None else math



class OObject(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("material", C.c_uint32), ("p", C.c_float * 9)]


class OMaterial(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("rho", C.c_float * 3), ("param", C.c_float)]


class OThinLens(C.Structure):
    _fields_ = [("transform", C.c_float * 16), ("focal_length", C.c_float), ("focus_distance", C.c_float),
                ("radius", C.c_float), ("n_blades", C.c_uint32)]


class OSensor(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("scene_width", C.c_float), ("scene_height", C.c_float)]


class OCounters(C.Structure):
    _fields_ = [("casts", C.c_uint64), ("hits", C.c_uint64), ("paths", C.c_uint64)]


class OBounce(C.Structure):
    _fields_ = [("object", C.c_int32), ("t", C.c_float), ("pos", C.c_float * 3), ("weight", C.c_float * 3),
                ("measurement", C.c_float * 3)]


_lib = None


def build(force: bool = False) -> Path:
    if force or not LIB.exists() or LIB.stat().st_mtime < (ROOT / "oracle" / "amber_oracle.cc").stat().st_mtime:
        subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
    return LIB


def load():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(str(LIB))
    vp, u32, u64, f = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float
    L.oracle_scene_cornell_box.restype = vp
    L.oracle_scene_cornell_box.argtypes = [f, f, u32, C.c_int]
    L.oracle_scene_create.restype = vp
    L.oracle_scene_create.argtypes = [C.POINTER(OObject), u32, C.POINTER(OMaterial), u32, C.POINTER(OThinLens), C.c_int]
    L.oracle_scene_destroy.argtypes = [vp]
    L.oracle_scene_set_accel.argtypes = [vp, C.c_int]
    L.oracle_cast_many.argtypes = [vp, C.c_int, u64, vp, vp, u32, vp, vp]
    L.oracle_collect_rays.restype = u64
    L.oracle_collect_rays.argtypes = [vp, C.POINTER(OSensor), u64, u32, u32, u32, u32, C.c_int, u32, u64, vp, vp]
    L.oracle_classify_path.argtypes = [vp, C.POINTER(OSensor), u64, u32, u32, u32, C.c_int, u32, C.c_int, C.c_int, C.POINTER(u32)]
    L.oracle_scene_object_count.argtypes = [vp]
    L.oracle_scene_object_count.restype = u32
    L.oracle_scene_material_count.argtypes = [vp]
    L.oracle_scene_material_count.restype = u32
    L.oracle_scene_get_object.argtypes = [vp, u32, C.POINTER(OObject), C.POINTER(f)]
    L.oracle_scene_get_material.argtypes = [vp, u32, C.POINTER(OMaterial), C.POINTER(f)]
    L.oracle_scene_get_lens.argtypes = [vp] + [C.POINTER(f)] * 6
    L.oracle_scene_bvh_stats.argtypes = [vp] + [C.POINTER(u32)] * 3
    L.oracle_scene_bvh_order.argtypes = [vp, vp]
    L.oracle_scene_bvh_digest.argtypes = [vp]
    L.oracle_scene_bvh_digest.restype = u64
    L.oracle_render_mt.argtypes = [vp, C.POINTER(OSensor), u64, u32, C.c_int, vp, C.POINTER(OCounters)]
    L.oracle_render_xorshift.argtypes = [vp, C.POINTER(OSensor), u64, u32, u32, u32, u32, C.c_int, u32, u32, u32, vp, C.POINTER(OCounters)]
    L.oracle_render_lt_xorshift.restype = u64
    L.oracle_render_lt_xorshift.argtypes = [vp, C.POINTER(OSensor), u64, u32, u32, C.c_int, u32, vp, C.POINTER(OCounters), vp, u64]
    L.oracle_path_signatures.argtypes = [vp, C.POINTER(OSensor), u64, u32, u32, u32, u32, C.c_int, u32, u32, vp]
    L.oracle_trace_path.restype = u32
    L.oracle_trace_path.argtypes = [vp, C.POINTER(OSensor), u64, u32, u32, u32, C.c_int, u32, C.POINTER(OBounce), u32, C.POINTER(f)]
    L.oracle_cast.restype = C.c_int32
    L.oracle_cast.argtypes = [vp, C.POINTER(f), C.POINTER(f), C.POINTER(f), C.POINTER(f), C.POINTER(f)]
    L.oracle_intersect.argtypes = [C.POINTER(OObject), C.POINTER(f), C.POINTER(f), C.POINTER(f), C.POINTER(f), C.POINTER(f)]
    L.oracle_aabb_intersect.argtypes = [C.POINTER(f)] * 4 + [f, C.POINTER(f), C.POINTER(f)]
    L.oracle_sample_material.restype = u32
    L.oracle_sample_material.argtypes = [C.POINTER(OMaterial), C.POINTER(f), C.POINTER(f), C.POINTER(C.c_double), u32, C.c_int,
                                         C.POINTER(f), C.POINTER(f)]
    L.oracle_radiance.argtypes = [C.POINTER(OMaterial), C.POINTER(f), C.POINTER(f), C.POINTER(f)]
    L.oracle_eye_ray.argtypes = [vp, C.POINTER(OSensor), u32, u32, C.POINTER(C.c_double), C.c_int, C.POINTER(f), C.POINTER(f),
                                 C.POINTER(f), C.POINTER(f), C.POINTER(u32)]
    L.oracle_mt_uniforms.argtypes = [u64, u32, vp, vp, vp]
    L.oracle_xorshift_seed.restype = u64
    L.oracle_xorshift_seed.argtypes = [u64, u32, u32]
    L.oracle_xorshift_uniforms.argtypes = [u64, u32, vp]
    L.oracle_sincos.argtypes = [f, C.c_int, C.POINTER(f), C.POINTER(f)]
    L.oracle_pow.restype = f
    L.oracle_pow.argtypes = [f, f, C.c_int]
    L.oracle_fnv1a64.restype = u64
    L.oracle_fnv1a64.argtypes = [vp, u64]
    L.oracle_tonemap.argtypes = [vp, u32, u32, vp]
    L.oracle_math_compare.restype = u64
    L.oracle_math_compare.argtypes = [C.c_int, C.c_int, C.c_int, u32, u32, f, C.POINTER(f)]
    L.oracle_pow_i.restype = C.c_double
    L.oracle_pow_i.argtypes = [f, C.c_int, C.c_int]
    _lib = L
    return L


def tonemap(img):
    """oracle_tonemap: Filmic -> Gamma -> 8-bit of an (H, W, 3) float32 image."""
    img = np.ascontiguousarray(img, np.float32)
    out = np.empty(img.shape, np.uint8)
    load().oracle_tonemap(img.ctypes.data, img.shape[1], img.shape[0], out.ctypes.data)
    return out


def sensor(width, height):
    """application.cc:89-94: Sensor(w, h, 0.036, 0.036 / w * h)."""
    return OSensor(width, height, np.float32(0.036), np.float32(0.036 / width * height))


def f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class Scene:
    def __init__(self, handle):
        self.h = C.c_void_p(handle)
        self.L = load()

    @classmethod
    def cornell(cls, accel=ACCEL_BVH, focal=0.050, radius=0.050, blades=6):
        return cls(load().oracle_scene_cornell_box(focal, radius, blades, accel))

    @classmethod
    def create(cls, objects, materials, transform, focal_length, focus_distance, radius, n_blades, accel=ACCEL_LIST):
        """objects: (kind, material, params) ; materials: (kind, (r,g,b), param) -- same tuples as amber_amd.HostScene.create."""
        objs = (OObject * max(1, len(objects)))()
        for i, (kind, mat, params) in enumerate(objects):
            objs[i].kind, objs[i].material = kind, mat
            for j, v in enumerate(list(params)[:9]):
                objs[i].p[j] = v
        mats = (OMaterial * max(1, len(materials)))()
        for i, (kind, rho, param) in enumerate(materials):
            mats[i].kind, mats[i].param = kind, param
            for j in range(3):
                mats[i].rho[j] = rho[j]
        lens = OThinLens((C.c_float * 16)(*[float(x) for x in transform]), focal_length, focus_distance, radius, n_blades)
        return cls(load().oracle_scene_create(objs, len(objects), mats, len(materials), C.byref(lens), accel))

    @classmethod
    def create_arrays(cls, kinds, material_index, params, materials, transform, focal_length, focus_distance, radius, n_blades, accel=ACCEL_LIST):
        """Bulk form of create() (numpy arrays as amber_amd.HostScene.create_arrays takes them): a million objects in a second."""
        n = len(kinds)
        dt = np.dtype([("kind", np.uint32), ("material", np.uint32), ("p", np.float32, (9,))])
        assert dt.itemsize == C.sizeof(OObject)
        arr = np.zeros(n, dt)
        arr["kind"], arr["material"] = kinds, material_index
        arr["p"] = np.asarray(params, np.float32)[:, :9]
        mats = (OMaterial * max(1, len(materials)))()
        for i, (kind, rho, param) in enumerate(materials):
            mats[i].kind, mats[i].param = kind, param
            for j in range(3):
                mats[i].rho[j] = rho[j]
        lens = OThinLens((C.c_float * 16)(*[float(x) for x in transform]), focal_length, focus_distance, radius, n_blades)
        return cls(load().oracle_scene_create(arr.ctypes.data_as(C.POINTER(OObject)), n, mats, len(materials), C.byref(lens), accel))

    def render_mt(self, w, h, seed, spp, math=MATH_LIBM):
        s = sensor(w, h)
        img = np.zeros((h, w, 3), np.float32)
        cnt = OCounters()
        self.L.oracle_render_mt(self.h, C.byref(s), seed, spp, math, img.ctypes.data, C.byref(cnt))
        return img, cnt

    def render_xorshift(self, w, h, seed, first, n, math=None, max_depth=0, threads=8, rows=None, out=None, chunk=ACCUM_CHUNK):
        s = sensor(w, h)
        img = np.zeros((h, w, 3), np.float32) if out is None else out
        cnt = OCounters()
        y0, y1 = rows if rows else (0, h)
        self.L.oracle_render_xorshift(self.h, C.byref(s), seed, first, n, y0, y1, _m(math), max_depth, threads, chunk, img.ctypes.data, C.byref(cnt))
        return img, cnt

    def path_signatures(self, w, h, seed, first, n, rows, math=None, max_depth=0, threads=8):
        """(y1 - y0, w, n) uint64 path signatures (oracle_path_signatures)."""
        s = sensor(w, h)
        y0, y1 = rows
        out = np.zeros((y1 - y0, w, n), np.uint64)
        self.L.oracle_path_signatures(self.h, C.byref(s), seed, first, n, y0, y1, _m(math), max_depth, threads, out.ctypes.data)
        return out

    def render_lt(self, w, h, seed, first, n, math=None, max_depth=0, max_records=1 << 16):
        """Light tracing: returns (sum image, counters, records (k,7) uint32 in accumulation order)."""
        s = sensor(w, h)
        img = np.zeros((h, w, 3), np.float32)
        cnt = OCounters()
        rec = np.zeros((max_records, 7), np.uint32)
        k = self.L.oracle_render_lt_xorshift(self.h, C.byref(s), seed, first, n, _m(math), max_depth, img.ctypes.data, C.byref(cnt), rec.ctypes.data, max_records)
        assert k <= max_records
        return img, cnt, rec[:k]

    def trace(self, w, h, seed, px, py, sample, math=None, max_depth=0, max_bounces=16):
        s = sensor(w, h)
        rec = (OBounce * max_bounces)()
        eye = (C.c_float * 7)()
        n = self.L.oracle_trace_path(self.h, C.byref(s), seed, px, py, sample, _m(math), max_depth, rec, max_bounces, eye)
        return n, rec, np.array(eye[:], np.float32)

    def set_accel(self, accel):
        """Switch the acceleration Scene::Cast uses (oracle_scene_set_accel); raises when the scene was not created for it."""
        if self.L.oracle_scene_set_accel(self.h, accel) != 0:
            raise ValueError(f"this oracle scene cannot serve acceleration {accel}")
        return self

    def cast_many(self, origins, dirs, accel, threads=8):
        """Closest hits of n rays through `accel`: (object index int32, -1 = miss; distance float32, NaN = miss)."""
        o, d = np.ascontiguousarray(origins, np.float32), np.ascontiguousarray(dirs, np.float32)
        n = len(o)
        idx, t = np.empty(n, np.int32), np.empty(n, np.float32)
        self.L.oracle_cast_many(self.h, accel, n, o.ctypes.data, d.ctypes.data, threads, idx.ctypes.data, t.ctypes.data)
        return idx, t

    def collect_rays(self, w, h, seed, first, n, rows, max_rays, math=None, max_depth=0):
        """The rays cast while rows x samples are rendered (every bounce, path order): (origins, dirs), at most max_rays."""
        s = sensor(w, h)
        o, d = np.zeros((max_rays, 3), np.float32), np.zeros((max_rays, 3), np.float32)
        k = self.L.oracle_collect_rays(self.h, C.byref(s), seed, first, n, rows[0], rows[1], _m(math), max_depth, max_rays, o.ctypes.data, d.ctypes.data)
        k = min(int(k), max_rays)
        return o[:k], d[:k]

    def classify_path(self, w, h, seed, px, py, sample, accel_a, accel_b, math=None, max_depth=0):
        """oracle_classify_path: None if the two accelerations agree on the whole path, else a dict describing the first cast they differ on."""
        s = sensor(w, h)
        out = (C.c_uint32 * 10)()
        if not self.L.oracle_classify_path(self.h, C.byref(s), seed, px, py, sample, _m(math), max_depth, accel_a, accel_b, out):
            return None
        f = lambda b: float(np.array([b], np.uint32).view(np.float32)[0])
        i = lambda v: -1 if v == 0xffffffff else int(v)
        return dict(cast=int(out[0]), object_a=i(out[1]), object_b=i(out[2]), object_list=i(out[3]), t_a=f(out[4]), t_b=f(out[5]), t_list=f(out[6]),
                    t_bits=(int(out[4]), int(out[5]), int(out[6])), list_object_box_hit=bool(out[7]), exact_tie=bool(out[8]), list_hit_inside_box=bool(out[9]))

    def cast(self, o, d):
        t = C.c_float()
        pos, nrm = (C.c_float * 3)(), (C.c_float * 3)()
        idx = self.L.oracle_cast(self.h, f3(o), f3(d), C.byref(t), pos, nrm)
        return idx, np.float32(t.value), np.array(pos[:], np.float32), np.array(nrm[:], np.float32)

    def objects(self):
        out = []
        for i in range(self.L.oracle_scene_object_count(self.h)):
            o, n = OObject(), (C.c_float * 3)()
            self.L.oracle_scene_get_object(self.h, i, C.byref(o), n)
            out.append((o.kind, o.material, np.array(o.p[:], np.float32), np.array(n[:], np.float32)))
        return out

    def materials(self):
        out = []
        for i in range(self.L.oracle_scene_material_count(self.h)):
            m, r0 = OMaterial(), C.c_float()
            self.L.oracle_scene_get_material(self.h, i, C.byref(m), C.byref(r0))
            out.append((m.kind, np.array(m.rho[:], np.float32), np.float32(m.param), np.float32(r0.value)))
        return out

    def lens(self):
        origin, g, l = (C.c_float * 3)(), (C.c_float * 9)(), (C.c_float * 9)()
        fd, sd, pa = C.c_float(), C.c_float(), C.c_float()
        self.L.oracle_scene_get_lens(self.h, origin, g, l, C.byref(fd), C.byref(sd), C.byref(pa))
        return (np.array(origin[:], np.float32), np.array(g[:], np.float32), np.array(l[:], np.float32),
                np.float32(fd.value), np.float32(sd.value), np.float32(pa.value))

    def bvh_stats(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self.L.oracle_scene_bvh_stats(self.h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def bvh_order(self):
        """objects_ after the reference's build: position -> insertion index"""
        out = np.zeros(self.L.oracle_scene_object_count(self.h), np.uint32)
        self.L.oracle_scene_bvh_order(self.h, out.ctypes.data)
        return out

    def bvh_digest(self):
        return int(self.L.oracle_scene_bvh_digest(self.h))

    def __del__(self):
        try:
            self.L.oracle_scene_destroy(self.h)
        except Exception:
            pass


def survey_hash(img: np.ndarray) -> int:
    """The hash SURVEY.md section 8(c) quotes: FNV-1a 64-bit multiply/xor over the raw f32 RGB bytes, started
    from the survey harness' offset basis 1469598103934665603 (the standard basis minus its last digit)."""
    h = 1469598103934665603
    data = np.ascontiguousarray(img, np.float32).tobytes()
    # vectorised per byte is not possible (sequential dependency); images here are <= 256x256
    for b in data:
        h ^= b
        h = (h * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h
