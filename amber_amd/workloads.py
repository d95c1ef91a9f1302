"""Procedural triangle-mesh workloads for engine BVH, written as Wavefront OBJ + MTL so that they enter through cli::ImportScene
(amber/import.cc; reference: /root/reference/src/amber/cli/import.cc:49-167, application.cc:74-87) -- the path a user of `--scene` takes.

Every generator returns a `MeshWorkload`: the OBJ / MTL text plus, built here INDEPENDENTLY of the importer, the arrays the import must
produce (object order of import.cc:109-157: meshes = (object block, material) pairs in first-appearance order, faces in file order,
aperture blades last), in the form HostScene.create_arrays / oracle_binding.Scene.create_arrays take.  Host-side data only.

  room_mesh(subdivisions)   a Cornell-like room (walls, lamp, a mirror box) holding a bumpy icosphere of 20 * 4**subdivisions glossy
                            triangles: subdivisions = 3 -> 1 280 + 24 triangles, "a typical imported scene".
  terrain_mesh(patches, k)  a displaced, tessellated terrain inside the same room: patches x patches tiles, tessellated k x k and
                            k/2 x k/2 in a checkerboard, the tiles a hair apart and stitched with NEEDLE triangles (one edge a
                            coarse cell long, 1e-5 wide; aspect ratio ~ 1 000): patches = 16, k = 56 -> 1.0 M triangles.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

LAMBERTIAN, PHONG, SPECULAR, REFRACTION, DIFFUSE_LIGHT = 0, 1, 2, 3, 4
LENS = dict(focal_length=0.050, focus_distance=4.0, radius=0.010, n_blades=6)      # import.cc:148-154


def camera_transform(cam):
    """import.cc:136-146 in binary32: zaxis = -lookAt, xaxis = lookAt ^ up, yaxis = zaxis ^ xaxis, position in column 4."""
    f = np.float32
    pos, look, up = [f(x) for x in cam[0:3]], [f(x) for x in cam[3:6]], [f(x) for x in cam[6:9]]

    def cross(u, v):
        return [f(f(u[1] * v[2]) - f(u[2] * v[1])), f(f(u[2] * v[0]) - f(u[0] * v[2])), f(f(u[0] * v[1]) - f(u[1] * v[0]))]
    z = [-c for c in look]
    x = cross(look, up)
    y = cross(z, x)
    return [x[0], y[0], z[0], pos[0], x[1], y[1], z[1], pos[1], x[2], y[2], z[2], pos[2], f(0), f(0), f(0), f(1)]


@dataclass
class MeshWorkload:
    name: str
    camera: tuple                                  # position, lookAt direction, up (9 numbers): the `#camera` line
    mtl: str                                       # text of the material library
    materials: list                                # (kind, (r, g, b), param) in library order
    vertices: np.ndarray = None                    # (nv, 3) float32
    groups: list = field(default_factory=list)     # [(object name, material index, faces (m, 3) int64 0-based)], file order

    @property
    def n_triangles(self) -> int:
        return int(sum(len(f) for _, _, f in self.groups))

    def write(self, directory) -> Path:
        """Writes <name>.obj and <name>.mtl into `directory`; returns the OBJ path."""
        directory = Path(directory)
        directory.mkdir(parents=True, exist_ok=True)
        (directory / f"{self.name}.mtl").write_text(self.mtl)
        names = [ln.split()[1] for ln in self.mtl.splitlines() if ln.startswith("newmtl")]
        path = directory / f"{self.name}.obj"
        with open(path, "w") as f:
            f.write(f"# {self.name}: {self.n_triangles} triangles (amber_amd/workloads.py)\nmtllib {self.name}.mtl\n")
            f.write("#camera " + " ".join("%.9g" % x for x in self.camera) + "\n")
            v = self.vertices
            f.write("\n".join("v %.9g %.9g %.9g" % (a, b, c) for a, b, c in v.tolist()))
            f.write("\n")
            for obj, mat, faces in self.groups:
                f.write(f"o {obj}\nusemtl {names[mat]}\n")
                f.write("\n".join("f %d %d %d" % (a, b, c) for a, b, c in (faces + 1).tolist()))
                f.write("\n")
        return path

    def arrays(self):
        """What cli::ImportScene must make of the file: keyword arguments of HostScene.create_arrays / oracle Scene.create_arrays
        (the oracle needs accel | BLADES_LAST: aperture objects after the meshes, import.cc:155-157)."""
        # meshes = (block, material) in first-appearance order; here every `o` block has one material, so file order
        tri = np.concatenate([self.vertices[f].reshape(len(f), 9) for _, _, f in self.groups]).astype(np.float32)
        material = np.concatenate([np.full(len(f), m, np.uint32) for _, m, f in self.groups])
        return dict(kinds=np.zeros(len(tri), np.uint32), material_index=material, params=tri, materials=self.materials,
                    transform=camera_transform(self.camera), **LENS)


# ---- shared pieces -----------------------------------------------------------------------------------------------------------
_ROOM_MTL = """newmtl white
Kd 0.75 0.75 0.75
newmtl red
Kd 0.75 0.25 0.25
newmtl green
Kd 0.25 0.75 0.25
newmtl lamp
Kd 0 0 0
Ke 18 17 15
newmtl mirror
Kd 0 0 0
Kr 0.9 0.9 0.9
Pr 1
newmtl glossy
Kd 0.2 0.2 0.2
Ks 0.8 0.75 0.6
Ns 40
illum 2
newmtl blue
Kd 0.3 0.4 0.75
"""
_ROOM_MATERIALS = [(LAMBERTIAN, (0.75, 0.75, 0.75), 0.0), (LAMBERTIAN, (0.75, 0.25, 0.25), 0.0), (LAMBERTIAN, (0.25, 0.75, 0.25), 0.0),
                   (DIFFUSE_LIGHT, (18.0, 17.0, 15.0), 0.0), (SPECULAR, (0.9, 0.9, 0.9), 0.0), (PHONG, (0.8, 0.75, 0.6), 40.0),
                   (LAMBERTIAN, (0.3, 0.4, 0.75), 0.0)]
WHITE, RED, GREEN, LAMP, MIRROR, GLOSSY, BLUE = range(7)


class _Builder:
    def __init__(self):
        self.v, self.groups = [], []

    def add_vertices(self, pts) -> int:
        base = sum(len(a) for a in self.v)
        self.v.append(np.asarray(pts, np.float32).reshape(-1, 3))
        return base

    def add_group(self, name, material, faces):
        self.groups.append((name, material, np.asarray(faces, np.int64).reshape(-1, 3)))

    def quad(self, name, material, p):
        b = self.add_vertices(p)
        self.add_group(name, material, [(b, b + 1, b + 2), (b + 2, b + 3, b)])

    def vertices(self):
        return np.concatenate(self.v)


def _room(b: _Builder):
    c = [(-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1)]
    b.quad("back", WHITE, [c[0], c[1], c[2], c[3]])
    b.quad("floor", WHITE, [c[0], c[4], c[5], c[1]])
    b.quad("ceiling", WHITE, [c[3], c[2], c[6], c[7]])
    b.quad("left", RED, [c[0], c[3], c[7], c[4]])
    b.quad("right", GREEN, [c[1], c[5], c[6], c[2]])
    b.quad("lamp", LAMP, [(-0.3, 0.99, -0.3), (0.3, 0.99, -0.3), (0.3, 0.99, 0.3), (-0.3, 0.99, 0.3)])


def _box(b: _Builder, name, material, lo, hi, turn=0.0):
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    ctr = 0.5 * (lo + hi)
    cs, sn = np.cos(turn), np.sin(turn)
    pts = []
    for k in range(8):
        p = np.array([hi[0] if k & 1 else lo[0], hi[1] if k & 2 else lo[1], hi[2] if k & 4 else lo[2]]) - ctr
        pts.append(ctr + np.array([cs * p[0] + sn * p[2], p[1], -sn * p[0] + cs * p[2]]))
    base = b.add_vertices(pts)
    quads = [(0, 2, 3, 1), (4, 5, 7, 6), (0, 1, 5, 4), (2, 6, 7, 3), (0, 4, 6, 2), (1, 3, 7, 5)]
    b.add_group(name, material, [(base + q[i], base + q[j], base + q[k]) for q in quads for i, j, k in ((0, 1, 2), (2, 3, 0))])


def _icosphere(subdivisions):
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)], np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
                  (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)], np.int64)
    for _ in range(subdivisions):
        e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
        ue, inv = np.unique(e, axis=0, return_inverse=True)
        mid = v[ue[:, 0]] + v[ue[:, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        m = len(v) + np.asarray(inv).reshape(3, -1)
        v = np.concatenate([v, mid])
        a, b, c = f[:, 0], f[:, 1], f[:, 2]
        f = np.concatenate([np.stack([a, m[0], m[2]], 1), np.stack([b, m[1], m[0]], 1), np.stack([c, m[2], m[1]], 1), np.stack([m[0], m[1], m[2]], 1)])
    return v, f


def room_mesh(subdivisions: int = 3, seed: int = 3) -> MeshWorkload:
    """Cornell-like room + mirror box + bumpy glossy icosphere (20 * 4**subdivisions triangles)."""
    b = _Builder()
    _room(b)
    _box(b, "mirror_box", MIRROR, (-0.75, -1.0, -0.65), (-0.2, 0.15, -0.1), turn=0.35)
    v, f = _icosphere(subdivisions)
    rng = np.random.Generator(np.random.MT19937(seed))
    bump = 1.0 + 0.03 * np.sin(9.0 * v[:, 0]) * np.sin(7.0 * v[:, 1] + 1.0) * np.sin(8.0 * v[:, 2] + 2.0) + 0.003 * rng.standard_normal(len(v))
    base = b.add_vertices(np.array([0.35, -0.55, 0.2]) + 0.42 * v * bump[:, None])
    b.add_group("ball", GLOSSY, f + base)
    return MeshWorkload(f"room_mesh_{subdivisions}", (0.0, 0.0, 4.0, 0.0, 0.0, -1.0, 0.0, 1.0, 0.0), _ROOM_MTL, list(_ROOM_MATERIALS), b.vertices(), b.groups)


def _height(x, z):
    return (-0.62 + 0.16 * np.sin(2.7 * x + 0.4) * np.cos(2.2 * z - 0.3) + 0.05 * np.sin(9.0 * x + 1.0) * np.sin(11.0 * z)
            + 0.012 * np.sin(37.0 * x) * np.cos(41.0 * z + 0.5))


def terrain_mesh(patches: int = 16, k: int = 56, gap: float = 1e-5) -> MeshWorkload:
    """Displaced terrain of patches x patches tiles over [-0.98, 0.98]^2, tile (i, j) tessellated k x k if (i + j) is even, else
    k/2 x k/2; neighbouring tiles are `gap` apart and the slit is closed by a strip of needle triangles."""
    assert k % 2 == 0
    b = _Builder()
    _room(b)
    size = 1.96 / patches
    side = size - gap
    tile_mat = [WHITE, BLUE, GLOSSY, WHITE, GREEN, WHITE, MIRROR, WHITE]
    edges = {}                                                                 # (i, j) -> vertex index arrays of the four borders
    for j in range(patches):
        for i in range(patches):
            n = k if (i + j) % 2 == 0 else k // 2
            x0, z0 = -0.98 + i * size + 0.5 * gap, -0.98 + j * size + 0.5 * gap
            xs, zs = x0 + side * np.arange(n + 1) / n, z0 + side * np.arange(n + 1) / n
            X, Z = np.meshgrid(xs, zs, indexing="xy")                          # row r = z index, column c = x index
            base = b.add_vertices(np.stack([X, _height(X, Z), Z], -1))
            idx = base + np.arange((n + 1) * (n + 1)).reshape(n + 1, n + 1)
            a, bb, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel(), idx[1:, :-1].ravel()
            b.add_group(f"tile_{i}_{j}", tile_mat[(3 * i + 5 * j) % len(tile_mat)], np.concatenate([np.stack([a, d, c], 1), np.stack([c, bb, a], 1)]))
            edges[(i, j)] = dict(west=idx[:, 0], east=idx[:, -1], south=idx[0, :], north=idx[-1, :])
    seams = []

    def stitch(e0, e1):
        """Strip between two border vertex rows (same direction), one of which has twice the cells of the other."""
        if len(e0) == len(e1):
            a, bb, c, d = e0[:-1], e0[1:], e1[1:], e1[:-1]
            return np.concatenate([np.stack([a, bb, c], 1), np.stack([c, d, a], 1)])
        flip = len(e0) < len(e1)
        fine, coarse = (e1, e0) if flip else (e0, e1)
        c0, c1 = coarse[:-1], coarse[1:]
        f0, f1, f2 = fine[0:-2:2], fine[1:-1:2], fine[2::2]
        t = np.concatenate([np.stack([c0, f1, f0], 1), np.stack([c0, c1, f1], 1), np.stack([c1, f2, f1], 1)])
        return t[:, ::-1] if flip else t
    for j in range(patches):
        for i in range(patches):
            if i + 1 < patches:
                seams.append(stitch(edges[(i, j)]["east"], edges[(i + 1, j)]["west"]))
            if j + 1 < patches:
                seams.append(stitch(edges[(i, j + 1)]["south"], edges[(i, j)]["north"]))
    b.add_group("seams", WHITE, np.concatenate(seams))
    _box(b, "pillar", MIRROR, (0.45, -0.9, -0.6), (0.7, 0.3, -0.35), turn=0.5)
    cam = (0.0, 0.63, 3.3, 0.0, -0.28, -0.96, 0.0, 0.96, -0.28)        # pitched down 16 degrees: the terrain fills the 16:9 frame from its front edge to the back wall
    return MeshWorkload(f"terrain_{patches}x{k}", cam, _ROOM_MTL, list(_ROOM_MATERIALS), b.vertices(), b.groups)
