"""Writes the OBJ + MTL test scene for cli::ImportScene and returns what the import must produce.

The expectation is built here independently of the importer: the (kind, material, params) / (kind, rho, param)
tuples of amber_amd.HostScene.create / oracle_binding.Scene.create, in the object order import.cc:109-157 defines
(meshes = (object block, material) in first-appearance order, polygons fanned from their first corner, aperture
blades last).
"""
import numpy as np

LAMBERTIAN, PHONG, SPECULAR, REFRACTION, DIFFUSE_LIGHT = 0, 1, 2, 3, 4
CAMERA = (0.0, 0.0, 4.0, 0.0, 0.0, -1.0, 0.0, 1.0, 0.0)        # position, lookAt direction, up


def camera_transform(cam):
    """import.cc:136-146 in binary32: zaxis = -lookAt, xaxis = lookAt ^ up, yaxis = zaxis ^ xaxis, position in column 4
    (signed zeros included: they reach the lens matrices)."""
    f = np.float32
    pos, look, up = [f(x) for x in cam[0:3]], [f(x) for x in cam[3:6]], [f(x) for x in cam[6:9]]

    def cross(u, v):
        return [f(u[1] * v[2]) - f(u[2] * v[1]), f(u[2] * v[0]) - f(u[0] * v[2]), f(u[0] * v[1]) - f(u[1] * v[0])]
    z = [-c for c in look]
    x = cross(look, up)
    y = cross(z, x)
    return [x[0], y[0], z[0], pos[0], x[1], y[1], z[1], pos[1], x[2], y[2], z[2], pos[2], f(0), f(0), f(0), f(1)]


TRANSFORM = camera_transform(CAMERA)
LENS = dict(focal_length=0.050, focus_distance=4.0, radius=0.010, n_blades=6)   # import.cc:148-154


def icosphere(center, radius, subdivisions):
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    v = [np.array(p, np.float64) / np.linalg.norm(p) for p in v]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(subdivisions):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m)); cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    c = np.array(center, np.float64)
    return [tuple(np.float32(c + radius * p)) for p in v], f


def write_scene(directory, subdivisions=2, camera=True):
    """Returns (obj_path, expected_objects, expected_materials)."""
    mtl = """# test materials
newmtl white
Kd 0.75 0.75 0.75
illum 1
newmtl red
Kd 0.75 0.25 0.25
Ks 0.5 0.5 0.5
newmtl lamp
Kd 0.1 0.1 0.1
Ke 12 11 9
newmtl mirror
Kd 0.2 0.2 0.2
Kr 0.9 0.95 1.0
Pr 0.5
newmtl shiny
Kd 0.3 0.3 0.3
Ks 0.6 0.7 0.8
Ns 24
illum 2
newmtl dark_mirror
Kr 0.9 0.9 0.9
Pr 0
Kd 0.25 0.5 0.25
newmtl unlit_lamp
Ke 0 0 0
newmtl phong_without_ns
Ks 0.5 0.5 0.5
illum 2
"""
    materials = [(LAMBERTIAN, (0.75, 0.75, 0.75), 0.0), (LAMBERTIAN, (0.75, 0.25, 0.25), 0.0), (DIFFUSE_LIGHT, (12.0, 11.0, 9.0), 0.0),
                 (SPECULAR, (0.9 * 0.5, 0.95 * 0.5, 1.0 * 0.5), 0.0), (PHONG, (0.6, 0.7, 0.8), 24.0), (LAMBERTIAN, (0.25, 0.5, 0.25), 0.0),
                 (LAMBERTIAN, (0.5, 0.5, 0.5), 0.0), (LAMBERTIAN, (0.5, 0.5, 0.5), 0.0),
                 (LAMBERTIAN, (0.5, 0.5, 0.5), 0.0)]                                        # [8]: faces without usemtl
    mat_id = {"white": 0, "red": 1, "lamp": 2, "mirror": 3, "shiny": 4, "dark_mirror": 5, "unlit_lamp": 6, "phong_without_ns": 7, None: 8}
    # float32(0.9f * 0.5f) etc.: the importer multiplies colour by reflectivity in binary32 (aiColor4D * float)
    materials[3] = (SPECULAR, tuple(float(np.float32(a) * np.float32(0.5)) for a in (0.9, 0.95, 1.0)), 0.0)

    verts, lines, meshes, order = [], [], {}, []
    block, material = [0], [None]

    def v(p):
        verts.append(tuple(np.float32(x) for x in p)); lines.append("v %.9g %.9g %.9g" % verts[-1]); return len(verts)

    def face(idx, fmt="%d"):
        """idx: 1-based absolute indices; fmt renders one corner."""
        lines.append("f " + " ".join(fmt % ((i,) * fmt.count("%d")) for i in idx))
        key = (block[0], material[0])
        if key not in meshes:
            meshes[key] = []; order.append(key)
        for k in range(1, len(idx) - 1):
            meshes[key].append((verts[idx[0] - 1], verts[idx[k] - 1], verts[idx[k + 1] - 1]))

    def usemtl(name):
        material[0] = name; lines.append("usemtl " + (name or "no_such_material"))

    lines += ["# scene for tests/test_scene_import.py", "mtllib scene.mtl"]
    if camera:
        lines.append("#camera " + " ".join("%g" % x for x in CAMERA))
        lines.append("#camera 9 9 9 1 0 0 0 1 0")           # a second camera is ignored (import.cc:135 takes mCameras[0])
    a = v((0.9, -0.99, 0.9)); b = v((0.9, -0.99, 0.5)); c = v((0.5, -0.99, 0.9))
    face([a, b, c])                                                                        # before any usemtl: default material
    lines.append("o room")
    block[0] += 1
    c8 = [v(p) for p in ((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1))]
    usemtl("white")
    face([c8[0], c8[1], c8[2], c8[3]])                                                      # back   (z = -1), a quad
    face([c8[0], c8[4], c8[5], c8[1]], "%d/%d")                                             # floor  (v/vt)
    usemtl("red")
    face([c8[0], c8[3], c8[7], c8[4]], "%d//%d")                                            # left   (v//vn)
    usemtl("white")                                                                         # back to white: SAME mesh as before
    face([c8[3], c8[2], c8[6], c8[7]], "%d/%d/%d")                                          # ceiling
    usemtl("dark_mirror")
    face([c8[1], c8[5], c8[6], c8[2]])                                                      # right
    lines.append("g lamp")
    block[0] += 1
    usemtl("lamp")
    for p in ((-0.3, 0.98, -0.3), (0.3, 0.98, -0.3), (0.3, 0.98, 0.3), (-0.3, 0.98, 0.3)):
        v(p)
    lines.append("f -4 -3 -2 -1")                                                           # negative (relative) indices
    n = len(verts)
    meshes[(block[0], "lamp")] = [(verts[n - 4], verts[n - 3], verts[n - 2]), (verts[n - 4], verts[n - 2], verts[n - 1])]
    order.append((block[0], "lamp"))
    lines.append("o ball")
    block[0] += 1
    usemtl("mirror")
    sv, sf = icosphere((0.35, -0.55, -0.2), 0.4, subdivisions)
    base = len(verts)
    for p in sv:
        v(p)
    for (i, j, k) in sf:
        face([base + i + 1, base + j + 1, base + k + 1])
    lines.append("o plate")
    block[0] += 1
    usemtl("shiny")
    pent = [v((-0.5 + 0.3 * np.cos(2 * np.pi * k / 5), -0.6, 0.2 + 0.3 * np.sin(2 * np.pi * k / 5))) for k in range(5)]
    face(pent[::-1])                                                                        # pentagon: fan of 3 triangles
    usemtl("unlit_lamp")
    face([pent[0], pent[1], c8[4]])
    usemtl("phong_without_ns")
    face([pent[2], pent[3], c8[4]])
    usemtl(None)                                                                            # unknown name: default material
    face([pent[3], pent[4], c8[4]])
    lines += ["l 1 2", "p 1", "vt 0 0", "vn 0 1 0", "s off"]                                # ignored records

    (directory / "scene.mtl").write_text(mtl)
    path = directory / "scene.obj"
    path.write_text("\n".join(lines) + "\n")
    objects = []
    for key in order:
        for tri in meshes[key]:
            objects.append((0, mat_id[key[1]], [float(x) for p in tri for x in p]))
    return path, objects, materials
