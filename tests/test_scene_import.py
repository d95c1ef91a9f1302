"""cli::ImportScene (import.cc:49-167) restated over an OBJ + MTL subset: host logic on the CPU, rendering on the GPU."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

import oracle_binding as O
import scene_files as SF

ROOT = Path(__file__).resolve().parent.parent


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _resolved(objs, mats):
    out = []
    for o in objs:
        m = mats[o.material]
        out.append((o.kind, tuple(bits(o.p[:])), m.kind, tuple(bits(m.rho[:])), int(bits([m.param])[0]), int(bits([m.r0])[0])))
    return out


def _oracle_scene(objects, materials, accel):
    return O.Scene.create(objects, materials, SF.TRANSFORM, accel=accel | O.BLADES_LAST, **SF.LENS)


def test_import_produces_the_reference_object_order_and_materials(amber, oracle, tmp_path):
    path, objects, materials = SF.write_scene(tmp_path)
    assert len(objects) == 1 + 8 + 2 + 2 + 320 + 3 + 3                     # fan triangulation of quads and the pentagon
    hs = amber.HostScene.import_file(path)
    objs, mats, lens = hs.flatten()
    osc = _oracle_scene(objects, materials, O.ACCEL_LIST)
    oobjs, omats = osc.objects(), osc.materials()
    assert len(objs) == len(oobjs) == len(objects) + 6
    assert lens.n_blades == 6 and lens.first_blade_object == len(objects)   # aperture objects LAST (import.cc:155-157)
    got = _resolved(objs, mats)
    for i, (g, (kind, mat, p, n)) in enumerate(zip(got, oobjs)):
        mk, rho, param, r0 = omats[mat]
        pp = np.zeros(12, np.float32)
        pp[:9], pp[9:] = p, n
        assert g == (kind, tuple(bits(pp)), mk, tuple(bits(rho)), int(bits([param])[0]), int(bits([r0])[0])), i
    kinds = [mats[o.material].kind for o in objs[:len(objects)]]
    assert kinds[0] == SF.LAMBERTIAN and SF.DIFFUSE_LIGHT in kinds and SF.SPECULAR in kinds and SF.PHONG in kinds
    assert all(mats[o.material].kind == 5 for o in objs[len(objects):])     # Eye
    # camera -> lens (import.cc:130-154): same bits as the oracle's MakeThinLens on the expected transform
    origin, g_, l_, fd, sd, pa = osc.lens()
    assert np.array_equal(bits(lens.origin[:]), bits(origin)) and np.array_equal(bits(lens.global_[:]), bits(g_))
    assert np.array_equal(bits(lens.local_[:]), bits(l_))
    assert bits([lens.focus_distance, lens.sensor_distance, lens.p_area]).tolist() == bits([fd, sd, pa]).tolist()
    assert tuple(lens.origin[:]) == SF.CAMERA[:3]


def test_import_error_behaviour(amber, tmp_path):
    from amber_amd.api import AmberError
    with pytest.raises(AmberError, match="Unable to open file"):
        amber.HostScene.import_file(tmp_path / "missing.obj")
    sub = tmp_path / "nocam"; sub.mkdir()
    path, _, _ = SF.write_scene(sub, subdivisions=0, camera=False)
    with pytest.raises(AmberError, match="scene file has no cameras"):     # import.cc:132-134
        amber.HostScene.import_file(path)
    bad = tmp_path / "bad.obj"
    bad.write_text("#camera 0 0 4 0 0 -1 0 1 0\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4\n")
    with pytest.raises(AmberError, match=r"bad\.obj:5: face references vertex 4 of 3"):
        amber.HostScene.import_file(bad)
    bad.write_text("#camera 0 0 4 0 0 -1 0 1 0\nv 0 0\n")
    with pytest.raises(AmberError, match="vertex needs three coordinates"):
        amber.HostScene.import_file(bad)
    # a missing material library is not an error (assimp's OBJ loader goes on with the default material)
    ok = tmp_path / "ok.obj"
    ok.write_text("mtllib nowhere.mtl\n#camera 0 0 4 0 0 -1 0 1 0\nv 0 0 0\r\nv 1 0 0\nv 0 1 0\nusemtl x\nf 1 2 3\n")
    objs, mats, _ = amber.HostScene.import_file(ok).flatten()
    assert len(objs) == 7 and mats[objs[0].material].kind == SF.LAMBERTIAN and tuple(mats[objs[0].material].rho[:]) == (0.5, 0.5, 0.5)


@pytest.mark.gpu
def test_imported_mesh_renders_like_the_oracle(amber, oracle, tmp_path):
    """A 339-triangle imported scene (BVH engine chosen automatically, application.cc:80) == oracle, == engine LIST."""
    path, objects, materials = SF.write_scene(tmp_path)
    hs = amber.HostScene.import_file(path)
    osc = _oracle_scene(objects, materials, O.ACCEL_BVH)
    W, H, spp = 96, 64, 24
    sn = amber.Sensor.default(W, H)
    ref, cnt = osc.render_xorshift(W, H, 31, 0, spp)
    for engine in (amber.ENGINE_AUTO, amber.ENGINE_LIST, amber.ENGINE_WAVEFRONT):
        pt = amber.PathTracer(hs, sn, seed=31, engine=engine)
        pt.render_pass(0, spp)
        img, rays = pt.download()
        pt.close()
        assert rays == cnt.casts, engine
        assert np.array_equal(bits(img), bits(ref)), engine
    assert (ref > 0).mean() > 0.2 and cnt.casts > 2 * W * H * spp          # lit, multi-bounce
    # per-bounce traces through the mirror ball
    pt = amber.PathTracer(hs, sn, seed=31)
    px = np.arange(0, W * H, 5, dtype=np.uint32); sm = (px % spp).astype(np.uint32)
    from test_gpu_parity import _compare_traces
    casts = _compare_traces(pt, osc, W, H, 31, px[::37], sm[::37])
    assert casts.max() >= 5


@pytest.mark.gpu
def test_cli_scene_option(amber, tmp_path):
    """`amber --scene FILE` (option.cc:77-80, application.cc:74-87) renders the imported scene; a bad file exits -1."""
    path, _, _ = SF.write_scene(tmp_path, subdivisions=1)
    exe = ROOT / "amber_amd" / "bin" / "amber"
    out = tmp_path / "img"
    r = subprocess.run([str(exe), "--scene", str(path), "--spp", "8", "--width", "64", "--height", "48", "--seed", "3", "--output", str(out)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "Loading scene ... done." in r.stderr
    assert (tmp_path / "img.png").stat().st_size > 1000 and (tmp_path / "img.exr").stat().st_size > 64 * 48 * 12
    img, _ = amber.HostScene.import_file(path).render(amber.Sensor.default(64, 48), 8, seed=3)
    from test_output_stage import parse_exr
    got = parse_exr(str(tmp_path / "img.exr"))[:, ::-1]                      # x-mirrored on export (cli/image.cc)
    assert np.array_equal(bits(got), bits(img))
    r = subprocess.run([str(exe), "--scene", str(tmp_path / "none.obj"), "--spp", "1"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "Unable to open file" in r.stderr
