"""Output stage: Filmic + Gamma tone mapping and the x-mirrored PNG / EXR files
(postprocess/filmic.cc:30-67, gamma.cc:36-52, cli/image.cc:45-71).  Host-only: no GPU needed.
Byte work: the checker is the oracle's C restatement with the host's powf (oracle_tonemap) and every byte must be EQUAL."""
import oracle_binding as O
import struct
import zlib

import numpy as np
import pytest


def parse_png(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, ihdr = 8, b"", None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + body)
        if typ == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    w, h, depth, ctype = ihdr[:4]
    assert (depth, ctype) == (8, 2)
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert not raw[:, 0].any()                                  # filter type 0 on every scanline
    return raw[:, 1:].reshape(h, w, 3)


def parse_exr(path):
    data = open(path, "rb").read()
    assert struct.unpack("<II", data[:8]) == (20000630, 2)
    pos, attrs = 8, {}
    while data[pos] != 0:
        e = data.index(b"\0", pos); name = data[pos:e].decode(); pos = e + 1
        e = data.index(b"\0", pos); typ = data[pos:e].decode(); pos = e + 1
        size = struct.unpack("<I", data[pos:pos + 4])[0]; pos += 4
        attrs[name] = (typ, data[pos:pos + size]); pos += size
    pos += 1
    x0, y0, x1, y1 = struct.unpack("<iiii", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    assert attrs["compression"][1] == b"\0" and attrs["channels"][1].count(b"\0") > 3
    offsets = struct.unpack("<%dQ" % h, data[pos:pos + 8 * h])
    img = np.zeros((h, w, 3), np.float32)
    for y, off in enumerate(offsets):
        yy, nbytes = struct.unpack("<ii", data[off:off + 8])
        assert yy == y and nbytes == 12 * w
        line = np.frombuffer(data[off + 8:off + 8 + nbytes], np.float32).reshape(3, w)      # B, G, R planes
        img[y, :, 2], img[y, :, 1], img[y, :, 0] = line[0], line[1], line[2]
    return img


def test_tonemap_matches_the_reference_formulas(amber):
    rng = np.random.default_rng(4)
    img = (rng.random((37, 53, 3)) ** 4 * 0.3).astype(np.float32)
    img[0, 0] = 0.0; img[0, 1] = 1e11; img[0, 2] = [1e-9, 0.0437, 0.7 / 16]
    img[0, 3] = [np.nan, -1.0, np.inf]; img[0, 4] = [-0.0, 1e-30, 3e38]
    got = amber.tonemap(img)
    ref = O.tonemap(img)
    assert np.array_equal(got, ref)                                  # every byte
    assert tuple(got[0, 0]) == (0, 0, 0) and tuple(got[0, 1]) == (255, 255, 255)
    big = (np.random.default_rng(5).random((256, 512, 3)) ** 6).astype(np.float32)       # 393 216 values across the whole curve
    assert np.array_equal(amber.tonemap(big), O.tonemap(big))


def test_png_and_exr_files_are_x_mirrored(amber, tmp_path):
    rng = np.random.default_rng(8)
    img = (rng.random((21, 34, 3)) * 0.05).astype(np.float32)
    png, exr = str(tmp_path / "o.png"), str(tmp_path / "o.exr")
    amber.export(img, png, exr)
    assert np.array_equal(parse_png(png), O.tonemap(img)[:, ::-1])              # cli/image.cc:66 column W-1-i; bytes == the oracle's
    assert np.array_equal(parse_exr(exr).view(np.uint32), img[:, ::-1].view(np.uint32))   # raw floats, bit for bit


@pytest.mark.gpu
def test_command_line_driver(amber, tmp_path):
    """bin/amber --algorithm pt: progress line, statistics, output.png / output.exr equal to the library's render."""
    import subprocess
    from pathlib import Path
    exe = Path(amber.library_path()).parent.parent / "bin" / "amber"
    out = str(tmp_path / "cli")
    r = subprocess.run([str(exe), "--algorithm", "pt", "--width", "72", "--height", "48", "--spp", "40", "--seed", "9",
                        "--samples-per-launch", "16", "--output", out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "40/40" in r.stderr and "iterations/second" in r.stderr and "Exporting" in r.stderr
    image, st = amber.HostScene.cornell_box().render(amber.Sensor.default(72, 48), 40, seed=9, samples_per_launch=16)
    assert st["passes"] == 40
    assert np.array_equal(parse_exr(out + ".exr").view(np.uint32), image[:, ::-1].view(np.uint32))
    assert np.array_equal(parse_png(out + ".png"), O.tonemap(image)[:, ::-1])
    bad = subprocess.run([str(exe), "--algorithm", "bdpt"], capture_output=True, text=True, timeout=60)
    assert bad.returncode != 0 and "Unknown algorithm" in bad.stderr              # application.cc:60-65
    # --time expiry (Context::Expire): --spp 0 runs until the limit and still writes a valid mean image
    r = subprocess.run([str(exe), "--width", "64", "--height", "64", "--spp", "0", "--time", "1", "--output", out + "t"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and np.isfinite(parse_exr(out + "t.exr")).all()
