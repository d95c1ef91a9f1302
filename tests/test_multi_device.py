"""Multi-GPU inside ONE Algorithm::Render call (rendering.h HipPathTracingOptions.devices) -- the
counterpart of the reference's thread fan-out inside Render (prelude/parallel.cc:29-40, rendering/parallel.h:57-68,
cli/application.cc:129).  The pool has one GPU per box, so N engine handles are put on device 0: the code path is the
multi-device one (N handles, concurrent creation, interleaved stripes, per-handle download, re-assembly), and the image
must be bit-identical to the single-handle render."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

from test_output_stage import parse_exr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("width,height,n", [(96, 64, 2), (80, 50, 3), (64, 12, 3), (40, 7, 4)])   # (64, 12, 3): rank 2 has no stripe; 7 rows: one partial stripe
def test_devices_option_is_bit_identical_to_one_device(amber, width, height, n):
    hs = amber.HostScene.cornell_box()
    sensor = amber.Sensor.default(width, height)
    one, st1 = hs.render(sensor, 48, seed=21, samples_per_launch=16)
    many, stn = hs.render(sensor, 48, seed=21, samples_per_launch=16, devices=[0] * n)
    assert np.array_equal(one.view(np.uint32), many.view(np.uint32))
    assert st1["rays"] == stn["rays"] and stn["passes"] == 48


def test_bad_device_ordinal_is_loud(amber):
    hs = amber.HostScene.cornell_box()
    with pytest.raises(amber.AmberError, match="device"):
        hs.render(amber.Sensor.default(32, 32), 4, devices=[0, 99])


def test_cli_device_list(amber, tmp_path):
    exe = Path(amber.library_path()).parent.parent / "bin" / "amber"
    a, b = str(tmp_path / "one"), str(tmp_path / "three")
    common = ["--algorithm", "pt", "--width", "72", "--height", "48", "--spp", "32", "--seed", "5", "--samples-per-launch", "16"]
    r1 = subprocess.run([str(exe), *common, "--output", a], capture_output=True, text=True, timeout=300)
    r3 = subprocess.run([str(exe), *common, "--device-list", "0,0,0", "--output", b], capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0 and r3.returncode == 0, r1.stderr + r3.stderr
    assert np.array_equal(parse_exr(a + ".exr").view(np.uint32), parse_exr(b + ".exr").view(np.uint32))
    assert (Path(a + ".png").read_bytes() == Path(b + ".png").read_bytes())
