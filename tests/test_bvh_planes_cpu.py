"""CPU test of the host side of engine BVH's node format (amber_amd/csrc/hip/bvh_build.h): the binary16 plane values of a box are
conservative, tight and never denormal, at every scale and offset ("binary16 planes")."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_binary16_planes_are_conservative_and_tight(tmp_path):
    exe = tmp_path / "plane_word_check"
    src = ROOT / "tests" / "cpp" / "plane_word_check.cpp"
    r = subprocess.run(["hipcc", "-x", "hip", "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off", str(src), "-o", str(exe)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "0 violations, 0 planes not tight" in r.stdout, r.stdout + r.stderr
