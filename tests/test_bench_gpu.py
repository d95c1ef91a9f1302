"""bench.py end to end on the GPU box: it starts its own ranks, the gathered image is the rendered image.

ADVICE round 1 (high): the gather used to run on torch's default stream while the render ran on the handle's private
non-blocking stream -- unordered.  The gather is now enqueued on the handle's stream (amber_hip_pt_stream); these tests
compare the image rank 0 gathers with a plain single-handle download, bit for bit, for the RCCL path (one rank,
AMBER_BENCH_FORCE_COLLECTIVE=1 -- the pool has one GPU per box) and for the multi-rank flow (ranks share GPU 0, gloo).
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _bench(tmp_path, *flags, env=None):
    out = tmp_path / "img.npy"
    e = dict(os.environ, AMBER_BENCH_SAVE_IMAGE=str(out), HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update(env or {})
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", *flags],
                       capture_output=True, text=True, env=e, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout                      # the contract: ONE JSON line on stdout
    return json.loads(lines[0]), np.load(out)


def _reference_image(amber, width, spp, seed=12345):
    pt = amber.PathTracer(amber.HostScene.cornell_box(), amber.Sensor.default(width, width), seed=seed)
    pt.render_pass(0, spp)
    img, rays = pt.download()
    return img, rays


def test_plain_gpus_flag_starts_its_own_ranks(amber, tmp_path):
    """`python bench.py --gpus 3 --rehearse-on-one-gpu` with no launcher around it: one JSON line, n_gpus 3, exact image."""
    line, img = _bench(tmp_path, "--gpus", "3", "--rehearse-on-one-gpu", "--width", "256", "--spp", "64")
    assert line["n_gpus"] == 3 and line["ranks"] == 3 and line["config"]["parallelism"] == "stripes3x8rows"
    ref, rays = _reference_image(amber, 256, 64)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    assert line["config"]["rays_per_step"] == rays
    # what makes a first real N > 1 run diagnosable from its one line: a per-rank table that sums to the totals ...
    table = line["per_rank"]
    assert [r["rank"] for r in table] == [0, 1, 2] and all(r["device"] == 0 for r in table)           # the rehearsal: every rank on GPU 0
    assert sum(r["rows"] for r in table) == 256 and sum(r["rays"] for r in table) == rays * line["steps"]
    assert all(r["kernel_ms"] > 0 and r["step_wall_ms"] >= r["kernel_ms"] and r["gather_ms"] >= 0 for r in table)
    assert max(r["step_wall_ms"] for r in table) <= line["ms_per_step"] * 1.5 + 50
    # ... and the oracle's verdict on 16 rows of the GATHERED image
    par = line["parity"]
    assert par["rows"] == 16 and par["pixels"] == 16 * 256 and par["pixels_differing"] == 0 and par["pixels_over_tol"] == 0


def test_a_dead_rank_ends_the_gpu_run_with_an_error(amber, tmp_path):
    """The same flow with rank 1 killed after the rendezvous: non-zero exit, no JSON line, well inside the time-out."""
    import time
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", AMBER_BENCH_KILL_RANK="1", AMBER_BENCH_DIST_TIMEOUT_S="20")
    t0 = time.time()
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--gpus", "3", "--rehearse-on-one-gpu",
                        "--width", "64", "--spp", "8"], capture_output=True, text=True, env=e, timeout=300)
    assert p.returncode != 0 and not [l for l in p.stdout.splitlines() if l.startswith('{"metric"')], p.stdout
    assert time.time() - t0 < 120


def test_rccl_gather_is_ordered_after_the_render(amber, tmp_path):
    line, img = _bench(tmp_path, "--width", "512", "--spp", "128", env={"AMBER_BENCH_FORCE_COLLECTIVE": "1"})
    assert line["n_gpus"] == 1 and line["collective_backend"] == "rccl"
    ref, _ = _reference_image(amber, 512, 128)
    assert np.array_equal(img.reshape(ref.shape).view(np.uint32), ref.view(np.uint32))


def test_more_ranks_than_stripes(amber, tmp_path):
    """16 rows = two 8-row stripes dealt to three ranks: rank 2 owns an empty band and still takes part in the gather."""
    line, img = _bench(tmp_path, "--gpus", "3", "--rehearse-on-one-gpu", "--width", "16", "--spp", "16")
    ref, _ = _reference_image(amber, 16, 16)
    assert line["n_gpus"] == 3 and np.array_equal(img.view(np.uint32), ref.view(np.uint32))
