"""world_size-2/3 gloo test of the N>1 path: stripe partition -> per-rank render -> one gather -> full image.

There is no GPU here, so each rank renders its band with the ORACLE (tests may use it as a stand-in
renderer); what is under test is amber_amd.distributed (partition, padding, the single gather, re-assembly)
and that the assembled image equals a single-process render bit for bit (the property that makes band
sharding valid: per-(pixel,sample) sampling makes pixels independent of the partitioning).
"""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent

WORKER = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
import numpy as np, torch, torch.distributed as dist
import oracle_binding as O
from amber_amd.distributed import stripe_partition, gather_rows
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
W, H, spp, seed = 40, {height}, 3, 31
parts = stripe_partition(H, world, 4)
idx = parts[rank]["index"]
sc = O.Scene.cornell(O.ACCEL_LIST)
full = np.zeros((H, W, 3), np.float32); casts = 0
for y in idx:                       # the oracle stands in for the engine: render exactly this rank's rows
    _, c = sc.render_xorshift(W, H, seed, 0, spp, rows=(int(y), int(y) + 1), threads=1, out=full); casts += c.casts
local = torch.from_numpy(full[idx].copy())
rays = torch.tensor([casts], dtype=torch.int64)
dist.reduce(rays, dst=0)
img = gather_rows(local, parts, W, rank, world)
if rank == 0:
    np.save({out!r}, img.numpy()); np.save({out!r} + ".rays.npy", rays.numpy())
dist.barrier(); dist.destroy_process_group()
"""


@pytest.mark.parametrize("world,height", [(2, 52), (3, 52), (3, 8)])       # (3, 8): two 4-row stripes, rank 2 owns an EMPTY band
def test_band_sharding_gloo(tmp_path, world, height):
    import subprocess
    import oracle_binding as O
    out = str(tmp_path / "img.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=str(ROOT), out=out, height=height))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + world + os.getpid() % 200))
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                    "--master-port", env["MASTER_PORT"], str(script)], check=True, env=env, timeout=300)
    got = np.load(out)
    ref, cnt = O.Scene.cornell(O.ACCEL_LIST).render_xorshift(40, height, 31, 0, 3)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert int(np.load(out + ".rays.npy")[0]) == cnt.casts


def _selftest(world, extra_env=None, timeout=120):
    """`bench.py --gpus N --dist-selftest`: bench.py's own launcher + rendezvous + gather + per-rank table on the CPU (gloo)."""
    import json
    import subprocess
    import time
    env = dict(os.environ, **(extra_env or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(world), "--dist-selftest", "--width", "64"],
                       capture_output=True, text=True, env=env, timeout=timeout)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith('{"selftest"')]
    return p.returncode, lines, time.time() - t0, p.stderr


def test_bench_rank_plumbing_reports_every_rank():
    """The one JSON line of a multi-rank run carries a per-rank table whose rows and rays sum to the job's (bench.py per_rank_table)."""
    rc, lines, _, err = _selftest(3)
    assert rc == 0 and len(lines) == 1, err[-2000:]
    line = lines[0]
    assert line["n_gpus"] == 3 and line["gathered_rows_in_order"] is True
    table = line["per_rank"]
    assert [r["rank"] for r in table] == [0, 1, 2]
    assert sum(r["rows"] for r in table) == 64 and sum(r["rays"] for r in table) == 64 * 64
    assert all(set(r) == {"rank", "device", "rows", "rays", "kernel_ms", "step_wall_ms", "gather_ms"} for r in table)


def test_bench_exits_nonzero_when_a_rank_dies():
    """A rank that dies after the rendezvous must end the run with a non-zero code within the time-out -- not leave the others in a barrier
    (VERDICT r04 item 3).  AMBER_BENCH_KILL_RANK makes rank 1 exit(17) before the first barrier; the survivors' watchdog is 10 + 5 s."""
    rc, lines, seconds, err = _selftest(3, {"AMBER_BENCH_KILL_RANK": "1", "AMBER_BENCH_DIST_TIMEOUT_S": "10"}, timeout=90)
    assert rc != 0 and not lines, (rc, lines)
    assert seconds < 60, seconds
    assert "AMBER_BENCH_KILL_RANK" in err


def test_bench_exits_nonzero_when_a_rank_never_arrives():
    """WORLD_SIZE says 2, one rank shows up: the rendezvous time-out (not the lease's) ends it."""
    import subprocess
    import time
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200),
               AMBER_BENCH_DIST_TIMEOUT_S="5")
    t0 = time.time()
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dist-selftest", "--width", "64"], capture_output=True, text=True, env=env, timeout=90)
    assert p.returncode != 0 and time.time() - t0 < 60, (p.returncode, p.stderr[-500:])
