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
