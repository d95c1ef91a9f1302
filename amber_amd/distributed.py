"""Framebuffer-band sharding across the GPUs of one node (one process per GPU, torch.distributed).

The path shards trivially: with the per-(pixel,sample) sampler every pixel is independent, so each
rank renders a band of rows with its own amber_hip_pt handle and full scene replica, and there is no
exchange while rendering.  The only collective is one gather of the per-band radiance sums to rank 0
at the end (RCCL over xGMI when the backend is "nccl"; gloo on CPU for the tests).  The reference
has no counterpart: it is single-process.
"""
from __future__ import annotations

import numpy as np

STRIPE_ROWS = 8  # rows per stripe of the interleaved sharding


def partition_rows(height: int, world_size: int):
    """Contiguous bands of rows (multiples of 8, as even as possible).  Returns [(y0, y1)] per rank; ranks beyond
    the number of 8-row groups get an empty band (y0 == y1).  Kept for callers that want contiguous output;
    interleaved stripes (stripe_partition) balance the load much better on the Cornell scene."""
    if height <= 0 or world_size <= 0:
        raise ValueError("height and world_size must be positive")
    n_tiles = (height + 7) // 8
    base, extra = divmod(n_tiles, world_size)
    bands, t = [], 0
    for r in range(world_size):
        nt = base + (1 if r < extra else 0)
        y0, y1 = min(t * 8, height), min((t + nt) * 8, height)
        bands.append((y0, y1))
        t += nt
    return bands


def stripe_partition(height: int, world_size: int, stripe_rows: int = STRIPE_ROWS):
    """Interleaved sharding: rank r renders the rows y with (y // stripe_rows) % world_size == r, i.e. every
    world_size-th stripe, so each rank sees the same mix of cheap and expensive image regions.
    Returns per rank a dict(rows=(y0, y1), stripe=(S, period) or None, index=np.ndarray of global rows)."""
    if height <= 0 or world_size <= 0 or stripe_rows <= 0:
        raise ValueError("height, world_size and stripe_rows must be positive")
    parts = []
    ys = np.arange(height)
    for r in range(world_size):
        if world_size == 1:
            parts.append(dict(rows=(0, height), stripe=None, index=ys))
            continue
        index = ys[(ys // stripe_rows) % world_size == r]
        y0 = min(r * stripe_rows, height)
        parts.append(dict(rows=(y0, height), stripe=(stripe_rows, stripe_rows * world_size), index=index))
    return parts


class RowGatherer:
    """The job's single collective with its buffers and index tensors allocated once: gathers every rank's rows
    (tensor (n_r, width, 3) on the backend's device) to rank `dst` and puts them at their global row positions.
    At 8 ranks a Cornell step is ~8 ms per rank, so per-step allocations, index uploads and small launches would show:
    the ranks' rows land in ONE buffer (world, max_rows, width, 3) and ONE index_select with a permutation built here
    moves them into global row order; a rank whose band fills its slot sends the band tensor itself (no staging copy)."""

    def __init__(self, parts, width: int, rank: int, world_size: int, device, dst: int = 0, force_collective: bool = False):
        import torch

        self.parts, self.width, self.rank, self.world, self.dst = parts, width, rank, world_size, dst
        self.n = len(parts[rank]["index"])
        self.max_rows = max(len(p["index"]) for p in parts)
        self.height = sum(len(p["index"]) for p in parts)
        self.collective = world_size > 1 or force_collective     # force_collective: run the gather even for one rank (RCCL smoke test)
        if self.collective:
            self.send = None if self.n == self.max_rows else torch.zeros((self.max_rows, width, 3), dtype=torch.float32, device=device)
            if rank == dst:
                self.recv_all = torch.empty((world_size, self.max_rows, width, 3), dtype=torch.float32, device=device)
                self.recv = [self.recv_all[r] for r in range(world_size)]
                self.full = torch.empty((self.height, width, 3), dtype=torch.float32, device=device)
                perm = np.empty(self.height, np.int64)                   # global row y <- slot r * max_rows + its local row
                for r, p in enumerate(parts):
                    perm[np.asarray(p["index"], np.int64)] = r * self.max_rows + np.arange(len(p["index"]), dtype=np.int64)
                self.perm = torch.as_tensor(perm, dtype=torch.long, device=device)

    def __call__(self, local_rows):
        """Returns the (height, width, 3) image on `dst` (a buffer reused by the next call), None elsewhere."""
        import torch
        import torch.distributed as dist

        if not self.collective:
            return local_rows.reshape(self.n, self.width, 3)
        rows = local_rows.reshape(self.n, self.width, 3)
        if self.send is None and rows.is_contiguous():
            send = rows
        else:
            if self.send is None:
                self.send = torch.zeros((self.max_rows, self.width, 3), dtype=torch.float32, device=rows.device)
            if self.n:
                self.send[: self.n].copy_(rows)
            send = self.send
        dist.gather(send, gather_list=self.recv if self.rank == self.dst else None, dst=self.dst)
        if self.rank != self.dst:
            return None
        torch.index_select(self.recv_all.view(self.world * self.max_rows, self.width, 3), 0, self.perm, out=self.full)
        return self.full


def gather_rows(local_rows, parts, width: int, rank: int, world_size: int, dst: int = 0):
    """One-shot form of RowGatherer (allocates its buffers per call)."""
    return RowGatherer(parts, width, rank, world_size, local_rows.device, dst)(local_rows)


def gather_bands(local_band, bands, width: int, rank: int, world_size: int, dst: int = 0):
    """gather_rows for contiguous bands [(y0, y1)]."""
    parts = [dict(rows=b, stripe=None, index=np.arange(b[0], b[1])) for b in bands]
    return gather_rows(local_band, parts, width, rank, world_size, dst)


class DeviceArray:
    """Zero-copy view of a device pointer for torch.as_tensor (CUDA array interface; works on ROCm)."""

    def __init__(self, ptr: int, shape, typestr: str = "<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2}


def band_tensor(tracer, device: str = "cuda"):
    """torch tensor aliasing the engine's band framebuffer (no copy)."""
    import torch

    ptr, n = tracer.device_framebuffer()
    shape = tracer.band_shape
    assert n == int(np.prod(shape))
    if n == 0:                                   # empty band (more ranks than stripes): nothing to alias
        return torch.empty(shape, dtype=torch.float32, device=device)
    return torch.as_tensor(DeviceArray(ptr, shape), device=device)
