"""Framebuffer-band sharding across the GPUs of one node (one process per GPU, torch.distributed).

The path shards trivially: with the per-(pixel,sample) sampler every pixel is independent, so each
rank renders a band of rows with its own amber_hip_pt handle and full scene replica, and there is no
exchange while rendering.  The only collective is one gather of the per-band radiance sums to rank 0
at the end (RCCL over xGMI when the backend is "nccl"; gloo on CPU for the tests).  The reference
has no counterpart: it is single-process.
"""
from __future__ import annotations

import numpy as np

TILE_ROWS = 8  # the engine works on 8x8 pixel tiles; bands are multiples of 8 rows


def partition_rows(height: int, world_size: int):
    """Contiguous bands of rows, multiples of TILE_ROWS, as even as possible.  Returns [(y0, y1)] per rank;
    ranks beyond the number of tile rows get an empty band (y0 == y1)."""
    if height <= 0 or world_size <= 0:
        raise ValueError("height and world_size must be positive")
    n_tiles = (height + TILE_ROWS - 1) // TILE_ROWS
    base, extra = divmod(n_tiles, world_size)
    bands, t = [], 0
    for r in range(world_size):
        nt = base + (1 if r < extra else 0)
        y0, y1 = min(t * TILE_ROWS, height), min((t + nt) * TILE_ROWS, height)
        bands.append((y0, y1))
        t += nt
    return bands


def gather_bands(local_band, bands, width: int, rank: int, world_size: int, dst: int = 0):
    """Gather the per-rank band sums (torch tensors, shape (rows_r, width, 3), on the backend's device) into
    the full (height, width, 3) image on rank `dst`; returns None elsewhere.  One collective."""
    import torch
    import torch.distributed as dist

    max_rows = max(y1 - y0 for y0, y1 in bands)
    dev = local_band.device
    send = torch.zeros((max_rows, width, 3), dtype=torch.float32, device=dev)
    rows = bands[rank][1] - bands[rank][0]
    if rows:
        send[:rows].copy_(local_band.reshape(rows, width, 3))
    if world_size == 1:
        return send[:rows].clone()
    recv = [torch.empty_like(send) for _ in range(world_size)] if rank == dst else None
    dist.gather(send, gather_list=recv, dst=dst)
    if rank != dst:
        return None
    height = bands[-1][1]
    full = torch.empty((height, width, 3), dtype=torch.float32, device=dev)
    for r, (y0, y1) in enumerate(bands):
        if y1 > y0:
            full[y0:y1].copy_(recv[r][: y1 - y0])
    return full


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
    return torch.as_tensor(DeviceArray(ptr, shape), device=device)
