#!/usr/bin/env python
"""bench.py -- Mrays/s of the path-tracing hot path on N GPUs of one node (driver contract in the task prompt).

A "step" is one complete render of BASELINE.json's configs[1]: Cornell box, 1024x1024, 1024 samples per
pixel (1.07e9 paths, ~2.25e9 rays), i.e. one pass of the hot path over the whole job.  For N > 1 the
framebuffer rows are dealt to the N ranks in interleaved 8-row stripes (strong scaling: the job is fixed), every
rank renders its rows with its own scene replica, and each step ends with the single gather of the row sums
to rank 0 (RCCL).

Prints ONE JSON line on rank 0.  `value` = rays of all ranks / max-over-ranks wall time of the K timed
steps (inputs resident in HBM; barrier + synchronize on both sides).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
BYTES_PER_RAY = 96.0      # algorithmic ray-state bytes per bounce (DESIGN.md "Roofline")


def measured_copy_bandwidth(torch, device) -> float:
    """Achievable HBM bandwidth of this device: a 1 GiB device-to-device copy (read + write), GB/s (SURVEY.md 8(d):
    the roofline fraction is also reported against what the box actually reaches, not only the vendor figure)."""
    n = 1 << 28                                          # 2^28 float32 = 1 GiB
    a = torch.empty(n, dtype=torch.float32, device=device).normal_()
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 10
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * 4.0 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def cpu_baseline(width: int, seed: int, budget_s: float = 15.0):
    """The CPU restatement (oracle, kind "port") on the host cores, reference threading model (rows of whole-image
    1-spp passes pulled by T threads), on a bounded sample of the SAME workload: full 1024x1024 frame, few spp."""
    sys.path.insert(0, str(ROOT / "tests"))
    import numpy as np
    import oracle_binding as O

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("AMBER_BENCH_CPU_THREADS", "16"))))   # a 1-GPU box's CPU share is 16 cores
    sc = O.Scene.cornell(O.ACCEL_BVH)          # the reference's own acceleration structure
    t0 = time.perf_counter()
    _, cnt = sc.render_xorshift(width, width, seed, 0, 1, math=O.MATH_LIBM, threads=cores)
    dt1 = time.perf_counter() - t0
    spp = max(1, min(512, int(budget_s / max(dt1, 1e-3))))
    t0 = time.perf_counter()
    _, cnt = sc.render_xorshift(width, width, seed, 1, spp, math=O.MATH_LIBM, threads=cores)
    dt = time.perf_counter() - t0
    out = {"value": round(cnt.casts / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
           "sample": f"Cornell {width}x{width} @ {spp} spp of the 1024 ({cnt.casts} rays in {dt:.2f} s), oracle in BVH/libm mode, "
                     f"{cores} threads; scale linearly in spp"}
    if cores > 1:                                # also one thread (a quarter of the frame's rows, 1 spp)
        t0 = time.perf_counter()
        _, c1 = sc.render_xorshift(width, width, seed, 0, 1, math=O.MATH_LIBM, threads=1, rows=(0, max(1, width // 4)))
        d1 = time.perf_counter() - t0
        out["single_thread_value"] = round(c1.casts / d1 / 1e6, 3)
        out["sample"] += f"; 1 thread: rows 0..{max(1, width // 4)} @ 1 spp ({c1.casts} rays in {d1:.2f} s)"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--spp-per-launch", type=int, default=1024)
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a single GPU: every rank uses device 0 and the gather runs over gloo on host copies")
    ap.add_argument("--engine", default="auto", choices=["auto", "list", "two_phase", "bvh", "wavefront"],
                    help="closest-hit / scheduling engine (default: auto = the fastest valid one; others for comparison)")
    args = ap.parse_args()

    import numpy as np
    import torch

    import amber_amd
    from amber_amd.distributed import RowGatherer, band_tensor, stripe_partition

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: amber_amd has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    force_collective = world == 1 and os.environ.get("AMBER_BENCH_FORCE_COLLECTIVE") == "1"   # one-rank RCCL rehearsal of the N > 1 code path
    if force_collective:
        os.environ.setdefault("MASTER_PORT", "29571")
    json_fd = 1
    if world > 1 or force_collective:
        # RCCL prints a version banner on STDOUT when the communicator is created; the contract is ONE JSON line on
        # stdout, so everything else written to fd 1 from here on goes to stderr and the JSON line to the saved fd
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    W = H = args.width
    sensor = amber_amd.Sensor.default(W, H)
    scene = amber_amd.HostScene.cornell_box()                      # etude::CornelBox(0.050, 0.050, 6), application.cc:68-73
    parts = stripe_partition(H, world)                             # interleaved 8-row stripes: equal work per rank
    mine = parts[rank]
    stream = torch.cuda.current_stream().cuda_stream              # launch on torch's stream: ordered with the gather
    tracer = amber_amd.PathTracer(scene, sensor, seed=args.seed, device=local_rank, rows=mine["rows"], stripe=mine["stripe"],
                                  stream=stream, engine={"auto": 0, "list": 1, "two_phase": 2, "bvh": 3, "wavefront": 4}[args.engine])
    fb = band_tensor(tracer, f"cuda:{local_rank}")
    launches = [(s, min(args.spp_per_launch, args.spp - s)) for s in range(0, args.spp, args.spp_per_launch)]
    gather = RowGatherer(parts, W, rank, world, "cpu" if (args.rehearse_on_one_gpu and world > 1) else fb.device, force_collective=force_collective)   # buffers allocated once

    def step():
        tracer.clear()
        for first, n in launches:
            tracer.render_pass(first, n)
        if args.rehearse_on_one_gpu and world > 1:
            torch.cuda.synchronize()
            return gather(fb.cpu())                                # gloo: host tensors
        return gather(fb)                                          # the single collective of the job

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    rays_before = tracer.ray_count()                               # clear() in step() resets it; take the last step's count below
    t0 = time.perf_counter()
    total_rays_local, kernel_ms, n_launch = 0, 0.0, 0
    for _ in range(args.steps):
        img = step()
        # ray counter and kernel times are read after the step's work is enqueued; download syncs the stream
        total_rays_local += tracer.ray_count()
        nl, ms = tracer.kernel_time()
        kernel_ms += ms; n_launch += nl
    fence()
    dt = time.perf_counter() - t0

    red_dev = "cpu" if args.rehearse_on_one_gpu else f"cuda:{local_rank}"
    t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    r = torch.tensor([total_rays_local], dtype=torch.int64, device=red_dev)
    k = torch.tensor([kernel_ms / max(n_launch, 1)], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        dist.all_reduce(k, op=dist.ReduceOp.MAX)
    dt_max, rays, kern_ms = float(t.item()), int(r.item()), float(k.item())

    if rank == 0 and os.environ.get("AMBER_BENCH_SAVE_IMAGE"):
        np.save(os.environ["AMBER_BENCH_SAVE_IMAGE"], img.detach().cpu().numpy())
    if rank == 0:
        rays_per_launch_local = total_rays_local / max(n_launch, 1)
        achieved = rays_per_launch_local * BYTES_PER_RAY / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        out = {
            "metric": "Mrays/sec + wall-clock, Cornell 1024^2 @1024spp; 1/2/4/8 GPU",
            "value": round(rays / dt_max / 1e6, 3),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"Cornell box (etude::CornelBox(0.050,0.050,6)) {W}x{H} @ {args.spp} spp, RR-only path tracing, "
                                   f"per-(pixel,sample) XorShift seed {args.seed}", "rays_per_step": rays // args.steps,
                       "paths_per_step": W * H * args.spp, "wall_s_per_step": round(dt_max / args.steps, 4),
                       "parallelism": f"stripes{world}x8rows", "launches_per_step": len(launches), "engine": "work-queue megakernel, two-phase closest hit" if args.engine == "auto" else args.engine},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "kernel": "pt_megakernel" if args.engine != "wavefront" else "wf_generate + wf_bounce launches of one batch", "kernel_ms": round(kern_ms, 3),
                         "note": "achieved = rays per launch (rank 0) x 96 B algorithmic ray-state bytes / mean launch duration (hipEvents)"},
        }
        profs = sorted((ROOT / "profiles").glob("r*_hbm_traffic.json"))      # latest committed rocprofv3 PMC summary
        if profs:
            try:
                out["roofline"]["traffic"] = json.loads(profs[-1].read_text()).get("hbm_bytes_per_launch")
                out["roofline"]["traffic_source"] = f"profiles/{profs[-1].name}"
            except Exception:
                pass
        if world == 1:
            try:
                bw = measured_copy_bandwidth(torch, f"cuda:{local_rank}")
                out["roofline"]["measured_copy_bw"] = round(bw, 1)
                out["roofline"]["frac_of_measured_copy_bw"] = round(achieved / bw, 5)
            except Exception as e:                                     # never lose the bench line over the side measurement
                out["roofline"]["measured_copy_bw"] = None
                out["roofline"]["measured_copy_bw_error"] = str(e)[:120]
        if not args.no_cpu_baseline and world == 1:                    # reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(W, args.seed)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
