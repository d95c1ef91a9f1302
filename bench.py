#!/usr/bin/env python
"""bench.py -- Mrays/s of the path-tracing hot path on N GPUs of one node (driver contract in the task prompt).

A "step" is one complete render of BASELINE.json's configs[1]: Cornell box, 1024x1024, 1024 samples per
pixel (1.07e9 paths, ~2.25e9 rays), i.e. one pass of the hot path over the whole job.  For N > 1 the
framebuffer rows are dealt to the N ranks in interleaved 8-row stripes (strong scaling: the job is fixed), every
rank renders its rows with its own scene replica, and each step ends with the single gather of the row sums
to rank 0 (RCCL), enqueued on the stream the render runs on.

`python bench.py --gpus N` starts its own N ranks (a torch.distributed.run child process) when it was not itself
started by a launcher; under a launcher (RANK / WORLD_SIZE in the environment) it is one of the ranks.

Prints ONE JSON line on rank 0.  `value` = rays of all ranks / max-over-ranks wall time of the K timed
steps (inputs resident in HBM; barrier + synchronize on both sides).

After the timed steps, and outside `value`, the N = 1 run starts ONE child process (`--side-child`; a child, never an exec; the
headline numbers are already on disk in gpurun_out/bench_headline.json by then, and a fault, hang or time-out of the child costs only
its own fields) that measures on this one GPU:
  `secondary`         the other BASELINE configurations: config 3 at its full size, and ONE rank's share of configs 4 and 5 (its stripes
                      of the real frame, one launch of the real samples-per-launch; config 5 with max depth 16); and where a `--scene` user
                      lands: the Cornell box through ENGINE_BVH (both schedulers) and just past the 32-object engine (33 / 49 objects), a room
                      with a 1.3k-triangle OBJ mesh, a 1.04M-triangle terrain OBJ -- written and imported through cli::ImportScene here;
  `projected_scaling` every rank's COMPLETE step of the headline job at N = 1 / 2 / 4 / 8 (clear, launch, record kernels, sync, the RCCL
                      gather of its rows run as a one-rank collective), max over the ranks' shares -- a projection, labelled as such;
  `cold_start`        create -> first downloaded frame of the headline job on a new handle (`cold_wall_ms`).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

ENGINES = {"auto": 0, "list": 1, "two_phase": 2, "bvh": 3, "wavefront": 4, "reference_bvh": 6}     # AMBER_ENGINE_* (include/amber_hip.h)
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


def host_cores():
    """(cores this process may run on, cores its cgroup's CPU quota lets it use at once)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = n
    try:
        q, p = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]            # cgroup v2
        if q != "max":
            quota = max(1, -(-int(q) // int(p)))
    except Exception:
        try:                                                                         # cgroup v1
            q = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
            p = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            if q > 0:
                quota = max(1, -(-q // p))
        except Exception:
            pass
    return n, min(n, quota)


def cpu_baseline(amber_amd, width: int, spp_job: int, seed: int, many_s: float = 5.0, one_s: float = 2.5):
    """The CPU leg (the only place bench.py touches oracle/): (1) the CPU restatement (oracle, kind "port") timed on the
    host cores with the reference's threading model (rows of whole-image 1-spp passes pulled by T threads,
    parallel.h:50-54), reference BVH and the host's libm, on a bounded sample of the SAME workload -- the full frame at a
    few spp -- with T = every core this process may use and with T = 1; (2) the same oracle as the CHECKER of the
    GPU's output: 16 full-width rows of the job at ALL its samples, images / ray counts / per-path signatures
    (tests/parity_rows.py).  Returns (cpu_baseline, parity)."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_binding as O
    from parity_rows import compare_rows

    nproc, usable = host_cores()
    cores = max(1, min(usable, int(os.environ.get("AMBER_BENCH_CPU_THREADS", "256"))))
    sc = O.Scene.cornell(O.ACCEL_BVH)          # the reference's own acceleration structure

    def timed(first, n, threads, rows=None):
        t0 = time.perf_counter()
        _, c = sc.render_xorshift(width, width, seed, first, n, math=O.MATH_LIBM, threads=threads, rows=rows)
        return c.casts, time.perf_counter() - t0

    timed(0, 1, cores)                                                    # warms the pages
    _, dt1 = timed(1, 4, cores)                                           # calibration (4 spp: long enough to see a CPU quota bite)
    spp = max(1, min(spp_job, int(many_s / max(dt1 / 4, 1e-3))))
    casts, dt = timed(5, spp, cores)
    out = {"value": round(casts / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port", "nproc": nproc,
           "sampler": "xorshift per (pixel, sample) -- the reference draws from one mt19937_64 per thread, about 13 % of its run time "
                      "(BASELINE.md section 2): this port is FASTER than the reference itself would be on these cores",
           "reference_calibration": None,
           "reference_calibration_reason": "the ratio port / true reference (SURVEY 8(d)) cannot be measured: the reference does not build in this image "
                                           "(boost headers absent, stand-ins not allowed); the survey's own figure for the unmodified reference is 1.46 Mrays/s "
                                           "per core of a 2.1 GHz Xeon (BASELINE.md section 2)",
           "sample": f"Cornell {width}x{width} @ {spp} spp of the {spp_job} ({casts} rays in {dt:.2f} s), oracle: XorShift sampler, "
                     f"reference BVH, host libm, {cores} threads (nproc {nproc}, cgroup CPU quota {usable}); scale linearly in spp"}
    if cores > 1:                                                         # also ONE thread, >= 2 s of it
        rows1 = (0, width)
        c0, d0 = timed(0, 1, 1, rows=(0, max(1, width // 8)))             # calibration on an eighth of the frame
        spp1 = max(1, int(one_s / max(d0 * 8, 1e-3)))
        c1, d1 = timed(1, spp1, 1, rows=rows1)
        out["single_thread_value"] = round(c1 / d1 / 1e6, 3)
        out["sample"] += f"; 1 thread: full frame @ {spp1} spp ({c1} rays in {d1:.2f} s)"
    # images and ray counts: the PRODUCT library, in this process.  Per-path signatures come from the signature instantiation of the same kernels,
    # which only the lab build has (include/amber_hip_lab.h): a child process on libamber_hip_lab.so adds them.
    parity = compare_rows(amber_amd, width, spp_job, seed, threads=cores, math=O.MATH_LIBM, accel=O.ACCEL_BVH, signatures=amber_amd.is_lab())
    parity["library"] = amber_amd.library_path().name
    if not amber_amd.is_lab():
        code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r); import amber_amd, oracle_binding as O; from parity_rows import compare_rows; "
                "r = compare_rows(amber_amd, %d, %d, %d, threads=%d, math=O.MATH_LIBM, accel=O.ACCEL_BVH); r['library'] = amber_amd.library_path().name; print('PARITY ' + json.dumps(r))"
                % (str(ROOT), str(ROOT / "tests"), width, spp_job, seed, cores))
        try:
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, AMBER_AMD_LIB="libamber_hip_lab.so"))
            line = [l for l in r.stdout.splitlines() if l.startswith("PARITY ")]
            parity["signatures_lab_build"] = json.loads(line[0][7:]) if line else {"error": (r.stderr or "no result")[-200:]}
        except Exception as e:
            parity["signatures_lab_build"] = {"error": str(e)[:200]}
    return out, parity


def latest_profile_summary():
    """VALU-side figures of the dominant kernel from the newest committed rocprofv3 PMC summary (profiles/rNN*_summary.json,
    written by tools/summarize_profile.py): the kernel is VALU-issue-bound, not HBM-bound, and the line says so."""
    best = None
    for p in sorted((ROOT / "profiles").glob("r*_summary.json")):
        try:
            d = json.loads(p.read_text())
        except Exception:
            continue
        if isinstance(d, dict) and "valu" in d and d.get("kernel") == "pt_megakernel":
            best = (p.name, d)
    return best


def secondary_workloads(amber_amd, np, seed: int, device: int):
    """BASELINE configs 3, 4 and 5 on THIS GPU, outside `value`: config 3 whole;
    for the multi-GPU configs the share of rank 0 -- its interleaved 8-row stripes of the real frame -- for ONE launch of the
    real samples-per-launch (1024).  Every entry: kernel time by hipEvents, rays, Mrays/s, the section-8(d) contract fraction."""
    from amber_amd import scenes
    from amber_amd.distributed import stripe_partition
    out = []

    def run(name, scene, W, H, spp, engine_name, max_depth=0, world=1, warm=8, engine=0, flags=0):
        part = stripe_partition(H, world)[0]
        t0 = time.perf_counter()
        pt = amber_amd.PathTracer(scene, amber_amd.Sensor.default(W, H), seed=seed, device=device, max_depth=max_depth,
                                  rows=part["rows"], stripe=part["stripe"], engine=engine, flags=flags)
        t_create = time.perf_counter() - t0
        pt.render_pass(0, warm); pt.sync(); pt.clear()                       # first launch of a handle: page-in + record-density probe
        t0 = time.perf_counter()
        pt.render_pass(0, spp); pt.sync()
        wall = time.perf_counter() - t0
        n_launch, ms = pt.kernel_time()
        rays = pt.ray_count()
        rows = len(part["index"])
        entry = {"workload": name, "frame": [W, H], "rows_of_this_rank": rows, "ranks_of_the_config": world, "spp_in_launch": spp, "max_depth": max_depth,
                 "engine": engine_name, "launches": n_launch, "kernel_ms": round(ms, 3), "wall_ms": round(wall * 1e3, 3), "rays": int(rays),
                 "paths": rows * W * spp, "Mrays_s": round(rays / ms / 1e3, 1) if ms > 0 else None,
                 "roofline_frac": round(rays * BYTES_PER_RAY / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if ms > 0 else None,
                 "create_s": round(t_create, 3)}
        pt.close()
        return entry

    t0 = time.perf_counter()
    spheres = amber_amd.HostScene.create_arrays(**scenes.random_spheres(1_000_000, 7))
    t_scene = time.perf_counter() - t0
    e = run("config 3: 1M random spheres (deep BVH) 1920x1080 @ 256 spp, 1 GPU", spheres, 1920, 1080, 256,
            "work-queue megakernel, engine BVH (16-bit quantised 2-wide tree, resumable LDS-stack traversal)")
    e["host_scene_s"] = round(t_scene, 3)
    out.append(e)
    # the same frame through the reference's OWN tree and traversal order (AMBER_ENGINE_REFERENCE_BVH): the hits the reference's command line finds (oracle in its reference-BVH mode),
    # bit for bit (the List engines differ from it on 117 of this frame's pixels; profiles/r05_reference_bvh_full_frame.txt); create_s = the reference's build
    r = run("config 3 through AMBER_ENGINE_REFERENCE_BVH: the reference's own BVH (acceleration_bvh.h:134-403), 1920x1080 @ 256 spp", spheres, 1920, 1080, 256,
            "path-granular megakernel, per-lane walk of the reference's tree in the reference's order (binary32 boxes, stack in global memory)", engine=amber_amd.ENGINE_REFERENCE_BVH)
    r["kernel_ms_over_engine_bvh"] = round(r["kernel_ms"] / e["kernel_ms"], 3) if e["kernel_ms"] > 0 else None
    out.append(r)
    spheres.close()
    cornell = amber_amd.HostScene.cornell_box()
    out.append(run("config 4: Cornell + glass/refractive 2048x2048 @ 4096 spp on 4 GPUs -- rank 0's stripes, one 1024-spp launch of its 4",
                   cornell, 2048, 2048, 1024, "work-queue megakernel, two-phase closest hit", world=4))
    out.append(run("config 5: 3840x2160 @ 8192 spp, max depth 16, on 8 GPUs -- rank 0's stripes, one 1024-spp launch of its 8",
                   cornell, 3840, 2160, 1024, "work-queue megakernel, two-phase closest hit", max_depth=16, world=8))
    # Engine BVH where the reference's `--scene` users land (application.cc:74-87, import.cc:49-167): triangle meshes
    # written as OBJ + MTL (amber_amd/workloads.py) and read back through cli::ImportScene, and the engine switch at 33 objects itself.
    import tempfile
    from amber_amd import workloads
    two = run("Cornell 1024x1024 @ 1024 spp, two-phase (the headline kernel, for the ratio below)", cornell, 1024, 1024, 1024, "two-phase")
    e = run("mesh (i): the Cornell box through ENGINE_BVH, config 2's frame (1024x1024 @ 1024 spp) -- the engine switch at 33 objects",
            cornell, 1024, 1024, 1024, "path-granular megakernel, one-shot per-lane BVH traversal (tree depth 6 <= 12)", engine=amber_amd.ENGINE_BVH)
    e["kernel_ms_over_two_phase"] = round(e["kernel_ms"] / two["kernel_ms"], 3) if two["kernel_ms"] > 0 else None
    e["two_phase_kernel_ms"] = two["kernel_ms"]
    out.append(e)
    e = run("mesh (i'): the same through pt_bvh_megakernel (AMBER_PT_FLAG_BVH_ITEMS), the scheduler of deep trees",
            cornell, 1024, 1024, 1024, "item megakernel, resumable per-lane BVH traversal", engine=amber_amd.ENGINE_BVH, flags=amber_amd.api.PT_FLAG_BVH_ITEMS)
    e["kernel_ms_over_two_phase"] = round(e["kernel_ms"] / two["kernel_ms"], 3) if two["kernel_ms"] > 0 else None
    out.append(e)
    e = run("Cornell through AMBER_ENGINE_REFERENCE_BVH (the reference's tree of the box: a root and two leaves), config 2's frame at 256 spp",
            cornell, 1024, 1024, 256, "path-granular megakernel, per-lane walk of the reference's tree in the reference's order", engine=amber_amd.ENGINE_REFERENCE_BVH)
    e["kernel_ms_over_two_phase_at_equal_spp"] = round(e["kernel_ms"] * 4 / two["kernel_ms"], 3) if two["kernel_ms"] > 0 else None
    out.append(e)
    cornell.close()
    from amber_amd import scenes as _scenes
    for extra in (8, 24):                                                    # 33 and 49 objects: just past the 32-object two-phase engine
        kw = _scenes.cornell_plus(extra)
        hs = amber_amd.HostScene.create_arrays(**kw)
        e = run(f"mesh (i''): the Cornell box + {extra} small triangles = {len(kw['kinds']) + kw['n_blades']} objects, engine auto, config 2's frame (1024x1024 @ 1024 spp)",
                hs, 1024, 1024, 1024, "path-granular megakernel, two-phase closest hit over two groups of <= 32 objects")
        e["kernel_ms_over_two_phase"] = round(e["kernel_ms"] / two["kernel_ms"], 3) if two["kernel_ms"] > 0 else None
        out.append(e)
        hs.close()
    with tempfile.TemporaryDirectory() as tmp:
        for wl, W, H, spp, what in ((workloads.room_mesh(3), 1024, 1024, 256, "mesh (ii): Cornell-like room + bumpy icosphere, imported OBJ"),
                                    (workloads.terrain_mesh(16, 56), 1920, 1080, 64, "mesh (iii): displaced terrain with needle triangles at the tile seams, imported OBJ")):
            t0 = time.perf_counter()
            path = wl.write(tmp)
            t_write = time.perf_counter() - t0
            t0 = time.perf_counter()
            hs = amber_amd.HostScene.import_file(path)
            t_import = time.perf_counter() - t0
            e = run(f"{what}: {wl.n_triangles} triangles, {W}x{H} @ {spp} spp", hs, W, H, spp,
                    "engine auto = BVH: two-stage triangle leaves; path-granular kernel for trees of depth <= 12, item kernel beyond")
            e["obj_write_s"], e["import_s"] = round(t_write, 3), round(t_import, 3)
            out.append(e)
            hs.close()
    return out


def projected_scaling(amber_amd, torch, seed: int, device: int, width: int, spp: int, reps: int = 3, engine: int = 0):
    """What a step of the headline job would cost on N GPUs, measured on ONE: every rank's share (its interleaved 8-row stripes, as
    `stripe_partition` deals them) runs its COMPLETE step here -- clear, the launch, the rank / scan / place / reduce kernels, the host
    synchronisation, and the gather of its rows through RCCL as a one-rank collective on the render stream (launch + re-ordering cost;
    the transfer of the other ranks' rows over xGMI is estimated, see `xgmi_estimate_ms`) -- and the step of N ranks is the slowest
    share (the bench takes the max over ranks).  A PROJECTION: no run on more than one GPU exists for this line."""
    import numpy as np
    from amber_amd.distributed import RowGatherer, band_tensor, stripe_partition
    import torch.distributed as dist

    collective = dist.is_available() and dist.is_initialized()
    sensor = amber_amd.Sensor.default(width, width)
    scene = amber_amd.HostScene.cornell_box()
    out = {"what": "PROJECTED from one GPU: step wall of the slowest rank's share, every share run alone on this GPU (strong scaling of the headline job); "
                   "not a multi-GPU measurement", "spp": spp, "reps": reps,
           "step_includes": ["clear", "pt_megakernel", "rec_rank/scan/place + reduce_flagged", "host sync"] + (["1-rank RCCL gather + row re-ordering on the render stream"] if collective else []),
           "gather": "rccl, one rank" if collective else None, "per_n": []}
    base = base_k = None
    for n in (1, 2, 4, 8):
        parts = stripe_partition(width, n)
        walls, kerns, gathers, rays = [], [], [], []
        for part in parts:
            pt = amber_amd.PathTracer(scene, sensor, seed=seed, device=device, rows=part["rows"], stripe=part["stripe"], engine=engine)
            stream = torch.cuda.ExternalStream(pt.stream(), device=torch.device("cuda", device))
            fb = band_tensor(pt, f"cuda:{device}")
            own = [dict(rows=(0, len(part["index"])), stripe=None, index=np.arange(len(part["index"])))]
            gather = RowGatherer(own, width, 0, 1, fb.device, force_collective=collective)
            pt.render_pass(0, 8); pt.sync(); pt.clear()
            best, best_k, best_g = 1e9, 0.0, 0.0
            for _ in range(reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                pt.clear()                                              # (also resets the handle's launch timer)
                pt.render_pass(0, spp)
                pt.sync()
                g_ms = 0.0
                if collective:
                    with torch.cuda.stream(stream):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(stream); gather(fb); e1.record(stream)
                    torch.cuda.synchronize()
                    g_ms = e0.elapsed_time(e1)
                wall = (time.perf_counter() - t0) * 1e3
                _, k_ms = pt.kernel_time()
                if wall < best:
                    best, best_k, best_g = wall, k_ms, g_ms
            walls.append(best); kerns.append(best_k); gathers.append(best_g); rays.append(pt.ray_count())
            del fb, gather
            pt.close()
        step = max(walls)
        if n == 1:
            base, base_k = step, max(kerns)
        band_bytes = len(parts[0]["index"]) * width * 12
        out["per_n"].append({"n_gpus": n, "step_wall_ms": round(step, 3), "mean_share_wall_ms": round(sum(walls) / len(walls), 3),
                             "kernel_ms_of_slowest_share": round(max(kerns), 3), "gather_ms": round(max(gathers), 4) if collective else None,
                             "xgmi_estimate_ms": round(band_bytes / 120e9 * 1e3, 4) if n > 1 else 0.0,
                             "projected_speedup": round(base / step, 3),
                             "projected_speedup_by_kernel_time": round(base_k / max(kerns), 3) if max(kerns) > 0 else None})
    out["note"] = ("xgmi_estimate_ms = one rank's rows (bytes) / 120 GB/s of one xGMI link (7 links x ~153 GB/s per GPU, 0.8 efficiency; the root ingests the other "
                   "ranks' rows on distinct links concurrently): NOT included in step_wall_ms")
    return out


def cold_start(amber_amd, seed: int, device: int, width: int, spp: int, engine: int = 0):
    """What a user of the CLI waits for at the start of a render (application.cc:120-215): a NEW handle on the headline job -- create
    (scene flattening and upload), the per-pixel candidate masks, the one-chunk record-density probe, the first full launch and the
    download of the frame.  The process is warm (HIP initialised, code objects loaded); the handle is not."""
    import numpy as np
    scene = amber_amd.HostScene.cornell_box()
    best = None
    runs = []
    for _ in range(5):
        t0 = time.perf_counter()
        pt = amber_amd.PathTracer(scene, amber_amd.Sensor.default(width, width), seed=seed, device=device, engine=engine)
        t1 = time.perf_counter()
        pt.render_pass(0, 8)                                                # what amber_hip_pt_render_pass does first anyway: the probe chunk
        pt.sync()
        t2 = time.perf_counter()
        pt.render_pass(8, spp - 8)
        img, rays = pt.download()
        t3 = time.perf_counter()
        n_launch, ms = pt.kernel_time()
        pt.close()
        cur = {"cold_wall_ms": round((t3 - t0) * 1e3, 3), "create_ms": round((t1 - t0) * 1e3, 3), "create_to_end_of_first_launch_ms": round((t2 - t0) * 1e3, 3),
               "first_launch_incl_pixel_masks_ms": round((t2 - t1) * 1e3, 3), "rest_and_download_ms": round((t3 - t2) * 1e3, 3),
               "launches": n_launch, "kernel_ms_total": round(ms, 3), "rays": int(rays)}
        runs.append(cur["create_to_end_of_first_launch_ms"])
        if best is None or cur["create_to_end_of_first_launch_ms"] < best["create_to_end_of_first_launch_ms"]:
            best = cur
    best["create_to_end_of_first_launch_ms_all_runs"] = runs
    best["what"] = (f"new handle, Cornell {width}x{width} @ {spp} spp: create -> frame downloaded; five new handles in a warm process, the fields are those of the run "
                    "with the shortest create -> end of first launch (all five listed)")
    return best


def side_child(args):
    """`bench.py --side-child`: the measurements that follow the headline (module docstring), in a process of their own.  One JSON line on
    the saved stdout; RCCL's banner and everything else goes to stderr."""
    import numpy as np
    import torch
    import amber_amd
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    device = 0
    torch.cuda.set_device(device)
    out = {}
    try:
        out["cold_start"] = cold_start(amber_amd, args.seed, device, args.width, args.spp, engine=ENGINES[args.engine])
        out["cold_start"]["engine"] = args.engine
    except Exception as e:
        out["cold_start"] = None; out["cold_start_error"] = str(e)[:200]
    try:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        try:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", device))
        except Exception as e:                                             # the projection still runs, without the gather
            out["projected_scaling_gather_error"] = str(e)[:200]
        out["projected_scaling"] = projected_scaling(amber_amd, torch, args.seed, device, args.width, args.spp, engine=ENGINES[args.engine])
        out["projected_scaling"]["engine"] = args.engine
        if dist.is_initialized():
            dist.destroy_process_group()
    except Exception as e:
        out["projected_scaling"] = None; out["projected_scaling_error"] = str(e)[:200]
    try:
        out["secondary"] = secondary_workloads(amber_amd, np, args.seed, device)
    except Exception as e:
        out["secondary"] = None; out["secondary_error"] = str(e)[:200]
    os.write(json_fd, (json.dumps({"side": out}) + "\n").encode())


def run_side_child(args, timeout_s: float = 420.0):
    """Starts side_child as a child process and returns its dict (or an error entry).  The parent has released its GPU buffers."""
    cmd = [sys.executable, str(Path(__file__).resolve()), "--side-child", "--seed", str(args.seed), "--width", str(args.width), "--spp", str(args.spp), "--engine", args.engine]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"side_error": f"the side-measurement child did not finish in {timeout_s:.0f} s"}
    for line in r.stdout.splitlines():
        if line.startswith('{"side"'):
            return json.loads(line)["side"]
    return {"side_error": f"the side-measurement child ended with code {r.returncode} and no result"}


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def dist_timeout_s() -> float:
    """Seconds a rank waits for the others at the rendezvous and at the first barrier before it gives up (AMBER_BENCH_DIST_TIMEOUT_S, default 120)."""
    return float(os.environ.get("AMBER_BENCH_DIST_TIMEOUT_S", "120"))


def watchdog(seconds: float, rank: int, what: str):
    """A rank that is stuck in `what` for `seconds` EXITS (code 3) instead of holding the GPU lease: the launcher then ends the other ranks
    and returns non-zero.  os._exit from a timer thread: no exec, no clean-up that could block on the very collective that hangs."""
    import threading

    def fire():
        sys.stderr.write(f"bench.py: rank {rank}: {what} did not complete in {seconds:.0f} s -- giving up (exit 3)\n")
        sys.stderr.flush()
        os._exit(3)
    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def rendezvous(torch, backend: str, rank: int, world: int, device=None):
    """init_process_group + the first barrier, both under a time-out (a rank that fails must make the run exit non-zero,
    not hang).  AMBER_BENCH_KILL_RANK=r makes rank r die right after the rendezvous (the test of exactly that)."""
    import datetime
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    limit = dist_timeout_s()
    dog = watchdog(limit + 5.0, rank, f"the {backend} rendezvous of {world} ranks")
    kw = dict(rank=rank, world_size=world, timeout=datetime.timedelta(seconds=limit))
    if device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, **kw)
    dog.cancel()
    if os.environ.get("AMBER_BENCH_KILL_RANK") == str(rank):
        sys.stderr.write(f"bench.py: rank {rank}: AMBER_BENCH_KILL_RANK -- exiting 17 before the first barrier\n")
        sys.stderr.flush()
        os._exit(17)
    dog = watchdog(limit + 5.0, rank, "the first barrier")
    dist.barrier()
    dog.cancel()
    return dist


def per_rank_table(torch, dist, world: int, device, row):
    """all_gather of one float64 row per rank -> list of dicts (rank 0 prints it): what makes a first real N > 1 run diagnosable from its one line."""
    keys = ("rank", "device", "rows", "rays", "kernel_ms", "step_wall_ms", "gather_ms")
    mine = torch.tensor([float(row[k]) for k in keys], dtype=torch.float64, device=device)
    if dist is None or world == 1:
        got = [mine]
    else:
        got = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
    out = []
    for t in got:
        v = t.tolist()
        out.append({"rank": int(v[0]), "device": int(v[1]), "rows": int(v[2]), "rays": int(v[3]), "kernel_ms": round(v[4], 3),
                    "step_wall_ms": round(v[5], 3), "gather_ms": round(v[6], 4)})
    return out


def dist_selftest(args) -> int:
    """`bench.py --gpus N --dist-selftest` (under the launcher `--gpus N` starts): the multi-rank plumbing on the CPU -- rendezvous and first
    barrier with their time-outs, stripe partition, the single gather (gloo) of stand-in rows, the per-rank table, one JSON line on rank 0."""
    import numpy as np
    import torch
    from amber_amd.distributed import RowGatherer, stripe_partition
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dist = rendezvous(torch, "gloo", rank, world) if world > 1 else None
    W = H = args.width
    parts = stripe_partition(H, world)
    mine = parts[rank]
    rows = torch.from_numpy((np.asarray(mine["index"], np.float32)[:, None, None] + np.zeros((1, W, 3), np.float32)).copy())   # row y holds the value y
    gather = RowGatherer(parts, W, rank, world, "cpu")
    t0 = time.perf_counter()
    img = gather(rows)
    g_ms = (time.perf_counter() - t0) * 1e3
    table = per_rank_table(torch, dist, world, "cpu", dict(rank=rank, device=-1, rows=len(mine["index"]), rays=len(mine["index"]) * W, kernel_ms=0.0,
                                                           step_wall_ms=g_ms, gather_ms=g_ms))
    ok = True
    if rank == 0:
        ok = bool(np.array_equal(img.numpy()[:, 0, 0], np.arange(H, dtype=np.float32)))
        print(json.dumps({"selftest": "dist", "n_gpus": world, "backend": "gloo", "gathered_rows_in_order": ok, "per_rank": table}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process (never exec: nothing here has
    touched the GPU yet, and nothing will in this parent), relay rank 0's JSON line, propagate the exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = []
    for line in proc.stdout:
        if line.startswith('{"metric"') or line.startswith('{"selftest"'):
            lines.append(line)
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if lines:
        sys.stdout.write(lines[-1])
        sys.stdout.flush()
    return rc if rc else (0 if lines else 1)


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
    ap.add_argument("--no-secondary", action="store_true", help="skip the child process that follows the timed steps at N = 1 (configs 3 / 4 / 5, projected scaling, cold start)")
    ap.add_argument("--side-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a single GPU: every rank uses device 0 and the gather runs over gloo on host copies")
    ap.add_argument("--no-parity", action="store_true", help="N > 1: skip the oracle check of 16 rows of the gathered image on rank 0 after the timed steps")
    ap.add_argument("--dist-selftest", action="store_true",
                    help="CPU-only self-test of the multi-rank plumbing (gloo): rendezvous with its time-outs, first barrier, the row gather on stand-in rows, "
                         "the per-rank table; no GPU, no render (tests/test_distributed_cpu.py)")
    ap.add_argument("--engine", default="auto", choices=["auto", "list", "two_phase", "bvh", "wavefront", "reference_bvh"],
                    help="closest-hit / scheduling engine (default: auto = the fastest valid one; others for comparison)")
    args = ap.parse_args()

    if args.side_child:
        side_child(args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))

    if args.dist_selftest:
        raise SystemExit(dist_selftest(args))

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
        os.environ.setdefault("MASTER_PORT", str(free_port()))
    json_fd = 1
    backend = None
    if world > 1 or force_collective:
        # RCCL prints a version banner on STDOUT when the communicator is created; the contract is ONE JSON line on
        # stdout, so everything else written to fd 1 from here on goes to stderr and the JSON line to the saved fd
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        backend = "gloo" if args.rehearse_on_one_gpu else "nccl"
        dist = rendezvous(torch, backend, rank, world, torch.device("cuda", local_rank) if backend == "nccl" else None)

    W = H = args.width
    sensor = amber_amd.Sensor.default(W, H)
    scene = amber_amd.HostScene.cornell_box()                      # etude::CornelBox(0.050, 0.050, 6), application.cc:68-73
    parts = stripe_partition(H, world)                             # interleaved 8-row stripes: equal work per rank
    mine = parts[rank]
    # The handle renders on a stream of its own; the gather is enqueued on THAT stream (torch sees it as an external
    # stream), so the collective is ordered after the render without a host synchronisation in between.
    tracer = amber_amd.PathTracer(scene, sensor, seed=args.seed, device=local_rank, rows=mine["rows"], stripe=mine["stripe"],
                                  engine=ENGINES[args.engine])
    render_stream = torch.cuda.ExternalStream(tracer.stream(), device=torch.device("cuda", local_rank))
    fb = band_tensor(tracer, f"cuda:{local_rank}")
    launches = [(s, min(args.spp_per_launch, args.spp - s)) for s in range(0, args.spp, args.spp_per_launch)]
    gather = RowGatherer(parts, W, rank, world, "cpu" if (args.rehearse_on_one_gpu and world > 1) else fb.device, force_collective=force_collective)   # buffers allocated once

    collective = world > 1 or force_collective
    gather_events = []                                                 # (start, end) on the render stream: the gather + row re-ordering of a step
    gather_host_ms = []                                                # the gloo rehearsal: host time of the gather call

    def step(timed=False):
        tracer.clear()
        for first, n in launches:
            tracer.render_pass(first, n)
        if args.rehearse_on_one_gpu and world > 1:
            tracer.sync()
            tg = time.perf_counter()
            img_ = gather(fb.cpu())                                # gloo: host tensors
            if timed:
                gather_host_ms.append((time.perf_counter() - tg) * 1e3)
            return img_
        if collective:
            tracer.sync()                                          # a launch that ran out of record slots is repeated HERE: the
                                                                   # gather below must not ship a framebuffer that still misses it
        with torch.cuda.stream(render_stream):
            if timed and collective:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(render_stream)
                img_ = gather(fb)                                  # the single collective of the job
                e1.record(render_stream)
                gather_events.append((e0, e1))
                return img_
            return gather(fb)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    total_rays_local, kernel_ms, n_launch = 0, 0.0, 0
    img = None
    step_walls = []                                                    # this rank's own step: clear -> launches -> gather enqueued -> its stream drained
    for _ in range(args.steps):
        ts = time.perf_counter()
        img = step(timed=True)
        # ray counter and kernel times are read after the step's work is enqueued; download syncs the stream
        total_rays_local += tracer.ray_count()
        step_walls.append((time.perf_counter() - ts) * 1e3)
        nl, ms = tracer.kernel_time()
        kernel_ms += ms; n_launch += nl
    fence()
    dt = time.perf_counter() - t0

    red_dev = "cpu" if args.rehearse_on_one_gpu else f"cuda:{local_rank}"
    t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    r = torch.tensor([total_rays_local], dtype=torch.int64, device=red_dev)
    k = torch.tensor([kernel_ms / max(n_launch, 1)], dtype=torch.float64, device=red_dev)
    gms = [a_.elapsed_time(b_) for a_, b_ in gather_events]            # (the fence above has synchronised them)
    g = torch.tensor([sum(gms) / len(gms) if gms else 0.0], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        dist.all_reduce(k, op=dist.ReduceOp.MAX)
        dist.all_reduce(g, op=dist.ReduceOp.MAX)
    dt_max, rays, kern_ms, gather_ms = float(t.item()), int(r.item()), float(k.item()), float(g.item())
    per_rank = None
    if collective:
        own_gather = (sum(gms) / len(gms)) if gms else ((sum(gather_host_ms) / len(gather_host_ms)) if gather_host_ms else 0.0)
        per_rank = per_rank_table(torch, dist, world, red_dev, dict(rank=rank, device=local_rank, rows=len(mine["index"]), rays=total_rays_local,
                                                                    kernel_ms=kernel_ms / max(n_launch, 1), step_wall_ms=sum(step_walls) / max(len(step_walls), 1),
                                                                    gather_ms=own_gather))

    if rank == 0 and os.environ.get("AMBER_BENCH_SAVE_IMAGE") and img is not None:
        torch.cuda.synchronize()
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
                       "parallelism": f"stripes{world}x8rows", "launches_per_step": len(launches), "engine": "work-queue megakernel, two-phase closest hit" if args.engine == "auto" else args.engine,
                       "math": {amber_amd.MATH_GLIBC: "glibc 2.35 sincosf/powf kernels (the reference's libm)", amber_amd.MATH_PORTABLE: "portable (measurement build)"}[amber_amd.math_mode()]},
            "ranks": world, "collective_backend": ("rccl" if backend == "nccl" else backend),
            "roofline": {"bound": "valu-issue", "contract": "hbm-equivalent (SURVEY 8(d): rays x 96 B of algorithmic ray state / kernel time, against the 8 TB/s HBM peak)",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "kernel": "pt_megakernel" if args.engine != "wavefront" else "wf_generate + wf_bounce launches of one batch", "kernel_ms": round(kern_ms, 3),
                         "note": "achieved = rays per launch (rank 0) x 96 B / mean launch duration (hipEvents on the render stream), frac = achieved / 8000: the contractual "
                                 "figure.  The kernel keeps ray state in VGPRs and never touches that roof: its measured HBM traffic is `traffic` bytes per launch; what binds "
                                 "it is VALU instruction issue -- see `valu` (from the committed rocprofv3 profile, printed only while that profile's kernel time matches this run's)"},
        }
        if collective:
            out["gather_ms"] = round(gather_ms, 4)                     # per step: the gather + re-ordering into global row order, on the render stream (max over ranks)
        if backend == "nccl":
            out["rccl_ranks"] = world
        if per_rank is not None:
            # rays: over the timed steps; kernel_ms: mean per launch; step_wall_ms: the rank's own step until its stream has drained (the gather included:
            # on rank 0 that waits for every rank's rows); gather_ms: the collective + re-ordering on the render stream (host time in the gloo rehearsal)
            out["per_rank"] = per_rank
        if world > 1 and not args.no_parity and img is not None:
            # the checker's leg of an N > 1 run: 16 rows of the GATHERED image (row sums of all samples) against the oracle, on rank 0, outside the timed region
            try:
                sys.path.insert(0, str(ROOT / "tests"))
                from parity_rows import compare_image_rows
                if not args.rehearse_on_one_gpu:
                    torch.cuda.synchronize()
                out["parity"] = compare_image_rows(img.detach().cpu().numpy().reshape(H, W, 3), W, args.spp, args.seed, threads=max(1, host_cores()[1]),
                                                   launches=launches)
            except Exception as e:                                     # the headline must survive its checker
                out["parity"] = None
                out["parity_error"] = str(e)[:200]
        profs = sorted((ROOT / "profiles").glob("r*_hbm_traffic.json"))      # latest committed rocprofv3 PMC summary
        if profs and args.engine == "auto" and args.spp == 1024 and args.width == 1024:
            try:
                out["roofline"]["traffic"] = json.loads(profs[-1].read_text()).get("hbm_bytes_per_launch")
                out["roofline"]["traffic_source"] = f"profiles/{profs[-1].name}"
                if out["roofline"]["traffic"] and kern_ms > 0:
                    out["roofline"]["measured_hbm_gbs"] = round(out["roofline"]["traffic"] / (kern_ms * 1e-3) / 1e9, 1)
            except Exception:
                pass
        summ = latest_profile_summary()
        if summ and args.engine == "auto":
            prof_ms = summ[1].get("kernel_trace", {}).get("average_ns", 0.0) * 1e-6
            # the counters describe the kernel of THAT profile: a kernel that has changed since must not inherit them
            if world == 1 and args.spp == 1024 and args.width == 1024 and prof_ms > 0 and abs(prof_ms - kern_ms) <= 0.05 * kern_ms:
                out["roofline"]["valu"] = dict(summ[1]["valu"], source=f"profiles/{summ[0]}", profile_kernel_ms=round(prof_ms, 3))
            else:
                out["roofline"]["valu"] = None
                out["roofline"]["valu_withheld"] = (f"profiles/{summ[0]} measured {prof_ms:.3f} ms per launch, this run {kern_ms:.3f} ms (or another workload / rank count): "
                                                    "more than 5 % apart, the profile is stale for this kernel -- re-run tools/profile.sh")
        if world == 1:
            try:
                bw = measured_copy_bandwidth(torch, f"cuda:{local_rank}")
                out["roofline"]["measured_copy_bw"] = round(bw, 1)
                out["roofline"]["frac_of_measured_copy_bw"] = round(achieved / bw, 5)
            except Exception as e:                                     # never lose the bench line over the side measurement
                out["roofline"]["measured_copy_bw"] = None
                out["roofline"]["measured_copy_bw_error"] = str(e)[:120]
        if world == 1 and not args.no_secondary:
            # The headline is measured: make it durable before anything else can go wrong, then run the side measurements
            # in a child process -- a GPU fault or a hang there costs only `secondary` / `projected_scaling` / `cold_start`.
            try:
                side_file = ROOT / "gpurun_out" / "bench_headline.json"
                side_file.parent.mkdir(exist_ok=True)
                side_file.write_text(json.dumps(out) + "\n")
            except Exception:
                pass
            del fb, img, gather                                        # aliases of the handle's framebuffer: gone before the handle is
            tracer.close()                                             # the child gets the whole GPU
            torch.cuda.empty_cache()
            side = run_side_child(args)
            out["secondary"] = side.get("secondary")
            out["projected_scaling"] = side.get("projected_scaling")
            cold = side.get("cold_start")
            out["cold_start"] = cold
            out["config"]["cold_wall_ms"] = cold.get("cold_wall_ms") if cold else None    # next to wall_s_per_step: create -> first downloaded frame, new handle
            for k, v in side.items():
                if k.endswith("_error"):
                    out[k] = v
        if not args.no_cpu_baseline and world == 1:                    # reported at N = 1 only
            out["cpu_baseline"], out["parity"] = cpu_baseline(amber_amd, W, args.spp, args.seed)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
