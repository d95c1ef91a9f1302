"""BASELINE configs 4 and 5 on the oracle's terms: rank 0's share of the real frame (its interleaved 8-row stripes) for ONE launch of the
real samples-per-launch (1024) -- what bench.py's `secondary` block times -- against oracle(XorShift, List, live libm): every pixel of
the share, image bits and ray count.  About half a minute of oracle time per configuration on 16 threads.
    python tools/full_share_parity.py [4|5] [spp]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import amber_amd as A
import oracle_binding as O
from amber_amd.distributed import stripe_partition
which = [int(a) for a in sys.argv[1:] if a in ("4", "5")] or [4, 5]
spp = next((int(a) for a in sys.argv[1:] if a.isdigit() and int(a) > 5), 1024)
seed = 12345
hs = A.HostScene.cornell_box(); osc = O.Scene.cornell(O.ACCEL_LIST)
threads = min(16, os.cpu_count() or 1)
for cfg in which:
    W, H, world, depth = ((2048, 2048, 4, 0), (3840, 2160, 8, 16))[cfg - 4]
    part = stripe_partition(H, world)[0]
    pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=seed, max_depth=depth, rows=part["rows"], stripe=part["stripe"])
    pt.render_pass(0, spp); img, rays = pt.download()
    rows = np.asarray(part["index"])
    full = np.zeros((H, W, 3), np.float32); casts = 0
    t = time.time()
    starts = rows[::8]
    for y0 in starts:
        y1 = min(int(y0) + 8, H)
        _, c = osc.render_xorshift(W, H, seed, 0, spp, math=O.MATH_LIBM, threads=threads, rows=(int(y0), y1), out=full, max_depth=depth)
        casts += c.casts
    dt = time.time() - t
    ref = full[rows]
    diff = (img.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
    print("config %d: %dx%d, rank 0 of %d (%d rows in %d stripes), %d spp, max depth %d: GPU %d rays; oracle(List, live libm) %d rays in %.1f s; "
          "pixels differing %d of %d; ray count delta %d; lit pixels %d" % (cfg, W, H, world, len(rows), len(starts), spp, depth, rays, casts, dt,
          int(diff.sum()), diff.size, int(rays) - int(casts), int((ref > 0).any(axis=2).sum())), flush=True)
    pt.close()
