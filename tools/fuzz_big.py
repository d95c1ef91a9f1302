"""Large random soups (tens of thousands of triangles, spheres, disks, cylinders): engine BVH (parallel build, deep tree)
against engine LIST, bit for bit.   python tools/fuzz_big.py [n_scenes] [n_objects] [sphere share] [triangle share]
(a sphere share near 1 makes most leaves all-sphere leaves: the two-stage sphere leaf; a triangle share near 1 all-triangle leaves: the
two-stage triangle leaf -- a fifth of those triangles are NEEDLES, one edge 1e-3 .. 1e-5 of the others, and a tenth lie in a common plane)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import amber_amd as A
_nums = [a for a in sys.argv[1:] if not a.startswith('--')]
n_scenes = int(_nums[0]) if len(_nums) > 0 else 4
n_obj = int(_nums[1]) if len(_nums) > 1 else 80000
p_sphere = float(_nums[2]) if len(_nums) > 2 else 1.0 / 6.0
p_tri = float(_nums[3]) if len(_nums) > 3 else None
W, H, spp = 64, 48, 4
for seed in range(n_scenes):
    rng = np.random.default_rng(1000 + seed)
    q = (1.0 - p_sphere) / 5.0
    if p_tri is None: kinds = rng.choice([0, 1, 2, 3], n_obj, p=[3 * q, p_sphere, q, q]).astype(np.uint32)
    else: kinds = rng.choice([0, 1, 2, 3], n_obj, p=[p_tri, (1 - p_tri) / 3, (1 - p_tri) / 3, (1 - p_tri) / 3]).astype(np.uint32)
    params = np.zeros((n_obj, 12), np.float32)
    c = rng.uniform(-1, 1, (n_obj, 3)) * rng.choice([1.0, 1.0, 30.0], (n_obj, 1))          # a third of the objects far out: deep, unbalanced tree
    size = (10.0 ** rng.uniform(-3, -0.5, n_obj))
    tri = kinds == 0
    params[:, 0:3] = c
    params[tri, 3:6] = (c + rng.normal(size=(n_obj, 3)) * size[:, None])[tri]
    params[tri, 6:9] = (c + rng.normal(size=(n_obj, 3)) * size[:, None])[tri]
    if p_tri is not None:
        needle = tri & (rng.random(n_obj) < 0.2)                                                  # second vertex a hair from the first
        params[needle, 3:6] = (c + rng.normal(size=(n_obj, 3)) * (size * 10.0 ** rng.uniform(-5, -3, n_obj))[:, None])[needle]
        flat = tri & (rng.random(n_obj) < 0.1)                                                    # coplanar clutter in the plane y = 0.25: exact ties and grazing rays
        for k in (1, 4, 7): params[flat, k] = 0.25
    params[kinds == 1, 3] = size[kinds == 1]
    nrm = rng.normal(size=(n_obj, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    dc = kinds >= 2
    params[dc, 3:6] = nrm[dc]; params[dc, 6] = size[dc]; params[kinds == 3, 7] = (size * 3)[kinds == 3]
    mats = [(4, (20.0, 20.0, 20.0), 0.0), (0, (0.7, 0.7, 0.7), 0.0), (2, (0.9, 0.9, 0.9), 0.0), (3, (1.0, 1.0, 1.0), 1.5), (1, (0.8, 0.8, 0.8), 16.0)]
    material = rng.choice(5, n_obj, p=[0.05, 0.55, 0.15, 0.15, 0.10]).astype(np.uint32)
    hs = A.HostScene.create_arrays(kinds=kinds, material_index=material, params=params, materials=mats,
                                   transform=[1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 4, 0, 0, 0, 1], focal_length=0.05, focus_distance=4.0, radius=0.02, n_blades=6)
    res = {}
    for e in (A.ENGINE_BVH, A.ENGINE_LIST):
        t = time.time(); pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=seed, engine=e); tc = time.time() - t
        t = time.time(); pt.render_pass(0, spp); img, rays = pt.download(); tr = time.time() - t; pt.close()
        res[e] = (img.view(np.uint32).copy(), rays)
        print("seed %d engine %d: create %.2f s, render %.2f s, rays %d" % (seed, e, tc, tr, rays), flush=True)
    if "--reference-bvh" in sys.argv:              # engine REFERENCE_BVH against oracle(ACCEL_BVH): the reference's tree of the same soup on both sides
        sys.path.insert(0, os.path.join(R, "tests"))
        import oracle_binding as O
        t = time.time(); pt = A.PathTracer(hs, A.Sensor.default(W, H), seed=seed, engine=A.ENGINE_REFERENCE_BVH); tc = time.time() - t
        pt.render_pass(0, spp); rimg, rrays = pt.download(); pt.close()
        osc = O.Scene.create_arrays(kinds=kinds, material_index=material, params=params, materials=mats, transform=[1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 4, 0, 0, 0, 1],
                                    focal_length=0.05, focus_distance=4.0, radius=0.02, n_blades=6, accel=O.ACCEL_BVH)
        oimg, cnt = osc.render_xorshift(W, H, seed, 0, spp)
        print("seed %d engine REFERENCE_BVH: create %.2f s, rays %d (oracle(BVH) %d), tree %s, differs from List on %d values" % (seed, tc, rrays, cnt.casts, osc.bvh_stats(), int((rimg.view(np.uint32) != res[A.ENGINE_LIST][0]).sum())), flush=True)
        if rrays != cnt.casts or not np.array_equal(rimg.view(np.uint32), oimg.view(np.uint32)):
            print("REFERENCE-BVH MISMATCH seed", seed); sys.exit(1)
    if res[A.ENGINE_BVH][1] != res[A.ENGINE_LIST][1] or not np.array_equal(res[A.ENGINE_BVH][0], res[A.ENGINE_LIST][0]):
        print("MISMATCH seed", seed); sys.exit(1)
print("big fuzz ok: %d scenes of %d objects" % (n_scenes, n_obj))
