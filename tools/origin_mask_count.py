"""Would per-origin candidate masks (origin object x direction octant -> objects reachable, a conservative box test)
replace Phase A for secondary rays?  Counts, on the Cornell box, how many candidates such a mask leaves for the secondary rays of real paths
(CPU, oracle traces).  Phase A + Phase B today: 1.3 exact tests per ray, 1.25 Phase-B trips per wave iteration."""
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, oracle_binding as O
sc = O.Scene.cornell(O.ACCEL_LIST)
objs = sc.objects()
n = len(objs)
# geometric boxes
def box(o):
    kind, mat, p, nrm = o
    if kind == 0:
        v = p[:9].reshape(3,3); return v.min(0), v.max(0)
    c, r = p[:3], p[3]; return c - r, c + r
B = [box(o) for o in objs]
tol = 1e-4
# mask[origin][octant]: object X reachable iff for every axis: (s=+ : max_X >= min_O - tol) (s=- : min_X <= max_O + tol)
masks = np.zeros((n, 8), np.uint32)
for i in range(n):
    for oc in range(8):
        m = 0
        for j in range(n):
            ok = True
            for a in range(3):
                pos = (oc >> a) & 1
                if pos: ok &= B[j][1][a] >= B[i][0][a] - tol
                else: ok &= B[j][0][a] <= B[i][1][a] + tol
            if ok: m |= 1 << j
        masks[i, oc] = m
pc = np.array([[bin(int(masks[i, oc])).count("1") for oc in range(8)] for i in range(n)])
print("objects", n, "mean popcount over all (origin, octant):", pc.mean())
# the workload's secondary rays
W = H = 256; seed = 12345
rng = np.random.default_rng(1)
cnt = []; miss_in_mask = 0; tot = 0
for _ in range(6000):
    px, py, s = int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(0, 1024))
    k, rec, eye = sc.trace(W, H, seed, px, py, s, math=O.MATH_LIBM, max_bounces=32)
    k = min(k, 32)
    for b in range(1, k):
        o0 = rec[b-1].object
        if o0 < 0: break
        p0 = np.array(rec[b-1].pos[:]);
        if rec[b].object < 0:
            continue     # direction unknown for a miss (not recorded): skip
        p1 = np.array(rec[b].pos[:]); d = p1 - p0
        oc = (1 if d[0] >= 0 else 0) | (2 if d[1] >= 0 else 0) | (4 if d[2] >= 0 else 0)
        m = int(masks[o0, oc]); cnt.append(bin(m).count("1")); tot += 1
        if not (m >> rec[b].object) & 1: miss_in_mask += 1
cnt = np.array(cnt)
print("secondary rays sampled:", tot, " hit object missing from its mask:", miss_in_mask)
print("candidates per secondary ray from (origin object x octant) masks: mean %.2f, median %d, p90 %d, max %d" % (cnt.mean(), np.median(cnt), np.percentile(cnt, 90), cnt.max()))
wave_max = [rng.choice(cnt, 40).max() for _ in range(2000)]
print("largest count among 40 rays drawn at random (a wave's Phase-B trip count): mean %.1f" % np.mean(wave_max))
