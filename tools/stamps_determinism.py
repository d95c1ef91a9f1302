"""Ray counts and image hashes of every engine, several times over, with the library named by AMBER_AMD_LIB: a diagnostic build
must cast exactly the product's rays, every time (EXPERIMENTS.md: the diagnostic build that lost 5 % of its rays).
    AMBER_AMD_LIB=libamber_hip_<variant>.so python tools/stamps_determinism.py [rounds] [spp]"""
import hashlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import amber_amd as A
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
sc = A.HostScene.cornell_box(); sn = A.Sensor.default(256, 256)
print(os.path.basename(str(A.library_path())))
ref = None
for e, name in ((A.ENGINE_LIST, "list"), (A.ENGINE_TWO_PHASE, "two_phase"), (A.ENGINE_BVH, "bvh")):
    seen, diff = [], []
    for k in range(rounds):
        pt = A.PathTracer(sc, sn, engine=e); pt.render_pass(0, spp); img, rays = pt.download(); pt.close()
        seen.append((rays, hashlib.sha1(img.tobytes()).hexdigest()[:10]))
        if ref is None: ref = img.copy()
        diff.append(int((img.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum()))
    print("  %-10s %s%s   pixels differing from list: %s of %d lit" % (name, " ".join("%d/%s" % s for s in sorted(set(seen))), "   <-- NOT deterministic" if len(set(seen)) > 1 else "",
                                                                     sorted(set(diff)), int((ref.sum(axis=2) > 0).sum())))
