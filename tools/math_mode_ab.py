"""Cost and distance of the two sin/cos/pow arithmetics.

  python tools/math_mode_ab.py            # on the GPU box (builds the portable variant if it is missing)
1. kernel time of config 2 (512 spp) for the product build (glibc kernels) and the -DAMBER_BUILD_PORTABLE_MATH build, interleaved
   in one process (tools/ab_lib.py);
2. for each build, its distance from oracle(XorShift, BVH, live libm) on 16 full-width rows of config 2 at all 1024 spp:
   cast delta, diverged paths, inexact paths, differing pixels, pixels over the 1e-4 tolerance (tests/parity_rows.py).
"""
import json, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, sys
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import amber_amd, oracle_binding as O
from parity_rows import compare_rows
r = compare_rows(amber_amd, 1024, 1024, 12345, threads=16, math=O.MATH_LIBM, accel=O.ACCEL_BVH)
r["math_mode"] = {1: "portable", 2: "glibc"}[amber_amd.math_mode()]
print(json.dumps(r))
''' % (R, R)
if __name__ == "__main__":
    if not os.path.exists(os.path.join(R, "amber_amd", "lib", "libamber_hip_portable.so")):   # measurement builds are built on demand
        subprocess.run(["make", "-C", os.path.join(R, "amber_amd", "csrc"), "portable"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run([sys.executable, os.path.join(R, "tools", "ab_lib.py"), "libamber_hip.so", "libamber_hip_portable.so", "512"], check=True)
    for lib in ("libamber_hip.so", "libamber_hip_portable.so"):
        env = dict(os.environ, AMBER_AMD_LIB=lib)
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, check=True, capture_output=True, text=True).stdout
        print(lib, out.strip().splitlines()[-1])
