#!/bin/bash
# Round 5's fuzz campaign on the final lab build (every engine, both BVH schedulers, light tracing, the oracle on every 8th small scene; large soups with
# all-triangle leaves and needles; legs 1, 2 and the soups also render through engine REFERENCE_BVH and compare with oracle(ACCEL_BVH)).  Scenes of 33 .. 128 objects also run the two-phase engine over groups of 32.  Each leg is time-limited; a MISMATCH ends its leg with exit code 1.  Usage (GPU box): bash tools/fuzz_round5.sh [seconds per leg] [seed base in millions]
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
T=${1:-400}
B=${2:-7}            # seed base, millions
timeout -k 10 $T python tools/fuzz_engines.py 100000 ${B}000000 --oracle --lt --reference-bvh > gpurun_out/fuzz_r05_1.txt 2>&1; echo "leg 1 exit $?"; tail -1 gpurun_out/fuzz_r05_1.txt
timeout -k 10 $T python tools/fuzz_engines.py 100000 ${B}100000 --extreme --scaled --oracle --reference-bvh > gpurun_out/fuzz_r05_2.txt 2>&1; echo "leg 2 exit $?"; tail -1 gpurun_out/fuzz_r05_2.txt
timeout -k 10 $T python tools/fuzz_engines.py 100000 ${B}200000 --scaled --lt --stripes > gpurun_out/fuzz_r05_3.txt 2>&1; echo "leg 3 exit $?"; tail -1 gpurun_out/fuzz_r05_3.txt
timeout -k 10 120 python tools/fuzz_big.py 4 80000 --reference-bvh > gpurun_out/fuzz_r05_big1.txt 2>&1; echo "big 1 exit $?"; tail -1 gpurun_out/fuzz_r05_big1.txt
timeout -k 10 120 python tools/fuzz_big.py 8 80000 0.03 0.9 --reference-bvh > gpurun_out/fuzz_r05_big2.txt 2>&1; echo "big 2 exit $?"; tail -1 gpurun_out/fuzz_r05_big2.txt
