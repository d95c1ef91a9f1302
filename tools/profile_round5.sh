#!/bin/bash
# Round 5's committed profiles of the final build (GPU box): the bench workload (kernel trace + PMC passes -> tools/summarize_profile.py r05), config 3 and
# the mesh workloads (kernel trace + stats).  Usage: bash tools/profile_round5.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
bash $ROOT/tools/profile.sh r05 > $ROOT/gpurun_out/profile_r05.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_r05_config3 -- python3 $ROOT/tools/bench_config3.py 256 > $ROOT/gpurun_out/prof_r05_config3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_r05_mesh -- python3 $ROOT/tools/mesh_workloads.py > $ROOT/gpurun_out/prof_r05_mesh.log 2>&1
tail -3 $ROOT/gpurun_out/prof_r05_config3.log $ROOT/gpurun_out/prof_r05_mesh.log
