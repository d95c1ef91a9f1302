#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of the bench workload.
# Usage: tools/profile.sh <tag>   -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary ${BENCH_ARGS:-}"
echo "== kernel trace + stats"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace_stdout.log" 2>&1 || echo "trace failed"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  echo "== pmc $pass"
  rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- $BENCH > "$OUT/pmc_${name}_stdout.log" 2>&1 || echo "pmc $pass failed"
done
find "$OUT" -name "*.csv" | head -40
