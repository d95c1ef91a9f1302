#!/bin/bash
# Runs on the GPU box: PMC passes (one small counter group per run, each under its own timeout: a group the hardware
# cannot collect makes rocprofv3 abort and hang) over tools/bench_config3.py (engine BVH, 1M spheres).  Usage: tools/profile_config3.sh <tag> [spp]
set -u
TAG=${1:-r01}; SPP=${2:-64}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof3_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/bench_config3.py $SPP"
for pass in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  echo "== pmc $pass"
  timeout -k 10 150 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- $CMD > "$OUT/pmc_${name}_stdout.log" 2>&1 || echo "pmc $pass failed"
  tail -1 "$OUT/pmc_${name}_stdout.log"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "megakernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print("%-34s last launch %.6g  (launches %d)" % (k, v[-1], len(v)))
PY
