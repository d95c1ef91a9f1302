#!/bin/bash
# Runs on the GPU box: instruction-delivery counters of the bench workload (branches, instruction fetches, SALU cycles).
# Usage: tools/pmc_branches.sh <tag>  -> gpurun_out/prof_<tag>_branches/
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_branches
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc" -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$OUT/stdout.log" 2>&1 || echo "pmc failed"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pt_megakernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print("%-24s %.6e per launch (%d launches)" % (k, tot[k] / max(n[k], 1), n[k]))
PY
