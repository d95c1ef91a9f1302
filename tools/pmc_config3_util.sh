#!/bin/bash
# Lane utilisation / wait counters of the config-3 kernel (engine BVH).  Usage: tools/pmc_config3_util.sh <tag> [env...]
set -u
TAG=${1:-x}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof3u_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -k 10 150 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- python3 $ROOT/tools/bench_config3.py 64 > "$OUT/pmc_${name}.log" 2>&1 || echo "pmc $pass failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "bvh" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
v = {k: x[-1] for k, x in agg.items()}
for k in sorted(v): print("%-28s %.6g" % (k, v[k]))
if "SQ_THREAD_CYCLES_VALU" in v: print("lane utilisation %.3f" % (v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64)))
if "SQ_WAIT_ANY" in v and "SQ_WAVE_CYCLES" in v: print("wait_any %.3f wait_inst %.3f" % (v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"]))
PY
