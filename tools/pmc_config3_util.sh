#!/bin/bash
# Lane utilisation / wait / memory-pipe counters of the config-3 kernel (engine BVH).  Usage: tools/pmc_config3_util.sh <tag> [ENV=VALUE ...]
# (AMBER_BVH_POOL=1 profiles pt_bvh_pool_kernel instead of the default pt_bvh_megakernel)
set -u
TAG=${1:-x}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof3u_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ROOT/tools/bench_config3.py 256 > "$OUT/trace.log" 2>&1 || echo "kernel trace failed"
grep -h "bvh" $OUT/trace/*/*_kernel_stats.csv | cut -c1-220
tail -1 $OUT/trace.log
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_TOTAL_WAVEFRONTS_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -k 10 150 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- python3 $ROOT/tools/bench_config3.py 64 > "$OUT/pmc_${name}.log" 2>&1 || echo "pmc $pass failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(dict)
for f in glob.glob(out + "/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "bvh" in r["Kernel_Name"]:
            d = agg[r["Counter_Name"]]; k = r.get("Dispatch_Id", "0")
            d[k] = d.get(k, 0.0) + float(r["Counter_Value"])
v = {k: x[max(x, key=lambda s: int(s))] for k, x in agg.items()}     # the last (timed) launch
for k in sorted(v): print("%-36s %.6g" % (k, v[k]))
if "SQ_THREAD_CYCLES_VALU" in v: print("lane utilisation %.3f" % (v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64)))
if "SQ_WAIT_ANY" in v and "SQ_WAVE_CYCLES" in v: print("wait_any %.3f wait_inst %.3f" % (v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"]))
PY
