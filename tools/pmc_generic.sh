#!/bin/bash
# Per-dispatch PMC counters of one kernel of a python script.  Usage: tools/pmc_generic.sh <tag> <kernel substring> <script> [args]   (env passes through)
set -u
TAG=$1; KERNEL=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmcg_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
# bench.py starts a child process for its side measurements, which would inherit the profiler and run under counter collection: not when profiled
case "$1" in *bench.py) case " $* " in *" --no-secondary "*) ;; *) set -- "$@" --no-secondary ;; esac ;; esac
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_TOTAL_WAVEFRONTS_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- python3 "$@" > "$OUT/pmc_${name}.log" 2>&1 || echo "pmc $pass failed"
done
for f in $OUT/pmc_*.log; do tail -n 1 $f; done | sort | uniq -c
python3 - "$OUT" "$KERNEL" <<'PY'
import csv, glob, sys, collections
out, kern = sys.argv[1], sys.argv[2]
import re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
names = {}
for f in glob.glob(out + "/pmc_*/*/*_counter_collection.csv"):
    rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    for r in rows:
        i = ids.index(int(r["Dispatch_Id"]))
        agg[r["Counter_Name"]][i] += float(r["Counter_Value"])
        names[i] = re.sub(r"^.*?(pt_\w+?kernel)(<[^>]*>)?.*$", r"\1\2", r["Kernel_Name"])[:28]     # (the same script dispatches the same kernels in every pass)
n = max(len(v) for v in agg.values())
print("%-40s" % "dispatch of" + " ".join("%28s" % names.get(i, "?") for i in range(n)))
for k in sorted(agg): print("%-40s" % k + " ".join("%28.6g" % agg[k].get(i, float("nan")) for i in range(n)))
d = {k: agg[k] for k in agg}
def col(name, i): return d.get(name, {}).get(i, float("nan"))
print("%-40s" % "lane utilisation (VALU)" + " ".join("%28.3f" % (col("SQ_THREAD_CYCLES_VALU", i) / (col("SQ_ACTIVE_INST_VALU", i) * 64)) for i in range(n)))
print("%-40s" % "SQ_WAIT_ANY / SQ_WAVE_CYCLES" + " ".join("%28.3f" % (col("SQ_WAIT_ANY", i) / col("SQ_WAVE_CYCLES", i)) for i in range(n)))
print("%-40s" % "SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES" + " ".join("%28.3f" % (col("SQ_WAIT_INST_ANY", i) / col("SQ_WAVE_CYCLES", i)) for i in range(n)))
print("%-40s" % "SIMD cycles per VALU wave-instruction" + " ".join("%28.3f" % (col("SQ_ACTIVE_INST_VALU", i) * 4 / col("SQ_INSTS_VALU", i)) for i in range(n)))
PY
