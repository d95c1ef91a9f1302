set -u
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/prof3_r02x; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|name)?\s*:?\s*(TA_|TCP_|TD_)" | head -80 > $OUT/counters_list.txt
rocprofv3 -L > $OUT/counters_full.txt 2>&1
for pass in "TA_TA_BUSY_sum TA_BUSY_avr" "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE TA_TOTAL_WAVEFRONTS_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -k 10 150 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- python3 $ROOT/tools/bench_config3.py 64 > "$OUT/pmc_${name}.log" 2>&1 || echo "pmc $pass failed"
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
    print("%-40s last launch %.6g  (launches %d)" % (k, v[-1], len(v)))
PY
