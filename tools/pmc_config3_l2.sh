#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r03z/l2; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for pass in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TCC_READ_REQ_sum"; do
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
            d = agg[r["Counter_Name"]]; k = r.get("Dispatch_Id", "0"); d[k] = d.get(k, 0.0) + float(r["Counter_Value"])
for k in sorted(agg): x = agg[k]; print("%-30s %.6g" % (k, x[max(x, key=lambda s: int(s))]))
PY
