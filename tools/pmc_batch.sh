#!/bin/bash
# SQ counters of the batched image stages (tools/batch_image_kernels.py): the largest launch of every kernel.
set -e
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/pmc_batch"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
    --output-format csv -d "$OUT" -o b -- python3 "$ROOT/tools/batch_image_kernels.py" > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
best = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][-40:]
    key = (k, r["Dispatch_Id"])
    best[key][r["Counter_Name"]] = float(r["Counter_Value"])
top = {}
for (k, d), c in best.items():
    if k not in top or c.get("SQ_WAVES", 0) > top[k].get("SQ_WAVES", 0):
        top[k] = c
for k, c in top.items():
    print(k, {n: "%.3g" % v for n, v in c.items()})
PY
