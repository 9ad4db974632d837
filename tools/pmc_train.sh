#!/bin/bash
# PMC pass over the UNet training step (batch 8): MFMA busy, waits and LDS activity per kernel.
# usage: tools/pmc_train.sh <tag>     (run on the GPU box from the repository root)
TAG=$1; R=$(pwd); OUT=$R/gpurun_out/pmc_train_$TAG
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
run() { timeout -s KILL 200 rocprofv3 --output-format csv --pmc $2 -d "$OUT/$1" -o p -- python3 $R/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/$1.log" 2>&1 || { echo "pass $1 failed"; tail -n 5 "$OUT/$1.log"; exit 1; }; }
run sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
run sq2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in ("sq", "sq2"):
    f = glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True)
    if not f:
        print(d, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "unetmi::" not in k:
            continue
        k = k[k.index("unetmi::") + 8:]
        k = k[:k.index("(")] if "(" in k else k
        k = k + " grid " + r["Grid_Size"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    for k in sorted(acc):
        print(d, "|", k, "|", cnt[k][next(iter(cnt[k]))], "|", " ".join("{}={:.4g}".format(c, acc[k][c] / cnt[k][c]) for c in sorted(acc[k])))
PY
